"""Deterministic synthetic checkpoints and inputs (used by bench.py, smoke() and the tests).

No trained checkpoint exists offline (every pretrained loader is a network
fetch), so the parity fixtures use seeded random weights with *randomised BatchNorm statistics*
(identity-like default BN hides folding bugs).  The same generator feeds the reference model in
``oracle/gen_golden.py`` (via ``load_state_dict``) and the HIP path on the GPU box, so only seeds and
outputs need to be committed, not 179 MB of weights.

Key names / shapes follow the reference ``resnets_shift.ResNet`` state dict
(/root/reference/resnets_shift.py:122-150, 169-187): 130 keys.
"""
import numpy as np
import torch

_STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))   # (planes, stride of first block); 2 BasicBlocks each


def resnet18_key_shapes(n_bag=16, include_aux_heads=True):
    """[(key, shape, kind)] in the reference's state-dict order."""
    out = []

    def bn(prefix, c):
        out.append((prefix + '.weight', (c,), 'bn_w'))
        out.append((prefix + '.bias', (c,), 'bn_b'))
        out.append((prefix + '.running_mean', (c,), 'bn_m'))
        out.append((prefix + '.running_var', (c,), 'bn_v'))
        out.append((prefix + '.num_batches_tracked', (), 'bn_n'))

    out.append(('conv1.weight', (64, 3, 7, 7), 'conv'))
    bn('bn1', 64)
    inpl = 64
    for li, (planes, stride) in enumerate(_STAGES, start=1):
        for bi in range(2):
            p = 'layer%d.%d' % (li, bi)
            out.append((p + '.conv1.weight', (planes, inpl if bi == 0 else planes, 3, 3), 'conv'))
            bn(p + '.bn1', planes)
            out.append((p + '.conv2.weight', (planes, planes, 3, 3), 'conv'))
            bn(p + '.bn2', planes)
            if bi == 0 and (stride != 1 or inpl != planes):
                out.append((p + '.downsample.0.weight', (planes, inpl, 1, 1), 'conv'))
                bn(p + '.downsample.1', planes)
        inpl = planes
    n = 512 * n_bag
    out.append(('fc.0.weight', (n // 2, n), 'lin_w'))
    out.append(('fc.0.bias', (n // 2,), 'lin_b'))
    out.append(('fc.2.weight', (4, n // 2), 'lin_w'))
    out.append(('fc.2.bias', (4,), 'lin_b'))
    out.append(('fc0.weight', (4, 512), 'lin_w'))
    out.append(('fc0.bias', (4,), 'lin_b'))
    if include_aux_heads:
        out.append(('fc1.0.weight', (16, 512), 'lin_w'))
        out.append(('fc1.0.bias', (16,), 'lin_b'))
        out.append(('fc2.0.weight', (4, 16 * n_bag), 'lin_w'))
        out.append(('fc2.0.bias', (4,), 'lin_b'))
    return out


def _fill(rng, shape, kind):
    if kind == 'conv':          # kaiming-normal, fan_out, relu gain (resnets_shift.py:154)
        fan_out = shape[0] * shape[2] * shape[3]
        return rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_out))
    if kind == 'bn_w':
        return rng.uniform(0.75, 1.25, shape).astype(np.float32)
    if kind == 'bn_b' or kind == 'bn_m':
        return (rng.standard_normal(shape) * 0.1).astype(np.float32)
    if kind == 'bn_v':
        return rng.uniform(0.75, 1.25, shape).astype(np.float32)
    if kind == 'bn_n':
        return np.int64(1)
    if kind == 'lin_w':
        bound = 1.0 / np.sqrt(shape[1])
        return rng.uniform(-bound, bound, shape).astype(np.float32)
    if kind == 'lin_b':
        return rng.uniform(-0.05, 0.05, shape).astype(np.float32)
    raise ValueError(kind)


def make_resnet18_state_dict(seed, with_fc=True):
    """Seeded state dict; ``with_fc=False`` skips the 33.5M-parameter ``fc.0`` draw (keys absent)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, shape, kind in resnet18_key_shapes():
        if not with_fc and key.startswith('fc.'):
            continue
        sd[key] = torch.from_numpy(np.asarray(_fill(rng, shape, kind)))
    return sd


def make_head_state_dict(seed, kind, num_features=512, num_classes=4):
    """``Classifier`` (fc.0) or ``Regressor`` (fc.0, fc.2) weights (/root/reference/models/models.py:20-58)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    if kind == 'classifier':
        sd['fc.0.weight'] = torch.from_numpy(_fill(rng, (num_classes, num_features), 'lin_w'))
        sd['fc.0.bias'] = torch.from_numpy(_fill(rng, (num_classes,), 'lin_b'))
    elif kind == 'regressor':
        sd['fc.0.weight'] = torch.from_numpy(_fill(rng, (num_features // 4, num_features), 'lin_w'))
        sd['fc.0.bias'] = torch.from_numpy(_fill(rng, (num_features // 4,), 'lin_b'))
        sd['fc.2.weight'] = torch.from_numpy(_fill(rng, (num_classes, num_features // 4), 'lin_w'))
        sd['fc.2.bias'] = torch.from_numpy(_fill(rng, (num_classes,), 'lin_b'))
    else:
        raise ValueError(kind)
    return sd


def make_u8_patches(seed, shape):
    """i.i.d. uniform u8 patches, e.g. shape (B, P, 3, H, W) planar or (N, H, W, 3) interleaved."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=shape, dtype=np.uint8)


# ---------------------------------------------------------------------------------------------- wider checkpoint families
def make_wide_resnet18_state_dict(seed, bn_stats=None):
    """A "trained-like" checkpoint family far from the default one (tests of the precision-mode margin): every conv output
    channel is scaled by a random factor (x 0.1 ... x 3.2, so BatchNorm's running variance spans ~[0.01, 10] of its usual
    size), BN gamma is uniform in [0.25, 3] and beta ~ N(0, 0.5^2).  The running statistics must then MATCH the data (as in
    any trained network: otherwise activations explode through the eight blocks): they are calibrated once with the
    reference model itself (oracle/gen_golden.py) and travel inside the golden fixture; pass them as
    bn_stats = {'<bn prefix>.running_mean': array, '<bn prefix>.running_var': array}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, shape, kind in resnet18_key_shapes():
        if key.startswith('fc.'):
            continue
        v = np.asarray(_fill(rng, shape, kind))
        if kind == 'conv':
            v = v * np.exp(rng.uniform(np.log(0.1), np.log(3.2), (shape[0], 1, 1, 1))).astype(np.float32)
        elif kind == 'bn_w':
            v = rng.uniform(0.25, 3.0, shape).astype(np.float32)
        elif kind == 'bn_b':
            v = (rng.standard_normal(shape) * 0.5).astype(np.float32)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v))
    if bn_stats is not None:
        for k, v in bn_stats.items():
            assert k in sd and tuple(sd[k].shape) == tuple(np.asarray(v).shape), k
            sd[k] = torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32)))
    return sd


def make_he_patches(seed, n, size=256):
    """H&E-like u8 tiles (N,3,H,W): saturated white background (255), pink stroma, purple nuclei blobs, mild noise - the
    input family a real slide gives (large flat 255 areas, strongly correlated channels), unlike i.i.d. uniform noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[:size, :size].astype(np.float32)
    out = np.empty((n, 3, size, size), np.uint8)
    for i in range(n):
        img = np.full((size, size, 3), 255.0, np.float32)
        tissue = np.zeros((size, size), bool)
        for _ in range(int(rng.integers(1, 4))):                       # stroma regions
            cy, cx, ry, rx = rng.uniform(0, size), rng.uniform(0, size), rng.uniform(40, 160), rng.uniform(40, 160)
            tissue |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
        img[tissue] = np.array([232, 160, 200], np.float32) + rng.normal(0, 12, (int(tissue.sum()), 3))
        for _ in range(int(rng.integers(20, 120))):                    # nuclei
            cy, cx, r = rng.uniform(0, size), rng.uniform(0, size), rng.uniform(2, 7)
            m = ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r) & tissue
            img[m] = np.array([90, 50, 140], np.float32) + rng.normal(0, 10, (int(m.sum()), 3))
        out[i] = np.clip(np.rint(img), 0, 255).astype(np.uint8).transpose(2, 0, 1)
    return out


def make_unet_state_dict(seed, classes=4):
    """Seeded state dict of the U-Net 'seg' model (encoder.* = the ResNet-18 trunk keys of make_resnet18_state_dict(seed),
    decoder.* per wsi_segmentation_pipeline_amd.unet.decoder_key_shapes), randomised BN statistics as everywhere."""
    from .unet import decoder_key_shapes
    sd = {'encoder.' + k: v for k, v in make_resnet18_state_dict(seed, with_fc=False).items()
          if not k.startswith('fc')}
    rng = np.random.Generator(np.random.PCG64(seed + 5000))
    for key, shape, kind in decoder_key_shapes(classes):
        sd[key] = torch.from_numpy(np.asarray(_fill(rng, shape, kind)))
    return sd
