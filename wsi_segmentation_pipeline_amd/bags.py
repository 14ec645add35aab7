"""Region-proposal sparse inference across ranks (BASELINE.json configs[3]; reference scannet.py:134-155, slic.py:82-99):
bags of 16 crops (64x64 at level 1) per candidate region -> ResNet.forward bag path -> class per region -> painted label
image.  Bags are independent, so they are sharded over the ranks by greedy cost balance, each rank runs its bags on the HIP
trunk + the `fc` bag head, ONE RCCL all-gather of the (R, C) ensemble logits follows, and the paint runs on the device.
"""
import ctypes as C

import numpy as np
import torch

from . import native
from .engine import _ptr, _require_gpu, _stream


# ------------------------------------------------------------------------------------------ sharding
def shard_bags(costs, world):
    """Greedy balance (largest cost first onto the least loaded rank; ties -> lower rank): returns `world` ascending index
    arrays covering range(len(costs)).  Every eval bag costs the same 16 crops, so this reduces to an even split, but
    callers may pass real costs (bags with fewer valid crops, regions of very different paint size)."""
    costs = np.asarray(costs, np.float64)
    order = np.argsort(-costs, kind='stable')
    load = np.zeros(world)
    owner = np.empty(len(costs), np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += costs[i]
    return [np.nonzero(owner == r)[0] for r in range(world)]


def gather_rows(local, shards, rank, world):
    """local (n_r, C) rows of this rank's shard -> (R, C) rows in original order on every rank: one all-gather of equal
    padded chunks (payload R x C fp32: latency-bound, a single collective), then an index scatter.  The collective goes through
    slide._Wire like every other one of the path: one op list for RCCL and gloo, only the wire buffer's place differs."""
    total = sum(len(s) for s in shards)
    from .slide import _Wire, _collective_forced
    if world == 1 and not _collective_forced():
        out = torch.empty((total, local.shape[1]), dtype=local.dtype, device=local.device)
        out[torch.as_tensor(shards[0], device=local.device)] = local
        return out
    chunk = max(len(s) for s in shards)
    wire = _Wire(local.device)
    buf = wire.new((chunk, local.shape[1]), local.dtype, 0)
    buf[:local.shape[0]] = wire.put(local)
    allb = wire.get(wire.all_gather(wire.new((world * chunk, local.shape[1]), local.dtype), buf))
    out = torch.empty((total, local.shape[1]), dtype=local.dtype, device=local.device)
    for r in range(world):
        if len(shards[r]):
            out[torch.as_tensor(shards[r], device=local.device)] = allb[r * chunk:r * chunk + len(shards[r])]
    return out


# ------------------------------------------------------------------------------------------ paint
def prepare_paint(label_shape, index_lists, device):
    """Upload the regions' pixel index lists once: (flat int64 indices, region number per entry) device tensors.
    index_lists: per region a tuple of index arrays (as np.nonzero returns) or a flat index array."""
    flat, region_of = [], []
    for r, idx in enumerate(index_lists):
        f = np.ravel_multi_index(tuple(np.asarray(a) for a in idx), label_shape) if isinstance(idx, tuple) else np.asarray(idx).ravel()
        flat.append(f.astype(np.int64))
        region_of.append(np.full(len(f), r, np.int32))
    flat = np.concatenate(flat) if flat else np.zeros(0, np.int64)
    region_of = np.concatenate(region_of) if region_of else np.zeros(0, np.int32)
    dev = torch.device(device)
    return torch.from_numpy(flat).to(dev), torch.from_numpy(region_of).to(dev)


def paint_regions(label_shape, index_lists, classes, device, prepared=None):
    """pred_mask[foreground_indices] = cls for every region, in order (scannet.py:154-155), on the device: where regions
    overlap the last one wins, as in the reference's loop.  classes (R,) uint8 GPU tensor.  Returns the int64 label image."""
    lib = native.load()
    npix = int(np.prod(label_shape))
    dev = torch.device(device)
    idx_d, reg_d = prepared if prepared is not None else prepare_paint(label_shape, index_lists, dev)
    classes = classes.to(dev, torch.uint8).contiguous()
    label = torch.zeros(tuple(label_shape), dtype=torch.int64, device=dev)
    if idx_d.numel() == 0:
        return label
    winner = torch.empty(npix, dtype=torch.int32, device=dev)
    native.check(lib.wsi_paint_regions(_ptr(idx_d), _ptr(reg_d), idx_d.numel(), _ptr(classes), _ptr(winner), _ptr(label), npix, _stream()),
                 'wsi_paint_regions')
    return label


# ------------------------------------------------------------------------------------------ bag head
def bag_ensemble(eng, feat, fc0_w, fc0_b, fc2_w, fc2_b, bag=16):
    """(n*bag, 512) pooled features -> (n, C) ensemble logits: Linear(8192 -> 4096) + ReLU on the fp32 MFMA GEMM, then
    Linear(4096 -> C) (reference resnets_shift.py:133-139,214-215; features concatenated patch-major per bag)."""
    n = feat.shape[0] // bag
    hidden = eng.linear(feat.view(n, bag * feat.shape[1]), fc0_w, fc0_b, relu=True)
    return eng.linear(hidden, fc2_w, fc2_b)


class BagWorkload:
    """bench.py --workload cfg4: R synthetic regions (seeded), 16 crop corners each on an HBM-resident synthetic level-1
    image, lognormal region areas (paint sizes).  One step = every bag of this rank through trunk + heads, gather, softmax /
    argmax, paint."""

    def __init__(self, eng, sd, regions, seed, device, rank=0, world=1, size=8192, label_hw=(512, 512)):
        from . import slide as S
        self.eng, self.rank, self.world, self.device = eng, rank, world, torch.device(device)
        rng = np.random.default_rng(seed)
        self.level1 = S.SyntheticRows(size, size, seed, device).full()
        self.R = int(regions)
        self.total_crops = self.R * 16
        xy = rng.integers(0, size - 64, (self.R, 16, 2)).astype(np.int32)
        self.shards = shard_bags(np.full(self.R, 16.0), world)
        mine = self.shards[rank]
        self.xy = torch.from_numpy(np.ascontiguousarray(xy[mine].reshape(-1, 2))).to(self.device)
        self.fc = [sd[k].to(self.device, torch.float32).contiguous() for k in ('fc.0.weight', 'fc.0.bias', 'fc.2.weight', 'fc.2.bias')]
        # irregular paint sizes: lognormal areas, random pixel sets (disjointness is not required: last region wins)
        self.label_hw = label_hw
        areas = np.minimum(np.maximum(rng.lognormal(4.0, 1.0, self.R).astype(np.int64), 4), 4096)
        self.index_lists = [rng.integers(0, label_hw[0] * label_hw[1], int(a)) for a in areas]
        self.prepared = prepare_paint(label_hw, self.index_lists, self.device)
        self.n_mine = len(mine)

    def step(self):
        from .engine import softmax_threshold_argmax
        if self.n_mine:
            feat, singles, _ = self.eng.forward_tiles(self.level1, self.xy, 64, 64, feat=True, logits=True)
            ens = bag_ensemble(self.eng, feat, *self.fc)
        else:
            ens = torch.zeros((0, 4), dtype=torch.float32, device=self.device)
        ens = gather_rows(ens, self.shards, self.rank, self.world)
        as_map = ens.t().to(torch.float64).contiguous().view(ens.shape[1], -1, 1)
        cls = softmax_threshold_argmax(as_map, (0., 0., 0., 0.), want_probs=False)[0].view(-1)
        label = paint_regions(self.label_hw, None, cls, self.device, self.prepared)
        return {'logits': ens, 'classes': cls, 'label': label}

    def roofline(self, per_kind):
        """The dominant kernels of the bag path are the same conv kernels on 16x16 ... 2x2 maps; report the stride-1 3x3 convs of
        layers 2-4 like the slide workloads do."""
        k = per_kind.get('conv3x3_s1')
        if not k:
            return None
        return {'kernel': 'stride-1 3x3 convs of layers 2-4 on 8x8 / 4x4 / 2x2 maps (64x64 crops)', 'bound': 'mfma',
                'achieved': round(k['tflops'], 2), 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': round(k['tflops'] / 2500.0, 4),
                'traffic': None, 'avg_launch_ms': round(k['avg_ms'], 4)}
