"""CPU restatement (PyTorch fp32, functional) of the reference bag-of-patches ResNet-18.

Test infrastructure only (see oracle/__init__.py).  Pinned against the reference itself by
tests/golden/resnet18_*.npz (oracle/gen_golden.py imports /root/reference/resnets_shift.py).

Follows, op for op and in the same order:
  * transform           /root/reference/utils/preprocessing.py:206-212  (ToTensor + Normalize)
  * stem                /root/reference/resnets_shift.py:196-199
  * BasicBlock.forward  /root/reference/resnets_shift.py:49-65
  * ResNet.forward      /root/reference/resnets_shift.py:189-217
  * Classifier/Regressor /root/reference/models/models.py:20-58
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5                      # nn.BatchNorm2d default, resnets_shift.py:117
DATASET_MEAN = (0.485, 0.456, 0.406)   # /root/reference/myargs.py:127
DATASET_STD = (0.229, 0.224, 0.225)    # /root/reference/myargs.py:129


def normalize_u8(u8_nchw, mean=DATASET_MEAN, std=DATASET_STD):
    """ToTensor (u8 -> f32, /255) then Normalize ((x-mean)/std), all in fp32 like torchvision."""
    x = torch.as_tensor(np.ascontiguousarray(u8_nchw)).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1)
    return x.sub(m).div(s)


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'], False, 0.0, BN_EPS)


def _block(sd, prefix, x, stride):
    out = F.conv2d(x, sd[prefix + '.conv1.weight'], None, stride, 1)
    out = F.relu(_bn(sd, prefix + '.bn1', out))
    out = F.conv2d(out, sd[prefix + '.conv2.weight'], None, 1, 1)
    out = _bn(sd, prefix + '.bn2', out)
    if (prefix + '.downsample.0.weight') in sd:
        x = _bn(sd, prefix + '.downsample.1', F.conv2d(x, sd[prefix + '.downsample.0.weight'], None, stride, 0))
    return F.relu(out + x)


def trunk(sd, x, taps=None):
    """x: (N,3,H,W) normalised fp32 -> (N,512,H/32,W/32).  ``taps`` (dict) collects intermediates."""
    x = F.conv2d(x, sd['conv1.weight'], None, 2, 3)
    x = F.relu(_bn(sd, 'bn1', x))
    if taps is not None:
        taps['stem'] = x
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps['pool'] = x
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = _block(sd, 'layer%d.0' % li, x, stride)
        if taps is not None:
            taps['layer%d.0' % li] = x
        x = _block(sd, 'layer%d.1' % li, x, 1)
        if taps is not None:
            taps['layer%d.1' % li] = x
    return x


def pooled_features(sd, x):
    """(N,3,H,W) -> (N,512): trunk + AdaptiveAvgPool2d((1,1)) + flatten."""
    return torch.flatten(F.adaptive_avg_pool2d(trunk(sd, x), 1), 1)


def resnet_forward(sd, xs):
    """Bag forward: xs (B,P,3,H,W) fp32 -> (singles (P*B,4) patch-major, ensemble (B,4))."""
    B, P = xs.shape[:2]
    xs = xs.transpose(0, 1)
    feats, singles = [], []
    for p in range(P):
        f = pooled_features(sd, xs[p])
        singles.append(F.linear(f, sd['fc0.weight'], sd['fc0.bias']))
        feats.append(f)
    features = torch.cat(feats, 1)
    h = F.relu(F.linear(features.view(B, -1), sd['fc.0.weight'], sd['fc.0.bias']))
    return torch.cat(singles, 0), F.linear(h, sd['fc.2.weight'], sd['fc.2.bias'])


def classifier(sd, fmap):
    """Classifier: avgpool(1,1) -> flatten -> Linear.  fmap (N,F,h,w)."""
    f = torch.flatten(F.adaptive_avg_pool2d(fmap, 1), 1)
    return F.linear(f, sd['fc.0.weight'], sd['fc.0.bias'])


def regressor(sd, fmap):
    """Regressor: avgpool -> Linear(F,F/4) -> ReLU -> Linear(F/4,C)."""
    f = torch.flatten(F.adaptive_avg_pool2d(fmap, 1), 1)
    return F.linear(F.relu(F.linear(f, sd['fc.0.weight'], sd['fc.0.bias'])), sd['fc.2.weight'], sd['fc.2.bias'])


def tile_logits(sd, cls_sd, u8_nchw):
    """The `predict_tumorbed(mode='cls')` per-batch compute: normalise -> encoder -> classifier
    (/root/reference/utils/eval.py:196-198)."""
    return classifier(cls_sd, trunk(sd, normalize_u8(u8_nchw)))
