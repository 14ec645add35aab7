#!/usr/bin/env python3
"""CPU simulation of candidate reduced-pass numerics for the conv stack (design study, not product code):
   conv(x, w) ~= conv(xh, wh)            main pass, 16-bit operands (fp16 or bf16), fp32 accumulate
              +  conv(q(xl), q(wh))      cross terms with block-scaled low-precision operands
              +  conv(q(xh), q(wl))      (MX: one E8M0 scale per 32 channels along K)
Reports max |logit - fp32 reference| on the cfg1 golden inputs (64 patches 256x256)."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import resnet_oracle as R
from oracle import weights as W

FP4 = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6.0])                       # e2m1 magnitudes
FP6 = torch.tensor(sorted({m * 2.0 ** e for e in range(0, 4) for m in (1, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875)}
                          | {k * 0.125 for k in range(8)}))              # e2m3 magnitudes (max 7.5)


def q_block(t, dim, grid):
    """Block-scaled quantisation along `dim` in blocks of 32 with a power-of-two scale (MX style)."""
    t = t.movedim(dim, -1)
    shp = t.shape
    c = shp[-1]
    pad = (-c) % 32
    if pad:
        t = F.pad(t, (0, pad))
    b = t.reshape(*t.shape[:-1], -1, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    scale = 2.0 ** torch.ceil(torch.log2(amax / grid.max()))
    y = (b / scale).abs()
    idx = torch.bucketize(y, (grid[1:] + grid[:-1]) / 2)
    q = grid[idx] * torch.sign(b) * scale
    q = q.reshape(*t.shape[:-1], -1)[..., :c].reshape(shp)
    return q.movedim(-1, dim)


def conv_scheme(x, w, stride, pad, main, cross):
    cast = (lambda t: t.half().float()) if main == 'fp16' else (lambda t: t.bfloat16().float())
    xh, wh = cast(x), cast(w)
    out = F.conv2d(xh, wh, None, stride, pad)
    if cross is None:
        return out
    xl, wl = x - xh, w - wh
    if cross == 'exact16':
        return out + F.conv2d(cast(xl), wh, None, stride, pad) + F.conv2d(xh, cast(wl), None, stride, pad)
    grid = FP4 if cross == 'fp4' else FP6
    qx = lambda t: q_block(t, 1, grid)
    qw = lambda t: q_block(t, 1, grid)
    return out + F.conv2d(qx(xl), qw(wh), None, stride, pad) + F.conv2d(qx(xh), qw(wl), None, stride, pad)


def store(x, cross, main):
    """What the next layer / the residual sees when activations are stored as (hi, q(lo))."""
    if cross in (None, 'exact16'):
        return x
    cast = (lambda t: t.half().float()) if main == 'fp16' else (lambda t: t.bfloat16().float())
    grid = FP4 if cross == 'fp4' else FP6
    xh = cast(x)
    return xh + q_block(x - xh, 1, grid)


def forward(sd, x, main, cross, stored=False):
    def bnfold(wk, bnk):
        s = sd[bnk + '.weight'].double() / torch.sqrt(sd[bnk + '.running_var'].double() + 1e-5)
        return (sd[wk].double() * s.view(-1, 1, 1, 1)).float(), (sd[bnk + '.bias'].double() - sd[bnk + '.running_mean'].double() * s).float()
    def cv(x, wk, bnk, stride, pad):
        w, b = bnfold(wk, bnk)
        return conv_scheme(x, w, stride, pad, main, cross) + b.view(1, -1, 1, 1)
    sto = (lambda t: store(t, cross, main)) if stored else (lambda t: t)
    x = F.relu(cv(x, 'conv1.weight', 'bn1', 2, 3))
    x = sto(F.max_pool2d(x, 3, 2, 1))
    for li, st in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in range(2):
            p = 'layer%d.%d' % (li, bi)
            s = st if bi == 0 else 1
            y = sto(F.relu(cv(x, p + '.conv1.weight', p + '.bn1', s, 1)))
            y = cv(y, p + '.conv2.weight', p + '.bn2', 1, 1)
            if (p + '.downsample.0.weight') in sd:
                x = sto(cv(x, p + '.downsample.0.weight', p + '.downsample.1', s, 0))
            x = sto(F.relu(y + x))
    f = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    return F.linear(f, sd['fc0.weight'], sd['fc0.bias'])


def main():
    torch.set_num_threads(8)
    sd = W.make_resnet18_state_dict(11, with_fc=False)
    u8 = W.make_u8_patches(1, (4, 16, 3, 256, 256)).reshape(-1, 3, 256, 256)[:32]
    x = R.normalize_u8(u8)
    with torch.no_grad():
        ref = F.linear(R.pooled_features(sd, x), sd['fc0.weight'], sd['fc0.bias'])
        for main_t, cross in (('bf16', None), ('fp16', None), ('bf16', 'exact16'), ('fp16', 'exact16'), ('fp16', 'fp6'), ('fp16', 'fp4'),
                              ('bf16', 'fp6')):
            out = forward(sd, x, main_t, cross)
            print('main %-5s cross %-8s max|dlogit| = %.2e' % (main_t, cross, float((out - ref).abs().max())), end='')
            if cross in ('fp4', 'fp6'):
                out = forward(sd, x, main_t, cross, stored=True)
                print('   with (hi, q(lo)) storage: %.2e' % float((out - ref).abs().max()), end='')
            print()


if __name__ == '__main__':
    main()
