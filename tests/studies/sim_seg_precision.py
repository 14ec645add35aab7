#!/usr/bin/env python3
"""CPU simulation of two-plane operand splits on the dense 'seg' model (ResNet-18 encoder + smp-style U-Net decoder): which 16-bit
pair holds the ABSOLUTE 1e-3 logit contract at |logit| ~ 16 per pixel (r05 design study; r04 measured 1.011e-3 with bf16 hi + bf16 lo)?
   x ~ hi + lo, conv = hi*hi + hi*lo + lo*hi (three MFMA passes, fp32 accumulate), activations stored as (hi, lo) between layers.
Prints max |logit - fp64 chain| and max |logit - fp32 oracle| for: bf16 pair, fp16 pair, fp16 pair with flushed fp16 subnormals."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import resnet_oracle as R
from oracle import unet_oracle as U
from oracle import weights as W

FTZ = False


def split(t, kind):
    if kind == 'bf16':
        hi = t.bfloat16().float()
        lo = (t - hi).bfloat16().float()
    else:
        tc = t.clamp(-65504, 65504)
        hi = tc.half().float()
        lo = (tc - hi).half().float()
        if FTZ:
            hi = torch.where(hi.abs() < 2.0 ** -14, torch.zeros_like(hi), hi)
            lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
    return hi, lo


def conv(x, w, stride, pad, kind, wscale=False):
    """x: stored activation (already hi + lo exactly); w: folded fp32 weights."""
    if kind is None:
        return F.conv2d(x, w, None, stride, pad)
    xh, xl = split(x, kind)
    if wscale:                                              # per-output-channel power-of-two scaling into [2^9, 2^10)
        amax = w.abs().amax((1, 2, 3), keepdim=True).clamp_min(1e-30)
        s = 2.0 ** (9 - torch.floor(torch.log2(amax)))
        wh, wl = split(w * s, kind)
        out = F.conv2d(xl, wh, None, stride, pad) + F.conv2d(xh, wl, None, stride, pad) + F.conv2d(xh, wh, None, stride, pad)
        return out / s.view(1, -1, 1, 1)
    wh, wl = split(w, kind)
    return F.conv2d(xl, wh, None, stride, pad) + F.conv2d(xh, wl, None, stride, pad) + F.conv2d(xh, wh, None, stride, pad)


def store(x, kind):
    if kind is None:
        return x
    h, l = split(x, kind)
    return h + l


def forward(sd, x, kind, dtype=torch.float32, wscale=False, taps=None):
    def fold(wk, bnk):
        s = sd[bnk + '.weight'].double() / torch.sqrt(sd[bnk + '.running_var'].double() + R.BN_EPS)
        return (sd[wk].double() * s.view(-1, 1, 1, 1)).to(dtype), (sd[bnk + '.bias'].double() - sd[bnk + '.running_mean'].double() * s).to(dtype)

    def cv(x, wk, bnk, stride, pad):
        w, b = fold(wk, bnk)
        return conv(x, w, stride, pad, kind, wscale) + b.view(1, -1, 1, 1)

    sto = lambda t: store(t, kind)
    x = x.to(dtype)
    e = 'encoder.'
    x0 = sto(F.relu(cv(x, e + 'conv1.weight', e + 'bn1', 2, 3)))      # (the product's stem is exact integer arithmetic: no operand split there)
    if kind is not None:
        w, b = fold(e + 'conv1.weight', e + 'bn1')
        x0 = sto(F.relu(F.conv2d(x.double(), w.double(), None, 2, 3).to(dtype) + b.view(1, -1, 1, 1)))
    x = F.max_pool2d(x0, 3, 2, 1)
    skips = [x0]
    for li, st in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in range(2):
            p = e + 'layer%d.%d' % (li, bi)
            s = st if bi == 0 else 1
            y = sto(F.relu(cv(x, p + '.conv1.weight', p + '.bn1', s, 1)))
            y = cv(y, p + '.conv2.weight', p + '.bn2', 1, 1)
            if (p + '.downsample.0.weight') in sd:
                x = cv(x, p + '.downsample.0.weight', p + '.downsample.1', s, 0)     # (folded into conv2's accumulators in the product)
            x = sto(F.relu(y + x))
        skips.append(x)
    enc = skips[::-1]                                         # x4, x3, x2, x1, x0
    x = enc[0]
    sk = list(enc[1:]) + [None]
    for L in range(5):
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        if sk[L] is not None:
            x = torch.cat([x, sk[L]], 1)
        for j in range(2):
            p = 'decoder.layer%d.block.%d.block' % (L + 1, j)
            x = sto(F.relu(cv(x, p + '.0.weight', p + '.1', 1, 1)))
        if taps is not None:
            taps.append(x)
    return F.conv2d(x, sd['decoder.final_conv.weight'].to(dtype), sd['decoder.final_conv.bias'].to(dtype))


def main():
    global FTZ
    torch.set_num_threads(8)
    sd = W.make_unet_state_dict(5, classes=4)
    for key in ('decoder.final_conv.weight', 'decoder.final_conv.bias'):
        sd[key] = sd[key] * (8.0 / 216.0)
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (2, 3, 256, 256), dtype=np.uint8)
    x = R.normalize_u8(u8)
    with torch.no_grad():
        ref64 = forward(sd, x, None, torch.float64)
        ref32 = U.unet_forward(sd, x)
        print('max |logit| %.2f;  fp32 oracle vs fp64 chain: %.2e' % (float(ref64.abs().max()), float((ref32.double() - ref64).abs().max())))
        for kind, ws, ftz in (('bf16', False, False), ('fp16', False, False), ('fp16', True, False), ('fp16', False, True), ('fp16', True, True)):
            FTZ = ftz
            out = forward(sd, x, kind, torch.float32, ws)
            print('%s pair%s%s:  vs fp64 %.2e   vs fp32 oracle %.2e' % (kind, ' + weight scaling' if ws else '', ' (fp16 subnormals flushed)' if ftz else '',
                  float((out.double() - ref64).abs().max()), float((out - ref32).abs().max())), flush=True)


if __name__ == '__main__':
    main()
