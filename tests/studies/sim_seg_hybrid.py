#!/usr/bin/env python3
"""CPU simulation (r05 design study): dense 'seg' model with the ENCODER in mx (fp16 + MX-fp6 cross terms, activations stored as
fp16 hi + fp6 lo lines) and the DECODER as an fp16 pair - does the hybrid hold the absolute 1e-3 contract at |logit| ~ 16?
(mx everywhere: 2.2e-3 measured; fp16 pair everywhere: 1.2e-4 measured.)  The encoder is 40 % of the path's arithmetic and runs 1.5x
faster in mx."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import resnet_oracle as R
from oracle import unet_oracle as U
from oracle import weights as W
import sim_mx_numerics as S
import sim_seg_precision as P


def forward(sd, x, enc_mode, dec_mode):
    def fold(wk, bnk):
        s = sd[bnk + '.weight'].double() / torch.sqrt(sd[bnk + '.running_var'].double() + R.BN_EPS)
        return (sd[wk].double() * s.view(-1, 1, 1, 1)).float(), (sd[bnk + '.bias'].double() - sd[bnk + '.running_mean'].double() * s).float()

    def conv(x, w, stride, pad, mode):
        if mode == 'mx':
            return S.conv_scheme(x, w, stride, pad, 'fp16', 'fp6')
        return P.conv(x, w, stride, pad, 'fp16', True)

    def sto(t, mode):
        return S.store(t, 'fp6', 'fp16') if mode == 'mx' else P.store(t, 'fp16')

    def cv(x, wk, bnk, stride, pad, mode):
        w, b = fold(wk, bnk)
        return conv(x, w, stride, pad, mode) + b.view(1, -1, 1, 1)
    e = 'encoder.'
    w, b = fold(e + 'conv1.weight', e + 'bn1')
    x0 = sto(F.relu(F.conv2d(x.double(), w.double(), None, 2, 3).float() + b.view(1, -1, 1, 1)), enc_mode)      # exact integer stem
    x = F.max_pool2d(x0, 3, 2, 1)
    skips = [x0]
    for li, st in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in range(2):
            p = e + 'layer%d.%d' % (li, bi)
            s = st if bi == 0 else 1
            y = sto(F.relu(cv(x, p + '.conv1.weight', p + '.bn1', s, 1, enc_mode)), enc_mode)
            y = cv(y, p + '.conv2.weight', p + '.bn2', 1, 1, enc_mode)
            if (p + '.downsample.0.weight') in sd:
                x = cv(x, p + '.downsample.0.weight', p + '.downsample.1', s, 0, enc_mode)
            x = sto(F.relu(y + x), enc_mode)
        skips.append(x)
    enc = skips[::-1]
    x = enc[0]
    sk = list(enc[1:]) + [None]
    for L in range(5):
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        if sk[L] is not None:
            x = torch.cat([x, sk[L]], 1)
        for j in range(2):
            p = 'decoder.layer%d.block.%d.block' % (L + 1, j)
            x = sto(F.relu(cv(x, p + '.0.weight', p + '.1', 1, 1, dec_mode)), dec_mode)
    return F.conv2d(x, sd['decoder.final_conv.weight'], sd['decoder.final_conv.bias'])


def main():
    torch.set_num_threads(8)
    sd = W.make_unet_state_dict(5, classes=4)
    for key in ('decoder.final_conv.weight', 'decoder.final_conv.bias'):
        sd[key] = sd[key] * (8.0 / 216.0)
    rng = np.random.default_rng(3)
    x = R.normalize_u8(rng.integers(0, 256, (2, 3, 256, 256), dtype=np.uint8))
    with torch.no_grad():
        ref = U.unet_forward(sd, x)
        for em, dm in (('pair', 'pair'), ('mx', 'pair'), ('mx', 'mx'), ('pair', 'mx')):
            out = forward(sd, x, em, dm)
            print('encoder %-4s decoder %-4s: max |logit - fp32 oracle| %.2e (max |logit| %.1f)' % (em, dm, float((out - ref).abs().max()), float(ref.abs().max())), flush=True)


if __name__ == '__main__':
    main()
