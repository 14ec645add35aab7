#!/usr/bin/env python3
"""CPU simulation: which layers of the mx trunk need their MX-fp6 cross-term pass?  (r05 design study)
Per (tile, line, tap) step the mx mode issues 2 fp16 MFMAs (Wh.Xh) + 1 MX-fp6 MFMA (Wh6.Xl6 + Wl6.Xh6): the third instruction is 1/3 of
the matrix time of layers 2-4.  Errors of late layers pass through fewer layers and are averaged by the pool, so the cross terms may be
droppable there.  Variants = set of stages whose stride-1 / stride-2 3x3 convs run WITHOUT the cross-term pass (single fp16 pass);
activations stay stored as (fp16 hi, fp6 lo) lines everywhere.  Prints max |logit - reference| per margin family."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import resnet_oracle as R
import sim_mx_numerics as S
import sim_mx_margin as M


def forward(sd, x, single):
    """single: set of conv names ('layer4.0.conv1', ...) or prefixes that run as ONE fp16 pass."""
    def is_single(name):
        return any(name.startswith(p) for p in single)

    def bnfold(wk, bnk):
        s = sd[bnk + '.weight'].double() / torch.sqrt(sd[bnk + '.running_var'].double() + 1e-5)
        return (sd[wk].double() * s.view(-1, 1, 1, 1)).float(), (sd[bnk + '.bias'].double() - sd[bnk + '.running_mean'].double() * s).float()

    def cv(x, name, wk, bnk, stride, pad):
        w, b = bnfold(wk, bnk)
        return S.conv_scheme(x, w, stride, pad, 'fp16', None if is_single(name) else 'fp6') + b.view(1, -1, 1, 1)
    sto = lambda t: S.store(t, 'fp6', 'fp16')
    w, b = bnfold('conv1.weight', 'bn1')
    x = F.relu(F.conv2d(x.double(), w.double(), None, 2, 3).float() + b.view(1, -1, 1, 1))      # the product's stem is exact integer arithmetic
    x = sto(F.max_pool2d(x, 3, 2, 1))
    for li, st in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in range(2):
            p = 'layer%d.%d' % (li, bi)
            s = st if bi == 0 else 1
            y = sto(F.relu(cv(x, p + '.conv1', p + '.conv1.weight', p + '.bn1', s, 1)))
            y = cv(y, p + '.conv2', p + '.conv2.weight', p + '.bn2', 1, 1)
            if (p + '.downsample.0.weight') in sd:
                x = cv(x, p + '.conv2', p + '.downsample.0.weight', p + '.downsample.1', s, 0)   # folded into conv2's launch in the product
            x = sto(F.relu(y + x))
    f = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    return F.linear(f, sd['fc0.weight'], sd['fc0.bias'])


VARIANTS = [('all mx', ()), ('layer4 single', ('layer4',)), ('layer4.1 single', ('layer4.1',)), ('layers 3-4 single', ('layer3', 'layer4')),
            ('layer4 + layer3.1 single', ('layer4', 'layer3.1')), ('conv2 of layer4 blocks single', ('layer4.0.conv2', 'layer4.1.conv2')),
            ('everything single', ('layer',))]


def main():
    torch.set_num_threads(8)
    cases = sys.argv[1:] or M.CASES
    for case in cases:
        sd, head, u8, ref = M.load_case(case)
        sd = dict(sd)
        sd['fc0.weight'], sd['fc0.bias'] = head['fc.0.weight'], head['fc.0.bias']
        x = R.normalize_u8(u8)
        with torch.no_grad():
            row = ['%s %.2e' % (name, float(np.abs(forward(sd, x, single).numpy() - ref).max())) for name, single in VARIANTS]
        print('%-22s |logit| %.1f : %s' % (case, float(np.abs(ref).max()), ' | '.join(row)), flush=True)


if __name__ == '__main__':
    main()
