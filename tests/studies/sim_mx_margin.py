#!/usr/bin/env python3
"""CPU simulation of the mx cross-term formats (fp4 e2m1 / fp6 e2m3) on the five reference-generated precision-margin
families (tests/golden/margin_*.npz): predicts max |logit - reference| before a kernel is written (r03 design study)."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import resnet_oracle as R
from oracle import weights as W
import sim_mx_numerics as S

CASES = ['wide_a_uniform', 'wide_a_he', 'wide_b_he_hot', 'default_he', 'default_uniform_hot']


def load_case(case):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'margin_%s.npz' % case))
    if str(g['weight_family']) == 'default':
        sd = W.make_resnet18_state_dict(int(g['weight_seed']), with_fc=False)
    else:
        stats = {k[4:].replace('__', '.'): g[k] for k in g.files if k.startswith('bn__')}
        sd = W.make_wide_resnet18_state_dict(int(g['weight_seed']), stats)
    head = W.make_head_state_dict(int(g['head_seed']), 'classifier')
    gain = float(g['head_gain'])
    head = {k: v * gain for k, v in head.items()}
    u8 = W.make_he_patches(int(g['input_seed']), 8) if str(g['input_kind']) == 'he' else W.make_u8_patches(int(g['input_seed']), (8, 3, 256, 256))
    return sd, head, u8, g['logits']


def main():
    torch.set_num_threads(8)
    for case in CASES:
        sd, head, u8, ref = load_case(case)
        sd = dict(sd)
        sd['fc0.weight'], sd['fc0.bias'] = head['fc.0.weight'], head['fc.0.bias']
        x = R.normalize_u8(u8)
        row = []
        with torch.no_grad():
            for cross in ('fp4', 'fp6'):
                out = S.forward(sd, x, 'fp16', cross, stored=True).numpy()
                row.append(float(np.abs(out - ref).max()))
        print('%-22s |logit| %.1f   fp4 %.2e   fp6 %.2e' % (case, float(np.abs(ref).max()), *row), flush=True)


if __name__ == '__main__':
    main()
