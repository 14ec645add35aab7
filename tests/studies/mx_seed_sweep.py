#!/usr/bin/env python3
"""Study (GPU): max |logit - fp32 CPU oracle| of the mx and parity modes over several seeded checkpoints and inputs
(256x256 patches, fc0 head): how much margin is there to the 1e-3 contract beyond the golden seed?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import resnet_oracle as R  # noqa: E402
from oracle import weights as W  # noqa: E402
from wsi_segmentation_pipeline_amd.engine import TrunkEngine  # noqa: E402

dev = torch.device('cuda:0')
for seed in (11, 12, 13, 14, 15, 16):
    sd = W.make_resnet18_state_dict(seed)
    u8 = W.make_u8_patches(100 + seed, (8, 3, 256, 256))
    x = R.normalize_u8(u8)
    with torch.no_grad():
        f = R.trunk(sd, x)
        feat = torch.flatten(torch.nn.functional.adaptive_avg_pool2d(f, 1), 1)
        ref = torch.nn.functional.linear(feat, sd['fc0.weight'], sd['fc0.bias'])
    out = []
    for planes in (3, 2):
        eng = TrunkEngine(sd, dev, planes=planes, head=(sd['fc0.weight'], sd['fc0.bias']))
        got = eng.forward_f32(x.to(dev), logits=True)[1].cpu()
        out.append(float((got - ref).abs().max()))
    print('seed %d: |logit| max %.2f  err mx %.2e  parity %.2e' % (seed, float(ref.abs().max()), out[0], out[1]))
