#!/usr/bin/env python3
"""One-off stress of the r05 U-Net routes: random batch sizes and patch shapes, fused tail + x0-from-stem against the three-launch /
separate-stem-conv routes (wsi_conv_set_mode +2097152 +8388608) on the same engine; tiles partly outside the slide."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_segmentation_pipeline_amd import native, synthetic as W
from wsi_segmentation_pipeline_amd.engine import PARITY
from wsi_segmentation_pipeline_amd.unet import UNetEngine
dev = torch.device('cuda:0')
lib = native.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
usd = W.make_unet_state_dict(5, classes=4)
for key in ('decoder.final_conv.weight', 'decoder.final_conv.bias'):
    usd[key] = usd[key] * (8.0 / 216.0)
eng = UNetEngine(usd, dev, planes=PARITY)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    h, w = (int(rng.choice([32, 64, 96, 128, 160, 192, 256])) for _ in range(2))
    n = int(rng.integers(1, 12))
    sh, sw = h + int(rng.integers(0, 90)), w + int(rng.integers(0, 90))
    slide = torch.from_numpy(rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)).to(dev)
    xy = torch.from_numpy(np.stack([rng.integers(-20, sw - w + 40, n), rng.integers(-20, sh - h + 40, n)], 1).astype(np.int32))
    a = eng.forward_tiles(slide, xy, h, w)
    lib.wsi_conv_set_mode(1 + 2097152 + 8388608)
    b = eng.forward_tiles(slide, xy, h, w)
    lib.wsi_conv_set_mode(1)
    assert torch.isfinite(a).all() and a.shape == b.shape == (n, 4, h, w)
    d = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
    worst = max(worst, d)
    assert d <= 3e-5, (it, n, h, w, d)
    eng.release_workspaces()
print('ok: worst relative logit difference fused / plain routes %.3g' % worst)
