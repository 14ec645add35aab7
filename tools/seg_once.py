#!/usr/bin/env python3
"""One U-Net engine and a few forward passes over HBM-resident tiles: the command behind the seg kernel traces / counter passes.
usage: seg_once.py [--mode parity|mx|speed] [--n 128] [--reps 2] [--set-mode FLAGS]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_segmentation_pipeline_amd import native, synthetic as W
from wsi_segmentation_pipeline_amd.engine import PARITY, MX, SPEED
from wsi_segmentation_pipeline_amd.unet import UNetEngine

ap = argparse.ArgumentParser()
ap.add_argument('--mode', default='parity')
ap.add_argument('--n', type=int, default=128)
ap.add_argument('--reps', type=int, default=2)
ap.add_argument('--set-mode', type=int, default=0)
a = ap.parse_args()
dev = torch.device('cuda:0')
usd = W.make_unet_state_dict(5, classes=4)
for key in ('decoder.final_conv.weight', 'decoder.final_conv.bias'):
    usd[key] = usd[key] * (8.0 / 216.0)
eng = UNetEngine(usd, dev, planes={'parity': PARITY, 'mx': MX, 'speed': SPEED}[a.mode], max_batch=a.n)
if a.set_mode:
    native.load().wsi_conv_set_mode(1 + a.set_mode)
side = 1
while side * side < a.n:
    side += 1
level = torch.randint(0, 256, (side * 256, side * 256, 3), dtype=torch.uint8, device=dev)
xy = torch.tensor([[256 * (i % side), 256 * (i // side)] for i in range(a.n)], dtype=torch.int32, device=dev)
for _ in range(a.reps):
    out = eng.forward_tiles(level, xy, 256, 256)
torch.cuda.synchronize()
print('ok', tuple(out.shape), float(out.abs().max()))
