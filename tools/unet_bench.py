#!/usr/bin/env python3
"""Throughput of the dense 'seg' model (ResNet-18 encoder + smp-style U-Net decoder, reference eval_tumorbed.py default mode):
tiles of 256x256 read from an HBM-resident u8 slide -> (N, classes, 256, 256) logits.  Run on the GPU box:
  python tools/unet_bench.py [--planes 2|3] [--n 512] [--batch 128]      (under rocprofv3 --kernel-trace --stats for the shares)"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_segmentation_pipeline_amd import synthetic as W  # noqa: E402
from wsi_segmentation_pipeline_amd.unet import UNetEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--planes', type=int, default=2)
    ap.add_argument('--n', type=int, default=512)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    sd = W.make_unet_state_dict(5, classes=4)
    eng = UNetEngine(sd, dev, planes=a.planes, max_batch=a.batch)
    side = int(np.ceil(np.sqrt(a.n)))
    g = torch.Generator(device=dev).manual_seed(1)
    slide = torch.randint(0, 256, (side * 256, side * 256, 3), dtype=torch.uint8, device=dev, generator=g)
    xy = torch.tensor([[256 * (i % side), 256 * (i // side)] for i in range(a.n)], dtype=torch.int32, device=dev)
    eng.forward_tiles(slide, xy[:a.batch], 256, 256)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        t0 = time.perf_counter()
        out = eng.forward_tiles(slide, xy, 256, 256)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    gf = 3.63 + 6.04                                          # encoder trunk + decoder GFLOP per 256x256 tile (real channels)
    print(json.dumps({'planes': a.planes, 'tiles': a.n, 'batch': a.batch, 'tiles_per_s': round(a.n / best, 1),
                      'ms_per_tile': round(best / a.n * 1e3, 4), 'algorithmic_TFLOPs': round(gf * a.n / best / 1e3, 1),
                      'logits_shape': list(out.shape)}))


if __name__ == '__main__':
    main()
