#!/usr/bin/env python3
"""Per-launch HIP-event times of one trunk forward (batch of synthetic 256x256 patches), in launch order.
Usage: python tools/launch_times.py [--planes 3] [--n 1000] [--reps 5]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_segmentation_pipeline_amd import native, synthetic as W  # noqa: E402
from wsi_segmentation_pipeline_amd.engine import TrunkEngine  # noqa: E402

NAMES = {1: 'conv3x3_s1', 5: 'conv3x3_s1', 2: 'conv3x3_s2(+ds)', 3: 'conv1x1_s2', 4: 'stem+maxpool'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--planes', type=int, default=3)
    ap.add_argument('--n', type=int, default=1000)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--tile', type=int, default=256, help='patch size (multiple of 32)')
    ap.add_argument('--slide-tiles-per-row', type=int, default=0, help='tiles per slide row (default: square slide); 156 = the 40k-wide cfg3 slide, 1 = tiles stacked in one column (pitch 768 B)')
    ap.add_argument('--stem-rows', type=int, default=0, help='pooled rows per stem workgroup (default: library default)')
    ap.add_argument('--s2', type=int, default=-1, help='wsi_conv_set_mode value (0 gather, 1 slab with 64-pixel tiles, 3 slab with 128-pixel tiles)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = native.load()
    if args.s2 >= 0:
        native.check(lib.wsi_conv_set_mode(args.s2), 'wsi_conv_set_mode')
    if args.stem_rows:
        native.check(lib.wsi_stem_set_mode(1, args.stem_rows), 'wsi_stem_set_mode')
    sd = W.make_resnet18_state_dict(11, with_fc=False)
    cls = W.make_head_state_dict(22, 'classifier')
    eng = TrunkEngine(sd, dev, planes=args.planes, head=(cls['fc.0.weight'], cls['fc.0.bias']), max_batch=args.n)
    g = torch.Generator(device=dev).manual_seed(3)
    side = args.slide_tiles_per_row if args.slide_tiles_per_row > 0 else int(np.ceil(np.sqrt(args.n)))
    T = args.tile
    rows_of_tiles = -(-args.n // side)
    slide = torch.randint(0, 256, (rows_of_tiles * T, side * T, 3), dtype=torch.uint8, device=dev, generator=g)
    xy = torch.tensor([[T * (i % side), T * (i // side)] for i in range(args.n)], dtype=torch.int32, device=dev)
    eng.forward_tiles(slide, xy, T, T, logits=True)
    torch.cuda.synchronize()
    cap = 64 * args.reps
    native.check(lib.wsi_prof_begin(cap), 'prof')
    for _ in range(args.reps):
        eng.forward_tiles(slide, xy, T, T, logits=True)
    torch.cuda.synchronize()
    ms = np.zeros(cap, np.float32); kind = np.zeros(cap, np.int32); fl = np.zeros(cap, np.float64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = lib.wsi_prof_end(p(ms), p(kind), p(fl), cap)
    per = n // args.reps
    m = ms[:n].reshape(args.reps, per)
    med = np.median(m, 0)
    for i in range(per):
        print('%2d %-16s %7.3f ms  %7.1f TFLOP/s' % (i, NAMES.get(int(kind[i]), '?'), med[i], fl[i] / med[i] / 1e9))
    print('sum of kernels %.3f ms per batch of %d %dx%d patches -> %.0f patches/s (kernel time only)' % (med.sum(), args.n, T, T, args.n / med.sum() * 1e3))


if __name__ == '__main__':
    main()
