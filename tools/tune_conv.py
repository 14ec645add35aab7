#!/usr/bin/env python3
"""On-GPU tuning of the stride-1 3x3 slab kernel: times every tile configuration on the four
ResNet-18 layer shapes (interleaved rounds in one process) and checks all configurations give
bit-identical outputs.  Usage: python tools/tune_conv.py [--n 250] [--planes 2] [--rounds 5]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_segmentation_pipeline_amd import engine as E, native  # noqa: E402

SHAPES = [(64, 64, 64), (128, 32, 32), (256, 16, 16), (512, 8, 8)]      # (C, H, W) for 256x256 patches
NCFG = 92


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=250)
    ap.add_argument('--planes', type=int, default=2)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--cfgs', type=str, default='')
    ap.add_argument('--ablate', type=str, default='1', help='comma list of relu/ablation bit masks')
    ap.add_argument('--noresid', action='store_true')
    ap.add_argument('--zero', action='store_true', help='all-zero activations (clock / power test)')
    ap.add_argument('--wcopies', type=int, default=1, help='study: replicate the packed weights N times (<= 16), workgroups spread over the copies')
    ap.add_argument('--scale', type=int, default=1, help='divide H,W by this (64x64 patches: 4)')
    ap.add_argument('--shapes', type=str, default='', help='comma list of indices into SHAPES')
    ap.add_argument('--set-mode', type=int, default=-1, help='wsi_conv_set_mode value (A/B switches: include/wsi_hip.h)')
    ap.add_argument('--shape', type=str, default='', help='one explicit C,H,W shape (e.g. 64,256,256: the U-Net decoder\'s last level)')
    args = ap.parse_args()
    lib = native.load()
    if args.set_mode >= 0:
        native.check(lib.wsi_conv_set_mode(args.set_mode), 'wsi_conv_set_mode')
    dev = torch.device('cuda:0')
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    shapes = [tuple(int(v) for v in args.shape.split(','))] if args.shape else ([SHAPES[int(i)] for i in args.shapes.split(',')] if args.shapes else SHAPES)
    for (c, h, w) in shapes:
        h, w = h // args.scale, w // args.scale
        n = args.n
        x = torch.randn(n, c, h, w, generator=g).abs_()
        if args.zero:
            x.zero_()
        wt = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
        wpk, bias = E.prepack_conv(wt, None, args.planes, dev)
        if args.wcopies > 1:
            wpk = wpk.repeat(args.wcopies).contiguous()
        xpf = E.pf_pack(x.to(dev), args.planes)
        rpf = E.pf_pack(torch.randn(n, c, h, w, generator=g).to(dev), args.planes)
        outs = {cfg: E.pf_zeros(n, c, h, w, args.planes, dev) for cfg in range(NCFG)}
        flops = 2.0 * n * h * w * c * c * 9

        flag = [1]
        wc = (args.wcopies - 1) << 10

        def run(cfg):
            return lib.wsi_conv3x3_bn_act_cfg(xpf.data_ptr(), outs[cfg].data_ptr(), None if args.noresid else rpf.data_ptr(),
                                              wpk.data_ptr(), bias.data_ptr(), n, h, w, c, c, 1, flag[0] | wc, args.planes, cfg, st())
        want = [int(v) for v in args.cfgs.split(',')] if args.cfgs else list(range(NCFG))
        valid = [cfg for cfg in want if run(cfg) == 0]
        torch.cuda.synchronize()
        if not valid:
            continue
        ref = outs[valid[0]]
        same = {cfg: bool(torch.equal(outs[cfg], ref)) for cfg in valid}            # (ablation cfgs 50-53 differ by design)
        masks = [int(v) for v in args.ablate.split(',')]
        valid = [(cfg, mk) for cfg in valid for mk in masks]
        times = {cfg: [] for cfg in valid}
        same = {k: same[k[0]] for k in valid}
        for _ in range(args.rounds):
            for cfg in valid:
                flag[0] = cfg[1]
                cfg_id = cfg[0]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    run(cfg_id)
                e1.record()
                torch.cuda.synchronize()
                times[cfg].append(e0.elapsed_time(e1) / args.iters)
        print('shape C=%d %dx%d n=%d planes=%d  (%.1f GFLOP)' % (c, h, w, n, args.planes, flops / 1e9))
        for cfg in valid:
            med, mn = float(np.median(times[cfg])), float(np.min(times[cfg]))
            print('  cfg %s: median %.3f ms  min %.3f ms  -> %.1f TFLOP/s algorithmic  identical=%s' %
                  (cfg, med, mn, flops / med / 1e9, same[cfg]))


if __name__ == '__main__':
    main()
