#!/usr/bin/env python3
"""Study builds (make -C wsi_segmentation_pipeline_amd/csrc clean all STUDY=1): where the waves of conv3x3s1_wide_kernel spend
their cycles on the layer-2/3/4 shapes - s_memtime stamps around the phases, summed over a launch's waves
(csrc/conv.hip: g_wide_stamps).  Usage: python tools/wide_stamps.py [--n 2000] [--planes 3]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_segmentation_pipeline_amd import engine as E, native  # noqa: E402

SHAPES = [(128, 32, 32), (256, 16, 16), (512, 8, 8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=2000)
    ap.add_argument('--planes', type=int, default=3)
    ap.add_argument('--cfg', type=int, default=60)
    ap.add_argument('--shape', type=str, default='', help='one C,H,W shape instead of the layer-2/3/4 ones (64,64,64 with --cfg 38: the layer-1 slab3 kernel)')
    args = ap.parse_args()
    lib = native.load()
    if not hasattr(lib, 'wsi_study_wide_stamps'):
        raise SystemExit('not a study build: make -C wsi_segmentation_pipeline_amd/csrc clean all STUDY=1')
    lib.wsi_study_wide_stamps.restype = C.c_int
    lib.wsi_study_wide_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    dev = torch.device('cuda:0')
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    print('%-26s %8s %9s | %6s %6s %6s %6s %6s %6s' % ('shape', 'ms', 'cyc/wave', 'setup', 'prolog', 'lines', 'taps', 'tail', 'mult'))
    for (c, h, w) in ([tuple(int(v) for v in args.shape.split(','))] if args.shape else SHAPES):
        n = args.n
        x = torch.randn(n, c, h, w, generator=g).abs_()
        wt = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
        wpk, bias = E.prepack_conv(wt, None, args.planes, dev)
        xpf = E.pf_pack(x.to(dev), args.planes)
        rpf = E.pf_pack(torch.randn(n, c, h, w, generator=g).to(dev), args.planes)
        out = E.pf_zeros(n, c, h, w, args.planes, dev)
        for resid in (False, True):
            def run():
                return lib.wsi_conv3x3_bn_act_cfg(xpf.data_ptr(), out.data_ptr(), rpf.data_ptr() if resid else None, wpk.data_ptr(),
                                                  bias.data_ptr(), n, h, w, c, c, 1, 1, args.planes, args.cfg, st())
            native.check(run(), 'conv')
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * 8)()
            lib.wsi_study_wide_stamps(buf, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            lib.wsi_study_wide_stamps(buf, 1)
            waves, total, setup, pro, lines, taps, tail = (float(buf[i]) for i in range(7))
            mult = total - setup - pro - lines - taps - tail
            f = lambda v: '%5.1f%%' % (100.0 * v / total)
            print('%-26s %8.3f %9.0f | %s %s %s %s %s %s' % ('C=%d %dx%d%s' % (c, h, w, ' +resid' if resid else ''), e0.elapsed_time(e1),
                                                             total / waves, f(setup), f(pro), f(lines), f(taps), f(tail), f(mult)))


if __name__ == '__main__':
    main()
