#!/usr/bin/env python3
"""Study build only (make -C wsi_segmentation_pipeline_amd/csrc STUDY=1): phase anatomy of the ping-pong conv kernel.
Runs cfg 75 / 76 (cfg 70 / 71 with s_memtime stamps) on the layer shapes and prints, per wave of two workgroups, the
summed cycles of {load phase, wait at its barrier, multiply phase, wait at its barrier} and the per-step averages."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_segmentation_pipeline_amd import engine as E, native  # noqa: E402


def main():
    lib = C.CDLL(native.LIB_PATH)
    native.load()
    dev = torch.device('cuda:0')
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    for (c, h, w, cfg) in ((128, 32, 32, 76), (256, 16, 16, 75), (512, 8, 8, 75)):
        x = torch.randn(n, c, h, w, generator=g).abs_()
        wt = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
        wpk, bias = E.prepack_conv(wt, None, 3, dev)
        xpf = E.pf_pack(x.to(dev), 3)
        rpf = E.pf_pack(torch.randn(n, c, h, w, generator=g).to(dev), 3)
        out = E.pf_zeros(n, c, h, w, 3, dev)
        dbg = torch.zeros(96, dtype=torch.int64, device=dev)
        lib.wsi_study_set_debug(C.c_void_p(dbg.data_ptr()))
        for _ in range(3):
            rc = native.load().wsi_conv3x3_bn_act_cfg(xpf.data_ptr(), out.data_ptr(), rpf.data_ptr(), wpk.data_ptr(), bias.data_ptr(),
                                                      n, h, w, c, c, 1, 1, 3, cfg, st())
            assert rc == 0, rc
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(2, 8, 6)
        steps = (c // 32) * 9
        print('shape C=%d %dx%d cfg %d: %d steps; cycles per step and wave [load (reads + wait), barrier, multiply, barrier, DMA issue]' % (c, h, w, cfg, steps))
        for wg in range(2):
            for wv in range(8):
                print('  wg %d wave %d: %s  sum %.0f' % (wg, wv, np.round(d[wg, wv, :5] / steps).astype(int).tolist(), d[wg, wv].sum() / steps))


if __name__ == '__main__':
    main()
