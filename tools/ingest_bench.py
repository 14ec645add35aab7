#!/usr/bin/env python3
"""Host -> HBM ingestion rate of the pinned ring (wsi_ring_*) against a plain pageable `.to(device)` of the same level, and the
PCIe-inclusive patch rate it implies for the cfg2 tiling (256x256 tiles at stride 256: 196608 level bytes per patch).
Run on the GPU box:  python tools/ingest_bench.py [--size 16384] [--channels 4]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wsi_segmentation_pipeline_amd import ingest  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=16384)
    ap.add_argument('--channels', type=int, default=4)
    ap.add_argument('--slots', type=int, default=4)
    ap.add_argument('--slot-mb', type=int, default=64)
    ap.add_argument('--workers', type=int, default=4)
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    h = w = a.size
    src = np.random.default_rng(0).integers(0, 256, (h, w, a.channels), dtype=np.uint8)
    ring = ingest.IngestRing(a.slots, a.slot_mb << 20)
    out = torch.empty((h, w, 3), dtype=torch.uint8, device='cuda')

    def read_band(y0, rows, dst):
        np.copyto(dst, src[y0:y0 + rows])
    res = {}
    for name, fn in (('ring', lambda: ring.upload_level(read_band, h, w, a.channels, 'cuda:0', out=out, workers=a.workers)),
                     ('pageable', lambda: out.copy_(torch.from_numpy(np.ascontiguousarray(src[..., :3])).to('cuda')))):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        res[name] = {'seconds': round(best, 4), 'level_GBps': round(h * w * 3 / best / 1e9, 2), 'host_GBps': round(src.nbytes / best / 1e9, 2),
                     'patches_per_s_256_stride256': round(h * w / 65536 / best, 0)}
    assert np.array_equal(out.cpu().numpy()[::97], src[::97, :, :3])
    print(json.dumps({'size': a.size, 'channels': a.channels, 'slots': a.slots, 'slot_mb': a.slot_mb, 'workers': a.workers, **res}))


if __name__ == '__main__':
    main()
