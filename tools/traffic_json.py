#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch."""
import csv
import glob
import json
import sys
from collections import defaultdict

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

out_dir, dst = sys.argv[1], sys.argv[2]


def launch_batch(argv):
    """Tiles per trunk launch of the profiled bench command (cfg3 unless --workload cfg2): the engine cuts the slide's tiles into
    equal batches under the --batch cap (engine.batch_sizes), e.g. 24 648 tiles at cap 6 200 -> 6 162, at cap 1 000 -> 986
    (r03 advisor finding: this file used to record the cap, 1000, and bench.py's scaled figure came out 1.4 % low)."""
    from wsi_segmentation_pipeline_amd.engine import batch_sizes
    from wsi_segmentation_pipeline_amd import slide as S

    def opt(name, default):
        return type(default)(argv[argv.index(name) + 1]) if name in argv else default
    cap = opt('--batch', 6200)
    if opt('--workload', 'cfg3') == 'cfg2':
        tiles = opt('--tiles', 10000)
    else:
        size = opt('--size', 40000)
        tiles = len(S.tile_grid(size, size, 256, 256, 256, 256))
    sizes = batch_sizes(tiles, cap)
    return sum(sizes) / len(sizes), tiles, cap


acc = defaultdict(lambda: {'FETCH_SIZE': [], 'WRITE_SIZE': []})
for which in ('fetch', 'write'):
    for f in glob.glob('%s/%s/*/*_counter_collection.csv' % (out_dir, which)):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for name, d in acc.items():
    if not any(k in name for k in ('conv3x3', 'stem_pool', 'conv_gather', 'avgpool', 'stitch', 'softmax')):
        continue
    n = max(len(d['FETCH_SIZE']), len(d['WRITE_SIZE']), 1)
    fetch = 2.0 * 1024.0 * sum(d['FETCH_SIZE']) / max(len(d['FETCH_SIZE']), 1)      # KiB -> bytes, gfx950 x2 correction
    write = 1024.0 * sum(d['WRITE_SIZE']) / max(len(d['WRITE_SIZE']), 1)
    res[name] = {'launches': n, 'read_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
                 'hbm_bytes_per_launch': fetch + write}
slab = {k: v for k, v in res.items() if 'conv3x3s1_' in k}      # slab3 + wide: all stride-1 3x3 launches
tot_l = sum(v['launches'] for v in slab.values())
batch, tiles, cap = launch_batch(sys.argv[3:])
summary = {'bench_args': sys.argv[3:], 'batch': batch, 'tiles': tiles, 'batch_cap': cap, 'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH x2 (gfx950), KiB units',
           'kernels': res,
           'conv3x3_s1_hbm_bytes_per_launch': sum(v['hbm_bytes_per_launch'] * v['launches'] for v in slab.values()) / max(tot_l, 1)}
json.dump(summary, open(dst, 'w'), indent=1)
print(json.dumps(summary)[:600])
