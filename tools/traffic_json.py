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
# Layer-1 kernels read 96-byte lines (r03+), for which FETCH_SIZE needs its own factors (tools/probes/fetch_calib96 under
# --pmc FETCH_SIZE, profiles/r04_fetch_calib96.txt): the slab pattern is tallied at CAL96_SLAB of its bytes, the tail's residual
# tile pattern at CAL96_RESID.  Launches alternate without / with a residual, so the per-dispatch counters fall into two
# clusters: reads(no residual) = A / CAL96_SLAB, reads(residual) = A / CAL96_SLAB + (B - A) / CAL96_RESID.
# r05: 96-byte lines are line-planar - contiguous reads, the guide's x2 rule (factor 0.5) applies to both patterns (kplanar of the probe);
# r03-r04 (lines interleaved per pixel): slab 0.945, residual 0.5
CAL96_SLAB = float(os.environ.get('WSI_CAL96_SLAB', '0.5'))
CAL96_RESID = float(os.environ.get('WSI_CAL96_RESID', '0.5'))


def layer1_reads(values_kib):
    v = sorted(values_kib)
    if len(v) < 2:
        return None
    lo, hi = v[:len(v) // 2], v[len(v) // 2:]
    a, b = 1024.0 * sum(lo) / len(lo), 1024.0 * sum(hi) / len(hi)
    return 0.5 * (a / CAL96_SLAB + a / CAL96_SLAB + max(b - a, 0.0) / CAL96_RESID), a, b


res = {}
for name, d in acc.items():
    if not any(k in name for k in ('conv3x3', 'stem_pool', 'conv_gather', 'avgpool', 'stitch', 'softmax')):
        continue
    n = max(len(d['FETCH_SIZE']), len(d['WRITE_SIZE']), 1)
    fetch = 2.0 * 1024.0 * sum(d['FETCH_SIZE']) / max(len(d['FETCH_SIZE']), 1)      # KiB -> bytes, gfx950 x2 correction
    write = 1024.0 * sum(d['WRITE_SIZE']) / max(len(d['WRITE_SIZE']), 1)
    res[name] = {'launches': n, 'read_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
                 'hbm_bytes_per_launch': fetch + write}
    if ('rows_kernel' in name or 'slab3_kernel<4, 2, 2, 3' in name) and '--s2' not in ' '.join(sys.argv[3:]):
        cal = layer1_reads(d['FETCH_SIZE'])
        if cal:
            res[name]['read_bytes_per_launch_calibrated96'] = cal[0]
            res[name]['hbm_bytes_per_launch_calibrated96'] = cal[0] + write
            res[name]['fetch_raw_bytes_no_residual_launches'] = cal[1]
            res[name]['fetch_raw_bytes_residual_launches'] = cal[2]
            res[name]['calibration'] = {'slab_pattern_factor': CAL96_SLAB, 'residual_pattern_factor': CAL96_RESID,
                                        'source': 'tools/probes/fetch_calib96 under rocprofv3 --pmc FETCH_SIZE'}
slab = {k: v for k, v in res.items() if 'conv3x3s1_' in k}      # slab3 + wide: all stride-1 3x3 launches
tot_l = sum(v['launches'] for v in slab.values())
batch, tiles, cap = launch_batch(sys.argv[3:])
summary = {'bench_args': sys.argv[3:], 'batch': batch, 'tiles': tiles, 'batch_cap': cap, 'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH x2 (gfx950), KiB units',
           'kernels': res,
           'conv3x3_s1_hbm_bytes_per_launch': sum(v['hbm_bytes_per_launch'] * v['launches'] for v in slab.values()) / max(tot_l, 1)}
json.dump(summary, open(dst, 'w'), indent=1)
print(json.dumps(summary)[:600])
