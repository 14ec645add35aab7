#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch."""
import csv
import glob
import json
import sys
from collections import defaultdict

out_dir, dst = sys.argv[1], sys.argv[2]
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
summary = {'bench_args': sys.argv[3:], 'batch': 1000, 'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH x2 (gfx950), KiB units',
           'kernels': res,
           'conv3x3_s1_hbm_bytes_per_launch': sum(v['hbm_bytes_per_launch'] * v['launches'] for v in slab.values()) / max(tot_l, 1)}
json.dump(summary, open(dst, 'w'), indent=1)
print(json.dumps(summary)[:600])
