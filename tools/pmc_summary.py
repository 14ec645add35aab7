#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel name (+ grid)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'][:70]
        key = (name, r.get('Grid_Size', ''), r.get('Workgroup_Size', ''))
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key in sorted(acc):
    if 'conv' not in key[0] and 'stem' not in key[0] and 'maxpool' not in key[0]:
        continue
    vals = {k: sum(v) / len(v) for k, v in acc[key].items()}
    print(key[0], 'grid', key[1], 'wg', key[2], 'n=%d' % len(next(iter(acc[key].values()))))
    for k in sorted(vals):
        print('    %-32s %.4g' % (k, vals[k]))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in vals and 'SQ_BUSY_CYCLES' in vals:
        pass
