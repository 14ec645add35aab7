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
    if 'conv' not in key[0] and 'stem' not in key[0] and 'maxpool' not in key[0] and 'unet' not in key[0]:
        continue
    vals = {k: sum(v) / len(v) for k, v in acc[key].items()}
    print(key[0], 'grid', key[1], 'wg', key[2], 'n=%d' % len(next(iter(acc[key].values()))))
    for k in sorted(vals):
        print('    %-32s %.4g' % (k, vals[k]))
    g = vals.get
    if g('SQ_VALU_MFMA_BUSY_CYCLES') and g('SQ_BUSY_CYCLES'):
        # SQ_BUSY_CYCLES counts per SE-quad; the *_BUSY / ACTIVE ratios below are per-SIMD shares of the kernel's busy time
        derived = {}
        if g('SQ_INSTS_VALU') and g('SQ_INSTS_MFMA'):
            derived['VALU instructions per MFMA (SQ_INSTS_VALU includes the MFMAs)'] = (g('SQ_INSTS_VALU') - g('SQ_INSTS_MFMA')) / g('SQ_INSTS_MFMA')
        if g('SQ_LDS_IDX_ACTIVE') and g('SQ_LDS_BANK_CONFLICT') is not None:
            derived['LDS bank-conflict cycles / LDS active cycles'] = g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')
        if g('SQ_WAVE_CYCLES') and g('SQ_WAIT_ANY') is not None:
            derived['share of wave lifetime waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES)'] = g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES')
        if g('GRBM_GUI_ACTIVE'):
            # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles; SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs' matrix-pipe cycles
            derived['matrix pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))'] = g('SQ_VALU_MFMA_BUSY_CYCLES') / (g('GRBM_GUI_ACTIVE') * 128.0)
        for k, v in derived.items():
            print('    -> %-96s %.3f' % (k, v))
