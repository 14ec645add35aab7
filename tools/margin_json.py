#!/usr/bin/env python3
"""tests/test_gpu_margin.py -s output -> profiles/rNN_margin_families.json (the `contract.families_max` of bench.py):
   python -m pytest tests/test_gpu_margin.py -s -q > margin.txt; python tools/margin_json.py margin.txt profiles/r04_margin_families.json "note" """
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ''
per = {}
pat = re.compile(r'(\w+): max \|logit\| ([\d.]+); max abs err parity f32 ([\d.e+-]+) / u8 ([\d.e+-]+); mx f32 ([\d.e+-]+) / u8 ([\d.e+-]+)')
for line in open(src):
    m = pat.search(line)
    if m:
        per[m.group(1)] = {'max_abs_logit': float(m.group(2)), 'parity': [float(m.group(3)), float(m.group(4))],
                           'mx': [float(m.group(5)), float(m.group(6))]}
if len(per) < 5:
    raise SystemExit('expected five margin families in %s, found %s' % (src, sorted(per)))
out = {'source': 'tests/test_gpu_margin.py -s on MI355X%s: max over the five reference-generated margin families (tests/golden/margin_*.npz) and both input '
                 'paths (f32 tensor / fused u8 slide) of max |logit - reference|' % ((' (' + note + ')') if note else ''),
       'mx': max(max(v['mx']) for v in per.values()), 'parity': max(max(v['parity']) for v in per.values()), 'per_family': per}
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps({k: out[k] for k in ('mx', 'parity')}))
