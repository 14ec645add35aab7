#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per-kernel average duration over the launches that ran ALONE on the device.

bench.py keeps two batches in flight in its timed region (two HIP streams), so launches of the two streams overlap there and a
kernel's wall duration in the trace is not its stand-alone duration; the per-kernel HIP-event leg of bench.py (the `kernels` /
`roofline` objects) runs the same steps once more with ONE batch in flight.  The plain `--stats` summary averages over both.
This tool reads the dispatch trace, marks every dispatch whose [start, end) interval intersects no other dispatch's, and prints
calls / average / min / max over those - the figure that must agree with the HIP-event average.
   python tools/rocprof_solo_stats.py <dir with *_kernel_trace.csv> [name substring ...]"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
subs = sys.argv[2:]
solo = [True] * len(rows)
max_end, max_i = -1, -1
for i, (s, e, _) in enumerate(rows):                      # sweep: a dispatch overlaps if it starts before the running maximum end
    if s < max_end:
        solo[i] = False
        solo[max_i] = False
    if e > max_end:
        max_end, max_i = e, i
agg = defaultdict(lambda: [[], []])
for (s, e, name), alone in zip(rows, solo):
    agg[name][0].append(e - s)
    if alone:
        agg[name][1].append(e - s)
print('kernel, calls, average ms (all launches), launches that ran alone, average ms (alone), min ms, max ms')
for name, (al, so) in sorted(agg.items(), key=lambda kv: -sum(kv[1][0])):
    if subs and not any(x in name for x in subs):
        continue
    if not subs and sum(al) < 1e6:
        continue
    print('"%s", %d, %.4f, %d, %s' % (name[:90], len(al), sum(al) / len(al) / 1e6, len(so),
                                      ('%.4f, %.4f, %.4f' % (sum(so) / len(so) / 1e6, min(so) / 1e6, max(so) / 1e6)) if so else '-, -, -'))
