#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: for one kernel (substring match), the distribution of its duration,
the start-to-start period and the idle gap to the previous dispatch, plus what else runs in between."""
import csv, glob, os, sys
from collections import Counter
root, pat = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
med = lambda v: sorted(v)[len(v) // 2] if v else float('nan')
dur, period, gap, between = [], [], [], Counter()
prev_end, prev_start, last_hit = None, None, None
for i, r in enumerate(rows):
    s, e, name = int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']
    if pat in name:
        dur.append((e - s) / 1e3)
        if last_hit is not None:
            period.append((s - int(rows[last_hit]['Start_Timestamp'])) / 1e3)
            gap.append((s - int(rows[i - 1]['End_Timestamp'])) / 1e3)
            for j in range(last_hit + 1, i):
                between[rows[j]['Kernel_Name'][:60]] += 1
        last_hit = i
print('%s: %d dispatches' % (pat, len(dur)))
print('  duration      median %.1f us' % med(dur))
print('  period        median %.1f us   (start to start)' % med(period))
print('  gap to prev   median %.1f us   (previous dispatch end -> this start)' % med(gap))
print('  other kernels between consecutive dispatches:', dict(between))
