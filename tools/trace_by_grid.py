#!/usr/bin/env python3
"""Group a rocprofv3 kernel trace by (kernel, grid): calls, average and total duration - launches of one kernel at
different problem sizes (half-resolution Path-Restore, grouped and single members) are then told apart.
usage: tools/trace_by_grid.py <d_kernel_trace.csv> [min_total_us]"""
import csv, sys
from collections import defaultdict
rows = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:48]
    grid = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y'])),
            int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
    k = (name, grid)
    rows[k][0] += 1
    rows[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in rows.values())
floor = float(sys.argv[2]) if len(sys.argv) > 2 else tot / 400
print('%-50s %-18s %7s %10s %12s %6s' % ('kernel', 'workgroups', 'calls', 'avg_us', 'total_us', 'pct'))
for (name, grid), (calls, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if t >= floor:
        print('%-50s %-18s %7d %10.1f %12.1f %6.2f' % (name, 'x'.join(map(str, grid)), calls, t / calls, t, 100 * t / tot))
