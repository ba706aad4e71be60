#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into a small
text summary that is committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    hits = glob.glob(os.path.join(root, sub, '**', pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace('void ', '').replace('(anonymous namespace)::', '')
    for cut in ('(risp_conv_desc)', '(ChainArgs)'):
        name = name.replace(cut, '')
    return name[:70]


f = find('trace', '*kernel_stats.csv')
if f:
    print('== kernel stats (%s)' % os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    print('%-72s %8s %12s %12s %8s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
    for r in rows[:25]:
        print('%-72s %8s %12.1f %12.2f %8s' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                                float(r['AverageNs']) / 1e3, r['Percentage']))
for tag, counter in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    f = find(tag, '*counter_collection.csv')
    if not f:
        continue
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == counter:
            acc[r['Kernel_Name']] += float(r['Counter_Value'])
            cnt[r['Kernel_Name']] += 1
    print('== %s per launch (raw counter units = KiB; FETCH_SIZE under-reports wide streaming reads 2x on gfx950)' % counter)
    top = sorted(acc, key=lambda k: -acc[k])
    # the twelve largest, and ALWAYS the two kernels of the bench line (tools/collect_r05.py reads their rows into profiles/traffic.json)
    for k in top[:12] + [k for k in top[12:] if 'chain_kernel' in k]:
        print('%-72s launches %5d  avg %14.1f KiB' % (short(k), cnt[k], acc[k] / cnt[k]))
