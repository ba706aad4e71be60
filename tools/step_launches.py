#!/usr/bin/env python3
"""Launches of ONE steady-state DARTS iteration from a rocprofv3 kernel trace of tools/bench_darts.py: the trace is cut at
every 20th (n_step 2) / 25th (n_step 3) prune_softmax_fwd_kernel - one per slot and forward, 5 forwards per iteration -
so model construction (weight packing, broadcasts) is not averaged into the per-step figure.
usage: tools/step_launches.py <d_kernel_trace.csv> [slots=4]"""
import csv, sys
from collections import Counter
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4
marks = [i for i, r in enumerate(rows) if 'prune_softmax_fwd_kernel' in r['Kernel_Name']]
per = 5 * slots
cuts = marks[::per]
steps = [rows[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
steady = steps[len(steps) // 2:]                 # second half: warm caches
n = sum(len(s) for s in steady) / len(steady)
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for s in steady for r in s) / len(steady) / 1e6
wall = (int(steady[-1][-1]['End_Timestamp']) - int(steady[0][0]['Start_Timestamp'])) / len(steady) / 1e6
def cls(name):
    if 'rocclr' in name: return 'runtime copy/fill'
    if name.startswith('Cijk'): return 'Tensile GEMM'
    if 'at::native' in name or 'rocprim' in name or 'hipcub' in name: return 'PyTorch'
    return 'own (libreconfigisp_hip)'
c = Counter(cls(r['Kernel_Name']) for s in steady for r in s)
print('steady-state iterations analysed: %d; launches per iteration: %.0f; kernel time %.2f ms; wall %.2f ms' % (len(steady), n, busy, wall))
for k, v in c.most_common():
    print('  %-28s %7.1f per iteration (%.1f %%)' % (k, v / len(steady), 100.0 * v / len(steady) / n))
names = Counter(r['Kernel_Name'].replace('void ', '')[:100] for s in steady for r in s if cls(r['Kernel_Name']) != 'own (libreconfigisp_hip)')
for k, v in names.most_common(25):
    print('     %6.1f  %s' % (v / len(steady), k))
