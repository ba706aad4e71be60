#!/bin/bash
# GPU box: kernel trace of the driver's short bench window (--steps 20 --warmup 5): where the 20 timed steps spend their time.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/k20
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof" -o k -- python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu --no-cnn --no-search > "$OUT/bench.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
f = glob.glob(os.path.join(root, 'prof', '**', '*kernel_trace.csv'), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
# groups of consecutive bilateral_chain launches separated by idle gaps > 25 us
groups, cur = [], []
for s, e, n in ks:
    if 'bilateral_chain_kernel' not in n:
        if cur: groups.append(cur); cur = []
        continue
    if cur and s - cur[-1][1] > 25000:
        groups.append(cur); cur = []
    cur.append((s, e))
if cur: groups.append(cur)
for g in groups:
    if len(g) in (20, 25) or 18 <= len(g) <= 30:
        durs = [(e - s) / 1e3 for s, e in g]
        gaps = [(g[i + 1][0] - g[i][1]) / 1e3 for i in range(len(g) - 1)]
        print('group of %d launches: span %.1f us, kernel durations first 6 %s ... last 3 %s, mean %.2f; gaps mean %.2f max %.2f'
              % (len(g), (g[-1][1] - g[0][0]) / 1e3, ['%.1f' % d for d in durs[:6]], ['%.1f' % d for d in durs[-3:]], sum(durs) / len(durs),
                 sum(gaps) / max(len(gaps), 1), max(gaps) if gaps else 0))
print(open(os.path.join(root, 'bench.log')).read().strip().splitlines()[-1][:300])
PY
