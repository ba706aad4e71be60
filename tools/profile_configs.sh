#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats of BASELINE.json configs 3 (DARTS search step) and 5 (tiled full frame),
# plus the counter passes of the dominant 64->64 3x3 convolution layer.  -> gpurun_out/cfg_<tag>/
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/cfg_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/darts" -o d -- python3 "$REPO/tools/bench_darts.py" 32 256 3 2 > "$OUT/darts.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/split" -o s -- python3 "$REPO/tools/bench_split.py" > "$OUT/split.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    for cut in ('(risp_conv_desc)', '(ChainArgs)'): n = n.replace(cut, '')
    return n[:84]
for sub, log in (('darts', 'darts.log'), ('split', 'split.log')):
    f = glob.glob(os.path.join(root, sub, '**', '*kernel_stats.csv'), recursive=True)
    with open(os.path.join(root, sub + '_summary.txt'), 'w') as out:
        lines = [l for l in open(os.path.join(root, log)).read().splitlines() if l.strip() and not l[:5] in ("E2026", "W2026", "I2026")]
        out.write("== %s\n" % (lines[-1] if lines else ""))
        if f:
            rows = list(csv.DictReader(open(f[0])))
            tot = sum(float(r['TotalDurationNs']) for r in rows)
            out.write('%-86s %7s %12s %10s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
            for r in rows[:22]:
                out.write('%-86s %7s %12.1f %10.2f %7.2f\n' % (short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                                              float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
            conv = sum(float(r['TotalDurationNs']) for r in rows if 'conv_mfma' in r['Name'] or 'conv_wgrad' in r['Name'])
            out.write('matrix-core convolution kernels: %.1f %% of GPU kernel time\n' % (100 * conv / tot))
    print(open(os.path.join(root, sub + '_summary.txt')).read())
PY
bash "$REPO/tools/conv_pmc.sh" "$TAG" > "$OUT/conv_pmc.txt" 2>&1
cat "$OUT/conv_pmc.txt"
