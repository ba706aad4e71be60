#!/bin/bash
# GPU box: kernel statistics of the DARTS search step at a given batch (one stream, so that per-launch durations can
# be attributed).  usage: tools/profile_darts.sh <tag> <batch> <n_step> <iters> [size]   -> gpurun_out/darts_<tag>/summary.txt
set -u
TAG=${1:-r03}; BATCH=${2:-32}; NSTEP=${3:-2}; ITERS=${4:-3}; SIZE=${5:-256}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/darts_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$REPO/tools/bench_darts.py" $BATCH $SIZE $NSTEP $ITERS 2>&1 | tail -1 > "$OUT/two_streams.log"
export RISP_SLOT_STREAMS=1
python3 "$REPO/tools/bench_darts.py" $BATCH $SIZE $NSTEP $ITERS 2>&1 | tail -1 > "$OUT/one_stream.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o d -- python3 "$REPO/tools/bench_darts.py" $BATCH $SIZE $NSTEP $ITERS > "$OUT/prof.log" 2>&1
python3 - "$OUT" $ITERS <<'PY'
import csv, glob, os, sys
root, iters = sys.argv[1], int(sys.argv[2]) + 1
f = glob.glob(os.path.join(root, 'prof', '**', '*kernel_stats.csv'), recursive=True)
with open(os.path.join(root, 'summary.txt'), 'w') as out:
    out.write('two streams : %s\n' % open(os.path.join(root, 'two_streams.log')).read().strip())
    out.write('one stream  : %s\n' % open(os.path.join(root, 'one_stream.log')).read().strip())
    lines = [l for l in open(os.path.join(root, 'prof.log')).read().splitlines() if l.startswith('search step')]
    out.write('profiled (one stream): %s\n' % (lines[-1] if lines else ''))
    if f:
        rows = list(csv.DictReader(open(f[0])))
        tot = sum(float(r['TotalDurationNs']) for r in rows)
        calls = sum(int(r['Calls']) for r in rows)
        out.write('kernel time over the profiled run (%d steps): %.1f ms in %d launches = %.1f ms and %d launches per step\n'
                  % (iters, tot / 1e6, calls, tot / 1e6 / iters, calls // iters))
        out.write('%-86s %7s %12s %10s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
        for r in rows[:40]:
            n = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:84]
            out.write('%-86s %7s %12.1f %10.2f %7.2f\n' % (n, r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                                          float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print(open(os.path.join(root, 'summary.txt')).read())
PY
