#!/bin/bash
# GPU box: the search step at the per-rank batch of the 8-GPU configuration (BASELINE config 4: 4 images, n_step 2):
# wall time with the slot ops on two streams and on one, and the kernel-time sum of the one-stream run (what the GPU
# is busy for; the rest of the wall time is launch / host overhead).  -> gpurun_out/small_<tag>/
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/small_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$REPO/tools/bench_darts.py" 4 256 2 10 2>&1 | tail -1 > "$OUT/two_streams.log"
export RISP_SLOT_STREAMS=1
python3 "$REPO/tools/bench_darts.py" 4 256 2 10 2>&1 | tail -1 > "$OUT/one_stream.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o d -- python3 "$REPO/tools/bench_darts.py" 4 256 2 10 > "$OUT/prof.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
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
        out.write('kernel time over the profiled run (11 steps): %.1f ms in %d launches = %.1f ms and %d launches per step\n'
                  % (tot / 1e6, calls, tot / 1e6 / 11, calls // 11))
        out.write('%-86s %7s %12s %10s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
        for r in rows[:25]:
            n = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:84]
            out.write('%-86s %7s %12.1f %10.2f %7.2f\n' % (n, r['Calls'], float(r['TotalDurationNs']) / 1e3,
                                                          float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print(open(os.path.join(root, 'summary.txt')).read())
PY
