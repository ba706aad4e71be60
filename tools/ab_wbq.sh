#!/bin/bash
# GPU box: the WbQuadratic parameter-sum kernels at 2 / 3 / 4 waves per SIMD (register caps 256 / 168 / 128): rocprofv3 kernel durations
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for w in ${WAVES:-2}; do
  touch "$REPO/reconfigisp_amd/csrc/risp_slot.hip" "$REPO/reconfigisp_amd/csrc/risp_pointwise.hip"
  make -C "$REPO/reconfigisp_amd/csrc" -j8 EXTRA="-DRISP_WBQ_WAVES=$w" > /dev/null 2>&1
  rm -rf /tmp/abw; RISP_OPS_REPS=24 RISP_OPS_ONLY="quadratic forward +" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abw -o o -- python3 "$REPO/tools/bench_ops.py" > /dev/null 2>&1
  python3 - $w <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/abw/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'Wbq' in r['Name'] or 'wbq' in r['Name']:
            print('waves', sys.argv[1], r['Name'][:70], '%.1f us' % (float(r['AverageNs']) / 1e3))
PY
  rm -rf /tmp/abw; RISP_OPS_REPS=24 RISP_OPS_ONLY="slot mixture fused forward +" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abw -o o -- python3 "$REPO/tools/bench_ops.py" > /dev/null 2>&1
  python3 - $w <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/abw/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'slot_' in r['Name']:
            print('waves', sys.argv[1], r['Name'][:70], '%.1f us' % (float(r['AverageNs']) / 1e3))
PY
done
