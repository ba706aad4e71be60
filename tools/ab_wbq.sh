#!/bin/bash
# GPU box: WbQuadratic's backward kernels under different build flags (one argument = one set of -D flags for risp_slot.hip /
# risp_pointwise.hip, e.g. "-DRISP_WBQ_WAVES=3 -DRISP_WBQ_AHEAD=1"): rocprofv3 kernel durations.  Every build goes to /tmp (tools/build_variant.sh); the
# in-tree library is not touched.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i + 1))
  export RISP_HIP_LIBRARY=$(bash "$REPO/tools/build_variant.sh" /tmp/ab_wbq_$i "$cfg" risp_slot.hip risp_pointwise.hip risp_core.cpp risp_reduce.hip) || exit 1
  for what in "quadratic forward +" "slot mixture fused forward +"; do
    rm -rf /tmp/abw; RISP_OPS_REPS=24 RISP_OPS_ONLY="$what" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abw -o o -- python3 "$REPO/tools/bench_ops.py" > /dev/null 2>&1
    python3 - "$cfg" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/abw/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'Wbq' in r['Name'] or 'wbq' in r['Name'] or 'slot_mix_bwd' in r['Name']:
            print('[%s] %-60s %.1f us' % (sys.argv[1], r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:60], float(r['AverageNs']) / 1e3))
PY
  done
done
