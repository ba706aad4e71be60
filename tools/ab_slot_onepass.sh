#!/bin/bash
# GPU box: the fused slot mixture's backward with WbQuadratic's 30 sums inside the one launch (the default since round 6) against a second
# launch (-DRISP_SLOT_WBQ_ONE_PASS=0) and other builds of risp_slot.hip given as arguments: rocprofv3 kernel durations of
# tools/bench_ops.py (slot section only).  Every build goes to /tmp (tools/build_variant.sh); the in-tree library is not touched.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for v in "" "$@"; do
  i=$((i + 1))
  export RISP_HIP_LIBRARY=$(bash "$REPO/tools/build_variant.sh" /tmp/ab_slot_$i "$v" ${RISP_AB_SRC:-risp_slot.hip}) || exit 1
  rm -rf /tmp/ab_slot_prof
  RISP_OPS_ONLY=slot RISP_OPS_REPS=24 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_slot_prof -o o -- python3 "$REPO/tools/bench_ops.py" > /tmp/ab_slot.log 2>&1
  echo "== build [$v]"
  python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ab_slot_prof/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'slot_' in r['Name']:
        print('   %-70s calls %5s  avg %8.2f us' % (r['Name'].replace('(anonymous namespace)::', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
