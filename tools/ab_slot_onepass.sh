#!/bin/bash
# GPU box: the fused slot mixture's backward with WbQuadratic's 30 sums in a second launch (default) against one launch
# (-DRISP_SLOT_WBQ_ONE_PASS=1): rocprofv3 kernel durations of tools/bench_ops.py (slot section only) for both builds.  The library
# is rebuilt in the box's scratch copy and left in its default configuration.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
trap 'touch "$REPO"/reconfigisp_amd/csrc/risp_slot.hip; make -s -C "$REPO/reconfigisp_amd/csrc" -j8 > /tmp/ab_build.log 2>&1' EXIT
for v in "" "$@"; do
  touch "$REPO"/reconfigisp_amd/csrc/risp_slot.hip
  make -s -C "$REPO/reconfigisp_amd/csrc" -j8 EXTRA="$v" > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; exit 1; }
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
