#!/bin/bash
# GPU box: rocprofv3 kernel durations of the classical stencils (tools/bench_ops.py, RISP_OPS_ONLY=origin)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/abo; RISP_OPS_REPS=24 RISP_OPS_ONLY="origin" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abo -o o -- python3 "$REPO/tools/bench_ops.py" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/abo/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'anonymous' in r['Name']:
            print('%-80s %8.1f us x %s' % (r['Name'].replace('void (anonymous namespace)::', '')[:80], float(r['AverageNs']) / 1e3, r['Calls']))
PY
