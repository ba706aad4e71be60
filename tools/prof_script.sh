#!/bin/bash
# GPU box: rocprofv3 kernel stats of an arbitrary python script.  Usage: tools/prof_script.sh <script.py> [args]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_script
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_script -o o -- python3 "$REPO/$1" "${@:2}" > /tmp/prof_script.log 2>&1
tail -3 /tmp/prof_script.log; python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/prof_script/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print('%-70s calls %6s avg_us %10.2f pct %s' % (r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
