#!/bin/bash
# GPU box: bank conflicts of candidate tile layouts of the tap-row kernel (tools/lds_probe_tapout.hip) -> conflict cycles per LDS instruction
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 "$REPO/tools/lds_probe_tapout.hip" -o /tmp/lds_probe_tapout || exit 1
rm -rf /tmp/lpt
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/lpt -o p -- /tmp/lds_probe_tapout > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob('/tmp/lpt/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in acc.items():
    print('%-14s conflict cycles per LDS instruction %6.2f   conflict / active %.3f' % (k, v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_INSTS_LDS'], 1), v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1)))
PY
