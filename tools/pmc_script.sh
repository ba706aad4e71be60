#!/bin/bash
# GPU box: PMC counters (two passes) for the kernels of a python script, averaged per launch.
# Usage: tools/pmc_script.sh <kernel-substring> <script.py> [args]
KSUB=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_a /tmp/pmc_b
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_a -o c -- python3 "$REPO/$1" "${@:2}" > /tmp/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pmc_b -o c -- python3 "$REPO/$1" "${@:2}" > /tmp/pmc_b.log 2>&1
python3 - "$KSUB" <<'PY'
import csv, glob, sys
from collections import defaultdict
for sub in ('a', 'b'):
    for f in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % sub, recursive=True):
        acc, cnt = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(f)):
            if sys.argv[1] in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
        for k in sorted(acc): print('%-24s avg/launch %16.0f  (%d launches)' % (k, acc[k] / cnt[k], cnt[k]))
PY
tail -2 /tmp/pmc_a.log | head -1
