#!/bin/bash
# GPU box: counters for one conv layer (clock, MFMA busy, wait breakdown).  Usage: tools/conv_pmc.sh <tag> [conv_bench args]
# RISP_PMC_PROG = another program under tools/ (few_channel_bench.py ...), RISP_PMC_KERNELS = the kernel-name fragments to keep
set -u
PROG=${RISP_PMC_PROG:-conv_bench.py}
export RISP_PMC_KERNELS=${RISP_PMC_KERNELS:-conv_mfma,conv_wino,conv_f16x2}
TAG=${1:-conv}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/a" -o c -- python3 "$REPO/tools/$PROG" "$@" > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/b" -o c -- python3 "$REPO/tools/$PROG" "$@" > "$OUT/b.log" 2>&1
if [ "${RISP_PMC_HBM:-0}" = 1 ]; then      # HBM bytes: FETCH_SIZE and WRITE_SIZE in their own passes (KiB; FETCH_SIZE x 2 on gfx950, profiles/r02_fetch_size_calibration.txt)
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -o c -- python3 "$REPO/tools/$PROG" "$@" > "$OUT/f.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -o c -- python3 "$REPO/tools/$PROG" "$@" > "$OUT/w.log" 2>&1
fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o c -- python3 "$REPO/tools/$PROG" "$@" > "$OUT/t.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
root = sys.argv[1]
KEEP = os.environ['RISP_PMC_KERNELS'].split(',')
for sub in ('a', 'b', 'f', 'w'):
    for f in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        acc, cnt = defaultdict(float), defaultdict(int)
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in KEEP):
                acc[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
        for k in acc: print('%-28s avg/launch %16.0f' % (k, acc[k] / cnt[k]))
for f in glob.glob(os.path.join(root, 't', '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in KEEP): print('kernel avg us %.1f calls %s' % (float(r['AverageNs']) / 1e3, r['Calls']))
PY
tail -1 "$OUT/t.log"
