#!/bin/bash
# GPU box: counters of WbQuadratic's backward kernels (stand-alone and the slot's second launch): where their time goes.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_wbq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RISP_OPS_REPS=12
for what in "quadratic forward +" "slot mixture fused forward +"; do
  tag=$(echo "$what" | cut -c1-4)
  RISP_OPS_ONLY="$what" rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/a_$tag" -o c -- python3 "$REPO/tools/bench_ops.py" > "$OUT/a.log" 2>&1
  RISP_OPS_ONLY="$what" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d "$OUT/b_$tag" -o c -- python3 "$REPO/tools/bench_ops.py" > "$OUT/b.log" 2>&1
  RISP_OPS_ONLY="$what" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t_$tag" -o c -- python3 "$REPO/tools/bench_ops.py" > "$OUT/t.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
root = sys.argv[1]
acc, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(root, '[ab]_*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("wbq", "Wbq")):
            key = (r["Kernel_Name"].replace('void ', '').replace('(anonymous namespace)::', '')[:60], r['Counter_Name'])
            acc[key] += float(r['Counter_Value']); cnt[key] += 1
for k in sorted(acc): print('%-62s %-24s avg/launch %16.0f' % (k[0], k[1], acc[k] / cnt[k]))
for f in glob.glob(os.path.join(root, 't_*', '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("wbq", "Wbq")): print('%-62s avg us %.1f calls %s' % (r["Name"][:60], float(r['AverageNs']) / 1e3, r['Calls']))
PY
find "$OUT" -name "*.csv" -size +256k -delete
