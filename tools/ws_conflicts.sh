#!/bin/bash
# GPU box: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the wave-specialised 3x3 kernel, full and with parts of the producers' LDS writes
# compiled out (diagnostic builds in /tmp through tools/build_variant.sh, wrong results; the in-tree library is not touched): which accesses conflict
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "" "-DRISP_WS_ABL_NO_HALO" "-DRISP_WS_ABL_NO_HALO -DRISP_WS_ABL_NO_QUAD"; do
  i=$((i + 1))
  export RISP_HIP_LIBRARY=$(bash "$REPO/tools/build_variant.sh" /tmp/ab_wsc_$i "$flags" risp_conv_f16x2_ws.hip) || exit 1
  rm -rf /tmp/wsc
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/wsc -o c -- python3 "$REPO/tools/conv_bench.py" 64 64 3 32 256 256 > /dev/null 2>&1
  python3 - "$flags" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob('/tmp/wsc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_f16x2' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
print('[%s]' % sys.argv[1], {k: round(acc[k] / cnt[k]) for k in acc}, 'conflict share %.3f' % (acc['SQ_LDS_BANK_CONFLICT'] / max(acc['SQ_LDS_IDX_ACTIVE'], 1)))
PY
done
