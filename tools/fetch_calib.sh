#!/bin/bash
# GPU box: FETCH_SIZE correction factors per load width (tools/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/fetch_calib
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib "$REPO/tools/fetch_calib.hip" || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc" -o c -- /tmp/fetch_calib > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY' | tee "$OUT/../fetch_calib.txt"
import csv, glob, os, sys
from collections import defaultdict
acc, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], 'pmc', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE':
            acc[r['Kernel_Name']] += float(r['Counter_Value']); cnt[r['Kernel_Name']] += 1
print('# FETCH_SIZE calibration, 1 GiB read once per launch (tools/fetch_calib.hip); counter unit = KiB')
for k in sorted(acc):
    kib = acc[k] / cnt[k]
    print('%-60s FETCH_SIZE %12.1f KiB  -> factor (bytes read / counter bytes) = %.3f' % (k[:60], kib, (1 << 30) / (kib * 1024)))
PY
