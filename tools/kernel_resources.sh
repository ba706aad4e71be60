#!/bin/bash
# Registers / scratch / occupancy of every kernel of one source file, as hipcc reports them (no GPU needed).
# usage: tools/kernel_resources.sh reconfigisp_amd/csrc/risp_conv_small.hip [extra hipcc flags]
SRC=$1; shift
DIR=$(dirname "$SRC")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I"$DIR/../../include" -I"$DIR" "$@" -x hip -c "$SRC" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r'remark: +(.*?) *\[-Rpass', line)
    if not m: continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
    elif ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
        if k.strip().startswith('LDS Size'):
            import subprocess
            name = subprocess.run(['c++filt', cur['name']], capture_output=True, text=True).stdout.strip()
            name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            print('%-52s VGPR %4s  AGPR %3s  SGPR %4s  scratch %5s  waves/SIMD %2s  LDS %6s' % (
                name[:52], cur.get('VGPRs'), cur.get('AGPRs'), cur.get('TotalSGPRs'), cur.get('ScratchSize [bytes/lane]'),
                cur.get('Occupancy [waves/SIMD]'), cur.get('LDS Size [bytes/block]')))
"
