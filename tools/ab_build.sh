#!/bin/bash
# GPU box: A/B of a COMPILE-TIME switch of the library on the DARTS search step - rebuilds the library in the box's scratch
# copy between runs (hipcc is on the box), alternating A / B.   usage: tools/ab_build.sh "-DX=0" "-DX=1" [batch] [n_step] [iters] [rounds]
A=$1; B=$2; BATCH=${3:-32}; NSTEP=${4:-2}; ITERS=${5:-4}; ROUNDS=${6:-2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq $ROUNDS); do
  for v in "$A" "$B"; do
    touch "$REPO"/reconfigisp_amd/csrc/*.hip
    make -s -C "$REPO/reconfigisp_amd/csrc" -j8 EXTRA="$v" > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; exit 1; }
    echo -n "[$v]  "; (cd /tmp && python3 "$REPO/tools/bench_darts.py" $BATCH 256 $NSTEP $ITERS 2>&1 | tail -1 | cut -c1-80)
  done
done
# leave the box's library in its default configuration (the last variant would otherwise stay installed for whatever runs next)
touch "$REPO"/reconfigisp_amd/csrc/*.hip
make -s -C "$REPO/reconfigisp_amd/csrc" -j8 > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; exit 1; }
