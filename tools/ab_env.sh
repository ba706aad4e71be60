#!/bin/bash
# GPU box: A/B of one environment switch on the DARTS search step, alternating runs on the same box.
# usage: tools/ab_env.sh VAR A B [batch] [n_step] [iters] [rounds]
VAR=$1; A=$2; B=$3; BATCH=${4:-4}; NSTEP=${5:-2}; ITERS=${6:-10}; ROUNDS=${7:-3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for r in $(seq $ROUNDS); do
  for v in $A $B; do
    echo -n "$VAR=$v  "; env $VAR=$v python3 "$REPO/tools/bench_darts.py" $BATCH 256 $NSTEP $ITERS 2>&1 | tail -1 | cut -c1-75
  done
done
