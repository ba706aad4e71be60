#!/bin/bash
# GPU box: A/B of a COMPILE-TIME switch of the library on the DARTS search step, alternating A / B on one box.  Both builds go to /tmp
# (tools/build_variant.sh) and are loaded through RISP_HIP_LIBRARY; the in-tree library is not touched.
# usage: tools/ab_build.sh "-DX=0" "-DX=1" [batch] [n_step] [iters] [rounds] [sources the flags reach, default all]
A=$1; B=$2; BATCH=${3:-32}; NSTEP=${4:-2}; ITERS=${5:-4}; ROUNDS=${6:-2}; SRC=${7:-all}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
LA=$(bash "$REPO/tools/build_variant.sh" /tmp/ab_build_a "$A" $SRC) || exit 1
LB=$(bash "$REPO/tools/build_variant.sh" /tmp/ab_build_b "$B" $SRC) || exit 1
for r in $(seq $ROUNDS); do
  echo -n "[$A]  "; (cd /tmp && RISP_HIP_LIBRARY=$LA python3 "$REPO/tools/bench_darts.py" $BATCH 256 $NSTEP $ITERS 2>&1 | tail -1 | cut -c1-80)
  echo -n "[$B]  "; (cd /tmp && RISP_HIP_LIBRARY=$LB python3 "$REPO/tools/bench_darts.py" $BATCH 256 $NSTEP $ITERS 2>&1 | tail -1 | cut -c1-80)
done
