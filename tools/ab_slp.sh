#!/bin/bash
# GPU box: the library with and without hipcc's SLP vectoriser (packed fp32 vector operations) on the same box.  The product is built
# with -fno-slp-vectorize (Makefile); the variant re-enables the vectoriser (the later flag wins).  Layers first, then the search step in
# ALTERNATING runs (the first process after an idle period measures 3-6 % slow whatever it loads: an A-then-B order is not an A/B).
cd /tmp
L=$(bash $GRAFT_REPO_ROOT/tools/build_variant.sh /tmp/withslp "-fslp-vectorize" all) || exit 1
python3 $GRAFT_REPO_ROOT/tools/bench_darts.py 32 256 3 1 > /dev/null 2>&1          # (warm the box)
for lib in "" "$L"; do
  echo "== library [${lib:-product (no SLP vectoriser)}]"
  export RISP_HIP_LIBRARY=$lib
  python3 $GRAFT_REPO_ROOT/tools/conv_bench.py 64 64 3 32 256 256 30 2>&1 | tail -1
  python3 $GRAFT_REPO_ROOT/tools/conv_bench.py 64 32 5 32 256 256 30 2>&1 | tail -1
  RISP_BENCH_GRAD=1 python3 $GRAFT_REPO_ROOT/tools/conv_bench.py 64 64 3 32 256 256 30 2>&1 | tail -1
  for m in first_exact bwd9_sums fwd5; do python3 $GRAFT_REPO_ROOT/tools/few_channel_bench.py $m 32 256 256 8 20 2>&1 | tail -1; done
done
for r in 1 2 3; do
  echo -n "product (no SLP vectoriser): "; RISP_HIP_LIBRARY= python3 $GRAFT_REPO_ROOT/tools/bench_darts.py 32 256 3 2 2>&1 | tail -1 | cut -c1-90
  echo -n "with the SLP vectoriser    : "; RISP_HIP_LIBRARY=$L python3 $GRAFT_REPO_ROOT/tools/bench_darts.py 32 256 3 2 2>&1 | tail -1 | cut -c1-90
done
