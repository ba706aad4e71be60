cd /tmp
L=$(bash $GRAFT_REPO_ROOT/tools/build_variant.sh /tmp/noslp "-fno-slp-vectorize" all) || exit 1
for lib in "" "$L"; do
  echo "== library [${lib:-default}]"
  export RISP_HIP_LIBRARY=$lib
  python3 $GRAFT_REPO_ROOT/tools/conv_bench.py 64 64 3 32 256 256 30 2>&1 | tail -1
  python3 $GRAFT_REPO_ROOT/tools/conv_bench.py 64 32 5 32 256 256 30 2>&1 | tail -1
  RISP_BENCH_GRAD=1 python3 $GRAFT_REPO_ROOT/tools/conv_bench.py 64 64 3 32 256 256 30 2>&1 | tail -1
  for m in first_exact bwd9_sums fwd5; do python3 $GRAFT_REPO_ROOT/tools/few_channel_bench.py $m 32 256 256 8 20 2>&1 | tail -1; done
  python3 $GRAFT_REPO_ROOT/tools/bench_darts.py 32 256 3 2 2>&1 | tail -1 | cut -c1-90
done
