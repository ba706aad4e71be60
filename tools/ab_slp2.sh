cd /tmp
L=$(bash $GRAFT_REPO_ROOT/tools/build_variant.sh /tmp/noslp "-fno-slp-vectorize" all) || exit 1
for r in 1 2; do for lib in "" "$L"; do
  echo "== library [${lib:-default}]"
  RISP_HIP_LIBRARY=$lib python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-cnn --no-search --no-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
for lib in "" "$L"; do echo "== ops [${lib:-default}]"; RISP_HIP_LIBRARY=$lib RISP_OPS_REPS=24 python3 $GRAFT_REPO_ROOT/tools/bench_ops.py 2>&1 | tail -45 | cut -c1-110; done
