#!/bin/bash
# GPU box: the library with and without hipcc's SLP vectoriser (packed fp32 vector operations), same box, one after the other
cd /tmp
# the product is built with -fno-slp-vectorize (Makefile); the variant re-enables the SLP vectoriser (the later flag wins)
L=$(bash $GRAFT_REPO_ROOT/tools/build_variant.sh /tmp/withslp "-fslp-vectorize" all) || exit 1
for r in 1 2; do for lib in "" "$L"; do
  echo "== library [${lib:-product (no SLP vectoriser)}]"
  RISP_HIP_LIBRARY=$lib python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-cnn --no-search --no-configs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
for lib in "" "$L"; do echo "== ops [${lib:-product (no SLP vectoriser)}]"; RISP_HIP_LIBRARY=$lib RISP_OPS_REPS=24 python3 $GRAFT_REPO_ROOT/tools/bench_ops.py 2>&1 | tail -45 | cut -c1-110; done
