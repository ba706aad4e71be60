#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + HBM PMC passes of `bench.py`.
# Usage: tools/profile_bench.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="${*:---steps 1000 --warmup 100 --no-cpu}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- python3 "$REPO/bench.py" --steps 200 --warmup 20 --no-cpu --no-cnn --no-search > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- python3 "$REPO/bench.py" --steps 200 --warmup 20 --no-cpu --no-cnn --no-search > "$OUT/pmc_write.log" 2>&1
python3 "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
