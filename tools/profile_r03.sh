#!/bin/bash
# GPU box: everything round 3 commits under profiles/ in one call.  -> gpurun_out/r03/*
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. the bench line: default window, then the round driver's flags
timeout 900 python3 "$REPO/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 900 python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
# 2. kernel trace + HBM counters of the bench command
timeout 1200 bash "$REPO/tools/profile_bench.sh" r03 > "$OUT/profile_bench.log" 2>&1
# 3. search step: BASELINE config 3 (batch 32, n_step 3) and the per-rank batch of config 4 (batch 4, n_step 2); one stream in the trace
timeout 900 bash "$REPO/tools/profile_darts.sh" r03_c3 32 3 2 > "$OUT/config3.log" 2>&1
python3 "$REPO/tools/step_launches.py" "$REPO/gpurun_out/darts_r03_c3/prof/d_kernel_trace.csv" 5 >> "$OUT/config3.log" 2>&1
python3 "$REPO/tools/trace_by_grid.py" "$REPO/gpurun_out/darts_r03_c3/prof/d_kernel_trace.csv" >> "$OUT/config3.log" 2>&1
timeout 900 bash "$REPO/tools/profile_darts.sh" r03_b4 4 2 10 > "$OUT/small_batch.log" 2>&1
python3 "$REPO/tools/step_launches.py" "$REPO/gpurun_out/darts_r03_b4/prof/d_kernel_trace.csv" 4 >> "$OUT/small_batch.log" 2>&1
python3 "$REPO/tools/trace_by_grid.py" "$REPO/gpurun_out/darts_r03_b4/prof/d_kernel_trace.csv" >> "$OUT/small_batch.log" 2>&1
timeout 600 bash "$REPO/tools/profile_darts.sh" r03_b32 32 2 3 > "$OUT/batch32_nstep2.log" 2>&1
# 4. config 5: wall time with the default two tile streams, kernel trace on one stream
python3 "$REPO/tools/bench_split.py" 21 2>&1 | tail -1 > "$OUT/config5.log"
RISP_TILE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5" -o s -- python3 "$REPO/tools/bench_split.py" 21 > "$OUT/cfg5_prof.log" 2>&1
tail -1 "$OUT/cfg5_prof.log" >> "$OUT/config5.log"
python3 "$REPO/tools/trace_by_grid.py" "$OUT/cfg5/s_kernel_trace.csv" 200 >> "$OUT/config5.log" 2>&1
# 5. counters of the 64 -> 64 3x3 layer
timeout 600 bash "$REPO/tools/conv_pmc.sh" r03 > "$OUT/conv_pmc.txt" 2>&1
ls -la "$OUT"
