#!/bin/bash
# GPU box: everything round 5 commits under profiles/ in one call.  -> gpurun_out/r05/*
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. the bench line: default window, then the round driver's flags
timeout 900 python3 "$REPO/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 900 python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
# 2. kernel trace + HBM counters of the bench command (-> gpurun_out/prof_r05; tools/collect_r05.py writes profiles/traffic.json from it)
timeout 1200 bash "$REPO/tools/profile_bench.sh" r05 > "$OUT/profile_bench.log" 2>&1
# 3. search step: BASELINE config 3 (batch 32, n_step 3) and the per-rank batch of config 4 (batch 4, n_step 2); one stream in the trace
timeout 900 bash "$REPO/tools/profile_darts.sh" r05_c3 32 3 2 > "$OUT/config3.log" 2>&1
python3 "$REPO/tools/step_launches.py" "$REPO/gpurun_out/darts_r05_c3/prof/d_kernel_trace.csv" 5 >> "$OUT/config3.log" 2>&1
python3 "$REPO/tools/trace_by_grid.py" "$REPO/gpurun_out/darts_r05_c3/prof/d_kernel_trace.csv" >> "$OUT/config3.log" 2>&1
timeout 900 bash "$REPO/tools/profile_darts.sh" r05_b4 4 2 10 > "$OUT/small_batch.log" 2>&1
python3 "$REPO/tools/step_launches.py" "$REPO/gpurun_out/darts_r05_b4/prof/d_kernel_trace.csv" 4 >> "$OUT/small_batch.log" 2>&1
timeout 600 bash "$REPO/tools/profile_darts.sh" r05_b32 32 2 3 > "$OUT/batch32_nstep2.log" 2>&1
# ... and the fp32 arithmetic on the same box, wall time only
{ RISP_CONV_ARITH=f32 python3 "$REPO/tools/bench_darts.py" 32 256 3 2 2>&1 | tail -1; RISP_CONV_ARITH=f32 python3 "$REPO/tools/bench_darts.py" 4 256 2 8 2>&1 | tail -1; RISP_CONV_ARITH=f32 python3 "$REPO/tools/bench_split.py" 16 2>&1 | tail -1; } > "$OUT/f32_arith_same_box.log" 2>&1
# ... and round 4's training first layers (fp32 kernel) for the first-layer default
{ RISP_CONV_TOEP_FIRST=infer python3 "$REPO/tools/bench_darts.py" 32 256 3 2 2>&1 | tail -1; RISP_CONV_TOEP_FIRST=infer python3 "$REPO/tools/bench_darts.py" 4 256 2 8 2>&1 | tail -1; } > "$OUT/first_layer_infer_same_box.log" 2>&1
# 4. config 5: wall time with the default two tile streams, kernel trace on one stream
for b in 16 21 32 63; do python3 "$REPO/tools/bench_split.py" $b 2>&1 | tail -1; done > "$OUT/config5.log"
RISP_TILE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5" -o s -- python3 "$REPO/tools/bench_split.py" 16 > "$OUT/cfg5_prof.log" 2>&1
tail -1 "$OUT/cfg5_prof.log" >> "$OUT/config5.log"
python3 "$REPO/tools/trace_by_grid.py" "$OUT/cfg5/s_kernel_trace.csv" 200 >> "$OUT/config5.log" 2>&1
# 5. counters of the 64 -> 64 3x3 layer and of the 5x5 64 -> 32 layer (the wave-specialised split-precision kernel)
timeout 600 bash "$REPO/tools/conv_pmc.sh" r05 64 64 3 32 256 256 > "$OUT/conv_pmc.txt" 2>&1
timeout 600 bash "$REPO/tools/conv_pmc.sh" r05_5x5 64 32 5 32 256 256 > "$OUT/conv_pmc_5x5.txt" 2>&1
# 6. the wave-specialised kernel against the round-4 kernel (interleaved rounds, one process), every epilogue; where its waves' life goes
{
  python3 "$REPO/tools/ab_ws.py" 2>&1 | grep "round-4"
  python3 "$REPO/tools/ab_ws.py" 21 256 256 2>&1 | grep "round-4"
  python3 "$REPO/tools/ab_ws.py" 4 256 256 2>&1 | grep "round-4"
  RISP_AB_K=5 python3 "$REPO/tools/ab_ws.py" 32 256 256 64 32 2>&1 | grep "round-4"
  RISP_AB_K=5 python3 "$REPO/tools/ab_ws.py" 32 256 256 32 64 2>&1 | grep "round-4"
  RISP_AB_EPIS="0 2 3 4" python3 "$REPO/tools/ab_ws_build.py" "" "-DWS_SPLIT=0" "-DWS_SPLIT=2" 2>&1 | grep "bits"
  RISP_AB_SHAPE="4 256 256" RISP_AB_EPIS="0 2" python3 "$REPO/tools/ab_ws_build.py" "" "-DWS_SPLIT=0" 2>&1 | grep "bits"
  for m in 0 1 2 3; do python3 "$REPO/tools/ws_stamps.py" $m 2>&1 | tail -3; done
  for m in 1 3; do python3 "$REPO/tools/ws_stamps.py" $m -DWS_SPLIT=0 2>&1 | tail -3; done
} > "$OUT/ws_ladder.txt" 2>&1
# 7. weight gradients and the proxy fine-tuning step
{ python3 "$REPO/tools/bench_wgrad.py" 2>&1 | tail -3; python3 "$REPO/tools/bench_ft.py" 2>&1 | tail -1; } > "$OUT/wgrad_ft.txt" 2>&1
# 8. every stand-alone kernel for the per-op table
timeout 900 bash "$REPO/tools/profile_ops.sh" r05 > "$OUT/ops.log" 2>&1
# 9. where the wave-specialised kernel's LDS bank conflicts come from (ablation builds under --pmc) and what WbQuadratic's backward
#    kernels spend outside their loop (ablation builds: empty launch / one vector per thread / the walk without the arithmetic / full).
#    Both rebuild the library on the box and leave it in the default configuration: last.
timeout 900 bash "$REPO/tools/ws_conflicts.sh" > "$OUT/ws_conflicts.txt" 2>&1
timeout 900 bash "$REPO/tools/ab_wbq.sh" "-DRISP_WBQ_ABL=3" "-DRISP_WBQ_ABL=2" "-DRISP_WBQ_ABL=1" "" 2>&1 | grep "wbq_params\|bwd_kernel<risp_ops::Wbq" > "$OUT/wbq_ablation.txt"
# gpurun copies back at most 64 MiB: the raw traces and counter dumps stay on the box, the summaries made from them travel
find "$REPO/gpurun_out" -name "*.csv" -size +256k -delete
du -sh "$REPO/gpurun_out"
ls -la "$OUT"
