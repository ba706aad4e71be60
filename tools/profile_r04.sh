#!/bin/bash
# GPU box: everything round 4 commits under profiles/ in one call.  -> gpurun_out/r04/*
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. the bench line: default window, then the round driver's flags
timeout 900 python3 "$REPO/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 900 python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
# 2. kernel trace + HBM counters of the bench command
timeout 1200 bash "$REPO/tools/profile_bench.sh" r04 > "$OUT/profile_bench.log" 2>&1
# 3. search step: BASELINE config 3 (batch 32, n_step 3) and the per-rank batch of config 4 (batch 4, n_step 2); one stream in the trace
timeout 900 bash "$REPO/tools/profile_darts.sh" r04_c3 32 3 2 > "$OUT/config3.log" 2>&1
python3 "$REPO/tools/step_launches.py" "$REPO/gpurun_out/darts_r04_c3/prof/d_kernel_trace.csv" 5 >> "$OUT/config3.log" 2>&1
python3 "$REPO/tools/trace_by_grid.py" "$REPO/gpurun_out/darts_r04_c3/prof/d_kernel_trace.csv" >> "$OUT/config3.log" 2>&1
timeout 900 bash "$REPO/tools/profile_darts.sh" r04_b4 4 2 10 > "$OUT/small_batch.log" 2>&1
python3 "$REPO/tools/step_launches.py" "$REPO/gpurun_out/darts_r04_b4/prof/d_kernel_trace.csv" 4 >> "$OUT/small_batch.log" 2>&1
python3 "$REPO/tools/trace_by_grid.py" "$REPO/gpurun_out/darts_r04_b4/prof/d_kernel_trace.csv" >> "$OUT/small_batch.log" 2>&1
timeout 600 bash "$REPO/tools/profile_darts.sh" r04_b32 32 2 3 > "$OUT/batch32_nstep2.log" 2>&1
# ... and the fp32 arithmetic on the same box, wall time only
for a in f32; do RISP_CONV_ARITH=$a python3 "$REPO/tools/bench_darts.py" 32 256 3 2 2>&1 | tail -1; RISP_CONV_ARITH=$a python3 "$REPO/tools/bench_darts.py" 4 256 2 8 2>&1 | tail -1; RISP_CONV_ARITH=$a python3 "$REPO/tools/bench_split.py" 21 2>&1 | tail -1; done > "$OUT/f32_arith_same_box.log" 2>&1
# 4. config 5: wall time with the default two tile streams, kernel trace on one stream
python3 "$REPO/tools/bench_split.py" 21 2>&1 | tail -1 > "$OUT/config5.log"
python3 "$REPO/tools/bench_split.py" 16 2>&1 | tail -1 >> "$OUT/config5.log"
python3 "$REPO/tools/bench_split.py" 63 2>&1 | tail -1 >> "$OUT/config5.log"
RISP_TILE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5" -o s -- python3 "$REPO/tools/bench_split.py" 21 > "$OUT/cfg5_prof.log" 2>&1
tail -1 "$OUT/cfg5_prof.log" >> "$OUT/config5.log"
python3 "$REPO/tools/trace_by_grid.py" "$OUT/cfg5/s_kernel_trace.csv" 200 >> "$OUT/config5.log" 2>&1
# 5. counters of the 64 -> 64 3x3 layer (split precision, then the fp32 F(4,3) kernel) and of the 5x5 64 -> 32 layer
timeout 600 bash "$REPO/tools/conv_pmc.sh" r04 64 64 3 32 256 256 > "$OUT/conv_pmc.txt" 2>&1
RISP_CONV_ARITH=f32 timeout 600 bash "$REPO/tools/conv_pmc.sh" r04_f32 64 64 3 32 256 256 > "$OUT/conv_pmc_f32.txt" 2>&1
timeout 600 bash "$REPO/tools/conv_pmc.sh" r04_5x5 64 32 5 32 256 256 > "$OUT/conv_pmc_5x5.txt" 2>&1
# 6. the split-precision kernel: error and time beside the fp32 kernels, every epilogue; where a wave's life goes
{
  for e in 0 1 2 3; do echo "== 3x3 64->64 on 32 x 256 x 256, epilogue mode $e (0 ReLU, 1 residual + ReLU, 2 mask, 3 residual + mask)"; RISP_AB_EPI=$e python3 "$REPO/tools/ab_f16x2.py" "" 2>&1 | grep "status\|median"; done
  echo "== 5x5 64->32"; RISP_AB_K=5 RISP_AB_CH="64 32" python3 "$REPO/tools/ab_f16x2.py" "" 2>&1 | grep "status\|median"
  echo "== 5x5 32->64, mask epilogue (a backward-data pass)"; RISP_AB_EPI=2 RISP_AB_K=5 RISP_AB_CH="32 64" python3 "$REPO/tools/ab_f16x2.py" "" 2>&1 | grep "status\|median"
  echo "== where a wave's life goes (in-kernel stamps; two workgroups per CU, then one)"
  python3 "$REPO/tools/ab_f16x2.py" "-DRISP_H2_STAMPS" "-DRISP_H2_STAMPS,-DRISP_H2_WGS=1" 2>&1 | grep -v "^/opt\|status"
  RISP_AB_K=5 RISP_AB_CH="64 32" python3 "$REPO/tools/ab_f16x2.py" "-DRISP_H2_STAMPS" 2>&1 | grep -v "^/opt\|status"
} > "$OUT/f16x2_ladder.txt" 2>&1
# 7. every stand-alone kernel (unchanged set) for the per-op table
timeout 900 bash "$REPO/tools/profile_ops.sh" > "$OUT/ops.log" 2>&1
ls -la "$OUT"
