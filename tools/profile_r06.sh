#!/bin/bash
# GPU box: everything round 6 commits under profiles/ in one call.  -> gpurun_out/r06/*     (sections can be picked: tools/profile_r06.sh 3 6)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r06
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
WANT=" ${*:-1 2 3 4 5 6 7 8 9} "
want() { [[ "$WANT" == *" $1 "* ]]; }
if want 1; then   # the bench line: default window, then the round driver's flags
  timeout 900 python3 "$REPO/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
  timeout 900 python3 "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> "$OUT/bench_driver_flags.err"
fi
if want 2; then   # kernel trace + HBM counters of the bench command (-> gpurun_out/prof_r06; tools/collect_r06.py writes profiles/traffic.json from it)
  timeout 1200 bash "$REPO/tools/profile_bench.sh" r06 > "$OUT/profile_bench.log" 2>&1
fi
if want 3; then   # search step: BASELINE config 3, the per-rank batch of config 4, the shipped geometry of the search YAMLs (batch 4 of 48 x 48, n_step 3)
  timeout 900 bash "$REPO/tools/profile_darts.sh" r06_c3 32 3 2 > "$OUT/config3.log" 2>&1
  python3 "$REPO/tools/step_launches.py" "$REPO/gpurun_out/darts_r06_c3/prof/d_kernel_trace.csv" 5 >> "$OUT/config3.log" 2>&1
  python3 "$REPO/tools/trace_by_grid.py" "$REPO/gpurun_out/darts_r06_c3/prof/d_kernel_trace.csv" >> "$OUT/config3.log" 2>&1
  timeout 900 bash "$REPO/tools/profile_darts.sh" r06_b4 4 2 10 > "$OUT/small_batch.log" 2>&1
  python3 "$REPO/tools/step_launches.py" "$REPO/gpurun_out/darts_r06_b4/prof/d_kernel_trace.csv" 4 >> "$OUT/small_batch.log" 2>&1
  timeout 900 bash "$REPO/tools/profile_darts.sh" r06_ship 4 3 20 48 > "$OUT/shipped_geometry.log" 2>&1
  python3 "$REPO/tools/step_launches.py" "$REPO/gpurun_out/darts_r06_ship/prof/d_kernel_trace.csv" 5 >> "$OUT/shipped_geometry.log" 2>&1
  python3 "$REPO/tools/host_profile_darts.py" 4 48 3 2>&1 | tail -40 >> "$OUT/shipped_geometry.log"
  timeout 600 bash "$REPO/tools/profile_darts.sh" r06_b32 32 2 3 > "$OUT/batch32_nstep2.log" 2>&1
  # ... the round-5 few-channel kernels (Toeplitz bands) against this round's on the same box, alternating: config 3 and the rank-of-8 shard
  OLD=$(bash "$REPO/tools/build_variant.sh" /tmp/r06_xwin_off "-DRISP_XWIN_OFF" risp_conv_toep_first.hip)
  python3 "$REPO/tools/bench_darts.py" 32 256 3 1 > /dev/null 2>&1          # (warm the box: the first process after an idle period measures slow)
  { for r in 1 2 3; do
      echo -n "round 6 kernels        : "; python3 "$REPO/tools/bench_darts.py" 32 256 3 2 2>&1 | tail -1
      echo -n "round 5 kernels        : "; RISP_BENCH_NO_TAPOUT=1 RISP_BENCH_NO_THIN5=1 RISP_HIP_LIBRARY=$OLD python3 "$REPO/tools/bench_darts.py" 32 256 3 2 2>&1 | tail -1
    done
    for r in 1 2 3; do
      echo -n "round 6 kernels        : "; python3 "$REPO/tools/bench_darts.py" 32 256 3 2 2>&1 | tail -1
      echo -n "... without thin5      : "; RISP_BENCH_NO_THIN5=1 python3 "$REPO/tools/bench_darts.py" 32 256 3 2 2>&1 | tail -1
    done
    for r in 1 2; do
      echo -n "round 6 kernels        : "; python3 "$REPO/tools/bench_darts.py" 4 256 2 8 2>&1 | tail -1
      echo -n "round 5 kernels        : "; RISP_BENCH_NO_TAPOUT=1 RISP_BENCH_NO_THIN5=1 RISP_HIP_LIBRARY=$OLD python3 "$REPO/tools/bench_darts.py" 4 256 2 8 2>&1 | tail -1
    done; } > "$OUT/few_channel_same_box.log" 2>&1
  # ... and the fp32 arithmetic on the same box, wall time only
  { RISP_CONV_ARITH=f32 python3 "$REPO/tools/bench_darts.py" 32 256 3 2 2>&1 | tail -1; RISP_CONV_ARITH=f32 python3 "$REPO/tools/bench_darts.py" 4 256 2 8 2>&1 | tail -1; RISP_CONV_ARITH=f32 python3 "$REPO/tools/bench_split.py" 16 2>&1 | tail -1; } > "$OUT/f32_arith_same_box.log" 2>&1
fi
if want 4; then   # config 5: wall time with the default two tile streams, kernel trace on one stream
  for b in 16 21 32 63; do python3 "$REPO/tools/bench_split.py" $b 2>&1 | tail -1; done > "$OUT/config5.log"
  RISP_TILE_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5" -o s -- python3 "$REPO/tools/bench_split.py" 16 > "$OUT/cfg5_prof.log" 2>&1
  tail -1 "$OUT/cfg5_prof.log" >> "$OUT/config5.log"
  python3 "$REPO/tools/trace_by_grid.py" "$OUT/cfg5/s_kernel_trace.csv" 200 >> "$OUT/config5.log" 2>&1
fi
if want 5; then   # counters: the 64 -> 64 3x3 and 5x5 64 -> 32 layers (wave-specialised split precision), then this round's few-channel kernels with their HBM bytes
  timeout 600 bash "$REPO/tools/conv_pmc.sh" r06 64 64 3 32 256 256 > "$OUT/conv_pmc.txt" 2>&1
  timeout 600 bash "$REPO/tools/conv_pmc.sh" r06_5x5 64 32 5 32 256 256 > "$OUT/conv_pmc_5x5.txt" 2>&1
  export RISP_PMC_PROG=few_channel_bench.py RISP_PMC_HBM=1
  RISP_PMC_KERNELS=conv_xwin timeout 600 bash "$REPO/tools/conv_pmc.sh" r06_first first > "$OUT/conv_pmc_first.txt" 2>&1
  RISP_PMC_KERNELS=conv_tapout timeout 600 bash "$REPO/tools/conv_pmc.sh" r06_bwd9 bwd9 > "$OUT/conv_pmc_bwd9.txt" 2>&1
  RISP_PMC_KERNELS=conv_tapout timeout 600 bash "$REPO/tools/conv_pmc.sh" r06_fwd5 fwd5 > "$OUT/conv_pmc_fwd5.txt" 2>&1
  RISP_PMC_KERNELS=conv_thin5 timeout 600 bash "$REPO/tools/conv_pmc.sh" r06_bwd5 bwd5 > "$OUT/conv_pmc_bwd5.txt" 2>&1
  unset RISP_PMC_PROG RISP_PMC_HBM
fi
if want 6; then   # this round's few-channel kernels against the Toeplitz-band kernels they replace (interleaved rounds, one process), and where their waves' time goes
  {
    python3 "$REPO/tools/ab_tapout.py" 2>&1 | grep "band"
    python3 "$REPO/tools/ab_tapout.py" 4 256 256 2>&1 | grep "band"
    python3 "$REPO/tools/ab_tapout.py" 4 48 48 2>&1 | grep "band"
    python3 "$REPO/tools/ab_xwin.py" 2>&1 | grep "band"
    python3 "$REPO/tools/ab_xwin.py" 4 256 256 2>&1 | grep "band"
    python3 "$REPO/tools/ab_xwin.py" 4 48 48 2>&1 | grep "band"
    for r in 1 2; do python3 "$REPO/tools/few_channel_bench.py" bwd5 2>&1 | tail -1; RISP_FCB_WINO=1 python3 "$REPO/tools/few_channel_bench.py" bwd5 2>&1 | tail -1 | sed 's/^bwd5/bwd5 (fp32 F(4,5) kernel)/'; done
    python3 "$REPO/tools/tapout_stamps.py" 2>&1 | tail -12
    python3 "$REPO/tools/xwin_stamps.py" 2>&1 | tail -12
  } > "$OUT/few_channel_ladder.txt" 2>&1
  # the 3x3 / 5x5 wave-specialised kernel against the uniform kernel of round 4 (now two entry points)
  {
    python3 "$REPO/tools/ab_ws.py" 2>&1 | grep "round-4"
    python3 "$REPO/tools/ab_ws.py" 4 256 256 2>&1 | grep "round-4"
    RISP_AB_K=5 python3 "$REPO/tools/ab_ws.py" 32 256 256 64 32 2>&1 | grep "round-4"
  } > "$OUT/ws_ladder.txt" 2>&1
fi
if want 7; then   # weight gradients and the proxy fine-tuning step
  { python3 "$REPO/tools/bench_wgrad.py" 2>&1 | tail -3; python3 "$REPO/tools/bench_ft.py" 2>&1 | tail -1; } > "$OUT/wgrad_ft.txt" 2>&1
fi
if want 8; then   # every stand-alone kernel for the per-op table
  timeout 900 bash "$REPO/tools/profile_ops.sh" r06 > "$OUT/ops.log" 2>&1
fi
if want 9; then   # the slot mixture's backward: WbQuadratic's sums inside the launch (default) against a second launch, streaming stores, batched loads (builds in /tmp)
  timeout 900 bash "$REPO/tools/ab_slot_onepass.sh" "-DRISP_SLOT_WBQ_ONE_PASS=0" "-DRISP_SLOT_NT=1" "-DSLOT_TG=4" > "$OUT/slot_one_pass.txt" 2>&1
fi
# gpurun copies back at most 64 MiB: the raw traces and counter dumps stay on the box, the summaries made from them travel
find "$REPO/gpurun_out" -name "*.csv" -size +256k -delete
du -sh "$REPO/gpurun_out"
ls -la "$OUT"
