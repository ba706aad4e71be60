#!/usr/bin/env python3
"""Copy what tools/profile_r05.sh left under gpurun_out/ into profiles/r05_* (run in the dev container after the gpurun call;
gpurun_out/ is scratch, profiles/ is committed) and REGENERATE profiles/traffic.json from this round's FETCH_SIZE / WRITE_SIZE passes
(bench.py reports roofline.traffic only while the file's `kernel` is the instance it launches)."""
import csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
R, TAG = 'gpurun_out/r05', 'r05'


def last_json(path):
    return json.loads([ln for ln in open(path).read().splitlines() if ln.startswith('{')][-1])


def clean(text):
    keep = []
    for ln in text.splitlines():
        if re.match(r'^[EWI]20\d\d', ln) or 'warning:' in ln or re.match(r'^\s+\d+ \|', ln) or re.match(r'^\s+\|\s+\^', ln) or 'warnings generated' in ln \
                or 'amdgpu.ids' in ln:
            continue
        keep.append(ln)
    return '\n'.join(keep)


bench = last_json(os.path.join(R, 'bench.json'))
for src, dst in (('bench.json', TAG + '_bench.json'), ('bench_driver_flags.json', TAG + '_bench_driver_flags.json')):
    json.dump(last_json(os.path.join(R, src)), open(os.path.join('profiles', dst), 'w'), indent=1)
shutil.copy('gpurun_out/prof_%s/summary.txt' % TAG, 'profiles/%s_bench_rocprofv3_summary.txt' % TAG)
shutil.copy(glob.glob('gpurun_out/prof_%s/trace/**/*kernel_stats.csv' % TAG, recursive=True)[0], 'profiles/%s_bench_kernel_stats.csv' % TAG)

# ---- traffic.json from the counter passes of THIS round
sys.path.insert(0, ROOT)
from reconfigisp_amd import lib as L          # noqa: E402  (the name query touches no GPU)
kernel = L.load().risp_bilateral_chain_kernel(1, 3, 0).decode()           # what a bench step launches, as rocprofv3 prints it


def per_launch(sub, counter, name):
    """average raw counter value (KiB) per launch of the kernel whose rocprofv3 name - spaces aside - starts with `name`, from the
    summary tools/summarize_prof.py wrote on the GPU box (the raw counter dumps are too large to travel)"""
    want, section = name.replace(' ', ''), None
    for ln in open('gpurun_out/prof_%s/summary.txt' % TAG):
        if ln.startswith('== '):
            section = ln.split()[1]
        elif section == counter and ln.replace(' ', '').replace('void(anonymousnamespace)::', '').startswith(want):
            m = re.search(r'launches\s+(\d+)\s+avg\s+([\d.]+) KiB', ln)
            return float(m.group(2)), int(m.group(1))
    raise SystemExit('no %s row for %s in gpurun_out/prof_%s/summary.txt' % (counter, name, TAG))


fetch, nf = per_launch('pmc_fetch', 'FETCH_SIZE', kernel)
write, nw = per_launch('pmc_write', 'WRITE_SIZE', kernel)
pw_fetch, _ = per_launch('pmc_fetch', 'FETCH_SIZE', 'chain_kernel<2,false>')
pw_write, _ = per_launch('pmc_write', 'WRITE_SIZE', 'chain_kernel<2,false>')
old = json.load(open('profiles/traffic.json'))
alg = 64 * 64 * 256 * 256                       # bench.py: BYTES_PER_PIX_ISP x the pixels of one launch (batch 64 of 256 x 256)
traffic = {
    '_comment': 'HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of round 5 (tools/profile_bench.sh r05 through tools/profile_r05.sh: separate '
                '--pmc FETCH_SIZE and --pmc WRITE_SIZE runs of `bench.py --steps 200 --warmup 20 --no-cpu --no-cnn`; summary in profiles/r05_bench_rocprofv3_summary.txt), '
                'written by tools/collect_r05.py.  Counter units are KiB.  WRITE_SIZE is exact for 16-byte-per-lane stores (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is '
                'multiplied by 2 (calibrated on this chip with tools/fetch_calib.hip: factor 2.000 for 4-, 8- and 16-byte loads and for the two-row quad pattern the kernel '
                'stages its mosaic with, profiles/r02_fetch_size_calibration.txt).  `kernel` is what risp_bilateral_chain_kernel() names for the bench launch: bench.py '
                'reports these bytes only while that is the kernel it launches.',
    'round': TAG, 'kernel': kernel, 'launches_averaged': [nf, nw],
    'fetch_size_kib_raw': round(fetch, 1), 'fetch_size_factor_calibrated': 2.0, 'write_size_kib_raw': round(write, 1),
    'traffic_bytes_per_launch': int(round((2.0 * fetch + write) * 1024)), 'algorithmic_bytes_per_launch': alg,
    'before_xcd_aware_order': old.get('before_xcd_aware_order'),
    'pointwise_kernel': {'kernel': 'chain_kernel<2,false>', 'fetch_size_kib_raw': round(pw_fetch, 1), 'write_size_kib_raw': round(pw_write, 1),
                         'traffic_bytes_per_launch': int(round((2.0 * pw_fetch + pw_write) * 1024)), 'algorithmic_bytes_per_launch': 52 * 64 * 256 * 256},
}
json.dump(traffic, open('profiles/traffic.json', 'w'), indent=1)
print('traffic.json: %s, %.1f MB per launch = %.3f x algorithmic' % (kernel, traffic['traffic_bytes_per_launch'] / 1e6, traffic['traffic_bytes_per_launch'] / alg))

heads = {
    'config3.log': (TAG + '_config3_darts_step_kernel_stats.txt', '# tools/profile_r05.sh: BASELINE config 3 - DARTS search step, batch 32, 256 x 256, n_step 3 (5-slot super-net), ONE stream in the trace; round-5 defaults (wave-specialised split-precision kernels, split-precision first layers in training)'),
    'config5.log': (TAG + '_config5_test_split_kernel_stats.txt', '# tools/profile_r05.sh: BASELINE config 5 - 4000 x 3000 frame, 63 tiles of 512 / 480, Bayer_01_Demosaic_02_sRGB_13; first lines: wall time with the default two tile streams at tile batches 16 / 21 / 32 / 63; then the kernel trace on ONE stream (RISP_TILE_STREAMS=1, tile batch 16)'),
    'small_batch.log': (TAG + '_small_batch_search_step.txt', '# tools/profile_r05.sh (profile_darts.sh r05_b4 4 2 10 + step_launches.py): the search step at the per-rank batch of the 8-GPU configuration (4 images, n_step 2)'),
    'batch32_nstep2.log': (TAG + '_batch32_search_step.txt', "# tools/profile_r05.sh (profile_darts.sh r05_b32 32 2 3): config 4's network at the global batch of 32 on one GPU"),
    'f32_arith_same_box.log': (TAG + '_f32_arith_same_box.txt', '# tools/profile_r05.sh: RISP_CONV_ARITH=f32 (the fp32 matrix-core kernels) on the same box, wall time: config 3, the rank-of-8 shard, config 5'),
    'first_layer_infer_same_box.log': (TAG + '_first_layer_infer_same_box.txt', "# tools/profile_r05.sh: RISP_CONV_TOEP_FIRST=infer (round 4's default: training forwards of the 9x9 first layers on the fp32 kernel) on the same box: config 3, the rank-of-8 shard"),
    'ws_ladder.txt': (TAG + '_ws_ladder.txt', '# tools/profile_r05.sh section 6: risp_conv2d_f16x2, wave-specialised kernel (variant 1, the default) against the round-4 kernel (variant 0), interleaved rounds in one process\n# (tools/ab_ws.py: 3x3 64->64 at 32 / 21 / 4 images of 256 x 256, 5x5 64->32 and 32->64), then builds of the kernel against each other (tools/ab_ws_build.py: default = the epilogue of residual / mask launches rides on the matrix steps,\n# -DWS_SPLIT=0 = never, -DWS_SPLIT=2 = also without a residual or mask), then in-kernel stamps of the 3x3 kernel (tools/ws_stamps.py: -DRISP_WS_STAMPS; epilogue modes 0-3 = ReLU / residual + ReLU /\n# mask / residual + mask; the last two blocks: modes 1 and 3 with -DWS_SPLIT=0)'),
    'ws_conflicts.txt': (TAG + '_ws_conflicts.txt', '# tools/ws_conflicts.sh through tools/profile_r05.sh section 9: LDS bank conflicts of conv_f16x2_ws_kernel<3, 2> (3x3 64 -> 64, 32 x 256 x 256) per launch: the full kernel,\n# without the producers\' 4-byte halo writes, without any producer write (diagnostic builds, wrong results): ALL conflicts are in the halo writes'),
    'wbq_ablation.txt': (TAG + '_wbq_ablation.txt', "# tools/ab_wbq.sh through tools/profile_r05.sh section 9, 64 x 256 x 256: WbQuadratic's backward kernels as ablation builds (-DRISP_WBQ_ABL: 3 = the empty launch,\n# 2 = ONE vector per thread: launch + prologue + block reduction + partial row, 1 = the whole walk without the arithmetic (slot_wbq_params_kernel only), none = the product)"),
    'wgrad_ft.txt': (TAG + '_wgrad_ft.txt', '# tools/profile_r05.sh section 7: risp_conv2d_wgrad on the three layers of SRCNNRes (tools/bench_wgrad.py) and one finetune_proxies() call (tools/bench_ft.py)'),
}
for src, (dst, head) in heads.items():
    open(os.path.join('profiles', dst), 'w').write(head + '\n' + clean(open(os.path.join(R, src)).read()) + '\n')


def pmc_block(path, what):
    pmc = clean(open(path).read())
    vals = dict(re.findall(r'(\w+)\s+avg/launch\s+(\d+)', pmc))
    us = float(re.search(r'kernel avg us ([\d.]+)', pmc).group(1))
    clk = float(vals['GRBM_GUI_ACTIVE']) / 8 / us / 1e3
    busy = float(vals['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024 / (float(vals['GRBM_GUI_ACTIVE']) / 8)
    return ('## %s\n%s\n# derived: matrix pipes busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = %.3f; '
            '%.1f busy cycles per matrix instruction; LDS: %.1f %% of its active cycles stalled on bank conflicts; '
            'GRBM_GUI_ACTIVE / 8 / %.1f us = %.2f GHz under the profiler (profiled runs are serialised and slower than back-to-back launches)\n'
            % (what, pmc, busy, float(vals['SQ_VALU_MFMA_BUSY_CYCLES']) / float(vals['SQ_INSTS_MFMA']),
               100 * float(vals['SQ_LDS_BANK_CONFLICT']) / max(float(vals['SQ_LDS_IDX_ACTIVE']), 1), us, clk))


open('profiles/%s_conv_pmc.txt' % TAG, 'w').write(
    '# tools/conv_pmc.sh through tools/profile_r05.sh, MI355X: counters of single convolution layers on 32 x 256 x 256 (tools/conv_bench.py)\n'
    + pmc_block(R + '/conv_pmc.txt', '3x3 64->64, conv_f16x2_ws_kernel<3, 2> (wave-specialised split precision on the f16 matrix pipe, the default)')
    + pmc_block(R + '/conv_pmc_5x5.txt', '5x5 64->32, conv_f16x2_ws_kernel<5, 1>'))
if os.path.exists('gpurun_out/ops_r05/summary.txt'):
    shutil.copy('gpurun_out/ops_r05/summary.txt', 'profiles/%s_ops_kernel_stats.txt' % TAG)
print('profiles/%s_* refreshed' % TAG)
