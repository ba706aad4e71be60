#!/usr/bin/env python3
"""Copy what tools/profile_r04.sh left under gpurun_out/ into profiles/r04_* (run in the dev container after the gpurun call;
gpurun_out/ is scratch, profiles/ is committed)."""
import glob, json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
R = 'gpurun_out/r04'


def last_json(path):
    return json.loads([ln for ln in open(path).read().splitlines() if ln.startswith('{')][-1])


def clean(text):
    keep = []
    for ln in text.splitlines():
        if re.match(r'^[EWI]20\d\d', ln) or 'warning:' in ln or re.match(r'^\s+\d+ \|', ln) or re.match(r'^\s+\|\s+\^', ln) or 'warnings generated' in ln:
            continue
        keep.append(ln)
    return '\n'.join(keep)


for src, dst in (('bench.json', 'r04_bench.json'), ('bench_driver_flags.json', 'r04_bench_driver_flags.json')):
    json.dump(last_json(os.path.join(R, src)), open(os.path.join('profiles', dst), 'w'), indent=1)
shutil.copy('gpurun_out/prof_r04/summary.txt', 'profiles/r04_bench_rocprofv3_summary.txt')
shutil.copy(glob.glob('gpurun_out/prof_r04/trace/**/*kernel_stats.csv', recursive=True)[0], 'profiles/r04_bench_kernel_stats.csv')
heads = {
    'config3.log': ('r04_config3_darts_step_kernel_stats.txt', '# tools/profile_r04.sh: BASELINE config 3 - DARTS search step, batch 32, 256 x 256, n_step 3 (5-slot super-net), ONE stream in the trace; split-precision arithmetic (the default)'),
    'config5.log': ('r04_config5_test_split_kernel_stats.txt', '# tools/profile_r04.sh: BASELINE config 5 - 4000 x 3000 frame, 63 tiles of 512 / 480, Bayer_01_Demosaic_02_sRGB_13; first lines: wall time with the default two tile streams at tile batches 21 / 16 / 63; then the kernel trace on ONE stream (RISP_TILE_STREAMS=1, tile batch 21)'),
    'small_batch.log': ('r04_small_batch_search_step.txt', '# tools/profile_r04.sh (profile_darts.sh r04_b4 4 2 10 + step_launches.py + trace_by_grid.py): the search step at the per-rank batch of the 8-GPU configuration (4 images, n_step 2)'),
    'batch32_nstep2.log': ('r04_batch32_search_step.txt', "# tools/profile_r04.sh (profile_darts.sh r04_b32 32 2 3): config 4's network at the global batch of 32 on one GPU"),
    'f32_arith_same_box.log': ('r04_f32_arith_same_box.txt', '# tools/profile_r04.sh: RISP_CONV_ARITH=f32 (the fp32 matrix-core kernels of round 3) on the same box, wall time: config 3, the rank-of-8 shard, config 5'),
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
            '%.1f busy cycles per matrix instruction; LDS: %.0f %% of its active cycles stalled on bank conflicts; '
            'GRBM_GUI_ACTIVE / 8 / %.1f us = %.2f GHz under the profiler (profiled runs are serialised and slower than back-to-back launches)\n'
            % (what, pmc, busy, float(vals['SQ_VALU_MFMA_BUSY_CYCLES']) / float(vals['SQ_INSTS_MFMA']),
               100 * float(vals['SQ_LDS_BANK_CONFLICT']) / max(float(vals['SQ_LDS_IDX_ACTIVE']), 1), us, clk))


open('profiles/r04_conv_pmc.txt', 'w').write(
    '# tools/conv_pmc.sh through tools/profile_r04.sh, MI355X: counters of single convolution layers on 32 x 256 x 256 (tools/conv_bench.py)\n'
    + pmc_block(R + '/conv_pmc.txt', '3x3 64->64, conv_f16x2_kernel<3, 2> (split precision on the f16 matrix pipe, the default)')
    + pmc_block(R + '/conv_pmc_f32.txt', '3x3 64->64, conv_wino43_b2_kernel (RISP_CONV_ARITH=f32: F(4,3) on the fp32 matrix pipe)')
    + pmc_block(R + '/conv_pmc_5x5.txt', '5x5 64->32, conv_f16x2_kernel<5, 1>'))
open('profiles/r04_f16x2_ladder.txt', 'w').write(
    '# tools/profile_r04.sh section 6 (tools/ab_f16x2.py): risp_conv2d_f16x2 beside the fp32 kernels of the product library on the same data - error\n'
    '# against the float64 convolution (rms / max of max|y|), interleaved timing rounds in one process; then in-kernel stamps of a wave\'s life\n'
    + clean(open(R + '/f16x2_ladder.txt').read()) + '\n')
if os.path.exists('gpurun_out/ops_r02/summary.txt'):
    shutil.copy('gpurun_out/ops_r02/summary.txt', 'profiles/r04_ops_kernel_stats.txt')
print('profiles/r04_* refreshed')
