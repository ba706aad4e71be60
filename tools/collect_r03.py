#!/usr/bin/env python3
"""Copy what tools/profile_r03.sh and tools/profile_ops.sh left under gpurun_out/ into profiles/r03_* (run in the dev
container after the gpurun call; gpurun_out/ is scratch, profiles/ is committed)."""
import glob, json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
R = 'gpurun_out/r03'


def last_json(path):
    return json.loads([ln for ln in open(path).read().splitlines() if ln.startswith('{')][-1])


def clean(text):
    return '\n'.join(ln for ln in text.splitlines() if not re.match(r'^[EWI]20\d\d', ln))


for src, dst in (('bench.json', 'r03_bench.json'), ('bench_driver_flags.json', 'r03_bench_driver_flags.json')):
    json.dump(last_json(os.path.join(R, src)), open(os.path.join('profiles', dst), 'w'), indent=1)
shutil.copy('gpurun_out/prof_r03/summary.txt', 'profiles/r03_bench_rocprofv3_summary.txt')
shutil.copy(glob.glob('gpurun_out/prof_r03/trace/**/*kernel_stats.csv', recursive=True)[0], 'profiles/r03_bench_kernel_stats.csv')
heads = {
    'config3.log': ('r03_config3_darts_step_kernel_stats.txt', '# tools/profile_r03.sh: BASELINE config 3 - DARTS search step, batch 32, 256 x 256, n_step 3 (5-slot super-net), ONE stream in the trace'),
    'config5.log': ('r03_config5_test_split_kernel_stats.txt', '# tools/profile_r03.sh: BASELINE config 5 - 4000 x 3000 frame, 63 tiles of 512 / 480, Bayer_01_Demosaic_02_sRGB_13; first line: wall time with the default two tile streams; then the kernel trace on ONE stream (RISP_TILE_STREAMS=1)'),
    'small_batch.log': ('r03_small_batch_search_step.txt', '# tools/profile_r03.sh (profile_darts.sh r03_b4 4 2 10 + step_launches.py + trace_by_grid.py): the search step at the per-rank batch of the 8-GPU configuration (4 images, n_step 2)'),
    'batch32_nstep2.log': ('r03_batch32_search_step.txt', "# tools/profile_r03.sh (profile_darts.sh r03_b32 32 2 3): config 4's network at the global batch of 32 on one GPU"),
}
for src, (dst, head) in heads.items():
    open(os.path.join('profiles', dst), 'w').write(head + '\n' + clean(open(os.path.join(R, src)).read()) + '\n')
pmc = clean(open(R + '/conv_pmc.txt').read())
vals = dict(re.findall(r'(\w+)\s+avg/launch\s+(\d+)', pmc))
us = re.search(r'kernel avg us ([\d.]+)', pmc)
busy = float(vals['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024 / (float(vals['GRBM_GUI_ACTIVE']) / 8)
open('profiles/r03_conv3x3_64to64_pmc.txt', 'w').write(
    '# tools/conv_pmc.sh r03 (conv_wino43_b2_kernel, 64->64 3x3 on 64 x 128 x 128, MI355X; profiled runs are serialised and slower than back-to-back launches).\n'
    "# Round 3's two-block kernel (DESIGN.md 4.3a'''): both cout blocks of the layer in one wave, 12 accumulator tiles, two workgroups per CU.\n" + pmc + '''
# derived: matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = %.3f (r02: 0.676, r01: 0.563); LDS bank conflicts %s;
#          the counter is 64 cycles x 9 437 184 MFMA instructions; GRBM_GUI_ACTIVE / 8 / %s us = %.2f GHz under the profiler.
#          Back to back in one process (tools/ab_wino43.py, round 3): 350 us on this shape (one-block kernel 364), 684 us on 32 x 256 x 256 (733)
#          = 110-113 issued TFLOP/s = 0.70-0.72 of the 157.3 TFLOP/s nameplate.  SQ_INSTS_VALU 41.0M (one-block kernel 59.5M): the input transform now
#          feeds twelve matrix instructions.  Why not more: profiles/r03_mfma_ceiling.txt - every vector instruction costs its 4-5 cycles and every
#          LDS-DMA piece ~60 cycles of matrix time (the fp32 matrix and vector pipes do not overlap), the store epilogue ~5-10 %%.
''' % (busy, vals.get('SQ_LDS_BANK_CONFLICT', '?'), us.group(1), float(vals['GRBM_GUI_ACTIVE']) / 8 / float(us.group(1)) / 1e3))
if os.path.exists('gpurun_out/ops_r03/summary.txt'):
    shutil.copy('gpurun_out/ops_r03/summary.txt', 'profiles/r03_ops_kernel_stats.txt')
print('profiles/r03_* refreshed')
