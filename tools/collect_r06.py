#!/usr/bin/env python3
"""Copy what tools/profile_r06.sh left under gpurun_out/ into profiles/r06_* (run in the dev container after the gpurun call;
gpurun_out/ is scratch, profiles/ is committed) and REGENERATE profiles/traffic.json from this round's FETCH_SIZE / WRITE_SIZE passes
(bench.py reports roofline.traffic only while the file's `kernel` is the instance it launches)."""
import csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
R, TAG = 'gpurun_out/r06', 'r06'


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
    '_comment': 'HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of round 6 (tools/profile_bench.sh r06 through tools/profile_r06.sh: separate '
                '--pmc FETCH_SIZE and --pmc WRITE_SIZE runs of `bench.py --steps 200 --warmup 20 --no-cpu --no-cnn`; summary in profiles/r06_bench_rocprofv3_summary.txt), '
                'written by tools/collect_r06.py.  Counter units are KiB.  WRITE_SIZE is exact for 16-byte-per-lane stores (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is '
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
    'config3.log': (TAG + '_config3_darts_step_kernel_stats.txt', '# tools/profile_r06.sh: BASELINE config 3 - DARTS search step, batch 32, 256 x 256, n_step 3 (5-slot super-net), ONE stream in the trace; round-6 defaults (tap-row kernel on the 3-cout layers, tap-index kernel on the 9x9 first layers)'),
    'config5.log': (TAG + '_config5_test_split_kernel_stats.txt', '# tools/profile_r06.sh: BASELINE config 5 - 4000 x 3000 frame, 63 tiles of 512 / 480, Bayer_01_Demosaic_02_sRGB_13; first lines: wall time with the default two tile streams at tile batches 16 / 21 / 32 / 63; then the kernel trace on ONE stream (RISP_TILE_STREAMS=1, tile batch 16)'),
    'small_batch.log': (TAG + '_small_batch_search_step.txt', '# tools/profile_r06.sh (profile_darts.sh r06_b4 4 2 10 + step_launches.py): the search step at the per-rank batch of the 8-GPU configuration (4 images, n_step 2)'),
    'shipped_geometry.log': (TAG + '_shipped_geometry_search_step.txt', "# tools/profile_r06.sh (profile_darts.sh r06_ship 4 3 20 48 + step_launches.py + host_profile_darts.py): the search step at the geometry the reference's search YAMLs ship (batch 4 of 48 x 48, n_step 3): launches per iteration, kernel time, wall time, host profile"),
    'batch32_nstep2.log': (TAG + '_batch32_search_step.txt', "# tools/profile_r06.sh (profile_darts.sh r06_b32 32 2 3): config 4's network at the global batch of 32 on one GPU"),
    'few_channel_same_box.log': (TAG + '_few_channel_same_box.txt', "# tools/profile_r06.sh section 3: the search step with this round's few-channel kernels against round 5's Toeplitz-band kernels on the SAME box, alternating runs (RISP_BENCH_NO_TAPOUT=1 RISP_BENCH_NO_THIN5=1 + a -DRISP_XWIN_OFF build of the library in /tmp): config 3 (batch 32, n_step 3) against the round-5 kernels, then with and without risp_conv2d_thin5 alone (RISP_BENCH_NO_THIN5=1), then the rank-of-8 shard (batch 4, n_step 2)"),
    'f32_arith_same_box.log': (TAG + '_f32_arith_same_box.txt', '# tools/profile_r06.sh: RISP_CONV_ARITH=f32 (the fp32 matrix-core kernels) on the same box, wall time: config 3, the rank-of-8 shard, config 5'),
    'few_channel_ladder.txt': (TAG + '_few_channel_ladder.txt', "# tools/profile_r06.sh section 6: risp_conv2d_tapout (filter rows in the rows of the matrix instruction) and the tap-index first layer (risp_conv_xwin.hip) against the Toeplitz-band kernels they replace,\n# interleaved rounds in one process (tools/ab_tapout.py, tools/ab_xwin.py: grouped launches of 8 members x 32 / 4 images of 256 x 256 and 4 of 48 x 48), then in-kernel stamps (tools/tapout_stamps.py, tools/xwin_stamps.py)"),
    'ws_ladder.txt': (TAG + '_ws_ladder.txt', '# tools/profile_r06.sh section 6: risp_conv2d_f16x2 (wave-specialised) against risp_conv2d_f16x2_uniform (the round-4 kernel), interleaved rounds in one process (tools/ab_ws.py)'),
    'slot_one_pass.txt': (TAG + '_slot_one_pass.txt', "# tools/ab_slot_onepass.sh through tools/profile_r06.sh section 9, 64 x 256 x 256: the fused slot mixture's backward - default build (WbQuadratic's 30 sums inside the one launch), -DRISP_SLOT_WBQ_ONE_PASS=0 (round 5: a second launch),\n# -DRISP_SLOT_NT=1 (streaming stores), -DSLOT_TG=4 (the loads of four tensor operands issued together)"),
    'wgrad_ft.txt': (TAG + '_wgrad_ft.txt', '# tools/profile_r06.sh section 7: risp_conv2d_wgrad on the three layers of SRCNNRes (tools/bench_wgrad.py) and one finetune_proxies() call (tools/bench_ft.py)'),
}
for src, (dst, head) in heads.items():
    if not os.path.exists(os.path.join(R, src)):
        print('missing', src); continue
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


def hbm_block(path, what, alg_bytes):
    """counters of a few-channel kernel + its HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB) against the tensors it must move"""
    txt = pmc_block(path, what)
    vals = dict(re.findall(r'(\w+)\s+avg/launch\s+(\d+)', clean(open(path).read())))
    if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
        traffic = (2.0 * float(vals['FETCH_SIZE']) + float(vals['WRITE_SIZE'])) * 1024
        us = float(re.search(r'kernel avg us ([\d.]+)', txt).group(1))
        txt += ('# HBM: 2 x FETCH_SIZE + WRITE_SIZE = %.2f GB per launch = %.2f x the tensors in and out (%.2f GB); at %.0f us that is %.2f TB/s = %.2f of 8 TB/s\n'
                % (traffic / 1e9, traffic / alg_bytes, alg_bytes / 1e9, us, traffic / us / 1e6, traffic / us / 1e6 / 8.0))
    return txt


PIX = 8 * 32 * 256 * 256
open('profiles/%s_conv_pmc.txt' % TAG, 'w').write(
    '# tools/conv_pmc.sh through tools/profile_r06.sh, MI355X: counters of single convolution layers on 32 x 256 x 256 (tools/conv_bench.py) and of the grouped few-channel layers\n'
    '# (tools/few_channel_bench.py: 8 members x 32 images of 256 x 256)\n'
    + pmc_block(R + '/conv_pmc.txt', '3x3 64->64, conv_f16x2_ws_kernel<3, 2> (wave-specialised split precision on the f16 matrix pipe, the default)')
    + pmc_block(R + '/conv_pmc_5x5.txt', '5x5 64->32, conv_f16x2_ws_kernel<5, 1>')
    + hbm_block(R + '/conv_pmc_first.txt', '9x9 3->64 first layer, grouped 8 x 32: conv_xwin_kernel (risp_conv_xwin.hip)', (64.0 * PIX + 3 * PIX / 8) * 4)
    + hbm_block(R + '/conv_pmc_bwd9.txt', '9x9 64->3 backward-data + residual, grouped 8 x 32: conv_tapout_kernel<9, ...> (risp_conv_tapout.hip)', (64.0 + 6) * PIX * 4)
    + hbm_block(R + '/conv_pmc_fwd5.txt', '5x5 32->3 forward + residual, grouped 8 x 32: conv_tapout_kernel<5, ...>', (32.0 + 6) * PIX * 4)
    + hbm_block(R + '/conv_pmc_bwd5.txt', '5x5 3->32 backward-data with a ReLU mask, grouped 8 x 32: conv_thin5_kernel (risp_conv_thin5.hip)', (3.0 + 64) * PIX * 4))
if os.path.exists('gpurun_out/ops_r06/summary.txt'):
    shutil.copy('gpurun_out/ops_r06/summary.txt', 'profiles/%s_ops_kernel_stats.txt' % TAG)
print('profiles/%s_* refreshed' % TAG)
