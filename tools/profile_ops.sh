#!/bin/bash
# GPU box: kernel durations (rocprofv3 kernel trace) of the stand-alone ISP kernels that tools/bench_ops.py calls,
# with their algorithmic HBM rates at 64 x 256 x 256.  -> gpurun_out/ops_<tag>/summary.txt
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ops_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RISP_OPS_REPS=24
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o o -- python3 "$REPO/tools/bench_ops.py" > "$OUT/wall.log" 2>&1
# the grouped convolution layers of the SRCNNRes family, one by one (32 = 8 members x 4 images)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/layers" -o l -- python3 "$REPO/tools/bench_layers.py" 4 256 256 > "$OUT/layers.log" 2>&1
# instruction counts of the classical stencils: the compute-side floor = vector instructions x 4 cycles / 1024 SIMDs / clock
RISP_OPS_ONLY=origin rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$OUT/pmc" -o p -- python3 "$REPO/tools/bench_ops.py" > "$OUT/pmc.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
PIX = 64 * 256 * 256
# kernel-name fragment -> (label, algorithmic bytes per pixel of ONE launch)
ALG = [('origin_demosaic_kernel<false>', 'demosaic bilinear', 16), ('origin_demosaic_kernel<true>', 'demosaic Malvar-He-Cutler', 16),
       ('demosaic4_kernel<false>', 'demosaic bilinear', 16), ('demosaic4_kernel<true>', 'demosaic Malvar-He-Cutler', 16),
       ('bilateral4_kernel<1>', 'bilateral 3x3', 24), ('bilateral4_kernel<2>', 'bilateral 5x5', 24), ('bilateral_kernel', 'bilateral (general form)', 24),
       ('median3x4_kernel', 'median 3x3', 24), ('median4_kernel<5>', 'median 5x5', 24), ('median4_kernel<7>', 'median 7x7', 24), ('median4_kernel<9>', 'median 9x9', 24), ('median_kernel', 'median 11x11 (general form)', 24),
       ('fastnlm4_kernel<true>', 'fast-NLM 3/3', 24), ('fastnlm4_kernel<false>', 'fast-NLM (general)', 24), ('fastnlm_kernel', 'fast-NLM (general form)', 24),
       ('tonemap_kernel<0>', 'Reinhard curve', 24), ('tonemap_kernel<1>', 'Crysis curve', 24), ('tonemap_kernel<2>', 'filmic curve', 24),
       ('tonemap_kernel<3>', 'white-world gain', 24), ('loglum_kernel', 'log-average luminance', 12),
       ('stats_partial_kernel', 'channel statistics', 12), ('histc_kernel', 'histogram 3 x 32 bins', 12),
       ('bgr_fwd_kernel<risp_ops::WbManualCtx>', 'WbManual forward', 24), ('bgr_fwd_kernel<risp_ops::GammaCtx>', 'Gamma forward', 24),
       ('bgr_fwd_kernel<risp_ops::GtmCtx>', 'GtmManual forward', 24), ('bgr_fwd_kernel<risp_ops::WbqCtx>', 'WbQuadratic forward', 24),
       ('bgr_fwd_kernel<risp_ops::Gain3Ctx>', 'per-image gain (gray-world)', 24),
       ('bgr_bwd_kernel<risp_ops::WbManualCtx>', 'WbManual backward', 36), ('bgr_bwd_kernel<risp_ops::GammaCtx>', 'Gamma backward', 36),
       ('bgr_bwd_kernel<risp_ops::GtmCtx>', 'GtmManual backward', 36), ('bgr_bwd_kernel<risp_ops::WbqCtx>', 'WbQuadratic backward', 36),
       ('param_finish_kernel', 'parameter-gradient finish', 0), ('chain_kernel', 'nearest demosaic (chain of 1)', 16),
       ('mix_fwd_kernel', 'mixture forward, 8 operands', 108), ('mix_bwd_kernel', 'mixture backward, 8 operands', 204), ('mix_finish_kernel', 'mixture alpha-gradient finish', 0),
       ('slot_mix_fwd_kernel', 'fused slot mixture forward (9 tensors + 6 element-wise)', 132), ('slot_mix_bwd_kernel', 'fused slot mixture backward', 252),
       ('slot_wbq_params_kernel', 'fused slot mixture backward: the 30 WbQuadratic sums (second launch)', 24), ('slot_mix_finish_kernel', 'fused slot mixture finish', 0)]
f = glob.glob(os.path.join(root, 'prof', '**', '*kernel_stats.csv'), recursive=True)
with open(os.path.join(root, 'summary.txt'), 'w') as out:
    out.write('# tools/profile_ops.sh: rocprofv3 kernel durations at 64 x 256 x 256 (4.19 MPix); rate = algorithmic bytes (tensors read + written once) / duration\n')
    out.write('%-64s %6s %10s %6s %10s %8s\n' % ('kernel', 'calls', 'avg_us', 'B/pix', 'GB/s', 'of 8TB/s'))
    for r in csv.DictReader(open(f[0])):
        name = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')
        hit = [a for a in ALG if name.startswith(a[0])]
        if not hit:
            continue
        us = float(r['AverageNs']) / 1e3
        bpp = hit[0][2]
        rate = bpp * PIX / us / 1e3
        out.write('%-64s %6s %10.2f %6d %10s %8s\n' % ((hit[0][1] + '  [' + name.split('(')[0] + ']')[:64], r['Calls'], us, bpp,
                                                    '%.0f' % rate if bpp else '-', '%.2f' % (rate / 8000.) if bpp else '-'))
    # ---- grouped convolution layers (tools/bench_layers.py): kernel duration against the matrix / packed-FMA peak and HBM
    LPIX = 32 * 256 * 256
    LAY = [('conv_lin_kernel<9, 3', '9x9 3->64 forward (linear-k MFMA)', 2 * 244 * 64, 4 * (3 / 8 + 64)),
           ('conv_wino45_r2_kernel', '5x5 F(4,5): 64->32 fwd, 32->64 bwd, 3->32 bwd (mean of the three)', 2 * 10 * (64 * 32 + 32 * 64 + 4 * 32) / 3, 4 * (96 + 160 + 67) / 3),
           ('conv_wino5_glds_kernel', '5x5 64->32 fwd / 32->64 bwd F(2,5)', 2 * 15 * 64 * 32, 4 * 128),
           ('conv_small_kernel<5, 2', '5x5 32->3 forward, packed FMA', 2 * 25 * 32 * 4, 4 * 35.4),
           ('conv_wino5_kernel', '5x5 3->32 backward-data F(2,5)', 2 * 15 * 4 * 32, 4 * 67),
           ('conv_small_kernel<9, 2', '9x9 64->3 backward-data, packed FMA', 2 * 81 * 64 * 4, 4 * 70)]
    f2 = glob.glob(os.path.join(root, 'layers', '**', '*kernel_stats.csv'), recursive=True)
    if f2:
        out.write('\n# grouped SRCNNRes layers, 8 members x 4 images of 256 x 256 per launch (tools/bench_layers.py); FLOPs = what the launch ISSUES\n')
        out.write('# (MFMA layers: padded channel pairs, Winograd taps; direct layers: cout padded to 4) against the 157.3 TFLOP/s fp32 peak\n')
        out.write('%-64s %6s %10s %9s %8s %8s\n' % ('layer', 'calls', 'avg_us', 'TFLOP/s', 'of peak', 'of 8TB/s'))
        for r in csv.DictReader(open(f2[0])):
            name = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')
            hit = [a for a in LAY if name.startswith(a[0])]
            if hit:
                us = float(r['AverageNs']) / 1e3
                fl = hit[0][2] * LPIX / (us * 1e-6)
                out.write('%-64s %6s %10.2f %9.1f %8.3f %8.3f\n' % ((hit[0][1] + ' [' + name.split('(')[0] + ']')[:64], r['Calls'], us, fl / 1e12,
                                                                  fl / 157.3e12, hit[0][3] * LPIX / (us * 1e-6) / 8e12))
    # ---- instruction counts of the classical stencils -> compute-side floor
    f3 = glob.glob(os.path.join(root, 'pmc', '**', '*counter_collection.csv'), recursive=True)
    if f3:
        from collections import defaultdict
        acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(int)
        for r in csv.DictReader(open(f3[0])):
            k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'SQ_INSTS_VALU':
                cnt[k] += 1
        out.write('\n# vector / LDS / scalar instructions per launch (rocprofv3 --pmc, wave-instructions) and the issue floor they imply:\n')
        out.write('# floor_us = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz (every vector instruction holds its SIMD 4 cycles; transcendentals 8)\n')
        out.write('%-40s %8s %14s %14s %14s %10s %12s\n' % ('kernel', 'launches', 'VALU/launch', 'LDS/launch', 'SALU/launch', 'VALU/pixel', 'floor_us'))
        for k in sorted(acc, key=lambda k: -acc[k]['SQ_INSTS_VALU']):
            if not cnt[k] or 'origin' not in k and 'bilateral' not in k and 'median' not in k and 'fastnlm' not in k and 'demosaic' not in k and 'tonemap' not in k and 'loglum' not in k:
                continue
            v = acc[k]['SQ_INSTS_VALU'] / cnt[k]
            out.write('%-40s %8d %14.0f %14.0f %14.0f %10.1f %12.1f\n' % (k[:40], cnt[k], v, acc[k]['SQ_INSTS_LDS'] / cnt[k],
                                                                          acc[k]['SQ_INSTS_SALU'] / cnt[k], v * 64 / PIX, v * 4 / 1024 / 2.4e3))
print(open(os.path.join(root, 'summary.txt')).read())
PY
