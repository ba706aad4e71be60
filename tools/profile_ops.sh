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
       ('mix_fwd_kernel', 'mixture forward, 8 operands', 108), ('mix_bwd_kernel', 'mixture backward, 8 operands', 204), ('mix_finish_kernel', 'mixture alpha-gradient finish', 0)]
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
print(open(os.path.join(root, 'summary.txt')).read())
PY
