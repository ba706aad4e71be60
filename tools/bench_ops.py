#!/usr/bin/env python3
"""GPU box: call time (Python wrapper included: calls shorter than ~40 us are host-bound - the kernel durations come
from tools/profile_ops.sh) and HBM rate of every stand-alone ISP kernel (classical "Origin" stencils and tone curves,
differentiable point-wise ops forward and backward, reductions, the slot mixture) at BASELINE config 2's shape
(64 x 256 x 256).  Algorithmic bytes = tensors read + written once; rate against the 8 TB/s HBM peak.
python tools/bench_ops.py [N H W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reconfigisp_amd.functional as F
from reconfigisp_amd.codes.data.synthetic_raw import make_batch

n, h, w = (int(v) for v in (sys.argv[1:4] + ['64', '256', '256'][len(sys.argv) - 1:]))
pix = n * h * w
SETS = 4                                           # rotate over several resident inputs: no cache-assisted reads
bay = [make_batch(n, h, w, seed=10 + k)[0].cuda() for k in range(SETS)]
bgr = [F.demosaic_nearest(b) for b in bay]
gy = [torch.rand_like(b) for b in bgr]
sig = lambda v: torch.sigmoid(torch.tensor(v, dtype=torch.float32)).repeat(n, 1).cuda()
ones = lambda v: torch.full((n,), float(v)).cuda()


REPS = int(os.environ.get('RISP_OPS_REPS', '0'))   # short runs for a rocprofv3 kernel trace (tools/profile_ops.sh)
ONLY = os.environ.get('RISP_OPS_ONLY', '')               # run only the cases whose name contains this


def timed(fn, reps=100):
    reps = REPS or reps
    for k in range(8):
        fn(k % SETS)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        fn(k % SETS)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


rows = []


def case(name, bytes_per_pix, fn, reps=100):
    if ONLY and ONLY not in name:
        return
    us = timed(fn, reps)
    gbs = bytes_per_pix * pix / us / 1e3
    rows.append((name, bytes_per_pix, us, gbs))
    print('%-44s %4d B/pix %9.1f us %8.1f GB/s  %.2f of 8 TB/s' % (name, bytes_per_pix, us, gbs, gbs / 8000.), flush=True)


print('== %d x %d x %d (%.2f MPix), launches rotate over %d resident inputs' % (n, h, w, pix / 1e6, SETS))
# ---- classical operators (tools_origin.py:445-804), [0,1] in and out (scales (255, 255): the fused inference form)
S = (255.0, 255.0)
case('origin demosaic bilinear', 16, lambda k: F.origin_demosaic(bay[k], 'bilinear', S))
case('origin demosaic laplacian (MHC)', 16, lambda k: F.origin_demosaic(bay[k], 'laplacian', S))
bp = dict(window_length=torch.full((n,), 3, dtype=torch.int32).cuda(), sigma_color=ones(50.5), sigma_space=ones(50.5), max_window=3)
case('origin bilateral 3x3', 24, lambda k: F.origin_denoise(bgr[k], 'bilateral', bp, S))
bp5 = dict(bp, window_length=torch.full((n,), 5, dtype=torch.int32).cuda(), max_window=5)
case('origin bilateral 5x5', 24, lambda k: F.origin_denoise(bgr[k], 'bilateral', bp5, S))
case('origin median 3x3', 24, lambda k: F.origin_denoise(bgr[k], 'median', dict(size=3), S))
case('origin median 5x5', 24, lambda k: F.origin_denoise(bgr[k], 'median', dict(size=5), S), 50)
case('origin median 9x9 (default)', 24, lambda k: F.origin_denoise(bgr[k], 'median', dict(size=9), S), 30)
case('origin median 11x11 (general form)', 24, lambda k: F.origin_denoise(bgr[k], 'median', dict(size=11), S), 10)
nl = dict(block_size=torch.full((n,), 3, dtype=torch.int32).cuda(), search_block=torch.full((n,), 3, dtype=torch.int32).cuda(),
          decay_factor=ones(50.5), max_block=3, max_search=3)
case('origin fast-NLM 3/3', 24, lambda k: F.origin_denoise(bgr[k], 'fastnlm', nl, S))
case('origin reinhard (stats pass + curve)', 36, lambda k: F.origin_tonemap(bgr[k], 'reinhard', dict(white_point=ones(0.8), middle_grey=ones(0.5)), S))
case('origin crysis', 24, lambda k: F.origin_tonemap(bgr[k], 'crysisengine', dict(lum_adapted=ones(0.5)), S))
case('origin filmic', 24, lambda k: F.origin_tonemap(bgr[k], 'filmic', dict(white_point=ones(0.8), exposure_bias=ones(5.5)), S))
case('origin white-world (stats pass + gain)', 36, lambda k: F.origin_whiteworld(bgr[k], ones(0.9), S))
# ---- differentiable point-wise ops, forward (x in, y out) and backward (x, gy in; gx out; parameter gradients)
pw, pg, pt, pq = sig([-1.38] * 3), sig([0.]), sig([-1.099, 0., 1.099]), sig([0.1 * (i % 7 - 3) for i in range(30)])
for name, fn, p in (('wb_manual', F.wb_manual, pw), ('gamma', F.gamma, pg), ('gtm_manual', F.gtm_manual, pt),
                    ('wb_quadratic', F.wb_quadratic, pq)):
    with torch.no_grad():
        case(name + ' forward', 24, lambda k, fn=fn, p=p: fn(bgr[k], p))
    xs = [b.clone().requires_grad_(True) for b in bgr]
    pr = p.clone().requires_grad_(True)
    def bwd(k, fn=fn, xs=xs, pr=pr):
        y = fn(xs[k], pr)
        torch.autograd.grad(y, (xs[k], pr), gy[k])
    case(name + ' forward + backward', 24 + 36, bwd)
with torch.no_grad():
    case('demosaic_nearest forward', 16, lambda k: F.demosaic_nearest(bay[k]))
    case('grayworld forward (stats pass + gain)', 36, lambda k: F.grayworld(bgr[k]))
    case('channel_stats (min / sum / max / arg)', 12, lambda k: F.channel_stats(bgr[k]))
    case('histc 3 x 32 bins', 12, lambda k: F.hist_features(bgr[k], 32))
    K = 8
    outs = [[torch.rand_like(bgr[0]) for _ in range(K)] for _ in range(2)]
    wmix = torch.softmax(torch.zeros(K), 0).cuda()
    case('mix forward, 8 operands', 12 * (K + 1), lambda k: F.mix(wmix, outs[k % 2], [1.0 / K] * K), 50)
wm = wmix.clone().requires_grad_(True)
oo = [[o.clone().requires_grad_(True) for o in outs[s]] for s in range(2)]
def mixb(k):
    y = F.mix(wm, oo[k % 2], [1.0 / K] * K)
    torch.autograd.grad(y, [wm] + oo[k % 2], gy[k])
case('mix forward + backward, 8 operands', 12 * (K + 1) + 12 * (2 * K + 1), mixb, 50)

# ---- the mixture of an sRGB slot with its element-wise operators evaluated in the kernel (functional.slot_mix): 9 tensor
# operands (the CNN proxies) + gamma, gray world, skip, manual white balance, quadratic white balance, tone curve
T9 = [[torch.rand_like(bgr[0]) for _ in range(9)] for _ in range(2)]
order = ['gamma', 'T', 'T', 'T', 'grayworld', 'T', 'T', 'T', 'T', 'skip', 'wb_manual', 'T', 'wb_quadratic', 'gtm_manual', 'T']
w15 = torch.softmax(torch.zeros(15), 0).cuda()
blocks = {'gamma': pg, 'wb_manual': pw, 'wb_quadratic': pq, 'gtm_manual': pt}


def entries(k, ts, grad=False):
    it = iter(ts)
    return [('tensor', next(it)) if o == 'T' else ('op', o, blocks.get(o)) for o in order]


with torch.no_grad():
    # algorithmic bytes: x + 9 tensors read, y written
    case('slot mixture fused forward, 9 tensors + 6 element-wise', 12 * 11, lambda k: F.slot_mix(w15, bgr[k], entries(k, T9[k % 2]), [1 / 15.] * 15), 50)
wq = w15.clone().requires_grad_(True)
xs = [b.clone().requires_grad_(True) for b in bgr]
tt = [[t.clone().requires_grad_(True) for t in T9[s]] for s in range(2)]
def slotb(k):
    y = F.slot_mix(wq, xs[k], entries(k, tt[k % 2]), [1 / 15.] * 15)
    torch.autograd.grad(y, [wq, xs[k]] + tt[k % 2], gy[k])
# forward as above; backward: gy, x, 9 tensors read; 9 tensor gradients + gx written
case('slot mixture fused forward + backward', 12 * 11 + 12 * (2 + 9 + 9 + 1), slotb, 50)
