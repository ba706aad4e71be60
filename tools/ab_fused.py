#!/usr/bin/env python3
"""GPU box: in-process A/B of builds of risp_fused.hip with different -D flags (interleaved rounds)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = sys.argv[1:] or ['', '-DRISP_FUSED_WAVES=6', '-DRISP_FUSED_WAVES=8']
base = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
        '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared']
core = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_core.cpp')
default_src = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_fused.hip')
import torch  # before dlopen: the library must bind to the HIP runtime PyTorch already loaded
libs = {}
for i, v in enumerate(variants):
    so = '/tmp/fused_%d.so' % i
    # a variant is "-Dflags" (current source) or "path/to/other_fused.hip[,-Dflags]" (another source file)
    parts = v.split(',') if v else []
    srcf = default_src
    if parts and parts[0].endswith('.hip'):
        srcf, parts = os.path.join(ROOT, parts[0]), parts[1:]
    subprocess.check_call(base + parts + ['-o', so, srcf, core])
    libs[v or 'base'] = C.CDLL(so)
import torch
from reconfigisp_amd import lib as L
import reconfigisp_amd.functional as F
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
n = 64
bay = make_batch(n, 256, 256, seed=10)[0].cuda()
sig = lambda v: torch.sigmoid(torch.tensor(v)).repeat(n, 1).cuda()
pw, pg, pt = sig([-1.38] * 3) * 5, sig([0.]), sig([-1.099, 0., 1.099])
sc = torch.full((n,), 50.5).cuda(); ss = torch.full((n,), 50.5).cuda()
w = torch.full((n,), 3, dtype=torch.int32).cuda()
NSETS = int(os.environ.get('RISP_AB_SETS', '4'))
plans = [F.BilateralChainPlan(make_batch(n, 256, 256, seed=10 + k)[0].cuda(), True, w, sc, ss, 3,
                              [F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL], [pw, pg, pt]) for k in range(NSETS)]
plan = plans[0]
sig_ = L.SIGNATURES['risp_bilateral_chain_fwd']
for l in libs.values():
    l.risp_bilateral_chain_fwd.restype, l.risp_bilateral_chain_fwd.argtypes = sig_
for name, l in libs.items():
    st = l.risp_bilateral_chain_fwd(*plan._args, None)
    l.risp_last_error.restype = C.c_char_p
    print(name, 'status', st, l.risp_last_error())
outs0 = None
for name, l in libs.items():        # all variants must agree on the 8-bit codes (<= 1 code on a tiny fraction)
    l.risp_bilateral_chain_fwd(*plan._args, None)
    torch.cuda.synchronize()
    cur = [t.clone() for t in plan.outs]
    if outs0 is None:
        outs0 = cur
    else:
        for k, (a_, b_) in enumerate(zip(outs0, cur)):
            d = (a_ - b_).abs()
            print('  %s stage %d: max diff %.3g, differing %.4f %%' % (name, k, d.max().item(), 100. * (d > 1e-6).float().mean().item()))
REPS = int(os.environ.get('RISP_AB_REPS', '48'))      # 48 = short bursts (boost clocks); 2000+ = sustained load
for nsets in (1, NSETS):
    res = {k: [] for k in libs}
    for rnd in range(7):
        for name, l in libs.items():
            for k in range(4): l.risp_bilateral_chain_fwd(*plans[k % nsets]._args, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for k in range(REPS): l.risp_bilateral_chain_fwd(*plans[k % nsets]._args, None)
            e1.record(); e1.synchronize()
            res[name].append(e0.elapsed_time(e1) / REPS * 1e3)
    for k, v in res.items():
        print('%d set(s) %-40s median %.1f us  min %.1f' % (nsets, k, sorted(v)[len(v) // 2], min(v)))
