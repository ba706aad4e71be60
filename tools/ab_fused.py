#!/usr/bin/env python3
"""GPU box: in-process A/B of builds of risp_fused.hip with different -D flags (interleaved rounds)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = sys.argv[1:] or ['', '-DRISP_FUSED_WAVES=6', '-DRISP_FUSED_WAVES=8']
base = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off',
        '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared']
src = [os.path.join(ROOT, 'reconfigisp_amd/csrc', f) for f in ('risp_fused.hip', 'risp_core.cpp')]
import torch  # before dlopen: the library must bind to the HIP runtime PyTorch already loaded
libs = {}
for i, v in enumerate(variants):
    so = '/tmp/fused_%d.so' % i
    subprocess.check_call(base + ([v] if v else []) + ['-o', so] + src)
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
plan = F.BilateralChainPlan(bay, True, w, sc, ss, 3, [F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL], [pw, pg, pt])
sig_ = L.SIGNATURES['risp_bilateral_chain_fwd']
for l in libs.values():
    l.risp_bilateral_chain_fwd.restype, l.risp_bilateral_chain_fwd.argtypes = sig_
for name, l in libs.items():
    st = l.risp_bilateral_chain_fwd(*plan._args, None)
    l.risp_last_error.restype = C.c_char_p
    print(name, 'status', st, l.risp_last_error())
res = {k: [] for k in libs}
for rnd in range(7):
    for name, l in libs.items():
        for _ in range(3): l.risp_bilateral_chain_fwd(*plan._args, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): l.risp_bilateral_chain_fwd(*plan._args, None)
        e1.record(); e1.synchronize()
        res[name].append(e0.elapsed_time(e1) * 20)
for k, v in res.items():
    print('%-28s median %.1f us  min %.1f' % (k, sorted(v)[len(v) // 2], min(v)))
