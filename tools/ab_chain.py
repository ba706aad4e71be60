#!/usr/bin/env python3
"""GPU box: in-process A/B of builds of risp_pointwise.hip (-D flags): the point-wise chain launch over 1 and 4
rotating buffer sets, and a producer -> consumer pair (demosaic launch, then a white-balance launch reading it)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = sys.argv[1:] or ['', '-DRISP_CHAIN_NT']
base = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
        '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared']
src = [os.path.join(ROOT, 'reconfigisp_amd/csrc', f) for f in ('risp_pointwise.hip', 'risp_core.cpp')]
import torch  # before dlopen: the library must bind to the HIP runtime PyTorch already loaded
libs = {}
for i, v in enumerate(variants):
    so = '/tmp/chain_%d.so' % i
    subprocess.check_call(base + (v.split(',') if v else []) + ['-o', so] + src)
    libs[v or 'base'] = C.CDLL(so)
from reconfigisp_amd import lib as L
import reconfigisp_amd.functional as F
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
n = 64
sig = lambda v: torch.sigmoid(torch.tensor(v)).repeat(n, 1).cuda()
pw, pg, pt = sig([-1.38] * 3) * 5, sig([0.]), sig([-1.099, 0., 1.099])
NSETS = 4
plans = [F.ChainPlan(make_batch(n, 256, 256, seed=10 + k)[0].cuda(),
                     [F.OP_DEMOSAIC_NEAREST, F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL], [None, pw, pg, pt]) for k in range(NSETS)]
dem = [F.ChainPlan(p.x, [F.OP_DEMOSAIC_NEAREST], [None]) for p in plans]
wbo = [torch.empty_like(d.outs[0]) for d in dem]
for l in libs.values():
    for f in ('risp_chain_fwd', 'risp_wb_manual_fwd'):
        getattr(l, f).restype, getattr(l, f).argtypes = L.SIGNATURES[f]

def timeit(fn, reps=48):
    for k in range(4): fn(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for k in range(reps): fn(k)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for nsets in (1, NSETS):
    res, res2 = {k: [] for k in libs}, {k: [] for k in libs}
    for rnd in range(7):
        for name, l in libs.items():
            res[name].append(timeit(lambda k: l.risp_chain_fwd(*plans[k % nsets]._args, None)))
            def pair(k):
                i = k % nsets
                l.risp_chain_fwd(*dem[i]._args, None)
                l.risp_wb_manual_fwd(dem[i].outs[0].data_ptr(), pw.data_ptr(), wbo[i].data_ptr(), n, 65536, None)
            res2[name].append(timeit(pair))
    for k in libs:
        print('%d set(s) %-20s chain median %.1f us | demosaic -> wb pair median %.1f us'
              % (nsets, k, sorted(res[k])[3], sorted(res2[k])[3]))
