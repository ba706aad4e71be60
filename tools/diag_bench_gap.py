#!/usr/bin/env python3
"""GPU box: why does bench.py's kernel_time_ms differ from tools/ab_fused.py?  Times the same fused launch
(a) through plan.launch() and (b) through a bare ctypes call, with 4 rotating batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reconfigisp_amd.functional as F
from reconfigisp_amd import lib as L
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
n = 64
sig = lambda v: torch.sigmoid(torch.tensor(v)).repeat(n, 1).cuda()
pw, pg, pt = sig([-1.38] * 3) * 5, sig([0.]), sig([-1.099, 0., 1.099])
sc = torch.full((n,), 50.5).cuda(); ss = torch.full((n,), 50.5).cuda(); w = torch.full((n,), 3, dtype=torch.int32).cuda()
plans = [F.BilateralChainPlan(make_batch(n, 256, 256, seed=100 + k)[0].cuda(), True, w, sc, ss, 3,
                              [F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL], [pw, pg, pt]) for k in range(4)]
lib = L.load()
fn = lib.risp_bilateral_chain_fwd

def timeit(f, reps):
    for k in range(8): f(k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
    for k in range(reps): f(k)
    t1 = time.perf_counter()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, (t1 - t0) / reps * 1e6

import ctypes as C, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
        '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared']
subprocess.check_call(base + ['-o', '/tmp/f_only.so', os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_fused.hip'),
                              os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_core.cpp')])
l2 = C.CDLL('/tmp/f_only.so')
l2.risp_bilateral_chain_fwd.restype, l2.risp_bilateral_chain_fwd.argtypes = L.SIGNATURES['risp_bilateral_chain_fwd']
fn2 = l2.risp_bilateral_chain_fwd
for rnd in range(4):
    b = timeit(lambda k: fn(*plans[k % 4]._args, None), 200)
    c = timeit(lambda k: fn2(*plans[k % 4]._args, None), 200)
    print('round %d: full library %.1f us | fused-only library built on the box %.1f us' % (rnd, b[0], c[0]))
for reps in (200,):
    a = timeit(lambda k: plans[k % 4].launch(), reps)
    b = timeit(lambda k: fn(*plans[k % 4]._args, None), reps)
    s = torch.cuda.current_stream().cuda_stream
    c = timeit(lambda k: fn(*plans[k % 4]._args, s), reps)
    print('reps %4d: plan.launch() %.1f us (host issue %.1f us/launch) | bare ctypes, null stream %.1f us (host %.1f) | bare ctypes, torch stream %.1f us (host %.1f)'
          % (reps, a[0], a[1], b[0], b[1], c[0], c[1]))
