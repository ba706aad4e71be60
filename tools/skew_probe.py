#!/usr/bin/env python3
"""GPU box: does the relative placement of the five stage-output buffers matter?  The fused ISP launch writes 15
planes concurrently; each buffer is exactly 48 MiB, so back-to-back allocations put the streams at power-of-two-ish
distances.  Carve the buffers out of one slab with an extra skew between consecutive buffers and time the launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reconfigisp_amd.functional as F
from reconfigisp_amd import lib as L
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
n, h, w = 64, 256, 256
sig = lambda v: torch.sigmoid(torch.tensor(v)).repeat(n, 1).cuda()
pw, pg, pt = sig([-1.38] * 3) * 5, sig([0.]), sig([-1.099, 0., 1.099])
sc = torch.full((n,), 50.5).cuda(); ss = torch.full((n,), 50.5).cuda(); win = torch.full((n,), 3, dtype=torch.int32).cuda()
ops = [F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL]
lib = L.load()
fn = lib.risp_bilateral_chain_fwd
NSETS = int(os.environ.get('RISP_SETS', '8'))
bays = [make_batch(n, h, w, seed=10 + k)[0].cuda() for k in range(NSETS)]
buf_bytes = n * 3 * h * w * 4
slab = torch.empty((NSETS * 5 * (buf_bytes + (8 << 20)) + (64 << 20)) // 4, device='cuda')

def make_args(skew, set_skew):
    out = []
    base = slab.data_ptr()
    base = (base + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    for s in range(NSETS):
        ptrs = [base + (s * 5 + k) * buf_bytes + k * skew + s * set_skew for k in range(5)]
        out.append((bays[s].data_ptr(), 1, ptrs[0], ptrs[1], win.data_ptr(), sc.data_ptr(), ss.data_ptr(), 3, 3,
                    (C.c_int * 3)(*ops), L.ptr_array([pw.data_ptr(), pg.data_ptr(), pt.data_ptr()]),
                    L.ptr_array(ptrs[2:]), n, h, w))
    return out

def timeit(args, reps=64):
    for k in range(NSETS): fn(*args[k % NSETS], None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for k in range(reps): fn(*args[k % NSETS], None)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

skews = [0, 256, 1024, 4096, 4096 + 256, 16384 + 256, 65536 + 4096 + 256, (1 << 20) + 65536 + 4096 + 256, 3 << 20]
res = {s: [] for s in skews}
for rnd in range(5):
    for s in skews:
        res[s].append(timeit(make_args(s, 5 * s)))
for s in skews:
    v = sorted(res[s])
    print('skew %8d B between stage buffers: median %.1f us  min %.1f  (%d rotating sets)' % (s, v[len(v) // 2], v[0], NSETS))
