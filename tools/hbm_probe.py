#!/usr/bin/env python3
"""GPU box: reference points for a WRITE-dominated stream on this device (torch fill / copy on buffers that do
not fit the 256 MiB Infinity Cache), next to the fused ISP kernels run over rotating buffer sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reconfigisp_amd.functional as F
from reconfigisp_amd.codes.data.synthetic_raw import make_batch

def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

for gib in (1, 4, 16):
    big = torch.empty(gib << 28, device='cuda')
    t = timeit(lambda: big.fill_(1.0), 10); print('fill %2d GiB           : %.2f TB/s written' % (gib, big.numel() * 4 / t / 1e12))
    del big
big = torch.empty(1 << 28, device='cuda')            # 1 GiB fp32
src = torch.rand(1 << 28, device='cuda')
t = timeit(lambda: big.copy_(src)); print('copy 1 GiB -> 1 GiB   : %.2f TB/s (read+write)' % (2 * big.numel() * 4 / t / 1e12))
del big, src
small = torch.empty(48 << 20, device='cuda')          # 192 MiB: fits the Infinity Cache
t = timeit(lambda: small.fill_(1.0), 50); print('fill 192 MiB (cached) : %.2f TB/s written' % (small.numel() * 4 / t / 1e12))

n = 64
sig = lambda v: torch.sigmoid(torch.tensor(v)).repeat(n, 1).cuda()
pw, pg, pt = sig([-1.38] * 3) * 5, sig([0.]), sig([-1.099, 0., 1.099])
sc = torch.full((n,), 50.5).cuda(); ss = torch.full((n,), 50.5).cuda(); w = torch.full((n,), 3, dtype=torch.int32).cuda()
for sets in (1, 2, 4, 8, 16, 32):
    bays = [make_batch(n, 256, 256, seed=10 + k)[0].cuda() for k in range(sets)]
    fused = [F.BilateralChainPlan(b, True, w, sc, ss, 3, [F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL], [pw, pg, pt]) for b in bays]
    chain = [F.ChainPlan(b, [F.OP_DEMOSAIC_NEAREST, F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL], [None, pw, pg, pt]) for b in bays]
    def run(plans):
        for p in plans: p.launch()
    tf = timeit(lambda: run(fused), 30) / sets
    tc = timeit(lambda: run(chain), 30) / sets
    print('%d buffer set(s): fused ISP %.1f us = %.2f TB/s (64 B/pix) | point-wise chain %.1f us = %.2f TB/s (52 B/pix)'
          % (sets, tf * 1e6, 64 * n * 65536 / tf / 1e12, tc * 1e6, 52 * n * 65536 / tc / 1e12))
