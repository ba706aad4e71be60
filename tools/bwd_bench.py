#!/usr/bin/env python3
"""GPU box: backward-data 3x3 64->64 with mask (+ residual add) epilogue: F(2,3) vs F(4,3) vs direct."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconfigisp_amd import convnets as CN
n, h, w = (int(v) for v in (sys.argv[1:4] + ['32', '256', '256'][len(sys.argv) - 1:]))
wt = torch.randn(64, 64, 3, 3, device='cuda') * 0.05
pc = CN.PackedConv(wt, torch.zeros(64, device='cuda'))
gy, add, mask = (torch.randn(n, 64, h, w, device='cuda') for _ in range(3))
out = torch.empty_like(gy)
def run(epi, **kw):
    for _ in range(3): CN.conv(gy, pc, n, h, w, transpose=True, epi=epi, out=out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): CN.conv(gy, pc, n, h, w, transpose=True, epi=epi, out=out, **kw)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 20
keep = pc.wino43_bwd
for name in ('F(4,3)', 'F(2,3)'):
    pc.wino43_bwd = keep if name == 'F(4,3)' else None
    res = []
    for rnd in range(3):
        res.append((run(CN.EPI_MASK, mask=mask), run(CN.EPI_MASK | CN.EPI_ADD, mask=mask, add=add, add_c=64), run(0)))
    print('%s: mask %.3f ms | mask+add %.3f ms | plain %.3f ms' % ((name,) + tuple(sorted(r[i] for r in res)[1] for i in range(3))))
