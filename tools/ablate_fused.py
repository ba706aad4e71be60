#!/usr/bin/env python3
"""GPU box: ablation timings of risp_bilateral_chain_fwd (which part of the fused segment costs what)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reconfigisp_amd.functional as F
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
n = 64
bay = make_batch(n, 256, 256, seed=10)[0].cuda()
sig = lambda v: torch.sigmoid(torch.tensor(v)).repeat(n, 1).cuda()
pw, pg, pt = sig([-1.38] * 3) * 5, sig([0.]), sig([-1.099, 0., 1.099])
sc = torch.full((n,), 50.5).cuda(); ss = torch.full((n,), 50.5).cuda()
def run(name, from_bayer, win, ops, params, x):
    w = torch.full((n,), win, dtype=torch.int32).cuda()
    plan = F.BilateralChainPlan(x, from_bayer, w, sc, ss, 3, ops, params)
    for _ in range(5): plan.launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): plan.launch()
    e1.record(); e1.synchronize()
    print('%-46s %.1f us' % (name, e0.elapsed_time(e1) * 10))
full_ops = [F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL]
run('full: bayer, 3x3, wb+gamma+gtm', True, 3, full_ops, [pw, pg, pt], bay)
run('no taps (window 1), chain kept', True, 1, full_ops, [pw, pg, pt], bay)
run('3x3, no chain', True, 3, [], [], bay)
run('no taps, no chain (staging + 2 outputs)', True, 1, [], [], bay)
run('3x3, chain = wb only', True, 3, [F.OP_WB_MANUAL], [pw], bay)
run('3x3, chain = wb+gamma', True, 3, [F.OP_WB_MANUAL, F.OP_GAMMA], [pw, pg], bay)
bgr = torch.rand(n, 3, 256, 256).cuda()
run('BGR input, 3x3, no chain', False, 3, [], [], bgr)
