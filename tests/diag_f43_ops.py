#!/usr/bin/env python3
"""Diagnostic script (not collected by pytest; it lives under tests/ because it uses the test helpers and,
through them, the oracle).  GPU box: per-op forward / input-gradient difference between the F(2,3) and F(4,3) 3x3 kernels at 16x16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import conftest  # noqa: F401  (puts oracle/ on the path)
import torch
import test_host_logic as H
from reconfigisp_amd import convnets as CN
import reconfigisp_amd.functional as F
net = H.build_supernet(2, 'cuda')
torch.manual_seed(0)
if os.environ.get('RISP_DIAG_POSITIVE') == '1':      # no ReLU ever clips: kernels must then agree to rounding
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 4 or p.dim() == 1:
                p.abs_()
def run(mod, x, par):
    x = x.clone().requires_grad_(True)
    y = mod(x, par)
    gy = torch.ones_like(y) * 0.5 + 0.1 * torch.sin(torch.arange(y.numel(), device='cuda').float()).view_as(y)
    g, = torch.autograd.grad(y, x, gy)
    return y.detach(), g
for s, (mods, names) in enumerate(zip(net.all_modules, net.slot_names)):
    for mod, name in zip(mods, names):
        if 'path' not in name:
            continue
        if os.environ.get('RISP_DIAG_POSITIVE') == '1':
            with torch.no_grad():
                for p in mod.parameters():
                    p.abs_()
        cin = 1 if s == 0 else 3
        for hw in ((16, 16), (8, 8), (16, 32), (32, 128), (32, 256)):
            x = torch.rand(2, cin, *hw, device='cuda')
            res = []
            for f43 in (False, True):
                CN.WINO_F43 = f43
                for m in mod.modules():
                    m.__dict__.pop('_risp_pack_cache', None)
                res.append(run(mod, x, None))
            print('slot %d %-12s %s  fwd diff %.2e  grad diff %.2e (grad scale %.2e)' % (
                s, name, hw, (res[0][0] - res[1][0]).abs().max().item(), (res[0][1] - res[1][1]).abs().max().item(),
                res[0][1].abs().max().item()))
