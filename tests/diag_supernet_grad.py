#!/usr/bin/env python3
"""Diagnostic script (not collected by pytest; it lives under tests/ because it uses the test helpers and,
through them, the oracle).  GPU box: alpha / parameter gradient errors of the super-net golden under the three 3x3 kernels (ReLU-kink check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conftest import load_golden
import test_host_logic as H
g = load_golden('supernet_n2')
print('input', g['x'].shape)
net = H.build_supernet(2, 'cuda')
with torch.no_grad():
    for k, v in net.named_parameters():
        v.copy_(H.T(g['p_' + k]))
y = net(H.T(g['x']).cuda())
named = dict(net.named_parameters()); keys = sorted(named)
grads = torch.autograd.grad(y, [named[k] for k in keys], H.T(g['gy']).cuda(), allow_unused=True)
worst = []
for k, gr in zip(keys, grads):
    ref = H.T(g['g_' + k])
    gr = torch.zeros_like(named[k]).cpu() if gr is None else gr.cpu()
    scale = max(ref.abs().max().item(), 1e-30)
    worst.append(((gr - ref).abs().max().item() / scale, k))
worst.sort(reverse=True)
print(' | '.join('%s %.2e' % (k, e) for e, k in worst[:5]))
print('output max err %.2e' % (y.detach().cpu() - H.T(g['mid%d' % (len(net.intermediate_results) - 1)])).abs().max().item())
