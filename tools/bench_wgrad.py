#!/usr/bin/env python3
"""GPU box: risp_conv2d_wgrad on the three layers of SRCNNRes (srcnn_res_arch.py:18-22) at 32 x 256 x 256, run-to-run bits, error against float64
on one image.  python tools/bench_wgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF
from reconfigisp_amd import convnets as CN
n, h, w = 32, 256, 256
torch.manual_seed(0)
for cin, cout, k, load in ((17, 64, 9, 'const'), (64, 32, 5, 'plain'), (32, 3, 5, 'plain')):
    gy = torch.randn(n, cout, h, w, device='cuda') * 1e-4
    if load == 'const':
        x, cv = torch.rand(n, 3, h, w, device='cuda'), torch.rand(n, cin - 3, device='cuda')
        run = lambda: CN.conv_wgrad(x, gy, cin, cout, k, n, h, w, load=CN.LOAD_CONSTCH, cin_img=3, cvals=cv)
        full = torch.cat([x[:1], cv[:1, :, None, None].expand(-1, -1, h, w)], dim=1)
    else:
        x = torch.rand(n, cin, h, w, device='cuda')
        run = lambda: CN.conv_wgrad(x, gy, cin, cout, k, n, h, w)
        full = x[:1]
    a, b = run()[0], run()[0]
    wd = torch.zeros(cout, cin, k, k, device='cuda', dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(TF.conv2d(full.double(), wd, None, padding=k // 2), wd, gy[:1].double())
    if load == 'const':
        one = CN.conv_wgrad(x[:1].contiguous(), gy[:1].contiguous(), cin, cout, k, 1, h, w, load=CN.LOAD_CONSTCH, cin_img=3, cvals=cv[:1].contiguous())[0]
    else:
        one = CN.conv_wgrad(x[:1].contiguous(), gy[:1].contiguous(), cin, cout, k, 1, h, w)[0]
    err = ((one.double() - ref).abs().max() / ref.abs().max()).item()
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    flop = 2.0 * cin * cout * k * k * n * h * w
    print('dW %2d -> %2d %dx%d: %.0f us (with the bias sums), %.1f TFLOP/s of the layer, same bits twice %s, max error of one image vs float64 %.1e'
          % (cin, cout, k, k, us, flop / us / 1e6, bool(torch.equal(a, b)), err))
