#!/usr/bin/env python3
"""Micro-benchmark of one convolution layer (GPU box): python tools/conv_bench.py [cin cout k N H W reps]
The kernel is whatever convnets.route() picks (RISP_CONV_ARITH=f32: the fp32 kernels); RISP_BENCH_GRAD=1 times a training launch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reconfigisp_amd import convnets as CN

cin, cout, k, n, h, w, reps = (int(v) for v in (sys.argv[1:8] + ['64', '64', '3', '64', '128', '128', '20'][len(sys.argv) - 1:]))
torch.manual_seed(0)
dev = torch.device('cuda')
wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
b = torch.randn(cout, device=dev) * 0.01
pc = CN.PackedConv(wt, b)
x = torch.rand(n, cin, h, w, device=dev)
INFER = os.environ.get('RISP_BENCH_GRAD') != '1'
y = CN.conv(x, pc, n, h, w, epi=CN.EPI_RELU, infer=INFER)
ref = torch.relu(torch.nn.functional.conv2d(x[:2], wt, b, padding=k // 2))
err = (y[:2] - ref).abs().max().item()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    CN.conv(x, pc, n, h, w, epi=CN.EPI_RELU, out=y, infer=INFER)
e0.record()
for _ in range(reps):
    CN.conv(x, pc, n, h, w, epi=CN.EPI_RELU, out=y, infer=INFER)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flop = 2.0 * cin * cout * k * k * n * h * w
print('conv %dx%d k%d N%d %dx%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)  max|err| vs torch %.2e'
      % (cin, cout, k, n, h, w, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100, err))
