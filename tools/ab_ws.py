#!/usr/bin/env python3
"""GPU box: risp_conv2d_f16x2 3x3 / 5x5 (env RISP_AB_K), wave-specialised kernel (risp_conv2d_f16x2) against the round-4 form (risp_conv2d_f16x2_uniform), interleaved rounds in ONE
process on the same tensors; plain / residual + ReLU / mask / residual + mask epilogues.  python tools/ab_ws.py [n h w [cin cout]]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reconfigisp_amd import lib as L, convnets as CN
a = [int(v) for v in sys.argv[1:]]
n, h, w = (a + [32, 256, 256])[:3] if len(a) >= 3 else (32, 256, 256)
cin, cout = (a[3], a[4]) if len(a) >= 5 else (64, 64)
K = int(os.environ.get('RISP_AB_K', '3'))
torch.manual_seed(0)
wt = torch.randn(cout, cin, K, K, device='cuda') * 0.05
b = torch.randn(cout, device='cuda') * 0.01
x = torch.rand(n, cin, h, w, device='cuda')
add, mask = torch.rand(n, cout, h, w, device='cuda'), torch.randn(n, cout, h, w, device='cuda')
y = torch.empty(n, cout, h, w, device='cuda')
pack = CN.f16x2_weights(wt, False)
lib = L.load()
ENTRY = {0: 'risp_conv2d_f16x2_uniform', 1: 'risp_conv2d_f16x2'}
def desc(epi):
    return L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=K, load_mode=0, cin_img=0, epilogue=epi, add_c=cout if epi & CN.EPI_ADD else 0,
                      x=x.data_ptr(), wpack=pack.data_ptr(), bias=b.data_ptr(), cvals=None, add=add.data_ptr() if epi & CN.EPI_ADD else None,
                      mask=mask.data_ptr() if epi & CN.EPI_MASK else None, y=y.data_ptr())
flop = 3 * 2.0 * cin * cout * K * K * n * h * w
for what, epi in (('plain', 0), ('relu', CN.EPI_RELU), ('add+relu', CN.EPI_ADD | CN.EPI_RELU), ('mask', CN.EPI_MASK), ('add+mask', CN.EPI_ADD | CN.EPI_MASK)):
    d = desc(epi)
    outs, res = {}, {0: [], 1: []}
    for v in (0, 1):
        y.fill_(float('nan'))
        L.call(ENTRY[v], C.byref(d), None)
        torch.cuda.synchronize()
        outs[v] = y.clone()
    for rnd in range(5):
        for v in (0, 1):
            for _ in range(2):
                L.call(ENTRY[v], C.byref(d), None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10):
                L.call(ENTRY[v], C.byref(d), None)
            e1.record(); e1.synchronize()
            res[v].append(e0.elapsed_time(e1) / 10 * 1e3)
    m0, m1 = sorted(res[0])[2], sorted(res[1])[2]
    print('%-9s %dx%d %dx%dx%dx%d->%d  round-4 %.1f us (min %.1f)  wave-specialised %.1f us (min %.1f)  x%.3f  %.0f TFLOP/s issued (%.2f of 2516.6)  same bits %s nan %d'
          % (what, K, K, n, cin, h, w, cout, m0, min(res[0]), m1, min(res[1]), m0 / m1, flop / m1 / 1e6, flop / m1 / 1e6 / 2516.6,
             bool(torch.equal(outs[0], outs[1])), int(torch.isnan(outs[1]).sum())))
