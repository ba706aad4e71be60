#!/usr/bin/env python3
"""GPU box: in-process A/B of two builds of risp_conv.hip (default flags vs EXTRA_DEFS), interleaved rounds."""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
defs = sys.argv[1:] or ['-DRISP_CONV_STAGGER']
base = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
        '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared']
src = [os.path.join(ROOT, 'reconfigisp_amd/csrc', f) for f in ('risp_conv.hip', 'risp_core.cpp')]
subprocess.check_call(base + ['-o', '/tmp/conv_a.so'] + src)
subprocess.check_call(base + defs + ['-o', '/tmp/conv_b.so'] + src)
import torch
from reconfigisp_amd import lib as L
from reconfigisp_amd import convnets as CN
libs = {'base': C.CDLL('/tmp/conv_a.so'), ' '.join(defs): C.CDLL('/tmp/conv_b.so')}
for l in libs.values():
    l.risp_conv2d.restype = C.c_int
    l.risp_conv2d.argtypes = [C.POINTER(L.ConvDesc), C.c_void_p]
cfgs = [(64, 64, 3, 64, 128, 128), (64, 64, 3, 16, 256, 256), (64, 32, 5, 64, 128, 128), (17, 64, 9, 16, 256, 256)]
for cin, cout, k, n, h, w in cfgs:
    wt = torch.randn(cout, cin, k, k, device='cuda') * 0.05
    pc = CN.PackedConv(wt, torch.zeros(cout, device='cuda'))
    x = torch.rand(n, cin, h, w, device='cuda'); y = torch.empty(n, cout, h, w, device='cuda')
    d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=k, load_mode=0, cin_img=0, epilogue=1, add_c=0, x=x.data_ptr(),
                   wpack=pc.fwd.data_ptr(), bias=pc.bias.data_ptr(), cvals=None, add=None, mask=None, y=y.data_ptr())
    res = {k_: [] for k_ in libs}
    for rnd in range(5):
        for name, l in libs.items():
            for _ in range(2): l.risp_conv2d(C.byref(d), None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): l.risp_conv2d(C.byref(d), None)
            e1.record(); e1.synchronize()
            res[name].append(e0.elapsed_time(e1) / 10)
    flop = 2.0 * cin * cout * k * k * n * h * w
    for name, v in res.items():
        m = sorted(v)[len(v) // 2]
        print('%dx%d k%d N%d %dx%d  %-24s median %.3f ms  %.1f TFLOP/s' % (cin, cout, k, n, h, w, name, m, flop / m / 1e9))
