#!/usr/bin/env python3
"""GPU box: in-process A/B of builds of risp_conv_wino.hip with different -D flags (interleaved rounds) on one
64 -> 64 3x3 layer.  python tools/ab_wino.py "" "-DRISP_WINO_STAGGER=0" ...   [env RISP_AB_SHAPE="n h w"]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = sys.argv[1:] or ['', '-DRISP_WINO_STAGGER=0']
base = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off',
        '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared']
core = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_core.cpp')
default_src = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_conv_wino.hip')
import torch
libs = {}
for i, v in enumerate(variants):
    so = '/tmp/wino_%d.so' % i
    parts = v.split(',') if v else []            # "-Dflags" or "other_source.hip[,-Dflags]"
    srcf = default_src
    if parts and parts[0].endswith('.hip'):
        srcf, parts = os.path.join(ROOT, parts[0]), parts[1:]
    subprocess.check_call(base + parts + ['-o', so, srcf, core])
    libs[v or 'base'] = C.CDLL(so)
from reconfigisp_amd import lib as L
from reconfigisp_amd import convnets as CN
n, h, w = (int(v) for v in os.environ.get('RISP_AB_SHAPE', '64 128 128').split())
cin = cout = 64
wt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
b = torch.randn(cout, device='cuda') * 0.01
x = torch.rand(n, cin, h, w, device='cuda')
y = torch.empty(n, cout, h, w, device='cuda')
ref = torch.relu(torch.nn.functional.conv2d(x[:2], wt, b, padding=1))
res = {k: [] for k in libs}
for name, l in libs.items():
    l.risp_conv2d_wino3.restype, l.risp_conv2d_wino3.argtypes = L.SIGNATURES['risp_conv2d_wino3']
    l.risp_conv_wino3_chunk.restype = C.c_int
    ck = l.risp_conv_wino3_chunk()
    # pack for this variant's chunk depth
    g0, g1, g2 = wt[..., 0], wt[..., 1], wt[..., 2]
    u = torch.stack([g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2], dim=-1)
    nch = (cin + ck - 1) // ck
    p = torch.zeros((nch * ck, 3, 4, 64), device='cuda')
    p[:cin, :, :, :cout] = u.permute(1, 2, 3, 0)
    pack = p.view(nch, ck, 3, 4, 64).permute(0, 2, 3, 1, 4).contiguous()
    d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=3, load_mode=0, cin_img=0, epilogue=CN.EPI_RELU, add_c=0,
                   x=x.data_ptr(), wpack=pack.data_ptr(), bias=b.data_ptr(), cvals=None, add=None, mask=None, y=y.data_ptr())
    libs[name] = (l, d, pack)
    st = l.risp_conv2d_wino3(C.byref(d), None)
    torch.cuda.synchronize()
    print('%-40s status %d max|err| %.2e' % (name, st, (y[:2] - ref).abs().max().item()))
for rnd in range(7):
    for name, (l, d, pack) in libs.items():
        for _ in range(2): l.risp_conv2d_wino3(C.byref(d), None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): l.risp_conv2d_wino3(C.byref(d), None)
        e1.record(); e1.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
flop = 2.0 * cin * cout * 9 * n * h * w
for k, v in res.items():
    m = sorted(v)[len(v) // 2]
    print('%-40s median %.1f us  min %.1f   (%.1f algorithmic TFLOP/s)' % (k, m, min(v), flop / m / 1e6))
