#!/usr/bin/env python3
"""GPU box: in-process A/B of builds of risp_conv_wino.hip with different -D flags (interleaved rounds) on one
64 -> 64 3x3 layer through risp_conv2d_wino43 (F(4,3)), or - RISP_AB_ENTRY=wino45 - one 64 -> 32 5x5 layer through
risp_conv2d_wino45 (F(4,5)).  python tools/ab_wino43.py "" "-DRISP_W43_NO_GLDS" ...
[env RISP_AB_SHAPE="n h w", RISP_AB_EPI=1 for the residual + ReLU epilogue of a Path-Restore block, RISP_AB_CH="cin cout"]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = sys.argv[1:] or ['', '-DRISP_W43_GLDS=3']
base = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
        '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared']
core = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_core.cpp')
src = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_conv_wino.hip')
import torch
libs = {}
for i, v in enumerate(variants):
    so = '/tmp/wino43_%d.so' % i
    parts = v.split(',') if v else []               # "-Dflags" or "other_source.hip[,-Dflags]"
    srcf = src
    if parts and parts[0].endswith('.hip'):
        srcf, parts = os.path.join(ROOT, parts[0]), parts[1:]
    subprocess.check_call(base + parts + ['-o', so, srcf, core])
    libs[v or 'base'] = C.CDLL(so)
from reconfigisp_amd import lib as L
from reconfigisp_amd import convnets as CN
n, h, w = (int(v) for v in os.environ.get('RISP_AB_SHAPE', '64 128 128').split())
W45 = os.environ.get('RISP_AB_ENTRY') == 'wino45'                 # F(4,5): the pack follows each build's risp_conv_wino45_layout()
W5 = W45
K = 5 if W5 else 3
ENTRY = 'risp_conv2d_wino45' if W45 else 'risp_conv2d_wino43'
cin, cout = (int(v) for v in os.environ.get('RISP_AB_CH', '64 32' if W5 else '64 64').split())
torch.manual_seed(0)
wt = torch.randn(cout, cin, K, K, device='cuda') * (0.05 if K == 3 else 0.02)
b = torch.randn(cout, device='cuda') * 0.01
x = torch.rand(n, cin, h, w, device='cuda')
res_in = torch.rand(n, cout, h, w, device='cuda')
y = torch.empty(n, cout, h, w, device='cuda')
full_epi = os.environ.get('RISP_AB_EPI') == '1'
ref = torch.nn.functional.conv2d(x[:2], wt, b, padding=K // 2)
ref = torch.relu(ref + res_in[:2]) if full_epi else torch.relu(ref)
pack = CN.wino5_weights(wt, False, 4) if W5 else CN.wino43_weights(wt, False, 4)
res = {k: [] for k in libs}
packs = {}
for name, l in libs.items():
    if W45:
        pack = packs[name] = CN.wino45_weights(wt, False, 4, l.risp_conv_wino45_layout())
    getattr(l, ENTRY).restype, getattr(l, ENTRY).argtypes = L.SIGNATURES[ENTRY]
    l.risp_last_error.restype = C.c_char_p
    d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=K, load_mode=0, cin_img=0,
                   epilogue=CN.EPI_RELU | (CN.EPI_ADD if full_epi else 0), add_c=cout if full_epi else 0,
                   x=x.data_ptr(), wpack=pack.data_ptr(), bias=b.data_ptr(), cvals=None,
                   add=res_in.data_ptr() if full_epi else None, mask=None, y=y.data_ptr())
    libs[name] = (l, d)
    y.zero_()
    st = getattr(l, ENTRY)(C.byref(d), None)
    torch.cuda.synchronize()
    full = torch.relu(torch.nn.functional.conv2d(x[-1:], wt, b, padding=K // 2) + (res_in[-1:] if full_epi else 0))
    print('%-40s status %d %s max|err| %.2e (last image %.2e)' % (name, st, l.risp_last_error(), (y[:2] - ref).abs().max().item(),
                                                                (y[-1:] - full).abs().max().item()))
for rnd in range(7):
    for name, (l, d) in libs.items():
        for _ in range(2): getattr(l, ENTRY)(C.byref(d), None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): getattr(l, ENTRY)(C.byref(d), None)
        e1.record(); e1.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
flop = 2.0 * cin * cout * K * K * n * h * w
for k, v in res.items():
    m = sorted(v)[len(v) // 2]
    print('%-40s median %.1f us  min %.1f   (%.1f algorithmic TFLOP/s, %.1f issued)' % (k, m, min(v), flop / m / 1e6, flop * (0.4 if W45 else (0.6 if W5 else 0.5)) / m / 1e6))
