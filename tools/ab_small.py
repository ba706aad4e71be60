#!/usr/bin/env python3
"""GPU box: in-process A/B of builds of risp_conv_small.hip (-D flags) and channel-group counts on the 9x9 64 -> 3
backward-data layer of SRCNNRes at a small batch.  python tools/ab_small.py "" "-DRISP_SMALL_SPLIT_SPY2=512"
[env RISP_AB_SHAPE="n h w", RISP_AB_GROUPS="1 2 4 8", RISP_AB_LAYER="cin cout k shuffle(0|1)"]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = sys.argv[1:] or ['', '-DRISP_SMALL_SPLIT_SPY2=512']
base = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
        '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared']
core = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_core.cpp')
src = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_conv_small.hip')
import torch
libs = {}
for i, v in enumerate(variants):
    so = '/tmp/small_%d.so' % i
    subprocess.check_call(base + (v.split(',') if v else []) + ['-o', so, src, core])
    libs[v or 'base'] = C.CDLL(so)
from reconfigisp_amd import lib as L
from reconfigisp_amd import convnets as CN
n, h, w = (int(v) for v in os.environ.get('RISP_AB_SHAPE', '4 256 256').split())
cin, cout, k, shuf = (int(v) for v in os.environ.get('RISP_AB_LAYER', '64 3 9 0').split())
torch.manual_seed(0)
wt = torch.randn(cout, cin, k, k, device='cuda') * 0.02
x = torch.rand(n, cin, h, w, device='cuda')
y = torch.empty((n, cout // 4, 2 * h, 2 * w) if shuf else (n, cout, h, w), device='cuda')
pack, _ = CN.small_weights(wt)
ref = torch.nn.functional.conv2d(x[:1], wt, None, padding=k // 2)
if shuf:
    ref = torch.nn.functional.pixel_shuffle(ref, 2)
for name, l in libs.items():
    l.risp_conv2d_small_split.restype, l.risp_conv2d_small_split.argtypes = L.SIGNATURES['risp_conv2d_small_split']
for groups in (int(g) for g in os.environ.get('RISP_AB_GROUPS', '1 2 4 8').split()):
    scratch = torch.empty((max(groups, 1), n, cout, h, w), device='cuda')
    d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=k, load_mode=0, cin_img=0, epilogue=CN.EPI_NOBIAS | (CN.EPI_SHUFFLE2 if shuf else 0), add_c=0,
                   x=x.data_ptr(), wpack=pack.data_ptr(), bias=None, cvals=None, add=None, mask=None, y=y.data_ptr())
    res = {k_: [] for k_ in libs}
    for rnd in range(5):
        for name, l in libs.items():
            y.zero_()
            st = l.risp_conv2d_small_split(C.byref(d), scratch.data_ptr(), groups, None)
            torch.cuda.synchronize()
            err = (y[:1] - ref).abs().max().item()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): l.risp_conv2d_small_split(C.byref(d), scratch.data_ptr(), groups, None)
            e1.record(); e1.synchronize()
            res[name].append((e0.elapsed_time(e1) / 10 * 1e3, st, err))
    for name, v in res.items():
        print('groups %d %-34s median %.1f us  (status %d, max|err| %.1e)' % (groups, name, sorted(t for t, _, _ in v)[len(v) // 2], v[0][1], v[0][2]))
