#!/usr/bin/env python3
"""GPU box: the 9x9 3 -> 64 first layer (risp_conv2d_toep_first) with the (channel, tap) reduction index of risp_conv_xwin.hip against the
band kernel of risp_conv_toep_first.hip (a second build of the library with -DRISP_XWIN_OFF in /tmp), interleaved rounds in ONE process on
the same tensors; error of both against float64 on one image.  python tools/ab_xwin.py [images h w [members]]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as TF
from reconfigisp_amd import lib as L, convnets as CN
variants = [v for v in sys.argv[1:] if v.startswith('-D') or v == '']          # extra builds of the new kernel: "-DXW_SPLIT=12" ...
a = [int(v) for v in sys.argv[1:] if not v.startswith('-D') and v != '']
n, h, w = (a + [32, 256, 256])[:3] if len(a) >= 3 else (32, 256, 256)
G = a[3] if len(a) >= 4 else 8
csrc = os.path.join(ROOT, 'reconfigisp_amd/csrc')
so = '/tmp/ab_xwin_off.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize', '-DRISP_XWIN_OFF',
                       '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-x', 'hip', '-shared', '-o', so] +
                      [os.path.join(csrc, f) for f in ('risp_conv_toep_first.hip', 'risp_conv_xwin.hip', 'risp_core.cpp')])
old = C.CDLL(so)
old.risp_conv2d_toep_first.restype, old.risp_conv2d_toep_first.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
new = L.load()
extra = {}
for i, v in enumerate(variants):
    so_v = '/tmp/ab_xwin_v%d.so' % i
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize'] + v.split() +
                          ['-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-x', 'hip', '-shared', '-o', so_v] +
                          [os.path.join(csrc, f) for f in ('risp_conv_toep_first.hip', 'risp_conv_xwin.hip', 'risp_core.cpp')])
    lv = C.CDLL(so_v)
    lv.risp_conv2d_toep_first.restype, lv.risp_conv2d_toep_first.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    extra[v] = lv
torch.manual_seed(0)
ws = [torch.randn(64, 3, 9, 9, device='cuda') * 0.05 for _ in range(G)]
bs = torch.stack([torch.randn(64, device='cuda') * 0.1 for _ in range(G)])
packs = torch.stack([CN.toep_first_weights(t) for t in ws])
x = torch.rand(n, 3, h, w, device='cuda')
table = torch.randn(G * n, 64 * 81, device='cuda') * 0.01
y = torch.empty(G * n, 64, h, w, device='cuda')
d = L.ConvDesc(N=G * n, H=h, W=w, cin=3, cout=64, ksize=9, load_mode=0, cin_img=0, epilogue=CN.EPI_RELU | CN.EPI_CASEBIAS, add_c=0, x=x.data_ptr(),
               wpack=packs.data_ptr(), bias=bs.data_ptr(), cvals=table.data_ptr(), add=None, mask=None, y=y.data_ptr())
d.group_n, d.group_flags, d.wpack_gs, d.bias_gs = n, L.GROUP_SHARED_X, packs.stride(0) * packs.element_size() // 4, bs.stride(0)
calls = {'band': lambda: old.risp_conv2d_toep_first(C.byref(d), None), 'tap index': lambda: new.risp_conv2d_toep_first(C.byref(d), None)}
for v, lv in extra.items():
    calls['tap index ' + v] = (lambda lv_: (lambda: lv_.risp_conv2d_toep_first(C.byref(d), None)))(lv)
# float64 reference of image 0 / member 0 (the border-case table as an explicit per-pixel bias)
def bc(v, L_):
    return torch.where(v < 4, v, torch.where(v >= L_ - 4, 8 - (L_ - 1 - v), torch.full_like(v, 4)))
yy, xx = bc(torch.arange(h, device='cuda'), h), bc(torch.arange(w, device='cuda'), w)
tb = table[0].view(64, 9, 9)[:, yy][:, :, xx]
ref = torch.relu(TF.conv2d(x[:1].double(), ws[0].double(), bs[0].double(), padding=4) + tb.double())
res, errs = {k: [] for k in calls}, {}
for k, fn in calls.items():
    y.fill_(float('nan'))
    assert fn() == 0
    torch.cuda.synchronize()
    e = y[:1].double() - ref
    errs[k] = (e.pow(2).mean().sqrt().item() / ref.abs().max().item(), e.abs().max().item() / ref.abs().max().item(), int(torch.isnan(y).sum()))
for rnd in range(5):
    for k, fn in calls.items():
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            fn()
        e1.record(); e1.synchronize()
        res[k].append(e0.elapsed_time(e1) / 5 * 1e3)
mb, mt = sorted(res['band'])[2], sorted(res['tap index'])[2]
issued = 3 * 2.0 * 9 * 32 * 64 * G * n * h * w
print('9x9 3 -> 64 first layer, %d x %d x 3 x %d x %d: band %.0f us (min %.0f)  tap index %.0f us (min %.0f)  x%.2f  issues %.0f TFLOP/s (%.2f of 2516.6), useful / issued %.2f; '
      'rms / max error vs float64 (nan): band %.1e / %.1e (%d), tap index %.1e / %.1e (%d)'
      % (G, n, h, w, mb, min(res['band']), mt, min(res['tap index']), mb / mt, issued / mt / 1e6, issued / mt / 1e6 / 2516.6, 27 / 32.,
         errs['band'][0], errs['band'][1], errs['band'][2], errs['tap index'][0], errs['tap index'][1], errs['tap index'][2]))
for v in extra:
    k = 'tap index ' + v
    print('   build [%s]: %.0f us (min %.0f), max error %.1e' % (v, sorted(res[k])[2], min(res[k]), errs[k][1]))
