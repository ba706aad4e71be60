#!/usr/bin/env python3
"""GPU box: risp_conv2d_tapout (filter rows in the rows of the matrix instruction) against risp_conv2d_toep (Toeplitz bands) on the two
layers it takes over, interleaved rounds in ONE process on the same tensors, error of both against float64 on one image.
python tools/ab_tapout.py [images h w [members]]   (default: the grouped launch of config 3: 8 members x 32 images of 256 x 256)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as TF
from reconfigisp_amd import lib as L, convnets as CN
a = [int(v) for v in sys.argv[1:]]
n, h, w = (a + [32, 256, 256])[:3] if len(a) >= 3 else (32, 256, 256)
G = a[3] if len(a) >= 4 else 8
torch.manual_seed(0)
lib = L.load()


def run(name, k, cin, wt_list, transpose, epi, add_c):
    packs_b = torch.stack([CN.toep_weights(t, transpose, 3 if transpose else None) for t in wt_list])
    packs_t = torch.stack([CN.tapout_weights(t, transpose, 3 if transpose else None) for t in wt_list])
    x = torch.randn(G * n, cin, h, w, device='cuda') * (torch.rand(G * n, cin, h, w, device='cuda') > 0.5)
    add = torch.randn(G * n, 3, h, w, device='cuda')
    y = torch.empty(G * n, 3, h, w, device='cuda')
    def desc(pack):
        d = L.ConvDesc(N=G * n, H=h, W=w, cin=cin, cout=3, ksize=k, load_mode=0, cin_img=0, epilogue=epi | 16, add_c=add_c, x=x.data_ptr(),
                       wpack=pack.data_ptr(), bias=None, cvals=None, add=add.data_ptr() if add_c else None, mask=None, y=y.data_ptr())
        d.group_n, d.group_flags, d.wpack_gs, d.bias_gs = n, 0, pack.stride(0) * pack.element_size() // 4, 0
        return d
    db, dt = desc(packs_b), desc(packs_t)
    calls = {'band': lambda: L.call('risp_conv2d_toep', C.byref(db), None), 'taprow': lambda: L.call('risp_conv2d_tapout', C.byref(dt), 0, None)}
    wt0 = wt_list[0]
    w_eff = wt0[:, :3].flip(2, 3).transpose(0, 1) if transpose else wt0
    ref = TF.conv2d(x[:1].double(), w_eff.double(), padding=k // 2) + (add[:1].double() if add_c else 0)
    res, errs = {kk: [] for kk in calls}, {}
    for kk, fn in calls.items():
        y.fill_(float('nan'))
        fn()
        torch.cuda.synchronize()
        e = (y[:1].double() - ref)
        errs[kk] = (e.pow(2).mean().sqrt().item() / ref.abs().max().item(), e.abs().max().item() / ref.abs().max().item())
    for rnd in range(5):
        for kk, fn in calls.items():
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(5):
                fn()
            e1.record(); e1.synchronize()
            res[kk].append(e0.elapsed_time(e1) / 5 * 1e3)
    mb, mt = sorted(res['band'])[2], sorted(res['taprow'])[2]
    useful = 2.0 * 3 * cin * k * k * G * n * h * w * 3          # 3 products per tap
    issued = 3 * 2.0 * 32 * k * cin * G * n * h * w
    print('%-28s %dx%d %d x %d x %d x %d x %d -> 3: band %.0f us (min %.0f)  tap-row %.0f us (min %.0f)  x%.2f  tap-row issues %.0f TFLOP/s (%.2f of 2516.6), useful/issued %.2f; '
          'rms / max error vs float64: band %.1e / %.1e, tap-row %.1e / %.1e'
          % (name, k, k, G, n, cin, h, w, mb, min(res['band']), mt, min(res['taprow']), mb / mt, issued / mt / 1e6, issued / mt / 1e6 / 2516.6, useful / issued,
             errs['band'][0], errs['band'][1], errs['taprow'][0], errs['taprow'][1]))


run('9x9 64 -> 3 backward-data', 9, 64, [torch.randn(64, 12, 9, 9, device='cuda') * 0.05 for _ in range(G)], True, CN.EPI_ADD, 3)
run('5x5 32 -> 3 forward', 5, 32, [torch.randn(3, 32, 5, 5, device='cuda') * 0.05 for _ in range(G)], False, CN.EPI_ADD, 3)
