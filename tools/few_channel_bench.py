#!/usr/bin/env python3
"""GPU box: the three few-channel layers of the SRCNNRes family as the search step launches them (grouped: 8 members x 32 images of
256 x 256), product kernels only - the program tools/conv_pmc.sh profiles for their counters.
python tools/few_channel_bench.py first|first_exact|bwd9|bwd9_sums|fwd5|bwd5|tail3 [images h w members reps]
  (first_exact = the training forward: ties of the ReLU listed and recomputed; bwd9_sums = with the per-item channel sums of the input;
   RISP_FCB_SERIES=1: the launch time over consecutive blocks of launches - what sustained load does to it)
  first = 9x9 3 -> 64, ReLU + border-case bias (risp_conv2d_toep_first -> conv_xwin_kernel)
  bwd9  = 9x9 64 -> 3 backward-data + residual (risp_conv2d_tapout)        fwd5 = 5x5 32 -> 3 forward + residual (risp_conv2d_tapout)
  tail3 = 3x3 64 -> 3, Path-Restore's last layer, members x images in one launch (risp_conv2d_narrow3; RISP_FCB_SMALL=1: the vector kernel)
  bwd5  = 5x5 3 -> 32 backward-data with a ReLU mask (risp_conv2d_thin5; RISP_FCB_WINO=1: the fp32 F(4,5) kernel it replaces)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reconfigisp_amd import lib as L, convnets as CN
what = sys.argv[1] if len(sys.argv) > 1 else 'first'
n, h, w, G, reps = (int(v) for v in (sys.argv[2:7] + ['32', '256', '256', '8', '10'][len(sys.argv) - 2:]))
torch.manual_seed(0)
if what.startswith('first'):
    packs = torch.stack([CN.toep_first_weights(torch.randn(64, 3, 9, 9, device='cuda') * 0.05) for _ in range(G)])
    bs = torch.randn(G, 64, device='cuda') * 0.1
    x = torch.rand(n, 3, h, w, device='cuda')
    table = torch.randn(G * n, 64 * 81, device='cuda') * 0.01
    y = torch.empty(G * n, 64, h, w, device='cuda')
    d = L.ConvDesc(N=G * n, H=h, W=w, cin=3, cout=64, ksize=9, load_mode=0, cin_img=0, epilogue=CN.EPI_RELU | CN.EPI_CASEBIAS, add_c=0, x=x.data_ptr(),
                   wpack=packs.data_ptr(), bias=bs.data_ptr(), cvals=table.data_ptr(), add=None, mask=None, y=y.data_ptr())
    d.group_n, d.group_flags, d.wpack_gs, d.bias_gs = n, L.GROUP_SHARED_X, packs.stride(0) * packs.element_size() // 4, bs.stride(0)
    fn = lambda: L.call('risp_conv2d_toep_first', C.byref(d), None)
    if what == 'first_exact':
        w32 = torch.stack([torch.randn(64, 3, 9, 9, device='cuda') * 0.05 for _ in range(G)])
        ties = torch.zeros(1 + CN.TIES_MAX, device='cuda', dtype=torch.int32)
        fn = lambda: L.call('risp_conv2d_toep_first_exact', C.byref(d), w32.data_ptr(), w32.stride(0), ties.data_ptr(), CN.TIES_MAX, None)
    useful, alg = 3 * 2.0 * 81 * 3 * 64 * G * n * h * w, (64 * G + 3) * 4.0 * n * h * w
elif what == 'tail3':
    wt = torch.randn(3, 64, 3, 3, device='cuda') * 0.05
    b = torch.randn(3, device='cuda') * 0.1
    x = torch.randn(G * n, 64, h, w, device='cuda')
    y = torch.empty(G * n, 3, h, w, device='cuda')
    small = os.environ.get('RISP_FCB_SMALL') == '1'
    pack = CN.SmallConv(wt, b).wpack if small else CN.narrow3_weights(wt)
    d = L.ConvDesc(N=G * n, H=h, W=w, cin=64, cout=3, ksize=3, load_mode=0, cin_img=0, epilogue=0, add_c=0, x=x.data_ptr(),
                   wpack=pack.data_ptr(), bias=b.data_ptr(), cvals=None, add=None, mask=None, y=y.data_ptr())
    fn = lambda: L.call('risp_conv2d_small' if small else 'risp_conv2d_narrow3', C.byref(d), None)
    useful, alg = 3 * 2.0 * 9 * 64 * 3 * G * n * h * w, (64 + 3) * 4.0 * G * n * h * w
elif what == 'bwd5':
    wfs = [torch.randn(3, 32, 5, 5, device='cuda') * 0.05 for _ in range(G)]
    gy = torch.randn(G * n, 3, h, w, device='cuda') * 1e-3
    act = torch.randn(G * n, 32, h, w, device='cuda')
    y = torch.empty(G * n, 32, h, w, device='cuda')
    if os.environ.get('RISP_FCB_WINO') == '1':
        packs = torch.stack([CN.wino45_weights(t, True) for t in wfs])
        entry = 'risp_conv2d_wino45'
    else:
        packs = torch.stack([CN.thin5_weights(t, True) for t in wfs])
        entry = 'risp_conv2d_thin5'
    d = L.ConvDesc(N=G * n, H=h, W=w, cin=3, cout=32, ksize=5, load_mode=0, cin_img=0, epilogue=CN.EPI_MASK | CN.EPI_NOBIAS, add_c=0, x=gy.data_ptr(),
                   wpack=packs.data_ptr(), bias=None, cvals=None, add=None, mask=act.data_ptr(), y=y.data_ptr())
    d.group_n, d.group_flags, d.wpack_gs, d.bias_gs = n, 0, packs.stride(0) * packs.element_size() // 4, 0
    fn = lambda: L.call(entry, C.byref(d), None)
    useful, alg = 3 * 2.0 * 25 * 3 * 32 * G * n * h * w, (3 + 32 + 32) * 4.0 * G * n * h * w
else:
    k, cin, tr = (9, 64, True) if what.startswith('bwd9') else (5, 32, False)
    wts = [torch.randn(64, 12, 9, 9, device='cuda') * 0.05 if tr else torch.randn(3, 32, 5, 5, device='cuda') * 0.05 for _ in range(G)]
    packs = torch.stack([CN.tapout_weights(t, tr, 3 if tr else None) for t in wts])
    x = torch.randn(G * n, cin, h, w, device='cuda') * (torch.rand(G * n, cin, h, w, device='cuda') > 0.5)
    add = torch.randn(G * n, 3, h, w, device='cuda')
    y = torch.empty(G * n, 3, h, w, device='cuda')
    d = L.ConvDesc(N=G * n, H=h, W=w, cin=cin, cout=3, ksize=k, load_mode=0, cin_img=0, epilogue=CN.EPI_ADD | 16, add_c=3, x=x.data_ptr(),
                   wpack=packs.data_ptr(), bias=None, cvals=None, add=add.data_ptr(), mask=None, y=y.data_ptr())
    d.group_n, d.group_flags, d.wpack_gs, d.bias_gs = n, 0, packs.stride(0) * packs.element_size() // 4, 0
    seg = CN.tapout_seg(G * n, h, w, False)
    fn = lambda: L.call('risp_conv2d_tapout', C.byref(d), seg, None)
    if what == 'bwd9_sums':
        ps = torch.empty((G * n, L.load().risp_conv_tapout_items(G * n, h, w, seg), 64), device='cuda')
        fn = lambda: L.call('risp_conv2d_tapout_sums', C.byref(d), seg, ps.data_ptr(), None)
    useful, alg = 3 * 2.0 * k * k * cin * 3 * G * n * h * w, (cin + 6) * 4.0 * G * n * h * w
for _ in range(3):
    fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(reps):
    fn()
e1.record(); e1.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
if os.environ.get('RISP_FCB_SERIES') == '1':
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    ev[0].record()
    for i in range(20):
        for _ in range(reps):
            fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    print('  consecutive blocks of %d launches, us per launch: %s' % (reps, ' '.join('%.0f' % (ev[i].elapsed_time(ev[i + 1]) / reps * 1e3) for i in range(20))))
print('%s %d x %d x %d x %d: %.0f us per launch, useful split-precision products %.0f TFLOP/s, tensors in + out %.0f GB/s' % (what, G, n, h, w, us, useful / us / 1e6, alg / us / 1e3))
