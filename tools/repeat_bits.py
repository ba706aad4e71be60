#!/usr/bin/env python3
"""GPU box: bit-repeatability of the round-6 convolution kernels at the search step's full size (grouped 8 members x 32 images of
256 x 256): every launch form repeated, outputs compared bit for bit with the first run - under load from a second stream that streams
matrix work of its own.  python tools/repeat_bits.py [repeats]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reconfigisp_amd import lib as L, convnets as CN
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G, n, h, w = 8, 32, 256, 256
torch.manual_seed(0)
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)


def noise():
    with torch.cuda.stream(side):
        for _ in range(4):
            a @ a


def check(name, fn, outs):
    fn(); torch.cuda.synchronize()
    first = [o.clone() for o in outs]
    bad = 0
    for r in range(reps):
        for o in outs:
            o.fill_(float('nan'))
        if r % 2:
            noise()
        fn(); torch.cuda.synchronize()
        bad += any(not torch.equal(o, f) for o, f in zip(outs, first))
    print('%-46s %d / %d runs differ from the first' % (name, bad, reps), flush=True)
    return bad


def desc(**kw):
    base = dict(N=G * n, H=h, W=w, load_mode=0, cin_img=0, add_c=0, bias=None, cvals=None, add=None, mask=None)
    base.update(kw)
    d = L.ConvDesc(**base)
    return d


total = 0
# tap-index first layer, training form
packs = torch.stack([CN.toep_first_weights(torch.randn(64, 3, 9, 9, device='cuda') * 0.05) for _ in range(G)])
w32 = torch.stack([torch.randn(64, 3, 9, 9, device='cuda') * 0.05 for _ in range(G)])
bs = torch.randn(G, 64, device='cuda') * 0.1
x = torch.rand(n, 3, h, w, device='cuda')
table = torch.randn(G * n, 64 * 81, device='cuda') * 0.01
y = torch.empty(G * n, 64, h, w, device='cuda')
ties = torch.zeros(1 + CN.TIES_MAX, device='cuda', dtype=torch.int32)
d = desc(cin=3, cout=64, ksize=9, epilogue=CN.EPI_RELU | CN.EPI_CASEBIAS, x=x.data_ptr(), wpack=packs.data_ptr(), bias=bs.data_ptr(), cvals=table.data_ptr(), y=y.data_ptr())
d.group_n, d.group_flags, d.wpack_gs, d.bias_gs = n, L.GROUP_SHARED_X, packs.stride(0) * packs.element_size() // 4, bs.stride(0)
total += check('tap-index first layer with exact ties', lambda: L.call('risp_conv2d_toep_first_exact', C.byref(d), w32.data_ptr(), w32.stride(0), ties.data_ptr(), CN.TIES_MAX, None), [y])
del y, table
# tap-row 9x9 backward-data with the channel sums
wts = [torch.randn(64, 12, 9, 9, device='cuda') * 0.05 for _ in range(G)]
packs2 = torch.stack([CN.tapout_weights(t, True, 3) for t in wts])
g1 = torch.randn(G * n, 64, h, w, device='cuda') * (torch.rand(G * n, 64, h, w, device='cuda') > 0.5)
add = torch.randn(G * n, 3, h, w, device='cuda')
y3 = torch.empty(G * n, 3, h, w, device='cuda')
seg = CN.tapout_seg(G * n, h, w, False)
ps = torch.empty((G * n, L.load().risp_conv_tapout_items(G * n, h, w, seg), 64), device='cuda')
d2 = desc(cin=64, cout=3, ksize=9, epilogue=CN.EPI_ADD | 16, add_c=3, x=g1.data_ptr(), wpack=packs2.data_ptr(), add=add.data_ptr(), y=y3.data_ptr())
d2.group_n, d2.group_flags, d2.wpack_gs, d2.bias_gs = n, 0, packs2.stride(0) * packs2.element_size() // 4, 0
total += check('tap-row 9x9 backward-data + channel sums', lambda: L.call('risp_conv2d_tapout_sums', C.byref(d2), seg, ps.data_ptr(), None), [y3, ps])
del g1
# thin-input 5x5 backward-data with a mask
wf = [torch.randn(3, 32, 5, 5, device='cuda') * 0.05 for _ in range(G)]
packs3 = torch.stack([CN.thin5_weights(t, True) for t in wf])
gy = torch.randn(G * n, 3, h, w, device='cuda') * 1e-3
act = torch.randn(G * n, 32, h, w, device='cuda')
y32 = torch.empty(G * n, 32, h, w, device='cuda')
d3 = desc(cin=3, cout=32, ksize=5, epilogue=CN.EPI_MASK | CN.EPI_NOBIAS, x=gy.data_ptr(), wpack=packs3.data_ptr(), mask=act.data_ptr(), y=y32.data_ptr())
d3.group_n, d3.group_flags, d3.wpack_gs, d3.bias_gs = n, 0, packs3.stride(0) * packs3.element_size() // 4, 0
total += check('thin-input 5x5 backward-data with a mask', lambda: L.call('risp_conv2d_thin5', C.byref(d3), None), [y32])
# tap-row 5x5 forward
wt5 = [torch.randn(3, 32, 5, 5, device='cuda') * 0.05 for _ in range(G)]
packs4 = torch.stack([CN.tapout_weights(t) for t in wt5])
d4 = desc(cin=32, cout=3, ksize=5, epilogue=CN.EPI_ADD | 16, add_c=3, x=act.data_ptr(), wpack=packs4.data_ptr(), add=add.data_ptr(), y=y3.data_ptr())
d4.group_n, d4.group_flags, d4.wpack_gs, d4.bias_gs = n, 0, packs4.stride(0) * packs4.element_size() // 4, 0
total += check('tap-row 5x5 forward', lambda: L.call('risp_conv2d_tapout', C.byref(d4), seg, None), [y3])
# narrow 3x3 tail
wt3 = torch.randn(3, 64, 3, 3, device='cuda') * 0.05
pk = CN.narrow3_weights(wt3)
x64 = torch.randn(n, 64, h, w, device='cuda')
yn = torch.empty(n, 3, h, w, device='cuda')
d5 = L.ConvDesc(N=n, H=h, W=w, cin=64, cout=3, ksize=3, load_mode=0, cin_img=0, epilogue=CN.EPI_NOBIAS, add_c=0, x=x64.data_ptr(), wpack=pk.data_ptr(), bias=None,
                cvals=None, add=None, mask=None, y=yn.data_ptr())
total += check('narrow 3x3 tail', lambda: L.call('risp_conv2d_narrow3', C.byref(d5), None), [yn])
sys.exit(1 if total else 0)
