#!/usr/bin/env python3
"""GPU box: the split-precision (2 x f16, 3 products) convolution risp_conv2d_f16x2 - builds of risp_conv_f16x2.hip with
different -D flags, interleaved rounds, on one 3x3 layer; error against float64 next to the product library's F(4,3) kernel
(risp_conv2d_wino43) on the same data.  python tools/ab_f16x2.py "" "-DRISP_H2_ABL=1" ...
[env RISP_AB_SHAPE="n h w", RISP_AB_EPI=1 residual + ReLU epilogue, RISP_AB_CH="cin cout", RISP_AB_XSCALE=1e-5 gradient-like input]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
variants = sys.argv[1:] or ['']
base = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
        '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared']
core = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_core.cpp')
src = os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_conv_f16x2.hip')
import torch
libs = {}
for i, v in enumerate(variants):
    so = '/tmp/f16x2_%d.so' % i
    parts = v.split(',') if v else []
    srcf = src
    if parts and parts[0].endswith('.hip'):
        srcf, parts = os.path.join(ROOT, parts[0]), parts[1:]
    subprocess.check_call(base + parts + ['-o', so, srcf, core])
    libs[v or 'base'] = C.CDLL(so)
from reconfigisp_amd import lib as L
from reconfigisp_amd import convnets as CN
n, h, w = (int(v) for v in os.environ.get('RISP_AB_SHAPE', '32 256 256').split())
K = int(os.environ.get('RISP_AB_K', '3'))
cin, cout = (int(v) for v in os.environ.get('RISP_AB_CH', '64 64').split())
xscale = float(os.environ.get('RISP_AB_XSCALE', '1'))
torch.manual_seed(0)
wt = torch.randn(cout, cin, K, K, device='cuda') * (0.05 if K == 3 else 0.02)
b = torch.randn(cout, device='cuda') * 0.01
x = torch.rand(n, cin, h, w, device='cuda')
if xscale != 1:
    x = torch.randn(n, cin, h, w, device='cuda') * xscale * (torch.rand(n, cin, h, w, device='cuda') > 0.5)
res_in = torch.rand(n, cout, h, w, device='cuda') * xscale
y = torch.empty(n, cout, h, w, device='cuda')
EPI_MODE = int(os.environ.get('RISP_AB_EPI', '0'))          # 1 residual + ReLU, 2 mask, 3 residual + mask
full_epi = EPI_MODE in (1, 3)
use_mask = EPI_MODE in (2, 3)
mask_t = torch.randn(n, cout, h, w, device='cuda')
relu = os.environ.get('RISP_AB_RELU', '1') == '1'
def ref64(sl):
    r = torch.nn.functional.conv2d(x[sl].double(), wt.double(), b.double(), padding=K // 2)
    if full_epi:
        r = r + res_in[sl].double()
    if use_mask:
        return r * (mask_t[sl] > 0)
    return torch.relu(r) if relu else r
refs = {0: ref64(slice(0, 1)), n - 1: ref64(slice(n - 1, n))}
scale = max(r.abs().max().item() for r in refs.values())
def err(yy):
    e = torch.cat([(yy[i:i + 1].double() - r).flatten() for i, r in refs.items()])
    return e.pow(2).mean().sqrt().item() / scale, e.abs().max().item() / scale
relu = relu and not use_mask
epi = (CN.EPI_RELU if relu else 0) | (CN.EPI_ADD if full_epi else 0) | (CN.EPI_MASK if use_mask else 0)
def desc(pack):
    return L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=K, load_mode=0, cin_img=0, epilogue=epi, add_c=cout if full_epi else 0,
                      x=x.data_ptr(), wpack=pack.data_ptr(), bias=b.data_ptr(), cvals=None,
                      add=res_in.data_ptr() if full_epi else None, mask=mask_t.data_ptr() if use_mask else None, y=y.data_ptr())
# the product library's fp32 kernels on the same data
prod = L.load()
runs = {}
if K == 3 and hasattr(prod, 'risp_conv2d_wino43'):
    p43 = CN.wino43_weights(wt, False, prod.risp_conv_wino43_chunk())
    d43 = desc(p43)
    runs['product F(4,3) fp32'] = (lambda: prod.risp_conv2d_wino43(C.byref(d43), None))
if K == 5:
    p45 = CN.wino45_weights(wt, False)
    d45 = desc(p45)
    runs['product F(4,5) fp32'] = (lambda: prod.risp_conv2d_wino45(C.byref(d45), None))
pd = torch.empty(prod.risp_conv_wpack_floats(cin, cout, K), device='cuda')
prod.risp_conv_pack_weights(C.c_void_p(wt.data_ptr()), cin, cout, K, 0, C.c_void_p(pd.data_ptr()), None)
dd = desc(pd)
runs['product direct fp32'] = (lambda: prod.risp_conv2d(C.byref(dd), None))
ph = CN.f16x2_weights(wt, False)
dh = desc(ph)
for name, l in libs.items():
    l.risp_conv2d_f16x2.restype, l.risp_conv2d_f16x2.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    l.risp_last_error.restype = C.c_char_p
    runs['f16x2 ' + name] = (lambda l=l: l.risp_conv2d_f16x2(C.byref(dh), None))
yt = torch.nn.functional.conv2d(x[:1], wt, b, padding=K // 2)
if full_epi: yt = yt + res_in[:1]
if use_mask: yt = yt * (mask_t[:1] > 0)
print('torch fp32 conv (image 0)               rms %.2e max %.2e of max|y|' % (
    ((torch.relu(yt) if relu else yt).double() - refs[0]).pow(2).mean().sqrt().item() / scale,
    ((torch.relu(yt) if relu else yt).double() - refs[0]).abs().max().item() / scale))
for name, f in runs.items():
    y.fill_(float('nan'))
    st = f()
    torch.cuda.synchronize()
    msg = ''
    if st and name.startswith('f16x2'):
        msg = libs[name[6:]].risp_last_error().decode()
    print('%-40s status %d %s rms %.2e max %.2e of max|y|  nan %d' % (name, st, msg, *err(y), int(torch.isnan(y).sum().item())))
res = {k: [] for k in runs}
for rnd in range(5):
    for name, f in runs.items():
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): f()
        e1.record(); e1.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
for name, l in libs.items():                       # diagnostic builds (-DRISP_H2_STAMPS): where a wave's life goes
    if '-DRISP_H2_STAMPS' not in name:
        continue
    l.risp_conv_f16x2_occupancy.restype = C.c_int
    nwg = min(((w + 63) // 64) * ((h + 7) // 8) * n * (cout // 32 if K == 5 else 1), (1 if 'RISP_H2_WGS=1' in name else 2) * torch.cuda.get_device_properties(0).multi_processor_count)
    buf = torch.zeros(nwg * 4 * 10, dtype=torch.int64, device='cuda')
    ds = desc(ph)
    ds.cvals = buf.data_ptr()
    for _ in range(3):
        l.risp_conv2d_f16x2(C.byref(ds), None)
    torch.cuda.synchronize()
    raw = buf.view(nwg * 4, 10)
    steps = raw[:, 8].double()                        # issue time of the matrix-instruction steps alone
    t = raw.double()
    life = t[:, 4]
    clk = (life / ((t[:, 6] - t[:, 5]) * 10e-9)).median().item() / 1e9       # s_memrealtime ticks at 100 MHz
    print('   matrix-instruction steps alone: %.3f of a wave life' % (steps.sum().item() / life.sum().item()))
    print('%s: workgroups per CU %d; in-kernel clock %.2f GHz; wave life %.0f cycles (median); shares: wait for tile %.3f, staging '
          '(barrier A .. barrier B) %.3f, matrix phase %.3f, epilogue %.3f' % (
              name, l.risp_conv_f16x2_occupancy(), clk, life.median().item(), *(t[:, i].sum().item() / life.sum().item() for i in range(4))))
    hw_id = buf.view(nwg * 4, 10)[:, 7]
    cu = ((hw_id >> 8) & 0xf) | (((hw_id >> 13) & 0x7) << 4) | (((hw_id >> 32) & 0xf) << 8)        # cu_id, se_id (gfx9 HW_ID layout), xcc_id
    if nwg % 2 == 0:
        cw = cu.view(nwg, 4)[:, 0]
        print('   workgroups b and b + %d on the same CU: %d of %d' % (nwg // 2, int((cw[:nwg // 2] == cw[nwg // 2:]).sum().item()), nwg // 2))
    start = buf.view(nwg * 4, 10)[:, 5]
    print('   start spread (100 MHz ticks): %d; distinct (se, cu) ids seen %d' % ((start.max() - start.min()).item(), len(set(cu.tolist()))))
flop = 2.0 * cin * cout * K * K * n * h * w
byts = 4.0 * n * h * w * (cin + cout * (1 + full_epi + use_mask))
for k, v in res.items():
    m = sorted(v)[len(v) // 2]
    print('%-40s median %.1f us  min %.1f   (%.1f direct-convolution TFLOP/s, %.2f TB/s of tensors)' % (k, m, min(v), flop / m / 1e6, byts / m / 1e6))
