#!/usr/bin/env python3
"""GPU box: where the waves of the wave-specialised split-precision 3x3 kernel (risp_conv_f16x2_ws.hip) spend their life - a
diagnostic build with in-kernel stamps (-DRISP_WS_STAMPS; extra flags as arguments).  python tools/ws_stamps.py [epi 0|1|2|3] [-D...]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reconfigisp_amd import lib as L, convnets as CN
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
extra = sys.argv[2:]
so = '/tmp/ws_stamps.so'
csrc = os.path.join(ROOT, 'reconfigisp_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize', '-DRISP_WS_STAMPS',
                       '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-x', 'hip', '-shared', '-o', so] + extra +
                      [os.path.join(csrc, f) for f in ('risp_conv_f16x2.hip', 'risp_conv_f16x2_ws.hip', 'risp_core.cpp')])
lib = C.CDLL(so)
n, h, w, cin, cout = 32, 256, 256, 64, 64
torch.manual_seed(0)
wt = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
b = torch.randn(cout, device='cuda') * 0.01
x = torch.rand(n, cin, h, w, device='cuda')
add, mask = torch.rand(n, cout, h, w, device='cuda'), torch.randn(n, cout, h, w, device='cuda')
y = torch.empty(n, cout, h, w, device='cuda')
pack = CN.f16x2_weights(wt, False)
epi = {0: CN.EPI_RELU, 1: CN.EPI_ADD | CN.EPI_RELU, 2: CN.EPI_MASK, 3: CN.EPI_ADD | CN.EPI_MASK}[mode]
nwg = torch.cuda.get_device_properties(0).multi_processor_count
buf = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device='cuda')
d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=3, load_mode=0, cin_img=0, epilogue=epi, add_c=cout if epi & CN.EPI_ADD else 0,
               x=x.data_ptr(), wpack=pack.data_ptr(), bias=b.data_ptr(), cvals=buf.data_ptr(), add=add.data_ptr() if epi & CN.EPI_ADD else None,
               mask=mask.data_ptr() if epi & CN.EPI_MASK else None, y=y.data_ptr())
lib.risp_conv2d_f16x2.restype, lib.risp_conv2d_f16x2.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
for _ in range(200):                                 # ~0.1 s of back-to-back launches: the clock the chip holds under this load
    assert lib.risp_conv2d_f16x2(C.byref(d), None) == 0
torch.cuda.synchronize()
t = buf.view(nwg, 8, 8).double()
cons, prod = t[:, :4], t[:, 4:]
life = cons[..., 4]
clk = (life / ((cons[..., 6] - cons[..., 5]) * 10e-9)).median().item() / 1e9
tiles = n * (h // 8) * (w // 64) / nwg
print('epilogue mode %d: consumer wave life %.0f cycles (median) = %.0f per tile; in-kernel clock %.2f GHz -> %.1f us; matrix instructions alone: %.0f cycles per tile'
      % (mode, life.median().item(), life.median().item() / tiles, clk, life.median().item() / clk / 1e3, 4 * 3 * 72 * 32))
print('  consumers: barrier wait %.3f, chunk head (exponent, first operand reads) %.3f, matrix steps %.3f, epilogue %.3f of the life'
      % tuple(cons[..., i].sum().item() / life.sum().item() for i in (0, 3, 1, 2)))
pl = prod[..., 0] + prod[..., 1]
print('  producers: work %.3f, barrier wait %.3f of their life' % (prod[..., 0].sum().item() / pl.sum().item(), prod[..., 1].sum().item() / pl.sum().item()))
