#!/usr/bin/env python3
"""GPU box: where the waves of risp_conv2d_tapout (risp_conv_tapout.hip) spend their life - a diagnostic build with in-kernel stamps
(-DRISP_TO_STAMPS; extra -D flags as arguments).  python tools/tapout_stamps.py [k 9|5] [sums] [-D...]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reconfigisp_amd import lib as L, convnets as CN
k = int(sys.argv[1]) if len(sys.argv) > 1 else 9
SUMS = 'sums' in sys.argv                                # risp_conv2d_tapout_sums (the training launch of config 3) instead
extra = [a for a in sys.argv[2:] if a != 'sums']
so = '/tmp/tapout_stamps.so'
csrc = os.path.join(ROOT, 'reconfigisp_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize', '-DRISP_TO_STAMPS',
                       '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-x', 'hip', '-shared', '-o', so] + extra +
                      [os.path.join(csrc, f) for f in ('risp_conv_tapout.hip', 'risp_core.cpp')])
lib = C.CDLL(so)
G, n, h, w = 8, 32, 256, 256
cin = 64 if k == 9 else 32
torch.manual_seed(0)
if k == 9:
    packs = torch.stack([CN.tapout_weights(torch.randn(64, 12, 9, 9, device='cuda') * 0.05, True, 3) for _ in range(G)])
else:
    packs = torch.stack([CN.tapout_weights(torch.randn(3, 32, 5, 5, device='cuda') * 0.05) for _ in range(G)])
x = torch.randn(G * n, cin, h, w, device='cuda') * (torch.rand(G * n, cin, h, w, device='cuda') > 0.5)
add = torch.randn(G * n, 3, h, w, device='cuda')
y = torch.empty(G * n, 3, h, w, device='cuda')
nwg = torch.cuda.get_device_properties(0).multi_processor_count
buf = torch.zeros(nwg * 8 * 4, dtype=torch.int64, device='cuda')
d = L.ConvDesc(N=G * n, H=h, W=w, cin=cin, cout=3, ksize=k, load_mode=0, cin_img=0, epilogue=CN.EPI_ADD | 16, add_c=3, x=x.data_ptr(),
               wpack=packs.data_ptr(), bias=None, cvals=buf.data_ptr(), add=add.data_ptr(), mask=None, y=y.data_ptr())
d.group_n, d.group_flags, d.wpack_gs, d.bias_gs = n, 0, packs.stride(0) * packs.element_size() // 4, 0
lib.risp_conv2d_tapout.restype, lib.risp_conv2d_tapout.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_void_p]
lib.risp_conv2d_tapout_sums.restype, lib.risp_conv2d_tapout_sums.argtypes = C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
lib.risp_conv_tapout_items.restype, lib.risp_conv_tapout_items.argtypes = C.c_int, [C.c_int] * 4
lib.risp_conv_tapout_seg_rows.restype, lib.risp_conv_tapout_seg_rows.argtypes = C.c_int, [C.c_int] * 3
if SUMS:
    seg = lib.risp_conv_tapout_seg_rows(G * n, h, w)
    ps = torch.empty(G * n, lib.risp_conv_tapout_items(G * n, h, w, seg), 64, device='cuda')
    run = lambda: lib.risp_conv2d_tapout_sums(C.byref(d), 0, ps.data_ptr(), None)
else:
    run = lambda: lib.risp_conv2d_tapout(C.byref(d), 0, None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    assert run() == 0
e0.record()
for _ in range(20):
    assert run() == 0
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
t = buf.view(nwg, 8, 4).double()
cons, prod = t[:, :4], t[:, 4:]
chunks = G * n * ((w + 127) // 128) * ((h + 3) // 4) * (cin // 16) / nwg
life = cons[..., 3].median().item()
print('%dx%d %d -> 3, %d x %d x %d x %d: %.0f us per launch (stamped build); wave life %.0f cycles (median) = %.0f per chunk (%d matrix instructions = %d cycles); in-kernel clock ~%.2f GHz'
      % (k, k, cin, G * n, cin, h, w, us, life, life / chunks, 12 * k, 12 * k * 32, life / us / 1e3))
print('  consumers: barrier wait %.3f, matrix phase (exponent, operand reads, products) %.3f, ring / retiring rows / stores %.3f of the life'
      % tuple(cons[..., i].sum().item() / cons[..., 3].sum().item() for i in (0, 1, 2)))
print('  producers: work (stage, maxima, requests, waits for memory) %.3f, barrier wait %.3f of the life'
      % tuple(prod[..., i].sum().item() / prod[..., 3].sum().item() for i in (1, 0)))
if any(a.startswith('-DTO_PSPLIT') for a in extra):
    print('  producers, phase start up to the split point (%s): %.0f cycles per chunk; from there to the barrier: %.0f'
          % ([a for a in extra if a.startswith('-DTO_PSPLIT')][0], prod[..., 2].median().item() / chunks, prod[..., 1].median().item() / chunks))
