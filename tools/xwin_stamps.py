#!/usr/bin/env python3
"""GPU box: where the waves of the 3-channel first-layer kernel (risp_conv_xwin.hip) spend their life - a diagnostic build with in-kernel
stamps (-DRISP_XW_STAMPS; extra -D flags as arguments).  python tools/xwin_stamps.py [-D...]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reconfigisp_amd import lib as L, convnets as CN
so = '/tmp/xwin_stamps.so'
csrc = os.path.join(ROOT, 'reconfigisp_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize', '-DRISP_XW_STAMPS',
                       '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-x', 'hip', '-shared', '-o', so] + sys.argv[1:] +
                      [os.path.join(csrc, f) for f in ('risp_conv_toep_first.hip', 'risp_conv_xwin.hip', 'risp_core.cpp')])
lib = C.CDLL(so)
G, n, h, w = 8, 32, 256, 256
torch.manual_seed(0)
packs = torch.stack([CN.toep_first_weights(torch.randn(64, 3, 9, 9, device='cuda') * 0.05) for _ in range(G)])
bs = torch.randn(G, 64, device='cuda') * 0.1
x = torch.rand(n, 3, h, w, device='cuda')
table = torch.randn(G * n, 64 * 81, device='cuda') * 0.01
y = torch.empty(G * n, 64, h, w, device='cuda')
nwg = torch.cuda.get_device_properties(0).multi_processor_count
buf = torch.zeros(nwg * 8 * 8 + 16, dtype=torch.int64, device='cuda')
w32 = torch.zeros(G, 64, 3, 9, 9, device='cuda')
d = L.ConvDesc(N=G * n, H=h, W=w, cin=3, cout=64, ksize=9, load_mode=0, cin_img=0, epilogue=CN.EPI_RELU | CN.EPI_CASEBIAS, add_c=0, x=x.data_ptr(),
               wpack=packs.data_ptr(), bias=bs.data_ptr(), cvals=table.data_ptr(), add=None, mask=None, y=y.data_ptr())
d.group_n, d.group_flags, d.wpack_gs, d.bias_gs = n, L.GROUP_SHARED_X, packs.stride(0) * packs.element_size() // 4, bs.stride(0)
fn = lib.risp_conv2d_toep_first_exact
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_uint, C.c_void_p]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    assert fn(C.byref(d), w32.data_ptr(), w32.stride(0), buf.data_ptr(), 0xABCD, None) == 0, lib.risp_last_error()
e0.record()
for _ in range(10):
    fn(C.byref(d), w32.data_ptr(), w32.stride(0), buf.data_ptr(), 0xABCD, None)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
t = buf[:nwg * 64].view(nwg, 8, 8).double()
cons, prod = t[:, :4], t[:, 4:]
life = cons[..., 4].median().item()
phases = G * n * ((w + 31) // 32) * (h // 4 + 1) / nwg
print('9x9 3 -> 64, %d x 3 x %d x %d: %.0f us per launch (stamped build, with the tie-recompute launch); wave life %.0f cycles = %.0f per phase (the 216 16x16x32 matrix instructions of a team every other phase = 3456 cycles); clock ~%.2f GHz'
      % (G * n, h, w, us, life, life / phases, life / us / 1e3))
for nm, grp in (('team 0 (waves 0-3)', cons), ('team 1 (waves 4-7)', prod)):
    print('  %s: item preamble (weights, strip maximum) %.3f, barrier wait %.3f, products + expansion of input rows %.3f, epilogue %.3f of the life'
          % ((nm,) + tuple(grp[..., i].sum().item() / grp[..., 4].sum().item() for i in (0, 1, 2, 3))))
