#!/usr/bin/env python3
"""GPU box diagnostic: build risp_conv.hip with -DRISP_CONV_STAMPS into /tmp and print the per-wave
cycle shares (barrier wait / MFMA loop / publish) of one conv layer.  Not part of the product."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = '/tmp/librisp_stamps.so'
import glob
src = glob.glob(os.path.join(ROOT, 'reconfigisp_amd/csrc', '*.hip')) + [os.path.join(ROOT, 'reconfigisp_amd/csrc/risp_core.cpp')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
                       '-DRISP_CONV_STAMPS', '-DRISP_W43_NO_GLDS', '-DRISP_W5_NO_GLDS',      # the stamps live in the register-staged kernels
                       '-I' + os.path.join(ROOT, 'include'),
                       '-I' + os.path.join(ROOT, 'reconfigisp_amd/csrc'), '-x', 'hip', '-shared', '-o', so] + src)
import torch
from reconfigisp_amd import lib as L
L.LIB_PATH = so
from reconfigisp_amd import convnets as CN

# python tools/conv_stamps.py wino|wino43 [cin cout 3 n h w]: the F(2,3) / F(4,3) Winograd kernels
wino = len(sys.argv) > 1 and sys.argv[1] in ('wino', 'wino43')
w43 = wino and sys.argv[1] == 'wino43'
if wino:
    del sys.argv[1]
cin, cout, k, n, h, w = (int(v) for v in (sys.argv[1:7] + ['64', '64', '3', '64', '128', '128'][len(sys.argv) - 1:]))
dev = torch.device('cuda')
wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
pc = CN.PackedConv(wt, torch.zeros(cout, device=dev))
x = torch.rand(n, cin, h, w, device=dev)
nwg = (((w + 127) // 128) * ((h + 3) // 4) * n * ((cout + 31) // 32)) if w43 else \
      (((w + 63) // 64) * ((h + 3) // 4) * n) if wino else (((w + 31) // 32) * ((h + 15) // 16) * n)
stamps = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
y = torch.empty(n, cout, h, w, device=dev)
for _ in range(2):
    d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=k, load_mode=0, cin_img=0, epilogue=CN.EPI_RELU, add_c=0,
                   x=x.data_ptr(), wpack=(pc.wino43_fwd if w43 else pc.wino_fwd if wino else pc.fwd).data_ptr(), bias=pc.bias.data_ptr(), cvals=None,
                   add=None, mask=stamps.data_ptr(), y=y.data_ptr())
    L.call('risp_conv2d_wino43' if w43 else pc.wino_entry if wino else 'risp_conv2d', C.byref(d), None)
torch.cuda.synchronize()
s = stamps.view(nwg * 4, 8).cpu().double()
life_ticks = s[:, 7] - s[:, 6]
cycles = s[:, 3] + s[:, 4] + s[:, 5]
span = (s[:, 7].max() - s[:, 6].min()).item()
print('in-kernel shader clock %.3f GHz; kernel span %.1f us; mean wave life %.1f us; mean concurrent waves %.0f (of %d slots)'
      % ((cycles / life_ticks).median().item() * 0.1, span / 100., life_ticks.mean().item() / 100.,
         life_ticks.sum().item() / span, 256 * 4 * 2))
m = lambda c: s[:, c].mean().item()
tot = cycles.mean().item()
print('cycles/wave: total %.0f | prologue %.0f (%.1f%%) | chunk loop %.0f (%.1f%%) [barrier wait %.0f, mfma loop %.0f, publish %.0f] | epilogue %.0f (%.1f%%)'
      % (tot, m(3), 100 * m(3) / tot, m(4), 100 * m(4) / tot, m(0), m(1), m(2), m(5), 100 * m(5) / tot))
