#!/usr/bin/env python3
"""GPU box: in-process A/B of BUILDS of the wave-specialised split-precision kernel (risp_conv_f16x2_ws.hip + risp_conv_f16x2.hip) with
different -D flags (or "src=tools/other_ws.hip [-D...]": another source in its place), interleaved rounds on the same tensors, every epilogue; the first build is the reference for the bits.
python tools/ab_ws_build.py "" "-DWS_SPLIT=0" ...   [env RISP_AB_SHAPE="n h w", RISP_AB_CH="cin cout", RISP_AB_K=3|5, RISP_AB_EPIS="0 1 2 3 4"]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reconfigisp_amd import lib as L, convnets as CN
variants = sys.argv[1:] or ['', '-DWS_SPLIT=0']
csrc = os.path.join(ROOT, 'reconfigisp_amd/csrc')
libs = []
for i, v in enumerate(variants):
    so = '/tmp/ab_ws_%d.so' % i
    flags = v.split()
    ws = os.path.join(csrc, 'risp_conv_f16x2_ws.hip')
    if flags and flags[0].startswith('src='):                    # another source file in place of risp_conv_f16x2_ws.hip
        ws, flags = os.path.join(ROOT, flags[0][4:]), flags[1:]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
                           '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-x', 'hip', '-shared', '-o', so] + flags +
                          [os.path.join(csrc, 'risp_conv_f16x2.hip'), ws, os.path.join(csrc, 'risp_core.cpp')])
    lib = C.CDLL(so)
    lib.risp_conv2d_f16x2.restype, lib.risp_conv2d_f16x2.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
    libs.append(lib)
n, h, w = (int(v) for v in os.environ.get('RISP_AB_SHAPE', '32 256 256').split())
K = int(os.environ.get('RISP_AB_K', '3'))
cin, cout = (int(v) for v in os.environ.get('RISP_AB_CH', '64 64' if K == 3 else '64 32').split())
torch.manual_seed(0)
wt = torch.randn(cout, cin, K, K, device='cuda') * 0.05
b = torch.randn(cout, device='cuda') * 0.01
x = torch.rand(n, cin, h, w, device='cuda')
add, mask = torch.rand(n, cout, h, w, device='cuda'), torch.randn(n, cout, h, w, device='cuda')
y = torch.empty(n, cout, h, w, device='cuda')
pack = CN.f16x2_weights(wt, False)
flop = 3 * 2.0 * cin * cout * K * K * n * h * w
EPIS = (('plain', 0), ('relu', CN.EPI_RELU), ('add+relu', CN.EPI_ADD | CN.EPI_RELU), ('mask', CN.EPI_MASK), ('add+mask', CN.EPI_ADD | CN.EPI_MASK))
for idx in (int(v) for v in os.environ.get('RISP_AB_EPIS', '0 2 3 4').split()):
    what, epi = EPIS[idx]
    d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=K, load_mode=0, cin_img=0, epilogue=epi, add_c=cout if epi & CN.EPI_ADD else 0,
                   x=x.data_ptr(), wpack=pack.data_ptr(), bias=b.data_ptr(), cvals=None, add=add.data_ptr() if epi & CN.EPI_ADD else None,
                   mask=mask.data_ptr() if epi & CN.EPI_MASK else None, y=y.data_ptr())
    outs, res = [], [[] for _ in libs]
    for lib in libs:
        y.fill_(float('nan'))
        assert lib.risp_conv2d_f16x2(C.byref(d), None) == 0
        torch.cuda.synchronize()
        outs.append(y.clone())
    for rnd in range(5):
        for i, lib in enumerate(libs):
            for _ in range(2):
                lib.risp_conv2d_f16x2(C.byref(d), None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10):
                lib.risp_conv2d_f16x2(C.byref(d), None)
            e1.record(); e1.synchronize()
            res[i].append(e0.elapsed_time(e1) / 10 * 1e3)
    med = [sorted(r)[2] for r in res]
    print('%-9s %dx%d %dx%dx%dx%d->%d: ' % (what, K, K, n, cin, h, w, cout) +
          ' | '.join('[%s] %.1f us (min %.1f) x%.3f %.2f of peak, bits %s nan %d' % (variants[i] or 'default', med[i], min(res[i]), med[0] / med[i],
                                                                                      flop / med[i] / 1e6 / 2516.6, bool(torch.equal(outs[0], outs[i])),
                                                                                      int(torch.isnan(outs[i]).sum())) for i in range(len(libs))))
