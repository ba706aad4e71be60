#!/usr/bin/env python3
"""GPU box: the convolution layers of the SRCNNRes family as the grouped launches of a search step issue them (8 members,
super_prune_fifteen_demos_four_bayer_two.py:35-52; srcnn_res_arch.py:15-24), one by one, with the ceiling that bounds
each: matrix-core layers against the FLOPs their instructions issue at the 157.3 TFLOP/s fp32 MFMA peak (155 measured,
tools/mfma_peak.hip), the direct small-cout layers against their packed-FMA FLOPs at the same peak, and every layer
against its HBM bytes at 8 TB/s.  Kernel durations come from the rocprofv3 trace (tools/profile_ops.sh); this prints the
wall time per call.   python tools/bench_layers.py [images_per_member=4] [H=256] [W=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconfigisp_amd import convnets as CN
from reconfigisp_amd import lib as L

n, h, w = (int(v) for v in (sys.argv[1:4] + ['4', '256', '256'][len(sys.argv) - 1:]))
G = 8
REPS = int(os.environ.get('RISP_OPS_REPS', '0')) or 20
torch.manual_seed(0)
dev = 'cuda'
rnd = lambda *s: torch.randn(*s, device=dev)


def layer(cout, cin, k):
    return [CN.PackedConv(rnd(cout, cin, k, k) * 0.05, rnd(cout) * 0.01) for _ in range(G)]


first = layer(64, 3, 9)
for pc in first:
    pc.k3 = CN.k3_weights(rnd(64, 3, 9, 9) * 0.05)
c2, c3 = layer(32, 64, 5), layer(3, 32, 5)
img, c2s, c3s = CN.stack_packed(first), CN.stack_packed(c2), CN.stack_packed(c3)
tail = CN.stack_small([CN.SmallConv(rnd(3, 32, 5, 5) * 0.05, rnd(3) * 0.01) for _ in range(G)])
bwd_img = CN.stack_small([CN.SmallConv(rnd(64, 12, 9, 9) * 0.05, None, transpose=True, keep=3) for _ in range(G)])
x = torch.rand(n, 3, h, w, device=dev)
table = rnd(G * n, 64 * 81) * 0.01
t1, t2 = torch.rand(G * n, 64, h, w, device=dev), torch.rand(G * n, 32, h, w, device=dev)
gy = rnd(G * n, 3, h, w)
g2, g1 = rnd(G * n, 32, h, w), rnd(G * n, 64, h, w)
pix = G * n * h * w
PEAK = 157.3e12


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


print('== grouped launches: %d members x %d images of %d x %d (%.2f MPix per launch)' % (G, n, h, w, pix / 1e6))
print('%-58s %9s %8s %8s %8s' % ('layer (kernel)', 'us', 'TFLOP/s', 'of peak', 'of HBM'))
rows = [
    ('9x9 3->64 forward, case table + ReLU (conv_lin_kernel)', 2 * 244 * 64, 4 * (3 / G + 64),
     lambda: CN.conv(x, img, n, h, w, epi=CN.EPI_RELU | CN.EPI_CASEBIAS, cvals=table, group=(G, L.GROUP_SHARED_X))),
    ('5x5 64->32 forward F(4,5) (conv_wino45_r2_kernel)', 2 * 10 * 64 * 32, 4 * (64 + 32),
     lambda: CN.conv(t1, c2s, n, h, w, epi=CN.EPI_RELU, group=(G, 0))),
    ('5x5 32->3 forward + residual (conv_small_kernel<5>)', 2 * 25 * 32 * 4, 4 * (32 + 3 / G + 3),
     lambda: CN.conv_small(t2, tail, n, h, w, epi=CN.EPI_ADD, add=x, add_c=3, group=(G, L.GROUP_SHARED_ADD))),
    ('5x5 3->32 backward-data + mask F(4,5) (conv_wino45_r2_kernel)', 2 * 10 * 4 * 32, 4 * (3 + 32 + 32),
     lambda: CN.conv(gy, c3s, n, h, w, transpose=True, epi=CN.EPI_MASK, mask=t2, group=(G, 0))),
    ('5x5 32->64 backward-data + mask F(4,5) (conv_wino45_r2_kernel)', 2 * 10 * 32 * 64, 4 * (32 + 64 + 64),
     lambda: CN.conv(g2, c2s, n, h, w, transpose=True, epi=CN.EPI_MASK, mask=t1, group=(G, 0))),
    ('9x9 64->3 backward-data + residual (conv_small_kernel<9>)', 2 * 81 * 64 * 4, 4 * (64 + 3 + 3),
     lambda: CN.conv_small(g1, bwd_img, n, h, w, epi=CN.EPI_ADD, add=gy, add_c=3, group=(G, 0))),
]
for name, flop_pix, bytes_pix, fn in rows:
    us = timed(fn)
    flops = flop_pix * pix / (us * 1e-6)
    print('%-58s %9.1f %8.1f %8.3f %8.3f' % (name, us, flops / 1e12, flops / PEAK, bytes_pix * pix / (us * 1e-6) / 8e12), flush=True)
