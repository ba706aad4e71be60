#!/usr/bin/env python3
"""GPU box: bench.py's exact pre-timing sequence (spin-up with periodic synchronize, W warm-up steps, synchronize)
followed by per-10-step device times."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconfigisp_amd.codes.models import networks
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
net = networks.define_G({'network_G': {'which_model_G': 'OriginUniversal', 'architecture': 'Demosaic_01_sRGB_07_11_01_14',
                                       'module_path': None}}).cuda().eval()
bs = [make_batch(64, 256, 256, seed=100 + k)[0].cuda() for k in range(4)]
gc.collect(); gc.disable()
i = [0]
def fn():
    net(bs[i[0] % 4]); i[0] += 1
with torch.no_grad():
    for spin_sync in (20, 200, 0):
        time.sleep(0.3)                                   # idle, as after building the networks
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.04:
            for _ in range(spin_sync or 20): fn()
            if spin_sync: torch.cuda.synchronize()
        for _ in range(20): fn()
        torch.cuda.synchronize(); torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
        evs[0].record()
        for k in range(400):
            fn()
            if k % 10 == 9: evs[k // 10 + 1].record()
        torch.cuda.synchronize()
        t = [evs[j].elapsed_time(evs[j + 1]) / 10 * 1e3 for j in range(40)]
        print('spin-up sync every %3d: %s ... first 200: %.1f, last 200: %.1f' % (spin_sync, ' '.join('%.0f' % v for v in t[:14]), sum(t[:20]) / 20, sum(t[20:]) / 20))
