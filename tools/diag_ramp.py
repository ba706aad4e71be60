#!/usr/bin/env python3
"""GPU box: per-10-step device time of the headline forward right after a synchronize (is there a ramp?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconfigisp_amd.codes.models import networks
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
net = networks.define_G({'network_G': {'which_model_G': 'OriginUniversal', 'architecture': 'Demosaic_01_sRGB_07_11_01_14',
                                       'module_path': None}}).cuda().eval()
bs = [make_batch(64, 256, 256, seed=100 + k)[0].cuda() for k in range(4)]
import gc
gc.collect(); gc.disable()
with torch.no_grad():
    for idle_ms in (0, 0, 0, 0, 5, 0, 50, 0):
        if idle_ms: 
            for k in range(40): net(bs[k % 4])
        torch.cuda.synchronize()
        time.sleep(idle_ms / 1e3)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
        evs[0].record()
        for k in range(400):
            net(bs[k % 4])
            if k % 10 == 9: evs[k // 10 + 1].record()
        torch.cuda.synchronize()
        t = [evs[i].elapsed_time(evs[i + 1]) / 10 * 1e3 for i in range(40)]
        print('idle %2d ms before: us/step per 10-step group: %s ... mean first 200 steps %.1f, last 200 %.1f'
              % (idle_ms, ' '.join('%.0f' % v for v in t[:12]), sum(t[:20]) / 20, sum(t[20:]) / 20))
