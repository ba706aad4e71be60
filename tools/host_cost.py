#!/usr/bin/env python3
"""GPU box: where does the host time of one eager headline forward go?  (cProfile over 2000 forwards)"""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconfigisp_amd.codes.models import networks
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
which = sys.argv[2] if len(sys.argv) > 2 else 'OriginUniversal'
net = networks.define_G({'network_G': {'which_model_G': which, 'architecture': sys.argv[1] if len(sys.argv) > 1 else 'Demosaic_01_sRGB_07_11_01_14',
                                       'module_path': None, 'individual_module_paths': [None] * 8}}).cuda().eval()
x = make_batch(64, 256, 256, seed=1)[0].cuda()
with torch.no_grad():
    for _ in range(20): net(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):                      # un-queued: the GPU is idle when the call is made
        torch.cuda.synchronize()
        t0 = time.perf_counter(); net(x); ts.append(time.perf_counter() - t0)
    print('host time of one forward call (GPU idle): median %.1f us' % (sorted(ts)[15] * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): net(x)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
