#!/usr/bin/env python3
"""GPU box diagnostic: where the host time of one registry forward of the headline pipeline goes (cProfile over 3000 calls)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
dev = torch.device('cuda')
net = bench.build_pipeline('Demosaic_01_sRGB_07_11_01_14', dev, which='OriginUniversal')
bay = make_batch(64, 256, 256, seed=10)[0].to(dev)
with torch.no_grad():
    for _ in range(200):
        net(bay)
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for _ in range(3000):
        net(bay)
    host = (time.perf_counter() - t) / 3000 * 1e6
    torch.cuda.synchronize()
    print('host issue %.1f us per forward (GPU-bound loop: includes back-pressure)' % host)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3000):
        net(bay)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:50]))
