#!/usr/bin/env python3
"""GPU box: time the literal 5-stage pipeline (nearest demosaic, bilateral, WbManual, Gamma, GtmManual)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconfigisp_amd.codes.models import networks
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
from reconfigisp_amd.graphs import GraphedForward
opt = {'network_G': {'which_model_G': 'OriginUniversal', 'architecture': 'Demosaic_01_sRGB_07_11_01_14', 'module_path': None}}
net = networks.define_G(opt).cuda().eval()
bay = make_batch(64, 256, 256, seed=10)[0].cuda()
g = GraphedForward(net, bay)
for fn, name in ((lambda: g(), 'graph'), (lambda: net(bay), 'eager')):
    with torch.no_grad():
        for _ in range(10): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 200
    print(name, 'us/step %.1f  GPix/s %.1f' % (dt * 1e6, 64 * 65536 / dt / 1e9))
