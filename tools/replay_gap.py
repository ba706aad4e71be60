#!/usr/bin/env python3
"""GPU box: per-step time of the headline forward under different launch forms (4 resident batches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reconfigisp_amd.codes.models import networks
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
from reconfigisp_amd.graphs import GraphedForward, GraphedQueue
torch.cuda.set_device(0)
net = networks.define_G({'network_G': {'which_model_G': 'OriginUniversal', 'architecture': 'Demosaic_01_sRGB_07_11_01_14',
                                       'module_path': None}}).cuda().eval()
bs = [make_batch(64, 256, 256, seed=100 + k)[0].cuda() for k in range(4)]

def timeit(fn, calls, steps_per_call):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
    for _ in range(calls): fn()
    t1 = time.perf_counter(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (calls * steps_per_call) * 1e3, (t1 - t0) / calls * 1e6

with torch.no_grad():
    q4 = GraphedQueue(net, bs)
    print('one graph x 4 batches      : %.1f us/step (host %.1f us per replay)' % timeit(lambda: q4(), 50, 4))
    qa, qb = GraphedQueue(net, bs[:2]), GraphedQueue(net, bs[2:])
    def pingpong():
        qa(); qb()
    print('two graphs x 2, alternating: %.1f us/step (host %.1f us per pair)' % timeit(pingpong, 50, 4))
    g1 = [GraphedForward(net, b) for b in bs]
    def four():
        for g in g1: g()
    print('four graphs x 1            : %.1f us/step (host %.1f us per 4)' % timeit(four, 50, 4))
    def eager():
        for b in bs: net(b)
    print('eager python forward       : %.1f us/step (host %.1f us per 4)' % timeit(eager, 50, 4))
