#!/usr/bin/env python3
"""GPU box diagnostic: the config-3 super-net with the ops of a slot FORCED onto two streams from two jobs (the policy round 6 tried and
dropped): forward + two backward passes of the same graph, repeated with allocator churn in between; counts the repetitions whose two
backward passes differ in a bit.  python tools/stream_stress.py [repetitions] [min_jobs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch
from reconfigisp_amd.codes.models.modules import super_prune_fifteen_demos_four_bayer_two as SP
from test_host_logic import build_supernet
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SP.SLOT_STREAMS_MIN_JOBS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
net = build_supernet(3, 'cuda')
bay, gt = make_batch(32, 256, 256, seed=7)
bay, gt = bay.cuda(), gt.cuda()
named = dict(net.named_parameters())
keys = [k for k in sorted(named) if named[k].requires_grad]
g = torch.Generator().manual_seed(0)
bad = 0
for r in range(reps):
    junk = [torch.empty(int(torch.randint(1 << 16, 1 << 26, (1,), generator=g)), device='cuda') for _ in range(6)]      # allocator churn
    del junk
    y = net(bay)
    gy = (y.detach() - gt) * (2.0 / y.numel())
    g1 = torch.autograd.grad(y, [named[k] for k in keys], gy, retain_graph=True, allow_unused=True)
    g1b = torch.autograd.grad(y, [named[k] for k in keys], gy, allow_unused=True)
    diff = [k for k, a, b in zip(keys, g1, g1b) if a is not None and not torch.equal(a, b)]
    if diff:
        bad += 1
        print('repetition %d: %d gradients differ: %s' % (r, len(diff), diff[:6]), flush=True)
    del y, gy, g1, g1b
print('%d of %d repetitions with two backward passes that differ (streams from %d jobs)' % (bad, reps, SP.SLOT_STREAMS_MIN_JOBS))
