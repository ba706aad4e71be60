#!/usr/bin/env python3
"""GPU box diagnostic: host-side profile (cProfile) of one DARTS search iteration at a given geometry (default: the reference's shipped
one - SID_search.yml: batch 4, 48 x 48, n_step 3), wall time with and without the profiler, time of an iteration whose launches are
not waited for (host issue time alone).  python tools/host_profile_darts.py [batch size n_step]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from reconfigisp_amd.codes.models import create_model
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
a = [int(v) for v in sys.argv[1:]]
batch, size, n_step = (a + [4, 48, 3])[:3] if len(a) >= 3 else (4, 48, 3)
opt = OrderedDict(model='darts', gpu_ids=[0], dist=False, is_train=True,
                  network_G=dict(which_model_G='SuperPruneFifteenDemosFourBayerTwo', n_step=n_step, n_modules=15, prune_threshold=0.2, module_path=None),
                  path=dict(pretrain_model_G=None, strict_load=True),
                  train=dict(lr_G=1e-4, momentum_G=0.9, lr_meta=1e-4, beta1=0.9, beta2=0.99, pixel_criterion='l2',
                             lr_scheme='MultiStepLR', lr_steps=[100000], restarts=None, restart_weights=None, lr_gamma=0.5, clear_state=False))
torch.manual_seed(10)
model = create_model(opt)
x, g = make_batch(batch, size, size, seed=1)
y, h = make_batch(batch, size, size, seed=2)
data = (x.cuda(), g.cuda(), y.cuda(), h.cuda())
def step(i):
    model.feed_data(data); model.update_learning_rate(i, warmup_iter=-1); model.optimize_alphas(); model.optimize_parameters()
for i in range(5): step(i)
torch.cuda.synchronize(); t = time.perf_counter()
N = 30
for i in range(N): step(i + 5)
torch.cuda.synchronize(); print('%.2f ms per iteration (batch %d, %dx%d, n_step %d)' % ((time.perf_counter() - t) / N * 1e3, batch, size, size, n_step))
pr = cProfile.Profile(); pr.enable()
for i in range(N): step(i + 100)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45)
print('\n'.join(l[:170] for l in s.getvalue().splitlines()[:62]))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40)
print('\n'.join(l[:170] for l in s.getvalue().splitlines()[:55]))
