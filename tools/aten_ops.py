#!/usr/bin/env python3
"""GPU box diagnostic: which Python lines issue the small torch (aten) launches of one search iteration.
python tools/aten_ops.py [batch n_step]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ['4', '2'])
batch, n_step = int(sys.argv[1]), int(sys.argv[2])
import torch
from collections import OrderedDict
from reconfigisp_amd.codes.models import create_model
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
opt = OrderedDict(model='darts', gpu_ids=[0], dist=False, is_train=True,
                  network_G=dict(which_model_G='SuperPruneFifteenDemosFourBayerTwo', n_step=n_step, n_modules=15,
                                 prune_threshold=0.2, module_path=None),
                  path=dict(pretrain_model_G=None, strict_load=True),
                  train=dict(lr_G=1e-4, momentum_G=0.9, lr_meta=1e-4, beta1=0.9, beta2=0.99, pixel_criterion='l2',
                             lr_scheme='MultiStepLR', lr_steps=[100000], restarts=None, restart_weights=None,
                             lr_gamma=0.5, clear_state=False))
torch.manual_seed(10)
model = create_model(opt)
a, ga = make_batch(batch, 256, 256, seed=1)
b, gb = make_batch(batch, 256, 256, seed=2)
data = (a.cuda(), ga.cuda(), b.cuda(), gb.cuda())
def step(i):
    model.feed_data(data)
    model.update_learning_rate(i, warmup_iter=-1)
    model.optimize_alphas()
    model.optimize_parameters()
step(0); step(1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(2)
    torch.cuda.synchronize()
names = collections.Counter(ev.name for ev in prof.events())
for name, k in names.most_common(70):
    print('%6d  %s' % (k, name))
by = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.name in ('aten::sum', 'aten::copy_', 'aten::add_', 'aten::mul', 'aten::contiguous', 'aten::clone'):
        frames = [f for f in (ev.stack or []) if 'reconfigisp_amd' in f]
        by[ev.name][' <- '.join(fr[-70:] for fr in frames[:2]) if frames else '(no python frame)'] += 1
for name, c in sorted(by.items(), key=lambda kv: -sum(kv[1].values())):
    print('%-14s %5d' % (name, sum(c.values())))
    for where, k in c.most_common(10):
        print('      %5d  %s' % (k, where))
