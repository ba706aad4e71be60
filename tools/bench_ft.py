#!/usr/bin/env python3
"""GPU box: one finetune_proxies() call of the darts_ft search (models/darts_ft_model.py:206-246): every flagged proxy of
the last sRGB slot trained for ft_steps Adam steps against its classical teacher on replay-memory batches.
python tools/bench_ft.py [batch n_step ft_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from reconfigisp_amd.codes.models import create_model
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
batch, n_step, ft_steps = (int(v) for v in (sys.argv[1:4] + ['32', '3', '5'][len(sys.argv) - 1:]))
opt = OrderedDict(model='darts_ft', gpu_ids=[0], dist=False, is_train=True,
                  network_G=dict(which_model_G='SuperPruneFifteenDemosFourBayerTwoFt', n_step=n_step, n_modules=15,
                                 prune_threshold=0.2, module_path=None),
                  proxy_ft_params=dict(memory_size=1000, ft_interval=100, ft_steps=ft_steps),
                  path=dict(pretrain_model_G=None, strict_load=True),
                  train=dict(lr_G=1e-4, momentum_G=0.9, lr_meta=1e-4, beta1=0.9, beta2=0.99, pixel_criterion='l2',
                             lr_scheme='MultiStepLR', lr_steps=[100000], restarts=None, restart_weights=None,
                             lr_gamma=0.5, clear_state=False))
torch.manual_seed(10)
import random
random.seed(10)                                  # the replay-memory entries finetune_proxies() draws
model = create_model(opt)
a, ga = make_batch(batch, 256, 256, seed=1)
b, gb = make_batch(batch, 256, 256, seed=2)
model.feed_data((a.cuda(), ga.cuda(), b.cuda(), gb.cuda()))
model.update_learning_rate(0, warmup_iter=-1)
model.optimize_alphas()
model.optimize_parameters()                      # fills the replay memory
model.finetune_proxies()                         # warm-up (packs, allocator)
torch.cuda.synchronize()
t = time.perf_counter()
model.finetune_proxies()
torch.cuda.synchronize()
dt = time.perf_counter() - t
n = len(model.ft_nets) * ft_steps
print('finetune_proxies: %d proxies x %d steps on batches of %d x 256 x 256: %.3f s = %.1f ms per proxy step (teacher, proxy '
      'forward, loss, backward with weight gradients, Adam); losses %s'
      % (len(model.ft_nets), ft_steps, batch, dt, dt / n * 1e3,
         {k[8:]: round(v, 5) for k, v in model.log_dict.items() if k.startswith('ft_loss_')}))
