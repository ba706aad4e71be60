#!/usr/bin/env python3
"""GPU box: 60 DARTS search iterations (n_step 3, fresh synthetic batches every iteration, lr 1e-3) through every round-3 path
(grouped launches, fused slot mixture, step reuse, lean backward passes, F(4,5)): parameters stay finite, operators get pruned,
the loss decreases."""
import sys, os, torch, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
from collections import OrderedDict
from reconfigisp_amd.codes.models import create_model
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
opt = OrderedDict(model='darts', gpu_ids=[0], dist=False, is_train=True,
                  network_G=dict(which_model_G='SuperPruneFifteenDemosFourBayerTwo', n_step=3, n_modules=15, prune_threshold=0.2, module_path=None),
                  path=dict(pretrain_model_G=None, strict_load=True),
                  train=dict(lr_G=1e-3, momentum_G=0.9, lr_meta=1e-3, beta1=0.9, beta2=0.99, pixel_criterion='l2', lr_scheme='MultiStepLR',
                             lr_steps=[100000], restarts=None, restart_weights=None, lr_gamma=0.5, clear_state=False))
torch.manual_seed(10)
model = create_model(opt)
losses, pruned = [], []
for it in range(60):
    a, ga = make_batch(4, 128, 128, seed=2 * it)
    b, gb = make_batch(4, 128, 128, seed=2 * it + 1)
    model.feed_data((a.cuda(), ga.cuda(), b.cuda(), gb.cuda()))
    model.update_learning_rate(it, warmup_iter=-1)
    model.optimize_alphas()
    model.optimize_parameters()
    losses.append(model.log_dict['loss'])
    pruned.append(list(model.netG.pruned_paths))
    for p in model.netG.parameters():
        assert torch.isfinite(p).all(), 'non-finite parameter at iteration %d' % it
print('loss first 5', [round(v, 5) for v in losses[:5]], 'last 5', [round(v, 5) for v in losses[-5:]])
print('pruned paths first', pruned[0], 'last', pruned[-1])
print('alpha_step1', model.netG.alpha_step1.detach().cpu().numpy().round(3))
assert sum(losses[-10:]) < sum(losses[:10]), 'the loss did not decrease over 60 search iterations'
print('soak OK: 60 search iterations (n_step 3, fresh batches), finite, loss decreasing')
