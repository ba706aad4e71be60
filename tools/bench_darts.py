#!/usr/bin/env python3
"""GPU box: one DARTS search iteration (optimize_alphas + optimize_parameters = 5 forwards + 5 backwards of
the super-net), BASELINE.json config 3: batch 32, 256x256, n_step=3, prune_threshold 0.2, alpha = 0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from reconfigisp_amd.codes.models import create_model
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
if os.environ.get('RISP_BENCH_NO_TAPOUT') == '1':          # A/B on one box: the 3-cout layers on the Toeplitz-band kernel of round 4
    import reconfigisp_amd.convnets as CN
    CN.small_has_tapout = lambda *a: False

if os.environ.get('RISP_BENCH_NO_NARROW3') == '1':         # A/B on one box: Path-Restore's 3x3 tails back on the vector kernel
    import reconfigisp_amd.convnets as CN
    CN.small_has_narrow3 = lambda *a: False
if os.environ.get('RISP_BENCH_NO_THIN5') == '1':           # A/B on one box: the 5x5 3 -> 32 backward-data layers on the fp32 F(4,5) kernel of round 5
    import reconfigisp_amd.convnets as CN
    _kinds = CN.pack_kinds
    CN.pack_kinds = lambda *a, **k: [x for x in _kinds(*a, **k) if x != 'thin5']
if os.environ.get('RISP_BENCH_STREAM_PIXELS'):             # A/B: planes from which the ops of a slot use two streams (product: 2^16 pixels)
    from reconfigisp_amd.codes.models.modules import super_prune_fifteen_demos_four_bayer_two as SP
    SP.SLOT_STREAMS_MIN_PIXELS = int(os.environ['RISP_BENCH_STREAM_PIXELS'])
if os.environ.get('RISP_BENCH_STREAM_JOBS'):               # A/B: jobs of a slot from which its ops use two streams (product: 3)
    from reconfigisp_amd.codes.models.modules import super_prune_fifteen_demos_four_bayer_two as SP
    SP.SLOT_STREAMS_MIN_JOBS = int(os.environ['RISP_BENCH_STREAM_JOBS'])
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_step = int(sys.argv[3]) if len(sys.argv) > 3 else 3
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
opt = OrderedDict(model='darts', gpu_ids=[0], dist=False, is_train=True,
                  network_G=dict(which_model_G='SuperPruneFifteenDemosFourBayerTwo', n_step=n_step, n_modules=15,
                                 prune_threshold=0.2, module_path=None),
                  path=dict(pretrain_model_G=None, strict_load=True),
                  train=dict(lr_G=1e-4, momentum_G=0.9, lr_meta=1e-4, beta1=0.9, beta2=0.99, pixel_criterion='l2',
                             lr_scheme='MultiStepLR', lr_steps=[100000], restarts=None, restart_weights=None,
                             lr_gamma=0.5, clear_state=False))
torch.manual_seed(10)
model = create_model(opt)
a, ga = make_batch(batch, size, size, seed=1)
b, gb = make_batch(batch, size, size, seed=2)
data = (a.cuda(), ga.cuda(), b.cuda(), gb.cuda())
FLOP_FWD = {2: 6.12e6, 3: 9.05e6}.get(n_step, 0) * batch * size * size
def step(i):
    model.feed_data(data)
    model.update_learning_rate(i, warmup_iter=-1)
    model.optimize_alphas()
    model.optimize_parameters()
step(0)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(iters):
    step(i + 1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / iters
print('search step: batch %d %dx%d n_step %d: %.3f s/step, %.2f MPix/s (train+val pixels), ~%.1f TFLOP/s of conv work '
      '(10 x forward FLOPs), loss %.4f, peak mem %.1f GB'
      % (batch, size, size, n_step, dt, 2 * batch * size * size / dt / 1e6, 10 * FLOP_FWD / dt / 1e12,
         model.log_dict['loss'], torch.cuda.max_memory_allocated() / 1e9))
