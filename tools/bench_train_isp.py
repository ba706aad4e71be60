#!/usr/bin/env python3
"""GPU box: IspModel.optimize_parameters (fixed-pipeline training step: forward, L2, backward, Adam) on batch 64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from reconfigisp_amd.codes.models import create_model
from reconfigisp_amd.codes.data.synthetic_raw import make_batch
arch = sys.argv[1] if len(sys.argv) > 1 else 'Bayer_02_Demosaic_01_sRGB_11_01_13_14'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
opt = OrderedDict(model='isp', gpu_ids=[0], dist=False, is_train=True,
                  network_G=dict(which_model_G='IspUniversal', architecture=arch, individual_module_paths=[None] * 8, module_path=None),
                  path=dict(pretrain_model_G=None, strict_load=True),
                  train=dict(lr_G=1e-3, beta1=0.9, beta2=0.99, pixel_criterion='l2', lr_scheme='MultiStepLR', lr_steps=[100000],
                             restarts=None, restart_weights=None, lr_gamma=0.5, clear_state=False))
model = create_model(opt)
bay, gt = make_batch(n, 256, 256, seed=1)
data = (bay.cuda(), gt.cuda())
def step(i):
    model.feed_data(data); model.update_learning_rate(i); model.optimize_parameters()
for i in range(5): step(i)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(50): step(i + 5)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
print('%s batch %d: %.3f ms/step, %.1f MPix/s, loss %.5f' % (arch, n, dt * 1e3, n * 65536 / dt / 1e6, model.log_dict['loss']))
