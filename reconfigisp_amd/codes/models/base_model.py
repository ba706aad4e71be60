"""Shared behaviour of the model wrappers: learning-rate warm-up, checkpoint and training-state files
(same method names and on-disk formats as the reference's models/base_model.py:8-119, so checkpoints
written by either side load in the other)."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel

_WRAPPERS = (nn.DataParallel, DistributedDataParallel)


def _bare(network):
    return network.module if isinstance(network, _WRAPPERS) else network


def _strip_module_prefix(state):
    return OrderedDict((key[len('module.'):] if key.startswith('module.') else key, value) for key, value in state.items())


class BaseModel(object):
    def __init__(self, opt):
        self.opt, self.is_train = opt, opt['is_train']
        self.device = torch.device('cpu' if opt['gpu_ids'] is None else 'cuda')
        self.optimizers, self.schedulers = [], []

    # interface filled in by the subclasses
    def feed_data(self, data): pass
    def optimize_parameters(self): pass
    def get_current_visuals(self): pass
    def get_current_losses(self): pass
    def print_network(self): pass
    def save(self, label): pass
    def load(self): pass

    # ---- learning rate
    def _get_init_lr(self):
        return [[group['initial_lr'] for group in opt.param_groups] for opt in self.optimizers]

    def _set_lr(self, lr_groups_l):
        for opt, lrs in zip(self.optimizers, lr_groups_l):
            for group, lr in zip(opt.param_groups, lrs):
                group['lr'] = lr

    def update_learning_rate(self, cur_iter, warmup_iter=-1):
        for sched in self.schedulers:
            sched.step()
        if cur_iter < warmup_iter:                      # linear ramp towards the scheduler's initial value
            scale = cur_iter / warmup_iter
            self._set_lr([[lr * scale for lr in lrs] for lrs in self._get_init_lr()])

    def get_current_learning_rate(self):
        return self.optimizers[0].param_groups[0]['lr']

    # ---- networks on disk: '<iter>_<label>.pth' holds a CPU state dict
    def get_network_description(self, network):
        net = _bare(network)
        return str(net), sum(p.numel() for p in net.parameters())

    def save_network(self, network, network_label, iter_label):
        target = os.path.join(self.opt['path']['models'], '{}_{}.pth'.format(iter_label, network_label))
        torch.save(OrderedDict((k, v.cpu()) for k, v in _bare(network).state_dict().items()), target)

    def load_network(self, load_path, network, strict=True):
        _bare(network).load_state_dict(_strip_module_prefix(torch.load(load_path, map_location='cpu')), strict=strict)

    # ---- optimiser / scheduler state: '<iter>.state'
    def save_training_state(self, epoch, iter_step):
        blob = dict(epoch=epoch, iter=iter_step, schedulers=[s.state_dict() for s in self.schedulers],
                    optimizers=[o.state_dict() for o in self.optimizers])
        torch.save(blob, os.path.join(self.opt['path']['training_state'], '{}.state'.format(iter_step)))

    def resume_training(self, resume_state):
        for kind, mine in (('optimizers', self.optimizers), ('schedulers', self.schedulers)):
            saved = resume_state[kind]
            assert len(saved) == len(mine), 'Wrong lengths of ' + kind
            for obj, state in zip(mine, saved):
                obj.load_state_dict(state)
