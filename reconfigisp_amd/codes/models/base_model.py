"""BaseModel - LR warm-up, save / load / resume plumbing shared by the model wrappers
(mirror of models/base_model.py:8-119; same method names and checkpoint formats)."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel


def _unwrap(net):
    return net.module if isinstance(net, (nn.DataParallel, DistributedDataParallel)) else net


class BaseModel(object):
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device('cuda' if opt['gpu_ids'] is not None else 'cpu')
        self.is_train = opt['is_train']
        self.schedulers = []
        self.optimizers = []

    def feed_data(self, data):
        pass

    def optimize_parameters(self):
        pass

    def get_current_visuals(self):
        pass

    def get_current_losses(self):
        pass

    def print_network(self):
        pass

    def save(self, label):
        pass

    def load(self):
        pass

    def _set_lr(self, lr_groups_l):
        for optimizer, lr_groups in zip(self.optimizers, lr_groups_l):
            for group, lr in zip(optimizer.param_groups, lr_groups):
                group['lr'] = lr

    def _get_init_lr(self):
        return [[g['initial_lr'] for g in o.param_groups] for o in self.optimizers]

    def update_learning_rate(self, cur_iter, warmup_iter=-1):
        for s in self.schedulers:
            s.step()
        if cur_iter < warmup_iter:      # linear warm-up from 0 to the scheduler's initial lr
            self._set_lr([[v / warmup_iter * cur_iter for v in grp] for grp in self._get_init_lr()])

    def get_current_learning_rate(self):
        return self.optimizers[0].param_groups[0]['lr']

    def get_network_description(self, network):
        network = _unwrap(network)
        return str(network), sum(p.numel() for p in network.parameters())

    def save_network(self, network, network_label, iter_label):
        path = os.path.join(self.opt['path']['models'], '{}_{}.pth'.format(iter_label, network_label))
        state = OrderedDict((k, v.cpu()) for k, v in _unwrap(network).state_dict().items())
        torch.save(state, path)

    def load_network(self, load_path, network, strict=True):
        state = torch.load(load_path, map_location='cpu')
        clean = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state.items())
        _unwrap(network).load_state_dict(clean, strict=strict)

    def save_training_state(self, epoch, iter_step):
        state = {'epoch': epoch, 'iter': iter_step,
                 'schedulers': [s.state_dict() for s in self.schedulers],
                 'optimizers': [o.state_dict() for o in self.optimizers]}
        torch.save(state, os.path.join(self.opt['path']['training_state'], '{}.state'.format(iter_step)))

    def resume_training(self, resume_state):
        assert len(resume_state['optimizers']) == len(self.optimizers), 'Wrong lengths of optimizers'
        assert len(resume_state['schedulers']) == len(self.schedulers), 'Wrong lengths of schedulers'
        for o, s in zip(self.optimizers, resume_state['optimizers']):
            o.load_state_dict(s)
        for sch, s in zip(self.schedulers, resume_state['schedulers']):
            sch.load_state_dict(s)
