"""DartsFtModel - DARTS search with online proxy fine-tuning (mirror of models/darts_ft_model.py:20-368).

On top of DartsModel: every weight step appends the detached sRGB-domain slot outputs to a FIFO replay
memory (:194-201); ``finetune_proxies()`` (called by the driver every proxy_ft_params.ft_interval iterations)
trains each flagged proxy of the LAST sRGB slot for ft_steps Adam steps on a random memory entry with random
parameters against its classical teacher (the Origin* HIP stencils), then copies the weights into every slot.
The proxies' weight gradients come from risp_conv2d_wgrad.  Distributed: the reference wraps each proxy in DDP;
here its ~0.5 MB gradient is averaged with one flat all-reduce per step (RCCL)."""
import random

import torch

from .darts_model import DartsModel
from .modules import tools_origin as T

_TEACHERS = {'reinhard': (T.OriginToneReinhard, 2), 'crysisengine': (T.OriginToneCrysis, 1),
             'filmic': (T.OriginToneFilmic, 2), 'whiteworld': (T.OriginWbWhiteworld, 1),
             'bilateral': (T.OriginNoiseBilateral, 3), 'median': (T.OriginNoiseMedian, 1),
             'fastnlm': (T.OriginNoiseFastnlm, 3)}


class DartsFtModel(DartsModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.ft_nets, self.ft_data = [], []
        if not self.is_train:
            return
        ft = opt['proxy_ft_params']
        self.memory_size, self.ft_steps = ft['memory_size'], ft['ft_steps']
        self.param_num_dict = {name: p for name, (_, p) in _TEACHERS.items()}
        t = opt['train']
        for (name, enabled), proxy in zip(self.netG_attr.proxy_ft_flag, self.netG_attr.all_modules[-1]):
            if not enabled:
                continue
            proxy.train()
            optimizer = torch.optim.Adam(proxy.parameters(), lr=t['lr_G'], betas=(t['beta1'], t['beta2']))
            # [name, net_proxy, net_proxy_attr, net_target, optimizer] as in the reference (:98)
            self.ft_nets.append([name, proxy, proxy, _TEACHERS[name][0]().to(self.device), optimizer])

    def save(self, iter_label):
        super().save(iter_label)
        for name, _, proxy, _, _ in self.ft_nets:
            self.save_network(proxy, name, iter_label)

    def optimize_parameters(self):
        super().optimize_parameters()
        # replay memory: sRGB-domain slot outputs of this step, first in first out (kept in HBM)
        self.ft_data.extend(m.detach().clone() for m in self.netG_attr.intermediate_results if m.size(1) == 3)
        if len(self.ft_data) > self.memory_size:
            del self.ft_data[:len(self.ft_data) - self.memory_size]

    def finetune_proxies(self):
        if not self.is_train:
            return
        tuned = {}
        for name, proxy, proxy_attr, teacher, optimizer in self.ft_nets:
            if not self.ft_data:
                print('[Warning] Data is not ready for proxy fine-tuning!')
                continue
            proxy_attr.train_weights = True              # route through the weight-gradient path
            try:
                for _ in range(self.ft_steps):
                    data = self.ft_data[int(random.random() * len(self.ft_data))].to(self.device)
                    param = torch.rand(1, self.param_num_dict[name]).repeat(data.size(0), 1).to(self.device)
                    output = proxy(data, param)
                    with torch.no_grad():
                        target = teacher(data, param)
                    loss = self.cri_pix(output, target)
                    optimizer.zero_grad()
                    loss.backward()
                    self._allreduce_mean([p.grad for p in proxy_attr.parameters()])
                    optimizer.step()
                    self.log_dict['ft_loss_' + name] = loss.item()
            finally:
                proxy_attr.train_weights = False
            tuned[name] = proxy_attr
        self.netG_attr.load_proxy_nets(tuned)
