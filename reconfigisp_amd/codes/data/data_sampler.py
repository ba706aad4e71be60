"""Iteration-oriented distributed samplers (mirror of data/data_sampler.py:69-150).

One dataset serves both DARTS streams: its first half is the weight-training split, its second half
the architecture ("train-val") split.  Every epoch reshuffles a ``len(dataset) * ratio`` index space
with seed = epoch, keeps the indices of the wanted half, and rank r takes every world-th of them."""
import math

import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler


class _HalfSplitSampler(Sampler):
    second_half = False

    def __init__(self, dataset, num_replicas=None, rank=None, ratio=128):
        if num_replicas is None or rank is None:
            if not dist.is_available() or not dist.is_initialized():
                raise RuntimeError('Requires distributed package to be available')
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.dataset, self.num_replicas, self.rank, self.epoch = dataset, num_replicas, rank, 0
        self.num_samples = math.ceil((len(dataset) // 2) * ratio / num_replicas)
        self.total_size = len(dataset) * ratio

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        size, half = len(self.dataset), len(self.dataset) // 2
        lo, hi = (half, 2 * half) if self.second_half else (0, half)
        picked = [v % size for v in torch.randperm(self.total_size, generator=g).tolist() if lo <= v % size < hi]
        mine = picked[self.rank::self.num_replicas]
        assert len(mine) == self.num_samples
        return iter(mine)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class DistIterTrainSampler(_HalfSplitSampler):
    second_half = False


class DistIterValSampler(_HalfSplitSampler):
    second_half = True
