#!/usr/bin/env python3
"""Search / training driver - same command line and loop order as the reference's train.py:58-301:

    python train.py --opt options/train/X.yml                       # single GPU
    python -m torch.distributed.run --nproc-per-node N train.py --opt X.yml --launcher pytorch

Per iteration: feed (img, gt, val_img, val_gt) -> update_learning_rate -> [optimize_alphas] ->
optimize_parameters; first half of the dataset feeds the weight step, second half the architecture
step.  One process per GPU; ``backend='nccl'`` is RCCL on ROCm."""
import argparse
import logging
import math
import os
import random
import sys
import time

if __package__ in (None, ''):
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir, os.pardir)))
    __package__ = 'reconfigisp_amd.codes'

import torch
import torch.distributed as dist

from .data import create_dataloader, create_dataset
from .data.data_sampler import DistIterTrainSampler, DistIterValSampler
from .models import create_model
from .options import options as option
from .utils import util


def init_dist(launcher, backend='nccl', port=29500):
    if launcher == 'pytorch':
        rank = int(os.environ['RANK'])
    elif launcher == 'slurm':
        rank = int(os.environ['SLURM_PROCID'])
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(port))
        os.environ['WORLD_SIZE'] = os.environ['SLURM_NTASKS']
        os.environ['RANK'] = str(rank)
    else:
        raise ValueError('Invalid launcher type: {}'.format(launcher))
    if torch.cuda.is_available():
        torch.cuda.set_device(rank % torch.cuda.device_count())
    else:
        backend = 'gloo'
    dist.init_process_group(backend=backend)


def batch_tuple(opt, a, b):
    if 'local_global' in opt['train']['pixel_criterion']:
        return a['noisy'], a['gt'], a['glb_flag'], b['noisy'], b['gt'], b['glb_flag']
    return a['noisy'], a['gt'], b['noisy'], b['gt']


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--opt', type=str, help='Path to option YAML file.')
    ap.add_argument('--launcher', choices=['none', 'pytorch', 'slurm'], default='none')
    ap.add_argument('--local_rank', type=int, default=0)
    ap.add_argument('--port', type=int, default=29500)
    args = ap.parse_args(argv)
    opt = option.parse(args.opt, is_train=True)

    opt['dist'] = args.launcher != 'none'
    rank = -1
    if opt['dist']:
        init_dist(args.launcher, port=args.port)
        rank = dist.get_rank()
    else:
        print('Disabled distributed training.')

    if rank <= 0:
        util.mkdir_and_rename(opt['path']['experiments_root'])
        util.mkdirs(p for k, p in opt['path'].items()
                    if p and k != 'experiments_root' and 'pretrain_model' not in k and 'resume' not in k
                    and k not in ('strict_load', 'root'))
        util.setup_logger('base', opt['path']['log'], 'train_' + opt['name'], level=logging.INFO, screen=True, tofile=True)
        util.setup_logger('val', opt['path']['log'], 'val_' + opt['name'], level=logging.INFO, screen=True, tofile=True)
    else:
        util.setup_logger('base', opt['path']['log'], 'train', level=logging.INFO, screen=True)
    logger = logging.getLogger('base')
    if rank <= 0:
        logger.info(option.dict2str(opt))

    seed = opt['train']['manual_seed']
    util.set_random_seed(random.randint(1, 10000) if seed is None else seed)

    dopt = opt['datasets']['train']
    train_set = create_dataset(dopt)
    total_iters = int(opt['train']['niter'])
    iters_per_epoch = max(1, int(math.ceil(len(train_set) // 2 / dopt['batch_size'])))
    if rank == -1:
        logger.info('Number of train/val images: {:,d}, iters: {:,d}'.format(len(train_set) // 2, iters_per_epoch))
    if opt['dist']:
        samplers = (DistIterTrainSampler(train_set, dist.get_world_size(), rank),
                    DistIterValSampler(train_set, dist.get_world_size(), rank))
    else:
        half = len(train_set) // 2
        idx = list(range(len(train_set)))
        samplers = (torch.utils.data.sampler.SubsetRandomSampler(idx[:half]),
                    torch.utils.data.sampler.SubsetRandomSampler(idx[half:]))
    loaders = [create_dataloader(train_set, dopt, opt, s) for s in samplers]

    model = create_model(opt)
    kind = opt['model']
    print_freq = opt['logger']['print_freq']
    save_freq = opt['logger']['save_checkpoint_freq']
    step, epoch, tick = 0, 0, time.time()
    while step <= total_iters:
        epoch += 1
        if opt['dist']:
            for s in samplers:
                s.set_epoch(epoch)
        for trn, val in zip(*loaders):
            step += 1
            if step > total_iters:
                break
            model.feed_data(batch_tuple(opt, trn, val))
            model.update_learning_rate(step, warmup_iter=opt['train']['warmup_iter'])
            if kind == 'darts_ft' and step % opt['proxy_ft_params']['ft_interval'] == 0:
                model.finetune_proxies()
                if rank <= 0:
                    print('proxy nets fine-tuned!')
            if kind in ('darts', 'darts_ft'):
                model.optimize_alphas()
            model.optimize_parameters()

            if rank <= 0 and step % print_freq == 0:
                print('Average time per iter: {:.6f}'.format((time.time() - tick) / print_freq))
                msg = '<epoch:{:3d}, iter:{:8,d}, lr:{:.3e}> '.format(epoch, step, model.get_current_learning_rate())
                msg += ' '.join('{:s}: {:.4e}'.format(k, v) for k, v in model.get_current_log().items())
                if kind in ('darts', 'darts_ft'):
                    print('train_val_loss: {}'.format(model.val_loss.item()))
                    print('Pruned paths: {}'.format(model.netG_attr.pruned_paths))
                logger.info(msg)
                tick = time.time()
            if rank <= 0 and step % save_freq == 0:
                logger.info('Saving models and training states.')
                model.save(step)
                model.save_training_state(epoch, step)
    if opt['dist']:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
