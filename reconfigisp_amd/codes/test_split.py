#!/usr/bin/env python3
"""Full-frame tiled inference (reference test_split.py:57-142): ``python test_split.py --opt X.yml``
with datasets.test.patch_size / patch_stride.  The frame is cut into overlapped tiles, every tile goes
through the pipeline, and the last stage's outputs are blended back with the linear edge-ramp mask.

Unlike the reference's one-tile-at-a-time host loop (numpy crop -> H2D -> forward -> D2H per tile), the
frame stays in HBM: one gather kernel makes the (T,C,h,w) tile batch, the pipeline runs on tile
batches, one blend kernel writes the frame."""
import argparse
import logging
import os
import os.path as osp
import random
import sys

if __package__ in (None, ''):
    sys.path.insert(0, osp.abspath(osp.join(osp.dirname(__file__), os.pardir, os.pardir)))
    __package__ = 'reconfigisp_amd.codes'

import numpy as np
import torch

from .data import create_dataloader, create_dataset
from .models import create_model
from .options import options as option
from .test import as_three, write_ppm
from .utils import util
from .utils.util_path_restore import blend_tiles, gather_tiles, tile_grid


TILE_STREAMS = int(os.environ.get('RISP_TILE_STREAMS', '2'))      # 1 = everything on the current stream
_SIDE = {}


def run_frame(model, frame, size, stride, tile_batch=16, rank=0, world=1, gather=gather_tiles, blend=blend_tiles,
              collective=False):
    """frame: (1,C,H,W) tensor.  Returns the blended (1,3,H,W) output of the last pipeline stage.

    ``world > 1`` (one process per GPU, torch.distributed initialised): the tiles are independent, so rank r runs
    tiles r, r + world, ... ; the last-stage tiles are exchanged with ONE all_gather (RCCL over xGMI: 63 x 3 MB for a
    3000 x 4000 frame) and every rank blends the frame - no other communication.  The result is bit-identical to
    the single-process run: a tile's arithmetic does not depend on which tiles share its batch.  A rank may own no tile
    at all (fewer tiles than ranks): it contributes padding.  ``collective``: take the all_gather path even when
    world == 1 (the RCCL self-test on a one-GPU box, tests/test_gpu_rccl.py)."""
    dev = model.device
    img = frame[0].to(dev)
    _, H, W = img.shape
    positions = tile_grid(H, W, size, stride)
    mine = np.arange(rank, len(positions), world)
    tiles = gather(img, positions[mine], size) if len(mine) else img.new_zeros((0, img.shape[0]) + tuple(int(v) for v in size))
    if rank <= 0:
        print('Split into {} patches'.format(len(positions)))
    # Tile batches are independent: consecutive batches alternate between two HIP streams, so the convolution launches of
    # one batch fill the prologue / store-drain gaps of the other's (each launch runs its workgroups in lockstep
    # rounds).  Same kernels on the same data: the result does not change.
    main = torch.cuda.current_stream() if img.is_cuda else None
    side = None
    if main is not None and TILE_STREAMS > 1 and len(mine) > tile_batch:
        side = _SIDE.setdefault(img.device.index, torch.cuda.Stream(device=img.device))
        tiles.record_stream(side)
    # every tile batch writes its last stage straight into its rows of ONE stack (round 6: the torch.cat of the batches' outputs was
    # 4 x 90 us of a 13 ms frame)
    local = img.new_empty((len(mine), 3) + tuple(int(v) for v in size)) if len(mine) else img.new_zeros((0, 3) + tuple(int(v) for v in size))
    netG = getattr(model, 'netG', None)
    for k, at in enumerate(range(0, len(mine), tile_batch)):
        chunk = tiles[at: at + tile_batch]
        dest = local[at: at + tile_batch]
        stream = side if (side is not None and k % 2) else main
        if k == 1 and side is not None:
            # first use of the side stream: AFTER batch 0 has been issued on the main stream - the model fills its lazily
            # built caches (packed weights, per-image parameter blocks) with launches on the stream of their first use
            # and later batches find them by host-side keys, so those launches must be ordered before the side stream
            side.wait_stream(main)
        if netG is not None:
            netG.__dict__['final_out'] = dest if dest.is_cuda else None
        try:
            if stream is None:
                model.feed_data((chunk, chunk))        # dummy ground truth, as in the reference (:93)
                _, mids = model.test()
            else:
                with torch.cuda.stream(stream):
                    model.feed_data((chunk, chunk))
                    _, mids = model.test()
                    if mids[-1].data_ptr() != dest.data_ptr():
                        dest.copy_(mids[-1])
            if stream is None and mids[-1].data_ptr() != dest.data_ptr():
                dest.copy_(mids[-1])
        finally:
            if netG is not None:
                netG.__dict__.pop('final_out', None)
    if side is not None:
        local.record_stream(side)
        main.wait_stream(side)
    if world > 1 or collective:
        import torch.distributed as dist
        per = (len(positions) + world - 1) // world
        if local.shape[0] < per:                   # ranks at the tail own one tile less: pad to a common shape
            local = torch.cat([local, local.new_zeros((per - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous())
        merged = local.new_empty((len(positions),) + tuple(local.shape[1:]))
        for r, part in enumerate(parts):
            idx = np.arange(r, len(positions), world)
            merged[torch.as_tensor(idx, device=merged.device)] = part[:len(idx)]
        local = merged
    return blend(local, positions, (H, W), stride).unsqueeze(0)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--opt', type=str, help='Path to option YAML file.')
    ap.add_argument('--tile_batch', type=int, default=16, help='tiles per forward (the reference uses 1)')
    ap.add_argument('--launcher', choices=['none', 'pytorch'], default='none',
                    help='pytorch: one process per GPU (torch.distributed.run); the tiles of a frame are sharded over the ranks')
    args = ap.parse_args(argv)
    rank, world = 0, 1
    if args.launcher == 'pytorch':
        import torch.distributed as dist
        from .train import init_dist
        init_dist('pytorch')
        rank, world = dist.get_rank(), dist.get_world_size()
    opt = option.parse(args.opt, is_train=False)
    util.mkdirs(p for k, p in opt['path'].items()
                if p and k not in ('experiments_root', 'strict_load', 'root') and 'pretrain_model' not in k
                and 'resume' not in k)
    util.setup_logger('base', opt['path']['log'], 'test_' + opt['name'], level=logging.INFO, screen=True, tofile=True)
    logger = logging.getLogger('base')
    logger.info(option.dict2str(opt))

    loaders = []
    for _, dopt in sorted(opt['datasets'].items()):
        loaders.append(create_dataloader(create_dataset(dopt), dopt))
    model = create_model(opt)
    seed = opt.get('test_seed')
    util.set_random_seed(random.randint(1, 10000) if seed is None else seed)
    size = (opt['datasets']['test']['patch_size'],) * 2
    stride = (opt['datasets']['test']['patch_stride'],) * 2

    for loader in loaders:
        name = loader.dataset.opt['mode']
        out_dir = osp.join(opt['path']['results_root'], name)
        util.mkdir(out_dir)
        psnr_in, psnr_out = [], []
        for idx, data in enumerate(loader):
            print('Image No. {}'.format(idx + 1))
            merged = run_frame(model, data['noisy'], size, stride, args.tile_batch, rank, world)
            if rank > 0:
                continue
            out_u8 = (np.clip(merged[0].permute(1, 2, 0).cpu().numpy(), 0, 1) * 255.).astype(np.uint8)
            img_in, img_gt = as_three(util.tensor2bgr(data['noisy'])), util.tensor2bgr(data['gt'])
            psnr_in.append(util.psnr(img_in, img_gt))
            psnr_out.append(util.psnr(out_u8, img_gt))
            write_ppm(osp.join(out_dir, str(data['name'][0]) + '_out.ppm'), np.concatenate([img_in, out_u8, img_gt], axis=1))
        if rank > 0:
            continue
        for tag, v in (('in', np.asarray(psnr_in)), ('out', np.asarray(psnr_out))):
            print('PSNR {}: min {}, max {}, mean {}, std {}'.format(tag, v.min(), v.max(), v.mean(), v.std()))


if __name__ == '__main__':
    main()
