#!/usr/bin/env python3
"""Per-image test driver (reference test.py:22-107): ``python test.py --opt options/test/X.yml``.
Runs model.test() on every image, reports PSNR of input and output against the ground truth with the
reference's truncating-uint8 metric, and writes [input | every stage | gt] side by side as one PNM
file per image (cv2 is not required)."""
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

from .data import create_dataloader, create_dataset
from .models import create_model
from .options import options as option
from .utils import util


def as_three(img):
    return np.concatenate([img] * 3, axis=2) if img.shape[2] == 1 else img


def write_ppm(path, bgr):
    rgb = np.ascontiguousarray(bgr[:, :, ::-1])
    with open(path, 'wb') as f:
        f.write(b'P6\n%d %d\n255\n' % (rgb.shape[1], rgb.shape[0]))
        f.write(rgb.tobytes())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--opt', type=str, help='Path to option YAML file.')
    args = ap.parse_args(argv)
    opt = option.parse(args.opt, is_train=False)
    util.mkdirs(p for k, p in opt['path'].items()
                if p and k not in ('experiments_root', 'strict_load', 'root') and 'pretrain_model' not in k
                and 'resume' not in k)
    util.setup_logger('base', opt['path']['log'], 'test_' + opt['name'], level=logging.INFO, screen=True, tofile=True)
    logger = logging.getLogger('base')
    logger.info(option.dict2str(opt))

    loaders = []
    for _, dopt in sorted(opt['datasets'].items()):
        ds = create_dataset(dopt)
        loaders.append(create_dataloader(ds, dopt))
        logger.info('Number of test images in [{:s}]: {:d}'.format(dopt['mode'], len(ds)))
    model = create_model(opt)
    seed = opt.get('test_seed')
    util.set_random_seed(random.randint(1, 10000) if seed is None else seed)

    for loader in loaders:
        name = loader.dataset.opt['mode']
        logger.info('\nTesting [{:s}]...'.format(name))
        out_dir = osp.join(opt['path']['results_root'], name)
        util.mkdir(out_dir)
        psnr_in, psnr_out = [], []
        for idx, data in enumerate(loader):
            print('Image No. {}'.format(idx + 1))
            model.feed_data((data['noisy'], data['gt']))
            out, mids = model.test()
            img_in, img_gt = as_three(util.tensor2bgr(data['noisy'])), util.tensor2bgr(data['gt'])
            psnr_in.append(util.psnr(img_in, img_gt))
            psnr_out.append(util.psnr(util.tensor2bgr(out), img_gt))
            panels = [img_in] + [as_three(util.tensor2bgr(m)) for m in mids] + [img_gt]
            write_ppm(osp.join(out_dir, '{:03d}.ppm'.format(idx + 1)), np.concatenate(panels, axis=1))
        for tag, v in (('in', np.asarray(psnr_in)), ('out', np.asarray(psnr_out))):
            print('PSNR {}: min {}, max {}, mean {}, std {}'.format(tag, v.min(), v.max(), v.mean(), v.std()))


if __name__ == '__main__':
    main()
