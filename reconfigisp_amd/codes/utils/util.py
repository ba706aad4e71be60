"""Helpers of the drivers (loggers, folders, seeding, YAML) plus the two functions that DEFINE "PSNR" for
this project: ``tensor2bgr`` clips and TRUNCATES to uint8, ``psnr`` uses peak 1.0 (the reference's
utils/util.py:118-154, pinned by golden vectors).  ``psnr_tensors`` evaluates the same metric on the GPU
(risp_sse_uint8).  Neither cv2 nor torchvision is needed here."""
import logging
import math
import os
import random
import time
from collections import OrderedDict

import numpy as np
import torch
import yaml

try:
    from yaml import CDumper as Dumper, CLoader as Loader
except ImportError:  # pragma: no cover
    from yaml import Dumper, Loader


# ---------------------------------------------------------------- YAML / folders / logging
def OrderedYaml():
    """Loader / Dumper pair that keeps mapping order (OrderedDict in, OrderedDict out)."""
    Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                           lambda loader, node: OrderedDict(loader.construct_pairs(node)))
    Dumper.add_representer(OrderedDict, lambda dumper, data: dumper.represent_dict(data.items()))
    return Loader, Dumper


def get_timestamp():
    return time.strftime('%y%m%d-%H%M%S')


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def mkdirs(paths):
    for path in ((paths,) if isinstance(paths, str) else paths):
        mkdir(path)


def mkdir_and_rename(path):
    """fresh folder; an existing one is archived under a timestamped name first"""
    if os.path.exists(path):
        archived = '{}_archived_{}'.format(path, get_timestamp())
        note = 'Path already exists. Rename it to [{:s}]'.format(archived)
        print(note)
        logging.getLogger('base').info(note)
        os.rename(path, archived)
    os.makedirs(path)


def set_random_seed(seed):
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def setup_logger(logger_name, root, phase, level=logging.INFO, screen=False, tofile=False):
    log = logging.getLogger(logger_name)
    log.setLevel(level)
    handlers = []
    if tofile:
        handlers.append(logging.FileHandler(os.path.join(root, '{}_{}.log'.format(phase, get_timestamp())), mode='w'))
    if screen:
        handlers.append(logging.StreamHandler())
    for h in handlers:
        h.setFormatter(logging.Formatter('%(asctime)s.%(msecs)03d - %(levelname)s: %(message)s', '%y-%m-%d %H:%M:%S'))
        log.addHandler(h)


# ---------------------------------------------------------------- tensors <-> images, metrics
def state2tensor(state):
    """NHWC 10-bit array -> NCHW float tensor in [0,1]"""
    return torch.from_numpy(np.ascontiguousarray(state.astype(np.float32).transpose(0, 3, 1, 2) / 1023.))


def tensor2state(tensor):
    """NCHW float tensor -> NHWC int16 10-bit codes (truncated, negatives floored to 0)"""
    codes = (tensor.numpy().transpose(0, 2, 3, 1) * 1023).astype(np.int16)
    return np.maximum(codes, 0)


def tensor2bgr(tensor, is_uint8=True):
    """(1,C,H,W) or (C,H,W) tensor in [0,1] -> (H,W,C) BGR array; uint8 = clip then truncate."""
    chw = tensor.detach().cpu().numpy()
    chw = chw[0] if chw.ndim == 4 else chw
    hwc = chw.transpose(1, 2, 0)
    if is_uint8:
        hwc = np.clip(hwc * 255, 0, 255).astype(np.uint8)
    return np.array(hwc)        # C-ordered copy


def _unit_range(img):
    if img.dtype == np.uint8:
        return img.astype(np.float32) / 255.
    if img.dtype == np.int16:       # 10-bit codes
        return img.astype(np.float32) / 1023.
    return img


def psnr(img1, img2):
    mse = np.mean((_unit_range(img1) - _unit_range(img2)) ** 2)
    return float('inf') if mse == 0 else 10 * math.log10(1. / mse)


def psnr_tensors(a, b):
    """psnr(tensor2bgr(a), tensor2bgr(b)) for device tensors of equal shape, without leaving the GPU."""
    import ctypes as C
    from ... import functional as F
    from ... import lib as L
    a, b = F._dev(a), F._dev(b)
    if a.shape != b.shape:
        raise ValueError('shape mismatch %s vs %s' % (tuple(a.shape), tuple(b.shape)))
    sse = torch.empty(L.load().risp_sse_uint8_doubles(), device=a.device, dtype=torch.float64)
    L.call('risp_sse_uint8', F._p(a), F._p(b), C.c_void_p(sse.data_ptr()), sse.numel(), a.numel(), F._stream())
    mse = sse[0].item() / a.numel()
    return float('inf') if mse == 0 else 10 * math.log10(1. / mse)     # identical images: inf, like numpy's 1./0.
