"""Logging / filesystem helpers and the two metric functions that define "PSNR" for this
project (mirror of utils/util.py).  ``tensor2bgr`` truncates (not rounds) to uint8 and ``psnr``
uses peak 1.0 (:118-154) - both pinned by golden vectors; ``psnr_tensors`` is the same metric
evaluated on the device (risp_sse_uint8) without the host round trip.
cv2 / torchvision are not needed by anything here and are not imported."""
import logging
import math
import os
import random
from collections import OrderedDict
from datetime import datetime

import numpy as np
import torch
import yaml

try:
    from yaml import CDumper as Dumper, CLoader as Loader
except ImportError:  # pragma: no cover
    from yaml import Dumper, Loader


def OrderedYaml():
    """yaml <-> OrderedDict"""
    tag = yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG
    Dumper.add_representer(OrderedDict, lambda dumper, data: dumper.represent_dict(data.items()))
    Loader.add_constructor(tag, lambda loader, node: OrderedDict(loader.construct_pairs(node)))
    return Loader, Dumper


def get_timestamp():
    return datetime.now().strftime('%y%m%d-%H%M%S')


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def mkdirs(paths):
    for p in ([paths] if isinstance(paths, str) else paths):
        mkdir(p)


def mkdir_and_rename(path):
    if os.path.exists(path):
        new_name = path + '_archived_' + get_timestamp()
        print('Path already exists. Rename it to [{:s}]'.format(new_name))
        logging.getLogger('base').info('Path already exists. Rename it to [{:s}]'.format(new_name))
        os.rename(path, new_name)
    os.makedirs(path)


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def setup_logger(logger_name, root, phase, level=logging.INFO, screen=False, tofile=False):
    lg = logging.getLogger(logger_name)
    fmt = logging.Formatter('%(asctime)s.%(msecs)03d - %(levelname)s: %(message)s', datefmt='%y-%m-%d %H:%M:%S')
    lg.setLevel(level)
    if tofile:
        fh = logging.FileHandler(os.path.join(root, phase + '_{}.log'.format(get_timestamp())), mode='w')
        fh.setFormatter(fmt)
        lg.addHandler(fh)
    if screen:
        sh = logging.StreamHandler()
        sh.setFormatter(fmt)
        lg.addHandler(sh)


def state2tensor(state):
    """NHWC 10-bit numpy -> NCHW float tensor in [0,1]"""
    return torch.from_numpy(np.transpose(state.astype(np.float32) / 1023., (0, 3, 1, 2)).copy())


def tensor2state(tensor):
    """NCHW float tensor -> NHWC int16 10-bit (truncated, floored at 0)"""
    state = (np.transpose(tensor.numpy(), (0, 2, 3, 1)) * 1023).astype(np.int16)
    return np.maximum(state, 0)


def tensor2bgr(tensor, is_uint8=True):
    """1CHW / CHW tensor in [0,1] -> HWC BGR image; uint8 conversion clips then TRUNCATES."""
    image = tensor.detach().cpu().numpy()
    if image.ndim == 4:
        image = image[0]
    image = np.transpose(image, (1, 2, 0))
    if is_uint8:
        image = np.clip(image * 255, 0, 255).astype(np.uint8)
    return image.copy()


def psnr(img1, img2):
    """PSNR with peak 1.0; int16 inputs are 10-bit, uint8 inputs 8-bit."""
    def unit(a):
        if a.dtype == np.int16:
            return a.astype(np.float32) / 1023.
        if a.dtype == np.uint8:
            return a.astype(np.float32) / 255.
        return a
    mse = ((unit(img1) - unit(img2)) ** 2).mean()
    return float('inf') if mse == 0 else 10 * math.log10(1. / mse)


def psnr_tensors(a, b):
    """psnr(tensor2bgr(a), tensor2bgr(b)) evaluated on the GPU (device tensors, any matching shape)."""
    import ctypes as C
    from ... import functional as F
    from ... import lib as L
    a, b = F._dev(a), F._dev(b)
    if a.shape != b.shape:
        raise ValueError('shape mismatch %s vs %s' % (tuple(a.shape), tuple(b.shape)))
    sse = torch.empty(1, device=a.device, dtype=torch.float64)
    L.call('risp_sse_uint8', F._p(a), F._p(b), C.c_void_p(sse.data_ptr()), a.numel(), F._stream())
    mse = sse.item() / a.numel()
    return float('inf') if mse == 0 else 10 * math.log10(1. / mse)   # identical images: inf, like numpy's 1./0.
