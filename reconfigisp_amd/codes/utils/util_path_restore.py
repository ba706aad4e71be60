"""Overlapped tiling for full-frame inference - mirror of utils/util_path_restore.py:47-134.

``whole2patch`` / ``patch2whole`` / ``create_patch_mask`` keep the reference's numpy signatures
(HWC arrays, positions, count map) but the gather / blend run on the GPU
(risp_tile_gather / risp_tile_blend); ``tile_positions``, ``gather_tiles`` and ``blend_tiles``
are the device-resident forms used by test_split.py so a frame never leaves HBM between stages.
"""
import ctypes as C

import numpy as np
import torch

from ... import functional as F
from ... import lib as L


def get_mse_psnr(x, y):
    if x.ndim == 4:
        pairs = [get_mse_psnr(a, b) for a, b in zip(x, y)]
        return np.asarray([p[0] for p in pairs]), np.asarray([p[1] for p in pairs])
    if x.ndim == 3:
        mse = np.mean((x - y) ** 2)
        return mse, 10 * np.log10(1. / mse)
    raise ValueError('Invalid data!')


def tile_positions(full, size, stride):
    """range(0, full-size, stride) + [full-size]  (:88-89)"""
    return list(range(0, full - size, stride)) + [full - size]


def tile_grid(H, W, size, stride):
    """int32 (T,2) array of (y,x) tile origins, row-major like the reference's nested loops."""
    return np.asarray([[y, x] for y in tile_positions(H, size[0], stride[0])
                       for x in tile_positions(W, size[1], stride[1])], dtype=np.int32)


def _edges(size, stride):
    return (size[0] - stride[0]) // 2, (size[1] - stride[1]) // 2


def create_patch_mask(size, edge):
    """ones with (i+1)/(e+1) ramps over the outer e rows / columns (min of both)  (:47-64)"""
    (h, w), (eh, ew) = size, edge
    assert eh <= h // 2 and ew <= w // 2
    ramp = lambda n, e: np.minimum(np.minimum(np.arange(1, n + 1), np.arange(n, 0, -1)) / np.float64(e + 1),
                                   1.0).astype(np.float32) if e > 0 else np.ones(n, np.float32)
    return np.minimum(ramp(h, eh)[:, None], ramp(w, ew)[None, :]).astype(np.float32)


def gather_tiles(img, positions, size):
    """img (C,H,W) device tensor -> (T,C,h,w) device tensor."""
    img = F._dev(img)
    c, H, W = img.shape
    pos = torch.as_tensor(np.ascontiguousarray(positions, dtype=np.int32), device=img.device)
    out = torch.empty((len(positions), c, size[0], size[1]), device=img.device, dtype=torch.float32)
    L.call('risp_tile_gather', F._p(img), F._p(out), C.c_void_p(pos.data_ptr()), len(positions), c, H, W,
           size[0], size[1], F._stream())
    return out


def blend_tiles(patches, positions, full, stride):
    """patches (T,C,h,w) device tensor -> (C,H,W): sum(patch*mask)/sum(mask) with the edge-ramp mask."""
    patches = F._dev(patches)
    t, c, h, w = patches.shape
    eh, ew = _edges((h, w), stride)
    pos = torch.as_tensor(np.ascontiguousarray(positions, dtype=np.int32), device=patches.device)
    out = torch.empty((c, full[0], full[1]), device=patches.device, dtype=torch.float32)
    L.call('risp_tile_blend', F._p(patches), F._p(out), C.c_void_p(pos.data_ptr()), t, c, full[0], full[1], h, w,
           eh, ew, F._stream())
    return out


def whole2patch(img, size, stride, is_mask=True):
    """HWC numpy image -> (patches (T,h,w,C), positions (T,2), count_map (H,W))."""
    H, W, ch = img.shape
    (h, w), (sh, sw) = size, stride
    assert sh <= h <= H and sw <= w <= W and ch >= 1
    positions = tile_grid(H, W, size, stride)
    mask = create_patch_mask(size, _edges(size, stride)) if is_mask else np.ones(size, np.float32)
    count_map = np.zeros((H, W), np.float32)
    for y, x in positions:
        count_map[y:y + h, x:x + w] += mask
    dev = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 0, 1)), dtype=np.float32)).cuda()
    patches = gather_tiles(dev, positions, size).permute(0, 2, 3, 1).cpu().numpy()
    return patches, positions.astype(np.int64), count_map


def patch2whole(patches, positions, count_map, stride, is_mask=True):
    """inverse of whole2patch: (T,h,w,C) numpy patches -> blended HWC image."""
    if not is_mask:
        raise NotImplementedError('unmasked blending is never used by the reference drivers')
    H, W = count_map.shape
    dev = torch.from_numpy(np.ascontiguousarray(np.transpose(patches, (0, 3, 1, 2)), dtype=np.float32)).cuda()
    return blend_tiles(dev, positions, (H, W), stride).permute(1, 2, 0).cpu().numpy()
