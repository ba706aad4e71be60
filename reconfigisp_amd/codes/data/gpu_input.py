"""Device-resident input path: raw frames live in HBM as uint16 / uint8 and are
cropped + normalised by a kernel instead of numpy on the host - the same tensors the reference's datasets
produce (``noisy`` (N,1,h,w) fp32 in [0,1] with an even-aligned RGGB crop, ``gt`` (N,3,h,w) fp32 /255)."""
import ctypes as C
import random

import torch

from ... import functional as F
from ... import lib as L


def even_crop_positions(n, n_frames, full, size, rng=random):
    """(N,3) {frame, row, col}: uniform crops snapped down to even coordinates
    (sid_sony_ratio_rggb2bgr_dataset.py:121-125 - keeps the RGGB pattern)."""
    rows = [[rng.randrange(n_frames), (rng.randint(0, full[0] - size[0]) // 2) * 2,
             (rng.randint(0, full[1] - size[1]) // 2) * 2] for _ in range(n)]
    return torch.tensor(rows, dtype=torch.int32)


def _frames(t, dtype, ndim):
    if not t.is_cuda or t.dtype != dtype or t.dim() != ndim:
        raise ValueError('expected a %s CUDA tensor with %d dims, got %s %s' % (dtype, ndim, t.dtype, tuple(t.shape)))
    return t.contiguous()


def raw_crops(frames_u16, sel, size, white_level=1023.0):
    """frames (F,H0,W0) uint16 on the GPU, sel (N,3) int32 -> (N,1,h,w) fp32 = sample / white_level."""
    frames = _frames(frames_u16, torch.uint16, 3)
    sel = sel.to(device=frames.device, dtype=torch.int32).contiguous()
    n = sel.shape[0]
    out = torch.empty((n, 1, size[0], size[1]), device=frames.device, dtype=torch.float32)
    L.call('risp_raw_crop', C.c_void_p(frames.data_ptr()), F._p(out), C.c_void_p(sel.data_ptr()), n, frames.shape[1],
           frames.shape[2], size[0], size[1], float(white_level), F._stream())
    return out


def gt_crops(frames_u8, sel, size):
    """frames (F,H0,W0,3) uint8 HWC BGR on the GPU -> (N,3,h,w) fp32 / 255."""
    frames = _frames(frames_u8, torch.uint8, 4)
    sel = sel.to(device=frames.device, dtype=torch.int32).contiguous()
    n = sel.shape[0]
    out = torch.empty((n, 3, size[0], size[1]), device=frames.device, dtype=torch.float32)
    L.call('risp_gt_crop', C.c_void_p(frames.data_ptr()), F._p(out), C.c_void_p(sel.data_ptr()), n, frames.shape[1],
           frames.shape[2], size[0], size[1], F._stream())
    return out


def resize_rggb_letterbox(frame_u16, desired_size=1024):
    """OnePlus pre-processing (oneplus_rggb2obj_dataset.py:109-145): width -> desired_size, height scaled with it
    (multiple of 4), planes resized by nearest neighbour, zero rows above and below -> (desired, desired) uint16.
    Returns (image, top) with `top` the number of padded rows above."""
    if not frame_u16.is_cuda or frame_u16.dtype != torch.uint16 or frame_u16.dim() != 2:
        raise ValueError('expected a 2-D uint16 CUDA frame')
    frame = frame_u16.contiguous()
    h0, w0 = frame.shape
    rh = h0 * desired_size // w0
    rh -= rh % 4
    top = (desired_size - rh) // 2
    out = torch.empty((desired_size, desired_size), device=frame.device, dtype=torch.uint16)
    L.call('risp_resize_rggb', C.c_void_p(frame.data_ptr()), C.c_void_p(out.data_ptr()), h0, w0, desired_size,
           desired_size, rh, (top // 2) * 2, F._stream())
    return out, top
