"""Train-time loader with the dataset's frames resident in HBM.

The reference's datasets crop and normalise on the host, one sample per ``__getitem__``
(data/sid_sony_ratio_rggb2bgr_dataset.py:119-136), and the loader ships float32 batches over PCIe.  With 288 GB of
HBM the uint16 mosaics and uint8 ground truths of a whole training set stay on the device; a batch is then two kernel
launches (``risp_raw_crop`` / ``risp_gt_crop``, codes/data/gpu_input.py): even-aligned random crops, ``/ white level``
and ``/ 255`` - the same tensors the host path produces, bit for bit (integer gathers and one fp32 division).

Selected with ``device_resident: true`` in a train dataset's options (RGGB2BGR modes, frames of one size); it yields
``{'noisy': (B,1,h,w), 'gt': (B,3,h,w)}`` CUDA tensors, which ``model.feed_data`` takes as they are."""
import random

import numpy as np
import torch

from . import gpu_input


class DeviceCropLoader:
    def __init__(self, dataset, batch_size, indices, device, n_batches=None):
        self.size = int(dataset.opt['data_size'])
        self.white = float(dataset.white)
        self.batch = int(batch_size)
        frames = sorted(set(int(i) for i in indices))
        if not frames:
            raise ValueError('DeviceCropLoader: no frames selected')
        raws, gts = [], []
        for i in frames:
            noisy, gt, _ = dataset._frames(i)
            raws.append(np.ascontiguousarray(noisy[:, :, 0]).astype(np.uint16))      # lmdb frames are int16 >= 0
            gts.append(np.ascontiguousarray(gt))
        if len({r.shape for r in raws}) != 1:
            raise ValueError('DeviceCropLoader needs frames of one size; found %s' % sorted({r.shape for r in raws}))
        self.raw = torch.from_numpy(np.stack(raws)).to(device)                       # (F,H,W) uint16
        self.gt = torch.from_numpy(np.stack(gts)).to(device)                         # (F,H,W,3) uint8, BGR
        self.full = tuple(self.raw.shape[1:])
        self.n_batches = int(n_batches) if n_batches is not None else max(1, len(frames) // self.batch)
        self.last_selection = None

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        for _ in range(self.n_batches):
            sel = gpu_input.even_crop_positions(self.batch, self.raw.shape[0], self.full, (self.size, self.size), random)
            self.last_selection = sel                                                # (B,3): frame, row, col
            yield {'noisy': gpu_input.raw_crops(self.raw, sel, (self.size, self.size), self.white),
                   'gt': gpu_input.gt_crops(self.gt, sel, (self.size, self.size))}
