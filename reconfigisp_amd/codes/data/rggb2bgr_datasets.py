"""The reference's four file-backed RGGB -> BGR datasets (data/sid_sony_ratio_rggb2bgr_dataset.py,
sid_sony_ratio_test_rggb2bgr_dataset.py, s7isp_rggb2bgr_dataset.py, s7isp_rggb2bgr_test_dataset.py) behind their own
``mode`` names and option keys (``dataroot``, ``data_type``, ``data_size``, ``sid_expo_in`` / ``sid_expo_gt``).

Same contract: ``meta_info.pkl`` in ``dataroot`` lists the frame keys (``keys_ratio`` / ``keys_noisy``, ``keys_gt``,
``resolution``); an item is ``{'noisy': (1,h,w) float32, 'gt': (3,h,w) float32[, 'name']}`` with the mosaic divided by
its white level (16383 for SID, 1023 for S7-ISP), the ground truth by 255, and every crop snapped to even coordinates
so that the RGGB phase survives.  The four reference classes differ only in four constants, so they are ONE class
here, configured per mode.

``data_type``: 'img' reads .png / .npy frames (data/image_io.py: cv2 is not in this image); 'lmdb' needs the ``lmdb``
module (the reference's raw-buffer layout, data/util.py:13-21); 'mc' (SenseTime's memcached client) raises."""
import os.path as osp
import pickle
import random

import numpy as np
import torch.utils.data as data

from .image_io import read_image

# mode -> (meta key of the noisy frames, white level, train (random crop) or test (whole frame), exposure filter)
_MODES = {
    'SID_Sony_Ratio_RGGB2BGR': ('keys_ratio', 16383., True, True),
    'SID_Sony_Ratio_Test_RGGB2BGR': ('keys_ratio', 16383., False, True),
    'S7ISP_RGGB2BGR': ('keys_noisy', 1023., True, False),
    'S7ISP_RGGB2BGR_Test': ('keys_noisy', 1023., False, False),
}


class Rggb2BgrDataset(data.Dataset):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.noisy_key, self.white, self.train, has_expo = _MODES[opt['mode']]
        self.data_type = opt['data_type']
        if self.data_type == 'mc':
            raise NotImplementedError("data_type 'mc' needs SenseTime's memcached client; use 'img' or 'lmdb'")
        if self.data_type not in ('img', 'lmdb'):
            raise ValueError('Invalid data type {}'.format(self.data_type))
        self.lmdb_env = None
        with open(osp.join(opt['dataroot'], 'meta_info.pkl'), 'rb') as f:
            meta = pickle.load(f)
        self.keys_noisy, self.keys_gt = list(meta[self.noisy_key]), list(meta['keys_gt'])
        self.image_size = meta.get('resolution')
        expo_in, expo_gt = opt.get('sid_expo_in'), opt.get('sid_expo_gt')
        if has_expo and not (expo_in is None and expo_gt is None):        # keep the pairs of one exposure setting
            keep = [(a, b) for a, b in zip(self.keys_noisy, self.keys_gt) if expo_in in a and expo_gt in b]
            self.keys_noisy, self.keys_gt = [a for a, _ in keep], [b for _, b in keep]

    def __len__(self):
        return len(self.keys_gt)

    # ---- frame access
    def _lmdb(self, key, shape, dtype):
        if self.lmdb_env is None:
            import lmdb                                                   # optional dependency
            self.lmdb_env = lmdb.open(self.opt['dataroot'], readonly=True, lock=False, readahead=False, meminit=False)
        with self.lmdb_env.begin(write=False) as txn:
            buf = txn.get(key.encode('ascii'))
        c, h, w = shape
        return np.frombuffer(buf, dtype=dtype).reshape(h, w, c)

    def _frames(self, index):
        key_noi, key_gt = self.keys_noisy[index], self.keys_gt[index]
        if self.data_type == 'lmdb':
            s = self.image_size
            noisy, gt = self._lmdb(key_noi, (1, s, s), np.int16), self._lmdb(key_gt, (3, s, s), np.uint8)
        else:
            noisy = read_image(osp.join(self.opt['dataroot'], key_noi))
            gt = read_image(osp.join(self.opt['dataroot'], key_gt))
            if noisy.ndim == 2:
                noisy = noisy[:, :, None]                                 # HW -> HWC
        return noisy, gt[:, :, :3], key_noi

    def __getitem__(self, index):
        noisy, gt, key = self._frames(index)
        size = self.opt.get('data_size')
        if self.train:                                                    # random crop on even coordinates
            full = self.image_size if self.image_size is not None else min(noisy.shape[:2])
            r = (random.randint(0, full - size) // 2) * 2
            c = (random.randint(0, full - size) // 2) * 2
            noisy, gt = noisy[r:r + size, c:c + size], gt[r:r + size, c:c + size]
        elif self.white == 16383.:                                        # SID test: top-left corner if data_size is set
            if size is not None:
                noisy, gt = noisy[:size, :size], gt[:size, :size]
        else:                                                             # S7-ISP test: whole frame, even sizes
            h, w = noisy.shape[0] - noisy.shape[0] % 2, noisy.shape[1] - noisy.shape[1] % 2
            noisy, gt = noisy[:h, :w], gt[:h, :w]
        item = {'noisy': np.transpose(noisy, (2, 0, 1)).astype(np.float32) / np.float32(self.white),
                'gt': np.transpose(gt, (2, 0, 1)).astype(np.float32) / np.float32(255.)}
        if not self.train:
            item['name'] = osp.splitext(osp.basename(key))[0]
        return item
