"""Synthetic RAW source (there is no network for datasets): smooth random scene -> RGGB mosaic ->
Poisson-Gaussian noise -> 10-bit codes k/1023, the shape / scale contract of the reference's
datasets (fp32 (N,1,H,W) in [0,1], even H/W, RGGB; GT = clean BGR k/255;
data/oneplus_rggb2obj_dataset.py:201, data/util.py:46-49)."""
import numpy as np
import torch
import torch.utils.data as data


def make_batch(n, h, w, seed=10, bits=10):
    """-> (noisy bayer (N,1,H,W), gt (N,3,H,W)) CPU fp32 tensors."""
    if h % 2 or w % 2:
        raise ValueError('H and W must be even')
    rng = np.random.default_rng(seed)
    peak = float(2 ** bits - 1)
    # band-limited scene: a few random low-frequency cosines per channel
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing='ij')
    scene = np.empty((n, 3, h, w), np.float32)
    for i in range(n):
        for c in range(3):
            f = rng.uniform(0.5, 4.0, size=(4, 2)).astype(np.float32)
            ph = rng.uniform(0, 2 * np.pi, size=4).astype(np.float32)
            amp = rng.uniform(0.05, 0.25, size=4).astype(np.float32)
            acc = np.full((h, w), rng.uniform(0.3, 0.6), np.float32)
            for k in range(4):
                acc += amp[k] * np.cos(2 * np.pi * (f[k, 0] * yy + f[k, 1] * xx) + ph[k])
            scene[i, c] = acc
    scene = np.clip(scene, 0.02, 0.98)
    gt = np.floor(scene * 255 + 0.5) / 255.
    lin = 0.5 * scene ** 2.2                                # linear sensor response, half exposure
    mosaic = np.empty((n, 1, h, w), np.float32)
    mosaic[:, 0, 0::2, 0::2] = lin[:, 2, 0::2, 0::2]        # R
    mosaic[:, 0, 0::2, 1::2] = lin[:, 1, 0::2, 1::2]        # G1
    mosaic[:, 0, 1::2, 0::2] = lin[:, 1, 1::2, 0::2]        # G2
    mosaic[:, 0, 1::2, 1::2] = lin[:, 0, 1::2, 1::2]        # B
    shot = rng.poisson(mosaic * 500.).astype(np.float32) / 500.
    noisy = shot + rng.normal(0, 0.003, size=mosaic.shape).astype(np.float32)
    codes = np.floor(np.clip(noisy, 0, 1) * peak + 0.5) / peak
    return torch.from_numpy(codes.astype(np.float32)), torch.from_numpy(gt.astype(np.float32))


class SyntheticRawDataset(data.Dataset):
    """dataset mode 'Synthetic_RGGB2BGR': items {'noisy': (1,H,W), 'gt': (3,H,W), 'name': str}."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        size = opt.get('data_size') or 256
        self.length = int(opt.get('n_images') or 64)
        self.noisy, self.gt = make_batch(self.length, size, size, seed=int(opt.get('seed') or 10))

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        idx = idx % self.length
        return {'noisy': self.noisy[idx], 'gt': self.gt[idx], 'name': 'synthetic_%04d' % idx}
