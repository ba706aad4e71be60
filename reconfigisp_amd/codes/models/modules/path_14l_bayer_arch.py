"""Path14lBayer - 14-layer Path-Restore denoiser in the Bayer domain.

Mirror of models/modules/path_14l_bayer_arch.py:6-88 (state-dict keys
``path_restore_14l.0``, ``path_restore_14l.1.<k>.basic.{1,3}``, ``path_restore_14l.3``).
Reference quirks kept: the block's first ReLU is in-place, so the skip adds relu(x) (:9-21);
there is no outer residual (:86).
"""
import torch.nn as nn

from .... import functional as F


class ResidualBlock(nn.Module):
    """Parameter holder: ReLU, conv3x3, ReLU, conv3x3 (+ relu(x))."""

    def __init__(self, inchannel, outchannel, shortcut=None):
        super().__init__()
        if shortcut is not None:
            raise NotImplementedError('projection shortcuts are never used by the reference pipelines')
        self.basic = nn.Sequential(nn.ReLU(inplace=True), nn.Conv2d(inchannel, outchannel, 3, 1, 1),
                                   nn.ReLU(inplace=True), nn.Conv2d(outchannel, outchannel, 3, 1, 1))


def _path_restore_stack(cio, shuffle):
    body = nn.Sequential(*[ResidualBlock(64, 64) for _ in range(6)])
    tail = [nn.Conv2d(cio, 64, 3, 1, 1), body, nn.ReLU(inplace=True), nn.Conv2d(64, cio, 3, 1, 1)]
    if shuffle:
        tail.append(nn.PixelShuffle(2))
    return nn.Sequential(*tail)


class Path14lBayer(nn.Module):
    def __init__(self, param_channel):
        super().__init__()
        if param_channel:
            raise NotImplementedError('Path14lBayer is only instantiated with 0 parameter channels')
        self.path_restore_14l = _path_restore_stack(4, shuffle=True)

    def forward(self, x, param_vec=None):
        if x.shape[2] % 2 or x.shape[3] % 2:
            raise ValueError('H and W must be even, got %s' % (tuple(x.shape),))
        return F.path14l_bayer(x, self)
