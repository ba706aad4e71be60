"""SRCNNRes - residual SRCNN proxy for the classical sRGB operators (reinhard, crysisengine,
filmic, whiteworld, bilateral, median, fastnlm, bm3d).

Mirror of the reference interface (models/modules/srcnn_res_arch.py:6-53): same class name,
constructor, ``forward(x, param_vec)`` and state_dict keys (``srcnn.{0,2,4}.{weight,bias}``)
so the published ``*_G.pth`` proxy weights load unchanged.  The ``nn.Conv2d`` objects only
own the parameters; the arithmetic runs in the MFMA convolution kernels.
"""
import torch.nn as nn

from .... import functional as F


def _srcnn_stack(cin, specs, shuffle=False):
    layers, c = [], cin
    for i, (cout, k) in enumerate(specs):
        layers.append(nn.Conv2d(c, cout, k, stride=1, padding=k // 2))
        if i + 1 < len(specs):
            layers.append(nn.ReLU())
        c = cout
    if shuffle:
        layers.append(nn.PixelShuffle(2))
    return nn.Sequential(*layers)


class SRCNNRes(nn.Module):
    def __init__(self, param_channel):
        super().__init__()
        self.param_channel = param_channel
        # 3 image planes + per-image min/mean/max (9) + P broadcast parameter planes
        self.srcnn = _srcnn_stack(3 + 9 + param_channel, [(64, 9), (32, 5), (3, 5)])

    def forward(self, x, param_vec):
        if self.param_channel and (param_vec is None or param_vec.shape[1] != self.param_channel):
            raise ValueError((tuple(x.shape), None if param_vec is None else tuple(param_vec.shape)))
        return F.srcnn_res(x, param_vec, self)
