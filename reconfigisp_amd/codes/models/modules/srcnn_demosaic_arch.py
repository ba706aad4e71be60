"""SRCNNDemosaic - proxy for bilinear / Laplacian demosaicking.

Mirror of models/modules/srcnn_demosaic_arch.py:6-55: RGGB mosaic (N,1,H,W) -> space-to-depth
-> conv9x9(4->64) ReLU conv1x1(64->32) ReLU conv5x5(32->12) -> PixelShuffle(2) -> (N,3,H,W).
State-dict keys ``srcnn.{0,2,4}.*`` as in the reference.
"""
import torch.nn as nn

from .... import functional as F
from .srcnn_res_arch import _srcnn_stack


class SRCNNDemosaic(nn.Module):
    def __init__(self, param_channel):
        super().__init__()
        if param_channel:
            raise NotImplementedError('the registry only instantiates SRCNNDemosaic with 0 parameter channels '
                                      '(super_prune_fifteen_demos_four_bayer_two.py:43-44)')
        self.srcnn = _srcnn_stack(4, [(64, 9), (32, 1), (12, 5)], shuffle=True)

    def forward(self, x, param_vec=None):
        if x.shape[2] % 2 or x.shape[3] % 2:
            raise ValueError('H and W must be even, got %s' % (tuple(x.shape),))
        return F.srcnn_demosaic(x, self)
