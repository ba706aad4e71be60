"""Path14lBgr - 14-layer Path-Restore denoiser on BGR images (trained on RGB; the channel flips
of models/modules/path_14l_bgr_arch.py:59,84 are folded into the packed weights)."""
import torch.nn as nn

from .... import functional as F
from .path_14l_bayer_arch import ResidualBlock, _path_restore_stack  # noqa: F401


class Path14lBgr(nn.Module):
    def __init__(self, param_channel):
        super().__init__()
        if param_channel:
            raise NotImplementedError('Path14lBgr is only instantiated with 0 parameter channels')
        self.path_restore_14l = _path_restore_stack(3, shuffle=False)

    def forward(self, x, param_vec=None):
        return F.path14l_bgr(x, self)
