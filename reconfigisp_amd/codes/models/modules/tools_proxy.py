"""Weight-loading subclasses of the learned operators (mirror of models/modules/tools_proxy.py).

``load_path=None`` keeps the constructor's random initialisation (the reference's own
behaviour at :25-26); a path is loaded with the reference's 'module.' prefix stripping
(:28-39) and raises if the file is absent, exactly like ``torch.load`` there.
"""
import logging
from collections import OrderedDict

import torch

from .path_14l_bayer_arch import Path14lBayer
from .path_14l_bgr_arch import Path14lBgr
from .srcnn_demosaic_arch import SRCNNDemosaic
from .srcnn_res_arch import SRCNNRes


class _Loadable:
    def _init_weights(self, load_path, strict_load):
        self.logger = logging.getLogger('base')
        if load_path is not None:
            self.load(load_path, strict_load)

    def load(self, load_path, strict_load):
        self.logger.info('Loading model for ProxyNet [{:s}] ...'.format(load_path))
        state = torch.load(load_path, map_location='cpu')
        clean = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in state.items())
        self.load_state_dict(clean, strict=strict_load)


class ProxyNet(SRCNNRes, _Loadable):
    def __init__(self, param_channel, load_path, strict_load=True):
        super().__init__(param_channel)
        self._init_weights(load_path, strict_load)


class ProxyDemosaicNet(SRCNNDemosaic, _Loadable):
    def __init__(self, param_channel, load_path, strict_load=True):
        super().__init__(param_channel)
        self._init_weights(load_path, strict_load)


class PathRestore14lBayer(Path14lBayer, _Loadable):
    def __init__(self, param_channel, load_path, strict_load=True):
        super().__init__(param_channel)
        self._init_weights(load_path, strict_load)


class PathRestore14lBgr(Path14lBgr, _Loadable):
    def __init__(self, param_channel, load_path, strict_load=True):
        super().__init__(param_channel)
        self._init_weights(load_path, strict_load)
