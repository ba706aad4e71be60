"""The operator registry shared by the fixed pipelines and the DARTS super-net.

Pool order and 1-based indices, parameter initialisations and proxy-weight locations are the
reference's (isp_universal.py:34-127, origin_universal.py:28-85,
super_prune_fifteen_demos_four_bayer_two.py:34-140); they are kept as DATA here so the three
graph builders cannot drift apart.
"""
from . import tools_origin as T
from . import tools_proxy as TP

_EXP = 'proxy_nets/experiments/'
# name -> (number of parameter channels, weight file below module_path)
PROXY_NETS = {
    'reinhard': (2, _EXP + '006_reinhard_residual_multistepLR2/models/400000_G.pth'),
    'crysisengine': (1, _EXP + '007_crysis_residual_multistepLR/models/400000_G.pth'),
    'filmic': (2, _EXP + '009_filmic_residual_multistepLR/models/400000_G.pth'),
    'whiteworld': (1, _EXP + '008_whiteworld_residual_multistepLR/models/400000_G.pth'),
    'bilateral': (3, _EXP + '013_bilateral_residual_multistepLR2/models/400000_G.pth'),
    'median': (1, _EXP + '010_median_residual_multistepLR/models/400000_G.pth'),
    'fastnlm': (3, _EXP + '014_fastnlm_residual_multistepLR2/models/400000_G.pth'),
    'bilinear': (0, _EXP + '015_demosaic_bilinear_multistepLR/models/400000_G.pth'),
    'laplacian': (0, _EXP + '016_demosaic_laplacian_multistepLR/models/400000_G.pth'),
    'path_bayer': (0, _EXP + '020_denoise_path_restore_14l_bayer_aug_multistepLR/models/800000_G.pth'),
    'path_bgr': (0, _EXP + '019_path_restore_14l_rgb/models/path_restore_14l_rgb.pth'),
    'bm3d': (5, _EXP + '022_bm3d_residual_multistepLR_mc/models/400000_G.pth'),
}


def _wbq_init():
    v = [0.] * 30
    for i in (6, 17, 28):      # linear diagonal: sigmoid(0.406)*10-5 ~ 1
        v[i] = 0.406
    return v


# raw (pre-sigmoid) initial parameters; [] = parameter-free
PARAM_INIT = {
    'path_bayer': [], 'skip': [], 'nearest': [], 'bilinear': [], 'laplacian': [], 'demosaicnet': [],
    'gamma': [0.], 'reinhard': [0., 0.], 'crysisengine': [0.], 'filmic': [0., 0.], 'grayworld': [],
    'whiteworld': [0.], 'bilateral': [0., 0., 0.], 'median': [0.], 'fastnlm': [0., 0., 0.],
    'wbmanual': [-1.38, -1.38, -1.38], 'path_bgr': [], 'wbquadratic': _wbq_init(),
    'gtmmanual': [-1.099, 0., 1.099],
    # BM3D: cff, n1, cspace, wtransform, neighborhood; init probs .125 .75 .25 .25 .9375
    'bm3d': [-1.946, 1.099, -1.099, -1.099, 2.708],
    'conditional_gamma': [0.], 'conditional_wb_manual': [-1.38, -1.38, -1.38],
    'conditional_wb_quadratic': _wbq_init(),
}

NAMES_BAYER = ['path_bayer', 'skip']
NAMES_DEMOSAIC = ['nearest', 'bilinear', 'laplacian', 'demosaicnet']
NAMES_SRGB = ['gamma', 'reinhard', 'crysisengine', 'filmic', 'grayworld', 'whiteworld', 'bilateral', 'median',
              'fastnlm', 'skip', 'wbmanual', 'path_bgr', 'wbquadratic', 'gtmmanual', 'bm3d']
NAMES_SRGB_EXT = NAMES_SRGB + ['conditional_gamma', 'conditional_wb_manual', 'conditional_wb_quadratic',
                               'ten_layer_net', 'two_layer_net', 'toy_net']   # 16-18 conditional, 19-21 undefined
CONDITIONAL_KW = {'conditional_gamma': 'gamma_in_channels', 'conditional_wb_manual': 'wb_manual_in_channels',
                  'conditional_wb_quadratic': 'wb_quadratic_in_channels'}

_PROXY_CLASS = {'path_bayer': TP.PathRestore14lBayer, 'path_bgr': TP.PathRestore14lBgr,
                'bilinear': TP.ProxyDemosaicNet, 'laplacian': TP.ProxyDemosaicNet}
_PLAIN_CLASS = {'skip': T.Skip, 'nearest': T.DemosaicNearest, 'demosaicnet': T.DemosaicNet, 'gamma': T.Gamma,
                'grayworld': T.Grayworld, 'wbmanual': T.WbManual, 'wbquadratic': T.WbQuadratic}
_ORIGIN_CLASS = {'bilinear': T.OriginDemosBilinear, 'laplacian': T.OriginDemosLaplacian,
                 'reinhard': T.OriginToneReinhard, 'crysisengine': T.OriginToneCrysis,
                 'filmic': T.OriginToneFilmic, 'whiteworld': T.OriginWbWhiteworld,
                 'bilateral': T.OriginNoiseBilateral, 'median': T.OriginNoiseMedian,
                 'fastnlm': T.OriginNoiseFastnlm}
_CONDITIONAL_CLASS = {'conditional_gamma': T.ConditionalGamma, 'conditional_wb_manual': T.ConditionalWbManual,
                      'conditional_wb_quadratic': T.ConditionalWbQuadratic}


def weight_path(name, module_path, override=None):
    """Default weight file of a proxy, or None when module_path is None (random init, tests/bench)."""
    if override is not None:
        return override
    return None if module_path is None else module_path + PROXY_NETS[name][1]


def make_op(name, module_path, origin=False, weight_override=None, conditional_channels=None):
    """Instantiate registry entry `name`.  origin=True selects the classical (non-proxy) kernels
    where they exist (OriginUniversal); otherwise the differentiable proxies."""
    if name in _PLAIN_CLASS:
        return _PLAIN_CLASS[name]()
    if name == 'gtmmanual':
        return T.GtmManual(4)      # hard-coded 4 segments (isp_universal.py:179)
    if name in _CONDITIONAL_CLASS:
        if conditional_channels is None:
            raise AssertionError('%s needs conditional_modules.%s in the options' % (name, CONDITIONAL_KW[name]))
        return _CONDITIONAL_CLASS[name](in_channels=tuple(conditional_channels))
    if origin and name in _ORIGIN_CLASS:
        return _ORIGIN_CLASS[name]()
    if name in PROXY_NETS:
        cls = _PROXY_CLASS.get(name, TP.ProxyNet)
        return cls(PROXY_NETS[name][0], weight_path(name, module_path, weight_override))
    raise NotImplementedError(
        'registry entry %r has no implementation (the reference references undefined classes for it, '
        'isp_universal.py:92-94)' % (name,))


def parse_architecture(architecture, srgb_names):
    """'Bayer_xx_Demosaic_xx_sRGB_xx_..' with 1-based indices -> [(domain, name), ...]."""
    pools = {'Bayer': NAMES_BAYER, 'Demosaic': NAMES_DEMOSAIC, 'sRGB': srgb_names}
    domain, steps = None, []
    for token in architecture.split('_'):
        if token in pools:
            domain = token
            continue
        if domain is None:
            raise ValueError('Domain (Bayer, Demosaic, sRGB) is not specified in ISP architecture!')
        index = int(token)
        if not 1 <= index <= len(pools[domain]):
            raise AssertionError('module index %d out of range for domain %s' % (index, domain))
        steps.append((domain, pools[domain][index - 1]))
    return steps
