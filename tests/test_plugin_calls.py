"""B1 boundary: the reference wrappers' OWN call arguments, replayed.

tests/golden/plugin_calls.npz holds, for each of the 16 ``self.kernel.run(img, option, params)`` call sites of the
reference's models/modules/tools_origin.py (:33-41 grayworld ... :775-797 fastnlm), exactly what the imported
reference passed: option string, the image with its shape / strides (permuted NHWC views, x255 for the classical
ops), and every params entry with its Python kind (tensor / ndarray / int / nested dict).  Each call is rebuilt with
the same layout and kinds and sent into ``reconfigisp_amd.isp_kernels.<module>.<Class>().run``; the result must have
the layout the wrapper expects back and the oracle's values.  Runs on the CPU seam (binding logic) and on the GPU."""
import json

import numpy as np
import pytest
import torch

import isp_oracle as O
from conftest import assert_close, load_golden

_MODULE = {'Grayworld': ('whitebalance', 'WhiteBalance'), 'WbManual': ('whitebalance', 'WhiteBalance'),
           'ConditionalWbManual': ('whitebalance', 'WhiteBalance'), 'OriginWbWhiteworld': ('whitebalance', 'WhiteBalance'),
           'Gamma': ('gamma', 'Gamma'), 'ConditionalGamma': ('gamma', 'Gamma'),
           'DemosaicNearest': ('demosaic', 'Demosaic'), 'DemosaicNet': ('demosaic', 'Demosaic'),
           'OriginDemosBilinear': ('demosaic', 'Demosaic'), 'OriginDemosLaplacian': ('demosaic', 'Demosaic'),
           'OriginToneReinhard': ('globaltonemapping', 'GlobalToneMapping'),
           'OriginToneCrysis': ('globaltonemapping', 'GlobalToneMapping'),
           'OriginToneFilmic': ('globaltonemapping', 'GlobalToneMapping'),
           'OriginNoiseBilateral': ('spatialnoisereduction', 'SpatialNoiseReduction'),
           'OriginNoiseMedian': ('spatialnoisereduction', 'SpatialNoiseReduction'),
           'OriginNoiseFastnlm': ('spatialnoisereduction', 'SpatialNoiseReduction')}


@pytest.fixture(params=['cpu-oracle-seam', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request, monkeypatch):
    import reconfigisp_amd.functional as F
    if request.param == 'hip':
        return torch.device('cuda')
    from oracle_backend import OracleImpl
    monkeypatch.setattr(F, '_IMPL', OracleImpl)
    return torch.device('cpu')


def _rebuild(g, k, dev):
    """(site, option, img with the recorded shape AND strides, params with the recorded kinds)"""
    pre = 'call%02d_' % k
    shape, strides = tuple(int(v) for v in g[pre + 'img_shape']), tuple(int(v) for v in g[pre + 'img_strides'])
    storage = torch.from_numpy(np.ascontiguousarray(g[pre + 'img_storage'])).to(dev)
    img = torch.as_strided(storage.reshape(-1), shape, strides)
    assert tuple(img.stride()) == strides
    params = {}
    for key, m in json.loads(str(g[pre + 'params_meta'])).items():
        if m['kind'] == 'dict':
            params[key] = m['value']
        elif m['kind'] == 'tensor':
            params[key] = torch.from_numpy(g[pre + 'param_' + key]).to(dev)      # the wrappers pass device tensors
            assert str(params[key].dtype) == m['dtype']
        elif m['kind'] == 'ndarray':
            params[key] = np.array(g[pre + 'param_' + key])                      # ... or host numpy arrays (.cpu().numpy())
        else:
            params[key] = m['value']
    return str(g[pre + 'site']), str(g[pre + 'option']), img, params


def _expected(option, img, params):
    """the oracle on the same arguments, in the layout the wrapper expects back"""
    x = img.detach().cpu()
    if option == 'nearestneighbor':                       # NCHW in, NCHW out (tools_origin.py:276-284: 'no need to permute')
        return O.demosaic_nearest(x)
    x = x.permute(0, 3, 1, 2)                             # every other site hands over an NHWC view
    cpu = lambda v: v.detach().cpu() if isinstance(v, torch.Tensor) else v
    if option == 'grayworld':
        y = O.grayworld(x)
    elif option == 'manual' and 'gamma' in params:
        y = O.gamma_manual(x, cpu(params['gamma']))
    elif option == 'manual':
        y = O.wb_manual(x, cpu(params['gain']) / 5.0)
    elif option in ('bilinear', 'laplacian'):
        y = O.origin_demosaic(x, option)
    elif option in ('reinhard', 'crysisengine', 'filmic'):
        y = O.origin_tonemap(x, option, params)
    elif option == 'whiteworld':
        y = O.origin_whiteworld(x, params['white_point_ratio'])
    else:
        y = O.origin_denoise(x, option, {k: cpu(v) for k, v in params.items()})
    return y.permute(0, 2, 3, 1)


def test_recorded_reference_calls_replayed_into_the_plugin_modules(dev):
    import importlib
    g = load_golden('plugin_calls')
    n = int(g['n_calls'])
    assert n == 16 and {str(g['call%02d_site' % k]) for k in range(n)} == set(_MODULE)
    for k in range(n):
        site, option, img, params = _rebuild(g, k, dev)
        mod, cls = _MODULE[site]
        kernel = getattr(importlib.import_module('reconfigisp_amd.isp_kernels.' + mod), cls)()
        if option == 'demosaicnet':                        # the one site that cannot be served (weights are private)
            with pytest.raises(NotImplementedError, match='private'):
                kernel.run(img, option, params)
            continue
        out = kernel.run(img, option, params)
        ref = _expected(option, img, params)
        assert tuple(out.shape) == tuple(ref.shape), '%s: output layout %s, wrapper expects %s' % (site, out.shape, ref.shape)
        # the wrapper then does output.permute(0, 3, 1, 2) (or nothing, for the NCHW demosaics): must be the wrapper's
        # own output shape
        back = out if option == 'nearestneighbor' else out.permute(0, 3, 1, 2)
        assert tuple(back.shape) == tuple(int(v) for v in g['call%02d_wrapper_out_shape' % k]), site
        if option in ('median',):
            assert torch.equal(out.cpu(), ref), site
        elif option in ('bilinear', 'laplacian', 'reinhard', 'crysisengine', 'filmic', 'whiteworld', 'bilateral', 'fastnlm'):
            d = (out.cpu() - ref).abs()                    # 8-bit codes: <= 1 code, <= 0.2 % differ
            assert d.max().item() <= 1 and (d > 0).float().mean().item() <= 2e-3, '%s: %g' % (site, d.max().item())
        else:
            assert_close(out, ref, what=site)
