"""GPU parity: the MFMA convolution kernels (through the C ABI and the reference-shaped
modules) vs golden vectors of the imported reference and vs the CPU oracle."""
import numpy as np
import pytest
import torch

import isp_oracle as O
from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu

T = lambda a: torch.from_numpy(np.asarray(a))


def _modules():
    from reconfigisp_amd.codes.models.modules import tools_proxy as TP
    return TP


def _build(kind, seed, P=0):
    TP = _modules()
    cls = {'srcnn_res': TP.ProxyNet, 'srcnn_demosaic': TP.ProxyDemosaicNet,
           'path14l_bayer': TP.PathRestore14lBayer, 'path14l_bgr': TP.PathRestore14lBgr}[kind]
    m = cls(P, None)
    m.load_state_dict(O.make_weights(kind, seed, P))
    return m.cuda()


def _golden_case(tag, kind, P=0, rtol=1e-4):
    g = load_golden('cnn_' + tag)
    m = _build(kind, int(g['seed']), P)
    x = T(g['x']).cuda().requires_grad_(True)
    pv = T(g['pv']).cuda().requires_grad_(True) if P else None
    y = m(x, pv)
    assert_close(y, g['y'], rtol=rtol, what=tag + ' y')
    if 'gx' in g:
        grads = torch.autograd.grad(y, (x, pv) if P else (x,), T(g['gy']).cuda())
        assert_close(grads[0], g['gx'], rtol=rtol, what=tag + ' gx')
        if P:
            assert_close(grads[1], g['gpv'], rtol=rtol, what=tag + ' gpv')


def test_srcnn_res_vs_reference_golden():
    _golden_case('srcnn_res_p2', 'srcnn_res', 2)
    _golden_case('srcnn_res_p5', 'srcnn_res', 5)
    _golden_case('srcnn_res_p3_36x70', 'srcnn_res', 3)


def test_srcnn_demosaic_vs_reference_golden():
    _golden_case('srcnn_demosaic', 'srcnn_demosaic')


@pytest.mark.parametrize('arith', ['f16x2', 'f32'])
def test_path14l_vs_reference_golden(arith, monkeypatch):
    """forward + input gradient of both Path-Restore nets against the reference goldens, with the 3x3 layers in split precision
    on the f16 matrix pipe (the default) and on the fp32 F(4,3) kernels (RISP_CONV_ARITH=f32)"""
    from reconfigisp_amd import convnets as CN
    monkeypatch.setattr(CN, 'CONV_ARITH', arith)
    _golden_case('path14l_bayer', 'path14l_bayer')
    _golden_case('path14l_bayer_40x72', 'path14l_bayer')
    _golden_case('path14l_bgr', 'path14l_bgr')


def kink_free_weights(kind, seed, P):
    """Weights whose hidden ReLUs are decided by a per-channel bias of +-1 (even channels live, odd
    channels dead) with the convolution contributing only a few percent on top.  A dense input-gradient
    can only be compared between two fp32 implementations when no pre-activation sits on the ReLU kink
    (a 1-ulp difference would flip a mask bit); random weights at multi-tile sizes always have some.
    The ReLU-mask / residual logic is still exercised (half the channels masked), the convolution
    arithmetic itself is pinned by test_gpu_conv_modes.py and the golden cases."""
    w = O.make_weights(kind, seed, P, gain=0.02)
    keys = [k for k in w if k.endswith('.bias')]
    last = keys[-1]
    for k in keys:
        if k != last:
            sign = torch.ones_like(w[k])
            sign[1::2] = -1.0
            w[k] = sign
    return w


@pytest.mark.parametrize('kind,P,cin,hw', [('srcnn_res', 1, 3, (64, 96)), ('srcnn_demosaic', 0, 1, (64, 96)),
                                           ('path14l_bayer', 0, 1, (96, 64)), ('path14l_bgr', 0, 3, (48, 80))])
def test_cnn_vs_oracle_multi_tile(kind, P, cin, hw):
    """Sizes spanning several 16x32 output tiles: forward with random weights, forward + backward with
    kink-free weights, vs the CPU oracle."""
    g = np.random.Generator(np.random.PCG64(5))
    x = torch.from_numpy(g.random((2, cin) + hw).astype(np.float32))
    pv = torch.from_numpy(g.random((2, P)).astype(np.float32)) if P else None
    TP = _modules()
    cls = {'srcnn_res': TP.ProxyNet, 'srcnn_demosaic': TP.ProxyDemosaicNet,
           'path14l_bayer': TP.PathRestore14lBayer, 'path14l_bgr': TP.PathRestore14lBgr}[kind]

    def oracle(a, b, w):
        return {'srcnn_res': lambda: O.srcnn_res(a, b, w), 'srcnn_demosaic': lambda: O.srcnn_demosaic(a, w),
                'path14l_bayer': lambda: O.path14l_bayer(a, w), 'path14l_bgr': lambda: O.path14l_bgr(a, w)}[kind]()

    w = O.make_weights(kind, 321, P)
    m = cls(P, None)
    m.load_state_dict(w)
    m = m.cuda()
    with torch.no_grad():
        assert_close(m(x.cuda(), pv.cuda() if P else None), oracle(x, pv, w), what=kind + ' y (random weights)')

    w = kink_free_weights(kind, 654, P)
    m.load_state_dict(w)
    xc = x.clone().requires_grad_(True)
    pc = pv.clone().requires_grad_(True) if P else None
    yc = oracle(xc, pc, w)
    gy = torch.from_numpy(g.standard_normal(tuple(yc.shape)).astype(np.float32))
    gc = torch.autograd.grad(yc, (xc, pc) if P else (xc,), gy)
    xg = x.cuda().requires_grad_(True)
    pg = pv.cuda().requires_grad_(True) if P else None
    yg = m(xg, pg)
    gg = torch.autograd.grad(yg, (xg, pg) if P else (xg,), gy.cuda())
    assert_close(yg, yc, what=kind + ' y')
    assert_close(gg[0], gc[0], what=kind + ' gx')
    if P:
        assert_close(gg[1], gc[1], what=kind + ' gpv')


def test_repack_on_weight_update():
    m = _build('srcnn_demosaic', 1)
    x = torch.rand(1, 1, 16, 16).cuda()
    y0 = m(x, None).clone()
    with torch.no_grad():
        m.srcnn[0].weight.mul_(0.5)
    y1 = m(x, None)
    assert (y0 - y1).abs().max() > 1e-4
    m.load_state_dict(O.make_weights('srcnn_demosaic', 1))
    assert_close(m.cuda()(x, None), y0)


@pytest.mark.parametrize('kind,P,cin', [('srcnn_res', 2, 3), ('srcnn_demosaic', 0, 1), ('path14l_bayer', 0, 1), ('path14l_bgr', 0, 3)])
def test_inference_result_independent_of_batch(kind, P, cin):
    """test_split.py batches tiles (test_split.py:100-121): without autograd an image's output must be the same bits
    whether it travels alone or in a batch - no grid-dependent kernel form (channel-group split of the direct kernel,
    F(2,3) / F(4,3) switch) may be taken on the inference path."""
    m = _build(kind, 5, P)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(6, cin, 64, 96, generator=g).cuda()
    pv = torch.rand(6, P, generator=g).cuda() if P else None
    with torch.no_grad():
        full = m(x, pv)
        for k in (0, 5):
            alone = m(x[k:k + 1].contiguous(), pv[k:k + 1].contiguous() if P else None)
            assert torch.equal(alone, full[k:k + 1]), '%s: image %d differs between batch 1 and batch 6 (max %g)' % (
                kind, k, (alone - full[k:k + 1]).abs().max().item())
