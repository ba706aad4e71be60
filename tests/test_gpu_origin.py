"""GPU: the classical "Origin" stencil / tone kernels vs the CPU oracle's OPSPEC (build-defined, parity
unpinned against the absent plugin), through the plugin-shaped modules and OriginUniversal.

Outputs are 8-bit codes: a last-ulp difference in an exp() or a division can move a value across a
rounding boundary, so the bar is: no code differs by more than 1, and at most 0.2 % of the codes differ
at all (index maps / medians are exact)."""
import numpy as np
import pytest
import torch

import isp_oracle as O

pytestmark = pytest.mark.gpu


def rnd(*shape, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.random(shape).astype(np.float32))


def codes_close(got, ref, what, exact=False):
    got, ref = got.detach().cpu(), ref.detach().cpu()
    assert got.shape == ref.shape, what
    assert torch.equal(got, got.round()) and got.min() >= 0 and got.max() <= 255, what + ': not 8-bit codes'
    d = (got - ref).abs()
    frac = (d > 0).float().mean().item()
    assert d.max().item() <= (0 if exact else 1), '%s: max code difference %g' % (what, d.max().item())
    assert frac <= (0 if exact else 2e-3), '%s: %.4f%% of the codes differ' % (what, 100 * frac)


@pytest.mark.parametrize('hw', [(8, 8), (34, 50), (256, 256)])
@pytest.mark.parametrize('option', ['bilinear', 'laplacian'])
def test_origin_demosaic(option, hw):
    import reconfigisp_amd.functional as F
    x = torch.floor(rnd(2, 1, *hw, seed=1) * 1023) / 1023 * 255          # 10-bit codes scaled to 0..255
    codes_close(F.origin_demosaic(x.cuda(), option), O.origin_demosaic(x, option), option)
    flat = torch.full((1, 1, 8, 8), 77.0)
    assert F.origin_demosaic(flat.cuda(), option).unique().tolist() == [77.0]   # constant in -> constant out


@pytest.mark.parametrize('hw', [(16, 16), (33 * 2, 45 * 2)])
def test_origin_tonemaps_and_whiteworld(hw):
    import reconfigisp_amd.functional as F
    x = rnd(3, 3, *hw, seed=2) * 255
    p = {'white_point': np.array([0.5, 0.9, 0.1], np.float32), 'middle_grey': np.array([0.5, 0.2, 0.8], np.float32),
         'lum_adapted': np.array([0.5, 0.05, 0.95], np.float32), 'exposure_bias': np.array([5.5, 1.0, 10.0], np.float32)}
    for option in ('reinhard', 'crysisengine', 'filmic'):
        codes_close(F.origin_tonemap(x.cuda(), option, p), O.origin_tonemap(x, option, p), option)
    ratio = np.array([0.5, 0.0, 1.0], np.float32)
    codes_close(F.origin_whiteworld(x.cuda(), ratio), O.origin_whiteworld(x, ratio), 'whiteworld')


@pytest.mark.parametrize('hw', [(16, 16), (40, 70), (24, 132), (70, 64)])
def test_origin_denoisers(hw):
    """W % 4 == 0 shapes take the 64 x 16 / 4-pixels-per-thread kernels (132: a last tile with 4 valid columns, whose
    windows reflect inside the tile), the others the general one-pixel-per-thread kernels."""
    import reconfigisp_amd.functional as F
    x = rnd(2, 3, *hw, seed=3) * 255
    for size in (3, 5, 7, 9) + ((11, 13, 15, 17) if hw[0] < 40 else ()):
        codes_close(F.origin_denoise(x.cuda(), 'median', {'size': size}), O.origin_denoise(x, 'median', {'size': size}),
                    'median %d' % size, exact=True)
    bp = {'window_length': torch.tensor([3, 5]), 'sigma_color': torch.tensor([50.5, 12.0]),
          'sigma_space': torch.tensor([50.5, 1.5])}
    codes_close(F.origin_denoise(x.cuda(), 'bilateral', bp), O.origin_denoise(x, 'bilateral', bp), 'bilateral')
    npar = {'block_size': torch.tensor([3, 3]), 'search_block': torch.tensor([3, 5]),
            'decay_factor': torch.tensor([50.5, 8.0])}
    codes_close(F.origin_denoise(x.cuda(), 'fastnlm', npar), O.origin_denoise(x, 'fastnlm', npar), 'fastnlm')
    with pytest.raises(ValueError, match='odd size'):
        F.origin_denoise(x.cuda(), 'median', {'size': 4})


def test_origin_universal_test_yaml_pipeline():
    """The shipped test YAMLs select OriginUniversal (options/test/SID_test.yml:28:
    Bayer_01_Demosaic_03_sRGB_01_13_11); here with every classical op in one pipeline, through the
    reference-shaped wrappers (x255 / detach / plugin.run / /255), vs the oracle's restatement."""
    from reconfigisp_amd.codes.models.modules.origin_universal import OriginUniversal
    arch = 'Bayer_02_Demosaic_03_sRGB_07_06_11_01_02_08_09_04_03_14'
    net = OriginUniversal(module_path=None, architecture=arch).cuda().eval()
    bay, _ = O.synthetic_raw(2, 32, 48, seed=5)
    with torch.no_grad():
        y = net(bay.cuda())
    names = O.parse_architecture(arch)
    assert names == ['skip', 'laplacian', 'bilateral', 'whiteworld', 'wbmanual', 'gamma', 'reinhard', 'median',
                     'fastnlm', 'filmic', 'crysisengine', 'gtmmanual']
    sig = lambda k: torch.sigmoid(torch.tensor(O.PARAM_INIT[k])).repeat(2, 1)
    x = bay
    for k, got in zip(names, net.intermediate_results):
        p = sig(k) if O.PARAM_INIT[k] else None
        if k == 'skip':
            ref = x
        elif k == 'laplacian':
            ref = O.origin_demosaic(x * 255, 'laplacian') / 255
        elif k == 'bilateral':   # window: .int() before *7 -> always 3 (tools_origin.py:698)
            ref = O.origin_denoise(x * 255, 'bilateral', {'window_length': (p[:, 0].int() * 7) * 2 + 3,
                                                           'sigma_color': p[:, 1] * 99 + 1, 'sigma_space': p[:, 2] * 99 + 1}) / 255
        elif k == 'whiteworld':
            ref = O.origin_whiteworld(x * 255, p[:, 0].numpy()) / 255
        elif k == 'reinhard':
            ref = O.origin_tonemap(x * 255, 'reinhard', {'white_point': p[:, 0].numpy(), 'middle_grey': p[:, 1].numpy()}) / 255
        elif k == 'median':
            ref = O.origin_denoise(x * 255, 'median', {'size': 2 * int(p[0, 0].item() * 7) + 3}) / 255
        elif k == 'fastnlm':
            ref = O.origin_denoise(x * 255, 'fastnlm', {'block_size': (p[:, 0].int() * 7) * 2 + 3,
                                                         'search_block': (p[:, 1].int() * 7) * 2 + 3,
                                                         'decay_factor': p[:, 2] * 99 + 1}) / 255
        elif k == 'filmic':
            ref = O.origin_tonemap(x * 255, 'filmic', {'white_point': p[:, 0].numpy(),
                                                       'exposure_bias': p[:, 1].numpy() * 9. + 1.}) / 255
        elif k == 'crysisengine':
            ref = O.origin_tonemap(x * 255, 'crysisengine', {'lum_adapted': p[:, 0].numpy()}) / 255
        else:
            ref = O.apply_op(k, x, p)
        d = (got.cpu() - ref).abs()
        assert d.max().item() <= 1.01 / 255 and (d > 1e-5).float().mean().item() < 5e-3, 'stage %s: %g' % (k, d.max().item())
        x = got.cpu()            # continue from the GPU result so single-code flips do not cascade


@pytest.mark.parametrize('arch,cls', [('Demosaic_01_sRGB_07_11_01_14', 'OriginUniversal'),
                                      ('Bayer_02_Demosaic_01_sRGB_11_07_01_13', 'OriginUniversal'),
                                      ('Demosaic_01_sRGB_07', 'OriginUniversal')])
@pytest.mark.parametrize('seed', range(6))
def test_fused_stencil_segment_equals_unfused(arch, cls, seed):
    """risp_bilateral_chain_fwd ([demosaic ->] bilateral -> element-wise tail, one launch) must reproduce
    the stage-by-stage path (reference-shaped wrappers: x255, plugin.run, /255) bit for bit."""
    from reconfigisp_amd.codes.models import networks
    opt = {'network_G': {'which_model_G': cls, 'architecture': arch, 'module_path': None}}
    net = networks.define_G(opt).cuda().eval()
    torch.manual_seed(seed)
    with torch.no_grad():
        for p in net.all_params:
            if p.numel():
                p.add_(torch.randn_like(p) * 0.3)
    bay, _ = O.synthetic_raw(3, 48, 80, seed=6 + seed)
    x = bay.cuda()
    with torch.no_grad():
        y_fused = net(x).clone()
        fused = [m.clone() for m in net.intermediate_results]
    pars = net._build_stage_params(3)
    cur, unfused = x, []
    with torch.no_grad():
        for op, par in zip(net.all_modules, pars):
            cur = op(cur, par)
            unfused.append(cur)
    assert len(fused) == len(unfused)
    for k, (a, b) in enumerate(zip(fused, unfused)):
        assert torch.equal(a, b), 'stage %d (%s): max diff %g' % (k, net.step_names[k], (a - b).abs().max().item())
    assert torch.equal(y_fused, unfused[-1])


import os
_FUZZ = int(os.environ.get('RISP_TEST_SEEDS', '8'))               # soak runs: RISP_TEST_SEEDS=64


@pytest.mark.parametrize('seed', range(_FUZZ))
def test_random_classical_kernels(seed):
    """Random sizes, value ranges (negative and over-range samples included) and plugin parameters for every classical
    kernel against the oracle's OPSPEC."""
    import reconfigisp_amd.functional as F
    rng = np.random.default_rng(700 + seed)
    n, h, w = int(rng.integers(1, 4)), 2 * int(rng.integers(4, 24)), 4 * int(rng.integers(2, 14))
    lo, hi = (-40.0, 300.0) if rng.random() < 0.4 else (0.0, 255.0)
    x = torch.from_numpy(rng.uniform(lo, hi, size=(n, 3, h, w)).astype(np.float32))
    vec = lambda a, b: rng.uniform(a, b, size=n).astype(np.float32)
    p = {'white_point': vec(0, 1), 'middle_grey': vec(0, 1), 'lum_adapted': vec(0, 1), 'exposure_bias': vec(1, 10)}
    for option in ('reinhard', 'crysisengine', 'filmic'):
        codes_close(F.origin_tonemap(x.cuda(), option, p), O.origin_tonemap(x, option, p), '%s seed %d' % (option, seed))
    ratio = vec(0, 1)
    codes_close(F.origin_whiteworld(x.cuda(), ratio), O.origin_whiteworld(x, ratio), 'whiteworld seed %d' % seed)
    size = 2 * int(rng.integers(1, 5)) + 1
    if size // 2 < min(h, w):
        codes_close(F.origin_denoise(x.cuda(), 'median', {'size': size}), O.origin_denoise(x, 'median', {'size': size}),
                    'median %d seed %d' % (size, seed), exact=True)
    bp = {'window_length': torch.from_numpy(rng.choice([3, 5, 7], size=n)), 'sigma_color': torch.from_numpy(vec(1, 100)),
          'sigma_space': torch.from_numpy(vec(1, 100))}
    codes_close(F.origin_denoise(x.cuda(), 'bilateral', bp), O.origin_denoise(x, 'bilateral', bp), 'bilateral seed %d' % seed)
    npar = {'block_size': torch.from_numpy(rng.choice([3, 5], size=n)), 'search_block': torch.from_numpy(rng.choice([3, 5], size=n)),
            'decay_factor': torch.from_numpy(vec(1, 100))}
    codes_close(F.origin_denoise(x.cuda(), 'fastnlm', npar), O.origin_denoise(x, 'fastnlm', npar), 'fastnlm seed %d' % seed)
    bay = torch.floor(torch.from_numpy(rng.random((n, 1, h, w)).astype(np.float32)) * 1023) / 1023 * 255
    for option in ('bilinear', 'laplacian'):
        codes_close(F.origin_demosaic(bay.cuda(), option), O.origin_demosaic(bay, option), '%s seed %d' % (option, seed))


@pytest.mark.parametrize('hw', [(16, 16), (66, 92)])
def test_classical_kernels_before_quantisation_at_float_tolerance(hw):
    """The 8-bit criterion above (<= 1 code, <= 0.2 % differ) cannot see a systematic sub-code bias.  The kernels'
    diagnostic form (out_div < 0: no clip-and-round) exposes the float in FRONT of the rounding; it is compared
    with the oracle's unquantised value at the 1e-4 bar of every other operator."""
    import reconfigisp_amd.functional as F
    from conftest import assert_close
    RAW = (1.0, -1.0)
    n = 3
    x = rnd(n, 3, *hw, seed=11) * 255
    bay = torch.floor(rnd(n, 1, *hw, seed=12) * 1023) / 1023 * 255
    p = {'white_point': np.array([0.5, 0.9, 0.1], np.float32), 'middle_grey': np.array([0.5, 0.2, 0.8], np.float32),
         'lum_adapted': np.array([0.5, 0.05, 0.95], np.float32), 'exposure_bias': np.array([5.5, 1.0, 10.0], np.float32)}
    ratio = np.array([0.5, 0.0, 1.0], np.float32)
    bp = {'window_length': torch.tensor([3, 5, 7]), 'sigma_color': torch.tensor([50.5, 12.0, 90.0]),
          'sigma_space': torch.tensor([50.5, 1.5, 3.0])}
    npar = {'block_size': torch.tensor([3, 3, 5]), 'search_block': torch.tensor([3, 5, 3]),
            'decay_factor': torch.tensor([50.5, 8.0, 20.0])}
    with O.unquantized():
        for option in ('bilinear', 'laplacian'):
            assert_close(F.origin_demosaic(bay.cuda(), option, RAW), O.origin_demosaic(bay, option), what=option)
        for option in ('reinhard', 'crysisengine', 'filmic'):
            assert_close(F.origin_tonemap(x.cuda(), option, p, RAW), O.origin_tonemap(x, option, p), what=option)
        assert_close(F.origin_whiteworld(x.cuda(), ratio, RAW), O.origin_whiteworld(x, ratio), what='whiteworld')
        assert_close(F.origin_denoise(x.cuda(), 'bilateral', bp, RAW), O.origin_denoise(x, 'bilateral', bp), what='bilateral')
        assert_close(F.origin_denoise(x.cuda(), 'fastnlm', npar, RAW), O.origin_denoise(x, 'fastnlm', npar), what='fastnlm')
    # and the quantised form is exactly the rounding of the diagnostic form
    for fn in (lambda s: F.origin_denoise(x.cuda(), 'bilateral', bp, s), lambda s: F.origin_tonemap(x.cuda(), 'filmic', p, s)):
        assert torch.equal(fn((1.0, 1.0)), torch.floor(fn(RAW).clamp(0, 255) + 0.5))


def test_window_17_of_a_saturated_parameter():
    """(p.int() * 7) * 2 + 3 and 2 * int(p * 7) + 3 reach 17 when a sigmoid saturates to exactly 1.0
    (tools_origin.py:698, 746, 787): the stencils take radius 8."""
    import reconfigisp_amd.functional as F
    x = rnd(1, 3, 40, 48, seed=13) * 255
    codes_close(F.origin_denoise(x.cuda(), 'median', {'size': 17}), O.origin_denoise(x, 'median', {'size': 17}),
                'median 17', exact=True)
    bp = {'window_length': torch.tensor([17]), 'sigma_color': torch.tensor([30.0]), 'sigma_space': torch.tensor([6.0])}
    codes_close(F.origin_denoise(x.cuda(), 'bilateral', bp), O.origin_denoise(x, 'bilateral', bp), 'bilateral 17')
    npar = {'block_size': torch.tensor([3]), 'search_block': torch.tensor([17]), 'decay_factor': torch.tensor([20.0])}
    codes_close(F.origin_denoise(x.cuda(), 'fastnlm', npar), O.origin_denoise(x, 'fastnlm', npar), 'fastnlm search 17')
    with pytest.raises(ValueError, match='odd size'):
        F.origin_denoise(x.cuda(), 'median', {'size': 19})


def test_bilateral_on_odd_height_takes_the_standalone_kernel():
    """The fused stencil segment needs even H and W % 4 == 0; other shapes must fall back to risp_origin_bilateral
    instead of raising (pipeline_fusion.py dispatch predicate)."""
    from reconfigisp_amd.codes.models import networks
    net = networks.define_G({'network_G': {'which_model_G': 'OriginUniversal', 'architecture': 'sRGB_07_11',
                                           'module_path': None}}).cuda().eval()
    for hw in ((15, 20), (6, 8)):
        x = rnd(2, 3, *hw, seed=14)
        with torch.no_grad():
            y = net(x.cuda())
        ref, _ = O.fixed_pipeline(x, ['bilateral', 'wbmanual'], [torch.tensor(O.PARAM_INIT[k]) for k in ('bilateral', 'wbmanual')],
                                  [None, None], origin=True)
        d = (y.cpu() - ref).abs()
        assert d.max().item() <= 1.01 * 5 * 0.2 / 255 + 1e-6 and (d > 1e-5).float().mean().item() < 5e-3


@pytest.mark.parametrize('hw', [(16, 16), (40, 132), (22, 64)])
@pytest.mark.parametrize('form', ['codes', 'raw'])
def test_tile_forms_give_identical_bits(hw, form):
    """Every classical stencil has a general kernel (32 x 8 tiles, one pixel per thread) and, for W % 4 == 0 and a
    16-byte aligned output, a 64 x 16 form with 4 pixels per thread (packed bytes + v_sad_u8 for the median).  Same
    input, the output pointer moved by one float to force the general kernel: the results must be the same bits -
    quantised codes and the float in front of the quantisation alike."""
    import ctypes as C
    from reconfigisp_amd import lib as L
    h, w = hw
    n = 2
    dev = torch.device('cuda')
    x = (rnd(n, 3, h, w, seed=91) * 255).to(dev)
    bay = (rnd(n, 1, h, w, seed=92) * 255).to(dev)
    so = 1.0 if form == 'codes' else -1.0
    p = lambda t: C.c_void_p(t.data_ptr())

    def both(call, shape):
        outs = []
        for shift in (0, 1):
            buf = torch.zeros(int(np.prod(shape)) + 4, device=dev)
            y = buf[shift:shift + int(np.prod(shape))].view(shape)
            assert (y.data_ptr() % 16 == 0) == (shift == 0)
            call(y)
            outs.append(y.clone())
        assert torch.equal(outs[0], outs[1]), 'max difference %g' % (outs[0] - outs[1]).abs().max().item()

    win = torch.tensor([3, 5], dtype=torch.int32, device=dev)
    sc, ss = torch.tensor([50.5, 12.0], device=dev), torch.tensor([50.5, 1.5], device=dev)
    both(lambda y: L.call('risp_origin_bilateral', p(x), p(y), p(win), p(sc), p(ss), 5, n, h, w, 1.0, so, None), x.shape)
    win3 = torch.tensor([3, 3], dtype=torch.int32, device=dev)
    both(lambda y: L.call('risp_origin_bilateral', p(x), p(y), p(win3), p(sc), p(ss), 3, n, h, w, 1.0, so, None), x.shape)
    blk, srch = torch.tensor([3, 3], dtype=torch.int32, device=dev), torch.tensor([3, 3], dtype=torch.int32, device=dev)
    dec = torch.tensor([50.5, 8.0], device=dev)
    both(lambda y: L.call('risp_origin_fastnlm', p(x), p(y), p(blk), p(srch), p(dec), 3, 3, n, h, w, 1.0, so, None), x.shape)
    srch5 = torch.tensor([3, 5], dtype=torch.int32, device=dev)
    both(lambda y: L.call('risp_origin_fastnlm', p(x), p(y), p(blk), p(srch5), p(dec), 3, 5, n, h, w, 1.0, so, None), x.shape)
    for lap in (0, 1):
        both(lambda y: L.call('risp_origin_demosaic', p(bay), p(y), lap, n, h, w, 1.0, so, None), (n, 3, h, w))
    if form == 'codes':
        for size in (3, 5, 7, 9, 13, 17):
            if size // 2 < min(h, w):
                both(lambda y: L.call('risp_origin_median', p(x), p(y), size, n, h, w, 1.0, 1.0, None), x.shape)


def test_unaligned_input_views_take_the_general_kernels():
    """The 4-pixel forms read the INPUT in 16-byte vectors too: a contiguous view whose storage offset is not a multiple
    of 4 floats (aligned output) must give the bits of the aligned call - every classical stencil and the histogram."""
    import ctypes as C
    from reconfigisp_amd import lib as L
    n, h, w = 2, 40, 72
    dev = torch.device('cuda')
    p = lambda t: C.c_void_p(t.data_ptr())

    def shifted(t):
        buf = torch.zeros(t.numel() + 4, device=dev)
        v = buf[1:1 + t.numel()].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4 and v.is_contiguous()
        return v

    x = (rnd(n, 3, h, w, seed=93) * 255).to(dev)
    bay = (rnd(n, 1, h, w, seed=94) * 255).to(dev)
    win = torch.tensor([3, 3], dtype=torch.int32, device=dev)
    sc, ss = torch.tensor([50.5, 12.0], device=dev), torch.tensor([50.5, 1.5], device=dev)
    dec = torch.tensor([50.5, 8.0], device=dev)
    calls = [
        (x, lambda s, y: L.call('risp_origin_bilateral', p(s), p(y), p(win), p(sc), p(ss), 3, n, h, w, 1.0, 1.0, None), x.shape),
        (x, lambda s, y: L.call('risp_origin_fastnlm', p(s), p(y), p(win), p(win), p(dec), 3, 3, n, h, w, 1.0, 1.0, None), x.shape),
        (x, lambda s, y: L.call('risp_origin_median', p(s), p(y), 3, n, h, w, 1.0, 1.0, None), x.shape),
        (x, lambda s, y: L.call('risp_origin_median', p(s), p(y), 9, n, h, w, 1.0, 1.0, None), x.shape),
        (bay, lambda s, y: L.call('risp_origin_demosaic', p(s), p(y), 1, n, h, w, 1.0, 1.0, None), (n, 3, h, w)),
        (x / 255, lambda s, y: L.call('risp_histc', p(s), p(y), n * 3, h * w, 32, None), (n * 3, 32)),
    ]
    for src, call, shape in calls:
        src = src.contiguous()
        a, b = torch.zeros(shape, device=dev), torch.zeros(shape, device=dev)
        call(src, a)
        call(shifted(src), b)
        assert torch.equal(a, b)
