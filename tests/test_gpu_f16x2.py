"""GPU parity of the split-precision convolution ``risp_conv2d_f16x2`` (two f16 halves per fp32 operand, three f16 matrix
products, fp32 accumulation; reconfigisp_amd/csrc/risp_conv_f16x2.hip) through the C ABI: against the float64 convolution -
the yardstick its accuracy claim is made on - next to the fp32 matrix-core kernels on the same data, every epilogue, forward
and backward-data packs, aligned and ragged shapes, inputs of gradient magnitude, zeros, NaNs, run-to-run and batch
independence.  Layers: Path-Restore's 64 -> 64 3x3 blocks (path_14l_bgr_arch.py:6-21, path_14l_bayer_arch.py:59-88)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rnd(*shape, seed, dtype=np.float32):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(shape).astype(dtype)).cuda()


def launch(entry, x, pack, bias, n, h, w, cin, cout, epi=0, add=None, mask=None):
    from reconfigisp_amd import lib as L
    y = torch.full((n, cout, h, w), float('nan'), device='cuda')
    d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=3, load_mode=0, cin_img=0, epilogue=epi | (0 if bias is not None else 16),
                   add_c=cout if add is not None else 0, x=x.data_ptr(), wpack=pack.data_ptr(),
                   bias=bias.data_ptr() if bias is not None else None, cvals=None, add=add.data_ptr() if add is not None else None,
                   mask=mask.data_ptr() if mask is not None else None, y=y.data_ptr())
    L.call(entry, C.byref(d), None)
    torch.cuda.synchronize()
    return y


def err(y, ref):
    """(rms, max) error relative to the reference's largest magnitude"""
    m = ref.abs().max().item() or 1.0
    e = y.double() - ref
    return e.pow(2).mean().sqrt().item() / m, e.abs().max().item() / m


SHAPES = [(1, 8, 64), (2, 16, 64), (3, 13, 68), (2, 40, 100), (1, 7, 4), (2, 33, 128), (5, 9, 196)]


@pytest.mark.parametrize('cin,cout', [(64, 64), (64, 32), (32, 64), (16, 32)])
@pytest.mark.parametrize('nhw', SHAPES)
def test_forward_against_float64_next_to_the_fp32_kernels(cin, cout, nhw):
    from reconfigisp_amd import convnets as CN, lib as L
    n, h, w = nhw
    wt, b = rnd(cout, cin, 3, 3, seed=1) * 0.05, rnd(cout, seed=2) * 0.1
    x = torch.rand(n, cin, h, w, device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    ref = TF.conv2d(x.double(), wt.double(), b.double(), padding=1)
    y = launch('risp_conv2d_f16x2', x, CN.f16x2_weights(wt, False), b, n, h, w, cin, cout)
    p32 = torch.empty(L.load().risp_conv_wpack_floats(cin, cout, 3), device='cuda')
    L.call('risp_conv_pack_weights', C.c_void_p(wt.data_ptr()), cin, cout, 3, 0, C.c_void_p(p32.data_ptr()), None)
    y32 = launch('risp_conv2d', x, p32, b, n, h, w, cin, cout)
    (rms, mx), (rms32, mx32) = err(y, ref), err(y32, ref)
    assert not torch.isnan(y).any()
    # the claim: fp32-level accuracy - no worse than the exact-fp32 matrix-core kernel on the same data (small slack for the
    # tiny shapes, where a handful of outputs decide the maximum), and far inside the 1e-4 bar
    assert rms <= 1.25 * rms32 + 1e-9 and mx <= 2.0 * mx32 + 1e-8, (rms, rms32, mx, mx32)
    assert mx < 5e-6


@pytest.mark.parametrize('nhw', [(2, 16, 64), (3, 13, 68), (2, 21, 36)])
def test_epilogues_and_backward_data_pack(nhw):
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    c = 64
    wt, b = rnd(c, c, 3, 3, seed=5) * 0.05, rnd(c, seed=6) * 0.1
    x, add, mask = rnd(n, c, h, w, seed=7), rnd(n, c, h, w, seed=8), rnd(n, c, h, w, seed=9)
    pf, pb = CN.f16x2_weights(wt, False), CN.f16x2_weights(wt, True)
    lin = TF.conv2d(x.double(), wt.double(), b.double(), padding=1)
    E = CN
    cases = [('plain', 0, None, None, lin),
             ('relu', E.EPI_RELU, None, None, torch.relu(lin)),
             ('add+relu', E.EPI_ADD | E.EPI_RELU, add, None, torch.relu(lin + add.double())),
             ('mask', E.EPI_MASK, None, mask, lin * (mask > 0)),
             ('add+mask', E.EPI_ADD | E.EPI_MASK, add, mask, (lin + add.double()) * (mask > 0))]
    for what, epi, a, m, ref in cases:
        y = launch('risp_conv2d_f16x2', x, pf, b, n, h, w, c, c, epi, a, m)
        assert err(y, ref)[1] < 3e-6, what
    lin_t = TF.conv_transpose2d(x.double(), wt.double(), padding=1)       # == backward-data of the forward layer
    y = launch('risp_conv2d_f16x2', x, pb, None, n, h, w, c, c, E.EPI_ADD | E.EPI_MASK, add, mask)
    assert err(y, (lin_t + add.double()) * (mask > 0))[1] < 3e-6
    y = launch('risp_conv2d_f16x2', x, pb, None, n, h, w, c, c, E.EPI_MASK, None, mask)
    assert err(y, lin_t * (mask > 0))[1] < 3e-6


@pytest.mark.parametrize('scale', [1e-8, 1e-5, 1.0, 3e4])
def test_accuracy_does_not_depend_on_the_magnitude_of_the_input(scale):
    """upstream gradients of magnitude 1e-8 are as exact as activations of magnitude 1: the scale is taken per tile and chunk"""
    from reconfigisp_amd import convnets as CN
    n, h, w, c = 2, 24, 64, 64
    wt = rnd(c, c, 3, 3, seed=11) * 0.05
    x = rnd(n, c, h, w, seed=12) * scale * (rnd(n, c, h, w, seed=13) > 0)
    ref = TF.conv2d(x.double(), wt.double(), None, padding=1)
    y = launch('risp_conv2d_f16x2', x, CN.f16x2_weights(wt, False), None, n, h, w, c, c)
    rms, mx = err(y, ref)
    assert rms < 1.5e-7 and mx < 3e-6, (scale, rms, mx)


def test_mixed_magnitudes_inside_one_tile_and_between_chunks():
    """channels 0-15 of magnitude 1e-6, 16-31 of 1, 32-47 of 1e3, 48-63 zero: the running exponent is lowered between the chunks
    of a tile (accumulators rescaled exactly), and a chunk of zeros is harmless"""
    from reconfigisp_amd import convnets as CN
    n, h, w, c = 1, 16, 128, 64
    wt = rnd(c, c, 3, 3, seed=21) * 0.05
    x = rnd(n, c, h, w, seed=22)
    x[:, :16] *= 1e-6
    x[:, 32:48] *= 1e3
    x[:, 48:] = 0
    ref = TF.conv2d(x.double(), wt.double(), None, padding=1)
    y = launch('risp_conv2d_f16x2', x, CN.f16x2_weights(wt, False), None, n, h, w, c, c)
    assert err(y, ref)[1] < 3e-6
    # the small channels alone (what the big ones would drown): their own convolution at their own magnitude
    x2 = x.clone()
    x2[:, 16:] = 0
    ref2 = TF.conv2d(x2.double(), wt.double(), None, padding=1)
    y2 = launch('risp_conv2d_f16x2', x2, CN.f16x2_weights(wt, False), None, n, h, w, c, c)
    assert err(y2, ref2)[1] < 3e-6


def test_zeros_bias_only_and_nan_locality():
    from reconfigisp_amd import convnets as CN
    n, h, w, c = 1, 16, 64, 64
    wt, b = rnd(c, c, 3, 3, seed=31) * 0.05, rnd(c, seed=32)
    pf = CN.f16x2_weights(wt, False)
    y = launch('risp_conv2d_f16x2', torch.zeros(n, c, h, w, device='cuda'), pf, b, n, h, w, c, c)
    assert torch.equal(y, b.view(1, c, 1, 1).expand(n, c, h, w))
    x = rnd(n, c, h, w, seed=33)
    clean = launch('risp_conv2d_f16x2', x, pf, b, n, h, w, c, c)
    x[0, 5, 9, 40] = float('nan')
    y = launch('risp_conv2d_f16x2', x, pf, b, n, h, w, c, c)
    hit = torch.zeros(h, w, dtype=torch.bool, device='cuda')
    hit[8:11, 39:42] = True
    assert torch.isnan(y[0][:, hit]).all() and not torch.isnan(y[0][:, ~hit]).any()
    # ... and outside the NaN's 3x3 footprint, but inside its 8 x 64 tile, the values are the clean ones to rounding: the tile's
    # scale ignores the NaN
    assert (y[0][:, ~hit] - clean[0][:, ~hit]).abs().max().item() < 1e-5


def test_bit_repeatable_and_independent_of_the_batch():
    from reconfigisp_amd import convnets as CN
    n, h, w, c = 7, 24, 132, 64
    wt, b = rnd(c, c, 3, 3, seed=41) * 0.05, rnd(c, seed=42) * 0.1
    pf = CN.f16x2_weights(wt, False)
    x = rnd(n, c, h, w, seed=43)
    y1 = launch('risp_conv2d_f16x2', x, pf, b, n, h, w, c, c, CN.EPI_RELU)
    y2 = launch('risp_conv2d_f16x2', x, pf, b, n, h, w, c, c, CN.EPI_RELU)
    assert torch.equal(y1, y2)
    for i in (0, 3, 6):                          # an image alone == the same image inside the batch (tiles are per image)
        yi = launch('risp_conv2d_f16x2', x[i:i + 1].contiguous(), pf, b, 1, h, w, c, c, CN.EPI_RELU)
        assert torch.equal(yi[0], y1[i])
    big = rnd(600, 64, 8, 64, seed=44)           # more tiles than persistent workgroups: every workgroup walks several tiles
    yb = launch('risp_conv2d_f16x2', big, pf, b, 600, 8, 64, c, c)
    ref = TF.conv2d(big[::97].double(), wt.double(), b.double(), padding=1)
    assert err(yb[::97], ref)[1] < 3e-6


def test_argument_checks():
    from reconfigisp_amd import convnets as CN, lib as L
    wt = rnd(64, 64, 3, 3, seed=51) * 0.05
    pf = CN.f16x2_weights(wt, False)
    x = rnd(1, 64, 8, 64, seed=52)
    with pytest.raises(RuntimeError, match='cout 32 or 64'):
        launch('risp_conv2d_f16x2', x, pf, None, 1, 8, 64, 64, 48)
    with pytest.raises(RuntimeError, match='cin %% 16|cin % 16'):
        launch('risp_conv2d_f16x2', x, pf, None, 1, 8, 64, 24, 64)
    with pytest.raises(RuntimeError, match='W %% 4|W % 4'):
        launch('risp_conv2d_f16x2', x, pf, None, 1, 8, 62, 64, 64)
    assert L.load().risp_conv_f16x2_wpack_bytes(64, 64, 3) == pf.numel() * 2


@pytest.mark.parametrize('arith', ['f16x2', 'f32'])
def test_dispatch_switch_through_conv(arith, monkeypatch):
    """convnets.conv routes the wide 3x3 layers by CONV_ARITH (env RISP_CONV_ARITH); both routes meet the float64 reference"""
    from reconfigisp_amd import convnets as CN
    monkeypatch.setattr(CN, 'CONV_ARITH', arith)
    calls = []
    real = CN.L.call
    monkeypatch.setattr(CN.L, 'call', lambda name, *a: (calls.append(name), real(name, *a))[1])
    n, h, w, c = 2, 20, 36, 64
    wt, b = rnd(c, c, 3, 3, seed=61) * 0.05, rnd(c, seed=62) * 0.1
    pc = CN.PackedConv(wt, b)
    x, mask = rnd(n, c, h, w, seed=63), rnd(n, c, h, w, seed=64)
    calls.clear()
    y = CN.conv(x, pc, n, h, w, epi=CN.EPI_RELU)
    g = CN.conv(x, pc, n, h, w, transpose=True, epi=CN.EPI_MASK, mask=mask)
    assert calls == (['risp_conv2d_f16x2'] * 2 if arith == 'f16x2' else ['risp_conv2d_wino43'] * 2)
    assert err(y, torch.relu(TF.conv2d(x.double(), wt.double(), b.double(), padding=1)))[1] < 5e-6
    assert err(g, TF.conv_transpose2d(x.double(), wt.double(), padding=1) * (mask > 0))[1] < 5e-6


def launch_k(entry, x, pack, bias, n, h, w, cin, cout, k, epi=0, add=None, mask=None, group=None):
    from reconfigisp_amd import lib as L
    nn = n * (group[0] if group else 1)
    y = torch.full((nn, cout, h, w), float('nan'), device='cuda')
    d = L.ConvDesc(N=nn, H=h, W=w, cin=cin, cout=cout, ksize=k, load_mode=0, cin_img=0, epilogue=epi | (0 if bias is not None else 16),
                   add_c=cout if add is not None else 0, x=x.data_ptr(), wpack=pack.data_ptr(),
                   bias=bias.data_ptr() if bias is not None else None, cvals=None, add=add.data_ptr() if add is not None else None,
                   mask=mask.data_ptr() if mask is not None else None, y=y.data_ptr())
    if group:
        d.group_n, d.group_flags = n, group[1]
        d.wpack_gs = pack.stride(0) * pack.element_size() // 4
        d.bias_gs = bias.stride(0) if bias is not None else 0
    L.call(entry, C.byref(d), None)
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize('cin,cout', [(64, 32), (32, 64), (64, 64), (16, 32)])
@pytest.mark.parametrize('nhw', [(1, 8, 64), (3, 13, 68), (2, 40, 100), (1, 7, 4), (2, 33, 128)])
def test_5x5_forward_and_backward_pack_against_float64(cin, cout, nhw):
    """the 5x5 form (SRCNNRes' 64 -> 32 layer and its 32 -> 64 backward, srcnn_res_arch.py:20): one cout block per tile, a
    64-cout layer as two tiles per pixel tile"""
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    wt, b = rnd(cout, cin, 5, 5, seed=71) * 0.02, rnd(cout, seed=72) * 0.1
    x, mask = rnd(n, cin, h, w, seed=73), rnd(n, cout, h, w, seed=74)
    ref = TF.conv2d(x.double(), wt.double(), b.double(), padding=2)
    y = launch_k('risp_conv2d_f16x2', x, CN.f16x2_weights(wt, False), b, n, h, w, cin, cout, 5, CN.EPI_RELU)
    assert err(y, torch.relu(ref))[1] < 3e-6
    wb = rnd(cin, cout, 5, 5, seed=75) * 0.02            # a forward layer cout_f = cin here: its backward-data maps cin -> cout
    refb = TF.conv_transpose2d(x.double(), wb.double(), padding=2) * (mask > 0)
    yb = launch_k('risp_conv2d_f16x2', x, CN.f16x2_weights(wb, True), None, n, h, w, cin, cout, 5, CN.EPI_MASK, None, mask)
    rms, mx = err(yb, refb)
    assert mx < 3e-6 and rms < 2e-7, (rms, mx)


def test_5x5_accuracy_next_to_the_fp32_winograd_kernel():
    from reconfigisp_amd import convnets as CN
    n, h, w, cin, cout = 2, 64, 128, 64, 32
    wt, b = rnd(cout, cin, 5, 5, seed=81) * 0.02, rnd(cout, seed=82) * 0.1
    x = torch.rand(n, cin, h, w, device='cuda', generator=torch.Generator('cuda').manual_seed(83))
    ref = TF.conv2d(x.double(), wt.double(), b.double(), padding=2)
    y = launch_k('risp_conv2d_f16x2', x, CN.f16x2_weights(wt, False), b, n, h, w, cin, cout, 5)
    y32 = launch_k('risp_conv2d_wino45', x, CN.wino45_weights(wt, False), b, n, h, w, cin, cout, 5)
    (rms, mx), (rms32, mx32) = err(y, ref), err(y32, ref)
    assert rms <= rms32 and mx <= 1.5 * mx32, (rms, rms32, mx, mx32)


@pytest.mark.parametrize('k,cin,cout', [(5, 64, 32), (5, 32, 64), (3, 64, 64)])
def test_grouped_launch_equals_member_launches(k, cin, cout):
    """risp_conv_desc.group_n: G members' images stacked along N, weights / bias by member - the bits of the member launches"""
    from reconfigisp_amd import convnets as CN, lib as L
    G, n, h, w = 3, 2, 21, 68
    wts = [rnd(cout, cin, k, k, seed=90 + g) * (0.02 * (g + 1)) for g in range(G)]
    bs = torch.stack([rnd(cout, seed=95 + g) * 0.1 for g in range(G)])
    packs = torch.stack([CN.f16x2_weights(wt, False) for wt in wts])
    x, mask = rnd(G * n, cin, h, w, seed=99), rnd(G * n, cout, h, w, seed=100)
    yg = launch_k('risp_conv2d_f16x2', x, packs, bs, n, h, w, cin, cout, k, CN.EPI_MASK, None, mask, group=(G, 0))
    for g in range(G):
        s = slice(g * n, (g + 1) * n)
        ym = launch_k('risp_conv2d_f16x2', x[s].contiguous(), packs[g], bs[g], n, h, w, cin, cout, k, CN.EPI_MASK, None, mask[s].contiguous())
        assert torch.equal(yg[s], ym), g
    # one input shared by all members (the slot input as x of a first layer)
    ys = launch_k('risp_conv2d_f16x2', x[:n].contiguous(), packs, bs, n, h, w, cin, cout, k, 0, None, None, group=(G, L.GROUP_SHARED_X))
    for g in range(G):
        ym = launch_k('risp_conv2d_f16x2', x[:n].contiguous(), packs[g], bs[g], n, h, w, cin, cout, k)
        assert torch.equal(ys[g * n:(g + 1) * n], ym), g


@pytest.mark.parametrize('k,cin,cout', [(3, 64, 64), (3, 16, 64), (3, 64, 32), (3, 32, 32), (5, 64, 32), (5, 32, 64), (5, 16, 32)])
@pytest.mark.parametrize('nhw', [(1, 8, 64), (3, 13, 68), (2, 40, 100), (1, 7, 4), (5, 24, 196), (300, 8, 64)])
def test_wave_specialised_kernel_gives_the_bits_of_the_round4_kernel(k, cin, cout, nhw):
    """risp_conv2d_f16x2 = one 8-wave workgroup per CU with producer and consumer waves (risp_conv_f16x2_ws.hip),
    risp_conv2d_f16x2_uniform = the round-4 form in which every wave does both.  Same products in the same order into every
    accumulator: identical bits, every epilogue, one tile to several tiles per persistent workgroup."""
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    wt, b = rnd(cout, cin, k, k, seed=111) * 0.05, rnd(cout, seed=112) * 0.1
    x, add, mask = rnd(n, cin, h, w, seed=113), rnd(n, cout, h, w, seed=114), rnd(n, cout, h, w, seed=115)
    x[:, : cin // 2] *= 1e-3                                     # the running exponent moves between the chunks
    pf = CN.f16x2_weights(wt, False)
    for epi, a, m in ((0, None, None), (CN.EPI_RELU, None, None), (CN.EPI_ADD | CN.EPI_RELU, add, None), (CN.EPI_MASK, None, mask),
                      (CN.EPI_ADD | CN.EPI_MASK, add, mask)):
        y1 = launch_k('risp_conv2d_f16x2', x, pf, b, n, h, w, cin, cout, k, epi, a, m)
        y0 = launch_k('risp_conv2d_f16x2_uniform', x, pf, b, n, h, w, cin, cout, k, epi, a, m)
        assert not torch.isnan(y1).any() and torch.equal(y0, y1), epi
