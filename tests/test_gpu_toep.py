"""GPU parity of ``risp_conv2d_toep`` (reconfigisp_amd/csrc/risp_conv_toep.hip) through the C ABI: the small-cout layers with 5-
and 9-tap rows on the f16 matrix pipe, rows of the matrix instruction = (cout, position inside a block of 8 pixels), reduction
index = a window of 16 input pixels (a Toeplitz band of the filter row as the A operand), split precision as in
``risp_conv2d_f16x2``.  Against the float64 convolution next to the vector-FMA kernel ``risp_conv2d_small`` it replaces, every
epilogue, grouped launches, ragged shapes, gradient-sized inputs, and the dispatch in ``convnets.conv_small``.
Layers: srcnn_res_arch.py:18 (backward-data, 64 -> 3), :22 (5x5 32 -> 3); srcnn_demosaic_arch.py:14-16 (backward-data through
PixelShuffle, 64 -> 4)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rnd(*shape, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(shape).astype(np.float32)).cuda()


def launch(entry, x, pack, bias, n, h, w, cin, cout, k, epi=0, add=None, add_c=0, group=None, out=None):
    """one launch through the C ABI; ``group`` = (G, flags, stacked bias or None): ``pack`` then holds the G members' packs"""
    from reconfigisp_amd import lib as L
    G = group[0] if group else 1
    shape = (G * n, cout // 4, 2 * h, 2 * w) if epi & 8 else (G * n, cout, h, w)
    y = torch.full(shape, float('nan'), device='cuda') if out is None else out
    d = L.ConvDesc(N=G * n, H=h, W=w, cin=cin, cout=cout, ksize=k, load_mode=0, cin_img=0, epilogue=epi | (0 if bias is not None else 16),
                   add_c=add_c, x=x.data_ptr(), wpack=pack.data_ptr(), bias=bias.data_ptr() if bias is not None else None, cvals=None,
                   add=add.data_ptr() if add is not None else None, mask=None, y=y.data_ptr())
    if group:
        d.group_n, d.group_flags = n, group[1]
        d.wpack_gs = pack.stride(0) * pack.element_size() // 4
        d.bias_gs = bias.stride(0) if bias is not None else 0
    L.call(entry, C.byref(d), None)
    torch.cuda.synchronize()
    return y


def err(y, ref):
    m = ref.abs().max().item() or 1.0
    e = y.double() - ref
    return e.pow(2).mean().sqrt().item() / m, e.abs().max().item() / m


SHAPES = [(1, 16, 256), (2, 37, 64), (3, 16, 260), (1, 5, 8), (2, 50, 512), (1, 33, 4), (5, 20, 132)]


@pytest.mark.parametrize('k,cin,cout', [(9, 64, 3), (5, 32, 3), (9, 64, 4), (5, 7, 1), (9, 3, 2), (5, 32, 12), (5, 8, 6)])
@pytest.mark.parametrize('nhw', SHAPES)
def test_forward_against_float64_next_to_the_vector_kernel(k, cin, cout, nhw):
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    wt, b = rnd(cout, cin, k, k, seed=1) * 0.05, rnd(cout, seed=2) * 0.1
    x = rnd(n, cin, h, w, seed=3)
    ref = TF.conv2d(x.double(), wt.double(), b.double(), padding=k // 2)
    y = launch('risp_conv2d_toep', x, CN.toep_weights(wt), b, n, h, w, cin, cout, k)
    sc = CN.SmallConv(wt, b)
    y32 = launch('risp_conv2d_small', x, sc.wpack, b, n, h, w, cin, cout, k)
    (rms, mx), (rms32, mx32) = err(y, ref), err(y32, ref)
    assert not torch.isnan(y).any()
    # fp32-level accuracy: no worse than the fp32 vector-FMA kernel on the same data (slack for the tiny shapes), far inside 1e-4
    assert rms <= 1.25 * rms32 + 1e-9 and mx <= 2.0 * mx32 + 1e-8, (rms, rms32, mx, mx32)
    assert mx < 5e-6


@pytest.mark.parametrize('nhw', [(2, 16, 256), (3, 13, 68), (1, 40, 300)])
def test_epilogues_and_the_backward_data_packs(nhw):
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    E = CN
    # SRCNNRes tail: 5x5 32 -> 3 with bias + residual of 3 channels
    wt, b = rnd(3, 32, 5, 5, seed=5) * 0.05, rnd(3, seed=6) * 0.1
    x, add = rnd(n, 32, h, w, seed=7), rnd(n, 3, h, w, seed=8)
    lin = TF.conv2d(x.double(), wt.double(), b.double(), padding=2)
    p = CN.toep_weights(wt)
    for what, epi, a, ref in (('plain', 0, None, lin), ('relu', E.EPI_RELU, None, torch.relu(lin)), ('add', E.EPI_ADD, add, lin + add.double()),
                              ('add+relu', E.EPI_ADD | E.EPI_RELU, add, torch.relu(lin + add.double()))):
        y = launch('risp_conv2d_toep', x, p, b, n, h, w, 32, 3, 5, epi, a, 3 if a is not None else 0)
        assert err(y, ref)[1] < 3e-6, what
    # a residual narrower than the layer: only the first add_c couts take it
    wt4 = rnd(4, 8, 5, 5, seed=9) * 0.05
    x8, add2 = rnd(n, 8, h, w, seed=10), rnd(n, 2, h, w, seed=11)
    ref = TF.conv2d(x8.double(), wt4.double(), padding=2)
    ref[:, :2] += add2.double()
    y = launch('risp_conv2d_toep', x8, CN.toep_weights(wt4), None, n, h, w, 8, 4, 5, E.EPI_ADD, add2, 2)
    assert err(y, ref)[1] < 3e-6
    # SRCNNRes first layer, backward-data restricted to the image channels: forward weight (64, 12, 9, 9), gradient (n, 64, h, w)
    w1 = rnd(64, 12, 9, 9, seed=12) * 0.05
    g, gy = rnd(n, 64, h, w, seed=13) * 1e-4, rnd(n, 3, h, w, seed=14) * 1e-4
    ref = TF.conv_transpose2d(g.double(), w1[:, :3].double(), padding=4) + gy.double()
    y = launch('risp_conv2d_toep', g, CN.toep_weights(w1, True, 3), None, n, h, w, 64, 3, 9, E.EPI_ADD, gy, 3)
    assert err(y, ref)[1] < 5e-6
    # SRCNNDemosaic first layer, backward-data through PixelShuffle: forward weight (64, 4, 9, 9) on the unshuffled mosaic
    w1 = rnd(64, 4, 9, 9, seed=15) * 0.05
    ref = TF.pixel_shuffle(TF.conv_transpose2d(g.double(), w1.double(), padding=4), 2)
    y = launch('risp_conv2d_toep', g, CN.toep_weights(w1, True, 4), None, n, h, w, 64, 4, 9, E.EPI_SHUFFLE2)
    assert y.shape == (n, 1, 2 * h, 2 * w) and err(y, ref)[1] < 5e-6
    # SRCNNDemosaic tail: 5x5 32 -> 12 through PixelShuffle, with bias (srcnn_demosaic_arch.py:21-22)
    w3, b3 = rnd(12, 32, 5, 5, seed=17) * 0.05, rnd(12, seed=18) * 0.1
    ref = TF.pixel_shuffle(TF.conv2d(x.double(), w3.double(), b3.double(), padding=2), 2)
    y = launch('risp_conv2d_toep', x, CN.toep_weights(w3), b3, n, h, w, 32, 12, 5, E.EPI_SHUFFLE2)
    assert y.shape == (n, 3, 2 * h, 2 * w) and err(y, ref)[1] < 3e-6
    bias4 = rnd(4, seed=16) * 1e-4
    ref = TF.pixel_shuffle(TF.conv_transpose2d(g.double(), w1.double(), padding=4) + bias4.double().view(1, 4, 1, 1), 2)
    y = launch('risp_conv2d_toep', g, CN.toep_weights(w1, True, 4), bias4, n, h, w, 64, 4, 9, E.EPI_SHUFFLE2)
    assert err(y, ref)[1] < 5e-6


@pytest.mark.parametrize('scale', [1e-8, 1e-5, 1.0, 3e4])
def test_accuracy_does_not_depend_on_the_magnitude_of_the_input(scale):
    from reconfigisp_amd import convnets as CN
    n, h, w = 2, 24, 256
    w1 = rnd(64, 3, 9, 9, seed=21) * 0.05
    g = rnd(n, 64, h, w, seed=22) * scale * (rnd(n, 64, h, w, seed=23) > 0)
    ref = TF.conv_transpose2d(g.double(), w1.double(), padding=4)
    y = launch('risp_conv2d_toep', g, CN.toep_weights(w1, True, 3), None, n, h, w, 64, 3, 9)
    rms, mx = err(y, ref)
    assert rms < 3e-7 and mx < 4e-6, (scale, rms, mx)


def test_channels_of_very_different_magnitude_zeros_and_nan_locality():
    """the scale is taken per input channel and tile: channels of magnitude 1e-6 and 1e3 in one layer, a channel of zeros; a NaN
    poisons the outputs whose windows contain it - and, through the tile's maximum, nothing outside its own tile"""
    from reconfigisp_amd import convnets as CN
    n, h, w, c = 1, 48, 512, 16
    wt = rnd(3, c, 5, 5, seed=31) * 0.05
    x = rnd(n, c, h, w, seed=32)
    x[:, :4] *= 1e-6
    x[:, 8:12] *= 1e3
    x[:, 12:] = 0
    ref = TF.conv2d(x.double(), wt.double(), padding=2)
    y = launch('risp_conv2d_toep', x, CN.toep_weights(wt), None, n, h, w, c, 3, 5)
    assert err(y, ref)[1] < 3e-6
    z = launch('risp_conv2d_toep', torch.zeros_like(x), CN.toep_weights(wt), None, n, h, w, c, 3, 5)
    assert (z == 0).all()
    x[0, 5, 20, 300] = float('nan')                          # tile rows 16-31, columns 256-511
    y2 = launch('risp_conv2d_toep', x, CN.toep_weights(wt), None, n, h, w, c, 3, 5)
    bad = torch.isnan(y2[0]).any(0)
    assert bad[18:23, 298:303].all()
    bad[16:32, 256:] = False                                  # the NaN's own tile may be poisoned through its maximum
    assert not bad.any()
    keep = torch.ones(h, w, dtype=torch.bool, device='cuda')
    keep[16:32, 256:] = False
    assert torch.equal(y2[0][:, keep], y[0][:, keep])


def test_grouped_launch_equals_the_members_bit_for_bit_and_runs_are_repeatable():
    from reconfigisp_amd import convnets as CN, lib as L
    n, h, w, G = 2, 40, 260, 3
    ws = [rnd(64, 12, 9, 9, seed=40 + g) * 0.05 * (g + 1) for g in range(G)]
    packs = torch.stack([CN.toep_weights(wg, True, 3) for wg in ws])
    g1 = rnd(G * n, 64, h, w, seed=44) * 1e-3
    gy = rnd(G * n, 3, h, w, seed=45) * 1e-3
    y = launch('risp_conv2d_toep', g1, packs, None, n, h, w, 64, 3, 9, CN.EPI_ADD, gy, 3, group=(G, 0, None))
    for g in range(G):
        s = slice(g * n, (g + 1) * n)
        ym = launch('risp_conv2d_toep', g1[s].contiguous(), packs[g], None, n, h, w, 64, 3, 9, CN.EPI_ADD, gy[s].contiguous(), 3)
        assert torch.equal(y[s], ym), g
    assert torch.equal(y, launch('risp_conv2d_toep', g1, packs, None, n, h, w, 64, 3, 9, CN.EPI_ADD, gy, 3, group=(G, 0, None)))
    # forward tails sharing the residual operand (SHARED_ADD) and with per-member bias
    wt = [rnd(3, 32, 5, 5, seed=50 + g) * 0.05 for g in range(G)]
    bs = torch.stack([rnd(3, seed=60 + g) * 0.1 for g in range(G)])
    packs = torch.stack([CN.toep_weights(t) for t in wt])
    t2, x = rnd(G * n, 32, h, w, seed=70), rnd(n, 3, h, w, seed=71)
    y = launch('risp_conv2d_toep', t2, packs, bs, n, h, w, 32, 3, 5, CN.EPI_ADD, x, 3, group=(G, L.GROUP_SHARED_ADD, bs))
    for g in range(G):
        s = slice(g * n, (g + 1) * n)
        ym = launch('risp_conv2d_toep', t2[s].contiguous(), packs[g], bs[g], n, h, w, 32, 3, 5, CN.EPI_ADD, x, 3)
        assert torch.equal(y[s], ym), g
    # an image's result does not depend on the batch it travels in
    one = launch('risp_conv2d_toep', t2[3:4].contiguous(), packs[1], bs[1], 1, h, w, 32, 3, 5, CN.EPI_ADD, x[1:2].contiguous(), 3)
    assert torch.equal(one, y[3:4])


def test_arguments_outside_the_kernel_are_refused():
    from reconfigisp_amd import convnets as CN, lib as L
    x = rnd(1, 8, 16, 64, seed=80)
    p = CN.toep_weights(rnd(3, 8, 5, 5, seed=81))
    cases = [dict(cout=13), dict(cout=5, k=9), dict(k=3), dict(k=7), dict(w=62), dict(epi=4), dict(epi=8), dict(epi=2)]   # mask; shuffle with cout 3; add without tensor
    for c in cases:
        kw = dict(cout=3, k=5, w=64, epi=0)
        kw.update(c)
        with pytest.raises(RuntimeError, match='risp_conv2d_toep'):
            launch('risp_conv2d_toep', x, p, None, 1, 16, kw['w'], 8, kw['cout'], kw['k'], kw['epi'])
    assert L.load().risp_conv_toep_wpack_bytes(8, 3, 5) == p.numel() * 2


def test_conv_small_dispatch(monkeypatch):
    """``convnets.conv_small`` takes the matrix-pipe kernel for inference and for training grids that fill the chip, the vector
    kernel otherwise and under RISP_CONV_ARITH=f32; the per-member form of a grouped layer follows the grouped decision"""
    from reconfigisp_amd import convnets as CN, lib as L
    calls = []
    real = L.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    monkeypatch.setattr(CN.L, 'call', spy)
    monkeypatch.setattr(CN, 'TOEP_MIN_TILES', 256)
    wt, b = rnd(4, 32, 5, 5, seed=90) * 0.05, rnd(4, seed=91) * 0.1      # 4 couts: the band kernel's layer (3 couts: test_gpu_tapout.py)
    sc = CN.SmallConv(wt, b)
    x = rnd(2, 32, 32, 256, seed=92)
    ref = TF.conv2d(x.double(), wt.double(), b.double(), padding=2)

    def last():
        return [c for c in calls if c.startswith('risp_conv2d')][-1]
    y = CN.conv_small(x, sc, 2, 32, 256)                       # 4 tiles: the vector kernel (with its channel split)
    assert last().startswith('risp_conv2d_small') and err(y, ref)[1] < 3e-6
    y = CN.conv_small(x, sc, 2, 32, 256, infer=True)           # inference: always the same kernel, whatever the batch
    assert last() == 'risp_conv2d_toep' and err(y, ref)[1] < 3e-6
    monkeypatch.setattr(CN, 'TOEP_MIN_TILES', 4)
    y = CN.conv_small(x, sc, 2, 32, 256)
    assert last() == 'risp_conv2d_toep' and err(y, ref)[1] < 3e-6
    assert CN._small_split(x, sc, 2, 32, 256, 0, 0) == 0
    CN.conv_small(x[..., :128].contiguous(), sc, 2, 32, 128, infer=True)       # narrow planes: two rows folded into the 32 columns
    assert last() == 'risp_conv2d_toep'
    y = CN.conv_small(x, sc, 2, 32, 256, mask=torch.ones_like(ref, dtype=torch.float32), epi=CN.EPI_MASK)
    assert last().startswith('risp_conv2d_small')              # epilogues the kernel does not have stay where they were
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f32')
    y = CN.conv_small(x, sc, 2, 32, 256, infer=True)
    assert last().startswith('risp_conv2d_small') and err(y, ref)[1] < 3e-6
    assert CN._small_split(x, sc, 2, 32, 256, 0, 0) != 0


# --------------------------------------------------------------------------- the 9x9 first layers (risp_conv2d_toep_first)
def border_case(v, L, P=4):
    return v if v < P else (2 * P - (L - 1 - v) if v >= L - P else P)


def first_launch(x, pack, bias, n, h, w, cin, cout, epi=0, table=None, load=0, group=None):
    from reconfigisp_amd import lib as L
    G = group[0] if group else 1
    y = torch.full((G * n, cout, h, w), float('nan'), device='cuda')
    d = L.ConvDesc(N=G * n, H=h, W=w, cin=cin, cout=cout, ksize=9, load_mode=load, cin_img=0, epilogue=epi | (0 if bias is not None else 16),
                   add_c=0, x=x.data_ptr(), wpack=pack.data_ptr(), bias=bias.data_ptr() if bias is not None else None,
                   cvals=table.data_ptr() if table is not None else None, add=None, mask=None, y=y.data_ptr())
    if group:
        d.group_n, d.group_flags = n, group[1]
        d.wpack_gs = pack.stride(0) * pack.element_size() // 4
        d.bias_gs = bias.stride(0) if bias is not None else 0
    L.call('risp_conv2d_toep_first', C.byref(d), None)
    torch.cuda.synchronize()
    return y


def with_table(lin, table, h, w):
    """lin (n, cout, h, w) float64 + table (n, cout, 9, 9) looked up by the border case of the row and of the column"""
    iy = torch.tensor([border_case(v, h) for v in range(h)], device='cuda')
    ix = torch.tensor([border_case(v, w) for v in range(w)], device='cuda')
    return lin + table.double()[:, :, iy][:, :, :, ix]


@pytest.mark.parametrize('cout', [64, 48, 20])
@pytest.mark.parametrize('nhw', [(2, 8, 8), (1, 16, 32), (3, 20, 36), (2, 40, 72), (1, 64, 256), (2, 30, 260), (1, 9, 516)])
def test_first_layer_three_channels_with_case_table_next_to_the_fp32_kernel(cout, nhw):
    """SRCNNRes' folded first layer (srcnn_res_arch.py:18, 41-46): bias + border-case table + ReLU against float64, next to
    risp_conv2d_k3 on the same data"""
    from reconfigisp_amd import convnets as CN, lib as L
    n, h, w = nhw
    wt, b = rnd(cout, 3, 9, 9, seed=100) * 0.1, rnd(cout, seed=101) * 0.1
    x, table = rnd(n, 3, h, w, seed=102), rnd(n, cout, 9, 9, seed=103) * 0.1
    lin = TF.conv2d(x.double(), wt.double(), b.double(), padding=4)
    pack = CN.toep_first_weights(wt)
    assert L.load().risp_conv_toep_first_wpack_bytes(3, cout) == pack.numel() * 2
    y = first_launch(x, pack, b, n, h, w, 3, cout)
    assert err(y, lin)[1] < 3e-6
    y = first_launch(x, pack, b, n, h, w, 3, cout, CN.EPI_RELU)
    assert err(y, torch.relu(lin))[1] < 3e-6
    ref = torch.relu(with_table(lin, table, h, w))
    y = first_launch(x, pack, b, n, h, w, 3, cout, CN.EPI_RELU | CN.EPI_CASEBIAS, table)
    pc = CN.PackedConv(wt, b)
    y32 = torch.empty_like(y)
    d = L.ConvDesc(N=n, H=h, W=w, cin=3, cout=cout, ksize=9, load_mode=0, cin_img=0, epilogue=CN.EPI_RELU | CN.EPI_CASEBIAS, add_c=0,
                   x=x.data_ptr(), wpack=pc.k3.data_ptr(), bias=b.data_ptr(), cvals=table.data_ptr(), add=None, mask=None, y=y32.data_ptr())
    L.call('risp_conv2d_k3', C.byref(d), None)
    (rms, mx), (rms32, mx32) = err(y, ref), err(y32, ref)
    assert rms <= 1.25 * rms32 + 1e-9 and mx <= 2.0 * mx32 + 1e-8, (rms, rms32, mx, mx32)
    assert mx < 3e-6
    y = first_launch(x, pack, None, n, h, w, 3, cout, CN.EPI_CASEBIAS, table)              # no bias, no ReLU
    assert err(y, with_table(TF.conv2d(x.double(), wt.double(), padding=4), table, h, w))[1] < 3e-6


@pytest.mark.parametrize('cout', [64, 40])
@pytest.mark.parametrize('nhw', [(2, 8, 8), (3, 20, 36), (1, 128, 128), (2, 33, 260)])
def test_first_layer_on_the_mosaic(cout, nhw):
    """SRCNNDemosaic's first layer (srcnn_demosaic_arch.py:14-16, 39-43): the four planes are read out of the (N,1,2H,2W) mosaic"""
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    wt, b = rnd(cout, 4, 9, 9, seed=110) * 0.1, rnd(cout, seed=111) * 0.1
    bay = rnd(n, 1, 2 * h, 2 * w, seed=112)
    ref = torch.relu(TF.conv2d(TF.pixel_unshuffle(bay.double(), 2), wt.double(), b.double(), padding=4))
    y = first_launch(bay, CN.toep_first_weights(wt), b, n, h, w, 4, cout, CN.EPI_RELU, load=CN.LOAD_UNSHUFFLE2)
    assert not torch.isnan(y).any() and err(y, ref)[1] < 3e-6
    planes = TF.pixel_unshuffle(bay, 2).contiguous()              # the same layer fed with the planes: same bits
    assert torch.equal(y, first_launch(planes, CN.toep_first_weights(wt), b, n, h, w, 4, cout, CN.EPI_RELU))


def test_first_layer_grouped_on_one_shared_input_and_small_inputs():
    from reconfigisp_amd import convnets as CN, lib as L
    n, h, w, G, cout = 2, 40, 72, 3, 64
    ws = [rnd(cout, 3, 9, 9, seed=120 + g) * 0.1 * (g + 1) for g in range(G)]
    bs = torch.stack([rnd(cout, seed=130 + g) * 0.1 for g in range(G)])
    packs = torch.stack([CN.toep_first_weights(t) for t in ws])
    x, table = rnd(n, 3, h, w, seed=140) * 1e-6, rnd(G * n, cout, 9, 9, seed=141) * 1e-7
    epi = CN.EPI_RELU | CN.EPI_CASEBIAS
    y = first_launch(x, packs, bs * 1e-7, n, h, w, 3, cout, epi, table, group=(G, L.GROUP_SHARED_X))
    for g in range(G):
        s = slice(g * n, (g + 1) * n)
        ym = first_launch(x, packs[g], bs[g] * 1e-7, n, h, w, 3, cout, epi, table[s].contiguous())
        assert torch.equal(y[s], ym), g
        ref = torch.relu(with_table(TF.conv2d(x.double(), ws[g].double(), (bs[g] * 1e-7).double(), padding=4), table[s], h, w))
        assert err(ym, ref)[1] < 3e-6                              # inputs of magnitude 1e-6: the scale is the tile's own
    assert torch.equal(y, first_launch(x, packs, bs * 1e-7, n, h, w, 3, cout, epi, table, group=(G, L.GROUP_SHARED_X)))
    with pytest.raises(RuntimeError, match='risp_conv2d_toep_first'):
        first_launch(x, packs[0], bs[0], n, h, 70, 3, cout)                                 # W % 4
    with pytest.raises(RuntimeError, match='risp_conv2d_toep_first'):
        first_launch(x, packs[0], bs[0], n, h, w, 3, cout, CN.EPI_CASEBIAS)                # table missing
    with pytest.raises(RuntimeError, match='risp_conv2d_toep_first'):
        first_launch(x, packs[0], bs[0], n, h, w, 3, cout, CN.EPI_ADD)


def test_first_layer_dispatch(monkeypatch):
    """``convnets.conv`` sends the 9x9 first layers to risp_conv2d_toep_first, the 3x3 ones and RISP_CONV_ARITH=f32 to risp_conv2d_k3"""
    from reconfigisp_amd import convnets as CN, lib as L
    calls = []
    real = L.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)
    n, h, w = 2, 24, 200
    x = rnd(n, 3, h, w, seed=150)
    pc9, pc3 = CN.PackedConv(rnd(64, 3, 9, 9, seed=151) * 0.1, rnd(64, seed=152)), CN.PackedConv(rnd(64, 3, 3, 3, seed=153) * 0.1, rnd(64, seed=154))
    monkeypatch.setattr(CN.L, 'call', spy)
    assert CN.TOEP_FIRST == 'train'                                # the default: one first-layer kernel for inference and training
    y = CN.conv(x, pc9, n, h, w, epi=CN.EPI_RELU, infer=True)
    assert calls[-1] == 'risp_conv2d_toep_first'
    yt = CN.conv(x, pc9, n, h, w, epi=CN.EPI_RELU)                # training forward: the same kernel + exact recomputation of the ReLU ties
    assert calls[-1] == 'risp_conv2d_toep_first_exact' and (yt - y).abs().max().item() <= 1e-6 * y.abs().max().item()
    assert ((yt > 0) != (y > 0)).float().mean().item() < 1e-4     # ... which can only move outputs that sit on the kink
    monkeypatch.setattr(CN, 'TOEP_FIRST', 'infer')                 # round 4's default: training forwards on the fp32 kernel
    CN.conv(x, pc9, n, h, w, epi=CN.EPI_RELU)
    assert calls[-1] == 'risp_conv2d_k3'
    monkeypatch.setattr(CN, 'TOEP_FIRST', 'plain')
    assert torch.equal(y, CN.conv(x, pc9, n, h, w, epi=CN.EPI_RELU)) and calls[-1] == 'risp_conv2d_toep_first'
    monkeypatch.setattr(CN, 'TOEP_FIRST', '0')
    CN.conv(x, pc9, n, h, w, epi=CN.EPI_RELU, infer=True)
    assert calls[-1] == 'risp_conv2d_k3'
    monkeypatch.setattr(CN, 'TOEP_FIRST', 'train')
    CN.conv(x, pc3, n, h, w, epi=CN.EPI_RELU, infer=True)
    assert calls[-1] == 'risp_conv2d_k3'
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f32')
    y32 = CN.conv(x, pc9, n, h, w, epi=CN.EPI_RELU, infer=True)
    assert calls[-1] == 'risp_conv2d_k3'
    assert (y - y32).abs().max().item() < 1e-5 * y32.abs().max().item()


# --------------------------------------------------------------------------- rectangle sums out of the backward-data launch
@pytest.mark.parametrize('nchw', [(2, 64, 40, 72), (1, 64, 16, 256), (3, 16, 33, 260), (1, 8, 5, 8), (2, 3, 4, 4), (1, 64, 70, 516)])
def test_tile_sums_finish_into_the_rectangle_sums(nchw):
    """risp_conv2d_toep_sums writes per-tile sums of its input planes; risp_rect_sums_tiles turns them plus the border rows and
    columns into what risp_rect_sums computes from the whole planes (srcnn_res_arch.py:41-46: the gradient of the constant planes).
    Summation order: per thread its staged quads in task order, DPP butterflies, the four waves, the tiles in index order, then
    T - rows - columns + corner - another order than risp_rect_sums: agreement to 2e-6 of the largest sum, float64 as the referee."""
    from reconfigisp_amd import convnets as CN, lib as L
    n, c, h, w = nchw
    w1 = rnd(c, 3, 9, 9, seed=200) * 0.05
    g = rnd(n, c, h, w, seed=201)
    pack = CN.toep_weights(w1, True, 3)
    y0 = launch('risp_conv2d_toep', g, pack, None, n, h, w, c, 3, 9)
    tiles = L.load().risp_conv_toep_tiles(h, w)
    assert tiles == (((h + 31) // 32) * ((w + 127) // 128) if w <= 128 else ((h + 15) // 16) * ((w + 255) // 256))
    ps = torch.full((n, tiles, c), float('nan'), device='cuda')
    y = torch.full((n, 3, h, w), float('nan'), device='cuda')
    d = L.ConvDesc(N=n, H=h, W=w, cin=c, cout=3, ksize=9, load_mode=0, cin_img=0, epilogue=16, add_c=0, x=g.data_ptr(), wpack=pack.data_ptr(),
                   bias=None, cvals=None, add=None, mask=None, y=y.data_ptr())
    L.call('risp_conv2d_toep_sums', C.byref(d), C.c_void_p(ps.data_ptr()), None)
    assert torch.equal(y, y0)                                     # the convolution itself is untouched
    assert (ps.sum(1).double() - g.double().sum((2, 3))).abs().max().item() <= 2e-6 * g.double().abs().sum((2, 3)).max().item()
    rs_t, rs_f = torch.empty(n, c * 81, device='cuda'), torch.empty(n, c * 81, device='cuda')
    L.call('risp_rect_sums_tiles', C.c_void_p(g.data_ptr()), C.c_void_p(ps.data_ptr()), C.c_void_p(rs_t.data_ptr()), n, c, h, w, tiles, None)
    L.call('risp_rect_sums', C.c_void_p(g.data_ptr()), C.c_void_p(rs_f.data_ptr()), n * c, h, w, 9, None)
    torch.cuda.synchronize()
    gd = g.double().cpu().numpy()
    ref = np.zeros((n, c, 9, 9))
    for i in range(9):
        for j in range(9):
            dy, dx = i - 4, j - 4
            ys = slice(max(0, -dy), h - max(0, dy))
            xs = slice(max(0, -dx), w - max(0, dx))
            ref[:, :, i, j] = gd[:, :, ys, xs].sum((2, 3))
    ref = torch.from_numpy(ref.reshape(n, -1)).cuda()
    scale = np.abs(gd).sum((2, 3)).max()
    assert (rs_t.double() - ref).abs().max().item() <= 2e-6 * scale
    assert (rs_f.double() - ref).abs().max().item() <= 2e-6 * scale
    again = torch.empty_like(rs_t)
    L.call('risp_rect_sums_tiles', C.c_void_p(g.data_ptr()), C.c_void_p(ps.data_ptr()), C.c_void_p(again.data_ptr()), n, c, h, w, tiles, None)
    assert torch.equal(again, rs_t)


@pytest.mark.parametrize('hw', [(40, 72), (20, 260)])            # two rows folded into an instruction's columns / one 256-pixel strip
@pytest.mark.parametrize('mosaic', [False, True])
def test_first_layer_exact_relu_decisions(mosaic, hw):
    """risp_conv2d_toep_first_exact: outputs whose pre-activation lies within the arithmetic's error of zero are listed and recomputed in
    double - they equal the float64 convolution rounded once, and their ReLU decisions are float64's; everything else is the plain
    kernel's output bit for bit; a grouped launch on a shared input; a list that overflows is cut, not overrun."""
    from reconfigisp_amd import convnets as CN, lib as L
    n, (h, w), cout, G = 2, hw, 64, 2
    cin = 4 if mosaic else 3
    ws = torch.stack([rnd(cout, cin, 9, 9, seed=300 + g) * 0.1 for g in range(G)])
    bs = torch.stack([rnd(cout, seed=310 + g) * 0.05 for g in range(G)])
    packs = torch.stack([CN.toep_first_weights(ws[g]) for g in range(G)])
    x = rnd(n, 1, 2 * h, 2 * w, seed=320) if mosaic else rnd(n, 3, h, w, seed=320)
    table = None if mosaic else rnd(G * n, cout, 9, 9, seed=321) * 0.1
    epi = CN.EPI_RELU | (0 if mosaic else CN.EPI_CASEBIAS)
    load = CN.LOAD_UNSHUFFLE2 if mosaic else 0
    plain = first_launch(x, packs, bs, n, h, w, cin, cout, epi, table, load=load, group=(G, L.GROUP_SHARED_X))
    y = torch.full_like(plain, float('nan'))
    d = L.ConvDesc(N=G * n, H=h, W=w, cin=cin, cout=cout, ksize=9, load_mode=load, cin_img=0, epilogue=epi, add_c=0, x=x.data_ptr(),
                   wpack=packs.data_ptr(), bias=bs.data_ptr(), cvals=table.data_ptr() if table is not None else None, add=None, mask=None,
                   y=y.data_ptr())
    d.group_n, d.group_flags = n, L.GROUP_SHARED_X
    d.wpack_gs, d.bias_gs = packs.stride(0) * 2 // 4, bs.stride(0)
    cap = 1 << 16
    ties = torch.full((1 + cap,), -1, device='cuda', dtype=torch.int32)
    L.call('risp_conv2d_toep_first_exact', C.byref(d), C.c_void_p(ws.data_ptr()), ws.stride(0), C.c_void_p(ties.data_ptr()), cap, None)
    torch.cuda.synchronize()
    count = int(ties[0].item())
    assert 0 < count < cap, count                      # a few outputs in 10^4 sit that close to zero
    idx = ties[1:1 + count].long()
    assert idx.unique().numel() == count
    xs = TF.pixel_unshuffle(x.double(), 2) if mosaic else x.double()
    ref = torch.cat([TF.conv2d(xs, ws[g].double(), bs[g].double(), padding=4) for g in range(G)])
    if table is not None:
        ref = with_table(ref, table, h, w)
    same = torch.ones(y.numel(), dtype=torch.bool, device='cuda')
    same[idx] = False
    assert torch.equal(y.flatten()[same], plain.flatten()[same])                 # untouched outside the list
    zt, yt = ref.flatten()[idx], y.flatten()[idx]
    assert torch.equal(yt > 0, zt > 0)                                             # float64's ReLU decisions
    exact = torch.relu(zt).float()
    ulp = torch.finfo(torch.float32).eps * exact.abs().clamp_min(1e-30)
    assert ((yt - exact).abs() <= ulp).all()                                       # rounded once
    assert err(y, torch.relu(ref))[1] < 3e-6
    # the plain kernel's own decisions on the listed outputs: some differ from float64's - that is what the list is for
    assert ((plain.flatten()[idx] > 0) != (zt > 0)).sum().item() >= 0
    # a list too short for the ties of the launch is cut, not overrun: the counter keeps counting, the outputs beyond it stay the
    # plain kernel's
    tiny = torch.full((1 + 16 + 8,), -1, device='cuda', dtype=torch.int32)
    y2 = torch.full_like(plain, float('nan'))
    d.y = y2.data_ptr()
    L.call('risp_conv2d_toep_first_exact', C.byref(d), C.c_void_p(ws.data_ptr()), ws.stride(0), C.c_void_p(tiny.data_ptr()), 16, None)
    torch.cuda.synchronize()
    assert int(tiny[0].item()) == count and (tiny[17:] == -1).all()
    listed = tiny[1:17].long()
    rest = torch.ones(y.numel(), dtype=torch.bool, device='cuda')
    rest[listed] = False
    assert torch.equal(y2.flatten()[rest], plain.flatten()[rest]) and torch.equal(y2.flatten()[listed], y.flatten()[listed])
    # an all-zero input has no ties: exact zeros are exact in every arithmetic
    zero = torch.zeros_like(x)
    d.x, d.bias, d.epilogue, d.cvals, d.y = zero.data_ptr(), None, CN.EPI_RELU | CN.EPI_NOBIAS, None, y.data_ptr()
    L.call('risp_conv2d_toep_first_exact', C.byref(d), C.c_void_p(ws.data_ptr()), ws.stride(0), C.c_void_p(ties.data_ptr()), cap, None)
    torch.cuda.synchronize()
    assert int(ties[0].item()) == 0 and (y == 0).all()


def test_first_layer_tie_list_holds_far_more_ties_than_a_workgroup_collects():
    """the tap-index kernel collects an item's ties in a 511-entry LDS list and hands them over once per item; an item with more ties
    (here: half of the couts have zero weights and no bias - every one of their pre-activations is an exact zero beside a non-zero
    input, 40 960 ties per work item) sends the rest straight to the global list: every tie listed exactly once, none invented"""
    from reconfigisp_amd import convnets as CN, lib as L
    n, h, w, cout = 2, 40, 72, 64
    wt = rnd(cout, 3, 9, 9, seed=330) * 0.1
    wt[1::2] = 0.
    ws = wt.unsqueeze(0).contiguous()
    pack = CN.toep_first_weights(wt)
    x = rnd(n, 3, h, w, seed=331)
    y = torch.full((n, cout, h, w), float('nan'), device='cuda')
    d = L.ConvDesc(N=n, H=h, W=w, cin=3, cout=cout, ksize=9, load_mode=0, cin_img=0, epilogue=CN.EPI_RELU | CN.EPI_NOBIAS, add_c=0, x=x.data_ptr(),
                   wpack=pack.data_ptr(), bias=None, cvals=None, add=None, mask=None, y=y.data_ptr())
    cap = 1 << 20
    ties = torch.full((1 + cap,), -1, device='cuda', dtype=torch.int32)
    L.call('risp_conv2d_toep_first_exact', C.byref(d), C.c_void_p(ws.data_ptr()), 0, C.c_void_p(ties.data_ptr()), cap, None)
    torch.cuda.synchronize()
    count = int(ties[0].item())
    zero_outputs = n * (cout // 2) * h * w
    assert zero_outputs <= count < zero_outputs + 1000, (count, zero_outputs)          # + the few genuine near-zeros of the other couts
    idx = ties[1:1 + count].long()
    assert idx.unique().numel() == count
    listed = torch.zeros(y.numel(), dtype=torch.bool, device='cuda')
    listed[idx] = True
    assert listed.view(n, cout, h, w)[:, 1::2].all()
    ref = torch.relu(TF.conv2d(x.double(), wt.double(), padding=4))
    assert (y[:, 1::2] == 0).all() and err(y, ref)[1] < 3e-6
