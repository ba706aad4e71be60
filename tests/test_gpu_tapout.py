"""GPU parity of ``risp_conv2d_tapout`` (reconfigisp_amd/csrc/risp_conv_tapout.hip) through the C ABI: the layers with at most 3
output channels and whole chunks of 16 input channels, filter ROWS in the rows of the matrix instruction, one matrix pass per input
row, the vertical shift-add through a ring of output rows in LDS.  Against the float64 convolution next to the Toeplitz-band kernel
``risp_conv2d_toep`` it replaces for these layers: every epilogue, every segmentation, ragged shapes, gradient-sized inputs, grouped
launches, the channel sums of the backward-data launch, and the dispatch in ``convnets.conv_small``.
Layers: srcnn_res_arch.py:18 (backward-data, 64 -> 3), :22 (5x5 32 -> 3)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rnd(*shape, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(shape).astype(np.float32)).cuda()


def desc(x, pack, bias, n, h, w, cin, cout, k, epi, add, add_c, group, y):
    from reconfigisp_amd import lib as L
    G = group[0] if group else 1
    d = L.ConvDesc(N=G * n, H=h, W=w, cin=cin, cout=cout, ksize=k, load_mode=0, cin_img=0, epilogue=epi | (0 if bias is not None else 16),
                   add_c=add_c, x=x.data_ptr(), wpack=pack.data_ptr(), bias=bias.data_ptr() if bias is not None else None, cvals=None,
                   add=add.data_ptr() if add is not None else None, mask=None, y=y.data_ptr())
    if group:
        d.group_n, d.group_flags = n, group[1]
        d.wpack_gs = pack.stride(0) * pack.element_size() // 4
        d.bias_gs = bias.stride(0) if bias is not None else 0
    return d


def launch(x, pack, bias, n, h, w, cin, cout, k, epi=0, add=None, add_c=0, group=None, seg=0, entry='risp_conv2d_tapout'):
    from reconfigisp_amd import lib as L
    G = group[0] if group else 1
    y = torch.full((G * n, cout, h, w), float('nan'), device='cuda')
    d = desc(x, pack, bias, n, h, w, cin, cout, k, epi, add, add_c, group, y)
    if entry == 'risp_conv2d_tapout':
        L.call(entry, C.byref(d), seg, None)
    else:
        L.call(entry, C.byref(d), None)
    torch.cuda.synchronize()
    return y


def err(y, ref):
    m = ref.abs().max().item() or 1.0
    e = y.double() - ref
    return e.pow(2).mean().sqrt().item() / m, e.abs().max().item() / m


SHAPES = [(1, 16, 256), (2, 37, 64), (3, 16, 260), (1, 5, 8), (2, 50, 512), (1, 33, 4), (5, 20, 132), (2, 130, 256), (1, 300, 384)]


@pytest.mark.parametrize('k,cin,cout', [(9, 64, 3), (5, 32, 3), (9, 16, 2), (5, 48, 1)])
@pytest.mark.parametrize('nhw', SHAPES)
def test_forward_against_float64_next_to_the_band_kernel(k, cin, cout, nhw):
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    wt, b = rnd(cout, cin, k, k, seed=1) * 0.05, rnd(cout, seed=2) * 0.1
    x = rnd(n, cin, h, w, seed=3)
    ref = TF.conv2d(x.double(), wt.double(), b.double(), padding=k // 2)
    yb = launch(x, CN.toep_weights(wt), b, n, h, w, cin, cout, k, entry='risp_conv2d_toep')
    rms_b, mx_b = err(yb, ref)
    p = CN.tapout_weights(wt)
    for seg in (0, 32, 64, 8):
        y = launch(x, p, b, n, h, w, cin, cout, k, seg=seg)
        assert not torch.isnan(y).any(), seg
        rms, mx = err(y, ref)
        # fp32-level accuracy: no worse than the band kernel on the same data (slack for the tiny shapes), far inside 1e-4
        assert rms <= 1.5 * rms_b + 1e-8 and mx <= 2.5 * mx_b + 1e-7 and mx < 5e-6, (seg, rms, rms_b, mx, mx_b)


@pytest.mark.parametrize('nhw', [(2, 16, 256), (3, 13, 68), (1, 40, 300)])
def test_epilogues_and_the_backward_data_pack(nhw):
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    E = CN
    # SRCNNRes tail: 5x5 32 -> 3 with bias + residual of 3 channels
    wt, b = rnd(3, 32, 5, 5, seed=5) * 0.05, rnd(3, seed=6) * 0.1
    x, add = rnd(n, 32, h, w, seed=7), rnd(n, 3, h, w, seed=8)
    lin = TF.conv2d(x.double(), wt.double(), b.double(), padding=2)
    p = CN.tapout_weights(wt)
    for what, epi, a, ref in (('plain', 0, None, lin), ('relu', E.EPI_RELU, None, torch.relu(lin)), ('add', E.EPI_ADD, add, lin + add.double()),
                              ('add+relu', E.EPI_ADD | E.EPI_RELU, add, torch.relu(lin + add.double()))):
        y = launch(x, p, b, n, h, w, 32, 3, 5, epi, a, 3 if a is not None else 0)
        assert err(y, ref)[1] < 3e-6, what
    # a residual narrower than the layer: only the first add_c couts take it
    add2 = rnd(n, 2, h, w, seed=11)
    ref = lin.clone()
    ref[:, :2] += add2.double()
    assert err(launch(x, p, b, n, h, w, 32, 3, 5, E.EPI_ADD, add2, 2), ref)[1] < 3e-6
    # SRCNNRes first layer, backward-data restricted to the image channels: forward weight (64, 12, 9, 9), gradient (n, 64, h, w)
    w1 = rnd(64, 12, 9, 9, seed=12) * 0.05
    g, gy = rnd(n, 64, h, w, seed=13) * 1e-4, rnd(n, 3, h, w, seed=14) * 1e-4
    ref = TF.conv_transpose2d(g.double(), w1[:, :3].double(), padding=4) + gy.double()
    y = launch(g, CN.tapout_weights(w1, True, 3), None, n, h, w, 64, 3, 9, E.EPI_ADD, gy, 3)
    assert err(y, ref)[1] < 5e-6


@pytest.mark.parametrize('scale', [1e-8, 1e-5, 1.0, 3e4])
def test_accuracy_does_not_depend_on_the_magnitude_of_the_input(scale):
    from reconfigisp_amd import convnets as CN
    n, h, w = 2, 24, 256
    w1 = rnd(64, 3, 9, 9, seed=21) * 0.05
    g = rnd(n, 64, h, w, seed=22) * scale * (rnd(n, 64, h, w, seed=23) > 0)
    ref = TF.conv_transpose2d(g.double(), w1.double(), padding=4)
    y = launch(g, CN.tapout_weights(w1, True, 3), None, n, h, w, 64, 3, 9)
    rms, mx = err(y, ref)
    assert rms < 3e-7 and mx < 4e-6, (scale, rms, mx)


def test_chunks_of_very_different_magnitude_zeros_and_nan_locality():
    """the scale is taken per chunk of 16 channels and step tile (4 rows x 136 columns): chunks of magnitude 1e-6 and 1e3 in one layer, a
    chunk of zeros; a NaN poisons the outputs whose windows contain it - and, through its tile's maximum, nothing further than the
    rows and the strip its step reaches"""
    from reconfigisp_amd import convnets as CN
    n, h, w, c = 1, 48, 512, 64
    wt = rnd(3, c, 5, 5, seed=31) * 0.05
    x = rnd(n, c, h, w, seed=32)
    x[:, :16] *= 1e-6
    x[:, 32:48] *= 1e3
    x[:, 48:] = 0
    ref = TF.conv2d(x.double(), wt.double(), padding=2)
    p = CN.tapout_weights(wt)
    y = launch(x, p, None, n, h, w, c, 3, 5, seg=48)
    assert err(y, ref)[1] < 3e-6
    z = launch(torch.zeros_like(x), p, None, n, h, w, c, 3, 5)
    assert (z == 0).all()
    x[0, 5, 21, 300] = float('nan')          # segment rows 0-47: input rows -2 .. 49 in steps of 4 -> the step of rows 18-21; strip 256-383
    y2 = launch(x, p, None, n, h, w, c, 3, 5, seg=48)
    bad = torch.isnan(y2[0]).any(0)
    assert bad[19:24, 298:303].all()
    bad[16:24, 254:386] = False              # the NaN's own step tile (rows 18-21 + 2 rows of filter reach, the strip + its halo columns)
    assert not bad.any()
    keep = torch.ones(h, w, dtype=torch.bool, device='cuda')
    keep[16:24, 254:386] = False
    assert torch.equal(y2[0][:, keep], y[0][:, keep])


def test_grouped_launch_equals_the_members_bit_for_bit_and_runs_are_repeatable():
    from reconfigisp_amd import convnets as CN, lib as L
    n, h, w, G = 2, 40, 260, 3
    ws = [rnd(64, 12, 9, 9, seed=40 + g) * 0.05 * (g + 1) for g in range(G)]
    packs = torch.stack([CN.tapout_weights(wg, True, 3) for wg in ws])
    g1 = rnd(G * n, 64, h, w, seed=44) * 1e-3
    gy = rnd(G * n, 3, h, w, seed=45) * 1e-3
    seg = L.load().risp_conv_tapout_seg_rows(G * n, h, w)
    y = launch(g1, packs, None, n, h, w, 64, 3, 9, CN.EPI_ADD, gy, 3, group=(G, 0, None), seg=seg)
    assert torch.equal(y, launch(g1, packs, None, n, h, w, 64, 3, 9, CN.EPI_ADD, gy, 3, group=(G, 0, None)))      # seg 0 = that choice
    for g in range(G):
        s = slice(g * n, (g + 1) * n)
        ym = launch(g1[s].contiguous(), packs[g], None, n, h, w, 64, 3, 9, CN.EPI_ADD, gy[s].contiguous(), 3, seg=seg)
        assert torch.equal(y[s], ym), g
    for _ in range(3):
        assert torch.equal(y, launch(g1, packs, None, n, h, w, 64, 3, 9, CN.EPI_ADD, gy, 3, group=(G, 0, None), seg=seg))
    # forward tails sharing the residual operand (SHARED_ADD) and with per-member bias
    wt = [rnd(3, 32, 5, 5, seed=50 + g) * 0.05 for g in range(G)]
    bs = torch.stack([rnd(3, seed=60 + g) * 0.1 for g in range(G)])
    packs = torch.stack([CN.tapout_weights(t) for t in wt])
    t2, x = rnd(G * n, 32, h, w, seed=70), rnd(n, 3, h, w, seed=71)
    y = launch(t2, packs, bs, n, h, w, 32, 3, 5, CN.EPI_ADD, x, 3, group=(G, L.GROUP_SHARED_ADD, bs), seg=64)
    for g in range(G):
        s = slice(g * n, (g + 1) * n)
        ym = launch(t2[s].contiguous(), packs[g], bs[g], n, h, w, 32, 3, 5, CN.EPI_ADD, x, 3, seg=64)
        assert torch.equal(y[s], ym), g
    # with a fixed segment height an image's result does not depend on the batch it travels in
    one = launch(t2[3:4].contiguous(), packs[1], bs[1], 1, h, w, 32, 3, 5, CN.EPI_ADD, x[1:2].contiguous(), 3, seg=64)
    assert torch.equal(one, y[3:4])
    # a shared input (SHARED_X): the members read the same n images
    xs = rnd(n, 32, h, w, seed=72)
    y = launch(xs, packs, bs, n, h, w, 32, 3, 5, 0, None, 0, group=(G, L.GROUP_SHARED_X, bs), seg=64)
    for g in range(G):
        assert torch.equal(y[g * n:(g + 1) * n], launch(xs, packs[g], bs[g], n, h, w, 32, 3, 5, seg=64)), g


def test_arguments_outside_the_kernel_are_refused():
    from reconfigisp_amd import convnets as CN, lib as L
    x = rnd(1, 16, 16, 64, seed=80)
    p = CN.tapout_weights(rnd(3, 16, 5, 5, seed=81))
    cases = [dict(cout=4), dict(k=3), dict(k=7), dict(w=62), dict(epi=4), dict(epi=8), dict(epi=2), dict(cin=8), dict(seg=6), dict(seg=-4)]
    for c in cases:
        kw = dict(cout=3, k=5, w=64, epi=0, cin=16, seg=0)
        kw.update(c)
        with pytest.raises(RuntimeError, match='risp_conv2d_tapout'):
            launch(x, p, None, 1, 16, kw['w'], kw['cin'], kw['cout'], kw['k'], kw['epi'], seg=kw['seg'])
    assert L.load().risp_conv_tapout_wpack_bytes(16, 5) == p.numel() * 2
    lib = L.load()
    # the launch's own choice of the segment height: whole images when there are enough of them, never below 32 rows, a multiple of 4
    assert lib.risp_conv_tapout_seg_rows(256, 256, 256) == 256 and lib.risp_conv_tapout_seg_rows(1, 256, 256) == 32
    assert lib.risp_conv_tapout_seg_rows(1, 50, 64) == 50 and lib.risp_conv_tapout_seg_rows(1, 30, 64) == 30 and lib.risp_conv_tapout_seg_rows(1, 100, 64) == 52
    assert lib.risp_conv_tapout_items(1, 256, 256, 64) == 8 and lib.risp_conv_tapout_items(7, 256, 256, 0) == 2 * (256 // lib.risp_conv_tapout_seg_rows(7, 256, 256))


@pytest.mark.parametrize('nhw', [(2, 32, 256), (3, 45, 132), (1, 100, 260)])
def test_backward_data_launch_also_sums_its_input_planes(nhw):
    """risp_conv2d_tapout_sums writes per-work-item sums of its input planes; risp_rect_sums_tiles turns them plus the border rows and
    columns into the rectangle sums of the constant-plane gradient (srcnn_res_arch.py:41-46) - against risp_rect_sums on the planes"""
    from reconfigisp_amd import convnets as CN, lib as L
    n, h, w = nhw
    c = 64
    w1 = rnd(64, 12, 9, 9, seed=90) * 0.05
    g = rnd(n, c, h, w, seed=91)
    pack = CN.tapout_weights(w1, True, 3)
    for seg in (0, 32, 16):
        y0 = launch(g, pack, None, n, h, w, c, 3, 9, seg=seg)
        items = L.load().risp_conv_tapout_items(n, h, w, seg)
        ps = torch.full((n, items, c), float('nan'), device='cuda')
        y = torch.full((n, 3, h, w), float('nan'), device='cuda')
        d = desc(g, pack, None, n, h, w, c, 3, 9, 0, None, 0, None, y)
        L.call('risp_conv2d_tapout_sums', C.byref(d), seg, C.c_void_p(ps.data_ptr()), None)
        torch.cuda.synchronize()
        assert torch.equal(y, y0) and not torch.isnan(ps).any()
        tot = g.double().sum(dim=(2, 3))
        assert (ps.double().sum(1) - tot).abs().max().item() <= 1e-5 * g.abs().double().sum(dim=(2, 3)).max().item()
        rs = torch.empty((n, c * 81), device='cuda')
        rs0 = torch.empty((n, c * 81), device='cuda')
        L.call('risp_rect_sums_tiles', C.c_void_p(g.data_ptr()), C.c_void_p(ps.data_ptr()), C.c_void_p(rs.data_ptr()), n, c, h, w, items, None)
        L.call('risp_rect_sums', C.c_void_p(g.data_ptr()), C.c_void_p(rs0.data_ptr()), n * c, h, w, 9, None)
        torch.cuda.synchronize()
        assert (rs - rs0).abs().max().item() <= 2e-5 * g.abs().sum(dim=(2, 3)).max().item(), seg


def test_conv_small_dispatch(monkeypatch):
    """``convnets.conv_small`` takes the tap-row kernel for inference (fixed segment height) and for training grids that fill the chip,
    the vector kernel otherwise and under RISP_CONV_ARITH=f32; layers it cannot take stay on the band kernel"""
    from reconfigisp_amd import convnets as CN, lib as L
    calls = []
    real = L.call

    def spy(name, *a):
        calls.append((name, a))
        return real(name, *a)
    monkeypatch.setattr(CN.L, 'call', spy)
    wt, b = rnd(3, 32, 5, 5, seed=90) * 0.05, rnd(3, seed=91) * 0.1
    sc = CN.SmallConv(wt, b)
    x = rnd(2, 32, 64, 256, seed=92)
    ref = TF.conv2d(x.double(), wt.double(), b.double(), padding=2)

    def last():
        return [c for c in calls if c[0].startswith('risp_conv2d')][-1]
    y = CN.conv_small(x, sc, 2, 64, 256)                       # 2 images x 2 strips x 2 segments of 32 rows: the vector kernel
    assert last()[0].startswith('risp_conv2d_small') and err(y, ref)[1] < 3e-6
    y = CN.conv_small(x, sc, 2, 64, 256, infer=True)           # inference: always the same kernel and the same cut, whatever the batch
    assert last()[0] == 'risp_conv2d_tapout' and last()[1][1] == 64 and err(y, ref)[1] < 3e-6
    y1 = CN.conv_small(x[:1].contiguous(), sc, 1, 64, 256, infer=True)
    assert torch.equal(y1, y[:1])
    monkeypatch.setattr(CN, 'TAPOUT_MIN_ITEMS', 4)
    y = CN.conv_small(x, sc, 2, 64, 256)
    assert last()[0] == 'risp_conv2d_tapout' and last()[1][1] == L.load().risp_conv_tapout_seg_rows(2, 64, 256) and err(y, ref)[1] < 3e-6
    assert CN._small_split(x, sc, 2, 64, 256, 0, 0) == 0
    y = CN.conv_small(x, sc, 2, 64, 256, mask=torch.ones_like(ref, dtype=torch.float32), epi=CN.EPI_MASK)
    assert last()[0].startswith('risp_conv2d_small')           # epilogues the kernel does not have stay where they were
    sc4 = CN.SmallConv(rnd(4, 64, 9, 9, seed=93) * 0.05)       # 4 couts: the band kernel
    CN.conv_small(rnd(1, 64, 64, 256, seed=94), sc4, 1, 64, 256, infer=True)
    assert last()[0] == 'risp_conv2d_toep'
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f32')
    y = CN.conv_small(x, sc, 2, 64, 256, infer=True)
    assert last()[0].startswith('risp_conv2d_small') and err(y, ref)[1] < 3e-6
    assert CN._small_split(x, sc, 2, 64, 256, 0, 0) != 0
