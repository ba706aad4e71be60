"""GPU parity of ``risp_conv2d_thin5`` (reconfigisp_amd/csrc/risp_conv_thin5.hip) through the C ABI: 5x5 layers with at most 3 input
channels and 32 / 64 output channels in split precision - (filter row, channel) pairs as the reduction index of one matrix instruction.
Against the float64 convolution next to the fp32 Winograd kernel it replaces, every epilogue, ragged shapes, gradient-sized inputs,
grouped launches, repeatability, and the dispatch in ``convnets.conv``.  Layer: the backward-data pass of srcnn_res_arch.py:22 (3 -> 32,
masked by the ReLU of :20)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rnd(*shape, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(shape).astype(np.float32)).cuda()


def launch(x, pack, bias, n, h, w, cin, cout, epi=0, mask=None, group=None):
    from reconfigisp_amd import lib as L
    G = group[0] if group else 1
    y = torch.full((G * n, cout, h, w), float('nan'), device='cuda')
    d = L.ConvDesc(N=G * n, H=h, W=w, cin=cin, cout=cout, ksize=5, load_mode=0, cin_img=0, epilogue=epi | (0 if bias is not None else 16), add_c=0,
                   x=x.data_ptr(), wpack=pack.data_ptr(), bias=bias.data_ptr() if bias is not None else None, cvals=None, add=None,
                   mask=mask.data_ptr() if mask is not None else None, y=y.data_ptr())
    if group:
        d.group_n, d.group_flags = n, group[1]
        d.wpack_gs = pack.stride(0) * pack.element_size() // 4
        d.bias_gs = bias.stride(0) if bias is not None else 0
    L.call('risp_conv2d_thin5', C.byref(d), None)
    torch.cuda.synchronize()
    return y


def err(y, ref):
    m = ref.abs().max().item() or 1.0
    e = y.double() - ref
    return e.pow(2).mean().sqrt().item() / m, e.abs().max().item() / m


SHAPES = [(1, 16, 256), (2, 37, 64), (3, 33, 260), (1, 5, 8), (2, 70, 130), (1, 1, 4), (5, 20, 132), (1, 64, 37)]


@pytest.mark.parametrize('cin,cout', [(3, 32), (3, 64), (1, 32), (2, 64)])
@pytest.mark.parametrize('nhw', SHAPES)
def test_forward_against_float64_next_to_the_winograd_kernel(cin, cout, nhw):
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    wt, b = rnd(cout, cin, 5, 5, seed=1) * 0.05, rnd(cout, seed=2) * 0.1
    x = rnd(n, cin, h, w, seed=3)
    ref = torch.relu(TF.conv2d(x.double(), wt.double(), b.double(), padding=2))
    y = launch(x, CN.thin5_weights(wt), b, n, h, w, cin, cout, epi=CN.EPI_RELU)
    assert not torch.isnan(y).any()
    rms, mx = err(y, ref)
    assert rms < 1e-7 and mx < 2e-6, (rms, mx)
    if w % 4 == 0:                                                   # no worse than the fp32 kernel it replaces
        pc = CN.PackedConv(wt, b)
        pc.thin5_fwd = None
        y32 = CN.conv(x, pc, n, h, w, epi=CN.EPI_RELU)
        assert rms <= 1.5 * err(y32, ref)[0] + 1e-9


@pytest.mark.parametrize('nhw', [(2, 40, 72), (1, 33, 260)])
def test_backward_data_pack_mask_and_bias_free_epilogues(nhw):
    """the layer it serves: gy (3 channels) -> 32 channels through the transposed pack, masked by an activation tensor"""
    from reconfigisp_amd import convnets as CN
    n, h, w = nhw
    wf = rnd(3, 32, 5, 5, seed=10) * 0.05                             # the FORWARD layer's weight: 32 -> 3
    gy = rnd(n, 3, h, w, seed=11) * 1e-3
    act = rnd(n, 32, h, w, seed=12)
    act[0, :, :2] = 0.                                               # exact zeros in the mask: not > 0
    ref = TF.conv_transpose2d(gy.double(), wf.double(), padding=2) * (act > 0).double()
    y = launch(gy, CN.thin5_weights(wf, True), None, n, h, w, 3, 32, epi=CN.EPI_MASK, mask=act)
    rms, mx = err(y, ref)
    assert rms < 1e-7 and mx < 2e-6, (rms, mx)
    assert (y[0, :, :2] == 0).all()
    plain = launch(gy, CN.thin5_weights(wf, True), None, n, h, w, 3, 32)
    assert err(plain, TF.conv_transpose2d(gy.double(), wf.double(), padding=2))[1] < 2e-6
    assert torch.equal(y, torch.where(act > 0, plain, torch.zeros_like(plain)))      # the mask selects, nothing else changes
    # through conv(): the dispatch picks the kernel for the backward-data direction of a (3, 32, 5, 5) layer
    pc = CN.PackedConv(wf, rnd(3, seed=13))
    calls = []
    real = CN.L.call
    CN.L.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    try:
        yc = CN.conv(gy, pc, n, h, w, transpose=True, epi=CN.EPI_MASK, mask=act)
    finally:
        CN.L.call = real
    if w % 4 == 0 and CN.CONV_ARITH == 'f16x2':
        assert calls == ['risp_conv2d_thin5'] and torch.equal(yc, y)
    else:
        assert calls != ['risp_conv2d_thin5']


@pytest.mark.parametrize('scale', [1e-8, 1.0, 1e6])
def test_accuracy_does_not_depend_on_the_magnitude_of_the_input(scale):
    from reconfigisp_amd import convnets as CN
    n, h, w = 2, 48, 96
    wt = rnd(32, 3, 5, 5, seed=20) * 0.05
    x = rnd(n, 3, h, w, seed=21) * scale
    ref = TF.conv2d(x.double(), wt.double(), padding=2)
    y = launch(x, CN.thin5_weights(wt), None, n, h, w, 3, 32)
    assert err(y, ref)[0] < 1e-7


def test_regions_of_very_different_magnitude_zeros_and_nan_locality():
    """one scale per work item (image, 128-column strip, 32-row segment): a loud item does not cost a quiet one its precision; an
    all-zero image gives exact zeros; a NaN stays inside the reach of the filter"""
    from reconfigisp_amd import convnets as CN
    n, h, w = 3, 64, 256
    wt = rnd(32, 3, 5, 5, seed=30) * 0.05
    x = rnd(n, 3, h, w, seed=31)
    x[0, :, :32, :128] *= 1e6
    x[1] = 0.
    ref = TF.conv2d(x.double(), wt.double(), padding=2)
    y = launch(x, CN.thin5_weights(wt), None, n, h, w, 3, 32)
    quiet = ref[0, :, 36:, 132:]
    assert (y[0, :, 36:, 132:].double() - quiet).abs().max().item() < 2e-6 * quiet.abs().max().item()
    assert (y[1] == 0).all()
    x[2, 1, 40, 200] = float('nan')
    y = launch(x, CN.thin5_weights(wt), None, n, h, w, 3, 32)
    bad = torch.isnan(y[2])
    assert bad[:, 38:43, 198:203].all() and not bad[:, :32].any() and not bad[:, :, :128].any() and not torch.isnan(y[0]).any()


def test_grouped_launch_equals_the_members_bit_for_bit_and_runs_are_repeatable():
    from reconfigisp_amd import convnets as CN, lib as L
    G, n, h, w = 3, 2, 40, 136
    wfs = [rnd(3, 32, 5, 5, seed=40 + g) * 0.05 for g in range(G)]
    packs = torch.stack([CN.thin5_weights(t, True) for t in wfs])
    gy = rnd(G * n, 3, h, w, seed=50) * 1e-2
    act = rnd(G * n, 32, h, w, seed=51)
    yg = launch(gy, packs, None, n, h, w, 3, 32, epi=CN.EPI_MASK, mask=act, group=(G, 0))
    for g in range(G):
        ym = launch(gy[g * n:(g + 1) * n].contiguous(), packs[g], None, n, h, w, 3, 32, epi=CN.EPI_MASK, mask=act[g * n:(g + 1) * n].contiguous())
        assert torch.equal(yg[g * n:(g + 1) * n], ym)
    assert torch.equal(yg, launch(gy, packs, None, n, h, w, 3, 32, epi=CN.EPI_MASK, mask=act, group=(G, 0)))
    # one shared input for every member
    shared = launch(gy[:n].contiguous(), packs, None, n, h, w, 3, 32, group=(G, L.GROUP_SHARED_X))
    for g in range(G):
        assert torch.equal(shared[g * n:(g + 1) * n], launch(gy[:n].contiguous(), packs[g], None, n, h, w, 3, 32))
    # a result does not depend on the batch an image travels in
    alone = launch(gy[3:4].contiguous(), packs[1], None, 1, h, w, 3, 32)
    assert torch.equal(alone, launch(gy[2:4].contiguous(), packs[1], None, 2, h, w, 3, 32)[1:2])


def test_arguments_outside_the_kernel_are_refused():
    from reconfigisp_amd import convnets as CN
    wt = rnd(32, 3, 5, 5, seed=60)
    x = rnd(1, 3, 8, 8, seed=61)
    pack = CN.thin5_weights(wt)
    with pytest.raises(RuntimeError):
        launch(x, pack, None, 1, 8, 8, 4, 32)                        # 4 input channels
    with pytest.raises(RuntimeError):
        launch(x, pack, None, 1, 8, 8, 3, 48)                        # couts in blocks of 32
    with pytest.raises(RuntimeError):
        launch(x, pack, None, 1, 8, 8, 3, 32, epi=2)                 # residual epilogue
    with pytest.raises(RuntimeError):
        launch(x, pack, None, 1, 8, 8, 3, 32, epi=4)                 # mask flag without a mask
