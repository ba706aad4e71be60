"""GPU parity of single risp_conv2d launches: every load mode / epilogue flag / transpose pack, at tile-
aligned and ragged sizes, against plain PyTorch fp32 (the floating-point kernel's reference)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from conftest import assert_close

pytestmark = pytest.mark.gpu


def rnd(*shape, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(shape).astype(np.float32)).cuda()


SIZES = [(8, 8), (16, 32), (20, 36), (40, 72), (33, 30)]


@pytest.mark.parametrize('cin,cout,k', [(64, 64, 3), (4, 64, 3), (64, 4, 3), (3, 64, 3), (64, 3, 3), (64, 32, 5),
                                        (32, 3, 5), (32, 12, 5), (17, 64, 9), (4, 64, 9), (64, 32, 1)])
@pytest.mark.parametrize('hw', SIZES)
def test_forward_and_transpose(cin, cout, k, hw):
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n = 2
    wt, b = rnd(cout, cin, k, k, seed=1) * 0.1, rnd(cout, seed=2) * 0.1
    pc = CN.PackedConv(wt, b)
    x = rnd(n, cin, h, w, seed=3).requires_grad_(True)
    ref = TF.conv2d(x, wt, b, padding=k // 2)
    y = CN.conv(x.detach(), pc, n, h, w)
    assert_close(y, ref, what='fwd')
    gy = rnd(n, cout, h, w, seed=4)
    gref, = torch.autograd.grad(ref, x, gy)
    gx = CN.conv(gy, pc, n, h, w, transpose=True)
    assert_close(gx, gref, what='bwd-data')


@pytest.mark.parametrize('hw', SIZES)
def test_epilogues(hw):
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n, c = 2, 64
    wt, b = rnd(c, c, 3, 3, seed=5) * 0.05, rnd(c, seed=6) * 0.1
    pc = CN.PackedConv(wt, b)
    x, add, mask = rnd(n, c, h, w, seed=7), rnd(n, c, h, w, seed=8), rnd(n, c, h, w, seed=9)
    lin = TF.conv2d(x, wt, b, padding=1)
    assert_close(CN.conv(x, pc, n, h, w, epi=CN.EPI_RELU), torch.relu(lin), what='relu')
    assert_close(CN.conv(x, pc, n, h, w, epi=CN.EPI_ADD | CN.EPI_RELU, add=add, add_c=c), torch.relu(lin + add),
                 what='add+relu')
    assert_close(CN.conv(x, pc, n, h, w, epi=CN.EPI_MASK, mask=mask), lin * (mask > 0), what='mask')
    part = lin.clone()
    part[:, :3] += add[:, :3]
    assert_close(CN.conv(x, pc, n, h, w, epi=CN.EPI_ADD, add=add[:, :3].contiguous(), add_c=3), part, what='partial add')
    lin_t = TF.conv_transpose2d(x, wt, padding=1)       # == backward-data of the forward layer
    assert_close(CN.conv(x, pc, n, h, w, transpose=True, epi=CN.EPI_ADD | CN.EPI_MASK, add=add, add_c=c, mask=mask),
                 (lin_t + add) * (mask > 0), what='bwd add+mask')


@pytest.mark.parametrize('hw', [(8, 8), (20, 36), (33, 30)])
def test_unshuffle_load_and_shuffle_store(hw):
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n = 2
    wt, b = rnd(64, 4, 3, 3, seed=10) * 0.2, rnd(64, seed=11) * 0.1
    pc = CN.PackedConv(wt, b)
    bay = rnd(n, 1, 2 * h, 2 * w, seed=12)
    planes = TF.pixel_unshuffle(bay, 2)                                 # [R,G1,G2,B] = 2i+j order
    assert_close(CN.conv(bay, pc, n, h, w, load=CN.LOAD_UNSHUFFLE2), TF.conv2d(planes, wt, b, padding=1), what='unshuffle')
    wt2, b2 = rnd(12, 64, 3, 3, seed=13) * 0.05, rnd(12, seed=14) * 0.1
    pc2 = CN.PackedConv(wt2, b2)
    x = rnd(n, 64, h, w, seed=15)
    assert_close(CN.conv(x, pc2, n, h, w, epi=CN.EPI_SHUFFLE2), TF.pixel_shuffle(TF.conv2d(x, wt2, b2, padding=1), 2),
                 what='shuffle')
    g3 = rnd(n, 3, 2 * h, 2 * w, seed=16)                                # backward of the shuffle store
    assert_close(CN.conv(g3, pc2, n, h, w, transpose=True, load=CN.LOAD_UNSHUFFLE2),
                 TF.conv_transpose2d(TF.pixel_unshuffle(g3, 2), wt2, padding=1), what='bwd of shuffle')


def test_const_channel_load():
    from reconfigisp_amd import convnets as CN
    n, h, w, P = 2, 20, 36, 3
    wt, b = rnd(64, 12 + P, 9, 9, seed=17) * 0.05, rnd(64, seed=18) * 0.1
    pc = CN.PackedConv(wt, b)
    x, cv = rnd(n, 3, h, w, seed=19), rnd(n, 9 + P, seed=20)
    full = torch.cat([x, cv[:, :, None, None].expand(-1, -1, h, w)], dim=1)
    assert_close(CN.conv(x, pc, n, h, w, load=CN.LOAD_CONSTCH, cin_img=3, cvals=cv), TF.conv2d(full, wt, b, padding=4),
                 what='const channels')


@pytest.mark.parametrize('cin,cout,k,hw', [(32, 3, 5, (16, 16)), (64, 32, 5, (24, 40)), (17, 64, 9, (20, 36)),
                                           (64, 64, 3, (33, 30)), (3, 64, 3, (8, 8)), (3, 64, 9, (20, 36)), (6, 40, 5, (17, 19)),
                                           (40, 6, 5, (19, 35)), (24, 3, 9, (20, 36)), (10, 10, 3, (9, 18)), (5, 5, 1, (8, 8))])
def test_backward_weight(cin, cout, k, hw):
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n = 3
    wt = (rnd(cout, cin, k, k, seed=21) * 0.1).requires_grad_(True)
    b = (rnd(cout, seed=22) * 0.1).requires_grad_(True)
    x, gy = rnd(n, cin, h, w, seed=23), rnd(n, cout, h, w, seed=24)
    ref = TF.conv2d(x, wt, b, padding=k // 2)
    gw_ref, gb_ref = torch.autograd.grad(ref, (wt, b), gy)
    gw, gb = CN.conv_wgrad(x, gy, cin, cout, k, n, h, w)
    assert_close(gw, gw_ref, what='dW')
    assert_close(gb, gb_ref, what='db')
    gw2, gb2 = CN.conv_wgrad(x, gy, cin, cout, k, n, h, w)               # per-workgroup partial sums added in index order: no atomics
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    big = 40                                                            # more pixel tiles than slices: every workgroup walks several
    xb, gyb = rnd(big, cin, h, w, seed=25), rnd(big, cout, h, w, seed=26)
    gwb, _ = CN.conv_wgrad(xb, gyb, cin, cout, k, big, h, w)
    wd = wt.double().detach().requires_grad_(True)
    gwb_ref, = torch.autograd.grad(TF.conv2d(xb.double(), wd, None, padding=k // 2), wd, gyb.double())
    assert_close(gwb, gwb_ref, what='dW, 40 images')
    assert torch.equal(gwb, CN.conv_wgrad(xb, gyb, cin, cout, k, big, h, w)[0])


@pytest.mark.parametrize('P,hw', [(1, (20, 36)), (5, (33, 30))])
def test_backward_weight_of_the_first_layer_with_constant_planes(P, hw):
    """SRCNNRes' first layer (srcnn_res_arch.py:41-46): 3 image channels through the matrix kernel, the 9 + P constant planes as
    cvals^T @ rectangle sums - against autograd on the materialised (N, 12 + P, H, W) input"""
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n, k, cout = 3, 9, 64
    x, cv, gy = rnd(n, 3, h, w, seed=31), rnd(n, 9 + P, seed=32), rnd(n, cout, h, w, seed=33)
    wd = torch.zeros(cout, 12 + P, k, k, device='cuda', dtype=torch.float64, requires_grad=True)
    full = torch.cat([x, cv[:, :, None, None].expand(-1, -1, h, w)], dim=1).double()
    ref, = torch.autograd.grad(TF.conv2d(full, wd, None, padding=k // 2), wd, gy.double())
    gw, gb = CN.conv_wgrad(x, gy, 12 + P, cout, k, n, h, w, load=CN.LOAD_CONSTCH, cin_img=3, cvals=cv)
    assert_close(gw, ref, what='dW of the first layer')
    assert_close(gb, gy.double().sum(dim=(0, 2, 3)), what='db')
    assert torch.equal(gw, CN.conv_wgrad(x, gy, 12 + P, cout, k, n, h, w, load=CN.LOAD_CONSTCH, cin_img=3, cvals=cv)[0])


def test_srcnn_res_weight_gradients_vs_oracle():
    """Proxy fine-tuning path: gradients of an MSE-like loss w.r.t. all six SRCNNRes tensors vs CPU autograd."""
    import isp_oracle as O
    from reconfigisp_amd.codes.models.modules.tools_proxy import ProxyNet
    P, n, h, w = 3, 2, 24, 40
    wts = O.make_weights('srcnn_res', 55, P)
    for k in list(wts):                                   # kink-free (see test_gpu_cnn.kink_free_weights)
        if k.endswith('.bias') and not k.startswith('srcnn.4'):
            s = torch.ones_like(wts[k]); s[1::2] = -1.0; wts[k] = s
        elif k.endswith('.weight'):
            wts[k] = wts[k] * 0.02
    x, pv = rnd(n, 3, h, w, seed=25).cpu().abs().clamp(0, 1), rnd(n, P, seed=26).cpu().abs().clamp(0, 1)
    gy = rnd(n, 3, h, w, seed=27).cpu()
    leaf = {k: v.clone().requires_grad_(True) for k, v in wts.items()}
    yc = O.srcnn_res(x, pv, leaf)
    keys = sorted(leaf)
    gref = torch.autograd.grad(yc, [leaf[k] for k in keys], gy)
    m = ProxyNet(P, None)
    m.load_state_dict(wts)
    m = m.cuda()
    m.train_weights = True
    yg = m(x.cuda(), pv.cuda())
    assert_close(yg, yc, what='y')
    params = dict(m.named_parameters())
    ggot = torch.autograd.grad(yg, [params[k] for k in keys], gy.cuda())
    for k, a, b in zip(keys, ggot, gref):
        assert_close(a, b, rtol=2e-4, what='grad ' + k)


# ---------------------------------------------------------------------------------------------------
# direct small-cout kernel (risp_conv2d_small), rectangle sums, border-case bias
@pytest.mark.parametrize('cin,cout,k', [(32, 3, 5), (64, 3, 9), (64, 4, 3), (5, 1, 3), (13, 2, 9), (64, 3, 3)])
@pytest.mark.parametrize('hw', SIZES + [(9, 10), (17, 66)])
def test_small_cout_forward_and_backward_data(cin, cout, k, hw):
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n = 2
    wt, b = rnd(cout, cin, k, k, seed=11) * 0.1, rnd(cout, seed=12) * 0.1
    x = rnd(n, cin, h, w, seed=13)
    ref = TF.conv2d(x, wt, b, padding=k // 2)
    assert_close(CN.conv_small(x, CN.SmallConv(wt, b), n, h, w), ref, what='small fwd')
    # backward-data of a FORWARD layer (cout_f -> the first `keep` input channels)
    wf = rnd(cin, 7, k, k, seed=14) * 0.1                     # forward layer 7 -> cin channels
    keep = min(cout, 4)
    xin = rnd(n, 7, h, w, seed=15).requires_grad_(True)
    gref, = torch.autograd.grad(TF.conv2d(xin, wf, None, padding=k // 2), xin, x)
    got = CN.conv_small(x, CN.SmallConv(wf, None, transpose=True, keep=keep), n, h, w)
    assert_close(got, gref[:, :keep], what='small bwd-data')


@pytest.mark.parametrize('cin,cout', [(64, 32), (32, 64), (32, 3), (8, 40)])
@pytest.mark.parametrize('hw', [(33, 36), (7, 132), (1, 8), (2, 260), (130, 128)])
def test_f45_two_rows_per_wave_ragged(cin, cout, hw):
    """conv_wino45_r2_kernel: a wave owns two output rows - odd heights (the second row of the last pair does not exist), a
    single row, widths that leave the last 64-pixel half of a tile partly or wholly outside the image; forward with every
    epilogue, and the backward-data pack (for 32 -> 3 that is the 3 -> 32 launch with a partial channel chunk)."""
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n = 3
    wt, b = rnd(cout, cin, 5, 5, seed=91) * 0.05, rnd(cout, seed=92) * 0.1
    pc = CN.PackedConv(wt, b)
    x, add, mask = rnd(n, cin, h, w, seed=93), rnd(n, cout, h, w, seed=94), rnd(n, cout, h, w, seed=95)
    if pc.wino45_fwd is not None:
        lin = TF.conv2d(x, wt, b, padding=2)
        assert_close(CN.conv(x, pc, n, h, w), lin, what='plain')
        assert_close(CN.conv(x, pc, n, h, w, epi=CN.EPI_ADD | CN.EPI_RELU, add=add, add_c=cout), torch.relu(lin + add), what='add+relu')
        assert_close(CN.conv(x, pc, n, h, w, epi=CN.EPI_MASK, mask=mask), lin * (mask > 0), what='mask')
    if pc.wino45_bwd is not None:
        gy, xm = rnd(n, cout, h, w, seed=96), rnd(n, cin, h, w, seed=97)
        ref = TF.conv_transpose2d(gy, wt, padding=2) * (xm > 0)
        assert_close(CN.conv(gy, pc, n, h, w, transpose=True, epi=CN.EPI_MASK, mask=xm), ref, what='bwd-data + mask')


@pytest.mark.parametrize('k,cin', [(3, 16), (5, 12), (9, 6)])
def test_small_three_couts_on_four_lanes(k, cin):
    """conv_small_kernel's C3 form (3 couts: the fourth lane of the packed FMA carries the third cout of the thread's second
    output row) is what grids of >= 384 tiles and the channel split on 64 x 32 tiles launch; small test images never reach it.
    Ragged image (odd height, last tile column partial), every epilogue, and the same layer through the one-row form
    (inference: no split, small grid) - the four-lane form adds exact zeros only, so the two agree to the bit."""
    from reconfigisp_amd import convnets as CN
    h, w = 33, 132
    wt, b = rnd(3, cin, k, k, seed=31) * 0.1, rnd(3, seed=32) * 0.1
    sc = CN.SmallConv(wt, b)
    n = 64                                                 # 3 x 2 tiles x 64 images = 384 tiles: the full-grid form
    x, add, mask = rnd(n, cin, h, w, seed=33), rnd(n, 3, h, w, seed=34), rnd(n, 3, h, w, seed=35)
    lin = TF.conv2d(x, wt, b, padding=k // 2)
    y = CN.conv_small(x, sc, n, h, w, infer=True)
    assert_close(y, lin, what='plain')
    assert_close(CN.conv_small(x, sc, n, h, w, epi=CN.EPI_ADD | CN.EPI_RELU, add=add, add_c=3, infer=True), torch.relu(lin + add), what='add+relu')
    assert_close(CN.conv_small(x, sc, n, h, w, epi=CN.EPI_MASK, mask=mask, infer=True), lin * (mask > 0), what='mask')
    one_row = torch.cat([CN.conv_small(x[i:i + 8], sc, 8, h, w, infer=True) for i in range(0, n, 8)])      # 48 tiles: 64 x 16 form
    assert torch.equal(y, one_row)
    m = 8                                                  # channel split on 64 x 32 tiles: 3 x 2 x 8 images x 8 groups
    ys = CN.conv_small(x[:m], sc, m, h, w, epi=CN.EPI_ADD, add=add[:m], add_c=3, split=min(8, cin // 2))
    assert_close(ys, lin[:m] + add[:m], what='split')


@pytest.mark.parametrize('hw', [(16, 32), (33, 30), (40, 72)])
def test_small_cout_epilogues(hw):
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n, cin = 2, 32
    wt, b = rnd(3, cin, 5, 5, seed=21) * 0.1, rnd(3, seed=22) * 0.1
    sc = CN.SmallConv(wt, b)
    x, add, mask = rnd(n, cin, h, w, seed=23), rnd(n, 3, h, w, seed=24), rnd(n, 3, h, w, seed=25)
    lin = TF.conv2d(x, wt, b, padding=2)
    assert_close(CN.conv_small(x, sc, n, h, w, epi=CN.EPI_RELU), torch.relu(lin), what='relu')
    assert_close(CN.conv_small(x, sc, n, h, w, epi=CN.EPI_ADD | CN.EPI_RELU, add=add, add_c=3), torch.relu(lin + add),
                 what='add+relu')
    assert_close(CN.conv_small(x, sc, n, h, w, epi=CN.EPI_MASK, mask=mask), lin * (mask > 0), what='mask')
    part = lin.clone()
    part[:, :2] += add[:, :2]
    assert_close(CN.conv_small(x, sc, n, h, w, epi=CN.EPI_ADD, add=add[:, :2].contiguous(), add_c=2), part, what='partial add')


@pytest.mark.parametrize('k', [3, 5, 9])
@pytest.mark.parametrize('hw', [(8, 8), (9, 13), (33, 30), (64, 256)])
def test_rect_sums_are_the_constant_channel_gradient(k, hw):
    """risp_rect_sums(g) @ w == d/dc of sum(g * conv(c * ones)) for a spatially constant, zero-padded channel."""
    from reconfigisp_amd import lib as L
    from reconfigisp_amd.functional import _p, _stream
    h, w = hw
    n, cout, nconst = 2, 6, 4
    g = rnd(n, cout, h, w, seed=31)
    wt = rnd(cout, nconst, k, k, seed=32)
    rs = torch.empty((n, cout * k * k), device='cuda')
    L.call('risp_rect_sums', _p(g), _p(rs), n * cout, h, w, k, _stream())
    wconst = wt.permute(0, 2, 3, 1).reshape(-1, nconst).contiguous()
    got = torch.empty((n, nconst), device='cuda')
    L.call('risp_srcnn_const_grad', _p(rs), _p(wconst), _p(got), n, rs.shape[1], nconst, _stream())
    assert_close(got, rs.double() @ wconst.double(), what='const grad product', rtol=1e-5, floor=1.0)
    cv = rnd(n, nconst, seed=33).requires_grad_(True)
    planes = cv[:, :, None, None].expand(n, nconst, h, w)
    (TF.conv2d(planes, wt, None, padding=k // 2) * g).sum().backward()
    assert_close(got, cv.grad, what='const-channel grad', rtol=2e-4, floor=1.0)


@pytest.mark.parametrize('P', [0, 2, 5])
def test_srcnn_case_table(P):
    """risp_srcnn_case_table = (min | mean | max | params) @ rcase, against torch on the statistics torch computes."""
    from reconfigisp_amd import lib as L
    from reconfigisp_amd.functional import _p, _stream, channel_stats
    n, h, w, m = 3, 12, 20, 64 * 81
    x = torch.rand(n, 3, h, w, device='cuda')
    pv = torch.rand(n, P, device='cuda') if P else None
    rcase = rnd(9 + P, m, seed=35)
    stats, _ = channel_stats(x)
    table = torch.empty((n, m), device='cuda')
    L.call('risp_srcnn_case_table', _p(stats), _p(pv), _p(rcase), _p(table), n, P, h * w, m, _stream())
    cv = torch.cat([x.amin(dim=(2, 3)), x.mean(dim=(2, 3)), x.amax(dim=(2, 3))] + ([pv] if P else []), dim=1)
    assert_close(table, cv.double() @ rcase.double(), what='case table', rtol=1e-5, floor=1.0)


@pytest.mark.parametrize('P', [1, 3, 5])
@pytest.mark.parametrize("hw", [(8, 8), (8, 12), (20, 36), (40, 72), (34, 30)])
def test_srcnn_res_folded_equals_unfolded(P, hw):
    """The broadcast planes folded out of the 9x9 layer (border-case bias forward, rectangle sums backward, direct
    kernels for the 3-channel ends) against the plain 12+P-channel convolution path: outputs, input gradient,
    parameter gradient."""
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n = 2
    seq = torch.nn.Sequential(torch.nn.Conv2d(12 + P, 64, 9, padding=4), torch.nn.ReLU(), torch.nn.Conv2d(64, 32, 5, padding=2),
                              torch.nn.ReLU(), torch.nn.Conv2d(32, 3, 5, padding=2)).cuda()
    torch.manual_seed(5)
    for m in seq:
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.normal_(m.weight, std=0.05)
            torch.nn.init.normal_(m.bias, std=0.1)
    packs = CN.build_srcnn_packs(seq, residual=True)
    assert packs[0].fold is not None
    x0, pv0, gy = torch.rand(n, 3, h, w, device='cuda'), torch.rand(n, P, device='cuda'), rnd(n, 3, h, w, seed=41)
    res = []
    for fn in (CN._SrcnnResFolded, CN._SrcnnRes):
        x, pv = x0.clone().requires_grad_(True), pv0.clone().requires_grad_(True)
        y = fn.apply(x, pv, packs)
        y.backward(gy)
        res.append((y.detach(), x.grad, pv.grad))
    # The two paths are two fp32 arithmetics of the same layers (the folded first layer even decides its ReLU ties exactly, convnets.
    # TOEP_FIRST): a pre-activation within rounding of zero may be clipped by one and passed by the other, and the input gradient then
    # differs by O(1) over that activation's footprint (9 x 9 for the first layer, 13 x 13 through it for the second).  Ties are
    # located with a float64 evaluation; the comparison holds everywhere outside their footprints (typically everywhere).
    with torch.no_grad():
        xd = x0.double()
        feat = torch.cat([xd.amin(dim=(2, 3)), xd.mean(dim=3).mean(dim=2), xd.amax(dim=(2, 3)), pv0.double()], dim=1)
        pre1 = TF.conv2d(torch.cat([xd, feat[:, :, None, None].expand(-1, -1, h, w)], dim=1), seq[0].weight.double(), seq[0].bias.double(), padding=4)
        pre2 = TF.conv2d(torch.relu(pre1), seq[2].weight.double(), seq[2].bias.double(), padding=2)
        tie = lambda pre: (pre.abs() < 2e-6 * pre.abs().max()).any(dim=1, keepdim=True).float()      # ~10 x the kernels' rounding
        near = (TF.max_pool2d(tie(pre1), 9, 1, 4) + TF.max_pool2d(tie(pre2), 13, 1, 6)) > 0            # (n, 1, h, w)
        # ... and the pixels that hold a channel's minimum / maximum: the gradient of the min / max planes (srcnn_res_arch.py:36-40) lands
        # there, and a flipped activation changes it by that activation's whole contribution
        flat = x0.flatten(2)
        for i in range(n):
            if near[i].any():
                for pos in torch.cat([flat[i].argmin(dim=1), flat[i].argmax(dim=1)]).tolist():
                    near[i, 0, pos // w, pos % w] = True
    assert h * w < 1000 or near.float().mean().item() < 0.25           # (a tie's footprint covers most of a tiny image)
    for a, b, what in zip(res[0], res[1], ('output', 'input grad', 'param grad')):
        if what == 'input grad':
            a, b = torch.where(near, torch.zeros_like(a), a), torch.where(near, torch.zeros_like(b), b)
        if what == 'param grad':                       # a sum over the image: an image with a tie carries that activation's whole share
            tied = torch.tensor([bool(near[i].any()) for i in range(n)], device=a.device)
            assert_close(a[tied], b[tied], what=what + ' (images with a ReLU tie)', rtol=2e-2, floor=1.0)
            a, b = a[~tied], b[~tied]
        assert_close(a, b, what=what, rtol=2e-4, floor=1.0)


@pytest.mark.parametrize('hw', [(8, 8), (20, 36), (33, 30)])
def test_small_cout_pixelshuffle_store(hw):
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n, cin = 2, 64
    wt, b = rnd(4, cin, 3, 3, seed=51) * 0.1, rnd(4, seed=52) * 0.1
    x = rnd(n, cin, h, w, seed=53)
    ref = TF.pixel_shuffle(TF.conv2d(x, wt, b, padding=1), 2)
    assert_close(CN.conv_small(x, CN.SmallConv(wt, b), n, h, w, epi=CN.EPI_SHUFFLE2), ref, what='small shuffle2')


@pytest.mark.parametrize('cout', [5, 8, 12])
@pytest.mark.parametrize('hw', [(8, 8), (20, 36), (33, 30)])
def test_small_cout_up_to_12(cout, hw):
    """cout <= 12 form of the direct kernel (SRCNNDemosaic's 5x5 32 -> 12 tail), plain and through PixelShuffle."""
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n, cin = 2, 32
    wt, b = rnd(cout, cin, 5, 5, seed=61) * 0.1, rnd(cout, seed=62) * 0.1
    x = rnd(n, cin, h, w, seed=63)
    lin = TF.conv2d(x, wt, b, padding=2)
    sc = CN.SmallConv(wt, b)
    assert_close(CN.conv_small(x, sc, n, h, w), lin, what='small cout<=12')
    if cout % 4 == 0:
        assert_close(CN.conv_small(x, sc, n, h, w, epi=CN.EPI_SHUFFLE2), TF.pixel_shuffle(lin, 2), what='small shuffle2 groups')


@pytest.mark.parametrize('cin,cout,k', [(64, 4, 9), (32, 12, 5), (64, 4, 3)])
@pytest.mark.parametrize('groups', [2, 5, 8])
def test_small_split_pixelshuffle_store(cin, cout, k, groups):
    """Channel-group split of the direct kernel with the PixelShuffle store (first-layer backward and tail of
    SRCNNDemosaic at small batches): against PyTorch and against the single-launch form."""
    import ctypes as C
    from reconfigisp_amd import convnets as CN, lib as L
    n, h, w = 2, 20, 36
    wt, b = rnd(cout, cin, k, k, seed=81) * 0.1, rnd(cout, seed=82) * 0.1
    x = rnd(n, cin, h, w, seed=83)
    ref = TF.pixel_shuffle(TF.conv2d(x, wt, b, padding=k // 2), 2)
    sc = CN.SmallConv(wt, b)
    outs = []
    for g in (1, groups):
        y = torch.zeros_like(ref)
        scratch = torch.empty((g, n, cout, h, w), device=x.device)
        d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=k, load_mode=CN.LOAD_PLAIN, cin_img=0, epilogue=CN.EPI_SHUFFLE2,
                       add_c=0, x=x.data_ptr(), wpack=sc.wpack.data_ptr(), bias=sc.bias.data_ptr(), cvals=None, add=None,
                       mask=None, y=y.data_ptr())
        L.call('risp_conv2d_small_split', C.byref(d), scratch.data_ptr(), g, None)
        assert_close(y, ref, what='%d groups' % g)
        outs.append(y)
    assert_close(outs[1], outs[0], what='split vs single launch')      # summation order differs: fp32 noise only


@pytest.mark.parametrize('cin,cout', [(64, 64), (4, 64), (64, 33), (3, 64), (64, 3)])
@pytest.mark.parametrize('hw', SIZES + [(12, 128), (6, 260)])
def test_inference_dispatch_f43(cin, cout, hw, monkeypatch):
    """3x3 layers on the fp32 route go through risp_conv2d_wino43 (F(4,3)), inference and training launches alike: against PyTorch,
    every epilogue the kernel has."""
    from reconfigisp_amd import convnets as CN
    h, w = hw
    n = 2
    wt, b = rnd(cout, cin, 3, 3, seed=71) * 0.1, rnd(cout, seed=72) * 0.1
    pc = CN.PackedConv(wt, b)
    assert pc.wino43_fwd is not None
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f32')
    x, add, mask = rnd(n, cin, h, w, seed=73), rnd(n, cout, h, w, seed=74), rnd(n, cout, h, w, seed=75)
    lin = TF.conv2d(x, wt, b, padding=1)
    for infer in (True, False):
        assert_close(CN.conv(x, pc, n, h, w, infer=infer), lin, what='plain infer=%s' % infer)
        assert_close(CN.conv(x, pc, n, h, w, epi=CN.EPI_ADD | CN.EPI_RELU, add=add, add_c=cout, infer=infer),
                     torch.relu(lin + add), what='add+relu infer=%s' % infer)
        assert_close(CN.conv(x, pc, n, h, w, epi=CN.EPI_MASK, mask=mask, infer=infer), lin * (mask > 0),
                     what='mask infer=%s' % infer)


@pytest.mark.parametrize('cout', [64, 33])
@pytest.mark.parametrize('hw', [(8, 8), (20, 36), (33, 32), (64, 256), (6, 260)])
def test_f43_both_cout_blocks_in_one_wave_equals_per_block_launches(cout, hw, monkeypatch):
    """conv_wino43_b2_kernel keeps both 32-cout blocks of a 33..64-cout layer in one wave (one input transform feeds twelve
    matrix instructions); per output it performs the one-block kernel's arithmetic in the same order, so the layer launched
    as two separate 32-cout layers (the one-block kernel) must give the same bits - forward and backward-data packs, every
    epilogue.  (The fp32 route: RISP_CONV_ARITH=f32.)"""
    from reconfigisp_amd import convnets as CN
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f32')
    h, w = hw
    n, cin = 3, 64
    wt, b = rnd(cout, cin, 3, 3, seed=81) * 0.05, rnd(cout, seed=82) * 0.1
    x, add, mask = rnd(n, cin, h, w, seed=83), rnd(n, cout, h, w, seed=84), rnd(n, cout, h, w, seed=85)
    whole = CN.PackedConv(wt, b)
    parts = [(CN.PackedConv(wt[lo:hi].contiguous(), b[lo:hi].contiguous()), lo, hi) for lo, hi in ((0, 32), (32, cout))]
    for kw in (dict(), dict(epi=CN.EPI_RELU), dict(epi=CN.EPI_ADD | CN.EPI_RELU, add=add, add_c=cout), dict(epi=CN.EPI_MASK, mask=mask)):
        y = CN.conv(x, whole, n, h, w, infer=True, **kw)
        for pc, lo, hi in parts:
            kws = dict(kw)
            if 'add' in kws:
                kws.update(add=add[:, lo:hi].contiguous(), add_c=hi - lo)
            if 'mask' in kws:
                kws.update(mask=mask[:, lo:hi].contiguous())
            yp = CN.conv(x, pc, n, h, w, infer=True, **kws)
            assert torch.equal(y[:, lo:hi], yp), 'forward %s couts %d..%d' % (kw.get('epi'), lo, hi)
    if cout == 64:                                        # backward-data of a cin-64 layer with 64 couts is a 64 -> 64 layer again
        gy = rnd(n, cout, h, w, seed=86)
        gx = CN.conv(gy, whole, n, h, w, transpose=True, epi=CN.EPI_MASK, mask=x)
        wt_t = wt.flip(2, 3).transpose(0, 1).contiguous()                    # the same backward-data as a forward layer
        for lo, hi in ((0, 32), (32, 64)):
            pc = CN.PackedConv(wt_t[lo:hi].contiguous(), torch.zeros(hi - lo, device='cuda'))
            gp = CN.conv(gy, pc, n, h, w, infer=True, epi=CN.EPI_MASK, mask=x[:, lo:hi].contiguous())
            assert torch.equal(gx[:, lo:hi], gp), 'backward-data cins %d..%d' % (lo, hi)


# ---------------------------------------------------------------------------------------------------
# randomized layer shapes through the dispatching wrappers (direct / F(4,3) / F(4,5) / split-precision / small-cout kernels)
import os
_FUZZ = int(os.environ.get('RISP_TEST_SEEDS', '8')) * 6            # soak runs: RISP_TEST_SEEDS=64


@pytest.mark.parametrize('seed', range(_FUZZ))
def test_random_layer_shapes(seed, monkeypatch):
    from reconfigisp_amd import convnets as CN
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f16x2' if seed % 2 == 0 else 'f32')      # every other case on the fp32 routes
    rng = np.random.default_rng(9000 + seed)
    k = int(rng.choice([1, 3, 3, 5, 5, 9]))
    cin, cout = int(rng.integers(1, 65)), int(rng.integers(1, 65))
    n, h = int(rng.integers(1, 4)), int(rng.integers(max(2, k // 2 + 1), 72))
    w = int(rng.integers(max(2, k // 2 + 1), 150))
    if rng.random() < 0.6:
        w = max(4, w // 4 * 4)                                       # the vector paths need W % 4 == 0
    wt, b = rnd(cout, cin, k, k, seed=seed) * (0.5 / k), rnd(cout, seed=seed + 1) * 0.1
    x = rnd(n, cin, h, w, seed=seed + 2)
    add, mask = rnd(n, cout, h, w, seed=seed + 3), rnd(n, cout, h, w, seed=seed + 4)
    pc = CN.PackedConv(wt, b)
    lin = TF.conv2d(x, wt, b, padding=k // 2)
    tag = 'k%d %d->%d N%d %dx%d' % (k, cin, cout, n, h, w)
    mode = int(rng.integers(0, 4))
    for infer in (False, True):
        if mode == 0:
            got, ref = CN.conv(x, pc, n, h, w, infer=infer), lin
        elif mode == 1:
            got, ref = CN.conv(x, pc, n, h, w, epi=CN.EPI_RELU, infer=infer), torch.relu(lin)
        elif mode == 2:
            got, ref = CN.conv(x, pc, n, h, w, epi=CN.EPI_ADD | CN.EPI_RELU, add=add, add_c=cout, infer=infer), torch.relu(lin + add)
        else:
            got, ref = CN.conv(x, pc, n, h, w, epi=CN.EPI_MASK, mask=mask, infer=infer), lin * (mask > 0)
        assert_close(got, ref, what='%s mode %d infer=%s' % (tag, mode, infer))
    gy = rnd(n, cout, h, w, seed=seed + 5)
    gref = TF.conv_transpose2d(gy, wt, padding=k // 2)
    addb, maskb = rnd(n, cin, h, w, seed=seed + 6), rnd(n, cin, h, w, seed=seed + 7)
    got = CN.conv(gy, pc, n, h, w, transpose=True, epi=CN.EPI_ADD | CN.EPI_MASK, add=addb, add_c=cin, mask=maskb)
    assert_close(got, (gref + addb) * (maskb > 0), what=tag + ' bwd-data add+mask')
    if cout <= 12 and k in (3, 5) or cout <= 4 and k == 9:
        assert_close(CN.conv_small(x, CN.SmallConv(wt, b), n, h, w, epi=CN.EPI_RELU), torch.relu(lin), what=tag + ' small')


@pytest.mark.parametrize('hw', [(8, 8), (16, 32), (20, 36), (40, 72), (256, 256)])
@pytest.mark.parametrize('cout', [64, 48])
def test_linear_k_9x9_over_three_channels(cout, hw, monkeypatch):
    """risp_conv2d_k3 (SRCNNRes' folded first layer, srcnn_res_arch.py:18): against PyTorch, with the border-case table
    and ReLU epilogue, ungrouped and as a grouped launch of three members on one shared input; and against the general
    kernel (another summation order: float tolerance).  (RISP_CONV_ARITH=f32: by default the 9x9 first layers run on
    risp_conv2d_toep_first - tests/test_gpu_toep.py.)"""
    from reconfigisp_amd import convnets as CN
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f32')
    import ctypes as C
    from reconfigisp_amd import lib as L
    from reconfigisp_amd.functional import _p, _stream
    h, w = hw
    n = 2
    packs, ws, bs = [], [], []
    for g in range(3):
        wt, b = rnd(cout, 3, 9, 9, seed=30 + g) * 0.1, rnd(cout, seed=40 + g) * 0.1
        pc = CN.PackedConv(wt, b)
        pc.k3 = CN.k3_weights(wt)
        packs.append(pc); ws.append(wt); bs.append(b)
    x = rnd(n, 3, h, w, seed=50)
    table = rnd(3 * n, cout, 9, 9, seed=51) * 0.1

    def launch(pc, xx, tab, epi, group=None):
        calls = []
        real = L.call
        L.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        try:
            y = CN.conv(xx, pc, n, h, w, epi=epi, cvals=tab, group=group)
        finally:
            L.call = real
        assert calls == ['risp_conv2d_k3']
        return y

    ref = [TF.conv2d(x, ws[g], bs[g], padding=4) for g in range(3)]
    assert_close(launch(packs[0], x, None, 0), ref[0], what='k3 plain')
    # border-case table + ReLU
    def case_of(v, L_):
        return torch.tensor([i if i < 4 else (8 - (L_ - 1 - i) if i >= L_ - 4 else 4) for i in range(L_)])
    cy, cx = case_of(0, h).cuda(), case_of(0, w).cuda()
    def with_table(g):
        t = table[g * n:(g + 1) * n]
        return torch.relu(ref[g] + t[:, :, cy][:, :, :, cx])
    y = launch(packs[1], x, table[n:2 * n].contiguous(), CN.EPI_RELU | CN.EPI_CASEBIAS)
    assert_close(y, with_table(1), what='k3 casebias + relu')
    # grouped: three members, one shared input
    stacked = CN.stack_packed(packs)
    yg = launch(stacked, x, table, CN.EPI_RELU | CN.EPI_CASEBIAS, group=(3, L.GROUP_SHARED_X))
    for g in range(3):
        single = launch(packs[g], x, table[g * n:(g + 1) * n].contiguous(), CN.EPI_RELU | CN.EPI_CASEBIAS)
        assert torch.equal(yg[g * n:(g + 1) * n], single)
        assert_close(single, with_table(g), what='k3 member %d' % g)
    # the general kernel on the same layer
    keep, packs[0].k3 = packs[0].k3, None               # (no pack, no route: route() falls through to risp_conv2d)
    try:
        general = CN.conv(x, packs[0], n, h, w)
    finally:
        packs[0].k3 = keep
    assert_close(launch(packs[0], x, None, 0), general, rtol=1e-5, floor=1.0, what='k3 vs general kernel')


@pytest.mark.parametrize('hw', [(8, 8), (16, 32), (20, 36), (40, 72), (128, 128)])
@pytest.mark.parametrize('k,cin,cout', [(3, 3, 64), (3, 4, 64), (9, 4, 64), (9, 4, 40), (3, 4, 24), (9, 3, 20)])
def test_linear_k_first_layers(k, cin, cout, hw, monkeypatch):
    """risp_conv2d_k3 on every first-layer form (path_14l_bgr_arch.py:40-43, path_14l_bayer_arch.py:37-40 + :70-75,
    srcnn_demosaic_arch.py:14-16 + :39-43, srcnn_res_arch.py:18): 3 plain channels or the 4 planes of the space-to-depth
    mosaic, bias + ReLU, against PyTorch; the launch really is the linear-k kernel (under RISP_CONV_ARITH=f32 for the 9x9 ones)."""
    from reconfigisp_amd import convnets as CN
    monkeypatch.setattr(CN, 'CONV_ARITH', 'f32')
    from reconfigisp_amd import lib as L
    h, w = hw
    n = 3
    wt, b = rnd(cout, cin, k, k, seed=60 + k + cin) * 0.1, rnd(cout, seed=61) * 0.1
    pc = CN.PackedConv(wt, b)
    assert pc.k3 is not None
    if cin == 4:
        bay = rnd(n, 1, 2 * h, 2 * w, seed=62)
        src, load, planes = bay, CN.LOAD_UNSHUFFLE2, TF.pixel_unshuffle(bay, 2)
    else:
        src = planes = rnd(n, 3, h, w, seed=63)
        load = CN.LOAD_PLAIN
    calls = []
    real = L.call
    L.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
    try:
        y = CN.conv(src, pc, n, h, w, load=load, epi=CN.EPI_RELU)
    finally:
        L.call = real
    assert calls == ['risp_conv2d_k3'], calls
    assert_close(y, torch.relu(TF.conv2d(planes, wt, b, padding=k // 2)), what='linear-k %dx%d %d->%d' % (k, k, cin, cout))


@pytest.mark.parametrize('arith', ['f16x2', 'f32'])
def test_conv_launches_what_route_says(arith, monkeypatch):
    """``conv`` / ``conv_small`` launch the entry point ``route`` / ``route_small`` name, for the layers of tests/test_route_table_cpu.py
    (the packs a layer holds are those ``pack_kinds`` lists), and every route meets PyTorch"""
    from reconfigisp_amd import convnets as CN, lib as L
    monkeypatch.setattr(CN, 'CONV_ARITH', arith)
    calls = []
    real = L.call
    monkeypatch.setattr(CN.L, 'call', lambda name, *a: (calls.append(name), real(name, *a))[1])
    n, h, w = 2, 24, 64
    for k, cin, cout in ((9, 3, 64), (5, 64, 32), (5, 32, 3), (1, 64, 32), (3, 64, 64), (3, 3, 64), (3, 64, 3), (5, 32, 12)):
        wt, b = rnd(cout, cin, k, k, seed=300 + k + cin) * (0.5 / k), rnd(cout, seed=301) * 0.1
        pc = CN.PackedConv(wt, b)
        for transpose in (False, True):
            assert CN._have(pc, transpose) == CN.pack_kinds(k, cin, cout, transpose)
            ci, co = (cout, cin) if transpose else (cin, cout)
            x = rnd(n, ci, h, w, seed=302)
            for infer in (False, True):
                if transpose and infer:
                    continue
                calls.clear()
                epi = 0 if transpose else CN.EPI_RELU
                y = CN.conv(x, pc, n, h, w, transpose=transpose, epi=epi, infer=infer)
                want = CN.route(k, ci, co, h, w, transpose, CN.LOAD_PLAIN, epi | (CN.EPI_NOBIAS if transpose else 0), 0, infer, True,
                                CN.pack_kinds(k, cin, cout, transpose))
                assert calls[-1] == want, (k, cin, cout, transpose, infer, calls, want)
                ref = TF.conv_transpose2d(x, wt, padding=k // 2) if transpose else torch.relu(TF.conv2d(x, wt, b, padding=k // 2))
                assert_close(y, ref, what='%s k%d %d->%d T%d' % (want, k, cin, cout, transpose))
        if cout <= 12:
            sc = CN.SmallConv(wt, b)
            x = rnd(n, cin, h, w, seed=303)
            for infer in (False, True):
                calls.clear()
                y = CN.conv_small(x, sc, n, h, w, infer=infer)
                want = CN.route_small(k, cin, cout, h, w, n, infer, False, CN.small_has_toep(k, cout), None, CN.small_has_tapout(k, cin, cout),
                                      CN.small_has_narrow3(k, cin, cout))
                assert calls[-1] in (want, want + '_split'), (k, cin, cout, infer, calls, want)
                assert_close(y, TF.conv2d(x, wt, b, padding=k // 2), what='%s k%d %d->%d' % (want, k, cin, cout))
