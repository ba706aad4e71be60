"""GPU: the same-geometry proxies of a super-net slot as ONE launch per layer (convnets.srcnn_res_group /
srcnn_demosaic_group, risp_conv_desc.group_n) - bit-identical to the members launched one by one, to the per-operator
path of round 2, and checked against the CPU oracle.  Reference: super_prune_fifteen_demos_four_bayer_two.py:35-52,
183-212; srcnn_res_arch.py:15-53; srcnn_demosaic_arch.py:14-55."""
import pytest
import torch

import isp_oracle as O
from conftest import assert_close

pytestmark = pytest.mark.gpu

# the SRCNNRes family of an sRGB slot, registry order (registry.PROXY_NETS): parameter channels per member
FAMILY_P = (2, 1, 2, 1, 3, 1, 3, 5)


def _family(seed0=40):
    from reconfigisp_amd.codes.models.modules import tools_proxy as TP
    mods = []
    for j, p in enumerate(FAMILY_P):
        m = TP.ProxyNet(p, None)
        m.load_state_dict(O.make_weights('srcnn_res', seed0 + j, p))
        mods.append(m.cuda())
    return mods


def _inputs(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((n, 3, h, w), generator=g).cuda()
    pvs = [torch.rand((n, p), generator=g).cuda() for p in FAMILY_P]
    gys = [torch.randn((n, 3, h, w), generator=g).cuda() * (0.5 + 0.1 * j) for j in range(len(FAMILY_P))]
    return x, pvs, gys


def _run_group(mods, x, pvs, gys, grouped):
    import reconfigisp_amd.functional as F
    from reconfigisp_amd import convnets as CN
    CN.GROUP_LAUNCH, CN.LAUNCHES = grouped, [0]
    try:
        xg = x.clone().requires_grad_(True)
        pg = [p.clone().requires_grad_(True) for p in pvs]
        outs = F.srcnn_res_group(xg, pg, mods, {})
        grads = torch.autograd.grad(outs, [xg] + pg, gys)
        return outs, grads, CN.LAUNCHES[0]
    finally:
        CN.GROUP_LAUNCH, CN.LAUNCHES = True, None


@pytest.mark.parametrize('shape', [(4, 256, 256), (32, 256, 256), (3, 40, 72), (2, 16, 24)])
def test_grouped_srcnn_res_equals_member_launches_and_per_op_path(shape):
    """batch 4 (a rank of the 8-GPU search) and batch 32 (BASELINE config 3) at 256 x 256, plus ragged shapes: one launch
    per layer == eight launches per layer, bit for bit.  Against the per-operator path of round 2: the same bits of every image
    tensor when both take the matrix-pipe kernels (batch 32: risp_conv2d_tapout's step tiles sit on one 4-row grid however the
    launch cuts its segments); the parameter gradients pass through per-work-item channel sums, which a launch of 32 images cuts
    finer than one of 256 - the same sums in another order, compared at float tolerance, like everything on a small grid, where
    the per-operator launches split their input channels over workgroups (risp_conv2d_small_split) and the grouped grid does not."""
    import reconfigisp_amd.functional as F
    n, h, w = shape
    mods = _family()
    x, pvs, gys = _inputs(n, h, w, seed=n * 1000 + h)
    outs_g, grads_g, launches_g = _run_group(mods, x, pvs, gys, True)
    outs_m, grads_m, launches_m = _run_group(mods, x, pvs, gys, False)
    for a, b in zip(list(outs_g) + list(grads_g), list(outs_m) + list(grads_m)):
        assert torch.equal(a, b)
    assert launches_g == 12 and launches_m == 6 + 6 * len(mods)
    # the per-operator path of round 2 (one autograd.Function per member)
    gx_sum = None
    for j, m in enumerate(mods):
        xo = x.clone().requires_grad_(True)
        po = pvs[j].clone().requires_grad_(True)
        y = F.srcnn_res(xo, po, m)
        gx, gp = torch.autograd.grad(y, (xo, po), gys[j])
        if n == 32:
            assert torch.equal(y, outs_g[j]), 'member %d forward' % j
            assert_close(gp, grads_g[1 + j], rtol=1e-5, floor=1.0, what='member %d parameter gradient' % j)
        else:
            assert_close(y, outs_g[j], rtol=1e-5, floor=1.0, what='member %d forward' % j)
            assert_close(gp, grads_g[1 + j], rtol=1e-5, floor=1.0, what='member %d parameter gradient' % j)
        gx_sum = gx if gx_sum is None else gx_sum + gx
    if n == 32:
        # (the statistics' gradients - mean plane: gconst / HW added to every pixel - pass through the same per-work-item channel sums)
        assert_close(gx_sum, grads_g[0], rtol=1e-6, floor=1.0, what='input gradient')
    else:
        assert_close(gx_sum, grads_g[0], rtol=1e-5, floor=1.0, what='input gradient')


def test_grouped_srcnn_res_against_the_oracle():
    n, h, w = 2, 24, 32
    mods = _family(seed0=70)
    x, pvs, gys = _inputs(n, h, w, seed=5)
    outs, grads, _ = _run_group(mods, x, pvs, gys, True)
    xc = x.cpu().requires_grad_(True)
    pc = [p.cpu().requires_grad_(True) for p in pvs]
    ref = [O.srcnn_res(xc, pc[j], {k: v.detach().cpu() for k, v in m.state_dict().items()}) for j, m in enumerate(mods)]
    rg = torch.autograd.grad(ref, [xc] + pc, [g.cpu() for g in gys])
    for j in range(len(mods)):
        assert_close(outs[j], ref[j], what='member %d y' % j)
        assert_close(grads[1 + j], rg[1 + j], rtol=2e-4, what='member %d gpv' % j)
    assert_close(grads[0], rg[0], rtol=2e-4, floor=1.0, what='gx')


@pytest.mark.parametrize('shape', [(4, 256, 256), (32, 256, 256), (2, 40, 72)])
def test_grouped_srcnn_demosaic_equals_member_launches_and_per_op_path(shape):
    import reconfigisp_amd.functional as F
    from reconfigisp_amd import convnets as CN
    from reconfigisp_amd.codes.models.modules import tools_proxy as TP
    n, h, w = shape
    mods = []
    for seed in (11, 12):
        m = TP.ProxyDemosaicNet(0, None)
        m.load_state_dict(O.make_weights('srcnn_demosaic', seed))
        mods.append(m.cuda())
    g = torch.Generator().manual_seed(n + h)
    x = torch.rand((n, 1, h, w), generator=g).cuda()
    gys = [torch.randn((n, 3, h, w), generator=g).cuda() for _ in mods]
    res = {}
    for grouped in (True, False):
        CN.GROUP_LAUNCH = grouped
        try:
            xg = x.clone().requires_grad_(True)
            outs = F.srcnn_demosaic_group(xg, mods, {})
            res[grouped] = (outs, torch.autograd.grad(outs, xg, gys)[0])
        finally:
            CN.GROUP_LAUNCH = True
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    assert torch.equal(res[True][1], res[False][1])
    gx_sum = None
    for j, m in enumerate(mods):
        xo = x.clone().requires_grad_(True)
        y = F.srcnn_demosaic(xo, m)
        gx, = torch.autograd.grad(y, xo, gys[j])
        assert_close(y, res[True][0][j], rtol=1e-5, floor=1.0, what='demosaic member %d vs per-op' % j)
        gx_sum = gx if gx_sum is None else gx_sum + gx
    # (small grids: the per-operator tail / first-layer backward split their input channels, the grouped grid may not)
    assert_close(gx_sum, res[True][1], rtol=1e-5, floor=1.0, what='demosaic group gx')
    xc = x.cpu().requires_grad_(True)
    ref = [O.srcnn_demosaic(xc, {k: v.detach().cpu() for k, v in m.state_dict().items()}) for m in mods]
    if h * w <= 64 * 128:
        rg, = torch.autograd.grad(ref, xc, [g_.cpu() for g_ in gys])
        for j in range(2):
            assert_close(res[True][0][j], ref[j], what='demosaic member %d' % j)
        assert_close(res[True][1], rg, rtol=2e-4, floor=1.0, what='demosaic gx vs oracle')


@pytest.mark.parametrize('batch', [4, 32])
def test_search_network_step_identical_with_and_without_grouped_launches(batch):
    """The whole super-net (n_step 2, BASELINE config 4's per-rank batch and config 3's batch): outputs, architecture
    gradients and parameter gradients of a forward + backward are the same bits whether the proxies of a slot run as one
    launch per layer or one by one; the grouped form needs less than half the launches."""
    from reconfigisp_amd import convnets as CN
    from reconfigisp_amd.codes.models.modules.super_prune_fifteen_demos_four_bayer_two import (
        SuperPruneFifteenDemosFourBayerTwo)
    torch.manual_seed(3)
    net = SuperPruneFifteenDemosFourBayerTwo(2, 0.2, None).cuda()
    with torch.no_grad():
        for a in net.alphas:
            a.copy_(torch.randn_like(a) * 0.3)
    size = 256 if batch == 4 else 64
    bay, gt = O.synthetic_raw(batch, size, size, seed=21)
    bay, gt = bay.cuda(), gt.cuda()
    res = {}
    for grouped in (True, False):
        CN.GROUP_LAUNCH = grouped
        try:
            out = net(bay)
            loss = torch.nn.functional.mse_loss(out, gt)
            wanted = [p for p in net.parameters_and_alpha if p.numel()]
            res[grouped] = (out.detach(), torch.autograd.grad(loss, wanted, allow_unused=True))
        finally:
            CN.GROUP_LAUNCH = True
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
