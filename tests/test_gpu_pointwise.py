"""GPU parity: element-wise HIP kernels (through the C ABI) vs the CPU oracle and the
golden vectors of the imported reference."""
import numpy as np
import pytest
import torch

import isp_oracle as O
from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu

T = lambda a: torch.from_numpy(np.asarray(a))


@pytest.fixture(scope='module')
def F():
    import reconfigisp_amd.functional as F
    from reconfigisp_amd import lib
    assert lib.load().risp_version() >= 100
    return F


def rnd(*shape, seed, lo=0.0, hi=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.random(shape) * (hi - lo) + lo).astype(np.float32))


def _fwd_bwd(fn_gpu, fn_cpu, x, p, gy, what, rtol=1e-4, floor=0.0):
    """Element-wise operators are held to north_star's bar ELEMENT-wise: |a - b| <= 1e-6 + 1e-4 |b| for the output and for the
    input gradient (``floor`` = 0: no relief for elements that are small beside the tensor's largest - that relief is for the
    14-layer CNNs).  Parameter gradients are sums over the image: they keep conftest's default floor."""
    xc, pc = x.clone().requires_grad_(True), (p.clone().requires_grad_(True) if p is not None else None)
    yc = fn_cpu(xc, pc) if p is not None else fn_cpu(xc)
    gc = torch.autograd.grad(yc, (xc, pc) if p is not None else (xc,), gy)
    xg = x.cuda().requires_grad_(True)
    pg = p.cuda().requires_grad_(True) if p is not None else None
    yg = fn_gpu(xg, pg) if p is not None else fn_gpu(xg)
    gg = torch.autograd.grad(yg, (xg, pg) if p is not None else (xg,), gy.cuda())
    assert_close(yg, yc, rtol=rtol, what=what + ' y', floor=floor)
    assert_close(gg[0], gc[0], rtol=rtol, what=what + ' gx', floor=floor)
    if p is not None:
        assert_close(gg[1], gc[1], rtol=rtol, what=what + ' gp')
    return yg


@pytest.mark.parametrize('shape', [(2, 3, 16, 16), (3, 3, 34, 50), (1, 3, 256, 256)])
def test_pointwise_vs_oracle(F, shape):
    n = shape[0]
    x = rnd(*shape, seed=1, lo=-0.15, hi=1.2)
    gy = rnd(*shape, seed=2, lo=-0.5, hi=0.5)
    _fwd_bwd(lambda a, p: F.wb_manual(a, p * 5), O.wb_manual, x, rnd(n, 3, seed=3), gy, 'wb_manual')
    _fwd_bwd(F.gamma, O.gamma_manual, x, rnd(n, 1, seed=4, lo=0.2, hi=0.9), gy, 'gamma')
    _fwd_bwd(F.gtm_manual, O.gtm_manual, x, rnd(n, 3, seed=5), gy, 'gtm')
    _fwd_bwd(F.wb_quadratic, O.wb_quadratic, x, rnd(n, 30, seed=6, lo=0.4, hi=0.6), gy, 'wbq')
    _fwd_bwd(F.grayworld, O.grayworld, rnd(*shape, seed=7), None, gy, 'grayworld')


@pytest.mark.parametrize('shape', [(512, 3, 64, 64), (200, 3, 64, 64), (300, 3, 40, 52), (64, 3, 96, 128)])
def test_wbq_backward_walks_of_several_vectors_per_thread(F, shape):
    """The quadratic white balance's backward runs ONE resident round of workgroups (risp_common.h: risp_bwd_blocks_wbq): with
    many images a thread walks several vectors, prefetched through LDS, and a ragged end follows (bgr_walk_lds).  Shapes: 4 whole iterations per thread, 1 iteration + a ragged one, a ragged row width, 2 iterations
    on larger planes (mixed with 1).  Against the float64 oracle: input gradient element by element, parameter sums as sums; twice: same bits."""
    n = shape[0]
    x = rnd(*shape, seed=11, lo=-0.15, hi=1.2)
    gy = rnd(*shape, seed=12, lo=-0.5, hi=0.5)
    # coefficients that keep every pre-activation inside (0.2, 0.8): no pixel sits at the clamp's edges, where fp32 and float64 gate
    # differently (the gates have their own tests above) - this one is about which pixels a thread visits and how the sums are added
    p = rnd(n, 30, seed=13, lo=0.497, hi=0.503)
    p[:, 9::10] = 0.55
    xc, pc = x.double().requires_grad_(True), p.double().requires_grad_(True)
    yc = O.wb_quadratic(xc, pc)
    assert 0.2 < float(yc.detach().min()) and float(yc.detach().max()) < 0.8
    gc = torch.autograd.grad(yc, (xc, pc), gy.double())
    outs = []
    for _ in range(2):
        xg, pg = x.cuda().requires_grad_(True), p.cuda().requires_grad_(True)
        outs.append(torch.autograd.grad(F.wb_quadratic(xg, pg), (xg, pg), gy.cuda()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert_close(outs[0][0], gc[0].float(), what='wbq gx', floor=0.0)
    assert_close(outs[0][1], gc[1].float(), what='wbq gp')


def test_wbq_gtm_vs_reference_golden(F):
    g = load_golden('pointwise')
    x, p = T(g['x']).cuda().requires_grad_(True), T(g['wbq_p']).cuda().requires_grad_(True)
    y = F.wb_quadratic(x, p)
    gx, gp = torch.autograd.grad(y, (x, p), T(g['gy']).cuda())
    assert_close(y, g['wbq_y'], what='wbq y', floor=0.0)                  # element-wise bar (tools_origin.py:317-359)
    assert_close(gx, g['wbq_gx'], what='wbq gx', floor=0.0)
    assert_close(gp, g['wbq_gp'], what='wbq gp')                          # 30 sums over the image: the default floor
    x, p = T(g['gtm_x']).cuda().requires_grad_(True), T(g['gtm_p']).cuda().requires_grad_(True)
    y = F.gtm_manual(x, p)
    gx, gp = torch.autograd.grad(y, (x, p), T(g['gy']).cuda())
    assert_close(y, g['gtm_y'], what='gtm y', floor=0.0)                  # (tools_origin.py:414-440)
    assert_close(gx, g['gtm_gx'], what='gtm gx', floor=0.0)
    assert_close(gp, g['gtm_gp'], what='gtm gp')
    # known answer of the reference's own smoke block (tools_origin.py:807-820)
    k = F.gtm_manual(torch.full((1, 3, 64, 64), 0.9).cuda(), torch.tensor([[0.3, 0.5, 0.7]]).cuda())
    assert abs(k.min().item() - 0.88) < 1e-6 and abs(k.max().item() - 0.88) < 1e-6


@pytest.mark.parametrize('hw', [(6, 8), (10, 6), (256, 256)])
def test_demosaic_nearest_bit_exact(F, hw):
    h, w = hw
    x = torch.arange(2 * h * w, dtype=torch.float32).view(2, 1, h, w)
    y = F.demosaic_nearest(x.cuda())
    assert torch.equal(y.cpu(), O.demosaic_nearest(x))                 # index map: bit exact
    xr = rnd(2, 1, h, w, seed=9).requires_grad_(True)
    gy = rnd(2, 3, h, w, seed=10)
    gc, = torch.autograd.grad(O.demosaic_nearest(xr), xr, gy)
    xg = xr.detach().cuda().requires_grad_(True)
    gg, = torch.autograd.grad(F.demosaic_nearest(xg), xg, gy.cuda())
    assert_close(gg, gc, what='demosaic gx', floor=0.0)


def test_gtm_segment_boundaries_and_passthrough(F):
    x = torch.tensor([-0.5, 0.0, 0.25, 0.5, 0.75, 1.0, 1.5, 0.2499999, 0.9999999] + [0.3] * 3).view(1, 3, 2, 2)
    p = torch.tensor([[0.1, 0.6, 0.7]])
    assert torch.equal(F.gtm_manual(x.cuda(), p.cuda()).cpu(), O.gtm_manual(x, p))


def test_chain_matches_per_op_and_oracle(F):
    n, h, w = 3, 32, 48
    bay = rnd(n, 1, h, w, seed=11)
    sig = lambda v: torch.sigmoid(torch.tensor(v)).repeat(n, 1)
    pw, pg, pt = sig(O.PARAM_INIT['wbmanual']), torch.full((n, 1), 0.45), rnd(1, 3, seed=12).repeat(n, 1)
    pq = rnd(n, 30, seed=13, lo=0.45, hi=0.55)
    ops = [F.OP_SKIP, F.OP_DEMOSAIC_NEAREST, F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL, F.OP_WB_QUADRATIC]
    # SKIP before the demosaic is resolved by the host; the chain itself starts at the demosaic
    outs = F.chain_forward(bay.cuda(), ops[1:], [None, (pw * 5).cuda(), pg.cuda(), pt.cuda(), pq.cuda()])
    ref = O.demosaic_nearest(bay)
    assert torch.equal(outs[0].cpu(), ref)
    for o, fn, p in zip(outs[1:], (O.wb_manual, O.gamma_manual, O.gtm_manual, O.wb_quadratic), (pw, pg, pt, pq)):
        ref = fn(ref, p)
        assert_close(o, ref, what=fn.__name__, floor=0.0)
    # odd quad count per row (W % 4 == 2) takes the float2 path
    bay2 = rnd(1, 1, 6, 10, seed=14)
    o2 = F.chain_forward(bay2.cuda(), [F.OP_DEMOSAIC_NEAREST, F.OP_GAMMA], [None, torch.full((1, 1), 0.5).cuda()])
    assert_close(o2[1], O.gamma_manual(O.demosaic_nearest(bay2), torch.full((1, 1), 0.5)), floor=0.0)
    # BGR-input chain with a leading skip aliasing its input
    xb = rnd(2, 3, 8, 8, seed=15).cuda()
    o3 = F.chain_forward(xb, [F.OP_SKIP, F.OP_WB_MANUAL], [None, (sig(O.PARAM_INIT['wbmanual'])[:2] * 5).cuda()])
    assert o3[0].data_ptr() == xb.data_ptr()
    assert_close(o3[1], O.wb_manual(xb.cpu(), sig(O.PARAM_INIT['wbmanual'])[:2]), floor=0.0)


def test_channel_stats_first_occurrence(F):
    x = rnd(2, 3, 16, 24, seed=16)
    x[0, 0, 3, 5] = x[0, 0, 9, 1] = -1.0          # tie on the min: first (row-major) wins
    x[1, 2, 0, 0] = x[1, 2, 15, 23] = 2.0
    stats, arg = F.channel_stats(x.cuda())
    flat = x.view(2, 3, -1)
    assert torch.equal(stats[..., 0].cpu(), flat.min(dim=2)[0])
    assert torch.equal(stats[..., 2].cpu(), flat.max(dim=2)[0])
    assert_close(stats[..., 1], flat.sum(dim=2), rtol=1e-5)
    assert arg[0, 0, 0].item() == 3 * 24 + 5 and arg[1, 2, 1].item() == 0


def test_mix_fwd_bwd(F):
    outs = [rnd(2, 3, 8, 8, seed=20 + k) for k in range(5)]
    w = torch.tensor([0.1, 0.0, 0.4, 0.3, 0.2])
    gy = rnd(2, 3, 8, 8, seed=30, lo=-1, hi=1)
    oc = [o.clone().requires_grad_(True) for o in outs]
    wc = w.clone().requires_grad_(True)
    yc = sum(o * wk for o, wk in zip(oc, wc))
    gc = torch.autograd.grad(yc, [wc] + oc, gy)
    og = [o.cuda().requires_grad_(k != 1) for k, o in enumerate(outs)]
    wg = w.cuda().requires_grad_(True)
    yg = F.mix(wg, og)
    gg = torch.autograd.grad(yg, [wg] + [o for o in og if o.requires_grad], gy.cuda())
    assert_close(yg, yc, floor=0.0)
    assert_close(gg[0], gc[0], what='gw')
    assert_close(gg[1], gc[1], what='go0')


def test_cpu_tensor_raises(F):
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        F.gamma(torch.rand(1, 3, 4, 4), torch.rand(1, 1))


@pytest.mark.parametrize('widths', [(12, 8, 1), (24, 8, 30), (30, 16, 7, 3), (96, 3)])
def test_conditional_head_mlp_forward_and_parameter_gradient(widths):
    """risp_cond_fc_fwd / _bwd (histogram -> MLP sliced from one flat vector -> + global scalar -> sigmoid) against
    the oracle's restatement of tools_origin.py:109-163, forward and d/d(flat)."""
    import reconfigisp_amd.functional as F
    import isp_oracle as O
    n = 3
    g = torch.Generator().manual_seed(len(widths) * 100 + widths[0])
    img = torch.rand(n, 3, 16, 20, generator=g)
    total = O.conditional_total_params(widths[:-1], widths[-1])
    flat = (torch.randn(total, generator=g) * 0.05)
    flat[-widths[-1]:] = torch.randn(widths[-1], generator=g)          # the "global" block: only its first entry counts
    gout = torch.randn(n, widths[-1], generator=g)
    ref_flat = flat.clone().requires_grad_(True)
    ref = O.conditional_fc(img, ref_flat, tuple(widths[:-1]), widths[-1])
    ref.backward(gout)
    dev_flat = flat.clone().cuda().requires_grad_(True)
    got = F.conditional_fc(img.cuda(), dev_flat, widths)
    got.backward(gout.cuda())
    assert_close(got.detach(), ref.detach(), what='head output')
    assert_close(dev_flat.grad, ref_flat.grad, rtol=2e-4, what='d/d flat')
    assert torch.count_nonzero(dev_flat.grad[-widths[-1] + 1:]) == 0 or widths[-1] == 1   # unused tail of the global block


def test_conditional_head_matches_reference_golden():
    import reconfigisp_amd.functional as F
    g = load_golden('conditional')
    x = torch.from_numpy(np.asarray(g['x'])).cuda()
    for tag, nout in (('gamma', 1), ('wbm', 3), ('wbq', 30)):
        inch = tuple(int(v) for v in g[tag + '_inch'])
        flat = torch.from_numpy(np.asarray(g[tag + '_flat'])).cuda()
        assert_close(F.conditional_fc(x, flat, inch + (nout,)), g[tag + '_fc'], what=tag)


import os
_FUZZ = int(os.environ.get('RISP_TEST_SEEDS', '8'))               # soak runs: RISP_TEST_SEEDS=64


@pytest.mark.parametrize('seed', range(_FUZZ))
def test_random_reductions(seed):
    """Random shapes / value ranges for the reduction kernels: channel statistics (ties included), histograms with
    torch.histc semantics (values outside [0,1] dropped, exactly 1.0 in the last bin), the slot mixture and its
    gradients, the truncating uint8 PSNR."""
    import reconfigisp_amd.functional as F
    from reconfigisp_amd.codes.utils import util
    rng = np.random.default_rng(800 + seed)
    n, c = int(rng.integers(1, 5)), 3
    h, w = int(rng.integers(2, 60)), 4 * int(rng.integers(1, 40))
    x = torch.from_numpy(rng.uniform(-0.3, 1.3, size=(n, c, h, w)).astype(np.float32))
    x.view(-1)[rng.integers(0, x.numel(), size=6)] = 1.0            # exact upper edge
    x.view(-1)[rng.integers(0, x.numel(), size=6)] = 0.0
    ties = x.view(n, c, -1)
    ties[0, 0, rng.integers(0, h * w, size=3)] = -0.5                # tied minimum: the first index wins
    stats, arg = F.channel_stats(x.cuda())
    flat = x.view(n, c, -1)
    assert torch.equal(stats[..., 0].cpu(), flat.min(dim=2)[0]) and torch.equal(stats[..., 2].cpu(), flat.max(dim=2)[0])
    assert_close(stats[..., 1], flat.sum(dim=2), rtol=1e-5, what='sum')
    first_min = (flat == flat.min(dim=2, keepdim=True)[0]).float().argmax(dim=2)
    first_max = (flat == flat.max(dim=2, keepdim=True)[0]).float().argmax(dim=2)
    assert torch.equal(arg[..., 0].cpu().long(), first_min) and torch.equal(arg[..., 1].cpu().long(), first_max)
    bins = int(rng.integers(2, 40))
    ref = torch.stack([torch.cat([torch.histc(ch, bins=bins, min=0, max=1) for ch in im]) for im in x])
    assert torch.equal(F.hist_features(x.cuda(), bins).cpu(), ref), 'histc bins=%d' % bins
    k = int(rng.integers(2, 9))
    outs = [torch.from_numpy(rng.random((n, c, h, w), dtype=np.float32)) for _ in range(k)]
    wts = torch.from_numpy(rng.random(k).astype(np.float32))
    wts[rng.integers(0, k)] = 0.0
    gy = torch.from_numpy(rng.uniform(-1, 1, size=(n, c, h, w)).astype(np.float32))
    oc, wc = [o.clone().requires_grad_(True) for o in outs], wts.clone().requires_grad_(True)
    gref = torch.autograd.grad(sum(o * wk for o, wk in zip(oc, wc)), [wc] + oc, gy)
    og, wg = [o.cuda().requires_grad_(True) for o in outs], wts.cuda().requires_grad_(True)
    yg = F.mix(wg, og)
    gg = torch.autograd.grad(yg, [wg] + og, gy.cuda())
    assert_close(yg, sum(o * wk for o, wk in zip(outs, wts)), what='mix', floor=0.0)
    for a, b in zip(gg, gref):
        assert_close(a, b, rtol=2e-4, what='mix grad')
    a, b = torch.from_numpy(rng.uniform(-0.1, 1.1, size=(n, 3, h, w)).astype(np.float32)), x
    ref_psnr = O.psnr_uint8(np.clip(a.numpy() * 255, 0, 255).astype(np.uint8), np.clip(b.numpy() * 255, 0, 255).astype(np.uint8))
    assert abs(util.psnr_tensors(a.cuda(), b.cuda()) - ref_psnr) < 1e-4


@pytest.mark.parametrize('seed', range(_FUZZ))
def test_random_pointwise_ops_forward_and_gradients(F, seed):
    """Random shapes, sample ranges and per-image parameters (full [0,1] range) for every differentiable element-wise
    op: output, input gradient and parameter gradient against the oracle's autograd."""
    rng = np.random.default_rng(900 + seed)
    n, h, w = int(rng.integers(1, 5)), 2 * int(rng.integers(2, 40)), 4 * int(rng.integers(1, 40))
    shape = (n, 3, h, w)
    lo, hi = (-0.3, 1.4) if rng.random() < 0.5 else (0.0, 1.0)
    x = rnd(*shape, seed=seed, lo=lo, hi=hi)
    gy = rnd(*shape, seed=seed + 1, lo=-1.0, hi=1.0)
    par = lambda k, a=0.02, b=0.98: torch.from_numpy(rng.uniform(a, b, size=(n, k)).astype(np.float32))
    _fwd_bwd(lambda a, p: F.wb_manual(a, p * 5), O.wb_manual, x, par(3), gy, 'wb_manual', rtol=2e-4)
    _fwd_bwd(F.gamma, O.gamma_manual, x, par(1, 0.05, 0.95), gy, 'gamma', rtol=2e-4)
    _fwd_bwd(F.gtm_manual, O.gtm_manual, x, par(3), gy, 'gtm', rtol=2e-4)
    _fwd_bwd(F.wb_quadratic, O.wb_quadratic, x, par(30, 0.3, 0.7), gy, 'wbq', rtol=2e-4)
    _fwd_bwd(F.grayworld, O.grayworld, rnd(*shape, seed=seed + 2, lo=0.01, hi=1.0), None, gy, 'grayworld', rtol=2e-4)
    bay = rnd(n, 1, h, w, seed=seed + 3)
    assert torch.equal(F.demosaic_nearest(bay.cuda()).cpu(), O.demosaic_nearest(bay))


@pytest.mark.parametrize('seed', range(6))
def test_prune_softmax_matches_the_reference_sequence(F, seed):
    """risp_prune_softmax_fwd/_bwd against the reference's own tensor operations
    (super_prune_fifteen_demos_four_bayer_two.py:185-193), values, prune mask and alpha-gradient."""
    g = torch.Generator().manual_seed(seed)
    k = [2, 4, 15, 15, 21, 64][seed]
    alpha = (torch.randn(k, generator=g) * (0.3 if seed % 2 else 2.0))
    unavailable = None
    if seed in (1, 3):
        unavailable = torch.zeros(k, dtype=torch.uint8)
        unavailable[k - 1] = 1
    thr = 0.2
    a_ref = alpha.clone().requires_grad_(True)
    am = a_ref if unavailable is None else a_ref.masked_fill(unavailable.bool(), float('-inf'))
    probs = torch.softmax(am, dim=0)
    below = probs.detach() < thr * probs.detach().max()
    post = probs.clone()
    post[below] = 0
    post = post / post.sum().detach()
    gpost = torch.randn(k, generator=g)
    ga_ref, = torch.autograd.grad(post, a_ref, gpost)
    a_gpu = alpha.cuda().requires_grad_(True)
    got = F.prune_softmax(a_gpu, thr, None if unavailable is None else unavailable.cuda())
    ga, = torch.autograd.grad(got, a_gpu, gpost.cuda())
    assert torch.equal(got.cpu() == 0, post.detach() == 0)                    # the prune mask: exact
    assert_close(got, post.detach(), atol=1e-7, what='post')
    assert_close(ga, ga_ref, atol=1e-7, what='d post / d alpha')
    if unavailable is not None:
        assert got[k - 1].item() == 0.0 and ga[k - 1].item() == 0.0


def test_param_blocks_match_sigmoid_repeat(F):
    g = torch.Generator().manual_seed(3)
    widths = [1, 3, 30, 2, 5, 3, 1, 1, 2, 3, 3, 30, 3, 5, 1, 2, 3, 1]              # 18 ops: two launches
    raws = [(torch.randn(w, generator=g) * 2).requires_grad_(True) for w in widths]
    n = 5
    ref = [torch.sigmoid(r).repeat(n, 1) for r in raws]
    gs = [torch.randn(n, w, generator=g) for w in widths]
    gs[4] = None                                                                    # an op whose block nothing used
    live = [(b, gb) for b, gb in zip(ref, gs) if gb is not None]
    gref = torch.autograd.grad([b for b, _ in live], raws, [gb for _, gb in live], allow_unused=True)
    raws_g = [r.detach().cuda().requires_grad_(True) for r in raws]
    got = F.param_blocks(raws_g, n)
    for a, b in zip(got, ref):
        assert a.shape == b.shape and a.is_contiguous()
        assert_close(a, b.detach(), atol=1e-7, what='block')
    live_g = [(b, gb.cuda()) for b, gb in zip(got, gs) if gb is not None]
    ggot = torch.autograd.grad([b for b, _ in live_g], raws_g, [gb for _, gb in live_g], allow_unused=True)
    for a, b in zip(ggot, gref):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0
        else:
            assert_close(a, b, atol=1e-7, what='d block / d raw')
