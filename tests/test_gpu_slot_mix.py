"""GPU: the mixture of a super-net slot with its element-wise operators evaluated inside the kernel (risp_slot_mix_fwd /
_bwd, functional.slot_mix) against the unfused slot - every operator run as its own launch, then risp_mix - and against
the CPU oracle.  Reference: super_prune_fifteen_demos_four_bayer_two.py:183-212; tools_origin.py:27-45, 53-73, 205-225,
256-262, 317-359, 414-440."""
import numpy as np
import pytest
import torch

import isp_oracle as O
from conftest import assert_close

pytestmark = pytest.mark.gpu


def _entries(n, h, w, seed, with_wbq=True, n_tensors=3):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand((n, 3, h, w), generator=g) * 1.1 - 0.05).cuda()
    blocks = {'gamma': torch.rand((n, 1), generator=g) * 0.8 + 0.1, 'wb_manual': torch.rand((n, 3), generator=g) * 0.4 + 0.1,
              'gtm_manual': torch.tensor([[0.2, 0.55, 0.8]]).repeat(n, 1), 'wb_quadratic': torch.rand((n, 30), generator=g) * 0.1 + 0.45}
    tensors = [torch.rand((n, 3, h, w), generator=g).cuda() for _ in range(n_tensors)]
    order = ['gamma', 'T', 'grayworld', 'T', 'skip', 'wb_manual', 'T'] + (['wb_quadratic'] if with_wbq else []) + ['gtm_manual']
    order = [o for o in order if o != 'T' or tensors]
    return x, {k: v.cuda() for k, v in blocks.items()}, tensors, order


def _run(x, blocks, tensors, order, w, fused, gy):
    import reconfigisp_amd.functional as F
    xg = x.clone().requires_grad_(True)
    bg = {k: v.clone().requires_grad_(True) for k, v in blocks.items()}
    tg = [t.clone().requires_grad_(True) for t in tensors]
    wg = w.clone().requires_grad_(True)
    it = iter(tg)
    if fused:
        entries = [('tensor', next(it)) if o == 'T' else ('op', o, bg.get(o)) for o in order]
        y = F.slot_mix(wg, xg, entries, w_host=w.tolist())
    else:
        ops = {'skip': lambda: xg, 'gamma': lambda: F.gamma(xg, bg['gamma']), 'wb_manual': lambda: F.wb_manual(xg, bg['wb_manual'] * 5),
               'gtm_manual': lambda: F.gtm_manual(xg, bg['gtm_manual']), 'wb_quadratic': lambda: F.wb_quadratic(xg, bg['wb_quadratic']),
               'grayworld': lambda: F.grayworld(xg)}
        y = F.mix(wg, [next(it) if o == 'T' else ops[o]() for o in order], w_host=w.tolist())
    live = [k for k in ('gamma', 'wb_manual', 'gtm_manual', 'wb_quadratic') if k in order]
    grads = torch.autograd.grad(y, [wg, xg] + tg + [bg[k] for k in live], gy)
    return y.detach(), grads, live


@pytest.mark.parametrize('shape,with_wbq', [((2, 16, 24), True), ((3, 40, 72), True), ((4, 256, 256), True), ((4, 256, 256), False),
                                            ((2, 64, 64), False), ((160, 64, 64), True), ((40, 128, 128), True)])
def test_fused_slot_mixture_equals_unfused(shape, with_wbq):
    n, h, w_ = shape
    x, blocks, tensors, order = _entries(n, h, w_, seed=h + n, with_wbq=with_wbq)
    w = torch.softmax(torch.linspace(-1, 1, len(order)), 0).cuda()
    gy = torch.randn_like(x)
    y_f, g_f, live = _run(x, blocks, tensors, order, w, True, gy)
    y_u, g_u, _ = _run(x, blocks, tensors, order, w, False, gy)
    assert torch.equal(y_f, y_u)                                          # same per-operand bits, same summation order
    nt = len(tensors)
    for a, b in zip(g_f[2:2 + nt], g_u[2:2 + nt]):                         # w_k * gy
        assert torch.equal(a, b)
    for name, a, b in zip(live, g_f[2 + nt:], g_u[2 + nt:]):               # parameter gradients: same block partition, same order
        if name == 'wb_quadratic':
            # round 6: the 30 sums ride in the slot's ONE backward launch, on the partition of the other operators; the stand-alone
            # backward keeps its own single round of workgroups (risp_bwd_blocks_wbq): the same sums cut differently
            assert_close(a, b, rtol=1e-6, floor=1.0, what='grad wb_quadratic')
        else:
            assert torch.equal(a, b), name
    assert_close(g_f[0], g_u[0], rtol=1e-5, floor=1.0, what='architecture terms <gy, o_k>')      # another summation order
    assert_close(g_f[1], g_u[1], rtol=1e-5, floor=1.0, what='input gradient')                    # operands added in kind order


def test_fused_slot_mixture_without_tensor_operands_and_against_oracle():
    n, h, w_ = 2, 24, 32
    x, blocks, tensors, order = _entries(n, h, w_, seed=9, n_tensors=0)
    w = torch.softmax(torch.linspace(0.5, -0.5, len(order)), 0).cuda()
    gy = torch.randn_like(x)
    y, grads, live = _run(x, blocks, [], order, w, True, gy)
    xc = x.cpu().requires_grad_(True)
    bc = {k: v.cpu().requires_grad_(True) for k, v in blocks.items()}
    wc = w.cpu().requires_grad_(True)
    ops = {'skip': lambda: xc, 'gamma': lambda: O.gamma_manual(xc, bc['gamma']), 'wb_manual': lambda: O.wb_manual(xc, bc['wb_manual']),
           'gtm_manual': lambda: O.gtm_manual(xc, bc['gtm_manual']), 'wb_quadratic': lambda: O.wb_quadratic(xc, bc['wb_quadratic']),
           'grayworld': lambda: O.grayworld(xc)}
    ref = sum(ops[o]() * wc[i] for i, o in enumerate(order))
    rg = torch.autograd.grad(ref, [wc, xc] + [bc[k] for k in live], gy.cpu())
    assert_close(y, ref, what='y')
    assert_close(grads[0], rg[0], rtol=2e-4, floor=1.0, what='gw')
    assert_close(grads[1], rg[1], rtol=2e-4, floor=1.0, what='gx')
    for name, a, b in zip(live, grads[2:], rg[2:]):
        assert_close(a, b, rtol=2e-4, floor=1.0, what='grad ' + name)


@pytest.mark.parametrize('batch', [4])
def test_search_network_with_and_without_the_fused_slot(batch):
    from reconfigisp_amd.codes.models.modules import super_prune_fifteen_demos_four_bayer_two as SP
    torch.manual_seed(3)
    net = SP.SuperPruneFifteenDemosFourBayerTwo(2, 0.2, None).cuda()
    with torch.no_grad():
        for a in net.alphas:
            a.copy_(torch.randn_like(a) * 0.3)
    bay, gt = O.synthetic_raw(batch, 64, 64, seed=21)
    bay, gt = bay.cuda(), gt.cuda()
    res = {}
    for fused in (True, False):
        SP.FUSE_SLOT = fused
        try:
            out = net(bay)
            loss = torch.nn.functional.mse_loss(out, gt)
            wanted = [p for p in net.parameters_and_alpha if p.numel()]
            res[fused] = (out.detach(), torch.autograd.grad(loss, wanted, allow_unused=True))
        finally:
            SP.FUSE_SLOT = True
    assert torch.equal(res[True][0], res[False][0])                       # the forward pass is the same bits
    for a, b in zip(res[True][1], res[False][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert_close(a, b, rtol=2e-5, floor=1.0, what='gradient')
