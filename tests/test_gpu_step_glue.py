"""GPU parity of the DARTS step glue (risp_step.hip): the pixel loss with its gradient against nn.MSELoss / nn.L1Loss, and the
list-wide kernels against the reference's per-parameter loops (models/darts_model.py:159-180, 204-222, 254-265, 299-323) - the
same arithmetic operation by operation, so the list kernels must give the SAME BITS as the torch loops."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(*shape, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(shape).astype(np.float32)).cuda()


@pytest.mark.parametrize('kind', ['l2', 'l1'])
@pytest.mark.parametrize('shape', [(4, 3, 256, 256), (1, 3, 8, 12), (2, 3, 62, 34)])
def test_pixel_loss_value_and_gradient(kind, shape):
    from reconfigisp_amd import functional as F
    y, gt = rnd(*shape, seed=1).requires_grad_(True), rnd(*shape, seed=2)
    ref_fn = torch.nn.functional.mse_loss if kind == 'l2' else torch.nn.functional.l1_loss
    ref = ref_fn(y.double(), gt.double())
    gref, = torch.autograd.grad(ref, y)
    loss = F.pixel_loss(y, gt, kind)
    assert loss.shape == () and abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item())
    for gout in (None, torch.tensor(0.37, device='cuda')):
        g, = torch.autograd.grad(loss, y, gout, retain_graph=True)
        want = gref * (1.0 if gout is None else 0.37)
        assert (g.double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()
    assert torch.equal(F.pixel_loss(y, gt, kind), loss)                 # fixed summation order


def test_pixel_loss_module_takes_what_the_kernel_does_not():
    from reconfigisp_amd.codes.models.darts_model import PixelLoss
    y, gt = rnd(1, 3, 5, 5, seed=3), rnd(1, 3, 5, 5, seed=4)            # 75 values: numel % 4 != 0 -> torch
    assert torch.allclose(PixelLoss('l2')(y, gt), torch.nn.functional.mse_loss(y, gt))
    v = rnd(2, 3, 16, 17, seed=5)[:, :, :, 1:]                          # a non-contiguous view
    assert torch.allclose(PixelLoss('l1')(v, torch.zeros_like(v)), v.abs().mean())


def _tensors(seed, sizes=(1, 3, 5, 15, 2, 64, 7)):
    return [rnd(n, seed=seed + i) for i, n in enumerate(sizes)]


def test_virtual_step_is_the_reference_loop_bit_for_bit():
    from reconfigisp_amd import functional as F
    p, g, buf = _tensors(10), _tensors(20), _tensors(30)
    vp = [torch.full_like(t, float('nan')) for t in p]
    g[2] = None                                                         # no gradient arrived: plain copy
    buf[4] = None                                                       # no momentum buffer yet
    mom, lr = 0.9, 1e-4
    before = [t._version for t in vp]
    F.darts_virtual_step(list(zip(vp, p, g, buf)), mom, lr)
    for k in range(len(p)):
        if g[k] is None:
            want = p[k]
        else:
            upd = (buf[k] * mom if buf[k] is not None else 0. * mom) + g[k]
            want = p[k] - upd * lr
        assert torch.equal(vp[k], want), k
    assert all(t._version > v for t, v in zip(vp, before))             # in-place through the C ABI, version counters bumped


def test_norm_eps_shifts_and_architecture_gradient():
    from reconfigisp_amd import functional as F
    dp = _tensors(40)
    ne = F.list_norm_eps(dp + [None])
    norm = torch.cat([t.reshape(-1) for t in dp]).double().norm()
    assert abs(ne[0].item() - norm.item()) <= 2e-6 * norm.item() and abs(ne[1].item() - 0.01 / norm.item()) <= 2e-6 * 0.01 / norm.item()
    assert F.list_norm_eps([torch.zeros(5, device='cuda')])[1].item() == 0.0           # norm < 1e-6 -> eps = 0 (:276-277)
    eps = ne[1:2]
    p = _tensors(50)
    q = [t.clone() for t in p]
    F.list_axpy_scalar(list(zip(q, dp)), eps, -2.)
    for a, b, d in zip(q, p, dp):
        assert torch.equal(a, b + d * (-2. * eps.reshape(())))
    # architecture gradient with a missing term and a NaN
    da, pos, neg = _tensors(60, (4, 2, 15, 8)), _tensors(70, (4, 2, 15, 8)), _tensors(80, (4, 2, 15, 8))
    pos[1] = None
    neg[3][2] = float('nan')
    out = [torch.full_like(t, 7.0) for t in da]
    flags = F.darts_alpha_grad(list(zip(out, da, pos, neg)), eps, 1e-4)
    assert flags.tolist() == [0, 0, 0, 1]
    for k in (0, 2):
        assert torch.equal(out[k], da[k] - 1e-4 * ((pos[k] - neg[k]) / 2. * eps.reshape(()))), k
    assert out[1].abs().max().item() == 0 and out[3].abs().max().item() == 0


def test_tables_longer_than_one_launch():
    """n_step >= 6 gives more parametrised operators than one risp_list_desc holds (64): the norm runs in pieces that carry the
    sum of squares, the architecture gradient in pieces with their own slice of the flags; empty tensors are skipped."""
    from reconfigisp_amd import functional as F, lib as L
    n = 2 * L.LIST_MAX + 9
    sizes = tuple(1 + (7 * i) % 30 for i in range(n))
    dp = _tensors(300, sizes)
    ne = F.list_norm_eps(dp[:5] + [None, torch.zeros(0, device='cuda')] + dp[5:])
    norm = torch.cat([t.reshape(-1) for t in dp]).double().norm()
    assert abs(ne[0].item() - norm.item()) <= 2e-6 * norm.item() and abs(ne[1].item() - 0.01 / norm.item()) <= 2e-6 * 0.01 / norm.item()
    one = F.list_norm_eps(dp[:L.LIST_MAX])
    assert torch.equal(one, F.list_norm_eps(dp[:L.LIST_MAX] + [None]))                 # a single piece: the round-4 bits
    eps = ne[1:2]
    asz = tuple(2 + i % 14 for i in range(n))
    da, pos, neg = _tensors(400, asz), _tensors(500, asz), _tensors(600, asz)
    pos[70] = None
    neg[L.LIST_MAX + 3][1] = float('nan')
    neg[n - 1][0] = float('nan')
    out = [torch.full_like(t, 7.0) for t in da]
    flags = F.darts_alpha_grad(list(zip(out, da, pos, neg)), eps, 1e-4).tolist()
    assert [k for k, f in enumerate(flags) if f] == [L.LIST_MAX + 3, n - 1]
    for k in range(n):
        if k in (70, L.LIST_MAX + 3, n - 1):
            assert out[k].abs().max().item() == 0, k
        else:
            assert torch.equal(out[k], da[k] - 1e-4 * ((pos[k] - neg[k]) / 2. * eps.reshape(()))), k


def test_tables_refuse_cpu_tensors():
    from reconfigisp_amd import functional as F
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        F.darts_virtual_step([(torch.zeros(3), torch.zeros(3), None, None)], 0.9, 1e-4)


def test_list_optimizers_follow_torch_optim():
    """ListSGD / ListAdam (codes/models/list_optim.py): torch.optim's state keys and arithmetic, step() as one launch"""
    from reconfigisp_amd.codes.models.list_optim import ListAdam, ListSGD
    sizes = (1, 3, 15, 2, 30, 7)
    for cls, ref_cls, kw in ((ListSGD, torch.optim.SGD, dict(lr=1e-2, momentum=0.9)),
                             (ListAdam, torch.optim.Adam, dict(lr=1e-3, betas=(0.9, 0.99))),
                             # torch's defaults: 1 - beta2 must be float(0.001), not 1.f - 0.999f (1.3e-5 apart)
                             (ListAdam, torch.optim.Adam, dict(lr=1e-3, betas=(0.9, 0.999)))):
        a = [torch.nn.Parameter(rnd(n, seed=100 + i)) for i, n in enumerate(sizes)]
        b = [torch.nn.Parameter(p.detach().clone()) for p in a]
        oa, ob = cls(a, **kw), ref_cls(b, **kw)
        for it in range(4):
            for k, (p, q) in enumerate(zip(a, b)):
                g = rnd(p.numel(), seed=1000 * it + k)
                p.grad, q.grad = (None, None) if (k == 3 and it == 1) else (g.clone(), g.clone())     # a parameter without a gradient
            va = [p._version for p in a]
            oa.step()
            ob.step()
            assert all(p._version > v for p, v, q in zip(a, va, b) if q.grad is not None)
            for p, q in zip(a, b):
                assert (p - q).abs().max().item() <= 1e-6 * max(q.abs().max().item(), 1e-3), (cls.__name__, it)
        sa, sb = oa.state_dict()['state'], ob.state_dict()['state']
        assert sa.keys() == sb.keys() and all(sa[k].keys() == sb[k].keys() for k in sa)
        for k in sa:
            for name in sa[k]:
                assert torch.allclose(sa[k][name].float().cpu(), sb[k][name].float().cpu(), rtol=2e-6, atol=1e-9), (cls.__name__, k, name)


def test_fan_out_sums_the_gradients_like_autograd():
    from reconfigisp_amd import functional as F
    x = rnd(2, 3, 16, 20, seed=200).requires_grad_(True)
    a, b, c = F.fan_out(x, 3)
    assert a.data_ptr() == x.data_ptr() and a is not b
    ws = [rnd(2, 3, 16, 20, seed=201 + i) for i in range(3)]
    g, = torch.autograd.grad((a * ws[0]).sum() + (b * ws[1]).sum() + (c * ws[2]).sum(), x)
    assert torch.equal(g, (ws[0] + ws[1]) + ws[2])
    a, b = F.fan_out(x, 2)
    g, = torch.autograd.grad((a * ws[0]).sum(), x)               # an alias nobody differentiates through
    assert torch.equal(g, ws[0])
    assert F.fan_out(x.detach(), 3)[0] is not None and all(t.data_ptr() == x.data_ptr() for t in F.fan_out(x.detach(), 3))
