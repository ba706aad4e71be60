"""GPU: step-level reuse inside a DARTS iteration (SuperPrune...begin_reuse, DartsModel.optimize_alphas): forwards #1, #3
and #4 (darts_model.py:182-222, 270-324) see the same train batch and alphas, so the parameter-free CNN ops upstream of
the first shifted parameter are computed once.  Architecture gradients, validation loss and the state after the step are
the SAME BITS with the cache on and off; the cache removes launches."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_host_logic import darts_opt, seed_darts

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.asarray(a))


def _run(reuse, iters=2, alpha_grads=False):
    from reconfigisp_amd import convnets as CN
    from reconfigisp_amd.codes.models import create_model
    g = load_golden('darts_step')
    opt = darts_opt(torch.device('cuda'))
    opt['train']['step_reuse'] = reuse
    opt['train']['weight_step_alpha_grads'] = alpha_grads
    torch.manual_seed(0)
    model = create_model(opt)
    seed_darts(model)
    data = tuple(T(g[k]) for k in ('img', 'gt', 'val_img', 'val_gt'))
    out = []
    calls = [0]
    real = CN.L.call

    def counting(name, *a):
        calls[0] += 1
        return real(name, *a)

    CN.L.call = counting
    try:
        for it in range(iters):
            model.feed_data(data)
            model.update_learning_rate(it, warmup_iter=-1)
            model.optimize_alphas()
            out.append(model.val_loss.detach().clone())
            out += [a.grad.clone() for a in model.netG.alphas]
            model.optimize_parameters()
            out += [v.detach().clone() for v in model.netG.state_dict().values()]
    finally:
        CN.L.call = real
    return out, calls[0], model


@pytest.mark.filterwarnings('ignore:Detected call of')
def test_search_iteration_identical_with_and_without_step_reuse():
    on, calls_on, model = _run(True)
    off, calls_off, _ = _run(False)
    assert len(on) == len(off)
    for a, b in zip(on, off):
        assert torch.equal(a, b)
    # 2 of the 3 train-batch forwards skip Path-Restore-Bayer (14 launches), the proxy demosaics (3) and the Path-Restore op
    # of the first sRGB slot (14) in every iteration
    assert calls_off - calls_on >= 2 * 2 * (14 + 3 + 14)
    assert model.netG._reuse is None                                  # the scope is closed after the architecture step
    assert all('_risp_reuse' not in m.__dict__ for mods in model.netG.all_modules for m in mods)


@pytest.mark.filterwarnings('ignore:Detected call of')
def test_reuse_scope_misses_when_the_input_or_a_parameter_changes():
    """outside DartsModel: same input twice hits, an in-place change of the input or of a slot parameter misses"""
    from test_host_logic import build_supernet
    import isp_oracle as O
    net = build_supernet(2, torch.device('cuda'))
    bay, _ = O.synthetic_raw(2, 32, 32, seed=4)
    bay = bay.cuda()
    ref = net(bay).detach()
    net.begin_reuse()
    try:
        a = net(bay).detach()
        n_rec = len(net._reuse)
        b = net(bay).detach()
        assert len(net._reuse) == n_rec and torch.equal(a, ref) and torch.equal(b, ref)
        with torch.no_grad():
            net.param_step1_gamma.add_(0.25)                          # first sRGB slot: downstream Path-Restore must recompute
        c = net(bay).detach()
        assert len(net._reuse) > n_rec
        bay.mul_(0.5)                                                 # new input value, same storage
        d = net(bay).detach()
    finally:
        net.end_reuse()
    assert torch.equal(c, net(bay.mul(2.0)).detach()) and not torch.equal(c, ref)
    assert torch.equal(d, net(bay).detach())


@pytest.mark.filterwarnings('ignore:Detected call of')
def test_weight_step_without_alpha_gradients_changes_no_result():
    """optimize_parameters asks autograd for the module parameters' gradients only (the reference's l_pix.backward() also
    fills alpha.grad, which nothing reads: models/darts_model.py:173, 226): alphas, parameters, losses over two iterations are
    the same bits either way; with the reference's form the transient alpha.grad is written."""
    lean, calls_lean, m_lean = _run(True)
    full, calls_full, m_full = _run(True, alpha_grads=True)
    for a, b in zip(lean, full):
        assert torch.equal(a, b)
    assert calls_lean < calls_full
    # after the weight step: the lean form leaves alpha.grad at what optimize_alphas set, the reference's form adds d loss / d alpha
    ga, gb = m_lean.netG.alpha_step1.grad, m_full.netG.alpha_step1.grad
    assert ga is not None and gb is not None and not torch.equal(ga, gb)


@pytest.mark.filterwarnings('ignore:Detected call of')
def test_search_iteration_identical_on_one_and_two_slot_streams(monkeypatch):
    """the ops of a slot on two HIP streams (planes of >= 2^16 pixels by default; forced here on the golden's small planes) against one
    stream: same kernels, same order of every sum - the same bits (super_prune_fifteen_demos_four_bayer_two.py:_run_jobs)"""
    from reconfigisp_amd.codes.models.modules import super_prune_fifteen_demos_four_bayer_two as SP
    assert SP.SLOT_STREAMS == 2 and SP.SLOT_STREAMS_MIN_PIXELS == 1 << 16
    used = []
    real = SP.SuperPruneFifteenDemosFourBayerTwo._run_jobs

    def spy(self, jobs, n_out, x, args, xs=None):
        pixels = x.shape[0] * x.shape[2] * x.shape[3]
        used.append(x.is_cuda and len(jobs) >= SP.SLOT_STREAMS_MIN_JOBS and SP.SLOT_STREAMS_MIN_PIXELS <= pixels)
        return real(self, jobs, n_out, x, args, xs)

    monkeypatch.setattr(SP.SuperPruneFifteenDemosFourBayerTwo, '_run_jobs', spy)
    one, _, _ = _run(True, iters=1)
    assert used and not any(used)                                     # the golden's planes are below the threshold: one stream
    del used[:]
    monkeypatch.setattr(SP, 'SLOT_STREAMS_MIN_PIXELS', 0)
    monkeypatch.setattr(SP, 'SLOT_STREAMS_MIN_JOBS', 2)               # (grouped launches leave the golden's slots two or three jobs)
    two, _, _ = _run(True, iters=1)
    assert any(used)
    assert len(one) == len(two)
    for a, b in zip(one, two):
        assert torch.equal(a, b)
