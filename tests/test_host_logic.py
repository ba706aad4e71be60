"""Host logic of the product (registry, fixed pipelines, super-net combiner, DARTS step) against
golden vectors of the imported reference.

Runs twice: on CPU with the operator seam bound to the oracle (``-m "not gpu"``: checks the host
logic only) and on the GPU with the real HIP kernels (``-m gpu``: the end-to-end parity test)."""
import numpy as np
import pytest
import torch

import isp_oracle as O
from conftest import assert_close as _assert_close, load_golden, ref_rtol


def assert_close(a, b, **kw):
    """Multi-stage graphs: every stage re-amplifies the fp32 noise of the stages before it (gamma's
    slope is 32 below 1/1024, the CNNs are 3-14 layers deep), so stage outputs are judged norm-wise:
    |a-b| <= 1e-4 * max|ref| (floor=1.0).  Single-operator tests keep the element-wise bound."""
    kw.setdefault('floor', 1.0)
    _assert_close(a, b, **kw)


T = lambda a: torch.from_numpy(np.asarray(a))


@pytest.fixture(params=['cpu-oracle-seam', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request, monkeypatch):
    import reconfigisp_amd.functional as F
    if request.param == 'hip':
        return torch.device('cuda')
    from oracle_backend import OracleImpl
    monkeypatch.setattr(F, '_IMPL', OracleImpl)
    return torch.device('cpu')


def weight_kind(name):
    if name in O.PROXY_P:
        return 'srcnn_res', O.PROXY_P[name]
    return {'bilinear': ('srcnn_demosaic', 0), 'laplacian': ('srcnn_demosaic', 0),
            'path_bayer': ('path14l_bayer', 0), 'path_bgr': ('path14l_bgr', 0)}.get(name, (None, 0))


def seed_ops(ops, names, base):
    for k, (op, name) in enumerate(zip(ops, names)):
        kind, P = weight_kind(name)
        if kind and hasattr(op, 'load_state_dict') and len(op.state_dict()):
            op.load_state_dict(O.make_weights(kind, base + k, P))


def build_supernet(n_step, dev):
    from reconfigisp_amd.codes.models.modules.super_prune_fifteen_demos_four_bayer_two import \
        SuperPruneFifteenDemosFourBayerTwo
    net = SuperPruneFifteenDemosFourBayerTwo(n_step=n_step, threshold=0.2, module_path=None)
    for s, (mods, names) in enumerate(zip(net.all_modules, net.slot_names)):
        seed_ops(mods, names, 1000 + 100 * s)
    return net.to(dev)


def test_supernet_matches_reference(dev):
    g, g64 = load_golden('supernet_n2'), load_golden('supernet_n2_f64')
    net = build_supernet(2, dev)
    assert list(net.state_dict().keys()) == list(g['state_keys'])
    assert len(net.trainable_parameters) == int(g['n_trainable'])
    with torch.no_grad():
        for k, v in net.named_parameters():
            v.copy_(T(g['p_' + k]))
    y = net(T(g['x']).to(dev))
    assert net.pruned_paths == list(g['pruned_paths'])                       # prune mask: bit exact
    for i, m in enumerate(net.intermediate_results):
        assert_close(m, g['mid%d' % i], what='slot %d' % i)
    named = dict(net.named_parameters())
    keys = sorted(named)
    grads = torch.autograd.grad(y, [named[k] for k in keys], T(g['gy']).to(dev), allow_unused=True)
    for k, gr in zip(keys, grads):
        gr = torch.zeros_like(named[k]) if gr is None else gr
        assert_close(gr, g['g_' + k], rtol=ref_rtol(g, g64, 'g_' + k), what='grad ' + k)


def test_origin_universal_matches_reference(dev):
    from reconfigisp_amd.codes.models.modules.origin_universal import OriginUniversal
    g = load_golden('fixed_origin')
    net = OriginUniversal(module_path=None, architecture=str(g['arch']))
    seed_ops(net.all_modules, net.step_names, 2000)
    net = net.to(dev)
    assert list(net.state_dict().keys()) == list(g['state_keys'])
    x = T(g['x']).to(dev)
    y = net(x)                                                              # autograd (per-op) path
    for i, m in enumerate(net.intermediate_results):
        assert_close(m, g['mid%d' % i], what='stage %d' % i)
    assert_close(y, g['y'])
    assert net.intermediate_results[1] is net.intermediate_results[0]       # Skip returns the same tensor
    if dev.type == 'cuda':
        with torch.no_grad():
            y2 = net(x)                                                     # fused inference path
        for i, m in enumerate(net.intermediate_results):
            assert_close(m, g['mid%d' % i], what='fused stage %d' % i)
        assert_close(y2, g['y'])


def test_isp_universal_conditional_matches_reference(dev):
    from reconfigisp_amd.codes.models.modules.isp_universal import IspUniversal
    g = load_golden('fixed_isp')
    net = IspUniversal(module_path=None, indiv_module_paths=(None,) * 8, architecture=str(g['arch']),
                       gamma_in_channels=(12, 8), wb_manual_in_channels=(12, 8), wb_quadratic_in_channels=(24, 8))
    seed_ops(net.all_modules, net.step_names, 3000)
    assert list(net.state_dict().keys()) == list(g['state_keys'])
    net.load_state_dict({k: T(g['p_' + k]) for k in net.state_dict()})
    net = net.to(dev)
    x = T(g['x']).to(dev)
    y = net(x)
    for i, m in enumerate(net.intermediate_results):
        assert_close(m, g['mid%d' % i], what='stage %d %s' % (i, net.step_names[i]))
    assert_close(y, g['y'])
    if dev.type == 'cuda':
        with torch.no_grad():
            net(x)
        for i, m in enumerate(net.intermediate_results):
            assert_close(m, g['mid%d' % i], what='fused stage %d %s' % (i, net.step_names[i]))


def test_registry_errors():
    from reconfigisp_amd.codes.models.modules import registry as R
    from reconfigisp_amd.codes.models import networks, create_model
    with pytest.raises(ValueError, match='Domain'):
        R.parse_architecture('01_02', R.NAMES_SRGB)
    with pytest.raises(AssertionError):
        R.parse_architecture('sRGB_16', R.NAMES_SRGB)
    with pytest.raises(NotImplementedError):
        R.make_op('ten_layer_net', None)
    with pytest.raises(NotImplementedError, match='not recognized'):
        networks.define_G({'network_G': {'which_model_G': 'Nope'}})
    with pytest.raises(NotImplementedError, match='not recognized'):
        create_model({'model': 'nope'})
    with pytest.raises(NotImplementedError, match='outside the hot-path scope'):
        create_model({'model': 'darts_yolo'})
    assert [n for _, n in R.parse_architecture('Bayer_01_Demosaic_03_sRGB_01_13_11', R.NAMES_SRGB)] == \
        ['path_bayer', 'laplacian', 'gamma', 'wbquadratic', 'wbmanual']


def darts_opt(dev, n_step=2):
    from collections import OrderedDict
    return OrderedDict(model='darts', gpu_ids=[0] if dev.type == 'cuda' else None, dist=False, is_train=True,
                       network_G=dict(which_model_G='SuperPruneFifteenDemosFourBayerTwo', n_step=n_step, n_modules=15,
                                      prune_threshold=0.2, module_path=None),
                       path=dict(pretrain_model_G=None, strict_load=True),
                       train=dict(lr_G=1e-2, momentum_G=0.9, lr_meta=1e-2, beta1=0.9, beta2=0.99,
                                  pixel_criterion='l2', lr_scheme='MultiStepLR', lr_steps=[1000], restarts=None,
                                  restart_weights=None, lr_gamma=0.5, clear_state=False))


def seed_darts(model):
    for net in (model.netG, model.netV):
        for s, (mods, names) in enumerate(zip(net.all_modules, net.slot_names)):
            seed_ops(mods, names, 1000 + 100 * s)
        net.to(model.device)
        with torch.no_grad():
            net.alpha_demosaic[3] = -20.0


# The DARTS fixtures (tests/golden/make_golden.py): 'darts_step_kf' - the 4-slot scenario on a data seed where four arithmetics of the
# reference itself (oneDNN / native / one-thread fp32, float64) agree to 1e-5, so the 1e-4 bar judges the step logic and not a ReLU
# coin toss; 'darts_step_n3' - the reference's shipped search geometry (SID_search.yml:16-17,31-32: n_step 3, batch 4, 48 x 48, prune
# 0.2), where the reference's own fp32 run is 0.9e-4 from its float64 run (no tie-free seed exists at that size: ref_rtol adds the
# golden's own distance); 'darts_step' - the round-1 scenario, which holds one first-layer pre-activation at 3.6e-9 of its layer: kept
# as the regression of the fp32 first-layer route (RISP_CONV_TOEP_FIRST=0), whose rounding happens to take the reference's side of it.
# 'darts_step_kf5' (round 6): the FIVE-slot super-net (n_step 3, the slot count of the shipped search) at batch 2, 16 x 16, on the data seed
# where the reference's four arithmetics agree best (2.4e-5): every operator's gradient and state TENSOR BY TENSOR at the 1e-4 bar -
# the per-operator pin that the shipped 48 x 48 geometry ('darts_step_n3': judged slot by slot against float64) cannot give.  Judged
# on its FIRST iteration only (KF_ITERS): the second one is not tie-free for arithmetics outside the four it was selected on - the
# reference's own torch on another CPU (the GPU box's EPYC) moves it1_alpha_step1 by 3.3e-4, this build's kernels move all twelve
# operator gradients of slot step1 together by 1e-4 .. 4e-4 (one mask bit downstream of the slot).
KF_ITERS = {'darts_step_kf5': 1}
DARTS_FIXTURES = [('darts_step_kf', 2, None), ('darts_step_kf5', 3, None), ('darts_step_n3', 3, None), ('darts_step', 2, '0')]


@pytest.mark.filterwarnings('ignore:Detected call of')
@pytest.mark.parametrize('fixture,n_step,toep_first', DARTS_FIXTURES, ids=[f[0] for f in DARTS_FIXTURES])
def test_darts_search_step_matches_reference(dev, fixture, n_step, toep_first, monkeypatch):
    from reconfigisp_amd.codes.models import create_model
    from reconfigisp_amd import convnets as CN
    if toep_first is not None:
        monkeypatch.setattr(CN, 'TOEP_FIRST', toep_first)
    g, g64 = load_golden(fixture), load_golden(fixture + '_f64')
    if dev.type == 'cuda' and float(g.get('fp32_vs_f64', 0.0)) > 2e-5:
        # a scenario on which the reference's fp32 run is ~1e-4 from its own float64 run cannot hold a tensor-by-tensor fp32-vs-fp32 bar
        # for ANY second implementation; the kernels are judged on it against float64 (tests/test_gpu_error_budget.py), the step logic
        # here on the seam, where the oracle's arithmetic is torch's own
        pytest.skip('judged against the float64 run in test_gpu_error_budget.py::test_darts_iterations_within_budget')
    model = create_model(darts_opt(dev, n_step))
    seed_darts(model)
    data = tuple(T(g[k]) for k in ('img', 'gt', 'val_img', 'val_gt'))
    for it in range(KF_ITERS.get(fixture, 2)):
        model.feed_data(data)
        model.update_learning_rate(it, warmup_iter=-1)
        model.optimize_alphas()
        assert_close(model.val_loss, g['it%d_val_loss' % it], rtol=1e-4, what='val loss')
        for k, a in enumerate(model.netG.alphas):
            # fp32 vs the reference's fp32: the 1e-4 bar plus the golden's OWN distance from the float64 evaluation of the
            # reference (up to 7.1e-5 of the gradient's magnitude, tests/golden/darts_step_f64.npz) - conftest.ref_rtol
            key = 'it%d_alpha_grad%d' % (it, k)
            assert_close(a.grad, g[key], rtol=ref_rtol(g, g64, key), atol=1e-7, what='alpha grad %d' % k)
        model.optimize_parameters()
        assert abs(model.log_dict['loss'] - float(g['it%d_loss' % it])) <= 1e-4 * abs(float(g['it%d_loss' % it]))
        for k, v in model.netG.state_dict().items():
            # (the reference's fp32 state is up to 3.7e-4 off its own float64 evaluation on the near-zero parameters)
            key = 'it%d_%s' % (it, k)
            if k == 'alpha_demosaic':
                # DemosaicNet has no implementation in this build (the plugin's weights are not distributed): its
                # probability is exactly 0 and its logit receives NO gradient, where the reference's e^-20 softmax tail
                # hands Adam a 1e-8 gradient that it turns into a 5e-3 move.  A documented difference, not a tolerance:
                # the masked logit stays where it was, the three live ones are compared like everything else.
                assert v[3].item() == -20.0
                live = {key: g[key][:3]}, {key: g64[key][:3]}
                assert_close(v[:3], g[key][:3], rtol=ref_rtol(live[0], live[1], key), atol=1e-6, what=key)
                continue
            assert_close(v, g[key], rtol=ref_rtol(g, g64, key), atol=1e-6, what=key)


@pytest.mark.filterwarnings('ignore:Detected call of')
def test_darts_ft_finetunes_proxies_against_teachers(dev):
    """darts_ft (darts_ft_model.py:206-246): replay memory fills during the weight step, finetune_proxies() lowers
    the proxy-vs-teacher loss for the flagged proxies only and copies the weights into every sRGB slot."""
    import random
    from reconfigisp_amd.codes.models import create_model
    opt = darts_opt(dev)
    opt['model'] = 'darts_ft'
    opt['network_G']['which_model_G'] = 'SuperPruneFifteenDemosFourBayerTwoFt'
    opt['proxy_ft_params'] = dict(memory_size=3, ft_interval=1, ft_steps=3)
    opt['train']['lr_G'] = 1e-3
    model = create_model(opt)
    seed_darts(model)
    assert [n for n, _, _, _, _ in model.ft_nets] == ['crysisengine', 'whiteworld', 'bilateral', 'median', 'fastnlm']
    g = load_golden('darts_step')
    data = tuple(T(g[k]) for k in ('img', 'gt', 'val_img', 'val_gt'))
    model.feed_data(data)
    model.finetune_proxies()                                    # memory empty: warns, changes nothing
    model.optimize_parameters()
    model.optimize_parameters()
    assert len(model.ft_data) == 3 and all(t.shape[1] == 3 for t in model.ft_data)      # FIFO of size memory_size
    idx = [n for n, _ in model.netG.proxy_ft_flag].index('median')
    before = {k: v.clone() for k, v in model.netG.all_modules[-1][idx].state_dict().items()}
    frozen = {k: v.clone() for k, v in model.netG.all_modules[-1][1].state_dict().items()}     # reinhard: flagged off
    random.seed(3)
    torch.manual_seed(3)
    model.finetune_proxies()
    first = dict(model.log_dict)
    for _ in range(4):
        model.finetune_proxies()
    assert all(k.startswith('ft_loss_') or k == 'loss' for k in model.log_dict)
    assert model.log_dict['ft_loss_median'] < first['ft_loss_median']                  # the proxy learns its teacher
    after = model.netG.all_modules[-1][idx].state_dict()
    assert any((after[k] - before[k]).abs().max() > 0 for k in before)
    for k, v in model.netG.all_modules[-2][idx].state_dict().items():                  # copied into the other slot
        assert torch.equal(v, after[k])
    for k, v in model.netG.all_modules[-1][1].state_dict().items():
        assert torch.equal(v, frozen[k])


def isp_opt(dev, which, arch, criterion):
    from collections import OrderedDict
    return OrderedDict(model='isp', gpu_ids=[0] if dev.type == 'cuda' else None, dist=False, is_train=True,
                       network_G=dict(which_model_G=which, architecture=arch, individual_module_paths=[None] * 8,
                                      module_path=None),
                       path=dict(pretrain_model_G=None, strict_load=True),
                       train=dict(lr_G=1e-2, beta1=0.9, beta2=0.99, pixel_criterion=criterion, lr_scheme='MultiStepLR',
                                  lr_steps=[1000], restarts=None, restart_weights=None, lr_gamma=0.5, clear_state=False))


@pytest.mark.filterwarnings('ignore:Detected call of')
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_isp_model_training_step_matches_reference(dev, tag):
    """IspModel (models/isp_model.py:19-151): two optimize_parameters() - forward, MSE (case a) / L1 (case b),
    backward, Adam - and one test() against the imported reference (tests/golden/make_golden.py::gold_isp_model)."""
    from reconfigisp_amd.codes.models import create_model
    g = load_golden('isp_model')
    model = create_model(isp_opt(dev, str(g[tag + '_which']), str(g[tag + '_arch']), str(g[tag + '_criterion'])))
    seed_ops(model.netG.all_modules, model.netG.step_names, 4000)
    model.netG.to(model.device)
    assert list(model.netG.state_dict().keys()) == list(g[tag + '_state_keys'])
    data = (T(g[tag + '_img']), T(g[tag + '_gt']))
    for it in range(2):
        model.feed_data(data)
        model.update_learning_rate(it, warmup_iter=-1)
        model.optimize_parameters()
        ref_loss = float(g['%s_it%d_loss' % (tag, it)])
        assert abs(model.log_dict['loss'] - ref_loss) <= 1e-4 * abs(ref_loss)
        assert_close(model.output, g['%s_it%d_output' % (tag, it)], what='output it%d' % it)
        for k, v in model.netG.named_parameters():
            key = '%s_it%d_grad_%s' % (tag, it, k)
            if key in g:
                assert_close(v.grad, g[key], rtol=2e-4, atol=1e-7, what='it%d grad %s' % (it, k))
        for k, v in model.netG.state_dict().items():
            # Adam's first steps move every element by ~lr whatever the gradient's size: judged against lr = 1e-2
            assert_close(v, g['%s_it%d_%s' % (tag, it, k)], rtol=1e-4, atol=1e-5, what='it%d %s' % (it, k))
    y, mids = model.test()
    assert_close(y, g[tag + '_test_y'], what='test() output')
    for i, m in enumerate(mids):
        assert_close(m, g['%s_test_mid%d' % (tag, i)], what='test() stage %d' % i)


@pytest.mark.filterwarnings('ignore:Detected call of')
def test_virtual_net_evaluates_the_same_frozen_proxies(dev):
    """With module_path None both super-nets draw random proxies; the twin used for the virtual step must hold the
    searched net's operators (the reference gets that from loading the same weight files twice,
    darts_model.py:31-44).  lr_meta = 0: the virtual step copies parameters and alphas, so netV == netG exactly."""
    from reconfigisp_amd.codes.models import create_model
    opt = darts_opt(dev)
    opt['train']['lr_meta'] = 0.0
    torch.manual_seed(123)
    model = create_model(opt)                      # NOT re-seeded: constructor state
    with torch.no_grad():
        for net in (model.netG, model.netV):
            net.alpha_demosaic[3] = -20.0
    g = load_golden('darts_step')
    model.feed_data(tuple(T(g[k]) for k in ('img', 'gt', 'val_img', 'val_gt')))
    model.virtual_step()
    with torch.no_grad():
        a, b = model.netG(model.img), model.netV(model.img)
    assert torch.equal(a, b)
    for ma, mb in zip(model.netG.intermediate_results, model.netV.intermediate_results):
        assert torch.equal(ma, mb)


def test_mixture_weight_cache_follows_in_place_updates(dev):
    """The host copy of a slot's mixture weights is reused while alpha is unchanged (one device synchronisation per VALUE
    of alpha, not per forward) and refreshed by any in-place update - an optimiser step, copy_, load_state_dict."""
    net = build_supernet(1, dev)
    x = T(load_golden('supernet_n2')['x']).to(dev)
    with torch.no_grad():
        net(x)
        assert net.pruned_paths == [0, 1, 0]                         # alpha = 0: nothing pruned but the unavailable DemosaicNet
        net(x)
        assert net.pruned_paths == [0, 1, 0]
        net.alpha_step1[3] = 5.0                                     # in-place: one op now dominates its slot
        net(x)
        assert net.pruned_paths[2] == 14
        state = {k: v.clone() for k, v in net.state_dict().items()}
        state['alpha_step1'] = torch.zeros_like(state['alpha_step1'])
        net.load_state_dict(state)
        net(x)
        assert net.pruned_paths == [0, 1, 0]


# ---------------------------------------------------------------- local/global and latency losses (utils/util_loss.py:8-64)
def test_local_global_and_latency_losses_match_reference(dev):
    """value and image gradient of the restated losses against the imported reference: mixed / all-local / all-global
    flags, the gain clamp (darker than half the target; negative mean), the latency term"""
    from reconfigisp_amd.codes.utils.util_loss import latency_loss, local_global_loss
    g = load_golden('losses')
    mse = torch.nn.MSELoss()
    b = T(g['lg_gt']).to(dev)
    for tag in ('mixed', 'local', 'global'):
        a = T(g['lg_in']).to(dev).requires_grad_(True)
        loss = local_global_loss(a, b, T(g['lg_%s_flags' % tag]).to(dev), mse)
        grad, = torch.autograd.grad(loss, a)
        _assert_close(loss, g['lg_%s_loss' % tag], rtol=1e-5, what=tag + ' loss')
        _assert_close(grad, g['lg_%s_grad' % tag], rtol=1e-5, floor=1.0, what=tag + ' grad')
    if dev.type == 'cuda':
        # the same three cases through the criterion the models build (darts_model._criterion: PixelLoss('l2')): both branches on the
        # device in one call (risp_local_global_l2), no boolean indexing, no host read of the flags
        from reconfigisp_amd.codes.models.darts_model import PixelLoss
        from reconfigisp_amd import lib as L
        L.CALLS = {}
        try:
            for tag in ('mixed', 'local', 'global'):
                a = T(g['lg_in']).to(dev).requires_grad_(True)
                loss = local_global_loss(a, b, T(g['lg_%s_flags' % tag]).to(dev), PixelLoss('l2'))
                grad, = torch.autograd.grad(loss, a)
                _assert_close(loss, g['lg_%s_loss' % tag], rtol=1e-5, what=tag + ' loss (device)')
                _assert_close(grad, g['lg_%s_grad' % tag], rtol=1e-5, floor=1.0, what=tag + ' grad (device)')
            a = T(g['lg_clamp_in']).to(dev).requires_grad_(True)
            loss = local_global_loss(a, b[:2], torch.zeros(2, dtype=torch.int64, device=dev), PixelLoss('l2'))
            _assert_close(loss, g['lg_clamp_loss'], rtol=1e-5, what='clamped gain loss (device)')
            _assert_close(torch.autograd.grad(loss, a)[0], g['lg_clamp_grad'], rtol=1e-5, floor=1.0, what='clamped gain grad (device)')
            assert L.CALLS.get('risp_local_global_l2') == 4, L.CALLS
        finally:
            L.CALLS = None
    a = T(g['lg_clamp_in']).to(dev).requires_grad_(True)
    loss = local_global_loss(a, b[:2], torch.zeros(2, dtype=torch.int64, device=dev), mse)
    _assert_close(loss, g['lg_clamp_loss'], rtol=1e-5, what='clamped gain loss')
    _assert_close(torch.autograd.grad(loss, a)[0], g['lg_clamp_grad'], rtol=1e-5, floor=1.0, what='clamped gain grad')
    a = T(g['lg_in']).to(dev).requires_grad_(True)
    lat = T(g['lat_latency']).to(dev).requires_grad_(True)
    loss, term = latency_loss(a, b, lat, target_latency=2.5, w=0.07, fidelity_loss=mse)
    ga, gl = torch.autograd.grad(loss, (a, lat))
    _assert_close(loss, g['lat_loss'], rtol=1e-5, what='latency loss')
    _assert_close(term, g['lat_term'], rtol=1e-6, what='latency term')
    _assert_close(ga, g['lat_grad'], rtol=1e-5, floor=1.0, what='latency grad image')
    _assert_close(gl, g['lat_grad_latency'], rtol=1e-5, what='latency grad latency')


@pytest.mark.filterwarnings('ignore:Detected call of')
def test_darts_iteration_with_local_global_loss_and_six_tuple_batch(dev):
    """models/darts_model.py:131-133, 149-167: pixel_criterion 'local_global_l2', feed_data with the 6-tuple
    (img, gt, flag, val_img, val_gt, val_flag) - one search iteration against the reference"""
    from reconfigisp_amd.codes.models import create_model
    g = load_golden('losses')
    opt = darts_opt(dev)
    opt['train']['pixel_criterion'] = 'local_global_l2'
    model = create_model(opt)
    seed_darts(model)
    assert model.is_local_global and not model.is_latency
    model.feed_data(tuple(T(g[k]) for k in ('d_img', 'd_gt', 'd_flag', 'd_val_img', 'd_val_gt', 'd_val_flag')))
    assert model.glb_flag.tolist() == [0, 1] and model.val_glb_flag.tolist() == [1, 0]
    model.update_learning_rate(0, warmup_iter=-1)
    model.optimize_alphas()
    assert_close(model.val_loss, g['d_val_loss'], rtol=2e-4, what='val loss')
    for k, a in enumerate(model.netG.alphas):
        assert_close(a.grad, g['d_alpha_grad%d' % k], rtol=2e-4, atol=1e-7, what='alpha grad %d' % k)
    model.optimize_parameters()
    assert abs(model.log_dict['loss'] - float(g['d_loss'])) <= 2e-4 * abs(float(g['d_loss']))
    for k, v in model.netG.state_dict().items():
        if k == 'alpha_demosaic':                 # the masked DemosaicNet logit: see test_darts_search_step_matches_reference
            assert v[3].item() == -20.0
            v, ref = v[:3], g['d_' + k][:3]
        else:
            ref = g['d_' + k]
        # (no float64 evaluation of this case: the bar plus the 3.7e-4 the reference's own fp32 state is off in darts_step)
        assert_close(v, ref, rtol=5e-4, atol=1e-6, what=k)
    with pytest.raises(ValueError, match='Invalid data format'):
        model.feed_data((1, 2, 3))
