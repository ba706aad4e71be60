"""Pin the CPU oracle (oracle/isp_oracle.py) to golden vectors produced by the
imported reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import torch

import isp_oracle as O
from conftest import assert_close, load_golden

T = lambda a: torch.from_numpy(np.asarray(a))


def test_gtm_manual_known_answer():
    # the reference's only executable smoke block: tools_origin.py:807-820 -> 0.88 0.88
    y = O.gtm_manual(torch.full((1, 3, 64, 64), 0.9), torch.tensor([[0.3, 0.5, 0.7]]))
    assert abs(y.min().item() - 0.88) < 1e-6 and abs(y.max().item() - 0.88) < 1e-6
    g = load_golden('pointwise')
    assert_close(np.array([y.min().item(), y.max().item()]), g['gtm_kat'])


def test_wb_quadratic_fwd_bwd():
    g = load_golden('pointwise')
    x, p = T(g['x']).requires_grad_(True), T(g['wbq_p']).requires_grad_(True)
    y = O.wb_quadratic(x, p)
    assert_close(y, g['wbq_y'], what='wbq y')
    gx, gp = torch.autograd.grad(y, (x, p), T(g['gy']))
    assert_close(gx, g['wbq_gx'], what='wbq gx')
    assert_close(gp, g['wbq_gp'], what='wbq gp')


def test_gtm_manual_fwd_bwd():
    g = load_golden('pointwise')
    x, p = T(g['gtm_x']).requires_grad_(True), T(g['gtm_p']).requires_grad_(True)
    y = O.gtm_manual(x, p)
    assert np.array_equal(y.detach().numpy(), g['gtm_y'])       # same op order -> bit exact
    gx, gp = torch.autograd.grad(y, (x, p), T(g['gy']))
    assert_close(gx, g['gtm_gx'], what='gtm gx')
    assert_close(gp, g['gtm_gp'], what='gtm gp')
    assert float(gp[1].abs().max()) == 0.0                         # params[0] only


def test_identity_at_shipped_inits():
    # at the shipped initial parameters every parametrised op ~ identity
    x = torch.rand(2, 3, 8, 8)
    sig = lambda v: torch.sigmoid(torch.tensor(v)).repeat(2, 1)
    assert (O.wb_quadratic(x, sig(O.PARAM_INIT['wbquadratic'])) - x).abs().max() < 2e-3
    assert (O.gtm_manual(x, sig(O.PARAM_INIT['gtmmanual'])) - x).abs().max() < 1e-4
    assert (O.wb_manual(x, sig(O.PARAM_INIT['wbmanual'])) - x).abs().max() < 6e-3


def test_conditional_heads():
    g = load_golden('conditional')
    x = T(g['x'])
    for tag, nout in (('gamma', 1), ('wbm', 3), ('wbq', 30)):
        inch = tuple(int(v) for v in g[tag + '_inch'])
        flat = T(g[tag + '_flat'])
        assert flat.numel() == O.conditional_total_params(inch, nout)
        assert_close(O.conditional_fc(x, flat, inch, nout), g[tag + '_fc'], what=tag)
    p = O.conditional_fc(x, T(g['wbq_flat']), tuple(int(v) for v in g['wbq_inch']), 30)
    assert_close(O.wb_quadratic(x, p), g['wbq_y'], what='cond wbq y')


def _cnn(tag, kind, fn, P=0):
    g = load_golden('cnn_' + tag)
    w = O.make_weights(kind, int(g['seed']), P)
    x = T(g['x']).requires_grad_(True)
    if P:
        pv = T(g['pv']).requires_grad_(True)
        y = fn(x, pv, w)
    else:
        y = fn(x, w)
    assert_close(y, g['y'], what=tag + ' y')
    if 'gx' in g:
        grads = torch.autograd.grad(y, (x, pv) if P else (x,), T(g['gy']))
        assert_close(grads[0], g['gx'], what=tag + ' gx')
        if P:
            assert_close(grads[1], g['gpv'], what=tag + ' gpv')


def test_srcnn_res():
    _cnn('srcnn_res_p2', 'srcnn_res', O.srcnn_res, 2)
    _cnn('srcnn_res_p5', 'srcnn_res', O.srcnn_res, 5)
    _cnn('srcnn_res_p3_36x70', 'srcnn_res', O.srcnn_res, 3)


def test_srcnn_demosaic():
    _cnn('srcnn_demosaic', 'srcnn_demosaic', O.srcnn_demosaic)


def test_path14l():
    _cnn('path14l_bayer', 'path14l_bayer', O.path14l_bayer)
    _cnn('path14l_bayer_40x72', 'path14l_bayer', O.path14l_bayer)
    _cnn('path14l_bgr', 'path14l_bgr', O.path14l_bgr)


def supernet_weights(names_per_slot, base):
    out = []
    for s, names in enumerate(names_per_slot):
        row = []
        for k, name in enumerate(names):
            kind, P = weight_kind(name)
            row.append(O.make_weights(kind, base + 100 * s + k, P) if kind else None)
        out.append(row)
    return out


def weight_kind(name):
    if name in O.PROXY_P:
        return 'srcnn_res', O.PROXY_P[name]
    return {'bilinear': ('srcnn_demosaic', 0), 'laplacian': ('srcnn_demosaic', 0),
            'path_bayer': ('path14l_bayer', 0), 'path_bgr': ('path14l_bgr', 0)}.get(name, (None, 0))


def test_supernet_forward_and_grads():
    g = load_golden('supernet_n2')
    slots = [O.NAMES_BAYER, O.NAMES_DEMOSAIC, O.NAMES_SRGB, O.NAMES_SRGB]
    wts = supernet_weights(slots, 1000)
    alphas = [T(g['p_alpha_bayer']), T(g['p_alpha_demosaic']), T(g['p_alpha_step1']), T(g['p_alpha_step2'])]
    alphas = [a.clone().requires_grad_(True) for a in alphas]
    params, leaves = [], {}
    for s, names in enumerate(slots):
        row = []
        for name in names:
            key = 'p_param_step%d_%s' % (s - 1, name)
            if s >= 2 and key in g:
                t = T(g[key]).clone().requires_grad_(True)
                leaves[key[2:]] = t
            else:
                t = torch.zeros(0)
            row.append(t)
        params.append(row)
    x = T(g['x'])
    pruned = []
    for s, names in enumerate(slots):
        x, npr = O.mixed_slot(x, names, params[s], alphas[s], wts[s], 0.2)
        pruned.append(npr)
        assert_close(x, g['mid%d' % s], what='slot %d' % s)
    assert pruned == list(g['pruned_paths'])                       # prune mask: bit exact
    assert pruned[2] >= 1
    assert_close(x, g['y'], what='supernet y')
    names_a = ['alpha_bayer', 'alpha_demosaic', 'alpha_step1', 'alpha_step2']
    keys = sorted(leaves)
    grads = torch.autograd.grad(x, alphas + [leaves[k] for k in keys], T(g['gy']), allow_unused=True)
    for n, gr in zip(names_a + keys, grads):
        ref = g['g_' + n]
        gr = torch.zeros_like(T(ref)) if gr is None else gr
        assert_close(gr, ref, rtol=2e-4, what='grad ' + n)
    assert len(g['state_keys']) == 4 + 12 * 2                      # alphas + param_step* only


def test_fixed_pipeline_origin():
    g = load_golden('fixed_origin')
    names = O.parse_architecture(str(g['arch']))
    assert names == ['path_bayer', 'skip', 'nearest', 'wbmanual', 'gamma', 'wbquadratic', 'grayworld', 'bm3d']
    raw = [torch.tensor(O.PARAM_INIT[n]) for n in names]
    wts = []
    for k, n in enumerate(names):
        kind, P = weight_kind(n)
        wts.append(O.make_weights(kind, 2000 + k, P) if kind else None)
    y, mids = O.fixed_pipeline(T(g['x']), names, raw, wts)
    for i, m in enumerate(mids):
        assert_close(m, g['mid%d' % i], what='stage %d %s' % (i, names[i]))
    assert_close(y, g['y'])
    assert list(g['state_keys']) == ['param_step%d_%s' % (i + 1, n) for i, n in enumerate(names)
                                     if len(O.PARAM_INIT[n])]


def test_tiling_and_metrics():
    g = load_golden('tiling')
    patches, pos, cnt = O.whole2patch(g['img'], (16, 20), (12, 14))
    assert np.array_equal(pos, g['positions'])                     # index map: bit exact
    assert np.array_equal(patches, g['patches'])
    assert np.array_equal(cnt, g['count_map'])
    assert np.array_equal(O.create_patch_mask((16, 20), (2, 3)), g['mask'])
    assert np.array_equal(O.patch2whole(g['processed'], pos, cnt, (12, 14)), g['whole'])
    assert O.tile_positions(3000, 512, 480) == [0, 480, 960, 1440, 1920, 2400, 2488]
    assert len(O.tile_positions(4000, 512, 480)) == 9
    m = load_golden('metrics')
    a, b = O.tensor2bgr_uint8(T(m['t'])), O.tensor2bgr_uint8(T(m['u']))
    assert np.array_equal(a, m['t_u8']) and np.array_equal(b, m['u_u8'])   # truncation, not rounding
    assert abs(O.psnr_uint8(a, b) - float(m['psnr'])) < 1e-9


def test_demosaic_nearest_index_map():
    # build-defined OPSPEC (parity unpinned): bit-exact index map on integer-coded data
    x = torch.arange(2 * 6 * 8, dtype=torch.float32).view(2, 1, 6, 8)
    y = O.demosaic_nearest(x)
    for n in range(2):
        for i in range(6):
            for j in range(8):
                qi, qj = i - i % 2, j - j % 2
                assert y[n, 2, i, j] == x[n, 0, qi, qj]            # R
                assert y[n, 0, i, j] == x[n, 0, qi + 1, qj + 1]    # B
                assert y[n, 1, i, j] == (x[n, 0, qi, qj + 1] if i % 2 == 0 else x[n, 0, qi + 1, qj])
