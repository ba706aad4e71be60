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
                                           (64, 64, 3, (33, 30)), (3, 64, 3, (8, 8))])
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
