"""CPU checks of the weight algebra behind three kernels (pure tensor code in reconfigisp_amd/convnets.py; the
kernels themselves are tested on the GPU): the Winograd F(4,3) / F(4,5) packs, the small-cout packs, and the tables that
fold SRCNNRes' broadcast planes out of its 9x9 layer.  Each is emulated with plain torch ops in the exact way the
kernel consumes the pack and compared with torch.nn.functional.conv2d (fp64, so only the algebra is on trial)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from reconfigisp_amd import convnets as CN


def rnd(*shape, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(shape))


def wino43_emulate(x, pack, ck, cout):
    """What conv_wino43_kernel computes from its pack: F(4,3), four outputs per quad from 6 products per filter row."""
    n, cin, h, w = x.shape
    nq = (w + 3) // 4
    xp = TF.pad(x, (1, 4 * nq - w + 4, 1, 1))
    d = [xp[:, :, :, j:j + 4 * nq:4] for j in range(6)]                  # d_j = x[4q - 1 + j]
    s12, s34, m12, m34, m13, m24 = d[1] + d[2], d[3] + d[4], d[1] - d[2], d[3] - d[4], d[1] - d[3], d[2] - d[4]
    v = torch.stack([4 * d[0] - 5 * d[2] + d[4], 4 * s12 - s34, m34 - 4 * m12, -2 * m13 - m24, 2 * m13 - m24,
                     4 * d[1] - 5 * d[3] + d[5]], dim=-1)
    ncb, nch = pack.shape[0], pack.shape[1]
    u = pack.permute(0, 5, 1, 4, 2, 3).reshape(ncb * 32, nch * ck, 3, 6)[:cout, :cin]
    m = torch.zeros(n, cout, h, nq, 6, dtype=x.dtype)
    for ky in range(3):
        m += torch.einsum('nchqt,oct->nohqt', v[:, :, ky:ky + h], u[:, :, ky])
    a12, s12, a34, s34 = m[..., 1] + m[..., 2], m[..., 1] - m[..., 2], m[..., 3] + m[..., 4], m[..., 3] - m[..., 4]
    y = torch.stack([m[..., 0] + a12 + a34, s12 + 2 * s34, a12 + 4 * a34, s12 + 8 * s34 + m[..., 5]], dim=-1)
    return y.reshape(n, cout, h, 4 * nq)[..., :w]


@pytest.mark.parametrize('cin,cout,ck', [(64, 64, 4), (5, 3, 4), (12, 33, 2)])
def test_winograd43_pack_forward_and_backward_data(cin, cout, ck):
    wt = rnd(cout, cin, 3, 3, seed=21)
    x = rnd(2, cin, 6, 12, seed=22).requires_grad_(True)
    ref = TF.conv2d(x, wt, None, padding=1)
    got = wino43_emulate(x.detach(), CN.wino43_weights(wt, False, ck), ck, cout)
    assert torch.allclose(got, ref.detach(), rtol=1e-9, atol=1e-9)
    gy = rnd(2, cout, 6, 12, seed=23)
    gref, = torch.autograd.grad(ref, x, gy)
    got = wino43_emulate(gy, CN.wino43_weights(wt, True, ck), ck, cin)
    assert torch.allclose(got, gref, rtol=1e-9, atol=1e-9)


def small_emulate(x, pack, cout, k):
    w = pack[:, :k, :, :cout].permute(3, 0, 1, 2)             # back to (cout, cin, k, k)
    return TF.conv2d(x, w, None, padding=k // 2)


def small_emulate_c3(x, pack, k):
    """The four-lane form of a 3-cout layer (conv_small_kernel<.., C3>): per pair of output rows (2j, 2j + 1) and tile row r,
    lane 3 of the packed FMA pairs the third cout of row 2j (pack slot 2 of filter row r) with that of row 2j + 1 (slot 3 of
    filter row r = the third cout's weight of filter row r - 1)."""
    n, cin, h, w = x.shape
    p = k // 2
    xp = TF.pad(x, (p, p, p, p + 1))
    out = torch.zeros(n, 3, h + (h & 1), w, dtype=x.dtype)
    for y in range(0, h, 2):
        for r in range(k + 1):
            rowv = xp[:, :, y + r]                                          # tile row r of this row pair: (n, cin, w + 2p)
            for kx in range(k):
                a = rowv[:, :, kx:kx + w]
                if r < k:
                    out[:, :2, y] += torch.einsum('ncw,co->now', a, pack[:, r, kx, :2])
                if r >= 1:
                    out[:, :2, y + 1] += torch.einsum('ncw,co->now', a, pack[:, r - 1, kx, :2])
                out[:, 2, y] += torch.einsum('ncw,c->nw', a, pack[:, r, kx, 2])
                out[:, 2, y + 1] += torch.einsum('ncw,c->nw', a, pack[:, r, kx, 3])
    return out[:, :, :h]


@pytest.mark.parametrize('k', [3, 5, 9])
def test_small_cout_pack_forward_and_backward_data(k):
    wf = rnd(16, 7, k, k, seed=4)                               # forward layer 7 -> 16
    x = rnd(2, 7, 9, 11, seed=5).requires_grad_(True)
    ref = TF.conv2d(x, wf[:3], None, padding=k // 2)            # a 3-cout forward layer
    pack, cout = CN.small_weights(wf[:3].float())
    assert cout == 3 and pack.shape == (7, k + 1, k, 4)         # 3 couts: k + 1 filter rows, slot 3 = the third cout of the row above
    assert torch.equal(pack[:, 1:, :, 3], pack[:, :k, :, 2]) and pack[:, 0, :, 3].abs().max() == 0 and pack[:, k, :, :3].abs().max() == 0
    pack3 = CN.small_weights(wf[:3])[0]
    assert torch.allclose(small_emulate(x.detach(), pack3, 3, k), ref.detach(), rtol=1e-10, atol=1e-10)
    assert torch.allclose(small_emulate_c3(x.detach(), pack3, k), ref.detach(), rtol=1e-10, atol=1e-10)
    gy = rnd(2, 16, 9, 11, seed=6)
    gref, = torch.autograd.grad(TF.conv2d(x, wf, None, padding=k // 2), x, gy)
    pack, cout = CN.small_weights(wf, transpose=True, keep=3)   # backward-data restricted to 3 input channels
    assert torch.allclose(small_emulate(gy, pack, cout, k), gref[:, :3], rtol=1e-10, atol=1e-10)
    assert torch.allclose(small_emulate_c3(gy, pack, k), gref[:, :3], rtol=1e-10, atol=1e-10)
    pack4, cout4 = CN.small_weights(wf[:4])
    assert cout4 == 4 and pack4.shape == (7, k, k, 4)
    pack12, cout12 = CN.small_weights(wf[:12])
    assert cout12 == 12 and pack12.shape == (7, k, k, 12)
    with pytest.raises(ValueError):
        CN.small_weights(wf)                                    # 16 output channels


def border_case(v, L, p):
    return v if v < p else (2 * p - (L - 1 - v) if v >= L - p else p)


def rect_sums(g, k):
    """numpy restatement of risp_rect_sums: S[n,c,ky,kx] = sum of g over q with q + (ky-p, kx-p) inside."""
    n, c, h, w = g.shape
    p = k // 2
    out = torch.zeros(n, c, k, k, dtype=g.dtype)
    for ky in range(k):
        for kx in range(k):
            dy, dx = ky - p, kx - p
            out[:, :, ky, kx] = g[:, :, max(0, -dy):h - max(0, dy), max(0, -dx):w - max(0, dx)].sum(dim=(2, 3))
    return out


@pytest.mark.parametrize('P,hw', [(1, (8, 8)), (3, (9, 13)), (5, (12, 8))])
def test_srcnn_broadcast_planes_fold(P, hw):
    """conv over [image, constant planes] == conv over the image + table[case(y), case(x)], and the gradient of the
    constants == rect_sums(upstream) @ wconst."""
    h, w = hw
    n, cout, k = 2, 6, 9
    w1 = rnd(cout, 12 + P, k, k, seed=7)
    x = rnd(n, 3, h, w, seed=8)
    cv = rnd(n, 9 + P, seed=9).requires_grad_(True)
    full = TF.conv2d(torch.cat([x, cv[:, :, None, None].expand(n, 9 + P, h, w)], dim=1), w1, None, padding=k // 2)
    rcase, wconst = CN.srcnn_fold_tables(w1)
    table = (cv.detach() @ rcase).view(n, cout, k, k)
    iy = torch.tensor([border_case(y, h, k // 2) for y in range(h)])
    ix = torch.tensor([border_case(xx, w, k // 2) for xx in range(w)])
    folded = TF.conv2d(x, w1[:, :3], None, padding=k // 2) + table[:, :, iy][:, :, :, ix]
    assert torch.allclose(folded, full.detach(), rtol=1e-10, atol=1e-10)
    g = rnd(n, cout, h, w, seed=10)
    gref, = torch.autograd.grad(full, cv, g)
    got = rect_sums(g, k).reshape(n, -1) @ wconst
    assert torch.allclose(got, gref, rtol=1e-10, atol=1e-10)


def wino45_emulate(x, pack, ck, cout, layout=0):
    """What conv_wino45_glds_kernel computes from its pack: F(4,5), four outputs per quad from 8 products per filter row
    (the input transform and the output combination exactly as the kernel writes them)."""
    n, cin, h, w = x.shape
    nq = (w + 3) // 4
    xp = TF.pad(x, (2, 4 * nq - w + 6, 2, 2))
    d = [xp[:, :, :, j:j + 4 * nq:4] for j in range(8)]                  # d_j = x[4q - 2 + j]
    e1, o1 = 4 * (d[2] + d[6]) - 17 * d[4], 4 * (d[1] + d[5]) - 17 * d[3]
    e3, o3 = d[2] - 5 * d[4] + 4 * d[6], d[1] - 5 * d[3] + 4 * d[5]
    e5, o5 = 4 * d[2] - 5 * d[4] + d[6], 4 * d[1] - 5 * d[3] + d[5]
    v = torch.stack([(d[0] - d[6]) + 5.25 * (d[4] - d[2]), e1 + o1, e1 - o1, e3 + 2 * o3, e3 - 2 * o3, 2 * e5 + o5, 2 * e5 - o5,
                     (d[7] - d[1]) + 5.25 * (d[3] - d[5])], dim=-1)     # (n,ci,h+4,quads,8)
    ncb, nch = pack.shape[0], pack.shape[1]
    if layout == 0:                                 # [cb][chunk][ky][t][ci][32]
        u = pack.permute(0, 5, 1, 4, 2, 3).reshape(ncb * 32, nch * ck, 5, 8)[:cout, :cin]     # (co, ci, ky, t)
    else:                                           # [cb][chunk][ky][point group][cout block of 16][ci][cout 16][4 points]
        u = pack.permute(0, 4, 6, 1, 5, 2, 3, 7).reshape(ncb * 32, nch * ck, 5, 8)[:cout, :cin]
    m = torch.zeros(n, cout, h, nq, 8, dtype=x.dtype)
    for ky in range(5):
        m += torch.einsum('nchqt,oct->nohqt', v[:, :, ky:ky + h], u[:, :, ky])
    a12, s12, a34, s34, a56, s56 = (m[..., 1] + m[..., 2], m[..., 1] - m[..., 2], m[..., 3] + m[..., 4], m[..., 3] - m[..., 4],
                                    m[..., 5] + m[..., 6], m[..., 5] - m[..., 6])
    y = torch.stack([m[..., 0] + a12 + a34 + a56, s12 + 2 * s34 + 0.5 * s56, a12 + 4 * a34 + 0.25 * a56,
                     s12 + 8 * s34 + 0.125 * s56 + m[..., 7]], dim=-1)
    return y.reshape(n, cout, h, 4 * nq)[..., :w]


@pytest.mark.parametrize('cin,cout', [(64, 32), (32, 64), (8, 3), (4, 40)])
def test_winograd45_pack_forward_and_backward_data(cin, cout):
    wt = rnd(cout, cin, 5, 5, seed=21)
    x = rnd(2, cin, 6, 10, seed=22).requires_grad_(True)
    ref = TF.conv2d(x, wt, None, padding=2)
    for layout in (0, 1):                           # the one-row kernel's slab and the two-row kernel's
        got = wino45_emulate(x.detach(), CN.wino45_weights(wt, False, 4, layout), 4, cout, layout)
        assert torch.allclose(got, ref.detach(), rtol=1e-9, atol=1e-9)
        if cout % 4 == 0 or cout < 4:
            gy = rnd(2, cout, 6, 10, seed=23)
            gref, = torch.autograd.grad(ref, x, gy, retain_graph=True)
            got = wino45_emulate(gy, CN.wino45_weights(wt, True, 4, layout), 4, cin, layout)
            assert torch.allclose(got, gref, rtol=1e-9, atol=1e-9)


# --------------------------------------------------------------------------- split precision (risp_conv2d_f16x2)
def _split16(v, s):
    vs = v * s
    hi = vs.half()
    return hi, (vs - hi.float()).half()


def _pow2_scale(t):
    """s = 2^k with max|t| s in [2^14, 2^15) - the rule of convnets.f16x2_weights and of the kernel's per-tile scale"""
    _, e = torch.frexp(t.abs().max())
    return torch.ldexp(torch.ones(()), 15 - e)


def _emulate_f16x2(x, w, k):
    """x w ~ (x_lo w_hi + x_hi w_lo + x_hi w_hi) / (s_x s_w) with f16 halves; per tap and 16-channel block the products are
    summed exactly (they are exact in fp32; float64 here) and added to an fp32 accumulator - one rounding per matrix instruction,
    in the kernel's order (lo-hi, hi-lo, hi-hi)"""
    sx, sw = _pow2_scale(x), _pow2_scale(w)
    (xh, xl), (wh, wl) = _split16(x, sx), _split16(w, sw)
    n, c, h, ww = x.shape
    p = k // 2
    xp = [torch.nn.functional.pad(t.double(), (p, p, p, p)) for t in (xh, xl)]
    wp = [wh.double(), wl.double()]
    acc = torch.zeros((n, w.shape[0], h, ww), dtype=torch.float32)
    for c0 in range(0, c, 16):
        for ky in range(k):
            for kx in range(k):
                for i, j in ((1, 0), (0, 1), (0, 0)):
                    part = torch.einsum('nchw,oc->nohw', xp[i][:, c0:c0 + 16, ky:ky + h, kx:kx + ww], wp[j][:, c0:c0 + 16, ky, kx])
                    acc = (acc.double() + part).float()
    return acc / (sx * sw)


def _fma_chain(x, w, k):
    """what v_mfma_f32_32x32x2_f32 is: one fp32 rounding of the running sum per product"""
    n, c, h, ww = x.shape
    p = k // 2
    xp = torch.nn.functional.pad(x, (p, p, p, p))
    acc = torch.zeros((n, w.shape[0], h, ww), dtype=torch.float32)
    for c0 in range(c):
        for ky in range(k):
            for kx in range(k):
                acc = (acc.double() + xp[:, c0, ky:ky + h, kx:kx + ww].double()[:, None] * w[:, c0, ky, kx].double()[None, :, None, None]).float()
    return acc


@pytest.mark.parametrize('case', ['3x3 activations', '5x5 activations', '3x3 gradients'])
def test_f16x2_split_numerics(case):
    """The scheme of risp_conv_f16x2.hip against float64, beside the fp32 FMA chain the fp32 matrix instruction is: its rms error
    must not be larger (measured: about half - its error is per product, a random walk, where the chain rounds the running sum
    at every step), for activations in [0,1) and for sparse upstream gradients of magnitude 1e-5."""
    torch.manual_seed(0)
    k, cout = (5, 32) if case.startswith('5x5') else (3, 64)
    x = torch.rand(1, 64, 20, 20)
    if case.endswith('gradients'):
        x = torch.randn(1, 64, 20, 20) * 1e-5 * (torch.rand(1, 64, 20, 20) > 0.5)
    w = (torch.rand(cout, 64, k, k) - 0.5) * 2 / (64 * k * k) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=k // 2)
    m = ref.abs().max()
    rms = lambda y: ((y.double() - ref).pow(2).mean().sqrt() / m).item()
    e_split, e_chain = rms(_emulate_f16x2(x, w, k)), rms(_fma_chain(x, w, k))
    assert e_split <= e_chain, (e_split, e_chain)
    assert e_split < 1.2e-7


def test_f16x2_weight_pack_layout_and_scale():
    """convnets.f16x2_weights: header = 1 / s_w, body [chunk][tap][hi, lo][channel half][cout][8]; hi + lo reproduces w s_w to 2^-22;
    the transposed pack is the backward-data layer"""
    from reconfigisp_amd import convnets as CN
    torch.manual_seed(1)
    for co, ci in ((64, 64), (32, 64), (40, 32)):
        w = torch.randn(co, ci, 3, 3) * 0.05
        for tr in (False, True):
            p = CN.f16x2_weights(w, tr)
            wt = w.flip(2, 3).transpose(0, 1) if tr else w
            o, i = wt.shape[0], wt.shape[1]
            nt, nch = (o + 31) // 32, (i + 15) // 16
            assert p.dtype == torch.float16 and p.numel() == 8 + nch * 9 * 2 * 2 * nt * 32 * 8
            inv = p[:2].view(torch.float32).item()
            sw = 1.0 / inv
            assert 2.0 ** 14 <= wt.abs().max().item() * sw < 2.0 ** 15 and sw == 2.0 ** round(np.log2(sw))
            body = p[8:].view(nch, 9, 2, 2, nt * 32, 8).float()
            rec = ((body[:, :, 0] + body[:, :, 1]) * inv).permute(3, 0, 2, 4, 1).reshape(nt * 32, nch * 16, 3, 3)
            assert (rec[:o, :i] - wt).abs().max().item() <= 2.0 ** -22 * wt.abs().max().item()
            assert rec[o:].abs().max().item() == 0 if o < nt * 32 else True
            assert body[:, :, 0].abs().max().item() < 2.0 ** 15


# --------------------------------------------------------------------------- Toeplitz bands (risp_conv2d_toep)
@pytest.mark.parametrize('k,co,ci,tr', [(9, 3, 5, False), (5, 3, 6, False), (9, 3, 7, True), (9, 4, 3, True), (5, 1, 2, False), (5, 12, 4, False), (5, 7, 3, False)])
def test_toeplitz_band_pack_reproduces_the_convolution(k, co, ci, tr):
    """convnets.toep_weights: header = 1 / s_w, body [cin][ky][hi, lo][window half][row 8 cout + j][8]; with the window of block b
    starting 4 pixels left of it, band @ window == the filter row applied at the block's 8 pixels - the whole convolution (or the
    backward-data one of the transposed pack) comes out of 32 x 16 products per (ci, ky, block)"""
    from reconfigisp_amd import convnets as CN
    torch.manual_seed(k + co)
    w = torch.randn(co, ci, k, k) * 0.05 if not tr else torch.randn(ci, co + 2, k, k) * 0.05
    p = CN.toep_weights(w, tr, co if tr else None)
    rows = 32 if co <= 4 else 96
    assert p.dtype == torch.float16 and p.numel() == 8 + ci * k * 2 * 2 * rows * 8
    inv = p[:2].view(torch.float32).item()
    wt = w[:, :co].flip(2, 3).transpose(0, 1) if tr else w
    sw = 1.0 / inv
    assert 2.0 ** 14 <= wt.abs().max().item() * sw < 2.0 ** 15 and sw == 2.0 ** round(np.log2(sw))
    body = p[8:].view(ci, k, 2, 2, rows, 8).double()
    band = ((body[:, :, 0] + body[:, :, 1]) * inv).permute(0, 1, 3, 2, 4).reshape(ci, k, rows, 16)      # (ci, ky, m, u)
    assert 8 * co == rows or band[:, :, 8 * co:].abs().max().item() == 0                                      # rows of couts the layer has not
    h, wd, pad = 7, 24, k // 2
    x = torch.randn(1, ci, h, wd, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(x, wt.double(), padding=pad)[0]
    xp = torch.nn.functional.pad(x, (4, 12, pad, pad))[0]
    out = torch.zeros(co, h, wd, dtype=torch.float64)
    for y in range(h):
        for b in range(wd // 8):
            acc = sum(band[c, ky] @ xp[c, y + ky, 8 * b:8 * b + 16] for c in range(ci) for ky in range(k))
            out[:, y, 8 * b:8 * b + 8] = acc[:8 * co].view(co, 8)
    assert (out - ref).abs().max().item() <= 2.0 ** -21 * ref.abs().max().item()


# --------------------------------------------------------------------------- filter rows in the matrix rows (risp_conv2d_tapout)
@pytest.mark.parametrize('k,co,ci,tr', [(9, 3, 64, True), (5, 3, 32, False), (9, 2, 16, False), (5, 1, 48, True), (9, 3, 32, False)])
def test_tap_row_pack_reproduces_the_convolution(k, co, ci, tr):
    """convnets.tapout_weights: header = 1 / s_w, body [chunk of 16 cin][kx][hi, lo][channel half][row m][8 channels] with row
    m = 4 ky + co (ky < 8) or 4 co + 3 (ky = 8).  As the kernel consumes it: per INPUT row y' one matrix pass
    D[m][x] = sum_kx sum_ci W[m][ci][kx] in[ci][y'][x + kx - P], then out[co][y] = sum_ky D[m(co, ky)][y + ky - P] - the whole
    convolution (or the backward-data one of the transposed pack)."""
    torch.manual_seed(k + co + ci)
    w = torch.randn(co, ci, k, k) * 0.05 if not tr else torch.randn(ci, co + 2, k, k) * 0.05
    p = CN.tapout_weights(w, tr, co if tr else None)
    nch, pad = ci // 16, k // 2
    assert p.dtype == torch.float16 and p.numel() == 8 + nch * k * 2 * 2 * 32 * 8
    inv = p[:2].view(torch.float32).item()
    wt = w[:, :co].flip(2, 3).transpose(0, 1) if tr else w
    sw = 1.0 / inv
    assert 2.0 ** 14 <= wt.abs().max().item() * sw < 2.0 ** 15 and sw == 2.0 ** round(np.log2(sw))
    body = p[8:].view(nch, k, 2, 2, 32, 8).double()
    #     (chunk, kx, half, m, 8) -> W[m][ci = 16 chunk + 8 half + e][kx]
    W = ((body[:, :, 0] + body[:, :, 1]) * inv).permute(3, 0, 2, 4, 1).reshape(32, ci, k)
    row = lambda c, ky: 4 * ky + c if ky < 8 else 4 * c + 3
    used = sorted(row(c, ky) for c in range(co) for ky in range(k))
    assert len(set(used)) == co * k and max(used) < 32
    unused = [m for m in range(32) if m not in used]
    assert not unused or W[unused].abs().max().item() == 0
    h, wd = 11, 20
    x = torch.randn(2, ci, h, wd, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(x, wt.double(), padding=pad)
    D = torch.nn.functional.conv2d(x, W.view(32, ci, 1, k), padding=(0, pad))              # (n, 32, h, wd): one pass per input row
    out = torch.zeros(2, co, h, wd, dtype=torch.float64)
    for c in range(co):
        for ky in range(k):
            for y in range(h):
                if 0 <= y + ky - pad < h:
                    out[:, c, y] += D[:, row(c, ky), y + ky - pad]
    assert (out - ref).abs().max().item() <= 2.0 ** -21 * ref.abs().max().item()
    with pytest.raises(ValueError):
        CN.tapout_weights(torch.randn(4, 16, 9, 9))
    with pytest.raises(ValueError):
        CN.tapout_weights(torch.randn(3, 12, 5, 5))


@pytest.mark.parametrize('co,ci', [(64, 3), (40, 4), (7, 2)])
def test_first_layer_window_pack_reproduces_the_convolution(co, ci):
    """convnets.toep_first_weights: header = 1 / s_w, body [cout block][cin][ky][hi, lo][taps 0-7 | tap 8 + zeros][cout][8].  The kernel
    pads a filter row to (8 zeros, 9 taps, zeros) and takes the operand of pixel position j as slots 8 - j .. 15 - j of the 16 the
    lane holds (lanes of the upper window half hold the row from slot 8 on): with the window of block b starting 4 pixels left of
    it, sum_u A_j[u] window[u] is the filter row applied at pixel 8 b + j."""
    from reconfigisp_amd import convnets as CN
    torch.manual_seed(co)
    w = torch.randn(co, ci, 9, 9) * 0.05
    p = CN.toep_first_weights(w)
    nb = (co + 31) // 32
    assert p.dtype == torch.float16 and p.numel() == 8 + nb * ci * 9 * 2 * 2 * 32 * 8
    inv = p[:2].view(torch.float32).item()
    body = p[8:].view(nb, ci, 9, 2, 2, 32, 8).double()
    rows = ((body[:, :, :, 0] + body[:, :, :, 1]) * inv).permute(0, 4, 1, 2, 3, 5).reshape(nb * 32, ci, 9, 16)   # (cout, ci, ky, 16): taps 0-8, zeros
    assert (rows[:co, :, :, :9] - w.double()).abs().max().item() <= 2.0 ** -22 * w.abs().max().item()
    assert rows[:, :, :, 9:].abs().max().item() == 0 and (co == nb * 32 or rows[co:].abs().max().item() == 0)
    padded = torch.cat([torch.zeros(nb * 32, ci, 9, 8, dtype=torch.float64), rows, torch.zeros(nb * 32, ci, 9, 8, dtype=torch.float64)], dim=3)
    h, wd = 5, 16
    x = torch.randn(1, ci, h, wd, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(x, w.double(), padding=4)[0]
    xp = torch.nn.functional.pad(x, (4, 12, 4, 4))[0]
    out = torch.zeros(co, h, wd, dtype=torch.float64)
    for y in range(h):
        for b in range(wd // 8):
            for j in range(8):
                acc = torch.zeros(nb * 32, dtype=torch.float64)
                for c in range(ci):
                    for ky in range(9):
                        window = xp[c, y + ky, 8 * b:8 * b + 16]
                        a_j = torch.cat([padded[:, c, ky, 8 - j:16 - j], padded[:, c, ky, 16 - j:24 - j]], dim=1)     # lower / upper half lanes
                        acc += a_j @ window
                out[:, y, 8 * b + j] = acc[:co]
    assert (out - ref).abs().max().item() <= 2.0 ** -21 * ref.abs().max().item()


def test_f16x2_buffer_offsets_stay_inside_the_resource():
    """risp_conv2d_f16x2 range-checks per-lane byte offsets against 2^31 - 1 (risp_f16x2.h::h2_rsrc): the largest offset a lane forms
    - the last pixel of the last input plane a staging task reads, or of the last cout plane the epilogue stores / reads the residual
    and mask rows at - must stay below 2^31 for everything convnets.f16x2_addressable() lets through, and an untiled 3000 x 4000
    frame through a 64-channel layer (cout planes past 2 GiB: their stores would be dropped silently) must be refused."""
    def largest_offset(cin, cout, h, w):
        hw4 = 4 * h * w
        stage = 14 * hw4 + hw4 + (hw4 - 16)          # channel pair 7 of a chunk (2 cp planes), + 1 plane, + the pixel
        store = (cout - 1) * hw4 + (hw4 - 16)
        return max(stage, store)

    for cin, cout, h, w in ((64, 64, 256, 256), (64, 64, 1448, 1448), (64, 32, 2048, 2044), (16, 64, 2896, 2892), (64, 64, 2048, 4092)):
        ok = CN.f16x2_addressable(cin, cout, h, w)
        assert ok == (largest_offset(cin, cout, h, w) < (1 << 31) and max(cin, cout) * h * w * 4 < (1 << 31)), (cin, cout, h, w)
    assert CN.f16x2_addressable(64, 64, 2048, 4092)                   # 2^31 - 2^21 bytes: the largest tile class that fits
    assert not CN.f16x2_addressable(64, 64, 2048, 4096)               # exactly 2^31
    assert not CN.f16x2_addressable(64, 64, 3000, 4000)               # the frame of test_split.py, untiled
    assert not CN.f16x2_addressable(64, 32, 3000, 4000)


@pytest.mark.parametrize('co,ci,tr', [(32, 3, False), (64, 2, False), (32, 3, True), (64, 1, True)])
def test_thin_input_pack_reproduces_the_convolution(co, ci, tr):
    """convnets.thin5_weights: header = 1 / s_w, body [cout block][kx][hi, lo][half of the reduction index][cout][8] with the reduction
    index k = 3 ky + c.  As the kernel consumes it: out[co][y][x] = sum_kx sum_k W[co][kx][k] E[y][x + kx - 2][k], E[y][x'][3 ky + c] =
    in[c][y + ky - 2][x'] - the whole convolution (or the backward-data one of the transposed pack)."""
    torch.manual_seed(co + ci)
    w = torch.randn(co, ci, 5, 5) * 0.05 if not tr else torch.randn(ci, co, 5, 5) * 0.05
    p = CN.thin5_weights(w, tr)
    nb = co // 32
    assert p.dtype == torch.float16 and p.numel() == 8 + nb * 5 * 2 * 2 * 32 * 8
    inv = p[:2].view(torch.float32).item()
    wt = w.flip(2, 3).transpose(0, 1) if tr else w
    sw = 1.0 / inv
    assert 2.0 ** 14 <= wt.abs().max().item() * sw < 2.0 ** 15 and sw == 2.0 ** round(np.log2(sw))
    body = p[8:].view(nb, 5, 2, 2, 32, 8).double()
    #     (block, kx, half, m, 8) -> W[cout = 32 block + m][kx][k = 8 half + e]
    W = ((body[:, :, 0] + body[:, :, 1]) * inv).permute(0, 3, 1, 2, 4).reshape(co, 5, 16)
    assert W[:, :, 15].abs().max().item() == 0
    if ci < 3:
        assert W[:, :, :15].view(co, 5, 5, 3)[..., ci:].abs().max().item() == 0
    h, wd = 9, 14
    x = torch.randn(2, ci, h, wd, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(x, wt.double(), padding=2)
    xp = torch.nn.functional.pad(x, (2, 2, 2, 2))
    E = torch.zeros(2, h, wd + 4, 16, dtype=torch.float64)
    for ky in range(5):
        for c in range(ci):
            E[:, :, :, 3 * ky + c] = xp[:, c, ky:ky + h, :]
    out = torch.zeros(2, co, h, wd, dtype=torch.float64)
    for kx in range(5):
        out += torch.einsum('ok,nyxk->noyx', W[:, kx], E[:, :, kx:kx + wd])
    assert (out - ref).abs().max().item() <= 2.0 ** -21 * ref.abs().max().item()
    with pytest.raises(ValueError):
        CN.thin5_weights(torch.randn(32, 4, 5, 5))
    with pytest.raises(ValueError):
        CN.thin5_weights(torch.randn(16, 3, 5, 5))


@pytest.mark.parametrize('co,ci,tr', [(3, 64, False), (4, 16, False), (4, 64, True), (1, 32, True)])
def test_narrow_3x3_pack_reproduces_the_convolution(co, ci, tr):
    """convnets.narrow3_weights: header = 1 / s_w, body [chunk of 16 cin][kx][hi, lo][channel half][row m][8 channels] with row m = 4 ky + co.
    As the kernel consumes it: per INPUT row y' one matrix pass D[m][x] = sum_kx sum_ci W[m][ci][kx] in[ci][y'][x + kx - 1], then
    out[co][y] = (D[4 * 0 + co][y - 1] + D[4 * 1 + co][y]) + D[4 * 2 + co][y + 1]."""
    torch.manual_seed(co + ci)
    w = torch.randn(co, ci, 3, 3) * 0.05 if not tr else torch.randn(ci, co + 2, 3, 3) * 0.05
    p = CN.narrow3_weights(w, tr, co if tr else None)
    nch = ci // 16
    assert p.dtype == torch.float16 and p.numel() == 8 + nch * 3 * 2 * 2 * 32 * 8
    inv = p[:2].view(torch.float32).item()
    wt = w[:, :co].flip(2, 3).transpose(0, 1) if tr else w
    sw = 1.0 / inv
    assert 2.0 ** 14 <= wt.abs().max().item() * sw < 2.0 ** 15 and sw == 2.0 ** round(np.log2(sw))
    body = p[8:].view(nch, 3, 2, 2, 32, 8).double()
    #     (chunk, kx, half, m, 8) -> W[m][ci = 16 chunk + 8 half + e][kx]
    W = ((body[:, :, 0] + body[:, :, 1]) * inv).permute(3, 0, 2, 4, 1).reshape(32, ci, 3)
    used = sorted(4 * ky + c for c in range(co) for ky in range(3))
    unused = [m for m in range(32) if m not in used]
    assert W[unused].abs().max().item() == 0
    h, wd = 7, 12
    x = torch.randn(2, ci, h, wd, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(x, wt.double(), padding=1)
    D = torch.nn.functional.conv2d(x, W.view(32, ci, 1, 3), padding=(0, 1))               # (n, 32, h, wd): one pass per input row
    Dp = torch.nn.functional.pad(D, (0, 0, 1, 1))                                         # rows -1 and h are zero
    out = torch.stack([(Dp[:, c, 0:h] + Dp[:, 4 + c, 1:h + 1]) + Dp[:, 8 + c, 2:h + 2] for c in range(co)], 1)
    assert (out - ref).abs().max().item() <= 2.0 ** -21 * ref.abs().max().item()
    with pytest.raises(ValueError):
        CN.narrow3_weights(torch.randn(5, 64, 3, 3))
    with pytest.raises(ValueError):
        CN.narrow3_weights(torch.randn(3, 24, 3, 3))
