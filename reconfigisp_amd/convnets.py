"""The four learned-proxy CNN families as sequences of fused MFMA convolution launches.

One ``torch.autograd.Function`` per network: forward = the chain of ``risp_conv2d``
launches (bias / ReLU / residual / space-to-depth / PixelShuffle / broadcast planes folded
into each launch), backward = the transposed chain (backward-data only: the proxies'
weights are frozen on the hot path - super_prune_fifteen_demos_four_bayer_two.py:50-52
keeps them out of every optimizer).  References:
  SRCNNRes       models/modules/srcnn_res_arch.py:26-53
  SRCNNDemosaic  models/modules/srcnn_demosaic_arch.py:27-55
  Path14lBayer   models/modules/path_14l_bayer_arch.py:59-88 (+ ResidualBlock :6-21)
  Path14lBgr     models/modules/path_14l_bgr_arch.py:58-86
"""
import ctypes as C
import os

import torch

from . import lib as L
from .functional import _dev, _p, _stream, channel_stats

LOAD_PLAIN, LOAD_UNSHUFFLE2, LOAD_CONSTCH = 0, 1, 2
EPI_RELU, EPI_ADD, EPI_MASK, EPI_SHUFFLE2, EPI_NOBIAS, EPI_CASEBIAS = 1, 2, 4, 8, 16, 32


_WINO_EPI = EPI_RELU | EPI_ADD | EPI_MASK | EPI_NOBIAS

_W43_G = ((1 / 4, 0, 0), (1 / 6, 1 / 6, 1 / 6), (1 / 6, -1 / 6, 1 / 6), (1 / 24, 1 / 12, 1 / 6), (1 / 24, -1 / 12, 1 / 6), (0, 0, 1))


def _wino_blocked_pack(u, ck):
    """(co, ci, ky, t) transformed weights -> [cout block of 32][chunk of ck cin][ky][t][ci][32]."""
    co, ci, kk, nt = u.shape
    ncb, nch = (co + 31) // 32, (ci + ck - 1) // ck
    p = torch.zeros((ncb * 32, nch * ck, kk, nt), device=u.device, dtype=u.dtype)
    p[:co, :ci] = u
    return p.view(ncb, 32, nch, ck, kk, nt).permute(0, 2, 4, 5, 3, 1).contiguous()


def wino43_weights(w, transpose, ck):
    """Winograd F(4,3)-along-x weights of a 3x3 layer in the blocked layout of ``risp_conv2d_wino43`` (fp64 algebra,
    stored in the dtype of ``w``)."""
    if transpose:
        w = w.flip(2, 3).transpose(0, 1)
    g = torch.tensor(_W43_G, dtype=torch.float64, device=w.device)
    return _wino_blocked_pack(torch.einsum('tk,oiyk->oiyt', g, w.double()).to(w.dtype), ck)


_W45_G = ((1, 0, 0, 0, 0), (1, 1, 1, 1, 1), (1, -1, 1, -1, 1), (1, 2, 4, 8, 16), (1, -2, 4, -8, 16),
          (1, 1 / 2, 1 / 4, 1 / 8, 1 / 16), (1, -1 / 2, 1 / 4, -1 / 8, 1 / 16), (0, 0, 0, 0, 1))
_W45_S = (1., -18., -18., 360., 360., 45. / 16., 45. / 16., 1.)


def wino45_weights(w, transpose, ck=4, layout=1):
    """Winograd F(4,5)-along-x weights of a 5x5 layer (include/risp.h: risp_conv2d_wino45), pure tensor algebra computed in
    fp64 and stored in the dtype of ``w``: [cout block of 32][chunk of 4 cin][ky][point group 2][cout block of 16: 2][ci 4][cout 16]
    [4 points] (one ds_read_b128 = the A operands of four points).  ``layout`` 0 = the plain blocked form [cout block of 32][chunk of
    4 cin][ky][t][ci][32] (tests/test_pack_algebra_cpu.py checks the algebra on it)."""
    if transpose:                                   # backward-data: roles swapped, taps rotated by 180 degrees
        w = w.flip(2, 3).transpose(0, 1)
    g = torch.tensor(_W45_G, dtype=torch.float64, device=w.device) / torch.tensor(_W45_S, dtype=torch.float64,
                                                                                  device=w.device)[:, None]
    u = torch.einsum('tk,oiyk->oiyt', g, w.double()).to(w.dtype)              # (co, ci, ky, t)
    if layout == 0:
        return _wino_blocked_pack(u, ck)
    assert ck == 4
    co, ci = u.shape[0], u.shape[1]
    ncb, nch = (co + 31) // 32, (ci + 3) // 4
    p = torch.zeros((ncb * 32, nch * 4, 5, 8), device=u.device, dtype=u.dtype)
    p[:co, :ci] = u
    #        (ncb, mb, m, nch, k, ky, pg, pt) -> (ncb, nch, ky, pg, mb, k, m, pt)
    return p.view(ncb, 2, 16, nch, 4, 5, 2, 4).permute(0, 3, 5, 6, 1, 4, 2, 7).contiguous()


def f16x2_weights(w, transpose):
    """Pre-split pack of ``risp_conv2d_f16x2`` (include/risp.h): a 16-byte header whose first float is 1 / s_w, then
    [chunk of 16 cin][tap][part: hi, lo][channel half][cout padded to 32 or 64][8 channels] halves of w * s_w, where
    s_w = 2^k puts max|w| into [2^14, 2^15), hi = rn_f16(w s_w), lo = rn_f16(w s_w - hi).  ``transpose``: the backward-data
    layer of a forward weight (roles swapped, taps rotated by 180 degrees).  Pure tensor algebra on the device of ``w`` (no
    host synchronisation); tests/test_pack_algebra_cpu.py pins it on the CPU.  Returns a float16 tensor."""
    if transpose:
        w = w.flip(2, 3).transpose(0, 1)
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    nt, nch = (co + 31) // 32, (ci + 15) // 16
    _, e = torch.frexp(w.detach().abs().max())                # max|w| = m 2^e, m in [0.5, 1)
    sw = torch.ldexp(torch.ones((), device=w.device), 15 - e)       # max|w| s_w = m 2^15
    ws = w.detach().float() * sw
    hi = ws.half()
    lo = (ws - hi.float()).half()
    p = torch.zeros((2, nch * 16, k * k, nt * 32), device=w.device, dtype=torch.float16)
    p[0, :ci, :, :co] = hi.permute(1, 2, 3, 0).reshape(ci, k * k, co)
    p[1, :ci, :, :co] = lo.permute(1, 2, 3, 0).reshape(ci, k * k, co)
    #       (part, chunk, half, 8, tap, cout) -> (chunk, tap, part, half, cout, 8)
    p = p.view(2, nch, 2, 8, k * k, nt * 32).permute(1, 4, 0, 2, 5, 3).contiguous()
    hdr = torch.zeros(4, device=w.device, dtype=torch.float32)
    hdr[0] = 1.0 / sw
    return torch.cat([hdr.view(torch.float16), p.reshape(-1)])


def toep_weights(w, transpose=False, keep=None):
    """Pre-split Toeplitz-band pack of ``risp_conv2d_toep`` (include/risp.h) from a layer's (cout <= 4, cin, k, k) tensor, k = 5 or
    9: a 16-byte header whose first float is 1 / s_w, then [cin][ky][part: hi, lo][window half][row m = 8 cout + j, padded to 32 - or to
    96 for the 5 to 12 couts of a 5-tap layer][8 window slots] halves with band[m][u] = w[co][ci][ky][u - j + k // 2 - 4] * s_w - the filter row as seen by the j-th pixel of
    a block of 8 from a window of 16 input pixels that starts 4 pixels left of the block.  ``transpose``: the backward-data layer
    of a FORWARD weight (roles swapped, taps rotated by 180 degrees) restricted to its first ``keep`` input channels.  Scale and
    split as in ``f16x2_weights``; pure tensor algebra on the device of ``w``.  Returns a float16 tensor."""
    if transpose:
        w = w[:, :keep].flip(2, 3).transpose(0, 1)
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    if k not in (5, 9) or co > (12 if k == 5 else 4):
        raise ValueError('Toeplitz-band pack: %d output channels (at most 4; 12 for 5 taps), %d taps (5 or 9)' % (co, k))
    nb = 1 if co <= 4 else 3                                        # row blocks of 4 couts x 8 positions
    _, e = torch.frexp(w.detach().abs().max())
    sw = torch.ldexp(torch.ones((), device=w.device), 15 - e)
    ws = w.detach().float() * sw
    hi = ws.half()
    parts = (hi, (ws - hi.float()).half())
    band = torch.zeros((2, ci, k, 4 * nb, 8, 16), device=w.device, dtype=torch.float16)      # (part, ci, ky, cout, j, u)
    for j in range(8):
        for part in range(2):
            band[part, :, :, :co, j, j + 4 - k // 2:j + 4 - k // 2 + k] = parts[part].permute(1, 2, 0, 3)
    #       (part, ci, ky, m, half, 8) -> (ci, ky, part, half, m, 8)
    band = band.view(2, ci, k, 32 * nb, 2, 8).permute(1, 2, 0, 4, 3, 5).contiguous()
    hdr = torch.zeros(4, device=w.device, dtype=torch.float32)
    hdr[0] = 1.0 / sw
    return torch.cat([hdr.view(torch.float16), band.reshape(-1)])


_TAPOUT_ROWS = {}


def tapout_weights(w, transpose=False, keep=None):
    """Pre-split pack of ``risp_conv2d_tapout`` (include/risp.h) from a layer's (cout <= 3, cin % 16 == 0, k, k) tensor, k = 5 or 9: a
    16-byte header whose first float is 1 / s_w, then [chunk of 16 cin][kx][part: hi, lo][channel half][row m, 32][8 channels] halves of
    w * s_w with row m = 4 ky + co for ky < 8 and 4 co + 3 for ky = 8 - the filter ROWS sit in the rows of the matrix instruction, the
    filter column is a shift of the pixel operand.  ``transpose``: the backward-data layer of a FORWARD weight (roles swapped, taps
    rotated by 180 degrees) restricted to its first ``keep`` input channels.  Scale and split as in ``f16x2_weights``; pure tensor
    algebra on the device of ``w``.  Returns a float16 tensor."""
    if transpose:
        w = w[:, :keep].flip(2, 3).transpose(0, 1)
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    if k not in (5, 9) or co > 3 or ci % 16:
        raise ValueError('tap-row pack: %d output channels (at most 3), %d input channels (a multiple of 16), %d taps (5 or 9)' % (co, ci, k))
    _, e = torch.frexp(w.detach().abs().max())
    sw = torch.ldexp(torch.ones((), device=w.device), 15 - e)
    ws = w.detach().float() * sw
    hi = ws.half()
    parts = (hi, (ws - hi.float()).half())
    rows = torch.zeros((2, ci, k, 32), device=w.device, dtype=torch.float16)                 # (part, ci, kx, m)
    key = (co, k, w.device)
    m_idx = _TAPOUT_ROWS.get(key)
    if m_idx is None:                                # row of (cout c, filter row ky), in (c, ky) order
        m_idx = _TAPOUT_ROWS[key] = torch.tensor([4 * ky + c if ky < 8 else 4 * c + 3 for c in range(co) for ky in range(k)], device=w.device)
    for part in range(2):                            # (co, ci, ky, kx) -> (ci, kx, co * ky): one indexed store per part
        rows[part][:, :, m_idx] = parts[part].permute(1, 3, 0, 2).reshape(ci, k, co * k)
    #       (part, chunk, half, 8, kx, m) -> (chunk, kx, part, half, m, 8)
    p = rows.view(2, ci // 16, 2, 8, k, 32).permute(1, 4, 0, 2, 5, 3).contiguous()
    hdr = torch.zeros(4, device=w.device, dtype=torch.float32)
    hdr[0] = 1.0 / sw
    return torch.cat([hdr.view(torch.float16), p.reshape(-1)])


def toep_first_weights(w):
    """Pre-split pack of ``risp_conv2d_toep_first`` (include/risp.h) from a 9x9 first layer's (cout, cin <= 16, 9, 9) tensor: a
    16-byte header whose first float is 1 / s_w, then [cout block of 32][cin][ky][part: hi, lo][taps 0-7 | tap 8 and 7 zeros][cout][8]
    halves of w * s_w - per cout the filter row in two 16-byte slots, from which the kernel cuts its 8 shifted windows.  Scale and
    split as in ``f16x2_weights``.  Returns a float16 tensor."""
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    if k != 9 or ci > 16:
        raise ValueError('first-layer band pack: a 9x9 layer with at most 16 input channels, got %dx%d with %d' % (k, k, ci))
    nb = (co + 31) // 32
    _, e = torch.frexp(w.detach().abs().max())
    sw = torch.ldexp(torch.ones((), device=w.device), 15 - e)
    ws = w.detach().float() * sw
    hi = ws.half()
    p = torch.zeros((2, nb * 32, ci, k, 16), device=w.device, dtype=torch.float16)
    p[0, :co, :, :, :k] = hi
    p[1, :co, :, :, :k] = (ws - hi.float()).half()
    #       (part, block, cout, ci, ky, slot, 8) -> (block, ci, ky, part, slot, cout, 8)
    p = p.view(2, nb, 32, ci, k, 2, 8).permute(1, 3, 4, 0, 5, 2, 6).contiguous()
    hdr = torch.zeros(4, device=w.device, dtype=torch.float32)
    hdr[0] = 1.0 / sw
    return torch.cat([hdr.view(torch.float16), p.reshape(-1)])


def thin5_weights(w, transpose=False):
    """Pre-split pack of ``risp_conv2d_thin5`` (include/risp.h) from a 5x5 layer's (cout 32 | 64, cin <= 3, 5, 5) tensor: a 16-byte header
    whose first float is 1 / s_w, then [cout block of 32][kx][part: hi, lo][half of the reduction index][cout][8] halves of w * s_w with the
    reduction index k = 3 ky + c (k = 15 and missing channels zero) - (filter row, channel) pairs fill the 16 reduction slots of one matrix
    instruction, the filter column shifts the pixel operand.  ``transpose``: the backward-data layer of a FORWARD (cin <= 3 couts, cout,
    5, 5) weight (roles swapped, taps rotated by 180 degrees).  Scale and split as in ``f16x2_weights``.  Returns a float16 tensor."""
    if transpose:
        w = w.flip(2, 3).transpose(0, 1)
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    if k != 5 or ci > 3 or co not in (32, 64):
        raise ValueError('thin-input pack: a 5x5 layer with at most 3 input and 32 or 64 output channels, got %dx%d %d -> %d' % (k, k, ci, co))
    _, e = torch.frexp(w.detach().abs().max())
    sw = torch.ldexp(torch.ones((), device=w.device), 15 - e)
    ws = w.detach().float() * sw
    hi = ws.half()
    parts = (hi, (ws - hi.float()).half())
    p = torch.zeros((2, co, k, 16), device=w.device, dtype=torch.float16)                   # (part, cout, kx, 3 ky + c)
    for part in range(2):
        p[part][:, :, :15].view(co, k, k, 3)[..., :ci] = parts[part].permute(0, 3, 2, 1)   # (cout, kx, ky, c)
    #       (part, block, m, kx, half, 8) -> (block, kx, part, half, m, 8)
    p = p.view(2, co // 32, 32, k, 2, 8).permute(1, 3, 0, 4, 2, 5).contiguous()
    hdr = torch.zeros(4, device=w.device, dtype=torch.float32)
    hdr[0] = 1.0 / sw
    return torch.cat([hdr.view(torch.float16), p.reshape(-1)])


def narrow3_weights(w, transpose=False, keep=None):
    """Pre-split pack of ``risp_conv2d_narrow3`` (include/risp.h) from a 3x3 layer's (cout <= 4, cin in 16 .. 64 and a multiple of 16, 3, 3)
    tensor: a 16-byte header whose first float is 1 / s_w, then [chunk of 16 cin][kx][part: hi, lo][channel half][row m, 32][8 channels]
    halves of w * s_w with row m = 4 ky + co - (filter row, cout) pairs in the rows of the matrix instruction, the filter column a
    shift of the pixel operand.  ``transpose``: the backward-data layer of a FORWARD weight (roles swapped, taps rotated by 180
    degrees) restricted to its first ``keep`` input channels.  Scale and split as in ``f16x2_weights``.  Returns a float16 tensor."""
    if transpose:
        w = w[:, :keep].flip(2, 3).transpose(0, 1)
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    if k != 3 or co > 4 or ci % 16 or not 16 <= ci <= 64:
        raise ValueError('narrow 3x3 pack: at most 4 output channels, 16 .. 64 input channels (a multiple of 16), 3 taps; got %d -> %d, %d taps' % (ci, co, k))
    _, e = torch.frexp(w.detach().abs().max())
    sw = torch.ldexp(torch.ones((), device=w.device), 15 - e)
    ws = w.detach().float() * sw
    hi = ws.half()
    parts = (hi, (ws - hi.float()).half())
    rows = torch.zeros((2, ci, k, 32), device=w.device, dtype=torch.float16)                 # (part, ci, kx, m)
    for part in range(2):                            # (co, ci, ky, kx) -> (ci, kx, ky, co) -> rows 4 ky + co
        rows[part][:, :, :12].view(ci, k, 3, 4)[..., :co] = parts[part].permute(1, 3, 2, 0)
    #       (part, chunk, half, 8, kx, m) -> (chunk, kx, part, half, m, 8)
    p = rows.view(2, ci // 16, 2, 8, k, 32).permute(1, 4, 0, 2, 5, 3).contiguous()
    hdr = torch.zeros(4, device=w.device, dtype=torch.float32)
    hdr[0] = 1.0 / sw
    return torch.cat([hdr.view(torch.float16), p.reshape(-1)])


# Arithmetic of the wide hidden layers (3x3, cin % 16 == 0, cout 32 / 64): 'f16x2' (default) = split precision on the f16 matrix
# pipe - each fp32 operand as two f16 halves, three products, fp32 accumulation (risp_conv2d_f16x2: fp32 tensors in and out,
# error against float64 no larger than the fp32 kernels'); 'f32' = the fp32 matrix-core kernels (Winograd F(4,3) / F(2,3)).
CONV_ARITH = os.environ.get('RISP_CONV_ARITH', 'f16x2')
if CONV_ARITH not in ('f16x2', 'f32'):
    raise ValueError("RISP_CONV_ARITH must be 'f16x2' or 'f32', got %r" % CONV_ARITH)


def f16x2_addressable(cin, cout, h, w):
    """``risp_conv2d_f16x2`` addresses an image through buffer resources of 2^31 - 1 bytes whose range check covers the per-lane
    offset: pixel + up to 16 input planes on the way in, pixel + the cout plane on the way out (residual / mask rows alike).  Every
    such offset stays below 2^31 iff max(cin, cout) * H * W * 4 < 2^31 (the entry point checks the same bound); larger images -
    an untiled 3000 x 4000 frame through a 64-channel layer - take the fp32 kernels with their 64-bit addresses."""
    return max(cin, cout) * h * w * 4 < (1 << 31)


def small_weights(w, transpose=False, keep=None):
    """Weights of ``risp_conv2d_small`` from a layer's (cout,cin,k,k) tensor: [cin][k][k][P] with P = 4 for cout <= 4, else 12
    (couts zero-padded) - except for cout == 3, whose pack is [cin][k + 1][k][4] = (w0, w1, w2[ky], w2[ky - 1]), row k =
    (0, 0, 0, w2[k - 1]): the fourth slot carries the third cout's weight of the filter row above, so that one packed FMA
    serves cout 2 of two output rows (include/risp.h).  ``transpose``: the backward-data layer of a FORWARD weight (roles
    swapped, taps rotated by 180 degrees) restricted to its first ``keep`` input channels."""
    if transpose:
        w = w[:, :keep].flip(2, 3).permute(0, 2, 3, 1)                # [cin_b = cout_f][ky][kx][cout_b = cin_f]
    else:
        w = w.permute(1, 2, 3, 0)                                      # [cin][ky][kx][cout]
    cout = w.shape[3]
    if cout > 12:
        raise ValueError('small-cout layer: %d output channels (at most 12)' % cout)
    rows = w.shape[1] + (1 if cout == 3 else 0)
    pack = torch.zeros((w.shape[0], rows, w.shape[2], 4 if cout <= 4 else 12), device=w.device, dtype=w.dtype)
    pack[:, :w.shape[1], :, :cout] = w
    if cout == 3:
        pack[:, 1:, :, 3] = w[..., 2]
    return pack, cout


def k3_weights(w):
    """Linear-k weights of ``risp_conv2d_k3`` from a (cout <= 64, cin, k, k) first layer: [cout block][2 ceil(cin k k / 2)]
    [block width] with row r = w[co][ci][ky][kx] at r = (ci k + ky) k + kx (include/risp.h)."""
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    cp = L.load().risp_conv_k3_cout_block(ci, k)
    rows, ncb = 2 * ((ci * k * k + 1) // 2), (co + cp - 1) // cp
    p = torch.zeros((rows, ncb * cp), device=w.device, dtype=w.dtype)
    p[:ci * k * k, :co] = w.reshape(co, -1).t()
    return p.view(rows, ncb, cp).permute(1, 0, 2).contiguous()


def srcnn_fold_tables(w1):
    """(rcase (9+P, cout*k*k), wconst (cout*k*k, 9+P)) of SRCNNRes' first layer w1 (cout, 12+P, k, k): see SrcnnResFold."""
    k, p = w1.shape[2], w1.shape[2] // 2
    wc = w1[:, 3:]
    # border case i keeps the taps lo..hi of a filter row / column; all k x k case sums at once as two contractions with
    # the 0/1 case-membership matrix (float64, rounded once: 81 slice sums per layer were 81 launches each, and the packs
    # of every slot are rebuilt after each proxy fine-tuning round)
    member = torch.zeros((k, k), dtype=torch.float64, device=w1.device)
    for i in range(k):
        lo, hi = (p - i, k - 1) if i < p else (0, k - 1 - (i - p))
        member[i, lo:hi + 1] = 1.0
    rc = torch.einsum('ocyx,iy,jx->ocij', wc.double(), member, member).to(w1.dtype)     # (cout, 9+P, k, k)
    rcase = rc.permute(1, 0, 2, 3).reshape(wc.shape[1], -1).contiguous()
    wconst = wc.permute(0, 2, 3, 1).reshape(-1, wc.shape[1]).contiguous()
    return rcase, wconst


def pack_kinds(k, cin, cout, transpose=False):
    """Which packs a (cout, cin, k, k) layer holds for its forward (``transpose`` False) or backward-data direction - what
    ``PackedConv`` builds and ``route`` may choose from (the general pack of risp_conv2d always exists)"""
    c_in, c_out = (cout, cin) if transpose else (cin, cout)
    have = []
    if k == 3:
        have.append('wino43')
    if k == 5 and (c_in % 4 == 0 or c_in < 4):
        have.append('wino45')
    if k in (3, 5) and c_in % 16 == 0 and c_out in (32, 64):
        have.append('f16x2')
    if k == 5 and c_in <= 3 and c_out in (32, 64):             # split precision for 5x5 layers with at most 3 input channels (risp_conv_thin5.hip)
        have.append('thin5')
    if k in (3, 9) and cin in (3, 4) and cout <= 64:           # first layers (forward geometry; route() uses them forward only)
        have.append('k3')
    if k == 9 and cin in (3, 4):
        have.append('toep_first')
    return have


def small_has_toep(k, cout):
    """does a small-cout layer (``SmallConv``) hold a Toeplitz-band pack for risp_conv2d_toep"""
    return (k == 9 and cout <= 4) or (k == 5 and cout <= 12)


def small_has_tapout(k, cin, cout):
    """... and a tap-row pack for risp_conv2d_tapout (at most 3 couts, whole chunks of 16 input channels)"""
    return k in (5, 9) and cout <= 3 and cin % 16 == 0 and cin > 0


def small_has_narrow3(k, cin, cout):
    """... and a (filter row, cout) pack for risp_conv2d_narrow3: 3x3, at most 4 couts, 16 .. 64 input channels in whole chunks of 16"""
    return k == 3 and cout <= 4 and cin % 16 == 0 and 16 <= cin <= 64


class PackedConv:
    """Device-side packed weights of one layer, forward and backward-data.  Every pack is built on FIRST USE (round 6): a layer holds up to
    nine packs and a launch reads one - the fp32 Winograd packs (einsum -> a Tensile GEMM and ~10 torch launches each) only serve
    RISP_CONV_ARITH=f32 and the shapes the split-precision kernels do not take, and in the proxy fine-tuning loop every optimizer step of a
    proxy rebuilds its packs (darts_ft_model.py:206-246)."""

    def __init__(self, weight, bias):
        w = _dev(weight.detach(), 'weight')
        self._w = w
        self.cout, self.cin, self.k = w.shape[0], w.shape[1], w.shape[2]
        self.bias = _dev(bias.detach(), 'bias')
        self._kf, self._kb = pack_kinds(self.k, self.cin, self.cout, False), pack_kinds(self.k, self.cin, self.cout, True)

    def _general(self, transpose):
        lib = L.load()
        ci, co = (self.cout, self.cin) if transpose else (self.cin, self.cout)
        pack = torch.empty(lib.risp_conv_wpack_floats(ci, co, self.k), device=self._w.device)
        L.call('risp_conv_pack_weights', _p(self._w), ci, co, self.k, int(transpose), _p(pack), _stream())
        return pack

    def __getattr__(self, name):                     # only reached for packs that have not been built yet
        w = self.__dict__.get('_w')
        if w is None or name.startswith('_'):
            raise AttributeError(name)
        kf, kb = self._kf, self._kb
        build = {
            'fwd': lambda: self._general(False), 'bwd': lambda: self._general(True),
            # fp32 Winograd packs: RISP_CONV_ARITH=f32, and the layers the split-precision kernels do not take
            'wino43_fwd': lambda: wino43_weights(w, False, L.load().risp_conv_wino43_chunk()) if 'wino43' in kf else None,
            'wino43_bwd': lambda: wino43_weights(w, True, L.load().risp_conv_wino43_chunk()) if 'wino43' in kb else None,
            'wino45_fwd': lambda: wino45_weights(w, False) if 'wino45' in kf else None,
            'wino45_bwd': lambda: wino45_weights(w, True) if 'wino45' in kb else None,
            # split-precision packs (see CONV_ARITH); either direction on its own
            'f16x2_fwd': lambda: f16x2_weights(w, False) if 'f16x2' in kf else None,
            'f16x2_bwd': lambda: f16x2_weights(w, True) if 'f16x2' in kb else None,
            'thin5_fwd': lambda: thin5_weights(w, False) if 'thin5' in kf else None,
            'thin5_bwd': lambda: thin5_weights(w, True) if 'thin5' in kb else None,
            # first layers (3 plain or 4 space-to-depth input channels): the linear-k kernel, risp_conv_k3.hip
            'k3': lambda: k3_weights(w) if 'k3' in kf else None,
            # ... and the 9x9 ones on the f16 matrix pipe in split precision, risp_conv_toep_first.hip
            'toep_first': lambda: toep_first_weights(w) if 'toep_first' in kf else None,
            'w32': lambda: w.float().contiguous() if 'toep_first' in kf else None,     # for the exact recomputation of ReLU ties
        }.get(name)
        if build is None:
            raise AttributeError(name)
        val = build()
        self.__dict__[name] = val
        return val


class SmallConv:
    """Weights of a layer with at most 4 output channels in the [cin][k][k][4] layout of ``risp_conv2d_small``.
    ``weight`` is the layer's (cout,cin,k,k) tensor; ``transpose=True`` builds the backward-data layer of a
    FORWARD weight (roles swapped, taps rotated by 180 degrees), restricted to its first ``keep`` input channels."""

    def __init__(self, weight, bias=None, transpose=False, keep=None):
        w = _dev(weight.detach(), 'weight')
        self.wpack, self.cout = small_weights(w, transpose, keep)
        self.cin, self.k = self.wpack.shape[0], self.wpack.shape[2]
        self._src = (w, transpose, keep)
        self.bias = _dev(bias.detach(), 'bias') if bias is not None else None

    def __getattr__(self, name):                     # the matrix-pipe packs, built when a launch first asks for them
        src = self.__dict__.get('_src')
        if src is None or name not in ('toep', 'tapout', 'narrow3'):
            raise AttributeError(name)
        w, transpose, keep = src
        if name == 'narrow3':   # 3x3 tails on the f16 matrix pipe (risp_conv2d_narrow3)
            val = narrow3_weights(w, transpose, keep) if small_has_narrow3(self.k, self.cin, self.cout) else None
        elif name == 'toep':      # the same layer for the f16 matrix pipe (risp_conv2d_toep): 5- and 9-tap rows, at most 4 couts (12 for 5 taps)
            val = toep_weights(w, transpose, keep) if small_has_toep(self.k, self.cout) else None
        else:                   # ... and with the filter rows in the rows of the matrix instruction (risp_conv2d_tapout)
            val = tapout_weights(w, transpose, keep) if small_has_tapout(self.k, self.cin, self.cout) else None
        self.__dict__[name] = val
        return val


def _group_fields(d, n, group, wpack, bias):
    """Fill the grouped-launch fields of a ConvDesc: ``group`` = (G, flags) with the members' packs stacked along
    dimension 0 of ``wpack`` / ``bias``; the launch then covers G * n images (include/risp.h: group_n)."""
    if group is None:
        return n
    g, flags = group
    if wpack.shape[0] != g or (bias is not None and bias.shape[0] != g):
        raise ValueError('grouped launch: %d members but the stacked packs hold %d' % (g, wpack.shape[0]))
    d.group_n, d.group_flags = n, flags
    d.wpack_gs = wpack.stride(0) * wpack.element_size() // 4        # in floats, whatever the pack's dtype
    d.bias_gs = bias.stride(0) if bias is not None else 0
    d.N = g * n
    return g * n


def toep_grid_ok(images, h, w):
    """training launches: enough 16 x 256 tiles for the persistent grid of ``risp_conv2d_toep`` (else the vector-FMA kernel with
    its input-channel split fills the chip better)"""
    tiles = ((h + 31) // 32) * ((w + 127) // 128) if w <= 128 else ((h + 15) // 16) * ((w + 255) // 256)
    return images * tiles >= TOEP_MIN_TILES


def _tapout_ok(sc, h, w, epi=0):
    return (CONV_ARITH == 'f16x2' and small_has_tapout(sc.k, sc.cin, sc.cout) and not (epi & EPI_SHUFFLE2) and w % 4 == 0
            and sc.cin * h * w < (1 << 30) and h * w < (1 << 24))


def _toep_ok(sc, h, w):
    # (any width: planes of at most 128 pixels run with two rows folded into the 32 columns of the matrix instruction; the 12-cout
    # form cannot fold and is ~20 % slower than the vector kernel there, but a layer keeps ONE arithmetic whatever the crop it sees)
    return CONV_ARITH == 'f16x2' and small_has_toep(sc.k, sc.cout) and w % 4 == 0 and sc.cin * h * w < (1 << 30)


# risp_conv2d_toep_first (9x9 first layers on the f16 matrix pipe): 'train' (default) = inference AND training forwards, the latter with
# EXACT ReLU decisions (risp_conv2d_toep_first_exact: outputs whose pre-activation is within the arithmetic's own error of zero are
# recomputed in double) - one first-layer kernel for model.test() and the training forward; 'infer' = inference launches only, training
# forwards on the fp32 kernel risp_conv2d_k3 (the default of round 4); 'plain' = training without the exact decisions; '0' = never.
# Round 4 kept training on the fp32 kernel because its DARTS fixture (tests/golden/darts_step.npz) holds one first-layer pre-activation
# at 3.6e-9 of its layer: which side of the ReLU it lands on is decided by the last bits of the slot's INPUT - the rounding of whatever
# ran upstream - and that one mask bit moves iteration 1 by 9e-4.  The reference's own fp32 and float64 runs disagree on that scenario
# by 3.7e-4, so it cannot pin an arithmetic; round 5 pins the step on scenarios the reference agrees with itself on
# (darts_step_kf, darts_step_n3: tests/golden/make_golden.py) and keeps the old one as the regression of the '0' route.
TOEP_FIRST = os.environ.get('RISP_CONV_TOEP_FIRST', 'train')
if TOEP_FIRST not in ('0', 'infer', 'train', 'plain'):
    raise ValueError("RISP_CONV_TOEP_FIRST must be '0', 'infer', 'train' or 'plain', got %r" % TOEP_FIRST)
_TIES = {}                                              # per (device, stream): the tie list of risp_conv2d_toep_first_exact
TIES_MAX = 1 << 20


def _tie_list(device):
    """one list per stream: the launches of a slot's operators run on two streams (section 5.1 of DESIGN.md), and a list shared by two
    launches in flight would lose ties"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _TIES.get(key)
    if t is None:
        t = _TIES[key] = torch.zeros(1 + TIES_MAX, device=device, dtype=torch.int32)
    return t


TOEP_MIN_TILES = 256                        # training launches: tiles below which the vector-FMA kernel with its channel split serves
TAPOUT_MIN_ITEMS = 128                      # ... and work items (image, 128-column strip, 32-row segment) below which it serves instead of risp_conv2d_tapout
TAPOUT_INFER_SEG = 64                       # rows of a work item's segment in inference launches (fixed: a result must not depend on the batch)


def tapout_grid_ok(images, h, w):
    """training launches: enough (strip, segment) work items for the persistent grid of ``risp_conv2d_tapout``"""
    return images * ((w + 127) // 128) * max(1, h // 32) >= TAPOUT_MIN_ITEMS


def tapout_seg(images, h, w, infer):
    """rows of a work item's segment: fixed for inference, chosen by the grid of ALL the images of a (grouped) launch for training -
    the per-member form of a grouped launch passes the grouped launch's value, so that both cut the planes alike (same bits)"""
    return min(TAPOUT_INFER_SEG, h) if infer else L.load().risp_conv_tapout_seg_rows(images, h, w)


def route_small(k, cin, cout, h, w, images, infer=False, has_mask=False, has_toep=True, split=None, has_tapout=False, has_narrow3=False):
    """The dispatch of ``conv_small`` (layers with at most 12 output channels).  Under RISP_CONV_ARITH=f16x2, W % 4 == 0, no mask:
    layers that hold a tap-row pack (at most 3 couts, cin % 16 == 0, 5 or 9 taps) run on risp_conv2d_tapout, the other 5- and 9-tap
    layers that hold a Toeplitz-band pack (4 couts, or 5 .. 12 with 5 taps, or odd channel counts) on risp_conv2d_toep (both: f16 matrix
    pipe, split precision) - ALWAYS for inference (a tile's result must not depend on the batch it travels in), for training when the
    grid holds enough work (TAPOUT_MIN_ITEMS / TOEP_MIN_TILES; or the caller forces ``split`` = 0: the per-member form of a grouped
    launch follows the grouped grid); everything else on risp_conv2d_small (vector FMAs; small training grids split their input
    channels over several workgroups per tile: risp_conv2d_small_split, never for inference).  3x3 layers that hold a (filter row,
    cout) pack (at most 4 couts, 16 .. 64 input channels; ``has_narrow3``: the caller also checks the epilogue) run on
    risp_conv2d_narrow3 in inference launches, whatever the grid."""
    mat = CONV_ARITH == 'f16x2' and w % 4 == 0 and cin * h * w < (1 << 30) and not has_mask
    if mat and has_narrow3 and infer:                 # 3x3 tails (at most 4 couts) of inference launches: one scale per wave and row - any grid,
        return 'risp_conv2d_narrow3'                  # any batch.  (Training keeps the vector kernel: equal at batch 32 - both read 64 planes at
        #                                               2.6 TB/s - and the search step's goldens keep their arithmetic.)
    if mat and has_tapout and h * w < (1 << 24) and (infer or (split == 0 if split is not None else tapout_grid_ok(images, h, w))):
        return 'risp_conv2d_tapout'
    if mat and has_toep and (infer or (split == 0 if split is not None else toep_grid_ok(images, h, w))):
        return 'risp_conv2d_toep'
    return 'risp_conv2d_small'


def conv_small(x, sc, n, h, w, epi=0, add=None, add_c=0, mask=None, infer=False, out=None, group=None, split=None, tile_sums=None,
               seg_rows=None):
    """One launch of a layer with at most 12 output channels (``route_small``).  ``infer``: never split the input channels over
    workgroups - the split depends on the grid, and an inference result must not depend on the batch a tile travels in
    (test_split.py batches tiles).  ``group``: see ``_group_fields``; ``split``: force the channel split (the per-member form of a
    grouped launch uses the split the grouped grid would take; 0 = the matrix-pipe kernel) and ``seg_rows`` the segment height of
    risp_conv2d_tapout (``tapout_seg`` of the grouped launch).  ``tile_sums``: a list that receives the per-tile sums of the input
    planes when a matrix-pipe kernel serves the launch (see ``rect_sums``)."""
    if sc.bias is None:
        epi |= EPI_NOBIAS
    nn_ = n * (group[0] if group else 1)
    if out is None:
        shape = (nn_, sc.cout // 4, 2 * h, 2 * w) if epi & EPI_SHUFFLE2 else (nn_, sc.cout, h, w)
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    d = L.ConvDesc(N=n, H=h, W=w, cin=sc.cin, cout=sc.cout, ksize=sc.k, load_mode=LOAD_PLAIN, cin_img=0, epilogue=epi,
                   add_c=add_c, x=_p(x), wpack=_p(sc.wpack), bias=_p(sc.bias), cvals=None, add=_p(add), mask=_p(mask),
                   y=_p(out))
    has_toep = small_has_toep(sc.k, sc.cout)          # (the packs themselves are built when a launch first asks for them)
    has_tapout = small_has_tapout(sc.k, sc.cin, sc.cout) and not (epi & EPI_SHUFFLE2)
    has_narrow3 = small_has_narrow3(sc.k, sc.cin, sc.cout) and not (epi & ~(EPI_RELU | EPI_SHUFFLE2 | EPI_NOBIAS)) and add is None
    entry = route_small(sc.k, sc.cin, sc.cout, h, w, nn_, infer, mask is not None, has_toep, split, has_tapout, has_narrow3)
    if entry == 'risp_conv2d_narrow3':
        pack = sc.narrow3
        d.wpack = _p(pack)
        _group_fields(d, n, group, pack, sc.bias)
        L.call(entry, C.byref(d), _stream())
        if MFMA_ISSUED_F16 is not None:                # per filter column and chunk of 16 channels: 3 products of 32 rows x 16 channels per pixel
            MFMA_ISSUED_F16[0] += 3 * 2.0 * 32 * sc.k * sc.cin * nn_ * h * w
        return out
    if entry == 'risp_conv2d_tapout':
        # the f16 matrix pipe with the filter rows as the rows of the matrix instruction: half the matrix work of the band form
        tapout = sc.tapout
        d.wpack = _p(tapout)
        _group_fields(d, n, group, tapout, sc.bias)
        seg = seg_rows if seg_rows is not None else tapout_seg(nn_, h, w, infer)
        if tile_sums is not None and sc.k == 9 and sc.cin == 64:
            ps = torch.empty((nn_, L.load().risp_conv_tapout_items(nn_, h, w, seg), 64), device=x.device, dtype=torch.float32)
            L.call('risp_conv2d_tapout_sums', C.byref(d), seg, _p(ps), _stream())
            tile_sums.append(ps)
        else:
            L.call('risp_conv2d_tapout', C.byref(d), seg, _stream())
        if MFMA_ISSUED_F16 is not None:                # per filter column and chunk of 16 channels: 3 products of 32 rows x 16 channels per pixel
            MFMA_ISSUED_F16[0] += 3 * 2.0 * 32 * sc.k * sc.cin * nn_ * h * w
        return out
    if entry == 'risp_conv2d_toep':
        # the f16 matrix pipe (Toeplitz bands of the filter rows as the A operand): 2.5-3 x the vector-FMA kernel on full grids
        toep = sc.toep
        d.wpack = _p(toep)
        _group_fields(d, n, group, toep, sc.bias)
        if tile_sums is not None:
            ps = torch.empty((nn_, L.load().risp_conv_toep_tiles(h, w), sc.cin), device=x.device, dtype=torch.float32)
            L.call('risp_conv2d_toep_sums', C.byref(d), _p(ps), _stream())
            tile_sums.append(ps)
        else:
            L.call('risp_conv2d_toep', C.byref(d), _stream())
        if MFMA_ISSUED_F16 is not None:                # per (ci, ky) and block of 8 pixels: 3 products of 32 rows x 16 window slots
            MFMA_ISSUED_F16[0] += 3 * 2.0 * (32 if sc.cout <= 4 else 96) * 16 / 8 * sc.cin * sc.k * nn_ * h * w
        return out
    _group_fields(d, n, group, sc.wpack, sc.bias)
    groups = 1 if infer else (split if split is not None else L.load().risp_conv_small_groups(C.byref(d)))
    if groups > 1:                                  # small grid: split the input channels over several workgroups per tile
        scratch = torch.empty((groups,) + tuple(out.shape), device=x.device, dtype=torch.float32)
        L.call('risp_conv2d_small_split', C.byref(d), _p(scratch), groups, _stream())
    else:
        L.call('risp_conv2d_small', C.byref(d), _stream())
    return out


def rect_sums(g1, rs, images, planes, h, w, k, tile_sums):
    """rs (images, planes * k * k) = the rectangle sums of the planes of g1 (include/risp.h: risp_rect_sums).  ``tile_sums``: what
    ``conv_small(..., tile_sums=[])`` collected while the same planes went through the backward-data convolution - then only their
    border rows and columns are read again (risp_rect_sums_tiles); empty or None: the whole planes (risp_rect_sums)."""
    if tile_sums and k == 9:
        ps = tile_sums[0]
        L.call('risp_rect_sums_tiles', _p(g1), _p(ps), _p(rs), images, planes, h, w, ps.shape[1], _stream())
    else:
        L.call('risp_rect_sums', _p(g1), _p(rs), images * planes, h, w, k, _stream())


# bench.py sets this to [0.0] to count the FLOPs the launches ISSUE on the matrix cores (diagnostic; None = off)
MFMA_ISSUED = None
_TAPS = {'risp_conv2d_wino43': (18, 4), 'risp_conv2d_wino45': (40, 4)}
# ... and, separately, the FLOPs the split-precision launches issue on the f16 matrix pipe (3 products per tap) and the time-free
# count of such launches: bench.py prices the two pipes against their own peaks
MFMA_ISSUED_F16 = None


def _issued_flops(entry, cin, cout, k, pixels):
    """2 x MACs of the MFMA instructions one launch executes, tile-edge padding not counted: the direct kernel
    multiplies k*k taps per pixel, the Winograd-x kernels 18 / 40 transformed taps per 4 pixels; output
    channels are padded to a multiple of 32 (the MFMA tile), input channels to the kernel's pair granularity."""
    if entry == 'risp_conv2d_k3':
        return 2.0 * (2 * ((cin * k * k + 1) // 2)) * ((cout + 31) // 32 * 32) * pixels
    taps, per = _TAPS.get(entry, (k * k, 1))
    return 2.0 * taps * (cin + cin % 2) * ((cout + 31) // 32 * 32) * pixels / per


def route(k, cin, cout, h, w, transpose=False, load=LOAD_PLAIN, epi=0, add_c=0, infer=False, aligned=True, have=()):
    """THE dispatch table of ``conv``: which entry point of the library serves one launch of a layer.  Pure function of the launch's
    geometry (``cin`` / ``cout`` as the launch sees them: swapped for a backward-data pass), its load mode and epilogue flags, whether
    every tensor is 16-byte aligned, whether it is an inference launch, and which packs the layer holds (``have``: a subset of
    'wino43', 'wino45', 'f16x2', 'k3', 'toep_first').  Rules, first match wins (tests/test_gpu_conv_modes.py::test_route_table pins the
    result for every layer of the four proxy families):

      1. first layers (3 plain / 4 space-to-depth input channels, 3x3 or 9x9, forward only, epilogue RELU | NOBIAS | CASEBIAS):
         9x9 under RISP_CONV_ARITH=f16x2 -> risp_conv2d_toep_first (inference; TOEP_FIRST 'train' / 'plain' also training, 'train'
         through risp_conv2d_toep_first_exact), otherwise -> risp_conv2d_k3 (fp32, linear reduction index);
      2. wide 3x3 / 5x5 layers (cin % 16 == 0, cout 32 or 64, plain load, epilogue within RELU | ADD | MASK | NOBIAS with a full-width
         residual, addressable through 2^31-byte buffers) under f16x2 -> risp_conv2d_f16x2 (split precision, f16 matrix pipe);
      2b. 5x5 with at most 3 input channels and 32 / 64 couts (the backward-data pass of a proxy's 3-cout tail), plain loads, epilogue
         within RELU | MASK | NOBIAS, under f16x2 -> risp_conv2d_thin5 (split precision; (filter row, channel) as the reduction index);
      3. 3x3 with the plain-16 conditions -> risp_conv2d_wino43 (fp32 F(4,3)); 5x5 with cin % 4 == 0 or cin < 4 -> risp_conv2d_wino45;
      4. everything else -> risp_conv2d (fp32 matrix cores, direct).
    (Layers with at most 12 output channels never come here: ``conv_small`` / ``route_small``.)"""
    plain16 = load == LOAD_PLAIN and w % 4 == 0 and not (epi & ~_WINO_EPI) and aligned
    if ('k3' in have and not transpose and w % 4 == 0 and h >= k - 1 and w >= k - 1 and aligned
            and ((load == LOAD_PLAIN and cin == 3) or (load == LOAD_UNSHUFFLE2 and cin == 4))
            and not (epi & ~(EPI_RELU | EPI_NOBIAS | EPI_CASEBIAS))):
        if ('toep_first' in have and CONV_ARITH == 'f16x2' and cin * h * w < (1 << 30)
                and (TOEP_FIRST in ('train', 'plain') or (TOEP_FIRST == 'infer' and infer))):
            return 'risp_conv2d_toep_first_exact' if (TOEP_FIRST == 'train' and not infer) else 'risp_conv2d_toep_first'
        return 'risp_conv2d_k3'
    if (CONV_ARITH == 'f16x2' and 'f16x2' in have and plain16 and (not (epi & EPI_ADD) or add_c == cout)
            and f16x2_addressable(cin, cout, h, w)):
        return 'risp_conv2d_f16x2'
    if (CONV_ARITH == 'f16x2' and 'thin5' in have and load == LOAD_PLAIN and not (epi & ~(EPI_RELU | EPI_MASK | EPI_NOBIAS)) and aligned
            and w % 4 == 0 and cout * h * w * 4 < (1 << 31)):
        return 'risp_conv2d_thin5'
    if k == 3 and 'wino43' in have and plain16:
        return 'risp_conv2d_wino43'
    if k == 5 and 'wino45' in have and plain16:
        return 'risp_conv2d_wino45'
    return 'risp_conv2d'


def _have(pc, transpose):
    """the packs of a layer that ``route`` may pick from, for this direction: what ``pack_kinds`` says the layer can hold (nothing is
    built by asking), minus what has been set to None on the object"""
    sfx = '_bwd' if transpose else '_fwd'
    held = pc.__dict__
    kinds = pack_kinds(pc.k, pc.cin, pc.cout, transpose)
    have = [name for name in ('wino43', 'wino45', 'f16x2', 'thin5') if name in kinds and held.get(name + sfx, 1) is not None]
    return have + [name for name in ('k3', 'toep_first') if name in kinds and held.get(name, 1) is not None]


def conv(x, pc, n, h, w, transpose=False, load=LOAD_PLAIN, cin_img=0, cvals=None, epi=0, add=None, add_c=0,
         mask=None, out=None, infer=False, group=None):
    """One fused convolution launch at resolution (h,w); returns the output tensor.  ``infer``: no backward pass will read this
    layer's activations (an inference launch: its result must not depend on the batch a tile travels in).  ``group`` = (G, flags):
    ``pc`` holds the packs of G same-geometry layers stacked along dimension 0 and the launch covers G * n images
    (``_group_fields``).  The kernel is chosen by ``route``."""
    cin, cout = (pc.cout, pc.cin) if transpose else (pc.cin, pc.cout)
    if transpose:
        epi |= EPI_NOBIAS
    if group is not None and pc.k == 3:
        raise ValueError('grouped launches are not available for 3x3 layers')
    nn_ = n * (group[0] if group else 1)
    if out is None:
        shape = (nn_, cout // 4, 2 * h, 2 * w) if epi & EPI_SHUFFLE2 else (nn_, cout, h, w)
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    aligned = (x.data_ptr() | out.data_ptr() | (add.data_ptr() if add is not None else 0) |
               (mask.data_ptr() if mask is not None else 0)) % 16 == 0
    have = _have(pc, transpose)
    if add is not None:
        epi_r = epi | EPI_ADD
    else:
        epi_r = epi
    entry = route(pc.k, cin, cout, h, w, transpose, load, epi_r, add_c, infer, aligned, have)
    if entry == 'risp_conv2d_toep_first_exact' and nn_ * cout * h * w >= (1 << 32):
        entry = 'risp_conv2d_toep_first'
    sfx = '_bwd' if transpose else '_fwd'
    wpack = {'risp_conv2d_k3': lambda: pc.k3, 'risp_conv2d_toep_first': lambda: pc.toep_first,
             'risp_conv2d_toep_first_exact': lambda: pc.toep_first, 'risp_conv2d_f16x2': lambda: getattr(pc, 'f16x2' + sfx),
             'risp_conv2d_thin5': lambda: getattr(pc, 'thin5' + sfx),
             'risp_conv2d_wino43': lambda: getattr(pc, 'wino43' + sfx), 'risp_conv2d_wino45': lambda: getattr(pc, 'wino45' + sfx),
             'risp_conv2d': lambda: pc.bwd if transpose else pc.fwd}[entry]()
    d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=pc.k, load_mode=load, cin_img=cin_img,
                   epilogue=epi, add_c=add_c, x=_p(x), wpack=_p(wpack),
                   bias=_p(pc.bias), cvals=_p(cvals), add=_p(add), mask=_p(mask), y=_p(out))
    _group_fields(d, n, group, wpack, None if transpose else pc.bias)
    if entry == 'risp_conv2d_toep_first_exact':
        w32 = pc.w32                                    # exact ReLU decisions (see TOEP_FIRST)
        L.call(entry, C.byref(d), _p(w32), w32.stride(0) if group else 0, _p(_tie_list(x.device)), TIES_MAX, _stream())
    else:
        L.call(entry, C.byref(d), _stream())
    if entry == 'risp_conv2d_f16x2':
        if MFMA_ISSUED_F16 is not None:
            MFMA_ISSUED_F16[0] += 3 * 2.0 * pc.k * pc.k * cin * cout * nn_ * h * w
    elif entry == 'risp_conv2d_thin5':
        if MFMA_ISSUED_F16 is not None:                # per filter column: 3 products of 16 (filter row, channel) slots x 32 couts per pixel
            MFMA_ISSUED_F16[0] += 3 * 2.0 * 5 * 16 * cout * nn_ * h * w
    elif entry.startswith('risp_conv2d_toep_first'):
        if MFMA_ISSUED_F16 is not None:
            MFMA_ISSUED_F16[0] += first_layer_issued(cin, cout, h, w, load == LOAD_UNSHUFFLE2) * nn_ * h * w
    elif MFMA_ISSUED is not None:
        MFMA_ISSUED[0] += _issued_flops(entry, cin, cout, pc.k, nn_ * h * w)
    return out


def first_layer_form(cin, cout, h, w, unshuffle=False):
    """which kernel ``risp_conv2d_toep_first`` launches (the dispatch of toep_first_impl, risp_conv_toep_first.hip): 'xwin' = the
    (channel, tap) reduction index of risp_conv_xwin.hip for 3 plain input channels, 'band' = the 16-slot window of round 5"""
    return 'xwin' if (not unshuffle and cin == 3 and cout <= 64 and cout * h * w < (1 << 29)) else 'band'


def first_layer_issued(cin, cout, h, w, unshuffle=False):
    """f16 matrix flops ISSUED per output pixel by the 9x9 first layer's kernel (3 split-precision products): 'xwin' spends one 32-slot
    reduction step per filter row (27 slots carry a (channel, tap)) on 64 couts; 'band' spends 16 window slots per (channel, filter
    row) (9 carry a tap) on the couts padded to 32"""
    if first_layer_form(cin, cout, h, w, unshuffle) == 'xwin':
        return 3 * 2.0 * 9 * 32 * 64
    return 3 * 2.0 * 9 * 16 * cin * ((cout + 31) // 32 * 32)


_WGRAD_SCRATCH = {}                                     # per (device, stream): the partial-sum slots of risp_conv2d_wgrad, grown on demand


def _wgrad_scratch(device, floats):
    """one scratch per stream (launches on a stream are ordered: the finishing launch of one call has read the slots before the next
    call writes them) instead of up to 255 MB from the caching allocator per call"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _WGRAD_SCRATCH.get(key)
    if t is None or t.numel() < floats:
        t = _WGRAD_SCRATCH[key] = torch.empty(floats, device=device, dtype=torch.float32)
    return t


def conv_wgrad(x, gy, cin, cout, k, n, h, w, load=LOAD_PLAIN, cin_img=0, cvals=None):
    """(dW (cout,cin,k,k), db (cout,)) of one layer from its input x and the gradient gy at its output (risp_conv2d_wgrad: per-workgroup
    partial sums added in index order - the same bits on every run).  ``load`` LOAD_CONSTCH (SRCNNRes' first layer, srcnn_res_arch.py:
    41-46: ``cin_img`` image channels followed by planes that hold the per-image constants ``cvals``): the gradient of a constant plane's
    taps is cvals^T @ (rectangle sums of gy) - risp_rect_sums, the sums its backward-data pass needs anyway - and only the image channels
    go through the matrix kernel (cin_img * k <= 32: one chain per filter ROW with (channel, column) pairs as the matrix columns)."""
    gy = _dev(gy, 'grad')
    if load == LOAD_CONSTCH and cin_img * k <= 32 and cvals is not None and h >= k // 2 and k // 2 <= w <= 8192:   # (risp_rect_sums: W <= 8192)
        dw_img, db = conv_wgrad(x, gy, cin_img, cout, k, n, h, w)
        rs = torch.empty((n, cout * k * k), device=gy.device, dtype=torch.float32)
        L.call('risp_rect_sums', _p(gy), _p(rs), n * cout, h, w, k, _stream())
        # (C x n) @ (n x cout k k) as products and ONE reduction over the images (no BLAS: its blocking would own the summation order)
        dw_c = (cvals.t().unsqueeze(2) * rs.unsqueeze(0)).sum(dim=1).view(cin - cin_img, cout, k, k).transpose(0, 1)
        return torch.cat([dw_img, dw_c], dim=1).contiguous(), db
    dw = torch.empty((cout, cin, k, k), device=gy.device, dtype=torch.float32)
    ws = _wgrad_scratch(gy.device, L.load().risp_conv_wgrad_scratch_floats(cin, cout, k))
    d = L.ConvDesc(N=n, H=h, W=w, cin=cin, cout=cout, ksize=k, load_mode=load, cin_img=cin_img, epilogue=0, add_c=0,
                   x=_p(x), wpack=None, bias=None, cvals=_p(cvals), add=None, mask=None, y=None)
    L.call('risp_conv2d_wgrad', C.byref(d), _p(gy), _p(dw), _p(ws), ws.numel(), _stream())
    sums = torch.empty((n, cout), device=gy.device, dtype=torch.float32)
    L.call('risp_plane_sums', _p(gy), _p(sums), n, cout, 0, cout, h * w, _stream())
    return dw, sums.sum(dim=0)


class PackCache:
    """Re-pack a module's conv weights only when a parameter changed (version counter / storage)."""

    def __init__(self):
        self._key, self._packs, self._params, self._age = None, None, None, 0

    def get(self, module, build):
        # the module tree is walked once and then every 64th call (module.parameters() recurses through named_modules: 35 us for a
        # Path-Restore proxy, 150 times per search iteration - a quarter of the host time of the reference's shipped 48 x 48 geometry);
        # in between the SAME Parameter objects are watched: in-place updates (optimizer steps, load_state_dict, .to()) move their
        # version counter or storage
        self._age -= 1
        if self._params is None or self._age < 0:
            self._params, self._age = list(module.parameters()), 64
        key = tuple((p.data_ptr(), p._version) for p in self._params)              # (a device move changes data_ptr)
        if key != self._key:
            self._params = list(module.parameters())
            key = tuple((p.data_ptr(), p._version) for p in self._params)
            self._packs, self._key = build(), key
        return self._packs


# --------------------------------------------------------------------------- Path-Restore (14 layers)
class _Path14l(torch.autograd.Function):
    """``record`` (a dict, or None): step-level reuse.  The first call stores the output and the saved activations in it;
    a later call with the same record returns them without launching anything - the caller guarantees that input and
    weights are unchanged (DartsModel: forwards #1, #3, #4 of an iteration see the same batch, darts_model.py:182-222,
    270-324).  Each call is its own autograd node, so every backward pass runs with its own upstream gradient."""

    @staticmethod
    def forward(ctx, x, packs, bayer, infer, record=None):
        x = _dev(x, 'img')
        n = x.shape[0]
        h, w = (x.shape[2] // 2, x.shape[3] // 2) if bayer else (x.shape[2], x.shape[3])
        ctx.packs, ctx.bayer, ctx.dims = packs, bayer, (n, h, w)
        if record is not None and 'y' in record:
            ctx.save_for_backward(*record['saved'])
            return record['y'].detach()
        first, blocks, last = packs
        r = conv(x, first, n, h, w, load=LOAD_UNSHUFFLE2 if bayer else LOAD_PLAIN, epi=EPI_RELU, infer=infer)
        saved = [r]
        for c1, c2 in blocks:
            u = conv(r, c1, n, h, w, epi=EPI_RELU, infer=infer)
            r = conv(u, c2, n, h, w, epi=EPI_ADD | EPI_RELU, add=r, add_c=64, infer=infer)
            saved += [u, r]
        # the 64 -> 4 / 3 tail runs on the direct small-cout kernel (the matrix-core kernel pads cout to 32)
        y = conv_small(r, last.small, n, h, w, epi=EPI_SHUFFLE2 if bayer else 0, infer=infer)
        ctx.save_for_backward(*saved)
        if record is not None:
            record['y'], record['saved'] = y.detach(), saved
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        first, blocks, last = ctx.packs
        n, h, w = ctx.dims
        gy = _dev(gy, 'grad')
        # gradient at the pre-activation of the last ReLU output r6
        g = conv(gy, last, n, h, w, transpose=True, load=LOAD_UNSHUFFLE2 if ctx.bayer else LOAD_PLAIN,
                 epi=EPI_MASK, mask=saved[-1])
        for k in range(len(blocks) - 1, -1, -1):
            c1, c2 = blocks[k]
            u, r_in = saved[1 + 2 * k], saved[2 * k]
            gu = conv(g, c2, n, h, w, transpose=True, epi=EPI_MASK, mask=u)
            g = conv(gu, c1, n, h, w, transpose=True, epi=EPI_ADD | EPI_MASK, add=g, add_c=64, mask=r_in)
        gx = conv_small(g, first.small_bwd, n, h, w, epi=EPI_SHUFFLE2 if ctx.bayer else 0) if ctx.needs_input_grad[0] else None
        return gx, None, None, None, None


def build_path14l_packs(seq, flip_bgr):
    """seq = module.path_restore_14l (conv_first, Sequential(6 blocks), ReLU, conv_last[, PixelShuffle])."""
    first_w, first_b = seq[0].weight, seq[0].bias
    last_w, last_b = seq[3].weight, seq[3].bias
    if flip_bgr:  # x[:, [2,1,0]] on the way in and out == permuted weights (path_14l_bgr_arch.py:59,84)
        first_w = first_w.detach().flip(1)
        last_w, last_b = last_w.detach().flip(0), last_b.detach().flip(0)
    blocks = [(PackedConv(b.basic[1].weight, b.basic[1].bias), PackedConv(b.basic[3].weight, b.basic[3].bias))
              for b in seq[1]]
    first, last = PackedConv(first_w, first_b), PackedConv(last_w, last_b)
    first.small_bwd = SmallConv(first_w, None, transpose=True, keep=first_w.shape[1])    # 64 -> 4 / 3, backward-data
    last.small = SmallConv(last_w, last_b)                                                # 64 -> 4 / 3
    return first, blocks, last


def path14l(x, packs, bayer, record=None):
    # autograd globally off (test.py / test_split.py / serving): nothing in the process will differentiate through
    # these activations, so the 3x3 layers may take the F(4,3) form.  With autograd on - even for an op whose own
    # input needs no gradient - the F(2,3) form keeps the rounding, and with it the ReLU masks of every downstream
    # op, as close to the reference's arithmetic as the direct kernel does.
    return _Path14l.apply(x, packs, bayer, not torch.is_grad_enabled(), record)


# --------------------------------------------------------------------------- SRCNN (residual proxy)
class _SrcnnRes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pv, packs):
        x = _dev(x, 'img')
        n, _, h, w = x.shape
        c1, c2, c3 = packs
        P = c1.cin - 12
        pv = _dev(pv, 'params') if P else None
        if P and (pv.dim() != 2 or pv.shape[0] != n or pv.shape[1] != P):
            raise ValueError('SRCNNRes: param_vec must be (N=%d,%d), got %s' % (n, P, tuple(pv.shape)))
        stats, arg = channel_stats(x)
        cvals = torch.empty((n, 9 + P), device=x.device, dtype=torch.float32)
        L.call('risp_srcnn_cvals', _p(stats), _p(pv), _p(cvals), n, P, h * w, _stream())
        t1 = conv(x, c1, n, h, w, load=LOAD_CONSTCH, cin_img=3, cvals=cvals, epi=EPI_RELU)
        t2 = conv(t1, c2, n, h, w, epi=EPI_RELU)
        y = conv(t2, c3, n, h, w, epi=EPI_ADD, add=x, add_c=3)
        ctx.save_for_backward(t1, t2, arg)
        ctx.packs, ctx.dims = packs, (n, h, w, P)
        return y

    @staticmethod
    def backward(ctx, gy):
        t1, t2, arg = ctx.saved_tensors
        c1, c2, c3 = ctx.packs
        n, h, w, P = ctx.dims
        gy = _dev(gy, 'grad')
        g2 = conv(gy, c3, n, h, w, transpose=True, epi=EPI_MASK, mask=t2)
        g1 = conv(g2, c2, n, h, w, transpose=True, epi=EPI_MASK, mask=t1)
        gf = conv(g1, c1, n, h, w, transpose=True, epi=EPI_ADD, add=gy, add_c=3)      # (N,12+P,H,W)
        sums = []
        for c0, nc in ((3, 3), (6, 3), (9, 3), (12, P)):
            if nc == 0:
                sums.append(None)
                continue
            s = torch.empty((n, nc), device=gy.device, dtype=torch.float32)
            L.call('risp_plane_sums', _p(gf), _p(s), n, 12 + P, c0, nc, h * w, _stream())
            sums.append(s)
        gx = gf[:, :3].contiguous()
        L.call('risp_stats_bwd', _p(gx), _p(sums[0]), _p(sums[1]), _p(sums[2]), _p(arg), n * 3, h * w, _stream())
        return gx, sums[3], None


class SrcnnResFold:
    """Weight-only tables that take the 9+P broadcast planes (srcnn_res_arch.py:41-46: per-image constants inside
    the image, zero in the padding) out of the 9x9 first layer, and the small-cout forms of the 3-channel ends:

      img     the layer restricted to its 3 image channels (matrix-core pack)
      rcase   (9+P, 64*81): rcase[c, co, i, j] = sum of w[co, 3+c] over the taps inside the image for border case
              (i, j) - forward, the constants contribute ``cvals @ rcase`` per (image, cout, case): RISP_EPI_CASEBIAS
      wconst  (64*81, 9+P): backward, d loss / d constant = risp_rect_sums(upstream) @ wconst
      bwd_img backward-data of the layer for the 3 image channels (direct kernel)
      tail    the 5x5 32->3 last layer (direct kernel)
    """

    def __init__(self, conv1, conv3):
        w1 = _dev(conv1.weight.detach(), 'weight')                    # (64, 12+P, 9, 9)
        self.k = w1.shape[2]
        self.img = PackedConv(w1[:, :3].contiguous(), conv1.bias)
        self.rcase, self.wconst = srcnn_fold_tables(w1)
        self.bwd_img = SmallConv(w1, None, transpose=True, keep=3)
        self.tail = SmallConv(conv3.weight, conv3.bias)


def _srcnn_fold_ok(h, w, k=9):
    return h >= k - 1 and k - 1 <= w <= 8192


class _SrcnnResFolded(torch.autograd.Function):
    """SRCNNRes with the broadcast planes folded out of the first layer (see SrcnnResFold): the 9x9 layer runs
    over 3 channels instead of 12+P, its backward-data over 3 output channels instead of 32 padded ones."""

    @staticmethod
    def forward(ctx, x, pv, packs, infer=False):
        x = _dev(x, 'img')
        n, _, h, w = x.shape
        c1, c2, c3 = packs
        fold = c1.fold
        P = c1.cin - 12
        pv = _dev(pv, 'params') if P else None
        stats, arg = channel_stats(x)
        table = torch.empty((n, fold.rcase.shape[1]), device=x.device, dtype=torch.float32)
        L.call('risp_srcnn_case_table', _p(stats), _p(pv), _p(fold.rcase), _p(table), n, P, h * w, fold.rcase.shape[1],
               _stream())                                                # (N, 64*81) border-case constants = cvals @ rcase
        t1 = conv(x, fold.img, n, h, w, epi=EPI_RELU | EPI_CASEBIAS, cvals=table, infer=infer)
        t2 = conv(t1, c2, n, h, w, epi=EPI_RELU)
        y = conv_small(t2, fold.tail, n, h, w, epi=EPI_ADD, add=x, add_c=3, infer=infer)
        ctx.save_for_backward(t1, t2, arg)
        ctx.packs, ctx.dims = packs, (n, h, w, P)
        return y

    @staticmethod
    def backward(ctx, gy):
        t1, t2, arg = ctx.saved_tensors
        c1, c2, c3 = ctx.packs
        fold = c1.fold
        n, h, w, P = ctx.dims
        gy = _dev(gy, 'grad')
        g2 = conv(gy, c3, n, h, w, transpose=True, epi=EPI_MASK, mask=t2)
        g1 = conv(g2, c2, n, h, w, transpose=True, epi=EPI_MASK, mask=t1)
        ts = []
        gx = conv_small(g1, fold.bwd_img, n, h, w, epi=EPI_ADD, add=gy, add_c=3, tile_sums=ts)     # image channels + residual path
        rs = torch.empty((n, c1.cout * fold.k * fold.k), device=gy.device, dtype=torch.float32)
        rect_sums(g1, rs, n, c1.cout, h, w, fold.k, ts)
        gconst = torch.empty((n, 9 + P), device=gy.device, dtype=torch.float32)     # min, mean, max planes, then params
        L.call('risp_srcnn_const_grad', _p(rs), _p(fold.wconst), _p(gconst), n, rs.shape[1], 9 + P, _stream())
        row = gconst.shape[1]                                           # columns 0-2 / 3-5 / 6-8 of each row, read in place
        gb = gconst.data_ptr()
        L.call('risp_stats_bwd_rows', _p(gx), C.c_void_p(gb), C.c_void_p(gb + 12), C.c_void_p(gb + 24), _p(arg), n, 3, h * w,
               row, _stream())
        return gx, (gconst[:, 9:] if P else None), None, None


class _SrcnnResTrain(torch.autograd.Function):
    """SRCNNRes whose six conv tensors are autograd inputs: backward also returns their gradients
    (risp_conv2d_wgrad).  Used only while a proxy is being fine-tuned against its classical teacher."""

    @staticmethod
    def forward(ctx, x, pv, packs, w1, b1, w2, b2, w3, b3):
        x = _dev(x, 'img')
        n, _, h, w = x.shape
        c1, c2, c3 = packs
        P = c1.cin - 12
        pv = _dev(pv, 'params') if P else None
        stats, arg = channel_stats(x)
        cvals = torch.empty((n, 9 + P), device=x.device, dtype=torch.float32)
        L.call('risp_srcnn_cvals', _p(stats), _p(pv), _p(cvals), n, P, h * w, _stream())
        t1 = conv(x, c1, n, h, w, load=LOAD_CONSTCH, cin_img=3, cvals=cvals, epi=EPI_RELU)
        t2 = conv(t1, c2, n, h, w, epi=EPI_RELU)
        y = conv(t2, c3, n, h, w, epi=EPI_ADD, add=x, add_c=3)
        ctx.save_for_backward(x, cvals, t1, t2, arg)
        ctx.packs, ctx.dims = packs, (n, h, w, P)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, cvals, t1, t2, arg = ctx.saved_tensors
        c1, c2, c3 = ctx.packs
        n, h, w, P = ctx.dims
        gy = _dev(gy, 'grad')
        g2 = conv(gy, c3, n, h, w, transpose=True, epi=EPI_MASK, mask=t2)
        g1 = conv(g2, c2, n, h, w, transpose=True, epi=EPI_MASK, mask=t1)
        dw3, db3 = conv_wgrad(t2, gy, c3.cin, c3.cout, c3.k, n, h, w)
        dw2, db2 = conv_wgrad(t1, g2, c2.cin, c2.cout, c2.k, n, h, w)
        dw1, db1 = conv_wgrad(x, g1, c1.cin, c1.cout, c1.k, n, h, w, load=LOAD_CONSTCH, cin_img=3, cvals=cvals)
        gx = gpv = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gf = conv(g1, c1, n, h, w, transpose=True, epi=EPI_ADD, add=gy, add_c=3)
            sums = []
            for c0, nc in ((3, 3), (6, 3), (9, 3), (12, P)):
                if nc == 0:
                    sums.append(None)
                    continue
                s_ = torch.empty((n, nc), device=gy.device, dtype=torch.float32)
                L.call('risp_plane_sums', _p(gf), _p(s_), n, 12 + P, c0, nc, h * w, _stream())
                sums.append(s_)
            gx = gf[:, :3].contiguous()
            L.call('risp_stats_bwd', _p(gx), _p(sums[0]), _p(sums[1]), _p(sums[2]), _p(arg), n * 3, h * w, _stream())
            gpv = sums[3]
        return gx, gpv, None, dw1, db1, dw2, db2, dw3, db3


def srcnn_res(x, pv, packs, train_module=None):
    if train_module is not None:
        seq = train_module.srcnn
        return _SrcnnResTrain.apply(x, pv, packs, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias,
                                    seq[4].weight, seq[4].bias)
    if getattr(packs[0], 'fold', None) is not None and _srcnn_fold_ok(x.shape[2], x.shape[3]):
        return _SrcnnResFolded.apply(x, pv, packs, not torch.is_grad_enabled())
    return _SrcnnRes.apply(x, pv, packs)


# --------------------------------------------------------------------------- SRCNN demosaic proxy
class _SrcnnDemosaic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, packs, infer=False):
        x = _dev(x, 'img')
        n, h, w = x.shape[0], x.shape[2] // 2, x.shape[3] // 2
        c1, c2, c3 = packs
        t1 = conv(x, c1, n, h, w, load=LOAD_UNSHUFFLE2, epi=EPI_RELU, infer=infer)
        t2 = conv(t1, c2, n, h, w, epi=EPI_RELU)
        if getattr(c3, 'small', None) is not None:           # 5x5 32 -> 12 + PixelShuffle: direct small-cout kernel
            y = conv_small(t2, c3.small, n, h, w, epi=EPI_SHUFFLE2, infer=infer)
        else:
            y = conv(t2, c3, n, h, w, epi=EPI_SHUFFLE2)
        ctx.save_for_backward(t1, t2)
        ctx.packs, ctx.dims = packs, (n, h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        t1, t2 = ctx.saved_tensors
        c1, c2, c3 = ctx.packs
        n, h, w = ctx.dims
        gy = _dev(gy, 'grad')
        g2 = conv(gy, c3, n, h, w, transpose=True, load=LOAD_UNSHUFFLE2, epi=EPI_MASK, mask=t2)
        g1 = conv(g2, c2, n, h, w, transpose=True, epi=EPI_MASK, mask=t1)
        if getattr(c1, 'small_bwd', None) is not None:      # 64 -> 4 through PixelShuffle: direct small-cout kernel
            gx = conv_small(g1, c1.small_bwd, n, h, w, epi=EPI_SHUFFLE2)
        else:
            gx = conv(g1, c1, n, h, w, transpose=True, epi=EPI_SHUFFLE2)
        return gx, None, None


def srcnn_demosaic(x, packs):
    return _SrcnnDemosaic.apply(x, packs, not torch.is_grad_enabled())


def build_srcnn_packs(seq, residual=False):
    """seq = module.srcnn (conv, ReLU, conv, ReLU, conv[, PixelShuffle]).  ``residual``: the SRCNNRes form, whose
    first layer also gets the folded tables (SrcnnResFold)."""
    packs = tuple(PackedConv(seq[i].weight, seq[i].bias) for i in (0, 2, 4))
    packs[0].fold = SrcnnResFold(seq[0], seq[4]) if residual and seq[4].weight.shape[0] <= 4 else None
    if not residual and seq[0].weight.shape[1] == 4:       # SRCNNDemosaic: backward-data of the 4 -> 64 first layer
        packs[0].small_bwd = SmallConv(seq[0].weight, None, transpose=True, keep=4)
    if not residual and seq[4].weight.shape[0] <= 12 and seq[4].weight.shape[2] in (3, 5):
        packs[2].small = SmallConv(seq[4].weight, seq[4].bias)
    return packs


# --------------------------------------------------------------------------- grouped launches: one launch per LAYER
# The same-geometry proxies of a super-net slot - the 8 SRCNNRes of an sRGB slot, the 2 SRCNNDemosaic of the demosaic slot
# (super_prune_fifteen_demos_four_bayer_two.py:35-52, looped at :183-212) - run every layer as ONE launch: the members'
# images are stacked along N, the member index sits in the grid and selects the weights (include/risp.h: group_n).  At the
# 4-image per-GPU batch of the 8-GPU search a single member's launch is one or two rounds of workgroups; eight of them
# fill the chip.  GROUP_LAUNCH=0 keeps the data flow (stacked buffers, one summed input gradient) but issues the members'
# launches one by one - same kernels, same per-image arithmetic, bit-identical results (tests/test_gpu_group.py).
GROUP_LAUNCH = os.environ.get('RISP_GROUP_LAUNCH', '1') != '0'
LAUNCHES = None           # tests / tools set this to [0] to count C-ABI launches of the grouped paths


class _Stacked:
    """Packs of G same-shape layers stacked along dimension 0 (what ``conv`` / ``conv_small`` take with ``group=``); a pack is
    stacked when a launch first asks for it (``PackedConv`` builds its packs on first use)."""

    def __init__(self, members, tensors, scalars):
        for a in scalars:
            vals = {getattr(m, a) for m in members}
            if len(vals) != 1:
                raise ValueError('grouped layer: members disagree on %s: %s' % (a, sorted(vals)))
            setattr(self, a, vals.pop())
        self._tensors = tuple(tensors)
        self.wino43_fwd = self.wino43_bwd = None
        self.members = list(members)

    def __getattr__(self, name):
        if name.startswith('_') or name not in self.__dict__.get('_tensors', ()):
            raise AttributeError(name)
        ts = [getattr(m, name, None) for m in self.members]
        val = torch.stack(ts) if all(t is not None for t in ts) else None
        self.__dict__[name] = val
        return val


def stack_packed(pcs):
    return _Stacked(pcs, ('fwd', 'bwd', 'bias', 'k3', 'wino45_fwd', 'wino45_bwd', 'f16x2_fwd', 'f16x2_bwd', 'thin5_fwd', 'thin5_bwd', 'toep_first', 'w32'),
                    ('cin', 'cout', 'k'))


def stack_small(scs):
    return _Stacked(scs, ('wpack', 'bias', 'toep', 'tapout', 'narrow3'), ('cin', 'cout', 'k'))


def _stack_grads(gys, like):
    """The members' upstream gradients as ONE (G*N, ...) tensor: in place when they already are consecutive slices of
    one buffer (functional._Mix.backward allocates them that way), else copied; None entries count as zeros."""
    g0 = gys[0]
    if all(g is not None and g.is_contiguous() and g.dtype == torch.float32 for g in gys):
        step = g0.numel() * 4
        if all(g.data_ptr() == g0.data_ptr() + k * step for k, g in enumerate(gys)):
            try:
                return g0.as_strided((len(gys) * g0.shape[0],) + tuple(g0.shape[1:]), g0.stride())
            except RuntimeError:
                pass                                    # adjacent by coincidence, not one storage
    return torch.cat([_dev(g, 'grad') if g is not None else torch.zeros_like(like) for g in gys])


class SrcnnResGroup:
    """Stacked packs of G SRCNNRes members with folded first layers (see SrcnnResFold)."""

    def __init__(self, packs_list):
        self.packs_list = list(packs_list)
        self.G = len(packs_list)
        folds = [p[0].fold for p in packs_list]
        if any(f is None for f in folds):
            raise ValueError('grouped SRCNNRes: every member needs the folded first layer')
        self.P = [p[0].cin - 12 for p in packs_list]
        self.k = folds[0].k
        self.img = stack_packed([f.img for f in folds])            # 9x9 3 -> 64
        self.c2 = stack_packed([p[1] for p in packs_list])         # 5x5 64 -> 32
        self.c3 = stack_packed([p[2] for p in packs_list])         # 5x5 32 -> 3 (its backward-data runs on the matrix cores)
        self.tail = stack_small([f.tail for f in folds])           # 5x5 32 -> 3 forward, direct kernel
        self.bwd_img = stack_small([f.bwd_img for f in folds])     # 9x9 64 -> 3 backward-data, direct kernel
        self.rcase = [f.rcase for f in folds]
        self.wconst = [f.wconst for f in folds]
        self.M = self.rcase[0].shape[1]

    def desc(self, n, hw, pvs=None):
        d = L.SrcnnGroupDesc()
        d.G, d.N, d.HW, d.M = self.G, n, hw, self.M
        for g in range(self.G):
            d.P[g] = self.P[g]
            d.rcase[g], d.wconst[g] = self.rcase[g].data_ptr(), self.wconst[g].data_ptr()
            d.pv[g] = pvs[g].data_ptr() if pvs is not None and self.P[g] else None
        return d


def _count(k=1):
    if LAUNCHES is not None:
        LAUNCHES[0] += k


class _SrcnnResGroupFn(torch.autograd.Function):
    """G SRCNNRes members on one shared input: outputs (y_0, ..., y_{G-1}), each (N,3,H,W) - slices of one buffer."""

    @staticmethod
    def forward(ctx, x, gp, grouped, flags, *pvs):
        x = _dev(x, 'img')
        n, _, h, w = x.shape
        G = gp.G
        pvs = [_dev(p, 'params') if gp.P[g] else None for g, p in enumerate(pvs)]
        for g, p in enumerate(pvs):
            if gp.P[g] and (p.dim() != 2 or p.shape[0] != n or p.shape[1] != gp.P[g]):
                raise ValueError('SRCNNRes group member %d: param_vec must be (N=%d,%d), got %s' % (g, n, gp.P[g], tuple(p.shape)))
        stats, arg = channel_stats(x)                   # of the shared input: once for the whole group
        table = torch.empty((G * n, gp.M), device=x.device, dtype=torch.float32)
        L.call('risp_srcnn_case_table_group', _p(stats), C.byref(gp.desc(n, h * w, pvs)), _p(table), _stream())
        if grouped:
            t1 = conv(x, gp.img, n, h, w, epi=EPI_RELU | EPI_CASEBIAS, cvals=table, group=(G, L.GROUP_SHARED_X))
            t2 = conv(t1, gp.c2, n, h, w, epi=EPI_RELU, group=(G, 0))
            y = conv_small(t2, gp.tail, n, h, w, epi=EPI_ADD, add=x, add_c=3, group=(G, L.GROUP_SHARED_ADD))
            _count(5 + 1)
        else:
            dev = dict(device=x.device, dtype=torch.float32)
            t1, t2 = torch.empty((G * n, 64, h, w), **dev), torch.empty((G * n, 32, h, w), **dev)
            y = torch.empty((G * n, 3, h, w), **dev)
            split = _small_split(t2, gp.tail, G * n, h, w, EPI_ADD, 3)
            for g, (c1, c2, c3) in enumerate(gp.packs_list):
                s = slice(g * n, (g + 1) * n)
                conv(x, c1.fold.img, n, h, w, epi=EPI_RELU | EPI_CASEBIAS, cvals=table[s], out=t1[s])
                conv(t1[s], c2, n, h, w, epi=EPI_RELU, out=t2[s])
                conv_small(t2[s], c1.fold.tail, n, h, w, epi=EPI_ADD, add=x, add_c=3, out=y[s], split=split,
                           seg_rows=tapout_seg(G * n, h, w, False))
            _count(3 + 3 * G)
        ctx.save_for_backward(t1, t2, arg, x)
        ctx.gp, ctx.dims, ctx.grouped, ctx.flags = gp, (n, h, w), grouped, flags
        return tuple(y.view(G, n, 3, h, w).unbind(0))

    @staticmethod
    def backward(ctx, *gys):
        t1, t2, arg, x = ctx.saved_tensors
        gp, (n, h, w), G = ctx.gp, ctx.dims, ctx.gp.G
        gy = _stack_grads(gys, x)                       # (G*N,3,H,W)
        dev = dict(device=gy.device, dtype=torch.float32)
        # the caller may know that nobody will consume this node's input gradient (DartsModel.virtual_step asks for the
        # parameter gradients only and the slots below the first parametrised one have none: flags['skip_gx']); autograd
        # cannot tell a custom Function that - needs_input_grad is a property of the forward graph
        want_gx = not (ctx.flags is not None and ctx.flags.get('skip_gx'))
        ts, tsm = [], []                                # per-tile plane sums out of the backward-data launch(es): see rect_sums
        if ctx.grouped and not want_gx:
            g2 = conv(gy, gp.c3, n, h, w, transpose=True, epi=EPI_MASK, mask=t2, group=(G, 0))
            g1 = conv(g2, gp.c2, n, h, w, transpose=True, epi=EPI_MASK, mask=t1, group=(G, 0))
            gxs = None
            _count(2)
        elif ctx.grouped:
            g2 = conv(gy, gp.c3, n, h, w, transpose=True, epi=EPI_MASK, mask=t2, group=(G, 0))
            g1 = conv(g2, gp.c2, n, h, w, transpose=True, epi=EPI_MASK, mask=t1, group=(G, 0))
            gxs = conv_small(g1, gp.bwd_img, n, h, w, epi=EPI_ADD, add=gy, add_c=3, group=(G, 0), tile_sums=ts)
            _count(3)
        else:
            g2, g1 = torch.empty((G * n, 32, h, w), **dev), torch.empty((G * n, 64, h, w), **dev)
            gxs = torch.empty((G * n, 3, h, w), **dev)
            split = _small_split(g1, gp.bwd_img, G * n, h, w, EPI_ADD, 3)
            for g, (c1, c2, c3) in enumerate(gp.packs_list):
                s = slice(g * n, (g + 1) * n)
                conv(gy[s], c3, n, h, w, transpose=True, epi=EPI_MASK, mask=t2[s], out=g2[s])
                conv(g2[s], c2, n, h, w, transpose=True, epi=EPI_MASK, mask=t1[s], out=g1[s])
                conv_small(g1[s], c1.fold.bwd_img, n, h, w, epi=EPI_ADD, add=gy[s], add_c=3, out=gxs[s], split=split, tile_sums=tsm,
                           seg_rows=tapout_seg(G * n, h, w, False))
            if tsm:
                ts.append(torch.cat(tsm))
            _count(3 * G)
        rs = torch.empty((G * n, gp.M), **dev)
        rect_sums(g1, rs, G * n, 64, h, w, gp.k, ts)
        row = 9 + max(gp.P)
        gconst = torch.empty((G * n, row), **dev)       # min, mean, max planes, then the members' parameters
        L.call('risp_srcnn_const_grad_group', _p(rs), C.byref(gp.desc(n, h * w)), _p(gconst), row, _stream())
        gx = None
        if gxs is not None:
            gx = torch.empty((n, 3, h, w), **dev)
            L.call('risp_group_sum', _p(gxs), _p(gx), G, n, 3, h * w, _p(gconst), row, _p(arg), _stream())
        _count(3 if gxs is not None else 2)
        gpvs = tuple(gconst[g * n:(g + 1) * n, 9:9 + gp.P[g]] if gp.P[g] else None for g in range(G))
        return (gx, None, None, None) + gpvs


def _small_split(x, sc, n_total, h, w, epi, add_c):
    """the channel split ``conv_small`` would take for the grouped launch (so that the per-member form uses the same); 0 = the
    grouped launch runs on ``risp_conv2d_toep``"""
    if _tapout_ok(sc, h, w, epi):
        if tapout_grid_ok(n_total, h, w):
            return 0
    elif _toep_ok(sc, h, w) and toep_grid_ok(n_total, h, w):
        return 0
    d = L.ConvDesc(N=n_total, H=h, W=w, cin=sc.cin, cout=sc.cout, ksize=sc.k, load_mode=LOAD_PLAIN, cin_img=0, epilogue=epi,
                   add_c=add_c, x=_p(x), wpack=_p(sc.wpack), bias=None, cvals=None, add=None, mask=None, y=None)
    return L.load().risp_conv_small_groups(C.byref(d))


def srcnn_res_group(x, pvs, packs_list, cache):
    """[SRCNNRes_g(x, pvs[g]) for g] with one launch per layer.  ``cache``: a dict owned by the caller (the slot) that keeps
    the stacked packs while the members' packs are unchanged."""
    key = tuple(id(p) for p in packs_list)
    gp = cache.get('srcnn_res')
    if gp is None or gp[0] != key:
        gp = cache['srcnn_res'] = (key, SrcnnResGroup(packs_list))
    return list(_SrcnnResGroupFn.apply(x, gp[1], GROUP_LAUNCH, cache, *pvs))


class SrcnnDemosaicGroup:
    """Stacked packs of G SRCNNDemosaic members (srcnn_demosaic_arch.py:14-55)."""

    def __init__(self, packs_list):
        self.packs_list = list(packs_list)
        self.G = len(packs_list)
        if any(getattr(p[0], 'small_bwd', None) is None or getattr(p[2], 'small', None) is None for p in packs_list):
            raise ValueError('grouped SRCNNDemosaic: members need the direct-kernel forms of their first and last layer')
        self.c1 = stack_packed([p[0] for p in packs_list])         # 9x9 4 -> 64 (space-to-depth load)
        self.c2 = stack_packed([p[1] for p in packs_list])         # 1x1 64 -> 32
        self.c3 = stack_packed([p[2] for p in packs_list])         # 5x5 32 -> 12: backward-data on the matrix cores
        self.tail = stack_small([p[2].small for p in packs_list])  # 5x5 32 -> 12 + PixelShuffle, direct kernel
        self.bwd_first = stack_small([p[0].small_bwd for p in packs_list])   # 9x9 64 -> 4 through PixelShuffle


class _SrcnnDemosaicGroupFn(torch.autograd.Function):
    """G SRCNNDemosaic members on one shared mosaic: outputs (y_0, ..., y_{G-1}), each (N,3,2h,2w)."""

    @staticmethod
    def forward(ctx, x, gp, grouped, infer, record=None):
        x = _dev(x, 'img')
        n, h, w = x.shape[0], x.shape[2] // 2, x.shape[3] // 2
        G = gp.G
        ctx.gp, ctx.dims, ctx.grouped = gp, (n, h, w), grouped
        if record is not None and 'y' in record:           # step-level reuse, see _Path14l
            ctx.save_for_backward(*record['saved'])
            return tuple(record['y'].detach().view(G, n, 3, 2 * h, 2 * w).unbind(0))
        if grouped:
            t1 = conv(x, gp.c1, n, h, w, load=LOAD_UNSHUFFLE2, epi=EPI_RELU, group=(G, L.GROUP_SHARED_X))
            t2 = conv(t1, gp.c2, n, h, w, epi=EPI_RELU, group=(G, 0))
            y = conv_small(t2, gp.tail, n, h, w, epi=EPI_SHUFFLE2, infer=infer, group=(G, 0))
            _count(3)
        else:
            dev = dict(device=x.device, dtype=torch.float32)
            t1, t2 = torch.empty((G * n, 64, h, w), **dev), torch.empty((G * n, 32, h, w), **dev)
            y = torch.empty((G * n, 3, 2 * h, 2 * w), **dev)
            split = _small_split(t2, gp.tail, G * n, h, w, EPI_SHUFFLE2, 0)
            for g, (c1, c2, c3) in enumerate(gp.packs_list):
                s = slice(g * n, (g + 1) * n)
                conv(x, c1, n, h, w, load=LOAD_UNSHUFFLE2, epi=EPI_RELU, out=t1[s])
                conv(t1[s], c2, n, h, w, epi=EPI_RELU, out=t2[s])
                conv_small(t2[s], c3.small, n, h, w, epi=EPI_SHUFFLE2, infer=infer, out=y[s], split=split)
            _count(3 * G)
        ctx.save_for_backward(t1, t2, x)
        if record is not None:
            record['y'], record['saved'] = y.detach(), (t1, t2, x)
        return tuple(y.view(G, n, 3, 2 * h, 2 * w).unbind(0))

    @staticmethod
    def backward(ctx, *gys):
        t1, t2, x = ctx.saved_tensors
        gp, (n, h, w), G = ctx.gp, ctx.dims, ctx.gp.G
        like = torch.empty(0)
        gy = _stack_grads(gys, gys[0] if gys[0] is not None else like)
        dev = dict(device=gy.device, dtype=torch.float32)
        if ctx.grouped:
            g2 = conv(gy, gp.c3, n, h, w, transpose=True, load=LOAD_UNSHUFFLE2, epi=EPI_MASK, mask=t2, group=(G, 0))
            g1 = conv(g2, gp.c2, n, h, w, transpose=True, epi=EPI_MASK, mask=t1, group=(G, 0))
            gxs = conv_small(g1, gp.bwd_first, n, h, w, epi=EPI_SHUFFLE2, group=(G, 0))
            _count(3)
        else:
            g2, g1 = torch.empty((G * n, 32, h, w), **dev), torch.empty((G * n, 64, h, w), **dev)
            gxs = torch.empty((G * n, 1, 2 * h, 2 * w), **dev)
            split = _small_split(g1, gp.bwd_first, G * n, h, w, EPI_SHUFFLE2, 0)
            for g, (c1, c2, c3) in enumerate(gp.packs_list):
                s = slice(g * n, (g + 1) * n)
                conv(gy[s], c3, n, h, w, transpose=True, load=LOAD_UNSHUFFLE2, epi=EPI_MASK, mask=t2[s], out=g2[s])
                conv(g2[s], c2, n, h, w, transpose=True, epi=EPI_MASK, mask=t1[s], out=g1[s])
                conv_small(g1[s], c1.small_bwd, n, h, w, epi=EPI_SHUFFLE2, out=gxs[s], split=split)
            _count(3 * G)
        gx = torch.empty_like(x)
        L.call('risp_group_sum', _p(gxs), _p(gx), G, n, 1, 4 * h * w, None, 0, None, _stream())
        _count(1)
        return gx, None, None, None, None


def srcnn_demosaic_group(x, packs_list, cache, record=None):
    """[SRCNNDemosaic_g(x) for g] with one launch per layer (see srcnn_res_group); ``record``: see _Path14l."""
    key = tuple(id(p) for p in packs_list)
    gp = cache.get('srcnn_demosaic')
    if gp is None or gp[0] != key:
        gp = cache['srcnn_demosaic'] = (key, SrcnnDemosaicGroup(packs_list))
    return list(_SrcnnDemosaicGroupFn.apply(x, gp[1], GROUP_LAUNCH, not torch.is_grad_enabled(), record))
