"""Differentiable ISP operators on top of the C ABI (``include/risp.h``).

Every function here is a ``torch.autograd.Function`` whose forward AND backward are
hand-written HIP kernels in ``libreconfigisp_hip.so``; PyTorch only owns the device
buffers and the stream.  Tensors must live on the GPU - a CPU tensor raises, there is
no fallback.

``_IMPL`` is the dispatch seam: the host logic above (registry, super-net, DARTS
step) calls ``F.<op>`` which forwards to ``_IMPL``.  The product never rebinds it;
the CPU test-suite does (tests/conftest.py) so that the host logic can be exercised
without a GPU.
"""
import ctypes as C

import torch

from . import lib as L

OP_SKIP, OP_DEMOSAIC_NEAREST, OP_WB_MANUAL, OP_GAMMA, OP_GTM_MANUAL, OP_WB_QUADRATIC, OP_GAIN3 = range(7)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """The current HIP stream of the current device as a void* (the raw-handle getter is ~10x cheaper than building
    a torch.cuda.Stream object, which matters for launches of ~50 us)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, what='tensor'):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('reconfigisp_amd: %s must be a CUDA/HIP tensor (got %s); the ops are GPU-only '
                           'and there is no CPU fallback' % (what, getattr(t, 'device', type(t))))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _grad_scratch(n, device):
    """workspace of the element-wise backward kernels (per-workgroup partial parameter gradients)"""
    return torch.empty(L.load().risp_param_grad_scratch_floats(n), device=device, dtype=torch.float32)


def _check_bgr(x):
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError('expected a (N,3,H,W) BGR tensor, got %s' % (tuple(x.shape),))
    if (x.shape[2] * x.shape[3]) % 4:
        raise ValueError('H*W must be a multiple of 4, got %s' % (tuple(x.shape),))


def _check_params(p, n, width, name):
    if p.dim() != 2 or p.shape[0] != n or p.shape[1] != width:
        raise ValueError('%s: params must be (N=%d,%d), got %s' % (name, n, width, tuple(p.shape)))


class _Pointwise(torch.autograd.Function):
    """y = op(x, p) for the planar BGR ops; p is the (N,P) per-image block."""

    @staticmethod
    def forward(ctx, x, p, name, width):
        x, p = _dev(x, 'img'), _dev(p, 'params')
        _check_bgr(x)
        _check_params(p, x.shape[0], width, name)
        y = torch.empty_like(x)
        n, hw = x.shape[0], x.shape[2] * x.shape[3]
        L.call('risp_%s_fwd' % name, _p(x), _p(p), _p(y), n, hw, _stream())
        ctx.save_for_backward(x, p)
        ctx.name = name
        return y

    @staticmethod
    def backward(ctx, gy):
        x, p = ctx.saved_tensors
        gy = _dev(gy, 'grad')
        gx, gp = torch.empty_like(x), torch.empty_like(p)
        n, hw = x.shape[0], x.shape[2] * x.shape[3]
        L.call('risp_%s_bwd' % ctx.name, _p(x), _p(p), _p(gy), _p(gx), _p(gp), _p(_grad_scratch(n, x.device)), n, hw, _stream())
        return gx, gp, None, None


class _DemosaicNearest(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _dev(x, 'img')
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] % 2 or x.shape[3] % 2:
            raise ValueError('expected a (N,1,H,W) RGGB mosaic with even H, W; got %s' % (tuple(x.shape),))
        n, _, h, w = x.shape
        y = torch.empty((n, 3, h, w), device=x.device, dtype=torch.float32)
        L.call('risp_demosaic_nearest_fwd', _p(x), _p(y), n, h, w, _stream())
        ctx.shape = (n, h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        n, h, w = ctx.shape
        gy = _dev(gy, 'grad')
        gx = torch.empty((n, 1, h, w), device=gy.device, dtype=torch.float32)
        L.call('risp_demosaic_nearest_bwd', _p(gy), _p(gx), n, h, w, _stream())
        return gx


def channel_stats(x, want_arg=True):
    """(N,C,H,W) -> stats (N,C,4) = {min,sum,max,0}, arg (N,C,2) int32 (first argmin/argmax)."""
    x = _dev(x)
    n, c, h, w = x.shape
    stats = torch.empty((n, c, 4), device=x.device, dtype=torch.float32)
    arg = torch.empty((n, c, 2), device=x.device, dtype=torch.int32) if want_arg else None
    ws = torch.empty(L.load().risp_channel_stats_scratch_floats(n * c, h * w), device=x.device, dtype=torch.float32)
    L.call('risp_channel_stats', _p(x), _p(stats), _p(arg), _p(ws), n * c, h * w, _stream())
    return stats, arg


def histc01(x, bins):
    """Per-(n,c) histogram of an (N,C,H,W) tensor with torch.histc(.., bins, 0, 1) semantics -> (N, C*bins)."""
    x = _dev(x.detach())
    n, c, h, w = x.shape
    hist = torch.empty((n, c * bins), device=x.device, dtype=torch.float32)
    L.call('risp_histc', _p(x), _p(hist), n * c, h * w, bins, _stream())
    return hist


class _CondFc(torch.autograd.Function):
    """sigmoid(MLP(hist) + flat[global]) of the conditional heads (tools_origin.py:109-163): risp_cond_fc_fwd /
    risp_cond_fc_bwd.  ``hist`` carries no gradient; the gradient of the flat parameter vector is fully written."""

    @staticmethod
    def forward(ctx, hist, flat, widths):
        hist, flat = _dev(hist.detach(), 'hist'), _dev(flat, 'params')
        n, nl = hist.shape[0], len(widths) - 1
        cw = (C.c_int * len(widths))(*widths)
        row = L.load().risp_cond_fc_row_floats(cw, nl)
        acts = torch.empty((n, row), device=hist.device, dtype=torch.float32)
        out = torch.empty((n, widths[-1]), device=hist.device, dtype=torch.float32)
        L.call('risp_cond_fc_fwd', _p(hist), _p(flat), cw, nl, _p(acts), _p(out), n, _stream())
        ctx.save_for_backward(flat, acts, out)
        ctx.widths = tuple(widths)
        return out

    @staticmethod
    def backward(ctx, gout):
        flat, acts, out = ctx.saved_tensors
        widths, nl = ctx.widths, len(ctx.widths) - 1
        cw = (C.c_int * len(widths))(*widths)
        gout = _dev(gout, 'grad')
        deltas = torch.empty_like(acts)
        dflat = torch.empty_like(flat)
        L.call('risp_cond_fc_bwd', _p(flat), cw, nl, _p(acts), _p(out), _p(gout), _p(deltas), _p(dflat), flat.numel(),
               acts.shape[0], _stream())
        return None, dflat, None


def _hip_conditional_fc(img, flat, widths):
    bins = widths[0] // 3
    return _CondFc.apply(histc01(img, bins), flat, tuple(int(v) for v in widths))


class _Grayworld(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _dev(x, 'img')
        _check_bgr(x)
        n, hw = x.shape[0], x.shape[2] * x.shape[3]
        stats, _ = channel_stats(x, want_arg=False)
        gains = torch.empty((n, 3), device=x.device, dtype=torch.float32)
        L.call('risp_grayworld_gains_fwd', _p(stats), _p(gains), n, hw, _stream())
        y = torch.empty_like(x)
        L.call('risp_gain3_fwd', _p(x), _p(gains), _p(y), n, hw, _stream())
        ctx.save_for_backward(x, stats, gains)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, stats, gains = ctx.saved_tensors
        gy = _dev(gy, 'grad')
        n, hw = x.shape[0], x.shape[2] * x.shape[3]
        gx, gk, gm = torch.empty_like(x), torch.empty_like(gains), torch.empty_like(gains)
        L.call('risp_gain3_bwd', _p(x), _p(gains), _p(gy), _p(gx), _p(gk), _p(_grad_scratch(n, x.device)), n, hw, _stream())
        L.call('risp_grayworld_gains_bwd', _p(stats), _p(gk), _p(gm), n, hw, _stream())
        L.call('risp_stats_bwd', _p(gx), None, _p(gm), None, None, n * 3, hw, _stream())
        return gx


def grayworld_gains(x):
    """(N,3) gray-world gains of an (N,3,H,W) image (inference helper for the fused chain)."""
    x = _dev(x, 'img')
    _check_bgr(x)
    n, hw = x.shape[0], x.shape[2] * x.shape[3]
    stats, _ = channel_stats(x, want_arg=False)
    gains = torch.empty((n, 3), device=x.device, dtype=torch.float32)
    L.call('risp_grayworld_gains_fwd', _p(stats), _p(gains), n, hw, _stream())
    return gains


class _Mix(torch.autograd.Function):
    """y = sum_k w[k] * o_k ; w is a 1-D tensor (gradient flows to it), o_k the op outputs."""

    @staticmethod
    def forward(ctx, w, w_host, stacks, *outs):
        outs = [_dev(o) for o in outs]
        k = len(outs)
        ctx.stacks = stacks
        if w_host is None:                  # the caller usually has the values on the host already (no second D2H)
            w_host = w.detach().cpu().tolist()
        w_host = [float(v) for v in w_host]
        if len(w_host) != k or w.numel() != k:
            raise ValueError('mix: %d weights for %d operands' % (len(w_host), k))
        y = torch.empty_like(outs[0])
        L.call('risp_mix_fwd', L.ptr_array([o.data_ptr() for o in outs]), (C.c_float * k)(*w_host), k, _p(y),
               y.numel(), _stream())
        ctx.save_for_backward(*outs)
        ctx.w_host = w_host
        ctx.w_meta = (w.device, w.dtype)
        return y

    @staticmethod
    def backward(ctx, gy):
        outs = ctx.saved_tensors
        gy = _dev(gy, 'grad')
        k = len(outs)
        need = ctx.needs_input_grad[3:]
        gos = [None] * k
        for members in (ctx.stacks or ()):          # operands that feed ONE grouped launch: consecutive slices of one buffer
            if all(need[i] for i in members):
                buf = torch.empty((len(members),) + tuple(outs[members[0]].shape), device=gy.device, dtype=torch.float32)
                for j, i in enumerate(members):
                    gos[i] = buf[j]
        gos = [g if g is not None else (torch.empty_like(o) if nd else None) for g, o, nd in zip(gos, outs, need)]
        gw = torch.empty(k, device=gy.device, dtype=torch.float32)
        scratch = torch.empty(L.load().risp_mix_scratch_floats(), device=gy.device, dtype=torch.float32)
        L.call('risp_mix_bwd', L.ptr_array([o.data_ptr() for o in outs]), (C.c_float * k)(*ctx.w_host), k, _p(gy),
               L.ptr_array([g.data_ptr() if g is not None else None for g in gos]), _p(gw), _p(scratch), gy.numel(), _stream())
        return (gw.to(device=ctx.w_meta[0], dtype=ctx.w_meta[1]), None, None) + tuple(gos)


SLOT_KINDS = {'skip': OP_SKIP, 'wb_manual': OP_WB_MANUAL, 'gamma': OP_GAMMA, 'gtm_manual': OP_GTM_MANUAL,
              'wb_quadratic': OP_WB_QUADRATIC, 'grayworld': OP_GAIN3}
_SLOT_WIDTH = {OP_WB_MANUAL: 3, OP_GAMMA: 1, OP_GTM_MANUAL: 3, OP_WB_QUADRATIC: 30}


class _SlotMix(torch.autograd.Function):
    """y = sum_k w[k] o_k of one super-net slot with the element-wise operators evaluated on the fly (risp_slot_mix_fwd /
    _bwd): operand k is a materialised tensor (kinds[k] == L.SLOT_TENSOR) or op(x, block) for an element-wise kind.
    ``flat``: per operand the tensor / the (N,P) parameter block / nothing (skip, gray world)."""

    @staticmethod
    def forward(ctx, w, w_host, x, kinds, stacks, *flat):
        x = _dev(x, 'img')
        _check_bgr(x)
        n, hw = x.shape[0], x.shape[2] * x.shape[3]
        k = len(kinds)
        w_host = [float(v) for v in (w_host if w_host is not None else w.detach().cpu().tolist())]
        if len(w_host) != k or w.numel() != k or k > L.MIX_MAX:
            raise ValueError('slot_mix: %d weights for %d operands (at most %d)' % (len(w_host), k, L.MIX_MAX))
        d = L.SlotMixDesc()
        d.K, d.N, d.HW = k, n, hw
        keep, it, stats = [], iter(flat), None
        for i, kind in enumerate(kinds):
            d.kind[i], d.w[i], d.pmul[i] = kind, w_host[i], 1.0
            if kind == L.SLOT_TENSOR:
                t = _dev(next(it))
                if t.shape != x.shape:
                    raise ValueError('slot_mix: operand %d has shape %s, the slot input %s' % (i, tuple(t.shape), tuple(x.shape)))
            elif kind == OP_SKIP:
                t = None
            elif kind == OP_GAIN3:                   # gray world: gains from the statistics of x (no parameters)
                stats, _ = channel_stats(x, want_arg=False)
                t = torch.empty((n, 3), device=x.device, dtype=torch.float32)
                L.call('risp_grayworld_gains_fwd', _p(stats), _p(t), n, hw, _stream())
            else:
                t = _dev(next(it), 'params')
                _check_params(t, n, _SLOT_WIDTH[kind], 'slot operand %d' % i)
                if kind == OP_WB_MANUAL:
                    d.pmul[i] = 5.0                   # the wrapper's params * 5 (tools_origin.py:214)
            keep.append(t)
            d.ptr[i] = t.data_ptr() if t is not None else None
        y = torch.empty_like(x)
        d.x, d.y = x.data_ptr(), y.data_ptr()
        L.call('risp_slot_mix_fwd', C.byref(d), _stream())
        ctx.save_for_backward(x, stats, *[t for t in keep if t is not None])
        ctx.kinds, ctx.w_host, ctx.stacks, ctx.w_meta = tuple(kinds), w_host, stacks, (w.device, w.dtype)
        ctx.has = [t is not None for t in keep]
        return y

    @staticmethod
    def backward(ctx, gy):
        x, stats, *rest = ctx.saved_tensors
        kinds, k = ctx.kinds, len(ctx.kinds)
        gy = _dev(gy, 'grad')
        n, hw = x.shape[0], x.shape[2] * x.shape[3]
        it = iter(rest)
        keep = [next(it) if h else None for h in ctx.has]
        dev = dict(device=gy.device, dtype=torch.float32)
        d = L.SlotMixDesc()
        d.K, d.N, d.HW = k, n, hw
        d.x = x.data_ptr()
        # gradient buffers: tensor operands of one grouped launch share a stacked buffer (convnets._stack_grads)
        go = [None] * k
        for members in (ctx.stacks or ()):
            if all(kinds[i] == L.SLOT_TENSOR for i in members):
                buf = torch.empty((len(members),) + tuple(x.shape), **dev)
                for j, i in enumerate(members):
                    go[i] = buf[j]
        gp = [None] * k
        pointwise = False
        # only what the graph consumes is allocated and written (risp_slot_mix_bwd takes NULL for go[k] / gp[k]): 12 B/pixel per
        # tensor operand whose producer needs no gradient.  forward inputs: (w, w_host, x, kinds, stacks, *flat)
        need_flat, need_x = iter(ctx.needs_input_grad[5:]), ctx.needs_input_grad[2]
        for i, kind in enumerate(kinds):
            d.kind[i], d.w[i], d.pmul[i] = kind, ctx.w_host[i], 5.0 if kind == OP_WB_MANUAL else 1.0
            d.ptr[i] = keep[i].data_ptr() if keep[i] is not None else None
            if kind == L.SLOT_TENSOR:
                need = next(need_flat)
                if go[i] is None and need:
                    go[i] = torch.empty_like(x)
                d.go[i] = go[i].data_ptr() if go[i] is not None else None
            else:
                pointwise = True
                need = next(need_flat) if kind not in (OP_SKIP, OP_GAIN3) else need_x       # gray world: its gains lead back to x
                if kind != OP_SKIP and need:
                    gp[i] = torch.empty_like(keep[i])
                    d.gp[i] = gp[i].data_ptr()
        gx = torch.empty_like(x) if pointwise else None
        gw = torch.empty(k, **dev)
        scratch = torch.empty(L.load().risp_slot_mix_scratch_floats(n, hw), **dev)
        L.call('risp_slot_mix_bwd', C.byref(d), _p(gy), _p(gx), _p(gw), _p(scratch), _stream())
        for i, kind in enumerate(kinds):
            if kind == OP_GAIN3 and gp[i] is not None:        # gray world: the gains' gradient flows back through the channel means
                gm = torch.empty((n, 3), **dev)
                L.call('risp_grayworld_gains_bwd', _p(stats), _p(gp[i]), _p(gm), n, hw, _stream())
                L.call('risp_stats_bwd', _p(gx), None, _p(gm), None, None, n * 3, hw, _stream())
        grads = []
        for i, kind in enumerate(kinds):
            if kind == L.SLOT_TENSOR:
                grads.append(go[i])
            elif kind not in (OP_SKIP, OP_GAIN3):
                grads.append(gp[i])
        return (gw.to(device=ctx.w_meta[0], dtype=ctx.w_meta[1]), None, gx, None, None) + tuple(grads)


class _PruneSoftmax(torch.autograd.Function):
    """post = pruned, renormalised softmax(alpha) of one super-net slot (risp_prune_softmax_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, alpha, threshold, unavailable):
        a = _dev(alpha, 'alpha')
        k = a.numel()
        buf = torch.empty((3, k), device=a.device, dtype=torch.float32)          # probs | coef | post
        L.call('risp_prune_softmax_fwd', _p(a), _p(unavailable), float(threshold), k, _p(buf[0]), _p(buf[1]), _p(buf[2]),
               _stream())
        ctx.save_for_backward(buf)
        return buf[2]

    @staticmethod
    def backward(ctx, gpost):
        buf, = ctx.saved_tensors
        k = buf.shape[1]
        galpha = torch.empty(k, device=buf.device, dtype=torch.float32)
        L.call('risp_prune_softmax_bwd', _p(buf[0]), _p(buf[1]), _p(_dev(gpost, 'grad')), k, _p(galpha), _stream())
        return galpha, None, None


class _ParamBlocks(torch.autograd.Function):
    """blocks[k] = sigmoid(raw[k]).repeat(N, 1) for all parametrised ops of a slot in ONE launch each way."""

    @staticmethod
    def forward(ctx, n, *raws):
        raws = [_dev(r, 'params') for r in raws]
        d = L.ParamBlocksDesc()
        d.n_ops, d.N = len(raws), n
        flat = torch.empty(n * sum(r.numel() for r in raws), device=raws[0].device, dtype=torch.float32)
        blocks, at = [], 0
        for k, r in enumerate(raws):
            w = r.numel()
            blk = flat[at: at + n * w].view(n, w)
            at += n * w
            d.width[k], d.raw[k], d.block[k] = w, r.data_ptr(), blk.data_ptr()
            blocks.append(blk)
        L.call('risp_param_blocks_fwd', C.byref(d), _stream())
        ctx.save_for_backward(*raws)
        ctx.n = n
        return tuple(blocks)

    @staticmethod
    def backward(ctx, *gblocks):
        raws = ctx.saved_tensors
        d = L.ParamBlocksDesc()
        d.n_ops, d.N = len(raws), ctx.n
        keep, grads = [], []
        for k, (r, g) in enumerate(zip(raws, gblocks)):
            if g is not None and not (g.is_cuda and g.dtype == torch.float32 and g.dim() == 2 and g.stride(1) == 1
                                      and g.stride(0) >= g.shape[1]):
                g = _dev(g, 'grad')               # (a column range of a wider matrix is read in place: SRCNNRes' folded backward)
            keep.append(g)
            gr = torch.empty_like(r)
            grads.append(gr)
            d.width[k], d.raw[k], d.graw[k] = r.numel(), r.data_ptr(), gr.data_ptr()
            d.gblock[k] = g.data_ptr() if g is not None else None
            d.gstride[k] = g.stride(0) if g is not None else 0
        L.call('risp_param_blocks_bwd', C.byref(d), _stream())
        return (None,) + tuple(grads)


class _ZeroGrad(torch.autograd.Function):
    """y passes through untouched; the extra parameters join the graph with an all-zero gradient.

    Stands for the reference's 'dummy gradients' term ``zeros(x.shape) * par.sum()``
    (super_prune_fifteen_demos_four_bayer_two.py:198-201, needed by DDP) without a pass over y."""

    @staticmethod
    def forward(ctx, y, *pars):
        ctx.shapes = [(p.shape, p.dtype, p.device) for p in pars]
        return y.view_as(y)

    @staticmethod
    def backward(ctx, gy):
        return (gy,) + tuple(torch.zeros(s, dtype=d, device=dev) for s, d, dev in ctx.shapes)


def attach_zero_grad(y, pars):
    return _ZeroGrad.apply(y, *pars)


class ChainPlan:
    """A fused element-wise segment with its output buffers and marshalled arguments prepared once;
    ``launch()`` is a single C-ABI call (``risp_chain_fwd``).  ``outs[k]`` is stage k's output
    (SKIP stages alias their input)."""

    def __init__(self, x, ops, params, out_last=None):
        x = _dev(x, 'img')
        n, cin, h, w = x.shape
        if h % 2 or w % 2:
            raise ValueError('H and W must be even, got %s' % (tuple(x.shape),))
        if cin != (1 if ops[0] == OP_DEMOSAIC_NEAREST else 3):
            raise ValueError('chain input has %d channels' % cin)
        self.x, self.outs, cur = x, [], x
        count = sum(1 for op in ops if op != OP_SKIP)
        # ``out_last``: the caller's buffer for the segment's LAST computed stage (a contiguous fp32 (N,3,H,W) device tensor, 16-byte
        # aligned - test_split.run_frame hands a slice of the frame's tile stack); the other stages share ONE allocation
        if out_last is not None and (not count or tuple(out_last.shape) != (n, 3, h, w) or not out_last.is_contiguous()
                                     or out_last.dtype != torch.float32 or out_last.device != x.device or out_last.data_ptr() % 16):
            out_last = None
        own = count - (1 if out_last is not None else 0)
        bufs = iter(torch.empty((own, n, 3, h, w), device=x.device, dtype=torch.float32).unbind(0)) if own else iter(())
        left = count
        for op in ops:
            if op != OP_SKIP:
                left -= 1
                cur = out_last if (left == 0 and out_last is not None) else next(bufs)      # every stage output is a (N,3,H,W) view of ONE allocation
            self.outs.append(cur)
        self.params = [_dev(p) if p is not None else None for p in params]   # keep alive
        self._args = (_p(x), len(ops), (C.c_int * len(ops))(*ops),
                      L.ptr_array([p.data_ptr() if p is not None else None for p in self.params]),
                      L.ptr_array([o.data_ptr() if op != OP_SKIP else None for o, op in zip(self.outs, ops)]),
                      n, h, w)

    def launch(self):
        L.call('risp_chain_fwd', *self._args, _stream())
        return self.outs


def chain_forward(x, ops, params, out_last=None):
    """Fused element-wise segment: returns the list of stage outputs (SKIP aliases its input).

    Inference-only fast path (no autograd graph is recorded)."""
    return ChainPlan(x, ops, params, out_last).launch()


class BilateralChainPlan:
    """[nearest demosaic ->] bilateral -> element-wise chain as ONE launch (risp_bilateral_chain_fwd).
    ``outs`` lists the stage outputs in pipeline order: [demosaic,] bilateral, chain stages..."""

    def __init__(self, x, from_bayer, window, sigma_color, sigma_space, max_window, ops, params):
        x = _dev(x, 'img')
        n, cin, h, w = x.shape
        if cin != (1 if from_bayer else 3) or h % 2 or w % 4:
            raise ValueError('fused stencil segment: unsupported input %s' % (tuple(x.shape),))
        count = (2 if from_bayer else 1) + sum(1 for op in ops if op != OP_SKIP)
        bufs = iter(torch.empty((count, n, 3, h, w), device=x.device, dtype=torch.float32).unbind(0))
        new = lambda: next(bufs)              # every stage output is a (N,3,H,W) view of ONE allocation
        self.x = x
        self.dem = new() if from_bayer else None
        self.bil = new()
        self.outs = ([self.dem] if from_bayer else []) + [self.bil]
        cur, chain_outs = self.bil, []
        for op in ops:
            if op != OP_SKIP:
                cur = new()
            chain_outs.append(cur)
        self.outs += chain_outs
        if window.dtype != torch.int32 or not window.is_cuda:
            raise ValueError('window must be an int32 device tensor')
        self.keep = [window.contiguous(), _dev(sigma_color), _dev(sigma_space)] + \
                    [_dev(p) if p is not None else None for p in params]
        win, sc, ss = self.keep[:3]
        self._args = (_p(x), int(from_bayer), _p(self.dem), _p(self.bil), _p(win), _p(sc), _p(ss), int(max_window),
                      len(ops), (C.c_int * max(1, len(ops)))(*ops),
                      L.ptr_array([p.data_ptr() if p is not None else None for p in self.keep[3:]] or [None]),
                      L.ptr_array([o.data_ptr() if op != OP_SKIP else None for o, op in zip(chain_outs, ops)] or [None]),
                      n, h, w)

    def launch(self):
        L.call('risp_bilateral_chain_fwd', *self._args, _stream())
        return self.outs


class _FanOut(torch.autograd.Function):
    """k aliases of one tensor whose gradients are added by ONE launch (operand order, the order autograd's own pairwise
    additions take: the same bits) - the slot input of a super-net feeds the proxy group, Path-Restore and the fused mixture, and
    autograd otherwise spends k - 1 element-wise launches per slot and backward pass on the sum."""

    @staticmethod
    def forward(ctx, x, k):
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        live = [_dev(g, 'grad') for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        y = torch.empty_like(live[0])
        k = len(live)
        L.call('risp_mix_fwd', L.ptr_array([g.data_ptr() for g in live]), (C.c_float * k)(*([1.0] * k)), k, _p(y), y.numel(), _stream())
        return y, None


def fan_out(x, k):
    """[x] * k, or k aliases with a one-launch gradient sum (see _FanOut) when a gradient will flow into x on the device"""
    if k < 2 or not (x.is_cuda and x.requires_grad and torch.is_grad_enabled() and x.dtype == torch.float32 and k <= L.MIX_MAX
                     and x.is_contiguous() and x.numel() % 4 == 0 and x.data_ptr() % 16 == 0):
        return [x] * k
    return list(_FanOut.apply(x, k))


class _PixelLoss(torch.autograd.Function):
    """mean((y - gt)^2) / mean(|y - gt|) with the gradient formed in the same pass (risp_pixel_loss): two launches forward, a
    scaling backward - nn.MSELoss / nn.L1Loss are an element-wise launch + a reduction forward and two launches backward."""

    @staticmethod
    def forward(ctx, y, gt, kind):
        y, gt = _dev(y, 'output'), _dev(gt, 'target')
        if y.shape != gt.shape or y.numel() % 4 or (y.data_ptr() | gt.data_ptr()) % 16:
            raise ValueError('pixel_loss: shapes %s / %s must agree, numel %% 4 == 0, 16-byte aligned' % (tuple(y.shape), tuple(gt.shape)))
        lib = L.load()
        g = torch.empty_like(y) if ctx.needs_input_grad[0] else None
        loss = torch.empty((), device=y.device, dtype=torch.float32)
        scratch = torch.empty(lib.risp_loss_scratch_floats(), device=y.device, dtype=torch.float32)
        L.call('risp_pixel_loss', _p(y), _p(gt), _p(g), _p(loss), _p(scratch), y.numel(), kind, _stream())
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        g, = ctx.saved_tensors
        return (g * gl if g is not None else None), None, None


class _LocalGlobalL2(torch.autograd.Function):
    """local_global_loss with the mean-squared loss (utils/util_loss.py:26-64) - both branches on the device in one call
    (risp_local_global_l2), the flags never leave it; backward is a scaling of the gradient formed in the same pass."""

    @staticmethod
    def forward(ctx, y, gt, flag):
        y, gt = _dev(y, 'output'), _dev(gt, 'target')
        n, c, h, w = y.shape
        lib = L.load()
        flag = flag.to(device=y.device, dtype=torch.float32).contiguous()
        g = torch.empty_like(y) if ctx.needs_input_grad[0] else None
        loss = torch.empty((), device=y.device, dtype=torch.float32)
        scratch = torch.empty(lib.risp_local_global_scratch_floats(n, c), device=y.device, dtype=torch.float32)
        L.call('risp_local_global_l2', _p(y), _p(gt), _p(flag), _p(g), _p(loss), _p(scratch), n, c, h, w, _stream())
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gl):
        g, = ctx.saved_tensors
        return (g * gl if g is not None else None), None, None


def _written(tensors):
    """The C ABI wrote these tensors in place: bump their version counters as a torch in-place op would - the mixture-weight
    cache and the step-level reuse of the super-net key on them (a shifted parameter must miss)."""
    torch.autograd.graph.increment_version(list(tensors))


def _tensor_table(rows, with_numel_of=0):
    """risp_list_desc from rows of (a, b, c, e) tensors (None = NULL); numel of column ``with_numel_of``"""
    if len(rows) > L.LIST_MAX:
        raise ValueError('at most %d tensors per table, got %d' % (L.LIST_MAX, len(rows)))
    d = L.ListDesc()
    d.n = len(rows)
    keep = []
    for t, row in enumerate(rows):
        d.numel[t] = row[with_numel_of].numel()
        for name, v in zip('abce', row):
            if v is not None:
                if not (v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()):
                    raise RuntimeError('reconfigisp_amd: table entries must be contiguous fp32 CUDA/HIP tensors (no CPU fallback)')
                getattr(d, name)[t] = v.data_ptr()
                keep.append(v)
    return d, keep


class _HipImpl:
    """The product implementation: every op runs in libreconfigisp_hip.so."""

    @staticmethod
    def skip(x, p=None):
        return x

    @staticmethod
    def pixel_loss(y, gt, kind):
        return _PixelLoss.apply(y, gt, {'l2': 0, 'l1': 1}[kind])

    @staticmethod
    def local_global_l2(y, gt, flag):
        return _LocalGlobalL2.apply(y, gt, flag)

    @staticmethod
    def darts_virtual_step(rows, momentum, lr_meta):
        """rows of (twin parameter, parameter, gradient or None, momentum buffer or None): one launch (risp_darts_virtual_step)"""
        for at in range(0, len(rows), L.LIST_MAX):
            d, _keep = _tensor_table(rows[at: at + L.LIST_MAX], 1)
            L.call('risp_darts_virtual_step', C.byref(d), float(momentum), float(lr_meta), _stream())
        _written([r[0] for r in rows])

    @staticmethod
    def list_norm_eps(tensors):
        """(2,) device tensor: the 2-norm of the concatenated tensors (None skipped) and eps = 0 if norm < 1e-6 else 0.01 / norm"""
        live = [t for t in tensors if t is not None and t.numel()]
        dev = next(t.device for t in tensors if t is not None)
        out = torch.empty(2, device=dev, dtype=torch.float32)
        pieces = [live[at: at + L.LIST_MAX] for at in range(0, len(live), L.LIST_MAX)] or [[]]
        for k, piece in enumerate(pieces):          # any number of tensors (torch.cat(...).norm() has no limit): pieces of LIST_MAX
            d, _keep = _tensor_table([(t, None, t, None) for t in piece], 0)
            L.call('risp_list_norm_eps_part', C.byref(d), _p(out), int(k == 0), int(k == len(pieces) - 1), _stream())
        return out

    @staticmethod
    def list_axpy_scalar(pairs, scalar, factor):
        """p += (factor * scalar) * d for every (p, d), scalar a one-element device tensor"""
        for at in range(0, len(pairs), L.LIST_MAX):
            d, _keep = _tensor_table([(p, None, dd, None) for p, dd in pairs[at: at + L.LIST_MAX]], 0)
            L.call('risp_list_axpy_scalar', C.byref(d), _p(scalar), float(factor), _stream())
        _written([p for p, _ in pairs])

    @staticmethod
    def darts_alpha_grad(rows, eps, lr_meta):
        """rows of (out, dalpha or None, pos or None, neg or None) -> out = dalpha - lr_meta * (pos - neg) / 2 * eps, zeros where an
        input is missing or the finite-difference term holds a NaN; returns the (len(rows),) int32 NaN flags (on the device)"""
        flags = torch.zeros(len(rows), device=rows[0][0].device, dtype=torch.int32)
        for at in range(0, len(rows), L.LIST_MAX):
            d, _keep = _tensor_table(rows[at: at + L.LIST_MAX], 0)
            L.call('risp_darts_alpha_grad', C.byref(d), _p(eps), float(lr_meta), C.c_void_p(flags.data_ptr() + 4 * at), _stream())
        _written([r[0] for r in rows])
        return flags

    @staticmethod
    def wb_manual(x, p):
        return _Pointwise.apply(x, p, 'wb_manual', 3)

    @staticmethod
    def gamma(x, p):
        return _Pointwise.apply(x, p, 'gamma', 1)

    @staticmethod
    def gtm_manual(x, p):
        return _Pointwise.apply(x, p, 'gtm_manual', 3)

    @staticmethod
    def wb_quadratic(x, p):
        return _Pointwise.apply(x, p, 'wb_quadratic', 30)

    @staticmethod
    def grayworld(x, p=None):
        return _Grayworld.apply(x)

    @staticmethod
    def demosaic_nearest(x, p=None):
        return _DemosaicNearest.apply(x)

    @staticmethod
    def mix(w, outs, w_host=None, stacks=None):
        return _Mix.apply(w, w_host, stacks, *outs)

    @staticmethod
    def slot_mix(w, x, entries, w_host=None, stacks=None):
        kinds, flat = [], []
        for e in entries:
            if e[0] == 'tensor':
                kinds.append(L.SLOT_TENSOR)
                flat.append(e[1])
            else:
                kinds.append(SLOT_KINDS[e[1]])
                if e[1] not in ('skip', 'grayworld'):
                    flat.append(e[2])
        return _SlotMix.apply(w, w_host, x, tuple(kinds), stacks, *flat)

    @staticmethod
    def can_fuse_slot(x, names, tensors=()):
        """element-wise operands can be evaluated inside the mixture kernel: BGR input, 16-byte planes (the slot input and every
        materialised operand in ``tensors``), one of each kind"""
        return (x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and (x.shape[2] * x.shape[3]) % 4 == 0 and
                x.data_ptr() % 16 == 0 and len(set(names)) == len(names) and all(nm in SLOT_KINDS for nm in names) and
                all(t.data_ptr() % 16 == 0 and t.is_contiguous() for t in tensors))

    @staticmethod
    def prune_softmax(alpha, threshold, unavailable=None):
        return _PruneSoftmax.apply(alpha, threshold, unavailable)

    @staticmethod
    def param_blocks(raws, n):
        out = []
        for at in range(0, len(raws), L.PARAM_OPS_MAX):
            out += list(_ParamBlocks.apply(n, *raws[at: at + L.PARAM_OPS_MAX]))
        return out

    @staticmethod
    def histc01(x, bins):
        return histc01(x, bins)

    @staticmethod
    def conditional_fc(img, flat, widths):
        return _hip_conditional_fc(img, flat, widths)

    @staticmethod
    def origin_demosaic(x, option, scales=(1.0, 1.0)):
        return _hip_origin_demosaic(x, option, scales)

    @staticmethod
    def origin_tonemap(x, option, params, scales=(1.0, 1.0)):
        return _hip_origin_tonemap(x, option, params, scales)

    @staticmethod
    def origin_whiteworld(x, ratio, scales=(1.0, 1.0)):
        return _hip_origin_whiteworld(x, ratio, scales)

    @staticmethod
    def origin_denoise(x, option, params, scales=(1.0, 1.0)):
        return _hip_origin_denoise(x, option, params, scales)

    # CNN families: `module` is the nn.Module that owns the reference-shaped parameters
    @staticmethod
    def srcnn_res(x, pv, module):
        from . import convnets as CN
        packs = _packs(module, lambda: CN.build_srcnn_packs(module.srcnn, residual=True))
        # weight gradients only on request (proxy fine-tuning sets module.train_weights); the search itself never
        # uses them, although the proxies' tensors nominally require grad
        training = getattr(module, 'train_weights', False) and torch.is_grad_enabled()
        return CN.srcnn_res(x, pv, packs, module if training else None)

    @staticmethod
    def srcnn_demosaic(x, module):
        from . import convnets as CN
        packs = _packs(module, lambda: CN.build_srcnn_packs(module.srcnn))
        return CN.srcnn_demosaic(x, packs)

    @staticmethod
    def srcnn_res_group(x, pvs, modules, cache):
        from . import convnets as CN
        packs = [_packs(m, (lambda m=m: CN.build_srcnn_packs(m.srcnn, residual=True))) for m in modules]
        return CN.srcnn_res_group(x, pvs, packs, cache)

    @staticmethod
    def srcnn_demosaic_group(x, modules, cache, record=None):
        from . import convnets as CN
        packs = [_packs(m, (lambda m=m: CN.build_srcnn_packs(m.srcnn))) for m in modules]
        return CN.srcnn_demosaic_group(x, packs, cache, record)

    @staticmethod
    def can_group(modules, x):
        """True when these same-class proxies can run as one grouped launch per layer on input x"""
        from . import convnets as CN
        from .codes.models.modules.srcnn_res_arch import SRCNNRes
        if len(modules) < 2 or not x.is_cuda or x.shape[3] % 4 or x.data_ptr() % 16:
            return False
        if any(getattr(m, 'train_weights', False) for m in modules) and torch.is_grad_enabled():
            return False                             # a proxy under fine-tuning needs its weight gradients: own launches
        if isinstance(modules[0], SRCNNRes):
            return CN._srcnn_fold_ok(x.shape[2], x.shape[3]) and all(m.srcnn[4].weight.shape[0] <= 4 for m in modules)
        return x.shape[2] % 2 == 0 and x.shape[3] % 8 == 0

    @staticmethod
    def path14l_bayer(x, module):
        from . import convnets as CN
        packs = _packs(module, lambda: CN.build_path14l_packs(module.path_restore_14l, False))
        return CN.path14l(x, packs, True, module.__dict__.get('_risp_reuse'))

    @staticmethod
    def path14l_bgr(x, module):
        from . import convnets as CN
        packs = _packs(module, lambda: CN.build_path14l_packs(module.path_restore_14l, True))
        return CN.path14l(x, packs, False, module.__dict__.get('_risp_reuse'))


def _packs(module, build):
    from . import convnets as CN
    cache = module.__dict__.get('_risp_pack_cache')
    if cache is None:
        cache = module.__dict__['_risp_pack_cache'] = CN.PackCache()
    return cache.get(module, build)


_IMPL = _HipImpl


def skip(x, p=None):
    return _IMPL.skip(x, p)


def wb_manual(x, p):
    return _IMPL.wb_manual(x, p)


def gamma(x, p):
    return _IMPL.gamma(x, p)


def gtm_manual(x, p):
    return _IMPL.gtm_manual(x, p)


def wb_quadratic(x, p):
    return _IMPL.wb_quadratic(x, p)


def grayworld(x, p=None):
    return _IMPL.grayworld(x, p)


def demosaic_nearest(x, p=None):
    return _IMPL.demosaic_nearest(x, p)


def mix(w, outs, w_host=None, stacks=None):
    """sum_k w[k] * outs[k]; ``w_host``: the same weights as Python floats when the caller already holds them;
    ``stacks``: lists of operand positions whose gradients should come back as consecutive slices of one buffer (the
    members of a grouped launch read them in place)."""
    return _IMPL.mix(w, outs, w_host, stacks)


def slot_mix(w, x, entries, w_host=None, stacks=None):
    """The mixture of a slot with its element-wise operators evaluated on the fly.  entries[k] = ('tensor', o_k) for a
    materialised operand or ('op', name, block) with name in SLOT_KINDS and block the (N,P) parameter block the operator
    module would receive (None for skip / grayworld).  Same value as running the modules and ``mix``."""
    return _IMPL.slot_mix(w, x, entries, w_host, stacks)


def can_fuse_slot(x, names, tensors=()):
    return _IMPL.can_fuse_slot(x, names, tensors)


def pixel_loss(y, gt, kind='l2'):
    """nn.MSELoss ('l2') / nn.L1Loss ('l1') of the reference (models/darts_model.py:58-63, isp_model.py:29-34): a 0-dim tensor"""
    return _IMPL.pixel_loss(y, gt, kind)


def local_global_l2(y, gt, flag):
    """local_global_loss(y, gt, flag, nn.MSELoss()) of the reference (utils/util_loss.py:26-64): a 0-dim tensor"""
    return _IMPL.local_global_l2(y, gt, flag)


def darts_virtual_step(rows, momentum, lr_meta):
    return _IMPL.darts_virtual_step(rows, momentum, lr_meta)


def list_norm_eps(tensors):
    return _IMPL.list_norm_eps(tensors)


def list_axpy_scalar(pairs, scalar, factor):
    return _IMPL.list_axpy_scalar(pairs, scalar, factor)


def darts_alpha_grad(rows, eps, lr_meta):
    return _IMPL.darts_alpha_grad(rows, eps, lr_meta)


def prune_softmax(alpha, threshold, unavailable=None):
    """Mixture weights of a slot (super_prune_fifteen_demos_four_bayer_two.py:185-193): softmax, strict-< pruning
    against threshold * max on detached values, renormalisation by the detached sum.  ``unavailable``: uint8 mask of
    ops whose probability is forced to 0.  Returns post (K,), differentiable in alpha."""
    return _IMPL.prune_softmax(alpha, threshold, unavailable)


def param_blocks(raws, n):
    """[sigmoid(r).repeat(n, 1) for r in raws] (:204-209), one launch for the whole list."""
    return _IMPL.param_blocks(list(raws), n) if len(raws) else []


def hist_features(x, bins):
    return _IMPL.histc01(x, bins)


def conditional_fc(img, flat, widths):
    """(N, widths[-1]) = sigmoid(MLP(per-channel histograms of img) + flat[global]); widths = (3*bins, ..., out)."""
    return _IMPL.conditional_fc(img, flat, widths)


def srcnn_res(x, pv, module):
    return _IMPL.srcnn_res(x, pv, module)


def srcnn_demosaic(x, module):
    return _IMPL.srcnn_demosaic(x, module)


def srcnn_res_group(x, pvs, modules, cache):
    """[srcnn_res(x, pvs[g], modules[g]) for g] as ONE launch per layer (convnets.srcnn_res_group)."""
    return _IMPL.srcnn_res_group(x, pvs, modules, cache)


def srcnn_demosaic_group(x, modules, cache, record=None):
    return _IMPL.srcnn_demosaic_group(x, modules, cache, record)


def can_group(modules, x):
    return _IMPL.can_group(modules, x)


def path14l_bayer(x, module):
    return _IMPL.path14l_bayer(x, module)


def path14l_bgr(x, module):
    return _IMPL.path14l_bgr(x, module)


# ---------------------------------------------------------------------------
# classical "Origin" kernels (0..255 domain, non-differentiable)
# ---------------------------------------------------------------------------
def _vec(v, n, device, dtype=torch.float32):
    """per-image plugin parameter (numpy array / tensor / scalar) -> contiguous (N,) device tensor"""
    t = torch.as_tensor(v).detach().to(device=device, dtype=dtype).reshape(-1)
    if t.numel() == 1 and n > 1:
        t = t.repeat(n)
    if t.numel() != n:
        raise ValueError('expected %d per-image values, got %d' % (n, t.numel()))
    return t.contiguous()


def _check_odd(name, v):
    v = int(v)
    if v < 1 or v > 17 or v % 2 == 0:     # 17 = (1 * 7) * 2 + 3: a saturated sigmoid parameter (tools_origin.py:698,746,787)
        raise ValueError('%s must be an odd size in 1..17, got %d' % (name, v))
    return v


def _hip_origin_demosaic(x, option, scales=(1.0, 1.0)):
    x = _dev(x.detach(), 'img')
    if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] % 2 or x.shape[3] % 2:
        raise ValueError('expected a (N,1,H,W) RGGB mosaic with even H, W; got %s' % (tuple(x.shape),))
    n, _, h, w = x.shape
    y = torch.empty((n, 3, h, w), device=x.device, dtype=torch.float32)
    L.call('risp_origin_demosaic', _p(x), _p(y), int(option == 'laplacian'), n, h, w, scales[0], scales[1], _stream())
    return y


_TONEMAP = {'reinhard': (0, 'white_point', 'middle_grey'), 'crysisengine': (1, 'lum_adapted', None),
            'filmic': (2, 'white_point', 'exposure_bias')}


def _hip_origin_tonemap(x, option, params, scales=(1.0, 1.0)):
    x = _dev(x.detach(), 'img')
    _check_bgr(x)
    n, hw = x.shape[0], x.shape[2] * x.shape[3]
    mode, ka, kb = _TONEMAP[option]
    a = _vec(params[ka], n, x.device)
    b = _vec(params[kb], n, x.device) if kb else None
    y = torch.empty_like(x)
    ws = torch.empty(L.load().risp_origin_tonemap_scratch_floats(n), device=x.device, dtype=torch.float32)
    L.call('risp_origin_tonemap', _p(x), _p(y), mode, _p(a), _p(b), None, _p(ws), n, hw, scales[0], scales[1],
           _stream())
    return y


def _hip_origin_whiteworld(x, ratio, scales=(1.0, 1.0)):
    x = _dev(x.detach(), 'img')
    _check_bgr(x)
    n, hw = x.shape[0], x.shape[2] * x.shape[3]
    stats, _ = channel_stats(x, want_arg=False)
    y = torch.empty_like(x)
    ws = torch.empty(L.load().risp_origin_tonemap_scratch_floats(n), device=x.device, dtype=torch.float32)
    r = _vec(ratio, n, x.device)          # keep every operand alive until the launch is enqueued
    L.call('risp_origin_tonemap', _p(x), _p(y), 3, _p(r), None, _p(stats), _p(ws), n, hw, scales[0], scales[1],
           _stream())
    return y


def _hip_origin_denoise(x, option, params, scales=(1.0, 1.0)):
    x = _dev(x.detach(), 'img')
    _check_bgr(x)
    n, _, h, w = x.shape
    y = torch.empty_like(x)
    if option == 'median':
        L.call('risp_origin_median', _p(x), _p(y), _check_odd('median size', params['size']), n, h, w, scales[0],
               scales[1], _stream())
    elif option == 'bilateral':
        win = _vec(params['window_length'], n, x.device, torch.int32)
        sc, ss = _vec(params['sigma_color'], n, x.device), _vec(params['sigma_space'], n, x.device)
        wmax = _check_odd('window_length', params.get('max_window') or win.max().item())
        L.call('risp_origin_bilateral', _p(x), _p(y), _p(win), _p(sc), _p(ss), wmax, n, h, w, scales[0], scales[1],
               _stream())
    elif option == 'fastnlm':
        blk = _vec(params['block_size'], n, x.device, torch.int32)
        srch = _vec(params['search_block'], n, x.device, torch.int32)
        dec = _vec(params['decay_factor'], n, x.device)
        bmax = _check_odd('block_size', params.get('max_block') or blk.max().item())
        smax = _check_odd('search_block', params.get('max_search') or srch.max().item())
        L.call('risp_origin_fastnlm', _p(x), _p(y), _p(blk), _p(srch), _p(dec), bmax, smax, n, h, w, scales[0],
               scales[1], _stream())
    else:
        raise ValueError('unknown denoiser %r' % (option,))
    return y


def origin_demosaic(x, option, scales=(1.0, 1.0)):
    return _IMPL.origin_demosaic(x, option, scales)


def origin_tonemap(x, option, params, scales=(1.0, 1.0)):
    return _IMPL.origin_tonemap(x, option, params, scales)


def origin_whiteworld(x, ratio, scales=(1.0, 1.0)):
    return _IMPL.origin_whiteworld(x, ratio, scales)


def origin_denoise(x, option, params, scales=(1.0, 1.0)):
    return _IMPL.origin_denoise(x, option, params, scales)
