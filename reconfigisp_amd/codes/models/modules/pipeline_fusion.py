"""Segment fusion for fixed pipelines (inference): maximal runs of element-wise stages are
executed by ONE ``risp_chain_fwd`` launch that reads the segment input once and writes every
stage output (``intermediate_results`` is API: test.py:74 consumes every stage).

Stages that need a whole-image quantity first (gray-world means, conditional-head histograms)
start a new segment: the reduction runs on the segment input, then the stage joins the next
chain as a per-image-parameter op.  CNN stages and the classical stencils run as themselves.
"""
import torch

from .... import functional as F
from . import tools_origin as T

_CHAIN_OP = {T.WbManual: F.OP_WB_MANUAL, T.Gamma: F.OP_GAMMA, T.GtmManual: F.OP_GTM_MANUAL,
             T.WbQuadratic: F.OP_WB_QUADRATIC, T.Skip: F.OP_SKIP}
_COND_OP = {T.ConditionalGamma: (F.OP_GAMMA, 1.0), T.ConditionalWbManual: (F.OP_WB_MANUAL, 5.0),
            T.ConditionalWbQuadratic: (F.OP_WB_QUADRATIC, 1.0)}
MAX_CHAIN = 8
_UNIT = (255.0, 255.0)     # the classical kernels work in 0..255: x255 on load, /255 on store, in the kernel


def _origin_call(mod, par):
    """Closure running one classical op directly on [0,1] tensors (no x255 / /255 passes, no host
    round trip per call): the plugin parameters are derived once per parameter version."""
    opt = mod.option
    if isinstance(mod, (T.OriginDemosBilinear, T.OriginDemosLaplacian)):
        return lambda x: F.origin_demosaic(x, opt, _UNIT)
    p = par.detach()
    if isinstance(mod, T.OriginToneReinhard):
        d = {'white_point': p[:, 0].contiguous(), 'middle_grey': p[:, 1].contiguous()}
        return lambda x: F.origin_tonemap(x, opt, d, _UNIT)
    if isinstance(mod, T.OriginToneCrysis):
        d = {'lum_adapted': p[:, 0].contiguous()}
        return lambda x: F.origin_tonemap(x, opt, d, _UNIT)
    if isinstance(mod, T.OriginToneFilmic):
        d = {'white_point': p[:, 0].contiguous(), 'exposure_bias': (p[:, 1] * 9. + 1.).contiguous()}
        return lambda x: F.origin_tonemap(x, opt, d, _UNIT)
    if isinstance(mod, T.OriginWbWhiteworld):
        r = p[:, 0].contiguous()
        return lambda x: F.origin_whiteworld(x, r, _UNIT)
    d = mod._params(p, {})                       # bilateral / median / fastnlm: the wrapper's own scaling rules
    if isinstance(mod, T.OriginNoiseBilateral):
        d['max_window'] = int(d['window_length'].max().item())
    elif isinstance(mod, T.OriginNoiseFastnlm):
        d['max_block'], d['max_search'] = int(d['block_size'].max().item()), int(d['search_block'].max().item())
    return lambda x: F.origin_denoise(x, opt, d, _UNIT)


def _run_origin(mod, x, par):
    key = None if par is None else (par.data_ptr(), par._version, tuple(par.shape))
    plan = mod.__dict__.get('_risp_origin_plan')
    if plan is None or plan[0] != key:
        plan = mod.__dict__['_risp_origin_plan'] = (key, _origin_call(mod, par))
    return plan[1](x)


def _wb_gain(mod, par):
    """params * 5 (tools_origin.py:214), computed once per parameter version."""
    key = (par.data_ptr(), par._version, tuple(par.shape))
    cached = mod.__dict__.get('_risp_gain')
    if cached is None or cached[0] != key:
        cached = mod.__dict__['_risp_gain'] = (key, par.detach() * 5)
    return cached[1]


def _chain_param(mod, par):
    return _wb_gain(mod, par) if type(mod) is T.WbManual else par


def _flush(x, ops, params, results, out_last=None):
    if not ops:
        return x
    if all(o == F.OP_SKIP for o in ops):
        results.extend([x] * len(ops))
        return x
    outs = F.chain_forward(x, ops, params, out_last)
    results.extend(outs)
    return outs[-1]


def _bilateral_segment(mod, par, x, from_bayer, tail_ops, tail_params):
    """[demosaic ->] bilateral -> element-wise tail as one launch; returns the stage outputs."""
    key = (par.data_ptr(), par._version, tuple(par.shape))
    cached = mod.__dict__.get('_risp_bilateral_args')
    if cached is None or cached[0] != key:
        d = mod._params(par.detach(), {})
        cached = mod.__dict__['_risp_bilateral_args'] = (
            key, d['window_length'].to(torch.int32).contiguous(), d['sigma_color'].contiguous(),
            d['sigma_space'].contiguous(), int(d['window_length'].max().item()))
    _, win, sc, ss, wmax = cached
    return F.BilateralChainPlan(x, from_bayer, win, sc, ss, wmax, tail_ops, tail_params).launch()


def fused_forward(modules, param_tensors, x, final_out=None):
    """modules[k](x, param_tensors[k]) for all k, fusing where possible.  Returns (y, stage outputs).  ``final_out``: where the caller
    wants the LAST stage (test_split: a slice of the frame's tile stack - no concatenation afterwards); honoured when the pipeline ends in
    an element-wise segment, otherwise the last stage is copied there."""
    results, ops, params = [], [], []
    seg_in = x                                    # input of the pending element-wise chain
    k, count = 0, len(modules)
    while k < count:
        mod, par = modules[k], param_tensors[k]
        kind = type(mod)
        if kind is T.DemosaicNearest:
            x = _flush(x, ops, params, results)
            seg_in, ops, params = x, [F.OP_DEMOSAIC_NEAREST], [None]
        elif kind is T.OriginNoiseBilateral and x.shape[3] % 4 == 0 and x.shape[2] % 2 == 0 and min(x.shape[2:]) > 8:
            from_bayer = ops == [F.OP_DEMOSAIC_NEAREST]
            if not from_bayer:
                x = _flush(x, ops, params, results)
                seg_in = x
            tail_ops, tail_params, j = [], [], k + 1
            while j < count and type(modules[j]) in _CHAIN_OP and len(tail_ops) < MAX_CHAIN:
                tail_ops.append(_CHAIN_OP[type(modules[j])])
                tail_params.append(_chain_param(modules[j], param_tensors[j]))
                j += 1
            outs = _bilateral_segment(mod, par, seg_in, from_bayer, tail_ops, tail_params)
            results.extend(outs)
            x, ops, params, k = outs[-1], [], [], j - 1
        elif kind in _CHAIN_OP and len(ops) < MAX_CHAIN:
            if not ops:
                seg_in = x
            ops.append(_CHAIN_OP[kind])
            params.append(_chain_param(mod, par))
        elif kind is T.Grayworld or kind in _COND_OP or kind in _CHAIN_OP:
            x = _flush(x, ops, params, results)
            seg_in = x
            if kind is T.Grayworld:
                ops, params = [F.OP_GAIN3], [F.grayworld_gains(x)]
            elif kind in _COND_OP:
                ops, params = [_COND_OP[kind][0]], [mod._fc_forward(x, par) * _COND_OP[kind][1]]
            else:
                ops, params = [_CHAIN_OP[kind]], [_chain_param(mod, par)]
        else:
            x = _flush(x, ops, params, results)
            ops, params = [], []
            x = _run_origin(mod, x, par) if isinstance(mod, T._OriginOp) else mod(x, par)
            results.append(x)
        k += 1
    x = _flush(x, ops, params, results, final_out)
    if final_out is not None and results and x.data_ptr() != final_out.data_ptr() and tuple(x.shape) == tuple(final_out.shape):
        final_out.copy_(x)
        x = results[-1] = final_out
    return x, results


def wants_grad(x, raw_params):
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in raw_params))
