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
_COND_OP = {T.ConditionalGamma: (F.OP_GAMMA, 1.0), T.ConditionalWbManual: (F.OP_WB_MANUAL, 1.0),
            T.ConditionalWbQuadratic: (F.OP_WB_QUADRATIC, 1.0)}
MAX_CHAIN = 8


def _flush(x, ops, params, results):
    if not ops:
        return x
    if all(o == F.OP_SKIP for o in ops):
        results.extend([x] * len(ops))
        return x
    outs = F.chain_forward(x, ops, params)
    results.extend(outs)
    return outs[-1]


def fused_forward(modules, param_tensors, x):
    """modules[k](x, param_tensors[k]) for all k, fusing where possible.  Returns (y, stage outputs)."""
    results, ops, params = [], [], []
    for mod, par in zip(modules, param_tensors):
        kind = type(mod)
        if kind is T.DemosaicNearest:
            x = _flush(x, ops, params, results)
            ops, params = [F.OP_DEMOSAIC_NEAREST], [None]
        elif kind in _CHAIN_OP and len(ops) < MAX_CHAIN:
            ops.append(_CHAIN_OP[kind])
            params.append(par)
        elif kind is T.Grayworld or kind in _COND_OP or kind in _CHAIN_OP:
            x = _flush(x, ops, params, results)
            if kind is T.Grayworld:
                ops, params = [F.OP_GAIN3], [F.grayworld_gains(x)]
            elif kind in _COND_OP:
                ops, params = [_COND_OP[kind][0]], [mod._fc_forward(x, par)]
            else:
                ops, params = [_CHAIN_OP[kind]], [par]
        else:
            x = _flush(x, ops, params, results)
            ops, params = [], []
            x = mod(x, par)
            results.append(x)
    x = _flush(x, ops, params, results)
    return x, results


def wants_grad(x, raw_params):
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in raw_params))
