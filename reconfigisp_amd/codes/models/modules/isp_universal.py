"""IspUniversal - a fixed ISP built from an architecture string, proxy variant.

Host-side mirror of models/modules/isp_universal.py:12-236: same constructor arguments,
``forward``, ``trainable_parameters``, ``intermediate_results`` and state-dict keys
(``param_step<k>_<name>``).  Differences from the shipped reference, both documented in
SURVEY.md: the reference crashes at construction on undefined names (:92-94) - here pool
entries 19-21 raise only when selected; inference forwards run fused (pipeline_fusion.py).
"""
import numpy as np
import torch
import torch.nn as nn

from .... import functional as F
from . import registry as R
from .pipeline_fusion import fused_forward, wants_grad


class _FixedPipeline(nn.Module):
    """Shared machinery of IspUniversal / OriginUniversal."""
    srgb_names = R.NAMES_SRGB
    use_origin_kernels = False

    def _build(self, module_path, architecture, indiv_module_paths=None, conditional=None):
        conditional = conditional or {}
        self.architecture = architecture
        self.all_modules, self.all_params, self.is_conditional, self.step_names = [], [], [], []
        for step, (_, name) in enumerate(R.parse_architecture(architecture, self.srgb_names), start=1):
            override = indiv_module_paths[step - 1] if indiv_module_paths is not None and name in R.PROXY_NETS else None
            cond_ch = conditional.get(R.CONDITIONAL_KW.get(name))
            op = R.make_op(name, module_path, origin=self.use_origin_kernels, weight_override=override,
                           conditional_channels=cond_ch)
            init = list(R.PARAM_INIT[name])
            if name in R.CONDITIONAL_KW:
                # FC weights ~ N(0, 0.01^2), then the 'global' module parameters (isp_universal.py:185-190)
                init = list(np.random.randn(op.total_params - len(init)) * 0.01) + init
            self.all_modules.append(op)
            self.is_conditional.append(name in R.CONDITIONAL_KW)
            self.step_names.append(name)
            if init:
                key = 'param_step{}_{}'.format(step, name)
                setattr(self, key, nn.Parameter(torch.tensor(init, dtype=torch.float32)))
                self.all_params.append(getattr(self, key))
            else:
                self.all_params.append(nn.Parameter(torch.zeros(0)))
        self.intermediate_results = []

    def _apply(self, fn, *args, **kwargs):
        # sub-modules and zero-size placeholders live in plain lists (as in the reference, so the
        # state dict holds only param_step*); unlike the reference they still follow .to()/.cuda()
        super()._apply(fn, *args, **kwargs)
        for m in self.all_modules:
            m._apply(fn, *args, **kwargs)
        for p in self.all_params:
            if p.numel() == 0:
                p.data = fn(p.data)
        return self

    def _stage_params(self, n):
        if not torch.is_grad_enabled():
            # inference: sigmoid(p).repeat(N,1) only changes when a parameter does - keep the (N,P)
            # blocks resident instead of re-launching 2 tiny kernels per stage per call
            key = (n,) + tuple((p._version, p.data_ptr()) for p in self.all_params)
            if getattr(self, '_stage_cache_key', None) != key:
                self._stage_cache, self._stage_cache_key = self._build_stage_params(n), key
            return self._stage_cache
        return self._build_stage_params(n)

    def _build_stage_params(self, n):
        plain = [p for p, cond in zip(self.all_params, self.is_conditional) if p.numel() and not cond]
        blocks = iter(F.param_blocks(plain, n))                # sigmoid(p).repeat(n, 1), (N, P) in [0,1]: one launch
        out = []
        for p, cond in zip(self.all_params, self.is_conditional):
            if p.numel() == 0:
                out.append(None)
            elif cond:
                out.append(p)                                  # raw flat vector, no sigmoid / repeat
            else:
                out.append(next(blocks))
        return out

    def forward(self, x):
        pars = self._stage_params(x.size(0))
        # segment fusion works on 2 x 4 pixel patches: odd sizes (sRGB-only pipelines may see them) go op by op
        if x.is_cuda and not wants_grad(x, self.all_params) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            with torch.no_grad():
                # (final_out: set by test_split.run_frame around model.test() - the last stage written into the frame's tile stack)
                x, self.intermediate_results = fused_forward(self.all_modules, pars, x, self.__dict__.get('final_out'))
            return x
        self.intermediate_results = []
        for op, par in zip(self.all_modules, pars):
            x = op(x, par)
            self.intermediate_results.append(x)
        return x

    @property
    def trainable_parameters(self):
        return self.all_params


class IspUniversal(_FixedPipeline):
    srgb_names = R.NAMES_SRGB_EXT

    def __init__(self, module_path, indiv_module_paths, architecture, **kwargs):
        """kwargs: gamma_in_channels / wb_manual_in_channels / wb_quadratic_in_channels for the
        conditional modules (options key network_G.conditional_modules)."""
        super().__init__()
        self._build(module_path, architecture, indiv_module_paths, kwargs)
