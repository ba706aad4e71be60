"""Fused training step of an element-wise fixed pipeline (``risp_chain_train_step``).

``FusedIspStep.build(netG, criterion, optimizer)`` returns a callable replacing the body of the reference's
``IspModel.optimize_parameters`` (models/isp_model.py:128-142) - forward, pixel loss, backward, Adam - by two kernel
launches, or ``None`` when the pipeline has a stage without a fused training form (CNN proxies, gray-world,
conditional heads, classical kernels), the criterion is not nn.MSELoss / nn.L1Loss (mean) or the optimiser is not a
plain torch.optim.Adam; the caller then keeps the op-by-op autograd path.

The torch optimiser stays the owner of its state: ``exp_avg`` / ``exp_avg_sq`` are the tensors in
``optimizer.state[p]`` (created here exactly as Adam's lazy initialisation would), ``state['step']`` is advanced on
the host, the learning rate is read from ``param_groups`` every step (schedulers keep working) and
``optimizer.state_dict()`` checkpoints what it always did.  ``p.grad`` is filled as ``backward()`` would.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import functional as F
from . import lib as L


class LazyScalar:
    """A device scalar that converts to a Python float only when somebody looks at it (``log_dict['loss']`` is
    written every step and read every print_freq steps: the reference's ``.item()`` per step, isp_model.py:142, is a
    host-device synchronisation per step)."""
    __slots__ = ('_t', '_v')

    def __init__(self, t):
        self._t, self._v = t, None

    def __float__(self):
        if self._v is None:
            self._v = float(self._t.item())
        return self._v

    item = __float__
    __format__ = lambda self, spec: format(float(self), spec)
    __repr__ = lambda self: repr(float(self))
    __str__ = lambda self: str(float(self))
    __sub__ = lambda self, o: float(self) - o
    __rsub__ = lambda self, o: o - float(self)
    __add__ = lambda self, o: float(self) + o
    __radd__ = __add__
    __mul__ = lambda self, o: float(self) * o
    __rmul__ = __mul__
    __truediv__ = lambda self, o: float(self) / o
    __lt__ = lambda self, o: float(self) < o
    __gt__ = lambda self, o: float(self) > o
    __le__ = lambda self, o: float(self) <= o
    __ge__ = lambda self, o: float(self) >= o
    __eq__ = lambda self, o: float(self) == o
    __abs__ = lambda self: abs(float(self))
    __hash__ = None


class FusedIspStep:
    def __init__(self, net, stages, from_bayer, loss_kind, optimizer):
        self.net, self.stages, self.from_bayer, self.loss_kind, self.opt = net, stages, from_bayer, loss_kind, optimizer
        self.group = optimizer.param_groups[0]
        self._plan = None

    # ------------------------------------------------------------------ applicability
    @staticmethod
    def build(net, criterion, optimizer):
        from .codes.models.modules import tools_origin as T
        ops = {T.WbManual: (F.OP_WB_MANUAL, 3), T.Gamma: (F.OP_GAMMA, 1), T.GtmManual: (F.OP_GTM_MANUAL, 3),
               T.WbQuadratic: (F.OP_WB_QUADRATIC, 30)}
        if type(criterion) is nn.MSELoss and criterion.reduction == 'mean':
            loss_kind = 0
        elif type(criterion) is nn.L1Loss and criterion.reduction == 'mean':
            loss_kind = 1
        else:
            return None
        if type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
            return None
        grp = optimizer.param_groups[0]
        if grp.get('weight_decay', 0) or grp.get('amsgrad') or grp.get('maximize') or grp.get('capturable') or \
                grp.get('differentiable'):
            return None
        mods, pars = getattr(net, 'all_modules', None), getattr(net, 'all_params', None)
        if not mods or any(getattr(net, 'is_conditional', [False])):
            return None
        stages, from_bayer, seen_demosaic = [], False, False
        for m, p in zip(mods, pars):
            if type(m) is T.Skip:                             # returns its input (tools_origin.py:256-262)
                continue
            if type(m) is T.DemosaicNearest and not seen_demosaic and not stages:
                from_bayer = seen_demosaic = True
                continue
            if type(m) not in ops:
                return None
            code, width = ops[type(m)]
            if p.numel() != width or not p.is_cuda or p.dtype != torch.float32:
                return None
            stages.append((code, width, p))
        if not 1 <= len(stages) <= L.TRAIN_MAX or sum(1 for s in stages if s[0] == F.OP_WB_QUADRATIC) > 1:
            return None
        if {id(s[2]) for s in stages} != {id(p) for p in grp['params'] if p.numel()}:
            return None                                        # the optimiser must own exactly these parameters
        return FusedIspStep(net, stages, from_bayer, loss_kind, optimizer)

    def accepts(self, img, gt):
        return (img.is_cuda and gt.is_cuda and img.dim() == 4 and img.shape[1] == (1 if self.from_bayer else 3) and
                img.shape[2] % 2 == 0 and img.shape[3] % 2 == 0 and gt.shape == (img.shape[0], 3) + tuple(img.shape[2:]) and
                img.dtype == torch.float32 and gt.dtype == torch.float32)

    # ------------------------------------------------------------------ per-shape plan
    def _state(self, p):
        st = self.opt.state[p]
        if len(st) == 0:                                       # Adam's lazy state initialisation (torch/optim/adam.py)
            st['step'] = torch.tensor(0.0, dtype=torch.float32)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _make_plan(self, n, device):
        blocks = []
        with torch.no_grad():
            for code, width, p in self.stages:
                b = torch.sigmoid(p).repeat(n, 1)
                blocks.append((b * 5 if code == F.OP_WB_MANUAL else b).contiguous())
        loss = torch.zeros(1, device=device, dtype=torch.float32)
        scratch = torch.empty(L.load().risp_train_scratch_floats(n), device=device, dtype=torch.float32)
        return {'n': n, 'blocks': blocks, 'loss': loss, 'scratch': scratch,
                'versions': [p._version for _, _, p in self.stages]}

    def __call__(self, img, gt):
        """-> (output (N,3,H,W), loss as a LazyScalar).  Parameters, Adam state and .grad are updated in place."""
        img, gt = img.contiguous(), gt.contiguous()
        n, _, h, w = img.shape
        plan = self._plan
        # somebody else wrote the parameters (load_state_dict, resume, a manual edit): rebuild the per-image blocks
        if plan is None or plan['n'] != n or plan['versions'] != [p._version for _, _, p in self.stages] or \
                plan['loss'].device != img.device:
            plan = self._plan = self._make_plan(n, img.device)
        states = [self._state(p) for _, _, p in self.stages]
        for _, _, p in self.stages:
            if p.grad is None:                                 # zero_grad(set_to_none=True) was here
                p.grad = torch.zeros_like(p)
        step = int(states[0]['step'].item()) + 1               # CPU scalar tensor: no device synchronisation
        beta1, beta2 = self.group['betas']
        lr, eps = float(self.group['lr']), float(self.group['eps'])
        y = torch.empty((n, 3, h, w), device=img.device, dtype=torch.float32)
        loss = torch.empty(1, device=img.device, dtype=torch.float32)
        d = L.TrainDesc()
        d.in_, d.gt, d.y = F._p(img), F._p(gt), F._p(y)
        d.from_bayer, d.n_ops, d.loss_kind = int(self.from_bayer), len(self.stages), self.loss_kind
        for k, ((code, _, p), st, blk) in enumerate(zip(self.stages, states, plan['blocks'])):
            d.ops[k] = code
            d.blocks[k], d.raw[k], d.grad[k] = blk.data_ptr(), p.data_ptr(), p.grad.data_ptr()
            d.exp_avg[k], d.exp_avg_sq[k] = st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
        d.N, d.H, d.W = n, h, w
        d.lr_step = lr / (1.0 - beta1 ** step)
        d.beta1, d.beta2, d.bias2_sqrt, d.eps = beta1, beta2, math.sqrt(1.0 - beta2 ** step), eps
        d.one_minus_beta1, d.one_minus_beta2 = 1.0 - beta1, 1.0 - beta2            # in double, rounded once (torch's lerp_ / addcmul_ weights)
        d.loss, d.scratch = F._p(loss), F._p(plan['scratch'])
        L.call('risp_chain_train_step', C.byref(d), F._stream())
        for st in states:
            st['step'] += 1
        for _, _, p in self.stages:                            # the kernel wrote the parameters behind autograd's back:
            torch.autograd.graph.increment_version(p)          # tell everybody who caches on p._version (host-side only)
        plan['versions'] = [p._version for _, _, p in self.stages]
        return y, LazyScalar(loss)
