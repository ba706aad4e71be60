"""The two optimizers of the search (models/darts_model.py:78-81: SGD with momentum for the operator parameters, Adam for the
alphas) with ``step()`` as ONE launch over the table of tensors instead of torch's list-wide ("foreach") sequence of 3 / 8
launches - the parameters of a super-net are <= 216 floats in ~45 tensors, every launch is pure latency.  Same state keys and
tensors as torch.optim (``momentum_buffer``; ``step``, ``exp_avg``, ``exp_avg_sq``): ``state_dict()``, schedulers and
checkpoints are untouched.  Anything the kernels do not cover (CPU tensors, weight decay, nesterov, amsgrad, closures, ...) goes
to the parent class."""
import ctypes as C
import math

import torch


def _table(rows):
    from ... import functional as F
    return F._tensor_table(rows, 0)


def _ok(p):
    return p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad is not None and not p.grad.is_sparse \
        and p.grad.is_contiguous() and p.grad.dtype == torch.float32


class ListSGD(torch.optim.SGD):
    @torch.no_grad()
    def step(self, closure=None):
        from ... import functional as F, lib as L
        groups = self.param_groups
        plain = closure is None and all(g['momentum'] != 0 and g['dampening'] == 0 and g['weight_decay'] == 0 and not g['nesterov']
                                        and not g.get('maximize') and not g.get('differentiable') for g in groups)
        live = [[p for p in g['params'] if p.grad is not None and p.numel()] for g in groups]
        if not plain or not all(_ok(p) for ps in live for p in ps):
            return super().step(closure)
        for g, ps in zip(groups, live):
            first, later = [], []
            for p in ps:
                st = self.state[p]
                buf = st.get('momentum_buffer')
                if buf is None:                             # torch: buf = clone(grad).detach() - here written by the launch
                    buf = st['momentum_buffer'] = torch.empty_like(p)
                    first.append((p, None, p.grad, buf))
                else:
                    later.append((p, None, p.grad, buf))
            for rows, flag in ((first, 1), (later, 0)):
                for at in range(0, len(rows), L.LIST_MAX):
                    d, _keep = _table(rows[at: at + L.LIST_MAX])
                    L.call('risp_sgd_momentum_step', C.byref(d), float(g['lr']), float(g['momentum']), flag, F._stream())
            F._written([p for p in ps] + [self.state[p]['momentum_buffer'] for p in ps])
        return None


class ListAdam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        from ... import functional as F, lib as L
        groups = self.param_groups
        plain = closure is None and all(g['weight_decay'] == 0 and not g['amsgrad'] and not g.get('maximize') and not g.get('capturable')
                                        and not g.get('differentiable') and not g.get('fused') and not torch.is_tensor(g['lr'])
                                        for g in groups)
        live = [[p for p in g['params'] if p.grad is not None and p.numel()] for g in groups]
        if not plain or not all(_ok(p) for ps in live for p in ps):
            return super().step(closure)
        for g, ps in zip(groups, live):
            beta1, beta2 = g['betas']
            by_step = {}
            for p in ps:
                st = self.state[p]
                if len(st) == 0:                            # Adam's lazy state initialisation (torch/optim/adam.py)
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1                             # a CPU scalar tensor: no device synchronisation
                by_step.setdefault(int(st['step'].item()), []).append((p, st['exp_avg'], p.grad, st['exp_avg_sq']))
            for t, rows in by_step.items():
                for at in range(0, len(rows), L.LIST_MAX):
                    d, _keep = _table(rows[at: at + L.LIST_MAX])
                    L.call('risp_adam_step', C.byref(d), float(g['lr']) / (1.0 - beta1 ** t), float(beta2),
                           1.0 - beta1, 1.0 - beta2, math.sqrt(1.0 - beta2 ** t), float(g['eps']), F._stream())
            F._written([p for p in ps] + [self.state[p][k] for p in ps for k in ('exp_avg', 'exp_avg_sq')])
        return None
