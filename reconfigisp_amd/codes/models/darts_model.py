"""DartsModel - one architecture-search iteration = optimize_alphas() + optimize_parameters()
(mirror of models/darts_model.py:19-330).

Second-order DARTS as the reference does it: virtual SGD step into a twin network netV,
validation loss at the virtual weights, finite-difference Hessian-vector product with
eps = 0.01/||dp||, alpha gradient ``dalpha - lr_meta * hessian`` where the reference computes
``hessian = (pos - neg) / 2. * eps`` (sic, :323 - kept), Adam on alpha, SGD(momentum) on the
module parameters.  5 forwards + 5 backwards of the super-net per iteration.

Multi-GPU (one process per GPU, RCCL over xGMI): the weight step's gradients are averaged
across ranks (what DDP's bucket all-reduce does in the reference, :31,173) by ONE flat
all-reduce of the 146..216-float gradient vector; ``train.sync_arch_grads`` (default true, new)
also averages the three architecture-step gradient sets so that W ranks reproduce one process
with a W-times larger batch (the reference leaves them rank-local).
"""
import logging
import time
from collections import OrderedDict
from functools import partial

import torch
import torch.distributed as dist
import torch.nn as nn

from ... import functional as F
from ..utils.util_loss import latency_loss, local_global_loss
from . import networks
from .base_model import BaseModel
from .isp_model import make_schedulers
from .list_optim import ListAdam, ListSGD

logger = logging.getLogger('base')


class PixelLoss(nn.Module):
    """nn.MSELoss ('l2') / nn.L1Loss ('l1') of the reference (:58-63) - loss and gradient in one pass on the device
    (functional.pixel_loss); tensors the kernel does not take (numel % 4 != 0, unaligned views, other dtypes) and a target that
    itself needs a gradient (the kernel returns none for it) go to torch.  CPU tensors raise, like every other operator of this
    build (no CPU fallback); a second-order pass through the loss raises as well (the gradient is formed once, as a constant)."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind

    def forward(self, out, gt):
        if (out.dtype == torch.float32 and gt.dtype == torch.float32 and out.shape == gt.shape and out.numel() % 4 == 0
                and not gt.requires_grad and out.is_contiguous() and gt.is_contiguous() and (out.data_ptr() | gt.data_ptr()) % 16 == 0):
            return F.pixel_loss(out, gt, self.kind)
        return nn.functional.mse_loss(out, gt) if self.kind == 'l2' else nn.functional.l1_loss(out, gt)


def _criterion(kind, train_opt, device):
    if kind == 'l1':
        return PixelLoss('l1').to(device)
    mse = PixelLoss('l2').to(device)
    if kind == 'l2':
        return mse
    if kind == 'local_global_l2':
        return partial(local_global_loss, loss_func=mse)
    if kind == 'l2_latency':
        return partial(latency_loss, target_latency=train_opt['target_latency'], w=train_opt['w'],
                       fidelity_loss=mse)
    raise NotImplementedError('pixel_criterion [{}]'.format(kind))


class DartsModel(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.distributed = bool(opt['dist'])
        self.rank = dist.get_rank() if self.distributed else -1
        self.world = dist.get_world_size() if self.distributed else 1
        self.netG = networks.define_G(opt).to(self.device)
        self.netV = networks.define_G(opt).to(self.device)     # twin for the virtual step (not a deepcopy)
        self.netG_attr = self.netG
        self._sync_frozen_ops()
        if self.rank <= 0:
            self.print_network()
        self.load()
        self.img = self.gt = self.output = self.glb_flag = None
        self.val_img = self.val_gt = self.val_loss = self.val_glb_flag = None
        self.latency = self.latency_term = None
        self.comm_seconds = None

        if self.is_train:
            t = opt['train']
            self.netG.train()
            kind = t['pixel_criterion']
            self.is_local_global = 'local_global' in kind
            self.is_latency = 'latency' in kind
            self.cri_pix = _criterion(kind, t, self.device)
            self.cri_pix_v = _criterion(kind, t, self.device)
            self.momentum_G = t['momentum_G']
            self.lr_meta = t['lr_meta']
            self.sync_arch_grads = bool(t.get('sync_arch_grads', True)) if hasattr(t, 'get') else True
            self.step_reuse = bool(t.get('step_reuse', True)) if hasattr(t, 'get') else True
            self.weight_step_alpha_grads = bool(t.get('weight_step_alpha_grads', False)) if hasattr(t, 'get') else False
            # torch.optim.SGD / Adam with step() as one launch each (list_optim.py; same state, same arithmetic)
            self.optimizer_G = ListSGD(self.netG_attr.trainable_parameters, t['lr_G'], momentum=self.momentum_G)
            self.optimizer_alpha = ListAdam(self.netG_attr.alphas, lr=t['lr_G'], betas=(t['beta1'], t['beta2']))
            self.optimizers += [self.optimizer_G, self.optimizer_alpha]
            self.schedulers += make_schedulers(self.optimizers, t)
        else:
            self.netG.eval()
        self.log_dict = OrderedDict()

    # ------------------------------------------------------------------ plumbing
    def _sync_frozen_ops(self):
        """The searched net, its virtual-step twin and every rank must evaluate the SAME frozen operators.  The
        reference gets this from loading the same weight files into both nets (tools_proxy.py:28-39); with
        ``module_path: None`` each constructor draws its own random proxies, so rank 0's netG is broadcast (every
        registered parameter plus the proxies kept in plain lists, darts_model.py:31-44) and copied into netV."""
        def ops(net):       # the operators sit in plain (nested) lists, not in the module tree
            for entry in getattr(net, 'all_modules', []):
                for m in (entry if isinstance(entry, (list, tuple)) else [entry]):
                    yield m

        if self.distributed:
            tensors = list(self.netG.parameters())
            for m in ops(self.netG):
                tensors += list(m.parameters()) + list(m.buffers())
            with torch.no_grad():
                for t in tensors:
                    if t.numel():
                        dist.broadcast(t, src=0)          # on the tensor itself: the in-place write bumps its version
            for net in (self.netG, self.netV):
                if hasattr(net, 'invalidate_weight_cache'):
                    net.invalidate_weight_cache()
        with torch.no_grad():
            for g, v in zip(ops(self.netG), ops(self.netV)):
                v.load_state_dict(g.state_dict())
            for g, v in zip(self.netG.parameters(), self.netV.parameters()):
                v.copy_(g)

    def print_network(self):
        s, n = self.get_network_description(self.netG)
        logger.info('Network G structure: {}, with parameters: {:,d}'.format(self.netG.__class__.__name__, n))
        logger.info(s)

    def get_current_log(self):
        return self.log_dict

    def load(self):
        path = self.opt['path']['pretrain_model_G']
        if path is not None:
            logger.info('Loading model for G [{:s}] ...'.format(path))
            self.load_network(path, self.netG, self.opt['path']['strict_load'])

    def save(self, iter_label):
        self.save_network(self.netG, 'G', iter_label)

    def feed_data(self, data):
        """(img, gt) | (img, gt, val_img, val_gt) | (img, gt, flag, val_img, val_gt, val_flag)"""
        if len(data) == 4:
            self.val_img, self.val_gt = data[2].to(self.device), data[3].to(self.device)
        elif len(data) == 6:
            self.glb_flag, self.val_glb_flag = data[2].to(self.device), data[5].to(self.device)
            self.val_img, self.val_gt = data[3].to(self.device), data[4].to(self.device)
        elif len(data) != 2:
            raise ValueError('Invalid data format.')
        self.img, self.gt = data[0].to(self.device), data[1].to(self.device)

    def _loss(self, net, img, gt, flag, criterion):
        if self.is_latency:
            out, lat = net(img)
            loss, term = criterion(out, gt, lat)
            return loss, out, lat, term
        out = net(img)
        loss = criterion(out, gt, flag) if self.is_local_global else criterion(out, gt)
        return loss, out, None, None

    def _allreduce_mean(self, tensors):
        """Average a list of small gradient tensors over the ranks with ONE flat all-reduce (RCCL): a persistent flat buffer per
        gradient set, one list-wide copy in, the collective (averaging on RCCL: no division launch), one list-wide copy out -
        three launches whatever the number of tensors (was torch.cat of ~45 tensors + ~45 copy_ launches, four times per
        iteration)."""
        if not self.distributed:
            return tensors                          # (a world of one still goes through the collective: same code path)
        live = [t for t in tensors if t is not None and t.numel()]
        if not live:
            return tensors
        sizes = tuple(t.numel() for t in live)
        key = (sizes, live[0].device, live[0].dtype)
        cache = self.__dict__.setdefault('_flat_grads', {})
        if key not in cache:                        # the views are kept with the buffer: nothing is re-sliced per call
            flat = torch.empty(sum(sizes), device=live[0].device, dtype=live[0].dtype)
            cache[key] = (flat, [v.view_as(t) for v, t in zip(flat.split(sizes), live)])
        flat, views = cache[key]
        torch._foreach_copy_(views, live)
        probe = self.comm_seconds is not None       # bench.py: seconds inside the collectives (synchronising)
        if probe:
            torch.cuda.synchronize(flat.device) if flat.is_cuda else None
            t0 = time.perf_counter()
        if flat.is_cuda and dist.get_backend() == 'nccl':
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:                                       # gloo (CPU tests, dry runs) has no averaging reduction
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat /= self.world
        if probe:
            torch.cuda.synchronize(flat.device) if flat.is_cuda else None
            self.comm_seconds += time.perf_counter() - t0
        torch._foreach_copy_(live, views)
        return tensors

    # ------------------------------------------------------------------ weight step
    def optimize_parameters(self):
        l_pix, self.output, self.latency, self.latency_term = self._loss(self.netG, self.img, self.gt,
                                                                         self.glb_flag, self.cri_pix)
        self.optimizer_G.zero_grad()
        if self.weight_step_alpha_grads:
            l_pix.backward()
        else:
            # The weight step uses the gradients of the module parameters only (optimizer_G, :86-88).  The reference's
            # l_pix.backward() (:173) also accumulates d loss / d alpha into alpha.grad, which nothing reads: the next
            # optimize_alphas() starts with optimizer_alpha.zero_grad() (:226).  Asking for what is used lets autograd skip
            # every node that leads only to alphas - the backward of the demosaic proxies and of the first sRGB slot's
            # Path-Restore - and the grouped backward its unneeded input gradient.  Parameters, alphas, losses and outputs
            # are unchanged; only that transient alpha.grad is not written (train.weight_step_alpha_grads: true restores it).
            params = [p for p in self.netG_attr.trainable_parameters if p.numel()]
            only = getattr(self.netG_attr, 'params_only_backward', None)
            if only:
                only(True, params)
            try:
                grads = torch.autograd.grad(l_pix, params, allow_unused=True)
            finally:
                if only:
                    only(False)
            for p, g in zip(params, grads):
                p.grad = g
        self._allreduce_mean([p.grad for p in self.netG_attr.trainable_parameters])
        self.optimizer_G.step()
        self.log_dict['loss'] = l_pix.item()
        self._report_nan_flags()                    # (the queue is drained by the read-out above anyway)
        if self.is_latency:
            self.log_dict['latency'] = self.latency.item()
            self.log_dict['latency_term'] = self.latency_term.item()

    def _report_nan_flags(self):
        """The reference prints its NaN warning inside optimize_alphas() (:258-261); here the flags of an architecture step are read
        where the iteration next synchronises - the loss read-out of optimize_parameters(), or the start of the next
        optimize_alphas() when no weight step came in between - so no warning is lost or overwritten."""
        flags, self._nan_flags = getattr(self, '_nan_flags', None), None
        if flags is not None:
            for idx, bad in enumerate(flags.tolist()):
                if bad:
                    print('Warning: NaN in hessian, for the {}-th alpha'.format(idx + 1))

    # ------------------------------------------------------------------ architecture step
    def virtual_step(self):
        """p' = p - lr_meta * (momentum * buf + dL_trn/dp) written into netV; alphas copied."""
        loss = self._loss(self.netG, self.img, self.gt, self.glb_flag, self.cri_pix)[0]
        params = self.netG_attr.trainable_parameters
        only = getattr(self.netG_attr, 'params_only_backward', None) if self.step_reuse else None
        if only:
            only(True, params)    # nobody consumes the input gradient of the first parametrised slot in this pass
        try:
            grads = list(torch.autograd.grad(loss, params, allow_unused=True))
        finally:
            if only:
                only(False)
        if self.sync_arch_grads:
            self._allreduce_mean(grads)
        # the reference's per-parameter loop (darts_model.py:208-218) as ONE launch over a table of the tensors (functional.
        # darts_virtual_step -> risp_darts_virtual_step: the same operations in the same order on every element); the alphas are
        # rows without a gradient (plain copies)
        with torch.no_grad():
            rows = []
            for p, vp, g in zip(params, self.netV.trainable_parameters, grads):
                if len(p):
                    rows.append((vp, p, g, self.optimizer_G.state[p].get('momentum_buffer') if g is not None else None))
            rows += [(va, a, None, None) for va, a in zip(self.netV.alphas, self.netG_attr.alphas)]
            if rows:
                F.darts_virtual_step(rows, self.momentum_G, self.lr_meta)

    def optimize_alphas(self):
        # forwards #1 (virtual step), #3 and #4 (Hessian) evaluate netG on the same train batch with the same alphas: the
        # parameter-free CNN ops upstream of the first shifted parameter are computed once (SuperPrune...begin_reuse)
        reuse = self.step_reuse and hasattr(self.netG_attr, 'begin_reuse')
        if reuse:
            self.netG_attr.begin_reuse()
        try:
            self._optimize_alphas()
        finally:
            if reuse:
                self.netG_attr.end_reuse()

    def _optimize_alphas(self):
        self._report_nan_flags()                    # a previous architecture step that no weight step followed
        self.optimizer_alpha.zero_grad()
        self.virtual_step()
        loss = self._loss(self.netV, self.val_img, self.val_gt, self.val_glb_flag, self.cri_pix_v)[0]
        self.val_loss = loss
        v_alphas, v_params = tuple(self.netV.alphas), tuple(self.netV.trainable_parameters)
        grads = list(torch.autograd.grad(loss, v_alphas + v_params, allow_unused=True))
        if self.sync_arch_grads:
            self._allreduce_mean(grads)
        dalpha, dp = grads[:len(v_alphas)], grads[len(v_alphas):]
        pos, neg, eps = self.compute_hessian(dp)
        with torch.no_grad():
            # alpha.grad = dalpha - lr_meta * (pos - neg) / 2 * eps, zeros where a term is missing or the finite difference holds a
            # NaN (:254-265 with :313-323), for all alphas in ONE launch.  The reference's per-alpha `torch.isnan(h).any()` is a
            # device synchronisation each; the flags stay on the device and are looked at where the iteration synchronises anyway
            # (the loss read-out of optimize_parameters).
            alphas = list(self.netG_attr.alphas)
            for a in alphas:
                if a.grad is None:
                    a.grad = torch.empty_like(a)
            self._nan_flags = F.darts_alpha_grad([(a.grad, da, p, n) for a, da, p, n in zip(alphas, dalpha, pos, neg)], eps, self.lr_meta)
        self.optimizer_alpha.step()

    def compute_hessian(self, dp):
        """The two shifted architecture gradients and eps of (dalpha L_trn(p + eps dp) - dalpha L_trn(p - eps dp)) / 2. * eps,
        eps = 0.01/||dp|| (:270-324); the difference itself is formed with the architecture gradient (one launch)."""
        # eps = 0 if norm < 1e-6 else 0.01 / norm (:276-277) without reading norm back: the comparison on the host is a
        # device synchronisation in the middle of the iteration
        live_dp = [w for w in dp if w is not None]
        if live_dp:
            eps = F.list_norm_eps(live_dp)[1:2]
        else:
            eps = torch.zeros(1, device=self.device)
        params = self.netG_attr.trainable_params
        live = [(p, d) for p, d in zip(params, dp) if len(p) > 0 and d is not None]

        def shift(factor):                                  # p += factor * eps * dp for every parameter: one launch
            if live:
                with torch.no_grad():
                    F.list_axpy_scalar(live, eps, factor)

        def dalpha_at():
            loss = self._loss(self.netG, self.img, self.gt, self.glb_flag, self.cri_pix)[0]
            return list(torch.autograd.grad(loss, self.netG_attr.alphas, allow_unused=True))

        shift(1.)
        pos = dalpha_at()
        shift(-2.)
        neg = dalpha_at()
        shift(1.)
        if self.sync_arch_grads:
            self._allreduce_mean(pos + neg)
        return pos, neg, eps

    def test(self):
        with torch.no_grad():
            self.output = self.netG(self.img)
        return self.output, self.netG_attr.intermediate_results
