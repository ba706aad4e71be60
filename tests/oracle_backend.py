"""TEST-ONLY operator backend: binds the functional seam (reconfigisp_amd.functional._IMPL) to the
CPU oracle so that the HOST logic of the product (registry, fixed pipelines, super-net, DARTS
step, gloo multi-process path) can be exercised on a machine without a GPU.  The product never
imports this file; `-m gpu` tests never use it."""
import torch

import isp_oracle as O


def _sd(module):
    return {k: v.detach() for k, v in module.state_dict().items()}


def _virtual_step(rows, momentum, lr_meta):
    # the reference's per-parameter loop, darts_model.py:208-218
    with torch.no_grad():
        for vp, p, g, buf in rows:
            if g is None:
                vp.copy_(p)
            else:
                upd = (buf * momentum if buf is not None else 0. * momentum) + g
                vp.copy_(p - upd * lr_meta)


def _norm_eps(tensors):
    norm = torch.cat([w.reshape(-1) for w in tensors if w is not None]).norm()
    return torch.stack([norm, torch.where(norm < 1e-6, torch.zeros_like(norm), 0.01 / norm)])


def _axpy_scalar(pairs, scalar, factor):
    with torch.no_grad():
        for p, d in pairs:
            p.add_(d * (factor * scalar.reshape(())))


def _alpha_grad(rows, eps, lr_meta):
    flags = torch.zeros(len(rows), dtype=torch.int32)
    with torch.no_grad():
        for t, (out, da, pos, neg) in enumerate(rows):
            if da is None or pos is None or neg is None:
                out.zero_()
                continue
            h = (pos - neg) / 2. * eps.reshape(())
            if torch.isnan(h).any():
                flags[t] = 1
                out.zero_()
            else:
                out.copy_(da - lr_meta * h)
    return flags


class OracleImpl:
    skip = staticmethod(lambda x, p=None: x)
    pixel_loss = staticmethod(lambda y, gt, kind: torch.nn.functional.mse_loss(y, gt) if kind == 'l2' else torch.nn.functional.l1_loss(y, gt))
    darts_virtual_step = staticmethod(_virtual_step)
    list_norm_eps = staticmethod(_norm_eps)
    list_axpy_scalar = staticmethod(_axpy_scalar)
    darts_alpha_grad = staticmethod(_alpha_grad)
    wb_manual = staticmethod(lambda x, gain: x * gain.view(-1, 3, 1, 1))     # the seam takes the gain (= 5p)
    gamma = staticmethod(O.gamma_manual)
    gtm_manual = staticmethod(O.gtm_manual)
    wb_quadratic = staticmethod(O.wb_quadratic)
    grayworld = staticmethod(lambda x, p=None: O.grayworld(x))
    conditional_fc = staticmethod(lambda img, flat, widths: O.conditional_fc(img, flat, tuple(widths[:-1]), widths[-1]))
    demosaic_nearest = staticmethod(lambda x, p=None: O.demosaic_nearest(x))

    # grouped launches are a launch-count matter of the HIP path; on the seam they are the members one by one
    can_group = staticmethod(lambda modules, x: len(modules) >= 2)
    srcnn_res_group = staticmethod(lambda x, pvs, ms, cache: [OracleImpl.srcnn_res(x, pv, m) for pv, m in zip(pvs, ms)])
    srcnn_demosaic_group = staticmethod(lambda x, ms, cache, record=None: [OracleImpl.srcnn_demosaic(x, m) for m in ms])

    can_fuse_slot = staticmethod(lambda x, names, tensors=(): x.dim() == 4 and x.shape[1] == 3 and len(set(names)) == len(names))

    @staticmethod
    def slot_mix(w, x, entries, w_host=None, stacks=None):
        # the fused slot kernel is a traffic matter of the HIP path; on the seam: the operators one by one, then the mixture
        ops = {'skip': lambda p: x, 'wb_manual': lambda p: OracleImpl.wb_manual(x, p * 5), 'gamma': lambda p: O.gamma_manual(x, p),
               'gtm_manual': lambda p: O.gtm_manual(x, p), 'wb_quadratic': lambda p: O.wb_quadratic(x, p),
               'grayworld': lambda p: O.grayworld(x)}
        outs = [e[1] if e[0] == 'tensor' else ops[e[1]](e[2]) for e in entries]
        return OracleImpl.mix(w, outs)

    @staticmethod
    def mix(w, outs, w_host=None, stacks=None):
        y = 0
        for wk, o in zip(w, outs):
            y = y + o * wk
        return y

    @staticmethod
    def prune_softmax(alpha, threshold, unavailable=None):
        # the reference's own sequence of tensor operations (super_prune_fifteen_demos_four_bayer_two.py:185-193)
        if unavailable is not None:
            alpha = alpha.masked_fill(unavailable.to(torch.bool), float('-inf'))
        probs = torch.softmax(alpha, dim=0)
        below = probs.detach() < threshold * probs.detach().max()
        post = probs.clone()
        post[below] = 0
        return post / post.sum().detach()

    @staticmethod
    def param_blocks(raws, n):
        return [torch.sigmoid(r).repeat(n, 1) for r in raws]

    @staticmethod
    def histc01(x, bins):
        return torch.stack([torch.cat([torch.histc(ch.detach(), bins=bins, min=0, max=1) for ch in im]) for im in x])

    # weights stay attached to the graph while a proxy is being fine-tuned (module.train_weights)
    srcnn_res = staticmethod(lambda x, pv, m: O.srcnn_res(
        x, pv, dict(m.named_parameters()) if getattr(m, 'train_weights', False) else _sd(m)))
    srcnn_demosaic = staticmethod(lambda x, m: O.srcnn_demosaic(x, _sd(m)))
    path14l_bayer = staticmethod(lambda x, m: O.path14l_bayer(x, _sd(m)))
    path14l_bgr = staticmethod(lambda x, m: O.path14l_bgr(x, _sd(m)))

    # classical kernels: the plugin boundary hands over x255 images (scales = (1, 1))
    origin_demosaic = staticmethod(lambda x, option, scales=(1.0, 1.0): O.origin_demosaic(x, option))
    origin_whiteworld = staticmethod(lambda x, ratio, scales=(1.0, 1.0): O.origin_whiteworld(x, _np(ratio)))

    @staticmethod
    def origin_tonemap(x, option, params, scales=(1.0, 1.0)):
        return O.origin_tonemap(x, option, {k: _np(v) for k, v in params.items() if k not in ('input', 'output')})

    @staticmethod
    def origin_denoise(x, option, params, scales=(1.0, 1.0)):
        return O.origin_denoise(x, option, params)


def _np(v):
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v
