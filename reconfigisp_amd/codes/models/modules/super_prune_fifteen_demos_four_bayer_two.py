"""SuperPruneFifteenDemosFourBayerTwo - the DARTS super-net with online pruning.

Slot 0: 2 Bayer ops, slot 1: 4 demosaic ops, slots 2..n_step+1: 15 sRGB ops each.  Mirror of
models/modules/super_prune_fifteen_demos_four_bayer_two.py:13-230: same constructor, forward
semantics (softmax -> prune strictly below threshold*max on detached probs -> renormalise by the
detached sum -> weighted sum, 'dummy gradient' for pruned parametrised ops), properties and
state-dict keys (alpha_bayer, alpha_demosaic, alpha_step<k>, param_step<k>_<name>).
The weighted sum of a slot is one fused kernel (risp_mix_fwd/bwd) instead of K multiply-adds.
"""
import logging
import os

import torch
import torch.nn as nn

from .... import functional as F
from ....isp_kernels import demosaic as _dm
from . import registry as R
from . import tools_origin as T
from . import tools_proxy as TP


# HIP streams the ops of a slot are spread over when the batch is small (1 = off), and what "small" means
FUSE_SLOT = os.environ.get('RISP_FUSE_SLOT', '1') != '0'     # element-wise operators evaluated inside the mixture kernel
# module class -> name of the element-wise operator in functional.SLOT_KINDS
_POINTWISE = {T.Skip: 'skip', T.WbManual: 'wb_manual', T.Gamma: 'gamma', T.GtmManual: 'gtm_manual', T.WbQuadratic: 'wb_quadratic',
              T.Grayworld: 'grayworld'}
SLOT_STREAMS = int(os.environ.get('RISP_SLOT_STREAMS', '2'))
SLOT_STREAMS_MAX_PIXELS = 1 << 40
SLOT_STREAMS_MIN_PIXELS = 1 << 16           # batch x H x W below which a second stream buys nothing (batch 4 of 48 x 48: 19-20 ms per iteration either way)
SLOT_STREAMS_MIN_JOBS = 3                   # jobs of a slot from which the streams are used
# (Round 6 tried two jobs - the grouped proxies and Path-Restore of an sRGB slot side by side - on planes of at least 2^20 pixels: config 3
# 0.354 -> 0.350 s in alternating runs, but test_search_network_full_size_gradient_properties (two backward passes of the same graph give the
# same bits) then failed once in four long suite runs and never alone: not kept.)

class SuperPruneFifteenDemosFourBayerTwo(nn.Module):
    def __init__(self, n_step, threshold, module_path):
        super().__init__()
        self.threshold = threshold
        self.middle_results = None
        self.pruned_paths = [0] * (n_step + 2)     # bayer, demosaic, then the sRGB steps
        self.all_modules, self.all_params, self.all_alphas = [], [], []
        self.trainable_params = []
        self.slot_names = []
        self._weight_cache = {}
        self._reuse = None          # step-level reuse scope (begin_reuse / end_reuse)

        def empty():
            return nn.Parameter(torch.zeros(0))

        for alpha_key, names in (('alpha_bayer', R.NAMES_BAYER), ('alpha_demosaic', R.NAMES_DEMOSAIC)):
            setattr(self, alpha_key, nn.Parameter(torch.zeros(len(names))))
            self.all_modules.append([R.make_op(n, module_path) for n in names])
            self.all_params.append([empty() for _ in names])
            self.all_alphas.append(getattr(self, alpha_key))
            self.slot_names.append(list(names))

        for k in range(1, n_step + 1):
            mods, pars = [], []
            for name in R.NAMES_SRGB:
                mods.append(R.make_op(name, module_path))
                init = R.PARAM_INIT[name]
                if init:
                    key = 'param_step{}_{}'.format(k, name)
                    setattr(self, key, nn.Parameter(torch.tensor(init, dtype=torch.float32)))
                    pars.append(getattr(self, key))
                else:
                    pars.append(empty())
            setattr(self, 'alpha_step{}'.format(k), nn.Parameter(torch.zeros(len(R.NAMES_SRGB))))
            self.trainable_params += pars          # includes the zero-size placeholders (:163)
            self.all_modules.append(mods)
            self.all_params.append(pars)
            self.all_alphas.append(getattr(self, 'alpha_step{}'.format(k)))
            self.slot_names.append(list(R.NAMES_SRGB))
        self.param_and_alpha = self.trainable_params + self.all_alphas

    def _apply(self, fn, *args, **kwargs):
        # ops and zero-size placeholders are kept in plain lists (only alphas / param_step* are
        # registered, as in the reference); make them follow .to()/.cuda() all the same
        super()._apply(fn, *args, **kwargs)
        self._weight_cache.clear()
        for mods in self.all_modules:
            for m in mods:
                m._apply(fn, *args, **kwargs)
        for pars in self.all_params:
            for p in pars:
                if p.numel() == 0:
                    p.data = fn(p.data)
        return self

    def load_state_dict(self, *args, **kwargs):
        self._weight_cache.clear()             # keyed on alpha._version, which copy_ bumps - and writes through .data do not
        return super().load_state_dict(*args, **kwargs)

    def invalidate_weight_cache(self):
        """call after writing an alpha through ``.data`` (no version bump): the host copy of the mixture weights is stale"""
        self._weight_cache.clear()

    def _unavailable(self, mods, device):
        """uint8 mask of ops that cannot run in this build (DemosaicNet without a registered implementation): they are
        left out of the softmax - probability exactly 0, zero alpha-gradient - instead of crashing the search.  With an
        implementation registered (isp_kernels.demosaic.register_demosaicnet) the reference behaviour applies
        unchanged.  None when every op of the slot is available."""
        if _dm.demosaicnet_available():
            return None
        mask = [isinstance(m, T.DemosaicNet) for m in mods]
        if not any(mask):
            return None
        if not getattr(self, '_warned_unavailable', False):
            logging.getLogger('base').warning('DemosaicNet has no implementation in this build; its mixing '
                                              'probability is fixed to 0')
            self._warned_unavailable = True
        cache = self.__dict__.setdefault('_unavailable_masks', {})
        key = (tuple(mask), str(device))
        if key not in cache:
            cache[key] = torch.tensor(mask, dtype=torch.uint8, device=device)
        return cache[key]

    # ------------------------------------------------------------------ step-level reuse
    def begin_reuse(self):
        """Until end_reuse(): forwards whose input batch, alphas and (for the ops concerned) parameters are unchanged take
        the outputs and saved activations of the parameter-free CNN ops from the first such forward instead of
        recomputing them.  DartsModel opens the scope around the architecture step: forwards #1, #3 and #4 of an
        iteration (darts_model.py:182-222, 270-324) see the same train batch and the same alphas, slots 0-1 have no
        trainable parameters, and the Path-Restore op of the first sRGB slot sees an identical input.  Identity is
        tracked by value tokens: (data_ptr, _version) of the network input, then per slot (input token, alpha version,
        versions of the slot's parameters) - the finite-difference shifts of the parameters bump their versions, so
        everything downstream of a shifted parameter misses the cache by construction.  Results are bit-identical."""
        self._reuse = {}

    def end_reuse(self):
        self._reuse = None
        for mods in self.all_modules:
            for m in mods:
                m.__dict__.pop('_risp_reuse', None)

    def params_only_backward(self, on, params=None):
        """While on: the next backward pass is asked for the PARAMETER gradients only (DartsModel.virtual_step,
        darts_model.py:204 `autograd.grad(loss, trainable_parameters)`).  The slots below the first parametrised one hold no
        parameters, so the input gradient of that slot's operators feeds nothing: the grouped SRCNNRes backward skips its
        9x9 64->3 backward-data convolution and the member sum (autograd already skips the nodes that lead only to alphas).
        ``params``: the tensors the pass will ask for - the switch stays OFF unless every one of them is an operator parameter of
        the first parametrised slot or a later one, and nothing below that slot requires a gradient (proxy weights being
        fine-tuned in the graph, say): an undefined input gradient would otherwise reach a consumer silently."""
        first = next((s for s, pars in enumerate(self.all_params) if any(p.numel() for p in pars)), None)
        if first is None:
            return False
        if on:
            later = {id(p) for pars in self.all_params[first:] for p in pars}
            below = [p for mods in self.all_modules[:first] for m in mods for p in m.parameters() if p.requires_grad]
            if below or (params is not None and any(id(p) not in later for p in params)):
                on = False
        self.__dict__.setdefault('_group_cache', {}).setdefault(first, {})['skip_gx'] = bool(on)
        return bool(on)

    def _record(self, slot, k, token, mod):
        """the reuse record of op k of `slot` for this input token (None outside a scope / for ops with parameters)"""
        if self._reuse is None or token is None:
            mod.__dict__.pop('_risp_reuse', None)
            return None
        rec = self._reuse.setdefault((slot, k, token), {})
        mod.__dict__['_risp_reuse'] = rec
        return rec

    def _jobs(self, slot, mods, index, args, x, token=None, skip=()):
        """The surviving ops of a slot as launch jobs [(positions in `index`, callable -> list of outputs)].  Same-geometry
        proxies - the SRCNNRes family of an sRGB slot (:35-52), the two proxy demosaics - form ONE job that runs every layer
        as a single grouped launch (convnets.srcnn_res_group); everything else is a job of its own.  Heavy jobs come first
        so that the round-robin over the streams puts the group and Path-Restore on different ones."""
        families = ((TP.ProxyNet, 'res'), (TP.ProxyDemosaicNet, 'demosaic'))
        jobs, taken = [], set(skip)               # `skip`: positions the fused mixture evaluates itself
        cache = self.__dict__.setdefault('_group_cache', {}).setdefault(slot, {})
        for cls, kind in families:
            pos = [i for i, k in enumerate(index) if isinstance(mods[k], cls)]
            members = [mods[index[i]] for i in pos]
            if len(pos) >= 2 and F.can_group(members, x):
                if kind == 'res':
                    fn = (lambda xj, members=members, pos=pos: F.srcnn_res_group(xj, [args[i] for i in pos], members, cache))
                else:
                    rec = self._record(slot, index[pos[0]], token, members[0])
                    fn = (lambda xj, members=members, rec=rec: F.srcnn_demosaic_group(xj, members, cache, rec))
                jobs.append((pos, fn))
                taken.update(pos)
        heavy = [i for i in range(len(index)) if i not in taken and isinstance(mods[index[i]], (TP.PathRestore14lBgr,
                                                                                                  TP.PathRestore14lBayer))]
        rest = [i for i in range(len(index)) if i not in taken and i not in heavy]
        for i in heavy:
            self._record(slot, index[i], token, mods[index[i]])      # read by F.path14l_* through the module
        single = [([i], (lambda xj, i=i: [mods[index[i]](xj, args[i])])) for i in heavy + rest]
        # group, Path-Restore, then the light ops: with two streams the two heavy jobs land on different streams
        return jobs[:1] + single[:len(heavy)] + jobs[1:] + single[len(heavy):]

    def _run_jobs(self, jobs, n_out, x, args, xs=None):
        """outs[i] for every surviving op; ``xs``: the slot input per job (aliases of x with a one-launch gradient sum,
        functional.fan_out).  The jobs of a slot are independent given the slot input, so they are issued
        round-robin on two HIP streams: launches of different jobs overlap and fill each other's gaps - a convolution
        launch runs its workgroups in lockstep rounds and leaves the matrix pipes idle through each prologue / store
        drain.  The backward pass inherits the streams (autograd runs a node on the stream of its forward).  Same
        kernels, same arguments, same summation order in the mixture: results are bit-identical to the single-stream
        order.  Measured in round 2 (tools/bench_darts.py, n_step 2, op-by-op jobs): batch 4 0.113 -> 0.097 s per
        iteration, batch 32 0.62 -> 0.60 s; 3 and 4 streams are no faster."""
        pixels = x.shape[0] * x.shape[2] * x.shape[3]
        n_streams = SLOT_STREAMS if (x.is_cuda and len(jobs) >= SLOT_STREAMS_MIN_JOBS and
                                     SLOT_STREAMS_MIN_PIXELS <= pixels <= SLOT_STREAMS_MAX_PIXELS) else 1
        outs = [None] * n_out
        xs = xs if xs is not None else [x] * len(jobs)
        if n_streams <= 1:
            for (pos, fn), xj in zip(jobs, xs):
                for i, o in zip(pos, fn(xj)):
                    outs[i] = o
            return outs
        main = torch.cuda.current_stream()
        pool = self.__dict__.setdefault('_side_streams', {})
        key = (x.device.index, n_streams)
        if key not in pool:
            pool[key] = [torch.cuda.Stream(device=x.device) for _ in range(n_streams - 1)]
        streams = [main] + pool[key]
        for s in streams[1:]:
            s.wait_stream(main)                           # the slot input (and the parameter blocks) are ready
            x.record_stream(s)
        for j, (pos, fn) in enumerate(jobs):
            s = streams[j % n_streams]
            if s is not main:
                for i in pos:
                    if args[i] is not None:
                        args[i].record_stream(s)
            with torch.cuda.stream(s):
                res = fn(xs[j])
            for i, o in zip(pos, res):
                if s is not main and o is not x:
                    o.record_stream(main)                 # consumed by the mixture kernel on the main stream
                outs[i] = o
        for s in streams[1:]:
            main.wait_stream(s)
        return outs

    def _select(self, post, index, k):
        """post[index] for the surviving ops.  A leading run is a view; anything else goes through index_select with a
        cached device index (indexing with a Python list uploads the list on every call and sorts in its backward)."""
        if len(index) == k:
            return post
        if index == list(range(len(index))):
            return post[:len(index)]
        cache = self.__dict__.setdefault('_index_cache', {})
        key = (tuple(index), str(post.device))
        if key not in cache:
            cache[key] = torch.tensor(index, dtype=torch.long, device=post.device)
        return post.index_select(0, cache[key])

    def _mixture_weights(self):
        """(post, host weights) of every slot.  post = pruned, renormalised softmax(alpha) (one launch per slot); the host
        copy (the reference's .item(): a device synchronisation) is kept per VALUE of alpha - four of the five forwards of
        a DARTS iteration see unchanged logits (in-place updates bump _version) - and the slots whose copy is stale are
        fetched with ONE device-to-host transfer for the whole network instead of one per slot: each transfer drains the
        launch queue, and at the per-GPU batch of the 8-GPU search the host needs ~0.2 ms to refill it."""
        posts, keys, stale = [], [], []
        for slot, (mods, alpha) in enumerate(zip(self.all_modules, self.all_alphas)):
            unavailable = self._unavailable(mods, alpha.device)
            posts.append(F.prune_softmax(alpha, self.threshold, unavailable))
            keys.append((alpha._version, alpha.data_ptr(), self.threshold, unavailable is None))
            cached = self._weight_cache.get(slot)
            if cached is None or cached[0] != keys[slot]:
                stale.append(slot)
        if stale:
            flat = torch.cat([posts[s].detach() for s in stale]).cpu().tolist() if len(stale) > 1 else \
                posts[stale[0]].detach().cpu().tolist()
            at = 0
            for s in stale:
                k = posts[s].numel()
                self._weight_cache[s] = (keys[s], flat[at: at + k])
                at += k
        return posts, [self._weight_cache[s][1] for s in range(len(posts))]

    def forward(self, x):
        n = x.size(0)
        self.middle_results = []
        token = (x.data_ptr(), x._version, tuple(x.shape)) if self._reuse is not None else None
        posts, host_weights = self._mixture_weights()
        for slot, (mods, pars, alpha) in enumerate(zip(self.all_modules, self.all_params, self.all_alphas)):
            # softmax -> strict-< prune against threshold * max (detached) -> renormalise by the detached sum (:185-193)
            post, weights = posts[slot], host_weights[slot]
            self.pruned_paths[slot] = sum(1 for w in weights if w == 0.0)

            index, pruned_pars, live_pars = [], [], []
            for k, par in enumerate(pars):
                if weights[k] < 1e-9:
                    if par.nelement() > 0:               # pruned, but must still receive a (zero) gradient
                        pruned_pars.append(par)
                    continue
                index.append(k)
                if par.nelement() > 0:
                    live_pars.append(par)
            blocks = iter(F.param_blocks(live_pars, n))  # sigmoid(par).repeat(n, 1) of every surviving op: one launch
            args = [next(blocks) if pars[k].nelement() > 0 else None for k in index]
            # element-wise operators (gamma, white balances, tone curve, gray world, skip: :195-210) are not run at all -
            # the mixture kernel evaluates them from the slot input in registers (F.slot_mix)
            fused = {}
            if FUSE_SLOT and x.dim() == 4 and x.shape[1] == 3:
                names = {i: _POINTWISE[type(mods[k])] for i, k in enumerate(index) if type(mods[k]) in _POINTWISE}
                if len(names) >= 1 and F.can_fuse_slot(x, list(names.values())):
                    fused = names
            jobs = self._jobs(slot, mods, index, args, x, token, skip=set(fused))
            if token is not None:       # value token of this slot's output
                token = (token, slot, alpha.data_ptr(), alpha._version, tuple((p.data_ptr(), p._version) for p in live_pars))
            # every consumer of the slot input gets its own alias: their gradients are added by one launch instead of autograd's
            # pairwise additions
            xs = F.fan_out(x, len(jobs) + (1 if fused else 0))
            outs = self._run_jobs(jobs, len(index), x, args, xs[:len(jobs)])
            sel = self._select(post, index, len(weights))
            stacks = [pos for pos, _ in jobs if len(pos) > 1]
            if fused and not F.can_fuse_slot(x, list(fused.values()), [outs[i] for i in range(len(index)) if i not in fused]):
                # an operand view that is not 16-byte aligned: the element-wise operators as their own launches, plain mixture
                for i, nm in fused.items():
                    outs[i] = mods[index[i]](x, args[i])
                fused = {}
            if fused:
                entries = [('op', fused[i], args[i]) if i in fused else ('tensor', outs[i]) for i in range(len(index))]
                y = F.slot_mix(sel, xs[-1], entries, w_host=[weights[k] for k in index], stacks=stacks)
            else:
                y = F.mix(sel, outs, w_host=[weights[k] for k in index], stacks=stacks)
            if pruned_pars:
                y = F.attach_zero_grad(y, pruned_pars)
            self.middle_results.append(y)
            x = y
        return x

    @property
    def trainable_parameters(self):
        return self.trainable_params

    @property
    def parameters_and_alpha(self):
        return self.param_and_alpha

    @property
    def alphas(self):
        return self.all_alphas

    @property
    def intermediate_results(self):
        return self.middle_results
