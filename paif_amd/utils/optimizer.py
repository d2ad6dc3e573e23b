"""PolyWarmupAdamW -- counterpart of the reference's utils/optimizer.py:3-33 (torch.optim.AdamW with a linear-warm-up /
poly-decay learning rate recomputed inside step()), as ONE multi-tensor HIP kernel over flat arenas.

`ParamArena` lays every parameter out in one flat fp32 buffer (each parameter on whole 1024-float chunks) and re-points
`param.data` into it; a second arena holds the gradients (`param.grad` are views into it, handed out by ops.grad_of when
the wgrad kernels first write a parameter's gradient), two more the Adam moments.  One launch of paif_adamw_step updates
all 44.9 M parameters of Network_MM_Searched(mit_b3); the gradient arena is also what the bucketed RCCL all-reduce of
paif_amd.dist_utils.GradAllReduce works on (contiguous buckets, in the order the backward finishes them).

Semantics kept from torch.optim.AdamW (single-tensor path, eps 1e-8, no amsgrad / maximize): parameters whose .grad is
None are skipped (no weight decay either) -- `classifier.weight` and `decompation.relu.weight` never receive a gradient;
the step count used for the bias corrections is PER PARAMETER (torch's state['step']: it advances only on the steps where
that parameter has a gradient) -- parameters at different counts are updated by separate launches of the same kernel;
`state_dict()` / `load_state_dict()` carry exp_avg / exp_avg_sq / step per parameter in torch.optim.AdamW's own layout
(sliced from / copied into the arenas), so checkpoints move both ways between this class and the reference's.
"""
import ctypes
import math

import numpy as np
import torch

CHUNK = 1024   # floats; the AdamW kernel's unit (256 lanes x float4)


class ParamArena:
    """Flat parameter / gradient / moment buffers.  `params`: list of (parameter, group index), already in the order the
    backward pass completes their gradients (so contiguous gradient buckets become ready one after the other)."""

    def __init__(self, params, with_moments=True):
        assert params, "no parameters"
        dev = params[0][0].device
        for p, _ in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError("ParamArena: every parameter must be float32 on one device")
        self.entries = []           # (param, group, offset, numel)
        off = 0
        for p, g in params:
            n = p.numel()
            self.entries.append((p, g, off, n))
            off += (n + CHUNK - 1) // CHUNK * CHUNK
        self.total = off
        self.nchunks = off // CHUNK
        self.param = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.m = torch.zeros(off, device=dev, dtype=torch.float32) if with_moments else None
        self.v = torch.zeros(off, device=dev, dtype=torch.float32) if with_moments else None
        self.range_of = {}
        with torch.no_grad():
            for p, g, o, n in self.entries:
                view = self.param[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view                                   # the module now reads / the kernel updates the arena in place
                p._paif_grad_view = self.grad[o:o + n].view(p.shape)
                p.grad = None
                self.range_of[id(p)] = (o, o + (n + CHUNK - 1) // CHUNK * CHUNK)

    def zero_grad(self):
        self.grad.zero_()
        for p, _, _, _ in self.entries:
            p.grad = None

    def adopt_foreign_grads(self):
        """Gradients torch autograd (or the user) attached as separate tensors are copied into the arena."""
        for p, _, _, _ in self.entries:
            if p.grad is not None and p.grad.data_ptr() != p._paif_grad_view.data_ptr():
                p._paif_grad_view.copy_(p.grad)
                p.grad = p._paif_grad_view

    def chunk_groups(self):
        """uint8 [nchunks]: parameter group of each chunk, 255 = no gradient this step (skipped, like torch)."""
        cg = np.full(self.nchunks, 255, dtype=np.uint8)
        for p, g, o, n in self.entries:
            if p.grad is not None:
                cg[o // CHUNK:(o + n + CHUNK - 1) // CHUNK] = g
        return cg


def arena_order(params):
    """Order parameters for the arena: by `_paif_order` (assigned by the models in backward-completion order, see
    paif_amd.core.model_fusion_auto.assign_grad_order) where present, otherwise reversed registration order (the last
    layers finish first); parameters flagged `_paif_never_grad` go to the tail, outside every all-reduce bucket."""
    idx = {id(p): i for i, (p, _) in enumerate(params)}
    n = len(params)

    def key(item):
        p = item[0]
        never = 1 if getattr(p, "_paif_never_grad", False) else 0
        return (never, getattr(p, "_paif_order", 10 ** 9 + (n - idx[id(p)])))

    return sorted(params, key=key)


class PolyWarmupAdamW(torch.optim.AdamW):
    """utils/optimizer.py:3-33.  Same constructor and schedule; step() is the paif_adamw_step kernel over the arenas."""

    def __init__(self, params, lr, weight_decay, betas, warmup_iter=None, max_iter=None, warmup_ratio=None, power=None):
        super().__init__(params, lr=lr, betas=betas, weight_decay=weight_decay, eps=1e-8)
        self.global_step = 0
        self.warmup_iter = warmup_iter
        self.warmup_ratio = warmup_ratio
        self.max_iter = max_iter
        self.power = power
        self.__init_lr = [group['lr'] for group in self.param_groups]
        self._arena = None
        self._cg_cache = {}
        self._steps = None          # torch's per-parameter state['step'] (int64 per arena entry), independent of global_step
        self._pending_state = None  # a state_dict loaded before the arenas exist
        if len(self.param_groups) > 8:
            raise NotImplementedError("PolyWarmupAdamW: at most 8 parameter groups (the reference uses 3)")

    # ---- schedule (utils/optimizer.py:17-28) ----
    def lr_mult(self):
        if self.global_step < self.warmup_iter:
            return 1 - (1 - self.global_step / self.warmup_iter) * (1 - self.warmup_ratio)
        if self.global_step < self.max_iter:
            return (1 - self.global_step / self.max_iter) ** self.power
        return None

    # ---- arenas ----
    def arena(self):
        if self._arena is None:
            pairs = [(p, gi) for gi, g in enumerate(self.param_groups) for p in g['params'] if p.requires_grad]
            if not pairs[0][0].is_cuda:
                raise RuntimeError("PolyWarmupAdamW: parameters must be on the GPU before the first zero_grad()/step() "
                                   "(there is no CPU optimizer path)")
            self._arena = ParamArena(arena_order(pairs))
            self._steps = np.zeros(len(self._arena.entries), dtype=np.int64)
            from ..operations_m import invalidate_weight_caches
            invalidate_weight_caches()          # param.data moved
            if self._pending_state is not None:
                st, self._pending_state = self._pending_state, None
                self._restore_state(st)
        return self._arena

    # ---- checkpoint / resume: torch.optim.AdamW's layout (state[i] = {step, exp_avg, exp_avg_sq}, i = position in the
    # flattened param_groups) -- the moments live in the arenas, not in self.state ----
    def _flat_params(self):
        return [p for g in self.param_groups for p in g['params']]

    def state_dict(self):
        flat = self._flat_params()
        index = {id(p): i for i, p in enumerate(flat)}
        groups, start = [], 0
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != 'params'}
            d['params'] = list(range(start, start + len(g['params'])))
            start += len(g['params'])
            groups.append(d)
        state = {}
        if self._pending_state is not None:
            state = {i: dict(v, step=torch.tensor(float(v['step']))) for i, v in self._pending_state.items()}
        elif self._arena is not None:
            A = self._arena
            for e, (p, _, o, n) in enumerate(A.entries):
                if self._steps[e] > 0:
                    state[index[id(p)]] = {'step': torch.tensor(float(self._steps[e])),
                                           'exp_avg': A.m[o:o + n].view(p.shape).clone(),
                                           'exp_avg_sq': A.v[o:o + n].view(p.shape).clone()}
        return {'state': state, 'param_groups': groups}

    def _restore_state(self, state):
        A, flat = self._arena, self._flat_params()
        pos = {id(p): e for e, (p, _, _, _) in enumerate(A.entries)}
        A.m.zero_(); A.v.zero_(); self._steps[:] = 0
        with torch.no_grad():
            for i, st in state.items():
                p = flat[int(i)]
                if id(p) not in pos:
                    raise KeyError("load_state_dict: state for parameter %d, which this optimizer does not update" % int(i))
                e = pos[id(p)]
                _, _, o, n = A.entries[e]
                if tuple(st['exp_avg'].shape) != tuple(p.shape) or tuple(st['exp_avg_sq'].shape) != tuple(p.shape):
                    raise ValueError("load_state_dict: moment shape %s does not match parameter %d %s"
                                     % (tuple(st['exp_avg'].shape), int(i), tuple(p.shape)))
                A.m[o:o + n].view(p.shape).copy_(st['exp_avg'])
                A.v[o:o + n].view(p.shape).copy_(st['exp_avg_sq'])
                self._steps[e] = int(round(float(st['step'])))

    def load_state_dict(self, state_dict):
        groups = state_dict['param_groups']
        if len(groups) != len(self.param_groups) or any(len(a['params']) != len(b['params']) for a, b in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters per group")
        for g, loaded in zip(self.param_groups, groups):
            for k, v in loaded.items():
                if k != 'params':
                    g[k] = v
        # 'step' is read NOW (a caller -- e.g. a torch optimizer the same dict was also loaded into -- may go on incrementing that
        # tensor in place); the moment tensors are referenced until the arenas exist, then copied
        state = {int(k): dict(v, step=float(v['step'])) for k, v in state_dict['state'].items()}
        if self._arena is None:
            self._pending_state = state         # parameters may still be on the host: applied when the arenas are built
        else:
            self._restore_state(state)

    def zero_grad(self, set_to_none=True):
        """One memset of the gradient arena; .grad becomes None (torch's set_to_none semantics) until the next backward."""
        self.arena().zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        m = self.lr_mult()
        if m is not None:
            for i in range(len(self.param_groups)):
                self.param_groups[i]['lr'] = self.__init_lr[i] * m
        from .. import _lib, ops
        from ..operations_m import invalidate_weight_caches
        A = self.arena()
        A.adopt_foreign_grads()
        g0 = self.param_groups[0]
        for g in self.param_groups:
            if g['betas'] != g0['betas'] or g['eps'] != g0['eps'] or g.get('amsgrad') or g.get('maximize'):
                raise NotImplementedError("PolyWarmupAdamW: per-group betas / eps, amsgrad and maximize are not built")
        has_grad = np.array([p.grad is not None for p, _, _, _ in A.entries], dtype=bool)
        if ops.CONFIG["wgrad_f16x3"] and ops.CONFIG.get("wgrad_guard", True) and has_grad.any() and not torch.cuda.is_current_stream_capturing():
            # the Linear weight gradients of a training step run on fp16 pairs with dY pre-scaled by a power of two
            # guessed from the pixel count (ops.wgrad_scale).  A loss multiplier / sum reduction / GradScaler-style scaling can push
            # |dY| * scale past 65504: the kernel then accumulates inf / NaN.  One pass over the gradient arena and one host read per
            # step, BEFORE anything is written: a non-finite gradient raises instead of reaching the weights and the Adam moments
            if not bool(torch.isfinite(A.grad).all()):
                raise FloatingPointError(
                    "PolyWarmupAdamW.step: non-finite gradient in the arena; nothing was updated.  If the loss is scaled (multiplier, "
                    "sum reduction, GradScaler) the fp16-pair weight-gradient GEMMs may have left fp16's exponent range: set "
                    "ops.CONFIG['wgrad_f16x3'] = False (or PAIF_WGRAD_F16X3=0) to run them on the exact fp32 MFMA kernels")
        self._steps[has_grad] += 1                      # torch: state['step'] += 1 for the parameters updated now
        beta1, beta2 = g0['betas']
        ng = len(self.param_groups)
        decay = (ctypes.c_float * ng)(*[1 - g['lr'] * g['weight_decay'] for g in self.param_groups])
        # one launch per distinct step count among the parameters that have a gradient (normally exactly one)
        for t in np.unique(self._steps[has_grad]).tolist():
            cg = np.full(A.nchunks, 255, dtype=np.uint8)
            for e, (p, g, o, n) in enumerate(A.entries):
                if has_grad[e] and self._steps[e] == t:
                    cg[o // CHUNK:(o + n + CHUNK - 1) // CHUNK] = g
            key = cg.tobytes()
            if key not in self._cg_cache:
                if len(self._cg_cache) > 8:
                    self._cg_cache.clear()
                self._cg_cache[key] = torch.from_numpy(cg).to(A.param.device)
            bc1 = 1 - beta1 ** t
            bc2_sqrt = math.sqrt(1 - beta2 ** t)
            step_size = (ctypes.c_float * ng)(*[g['lr'] / bc1 for g in self.param_groups])
            _lib.check(ops.lib().paif_adamw_step(ops._p(A.param), ops._p(A.grad), ops._p(A.m), ops._p(A.v),
                                                 ctypes.c_void_p(self._cg_cache[key].data_ptr()), A.nchunks, ng, decay, step_size,
                                                 1 - beta1, beta2, 1 - beta2, bc2_sqrt, g0['eps'], ops._stream()), "adamw_step")
        self.global_step += 1
        invalidate_weight_caches()                     # the kernel wrote the weights behind torch's version counters
        return loss


class PolyWarmupAdamW_seg(PolyWarmupAdamW):
    """utils/optimizer.py:36-66: the same optimizer started at schedule position `iter_curr` (resuming a run)."""

    def __init__(self, params, lr, weight_decay, betas, iter_curr, warmup_iter=None, max_iter=None, warmup_ratio=None, power=None):
        super().__init__(params, lr, weight_decay, betas, warmup_iter=warmup_iter, max_iter=max_iter, warmup_ratio=warmup_ratio,
                         power=power)
        self.global_step = iter_curr
