"""PolyWarmupAdamW -- counterpart of the reference's utils/optimizer.py:3-33 (torch.optim.AdamW with a linear-warm-up /
poly-decay learning rate recomputed inside step()), as ONE multi-tensor HIP kernel over flat arenas.

`ParamArena` lays every parameter out in one flat fp32 buffer (each parameter on whole 1024-float chunks) and re-points
`param.data` into it; a second arena holds the gradients (`param.grad` are views into it, handed out by ops.grad_of when
the wgrad kernels first write a parameter's gradient), two more the Adam moments.  One launch of paif_adamw_step updates
all 44.9 M parameters of Network_MM_Searched(mit_b3); the gradient arena is also what the bucketed RCCL all-reduce of
paif_amd.dist_utils.GradAllReduce works on (contiguous buckets, in the order the backward finishes them).

Semantics kept from torch.optim.AdamW (single-tensor path, eps 1e-8, no amsgrad / maximize): parameters whose .grad is
None are skipped (no weight decay either) -- `classifier.weight` and `decompation.relu.weight` never receive a gradient;
the step count used for the bias corrections is the number of step() calls.
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
        self._cg_cache = (None, None)
        self._adam_t = 0            # torch's per-parameter state['step']: counts step() calls, independent of global_step
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
            from ..operations_m import invalidate_weight_caches
            invalidate_weight_caches()          # param.data moved
        return self._arena

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
        cg = A.chunk_groups()
        key = cg.tobytes()
        if self._cg_cache[0] != key:
            self._cg_cache = (key, torch.from_numpy(cg).to(A.param.device))
        self._adam_t += 1
        t = self._adam_t                               # bias corrections: step count = number of step() calls
        beta1, beta2 = g0['betas']
        bc1 = 1 - beta1 ** t
        bc2_sqrt = math.sqrt(1 - beta2 ** t)
        ng = len(self.param_groups)
        decay = (ctypes.c_float * ng)(*[1 - g['lr'] * g['weight_decay'] for g in self.param_groups])
        step_size = (ctypes.c_float * ng)(*[g['lr'] / bc1 for g in self.param_groups])
        _lib.check(ops.lib().paif_adamw_step(ops._p(A.param), ops._p(A.grad), ops._p(A.m), ops._p(A.v),
                                             ctypes.c_void_p(self._cg_cache[1].data_ptr()), A.nchunks, ng, decay, step_size,
                                             1 - beta1, beta2, 1 - beta2, bc2_sqrt, g0['eps'], ops._stream()), "adamw_step")
        self.global_step += 1
        invalidate_weight_caches()                     # the kernel wrote the weights behind torch's version counters
        return loss
