"""NAS-cell primitive operators -- MI355X-native counterparts of the reference's operations_m.py.

Same class names, constructor signatures, OPS registry, sub-module names and state_dict keys as
the reference (operations_m.py:9-18, 114-145, 340-393, 435-525), so a reference checkpoint loads
with strict=True and `OPS[name](C, kernel, dilation, affine)` is a drop-in.  forward(x) takes and
returns [B,C,H,W] tensors like the reference; the arithmetic runs in the hand-written gfx950
kernels of libpaif_hip.so (paif_amd/csrc), in NHWC (= torch channels_last) internally.

There is NO torch/eager fallback: the parameter-holder modules below raise if ever called.
"""
import torch
import torch.nn as nn

from . import ops


# ---------------------------------------------------------------------------------------------
# parameter holders: same constructors / init / state_dict keys as the torch layers the reference
# uses, but never executed -- the HIP kernels read their tensors.
# ---------------------------------------------------------------------------------------------
class _NeverCalled:
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("%s is a parameter holder; the HIP path must be used" % type(self).__name__)


class Conv2dParams(_NeverCalled, nn.Conv2d):
    pass


class Conv1dParams(_NeverCalled, nn.Conv1d):
    pass


class BatchNormParams(_NeverCalled, nn.BatchNorm2d):
    pass


class PReLUParams(_NeverCalled, nn.PReLU):
    pass


class LinearParams(_NeverCalled, nn.Linear):
    pass


class LayerNormParams(_NeverCalled, nn.LayerNorm):
    pass


def invalidate_weight_caches():
    """Drop every packed-weight / folded-BN cache entry at its next use.  The caches are keyed on (data_ptr, tensor version):
    in-place torch ops and load_state_dict bump the version, but writes through `param.data` and raw-pointer kernels (the
    AdamW kernel of paif_amd.utils.optimizer, which calls this after every step) do not -- call this after any such write."""
    ops.CONFIG["weights_generation"] = ops.CONFIG.get("weights_generation", 0) + 1


class _PackCache:
    """Packed-weight cache: one entry per (name, conv precision), valid for (data_ptr, version, device) of the source parameter(s) +
    the global weights generation (invalidate_weight_caches).  The precision is part of the SLOT, not of the validity key: the attack
    loops switch the arithmetic (ops.attack_arithmetic: bf16x3 <-> bf16x6) around every call, and both packs must survive that --
    otherwise every attack + clean forward rebuilt ~150 packs twice per batch, and under a hipGraph capture the rebuilt pack would
    live in capture-pool memory that holds garbage until the first replay."""

    def __init__(self):
        self._store = {}

    def get(self, name, params, builder):
        slot = (name, ops.pack_precision())      # ("f16" inside an fp16-storage inference forward)
        key = tuple((p.data_ptr(), p._version, str(p.device)) for p in params) + (ops.CONFIG.get("weights_generation", 0),)
        hit = self._store.get(slot)
        if hit is None or hit[0] != key:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("weight pack %r (%s) is missing while a hipGraph is being captured: run one eager step in this "
                                   "arithmetic first (the pack kernels and their buffers must not live inside the capture)" % slot)
            hit = (key, builder())
            self._store[slot] = hit
        return hit[1]


def _pad_for(k, d):
    # operations_m.py:121-132 -- anything outside the table gets padding 0 in the reference, which would
    # change the output size; the HIP conv is "same"-padded, so refuse other combinations.
    table = {(3, 1): 1, (3, 2): 2, (5, 1): 2, (5, 2): 4, (7, 1): 3, (7, 2): 6, (1, 1): 0}
    if (k, d) not in table:
        raise NotImplementedError("BasicConv kernel %d dilation %d is outside the reference's padding table" % (k, d))
    return table[(k, d)]


_slope_checked = {}


def check_positive_slope(param, what):
    """Backward kernels that read the sign of a PReLU OUTPUT (instead of a saved pre-activation) need a
    non-negative slope.  One host sync per parameter version."""
    key = (param.data_ptr(), param._version, ops.CONFIG.get("weights_generation", 0))
    if _slope_checked.get(id(param)) != key:
        if float(param.detach().min()) < 0.0:
            raise NotImplementedError("%s: negative PReLU slope is not supported by the backward kernels" % what)
        _slope_checked[id(param)] = key


class _HipOp(nn.Module):
    """Base: NCHW boundary <-> NHWC body."""

    def __init__(self):
        super().__init__()
        self._packs = _PackCache()

    def forward(self, x):
        wg = ops.want_param_grads(self)
        if torch.is_grad_enabled() and (x.requires_grad or wg):
            return _OpFn.apply(x, self, wg, grad_anchor(x.device))
        with torch.no_grad():
            return ops.to_nchw_view(self.forward_nhwc(ops.to_nhwc(x)))

    def forward_nhwc(self, x, res=(), tape=None):
        raise NotImplementedError

    def backward_nhwc(self, g, tape, wgrad=False):
        """g = d/d(output) -> d/d(x), INCLUDING the op's own residual path when it has one.  wgrad: also accumulate the
        parameter gradients into ops.grad_of(param) (the tape must have been recorded under ops.tape_mode("wgrad"))."""
        raise NotImplementedError("%s: backward kernels not built" % type(self).__name__)

    def _dgrad_w(self, name, w, coff=0, cs=32):
        return self._packs.get("dg_" + name, [w], lambda: ops.pack_conv_dgrad_weight(w, coff, cs))

    @staticmethod
    def _add_res(y, res):
        for r in res:
            y = ops.add(y, r)
        return y


_ANCHORS = {}


def grad_anchor(device):
    """A 0-d tensor that requires grad: handed to the autograd nodes so that their backward runs (and deposits the
    parameter gradients into param.grad) even when no INPUT requires grad -- the training step's case."""
    a = _ANCHORS.get(str(device))
    if a is None:
        a = _ANCHORS[str(device)] = torch.zeros((), device=device, requires_grad=True)
    return a


class _OpFn(torch.autograd.Function):
    """Autograd node of one NAS-cell operator: hand-written reverse pass; parameter gradients (when `wgrad`) are
    accumulated straight into param.grad by the wgrad kernels, like loss.backward() does in the reference."""

    @staticmethod
    def forward(ctx, x, module, wgrad, anchor):
        tape = []
        with ops.tape_mode("wgrad" if wgrad else "dgrad"):
            y = module.forward_nhwc(ops.to_nhwc(x.detach()), (), tape)
        # single ops record one entry; containers (Cell_Chain) consume the whole list
        ctx.entry = tape if getattr(module, "_tape_is_list", False) else tape[0]
        ctx.module, ctx.wgrad = module, wgrad
        return ops.to_nchw_view(y)

    @staticmethod
    def backward(ctx, g):
        d = ctx.module.backward_nhwc(ops.to_nhwc(g), ctx.entry, ctx.wgrad)
        ctx.entry = None
        return ops.to_nchw_view(d), None, None, None


class BasicConv(_HipOp):
    """operations_m.py:114-145.  On the hot path always relu=False, bn=False, bias=False."""

    def __init__(self, in_planes, out_planes, kernel_size, dilation=1, groups=1, relu=True, bn=False, bias=False):
        super().__init__()
        padding = _pad_for(kernel_size, dilation)
        self.out_channels = out_planes
        self.kernel_size, self.dilation, self.groups = kernel_size, dilation, groups
        self.conv = Conv2dParams(in_planes, out_planes, kernel_size=kernel_size, stride=1, padding=padding,
                                 dilation=dilation, groups=groups, bias=bias)
        self.bn = BatchNormParams(out_planes, eps=1e-5, momentum=0.01, affine=True) if bn else None
        self.relu = PReLUParams() if relu else None
        if bias:
            raise NotImplementedError("BasicConv(bias=True) is never used by the reference path")

    def wpk(self, nsrc, cin):
        return self._packs.get("w", [self.conv.weight], lambda: ops.pack_conv_weight(self.conv.weight, nsrc, cin, self.kernel_size))

    def forward_nhwc(self, x, res=(), tape=None):
        if tape is not None:
            raise NotImplementedError("BasicConv used stand-alone has no backward; it is differentiated inside its parent op")
        C = x.shape[-1]
        if self.groups == C and self.groups == self.conv.out_channels and self.groups > 1:
            y = ops.dwconv(x, self.conv.weight, self.kernel_size, self.dilation, in_relu=False)
            if self.bn is not None or self.relu is not None:
                raise NotImplementedError("depthwise BasicConv with bn/relu")
            return self._add_res(y, res)
        if self.groups != 1:
            raise NotImplementedError("grouped BasicConv other than depthwise")
        if C not in (32, 64, 96) or self.conv.out_channels != 32:
            raise NotImplementedError("BasicConv %d->%d: the HIP conv is built for 32-channel NAS cells" % (C, self.conv.out_channels))
        nsrc = C // 32
        srcs = [x] if nsrc == 1 else [x[..., 32 * i:32 * (i + 1)].contiguous() for i in range(nsrc)]
        scale = shift = None
        if self.bn is not None:
            scale, shift = _bn_scale_shift(self.bn, self._packs)
        act, slope = (ops.ACT_PRELU, self.relu.weight) if self.relu is not None else (ops.ACT_NONE, None)
        return ops.conv2d(srcs, self.wpk(nsrc, 32), self.kernel_size, self.dilation, scale=scale, shift=shift, act=act,
                          prelu=slope, res=res)


def conv3x3(in_planes, out_planes, stride=1):
    """operations_m.py:283-284."""
    return Conv2dParams(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class ResidualDenseBlock(_HipOp):
    """operations_m.py:435-449: x1=P(c1 x); x2=P(c2 [x,x1]); x3=P(c3 [x,x1,x2]); x3*0.333333 + x.
    The concatenations are virtual (the conv kernel walks up to three NHWC sources)."""

    def __init__(self, in_channels, kernel_size, dialtions=1, bias=False):
        super().__init__()
        self.k, self.d = kernel_size, dialtions
        self.conv1 = BasicConv(in_channels, in_channels, kernel_size, dilation=dialtions, relu=False)
        self.conv2 = BasicConv(in_channels * 2, in_channels, kernel_size, dilation=dialtions, relu=False)
        self.conv3 = BasicConv(in_channels * 3, in_channels, kernel_size, dilation=dialtions, relu=False)
        self.lrelu = PReLUParams()

    takes_cpool = True      # forward_nhwc(..., cpool=(comp, off)): ChannelPool of the block's output from its last conv's epilogue (inference)

    def forward_nhwc(self, x, res=(), tape=None, cpool=None):
        a = self.lrelu.weight
        k, d = self.k, self.d
        kw = dict(act=ops.ACT_PRELU, prelu=a)
        if tape is not None and ops.taping_wgrad():     # parameter gradients need every pre-activation and the block input
            x1, z1 = ops.conv2d([x], self.conv1.wpk(1, 32), k, d, want_aux=True, **kw)
            x2, z2 = ops.conv2d([x, x1], self.conv2.wpk(2, 32), k, d, want_aux=True, **kw)
            out, z3 = ops.conv2d([x, x1, x2], self.conv3.wpk(3, 32), k, d, alpha=0.333333, res=(x,) + tuple(res), want_aux=True, **kw)
            tape.append(dict(x=x, x1=x1, x2=x2, z1=z1, z2=z2, z3=z3))
            return out
        if (tape is None and x.dtype in ops.H16 and k == 3 and d == 1 and len(res) <= 2 and ops.CONFIG.get("rdb_fused", False)
                and x.shape[0] * ((x.shape[1] + 7) // 8) * ((x.shape[2] + 27) // 28) >= 512):
            # 16-bit inference forward: the whole block as one kernel -- x read once, out written once, x1 / x2 stay in LDS
            ws = [self.conv1.conv.weight, self.conv2.conv.weight, self.conv3.conv.weight]
            wpk = self._packs.get("rdb_" + str(x.dtype), ws, lambda: ops.rdb_fused_pack(ws[0], ws[1], ws[2], x.dtype))
            return ops.rdb_fused(x, wpk, a, alpha=0.333333, res=tuple(res), cpool=cpool)
        x1 = ops.conv2d([x], self.conv1.wpk(1, 32), k, d, **kw)
        x2 = ops.conv2d([x, x1], self.conv2.wpk(2, 32), k, d, **kw)
        if tape is None:
            return ops.conv2d([x, x1, x2], self.conv3.wpk(3, 32), k, d, alpha=0.333333, res=(x,) + tuple(res), cpool=cpool, **kw)
        out, z3 = ops.conv2d([x, x1, x2], self.conv3.wpk(3, 32), k, d, alpha=0.333333, res=(x,) + tuple(res), want_aux=True, **kw)
        tape.append(dict(x1=x1, x2=x2, z3=z3))
        return out

    bwd_takes_res = True    # backward_nhwc(..., res=(m,)): m is added to the returned gradient in the last dgrad conv's epilogue

    def backward_nhwc(self, g, t, wgrad=False, res=()):
        a = self.lrelu.weight
        k, d = self.k, self.d
        w1, w2, w3 = self.conv1.conv.weight, self.conv2.conv.weight, self.conv3.conv.weight
        # sign source of PReLU': the saved pre-activation when there is one, else the output (needs a non-negative slope)
        s1, s2 = t.get("z1"), t.get("z2")
        if s1 is None:
            check_positive_slope(a, "ResidualDenseBlock")
            s1, s2 = t["x1"], t["x2"]
        # g * PReLU'(z3) feeds three dgrad convs and d_x2 * PReLU'(z2) two: formed ONCE each by the elementwise kernel (which is also
        # the slope-gradient pass of the training step), so the five convs are plain single-source convs on the fast forward kernels
        # instead of five gradient-hook launches that each re-read the pre-activation (6 x 328 us -> ~1.6 ms per block at the bench shape)
        ds = ops.grad_of(a) if wgrad else None
        t3 = ops.prelu_bwd(g, t["z3"], a, ds, want_dx=True, factor=0.333333)
        d_x = ops.conv2d([t3], self._dgrad_w("c3s0", w3, 0), k, d, alpha=0.333333, res=(g,))
        d_x1 = ops.conv2d([t3], self._dgrad_w("c3s1", w3, 32), k, d, alpha=0.333333)
        d_x2 = ops.conv2d([t3], self._dgrad_w("c3s2", w3, 64), k, d, alpha=0.333333)
        t2 = ops.prelu_bwd(d_x2, s2, a, ds, want_dx=True)       # wgrad: s2 is the pre-activation z2
        d_x = ops.conv2d([t2], self._dgrad_w("c2s0", w2, 0), k, d, res=(d_x,))
        d_x1 = ops.conv2d([t2], self._dgrad_w("c2s1", w2, 32), k, d, res=(d_x1,))
        if wgrad:
            x, x1, x2 = t["x"], t["x1"], t["x2"]
            for w, srcs, dout, z, alpha in ((w3, [x, x1, x2], g, t["z3"], 0.333333), (w2, [x, x1], d_x2, t["z2"], 1.0),
                                            (w1, [x], d_x1, t["z1"], 1.0)):
                gw = ops.grad_of(w)
                if gw is not None:
                    ops.conv2d_wgrad(srcs, dout, k, d, z=z, act=ops.ACT_PRELU, prelu=a, alpha=alpha, out=gw)
            if ds is not None:
                ops.prelu_bwd(d_x1, t["z1"], a, ds, factor=1.0)
        return ops.conv2d([d_x1], self._dgrad_w("c1", w1, 0), k, d, res=(d_x,) + tuple(res), in_act=ops.IN_DPRELU, in_aux=s1, in_prelu=a)


class ResidualModule(_HipOp):
    """operations_m.py:451-464: conv kxk -> conv3x3 dil2 -> conv1x1 -> BN -> PReLU; + x."""

    def __init__(self, in_channels, kernel_size, dialtions=1, bias=False):
        super().__init__()
        self.k, self.d = kernel_size, dialtions
        self.op = nn.Sequential(
            BasicConv(in_channels, in_channels, kernel_size, dilation=dialtions, relu=False),
            Conv2dParams(in_channels, in_channels, kernel_size=3, stride=1, padding=2, dilation=2, bias=False),
            Conv2dParams(in_channels, in_channels, kernel_size=1, padding=0, bias=False),
            BatchNormParams(in_channels),
            PReLUParams(),
        )

    takes_out_f32 = True     # forward_nhwc(..., out_f32=True): the composed conv can write an fp32 map from fp16 sources

    def forward_nhwc(self, x, res=(), tape=None, out_f32=False):
        op = self.op
        t1 = ops.conv2d([x], op[0].wpk(1, 32), self.k, self.d)
        if tape is None and not op[3].training and ops.CONFIG.get("resmod_compose", True):
            # inference forward (round 4): conv3x3 dil 2 and the conv1x1 behind it have nothing in between -- ONE dilated 3x3 conv with
            # W[co][ci][tap] = sum_m w1x1[co][m] w3x3[m][ci][tap] and the BN / PReLU / residual epilogue; the map between them never
            # goes to HBM (fp32 maps: 206 us per forward at B=8 480x640, bf16 maps: 131 us)
            wc = self._packs.get("w23", [op[1].weight, op[2].weight],
                                 lambda: ops.pack_conv_weight(ops.compose_pw_conv_weight(op[2].weight, op[1].weight), 1, 32, 3))
            scale, shift = _bn_scale_shift(op[3], self._packs)
            return ops.conv2d([t1], wc, 3, 2, scale=scale, shift=shift, act=ops.ACT_PRELU, prelu=op[4].weight, res=(x,) + tuple(res),
                              out_f32=out_f32)
        w2 = self._packs.get("w2", [op[1].weight], lambda: ops.pack_conv_weight(op[1].weight, 1, 32, 3))
        t2 = ops.conv2d([t1], w2, 3, 2)
        w3 = self._packs.get("w3", [op[2].weight], lambda: ops.pack_conv_weight(op[2].weight, 1, 32, 1))
        if op[3].training or (tape is not None and ops.taping_wgrad()):
            # split form: raw 1x1 -> BatchNorm statistics (batch or running) -> affine + PReLU + residuals
            c = ops.conv2d([t2], w3, 1, 1)
            out, stats = _bn_split(op[3], c, ops.ACT_PRELU, op[4].weight, (x,) + tuple(res))
            if tape is not None:
                tape.append(dict(x=x, t1=t1, t2=t2, c=c, stats=stats, bn_training=op[3].training))
            return out
        scale, shift = _bn_scale_shift(op[3], self._packs)
        kw = dict(scale=scale, shift=shift, act=ops.ACT_PRELU, prelu=op[4].weight, res=(x,) + tuple(res))
        if tape is None:
            return ops.conv2d([t2], w3, 1, 1, **kw)
        out, z = ops.conv2d([t2], w3, 1, 1, want_aux=True, **kw)
        tape.append(dict(z=z))
        return out

    def backward_nhwc(self, g, t, wgrad=False):
        op = self.op
        if "c" in t:   # split form (train-mode BatchNorm and / or parameter gradients)
            bn = op[3]
            d_c = ops.bn_act_bwd(g, t["c"], t["stats"], ops.ACT_PRELU, op[4].weight,
                                 ops.grad_of(bn.weight) if wgrad else None, ops.grad_of(bn.bias) if wgrad else None,
                                 ops.grad_of(op[4].weight) if wgrad else None, training=t["bn_training"])
            d_t2 = ops.conv2d([d_c], self._dgrad_w("c1x1", op[2].weight), 1, 1)
            d_t1 = ops.conv2d([d_t2], self._dgrad_w("c3d2", op[1].weight), 3, 2)
            if wgrad:
                for w, src, dout, kk, dd in ((op[2].weight, t["t2"], d_c, 1, 1), (op[1].weight, t["t1"], d_t2, 3, 2),
                                             (op[0].conv.weight, t["x"], d_t1, self.k, self.d)):
                    gw = ops.grad_of(w)
                    if gw is not None:
                        ops.conv2d_wgrad([src], dout, kk, dd, out=gw)
            return ops.conv2d([d_t1], self._dgrad_w("ck", op[0].conv.weight), self.k, self.d, res=(g,))
        scale, _ = _bn_scale_shift(op[3], self._packs)
        d_t2 = ops.conv2d([g], self._dgrad_w("c1x1", op[2].weight), 1, 1, in_act=ops.IN_DPRELU, in_aux=t["z"], in_scale=scale,
                          in_prelu=op[4].weight)
        d_t1 = ops.conv2d([d_t2], self._dgrad_w("c3d2", op[1].weight), 3, 2)
        return ops.conv2d([d_t1], self._dgrad_w("ck", op[0].conv.weight), self.k, self.d, res=(g,))


class DilConv(_HipOp):
    """operations_m.py:494-506: ReLU -> depthwise kxk dil d -> conv1x1 -> BN(affine); + x."""

    def __init__(self, C_in, C_out, kernel_size, dilation, affine=True):
        super().__init__()
        self.k, self.d = kernel_size, dilation
        self.op = nn.Sequential(
            nn.ReLU(inplace=False),
            BasicConv(C_in, C_out, kernel_size, dilation=dilation, relu=False, groups=C_in),
            Conv2dParams(C_in, C_out, kernel_size=1, padding=0, bias=False),
            BatchNormParams(C_out, affine=affine),
        )

    takes_cpool = True

    def forward_nhwc(self, x, res=(), tape=None, cpool=None):
        op = self.op
        dense = ops.CONFIG.get("dilconv_dense", True) if x.dtype in ops.H16 else ops.CONFIG.get("dilconv_dense_f32", True)
        if self.k == 3 and tape is None and not op[3].training and dense:
            # inference forward (bf16 maps, and since round 4 fp32 maps too): depthwise + 1x1 as ONE dense k x k conv with
            # W[co][ci][tap] = pw[co][ci] * dw[ci][tap] (one fp32 rounding of the product, then the usual split) and the ReLU as its input
            # activation -- the depthwise map (a 32-channel map written and read back) never goes to HBM
            wc = self._packs.get("wc", [op[1].conv.weight, op[2].weight],
                                 lambda: ops.pack_conv_weight(ops.compose_dw_pw_weight(op[1].conv.weight, op[2].weight), 1, 32, self.k))
            scale, shift = _bn_scale_shift(op[3], self._packs)
            return ops.conv2d([x], wc, self.k, self.d, in_act=ops.ACT_RELU, scale=scale, shift=shift, res=(x,) + tuple(res), cpool=cpool)
        t = ops.dwconv(x, op[1].conv.weight, self.k, self.d, in_relu=True)
        w = self._packs.get("w", [op[2].weight], lambda: ops.pack_conv_weight(op[2].weight, 1, 32, 1))
        if op[3].training or (tape is not None and ops.taping_wgrad()):
            c = ops.conv2d([t], w, 1, 1)
            out, stats = _bn_split(op[3], c, ops.ACT_NONE, None, (x,) + tuple(res))
            if tape is not None:
                tape.append(dict(x=x, t=t, c=c, stats=stats, bn_training=op[3].training))
            if cpool is not None:            # (train-mode forward without a tape: the split BatchNorm form has no pooling epilogue)
                ops.channel_pool1(out, *cpool)
            return out
        scale, shift = _bn_scale_shift(op[3], self._packs)
        if tape is not None:
            tape.append(dict(x=x))
        return ops.conv2d([t], w, 1, 1, scale=scale, shift=shift, res=(x,) + tuple(res), cpool=cpool)

    def backward_nhwc(self, g, t, wgrad=False):
        op = self.op
        if "c" in t:
            bn = op[3]
            d_c = ops.bn_act_bwd(g, t["c"], t["stats"], ops.ACT_NONE, None,
                                 ops.grad_of(bn.weight) if wgrad and bn.affine else None,
                                 ops.grad_of(bn.bias) if wgrad and bn.affine else None, None, training=t["bn_training"])
            d_t = ops.conv2d([d_c], self._dgrad_w("c1x1", op[2].weight), 1, 1)
            if wgrad:
                gw = ops.grad_of(op[2].weight)
                if gw is not None:
                    ops.conv2d_wgrad([t["t"]], d_c, 1, 1, out=gw)
                gd = ops.grad_of(op[1].conv.weight)
                if gd is not None:
                    ops.dwconv_wgrad(t["x"], d_t, self.k, self.d, True, gd)
            return ops.dwconv_bwd(d_t, op[1].conv.weight, self.k, self.d, aux=t["x"], add=g)
        scale, _ = _bn_scale_shift(op[3], self._packs)
        d_t = ops.conv2d([g], self._dgrad_w("c1x1", op[2].weight), 1, 1, in_act=ops.IN_SCALE, in_scale=scale)
        return ops.dwconv_bwd(d_t, op[1].conv.weight, self.k, self.d, aux=t["x"], add=g)


def _bn_split(bn, c, act, slope, res):
    """act(BatchNorm(c)) + residuals in split form: statistics kernel (train mode: batch statistics + running-statistics
    update, as nn.BatchNorm2d does; eval mode: the running statistics) then one affine/activation/residual pass.
    Returns (out, (mean, invstd, scale, shift))."""
    g, b = (bn.weight, bn.bias) if bn.affine else (None, None)
    if bn.training:
        if bn.momentum is None or not bn.track_running_stats:
            raise NotImplementedError("BatchNorm with momentum=None / track_running_stats=False is not used by the reference")
        stats = ops.bn_stats(c, g, b, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        bn.num_batches_tracked.add_(1)
    else:
        stats = ops.bn_eval_stats(g, b, bn.running_mean, bn.running_var, bn.eps)
    return ops.affine_act_res(c, stats[2], stats[3], act, slope, res), stats


def _bn_scale_shift(bn, cache=None, name="bn"):
    """Eval-mode BatchNorm folded to per-channel scale/shift (cached on parameter/buffer versions)."""
    if bn.training:
        raise RuntimeError("_bn_scale_shift folds RUNNING statistics: train-mode BatchNorm goes through _bn_split (batch statistics)")

    def build():
        if bn.affine:
            return ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
        one = torch.ones_like(bn.running_mean)
        return ops.bn_fold(one, torch.zeros_like(one), bn.running_mean, bn.running_var, bn.eps)

    if cache is None:
        return build()
    keys = [bn.running_mean, bn.running_var] + ([bn.weight, bn.bias] if bn.affine else [])
    return cache.get(name, keys, build)


class SepConv(_HipOp):
    """operations_m.py:509-525 (no residual)."""

    def __init__(self, C_in, C_out, kernel_size, stride, padding, affine=True):
        super().__init__()
        if stride != 1 or padding != kernel_size // 2:
            raise NotImplementedError("SepConv is only reachable with stride 1, padding k//2 (OPS registry)")
        self.k = kernel_size
        self.op = nn.Sequential(
            nn.ReLU(inplace=False),
            Conv2dParams(C_in, C_in, kernel_size=kernel_size, stride=stride, padding=padding, groups=C_in, bias=False),
            Conv2dParams(C_in, C_in, kernel_size=1, padding=0, bias=False),
            BatchNormParams(C_in, affine=affine),
            nn.ReLU(inplace=False),
            Conv2dParams(C_in, C_in, kernel_size=kernel_size, stride=1, padding=padding, groups=C_in, bias=False),
            Conv2dParams(C_in, C_out, kernel_size=1, padding=0, bias=False),
            BatchNormParams(C_out, affine=affine),
        )

    def backward_nhwc(self, g, t, wgrad=False):
        op = self.op
        if "c2" in t:   # split form (train-mode BatchNorm and / or parameter gradients)
            go = lambda p_: ops.grad_of(p_) if wgrad else None
            aff = lambda bn: (go(bn.weight), go(bn.bias)) if bn.affine else (None, None)
            d_c2 = ops.bn_act_bwd(g, t["c2"], t["stats2"], ops.ACT_NONE, None, *aff(op[7]), None, training=t["bn_training"])
            d_t2 = ops.conv2d([d_c2], self._dgrad_w("c6", op[6].weight), 1, 1)
            d_a = ops.dwconv_bwd(d_t2, op[5].weight, self.k, 1)                         # a = ReLU(BN1(c1)): the gate is in bn_act_bwd
            d_c1 = ops.bn_act_bwd(d_a, t["c1"], t["stats1"], ops.ACT_RELU, None, *aff(op[3]), None, training=t["bn_training"])
            d_t = ops.conv2d([d_c1], self._dgrad_w("c2", op[2].weight), 1, 1)
            if wgrad:
                for w, src, dout in ((op[6].weight, t["t2"], d_c2), (op[2].weight, t["t"], d_c1)):
                    gw = ops.grad_of(w)
                    if gw is not None:
                        ops.conv2d_wgrad([src], dout, 1, 1, out=gw)
                for w, src, dout, relu in ((op[5].weight, t["a"], d_t2, False), (op[1].weight, t["x"], d_t, True)):
                    gw = ops.grad_of(w)
                    if gw is not None:
                        ops.dwconv_wgrad(src, dout, self.k, 1, relu, gw)
            return ops.dwconv_bwd(d_t, op[1].weight, self.k, 1, aux=t["x"])
        s1, _ = _bn_scale_shift(op[3], self._packs, 'bn1')
        s2, _ = _bn_scale_shift(op[7], self._packs, 'bn2')
        d_t5 = ops.conv2d([g], self._dgrad_w("c6", op[6].weight), 1, 1, in_act=ops.IN_SCALE, in_scale=s2)
        d_z3 = ops.dwconv_bwd(d_t5, op[5].weight, self.k, 1, aux=t["z3"])          # through dw5 and the ReLU after BN3
        d_t1 = ops.conv2d([d_z3], self._dgrad_w("c2", op[2].weight), 1, 1, in_act=ops.IN_SCALE, in_scale=s1)
        return ops.dwconv_bwd(d_t1, op[1].weight, self.k, 1, aux=t["x"])            # through dw1 and the leading ReLU

    def forward_nhwc(self, x, res=(), tape=None):
        op = self.op
        t = ops.dwconv(x, op[1].weight, self.k, 1, in_relu=True)
        w1 = self._packs.get("w1", [op[2].weight], lambda: ops.pack_conv_weight(op[2].weight, 1, 32, 1))
        w2 = self._packs.get("w2", [op[6].weight], lambda: ops.pack_conv_weight(op[6].weight, 1, 32, 1))
        if op[3].training or (tape is not None and ops.taping_wgrad()):
            c1 = ops.conv2d([t], w1, 1, 1)
            a, stats1 = _bn_split(op[3], c1, ops.ACT_RELU, None, ())
            t2 = ops.dwconv(a, op[5].weight, self.k, 1, in_relu=False)
            c2 = ops.conv2d([t2], w2, 1, 1)
            out, stats2 = _bn_split(op[7], c2, ops.ACT_NONE, None, tuple(res))
            if tape is not None:
                tape.append(dict(x=x, t=t, c1=c1, a=a, t2=t2, c2=c2, stats1=stats1, stats2=stats2, bn_training=op[3].training))
            return out
        s1, b1 = _bn_scale_shift(op[3], self._packs, 'bn1')
        if tape is not None:
            t, z3 = ops.conv2d([t], w1, 1, 1, scale=s1, shift=b1, act=ops.ACT_RELU, want_aux=True)
            tape.append(dict(x=x, z3=z3))
        else:
            t = ops.conv2d([t], w1, 1, 1, scale=s1, shift=b1, act=ops.ACT_RELU)
        t = ops.dwconv(t, op[5].weight, self.k, 1, in_relu=False)
        s2, b2 = _bn_scale_shift(op[7], self._packs, 'bn2')
        return ops.conv2d([t], w2, 1, 1, scale=s2, shift=b2, res=res)


class eca_layer(nn.Module):
    """operations_m.py:340-367 (parameter holder for the Conv1d over channels; the arithmetic is fused
    into ECABasicBlock's kernels)."""

    def __init__(self, channel, c_out, stride, k_size=3):
        super().__init__()
        self.k_size = k_size
        self.conv = Conv1dParams(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)

    def forward(self, x):
        """Stand-alone call (operations_m.py:353-367); inside ECABasicBlock the same arithmetic runs fused (conv2's epilogue pools,
        eca_finish applies the gate).  Inference only, like every stand-alone helper of the import surface."""
        ops.require_no_grad(x)
        if x.shape[1] != 32:
            raise NotImplementedError("eca_layer: the HIP kernels are built for the 32-channel maps of the search space")
        with torch.no_grad():
            return ops.to_nchw_view(ops.eca_layer_fwd(ops.to_nhwc(x), self.conv.weight, self.k_size))


class ECABasicBlock(_HipOp):
    """operations_m.py:368-393: r=conv3x3(x); o=BasicConv_k(PReLU(r)); o=o*sigmoid(conv1d(avgpool o)); PReLU(o+r).
    The residual is conv1's output, not the block input."""

    def __init__(self, inplanes, planes, kernel=3, dilation=1, stride=1, reduction=64, with_norm=False):
        super().__init__()
        if with_norm:
            raise NotImplementedError("ECABasicBlock(with_norm=True) is never constructed by the reference")
        self.with_norm = with_norm
        self.k = kernel
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.conv2 = BasicConv(inplanes, inplanes, kernel, relu=False)
        self.se = eca_layer(planes, planes, stride, k_size=kernel)
        self.relu = PReLUParams()

    def forward_nhwc(self, x, res=(), tape=None):
        a = self.relu.weight
        w1 = self._packs.get("w1", [self.conv1.weight], lambda: ops.pack_conv_weight(self.conv1.weight, 1, 32, 3))
        r = ops.conv2d([x], w1, 3, 1)
        o, partial = ops.conv2d([r], self.conv2.wpk(1, 32), self.k, 1, in_act=ops.ACT_PRELU, in_prelu=a, pool=True)
        if tape is None:
            out = ops.eca_finish(o, r, partial, self.se.conv.weight, self.k, a)
        else:
            out, u, gate = ops.eca_finish(o, r, partial, self.se.conv.weight, self.k, a, save=True)
            tape.append(dict(r=r, o=o, u=u, gate=gate, x=x, pool=partial) if ops.taping_wgrad() else dict(r=r, o=o, u=u, gate=gate))
        return self._add_res(out, res)

    bwd_takes_res = True    # see ResidualDenseBlock

    def backward_nhwc(self, g, t, wgrad=False, res=()):
        a = self.relu.weight
        if not wgrad:
            d_o, d_r = ops.eca_bwd(g, t["u"], t["o"], t["gate"], self.se.conv.weight, self.k, a)
            # o = conv2(PReLU(r)):  d_r += conv2^T(d_o) * PReLU'(r)
            d_r = ops.conv2d([d_o], self._dgrad_w("c2", self.conv2.conv.weight), self.k, 1, epi_dact=1, epi_aux=t["r"], prelu=a, res=(d_r,))
            return ops.conv2d([d_r], self._dgrad_w("c1", self.conv1.weight), 3, 1, res=tuple(res))
        B, H, W, _ = g.shape
        ds = ops.grad_of(a)
        if ds is not None:
            ops.prelu_bwd(g, t["u"], a, ds)                                   # out = PReLU(u)
        d_o, d_r, dgate = ops.eca_bwd(g, t["u"], t["o"], t["gate"], self.se.conv.weight, self.k, a, want_partial=True)
        g1d = ops.grad_of(self.se.conv.weight)
        if g1d is not None:
            ops.eca_wgrad(t["pool"], dgate, t["gate"], self.k, g1d, B, H, W)
        gw2 = ops.grad_of(self.conv2.conv.weight)
        if gw2 is not None:                                                   # conv2's input is PReLU(r): materialised for the wgrad
            ops.conv2d_wgrad([ops.affine_act_res(t["r"], None, None, ops.ACT_PRELU, a)], d_o, self.k, 1, out=gw2)
        tt = ops.conv2d([d_o], self._dgrad_w("c2", self.conv2.conv.weight), self.k, 1)     # d/d PReLU(r)
        d_r = ops.prelu_bwd(tt, t["r"], a, ds, add=d_r, want_dx=True)
        gw1 = ops.grad_of(self.conv1.weight)
        if gw1 is not None:
            ops.conv2d_wgrad([t["x"]], d_r, 3, 1, out=gw1)
        return ops.conv2d([d_r], self._dgrad_w("c1", self.conv1.weight), 3, 1, res=tuple(res))


class Spatial_BasicBlock(_HipOp):
    """operations_m.py:179-204 (SPAattention): in the search space, not in the shipped genotype."""

    def __init__(self, inplanes, planes, kernel=3, dilation=1, stride=1, reduction=64, with_norm=False):
        super().__init__()
        self.k = kernel
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.conv2 = BasicConv(inplanes, inplanes, kernel, relu=False)
        self.se = spatial_attn_layer(kernel)
        self.relu = PReLUParams()

    def forward_nhwc(self, x, res=(), tape=None):
        a = self.relu.weight
        w1 = self._packs.get("w1", [self.conv1.weight], lambda: ops.pack_conv_weight(self.conv1.weight, 1, 32, 3))
        r = ops.conv2d([x], w1, 3, 1)
        o = ops.conv2d([r], self.conv2.wpk(1, 32), self.k, 1, in_act=ops.ACT_PRELU, in_prelu=a)
        wsp = self.se.spatial.conv.weight
        if tape is None:
            out = ops.spa1(o, r, wsp, self.k, a)
        elif ops.taping_wgrad():
            out, u, s, comp = ops.spa1(o, r, wsp, self.k, a, save=True, want_comp=True)
            tape.append(dict(r=r, o=o, u=u, s=s, x=x, comp=comp))
        else:
            out, u, s = ops.spa1(o, r, wsp, self.k, a, save=True)
            tape.append(dict(r=r, o=o, u=u, s=s))
        return self._add_res(out, res)

    def backward_nhwc(self, g, t, wgrad=False):
        a = self.relu.weight
        wsp = self.se.spatial.conv.weight
        if not wgrad:
            d_o, d_r = ops.spa1_bwd(g, t["u"], t["o"], t["s"], wsp, self.k, a)
            d_r = ops.conv2d([d_o], self._dgrad_w("c2", self.conv2.conv.weight), self.k, 1, epi_dact=1, epi_aux=t["r"], prelu=a, res=(d_r,))
            return ops.conv2d([d_r], self._dgrad_w("c1", self.conv1.weight), 3, 1)
        ds = ops.grad_of(a)
        if ds is not None:
            ops.prelu_bwd(g, t["u"], a, ds)                                   # out = PReLU(u)
        d_o, d_r, dpre = ops.spa1_bwd(g, t["u"], t["o"], t["s"], wsp, self.k, a, want_dpre=True)
        gsp = ops.grad_of(wsp)
        if gsp is not None:   # 2 -> 1 conv: the pooled map zero padded to 4 channels, first 2*k*k values of the 4-channel gradient
            tmp = torch.zeros(4 * self.k * self.k, device=g.device, dtype=torch.float32)
            ops.corr1_wgrad(dpre, ops.pad_channels(t["comp"], 4), self.k, tmp)
            ops.axpy_(gsp.view(-1), tmp[:2 * self.k * self.k])
        gw2 = ops.grad_of(self.conv2.conv.weight)
        if gw2 is not None:
            ops.conv2d_wgrad([ops.affine_act_res(t["r"], None, None, ops.ACT_PRELU, a)], d_o, self.k, 1, out=gw2)
        tt = ops.conv2d([d_o], self._dgrad_w("c2", self.conv2.conv.weight), self.k, 1)
        d_r = ops.prelu_bwd(tt, t["r"], a, ds, add=d_r, want_dx=True)
        gw1 = ops.grad_of(self.conv1.weight)
        if gw1 is not None:
            ops.conv2d_wgrad([t["x"]], d_r, 3, 1, out=gw1)
        return ops.conv2d([d_r], self._dgrad_w("c1", self.conv1.weight), 3, 1)


class ChannelPool(nn.Module):
    """operations_m.py:148-150 (1-argument form, only inside spatial_attn_layer)."""

    def forward(self, x):
        """cat(max_c x, mean_c x) -> [B,2,H,W]; inside Spatial_BasicBlock it runs fused in the spa1 kernel, whose pooled map this
        stand-alone call returns."""
        ops.require_no_grad(x)
        if x.shape[1] != 32:
            raise NotImplementedError("ChannelPool: the HIP kernels are built for the 32-channel maps of the search space")
        with torch.no_grad():
            xh = ops.to_nhwc(x)
            one = torch.ones(1, device=x.device, dtype=torch.float32)
            comp = ops.spa1(xh, torch.zeros_like(xh), torch.zeros(1, 2, 3, 3, device=x.device), 3, one, save=True, want_comp=True)[3]
            return comp.permute(0, 3, 1, 2).contiguous()


class spatial_attn_layer(nn.Module):
    """operations_m.py:153-164 (parameter holder)."""

    def __init__(self, kernel_size=5):
        super().__init__()
        self.compress = ChannelPool()
        self.spatial = BasicConv(2, 1, kernel_size, relu=False)

    def forward(self, x):
        """x * sigmoid(conv_k(ChannelPool(x)))  (operations_m.py:159-164): the fused block's spa1 kernel with a zero residual and
        PReLU slope 1 (= identity)."""
        ops.require_no_grad(x)
        if x.shape[1] != 32:
            raise NotImplementedError("spatial_attn_layer: the HIP kernels are built for the 32-channel maps of the search space")
        with torch.no_grad():
            xh = ops.to_nhwc(x)
            one = torch.ones(1, device=x.device, dtype=torch.float32)
            k = self.spatial.conv.weight.shape[-1]
            return ops.to_nchw_view(ops.spa1(xh, torch.zeros_like(xh), self.spatial.conv.weight, k, one))


def _selfpath(*a, **k):
    raise NotImplementedError("SelAttention/SelfPath is unreachable from MixedOp in the reference (SURVEY.md 8(a) F4)")


# operations_m.py:9-18
OPS = {
    'Denseblocks': lambda C, kernel, dialtion, affine: ResidualDenseBlock(C, kernel, dialtion),
    'Residualblocks': lambda C, kernel, dialtion, affine: ResidualModule(C, kernel, dialtion),
    'ECAattention': lambda C, kernel, dialtion, affine: ECABasicBlock(C, C, kernel, dialtion),
    'SPAattention': lambda C, kernel, dialtion, affine: Spatial_BasicBlock(C, C, kernel, dialtion),
    'DilConv': lambda C, kernel, dialtion, affine: DilConv(C, C, kernel, dialtion),
    'SepConv': lambda C, kernel, dialtion, affine: SepConv(C, C, kernel, 1, kernel // 2),
    'SelAttention': lambda C, kernel, dialtion, affine: _selfpath(),
}
