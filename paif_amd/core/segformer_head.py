"""SegFormer all-MLP decode head -- MI355X-native counterpart of the reference's core/segformer_head.py.
Same classes and state_dict keys (`linear_fuse.{conv.weight, bn.*}` follow mmcv.cnn.ConvModule's naming;
semantics per oracle/shims/mmcv: conv without bias -> BatchNorm -> ReLU)."""
import torch
import torch.nn as nn

from .. import ops
from .mix_transformer import _linear_wgrad
from ..operations_m import BatchNormParams, Conv2dParams, LinearParams, _PackCache, _bn_scale_shift, _bn_split


class MLP(nn.Module):
    """core/segformer_head.py:13-24."""

    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = LinearParams(input_dim, embed_dim)


class ConvModule(nn.Module):
    """Parameter holder with mmcv.cnn.ConvModule's sub-module names (conv, bn, activate)."""

    def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None):
        super().__init__()
        self.conv = Conv2dParams(in_channels, out_channels, kernel_size, bias=(norm_cfg is None))
        self.with_norm = norm_cfg is not None
        if self.with_norm:
            self.bn = BatchNormParams(out_channels)
        self.activate = nn.ReLU(inplace=True)


class SegFormerHead(nn.Module):
    """core/segformer_head.py:27-82."""

    def __init__(self, feature_strides=None, in_channels=128, embedding_dim=256, num_classes=20, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        assert len(feature_strides) == len(self.in_channels)
        assert min(feature_strides) == feature_strides[0]
        self.feature_strides = feature_strides
        c1_in, c2_in, c3_in, c4_in = self.in_channels
        self.embedding_dim = embedding_dim
        self.linear_c4 = MLP(input_dim=c4_in, embed_dim=embedding_dim)
        self.linear_c3 = MLP(input_dim=c3_in, embed_dim=embedding_dim)
        self.linear_c2 = MLP(input_dim=c2_in, embed_dim=embedding_dim)
        self.linear_c1 = MLP(input_dim=c1_in, embed_dim=embedding_dim)
        self.dropout = nn.Dropout2d(0.1)
        self.linear_fuse = ConvModule(in_channels=embedding_dim * 4, out_channels=embedding_dim, kernel_size=1,
                                      norm_cfg=dict(type='BN', requires_grad=True))
        self.linear_pred = Conv2dParams(embedding_dim, self.num_classes, kernel_size=1)
        self._packs = _PackCache()

    def forward_nhwc(self, feats, tape=None):
        """feats: 4 NHWC maps -> logits NHWC [B,H/4,W/4,num_classes].  Train mode: batch statistics in linear_fuse's BatchNorm
        (mmcv ConvModule: conv -> BN -> ReLU) and Dropout2d(0.1) on its output (core/segformer_head.py:47,79-80)."""
        c1, c2, c3, c4 = feats
        B, H1, W1, _ = c1.shape
        E = self.embedding_dim
        bn = self.linear_fuse.bn
        wg = tape is not None and ops.taping_wgrad()
        if not (bn.training or wg or self.training) and ops.CONFIG.get("head_fold", True):
            # inference / input-gradient mode: linear_fuse folded in front of the upsampling (1.2 instead of 10.1 GFLOP per pair)
            M, v = self._folded()
            zs = [ops.gemm(c, M[i], shift=v[i]) for i, c in zip((3, 2, 1, 0), (c1, c2, c3, c4))]     # z1 (full size), z2, z3, z4
            scale, shift = _bn_scale_shift(bn, self._packs)
            x = ops.head_sum(zs, scale, shift)
            if tape is not None:
                tape.update(x=x, shapes=[tuple(c.shape) for c in feats], pre=None, stats=None, drop=None, bn_training=False, folded=True)
            pw = self.linear_pred.weight.view(self.num_classes, E)
            return ops.gemm(x, pw, shift=self.linear_pred.bias)
        cat = torch.empty((B, H1, W1, 4 * E), device=c1.device, dtype=torch.float32)
        # concat order [_c4, _c3, _c2, _c1] (core/segformer_head.py:77)
        for i, (c, lin) in enumerate(((c4, self.linear_c4), (c3, self.linear_c3), (c2, self.linear_c2))):
            y = ops.gemm(c, lin.proj.weight, shift=lin.proj.bias)
            ops.resize_bilinear_into(y, cat, i * E)
        ops.gemm(c1, self.linear_c1.proj.weight, shift=self.linear_c1.proj.bias, out=cat, col_offset=3 * E)
        fw = self.linear_fuse.conv.weight.view(E, 4 * E)
        pre = stats = drop = None
        if bn.training or wg:     # split form: raw GEMM -> BatchNorm statistics (batch or running) -> affine + ReLU
            pre = ops.gemm(cat, fw)
            x, stats = _bn_split(bn, pre, ops.ACT_RELU, None, ())
        else:
            scale, shift = _bn_scale_shift(bn, self._packs)
            x = ops.gemm(cat, fw, scale=scale, shift=shift, act=ops.ACT_RELU)
        xd = x
        if self.training and self.dropout.p > 0:
            drop = ops.DROP_RNG.keep_mask(B * E, self.dropout.p, x.device).view(B, E)     # Dropout2d: whole channels per sample
            xd = ops.rowscale_add(x, drop, None, per_channel=True)
        if tape is not None:
            tape.update(x=x, shapes=[tuple(c.shape) for c in feats], pre=pre, stats=stats, drop=drop, bn_training=bn.training)
            if wg:
                tape.update(feats=list(feats), cat=cat, xd=xd)
        pw = self.linear_pred.weight.view(self.num_classes, E)
        return ops.gemm(xd, pw, shift=self.linear_pred.bias)

    def _folded(self):
        """M[i] = W_i L_i ([E, C_i], nn.Linear layout) and v[i] = W_i b_i for i = 0..3 in concat order (c4, c3, c2, c1): the
        stage's MLP followed by its 256-column block W_i of linear_fuse.conv.weight, as one Linear.  Cached per weight version."""
        E = self.embedding_dim
        lins = (self.linear_c4, self.linear_c3, self.linear_c2, self.linear_c1)

        def build():
            fw = self.linear_fuse.conv.weight.detach().view(E, 4 * E)
            M, v = [], []
            for i, lin in enumerate(lins):
                Wi = fw[:, i * E:(i + 1) * E].contiguous()                                   # weight preparation, once per version
                M.append(ops.gemm(Wi, ops.transpose_pad(lin.proj.weight)))                   # [E, C_i] = W_i @ L_i
                v.append(ops.gemm(lin.proj.bias.detach().view(1, E), Wi).view(E))            # W_i @ b_i
            return M, v

        keys = [self.linear_fuse.conv.weight] + [p for lin in lins for p in (lin.proj.weight, lin.proj.bias)]
        return self._packs.get("folded", keys, build)

    def _backward_folded(self, d_logits, tape):
        E = self.embedding_dim
        pw = self.linear_pred.weight.view(self.num_classes, E)
        pwt = self._packs.get("predT", [self.linear_pred.weight], lambda: ops.transpose_pad(pw))
        scale, _ = _bn_scale_shift(self.linear_fuse.bn, self._packs)
        M, _ = self._folded()
        d_pre = ops.relu_mask_scale(ops.gemm(d_logits, pwt), tape["x"], scale)               # through ReLU and the folded BN
        grads = [None] * 4
        for i, idx in ((0, 3), (1, 2), (2, 1), (3, 0)):                                      # concat slot i <-> stage idx
            _, h, w, _ = tape["shapes"][idx]
            d_z = d_pre if idx == 0 else ops.resize_bilinear_adjoint(d_pre, 0, E, h, w)
            mt = self._packs.get("foldT%d" % i, [self.linear_fuse.conv.weight, (self.linear_c4, self.linear_c3, self.linear_c2,
                                                                               self.linear_c1)[i].proj.weight],
                                 lambda i=i: ops.transpose_pad(M[i]))
            grads[idx] = ops.gemm(d_z, mt)
        return grads

    def backward_nhwc(self, d_logits, tape, wgrad=False):
        """d_logits NHWC [B,H/4,W/4,32] (channels >= num_classes zero) -> [d_c1, d_c2, d_c3, d_c4] NHWC; wgrad: the head's
        parameter gradients too."""
        if tape.get("folded"):
            assert not wgrad, "the folded head records no parameter-gradient tape"
            return self._backward_folded(d_logits, tape)
        E = self.embedding_dim
        pw = self.linear_pred.weight.view(self.num_classes, E)
        fw = self.linear_fuse.conv.weight.view(E, 4 * E)
        pwt = self._packs.get("predT", [self.linear_pred.weight], lambda: ops.transpose_pad(pw))     # [E, 32]
        fwt = self._packs.get("fuseT", [self.linear_fuse.conv.weight], lambda: ops.transpose_pad(fw))  # [4E, E]
        assert d_logits.shape[-1] == pwt.shape[1]
        if wgrad:
            gw, gb = ops.grad_of(self.linear_pred.weight), ops.grad_of(self.linear_pred.bias)
            if gw is not None:
                ops.gemm_wgrad(d_logits, tape["xd"], out_w=gw, out_b=gb, n=self.num_classes)
            elif gb is not None:
                ops.colsum(d_logits, gb, ncols=4 * ((self.num_classes + 3) // 4))
        d_x = ops.gemm(d_logits, pwt)                                           # [B,H1,W1,E]
        if tape["drop"] is not None:
            d_x = ops.rowscale_add(d_x, tape["drop"], None, per_channel=True)
        if tape["pre"] is not None:   # split form: BatchNorm (+ReLU) backward as its own kernels
            bn = self.linear_fuse.bn
            d_pre = ops.bn_act_bwd(d_x, tape["pre"], tape["stats"], ops.ACT_RELU, None, ops.grad_of(bn.weight) if wgrad else None,
                                   ops.grad_of(bn.bias) if wgrad else None, None, training=tape["bn_training"])
            if wgrad:
                gw = ops.grad_of(self.linear_fuse.conv.weight)
                if gw is not None:
                    ops.gemm_wgrad(d_pre, tape["cat"], want_bias=False, out_w=gw)
            d_cat = ops.gemm(d_pre, fwt)
        else:
            scale, _ = _bn_scale_shift(self.linear_fuse.bn, self._packs)
            d_cat = ops.gemm(d_x, fwt, a_mask=tape["x"], a_scale=scale)         # through ReLU and the folded BN
        grads = [None] * 4
        for i, (idx, lin) in enumerate(((3, self.linear_c4), (2, self.linear_c3), (1, self.linear_c2))):
            _, h, w, _ = tape["shapes"][idx]
            d_y = ops.resize_bilinear_adjoint(d_cat, i * E, E, h, w)
            if wgrad:
                _linear_wgrad(lin.proj, d_y, tape["feats"][idx])
            lt = self._packs.get("linT%d" % idx, [lin.proj.weight], lambda lin=lin: ops.transpose_pad(lin.proj.weight))
            grads[idx] = ops.gemm(d_y, lt)
        if wgrad:
            lin = self.linear_c1.proj
            gw, gb = ops.grad_of(lin.weight), ops.grad_of(lin.bias)
            if gw is not None:   # d_y of linear_c1 = columns [3E, 4E) of d_cat
                ops.gemm_wgrad(d_cat, tape["feats"][0], out_w=gw, out_b=gb, n=E, dy_col0=3 * E)
        l1t = self._packs.get("linT0", [self.linear_c1.proj.weight], lambda: ops.transpose_pad(self.linear_c1.proj.weight))
        grads[0] = ops.gemm(d_cat, l1t, a_cols=(3 * E, E))
        if wgrad:
            ops.grads_ready(self)
        return grads

    def forward(self, x):
        ops.require_no_grad(*x)
        with torch.no_grad():
            out = self.forward_nhwc([ops.to_nhwc(t) for t in x])
            return ops.nhwc_to_nchw(out)
