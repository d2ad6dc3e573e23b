"""MixTransformer (MiT-B0..B5) encoder -- MI355X-native counterpart of the reference's
core/mix_transformer.py (NVIDIA SegFormer code).  Same classes, constructor arguments, init and
state_dict keys; forward bodies launch the gfx950 kernels (fp32 MFMA GEMM, fused SR attention, LayerNorm,
depthwise-conv+GELU).  Tokens [B,N,C] are NHWC images, so the encoder never transposes.

Forward, input-gradient and parameter-gradient passes; DropPath (timm semantics, core/mix_transformer.py:126) is the
identity in eval and a per-sample keep/scale in train mode (counter-based stream, ops.DROP_RNG).
"""
import math
from functools import partial

import torch
import torch.nn as nn

from .. import ops
from ..operations_m import Conv2dParams, LayerNormParams, LinearParams, _PackCache


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _init_weights(m):
    """core/mix_transformer.py:29-43 (identical in every class there)."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
        fan_out //= m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            m.bias.data.zero_()


class DropPath(nn.Module):
    """Stochastic depth (timm.models.layers.DropPath, core/mix_transformer.py:126,152-153): identity when p == 0 or in eval;
    in train mode each sample's branch is kept with probability 1-p and scaled by 1/(1-p).  The per-sample factors come
    from the counter-based stream ops.DROP_RNG (reproducible from seed / rank / step); `Block` applies them."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def sample_scale(self, B, device):
        """-> per-sample factors [B] (0 or 1/(1-p)), or None when the layer is the identity."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        return ops.DROP_RNG.keep_mask(B, self.drop_prob, device)

    def forward(self, x):
        s = self.sample_scale(x.shape[0], x.device)
        return x if s is None else ops.rowscale_add(x.contiguous(), s)


class DWConv(nn.Module):
    """core/mix_transformer.py:376-387 (parameter holder; fused with GELU in Mlp)."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = Conv2dParams(dim, dim, 3, 1, 1, bias=True, groups=dim)


class Mlp(nn.Module):
    """core/mix_transformer.py:18-53: fc1 -> DWConv -> GELU -> fc2 (drop = 0)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU or drop != 0.:
            raise NotImplementedError("Mlp: only GELU / drop=0 (every mit_b* config)")
        self.fc1 = LinearParams(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.act = act_layer()
        self.fc2 = LinearParams(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self._packs = _PackCache()
        self.apply(_init_weights)

    def forward_tokens(self, x, H, W, res, tape=None):
        """x [B,N,C] -> fc2(gelu(dwconv(fc1 x))) + res   (res None: the bare branch)."""
        B, N, _ = x.shape
        hid1 = ops.gemm(x, self.fc1.weight, shift=self.fc1.bias)
        hid = ops.dwconv3_bias_gelu(hid1.view(B, H, W, -1), self.dwconv.dwconv.weight, self.dwconv.dwconv.bias).view(B, N, -1)
        if tape is not None:
            tape.update(hid1=hid1, H=H, W=W)
            if ops.taping_wgrad():
                tape.update(x=x, hid=hid)
        return ops.gemm(hid, self.fc2.weight, shift=self.fc2.bias, res=res)

    def backward_tokens(self, dy, tape, wgrad=False):
        """dy = d/d(fc2 output) -> d/d(x); wgrad: fc1 / dwconv / fc2 parameter gradients too."""
        B, N, _ = dy.shape
        H, W = tape["H"], tape["W"]
        fc2t = self._packs.get("fc2T", [self.fc2.weight], lambda: ops.transpose_pad(self.fc2.weight))
        fc1t = self._packs.get("fc1T", [self.fc1.weight], lambda: ops.transpose_pad(self.fc1.weight))
        d_h2 = ops.gemm(dy, fc2t)
        dw = self.dwconv.dwconv
        hid1 = tape["hid1"].view(B, H, W, -1)
        if not wgrad:
            d_h1 = ops.dwconv3_bias_gelu_bwd(hid1, dw.weight, dw.bias, d_h2.view(B, H, W, -1)).view(B, N, -1)
            return ops.gemm(d_h1, fc1t)
        _linear_wgrad(self.fc2, dy, tape["hid"])
        d_h1, d_pre = ops.dwconv3_bias_gelu_bwd(hid1, dw.weight, dw.bias, d_h2.view(B, H, W, -1), want_dpre=True)
        gw, gb = ops.grad_of(dw.weight), ops.grad_of(dw.bias)
        if gw is not None or gb is not None:
            C = hid1.shape[-1]
            ops.dwconv_wgrad(hid1, d_pre, 3, 1, False, gw if gw is not None else torch.zeros(C * 9, device=dy.device), gb)
        d_h1 = d_h1.view(B, N, -1)
        _linear_wgrad(self.fc1, d_h1, tape["x"])
        return ops.gemm(d_h1, fc1t)


def _linear_wgrad(lin, dy, x):
    """nn.Linear parameter gradients accumulated into .grad (dW = dy^T x, db = column sums of dy)."""
    gw, gb = ops.grad_of(lin.weight), ops.grad_of(lin.bias) if lin.bias is not None else None
    if gw is None and gb is None:
        return
    if gw is None:
        ops.colsum(dy, gb)
        return
    ops.gemm_wgrad(dy, x, out_w=gw, out_b=gb)


def _ln_wgrad(norm, x, dy):
    """nn.LayerNorm affine gradients accumulated into .grad."""
    gg, gb = ops.grad_of(norm.weight), ops.grad_of(norm.bias)
    if gg is None and gb is None:
        return
    C = x.shape[-1]
    z = lambda: torch.zeros(C, device=x.device, dtype=torch.float32)
    ops.layernorm_wgrad(x, dy, norm.eps, gg if gg is not None else z(), gb if gb is not None else z())


def _conv_as_gemm_wgrad(conv, d_out, col):
    """Gradients of a conv evaluated as im2col + GEMM (patch embeddings, SR convs): d_out [.., Cout], col [.., Kpad]."""
    gw, gb = ops.grad_of(conv.weight), ops.grad_of(conv.bias) if conv.bias is not None else None
    if gw is None and gb is None:
        return
    Cout, kpad = d_out.shape[-1], col.shape[-1]
    dwp, _ = ops.gemm_wgrad(d_out.reshape(-1, Cout), col.reshape(-1, kpad), want_bias=False)
    if gb is not None:
        ops.colsum(d_out, gb)
    if gw is not None:
        ops.unpack_conv_gemm_wgrad(dwp, gw)


class Attention(nn.Module):
    """core/mix_transformer.py:56-115: spatial-reduction attention."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., sr_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        if qk_scale is not None or attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError("Attention: qk_scale / dropout are unused by every mit_b* config")
        self.dim = dim
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.q = LinearParams(dim, dim, bias=qkv_bias)
        self.kv = LinearParams(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = LinearParams(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = Conv2dParams(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = LayerNormParams(dim)
        self._packs = _PackCache()
        self.apply(_init_weights)

    def forward_tokens(self, x, H, W, res, tape=None):
        """x = norm1(tokens) [B,N,C]; returns proj(attn) + res   (res None: the bare branch)."""
        B, N, C = x.shape
        q = ops.gemm(x, self.q.weight, shift=self.q.bias)
        x_sr = None
        if self.sr_ratio > 1:
            sr = self.sr_ratio
            wsr = self._packs.get("sr", [self.sr.weight], lambda: ops.pack_conv_gemm_weight(self.sr.weight))
            x_sr, _, _ = ops.conv_gemm(x.view(B, H, W, C), wsr, sr, sr, 0, shift=self.sr.bias)   # the im2col matrix is gathered in the GEMM's loader
            x_ = ops.layernorm(x_sr, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            x_ = x
        kv = ops.gemm(x_, self.kv.weight, shift=self.kv.bias)
        if tape is None:
            o = ops.sr_attention(q, kv, self.num_heads)
        else:
            o, lse = ops.sr_attention(q, kv, self.num_heads, want_lse=True)
            tape.update(q=q, kv=kv, o=o, lse=lse, x_sr=x_sr, H=H, W=W)
            if ops.taping_wgrad():
                tape.update(x=x, x_kv=x_)
        return ops.gemm(o, self.proj.weight, shift=self.proj.bias, res=res)

    def backward_tokens(self, dy, tape, wgrad=False):
        """dy = d/d(proj output) -> d/d(x) where x = norm1(tokens); wgrad: q / kv / sr / norm / proj parameter gradients too."""
        B, N, C = dy.shape
        H, W = tape["H"], tape["W"]
        pt = self._packs.get("projT", [self.proj.weight], lambda: ops.transpose_pad(self.proj.weight))
        qt = self._packs.get("qT", [self.q.weight], lambda: ops.transpose_pad(self.q.weight))
        kvt = self._packs.get("kvT", [self.kv.weight], lambda: ops.transpose_pad(self.kv.weight))
        d_o = ops.gemm(dy, pt)
        dq, dkv = ops.sr_attention_bwd(tape["q"], tape["kv"], tape["o"], d_o, tape["lse"], self.num_heads)
        d_xkv = ops.gemm(dkv, kvt)
        if wgrad:
            _linear_wgrad(self.proj, dy, tape["o"])
            _linear_wgrad(self.kv, dkv, tape["x_kv"])
            _linear_wgrad(self.q, dq, tape["x"])
        if self.sr_ratio > 1:
            sr = self.sr_ratio
            wsr = self._packs.get("sr", [self.sr.weight], lambda: ops.pack_conv_gemm_weight(self.sr.weight))
            wsrt = self._packs.get("srT", [self.sr.weight], lambda: ops.transpose_pad(wsr))
            if wgrad:
                _ln_wgrad(self.norm, tape["x_sr"], d_xkv)
            d_sr = ops.layernorm_bwd(tape["x_sr"], self.norm.weight, d_xkv, self.norm.eps)
            if wgrad:   # the SR conv's im2col columns are rebuilt from the saved input (cheaper than keeping them)
                _conv_as_gemm_wgrad(self.sr, d_sr, ops.im2col(tape["x"].view(B, H, W, C), sr, sr, 0, wsr.shape[1]))
            d_xkv = ops.gemm_col2im(d_sr.reshape(-1, d_sr.shape[-1]), wsrt, B, H, W, C, sr).view(B, N, C)   # col2im by the GEMM epilogue's addresses
        return ops.gemm(dq, qt, res=d_xkv)


class Block(nn.Module):
    """core/mix_transformer.py:118-155."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, sr_ratio=1):
        super().__init__()
        self.norm1 = _make_norm(norm_layer, dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, sr_ratio=sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = _make_norm(norm_layer, dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.apply(_init_weights)

    def forward_tokens(self, x, H, W, tape=None):
        """x + drop_path(attn(norm1(x))), then + drop_path(mlp(norm2(.)))   (core/mix_transformer.py:151-155)."""
        t_attn = t_mlp = None
        if tape is not None:
            t_attn, t_mlp = {}, {}
        dp = self.drop_path if isinstance(self.drop_path, DropPath) else None
        B = x.shape[0]
        ln1 = ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        s1 = dp.sample_scale(B, x.device) if dp is not None else None
        if s1 is None:
            x1 = self.attn.forward_tokens(ln1, H, W, res=x, tape=t_attn)
        else:   # stochastic depth: the residual add moves out of the GEMM epilogue
            x1 = ops.rowscale_add(self.attn.forward_tokens(ln1, H, W, res=None, tape=t_attn), s1, x)
        ln2 = ops.layernorm(x1, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        s2 = dp.sample_scale(B, x.device) if dp is not None else None
        if s2 is None:
            x2 = self.mlp.forward_tokens(ln2, H, W, res=x1, tape=t_mlp)
        else:
            x2 = ops.rowscale_add(self.mlp.forward_tokens(ln2, H, W, res=None, tape=t_mlp), s2, x1)
        if tape is not None:
            tape.append(dict(x=x, x1=x1, attn=t_attn, mlp=t_mlp, s1=s1, s2=s2))
        return x2

    def backward_tokens(self, d_x2, t, wgrad=False):
        d_b2 = d_x2 if t["s2"] is None else ops.rowscale_add(d_x2, t["s2"])
        d_ln2 = self.mlp.backward_tokens(d_b2, t["mlp"], wgrad)
        if wgrad:
            _ln_wgrad(self.norm2, t["x1"], d_ln2)
        d_x1 = ops.layernorm_bwd(t["x1"], self.norm2.weight, d_ln2, self.norm2.eps, add=d_x2)
        d_b1 = d_x1 if t["s1"] is None else ops.rowscale_add(d_x1, t["s1"])
        d_ln1 = self.attn.backward_tokens(d_b1, t["attn"], wgrad)
        if wgrad:
            _ln_wgrad(self.norm1, t["x"], d_ln1)
        d_x = ops.layernorm_bwd(t["x"], self.norm1.weight, d_ln1, self.norm1.eps, add=d_x1)
        if wgrad:
            ops.grads_ready(self)
        return d_x


def _make_norm(norm_layer, dim):
    """norm_layer is nn.LayerNorm or partial(nn.LayerNorm, eps=...) in the reference; keep eps, hold params."""
    probe = norm_layer(dim)
    if not isinstance(probe, nn.LayerNorm):
        raise NotImplementedError("only LayerNorm norm layers")
    return LayerNormParams(dim, eps=probe.eps)


class OverlapPatchEmbed(nn.Module):
    """core/mix_transformer.py:158-198: strided conv (im2col + MFMA GEMM) + LayerNorm."""

    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.stride = stride
        self.H, self.W = img_size[0] // patch_size[0], img_size[1] // patch_size[1]
        self.num_patches = self.H * self.W
        self.proj = Conv2dParams(in_chans, embed_dim, kernel_size=patch_size, stride=stride,
                                 padding=(patch_size[0] // 2, patch_size[1] // 2))
        self.norm = LayerNormParams(embed_dim)
        self._packs = _PackCache()
        self.apply(_init_weights)

    def forward_nhwc(self, x, tape=None):
        """x NHWC [B,H,W,Cin] -> tokens [B, OH*OW, D], OH, OW."""
        k = self.patch_size[0]
        w = self._packs.get("w", [self.proj.weight], lambda: ops.pack_conv_gemm_weight(self.proj.weight))
        t, OH, OW = ops.conv_gemm(x, w, k, self.stride, k // 2, shift=self.proj.bias)
        if tape is not None:
            tape.append(dict(pre=t, in_shape=tuple(x.shape), x_in=x if ops.taping_wgrad() else None))
        return ops.layernorm(t, self.norm.weight, self.norm.bias, self.norm.eps), OH, OW

    def backward_nhwc(self, d_tok, t, wgrad=False):
        """d/d(tokens after the norm) -> d/d(input map) NHWC; wgrad: proj / norm parameter gradients too."""
        k = self.patch_size[0]
        B, H, W, Cin = t["in_shape"]
        w = self._packs.get("w", [self.proj.weight], lambda: ops.pack_conv_gemm_weight(self.proj.weight))
        wt = self._packs.get("wT", [self.proj.weight], lambda: ops.transpose_pad(w))
        if wgrad:
            _ln_wgrad(self.norm, t["pre"], d_tok)
        d_pre = ops.layernorm_bwd(t["pre"], self.norm.weight, d_tok, self.norm.eps)
        if wgrad:
            _conv_as_gemm_wgrad(self.proj, d_pre, ops.im2col(t["x_in"], k, self.stride, k // 2, w.shape[1]))
        d_col = ops.gemm(d_pre, wt)
        if wgrad:
            ops.grads_ready(self)
        return ops.col2im(d_col, B, H, W, Cin, k, self.stride, k // 2)


class MixVisionTransformer(nn.Module):
    """core/mix_transformer.py:201-375."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=[64, 128, 256, 512],
                 num_heads=[1, 2, 4, 8], mlp_ratios=[4, 4, 4, 4], qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm,
                 depths=[3, 4, 6, 3], sr_ratios=[8, 4, 2, 1]):
        super().__init__()
        self.num_classes = num_classes
        self.depths = depths
        self.embed_dims = embed_dims
        self.patch_embed1 = OverlapPatchEmbed(img_size=img_size, patch_size=7, stride=4, in_chans=in_chans, embed_dim=embed_dims[0])
        self.patch_embed2 = OverlapPatchEmbed(img_size=img_size // 4, patch_size=3, stride=2, in_chans=embed_dims[0], embed_dim=embed_dims[1])
        self.patch_embed3 = OverlapPatchEmbed(img_size=img_size // 8, patch_size=3, stride=2, in_chans=embed_dims[1], embed_dim=embed_dims[2])
        self.patch_embed4 = OverlapPatchEmbed(img_size=img_size // 16, patch_size=3, stride=2, in_chans=embed_dims[2], embed_dim=embed_dims[3])
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for s in range(4):
            blocks = nn.ModuleList([Block(
                dim=embed_dims[s], num_heads=num_heads[s], mlp_ratio=mlp_ratios[s], qkv_bias=qkv_bias, qk_scale=qk_scale,
                drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[cur + i], norm_layer=norm_layer,
                sr_ratio=sr_ratios[s]) for i in range(depths[s])])
            setattr(self, "block%d" % (s + 1), blocks)
            setattr(self, "norm%d" % (s + 1), _make_norm(norm_layer, embed_dims[s]))
            cur += depths[s]
        self.apply(_init_weights)

    def reset_drop_path(self, drop_path_rate):
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(self.depths))]
        cur = 0
        for s in range(4):
            for i in range(self.depths[s]):
                getattr(self, "block%d" % (s + 1))[i].drop_path.drop_prob = dpr[cur + i]
            cur += self.depths[s]

    def freeze_patch_emb(self):
        self.patch_embed1.requires_grad = False

    def forward_features_nhwc(self, x, tape=None):
        """x NHWC [B,H,W,3] -> 4 NHWC stage outputs [B,H/4,W/4,D0] ... [B,H/32,W/32,D3].
        tape (list): filled with what the input-gradient pass needs (one dict per stage)."""
        outs = []
        B = x.shape[0]
        for s in range(4):
            st = None if tape is None else dict(pe=[], blocks=[])
            t, H, W = getattr(self, "patch_embed%d" % (s + 1)).forward_nhwc(x, None if st is None else st["pe"])
            for blk in getattr(self, "block%d" % (s + 1)):
                t = blk.forward_tokens(t, H, W, None if st is None else st["blocks"])
            n = getattr(self, "norm%d" % (s + 1))
            if st is not None:
                st.update(pre_norm=t, H=H, W=W)
                tape.append(st)
            t = ops.layernorm(t, n.weight, n.bias, n.eps)
            x = t.view(B, H, W, -1)
            outs.append(x)
        return outs

    def backward_features_nhwc(self, d_feats, tape, wgrad=False):
        """d/d(4 stage outputs, NHWC) -> d/d(input) NHWC [B,H,W,3]; wgrad: every encoder parameter gradient too."""
        d_next = None
        for s in (3, 2, 1, 0):
            st = tape[s]
            d_out = d_feats[s] if d_next is None else ops.add(d_feats[s].contiguous(), d_next)
            B = d_out.shape[0]
            n = getattr(self, "norm%d" % (s + 1))
            d_out = d_out.reshape(B, st["H"] * st["W"], -1)
            if wgrad:
                _ln_wgrad(n, st["pre_norm"], d_out)
                ops.grads_ready(n)
            d_t = ops.layernorm_bwd(st["pre_norm"], n.weight, d_out, n.eps)
            blocks = getattr(self, "block%d" % (s + 1))
            for i in range(len(blocks) - 1, -1, -1):
                d_t = blocks[i].backward_tokens(d_t, st["blocks"][i], wgrad)
            d_next = getattr(self, "patch_embed%d" % (s + 1)).backward_nhwc(d_t, st["pe"][0], wgrad)
        return d_next

    def forward_features(self, x):
        ops.require_no_grad(x)
        with torch.no_grad():
            xin = ops.nchw_to_nhwc(x)
            return [ops.to_nchw_view(o) for o in self.forward_features_nhwc(xin)]

    def forward(self, x):
        return self.forward_features(x)


def _mit(embed_dims, depths):
    def ctor(self, **kwargs):
        MixVisionTransformer.__init__(
            self, patch_size=4, embed_dims=embed_dims, num_heads=[1, 2, 5, 8], mlp_ratios=[4, 4, 4, 4], qkv_bias=True,
            norm_layer=partial(nn.LayerNorm, eps=1e-6), depths=depths, sr_ratios=[8, 4, 2, 1], drop_rate=0.0,
            drop_path_rate=0.1)
    return ctor


# core/mix_transformer.py:389-433
mit_b0 = type("mit_b0", (MixVisionTransformer,), {"__init__": _mit([32, 64, 160, 256], [2, 2, 2, 2])})
mit_b1 = type("mit_b1", (MixVisionTransformer,), {"__init__": _mit([64, 128, 320, 512], [2, 2, 2, 2])})
mit_b2 = type("mit_b2", (MixVisionTransformer,), {"__init__": _mit([64, 128, 320, 512], [3, 4, 6, 3])})
mit_b3 = type("mit_b3", (MixVisionTransformer,), {"__init__": _mit([64, 128, 320, 512], [3, 4, 18, 3])})
mit_b4 = type("mit_b4", (MixVisionTransformer,), {"__init__": _mit([64, 128, 320, 512], [3, 8, 27, 3])})
mit_b5 = type("mit_b5", (MixVisionTransformer,), {"__init__": _mit([64, 128, 320, 512], [3, 6, 40, 3])})
