"""TEST INFRASTRUCTURE -- CPU oracle for the PAIF hot path.  NOT product code.

A plain fp32 torch-CPU *functional* restatement of the reference's algorithm for the path
named in BASELINE.json (fusion net forward -> colour glue -> SegFormer -> PGD loop).  Every
function cites the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product (paif_amd/) never does and
fails loudly when its HIP library is missing.

Pinning status
--------------
* Pinned against the reference ITSELF, imported in the build container with oracle/shims
  (oracle/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py, and live in
  tests/test_oracle_vs_reference.py when /root/reference is present).  The reference has no
  tests, golden vectors or fixtures of its own.
* PARITY UNPINNED at four third-party boundaries whose source is not under /root/reference
  and whose packages are not installable here: guided_filter_pytorch.GuidedFilter (version
  unpinned by the reference; restated from He et al. + the package's cumsum box filter),
  mmcv.cnn.ConvModule (conv(no bias)->BN->ReLU), timm DropPath/trunc_normal_ (eval identity /
  init only) and sklearn.metrics.confusion_matrix.  oracle/shims/* are this build's spec of
  record for those.

All tensors are NCHW float32 on CPU; `sd` is a flat state_dict (key -> tensor) with exactly
the reference's key layout; `p` is a key prefix such as
"enhance_net." or "denoise_net.".
"""
import collections
import math

import numpy as np
import torch
import torch.nn.functional as F

Genotype = collections.namedtuple(
    "Genotype", "normal_1 normal_1_concat normal_2 normal_2_concat normal_3 normal_3_concat"
)  # test_original.py:709

# the only genotype either entry script instantiates (test_original.py:711-713, robust_test.py:255-257)
FUSION_AT = Genotype(
    normal_1=[("Denseblocks_3_1", 0), ("DilConv_3_2", 1)], normal_1_concat=[1, 2],
    normal_2=[("Denseblocks_3_1", 0), ("Denseblocks_3_1", 1)], normal_2_concat=[1, 2],
    normal_3=[("ECAattention_3", 0), ("Residualblocks_7_1", 1)], normal_3_concat=[1, 2],
)

MIT_CFG = {  # core/mix_transformer.py:389-433
    "mit_b0": dict(dims=[32, 64, 160, 256], heads=[1, 2, 5, 8], depths=[2, 2, 2, 2], sr=[8, 4, 2, 1]),
    "mit_b1": dict(dims=[64, 128, 320, 512], heads=[1, 2, 5, 8], depths=[2, 2, 2, 2], sr=[8, 4, 2, 1]),
    "mit_b2": dict(dims=[64, 128, 320, 512], heads=[1, 2, 5, 8], depths=[3, 4, 6, 3], sr=[8, 4, 2, 1]),
    "mit_b3": dict(dims=[64, 128, 320, 512], heads=[1, 2, 5, 8], depths=[3, 4, 18, 3], sr=[8, 4, 2, 1]),
    "mit_b4": dict(dims=[64, 128, 320, 512], heads=[1, 2, 5, 8], depths=[3, 8, 27, 3], sr=[8, 4, 2, 1]),
    "mit_b5": dict(dims=[64, 128, 320, 512], heads=[1, 2, 5, 8], depths=[3, 6, 40, 3], sr=[8, 4, 2, 1]),
}


# --------------------------------------------------------------------------------------
# colour transforms -- core/model_fusion_auto.py:69-111
# --------------------------------------------------------------------------------------
def rgb2ycrcb(x):
    """core/model_fusion_auto.py:69-92."""
    R, G, B = x[:, 0:1], x[:, 1:2], x[:, 2:3]
    Y = 0.299 * R + 0.587 * G + 0.114 * B
    Cr = (R - Y) * 0.713 + 0.5
    Cb = (B - Y) * 0.564 + 0.5
    return torch.cat([Y, Cr, Cb], dim=1)


def ycrcb2rgb(x):
    """core/model_fusion_auto.py:94-111: (x + bias) @ mat on NHWC-flattened pixels."""
    mat = torch.tensor([[1.0, 1.0, 1.0], [1.403, -0.714, 0.0], [0.0, -0.344, 1.773]], dtype=x.dtype)
    bias = torch.tensor([0.0 / 255, -0.5, -0.5], dtype=x.dtype)
    B, _, H, W = x.shape
    flat = x.permute(0, 2, 3, 1).reshape(-1, 3)
    out = (flat + bias).mm(mat)
    return out.reshape(B, H, W, 3).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------
# guided filter -- third party (see header); call sites core/model_fusion_auto.py:522-535
# --------------------------------------------------------------------------------------
def _diff(t, r, dim):
    n = t.shape[dim]
    left = t.narrow(dim, r, r + 1)
    middle = t.narrow(dim, 2 * r + 1, n - 2 * r - 1) - t.narrow(dim, 0, n - 2 * r - 1)
    right = t.narrow(dim, n - 1, 1) - t.narrow(dim, n - 2 * r - 1, r)
    return torch.cat([left, middle, right], dim=dim)


def box_filter(x, r):
    """Border-clipped (2r+1)^2 window sums as cumsum-then-difference (package behaviour)."""
    return _diff(_diff(x.cumsum(dim=2), r, 2).cumsum(dim=3), r, 3)


def guided_filter(guide, y, r, eps):
    """He et al. guided filter with a 1-channel guide."""
    h, w = guide.shape[2:]
    assert h > 2 * r + 1 and w > 2 * r + 1
    N = box_filter(guide.new_ones((1, 1, h, w)), r)
    mean_x = box_filter(guide, r) / N
    mean_y = box_filter(y, r) / N
    cov_xy = box_filter(guide * y, r) / N - mean_x * mean_y
    var_x = box_filter(guide * guide, r) / N - mean_x * mean_x
    A = cov_xy / (var_x + eps)
    b = mean_y - A * mean_x
    return (box_filter(A, r) / N) * guide + box_filter(b, r) / N


# --------------------------------------------------------------------------------------
# NAS-cell primitives -- operations_m.py
# --------------------------------------------------------------------------------------
_PAD = {(3, 1): 1, (3, 2): 2, (5, 1): 2, (5, 2): 4, (7, 1): 3, (7, 2): 6}  # operations_m.py:121-132


def basic_conv(x, w, k, d, groups=1):
    """BasicConv with relu=False, bn=False (the only form on the path), operations_m.py:114-145."""
    return F.conv2d(x, w, None, 1, _PAD.get((k, d), 0), d, groups)


class TrainCtx:
    """Train-mode knobs of the oracle (BASELINE configs[4], the adversarial-training step): while `TRAIN` holds one,
    BatchNorm uses batch statistics and updates the running ones in `sd` (nn.BatchNorm2d defaults: momentum 0.1), and
    DropPath (timm semantics, core/mix_transformer.py:126,152-153) / Dropout2d(0.1) (core/segformer_head.py:47,79) draw their
    keep masks from the counter-based stream the product uses (paif_amd.ops.DropRNG: u = hash_uniform(seed, offset + i),
    keep iff u >= p, scaled by 1/(1-p)) -- so both sides drop the same samples / channels."""

    def __init__(self, seed=0, rank=0, step=0, drop_path_rate=0.1, dropout=0.1):
        self.seed = (int(seed) * 1000003 + int(rank) * 7919 + int(step) * 104729) & 0xFFFFFFFF
        self.offset = 0
        self.drop_path_rate, self.dropout = drop_path_rate, dropout

    def keep(self, n, p):
        from paif_amd.synthetic import hash_uniform
        u = hash_uniform(self.seed, n, offset=self.offset)
        self.offset += n
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
        return torch.from_numpy(np.where(u >= np.float64(np.float32(p)), scale, np.float32(0.0)).astype(np.float32))


TRAIN = None   # a TrainCtx while the oracle runs in train mode


def bn_eval(x, sd, p, eps=1e-5):
    """BatchNorm2d: running statistics in eval mode; batch statistics + running update while TRAIN is set."""
    if TRAIN is not None:
        if p + "num_batches_tracked" in sd:
            sd[p + "num_batches_tracked"] += 1
        return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], True, 0.1, eps)
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        False, 0.0, eps)


def residual_dense_block(x, sd, p, k, d):
    """operations_m.py:435-449."""
    a = sd[p + "lrelu.weight"]
    x1 = F.prelu(basic_conv(x, sd[p + "conv1.conv.weight"], k, d), a)
    x2 = F.prelu(basic_conv(torch.cat((x, x1), 1), sd[p + "conv2.conv.weight"], k, d), a)
    x3 = F.prelu(basic_conv(torch.cat((x, x1, x2), 1), sd[p + "conv3.conv.weight"], k, d), a)
    return x3 * 0.333333 + x


def residual_module(x, sd, p, k, d):
    """operations_m.py:451-464."""
    r = basic_conv(x, sd[p + "op.0.conv.weight"], k, d)
    r = F.conv2d(r, sd[p + "op.1.weight"], None, 1, 2, 2)
    r = F.conv2d(r, sd[p + "op.2.weight"])
    r = bn_eval(r, sd, p + "op.3.")
    r = F.prelu(r, sd[p + "op.4.weight"])
    return x + r


def dil_conv(x, sd, p, k, d):
    """operations_m.py:494-506."""
    C = x.shape[1]
    r = F.relu(x)
    r = basic_conv(r, sd[p + "op.1.conv.weight"], k, d, groups=C)
    r = F.conv2d(r, sd[p + "op.2.weight"])
    r = bn_eval(r, sd, p + "op.3.")
    return r + x


def sep_conv(x, sd, p, k):
    """operations_m.py:509-525 (stride 1, padding k//2); note: no residual."""
    C = x.shape[1]
    r = F.relu(x)
    r = F.conv2d(r, sd[p + "op.1.weight"], None, 1, k // 2, 1, C)
    r = F.conv2d(r, sd[p + "op.2.weight"])
    r = bn_eval(r, sd, p + "op.3.")
    r = F.relu(r)
    r = F.conv2d(r, sd[p + "op.5.weight"], None, 1, k // 2, 1, C)
    r = F.conv2d(r, sd[p + "op.6.weight"])
    return bn_eval(r, sd, p + "op.7.")


def eca_basic_block(x, sd, p, k):
    """operations_m.py:340-393.  Residual is conv1's OUTPUT, not the block input."""
    a = sd[p + "relu.weight"]
    r = F.conv2d(x, sd[p + "conv1.weight"], None, 1, 1)
    o = F.prelu(r, a)
    o = basic_conv(o, sd[p + "conv2.conv.weight"], k, 1)
    y = o.mean(dim=(2, 3), keepdim=True)  # AdaptiveAvgPool2d(1)
    y = F.conv1d(y.squeeze(-1).transpose(-1, -2), sd[p + "se.conv.weight"], None, 1, (k - 1) // 2)
    y = torch.sigmoid(y.transpose(-1, -2).unsqueeze(-1))
    o = o * y
    return F.prelu(o + r, a)


def spatial_basic_block(x, sd, p, k):
    """operations_m.py:148-204 (SPAattention; in the search space, not in the shipped genotype)."""
    a = sd[p + "relu.weight"]
    r = F.conv2d(x, sd[p + "conv1.weight"], None, 1, 1)
    o = F.prelu(r, a)
    o = basic_conv(o, sd[p + "conv2.conv.weight"], k, 1)
    comp = torch.cat((o.max(1, keepdim=True)[0], o.mean(1, keepdim=True)), 1)
    s = torch.sigmoid(basic_conv(comp, sd[p + "se.spatial.conv.weight"], k, 1))
    o = o * s
    return F.prelu(o + r, a)


def parse_primitive(primitive):
    """MixedOp's parser, core/model_fusion_auto.py:397-415: 'Name_k' iff the name contains
    lowercase 'attention', else 'Name_k_d' (IndexError when a field is missing)."""
    parts = primitive.split("_")
    name, kernel, dilation = parts[0], int(parts[1]), 1
    if primitive.find("attention") == -1:
        dilation = int(parts[2])
    return name, kernel, dilation


def mixed_op(x, sd, p, primitive):
    """OPS registry dispatch, operations_m.py:9-18."""
    name, k, d = parse_primitive(primitive)
    p = p + "_op."
    if name == "Denseblocks":
        return residual_dense_block(x, sd, p, k, d)
    if name == "Residualblocks":
        return residual_module(x, sd, p, k, d)
    if name == "ECAattention":
        return eca_basic_block(x, sd, p, k)
    if name == "SPAattention":
        return spatial_basic_block(x, sd, p, k)
    if name == "DilConv":
        return dil_conv(x, sd, p, k, d)
    if name == "SepConv":
        return sep_conv(x, sd, p, k)
    raise KeyError(name)


def cell_chain(x, sd, p, types):
    """core/model_fusion_auto.py:418-445: inp + ops(inp)."""
    s = x
    for i, (prim, _) in enumerate(types):
        s = mixed_op(s, sd, "%s_ops.%d." % (p, i), prim)
    return x + s


def get_residue(t):
    """core/model_fusion_auto.py:517-521."""
    return t.max(dim=1, keepdim=True)[0] - t.min(dim=1, keepdim=True)[0]


def decomposition(x):
    """core/model_fusion_auto.py:522-535: r=4, eps in {1e-3, 1e-4}."""
    res = get_residue(x)
    LF, HF = [], []
    for eps in (0.001, 0.0001):
        lf = guided_filter(res, x, 4, eps)
        LF.append(lf)
        HF.append(x - lf)
    return torch.cat(LF, 1), torch.cat(HF, 1)


def cell_decom(fir, fvis, sd, p, geno, inter=None):
    """core/model_fusion_auto.py:492-516."""
    lf_ir, hf_ir = decomposition(fir)
    lf_vis, hf_vis = decomposition(fvis)
    lf = F.conv2d(torch.cat([lf_ir, hf_ir], 1), sd[p + "conv1x1_lf.weight"], sd[p + "conv1x1_lf.bias"])
    hf = F.conv2d(torch.cat([lf_vis, hf_vis], 1), sd[p + "conv1x1_hf.weight"], sd[p + "conv1x1_hf.bias"])
    lf_re = cell_chain(lf, sd, p + "chain.", geno.normal_1)
    hf_re = cell_chain(hf, sd, p + "chain2.", geno.normal_2)
    if inter is not None:
        inter.update(lf_ir=lf_ir, lf_vis=lf_vis, lf=lf, hf=hf, lf_re=lf_re, hf_re=hf_re)
    return lf_re + fir, hf_re + fvis


def spatial_attn_m(ir, vis, sd, p):
    """ChannelPool(2-arg) + spatial_attn_layer_M, core/model_fusion_auto.py:1352-1368."""
    comp = torch.cat((ir.max(1, keepdim=True)[0], ir.mean(1, keepdim=True),
                      vis.max(1, keepdim=True)[0], vis.mean(1, keepdim=True)), 1)
    return torch.sigmoid(basic_conv(comp, sd[p + "spatial.conv.weight"], 5, 1))


def fusion_forward(ir, vis_y, sd, p="", geno=FUSION_AT, inter=None):
    """Network_Fusion_Searched.forward, core/model_fusion_auto.py:625-635."""
    vis_y = vis_y[:, 0:1]
    ir = ir[:, 0:1]
    fir = F.prelu(F.conv2d(ir, sd[p + "stem_1.0.weight"], None, 1, 1), sd[p + "stem_1.1.weight"])
    fvis = F.prelu(F.conv2d(vis_y, sd[p + "stem_2.0.weight"], None, 1, 1), sd[p + "stem_2.1.weight"])
    ir_f, vis_f = cell_decom(fir, fvis, sd, p + "decompation.", geno, inter)
    scale = spatial_attn_m(ir_f, vis_f, sd, p + "spa.")
    agg = scale * ir_f + (1 - scale) * vis_f
    feat2 = cell_chain(agg, sd, p + "chain.", geno.normal_3)
    o = F.conv2d(feat2, sd[p + "stem_out.0.weight"], None, 1, 1)
    o = F.conv2d(o, sd[p + "stem_out.1.weight"], None, 1, 1)
    o = F.prelu(o, sd[p + "stem_out.2.weight"])
    out = torch.tanh(o)
    if inter is not None:
        inter.update(fir=fir, fvis=fvis, ir_feature=ir_f, vis_feature=vis_f, scale=scale, agg=agg,
                     feature2=feat2, fused=out)
    return out


def fused_image_uint8(fused, vis):
    """Fused-image writer post-processing, test_original.py:181-197: RGB recomposition with the visible chroma, clamp to
    [0,1], np.uint8(255*x) (truncation), batch-global min-max of the uint8 array in float64, np.uint8(255*x) again.
    fused [B,1,H,W], vis [B,3,H,W] -> uint8 [B,H,W,3]."""
    import numpy as np
    ycc = rgb2ycrcb(vis)
    rgb = ycrcb2rgb(torch.cat((fused, ycc[:, 1:2], ycc[:, 2:]), 1))
    rgb = torch.where(rgb > 1, torch.ones_like(rgb), rgb)
    rgb = torch.where(rgb < 0, torch.zeros_like(rgb), rgb)
    q = np.uint8(255.0 * rgb.detach().numpy()).transpose((0, 2, 3, 1))
    q = (q - np.min(q)) / (np.max(q) - np.min(q))
    return np.uint8(255.0 * q)


def fusion_forward2(ir, vis_y, sd, p="", geno=FUSION_AT):
    """Network_Fusion_Searched_showfeatures.forward2 (core/model_fusion_auto.py:669-679) with Cell_Decom_decom
    (:536-581): the fused image plus the decomposition intermediates (LF / HF of both eps concatenated, residue)."""
    inter = {}
    out = fusion_forward(ir, vis_y, sd, p, geno, inter)
    fir, fvis = inter["fir"], inter["fvis"]
    res_ir = fir.max(1, keepdim=True)[0] - fir.min(1, keepdim=True)[0]
    res_vis = fvis.max(1, keepdim=True)[0] - fvis.min(1, keepdim=True)[0]
    lf_ir, lf_vis = inter["lf_ir"], inter["lf_vis"]
    hf_ir = torch.cat([fir, fir], 1) - lf_ir
    hf_vis = torch.cat([fvis, fvis], 1) - lf_vis
    return out, inter["ir_feature"], inter["vis_feature"], lf_ir, hf_ir, res_ir, lf_vis, hf_vis, res_vis


# --------------------------------------------------------------------------------------
# MiT encoder + SegFormer head -- core/mix_transformer.py, core/segformer_head.py
# --------------------------------------------------------------------------------------
def _ln(x, sd, p, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], eps)


def mit_attention(x, H, W, sd, p, heads, sr):
    """core/mix_transformer.py:93-115."""
    B, N, C = x.shape
    hd = C // heads
    q = F.linear(x, sd[p + "q.weight"], sd[p + "q.bias"]).reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = x.permute(0, 2, 1).reshape(B, C, H, W)
        x_ = F.conv2d(x_, sd[p + "sr.weight"], sd[p + "sr.bias"], sr).reshape(B, C, -1).permute(0, 2, 1)
        x_ = _ln(x_, sd, p + "norm.", 1e-5)  # Attention.norm = nn.LayerNorm(dim): default eps (line 75)
    else:
        x_ = x
    kv = F.linear(x_, sd[p + "kv.weight"], sd[p + "kv.bias"]).reshape(B, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def mit_mlp(x, H, W, sd, p):
    """core/mix_transformer.py:46-53 + DWConv :376-387."""
    B, N, _ = x.shape
    x = F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"])
    C = x.shape[-1]
    x = x.transpose(1, 2).reshape(B, C, H, W)
    x = F.conv2d(x, sd[p + "dwconv.dwconv.weight"], sd[p + "dwconv.dwconv.bias"], 1, 1, 1, C)
    x = x.flatten(2).transpose(1, 2)
    x = F.gelu(x)
    return F.linear(x, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def _drop_path(y, dp):
    """timm DropPath: per-sample keep / scale in train mode (dp = this block's rate; 0 = nn.Identity, no draw)."""
    if TRAIN is None or dp == 0.0:
        return y
    return y * TRAIN.keep(y.shape[0], dp).to(y.dtype).view(-1, 1, 1)


def mit_block(x, H, W, sd, p, heads, sr, dp=0.0):
    """core/mix_transformer.py:151-155 (eval: DropPath = identity)."""
    x = x + _drop_path(mit_attention(_ln(x, sd, p + "norm1.", 1e-6), H, W, sd, p + "attn.", heads, sr), dp)
    x = x + _drop_path(mit_mlp(_ln(x, sd, p + "norm2.", 1e-6), H, W, sd, p + "mlp."), dp)
    return x


def mit_forward(x, sd, p, backbone):
    """MixVisionTransformer.forward_features, core/mix_transformer.py:312-348."""
    cfg = MIT_CFG[backbone]
    B = x.shape[0]
    outs = []
    nblk = sum(cfg["depths"])
    rate = TRAIN.drop_path_rate if TRAIN is not None else 0.0
    dpr = [x_.item() for x_ in torch.linspace(0, rate, nblk)]     # core/mix_transformer.py:245
    cur = 0
    for s in range(4):
        k, st = (7, 4) if s == 0 else (3, 2)
        pe = "%spatch_embed%d." % (p, s + 1)
        x = F.conv2d(x, sd[pe + "proj.weight"], sd[pe + "proj.bias"], st, k // 2)
        H, W = x.shape[2:]
        x = x.flatten(2).transpose(1, 2)
        x = _ln(x, sd, pe + "norm.", 1e-5)  # OverlapPatchEmbed.norm = nn.LayerNorm: default eps (line 172)
        for i in range(cfg["depths"][s]):
            x = mit_block(x, H, W, sd, "%sblock%d.%d." % (p, s + 1, i), cfg["heads"][s], cfg["sr"][s], dpr[cur + i])
        cur += cfg["depths"][s]
        x = _ln(x, sd, "%snorm%d." % (p, s + 1), 1e-6)
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs


def segformer_head(feats, sd, p):
    """core/segformer_head.py:59-82 (eval: Dropout2d = identity; ConvModule per oracle/shims/mmcv)."""
    c1, c2, c3, c4 = feats
    n = c4.shape[0]
    size = c1.shape[2:]

    def mlp(c, name):
        y = F.linear(c.flatten(2).transpose(1, 2), sd[p + name + ".proj.weight"], sd[p + name + ".proj.bias"])
        return y.permute(0, 2, 1).reshape(n, -1, c.shape[2], c.shape[3])

    _c4 = F.interpolate(mlp(c4, "linear_c4"), size=size, mode="bilinear", align_corners=False)
    _c3 = F.interpolate(mlp(c3, "linear_c3"), size=size, mode="bilinear", align_corners=False)
    _c2 = F.interpolate(mlp(c2, "linear_c2"), size=size, mode="bilinear", align_corners=False)
    _c1 = mlp(c1, "linear_c1")
    x = F.conv2d(torch.cat([_c4, _c3, _c2, _c1], 1), sd[p + "linear_fuse.conv.weight"])
    x = F.relu(bn_eval(x, sd, p + "linear_fuse.bn."))
    if TRAIN is not None and TRAIN.dropout > 0:                    # nn.Dropout2d: whole channels per sample
        x = x * TRAIN.keep(x.shape[0] * x.shape[1], TRAIN.dropout).to(x.dtype).view(x.shape[0], x.shape[1], 1, 1)
    return F.conv2d(x, sd[p + "linear_pred.weight"], sd[p + "linear_pred.bias"])


def wetr_forward(x, sd, p, backbone, inter=None):
    """WeTr.forward, core/model_fusion_auto.py:62-68 (the classifier result is discarded)."""
    feats = mit_forward(x, sd, p + "encoder.", backbone)
    if inter is not None:
        inter.update(c1=feats[0], c2=feats[1], c3=feats[2], c4=feats[3])
    return segformer_head(feats, sd, p + "decoder.")


# --------------------------------------------------------------------------------------
# composite model -- core/model_fusion_auto.py:698-806 / 1029-1137
# --------------------------------------------------------------------------------------
SEG_MEAN = [123.675, 116.28, 103.53]
SEG_STD = [58.395, 57.12, 57.375]


def seg_input_from_fused(fused, vis_ycrcb):
    """core/model_fusion_auto.py:715-727: recompose RGB, clamp, BATCH-GLOBAL min-max, x255, mean/std."""
    x = ycrcb2rgb(torch.cat((fused, vis_ycrcb[:, 1:2], vis_ycrcb[:, 2:]), dim=1))
    ones, zeros = torch.ones_like(x), torch.zeros_like(x)
    x = torch.where(x > ones, ones, x)
    x = torch.where(x < zeros, zeros, x)
    x = (x - torch.min(x)) / (torch.max(x) - torch.min(x))
    x = x * 255
    chans = [(x[:, i] - SEG_MEAN[i]) / SEG_STD[i] for i in range(3)]
    return torch.stack(chans, dim=1)


def model_forward(ir, vis, sd, backbone="mit_b3", geno=FUSION_AT, inter=None):
    """Network_MM_CompModel.forward / Network_MM_Searched.forward (:712-729 / :1043-1060)."""
    ycc = rgb2ycrcb(vis)
    fused = fusion_forward(ir[:, 0:1], ycc[:, 0:1], sd, "enhance_net.", geno, inter)
    seg_in = seg_input_from_fused(fused, ycc)
    if inter is not None:
        inter.update(seg_in=seg_in)
    seg = wetr_forward(seg_in, sd, "denoise_net.", backbone, inter)
    return fused, seg


def model_forward_object(ir, vis, sd, backbone="mit_b3", geno=FUSION_AT):
    """Network_MM_CompModel.forward_object / Network_MM_Searched.forward_object (core/model_fusion_auto.py:736-766 / :1067-1097):
    the fused plane is clamped to [0,1] and min-max normalised over the whole batch BEFORE the recomposition; that plane is what
    the method returns next to the segmentation map."""
    ycc = rgb2ycrcb(vis)
    fused = fusion_forward(ir[:, 0:1], ycc[:, 0:1], sd, "enhance_net.", geno)
    ones, zeros = torch.ones_like(fused), torch.zeros_like(fused)
    fused = torch.where(fused > ones, ones, fused)
    fused = torch.where(fused < zeros, zeros, fused)
    fused = (fused - torch.min(fused)) / (torch.max(fused) - torch.min(fused))
    seg = wetr_forward(seg_input_from_fused(fused, ycc), sd, "denoise_net.", backbone)
    return fused, seg


def detection_loss(ir, vis, labels, sd, backbone="mit_b3"):
    """_detection_loss (core/model_fusion_auto.py:796-800 / :1123-1128): CrossEntropyLoss(ignore_index=255) on forward_object's map."""
    _, seg = model_forward_object(ir, vis, sd, backbone)
    return F.cross_entropy(F.interpolate(seg, size=labels.shape[1:], mode="bilinear", align_corners=False), labels.long(), ignore_index=255)


# --------------------------------------------------------------------------------------
# attack -- attack/attack.py
# --------------------------------------------------------------------------------------
def seg_loss(outputs, labels):
    """Seg_loss, attack/attack.py:103-114: CrossEntropyLoss(ignore_index=255), mean over valid px."""
    return F.cross_entropy(outputs, labels.long(), ignore_index=255)


def attack_loss_value(outputs, label, attack_way, i, attack_iters):
    """The per-iteration loss of attack_both, attack/attack.py:447-499."""
    if attack_way == "PGD":
        return seg_loss(outputs, label)
    if attack_way == "segPGD":
        lamb = (i - 1) / (attack_iters * 2)
        pred = torch.max(outputs, 1).values.unsqueeze(1)  # NB: max VALUE vs label (reference quirk)
        mask_t = (pred == label.unsqueeze(1)).int()
        mask_f = (pred != label.unsqueeze(1)).int()
        return (1 - lamb) * seg_loss(mask_t * outputs, label) + lamb * seg_loss(mask_f * outputs, label)
    if attack_way == "cosPGD":
        pred = torch.squeeze(torch.max(outputs, 1).values).flatten()
        _label = torch.squeeze(label).flatten()
        cossim = F.cosine_similarity(pred, _label, dim=0)
        return cossim * seg_loss(outputs, label)
    if attack_way == "newPGD":
        # attack/attack.py:472-499: pred_t and pred_f are both overwritten with max_c(outputs) (:486-492), so the factor is
        # cos/cos of the SAME vector: exactly 1 in value, and autograd's two gradient terms cancel to rounding noise
        pred = torch.squeeze(torch.max(outputs, 1).values.unsqueeze(1)).flatten()
        _label = torch.squeeze(label).flatten()
        cos_t = F.cosine_similarity(pred, _label, dim=0)
        cos_f = F.cosine_similarity(pred, _label, dim=0)
        return (cos_t / cos_f) * seg_loss(outputs, label)
    raise KeyError(attack_way)


def attack_both(forward_fn, X_vis, X_ir, label, delta0_ir, delta0_vis, epsilon=8 / 255.0, alpha=2 / 255.0,
                attack_iters=50, attack_way="PGD", trace=None):
    """attack_both, attack/attack.py:417-514, with the start perturbation passed in (the
    reference draws it from the global RNG, :434,439).  Reproduces the reference's
    NEVER-ZEROED delta.grad: the update uses the sign of the RUNNING SUM of gradients (:501-512).

    forward_fn(ir, vis) -> (fused, seg_map)."""
    d_ir = torch.clamp(delta0_ir.clone(), 0 - X_ir, 1 - X_ir).requires_grad_(True)
    d_vis = torch.clamp(delta0_vis.clone(), 0 - X_vis, 1 - X_vis).requires_grad_(True)
    for i in range(attack_iters):
        with torch.enable_grad():
            _, seg_map = forward_fn(X_ir + d_ir, X_vis + d_vis)
            outputs = F.interpolate(seg_map, size=label.shape[1:], mode="bilinear", align_corners=False)
            loss = attack_loss_value(outputs, label, attack_way, i, attack_iters)
        loss.backward()  # accumulates into .grad
        if trace is not None:
            trace.append(dict(loss=float(loss.detach()), g_ir=d_ir.grad.detach().clone(),
                              g_vis=d_vis.grad.detach().clone()))
        with torch.no_grad():
            n_ir = torch.clamp(d_ir.data + alpha * torch.sign(d_ir.grad.data), min=-epsilon, max=epsilon)
            d_ir.data = torch.max(torch.min(n_ir, 1 - X_ir), 0 - X_ir)
            n_vis = torch.clamp(d_vis.data + alpha * torch.sign(d_vis.grad.data), min=-epsilon, max=epsilon)
            d_vis.data = torch.max(torch.min(n_vis, 1 - X_vis), 0 - X_vis)
    return d_ir.detach(), d_vis.detach()


# --------------------------------------------------------------------------------------
# metrics, losses, schedule
# --------------------------------------------------------------------------------------
def confusion_matrix(label, pred, n_class=9):
    """sklearn.metrics.confusion_matrix(labels=0..8) as used at test_original.py:209-211:
    rows = true class, cols = predicted; pairs with a class outside 0..n-1 are dropped."""
    label = np.asarray(label).reshape(-1).astype(np.int64)
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    ok = (label >= 0) & (label < n_class) & (pred >= 0) & (pred < n_class)
    return np.bincount(label[ok] * n_class + pred[ok], minlength=n_class * n_class).reshape(n_class, n_class)


def compute_results(conf_total):
    """util/util.py:31-55."""
    n = conf_total.shape[0]
    prec, rec, iou = np.zeros(n), np.zeros(n), np.zeros(n)
    for c in range(n):
        col, row, tp = conf_total[:, c].sum(), conf_total[c, :].sum(), conf_total[c, c]
        prec[c] = np.nan if col == 0 else float(tp) / float(col)
        rec[c] = np.nan if row == 0 else float(tp) / float(row)
        iou[c] = np.nan if (row + col - tp) == 0 else float(tp) / float(row + col - tp)
    return prec, rec, iou


def ssim(img1, img2, window_size=11):
    """pytorch_ssim/__init__.py:8-43,70-78 (gaussian sigma 1.5, zero padding, mean)."""
    ch = img1.shape[1]
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    win = g.mm(g.t()).float()[None, None].expand(ch, 1, window_size, window_size).contiguous().to(img1.dtype)   # built in fp32 like the reference
    pad = window_size // 2
    mu1, mu2 = F.conv2d(img1, win, padding=pad, groups=ch), F.conv2d(img2, win, padding=pad, groups=ch)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, win, padding=pad, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, win, padding=pad, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, win, padding=pad, groups=ch) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()


def fusionloss_grad2(image_ir, image_vis, generate_img, mask):
    """core/loss.py:490-502."""
    mask = mask[:, :1]
    return F.l1_loss(mask, generate_img) + 1.1 * (1 - ssim(generate_img, mask))


def model_losses(ir, vis, ir2, vis2, mask, labels, sd, backbone="mit_b0"):
    """Network_MM_Searched._loss / _loss_coupled / _fusion_loss_lower / _fusion_loss (core/model_fusion_auto.py:1093-1122)
    with Fusionloss_grad2 and CrossEntropyLoss(ignore_index=255)."""
    fused, seg = model_forward(ir, vis, sd, backbone)
    up = F.interpolate(seg, size=labels.shape[1:], mode="bilinear", align_corners=False)
    den = F.cross_entropy(up, labels.long(), ignore_index=255)
    enh = fusionloss_grad2(ir, rgb2ycrcb(vis), fused, mask)
    enh_c = fusionloss_grad2(ir2, rgb2ycrcb(vis2), fused, mask)     # the criterion ignores its image arguments
    return dict(loss=enh * 0.1 + den * 4, loss_coupled=enh_c * 0.1 + den * 4, fusion_loss_lower=enh, fusion_loss=enh)


def loss_coupled(ir_adv, vis_adv, mask, labels, sd, backbone="mit_b0"):
    """Network_MM_Searched._loss_coupled (core/model_fusion_auto.py:1102-1109): the ATTACKED pair goes through the model, the
    fusion criterion (which ignores its image arguments, core/loss.py:494-502) and the CE use the clean targets:
    0.1 * Fusionloss_grad2 + 4 * CrossEntropy(ignore 255)."""
    fused, seg = model_forward(ir_adv, vis_adv, sd, backbone)
    up = F.interpolate(seg, size=labels.shape[1:], mode="bilinear", align_corners=False)
    return fusionloss_grad2(None, None, fused, mask) * 0.1 + F.cross_entropy(up, labels.long(), ignore_index=255) * 4


def adamw_step(params, grads, m, v, t, lr, wd, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.AdamW, single-tensor form (what utils/optimizer.py:3-33 inherits), in place on lists of tensors;
    t = 1-based step count.  None gradients are skipped like torch does."""
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    for p, g, mm, vv in zip(params, grads, m, v):
        if g is None:
            continue
        p.mul_(1 - lr * wd)
        mm.lerp_(g, 1 - b1)
        vv.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (vv.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(mm, denom, value=-(lr / bc1))


def poly_warmup_lr_mult(step, warmup_iter, max_iter, warmup_ratio, power):
    """PolyWarmupAdamW.step's LR multiplier, utils/optimizer.py:17-28 (None = lr left unchanged)."""
    if step < warmup_iter:
        return 1 - (1 - step / warmup_iter) * (1 - warmup_ratio)
    if step < max_iter:
        return (1 - step / max_iter) ** power
    return None
