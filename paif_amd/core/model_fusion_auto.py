"""Fusion network, NAS cell containers and the composite fusion+segmentation models --
MI355X-native counterparts of the reference's core/model_fusion_auto.py classes that are on the hot
path (F1-F8, G1, S1).  Same names, constructor signatures and state_dict keys;
forward() bodies launch the hand-written gfx950 kernels of libpaif_hip.so.

Not reproduced (out of scope, never constructed by either entry script): DRDB, Fusion_Network*,
SKFF, the GAN pieces, Network_MM_TARDAL/UMF/Base/Auto, the *_showfeatures variants.
"""
import torch
import torch.nn as nn

from .. import ops
from ..operations_m import OPS, BasicConv, Conv2dParams, PReLUParams, _HipOp, _PackCache, grad_anchor


# ---------------------------------------------------------------------------------------------
# colour transforms (reference: core/model_fusion_auto.py:69-111)
# ---------------------------------------------------------------------------------------------
def RGB2YCrCb(input_im):
    """core/model_fusion_auto.py:69-92.  [B,3,H,W] -> [B,3,H,W] (Y, Cr, Cb)."""
    ops.require_no_grad(input_im)
    return ops.rgb2ycrcb(input_im)


def YCrCb2RGB(input_im):
    """core/model_fusion_auto.py:94-111.  [B,3,H,W] (Y, Cr, Cb) -> [B,3,H,W] RGB, not clamped.  (Inside the
    fusion->seg glue the same arithmetic runs fused with the clamp and the min/max pass: ops.seg_input_from_fused.)"""
    ops.require_no_grad(input_im)
    return ops.ycrcb2rgb(input_im)


class MixedOp(nn.Module):
    """core/model_fusion_auto.py:397-415.  'Name_k' iff the name contains lowercase 'attention', else
    'Name_k_d' (IndexError on a missing field, KeyError on an unknown name -- as in the reference)."""

    def __init__(self, C, primitive):
        super().__init__()
        self._ops = nn.ModuleList()
        kernel = 3
        dilation = 1
        if primitive.find('attention') != -1:
            name = primitive.split('_')[0]
            kernel = int(primitive.split('_')[1])
        else:
            name = primitive.split('_')[0]
            kernel = int(primitive.split('_')[1])
            dilation = int(primitive.split('_')[2])
        self._op = OPS[name](C, kernel, dilation, False)

    def forward(self, x):
        return self._op(x)

    def forward_nhwc(self, x, res=(), tape=None, out_f32=False, cpool=None):
        """out_f32 / cpool: requests of the inference forward to the LAST op of a chain -- an fp32 output map from fp16 sources; the
        ChannelPool of the output written into cpool = (comp, offset).  Honoured inside the op where its last conv can (one launch
        less), by a cast / a stand-alone pooling pass here otherwise."""
        if out_f32 and getattr(self._op, "takes_out_f32", False):
            return self._op.forward_nhwc(x, res, tape, out_f32=True)
        if cpool is not None and tape is None:
            if getattr(self._op, "takes_cpool", False):
                return self._op.forward_nhwc(x, res, tape, cpool=cpool)
            y = self._op.forward_nhwc(x, res, tape)
            ops.channel_pool1(y, *cpool)
            return y
        return self._op.forward_nhwc(x, res, tape)

    def backward_nhwc(self, g, t, wgrad=False, res=()):
        if res:
            return self._op.backward_nhwc(g, t, wgrad, res=res)
        return self._op.backward_nhwc(g, t, wgrad)

    @property
    def bwd_takes_res(self):
        return getattr(self._op, "bwd_takes_res", False)


class Cell_Chain(_HipOp):
    """core/model_fusion_auto.py:418-445: inp + ops(inp).  The outer residual (and any residuals the
    caller adds on top) ride in the last op's conv epilogue."""

    _tape_is_list = True

    def __init__(self, C, type, concat):
        super().__init__()
        op_names, indices = zip(*type)
        self._compile(C, op_names, indices, concat)

    def _compile(self, C, op_names, indices, concat):
        assert len(op_names) == len(indices)
        self._steps = len(op_names)
        self._concat = concat
        self.multiplier = len(concat)
        self._ops = nn.ModuleList()
        for name, index in zip(op_names, indices):
            self._ops += [MixedOp(C, name)]
        self._indices = indices

    def forward_nhwc(self, inp, res=(), tape=None, out_f32=False, cpool=None):
        """out_f32 (fp16-storage inference forward): ask the last op for an fp32 output map (honoured where its kernel exists);
        cpool = (comp, offset): the ChannelPool of the chain's output goes into comp[..., offset:offset+2] (MixedOp.forward_nhwc)."""
        s1 = inp
        for i in range(self._steps):
            last = i == self._steps - 1
            s1 = self._ops[i].forward_nhwc(s1, ((inp,) + tuple(res)) if last else (), tape, out_f32=out_f32 and last,
                                           cpool=cpool if last else None)
        return s1

    def backward_nhwc(self, g, tape, wgrad=False):
        """g = d/d(chain output) -> d/d(inp).  `tape`: this chain's entries in forward order (one per op)."""
        d = g
        for i in range(self._steps - 1, 0, -1):
            d = self._ops[i].backward_nhwc(d, tape[i], wgrad)
        # the chain's own residual (out = inp + ops(inp)): d/d(inp) = g + ops^T(g).  The first op's last dgrad conv adds g in its epilogue
        # when the op can (RDB, ECA block: every chain of the shipped genotype); an elementwise pass otherwise
        if self._ops[0].bwd_takes_res:
            return self._ops[0].backward_nhwc(d, tape[0], wgrad, res=(g,))
        return ops.add(self._ops[0].backward_nhwc(d, tape[0], wgrad), g)


class Cell_Decom(nn.Module):
    """core/model_fusion_auto.py:492-535.  Guided-filter decomposition (r=4, eps 1e-3 / 1e-4) of both
    streams, the two 128->32 1x1 convs (folded to 96->32 over [x, LF1, LF2]: HF = x - LF is never
    materialised), the two chains and the outer residuals."""

    def __init__(self, C, types, concat):
        super().__init__()
        self._C = C
        self.radiux = [4]
        self.eps_list = [0.001, 0.0001]
        self._ops_1 = nn.ModuleList()
        self._ops_2 = nn.ModuleList()
        self.conv1x1_lf = Conv2dParams(C * 4, C, kernel_size=1, bias=True)
        self.conv1x1_hf = Conv2dParams(C * 4, C, kernel_size=1, bias=True)
        self._steps = len(concat)
        self.relu = PReLUParams()  # unused in the reference too, but serialised (decompation.relu.weight)
        self.chain = Cell_Chain(C, types[0], concat)
        self.chain2 = Cell_Chain(C, types[1], concat)
        self._packs = _PackCache()
        if C != 32:
            raise NotImplementedError("the HIP kernels are built for C = 32 (the only width either entry script uses)")

    def get_residue(self, tensor):
        return ops.to_nchw_view(ops.channel_residue(ops.to_nhwc(tensor)).unsqueeze(-1))

    def decomposition_nhwc(self, x, guide=None, want_ab=False, out_bf16=False):
        """-> lf [2,B,H,W,32] (LF for eps 1e-3, 1e-4)."""
        if guide is None:
            guide = ops.channel_residue(x)
        return ops.guided_filter_pair(guide, x, tuple(self.eps_list), want_ab=want_ab, out_bf16=out_bf16)

    @staticmethod
    def _cat_lf_hf(x, lf):
        """-> (LF, HF) as [B,2C,H,W] views: eps 1e-3 then 1e-4 on the channel axis (the reference's list order, :533-534)."""
        lf_cat, hf_cat = ops.decomp_cat(x, lf)
        return ops.to_nchw_view(lf_cat), ops.to_nchw_view(hf_cat)

    def decomposition(self, x, C=None):
        """core/model_fusion_auto.py:522-535: (LF [B,2C,H,W], HF [B,2C,H,W]).  (forward() never materialises HF: it is folded
        into the 1x1 conv; this public form does, because returning it is the point.)"""
        ops.require_no_grad(x)
        xn = ops.to_nhwc(x)
        return self._cat_lf_hf(xn, self.decomposition_nhwc(xn))

    def forward_nhwc(self, fir, fvis, g_ir=None, g_vis=None, tape=None, feats=None, comp=None):
        """feats (dict): receives the decomposition intermediates (LF maps, residues) for the visualisation path.
        comp ([B,H,W,4] fp32, inference): receives ChannelPool(ir_feature, vis_feature) (:1352-1355) from the two chains' last convs."""
        if g_ir is None:
            g_ir = ops.channel_residue(fir)
        if g_vis is None:
            g_vis = ops.channel_residue(fvis)
        # 16-bit configurations (inference): the guided filter writes its two maps as bf16 (fp16: the two HIGH-frequency maps x - LF) and
        # the folded 1x1 takes them with the stem map's 16-bit twin -- everything behind the filter's fp32 statistics / A / b is a 16-bit map
        lf16 = tape is None and feats is None and ops._ACT_BF16[0]
        if tape is None:
            lf_ir = self.decomposition_nhwc(fir, g_ir, out_bf16=lf16)
            lf_vis = self.decomposition_nhwc(fvis, g_vis, out_bf16=lf16)
        else:
            lf_ir, ab_ir = self.decomposition_nhwc(fir, g_ir, want_ab=True)
            lf_vis, ab_vis = self.decomposition_nhwc(fvis, g_vis, want_ab=True)
        # fp16 configuration: the filter wrote HF = x - LF (fp16) and the 1x1 folds over [x, HF1, HF2]
        pack1, pn = (ops.pack_decomp1x1_hf_weight, "_hf16") if lf16 is torch.float16 else (ops.pack_decomp1x1_weight, "")
        w_lf = self._packs.get("lf" + pn, [self.conv1x1_lf.weight], lambda: pack1(self.conv1x1_lf.weight))
        w_hf = self._packs.get("hf" + pn, [self.conv1x1_hf.weight], lambda: pack1(self.conv1x1_hf.weight))
        if feats is not None:
            feats.update(lf_ir=lf_ir, lf_vis=lf_vis, g_ir=g_ir, g_vis=g_vis)
        x_ir, x_vis = (ops.cast_storage(fir, True), ops.cast_storage(fvis, True)) if lf16 else (fir, fvis)   # the stems' bf16 twins
        lf = ops.conv2d([x_ir, lf_ir[0], lf_ir[1]], w_lf, 1, 1, shift=self.conv1x1_lf.bias)
        hf = ops.conv2d([x_vis, lf_vis[0], lf_vis[1]], w_hf, 1, 1, shift=self.conv1x1_hf.bias)
        t1 = None if tape is None else []
        t2 = None if tape is None else []
        cp = comp if tape is None else None
        ir_feature = self.chain.forward_nhwc(lf, (fir,), t1, cpool=None if cp is None else (cp, 0))      # lf_re + inp_ir
        vis_feature = self.chain2.forward_nhwc(hf, (fvis,), t2, cpool=None if cp is None else (cp, 2))   # hf_re + inp_vis
        if tape is not None:
            tape.update(fir=fir, fvis=fvis, g_ir=g_ir, g_vis=g_vis, ab_ir=ab_ir, ab_vis=ab_vis, chain=t1, chain2=t2)
            if ops.taping_wgrad():
                tape.update(lf_ir=lf_ir, lf_vis=lf_vis)
        return ir_feature, vis_feature

    def branch_nhwc(self, f, g, which, comp=None):
        """One stream of forward_nhwc (inference, no tape): guided-filter decomposition of the stem map f with guide g, the folded 1x1,
        the stream's chain (+ f).  which: 0 = infrared (conv1x1_lf, chain), 1 = visible (conv1x1_hf, chain2).  The two streams are
        independent up to the spatial blend: Network_Fusion_Searched runs them on two HIP streams (ops.CONFIG["two_stream"])."""
        lf16 = ops._ACT_BF16[0]
        lf = self.decomposition_nhwc(f, g, out_bf16=lf16)
        conv = self.conv1x1_lf if which == 0 else self.conv1x1_hf
        pack1, pn = (ops.pack_decomp1x1_hf_weight, "_hf16") if lf16 is torch.float16 else (ops.pack_decomp1x1_weight, "")
        w = self._packs.get(("lf" if which == 0 else "hf") + pn, [conv.weight], lambda: pack1(conv.weight))
        x = ops.cast_storage(f, True) if lf16 else f
        y = ops.conv2d([x, lf[0], lf[1]], w, 1, 1, shift=conv.bias)
        return (self.chain if which == 0 else self.chain2).forward_nhwc(y, (f,), None, cpool=None if comp is None else (comp, 2 * which))

    def _stream_backward(self, d_feat, chain, chain_tape, conv, name, feat, guide, ab, lf=None):
        """One stream: d/d(ir_feature) -> d/d(stem feature).  lf (the stream's two LF maps): also the 1x1's parameter gradients."""
        wgrad = lf is not None
        d_l = chain.backward_nhwc(d_feat, chain_tape, wgrad)                 # through chain(lf) (+ its residual)
        if wgrad:
            gb, gw = ops.grad_of(conv.bias), ops.grad_of(conv.weight)
            if gb is not None:
                ops.colsum(d_l, gb)
            if gw is not None:   # gradient of the folded 96-channel weight over [x, LF1, LF2], unfolded to cat[LF1, LF2, x-LF1, x-LF2]
                ops.unfold_decomp1x1_wgrad(ops.conv2d_wgrad([feat, lf[0], lf[1]], d_l, 1), gw)
        wf = self._packs.get("fold_" + name, [conv.weight], lambda: ops.fold_decomp1x1_weight(conv.weight))
        dg = lambda s: self._packs.get("dg_%s%d" % (name, s), [conv.weight], lambda: ops.pack_conv_dgrad_weight(wf, 32 * s, 32))
        d_x = ops.conv2d([d_l], dg(0), 1, 1, res=(d_feat,))                  # x term of the folded 1x1 + the outer "+ inp"
        dlf = torch.empty((2,) + tuple(feat.shape), device=feat.device, dtype=torch.float32)
        ops.conv2d([d_l], dg(1), 1, 1, out=dlf[0])
        ops.conv2d([d_l], dg(2), 1, 1, out=dlf[1])
        return ops.guided_filter_bwd(guide, feat, ab, dlf, tuple(self.eps_list), add=d_x)

    def backward_nhwc(self, d_ir_feature, d_vis_feature, t, wgrad=False):
        d_fir = self._stream_backward(d_ir_feature, self.chain, t["chain"], self.conv1x1_lf, "lf", t["fir"], t["g_ir"], t["ab_ir"],
                                      t["lf_ir"] if wgrad else None)
        d_fvis = self._stream_backward(d_vis_feature, self.chain2, t["chain2"], self.conv1x1_hf, "hf", t["fvis"], t["g_vis"], t["ab_vis"],
                                       t["lf_vis"] if wgrad else None)
        return d_fir, d_fvis

    def forward(self, inp_ir, inp_vis):
        ops.require_no_grad(inp_ir, inp_vis)
        with torch.no_grad():
            a, b = self.forward_nhwc(ops.to_nhwc(inp_ir), ops.to_nhwc(inp_vis))
        return ops.to_nchw_view(a), ops.to_nchw_view(b)


class ChannelPool(nn.Module):
    """core/model_fusion_auto.py:1352-1355 (2-argument form)."""

    def forward(self, ir, vis):
        ops.require_no_grad(ir, vis)
        return ops.to_nchw_view(ops.channel_pool2(ops.to_nhwc(ir), ops.to_nhwc(vis)))


class Cell_Decom_decom(Cell_Decom):
    """core/model_fusion_auto.py:536-581 -- the decomposition cell of the feature-visualisation network: same
    parameters and arithmetic as Cell_Decom, but it also RETURNS the decomposition (LF and HF of both eps concatenated
    on the channel axis, and the residue guide).  HF = x - LF is materialised here because showing it is the point."""

    def decomposition(self, x, C=None):
        """(LF [B,2C,H,W], HF [B,2C,H,W], res [B,1,H,W]) -- :567-581."""
        ops.require_no_grad(x)
        xn = ops.to_nhwc(x)
        g = ops.channel_residue(xn)
        lf, hf = self._cat_lf_hf(xn, self.decomposition_nhwc(xn, g))
        return lf, hf, ops.to_nchw_view(g.unsqueeze(-1))

    def forward_feats_nhwc(self, fir, fvis, g_ir=None, g_vis=None):
        feats = {}
        ir_feature, vis_feature = self.forward_nhwc(fir, fvis, g_ir, g_vis, None, feats)
        lf_ir, hf_ir = self._cat_lf_hf(fir, feats["lf_ir"])
        lf_vis, hf_vis = self._cat_lf_hf(fvis, feats["lf_vis"])
        res_ir = ops.to_nchw_view(feats["g_ir"].unsqueeze(-1))
        res_vis = ops.to_nchw_view(feats["g_vis"].unsqueeze(-1))
        return ir_feature, vis_feature, lf_ir, hf_ir, res_ir, lf_vis, hf_vis, res_vis

    def forward(self, inp_ir, inp_vis):
        ops.require_no_grad(inp_ir, inp_vis)
        r = self.forward_feats_nhwc(ops.to_nhwc(inp_ir), ops.to_nhwc(inp_vis))
        return (ops.to_nchw_view(r[0]), ops.to_nchw_view(r[1])) + r[2:]


class spatial_attn_layer_M(nn.Module):
    """core/model_fusion_auto.py:1358-1368: sigmoid(conv5x5 4->1 (ChannelPool(ir, vis)))."""

    def __init__(self, kernel_size=5):
        super().__init__()
        if kernel_size != 5:
            raise NotImplementedError("spatial_attn_layer_M: only the 5x5 kernel of the reference is built")
        self.compress = ChannelPool()
        self.spatial = BasicConv(4, 1, kernel_size, relu=False)

    def blend_nhwc(self, ir, vis, want_scale=False, want_comp=False, comp=None):
        """comp: ChannelPool(ir, vis) already written by the producers of ir / vis (paif_conv_desc.cpool); None = pooled here."""
        if comp is None:
            comp = ops.channel_pool2(ir, vis)
        r = ops.spa_blend(comp, self.spatial.conv.weight, ir, vis, want_scale)
        return (r + (comp,)) if want_comp else r

    def blend_backward_nhwc(self, d_agg, ir, vis, scale, comp=None):
        """comp (the pooled 4-channel map of the forward): also accumulate the 5x5 conv's weight gradient."""
        w = self.spatial.conv.weight
        if comp is None:
            return ops.spa_blend_bwd(d_agg, w, ir, vis, scale)
        d_ir, d_vis, dpre = ops.spa_blend_bwd(d_agg, w, ir, vis, scale, want_dpre=True)
        gw = ops.grad_of(w)
        if gw is not None:
            ops.corr1_wgrad(dpre, comp, 5, gw)
        return d_ir, d_vis

    def forward(self, ir, vis):
        ops.require_no_grad(ir, vis)
        with torch.no_grad():
            _, scale = self.blend_nhwc(ops.to_nhwc(ir), ops.to_nhwc(vis), want_scale=True)
        return scale.unsqueeze(1)


class Network_Fusion_Searched(nn.Module):
    """core/model_fusion_auto.py:599-640."""

    def __init__(self, C, criterion, genotype_feature, steps=4, multiplier=3):
        super().__init__()
        self._C = C
        self._criterion = criterion
        self._steps = steps
        self._multiplier = multiplier
        self._genotype = genotype_feature
        self.stem_1 = nn.Sequential(Conv2dParams(1, C, 3, padding=1, bias=False), PReLUParams())
        self.stem_2 = nn.Sequential(Conv2dParams(1, C, 3, padding=1, bias=False), PReLUParams())
        self.stem_out = nn.Sequential(
            Conv2dParams(C, C // 2, 3, padding=1, bias=False),
            Conv2dParams(C // 2, 1, 3, padding=1, bias=False),
            PReLUParams(),
        )
        self.tanh = nn.Tanh()
        self.spa = spatial_attn_layer_M()
        self.decompation = self._decom_cls(C, [self._genotype.normal_1, self._genotype.normal_2], self._genotype.normal_1_concat)
        self.chain = Cell_Chain(C, self._genotype.normal_3, self._genotype.normal_1_concat)
        self._packs = _PackCache()

    _decom_cls = Cell_Decom

    def forward(self, ir, vis, inter=None):
        wg = ops.want_param_grads(self)
        if torch.is_grad_enabled() and (ir.requires_grad or vis.requires_grad or wg):
            return _FusionFn.apply(ir, vis, self, wg, grad_anchor(ir.device))
        # ops.set_storage("bf16"): bf16 maps behind the guided-filter block (eval mode: the train-mode BatchNorm path is fp32 only)
        # (a forward that returns the decomposition intermediates -- inter["want_decomposition"] -- keeps fp32 maps: the guided filter
        # then writes fp32 LF maps, which the 16-bit chain kernels do not take)
        want_feats = inter is not None and bool(inter.get("want_decomposition"))
        with torch.no_grad(), ops.bf16_activations(enable=not self.training and not want_feats):
            return self.forward_impl(ir, vis, inter=inter)

    def forward_impl(self, ir, vis, inter=None, tape=None):
        """ir, vis: [B,>=1,H,W] (channel 0 is used, :626-627).  tape (dict): filled for the input-gradient pass."""
        vis = vis[:, 0:1, :, :]
        ir = ir[:, 0:1, :, :]
        if tape is None and inter is None and ir.is_cuda and ops.CONFIG.get("two_stream", False):
            # the infrared and the visible stream (stem -> guided filter -> 1x1 -> chain) are independent up to the blend: the visible one
            # runs on a second HIP stream, so that its kernels fill the CUs the other stream's launches leave idle (a guided-filter launch
            # holds 224 of the 256 CUs with one workgroup each; every kernel's last round of workgroups leaves a tail)
            main = torch.cuda.current_stream()
            side = self.__dict__.get("_side_stream")
            if side is None or side.device != ir.device:
                side = self.__dict__["_side_stream"] = torch.cuda.Stream(device=ir.device)
            comp = self._new_comp(ir)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fvis, g_vis = ops.stem(vis, self.stem_2[0].weight, self.stem_2[1].weight)
                vis_feature = self.decompation.branch_nhwc(fvis, g_vis, 1, comp)
            fir, g_ir = ops.stem(ir, self.stem_1[0].weight, self.stem_1[1].weight)
            ir_feature = self.decompation.branch_nhwc(fir, g_ir, 0, comp)
            main.wait_stream(side)
            # vis_feature lives in the side stream's allocator pool and is consumed here.  Its block can only be handed out again to a
            # side-stream allocation, i.e. after the next forward's side.wait_stream(main) -- behind everything this stream does with it;
            # record_stream states that explicitly outside a graph capture (it is not permitted on a capture's private pool)
            if not torch.cuda.is_current_stream_capturing():
                vis_feature.record_stream(main)
            del fvis, g_vis
            agg = self.spa.blend_nhwc(ir_feature, vis_feature, comp=comp)
            return self._tail_nhwc(self.chain.forward_nhwc(agg, (), None, out_f32=self._last_f32()))
        fir, g_ir = ops.stem(ir, self.stem_1[0].weight, self.stem_1[1].weight)
        fvis, g_vis = ops.stem(vis, self.stem_2[0].weight, self.stem_2[1].weight)
        t_dec = None if tape is None else {}
        feats = {} if (inter is not None and inter.get("want_decomposition")) else None
        comp = self._new_comp(ir) if tape is None else None
        ir_feature, vis_feature = self.decompation.forward_nhwc(fir, fvis, g_ir, g_vis, t_dec, feats, comp=comp)
        if feats is not None:
            inter.update(feats)
        t_chain = None if tape is None else []
        if tape is None:
            agg = self.spa.blend_nhwc(ir_feature, vis_feature, comp=comp)
        elif ops.taping_wgrad():
            agg, scale, comp = self.spa.blend_nhwc(ir_feature, vis_feature, want_scale=True, want_comp=True)
        else:
            agg, scale = self.spa.blend_nhwc(ir_feature, vis_feature, want_scale=True)
        # fp16 storage: the forward's last 32-channel map (largest magnitudes, straight into the fused image) stays fp32
        feature2 = self.chain.forward_nhwc(agg, (), t_chain, out_f32=tape is None and self._last_f32())
        if tape is None:
            out = self._tail_nhwc(feature2)
        else:
            w0 = self._packs.get("so0", [self.stem_out[0].weight], lambda: ops.pack_conv_weight(self.stem_out[0].weight, 1, 32, 3))
            t16 = ops.conv2d([feature2], w0, 3, 1, cout=16)
            out, z = ops.tail(t16, self.stem_out[1].weight, self.stem_out[2].weight, save=True)
            tape.update(dec=t_dec, chain=t_chain, ir_feature=ir_feature, vis_feature=vis_feature, scale=scale, fused=out, z=z)
            if ops.taping_wgrad():
                tape.update(comp=comp, feature2=feature2, t16=t16, img_ir=ir, img_vis=vis)
        if inter is not None:
            inter.update(fir=fir, fvis=fvis, ir_feature=ir_feature, vis_feature=vis_feature, agg=agg, feature2=feature2)
        return out

    @staticmethod
    def _new_comp(img):
        """The [B,H,W,4] plane ChannelPool(ir_feature, vis_feature) is written into by the two chains' last convs (inference forward)."""
        if not ops.CONFIG.get("cpool_fused", True):
            return None
        B, _, H, W = img.shape
        return torch.empty((B, H, W, 4), device=img.device, dtype=torch.float32)

    @staticmethod
    def _last_f32():
        return ops._ACT_BF16[0] is torch.float16 and ops.CONFIG.get("f16_last_f32", True)      # (ablation switch)

    def _tail_nhwc(self, feature2):
        """stem_out + tanh of the inference forward (no tape)."""
        so = self.stem_out
        if feature2.dtype == torch.float16:     # (a genotype whose last op has no fp32-output kernel)
            feature2 = ops.cast_storage(feature2, torch.float32)
        if feature2.dtype == torch.float32:
            # fp32 maps: the one-kernel form takes x as bf16 hi + lo (the split-bf16 convs' own operand precision); under
            # set_conv_precision("f32" | "bf16x6") the two packed convs run at the requested precision instead
            fused_on = ops.CONFIG.get("stem_out_fused_f32", True) and ops.CONFIG["conv_precision"] == "bf16x3"
        else:
            fused_on = ops.CONFIG.get("stem_out_fused", True)
        if feature2.shape[1] >= 3 and feature2.shape[2] >= 3 and fused_on:
            # inference forward (bf16 maps; since round 4 fp32 maps too): both stem_out convs + PReLU + tanh as one launch pair (csrc/stem_out.hip)
            wso = self._packs.get("so_fused", [so[0].weight, so[1].weight], lambda: ops.stem_out_pack(so[0].weight, so[1].weight))
            return ops.stem_out_fused(feature2, wso, so[2].weight)
        if ops._ACT_BF16[0] is torch.float16:  # the fp16 range guard rides on the one-kernel form; on this two-launch path a torch reduction feeds it
            ops.f16_guard_flag(feature2.device).bitwise_or_((~torch.isfinite(feature2).all()).to(torch.int32))
        if feature2.dtype == torch.float32:   # (also the fp32 last map of an fp16-storage forward: a split-bf16 pack whatever the forward's packs are)
            prec = ops.CONFIG["conv_precision"]
            w0 = self._packs.get("so0_" + prec, [so[0].weight], lambda: ops.pack_conv_weight(so[0].weight, 1, 32, 3, precision=prec))
        else:
            w0 = self._packs.get("so0", [so[0].weight], lambda: ops.pack_conv_weight(so[0].weight, 1, 32, 3))
        return ops.tail(ops.conv2d([feature2], w0, 3, 1, cout=16), so[1].weight, so[2].weight)

    def backward_impl(self, d_fused, tape, wgrad=False):
        """d/d(fused) [B,1,H,W] -> (d/d(ir), d/d(vis channel 0)) as [B,1,H,W] each; wgrad: parameter gradients too."""
        from ..operations_m import check_positive_slope
        for prm, what in ((self.stem_1[1].weight, "stem_1"), (self.stem_2[1].weight, "stem_2")):
            check_positive_slope(prm, what)
        so = self.stem_out
        if wgrad:
            dz = ops.tail_dz(d_fused, tape["fused"], tape["z"], so[2].weight, ops.grad_of(so[2].weight))
            gw = ops.grad_of(so[1].weight)
            if gw is not None:
                ops.corr1_wgrad(dz, tape["t16"], 3, gw)
        d_t16 = ops.tail_bwd(d_fused, tape["fused"], tape["z"], so[1].weight, so[2].weight)
        if wgrad:
            gw = ops.grad_of(so[0].weight)
            if gw is not None:   # 32 -> 16 conv: the 16-channel gradient zero padded, first 16 rows of dW
                ops.conv2d_wgrad([tape["feature2"]], ops.pad_channels(d_t16, 32), 3, 1, out=gw, cout=16)
        w0t = self._packs.get("so0T", [so[0].weight], lambda: ops.pack_conv_dgrad_weight(so[0].weight, 0, 32))
        d_feat2 = ops.conv2d([d_t16], w0t, 3, 1, cin=16, cout=32)
        d_agg = self.chain.backward_nhwc(d_feat2, tape["chain"], wgrad)
        d_irf, d_visf = self.spa.blend_backward_nhwc(d_agg, tape["ir_feature"], tape["vis_feature"], tape["scale"],
                                                     tape["comp"] if wgrad else None)
        d_fir, d_fvis = self.decompation.backward_nhwc(d_irf, d_visf, tape["dec"], wgrad)
        t = tape["dec"]
        if wgrad:
            for stem, img, d_f in ((self.stem_1, tape["img_ir"], d_fir), (self.stem_2, tape["img_vis"], d_fvis)):
                gw, gs = ops.grad_of(stem[0].weight), ops.grad_of(stem[1].weight)
                if gw is not None or gs is not None:
                    ops.stem_wgrad(img, d_f, stem[0].weight, stem[1].weight, gw, gs)
        d_ir = ops.stem_bwd(d_fir, t["fir"], self.stem_1[0].weight, self.stem_1[1].weight)
        d_y = ops.stem_bwd(d_fvis, t["fvis"], self.stem_2[0].weight, self.stem_2[1].weight)
        if wgrad:
            ops.grads_ready(self)
        return d_ir, d_y

    def _loss(self, ir, vis, mask):
        logits = self(ir, vis)
        return self._criterion(ir, vis, logits, mask)


class Network_Fusion_Searched_showfeatures(Network_Fusion_Searched):
    """core/model_fusion_auto.py:643-695 -- the feature-visualisation variant (caller: `val_fusion_train`,
    test_original.py:548-662).  Same parameters (state_dict keys identical), same fused output; `forward2` also returns
    the decomposition intermediates."""

    _decom_cls = Cell_Decom_decom

    def forward2(self, ir, vis):
        """-> (fused, ir_feature, vis_feature, lf_ir, hf_ir, res_ir, lf_vis, hf_vis, res_vis)   (:669-679)"""
        ops.require_no_grad(ir, vis)
        with torch.no_grad():
            inter = {"want_decomposition": True}
            out = self.forward_impl(ir, vis, inter=inter)
            fir, fvis = inter["fir"], inter["fvis"]
            lf_ir, hf_ir = Cell_Decom_decom._cat_lf_hf(fir, inter["lf_ir"])
            lf_vis, hf_vis = Cell_Decom_decom._cat_lf_hf(fvis, inter["lf_vis"])
            res_ir = ops.to_nchw_view(inter["g_ir"].unsqueeze(-1))
            res_vis = ops.to_nchw_view(inter["g_vis"].unsqueeze(-1))
            return (out, ops.to_nchw_view(inter["ir_feature"]), ops.to_nchw_view(inter["vis_feature"]), lf_ir, hf_ir, res_ir,
                    lf_vis, hf_vis, res_vis)


class _FusionFn(torch.autograd.Function):
    """Autograd node of the fusion network: hand-written reverse pass; with `wgrad` the parameter gradients are accumulated
    into param.grad by the wgrad kernels (the reference's loss.backward() does the same through torch autograd)."""

    @staticmethod
    def forward(ctx, ir, vis, module, wgrad, anchor):
        tape = {}
        with ops.tape_mode("wgrad" if wgrad else "dgrad"):
            out = module.forward_impl(ir.detach(), vis.detach(), tape=tape)
        ctx.tape, ctx.module, ctx.wgrad = tape, module, wgrad
        ctx.shapes = (tuple(ir.shape), tuple(vis.shape))
        return out

    @staticmethod
    def backward(ctx, d_fused):
        d_ir1, d_y1 = ctx.module.backward_impl(d_fused.contiguous(), ctx.tape, ctx.wgrad)
        ctx.tape = None
        outs = []
        for g1, shp in ((d_ir1, ctx.shapes[0]), (d_y1, ctx.shapes[1])):
            if shp[1] == 1:
                outs.append(g1)
            else:  # forward used channel 0 only: zero gradient for the others
                full = torch.zeros(shp, device=g1.device, dtype=g1.dtype)
                full[:, 0:1] = g1
                outs.append(full)
        return outs[0], outs[1], None, None, None


# ---------------------------------------------------------------------------------------------
# segmentation network wrapper + composite models
# ---------------------------------------------------------------------------------------------
from . import mix_transformer  # noqa: E402
from .segformer_head import SegFormerHead  # noqa: E402


class WeTr(nn.Module):
    """core/model_fusion_auto.py:9-68: MiT encoder + SegFormer head (+ a dead `classifier`, :66, whose result
    the reference discards: it is kept as a parameter for state_dict parity and never evaluated)."""

    def __init__(self, backbone, num_classes=20, embedding_dim=256, pretrained=None):
        super().__init__()
        self.num_classes = num_classes
        self.embedding_dim = embedding_dim
        self.backbone = backbone
        self.feature_strides = [4, 8, 16, 32]
        self.encoder = getattr(mix_transformer, backbone)()
        self.in_channels = self.encoder.embed_dims
        if pretrained:
            self.initialize()
        self.decoder = SegFormerHead(feature_strides=self.feature_strides, in_channels=self.in_channels,
                                     embedding_dim=self.embedding_dim, num_classes=self.num_classes)
        self.classifier = Conv2dParams(in_channels=self.in_channels[-1], out_channels=self.num_classes, kernel_size=1, bias=False)

    def initialize(self):
        """core/model_fusion_auto.py:22-26,32-36: load 'pretrained/<backbone>.pth' minus the ImageNet head."""
        state_dict = torch.load('pretrained/' + self.backbone + '.pth')
        state_dict.pop('head.weight')
        state_dict.pop('head.bias')
        self.encoder.load_state_dict(state_dict)

    def get_param_groups(self):
        """core/model_fusion_auto.py:44-60."""
        param_groups = [[], [], []]
        for name, param in list(self.encoder.named_parameters()):
            if "norm" in name:
                param_groups[1].append(param)
            else:
                param_groups[0].append(param)
        for param in list(self.decoder.parameters()):
            param_groups[2].append(param)
        param_groups[2].append(self.classifier.weight)
        return param_groups

    def forward_nhwc(self, x, tape=None):
        """x NHWC [B,H,W,3] (normalised) -> logits NHWC [B,H/4,W/4,num_classes]."""
        enc_tape = None if tape is None else []
        head_tape = None if tape is None else {}
        if tape is None:       # inference: the GEMMs / attention on fp16 pairs (ops.inference_gemm_arithmetic)
            with ops.inference_gemm_arithmetic():
                return self.decoder.forward_nhwc(self.encoder.forward_features_nhwc(x, None), None)
        out = self.decoder.forward_nhwc(self.encoder.forward_features_nhwc(x, enc_tape), head_tape)
        if tape is not None:
            tape.update(enc=enc_tape, head=head_tape)
        return out

    def backward_nhwc(self, d_logits32, tape, wgrad=False):
        """d/d(logits) as NHWC [B,H/4,W/4,32] (zero padded) -> d/d(input) NHWC [B,H,W,3]; wgrad: parameter gradients too
        (every parameter except `classifier.weight`, whose output the forward discards -- it never receives a gradient in
        the reference either)."""
        d_feats = self.decoder.backward_nhwc(d_logits32, tape["head"], wgrad)
        return self.encoder.backward_features_nhwc(d_feats, tape["enc"], wgrad)

    def forward(self, x):
        wg = ops.want_param_grads(self)
        if torch.is_grad_enabled() and (x.requires_grad or wg):
            return _WeTrFn.apply(x, self, wg, grad_anchor(x.device))
        with torch.no_grad():
            return ops.nhwc_to_nchw(self.forward_nhwc(ops.nchw_to_nhwc(x)))


class _WeTrFn(torch.autograd.Function):
    """Autograd node of WeTr: hand-written reverse pass; with `wgrad` the parameter gradients are accumulated into param.grad."""

    @staticmethod
    def forward(ctx, x, module, wgrad, anchor):
        tape = {}
        with ops.tape_mode("wgrad" if wgrad else "dgrad"):
            out = ops.nhwc_to_nchw(module.forward_nhwc(ops.nchw_to_nhwc(x.detach()), tape))
        ctx.tape, ctx.module, ctx.wgrad = tape, module, wgrad
        return out

    @staticmethod
    def backward(ctx, d_out):
        d32 = ops.nchw_to_nhwc_pad(d_out, 32)
        d_x = ctx.module.backward_nhwc(d32, ctx.tape, ctx.wgrad)
        ctx.tape = None
        return ops.nhwc_to_nchw(d_x), None, None, None


class _CompositeBase(nn.Module):
    """Shared body of Network_MM_CompModel (:698-806) and Network_MM_Searched (:1029-1137)."""

    mean = [123.675, 116.28, 103.53]
    std = [58.395, 57.12, 57.375]

    def _setup(self, f_loss, segloss):
        self.fusion_nums = 2
        self.seg_nums = 2
        self.fusion_channel = 48
        self.seg_channel = 64
        self._criterion = f_loss
        self.seg_loss = segloss
        # optional: True = the glue's min/max is taken over ALL ranks' batches (one 2-float all-reduce),
        # reproducing one process that runs the whole batch; default False = per rank, what DDP around the reference does
        self.global_minmax = False

    def _minmax_sync(self):
        if not self.global_minmax:
            return None
        if torch.is_grad_enabled():
            raise NotImplementedError("global_minmax=True is an inference mode (the gradient of a min/max element living on "
                                      "another rank is not exchanged)")
        import torch.distributed as dist
        from ..dist_utils import global_minmax
        return lambda mn, mx: global_minmax(mn, mx, dist)

    def forward(self, ir, vis):
        """-> (fused [B,1,H,W] in tanh range, seg_map [B,num_classes,H/4,W/4]); :712-729 / :1043-1060."""
        wg = ops.want_param_grads(self)
        if torch.is_grad_enabled() and (ir.requires_grad or vis.requires_grad or wg):
            return _CompositeFn.apply(ir, vis, self, wg, grad_anchor(ir.device))
        with torch.no_grad():
            ycc = ops.rgb2ycrcb(vis)
            with ops.bf16_activations(enable=not self.enhance_net.training):     # ops.set_storage("bf16"), inference only
                fused = self.enhance_net.forward_impl(ir[:, 0:1, :, :], ycc)
            seg_in = ops.seg_input_from_fused(fused, ycc, minmax_sync=self._minmax_sync())   # clamp, BATCH-GLOBAL min-max, x255, mean/std
            seg_map = self.denoise_net(seg_in)
        return fused, seg_map

    # ---- taped forward / hand-written reverse pass (input gradients only) -------------------------
    def forward_taped(self, ir, vis, object_glue=False):
        """-> (fused [B,1,H,W], logits NHWC [B,H/4,W/4,ncls], tape).  object_glue: forward_object's form (:743-751) -- the fused
        plane is clamped to [0,1] and min-max normalised (batch-global) before the recomposition, and THAT plane is returned."""
        tape = dict(fus={}, seg={}, ir_shape=tuple(ir.shape))
        ycc = ops.rgb2ycrcb(vis)
        fused = self.enhance_net.forward_impl(ir[:, 0:1, :, :], ycc, tape=tape["fus"])
        if object_glue:
            tape["fused_raw"] = fused
            fused, tape["plane_minmax"] = ops.plane_clamp_minmax(fused)
        seg_in, mm = ops.seg_input_from_fused(fused, ycc, return_minmax=True)
        logits = self.denoise_net.forward_nhwc(ops.nchw_to_nhwc(seg_in), tape["seg"])
        tape.update(ycc=ycc, fused=fused, minmax=mm)
        return fused, logits, tape

    def backward_taped(self, d_logits32, tape, d_fused=None, wgrad=False):
        """d/d(logits) NHWC [B,H/4,W/4,32] (zero padded) [+ d/d(fused)] -> (d/d(ir) [B,Cir,H,W], d/d(vis) [B,3,H,W]);
        wgrad: parameter gradients of both networks too (the tape must come from forward_taped under tape_mode("wgrad"))."""
        d_seg_in = ops.nhwc_to_nchw(self.denoise_net.backward_nhwc(d_logits32, tape["seg"], wgrad))
        d_f, dcrcb = ops.glue_bwd(d_seg_in, tape["fused"], tape["ycc"], tape["minmax"], d_fused)
        if "fused_raw" in tape:                                   # forward_object: back through the plane's clamp + min-max
            d_f = ops.plane_clamp_minmax_bwd(d_f, tape["fused_raw"], tape["plane_minmax"])
        d_ir1, d_y = self.enhance_net.backward_impl(d_f, tape["fus"], wgrad)
        d_vis = ops.rgb2ycrcb_bwd(d_y, dcrcb)
        shp = tape["ir_shape"]
        if shp[1] != 1:   # forward used channel 0 only
            full = torch.zeros(shp, device=d_ir1.device, dtype=d_ir1.dtype)
            full[:, 0:1] = d_ir1
            d_ir1 = full
        return d_ir1, d_vis

    def forward_fusion(self, ir, vis):
        ops.require_no_grad(ir, vis)
        with torch.no_grad():
            return self.enhance_net.forward(ir[:, 0:1, :, :], ops.rgb2ycrcb(vis))

    def forward_object(self, ir, vis):
        """:736-766 / :1067-1097 -- the segmentation objective on a fused plane that is first clamped to [0,1] and min-max
        normalised over the batch (the name is historical: no detection head is involved) -> (that plane, seg_map)."""
        wg = ops.want_param_grads(self)
        if torch.is_grad_enabled() and (ir.requires_grad or vis.requires_grad or wg):
            return _CompositeFn.apply(ir, vis, self, wg, grad_anchor(ir.device), True)
        if self.global_minmax:
            # the plane's own min-max (ops.plane_clamp_minmax) is a per-rank kernel pair: with global_minmax the two min-max steps of this
            # form would use different scopes and no longer reproduce the single-process batch
            raise NotImplementedError("forward_object: global_minmax=True is built for forward() only")
        with torch.no_grad():
            ycc = ops.rgb2ycrcb(vis)
            with ops.bf16_activations(enable=not self.enhance_net.training):
                fused = self.enhance_net.forward_impl(ir[:, 0:1, :, :], ycc)
            fused, _ = ops.plane_clamp_minmax(fused)
            seg_map = self.denoise_net(ops.seg_input_from_fused(fused, ycc))
        return fused, seg_map

    # ---- training-API losses (:1093-1128): values, input gradients and parameter gradients (the models' autograd nodes +
    # the loss gradient kernels); `loss.backward()` fills param.grad like in the reference.
    def _seg_term(self, seg_map, labels):
        """self.seg_loss(F.interpolate(seg_map, size=labels.shape[1:], bilinear), labels.long())   (:1096-1098)"""
        labels = labels.type(torch.long).contiguous()
        sl = self.seg_loss
        if isinstance(sl, nn.CrossEntropyLoss) and sl.reduction == "mean" and sl.weight is None and sl.label_smoothing == 0.0:
            # fused bilinear upsample + cross entropy, forward and gradient (HIP; ops.UpsampleCE is the autograd node)
            return ops.upsample_ce(seg_map, labels, ignore_index=sl.ignore_index)
        # any other user criterion: the reference's own two calls on torch ops above the HIP model
        import torch.nn.functional as F
        return sl(F.interpolate(seg_map, size=labels.shape[1:], mode='bilinear', align_corners=False), labels)

    @staticmethod
    def _ycc(vis):
        return ops.rgb2ycrcb(vis.detach())   # the criterion ignores its image arguments (core/loss.py:494-502)

    def _loss(self, ir, vis, mask, labels):
        fused_img, seg_map = self(ir, vis)
        enhance_loss = self._criterion(ir, self._ycc(vis), fused_img, mask)
        return enhance_loss * 0.1 + self._seg_term(seg_map, labels) * 4

    def _loss_coupled(self, ir_, vis_, mask, labels):
        fused_img, seg_map = self(ir_[0], vis_[0])
        denoise_loss = self._seg_term(seg_map, labels)
        enhance_loss = self._criterion(ir_[1], self._ycc(vis_[1]), fused_img, mask)
        return enhance_loss * 0.1 + denoise_loss * 4

    def _fusion_loss_lower(self, ir, vis, mask):
        fused_img, _ = self(ir, vis)
        return self._criterion(ir, self._ycc(vis), fused_img, mask)

    def _fusion_loss(self, ir, vis, mask):
        fused_img = self.forward_fusion(ir, vis)
        return self._criterion(ir, self._ycc(vis), fused_img, mask)

    _fusion_loss_wogan = _fusion_loss

    def _detection_loss(self, ir, vis, labels):
        """:796-800 / :1123-1128: seg_loss on forward_object's map."""
        _, seg_map = self.forward_object(ir, vis)
        return self._seg_term(seg_map, labels)

    def enhance_net_parameters(self):
        return self.enhance_net.parameters()

    def denoise_net_parameters(self):
        return self.denoise_net.parameters()


class _CompositeFn(torch.autograd.Function):
    """Autograd node of the composite model: lets the reference's own training / attack code (`loss.backward()`) drive the
    HIP reverse pass.  With `wgrad` (autograd recording outside an attack, parameters requiring grad) the parameter
    gradients of both networks are accumulated straight into param.grad by the wgrad kernels."""

    @staticmethod
    def forward(ctx, ir, vis, module, wgrad, anchor, object_glue=False):
        with ops.tape_mode("wgrad" if wgrad else "dgrad"):
            fused, logits, tape = module.forward_taped(ir.detach(), vis.detach(), object_glue)
        ctx.tape, ctx.module, ctx.wgrad = tape, module, wgrad
        return fused, ops.nhwc_to_nchw(logits)

    @staticmethod
    def backward(ctx, d_fused, d_seg):
        d32 = ops.nchw_to_nhwc_pad(d_seg, 32)
        if ctx.wgrad:      # the training step: weight-gradient GEMMs on fp16 pairs, dY scaled by ~ the pixel count inside the kernels
            with ops.wgrad_scale(d_fused.numel()):
                d_ir, d_vis = ctx.module.backward_taped(d32, ctx.tape, d_fused.contiguous(), ctx.wgrad)
        else:
            d_ir, d_vis = ctx.module.backward_taped(d32, ctx.tape, d_fused.contiguous(), ctx.wgrad)
        ctx.tape = None
        return d_ir, d_vis, None, None, None, None


def grad_milestones(model):
    """The sub-modules of a composite model (or a bare WeTr / fusion net) in the order the reverse pass FINISHES their
    parameter gradients -- each calls ops.grads_ready(module) at that point: SegFormer head, then per encoder stage
    (last to first) the stage norm, its blocks (last to first) and its patch embedding, then the fusion network."""
    out = []
    seg = getattr(model, "denoise_net", None) or (model if isinstance(model, WeTr) else None)
    if seg is not None:
        out.append(seg.decoder)
        for s in (4, 3, 2, 1):
            out.append(getattr(seg.encoder, "norm%d" % s))
            out.extend(reversed(list(getattr(seg.encoder, "block%d" % s))))
            out.append(getattr(seg.encoder, "patch_embed%d" % s))
    fus = getattr(model, "enhance_net", None) or (model if isinstance(model, Network_Fusion_Searched) else None)
    if fus is not None:
        out.append(fus)
    return out


def assign_grad_order(model):
    """Tag every parameter with `_paif_order` = its position in the backward-completion order (the optimizer lays the
    arenas out in that order, so contiguous gradient buckets become ready one after the other), and the parameters that
    never receive a gradient -- WeTr.classifier.weight (its output is discarded, :66) and Cell_Decom.relu.weight (unused,
    :505) -- with `_paif_never_grad` (kept at the arena's tail, outside every all-reduce bucket)."""
    order = 0
    for m in grad_milestones(model):
        for p in reversed(list(m.parameters())):
            if not hasattr(p, "_paif_order"):
                p._paif_order = order
                order += 1
    for name, p in model.named_parameters():
        if name.endswith("classifier.weight") or name.endswith("decompation.relu.weight"):
            p._paif_never_grad = True
    return model


class Network_MM_CompModel(_CompositeBase):
    """core/model_fusion_auto.py:698-806."""

    def __init__(self, model, f_loss, segloss, backbone, num_classes=20, embedding_dim=256, pretrained=None):
        super().__init__()
        self._setup(f_loss, segloss)
        self.enhance_net = model
        self.denoise_net = WeTr(backbone, num_classes, embedding_dim, pretrained)
        assign_grad_order(self)


class Network_MM_Searched(_CompositeBase):
    """core/model_fusion_auto.py:1029-1137."""

    def __init__(self, C, genotype, f_loss, segloss, backbone, num_classes=20, embedding_dim=256, pretrained=None):
        super().__init__()
        self._setup(f_loss, segloss)
        self.enhance_net = Network_Fusion_Searched(C, f_loss, genotype)
        self.denoise_net = WeTr(backbone, num_classes, embedding_dim, pretrained)
        assign_grad_order(self)


class Network_MM_SearchedFusion(nn.Module):
    """core/model_fusion_auto.py:1141-1188 (fusion only)."""

    def __init__(self, C, genotype, f_loss):
        super().__init__()
        self.fusion_nums = 2
        self.seg_nums = 2
        self.fusion_channel = 48
        self.seg_channel = 64
        self._criterion = f_loss
        self.enhance_net = Network_Fusion_Searched(C, f_loss, genotype)
        self.denoise_net = None

    def forward(self, ir, vis):
        ops.require_no_grad(ir, vis)
        with torch.no_grad():
            return self.enhance_net.forward(ir, ops.rgb2ycrcb(vis))

    forward_fusion = forward

    def enhance_net_parameters(self):
        return self.enhance_net.parameters()
