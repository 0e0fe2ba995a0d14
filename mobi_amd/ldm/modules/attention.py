"""SpatialTransformer / BasicTransformerBlock / CrossAttention / FeedForward on the engine.

Same class names, constructor kwargs and `state_dict` keys as the reference's
ldm/modules/attention.py (CrossAttention :153-194, BasicTransformerBlock :197-266,
SpatialTransformer :269-313, FeedForward/GEGLU :38-65); tokens are `[N, T, C]`
engine tensors, which is the same memory as the channels-last feature map, so the
reference's rearranges (:307,:310) disappear.

Launch sequence of one transformer block (all HIP, include/mobi_engine.h):
  attn1      layernorm -> igemm [to_q;to_k;to_v] (one stacked projection, V row-major) -> attention
             -> igemm to_out (+ residual, + attn2 vector)
  attn2      one key => softmax == 1: to_out(to_v(ref token)) is a per-image vector,
             two skinny_linear calls, added in attn1's epilogue (exact)
  adapter    two context tokens (the MObI case): ONE pass over the tokens, mobi_two_key_adapter -- LayerNorm
             statistics, 8 gate logits and the gated sum of 8 per-image vectors (see _two_key_terms); any other
             token count: layernorm -> igemm to_q -> ctx_attention -> igemm [connector . to_out folded] (+ residual)
  cross-modal (camera then lidar, in place on the interleaved batch)
             layernorm(x[::2]) -> igemm to_q ; igemm to_k / to_v(T) on x[1::2] -> attention
             -> igemm [connector . to_out folded] (+ residual, written back into x[::2]); then
             the lidar half against the UPDATED camera half (attention.py:257-261)
  ff         layernorm -> igemm GEGLU -> igemm (+ residual)
"""
import os
import weakref

import torch
import torch.nn as nn

from ... import engine_dtype, ops
from ..._lib import ACT_NONE
from .diffusionmodules.util import Conv2d, GroupNorm32, LayerNorm, Linear, Marker, enter, leave, zero_module


FUSED_FF = os.environ.get("MOBI_FUSED_FF", "1") != "0"      # A/B: 0 = GEGLU projection and output projection as two launches
FUSED_FF_MIN_ROWS = int(os.environ.get("MOBI_FUSED_FF_MIN_ROWS", "24576"))   # 192 workgroups
FUSED_LN = os.environ.get("MOBI_FUSED_LN", "1") != "0"      # A/B: 0 = the cross-modal LayerNorms as launches of their own
ROW_CHAIN = os.environ.get("MOBI_ROW_CHAIN", "1") != "0"    # A/B: 0 = the launches between the attention kernels one by one
LN_FOLD = os.environ.get("MOBI_LN_FOLD", "1") != "0"        # A/B: 0 = norm1 / norm3 as LayerNorm launches in front of their projections
LN_FOLD_MIN_ROWS = int(os.environ.get("MOBI_LN_FOLD_MIN_ROWS", "512"))    # below: a LayerNorm launch in front of the small kernels; 2048 -> 512: -0.03 ms per step of both workloads (profiles/r05_ab_ln_fold_rows.txt)
GROUPED_Q = os.environ.get("MOBI_GROUPED_Q", "1") != "0"    # A/B: 0 = the two cross-modal to_q projections as launches of their own
ROW_CHAIN_MIN_ROWS = int(os.environ.get("MOBI_ROW_CHAIN_MIN_ROWS", "24576"))   # 128 rows per workgroup: 192 workgroups
# GroupNorm, proj_in, norm1 and the q | k | v projection of a C = 320 block as ONE chain launch (+ a statistics pass): built, tested,
# and SLOWER than the four launches (19.90 against 19.61 ms per step, profiles/r04_ab_prechain.txt: four products at the chain
# kernel's 46 cycles per MFMA lose against the tuned igemm launches) -- off unless MOBI_PRE_CHAIN=1
PRE_CHAIN = os.environ.get("MOBI_PRE_CHAIN", "0") == "1"


def _store_in_place(old, new):
    """`new` (tensor | tuple of tensors | None) copied into `old`'s storage when the layouts agree, else `new`."""
    if isinstance(new, torch.Tensor):
        if isinstance(old, torch.Tensor) and old.shape == new.shape and old.dtype == new.dtype \
                and old.device == new.device and old.is_contiguous() and old.data_ptr() != new.data_ptr():
            old.copy_(new)
            return old
        return new.contiguous()
    if isinstance(new, tuple):
        if not (isinstance(old, tuple) and len(old) == len(new)):
            old = (None,) * len(new)
        return tuple(_store_in_place(o, n) for o, n in zip(old, new))
    return new


def Normalize(in_channels):
    return GroupNorm32(32, in_channels, eps=1e-6)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)

    def forward(self, x, ln=None):
        """ln: the LayerNorm in front of the projection, folded into the launch (ops.fold_layernorm: the packed matrix is
        W diag(gamma), the launch normalises with the row statistics of its own operand)."""
        if ln is None:
            return ops.linear(x, self.proj.packed_geglu())
        ps = (self.proj.weight, self.proj.bias, ln.weight, ln.bias)
        key = tuple(p._version for p in ps) + (ps[0].data_ptr(), ps[0].device, engine_dtype(), ln.eps)
        c = self.__dict__.setdefault("_ln_cache", {})
        if c.get("key") != key:
            w, b = ops.fold_layernorm(self.proj.weight, self.proj.bias, ln.weight, ln.bias)
            c["key"], c["val"] = key, ops.with_row_sums(ops.pack_geglu(w, b, engine_dtype(), self.proj.weight.device), ln.eps)
        return ops.linear(x, c["val"])


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        if not glu:
            raise NotImplementedError("MObI's transformer blocks use gated_ff=True")
        inner_dim = int(dim * mult)
        dim_out = dim if dim_out is None else dim_out
        self.net = nn.Sequential(GEGLU(dim, inner_dim), Marker(), Linear(inner_dim, dim_out))

    def _fused(self):
        """Chunk images of the pair for mobi_ff_geglu (C = 320: the output accumulators of a 32-row tile fit the registers),
        or None; cached per weight version and storage type like the other packs."""
        proj, out = self.net[0].proj, self.net[2]
        c, hidden = out.weight.shape
        if not (FUSED_FF and ops.ff_geglu_supported(c, hidden) and proj.weight.shape == (2 * hidden, c)):
            return None
        ps = (proj.weight, proj.bias, out.weight, out.bias)
        key = tuple(p._version for p in ps) + (ps[0].data_ptr(), ps[0].device, engine_dtype())
        cache = self.__dict__.setdefault("_fused_cache", {})
        if cache.get("key") != key:
            cache["key"], cache["val"] = key, ops.pack_ff_geglu(proj.weight, proj.bias, out.weight, out.bias, engine_dtype(),
                                                                proj.weight.device)
        return cache["val"]

    def forward(self, x, residual=None, norm=None):
        """norm: the LayerNorm module applied to x first (the transformer block's norm3); the one-launch kernel normalises
        the rows in the registers it multiplies them from, elsewhere it is a launch of its own."""
        # (128 token rows per workgroup: with fewer rows than FUSED_FF_MIN_ROWS the launch leaves most of the chip idle and the
        #  two launches win -- mobi_nusc_256, 8192 rows: 6.99 vs 7.22 ms per step)
        rows = x.numel() // x.shape[-1]
        pf = self._fused() if (rows >= FUSED_FF_MIN_ROWS and x.is_contiguous()
                               and (residual is None or residual.is_contiguous())) else None
        if pf is not None:                                       # one launch, the hidden activation never leaves the chip
            if norm is not None and FUSED_LN:
                return ops.ff_geglu(x, pf, residual=residual, ln=(*norm.affine(), norm.eps))
            if norm is not None:
                x = ops.layernorm(x, *norm.affine(), norm.eps)
            return ops.ff_geglu(x, pf, residual=residual)
        if norm is not None and LN_FOLD and x.is_contiguous() and rows >= LN_FOLD_MIN_ROWS:
            return ops.linear(self.net[0](x, ln=norm), self.net[2].packed(), residual=residual)
        if norm is not None:
            x = ops.layernorm(x, *norm.affine(), norm.eps)
        return ops.linear(self.net[0](x), self.net[2].packed(), residual=residual)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.inner_dim = inner_dim
        self.to_q = Linear(query_dim, inner_dim, bias=False)
        self.to_k = Linear(context_dim, inner_dim, bias=False)
        self.to_v = Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(Linear(inner_dim, query_dim), Marker())

    # -- packed projections ---------------------------------------------------------------
    def _stacked(self, names, fold_q=False, ln=None):
        """One packed matrix for several projections of the same input (rows stacked in `names` order): the input
        is read once and the launch count drops; attention reads q / k / v as column ranges of the result.
        fold_q: the to_q rows carry scale * log2(e) (fp32 masters multiplied before the single rounding to the storage
        type), so the attention kernels exponentiate q.k as a power of two without touching q again
        (mobi_attention_params.q_log2_scaled; `sim = einsum(q, k) * self.scale`, attention.py:178)."""
        mods = [getattr(self, n) for n in names]
        dtype = engine_dtype()
        key = (names, fold_q, dtype, self.scale) + tuple(v for m in mods for v in (m.weight._version, m.weight.data_ptr()))
        if ln is not None:       # the LayerNorm in front of the stacked projection, folded into the launch (ops.fold_layernorm)
            key += (ln.weight._version, ln.bias._version, ln.weight.data_ptr(), ln.eps)
        c = self.__dict__.setdefault("_stack_cache", {})
        slot = (names, fold_q, ln is not None)
        hit = c.get(slot)
        if hit is None or hit[0] != key:
            ws = [m.weight.detach() for m in mods]
            if fold_q:
                assert names[0] == "to_q"
                ws[0] = ws[0].float() * (self.scale * ops.LOG2E)
            w = torch.cat([w_.float() for w_ in ws], dim=0) if (fold_q or ln is not None) else torch.cat(ws, dim=0)
            if ln is not None:
                wf, bf = ops.fold_layernorm(w, None, ln.weight, ln.bias)
                hit = (key, ops.with_row_sums(ops.pack_linear(wf, bf, dtype, w.device), ln.eps))
            else:
                hit = (key, ops.pack_linear(w, None, dtype, w.device))
            c[slot] = hit
        return hit[1]

    # -- forms of attention ------------------------------------------------------------------
    def self_attention(self, xn, ln=None):
        """xn: normed tokens [N,T,C] -> attention output before to_out.  ln: xn is the RAW token tensor and this LayerNorm is
        folded into the stacked projection (no LayerNorm launch, no normalised copy)."""
        c = self.inner_dim
        qkv = ops.linear(xn, self._stacked(("to_q", "to_k", "to_v"), fold_q=True, ln=ln))
        return ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], self.heads, self.scale, v_rows=True,
                             q_log2_scaled=True)

    def token_attention(self, xn, ctx, q=None):
        """Many queries against another token stream `ctx` [N,Tk,Cc] (engine tensor, may be a
        batch-strided view).  q: to_q(xn) * scale * log2 e when the caller has it already (the grouped launch of
        BasicTransformerBlock._cross_modal_queries)."""
        c = self.inner_dim
        if q is not None:
            kv = ops.linear(ctx, self._stacked(("to_k", "to_v")))
        else:
            q, kv = ops.concurrently(lambda: ops.linear(xn, self._stacked(("to_q",), fold_q=True)),
                                     lambda: ops.linear(ctx, self._stacked(("to_k", "to_v"))))
        return ops.attention(q, kv[..., :c], kv[..., c:], self.heads, self.scale, v_rows=True, q_log2_scaled=True)

    def context_kv(self, context):
        """fp32 context [N,tk,Cc] -> (k, v) fp32 [N,tk,C]."""
        n, tk, cc = context.shape
        flat = context.reshape(n * tk, cc)
        wk, _ = self.to_k.skinny()
        wv, _ = self.to_v.skinny()
        k = ops.skinny_linear(flat, wk).reshape(n, tk, self.inner_dim)
        v = ops.skinny_linear(flat, wv).reshape(n, tk, self.inner_dim)
        return k, v

    def few_token_attention(self, xn, context):
        """Queries against <= 8 fp32 context tokens."""
        q, (k, v) = ops.concurrently(lambda: ops.linear(xn, self.to_q.packed()), lambda: self.context_kv(context))
        return ops.ctx_attention(q, k, v, self.heads, self.scale)

    def single_token_vector(self, token, extra_bias=None):
        """One key => softmax == 1 => the output is to_out(to_v(token)) for every query.
        token: fp32 [N, Cc] (row stride free) -> fp32 [N, query_dim].  extra_bias: a parameter vector added to
        to_out's bias (the bias of the launch this per-image vector is added in)."""
        wv, _ = self.to_v.skinny()
        wo, bo = self.to_out[0].skinny()
        if extra_bias is not None:
            bo = bo + extra_bias.detach().float()
        return ops.skinny_linear(ops.skinny_linear(token, wv), wo, bo)

    def attend(self, x, context=None, q=None):
        """Attention output BEFORE `to_out` (the caller applies to_out, possibly folded with a connector)."""
        if context is None:
            return self.self_attention(x)
        if context.dtype == torch.float32:
            if context.shape[1] > 8:
                raise NotImplementedError("fp32 contexts with more than 8 tokens")
            return self.few_token_attention(x, context.contiguous())
        return self.token_attention(x, context, q=q)

    def forward(self, x, context=None, mask=None):
        """Reference-compatible call (attention.py:171-194); x: engine tokens [N,T,C];
        context: None (self-attention), fp32 [N,tk<=8,Cc], or engine tokens."""
        if mask is not None:
            raise NotImplementedError("attention masks are not used on MObI's path")
        if context is None:
            a = self.self_attention(x)
        elif context.dtype == torch.float32:
            if context.shape[1] > 8:
                raise NotImplementedError("fp32 contexts with more than 8 tokens")
            a = self.few_token_attention(x, context.contiguous())
        else:
            a = self.token_attention(x, context)
        return ops.linear(a, self.to_out[0].packed())


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=False,
                 bbox_cond=False, multimodal=False):
        super().__init__()
        self.bbox_cond = bbox_cond
        self.multimodal = multimodal
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        if bbox_cond:
            self.cond_adapter_attn = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads,
                                                    dim_head=d_head, dropout=dropout)
        if multimodal:
            self.cross_modal_attn_camera = CrossAttention(query_dim=dim, context_dim=dim, heads=n_heads,
                                                          dim_head=d_head, dropout=dropout)
            self.cross_modal_attn_lidar = CrossAttention(query_dim=dim, context_dim=dim, heads=n_heads,
                                                         dim_head=d_head, dropout=dropout)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        self.norm3 = LayerNorm(dim)
        if bbox_cond:
            self.cond_adapter_norm = LayerNorm(dim)
            self.cond_adapter_connector = zero_module(Linear(dim, dim))
        if multimodal:
            self.cross_modal_norm_camera = LayerNorm(dim)
            self.cross_modal_connector_camera = zero_module(Linear(dim, dim))
            self.cross_modal_norm_lidar = LayerNorm(dim)
            self.cross_modal_connector_lidar = zero_module(Linear(dim, dim))
        self.checkpoint = checkpoint

    def _folded(self, attn, connector, tag):
        """`connector(attn.to_out(a))` is two Linear layers with nothing in between
        (attention.py:237-243, 249-261): W = Wc Wo, b = Wc bo + bc, folded once in fp32.  Exact algebra;
        saves one [T, C] x [C, C] GEMM and one activation round trip per adapter."""
        lo, lc = attn.to_out[0], connector
        key = (tag, lo.weight._version, lo.bias._version, lc.weight._version, lc.bias._version,
               lo.weight.data_ptr(), lo.packed().w.dtype)
        cache = self.__dict__.setdefault("_fold_cache", {})
        hit = cache.get(tag)
        if hit is None or hit[0] != key:
            wc, wo = lc.weight.detach().double().cpu(), lo.weight.detach().double().cpu()      # load-time, host
            w = (wc @ wo).float()
            b = (wc @ lo.bias.detach().double().cpu() + lc.bias.detach().double().cpu()).float()
            hit = (key, ops.pack_linear(w, b, lo.packed().w.dtype, lo.weight.device))
            cache[tag] = hit
        return hit[1]

    @staticmethod
    def _ln(norm, x):
        g, b = norm.affine()
        return ops.layernorm(x, g, b, norm.eps)

    def _cross_modal_queries(self, ln_cam, ln_lidar):
        """to_q of BOTH cross-modal attentions as ONE grouped launch (mobi_igemm_params.groups = 2) when the two normalised
        halves are the halves of one [camera images ; lidar images] buffer (ops.two_key_adapter's `ln_pair`): the lidar's
        queries do not depend on the camera update (attention.py:249-261: only its keys / values do), and one launch over
        all rows fills the chip where two over half the rows each left half of it idle and needed split-K slabs.
        -> (q_cam, q_lid) or (None, None)."""
        cam, lid = self.cross_modal_attn_camera, self.cross_modal_attn_lidar
        if not (GROUPED_Q and ln_cam is not None and ln_lidar is not None and ln_cam.shape == ln_lidar.shape
                and ln_cam.is_contiguous() and ln_lidar.is_contiguous() and cam.inner_dim == lid.inner_dim
                and ln_lidar.data_ptr() == ln_cam.data_ptr() + ln_cam.numel() * ln_cam.element_size()):
            return None, None
        dtype = engine_dtype()
        ws = (cam.to_q.weight, lid.to_q.weight)
        key = (dtype, cam.scale, lid.scale) + tuple(v for w in ws for v in (w._version, w.data_ptr()))
        c = self.__dict__.setdefault("_q_pair_cache", {})
        if c.get("key") != key:
            w = torch.cat([cam.to_q.weight.detach().float() * (cam.scale * ops.LOG2E),
                           lid.to_q.weight.detach().float() * (lid.scale * ops.LOG2E)], dim=0)
            c["key"], c["val"] = key, ops.pack_linear(w, None, dtype, w.device)
        h, t, ch = ln_cam.shape
        both = ln_cam.as_strided((2 * h, t, ch), (t * ch, ch, 1))            # [camera images ; lidar images]
        q = ops.linear(both, c["val"], groups=2)
        return q[:h], q[h:]

    # -- row-resident chains (C = 320: csrc/chain.hip) ---------------------------------------------------------------
    def _chain_weights(self):
        """Chunk images of every [C, C] matrix between the attention kernels (ops.pack_chain_weight), cached per weight
        version and storage type: attn1.to_out; the cross-modal to_q with its LayerNorm and scale * log2 e folded in; the
        cross-modal to_k / to_v; connector o to_out of both cross-modal attentions (W = Wc Wo, b = Wc bo + bc, fp64)."""
        a1, cam, lid = self.attn1, self.cross_modal_attn_camera, self.cross_modal_attn_lidar
        nc, nl = self.cross_modal_norm_camera, self.cross_modal_norm_lidar
        cc, cl = self.cross_modal_connector_camera, self.cross_modal_connector_lidar
        ps = [a1.to_out[0].weight, cam.to_q.weight, cam.to_k.weight, cam.to_v.weight, cam.to_out[0].weight, cam.to_out[0].bias,
              lid.to_q.weight, lid.to_k.weight, lid.to_v.weight, lid.to_out[0].weight, lid.to_out[0].bias,
              nc.weight, nc.bias, nl.weight, nl.bias, cc.weight, cc.bias, cl.weight, cl.bias]
        dtype = engine_dtype()
        key = tuple(p._version for p in ps) + (ps[0].data_ptr(), ps[0].device, dtype)
        c = self.__dict__.setdefault("_chain_cache", {})
        if c.get("key") != key:
            dev = ps[0].device
            pk = lambda w, b=None, **kw: ops.pack_chain_weight(w, b, dtype, dev, **kw)

            def fold(attn, con):
                wc, wo = con.weight.detach().double().cpu(), attn.to_out[0].weight.detach().double().cpu()
                return pk((wc @ wo).float(), (wc @ attn.to_out[0].bias.detach().double().cpu() + con.bias.detach().double().cpu()).float())

            c["key"] = key
            c["val"] = {"to_out": pk(a1.to_out[0].weight),            # (its bias rides in the per-image attn2 vector)
                        "q_cam": pk(cam.to_q.weight, ln=(nc.weight, nc.bias), scale=cam.scale * ops.LOG2E),
                        "q_lid": pk(lid.to_q.weight, ln=(nl.weight, nl.bias), scale=lid.scale * ops.LOG2E),
                        "k_cam": pk(cam.to_k.weight), "v_cam": pk(cam.to_v.weight),
                        "k_lid": pk(lid.to_k.weight), "v_lid": pk(lid.to_v.weight),
                        "fold_cam": fold(cam, cc), "fold_lid": fold(lid, cl)}
        return c["val"]

    def _chain_ok(self, x, adapter, adapter_image):
        return (ROW_CHAIN and self.bbox_cond and adapter is not None and adapter_image is not None and self.multimodal
                and x.shape[0] % 2 == 0
                and x.is_contiguous() and x.shape[0] * x.shape[1] >= ROW_CHAIN_MIN_ROWS
                and ops.row_chain_supported(x.shape[2], x.shape[1]))

    def _forward_chained(self, x, a, ref_vec, adapter_image):
        """Everything between attn1's attention kernel and the feed-forward with the token rows resident in registers:
          chain 1 (all rows)     x = to_out(a) + attn2 vector + x;  x = two-key adapter(x), written back;
                                 even (camera) images: q = to_q_cam(LN_cam(x));
                                 odd (lidar) images:   q = to_q_lid(LN_lid(x)),  k | v = to_k | to_v of the CAMERA's attention (x)
          attention (camera queries, lidar keys)
          chain 2 (camera rows)  x_cam = x_cam + connector(to_out(a));  k | v of the lidar's attention from the UPDATED rows
          attention (lidar queries, camera keys), connector o to_out (+ residual) on the lidar rows (mobi_igemm)
        attention.py:234-261 of the reference.  The unchained sequence is `_forward`'s (MOBI_ROW_CHAIN=0)."""
        n, t, c = x.shape
        cw = self._chain_weights()
        cam, lid = self.cross_modal_attn_camera, self.cross_modal_attn_lidar
        nc, nl = self.cross_modal_norm_camera, self.cross_modal_norm_lidar
        new = lambda *shape: torch.empty(shape, device=x.device, dtype=x.dtype)
        q_cam, q_lid, kv_l, kv_c = new(n // 2, t, c), new(n // 2, t, c), new(n // 2, t, 2 * c), new(n // 2, t, 2 * c)
        x_in, x = x, new(n, t, c)                  # (the caller's tensor stays as it was, as in the unchained sequence)

        def head(prog):
            return prog.load(a, "s").load(x_in, "r").product(cw["to_out"], resid=True, to_s=True, bias=ref_vec,
                                                             bias_img_stride=c).adapter(dst=x)
        p_cam = head(ops.ChainProgram()).rowstats(nc.eps).product(cw["q_cam"], fold=True, dst=q_cam, dst_img_div=2)
        p_lid = head(ops.ChainProgram()).rowstats(nl.eps).product(cw["q_lid"], fold=True, dst=q_lid, dst_img_div=2)
        p_lid.product(cw["k_cam"], dst=kv_l[..., :c], dst_img_div=2).product(cw["v_cam"], dst=kv_l[..., c:], dst_img_div=2)
        rows = n * t
        ops.row_chain([p_cam, p_lid], n, t, x.dtype, adapter=(adapter_image, self.cond_adapter_norm.eps),
                      flops=2.0 * rows * c * c * 3.5, nbytes=2.0 * rows * c * 5.0, note=f"post_attn1 rows={rows}")
        xc, xl = x[::2], x[1::2]
        ac = ops.attention(q_cam, kv_l[..., :c], kv_l[..., c:], cam.heads, cam.scale, v_rows=True, q_log2_scaled=True)
        p = ops.ChainProgram().load(ac, "s").load(xc, "r").product(cw["fold_cam"], resid=True, to_s=True, dst=xc)
        p.product(cw["k_lid"], dst=kv_c[..., :c]).product(cw["v_lid"], dst=kv_c[..., c:])
        ops.row_chain([p], n // 2, t, x.dtype, flops=2.0 * (rows // 2) * c * c * 3, nbytes=2.0 * (rows // 2) * c * 5.0,
                      note=f"post_cam rows={rows // 2}")
        al = ops.attention(q_lid, kv_c[..., :c], kv_c[..., c:], lid.heads, lid.scale, v_rows=True, q_log2_scaled=True)
        ops.linear(al, self._folded(lid, self.cross_modal_connector_lidar, "lidar"), residual=xl, out=xl)
        return self.ff(x, residual=x, norm=self.norm3)

    def forward(self, x, context=None, qkv=None):
        return self._forward(x, context, qkv=qkv)

    def _context_terms(self, ctx):
        """Everything that depends on the conditioning tokens only -- attn2's per-image vector and the bbox
        adapter's keys / values -- is the same at every denoising step of a sampling run.  It is computed once
        per context tensor (a slot per live tensor object, see below; the key checks the tensor's version counter and
        the weights' versions) instead of once per UNet call."""
        ws = [self.attn2.to_v.weight, self.attn2.to_out[0].weight, self.attn2.to_out[0].bias, self.attn1.to_out[0].bias]
        if self.bbox_cond:
            ca = self.cond_adapter_attn
            ws += [ca.to_k.weight, ca.to_v.weight, ca.to_q.weight, ca.to_out[0].weight, ca.to_out[0].bias,
                   self.cond_adapter_connector.weight, self.cond_adapter_connector.bias,
                   self.cond_adapter_norm.weight, self.cond_adapter_norm.bias]
        key = (ctx._version, ctx.data_ptr(), tuple(ctx.shape), tuple(w._version for w in ws),
               ws[0].data_ptr(), self.attn2.to_v.skinny()[0].dtype)
        # ONE SLOT PER CONTEXT TENSOR (keyed on the tensor object, dropped when it dies): a step graph captures the
        # addresses of its slot's buffers, so nothing else may write there -- a second graph (other batch, guidance or
        # sampler kind), the PLMS sampler or an eager call with other tokens get slots of their own.  (With one shared
        # slot a graph replayed after such a call read the other caller's terms, or freed memory.)
        slots = self.__dict__.setdefault("_ctx_slots", {})
        sid = id(ctx)
        c = slots.get(sid)
        if c is None or c["ref"]() is not ctx:
            c = {"ref": weakref.ref(ctx, lambda _r, s_=slots, i_=sid: s_.pop(i_, None)), "key": None}
            slots[sid] = c
        if c["key"] != key:
            c["key"] = key
            # (+ attn1.to_out's bias: the launch that adds this vector then passes no separate bias)
            ref_vec = self.attn2.single_token_vector(ctx[:, 0], extra_bias=self.attn1.to_out[0].bias)
            kv = self.cond_adapter_attn.context_kv(ctx) if self.bbox_cond else None
            adapter = self._two_key_terms(kv) if self.bbox_cond and ctx.shape[1] == 2 else None
            # the same tables as the LDS images the row-chain kernel copies in (C = 320 blocks: csrc/chain.hip)
            image = None
            # (the chain kernel's adapter tables hold up to 8 heads: a num_head_channels config with more keeps the one-by-one launches)
            if adapter is not None and ROW_CHAIN and self.cond_adapter_attn.heads <= 8 and ops.row_chain_supported(adapter[0].shape[2], 128):
                image = ops.chain_adapter_image(adapter[0], adapter[2], adapter[3], adapter[4], engine_dtype())
            # results live in PERSISTENT buffers, refreshed in place while their shapes stay the same (they do for one
            # context tensor): a denoising step captured in a HIP graph (mobi_amd/graph.py) keeps reading these addresses
            for name, val in (("ref_vec", ref_vec), ("kv", kv), ("adapter", adapter), ("adapter_image", image)):
                c[name] = _store_in_place(c.get(name), val)
        return c["ref_vec"], c["kv"], c["adapter"], c["adapter_image"]

    def context_term_addresses(self, ctx):
        """Device addresses of the slot `ctx` owns (None if it has none): what a captured step reads."""
        c = self.__dict__.get("_ctx_slots", {}).get(id(ctx))
        if c is None or c["ref"]() is not ctx or c["key"] is None:
            return None
        flat = []
        for name in ("ref_vec", "kv", "adapter", "adapter_image"):
            v = c.get(name)
            for t in (v if isinstance(v, tuple) else (v,)):
                flat.append(None if t is None else t.data_ptr())
        return tuple(flat)

    def _two_key_params(self):
        """The parameter-only factors of the two-token fold below, prepared once per weight version on the host in fp64
        (load-time work like the weight packs) and kept as fp32 device tensors:
        W = connector o to_out [query, inner], b0 = Wc bo + bc, to_q^T [query, inner], gamma * scale, beta * scale, and
        the head mask [H, inner]."""
        ca, ln, con = self.cond_adapter_attn, self.cond_adapter_norm, self.cond_adapter_connector
        ps = [ca.to_q.weight, ca.to_out[0].weight, ca.to_out[0].bias, con.weight, con.bias, ln.weight, ln.bias]
        key = tuple(p._version for p in ps) + (ps[0].data_ptr(), ps[0].device)
        c = self.__dict__.setdefault("_two_key_cache", {})
        if c.get("key") != key:
            d = lambda t: t.detach().double().cpu()
            wq, wo, bo, wc, bc, gamma, beta = (d(p) for p in ps)
            h, inner = ca.heads, wq.shape[0]
            mask = torch.zeros(h, inner, dtype=torch.float64)
            for i in range(h):
                mask[i, i * (inner // h):(i + 1) * (inner // h)] = 1.0
            f = lambda t: t.float().contiguous().to(ps[0].device)
            c["key"], c["val"] = key, (f(wc @ wo), f(wc @ bo + bc), f(wq.t()), f(gamma * ca.scale), f(beta * ca.scale), f(mask))
        return c["val"]

    def _two_key_terms(self, kv):
        """The bbox adapter (attention.py:237-243 of the reference: `x + connector(attn(norm(x), context))`) against
        exactly TWO context tokens, folded into per-image vectors -- exact algebra:
          softmax over two keys = sigmoid of the score difference:  p0 = sigmoid(scale * q_h . (k0 - k1)_h)
          q = to_q(LN(x)) has no bias, so  q_h . dk_h = LN(x) . (Wq_h^T dk_h) = rstd * (x . a_h - mean * sum a_h) + c_h
          attention output = v1 + p0 * (v0 - v1), pushed through W = connector o to_out:  b + sum_h p0_h * u_h
        so no [T, C] x [C, C] product is left: `ops.two_key_adapter` makes one pass over the tokens.
        The token-dependent products run once per sampling run as fp32 FMA chains on the engine (`mobi_linear_f32`; round 2
        had fp64 torch products here, i.e. rocBLAS kernels inside the product path): head h's slice of the key / value
        difference, zero elsewhere, times to_q^T resp. W.
        Returns (a [N,H,C], a_sum [N,H], c [N,H], u [N,H,C], b [N,C]) fp32."""
        w, b0, wqt, gamma_s, beta_s, mask = self._two_key_params()
        k, v = kv[0].float(), kv[1].float()                                    # [N, 2, inner]
        n, _, inner = k.shape
        h = self.cond_adapter_attn.heads
        dk = ((k[:, 0] - k[:, 1])[:, None, :] * mask[None]).reshape(n * h, inner).contiguous()
        dv = ((v[:, 0] - v[:, 1])[:, None, :] * mask[None]).reshape(n * h, inner).contiguous()
        wt = ops.linear_f32(dk, wqt).view(n, h, -1)                            # Wq_h^T dk_h: [N, H, query]
        a = (wt * gamma_s).contiguous()
        c = (wt * beta_s).sum(-1).contiguous()
        u = ops.linear_f32(dv, w).view(n, h, -1).contiguous()                  # W_h dv_h: [N, H, query]
        b = ops.linear_f32(v[:, 1].contiguous(), w, b0)
        return a, a.sum(-1).contiguous(), c, u, b

    def _forward(self, x, context=None, qkv=None):
        """x: engine tokens [N,T,C]; context: fp32 [N, n_ctx, context_dim]; qkv: attn1's stacked projections of norm1(x)
        [N, T, 3C] when the caller has them already (SpatialTransformer's pre-attention chain; q carries scale * log2 e)."""
        ctx = context.float().contiguous()
        ref_vec, ctx_kv, adapter, adapter_image = self._context_terms(ctx)
        # attn1 (self) + attn2 (reference token; norm2 / to_q cancel out of a one-key softmax)
        if qkv is not None:
            c = self.attn1.inner_dim
            a = ops.attention(qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], self.attn1.heads, self.attn1.scale, v_rows=True,
                              q_log2_scaled=True)
        elif LN_FOLD and x.is_contiguous() and x.shape[0] * x.shape[1] >= LN_FOLD_MIN_ROWS:
            a = self.attn1.self_attention(x, ln=self.norm1)
        else:
            a = self.attn1.self_attention(self._ln(self.norm1, x))
        if self._chain_ok(x, adapter, adapter_image):
            return self._forward_chained(x, a, ref_vec, adapter_image)
        x = ops.linear(a, self.attn1.to_out[0].packed(), residual=x, rowvec=ref_vec, rowvec_has_bias=True)

        ln_cam = ln_lidar = None
        if self.bbox_cond and adapter is not None:
            nc, nl = (self.cross_modal_norm_camera, self.cross_modal_norm_lidar) if self.multimodal else (None, None)
            if (FUSED_LN and self.multimodal and x.shape[0] % 2 == 0 and nc.eps == nl.eps
                    and ops.two_key_adapter_fuses_ln(x.shape[2], x.shape[0] * x.shape[1])):
                # the adapter kernel holds each result row in registers: it also writes the two LayerNorms the cross-modal
                # step reads next (camera norm of the even images, lidar norm of the odd ones -- the camera update below
                # does not touch the lidar rows)
                x, (ln_cam, ln_lidar) = ops.two_key_adapter(x, *adapter, eps=self.cond_adapter_norm.eps, out=x,
                                                            ln_pair=(nc.affine(), nl.affine(), nc.eps))
            else:
                x = ops.two_key_adapter(x, *adapter, eps=self.cond_adapter_norm.eps, out=x)
        elif self.bbox_cond:
            ca = self.cond_adapter_attn
            q = ops.linear(self._ln(self.cond_adapter_norm, x), ca.to_q.packed())
            a = ops.ctx_attention(q, ctx_kv[0], ctx_kv[1], ca.heads, ca.scale)
            x = ops.linear(a, self._folded(self.cond_adapter_attn, self.cond_adapter_connector, "adapter"),
                           residual=x)

        if self.multimodal:
            if x.shape[0] % 2:
                raise ValueError("multimodal blocks need camera/lidar samples interleaved on an even batch")
            xc, xl = x[::2], x[1::2]
            q_cam, q_lid = self._cross_modal_queries(ln_cam, ln_lidar)
            a = self.cross_modal_attn_camera.attend(self._ln(self.cross_modal_norm_camera, xc) if ln_cam is None else ln_cam,
                                                    context=xl, q=q_cam)
            ops.linear(a, self._folded(self.cross_modal_attn_camera, self.cross_modal_connector_camera, "cam"),
                       residual=xc, out=xc)
            a = self.cross_modal_attn_lidar.attend(self._ln(self.cross_modal_norm_lidar, xl) if ln_lidar is None else ln_lidar,
                                                   context=xc, q=q_lid)
            ops.linear(a, self._folded(self.cross_modal_attn_lidar, self.cross_modal_connector_lidar, "lidar"),
                       residual=xl, out=xl)

        return self.ff(x, residual=x, norm=self.norm3)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, bbox_cond=False,
                 multimodal=False):
        super().__init__()
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.proj_in = Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim,
                                   bbox_cond=bbox_cond, multimodal=multimodal) for _ in range(depth)])
        self.proj_out = zero_module(Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))

    def _pre_weights(self):
        """Chunk images for the pre-attention chain: proj_in, and attn1's to_q / to_k / to_v of the first block with its norm1
        (and, for q, scale * log2 e) folded in."""
        blk = self.transformer_blocks[0]
        a1, n1 = blk.attn1, blk.norm1
        ps = [self.proj_in.weight, self.proj_in.bias, a1.to_q.weight, a1.to_k.weight, a1.to_v.weight, n1.weight, n1.bias]
        dtype = engine_dtype()
        key = tuple(p._version for p in ps) + (ps[0].data_ptr(), ps[0].device, dtype)
        c = self.__dict__.setdefault("_pre_cache", {})
        if c.get("key") != key:
            dev = ps[0].device
            pk = lambda w, b=None, **kw: ops.pack_chain_weight(w, b, dtype, dev, **kw)
            ln = (n1.weight, n1.bias)
            c["key"] = key
            c["val"] = {"proj_in": pk(self.proj_in.weight[:, :, 0, 0], self.proj_in.bias),
                        "q": pk(a1.to_q.weight, ln=ln, scale=a1.scale * ops.LOG2E), "k": pk(a1.to_k.weight, ln=ln),
                        "v": pk(a1.to_v.weight, ln=ln)}
        return c["val"]

    def _pre_chain_ok(self, x):
        n, h, w, c = x.shape
        return (ROW_CHAIN and PRE_CHAIN and len(self.transformer_blocks) == 1 and c == self.proj_in.weight.shape[0] and x.is_contiguous()
                and n * h * w >= ROW_CHAIN_MIN_ROWS and ops.row_chain_supported(c, h * w) and c % 64 == 0)

    def forward(self, x, context=None):
        x, ext = enter(x)
        n, h, w, c = x.shape
        g, b = self.norm.affine()
        if isinstance(x, ops.Deferred) and self._pre_chain_ok(x.tensor):
            x = x.finish()
        if not isinstance(x, ops.Deferred) and self._pre_chain_ok(x):
            # GroupNorm -> proj_in -> norm1 -> to_q | to_k | to_v as ONE chain launch on the token rows (+ a statistics pass that
            # turns the GroupNorm into a per-image scale / shift): attention.py:306-307 and :234 of the reference
            blk = self.transformer_blocks[0]
            cw = self._pre_weights()
            scale, shift = ops.groupnorm_scale_shift(x, g, b, self.norm.eps)
            t = torch.empty((n, h * w, c), device=x.device, dtype=x.dtype)
            qkv = torch.empty((n, h * w, 3 * c), device=x.device, dtype=x.dtype)
            prog = ops.ChainProgram().load(x.view(n, h * w, c), "s").affine(scale, shift).product(cw["proj_in"], to_s=True, dst=t)
            prog.rowstats(blk.norm1.eps).product(cw["q"], fold=True, dst=qkv[..., :c]).product(cw["k"], fold=True, dst=qkv[..., c:2 * c])
            prog.product(cw["v"], fold=True, dst=qkv[..., 2 * c:])
            rows = n * h * w
            ops.row_chain([prog], n, h * w, x.dtype, flops=2.0 * rows * c * c * 4, nbytes=2.0 * rows * c * 5.0, note=f"pre_attn1 rows={rows}")
            t = blk(t, context=context, qkv=qkv)
            y = ops.igemm(t.view(n, h, w, t.shape[2]), self.proj_out.packed(), residual=x)
            return leave(y, ext)
        # (a Deferred x -- the partial sums of the ResBlock's last convolution -- is summed by this GroupNorm, its first reader)
        xn = ops.groupnorm(x, g, b, self.norm.eps, silu=False)
        x = ops.finished(x)
        t = ops.igemm(xn, self.proj_in.packed())
        t = t.view(n, h * w, t.shape[3])
        for block in self.transformer_blocks:
            t = block(t, context=context)
        y = ops.igemm(t.view(n, h, w, t.shape[2]), self.proj_out.packed(), residual=x)
        return leave(y, ext)
