"""First slice of the training step: forward WITH a tape and the backward pass of one
`BasicTransformerBlock` on the engine -- every trainable tensor of the reference's optimizer filter that lives in a block
(`cond_adapter*`, `cross_modal*`: ldm/models/diffusion/ddpm.py:1616-1629 of the reference; 27 tensors per block) gets its
gradient, and the data gradient is handed on to the block's input (ldm/modules/attention.py:230-266 is what is
differentiated; torch.autograd does this for the reference).

What runs where: the products -- y = x W^T, dx = dy W, dW = dy^T x -- are mobi_igemm launches (the last on operands
transposed by mobi_transpose, token axis as k, fp32 result); LayerNorm, GEGLU and attention have backward kernels of their
own (csrc/backward.hip: correct and bit-reproducible, not tuned).  The training forward is the UN-folded sequence: the
sampling path's algebra (two-key adapter tables, connector o to_out, LayerNorm folded into to_q, the row chains) bakes
trainable weights into per-run constants and has no place here.  Frozen layers (attn1, attn2, the feed-forward, norm1 /
norm3) only pass the data gradient on.

Not built yet (the rest of row 4): the same for ResBlock / GroupNorm / the convolutions' data gradients, i.e. the chain
through the whole UNet, the loss scaling an fp16 run needs, the optimizer step.  `mobi_amd.dist.allreduce_gradients` is the
gradient collective (bucketed, RCCL; gloo on CPU in the tests).
"""
import torch

from . import engine_dtype, ops
from .ops import Packed

TRAINABLE_MARKERS = ("cond_adapter", "lidar", "cross_modal")          # ddpm.py:1622-1626 of the reference


def trainable_names(module):
    """Parameter names the reference's `configure_optimizers` would hand to AdamW (ddpm.py:1616-1633), relative to `module`."""
    return [n for n, _ in module.named_parameters() if any(m in n for m in TRAINABLE_MARKERS)]


def _packed_t(lin):
    """The layer's weight read as [in][out]: `dx = dy W` is then an ordinary linear launch (no bias)."""
    key = (lin.weight._version, lin.weight.data_ptr(), engine_dtype())
    c = lin.__dict__.setdefault("_packed_t", {})
    if c.get("key") != key:
        c["key"], c["val"] = key, ops.pack_linear(lin.weight.detach().t().contiguous(), None, engine_dtype(), lin.weight.device)
    return c["val"]


def _rows(t):
    return t.reshape(-1, t.shape[-1])


def _dgrad(dy, lin, residual=None, out=None):
    return ops.linear(dy, _packed_t(lin), residual=residual, out=out)


def _wgrad(grads, name, lin, dy, x):
    """dW (and db) of `y = lin(x)` into grads[name + '.weight' / '.bias'] (fp32)."""
    grads[name + ".weight"] = ops.linear_wgrad(_rows(dy), _rows(x))
    if lin.bias is not None:
        grads[name + ".bias"] = ops.colsum(_rows(dy))


class BlockTape:
    """Activations one block's backward pass reads (kept in the engine's 16-bit storage type, as the forward wrote them)."""


def block_forward(blk, x, context):
    """BasicTransformerBlock._forward for training: x T [N, T, C] dense, context fp32 [N, 2, ctx_dim] -> (out, tape)."""
    assert blk.bbox_cond and blk.multimodal and x.shape[0] % 2 == 0 and context.shape[1] == 2
    t = BlockTape()
    n, tok, c = x.shape
    ln = lambda norm, v: ops.layernorm(v, *norm.affine(), norm.eps)
    a1m = blk.attn1
    t.x = x
    t.xn1 = ln(blk.norm1, x)
    t.q1, t.k1, t.v1 = (ops.linear(t.xn1, m.packed()) for m in (a1m.to_q, a1m.to_k, a1m.to_v))
    t.a1 = ops.attention(t.q1, t.k1, t.v1, a1m.heads, a1m.scale, v_rows=True)
    ctx = context.float().contiguous()
    vec = blk.attn2.single_token_vector(ctx[:, 0])               # one key: softmax == 1 (frozen; no gradient path to x)
    t.x2 = ops.linear(t.a1, a1m.to_out[0].packed(), residual=x, rowvec=vec)
    # bbox adapter (trainable)
    ca = blk.cond_adapter_attn
    t.ctx_t = ctx.to(x.dtype)
    t.xn_a = ln(blk.cond_adapter_norm, t.x2)
    t.q_a = ops.linear(t.xn_a, ca.to_q.packed())
    t.k_a, t.v_a = ops.linear(t.ctx_t, ca.to_k.packed()), ops.linear(t.ctx_t, ca.to_v.packed())
    t.a_a = ops.ctx_attention(t.q_a, t.k_a.float().contiguous(), t.v_a.float().contiguous(), ca.heads, ca.scale)
    t.y_a = ops.linear(t.a_a, ca.to_out[0].packed())
    t.x3 = ops.linear(t.y_a, blk.cond_adapter_connector.packed(), residual=t.x2)
    # cross-modal: camera rows attend to the lidar rows, then lidar rows to the UPDATED camera rows (attention.py:249-261)
    xc, xl = t.x3[::2], t.x3[1::2]
    t.x4 = torch.empty_like(t.x3)
    cam, lid = blk.cross_modal_attn_camera, blk.cross_modal_attn_lidar
    t.xn_c = ln(blk.cross_modal_norm_camera, xc)
    t.q_c, t.k_c, t.v_c = ops.linear(t.xn_c, cam.to_q.packed()), ops.linear(xl, cam.to_k.packed()), ops.linear(xl, cam.to_v.packed())
    t.a_c = ops.attention(t.q_c, t.k_c, t.v_c, cam.heads, cam.scale, v_rows=True)
    t.y_c = ops.linear(t.a_c, cam.to_out[0].packed())
    ops.linear(t.y_c, blk.cross_modal_connector_camera.packed(), residual=xc, out=t.x4[::2])
    xc2 = t.x4[::2]
    t.xn_l = ln(blk.cross_modal_norm_lidar, xl)
    t.q_l, t.k_l, t.v_l = ops.linear(t.xn_l, lid.to_q.packed()), ops.linear(xc2, lid.to_k.packed()), ops.linear(xc2, lid.to_v.packed())
    t.a_l = ops.attention(t.q_l, t.k_l, t.v_l, lid.heads, lid.scale, v_rows=True)
    t.y_l = ops.linear(t.a_l, lid.to_out[0].packed())
    ops.linear(t.y_l, blk.cross_modal_connector_lidar.packed(), residual=xl, out=t.x4[1::2])
    # feed-forward (frozen), GEGLU un-fused so that its inputs are on the tape
    t.xn3 = ln(blk.norm3, t.x4)
    t.pre = ops.linear(t.xn3, blk.ff.net[0].proj.packed())
    t.h = ops.geglu_fwd(t.pre)
    out = ops.linear(t.h, blk.ff.net[2].packed(), residual=t.x4)
    return out, t


def _attn_grads(grads, name, attn, connector, cname, dres, t_q_in, t_kv_in, q, k, v, a, y, scale_heads):
    """Backward of `res + connector(to_out(attention(to_q(q_in), to_k(kv_in), to_v(kv_in))))` given d(res + ...) = dres
    (a [n, T, C] view): parameter gradients into `grads`, returns (d q_in, dk, dv) -- the key / value data gradients still
    as gradients of the projections' OUTPUTS (the caller pushes them through to_k / to_v and into the right rows)."""
    heads, scale = scale_heads
    dy = _dgrad(dres, connector)
    _wgrad(grads, cname, connector, dres, y)
    da = _dgrad(dy, attn.to_out[0])
    _wgrad(grads, name + ".to_out.0", attn.to_out[0], dy, a)
    dq, dk, dv = ops.attention_bwd(q, k, v, a, da, heads, scale)
    _wgrad(grads, name + ".to_q", attn.to_q, dq, t_q_in)
    return _dgrad(dq, attn.to_q), dk, dv


def block_backward(blk, tape, dout):
    """dout: T [N, T, C] dense, the gradient of the loss w.r.t. `block_forward`'s output -> (dx T [N, T, C], grads):
    grads = {parameter name relative to the block: fp32 tensor} for every name of `trainable_names(blk)`."""
    t, g = tape, {}
    n, tok, c = dout.shape
    ff = blk.ff
    # ---- feed-forward (frozen): out = W2 geglu(W1 LN3(x4)) + x4
    dh = _dgrad(dout, ff.net[2])
    dpre = ops.geglu_bwd(t.pre, dh)
    dxn3 = _dgrad(dpre, ff.net[0].proj)
    dx4, _, _ = ops.layernorm_bwd(t.x4, dxn3, blk.norm3.affine()[0], blk.norm3.eps, dx_add=dout)
    xc, xl, xc2 = t.x3[::2], t.x3[1::2], t.x4[::2]
    cam, lid = blk.cross_modal_attn_camera, blk.cross_modal_attn_lidar
    # ---- lidar rows: xl' = xl + connector(attn(LN_l(xl), context = xc'))
    dxl2 = dx4[1::2]
    dxn_l, dk_l, dv_l = _attn_grads(g, "cross_modal_attn_lidar", lid, blk.cross_modal_connector_lidar, "cross_modal_connector_lidar",
                                    dxl2, t.xn_l, xc2, t.q_l, t.k_l, t.v_l, t.a_l, t.y_l, (lid.heads, lid.scale))
    _wgrad(g, "cross_modal_attn_lidar.to_k", lid.to_k, dk_l, xc2)
    _wgrad(g, "cross_modal_attn_lidar.to_v", lid.to_v, dv_l, xc2)
    dxl, g["cross_modal_norm_lidar.weight"], g["cross_modal_norm_lidar.bias"] = ops.layernorm_bwd(
        xl, dxn_l, blk.cross_modal_norm_lidar.affine()[0], blk.cross_modal_norm_lidar.eps, dx_add=dxl2.contiguous())
    # ---- camera rows: xc' = xc + connector(attn(LN_c(xc), context = xl)); d xc' = its own rows of dx4 + what the lidar's keys / values send back
    dxc2 = _dgrad(dv_l, lid.to_v, residual=_dgrad(dk_l, lid.to_k, residual=dx4[::2]))
    dxn_c, dk_c, dv_c = _attn_grads(g, "cross_modal_attn_camera", cam, blk.cross_modal_connector_camera, "cross_modal_connector_camera",
                                    dxc2, t.xn_c, xl, t.q_c, t.k_c, t.v_c, t.a_c, t.y_c, (cam.heads, cam.scale))
    _wgrad(g, "cross_modal_attn_camera.to_k", cam.to_k, dk_c, xl)
    _wgrad(g, "cross_modal_attn_camera.to_v", cam.to_v, dv_c, xl)
    dxc, g["cross_modal_norm_camera.weight"], g["cross_modal_norm_camera.bias"] = ops.layernorm_bwd(
        xc, dxn_c, blk.cross_modal_norm_camera.affine()[0], blk.cross_modal_norm_camera.eps, dx_add=dxc2)
    dx3 = torch.empty((n, tok, c), device=dout.device, dtype=dout.dtype)
    _dgrad(dv_c, cam.to_v, residual=_dgrad(dk_c, cam.to_k, residual=dxl), out=dx3[1::2])      # lidar rows: own + the camera's keys / values
    dx3[::2].copy_(dxc)                                                                        # (placement, not arithmetic)
    # ---- bbox adapter: x3 = x2 + connector(attn(LN_a(x2), context tokens))
    ca = blk.cond_adapter_attn
    dxn_a, dk_a, dv_a = _attn_grads(g, "cond_adapter_attn", ca, blk.cond_adapter_connector, "cond_adapter_connector", dx3, t.xn_a,
                                    None, t.q_a, t.k_a, t.v_a, t.a_a, t.y_a, (ca.heads, ca.scale))
    # to_k / to_v saw the 2 N context tokens: a handful of rows -> fp32 FMA chains (mobi_linear_f32) instead of the matrix cores
    ctx32 = _rows(t.ctx_t).float()
    for nm, dd in (("to_k", dk_a), ("to_v", dv_a)):
        g[f"cond_adapter_attn.{nm}.weight"] = ops.linear_f32(_rows(dd).float().t().contiguous(), ctx32.t().contiguous())
    dx2, g["cond_adapter_norm.weight"], g["cond_adapter_norm.bias"] = ops.layernorm_bwd(
        t.x2, dxn_a, blk.cond_adapter_norm.affine()[0], blk.cond_adapter_norm.eps, dx_add=dx3)
    # ---- attn2 adds a per-image constant: identity for the data gradient.  attn1 (frozen): x2 = to_out(attention(qkv(LN1(x)))) + x
    a1m = blk.attn1
    da1 = _dgrad(dx2, a1m.to_out[0])
    dq, dk, dv = ops.attention_bwd(t.q1, t.k1, t.v1, t.a1, da1, a1m.heads, a1m.scale)
    dxn1 = _dgrad(dv, a1m.to_v, residual=_dgrad(dk, a1m.to_k, residual=_dgrad(dq, a1m.to_q)))
    dx, _, _ = ops.layernorm_bwd(t.x, dxn1, blk.norm1.affine()[0], blk.norm1.eps, dx_add=dx2)
    return dx, g
