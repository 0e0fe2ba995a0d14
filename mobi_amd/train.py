"""The training step of the adapter parameters on the engine: forward WITH a tape and the backward
pass through the WHOLE UNet -- every tensor the reference's optimizer filter selects (`cond_adapter*`, `cross_modal*`:
ldm/models/diffusion/ddpm.py:1616-1629 of the reference; 27 tensors in each of the 16 transformer blocks = 432 tensors,
180 M parameters at full width) gets its gradient.  ldm/modules/attention.py:230-266 and
ldm/modules/diffusionmodules/openaimodel.py:255-275, 861-898 are what is differentiated; torch.autograd does this for the
reference, Lightning's DDP all-reduces the result (main.py:510).

What runs where: the products -- y = x W^T, dx = dy W, dW = dy^T x, and the data gradients of the 3 x 3 / strided /
upsampling convolutions (the kernel rotated by 180 degrees with its in / out axes swapped; stride 2: dy spread onto the even
pixels first; nearest x2: 2 x 2 sums after) -- are mobi_igemm launches; LayerNorm, GroupNorm (+ SiLU), GEGLU and attention
have backward kernels of their own (csrc/backward.hip: correct and bit-reproducible, NOT tuned: fp32 vector arithmetic).
The training forward is the UN-folded sequence: the sampling path's algebra (two-key adapter tables, connector o to_out,
LayerNorm folded into to_q, the row chains, the un-materialised skip concat) bakes trainable weights into per-run constants
or hides tensors the backward pass needs, and has no place here.  Frozen layers only pass the data gradient on.

Pinned to torch.autograd through the CPU oracle (tests/test_gpu_backward.py): one block, and the reduced UNet end to end.
Also here: the conditioning stage's trainable part (the 3-D box embedder, `bbox_uncond_vector`: ddpm.py:1635-1647) and AdamW.
Not built: `logvar`, LR schedulers, activation checkpointing (the tape of a full-width step at 64 x 64 x 4 is 17 GB: fine in 288 GB).
`mobi_amd.dist.allreduce_gradients` is the gradient collective (bucketed, RCCL; gloo on CPU in the tests).
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
    grads = {parameter name relative to the block: fp32 tensor} for every name of `trainable_names(blk)`, plus
    `__dcontext__`: fp32 [N, 2, ctx_dim], the gradient w.r.t. the context tokens through the bbox adapter's to_k / to_v
    (what the conditioning stage's trainable tensors receive; attn2's path to token 0 ends in frozen tensors only)."""
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
    dctx = None                                  # d loss / d context tokens [2 N, ctx_dim] fp32, through to_k and to_v
    for nm, dd, lin in (("to_k", dk_a, ca.to_k), ("to_v", dv_a, ca.to_v)):
        d32 = _rows(dd).float().contiguous()
        g[f"cond_adapter_attn.{nm}.weight"] = ops.linear_f32(d32.t().contiguous(), ctx32.t().contiguous())
        part = ops.linear_f32(d32, lin.weight.detach().float().t().contiguous())
        dctx = part if dctx is None else ops.lincomb4([dctx, part], [1.0, 1.0])
    g["__dcontext__"] = dctx.view(n, 2, -1)
    dx2, g["cond_adapter_norm.weight"], g["cond_adapter_norm.bias"] = ops.layernorm_bwd(
        t.x2, dxn_a, blk.cond_adapter_norm.affine()[0], blk.cond_adapter_norm.eps, dx_add=dx3)
    # ---- attn2 adds a per-image constant: identity for the data gradient.  attn1 (frozen): x2 = to_out(attention(qkv(LN1(x)))) + x
    a1m = blk.attn1
    da1 = _dgrad(dx2, a1m.to_out[0])
    dq, dk, dv = ops.attention_bwd(t.q1, t.k1, t.v1, t.a1, da1, a1m.heads, a1m.scale)
    dxn1 = _dgrad(dv, a1m.to_v, residual=_dgrad(dk, a1m.to_k, residual=_dgrad(dq, a1m.to_q)))
    dx, _, _ = ops.layernorm_bwd(t.x, dxn1, blk.norm1.affine()[0], blk.norm1.eps, dx_add=dx2)
    return dx, g


# ======================================================================================================================
# The whole UNet (openaimodel.py:861-898 of the reference): forward with a tape, backward to every adapter tensor
# ======================================================================================================================
def _conv_dgrad_pack(conv):
    """Data gradient of a stride-1 convolution = the convolution of dy with the kernel rotated by 180 degrees and its
    in / out axes swapped (the same padding for the odd kernels used here): packed once like any other weight."""
    key = (conv.weight._version, conv.weight.data_ptr(), engine_dtype())
    c = conv.__dict__.setdefault("_dgrad_pack", {})
    if c.get("key") != key:
        w = conv.weight.detach().flip(2, 3).transpose(0, 1).contiguous()
        c["key"], c["val"] = key, ops.pack_conv(w, None, engine_dtype(), conv.weight.device)
    return c["val"]


def _res_forward(rb, x, embd, skip):
    lo, hi = rb._emb_slice
    emb_out = embd["proj"][:, lo:hi]                           # (carries conv1's bias: UNetModel._emb_projection)
    xin = x if skip is None else torch.cat([x, skip], dim=3)  # the concat IS materialised here: its gradient is split below
    n1, n2 = rb.in_layers[0], rb.out_layers[0]
    a1 = ops.groupnorm(xin, *n1.affine(), n1.eps, silu=True)
    h1 = ops.igemm(a1, rb.in_layers[2].packed(), rowvec=emb_out, rowvec_has_bias=True)
    a2 = ops.groupnorm(h1, *n2.affine(), n2.eps, silu=True)
    has_conv = not isinstance(rb.skip_connection, torch.nn.Identity) and hasattr(rb.skip_connection, "weight")
    xs = ops.igemm(xin, rb.skip_connection.packed()) if has_conv else xin
    out = ops.igemm(a2, rb.out_layers[3].packed(), residual=xs)
    return out, (xin, h1, x.shape[3], has_conv)


def _res_backward(rb, tape, dout):
    xin, h1, c0, has_conv = tape
    n1, n2 = rb.in_layers[0], rb.out_layers[0]
    da2 = ops.igemm(dout, _conv_dgrad_pack(rb.out_layers[3]))
    dh1 = ops.groupnorm_bwd(h1, da2, *n2.affine(), n2.eps, True)
    da1 = ops.igemm(dh1, _conv_dgrad_pack(rb.in_layers[2]))
    dside = ops.igemm(dout, _conv_dgrad_pack(rb.skip_connection)) if has_conv else dout
    dxin = ops.groupnorm_bwd(xin, da1, *n1.affine(), n1.eps, True, dx_add=dside)
    if c0 < xin.shape[3]:
        return dxin[..., :c0].contiguous(), dxin[..., c0:].contiguous()
    return dxin, None


def _st_forward(st, x, context):
    n, h, w, c = x.shape
    g, b = st.norm.affine()
    t = ops.igemm(ops.groupnorm(x, g, b, st.norm.eps, silu=False), st.proj_in.packed())
    t = t.view(n, h * w, t.shape[3])
    tapes = []
    for blk in st.transformer_blocks:
        t, tp = block_forward(blk, t, context)
        tapes.append(tp)
    return ops.igemm(t.view(n, h, w, t.shape[2]), st.proj_out.packed(), residual=x), (x, tapes)


def _st_backward(st, tape, dout, grads, prefix):
    x, tapes = tape
    n, h, w, c = x.shape
    dt = ops.igemm(dout, _conv_dgrad_pack(st.proj_out))
    dt = dt.view(n, h * w, dt.shape[3])
    for i in reversed(range(len(tapes))):
        dt, g = block_backward(st.transformer_blocks[i], tapes[i], dt)
        dc = g.pop("__dcontext__")
        grads["__dcontext__"] = dc if "__dcontext__" not in grads else ops.lincomb4([grads["__dcontext__"], dc], [1.0, 1.0])
        grads.update({f"{prefix}.transformer_blocks.{i}.{k}": v for k, v in g.items()})
    dxn = ops.igemm(dt.view(n, h, w, dt.shape[2]), _conv_dgrad_pack(st.proj_in))
    g, b = st.norm.affine()
    return ops.groupnorm_bwd(x, dxn, g, b, st.norm.eps, False, dx_add=dout)


def _seq_forward(seq, h, embd, context, skip=None):
    from .ldm.modules.attention import SpatialTransformer
    from .ldm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample, _plain_conv
    tapes = []
    for layer in seq:
        if isinstance(layer, ResBlock):
            h, tp = _res_forward(layer, h, embd, skip)
            skip = None
        elif isinstance(layer, SpatialTransformer):
            h, tp = _st_forward(layer, h, context)
        elif isinstance(layer, (Downsample, Upsample)):
            tp, h = tuple(h.shape), layer(h)
        else:
            h, tp = _plain_conv(layer, h), None          # input_blocks.0: nothing trainable in front of it, the gradient stops
        tapes.append(tp)
    return h, tapes


def _seq_backward(seq, tapes, dh, grads, prefix):
    from .ldm.modules.attention import SpatialTransformer
    from .ldm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample
    dskip = None
    for i in reversed(range(len(seq))):
        layer, tp = seq[i], tapes[i]
        if isinstance(layer, ResBlock):
            dh, dskip = _res_backward(layer, tp, dh)
        elif isinstance(layer, SpatialTransformer):
            dh = _st_backward(layer, tp, dh, grads, f"{prefix}.{i}")
        elif isinstance(layer, Upsample):
            dh = ops.sumpool2(ops.igemm(dh, _conv_dgrad_pack(layer.conv)))       # nearest x2, then the convolution
        elif isinstance(layer, Downsample):
            # stride 2: dy spread onto the even pixels of the input grid (placement), then the stride-1 data gradient
            n, ho, wo, c = dh.shape
            dz = torch.zeros((n, tp[1], tp[2], c), device=dh.device, dtype=dh.dtype)
            dz[:, : 2 * ho: 2, : 2 * wo: 2] = dh
            dh = ops.igemm(dz, _conv_dgrad_pack(layer.op))
        else:
            dh = None
    return dh, dskip


def unet_forward(net, x, timesteps, context):
    """UNetModel.forward for training: x fp32 [N, in_channels, h, w] -> (eps fp32 [N, out_channels, h, w], tape)."""
    from ._lib import ACT_SILU
    from .ldm.modules.diffusionmodules.util import timestep_embedding
    t_emb = timestep_embedding(timesteps, net.model_channels)
    w0, b0 = net.time_embed[0].skinny()
    w2, b2 = net.time_embed[2].skinny()
    emb = ops.skinny_linear(ops.skinny_linear(t_emb, w0, b0, post_act=ACT_SILU), w2, b2)
    we, be = net._emb_projection()
    embd = {"emb": emb, "proj": ops.skinny_linear(emb, we, be, pre_act=ACT_SILU)}
    context = context.float().contiguous()
    tape = {"in": [], "out": []}
    hs, h = [], x
    for module in net.input_blocks:
        h, tp = _seq_forward(module, h, embd, context)
        hs.append(h)
        tape["in"].append(tp)
    h, tape["mid"] = _seq_forward(net.middle_block, h, embd, context)
    for module in net.output_blocks:
        h, tp = _seq_forward(module, h, embd, context, skip=hs.pop())
        tape["out"].append(tp)
    n0 = net.out[0]
    tape["head"] = h
    a = ops.groupnorm(h, *n0.affine(), n0.eps, silu=True)
    return ops.conv_small_cout(a, net.out[2].packed_tap_major(), pad=net.out[2].padding), tape


def unet_backward(net, tape, deps):
    """deps: fp32 [N, out_channels, h, w], the gradient of the loss w.r.t. `unet_forward`'s result -> {parameter name relative
    to the UNet: fp32 gradient} for every tensor of `trainable_names(net)` (ddpm.py:1616-1629 of the reference)."""
    grads = {}
    conv = net.out[2]
    key = (conv.weight._version, conv.weight.data_ptr(), engine_dtype())
    c = conv.__dict__.setdefault("_dgrad_thin", {})
    if c.get("key") != key:                       # 4 -> 320 channels: the thin side zero-padded to 32 like the input convolution's
        w = conv.weight.detach().flip(2, 3).transpose(0, 1).contiguous()
        c["key"], c["val"] = key, ops.pack_conv_padded_cin(w, None, engine_dtype(), conv.weight.device)
    dy = ops.pack_sources([deps.float().contiguous()], engine_dtype())
    n0 = net.out[0]
    dh = ops.groupnorm_bwd(tape["head"], ops.igemm(dy, c["val"]), *n0.affine(), n0.eps, True)
    nin = len(net.input_blocks)
    dhs = {}
    for i in reversed(range(len(net.output_blocks))):
        dh, dskip = _seq_backward(net.output_blocks[i], tape["out"][i], dh, grads, f"output_blocks.{i}")
        dhs[nin - 1 - i] = dskip                   # output block i consumed hs.pop() = the result of input block nin - 1 - i
    dh, _ = _seq_backward(net.middle_block, tape["mid"], dh, grads, "middle_block")
    for j in reversed(range(1, nin)):              # input_blocks.0 is the bare input convolution: nothing trainable before it
        dh = ops.add(dh, dhs[j])
        dh, _ = _seq_backward(net.input_blocks[j], tape["in"][j], dh, grads, f"input_blocks.{j}")
    return grads


def loss_and_gradients(net, x_noisy, timesteps, context, target, loss_scale=1.0):
    """The eps-parameterised simple loss of `p_losses` (ddpm.py:1177-1217: mean squared error of the UNet's output against
    the noise) and its gradient w.r.t. every adapter tensor.  loss_scale multiplies the gradient that enters the backward
    pass and is divided out of the fp32 results (fp16 storage underflows without it at production sizes)."""
    eps, tape = unet_forward(net, x_noisy, timesteps, context)
    target = target.float().contiguous()
    loss = torch.mean((eps - target) ** 2)
    k = 2.0 * loss_scale / eps.numel()
    deps = ops.lincomb4([eps.contiguous(), target], [k, -k])
    grads = unet_backward(net, tape, deps)
    if loss_scale != 1.0:
        grads = {name: ops.lincomb4([g.contiguous()], [1.0 / loss_scale]) for name, g in grads.items()}
    return loss, grads                            # (grads["__dcontext__"]: the gradient w.r.t. the context tokens)


# ======================================================================================================================
# The conditioning stage's trainable part (ddpm.py:1635-1647 of the reference): the 3-D box embedder and `bbox_uncond_vector`
# ======================================================================================================================
def bbox_embedder_forward(emb, bbox):
    """BBoxEmbedder.forward (ldm/modules/encoders/modules.py:85-91 of the reference) with a tape: bbox fp32 [B, 8, 3] ->
    (token fp32 [B, 1, 768], tape).  Fourier features, then Linear -> Linear, SiLU, Linear, SiLU, Linear as fp32 GEMV chains."""
    from ._lib import ACT_SILU
    from .ldm.modules.encoders.modules import fourier_features
    e = fourier_features(bbox.float(), emb.num_freqs).reshape(bbox.shape[0], -1).contiguous()
    gemv = lambda lin, x, pre=0: ops.skinny_linear(x.contiguous(), *lin.skinny(), pre_act=pre)
    h0 = gemv(emb.bbox_proj, e)
    z1 = gemv(emb.second_linear[0], h0)
    z2 = gemv(emb.second_linear[2], z1, ACT_SILU)
    out = gemv(emb.second_linear[4], z2, ACT_SILU)
    return out.unsqueeze(1), (e, h0, z1, z2)


def bbox_embedder_backward(emb, tape, dtoken):
    """dtoken: fp32 [B, 1, 768] -> {`bbox_proj.weight`, ..., `second_linear.4.bias`: fp32 gradient} (8 tensors).  A handful of rows:
    every product is an fp32 FMA chain (mobi_linear_f32)."""
    e, h0, z1, z2 = tape
    silu = lambda z: z * torch.sigmoid(z)          # (recomputed activations: [B, 512] fp32 elementwise, the GEMV's pre-activation)
    g = {}
    ones = torch.ones((1, dtoken.shape[0]), device=dtoken.device, dtype=torch.float32)

    def layer(name, lin, dy, x, need_dx=True):
        g[name + ".weight"] = ops.linear_f32(dy.t().contiguous(), x.t().contiguous())
        g[name + ".bias"] = ops.linear_f32(dy.t().contiguous(), ones).reshape(-1)
        return ops.linear_f32(dy, lin.weight.detach().float().t().contiguous()) if need_dx else None
    dy = dtoken.reshape(dtoken.shape[0], -1).float().contiguous()
    da2 = layer("second_linear.4", emb.second_linear[4], dy, silu(z2))
    dz2 = ops.silu_bwd_f32(z2.contiguous(), da2)
    da1 = layer("second_linear.2", emb.second_linear[2], dz2, silu(z1))
    dz1 = ops.silu_bwd_f32(z1.contiguous(), da1)
    dh0 = layer("second_linear.0", emb.second_linear[0], dz1, h0)
    layer("bbox_proj", emb.bbox_proj, dh0, e, need_dx=False)
    return g


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR for the engine's optimizer (ddpm.py:1662): the rate is the optimizer's initial rate times
    `lr_lambda(step)`, set at construction (step 0) and after every `step()`."""

    def __init__(self, optimizer, lr_lambda):
        self.optimizer, self.lr_lambda = optimizer, lr_lambda
        self.base_lrs = [optimizer.lr]
        self.last_epoch = 0
        optimizer.lr = self.base_lrs[0] * lr_lambda(0)

    def step(self):
        self.last_epoch += 1
        self.optimizer.lr = self.base_lrs[0] * self.lr_lambda(self.last_epoch)

    def get_last_lr(self):
        return [self.optimizer.lr]


class AdamW:
    """torch.optim.AdamW's update (what `configure_optimizers` returns, ddpm.py:1649) on the engine: fp32 master parameters
    updated in place by `mobi_adamw_step`, moments kept per parameter name."""

    def __init__(self, named_params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = dict(named_params)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.state, self.steps = {}, 0

    def step(self, grads):
        """grads: {name: fp32 tensor}; names without an entry are left alone."""
        self.steps += 1
        for name, p in self.params.items():
            if name not in grads:
                continue
            st = self.state.setdefault(name, (torch.zeros_like(p.data, dtype=torch.float32), torch.zeros_like(p.data, dtype=torch.float32)))
            ops.adamw_step(p.data, grads[name].reshape(p.shape).contiguous(), st[0], st[1], self.steps, self.lr, self.betas, self.eps,
                           self.weight_decay)
            torch.autograd.graph.increment_version(p)          # the packed 16-bit copies are keyed on the version counter
        # captured step graphs read the OLD packed copies: a new weights epoch drops them (samplers key their graphs on it, so
        # training-then-sampling re-captures without the caller having to refresh any fingerprint)
        from .ldm.modules.diffusionmodules.util import WEIGHTS_EPOCH
        WEIGHTS_EPOCH[0] += 1
        return self
