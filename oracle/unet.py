"""Functional PyTorch-CPU restatement of MObI's UNet (TEST INFRASTRUCTURE).

Works on a flat `state_dict` with the reference's key names (the keys under
`model.diffusion_model.` in the released checkpoint), so the same synthetic
parameters drive the reference modules, this oracle and the HIP engine.

Reference: ldm/modules/diffusionmodules/openaimodel.py (UNetModel :558-898,
ResBlock :255-275, Upsample :109-119, Downsample :158-160),
ldm/modules/attention.py (SpatialTransformer :302-313, BasicTransformerBlock
:230-266, CrossAttention :171-194, GEGLU :38-46),
ldm/modules/diffusionmodules/util.py (timestep_embedding :151-171,
GroupNorm32 :214-216).
"""
import math
from dataclasses import dataclass, field
from typing import List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    """Subset of `UNetModel.__init__` kwargs that MObI's configs set
    (configs/mobi_nusc_512.yaml:63-82)."""
    in_channels: int = 9
    model_channels: int = 320
    out_channels: int = 4
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8
    context_dim: int = 768
    transformer_depth: int = 1
    bbox_cond: bool = True
    use_camera: bool = True
    use_lidar: bool = True

    @property
    def multimodal(self):
        return bool(self.use_camera and self.use_lidar)


# ----------------------------------------------------------------------------
# structure: which blocks exist, in the order openaimodel.py:681-836 builds them
# ----------------------------------------------------------------------------

def unet_layout(cfg: UNetConfig):
    """Returns (input_blocks, middle, output_blocks); each block is a list of
    layer descriptors ('conv'|'res'|'st'|'down'|'up', cin, cout)."""
    mc = cfg.model_channels
    inputs = [[("conv", cfg.in_channels, mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers.append(("st", ch, ch))
            inputs.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            inputs.append([("down", ch, ch)])
            chans.append(ch)
            ds *= 2
    middle = [("res", ch, ch), ("st", ch, ch), ("res", ch, ch)]
    outputs = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                layers.append(("st", ch, ch))
            if level and i == cfg.num_res_blocks:
                layers.append(("up", ch, ch))
                ds //= 2
            outputs.append(layers)
    return inputs, middle, outputs


def unet_param_shapes(cfg: UNetConfig):
    """{key: shape} of every UNet parameter, reference key names."""
    shapes = {}
    mc, ted, cd = cfg.model_channels, cfg.model_channels * 4, cfg.context_dim

    def lin(p, cin, cout, bias=True):
        shapes[p + ".weight"] = (cout, cin)
        if bias:
            shapes[p + ".bias"] = (cout,)

    def conv(p, cin, cout, k):
        shapes[p + ".weight"] = (cout, cin, k, k)
        shapes[p + ".bias"] = (cout,)

    def norm(p, c):
        shapes[p + ".weight"] = (c,)
        shapes[p + ".bias"] = (c,)

    def attn(p, qd, kd):
        lin(p + ".to_q", qd, qd, False)
        lin(p + ".to_k", kd, qd, False)
        lin(p + ".to_v", kd, qd, False)
        lin(p + ".to_out.0", qd, qd)

    def res(p, cin, cout):
        norm(p + ".in_layers.0", cin)
        conv(p + ".in_layers.2", cin, cout, 3)
        lin(p + ".emb_layers.1", ted, cout)
        norm(p + ".out_layers.0", cout)
        conv(p + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip_connection", cin, cout, 1)

    def st(p, c):
        norm(p + ".norm", c)
        conv(p + ".proj_in", c, c, 1)
        for d in range(cfg.transformer_depth):
            b = f"{p}.transformer_blocks.{d}"
            attn(b + ".attn1", c, c)
            lin(b + ".ff.net.0.proj", c, 8 * c)
            lin(b + ".ff.net.2", 4 * c, c)
            attn(b + ".attn2", c, cd)
            for n in ("norm1", "norm2", "norm3"):
                norm(f"{b}.{n}", c)
            if cfg.bbox_cond:
                attn(b + ".cond_adapter_attn", c, cd)
                norm(b + ".cond_adapter_norm", c)
                lin(b + ".cond_adapter_connector", c, c)
            if cfg.multimodal:
                for m in ("camera", "lidar"):
                    attn(f"{b}.cross_modal_attn_{m}", c, c)
                    norm(f"{b}.cross_modal_norm_{m}", c)
                    lin(f"{b}.cross_modal_connector_{m}", c, c)
        conv(p + ".proj_out", c, c, 1)

    def block(p, layers):
        for j, (kind, cin, cout) in enumerate(layers):
            q = f"{p}.{j}"
            if kind == "conv":
                conv(q, cin, cout, 3)
            elif kind == "res":
                res(q, cin, cout)
            elif kind == "st":
                st(q, cin)
            elif kind == "down":
                conv(q + ".op", cin, cout, 3)
            elif kind == "up":
                conv(q + ".conv", cin, cout, 3)

    lin("time_embed.0", mc, ted)
    lin("time_embed.2", ted, ted)
    inputs, middle, outputs = unet_layout(cfg)
    for i, layers in enumerate(inputs):
        block(f"input_blocks.{i}", layers)
    block("middle_block", middle)
    for i, layers in enumerate(outputs):
        block(f"output_blocks.{i}", layers)
    norm("out.0", mc)
    conv("out.2", mc, cfg.out_channels, 3)
    return shapes


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------

def timestep_embedding(t, dim, max_period=10000):
    """util.py:151-171 (repeat_only=False)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _gn(sd, p, x, eps):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def res_block(sd, p, x, emb):
    """ResBlock._forward, openaimodel.py:255-275 (no up/down, no scale-shift)."""
    h = _conv(sd, p + ".in_layers.2", F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)))
    h = h + _lin(sd, p + ".emb_layers.1", F.silu(emb))[:, :, None, None]
    h = _conv(sd, p + ".out_layers.3", F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)))
    if (p + ".skip_connection.weight") in sd:
        x = _conv(sd, p + ".skip_connection", x, padding=0)
    return x + h


def cross_attention(sd, p, x, context, heads):
    """CrossAttention.forward, attention.py:171-194 (mask=None, dropout 0)."""
    context = x if context is None else context
    q, k, v = _lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", context), _lin(sd, p + ".to_v", context)
    b, n, c = q.shape
    d = c // heads
    split = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (d ** -0.5)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, c)
    return _lin(sd, p + ".to_out.0", out)


def transformer_block(sd, p, x, context, cfg: UNetConfig):
    """BasicTransformerBlock._forward, attention.py:230-266."""
    h = cfg.num_heads
    if context is not None and context.shape[1] > 1 and not cfg.bbox_cond:
        context = context[:, [0]]
    x = cross_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, h) + x
    x = cross_attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context[:, [0]], h) + x
    if cfg.bbox_cond:
        a = cross_attention(sd, p + ".cond_adapter_attn", _ln(sd, p + ".cond_adapter_norm", x), context, h)
        x = _lin(sd, p + ".cond_adapter_connector", a) + x
    if cfg.multimodal:
        xc, xl = x[::2], x[1::2]
        a = cross_attention(sd, p + ".cross_modal_attn_camera", _ln(sd, p + ".cross_modal_norm_camera", xc), xl, h)
        xc = _lin(sd, p + ".cross_modal_connector_camera", a) + xc
        # lidar attends to the ALREADY UPDATED camera stream (attention.py:257-261)
        a = cross_attention(sd, p + ".cross_modal_attn_lidar", _ln(sd, p + ".cross_modal_norm_lidar", xl), xc, h)
        xl = _lin(sd, p + ".cross_modal_connector_lidar", a) + xl
        x = torch.stack([xc, xl], dim=1).reshape(-1, *xc.shape[1:])   # cat_interleave, ldm/util.py:213-221
    y = _lin(sd, p + ".ff.net.0.proj", _ln(sd, p + ".norm3", x))
    a, gate = y.chunk(2, dim=-1)
    x = _lin(sd, p + ".ff.net.2", a * F.gelu(gate)) + x
    return x


def spatial_transformer(sd, p, x, context, cfg: UNetConfig):
    """SpatialTransformer.forward, attention.py:302-313."""
    b, c, hh, ww = x.shape
    x_in = x
    x = F.group_norm(x, 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)
    x = _conv(sd, p + ".proj_in", x, padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    for d in range(cfg.transformer_depth):
        x = transformer_block(sd, f"{p}.transformer_blocks.{d}", x, context, cfg)
    x = x.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    x = _conv(sd, p + ".proj_out", x, padding=0)
    return x + x_in


def _run_block(sd, p, layers, h, emb, context, cfg):
    for j, (kind, cin, cout) in enumerate(layers):
        q = f"{p}.{j}"
        if kind == "conv":
            h = _conv(sd, q, h)
        elif kind == "res":
            h = res_block(sd, q, h, emb)
        elif kind == "st":
            h = spatial_transformer(sd, q, h, context, cfg)
        elif kind == "down":
            h = _conv(sd, q + ".op", h, stride=2)                        # openaimodel.py:150-160
        elif kind == "up":
            h = _conv(sd, q + ".conv", F.interpolate(h, scale_factor=2, mode="nearest"))   # :116-118
    return h


def unet_forward(sd, cfg: UNetConfig, x, timesteps, context):
    """UNetModel.forward, openaimodel.py:861-898."""
    inputs, middle, outputs = unet_layout(cfg)
    emb = timestep_embedding(timesteps, cfg.model_channels)
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", emb)))
    hs = []
    h = x.float()
    for i, layers in enumerate(inputs):
        h = _run_block(sd, f"input_blocks.{i}", layers, h, emb, context, cfg)
        hs.append(h)
    h = _run_block(sd, "middle_block", middle, h, emb, context, cfg)
    for i, layers in enumerate(outputs):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, f"output_blocks.{i}", layers, h, emb, context, cfg)
    h = F.silu(_gn(sd, "out.0", h, 1e-5))
    return _conv(sd, "out.2", h)
