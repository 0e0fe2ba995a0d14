"""Functional PyTorch-CPU restatement of AutoencoderKL (TEST INFRASTRUCTURE).

Reference: ldm/models/autoencoder.py:63-72 (encode/decode),
ldm/modules/diffusionmodules/model.py (Normalize :38, Upsample :53-57,
Downsample :72-79, ResnetBlock :121-141, AttnBlock :178-202, Encoder :454-489,
Decoder :587-630), ldm/modules/distributions/distributions.py:24-37.
Keys are the reference's (`first_stage_model.` / `lidar_stage_model.` prefix
stripped).
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass
class VAEConfig:
    """`ddconfig` + embed_dim (configs/mobi_nusc_512.yaml:84-130)."""
    in_channels: int = 3
    out_ch: int = 3
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    embed_dim: int = 4
    double_z: bool = True
    lidar_adapter: bool = False


def vae_param_shapes(cfg: VAEConfig):
    s = {}

    def conv(p, cin, cout, kh, kw):
        s[p + ".weight"] = (cout, cin, kh, kw)
        s[p + ".bias"] = (cout,)

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def res(p, cin, cout, kh=3, kw=3):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout, kh, kw)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, kh, kw)
        if cin != cout:
            conv(p + ".nin_shortcut", cin, cout, 1, 1)

    def attn(p, c):
        norm(p + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(f"{p}.{n}", c, c, 1, 1)

    ch, nres = cfg.ch, len(cfg.ch_mult)
    # encoder (model.py:383-453)
    if cfg.lidar_adapter:
        conv("encoder.conv_in_lidar", cfg.in_channels, ch, 1, 5)
        res("encoder.res_block_lidar1", ch, ch, 1, 5)
        res("encoder.res_block_lidar2", ch, ch, 1, 5)
    else:
        conv("encoder.conv_in", cfg.in_channels, ch, 3, 3)
    in_mult = (1,) + tuple(cfg.ch_mult)
    bi = ch
    for l in range(nres):
        bi, bo = ch * in_mult[l], ch * cfg.ch_mult[l]
        for j in range(cfg.num_res_blocks):
            res(f"encoder.down.{l}.block.{j}", bi, bo)
            bi = bo
        if l != nres - 1:
            conv(f"encoder.down.{l}.downsample.conv", bi, bi, 3, 3)
    res("encoder.mid.block_1", bi, bi)
    attn("encoder.mid.attn_1", bi)
    res("encoder.mid.block_2", bi, bi)
    norm("encoder.norm_out", bi)
    conv("encoder.conv_out", bi, 2 * cfg.z_channels if cfg.double_z else cfg.z_channels, 3, 3)
    conv("quant_conv", 2 * cfg.z_channels, 2 * cfg.embed_dim, 1, 1)
    conv("post_quant_conv", cfg.embed_dim, cfg.z_channels, 1, 1)
    # decoder (model.py:510-586)
    bi = ch * cfg.ch_mult[-1]
    conv("decoder.conv_in", cfg.z_channels, bi, 3, 3)
    res("decoder.mid.block_1", bi, bi)
    attn("decoder.mid.attn_1", bi)
    res("decoder.mid.block_2", bi, bi)
    for l in reversed(range(nres)):
        bo = ch * cfg.ch_mult[l]
        for j in range(cfg.num_res_blocks + 1):
            res(f"decoder.up.{l}.block.{j}", bi, bo)
            bi = bo
        if l != 0:
            conv(f"decoder.up.{l}.upsample.conv", bi, bi, 3, 3)
    if cfg.lidar_adapter:
        res("decoder.res_block_lidar1", bi, bi, 1, 5)
        norm("decoder.norm_out_lidar1", bi)
        res("decoder.res_block_lidar2", bi, bi, 1, 5)
        norm("decoder.norm_out_lidar2", bi)
        conv("decoder.conv_out_lidar", bi, cfg.out_ch, 1, 5)
    else:
        norm("decoder.norm_out", bi)
        conv("decoder.conv_out", bi, cfg.out_ch, 3, 3)
    return s


def _swish(x):
    return x * torch.sigmoid(x)                 # model.py:33-35


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def _conv(sd, p, x, stride=1, padding=None):
    w = sd[p + ".weight"]
    if padding is None:
        padding = (w.shape[2] // 2, w.shape[3] // 2)
    return F.conv2d(x, w, sd[p + ".bias"], stride=stride, padding=padding)


def resnet_block(sd, p, x):
    """ResnetBlock.forward with temb=None, model.py:121-141."""
    h = _conv(sd, p + ".conv1", _swish(_gn(sd, p + ".norm1", x)))
    h = _conv(sd, p + ".conv2", _swish(_gn(sd, p + ".norm2", h)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x)
    return x + h


def attn_block(sd, p, x):
    """AttnBlock.forward, model.py:178-202: single head over h*w tokens."""
    h_ = _gn(sd, p + ".norm", x)
    q, k, v = _conv(sd, p + ".q", h_), _conv(sd, p + ".k", h_), _conv(sd, p + ".v", h_)
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h_)


def encoder_forward(sd, cfg: VAEConfig, x):
    """Encoder.forward, model.py:454-489."""
    if cfg.lidar_adapter:
        h = _conv(sd, "encoder.conv_in_lidar", x)
        h = resnet_block(sd, "encoder.res_block_lidar1", h)
        h = resnet_block(sd, "encoder.res_block_lidar2", h)
    else:
        h = _conv(sd, "encoder.conv_in", x)
    nres = len(cfg.ch_mult)
    for l in range(nres):
        for j in range(cfg.num_res_blocks):
            h = resnet_block(sd, f"encoder.down.{l}.block.{j}", h)
        if l != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)           # model.py:74-76
            h = _conv(sd, f"encoder.down.{l}.downsample.conv", h, stride=2, padding=0)
    h = resnet_block(sd, "encoder.mid.block_1", h)
    h = attn_block(sd, "encoder.mid.attn_1", h)
    h = resnet_block(sd, "encoder.mid.block_2", h)
    return _conv(sd, "encoder.conv_out", _swish(_gn(sd, "encoder.norm_out", h)))


def decoder_forward(sd, cfg: VAEConfig, z):
    """Decoder.forward, model.py:587-630 (incl. the extra GN+swish after
    res_block_lidar1 that the reference flags as a mistake, :617-618)."""
    h = _conv(sd, "decoder.conv_in", z)
    h = resnet_block(sd, "decoder.mid.block_1", h)
    h = attn_block(sd, "decoder.mid.attn_1", h)
    h = resnet_block(sd, "decoder.mid.block_2", h)
    for l in reversed(range(len(cfg.ch_mult))):
        for j in range(cfg.num_res_blocks + 1):
            h = resnet_block(sd, f"decoder.up.{l}.block.{j}", h)
        if l != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up.{l}.upsample.conv", h)
    if cfg.lidar_adapter:
        h = resnet_block(sd, "decoder.res_block_lidar1", h)
        h = _swish(_gn(sd, "decoder.norm_out_lidar1", h))
        h = resnet_block(sd, "decoder.res_block_lidar2", h)
        h = _swish(_gn(sd, "decoder.norm_out_lidar2", h))
        return _conv(sd, "decoder.conv_out_lidar", h)
    return _conv(sd, "decoder.conv_out", _swish(_gn(sd, "decoder.norm_out", h)))


def encode_moments(sd, cfg: VAEConfig, x):
    """AutoencoderKL.encode up to the posterior parameters, autoencoder.py:63-67."""
    return _conv(sd, "quant_conv", encoder_forward(sd, cfg, x))


def posterior_sample(moments, noise):
    """DiagonalGaussianDistribution.__init__/sample with the noise supplied,
    distributions.py:25-37 (the reference draws it from the CPU generator)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def decode(sd, cfg: VAEConfig, z):
    """AutoencoderKL.decode, autoencoder.py:69-72."""
    return decoder_forward(sd, cfg, _conv(sd, "post_quant_conv", z))
