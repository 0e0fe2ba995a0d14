"""VAE Encoder / Decoder (incl. MObI's lidar adapter) on the gfx950 engine.

Names and `state_dict` keys follow the reference's
ldm/modules/diffusionmodules/model.py: Normalize :38, Upsample :42-57, Downsample
:60-79, ResnetBlock :82-141, AttnBlock :151-202, Encoder :368-489, Decoder :492-630.
Only what MObI's configs instantiate is built (attn_resolutions [], vanilla
mid attention, conv resampling, temb_channels 0).
"""
import os

import torch
import torch.nn as nn

from .... import engine_dtype, ops
from ...._lib import OUT_ROWS_F32, OUT_TRANSPOSED
from .util import Conv2d, GroupNorm32, enter, leave


# The decoder's residual trunk in fp32: every `x + h` of Decoder.forward accumulates in an fp32 tensor and the 16-bit copy the
# next GroupNorm / convolution reads is made from it, so the trunk is rounded once per consumer instead of re-rounded at
# every block.  End to end at production width (tests/test_gpu_production.py::test_end_to_end_pixel_space, fp16): camera
# picture 1.26-1.37e-3 -> 0.91-0.98e-3 rel-L2 (inside the north star's 1e-3), range view 1.69-1.82e-3 -> 1.24-1.37e-3; bf16
# 1.0-1.4e-2 -> 0.7-1.0e-2.  Costs one elementwise pass per block (mobi_trunk_add; profiles/r04_vae_trunk.txt).
# Default: on with fp16 storage (the storage type chosen for parity), off with bf16 (the throughput configuration);
# MOBI_VAE_FP32_TRUNK=1 / 0 forces it.
_TRUNK_ENV = os.environ.get("MOBI_VAE_FP32_TRUNK", "")


def fp32_trunk():
    if _TRUNK_ENV in ("0", "1"):
        return _TRUNK_ENV == "1"
    return engine_dtype() == torch.float16


# The lidar decoder's TAIL (model.py:612-623 of the reference: two 1 x 5 ResnetBlocks, two GroupNorm + swish, the 1 x 5 output
# convolution -- what the range view has on top of the camera decoder) with its activations to ~22 bits: every GroupNorm reads
# the fp32 stream and writes a hi | lo pair of the storage type, every convolution multiplies the pair by duplicated weights (k
# doubles) and writes fp32.  Superseded as the default by precise_level() below (every convolution of both decoders); what the tail
# ALONE buys is still measurable: MOBI_VAE_PRECISE=0 MOBI_VAE_PRECISE_TAIL=1 (tests/decoder_err.py; numbers in DESIGN.md section 5).
_TAIL_ENV = os.environ.get("MOBI_VAE_PRECISE_TAIL", "")


def precise_tail():
    if _TAIL_ENV in ("0", "1"):
        return _TAIL_ENV == "1"
    return fp32_trunk()


# fp32 STREAMS (round 5): with the trunk in fp32 a ResnetBlock still rounded five tensors to the storage type -- the trunk's copy its
# first GroupNorm reads, both GroupNorm outputs, conv1's output (read by the second GroupNorm) and conv2's output (the increment
# added to the trunk).  Only the two GroupNorm outputs HAVE to be 16-bit (they are the convolutions' matrix-core operands): the
# GroupNorms read fp32 (mobi_groupnorm_params.src_f32), the convolutions write fp32 (MOBI_OUT_ROWS_F32), the trunk update is an fp32
# add (mobi_lincomb4) -- two roundings per block instead of five, no extra matrix work, twice the bytes on the fp32 tensors.
# On with the fp32 trunk; MOBI_VAE_FP32_STREAMS=0 / 1 forces it.
_STREAMS_ENV = os.environ.get("MOBI_VAE_FP32_STREAMS", "")


def fp32_streams():
    if _STREAMS_ENV in ("0", "1"):
        return _STREAMS_ENV == "1"
    return fp32_trunk()


# PRECISE DECODERS (round 5): with fp32 streams what is left of a decoder's error are the roundings of the convolutions' 16-bit
# OPERANDS -- the activations (GroupNorm outputs, the upsampling convolutions' and 1 x 1 shortcuts' inputs, the latent) and the
# weights, about half each (tests/decoder_err.py: range view 8.8e-4 -> 6.2e-4 with the activations split, -> 1.8e-4 with the
# weights split too).  Level 1: every convolution multiplies hi | lo activation pairs by duplicated weights [W ; W] (k doubles).
# Level 2: hi | lo | hi against [W ; W ; W - T(W)] (k triples): T(W) hi + T(W) lo + (W - T(W)) hi -- both operands to ~22 bits, the
# lo x lo term is below fp32's own rounding.  No kernel knows about it: the GroupNorm / split launches write the operand form
# (mobi_groupnorm out_mode 1 / 3, mobi_split_f32), the implicit GEMM sees a wider k.  Both decoders; fp16 end to end (DDIM): camera
# picture 8.5e-4 -> 3.7e-4, range view 1.13e-3 -> 7.0e-4; a decode of 8 images at 512 x 512 takes 75 / 86 ms
# instead of 37 / 46 (camera / lidar).  Default: level 2 with the fp32 streams (fp16
# storage), 0 with bf16; MOBI_VAE_PRECISE=0 / 1 / 2 forces it.
_PRECISE_ENV = os.environ.get("MOBI_VAE_PRECISE", "")


def precise_level():
    if not (fp32_trunk() and fp32_streams()):    # the split operands are read from / written beside the fp32 streams
        return 0
    return int(_PRECISE_ENV) if _PRECISE_ENV in ("0", "1", "2") else 2


def _gn_split(norm, x32, level, silu=True):
    """GroupNorm (+ swish) of an fp32 stream -> the hi | lo (| hi) operand of a convolution with _w_split weights."""
    g, b = norm.affine()
    return ops.groupnorm(x32, g, b, norm.eps, silu=silu, out_mode=ops.GN_OUT_SPLIT3 if level == 2 else ops.GN_OUT_SPLIT,
                         dtype=engine_dtype())


def _split(t32, level):
    return ops.split_f32(t32, engine_dtype(), 3 if level == 2 else 2)


def _w_split(conv, level):
    return conv.packed_dup3() if level == 2 else conv.packed_dup()


def _gn32(norm, x32, silu=True):
    """GroupNorm (+ swish) of an fp32 stream -> the storage type (a convolution's operand)."""
    g, b = norm.affine()
    return ops.groupnorm(x32, g, b, norm.eps, silu=silu, dtype=engine_dtype())


def _gn_pair(norm, x32, silu=True):
    g, b = norm.affine()
    return ops.groupnorm(x32, g, b, norm.eps, silu=silu, out_mode=ops.GN_OUT_SPLIT, dtype=engine_dtype())


def Normalize(in_channels, num_groups=32):
    return GroupNorm32(num_groups, in_channels, eps=1e-6)


def _gn_swish(norm, x, silu=True):
    g, b = norm.affine()
    return ops.groupnorm(x, g, b, norm.eps, silu=silu)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        return ops.igemm(x, self.conv.packed(), upsample=True, pad=(1, 1))      # nearest x2 folded into the gather


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x):
        # F.pad(x, (0,1,0,1)) + stride-2 conv (model.py:72-76): zero rows/cols past the bottom/right
        # edge come from the gather's bounds check, no padded copy
        n, h, w, c = x.shape
        return ops.igemm(x, self.conv.packed(), stride=2, pad=(0, 0), hout=(h + 1 - 3) // 2 + 1,
                         wout=(w + 1 - 3) // 2 + 1)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 kernel_size=3, padding=1):
        super().__init__()
        assert temb_channels == 0 and not conv_shortcut, "MObI's VAEs use temb_ch=0, nin shortcuts"
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=1, padding=padding)
        self.norm2 = Normalize(out_channels)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=kernel_size, stride=1, padding=padding)
        if in_channels != out_channels:
            self.nin_shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, temb=None):
        x, ext = enter(x)
        h = ops.igemm(_gn_swish(self.norm1, x), self.conv1.packed(), pad=self.conv1.padding)
        h = _gn_swish(self.norm2, h)
        xs = ops.igemm(x, self.nin_shortcut.packed()) if self.in_channels != self.out_channels else x
        return leave(ops.igemm(h, self.conv2.packed(), pad=self.conv2.padding, residual=xs), ext)

    def forward_trunk(self, x, trunk):
        """The same block on an fp32 trunk: x = the trunk's 16-bit copy -> (x', trunk')."""
        h = ops.igemm(_gn_swish(self.norm1, x), self.conv1.packed(), pad=self.conv1.padding)
        h = ops.igemm(_gn_swish(self.norm2, h), self.conv2.packed(), pad=self.conv2.padding)
        if self.in_channels != self.out_channels:
            trunk = ops.igemm(x, self.nin_shortcut.packed(), out_mode=OUT_ROWS_F32)
        return ops.trunk_add(trunk, h, x.dtype), trunk

    def forward_stream(self, t32):
        """The block on an fp32 stream (see fp32_streams): fp32 [N,H,W,Cin] -> fp32 [N,H,W,Cout]."""
        h32 = ops.igemm(_gn32(self.norm1, t32), self.conv1.packed(), pad=self.conv1.padding, out_mode=OUT_ROWS_F32)
        d32 = ops.igemm(_gn32(self.norm2, h32), self.conv2.packed(), pad=self.conv2.padding, out_mode=OUT_ROWS_F32)
        if self.in_channels != self.out_channels:
            t32 = ops.igemm(ops.trunk_add(t32, None, engine_dtype()), self.nin_shortcut.packed(), out_mode=OUT_ROWS_F32)
        return ops.lincomb4([t32, d32], [1.0, 1.0])

    def forward_precise(self, t32, level=1):
        """The block on an fp32 stream, the convolutions' operands split (see precise_level): fp32 [N,H,W,Cin] -> fp32 [N,H,W,Cout]."""
        h32 = ops.igemm(_gn_split(self.norm1, t32, level), _w_split(self.conv1, level), pad=self.conv1.padding, out_mode=OUT_ROWS_F32)
        d32 = ops.igemm(_gn_split(self.norm2, h32, level), _w_split(self.conv2, level), pad=self.conv2.padding, out_mode=OUT_ROWS_F32)
        if self.in_channels != self.out_channels:
            t32 = ops.igemm(_split(t32, level), _w_split(self.nin_shortcut, level), out_mode=OUT_ROWS_F32)
        return ops.lincomb4([t32, d32], [1.0, 1.0])


class AttnBlock(nn.Module):
    """Single-head attention over all h*w positions (model.py:178-202).  c = 512 does not fit the
    fused kernel's register budget, so it runs as matrix-core GEMMs with the scores in fp32:
    S = q k^T / sqrt(c) (per-image weights = k), row softmax, O = P v (weights = v^T)."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = Conv2d(in_channels, in_channels, kernel_size=1)

    def forward(self, x, trunk=None):
        x, ext = enter(x)
        n, h, w, c = x.shape
        t = h * w
        hn = _gn_swish(self.norm, x, silu=False)
        q = ops.igemm(hn, self.q.packed()).view(n, t, 1, c)
        k = ops.igemm(hn, self.k.packed()).view(n, t, c)
        vt = ops.igemm(hn, self.v.packed(), out_mode=OUT_TRANSPOSED)                      # [n, c, t]
        kw = ops.Packed(k, None, 1, 1, c, t, t)
        s = ops.igemm(q, kw, weight_per_image=True, w_group_stride=t * c, out_mode=OUT_ROWS_F32,
                      scale=float(int(c) ** (-0.5)))                                      # [n, t, 1, t] fp32
        p = ops.softmax_rows(s.view(n * t, t), x.dtype).view(n, t, 1, t)
        vw = ops.Packed(vt, None, 1, 1, t, c, c)
        o = ops.igemm(p, vw, weight_per_image=True, w_group_stride=c * t).view(n, h, w, c)
        if trunk is not None:
            return ops.trunk_add(trunk, ops.igemm(o, self.proj_out.packed()), x.dtype), trunk
        return leave(ops.igemm(o, self.proj_out.packed(), residual=x), ext)

    def forward_stream(self, t32):
        """The block on an fp32 stream: fp32 [N,H,W,C] -> fp32 (GroupNorm reads fp32, proj_out writes fp32, fp32 add)."""
        n, h, w, c = t32.shape
        t = h * w
        dt = engine_dtype()
        hn = _gn32(self.norm, t32, silu=False)
        q = ops.igemm(hn, self.q.packed()).view(n, t, 1, c)
        k = ops.igemm(hn, self.k.packed()).view(n, t, c)
        vt = ops.igemm(hn, self.v.packed(), out_mode=OUT_TRANSPOSED)
        kw = ops.Packed(k, None, 1, 1, c, t, t)
        s = ops.igemm(q, kw, weight_per_image=True, w_group_stride=t * c, out_mode=OUT_ROWS_F32, scale=float(int(c) ** (-0.5)))
        p = ops.softmax_rows(s.view(n * t, t), dt).view(n, t, 1, t)
        vw = ops.Packed(vt, None, 1, 1, t, c, c)
        o = ops.igemm(p, vw, weight_per_image=True, w_group_stride=c * t).view(n, h, w, c)
        return ops.lincomb4([t32, ops.igemm(o, self.proj_out.packed(), out_mode=OUT_ROWS_F32)], [1.0, 1.0])


def make_attn(in_channels, attn_type="vanilla"):
    assert attn_type == "vanilla"
    return AttnBlock(in_channels)


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, lidar_adapter=False,
                 dropout=0.0, resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True,
                 use_linear_attn=False, attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        assert not use_linear_attn and len(attn_resolutions) == 0
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.lidar_adapter = resolution, in_channels, lidar_adapter
        if lidar_adapter:
            self.conv_in_lidar = Conv2d(in_channels, ch, kernel_size=(1, 5), stride=1, padding=(0, 2))
            self.res_block_lidar1 = ResnetBlock(in_channels=ch, out_channels=ch, temb_channels=0,
                                                kernel_size=(1, 5), padding=(0, 2), dropout=dropout)
            self.res_block_lidar2 = ResnetBlock(in_channels=ch, out_channels=ch, temb_channels=0,
                                                kernel_size=(1, 5), padding=(0, 2), dropout=dropout)
        else:
            self.conv_in = Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0,
                                         dropout=dropout))
                block_in = block_out
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1,
                               padding=1)

    def forward(self, x):
        """x: fp32 NCHW image / range view -> fp32 NCHW [B, 2*z, h, w]."""
        cin = self.conv_in_lidar if self.lidar_adapter else self.conv_in
        h = ops.igemm(ops.pack_sources([x.float().contiguous()], engine_dtype()), cin.packed_thin(), pad=cin.padding)
        if self.lidar_adapter:
            h = self.res_block_lidar2(self.res_block_lidar1(h))
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block](h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        h = _gn_swish(self.norm_out, h)
        return ops.conv_small_cout(h, self.conv_out.packed_tap_major(), pad=self.conv_out.padding)


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, lidar_adapter=False,
                 dropout=0.0, resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False,
                 tanh_out=False, use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        assert not use_linear_attn and len(attn_resolutions) == 0 and not give_pre_end and not tanh_out
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.lidar_adapter = resolution, in_channels, lidar_adapter
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0,
                                         dropout=dropout))
                block_in = block_out
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
        if lidar_adapter:
            self.res_block_lidar1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0,
                                                kernel_size=(1, 5), padding=(0, 2), dropout=dropout)
            self.norm_out_lidar1 = Normalize(block_in)
            self.res_block_lidar2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0,
                                                kernel_size=(1, 5), padding=(0, 2), dropout=dropout)
            self.norm_out_lidar2 = Normalize(block_in)
            self.conv_out_lidar = Conv2d(block_in, out_ch, kernel_size=(1, 5), stride=1, padding=(0, 2))
        else:
            self.norm_out = Normalize(block_in)
            self.conv_out = Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, z, clamp=None):
        """z: fp32 NCHW latent -> fp32 NCHW image; `clamp=(lo, hi)` fuses the torch.clamp the
        harness applies to every decode (ddpm.py:1476,1504)."""
        zin = ops.pack_sources([z.float().contiguous()], engine_dtype())
        if fp32_trunk() and fp32_streams():
            dt = engine_dtype()
            lvl = precise_level()
            rb = (lambda blk, t: blk.forward_precise(t, lvl)) if lvl else (lambda blk, t: blk.forward_stream(t))
            if lvl:
                # the latent as hi | lo (| hi) channels of the thin first convolution (4 -> 8 | 12 of its 32 padded input channels)
                zf = z.float().contiguous()
                zhi = zf.to(dt).float()
                ci = self.conv_in

                def thin_split():
                    w = ci.weight.detach().float()
                    parts = [w, w] + ([w - w.to(dt).float()] if lvl == 2 else [])
                    return ops.pack_conv_padded_cin(torch.cat(parts, dim=1), ci.bias, dt, ci.weight.device)
                t32 = ops.igemm(ops.pack_sources([zhi, zf - zhi] + ([zhi] if lvl == 2 else []), dt), ci._cached(f"thin_split{lvl}", thin_split),
                                pad=(1, 1), out_mode=OUT_ROWS_F32)
            else:
                t32 = ops.igemm(zin, self.conv_in.packed_thin(), pad=(1, 1), out_mode=OUT_ROWS_F32)
            t32 = rb(self.mid.block_2, self.mid.attn_1.forward_stream(rb(self.mid.block_1, t32)))
            for i_level in reversed(range(self.num_resolutions)):
                for i_block in range(self.num_res_blocks + 1):
                    t32 = rb(self.up[i_level].block[i_block], t32)
                if i_level != 0:
                    up = self.up[i_level].upsample
                    if lvl:
                        t32 = ops.igemm(_split(t32, lvl), _w_split(up.conv, lvl), upsample=True, pad=(1, 1), out_mode=OUT_ROWS_F32)
                    else:
                        t32 = ops.igemm(ops.trunk_add(t32, None, dt), up.conv.packed(), upsample=True, pad=(1, 1), out_mode=OUT_ROWS_F32)
            tail = lvl or (1 if precise_tail() else 0)            # the adapter's tail (model.py:612-623) alone: MOBI_VAE_PRECISE_TAIL
            if self.lidar_adapter:
                rbt = (lambda blk, t: blk.forward_precise(t, tail)) if tail else (lambda blk, t: blk.forward_stream(t))
                t32 = rbt(self.res_block_lidar1, t32)
                g, b = self.norm_out_lidar1.affine()                       # (a new fp32 stream -- model.py:617-618)
                u32 = ops.groupnorm(t32, g, b, self.norm_out_lidar1.eps, silu=True, out_mode=ops.GN_OUT_F32, dtype=dt)
                u32 = rbt(self.res_block_lidar2, u32)
                norm, cout = self.norm_out_lidar2, self.conv_out_lidar
            else:
                u32, norm, cout = t32, self.norm_out, self.conv_out
            if tail:                                               # (the small-cout kernel: hi | lo pairs at every level)
                return ops.conv_small_cout(_gn_pair(norm, u32), cout.packed_dup(), pad=cout.padding, clamp=clamp)
            return ops.conv_small_cout(_gn32(norm, u32), cout.packed_tap_major(), pad=cout.padding, clamp=clamp)
        if fp32_trunk():
            dt = engine_dtype()
            t32 = ops.igemm(zin, self.conv_in.packed_thin(), pad=(1, 1), out_mode=OUT_ROWS_F32)
            h = ops.trunk_add(t32, None, dt)
            h, t32 = self.mid.block_1.forward_trunk(h, t32)
            h, t32 = self.mid.attn_1(h, trunk=t32)
            h, t32 = self.mid.block_2.forward_trunk(h, t32)
            for i_level in reversed(range(self.num_resolutions)):
                for i_block in range(self.num_res_blocks + 1):
                    h, t32 = self.up[i_level].block[i_block].forward_trunk(h, t32)
                if i_level != 0:
                    up = self.up[i_level].upsample
                    t32 = ops.igemm(h, up.conv.packed(), upsample=True, pad=(1, 1), out_mode=OUT_ROWS_F32)
                    h = ops.trunk_add(t32, None, dt)
            if self.lidar_adapter and precise_tail():
                t32 = self.res_block_lidar1.forward_precise(t32)
                g, b = self.norm_out_lidar1.affine()                       # (a new fp32 stream -- model.py:617-618)
                u32 = ops.groupnorm(t32, g, b, self.norm_out_lidar1.eps, silu=True, out_mode=ops.GN_OUT_F32, dtype=dt)
                u32 = self.res_block_lidar2.forward_precise(u32)
                cout = self.conv_out_lidar
                return ops.conv_small_cout(_gn_pair(self.norm_out_lidar2, u32), cout.packed_dup(), pad=cout.padding, clamp=clamp)
            if self.lidar_adapter:
                h, t32 = self.res_block_lidar1.forward_trunk(h, t32)
                # (GroupNorm + swish of the adapter: a new stream, not a residual update -- model.py:617-618)
                h = _gn_swish(self.norm_out_lidar1, h)
                h = _gn_swish(self.norm_out_lidar2, self.res_block_lidar2(h))
                cout = self.conv_out_lidar
            else:
                h = _gn_swish(self.norm_out, h)
                cout = self.conv_out
            return ops.conv_small_cout(h, cout.packed_tap_major(), pad=cout.padding, clamp=clamp)
        h = ops.igemm(zin, self.conv_in.packed_thin(), pad=(1, 1))
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block](h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        if self.lidar_adapter:
            h = _gn_swish(self.norm_out_lidar1, self.res_block_lidar1(h))    # reference keeps this extra GN+swish (:617-618)
            h = _gn_swish(self.norm_out_lidar2, self.res_block_lidar2(h))
            cout = self.conv_out_lidar
        else:
            h = _gn_swish(self.norm_out, h)
            cout = self.conv_out
        return ops.conv_small_cout(h, cout.packed_tap_major(), pad=cout.padding, clamp=clamp)
