"""UNetModel / ResBlock / Upsample / Downsample on the gfx950 engine.

Class names, constructor kwargs and `state_dict` keys follow the reference's
ldm/modules/diffusionmodules/openaimodel.py (TimestepEmbedSequential :74-88,
Upsample :91-119, Downsample :134-160, ResBlock :163-275, UNetModel :528-898);
only the configuration MObI instantiates is supported (use_spatial_transformer=True,
no scale-shift norm, conv resampling, no class conditioning).

Data flow differences from the reference graph (results are the same function):
  * activations are channels-last 16-bit tensors between kernels;
  * the skip concat `th.cat([h, hs.pop()], 1)` (:893) is never materialised: the
    following ResBlock's GroupNorm and 1x1 skip conv read both sources;
  * nearest x2 upsampling (:116) is folded into the conv's gather;
  * all 22 `emb_layers` projections run as one skinny GEMV at the top of forward and
    are added per image in the first conv's epilogue.
"""
import torch
import torch.nn as nn

from .... import engine_dtype, ops
from ...._lib import ACT_SILU
from ..attention import SpatialTransformer
from .util import (Conv2d, Marker, conv_nd, enter, leave, linear, normalization, timestep_embedding, zero_module)


class TimestepBlock(nn.Module):
    pass


def _opens_with_groupnorm(layer):
    """The first reader of `layer`'s input is a GroupNorm over all of it (ResBlock.in_layers[0], SpatialTransformer.norm;
    a Sequential: its first layer's)."""
    if isinstance(layer, nn.Sequential):
        return len(layer) > 0 and _opens_with_groupnorm(layer[0])
    return isinstance(layer, (ResBlock, SpatialTransformer))


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    def forward(self, x, emb, context=None, skip=None, then_groupnorm=False):
        """then_groupnorm: the caller's next reader of the result is a GroupNorm (the next block's first layer, or
        UNetModel.out): a convolution that splits k may then hand over its partial sums (ops.Deferred) instead of running a
        reduce launch.  Between the layers of the sequence that is decided here."""
        layers = list(self)
        for i, layer in enumerate(layers):
            defer = _opens_with_groupnorm(layers[i + 1]) if i + 1 < len(layers) else then_groupnorm
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb, skip=skip, defer_out=defer)
                skip = None
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            elif isinstance(layer, Conv2d):
                x = _plain_conv(layer, ops.finished(x))
            elif isinstance(layer, (Upsample, Downsample)):
                x = layer(x, defer_out=defer)
            else:
                x = layer(ops.finished(x))
        return x


def _plain_conv(conv, x):
    """A bare Conv2d inside a TimestepEmbedSequential (input_blocks.0)."""
    if isinstance(x, (list, tuple)) or (x.dtype == torch.float32 and conv.in_channels <= 16):
        srcs = list(x) if isinstance(x, (list, tuple)) else [x]
        packed = ops.pack_sources([s.float().contiguous() for s in srcs], engine_dtype())
        return ops.igemm(packed, conv.packed_thin(), stride=conv.stride, pad=conv.padding)
    x, ext = enter(x)
    return leave(ops.igemm(x, conv.packed(), stride=conv.stride, pad=conv.padding), ext)


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2, "MObI uses conv_resample=True, dims=2"
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)

    def forward(self, x, defer_out=False):
        x, ext = enter(ops.finished(x))
        assert x.shape[3] == self.channels
        defer = "keep" if defer_out and not ext else None
        return leave(ops.igemm(x, self.conv.packed(), upsample=True, pad=self.conv.padding, defer=defer), ext)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2, "MObI uses conv_resample=True, dims=2"
        self.channels = channels
        self.out_channels = out_channels or channels
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)

    def forward(self, x, defer_out=False):
        x, ext = enter(ops.finished(x))
        assert x.shape[3] == self.channels
        defer = "keep" if defer_out and not ext else None
        return leave(ops.igemm(x, self.op.packed(), stride=2, pad=self.op.padding, defer=defer), ext)


class ResBlock(TimestepBlock):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down or use_conv:
            raise NotImplementedError("not used by MObI's UNet configuration")
        self.channels = channels
        self.emb_channels = emb_channels
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(normalization(channels), Marker(),
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(Marker(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), Marker(), Marker(),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = Marker()
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)
        self._emb_slice = None          # set by UNetModel: column range inside the batched projection

    def forward(self, x, emb, skip=None, defer_out=False):
        """x: engine tensor [N,H,W,C0] (or fp32 NCHW, or an ops.Deferred: the partial sums of the split-K launch that
        produces it -- in_layers' GroupNorm sums them); skip: optional second source whose channels
        follow x's (the un-materialised concat); emb: fp32 [N, emb_channels] or a dict holding the
        pre-projected `emb_layers` outputs (UNetModel).  defer_out: the caller's next reader of the result is a GroupNorm
        (TimestepEmbedSequential.forward): the result may then be an ops.Deferred."""
        x, ext = enter(x)
        if skip is not None:
            skip, _ = enter(ops.finished(skip))
        # emb_out already carries conv1's bias (one fp32 add at pack time instead of one per pixel; the launch
        # then needs only ONE per-image vector, which keeps it on the register-epilogue kernels)
        if isinstance(emb, dict):
            lo, hi = self._emb_slice
            emb_out = emb["proj"][:, lo:hi]
        else:
            w, b = self.emb_layers[1].skinny()
            emb_out = ops.skinny_linear(emb.float().contiguous(), w, b + self.in_layers[2].bias.detach().float(),
                                        pre_act=ACT_SILU)
        n1, n2 = self.in_layers[0], self.out_layers[0]

        # in_layers' GroupNorm is x's first reader: a Deferred x is summed there (and written: x has more readers below)
        g, b = n1.affine()
        h1 = ops.groupnorm(x, g, b, n1.eps, silu=True, x2=skip)
        x = ops.finished(x)

        def main_path():
            # conv1's only reader is out_layers' GroupNorm: a split launch leaves its partial sums to it
            h = ops.igemm(h1, self.in_layers[2].packed(), rowvec=emb_out, rowvec_has_bias=True, defer="drop")
            g, b = n2.affine()
            return ops.groupnorm(h, g, b, n2.eps, silu=True)

        if isinstance(self.skip_connection, Conv2d):       # the 1x1 skip conv is independent of the main path
            h, xs = ops.concurrently(main_path, lambda: ops.igemm(x, self.skip_connection.packed(), x2=skip))
        else:
            assert skip is None
            h, xs = main_path(), x
        defer = "keep" if defer_out and not ext else None
        return leave(ops.igemm(h, self.out_layers[3].packed(), residual=xs, defer=defer), ext)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True, add_conv_in_front_of_unet=False, bbox_cond=False,
                 use_camera=True, use_lidar=False):
        super().__init__()
        if not use_spatial_transformer or context_dim is None:
            raise NotImplementedError("the engine implements the use_spatial_transformer=True UNet MObI uses")
        if num_classes is not None or n_embed is not None or resblock_updown or add_conv_in_front_of_unet \
                or use_scale_shift_norm or not conv_resample or dims != 2:
            raise NotImplementedError("option not used by any MObI config")
        if isinstance(context_dim, (list, tuple)):
            context_dim = list(context_dim)[0] if len(context_dim) == 1 else context_dim
        if num_heads == -1 and num_head_channels == -1:
            raise ValueError("Either num_heads or num_head_channels has to be set")
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_res_blocks = out_channels, num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.channel_mult = channel_mult
        self.dtype = torch.float32
        self.use_camera, self.use_lidar = use_camera, use_lidar
        self.multimodal = bool(use_camera and use_lidar)

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, time_embed_dim), Marker(),
                                        linear(time_embed_dim, time_embed_dim))

        def heads_for(ch):
            if num_head_channels == -1:
                return num_heads, ch // num_heads
            return ch // num_head_channels, num_head_channels

        def transformer(ch):
            nh, dh = heads_for(ch)
            return SpatialTransformer(ch, nh, dh, depth=transformer_depth, context_dim=context_dim,
                                      bbox_cond=bbox_cond, multimodal=self.multimodal)

        self.input_blocks = nn.ModuleList(
            [TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        input_block_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, time_embed_dim, dropout, out_channels=mult * model_channels, dims=dims)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(transformer(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims,
                                                                            out_channels=ch)))
                input_block_chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, time_embed_dim, dropout, dims=dims), transformer(ch),
            ResBlock(ch, time_embed_dim, dropout, dims=dims))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = input_block_chans.pop()
                layers = [ResBlock(ch + ich, time_embed_dim, dropout, out_channels=model_channels * mult, dims=dims)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(transformer(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), Marker(),
                                 zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))

        # column ranges of the batched emb_layers projection
        off = 0
        self._res_blocks = [m for m in self.modules() if isinstance(m, ResBlock)]
        for rb in self._res_blocks:
            rb._emb_slice = (off, off + rb.out_channels)
            off += rb.out_channels
        self._emb_total = off

    def _emb_projection(self):
        """(T weight [sum cout, 4*mc], fp32 bias) of all ResBlock emb_layers stacked; the bias also carries the
        bias of the conv the projection is added to (`in_layers[2]`, openaimodel.py:262-270 of the reference:
        `h = in_layers(x); h = h + emb_out`)."""
        lins = [rb.emb_layers[1] for rb in self._res_blocks]
        convs = [rb.in_layers[2] for rb in self._res_blocks]
        key = (engine_dtype(), lins[0].weight.device, tuple(l.weight._version for l in lins),
               tuple(l.bias._version for l in lins), tuple(c.bias._version for c in convs), lins[0].weight.data_ptr())
        c = self.__dict__.setdefault("_embproj_cache", {})
        if c.get("key") != key:
            w = torch.cat([l.weight.detach() for l in lins], dim=0).to(engine_dtype()).contiguous()
            b = torch.cat([l.bias.detach().float() + c.bias.detach().float() for l, c in zip(lins, convs)],
                          dim=0).contiguous()
            c["key"], c["val"] = key, (w, b)
        return c["val"]

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """x: fp32 [N, in_channels, h, w], or a list of fp32 NCHW tensors whose channels
        concatenate to in_channels (the sampler passes [img, inpaint_image, inpaint_mask] so the
        torch.cat of ddim.py:170 is never materialised); timesteps: int64 [N];
        context: fp32 [N, n_ctx, context_dim].  Returns fp32 [N, out_channels, h, w]."""
        assert y is None, "the MObI UNet is not class-conditional"
        t_emb = timestep_embedding(timesteps, self.model_channels)
        w0, b0 = self.time_embed[0].skinny()
        w2, b2 = self.time_embed[2].skinny()
        emb = ops.skinny_linear(ops.skinny_linear(t_emb, w0, b0, post_act=ACT_SILU), w2, b2)
        we, be = self._emb_projection()
        embd = {"emb": emb, "proj": ops.skinny_linear(emb, we, be, pre_act=ACT_SILU)}
        context = context.float().contiguous()

        # (a block whose successor opens with a GroupNorm may return an ops.Deferred: the successor's GroupNorm sums the split-K
        #  slabs and writes the tensor -- which is what `hs` then holds by the time an output block pops it)
        hs = []
        h = x
        blocks = list(self.input_blocks) + [self.middle_block] + list(self.output_blocks)
        nin = len(self.input_blocks)
        for i, module in enumerate(blocks):
            then_gn = _opens_with_groupnorm(blocks[i + 1]) if i + 1 < len(blocks) else True      # self.out[0]
            if i < nin:
                h = module(h, embd, context, then_groupnorm=then_gn)
                hs.append(h)
            elif i == nin:
                h = module(h, embd, context, then_groupnorm=then_gn)
            else:
                h = module(h, embd, context, skip=hs.pop(), then_groupnorm=then_gn)
        n0 = self.out[0]
        g, b = n0.affine()
        h = ops.groupnorm(h, g, b, n0.eps, silu=True)
        return ops.conv_small_cout(h, self.out[2].packed_tap_major(), pad=self.out[2].padding)
