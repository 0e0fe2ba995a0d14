"""AutoencoderKL on the gfx950 engine (reference: ldm/models/autoencoder.py:16-72)."""
import torch
import torch.nn as nn

from ... import engine_dtype, ops
from ..modules.diffusionmodules.model import Decoder, Encoder
from ..modules.diffusionmodules.util import Conv2d
from ..modules.distributions.distributions import DiagonalGaussianDistribution


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None):
        super().__init__()
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        assert ddconfig["double_z"]
        self.quant_conv = Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        if monitor is not None:
            self.monitor = monitor
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=list()):
        """autoencoder.py:52-60 of the reference, plus the key diagnostics its LatentDiffusion loader prints."""
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                print("Deleting key {} from state_dict.".format(k))
                del sd[k]
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
        if len(missing) > 0:
            print(f"Missing Keys: {missing}")
        if len(unexpected) > 0:
            print(f"Unexpected Keys: {unexpected}")
        return missing, unexpected

    @torch.no_grad()
    def encode(self, x):
        h = self.encoder(x)
        w, b = self.quant_conv.packed_f32()
        moments = ops.conv_small_cin([h], w, b, 1, 1, (0, 0), engine_dtype(), out_f32_nchw=True)
        return DiagonalGaussianDistribution(moments)

    @torch.no_grad()
    def decode(self, z, clamp=None):
        w, b = self.post_quant_conv.packed_f32()
        z = ops.conv_small_cin([z.float().contiguous()], w, b, 1, 1, (0, 0), engine_dtype(), out_f32_nchw=True)
        return self.decoder(z, clamp=clamp)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior


class IdentityFirstStage(nn.Module):
    def encode(self, x, *a, **k):
        return x

    def decode(self, x, *a, **k):
        return x
