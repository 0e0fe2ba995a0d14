"""Conditioning producer: reference image -> CLIP ViT-L/14 pooled token -> 5-layer mapper -> LayerNorm, and the
3-D box -> Fourier features -> MLP token (reference: ldm/modules/encoders/modules.py --
FrozenCLIPImageEmbedder :142-180, BBoxEmbedder :182-215, Embedder / get_embedder :217-266).

Host-side PyTorch-ROCm by design: once per batch, outside the
denoising loop.  The CLIP vision tower is Hugging Face's `CLIPVisionModel`, exactly the class the reference
instantiates; it is built from its CONFIG (no download) and receives its weights from the checkpoint's
`cond_stage_model.transformer.*` keys.  `state_dict` keys equal the reference's.
"""
import torch
import torch.nn as nn

from .xf import LayerNorm, Transformer

# openai/clip-vit-large-patch14, vision tower
CLIP_VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    image_size=224, patch_size=14, projection_dim=768, hidden_act="quick_gelu")


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


def fourier_features(x, num_freqs=4):
    """[x, sin(x f), cos(x f) for f = 2^0 .. 2^(num_freqs-1)] on the last axis (include_input, log sampling)."""
    freqs = 2.0 ** torch.linspace(0.0, num_freqs - 1, steps=num_freqs)
    out = [x]
    for f in freqs:
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, -1)


class BBoxEmbedder(AbstractEncoder):
    def __init__(self, embedder_num_freqs=4, proj_dims=(768, 512, 512, 768)):
        super().__init__()
        self.num_freqs = embedder_num_freqs
        out_dim = 3 * (1 + 2 * embedder_num_freqs)
        self.bbox_proj = nn.Linear(out_dim * 8, proj_dims[0])
        self.second_linear = nn.Sequential(nn.Linear(proj_dims[0], proj_dims[1]), nn.SiLU(),
                                           nn.Linear(proj_dims[1], proj_dims[2]), nn.SiLU(),
                                           nn.Linear(proj_dims[2], proj_dims[3]))

    def forward(self, bbox):
        e = fourier_features(bbox, self.num_freqs).reshape(bbox.shape[0], -1).type_as(self.bbox_proj.weight)
        return self.second_linear(self.bbox_proj(e)).unsqueeze(1)

    def encode(self, cond):
        return {"ref_bbox_token": self(cond["ref_bbox"])}


class FrozenCLIPImageEmbedder(AbstractEncoder):
    def __init__(self, conditions, version="openai/clip-vit-large-patch14", clip_config=None):
        super().__init__()
        if "ref_image" in conditions:
            from transformers import CLIPVisionConfig, CLIPVisionModel
            self.transformer = CLIPVisionModel(CLIPVisionConfig(**(clip_config or CLIP_VIT_L14)))
            width = self.transformer.config.hidden_size
            self.final_ln = LayerNorm(width)
            self.mapper = Transformer(1, width, 5, 1)
        if "ref_bbox" in conditions:
            self.bbox_embedder = BBoxEmbedder()
        self.eval()
        for p in self.parameters():
            p.requires_grad = False
        self._register_load_state_dict_pre_hook(self._remap_clip_keys)

    def _remap_clip_keys(self, state_dict, prefix, *args):
        """The released checkpoint was written with transformers 4.19 (`transformer.vision_model.<...>`); newer
        transformers name the same tensors `transformer.<...>`.  Accept either spelling."""
        if not hasattr(self, "transformer"):
            return
        own_has = any(k.startswith("vision_model.") for k in self.transformer.state_dict().keys())
        old, new = prefix + "transformer.vision_model.", prefix + "transformer."
        for k in list(state_dict.keys()):
            if not own_has and k.startswith(old):
                state_dict[new + k[len(old):]] = state_dict.pop(k)
            elif own_has and k.startswith(new) and not k.startswith(old):
                state_dict[old + k[len(new):]] = state_dict.pop(k)

    def forward(self, image):
        z = self.transformer(pixel_values=image).pooler_output.unsqueeze(1)
        return self.final_ln(self.mapper(z))

    def encode(self, cond):
        ret = {}
        if "ref_image" in cond:
            ret["ref_image_token"] = self(cond["ref_image"])
        if "ref_bbox" in cond:
            ret["ref_bbox_token"] = self.bbox_embedder(cond["ref_bbox"])
        return ret
