"""Conditioning producer on the engine: reference image -> CLIP ViT-L/14 vision tower -> pooled
token -> five-block token mapper -> LayerNorm, and the 3-D box -> Fourier features -> MLP token
(reference: ldm/modules/encoders/modules.py -- FrozenCLIPImageEmbedder :142-180, BBoxEmbedder :182-215,
get_embedder :217-266; the mapper's blocks are ldm/modules/encoders/xf.py:47-130; the tower is Hugging Face's
CLIPVisionModel, `openai/clip-vit-large-patch14`).

Everything runs in the engine's kernels; nothing here depends on `transformers`:
  tower   patches [B, 256, 588 -> 608] x W^T (mobi_igemm) ; class token + position table ; pre_layrnorm ;
          24 x { layer_norm1 -> [q;k;v] projection (one stacked mobi_igemm, biases in the epilogue) -> mobi_attention
                 (T = 257, 16 heads x 64) -> out_proj (+ residual) ; layer_norm2 -> fc1 -> mobi_quick_gelu -> fc2 (+ residual) } ;
          post_layernorm of the class token
  mapper  ONE token per image, so its attention is a softmax over a single key == 1: the block reduces EXACTLY to
          x += c_proj(v(ln_1(x))) ; x += c_proj(gelu(c_fc(ln_2(x)))) with v = rows [2 W, 3 W) of c_qkv -- fp32 GEMVs
          (mobi_skinny_linear), fp32 row LayerNorm, no [T x T] anything
  box     Fourier features on the host tensor library (216 numbers per object), then four GEMVs with SiLU
`state_dict` keys are the reference's / Hugging Face's (461 tensors); a checkpoint written by transformers 4.19 spells
the tower `transformer.vision_model.*` -- both spellings load.  The pooled CLIP token of an image is cached: the harness
encodes the SAME reference image for the camera and for the lidar branch (ddpm.py:788, :816 of the reference).
"""
import torch
import torch.nn as nn

from .... import engine_dtype, ops
from ...._lib import ACT_GELU, ACT_NONE, ACT_SILU
from ..diffusionmodules.util import LayerNorm, Linear

# openai/clip-vit-large-patch14, vision tower
CLIP_VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    image_size=224, patch_size=14, projection_dim=768, hidden_act="quick_gelu", layer_norm_eps=1e-5)


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


def _rows(norm, x):
    g, b = norm.affine()
    return ops.layernorm_rows_f32(x, g, b, norm.eps)


def _gemv(lin, x, act=ACT_NONE, rows=None):
    """fp32 [m, k] -> fp32 [m, n] through mobi_skinny_linear; `rows` = (lo, hi) selects output rows of the layer."""
    w, b = lin.skinny()
    if rows is not None:
        w, b = w[rows[0]:rows[1]], (None if b is None else b[rows[0]:rows[1]])
    return ops.skinny_linear(x.contiguous(), w.contiguous(), None if b is None else b.contiguous(), post_act=act)


# ---------------------------------------------------------------------------------------------------------------------
# 3-D box token
# ---------------------------------------------------------------------------------------------------------------------
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
        self.bbox_proj = Linear(out_dim * 8, proj_dims[0])
        self.second_linear = nn.Sequential(Linear(proj_dims[0], proj_dims[1]), nn.Identity(),
                                           Linear(proj_dims[1], proj_dims[2]), nn.Identity(),
                                           Linear(proj_dims[2], proj_dims[3]))

    def forward(self, bbox):
        e = fourier_features(bbox.float(), self.num_freqs).reshape(bbox.shape[0], -1)
        h = _gemv(self.bbox_proj, e)
        h = _gemv(self.second_linear[0], h, ACT_SILU)
        h = _gemv(self.second_linear[2], h, ACT_SILU)
        return _gemv(self.second_linear[4], h).unsqueeze(1)

    def encode(self, cond):
        return {"ref_bbox_token": self(cond["ref_bbox"])}


# ---------------------------------------------------------------------------------------------------------------------
# token mapper (one token per image)
# ---------------------------------------------------------------------------------------------------------------------
class _MapperAttention(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.c_qkv = Linear(width, width * 3)
        self.c_proj = Linear(width, width)


class _MapperMLP(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.c_fc = Linear(width, width * 4)
        self.c_proj = Linear(width * 4, width)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, n_ctx, width, heads):
        super().__init__()
        if n_ctx != 1 or heads != 1:
            raise NotImplementedError("MObI's mapper is Transformer(1, 1024, 5, 1): one token, one head")
        self.width = width
        self.attn = _MapperAttention(width)
        self.ln_1 = LayerNorm(width)
        self.mlp = _MapperMLP(width)
        self.ln_2 = LayerNorm(width)

    def forward(self, x):
        """x: fp32 [B, width]."""
        w = self.width
        v = _gemv(self.attn.c_qkv, _rows(self.ln_1, x), rows=(2 * w, 3 * w))      # softmax over one key == 1: out = v
        x = ops.lincomb4([x, _gemv(self.attn.c_proj, v)], [1.0, 1.0])
        h = _gemv(self.mlp.c_fc, _rows(self.ln_2, x), ACT_GELU)
        return ops.lincomb4([x, _gemv(self.mlp.c_proj, h)], [1.0, 1.0])


class Transformer(nn.Module):
    def __init__(self, n_ctx, width, layers, heads):
        super().__init__()
        self.n_ctx, self.width, self.layers = n_ctx, width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(n_ctx, width, heads) for _ in range(layers)])

    def forward(self, x):
        for block in self.resblocks:
            x = block(x)
        return x


# ---------------------------------------------------------------------------------------------------------------------
# CLIP vision tower
# ---------------------------------------------------------------------------------------------------------------------
class _PatchEmbedding(nn.Module):
    def __init__(self, width, patch):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(width, 3, patch, patch).normal_(0, 0.02))


class _Table(nn.Module):
    def __init__(self, n, width):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, width).normal_(0, 0.02))


class _Embeddings(nn.Module):
    def __init__(self, width, patch, image):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.empty(width).normal_(0, 0.02))
        self.patch_embedding = _PatchEmbedding(width, patch)
        self.position_embedding = _Table((image // patch) ** 2 + 1, width)


class _SelfAttention(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.k_proj, self.v_proj = Linear(width, width), Linear(width, width)
        self.q_proj, self.out_proj = Linear(width, width), Linear(width, width)


class _MLP(nn.Module):
    def __init__(self, width, inner):
        super().__init__()
        self.fc1, self.fc2 = Linear(width, inner), Linear(inner, width)


class _EncoderLayer(nn.Module):
    def __init__(self, width, inner, heads, eps):
        super().__init__()
        self.heads = heads
        self.self_attn = _SelfAttention(width)
        self.layer_norm1 = LayerNorm(width, eps)
        self.mlp = _MLP(width, inner)
        self.layer_norm2 = LayerNorm(width, eps)

    def _qkv(self):
        a = self.self_attn
        mods = (a.q_proj, a.k_proj, a.v_proj)
        key = (engine_dtype(),) + tuple(p._version for m in mods for p in (m.weight, m.bias)) + (a.q_proj.weight.data_ptr(),)
        c = self.__dict__.setdefault("_qkv_cache", {})
        if c.get("key") != key:
            w = torch.cat([m.weight.detach() for m in mods], 0)
            b = torch.cat([m.bias.detach() for m in mods], 0)
            c["key"], c["val"] = key, ops.pack_linear(w, b, engine_dtype(), w.device)
        return c["val"]

    def forward(self, x):
        """x: engine tokens [B, T, width]."""
        width = x.shape[2]
        g, b = self.layer_norm1.affine()
        qkv = ops.linear(ops.layernorm(x, g, b, self.layer_norm1.eps), self._qkv())
        a = ops.attention(qkv[..., :width], qkv[..., width:2 * width], qkv[..., 2 * width:], self.heads,
                          (width // self.heads) ** -0.5, v_rows=True)
        x = ops.linear(a, self.self_attn.out_proj.packed(), residual=x)
        g, b = self.layer_norm2.affine()
        h = ops.quick_gelu(ops.linear(ops.layernorm(x, g, b, self.layer_norm2.eps), self.mlp.fc1.packed()))
        return ops.linear(h, self.mlp.fc2.packed(), residual=x)


class _Encoder(nn.Module):
    def __init__(self, width, inner, heads, layers, eps):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(width, inner, heads, eps) for _ in range(layers)])


class CLIPVisionTower(nn.Module):
    """The vision half of CLIP with Hugging Face's parameter names (`embeddings.*`, `pre_layrnorm`,
    `encoder.layers.N.{self_attn.{q,k,v,out}_proj, layer_norm1, mlp.{fc1,fc2}, layer_norm2}`, `post_layernorm`)."""

    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                 image_size=224, patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, **unused):
        super().__init__()
        if hidden_act != "quick_gelu":
            raise NotImplementedError("the CLIP vision tower uses quick_gelu")
        self.patch, self.image, self.width = patch_size, image_size, hidden_size
        self.embeddings = _Embeddings(hidden_size, patch_size, image_size)
        self.pre_layrnorm = LayerNorm(hidden_size, layer_norm_eps)            # (sic: the checkpoint's spelling)
        self.encoder = _Encoder(hidden_size, intermediate_size, num_attention_heads, num_hidden_layers, layer_norm_eps)
        self.post_layernorm = LayerNorm(hidden_size, layer_norm_eps)

    def _patch_weight(self):
        w = self.embeddings.patch_embedding.weight
        key = (engine_dtype(), w._version, w.data_ptr())
        c = self.__dict__.setdefault("_patch_cache", {})
        if c.get("key") != key:
            k = w[0].numel()
            kp = (k + 31) // 32 * 32                                          # the matrix kernel wants k % 32 == 0
            wp = torch.zeros((w.shape[0], kp), device=w.device, dtype=torch.float32)
            wp[:, :k] = w.detach().float().reshape(w.shape[0], k)
            c["key"], c["val"] = key, (ops.pack_linear(wp, None, engine_dtype(), w.device), k, kp)
        return c["val"]

    def pooled(self, pixel_values):
        """fp32 [B, 3, S, S] -> fp32 [B, width]: post_layernorm of the class token (CLIPVisionModel.pooler_output)."""
        b, _, s, _ = pixel_values.shape
        p, n = self.patch, s // self.patch
        pw, k, kp = self._patch_weight()
        # patch matrix [B, n*n, 3*p*p] (an index permutation of the image), zero-padded on k, in the storage type
        patches = pixel_values.float().reshape(b, 3, n, p, n, p).permute(0, 2, 4, 1, 3, 5).reshape(b, n * n, k)
        pm = torch.zeros((b, n * n, kp), device=pixel_values.device, dtype=engine_dtype())
        pm[..., :k] = patches
        tok = ops.linear(pm, pw)                                              # [B, n*n, width]
        e = self.embeddings
        x = torch.cat([e.class_embedding.detach().to(tok.dtype).expand(b, 1, -1), tok], dim=1)
        x = (x.float() + e.position_embedding.weight.detach().float()[: n * n + 1]).to(tok.dtype).contiguous()
        g, bb = self.pre_layrnorm.affine()
        x = ops.layernorm(x, g, bb, self.pre_layrnorm.eps)
        for layer in self.encoder.layers:
            x = layer(x)
        cls = x[:, 0].float().contiguous()                                    # [B, width]
        return _rows(self.post_layernorm, cls)


class FrozenCLIPImageEmbedder(AbstractEncoder):
    def __init__(self, conditions, version="openai/clip-vit-large-patch14", clip_config=None):
        super().__init__()
        if "ref_image" in conditions:
            self.transformer = CLIPVisionTower(**(clip_config or CLIP_VIT_L14))
            width = self.transformer.width
            self.final_ln = LayerNorm(width)
            self.mapper = Transformer(1, width, 5, 1)
        if "ref_bbox" in conditions:
            self.bbox_embedder = BBoxEmbedder()
        self.eval()
        for p in self.parameters():
            p.requires_grad = False
        self._register_load_state_dict_pre_hook(self._remap_clip_keys)

    @staticmethod
    def _remap_clip_keys(state_dict, prefix, *args):
        """transformers 4.19 (the released checkpoint) spells the tower `transformer.vision_model.<...>` and stores a
        `position_ids` buffer; accept that spelling."""
        old, new = prefix + "transformer.vision_model.", prefix + "transformer."
        for k in list(state_dict.keys()):
            if k.startswith(old):
                v = state_dict.pop(k)
                if not k.endswith("position_ids"):
                    state_dict[new + k[len(old):]] = v
            elif k.startswith(new) and k.endswith("position_ids"):
                state_dict.pop(k)

    def forward(self, image):
        # the same reference image is encoded for the camera and the lidar branch: one tower pass per distinct image
        # (the harness hands the camera branch and the lidar branch two tensors with the same pixels, ddpm.py:788,816).
        # The entry is tied to the weights it was computed with (load_state_dict / .to() bump the epoch, in-place edits
        # the parameters' version counters) and the storage type.
        from ..diffusionmodules.util import WEIGHTS_EPOCH
        wkey = (WEIGHTS_EPOCH[0], sum(p._version for p in self.transformer.parameters()), engine_dtype(), image.device)
        c = self.__dict__.setdefault("_pooled_cache", {})
        hit = c.get("image")
        # always a pixel compare (one device compare + read-back per call, i.e. per batch): the tensor's identity / version
        # counter would miss writes that do not bump `_version` -- this repo's own engine ops write through raw pointers
        same = hit is not None and c.get("wkey") == wkey and hit.shape == image.shape and hit.device == image.device \
            and torch.equal(hit, image)
        if not same:
            c["image"], c["pooled"], c["wkey"] = image.detach().clone(), self.transformer.pooled(image), wkey
        z = self.mapper(c["pooled"])
        return _rows(self.final_ln, z).unsqueeze(1)

    def encode(self, cond):
        ret = {}
        if "ref_image" in cond:
            ret["ref_image_token"] = self(cond["ref_image"])
        if "ref_bbox" in cond:
            ret["ref_bbox_token"] = self.bbox_embedder(cond["ref_bbox"])
        return ret
