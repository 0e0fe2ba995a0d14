"""Token mapper used by the conditioning producer (reference: ldm/modules/encoders/xf.py --
Transformer :104-130, ResidualAttentionBlock :78-101, MLP :47-57, QKVMultiheadAttention :60-75).

Host-side PyTorch: this runs once per batch on a single token per image
(0.3 % of the path's FLOPs), not inside the denoising loop, so it is NOT part of the HIP engine; it
exists so that `LatentDiffusion.get_input` is complete.  Same `state_dict` keys as the reference.
"""
import math

import torch
import torch.nn as nn


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return super().forward(x.float()).to(x.dtype)


class QKVMultiheadAttention(nn.Module):
    def __init__(self, n_heads, n_ctx):
        super().__init__()
        self.n_heads, self.n_ctx = n_heads, n_ctx

    def forward(self, qkv):
        bs, n_ctx, width = qkv.shape
        ch = width // self.n_heads // 3
        q, k, v = qkv.view(bs, n_ctx, self.n_heads, 3 * ch).split(ch, dim=-1)
        s = 1.0 / math.sqrt(math.sqrt(ch))
        w = torch.einsum("bthc,bshc->bhts", q * s, k * s)
        w = torch.softmax(w.float(), dim=-1).type(w.dtype)
        return torch.einsum("bhts,bshc->bthc", w, v).reshape(bs, n_ctx, -1)


class MultiheadAttention(nn.Module):
    def __init__(self, n_ctx, width, heads):
        super().__init__()
        self.c_qkv = nn.Linear(width, width * 3)
        self.c_proj = nn.Linear(width, width)
        self.attention = QKVMultiheadAttention(heads, n_ctx)

    def forward(self, x):
        return self.c_proj(self.attention(self.c_qkv(x)))


class MLP(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.c_fc = nn.Linear(width, width * 4)
        self.c_proj = nn.Linear(width * 4, width)
        self.gelu = nn.GELU()

    def forward(self, x):
        return self.c_proj(self.gelu(self.c_fc(x)))


class ResidualAttentionBlock(nn.Module):
    def __init__(self, n_ctx, width, heads):
        super().__init__()
        self.attn = MultiheadAttention(n_ctx, width, heads)
        self.ln_1 = LayerNorm(width)
        self.mlp = MLP(width)
        self.ln_2 = LayerNorm(width)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, n_ctx, width, layers, heads):
        super().__init__()
        self.n_ctx, self.width, self.layers = n_ctx, width, layers
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(n_ctx, width, heads) for _ in range(layers)])

    def forward(self, x):
        for block in self.resblocks:
            x = block(x)
        return x
