"""Seeded random weights / inputs for the lab scripts (bench.py's recipe: N(0, 1 / fan_in) matrices, 1 + 0.1 z norm scales,
0.05 z biases) -- torch generator streams, nothing from oracle/ (which only tests/, smoke() and bench.py's cpu_baseline leg use)."""
import math
import zlib

import torch


def fill_module_(module, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) / math.sqrt(p[0].numel()))
            elif name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    return module


def synth_input(name, shape, seed=0):
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed << 20)) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g)


def synth_weight(name, shape, seed=0):
    w = synth_input(name, shape, seed)
    return w / math.sqrt(max(1, w[0].numel())) if w.dim() >= 2 else w
