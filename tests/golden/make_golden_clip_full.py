#!/usr/bin/env python3
"""tests/golden/cond_producer_full.npz: the REFERENCE's FrozenCLIPImageEmbedder at FULL size (ViT-L/14: 224 x 224, 257
tokens, 24 layers, width 1024, 16 heads x 64) on two seeded images, seeded parameters (oracle.weights.fill_module_, seed 29)
-> the mapped reference token [2, 1, 1024].  Build container only (imports /root/reference).

    python tests/golden/make_golden_clip_full.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                     # noqa: E402  (stubs, reference import, save)
import torch                                                 # noqa: E402
from oracle import weights as W                             # noqa: E402

FULL_CLIP_CFG = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                     image_size=224, patch_size=14, projection_dim=768, hidden_act="quick_gelu")


def main():
    torch.manual_seed(0)
    mg.import_reference()
    import transformers
    import ldm.modules.encoders.modules as em
    orig = transformers.CLIPVisionModel.from_pretrained
    em.CLIPVisionModel.from_pretrained = classmethod(
        lambda cls, *a, **k: transformers.CLIPVisionModel(transformers.CLIPVisionConfig(**FULL_CLIP_CFG)))
    try:
        enc = em.FrozenCLIPImageEmbedder(conditions=["ref_image", "ref_bbox"]).eval()
    finally:
        em.CLIPVisionModel.from_pretrained = orig
    W.fill_module_(enc, seed=29)
    ref_image = W.synth_input("cond.full", (3, 3, 224, 224))[:2].contiguous()
    with torch.no_grad():
        out = enc.encode({"ref_image": ref_image})
    assert out["ref_image_token"].shape == (2, 1, 1024)
    mg.save("cond_producer_full", ref_image_token=out["ref_image_token"])


if __name__ == "__main__":
    main()
