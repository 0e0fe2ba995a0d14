"""CPU: structure of the conditioning producer -- parameter names / count of the released
checkpoint's `cond_stage_model.*`, both spellings of the CLIP tower's keys, no dependency on `transformers`.  Its arithmetic
runs on the engine: tests/test_gpu_cond_producer.py compares it with the reference's outputs."""
import sys

import torch


def test_full_size_cond_stage_key_layout():
    """ViT-L/14 tower + mapper + bbox MLP: the 461 `cond_stage_model.*` tensors of the released checkpoint
     -- parameter names follow Hugging Face's CLIPVisionModel and the reference's modules."""
    from mobi_amd.ldm.modules.encoders.modules import FrozenCLIPImageEmbedder
    enc = FrozenCLIPImageEmbedder(["ref_image", "ref_bbox"])
    keys = list(enc.state_dict().keys())
    assert len(keys) == 461                      # cond_stage_model.* holds 461 tensors
    assert any(k.endswith("encoder.layers.23.mlp.fc2.weight") for k in keys)
    assert "transformer.embeddings.patch_embedding.weight" in keys and "transformer.pre_layrnorm.weight" in keys
    # a checkpoint written with transformers 4.19 spells the tower `transformer.vision_model.*` (+ a position_ids buffer)
    sd = enc.state_dict()
    old_style = {(k.replace("transformer.", "transformer.vision_model.", 1) if k.startswith("transformer.") else k): v
                 for k, v in sd.items()}
    old_style["transformer.vision_model.embeddings.position_ids"] = torch.arange(257)[None]
    missing, unexpected = enc.load_state_dict(old_style, strict=False)
    assert not missing and not unexpected
    missing, unexpected = enc.load_state_dict(sd, strict=True)
    assert "mapper.resblocks.4.attn.c_qkv.weight" in keys and "final_ln.weight" in keys
    assert "bbox_embedder.second_linear.4.bias" in keys and "bbox_embedder.bbox_proj.weight" in keys
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    assert shapes["transformer.embeddings.patch_embedding.weight"] == (1024, 3, 14, 14)
    assert shapes["transformer.embeddings.position_embedding.weight"] == (257, 1024)
    assert shapes["mapper.resblocks.0.attn.c_qkv.weight"] == (3072, 1024)
    assert shapes["bbox_embedder.bbox_proj.weight"] == (768, 216)


def test_producer_does_not_need_transformers():
    import importlib
    saved = {k: v for k, v in sys.modules.items() if k == "transformers" or k.startswith("transformers.")}
    for k in saved:
        del sys.modules[k]
    sys.modules["transformers"] = None               # any `import transformers` now raises ImportError
    try:
        import mobi_amd.ldm.modules.encoders.modules as m
        importlib.reload(m)
        m.FrozenCLIPImageEmbedder(["ref_image", "ref_bbox"], clip_config=dict(hidden_size=64, intermediate_size=64,
                                                                              num_hidden_layers=1, num_attention_heads=4,
                                                                              image_size=28, patch_size=14))
    finally:
        del sys.modules["transformers"]
        sys.modules.update(saved)
