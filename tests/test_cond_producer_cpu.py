"""CPU: the conditioning producer (host PyTorch) against the reference's
FrozenCLIPImageEmbedder / BBoxEmbedder outputs (tests/golden/cond_producer.npz), and through
LatentDiffusion.get_learned_conditioning's token layout."""
import torch

from oracle import weights as W
from tests.golden_cases import load, rel_l2

CLIP_CFG = dict(hidden_size=1024, intermediate_size=256, num_hidden_layers=1, num_attention_heads=16,
                image_size=28, patch_size=14, projection_dim=64, hidden_act="quick_gelu")


def test_cond_producer_matches_reference():
    from mobi_amd.ldm.modules.encoders.modules import FrozenCLIPImageEmbedder
    g = load("cond_producer")
    enc = FrozenCLIPImageEmbedder(["ref_image", "ref_bbox"], clip_config=CLIP_CFG)
    W.fill_module_(enc, seed=13)
    with torch.no_grad():
        out = enc.encode({"ref_image": g["ref_image"], "ref_bbox": g["ref_bbox"]})
    assert out["ref_image_token"].shape == (2, 1, 1024) and out["ref_bbox_token"].shape == (2, 1, 768)
    assert rel_l2(out["ref_image_token"], g["ref_image_token"]) < 2e-5
    assert rel_l2(out["ref_bbox_token"], g["ref_bbox_token"]) < 2e-5


def test_full_size_cond_stage_key_layout():
    """ViT-L/14 tower + mapper + bbox MLP: the 461 `cond_stage_model.*` tensors of the released checkpoint
     -- parameter names follow Hugging Face's CLIPVisionModel and the reference's modules."""
    from mobi_amd.ldm.modules.encoders.modules import FrozenCLIPImageEmbedder
    enc = FrozenCLIPImageEmbedder(["ref_image", "ref_bbox"])
    keys = list(enc.state_dict().keys())
    assert len(keys) == 461                      # cond_stage_model.* holds 461 tensors
    assert any(k.endswith("encoder.layers.23.mlp.fc2.weight") for k in keys)
    # a checkpoint written with transformers 4.19 spells the tower `transformer.vision_model.*`: both load
    sd = enc.state_dict()
    old_style = {(k.replace("transformer.", "transformer.vision_model.", 1)
                  if k.startswith("transformer.") and not k.startswith("transformer.vision_model.") else k): v
                 for k, v in sd.items()}
    missing, unexpected = enc.load_state_dict(old_style, strict=False)
    assert not missing and not unexpected
    assert "mapper.resblocks.4.attn.c_qkv.weight" in keys and "final_ln.weight" in keys
    assert "bbox_embedder.second_linear.4.bias" in keys and "bbox_embedder.bbox_proj.weight" in keys
