"""CPU: the lidar alignment inside LatentDiffusion.get_input and its inverse in decode_sample (SURVEY.md 8(a) rows A14 /
A15) against tests/golden/plumbing_align.npz, which the REFERENCE's own get_input / decode_sample produced
(tests/golden/make_golden_align.py: centre crop of a WIDE range latent, zero-row padding of a SHORT one, the in-place
re-normalisation of the lidar bbox, camera/lidar interleave order).  This part of the path is index arithmetic on the
host, so it runs without the GPU: the VAE encodes and the conditioning producer are replaced by the same seeded stand-ins
the generator used.  Everything is compared BIT FOR BIT."""
import os
import sys

import pytest
import torch

from tests.golden_cases import load

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _stand_in_cond(cond):
    b = cond["ref_bbox"].shape[0]
    t0 = cond["ref_image"].reshape(b, -1)[:, :48].repeat(1, 16).reshape(b, 1, 768)
    t1 = cond["ref_bbox"].reshape(b, -1).repeat(1, 32).reshape(b, 1, 768)
    return torch.cat([t0, t1], dim=1)


@pytest.fixture(scope="module")
def model():
    from mobi_amd.ldm.models.diffusion.ddpm import LatentDiffusion
    unet = {"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
            "params": dict(image_size=8, in_channels=9, out_channels=4, model_channels=32, attention_resolutions=[1],
                           num_res_blocks=1, channel_mult=[1], num_heads=4, use_spatial_transformer=True,
                           transformer_depth=1, context_dim=768, legacy=False, bbox_cond=True, use_camera=True,
                           use_lidar=True)}
    return LatentDiffusion(cond_stage_config="__is_unconditional__", unet_config=unet, first_stage_key="inpaint",
                           cond_stage_key=["ref_image", "ref_bbox"], image_size=8, channels=4,
                           conditioning_key="crossattn", use_ema=False, use_camera=True, use_lidar=True)


@pytest.mark.parametrize("tag", ["wide", "short"])
def test_get_input_alignment_and_decode_sample(model, tag, monkeypatch):
    g = load("plumbing_align")
    z_image, z_lidar = g[f"{tag}_z_image"], g[f"{tag}_z_lidar_in"]
    monkeypatch.setattr(model, "encode_all_stages", lambda **kw: (z_image.clone(), z_lidar.clone()))
    monkeypatch.setattr(model, "process_conditioning",
                        lambda cond, force_c_encode=False: (_stand_in_cond(cond), cond))
    batch = {"image": {"cond": {"ref_image": g[f"{tag}_ref_image"].clone(), "ref_bbox": g[f"{tag}_bbox_image"].clone()}},
             "lidar": {"cond": {"ref_image": g[f"{tag}_ref_image"].clone(), "ref_bbox": g[f"{tag}_bbox_before"].clone()}}}
    with pytest.warns(UserWarning) if tag == "wide" else _nullcontext():
        data = model.get_input(batch, "inpaint", force_c_encode=True)
    assert torch.equal(data["z"], g[f"{tag}_z"])                       # crop / zero rows + camera, lidar interleave
    assert torch.equal(data["z_lidar"], g[f"{tag}_z_lidar"])
    assert torch.equal(batch["lidar"]["cond"]["ref_bbox"], g[f"{tag}_bbox_after"])    # edited IN PLACE, as the reference
    assert not torch.equal(g[f"{tag}_bbox_after"], g[f"{tag}_bbox_before"])
    assert torch.equal(data["cond"], g[f"{tag}_cond"])                 # the lidar tokens see the edited bbox
    h_cam, h_lid = model.decode_sample(g[f"{tag}_sample"].clone(), data["z_lidar"].clone())
    assert torch.equal(h_cam, g[f"{tag}_h_cam"]) and torch.equal(h_lid, g[f"{tag}_h_lid"])


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
