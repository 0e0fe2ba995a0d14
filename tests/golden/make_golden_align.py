#!/usr/bin/env python3
"""tests/golden/plumbing_align.npz: the lidar alignment of LatentDiffusion.get_input (ddpm.py:798-815 of the
reference: centre crop of the range latent to `image_size` columns, zero padding of missing rows, the IN-PLACE
re-normalisation of the lidar bbox) and its inverse in decode_sample (ddpm.py:1420-1447), produced by the REFERENCE's
own `get_input` / `decode_sample` run on CPU.

`encode_all_stages` of the reference can only build SQUARE range latents (its mask is resized to
`size=z.shape[-1]`, a square, and concatenated: ddpm.py:1030-1031), so the non-square cases the alignment code is
written for cannot come out of the VAE path; here `encode_all_stages` and `process_conditioning` of the reference object
are replaced by seeded stand-ins and everything between them -- the code under test -- is the reference's.
Cases: a WIDE latent (8 x 16 -> columns 4..12) and a SHORT one (4 x 8 -> two zero rows above and below).

    python tests/golden/make_golden_align.py          (build container only: imports /root/reference)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (stubs + reference import + save)
from oracle import weights as W                           # noqa: E402

CASES = {"wide": (8, 16), "short": (4, 8)}                 # range-latent (rows, columns); image latent is 8 x 8


def fake_batch(tag, B):
    bbox = lambda n: W.synth_input(f"al.{tag}.{n}", (B, 8, 3), kind="uniform") * 0.5 + 0.5
    return {"image": {"cond": {"ref_image": W.synth_input(f"al.{tag}.ref", (B, 3, 4, 4)), "ref_bbox": bbox("bi")}},
            "lidar": {"cond": {"ref_image": W.synth_input(f"al.{tag}.ref", (B, 3, 4, 4)), "ref_bbox": bbox("bl")}}}


def stand_in_cond(cond):
    """deterministic [B, 2, 768] tokens out of (ref_image, ref_bbox): enough to see order and the edited bbox"""
    b = cond["ref_bbox"].shape[0]
    t0 = cond["ref_image"].reshape(b, -1)[:, :48].repeat(1, 16).reshape(b, 1, 768)
    t1 = cond["ref_bbox"].reshape(b, -1).repeat(1, 32).reshape(b, 1, 768)
    return torch.cat([t0, t1], dim=1)


def main():
    R = mg.import_reference()
    LD = R.ddpm.LatentDiffusion
    out = {}
    B = 3
    for tag, (rows, cols) in CASES.items():
        z_image = W.synth_input(f"al.{tag}.zi", (B, 9, 8, 8))
        z_lidar = W.synth_input(f"al.{tag}.zl", (B, 9, rows, cols))

        class Fake:
            use_camera = True
            use_lidar = True
            image_size = 8
            first_stage_key = "inpaint"
            cond_stage_key = ["ref_image", "ref_bbox"]
            encode_all_stages = lambda self, **kw: (z_image.clone(), z_lidar.clone())
            process_conditioning = lambda self, cond, force_c_encode=False: (stand_in_cond(cond), cond)

        f = Fake()
        batch = fake_batch(tag, B)
        bbox_before = batch["lidar"]["cond"]["ref_bbox"].clone()
        # `super().get_input(batch, k)` (ddpm.py:760; DDPM.get_input returns the two modality dicts) cannot bind to the
        # stand-in object: the module-level name `super` is shadowed for this one call
        class _Super:
            get_input = staticmethod(lambda b, k: (b["image"], b["lidar"]))
        R.ddpm.__dict__["super"] = lambda *a: _Super()
        try:
            data = LD.get_input(f, batch, "inpaint", force_c_encode=True)
        finally:
            del R.ddpm.__dict__["super"]
        sample = W.synth_input(f"al.{tag}.smp", (2 * B, 4, 8, 8))
        h_cam, h_lid = LD.decode_sample(f, sample, data["z_lidar"].clone())
        out.update({f"{tag}_z_image": z_image, f"{tag}_z_lidar_in": z_lidar, f"{tag}_bbox_before": bbox_before,
                    f"{tag}_bbox_after": batch["lidar"]["cond"]["ref_bbox"], f"{tag}_z": data["z"],
                    f"{tag}_cond": data["cond"], f"{tag}_z_lidar": data["z_lidar"], f"{tag}_sample": sample,
                    f"{tag}_h_cam": h_cam, f"{tag}_h_lid": h_lid,
                    f"{tag}_ref_image": batch["image"]["cond"]["ref_image"],
                    f"{tag}_bbox_image": batch["image"]["cond"]["ref_bbox"]})
        print(tag, "z", tuple(data["z"].shape), "z_lidar", tuple(data["z_lidar"].shape), "h_lid", tuple(h_lid.shape))
    mg.save("plumbing_align", **out)


if __name__ == "__main__":
    main()
