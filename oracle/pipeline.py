"""LatentDiffusion inference plumbing on CPU (TEST INFRASTRUCTURE).

Reference: ldm/models/diffusion/ddpm.py -- encode_all_stages :1010-1033,
get_input :758-834, decode_sample :1420-1447, decode_first_stage :837-901;
cat_interleave ldm/util.py:213-221.

The conditioning producer (CLIP image embedder + bbox embedder,
ldm/modules/encoders/modules.py:142-215) is a SURVEY.md 8(f) "next" row: here
the `[B,2,768]` context of each modality is an input.
"""
import torch
import torch.nn.functional as F

from . import vae as ovae


def cat_interleave(tensors):
    """ldm/util.py:213-221: [a0,b0,a1,b1,...] along the batch axis."""
    return torch.stack(tensors, dim=1).reshape(-1, *tensors[0].shape[1:])


def encode_modality(sd, cfg, gt, inpaint, mask, noise_gt, noise_inpaint, scale_factor):
    """One branch of encode_all_stages (ddpm.py:1013-1021 / :1023-1031)."""
    z = scale_factor * ovae.posterior_sample(ovae.encode_moments(sd, cfg, gt), noise_gt)
    z_inp = scale_factor * ovae.posterior_sample(ovae.encode_moments(sd, cfg, inpaint), noise_inpaint)
    mask_r = F.interpolate(mask, size=z.shape[-1], mode="nearest")
    return torch.cat((z, z_inp, mask_r), dim=1)


def align_lidar(z_lidar, ref_bbox, image_size):
    """ddpm.py:798-815: centre-crop the lidar latent to `image_size` columns,
    zero-pad rows, and renormalise the bbox x/y to the cropped map.  Returns
    (aligned latent, adjusted bbox) -- the reference edits the bbox in place."""
    W = z_lidar.shape[-1]
    left, right = W // 2 - image_size // 2, W // 2 + image_size // 2
    pad = (image_size - z_lidar.shape[-2]) // 2
    z = F.pad(z_lidar[..., left:right], (0, 0, pad, pad), mode="constant", value=0)
    bbox = ref_bbox.clone()
    bbox[..., 0] = (bbox[..., 0] * W - left) / image_size
    bbox[..., 1] += pad / image_size
    return z, bbox


def decode_sample(sample, z_lidar, image_size):
    """ddpm.py:1420-1433 (camera+lidar case): de-interleave, undo pad / crop."""
    h_camera = sample[::2]
    lid = sample[1::2]
    bottom = (lid.shape[-2] - z_lidar.shape[-2]) // 2
    top = bottom + z_lidar.shape[-2]
    h_lidar = lid[:, :, bottom:top, :]
    if image_size != z_lidar.shape[-1]:
        z_lidar = z_lidar.clone()
        c = z_lidar.shape[-1] // 2
        z_lidar[..., c - image_size // 2: c + image_size // 2] = h_lidar
        h_lidar = z_lidar
    return h_camera, h_lidar


def decode_first_stage(sd, cfg, z, scale_factor):
    """ddpm.py:847-850,896-899 + the clamp of log_data (:1476,:1504)."""
    return torch.clamp(ovae.decode(sd, cfg, (1.0 / scale_factor) * z[:, :4]), -1.0, 1.0)
