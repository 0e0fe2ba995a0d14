"""Harness-side helpers of the sampling path on the device (reference: ldm/data/utils.py -- un_norm :362-363,
un_norm_clip :366-371, postprocess_range_depth_int :471-505, postprocess_range_depth :507-534,
depth_normalization :537-557, inverse_depth_normalization :560-580; scripts/inference_test_bench.py:478-527 camera
paste-back, :567-610 range-view paste).

The reference hops to numpy / cv2 per sample; here the arithmetic runs in `mobi_range_paste`, `mobi_range_denorm`,
`mobi_paste_patch`, `mobi_gaussian_blur`, `mobi_blend_frame` and the results come back in ONE copy per call.  The drawing
helpers (get_camera_vis boxes, visualize_lidar) stay with whoever has cv2."""
import numpy as np
import torch
import torch.nn.functional as F

from ... import ops
from .box_np_ops import box_planes
from .lidar_converter import LidarConverter


def _resize(x, size):
    """torchvision.transforms.Resize(size) on a float NCHW tensor as the reference's pinned torchvision (0.11) does it:
    bilinear, align_corners=False, no antialias; a no-op at the target size."""
    if tuple(x.shape[-2:]) == tuple(size):
        return x
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)


def un_norm(x, size=(512, 512)):
    return (_resize(x, size) + 1.0) / 2.0


def un_norm_clip(x, size=(512, 512)):
    x = _resize(x, size).clone()
    x[:, 0] = x[:, 0] * 0.26862954 + 0.48145466
    x[:, 1] = x[:, 1] * 0.26130258 + 0.4578275
    x[:, 2] = x[:, 2] * 0.27577711 + 0.40821073
    return x


def depth_normalization(depth, min_d, max_d, alpha=0.75):
    """Dataset-side forward map (piecewise linear); plain tensor expressions, any device."""
    out = torch.empty_like(depth)
    min_d, max_d = torch.as_tensor(min_d).to(depth.dtype), torch.as_tensor(max_d).to(depth.dtype)
    mid = (depth >= min_d) & (depth <= max_d)
    out[mid] = -alpha + 2 * alpha * (depth[mid] - min_d) / (max_d - min_d)
    low = (depth >= -1) & (depth < min_d)
    out[low] = -1 + -(alpha - 1) * (depth[low] + 1) / (min_d + 1)
    high = (depth > max_d) & (depth <= 1)
    out[high] = alpha + (1 - alpha) * (depth[high] - max_d) / (1 - max_d)
    return out


def inverse_depth_normalization(normalized_depth, min_d, max_d, alpha=0.75):
    """One sample [1, H, W] (or any shape) with scalar tensors min_d / max_d -> de-normalised depth (mobi_range_denorm)."""
    nd = normalized_depth.float()
    flat = nd.reshape(1, 1, -1, 1)
    smp = torch.cat([flat, torch.zeros_like(flat)], dim=1).contiguous()
    lo = torch.as_tensor(min_d, dtype=torch.float32, device=nd.device).reshape(1)
    hi = torch.as_tensor(max_d, dtype=torch.float32, device=nd.device).reshape(1)
    depth, _ = ops.range_denorm(smp, lo, hi, alpha=alpha, object_norm=True, int_norm=False)
    return depth.reshape(nd.shape)


def _as_dev(x, device):
    return torch.as_tensor(x).to(device=device)


def postprocess_range_depth_int(*, range_depth, range_depth_orig, range_int, range_int_orig, crop_left, width_crop,
                                zero_context=False):
    """[B, 1, Hc, Wc] samples + [B, H0, W0] originals -> (depth [B, H0, W0], intensity [B, H0, W0]) numpy arrays."""
    dev = range_depth.device if range_depth.is_cuda else torch.device("cuda")
    d, i = _as_dev(range_depth, dev).float()[:, 0], _as_dev(range_int, dev).float()[:, 0]
    d0, i0 = _as_dev(range_depth_orig, dev).float(), _as_dev(range_int_orig, dev).float()
    if zero_context:
        d0 = d0 * 0 - 1
    out = ops.range_paste(d, d0, crop_left, width_crop, sample_int=i, int_orig=i0)
    return out["depth_unc"].cpu().numpy(), out["int_unc"].cpu().numpy()


def postprocess_range_depth(*, range_depth, range_depth_orig, crop_left, width_crop, zero_context=False):
    dev = range_depth.device if range_depth.is_cuda else torch.device("cuda")
    d, d0 = _as_dev(range_depth, dev).float()[:, 0], _as_dev(range_depth_orig, dev).float()
    if zero_context:
        d0 = d0 * 0 - 1
    return ops.range_paste(d, d0, crop_left, width_crop)["depth_unc"].cpu().numpy()


def paste_range_objects(*, range_depth, range_int, range_depth_orig, range_int_orig, crop_left, width_crop, range_pitch,
                        range_yaw, bbox_3d, gt_instance_mask, depth_interval=(1.4, 54)):
    """The per-sample range-view paste of scripts/inference_test_bench.py:567-610 for the WHOLE batch in one launch:
    un-crop, range2pcd, points-in-box instance mask of the predicted object, np.where paste.  Device tensors in,
    device tensors out: dict(depth_unc, int_unc, depth_final, int_final, pred_mask)."""
    dev = range_depth.device
    planes = torch.from_numpy(box_planes(np.asarray(bbox_3d.detach().cpu() if isinstance(bbox_3d, torch.Tensor) else bbox_3d)))
    return ops.range_paste(range_depth.float()[:, 0], _as_dev(range_depth_orig, dev).float(), crop_left, width_crop,
                           sample_int=range_int.float()[:, 0], int_orig=_as_dev(range_int_orig, dev).float(),
                           pitch=_as_dev(range_pitch, dev).float(), yaw=_as_dev(range_yaw, dev).float(),
                           planes=planes.to(dev), gt_mask=_as_dev(gt_instance_mask, dev) != 0,
                           depth_interval=depth_interval)


def gaussian_kernel1d(ksize, sigma):
    """cv2.getGaussianKernel(ksize, sigma) for sigma > 0: exp(-(i - (ksize - 1) / 2)^2 / (2 sigma^2)), normalised,
    computed in float64 and handed to the fp32 filter (cv2 absent here: restated from its documentation)."""
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def paste_camera_patch(*, patch_pred, image, mask, crop, ksize=15, sigma=7.0):
    """The camera paste-back of scripts/inference_test_bench.py:478-510 for one sample on the device.
    patch_pred: fp32 [1, 3, hs, ws] in [-1, 1]; image: fp32 [3, H, W] in [-1, 1]; mask: [H, W] (1 = keep);
    crop = (left, top, crop_W, crop_H).  Returns (image_recon fp32 [H, W, 3] BGR, image_pred uint8 [H, W, 3] BGR)."""
    dev = patch_pred.device
    left, top, crop_w, crop_h = (int(v) for v in crop)
    image = _as_dev(image, dev).float().contiguous()
    h, w = image.shape[1:]
    frame = torch.zeros((h, w, 3), device=dev, dtype=torch.uint8)
    ops.paste_patch(patch_pred.float().reshape(3, *patch_pred.shape[-2:]).contiguous(), frame, top, left, crop_h, crop_w)
    blur = ops.gaussian_blur(_as_dev(mask, dev).float().contiguous(), torch.from_numpy(gaussian_kernel1d(ksize, sigma)).to(dev))
    return ops.blend_frame(blur, image, frame), frame


# --------------------------------------------------------------------------------------
# pictures (visualisation only; the drawing primitives are cv2's in the reference)
# --------------------------------------------------------------------------------------
_BOX_EDGES = ((0, 1), (0, 3), (3, 2), (1, 2), (1, 5), (0, 4), (3, 7), (2, 6), (4, 7), (4, 5), (5, 6), (6, 7))


def _line(img, p0, p1, color):
    """1-pixel DDA segment, clipped to the image (stand-in for cv2.line, thickness 1)."""
    n = int(max(abs(p1[0] - p0[0]), abs(p1[1] - p0[1]))) + 1
    xs = np.rint(np.linspace(p0[0], p1[0], n)).astype(int)
    ys = np.rint(np.linspace(p0[1], p1[1], n)).astype(int)
    ok = (xs >= 0) & (xs < img.shape[1]) & (ys >= 0) & (ys < img.shape[0])
    img[ys[ok], xs[ok]] = color


def draw_projected_bbox(image, bbox_coords, color=(0, 165, 255), thickness=2):
    """[H, W, 3] uint8 numpy + 8 projected corners in [0, 1] image coordinates -> outline drawn (ldm/data/utils.py:200-252)."""
    image = np.ascontiguousarray(image)
    h, w = image.shape[:2]
    pts = np.asarray(bbox_coords, dtype=np.float64)[:, :2] * np.array([w, h])
    for a_, b_ in _BOX_EDGES:
        _line(image, pts[a_], pts[b_], color)
    return image


def draw_boxes_if_possible(images_u8, ref_bboxes):
    """uint8 [B, 3, H, W] device tensors -> the same with the projected reference box outlined (one host round trip for
    the pictures; skipped, returning the inputs, when there is no box)."""
    if ref_bboxes is None:
        return images_u8
    boxes = ref_bboxes.detach().cpu().numpy()
    out = []
    for x in images_u8:
        arr = x.permute(0, 2, 3, 1).cpu().numpy()
        arr = np.stack([draw_projected_bbox(a_, boxes[i, :, :2], color=(255, 165, 0), thickness=1) for i, a_ in enumerate(arr)])
        out.append(torch.from_numpy(arr).permute(0, 3, 1, 2).to(x.device))
    return out


def focus_on_bbox(points, bbox_3d):
    """ldm/data/utils.py:340-359: centre on the box, rotate by -+45 deg about z and -60 deg about x."""
    center = np.mean(bbox_3d, axis=0)
    tz = (1 if center[0] > 0 else -1) * np.pi / 4
    rz = np.array([[np.cos(tz), -np.sin(tz), 0], [np.sin(tz), np.cos(tz), 0], [0, 0, 1]])
    tx = -np.pi / 3
    rx = np.array([[1, 0, 0], [0, np.cos(tx), -np.sin(tx)], [0, np.sin(tx), np.cos(tx)]])
    rot = np.dot(rx, rz)
    return np.dot(points - center, rot.T), np.dot(bbox_3d - center, rot.T)


def visualize_lidar(lidar=None, *, bboxes=None, xlim=(-10, 10), ylim=(-10, 10), bbox_color=(0, 165, 255),
                    points_color=(0, 128, 128), dpi=20):
    """Top-down raster of a point cloud with box outlines (ldm/data/utils.py:280-337; the orientation arrow is cv2's)."""
    img = np.ones((int((ylim[1] - ylim[0]) * dpi), int((xlim[1] - xlim[0]) * dpi), 3), dtype=np.uint8) * 255
    if bboxes is not None and len(bboxes) > 0:
        for box in (bboxes[None] if bboxes.ndim == 2 else bboxes):
            px = lambda v: (int(v[0] * dpi - xlim[0] * dpi), int((ylim[1] - v[1]) * dpi))
            for a_, b_ in _BOX_EDGES:
                _line(img, px(box[a_]), px(box[b_]), bbox_color)
    if lidar is not None:
        pts = lidar.copy()
        pts[:, 0] = (pts[:, 0] - xlim[0]) * dpi
        pts[:, 1] = (ylim[1] - pts[:, 1]) * dpi
        ok = (pts[:, 0] >= 0) & (pts[:, 0] < img.shape[1]) & (pts[:, 1] >= 0) & (pts[:, 1] < img.shape[0])
        pts = pts[ok].astype(int)
        img[pts[:, 1], pts[:, 0]] = points_color
    return img


def get_lidar_vis(*, sample, input, rec, bboxes, range_depth_orig, range_shift_left, range_pitch, range_yaw, width_crop):
    """ldm/data/utils.py:409-468: three top-down pictures per sample (prediction, input, reconstruction)."""
    bboxes = bboxes.detach().cpu().numpy()
    pitch, yaw = range_pitch.detach().cpu().numpy(), range_yaw.detach().cpu().numpy()
    unc = [postprocess_range_depth(range_depth=x, range_depth_orig=range_depth_orig, crop_left=range_shift_left,
                                   width_crop=width_crop, zero_context=True) for x in (sample, input, rec)]
    conv = LidarConverter()
    vis = [[], [], []]
    for i in range(len(unc[0])):
        box = bboxes[i]
        clouds = [focus_on_bbox(conv.range2pcd(u[i], pitch[i], yaw[i])[0], box) for u in unc]
        for k in range(3):
            vis[k].append(visualize_lidar(clouds[k][0], bboxes=clouds[2][1], bbox_color=(255, 165, 0)))
    return tuple(torch.from_numpy(np.stack(v)).permute(0, 3, 1, 2) for v in vis)


# ----------------------------------------------------------------------------------------------------------------------
# dataset side (reference: ldm/data/utils.py -- get_image_coords :44-73, rotate_bbox :75-103, translate_bbox :106-122,
# get_camera_coords :125-144, get_inpaint_mask :146-171, get_range_inpaint_mask :174-198, get_2d_bbox :254-265,
# expand_bbox_corners :268-278).  Eight corners per call: numpy, the reference's own operation order (float64).
# ----------------------------------------------------------------------------------------------------------------------
BOX_FACES = ((0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (0, 4, 7, 3), (1, 5, 6, 2))


def _homogeneous(bbox_corners, matrix):
    pts = np.concatenate([bbox_corners.reshape(-1, 3), np.ones((8, 1))], axis=-1)
    return (pts @ matrix.copy().reshape(4, 4).T).reshape(8, 4)


def get_image_coords(bbox_corners, lidar2image, include_depth=False):
    """[8, 3] lidar-frame corners -> pixel (x, y[, depth]); depth clipped to [1e-5, 1e5] before the divide."""
    pr = _homogeneous(bbox_corners, lidar2image)
    pr[..., 2] = np.clip(pr[..., 2], a_min=1e-5, a_max=1e5)
    pr[..., :2] /= pr[..., 2, None]
    return pr[..., :3].reshape(8, 3) if include_depth else pr[..., :2].reshape(8, 2)


def get_camera_coords(bbox_corners, lidar2camera):
    return _homogeneous(bbox_corners, lidar2camera)[..., :3]


def _about_centre(bbox_corners, fn):
    import copy
    box = copy.deepcopy(bbox_corners)
    centre = np.mean(box, axis=0)
    box -= centre
    return fn(box, centre)


def rotate_bbox(bbox_corners, angle=0):
    """Turn the box about the vertical axis through its centre (degrees)."""
    if angle == 0:
        return bbox_corners
    a = np.deg2rad(angle)
    rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])

    def turn(box, centre):
        box = box @ rot.T
        box += centre
        return box
    return _about_centre(bbox_corners, turn)


def translate_bbox(bbox_corners, new_center):
    def move(box, centre):
        box += new_center
        return box
    return _about_centre(bbox_corners, move)


def expand_bbox_corners(bbox_corners, expand_ratio=0.1):
    if expand_ratio == 0:
        return bbox_corners

    def grow(box, centre):
        box *= (1 + expand_ratio)
        box += centre
        return box
    return _about_centre(bbox_corners, grow)


def get_2d_bbox(bbox_corners, transform, H, W, expand_ratio=0.1):
    """Axis-aligned pixel box (x1, y1, x2, y2) of the projected (expanded) corners, clipped to the image."""
    xy = get_image_coords(expand_bbox_corners(bbox_corners, expand_ratio), transform)
    box = np.concatenate([np.min(xy, axis=-2), np.max(xy, axis=-2)], axis=-1).astype(int)
    box[0::2] = np.clip(box[0::2], a_min=0, a_max=W - 1)
    box[1::2] = np.clip(box[1::2], a_min=0, a_max=H - 1)
    return box


def fill_box_faces(coords_xy, H, W):
    """uint8 [H, W]: 1 where a pixel belongs to one of the six projected faces of a box (corner pixel coordinates
    truncated to int32 first, as before the reference's `cv2.fillPoly(mask, [points], 1, cv2.LINE_AA)`).

    cv2 is not in this image, so its polygon rasteriser is RESTATED, not pinned: a pixel is taken when its centre lies
    inside the (convex) face or within half a pixel of the face's outline -- OpenCV fills the scan-line interior and
    draws the outline on top, whose anti-aliased pixels reach the fill colour 1 at >= 50 % coverage.  The two can
    differ on single outline pixels only.  The batched device form is `ops.box_mask`."""
    mask = np.zeros((H, W), dtype=np.uint8)
    for face in BOX_FACES:
        q = coords_xy[list(face)].astype(np.int32).astype(np.float64)
        x0, x1 = int(max(0, np.floor(q[:, 0].min() - 1))), int(min(W - 1, np.ceil(q[:, 0].max() + 1)))
        y0, y1 = int(max(0, np.floor(q[:, 1].min() - 1))), int(min(H - 1, np.ceil(q[:, 1].max() + 1)))
        if x1 < x0 or y1 < y0:
            continue
        xs, ys = np.meshgrid(np.arange(x0, x1 + 1, dtype=np.float64), np.arange(y0, y1 + 1, dtype=np.float64))
        pos = np.ones_like(xs, dtype=bool)
        neg = np.ones_like(xs, dtype=bool)
        near = np.zeros_like(xs, dtype=bool)
        for k in range(4):
            ax, ay = q[k]
            bx, by = q[(k + 1) % 4]
            ex, ey = bx - ax, by - ay
            cross = ex * (ys - ay) - ey * (xs - ax)
            pos &= cross >= 0
            neg &= cross <= 0
            ll = ex * ex + ey * ey
            t = np.clip(((xs - ax) * ex + (ys - ay) * ey) / ll, 0.0, 1.0) if ll > 0 else np.zeros_like(xs)
            dx, dy = xs - (ax + t * ex), ys - (ay + t * ey)
            near |= dx * dx + dy * dy <= 0.25
        box = (xs >= q[:, 0].min()) & (xs <= q[:, 0].max()) & (ys >= q[:, 1].min()) & (ys <= q[:, 1].max())
        mask[y0:y1 + 1, x0:x1 + 1] |= (((pos | neg) & box) | near).astype(np.uint8)
    return mask


def get_inpaint_mask(bbox_corners, transform, H, W, expand_ratio=0.1, use_3d_edit_mask=True):
    """float [H, W]: 0 inside the edit region (the projected faces of the expanded box, or its 2D bounding box), 1 outside."""
    if use_3d_edit_mask:
        mask = fill_box_faces(get_image_coords(expand_bbox_corners(bbox_corners, expand_ratio), transform), H, W)
    else:
        x1, y1, x2, y2 = get_2d_bbox(bbox_corners, transform, H, W, expand_ratio)
        mask = np.zeros((H, W), dtype=np.uint8)
        mask[y1:y2, x1:x2] = 1
    return 1. - torch.tensor(mask > 0.5).float()


def get_range_inpaint_mask(bbox_corners, range_height, range_width, expand_ratio=0.1, crop_left=None, width_crop=None):
    conv = LidarConverter()
    coords = conv.get_range_coords(expand_bbox_corners(bbox_corners, expand_ratio))
    _, _, _, coords, _, _ = conv.apply_default_transforms(coords, height=range_height, width=range_width,
                                                          crop_left=crop_left, width_crop=width_crop)
    mask = fill_box_faces(coords[:, :2], range_height, range_width)
    return 1. - torch.tensor(mask > 0.5).float()
