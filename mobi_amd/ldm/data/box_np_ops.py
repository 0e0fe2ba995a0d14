"""3-D box geometry the sampling harness needs (reference: ldm/data/box_np_ops.py -- corner_to_surfaces_3d :406-427,
points_in_bbox_corners :453-471, surface_equ_3d :712-732, points_in_convex_polygon_3d_jit :736-797).

The six surface equations of a box are 24 numbers computed on the host; the test of every point of a sweep against them
runs on the device (`mobi_range_paste` fuses it with the range-view paste; `points_in_bbox_corners` here is the
stand-alone form for point arrays)."""
import numpy as np
import torch

# corner indices of the six faces, normals pointing inwards (the reference's table, :417-425)
_FACES = ((0, 1, 2, 3), (7, 6, 5, 4), (0, 3, 7, 4), (1, 5, 6, 2), (0, 4, 5, 1), (3, 2, 6, 7))


def corner_to_surfaces_3d(corners):
    """corners [N, 8, 3] -> surfaces [N, 6, 4, 3]."""
    corners = np.asarray(corners)
    return np.stack([np.stack([corners[:, i] for i in face], axis=1) for face in _FACES], axis=1)


def surface_equ_3d(polygon_surfaces):
    """(normal [N, S, 3], d [N, S]) of a x + b y + c z + d = 0 through the first three points of every surface."""
    p = np.asarray(polygon_surfaces)
    edge = p[:, :, :2, :] - p[:, :, 1:3, :]
    normal = np.cross(edge[:, :, 0, :], edge[:, :, 1, :])
    d = np.einsum("aij, aij->ai", normal, p[:, :, 0, :])
    return normal, -d


def box_planes(corners):
    """[N, 8, 3] corners -> fp32 [N, 6, 4] = (nx, ny, nz, d) in the corners' own precision, then cast."""
    normal, d = surface_equ_3d(corner_to_surfaces_3d(corners)[:, :, :3, :])
    return np.concatenate([normal, d[..., None]], axis=-1).astype(np.float32)


def points_in_bbox_corners(points, rbbox_corners):
    """points [P, >=3], corners [M, 8, 3] -> bool [P, M]: inside iff every surface's sign is negative.
    Tensors on the GPU stay there (one fused comparison); numpy inputs give a numpy result like the reference."""
    planes = box_planes(np.asarray(rbbox_corners.detach().cpu() if isinstance(rbbox_corners, torch.Tensor) else rbbox_corners))
    as_numpy = not isinstance(points, torch.Tensor)
    pts = torch.as_tensor(np.asarray(points) if as_numpy else points)[:, :3].to(torch.float32)
    pl = torch.from_numpy(planes).to(pts.device)                       # [M, 6, 4]
    # x nx + y ny + z nz + d in the reference's left-to-right fp32 order
    sign = pts[:, None, None, 0] * pl[None, :, :, 0] + pts[:, None, None, 1] * pl[None, :, :, 1]
    sign = sign + pts[:, None, None, 2] * pl[None, :, :, 2]
    sign = sign + pl[None, :, :, 3]
    inside = (sign < 0).all(dim=-1)
    return inside.cpu().numpy() if as_numpy else inside
