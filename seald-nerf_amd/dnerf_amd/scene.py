"""Synthetic, fully specified D-NeRF-like scenes (there is no dataset in the reference tree:
.gitignore:1 `data/`).  Everything here is host-side numpy and deterministic; SURVEY.md section 8(d)
gives the specification this follows.

  * camera: 800x800 pinhole, camera_angle_x = 0.6911 (fx = fy = 0.5*W/tan(0.5*angle)), centre of the
    image, pose on a circle of radius 4.0*0.8 = 3.2 at 30 degrees elevation looking at the origin, in
    the frame `nerf_matrix_to_ngp` produces (dnerf/provider.py:18-26: y up, camera looks along +z of
    its own frame, image y down); rays as `get_rays` builds them (nerf/utils.py:54-137: pixel centres
    +0.5, directions normalised, row-major pixels).
  * "jumpingjacks-like" occupancy: union of six capsules (torso, head, two arms, two legs) whose
    limb angle is 40 deg * sin(2 pi t); rasterised at cell centres into the 128^3 Morton-ordered
    bitfield layout the marching kernels read (bit i%8 of byte i/8, raymarching.cu:378-379).
  * "lego-like" occupancy: a 0.6-side box with a 4x4 stud pattern on top.
"""
import math

import numpy as np

CAMERA_ANGLE_X = 0.6911
CAMERA_RADIUS = 4.0 * 0.8


def intrinsics(H, W, camera_angle_x=CAMERA_ANGLE_X):
    fl = 0.5 * W / math.tan(0.5 * camera_angle_x)
    return np.array([fl, fl, W / 2.0, H / 2.0], dtype=np.float64)


def look_at_pose(azimuth_deg=30.0, elevation_deg=30.0, radius=CAMERA_RADIUS):
    """cam2world [4,4] float32; columns = (right, down, forward, position)."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    p = np.array([radius * math.cos(el) * math.sin(az), radius * math.sin(el), radius * math.cos(el) * math.cos(az)])
    fwd = -p / np.linalg.norm(p)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, p
    return pose.astype(np.float32)


def get_rays(pose, intr, H, W):
    """rays_o, rays_d [H*W, 3] float32, as nerf/utils.py:54-137 computes them for N = -1 (all pixels)."""
    fx, fy, cx, cy = [np.float32(v) for v in intr]
    i = (np.arange(W, dtype=np.float32) + np.float32(0.5))[None, :].repeat(H, 0).reshape(-1)
    j = (np.arange(H, dtype=np.float32) + np.float32(0.5))[:, None].repeat(W, 1).reshape(-1)
    xs = (i - cx) / fx
    ys = (j - cy) / fy
    zs = np.ones_like(xs)
    d = np.stack([xs, ys, zs], -1).astype(np.float32)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True).astype(np.float32)
    rays_d = (d @ pose[:3, :3].T.astype(np.float32)).astype(np.float32)
    rays_o = np.broadcast_to(pose[:3, 3].astype(np.float32), rays_d.shape).copy()
    return rays_o, rays_d


def _part1by2(v):
    v = v.astype(np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d(ix, iy, iz):
    return _part1by2(ix) | (_part1by2(iy) << np.uint32(1)) | (_part1by2(iz) << np.uint32(2))


def _capsule_dist(p, a, b):
    pa, ba = p - a, b - a
    h = np.clip((pa @ ba) / (ba @ ba), 0.0, 1.0)
    return np.linalg.norm(pa - h[:, None] * ba, axis=1)


def _cell_centres(H):
    c = (np.arange(H, dtype=np.float64) + 0.5) * 2.0 / H - 1.0
    ix, iy, iz = np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing="ij")
    pts = np.stack([c[ix.reshape(-1)], c[iy.reshape(-1)], c[iz.reshape(-1)]], -1)
    return pts, ix.reshape(-1), iy.reshape(-1), iz.reshape(-1)


def _occupancy_to_bitfield(occ, ix, iy, iz, H):
    idx = morton3d(ix, iy, iz)
    flat = np.zeros(H * H * H, dtype=np.uint8)
    flat[idx[occ]] = 1
    return np.packbits(flat.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)


def jumpingjacks_occupancy(t, H=128):
    """uint8 [H^3/8] Morton bitfield of the capsule figure at time t in [0,1] (cascade 0, bound 1)."""
    pts, ix, iy, iz = _cell_centres(H)
    ang = math.radians(40.0) * math.sin(2 * math.pi * t)
    parts = []
    parts.append((np.array([0.0, -0.15, 0.0]), np.array([0.0, 0.30, 0.0]), 0.11))          # torso
    parts.append((np.array([0.0, 0.47, 0.0]), np.array([0.0, 0.48, 0.0]), 0.10))            # head
    sh_l, sh_r = np.array([-0.12, 0.28, 0.0]), np.array([0.12, 0.28, 0.0])
    arm = 0.30
    parts.append((sh_l, sh_l + arm * np.array([-math.cos(ang), math.sin(ang), 0.0]), 0.045))  # arms
    parts.append((sh_r, sh_r + arm * np.array([math.cos(ang), math.sin(ang), 0.0]), 0.045))
    hip_l, hip_r = np.array([-0.06, -0.18, 0.0]), np.array([0.06, -0.18, 0.0])
    leg = 0.40
    la = 0.5 * abs(ang)
    parts.append((hip_l, hip_l + leg * np.array([-math.sin(la), -math.cos(la), 0.0]), 0.055))  # legs
    parts.append((hip_r, hip_r + leg * np.array([math.sin(la), -math.cos(la), 0.0]), 0.055))
    occ = np.zeros(pts.shape[0], dtype=bool)
    for a, b, r in parts:
        lo = np.minimum(a, b) - r - 2.0 / H
        hi = np.maximum(a, b) + r + 2.0 / H
        m = np.all((pts >= lo) & (pts <= hi), axis=1)
        sel = np.nonzero(m)[0]
        occ[sel] |= _capsule_dist(pts[sel], a, b) < r
    return _occupancy_to_bitfield(occ, ix, iy, iz, H)


def lego_occupancy(H=128):
    """uint8 [H^3/8] Morton bitfield: 0.6-side box with a 4x4 stud pattern on its top face."""
    pts, ix, iy, iz = _cell_centres(H)
    occ = np.all(np.abs(pts) <= np.array([0.3, 0.15, 0.3]), axis=1)
    for sx in range(4):
        for sz in range(4):
            cx, cz = -0.225 + 0.15 * sx, -0.225 + 0.15 * sz
            r2 = (pts[:, 0] - cx) ** 2 + (pts[:, 2] - cz) ** 2
            occ |= (r2 < 0.045 ** 2) & (pts[:, 1] > 0.15) & (pts[:, 1] < 0.21)
    return _occupancy_to_bitfield(occ, ix, iy, iz, H)


def density_bitfield_all_times(time_size=64, H=128, kind="jumpingjacks", times=None):
    """[time_size, H^3/8] uint8, one slice per density-grid time stamp (dnerf/renderer.py:93,99)."""
    out = np.zeros((time_size, H * H * H // 8), dtype=np.uint8)
    for k in range(time_size):
        if times is not None and k not in times:
            continue
        t = (k + 0.5) / time_size
        out[k] = jumpingjacks_occupancy(t, H) if kind == "jumpingjacks" else lego_occupancy(H)
    return out
