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
import functools
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


@functools.lru_cache(maxsize=4)
def _grid_cache(H):
    """(cell-centre coordinate per index [H] float64, Morton index of every cell in (ix, iy, iz) C order [H^3] uint32)."""
    c = (np.arange(H, dtype=np.float64) + 0.5) * 2.0 / H - 1.0
    ix, iy, iz = np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing="ij")
    return c, morton3d(ix.reshape(-1), iy.reshape(-1), iz.reshape(-1))


def _index_range(c, lo, hi):
    """[i0, i1) of the cell centres c (ascending) inside [lo, hi]."""
    return int(np.searchsorted(c, lo, side="left")), int(np.searchsorted(c, hi, side="right"))


def _block_points(c, lo, hi):
    """Cell centres inside the axis-aligned box [lo, hi] as ([n,3] points, the three index ranges)."""
    r = [_index_range(c, lo[a], hi[a]) for a in range(3)]
    gx, gy, gz = np.meshgrid(c[r[0][0]:r[0][1]], c[r[1][0]:r[1][1]], c[r[2][0]:r[2][1]], indexing="ij")
    return np.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], -1), r


def _occupancy_to_bitfield(occ3, H):
    """occ3 [H,H,H] bool in (ix, iy, iz) order -> Morton-ordered bitfield (bit i%8 of byte i/8)."""
    flat = np.zeros(H * H * H, dtype=np.uint8)
    flat[_grid_cache(H)[1]] = occ3.reshape(-1)
    return np.packbits(flat.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)


def _capsule_parts(t):
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
    return parts


def jumpingjacks_occupancy(t, H=128, half_extent=1.0):
    """uint8 [H^3/8] Morton bitfield of the capsule figure at time t in [0,1]; the grid spans [-half_extent, half_extent]^3
    (1 = cascade 0 of a bound-1 scene; 2 = the second cascade of a bound-2 scene)."""
    c = _grid_cache(H)[0] * half_extent
    occ = np.zeros((H, H, H), dtype=bool)
    for a, b, r in _capsule_parts(t):
        lo = np.minimum(a, b) - r - 2.0 * half_extent / H
        hi = np.maximum(a, b) + r + 2.0 * half_extent / H
        pts, rg = _block_points(c, lo, hi)
        if pts.shape[0]:
            hit = (_capsule_dist(pts, a, b) < r).reshape(rg[0][1] - rg[0][0], rg[1][1] - rg[1][0], rg[2][1] - rg[2][0])
            occ[rg[0][0]:rg[0][1], rg[1][0]:rg[1][1], rg[2][0]:rg[2][1]] |= hit
    return _occupancy_to_bitfield(occ, H)


def lego_occupancy(H=128, half_extent=1.0):
    """uint8 [H^3/8] Morton bitfield: 0.6-side box with a 4x4 stud pattern on its top face."""
    c = _grid_cache(H)[0] * half_extent
    X, Y, Z = c[:, None, None], c[None, :, None], c[None, None, :]
    occ = (np.abs(X) <= 0.3) & (np.abs(Y) <= 0.15) & (np.abs(Z) <= 0.3)
    for sx in range(4):
        for sz in range(4):
            cx, cz = -0.225 + 0.15 * sx, -0.225 + 0.15 * sz
            r2 = (X - cx) ** 2 + (Z - cz) ** 2
            occ = occ | ((r2 < 0.045 ** 2) & (Y > 0.15) & (Y < 0.21))
    return _occupancy_to_bitfield(occ, H)


def density_bitfield_all_times(time_size=64, H=128, kind="jumpingjacks", times=None):
    """[time_size, H^3/8] uint8, one slice per density-grid time stamp (dnerf/renderer.py:93,99)."""
    return density_bitfield_cascades(time_size, H, 1, 1.0, kind, times)


def density_bitfield_cascades(time_size=64, H=128, cascade=1, bound=1.0, kind="jumpingjacks", times=None):
    """[time_size, cascade * H^3/8] uint8: per time stamp the `cascade` Morton grids back to back (dnerf/renderer.py:92-93);
    cascade c spans [-min(2^c, bound), +min(2^c, bound)]^3 (raymarching.cu:371-379)."""
    per = H * H * H // 8
    out = np.zeros((time_size, cascade * per), dtype=np.uint8)
    lego = {}
    for k in range(time_size):
        if times is not None and k not in times:
            continue
        t = (k + 0.5) / time_size
        for cas in range(cascade):
            ext = min(float(2 ** cas), float(bound))
            if kind == "jumpingjacks":
                out[k, cas * per:(cas + 1) * per] = jumpingjacks_occupancy(t, H, ext)
            else:
                if cas not in lego:
                    lego[cas] = lego_occupancy(H, ext)
                out[k, cas * per:(cas + 1) * per] = lego[cas]
    return out
