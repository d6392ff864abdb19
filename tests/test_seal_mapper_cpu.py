"""The SealD bounding-box mapper (dnerf_amd/seal_mapper.py) on the CPU: colour conversions against vectors generated from the
reference's pure-torch `SealNeRF/color_utils.py` (tests/golden/gen_golden_color.py), the box / inside-test geometry against an analytic
oriented-box test, and the map_to_origin algebra (it inverts the configured edit)."""
import os

import numpy as np
import pytest
import torch

from dnerf_amd import seal_mapper as SM


def _rot(axis, deg):
    a = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    t = np.deg2rad(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * K @ K


def _cuboid(center, half, R, rng=None):
    corners = np.array([[sx, sy, sz] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)], np.float64) * half
    pts = corners @ R.T + center
    return pts[rng.permutation(8)] if rng is not None else pts


def test_colour_conversions_match_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "color_reference_torch.npz"))
    hsv = SM.rgb2hsv(torch.from_numpy(g["rgb"]))
    assert np.allclose(hsv.numpy(), g["hsv"], atol=1e-6)
    assert np.allclose(SM.hsv2rgb(torch.from_numpy(g["hsv"])).numpy(), g["back"], atol=1e-6)
    assert np.allclose(SM.hsv2rgb(torch.from_numpy(g["hsv_in"])).numpy(), g["rgb_from"], atol=1e-6)
    # modify_hsv with a zero modification is the identity up to the conversion error; a hue shift keeps value and saturation
    rgb = torch.from_numpy(g["rgb"])
    assert torch.allclose(SM.modify_hsv(rgb, [0.0, 0.0, 0.0]), rgb, atol=1e-6)
    shifted = SM.rgb2hsv(SM.modify_hsv(rgb, [0.25, 0.0, 0.0]))
    assert torch.allclose(shifted[:, 1:], hsv[:, 1:], atol=1e-5)
    tint = SM.modify_rgb(rgb, torch.tensor([0.2, 0.6, 0.9]), 0.0)
    assert tint.shape == rgb.shape and float(tint.min()) >= -1e-6 and float(tint.max()) <= 1 + 1e-6


def test_oriented_box_recognises_shuffled_cuboid_corners():
    rng = np.random.default_rng(0)
    for half in ([0.3, 0.2, 0.1], [0.2, 0.2, 0.2], [0.05, 0.4, 0.4]):     # distinct extents, a cube (PCA is degenerate), a plate
        R = _rot([1, 2, 3], 37.0)
        pts = _cuboid(np.array([0.1, -0.2, 0.05]), np.array(half), R, rng)
        verts, centre = SM.oriented_box(pts)
        assert np.allclose(centre, [0.1, -0.2, 0.05], atol=1e-9)
        d = np.linalg.norm(verts[:, None] - pts[None], axis=-1)
        assert (d.min(0) < 1e-9).all() and (d.min(1) < 1e-9).all()
        e1, e2, e3 = verts[1] - verts[0], verts[2] - verts[0], verts[4] - verts[0]
        assert abs(e1 @ e2) < 1e-9 and abs(e1 @ e3) < 1e-9 and abs(e2 @ e3) < 1e-9


def test_points_in_mesh_equals_analytic_box_test():
    rng = np.random.default_rng(1)
    R, c, half = _rot([0.3, 1, -0.5], 52.0), np.array([0.05, 0.1, -0.1]), np.array([0.3, 0.15, 0.22])
    verts, _ = SM.oriented_box(_cuboid(c, half, R, rng))
    tris = torch.from_numpy(verts[SM._BOX_FACES]).float()
    p = rng.uniform(-0.6, 0.6, (20000, 3))
    local = (p - c) @ R
    margin = np.abs(np.abs(local) - half).min(1)            # distance to the nearest face plane
    inside = (np.abs(local) < half).all(1)
    got = SM.points_in_mesh(torch.from_numpy(p).float(), tris).numpy()
    clear = margin > 1e-4
    assert np.array_equal(got[clear], inside[clear]) and 300 < inside.sum() < 18000


@pytest.mark.parametrize("bound_type", ["to", "from", "both"])
def test_bbox_mapper_inverts_the_edit(bound_type):
    rng = np.random.default_rng(2)
    R0, c0, half = _rot([0, 0, 1], 20.0), np.array([-0.2, 0.0, 0.1]), np.array([0.15, 0.1, 0.2])
    T = np.eye(4); T[:3, :3] = _rot([0, 1, 0], 30.0); T[:3, 3] = [0.45, 0.05, -0.1]
    scale = np.array([1.5, 1.0, 0.8])
    cfg = {"type": "bbox", "raw": _cuboid(c0, half, R0, rng).tolist(), "transform": T.tolist(), "scale": scale.tolist(),
           "boundType": bound_type, "hsv": [0.1, 0.0, 0.0]}
    m = SM.get_seal_mapper(cfg)
    # content at source position s ends up at  t = T (c + scale * (s - c)) : sample the target box, map back, compare
    s = (rng.uniform(-1, 1, (4000, 3)) * half * 0.98) @ R0.T + c0
    t = ((s - c0) * scale + c0) @ T[:3, :3].T + T[:3, 3]
    d = rng.standard_normal((4000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts, dirs = torch.from_numpy(t).float(), torch.from_numpy(d).float()
    far = torch.tensor([[0.9, 0.9, 0.9], [-0.9, 0.8, -0.7]])
    mp, md, mask = m.map_to_origin(torch.cat([pts, far]), torch.cat([dirs, dirs[:2]]))
    assert bool(mask[:4000].all()) or bound_type == "from"
    if bound_type != "from":
        assert not bool(mask[4000:].any())
        assert torch.allclose(mp[:4000], torch.from_numpy(s).float(), atol=2e-6)
        assert torch.allclose(md[:4000], dirs @ torch.from_numpy(T[:3, :3]).float(), atol=2e-6)   # R^-1 d == d R for a rotation
        assert torch.equal(mp[4000:], far)
    # bounds bookkeeping used by the trainer / pose generation
    assert m.map_data["force_fill_bound"].shape == (2, 2, 3) and m.map_triangles.shape[0] in (12, 24)
    assert float(m.map_data["pose_radius"]) == pytest.approx(float(np.linalg.norm(c0 - (T[:3, :3] @ c0 + T[:3, 3])) * 10), rel=1e-5)
    cols = torch.rand(50, 3)
    out = m.map_color(mp[:50], md[:50], cols)
    assert out.shape == cols.shape and not torch.allclose(out, cols)
