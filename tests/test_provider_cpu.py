"""D-NeRF dataset provider (dnerf_amd/provider.py) on synthetic datasets written to a temp directory: there is no dataset in the
reference tree, so expectations are computed by hand from the file contents (dnerf/provider.py:93-361 is the behaviour followed)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import tests_support  # noqa: F401


def _opt(path, **kw):
    base = dict(path=str(path), preload=False, scale=0.8, offset=[0.1, 0.2, 0.3], bound=1, fp16=False, num_rays=50, rand_pose=-1,
                error_map=False, color_space="srgb")
    base.update(kw)
    return SimpleNamespace(**base)


def _pose(k):
    a = 0.3 * k
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3], T[:3, 3] = R, [1.0 + k, 2.0, 3.0 - k]
    return T


def _write_blender(root, n=4, side=8, with_time=True):
    from PIL import Image
    rng = np.random.default_rng(0)
    images = []
    for split, count in (("train", n), ("val", 1), ("test", 2)):
        frames = []
        os.makedirs(root / split, exist_ok=True)
        for k in range(count):
            img = rng.integers(0, 256, (side, side, 4), dtype=np.uint8)
            # with times: blender naming without extension; without: "<frame index>.png", which the provider parses as the time
            name = f"r_{k:03d}" if with_time else f"{k:04d}.png"
            Image.fromarray(img, "RGBA").save(root / split / (name if name.endswith(".png") else name + ".png"))
            fr = {"file_path": f"./{split}/{name}", "transform_matrix": _pose(k).tolist()}
            if with_time:
                fr["time"] = k / max(count - 1, 1)
            frames.append(fr)
            if split == "train":
                images.append(img)
        with open(root / f"transforms_{split}.json", "w") as f:
            json.dump({"camera_angle_x": 0.6911, "frames": frames}, f)
    return images


def test_blender_split_poses_times_intrinsics_and_batches(tmp_path):
    from dnerf_amd.provider import NeRFDataset, nerf_matrix_to_ngp
    imgs = _write_blender(tmp_path)
    ds = NeRFDataset(_opt(tmp_path), "cpu", type="train")
    assert ds.mode == "blender" and (ds.H, ds.W) == (8, 8) and ds.training and ds.num_rays == 50
    assert ds.poses.shape == (4, 4, 4) and ds.images.shape == (4, 8, 8, 4) and ds.times.shape == (4, 1)
    # pose convention (dnerf/provider.py:18-26), by hand for frame 2
    p = _pose(2)
    want = np.array([[p[1, 0], -p[1, 1], -p[1, 2], p[1, 3] * 0.8 + 0.1], [p[2, 0], -p[2, 1], -p[2, 2], p[2, 3] * 0.8 + 0.2],
                     [p[0, 0], -p[0, 1], -p[0, 2], p[0, 3] * 0.8 + 0.3], [0, 0, 0, 1]], dtype=np.float32)
    assert np.array_equal(ds.poses[2].numpy(), want) and np.array_equal(nerf_matrix_to_ngp(p, 0.8, [0.1, 0.2, 0.3]), want)
    assert np.allclose(ds.times.view(-1).numpy(), [0, 1 / 3, 2 / 3, 1])
    assert np.array_equal(ds.images[1].numpy(), imgs[1].astype(np.float32) / 255)          # RGBA order kept
    f = 8 / (2 * np.tan(0.6911 / 2))
    assert np.allclose(ds.intrinsics, [f, f, 4, 4])
    assert abs(ds.radius - float(ds.poses[:, :3, 3].norm(dim=-1).mean())) < 1e-6
    torch.manual_seed(0)
    batch = ds.collate([2])
    assert batch["rays_o"].shape == (1, 50, 3) and batch["rays_d"].shape == (1, 50, 3) and batch["images"].shape == (1, 50, 4)
    assert batch["H"] == 8 and batch["W"] == 8 and float(batch["time"]) == pytest.approx(2 / 3)
    assert torch.allclose(batch["rays_d"].norm(dim=-1), torch.ones(1, 50), atol=1e-6)
    assert torch.equal(batch["rays_o"][0, 0], ds.poses[2, :3, 3])
    # the colours are the pixels the rays were drawn through (same RNG draw as get_rays)
    torch.manual_seed(0)
    inds = torch.randint(0, 64, size=[50])
    assert torch.equal(batch["images"][0], ds.images[2].view(64, 4)[inds])
    loader = ds.dataloader()
    assert len(loader) == 4 and loader.has_gt and loader._data is ds
    assert set(next(iter(loader))) == {"time", "H", "W", "rays_o", "rays_d", "images"}
    # evaluation split: whole frames
    val = NeRFDataset(_opt(tmp_path), "cpu", type="val")
    b = val.collate([0])
    assert not val.training and b["rays_o"].shape == (1, 64, 3) and b["images"].shape == (1, 8, 8, 4)
    assert len(NeRFDataset(_opt(tmp_path), "cpu", type="trainval").poses) == 5
    assert len(NeRFDataset(_opt(tmp_path), "cpu", type="all").poses) == 7


def test_downscale_error_map_random_poses_and_frame_index_times(tmp_path):
    from dnerf_amd.provider import NeRFDataset, rand_poses
    imgs = _write_blender(tmp_path, with_time=False)
    ds = NeRFDataset(_opt(tmp_path, error_map=True, rand_pose=2, num_rays=16), "cpu", type="train", downscale=2)
    assert (ds.H, ds.W) == (4, 4)
    area = imgs[0].astype(np.float64).reshape(4, 2, 4, 2, 4).mean(axis=(1, 3))                # 2x2 box average of the 8x8 image
    assert np.abs(ds.images[0].numpy() * 255 - area).max() <= 0.5 + 1e-6                       # (8-bit rounding of the resampler)
    assert np.allclose(ds.intrinsics, [4 / (2 * np.tan(0.6911 / 2))] * 2 + [2, 2])
    assert np.allclose(ds.times.view(-1).numpy(), np.arange(4) / (3 + 1e-8))                    # file-name indices, normalised by the max
    assert ds.error_map.shape == (4, 128 * 128)
    b = ds.collate([1])
    assert b["index"] == [1] and b["inds_coarse"].shape == (1, 16) and b["images"].shape == (1, 16, 4)
    loader = ds.dataloader()
    assert len(loader) == 4 + 4 // 2
    r = ds.collate([5])                                                                         # past the data: a random orbit camera
    assert set(r) == {"H", "W", "rays_o", "rays_d"} and r["rays_o"].shape == (1, r["H"] * r["W"], 3)
    assert abs(float(r["rays_o"][0, 0].norm()) - ds.radius) < 1e-4
    P = rand_poses(64, "cpu", radius=2.5)
    R = P[:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(64, 3, 3), atol=1e-5)     # orthonormal frames
    assert torch.allclose(P[:, :3, 3].norm(dim=-1), torch.full((64,), 2.5), atol=1e-5)
    assert torch.allclose(P[:, :3, 2], -P[:, :3, 3] / 2.5, atol=1e-5)                           # looking at the origin


def test_colmap_mode_splits_and_interpolated_test_cameras(tmp_path):
    from PIL import Image
    from dnerf_amd.provider import NeRFDataset
    frames = []
    for k in range(5):
        Image.fromarray(np.full((6, 10, 3), 40 * k, dtype=np.uint8), "RGB").save(tmp_path / f"{k:04d}.png")
        frames.append({"file_path": f"{k:04d}.png", "transform_matrix": _pose(k).tolist()})
    frames.append({"file_path": "0009.png", "transform_matrix": _pose(9).tolist()})           # (no such file) non-existent paths are skipped
    with open(tmp_path / "transforms.json", "w") as f:
        json.dump({"fl_x": 20.0, "cx": 5.5, "cy": 2.5, "h": 6, "w": 10, "frames": frames}, f)
    tr = NeRFDataset(_opt(tmp_path), "cpu", type="train")
    va = NeRFDataset(_opt(tmp_path), "cpu", type="val")
    assert tr.mode == "colmap" and len(tr.poses) == 4 and len(va.poses) == 1 and tr.images.shape == (4, 6, 10, 3)
    assert np.allclose(tr.intrinsics, [20, 20, 5.5, 2.5])
    assert np.allclose(tr.times.view(-1).numpy(), np.array([1, 2, 3, 4]) / (4 + 1e-8))
    np.random.seed(0)
    te = NeRFDataset(_opt(tmp_path), "cpu", type="test", n_test=6)
    assert te.images is None and te.poses.shape == (7, 4, 4) and te.times.shape == (7, 1)
    R = te.poses[:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(7, 3, 3), atol=1e-5)
    assert float(te.times.min()) >= 0 and float(te.times.max()) <= 1
    b = te.collate([3])
    assert "images" not in b and b["rays_o"].shape == (1, 60, 3)
    with pytest.raises(NotImplementedError):
        NeRFDataset(_opt(tmp_path / "nowhere"), "cpu", type="train")


def test_merged_param_groups_is_the_same_adam():
    """dnerf_amd/train_graph.merged_param_groups: groups with identical hyper-parameters folded together update identically."""
    from dnerf_amd.train_graph import merged_param_groups
    torch.manual_seed(0)
    def make():
        torch.manual_seed(1)
        return [torch.nn.Parameter(torch.randn(5)) for _ in range(5)]
    a, b = make(), make()
    def groups(p):
        return [{"params": [p[0]], "lr": 1e-2}, {"params": [p[1], p[2]], "lr": 1e-3}, {"params": [], "lr": 1e-2},
                {"params": [p[3]], "lr": 1e-3}, {"params": [p[4]], "lr": 1e-2}]
    merged = merged_param_groups(groups(b))
    assert sorted((g["lr"], len(g["params"])) for g in merged) == [(1e-3, 3), (1e-2, 2)]
    oa = torch.optim.Adam(groups(a), betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam(merged, betas=(0.9, 0.99), eps=1e-15)
    for step in range(3):
        for p, q in zip(a, b):
            g = torch.randn(5, generator=torch.Generator().manual_seed(10 * step + 3))
            p.grad, q.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    assert all(torch.equal(p, q) for p, q in zip(a, b))
