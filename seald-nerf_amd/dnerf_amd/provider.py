"""D-NeRF dataset provider ("next" row 8(f)3): `transforms_*.json` + images -> poses, frame times, intrinsics and ray batches.

Same constructor, attributes and batch dictionaries as the reference's `dnerf/provider.py:93-361` (`NeRFDataset(opt, device, type,
downscale, n_test)`, `.collate(index)`, `.dataloader()`), so its trainer can iterate it unchanged.  Differences:
  * images are read with PIL (cv2 / imageio are not dependencies); RGBA stays RGBA.  Down-scaling is a per-channel block mean, which
    is what `cv2.INTER_AREA` computes for integer factors (the only case `downscale` produces); non-integer ratios: parity unpinned;
  * rays come from `dnerf_amd.utils.get_rays` (one HIP kernel for a whole frame on the device);
  * no trimesh pose visualiser.
"""
import glob
import json
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from .utils import get_rays


def nerf_matrix_to_ngp(pose, scale=0.33, offset=(0, 0, 0)):
    """Blender / NeRF camera-to-world -> the renderer's frame (dnerf/provider.py:18-26): axes (x,y,z) -> (y,z,x), camera y and z
    flipped, translation scaled and offset."""
    p = np.asarray(pose, dtype=np.float32)
    out = np.eye(4, dtype=np.float32)
    for row, src in enumerate((1, 2, 0)):
        out[row, 0], out[row, 1], out[row, 2] = p[src, 0], -p[src, 1], -p[src, 2]
        out[row, 3] = p[src, 3] * scale + offset[row]
    return out


def rand_poses(size, device, radius=1, theta_range=(np.pi / 3, 2 * np.pi / 3), phi_range=(0, 2 * np.pi)):
    """Random orbit cameras looking at the origin (dnerf/provider.py:56-90) -> [size, 4, 4]."""
    unit = lambda v: v / (torch.norm(v, dim=-1, keepdim=True) + 1e-10)  # noqa: E731
    thetas = torch.rand(size, device=device) * (theta_range[1] - theta_range[0]) + theta_range[0]
    phis = torch.rand(size, device=device) * (phi_range[1] - phi_range[0]) + phi_range[0]
    centers = torch.stack([radius * torch.sin(thetas) * torch.sin(phis), radius * torch.cos(thetas),
                           radius * torch.sin(thetas) * torch.cos(phis)], dim=-1)
    forward = -unit(centers)
    up = torch.tensor([0.0, -1.0, 0.0], device=device).expand(size, 3)
    right = unit(torch.cross(forward, up, dim=-1))
    up = unit(torch.cross(right, forward, dim=-1))
    poses = torch.eye(4, dtype=torch.float32, device=device).repeat(size, 1, 1)
    poses[:, :3, :3] = torch.stack((right, up, forward), dim=-1)
    poses[:, :3, 3] = centers
    return poses


def _frame_time(frame):
    """'time' if the frame has one, else the integer in its file name (dnerf/provider.py:236-239)."""
    return frame["time"] if "time" in frame else int(os.path.basename(frame["file_path"])[:-4])


def _open_image(path):
    """uint8 [h, w, 3] or, when the file carries alpha (the mask channel, dnerf/provider.py:225-229), [h, w, 4]."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("RGB", "RGBA"):
            im = im.convert("RGBA" if ("A" in im.getbands() or "transparency" in im.info) else "RGB")
        return np.asarray(im, dtype=np.uint8)


def _area_resize(img, H, W):
    """cv2.INTER_AREA on 8-bit data: every channel on its own (alpha is NOT premultiplied, unlike PIL's RGBA resize), block means
    rounded to the nearest integer.  Integer reduction factors -- all that `downscale` produces -- are exact block means; anything
    else goes through PIL's box filter channel by channel (parity unpinned)."""
    h, w, c = img.shape
    if h % H == 0 and w % W == 0:
        blocks = img.reshape(H, h // H, W, w // W, c).astype(np.float64).mean(axis=(1, 3))
        return np.rint(blocks).astype(np.uint8)
    from PIL import Image
    return np.stack([np.asarray(Image.fromarray(img[..., k]).resize((W, H), Image.BOX)) for k in range(c)], axis=-1)


class NeRFDataset:
    def __init__(self, opt, device, type="train", downscale=1, n_test=10):
        self.opt, self.device, self.type, self.downscale = opt, device, type, downscale
        self.root_path, self.preload = opt.path, opt.preload
        self.scale, self.offset, self.bound, self.fp16 = opt.scale, opt.offset, opt.bound, opt.fp16
        self.training = type in ("train", "all", "trainval")
        self.num_rays = opt.num_rays if self.training else -1
        self.rand_pose = opt.rand_pose

        root = self.root_path
        if os.path.exists(os.path.join(root, "transforms.json")):
            self.mode = "colmap"       # one file, split by hand; the test set is a camera interpolation
        elif os.path.exists(os.path.join(root, "transforms_train.json")):
            self.mode = "blender"      # splits provided
        else:
            raise NotImplementedError(f"[NeRFDataset] Cannot find transforms*.json under {root}")
        transform = self._read_transforms(type)

        if "h" in transform and "w" in transform:
            self.H, self.W = int(transform["h"]) // downscale, int(transform["w"]) // downscale
        else:
            self.H = self.W = None     # taken from the first image
        frames = transform["frames"]

        if self.mode == "colmap" and type == "test":
            self._interpolated_test_set(frames, n_test)
        else:
            if self.mode == "colmap":  # the first frame is the validation set
                frames = frames[1:] if type == "train" else frames[:1] if type == "val" else frames
            self._load_frames(frames)

        self.poses = torch.from_numpy(np.stack(self.poses, axis=0))                        # [N,4,4]
        self.images = torch.from_numpy(np.stack(self.images, axis=0)) if self.images is not None else None   # [N,H,W,C]
        self.times = torch.from_numpy(np.asarray(self.times, dtype=np.float32)).view(-1, 1)                   # [N,1]
        if self.times.max() > 1:       # frame indices -> [0, 1]
            self.times = self.times / (self.times.max() + 1e-8)
        self.radius = self.poses[:, :3, 3].norm(dim=-1).mean(0).item()
        self.error_map = torch.ones([self.images.shape[0], 128 * 128], dtype=torch.float) if self.training and opt.error_map else None

        if self.preload:
            self.poses = self.poses.to(device)
            if self.images is not None:
                half = self.fp16 and opt.color_space != "linear"
                self.images = self.images.to(torch.half if half else torch.float).to(device)
            if self.error_map is not None:
                self.error_map = self.error_map.to(device)
            self.times = self.times.to(device)
        self.intrinsics = self._intrinsics(transform)

    # ------------------------------------------------------------------------------------------
    def _read_transforms(self, type):
        def load(name):
            with open(os.path.join(self.root_path, name), "r") as f:
                return json.load(f)
        if self.mode == "colmap":
            return load("transforms.json")
        if type == "all":              # every split in the directory
            merged = None
            for path in glob.glob(os.path.join(self.root_path, "*.json")):
                part = load(os.path.basename(path))
                if merged is None:
                    merged = part
                else:
                    merged["frames"].extend(part["frames"])
            return merged
        if type == "trainval":
            merged = load("transforms_train.json")
            merged["frames"].extend(load("transforms_val.json")["frames"])
            return merged
        return load(f"transforms_{type}.json")

    def _load_frames(self, frames):
        self.poses, self.images, self.times = [], [], []
        for f in frames:               # assumed sorted by time, as in the reference
            path = os.path.join(self.root_path, f["file_path"])
            if self.mode == "blender" and "." not in os.path.basename(path):
                path += ".png"
            if not os.path.exists(path):
                continue
            im = _open_image(path)
            if self.H is None or self.W is None:
                self.H, self.W = im.shape[0] // self.downscale, im.shape[1] // self.downscale
            if im.shape[:2] != (self.H, self.W):
                im = _area_resize(im, self.H, self.W)
            image = im.astype(np.float32) / 255                   # [H, W, 3|4]
            self.poses.append(nerf_matrix_to_ngp(np.array(f["transform_matrix"], dtype=np.float32), scale=self.scale, offset=self.offset))
            self.images.append(image)
            self.times.append(_frame_time(f))

    def _interpolated_test_set(self, frames, n_test):
        """colmap + test: n_test + 1 cameras on a slerp / lerp between two random frames, times interpolated alike (:163-194)."""
        from scipy.spatial.transform import Rotation, Slerp
        f0, f1 = np.random.choice(frames, 2, replace=False)
        p0, p1 = [nerf_matrix_to_ngp(np.array(f["transform_matrix"], dtype=np.float32), scale=self.scale, offset=self.offset) for f in (f0, f1)]
        t0, t1 = _frame_time(f0), _frame_time(f1)
        slerp = Slerp([0, 1], Rotation.from_matrix(np.stack([p0[:3, :3], p1[:3, :3]])))
        self.poses, self.images, self.times = [], None, []
        for i in range(n_test + 1):
            ratio = np.sin(((i / n_test) - 0.5) * np.pi) * 0.5 + 0.5
            pose = np.eye(4, dtype=np.float32)
            pose[:3, :3] = slerp(ratio).as_matrix()
            pose[:3, 3] = (1 - ratio) * p0[:3, 3] + ratio * p1[:3, 3]
            self.poses.append(pose)
            self.times.append((1 - ratio) * t0 + ratio * t1)
        if "time" not in f0:           # file-name times: normalise by the largest frame index
            top = max(int(os.path.basename(f["file_path"])[:-4]) for f in frames)
            self.times = [t / top for t in self.times]

    def _intrinsics(self, transform):
        """(fl_x, fl_y, cx, cy) at the loaded resolution (:287-303)."""
        ds = self.downscale
        if "fl_x" in transform or "fl_y" in transform:
            fl_x = (transform["fl_x"] if "fl_x" in transform else transform["fl_y"]) / ds
            fl_y = (transform["fl_y"] if "fl_y" in transform else transform["fl_x"]) / ds
        elif "camera_angle_x" in transform or "camera_angle_y" in transform:
            fl_x = self.W / (2 * np.tan(transform["camera_angle_x"] / 2)) if "camera_angle_x" in transform else None
            fl_y = self.H / (2 * np.tan(transform["camera_angle_y"] / 2)) if "camera_angle_y" in transform else None
            fl_x = fl_y if fl_x is None else fl_x
            fl_y = fl_x if fl_y is None else fl_y
        else:
            raise RuntimeError("Failed to load focal length, please check the transforms.json!")
        cx = (transform["cx"] / ds) if "cx" in transform else (self.W / 2)
        cy = (transform["cy"] / ds) if "cy" in transform else (self.H / 2)
        return np.array([fl_x, fl_y, cx, cy])

    # ------------------------------------------------------------------------------------------
    def collate(self, index):
        B = len(index)                 # the loader's batch size is 1: one frame per step
        if self.rand_pose == 0 or index[0] >= len(self.poses):   # a random camera, no ground truth (:309-322)
            poses = rand_poses(B, self.device, radius=self.radius)
            s = np.sqrt(self.H * self.W / self.num_rays)
            rH, rW = int(self.H / s), int(self.W / s)
            rays = get_rays(poses, self.intrinsics / s, rH, rW, -1)
            return {"H": rH, "W": rW, "rays_o": rays["rays_o"], "rays_d": rays["rays_d"]}
        poses = self.poses[index].to(self.device)
        times = self.times[index].to(self.device)
        error_map = None if self.error_map is None else self.error_map[index]
        rays = get_rays(poses, self.intrinsics, self.H, self.W, self.num_rays, error_map)
        out = {"time": times, "H": self.H, "W": self.W, "rays_o": rays["rays_o"], "rays_d": rays["rays_d"]}
        if self.images is not None:
            images = self.images[index].to(self.device)
            if self.training:
                C = images.shape[-1]
                images = torch.gather(images.view(B, -1, C), 1, torch.stack(C * [rays["inds"]], -1))
            out["images"] = images
        if error_map is not None:
            out["index"], out["inds_coarse"] = index, rays["inds_coarse"]
        return out

    def dataloader(self):
        size = len(self.poses)
        if self.training and self.rand_pose > 0:
            size += size // self.rand_pose         # indices past the data mean "random pose"
        loader = DataLoader(list(range(size)), batch_size=1, collate_fn=self.collate, shuffle=self.training, num_workers=0)
        loader._data = self                        # the reference trainer reaches error_map / poses through this
        loader.has_gt = self.images is not None
        return loader
