"""Builds the fully specified synthetic benchmark / test scene.

  * rays: 800x800 (or HxW) camera of `scene.py`, time = 0.5 (density-grid slice 32)
  * occupancy: the capsule figure ("jumpingjacks-like") or the studded box ("lego-like"), one Morton
    bitfield per time slice
  * network: `NeRFNetwork` initialised under torch.manual_seed(seed) on the CPU (so the weights are
    the same on every box), then, deterministically:
      - grid embeddings U(-1e-4, 1e-4) (grid.py:138-140) scaled by 1e3;
      - last deform layer scaled by 0.05 (random Kaiming weights would throw samples out of the unit cube;
        trained D-NeRF deformations are small);
      - row 0 of the last sigma layer made non-negative (its inputs are post-ReLU, so the density logit is
        positive) and scaled so that the median sigma*dt over probe points inside the figure is 0.05
        (no-bias nets: an additive shift is not available);
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import scene
from .network import NeRFNetwork

MEDIAN_SIGMA_DT = 0.05


def _probe_points(bitfield_slice, n, seed):
    """n points at the centres of occupied cells (Morton order -> xyz), seeded."""
    bits = np.unpackbits(bitfield_slice, bitorder="little")
    occ = np.nonzero(bits)[0]
    rng = np.random.default_rng(seed)
    idx = occ[rng.integers(0, occ.shape[0], n)].astype(np.uint32)

    def compact(v):
        v = v & np.uint32(0x49249249)
        v = (v | (v >> np.uint32(2))) & np.uint32(0xC30C30C3)
        v = (v | (v >> np.uint32(4))) & np.uint32(0x0F00F00F)
        v = (v | (v >> np.uint32(8))) & np.uint32(0xFF0000FF)
        v = (v | (v >> np.uint32(16))) & np.uint32(0x0000FFFF)
        return v

    c = np.stack([compact(idx), compact(idx >> np.uint32(1)), compact(idx >> np.uint32(2))], 1).astype(np.float32)
    return ((c + 0.5) * (2.0 / 128) - 1.0).astype(np.float32)


def build_model(seed=0, device="cuda", bound=1, bg_radius=-1):
    torch.manual_seed(seed)
    model = NeRFNetwork(bound=bound, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=bg_radius)
    with torch.no_grad():
        model.encoder.embeddings.mul_(1e3)
        model.deform_net[-1].weight.mul_(0.05)
        model.sigma_net[-1].weight[0].abs_()
        if bg_radius > 0:
            model.encoder_bg.embeddings.mul_(1e3)      # as for the density grid: U(-1e-4, 1e-4) features would give a constant grey
    return model.to(device).eval()


@torch.no_grad()
def calibrate_density(model, bitfield_slice, time, seed=0, max_steps=1024):
    """Scales row 0 of the last sigma layer so that median(sigma) * dt_min == MEDIAN_SIGMA_DT on the probe set."""
    dt = 2 * math.sqrt(3) / max_steps
    pts = torch.from_numpy(_probe_points(bitfield_slice, 8192, seed)).to(model.encoder.embeddings.device)
    deform = model._deform(pts, time)
    h = model.encoder(pts + deform, bound=model.bound)
    for layer in model.sigma_net[:-1]:
        h = torch.relu(layer(h))
    logit = h @ model.sigma_net[-1].weight[0]
    med = float(logit.median())
    assert med > 0, "density logit must be positive on the probe set"
    model.sigma_net[-1].weight[0].mul_(math.log(MEDIAN_SIGMA_DT / dt) / med)
    return med


def build_scene(H=800, W=800, device="cuda", seed=0, kind="jumpingjacks", time=0.5, azimuth=30.0, elevation=30.0, bound=1):
    """bound > 1 (main_dnerf.py --bound 2): cascade = 1 + ceil(log2(bound)) occupancy grids per time slice, the same figure."""
    model = build_model(seed, device, bound)
    t_idx = int(min(max(math.floor(time * model.time_size), 0), model.time_size - 1))
    bits = scene.density_bitfield_cascades(model.time_size, model.grid_size, model.cascade, float(bound), kind)   # every time slice
    model.density_bitfield.copy_(torch.from_numpy(bits))
    time_t = torch.tensor([[time]], dtype=torch.float32, device=device)
    calibrate_density(model, bits[t_idx][: model.grid_size ** 3 // 8], time_t, seed)
    pose = scene.look_at_pose(azimuth, elevation)
    ro, rd = scene.get_rays(pose, scene.intrinsics(H, W), H, W)
    return SimpleNamespace(model=model, rays_o=torch.from_numpy(ro).to(device), rays_d=torch.from_numpy(rd).to(device), time=time_t,
                           H=H, W=W, kind=kind, t_idx=t_idx, bitfield=bits[t_idx], bitfields=bits, pose=pose)


def camera_path(sc, n_frames, device=None, elevation=30.0):
    """A D-NeRF test sequence of `n_frames` frames of the scene `sc`: the camera orbits once (azimuth 30 + 360 f / n) while the time
    runs from 0 to 1 (frame 0 is the canonical frame t == 0, the last one t == 1) -- what the reference's test set is: one pose and
    one time stamp per frame (dnerf/provider.py test split, dnerf/utils.py:151-161).  -> (rays_o list, rays_d list, times list)."""
    device = device if device is not None else sc.rays_o.device
    ros, rds, times = [], [], []
    intr = scene.intrinsics(sc.H, sc.W)
    for f in range(n_frames):
        pose = scene.look_at_pose(30.0 + 360.0 * f / n_frames, elevation)
        ro, rd = scene.get_rays(pose, intr, sc.H, sc.W)
        ros.append(torch.from_numpy(ro).to(device))
        rds.append(torch.from_numpy(rd).to(device))
        times.append(float(f) / max(n_frames - 1, 1))
    return ros, rds, times
