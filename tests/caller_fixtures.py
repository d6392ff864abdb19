"""Rebuilds, on the test side, the scene behind tests/golden/caller_*.npz (fixtures produced by executing the reference's own
caller code over oracle-backed operator shims -- tests/golden/gen_caller_fixtures.py) and checks that it is the same scene."""
import hashlib
import math
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, f"caller_{name}.npz"))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fixture_model(device="cpu", check=True, bound=1, bg_radius=-1):
    """The mirror network with the fixture's weights: bench_scene.build_model(seed 0) + the calibrated sigma row stored by the
    generator; occupancy slices of the three fixture times.  With `check`, every state-dict entry must hash to the digest the
    generator recorded from the REFERENCE's NeRFNetwork (same names, same shapes, same values)."""
    from dnerf_amd import scene
    from dnerf_amd.bench_scene import build_model
    fx0 = load("scene")
    # the bound-2 model (cascade 2) and the model with a background sphere have their own digests / calibrated rows
    fx = load("bg") if bg_radius > 0 else (fx0 if bound == 1 else load(f"bound{bound}"))
    model = build_model(int(fx0["seed"]), "cpu", bound, bg_radius)
    slices = {int(min(max(math.floor(float(t) * model.time_size), 0), model.time_size - 1)) for t in fx0["times"]}
    bits = scene.density_bitfield_cascades(model.time_size, model.grid_size, model.cascade, float(bound), "jumpingjacks", times=slices)
    with torch.no_grad():
        model.density_bitfield.copy_(torch.from_numpy(bits))
        model.sigma_net[-1].weight[0].copy_(torch.from_numpy(fx["sigma_last_row0"]))
    if check:
        want = dict(zip(fx["digest_keys"].tolist(), fx["digest_vals"].tolist()))
        got = {k: _sha(v.detach().numpy()) for k, v in model.state_dict().items() if not k.startswith("density_grid")}
        assert set(want) == set(got), (sorted(set(want) ^ set(got)))
        bad = [k for k in want if want[k] != got[k]]
        assert not bad, f"state differs from the reference-built network: {bad}"
    return model.to(device).eval(), bits


def fixture_scene(device="cpu", H=64, W=64, time=0.5, azimuth=30.0, elevation=30.0, model_bits=None):
    from dnerf_amd import scene
    model, bits = model_bits if model_bits is not None else fixture_model(device)
    ro, rd = scene.get_rays(scene.look_at_pose(azimuth, elevation), scene.intrinsics(H, W), H, W)
    t_idx = int(min(max(math.floor(time * model.time_size), 0), model.time_size - 1))
    return SimpleNamespace(model=model, rays_o=torch.from_numpy(ro).to(device), rays_d=torch.from_numpy(rd).to(device),
                           time=torch.tensor([[time]], dtype=torch.float32, device=device), H=H, W=W, t_idx=t_idx, bitfield=bits[t_idx])


def fill_bitfield_host(bits, bounds, H=128, bound=1.0):
    """numpy form of dnerf_amd.seal_mapper.fill_bitfield (cells of cascade 0 whose centre lies strictly inside one of `bounds`
    [B,2,3], OR-ed into every time slice) -- the occupancy the edit fixtures were rendered with."""
    from dnerf_amd import scene
    c = (np.arange(H, dtype=np.float32) + np.float32(0.5)) * np.float32(2.0 * bound / H) - np.float32(bound)
    inside = np.zeros((H, H, H), bool)
    for lo, hi in np.asarray(bounds, np.float32).reshape(-1, 2, 3):
        m = [(c > lo[a]) & (c < hi[a]) for a in range(3)]
        inside |= m[0][:, None, None] & m[1][None, :, None] & m[2][None, None, :]
    ix, iy, iz = np.nonzero(inside)
    flat = np.zeros(H * H * H, np.uint8)
    flat[scene.morton3d(ix, iy, iz)] = 1
    packed = np.packbits(flat.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
    out = bits.copy()
    out[:, : packed.shape[0]] |= packed[None]
    return out
