"""Where does the first march of a frame (640 000 rays, one step each) spend its time?  Variants of the same launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
import raymarching  # noqa: E402
from dnerf_amd import bench_scene  # noqa: E402

sc = bench_scene.build_scene()
m = sc.model
N = sc.rays_o.shape[0]
bit = m.density_bitfield[m.time_slice(sc.time)].contiguous()
nears, fars = raymarching.near_far_from_aabb(sc.rays_o, sc.rays_d, m.aabb_infer, m.min_near)
alive = torch.arange(N, dtype=torch.int32, device="cuda")


def run(name, bitfield, n_step=1, cull=True, rays=None, live=True, reps=10):
    ro, rd = (sc.rays_o, sc.rays_d) if rays is None else rays
    cg = raymarching.build_cull_grid(bitfield) if cull else None
    t = nears.clone()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        out = raymarching.march_rays_ex(N, n_step, alive, t, ro, rd, m.bound, bitfield, m.cascade, m.grid_size, fars, cull_grid=cg, want_live_list=live)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
    print(f"{name:58s} {ms * 1e3:8.1f} us" + (f"   live samples {int(out[4])}" if live else ""))


run("first march as in the frame (cull grid, live list)", bit)
run("  without the live list", bit, live=False)
run("  without the cull grid", bit, cull=False)
run("  empty occupancy (every ray exits at the cull test)", torch.zeros_like(bit))
run("  full occupancy (every ray samples at once)", torch.full_like(bit, 255))
perm = torch.randperm(N, device="cuda")
run("  rays in random order", bit, rays=(sc.rays_o[perm].contiguous(), sc.rays_d[perm].contiguous()))
run("  8 steps per ray", bit, n_step=8)
