"""Launch-level timing of the fused field kernel on a fixed batch of sample points (the size one loop iteration of the
800x800 frame hands it), outside the render loop.  Prints one JSON line.  Usage: python tools/field_microbench.py [--points N]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
from dnerf_amd import bench_scene, fused, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=126976)    # 496 workgroups of 256: one iteration of the headline frame
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--frame-samples", type=int, default=0, metavar="NSTEP",
                    help="use the live samples of the first NSTEP marching steps of the 800x800 headline frame (ray order) instead of random cells")
    args = ap.parse_args()
    if args.frame_samples:
        import raymarching
        sc = bench_scene.build_scene()
        model = sc.model
        N = sc.rays_o.shape[0]
        nears, fars = raymarching.near_far_from_aabb(sc.rays_o, sc.rays_d, model.aabb_infer, 0.2)
        alive = torch.arange(N, dtype=torch.int32, device="cuda")
        x, d, dl = raymarching.march_rays(N, args.frame_samples, alive, nears.clone(), sc.rays_o, sc.rays_d, model.bound,
                                          model.density_bitfield[sc.t_idx], model.cascade, model.grid_size, nears, fars, 128, False, 0, 1024)
        keep = dl[:, 0] > 0
        xyz, dirs = x[keep].contiguous(), d[keep].contiguous()
        args.points = int(xyz.shape[0])
    else:
        model = bench_scene.build_model(seed=0)
        bf = scene.jumpingjacks_occupancy(0.5)
        xyz = torch.from_numpy(bench_scene._probe_points(bf, args.points, 1)).cuda()
        rng = np.random.default_rng(2)
        d = rng.standard_normal((args.points, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        dirs = torch.from_numpy(d).cuda()
    f = fused.FusedField(model, torch.tensor([[0.5]], device="cuda"), max_points=args.points)
    for _ in range(5):
        f(xyz, dirs)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(args.iters):
        f(xyz, dirs)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / args.iters
    sig, rgb = f(xyz, dirs)
    print(json.dumps({"points": args.points, "us_per_launch": round(us, 2), "tflops": round(235520 * args.points / us / 1e6, 1),
                      "checksum": [float(sig.double().sum()), float(rgb.double().sum())]}), flush=True)


if __name__ == "__main__":
    main()
