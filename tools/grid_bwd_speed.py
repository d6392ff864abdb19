"""grid_encode backward timing: a ray-ordered training batch (neighbouring samples share coarse cells) and a large unordered batch.
Usage: python tools/grid_bwd_speed.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
from gridencoder import GridEncoder  # noqa: E402

enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                  gridtype="tiled", align_corners=False).cuda()
g = torch.Generator(device="cuda").manual_seed(0)


def ray_batch(n_rays, per_ray):
    o = torch.rand(n_rays, 1, 3, device="cuda", generator=g) * 0.4 + 0.3
    d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, device="cuda", generator=g), dim=-1)
    t = torch.arange(per_ray, device="cuda").view(1, -1, 1) * (3.383e-3 / 2)
    return (o + d * t).clamp(0, 1).reshape(-1, 3).contiguous()


def timeit(x, dtype, reps=20):
    x = x.requires_grad_(False)
    with torch.autocast("cuda", dtype=torch.float16, enabled=dtype == torch.float16):
        y = enc(x * 2 - 1, bound=1)
    gy = torch.randn_like(y)
    for _ in range(3):
        enc.embeddings.grad = None
        y.backward(gy, retain_graph=True)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    import sdn_backend
    tm = sdn_backend.KernelTimers()
    sdn_backend.timers = tm
    for _ in range(reps):
        enc.embeddings.grad = None
        y.backward(gy, retain_graph=True)
    sdn_backend.timers = None
    torch.cuda.synchronize()
    return {k: round(v["avg_ms"] * 1e3, 1) for k, v in tm.summary().items()}


for name, x in (("ray-ordered 9 000 (600 rays x 15)", ray_batch(600, 15)), ("ray-ordered 262 144 (4096 x 64)", ray_batch(4096, 64)),
                ("uniform 2 097 152", torch.rand(1 << 21, 3, device="cuda", generator=g))):
    for dt in (torch.float16, torch.float32):
        print(name, str(dt).split(".")[-1], "us per launch (incl. input-gradient kernel if any):", timeit(x, dt))
