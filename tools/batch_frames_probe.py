"""Throughput probe: B frames' rays concatenated into one ray set per loop context (per-ray results do not depend on which rays share
a launch), optionally several such sets in flight.  Prints ms per FRAME.  Usage: python tools/batch_frames_probe.py B [contexts]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
from dnerf_amd import bench_scene, fused  # noqa: E402
from dnerf_amd.renderer import PipelinedDeviceLoop  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sets = int(sys.argv[3]) if len(sys.argv) > 3 else 12
sc = bench_scene.build_scene()
ro, rd = torch.cat([sc.rays_o] * B).contiguous(), torch.cat([sc.rays_d] * B).contiguous()
N = ro.shape[0]
f = fused.FusedField(sc.model, sc.time)
pl = PipelinedDeviceLoop(sc.model, f, N, "cuda", contexts=K)
pl.render_frames([ro] * K, [rd] * K, sc.time)
import gc
gc.collect(); gc.disable()
torch.cuda.synchronize()
t0 = time.perf_counter()
pl.render_frames([ro] * sets, [rd] * sets, sc.time)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"frames per set {B}, contexts {K}: {dt / (sets * B) * 1e3:.4f} ms per frame")
