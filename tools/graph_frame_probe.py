"""Probe: a shard-size frame as ONE captured graph (begin + ITERS worst-case-sized iterations + finish, no host read-back) against the
mailbox-driven loop.  Usage: python tools/graph_frame_probe.py [size] [contexts] [iters]"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
from dnerf_amd import bench_scene, fused  # noqa: E402
from dnerf_amd.renderer import DeviceLoop, PipelinedDeviceLoop  # noqa: E402
from sdn_backend import lib, check, ptr  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 283
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sc = bench_scene.build_scene(H=size, W=size)
N = sc.rays_o.shape[0]
f = fused.FusedField(sc.model, sc.time)
ref = DeviceLoop(sc.model, f, N, "cuda").render(sc.rays_o, sc.rays_d, sc.time)
print("rays", N, "samples", ref["n_samples"], "iterations", len(ref["trace"]))
loops, graphs, streams = [], [], []
for k in range(K):
    lp = DeviceLoop(sc.model, f, N, "cuda")
    lp.bind(sc.rays_o, sc.rays_d, sc.time)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
        st = torch.cuda.current_stream().cuda_stream
        check(lib.sdn_render_begin(ctypes.byref(lp.ctx), st), "begin")
        for it in range(ITERS):
            check(lib.sdn_render_step_f16(ctypes.byref(lp.ctx), N, st), "step")
        check(lib.sdn_render_finish(ctypes.byref(lp.ctx), 1.0, ptr(lp.image_out), ptr(lp.depth_out), st), "finish")
    loops.append(lp); graphs.append(g); streams.append(s)
torch.cuda.synchronize()
for k in range(K):
    with torch.cuda.stream(streams[k]):
        graphs[k].replay()
torch.cuda.synchronize()
print("bit-identical image:", all(torch.equal(lp.image_out, ref["image"]) for lp in loops),
      "alive after", ITERS, "iterations:", [int(lp.buf["state"][0]) for lp in loops])

def run(frames, kk):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        with torch.cuda.stream(streams[i % kk]):
            graphs[i % kk].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3

import gc
gc.collect(); gc.disable()
for kk in range(1, K + 1):
    run(10, kk)
    print(f"graph frames, {kk} in flight: {run(60, kk):.4f} ms per frame")
pl = PipelinedDeviceLoop(sc.model, f, N, "cuda", contexts=3)
pl.render_frames([sc.rays_o] * 3, [sc.rays_d] * 3, sc.time)
torch.cuda.synchronize(); t0 = time.perf_counter()
pl.render_frames([sc.rays_o] * 60, [sc.rays_d] * 60, sc.time)
torch.cuda.synchronize()
print(f"mailbox-driven pipelined, 3 contexts: {(time.perf_counter() - t0) / 60 * 1e3:.4f} ms per frame")
