"""The shader clock the power management grants WHILE the fused field kernel runs (`sdn_debug_shader_clock`: one idling wave on a side
stream that compares s_memtime ticks with the 100 MHz wall clock), next to the clock of an otherwise idle device.  Nominal peaks are
quoted at 2.4 GHz; a matrix-heavy kernel runs well below it (profiles/r04_mfma_roof_probe.txt)."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
import torch
import sdn_backend
from sdn_backend import check, ptr
from dnerf_amd.bench_scene import build_scene
from dnerf_amd import fused, fused_f32

dev = torch.device("cuda:0")
sc = build_scene(H=64, W=64, device=dev, seed=0)
side = torch.cuda.Stream()
out = torch.zeros(4, dtype=torch.int64, device=dev)


def watch(load, ms=20.0):
    """MHz of the shader clock over `ms` of wall time while `load()` keeps the main stream busy."""
    torch.cuda.synchronize()
    load(warm=True)
    torch.cuda.synchronize()
    load()                                   # enqueue well over `ms` of work
    with torch.cuda.stream(side):
        check(sdn_backend.lib.sdn_debug_shader_clock(ptr(out), int(ms * 1e5), side.cuda_stream), "shader_clock")
    load()
    torch.cuda.synchronize()
    o = out.cpu().tolist()
    return round(o[0] / o[1] * 100.0, 1)


M = 431616
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(M, 3, device=dev, generator=g) - 0.5
d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev, generator=g), dim=1)
f16 = fused.FusedField(sc.model, sc.time, fp16=True)
f32 = fused_f32.FusedFieldF32(sc.model, sc.time)


def idle(warm=False):
    pass


def load16(warm=False):
    for _ in range(4 if warm else 200):      # ~0.13 ms per launch
        f16(x, d)


def load32(warm=False):
    for _ in range(2 if warm else 25):       # ~1 ms per launch
        f32(x, d)


res = {"idle_device_mhz": watch(idle), "under_field_forward_f16_mhz": watch(load16), "under_field_forward_f32_mhz": watch(load32),
       "idle_again_mhz": watch(idle), "points_per_launch": M,
       "note": "one idling wave on a side stream (sdn_debug_shader_clock), 20 ms of wall clock per figure; the main stream holds back-to-back launches"}
print(json.dumps(res))
