"""Speed of the fp32 fused field kernel (csrc/field_f32.hip) next to the op-by-op fp32 network (hipBLASLt GEMMs + the encoder
operators) and the fp16 fused kernel, on probe points of the bench scene; and an 800x800 frame through the host-stepped loop."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
import numpy as np
import torch
from dnerf_amd.bench_scene import build_scene, _probe_points
from dnerf_amd import fused
from dnerf_amd.fused_f32 import FusedFieldF32
from dnerf_amd.renderer import render_frame

FLOP = 235520.0
sc = build_scene(H=800, W=800, device="cuda", seed=0)
model = sc.model.eval()
out = {}
for n in (65536, 262144, 1048576):
    x = torch.from_numpy(_probe_points(sc.bitfield, n, 3)).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, device="cuda"), dim=1).contiguous()
    f32, f16 = FusedFieldF32(model, sc.time, max_points=n, variant="mfma32"), fused.FusedField(model, sc.time, max_points=n)
    f32s = FusedFieldF32(model, sc.time, max_points=n, variant="split")

    def ops():
        model.fused_inference = False
        with torch.no_grad():
            return model(x, d, sc.time)

    row = {}
    for name, fn, reps in (("fused_f32_split", lambda: f32s(x, d), 30), ("fused_f32", lambda: f32(x, d), 20), ("fused_f16", lambda: f16(x, d), 50), ("op_by_op_f32", ops, 5)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
        row[name] = {"ms": round(ms, 4), "TFLOP/s": round(FLOP * n / ms / 1e9, 1)}
    out[str(n)] = row
    print(n, json.dumps(row), flush=True)
frames = {}
for name, kw in (("fused_f32_split", dict(fp16=False, field=FusedFieldF32(model, sc.time, variant="split"))),
                 ("fused_f32", dict(fp16=False, field=FusedFieldF32(model, sc.time, variant="mfma32"))), ("op_by_op_f32", dict(fp16=False)),
                 ("fused_f16", dict(fp16=True, field=fused.FusedField(model, sc.time)))):
    for _ in range(2):
        render_frame(model, sc.rays_o, sc.rays_d, sc.time, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        render_frame(model, sc.rays_o, sc.rays_d, sc.time, **kw)
    torch.cuda.synchronize(); frames[name] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
print("800x800 frame, host-stepped loop, ms:", json.dumps(frames))
