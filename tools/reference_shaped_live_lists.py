"""The reference-shaped 800x800 frame (run_cuda over NeRFNetwork.forward, dnerf/renderer.py:350-376) with and without the marcher's
live-sample lists, under -O (fp16 fused kernel) and without (fp32 fused kernel): ms per frame, slots launched and samples listed."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
import torch
from dnerf_amd.bench_scene import build_scene, camera_path
import raymarching
import sdn_backend

dev = torch.device("cuda:0")
sc = build_scene(H=800, W=800, device=dev, seed=0, kind="jumpingjacks")
cam_o, cam_d, cam_t = camera_path(sc, 20, dev)
model = sc.model.eval()


def frame(i, fp32):
    t = torch.tensor([[cam_t[i]]], dtype=torch.float32, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=not fp32):
        return model.render(cam_o[i][None], cam_d[i][None], t, staged=False, perturb=False, bg_color=1, max_steps=1024)


def run(fp32, on, k=6):
    raymarching.live_lists.update(on=on, pinned=True)
    model.fused_inference_f32 = fp32
    for i in range(3):
        img = frame(i, fp32)["image"]
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for i in range(k):
            frame(i, fp32)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3 / k)
    log = []
    with sdn_backend.launch_log(log):
        frame(0, fp32)
    slots = sum(u for n, u in log if n.startswith("field_forward"))
    iters = sum(1 for n, u in log if n.startswith("field_forward"))
    model.fused_inference_f32 = False
    return best, slots, iters, img


out = {}
for fp32 in (False, True):
    a = run(fp32, False)
    b = run(fp32, True)
    out["fp32" if fp32 else "fp16 (-O)"] = {"ms_per_frame_whole_slots": round(a[0], 3), "ms_per_frame_live_lists": round(b[0], 3),
                                            "slots_per_frame": a[1], "iterations": a[2], "images_identical": bool(torch.equal(a[3], b[3]))}
print(json.dumps(out))
