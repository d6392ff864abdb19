"""Reads the in-kernel cycle stamps of the cooperative marcher from the DIAGNOSTIC library (make -C seald-nerf_amd/csrc diag;
SDN_LIB_PATH=seald-nerf_amd/lib/libsdn_hip_diag.so): renders a few frames one at a time and prints, per kernel, the shader cycles per
workgroup and phase, tasks per workgroup, windows per task.  Stamps exist in the diagnostic build only; never quote its run time."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SDN_LIB_PATH", os.path.join(ROOT, "seald-nerf_amd", "lib", "libsdn_hip_diag.so"))
for p in (ROOT, os.path.join(ROOT, "seald-nerf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    import sdn_backend
    from dnerf_amd.bench_scene import build_scene, camera_path
    from dnerf_amd import fused
    from dnerf_amd.renderer import DeviceLoop
    lib = sdn_backend.lib
    lib.sdn_debug_stamps.argtypes = [ctypes.c_void_p]
    sc = build_scene(H=800, W=800, device="cuda", seed=0)
    cam_o, cam_d, cam_t = camera_path(sc, 4, torch.device("cuda"))
    field = fused.FusedField(sc.model, sc.time, fp16=True)
    loop = DeviceLoop(sc.model, field, sc.rays_o.shape[0], "cuda", keep_cull_grids=True)
    buf = (ctypes.c_ulonglong * 32)()
    for i in range(2):
        loop.render(cam_o[i], cam_d[i], cam_t[i], want_stats=False)
    torch.cuda.synchronize()
    lib.sdn_debug_stamps(buf)
    frames = 4
    for i in range(frames):
        loop.render(cam_o[i], cam_d[i], cam_t[i], want_stats=False)
    torch.cuda.synchronize()
    lib.sdn_debug_stamps(buf)
    v = list(buf)
    for name, b in (("k_composite_march_g", 0), ("k_march_rays_g", 8)):
        wg = max(v[b], 1)
        print(json.dumps({"kernel": name, "frames": frames, "workgroups_with_tasks_per_frame": v[b] / frames,
                          "cycles_per_wg": {"phase_A": v[b + 1] / wg, "fine_cache": v[b + 2] / wg, "phase_B": v[b + 3] / wg, "live_append": v[b + 4] / wg},
                          "tasks_per_wg": v[b + 5] / wg, "windows_per_task": v[b + 6] / max(v[b + 5], 1), "slow_windows": v[b + 7],
                          "phase_B_cycles_per_pass": v[b + 3] / max(1.0, (v[b + 5] / wg + 15) // 16 * wg) if v[b + 5] else None,
                          "phase_B_cycles_per_window_pass": None}))


if __name__ == "__main__":
    main()
