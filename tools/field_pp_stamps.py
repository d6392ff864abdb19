"""Slot timeline of the persistent two-set field kernel (csrc/field_pp.inc) from the DIAGNOSTIC library's stamps (make -C seald-nerf_amd/csrc
diag; never quote its run time).  Per launch size: median cycles per slot of the X role (layers D0..D6) and of the Y role (chunks 0..6),
for set 0 and set 1, over the workgroups' middle periods; period length; prologue (entry -> resident weights landed).

    python tools/field_pp_stamps.py [--points 431616 863232] [--frame-samples NSTEP]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SDN_LIB_PATH", os.path.join(ROOT, "seald-nerf_amd", "lib", "libsdn_hip_diag.so"))
for p in (ROOT, os.path.join(ROOT, "seald-nerf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, nargs="+", default=[431616, 863232])
    ap.add_argument("--frame-samples", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    import torch
    import sdn_backend
    from dnerf_amd import bench_scene, fused, scene
    lib = sdn_backend.lib
    lib.sdn_debug_pp_stamps.argtypes = [ctypes.c_void_p]
    raw = np.zeros(256 * 2 * 128, dtype=np.uint64)
    cases = []
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
        cases.append(("frame%d" % args.frame_samples, model, x[keep].contiguous(), d[keep].contiguous()))
    else:
        model = bench_scene.build_model(seed=0)
        bf = scene.jumpingjacks_occupancy(0.5)
        for n in args.points:
            xyz = torch.from_numpy(bench_scene._probe_points(bf, n, 1)).cuda()
            rng = np.random.default_rng(2)
            dd = rng.standard_normal((n, 3)).astype(np.float32)
            dd /= np.linalg.norm(dd, axis=1, keepdims=True)
            cases.append(("random", model, xyz, torch.from_numpy(dd).cuda()))
    for name, model, xyz, dirs in cases:
        n = xyz.shape[0]
        f = fused.FusedField(model, torch.tensor([[0.5]], device="cuda"), max_points=n)
        for _ in range(3):
            f(xyz, dirs)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f(xyz, dirs)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1000 / 20
        lib.sdn_debug_pp_stamps(raw.ctypes.data)
        f(xyz, dirs)
        lib.sdn_debug_pp_stamps(raw.ctypes.data)
        st = raw.reshape(256, 2, 128).astype(np.int64)
        used = st[:, 0, 0] > 0
        st = st[used]
        rec = {"points": n, "samples": name, "workgroups": int(used.sum()), "us_per_launch_diag_build": round(us, 2)}
        rec["prologue_cycles_median"] = int(np.median(st[:, 0, 1] - st[:, 0, 0]))
        # slots: stamp 2 + 7 p + s is taken at the END of slot s of period p; its length = the difference to the previous stamp
        for sset in (0, 1):
            a = st[:, sset, :]
            n_st = int((a[0] > 0).sum())
            n_per = (n_st - 2) // 7
            d = np.diff(a[:, 1:2 + 7 * n_per], axis=1).reshape(a.shape[0], n_per, 7)      # [wg, period, slot]
            # role of set s in period p: r = p - s; r < 0 idle; r even: Y; r odd: X
            xs = [p for p in range(n_per) if (p - sset) >= 0 and (p - sset) % 2 == 1]
            ys = [p for p in range(n_per) if (p - sset) >= 2 and (p - sset) % 2 == 0 and p < n_per - 1]
            if xs:
                rec[f"set{sset}_X_slot_cycles"] = [int(np.median(d[:, xs, k])) for k in range(7)]
            if ys:
                rec[f"set{sset}_Y_chunk_cycles"] = [int(np.median(d[:, ys, k])) for k in range(7)]
            rec[f"set{sset}_periods"] = n_per
            rec[f"set{sset}_period_cycles_median"] = int(np.median(d[:, 1:max(2, n_per - 1), :].sum(axis=2)))
        rec["span_cycles"] = int(st[:, :, 2:].max() - st[:, 0, 0].min())
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
