"""Per-phase shader cycles of the fused field kernel from the DIAGNOSTIC library's in-kernel stamps (make -C seald-nerf_amd/csrc diag;
stamps exist in that build only -- never quote its run time).  For each launch size: the average cycles a workgroup's wave 0 and
wave 7 spend in each phase of k_field_f16 (see g_field_stamps in csrc/field.hip), and the launch time.

    python tools/field_stamps.py [--points 63488 126976 431573] [--variant throughput|latency|auto]
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

PHASES = ["entry->freq", "stage barrier", "D0", "D1", "D2", "D3", "D4", "D5", "D6", "conv+tail barrier", "D7+deform", "grid", "sigma net",
          "SH+colour net", "sigmoid+store"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, nargs="+", default=[63488, 126976, 431616])
    ap.add_argument("--variant", default="throughput")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    if args.variant != "auto":
        os.environ["SDN_FIELD_VARIANT"] = args.variant
    import numpy as np
    import torch
    import sdn_backend
    from dnerf_amd import bench_scene, fused, scene
    lib = sdn_backend.lib
    lib.sdn_debug_field_stamps.argtypes = [ctypes.c_void_p]
    lib.sdn_debug_field_stamp_words.restype = ctypes.c_uint32
    words = int(lib.sdn_debug_field_stamp_words())
    raw = np.zeros(words, dtype=np.uint64)
    model = bench_scene.build_model(seed=0)
    bf = scene.jumpingjacks_occupancy(0.5)
    for n in args.points:
        xyz = torch.from_numpy(bench_scene._probe_points(bf, n, 1)).cuda()
        rng = np.random.default_rng(2)
        d = rng.standard_normal((n, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        dirs = torch.from_numpy(d).cuda()
        f = fused.FusedField(model, torch.tensor([[0.5]], device="cuda"), max_points=n)
        for _ in range(3):
            f(xyz, dirs)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            f(xyz, dirs)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1000 / args.iters
        lib.sdn_debug_field_stamps(raw.ctypes.data)      # clears
        f(xyz, dirs)                                      # ONE launch's stamps
        lib.sdn_debug_field_stamps(raw.ctypes.data)
        st = raw.reshape(-1, 2, 32).astype(np.int64)
        wgs = min((n + 255) // 256, st.shape[0])
        st = st[:wgs]
        rec = {"points": n, "workgroups": wgs, "us_per_launch_diag_build": round(us, 2), "variant": args.variant, "cycles_per_workgroup": {}}
        t0 = st[:, :, 0].min()
        for w, name in ((0, "wave0"), (1, "wave7")):
            dur = np.diff(st[:, w, :16], axis=1)
            ph = {PHASES[k]: int(round(float(np.median(dur[:, k])))) for k in range(15)}
            ph["lifetime_median"] = int(np.median(st[:, w, 15] - st[:, w, 0]))
            rec["cycles_per_workgroup"][name] = ph
        rec["launch_span_cycles"] = int(st[:, :, 15].max() - t0)
        # co-residency: workgroups on the same CU (XCC id, HW_ID bits 8..15) whose lifetimes overlap -- how far apart do they enter
        # the hidden layers (stamp 3), as a fraction of the lifetime?  0 = lockstep, 0.5 = perfectly out of step
        key = (st[:, 0, 17] << 16) | ((st[:, 0, 16] >> 8) & 0xFF)
        offs = []
        for k in np.unique(key):
            idx = np.nonzero(key == k)[0]
            idx = idx[np.argsort(st[idx, 0, 0])]
            for a in range(len(idx)):
                for b in range(a + 1, len(idx)):
                    i, j = idx[a], idx[b]
                    if st[j, 0, 0] < st[i, 0, 15]:     # j started before i ended
                        life = 0.5 * ((st[i, 0, 15] - st[i, 0, 0]) + (st[j, 0, 15] - st[j, 0, 0]))
                        offs.append(abs(float(st[j, 0, 3] - st[i, 0, 3])) / max(life, 1.0))
        if offs:
            offs = np.array(offs)
            rec["coresident_pairs"] = int(len(offs))
            rec["hidden_layer_entry_offset_over_lifetime"] = {"median": round(float(np.median(offs)), 3), "p25": round(float(np.percentile(offs, 25)), 3),
                                                               "p75": round(float(np.percentile(offs, 75)), 3)}
            rec["cus_seen"] = int(len(np.unique(key)))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
