"""Where does a resumed native training run differ from an uninterrupted one?  

Three runs of `NativeTrainStep` from the same state on the same batch (the setup of tests/test_gpu_train_native.py):
  A, B  six uninterrupted steps each          -> |A - B| is the run-to-run floor (the table gradient's fp16 atomics are the one
                                                 order-dependent sum of the step)
  C     three steps, save, fresh objects, load, refresh(optimizer_state=True), three steps
and, before C's resumed steps, a BITWISE comparison of everything the step reads -- fp32 parameters, their fp16 copies, both Adam
moments, the device step counts, the scaler's scale and growth tracker, the gradient accumulator -- with the state the
uninterrupted run had after its third step.  Prints one JSON line per parameter and a verdict line.
"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "seald-nerf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

N_RAYS = 1024


def setup():
    from dnerf_amd.bench_scene import build_scene
    from dnerf_amd.network import NeRFNetwork
    sc = build_scene(H=32, W=32, device="cuda", seed=0)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    model.load_state_dict(sc.model.state_dict())
    opt = torch.optim.Adam(model.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda")
    target = torch.rand(1, N_RAYS, 3, generator=torch.Generator().manual_seed(4)).cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model.render(sc.rays_o[None], sc.rays_d[None], sc.time, staged=False, perturb=False, bg_color=1, force_all_rays=False)
    model.mean_count = int(model.step_counter[0, 0].item()) + 256
    model.local_step = 0
    model.step_counter.zero_()
    return sc, model, opt, scaler, target


def snapshot(step, model, opt, scaler):
    rows = model.encoder.embeddings.shape[0]
    s = {"adam_steps": step.adam_steps.clone(), "scale": scaler._scale.clone(), "tracker": scaler._growth_tracker.clone(),
         "g_table": step.view("g_table", torch.float16, (rows, 2)).clone(),
         "w_table": step.view("w_table", torch.float16, (rows, 2)).clone(),
         "w_deform": step.view("w_deform", torch.float16, (128 * 80 + 6 * 128 * 128 + 16 * 128,)).clone(),
         "w_sigma0": step.view("w_sigma0", torch.float16, (64, 32)).clone(), "w_sigma1": step.view("w_sigma1", torch.float16, (16, 64)).clone(),
         "w_color": step.view("w_color", torch.float16, (64 * 32 + 64 * 64 + 16 * 64,)).clone()}
    for n, p in model.named_parameters():
        s["p." + n] = p.detach().clone()
        s["m." + n] = opt.state[p]["exp_avg"].clone()
        s["v." + n] = opt.state[p]["exp_avg_sq"].clone()
    return s


def run(n_steps, stop_at=None):
    from dnerf_amd.train_native import NativeTrainStep
    sc, model, opt, scaler, target = setup()
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False)
    mid = None
    for k in range(n_steps):
        step(sc.rays_o, sc.rays_d, target, sc.time)
        if stop_at is not None and k + 1 == stop_at:
            torch.cuda.synchronize()
            step.sync_optimizer_state()
            mid = {"snap": snapshot(step, model, opt, scaler), "model": copy.deepcopy(model.state_dict()), "opt": copy.deepcopy(opt.state_dict()),
                   "scaler": scaler.state_dict(), "mean_count": model.mean_count, "local_step": model.local_step}
    torch.cuda.synchronize()
    return {n: p.detach().clone() for n, p in model.named_parameters()}, mid, (sc, target)


def resumed(mid, sc, target, n_steps):
    from dnerf_amd.network import NeRFNetwork
    from dnerf_amd.train_native import NativeTrainStep
    model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    model.load_state_dict(mid["model"])
    model.mean_count, model.local_step = mid["mean_count"], mid["local_step"]
    opt = torch.optim.Adam(model.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda")
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False)
    opt.load_state_dict(mid["opt"])
    scaler.load_state_dict(mid["scaler"])
    step.refresh(optimizer_state=True)
    torch.cuda.synchronize()
    got = snapshot(step, model, opt, scaler)
    bitwise = {k: bool(torch.equal(got[k], v)) for k, v in mid["snap"].items()}
    for _ in range(n_steps):
        step(sc.rays_o, sc.rays_d, target, sc.time)
    torch.cuda.synchronize()
    return {n: p.detach().clone() for n, p in model.named_parameters()}, bitwise


def stats(a, b, lr):
    d = (a - b).abs()
    return {"max_over_lr": float(d.max()) / lr, "mean_over_lr": float(d.mean()) / lr, "frac_gt_1e-5": float((d > 1e-5).float().mean())}


def main():
    A, mid, (sc, target) = run(6, stop_at=3)
    B, _, _ = run(6)
    C, bitwise = resumed(mid, sc, target, 3)
    worst = {"floor": 0.0, "resumed": 0.0, "floor_mean": 0.0, "resumed_mean": 0.0}
    for n in A:
        lr = 1e-2 if n == "encoder.embeddings" else 1e-3
        f, r = stats(A[n], B[n], lr), stats(A[n], C[n], lr)
        worst["floor"], worst["resumed"] = max(worst["floor"], f["max_over_lr"]), max(worst["resumed"], r["max_over_lr"])
        worst["floor_mean"], worst["resumed_mean"] = max(worst["floor_mean"], f["mean_over_lr"]), max(worst["resumed_mean"], r["mean_over_lr"])
        print(json.dumps({"param": n, "floor_A_vs_B": f, "resumed_C_vs_A": r}))
    not_restored = [k for k, ok in bitwise.items() if not ok]
    print(json.dumps({"state_restored_bitwise": not not_restored, "not_restored": not_restored, "worst_in_units_of_lr": worst}))


if __name__ == "__main__":
    main()
