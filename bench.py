#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s + sampled-points/s, 800x800 D-NeRF "jumpingjacks-like" frame
(BASELINE.json config[1]: `-O`, i.e. fp16 field network, HIP gridencoder + raymarching + shencoder on one
MI355X; with --gpus N the frame's rays are sharded N-way and the rendered tiles all-gathered over RCCL).

    python bench.py --gpus N --steps K --warmup W        (N > 1: under torch.distributed.run, or alone -- it then starts the N ranks
                                                          itself as a child `python -m torch.distributed.run ... bench.py <same args>`)

One "step" = one full pass of the hot path over one batch = one rendered frame (all rays of the frame,
or this rank's shard of them).  Inputs (rays, weights, occupancy bitfield) are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "seald-nerf_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "rendered rays/sec + sampled-points/sec, 800×800 D-NeRF jumpingjacks"
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32: the fp32 vector rate (MI355X_MICROARCH.md)
GRID_BYTES_PER_POINT = {"f16": 588.0, "f32": 1164.0}  # gathers + 12 B in + outputs
FIELD_FLOP_PER_POINT = 235520.0                          # 117 760 MAC
PMC_SUMMARY = "r04_field_pmc_summary.json"               # rocprofv3 --pmc summary the static `traffic` figure comes from


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=384)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=800, help="image side (800 = BASELINE config)")
    ap.add_argument("--fp32", action="store_true", help="fp32 field network instead of the -O (fp16) configuration")
    ap.add_argument("--field", default="auto", choices=["auto", "ops", "fused"], help="field network implementation")
    ap.add_argument("--loop", default="auto", choices=["auto", "host", "device"], help="loop driver: per-iteration host sync, or device-driven")
    ap.add_argument("--time-every", type=int, default=8, help="record the per-launch HIP events of the tracked kernels on every K-th timed step "
                    "(each event record costs ~5 us of queue time between two dependent launches; K = 1 instruments every step)")
    ap.add_argument("--pipeline", type=int, default=2, metavar="DIV",
                    help="device loop: render the timed steps as a stream of frames through --contexts loop contexts; the next frame "
                         "starts when a context is free and the newest frame in flight is down to rays / DIV alive (1 = at once; 2 = behind the newest "
                         "frame's first march, measured 1-2 %% better than at once; "
                         "0 = strictly one frame at a time)")
    ap.add_argument("--contexts", type=int, default=4, help="--pipeline: loop contexts (frames in flight); the HIP runtime is given one "
                                                            "hardware queue per context + the default stream (GPU_MAX_HW_QUEUES, if unset)")
    ap.add_argument("--cameras", type=int, default=20, help="frames of the test sequence (camera orbit + time ramp 0..1); the timed steps cycle through it")
    ap.add_argument("--static-frame", action="store_true", help="render ONE camera at t = 0.5 over and over (the round-1 workload) instead of the sequence")
    ap.add_argument("--group-frames", type=int, default=0, help="frames rendered together by one loop (frame group); 0 = 4 x --gpus, at most 16; "
                                                                 "moved to the nearest divisor of --steps")
    ap.add_argument("--emulate-rank-of", type=int, default=1, metavar="N",
                    help="one GPU only: render rank 0's shard of an N-way ray split of every frame (what one rank of --gpus N does, without "
                         "the gathers) -- to measure shard-sized loops / frame groups on one GPU; rays_per_s then counts shard rays")
    ap.add_argument("--mode", default="render", choices=["render", "train", "seald", "seald-train", "density"],
                    help="render: the headline 800x800 inference frame; train: one dnerf training step on 4096 rays (BASELINE config 3)")
    ap.add_argument("--scene", default="jumpingjacks", choices=["jumpingjacks", "lego"],
                    help="synthetic occupancy: the jumpingjacks-like figure (headline, BASELINE configs 1-4) or the lego-like box (config 5)")
    ap.add_argument("--train-mlp", default="ffmlp", choices=["ffmlp", "linear"],
                    help="--mode train: deformation / colour MLPs through the fused-MLP operator (dnerf_amd/network_ff.py) or as the "
                         "reference's nn.Linear stack under autocast")
    ap.add_argument("--train-graph", type=int, default=1, help="--mode train with --train-mlp ffmlp: replay the step as one captured HIP "
                                                                "graph (dnerf_amd/train_graph.py); 0 = launch it from Python")
    ap.add_argument("--train-native", type=int, default=1, help="--mode train: the step as ONE native call (sdn_train_step_f16, "
                                                                 "dnerf_amd/train_native.py); 0 = the autograd step (graphed or eager)")
    ap.add_argument("--train-prefetch", type=int, default=1, help="--mode train, native step: march batch k+1 on a second stream beside step k")
    ap.add_argument("--train-overlap", type=int, default=1, help="--mode train, native step: the optimizer's pass over the embedding table on a "
                                                                 "second stream, beside the next step's deformation-MLP forward")
    ap.add_argument("--gather-dtype", default="f32", choices=["f32", "f16", "u8"],
                    help="--gpus N: what the per-frame all-gather moves (fp32 as rendered; fp16; or the 8-bit pixels the reference writes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-side", type=int, default=224, help="side of the CPU-baseline sample of the 800x800 camera (224: ~12 s of CPU work; "
                                                                       "800 = the whole frame SURVEY 8(d) names, ~160 s)")
    ap.add_argument("--min-timed-s", type=float, default=0.25,
                    help="the K-step stream is repeated inside the timed region until it lasts at least this long (ms_per_step = elapsed / "
                         "frames rendered; `repeats` in the output; 0 = exactly one pass)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the marcher / grid_encode side measurements (roofline_secondary, grid_gather_rate)")
    return ap.parse_args()


def train_mode(args):
    """BASELINE config 3: one dnerf training step (march_rays_train, field network under autocast, composite_rays_train forward +
    backward, grid_encode backward atomics, Adam) on 4096 rays of the 800x800 camera, fp16 autocast + GradScaler as `-O` sets it
    (main_dnerf.py:70-73,129; nerf/utils.py:869-886).  The MLPs run as torch (hipBLASLt) GEMMs: the fused kernel is inference-only."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    import sdn_backend
    from dnerf_amd.bench_scene import build_scene
    sc = build_scene(H=args.size, W=args.size, device=dev, seed=0)
    model = sc.model
    if args.train_mlp == "ffmlp" and not args.fp32:
        from dnerf_amd.network_ff import NeRFNetworkFF
        model = NeRFNetworkFF(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).to(dev)
        model.load_state_dict(sc.model.state_dict())
    model.train()
    n_rays = 4096
    g = torch.Generator(device="cpu").manual_seed(0)
    idx = torch.randint(0, sc.rays_o.shape[0], (n_rays,), generator=g).to(dev)
    rays_o, rays_d = sc.rays_o[idx][None].contiguous(), sc.rays_d[idx][None].contiguous()
    target = torch.rand(1, n_rays, 3, generator=torch.Generator(device="cpu").manual_seed(2)).to(dev)
    native = bool(args.train_native) and not args.fp32
    graphed = bool(args.train_graph) and type(model).__name__ == "NeRFNetworkFF" and not native
    # the graph needs the optimizer that takes GradScaler's found_inf on the device (fused + capturable Adam)
    groups = model.get_params(1e-2, 1e-3)
    if graphed:
        from dnerf_amd.train_graph import merged_param_groups
        groups = merged_param_groups(groups)   # 4 non-empty groups -> 2 (table at lr, all MLP weights at lr_net): same update, fewer launches
    opt = torch.optim.Adam(groups, betas=(0.9, 0.99), eps=1e-15, **({"fused": True, "capturable": True} if graphed else {}))
    scaler = torch.amp.GradScaler("cuda", enabled=not args.fp32)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=not args.fp32):
            out = model.render(rays_o, rays_d, sc.time, staged=False, perturb=True, bg_color=1, force_all_rays=False, max_steps=1024)
            loss = ((out["image"] - target) ** 2).mean()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        return loss

    torch.manual_seed(1)
    for _ in range(2):  # first steps: unknown point budget (M = N * max_steps buffers, host read-back), as in the reference
        step()
    model.mean_count = int(model.step_counter[:2, 0].sum().item() / 2)  # what update_extra_state does (dnerf/renderer.py:550-552)
    if native:
        from dnerf_amd.train_native import NativeTrainStep
        nstep = NativeTrainStep(model, opt, scaler, n_rays, dev, perturb=True, bg_color=1, overlap_table_update=bool(args.train_overlap))
        nstep.load(rays_o, rays_d, target, sc.time)
        if args.train_prefetch:
            # the next batch is marched on a second stream beside the running step (a loader knows it one step early)
            def step():  # noqa: F811
                loss = nstep(rays_o, rays_d, target, sc.time)
                nstep.prefetch(rays_o, rays_d, sc.time)
                return loss
        else:
            step = nstep  # noqa: F811  (inputs stay in the step's buffers, as with the graph)
    if graphed:
        from dnerf_amd.train_graph import GraphedTrainStep
        gstep = GraphedTrainStep(model, opt, scaler, n_rays, dev)
        gstep.load(rays_o, rays_d, target, sc.time)
        gstep.capture()
        step = gstep  # noqa: F811  (inputs stay in the graph's buffers: a data loader would copy each batch in, 3 x 48 KiB)
    for _ in range(args.warmup):
        step()
    n_points = int(model.step_counter[(model.local_step - 1) % 16, 0].item())
    timers = sdn_backend.KernelTimers()
    import gc
    gc.collect()
    gc.disable()   # no cyclic-collector pass inside the timed region (see main())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sdn_backend.timers = timers
        step()
    sdn_backend.timers = None
    host_dt = time.perf_counter() - t0          # the host's share: enqueueing only (the calls are asynchronous)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    summ = timers.summary()
    print(json.dumps({"metric": "dnerf training step, 4096 rays (march_rays_train + field + composite_rays_train fwd/bwd + grid backward + Adam)",
                      "value": args.steps / dt, "unit": "steps/s", "points_per_s": n_points * args.steps / dt, "rays_per_s": n_rays * args.steps / dt,
                      "ms_per_step": dt / args.steps * 1e3, "host_enqueue_ms_per_step": host_dt / args.steps * 1e3,
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "dtype": "f32" if args.fp32 else "f16",
                      "data": "synthetic", "config": {"workload": "BASELINE config 3", "rays": n_rays, "sampled_points_per_step": n_points,
                                                      "mean_count": model.mean_count,
                                                      "mlps": ("deform + colour MLPs on the fused-MLP kernels, density MLP in a per-sample dot2 kernel" if native else
                                                               "deform + colour MLPs on the fused-MLP operator (ffmlp), density MLP hipBLASLt" if type(model).__name__ == "NeRFNetworkFF" else "nn.Linear stack (hipBLASLt GEMMs)"),
                                                      "launch": ("one native call per step (sdn_train_step_f16: ~30 launches, Adam + fp16 copies + gradient clear in one pass)" + ("; the next batch's rays are marched on a second stream beside the running step" if args.train_prefetch else "") if native
                                                                 else "one captured HIP graph per step (fused capturable Adam)" if graphed else "eager (Python launches)")},
                      "kernel_times": summ}))


def seald_mode(args):
    """BASELINE config 4: the SealD-NeRF teacher's edit render -- the jumpingjacks-like frame with the figure's head copied
    0.35 to the side and hue-shifted by a bounding-box seal mapper -- through the reference-shaped loop (`SealDNeRFTeacher`, op by
    op) and through the native loop with the fused field (`render_frame(..., mapper=)`).  One JSON line."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    from dnerf_amd.bench_scene import build_scene
    from dnerf_amd.renderer import render_frame, FrameWorkspace
    from dnerf_amd import fused, seal_mapper as SM
    sc = build_scene(H=args.size, W=args.size, device=dev, seed=0)
    half, centre = 0.12, (0.0, 0.47, 0.0)
    raw = [[centre[0] + sx * half, centre[1] + sy * half, centre[2] + sz * half] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)]
    T = np.eye(4); T[0, 3] = 0.35
    mapper = SM.get_seal_mapper({"type": "bbox", "raw": raw, "transform": T.tolist(), "scale": [1.0, 1.0, 1.0], "boundType": "to",
                                 "hsv": [0.3, 0.0, 0.0]})
    SM.fill_bitfield(sc.model.density_bitfield, mapper.map_data["force_fill_bound"].cpu().numpy(), sc.model.grid_size, sc.model.bound)
    field = fused.FusedField(sc.model, sc.time, fp16=True, max_points=sc.rays_o.shape[0] + 128)
    from dnerf_amd.renderer import DeviceLoop, PipelinedDeviceLoop
    N = sc.rays_o.shape[0]
    one = DeviceLoop(sc.model, field, N, dev, T_thresh=1e-4, mapper=mapper)
    first = one.render(sc.rays_o, sc.rays_d, sc.time)
    # frames per loop (as in the default mode: a group of 4 copies of the frame through one loop; --group-frames 1 = one frame per loop)
    F = args.group_frames if args.group_frames > 0 else 4
    F = max(d for d in range(1, min(F, 16) + 1) if args.steps % d == 0)
    if F > 1:
        field = fused.FusedField(sc.model, sc.time, fp16=True, max_points=N * F + 128)
        grp_o, grp_d = torch.cat([sc.rays_o] * F).contiguous(), torch.cat([sc.rays_d] * F).contiguous()
        tval = float(sc.time.reshape(-1)[0])
        grp_t = [tval] * F
    else:
        grp_o, grp_d, grp_t = sc.rays_o, sc.rays_d, sc.time
    pl = PipelinedDeviceLoop(sc.model, field, N * F, dev, contexts=args.contexts, overlap_div=max(1, args.pipeline), T_thresh=1e-4, mapper=mapper, frames=F)
    n_loops = args.steps // F
    import gc

    def timed(fn):
        gc.collect(); gc.disable()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        return dt
    for _ in range(args.warmup):
        one.render(sc.rays_o, sc.rays_d, sc.time, want_stats=False)
    times = (lambda k: [grp_t] * k) if F > 1 else (lambda k: sc.time)
    pl.render_frames([grp_o] * args.contexts, [grp_d] * args.contexts, times(args.contexts))
    dt_one = timed(lambda: [one.render(sc.rays_o, sc.rays_d, sc.time, want_stats=False) for _ in range(args.steps)])
    dt = timed(lambda: pl.render_frames([grp_o] * n_loops, [grp_d] * n_loops, times(n_loops)))
    print(json.dumps({"metric": "SealD-NeRF teacher edit render (bbox seal mapper), 800x800 jumpingjacks-like frame", "value": first["n_samples"] * args.steps / dt,
                      "unit": "sampled-points/s", "rays_per_s": N * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                      "ms_per_step_one_frame_at_a_time": dt_one / args.steps * 1e3, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "dtype": "f16", "data": "synthetic",
                      "config": {"workload": "BASELINE config 4", "rays": int(N), "sampled_points_per_frame": first["n_samples"],
                                 "loop_iterations": len(first["trace"]), "T_thresh": 1e-4,
                                 "mapper": "SealBBoxMapper: sdn_seal_bbox_map / sdn_seal_modify_hsv inside the native frame driver, between the marcher and the fused field",
                                 "loop": "device", "frames_per_loop": F, "frames_in_flight": f"{args.contexts} loops of {F} frame(s)"}}))


def seald_train_mode(args):
    """SealD-NeRF edit-training step (StudentTrainer.train_gui, SealDNeRF/utils.py:667-777) on 4096 rays of the 800x800
    camera: the teacher renders the edited scene (bbox seal mapper, T_thresh 1e-4), the student -- deformation network frozen --
    trains on it.  Native: teacher through the device loop with the fused field + seal kernels, student step as one native call
    (dnerf_amd/seald_train.py -> train_native.NativeTrainStep).  Reference-shaped: `SealDNeRFTeacher.render` op by op + an eager autocast step on the nn.Linear
    network with torch's Adam.  One JSON line."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    import gc
    from dnerf_amd.bench_scene import build_scene
    from dnerf_amd.network_ff import NeRFNetworkFF
    from dnerf_amd.seald import SealDNeRFTeacher
    from dnerf_amd.seald_train import EditTrainStep, freeze_deformation
    from dnerf_amd import seal_mapper as SM
    sc = build_scene(H=args.size, W=args.size, device=dev, seed=0)
    half, centre = 0.12, (0.0, 0.47, 0.0)
    raw = [[centre[0] + sx * half, centre[1] + sy * half, centre[2] + sz * half] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)]
    T = np.eye(4); T[0, 3] = 0.35
    mapper = SM.get_seal_mapper({"type": "bbox", "raw": raw, "transform": T.tolist(), "scale": [1.0, 1.0, 1.0], "boundType": "to",
                                 "hsv": [0.3, 0.0, 0.0]})
    SM.fill_bitfield(sc.model.density_bitfield, mapper.map_data["force_fill_bound"].cpu().numpy(), sc.model.grid_size, sc.model.bound)
    n_rays = 4096
    idx = torch.randint(0, sc.rays_o.shape[0], (n_rays,), generator=torch.Generator(device="cpu").manual_seed(0)).to(dev)
    rays_o, rays_d = sc.rays_o[idx].contiguous(), sc.rays_d[idx].contiguous()
    kw = dict(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1)

    def budget(m):   # the point budget the reference's first epoch provides
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            m.render(rays_o[None], rays_d[None], sc.time, staged=False, perturb=False, bg_color=1, force_all_rays=False, max_steps=1024)
        m.mean_count = int(m.step_counter[0, 0].item()) + 1024
        m.local_step = 0

    def timed(fn, steps):
        gc.collect(); gc.disable()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        return dt / steps * 1e3

    # native
    student = NeRFNetworkFF(**kw).to(dev).train()
    student.load_state_dict(sc.model.state_dict())
    params = freeze_deformation(student)
    budget(student)
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
    edit = EditTrainStep(sc.model, student, mapper, opt, torch.amp.GradScaler("cuda"), n_rays, dev, sc.time)
    for _ in range(args.warmup):
        edit(rays_o, rays_d, sc.time)
    ms_native = timed(lambda: edit(rays_o, rays_d, sc.time), args.steps)
    ms_teacher = timed(lambda: edit.proxy_truth(rays_o, rays_d, sc.time), args.steps)
    edit.run([(rays_o, rays_d, sc.time)] * 4)
    ms_pipelined = timed(lambda: edit.run([(rays_o, rays_d, sc.time)] * args.steps), 1) / args.steps
    # reference-shaped
    from dnerf_amd.network import NeRFNetwork
    teacher = SealDNeRFTeacher(**kw).to(dev).eval()
    teacher.load_state_dict(sc.model.state_dict(), strict=False)
    teacher.init_mapper(mapper)
    ref_student = NeRFNetwork(**kw).to(dev).train()
    ref_student.load_state_dict(sc.model.state_dict())
    ref_params = freeze_deformation(ref_student)
    budget(ref_student)
    ref_opt = torch.optim.Adam(ref_params, lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    ref_scaler = torch.amp.GradScaler("cuda")

    def ref_step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            target = teacher.render(rays_o[None], rays_d[None], sc.time, staged=True, perturb=False, bg_color=1, force_all_rays=True)["image"]
        ref_opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            out = ref_student.render(rays_o[None], rays_d[None], sc.time, staged=False, perturb=True, bg_color=1, force_all_rays=False, max_steps=1024)
            loss = ((out["image"] - target) ** 2).mean()
        ref_scaler.scale(loss).backward()
        ref_scaler.step(ref_opt)
        ref_scaler.update()
    for _ in range(args.warmup):
        ref_step()
    ms_ref = timed(ref_step, max(5, args.steps // 2))
    print(json.dumps({"metric": "SealD-NeRF edit-training step (teacher proxy render with bbox seal mapper + student step), 4096 rays",
                      "value": 1e3 / ms_pipelined, "unit": "steps/s", "ms_per_step": ms_pipelined, "ms_per_step_one_after_the_other": ms_native,
                      "teacher_render_ms": ms_teacher,
                      "reference_shaped_ms_per_step": ms_ref, "speedup": ms_ref / ms_pipelined, "rays_per_s": n_rays * 1e3 / ms_pipelined,
                      "higher_is_better": True, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "dtype": "f16", "data": "synthetic",
                      "config": {"workload": "SURVEY 3.4 / BASELINE config 4 in training: StudentTrainer.train_gui step", "rays": n_rays,
                                 "student": "deform_net frozen, one native call per step (sdn_train_step_f16, deform_frozen)",
                                 "teacher": "one-pass ray-batch render (march, one fused-field launch, whole-ray compositing; the loop's image bit for bit) + seal kernels, T_thresh 1e-4",
                                 "reference_shaped": "SealDNeRFTeacher.render op by op + eager nn.Linear student step, torch Adam"}}))


def density_mode(args):
    """The density-grid maintenance pass between training epochs (update_extra_state, dnerf/renderer.py:453-555):
    64 time slices x 128^3 cells through the density network + EMA + mean + bitfield.  Times the full update (iter_density < 16)
    and the partial update (16..99) on the device path (csrc/density.hip + the CELLS variant of the fused field kernel, including
    the per-call weight re-packing) next to the reference-shaped op-by-op pass on the same HIP operators.  One JSON line."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    import gc
    from dnerf_amd.bench_scene import build_scene, build_model
    sc = build_scene(H=8, W=8, device=dev, seed=0)

    def fresh(native):
        m = build_model(0, dev)
        m.load_state_dict(sc.model.state_dict())
        m.reset_extra_state()
        if native:
            m.use_native_density_update(fp32=args.fp32)
        return m

    def timed(m, first_iter, steps):
        out = []
        for k in range(steps):
            m.iter_density = first_iter
            gc.collect(); gc.disable()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.autocast("cuda", dtype=torch.float16, enabled=not args.fp32):
                m.update_extra_state()
            torch.cuda.synchronize()
            out.append(time.perf_counter() - t0)
            gc.enable()
        return sorted(out)[len(out) // 2]

    nat, ref = fresh(True), fresh(False)
    cells = nat.time_size * nat.cascade * nat.grid_size ** 3
    res = {}
    for name, m, steps in (("native", nat, max(3, args.steps // 4)), ("op_by_op", ref, 3)):
        timed(m, 0, 1)                                           # warm-up (allocator, first launches)
        res[name + "_full_ms"] = timed(m, 0, steps) * 1e3
        res[name + "_partial_ms"] = timed(m, 16, steps) * 1e3
    t_full = res["native_full_ms"] * 1e-3
    # dominant kernel: the fused field kernel, sigma branch only: deform MLP + sigma MLP = 2 * (76*128 + 6*128*128 + 128*3 + 32*64 + 64*16) MAC
    flop = 2.0 * (76 * 128 + 6 * 128 * 128 + 128 * 3 + 32 * 64 + 64 * 16) * cells
    peak = MFMA_F32_PEAK_TFLOPS if args.fp32 else MFMA_F16_PEAK_TFLOPS
    print(json.dumps({"metric": "density-grid full update (update_extra_state), 64 x 128^3 cells", "value": cells / t_full, "unit": "cells/s",
                      "ms_per_step": res["native_full_ms"], "higher_is_better": True, "n_gpus": 1, "dtype": "f32" if args.fp32 else "f16", "data": "synthetic",
                      **{k: round(v, 3) for k, v in res.items()},
                      "speedup_full": res["op_by_op_full_ms"] / res["native_full_ms"], "speedup_partial": res["op_by_op_partial_ms"] / res["native_partial_ms"],
                      "whole_pass_mfma": {"achieved": flop / t_full / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flop / t_full / 1e12 / peak},
                      "config": {"workload": "SURVEY 8(f)2: density-grid maintenance", "time_slices": nat.time_size, "grid": nat.grid_size,
                                 "cells": cells, "partial_cells_per_slice": 2 * (nat.grid_size ** 3 // 4),
                                 "native": ("sdn_density_query_cells_f32" if args.fp32 else "sdn_density_query_cells_f16") + " per slice + sdn_density_grid_ema + sdn_density_grid_pack, incl. weight re-pack",
                                 "op_by_op": "renderer mirror on the HIP operators (the reference's structure)"}}))


def self_launch(args):
    """`python3 bench.py --gpus N` (N > 1) started as ONE process, the shape of the driver's N = 1 command: start the N ranks ourselves as
    a CHILD process -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    <same arguments>` -- before this process has made any GPU / HIP call (importing torch makes none), relay the child's output (rank 0
    prints the JSON line) and exit with its code.  Nothing is exec'ed: the parent only waits."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.mode == "render":
        self_launch(args)
    # One hardware queue per stream in use (the loop contexts + torch's default stream), set before the HIP runtime starts: with the
    # runtime's default of 4 a fifth stream shares a queue with another frame's chain of dependent launches (4 contexts: 0.42 ms per
    # shard-sized frame), with more queues than streams three contexts got slower (0.85 ms); measured best: contexts + 1.
    # (only the modes that keep `contexts` frames in flight: queues beyond the streams in use cost as much as too few -- the
    # two-stream edit-training epoch ran at 3.8 instead of 1.3 ms per step with 5 queues)
    if args.mode in ("render", "seald"):
        # (ranks of a multi-GPU job also run the per-loop all-gathers on a stream of their own: one queue more, so that a collective
        # never waits behind a loop's chain of dependent launches)
        multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, args.contexts + (2 if multi else 1))))
    if args.mode == "train":
        return train_mode(args)
    if args.mode == "density":
        return density_mode(args)
    if args.mode == "seald-train":
        return seald_train_mode(args)
    if args.mode == "seald":
        return seald_mode(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world} (start `python bench.py --gpus N` alone, or under "
                         f"`python -m torch.distributed.run --nproc-per-node N`)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
    # rehearsal switch for a one-GPU box: every rank renders on device 0 and the frame is gathered over gloo (host staging);
    # exercises the sharding / gather / timing protocol only -- never use its numbers
    rehearse = os.environ.get("SDN_REHEARSE_ON_ONE_GPU", "0") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from dnerf_amd.bench_scene import build_scene, camera_path
    from dnerf_amd.renderer import render_frame, FrameWorkspace
    from dnerf_amd import fused

    fp16 = not args.fp32
    sc = build_scene(H=args.size, W=args.size, device=dev, seed=0, kind=args.scene)
    n_total = sc.rays_o.shape[0]
    K = args.steps
    # The workload: a D-NeRF test sequence -- one camera pose AND one time stamp per frame (dnerf/provider.py test split,
    # dnerf/utils.py:151-161; slice select dnerf/renderer.py:285): the camera orbits once while t runs 0 -> 1 over `--cameras` frames,
    # repeated when --steps is larger.  --static-frame renders one camera at t = 0.5 over and over instead (the round-1 workload).
    n_cams = 1 if args.static_frame else max(1, min(args.cameras, K))
    if args.static_frame:
        cam_o, cam_d, cam_t = [sc.rays_o], [sc.rays_d], [0.5]
    else:
        cam_o, cam_d, cam_t = camera_path(sc, n_cams, dev)
    gather = None
    if world > 1 or args.emulate_rank_of > 1:
        # ray-parallel: this rank renders its interleaved 16x16 tiles of EVERY frame
        from dnerf_amd.dist import shard_rays, FrameGather
        parts = world if world > 1 else args.emulate_rank_of
        idx, per = shard_rays(n_total, args.size, rank if world > 1 else 0, parts)
        idx_t = torch.from_numpy(idx).to(dev)
        cam_o = [r[idx_t].contiguous() for r in cam_o]
        cam_d = [r[idx_t].contiguous() for r in cam_d]
        if world > 1:
            gather = FrameGather(n_total, args.size, world, dev, transport=args.gather_dtype)
    n_local = cam_o[0].shape[0]
    # Frame groups: F consecutive frames' shards rendered by ONE loop (per-ray time constants), so that the ~30 dependent launches
    # of a loop are paid once per F frames; default F = gpus (a rank's batch keeps the size of one full frame), 1 on one GPU.
    # (one GPU: 4 consecutive frames per loop -- launches of ~300 K points instead of ~100 K fill the chip's 512 workgroup slots 2.3
    # times instead of 0.9 times: field kernel 0.25 -> 0.35 of the MFMA peak, 2 % more frames per second; sweep in profiles/r03_*)
    # (N GPUs: 4 N frames' shards per loop, i.e. the same 2.56 M rays per loop as 4 full frames on one GPU -- one rank of 8 at --steps 20:
    # 0.0876 ms per frame with 5 frames per loop, 0.0713 with 10; profiles/r03_rank_frames_per_loop.txt)
    want_f = args.group_frames if args.group_frames > 0 else (min(16, 4 * world) if world > 1 else (4 if args.emulate_rank_of <= 1 else 1))
    if args.group_frames == 0 and (args.field == "ops" or args.loop == "host"):
        want_f = 1            # frame groups need the device loop with the fused field
    F = min((d for d in range(1, 17) if K % d == 0), key=lambda d: (abs(d - min(want_f, 16)), -d))    # the divisor of --steps nearest to it
    n_groups = K // F
    frame_cam = [f % n_cams for f in range(K)]
    if F == 1:
        grp_o, grp_d = [cam_o[c] for c in frame_cam], [cam_d[c] for c in frame_cam]
        grp_t = [cam_t[c] for c in frame_cam]
    else:
        grp_o = [torch.cat([cam_o[frame_cam[g * F + i]] for i in range(F)]).contiguous() for g in range(n_groups)]
        grp_d = [torch.cat([cam_d[frame_cam[g * F + i]] for i in range(F)]).contiguous() for g in range(n_groups)]
        grp_t = [[cam_t[frame_cam[g * F + i]] for i in range(F)] for g in range(n_groups)]
    n_loop = n_local * F
    ws = FrameWorkspace(n_local, dev)
    field_kind = args.field
    if field_kind == "auto":
        field_kind = "fused" if fused.available() else "ops"
    if field_kind == "fused" and not fp16:       # --fp32: the reference without -O through the fp32 fused kernel (csrc/field_f32.hip)
        from dnerf_amd.fused_f32 import FusedFieldF32
        field = FusedFieldF32(sc.model, sc.time)
        args.no_secondary = True                 # (the secondary figures describe the -O path)
    else:
        field = fused.FusedField(sc.model, sc.time, fp16=fp16) if field_kind == "fused" else None
    import sdn_backend
    timers = sdn_backend.KernelTimers()
    loop_kind = args.loop
    if loop_kind == "auto":
        loop_kind = "device" if field is not None else "host"
    dloop = None
    if loop_kind == "device":
        from dnerf_amd.renderer import DeviceLoop
        dloop = DeviceLoop(sc.model, field, n_loop, dev, frames=F, keep_cull_grids=True)
    elif F > 1:
        raise SystemExit("frame groups need the device loop (fused field)")

    ploop = None
    if args.pipeline == 2 and n_groups <= 2 * args.contexts and world == 1:
        args.pipeline = 1    # a stream of a few loops (frame groups of a short sequence): holding the later loops back costs more than the
                             # lockstep it avoids (4 loops of 5 frames: 0.127 against 0.115 ms per frame).  Not with real ranks: loops that
                             # end at different moments let all gathers but the last run under the rendering of the others.
    if args.pipeline > 0 and dloop is not None:
        from dnerf_amd.renderer import PipelinedDeviceLoop
        # (keep_cull_grids: the occupancy grid does not change while a sequence is rendered: each slice's cull grid is derived once)
        ploop = PipelinedDeviceLoop(sc.model, field, n_loop, dev, overlap_div=args.pipeline, contexts=args.contexts, frames=F, keep_cull_grids=True)

    def make_timing(frames):
        """Event pairs (created outside the timed region) for the field launches of `frames` instrumented frames."""
        import ctypes
        cur = torch.cuda.current_stream()
        out = []
        for _ in range(frames):
            recs = []
            for _ in range(DeviceLoop.MAX_TIMED):
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(cur); e_.record(cur)
                recs.append((s_, e_, 0))
            arr = (ctypes.c_void_p * (2 * DeviceLoop.MAX_TIMED))(*[h for r in recs for h in (r[0].cuda_event, r[1].cuda_event)])
            out.append((arr, recs))
        torch.cuda.synchronize()
        return out

    class GatherStats:
        """Event pairs around every per-loop gather (recorded on the stream the gather runs on: the helper thread's), so that the first
        real multi-GPU run says where the time went: per rank, the device-side span of the renders, the summed duration of the gathers,
        and how much of it was NOT hidden under rendering (gather time after the last render finished)."""

        def __init__(self):
            self.pairs = []

        def reset(self):
            self.pairs = []

        def wrap(self, fn):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            self.pairs.append((a, b))

        def summary(self, done_events, start_ev):
            if not self.pairs or not done_events:
                return {}
            gather_ms = sum(a.elapsed_time(b) for a, b in self.pairs)
            render_ms = max(start_ev.elapsed_time(e) for e in done_events)          # start of the stream -> the last loop's last kernel
            gathers_end = start_ev.elapsed_time(self.pairs[-1][1])
            exposed = max(0.0, gathers_end - render_ms)
            return {"loops": len(done_events), "render_ms": render_ms, "gathers": len(self.pairs), "gather_ms_total": gather_ms,
                    "gather_ms_mean": gather_ms / len(self.pairs), "gather_ms_exposed_after_last_render": exposed,
                    "gather_hidden_frac": 1.0 - min(1.0, exposed / max(gather_ms, 1e-9))}

    gather_stats, rank_stats = GatherStats(), {}

    def gather_outputs(img, dep):
        """One RCCL all-gather per LOOP (+ local un-permute): a frame, or the frames of a group together."""
        gather_stats.wrap((lambda: gather(img, dep)) if F == 1 else (lambda: gather.gather_group(img, dep, F)))

    def stream_of_frames(k, every=0, reps=1):
        """The first k loops (frames, or frame groups) of the sequence, `reps` times over, through the pipelined driver as ONE stream;
        every > 0: the field launches of every `every`-th loop are timed in place."""
        loops_o, loops_d, loops_t = grp_o[:k] * reps, grp_d[:k] * reps, grp_t[:k] * reps
        k = k * reps
        # instrumented loops: the last is rendered with nothing else in flight -- the pipeline is draining there anyway -- and its
        # launch durations are the kernel's own (the roofline figure); every `every`-th loop before it is instrumented while it
        # overlaps like all the others (a launch while it shares the device)
        excl_set = (exclusive_frames(k, F) if world == 1 else {k - 1}) if every else set()
        marked = sorted(set(range(0, k, every)) | excl_set) if every else []
        timing = make_timing(len(marked))
        per_frame, it = [None] * k, iter(timing)
        exclusive = [False] * k
        for f in marked:
            per_frame[f] = next(it)
            exclusive[f] = f in excl_set
        outputs = None
        if world > 1:  # every loop keeps its own shard output until it has been gathered (a ring of buffers: 4 loops are in flight)
            ring = [(torch.empty(n_loop, 3, dtype=torch.float32, device=dev), torch.empty(n_loop, dtype=torch.float32, device=dev))
                    for _ in range(min(k, 4 * args.contexts))]
            outputs = [ring[i % len(ring)] for i in range(k)]
        barrier()
        start_ev = torch.cuda.Event(enable_timing=True)
        start_ev.record()
        t0 = time.perf_counter()
        # multi-GPU: a helper thread issues the per-frame all-gathers of a loop the moment that loop's last kernel is enqueued, on its
        # own stream, while the later loops still render (the driver call itself only returns when every loop has finished)
        outs, iters = ploop.render_frames(loops_o, loops_d, loops_t, outputs=outputs,
                                          timing=[p[0] if p else None for p in per_frame] if every else None,
                                          exclusive=exclusive if every else None,
                                          on_done=(lambda f, img, dep: gather_outputs(img, dep)) if world > 1 else None)
        t1 = time.perf_counter()
        if world > 1:      # device-side spans of this rank, before the closing barrier: renders (first launch -> last loop's done event)
            torch.cuda.synchronize()
            rank_stats.update(gather_stats.summary(ploop.last_done_events, start_ev))
        barrier()
        elapsed = time.perf_counter() - t0
        if os.environ.get("SDN_DRIVER_STATS"):
            print(f"[bench] render_frames {1e3 * (t1 - t0):.2f} ms, final sync {1e3 * (time.perf_counter() - t1):.2f} ms", file=sys.stderr)
        for f, p in enumerate(per_frame):
            if p:
                key = "field_forward_f16" if exclusive[f] else "field_forward_f16_overlapped"
                timers.records.setdefault(key, []).extend(p[1][: min(DeviceLoop.MAX_TIMED, iters[f])])
        return elapsed, outs, marked, excl_set

    def time_tensor(t):
        return torch.tensor([[t]], dtype=torch.float32, device=dev)

    def step(g=0, count=False, timed=False):
        """Loop g of the sequence, one at a time."""
        sdn_backend.timers = timers if timed else None  # HIP events around the tracked launches, timed steps only
        if dloop is not None:
            out = dloop.render(grp_o[g], grp_d[g], grp_t[g], want_stats=count)
        else:
            out = render_frame(sc.model, grp_o[g], grp_d[g], time_tensor(grp_t[g]), fp16=fp16, workspace=ws, field=field, count_samples=count)
        sdn_backend.timers = None
        if world > 1:
            gather_outputs(out["image"], out["depth"])
        return out

    # untimed: the whole sequence once, one loop at a time, with statistics: this rank's sample counts (deterministic); also a warm-up
    n_samples_local, n_iters_sum, distinct = 0, 0, {}
    for g in range(n_groups):
        key = tuple(frame_cam[g * F:(g + 1) * F])
        if key not in distinct:
            st = step(g, count=True)
            distinct[key] = (st["n_samples"], len(st["trace"]))
        n_samples_local += distinct[key][0]
        n_iters_sum += distinct[key][1]
    n_iters = n_iters_sum / n_groups
    for i in range(args.warmup):
        step(i % n_groups)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    every = max(1, args.time_every)
    # (multi-GPU: every rank instruments the same loops -- the ranks stay symmetric -- and only the LAST loop, where the pipeline drains
    # anyway, runs with nothing else in flight on its GPU; rank 0's figures go into the line: a rank's shard-sized launches)
    n_instrumented = len(range(0, n_groups, every)) if every else 0
    # A generation-2 pass of CPython's cyclic collector costs tens of milliseconds with torch + numpy loaded and fires on allocation
    # counts, i.e. inside the timed region for some argument combinations and not for others (seen: 20 frames in 14 ms of driver
    # time reported as 50-78 ms).  Collect now and keep the collector off while the clock runs, as `timeit` does.
    import gc
    gc.collect()
    gc.disable()
    reps, marked, excl_set = 1, [], set()
    if ploop is not None:
        stream_of_frames(min(n_groups, max(args.contexts, args.warmup)))           # warm every context
        # The driver's command times K = 20 steps: 13 ms.  The K-step stream is therefore repeated inside the timed region until it
        # lasts >= --min-timed-s: a first un-instrumented pass of the K steps gives the rate, the timed pass then renders
        # ceil(min / rate) repetitions as ONE stream; ms_per_step = elapsed / loops rendered.
        if args.min_timed_s > 0:
            probe_s, _, _, _ = stream_of_frames(n_groups)
            reps = max(1, int(np.ceil(1.15 * args.min_timed_s / max(probe_s, 1e-6))))
        if world > 1:      # every rank must render the same number of loops
            rt = torch.tensor([reps], dtype=torch.int64, device="cpu" if rehearse else dev)
            dist.all_reduce(rt, op=dist.ReduceOp.MAX)
            reps = int(rt.item())
        gather_stats.reset()
        every_eff = every if every else 0
        if every_eff and n_groups * reps < 8:
            every_eff = 0
        elapsed, _, marked, excl_set = stream_of_frames(n_groups, every_eff, reps)
        every = every_eff
    else:
        one = None
        if dloop is not None and args.min_timed_s > 0:
            barrier(); t0 = time.perf_counter(); step(0); barrier()
            one = time.perf_counter() - t0
            reps = max(1, int(np.ceil(args.min_timed_s / max(one * n_groups, 1e-6))))
        n_instrumented = len(range(0, n_groups * reps, every)) if every else 0
        if dloop is not None:
            dloop.prepare_timing(n_instrumented)  # event pairs for the in-place timing of the fused-field launches, created up front
        barrier()
        t0 = time.perf_counter()
        for i in range(n_groups * reps):
            step(i % n_groups, timed=bool(every) and (i % every == 0))
        barrier()
        elapsed = time.perf_counter() - t0
        marked = list(range(0, n_groups * reps, every)) if every else []
    # latency of ONE loop with nothing else in flight (what --pipeline 0 reports as ms_per_step), same sequence, after the timed region
    latency_ms = None
    if dloop is not None:
        lat = []
        for g in range(min(n_groups, 8)):
            barrier()
            t0 = time.perf_counter()
            step(g)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        latency_ms = sorted(lat)[len(lat) // 2]
    gc.enable()
    all_rank_stats = None
    if world > 1:
        # per-rank device-side spans (render_ms: first launch -> last loop done; gather_ms_*: events on the gather stream; how much of
        # the gather time was hidden under rendering): what the first real N-GPU run should be read by
        all_rank_stats = [None] * world
        dist.all_gather_object(all_rank_stats, dict(rank_stats, rank=rank, elapsed_ms=elapsed * 1e3))
        cdev = "cpu" if rehearse else dev
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ns = torch.tensor([n_samples_local], dtype=torch.int64, device=cdev)
        dist.all_reduce(ns)
        n_samples = int(ns.item())
    else:
        n_samples = n_samples_local
    frames_rendered = K * reps
    n_samples *= reps
    ms_per_step = elapsed / frames_rendered * 1e3
    points_per_s = n_samples / elapsed
    n_rays_frame = n_total if args.emulate_rank_of <= 1 else n_local
    rays_per_s = n_rays_frame * frames_rendered / elapsed

    seq = ("one camera at t = 0.5, repeated" if args.static_frame else
           f"test sequence of {n_cams} frames (camera orbit, one time stamp per frame, t = 0 .. 1: every frame selects its own occupancy "
           f"slice, time encoding and the t == 0 canonical rule)")
    result = {
        "metric": METRIC, "value": points_per_s, "unit": "sampled-points/s", "rays_per_s": rays_per_s,
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "repeats": reps, "frames_rendered": frames_rendered, "timed_region_s": elapsed,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f16" if fp16 else "f32", "data": "synthetic",
        "latency_ms_one_loop_at_a_time": latency_ms,
        "config": {"workload": f"dnerf {args.scene}-like {args.size}x{args.size} full-frame inference render, "
                               f"{'-O (fp16 field network, fp16 grid table)' if fp16 else 'fp32'}, {seq}, "
                               f"T_thresh 1e-2, max_steps 1024, dt_gamma 0",
                   "rays": n_total, "sampled_points_per_frame": n_samples / frames_rendered, "loop_iterations": n_iters,
                   "field": field_kind, "loop": loop_kind, "frames_per_loop": F,
                   "rays_per_loop_on_this_gpu": n_loop,
                   "frames_in_flight": ("%d loops of %d frame(s) (the next starts when the newest is down to rays/%d alive)" % (args.contexts, F, args.pipeline)) if ploop is not None else F,
                   "parallelism": (f"ray-tiles x{world}, {F} frames per loop, one all_gather_into_tensor per loop ({args.gather_dtype})" + (" (ONE-GPU REHEARSAL, numbers invalid)" if rehearse else "")) if world > 1
                                  else ("single GPU" + (f" rendering rank 0's shard of a {args.emulate_rank_of}-way ray split (NOT a whole-frame figure)" if args.emulate_rank_of > 1 else ""))},
    }
    if rank == 0:
        def loop_samples(g):     # sampled points of loop g of the (repeated) stream
            g = g % n_groups
            return distinct[tuple(frame_cam[g * F:(g + 1) * F])][0]
        if not every:
            n_excl, excl_samples, over_samples = 0, 0, 0
        elif ploop is not None:
            n_instrumented, n_excl = len(marked), len(excl_set)
            excl_samples = sum(loop_samples(g) for g in excl_set)
            over_samples = sum(loop_samples(g) for g in marked if g not in excl_set)
        else:
            n_excl = n_instrumented = len(marked)
            excl_samples = sum(loop_samples(g) for g in marked)
            over_samples = 0
        result["roofline"], result["kernel_times"] = roofline(timers, fp16, excl_samples, over_samples) if every else (None, {})
        if result["roofline"]:
            result["roofline"]["instrumented_steps"] = (
                f"{n_instrumented} of {n_groups * reps} timed loops (every {every}th" + (", plus the last" if ploop is not None else "") + ")"
                + (f"; {n_excl} of them (the last" + (" and the quarter points" if n_excl > 2 else (" and the midpoint" if n_excl == 2 else "")) + ") rendered with nothing else in flight -- their launches give achieved / frac -- and "
                   f"{n_instrumented - n_excl} overlapped like the uninstrumented ones (the `overlapped` entry)" if ploop is not None else ""))
            # field FLOPs of the whole timed region over its wall time: a lower bound of the kernel's rate that no overlap can inflate
            result["roofline"]["whole_job_mfma_frac"] = FIELD_FLOP_PER_POINT * n_samples / elapsed / 1e12 / result["roofline"]["peak"]
        if world == 1 and dloop is not None and not args.no_secondary and args.emulate_rank_of <= 1:
            result["roofline_secondary"] = marcher_roofline(dloop, grp_o, grp_d, grp_t, n_groups, distinct, frame_cam, F)
            result["grid_gather_rate"] = grid_gather_rate(sc, dev)
            # ONE frame with nothing else in flight (F = 1, --pipeline 0): what an interactive edit preview waits for
            result["latency_ms_one_frame"] = one_frame_latency(sc, field, cam_o, cam_d, cam_t, dev)
            # the reference's own caller, unchanged, over the drop-in operators: model.render -> run_cuda's loop (dnerf/renderer.py:350-376),
            # eval + no_grad + fp16 autocast, NeRFNetwork.forward dispatching each iteration's field evaluation to the fused kernel
            result["reference_shaped"] = reference_shaped(sc, cam_o, cam_d, cam_t)
        elif world == 1 and dloop is not None and not fp16 and args.emulate_rank_of <= 1:
            # --fp32: one frame alone and the reference's caller (no autocast; forward dispatching to the fp32 fused kernel)
            result["latency_ms_one_frame"] = one_frame_latency(sc, field, cam_o, cam_d, cam_t, dev)
            result["reference_shaped"] = reference_shaped(sc, cam_o, cam_d, cam_t, fp32=True)
        if world > 1:
            result["ranks"] = all_rank_stats
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(sc, args.cpu_baseline_side)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def one_frame_latency(sc, field, cam_o, cam_d, cam_t, dev, frames=8):
    """Median wall time of ONE 800x800 frame rendered alone by the device-driven loop (enqueue -> image complete), over the first frames of
    the sequence, after one warm-up pass."""
    from dnerf_amd.renderer import DeviceLoop
    loop = DeviceLoop(sc.model, field, cam_o[0].shape[0], dev, keep_cull_grids=True)
    k = min(frames, len(cam_o))
    for i in range(k):
        loop.render(cam_o[i], cam_d[i], cam_t[i], want_stats=False)
    torch.cuda.synchronize()
    lat = []
    for i in range(k):
        t0 = time.perf_counter()
        loop.render(cam_o[i], cam_d[i], cam_t[i], want_stats=False)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    return sorted(lat)[len(lat) // 2]


def reference_shaped(sc, cam_o, cam_d, cam_t, frames=4, fp32=False):
    """ms per frame of `model.render` (the mirror of the reference's NeRFRenderer.render -> run_cuda, control flow unchanged: march_rays,
    self(xyzs, dirs, time), composite_rays, boolean-mask compaction with its host read-back every iteration) in the reference's `-O`
    inference mode: eval, no_grad, fp16 autocast."""
    model = sc.model.eval()
    dev = cam_o[0].device
    k = min(frames, len(cam_o))

    def frame(i):
        t = torch.tensor([[cam_t[i]]], dtype=torch.float32, device=dev)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=not fp32):
            return model.render(cam_o[i][None], cam_d[i][None], t, staged=False, perturb=False, bg_color=1, max_steps=1024)
    if fp32:      # the reference without -O: NeRFNetwork.forward dispatches to the fp32 fused kernel when asked to (network.py)
        model.fused_inference_f32 = True
    for i in range(k):
        frame(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        frame(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / k
    if fp32:
        model.fused_inference_f32 = False
    return {"ms_per_frame": ms, "rays_per_s": cam_o[0].shape[0] / ms * 1e3, "frames": k,
            "path": "dnerf_amd.NeRFNetwork.render -> run_cuda (reference control flow) on the drop-in operators; field = fused dispatch of NeRFNetwork.forward ("
                    + ("sdn_field_forward_f32, model.fused_inference_f32 = True" if fp32 else "sdn_field_forward_f16") + ")"}


MARCH_BYTES_PER_SAMPLE = 33.0       # 32 B written per emitted sample (xyz, dir, 2 deltas) + ~1 B of occupancy bits per probe
COMPOSITE_BYTES_PER_SAMPLE = 24.0   # sigma, rgb, 2 deltas read per composited sample


def marcher_roofline(dloop, grp_o, grp_d, grp_t, n_groups, distinct, frame_cam, F):
    """`roofline_secondary`: the inference marchers (k_march_rays* before steady mode, k_composite_march* in it; raymarching.cu:701-815,
    819-914) timed in place -- the frame driver's per-iteration event pairs are moved from the field launch to the marcher launch
    (sdn_render_time_kernel) for four frames rendered one at a time after the timed region.  Algorithmic bytes: 33 B per emitted
    sample + 24 B per composited sample (the fused kernel composites iteration k and marches k + 1); the kernels are latency-bound
    (one dependent chain per ray), so the fraction of the HBM roof is small by nature -- the figure to watch is ms per frame."""
    import sdn_backend
    frames = min(4, n_groups)
    t = sdn_backend.KernelTimers()
    sdn_backend.check(sdn_backend.lib.sdn_render_time_kernel(1), "render_time_kernel")
    samples = 0
    try:
        dloop.prepare_timing(frames)
        sdn_backend.timers = t
        for g in range(frames):
            dloop.render(grp_o[g], grp_d[g], grp_t[g], want_stats=False)
            samples += distinct[tuple(frame_cam[g * F:(g + 1) * F])][0]
        torch.cuda.synchronize()
    finally:
        sdn_backend.timers = None
        sdn_backend.check(sdn_backend.lib.sdn_render_time_kernel(0), "render_time_kernel")
    summ = t.summary().get("field_forward_f16")      # (the event pairs keep their name; they bracketed the marcher launches here)
    if not summ:
        return None
    loops, frames = frames, frames * F               # a loop renders F frames
    per_frame_ms = summ["total_ms"] / frames
    bytes_per_frame = (MARCH_BYTES_PER_SAMPLE + COMPOSITE_BYTES_PER_SAMPLE) * samples / frames
    achieved = bytes_per_frame / (per_frame_ms * 1e-3) / 1e9
    return {"kernel": "inference marchers (k_march_rays / k_composite_march, one lane per ray; the 16-lanes-per-ray forms with SDN_GROUP_MARCH=1)", "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "bytes_per_sample": MARCH_BYTES_PER_SAMPLE + COMPOSITE_BYTES_PER_SAMPLE, "launches_per_frame": summ["launches"] / frames,
            "avg_launch_ms": summ["avg_ms"], "ms_per_frame": per_frame_ms, "frames": frames, "frames_per_loop": F, "loops": loops,
            "note": "one loop at a time (nothing else in flight), HIP events around the marcher launch of every iteration; the steady-mode "
                    "entry march (one more launch per frame) is not bracketed; latency-bound kernels: read ms_per_frame, not frac"}


def grid_gather_rate(sc, dev, n_points=196352, launches=20):
    """Stand-alone `grid_encode` forward (k_grid_fwd, gridencoder.cu:87-245) on the fp16 table, clean run (no profiler): the first
    n_points samples the marcher emits for the frame's rays (ray order: neighbours along a ray and across neighbouring rays, what an
    iteration of the loop hands to the encoder), HIP events around every launch.  `achieved` is the ALGORITHMIC gather rate (588 B per
    point: 512 B of table gathers + 12 in + 64 out -- SURVEY 8(d)) over the average launch time; the HBM-side traffic under
    rocprofv3 --pmc is 0.29-0.50 of it (profiles/r02_grid_pmc_summary.json: the table is served by L2 / Infinity Cache).  A second
    figure uses uniformly random points of the figure's bounding box: no locality between neighbouring lanes."""
    import sdn_backend
    import raymarching
    m = sc.model
    with torch.no_grad():
        side = int(round(sc.rays_o.shape[0] ** 0.5))
        band = slice((side // 2 - side // 20) * side, (side // 2 + side // 20) * side)     # the middle tenth of the image rows
        ro, rd = sc.rays_o[band].contiguous(), sc.rays_d[band].contiguous()
        nears, fars = raymarching.near_far_from_aabb(ro, rd, m.aabb_infer, m.min_near)
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        xyzs, _, _, _ = raymarching.march_rays_train(ro, rd, m.bound, m.density_bitfield[32], m.cascade, m.grid_size, nears, fars,
                                                     counter, -1, False, 128, False, 0.0, 1024)
        total = int(counter[0].item())
    x = xyzs[:min(n_points, total)].clone()
    del xyzs
    n_points = int(x.shape[0])
    g = torch.Generator(device="cpu").manual_seed(5)
    lo, hi = torch.tensor([-0.35, -0.6, -0.2]), torch.tensor([0.35, 0.6, 0.2])
    xr = (lo + (hi - lo) * torch.rand(n_points, 3, generator=g)).to(dev)
    enc = m.encoder

    def timed(pts):
        t = sdn_backend.KernelTimers()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            enc(pts, bound=m.bound)
            torch.cuda.synchronize()
            sdn_backend.timers = t
            for _ in range(launches):
                enc(pts, bound=m.bound)
            sdn_backend.timers = None
        summ = t.summary()
        name = next((k for k in summ if k.startswith("grid_encode_fwd")), None)
        return name, (summ[name]["avg_ms"] if name else None)
    name, ms = timed(x)
    _, ms_r = timed(xr)
    if name is None:
        return None
    # (since round 4 an inference forward under -O reads the QUAD copy of the table from its second call on -- two 16-byte gathers per
    #  point and level instead of four 8-byte ones, gridencoder/grid.py; the plain kernel on the same points for comparison)
    from gridencoder import grid as _grid_mod
    _grid_mod.quad_forward = False
    try:
        _, ms_plain = timed(x)
        _, ms_plain_r = timed(xr)
    finally:
        _grid_mod.quad_forward = True
    achieved = GRID_BYTES_PER_POINT["f16"] * n_points / (ms * 1e-3) / 1e9
    out = {"kernel": name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "bytes_per_point_algorithmic": GRID_BYTES_PER_POINT["f16"], "points_per_launch": n_points, "avg_launch_ms": ms, "launches": launches,
           "uniformly_random_points": {"avg_launch_ms": ms_r, "achieved": GRID_BYTES_PER_POINT["f16"] * n_points / (ms_r * 1e-3) / 1e9,
                                       "frac": GRID_BYTES_PER_POINT["f16"] * n_points / (ms_r * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "plain_kernel": {"avg_launch_ms": ms_plain, "frac": GRID_BYTES_PER_POINT["f16"] * n_points / (ms_plain * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "uniformly_random_points_ms": ms_plain_r,
                            "note": "k_grid_fwd on the .half() table (four 8-byte gathers per point and level): the kernel of rounds 1-3"},
           "note": "algorithmic gather rate of the stand-alone op (inference: k_grid_fwd_quad on the QUAD copy of the table, bit-identical outputs) on a frame's own samples, clean run; counter bytes are static (separate rocprofv3 --pmc passes)"}
    pmc = os.path.join(ROOT, "profiles", "r02_grid_pmc_summary.json")
    if os.path.exists(pmc):
        try:
            k = json.load(open(pmc))["kernels"]["k_grid_fwd"]
            pts = k["FETCH_SIZE"]["dispatches"] * 196352.0
            out["traffic_static"] = {"hbm_bytes_per_point_raw": k["hbm"]["hbm_bytes_raw"] / pts, "hbm_bytes_per_point_fetch_x2": k["hbm"]["hbm_bytes_fetch_x2"] / pts,
                                     "source": "profiles/r02_grid_pmc_summary.json (FETCH_SIZE / WRITE_SIZE passes, 196 352 points per launch)"}
        except Exception:
            pass
    return out


def exclusive_frames(k, frames_per_loop=1):
    """Loops of a k-loop stream that are rendered with nothing else in flight so that their launch durations are the kernel's
    own: the LAST one (the pipeline is draining there anyway: holding it back until its predecessors are done costs the stream a
    fraction of one loop's latency; an exclusive FIRST frame -- round 1 and the start of round 2 -- delays every other context by a
    whole loop, 7 % of a 20-frame stream).  A stream of >= 64 loops affords more (each holds the pipeline back for about one loop
    latency): the quarter points when a loop is one frame, the midpoint when a loop is a frame group -- the figure is then an average over
    >= 4 (one frame per loop) / >= 2 x 4 (groups of 4) different frames."""
    if k < 64:
        return {k - 1}
    return {k - 1, k // 2} if frames_per_loop >= 4 else {k - 1, k // 4, k // 2, 3 * k // 4}


def roofline(timers, fp16, points_exclusive, points_overlapped=0):
    """Roofline entry of the dominant TRACKED kernel: achieved = algorithmic bytes (or flops) per launch / average launch
    duration, both from the HIP events recorded around the launches inside the timed region.  The fused field kernel
    evaluates only the live samples of each iteration (device-side list), so its units per launch are the sampled points of the
    instrumented loops / their launches, not the padded slot count the launch is sized for.  points_exclusive / _overlapped: sampled
    points of the instrumented loops that ran alone / while sharing the device."""
    summ = timers.summary()
    if not summ:
        return None, {}
    if "field_forward_f16" in summ:
        # the device-driven loop also launches (and times) one trailing no-op iteration per loop, counted as a launch with zero units
        f = summ["field_forward_f16"]
        f["units"] = int(points_exclusive)
        f["avg_units"] = f["units"] / f["launches"]
    over = summ.pop("field_forward_f16_overlapped", None)
    name = max(summ, key=lambda k: summ[k]["total_ms"])
    s = summ[name]
    if name.startswith("grid_encode_fwd"):
        per_unit = GRID_BYTES_PER_POINT["f16" if name.endswith("f16") else "f32"]
        achieved = per_unit * s["avg_units"] / (s["avg_ms"] * 1e-3) / 1e9
        roof = {"kernel": name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "bytes_per_point": per_unit,
                "avg_launch_ms": s["avg_ms"], "avg_points_per_launch": s["avg_units"]}
    else:
        achieved = FIELD_FLOP_PER_POINT * s["avg_units"] / (s["avg_ms"] * 1e-3) / 1e12
        peak = MFMA_F16_PEAK_TFLOPS if fp16 else (MFMA_F16_PEAK_TFLOPS / 3.0 if os.environ.get("SDN_FIELD_F32", "mfma32") == "split" else MFMA_F32_PEAK_TFLOPS)
        # (--fp32 with SDN_FIELD_F32=split: fp32 operands as fp16 pairs, THREE fp16 MFMAs per product -- the matrix roof of the network's
        #  FLOPs is a third of the fp16 peak)
        roof = {"kernel": name if fp16 else ("field_forward_f32x3" if os.environ.get("SDN_FIELD_F32", "mfma32") == "split" else "field_forward_f32"), "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": None, "flop_per_point": FIELD_FLOP_PER_POINT,
                "avg_launch_ms": s["avg_ms"], "avg_points_per_launch": s["avg_units"]}
        pmc = os.path.join(ROOT, "profiles", PMC_SUMMARY)
        if fp16 and os.path.exists(pmc):  # HBM bytes per point from the committed rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE), NOT measured in this run
            f = json.load(open(pmc))["field_forward_f16"]
            roof["traffic"] = f["hbm_bytes_per_point"] * s["avg_units"]
            roof["traffic_static"] = True
            roof["traffic_fetch_x2"] = f.get("hbm_bytes_per_point_fetch_x2", 0.0) * s["avg_units"] or None   # FETCH_SIZE tallies wide reads at half their bytes (gfx950)
            mp = f.get("matrix_pipe") or {}
            if mp:   # counter evidence of the same kernel (static, from the committed passes): SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES
                roof["matrix_pipe_counters_static"] = {"mfma_busy_over_busy_cu": mp.get("mfma_busy_over_busy_cu"),
                                                       "busy_frac_while_cu_busy": mp.get("matrix_pipe_busy_frac_while_cu_busy"),
                                                       "busy_frac_of_chip_wall_cycles": mp.get("matrix_pipe_busy_frac_of_chip_wall"),
                                                       "source": f"profiles/{PMC_SUMMARY} (frame groups of 4, one loop at a time)"}
            # (static: what a kernel of nothing but v_mfma_f32_32x32x16_f16 sustains on this pool's boxes -- the shader clock under matrix load
            #  is 1.74-1.77 GHz, not the 2.4 GHz of the nominal peak that `frac` keeps as its denominator)
            roof["sustained_mfma_only_static"] = {"tflops": 1754.0, "frac_of_peak": 0.70, "shader_mhz_under_load": 1734, "boxes": "0.70-0.72 over five boxes",
                                                  "source": "profiles/r04_mfma_roof_probe.txt (tools/probes/mfma_peak_probe.hip)"}
            roof["traffic_unit"] = f"HBM bytes per launch = bytes per point of profiles/{PMC_SUMMARY} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command) x points per launch; static, not measured in this run"
        if over and points_overlapped:  # the same launches while other loops' kernels share the device: durations are not the kernel's own
            units = points_overlapped / over["launches"]
            ach = FIELD_FLOP_PER_POINT * units / (over["avg_ms"] * 1e-3) / 1e12
            roof["overlapped"] = {"avg_launch_ms": over["avg_ms"], "launches": over["launches"], "achieved": ach,
                                  "frac": ach / peak,
                                  "note": "loops in flight share the CUs: this is the figure rocprofv3 --stats of the default command averages towards"}
            summ["field_forward_f16_overlapped"] = over
    return roof, summ


def cpu_baseline(sc, side):
    """The reference's pure-PyTorch renderer (`NeRFRenderer.run`, dnerf/renderer.py:129-258: 128 uniform samples per ray, no occupancy
    grid, max_ray_batch 4096) on the host cores, as SURVEY 8(d) lays it out: 1 warm-up + 3 timed frames at 64 x 64 (BASELINE config 1)
    and the 800 x 800 camera -- of which a bounded side x side sample is timed by default (the whole frame is 640 000 rays x 128 samples,
    ~160 s on 64 cores; `--cpu-baseline-side 800` runs it whole), keeping the default bench within a few minutes."""
    from dnerf_amd import scene
    from oracle import render as orender
    cores = os.cpu_count() or 1
    cores = min(cores, 64)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    torch.set_num_threads(cores)
    state = orender.state_of(sc.model)

    def frame(s):
        ro, rd = scene.get_rays(sc.pose, scene.intrinsics(s, s), s, s)
        t0 = time.perf_counter()
        orender.render_run_cpu(state, ro, rd, 0.5, threads=cores)
        return ro.shape[0], time.perf_counter() - t0
    frame(64)                                                  # warm-up (page-in, thread pools)
    small = [frame(64) for _ in range(3)]
    n64, t64 = small[0][0], sorted(t for _, t in small)[1]     # median of the three timed 64 x 64 frames
    n, dt = frame(side)
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": n * 128 / dt, "unit": "sampled-points/s", "rays_per_s": n / dt, "cores": cores, "cpu_model": model, "kind": "port",
            "sample": f"{side}x{side} rays of the 800x800 frame's camera and weights ({'the whole frame' if side >= 800 else 'bounded sample; the whole frame of 640 000 rays is projected at %.0f s' % (640000 / (n / dt))}), "
                      f"128 uniform samples/ray (the reference's non-cuda_ray sampler, upsample_steps=0, max_ray_batch=4096), fp32, {dt:.1f} s; "
                      f"it samples empty space too, so rays/s is the like-for-like figure",
            "config1_64x64": {"rays_per_s": n64 / t64, "value": n64 * 128 / t64, "unit": "sampled-points/s", "frames": "1 warm-up + 3 timed (median)",
                              "ms_per_frame": t64 * 1e3}}


if __name__ == "__main__":
    main()
