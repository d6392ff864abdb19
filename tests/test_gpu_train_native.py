"""NativeTrainStep (dnerf_amd/train_native.py -> sdn_train_step_f16, csrc/train.hip) against the reference-shaped training step:
the mirror network's op-by-op render under fp16 autocast + torch autograd + torch.optim.Adam + GradScaler (dnerf/utils.py:38-125,
nerf/utils.py:880-906).  The native step runs the reference's op sequence with its dtypes but its own kernels, so agreement is to
fp16 accumulation-order tolerance (stated per check), not bit for bit; integer results (sample counts, the ray table) are exact."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_RAYS = 1024


def _setup(seed=0, lr=1e-3, ema=None):
    from dnerf_amd.bench_scene import build_scene
    from dnerf_amd.network import NeRFNetwork
    sc = build_scene(H=32, W=32, device="cuda", seed=seed)
    model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    model.load_state_dict(sc.model.state_dict())
    opt = torch.optim.Adam(model.get_params(10 * lr, lr), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda")
    target = torch.rand(1, N_RAYS, 3, generator=torch.Generator().manual_seed(4)).cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model.render(sc.rays_o[None], sc.rays_d[None], sc.time, staged=False, perturb=False, bg_color=1, force_all_rays=False)
    model.mean_count = int(model.step_counter[0, 0].item()) + 256
    model.local_step = 0
    model.step_counter.zero_()
    return sc, model, opt, scaler, target


def _eager_backward(model, sc, target, scaler, time=None, bg_color=1):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        out = model.render(sc.rays_o[None], sc.rays_d[None], sc.time if time is None else time, staged=False, perturb=False, bg_color=bg_color,
                           force_all_rays=False, max_steps=1024)
        loss = torch.nn.MSELoss(reduction="none")(out["image"], target).mean(-1).mean()
    scaler.scale(loss).backward()
    return out, loss


def _flat_to_layers(flat, in_cols, in_ld, width, n_hidden, out_rows):
    """The operator's flat layout [W, in_ld] ++ (L-1) x [W, W] ++ [16, W] -> the Linear layers' weight-shaped pieces."""
    parts, at = [flat[:width * in_ld].view(width, in_ld)[:, :in_cols]], width * in_ld
    for _ in range(n_hidden - 1):
        parts.append(flat[at:at + width * width].view(width, width))
        at += width * width
    parts.append(flat[at:at + 16 * width].view(16, width)[:out_rows])
    return parts


def _native_grads(step, model):
    """name -> fp32 gradient (still multiplied by the loss scale) from the step's workspace."""
    rows = model.encoder.embeddings.shape[0]
    g = {"encoder.embeddings": step.view("g_table", torch.float16, (rows, 2)).float()}
    for i, w in enumerate(_flat_to_layers(step.view("g_deform", torch.float16, (128 * 80 + 6 * 128 * 128 + 16 * 128,)), 76, 80, 128, 7, 3)):
        g[f"deform_net.{i}.weight"] = w.float()
    # the colour MLP's input layer is kept as [SH 16 | one zero column under the log-density | geo 15] (csrc/train.hip: the input row is
    # [SH | sigma-MLP output])
    for i, w in enumerate(_flat_to_layers(step.view("g_color", torch.float16, (64 * 32 + 64 * 64 + 16 * 64,)), 32, 32, 64, 2, 3)):
        g[f"color_net.{i}.weight"] = (torch.cat([w[:, :16], w[:, 17:]], dim=1) if i == 0 else w).float()
    g["sigma_net.0.weight"] = step.view("g_sigma0", torch.float16, (64, 32)).float()
    g["sigma_net.1.weight"] = step.view("g_sigma1", torch.float16, (16, 64)).float()
    return g


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("time", [None, 0.0])
def test_forward_and_gradients_match_the_autograd_step(time):
    """Forward: sample count and the per-ray (offset, count) table exact, image 5e-4 (measured 4e-5), loss 1e-4 (measured equal).
    Backward: every weight gradient within 0.5 % of the autograd gradient in the L2 norm (measured 1e-5 .. 9e-4: fp16 gradients and
    the summation order of ~9000 samples), the table gradient likewise; at time == 0 the deformation MLP gets no gradient
    (dnerf/network.py:140)."""
    from dnerf_amd.train_native import NativeTrainStep
    sc, model, opt, scaler, target = _setup()
    tval = sc.time if time is None else torch.tensor([[time]], dtype=torch.float32, device="cuda")
    out, loss = _eager_backward(model, sc, target, scaler, time=tval)
    ref_counter = model.step_counter[0].clone()
    ref = {k: v.grad.detach().clone() if v.grad is not None else None for k, v in model.named_parameters()}
    model.local_step = 0
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False)
    got_loss = step(sc.rays_o, sc.rays_d, target, tval, grads_only=True)
    torch.cuda.synchronize()
    assert torch.equal(model.step_counter[0], ref_counter)
    np.testing.assert_allclose(float(got_loss), float(loss.detach()), rtol=1e-4)
    assert float((step.image - out["image"][0]).abs().max()) < 5e-4
    grads = _native_grads(step, model)
    for k, want in ref.items():
        if want is None:
            assert time == 0.0 and k.startswith("deform_net")
            continue
        assert _rel(grads[k], want.float()) < 5e-3, (k, _rel(grads[k], want.float()))
    # nothing was updated
    assert float(step.adam_steps.sum()) == 0 and scaler.get_scale() == 65536.0


def test_config3_at_its_real_size_4096_rays_of_the_800x800_camera(monkeypatch):
    """BASELINE config 3 as `bench.py --mode train` runs it: 4096 rays drawn with torch.randint(seed 0) from the 800 x 800 camera of
    the jumpingjacks-like scene, perturbed starts, `mean_count` from two first-epoch steps.  Native step against the autograd step
    (op-by-op render under autocast) on the same rays and the same per-ray offsets: sample count and ray table exact, loss 1e-4,
    image 5e-4, every gradient within 0.5 % in L2 -- the bars of the 1024-ray test, at four times its batch."""
    from dnerf_amd.bench_scene import build_scene
    from dnerf_amd.network import NeRFNetwork
    from dnerf_amd.train_native import NativeTrainStep
    n_rays = 4096
    sc = build_scene(H=800, W=800, device="cuda", seed=0)
    idx = torch.randint(0, sc.rays_o.shape[0], (n_rays,), generator=torch.Generator(device="cpu").manual_seed(0)).cuda()
    rays_o, rays_d = sc.rays_o[idx].contiguous(), sc.rays_d[idx].contiguous()
    target = torch.rand(1, n_rays, 3, generator=torch.Generator(device="cpu").manual_seed(2)).cuda()
    model = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    model.load_state_dict(sc.model.state_dict())
    opt = torch.optim.Adam(model.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
    scaler = torch.amp.GradScaler("cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for _ in range(2):        # first-epoch steps: unknown budget (dnerf/renderer.py:289-296)
            model.render(rays_o[None], rays_d[None], sc.time, staged=False, perturb=True, bg_color=1, force_all_rays=False, max_steps=1024)
    model.mean_count = int(model.step_counter[:2, 0].sum().item() / 2)     # update_extra_state, dnerf/renderer.py:550-552
    assert 6000 < model.mean_count < 14000            # the bench's ~9 070 samples per step
    noises = torch.rand(n_rays, generator=torch.Generator(device="cpu").manual_seed(1)).cuda()
    # autograd step with the same offsets: the operator draws them with torch.rand(N) (raymarching.py:190); replay `noises` there
    import raymarching.raymarching as rm_mod
    model.local_step = 0
    model.step_counter.zero_()
    model.zero_grad(set_to_none=True)
    real_rand = torch.rand

    def fake_rand(*size, **kw):
        n = size[0] if len(size) == 1 and isinstance(size[0], int) else None
        return noises.clone() if n == n_rays else real_rand(*size, **kw)
    monkeypatch.setattr(rm_mod.torch, "rand", fake_rand)
    with torch.autocast("cuda", dtype=torch.float16):
        out = model.render(rays_o[None], rays_d[None], sc.time, staged=False, perturb=True, bg_color=1, force_all_rays=False, max_steps=1024)
        loss = torch.nn.MSELoss(reduction="none")(out["image"], target).mean(-1).mean()
    monkeypatch.undo()
    scaler.scale(loss).backward()
    ref_counter = model.step_counter[0].clone()
    ref = {k: v.grad.detach().clone() for k, v in model.named_parameters()}
    model.local_step = 0
    step = NativeTrainStep(model, opt, scaler, n_rays, "cuda", perturb=True)
    step.noises = noises
    got = step(rays_o, rays_d, target, sc.time, grads_only=True)
    torch.cuda.synchronize()
    assert torch.equal(model.step_counter[0], ref_counter), (model.step_counter[0].tolist(), ref_counter.tolist())
    np.testing.assert_allclose(float(got), float(loss.detach()), rtol=1e-4)
    assert float((step.image - out["image"][0]).abs().max()) < 5e-4
    grads = _native_grads(step, model)
    worst = {k: _rel(grads[k], ref[k].float()) for k in ref}
    assert max(worst.values()) < 5e-3, worst


def test_full_step_is_torch_adam_on_the_native_gradients():
    """The optimizer pass against torch.optim.Adam (the reference's, non-fused) fed with the step's own gradients: parameters 1e-6
    relative to the update size, moments 1e-5; the fp16 copies equal the rounded parameters; the table's gradient accumulator is
    cleared; the scaler's growth tracker advanced."""
    from dnerf_amd.train_native import NativeTrainStep
    sc, model, opt, scaler, target = _setup()
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False)
    names = [n for n, _ in model.named_parameters()]
    ref_model = copy.deepcopy(model)
    ref_opt = torch.optim.Adam(ref_model.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
    for it in range(3):
        step(sc.rays_o, sc.rays_d, target, sc.time, grads_only=True)
        grads = _native_grads(step, model)
        inv = 1.0 / scaler.get_scale()
        for k, p in ref_model.named_parameters():
            p.grad = (grads[k] * inv).to(p.dtype).reshape(p.shape).clone()
        ref_opt.step()
        # the full step recomputes the same gradients (same batch, same parameters) up to the order of the table's fp16 atomics
        step(sc.rays_o, sc.rays_d, target, sc.time)
        torch.cuda.synchronize()
        got, want = dict(model.named_parameters()), dict(ref_model.named_parameters())
        for k in names:
            a, b = got[k].detach(), want[k].detach()
            if k == "encoder.embeddings":
                # an entry whose tiny gradient changes sign between the two evaluations moves the other way: bounded count
                d = (a - b).abs()
                assert float((d > 1e-6).float().mean()) < 5e-3 and float(d.max()) <= 2.1e-2 * (it + 1)
            else:
                assert float((a - b).abs().max()) <= 2e-5, (k, it, float((a - b).abs().max()))
        for k in names:
            if k == "encoder.embeddings":
                continue
            p, q = got[k], want[k]
            for key in ("exp_avg", "exp_avg_sq"):
                x, y = opt.state[p][key], ref_opt.state[q][key]
                assert float((x - y).abs().max()) <= 1e-3 * float(y.abs().max()) + 1e-12, (k, key)
    assert float(step.adam_steps[0]) == 3 and float(step.adam_steps[1]) == 3
    assert int(scaler._growth_tracker) == 3
    rows = model.encoder.embeddings.shape[0]
    assert float(step.view("g_table", torch.float16, (rows, 2)).abs().max()) == 0
    assert torch.equal(step.view("w_table", torch.float16, (rows, 2)), model.encoder.embeddings.detach().half())
    assert torch.equal(step.view("w_sigma0", torch.float16, (64, 32)), model.sigma_net[0].weight.detach().half())
    step.sync_optimizer_state()
    assert all(float(opt.state[p]["step"]) == 3 for p in step.params)


def test_training_tracks_the_reference_shaped_step():
    """20 steps of the native step against 20 eager steps (autocast + autograd + torch Adam + GradScaler) from the same state on
    the same batch: the loss falls, both trajectories agree to 1 % of the loss at every step and their total decrease to 25 %
    (a random target: the loss moves slowly)."""
    from dnerf_amd.train_native import NativeTrainStep
    sc, model, opt, scaler, target = _setup(lr=1e-3)
    ref_model = copy.deepcopy(model)
    ref_opt = torch.optim.Adam(ref_model.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
    ref_scaler = torch.amp.GradScaler("cuda")
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False)
    a, b = [], []
    for _ in range(20):
        a.append(float(step(sc.rays_o, sc.rays_d, target, sc.time)))
        _, loss = _eager_backward(ref_model, sc, target, ref_scaler)
        ref_scaler.step(ref_opt)
        ref_scaler.update()
        b.append(float(loss))
    assert a[-1] < a[0] and b[-1] < b[0]
    np.testing.assert_allclose(a, b, rtol=1e-2)
    assert abs((a[0] - a[-1]) - (b[0] - b[-1])) < 0.25 * (b[0] - b[-1])
    assert model.local_step == 20


def test_canonical_frame_freezes_the_deformation_mlp_and_overflow_skips_the_step():
    from dnerf_amd.train_native import NativeTrainStep
    sc, model, opt, scaler, target = _setup()
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False, ema_decay=0.95)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    step(sc.rays_o, sc.rays_d, target, 0.0)
    torch.cuda.synchronize()
    for k, v in model.named_parameters():
        same = torch.equal(v.detach(), before[k])
        assert same == k.startswith("deform_net"), k
    assert step.adam_steps.tolist() == [1.0, 0.0]
    # torch_ema: decay = min(0.95, (1 + 1) / (10 + 1)); shadow -= (1 - decay) * (shadow - param)
    d = min(0.95, 2 / 11)
    for i, p in enumerate(step.params):
        k = [n for n, q in model.named_parameters() if q is p][0]
        want = before[k] - (1 - d) * (before[k] - p.detach())
        assert float((step.ema_shadow[i] - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-9
    # a loss scale that overflows fp16 gradients: nothing moves, the scale backs off (AmpKernels.cu amp_update_scale)
    scaler._scale.fill_(2.0 ** 40)
    mid = {k: v.detach().clone() for k, v in model.named_parameters()}
    step(sc.rays_o, sc.rays_d, target, 0.5)
    torch.cuda.synchronize()
    for k, v in model.named_parameters():
        assert torch.equal(v.detach(), mid[k]), k
    assert step.adam_steps.tolist() == [1.0, 0.0]
    assert scaler.get_scale() == 2.0 ** 39 and int(scaler._growth_tracker) == 0
    rows = model.encoder.embeddings.shape[0]
    assert float(step.view("g_table", torch.float16, (rows, 2)).float().abs().nan_to_num(0, 0, 0).max()) == 0


def test_reference_training_fixture_gradients():
    """The reference's own training branch (fixture `caller_train.npz`, fp32 autograd of dnerf/renderer.py + dnerf/network.py run in
    the build container): the native fp16 step on the same rays, per-ray offsets and target reproduces its sample counts exactly,
    its loss to 1e-3, its MLP weight gradients to 8 % in the L2 norm (an fp16 backward through eight layers against the fixture's
    fp32 one: the first deformation layer, at the end of the chain, measures 4 %) and its hash-grid gradient (per level, on the sampled
    rows, and in the set of touched rows) at the fp16 bars stated below."""
    from caller_fixtures import fixture_model, fixture_scene, load
    from dnerf_amd.train_native import NativeTrainStep
    fx = load("train")
    model_bits = fixture_model("cuda")
    model = model_bits[0]
    sc = fixture_scene("cuda", model_bits=model_bits)
    sel = torch.from_numpy(fx["sel"]).long().cuda()
    ro, rd = sc.rays_o[sel].contiguous(), sc.rays_d[sel].contiguous()
    target = torch.from_numpy(fx["target"]).cuda()
    try:
        model.train()
        model.local_step, model.mean_count = 0, int(fx["perturb_counter"][0])
        model.step_counter.zero_()
        opt = torch.optim.Adam(model.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
        step = NativeTrainStep(model, opt, torch.amp.GradScaler("cuda"), ro.shape[0], "cuda", perturb=True)
        step.noises = torch.from_numpy(fx["noises"]).cuda()
        loss = step(ro, rd, target, sc.time, grads_only=True)
        torch.cuda.synchronize()
        assert model.step_counter[0].cpu().numpy().tolist() == fx["budget_counter"].tolist()
        np.testing.assert_allclose(float(loss), float(fx["budget_loss"]), rtol=1e-3)
        assert float(np.abs(step.image.cpu().numpy() - fx["budget_image"]).max()) < 2e-3
        grads = _native_grads(step, model)
        for k, g in grads.items():
            if k == "encoder.embeddings":
                continue
            ref = torch.from_numpy(fx[f"perturb_grad_{k}"]).cuda() * 65536.0     # `budget` repeats `perturb` with M = its sample count
            assert _rel(g, ref) < 8e-2, (k, _rel(g, ref))
        # the hash-grid gradient (gridencoder.cu:248-340 through the reference network's autograd; the fixture keeps per-level
        # (sum, sum |.|, sum of squares), 16 384 sampled rows and the count of touched rows): the native step's fp16 accumulator,
        # unscaled.  Bars are fp16-distance (an fp16 backward through the sigma / colour MLPs into fp16 atomics against an fp32 one):
        # per level the L2 norm and the absolute sum to 0.5 % (measured 0.07 %), the sampled rows to 1.5 % in L2 (measured 0.4 %); the set of
        # touched rows is an integer property of the samples and must be the fixture's up to entries whose fp16 gradient underflows
        # (measured 76 924 of 76 929).
        ge = (grads["encoder.embeddings"].double() / 65536.0).cpu().numpy()
        off = model.encoder.offsets.cpu().numpy()
        ref_lv = fx["perturb_grad_emb_levels"]
        lv = np.stack([[ge[off[l]:off[l + 1]].sum(), np.abs(ge[off[l]:off[l + 1]]).sum(), (ge[off[l]:off[l + 1]] ** 2).sum()] for l in range(16)])
        l2_err = np.abs(np.sqrt(lv[:, 2]) - np.sqrt(ref_lv[:, 2])) / np.sqrt(ref_lv[:, 2])
        l1_err = np.abs(lv[:, 1] - ref_lv[:, 1]) / ref_lv[:, 1]
        sum_err = np.abs(lv[:, 0] - ref_lv[:, 0]) / ref_lv[:, 1].max()
        rows, vals = fx["perturb_grad_emb_rows"], fx["perturb_grad_emb_vals"].astype(np.float64)
        row_err = float(np.linalg.norm(ge[rows] - vals) / np.linalg.norm(vals))
        nnz, ref_nnz = int((np.abs(ge).sum(1) != 0).sum()), int(fx["perturb_grad_emb_nnz_rows"])
        report = dict(level_l2=float(l2_err.max()), level_l1=float(l1_err.max()), level_sum=float(sum_err.max()), rows_l2=row_err, nnz=nnz, ref_nnz=ref_nnz)
        assert l2_err.max() < 5e-3 and l1_err.max() < 5e-3 and sum_err.max() < 5e-3 and row_err < 1.5e-2, report
        assert nnz <= ref_nnz and nnz >= 0.999 * ref_nnz, report
        print("table gradient vs reference fixture:", report)
    finally:
        model.eval()
        model.mean_count, model.local_step = 0, 0


def _dp_rank(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # one-GPU rehearsal: both ranks on cuda:0, gradients staged through the host
    try:
        import tests_support  # noqa: F401
        from dnerf_amd.bench_scene import build_scene
        from dnerf_amd.dist import GradSync
        from dnerf_amd.network import NeRFNetwork
        from dnerf_amd.train_native import NativeTrainStep
        sc = build_scene(H=64, W=64, device="cuda", seed=0)
        idx = torch.randint(0, sc.rays_o.shape[0], (N_RAYS,), generator=torch.Generator().manual_seed(rank)).cuda()   # own batch per rank
        rays_o, rays_d = sc.rays_o[idx].contiguous(), sc.rays_d[idx].contiguous()
        target = torch.rand(N_RAYS, 3, generator=torch.Generator().manual_seed(50 + rank)).cuda()
        m = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
        m.load_state_dict(sc.model.state_dict())
        with torch.no_grad():
            m.deform_net[0].weight.mul_(1 + 0.01 * rank)            # replicas start different; the broadcast makes them equal
        sync = GradSync(m)
        sync.broadcast_parameters()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            m.render(rays_o[None], rays_d[None], sc.time, staged=False, perturb=False, bg_color=1, force_all_rays=False, max_steps=1024)
        m.mean_count = int(m.step_counter[0, 0].item()) + 256       # headroom: the ranks' batches differ
        m.local_step = 0
        opt = torch.optim.Adam(m.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
        step = NativeTrainStep(m, opt, torch.amp.GradScaler("cuda"), N_RAYS, "cuda", perturb=False, grad_sync=sync)
        # rank 1 trains the canonical frame in its third step: its deformation MLP has no gradient of its own then, rank 0's arrives
        times = [0.5, 0.5, 0.0 if rank == 1 else 0.5, 0.5, 0.5]
        losses = [float(step(rays_o, rays_d, target, t)) for t in times]
        flat = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat)
        same = all(torch.equal(parts[0], q) for q in parts)
        np.save(os.path.join(out_dir, f"dp_{rank}.npy"), np.array([float(same), losses[0], losses[-1], float(step.adam_steps[1])]))
    finally:
        dist.destroy_process_group()


def test_data_parallel_native_step_keeps_replicas_identical(tmp_path):
    """Two ranks (both on this one GPU, gloo with host staging: mechanics only) train on different ray batches: backward, all-reduce
    of the fp16 gradient buffers, optimizer pass with the world size as divisor.  Parameters stay bit-identical across the ranks, the
    loss goes down, and the deformation MLP is stepped every time (also when one rank's batch is the canonical frame)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        same, first, last, deform_steps = np.load(tmp_path / f"dp_{r}.npy")
        assert same == 1.0 and deform_steps == 5.0, (same, deform_steps)
    assert np.load(tmp_path / "dp_0.npy")[2] < np.load(tmp_path / "dp_0.npy")[1]


def test_per_ray_background_budget_overflow_and_rebuild():
    """Per-ray background colours (utils.py:74) against the autograd step; a sample budget smaller than the batch needs (the rays
    past it are dropped, raymarching.py:200-233, the step stays finite); a changed `mean_count` re-sizes the step's buffers without
    losing the parameters' fp16 copies; a changed learning rate is picked up from the optimizer's groups."""
    from dnerf_amd.train_native import NativeTrainStep
    sc, model, opt, scaler, target = _setup()
    bg = torch.rand(N_RAYS, 3, generator=torch.Generator().manual_seed(9)).cuda()
    out, loss = _eager_backward(model, sc, target, scaler, bg_color=bg[None])
    ref = {k: v.grad.detach().clone() for k, v in model.named_parameters()}
    model.local_step = 0
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False, bg_color=bg)
    got = step(sc.rays_o, sc.rays_d, target, sc.time, grads_only=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(got), float(loss), rtol=1e-3)
    grads = _native_grads(step, model)
    for k in ("color_net.2.weight", "sigma_net.0.weight", "deform_net.3.weight"):
        assert _rel(grads[k], ref[k].float()) < 2e-2, k
    # overflow: a budget of about a third of the samples
    full = int(model.step_counter[(model.local_step - 1) % 16, 0])
    model.mean_count = max(128, full // 3)
    before = model.sigma_net[0].weight.detach().clone()
    l2 = step(sc.rays_o, sc.rays_d, target, sc.time)
    torch.cuda.synchronize()
    assert step._M == model.mean_count + (128 - model.mean_count % 128) and np.isfinite(float(l2))
    kept = step.view("rays", torch.int32, (N_RAYS, 3))
    assert int((kept[:, 1] + kept[:, 2] <= step._M).sum()) < N_RAYS            # some rays did not fit ...
    assert not torch.equal(model.sigma_net[0].weight.detach(), before)          # ... the others trained
    assert torch.equal(step.view("w_sigma0", torch.float16, (64, 32)), model.sigma_net[0].weight.detach().half())
    # learning rate 0 for the MLPs: only the table moves
    for g in opt.param_groups:
        if any(p is model.sigma_net[0].weight or p is model.deform_net[0].weight or p is model.color_net[0].weight for p in g["params"]):
            g["lr"] = 0.0
    mid = {k: v.detach().clone() for k, v in model.named_parameters()}
    step(sc.rays_o, sc.rays_d, target, sc.time)
    torch.cuda.synchronize()
    for k, v in model.named_parameters():
        assert torch.equal(v.detach(), mid[k]) == (k != "encoder.embeddings"), k


def test_perturbed_steps_are_reproducible_and_differ_by_seed():
    from dnerf_amd.train_native import NativeTrainStep
    losses = []
    for seed in (1, 1, 2):
        sc, model, opt, scaler, target = _setup()
        step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=True, seed=seed)
        losses.append([float(step(sc.rays_o, sc.rays_d, target, sc.time, grads_only=True)) for _ in range(2)])
    assert losses[0] == losses[1] and losses[0] != losses[2]
    assert losses[0][0] == losses[0][1]             # grads_only does not advance the noise stream (same step index)


def test_marching_the_next_batch_ahead_changes_nothing():
    """`prefetch()` (phase 1 of batch k+1 on a side stream beside step k, second sample buffer) against the plain sequence, over
    batches that differ in rays and time: losses and parameters bit for bit, counter ring included."""
    from dnerf_amd.train_native import NativeTrainStep
    runs = []
    for ahead in (False, True):
        sc, model, opt, scaler, target = _setup()
        step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=True, seed=3)
        g = torch.Generator().manual_seed(11)
        batches = []
        for k in range(5):
            idx = torch.randperm(sc.rays_o.shape[0], generator=g)[:N_RAYS].cuda()
            batches.append((sc.rays_o[idx].contiguous(), sc.rays_d[idx].contiguous(), target[0, idx % N_RAYS].contiguous(), [0.5, 0.25, 0.0, 0.75, 0.5][k]))
        losses = []
        for k, (ro, rd, tg, t) in enumerate(batches):
            losses.append(step(ro, rd, tg, t).clone())
            if ahead and k + 1 < len(batches):
                step.prefetch(batches[k + 1][0], batches[k + 1][1], batches[k + 1][3])
        torch.cuda.synchronize()
        runs.append(([float(x) for x in losses], {k: v.detach().clone() for k, v in model.named_parameters()}, model.step_counter.clone()))
    # the table gradient's fp16 atomics are the one order-dependent sum of the step: compare what does not pass through them exactly
    # and the rest to the usual bound
    assert runs[0][0][0] == runs[1][0][0] and torch.equal(runs[0][2], runs[1][2])
    np.testing.assert_allclose(runs[0][0], runs[1][0], rtol=2e-3)
    for k in runs[0][1]:
        d = (runs[0][1][k] - runs[1][1][k]).abs()
        assert float(d.max()) <= (1.1e-1 if k == "encoder.embeddings" else 1.1e-2), k


def _step_state(step, model, opt, scaler):
    """Everything a native step reads: fp32 parameters, their fp16 copies, both Adam moments, the device step counts, the scaler's
    scale and growth tracker, the table's gradient accumulator."""
    rows = model.encoder.embeddings.shape[0]
    s = {"adam_steps": step.adam_steps.clone(), "scale": scaler._scale.clone(), "tracker": scaler._growth_tracker.clone(),
         "g_table": step.view("g_table", torch.float16, (rows, 2)).clone(), "w_table": step.view("w_table", torch.float16, (rows, 2)).clone(),
         "w_deform": step.view("w_deform", torch.float16, (128 * 80 + 6 * 128 * 128 + 16 * 128,)).clone(),
         "w_sigma0": step.view("w_sigma0", torch.float16, (64, 32)).clone(), "w_sigma1": step.view("w_sigma1", torch.float16, (16, 64)).clone(),
         "w_color": step.view("w_color", torch.float16, (64 * 32 + 64 * 64 + 16 * 64,)).clone()}
    for n, p in model.named_parameters():
        s["p." + n], s["m." + n], s["v." + n] = p.detach().clone(), opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()
    return s


def test_checkpoint_round_trip_through_the_optimizer_and_scaler_state():
    """Three native steps, `sync_optimizer_state()`, state dicts of model / optimizer / scaler into fresh objects,
    `refresh(optimizer_state=True)`, three more steps, against six uninterrupted steps.

    What is asserted, and why (round 2 loosened this test after it failed; `tools/resume_floor.py` then measured the cause):
    * BEFORE any resumed step, everything the step reads is restored BITWISE (`_step_state`): parameters, fp16 copies, both Adam
      moments, step counts, loss scale, growth tracker, gradient accumulator.  A moment, copy or count that is not restored fails here.
    * The one order-dependent sum of a step is the table gradient's fp16 atomics, and Adam with eps = 1e-15 moves an element by
      ~lr x sign(gradient) in its first steps, so gradient noise flips whole steps: two UNINTERRUPTED six-step runs from the same
      state differ by up to 2.5 lr in single table entries and 1.8 lr in deformation-MLP weights, 4-9 % of whose elements differ by
      more than 1e-5 (measured: profiles/r03_resume_floor.txt).  That run-to-run floor is measured here (run B against run A)
      and the resumed run must stay within 3x of it per parameter, or inside the floor's recorded envelope -- it measures 20-50x BELOW
      the floor, since its first three steps are run A's own."""
    from dnerf_amd.network import NeRFNetwork
    from dnerf_amd.train_native import NativeTrainStep

    def run(n, stop_at=None):
        sc, model, opt, scaler, target = _setup()
        step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False)
        saved = None
        for k in range(n):
            step(sc.rays_o, sc.rays_d, target, sc.time)
            if k + 1 == stop_at:
                torch.cuda.synchronize()
                step.sync_optimizer_state()
                saved = {"state": _step_state(step, model, opt, scaler), "model": copy.deepcopy(model.state_dict()), "opt": copy.deepcopy(opt.state_dict()),
                         "scaler": scaler.state_dict(), "mean_count": model.mean_count, "local_step": model.local_step}
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in model.named_parameters()}, saved, (sc, target)

    A, saved, (sc, target) = run(6, stop_at=3)
    B, _, _ = run(6)
    # resume in fresh objects
    model2 = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    model2.load_state_dict(saved["model"])
    model2.mean_count, model2.local_step = saved["mean_count"], saved["local_step"]
    opt2 = torch.optim.Adam(model2.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
    scaler2 = torch.amp.GradScaler("cuda")
    step2 = NativeTrainStep(model2, opt2, scaler2, N_RAYS, "cuda", perturb=False)
    opt2.load_state_dict(saved["opt"])
    scaler2.load_state_dict(saved["scaler"])
    step2.refresh(optimizer_state=True)
    torch.cuda.synchronize()
    assert step2.adam_steps.tolist() == [3.0, 3.0]
    restored = _step_state(step2, model2, opt2, scaler2)
    not_restored = [k for k, v in saved["state"].items() if not torch.equal(restored[k], v)]
    assert not not_restored, not_restored
    for _ in range(3):
        step2(sc.rays_o, sc.rays_d, target, sc.time)
    torch.cuda.synchronize()
    # The floor of THIS process (run B against run A) is itself one draw of the noise: for a small tensor it can come out several times
    # below its usual size (seen once in ~15 runs of the suite: a resumed run 3.x times above that draw).  The resumed run therefore has to
    # stay within 3x of this draw OR inside the floor's recorded envelope (profiles/r03_resume_floor.txt: max 2.49 lr, mean 0.0054 lr over all
    # parameters; the resumed run measures 0.05 lr / 2.4e-5 lr) -- a moment, copy or step count that is not restored moves every element by
    # ~lr per step (mean >= 0.3 lr) and fails both, besides failing the bitwise check above.
    for k, v in model2.named_parameters():
        lr = 1e-2 if k == "encoder.embeddings" else 1e-3
        floor, got = (B[k] - A[k]).abs(), (v.detach() - A[k]).abs()
        assert float(got.max()) <= max(3 * float(floor.max()), 2.5 * lr) + 1e-7, (k, float(got.max()), float(floor.max()))
        assert float(got.mean()) <= max(3 * float(floor.mean()), 0.006 * lr) + 1e-9, (k, float(got.mean()), float(floor.mean()))
    assert float(step2.adam_steps[0]) == 6 and int(scaler2._growth_tracker) == 6


def test_deterministic_mode_makes_runs_and_a_resumed_run_bit_identical():
    """`NativeTrainStep(deterministic=True)` (SDN_DETERMINISTIC=1): the table gradient -- the one order-dependent sum of a step -- is
    accumulated in 64-bit fixed point with integer atomics.  Two six-step runs from the same state are then bit-identical in EVERY
    parameter, and so is a run that is checkpointed after three steps and resumed in fresh objects (the statistical bound of the test
    above is only needed in the default mode)."""
    from dnerf_amd.network import NeRFNetwork
    from dnerf_amd.train_native import NativeTrainStep

    def run(n, stop_at=None):
        sc, model, opt, scaler, target = _setup()
        step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=True, seed=5, deterministic=True)
        saved, losses = None, []
        for k in range(n):
            losses.append(float(step(sc.rays_o, sc.rays_d, target, sc.time)))
            if k + 1 == stop_at:
                torch.cuda.synchronize()
                step.sync_optimizer_state()
                saved = {"model": copy.deepcopy(model.state_dict()), "opt": copy.deepcopy(opt.state_dict()), "scaler": scaler.state_dict(),
                         "mean_count": model.mean_count, "local_step": model.local_step, "step_count": step.step_count}
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in model.named_parameters()}, saved, (sc, target), losses

    A, saved, (sc, target), la = run(6, stop_at=3)
    B, _, _, lb = run(6)
    assert la == lb and all(torch.equal(A[k], B[k]) for k in A), [k for k in A if not torch.equal(A[k], B[k])]
    model2 = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    model2.load_state_dict(saved["model"])
    model2.mean_count, model2.local_step = saved["mean_count"], saved["local_step"]
    opt2 = torch.optim.Adam(model2.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
    scaler2 = torch.amp.GradScaler("cuda")
    step2 = NativeTrainStep(model2, opt2, scaler2, N_RAYS, "cuda", perturb=True, seed=5, deterministic=True)
    opt2.load_state_dict(saved["opt"])
    scaler2.load_state_dict(saved["scaler"])
    step2.refresh(optimizer_state=True)
    step2.step_count = saved["step_count"]               # (the per-step noise stream continues where the first run stopped)
    l2 = [float(step2(sc.rays_o, sc.rays_d, target, sc.time)) for _ in range(3)]
    torch.cuda.synchronize()
    assert l2 == la[3:], (l2, la[3:])
    diff = [k for k, v in model2.named_parameters() if not torch.equal(v.detach(), A[k])]
    assert not diff, diff
    # ... and the mode computes the same gradient as the default path up to the half atomics' rounding: one step from the same state
    sc, model, opt, scaler, target = _setup()
    g = []
    for det in (False, True):
        st = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False, deterministic=det)
        st(sc.rays_o, sc.rays_d, target, sc.time, grads_only=True)
        rows = model.encoder.embeddings.shape[0]
        g.append(st.view("g_table", torch.float16, (rows, 2)).float().clone())
    assert float((g[0] - g[1]).norm()) <= 2e-3 * float(g[0].norm()) and float(g[0].norm()) > 0


def test_frozen_deformation_leaves_the_optimizer_serialisable():
    """SealD-NeRF's edit training (SealDNeRF/utils.py:692-694): the optimizer holds only the non-deformation parameters.  The native
    step must not plant state entries for parameters outside the optimizer's groups (torch then raises KeyError in
    `state_dict()`), and a save / load / refresh(optimizer_state=True) round trip must work and carry the step count."""
    from dnerf_amd.network import NeRFNetwork
    from dnerf_amd.seald_train import freeze_deformation
    from dnerf_amd.train_native import NativeTrainStep
    sc, model, _, scaler, target = _setup()
    opt = torch.optim.Adam(freeze_deformation(model), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False, train_deform=False)
    deform_before = [p.detach().clone() for p in model.deform_net.parameters()]
    sigma_before = model.sigma_net[0].weight.detach().clone()
    for _ in range(2):
        step(sc.rays_o, sc.rays_d, target, sc.time)
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.detach()) for a, p in zip(deform_before, model.deform_net.parameters()))
    assert not torch.equal(sigma_before, model.sigma_net[0].weight.detach())
    assert all(p not in opt.state for p in model.deform_net.parameters())
    step.sync_optimizer_state()
    saved = {"model": copy.deepcopy(model.state_dict()), "opt": copy.deepcopy(opt.state_dict()), "scaler": scaler.state_dict()}   # raised KeyError before
    model2 = NeRFNetwork(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    model2.load_state_dict(saved["model"])
    model2.mean_count, model2.local_step = model.mean_count, model.local_step
    opt2 = torch.optim.Adam(freeze_deformation(model2), lr=1e-3, betas=(0.9, 0.99), eps=1e-15)
    scaler2 = torch.amp.GradScaler("cuda")
    step2 = NativeTrainStep(model2, opt2, scaler2, N_RAYS, "cuda", perturb=False, train_deform=False)
    opt2.load_state_dict(saved["opt"])
    scaler2.load_state_dict(saved["scaler"])
    step2.refresh(optimizer_state=True)
    assert step2.adam_steps.tolist() == [2.0, 0.0]
    for p, q in zip(step.params, step2.params):
        if p in opt.state:
            assert torch.equal(opt.state[p]["exp_avg"], opt2.state[q]["exp_avg"]) and torch.equal(opt.state[p]["exp_avg_sq"], opt2.state[q]["exp_avg_sq"])
    l2 = step2(sc.rays_o, sc.rays_d, target, sc.time)
    torch.cuda.synchronize()
    assert np.isfinite(float(l2)) and float(step2.adam_steps[0]) == 3
    # a trained parameter missing from the optimizer is an error, not a silent skip
    opt3 = torch.optim.Adam([model.encoder.embeddings], lr=1e-3)
    with pytest.raises(ValueError):
        NativeTrainStep(model, opt3, scaler, N_RAYS, "cuda", perturb=False, train_deform=False)


def test_skip_grids_follow_in_place_rewrites_of_the_occupancy():
    """The step caches the marcher's coarse skip grid per time slice.  `load_state_dict`, `fill_bitfield` and `reset_extra_state`
    rewrite `density_bitfield` IN PLACE without a new `iter_density`: the cache must notice (tensor version), and `refresh()` drops
    it.  An emptied occupancy yields no samples; restored in place, the next step samples exactly what a fresh step object does."""
    from dnerf_amd.train_native import NativeTrainStep
    sc, model, opt, scaler, target = _setup()
    keep = model.density_bitfield.clone()
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False)
    step(sc.rays_o, sc.rays_d, target, sc.time, grads_only=True)
    full = int(model.step_counter[(model.local_step - 1) % 16, 0])
    assert full > 0
    model.density_bitfield.zero_()
    step(sc.rays_o, sc.rays_d, target, sc.time, grads_only=True)
    assert int(model.step_counter[(model.local_step - 1) % 16, 0]) == 0
    model.density_bitfield.copy_(keep)                   # in place: iter_density unchanged
    step(sc.rays_o, sc.rays_d, target, sc.time, grads_only=True)
    assert int(model.step_counter[(model.local_step - 1) % 16, 0]) == full
    # refresh() is the documented call after an outside change: it forgets the cached grids as well
    step._cull_cache[next(iter(step._cull_cache))].zero_()          # poison the cached grid behind the cache's back
    step.refresh()
    step(sc.rays_o, sc.rays_d, target, sc.time, grads_only=True)
    assert int(model.step_counter[(model.local_step - 1) % 16, 0]) == full


def test_table_pass_on_a_second_stream_changes_nothing_but_the_schedule():
    """`overlap_table_update`: the optimizer's pass over the embedding table runs on a second stream beside the next step's
    deformation-MLP forward.  Same arithmetic on the same operands, so five steps give the parameters of the plain sequence up to
    the table atomics' summation order (the bars of `test_marching_the_next_batch_ahead_changes_nothing`); `flush()` orders the
    caller's stream behind the pass: a stream-ordered read of the table right after it equals the read after a device-wide
    synchronisation; `refresh()` and `sync_optimizer_state()` flush by themselves."""
    from dnerf_amd.train_native import NativeTrainStep
    runs = []
    for overlap in (False, True):
        sc, model, opt, scaler, target = _setup()
        step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=True, seed=3, overlap_table_update=overlap)
        losses = [step(sc.rays_o, sc.rays_d, target, t).clone() for t in (0.5, 0.25, 0.0, 0.75, 0.5)]
        if overlap:                                                  # state_dict() orders the reading stream behind the pass by itself (pre-hook)
            snap = {k: v.clone() for k, v in model.state_dict().items() if k == "encoder.embeddings"}
            torch.cuda.synchronize()
            assert torch.equal(snap["encoder.embeddings"], model.encoder.embeddings.detach())
        step.flush()
        early = model.encoder.embeddings.detach().clone()          # stream-ordered read, no device-wide synchronisation before it
        torch.cuda.synchronize()
        assert torch.equal(early, model.encoder.embeddings.detach())
        rows = model.encoder.embeddings.shape[0]
        assert float(step.view("g_table", torch.float16, (rows, 2)).abs().max()) == 0      # cleared by the table pass
        assert torch.equal(step.view("w_table", torch.float16, (rows, 2)), model.encoder.embeddings.detach().half())
        step.sync_optimizer_state()
        assert float(opt.state[model.encoder.embeddings]["step"]) == 5
        runs.append(([float(x) for x in losses], {k: v.detach().clone() for k, v in model.named_parameters()}))
    assert runs[0][0][0] == runs[1][0][0]
    np.testing.assert_allclose(runs[0][0], runs[1][0], rtol=2e-3)
    for k in runs[0][1]:
        d = (runs[0][1][k] - runs[1][1][k]).abs()
        assert float(d.max()) <= (1.1e-1 if k == "encoder.embeddings" else 1.1e-2), k


def test_a_batch_without_a_single_sample_is_a_finite_step():
    """Empty occupancy (a freshly reset grid, or a batch of rays that all miss): the native step and the autograd step see the same loss --
    every pixel is the background -- produce zero gradients, and the step leaves every parameter finite."""
    from dnerf_amd.train_native import NativeTrainStep
    sc, model, opt, scaler, target = _setup()
    with torch.no_grad():
        model.density_bitfield.zero_()
    out, loss = _eager_backward(model, sc, target, scaler)
    assert int(model.step_counter[0, 0]) == 0                       # no sample anywhere
    np.testing.assert_allclose(float(loss.detach()), float(((1 - target) ** 2).mean()), rtol=1e-5)
    model.local_step = 0
    step = NativeTrainStep(model, opt, scaler, N_RAYS, "cuda", perturb=False)
    got = step(sc.rays_o, sc.rays_d, target, sc.time, grads_only=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(got), float(loss.detach()), rtol=1e-5)
    assert torch.equal(step.image, torch.ones_like(step.image))
    grads = _native_grads(step, model)
    assert all(not g.any() for g in grads.values())
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    got2 = step(sc.rays_o, sc.rays_d, target, sc.time)                 # the whole step: Adam on zero gradients
    step.flush() if hasattr(step, "flush") else None
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(got2), float(loss.detach()), rtol=1e-5)
    for k, v in model.named_parameters():
        assert bool(torch.isfinite(v).all()), k
        assert float((v.detach() - before[k]).abs().max()) <= 1e-6, k   # zero gradient, zero moments: nothing moves
