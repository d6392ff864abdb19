"""GPU parity of the fused MLP operator (csrc/ffmlp.hip through the reference-shaped `ffmlp` package and the C ABI)
against oracle/ffmlp.py on the same seeded inputs.

Bar: both sides form exact fp16 products and round once per layer output, the GPU summing in fp32 on the matrix cores and
the oracle in fp64, so a layer output may differ by one fp16 ulp where the two sums straddle a rounding boundary (and the
next layer inherits that, so the bound grows with depth).  Tolerances below: outputs / buffers within 2 (L + 1) fp16 ulps on
99.9% of the elements and 1% norm-wise overall; weight gradients (sums over the batch of fp16 products, fp32 on the GPU) 1% norm-wise.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ffmlp as F  # noqa: E402


def _close(a, b, what, L=2):
    a = a.astype(np.float32); b = b.astype(np.float32)
    scale = float(np.abs(b).max()) + 1e-12
    ulps = 2 * (L + 1)
    bad = np.abs(a - b) > ulps * 2.0 ** -10 * np.maximum(np.abs(b), scale / 64)
    assert bad.mean() <= 1e-3, f"{what}: {bad.mean():.2e} of the elements off by more than {ulps} ulp"
    assert np.linalg.norm(a - b) <= 1e-2 * np.linalg.norm(b) + 1e-6, what


def _case(in_dim, hidden, L, B, seed=0, wscale=1.0):
    rng = np.random.default_rng(seed)
    std = np.sqrt(3 / hidden) * wscale
    w = rng.uniform(-std, std, hidden * (in_dim + hidden * (L - 1) + 16)).astype(np.float16)
    x = rng.standard_normal((B, in_dim)).astype(np.float16)
    g = (rng.standard_normal((B, 16)) / 16).astype(np.float16)
    return w, x, g


DIMS = [(16, 16, 2), (16, 64, 2), (32, 32, 3), (64, 64, 4), (48, 128, 3), (32, 128, 8), (16, 256, 2), (288, 128, 2), (128, 16, 2)]


@pytest.mark.parametrize("dims", DIMS)
def test_ffmlp_forward_inference_and_buffers_match_oracle(dims):
    import ffmlp
    in_dim, hidden, L = dims
    B = 1000                                   # not a multiple of 128 or of the 256-point workgroup
    w, x, _ = _case(in_dim, hidden, L, B)
    out_ref, fwd_ref = F.ffmlp_forward(x, w, in_dim, hidden, L, 0)
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    out_inf = ffmlp.ffmlp_forward(xd, wd, in_dim, 16, hidden, L, 0, 6, True, False)
    _close(out_inf.cpu().numpy(), out_ref, "inference outputs", L)
    # training forward: same outputs bit for bit, and the saved post-activations
    wd2 = wd.clone().requires_grad_(True)
    out_tr = ffmlp.ffmlp_forward(xd, wd2, in_dim, 16, hidden, L, 0, 6, False, False)
    assert torch.equal(out_tr, out_inf)
    fwd = out_tr.grad_fn.saved_tensors[3]
    assert fwd.shape == (L, B, hidden)
    _close(fwd.cpu().numpy(), fwd_ref, "forward_buffer", L)


@pytest.mark.parametrize("act", ["relu", "exponential", "sine", "sigmoid", "squareplus", "softplus", "none"])
def test_ffmlp_activations_match_oracle(act):
    import ffmlp
    in_dim, hidden, L, B = 32, 64, 3, 640
    w, x, _ = _case(in_dim, hidden, L, B, seed=3, wscale=0.5)
    a = ffmlp.convert_activation(act)
    out_ref, _ = F.ffmlp_forward(x, w, in_dim, hidden, L, a)
    out = ffmlp.ffmlp_forward(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), in_dim, 16, hidden, L, a, 6, True, False)
    _close(out.cpu().numpy(), out_ref, act, L)


@pytest.mark.parametrize("act", ["relu", "exponential", "sigmoid", "squareplus", "softplus", "none"])
@pytest.mark.parametrize("dims", [(16, 64, 2), (32, 32, 3), (48, 128, 3), (16, 256, 2), (288, 128, 2), (32, 16, 2)])
def test_ffmlp_backward_matches_oracle(dims, act):
    import ffmlp
    in_dim, hidden, L = dims
    B = 3000
    a = ffmlp.convert_activation(act)
    w, x, g = _case(in_dim, hidden, L, B, seed=7, wscale=0.5 if act in ("exponential", "softplus") else 1.0)
    out_ref, fwd_ref = F.ffmlp_forward(x, w, in_dim, hidden, L, a)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    wd = torch.from_numpy(w).cuda().requires_grad_(True)
    out = ffmlp.ffmlp_forward(xd, wd, in_dim, 16, hidden, L, a, 6, False, True)
    # the oracle's backward runs on the GPU's own forward_buffer, so that ReLU gates are the same on both sides
    fwd = out.grad_fn.saved_tensors[3].cpu().numpy()
    out.backward(torch.from_numpy(g).cuda())
    _close(fwd, fwd_ref, "forward_buffer", L)
    gi_ref, gw_ref, _ = F.ffmlp_backward(g, x, w, fwd, in_dim, hidden, L, a)
    assert xd.grad.dtype == torch.float16 and wd.grad.dtype == torch.float16
    _close(xd.grad.cpu().numpy(), gi_ref, "grad_inputs", L)
    gw = wd.grad.cpu().numpy().astype(np.float32)
    gw_ref = gw_ref.astype(np.float32)
    assert np.linalg.norm(gw - gw_ref) <= 1e-2 * np.linalg.norm(gw_ref), "grad_weights"
    # per layer too, so that a wrong small block cannot hide behind a large one
    off = 0
    for r, c in [(hidden, in_dim)] + [(hidden, hidden)] * (L - 1) + [(16, hidden)]:
        blk, ref = gw[off:off + r * c], gw_ref[off:off + r * c]
        assert np.linalg.norm(blk - ref) <= 1e-2 * np.linalg.norm(ref) + 1e-6, f"grad_weights block at {off}"
        off += r * c


def test_ffmlp_module_trains_like_a_linear_stack():
    """testing/test_ffmlp.py's comparison: FFMLP under autocast vs the same weights in bias-free nn.Linear layers."""
    import ffmlp
    in_dim, out_dim, hidden, L, B = 16, 3, 64, 2, 5000
    net = ffmlp.FFMLP(in_dim, out_dim, hidden, L).cuda()
    mats = [m.clone().float().cuda().requires_grad_(True) for m in
            map(torch.from_numpy, F.split_weights(net.weights.detach().cpu().numpy(), in_dim, hidden, L))]
    x = torch.rand(B, in_dim, device="cuda") * 2 - 1
    x0 = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        y0 = net(x0)
    assert y0.shape == (B, out_dim) and y0.dtype == torch.float16
    h = x
    for m in mats[:-1]:
        h = torch.relu(h @ m.T)
    y1 = (h @ mats[-1].T)[:, :out_dim]
    assert torch.allclose(y0.detach().float(), y1.detach(), rtol=2e-2, atol=2e-2 * float(y1.detach().abs().max()))
    (y0.float() ** 2).mean().backward()
    (y1 ** 2).mean().backward()
    ref = torch.cat([m.grad.reshape(-1) for m in mats])
    got = net.weights.grad
    assert got.dtype == torch.float32 and got.shape == ref.shape
    # loss-scaled fp16 gradients are tiny here (1/B); compare directions and norms
    assert float((got - ref).norm()) <= 5e-2 * float(ref.norm())
    assert x0.grad is not None and x0.grad.shape == (B, in_dim)
    # eval mode goes through the inference entry point and agrees with training mode
    net.eval()
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        y2 = net(x)
    assert torch.equal(y2, y0.detach())


def test_ffmlp_rejects_what_the_reference_rejects():
    import ffmlp
    import sdn_backend
    x = torch.zeros(128, 16, device="cuda", dtype=torch.float16)
    w = torch.zeros(48 * (16 + 48 + 16), device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError):
        ffmlp.ffmlp_forward(x, w, 16, 16, 48, 2, 0, 6, True, False)          # hidden 48: "hidden_dim should in [...]" (ffmlp.cu:657)
    w = torch.zeros(64 * (16 + 64 + 16), device="cuda", dtype=torch.float16)
    with pytest.raises(sdn_backend.SdnError):
        ffmlp.ffmlp_forward(x.float(), w, 16, 16, 64, 2, 0, 6, True, False)  # CHECK_IS_HALF (ffmlp.cu:637)
    wd = w.clone().requires_grad_(True)
    y = ffmlp.ffmlp_forward(x, wd, 16, 16, 64, 2, 2, 6, False, False)       # sine forward is fine ...
    with pytest.raises(NotImplementedError):
        y.sum().backward()                                                   # ... its backward is not (utils.h:552-556)


@pytest.mark.parametrize("B", [1, 31, 33, 255, 257])
def test_ffmlp_ragged_batches_match_the_padded_result(B):
    """Any batch size works (the reference pads to 128, ffmlp.py:154-157): rows beyond B are neither read nor written."""
    import ffmlp
    in_dim, hidden, L = 32, 64, 3
    w, x, g = _case(in_dim, hidden, L, 512, seed=11)
    wd = torch.from_numpy(w).cuda()
    full = ffmlp.ffmlp_forward(torch.from_numpy(x).cuda(), wd, in_dim, 16, hidden, L, 0, 6, True, False)
    guard = torch.full((B + 64, 16), -7.0, dtype=torch.float16, device="cuda")
    xs = torch.from_numpy(x[:B]).cuda().contiguous()
    out = ffmlp.ffmlp_forward(xs, wd, in_dim, 16, hidden, L, 0, 6, True, False)
    assert out.shape == (B, 16) and torch.equal(out, full[:B])
    # training forward + backward on the ragged batch, weights only (calc_grad_inputs False)
    wg = wd.clone().requires_grad_(True)
    y = ffmlp.ffmlp_forward(xs, wg, in_dim, 16, hidden, L, 0, 6, False, False)
    assert torch.equal(y, full[:B])
    fwd = y.grad_fn.saved_tensors[3].cpu().numpy()
    y.backward(torch.from_numpy(g[:B]).cuda())
    _, gw_ref, _ = F.ffmlp_backward(g[:B], x[:B], w, fwd, in_dim, hidden, L, 0, calc_grad_inputs=False)
    gw, gw_ref = wg.grad.cpu().numpy().astype(np.float32), gw_ref.astype(np.float32)
    assert np.linalg.norm(gw - gw_ref) <= 1e-2 * np.linalg.norm(gw_ref) + 1e-6
    del guard


def test_dnerf_network_on_the_fused_operator_matches_the_linear_stack():
    """dnerf_amd/network_ff.py: same parameters, deformation and colour MLPs through ffmlp_forward; forward values and every
    parameter gradient against the autocast nn.Linear network (dnerf/network.py:123-169) on the same points."""
    from dnerf_amd.bench_scene import build_model
    from dnerf_amd.network_ff import NeRFNetworkFF
    ref = build_model(0, "cuda").train()
    ff = NeRFNetworkFF(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    ff.load_state_dict(ref.state_dict())
    g = torch.Generator(device="cuda").manual_seed(0)
    M = 9000
    x = (torch.rand(M, 3, device="cuda", generator=g) - 0.5)
    d = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda", generator=g), dim=1)
    t = torch.tensor([[0.5]], device="cuda")
    w_s, w_c, w_d = torch.rand(M, device="cuda", generator=g), torch.rand(M, 3, device="cuda", generator=g), torch.rand(M, 3, device="cuda", generator=g)
    outs, grads = [], []
    for net in (ref, ff):
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb, deform = net(x, d, t)
            # sigma is exp(logit) with logits up to ~6 here: weigh it down so that the three terms pull comparably
            loss = (torch.log1p(sigma.float()) * w_s).sum() + (rgb.float() * w_c).sum() + (deform.float() * w_d).sum() * 10
        (loss * 64.0).backward()
        outs.append((sigma.detach().float(), rgb.detach().float(), deform.detach().float()))
        grads.append({n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None})
    (s0, c0, d0), (s1, c1, d1) = outs
    assert d1.dtype == torch.float32 and float((d1 - d0).abs().max()) <= 2e-3 * float(d0.abs().max()) + 1e-4
    rel = (s1 - s0).abs() / s0.abs().clamp(min=1e-3)
    assert float(rel.median()) < 2e-3 and float(rel.max()) < 5e-2
    assert float((c1 - c0).abs().max()) < 1e-2
    assert set(grads[0]) == set(grads[1]) and any(k.startswith("deform_net") for k in grads[0]) and "encoder.embeddings" in grads[0]
    for name, g0 in grads[0].items():
        g1 = grads[1][name]
        assert float((g1 - g0).norm()) <= 3e-2 * float(g0.norm()) + 1e-6, (name, float((g1 - g0).norm()), float(g0.norm()))
    # inference mode and the fp32 fallback
    ff.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s2, c2, _ = ff(x, d, t)
    assert torch.equal(s2.float(), s1) and torch.equal(c2.float(), c1)
    with torch.no_grad():
        s3, _, _ = ff(x, d, t)
        s4, _, _ = ref.eval()(x, d, t)
    assert torch.equal(s3, s4)


def test_graphed_training_step_trains_like_the_eager_step():
    """dnerf_amd/train_graph.py: the step replayed from one HIP graph against the same step launched from Python (same fused Adam,
    same GradScaler, perturb off so both see the same samples): losses and parameters stay together over several steps; the only
    difference is the order of the fp16 atomics in the grid backward."""
    from dnerf_amd.bench_scene import build_scene
    from dnerf_amd.network_ff import NeRFNetworkFF
    from dnerf_amd.train_graph import GraphedTrainStep
    sc = build_scene(H=64, W=64, device="cuda", seed=0)
    n_rays = 2048
    idx = torch.randint(0, sc.rays_o.shape[0], (n_rays,), generator=torch.Generator().manual_seed(0)).cuda()
    rays_o, rays_d = sc.rays_o[idx][None].contiguous(), sc.rays_d[idx][None].contiguous()
    target = torch.rand(1, n_rays, 3, generator=torch.Generator().manual_seed(2)).cuda()
    losses, finals = [], []
    for graphed in (False, True):
        m = NeRFNetworkFF(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
        m.load_state_dict(sc.model.state_dict())
        opt = torch.optim.Adam(m.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
        scaler = torch.amp.GradScaler("cuda")
        # point budget from one eager pass, as the reference's first epoch provides it
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            m.render(rays_o, rays_d, sc.time, staged=False, perturb=False, bg_color=1, force_all_rays=False, max_steps=1024)
        m.mean_count = int(m.step_counter[0, 0].item())
        assert m.mean_count > 100
        m.local_step = 0
        gs = GraphedTrainStep(m, opt, scaler, n_rays, "cuda", warmup=0, perturb=False)
        gs.load(rays_o, rays_d, target, sc.time)
        run = []
        if graphed:
            gs.capture()
            assert m.local_step == 0
            for k in range(6):
                run.append(float(gs()))
            assert m.local_step == 6 and int(m.step_counter[5, 0]) == int(m.step_counter[0, 0]) > 0
        else:
            for k in range(6):
                opt.zero_grad(set_to_none=True)
                run.append(float(gs._step().detach()))
        losses.append(run)
        finals.append({n: p.detach().float().clone() for n, p in m.named_parameters()})
    a, b = losses
    assert a[-1] < a[0] and b[-1] < b[0]                       # it trains
    assert all(abs(x - y) <= 2e-2 * abs(x) for x, y in zip(a, b)), (a, b)
    # Parameters: Adam with eps = 1e-15 moves every element by ~lr per step in the direction of sign(grad), so elements whose gradient
    # is at the noise level of the atomics' summation order walk apart by up to 2 lr per step; bound, not tolerance.
    for name, p in finals[0].items():
        q = finals[1][name]
        lr = 1e-2 if name.startswith("encoder") else 1e-3
        assert float((p - q).abs().max()) <= 2.5 * 6 * lr, name


def _dp_rank(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # one-GPU rehearsal: both ranks on cuda:0, gradients staged through the host
    try:
        import tests_support  # noqa: F401
        from dnerf_amd.bench_scene import build_scene
        from dnerf_amd.network_ff import NeRFNetworkFF
        from dnerf_amd.train_graph import GraphedTrainStep
        from dnerf_amd.dist import GradSync
        sc = build_scene(H=64, W=64, device="cuda", seed=0)
        n_rays = 1024
        idx = torch.randint(0, sc.rays_o.shape[0], (n_rays,), generator=torch.Generator().manual_seed(rank)).cuda()   # own batch per rank
        rays_o, rays_d = sc.rays_o[idx][None].contiguous(), sc.rays_d[idx][None].contiguous()
        target = torch.rand(1, n_rays, 3, generator=torch.Generator().manual_seed(50 + rank)).cuda()
        m = NeRFNetworkFF(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
        m.load_state_dict(sc.model.state_dict())
        with torch.no_grad():
            m.deform_net[0].weight.mul_(1 + 0.01 * rank)            # replicas start different; the broadcast makes them equal
        sync = GradSync(m)
        sync.broadcast_parameters()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            m.render(rays_o, rays_d, sc.time, staged=False, perturb=False, bg_color=1, force_all_rays=False, max_steps=1024)
        m.mean_count = int(m.step_counter[0, 0].item()) + 256       # headroom: the ranks' batches differ
        m.local_step = 0
        opt = torch.optim.Adam(m.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
        step = GraphedTrainStep(m, opt, torch.amp.GradScaler("cuda"), n_rays, "cuda", warmup=1, grad_sync=sync, perturb=False)
        step.load(rays_o, rays_d, target, sc.time)
        losses = [float(step()) for _ in range(5)]
        flat = torch.cat([p.detach().float().reshape(-1) for p in m.parameters()]).cpu()
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat)
        same = all(torch.equal(parts[0], q) for q in parts)
        np.save(os.path.join(out_dir, f"dp_{rank}.npy"), np.array([float(same), losses[0], losses[-1], float(step.graph_opt is not None)]))
    finally:
        dist.destroy_process_group()


def test_data_parallel_graphed_step_keeps_replicas_identical(tmp_path):
    """Two ranks (both on this one GPU, gloo with host staging: mechanics only) train on different ray batches through the two-graph
    step with the gradient all-reduces in between: parameters stay bit-identical across ranks and the loss goes down."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        same, first, last, two_graphs = np.load(tmp_path / f"dp_{r}.npy")
        assert same == 1.0 and two_graphs == 1.0 and last < first, (same, first, last)


def test_seald_edit_training_step_learns_the_teachers_edit():
    """dnerf_amd/seald_train.py (StudentTrainer.train_gui, SealDNeRF/utils.py:667-777): the teacher's mapped render is the target of the
    student's graphed step; the deformation network stays frozen; the student moves towards the edited image."""
    from dnerf_amd.bench_scene import build_scene
    from dnerf_amd.network_ff import NeRFNetworkFF
    from dnerf_amd.renderer import render_frame
    from dnerf_amd.seald_train import EditTrainStep, freeze_deformation
    from dnerf_amd import fused, seal_mapper as SM
    sc = build_scene(H=64, W=64, device="cuda", seed=0)
    half, centre = 0.12, (0.0, 0.47, 0.0)
    raw = [[centre[0] + sx * half, centre[1] + sy * half, centre[2] + sz * half] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)]
    T = np.eye(4); T[0, 3] = 0.35
    mapper = SM.get_seal_mapper({"type": "bbox", "raw": raw, "transform": T.tolist(), "scale": [1.0, 1.0, 1.0], "boundType": "to", "hsv": [0.3, 0.0, 0.0]})
    SM.fill_bitfield(sc.model.density_bitfield, mapper.map_data["force_fill_bound"].cpu().numpy(), sc.model.grid_size, sc.model.bound)
    n_rays = sc.rays_o.shape[0]
    student = NeRFNetworkFF(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    student.load_state_dict(sc.model.state_dict())
    params = freeze_deformation(student)
    assert all(not p.requires_grad for p in student.deform_net.parameters()) and len(params) == 6   # table + 2 sigma + 3 colour matrices
    frozen = [p.detach().clone() for p in student.deform_net.parameters()]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        student.render(sc.rays_o[None], sc.rays_d[None], sc.time, staged=False, perturb=False, bg_color=1, force_all_rays=False, max_steps=1024)
    student.mean_count = int(student.step_counter[0, 0].item()) + 512
    student.local_step = 0
    opt = torch.optim.Adam(params, lr=2e-3, betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
    edit = EditTrainStep(sc.model, student, mapper, opt, torch.amp.GradScaler("cuda"), n_rays, "cuda", sc.time, perturb=False, warmup=1)
    # the target is exactly the native mapped render of the teacher
    want = render_frame(sc.model, sc.rays_o, sc.rays_d, sc.time, fp16=True, field=fused.FusedField(sc.model, sc.time, fp16=True), T_thresh=1e-4, mapper=mapper)["image"]
    got = edit.proxy_truth(sc.rays_o, sc.rays_d, sc.time)
    assert torch.equal(got, want)
    plain = render_frame(sc.model, sc.rays_o, sc.rays_d, sc.time, fp16=True, field=fused.FusedField(sc.model, sc.time, fp16=True), T_thresh=1e-4)["image"]
    assert float((want - plain).abs().max()) > 0.05                               # the edit is visible
    losses = [float(edit(sc.rays_o, sc.rays_d, sc.time)) for _ in range(30)]
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])
    assert all(torch.equal(a, b.detach()) for a, b in zip(frozen, student.deform_net.parameters()))
    # pipelined epoch (teacher of batch k+1 under the student's step k): same kind of progress, same frozen deformation
    before = float(edit.step.loss)
    batch = (sc.rays_o, sc.rays_d, sc.time)
    assert edit.run([batch] * 20) == 20
    torch.cuda.synchronize()
    assert float(edit.step.loss) < before
    assert all(torch.equal(a, b.detach()) for a, b in zip(frozen, student.deform_net.parameters()))
    assert edit.run([]) == 0
