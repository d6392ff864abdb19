"""Per-frame time and frame groups in the native drivers.

A D-NeRF test set carries its own time stamp for every frame (dnerf/utils.py:151-161): the occupancy slice
(dnerf/renderer.py:285), the time encoding of the deformation network and the t == 0 canonical rule (dnerf/network.py:130-141)
all follow it.  The native loops derive all three from the VALUE of the time stamp (never from a tensor's address), per frame
in a stream of frames and per ray in a frame group."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from caller_fixtures import fixture_model, fixture_scene, load  # noqa: E402


@pytest.fixture(scope="module")
def model_bits():
    return fixture_model("cuda")


def _rays(H, W, az, el=30.0):
    from dnerf_amd import scene
    ro, rd = scene.get_rays(scene.look_at_pose(az, el), scene.intrinsics(H, W), H, W)
    return torch.from_numpy(ro).cuda(), torch.from_numpy(rd).cuda()


def _eq(a, b):
    return torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))


def test_one_loop_object_follows_the_time_value_with_fresh_tensors(model_bits):
    """One FusedField + DeviceLoop (built at t = 0.5) renders t = 0.0, 0.26, 0.5 from freshly allocated time tensors -- the caching
    allocator may hand each of them the address of its predecessor -- and every frame matches the reference's run_cuda fixture of
    THAT time (fp16 distance), and the frame a loop built at that time renders (bit for bit)."""
    from dnerf_amd.fused import FusedField
    from dnerf_amd.renderer import DeviceLoop
    fx = load("infer")
    model, _ = model_bits
    sc = fixture_scene("cuda", model_bits=model_bits)
    field = FusedField(model, sc.time)
    loop = DeviceLoop(model, field, sc.rays_o.shape[0], "cuda")
    imgs = {}
    for t in (0.0, 0.26, 0.5, 0.0):
        time = torch.tensor([[t]], dtype=torch.float32, device="cuda")      # fresh tensor, likely a recycled address
        out = loop.render(sc.rays_o, sc.rays_d, time)
        torch.cuda.synchronize()
        img = out["image"].clone()
        del time
        ref = fx[f"t{t}_image"]
        assert np.abs(img.cpu().numpy() - ref).max() < 2e-2 and np.abs(img.cpu().numpy() - ref).mean() < 2e-4, t
        if t in imgs:
            assert torch.equal(imgs[t], img)
        imgs[t] = img
        own = DeviceLoop(model, FusedField(model, torch.tensor([[t]], device="cuda")), sc.rays_o.shape[0], "cuda")
        assert torch.equal(own.render(sc.rays_o, sc.rays_d, t)["image"], img)
    assert not torch.equal(imgs[0.0], imgs[0.5]) and not torch.equal(imgs[0.26], imgs[0.5])


def test_pipelined_stream_with_a_time_per_frame(model_bits):
    """5 cameras x 5 distinct times through 3 overlapping loop contexts: every frame is bit-identical to DeviceLoop.render of that
    camera at that time, one by one."""
    from dnerf_amd.fused import FusedField
    from dnerf_amd.renderer import DeviceLoop, PipelinedDeviceLoop
    model, _ = model_bits
    H = W = 64
    cams = [_rays(H, W, az) for az in (30.0, 100.0, 170.0, 240.0, 310.0)]
    times = [0.0, 0.26, 0.5, 0.26, 0.5078125]
    field = FusedField(model, 0.5)
    one = DeviceLoop(model, field, H * W, "cuda")
    want = []
    for (ro, rd), t in zip(cams, times):
        o = one.render(ro, rd, t)
        want.append((o["image"].clone(), o["depth"].clone()))
    pl = PipelinedDeviceLoop(model, field, H * W, "cuda", contexts=3)
    outs = [(torch.empty(H * W, 3, device="cuda"), torch.empty(H * W, device="cuda")) for _ in cams]
    pl.render_frames([c[0] for c in cams], [c[1] for c in cams], times, outputs=outs)
    torch.cuda.synchronize()
    for (img, dep), (wi, wd) in zip(outs, want):
        assert torch.equal(img, wi) and _eq(dep, wd)
    assert not torch.equal(want[0][0], want[2][0])


@pytest.mark.parametrize("H,W,frames", [(64, 64, 3), (50, 50, 4), (24, 40, 8)])
def test_frame_group_equals_frames_rendered_alone(model_bits, H, W, frames):
    """F frames (own camera, own time -- incl. the canonical t = 0) rendered TOGETHER by one loop with per-ray time constants:
    every frame is bit-identical to the frame rendered alone.  50x50 = 2500 rays per frame puts frame boundaries inside
    workgroups (the marcher's per-frame rounds); 24x40 x 8 frames exercises many boundaries per launch."""
    from dnerf_amd.fused import FusedField
    from dnerf_amd.renderer import DeviceLoop, PipelinedDeviceLoop
    model, _ = model_bits
    all_times = [0.5, 0.0, 0.26, 0.5078125, 0.26, 0.0, 0.5, 0.2578125]
    cams = [_rays(H, W, 30.0 + 41.0 * f) for f in range(frames)]
    times = all_times[:frames]
    field = FusedField(model, 0.5)
    one = DeviceLoop(model, field, H * W, "cuda")
    want, n_samples = [], 0
    for (ro, rd), t in zip(cams, times):
        o = one.render(ro, rd, t)
        want.append((o["image"].clone(), o["depth"].clone()))
        n_samples += o["n_samples"]
    grp = DeviceLoop(model, field, frames * H * W, "cuda", frames=frames)
    ro = torch.cat([c[0] for c in cams]).contiguous()
    rd = torch.cat([c[1] for c in cams]).contiguous()
    out = grp.render(ro, rd, times)
    torch.cuda.synchronize()
    n = H * W
    for f in range(frames):
        assert torch.equal(out["image"][f * n:(f + 1) * n], want[f][0]), f
        assert _eq(out["depth"][f * n:(f + 1) * n], want[f][1]), f
    # the group's schedule is n_step = clamp(N_group // n_alive_group, 1, 8): it may march a ray past its termination by a
    # different number of (discarded) samples than the one-frame schedule
    assert abs(out["n_samples"] - n_samples) <= 0.02 * n_samples
    # a second render of the same group object, and a stream of two groups through the pipelined driver
    again = grp.render(ro, rd, times)
    assert torch.equal(again["image"], out["image"])
    pl = PipelinedDeviceLoop(model, field, frames * n, "cuda", contexts=2, frames=frames)
    rev = list(reversed(times))
    outs = [(torch.empty(frames * n, 3, device="cuda"), torch.empty(frames * n, device="cuda")) for _ in range(2)]
    pl.render_frames([ro, ro], [rd, rd], [times, rev], outputs=outs)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], out["image"])
    b = grp.render(ro, rd, rev)
    assert torch.equal(outs[1][0], b["image"])


def test_on_done_hands_finished_frames_on_in_order(model_bits):
    """The hook bench.py --gpus N uses to start a frame's all-gather while later frames still render: called once per frame, in
    frame order, on a helper thread whose current stream is ordered behind that frame's last kernel."""
    from dnerf_amd.fused import FusedField
    from dnerf_amd.renderer import PipelinedDeviceLoop
    model, _ = model_bits
    H = W = 64
    cams = [_rays(H, W, 30.0 + 50.0 * f) for f in range(6)]
    times = [0.0, 0.26, 0.5, 0.26, 0.5, 0.0]
    pl = PipelinedDeviceLoop(model, FusedField(model, 0.5), H * W, "cuda", contexts=3)
    outs = [(torch.empty(H * W, 3, device="cuda"), torch.empty(H * W, device="cuda")) for _ in cams]
    seen, copies = [], []

    def hook(f, img, dep):
        seen.append(f)
        copies.append(img.clone())          # on the helper thread's stream, behind frame f

    pl.render_frames([c[0] for c in cams], [c[1] for c in cams], times, outputs=outs, on_done=hook)
    torch.cuda.synchronize()
    assert seen == list(range(6))
    for f in range(6):
        assert torch.equal(copies[f], outs[f][0])
    assert not torch.equal(copies[0], copies[2])


def test_kept_cull_grids_change_nothing_and_follow_the_bitfield(model_bits):
    """`keep_cull_grids=True`: each occupancy slice's cull grid is derived once and copied into the context afterwards (single
    frames, a stream, a frame group): frames bit-identical to the loops that derive it every time; after the occupancy grid was
    written, `invalidate_cull_grids` makes the next frame use the new one."""
    from dnerf_amd.fused import FusedField
    from dnerf_amd.renderer import DeviceLoop, PipelinedDeviceLoop
    model, _ = model_bits
    H = W = 64
    n = H * W
    cams = [_rays(H, W, az) for az in (30.0, 100.0, 170.0, 240.0)]
    times = [0.5, 0.0, 0.26, 0.5]
    field = FusedField(model, 0.5)
    plain = DeviceLoop(model, field, n, "cuda")
    want = [plain.render(ro, rd, t)["image"].clone() for (ro, rd), t in zip(cams, times)]
    kept = DeviceLoop(model, field, n, "cuda", keep_cull_grids=True)
    for k in range(2):                                      # second round: every grid comes from the cache
        for (ro, rd), t, w in zip(cams, times, want):
            assert torch.equal(kept.render(ro, rd, t)["image"], w)
    pl = PipelinedDeviceLoop(model, field, n, "cuda", contexts=3, keep_cull_grids=True)
    outs = [(torch.empty(n, 3, device="cuda"), torch.empty(n, device="cuda")) for _ in cams]
    pl.render_frames([c[0] for c in cams], [c[1] for c in cams], times, outputs=outs)
    torch.cuda.synchronize()
    assert all(torch.equal(o[0], w) for o, w in zip(outs, want))
    grp = DeviceLoop(model, field, 4 * n, "cuda", frames=4, keep_cull_grids=True)
    out = grp.render(torch.cat([c[0] for c in cams]).contiguous(), torch.cat([c[1] for c in cams]).contiguous(), times)
    assert all(torch.equal(out["image"][f * n:(f + 1) * n], want[f]) for f in range(4))
    # empty the occupancy grid: a kept grid is stale until invalidated
    saved = model.density_bitfield.clone()
    try:
        model.density_bitfield.zero_()
        DeviceLoop.invalidate_cull_grids(model)
        blank = kept.render(cams[0][0], cams[0][1], times[0])
        assert blank["n_samples"] == 0 and torch.equal(blank["image"], torch.ones_like(blank["image"]))
    finally:
        model.density_bitfield.copy_(saved)
        DeviceLoop.invalidate_cull_grids(model)
    assert torch.equal(kept.render(cams[0][0], cams[0][1], times[0])["image"], want[0])


def test_kept_cull_grids_follow_in_place_rewrites_of_the_occupancy(model_bits):
    """`DeviceLoop(keep_cull_grids=True)` keeps the marcher's coarse grid -- which carries the slice's packed occupancy bits -- per time
    slice.  `load_state_dict`, `fill_bitfield` and `reset_extra_state` rewrite `density_bitfield` in place without a new `iter_density`:
    the cache must notice (tensor version).  An emptied occupancy renders the background; restored in place, the frame is again the
    frame of a loop that keeps nothing, bit for bit."""
    from dnerf_amd.fused import FusedField
    from dnerf_amd.renderer import DeviceLoop
    model, _ = model_bits
    sc = fixture_scene("cuda", model_bits=model_bits)
    field = FusedField(model, sc.time)
    kept = DeviceLoop(model, field, sc.rays_o.shape[0], "cuda", keep_cull_grids=True)
    plain = DeviceLoop(model, field, sc.rays_o.shape[0], "cuda")
    want = plain.render(sc.rays_o, sc.rays_d, sc.time)["image"].clone()
    assert torch.equal(kept.render(sc.rays_o, sc.rays_d, sc.time)["image"], want) and float(want.min()) < 0.99
    keep = model.density_bitfield.clone()
    try:
        model.density_bitfield.zero_()
        blank = kept.render(sc.rays_o, sc.rays_d, sc.time)
        assert blank["n_samples"] == 0 and bool((blank["image"] == 1).all())
        model.density_bitfield.copy_(keep)                   # in place: iter_density unchanged
        assert torch.equal(kept.render(sc.rays_o, sc.rays_d, sc.time)["image"], want)
    finally:
        model.density_bitfield.copy_(keep)
