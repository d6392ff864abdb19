"""The frame loops on scenes at the edges of what the marcher sees: an EMPTY occupancy grid (every ray retires at the cull test: no
iteration has a live ray), a FULL one (no cull cell is unmarked, the fine image of the marked box does not fit the marchers' LDS copy,
every ray marches from the cube's face to its T_thresh), ray counts that fill no wave, and a camera INSIDE the volume.  The three
loops -- the reference-shaped `model.render` (dnerf/renderer.py:350-376 over the drop-in operators), the host-stepped `render_frame`
and the device-driven `DeviceLoop` with the fused field -- must agree bit for bit (the same operators, the same fused kernel, per-ray
results that do not depend on the schedule); the full grid's samples are also checked against the CPU oracle's marcher."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(H=32, W=32):
    from dnerf_amd.bench_scene import build_scene
    return build_scene(H=H, W=W, device="cuda", seed=0)


def _three_loops(sc, rays_o, rays_d):
    from dnerf_amd import fused
    from dnerf_amd.renderer import DeviceLoop, render_frame
    model = sc.model.eval()
    f = fused.FusedField(model, sc.time, fp16=True)
    host = render_frame(model, rays_o, rays_d, sc.time, fp16=True, field=f)
    loop = DeviceLoop(model, f, rays_o.shape[0], rays_o.device)
    dev = loop.render(rays_o, rays_d, sc.time)
    assert torch.equal(host["image"], dev["image"]) and torch.equal(host["weights_sum"], dev["weights_sum"])
    assert torch.equal(torch.nan_to_num(host["depth"]), torch.nan_to_num(dev["depth"]))
    assert [tuple(t) for t in host["trace"]] == [tuple(t) for t in dev["trace"]] and host["n_samples"] == dev["n_samples"]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        ref = model.render(rays_o[None], rays_d[None], sc.time, staged=False, perturb=False, bg_color=1)
    assert torch.equal(ref["image"][0], host["image"])        # the same operators, the same kernel, the same schedule
    return host


def _set_bits(model, value):
    with torch.no_grad():
        model.density_bitfield.fill_(value)       # (in place: the cached cull grids follow the tensor's version counter)


def test_empty_occupancy_every_ray_is_background():
    sc = _scene()
    saved = sc.model.density_bitfield.clone()
    try:
        _set_bits(sc.model, 0)
        out = _three_loops(sc, sc.rays_o, sc.rays_d)
        assert out["n_samples"] == 0 and not out["weights_sum"].any()
        assert torch.equal(out["image"], torch.ones_like(out["image"]))
    finally:
        with torch.no_grad():
            sc.model.density_bitfield.copy_(saved)


def test_full_occupancy_every_ray_marches_from_the_cube_face():
    sc = _scene()
    saved = sc.model.density_bitfield.clone()
    try:
        _set_bits(sc.model, 255)
        out = _three_loops(sc, sc.rays_o, sc.rays_d)
        hit = out["weights_sum"] > 0
        assert out["n_samples"] > sc.rays_o.shape[0] and float(hit.float().mean()) > 0.5
        # the samples against the CPU oracle's marcher on a handful of rays: same count per ray (bit-exact indices / counts)
        import raymarching
        m = sc.model
        sel = torch.arange(0, sc.rays_o.shape[0], 97, device="cuda")
        ro, rd = sc.rays_o[sel].contiguous(), sc.rays_d[sel].contiguous()
        nears, fars = raymarching.near_far_from_aabb(ro, rd, m.aabb_infer, m.min_near)
        alive = torch.arange(sel.shape[0], dtype=torch.int32, device="cuda")
        t_idx = int(min(max(np.floor(float(sc.time) * m.time_size), 0), m.time_size - 1))
        x, d, dl = raymarching.march_rays(sel.shape[0], 8, alive, nears.clone(), ro, rd, m.bound, m.density_bitfield[t_idx], m.cascade,
                                          m.grid_size, nears, fars, 128, False, 0.0, 1024)
        from tests_support import O
        xo, do, dlo = O.march_rays(sel.shape[0], 8, alive.cpu().numpy(), nears.cpu().numpy().copy(), ro.cpu().numpy(), rd.cpu().numpy(),
                                   float(m.bound), m.density_bitfield[t_idx].cpu().numpy(), int(m.cascade), int(m.grid_size),
                                   nears.cpu().numpy(), fars.cpu().numpy(), align=128)
        for got, want in ((x, xo), (d, do), (dl, dlo)):
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    finally:
        with torch.no_grad():
            sc.model.density_bitfield.copy_(saved)


@pytest.mark.parametrize("n", [1, 7, 65, 255])
def test_ray_counts_that_fill_no_wave_or_workgroup(n):
    sc = _scene()
    centre = sc.rays_o.shape[0] // 2 + 16            # rays through the middle of the image: they hit the object
    idx = torch.arange(centre - n // 2, centre - n // 2 + n, device="cuda")
    out = _three_loops(sc, sc.rays_o[idx].contiguous(), sc.rays_d[idx].contiguous())
    assert out["image"].shape == (n, 3)
    whole = _three_loops(sc, sc.rays_o, sc.rays_d)
    assert torch.equal(out["image"], whole["image"][idx])          # a ray's pixel does not depend on its batch


def test_camera_inside_the_volume():
    sc = _scene()
    ro = torch.zeros_like(sc.rays_o)                                # every ray starts at the origin, inside the object's box
    ro[:, 0] = 0.05
    out = _three_loops(sc, ro.contiguous(), sc.rays_d)
    assert bool(torch.isfinite(out["image"]).all()) and out["n_samples"] > 0
