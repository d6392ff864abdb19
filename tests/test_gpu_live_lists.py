"""Live-sample lists behind the reference's caller (dnerf/renderer.py:350-376 calls march_rays, self(xyzs, dirs, time), composite_rays
unchanged): the drop-in `march_rays` hangs the list of slots that received a sample on its xyzs tensor, and the fused dispatch of
`NeRFNetwork.forward` evaluates those slots only.  The operators' results must not depend on it: same three tensors from the
marcher, the same values in every slot that holds a sample, zeros elsewhere, the same rendered frame bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from caller_fixtures import fixture_model, fixture_scene  # noqa: E402


@pytest.fixture(scope="module")
def model_bits():
    return fixture_model("cuda")


@pytest.fixture()
def lists():
    import raymarching
    saved = dict(raymarching.live_lists)
    yield raymarching.live_lists
    raymarching.live_lists.update(saved)


def _march(sc, n_step, on, lists, perturb=False):
    import raymarching
    lists["on"], lists["pinned"] = on, True
    N = sc.rays_o.shape[0]
    m = sc.model
    nears, fars = raymarching.near_far_from_aabb(sc.rays_o, sc.rays_d, m.aabb_infer, m.min_near)
    alive = torch.arange(N, dtype=torch.int32, device="cuda")
    return raymarching.march_rays(N, n_step, alive, nears.clone(), sc.rays_o, sc.rays_d, m.bound, sc.model.density_bitfield[sc.t_idx],
                                  m.cascade, m.grid_size, nears, fars, 128, perturb, 0.0, 1024)


@pytest.mark.parametrize("n_step", [1, 4])
def test_marcher_hangs_the_list_of_filled_slots_on_its_output(model_bits, lists, n_step):
    sc = fixture_scene("cuda", model_bits=model_bits, time=0.5)
    x0, d0, dl0 = _march(sc, n_step, False, lists)
    assert getattr(x0, "_sdn_live", None) is None
    x1, d1, dl1 = _march(sc, n_step, True, lists)
    assert torch.equal(x0, x1) and torch.equal(d0, d1) and torch.equal(dl0, dl1)          # the operator's results do not change
    idx, cnt, ver = x1._sdn_live
    n = int(cnt.item())
    filled = torch.nonzero(dl1[:, 0] > 0).reshape(-1).to(torch.int32)
    assert ver == x1._version and n == filled.numel() and n > 0
    assert torch.equal(torch.sort(idx[:n]).values, filled)                               # unordered list of exactly the filled slots


def test_list_with_perturbed_starts_and_with_no_ray_alive(model_bits, lists):
    """perturb=True (the marcher draws a start offset per ray): the list still names exactly the filled slots; n_alive = 0: no list."""
    import raymarching
    sc = fixture_scene("cuda", model_bits=model_bits, time=0.5)
    torch.manual_seed(4)
    x, d, dl = _march(sc, 2, True, lists, perturb=True)
    idx, cnt, _ = x._sdn_live
    n = int(cnt.item())
    filled = torch.nonzero(dl[:, 0] > 0).reshape(-1).to(torch.int32)
    assert n == filled.numel() and n > 0 and torch.equal(torch.sort(idx[:n]).values, filled)
    m = sc.model
    nears, fars = raymarching.near_far_from_aabb(sc.rays_o, sc.rays_d, m.aabb_infer, m.min_near)
    none = torch.empty(0, dtype=torch.int32, device="cuda")
    x0, d0, dl0 = raymarching.march_rays(0, 8, none, nears.clone(), sc.rays_o, sc.rays_d, m.bound, m.density_bitfield[sc.t_idx], m.cascade,
                                         m.grid_size, nears, fars, 128, False, 0.0, 1024)
    assert x0.shape == (128, 3) and not x0.any() and not dl0.any() and getattr(x0, "_sdn_live", None) is None


@pytest.mark.parametrize("fp32", [False, True])
def test_forward_on_the_marchers_tensor_evaluates_the_listed_slots_only(model_bits, lists, fp32):
    sc = fixture_scene("cuda", model_bits=model_bits, time=0.5)
    model = sc.model.eval()
    x, d, dl = _march(sc, 4, True, lists)
    filled = dl[:, 0] > 0
    assert 0 < int(filled.sum()) < x.shape[0]
    model.fused_inference_f32 = fp32
    model.fused_live_lists_f16 = True       # (off by default under -O: the host-bound loop gains nothing from it)
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=not fp32):
            import sdn_backend
            log = []
            with sdn_backend.launch_log(log):
                s_l, c_l, f_l = model(x, d, sc.time)                  # the marcher's tensor: listed slots
                s_w, c_w, f_w = model(x.clone(), d, sc.time)          # a copy carries no list: every slot
            assert [n for n, _ in log if n.startswith("field_forward")] == ["field_forward_f32" if fp32 else "field_forward_f16"] * 2
            assert torch.equal(s_l[filled], s_w[filled]) and torch.equal(c_l[filled], c_w[filled])
            assert not s_l[~filled].any() and not c_l[~filled].any() and s_w[~filled].any()
            if fp32:
                assert torch.equal(f_l[filled], f_w[filled]) and not f_l[~filled].any()
            # written since the march (version counter): the list no longer describes the tensor
            x.mul_(1.0)
            s_m, c_m, _ = model(x, d, sc.time)
            assert torch.equal(s_m, s_w) and torch.equal(c_m, c_w)
            # switched off on the model
            x2, d2, _ = _march(sc, 4, True, lists)
            model.fused_live_lists = False
            s_o, c_o, _ = model(x2, d2, sc.time)
            assert torch.equal(s_o, s_w) and torch.equal(c_o, c_w)
    finally:
        model.fused_inference_f32 = False
        model.__dict__.pop("fused_live_lists", None)
        model.__dict__.pop("fused_live_lists_f16", None)


@pytest.mark.parametrize("fp32", [False, True])
def test_reference_shaped_frame_is_the_same_with_and_without_lists(model_bits, lists, fp32):
    sc = fixture_scene("cuda", model_bits=model_bits, time=0.26)
    model = sc.model.eval()
    model.fused_inference_f32 = fp32
    model.fused_live_lists_f16 = True

    def frame(on):
        lists["on"], lists["pinned"] = on, True
        import sdn_backend
        log = []
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=not fp32), sdn_backend.launch_log(log):
            out = model.render(sc.rays_o[None], sc.rays_d[None], sc.time, staged=False, perturb=False, bg_color=None)
        return out, sum(u for n, u in log if n.startswith("field_forward"))
    try:
        a, slots_a = frame(False)
        b, slots_b = frame(True)
    finally:
        model.fused_inference_f32 = False
        model.__dict__.pop("fused_live_lists_f16", None)
    assert torch.equal(a["image"], b["image"]) and torch.equal(torch.nan_to_num(a["depth"]), torch.nan_to_num(b["depth"]))
    assert slots_a == slots_b        # the launches are sized by slots either way; the kernel leaves at the list's end


def test_dispatch_switches_the_lists_on_by_itself(model_bits, lists):
    lists["on"], lists["pinned"] = False, False
    sc = fixture_scene("cuda", model_bits=model_bits, time=0.5)
    model = sc.model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model.render(sc.rays_o[None], sc.rays_d[None], sc.time, staged=False, perturb=False, bg_color=None)
    assert lists["on"] == bool(model.fused_live_lists_f16)      # under -O only when asked for (SDN_LIVE_LISTS_F16=1)
    lists["on"] = False
    model.fused_inference_f32 = True
    try:
        with torch.no_grad():
            model.render(sc.rays_o[None], sc.rays_d[None], sc.time, staged=False, perturb=False, bg_color=None)
    finally:
        model.fused_inference_f32 = False
    assert lists["on"]                                           # the fp32 dispatch uses them by default
