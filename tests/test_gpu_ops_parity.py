"""GPU parity: every HIP operator, called through the reference-shaped Python API (which goes through
the C ABI of libsdn_hip.so via ctypes), against the CPU oracle on the same seeded inputs.

Bars (north_star): bit-exact for integer / index / count outputs and -- because both sides evaluate the
same float operations in the same order without contraction -- bit-exact for the marching floats too;
1e-4 relative (fp32) where a transcendental or an atomic summation order is involved, with the
tolerance written at each assert.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402  (tests are one of the three places allowed to use the oracle)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def cam():
    from dnerf_amd import scene
    H = W = 96
    pose = scene.look_at_pose()
    ro, rd = scene.get_rays(pose, scene.intrinsics(H, W), H, W)
    bf = scene.jumpingjacks_occupancy(0.5)
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.2)
    return dict(ro=ro, rd=rd, bf=bf, aabb=aabb, nears=nears, fars=fars, N=ro.shape[0])


def test_library_is_the_hip_build():
    import sdn_backend
    assert b"gfx950" in sdn_backend.lib.sdn_version()


def test_near_far_from_aabb_bit_exact(cam):
    import raymarching
    rng = np.random.default_rng(0)
    # image rays + random rays that miss / graze / start inside the box
    ro = np.concatenate([cam["ro"], rng.uniform(-2, 2, (4096, 3)).astype(np.float32)])
    rd = rng.standard_normal((4096, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd = np.concatenate([cam["rd"], rd])
    n_ref, f_ref = O.near_far_from_aabb(ro, rd, cam["aabb"], 0.2)
    n, f = raymarching.near_far_from_aabb(_dev(ro), _dev(rd), _dev(cam["aabb"]), 0.2)
    assert np.array_equal(n.cpu().numpy().view(np.uint32), n_ref.view(np.uint32))
    assert np.array_equal(f.cpu().numpy().view(np.uint32), f_ref.view(np.uint32))
    assert (n_ref > 1e30).any() and (n_ref < 1e30).any()  # both branches covered


def test_sph_from_ray(cam):
    import raymarching
    ref = O.sph_from_ray(cam["ro"], cam["rd"], 4.0)
    out = raymarching.sph_from_ray(_dev(cam["ro"]), _dev(cam["rd"]), 4.0).cpu().numpy()
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5)  # atan2f/sqrtf: libm vs OCML


def test_morton_roundtrip_and_packbits_bit_exact():
    import raymarching
    rng = np.random.default_rng(1)
    coords = rng.integers(0, 128, (100000, 3)).astype(np.int32)
    idx = raymarching.morton3D(_dev(coords))
    assert idx.dtype == torch.int32
    assert np.array_equal(idx.cpu().numpy(), O.morton3D(coords))
    back = raymarching.morton3D_invert(idx)
    assert np.array_equal(back.cpu().numpy(), coords)
    assert np.array_equal(back.cpu().numpy(), O.morton3D_invert(idx.cpu().numpy()))
    grid = rng.standard_normal((2, 64 ** 3)).astype(np.float32)
    grid[0, :16] = 0.01  # exactly at threshold: '>' must be strict
    bits = raymarching.packbits(_dev(grid), 0.01)
    assert np.array_equal(bits.cpu().numpy(), O.packbits(grid, 0.01))
    # in-place form
    pre = torch.zeros(2 * 64 ** 3 // 8, dtype=torch.uint8, device="cuda")
    out = raymarching.packbits(_dev(grid), 0.5, pre)
    assert out.data_ptr() == pre.data_ptr()
    assert np.array_equal(pre.cpu().numpy(), O.packbits(grid, 0.5))


@pytest.mark.parametrize("plain", [False, True])
@pytest.mark.parametrize("n_step,dt_gamma", [(1, 0.0), (4, 0.0), (8, 0.0), (3, 1.0 / 128)])
def test_march_rays_bit_exact(cam, n_step, dt_gamma, plain):
    """The operator with the reference's signature.  An 8-byte aligned 128^3 slice takes the exact cull grid inside the op (kept per
    slice content); `plain`: a slice at a 4-byte offset takes the plain marcher -- the same bits either way."""
    import raymarching
    N = cam["N"]
    alive = np.arange(N, dtype=np.int32)[::-1].copy()[: N - 7]  # not the identity, ragged count
    n_alive = alive.shape[0]
    rays_t = cam["nears"].copy()
    ref = O.march_rays(n_alive, n_step, alive, rays_t, cam["ro"], cam["rd"], 1.0, cam["bf"], 1, 128, cam["nears"], cam["fars"],
                       align=128, dt_gamma=dt_gamma)
    bf = _dev(cam["bf"])
    if plain:
        store = torch.empty(bf.numel() + 8, dtype=torch.uint8, device="cuda")
        bf = store[4:4 + bf.numel()].copy_(bf)
        assert bf.data_ptr() % 8 == 4 and raymarching._cull_grid_of(bf, 1, 128) is None
    else:
        assert raymarching._cull_grid_of(bf, 1, 128) is not None
    out = raymarching.march_rays(n_alive, n_step, _dev(alive), _dev(rays_t), _dev(cam["ro"]), _dev(cam["rd"]), 1.0, bf, 1, 128,
                                 _dev(cam["nears"]), _dev(cam["fars"]), 128, False, dt_gamma, 1024)
    if not plain:
        # an in-place rewrite of the slice is seen (tensor version): emptied occupancy -> no sample; restored -> the samples again
        keep = bf.clone()
        bf.zero_()
        none = raymarching.march_rays(n_alive, n_step, _dev(alive), _dev(rays_t), _dev(cam["ro"]), _dev(cam["rd"]), 1.0, bf, 1, 128,
                                      _dev(cam["nears"]), _dev(cam["fars"]), 128, False, dt_gamma, 1024)
        assert float(none[2].abs().max()) == 0.0
        bf.copy_(keep)
        again = raymarching.march_rays(n_alive, n_step, _dev(alive), _dev(rays_t), _dev(cam["ro"]), _dev(cam["rd"]), 1.0, bf, 1, 128,
                                       _dev(cam["nears"]), _dev(cam["fars"]), 128, False, dt_gamma, 1024)
        assert all(torch.equal(a, b) for a, b in zip(again, out))
    for o, r, name in zip(out, ref, ("xyzs", "dirs", "deltas")):
        assert o.shape == r.shape, name  # includes the "+128 when already aligned" padding rule
        assert np.array_equal(o.cpu().numpy().view(np.uint32), r.view(np.uint32)), name
    assert (ref[2][:, 0] > 0).sum() > 100  # the test actually sampled something


@pytest.mark.parametrize("kind", ["lego", "scattered"])
def test_march_rays_ex_other_occupancies(cam, kind):
    """Cull grid / fine-bit cache on other occupancy shapes: a compact box (cache active, different bounding box) and voxels
    scattered over the whole volume (bounding box too large for LDS: the marcher must fall back to global bit loads)."""
    import raymarching
    from dnerf_amd import scene
    if kind == "lego":
        bf = scene.lego_occupancy()
    else:
        rng = np.random.default_rng(5)
        bits = (rng.random(128 ** 3) < 0.002).astype(np.uint8)
        bf = np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
    N = cam["N"]
    alive = np.arange(N, dtype=np.int32)
    cull = raymarching.build_cull_grid(_dev(bf))
    for n_step, advance in ((1, 0.0), (8, 0.4)):
        rays_t = (cam["nears"] + np.float32(advance)).astype(np.float32)
        ref = O.march_rays(N, n_step, alive, rays_t, cam["ro"], cam["rd"], 1.0, bf, 1, 128, cam["nears"], cam["fars"], align=128)
        out = raymarching.march_rays_ex(N, n_step, _dev(alive), _dev(rays_t), _dev(cam["ro"]), _dev(cam["rd"]), 1.0, _dev(bf), 1, 128,
                                        _dev(cam["fars"]), 128, 0.0, 1024, cull, True)
        for o, r, name in zip(out[:3], ref, ("xyzs", "dirs", "deltas")):
            assert np.array_equal(o.cpu().numpy().view(np.uint32), r.view(np.uint32)), (kind, name, n_step)
        assert int(out[4].item()) == int((ref[2][: N * n_step, 0] > 0).sum()) > 0


@pytest.mark.parametrize("n_step", [1, 8])
def test_march_rays_ex_cull_and_live_list_are_exact(cam, n_step):
    """The cull-grid early-out and the live list must not change a single bit of the samples, at any stage of a render:
    checked on fresh rays and on rays advanced part-way through / past the figure."""
    import raymarching
    N = cam["N"]
    cull = raymarching.build_cull_grid(_dev(cam["bf"]))
    # reference semantics of the cull grid itself: cell marked <=> an occupied voxel within its 3x3x3 coarse neighbourhood
    bits = np.unpackbits(cam["bf"], bitorder="little")
    from dnerf_amd import scene
    g = np.arange(128)
    ix, iy, iz = np.meshgrid(g, g, g, indexing="ij")
    occ = np.zeros((128, 128, 128), bool)
    occ[ix.reshape(-1), iy.reshape(-1), iz.reshape(-1)] = bits[scene.morton3d(ix.reshape(-1), iy.reshape(-1), iz.reshape(-1))] > 0
    coarse = occ.reshape(32, 4, 32, 4, 32, 4).any(axis=(1, 3, 5))
    pad = np.pad(coarse, 1)
    dil = np.zeros_like(coarse)
    for a in range(3):
        for b in range(3):
            for c in range(3):
                dil |= pad[a:a + 32, b:b + 32, c:c + 32]
    raw = cull.cpu().numpy()
    cull_bits = np.unpackbits(raw[:4096], bitorder="little").reshape(32, 32, 32).astype(bool)   # bit c = (z*32 + y)*32 + x
    assert np.array_equal(cull_bits, dil.transpose(2, 1, 0))
    xs, ys, zs = np.nonzero(dil)
    assert list(raw[4096:4120].view(np.int32)) == [xs.min(), ys.min(), zs.min(), xs.max(), ys.max(), zs.max()]
    alive = np.arange(N, dtype=np.int32)
    for advance in (0.0, 0.15, 0.6, 1.5):
        rays_t = (cam["nears"] + np.float32(advance)).astype(np.float32)
        ref = O.march_rays(N, n_step, alive, rays_t, cam["ro"], cam["rd"], 1.0, cam["bf"], 1, 128, cam["nears"], cam["fars"], align=128)
        out = raymarching.march_rays_ex(N, n_step, _dev(alive), _dev(rays_t), _dev(cam["ro"]), _dev(cam["rd"]), 1.0, _dev(cam["bf"]), 1, 128,
                                        _dev(cam["fars"]), 128, 0.0, 1024, cull, True)
        for o, r, name in zip(out[:3], ref, ("xyzs", "dirs", "deltas")):
            assert np.array_equal(o.cpu().numpy().view(np.uint32), r.view(np.uint32)), (name, advance)
        live = np.nonzero(ref[2][: N * n_step, 0] > 0)[0]
        cnt = int(out[4].item())
        assert cnt == live.shape[0]
        assert np.array_equal(np.sort(out[3][:cnt].cpu().numpy()), live)   # unordered, complete, no duplicates


def test_march_rays_perturb_uses_noise(cam):
    import raymarching
    N = cam["N"]
    alive = np.arange(N, dtype=np.int32)
    torch.manual_seed(3)
    a = raymarching.march_rays(N, 2, _dev(alive), _dev(cam["nears"]), _dev(cam["ro"]), _dev(cam["rd"]), 1.0, _dev(cam["bf"]), 1, 128,
                               _dev(cam["nears"]), _dev(cam["fars"]), 128, True, 0.0, 1024)
    torch.manual_seed(3)
    noises = torch.rand(N, dtype=torch.float32, device="cuda").cpu().numpy()
    ref = O.march_rays(N, 2, alive, cam["nears"], cam["ro"], cam["rd"], 1.0, cam["bf"], 1, 128, cam["nears"], cam["fars"], align=128,
                       perturb=True, noises=noises)
    assert np.array_equal(a[0].cpu().numpy().view(np.uint32), ref[0].view(np.uint32))


def _fake_field(xyzs, seed=0):
    """Deterministic stand-in for the network so that compositing sees dense-ish media."""
    rng = np.random.default_rng(seed)
    M = xyzs.shape[0]
    sigmas = (rng.random(M, dtype=np.float32) * 60).astype(np.float32)
    rgbs = rng.random((M, 3), dtype=np.float32)
    return sigmas, rgbs


def test_inference_loop_bit_exact_counts(cam):
    """march -> composite -> compact, the whole reference loop (dnerf/renderer.py:350-376) on both sides:
    alive counts, alive ids and the n_step schedule must agree exactly at every iteration; image / depth /
    weights to 1e-6 (only exp() differs: OCML double exp vs glibc, both rounded to float)."""
    import raymarching
    N = cam["N"]
    d = {k: _dev(cam[k]) for k in ("ro", "rd", "bf", "nears", "fars")}
    ws_r, dp_r, im_r = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    ws, dp, im = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, 3, device="cuda")
    alive_r = np.arange(N, dtype=np.int32)
    alive = torch.arange(N, dtype=torch.int32, device="cuda")
    t_r = cam["nears"].copy()
    t = d["nears"].clone()
    step, it = 0, 0
    total_samples = 0
    while step < 1024:
        n_alive = alive_r.shape[0]
        assert alive.shape[0] == n_alive
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xr, dr, lr = O.march_rays(n_alive, n_step, alive_r, t_r, cam["ro"], cam["rd"], 1.0, cam["bf"], 1, 128, cam["nears"], cam["fars"], align=128)
        x, dd, l = raymarching.march_rays(n_alive, n_step, alive, t, d["ro"], d["rd"], 1.0, d["bf"], 1, 128, d["nears"], d["fars"], 128, False, 0, 1024)
        assert np.array_equal(l.cpu().numpy().view(np.uint32), lr.view(np.uint32))
        total_samples += int((lr[:, 0] > 0).sum())
        sig, rgb = _fake_field(xr, seed=it)
        O.composite_rays(n_alive, n_step, alive_r, t_r, sig, rgb, lr, ws_r, dp_r, im_r, 1e-2)
        raymarching.composite_rays(n_alive, n_step, alive, t, _dev(sig), _dev(rgb), l, ws, dp, im, 1e-2)
        assert np.array_equal(alive.cpu().numpy(), alive_r), f"termination flags differ at iteration {it}"
        assert np.array_equal(t.cpu().numpy().view(np.uint32), t_r.view(np.uint32))
        # device-side stable compaction == the caller's boolean mask select
        comp, cnt = raymarching.compact_alive(alive)
        alive_r = alive_r[alive_r >= 0]
        assert int(cnt.item()) == alive_r.shape[0]
        alive = comp[: alive_r.shape[0]].clone()
        assert np.array_equal(alive.cpu().numpy(), alive_r)
        step += n_step
        it += 1
    assert it > 3 and total_samples > 1000
    np.testing.assert_allclose(ws.cpu().numpy(), ws_r, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dp.cpu().numpy(), dp_r, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(im.cpu().numpy(), im_r, rtol=1e-6, atol=1e-7)


def test_compact_alive_edge_cases():
    import raymarching
    for n, frac in ((1, 1.0), (1, 0.0), (63, 0.5), (64, 0.5), (1024, 0.0), (1025, 1.0), (5000, 0.3), (300000, 0.9)):
        rng = np.random.default_rng(n)
        a = np.arange(n, dtype=np.int32)
        a[rng.random(n) >= frac] = -1
        out, cnt = raymarching.compact_alive(_dev(a))
        ref = a[a >= 0]
        assert int(cnt.item()) == ref.shape[0]
        assert np.array_equal(out[: ref.shape[0]].cpu().numpy(), ref)


@pytest.mark.parametrize("perturb,force_all,mean_count", [(False, False, -1), (True, False, -1), (False, True, 500), (False, False, 20000)])
def test_march_rays_train_exact_per_ray(cam, perturb, force_all, mean_count):
    import raymarching
    N = 2048
    rng = np.random.default_rng(7)
    sel = rng.integers(0, cam["N"], N)
    ro, rd, nears, fars = cam["ro"][sel], cam["rd"][sel], cam["nears"][sel], cam["fars"][sel]
    torch.manual_seed(11)
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    out = raymarching.march_rays_train(_dev(ro), _dev(rd), 1.0, _dev(cam["bf"]), 1, 128, _dev(nears), _dev(fars), counter, mean_count, perturb,
                                       128, force_all, 0, 1024)
    torch.manual_seed(11)
    noises = torch.rand(N, dtype=torch.float32, device="cuda").cpu().numpy() if perturb else None
    counter_r = np.zeros(2, np.int32)
    ref = O.march_rays_train(ro, rd, 1.0, cam["bf"], 1, 128, nears, fars, counter_r, mean_count, perturb, 128, force_all, 0, 1024, noises=noises)
    assert np.array_equal(counter.cpu().numpy(), counter_r)
    for o, r, name in zip(out, ref, ("xyzs", "dirs", "deltas", "rays")):
        assert o.shape == r.shape, name
        assert np.array_equal(o.cpu().numpy().view(np.uint32), r.view(np.uint32)), name
    assert counter_r[0] > 1000


def test_march_rays_train_overflow_drops_rays_deterministically(cam):
    import raymarching
    N = 1024
    sel = np.arange(cam["N"])[cam["nears"] < 1e30][:N]
    ro, rd, nears, fars = cam["ro"][sel], cam["rd"][sel], cam["nears"][sel], cam["fars"][sel]
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    counter_r = np.zeros(2, np.int32)
    # a budget far below what the rays need: later rays must be dropped (offset + n > M), raymarching.cu:416
    out = raymarching.march_rays_train(_dev(ro), _dev(rd), 1.0, _dev(cam["bf"]), 1, 128, _dev(nears), _dev(fars), counter, 256, False, 128, False, 0, 1024)
    ref = O.march_rays_train(ro, rd, 1.0, cam["bf"], 1, 128, nears, fars, counter_r, 256, False, 128, False, 0, 1024)
    assert out[0].shape[0] == 384 == ref[0].shape[0]
    for o, r in zip(out, ref):
        assert np.array_equal(o.cpu().numpy().view(np.uint32), r.view(np.uint32))


def test_composite_rays_train_forward_backward(cam):
    import raymarching
    N = 2048
    rng = np.random.default_rng(9)
    sel = rng.integers(0, cam["N"], N)
    ro, rd, nears, fars = cam["ro"][sel], cam["rd"][sel], cam["nears"][sel], cam["fars"][sel]
    xyzs, dirs, deltas, rays = O.march_rays_train(ro, rd, 1.0, cam["bf"], 1, 128, nears, fars, np.zeros(2, np.int32), -1, False, 128)
    M = xyzs.shape[0]
    sig, rgb = _fake_field(xyzs, 5)
    sig *= 0.3
    ws_r, dp_r, im_r = O.composite_rays_train_forward(sig, rgb, deltas, rays, 1e-4)
    s_t = _dev(sig).requires_grad_(True)
    c_t = _dev(rgb).requires_grad_(True)
    ws, dp, im = raymarching.composite_rays_train(s_t, c_t, _dev(deltas), _dev(rays), 1e-4)
    # exp() is the only non-identical operation (OCML vs glibc double exp, rounded to float): 1e-6
    np.testing.assert_allclose(ws.detach().cpu().numpy(), ws_r, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dp.detach().cpu().numpy(), dp_r, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(im.detach().cpu().numpy(), im_r, rtol=1e-6, atol=1e-7)
    g_ws = rng.standard_normal(N).astype(np.float32)
    g_im = rng.standard_normal((N, 3)).astype(np.float32)
    gs_r, gc_r = O.composite_rays_train_backward(g_ws, g_im, sig, rgb, deltas, rays, ws_r, im_r, 1e-4)
    torch.autograd.backward([ws, dp, im], [_dev(g_ws), torch.zeros_like(dp), _dev(g_im)])
    # north_star tolerance: 1e-4 relative (fp32); measured error is ~1e-6
    np.testing.assert_allclose(s_t.grad.cpu().numpy(), gs_r, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(c_t.grad.cpu().numpy(), gc_r, rtol=1e-4, atol=1e-7)
    assert M > 1000 and np.abs(gs_r).max() > 0


# ------------------------------------------------------------------------------------------------
# encoders
# ------------------------------------------------------------------------------------------------
GRID_CASES = [
    # D, L, C, H, desired_res, log2T, gridtype, align, interp
    (3, 16, 2, 16, 2048, 19, "tiled", False, "linear"),   # the dnerf configuration (dnerf/network.py:12,59)
    (3, 16, 2, 16, 2048, 19, "hash", False, "linear"),
    (3, 4, 2, 4, 64, 8, "hash", True, "smoothstep"),
    (2, 4, 2, 16, 2048, 19, "hash", False, "linear"),     # the bg encoder shape (dnerf/network.py:102)
    (3, 6, 4, 8, 256, 12, "tiled", False, "smoothstep"),
    (3, 5, 1, 8, 128, 10, "hash", False, "linear"),
    (3, 3, 8, 4, 32, 9, "tiled", True, "linear"),
    (4, 3, 2, 4, 16, 10, "hash", False, "linear"),
    (5, 2, 2, 3, 6, 10, "hash", False, "linear"),
]


def _grid_setup(D, L, C, H, res, log2T, align, seed, B):
    offsets, pls = O.grid_offsets(D, L, C, 2, H, log2T, res, align)
    rng = np.random.default_rng(seed)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.random((B, D), dtype=np.float32)
    x[0] = 0.0
    x[1] = 1.0               # both ends of the closed range are in-bounds
    x[2, 0] = -1e-3          # out of range -> zeros
    x[3, D - 1] = 1.0 + 1e-3
    return offsets, pls, emb, x


@pytest.mark.parametrize("case", GRID_CASES)
@pytest.mark.parametrize("half", [False, True])
def test_grid_encode_forward_bit_exact(case, half):
    from gridencoder.grid import grid_encode
    D, L, C, H, res, log2T, gridtype, align, interp = case
    gid, iid = {"hash": 0, "tiled": 1}[gridtype], {"linear": 0, "smoothstep": 1}[interp]
    offsets, pls, emb, x = _grid_setup(D, L, C, H, res, log2T, align, 3, 3001)
    if half:
        emb = emb.astype(np.float16)
    ref, ref_dd = O.grid_encode_forward(x, emb, offsets, pls, H, True, gid, align, iid)
    xt = _dev(x).requires_grad_(True)
    out = grid_encode(xt, _dev(emb), _dev(offsets), pls, H, True, gid, align, iid)
    assert out.shape == (x.shape[0], L * C) and out.dtype == (torch.float16 if half else torch.float32)
    o = out.detach().cpu().numpy()
    bits = np.uint16 if half else np.uint32
    assert np.array_equal(o.view(bits), ref.view(bits)), "forward must be bit-identical (same ops, same order)"
    assert not o[2].any() and not o[3].any() and o[0].any() and o[1].any()


@pytest.mark.parametrize("case", GRID_CASES[:5])
@pytest.mark.parametrize("half", [False, True])
def test_grid_encode_backward(case, half):
    from gridencoder.grid import grid_encode
    D, L, C, H, res, log2T, gridtype, align, interp = case
    gid, iid = {"hash": 0, "tiled": 1}[gridtype], {"linear": 0, "smoothstep": 1}[interp]
    offsets, pls, emb, x = _grid_setup(D, L, C, H, res, log2T, align, 4, 2000)
    rng = np.random.default_rng(5)
    g = rng.standard_normal((x.shape[0], L * C)).astype(np.float32)
    if half:
        emb = emb.astype(np.float16)
        g = (g * 1e-2).astype(np.float16)
    _, dd = O.grid_encode_forward(x, emb, offsets, pls, H, True, gid, align, iid)
    ge_r, gi_r = O.grid_encode_backward(g, x, emb, offsets, pls, H, dd, gid, align, iid)
    xt = _dev(x).requires_grad_(True)
    et = _dev(emb).requires_grad_(True)
    out = grid_encode(xt, et, _dev(offsets), pls, H, True, gid, align, iid)
    out.backward(_dev(g))
    ge, gi = et.grad.cpu().numpy().astype(np.float32), xt.grad.cpu().numpy()
    ge_r = ge_r.astype(np.float32)
    if half:
        # half accumulation: each atomic rounds the running sum to fp16; order differs from the oracle's
        # serial order, so agreement is to a few fp16 ulps of the largest partial sum
        np.testing.assert_allclose(ge, ge_r, rtol=2e-2, atol=2e-3 * max(1.0, float(np.abs(ge_r).max())))
        np.testing.assert_allclose(gi, gi_r, rtol=2e-2, atol=2e-2 * float(np.abs(gi_r).max()))
    else:
        # fp32 atomics: summation order only -> 1e-4 relative (north_star), measured ~1e-6
        np.testing.assert_allclose(ge, ge_r, rtol=1e-4, atol=1e-5 * float(np.abs(ge_r).max()))
        np.testing.assert_allclose(gi, gi_r, rtol=1e-4, atol=1e-5 * float(np.abs(gi_r).max()))
    assert np.abs(ge_r).max() > 0 and np.abs(gi_r).max() > 0


def test_grid_encoder_module_matches_reference_layout():
    from gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                      gridtype="tiled").cuda()
    assert tuple(enc.embeddings.shape) == (6119864, 2)
    assert enc.offsets.dtype == torch.int32 and enc.offsets.shape[0] == 17
    assert list(enc.state_dict().keys()) == ["embeddings", "offsets"]
    assert float(enc.embeddings.abs().max()) <= 1e-4
    x = torch.rand(1000, 3, device="cuda") * 2 - 1
    y = enc(x, bound=1)
    assert y.shape == (1000, 32) and y.dtype == torch.float32
    with torch.autocast("cuda", dtype=torch.float16):
        yh = enc(x, bound=1)
    assert yh.dtype == torch.float16
    ref, _ = O.grid_encode_forward(((x + 1) / 2).cpu().numpy(), enc.embeddings.detach().cpu().numpy(), enc.offsets.cpu().numpy(),
                                   enc.per_level_scale, 16, False, 1, False, 0)
    assert np.array_equal(y.detach().cpu().numpy().view(np.uint32), ref.view(np.uint32))


def test_sh_encode_vs_oracle_and_reference_closed_forms(golden_dir):
    from shencoder import SHEncoder, sh_encode
    g = np.load(f"{golden_dir}/sh_reference_closed_form.npz")
    x = g["inputs"]
    for degree in range(1, 9):
        xt = _dev(x).requires_grad_(True)
        out = sh_encode(xt, degree, True)
        ref, ref_dd = O.sh_encode_forward(x, degree, True)
        C2 = degree * degree
        # fp32 polynomial evaluation in a different (factored) order than the float64 references:
        # 1e-4 relative with an absolute floor for values that cancel to ~0
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(out.detach().cpu().numpy(), g["outputs"][:, :C2], rtol=1e-4, atol=2e-5)
        gy = np.random.default_rng(degree).standard_normal(ref.shape).astype(np.float32)
        out.backward(_dev(gy))
        gi_ref = O.sh_encode_backward(gy, ref_dd, degree)
        gi_gold = np.stack([(gy * g[k][:, :C2]).sum(1) for k in ("dx", "dy", "dz")], 1)
        scale = float(np.abs(gi_ref).max())
        np.testing.assert_allclose(xt.grad.cpu().numpy(), gi_ref, rtol=1e-4, atol=1e-5 * scale)
        np.testing.assert_allclose(xt.grad.cpu().numpy(), gi_gold, rtol=1e-4, atol=1e-5 * scale)
    enc = SHEncoder(degree=4)
    y = enc(_dev(x))
    assert y.shape == (x.shape[0], 16) and not y.requires_grad


def test_freq_encode_vs_oracle_and_reference_torch(golden_dir):
    from freqencoder import FreqEncoder
    g = np.load(f"{golden_dir}/freq_reference_torch.npz")
    for name in ("xyz", "time"):
        x, deg = g[f"{name}_inputs"], int(g[f"{name}_degree"])
        enc = FreqEncoder(input_dim=x.shape[1], degree=deg)
        xt = _dev(x).requires_grad_(True)
        y = enc(xt)
        ref = O.freq_encode_forward(x, deg)
        # oracle: same float argument (incl. the reference's float pi/2 phase add), sin correctly rounded: 1e-6
        np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=0, atol=1e-6)
        # reference torch FreqEncoder computes cos(x*2^f) directly; the kernel's sin(x*2^f + fl(pi/2)) differs by
        # the rounding of the phase add (<= ulp(2^9)/2 = 3e-5): inside the stated 1e-4 absolute bar
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{name}_outputs"], rtol=0, atol=1e-4)
        y.backward(_dev(g[f"{name}_grad_outputs"]))
        gi_ref = O.freq_encode_backward(g[f"{name}_grad_outputs"], ref, x.shape[1], deg)
        scale = float(np.abs(gi_ref).max())
        np.testing.assert_allclose(xt.grad.cpu().numpy(), gi_ref, rtol=1e-4, atol=1e-5 * scale)
        np.testing.assert_allclose(xt.grad.cpu().numpy(), g[f"{name}_grad_inputs"], rtol=1e-4, atol=1e-4 * scale)


def test_ops_reject_bad_buffers():
    import raymarching
    from sdn_backend import SdnError
    a = torch.zeros(10, 3, device="cuda")
    with pytest.raises(SdnError):
        raymarching.march_rays(10, 1, torch.zeros(10, dtype=torch.int64, device="cuda"), torch.zeros(10, device="cuda"), a, a, 1.0,
                               torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device="cuda"), 1, 128, torch.zeros(10, device="cuda"),
                               torch.zeros(10, device="cuda"))


def test_get_rays_matches_reference_formula_and_scene_rays():
    """`get_rays` (nerf/utils.py:54-137): the full-frame HIP kernel against the reference's torch expression evaluated on the
    CPU (tolerance 2e-6: its norm / matmul summation orders are library-defined), against the numpy rays the benchmark scene
    uses, and the sampled variants' bookkeeping (indices in range, rays equal to the gathered full-frame rays)."""
    from dnerf_amd import scene, utils
    H, W = 37, 53
    pose = scene.look_at_pose(70.0, 20.0).astype(np.float32)
    intr = scene.intrinsics(H, W)
    poses = torch.from_numpy(np.stack([pose, scene.look_at_pose(200.0, 35.0).astype(np.float32)])).cuda()
    out = utils.get_rays(poses, intr, H, W)
    assert out["rays_o"].shape == (2, H * W, 3) and out["rays_d"].shape == (2, H * W, 3)
    ref = utils.get_rays(poses.cpu(), intr, H, W)          # the torch expression path (no device: never reaches the kernel)
    assert torch.allclose(out["rays_d"].cpu(), ref["rays_d"], atol=2e-6, rtol=0)
    assert torch.equal(out["rays_o"].cpu(), ref["rays_o"].contiguous())
    ro, rd = scene.get_rays(pose, intr, H, W)
    assert np.allclose(out["rays_d"][0].cpu().numpy(), rd, atol=2e-6) and np.array_equal(out["rays_o"][0].cpu().numpy(), ro)
    assert torch.allclose(out["rays_d"].norm(dim=-1), torch.ones(2, H * W, device="cuda"), atol=1e-6)
    torch.manual_seed(0)
    for kw in (dict(N=100), dict(N=64, patch_size=4), dict(N=50, error_map=torch.rand(2, 128 * 128))):
        smp = utils.get_rays(poses, intr, H, W, **kw)
        inds = smp["inds"]
        assert inds.shape[0] == 2 and int(inds.min()) >= 0 and int(inds.max()) < H * W
        picked = torch.gather(out["rays_d"], 1, inds[..., None].expand(-1, -1, 3))
        assert torch.allclose(smp["rays_d"], picked, atol=2e-6, rtol=0)


def test_load_reference_checkpoint_roundtrip(tmp_path):
    """A checkpoint in the reference trainer's layout ({'model': state_dict, 'mean_count': ...}) loads by name."""
    from dnerf_amd import utils
    from dnerf_amd.bench_scene import build_model
    a, b = build_model(seed=3), build_model(seed=4)
    path = str(tmp_path / "ngp.pth")
    torch.save({"model": a.state_dict(), "mean_count": 123, "epoch": 7}, path)
    missing, unexpected = utils.load_reference_checkpoint(b, path, map_location="cuda")
    assert not missing and not unexpected and b.mean_count == 123
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_load_full_reference_layout_checkpoint(tmp_path):
    """The reference trainer's FULL checkpoint layout (save_checkpoint(full=True), nerf/utils.py:1033-1069): epoch / global_step /
    stats / mean_count / mean_density / optimizer / lr_scheduler / scaler / ema / model; and the "best" layout, whose model entry has
    no density_grid (:1085-1086).  Everything is restored by name; the file is read with weights_only=True."""
    from dnerf_amd import utils
    from dnerf_amd.bench_scene import build_model
    a, b = build_model(seed=3).train(), build_model(seed=4).train()
    a.density_grid.uniform_(0, 1)

    def make(m):
        opt = torch.optim.Adam(m.get_params(1e-2, 1e-3), betas=(0.9, 0.99), eps=1e-15)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / 30000, 1))      # main_dnerf.py:134
        return opt, sched, torch.amp.GradScaler("cuda")
    opt_a, sched_a, scaler_a = make(a)
    for _ in range(2):                                   # give the optimizer / scheduler / scaler non-trivial state
        opt_a.zero_grad(set_to_none=True)
        loss = sum((p.float() ** 2).mean() for p in a.parameters())
        scaler_a.scale(loss).backward()
        scaler_a.step(opt_a); scaler_a.update(); sched_a.step()
    ema_state = {"decay": 0.95, "num_updates": 2, "shadow_params": [p.detach().clone() for p in a.parameters()], "collected_params": None}
    state = {"epoch": 5, "global_step": 1234, "stats": {"loss": [0.5, 0.25], "valid_loss": [], "results": [np.float64(0.1)], "checkpoints": [], "best_result": np.float64(0.1)},   # numpy scalars, as PSNRMeter.measure() returns them
             "mean_count": 4321, "mean_density": 0.125, "optimizer": opt_a.state_dict(), "lr_scheduler": sched_a.state_dict(),
             "scaler": scaler_a.state_dict(), "ema": ema_state, "model": a.state_dict()}
    full = str(tmp_path / "ngp_ep0005.pth")
    torch.save(state, full)
    opt_b, sched_b, scaler_b = make(b)

    class Ema:                                           # torch_ema is absent here: the loader only calls load_state_dict on it
        def load_state_dict(self, sd):
            self.sd = sd
    ema_b = Ema()
    missing, unexpected = utils.load_reference_checkpoint(b, full, map_location="cuda", optimizer=opt_b, lr_scheduler=sched_b, scaler=scaler_b,
                                                          ema=ema_b, model_only=False)
    info = utils.load_reference_checkpoint.last
    assert not missing and not unexpected and b.mean_count == 4321 and b.mean_density == 0.125
    assert info["epoch"] == 5 and info["global_step"] == 1234 and info["stats"]["best_result"] == 0.1
    assert sorted(info["restored"]) == ["ema", "lr_scheduler", "optimizer", "scaler"] and not info["failed"]
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    sa, sb = opt_a.state_dict()["state"], opt_b.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) for k in sa)
    assert scaler_b.get_scale() == scaler_a.get_scale() and sched_b.last_epoch == sched_a.last_epoch == 2
    assert ema_b.sd["num_updates"] == 2
    # "best" checkpoint: no density_grid in the model entry -> reported missing, everything else loaded
    best = dict(state, model={k: v for k, v in a.state_dict().items() if k != "density_grid"})
    for k in ("optimizer", "lr_scheduler", "scaler", "ema"):
        best.pop(k)
    path = str(tmp_path / "ngp.pth")
    torch.save(best, path)
    c = build_model(seed=5)
    missing, unexpected = utils.load_reference_checkpoint(c, path, map_location="cuda")
    assert missing == ["density_grid"] and not unexpected
    assert torch.equal(c.encoder.embeddings, a.encoder.embeddings) and torch.equal(c.density_bitfield, a.density_bitfield)


def test_seal_bbox_kernels_match_the_torch_restatement():
    """csrc/seal.hip against dnerf_amd/seal_mapper's torch restatement of SealNeRF/seal_utils.py (run on the CPU): identical
    masks except within rounding of a face (excluded by an analytic margin), mapped coordinates / directions to 2e-6, colours
    to 2e-6; empty (all-zero) slots are never mapped."""
    from dnerf_amd import seal_mapper as SM
    rng = np.random.default_rng(5)

    def rot(axis, deg):
        a = np.asarray(axis, np.float64) / np.linalg.norm(axis); t = np.deg2rad(deg)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * K @ K
    R0, c0, half = rot([0, 0, 1], 20.0), np.array([-0.2, 0.0, 0.1]), np.array([0.15, 0.1, 0.2])
    corners = (np.array([[sx, sy, sz] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)], np.float64) * half) @ R0.T + c0
    T = np.eye(4); T[:3, :3] = rot([0, 1, 0], 30.0); T[:3, 3] = [0.45, 0.05, -0.1]
    scale = np.array([1.5, 1.0, 0.8])
    for bound_type in ("to", "both"):
        cfg = {"type": "bbox", "raw": corners.tolist(), "transform": T.tolist(), "scale": scale.tolist(), "boundType": bound_type,
               "hsv": [0.15, -0.05, 0.02]}
        cpu, gpu = SM.get_seal_mapper(cfg), SM.get_seal_mapper(cfg)
        p = rng.uniform(-0.8, 0.8, (50000, 3)).astype(np.float32)
        p[:100] = 0.0                                                  # empty slots
        d = rng.standard_normal((50000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
        cp, cd, cm = cpu.map_to_origin(torch.from_numpy(p), torch.from_numpy(d))
        gp, gd, gm = gpu.map_to_origin(torch.from_numpy(p).cuda(), torch.from_numpy(d).cuda())
        assert gm.dtype == torch.bool and not bool(gm[:100].any())
        # analytic distance of every point to the faces of the mapped boxes: the masks may only differ within 1e-5 of a face
        def margin(verts):
            e = [verts[1] - verts[0], verts[2] - verts[0], verts[4] - verts[0]]
            loc = np.stack([((p - verts[0]) @ (ei / np.linalg.norm(ei))) for ei in e], 1)
            ln = np.array([np.linalg.norm(ei) for ei in e])
            return np.minimum(np.abs(loc), np.abs(loc - ln)).min(1)
        m = margin(gpu.to_vertices)
        if bound_type == "both":
            m = np.minimum(m, margin(gpu.from_vertices))
        clear = m > 1e-5
        assert np.array_equal(cm.numpy()[clear], gm.cpu().numpy()[clear]) and int(cm.sum()) > 200
        both = (cm & gm.cpu()).numpy()
        assert np.allclose(gp.cpu().numpy()[both], cp.numpy()[both], atol=2e-6) and np.allclose(gd.cpu().numpy()[both], cd.numpy()[both], atol=2e-6)
        assert torch.equal(gp.cpu()[~gm.cpu()], torch.from_numpy(p)[~gm.cpu()])      # untouched outside the mask
        cols = rng.random((50000, 3), dtype=np.float32)
        want = torch.from_numpy(cols).clone()
        want[cm] = cpu.map_color(None, None, want[cm])
        got = gpu.map_color_(torch.from_numpy(cols).cuda(), cm.cuda())
        assert np.allclose(got.cpu().numpy(), want.numpy(), atol=2e-6)


@pytest.mark.parametrize("max_steps,H,full", [(1024, 128, False), (1024, 128, True), (256, 128, False), (1024, 64, False), (512, 32, True)])
def test_march_rays_train_wave_per_ray_kernel_exact(cam, max_steps, H, full):
    """The constant-step (dt_gamma == 0) training march runs one WAVE per ray over 64-point windows of the step lattice
    (k_march_train_count_wave): counts, per-ray offsets and every sample bit-identical to the oracle's sequential march -- with
    perturbed starts, on coarser grids (no cull grid below 128^3), with a coarser step (max_steps 256), and on a fully occupied
    grid, where whole windows are sampled and rays collect hundreds of samples (the step is the cube diagonal / max_steps, so the
    max_steps cap itself is only ever reached through rounding)."""
    import raymarching
    from dnerf_amd import scene
    N = 1500
    rng = np.random.default_rng(21)
    sel = rng.integers(0, cam["N"], N)
    ro, rd = cam["ro"][sel], cam["rd"][sel]
    if full:
        bf = np.full(H ** 3 // 8, 255, np.uint8)
    else:
        bf = cam["bf"] if H == 128 else scene.jumpingjacks_occupancy(0.5, H)
    nears, fars = cam["nears"][sel], cam["fars"][sel]
    noises = rng.random(N, dtype=np.float32)
    real_rand = torch.rand
    import raymarching.raymarching as rm_mod
    nz = torch.from_numpy(noises).cuda()
    rm_mod.torch.rand = lambda *a, **k: nz.clone() if a == (N,) else real_rand(*a, **k)
    try:
        counter = torch.zeros(2, dtype=torch.int32, device="cuda")
        out = raymarching.march_rays_train(_dev(ro), _dev(rd), 1.0, _dev(bf), 1, H, _dev(nears), _dev(fars), counter, -1, True, 128, False, 0, max_steps)
    finally:
        rm_mod.torch.rand = real_rand
    counter_r = np.zeros(2, np.int32)
    ref = O.march_rays_train(ro, rd, 1.0, bf, 1, H, nears, fars, counter_r, -1, True, 128, False, 0, max_steps, noises=noises)
    assert np.array_equal(counter.cpu().numpy(), counter_r) and counter_r[0] > (50000 if full else 300)
    for o, r, name in zip(out, ref, ("xyzs", "dirs", "deltas", "rays")):
        assert o.shape == r.shape, name
        assert np.array_equal(o.cpu().numpy().view(np.uint32), r.view(np.uint32)), name


@pytest.mark.parametrize("H,dt_gamma", [(64, 0.0), (64, 1.0 / 128), (128, 1.0 / 256)])
def test_generic_marcher_bound2_cascade2_other_grid_sizes(cam, H, dt_gamma):
    """`MarcherT<false>` (bound 2, two cascades: mip_from_pos / mip_from_dt, the double-typed cell index, no cull grid / LDS image) on
    a random two-cascade occupancy grid of H^3 cells per cascade -- inference march (ragged alive list, 8 steps) and training march
    (counts, ray records, samples) bit for bit against the oracle; H = 64 exercises a grid size the scenes do not use."""
    import raymarching
    rng = np.random.default_rng(3 + H)
    bf = (rng.random(2 * H ** 3 // 8) < 0.35).astype(np.uint8) * rng.integers(1, 256, 2 * H ** 3 // 8).astype(np.uint8)
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    ro, rd = cam["ro"] * 1.6, cam["rd"]                                 # camera pulled back: rays cross both cascades
    nears, fars = O.near_far_from_aabb(ro, rd, aabb, 0.2)
    N = ro.shape[0]
    alive = np.arange(N, dtype=np.int32)[::-1].copy()[: N - 5]
    ref = O.march_rays(alive.shape[0], 8, alive, nears.copy(), ro, rd, 2.0, bf, 2, H, nears, fars, align=128, dt_gamma=dt_gamma)
    out = raymarching.march_rays(alive.shape[0], 8, _dev(alive), _dev(nears.copy()), _dev(ro), _dev(rd), 2.0, _dev(bf), 2, H, _dev(nears), _dev(fars),
                                 128, False, dt_gamma, 1024)
    for o, r, name in zip(out, ref, ("xyzs", "dirs", "deltas")):
        assert o.shape == r.shape and np.array_equal(o.cpu().numpy().view(np.uint32), r.view(np.uint32)), name
    assert (ref[2][:, 0] > 0).sum() > 1000
    sel = rng.integers(0, N, 1024)
    counter = torch.zeros(2, dtype=torch.int32, device="cuda")
    got = raymarching.march_rays_train(_dev(ro[sel]), _dev(rd[sel]), 2.0, _dev(bf), 2, H, _dev(nears[sel]), _dev(fars[sel]), counter, -1, False, 128,
                                       False, dt_gamma, 1024)
    counter_r = np.zeros(2, np.int32)
    want = O.march_rays_train(ro[sel], rd[sel], 2.0, bf, 2, H, nears[sel], fars[sel], counter_r, -1, False, 128, False, dt_gamma, 1024)
    assert np.array_equal(counter.cpu().numpy(), counter_r) and counter_r[0] > 1000
    for o, r, name in zip(got, want, ("xyzs", "dirs", "deltas", "rays")):
        assert o.shape == r.shape and np.array_equal(o.cpu().numpy().view(np.uint32), r.view(np.uint32)), name


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_grid_encode_backward_deterministic_mode(dtype, monkeypatch):
    """SDN_DETERMINISTIC=1: `grid_encode`'s table gradient is summed in 64-bit fixed point with integer atomics (sdn_grid_encode_backward_det):
    two calls on the same inputs give the same bits -- the default path's half / float atomics round in execution order -- and the
    gradient agrees with the default path's to the atomics' own rounding (fp16: 2e-2 relative in L2, the bar of the default path's
    test against the oracle; fp32: 1e-5)."""
    from gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                      gridtype="tiled", align_corners=False).cuda()
    g = torch.Generator(device="cuda").manual_seed(3)
    # many points in few cells: every coarse row receives thousands of contributions (the order-dependent case)
    x = (torch.rand(60000, 3, device="cuda", generator=g) * 0.2 + 0.4) * 2 - 1
    gy = None

    def grad_of():
        nonlocal gy
        enc.embeddings.grad = None
        with torch.autocast("cuda", dtype=torch.float16, enabled=dtype == torch.float16):
            y = enc(x, bound=1)
        if gy is None:
            gy = torch.randn(y.shape, device="cuda", generator=g).to(y.dtype)
        y.backward(gy)
        return enc.embeddings.grad.detach().clone()

    plain = [grad_of() for _ in range(2)]
    monkeypatch.setenv("SDN_DETERMINISTIC", "1")
    det = [grad_of() for _ in range(3)]
    assert torch.equal(det[0], det[1]) and torch.equal(det[0], det[2])
    rel = float((det[0].float() - plain[0].float()).norm() / plain[0].float().norm())
    assert rel < (2e-2 if dtype == torch.float16 else 1e-5), rel
    assert float(det[0].float().abs().max()) > 0


def test_grid_encode_forward_on_the_quad_copy_is_the_plain_forward_bit_for_bit():
    """Inference under `-O` on a table that does not change between calls (gridencoder/grid.py `_quad_table_for`): from the second call
    on, the dnerf geometry's fp16 forward reads the QUAD copy of the table (two 16-byte gathers per point and level, csrc/gridencoder.hip
    k_grid_fwd_quad) -- same bits as the plain kernel on the `.half()` table and as the oracle; an in-place update of the table is seen
    (torch's version counter) and the copy rebuilt; calls that need a gradient never take it."""
    from gridencoder import GridEncoder
    from gridencoder import grid as G
    import sdn_backend
    torch.manual_seed(3)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                      gridtype="tiled", align_corners=False).cuda()
    with torch.no_grad():
        enc.embeddings.uniform_(-1.0, 1.0)
    rng = np.random.default_rng(8)
    x = torch.from_numpy(np.concatenate([rng.uniform(-1, 1, (20000, 3)), rng.uniform(-1.3, 1.3, (501, 3)),
                                         np.array([[1.0, 1.0, 1.0], [-1.0, -1.0, -1.0], [1.0, -1.0, 0.999999]])]).astype(np.float32)).cuda()
    G._QUAD_TABLES.clear()

    def run():
        launches = []
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16), sdn_backend.launch_log(launches):
            out = enc(x, bound=1)
        return out, launches

    a, _ = run()                                   # first sight of this table version: the plain kernel
    ent = next(iter(G._QUAD_TABLES.values()))
    assert ent["quad"] is None and ent["seen"] == 1
    b, _ = run()                                   # second call: the copy is built and used
    assert ent["quad"] is not None and a.dtype == torch.float16
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    c, _ = run()
    assert torch.equal(a.view(torch.int16), c.view(torch.int16))
    emb16 = enc.embeddings.detach().half().cpu().numpy()
    off = enc.offsets.cpu().numpy().astype(np.int32)
    xin = ((x + 1) / 2).cpu().numpy()
    ref, _ = O.grid_encode_forward(xin[:3000], emb16, off, float(enc.per_level_scale), enc.base_resolution, False, 1, False, 0)
    assert np.array_equal(b[:3000].cpu().numpy().view(np.uint16), ref.view(np.uint16))
    with torch.no_grad():
        enc.embeddings.mul_(0.5)                   # in place: same address, new version
    d, _ = run()
    ent = next(iter(G._QUAD_TABLES.values()))
    assert ent["quad"] is None                     # the stale copy is gone; this call ran the plain kernel on the new values
    e, _ = run()
    assert torch.equal(d.view(torch.int16), e.view(torch.int16)) and not torch.equal(d, a)
    # a call that needs the table gradient keeps the plain path (and its saved tensors)
    with torch.autocast("cuda", dtype=torch.float16):
        out = enc(x[:1000], bound=1)
    out.float().sum().backward()
    assert enc.embeddings.grad is not None and float(enc.embeddings.grad.abs().sum()) > 0


def test_empty_and_ragged_inputs_of_the_drop_in_operators():
    """Zero-size inputs and leading batch dimensions through the operator layer (the reference's wrappers allocate outputs of the same
    shapes and launch nothing: raymarching.py:40-56,300-370, grid.py:24-80, sphere_harmonics.py:14-50, freq.py:15-45)."""
    import raymarching
    from freqencoder import FreqEncoder
    from gridencoder import GridEncoder
    from shencoder import SHEncoder
    dev = "cuda"
    aabb = torch.tensor([-1., -1, -1, 1, 1, 1], device=dev)
    e0 = torch.empty(0, 3, device=dev)
    assert [tuple(t.shape) for t in raymarching.near_far_from_aabb(e0, e0, aabb, 0.2)] == [(0,), (0,)]
    assert raymarching.morton3D(torch.empty(0, 3, dtype=torch.int32, device=dev)).shape == (0,)
    assert raymarching.morton3D_invert(torch.empty(0, dtype=torch.int32, device=dev)).shape == (0, 3)
    g = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(dev)
    assert g(e0, bound=1).shape == (0, 32)
    x0 = e0.clone().requires_grad_(True)
    g(x0, bound=1).sum().backward()
    assert x0.grad.shape == (0, 3) and float(g.embeddings.grad.abs().sum()) == 0.0
    x = torch.rand(4, 5, 3, device=dev) * 2 - 1
    assert torch.equal(g(x, bound=1), g(x.view(-1, 3), bound=1).view(4, 5, 32))
    xt = (torch.rand(3, 64, device=dev) * 2 - 1).t()                     # non-contiguous [64, 3]
    assert torch.equal(g(xt, bound=1), g(xt.contiguous(), bound=1))
    sh, fq = SHEncoder(input_dim=3, degree=4).to(dev), FreqEncoder(input_dim=3, degree=10).to(dev)
    assert sh(e0).shape == (0, 16) and fq(e0).shape == (0, 63)
    d = torch.nn.functional.normalize(torch.randn(4, 5, 3, device=dev), dim=-1)
    assert torch.equal(sh(d), sh(d.view(-1, 3)).view(4, 5, 16)) and torch.equal(fq(d), fq(d.view(-1, 3)).view(4, 5, 63))
    # no ray alive: padding only, zero-filled (the reference's torch.zeros buffers); nothing to composite
    N = 5
    ro = torch.tensor([[0., 0, -3]] * N, device=dev); rd = torch.tensor([[0., 0, 1]] * N, device=dev)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    full = torch.full((128 ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
    none = torch.empty(0, dtype=torch.int32, device=dev)
    torch.empty(128, 8, device=dev).fill_(7.0)                           # (dirty the allocator's next blocks)
    xs, ds, dl = raymarching.march_rays(0, 8, none, nears.clone(), ro, rd, 1.0, full, 1, 128, nears, fars, 128, False, 0, 1024)
    assert xs.shape == (128, 3) and not xs.any() and not ds.any() and not dl.any()
    z = torch.zeros(0, device=dev)
    raymarching.composite_rays(0, 4, none, z, z, torch.zeros(0, 3, device=dev), torch.zeros(0, 2, device=dev), z, z, torch.zeros(0, 3, device=dev))
    # align = -1: no padding; align dividing M: the reference adds a whole `align` (raymarching.py:331-332)
    alive = torch.arange(N, dtype=torch.int32, device=dev)
    assert raymarching.march_rays(N, 3, alive, nears.clone(), ro, rd, 1.0, full, 1, 128, nears, fars, -1, False, 0, 1024)[0].shape == (15, 3)
    assert raymarching.march_rays(N, 3, alive, nears.clone(), ro, rd, 1.0, full, 1, 128, nears, fars, 5, False, 0, 1024)[0].shape == (20, 3)
    # a training batch without a single sample (align = -1 on an empty grid): zero images, empty gradients
    s = torch.zeros(0, device=dev, requires_grad=True)
    c = torch.zeros(0, 3, device=dev, requires_grad=True)
    rays = torch.zeros(64, 3, dtype=torch.int32, device=dev)
    rays[:, 0] = torch.arange(64, dtype=torch.int32, device=dev)
    w, dep, img = raymarching.composite_rays_train(s, c, torch.zeros(0, 2, device=dev), rays)
    assert not w.any() and not dep.any() and not img.any() and img.shape == (64, 3)
    (w.sum() + img.sum()).backward()
    assert s.grad.shape == (0,) and c.grad.shape == (0, 3)


def test_large_batches_are_batch_independent():
    """20 M points through the encoders, 24 M sample slots through the marcher: every row equals the row of a small batch holding the same
    input (no index arithmetic wraps, no tail is dropped); the table gradient of 8 M points in one launch equals the sum over four
    launches up to the order of the float atomics."""
    import raymarching
    from freqencoder import FreqEncoder
    from gridencoder import GridEncoder
    from shencoder import SHEncoder
    dev = "cuda"
    B = 20_000_007
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand(B, 3, device=dev, generator=g) * 2 - 1
    sub = torch.randint(0, B, (50_000,), device=dev, generator=g)
    sub[:3] = torch.tensor([0, B - 1, B - 2], device=dev)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(dev)
    sh, fq = SHEncoder(3, 4).to(dev), FreqEncoder(3, 10).to(dev)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1, generator=g)
        y = enc(x, bound=1)
        assert y.shape == (B, 32) and torch.equal(y[sub], enc(x[sub].contiguous(), bound=1))
        del y
        with torch.autocast("cuda", dtype=torch.float16):
            y = enc(x, bound=1)
            assert y.dtype == torch.float16 and torch.equal(y[sub], enc(x[sub].contiguous(), bound=1))
        del y
        d = torch.nn.functional.normalize(x, dim=1)
        y = sh(d)
        assert torch.equal(y[sub], sh(d[sub].contiguous()))
        del y
        y = fq(x)
        assert torch.equal(y[sub], fq(x[sub].contiguous()))
        del y, d
    xb = x[:8_000_003].contiguous()
    del x
    enc.zero_grad()
    enc(xb, bound=1).sum().backward()
    big = enc.embeddings.grad.clone()
    enc.zero_grad()
    for c in xb.split(2_000_001):
        enc(c.contiguous(), bound=1).sum().backward()
    chunks = enc.embeddings.grad
    assert float((big - chunks).abs().max()) <= 1e-4 * float(chunks.abs().max())
    del xb, big
    N = 3_000_001
    ro = torch.zeros(N, 3, device=dev); ro[:, 2] = -3; ro[:, 0] = torch.linspace(-0.9, 0.9, N, device=dev)
    rd = torch.zeros(N, 3, device=dev); rd[:, 2] = 1
    aabb = torch.tensor([-1., -1, -1, 1, 1, 1], device=dev)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    full = torch.full((128 ** 3 // 8,), 255, dtype=torch.uint8, device=dev)
    xs, ds, dl = raymarching.march_rays(N, 8, torch.arange(N, dtype=torch.int32, device=dev), nears.clone(), ro, rd, 1.0, full, 1, 128, nears,
                                        fars, 128, False, 0, 1024)
    assert xs.shape == (N * 8 + 128 - (N * 8) % 128, 3)                 # raymarching.py:331-332
    k = torch.tensor([0, N // 2, N - 1], device=dev)
    x2, d2, dl2 = raymarching.march_rays(3, 8, torch.arange(3, dtype=torch.int32, device=dev), nears[k].clone(), ro[k].contiguous(),
                                         rd[k].contiguous(), 1.0, full, 1, 128, nears[k].contiguous(), fars[k].contiguous(), 128, False, 0, 1024)
    for i, r in enumerate(k.tolist()):
        assert torch.equal(xs[r * 8:r * 8 + 8], x2[i * 8:i * 8 + 8]) and torch.equal(dl[r * 8:r * 8 + 8], dl2[i * 8:i * 8 + 8])
    assert bool((dl2[:24, 0] > 0).all())
