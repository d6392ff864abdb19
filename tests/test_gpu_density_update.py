"""Density-grid maintenance on the device (csrc/density.hip + the CELLS variant of the fused field kernel) against the
renderer mirror's update_extra_state, which restates dnerf/renderer.py:453-555 op by op on the HIP operators.

Parity is pinned at two levels: the in-kernel cell -> point construction and density query are BIT-EXACT against the fused
field kernel fed with points built by the reference's torch expressions; the whole update (fp16 `-O` numerics) agrees with the
op-by-op mirror within the fused-vs-unfused tolerance of test_gpu_render_parity.py (median 2e-3, max 5e-2 relative)."""
import numpy as np
import pytest
import torch

import tests_support  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    from dnerf_amd.bench_scene import build_scene
    return build_scene(H=8, W=8, device="cuda", seed=0)


def _points(model, cells, noise, cas_bound=1.0):
    """dnerf/renderer.py:480-490 in torch, for Morton indices `cells`."""
    import raymarching
    coords = raymarching.morton3D_invert(cells.to(torch.int32))
    xyzs = 2 * coords.float() / (model.grid_size - 1) - 1
    half_grid = cas_bound / model.grid_size
    cas = xyzs * (cas_bound - half_grid)
    cas += (noise * 2 - 1) * half_grid
    return cas


def test_query_cells_is_bit_exact_against_the_field_kernel(scene):
    from dnerf_amd import fused
    m = scene.model
    up = fused.DensityGridUpdater(m)
    f = up.field
    f.set_time(torch.tensor([[0.37]], device="cuda"))
    H3 = m.grid_size ** 3
    g = torch.Generator(device="cuda").manual_seed(3)
    noise = torch.rand(H3, 3, device="cuda", generator=g)
    out = torch.full((H3,), -1.0, device="cuda")
    up.query_cells(out, f.bias0, f.zero_deform, 1.0, n=H3, noise=noise)
    cells = torch.arange(H3, device="cuda", dtype=torch.int32)
    x = _points(m, cells, noise).contiguous()
    s, _ = f(x, torch.zeros_like(x))
    assert torch.equal(out, s), int((out != s).sum())
    assert float(out.min()) > 0

    # listed cells with the live count on the device: the tail of the list and every unlisted cell stay untouched
    n_list, n_live = 50000, 41234
    perm = torch.randperm(H3, device="cuda", generator=g)[:n_list].to(torch.int32).contiguous()
    count = torch.tensor([n_live], dtype=torch.int32, device="cuda")
    nz = torch.rand(n_list, 3, device="cuda", generator=g)
    out2 = torch.full((H3,), -1.0, device="cuda")
    up.query_cells(out2, f.bias0, f.zero_deform, 1.0, cells=perm, cell_count=count, noise=nz)
    x2 = _points(m, perm[:n_live], nz[:n_live]).contiguous()
    s2, _ = f(x2, torch.zeros_like(x2))
    live = perm[:n_live].long()
    assert torch.equal(out2[live], s2)
    rest = torch.ones(H3, dtype=torch.bool, device="cuda"); rest[live] = False
    assert bool((out2[rest] == -1).all())

    # in-kernel generator: jitter stays inside the cell, so sigma stays near the noise-free value; different seeds differ
    a = torch.empty(H3, device="cuda"); b = torch.empty(H3, device="cuda")
    up.query_cells(a, f.bias0, f.zero_deform, 1.0, n=H3, seed=1)
    up.query_cells(b, f.bias0, f.zero_deform, 1.0, n=H3, seed=2)
    assert bool(torch.isfinite(a).all()) and not torch.equal(a, b)
    lo = torch.empty(H3, device="cuda"); hi = torch.empty(H3, device="cuda")
    up.query_cells(lo, f.bias0, f.zero_deform, 1.0, n=H3, noise=torch.zeros(H3, 3, device="cuda"))
    up.query_cells(hi, f.bias0, f.zero_deform, 1.0, n=H3, noise=torch.ones(H3, 3, device="cuda"))
    assert not torch.equal(lo, hi)

    # argument checks
    import sdn_backend
    with pytest.raises(sdn_backend.SdnError):
        up.query_cells(out, f.bias0, 0, 1.0, cells=perm, cell_count=None)
    with pytest.raises(sdn_backend.SdnError):
        up.query_cells(out, f.bias0, 0, 1.0, n=H3 + 1)


def test_ema_and_pack_kernels_exact():
    import raymarching
    import sdn_backend
    from sdn_backend import check, ptr, stream
    g = torch.Generator(device="cuda").manual_seed(0)
    n = 64 * 4096
    grid = torch.rand(n, device="cuda", generator=g) * 4 - 1        # a quarter negative (= untrained cells)
    tmp = torch.rand(n, device="cuda", generator=g) * 4 - 1
    tmp[::97] = float("nan")
    ref = grid.clone()
    valid = (ref >= 0) & (tmp >= 0)
    ref[valid] = torch.maximum(ref[valid] * 0.95, tmp[valid])
    total = torch.zeros(1, dtype=torch.float64, device="cuda")
    check(sdn_backend.lib.sdn_density_grid_ema(ptr(grid), ptr(tmp), n, 0.95, ptr(total), stream()), "ema")
    assert torch.equal(grid, ref)
    want = ref.clamp(min=0).double().sum()
    assert abs(float(total[0]) - float(want)) <= 1e-9 * float(want)
    mean = torch.zeros(2, device="cuda")
    bits = torch.zeros(n // 8, dtype=torch.uint8, device="cuda")
    for cap in (10.0, 0.3):                                          # mean below / above density_thresh
        check(sdn_backend.lib.sdn_density_grid_pack(ptr(grid), n, ptr(total), cap, ptr(mean), ptr(bits), stream()), "pack")
        m = float(want) / n
        assert abs(float(mean[0]) - m) < 1e-6 * m and float(mean[1]) == min(float(mean[0]), np.float32(cap))
        assert torch.equal(bits, raymarching.packbits(grid.view(1, -1), float(mean[1])))
    with pytest.raises(sdn_backend.SdnError):
        check(sdn_backend.lib.sdn_density_grid_ema(ptr(grid), ptr(tmp), n - 1, 0.95, ptr(total), stream()), "ema")


def _fresh_model(scene):
    from dnerf_amd.bench_scene import build_model
    m = build_model(0, "cuda")
    m.load_state_dict(scene.model.state_dict())
    m.reset_extra_state()
    return m


def test_full_update_matches_the_op_by_op_mirror(scene):
    """Same torch.rand draws on both sides: the mirror draws rand_like(points) then rand_like(time) per slice."""
    a, b = _fresh_model(scene), _fresh_model(scene)
    T, H3 = a.time_size, a.grid_size ** 3
    torch.manual_seed(11)
    with torch.autocast("cuda", dtype=torch.float16):
        a.update_extra_state()
    torch.manual_seed(11)
    noise = torch.empty(T, 1, H3, 3, device="cuda")
    tn = torch.empty(T, 1)
    coords_order = None
    for t in range(T):
        noise[t, 0] = torch.rand(H3, 3, device="cuda")
        tn[t, 0] = float(torch.rand(1, 1, device="cuda"))
    # the mirror enumerates cells in meshgrid order, the kernel in Morton order: permute the draws
    import raymarching
    ax = torch.arange(a.grid_size, dtype=torch.int32, device="cuda")
    xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing="ij")
    morton_of_mesh = raymarching.morton3D(torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1).contiguous()).long()
    by_cell = torch.empty_like(noise)
    by_cell[:, :, morton_of_mesh] = noise
    up = b.use_native_density_update()
    mean = up.update(0.95, noise=by_cell, time_noise=tn)
    assert a.iter_density == b.iter_density == 1
    da, db = a.density_grid, b.density_grid
    rel = (da - db).abs() / da.abs().clamp(min=1e-3)
    assert float(rel.median()) < 2e-3
    # A jittered centre of an outermost cell can sit within an fp16 rounding of the deformation from the grid's [0,1] domain
    # boundary, where the encoder switches to zero features (sigma = exp(0) = 1): a handful of such cells may land on different
    # sides in the fused and the op-by-op network.  Everything else agrees to the fused-vs-unfused tolerance.
    off = (rel >= 5e-2).nonzero()
    assert off.shape[0] <= 16, off.shape[0]
    if off.shape[0]:
        c = raymarching.morton3D_invert(off[:, 2].to(torch.int32).contiguous())
        assert bool(((c == 0) | (c == a.grid_size - 1)).any(dim=1).all())
        assert bool(((da[rel >= 5e-2] == 1) | (db[rel >= 5e-2] == 1)).all())
    assert abs(float(mean[0]) - a.mean_density) < 2e-3 * a.mean_density
    thresh = min(a.mean_density, a.density_thresh)
    diff = (a.density_bitfield ^ b.density_bitfield)
    if int(diff.count_nonzero()):
        # a bit may differ only for a cell whose density sits within the numeric tolerance of the threshold
        bit = torch.arange(8, device="cuda", dtype=torch.uint8)
        cells = ((diff.view(T, -1, 1) >> bit) & 1).bool().view(T, 1, H3)
        near = cells & (rel < 5e-2)                                   # (the boundary cells above are already accounted for)
        assert float(((da[near] - thresh).abs() / thresh).max()) < 6e-2
        assert int(cells.sum()) < 2e-2 * cells.numel()
    # the bitfield is exactly packbits of the native grid at the native threshold
    assert torch.equal(b.density_bitfield, torch.stack([raymarching.packbits(db[t], float(mean[1])) for t in range(T)]))


def test_partial_update_invariants(scene):
    import raymarching
    m = _fresh_model(scene)
    up = m.use_native_density_update()
    m.update_extra_state()                       # full pass through the hook (in-kernel noise): every cell now > 0
    assert m.iter_density == 1 and float(m.density_grid.min()) > 0
    m.density_grid[:, :, ::3] = 0                # make "occupied" a proper subset
    m.density_grid[5] = 0                        # and one slice with nothing occupied at all
    m.iter_density = 16
    before = m.density_grid.clone()
    torch.manual_seed(5)
    cells, counts = up.partial_cells()
    N = m.grid_size ** 3 // 4
    assert cells.shape == (m.time_size, 1, 2 * N) and cells.dtype == torch.int32
    assert int(counts[5, 0]) == N and bool((counts[torch.arange(m.time_size) != 5] == 2 * N).all())
    H3 = m.grid_size ** 3
    for t in (5, 9):
        live = cells[t, 0, :int(counts[t, 0])].long()
        assert 0 <= int(live.min()) and int(live.max()) < H3 and bool((live[1:] >= live[:-1]).all())   # valid, sorted by Morton index
    # N uniform draws over all cells (2/3 of them occupied here) + N draws among the occupied ones
    live = cells[9, 0].long()
    assert abs(float((before[9, 0][live] > 0).float().mean()) - (2 / 3 + 1) / 2) < 0.01
    assert abs(float(live.float().mean()) / H3 - 0.5) < 0.01
    assert bool((cells[5, 0, N:] == 0x7FFFFFFF).all())                     # nothing occupied in slice 5: sentinels past the live count
    torch.manual_seed(5)                         # same lists inside update()
    mean = up.update(0.95)
    after = m.density_grid
    for t in (0, 5, 9, 63):
        n_live = int(counts[t, 0])
        touched = torch.zeros(m.grid_size ** 3, dtype=torch.bool, device="cuda")
        touched[cells[t, 0, :n_live].long()] = True
        assert torch.equal(after[t, 0][~touched], before[t, 0][~touched])  # tmp = -1 there: no decay either
        assert bool((after[t, 0][touched] >= before[t, 0][touched] * 0.95).all())
        assert bool((after[t, 0][touched] > 0).all())
    want = after.clamp(min=0).double().mean()
    assert abs(float(mean[0]) - float(want)) < 1e-6 * float(want)
    assert abs(m.mean_density - float(want)) < 1e-6 * float(want)
    assert torch.equal(m.density_bitfield, torch.stack([raymarching.packbits(after[t], float(mean[1])) for t in range(m.time_size)]))
    m.iter_density = 100                         # no network queries any more: grid unchanged, bitfield re-packed
    frozen = after.clone()
    m.update_extra_state()
    assert torch.equal(m.density_grid, frozen) and m.iter_density == 101


def test_query_cells_against_the_cpu_oracle(scene):
    """The cell query tied to the CPU oracle (oracle/field.py, fp16 mode): cell -> point in numpy with the reference's fp32 operation order
    (dnerf/renderer.py:480-490), density through the oracle's field network, against the kernel's tmp_grid entries."""
    from oracle import render as orender
    from oracle.field import FieldOracle
    from dnerf_amd import fused
    m = scene.model
    up = fused.DensityGridUpdater(m)
    f = up.field
    t = 0.37
    f.set_time(torch.tensor([[t]], device="cuda"))
    H3, G = m.grid_size ** 3, m.grid_size
    rng = np.random.default_rng(5)
    cells = np.sort(rng.choice(H3, 4096, replace=False)).astype(np.int32)
    noise = rng.random((4096, 3), dtype=np.float32)
    out = torch.full((H3,), -1.0, device="cuda")
    cnt = torch.tensor([4096], dtype=torch.int32, device="cuda")
    up.query_cells(out, f.bias0, f.zero_deform, 1.0, cells=torch.from_numpy(cells).cuda(), cell_count=cnt, noise=torch.from_numpy(noise).cuda())
    got = out[torch.from_numpy(cells).long().cuda()].cpu().numpy()

    def compact(v):        # Morton code -> coordinate (raymarching.cu:282-289)
        v = v & 0x49249249
        v = (v | (v >> 2)) & 0xc30c30c3
        v = (v | (v >> 4)) & 0x0f00f00f
        v = (v | (v >> 8)) & 0xff0000ff
        return (v | (v >> 16)) & 0x0000ffff
    c = cells.astype(np.uint32)
    coords = np.stack([compact(c), compact(c >> 1), compact(c >> 2)], axis=1).astype(np.float32)
    f32 = np.float32
    half_grid = f32(1.0 / G)
    pts = ((f32(2) * coords) * (f32(1) / f32(G - 1)) - f32(1)) * f32(1.0 - 1.0 / G) + (noise * f32(2) - f32(1)) * half_grid
    dirs = np.tile(np.array([[0.0, 0.0, 1.0]], dtype=np.float32), (4096, 1))
    sigma, _, _ = FieldOracle(orender.state_of(m), mode="fp16").forward(pts.astype(np.float32), dirs, t)
    want = sigma * m.density_scale
    rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
    assert np.median(rel) < 2e-3 and rel.max() < 5e-2, (np.median(rel), rel.max())


# ---------------------------------------------------------------------------------------------------------------------------------
# the fp32 twin (a model trained WITHOUT -O): sdn_density_query_cells_f32 = the CELLS variant of the fp32 fused kernel
# ---------------------------------------------------------------------------------------------------------------------------------
def test_fp32_query_cells_is_bit_exact_against_the_fp32_field_kernel(scene):
    from dnerf_amd import fused
    m = scene.model
    up = fused.DensityGridUpdater(m, fp32=True)
    f = up.field
    assert up.fp32 and f.variant == "mfma32"
    f.set_time(torch.tensor([[0.37]], device="cuda"))
    H3 = m.grid_size ** 3
    g = torch.Generator(device="cuda").manual_seed(5)
    noise = torch.rand(H3, 3, device="cuda", generator=g)
    out = torch.full((H3,), -1.0, device="cuda")
    up.query_cells(out, f.bias0, f.zero_deform, 1.0, n=H3, noise=noise)
    cells = torch.arange(H3, device="cuda", dtype=torch.int32)
    x = _points(m, cells, noise).contiguous()
    d = torch.zeros_like(x); d[:, 2] = 1
    s, _ = f(x, d)
    assert torch.equal(out, s), int((out != s).sum())
    assert float(out.min()) > 0

    # listed cells, live count on the device, canonical frame (t = 0: no deformation)
    f.set_time(torch.tensor([[0.0]], device="cuda"))
    assert f.zero_deform == 1
    n_list, n_live = 30000, 21234
    perm = torch.randperm(H3, device="cuda", generator=g)[:n_list].to(torch.int32).contiguous()
    count = torch.tensor([n_live], dtype=torch.int32, device="cuda")
    nz = torch.rand(n_list, 3, device="cuda", generator=g)
    out2 = torch.full((H3,), -1.0, device="cuda")
    up.query_cells(out2, f.bias0, f.zero_deform, 1.0, cells=perm, cell_count=count, noise=nz)
    x2 = _points(m, perm[:n_live], nz[:n_live]).contiguous()
    d2 = torch.zeros_like(x2); d2[:, 2] = 1
    s2, _ = f(x2, d2)
    live = perm[:n_live].long()
    assert torch.equal(out2[live], s2)
    rest = torch.ones(H3, dtype=torch.bool, device="cuda"); rest[live] = False
    assert bool((out2[rest] == -1).all())

    # against the op-by-op fp32 network of the mirror (dnerf/network.py:171-206): the north star's 1e-4
    tt = torch.tensor([[0.37]], device="cuda")
    f.set_time(tt)
    up.query_cells(out, f.bias0, f.zero_deform, 1.0, n=H3, noise=noise)
    sub = torch.randperm(H3, device="cuda", generator=g)[:20000]
    with torch.no_grad():
        ref = m.density(x[sub].contiguous(), tt)["sigma"].float() * m.density_scale
    np.testing.assert_allclose(out[sub].cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-6)

    import sdn_backend
    with pytest.raises(sdn_backend.SdnError):
        up.query_cells(out, f.bias0, 0, 1.0, n=H3 + 1)


def test_fp32_full_update_matches_the_op_by_op_mirror(scene):
    """The whole update without autocast: the native fp32 path against the mirror's op-by-op fp32 update, same torch.rand draws."""
    import raymarching
    a, b = _fresh_model(scene), _fresh_model(scene)
    T, H3 = a.time_size, a.grid_size ** 3
    torch.manual_seed(13)
    a.update_extra_state()
    torch.manual_seed(13)
    noise = torch.empty(T, 1, H3, 3, device="cuda")
    tn = torch.empty(T, 1)
    for t in range(T):
        noise[t, 0] = torch.rand(H3, 3, device="cuda")
        tn[t, 0] = float(torch.rand(1, 1, device="cuda"))
    ax = torch.arange(a.grid_size, dtype=torch.int32, device="cuda")
    xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing="ij")
    morton_of_mesh = raymarching.morton3D(torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1).contiguous()).long()
    by_cell = torch.empty_like(noise)
    by_cell[:, :, morton_of_mesh] = noise
    up = b.use_native_density_update(fp32=True)
    assert up.fp32
    mean = up.update(0.95, noise=by_cell, time_noise=tn)
    da, db = a.density_grid, b.density_grid
    rel = (da - db).abs() / da.abs().clamp(min=1e-3)
    # fp32 both sides: 1e-4, except cells whose jittered centre lies within rounding of the grid's [0,1] boundary (zero features there)
    stats = {"median": float(rel.median()), "frac_ge_1e-4": float((rel >= 1e-4).float().mean()), "n_ge_1e-3": int((rel >= 1e-3).sum()),
             "max": float(rel.max()), "max_abs": float((da - db).abs().max())}
    print("fp32 density update, native vs op by op:", stats)
    off = (rel >= 1e-3).nonzero()
    assert stats["median"] < 1e-5 and stats["frac_ge_1e-4"] < 1e-4 and off.shape[0] <= 16, stats
    if off.shape[0]:
        c = raymarching.morton3D_invert(off[:, 2].to(torch.int32).contiguous())
        assert bool(((c == 0) | (c == a.grid_size - 1)).any(dim=1).all()), stats
    assert abs(float(mean[0]) - a.mean_density) < 1e-4 * a.mean_density
    diff = (a.density_bitfield ^ b.density_bitfield)
    assert int(diff.count_nonzero()) <= 16
    assert torch.equal(b.density_bitfield, torch.stack([raymarching.packbits(db[t], float(mean[1])) for t in range(T)]))
