"""CPU suite (-m "not gpu"): pins the oracle against the golden fixtures, checks its internal
consistency (analytic gradients vs finite differences, as testing/test_hashgrid_grad.py:51-61 of the
reference prescribes for the grid encoder), and checks the host-side logic that needs no GPU."""
import os
import re

import numpy as np
import pytest

import oracle as O


# ------------------------------------------------------------------------------------------------
# golden vectors
# ------------------------------------------------------------------------------------------------
def test_oracle_sh_matches_reference_closed_forms(golden_dir):
    g = np.load(f"{golden_dir}/sh_reference_closed_form.npz")
    out, dd = O.sh_encode_forward(g["inputs"], 8, True)
    # oracle evaluates in float64 and rounds once; the fixture is float64: float32 rounding only
    np.testing.assert_allclose(out, g["outputs"], rtol=2e-7, atol=1e-7 * np.abs(g["outputs"]).max())
    ref_dd = np.concatenate([g["dx"], g["dy"], g["dz"]], axis=1)
    np.testing.assert_allclose(dd, ref_dd, rtol=2e-7, atol=1e-7 * np.abs(ref_dd).max())
    for degree in (1, 2, 3, 4, 5, 6, 7):
        o, _ = O.sh_encode_forward(g["inputs"], degree)
        np.testing.assert_allclose(o, g["outputs"][:, : degree * degree], rtol=2e-7, atol=1e-5)


def test_oracle_sh_orthonormal_on_sphere():
    # size-independent property: the basis is orthonormal over the unit sphere (Fibonacci quadrature)
    n = 200000
    k = np.arange(n) + 0.5
    z = 1 - 2 * k / n
    phi = np.pi * (1 + 5 ** 0.5) * k
    r = np.sqrt(1 - z * z)
    v = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1)
    Y, _ = O.sh_encode_forward(v, 6)
    G = (Y.astype(np.float64).T @ Y.astype(np.float64)) * (4 * np.pi / n)
    np.testing.assert_allclose(G, np.eye(36), atol=2e-3)


def test_oracle_freq_matches_reference_torch(golden_dir):
    g = np.load(f"{golden_dir}/freq_reference_torch.npz")
    for name in ("xyz", "time"):
        x, deg = g[f"{name}_inputs"], int(g[f"{name}_degree"])
        y = O.freq_encode_forward(x, deg)
        assert y.shape == g[f"{name}_outputs"].shape
        # identity + sin columns agree to float rounding; cos columns carry the reference kernel's
        # float `+ pi/2` phase add (<= 3e-5 at 2^9): inside the 1e-4 absolute bar of SURVEY 7
        np.testing.assert_allclose(y, g[f"{name}_outputs"], rtol=0, atol=1e-4)
        D = x.shape[1]
        sin_cols = [c for c in range(y.shape[1]) if c < D or ((c // D - 1) % 2 == 0)]
        np.testing.assert_allclose(y[:, sin_cols], g[f"{name}_outputs"][:, sin_cols], rtol=0, atol=2e-6)
        gi = O.freq_encode_backward(g[f"{name}_grad_outputs"], y, D, deg)
        np.testing.assert_allclose(gi, g[f"{name}_grad_inputs"], rtol=1e-4, atol=1e-4 * np.abs(g[f"{name}_grad_inputs"]).max())


def test_oracle_trunc_exp_matches_reference_torch(golden_dir):
    g = np.load(f"{golden_dir}/trunc_exp_reference_torch.npz")
    np.testing.assert_allclose(O.trunc_exp_forward(g["x"]), g["y"], rtol=2e-7)
    np.testing.assert_allclose(O.trunc_exp_backward(g["grad_y"], g["x"]), g["grad_x"], rtol=1e-6)


def test_oracle_regression_fixtures(golden_dir):
    """The committed oracle_*.npz pins: the oracle must keep producing exactly these bits."""
    from tests_support import oracle_fixture_cases
    for name, fn in oracle_fixture_cases().items():
        path = f"{golden_dir}/oracle_{name}.npz"
        assert os.path.exists(path), f"missing fixture {path}: run tests/golden/gen_oracle_fixtures.py"
        g = np.load(path)
        got = fn()
        assert set(got) == set(g.files)
        for k in got:
            a, b = np.asarray(got[k]), g[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k)
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (name, k)


# ------------------------------------------------------------------------------------------------
# half emulation
# ------------------------------------------------------------------------------------------------
def test_half_roundtrip_matches_numpy():
    lib = O.lib()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(3000).astype(np.float32) * s for s in (1e-8, 6e-8, 1e-5, 1e-3, 1, 100, 7e4)])
    xs = np.concatenate([xs, np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25, np.inf, -np.inf], np.float32)])
    for v in xs:
        assert lib.orc_f2h(float(v)) == int(np.float32(v).astype(np.float16).view(np.uint16)), v
    for h in range(0, 65536, 5):
        f = lib.orc_h2f(h)
        r = np.array([h], np.uint16).view(np.float16).astype(np.float32)[0]
        assert (f == r) or (np.isnan(f) and np.isnan(r))


# ------------------------------------------------------------------------------------------------
# grid encoder: analytic gradients vs finite differences  (testing/test_hashgrid_grad.py:51-61)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gridtype,interp,align", [(0, 0, False), (1, 0, False), (0, 1, True)])
def test_oracle_grid_gradients_vs_finite_differences(gridtype, interp, align):
    D, L, C, H, log2T = 3, 4, 2, 4, 8  # the shape the reference's gradcheck uses
    offsets, pls = O.grid_offsets(D, L, C, 2, H, log2T, None, align)
    rng = np.random.default_rng(0)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    # keep points away from cell faces of every level so central differences stay inside one cell
    x = (rng.random((64, D)) * 0.9 + 0.05).astype(np.float32)
    y, dd = O.grid_encode_forward(x, emb, offsets, pls, H, True, gridtype, align, interp)
    g = rng.standard_normal(y.shape).astype(np.float32)
    ge, gi = O.grid_encode_backward(g, x, emb, offsets, pls, H, dd, gridtype, align, interp)
    # d/d embeddings: the encoder is linear in the table -> <g, f(E + dE) - f(E)> == <ge, dE> exactly
    dE = rng.standard_normal(emb.shape).astype(np.float32) * 0.1
    y2, _ = O.grid_encode_forward(x, emb + dE, offsets, pls, H, False, gridtype, align, interp)
    lhs = float((g.astype(np.float64) * (y2.astype(np.float64) - y)).sum())
    rhs = float((ge.astype(np.float64) * dE).sum())
    assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(rhs))
    # d/d inputs: central differences, eps / tolerances of the reference's gradcheck (eps 1e-2 is too coarse for
    # the finest level's cells here, so use 1e-4 with the same atol/rtol)
    eps = 1e-4
    ok = 0
    for d in range(D):
        xp, xm = x.copy(), x.copy()
        xp[:, d] += eps
        xm[:, d] -= eps
        yp, _ = O.grid_encode_forward(xp, emb, offsets, pls, H, False, gridtype, align, interp)
        ym, _ = O.grid_encode_forward(xm, emb, offsets, pls, H, False, gridtype, align, interp)
        fd = ((yp.astype(np.float64) - ym) / (xp[:, d] - xm[:, d]).astype(np.float64)[:, None] * g).sum(1)
        good = np.abs(fd - gi[:, d]) <= 1e-3 + 1e-2 * np.abs(fd) + 2e-2 * np.abs(fd).max()
        ok += good.sum()
    # a few points straddle a cell face inside +-eps at the finest level; everything else must agree
    assert ok >= 0.95 * D * x.shape[0]


def test_oracle_grid_level_table_matches_survey():
    offsets, pls = O.grid_offsets(3, 16, 2, 2, 16, 19, 2048, False)
    assert offsets[-1] == 6119864 and abs(pls - 1.3819128) < 1e-6
    rows = np.diff(offsets)
    assert list(rows[:5]) == [4920, 13824, 32768, 85184, 216000] and (rows[5:] == 524288).all()


# ------------------------------------------------------------------------------------------------
# raymarching properties (no reference fixture exists: "parity unpinned", see tests/golden/README.md)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def small_cam():
    from dnerf_amd import scene
    H = W = 48
    ro, rd = scene.get_rays(scene.look_at_pose(), scene.intrinsics(H, W), H, W)
    bf = scene.jumpingjacks_occupancy(0.5)
    nears, fars = O.near_far_from_aabb(ro, rd, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    return ro, rd, bf, nears, fars


def test_oracle_march_samples_lie_in_occupied_cells(small_cam):
    from dnerf_amd import scene
    ro, rd, bf, nears, fars = small_cam
    xyzs, dirs, deltas, rays = O.march_rays_train(ro, rd, 1.0, bf, 1, 128, nears, fars, np.zeros(2, np.int32), -1, False, 128)
    n_pts = int(rays[:, 2].sum())
    assert n_pts > 500
    assert xyzs.shape[0] == n_pts + (128 - n_pts % 128)  # the "+align even if aligned" padding
    live = xyzs[:n_pts]
    cell = np.clip((0.5 * (live + 1) * 128).astype(np.int64), 0, 127)
    idx = scene.morton3d(cell[:, 0], cell[:, 1], cell[:, 2]).astype(np.int64)
    assert ((bf[idx // 8] >> (idx % 8)) & 1).all()
    assert np.allclose(deltas[:n_pts, 0], 2 * 3 ** 0.5 / 1024)  # dt_gamma = 0 -> constant step
    assert not xyzs[n_pts:].any()
    # offsets are an exclusive scan of the counts in ray order
    assert np.array_equal(rays[:, 0], np.arange(rays.shape[0]))
    assert np.array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:-1, 2])]))


def test_oracle_inference_equals_training_march_per_ray(small_cam):
    """Chunked inference marching (any n_step schedule) visits exactly the samples of the one-shot
    training march: the property that makes the render independent of the compaction schedule."""
    ro, rd, bf, nears, fars = small_cam
    N = ro.shape[0]
    xyzs, dirs, deltas, rays = O.march_rays_train(ro, rd, 1.0, bf, 1, 128, nears, fars, np.zeros(2, np.int32), -1, False, 128)
    per_ray = {int(r[0]): xyzs[r[1]: r[1] + r[2]] for r in rays}
    got = {i: [] for i in range(N)}
    alive = np.arange(N, dtype=np.int32)
    t = nears.copy()
    for n_step in (1, 3, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8):
        if alive.shape[0] == 0:
            break
        x, d, l = O.march_rays(alive.shape[0], n_step, alive, t, ro, rd, 1.0, bf, 1, 128, nears, fars, align=128)
        sig = np.zeros(x.shape[0], np.float32)  # transparent medium: nothing terminates early
        rgb = np.zeros((x.shape[0], 3), np.float32)
        ws, dp, im = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
        for k, rid in enumerate(alive):
            seg = l[k * n_step:(k + 1) * n_step, 0] > 0
            got[int(rid)].append(x[k * n_step:(k + 1) * n_step][seg])
        O.composite_rays(alive.shape[0], n_step, alive, t, sig, rgb, l, ws, dp, im, 1e-2)
        alive = alive[alive >= 0]
    assert alive.shape[0] == 0
    for i in range(N):
        a = np.concatenate(got[i]) if got[i] else np.zeros((0, 3), np.float32)
        assert np.array_equal(a.view(np.uint32), per_ray[i].view(np.uint32)), i


def test_oracle_composite_train_matches_closed_form():
    # constant sigma / colour along one ray: T = exp(-sigma * sum dt), image = c (1 - T)
    n = 50
    dt = np.float32(2 * 3 ** 0.5 / 1024)
    sig = np.full(n, 30.0, np.float32)
    rgb = np.tile(np.array([[0.2, 0.5, 0.9]], np.float32), (n, 1))
    deltas = np.full((n, 2), dt, np.float32)
    rays = np.array([[0, 0, n]], np.int32)
    ws, depth, image = O.composite_rays_train_forward(sig, rgb, deltas, rays, 1e-10)
    T = np.exp(-30.0 * dt * n)
    np.testing.assert_allclose(ws[0], 1 - T, rtol=1e-5)
    np.testing.assert_allclose(image[0], np.array([0.2, 0.5, 0.9]) * (1 - T), rtol=1e-5)
    # gradient wrt sigma vs finite differences (fp32 forward -> loose)
    g_ws, g_im = np.ones(1, np.float32), np.ones((1, 3), np.float32)
    gs, gc = O.composite_rays_train_backward(g_ws, g_im, sig, rgb, deltas, rays, ws, image, 1e-10)
    k, eps = 7, 0.5
    sp, sm = sig.copy(), sig.copy()
    sp[k] += eps
    sm[k] -= eps
    fp = O.composite_rays_train_forward(sp, rgb, deltas, rays, 1e-10)
    fm = O.composite_rays_train_forward(sm, rgb, deltas, rays, 1e-10)
    fd = ((fp[0] - fm[0]).sum() + (fp[2] - fm[2]).sum()) / (2 * eps)
    np.testing.assert_allclose(gs[k], fd, rtol=2e-2)
    np.testing.assert_allclose(gc, np.repeat((ws_weights(sig, deltas))[:, None], 3, 1), rtol=1e-5)


def ws_weights(sig, deltas):
    alpha = 1 - np.exp(-(sig.astype(np.float64) * deltas[:, 0]))
    T = np.concatenate([[1.0], np.cumprod(1 - alpha)[:-1]])
    return (alpha * T).astype(np.float32)


# ------------------------------------------------------------------------------------------------
# host-side logic and the C ABI surface (no compute calls: there is no GPU here)
# ------------------------------------------------------------------------------------------------
def test_c_abi_exports_every_declared_symbol():
    import ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "sdn_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = sorted(set(re.findall(r"\b(sdn_[a-zA-Z0-9_]+)\s*\(", header)))
    assert len(names) >= 20
    lib = ctypes.CDLL(os.path.join(root, "seald-nerf_amd", "lib", "libsdn_hip.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    import sdn_backend
    declared = (set(sdn_backend.PROTOTYPES) | set(sdn_backend.PROTOTYPES_U64) | set(sdn_backend.PROTOTYPES_U32)
                | {"sdn_version", "sdn_host_mailbox_alloc", "sdn_field_select_kernel", "sdn_field_persistent_workgroups"})
    assert set(names) == declared, sorted(set(names) ^ declared)


def test_ctypes_records_match_the_c_header(tmp_path):
    """The Python mirrors of the C records (sdn_backend.py) against the header itself: a C program compiled from include/sdn_hip.h
    prints sizeof and the offset of every field; a record whose mirror drifts would hand the library misplaced pointers."""
    import ctypes
    import subprocess
    import sdn_backend as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    records = {"SdnFrameTime": B.SdnFrameTime, "SdnRenderCtx": B.SdnRenderCtx, "SdnSealBox": B.SdnSealBox, "SdnTrainParam": B.SdnTrainParam,
               "SdnTrainStep": B.SdnTrainStep, "SdnTrainLayout": B.SdnTrainLayout}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sdn_hip.h"', 'int main(void) {']
    for name, rec in records.items():
        lines.append(f'  printf("{name} size %zu\\n", sizeof({name}));')
        for field, _ in rec._fields_:
            lines.append(f'  printf("{name} {field} %zu\\n", offsetof({name}, {field}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for line in filter(None, out):
        name, field, value = line.split()
        rec = records[name]
        want = ctypes.sizeof(rec) if field == "size" else getattr(rec, field).offset
        assert int(value) == want, (name, field, int(value), want)
        seen += 1
    assert seen == sum(len(r._fields_) + 1 for r in records.values())


def test_ops_fail_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for the CPU-only container")
    import raymarching
    from sdn_backend import SdnError
    with pytest.raises(SdnError):
        raymarching.near_far_from_aabb(torch.zeros(4, 3), torch.ones(4, 3), torch.tensor([-1., -1, -1, 1, 1, 1]), 0.2)
    from gridencoder import GridEncoder
    enc = GridEncoder(num_levels=2, log2_hashmap_size=8)
    with pytest.raises(SdnError):
        enc(torch.zeros(4, 3))


def test_scene_rays_and_bitfield_are_deterministic():
    from dnerf_amd import scene
    a = scene.jumpingjacks_occupancy(0.25, 64)
    b = scene.jumpingjacks_occupancy(0.25, 64)
    assert np.array_equal(a, b) and a.dtype == np.uint8 and a.shape == (64 ** 3 // 8,)
    assert 0 < np.unpackbits(a).mean() < 0.05
    ro, rd = scene.get_rays(scene.look_at_pose(), scene.intrinsics(8, 8), 8, 8)
    assert ro.shape == (64, 3) and np.allclose(np.linalg.norm(rd, axis=1), 1, atol=1e-6)
    assert np.allclose(np.linalg.norm(ro[0]), scene.CAMERA_RADIUS, atol=1e-5)


def test_training_step_layout_and_argument_checks_on_the_host():
    """`sdn_train_layout` is host arithmetic: every buffer 256-byte aligned, inside the block, the two sample sets one stride apart,
    the block growing with the batch; a record that cannot be run is refused before anything is launched (no GPU here)."""
    import ctypes
    import sdn_backend as B
    off = (ctypes.c_int32 * 17)()
    rows, res = 0, [int(np.ceil(16 * np.exp2(np.log2(2048 / 16) / 15) ** i)) for i in range(16)]
    for i, r in enumerate(res):
        off[i] = rows
        rows += int(np.ceil(min(2 ** 19, (r + 1) ** 3) / 8) * 8)
    off[16] = rows
    assert rows == 6119864                                          # the dnerf grid of dnerf/network.py:55-60 (SURVEY level table)
    small, big = B.SdnTrainLayout(), B.SdnTrainLayout()
    assert B.lib.sdn_train_layout(4096, 9216, 1024, off, ctypes.byref(small)) == 0
    assert B.lib.sdn_train_layout(8192, 18432, 1024, off, ctypes.byref(big)) == 0
    fields = [f for f, _ in B.SdnTrainLayout._fields_ if f not in ("total_bytes", "sample_set_stride", "found_inf", "dirs", "deltas")]
    for f in fields:
        v = getattr(small, f)
        assert v % 256 == 0 and v < small.total_bytes, f
    assert small.total_bytes < big.total_bytes and small.sample_set_stride % 256 == 0
    assert small.xyzs + small.sample_set_stride < small.total_bytes
    assert small.g_table - small.w_table >= rows * 2 * 2          # the fp16 table copy fits in front of the next persistent buffer
    assert B.lib.sdn_train_layout(0, 9216, 1024, off, ctypes.byref(small)) == -1          # SDN_E_BADARG
    rec = B.SdnTrainStep()
    assert B.lib.sdn_train_step_f16(ctypes.byref(rec), None) == -1
    assert B.lib.sdn_train_refresh(ctypes.byref(rec), None) == -1
