"""Shared helpers for the test-suite: the seeded oracle cases behind tests/golden/oracle_*.npz."""
import numpy as np

import oracle as O


def _cam(H=24):
    from dnerf_amd import scene
    ro, rd = scene.get_rays(scene.look_at_pose(), scene.intrinsics(H, H), H, H)
    bf = scene.jumpingjacks_occupancy(0.5)
    nears, fars = O.near_far_from_aabb(ro, rd, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    return ro, rd, bf, nears, fars


def _case_march_train():
    ro, rd, bf, nears, fars = _cam()
    noises = np.random.default_rng(3).random(ro.shape[0], dtype=np.float32)
    counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = O.march_rays_train(ro, rd, 1.0, bf, 1, 128, nears, fars, counter, -1, True, 128, noises=noises)
    return dict(nears=nears, fars=fars, xyzs=xyzs, deltas=deltas, rays=rays, counter=counter)


def _case_infer_loop():
    ro, rd, bf, nears, fars = _cam()
    N = ro.shape[0]
    ws, dp, im = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    alive = np.arange(N, dtype=np.int32)
    t = nears.copy()
    trace = []
    step = 0
    while step < 1024 and alive.shape[0] > 0:
        n_alive = alive.shape[0]
        n_step = max(min(N // n_alive, 8), 1)
        x, d, l = O.march_rays(n_alive, n_step, alive, t, ro, rd, 1.0, bf, 1, 128, nears, fars, align=128)
        rng = np.random.default_rng(step)
        sig = (rng.random(x.shape[0], dtype=np.float32) * 40).astype(np.float32)
        rgb = rng.random((x.shape[0], 3), dtype=np.float32)
        O.composite_rays(n_alive, n_step, alive, t, sig, rgb, l, ws, dp, im, 1e-2)
        trace.append((n_alive, n_step, x.shape[0], int((l[:, 0] > 0).sum())))
        alive = alive[alive >= 0]
        step += n_step
    return dict(weights_sum=ws, depth=dp, image=im, trace=np.array(trace, np.int32))


def _case_grid(half):
    D, L, C, H = 3, 6, 2, 8
    offsets, pls = O.grid_offsets(D, L, C, 2, H, 11, 128, False)
    rng = np.random.default_rng(2)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float16 if half else np.float32)
    x = rng.random((97, D), dtype=np.float32)
    out = {}
    for gridtype in (0, 1):
        y, dd = O.grid_encode_forward(x, emb, offsets, pls, H, True, gridtype, False, 0)
        g = (rng.standard_normal(y.shape) * 0.01).astype(emb.dtype)
        ge, gi = O.grid_encode_backward(g, x, emb, offsets, pls, H, dd, gridtype, False, 0)
        out.update({f"y{gridtype}": y, f"dydx{gridtype}": dd, f"ge{gridtype}": ge, f"gi{gridtype}": gi})
    return out


def oracle_fixture_cases():
    return {
        "march_train": _case_march_train,
        "infer_loop": _case_infer_loop,
        "grid_f32": lambda: _case_grid(False),
        "grid_f16": lambda: _case_grid(True),
    }


def dist_stats(got, want, rel_floor=None):
    """Distribution of |got - want| (or of the relative error with denominator max(|want|, rel_floor)): max, p99.9, p99, mean, the
    count above 1e-3 and the size -- so a bar states more than one `max <`, and a failure message shows where the mass sits."""
    got, want = np.asarray(got, np.float64).ravel(), np.asarray(want, np.float64).ravel()
    err = np.abs(got - want)
    if rel_floor is not None:
        err = err / np.maximum(np.abs(want), rel_floor)
    return {"max": float(err.max()), "p99.9": float(np.quantile(err, 0.999)), "p99": float(np.quantile(err, 0.99)), "mean": float(err.mean()),
            "above_1e-3": int((err > 1e-3).sum()), "n": int(err.size)}


def assert_dist(got, want, what, rel_floor=None, **bars):
    """bars: any of max / p999 / p99 / mean / frac_above_1e3 (fraction of elements whose error exceeds 1e-3)."""
    st = dist_stats(got, want, rel_floor)
    key = {"max": "max", "p999": "p99.9", "p99": "p99", "mean": "mean"}
    bad = [k for k, v in bars.items() if (st["above_1e-3"] / st["n"] if k == "frac_above_1e3" else st[key[k]]) > v]
    assert not bad, (what, "exceeds", bad, "bars", bars, "measured", st)
    return st
