"""numpy front-end of the CPU oracle (test infrastructure only).

Every function mirrors one operator of the reference's L1 API and applies the
same Python-side shape / padding logic as the reference wrapper it cites, then
calls the C restatement in ``sdn_oracle.c`` (or evaluates in numpy where noted).
Paths cited are relative to /root/reference.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u32 = ctypes.c_uint32
f32 = ctypes.c_float
i32 = ctypes.c_int32


def build():
    """Compile libsdn_oracle.so with the committed Makefile (gcc, a second or two)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libsdn_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsdn_oracle.so")
        src = os.path.join(_HERE, "sdn_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_h2f.restype = ctypes.c_float
        _LIB.orc_h2f.argtypes = [ctypes.c_uint16]
        _LIB.orc_f2h.restype = ctypes.c_uint16
        _LIB.orc_f2h.argtypes = [ctypes.c_float]
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------
# raymarching utils
# --------------------------------------------------------------------------
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """raymarching/raymarching.py:22-47 + raymarching.cu:92-145."""
    rays_o = _f(rays_o).reshape(-1, 3)
    rays_d = _f(rays_d).reshape(-1, 3)
    aabb = _f(aabb)
    N = rays_o.shape[0]
    nears = np.empty(N, np.float32)
    fars = np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    """raymarching.py:55-78 + raymarching.cu:163-198."""
    rays_o = _f(rays_o).reshape(-1, 3)
    rays_d = _f(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().orc_sph_from_ray(_p(rays_o), _p(rays_d), f32(radius), u32(N), _p(coords))
    return coords


def morton3D(coords):
    """raymarching.py:85-102 + raymarching.cu:214-226."""
    coords = _i(coords)
    N = coords.shape[0]
    out = np.empty(N, np.int32)
    lib().orc_morton3D(_p(coords), u32(N), _p(out))
    return out


def morton3D_invert(indices):
    """raymarching.py:108-124 + raymarching.cu:237-254."""
    indices = _i(indices)
    N = indices.shape[0]
    out = np.empty((N, 3), np.int32)
    lib().orc_morton3D_invert(_p(indices), u32(N), _p(out))
    return out


def packbits(grid, thresh, bitfield=None):
    """raymarching.py:132-153 + raymarching.cu:268-289.  grid: [C, H^3]."""
    grid = _f(grid)
    N = grid.shape[0] * grid.shape[1] // 8
    if bitfield is None:
        bitfield = np.empty(N, np.uint8)
    lib().orc_packbits(_p(grid), u32(N), f32(thresh), _p(bitfield))
    return bitfield


# --------------------------------------------------------------------------
# train
# --------------------------------------------------------------------------
def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None,
                     mean_count=-1, perturb=False, align=-1, force_all_rays=False, dt_gamma=0,
                     max_steps=1024, noises=None):
    """raymarching.py:164-233 (buffer sizing, the '+align even when aligned' padding,
    the counter read-back) + raymarching.cu:312-480.  ``noises`` may be passed to
    make perturb reproducible; otherwise numpy's default_rng(0) is used."""
    rays_o = _f(rays_o).reshape(-1, 3)
    rays_d = _f(rays_d).reshape(-1, 3)
    density_bitfield = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    nears = _f(nears)
    fars = _f(fars)
    N = rays_o.shape[0]
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.empty((N, 3), np.int32)
    if step_counter is None:
        step_counter = np.zeros(2, np.int32)
    if noises is None:
        noises = np.random.default_rng(0).random(N, dtype=np.float32) if perturb else np.zeros(N, np.float32)
    noises = _f(noises)
    lib().orc_march_rays_train(_p(rays_o), _p(rays_d), _p(density_bitfield), f32(bound), f32(dt_gamma),
                               u32(max_steps), u32(N), u32(C), u32(H), u32(M), _p(nears), _p(fars), _p(xyzs),
                               _p(dirs), _p(deltas), _p(rays), _p(step_counter), _p(noises))
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0])
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    """raymarching.py:241-269 + raymarching.cu:501-577."""
    sigmas = _f(sigmas)
    rgbs = _f(rgbs)
    deltas = _f(deltas)
    rays = _i(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws = np.empty(N, np.float32)
    depth = np.empty(N, np.float32)
    image = np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), u32(M), u32(N),
                                           f32(T_thresh), _p(ws), _p(depth), _p(image))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, T_thresh=1e-4):
    """raymarching.py:273-288 + raymarching.cu:602-682 (grad_depth is ignored by the reference)."""
    sigmas = _f(sigmas)
    rgbs = _f(rgbs)
    deltas = _f(deltas)
    rays = _i(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    gs = np.zeros_like(sigmas)
    gc = np.zeros_like(rgbs)
    lib().orc_composite_rays_train_backward(_p(_f(grad_ws)), _p(_f(grad_image)), _p(sigmas), _p(rgbs), _p(deltas),
                                            _p(rays), _p(_f(weights_sum)), _p(_f(image)), u32(M), u32(N),
                                            f32(T_thresh), _p(gs), _p(gc))
    return gs, gc


# --------------------------------------------------------------------------
# inference
# --------------------------------------------------------------------------
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
               align=-1, perturb=False, dt_gamma=0, max_steps=1024, noises=None):
    """raymarching.py:300-346 + raymarching.cu:701-805."""
    rays_o = _f(rays_o).reshape(-1, 3)
    rays_d = _f(rays_d).reshape(-1, 3)
    rays_alive = _i(rays_alive)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    if noises is None:
        noises = np.random.default_rng(0).random(n_alive, dtype=np.float32) if perturb else np.zeros(n_alive, np.float32)
    lib().orc_march_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(_f(rays_t)), _p(rays_o), _p(rays_d), f32(bound),
                         f32(dt_gamma), u32(max_steps), u32(C), u32(H),
                         _p(np.ascontiguousarray(density_bitfield, dtype=np.uint8)), _p(_f(near)), _p(_f(far)),
                         _p(xyzs), _p(dirs), _p(deltas), _p(_f(noises)))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """raymarching.py:354-370 + raymarching.cu:819-905.  Mutates rays_alive, rays_t,
    weights_sum, depth, image in place (they must be contiguous arrays of the right dtype)."""
    for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32), (depth, np.float32), (image, np.float32)):
        assert a.dtype == dt and a.flags.c_contiguous
    lib().orc_composite_rays(u32(n_alive), u32(n_step), f32(T_thresh), _p(rays_alive), _p(rays_t), _p(_f(sigmas)),
                             _p(_f(rgbs)), _p(_f(deltas)), _p(weights_sum), _p(depth), _p(image))
    return ()


# --------------------------------------------------------------------------
# grid encoder
# --------------------------------------------------------------------------
def grid_offsets(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, align_corners=False):
    """gridencoder/grid.py:100-128: per-level row offsets (int32 [L+1]) and per_level_scale."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets = []
    offset = 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), per_level_scale


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                        gridtype=0, align_corners=False, interpolation=0):
    """gridencoder/grid.py:27-63 + gridencoder.cu:87-245.  ``embeddings`` float32 or float16
    ([rows, C]); returns (outputs [B, L*C], dy_dx [B, L*D*C] | None) in the table's dtype."""
    inputs = _f(inputs)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    C = embeddings.shape[1]
    S = np.log2(per_level_scale)
    is_half = embeddings.dtype == np.float16
    emb = np.ascontiguousarray(embeddings)
    outputs = np.empty((L, B, C), emb.dtype)
    dy_dx = np.empty((B, L * D * C), emb.dtype) if calc_grad_inputs else None
    lib().orc_grid_encode_forward(_p(inputs), _p(emb), _p(_i(offsets)), _p(outputs), u32(B), u32(D), u32(C), u32(L),
                                  f32(S), u32(base_resolution), _p(dy_dx), u32(gridtype), ctypes.c_int(bool(align_corners)),
                                  u32(interpolation), ctypes.c_int(is_half))
    return np.ascontiguousarray(outputs.transpose(1, 0, 2)).reshape(B, L * C), dy_dx


def grid_encode_backward(grad, inputs, embeddings, offsets, per_level_scale, base_resolution, dy_dx=None,
                         gridtype=0, align_corners=False, interpolation=0):
    """gridencoder/grid.py:68-89 + gridencoder.cu:248-369.  grad: [B, L*C] in the table's dtype.
    Returns (grad_embeddings [rows, C], grad_inputs [B, D] float32 | None)."""
    inputs = _f(inputs)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    C = embeddings.shape[1]
    S = np.log2(per_level_scale)
    is_half = embeddings.dtype == np.float16
    g = np.ascontiguousarray(np.asarray(grad, dtype=embeddings.dtype).reshape(B, L, C).transpose(1, 0, 2))
    grad_emb = np.zeros_like(embeddings)
    grad_inputs = np.zeros((B, D), embeddings.dtype) if dy_dx is not None else None
    lib().orc_grid_encode_backward(_p(g), _p(inputs), _p(_i(offsets)), _p(grad_emb), u32(B), u32(D), u32(C), u32(L),
                                   f32(S), u32(base_resolution), _p(dy_dx), _p(grad_inputs), u32(gridtype),
                                   ctypes.c_int(bool(align_corners)), u32(interpolation), ctypes.c_int(is_half))
    if grad_inputs is not None:
        grad_inputs = grad_inputs.astype(np.float32)
    return grad_emb, grad_inputs


# --------------------------------------------------------------------------
# frequency encoder
# --------------------------------------------------------------------------
def freq_encode_forward(inputs, degree, output_dim=None):
    """freqencoder/freq.py:18-35 + freqencoder.cu:30-58."""
    inputs = _f(inputs)
    B, D = inputs.shape
    C = D + D * 2 * degree if output_dim is None else output_dim
    out = np.empty((B, C), np.float32)
    lib().orc_freq_encode_forward(_p(inputs), u32(B), u32(D), u32(degree), u32(C), _p(out))
    return out


def freq_encode_backward(grad, outputs, input_dim, degree):
    """freqencoder/freq.py:40-49 + freqencoder.cu:63-94."""
    grad = _f(grad)
    outputs = _f(outputs)
    B, C = outputs.shape
    gi = np.zeros((B, input_dim), np.float32)
    lib().orc_freq_encode_backward(_p(grad), _p(outputs), u32(B), u32(input_dim), u32(degree), u32(C), _p(gi))
    return gi


# --------------------------------------------------------------------------
# spherical harmonics: evaluated in numpy float64 from the textbook definition
#   Y_l^0    = K_l^0 P_l(z)
#   Y_l^{+m} = (-1)^m sqrt2 K_l^m Re(x+iy)^m  d^m P_l/dz^m
#   Y_l^{-m} = (-1)^m sqrt2 K_l^m Im(x+iy)^m  d^m P_l/dz^m
# which, as polynomials on R^3, are the closed forms of shencoder.cu:49-121
# (checked term by term against tests/golden/sh_reference_closed_form.npz).
# --------------------------------------------------------------------------
def _legendre_deriv_coeffs(l, m):
    """Coefficients (ascending powers of z) of d^m/dz^m P_l(z)."""
    c = np.polynomial.legendre.leg2poly([0] * l + [1])
    for _ in range(m):
        c = np.polynomial.polynomial.polyder(c)
    return np.atleast_1d(c)


def sh_encode_forward(inputs, degree, calc_grad_inputs=False):
    """shencoder/sphere_harmonics.py:17-41 + shencoder.cu:27-355.  Returns (outputs [B, degree^2],
    dy_dx [B, 3*degree^2] | None), float32 (rounded once from float64)."""
    v = np.asarray(inputs, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    B = v.shape[0]
    C2 = degree * degree
    out = np.zeros((B, C2))
    dx = np.zeros((B, C2))
    dy = np.zeros((B, C2))
    dz = np.zeros((B, C2))
    # A_m = Re (x+iy)^m, B_m = Im (x+iy)^m
    A = [np.ones(B)]
    Bm = [np.zeros(B)]
    for m in range(1, degree):
        A.append(x * A[m - 1] - y * Bm[m - 1])
        Bm.append(x * Bm[m - 1] + y * A[m - 1])
    pv = np.polynomial.polynomial.polyval
    for l in range(degree):
        for m in range(0, l + 1):
            K = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            q = _legendre_deriv_coeffs(l, m)
            Q = pv(z, q)
            dQ = pv(z, np.polynomial.polynomial.polyder(q)) if len(q) > 1 else np.zeros(B)
            if m == 0:
                i0 = l * l + l
                out[:, i0] = K * Q
                dz[:, i0] = K * dQ
            else:
                c = (-1) ** m * math.sqrt(2.0) * K
                ip, im = l * l + l + m, l * l + l - m
                out[:, ip] = c * A[m] * Q
                out[:, im] = c * Bm[m] * Q
                dx[:, ip] = c * m * A[m - 1] * Q
                dy[:, ip] = -c * m * Bm[m - 1] * Q
                dz[:, ip] = c * A[m] * dQ
                dx[:, im] = c * m * Bm[m - 1] * Q
                dy[:, im] = c * m * A[m - 1] * Q
                dz[:, im] = c * Bm[m] * dQ
    outputs = out.astype(np.float32)
    if not calc_grad_inputs:
        return outputs, None
    dy_dx = np.concatenate([dx, dy, dz], axis=1).astype(np.float32)  # [B, D*C2], d-major
    return outputs, dy_dx


def sh_encode_backward(grad, dy_dx, degree):
    """shencoder/sphere_harmonics.py:46-55 + shencoder.cu:358-382."""
    grad = _f(grad)
    dy_dx = _f(dy_dx)
    B = grad.shape[0]
    gi = np.zeros((B, 3), np.float32)
    lib().orc_sh_encode_backward(_p(grad), u32(B), u32(3), u32(degree), _p(dy_dx), _p(gi))
    return gi


# --------------------------------------------------------------------------
# activations used by the field network (activation.py:5-17)
# --------------------------------------------------------------------------
def trunc_exp_forward(x):
    return np.exp(np.asarray(x, np.float32)).astype(np.float32)


def trunc_exp_backward(g, x):
    return (np.asarray(g, np.float32) * np.exp(np.clip(np.asarray(x, np.float32), -15, 15))).astype(np.float32)
