"""Operator API of the reference's raymarching extension, on libsdn_hip (MI355X).

Same callables, argument order, defaults and return values as
/root/reference/raymarching/raymarching.py (line numbers cited per function), so
dnerf/renderer.py and SealDNeRF/renderer.py can call them unchanged.  Differences,
all on the safe side of the reference's behaviour:
  * buffers are validated (device / dtype / contiguity) instead of unchecked;
  * kernels launch on torch's current stream, not the legacy default stream;
  * march_rays_train allocates sample slots with a deterministic scan in ray order
    (rays[i, 0] == i) instead of two racing atomics; per ray the samples are identical;
  * march_rays clears only the alignment tail of its outputs (the kernel writes the
    zeros for exhausted rays itself) instead of memset-ing three M-sized buffers.
"""
import weakref

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import sdn_backend as _sdn
from sdn_backend import lib as _lib, check as _check, ptr as _ptr, stream as _stream, to_device as _dev

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
           "composite_rays_train", "march_rays", "composite_rays", "compact_alive", "build_cull_grid", "march_rays_ex"]

_f32 = torch.float32
_i32 = torch.int32


def _rays(rays_o, rays_d):
    rays_o = _dev(rays_o).contiguous().view(-1, 3)
    rays_d = _dev(rays_d).contiguous().view(-1, 3)
    return rays_o, rays_d


# ----------------------------------------------------------------------------------------------
# utils
# ----------------------------------------------------------------------------------------------
class _near_far_from_aabb(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """raymarching.py:22-47.  rays_o/d [N,3], aabb [6] -> nears [N], fars [N]."""
        rays_o, rays_d = _rays(rays_o, rays_d)
        aabb = _dev(aabb).contiguous()
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=_f32, device=rays_o.device)
        fars = torch.empty(N, dtype=_f32, device=rays_o.device)
        _check(_lib.sdn_near_far_from_aabb(_ptr(rays_o, _f32, "rays_o"), _ptr(rays_d, _f32, "rays_d"), _ptr(aabb, _f32, "aabb"),
                                           N, float(min_near), _ptr(nears), _ptr(fars), _stream()), "near_far_from_aabb")
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, rays_o, rays_d, radius):
        """raymarching.py:55-78.  -> coords [N,2] in [-1,1]."""
        rays_o, rays_d = _rays(rays_o, rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=_f32, device=rays_o.device)
        _check(_lib.sdn_sph_from_ray(_ptr(rays_o, _f32, "rays_o"), _ptr(rays_d, _f32, "rays_d"), float(radius), N, _ptr(coords),
                                     _stream()), "sph_from_ray")
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """raymarching.py:85-102.  coords [N,3] int -> indices [N] int32."""
        coords = _dev(coords).int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=_i32, device=coords.device)
        _check(_lib.sdn_morton3D(_ptr(coords, _i32, "coords"), N, _ptr(indices), _stream()), "morton3D")
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """raymarching.py:108-124.  indices [N] int -> coords [N,3] int32."""
        indices = _dev(indices).int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=_i32, device=indices.device)
        _check(_lib.sdn_morton3D_invert(_ptr(indices, _i32, "indices"), N, _ptr(coords), _stream()), "morton3D_invert")
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, grid, thresh, bitfield=None):
        """raymarching.py:132-153.  grid [C, H^3] f32 -> bitfield [C*H^3/8] u8 (written in place if given)."""
        grid = _dev(grid).contiguous()
        N = grid.shape[0] * grid.shape[1] // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        _check(_lib.sdn_packbits(_ptr(grid, _f32, "grid"), N, float(thresh), _ptr(bitfield, torch.uint8, "bitfield"), _stream()),
               "packbits")
        return bitfield


packbits = _packbits.apply


# ----------------------------------------------------------------------------------------------
# train
# ----------------------------------------------------------------------------------------------
class _march_rays_train(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        """raymarching.py:164-233.  -> xyzs [M,3], dirs [M,3], deltas [M,2], rays [N,3] (id, offset, count)."""
        rays_o, rays_d = _rays(rays_o, rays_d)
        density_bitfield = _dev(density_bitfield).contiguous()
        N = rays_o.shape[0]
        M = N * max_steps
        # running-average point budget (raymarching.py:200-203), including the "+align even when aligned" quirk
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        dev = rays_o.device
        # zero-initialised like the reference's (slots past the emitted samples are evaluated by the network): one fill, three views
        flat = torch.zeros(M * 8, dtype=_f32, device=dev)
        xyzs, dirs, deltas = flat[:3 * M].view(M, 3), flat[3 * M:6 * M].view(M, 3), flat[6 * M:].view(M, 2)
        rays = torch.empty(N, 3, dtype=_i32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=_i32, device=dev)
        noises = torch.rand(N, dtype=_f32, device=dev) if perturb else torch.zeros(N, dtype=_f32, device=dev)
        scratch = torch.empty(int(_lib.sdn_march_rays_train_scratch_bytes(N, int(max_steps))), dtype=torch.uint8, device=dev)
        _check(_lib.sdn_march_rays_train(_ptr(rays_o, _f32, "rays_o"), _ptr(rays_d, _f32, "rays_d"),
                                         _ptr(density_bitfield, torch.uint8, "density_bitfield"), float(bound), float(dt_gamma),
                                         int(max_steps), N, int(C), int(H), M, _ptr(nears.contiguous(), _f32, "nears"),
                                         _ptr(fars.contiguous(), _f32, "fars"), _ptr(xyzs), _ptr(dirs), _ptr(deltas), _ptr(rays),
                                         _ptr(step_counter, _i32, "step_counter"), _ptr(noises), _ptr(scratch), _stream()),
               "march_rays_train")
        # only used at the first (few) epochs (raymarching.py:223-231): host read-back of the point count
        if force_all_rays or mean_count <= 0:
            m = int(step_counter[0].item())
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
            torch.cuda.empty_cache()
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        """raymarching.py:241-269.  -> weights_sum [N], depth [N], image [N,3]."""
        sigmas = sigmas.contiguous()
        rgbs = rgbs.contiguous()
        deltas = deltas.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=_f32, device=dev)
        depth = torch.empty(N, dtype=_f32, device=dev)
        image = torch.empty(N, 3, dtype=_f32, device=dev)
        _check(_lib.sdn_composite_rays_train_forward(_ptr(sigmas, _f32, "sigmas"), _ptr(rgbs, _f32, "rgbs"), _ptr(deltas, _f32, "deltas"),
                                                     _ptr(rays, _i32, "rays"), M, N, float(T_thresh), _ptr(weights_sum), _ptr(depth),
                                                     _ptr(image), _stream()), "composite_rays_train_forward")
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        return weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        """raymarching.py:273-288.  grad_depth is ignored, exactly as in the reference (:275)."""
        grad_weights_sum = grad_weights_sum.contiguous()
        grad_image = grad_image.contiguous()
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        flat = torch.zeros(4 * M, dtype=sigmas.dtype, device=sigmas.device)    # one fill for both gradients
        grad_sigmas, grad_rgbs = flat[:M], flat[M:].view(M, 3)
        _check(_lib.sdn_composite_rays_train_backward(_ptr(grad_weights_sum, _f32, "grad_weights_sum"), _ptr(grad_image, _f32, "grad_image"),
                                                      _ptr(sigmas), _ptr(rgbs), _ptr(deltas), _ptr(rays), _ptr(weights_sum), _ptr(image),
                                                      M, N, float(T_thresh), _ptr(grad_sigmas), _ptr(grad_rgbs), _stream()),
               "composite_rays_train_backward")
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _composite_rays_train.apply


# ----------------------------------------------------------------------------------------------
# inference
# ----------------------------------------------------------------------------------------------
_CULL_CACHE = {}          # (data_ptr, bytes, device) -> (tensor version, cull grid): at most 256 occupancy slices


def _cull_grid_of(bitfield, C, H):
    """The marcher's coarse skip grid of one occupancy slice (128^3, cascade 1, 8-byte aligned), derived once per content: keyed by the
    slice's address and its tensor version (a view shares its base's counter: `density_bitfield[t]` of a bitfield rewritten in place --
    update_extra_state, load_state_dict, fill_bitfield -- misses).  None: this slice takes the plain marcher."""
    if int(H) != 128 or int(C) != 1 or not bitfield.is_cuda or bitfield.dtype != torch.uint8 or not bitfield.is_contiguous():
        return None
    if bitfield.numel() != 128 * 128 * 128 // 8 or (bitfield.data_ptr() & 7) != 0:
        return None
    key = (bitfield.data_ptr(), bitfield.numel(), str(bitfield.device))
    hit = _CULL_CACHE.get(key)
    # (the buffer OBJECT too -- a slice is a fresh view per call, its base is not: the allocator hands a freed bitfield's address to the
    #  next model's, and the version counters of two buffers can coincide)
    owner = bitfield._base if bitfield._base is not None else bitfield
    if hit is not None and hit[0] == bitfield._version and hit[2]() is owner:
        return hit[1]
    if len(_CULL_CACHE) >= 256:
        _CULL_CACHE.clear()
    grid = torch.empty(int(_lib.sdn_cull_grid_bytes()), dtype=torch.uint8, device=bitfield.device)
    _check(_lib.sdn_build_cull_grid(_ptr(bitfield), 128, _ptr(grid), _stream()), "build_cull_grid")
    _CULL_CACHE[key] = (bitfield._version, grid, weakref.ref(owner))
    return grid


def _to_f32(t):
    return t if t.dtype == _f32 else t.float()


# Live-sample lists for the fused field networks behind the reference's caller: `march_rays` returns exactly the reference's three
# tensors; with the switch on it also hangs (slot list, count, tensor version) on the xyzs tensor as `_sdn_live`.  The caller
# (dnerf/renderer.py:350-376) hands that same tensor to `self(xyzs, dirs, time)`; a tensor that was modified, copied or mapped in
# between (SealD's mapper) carries no list and is evaluated slot by slot as before.  Off by default: two small allocations and one
# 4-byte fill per call; NeRFNetwork's fused dispatch switches it on when it is used (set "pinned" to keep a manual choice).
live_lists = {"on": False, "pinned": False}


class _march_rays(Function):
    """The reference wraps the inference marcher in an autograd Function only for `custom_fwd(cast_inputs=float32)`; it has no
    backward (raymarching.py:300-346).  Here `march_rays` / `composite_rays` are plain functions that do the same cast themselves:
    `Function.apply` + the autocast bookkeeping cost ~10 us of host time per call, and the reference-shaped render loop is bound by
    exactly that (profiles/r04_reference_shaped_host_time.txt).  The classes stay for callers that name them."""

    @staticmethod
    def forward(ctx, *args):
        return march_rays(*args)


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
               perturb=False, dt_gamma=0, max_steps=1024):
    """raymarching.py:300-346.  -> xyzs [Mp,3], dirs [Mp,3], deltas [Mp,2], Mp = n_alive*n_step padded
    with `M += align - M % align` (a full `align` when already aligned, as in the reference :331-332)."""
    rays_o, rays_d = _rays(_to_f32(rays_o), _to_f32(rays_d))
    rays_t, near, far = _to_f32(rays_t), _to_f32(near), _to_f32(far)
    dev = rays_o.device
    M0 = n_alive * n_step
    M = M0
    if align > 0:
        M += align - (M % align)
    xyzs = torch.empty(M, 3, dtype=_f32, device=dev)
    dirs = torch.empty(M, 3, dtype=_f32, device=dev)
    deltas = torch.empty(M, 2, dtype=_f32, device=dev)
    noises = torch.rand(n_alive, dtype=_f32, device=dev) if perturb else None
    cull = _cull_grid_of(density_bitfield, C, H)
    if cull is not None:
        live_idx = live_count = None
        if live_lists["on"] and M0 > 0:
            # the marcher appends every slot that receives a sample to a list as it goes (wave-aggregated, no extra pass); a fused
            # field network that is handed THIS xyzs tensor evaluates the listed slots only (dnerf_amd/network.py)
            live_idx = torch.empty(M0, dtype=_i32, device=dev)
            live_count = torch.zeros(1, dtype=_i32, device=dev)
        # the exact cull grid (sdn_build_cull_grid, kept per occupancy slice and tensor version): rays whose remaining segment stays
        # clear of every occupied voxel retire at once instead of probing hundreds of empty voxels -- the same samples, bit for bit
        # (tests/test_gpu_ops_parity.py), 122 -> ~25 us per call in the reference-shaped loop of an 800x800 frame; the kernel also
        # clears the padded tail itself
        _check(_lib.sdn_march_rays_ex(int(n_alive), int(n_step), _ptr(rays_alive, _i32, "rays_alive"), _ptr(rays_t, _f32, "rays_t"),
                                      _ptr(rays_o, _f32, "rays_o"), _ptr(rays_d, _f32, "rays_d"), float(bound), float(dt_gamma),
                                      int(max_steps), int(C), int(H), _ptr(density_bitfield, torch.uint8, "density_bitfield"),
                                      _ptr(far, _f32, "far"), _ptr(xyzs), _ptr(dirs), _ptr(deltas), _ptr(noises), int(M), _ptr(cull),
                                      _ptr(live_idx), _ptr(live_count), _stream()), "march_rays_ex")
        if live_idx is not None:
            xyzs._sdn_live = (live_idx, live_count, xyzs._version)
        return xyzs, dirs, deltas
    if M > M0:  # the kernel writes every slot below n_alive*n_step; only the tail needs clearing
        xyzs[M0:].zero_(); dirs[M0:].zero_(); deltas[M0:].zero_()
    _check(_lib.sdn_march_rays(int(n_alive), int(n_step), _ptr(rays_alive, _i32, "rays_alive"), _ptr(rays_t, _f32, "rays_t"),
                               _ptr(rays_o, _f32, "rays_o"), _ptr(rays_d, _f32, "rays_d"), float(bound), float(dt_gamma),
                               int(max_steps), int(C), int(H), _ptr(density_bitfield, torch.uint8, "density_bitfield"),
                               _ptr(near, _f32, "near"), _ptr(far, _f32, "far"), _ptr(xyzs), _ptr(dirs), _ptr(deltas),
                               _ptr(noises), _stream()), "march_rays")
    return xyzs, dirs, deltas


class _composite_rays(Function):
    @staticmethod
    def forward(ctx, *args):
        return composite_rays(*args)


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """raymarching.py:354-370.  Mutates rays_alive (-1 = terminated), rays_t, weights_sum, depth, image."""
    sigmas, rgbs, deltas = _to_f32(sigmas), _to_f32(rgbs), _to_f32(deltas)   # sigmas / rgbs arrive as fp16 under autocast (op-by-op network)
    _check(_lib.sdn_composite_rays(int(n_alive), int(n_step), float(T_thresh), _ptr(rays_alive, _i32, "rays_alive"),
                                   _ptr(rays_t, _f32, "rays_t"), _ptr(sigmas.contiguous(), _f32, "sigmas"),
                                   _ptr(rgbs.contiguous(), _f32, "rgbs"), _ptr(deltas.contiguous(), _f32, "deltas"),
                                   _ptr(weights_sum, _f32, "weights_sum"), _ptr(depth, _f32, "depth"), _ptr(image, _f32, "image"),
                                   _stream()), "composite_rays")
    return tuple()


def compact_alive(rays_alive, out=None, count=None, scratch=None):
    """Extension: stable device-side `rays_alive[rays_alive >= 0]` (dnerf/renderer.py:372) without the
    boolean-mask kernels.  Returns (out, count) where count is a device int32[1]; the caller decides when
    to read it back."""
    n = rays_alive.shape[0]
    dev = rays_alive.device
    if out is None:
        out = torch.empty(n, dtype=_i32, device=dev)
    if count is None:
        count = torch.empty(1, dtype=_i32, device=dev)
    if scratch is None:
        scratch = torch.empty(max(int(_lib.sdn_compact_alive_scratch_bytes(n)), 4), dtype=torch.uint8, device=dev)
    _check(_lib.sdn_compact_alive(_ptr(rays_alive, _i32, "rays_alive"), n, _ptr(out, _i32, "out"), _ptr(count, _i32, "count"),
                                  _ptr(scratch), _stream()), "compact_alive")
    return out, count


def build_cull_grid(density_bitfield, H=128):
    """Extension: 32^3 byte grid marking coarse cells within one cell of an occupied voxel (exact ray early-out)."""
    cull = torch.empty(int(_lib.sdn_cull_grid_bytes()), dtype=torch.uint8, device=density_bitfield.device)
    _check(_lib.sdn_build_cull_grid(_ptr(density_bitfield, torch.uint8, "density_bitfield"), int(H), _ptr(cull), _stream()), "build_cull_grid")
    return cull


def march_rays_ex(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, far, align=128, dt_gamma=0,
                  max_steps=1024, cull_grid=None, want_live_list=False):
    """Extension of march_rays used by the native loop: same samples bit for bit; optionally culls dead-end rays exactly and
    returns the (unordered) list of slots that received a sample.  -> xyzs, dirs, deltas[, live_idx, live_count]."""
    dev = rays_o.device
    M0 = n_alive * n_step
    M = M0 + (align - M0 % align) if align > 0 else M0
    xyzs = torch.empty(M, 3, dtype=_f32, device=dev)
    dirs = torch.empty(M, 3, dtype=_f32, device=dev)
    deltas = torch.empty(M, 2, dtype=_f32, device=dev)
    live_idx = torch.empty(max(M0, 1), dtype=_i32, device=dev) if want_live_list else None
    live_count = torch.zeros(1, dtype=_i32, device=dev) if want_live_list else None
    _check(_lib.sdn_march_rays_ex(int(n_alive), int(n_step), _ptr(rays_alive, _i32), _ptr(rays_t, _f32), _ptr(rays_o, _f32), _ptr(rays_d, _f32),
                                  float(bound), float(dt_gamma), int(max_steps), int(C), int(H), _ptr(density_bitfield, torch.uint8),
                                  _ptr(far, _f32), _ptr(xyzs), _ptr(dirs), _ptr(deltas), None, M, _ptr(cull_grid), _ptr(live_idx),
                                  _ptr(live_count), _stream()), "march_rays_ex")
    if want_live_list:
        return xyzs, dirs, deltas, live_idx, live_count
    return xyzs, dirs, deltas
