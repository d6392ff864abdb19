"""ctypes binding of libsdn_hip.so (the C ABI declared in include/sdn_hip.h).

This is the only place the Python layer touches native code.  There is no CPU
fallback: if the library is missing the import fails, and if an operator is
called without a HIP device (or with a tensor that is not on one) it raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDN_LIB_PATH") or os.path.join(_HERE, "lib", "libsdn_hip.so")

SDN_F32 = 0
SDN_F16 = 1

_c = ctypes
_vp, _u32, _f32, _i32, _u64 = _c.c_void_p, _c.c_uint32, _c.c_float, _c.c_int, _c.c_uint64

# name -> argtypes, exactly the prototypes of include/sdn_hip.h (restype int unless noted)
PROTOTYPES = {
    "sdn_near_far_from_aabb": [_vp, _vp, _vp, _u32, _f32, _vp, _vp, _vp],
    "sdn_sph_from_ray": [_vp, _vp, _f32, _u32, _vp, _vp],
    "sdn_get_rays": [_vp, _f32, _f32, _f32, _f32, _u32, _u32, _vp, _vp, _vp],
    "sdn_morton3D": [_vp, _u32, _vp, _vp],
    "sdn_morton3D_invert": [_vp, _u32, _vp, _vp],
    "sdn_packbits": [_vp, _u32, _f32, _vp, _vp],
    "sdn_march_rays_train": [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sdn_composite_rays_train_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp, _vp],
    "sdn_composite_whole_rays": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp, _vp],
    "sdn_composite_rays_train_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _f32, _vp, _vp, _vp],
    "sdn_march_rays": [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sdn_march_rays_ex": [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp],
    "sdn_build_cull_grid": [_vp, _u32, _vp, _vp],
    "sdn_composite_rays": [_u32, _u32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sdn_compact_alive": [_vp, _u32, _vp, _vp, _vp, _vp],
    "sdn_grid_encode_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _u32, _i32, _u32, _i32, _vp],
    "sdn_grid_encode_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _i32, _u32, _i32, _vp],
    "sdn_grid_encode_backward_det": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _vp, _vp, _u32, _i32, _u32, _i32, _vp, _vp],
    "sdn_sh_encode_forward": [_vp, _vp, _u32, _u32, _u32, _vp, _vp],
    "sdn_sh_encode_backward": [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp],
    "sdn_freq_encode_forward": [_vp, _u32, _u32, _u32, _u32, _vp, _vp],
    "sdn_freq_encode_backward": [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp],
    "sdn_render_begin": [_vp, _vp],
    "sdn_render_step_f16": [_vp, _u32, _vp],
    "sdn_render_step_f16_ev": [_vp, _u32, _vp, _vp, _vp],
    "sdn_render_finish": [_vp, _f32, _vp, _vp, _vp],
    "sdn_render_time_kernel": [_i32],
    "sdn_render_frame_f16": [_vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "sdn_render_frames_pipelined_f16": [_vp, _u32, _u32, _vp, _vp, _vp, _vp, _f32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp],
    "sdn_host_mailbox_free": [_vp],
    "sdn_seal_bbox_map": [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sdn_seal_modify_hsv": [_vp, _vp, _u32, _f32, _f32, _f32, _vp],
    "sdn_seal_bbox_map_source": [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sdn_seal_modify_rgb": [_vp, _vp, _u32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp],
    "sdn_field_forward_f16": [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _f32, _u32, _f32, _f32, _i32, _vp, _vp, _vp],
    "sdn_field_forward_f32": [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _f32, _u32, _f32, _f32, _i32, _vp, _vp, _vp, _vp],
    "sdn_field_forward_f32x3": [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _f32, _u32, _f32, _f32, _i32, _vp, _vp, _vp, _vp],
    "sdn_field_build_quad_table": [_vp, _i32, _vp, _f32, _u32, _vp, _vp],
    "sdn_grid_encode_forward_quad_f16": [_vp, _vp, _vp, _vp, _u32, _f32, _u32, _vp],
    "sdn_density_query_cells_f16": [_vp, _vp, _u32, _vp, _u32, _u32, _f32, _vp, _vp, _vp, _vp, _f32, _u32, _f32, _f32, _i32, _vp, _vp],
    "sdn_debug_shader_clock": [_vp, _u32, _vp],
    "sdn_density_query_cells_f32": [_vp, _vp, _u32, _vp, _u32, _u32, _f32, _vp, _vp, _vp, _vp, _f32, _u32, _f32, _f32, _i32, _vp, _vp],
    "sdn_density_grid_ema": [_vp, _vp, ctypes.c_uint64, _f32, _vp, _vp],
    "sdn_density_grid_pack": [_vp, ctypes.c_uint64, _vp, _f32, _vp, _vp, _vp],
    "sdn_ffmlp_forward": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp],
    "sdn_ffmlp_inference": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "sdn_ffmlp_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _i32, _vp, _vp, _vp, _vp, _vp],
}
MAX_GROUP_FRAMES = 16   # SDN_MAX_GROUP_FRAMES


class SdnFrameTime(ctypes.Structure):
    """Mirror of `SdnFrameTime` in include/sdn_hip.h: the time-dependent constants of one frame (or of each frame of a group)."""
    _fields_ = [("bitfield", _vp * MAX_GROUP_FRAMES), ("field_bias0", _vp), ("zero_deform", _u32), ("reserved_", _u32),
                ("cull_grid", _vp * MAX_GROUP_FRAMES)]


class SdnRenderCtx(ctypes.Structure):
    """Mirror of `SdnRenderCtx` in include/sdn_hip.h (field order and types must match)."""
    _fields_ = ([(n, _vp) for n in ("rays_o", "rays_d", "nears", "fars", "bitfield", "cull_bits", "alive_a", "alive_b", "rays_t",
                                   "weights_sum", "depth", "image", "xyzs", "dirs", "deltas", "sigmas", "rgbs", "live_idx",
                                   "live_counts", "state", "trace", "n_out", "block_totals", "field_weights", "field_bias0",
                                   "grid_table")]
                + [("grid_offsets", ctypes.c_int32 * 17), ("grid_S", _f32), ("grid_H", _u32)]
                + [(n, _u32) for n in ("N", "M_cap", "n_counters", "max_steps", "C", "H")]
                + [(n, _f32) for n in ("bound", "dt_gamma", "T_thresh", "density_scale")]
                + [("zero_deform", ctypes.c_int32), ("aabb", _vp), ("min_near", _f32), ("reserved_", ctypes.c_int32), ("rays_tend", _vp), ("seal", _vp), ("seal_mask", _vp),
                   ("n_group_frames", _u32), ("rays_per_frame", _u32), ("frame_bitfield", _vp * MAX_GROUP_FRAMES), ("slot_frame", _vp),
                   ("frame_cull", _vp * MAX_GROUP_FRAMES), ("field_f32", _i32), ("reserved2_", _i32)])


class SdnSealBox(ctypes.Structure):
    """Mirror of `SdnSealBox` in include/sdn_hip.h."""
    _fields_ = [("bounds", _f32 * 24), ("n_bounds", _u32), ("n_tris", _u32), ("tris", _vp), ("test_dir", _f32 * 3), ("tinv", _f32 * 12),
                ("rinv", _f32 * 9), ("scale", _f32 * 3), ("center", _f32 * 3), ("hsv", _f32 * 3), ("modify_hsv", ctypes.c_int32),
                ("rgb", _f32 * 3), ("rgb_light_offset", _f32), ("modify_rgb", ctypes.c_int32), ("has_map_source", ctypes.c_int32),
                ("source_bound", _f32 * 6), ("map_source", _f32 * 3), ("reserved_", _u32), ("scratch", _vp)]


TRAIN_N_PARAMS = 14   # SDN_TRAIN_N_PARAMS


class SdnTrainParam(ctypes.Structure):
    """Mirror of `SdnTrainParam` in include/sdn_hip.h."""
    _fields_ = [("param", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("ema", _vp), ("n", ctypes.c_uint64)]


class SdnTrainStep(ctypes.Structure):
    """Mirror of `SdnTrainStep` in include/sdn_hip.h (field order and types must match)."""
    _fields_ = ([(n, _vp) for n in ("rays_o", "rays_d", "target", "bg_color")]
                + [("bg_value", _f32), ("N", _u32), ("M", _u32), ("bitfield", _vp), ("cull_grid", _vp), ("aabb", _vp)]
                + [(n, _f32) for n in ("bound", "min_near", "dt_gamma", "density_scale", "T_thresh", "time")]
                + [(n, _u32) for n in ("cascade", "grid_size", "max_steps")]
                + [("perturb", ctypes.c_int32), ("noise_seed", ctypes.c_uint64), ("noises", _vp), ("counter", _vp),
                   ("grid_offsets", ctypes.c_int32 * 17), ("grid_S", _f32), ("grid_H", _u32),
                   ("params", SdnTrainParam * TRAIN_N_PARAMS)]
                + [(n, ctypes.c_double) for n in ("lr_table", "lr_net", "beta1", "beta2", "eps")]
                + [("adam_steps", _vp), ("loss_scale", _vp), ("growth_tracker", _vp), ("growth_factor", _f32), ("backoff_factor", _f32),
                   ("growth_interval", _u32), ("ema_decay", _f32), ("loss_out", _vp), ("image_out", _vp), ("workspace", _vp),
                   ("mode", ctypes.c_int32), ("keep_deform", ctypes.c_int32), ("grad_divisor", _f32), ("deform_frozen", ctypes.c_int32), ("phase", ctypes.c_int32), ("sample_set", ctypes.c_int32),
                   ("table_stream", _vp), ("table_ready", _vp), ("table_done", _vp), ("det_scratch", _vp)])


class SdnTrainLayout(ctypes.Structure):
    """Mirror of `SdnTrainLayout` in include/sdn_hip.h."""
    _fields_ = [(n, ctypes.c_uint64) for n in ("total_bytes", "w_table", "w_deform", "w_sigma0", "w_sigma1", "w_color", "g_table", "g_deform",
                                                "g_sigma0", "g_sigma1", "g_color", "xyzs", "dirs", "deltas", "rays", "sample_set_stride", "sigmas",
                                                "weights_sum", "depth", "image", "found_inf")]


PROTOTYPES.update({
    "sdn_train_layout": [_u32, _u32, _u32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(SdnTrainLayout)],
    "sdn_train_refresh": [ctypes.POINTER(SdnTrainStep), _vp],
    "sdn_train_step_f16": [ctypes.POINTER(SdnTrainStep), _vp],
    "sdn_train_flush": [ctypes.POINTER(SdnTrainStep), _vp],
})

PROTOTYPES_U32 = {
    "sdn_field_weight_blocks": [],
    "sdn_field_weight_floats_f32": [],
    "sdn_cull_grid_bytes": [],
}
PROTOTYPES_U64 = {
    "sdn_march_rays_train_scratch_bytes": [_u32, _u32],
    "sdn_compact_alive_scratch_bytes": [_u32],
    "sdn_ffmlp_scratch_bytes": [_u32, _u32, _u32, _u32, _u32],
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C seald-nerf_amd/csrc` (or __graft_entry__.build()). "
        "There is no CPU fallback for the SealD-NeRF operators.")

lib = ctypes.CDLL(LIB_PATH)
lib.sdn_version.restype = ctypes.c_char_p
lib.sdn_host_mailbox_alloc.restype = ctypes.c_void_p
lib.sdn_host_mailbox_alloc.argtypes = [ctypes.c_uint32]
lib.sdn_field_select_kernel.restype = None
lib.sdn_field_select_kernel.argtypes = [ctypes.c_int]
lib.sdn_field_persistent_workgroups.restype = None
lib.sdn_field_persistent_workgroups.argtypes = [ctypes.c_int]
for _name, _args in PROTOTYPES.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = ctypes.c_int
for _name, _args in PROTOTYPES_U32.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = ctypes.c_uint32
for _name, _args in PROTOTYPES_U64.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = ctypes.c_uint64


class SdnError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported configuration", -3: "timed out waiting for the device"}.get(rc, f"hipError_t {rc}")
        raise SdnError(f"{what} failed: {kind}")


class HostMailbox:
    """Coherent, device-mapped host memory for the frame drivers' read-back (sdn_host_mailbox_alloc): 32 bytes per ray group."""

    def __init__(self, groups=1):
        self.ptr = lib.sdn_host_mailbox_alloc(int(groups))
        if not self.ptr:
            raise SdnError("sdn_host_mailbox_alloc failed")

    def data_ptr(self):
        return self.ptr

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                lib.sdn_host_mailbox_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


_device_seen = False


def require_device():
    global _device_seen
    if _device_seen:
        return
    if not torch.cuda.is_available():
        raise SdnError("SealD-NeRF HIP operators need a ROCm device (torch.cuda.is_available() is False); "
                       "there is no CPU fallback")
    _device_seen = True


def to_device(t):
    """The reference silently moves CPU inputs with .cuda() (raymarching.py:34-35); same here."""
    require_device()
    return t if t.is_cuda else t.cuda()


def ptr(t, dtype=None, name="tensor"):
    """data_ptr of a contiguous device tensor, validated (the reference checks nothing in raymarching)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise SdnError(f"{name} must be a device tensor")
    if not t.is_contiguous():
        raise SdnError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise SdnError(f"{name} must be {dtype}, got {t.dtype}")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device (the raw getter skips building a torch.cuda.Stream object: the
    wrappers call this once per launch, and the reference-shaped loops are bound by exactly this kind of host work)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def dtype_id(dt):
    if dt == torch.float32:
        return SDN_F32
    if dt == torch.float16:
        return SDN_F16
    raise SdnError(f"unsupported table dtype {dt} (the reference dispatches float/half; double is not built)")


# ---------------------------------------------------------------------------------------------------
# optional per-kernel HIP-event timing (used by bench.py for the roofline object; off by default)
# ---------------------------------------------------------------------------------------------------
class KernelTimers:
    """Collects (start, end) HIP events around native launches, keyed by kernel family, together with the
    number of work units (points) each launch processed.  Events are recorded on torch's current stream,
    which is the stream the launch goes to.  Nothing is synchronised until `summary()`."""

    def __init__(self):
        self.records = {}

    def record(self, name, units):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        self.records.setdefault(name, []).append((s, e, units))
        return s, e

    def raw_pair(self, name, units):
        """Pair of events whose hipEvent_t handles are handed to a native entry point that records them itself."""
        s, e = self.record(name, units)
        cur = torch.cuda.current_stream()
        s.record(cur); e.record(cur)  # materialises the handles; the native call re-records them in place
        return s.cuda_event, e.cuda_event

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _ in recs]
            units = [u for _, _, u in recs]
            out[name] = {"launches": len(recs), "total_ms": float(sum(ms)), "avg_ms": float(sum(ms) / len(ms)),
                         "units": int(sum(units)), "avg_units": float(sum(units) / len(units))}
        return out


timers = None  # set to a KernelTimers instance to enable


_launch_log = None  # a list while a `launch_log` context is open


class launch_log:
    """`with launch_log(out): ...` appends (kernel family, units) of every native launch that goes through `timed` to the list `out`
    (tests use it to see WHICH operators a caller ran -- e.g. that a reference-shaped loop took the fused dispatch)."""

    def __init__(self, out):
        self.out = out

    def __enter__(self):
        global _launch_log
        self.prev, _launch_log = _launch_log, self.out
        return self.out

    def __exit__(self, *exc):
        global _launch_log
        _launch_log = self.prev
        return False


class timed:
    """`with timed("grid_encode_fwd", B): launch(...)` -- no-op unless `sdn_backend.timers` is set."""

    def __init__(self, name, units):
        if _launch_log is not None:
            _launch_log.append((name, units))
        self.ev = timers.record(name, units) if timers is not None else None

    def __enter__(self):
        if self.ev:
            self.ev[0].record()

    def __exit__(self, *exc):
        if self.ev:
            self.ev[1].record()
        return False
