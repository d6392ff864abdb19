"""Multiresolution hash / tiled grid encoder on libsdn_hip (MI355X).

Same API as /root/reference/gridencoder/grid.py: `grid_encode` (autograd Function,
:24-93) and `GridEncoder` (nn.Module, :96-161) with identical constructor arguments,
parameter / buffer names (`embeddings`, `offsets`) and initialisation, so reference
checkpoints load unchanged.
"""
import weakref

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import sdn_backend as _sdn
from sdn_backend import lib as _lib, check as _check, ptr as _ptr, stream as _stream, dtype_id as _dtype_id, require_device

_gridtype_to_id = {"hash": 0, "tiled": 1}
_interp_to_id = {"linear": 0, "smoothstep": 1}

def _host_offsets(offsets):
    """The kernels take the level offsets by value (SGPRs).  `offsets` is a small registered buffer that
    never changes after construction; its host copy is cached ON the tensor object (one D2H copy per
    tensor, none per call) together with the version counter that invalidates it on in-place writes."""
    hit = getattr(offsets, "_sdn_host", None)
    if hit is None or hit[2] != offsets._version:
        arr = np.ascontiguousarray(offsets.detach().cpu().numpy().astype(np.int32))
        hit = (arr, arr.ctypes.data, offsets._version)
        offsets._sdn_host = hit
    return hit[0], hit[1]


_SORT_MIN_BATCH = 1 << 20   # backward: batches at least this large are scattered in Morton order (see _grid_encode.backward)


def _deterministic():
    import os
    return os.environ.get("SDN_DETERMINISTIC", "0") == "1" or torch.are_deterministic_algorithms_enabled()


# Inference on a table that does not change between calls (every iteration of a render loop, every slice of a density-grid update):
# the fp16 forward of the dnerf geometry (D = 3, C = 2, 16 tiled levels, linear, no align_corners) reads a QUAD copy of the table --
# one 16-byte block per row = the four (x, y) corners of a cell, two gathers per (point, level) instead of four -- built once per
# table version (address + torch version counter) on the SECOND call that sees it, and bit-identical to the plain kernel
# (csrc/gridencoder.hip k_grid_fwd_quad).  It also replaces the reference's per-call `.half()` cast of the whole table (grid.py:43-44).
_QUAD_TABLES = {}


quad_forward = True        # (measurements: False keeps every forward on the plain kernel)


def _quad_table_for(embeddings, offsets_host, off_ptr, S, H):
    if not quad_forward:
        return None
    key = (embeddings.data_ptr(), str(embeddings.device))
    ent = _QUAD_TABLES.get(key)
    # (the tensor OBJECT too, not only its address and version: the allocator hands a freed table's address to the next model's table,
    #  and two fresh tables can carry the same version number)
    if ent is None or ent["ref"]() is not embeddings or ent["version"] != embeddings._version or ent["rows"] != embeddings.shape[0]:
        if len(_QUAD_TABLES) >= 4:
            _QUAD_TABLES.clear()
        _QUAD_TABLES[key] = {"ref": weakref.ref(embeddings), "version": embeddings._version, "rows": embeddings.shape[0], "seen": 1,
                             "quad": None, "bad": False}
        return None
    ent["seen"] += 1
    if ent["bad"]:
        return None
    if ent["quad"] is None:
        quad = torch.empty(embeddings.shape[0] + 32, 4, 2, dtype=torch.float16, device=embeddings.device)
        src = embeddings.detach()
        if src.dtype not in (torch.float32, torch.float16):
            src = src.float()
        src = src.contiguous()
        rc = _lib.sdn_field_build_quad_table(_ptr(src), _dtype_id(src.dtype), off_ptr, S, H, _ptr(quad), _stream())
        if rc != 0:            # a geometry the quad layout does not cover (a capped level that is neither dense nor a power of two)
            ent["bad"] = True
            return None
        ent["quad"] = quad
    return ent["quad"]


class _grid_encode(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0):
        """grid.py:27-63.  inputs [B,D] f32 in [0,1]; embeddings [rows,C]; offsets [L+1] int32 -> [B, L*C]."""
        require_device()
        inputs = inputs.contiguous()
        if inputs.dtype != torch.float32:
            inputs = inputs.float()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = int(base_resolution)
        # autocast handling of the reference (grid.py:41-44): half table iff autocast is on and C is even
        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        embeddings = embeddings.contiguous()
        dt = _dtype_id(embeddings.dtype)
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=embeddings.dtype)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype) if calc_grad_inputs else None
        _, off_ptr = _host_offsets(offsets)
        with _sdn.timed("grid_encode_fwd_f16" if dt == _sdn.SDN_F16 else "grid_encode_fwd_f32", B):
            _check(_lib.sdn_grid_encode_forward(_ptr(inputs, torch.float32, "inputs"), _ptr(embeddings, None, "embeddings"), off_ptr,
                                                _ptr(outputs), B, D, C, L, S, H, _ptr(dy_dx), int(gridtype), int(bool(align_corners)),
                                                int(interpolation), dt, _stream()), "grid_encode_forward")
        outputs = outputs.permute(1, 0, 2).reshape(B, L * C)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, H, gridtype, interpolation]
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        """grid.py:68-89: scatter-add into a zeroed table gradient (+ input gradient through dy_dx)."""
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation = ctx.dims
        if grad.dtype != embeddings.dtype:
            grad = grad.to(embeddings.dtype)
        # Very large batches are scattered in MORTON ORDER of their cells.  The table gradient is an order-free sum (gridencoder.cu:
        # 248-340: plain atomic adds), but the memory-side atomic units of the MI355X work at a rate set by the number of 64-byte
        # segments a wave-instruction touches (64 lanes in 64 rows ~17x slower than 64 lanes in one 256-B run).  With the points in
        # Z-order the 64 lanes of a wave sit in a small cube: on the levels whose cells are at least as large as that cube the kernel's
        # run aggregation merges the lanes that share a row and the rest coalesces; on the finer levels every lane still owns its row
        # (the nature of a hash grid: no ordering of the POINTS changes that).  One radix sort of B 30-bit keys + one permuted copy of
        # the operands.  Ray-ordered training batches (neighbouring samples of a ray share their coarse cells already; measured 2x
        # slower when re-sorted) and anything below _SORT_MIN_BATCH are left as they are.
        perm = None
        if B >= _SORT_MIN_BATCH and D == 3:
            q = (inputs.clamp(0, 1) * 1023.0).to(torch.int32)

            def spread(v):   # 10 bits -> every third bit
                v = (v | (v << 16)) & 0x030000FF
                v = (v | (v << 8)) & 0x0300F00F
                v = (v | (v << 4)) & 0x030C30C3
                return (v | (v << 2)) & 0x09249249
            key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
            perm = torch.sort(key).indices
            inputs = inputs[perm].contiguous()
            grad = grad.view(B, L * C)[perm]
            if dy_dx is not None:
                dy_dx = dy_dx[perm].contiguous()
        grad = grad.view(B, L, C).permute(1, 0, 2).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if dy_dx is not None else None
        _, off_ptr = _host_offsets(offsets)
        # deterministic mode (SDN_DETERMINISTIC=1 / torch.use_deterministic_algorithms(True)): the table gradient is summed in 64-bit
        # fixed point with integer atomics -- the same bits whatever order the hardware executes them in
        det = torch.empty(embeddings.numel(), dtype=torch.int64, device=embeddings.device) if _deterministic() else None
        with _sdn.timed("grid_encode_bwd_f16" if embeddings.dtype == torch.float16 else "grid_encode_bwd_f32", B):
            _check(_lib.sdn_grid_encode_backward_det(_ptr(grad), _ptr(inputs), off_ptr, _ptr(grad_embeddings), B, D, C, L, S, H, _ptr(dy_dx),
                                                     _ptr(grad_inputs), int(gridtype), int(bool(ctx.align_corners)), int(interpolation),
                                                     _dtype_id(embeddings.dtype), _ptr(det), _stream()), "grid_encode_backward")
        if dy_dx is not None:
            grad_inputs = grad_inputs.to(inputs.dtype)
            if perm is not None:
                grad_inputs = torch.empty_like(grad_inputs).index_copy_(0, perm, grad_inputs)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None


def _forward_on_quad_copy(inputs, embeddings, offsets, per_level_scale, base_resolution, gridtype, align_corners, interpolation):
    """The inference forward of the dnerf geometry under `-O` on the QUAD copy of the table (see _quad_table_for), or None."""
    if not (inputs.is_cuda and inputs.dim() == 2 and inputs.shape[1] == 3 and embeddings.dim() == 2 and embeddings.shape[1] == 2
            and offsets.shape[0] == 17 and int(gridtype) == 1 and not align_corners and int(interpolation) == 0 and embeddings.is_contiguous()
            and inputs.shape[0] > 0):
        return None
    S, H = float(np.log2(per_level_scale)), int(base_resolution)
    _, off_ptr = _host_offsets(offsets)
    quad = _quad_table_for(embeddings, None, off_ptr, S, H)
    if quad is None:
        return None
    inputs = inputs.contiguous()
    if inputs.dtype != torch.float32:
        inputs = inputs.float()
    B = inputs.shape[0]
    outputs = torch.empty(16, B, 2, device=inputs.device, dtype=torch.half)
    with _sdn.timed("grid_encode_fwd_f16", B):
        _check(_lib.sdn_grid_encode_forward_quad_f16(_ptr(inputs, torch.float32, "inputs"), _ptr(quad), off_ptr, _ptr(outputs), B, S, H,
                                                     _stream()), "grid_encode_forward_quad_f16")
    return outputs.permute(1, 0, 2).reshape(B, 32)


def grid_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False,
                interpolation=0):
    """grid.py:27-93 (`grid_encode = _grid_encode.apply`).  A call that can need no gradient -- grad mode off, or neither the inputs nor the
    table require one -- under fp16 autocast takes the QUAD-copy forward when the geometry has one; everything else is the autograd
    Function as before."""
    if (torch.is_autocast_enabled("cuda") and not calc_grad_inputs
            and not (torch.is_grad_enabled() and (inputs.requires_grad or embeddings.requires_grad))):
        require_device()
        out = _forward_on_quad_copy(inputs, embeddings, offsets, per_level_scale, base_resolution, gridtype, align_corners, interpolation)
        if out is not None:
            return out
    return _grid_encode.apply(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs, gridtype, align_corners,
                              interpolation)


class GridEncoder(nn.Module):
    """grid.py:96-161.  Level table sizes follow :118-127 exactly (rows rounded up to a multiple of 8,
    capped at 2**log2_hashmap_size); `embeddings` ~ U(-1e-4, 1e-4) (:138-140)."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False, interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.align_corners = align_corners

        offsets = []
        offset = 0
        self.max_params = 2 ** log2_hashmap_size
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            params_in_level = min(self.max_params, (resolution if align_corners else resolution + 1) ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self.register_buffer("offsets", torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners} interpolation={self.interpolation}")

    def forward(self, inputs, bound=1):
        """inputs [..., input_dim] in [-bound, bound] -> [..., num_levels*level_dim]  (grid.py:145-161)."""
        inputs = (inputs + bound) / (2 * bound)
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, inputs.requires_grad,
                              self.gridtype_id, self.align_corners, self.interp_id)
        return outputs.view(prefix_shape + [self.output_dim])

    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """grid.py:164-185 (kernel_grad_tv).  Not on the dnerf / SealD-NeRF path (no caller outside grid.py);
        out of scope for this build -- fails loudly rather than silently doing nothing."""
        raise NotImplementedError("grad_total_variation (TV regulariser) is outside the dynamic-NeRF rendering path")
