"""Caller-side helpers of the rendering path ("next" rows of the scope table): ray generation and checkpoint loading.

`get_rays` mirrors nerf/utils.py:54-137 of the reference (same arguments, same result dict).  The full-frame case (N = -1) runs as
one HIP kernel (`sdn_get_rays`); the sampled cases (random pixels, patches, error-map importance sampling) are index bookkeeping
in torch followed by the same per-pixel arithmetic.
"""
import torch

import sdn_backend as _sdn
from sdn_backend import lib as _lib, check as _check, ptr as _ptr, stream as _stream


def _meshgrid(*args):
    return torch.meshgrid(*args, indexing="ij")


@torch.no_grad()
def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, patch_size=1):
    """poses [B,4,4] cam2world, intrinsics (fx, fy, cx, cy) -> {'rays_o','rays_d': [B,N,3] (+ 'inds', 'inds_coarse')}."""
    device = poses.device
    B = poses.shape[0]
    fx, fy, cx, cy = [float(v) for v in intrinsics]
    results = {}
    if N <= 0 and poses.is_cuda:
        rays_o = torch.empty(B, H * W, 3, dtype=torch.float32, device=device)
        rays_d = torch.empty(B, H * W, 3, dtype=torch.float32, device=device)
        p = poses.detach().to(torch.float32).contiguous()
        for b in range(B):
            with _sdn.timed("get_rays", H * W):
                _check(_lib.sdn_get_rays(_ptr(p[b]), fx, fy, cx, cy, int(H), int(W), _ptr(rays_o[b]), _ptr(rays_d[b]), _stream()), "get_rays")
        results["rays_o"], results["rays_d"] = rays_o, rays_d
        return results

    i, j = _meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device))
    i = i.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
    j = j.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
    if N > 0:
        N = min(N, H * W)
        if patch_size > 1:  # nerf/utils.py:82-100
            num_patch = N // (patch_size ** 2)
            inds_x = torch.randint(0, H - patch_size, size=[num_patch], device=device)
            inds_y = torch.randint(0, W - patch_size, size=[num_patch], device=device)
            inds = torch.stack([inds_x, inds_y], dim=-1)
            pi, pj = _meshgrid(torch.arange(patch_size, device=device), torch.arange(patch_size, device=device))
            offsets = torch.stack([pi.reshape(-1), pj.reshape(-1)], dim=-1)
            inds = (inds.unsqueeze(1) + offsets.unsqueeze(0)).view(-1, 2)
            inds = (inds[:, 0] * W + inds[:, 1]).expand([B, N])
        elif error_map is None:  # :102-104
            inds = torch.randint(0, H * W, size=[N], device=device).expand([B, N])
        else:  # :105-118
            inds_coarse = torch.multinomial(error_map.to(device), N, replacement=False)
            inds_x, inds_y = inds_coarse // 128, inds_coarse % 128
            sx, sy = H / 128, W / 128
            inds_x = (inds_x * sx + torch.rand(B, N, device=device) * sx).long().clamp(max=H - 1)
            inds_y = (inds_y * sy + torch.rand(B, N, device=device) * sy).long().clamp(max=W - 1)
            inds = inds_x * W + inds_y
            results["inds_coarse"] = inds_coarse
        i = torch.gather(i, -1, inds)
        j = torch.gather(j, -1, inds)
        results["inds"] = inds
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    directions = torch.stack((xs, ys, zs), dim=-1)
    directions = directions / torch.norm(directions, dim=-1, keepdim=True)
    rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    results["rays_o"], results["rays_d"] = rays_o, rays_d
    return results


def _numpy_data_globals():
    """(callable, pickled name) pairs of numpy's data-only reconstructors, under the module paths numpy 1.x and 2.x write."""
    import numpy as np
    try:
        from numpy._core import multiarray as ma
    except ImportError:                                  # numpy 1.x
        from numpy.core import multiarray as ma
    out = []
    for mod in ("numpy.core.multiarray", "numpy._core.multiarray"):
        out += [(ma.scalar, mod + ".scalar"), (ma._reconstruct, mod + "._reconstruct")]
    out += [np.ndarray, np.dtype]
    out += list({type(np.dtype(t)) for t in ("float64", "float32", "float16", "int64", "int32", "int16", "int8", "uint8", "bool")})
    return out


def load_reference_checkpoint(model, path, map_location="cpu", optimizer=None, lr_scheduler=None, scaler=None, ema=None, model_only=True):
    """Loads a checkpoint written by the reference's trainer (`Trainer.save_checkpoint`, nerf/utils.py:1033-1093) into a
    `dnerf_amd.network.NeRFNetwork`, the way `Trainer.load_checkpoint` (:1095-1154) does:

      {'epoch', 'global_step', 'stats', 'mean_count', 'mean_density', 'model': state_dict,            -- always
       'optimizer', 'lr_scheduler', 'scaler', 'ema'}                                                    -- `full=True` checkpoints
      ("best" checkpoints drop `density_grid` from the model entry, :1085-1086); a bare state dict is accepted too (:1107-1110).

    Parameter and buffer names are the reference's (`encoder.embeddings`, `deform_net.N.weight`, `density_bitfield`, ...), so the
    model entry is a plain name match with strict=False.  With model_only=False the optimizer / lr_scheduler / scaler / ema given
    are restored from their entries when present (a failure to restore one of them is reported, not raised, as in the reference).
    The file is read with `weights_only=True` (nothing in it is executed).  The reference's `stats['results']` /
    `stats['best_result']` hold numpy.float64 scalars (`PSNRMeter.measure()`, nerf/utils.py:1017-1018,1073-1075), which the
    weights-only unpickler refuses by default: the data-only numpy reconstructors (scalar / dtype / ndarray rebuild, under both the
    numpy 1.x and 2.x module paths) are allow-listed for this one read -- they build values from bytes and call nothing.  Any
    other global in the file is still refused, and the error then names it.
    Returns (missing_keys, unexpected_keys); the trainer bookkeeping is left in `load_reference_checkpoint.last`
    ({'epoch', 'global_step', 'stats', 'restored': [...], 'failed': [...]})."""
    try:
        with torch.serialization.safe_globals(_numpy_data_globals()):
            blob = torch.load(path, map_location=map_location, weights_only=True)
    except Exception as exc:
        try:
            extra = sorted(torch.serialization.get_unsafe_globals_in_checkpoint(path))
        except Exception:
            extra = []
        raise RuntimeError(f"load_reference_checkpoint: {path} cannot be read by the weights-only loader"
                           + (f" (globals it would have to execute: {extra})" if extra else "") + f": {exc}") from exc
    info = {"epoch": None, "global_step": None, "stats": None, "restored": [], "failed": []}
    load_reference_checkpoint.last = info
    if not (isinstance(blob, dict) and "model" in blob):
        result = model.load_state_dict(blob)
        return result.missing_keys, result.unexpected_keys
    result = model.load_state_dict(blob["model"], strict=False)
    if ema is not None and "ema" in blob:
        ema.load_state_dict(blob["ema"])
        info["restored"].append("ema")
    if getattr(model, "cuda_ray", False):
        if "mean_count" in blob:
            model.mean_count = blob["mean_count"]
        if "mean_density" in blob:
            model.mean_density = blob["mean_density"]
    if not model_only:
        info.update(epoch=blob.get("epoch"), global_step=blob.get("global_step"), stats=blob.get("stats"))
        for name, obj in (("optimizer", optimizer), ("lr_scheduler", lr_scheduler), ("scaler", scaler)):
            if obj is not None and name in blob:
                try:
                    obj.load_state_dict(blob[name])
                    info["restored"].append(name)
                except Exception as exc:       # the reference logs "[WARN] Failed to load ..." and goes on
                    info["failed"].append((name, repr(exc)))
    return result.missing_keys, result.unexpected_keys
