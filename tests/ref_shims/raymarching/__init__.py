"""CPU stand-in for the reference's `raymarching` package (test infrastructure; see ../README.md).

Same call surface as raymarching/raymarching.py:19-372; every operator is evaluated by the CPU oracle.  `TRACE` collects one
`(n_alive, n_step, padded_points)` tuple per `march_rays` call and `LAST` keeps references to the tensors of the latest
`composite_rays` call so the fixture generator can read the per-ray state the reference's `run_cuda` does not return.
`NOISES` (optional) replaces the random per-ray offsets of `perturb=True` so that a fixture can be replayed exactly."""
import numpy as np
import torch
from torch.autograd import Function

from oracle import oracle as O

TRACE = []
LAST = {}
NOISES = {"train": None, "infer": None}


def _np(t, dtype=np.float32):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(dtype, copy=False))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    n, f = O.near_far_from_aabb(_np(rays_o).reshape(-1, 3), _np(rays_d).reshape(-1, 3), _np(aabb), float(min_near))
    return _t(n), _t(f)


def sph_from_ray(rays_o, rays_d, radius):
    return _t(O.sph_from_ray(_np(rays_o).reshape(-1, 3), _np(rays_d).reshape(-1, 3), float(radius)))


def morton3D(coords):
    return _t(O.morton3D(_np(coords, np.int32)))


def morton3D_invert(indices):
    return _t(O.morton3D_invert(_np(indices, np.int32)))


def packbits(grid, thresh, bitfield=None):
    out = O.packbits(_np(grid), float(thresh))
    if bitfield is None:
        return _t(out)
    bitfield.copy_(_t(out))
    return bitfield


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                     align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
    counter = np.zeros(2, np.int32)
    xyzs, dirs, deltas, rays = O.march_rays_train(_np(rays_o), _np(rays_d), float(bound), _np(density_bitfield, np.uint8), int(C), int(H),
                                                  _np(nears), _np(fars), counter, int(mean_count), bool(perturb), int(align),
                                                  bool(force_all_rays), float(dt_gamma), int(max_steps), noises=NOISES["train"])
    if step_counter is not None:
        step_counter.copy_(_t(counter))
    return _t(xyzs), _t(dirs), _t(deltas), _t(rays)


class _CompositeTrain(Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        ws, depth, image = O.composite_rays_train_forward(_np(sigmas), _np(rgbs), _np(deltas), _np(rays, np.int32), float(T_thresh))
        ws, depth, image = _t(ws), _t(depth), _t(image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, ws, image)
        ctx.T_thresh = float(T_thresh)
        return ws, depth, image

    @staticmethod
    def backward(ctx, grad_ws, grad_depth, grad_image):
        sigmas, rgbs, deltas, rays, ws, image = ctx.saved_tensors
        gs, gc = O.composite_rays_train_backward(_np(grad_ws), _np(grad_image), _np(sigmas), _np(rgbs), _np(deltas), _np(rays, np.int32),
                                                 _np(ws), _np(image), ctx.T_thresh)
        return _t(gs), _t(gc), None, None, None


composite_rays_train = _CompositeTrain.apply


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1, perturb=False,
               dt_gamma=0, max_steps=1024):
    xyzs, dirs, deltas = O.march_rays(int(n_alive), int(n_step), _np(rays_alive, np.int32), _np(rays_t), _np(rays_o), _np(rays_d), float(bound),
                                      _np(density_bitfield, np.uint8), int(C), int(H), _np(near), _np(far), int(align), bool(perturb),
                                      float(dt_gamma), int(max_steps), noises=NOISES["infer"] if perturb else None)
    TRACE.append((int(n_alive), int(n_step), int(xyzs.shape[0])))
    return _t(xyzs), _t(dirs), _t(deltas)


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """Mutates rays_alive, rays_t, weights_sum, depth, image (torch CPU tensors share memory with the numpy views)."""
    views = [t.numpy() for t in (rays_alive, rays_t, weights_sum, depth, image)]
    O.composite_rays(int(n_alive), int(n_step), views[0], views[1], _np(sigmas), _np(rgbs), _np(deltas), views[2], views[3], views[4],
                     float(T_thresh))
    LAST.update(weights_sum=weights_sum, depth=depth, image=image, rays_t=rays_t)
    return ()
