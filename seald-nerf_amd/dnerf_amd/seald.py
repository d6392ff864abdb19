"""SealD-NeRF teacher / student renderers on the HIP operators.

Host-side mirror of /root/reference/SealDNeRF/renderer.py:27-297.  Differences from the dnerf renderer are the
reference's own: `T_thresh` is an explicit argument defaulting to 1e-4 (:114,275), depth is NOT normalised (:203,284),
the time-slice index is kept in `self.time_frame` (:140), `weights_sum` is returned in training mode (:210), and a
"seal mapper" may be hooked between the marcher and the network (`map_to_origin`, :155-160,245-250) and after it
(`map_color`, :266-267).  The mapper itself (SealNeRF/seal_utils.py, built on pytorch3d / open3d / trimesh) is the next
row of the scope table; any object with those two methods can be plugged in here.
"""
import torch

import raymarching

from .network import NeRFNetwork


class SealDNeRFTeacher(NeRFNetwork):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.seal_mapper = None
        self.time_frame = None

    def init_mapper(self, mapper):
        """Plug an object offering map_to_origin(xyzs, dirs) -> (xyzs', dirs', mask) and map_color(xyzs, dirs, rgbs) -> rgbs."""
        self.seal_mapper = mapper

    def _mapped_field(self, xyzs, dirs, time, recolor):
        if self.seal_mapper is None:
            sigmas, rgbs, deform = self(xyzs, dirs, time)
            return self.density_scale * sigmas, rgbs, deform
        m_xyzs, m_dirs, mask = self.seal_mapper.map_to_origin(xyzs.view(-1, 3), dirs.view(-1, 3))
        sigmas, rgbs, deform = self(m_xyzs.view(xyzs.shape), m_dirs.view(dirs.shape), time)
        if recolor:  # the reference recolours in the inference branch only (:266-267; disabled in training, :183-185)
            rgbs[mask] = self.seal_mapper.map_color(m_xyzs[mask], m_dirs[mask], rgbs[mask]).to(rgbs.dtype)
        return self.density_scale * sigmas, rgbs, deform

    def run_cuda(self, rays_o, rays_d, time, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 T_thresh=1e-4, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer, self.min_near)
        if bg_color is None:
            bg_color = 1
        t = self.time_slice(time)
        self.time_frame = t
        results = {}
        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield[t], self.cascade,
                                                                    self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                    force_all_rays, dt_gamma, max_steps)
            sigmas, rgbs, deform = self._mapped_field(xyzs, dirs, time, recolor=False)
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            results["deform"] = deform
            results["weights_sum"] = weights_sum
        else:
            weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
            depth = torch.zeros(N, dtype=torch.float32, device=device)
            image = torch.zeros(N, 3, dtype=torch.float32, device=device)
            rays_alive = torch.arange(N, dtype=torch.int32, device=device)
            rays_t = nears.clone()
            bitfield = self.density_bitfield[t]
            step = 0
            while step < max_steps:
                n_alive = rays_alive.shape[0]
                if n_alive <= 0:
                    break
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound, bitfield,
                                                            self.cascade, self.grid_size, nears, fars, 128, perturb if step == 0 else False,
                                                            dt_gamma, max_steps)
                sigmas, rgbs, _ = self._mapped_field(xyzs, dirs, time, recolor=True)
                raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
                rays_alive = rays_alive[rays_alive >= 0]
                step += n_step
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        results["depth"] = depth.view(*prefix)  # raw accumulated depth: the Seal renderers do not normalise it
        results["image"] = image.view(*prefix, 3)
        return results


class SealDNeRFStudent(NeRFNetwork):
    """SealDNeRF/renderer.py:294-297: the student is a plain dnerf network / renderer."""
