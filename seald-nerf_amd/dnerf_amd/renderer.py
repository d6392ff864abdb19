"""Volume renderer of the dynamic-NeRF path on the HIP operators.

Host-side mirror of /root/reference/dnerf/renderer.py:61-590 (`NeRFRenderer`): same constructor
arguments, registered buffers (`aabb_train`, `aabb_infer`, `density_grid`, `density_bitfield`, `times`,
`step_counter`) and the same `render / run_cuda / run / update_extra_state` semantics, so checkpoints
and callers carry over.  `render_frame` is the MI355X-native inference loop built on the same
operators: it produces the same image / depth bit for bit (per-ray results do not depend on the
compaction schedule) with one 4-byte read-back per iteration and no boolean-mask kernels.
"""
import math

import numpy as np
import torch
import torch.nn as nn

import raymarching


def _meshgrid(*args):
    return torch.meshgrid(*args, indexing="ij")


def sample_pdf(bins, weights, n_samples, det=False):
    """Inverse-CDF resampling of the coarse samples (dnerf/renderer.py:12-46; the NeRF formulation)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0.0 + 0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, device=weights.device)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples], device=weights.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    inds_g = torch.stack([below, above], -1)
    shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1):
        super().__init__()
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.time_size = 64
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius
        aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())
        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros(self.time_size, self.cascade, self.grid_size ** 3))
            self.register_buffer("density_bitfield", torch.zeros(self.time_size, self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self._mean_density, self._mean_density_dev, self._density_updater = 0, None, None
            self.iter_density = 0
            times = ((torch.arange(self.time_size, dtype=torch.float32) + 0.5) / self.time_size).view(-1, 1, 1)
            self.register_buffer("times", times)
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0

    @property
    def mean_density(self):
        """Python float as in the reference; after a native update it is read from the device only when somebody asks."""
        if getattr(self, "_mean_density_dev", None) is not None:
            self._mean_density = float(self._mean_density_dev[0])
            self._mean_density_dev = None
        return self._mean_density

    @mean_density.setter
    def mean_density(self, value):
        self._mean_density, self._mean_density_dev = value, None

    def use_native_density_update(self, field=None, fp32=False):
        """Route update_extra_state through csrc/density.hip + the fused field kernel (`-O` numerics; dnerf_amd/fused.py); fp32=True:
        through the fp32 fused kernel instead (a model trained without -O: 1e-4 from the op-by-op fp32 network)."""
        from .fused import DensityGridUpdater
        self._density_updater = DensityGridUpdater(self, field, fp32=fp32)
        return self._density_updater

    def forward(self, x, d, t):
        raise NotImplementedError()

    def density(self, x, t):
        raise NotImplementedError()

    def color(self, x, d, t, mask=None, **kwargs):
        raise NotImplementedError()

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    def _bg(self, rays_o, rays_d, bg_color):
        """Background colour the rays end on: the background-sphere model when bg_radius > 0 (dnerf/renderer.py:237-239,277-279),
        else the caller's colour, else white."""
        if self.bg_radius > 0:
            sph = raymarching.sph_from_ray(rays_o, rays_d, self.bg_radius)        # [N,2] in [-1,1]
            return self.background(sph, rays_d)
        return 1 if bg_color is None else bg_color

    def time_slice(self, time):
        """Index of the density-grid time slice for `time` [B,1] (dnerf/renderer.py:285)."""
        return torch.floor(time[0][0] * self.time_size).clamp(min=0, max=self.time_size - 1).long()

    # ------------------------------------------------------------------------------------------
    # occupancy-grid path (dnerf/renderer.py:261-386)
    # ------------------------------------------------------------------------------------------
    def run_cuda(self, rays_o, rays_d, time, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 T_thresh=None, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer, self.min_near)
        bg_color = self._bg(rays_o, rays_d, bg_color)
        t = self.time_slice(time)
        results = {}
        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            # (index_select with the device index: `density_bitfield[t]` with a 0-dim device tensor reads t back to the host)
            bitfield = self.density_bitfield.index_select(0, t.reshape(1))[0]
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, bitfield, self.cascade,
                                                                    self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                    force_all_rays, dt_gamma, max_steps)
            sigmas, rgbs, deform = self(xyzs, dirs, time)
            sigmas = self.density_scale * sigmas
            args = (sigmas, rgbs, deltas, rays) if T_thresh is None else (sigmas, rgbs, deltas, rays, T_thresh)
            weights_sum, depth, image = raymarching.composite_rays_train(*args)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
            results["deform"] = deform
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
                sigmas, rgbs, _ = self(xyzs, dirs, time)
                sigmas = self.density_scale * sigmas
                cargs = (n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
                raymarching.composite_rays(*(cargs if T_thresh is None else cargs + (T_thresh,)))
                rays_alive = rays_alive[rays_alive >= 0]
                step += n_step
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        results["depth"] = depth.view(*prefix)
        results["image"] = image.view(*prefix, 3)
        return results

    # ------------------------------------------------------------------------------------------
    # uniform sampler (dnerf/renderer.py:129-258): no occupancy grid, fixed num_steps (+ optional upsampling)
    # ------------------------------------------------------------------------------------------
    def run(self, rays_o, rays_d, time, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
        z_vals = torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0).expand((N, num_steps))
        z_vals = nears + (fars - nears) * z_vals
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device) - 0.5) * sample_dist

        def points(z):
            p = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z.unsqueeze(-1)
            return torch.min(torch.max(p, aabb[:3]), aabb[3:])

        def weights_of(z, sigma):
            deltas = torch.cat([z[..., 1:] - z[..., :-1], sample_dist * torch.ones_like(z[..., :1])], dim=-1)
            alphas = 1 - torch.exp(-deltas * self.density_scale * sigma)
            shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
            return deltas, alphas * torch.cumprod(shifted, dim=-1)[..., :-1]

        xyzs = points(z_vals)
        dens = {k: v.view(N, num_steps, -1) for k, v in self.density(xyzs.reshape(-1, 3), time).items()}
        if upsample_steps > 0:
            with torch.no_grad():
                deltas, weights = weights_of(z_vals, dens["sigma"].squeeze(-1))
                z_mid = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
                new_z = sample_pdf(z_mid, weights[:, 1:-1], upsample_steps, det=not self.training).detach()
                new_xyzs = points(new_z)
            new_dens = {k: v.view(N, upsample_steps, -1) for k, v in self.density(new_xyzs.reshape(-1, 3), time).items()}
            z_vals, z_index = torch.sort(torch.cat([z_vals, new_z], dim=1), dim=1)
            xyzs = torch.cat([xyzs, new_xyzs], dim=1)
            xyzs = torch.gather(xyzs, dim=1, index=z_index.unsqueeze(-1).expand_as(xyzs))
            for k in dens:
                tmp = torch.cat([dens[k], new_dens[k]], dim=1)
                dens[k] = torch.gather(tmp, dim=1, index=z_index.unsqueeze(-1).expand_as(tmp))
        _, weights = weights_of(z_vals, dens["sigma"].squeeze(-1))
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        dens = {k: v.view(-1, v.shape[-1]) for k, v in dens.items()}
        mask = weights > 1e-4
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **dens).view(N, -1, 3)
        weights_sum = weights.sum(dim=-1)
        ori_z = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
        depth = torch.sum(weights * ori_z, dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        bg_color = self._bg(rays_o, rays_d, bg_color)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3), "deform": dens["deform"]}

    # ------------------------------------------------------------------------------------------
    # density-grid maintenance (dnerf/renderer.py:453-555)
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        if not self.cuda_ray:
            return
        if getattr(self, "_density_updater", None) is not None:
            self._density_updater.refresh()
            self._mean_density_dev = self._density_updater.update(decay)
            self._update_step_counter()
            return
        dev = self.density_bitfield.device
        tmp_grid = -torch.ones_like(self.density_grid)
        half_time = 0.5 / self.time_size

        def query(coords, indices, t_idx, time, cas):
            xyzs = 2 * coords.float() / (self.grid_size - 1) - 1
            bound = min(2 ** cas, self.bound)
            half_grid = bound / self.grid_size
            cas_xyzs = xyzs * (bound - half_grid)
            cas_xyzs += (torch.rand_like(cas_xyzs) * 2 - 1) * half_grid
            time_perturb = time + (torch.rand_like(time) * 2 - 1) * half_time
            sigmas = self.density(cas_xyzs, time_perturb)["sigma"].reshape(-1).detach().float()
            tmp_grid[t_idx, cas, indices] = sigmas * self.density_scale

        if self.iter_density < 16:
            axes = [torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S) for _ in range(3)]
            for t_idx, time in enumerate(self.times):
                for xs in axes[0]:
                    for ys in axes[1]:
                        for zs in axes[2]:
                            xx, yy, zz = _meshgrid(xs, ys, zs)
                            coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                            indices = raymarching.morton3D(coords).long()
                            for cas in range(self.cascade):
                                query(coords, indices, t_idx, time, cas)
        elif self.iter_density < 100:
            N = self.grid_size ** 3 // 4
            for t_idx, time in enumerate(self.times):
                for cas in range(self.cascade):
                    coords = torch.randint(0, self.grid_size, (N, 3), device=dev)
                    indices = raymarching.morton3D(coords).long()
                    occ = torch.nonzero(self.density_grid[t_idx, cas] > 0).squeeze(-1)
                    if occ.shape[0] > 0:
                        occ = occ[torch.randint(0, occ.shape[0], [N], dtype=torch.long, device=dev)]
                        indices = torch.cat([indices, occ], dim=0)
                        coords = torch.cat([coords, raymarching.morton3D_invert(occ)], dim=0)
                    query(coords, indices, t_idx, time, cas)
        valid = (self.density_grid >= 0) & (tmp_grid >= 0)
        self.density_grid[valid] = torch.maximum(self.density_grid[valid] * decay, tmp_grid[valid])
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        self.iter_density += 1
        density_thresh = min(self.mean_density, self.density_thresh)
        for t_idx in range(self.time_size):
            raymarching.packbits(self.density_grid[t_idx], density_thresh, self.density_bitfield[t_idx])
        self._update_step_counter()

    def _update_step_counter(self):
        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    def render(self, rays_o, rays_d, time, staged=False, max_ray_batch=4096, **kwargs):
        """dnerf/renderer.py:558-590: rays [B,N,3] -> {image [B,N,3], depth [B,N]}; staging only without the grid."""
        _run = self.run_cuda if self.cuda_ray else self.run
        B, N = rays_o.shape[:2]
        device = rays_o.device
        if staged and not self.cuda_ray:
            depth = torch.empty((B, N), device=device)
            image = torch.empty((B, N, 3), device=device)
            for b in range(B):
                for head in range(0, N, max_ray_batch):
                    tail = min(head + max_ray_batch, N)
                    r = _run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], time[b:b + 1], **kwargs)
                    depth[b:b + 1, head:tail] = r["depth"]
                    image[b:b + 1, head:tail] = r["image"]
            return {"depth": depth, "image": image}
        return _run(rays_o, rays_d, time, **kwargs)


# ==================================================================================================
# MI355X-native inference loop
# ==================================================================================================
class FrameWorkspace:
    """Per-(N) buffers of the native loop, allocated once and reused across frames: the reference
    re-allocates and memsets three M-sized sample buffers every iteration."""

    def __init__(self, N, device):
        self.N = N
        f32, i32 = torch.float32, torch.int32
        M = N + 128 + 8 * 128  # >= n_alive * n_step + align for every schedule (n_alive*n_step <= N)
        self.xyzs = torch.empty(M, 3, dtype=f32, device=device)
        self.dirs = torch.empty(M, 3, dtype=f32, device=device)
        self.deltas = torch.empty(M, 2, dtype=f32, device=device)
        self.alive = [torch.empty(N, dtype=i32, device=device), torch.empty(N, dtype=i32, device=device)]
        self.rays_t = torch.empty(N, dtype=f32, device=device)
        self.weights_sum = torch.empty(N, dtype=f32, device=device)
        self.depth = torch.empty(N, dtype=f32, device=device)
        self.image = torch.empty(N, 3, dtype=f32, device=device)
        self.count = torch.zeros(1, dtype=i32, device=device)
        self.count_host = torch.zeros(1, dtype=i32).pin_memory()
        from sdn_backend import lib
        self.scratch = torch.empty(max(int(lib.sdn_compact_alive_scratch_bytes(N)), 4), dtype=torch.uint8, device=device)
        self.cull = torch.empty(int(lib.sdn_cull_grid_bytes()), dtype=torch.uint8, device=device)
        self.live_idx = torch.empty(M, dtype=i32, device=device)          # slots that received a sample, per iteration
        self.live_counts = torch.zeros(1024 + 8, dtype=i32, device=device)  # one counter per loop iteration (<= max_steps)


@torch.no_grad()
def render_frame(model, rays_o, rays_d, time, fp16=False, dt_gamma=0.0, max_steps=1024, T_thresh=1e-2, bg_color=1.0,
                 workspace=None, field=None, count_samples=True, use_cull=True, mapper=None):
    """One inference frame: rays [N,3] -> {'image' [N,3], 'depth' [N], 'weights_sum' [N], 'n_samples', 'trace'}.

    Same schedule as dnerf/renderer.py:340-381 (n_step = clamp(N // n_alive, 1, 8), stop at max_steps),
    same operators, but: buffers come from a reusable workspace, alive-ray compaction is the device-side
    `compact_alive` (one 4-byte count read back per iteration instead of a boolean-mask select), and the
    field network may be the fused MFMA kernel (`field`, see dnerf_amd/fused.py) instead of the op-by-op
    network.  `trace` lists (n_alive, n_step, padded_points) per iteration.

    With the fused field the marcher also appends the slots that received a sample to a compact list and the network is
    evaluated on those only (the reference evaluates every padded slot, including the ~94 % empty ones of the first
    iteration); rays whose remaining segment provably cannot produce a sample are retired by the marcher's exact
    cull-grid test instead of stepping through the empty volume voxel by voxel.

    `mapper`: an optional SealD seal mapper (`dnerf_amd.seal_mapper`), hooked exactly where `SealDNeRF/renderer.py:245-267` hooks
    it -- sample positions / directions are mapped back to their origin before the field is evaluated, colours of the mapped
    samples are re-mapped after it -- so an edited scene renders through the fused field as well.
    """
    from sdn_backend import lib, check, ptr, stream
    device = rays_o.device
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N = rays_o.shape[0]
    ws = workspace if workspace is not None and workspace.N == N else FrameWorkspace(N, device)
    nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, model.aabb_infer, model.min_near)
    bitfield = model.density_bitfield[model.time_slice(time)]
    ws.weights_sum.zero_(); ws.depth.zero_(); ws.image.zero_()
    ws.rays_t.copy_(nears)
    torch.arange(N, dtype=torch.int32, device=device, out=ws.alive[0])
    cur = 0
    n_alive = N
    step = 0
    trace = []
    use_list = field is not None
    n_samples = torch.zeros((), dtype=torch.int64, device=device) if (count_samples and not use_list) else None
    st = stream()
    evaluate = field if field is not None else _field_eval(model, time, fp16)
    cull = None
    if use_cull and model.grid_size == 128 and model.cascade == 1:
        check(lib.sdn_build_cull_grid(ptr(bitfield), 128, ptr(ws.cull), st), "build_cull_grid")
        cull = ptr(ws.cull)
    if use_list:
        ws.live_counts.zero_()
    it = 0
    while step < max_steps and n_alive > 0:
        n_step = max(min(N // n_alive, 8), 1)
        M0 = n_alive * n_step
        M = M0 + (128 - M0 % 128)
        xyzs, dirs, deltas = ws.xyzs[:M], ws.dirs[:M], ws.deltas[:M]
        alive = ws.alive[cur]
        live_count = ws.live_counts[it:it + 1] if use_list else None
        check(lib.sdn_march_rays_ex(n_alive, n_step, ptr(alive), ptr(ws.rays_t), ptr(rays_o), ptr(rays_d), float(model.bound),
                                    float(dt_gamma), int(max_steps), int(model.cascade), int(model.grid_size), ptr(bitfield), ptr(fars),
                                    ptr(xyzs), ptr(dirs), ptr(deltas), None, M, cull, ptr(ws.live_idx) if use_list else None,
                                    ptr(live_count), st), "march_rays_ex")
        if n_samples is not None:  # ops path: live samples = slots with a non-zero step (bookkeeping, off in the timed loop)
            n_samples += (deltas[:M0, 0] > 0).sum()
        q_xyzs, q_dirs, mapped = xyzs, dirs, None
        native_map = mapper is not None and hasattr(mapper, "map_to_origin_") and mapper._native_ok(xyzs, dirs)
        if native_map:      # in place on the sample buffers (the marcher rewrites them every iteration)
            mapper.map_data_conversion(xyzs)
            mapped = mapper.map_to_origin_(xyzs, dirs)
        elif mapper is not None:
            q_xyzs, q_dirs, mapped = mapper.map_to_origin(xyzs, dirs)
            q_xyzs, q_dirs = q_xyzs.contiguous(), q_dirs.contiguous()
        if use_list:
            sigmas, rgbs = evaluate(q_xyzs, q_dirs, ws.live_idx, live_count)
        else:
            sigmas, rgbs = evaluate(q_xyzs, q_dirs)
        if native_map:
            mapper.map_color_(rgbs, mapped)
        elif mapped is not None and bool(mapped.any()):
            rgbs[mapped] = mapper.map_color(q_xyzs[mapped], q_dirs[mapped], rgbs[mapped]).to(rgbs.dtype)
        check(lib.sdn_composite_rays(n_alive, n_step, float(T_thresh), ptr(alive), ptr(ws.rays_t), ptr(sigmas), ptr(rgbs), ptr(deltas),
                                     ptr(ws.weights_sum), ptr(ws.depth), ptr(ws.image), st), "composite_rays")
        check(lib.sdn_compact_alive(ptr(alive), n_alive, ptr(ws.alive[1 - cur]), ptr(ws.count), ptr(ws.scratch), st), "compact_alive")
        ws.count_host.copy_(ws.count, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        trace.append((n_alive, n_step, M))
        n_alive = int(ws.count_host[0])
        cur = 1 - cur
        step += n_step
        it += 1
    if getattr(model, "bg_radius", -1) > 0:
        bg_color = model._bg(rays_o, rays_d, None).float()
    image = ws.image + (1 - ws.weights_sum).unsqueeze(-1) * bg_color
    depth = torch.clamp(ws.depth - nears, min=0) / (fars - nears)
    if count_samples:
        total = int(ws.live_counts[:it].sum().item()) if use_list else int(n_samples.item())
    else:
        total = None
    return {"image": image, "depth": depth, "weights_sum": ws.weights_sum.clone(), "trace": trace, "nears": nears, "fars": fars,
            "n_samples": total}


def _field_eval(model, time, fp16):
    """Op-by-op field evaluation (the reference's network on the drop-in operators), fp32 outputs."""
    def run(xyzs, dirs):
        keep = model.__dict__.get("fused_inference")
        model.fused_inference = False          # this IS the op-by-op path: never the fused dispatch of NeRFNetwork.forward
        try:
            if fp16:
                with torch.autocast("cuda", dtype=torch.float16):
                    sigmas, rgbs, _ = model(xyzs, dirs, time)
            else:
                sigmas, rgbs, _ = model(xyzs, dirs, time)
        finally:
            if keep is None:
                del model.fused_inference
            else:
                model.fused_inference = keep
        return (model.density_scale * sigmas).float().contiguous(), rgbs.float().contiguous()
    return run


# ==================================================================================================
# device-driven loop (csrc/render.hip): no host round trip inside an iteration
# ==================================================================================================
class DeviceLoop:
    """Owns the buffers and the `SdnRenderCtx` of the device-driven inference loop for frames of N rays.

    The host enqueues iteration k+1 while iteration k runs; grids are sized with the survivor count read back (async,
    pinned memory) one iteration earlier, which is an upper bound of the current one.  Requires the fused field."""

    RING = 4
    MAX_TIMED = 24  # iterations whose fused-field launch can be timed in place (the headline frame has 11 + 1)

    def __init__(self, model, field, N, device, max_steps=1024, T_thresh=1e-2, dt_gamma=0.0, mailbox=True, mapper=None, frames=1,
                 keep_cull_grids=False):
        """frames > 1: a FRAME GROUP -- the N rays are `frames` equal, frame-major blocks (the shards of consecutive frames of a camera
        path / of successive time steps) rendered together by one loop, each at its own time (`bind(..., time=[t_0, ..., t_F-1])`):
        the chain of ~30 dependent launches of a loop is paid once per group instead of once per frame.  Per-ray results do not
        depend on which rays share a loop, so every frame of the group is bit-identical to the frame rendered alone."""
        import ctypes
        from sdn_backend import lib, SdnRenderCtx, MAX_GROUP_FRAMES
        f32, i32 = torch.float32, torch.int32
        self.model, self.field, self.N = model, field, N
        self.frames = int(frames)
        self.keep_cull_grids = bool(keep_cull_grids)
        if not 1 <= self.frames <= MAX_GROUP_FRAMES or N % self.frames:
            raise ValueError(f"a frame group holds 1..{MAX_GROUP_FRAMES} frames of equal ray count (N = {N}, frames = {frames})")
        M = N + 128 + 8 * 128
        n_counters = max_steps + 8
        z = lambda *shape, dt=f32: torch.empty(*shape, dtype=dt, device=device)  # noqa: E731
        self.buf = dict(alive_a=z(N, dt=i32), alive_b=z(N, dt=i32), rays_t=z(N), weights_sum=z(N), depth=z(N), image=z(N, 3),
                        xyzs=z(M, 3), dirs=z(M, 3), deltas=z(M, 2), sigmas=z(M), rgbs=z(M, 3), live_idx=z(M, dt=i32),
                        live_counts=torch.zeros(n_counters, dtype=i32, device=device), state=torch.zeros(16, dtype=i32, device=device),
                        trace=torch.zeros(2 * n_counters + 16, dtype=i32, device=device), n_out=torch.zeros(1, dtype=i32, device=device),
                        block_totals=z((N + 255) // 256 + 1, dt=i32), nears=z(N), fars=z(N), rays_tend=z(N),
                        cull_bits=torch.empty(self.frames * int(lib.sdn_cull_grid_bytes()), dtype=torch.uint8, device=device))
        if self.frames > 1:
            self.buf["slot_frame"] = torch.zeros(M, dtype=torch.uint8, device=device)
        self.image_out, self.depth_out = z(N, 3), z(N)
        self.snap = self.buf["trace"][2 * n_counters: 2 * n_counters + 8].view(4, 2)  # device ring written by the advance
        from sdn_backend import HostMailbox
        # mailbox=True: coherent mapped host memory the loop kernels publish each iteration's survivor count into (the driver polls
        # it); False: ordinary pinned memory, which selects the driver's event + side-stream copy read-back
        self.host_state = HostMailbox(1) if mailbox else torch.zeros(self.RING, 2, dtype=i32).pin_memory()
        self.events = [torch.cuda.Event() for _ in range(self.RING)]
        self.copy_events = [torch.cuda.Event() for _ in range(self.RING)]
        self.side = torch.cuda.Stream(device=device)
        self._handles = None
        self._timing_queue = []
        c = SdnRenderCtx()
        for k, v in self.buf.items():
            setattr(c, k, v.data_ptr())
        c.field_weights, c.field_bias0, c.grid_table = field.weights.data_ptr(), field.bias0.data_ptr(), field.table.data_ptr()
        c.grid_offsets = (ctypes.c_int32 * 17)(*[int(v) for v in field.offsets_host])
        c.grid_S, c.grid_H = field.S, field.H
        c.N, c.M_cap, c.n_counters, c.max_steps, c.C, c.H = N, M, n_counters, int(max_steps), int(model.cascade), int(model.grid_size)
        c.bound, c.dt_gamma, c.T_thresh, c.density_scale = float(model.bound), float(dt_gamma), float(T_thresh), float(model.density_scale)
        c.n_group_frames, c.rays_per_frame = (self.frames, N // self.frames) if self.frames > 1 else (0, 0)
        # the fp32 fused field (dnerf_amd.fused_f32.FusedFieldF32: the reference without -O) in the same loop: its packed floats, the
        # model's fp32 table in place, the reference's offsets
        c.field_f32 = (2 if getattr(field, "variant", "") == "split" else 1) if type(field).__name__ == "FusedFieldF32" else 0
        self.ctx = c
        self.max_steps = int(max_steps)
        self.set_mapper(mapper)

    def set_mapper(self, mapper):
        """Hooks a SealD bounding-box seal mapper (`dnerf_amd.seal_mapper.SealBBoxMapper`, or None) into every iteration of the native
        loop: samples are mapped back to their origin before the field kernel, colours of the mapped samples re-mapped after it."""
        import ctypes
        from sdn_backend import SdnSealBox
        c = self.ctx
        self.mapper = mapper
        if mapper is None:
            c.seal, c.seal_mask, self._seal = None, None, None
            return
        dev = self.buf["xyzs"].device
        mapper.map_data_conversion(self.buf["xyzs"])
        a = mapper._native_args(dev)
        if self.frames > 1 and ("map_source" in mapper.map_data or "rgb" in mapper.map_data):
            # both options look at ALL samples of a loop iteration (the mean brightness of the masked ones, "does the call map anything"):
            # in a frame group an iteration holds several frames' samples, which the reference never mixes
            raise NotImplementedError("mapSource / rgb tint depend on the set of samples of an iteration: render such edits one frame per loop")
        box = SdnSealBox()
        for k in range(6 * a["n_bounds"]):
            box.bounds[k] = a["bounds"][k]
        box.n_bounds, box.n_tris, box.tris = a["n_bounds"], a["n_tris"], a["tris"].data_ptr()
        for name, n in (("test_dir", 3), ("tinv", 12), ("rinv", 9), ("scale", 3), ("center", 3)):
            for k in range(n):
                getattr(box, name)[k] = a[name][k]
        if "hsv" in mapper.map_data:
            h = [float(v) for v in mapper.map_data["hsv"].reshape(-1).tolist()]
            box.hsv[0], box.hsv[1], box.hsv[2], box.modify_hsv = h[0], h[1], h[2], 1
        if "rgb" in a:
            box.rgb[0], box.rgb[1], box.rgb[2], box.rgb_light_offset, box.modify_rgb = a["rgb"][0], a["rgb"][1], a["rgb"][2], a["rgb_light_offset"], 1
        if "map_source" in a:
            for k in range(6):
                box.source_bound[k] = a["source_bound"][k]
            for k in range(3):
                box.map_source[k] = a["map_source"][k]
            box.has_map_source = 1
        box.scratch = a["scratch"].data_ptr()
        mask = torch.empty(self.buf["sigmas"].shape[0], dtype=torch.uint8, device=dev)
        self._seal = (box, mask, a)      # keep the record, the mask and the triangle tensor alive
        c.seal, c.seal_mask = ctypes.addressof(box), mask.data_ptr()

    def prepare_timing(self, frames):
        """Pre-creates (outside any timed region) the HIP event pairs for `frames` timed renders: MAX_TIMED pairs per frame,
        recorded by the native loop around each fused-field launch."""
        import ctypes
        cur = torch.cuda.current_stream()
        self._timing_queue = []
        for _ in range(frames):
            recs = []
            for _ in range(self.MAX_TIMED):
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(cur); e_.record(cur)  # materialises the handles; the native call re-records them in place
                recs.append((s_, e_, 0))
            arr = (ctypes.c_void_p * (2 * self.MAX_TIMED))(*[h for r in recs for h in (r[0].cuda_event, r[1].cuda_event)])
            self._timing_queue.append((arr, recs))
        torch.cuda.synchronize()

    def frame_time(self, time):
        """`SdnFrameTime` record (+ the tensors it points into) for `time`: one time stamp, or -- for a frame group -- one per
        frame.  Everything is derived from the VALUE of the time stamp(s): the occupancy slice (dnerf/renderer.py:285), the
        time-encoding bias of the first deform layer and the canonical-frame rule t == 0 (dnerf/network.py:130-141)."""
        from sdn_backend import SdnFrameTime
        model = self.model
        times = list(time) if isinstance(time, (list, tuple)) else [time]
        if len(times) == 1 and self.frames > 1:
            times = times * self.frames
        if len(times) != self.frames:
            raise ValueError(f"{self.frames} frame(s) in this loop, {len(times)} time stamp(s) given")
        bias, mask, slices = self.field.group_constants(times)
        ft = SdnFrameTime()
        bits = [model.density_bitfield[s_] for s_ in slices]          # views (no host sync: python int index)
        for f, b_ in enumerate(bits):
            ft.bitfield[f] = b_.data_ptr()
        ft.field_bias0, ft.zero_deform = bias.data_ptr(), mask
        culls = None
        if self.keep_cull_grids and int(model.grid_size) == 128 and int(model.cascade) == 1:
            culls = [self._cull_grid(s_) for s_ in slices]
            for f, g_ in enumerate(culls):
                ft.cull_grid[f] = g_.data_ptr()
        return ft, (bias, bits, culls)

    def _cull_grid(self, t_idx):
        """The marcher's coarse skip grid of one occupancy slice (`sdn_build_cull_grid`), kept until the density grid is updated
        (`model.iter_density` counts the updates) or `invalidate_cull_grids()` is called: a slice is rendered many times in between,
        and deriving its cull grid again is two launches at the head of every frame's latency chain (eight grids per frame group)."""
        import sdn_backend as B
        # The kept grid carries the slice's occupancy bits themselves (the packed fine-bit image the marchers copy to LDS), so a stale
        # entry means marching on stale OCCUPANCY: the epoch includes the bitfield's version counter -- load_state_dict, fill_bitfield
        # and reset_extra_state rewrite the bitfield in place without a new iter_density -- and the cache lives ON the model object
        # (an id()-keyed table would hand a new model the entry of a collected one whose id Python reused).
        cache = self.model.__dict__.get("_sdn_cull_cache")
        if cache is None:
            cache = {"epoch": None, "grids": {}}
            self.model.__dict__["_sdn_cull_cache"] = cache
        bf = self.model.density_bitfield
        epoch = (self.model.iter_density, bf.data_ptr(), bf._version)
        if cache["epoch"] != epoch:
            cache["epoch"], cache["grids"] = epoch, {}
        hit = cache["grids"].get(t_idx)
        if hit is None:
            hit = torch.empty(int(B.lib.sdn_cull_grid_bytes()), dtype=torch.uint8, device=self.model.density_bitfield.device)
            B.check(B.lib.sdn_build_cull_grid(self.model.density_bitfield[t_idx].data_ptr(), 128, hit.data_ptr(), B.stream()), "build_cull_grid")
            cache["grids"][t_idx] = hit
        return hit

    @staticmethod
    def invalidate_cull_grids(model=None):
        """Forget the kept cull grids of `model`: only needed after writing `density_bitfield` behind torch's back (a raw pointer, DLPack,
        a custom kernel) -- torch-level writes (`update_extra_state`, `fill_bitfield`, `load_state_dict`, `copy_`) bump the tensor's
        version counter and are seen by themselves."""
        if model is not None:
            model.__dict__.pop("_sdn_cull_cache", None)

    def bind(self, rays_o, rays_d, time):
        """Points the context at this frame's rays and time constants (the native driver launches near_far_from_aabb itself)."""
        import ctypes
        from sdn_backend import ptr
        c, model = self.ctx, self.model
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        assert rays_o.shape[0] == self.N
        nears, fars = self.buf["nears"], self.buf["fars"]  # filled by the native driver (ctx.aabb / ctx.min_near)
        c.aabb, c.min_near = ptr(model.aabb_infer, torch.float32, "aabb_infer"), float(model.min_near)
        ft, refs = self.frame_time(time)
        c.rays_o, c.rays_d = ptr(rays_o, torch.float32, "rays_o"), ptr(rays_d, torch.float32, "rays_d")
        c.bitfield, c.field_bias0, c.zero_deform = ft.bitfield[0], ft.field_bias0, int(ft.zero_deform)
        for f in range(self.frames):
            c.frame_bitfield[f] = ft.bitfield[f]
            c.frame_cull[f] = ft.cull_grid[f]
        self._frame_refs = (rays_o, rays_d, nears, fars, refs)  # keep the tensors alive while the frame is in flight
        cur = torch.cuda.current_stream()
        if self._handles is None:  # materialise raw hipEvent_t / hipStream_t handles once
            for e in self.events + self.copy_events:
                e.record(cur)
            self._ev_main = (ctypes.c_void_p * self.RING)(*[e.cuda_event for e in self.events])
            self._ev_copy = (ctypes.c_void_p * self.RING)(*[e.cuda_event for e in self.copy_events])
            self._iters = ctypes.c_uint32(0)
            self._handles = True
        return nears, fars

    def collect(self, nears, fars, want_stats):
        out = {"image": self.image_out, "depth": self.depth_out, "weights_sum": self.buf["weights_sum"], "nears": nears, "fars": fars}
        if want_stats:
            iters = int(self.buf["state"][3].item())
            tr = self.buf["trace"][: 2 * iters].cpu().view(-1, 2).tolist()
            out["trace"] = [(a, s, a * s + (128 - (a * s) % 128)) for a, s in tr]
            out["n_samples"] = int(self.buf["live_counts"][:iters].sum().item())
        return out

    @torch.no_grad()
    def render(self, rays_o, rays_d, time, bg_color=1.0, want_stats=True):
        import ctypes
        from sdn_backend import lib, check, ptr, stream
        import sdn_backend
        nears, fars = self.bind(rays_o, rays_d, time)
        bg_model = getattr(self.model, "bg_radius", -1) > 0
        if bg_model:         # the background-sphere colours are mixed in after the loop (the native finish takes one scalar colour)
            bg_color = 0.0
        cur = torch.cuda.current_stream()
        ev_field, n_ev, recs = None, 0, None
        if sdn_backend.timers is not None and self._timing_queue:  # bench: time the fused-field launches in place
            ev_field, recs = self._timing_queue.pop()
            n_ev = self.MAX_TIMED
        self.side.wait_stream(cur)
        check(lib.sdn_render_frame_f16(ctypes.byref(self.ctx), float(bg_color), ptr(self.image_out), ptr(self.depth_out), stream(),
                                       self.side.cuda_stream, self._ev_main, self._ev_copy, self.host_state.data_ptr(), ev_field, n_ev,
                                       ctypes.byref(self._iters)), "render_frame_f16")
        if recs is not None:  # keep the pairs of the iterations that ran
            sdn_backend.timers.records.setdefault("field_forward_f16", []).extend(recs[: min(n_ev, int(self._iters.value))])
        if bg_model:
            ro, rd = self._frame_refs[0], self._frame_refs[1]
            self.image_out += (1 - self.buf["weights_sum"]).unsqueeze(-1) * self.model._bg(ro, rd, None).float()
        return self.collect(nears, fars, want_stats)


def _context_streams(k, device):
    """One HIP stream per loop context, created here with `hipStreamCreateWithPriority` rather than taken from torch's pool: which
    hardware queue a pool stream lands on depends on how many of the pool's streams the process has touched before, and a context
    that shares its queue with another stream loses its overlap (measured, 4 contexts: 0.437 ms per frame with 5-6 hardware queues,
    0.52 with 7, 8 or 16, 0.49 with 3-4 from the pool; own streams: 0.435-0.439 for 5..9 queues -- profiles/r03_hw_queues.txt).  The
    process still needs GPU_MAX_HW_QUEUES >= contexts + 1 (set before the HIP runtime starts; bench.py does).
    SDN_CTX_PRIORITIES="p0,p1,..." (HIP priorities, -1 high / 0 normal / 1 low, used in turn; measured: no effect);
    SDN_CTX_STREAMS=torch falls back to the pool.
    One set of streams per (device, k, priorities) is created and REUSED by every later loop object of the process: raw HIP streams are
    never destroyed by torch's ExternalStream wrapper, and each leaked set would shift the queue assignment of the next -- the very
    sensitivity this function removes.  (Under rocprofv3 the profiler's preloaded library starts the HIP runtime before bench.py can set
    GPU_MAX_HW_QUEUES: export it in the environment of such a run.)"""
    import os
    if os.environ.get("SDN_CTX_STREAMS", "") == "torch":
        return [torch.cuda.Stream(device=device) for _ in range(k)]
    import ctypes
    spec = os.environ.get("SDN_CTX_PRIORITIES", "").strip() or "0"
    try:
        prios = [int(x) for x in spec.split(",")]
    except ValueError:
        raise ValueError(f"SDN_CTX_PRIORITIES={spec!r}: expected comma-separated integers (HIP stream priorities, e.g. '-1,0,0,0')") from None
    if any(p < -1 or p > 1 for p in prios):
        raise ValueError(f"SDN_CTX_PRIORITIES={spec!r}: HIP stream priorities are -1 (high), 0, 1 (low)")
    key = (str(torch.device(device)), int(k), tuple(prios))
    if key in _CONTEXT_STREAMS:
        return _CONTEXT_STREAMS[key]
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        return [torch.cuda.Stream(device=device) for _ in range(k)]
    out = []
    with torch.cuda.device(device):
        for i in range(k):
            h = ctypes.c_void_p()
            rc = hip.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(prios[i % len(prios)]))   # 1 = hipStreamNonBlocking
            if rc != 0:
                raise RuntimeError(f"hipStreamCreateWithPriority failed ({rc})")
            out.append(torch.cuda.ExternalStream(h.value, device=device))
    _CONTEXT_STREAMS[key] = out
    return out


_CONTEXT_STREAMS = {}


class PipelinedDeviceLoop:
    """A stream of frames (camera path / time steps of one model) through `contexts` `DeviceLoop` contexts used in turn, each on its
    own HIP stream, all driven by one host thread in `sdn_render_frames_pipelined_f16`: the next frame starts as soon as a
    context is free and the newest frame in flight is down to N / overlap_div alive rays (1 = at once), so the latency-bound
    parts of one frame (marching chains, tail iterations, launch gaps) run under the throughput-bound field kernels of another.
    Every frame is the one `DeviceLoop.render` produces, bit for bit; what changes is frames per second."""

    def __init__(self, model, field, N, device, overlap_div=1, contexts=2, mailbox=True, **kw):
        import ctypes
        from sdn_backend import SdnRenderCtx, HostMailbox
        self.N, self.device, self.overlap_div, self.K = N, device, int(overlap_div), int(contexts)
        self.loops = [DeviceLoop(model, field, N, device, mailbox=mailbox, **kw) for _ in range(self.K)]
        self.streams = _context_streams(self.K, device)
        if self.K > 1:
            # several frames in flight: the persistent field launches leave an eighth of the CUs to the other frames' small kernels
            import sdn_backend as B
            B.lib.sdn_field_persistent_workgroups(int(torch.cuda.get_device_properties(device).multi_processor_count) * 7 // 8)
        # mailbox=False: ordinary pinned memory -> the driver's event + side-stream copy read-back (see DeviceLoop)
        self.host_state = HostMailbox(self.K) if mailbox else torch.zeros(self.K, 8, dtype=torch.int32).pin_memory()
        self._ctxs = (ctypes.POINTER(SdnRenderCtx) * self.K)(*[ctypes.pointer(lp.ctx) for lp in self.loops])
        self._fixed = None

    def _drive(self, lib, n, a_ro, a_rd, a_img, a_dep, bg_color, fx, a_ev, n_ev, exclusive, a_ft, a_done, iters):
        import ctypes
        from sdn_backend import check
        check(lib.sdn_render_frames_pipelined_f16(self._ctxs, self.K, n, a_ro, a_rd, a_img, a_dep, float(bg_color), self.overlap_div, fx["streams"],
                                                  fx["sides"], fx["ev_main"], fx["ev_copy"], self.host_state.data_ptr(), a_ev, n_ev,
                                                  (ctypes.c_uint8 * n)(*[1 if e else 0 for e in exclusive]) if exclusive is not None else None,
                                                  a_ft, a_done, iters),
              "render_frames_pipelined_f16")

    @torch.no_grad()
    def render_frames(self, rays_o, rays_d, time, bg_color=1.0, outputs=None, timing=None, exclusive=None, on_done=None):
        """rays_o / rays_d: lists of [N,3] tensors, one per frame (entries may repeat).  time: ONE time stamp for the whole stream
        (number or the reference's [1,1] tensor), or a list with one entry per frame -- a D-NeRF test set carries its own time for
        every frame (dnerf/utils.py:151-161); with frame-group contexts (`frames=F`) an entry is itself a list of F times.
        outputs: optional list of (image [N,3],
        depth [N]) tensors per frame; by default frame f lands in the output buffers of context f % contexts.  timing: optional list
        (per frame, None allowed) of ctypes arrays of 2 * DeviceLoop.MAX_TIMED hipEvent_t for that frame's field launches.
        exclusive: optional list of bools per frame; a flagged frame is rendered with nothing else in flight.
        on_done: optional callable (frame index, image, depth) run on a helper thread, in frame order, as soon as a frame's last
        kernel has been enqueued -- with torch's current stream (of that thread) already ordered behind the frame -- while later
        frames still render: the hook for the per-frame all-gather of a ray-sharded job.
        Returns the list of (image, depth) per frame and the per-frame iteration counts."""
        import ctypes
        import time as _t
        _tin = _t.perf_counter()
        from sdn_backend import lib, check, ptr
        n = len(rays_o)
        assert n == len(rays_d) and n >= 1
        cur = torch.cuda.current_stream()
        ro = [r.contiguous().view(-1, 3) for r in rays_o]
        rd = [r.contiguous().view(-1, 3) for r in rays_d]
        from sdn_backend import SdnFrameTime
        times = list(time) if isinstance(time, (list, tuple)) else [time] * n     # a list is ALWAYS one entry per frame
        if len(times) != n:
            raise ValueError(f"{n} frames, {len(times)} time entries")
        a_ft = (SdnFrameTime * n)()
        ft_refs = []
        for f in range(n):
            rec, refs = self.loops[0].frame_time(times[f])
            a_ft[f] = rec
            ft_refs.append(refs)
        for s, lp in enumerate(self.loops):
            self.streams[s].wait_stream(cur)
            with torch.cuda.stream(self.streams[s]):
                lp.bind(ro[min(s, n - 1)], rd[min(s, n - 1)], times[min(s, n - 1)])     # aabb, handles; rays and time constants are set per frame
                lp.side.wait_stream(self.streams[s])
        if outputs is None:
            outputs = [(self.loops[f % self.K].image_out, self.loops[f % self.K].depth_out) for f in range(n)]
        vp = ctypes.c_void_p
        if self._fixed is None:
            K = self.K
            self._fixed = dict(streams=(vp * K)(*[s.cuda_stream for s in self.streams]),
                               sides=(vp * K)(*[lp.side.cuda_stream for lp in self.loops]),
                               ev_main=(vp * (4 * K))(*[e.cuda_event for lp in self.loops for e in lp.events]),
                               ev_copy=(vp * (4 * K))(*[e.cuda_event for lp in self.loops for e in lp.copy_events]))
        fx = self._fixed
        a_ro = (vp * n)(*[ptr(t, torch.float32, "rays_o") for t in ro])
        a_rd = (vp * n)(*[ptr(t, torch.float32, "rays_d") for t in rd])
        a_img = (vp * n)(*[ptr(o[0], torch.float32, "image_out") for o in outputs])
        a_dep = (vp * n)(*[ptr(o[1], torch.float32, "depth_out") for o in outputs])
        iters = (ctypes.c_uint32 * n)()
        a_ev, n_ev = None, 0
        if timing is not None:
            a_ev = (vp * n)(*[ctypes.cast(t, vp).value if t is not None else None for t in timing])
            n_ev = DeviceLoop.MAX_TIMED
        import os, time as _t
        _dbg = os.environ.get("SDN_DRIVER_STATS")
        a_done, worker = None, None
        if on_done is not None:
            from .dist import InOrderHandOn
            # (timing-enabled: bench.py reads the device-side span of the renders and of the gathers from them)
            done = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
            for e in done:
                e.record(cur)                    # materialises the hipEvent_t handles; the driver re-records them
            a_done = (vp * n)(*[e.cuda_event for e in done])
            dev_index = self.streams[0].device_index
            self._done_stream = getattr(self, "_done_stream", None) or torch.cuda.Stream()
            side = self._done_stream

            def in_side_stream():                # the helper thread's CUDA context: this device, its own stream
                torch.cuda.set_device(dev_index)
                return torch.cuda.stream(side)
            # loop f is finished once the driver has published its iteration count; its `done` event orders the helper's stream behind it
            worker = InOrderHandOn(n, lambda f: iters[f] != 0, on_done, outputs, before=lambda f: side.wait_event(done[f]),
                                   context=in_side_stream).start()
            self.last_done_events, self.last_hand_on = done, worker
        _t0 = _t.perf_counter()
        try:
            self._drive(lib, n, a_ro, a_rd, a_img, a_dep, bg_color, fx, a_ev, n_ev, exclusive, a_ft, a_done, iters)
        except BaseException:
            if worker is not None:
                worker.cancel()
                worker.thread.join()
            raise
        if worker is not None:
            worker.join()
            cur.wait_stream(self._done_stream)
        self._ft_refs = ft_refs    # the per-frame constants stay alive until the next stream of frames
        _t1 = _t.perf_counter()
        for s in self.streams:
            cur.wait_stream(s)
        if _dbg:
            import sys
            print(f"[sdn py] entry->call {1e3 * (_t0 - _tin):.2f} ms, driver call {1e3 * (_t1 - _t0):.2f} ms, wait_stream {1e3 * (_t.perf_counter() - _t1):.2f} ms",
                  file=sys.stderr)
        return outputs, [int(v) for v in iters]



class RayBatchRenderer:
    """A SMALL ray batch (a training batch's proxy render, SealDNeRF/utils.py:632-656; a preview; an error-map pass) rendered in one
    pass instead of through the iteration loop.

    The inference branch (dnerf/renderer.py:333-381) marches `n_step` samples per alive ray, evaluates, composites, compacts, ~12-16
    times: the point of the loop is to stop marching rays that have terminated, which pays for a frame of 640 000 rays.  For 4 096
    rays it is a chain of ~30 dependent launches of almost no work (0.96 ms on the MI355X).  Here every ray's samples are listed up
    front by the training marcher (same positions and step lengths as the inference marcher, no perturbation), the fused field
    kernel evaluates them in ONE launch (the count is read on the device), and `sdn_composite_whole_rays` composites each ray with
    the inference arithmetic: image and weights_sum are the loop's bit for bit, depth to fp32 rounding.  A seal mapper, if given,
    sits between marcher and field exactly as in the loop.  Cost: the samples behind a ray's termination point are evaluated too.

    `samples_per_ray` sizes the sample buffer (N * samples_per_ray slots); a batch that needs more loses its last rays -- `render(...,
    check=True)` reads the count back and raises, `overflowed()` does the same on demand."""

    def __init__(self, model, field, N, device, max_steps=1024, T_thresh=1e-2, dt_gamma=0.0, mapper=None, samples_per_ray=96):
        import sdn_backend as B
        B.require_device()
        if mapper is not None and ("rgb" in mapper.map_data or "map_source" in mapper.map_data):
            # (the loop's iterations are the unit both options look at -- modify_rgb's mean brightness, map_to_origin's early return:
            #  one pass over all samples would tint and redirect differently from the reference's loop)
            raise NotImplementedError("mapSource / rgb tint depend on the loop's iterations: use render_frame / DeviceLoop for such edits")
        self.model, self.field, self.N, self.device, self.mapper = model, field, int(N), torch.device(device), mapper
        self.max_steps, self.T_thresh, self.dt_gamma = int(max_steps), float(T_thresh), float(dt_gamma)
        M = self.N * int(samples_per_ray)
        self.M = M + (128 - M % 128)
        f32, dev = torch.float32, self.device
        self.flat = torch.empty(self.M * 8, dtype=f32, device=dev)
        self.xyzs, self.dirs, self.deltas = self.flat[:3 * self.M].view(-1, 3), self.flat[3 * self.M:6 * self.M].view(-1, 3), self.flat[6 * self.M:].view(-1, 2)
        self.rays = torch.empty(self.N, 3, dtype=torch.int32, device=dev)
        self.counter = torch.zeros(2, dtype=torch.int32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.noises = torch.zeros(self.N, dtype=f32, device=dev)
        self.nears, self.fars = torch.empty(self.N, dtype=f32, device=dev), torch.empty(self.N, dtype=f32, device=dev)
        self.identity = torch.arange(self.M, dtype=torch.int32, device=dev)
        self.scratch = torch.empty(int(B.lib.sdn_march_rays_train_scratch_bytes(self.N, self.max_steps)), dtype=torch.uint8, device=dev)
        self.weights_sum, self.depth = torch.empty(self.N, dtype=f32, device=dev), torch.empty(self.N, dtype=f32, device=dev)
        self.image = torch.empty(self.N, 3, dtype=f32, device=dev)
        self.image_out, self.depth_out = torch.empty(self.N, 3, dtype=f32, device=dev), torch.empty(self.N, dtype=f32, device=dev)
        self._aabb = model.aabb_infer.detach().to(dev, f32).contiguous()

    @torch.no_grad()
    def render(self, rays_o, rays_d, time, bg_color=1.0, check=False):
        import sdn_backend as B
        m, lib, st = self.model, B.lib, B.stream()
        ro, rd = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        if ro.shape[0] != self.N:
            raise ValueError(f"RayBatchRenderer was built for {self.N} rays, got {ro.shape[0]}")
        self.field.set_time(time)                       # time bias, canonical-frame flag, occupancy slice -- by VALUE, cached
        bitfield = m.density_bitfield[self.field.t_idx]
        B.check(lib.sdn_near_far_from_aabb(B.ptr(ro, torch.float32, "rays_o"), B.ptr(rd, torch.float32, "rays_d"), B.ptr(self._aabb), self.N,
                                           float(m.min_near), B.ptr(self.nears), B.ptr(self.fars), st), "near_far_from_aabb")
        self.flat.zero_()
        self.counter.zero_()
        B.check(lib.sdn_march_rays_train(B.ptr(ro), B.ptr(rd), B.ptr(bitfield, torch.uint8, "density_bitfield"), float(m.bound), self.dt_gamma,
                                         self.max_steps, self.N, int(m.cascade), int(m.grid_size), self.M, B.ptr(self.nears), B.ptr(self.fars),
                                         B.ptr(self.xyzs), B.ptr(self.dirs), B.ptr(self.deltas), B.ptr(self.rays), B.ptr(self.counter),
                                         B.ptr(self.noises), B.ptr(self.scratch), st), "march_rays_train")
        torch.clamp(self.counter[:1], max=self.M, out=self.count)
        mask = self.mapper.map_to_origin_(self.xyzs, self.dirs) if self.mapper is not None else None
        sigmas, rgbs = self.field(self.xyzs, self.dirs, live_idx=self.identity, live_count=self.count)
        if mask is not None:
            self.mapper.map_color_(rgbs, mask)
        B.check(lib.sdn_composite_whole_rays(B.ptr(sigmas), B.ptr(rgbs), B.ptr(self.deltas), B.ptr(self.rays), B.ptr(self.nears), self.M, self.N,
                                             self.T_thresh, B.ptr(self.weights_sum), B.ptr(self.depth), B.ptr(self.image), st), "composite_whole_rays")
        bg = bg_color if isinstance(bg_color, torch.Tensor) else float(bg_color)
        torch.addcmul(self.image, (1 - self.weights_sum).unsqueeze(-1), bg if isinstance(bg, torch.Tensor) else torch.full((1, 3), bg, device=self.device),
                      out=self.image_out)
        torch.div(torch.clamp(self.depth - self.nears, min=0), self.fars - self.nears, out=self.depth_out)
        if check and self.overflowed():
            raise RuntimeError(f"RayBatchRenderer: the batch needs {int(self.counter[0])} samples, the buffer holds {self.M} (raise samples_per_ray)")
        return {"image": self.image_out, "depth": self.depth_out, "weights_sum": self.weights_sum, "n_samples": self.counter[:1]}

    def overflowed(self):
        """One host read-back: did the last batch need more sample slots than the buffer has?"""
        return int(self.counter[0]) > self.M
