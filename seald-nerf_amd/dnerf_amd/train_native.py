"""A dnerf training step as ONE native call (`sdn_train_step_f16`, csrc/train.hip): ~35 kernel launches, no autograd graph, no
host read-back.

What the step is (dnerf/utils.py:38-125 `train_step` with `-O` inside nerf/utils.py:849-930 `train_one_epoch`): render the ray batch
through the occupancy-grid path (`march_rays_train`, field network under fp16 autocast, `composite_rays_train`), MSE against the
ground truth, GradScaler-scaled backward, Adam, optionally the EMA shadow update.  `GraphedTrainStep` (train_graph.py) replays the
op-by-op step -- ~160 launches of autograd glue -- as one HIP graph; this class calls the native composition of the same
operators' kernels instead and is the faster of the two.  Both train the SAME objects: the model's own parameters, the
torch.optim.Adam instance's own moment tensors and the GradScaler's own scale, so eager steps, graphed steps, native steps,
`update_extra_state`, checkpoint save / load can be mixed freely (call `refresh()` after anything else changed the parameters).

Requirements: the occupancy-grid path with a known sample budget (`model.mean_count > 0`, the reference's steady state after the
first `update_extra_state`), the default network geometry of dnerf/network.py, no background model.
"""
import ctypes

import numpy as np
import torch

import sdn_backend as _sdn

_PARAM_NAMES = (["encoder.embeddings"] + [f"deform_net.{i}.weight" for i in range(8)] + [f"sigma_net.{i}.weight" for i in range(2)]
                + [f"color_net.{i}.weight" for i in range(3)])


def deterministic_mode():
    """SDN_DETERMINISTIC=1 in the environment, or torch.use_deterministic_algorithms(True): order-independent accumulations where the
    default path uses float / half atomics (the grid encoder's table gradient)."""
    import os
    return os.environ.get("SDN_DETERMINISTIC", "0") == "1" or torch.are_deterministic_algorithms_enabled()


def _budget(mean_count, align=128):
    """raymarching.py:200-203, including the `+ align` of an already aligned count."""
    return mean_count + (align - mean_count % align)


class NativeTrainStep:
    def __init__(self, model, optimizer, scaler, n_rays, device, ema_decay=None, perturb=True, bg_color=1, dt_gamma=0.0, max_steps=1024,
                 T_thresh=1e-4, seed=0, grad_sync=None, train_deform=True, overlap_table_update=False, deterministic=None):
        """optimizer: a torch.optim.Adam over `model.get_params(lr, lr_net)` (or merged groups); scaler: torch.amp.GradScaler.
        ema_decay: None, or the decay of a torch_ema-style shadow kept in `self.ema_shadow` (nerf/utils.py:906).
        train_deform=False: the deformation MLP is evaluated but not trained (SealD-NeRF's edit training, SealDNeRF/utils.py:692-694;
        the optimizer then need not hold its parameters).
        overlap_table_update: the optimizer's pass over the embedding table (366 MB of traffic, most of the pass) runs on a second
        stream beside the NEXT step's deformation-MLP forward, which does not read the table (~35 us of a 0.27 ms step).  The next
        step / `refresh()` / `flush()` order the caller's stream behind it; code that reads or writes `encoder.embeddings`, its Adam
        moments or its EMA shadow on its own right after a step must call `flush()` first (a device-wide synchronize also does).
        grad_sync: a `dnerf_amd.dist.GradSync` for data-parallel training -- every rank runs forward + backward on its own batch, the
        fp16 gradient buffers of the step (24 MB table gradient + one 250 KB block with every MLP's) are all-reduced over RCCL, and
        the optimizer pass divides by the world size on top of the loss scale."""
        if not getattr(model, "cuda_ray", False) or model.mean_count <= 0:
            raise ValueError("NativeTrainStep needs the occupancy-grid path with a known point budget (model.mean_count > 0)")
        if getattr(model, "bg_radius", 0) > 0:
            raise NotImplementedError("NativeTrainStep: the background model is not part of the native step")
        if not isinstance(optimizer, torch.optim.Adam) or any(g.get("weight_decay", 0) or g.get("amsgrad") or g.get("maximize") for g in optimizer.param_groups):
            raise ValueError("NativeTrainStep implements torch.optim.Adam without weight decay / amsgrad / maximize (main_dnerf.py:118)")
        _sdn.require_device()
        self.model, self.opt, self.scaler, self.device = model, optimizer, scaler, torch.device(device)
        self.n_rays, self.perturb, self.bg_color = int(n_rays), bool(perturb), bg_color
        self.dt_gamma, self.max_steps, self.T_thresh, self.seed = float(dt_gamma), int(max_steps), float(T_thresh), int(seed)
        named = dict(model.named_parameters())
        self.params = [named[n] for n in _PARAM_NAMES]
        enc = model.encoder
        if (enc.num_levels, enc.level_dim, enc.input_dim, enc.gridtype_id, bool(enc.align_corners), enc.interp_id) != (16, 2, 3, 1, False, 0):
            raise ValueError("NativeTrainStep: the native step is built for the dnerf network's tiled 16 x 2 grid (dnerf/network.py:55-60)")
        shapes = [tuple(p.shape) for p in self.params[1:]]
        if shapes != [(128, 76)] + [(128, 128)] * 6 + [(3, 128), (64, 32), (16, 64), (64, 31), (64, 64), (3, 64)]:
            raise ValueError(f"NativeTrainStep: network geometry {shapes} is not the default of dnerf/network.py:10-96")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device.type != "cuda":
                raise ValueError("NativeTrainStep: parameters must be contiguous fp32 tensors on the GPU")
        f32 = torch.float32
        self._rays = [(torch.zeros(n_rays, 3, dtype=f32, device=device), torch.zeros(n_rays, 3, dtype=f32, device=device)) for _ in range(2)]
        for _, d in self._rays:
            d[:, 2] = 1
        self.rays_o, self.rays_d = self._rays[0]           # the input buffers of the next step (sample set 0 unless prefetching)
        self.target = torch.zeros(n_rays, 3, dtype=f32, device=device)
        self._target_in = self.target
        self.bg = torch.zeros(n_rays, 3, dtype=f32, device=device)
        self.image = torch.zeros(n_rays, 3, dtype=f32, device=device)
        self.loss = torch.zeros(1, dtype=f32, device=device)
        self.adam_steps = torch.zeros(2, dtype=f32, device=device)
        self.time = 0.5
        self._offsets = (ctypes.c_int32 * 17)(*[int(v) for v in enc.offsets.cpu().tolist()])
        # Adam state: the optimizer's own tensors (created here if it has not stepped yet) -- for the parameters the optimizer
        # trains.  A parameter outside its param_groups (the frozen deformation MLP of edit training) gets NO state entry: an entry
        # for a foreign parameter makes `optimizer.state_dict()` raise KeyError, and the native step never reads a frozen segment's
        # moments (csrc/train.hip: step_ok, k_train_adam).
        in_groups = {id(q) for g in optimizer.param_groups for q in g["params"]}
        self._trained = [id(p) in in_groups for p in self.params]
        missing = [n for n, p, t in zip(_PARAM_NAMES, self.params, self._trained) if not t and not (not train_deform and n.startswith("deform_net."))]
        if missing:
            raise ValueError(f"NativeTrainStep: the optimizer does not hold {missing} (only a frozen deformation MLP may be left out)")
        steps = {}
        for i, p in enumerate(self.params):
            if not self._trained[i]:
                continue
            st = optimizer.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.zeros((), dtype=f32, device=p.device) if optimizer.defaults.get("capturable") else torch.tensor(0.0, dtype=f32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps[i] = float(st["step"])
        self.adam_steps[0], self.adam_steps[1] = steps[0], steps.get(1, 0.0)
        if scaler.is_enabled() and scaler._scale is None:
            scaler._lazy_init_scale_growth_tracker(self.device)
        if not scaler.is_enabled():
            raise ValueError("NativeTrainStep is the fp16 (`-O`) training step: it needs an enabled GradScaler")
        self.ema_decay, self.ema_updates = ema_decay, 0
        self.ema_shadow = [p.detach().clone() for p in self.params] if ema_decay is not None else None
        self.step_count, self._M, self._ws, self._rec = 0, None, None, None
        self._cull_cache, self._cull_epoch = {}, None
        self._set, self._pending, self._side, self._before_step = 0, None, None, None
        self.grad_sync, self.train_deform = grad_sync, bool(train_deform)
        self._table_side = None
        if overlap_table_update:
            self._table_side = (torch.cuda.Stream(device=self.device), torch.cuda.Event(), torch.cuda.Event())
            for e in self._table_side[1:]:
                e.record()                                # materialises the hipEvent_t handles; the native step re-records them
            # `model.state_dict()` (a checkpoint, an EMA copy for evaluation) reads the table: order the reader's stream behind the pass
            self._sd_hook = model.register_state_dict_pre_hook(lambda module, prefix, keep_vars: self.flush())
        # deterministic mode: the one order-dependent sum of a step -- the table gradient's half atomics -- is taken
        # in fixed point with integer atomics instead; two runs from the same state then give the same bits.  Default: SDN_DETERMINISTIC=1
        # in the environment, or torch.use_deterministic_algorithms(True).
        self.deterministic = deterministic_mode() if deterministic is None else bool(deterministic)
        self.noises = None          # optional [n_rays] f32 device tensor: the per-ray offsets of the next steps (instead of the generator)
        self._time_cache = (None, None, None)       # (tensor, _version, value) of the last `time` tensor read back
        self._lr_of = {}
        for g in optimizer.param_groups:
            for p in g["params"]:
                self._lr_of[id(p)] = g

    # ---- workspace / argument record ----------------------------------------------------------------------------------------------
    def _build(self):
        m = self.model
        M = _budget(int(m.mean_count))
        lay = _sdn.SdnTrainLayout()
        _sdn.check(_sdn.lib.sdn_train_layout(self.n_rays, M, self.max_steps, self._offsets, ctypes.byref(lay)), "train_layout")
        self.layout = lay
        self._ws = torch.empty(int(lay.total_bytes) + 256, dtype=torch.uint8, device=self.device)
        base = self._ws.data_ptr()
        self._ws_ptr = (base + 255) // 256 * 256
        r = _sdn.SdnTrainStep()
        r.target = self.target.data_ptr()
        r.N, r.M = self.n_rays, M
        self._pending = None                              # samples marched for another budget are of no use
        self._aabb = m.aabb_train.detach().to(self.device, torch.float32).contiguous()
        r.aabb = self._aabb.data_ptr()
        r.bound, r.min_near, r.dt_gamma = float(m.bound), float(m.min_near), self.dt_gamma
        r.density_scale, r.T_thresh = float(m.density_scale), self.T_thresh
        r.cascade, r.grid_size, r.max_steps = int(m.cascade), int(m.grid_size), self.max_steps
        r.perturb = int(self.perturb)
        for i in range(17):
            r.grid_offsets[i] = self._offsets[i]
        r.grid_S, r.grid_H = float(np.log2(m.encoder.per_level_scale)), int(m.encoder.base_resolution)
        for i, p in enumerate(self.params):
            q = r.params[i]
            q.param, q.n = p.data_ptr(), p.numel()
            if self._trained[i]:
                st = self.opt.state[p]
                q.exp_avg, q.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            else:
                q.exp_avg, q.exp_avg_sq = None, None      # frozen segment: never read (deform_frozen)
            q.ema = self.ema_shadow[i].data_ptr() if self.ema_shadow is not None else None
        r.adam_steps = self.adam_steps.data_ptr()
        r.loss_scale, r.growth_tracker = self.scaler._scale.data_ptr(), self.scaler._growth_tracker.data_ptr()
        r.growth_factor, r.backoff_factor = float(self.scaler.get_growth_factor()), float(self.scaler.get_backoff_factor())
        r.growth_interval = int(self.scaler.get_growth_interval())
        r.loss_out, r.image_out, r.workspace = self.loss.data_ptr(), self.image.data_ptr(), self._ws_ptr
        if self._table_side is not None:
            r.table_stream, r.table_ready, r.table_done = (self._table_side[0].cuda_stream, self._table_side[1].cuda_event, self._table_side[2].cuda_event)
        if self.deterministic:
            # order-independent table gradient (sdn_grid_encode_backward_det): one 64-bit fixed-point accumulator per table element
            self._det = torch.empty(int(self._offsets[16]) * 2, dtype=torch.int64, device=self.device)
            r.det_scratch = self._det.data_ptr()
        if self._rec is not None:
            self.flush()                                  # the old record's table pass may still be in flight: order this stream behind it
        self._rec, self._M = r, M
        self.refresh()

    def _cull_grid(self, t_idx):
        """The marcher's coarse skip grid of one occupancy slice, kept until the density grid is updated again (`iter_density` counts
        the updates, dnerf/renderer.py:555)."""
        m = self.model
        if int(m.grid_size) != 128 or int(m.cascade) != 1:
            return None
        # (the epoch also names the bitfield's storage and in-place version: `load_state_dict`, `fill_bitfield` and
        # `reset_extra_state` rewrite the occupancy without a new `iter_density` value)
        epoch = (m.iter_density, m.density_bitfield.data_ptr(), m.density_bitfield._version)
        if self._cull_epoch != epoch:
            self._cull_cache, self._cull_epoch = {}, epoch
        hit = self._cull_cache.get(t_idx)
        if hit is None:
            hit = torch.empty(int(_sdn.lib.sdn_cull_grid_bytes()), dtype=torch.uint8, device=self.device)
            _sdn.check(_sdn.lib.sdn_build_cull_grid(m.density_bitfield[t_idx].data_ptr(), 128, hit.data_ptr(), _sdn.stream()), "build_cull_grid")
            self._cull_cache[t_idx] = hit
        return hit.data_ptr()

    def flush(self):
        """Orders torch's current stream behind the table pass a previous step left on the second stream (overlap_table_update); a
        no-op otherwise.  Call it before touching `encoder.embeddings`, its optimizer state or its EMA shadow outside the step."""
        if self._table_side is not None and self._rec is not None:
            _sdn.check(_sdn.lib.sdn_train_flush(ctypes.byref(self._rec), _sdn.stream()), "train_flush")

    def invalidate_cull_grids(self):
        """Forget the cached skip grids (the occupancy bitfield was rewritten by something this object cannot see)."""
        self._cull_cache, self._cull_epoch = {}, None

    def refresh(self, optimizer_state=False):
        """Re-derive the fp16 copies the kernels read from the fp32 parameters (after a checkpoint load, an eager step, ...); the
        cached skip grids are dropped too (a checkpoint load rewrites the occupancy bitfield in place).
        optimizer_state=True: also re-read the optimizer's state -- `optimizer.load_state_dict()` REPLACES the moment tensors and
        carries the step counts -- and the scaler's scale tensors (`scaler.load_state_dict()` replaces them too)."""
        self.invalidate_cull_grids()
        self.flush()
        if optimizer_state:
            for i, p in enumerate(self.params):
                if not self._trained[i]:
                    continue
                st = self.opt.state[p]
                if "exp_avg" not in st:
                    raise RuntimeError("NativeTrainStep.refresh(optimizer_state=True): the optimizer holds no state for a trained parameter")
                if st["exp_avg"].device != p.device:
                    st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].to(p.device), st["exp_avg_sq"].to(p.device)
                if i in (0, 1):
                    self.adam_steps[i] = float(st["step"])
            if self.scaler._scale is None:
                self.scaler._lazy_init_scale_growth_tracker(self.device)
            self._rec = None
        if self._rec is None:
            return self._build()
        _sdn.check(_sdn.lib.sdn_train_refresh(ctypes.byref(self._rec), _sdn.stream()), "train_refresh")

    def view(self, name, dtype, shape):
        """A tensor view of a workspace buffer named in `SdnTrainLayout` (gradients `g_*`, fp16 parameter copies `w_*`, samples...)."""
        off = int(getattr(self.layout, name)) + (self._ws_ptr - self._ws.data_ptr())
        n = int(np.prod(shape)) * torch.empty(0, dtype=dtype).element_size()
        return self._ws[off:off + n].view(dtype).view(*shape)

    def gradient_buffers(self):
        """The step's fp16 gradients as two flat tensors: the table's, and the block g_deform .. g_color (alignment gaps included)."""
        rows = self.params[0].shape[0]
        lay = self.layout
        end = int(lay.g_color) + (64 * 32 + 64 * 64 + 16 * 64) * 2
        base = self._ws_ptr - self._ws.data_ptr()
        small = self._ws[base + int(lay.g_deform):base + end].view(torch.float16)
        return [self.view("g_table", torch.float16, (rows * 2,)), small]

    # ---- one step -------------------------------------------------------------------------------------------------------------------
    def _time_value(self, time):
        """The time stamp as a host float.  A host number costs nothing; a DEVICE tensor (what the reference's loader hands over,
        dnerf/provider.py) costs a blocking read-back, so its value is cached per tensor OBJECT and in-place version (the cache keeps
        the tensor alive: an address alone can be recycled by another tensor with another value): a loader that reuses its time
        tensor, or passes a float, keeps the step free of host synchronisation.  The version counter only sees writes made through
        torch: a time tensor changed behind torch's back (a raw pointer, DLPack, a custom kernel) must be handed over as a NEW tensor or
        as a float, or the step trains at the stale time stamp."""
        if not isinstance(time, torch.Tensor):
            return float(np.float32(float(time)))
        if time.device.type != "cuda":
            return float(np.float32(float(time.reshape(-1)[0])))
        if self._time_cache[0] is not time or self._time_cache[1] != time._version:
            self._time_cache = (time, time._version, float(np.float32(float(time.reshape(-1)[0]))))
        return self._time_cache[2]

    def _adopt(self, t, own):
        """The caller's tensor itself when the kernels can read it in place (fp32, contiguous, on this device): the step then starts
        without a device-to-device copy per input (~5 us each in a chain of ~30 launches); anything else is copied into `own`.
        Stream order keeps this safe: later writes to the tensor on the same stream queue up behind the step."""
        if isinstance(t, torch.Tensor) and t.is_cuda and t.device == own.device and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == own.numel():
            return t.detach().view(own.shape)
        own.copy_(t.reshape(own.shape))
        return own

    def load(self, rays_o, rays_d, target, time, bg_color=None):
        if self._pending is None:                         # (a prefetched batch already has its rays in place and its samples marched)
            self.rays_o, self.rays_d = self._adopt(rays_o, self._rays[self._set][0]), self._adopt(rays_d, self._rays[self._set][1])
            self.time = self._time_value(time)
        elif self._time_value(time) != self._pending["time"]:
            raise ValueError("NativeTrainStep: this call's time differs from the prefetched batch's")
        self._target_in = self._adopt(target, self.target)
        if bg_color is not None:
            self.bg_color = bg_color

    def _fill_scene(self, r, time_value, local_step):
        """Everything of the argument record that marching needs (phase 0 / 1)."""
        m = self.model
        T = m.time_size
        t_idx = int(min(max(np.floor(np.float32(time_value) * np.float32(T)), 0), T - 1))   # dnerf/renderer.py:285
        r.bitfield = m.density_bitfield[t_idx].data_ptr()
        r.cull_grid = self._cull_grid(t_idx)
        r.counter = m.step_counter[local_step % 16].data_ptr()
        r.noise_seed = (self.seed * 0x9E3779B1 + self.step_count) & 0xFFFFFFFFFFFFFFFF
        r.noises = self.noises.data_ptr() if self.noises is not None else None

    def prefetch(self, rays_o, rays_d, time):
        """March the NEXT batch now, on a side stream, beside whatever step is still running: call it right after `step(...)` with the
        batch of the following call (which then skips its own march; its `rays_o` / `rays_d` / `time` arguments are taken as given
        here; the rays must exist by the time that preceding `step(...)` was called).  Marching reads the occupancy grid only, so nothing of the running step is touched; the samples go to the workspace's
        second sample buffer.  Results are those of the un-prefetched sequence bit for bit (same noise stream, same counter slots).
        fp32 contiguous ray tensors are read IN PLACE by the side stream (no copy): do not overwrite them before the following
        `step(...)` call has been issued."""
        m = self.model
        if self._rec is None or self._M != _budget(int(m.mean_count)):
            self._build()
        if self._pending is not None:
            raise RuntimeError("NativeTrainStep.prefetch(): one batch is already marched ahead")
        q = 1 - self._set
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side
        # Ordering: the march may start once everything enqueued BEFORE the step still running has finished -- that covers the last
        # step that read sample set q.  It does not wait for the running step itself (that is the point), so the rays handed in here
        # must not be produced by work enqueued on the current stream after that step's call.
        if self._before_step is not None:
            side.wait_event(self._before_step)
        else:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ro, rd = self._adopt(rays_o, self._rays[q][0]), self._adopt(rays_d, self._rays[q][1])
            r = self._rec
            tv = self._time_value(time)
            r.time, r.rays_o, r.rays_d = tv, ro.data_ptr(), rd.data_ptr()
            self._fill_scene(r, tv, m.local_step)
            r.mode, r.phase, r.sample_set = 0, 1, q
            _sdn.check(_sdn.lib.sdn_train_step_f16(ctypes.byref(r), _sdn.stream()), "train_step_f16 (march ahead)")
            done = torch.cuda.Event()
            done.record(side)
        self._pending = {"set": q, "event": done, "time": tv, "local_step": m.local_step, "step_count": self.step_count, "rays": (ro, rd)}

    def __call__(self, rays_o=None, rays_d=None, target=None, time=None, bg_color=None, grads_only=False):
        """One training step on the loaded batch.  Arguments, if given: `time` is taken by value; fp32 contiguous tensors on this device
        are READ IN PLACE by the step's kernels (`_adopt`: no device-to-device copy) -- stream order makes later writes on the SAME
        stream safe, but a loader that refills its ray / target buffers on ANOTHER stream must order that stream behind the step (an
        event recorded after this call) -- anything else (other dtype / device / layout) is copied into the step's own buffers.
        `step.rays_o / rays_d` therefore may alias the caller's tensors, and `step.target` is the step's own buffer only in the copied case.
        A device `time` tensor is read back once per (tensor object, version): change it through torch operations, or pass a number.
        Returns the loss tensor (device, overwritten by the next step).  grads_only: forward + backward only -- gradients stay in the
        workspace (`view("g_deform", ...)`), nothing is updated."""
        if rays_o is not None:
            self.load(rays_o, rays_d, target, time, bg_color)
        m = self.model
        if self._rec is None or self._M != _budget(int(m.mean_count)):
            self._build()       # `update_extra_state` moved the budget (dnerf/renderer.py:550-552): new buffers, same parameters
        r = self._rec
        self._before_step = torch.cuda.Event()
        self._before_step.record()
        pend = self._pending
        if pend is not None and (pend["local_step"] != m.local_step or pend["step_count"] != self.step_count or grads_only or self.grad_sync is not None):
            pend = self._pending = None                   # not the batch this call is about: march again
        if pend is not None:
            torch.cuda.current_stream().wait_event(pend["event"])
            self._set, self.time = pend["set"], pend["time"]
            self.rays_o, self.rays_d = pend["rays"]
            self._pending = None
            r.phase = 2
        else:
            r.phase = 0
        r.sample_set = self._set
        r.time, r.rays_o, r.rays_d = self.time, self.rays_o.data_ptr(), self.rays_d.data_ptr()
        r.target = self._target_in.data_ptr()
        self._fill_scene(r, self.time, m.local_step)
        if isinstance(self.bg_color, torch.Tensor):
            self.bg.copy_(self.bg_color.reshape(-1, 3).expand(self.n_rays, 3))
            r.bg_color, r.bg_value = self.bg.data_ptr(), 0.0
        else:
            r.bg_color, r.bg_value = None, float(self.bg_color)
        g_table, g_net = self._lr_of[id(self.params[0])], self._lr_of[id(self.params[9])]       # encoder.embeddings, sigma_net.0
        r.lr_table, r.lr_net = float(g_table["lr"]), float(g_net["lr"])
        r.beta1, r.beta2, r.eps = float(g_net["betas"][0]), float(g_net["betas"][1]), float(g_net["eps"])
        if self.ema_decay is not None:
            n = self.ema_updates + (0 if grads_only else 1)
            r.ema_decay = min(self.ema_decay, (1 + n) / (10 + n))       # torch_ema: num_updates is incremented before use
        r.deform_frozen = 0 if self.train_deform else 1
        if self.grad_sync is not None and not grads_only:
            # data parallel: backward | all-reduce of the gradient buffers (sums; the optimizer divides) | optimizer
            r.mode, r.keep_deform, r.grad_divisor = 1, 1, float(self.grad_sync.world)
            _sdn.check(_sdn.lib.sdn_train_step_f16(ctypes.byref(r), _sdn.stream()), "train_step_f16")
            for g in self.gradient_buffers():
                self.grad_sync._reduce(g)
            r.mode = 2
        else:
            r.mode, r.keep_deform, r.grad_divisor = (1 if grads_only else 0), 0, 1.0
        _sdn.check(_sdn.lib.sdn_train_step_f16(ctypes.byref(r), _sdn.stream()), "train_step_f16")
        m.local_step += 1
        if not grads_only:
            self.step_count += 1
            self.ema_updates += 1
        return self.loss

    # ---- optimizer state the host owns ------------------------------------------------------------------------------------------------
    def sync_optimizer_state(self):
        """Writes the step counts the device keeps into the optimizer's per-parameter `step` entries (one host read-back): call before
        `optimizer.state_dict()` / an eager `optimizer.step()`."""
        self.flush()
        main, deform = [float(v) for v in self.adam_steps.tolist()]
        for i, p in enumerate(self.params):
            if self._trained[i]:
                self.opt.state[p]["step"].fill_(deform if 1 <= i <= 8 else main)
