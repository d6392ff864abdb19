"""A dnerf training step as ONE HIP graph (MI355X: the step is launch-bound -- ~200 kernels of a few microseconds each for 4096
rays / ~9000 samples, 1.6 ms of device time inside 2.9 ms of wall time when launched from Python).

What the step is (nerf/utils.py:849-930 `train_one_epoch` -> `train_step` with `-O`): render the ray batch through the
occupancy-grid path (`march_rays_train`, field network under autocast, `composite_rays_train`), a colour loss, GradScaler-scaled
backward (composite backward, MLP backward, `grid_encode` backward atomics), Adam.  To be capturable the step may not read anything
back to the host, so:
  * the point budget must be known (`model.mean_count > 0`: the reference's own steady state, raymarching.py:200-203; run the first
    steps eagerly and call `update_extra_state`, as the reference does);
  * the network must not branch on `t == 0` on the host (`NeRFNetworkFF.forward` selects on the device);
  * the optimizer is the fused, capturable Adam, which takes GradScaler's `found_inf` on the device (no `.item()` per step).
Inputs live in fixed buffers the caller's batch is copied into; `time` is read by the kernels from device memory, so the frame time
may change from step to step.  `step_counter` bookkeeping (`local_step`, the 16-slot ring read by `update_extra_state`) is replayed
outside the graph with one small device copy.
"""
import torch


def merged_param_groups(groups):
    """Parameter groups with identical hyper-parameters folded into one (the reference's `get_params` returns seven groups with two
    distinct learning rates, four of them non-empty): the same Adam, but the fused optimizer launches three kernels per GROUP."""
    merged = {}
    for g in groups:
        params = list(g["params"])
        if not params:
            continue
        key = tuple(sorted((k, v) for k, v in g.items() if k != "params"))
        merged.setdefault(key, []).extend(params)
    return [dict(key, params=params) for key, params in merged.items()]


class GraphedTrainStep:
    def __init__(self, model, optimizer, scaler, n_rays, device, loss_fn=None, warmup=3, grad_sync=None, **render_kw):
        """grad_sync: a `dnerf_amd.dist.GradSync` for data-parallel training -- the step is then two graphs (forward + backward |
        optimizer) with the gradient all-reduces between them."""
        if not getattr(model, "cuda_ray", False) or model.mean_count <= 0:
            raise ValueError("GraphedTrainStep needs the occupancy-grid path with a known point budget (model.mean_count > 0)")
        self.model, self.opt, self.scaler = model, optimizer, scaler
        self.loss_fn = loss_fn or (lambda out, target: ((out["image"] - target) ** 2).mean())
        self.render_kw = dict(staged=False, perturb=True, bg_color=1, force_all_rays=False, max_steps=1024)
        self.render_kw.update(render_kw)
        f32 = torch.float32
        self.rays_o = torch.zeros(1, n_rays, 3, dtype=f32, device=device)
        self.rays_d = torch.zeros(1, n_rays, 3, dtype=f32, device=device)
        self.rays_d[..., 2] = 1
        self.target = torch.zeros(1, n_rays, 3, dtype=f32, device=device)
        self.time = torch.full((1, 1), 0.5, dtype=f32, device=device)
        self.graph, self.graph_opt, self.loss, self.warmup, self.grad_sync = None, None, None, warmup, grad_sync
        self._loaded, self._budget, self.captures = False, None, 0

    def _step(self):
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.scaler.is_enabled()):
            out = self.model.render(self.rays_o, self.rays_d, self.time, **self.render_kw)
            loss = self.loss_fn(out, self.target)
        self.scaler.scale(loss).backward()
        if self.grad_sync is not None:
            self.grad_sync.reduce_all()
        self._optimize()
        return loss

    def _backward_only(self):
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.scaler.is_enabled()):
            out = self.model.render(self.rays_o, self.rays_d, self.time, **self.render_kw)
            loss = self.loss_fn(out, self.target)
        self.scaler.scale(loss).backward()
        return loss

    def _optimize(self):
        self.scaler.step(self.opt)
        self.scaler.update()

    def load(self, rays_o, rays_d, target, time):
        self.rays_o.copy_(rays_o.reshape(self.rays_o.shape))
        self.rays_d.copy_(rays_d.reshape(self.rays_d.shape))
        self.target.copy_(target.reshape(self.target.shape))
        self.time.copy_(time.reshape(1, 1))
        self._loaded = True

    # ---- the warm-up steps a capture needs must not train: everything they touch is snapshotted and put back IN PLACE (the graph
    # ---- bakes in the addresses of parameters, Adam moments and the scaler's scale) ------------------------------------------------
    def _snapshot(self):
        params = [p for g in self.opt.param_groups for p in g["params"]]
        snap = {"params": [(p, p.detach().clone()) for p in params], "opt": {}, "scaler": None,
                "counters": (self.model.step_counter.clone(), self.model.local_step)}
        for p in params:
            st = self.opt.state.get(p)
            if st:
                snap["opt"][p] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
        if self.scaler.is_enabled() and self.scaler._scale is not None:
            snap["scaler"] = (self.scaler._scale.clone(), self.scaler._growth_tracker.clone())
        return snap

    def _restore(self, snap):
        with torch.no_grad():
            for p, v in snap["params"]:
                p.copy_(v)
                st = self.opt.state.get(p)
                if not st:
                    continue
                old = snap["opt"].get(p)
                for k, v2 in st.items():
                    if torch.is_tensor(v2):
                        if old is not None and k in old:
                            v2.copy_(old[k])
                        else:
                            v2.zero_()             # state the warm-up created (a fresh optimizer): back to "no step taken"
            if self.scaler.is_enabled() and self.scaler._scale is not None:
                if snap["scaler"] is not None:
                    self.scaler._scale.copy_(snap["scaler"][0])
                    self.scaler._growth_tracker.copy_(snap["scaler"][1])
                else:
                    self.scaler._scale.fill_(self.scaler._init_scale)
                    self.scaler._growth_tracker.zero_()
            self.model.step_counter.copy_(snap["counters"][0])
            self.model.local_step = snap["counters"][1]

    def capture(self):
        """Warm-up steps on a side stream, then the capture.  The warm-up runs on the LOADED batch (capturing on the placeholder
        buffers would take optimizer steps towards a black image) and has no training effect: parameters, Adam moments, the
        GradScaler state and the step-counter ring are restored afterwards."""
        if not self._loaded:
            raise RuntimeError("GraphedTrainStep.capture(): load() a batch first -- the warm-up steps must run on real rays")
        m = self.model
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.opt.zero_grad(set_to_none=True)
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        self._budget = (m.mean_count, tuple(sorted((k, v) for k, v in self.render_kw.items() if isinstance(v, (bool, int, float)))))
        self.captures += 1
        self._slot = m.local_step % 16                 # the ring slot run_cuda will bake into the graph
        self.graph = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        if self.grad_sync is None:
            with torch.cuda.graph(self.graph):
                self.loss = self._step().detach()
        else:
            with torch.cuda.graph(self.graph):
                self.loss = self._backward_only().detach()
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool()):
                self._optimize()
        m.local_step -= 1                              # recording the step did not run it
        self._restore(snap)
        self._slot = (snap["counters"][1] + self.warmup) % 16   # the slot the captured step writes (local_step is restored below it)
        return self

    def __call__(self, rays_o=None, rays_d=None, target=None, time=None):
        """One training step; returns the (static) loss tensor.  Arguments, if given, are copied into the graph's input buffers."""
        if rays_o is not None:
            self.load(rays_o, rays_d, target, time)
        m = self.model
        # The graph bakes in the point budget M = mean_count (rounded up) and the render switches: `update_extra_state` recomputes
        # mean_count every epoch (dnerf/renderer.py:550-552), the reference re-sizes M each step -- a stale, smaller M would drop the
        # tail rays of every batch for good.  Re-capture when any of them changed.
        if self.graph is not None and self._budget != (m.mean_count, tuple(sorted((k, v) for k, v in self.render_kw.items() if isinstance(v, (bool, int, float))))):
            self.graph, self.graph_opt = None, None
        if self.graph is None:
            self.capture()
        self.graph.replay()
        if self.graph_opt is not None:
            self.grad_sync.reduce_all()                # RCCL all-reduces of the gradients the backward graph left in place
            self.graph_opt.replay()
        slot = m.local_step % 16
        if slot != self._slot:
            m.step_counter[slot].copy_(m.step_counter[self._slot])
        m.local_step += 1
        return self.loss
