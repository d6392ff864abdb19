"""Ray-parallel sharding of one frame across the GPUs of a node.

Rays are independent, so the path shards with no data-path collective: every rank renders its own rays with a full
replica of the model, then ONE collective per frame (`all_gather_into_tensor` of [rays_per_rank, 4] fp32 = rgb + depth,
RCCL over xGMI when the backend is "nccl") assembles the image, followed by a local un-permute.

Tiles of 16x16 pixels are dealt round-robin so that the object-covering centre of the image is balanced across ranks
(contiguous row bands would give the middle ranks most of the work).
"""
import numpy as np
import torch


def shard_rays(n_rays, W, rank, world, tile=16):
    """int64 ray indices owned by `rank`, padded (by repeating its last ray) to the common shard length `per`."""
    H = n_rays // W
    tx = (W + tile - 1) // tile
    ys, xs = np.divmod(np.arange(n_rays), W)
    tile_id = (ys // tile) * tx + (xs // tile)
    owner = tile_id % world
    counts = np.bincount(owner, minlength=world)
    per = int(counts.max())
    mine = np.nonzero(owner == rank)[0]
    if mine.shape[0] == 0:  # more ranks than tiles: render ray 0 redundantly
        mine = np.zeros(1, dtype=np.int64)
    pad = per - mine.shape[0]
    if pad > 0:
        mine = np.concatenate([mine, np.repeat(mine[-1:], pad)])
    assert H * W == n_rays
    return mine.astype(np.int64), per


class InOrderHandOn:
    """Hands finished loops (frames, or frame groups) on to `on_done(f, image, depth)` in FRAME ORDER from a helper thread while later
    loops still render -- the hook of the per-loop all-gather of a ray-sharded job.  Several loops are in flight and finish out of
    order; every rank must issue its collectives in the same order, so loop f is handed on only after loops 0 .. f-1, whenever it
    finished.  `finished(f)`: has loop f's last kernel been enqueued (the frame driver publishes its iteration count)?  `before(f)`:
    optional, called right before `on_done` (orders the helper's stream behind the loop's `done` event).  `context`: optional context
    manager factory entered by the thread (its CUDA stream).  `join()` re-raises what `on_done` raised; `cancel()` stops the thread.
    `order` records the loops in the order they were handed on; `stamps` the host times (perf_counter) around each `on_done`."""

    def __init__(self, n, finished, on_done, outputs, before=None, context=None, poll=2e-5):
        import threading
        self.n, self.finished, self.on_done, self.outputs, self.before, self.context, self.poll = n, finished, on_done, outputs, before, context, poll
        self.order, self.stamps, self.failed, self._cancel = [], [], [], False
        self.thread = threading.Thread(target=self._run, daemon=True)

    def start(self):
        self.thread.start()
        return self

    def cancel(self):
        self._cancel = True

    def _run(self):
        import contextlib
        import time as _t
        try:
            with (self.context() if self.context is not None else contextlib.nullcontext()):
                for f in range(self.n):
                    while not self.finished(f) and not self._cancel:
                        _t.sleep(self.poll)
                    if self._cancel:
                        return
                    if self.before is not None:
                        self.before(f)
                    t0 = _t.perf_counter()
                    self.on_done(f, self.outputs[f][0], self.outputs[f][1])
                    self.order.append(f)
                    self.stamps.append((t0, _t.perf_counter()))
        except BaseException as exc:     # surfaced by join()
            self.failed.append(exc)

    def join(self):
        self.thread.join()
        if self.failed:
            raise self.failed[0]


class FrameGather:
    """Pre-computed index maps + buffers for assembling the full frame on every rank.

    transport: what travels.  "f32" (default): the renderer's fp32 colours and depth, 16 B per ray -- at 8 ranks 9 MB arrive per frame
    and rank, 60-90 us on a ring over 153 GB/s xGMI links, about what a rank needs to RENDER its share of a frame (0.09-0.12 ms), so
    the gathers must overlap the rendering completely (they run on their own stream).  "f16": half the bytes, colours to 5e-4.
    "u8": a quarter -- the 8-bit pixels the reference finally writes (`nerf/utils.py` test(): `(pred * 255).astype(np.uint8)` for
    image and depth alike); the frame then is uint8 [n_rays, 4]."""

    def __init__(self, n_rays, W, world, device, tile=16, transport="f32"):
        self.world = world
        shards = [shard_rays(n_rays, W, r, world, tile) for r in range(world)]
        self.per = shards[0][1]
        self.all_idx = torch.from_numpy(np.concatenate([s[0] for s in shards])).to(device)
        self.transport = transport
        self.dtype = {"f32": torch.float32, "f16": torch.float16, "u8": torch.uint8}[transport]
        self.gathered = torch.empty(world * self.per, 4, dtype=self.dtype, device=device)
        self.frame = torch.empty(n_rays, 4, dtype=self.dtype, device=device)
        self._group = {}

    def _pack(self, image_local, depth_local):
        local = torch.cat([image_local, depth_local.unsqueeze(-1)], dim=1)
        if self.transport == "u8":      # rays that miss the box have depth 0 / 0 = NaN in the reference (dnerf/renderer.py:379): 0 on the wire
            local = (torch.nan_to_num(local, nan=0.0).clamp(0, 1) * 255).to(torch.uint8)
        elif self.transport == "f16":
            local = local.to(torch.float16)
        return local.contiguous()

    def _all_gather(self, out, local):
        import torch.distributed as dist
        if local.is_cuda and dist.get_backend() == "gloo":  # one-GPU rehearsal only: gloo moves host memory
            host = torch.empty(out.shape, dtype=self.dtype)
            dist.all_gather_into_tensor(host, local.cpu())
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, local)          # RCCL over xGMI when the backend is "nccl"

    def __call__(self, image_local, depth_local):
        """image_local [per,3], depth_local [per] of this rank -> full frame [n_rays, 4] (rgb, depth) on every rank."""
        self._all_gather(self.gathered, self._pack(image_local, depth_local))
        self.frame[self.all_idx] = self.gathered  # padding rows rewrite a pixel with its own value
        return self.frame

    def gather_group(self, image_local, depth_local, frames, keep=False):
        """A frame group's shard output (image_local [frames * per, 3], depth_local [frames * per], frame-major: what a
        `DeviceLoop(frames=F)` renders on this rank) -> the `frames` full frames [frames, n_rays, 4] with ONE all_gather_into_tensor
        for the whole group: the frames of a group are finished by the same loop at the same moment, so one collective of F times
        the size pays one latency instead of F and runs nearer the links' bandwidth (8 ranks, fp32: 51 MB per group instead of 5 x 10
        MB).  keep=False: the group's frame buffer [frames, n_rays, 4] is returned and reused by the next group (a consumer --
        display, encoder, metric -- takes the frames as they are assembled); keep=True: a list of copies."""
        per, world = self.per, self.world
        assert image_local.shape[0] == frames * per and depth_local.shape[0] == frames * per
        if frames == 1:
            full = self(image_local, depth_local)
            return [full.clone()] if keep else full.unsqueeze(0)
        bufs = self._group.get(frames)
        if bufs is None:
            bufs = self._group[frames] = (torch.empty(world * frames * per, 4, dtype=self.dtype, device=self.gathered.device),
                                          torch.empty(frames, self.frame.shape[0], 4, dtype=self.dtype, device=self.gathered.device))
        gathered, out = bufs
        self._all_gather(gathered, self._pack(image_local, depth_local))
        # [rank][frame][ray of the shard] -> [frame][rank-major shard list] -> pixel order
        out[:, self.all_idx] = gathered.view(world, frames, per, 4).permute(1, 0, 2, 3).reshape(frames, world * per, 4)
        return [f.clone() for f in out] if keep else out


class GradSync:
    """Data-parallel training: every rank trains on its own ray batch with a full replica, gradients are averaged
    before the optimizer step.  Two exchanges per step over RCCL / xGMI:

      * the embedding-table gradient (6 119 864 x 2 fp32 = 46.7 MiB, one ring all-reduce: ~0.55 ms at 153 GB/s per link) -- in the
        eager step it is issued from a post-accumulate hook, i.e. the moment `grid_encode` backward has produced it, and runs under
        the deformation MLP's backward, which autograd executes next;
      * one flat bucket with every other gradient (113 k values).

    Under GradScaler the ranks' scales stay identical: an inf / nan on one rank spreads through the sum, every rank skips the step.
    With the graphed step (train_graph.py) the exchange sits between the backward graph and the optimizer graph (`reduce_all`)."""

    def __init__(self, model, group=None, table_name="encoder.embeddings"):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.table = [p for n, p in named if n == table_name]
        self.rest = [p for n, p in named if n != table_name]
        self._handle, self._hook = None, None

    def broadcast_parameters(self, src=0):
        """Replicas start equal (the reference has no DP; this is what DistributedDataParallel does at construction)."""
        for p in self.table + self.rest:
            self.dist.broadcast(p.data, src, group=self.group)

    def _reduce(self, t, async_op=False):
        if t.is_cuda and self.dist.get_backend(self.group) == "gloo":  # one-GPU rehearsal: gloo moves host memory
            host = t.cpu()
            self.dist.all_reduce(host, group=self.group)
            t.copy_(host)
            return None
        return self.dist.all_reduce(t, group=self.group, async_op=async_op)

    def install_hook(self):
        """Eager step: start the table exchange as soon as its gradient exists."""
        def fire(p):
            self._handle = self._reduce(p.grad, async_op=True)
        self._hook = [p.register_post_accumulate_grad_hook(fire) for p in self.table]
        return self

    def remove_hook(self):
        for h in self._hook or []:
            h.remove()
        self._hook = None

    def reduce_all(self):
        """Call after backward, before the optimizer (GradScaler.step): finishes / performs the table exchange, exchanges the flat
        bucket, turns sums into means."""
        from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors
        if self._hook:
            if self._handle is not None:
                self._handle.wait()
            self._handle = None
        else:
            for p in self.table:
                if p.grad is not None:
                    self._reduce(p.grad)
        grads = [p.grad for p in self.rest if p.grad is not None]
        if grads:
            flat = _flatten_dense_tensors(grads)
            self._reduce(flat)
            flat.div_(self.world)
            for g, f in zip(grads, _unflatten_dense_tensors(flat, grads)):
                g.copy_(f)
        for p in self.table:
            if p.grad is not None:
                p.grad.div_(self.world)
