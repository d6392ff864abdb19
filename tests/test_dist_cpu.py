"""world_size-2 (and 3) gloo tests of the ray-sharding / frame-assembly logic that bench.py --gpus N uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, H, W, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dnerf_amd.dist import shard_rays, FrameGather
        n = H * W
        idx, per = shard_rays(n, W, rank, world)
        # a stand-in "renderer": any per-ray function of the ray index (rays are independent)
        ridx = torch.from_numpy(idx).float()
        image = torch.stack([ridx, ridx * 2, ridx * 3], 1)
        depth = ridx * 0.5
        frame = FrameGather(n, W, world, "cpu")(image, depth)
        ref = torch.arange(n).float()
        ok = torch.equal(frame, torch.stack([ref, ref * 2, ref * 3, ref * 0.5], 1))
        # narrower transports: fp16, and the 8-bit pixels the reference writes
        unit = torch.stack([ridx / n, ridx / (2 * n), ridx / (3 * n)], 1)
        dep = ridx / n
        dep[ridx.long() % 7 == 0] = float("nan")                # rays that miss the box (a function of the RAY: shards are padded with repeats)
        want = torch.stack([ref / n, ref / (2 * n), ref / (3 * n), ref / n], 1)
        f16 = FrameGather(n, W, world, "cpu", transport="f16")(unit, ridx / n)
        ok = ok and f16.dtype == torch.float16 and torch.equal(f16, want.half())
        u8 = FrameGather(n, W, world, "cpu", transport="u8")(unit, dep)
        want8 = (want.clamp(0, 1) * 255).to(torch.uint8)
        miss = torch.arange(n) % 7 == 0
        want8[miss, 3] = 0
        ok = ok and u8.dtype == torch.uint8 and torch.equal(u8, want8)
        # timing protocol of bench.py: max over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t) == float(world)
        np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([ok, per]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 64, 64), (3, 40, 56), (2, 17, 23)])
def test_sharded_frame_assembly_gloo(tmp_path, world, H, W):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, H, W, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"ok_{r}.npy") for r in range(world)]
    assert all(bool(r[0]) for r in res)
    assert len({int(r[1]) for r in res}) == 1  # equal shard lengths (all_gather_into_tensor requirement)


def test_shards_partition_the_image_and_are_balanced():
    from dnerf_amd.dist import shard_rays
    n, W = 800 * 800, 800
    seen = np.zeros(n, dtype=np.int32)
    sizes = []
    for r in range(8):
        idx, per = shard_rays(n, W, r, 8)
        seen[np.unique(idx)] += 1
        sizes.append(np.unique(idx).shape[0])
        assert idx.shape[0] == per
    assert (seen == 1).all()                       # every ray rendered exactly once
    assert max(sizes) - min(sizes) <= 16 * 16 * 2  # tiles dealt evenly
    # centre tiles (where the figure is) are spread over all ranks
    ys, xs = np.divmod(np.arange(n), W)
    centre = (abs(ys - 400) < 64) & (abs(xs - 400) < 64)
    owners = ((ys // 16) * 50 + (xs // 16)) % 8
    assert len(set(owners[centre])) == 8


def _group_worker(rank, world, port, H, W, frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dnerf_amd.dist import shard_rays, FrameGather
        n = H * W
        idx, per = shard_rays(n, W, rank, world)
        # what a rank's frame-group loop hands over: `frames` shard outputs back to back (frame-major); the stand-in renderer makes
        # every value a function of (frame, ray) so that a mix-up of frames or of shards is visible
        ridx = torch.from_numpy(idx).float()
        image = torch.cat([torch.stack([ridx + 1000 * f, ridx * 2, ridx * 3 + f], 1) for f in range(frames)])
        depth = torch.cat([ridx * 0.5 + f for f in range(frames)])
        fg = FrameGather(n, W, world, "cpu")
        full = fg.gather_group(image, depth, frames, keep=True)        # ONE all_gather_into_tensor for the group
        ref = torch.arange(n).float()
        ok = len(full) == frames
        for f in range(frames):
            ok = ok and torch.equal(full[f], torch.stack([ref + 1000 * f, ref * 2, ref * 3 + f, ref * 0.5 + f], 1))
        group = fg.gather_group(image, depth, frames)                  # streaming form: the group's (reused) frame buffer
        ok = ok and tuple(group.shape) == (frames, n, 4) and all(torch.equal(group[f], full[f]) for f in range(frames))
        np.save(os.path.join(out_dir, f"grp_{rank}.npy"), np.array([ok]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W,frames", [(2, 64, 64, 4), (3, 40, 56, 3)])
def test_frame_group_gather_gloo(tmp_path, world, H, W, frames):
    """bench.py --gpus N renders N frames' shards per loop (frame group) and assembles the group's frames with one all-gather."""
    mp.spawn(_group_worker, args=(world, _free_port(), H, W, frames, str(tmp_path)), nprocs=world, join=True)
    assert all(bool(np.load(tmp_path / f"grp_{r}.npy")[0]) for r in range(world))


class _ToyField(torch.nn.Module):
    """Stand-in with the parameter layout GradSync distinguishes: a big table named `encoder.embeddings` + small MLP weights."""

    def __init__(self):
        super().__init__()
        self.encoder = torch.nn.Module()
        self.encoder.embeddings = torch.nn.Parameter(torch.zeros(1000, 2))
        self.net = torch.nn.ModuleList([torch.nn.Linear(2, 8, bias=False), torch.nn.Linear(8, 3, bias=False)])

    def forward(self, idx):
        h = self.encoder.embeddings[idx]
        return self.net[1](torch.relu(self.net[0](h)))


def _dp_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dnerf_amd.dist import GradSync
        torch.manual_seed(100 + rank)                      # replicas start DIFFERENT; broadcast makes them equal
        model = _ToyField()
        torch.nn.init.normal_(model.encoder.embeddings)
        sync = GradSync(model)
        sync.broadcast_parameters()
        start = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
        ok = True
        for use_hook in (False, True):
            if use_hook:
                sync.install_hook()
            g = torch.Generator().manual_seed(7 + rank)    # every rank its own batch
            idx = torch.randint(0, 1000, (64,), generator=g)
            target = torch.randn(64, 3, generator=g)
            model.zero_grad(set_to_none=True)
            ((model(idx) - target) ** 2).mean().backward()
            local = [p.grad.clone() for p in model.parameters()]
            sync.reduce_all()
            # reference: gather every rank's local gradients and average them
            for p, mine in zip(model.parameters(), local):
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                ok = ok and torch.allclose(p.grad, torch.stack(parts).mean(0), rtol=1e-6, atol=1e-7)
            if use_hook:
                sync.remove_hook()
        # replicas that step on the averaged gradient stay identical
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        opt.step()
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        parts = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(parts, flat)
        ok = ok and all(torch.equal(parts[0], q) for q in parts) and not torch.equal(flat, start)
        np.save(os.path.join(out_dir, f"dp_{rank}.npy"), np.array([ok]))
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradient_exchange_gloo(tmp_path):
    world = 2
    mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(bool(np.load(tmp_path / f"dp_{r}.npy")[0]) for r in range(world))


def _hand_on_worker(rank, world, port, H, W, n_loops, out_dir):
    """One rank of a ray-sharded job whose loops finish OUT OF ORDER (4 in flight, as the pipelined frame driver keeps them): a fake
    driver thread publishes loop completions in a per-rank shuffled order, `InOrderHandOn` runs the per-loop gather from its helper
    thread.  Collectives must be issued in loop order on every rank -- a rank that gathered loop 3 while another gathered loop 1
    would exchange the wrong frames (or deadlock)."""
    import threading
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dnerf_amd.dist import FrameGather, InOrderHandOn, shard_rays
        n = H * W
        idx, per = shard_rays(n, W, rank, world)
        fg = FrameGather(n, W, world, "cpu")
        ref = torch.arange(n, dtype=torch.float32)
        # loop f renders "image" = ray id + 1000 f on this rank's shard
        outputs = [(torch.stack([ref[idx] + 1000 * f, ref[idx] * 2, ref[idx] * 3], 1), ref[idx] * 0.5 + f) for f in range(n_loops)]
        finished = [False] * n_loops
        frames = {}

        def on_done(f, img, dep):
            frames[f] = fg(img, dep).clone()

        def driver():   # 4 loops in flight; the one that finishes next is drawn at random (different on every rank)
            rng = np.random.default_rng(100 + rank)
            in_flight, nxt = list(range(min(4, n_loops))), min(4, n_loops)
            while in_flight:
                time.sleep(float(rng.uniform(0, 2e-3)))
                f = in_flight.pop(int(rng.integers(len(in_flight))))
                finished[f] = True
                if nxt < n_loops:
                    in_flight.append(nxt)
                    nxt += 1
        hand = InOrderHandOn(n_loops, lambda f: finished[f], on_done, outputs).start()
        th = threading.Thread(target=driver)
        th.start()
        th.join()
        hand.join()
        ok = hand.order == list(range(n_loops)) and len(hand.stamps) == n_loops
        ok = ok and all(hand.stamps[f][1] <= hand.stamps[f + 1][0] for f in range(n_loops - 1))       # one gather at a time, in order
        for f in range(n_loops):
            ok = ok and torch.equal(frames[f], torch.stack([ref + 1000 * f, ref * 2, ref * 3, ref * 0.5 + f], 1))
        np.save(os.path.join(out_dir, f"ho_{rank}.npy"), np.array([ok]))
    finally:
        dist.destroy_process_group()


def test_gathers_are_handed_on_in_loop_order_with_four_loops_in_flight(tmp_path):
    """nerf/utils.py:963-977 are the reference's only collectives (metric gathers); here the per-loop tile all-gather is issued by a
    helper thread while later loops render: order and content across two ranks whose loops finish in different orders."""
    world = 2
    mp.spawn(_hand_on_worker, args=(world, _free_port(), 48, 64, 12, str(tmp_path)), nprocs=world, join=True)
    assert all(bool(np.load(tmp_path / f"ho_{r}.npy")[0]) for r in range(world))


def test_hand_on_surfaces_errors_and_can_be_cancelled():
    from dnerf_amd.dist import InOrderHandOn

    def boom(f, a, b):
        raise RuntimeError("gather failed")
    h = InOrderHandOn(2, lambda f: True, boom, [(None, None)] * 2).start()
    with pytest.raises(RuntimeError, match="gather failed"):
        h.join()
    h = InOrderHandOn(2, lambda f: False, lambda *a: None, [(None, None)] * 2).start()
    h.cancel()
    h.join()
    assert h.order == []
