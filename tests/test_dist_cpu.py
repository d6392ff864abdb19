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
