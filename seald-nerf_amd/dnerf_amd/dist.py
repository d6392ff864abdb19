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


class FrameGather:
    """Pre-computed index maps + buffers for assembling the full frame on every rank."""

    def __init__(self, n_rays, W, world, device, tile=16):
        self.world = world
        shards = [shard_rays(n_rays, W, r, world, tile) for r in range(world)]
        self.per = shards[0][1]
        self.all_idx = torch.from_numpy(np.concatenate([s[0] for s in shards])).to(device)
        self.gathered = torch.empty(world * self.per, 4, dtype=torch.float32, device=device)
        self.frame = torch.empty(n_rays, 4, dtype=torch.float32, device=device)

    def __call__(self, image_local, depth_local):
        """image_local [per,3], depth_local [per] of this rank -> full frame [n_rays, 4] (rgb, depth) on every rank."""
        import torch.distributed as dist
        local = torch.cat([image_local, depth_local.unsqueeze(-1)], dim=1).contiguous()
        if local.is_cuda and dist.get_backend() == "gloo":  # one-GPU rehearsal only: gloo moves host memory
            host = torch.empty(self.gathered.shape, dtype=torch.float32)
            dist.all_gather_into_tensor(host, local.cpu())
            self.gathered.copy_(host)
        else:
            dist.all_gather_into_tensor(self.gathered, local)  # RCCL over xGMI when the backend is "nccl"
        self.frame[self.all_idx] = self.gathered  # padding rows rewrite a pixel with its own value
        return self.frame
