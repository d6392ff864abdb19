"""ONE 800x800 frame rendered as S independent shards (interleaved 16x16 tiles, dnerf_amd/dist.py:shard_rays) that are in flight
TOGETHER, each in its own loop context on its own stream -- rays are independent (DESIGN 3 "Why per-ray results do not depend on the
loop schedule"), so the reassembled image is the lone frame's bit for bit; what changes is what a lone frame's latency-bound marcher
chains run beside.  Prints the median enqueue -> image latency per S and whether the images match."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("SPLIT_QUEUES", "5"))
import torch
from dnerf_amd.bench_scene import build_scene, camera_path
from dnerf_amd.renderer import DeviceLoop, PipelinedDeviceLoop
from dnerf_amd import fused
from dnerf_amd.dist import shard_rays

dev = torch.device("cuda:0")
size = 800
sc = build_scene(H=size, W=size, device=dev, seed=0, kind="jumpingjacks")
cam_o, cam_d, cam_t = camera_path(sc, 20, dev)
n = cam_o[0].shape[0]
field = fused.FusedField(sc.model, torch.tensor([[cam_t[0]]], device=dev), fp16=True)
K = 8


def median(v):
    return sorted(v)[len(v) // 2]


lone = DeviceLoop(sc.model, field, n, dev, keep_cull_grids=True)
ref = []
for i in range(K):
    out = lone.render(cam_o[i], cam_d[i], cam_t[i], want_stats=False)
    ref.append((out["image"].clone(), out["depth"].clone()))
torch.cuda.synchronize()
lat = []
for i in range(K):
    t0 = time.perf_counter()
    lone.render(cam_o[i], cam_d[i], cam_t[i], want_stats=False)
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - t0) * 1e3)
res = {"lone_loop_ms": round(median(lat), 4)}

for S in [int(s) for s in os.environ.get("SPLIT_PARTS", "2,3,4").split(",")]:
    shards = [shard_rays(n, size, r, S) for r in range(S)]
    per = shards[0][1]
    idx = [torch.from_numpy(s[0]).to(dev) for s in shards]
    pl = PipelinedDeviceLoop(sc.model, field, per, dev, overlap_div=1, contexts=S, keep_cull_grids=True)
    img = torch.empty(n, 3, device=dev); dep = torch.empty(n, device=dev)

    def frame(i):
        ro = [cam_o[i][ix].contiguous() for ix in idx]
        rd = [cam_d[i][ix].contiguous() for ix in idx]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs, _ = pl.render_frames(ro, rd, [cam_t[i]] * S)
        for ix, (im, dp) in zip(idx, outs):
            img[ix] = im
            dep[ix] = dp
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3
    for i in range(K):
        frame(i)
    lat, same = [], True
    for i in range(K):
        lat.append(frame(i))
        same = same and torch.equal(img, ref[i][0]) and torch.equal(torch.nan_to_num(dep), torch.nan_to_num(ref[i][1]))
    res[f"shards_{S}_ms"] = round(median(lat), 4)
    res[f"shards_{S}_identical"] = bool(same)
    del pl
print(json.dumps(res))
