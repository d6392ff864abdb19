"""Where the reference-shaped frame (run_cuda over NeRFNetwork.forward) spends its time: host enqueue time per stage, then the same with
a synchronize after every stage (stage = host + device time, serialised)."""
import os, sys, time, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "seald-nerf_amd"))
import torch
from dnerf_amd.bench_scene import build_scene, camera_path
import raymarching
from dnerf_amd import renderer as R

dev = torch.device("cuda:0")
sc = build_scene(H=800, W=800, device=dev, seed=0, kind="jumpingjacks")
cam_o, cam_d, cam_t = camera_path(sc, 20, dev)
model = sc.model.eval()

def frame(i):
    t = torch.tensor([[cam_t[i]]], dtype=torch.float32, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        return model.render(cam_o[i][None], cam_d[i][None], t, staged=False, perturb=False, bg_color=1, max_steps=1024)

for i in range(4): frame(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(4, 8): frame(i)
torch.cuda.synchronize()
print("plain ms/frame %.3f" % ((time.perf_counter() - t0) * 250))

acc = collections.defaultdict(float); cnt = collections.Counter()
SYNC = False
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        if SYNC: torch.cuda.synchronize()
        acc[key] += time.perf_counter() - t; cnt[key] += 1
        return r
    setattr(obj, name, g)
wrap(R.raymarching, "march_rays", "march_rays")
wrap(R.raymarching, "composite_rays", "composite_rays")
wrap(R.raymarching, "near_far_from_aabb", "near_far")
orig_fwd = type(model).forward
def fwd(self, *a, **k):
    t = time.perf_counter()
    r = orig_fwd(self, *a, **k)
    if SYNC: torch.cuda.synchronize()
    acc["forward"] += time.perf_counter() - t; cnt["forward"] += 1
    return r
type(model).forward = fwd
for SYNC in (False, True):
    acc.clear(); cnt.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8, 12): frame(i)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) * 250
    print("sync-after-stage" if SYNC else "host-enqueue-only", "ms/frame %.3f" % tot,
          {k: "%.3f ms (%d calls)" % (v * 250, cnt[k] // 4) for k, v in acc.items()}, "rest %.3f" % (tot - sum(acc.values()) * 250))
# the compaction line alone
ra = torch.arange(640000, dtype=torch.int32, device=dev); ra[::3] = -1
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): rb = ra[ra >= 0]
torch.cuda.synchronize(); print("rays_alive[rays_alive >= 0] at 640 K: %.1f us" % ((time.perf_counter() - t0) / 20 * 1e6))
ra = ra[:20000].contiguous()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): rb = ra[ra >= 0]
torch.cuda.synchronize(); print("rays_alive[rays_alive >= 0] at 20 K: %.1f us" % ((time.perf_counter() - t0) / 20 * 1e6))
