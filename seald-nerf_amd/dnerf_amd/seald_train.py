"""One SealD-NeRF edit-training step on the device's terms.

The reference's step (`StudentTrainer.train_gui`, SealDNeRF/utils.py:667-777): for a batch of `num_rays` rays the
TEACHER renders the edited scene -- inference branch, seal mapper between marcher and network, `T_thresh` 1e-4, no perturbation
(`proxy_truth`, utils.py:632-656) -- and its colours replace the ground truth; the STUDENT then takes a normal training step on
them with its deformation network frozen (utils.py:692-694).  Here the teacher's render is the native loop with the fused field
kernel and the seal kernels inside the frame driver (`DeviceLoop(..., mapper=)`), and the student's step is the native training step
(`train_native.NativeTrainStep(train_deform=False)`: one call, ~30 launches; the frozen deformation MLP is evaluated, nothing flows
back through it, the grid encoder computes no input gradients) -- or, with `native=False`, the captured graph of the op-by-op step
(`train_graph.GraphedTrainStep`, student = `NeRFNetworkFF`).
"""
import torch

from . import fused
from .renderer import DeviceLoop
from .train_graph import GraphedTrainStep


def freeze_deformation(student):
    """utils.py:692-694: the edit only re-learns density and colour."""
    for p in student.deform_net.parameters():
        p.requires_grad_(False)
    return [p for p in student.parameters() if p.requires_grad]


class EditTrainStep:
    def __init__(self, teacher, student, mapper, optimizer, scaler, n_rays, device, time, native=True, one_pass=True, **render_kw):
        """one_pass: the teacher's proxy render in one pass (`RayBatchRenderer`: march, one field launch, whole-ray compositing --
        the loop's image bit for bit) instead of the iteration loop built for whole frames (a chain of ~30 launches per batch)."""
        self.teacher, self.student = teacher, student
        self.field = fused.FusedField(teacher, time, fp16=True)
        if one_pass:
            from .renderer import RayBatchRenderer
            self.loop, self._checked = RayBatchRenderer(teacher, self.field, n_rays, device, T_thresh=1e-4, mapper=mapper), 0
        else:
            self.loop = DeviceLoop(teacher, self.field, n_rays, device, T_thresh=1e-4, mapper=mapper)
        if native:
            from .train_native import NativeTrainStep
            kw = {k: v for k, v in render_kw.items() if k in ("perturb", "bg_color", "dt_gamma", "max_steps", "T_thresh", "seed", "ema_decay")}
            self.step = NativeTrainStep(student, optimizer, scaler, n_rays, device, train_deform=False, **kw)
        else:
            self.step = GraphedTrainStep(student, optimizer, scaler, n_rays, device, **render_kw)

    @torch.no_grad()
    def proxy_truth(self, rays_o, rays_d, time, bg_color=1.0):
        """Teacher colours [n_rays, 3] for these rays (the edited scene)."""
        # (the loop derives the time slice / time bias / canonical-frame flag from the VALUE of `time`, cached per value)
        if isinstance(self.loop, DeviceLoop):
            return self.loop.render(rays_o, rays_d, time, bg_color=bg_color, want_stats=False)["image"]
        self._checked += 1
        return self.loop.render(rays_o, rays_d, time, bg_color=bg_color, check=self._checked in (1, 2) or self._checked % 256 == 0)["image"]

    def __call__(self, rays_o, rays_d, time):
        target = self.proxy_truth(rays_o, rays_d, time)
        return self.step(rays_o, rays_d, target, time)

    def run(self, batches):
        """Software-pipelined epoch over an iterable of (rays_o, rays_d, time): the student's graph for batch k is launched (one call,
        asynchronous) and the host then drives the teacher's loop for batch k+1 on a second stream, so the two halves of the step --
        both chains of small launches -- overlap on the device.  The teacher is fixed, so results equal the one-after-the-other
        order.  Returns the number of steps; the last loss is `self.step.loss`."""
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream()
        side = self._side
        loaded, rendered = torch.cuda.Event(), torch.cuda.Event()
        it = iter(batches)
        cur = next(it, None)
        if cur is None:
            return 0
        if hasattr(self.step, "capture") and self.step.graph is None:            # capture before anything runs beside it
            self.step.load(cur[0], cur[1], torch.zeros_like(self.loop.image_out), cur[2])
            self.step.capture()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            target = self.proxy_truth(*cur)
            rendered.record(side)
        n = 0
        while cur is not None:
            nxt = next(it, None)
            main.wait_event(rendered)          # the teacher's colours of batch k
            self.step.load(cur[0], cur[1], target, cur[2])
            loaded.record(main)                # ... are in the graph's buffer: the teacher may overwrite its output
            self.step()
            n += 1
            if nxt is not None:
                with torch.cuda.stream(side):
                    side.wait_event(loaded)
                    target = self.proxy_truth(*nxt)
                    rendered.record(side)
            cur = nxt
        main.wait_stream(side)
        return n
