"""One SealD-NeRF edit-training step on the device's terms.

The reference's step (`StudentTrainer.train_gui`, SealDNeRF/utils.py:667-777): for a batch of `num_rays` rays the
TEACHER renders the edited scene -- inference branch, seal mapper between marcher and network, `T_thresh` 1e-4, no perturbation
(`proxy_truth`, utils.py:632-656) -- and its colours replace the ground truth; the STUDENT then takes a normal training step on
them with its deformation network frozen (utils.py:692-694).  Here the teacher's render is the native loop with the fused field
kernel and the seal kernels inside the frame driver (`DeviceLoop(..., mapper=)`), and the student's step is the captured graph of
`train_graph.GraphedTrainStep` (student = `NeRFNetworkFF`; with the deformation MLP frozen its forward is the inference kernel and
the grid encoder needs no input gradients).
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
    def __init__(self, teacher, student, mapper, optimizer, scaler, n_rays, device, time, **render_kw):
        self.teacher, self.student = teacher, student
        self.field = fused.FusedField(teacher, time, fp16=True)
        self._time_key = None
        self.loop = DeviceLoop(teacher, self.field, n_rays, device, T_thresh=1e-4, mapper=mapper)
        self.step = GraphedTrainStep(student, optimizer, scaler, n_rays, device, **render_kw)

    @torch.no_grad()
    def proxy_truth(self, rays_o, rays_d, time, bg_color=1.0):
        """Teacher colours [n_rays, 3] for these rays (the edited scene)."""
        key = (time.data_ptr(), time._version)
        if key != self._time_key:            # the time bias of the fused kernel is a per-timestep constant
            self.field.set_time(time)
            self._time_key = key
        return self.loop.render(rays_o, rays_d, time, bg_color=bg_color, want_stats=False)["image"]

    def __call__(self, rays_o, rays_d, time):
        target = self.proxy_truth(rays_o, rays_d, time)
        return self.step(rays_o, rays_d, target, time)
