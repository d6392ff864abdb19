"""`load_reference_checkpoint` on files shaped like the reference trainer's (nerf/utils.py:1033-1093), CPU only: the loader never
executes anything from the file (weights_only=True), yet must read the numpy scalars the reference's meters put into `stats`."""
import numpy as np
import pytest
import torch

import tests_support  # noqa: F401


class _Net(torch.nn.Module):
    cuda_ray = True

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(4, 3, bias=False)
        self.register_buffer("density_bitfield", torch.zeros(8, dtype=torch.uint8))
        self.mean_count, self.mean_density = 0, 0.0


def test_checkpoint_with_numpy_scalars_in_stats_loads(tmp_path):
    """`stats['results']` / `['best_result']` are numpy.float64 after the first evaluation epoch (PSNRMeter.measure,
    nerf/utils.py:1017-1018, 1073-1075); numpy arrays may sit there too.  The plain weights-only loader refuses such a file."""
    from dnerf_amd import utils
    a, b = _Net(), _Net()
    with torch.no_grad():
        a.lin.weight.copy_(torch.arange(12.0).view(3, 4))
        a.density_bitfield.fill_(7)
    state = {"epoch": 3, "global_step": 99, "mean_count": 17, "mean_density": np.float32(0.5),
             "stats": {"loss": [0.5], "valid_loss": [np.float64(0.25)], "results": [np.float64(31.5), np.float64(32.25)],
                       "checkpoints": [], "best_result": np.float64(32.25), "curve": np.arange(3, dtype=np.float32)},
             "model": a.state_dict()}
    path = str(tmp_path / "ngp_ep0003.pth")
    torch.save(state, path)
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)             # what round 2's loader did
    missing, unexpected = utils.load_reference_checkpoint(b, path, model_only=False)
    info = utils.load_reference_checkpoint.last
    assert not missing and not unexpected and torch.equal(a.lin.weight, b.lin.weight) and int(b.density_bitfield[0]) == 7
    assert b.mean_count == 17 and float(b.mean_density) == 0.5
    assert info["epoch"] == 3 and float(info["stats"]["best_result"]) == 32.25 and [float(v) for v in info["stats"]["results"]] == [31.5, 32.25]
    assert info["stats"]["curve"].tolist() == [0.0, 1.0, 2.0]


class _Evil:
    def __reduce__(self):
        return (print, ("executed",))


def test_checkpoint_with_an_executable_global_is_refused_and_named(tmp_path):
    from dnerf_amd import utils
    path = str(tmp_path / "bad.pth")
    torch.save({"model": _Net().state_dict(), "stats": {"x": _Evil()}}, path)
    with pytest.raises(RuntimeError, match="weights-only"):
        utils.load_reference_checkpoint(_Net(), path)
