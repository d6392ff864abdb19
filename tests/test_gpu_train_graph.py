"""GraphedTrainStep (dnerf_amd/train_graph.py): capturing the step must not train, the captured step must equal the eager step,
and a changed point budget must re-capture (the reference re-sizes M every step, raymarching.py:195-203)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed=0):
    from dnerf_amd.bench_scene import build_scene
    from dnerf_amd.network_ff import NeRFNetworkFF
    from dnerf_amd.train_graph import merged_param_groups
    sc = build_scene(H=32, W=32, device="cuda", seed=seed)
    model = NeRFNetworkFF(bound=1, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1).cuda().train()
    model.load_state_dict(sc.model.state_dict())
    opt = torch.optim.Adam(merged_param_groups(model.get_params(1e-3, 1e-3)), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
    scaler = torch.amp.GradScaler("cuda")
    target = torch.rand(1, 1024, 3, generator=torch.Generator().manual_seed(4)).cuda()
    # point budget from one reference-shaped eager march (no optimizer step)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model.render(sc.rays_o[None], sc.rays_d[None], sc.time, staged=False, perturb=False, bg_color=1, force_all_rays=False)
    model.mean_count = int(model.step_counter[0, 0].item()) + 256
    model.local_step = 0
    model.step_counter.zero_()
    return sc, model, opt, scaler, target


def _params(model):
    return {k: v.detach().clone() for k, v in model.named_parameters()}


def test_capture_has_no_training_effect_and_needs_a_loaded_batch():
    from dnerf_amd.train_graph import GraphedTrainStep
    sc, model, opt, scaler, target = _setup()
    before = _params(model)
    step = GraphedTrainStep(model, opt, scaler, 1024, "cuda", perturb=False)
    with pytest.raises(RuntimeError):
        step.capture()
    step.load(sc.rays_o, sc.rays_d, target, sc.time)
    step.capture()
    torch.cuda.synchronize()
    for k, v in model.named_parameters():
        assert torch.equal(v.detach(), before[k]), k                 # three warm-up Adam steps, all undone
    for p, st in opt.state.items():
        for name, val in st.items():
            if torch.is_tensor(val):
                assert float(val.abs().max()) == 0.0, name               # moments / step counter of a fresh optimizer
    assert model.local_step == 0 and int(model.step_counter.abs().sum()) == 0
    assert scaler.get_scale() == 65536.0


def test_graphed_step_equals_eager_step():
    from dnerf_amd.train_graph import GraphedTrainStep
    sc, model, opt, scaler, target = _setup()
    ref_model = copy.deepcopy(model)
    from dnerf_amd.train_graph import merged_param_groups
    ref_opt = torch.optim.Adam(merged_param_groups(ref_model.get_params(1e-3, 1e-3)), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
    ref_scaler = torch.amp.GradScaler("cuda")
    step = GraphedTrainStep(model, opt, scaler, 1024, "cuda", perturb=False)
    loss = step(sc.rays_o, sc.rays_d, target, sc.time)
    ref_opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        out = ref_model.render(sc.rays_o[None], sc.rays_d[None], sc.time, staged=False, perturb=False, bg_color=1, force_all_rays=False, max_steps=1024)
        ref_loss = ((out["image"] - target) ** 2).mean()
    ref_scaler.scale(ref_loss).backward()
    ref_scaler.step(ref_opt)
    ref_scaler.update()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss), float(ref_loss), rtol=1e-5)
    got, want = dict(model.named_parameters()), dict(ref_model.named_parameters())
    for k in got:
        a, b = got[k].detach(), want[k].detach()
        if k == "encoder.embeddings":
            # fp16 atomic accumulation order differs run to run; the first Adam step moves an entry by +-lr whatever the size of its
            # gradient, so an entry whose (tiny) gradient changes sign differs by 2 lr: bounded count, bounded size
            diff = (a - b).abs()
            assert float(diff.max()) <= 2.1e-3 and float((diff > 1e-6).float().mean()) < 2e-3
        else:
            assert float((a - b).abs().max()) <= 2.1e-3 and float(((a - b).abs() > 1e-6).float().mean()) < 2e-2, k
    assert model.local_step == 1 and int(model.step_counter[0, 0]) == int(ref_model.step_counter[0, 0]) > 0


def test_changed_point_budget_recaptures_and_drops_no_ray():
    from dnerf_amd.train_graph import GraphedTrainStep
    sc, model, opt, scaler, target = _setup()
    full = model.mean_count
    model.mean_count = full // 2                      # a stale, too small budget: tail rays would be dropped
    step = GraphedTrainStep(model, opt, scaler, 1024, "cuda", perturb=False)
    step(sc.rays_o, sc.rays_d, target, sc.time)
    assert step.captures == 1
    used = int(model.step_counter[(model.local_step - 1) % 16, 0])
    assert used > model.mean_count                    # the small budget does overflow on this batch
    model.mean_count = full                           # what update_extra_state would have computed
    step(sc.rays_o, sc.rays_d, target, sc.time)
    assert step.captures == 2                         # re-captured with the new M
    step()
    assert step.captures == 2
    used = int(model.step_counter[(model.local_step - 1) % 16, 0])
    assert 0 < used <= model.mean_count + 128
