"""GPU parity of the fused MLP operator (csrc/ffmlp.hip through the reference-shaped `ffmlp` package and the C ABI)
against oracle/ffmlp.py on the same seeded inputs.

Bar: both sides form exact fp16 products and round once per layer output, the GPU summing in fp32 on the matrix cores and
the oracle in fp64, so a layer output may differ by one fp16 ulp where the two sums straddle a rounding boundary (and the
next layer inherits that, so the bound grows with depth).  Tolerances below: outputs / buffers within 2 (L + 1) fp16 ulps on
99.9% of the elements and 1% norm-wise overall; weight gradients (sums over the batch of fp16 products, fp32 on the GPU) 1% norm-wise.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ffmlp as F  # noqa: E402


def _close(a, b, what, L=2):
    a = a.astype(np.float32); b = b.astype(np.float32)
    scale = float(np.abs(b).max()) + 1e-12
    ulps = 2 * (L + 1)
    bad = np.abs(a - b) > ulps * 2.0 ** -10 * np.maximum(np.abs(b), scale / 64)
    assert bad.mean() <= 1e-3, f"{what}: {bad.mean():.2e} of the elements off by more than {ulps} ulp"
    assert np.linalg.norm(a - b) <= 1e-2 * np.linalg.norm(b) + 1e-6, what


def _case(in_dim, hidden, L, B, seed=0, wscale=1.0):
    rng = np.random.default_rng(seed)
    std = np.sqrt(3 / hidden) * wscale
    w = rng.uniform(-std, std, hidden * (in_dim + hidden * (L - 1) + 16)).astype(np.float16)
    x = rng.standard_normal((B, in_dim)).astype(np.float16)
    g = (rng.standard_normal((B, 16)) / 16).astype(np.float16)
    return w, x, g


DIMS = [(16, 16, 2), (16, 64, 2), (32, 32, 3), (64, 64, 4), (48, 128, 3), (32, 128, 8), (16, 256, 2), (288, 128, 2), (128, 16, 2)]


@pytest.mark.parametrize("dims", DIMS)
def test_ffmlp_forward_inference_and_buffers_match_oracle(dims):
    import ffmlp
    in_dim, hidden, L = dims
    B = 1000                                   # not a multiple of 128 or of the 256-point workgroup
    w, x, _ = _case(in_dim, hidden, L, B)
    out_ref, fwd_ref = F.ffmlp_forward(x, w, in_dim, hidden, L, 0)
    xd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda()
    out_inf = ffmlp.ffmlp_forward(xd, wd, in_dim, 16, hidden, L, 0, 6, True, False)
    _close(out_inf.cpu().numpy(), out_ref, "inference outputs", L)
    # training forward: same outputs bit for bit, and the saved post-activations
    wd2 = wd.clone().requires_grad_(True)
    out_tr = ffmlp.ffmlp_forward(xd, wd2, in_dim, 16, hidden, L, 0, 6, False, False)
    assert torch.equal(out_tr, out_inf)
    fwd = out_tr.grad_fn.saved_tensors[3]
    assert fwd.shape == (L, B, hidden)
    _close(fwd.cpu().numpy(), fwd_ref, "forward_buffer", L)


@pytest.mark.parametrize("act", ["relu", "exponential", "sine", "sigmoid", "squareplus", "softplus", "none"])
def test_ffmlp_activations_match_oracle(act):
    import ffmlp
    in_dim, hidden, L, B = 32, 64, 3, 640
    w, x, _ = _case(in_dim, hidden, L, B, seed=3, wscale=0.5)
    a = ffmlp.convert_activation(act)
    out_ref, _ = F.ffmlp_forward(x, w, in_dim, hidden, L, a)
    out = ffmlp.ffmlp_forward(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), in_dim, 16, hidden, L, a, 6, True, False)
    _close(out.cpu().numpy(), out_ref, act, L)


@pytest.mark.parametrize("act", ["relu", "exponential", "sigmoid", "squareplus", "softplus", "none"])
@pytest.mark.parametrize("dims", [(16, 64, 2), (32, 32, 3), (48, 128, 3), (16, 256, 2), (288, 128, 2), (32, 16, 2)])
def test_ffmlp_backward_matches_oracle(dims, act):
    import ffmlp
    in_dim, hidden, L = dims
    B = 3000
    a = ffmlp.convert_activation(act)
    w, x, g = _case(in_dim, hidden, L, B, seed=7, wscale=0.5 if act in ("exponential", "softplus") else 1.0)
    out_ref, fwd_ref = F.ffmlp_forward(x, w, in_dim, hidden, L, a)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    wd = torch.from_numpy(w).cuda().requires_grad_(True)
    out = ffmlp.ffmlp_forward(xd, wd, in_dim, 16, hidden, L, a, 6, False, True)
    # the oracle's backward runs on the GPU's own forward_buffer, so that ReLU gates are the same on both sides
    fwd = out.grad_fn.saved_tensors[3].cpu().numpy()
    out.backward(torch.from_numpy(g).cuda())
    _close(fwd, fwd_ref, "forward_buffer", L)
    gi_ref, gw_ref, _ = F.ffmlp_backward(g, x, w, fwd, in_dim, hidden, L, a)
    assert xd.grad.dtype == torch.float16 and wd.grad.dtype == torch.float16
    _close(xd.grad.cpu().numpy(), gi_ref, "grad_inputs", L)
    gw = wd.grad.cpu().numpy().astype(np.float32)
    gw_ref = gw_ref.astype(np.float32)
    assert np.linalg.norm(gw - gw_ref) <= 1e-2 * np.linalg.norm(gw_ref), "grad_weights"
    # per layer too, so that a wrong small block cannot hide behind a large one
    off = 0
    for r, c in [(hidden, in_dim)] + [(hidden, hidden)] * (L - 1) + [(16, hidden)]:
        blk, ref = gw[off:off + r * c], gw_ref[off:off + r * c]
        assert np.linalg.norm(blk - ref) <= 1e-2 * np.linalg.norm(ref) + 1e-6, f"grad_weights block at {off}"
        off += r * c


def test_ffmlp_module_trains_like_a_linear_stack():
    """testing/test_ffmlp.py's comparison: FFMLP under autocast vs the same weights in bias-free nn.Linear layers."""
    import ffmlp
    in_dim, out_dim, hidden, L, B = 16, 3, 64, 2, 5000
    net = ffmlp.FFMLP(in_dim, out_dim, hidden, L).cuda()
    mats = [m.clone().float().cuda().requires_grad_(True) for m in
            map(torch.from_numpy, F.split_weights(net.weights.detach().cpu().numpy(), in_dim, hidden, L))]
    x = torch.rand(B, in_dim, device="cuda") * 2 - 1
    x0 = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        y0 = net(x0)
    assert y0.shape == (B, out_dim) and y0.dtype == torch.float16
    h = x
    for m in mats[:-1]:
        h = torch.relu(h @ m.T)
    y1 = (h @ mats[-1].T)[:, :out_dim]
    assert torch.allclose(y0.detach().float(), y1.detach(), rtol=2e-2, atol=2e-2 * float(y1.detach().abs().max()))
    (y0.float() ** 2).mean().backward()
    (y1 ** 2).mean().backward()
    ref = torch.cat([m.grad.reshape(-1) for m in mats])
    got = net.weights.grad
    assert got.dtype == torch.float32 and got.shape == ref.shape
    # loss-scaled fp16 gradients are tiny here (1/B); compare directions and norms
    assert float((got - ref).norm()) <= 5e-2 * float(ref.norm())
    assert x0.grad is not None and x0.grad.shape == (B, in_dim)
    # eval mode goes through the inference entry point and agrees with training mode
    net.eval()
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        y2 = net(x)
    assert torch.equal(y2, y0.detach())


def test_ffmlp_rejects_what_the_reference_rejects():
    import ffmlp
    import sdn_backend
    x = torch.zeros(128, 16, device="cuda", dtype=torch.float16)
    w = torch.zeros(48 * (16 + 48 + 16), device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError):
        ffmlp.ffmlp_forward(x, w, 16, 16, 48, 2, 0, 6, True, False)          # hidden 48: "hidden_dim should in [...]" (ffmlp.cu:657)
    w = torch.zeros(64 * (16 + 64 + 16), device="cuda", dtype=torch.float16)
    with pytest.raises(sdn_backend.SdnError):
        ffmlp.ffmlp_forward(x.float(), w, 16, 16, 64, 2, 0, 6, True, False)  # CHECK_IS_HALF (ffmlp.cu:637)
    wd = w.clone().requires_grad_(True)
    y = ffmlp.ffmlp_forward(x, wd, 16, 16, 64, 2, 2, 6, False, False)       # sine forward is fine ...
    with pytest.raises(NotImplementedError):
        y.sum().backward()                                                   # ... its backward is not (utils.h:552-556)


@pytest.mark.parametrize("B", [1, 31, 33, 255, 257])
def test_ffmlp_ragged_batches_match_the_padded_result(B):
    """Any batch size works (the reference pads to 128, ffmlp.py:154-157): rows beyond B are neither read nor written."""
    import ffmlp
    in_dim, hidden, L = 32, 64, 3
    w, x, g = _case(in_dim, hidden, L, 512, seed=11)
    wd = torch.from_numpy(w).cuda()
    full = ffmlp.ffmlp_forward(torch.from_numpy(x).cuda(), wd, in_dim, 16, hidden, L, 0, 6, True, False)
    guard = torch.full((B + 64, 16), -7.0, dtype=torch.float16, device="cuda")
    xs = torch.from_numpy(x[:B]).cuda().contiguous()
    out = ffmlp.ffmlp_forward(xs, wd, in_dim, 16, hidden, L, 0, 6, True, False)
    assert out.shape == (B, 16) and torch.equal(out, full[:B])
    # training forward + backward on the ragged batch, weights only (calc_grad_inputs False)
    wg = wd.clone().requires_grad_(True)
    y = ffmlp.ffmlp_forward(xs, wg, in_dim, 16, hidden, L, 0, 6, False, False)
    assert torch.equal(y, full[:B])
    fwd = y.grad_fn.saved_tensors[3].cpu().numpy()
    y.backward(torch.from_numpy(g[:B]).cuda())
    _, gw_ref, _ = F.ffmlp_backward(g[:B], x[:B], w, fwd, in_dim, hidden, L, 0, calc_grad_inputs=False)
    gw, gw_ref = wg.grad.cpu().numpy().astype(np.float32), gw_ref.astype(np.float32)
    assert np.linalg.norm(gw - gw_ref) <= 1e-2 * np.linalg.norm(gw_ref) + 1e-6
    del guard
