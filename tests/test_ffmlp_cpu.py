"""CPU checks of the ffmlp oracle (oracle/ffmlp.py) and of the host-side FFMLP module.

The reference holds no golden vectors for ffmlp (testing/test_ffmlp.py is a speed comparison against a bias-free
`nn.Linear` stack with the same flat initialisation), so the oracle is anchored on exactly that comparison: the
same network in torch fp32 autograd, tolerance = fp16 rounding of activations and gradients.
"""
import numpy as np
import pytest
import torch

from oracle import ffmlp as F


def _torch_mlp(x, mats, act):
    fn = {0: torch.relu, 1: torch.exp, 3: torch.sigmoid,
          4: lambda z: 0.5 * (10 * z + torch.sqrt(100 * z * z + 4)) / 10,
          5: lambda z: torch.log(torch.exp(10 * z) + 1) / 10, 6: lambda z: z}[act]
    h = x
    for w in mats[:-1]:
        h = fn(h @ w.T)
    return h @ mats[-1].T


@pytest.mark.parametrize("act", [0, 3, 4, 5, 6])
@pytest.mark.parametrize("dims", [(16, 64, 2), (32, 16, 3), (48, 128, 2)])
def test_oracle_ffmlp_matches_torch_fp32_autograd(act, dims):
    in_dim, hidden, L = dims
    rng = np.random.default_rng(5)
    B = 200
    std = np.sqrt(3 / hidden)
    w = rng.uniform(-std, std, hidden * (in_dim + hidden * (L - 1) + 16)).astype(np.float16)
    x = rng.standard_normal((B, in_dim)).astype(np.float16)
    g = (rng.standard_normal((B, 16)) / 16).astype(np.float16)

    out, fwd = F.ffmlp_forward(x, w, in_dim, hidden, L, act)
    gi, gw, bwd = F.ffmlp_backward(g, x, w, fwd, in_dim, hidden, L, act)

    xt = torch.tensor(x.astype(np.float32), requires_grad=True)
    mats = [torch.tensor(m.astype(np.float32), requires_grad=True) for m in F.split_weights(w, in_dim, hidden, L)]
    yt = _torch_mlp(xt, mats, act)
    yt.backward(torch.tensor(g.astype(np.float32)))
    # fp16 activations: 2^-11 relative per rounding, a few layers deep
    yt = yt.detach()
    np.testing.assert_allclose(out.astype(np.float32), yt.numpy(), rtol=2e-2, atol=2e-2 * float(yt.abs().max()))
    # gradients: a pre-activation within fp16 rounding of a ReLU kink flips its gate, so single elements may differ by a
    # whole term; the bar is the norm-wise error (2% of the gradient norm covers the fp16 rounding of every G_l and X_l)
    ref_gw = np.concatenate([m.grad.numpy().reshape(-1) for m in mats])
    assert np.linalg.norm(gw.astype(np.float32) - ref_gw) <= 2e-2 * np.linalg.norm(ref_gw)
    assert np.linalg.norm(gi.astype(np.float32) - xt.grad.numpy()) <= 2e-2 * np.linalg.norm(xt.grad.numpy())
    assert fwd.shape == (L, B, hidden) and bwd.shape == (L, B, hidden)


def test_oracle_ffmlp_identity_network_is_exact():
    """Known answer: identity weights + ReLU on non-negative inputs reproduce the input in the first 16 outputs."""
    hidden = in_dim = 16
    L = 2
    eye = np.eye(16, dtype=np.float16)
    w = np.concatenate([eye.reshape(-1)] * (L + 1))
    x = np.abs(np.random.default_rng(0).standard_normal((64, 16))).astype(np.float16)
    out, fwd = F.ffmlp_forward(x, w, in_dim, hidden, L, 0)
    assert np.array_equal(out, x) and np.array_equal(fwd[0], x) and np.array_equal(fwd[1], x)


def test_ffmlp_module_matches_reference_shape_rules():
    """ffmlp.py:100-131: parameter count, fixed-seed uniform init, assertion messages."""
    import ffmlp
    m = ffmlp.FFMLP(32, 3, 64, 3)
    assert m.padded_output_dim == 16 and m.num_parameters == 64 * (32 + 64 * 2 + 16)
    assert m.weights.shape == (m.num_parameters,) and m.weights.dtype == torch.float32
    torch.manual_seed(42)
    ref = torch.zeros(m.num_parameters).uniform_(-np.sqrt(3 / 64), np.sqrt(3 / 64))
    assert torch.equal(m.weights.data, ref)
    assert ffmlp.convert_activation("relu") == 0 and ffmlp.convert_activation("softplus") == 5 and ffmlp.convert_activation("none") == 6
    for bad in [dict(input_dim=20, output_dim=3, hidden_dim=64, num_layers=2), dict(input_dim=16, output_dim=17, hidden_dim=64, num_layers=2),
                dict(input_dim=16, output_dim=3, hidden_dim=48, num_layers=2), dict(input_dim=16, output_dim=3, hidden_dim=64, num_layers=1)]:
        with pytest.raises(AssertionError):
            ffmlp.FFMLP(**bad)
    # no device here: the operator must refuse to run rather than fall back
    if not torch.cuda.is_available():
        import sdn_backend
        with pytest.raises(sdn_backend.SdnError):
            m(torch.zeros(4, 32))
