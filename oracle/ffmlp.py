"""CPU oracle of the fully fused MLP operator `ffmlp` (test infrastructure only).

Restates, in numpy, the arithmetic of the reference's ffmlp extension:
  * layout and layer order: ffmlp/src/ffmlp.cu:630-633 (flat fp16 weights, row-major [hidden, in] ++
    (L-1) x [hidden, hidden] ++ [16, hidden]; inputs [B, in]; outputs [B, 16]; forward_buffer [L, B, hidden]);
  * forward: kernel_mlp_fused, ffmlp.cu:331-407 (activation on every hidden layer, none on the output);
  * activations and their derivatives from the stored post-activations: ffmlp/src/utils.h:425-470, :538-583
    (K_ACT = 10, :41); the derivative factor is rounded to fp16 and multiplied in fp16 as there;
  * backward chain and weight gradients dW_l = G_l^T X_l: ffmlp.cu:411-523, :770-894.

PARITY UNPINNED.  The reference accumulates every product in fp16 wmma fragments (`OUT_T = __half`, ffmlp.cu:68,
:168, :256) and sums its split-K weight gradients in fp16 CUTLASS epilogues; both are hardware-defined summation
orders that no CPU restatement can reproduce, the extension cannot be built here (CUDA + un-vendored CUTLASS,
ffmlp/setup.py:47-49), and the reference holds no golden vectors for it (testing/test_ffmlp.py is a speed
comparison against an `nn.Linear` stack).  This oracle therefore fixes the obvious contract: exact products,
wide accumulation, ONE rounding to fp16 per layer output -- the same contract `torch.nn.Linear` under autocast
follows, which is what test_ffmlp.py compares FFMLP against.
"""
import numpy as np

K_ACT = np.float32(10.0)

ACTIVATIONS = {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5, "none": 6}


def split_weights(weights, input_dim, hidden_dim, num_layers, padded_output_dim=16):
    """Flat weights -> list of [out, in] matrices in layer order (ffmlp.cu:631)."""
    w = np.asarray(weights)
    mats, off = [], 0
    shapes = [(hidden_dim, input_dim)] + [(hidden_dim, hidden_dim)] * (num_layers - 1) + [(padded_output_dim, hidden_dim)]
    for r, c in shapes:
        mats.append(w[off:off + r * c].reshape(r, c))
        off += r * c
    assert off == w.size
    return mats


def _h(x):
    return np.asarray(x, np.float32).astype(np.float16)


def _linear(x16, w16):
    """fp16 operands, exact products, wide accumulation, one rounding to fp16."""
    return (x16.astype(np.float64) @ w16.astype(np.float64).T).astype(np.float32).astype(np.float16)


def activation_forward(act, z16):
    """utils.h:425-470 on the fp16 layer output."""
    x = z16.astype(np.float32)
    with np.errstate(over="ignore"):
        if act == 0:
            return np.where(z16 > 0, z16, np.float16(0))
        if act == 1:
            return _h(np.exp(x))
        if act == 2:
            return _h(np.sin(x))
        if act == 3:
            return _h(np.float32(1) / (np.float32(1) + np.exp(-x)))
        if act == 4:
            s = x * K_ACT
            return _h(np.float32(0.5) * (s + np.sqrt(s * s + np.float32(4))) / K_ACT)
        if act == 5:
            return _h(np.log(np.exp(x * K_ACT) + np.float32(1)) / K_ACT)
    return z16


def activation_backward(act, g16, f16):
    """utils.h:538-583: gradient w.r.t. the pre-activation from the stored post-activation f."""
    f = f16.astype(np.float32)
    if act == 0:
        return np.where(f16 > 0, g16, np.float16(0))
    if act == 1:
        return g16 * f16
    if act == 2:
        raise NotImplementedError("sine: the reference has no backward for it (utils.h:552-556)")
    if act == 3:
        return g16 * (f16 * (np.float16(1) - f16))
    if act == 4:
        y = f * K_ACT
        return g16 * _h(y * y / (y * y + np.float32(1)))
    if act == 5:
        return g16 * _h(np.float32(1) - np.exp(-f * K_ACT))
    return g16


def ffmlp_forward(inputs, weights, input_dim, hidden_dim, num_layers, activation):
    """-> (outputs [B,16] f16, forward_buffer [L,B,hidden] f16)."""
    mats = [_h(m) for m in split_weights(weights, input_dim, hidden_dim, num_layers)]
    x = _h(inputs)
    fwd = []
    for w in mats[:-1]:
        x = activation_forward(activation, _linear(x, w))
        fwd.append(x)
    return _linear(x, mats[-1]), np.stack(fwd)


def ffmlp_backward(grad, inputs, weights, forward_buffer, input_dim, hidden_dim, num_layers, activation, calc_grad_inputs=True):
    """-> (grad_inputs [B,in] f16 | None, grad_weights flat f16, backward_buffer [L,B,hidden] f16 (output side first))."""
    mats = [_h(m) for m in split_weights(weights, input_dim, hidden_dim, num_layers)]
    g = _h(grad)
    x0 = _h(inputs)
    L = num_layers
    bwd, dws = [], [None] * (L + 1)
    dws[L] = _linear(g.T, forward_buffer[L - 1].T)                        # [16, hidden]
    for k in range(L):                                                    # k-th backward layer uses matrix L - k
        g = activation_backward(activation, _linear(g, mats[L - k].T), forward_buffer[L - 1 - k])
        bwd.append(g)
        x_prev = forward_buffer[L - 2 - k] if L - 2 - k >= 0 else x0
        dws[L - 1 - k] = _linear(g.T, x_prev.T)
    grad_inputs = _linear(g, mats[0].T) if calc_grad_inputs else None
    return grad_inputs, np.concatenate([d.reshape(-1) for d in dws]), np.stack(bwd)
