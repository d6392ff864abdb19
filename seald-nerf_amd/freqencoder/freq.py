"""NeRF frequency (positional) encoder on libsdn_hip (MI355X).

Drop-in for the reference's `freqencoder/freq.py`: `freq_encode(inputs, degree, output_dim)` (autograd Function, :15-52) and
`FreqEncoder(input_dim=3, degree=4)` (nn.Module, :55-77).  Output layout [x | sin 2^0 x | cos 2^0 x | sin 2^1 x | ...]; fp32 only
(the reference casts with `custom_fwd(cast_inputs=float32)`, so does this).
"""
import torch
from torch import nn
from torch.amp import custom_bwd, custom_fwd

import sdn_backend as _sdn


def _launch_forward(x, degree, width):
    rows, dim = x.shape
    y = x.new_empty((rows, width))
    _sdn.check(_sdn.lib.sdn_freq_encode_forward(_sdn.ptr(x, torch.float32, "inputs"), rows, dim, int(degree), int(width), _sdn.ptr(y),
                                                _sdn.stream()), "freq_encode_forward")
    return y


def _launch_backward(dy, y, shape, degree):
    rows, dim = shape
    dx = dy.new_zeros((rows, dim))
    _sdn.check(_sdn.lib.sdn_freq_encode_backward(_sdn.ptr(dy, torch.float32, "grad"), _sdn.ptr(y), rows, dim, int(degree), int(y.shape[1]),
                                                 _sdn.ptr(dx), _sdn.stream()), "freq_encode_backward")
    return dx


class _freq_encoder(torch.autograd.Function):
    """d/dx of [x, sin(2^k x), cos(2^k x)] needs only the forward outputs (cos / -sin are their neighbours), which is all that
    is kept for the backward pass."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, output_dim):
        x = _sdn.to_device(inputs).contiguous()
        y = _launch_forward(x, degree, output_dim)
        ctx.save_for_backward(y)
        ctx.sdn_meta = (tuple(x.shape), int(degree))
        return y

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        (y,) = ctx.saved_tensors
        shape, degree = ctx.sdn_meta
        return _launch_backward(grad.contiguous(), y, shape, degree), None, None


freq_encode = _freq_encoder.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = input_dim * (1 + 2 * degree)

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        lead = inputs.shape[:-1]
        flat = freq_encode(inputs.reshape(-1, self.input_dim), self.degree, self.output_dim)
        return flat.reshape(*lead, self.output_dim)
