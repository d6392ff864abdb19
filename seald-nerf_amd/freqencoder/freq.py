"""NeRF frequency (positional) encoder on libsdn_hip (MI355X).

Same API as /root/reference/freqencoder/freq.py: `freq_encode` (Function, :15-52) and
`FreqEncoder` (nn.Module, :55-77).  Output layout [x | sin 2^0 x | cos 2^0 x | sin 2^1 x | ...].
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from sdn_backend import lib as _lib, check as _check, ptr as _ptr, stream as _stream, to_device as _dev


class _freq_encoder(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, output_dim):
        """freq.py:18-35.  inputs [B, D] -> [B, output_dim], output_dim = D + 2*D*degree."""
        inputs = _dev(inputs).contiguous()
        B, input_dim = inputs.shape
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        _check(_lib.sdn_freq_encode_forward(_ptr(inputs, torch.float32, "inputs"), B, input_dim, int(degree), int(output_dim),
                                            _ptr(outputs), _stream()), "freq_encode_forward")
        ctx.save_for_backward(inputs, outputs)
        ctx.dims = [B, input_dim, degree, output_dim]
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        """freq.py:40-49."""
        grad = grad.contiguous()
        inputs, outputs = ctx.saved_tensors
        B, input_dim, degree, output_dim = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _check(_lib.sdn_freq_encode_backward(_ptr(grad, torch.float32, "grad"), _ptr(outputs), B, input_dim, int(degree), int(output_dim),
                                             _ptr(grad_inputs), _stream()), "freq_encode_backward")
        return grad_inputs, None, None


freq_encode = _freq_encoder.apply


class FreqEncoder(nn.Module):
    """freq.py:55-77."""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = freq_encode(inputs, self.degree, self.output_dim)
        return outputs.reshape(prefix_shape + [self.output_dim])
