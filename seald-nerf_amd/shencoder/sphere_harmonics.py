"""Real spherical-harmonics direction encoder on libsdn_hip (MI355X).

Same API as /root/reference/shencoder/sphere_harmonics.py: `sh_encode` (Function, :14-58)
and `SHEncoder` (nn.Module, :61-87).  fp32 only, like the reference (custom_fwd casts).
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from sdn_backend import lib as _lib, check as _check, ptr as _ptr, stream as _stream, require_device


class _sh_encoder(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        """sphere_harmonics.py:17-41.  inputs [B,3] -> [B, degree^2]."""
        require_device()
        inputs = inputs.contiguous()
        B, input_dim = inputs.shape
        output_dim = degree ** 2
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(B, input_dim * output_dim, dtype=inputs.dtype, device=inputs.device) if calc_grad_inputs else None
        _check(_lib.sdn_sh_encode_forward(_ptr(inputs, torch.float32, "inputs"), _ptr(outputs), B, input_dim, int(degree), _ptr(dy_dx),
                                          _stream()), "sh_encode_forward")
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = [B, input_dim, degree]
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        """sphere_harmonics.py:46-55."""
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        grad = grad.contiguous()
        B, input_dim, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _check(_lib.sdn_sh_encode_backward(_ptr(grad, torch.float32, "grad"), _ptr(inputs), B, input_dim, int(degree), _ptr(dy_dx),
                                           _ptr(grad_inputs), _stream()), "sh_encode_backward")
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    """sphere_harmonics.py:61-87."""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert self.degree > 0 and self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])
