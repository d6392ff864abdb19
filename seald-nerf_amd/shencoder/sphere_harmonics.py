"""Real spherical-harmonics direction encoder on libsdn_hip (MI355X).

Drop-in for the reference's `shencoder/sphere_harmonics.py`: `sh_encode(inputs, degree, calc_grad_inputs=False)` (autograd
Function, :14-58) and `SHEncoder(input_dim=3, degree=4)` (nn.Module, :61-87).  fp32 only, like the reference (`custom_fwd` casts).
"""
import torch
from torch import nn
from torch.amp import custom_bwd, custom_fwd

import sdn_backend as _sdn


class _sh_encoder(torch.autograd.Function):
    """[B,3] unit directions -> [B, degree^2] basis values; the [B, 3 * degree^2] Jacobian is produced only when the caller asks for
    input gradients (never on the dnerf path: directions are data)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        _sdn.require_device()
        d = inputs.contiguous()
        rows, dim = d.shape
        width = int(degree) ** 2
        basis = d.new_empty((rows, width))
        jac = d.new_empty((rows, dim * width)) if calc_grad_inputs else None
        _sdn.check(_sdn.lib.sdn_sh_encode_forward(_sdn.ptr(d, torch.float32, "inputs"), _sdn.ptr(basis), rows, dim, int(degree), _sdn.ptr(jac),
                                                  _sdn.stream()), "sh_encode_forward")
        ctx.save_for_backward(d, jac)
        ctx.sdn_degree = int(degree)
        return basis

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        d, jac = ctx.saved_tensors
        if jac is None:
            return None, None, None
        rows, dim = d.shape
        g = grad.contiguous()
        dd = d.new_zeros(d.shape)
        _sdn.check(_sdn.lib.sdn_sh_encode_backward(_sdn.ptr(g, torch.float32, "grad"), _sdn.ptr(d), rows, dim, ctx.sdn_degree, _sdn.ptr(jac),
                                                   _sdn.ptr(dd), _sdn.stream()), "sh_encode_backward")
        return dd, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < degree <= 8, "SH encoder only supports degree in [1, 8]"
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        scaled = inputs / size
        lead = scaled.shape[:-1]
        flat = scaled.reshape(-1, self.input_dim)
        return sh_encode(flat, self.degree, flat.requires_grad).reshape(*lead, self.output_dim)
