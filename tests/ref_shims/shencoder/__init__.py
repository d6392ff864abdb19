"""CPU stand-in for the reference's `shencoder` package (test infrastructure; see ../README.md): shencoder/sphere_harmonics.py:14-87."""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from oracle import oracle as O


class _SHEncode(Function):
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        out, dy_dx = O.sh_encode_forward(np.ascontiguousarray(inputs.detach().numpy(), np.float32), int(degree), bool(calc_grad_inputs))
        ctx.saved = (dy_dx, int(degree))
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, grad):
        dy_dx, degree = ctx.saved
        if dy_dx is None:
            return None, None, None
        return torch.from_numpy(O.sh_encode_backward(np.ascontiguousarray(grad.numpy(), np.float32), dy_dx, degree)), None, None


sh_encode = _SHEncode.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2

    def forward(self, inputs, size=1):
        inputs = inputs / size
        lead = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim)
        return sh_encode(flat, self.degree, flat.requires_grad).reshape(lead + [self.output_dim])
