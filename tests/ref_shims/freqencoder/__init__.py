"""CPU stand-in for the reference's `freqencoder` package (test infrastructure; see ../README.md): freqencoder/freq.py:15-77."""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from oracle import oracle as O


class _FreqEncode(Function):
    @staticmethod
    def forward(ctx, inputs, degree, output_dim):
        out = O.freq_encode_forward(np.ascontiguousarray(inputs.detach().numpy(), np.float32), int(degree), int(output_dim))
        ctx.saved = (out, inputs.shape[1], int(degree))
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, grad):
        out, D, degree = ctx.saved
        return torch.from_numpy(O.freq_encode_backward(np.ascontiguousarray(grad.numpy(), np.float32), out, D, degree)), None, None


freq_encode = _FreqEncode.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def forward(self, inputs, **kwargs):
        lead = list(inputs.shape[:-1])
        out = freq_encode(inputs.reshape(-1, self.input_dim), self.degree, self.output_dim)
        return out.reshape(lead + [self.output_dim])
