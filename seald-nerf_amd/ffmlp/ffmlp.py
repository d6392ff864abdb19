"""Fully fused fp16 MLP on libsdn_hip (MI355X MFMA kernels, csrc/ffmlp.hip).

Same API as /root/reference/ffmlp/ffmlp.py: `ffmlp_forward` (Function, :15-84), `convert_activation` (:87-95) and
`FFMLP` (nn.Module, :98-168) with its flat `.weights` parameter ([hidden, in] ++ (L-1) x [hidden, hidden] ++
[16, hidden], row-major), fixed-seed initialisation and padding rules.  Differences, all on the permissive side:
any batch size works (the reference pads to a multiple of 128, :154-157; the padding is kept so shapes of saved
tensors match), and `activation='sine'` raises in backward instead of silently returning wrong gradients
(utils.h:552-556).
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from sdn_backend import lib as _lib, check as _check, ptr as _ptr, stream as _stream, to_device as _dev, timed as _timed


def _scratch(B, input_dim, output_dim, hidden_dim, num_layers, device):
    nbytes = int(_lib.sdn_ffmlp_scratch_bytes(B, input_dim, output_dim, hidden_dim, num_layers))
    if nbytes == 0:
        raise RuntimeError(f"ffmlp: unsupported dimensions in={input_dim} out={output_dim} hidden={hidden_dim} layers={num_layers}")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


class _ffmlp_forward(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False):
        """ffmlp.py:17-49.  inputs [B, input_dim] f16, weights flat f16 -> outputs [B, output_dim (16)] f16."""
        inputs = _dev(inputs).contiguous()
        weights = _dev(weights).contiguous()
        B = inputs.shape[0]
        outputs = torch.empty(B, output_dim, device=inputs.device, dtype=inputs.dtype)
        scratch = _scratch(B, input_dim, output_dim, hidden_dim, num_layers, inputs.device)
        if not inference:
            forward_buffer = torch.empty(num_layers, B, hidden_dim, device=inputs.device, dtype=inputs.dtype)
            with _timed("ffmlp_forward", B):
                _check(_lib.sdn_ffmlp_forward(_ptr(inputs, torch.half, "inputs"), _ptr(weights, torch.half, "weights"), B, input_dim,
                                              output_dim, hidden_dim, num_layers, activation, output_activation,
                                              _ptr(forward_buffer), _ptr(outputs), _ptr(scratch), _stream()), "ffmlp_forward")
            ctx.save_for_backward(inputs, weights, outputs, forward_buffer)
            ctx.dims = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs)
        else:
            with _timed("ffmlp_inference", B):
                _check(_lib.sdn_ffmlp_inference(_ptr(inputs, torch.half, "inputs"), _ptr(weights, torch.half, "weights"), B, input_dim,
                                                output_dim, hidden_dim, num_layers, activation, output_activation,
                                                _ptr(outputs), _ptr(scratch), _stream()), "ffmlp_inference")
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        """ffmlp.py:51-81.  grad [B, output_dim] -> (grad_inputs | None, grad_weights)."""
        grad = grad.contiguous()
        B = grad.shape[0]
        inputs, weights, outputs, forward_buffer = ctx.saved_tensors
        input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs = ctx.dims
        if activation == 2:
            raise NotImplementedError("ffmlp: the sine activation has no backward (needs pre-activations, utils.h:552-556)")
        grad_inputs = torch.empty_like(inputs) if calc_grad_inputs else None
        grad_weights = torch.empty_like(weights)
        backward_buffer = torch.empty(num_layers, B, hidden_dim, device=grad.device, dtype=grad.dtype)
        scratch = _scratch(B, input_dim, output_dim, hidden_dim, num_layers, grad.device)
        with _timed("ffmlp_backward", B):
            _check(_lib.sdn_ffmlp_backward(_ptr(grad, torch.half, "grad"), _ptr(inputs), _ptr(weights), _ptr(forward_buffer), B, input_dim,
                                           output_dim, hidden_dim, num_layers, activation, output_activation, int(bool(calc_grad_inputs)),
                                           _ptr(backward_buffer), _ptr(grad_inputs), _ptr(grad_weights), _ptr(scratch), _stream()),
                   "ffmlp_backward")
        return grad_inputs, grad_weights, None, None, None, None, None, None, None, None


ffmlp_forward = _ffmlp_forward.apply


def convert_activation(act):
    """ffmlp.py:87-95."""
    return {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5}.get(act, 6)


class FFMLP(nn.Module):
    """ffmlp.py:98-168."""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation='relu'):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation('none')  # not supported by the reference either (:108)
        self.tensorcore_width = 16

        assert hidden_dim in [16, 32, 64, 128, 256], f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"

        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()

    def cleanup(self):
        """The reference frees its split-K side streams here (:134-136); this build owns none."""

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)
        std = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-std, std)

    def forward(self, inputs):
        """inputs [B, input_dim] -> [B, output_dim]  (:147-168, including its always-positive batch padding)."""
        B, C = inputs.shape
        pad = 128 - (B % 128)
        if pad > 0:
            inputs = torch.cat([inputs, torch.zeros(pad, C, dtype=inputs.dtype, device=inputs.device)], dim=0)
        outputs = ffmlp_forward(inputs, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers,
                                self.activation, self.output_activation, not self.training, inputs.requires_grad)
        if B != outputs.shape[0] or self.padded_output_dim != self.output_dim:
            outputs = outputs[:B, :self.output_dim]
        return outputs
