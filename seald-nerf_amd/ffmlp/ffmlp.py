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
    """One launch per direction: `sdn_ffmlp_forward` (keeps the hidden post-activations) or `sdn_ffmlp_inference`, and
    `sdn_ffmlp_backward` (activation-gradient chain + split-K weight gradients).  Argument order of `.apply` is the reference's:
    (inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference, calc_grad_inputs)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                inference=False, calc_grad_inputs=False):
        x, w = _dev(inputs).contiguous(), _dev(weights).contiguous()
        rows = x.shape[0]
        geom = (int(input_dim), int(output_dim), int(hidden_dim), int(num_layers))
        y = x.new_empty((rows, geom[1]))
        work = _scratch(rows, *geom, x.device)
        common = (_ptr(x, torch.half, "inputs"), _ptr(w, torch.half, "weights"), rows, *geom, int(activation), int(output_activation))
        if inference:
            with _timed("ffmlp_inference", rows):
                _check(_lib.sdn_ffmlp_inference(*common, _ptr(y), _ptr(work), _stream()), "ffmlp_inference")
            return y
        hidden = x.new_empty((geom[3], rows, geom[2]))        # post-activations of every hidden layer, for backward
        with _timed("ffmlp_forward", rows):
            _check(_lib.sdn_ffmlp_forward(*common, _ptr(hidden), _ptr(y), _ptr(work), _stream()), "ffmlp_forward")
        ctx.save_for_backward(x, w, y, hidden)
        ctx.dims = geom + (int(activation), int(output_activation), bool(calc_grad_inputs))
        return y

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        x, w, _, hidden = ctx.saved_tensors
        input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, want_dx = ctx.dims
        if activation == 2:
            raise NotImplementedError("ffmlp: the sine activation has no backward (needs pre-activations, utils.h:552-556)")
        g = grad.contiguous()
        rows = g.shape[0]
        dx = torch.empty_like(x) if want_dx else None
        dw = torch.empty_like(w)
        pre_grads = g.new_empty((num_layers, rows, hidden_dim))
        work = _scratch(rows, input_dim, output_dim, hidden_dim, num_layers, g.device)
        with _timed("ffmlp_backward", rows):
            _check(_lib.sdn_ffmlp_backward(_ptr(g, torch.half, "grad"), _ptr(x), _ptr(w), _ptr(hidden), rows, input_dim, output_dim, hidden_dim,
                                           num_layers, activation, output_activation, int(want_dx), _ptr(pre_grads), _ptr(dx), _ptr(dw),
                                           _ptr(work), _stream()), "ffmlp_backward")
        return (dx, dw) + (None,) * 8


ffmlp_forward = _ffmlp_forward.apply

_ACTIVATIONS = ("relu", "exponential", "sine", "sigmoid", "squareplus", "softplus")


def convert_activation(act):
    """Activation name -> the integer code of the kernels (ffmlp.py:87-95); anything else means 'none' (6)."""
    return _ACTIVATIONS.index(act) if act in _ACTIVATIONS else 6


class FFMLP(nn.Module):
    """Bias-free fp16 MLP  input_dim -> hidden_dim x num_layers -> output_dim (<= 16, padded to 16 inside) with ONE flat parameter
    `.weights` laid out [hidden, in] ++ (num_layers - 1) x [hidden, hidden] ++ [16, hidden] (ffmlp.py:98-168)."""

    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation='relu'):
        super().__init__()
        assert hidden_dim in [16, 32, 64, 128, 256], f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"
        self.input_dim, self.output_dim, self.hidden_dim, self.num_layers = input_dim, output_dim, hidden_dim, num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation('none')   # the reference supports nothing else either (:108)
        self.tensorcore_width = 16
        self.padded_output_dim = 16 * -(-output_dim // 16)
        self.num_parameters = hidden_dim * (input_dim + (num_layers - 1) * hidden_dim + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()

    def cleanup(self):
        """The reference frees its split-K side streams here (:134-136); this build owns none."""

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        """Fixed-seed U(-sqrt(3/hidden), +sqrt(3/hidden)), as the reference (:138-141)."""
        torch.manual_seed(42)
        bound = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-bound, bound)

    def forward(self, inputs):
        """[B, input_dim] -> [B, output_dim].  The batch is padded with 128 - B % 128 zero rows like the reference does (:154-157,
        always at least one: kept so that saved shapes match), though the kernels take any batch."""
        rows, cols = inputs.shape
        filler = inputs.new_zeros((128 - rows % 128, cols))
        padded = torch.cat([inputs, filler], dim=0)
        out = ffmlp_forward(padded, self.weights, self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers,
                            self.activation, self.output_activation, not self.training, padded.requires_grad)
        return out[:rows, :self.output_dim]
