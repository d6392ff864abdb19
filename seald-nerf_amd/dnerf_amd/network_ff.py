"""dnerf field network with its MLPs on the fused-MLP operator (MI355X-native training / op-by-op inference path).

The reference wires `ffmlp` only into the static NeRF (`nerf/network_ff.py`); its dnerf network (dnerf/network.py:38-96) is a
stack of bias-free `nn.Linear` layers under autocast -- 13 GEMMs forward, 26 backward, each with its own ReLU / cast kernels and
a round trip of the [M,128] activations through HBM.  `NeRFNetworkFF` keeps that network's parameters (same modules, same
state-dict names, checkpoints interchangeable) and evaluates the deformation and colour MLPs through `ffmlp_forward`
(one forward launch, one backward chain + split-K weight gradients per MLP; csrc/ffmlp.hip): the flat fp16 weight vector of the
operator is assembled from the Linear weights on every call, inside autograd, so gradients land on the Linear parameters.
Numerics are those of the autocast Linear stack (fp16 operands, fp32 accumulation, one rounding to fp16 per layer output).
The density MLP (one hidden layer) stays on hipBLASLt: the operator needs two hidden layers (ffmlp.py:103), like the reference's.
"""
import torch
import torch.nn.functional as F

from ffmlp import ffmlp_forward, convert_activation

from .network import NeRFNetwork, _run_mlp

_RELU, _NONE = convert_activation("relu"), convert_activation("none")


def flat_weights(layers):
    """Linear stack [in -> hidden x L -> out] -> the operator's flat fp16 layout (ffmlp.cu:631): [hidden, in16] ++ (L-1) x
    [hidden, hidden] ++ [16, hidden], the input columns zero-padded to a multiple of 16, the output rows to 16."""
    first, last = layers[0].weight, layers[-1].weight
    pad_in = -first.shape[1] % 16
    parts = [F.pad(first, (0, pad_in)).reshape(-1)]
    parts += [l.weight.reshape(-1) for l in layers[1:-1]]
    parts.append(F.pad(last, (0, 0, 0, 16 - last.shape[0])).reshape(-1))
    return torch.cat(parts).to(torch.float16)


def run_mlp_ff(layers, h, need_input_grad):
    """_run_mlp (ReLU between layers, none after the last) through the fused operator."""
    hidden, out_dim, in_dim = layers[0].out_features, layers[-1].out_features, layers[0].in_features
    if not torch.is_autocast_enabled("cuda"):   # fp32 run: the operator is an fp16 one
        return _run_mlp(layers, h)
    if len(layers) < 3 or hidden not in (16, 32, 64, 128, 256) or out_dim > 16 or any(l.out_features != hidden for l in layers[:-1]):
        return _run_mlp(layers, h)
    x = F.pad(h, (0, -in_dim % 16)).to(torch.float16)
    grad = torch.is_grad_enabled() and (x.requires_grad or layers[0].weight.requires_grad)
    y = ffmlp_forward(x, flat_weights(layers), x.shape[1], 16, hidden, len(layers) - 1, _RELU, _NONE, not grad,
                      bool(need_input_grad and x.requires_grad))
    return y[:, :out_dim]


class NeRFNetworkFF(NeRFNetwork):
    def forward(self, x, d, t):
        """dnerf/network.py:123-169; the canonical-frame rule `if t == 0: deform = 0` (:140) is a select on the device instead of a
        host branch, so that the call never synchronises (and can be captured into a graph, dnerf_amd/train_graph.py)."""
        deform = self._deform(x, t)
        deform = torch.where(t.reshape(()) == 0, torch.zeros_like(deform), deform)
        sigma, geo_feat = self._sigma(x + deform.to(x.dtype))
        return sigma, self._color(d, geo_feat), deform

    def _deform(self, x, t):
        enc_x = self.encoder_deform(x, bound=self.bound)
        enc_t = self.encoder_time(t)
        if enc_t.shape[0] == 1:
            enc_t = enc_t.expand(x.shape[0], -1)
        return run_mlp_ff(self.deform_net, torch.cat([enc_x, enc_t], dim=1), need_input_grad=False)

    def _color(self, d, geo_feat):
        h = torch.cat([self.encoder_dir(d).to(geo_feat.dtype), geo_feat], dim=-1)
        return torch.sigmoid(run_mlp_ff(self.color_net, h, need_input_grad=True))
