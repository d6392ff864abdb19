"""CPU stand-in for the reference's `gridencoder` package (test infrastructure; see ../README.md): gridencoder/grid.py:24-161 on
the CPU oracle, fp32 table."""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from oracle import oracle as O

_gridtype_to_id = {"hash": 0, "tiled": 1}
_interp_to_id = {"linear": 0, "smoothstep": 1}


class _GridEncode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False,
                interpolation=0):
        x = np.ascontiguousarray(inputs.detach().numpy(), np.float32)
        emb = np.ascontiguousarray(embeddings.detach().numpy(), np.float32)
        off = np.ascontiguousarray(offsets.numpy(), np.int32)
        out, dy_dx = O.grid_encode_forward(x, emb, off, per_level_scale, int(base_resolution), bool(calc_grad_inputs), int(gridtype),
                                           bool(align_corners), int(interpolation))
        ctx.saved = (x, emb, off, dy_dx, per_level_scale, int(base_resolution), int(gridtype), bool(align_corners), int(interpolation))
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, grad):
        x, emb, off, dy_dx, pls, H, gridtype, align, interp = ctx.saved
        g_emb, g_in = O.grid_encode_backward(np.ascontiguousarray(grad.numpy(), np.float32), x, emb, off, pls, H, dy_dx, gridtype, align, interp)
        return (torch.from_numpy(g_in) if g_in is not None else None), torch.from_numpy(g_emb), None, None, None, None, None, None, None


grid_encode = _GridEncode.apply


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False, interpolation="linear"):
        super().__init__()
        offsets, per_level_scale = O.grid_offsets(input_dim, num_levels, level_dim, per_level_scale, base_resolution, log2_hashmap_size,
                                                  desired_resolution, align_corners)
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype_id, self.interp_id, self.align_corners = _gridtype_to_id[gridtype], _interp_to_id[interpolation], align_corners
        self.register_buffer("offsets", torch.from_numpy(offsets))
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)
        lead = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        out = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution, inputs.requires_grad,
                          self.gridtype_id, self.align_corners, self.interp_id)
        return out.view(lead + [self.output_dim])
