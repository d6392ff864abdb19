"""MI355X-native fused field evaluation (encoders + MLPs on MFMA in one launch) for the inference loop.

Host side of csrc/field.hip: packs the network's Linear weights into MFMA fragment order once per model,
computes the per-frame time-encoding bias of the first deform layer, and launches the kernel through the C ABI.

Fragment order (v_mfma_f32_32x32x16_f16, weights = A operand): block (layer, m-tile, k-step) is 64 lanes x 8
halfs; lane l holds W[32*mt + (l & 31)][kmap(ks, l >> 5, j)], j = 0..7.  `kmap` encodes which input feature a
k position means for that layer:
  * hidden layers: the accumulator-as-operand order 32t + 16s + 8(j>>2) + 4h + (j&3) (t = previous m-tile);
  * first deform layer: lane-half h owns (freq, dim) pairs 15h..15h+14 as (sin, cos) plus x0,x1 / x2,pad;
  * first sigma layer: lane-half h owns grid levels 8h..8h+7 (2 channels each);
  * first colour layer: k-step 0 = the 16 outputs of the sigma net in accumulator order (the density logit gets a
    zero column), k-step 1 = SH coefficient 8h + j.
"""
import numpy as np
import torch

import sdn_backend
from sdn_backend import check, ptr, stream
from freqencoder import freq_encode

def available():
    return hasattr(sdn_backend.lib, "sdn_field_forward_f16")


def _acc_kmap(t, s):
    h = np.arange(2)[:, None]
    j = np.arange(8)[None, :]
    return 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)


def _d0_kmap(s):
    km = np.full((2, 8), -1, dtype=np.int64)
    for h in range(2):
        for j in range(8):
            q = 8 * s + j
            if q < 30:
                pr = q >> 1
                f, dd, trig = 5 * h + pr // 3, pr % 3, q & 1
                km[h, j] = 3 + (2 * f + trig) * 3 + dd   # column of [x | sin 2^0 x | cos 2^0 x | sin 2^1 x | ...]
            elif q == 30:
                km[h, j] = 2 if h else 0
            else:
                km[h, j] = -1 if h else 1
    return km


def _s0_kmap(s):
    h = np.arange(2)[:, None]
    j = np.arange(8)[None, :]
    return 2 * (8 * h + 4 * s + (j >> 1)) + (j & 1)


def _c0_kmap(s):
    h = np.arange(2)[:, None]
    j = np.arange(8)[None, :]
    if s == 0:
        i = 8 * (j >> 2) + 4 * h + (j & 3)       # index into the sigma net's 16 outputs
        return np.where(i >= 1, 15 + i, -1)        # colour input = [SH(16), geo_feat(15)], geo_feat[i-1] = h[i]
    return 8 * h + j + 0 * j                       # SH coefficient


def _pack_layer(W, n_mt, kmaps):
    """W [out, in] float32 -> [n_mt * len(kmaps), 64, 8] float16 in (mt, ks) block order."""
    out_dim = W.shape[0]
    lanes = np.arange(64)
    rows_in_tile, hh = lanes & 31, lanes >> 5
    blocks = np.zeros((n_mt, len(kmaps), 64, 8), dtype=np.float16)
    Wh = W.astype(np.float16)
    for mt in range(n_mt):
        rows = 32 * mt + rows_in_tile
        row_ok = rows < out_dim
        for ks, km in enumerate(kmaps):
            cols = km[hh]                                # [64, 8]
            ok = row_ok[:, None] & (cols >= 0)
            vals = Wh[np.clip(rows, 0, out_dim - 1)[:, None], np.clip(cols, 0, W.shape[1] - 1)]
            blocks[mt, ks] = np.where(ok, vals, np.float16(0))
    return blocks.reshape(-1, 64, 8)


def pack_weights(model):
    """All Linear weights of the field network in the kernel's block order: [240, 64, 8] float16."""
    g = lambda m: m.weight.detach().float().cpu().numpy()  # noqa: E731
    dn, sn, cn = model.deform_net, model.sigma_net, model.color_net
    assert len(dn) == 8 and len(sn) == 2 and len(cn) == 3, "fused kernel is built for the dnerf network shape"
    assert g(dn[0]).shape == (128, 76) and g(dn[7]).shape == (3, 128) and g(sn[0]).shape == (64, 32) and g(sn[1]).shape == (16, 64)
    assert g(cn[0]).shape == (64, 31) and g(cn[1]).shape == (64, 64) and g(cn[2]).shape == (3, 64)
    hidden128 = [_acc_kmap(t, s) for t in range(4) for s in range(2)]
    hidden64 = [_acc_kmap(t, s) for t in range(2) for s in range(2)]
    parts = [_pack_layer(g(dn[0])[:, :63], 4, [_d0_kmap(s) for s in range(4)])]
    parts += [_pack_layer(g(dn[l]), 4, hidden128) for l in range(1, 7)]
    parts.append(_pack_layer(g(dn[7]), 1, hidden128))
    parts.append(_pack_layer(g(sn[0]), 2, [_s0_kmap(s) for s in range(2)]))
    parts.append(_pack_layer(g(sn[1]), 1, hidden64))
    parts.append(_pack_layer(g(cn[0]), 2, [_c0_kmap(s) for s in range(2)]))
    parts.append(_pack_layer(g(cn[1]), 2, hidden64))
    parts.append(_pack_layer(g(cn[2]), 1, hidden64))
    packed = np.concatenate(parts, axis=0)
    assert packed.shape[0] == int(sdn_backend.lib.sdn_field_weight_blocks()), packed.shape
    return packed


class FusedField:
    """Callable (xyzs [M,3], dirs [M,3]) -> (sigmas [M] f32, rgbs [M,3] f32) with the reference's -O numerics."""

    def __init__(self, model, time, fp16=True, max_points=None):
        if not fp16:
            raise NotImplementedError("the fused field kernel implements the -O (fp16) configuration; use the op-by-op network for fp32")
        if not available():
            raise sdn_backend.SdnError("libsdn_hip was built without the fused field kernel")
        enc = model.encoder
        assert enc.gridtype == "tiled" and not enc.align_corners and enc.interpolation == "linear" and enc.num_levels == 16 and enc.level_dim == 2
        dev = enc.embeddings.device
        self.model = model
        self.weights = torch.from_numpy(pack_weights(model)).to(dev).contiguous()
        self.table = enc.embeddings.detach().to(torch.float16).contiguous()   # grid.py:43-44 under autocast
        self.offsets_host = np.ascontiguousarray(enc.offsets.cpu().numpy().astype(np.int32))
        self.S = float(np.log2(enc.per_level_scale))
        self.H = int(enc.base_resolution)
        self.bound = float(model.bound)
        self.density_scale = float(model.density_scale)
        self.set_time(time)
        self._buf = None
        if max_points:
            self._alloc(max_points)

    def set_time(self, time):
        """Per-frame constants: the time encoding's contribution to the first deform layer (fp16 operands, fp32 sum)."""
        with torch.no_grad():
            enc_t = freq_encode(time.reshape(1, 1).float(), 6, 13).reshape(13)
            w = self.model.deform_net[0].weight.detach()[:, 63:76]
            self.bias0 = (w.to(torch.float16).float() @ enc_t.to(torch.float16).float()).contiguous()
        self.zero_deform = int(float(time.reshape(-1)[0]) == 0.0)

    def _alloc(self, M):
        dev = self.weights.device
        self._buf = (torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, 3, dtype=torch.float32, device=dev))

    def __call__(self, xyzs, dirs, live_idx=None, live_count=None):
        M = xyzs.shape[0]
        if self._buf is None or self._buf[0].shape[0] < M:
            self._alloc(M)
        sigmas, rgbs = self._buf[0][:M], self._buf[1][:M]
        with sdn_backend.timed("field_forward_f16", M):
            check(sdn_backend.lib.sdn_field_forward_f16(ptr(xyzs, torch.float32, "xyzs"), ptr(dirs, torch.float32, "dirs"),
                                                        ptr(live_idx), ptr(live_count), M, ptr(self.weights), ptr(self.bias0),
                                                        ptr(self.table), self.offsets_host.ctypes.data, self.S, self.H, self.bound,
                                                        self.density_scale, self.zero_deform, ptr(sigmas), ptr(rgbs), stream()),
                  "field_forward_f16")
        return sigmas, rgbs
