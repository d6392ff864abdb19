"""Fused field evaluation in fp32 (the reference WITHOUT `-O`): host side of csrc/field_f32.hip.

Packs the network's fp32 Linear weights into the A-operand order of v_mfma_f32_32x32x2_f32 -- per layer a block [pair][lane][m-tile] of
floats: lane l of k-pair p and output tile mt holds W[32 mt + (l & 31)][kmap[p][l >> 5]] -- where `kmap` says which input feature the
two k positions of a pair mean for that layer:
  * layers fed by a previous layer: pair 16 t + v = (row, row + 4), row = 32 t + 8 (v >> 2) + (v & 3) -- accumulator register v of the
    previous layer's output tile t is that pair's B operand as it stands;
  * first deform layer: pair 3 f + d = (sin, cos) of 2^f x_d (ONE sine per lane, the reference's phase shift); pair 30 = (x0, x1),
    pair 31 = (x2, -); the time encoding's 13 columns become the bias row W0[:, 63:76] . freq(t, 6);
  * first sigma layer: pair l = the two channels of grid level l;
  * first colour layer: pairs 0..7 = SH coefficients (2p, 2p + 1), pairs 8..15 = accumulator registers 0..7 of the sigma net's output
    (rows (r, r + 4)); row 0, the density logit, gets a zero column.
The embedding table is used where it is (the model's own fp32 tensor, reference layout): no copy.
"""
import numpy as np
import torch

import sdn_backend
from sdn_backend import check, ptr, stream
from freqencoder import freq_encode


def available():
    return hasattr(sdn_backend.lib, "sdn_field_forward_f32")


def _acc_pairs(tiles):
    out = []
    for t in range(tiles):
        for v in range(16):
            r = 32 * t + 8 * (v >> 2) + (v & 3)
            out.append((r, r + 4))
    return out


def _pack_layer(W, pairs, mtiles):
    """W [out, in] float32, pairs: list of (k_lo, k_hi) input columns (-1 = none) -> [len(pairs), 64, mtiles] float32."""
    W = np.asarray(W, dtype=np.float32)
    out_dim, in_dim = W.shape
    Wp = np.zeros((32 * mtiles, in_dim + 1), dtype=np.float32)      # (+1: a zero column for k = -1; rows past out_dim stay zero)
    Wp[:out_dim, :in_dim] = W
    km = np.array([[k if 0 <= k < in_dim else in_dim for k in pr] for pr in pairs], dtype=np.int64)      # [P, 2]
    lane = np.arange(64)
    m = 32 * np.arange(mtiles)[None, None, :] + (lane & 31)[None, :, None]                                  # [1, 64, MT]
    k = km[:, lane >> 5][:, :, None]                                                                        # [P, 64, 1]
    return np.ascontiguousarray(Wp[m, k]).astype(np.float32)


def pack_weights_f32(model):
    """All Linear weights of the dnerf field network in the kernel's stage order (D0 | D1..D6 | D7 S0 S1 C0 C1 C2), flat float32."""
    g = lambda lin: lin.weight.detach().float().cpu().numpy()   # noqa: E731
    dn, sn, cn = model.deform_net, model.sigma_net, model.color_net
    d0 = []
    for p in range(30):
        f, d = p // 3, p % 3
        d0.append((3 + (2 * f) * 3 + d, 3 + (2 * f + 1) * 3 + d))     # columns of [x | sin 2^0 x | cos 2^0 x | sin 2^1 x | ...]
    d0 += [(0, 1), (2, -1)]
    parts = [_pack_layer(g(dn[0])[:, :63], d0, 4)]
    for l in range(1, 7):
        parts.append(_pack_layer(g(dn[l]), _acc_pairs(4), 4))
    parts.append(_pack_layer(g(dn[7]), _acc_pairs(4), 1))
    parts.append(_pack_layer(g(sn[0]), [(2 * l, 2 * l + 1) for l in range(16)], 2))
    parts.append(_pack_layer(g(sn[1]), _acc_pairs(2), 1))
    c0 = [(2 * p, 2 * p + 1) for p in range(8)]
    for v in range(8):                                                  # geo_feat[j] = h[1 + j] sits in column 16 + j
        r = 8 * (v >> 2) + (v & 3)
        c0.append((16 + r - 1 if r >= 1 else -1, 16 + r + 4 - 1))
    parts.append(_pack_layer(g(cn[0]), c0, 2))
    parts.append(_pack_layer(g(cn[1]), _acc_pairs(2), 2))
    parts.append(_pack_layer(g(cn[2]), _acc_pairs(2), 1))
    flat = np.concatenate([p.reshape(-1) for p in parts]).astype(np.float32)
    assert flat.shape[0] == int(sdn_backend.lib.sdn_field_weight_floats_f32()), flat.shape
    return flat


def pack_weights_f32_split(model):
    """The same weights for csrc/field_f32x3.hip: every fp32 weight as hi + lo (two fp16 values), in the fp16 kernel's operand order
    (fused.py kmaps), per layer [k-step][lane][m-tile][hi | lo][8 halves]; flat uint16, stage order D0 | D1..D6 | D7 S0 S1 C0 C1 C2."""
    from . import fused as F16
    g = lambda lin: lin.weight.detach().float().cpu().numpy()   # noqa: E731
    dn, sn, cn = model.deform_net, model.sigma_net, model.color_net
    hidden128 = [F16._acc_kmap(t, s) for t in range(4) for s in range(2)]
    hidden64 = [F16._acc_kmap(t, s) for t in range(2) for s in range(2)]

    def layer(W, n_mt, kmaps):
        W = np.asarray(W, dtype=np.float32) * np.float32(256.0)     # kWS of the kernel: keeps the lo parts out of the fp16 subnormals the MFMA flushes
        hi = W.astype(np.float16)
        lo = (W - hi.astype(np.float32)).astype(np.float16)
        out = []
        for part in (hi, lo):      # _pack_layer rounds its input to fp16: exact for both parts
            blk = F16._pack_layer(part.astype(np.float32), n_mt, kmaps).reshape(n_mt, len(kmaps), 64, 8)
            out.append(blk.transpose(1, 2, 0, 3))                       # [ks, lane, mt, 8]
        return np.ascontiguousarray(np.stack(out, axis=3)).reshape(-1)   # [ks, lane, mt, 2, 8]

    parts = [layer(g(dn[0])[:, :63], 4, [F16._d0_kmap(s) for s in range(4)])]
    parts += [layer(g(dn[l]), 4, hidden128) for l in range(1, 7)]
    parts.append(layer(g(dn[7]), 1, hidden128))
    parts.append(layer(g(sn[0]), 2, [F16._s0_kmap(s) for s in range(2)]))
    parts.append(layer(g(sn[1]), 1, hidden64))
    parts.append(layer(g(cn[0]), 2, [F16._c0_kmap(s) for s in range(2)]))
    parts.append(layer(g(cn[1]), 2, hidden64))
    parts.append(layer(g(cn[2]), 1, hidden64))
    flat = np.concatenate(parts).view(np.uint16)
    assert flat.shape[0] * 2 == 4 * int(sdn_backend.lib.sdn_field_weight_floats_f32()), flat.shape
    return flat


class FusedFieldF32:
    """Callable (xyzs [M,3], dirs [M,3]) -> (sigmas [M] f32, rgbs [M,3] f32): the fp32 network (dnerf/network.py:123-169 without
    autocast) in one launch.  Same interface as `fused.FusedField` (time constants per value, live lists)."""

    def __init__(self, model, time, max_points=None, variant=None):
        """variant: "mfma32" (default; csrc/field_f32.hip: v_mfma_f32_32x32x2_f32, 1e-4 from the fp32 network) or "split"
        (csrc/field_f32x3.hip: fp32 operands as hi + lo fp16 pairs on the fp16 MFMAs -- 22-bit operands: worst element 1.3e-4, faster);
        SDN_FIELD_F32 in the environment overrides the default."""
        import os
        self.variant = variant or os.environ.get("SDN_FIELD_F32", "mfma32")
        assert self.variant in ("split", "mfma32"), self.variant
        if not available():
            raise sdn_backend.SdnError("libsdn_hip was built without the fp32 fused field kernel")
        enc = model.encoder
        assert enc.gridtype == "tiled" and not enc.align_corners and enc.interpolation == "linear" and enc.num_levels == 16 and enc.level_dim == 2
        assert len(model.deform_net) == 8 and model.hidden_dim_deform == 128 and model.hidden_dim == 64 and model.geo_feat_dim == 15
        if enc.embeddings.dtype != torch.float32:
            raise sdn_backend.SdnError("the fp32 fused field reads the model's fp32 embedding table in place")
        dev = enc.embeddings.device
        self.model = model
        self.weights = self._packed().to(dev).contiguous()
        self.table = enc.embeddings.detach()
        self.offsets_host = np.ascontiguousarray(enc.offsets.cpu().numpy().astype(np.int32))
        self.S = float(np.log2(enc.per_level_scale))
        self.H = int(enc.base_resolution)
        self.bound = float(model.bound)
        self.density_scale = float(model.density_scale)
        self._time_cache, self._group_cache = {}, {}
        self.set_time(time)
        self._buf = None
        if max_points:
            self._alloc(max_points)

    def _packed(self):
        if self.variant == "split":
            return torch.from_numpy(pack_weights_f32_split(self.model).view(np.int16))
        return torch.from_numpy(pack_weights_f32(self.model))

    @staticmethod
    def time_value(time):
        v = float(time.reshape(-1)[0]) if isinstance(time, torch.Tensor) else float(time)
        return float(np.float32(v))

    def time_constants(self, time):
        """(bias0 [128] f32 = W0[:, 63:76] . freq(t, 6) in fp32, t == 0 flag (dnerf/network.py:139-141), occupancy slice index)."""
        t = self.time_value(time)
        hit = self._time_cache.get(t)
        if hit is None:
            dev = self.weights.device
            with torch.no_grad(), torch.autocast("cuda", enabled=False):
                enc_t = freq_encode(torch.tensor([[t]], dtype=torch.float32, device=dev), 6, 13).reshape(13)
                w = self.model.deform_net[0].weight.detach().float()[:, 63:76]
                bias0 = (w @ enc_t).contiguous()
            T = self.model.time_size
            t_idx = int(min(max(np.floor(np.float32(t) * np.float32(T)), 0), T - 1))
            hit = (bias0, int(t == 0.0), t_idx)
            if len(self._time_cache) >= 4096:
                self._time_cache.clear()
            self._time_cache[t] = hit
        return hit

    def set_time(self, time):
        self.bias0, self.zero_deform, self.t_idx = self.time_constants(time)

    def group_constants(self, times):
        """(bias0 [F,128] contiguous, zero_deform bit mask, slice indices) for the F frames of a frame group; cached by value."""
        key = tuple(self.time_value(t) for t in times)
        hit = self._group_cache.get(key)
        if hit is None:
            parts = [self.time_constants(t) for t in key]
            bias = torch.stack([p[0] for p in parts]).contiguous()
            mask = sum(p[1] << f for f, p in enumerate(parts))
            hit = (bias, mask, [p[2] for p in parts])
            if len(self._group_cache) >= 1024:
                self._group_cache.clear()
            self._group_cache[key] = hit
        return hit

    def refresh(self):
        """Re-pack after the weights changed (the table is read in place)."""
        self.weights.copy_(self._packed())
        self.table = self.model.encoder.embeddings.detach()
        self._time_cache.clear()
        self._group_cache.clear()

    def _alloc(self, M):
        dev = self.weights.device
        self._buf = (torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, 3, dtype=torch.float32, device=dev))

    def __call__(self, xyzs, dirs, live_idx=None, live_count=None, deform=None):
        """deform: optional [M,3] f32 tensor that receives the deformation network's output (the third value of NeRFNetwork.forward)."""
        M = xyzs.shape[0]
        if self._buf is None or self._buf[0].shape[0] < M:
            self._alloc(M)
        sigmas, rgbs = self._buf[0][:M], self._buf[1][:M]
        with sdn_backend.timed("field_forward_f32", M):
            entry = sdn_backend.lib.sdn_field_forward_f32x3 if self.variant == "split" else sdn_backend.lib.sdn_field_forward_f32
            check(entry(ptr(xyzs, torch.float32, "xyzs"), ptr(dirs, torch.float32, "dirs"),
                                                        ptr(live_idx), ptr(live_count), M, ptr(self.weights), ptr(self.bias0),
                                                        ptr(self.table, torch.float32, "embeddings"), self.offsets_host.ctypes.data,
                                                        self.S, self.H, self.bound, self.density_scale, self.zero_deform, ptr(sigmas),
                                                        ptr(rgbs), ptr(deform, torch.float32, "deform"), stream()), "field_forward_f32")
        return sigmas, rgbs
