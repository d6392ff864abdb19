"""CPU oracle of the dynamic-NeRF field network (test infrastructure only).

Restates `NeRFNetwork.forward` / `density` / `color` (dnerf/network.py:123-257 of the reference) on the
oracle encoders.  Weights arrive as a plain dict of numpy arrays (a model state_dict moved to the CPU).

Precision modes:
  * "fp32": every Linear is evaluated with float64 accumulation and rounded once to float32 (the
    "fp64 shadow" of the reference's fp32 GEMM: any correct fp32 GEMM agrees with it to ~1e-6 relative);
  * "fp16": emulates the reference under `-O` (torch autocast): Linear inputs and weights rounded to fp16,
    exact products accumulated wide, output rounded to fp16; grid table and output fp16 (grid.py:43-44);
    trunc_exp / SH / freq in fp32 (their custom_fwd casts); sigmoid output fp16.
"""
import numpy as np

from . import oracle as O


def _linear(x, w, mode):
    if mode == "fp16":
        y = x.astype(np.float16).astype(np.float64) @ w.astype(np.float16).astype(np.float64).T
        return y.astype(np.float32).astype(np.float16)
    return (x.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)


def _mlp(x, weights, mode):
    h = x
    for i, w in enumerate(weights):
        h = _linear(h, w, mode)
        if i != len(weights) - 1:
            h = np.maximum(h, 0)
    return h


def _weights(state, prefix):
    keys = sorted((k for k in state if k.startswith(prefix + ".") and k.endswith(".weight")), key=lambda k: int(k.split(".")[1]))
    return [np.asarray(state[k], np.float32) for k in keys]


class FieldOracle:
    def __init__(self, state, bound=1.0, density_scale=1.0, per_level_scale=None, base_resolution=16, mode="fp32"):
        self.deform = _weights(state, "deform_net")
        self.sigma = _weights(state, "sigma_net")
        self.color = _weights(state, "color_net")
        self.emb = np.asarray(state["encoder.embeddings"], np.float32)
        self.offsets = np.asarray(state["encoder.offsets"], np.int32)
        self.bound = float(bound)
        self.mode = mode
        self.density_scale = density_scale
        self.H = base_resolution
        L = self.offsets.shape[0] - 1
        self.pls = per_level_scale if per_level_scale is not None else np.exp2(np.log2(2048 * bound / base_resolution) / (L - 1))
        self._emb16 = self.emb.astype(np.float16) if mode == "fp16" else None
        # optional background-sphere model (dnerf/network.py:99-121): 2-D hash grid (4 levels, 16 -> 2048) ++ SH(dir) -> bg_net
        self.bg = _weights(state, "bg_net") if any(k.startswith("bg_net.") for k in state) else None
        if self.bg is not None:
            self.bg_emb = np.asarray(state["encoder_bg.embeddings"], np.float32)
            self.bg_offsets = np.asarray(state["encoder_bg.offsets"], np.int32)
            self.bg_pls = np.exp2(np.log2(2048 / 16) / (self.bg_offsets.shape[0] - 2))

    def deform_of(self, x, t):
        enc_x = O.freq_encode_forward(x, 10)
        enc_t = O.freq_encode_forward(np.asarray(t, np.float32).reshape(1, 1), 6)
        h = np.concatenate([enc_x, np.repeat(enc_t, x.shape[0], 0)], axis=1)
        return _mlp(h, self.deform, self.mode)

    def sigma_of(self, x):
        u = ((x + np.float32(self.bound)) / np.float32(2 * self.bound)).astype(np.float32)
        table = self._emb16 if self.mode == "fp16" else self.emb
        enc, _ = O.grid_encode_forward(u, table, self.offsets, self.pls, self.H, False, 1, False, 0)
        h = _mlp(enc, self.sigma, self.mode)
        sigma = np.exp(h[:, 0].astype(np.float32)).astype(np.float32)
        return sigma, h[:, 1:]

    def color_of(self, d, geo_feat):
        sh, _ = O.sh_encode_forward(d, 4)
        h = np.concatenate([sh, geo_feat.astype(np.float32)], axis=1)
        h = _mlp(h, self.color, self.mode).astype(np.float32)
        rgb = (1.0 / (1.0 + np.exp(-h.astype(np.float64)))).astype(np.float32)
        if self.mode == "fp16":
            rgb = rgb.astype(np.float16).astype(np.float32)
        return rgb

    def background(self, sph, d):
        """dnerf/network.py:208-223: sph [N,2] in [-1,1] (raymarching.sph_from_ray), d [N,3] -> rgb [N,3]."""
        u = ((np.asarray(sph, np.float32) + np.float32(1)) / np.float32(2)).astype(np.float32)
        table = self.bg_emb.astype(np.float16) if self.mode == "fp16" else self.bg_emb
        enc, _ = O.grid_encode_forward(u, table, self.bg_offsets, self.bg_pls, 16, False, 0, False, 0)      # gridtype 0 = hash
        sh, _ = O.sh_encode_forward(np.asarray(d, np.float32), 4)
        h = _mlp(np.concatenate([sh, enc.astype(np.float32)], axis=1), self.bg, self.mode).astype(np.float32)
        return (1.0 / (1.0 + np.exp(-h.astype(np.float64)))).astype(np.float32)

    def forward(self, x, d, t):
        """-> sigma [M] f32 (already times density_scale), rgb [M,3] f32, deform [M,3]."""
        x = np.asarray(x, np.float32)
        deform = self.deform_of(x, t)
        if float(np.asarray(t).reshape(-1)[0]) == 0.0:
            deform = np.zeros_like(x)
        xd = (x + deform.astype(np.float32)).astype(np.float32)
        sigma, geo = self.sigma_of(xd)
        return (np.float32(self.density_scale) * sigma).astype(np.float32), self.color_of(np.asarray(d, np.float32), geo), deform

    def density(self, x, t):
        x = np.asarray(x, np.float32)
        deform = self.deform_of(x, t)
        xd = x if float(np.asarray(t).reshape(-1)[0]) == 0.0 else (x + deform.astype(np.float32)).astype(np.float32)
        sigma, geo = self.sigma_of(xd)
        return {"deform": deform, "sigma": sigma, "geo_feat": geo}
