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
import os

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

    def __init__(self, model, time, fp16=True, max_points=None, table_layout=None):
        if not fp16:
            raise NotImplementedError("the fused field kernel implements the -O (fp16) configuration; use the op-by-op network for fp32")
        if not available():
            raise sdn_backend.SdnError("libsdn_hip was built without the fused field kernel")
        enc = model.encoder
        assert enc.gridtype == "tiled" and not enc.align_corners and enc.interpolation == "linear" and enc.num_levels == 16 and enc.level_dim == 2
        dev = enc.embeddings.device
        self.model = model
        self.weights = torch.from_numpy(pack_weights(model)).to(dev).contiguous()
        # fp16 copy of the table (grid.py:43-44 under autocast) in a layout of the kernel's own (csrc/field.hip, kLayout*):
        #   "quad" (default): one 16-byte block per row r of a level = rows {r, r+1, r+s1, r+s1+1} mod the level's size -- the four (x, y)
        #       corners of a cell -- so a level costs TWO gathers instead of four: the phase is bound by the rate at which a CU takes
        #       divergent gather addresses, not by bytes (4 x the table: 93 MiB for the default geometry, held in the Infinity Cache);
        #   "pad": level l moves down by l rows and is followed by one extra row that repeats its row 0, so that the (x, x+1) row pair of
        #       every gather is two consecutive rows even when x is the level's last row (`(index + 1) % hashmap_size` == 0,
        #       gridencoder.cu:66-84): four 8-byte gathers per level, no clamp / wrap bookkeeping.
        # Either is recognised by the kernel from the offsets that come with it (level sizes == 2 / 1 mod 8).
        self.layout = table_layout or os.environ.get("SDN_FIELD_TABLE", "quad")
        assert self.layout in ("quad", "pad"), self.layout
        off = enc.offsets.cpu().numpy().astype(np.int64)
        self._ref_offsets = np.ascontiguousarray(off.astype(np.int32))
        self._level_rows = [(int(off[l]), int(off[l + 1])) for l in range(16)]
        self.S = float(np.log2(enc.per_level_scale))
        self.H = int(enc.base_resolution)
        if self.layout == "quad":
            self.offsets_host = np.ascontiguousarray((off + 2 * np.arange(17)).astype(np.int32))
            self.table = torch.empty(int(self.offsets_host[-1]), 4, 2, dtype=torch.float16, device=dev)
        else:
            self.offsets_host = np.ascontiguousarray((off + np.arange(17)).astype(np.int32))
            self.table = torch.empty(int(self.offsets_host[-1]), 2, dtype=torch.float16, device=dev)
        self.load_table(enc.embeddings)
        self.bound = float(model.bound)
        self.density_scale = float(model.density_scale)
        self._time_cache, self._group_cache = {}, {}
        self.set_time(time)
        self._buf = None
        if max_points:
            self._alloc(max_points)

    @torch.no_grad()
    def load_table(self, embeddings):
        """(Re)fills the kernel's fp16 table from the network's embeddings [rows, 2] (float32 or float16, reference layout)."""
        if self.layout == "quad":
            emb = embeddings.detach()
            if emb.dtype not in (torch.float32, torch.float16):
                emb = emb.float()
            emb = emb.contiguous()
            check(sdn_backend.lib.sdn_field_build_quad_table(ptr(emb), sdn_backend.dtype_id(emb.dtype), self._ref_offsets.ctypes.data,
                                                             self.S, self.H, ptr(self.table), stream()), "field_build_quad_table")
            return
        for l, (a, b) in enumerate(self._level_rows):
            dst = int(self.offsets_host[l])
            self.table[dst:dst + (b - a)].copy_(embeddings[a:b])
            self.table[dst + (b - a)].copy_(embeddings[a])

    @staticmethod
    def time_value(time):
        """`time` (python number, or the reference's [B,1] tensor: one host read) as the float32 value the network sees."""
        v = float(time.reshape(-1)[0]) if isinstance(time, torch.Tensor) else float(time)
        return float(np.float32(v))

    def time_constants(self, time):
        """The constants one time stamp contributes to a frame, cached by VALUE: (bias0 [128] f32 -- the time encoding's
        contribution W0[:,63:76] . freq(t, 6) to the first deform layer, fp16 operands, fp32 sum --, zero_deform flag (t == 0: the
        canonical frame, dnerf/network.py:139-141), index of the occupancy-grid time slice (dnerf/renderer.py:285))."""
        t = self.time_value(time)
        hit = self._time_cache.get(t)
        if hit is None:
            dev = self.weights.device
            with torch.no_grad(), torch.autocast("cuda", enabled=False):
                enc_t = freq_encode(torch.tensor([[t]], dtype=torch.float32, device=dev), 6, 13).reshape(13)
                w = self.model.deform_net[0].weight.detach()[:, 63:76]
                bias0 = (w.to(torch.float16).float() @ enc_t.to(torch.float16).float()).contiguous()
            T = self.model.time_size
            t_idx = int(min(max(np.floor(np.float32(t) * np.float32(T)), 0), T - 1))
            hit = (bias0, int(t == 0.0), t_idx)
            if len(self._time_cache) >= 4096:
                self._time_cache.clear()
            self._time_cache[t] = hit
        return hit

    def invalidate_time_cache(self):
        """Call after the first deform layer's weights changed (training): cached biases were computed from the old weights."""
        self._time_cache.clear()

    def set_time(self, time):
        """Selects the time stamp `__call__` evaluates at (a frame loop passes times per frame instead)."""
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

    def _alloc(self, M):
        dev = self.weights.device
        self._buf = (torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, 3, dtype=torch.float32, device=dev))

    def __call__(self, xyzs, dirs, live_idx=None, live_count=None):
        M = xyzs.shape[0]
        if self._buf is None or self._buf[0].shape[0] < M:
            self._alloc(M)
        sigmas, rgbs = self._buf[0][:M], self._buf[1][:M]
        c = self.__dict__.get("_const")
        if c is None or c[0] is not self.weights or c[1] is not self.table or c[2] is not self.offsets_host:
            # (validated addresses of what does not change between calls: the reference-shaped loops are bound by host time)
            c = self._const = (self.weights, self.table, self.offsets_host, ptr(self.weights), ptr(self.table), self.offsets_host.ctypes.data)
        with sdn_backend.timed("field_forward_f16", M):
            check(sdn_backend.lib.sdn_field_forward_f16(ptr(xyzs, torch.float32, "xyzs"), ptr(dirs, torch.float32, "dirs"),
                                                        ptr(live_idx), ptr(live_count), M, c[3], ptr(self.bias0),
                                                        c[4], c[5], self.S, self.H, self.bound,
                                                        self.density_scale, self.zero_deform, ptr(sigmas), ptr(rgbs), stream()),
                  "field_forward_f16")
        return sigmas, rgbs


class DensityGridUpdater:
    """NeRFRenderer.update_extra_state (dnerf/renderer.py:453-555) on the device: the 64 x 128^3 density queries go through the
    CELLS variant of the fused field kernel (one launch per time slice and cascade, cell centres built and jittered in the kernel),
    the EMA / maximum update, the mean and the bitfield are three more kernels, and nothing is read back to the host.

    The partial update (`iter_density` 16..99) draws the same two populations as the reference -- N = H^3 / 4 uniformly random cells
    plus N uniformly random *occupied* cells, with repetition -- but picks the occupied ones by rank (cumsum + searchsorted) instead
    of `torch.nonzero`, which would synchronise once per slice, and hands the kernel the union sorted by Morton index.
    """

    def __init__(self, model, field=None, fp32=False):
        """fp32: the update of a model trained WITHOUT -O -- the queries go through the fp32 fused kernel (csrc/field_f32.hip, CELLS
        variant: `sdn_density_query_cells_f32`, within 1e-4 of the op-by-op fp32 network) on the model's fp32 table in place."""
        self.model = model
        dev = model.density_grid.device
        if field is None:
            t0 = model.times[0].reshape(1, 1).to(dev)
            if fp32:
                from .fused_f32 import FusedFieldF32
                field = FusedFieldF32(model, t0, variant="mfma32")
            else:
                field = FusedField(model, t0, fp16=True)
        self.field = field
        self.fp32 = not isinstance(field, FusedField)
        if self.fp32 and getattr(field, "variant", "mfma32") != "mfma32":
            raise sdn_backend.SdnError("the fp32 density query reads the fp32-MFMA kernel's weight packing (FusedFieldF32(variant='mfma32'))")
        n = model.grid_size ** 3
        self.tmp = torch.empty(n, dtype=torch.float32, device=dev)
        self.sum = torch.zeros(1, dtype=torch.float64, device=dev)
        self.mean = torch.zeros(2, dtype=torch.float32, device=dev)   # {mean density, threshold used}
        self.seed = 0x5EA1D

    def refresh(self):
        """Re-pack the (trained) weights and the fp16 table; call after optimizer steps."""
        f, enc = self.field, self.model.encoder
        if self.fp32:
            f.refresh()
            return
        f.weights.copy_(torch.from_numpy(pack_weights(self.model)))
        f.load_table(enc.embeddings.detach())
        f.invalidate_time_cache()
        f._group_cache.clear()

    def time_bias(self, times):
        """[T] (perturbed) times -> bias0 [T,128], the expression of FusedField.set_time for every slice at once."""
        with torch.no_grad(), torch.autocast("cuda", enabled=False):   # (the trainer calls update_extra_state under autocast)
            enc_t = freq_encode(times.reshape(-1, 1).float(), 6, 13)
            w = self.model.deform_net[0].weight.detach()[:, 63:76]
            if self.fp32:
                return (enc_t @ w.float().t()).contiguous()
            return (enc_t.to(torch.float16).float() @ w.to(torch.float16).float().t()).contiguous()

    def query_cells(self, out, bias0, zero_deform, cas_bound, cells=None, cell_count=None, n=None, noise=None, seed=0):
        """out[cell] = density_scale * sigma for the listed (or the first n) cells of one slice."""
        f = self.field
        if cells is not None:
            n = cells.shape[0]
        if self.fp32:
            with sdn_backend.timed("density_query_cells_f32", n):
                check(sdn_backend.lib.sdn_density_query_cells_f32(ptr(cells, torch.int32, "cells"), ptr(cell_count), n, ptr(noise, torch.float32, "noise"),
                                                                  int(seed) & 0xFFFFFFFF, self.model.grid_size, float(cas_bound), ptr(f.weights),
                                                                  ptr(bias0, torch.float32, "bias0"), ptr(f.table, torch.float32, "embeddings"),
                                                                  f.offsets_host.ctypes.data, f.S, f.H, f.bound, f.density_scale, int(zero_deform),
                                                                  ptr(out, torch.float32, "out"), stream()), "density_query_cells_f32")
            return out
        with sdn_backend.timed("density_query_cells_f16", n):
            check(sdn_backend.lib.sdn_density_query_cells_f16(ptr(cells, torch.int32, "cells"), ptr(cell_count), n, ptr(noise, torch.float32, "noise"),
                                                              int(seed) & 0xFFFFFFFF, self.model.grid_size, float(cas_bound), ptr(f.weights),
                                                              ptr(bias0, torch.float32, "bias0"), ptr(f.table), f.offsets_host.ctypes.data,
                                                              f.S, f.H, f.bound, f.density_scale, int(zero_deform), ptr(out, torch.float32, "out"),
                                                              stream()), "density_query_cells_f16")
        return out

    @torch.no_grad()
    def partial_cells(self):
        """[T, C, 2N] int32 cell lists and [T, C] live counts of the partial update (dnerf/renderer.py:503-515), no host sync."""
        import raymarching
        m = self.model
        G, T, C = m.grid_size, m.time_size, m.cascade
        dev = m.density_grid.device
        N = G ** 3 // 4
        coords = torch.randint(0, G, (T * C * N, 3), device=dev, dtype=torch.int32)
        uniform = raymarching.morton3D(coords).view(T, C, N)
        ranks = torch.cumsum((m.density_grid > 0).view(T * C, -1), dim=1, dtype=torch.int32)           # rank of every occupied cell
        total = ranks[:, -1:]                                                                            # [T*C, 1] occupied cells
        pick = torch.minimum((torch.rand(T * C, N, device=dev) * total).to(torch.int32), (total - 1).clamp(min=0))
        occupied = torch.searchsorted(ranks, pick, right=True).to(torch.int32)
        occupied = torch.where(total > 0, occupied, 0x7FFFFFFF).view(T, C, N)      # nothing occupied: sentinels, sorted past the live count
        # Sorted by Morton index: neighbouring lanes then gather neighbouring table rows (measured 0.33 ms instead of 0.64 ms per
        # slice for the query).  The order of the list carries no meaning in the reference (tmp_grid[indices] = sigmas).
        cells = torch.sort(torch.cat([uniform, occupied], dim=2), dim=2).values.contiguous()
        counts = (N + N * (total.view(T, C) > 0)).to(torch.int32).contiguous()                          # no occupied cell: uniform part only
        return cells, counts

    @torch.no_grad()
    def update(self, decay=0.95, noise=None, time_noise=None):
        """One update_extra_state pass (density grid + bitfield).  noise [T, C, n, 3] / time_noise [T, C] uniform [0,1): supply what
        torch.rand_like would have drawn (tests); default = in-kernel generator / host generator."""
        m = self.model
        G, T, C = m.grid_size, m.time_size, m.cascade
        H3 = G ** 3
        dev = m.density_grid.device
        full = m.iter_density < 16
        partial = (not full) and m.iter_density < 100
        if time_noise is None:
            time_noise = torch.rand(T, C)
        times = m.times.reshape(T, 1).cpu() + (time_noise.cpu().float() * 2 - 1) * (0.5 / m.time_size)     # [T, C] dnerf/renderer.py:486,492
        bias = self.time_bias(times.reshape(-1).to(dev)).view(T, C, 128)
        zero = (times == 0).tolist()
        self.sum.zero_()
        cells = counts = None
        if partial:
            cells, counts = self.partial_cells()
        self.seed = (self.seed * 1664525 + 1013904223) & 0xFFFFFFFF
        grid = m.density_grid
        for t in range(T):
            for cas in range(C):
                cas_bound = min(2 ** cas, m.bound)
                nz = None if noise is None else noise[t, cas]
                if full:
                    self.query_cells(self.tmp, bias[t, cas], zero[t][cas], cas_bound, n=H3, noise=nz, seed=self.seed + 64 * t + cas)
                else:
                    self.tmp.fill_(-1.0)
                    if partial:
                        self.query_cells(self.tmp, bias[t, cas], zero[t][cas], cas_bound, cells=cells[t, cas], cell_count=counts[t, cas:cas + 1],
                                         noise=nz, seed=self.seed + 64 * t + cas)
                check(sdn_backend.lib.sdn_density_grid_ema(ptr(grid[t, cas]), ptr(self.tmp), H3, float(decay), ptr(self.sum), stream()),
                      "density_grid_ema")
        check(sdn_backend.lib.sdn_density_grid_pack(ptr(grid), grid.numel(), ptr(self.sum), float(m.density_thresh), ptr(self.mean),
                                                    ptr(m.density_bitfield), stream()), "density_grid_pack")
        m.iter_density += 1
        return self.mean
