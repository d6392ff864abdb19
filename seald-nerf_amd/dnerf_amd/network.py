"""Dynamic-NeRF field network (deform + sigma + color MLPs over the HIP encoders).

Host-side mirror of /root/reference/dnerf/network.py:10-275 (`NeRFNetwork`): the same sub-module and
parameter names (`encoder_deform`, `encoder_time`, `deform_net.{l}.weight`, `encoder.embeddings`,
`encoder.offsets`, `sigma_net`, `encoder_dir`, `color_net`), so a reference checkpoint's model state
dict loads with `strict=False` exactly as the reference's own loader does (nerf/utils.py:1095-1154).

Two evaluation paths:
  * `forward` / `density` / `color`  -- the reference's op sequence on the drop-in operators
    (`freq_encode`, `grid_encode`, `sh_encode`) with the MLPs as `F.linear` calls; differentiable.
  * the fused dispatch inside `forward`: in eval mode, without autograd, under fp16 autocast (the reference's `-O`) and with
    the default geometry the whole network is ONE launch of the fused MFMA kernel (csrc/field.hip, `sdn_field_forward_f16`), so the
    reference-shaped loop of `run_cuda` (dnerf/renderer.py:350-376; SealDNeRF/renderer.py:250-276 with the mapper hooks) costs
    march + one field launch + composite per iteration.  Same numbers as the native loops' field (`FusedField`), fp16 distance
    from the op-by-op path; `model.fused_inference = False` switches it off.  Without autocast (no `-O`) the same dispatch goes to
    the fp32 fused kernel (csrc/field_f32.hip, 1e-4 from the op-by-op fp32 network) when `model.fused_inference_f32 = True`.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from freqencoder import FreqEncoder
from gridencoder import GridEncoder
from shencoder import SHEncoder

from .renderer import NeRFRenderer


class _TruncExp(torch.autograd.Function):
    """activation.py:5-17 of the reference: exp forward in fp32, backward clamps the argument to [-15, 15]."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def _mlp(dims):
    """bias-free Linear stack, `dims` = [in, hidden..., out] (dnerf/network.py:38-52,61-75,82-96)."""
    return nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=False) for i in range(len(dims) - 1)])


def _run_mlp(layers, h):
    last = len(layers) - 1
    for i, layer in enumerate(layers):
        h = layer(h)
        if i != last:
            h = F.relu(h, inplace=True)
    return h


class NeRFNetwork(NeRFRenderer):
    def __init__(self, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64,
                 num_layers_bg=2, hidden_dim_bg=64, num_layers_deform=8, hidden_dim_deform=128, bound=1, **kwargs):
        super().__init__(bound, **kwargs)
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.num_layers_deform, self.hidden_dim_deform = num_layers_deform, hidden_dim_deform

        # deformation field: freq(xyz, 10) ++ freq(t, 6) -> 8 x 128 -> 3          (network.py:31-52)
        self.encoder_deform = FreqEncoder(input_dim=3, degree=10)
        self.encoder_time = FreqEncoder(input_dim=1, degree=6)
        self.in_dim_deform, self.in_dim_time = self.encoder_deform.output_dim, self.encoder_time.output_dim
        self.deform_net = _mlp([self.in_dim_deform + self.in_dim_time] + [hidden_dim_deform] * (num_layers_deform - 1) + [3])

        # density: tiled grid (16 levels x 2, 16 -> 2048*bound) -> 64 -> 1 + 15      (network.py:55-75)
        self.encoder = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                                   desired_resolution=2048 * bound, gridtype="tiled", align_corners=False)
        self.in_dim = self.encoder.output_dim
        self.sigma_net = _mlp([self.in_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim])

        # colour: SH(dir, 4) ++ geo_feat -> 64 -> 64 -> 3                             (network.py:78-96)
        self.encoder_dir = SHEncoder(input_dim=3, degree=4)
        self.in_dim_dir = self.encoder_dir.output_dim
        self.color_net = _mlp([self.in_dim_dir + geo_feat_dim] + [hidden_dim_color] * (num_layers_color - 1) + [3])

        # background sphere (bg_radius > 0): 2-D hash grid over (theta, phi) ++ SH(dir) -> 64 -> 3        (network.py:99-121)
        if self.bg_radius > 0:
            self.num_layers_bg, self.hidden_dim_bg = num_layers_bg, hidden_dim_bg
            self.encoder_bg = GridEncoder(input_dim=2, num_levels=4, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                                          desired_resolution=2048, gridtype="hash", align_corners=False)
            self.in_dim_bg = self.encoder_bg.output_dim
            self.bg_net = _mlp([self.in_dim_bg + self.in_dim_dir] + [hidden_dim_bg] * (num_layers_bg - 1) + [3])
        else:
            self.bg_net = None

    # ------------------------------------------------------------------------------------------
    def _deform(self, x, t):
        enc_x = self.encoder_deform(x, bound=self.bound)
        enc_t = self.encoder_time(t)
        if enc_t.shape[0] == 1:
            enc_t = enc_t.repeat(x.shape[0], 1)
        return _run_mlp(self.deform_net, torch.cat([enc_x, enc_t], dim=1))

    def _sigma(self, x):
        h = _run_mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        return trunc_exp(h[..., 0]), h[..., 1:]

    def _color(self, d, geo_feat):
        h = torch.cat([self.encoder_dir(d), geo_feat], dim=-1)
        return torch.sigmoid(_run_mlp(self.color_net, h))

    # -- fused inference dispatch -----------------------------------------------------------------
    fused_inference = True       # class default; set False on a model to keep `forward` on the op-by-op path in every mode
    fused_inference_f32 = False  # opt-in: without autocast (the reference without -O) dispatch to the fp32 fused kernel as well

    def _fused_inference_ok(self, x, d):
        # (cheapest tests first: this runs on every forward call, and the reference-shaped render loop is bound by host time)
        if not self.fused_inference or self.training or torch.is_grad_enabled() or not x.is_cuda or x.dim() != 2 or x.shape[0] == 0:
            return False
        if x.dtype != torch.float32 or d.dtype != torch.float32:
            return False
        if not torch.is_autocast_enabled("cuda"):
            if not self.fused_inference_f32:
                return False   # the default without -O stays the op-by-op network: hipBLASLt fp32 GEMMs, what the fp32 fixtures pin
            mode = 32          # opt-in: the fp32 fused kernel (csrc/field_f32.hip), 1e-4 from the op-by-op network, 3 x faster
        elif torch.get_autocast_dtype("cuda") == torch.float16:
            mode = 16          # -O: the fp16 fused kernel with autocast's roundings
        else:
            return False
        ok = self.__dict__.get("_fused_arch_ok")
        if ok is None:         # the architecture does not change after construction
            from . import fused
            enc = self.encoder
            ok = bool(fused.available() and len(self.deform_net) == 8 and self.hidden_dim_deform == 128 and self.hidden_dim == 64
                      and self.num_layers == 2 and self.geo_feat_dim == 15 and self.num_layers_color == 3 and self.hidden_dim_color == 64
                      and enc.gridtype == "tiled" and not enc.align_corners and enc.interpolation == "linear" and enc.num_levels == 16
                      and enc.level_dim == 2)
            self.__dict__["_fused_arch_ok"] = ok
        if not ok:
            return False
        if mode == 32:
            ok32 = self.__dict__.get("_fused_f32_ok")
            if ok32 is None:
                from . import fused_f32
                ok32 = bool(fused_f32.available())
                self.__dict__["_fused_f32_ok"] = ok32
            if not ok32 or self.encoder.embeddings.dtype != torch.float32 or self.deform_net[0].weight.dtype != torch.float32:
                return False
        return mode

    # the same under -O: off by default -- that loop is bound by host time per call, and the list's two allocations + fill cost what the
    # skipped slots save (800x800: 2.28 ms per frame without, 2.33 with; fp32: 6.69 -> 5.26; profiles/r04_reference_shaped_live_lists.json)
    fused_live_lists_f16 = os.environ.get("SDN_LIVE_LISTS_F16", "0") == "1"
    fused_live_lists = True      # evaluate only the slots the marcher filled when `x` is the marcher's own output tensor (below)

    def _live_of(self, x):
        """(slot list, count, version) the drop-in `raymarching.march_rays` hung on ITS output tensor, or None.  The reference's caller
        (dnerf/renderer.py:350-376) pads every ray to n_step slots and evaluates the padding too; composite_rays stops at a ray's first
        empty slot, so what the network returns there is never read.  With the list the fused kernels skip those slots (their outputs
        are zeros).  A tensor that is not the marcher's (or was written since: version counter) has no list and is evaluated whole."""
        if not self.fused_live_lists:
            return None
        live = getattr(x, "_sdn_live", None)
        if live is None:
            import raymarching
            ll = raymarching.live_lists
            if not ll["pinned"]:
                ll["on"] = True      # from the next march_rays call on
            return None
        if live[2] != x._version or not x.is_contiguous() or live[0].shape[0] > x.shape[0]:
            return None
        return live

    def _parameter_epoch(self):
        ps = self.__dict__.get("_fused_params")
        if ps is None:         # the Parameter OBJECTS (load_state_dict, .to(), optimizer steps keep them; their address / version move)
            ps = tuple([self.encoder.embeddings] + [l.weight for l in self.deform_net] + [l.weight for l in self.sigma_net]
                       + [l.weight for l in self.color_net])
            self.__dict__["_fused_params"] = ps
        return tuple([(p.data_ptr(), p._version) for p in ps])

    def _forward_fused32(self, x, d, t):
        """The fp32 network in one launch: sigma [M], rgb [M,3], deform [M,3] (zeros on the canonical frame), all float32, within 1e-4 of
        the op-by-op evaluation (tests/test_gpu_field_f32.py)."""
        epoch = self._parameter_epoch()
        cache = self.__dict__.get("_fused_cache32")
        if cache is None or cache[0] != epoch[1:] or cache[2][0] != epoch[0][0]:
            from . import fused_f32
            # (weights packed once per parameter version; the embedding table is read where it is, so its in-place updates need nothing)
            field = fused_f32.FusedFieldF32(self, t)
            field.density_scale = 1.0
            cache = (epoch[1:], field, epoch[0])
            self.__dict__["_fused_cache32"] = cache
        field = cache[1]
        seen = self.__dict__.get("_fused_time")
        if not isinstance(t, torch.Tensor) or seen is None or seen[0]() is not t or seen[1] != t._version:
            value = field.time_value(t)
            if isinstance(t, torch.Tensor):
                import weakref
                self.__dict__["_fused_time"] = (weakref.ref(t), t._version, value)
        else:
            value = seen[2]
        if field.__dict__.get("_time_set") != value:
            field.set_time(value)
            field._time_set = value
        live = self._live_of(x)
        M = x.shape[0]
        if live is not None:
            flat = torch.zeros(7 * M, dtype=torch.float32, device=x.device)      # one fill for the three outputs' skipped slots
            field._buf = (flat[:M], flat[M:4 * M].view(M, 3))
            deform = flat[4 * M:].view(M, 3)
            sig, rgb = field(x, d.contiguous(), live[0], live[1], deform=deform)
            field._buf = None
            return sig, rgb, deform
        field._buf = None
        deform = torch.empty(M, 3, dtype=torch.float32, device=x.device)
        sig, rgb = field(x.contiguous(), d.contiguous(), deform=deform)
        return sig, rgb, deform

    def _forward_fused(self, x, d, t):
        """sigma [M] f32 (trunc_exp, density_scale NOT applied: the caller multiplies, dnerf/renderer.py:368), rgb [M,3] (the fp16 values
        of torch.sigmoid on the half logits, held in f32), deform = None (callers of the inference branch discard it)."""
        epoch = self._parameter_epoch()
        cache = self.__dict__.get("_fused_cache")
        if cache is None or cache[0] != epoch[1:]:
            from . import fused
            # (the fp16 table cast of grid.py:43-44 and the packed weights are made once per parameter version, not once per call)
            field = fused.FusedField(self, t, fp16=True)
            field.density_scale = 1.0
            cache = (epoch[1:], field, epoch[0])
            self.__dict__["_fused_cache"] = cache
        elif cache[2] != epoch[0]:
            cache[1].load_table(self.encoder.embeddings.detach())
            cache = (cache[0], cache[1], epoch[0])
            self.__dict__["_fused_cache"] = cache
        field = cache[1]
        # by VALUE: one host read of t per distinct tensor content, where the reference's `if t == 0` reads it on every call
        # (network.py:140).  A loop hands the same tensor to every iteration: its value is kept while the tensor's address and version
        # counter stand (torch-level writes bump the counter; a write behind torch's back -- raw pointer, DLPack -- needs a fresh tensor).
        # (the tensor OBJECT, not its address: a new tensor the allocator placed at a recycled address is another object)
        seen = self.__dict__.get("_fused_time")
        if not isinstance(t, torch.Tensor) or seen is None or seen[0]() is not t or seen[1] != t._version:
            value = field.time_value(t)
            if isinstance(t, torch.Tensor):
                import weakref
                self.__dict__["_fused_time"] = (weakref.ref(t), t._version, value)
        else:
            value = seen[2]
        if cache[1].__dict__.get("_time_set") != value:
            field.set_time(value)
            field._time_set = value
        # (Evaluating only the slots that hold a sample from a list built HERE -- a kernel over the zero direction vectors of the empty
        #  slots -- was measured and dropped, 2.73 -> 3.08 ms per frame; the list the marcher builds as it goes costs no pass.)
        live = self._live_of(x) if self.fused_live_lists_f16 else None
        if live is not None:
            M = x.shape[0]
            flat = torch.zeros(4 * M, dtype=torch.float32, device=x.device)
            field._buf = (flat[:M], flat[M:].view(M, 3))
            sig, rgb = field(x, d.contiguous(), live[0], live[1])
            field._buf = None
            return sig, rgb, None
        field._buf = None          # fresh output tensors per call (caching allocator, no launch): the caller owns them, as on the op-by-op path
        sig, rgb = field(x.contiguous(), d.contiguous())
        return sig, rgb, None

    def forward(self, x, d, t):
        """x [M,3] in [-bound,bound], d [M,3] unit, t [1,1] -> sigma [M], rgb [M,3], deform [M,3]  (network.py:123-169)."""
        mode = self._fused_inference_ok(x, d)
        if mode:
            return self._forward_fused(x, d, t) if mode == 16 else self._forward_fused32(x, d, t)
        deform = self._deform(x, t)
        if t == 0:  # canonical frame: no deformation (device compare => host sync, as in the reference :140)
            deform = torch.zeros_like(x)
        # (explicit promotion: torch's mixed fp32 + fp16 add kernel takes 73 us for these 27 000 elements on gfx950, the cast + add 9 us)
        sigma, geo_feat = self._sigma(x + deform.to(x.dtype))
        return sigma, self._color(d, geo_feat), deform

    def density(self, x, t):
        """network.py:171-206."""
        deform = self._deform(x, t)
        if t != 0:
            x = x + deform.to(x.dtype)
        sigma, geo_feat = self._sigma(x)
        return {"deform": deform, "sigma": sigma, "geo_feat": geo_feat}

    def background(self, x, d):
        """x [N,2] in [-1,1] (theta, phi on the background sphere: raymarching.sph_from_ray), d [N,3] -> rgb [N,3]   (network.py:208-223)."""
        h = torch.cat([self.encoder_dir(d), self.encoder_bg(x)], dim=-1)
        return torch.sigmoid(_run_mlp(self.bg_net, h))

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """network.py:225-257 (masked colour query used by the non-cuda-ray sampler)."""
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            d, geo_feat = d[mask], geo_feat[mask]
        h = self._color(d, geo_feat)
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
            return rgbs
        return h

    def get_params(self, lr, lr_net):
        """network.py:260-275."""
        return [
            {"params": self.encoder.parameters(), "lr": lr},
            {"params": self.sigma_net.parameters(), "lr": lr_net},
            {"params": self.encoder_dir.parameters(), "lr": lr},
            {"params": self.color_net.parameters(), "lr": lr_net},
            {"params": self.encoder_deform.parameters(), "lr": lr},
            {"params": self.encoder_time.parameters(), "lr": lr},
            {"params": self.deform_net.parameters(), "lr": lr_net},
        ] + ([{"params": self.encoder_bg.parameters(), "lr": lr}, {"params": self.bg_net.parameters(), "lr": lr_net}] if self.bg_radius > 0 else [])
