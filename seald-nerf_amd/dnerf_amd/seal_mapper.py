"""SealD-NeRF bounding-box mapper on the device (scope row "next" #1): the object the teacher / student renderers hook
between the marcher and the field network (`map_to_origin`) and after it (`map_color`).

Mirrors, with the same names and `map_data` keys, the pieces of the reference that sit inside the render loop:
  * `SealMapper.map_mask` / `map_color` / `map_data_conversion`      SealNeRF/seal_utils.py:40-153
  * `SealBBoxMapper.__init__` / `map_to_origin`                      SealNeRF/seal_utils.py:156-286
  * `moller_trumbore`, `points_in_mesh`                              SealNeRF/seal_utils.py:638-693
  * `modify_hsv`, `modify_rgb`                                       SealNeRF/seal_utils.py:747-777
  * `rgb2hsv_torch`, `hsv2rgb_torch` ([N,3] form)                    SealNeRF/color_utils.py:31-63
without pytorch3d / trimesh / open3d (none is installed here): the "from" box is the oriented box of the config's `raw`
points, the "to" box its scaled + transformed image, both as 8 vertices / 12 triangles.

One documented difference: trimesh's `bounding_box_oriented` searches a minimum-volume box over the convex hull; here `raw` is
expected to be (what the Seal GUI writes) the 8 corners of a cuboid, which are recognised exactly; any other point set gets its
PCA-aligned box.  Everything is plain torch on whatever device the points live on -- it runs on the sample stream of the HIP
operators; colour conversions are pinned by vectors generated from the reference's pure-torch `color_utils` (tests/golden).
"""
import itertools

import numpy as np
import torch

_TEST_DIR = (0.4395064455, 0.617598629942, 0.652231566745)   # trimesh's "magic" ray direction, seal_utils.py:684-686


# ----------------------------------------------------------------------------------------------------------------------
# colour (color_utils.py:31-63 on [N,3] tensors; seal_utils.py:747-777)
# ----------------------------------------------------------------------------------------------------------------------
def rgb2hsv(rgb):
    cmax, cmax_idx = torch.max(rgb, dim=1, keepdim=True)
    cmin = torch.min(rgb, dim=1, keepdim=True)[0]
    delta = cmax - cmin
    r, g, b = rgb[:, 0:1], rgb[:, 1:2], rgb[:, 2:3]
    safe = torch.where(delta == 0, torch.ones_like(delta), delta)
    h = torch.where(cmax_idx == 0, ((g - b) / safe) % 6, torch.where(cmax_idx == 1, (b - r) / safe + 2, (r - g) / safe + 4))
    h = torch.where(delta == 0, torch.zeros_like(h), h) / 6.0
    s = torch.where(cmax == 0, torch.zeros_like(cmax), delta / torch.where(cmax == 0, torch.ones_like(cmax), cmax))
    return torch.cat([h, s, cmax], dim=1)


def hsv2rgb(hsv):
    h, s, v = hsv[:, 0:1], hsv[:, 1:2], hsv[:, 2:3]
    c = v * s
    x = c * (-torch.abs(h * 6.0 % 2.0 - 1) + 1.0)
    m = v - c
    o = torch.zeros_like(c)
    idx = (h * 6.0).to(torch.uint8) % 6            # the reference's `.type(torch.uint8)` truncation, then % 6
    table = [(c, x, o), (x, c, o), (o, c, x), (o, x, c), (x, o, c), (c, o, x)]
    rgb = torch.zeros_like(hsv)
    for k, (rr, gg, bb) in enumerate(table):
        rgb = torch.where(idx == k, torch.cat([rr, gg, bb], dim=1), rgb)
    return rgb + m


def modify_hsv(rgb, modification):
    if rgb.shape[0] == 0:
        return rgb
    hsv = rgb2hsv(rgb)
    mod = torch.as_tensor(modification, device=rgb.device, dtype=rgb.dtype).view(1, 3)
    return hsv2rgb(hsv + mod)


def modify_rgb(rgb, modification, light_offset=0.0):
    if rgb.shape[0] == 0:
        return rgb
    hsl = rgb2hsv(rgb)
    mod = rgb2hsv(torch.as_tensor(modification, device=rgb.device, dtype=rgb.dtype).view(-1, 3))
    raw_l = hsl[:, 2:3]
    raw_l_offset = raw_l - raw_l.mean()
    out = torch.cat([mod[:, :2].expand(rgb.shape[0], 2), (mod[:, 2:3] + raw_l_offset + light_offset).clamp(0, 1)], dim=1)
    return hsv2rgb(out)


# ----------------------------------------------------------------------------------------------------------------------
# geometry (seal_utils.py:638-693)
# ----------------------------------------------------------------------------------------------------------------------
def moller_trumbore(ray_o, ray_d, tris, eps=1e-8):
    """[m,3] rays x [n,3,3] triangles -> bool [m]: does the ray (t >= 0) hit any triangle."""
    e1 = tris[:, 1] - tris[:, 0]
    e2 = tris[:, 2] - tris[:, 0]
    n = torch.cross(e1, e2, dim=-1)
    invdet = 1.0 / -(torch.einsum("md,nd->mn", ray_d, n) + eps)
    a0 = ray_o[:, None] - tris[None, :, 0]
    da0 = torch.cross(a0, ray_d[:, None].expand(*a0.shape), dim=-1)
    u = torch.einsum("mnd,nd->mn", da0, e2) * invdet
    v = -torch.einsum("mnd,nd->mn", da0, e1) * invdet
    t = torch.einsum("mnd,nd->mn", a0, n) * invdet
    return ((t >= 0.0) & (u >= 0.0) & (v >= 0.0) & ((u + v) <= 1.0)).any(1)


def points_in_mesh(points, triangles, rays_d=None):
    """A point is inside iff both the ray along the test direction and the opposite ray hit the mesh."""
    if rays_d is None:
        rays_d = torch.tensor([_TEST_DIR], device=points.device, dtype=points.dtype)
    d = rays_d.to(points.dtype).repeat(points.shape[0], 1)
    mask = moller_trumbore(torch.cat([points, points]), torch.cat([d, -d]), triangles.to(points.dtype))
    return mask[:points.shape[0]] & mask[-points.shape[0]:]


_BOX_FACES = np.array([[0, 1, 3], [0, 3, 2], [4, 7, 5], [4, 6, 7], [0, 5, 1], [0, 4, 5], [2, 3, 7], [2, 7, 6],
                       [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], dtype=np.int64)   # vertex k = origin + (k&1) e1 + (k>>1&1) e2 + (k>>2) e3


def oriented_box(points):
    """-> (vertices [8,3] float64 in the corner order of _BOX_FACES, centre [3]).  8 cuboid corners are recognised exactly
    (any order); other point sets get their PCA-aligned bounding box."""
    p = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    if p.shape[0] == 8:
        diam = np.linalg.norm(p - p.mean(0), axis=1).max() * 2
        tol = 1e-6 * max(diam, 1e-12)
        rest = list(range(1, 8))
        for trio in itertools.combinations(rest, 3):
            e = p[list(trio)] - p[0]
            if max(abs(e[0] @ e[1]), abs(e[0] @ e[2]), abs(e[1] @ e[2])) > 1e-6 * diam * diam:
                continue
            verts = np.stack([p[0] + (k & 1) * e[0] + ((k >> 1) & 1) * e[1] + (k >> 2) * e[2] for k in range(8)])
            d = np.linalg.norm(verts[:, None] - p[None], axis=-1)
            if (d.min(1) < tol).all() and (d.min(0) < tol).all():
                return verts, verts.mean(0)
    c = p.mean(0)
    _, _, vt = np.linalg.svd(p - c, full_matrices=False)
    q = (p - c) @ vt.T
    lo, hi = q.min(0), q.max(0)
    verts = np.stack([c + (np.array([(hi if (k >> a) & 1 else lo)[a] for a in range(3)])) @ vt for k in range(8)])
    return verts, verts.mean(0)


def _bounds(verts):
    return np.stack([verts.min(0), verts.max(0)])


class SealMapper:
    """seal_utils.py:18-153 (the parts used inside the render loop)."""

    def __init__(self, seal_config):
        self.config = seal_config
        self.device = "cpu"
        self.dtype = torch.float32
        self.map_data = {}
        self.map_triangles = None
        self.map_test_dir = None

    def map_data_conversion(self, T=None, force=False):
        if T is None and not force:
            return
        if T is not None and (self.device != T.device or self.dtype != T.dtype):
            self.device, self.dtype = T.device, T.dtype
        elif not force:
            return
        for k, v in self.map_data.items():
            self.map_data[k] = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).to(self.device, self.dtype)
        if self.map_triangles is not None:
            self.map_triangles = self.map_triangles.to(self.device, self.dtype)

    def map_mask(self, points):
        bounds = self.map_data["map_bound"]
        if bounds.ndim == 2:
            bounds = bounds[None]
        bound_mask = None
        for i in range(bounds.shape[0]):
            cur = torch.logical_and(points.all(1), torch.logical_and(bounds[i][1] > points, points > bounds[i][0]).all(1))
            bound_mask = cur if bound_mask is None else torch.logical_or(bound_mask, cur)
        if not bound_mask.any():
            return bound_mask
        shape_mask = points_in_mesh(points[bound_mask], self.map_triangles, self.map_test_dir)
        bound_mask[bound_mask.clone()] = shape_mask
        return bound_mask

    def map_color(self, points, dirs, colors):
        if "hsv" in self.map_data:
            colors = modify_hsv(colors, self.map_data["hsv"])
        if "rgb" in self.map_data:
            colors = modify_rgb(colors, self.map_data["rgb"], float(self.map_data["rgb_light_offset"]))
        return colors

    def map_to_origin(self, points, dirs=None):
        raise NotImplementedError()


class SealBBoxMapper(SealMapper):
    """seal_utils.py:156-286.  seal_config: {type: 'bbox', raw: [N,3], transform: [4,4], scale: [3], boundType: 'from' | 'to' |
    'both', hsv / rgb / rgbLightOffset / mapSource optional}."""

    def __init__(self, seal_config, config_path=None):
        super().__init__(seal_config)
        T = np.array(seal_config["transform"], dtype=np.float64)
        R = T[:3, :3]
        scale = np.array(seal_config["scale"], dtype=np.float64)
        from_verts, from_center = oriented_box(seal_config["raw"])
        to_verts = (from_verts - from_center) * scale + from_center
        to_verts = to_verts @ R.T + T[:3, 3]
        to_center = to_verts.mean(0)
        self.from_vertices, self.to_vertices = from_verts, to_verts
        bound_type = seal_config.get("boundType", "to")
        fill_bounds = np.stack([_bounds(to_verts), _bounds(from_verts)])       # [2, 2, 3] like Meshes.get_bounding_boxes().transpose(1, 2)
        if bound_type == "to":
            bounds, tri = _bounds(to_verts), to_verts[_BOX_FACES]
        elif bound_type == "from":
            bounds, tri = _bounds(from_verts), from_verts[_BOX_FACES]
        elif bound_type == "both":
            bounds, tri = fill_bounds, np.concatenate([to_verts[_BOX_FACES], from_verts[_BOX_FACES]])
        else:
            raise ValueError(f"boundType {bound_type!r}")
        self.map_triangles = torch.from_numpy(tri)
        self.map_data = {
            "force_fill_bound": fill_bounds,
            "map_bound": bounds,
            "pose_center": (from_center + to_center) / 2,
            "pose_radius": np.linalg.norm(from_center - to_center, 2) * 10,
            "transform": np.linalg.inv(T),
            "rotation": np.linalg.inv(R),
            "scale": 1 / scale,
            "center": from_center,
        }
        if "hsv" in seal_config:
            self.map_data["hsv"] = seal_config["hsv"]
        if "rgb" in seal_config:
            self.map_data["rgb"] = seal_config["rgb"]
            self.map_data["rgb_light_offset"] = seal_config.get("rgbLightOffset", 0)
        if seal_config.get("mapSource"):
            self.map_data["empty_bound"] = _bounds(from_verts)
            self.map_data["map_source"] = seal_config["mapSource"]
        self.map_data_conversion(force=True)

    # ---- device fast path: csrc/seal.hip, one lane per sample slot, in place ---------------------------------------------
    def _native_ok(self, points, dirs):
        return (points.is_cuda and dirs is not None and points.dtype == torch.float32 and dirs.dtype == torch.float32
                and points.is_contiguous() and dirs.is_contiguous())

    def _native_args(self, device):
        import ctypes
        key = str(device)
        if getattr(self, "_native_key", None) != key:
            f32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64).astype(np.float32))  # noqa: E731
            cfl = lambda a: (ctypes.c_float * a.size)(*a.reshape(-1).tolist())                          # noqa: E731
            md = {k: (v.detach().cpu().double().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, dtype=np.float64))
                  for k, v in self.map_data.items()}
            tri = self.map_triangles.detach().cpu().double().numpy()
            e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
            tris12 = np.concatenate([tri[:, 0], e1, e2, np.cross(e1, e2)], axis=1)
            bounds = md["map_bound"].reshape(-1, 2, 3)
            self._native = dict(tris=torch.from_numpy(f32(tris12)).to(device), n_tris=int(tris12.shape[0]),
                                bounds=cfl(f32(bounds.reshape(-1, 6))), n_bounds=int(bounds.shape[0]),
                                test_dir=cfl(f32(_TEST_DIR if self.map_test_dir is None else self.map_test_dir.cpu().numpy())),
                                tinv=cfl(f32(md["transform"][:3, :4])), rinv=cfl(f32(md["rotation"])), scale=cfl(f32(md["scale"])),
                                center=cfl(f32(md["center"])),
                                # [0..15] modify_rgb's sum / count, [16..19] the mapSource flag word (only ever raised), zeroed once
                                scratch=torch.zeros(32, dtype=torch.uint8, device=device))
            if "map_source" in md:
                self._native["source_bound"] = cfl(f32(md["empty_bound"].reshape(2, 3)))
                self._native["map_source"] = cfl(f32(md["map_source"].reshape(3)))
            if "rgb" in md:
                self._native["rgb"] = [float(v) for v in f32(md["rgb"].reshape(3))]
                self._native["rgb_light_offset"] = float(np.float32(md["rgb_light_offset"]))
            self._native_key = key
        return self._native

    @torch.no_grad()
    def map_to_origin_(self, points, dirs):
        """In-place `map_to_origin` on the sample buffers (CUDA fp32 contiguous) -> bool mask [M]; the HIP kernel of csrc/seal.hip."""
        from sdn_backend import lib, check, ptr, stream
        a = self._native_args(points.device)
        M = points.shape[0]
        mask = torch.empty(M, dtype=torch.uint8, device=points.device)
        if "map_source" in a:
            # (`mapSource`, seal_utils.py:269-273: the call's unmapped samples inside the source box move to one point -- if the call maps
            #  any sample at all, :251-252; every slot of the buffers is a sample of the call here)
            check(lib.sdn_seal_bbox_map_source(ptr(points), ptr(dirs), M, a["bounds"], a["n_bounds"], ptr(a["tris"]), a["n_tris"], a["test_dir"],
                                               a["tinv"], a["rinv"], a["scale"], a["center"], a["source_bound"], a["map_source"],
                                               a["scratch"].data_ptr() + 16, ptr(mask), None, None, None, stream()), "seal_bbox_map_source")
        else:
            check(lib.sdn_seal_bbox_map(ptr(points), ptr(dirs), M, a["bounds"], a["n_bounds"], ptr(a["tris"]), a["n_tris"], a["test_dir"], a["tinv"],
                                        a["rinv"], a["scale"], a["center"], ptr(mask), stream()), "seal_bbox_map")
        return mask.view(torch.bool)

    @torch.no_grad()
    def map_color_(self, rgbs, mask):
        """In-place `map_color` of the masked samples (seal_utils.py:48-57: the hsv modification, then the rgb tint): HIP kernels on CUDA
        fp32 buffers, the torch restatement otherwise (and for an `image` modification, which the bbox mapper of this build does not carry)."""
        if rgbs.is_cuda and rgbs.dtype == torch.float32 and rgbs.is_contiguous() and "image" not in self.map_data:
            from sdn_backend import lib, check, ptr, stream
            m8 = mask.view(torch.uint8)
            if "hsv" in self.map_data:
                h = [float(v) for v in self.map_data["hsv"].reshape(-1).tolist()]
                check(lib.sdn_seal_modify_hsv(ptr(rgbs), ptr(m8), rgbs.shape[0], h[0], h[1], h[2], stream()), "seal_modify_hsv")
            if "rgb" in self.map_data:
                a = self._native_args(rgbs.device)
                c = a["rgb"]
                check(lib.sdn_seal_modify_rgb(ptr(rgbs), ptr(m8), rgbs.shape[0], c[0], c[1], c[2], a["rgb_light_offset"], ptr(a["scratch"]),
                                              None, None, None, stream()), "seal_modify_rgb")
        elif bool(mask.any()):
            rgbs[mask] = self.map_color(None, None, rgbs[mask]).to(rgbs.dtype)
        return rgbs

    @torch.no_grad()
    def map_to_origin(self, points, dirs=None):
        """Samples inside the target box are taken back to where their content comes from: inverse transform, inverse scale
        about the source centre, directions by the inverse rotation.  -> (points', dirs', mask).  CUDA fp32 inputs take the
        HIP kernel (on copies, as the reference returns copies); everything else the torch restatement below."""
        if self._native_ok(points, dirs):
            self.map_data_conversion(points)
            p, d = points.clone(), dirs.clone()
            return p, d, self.map_to_origin_(p, d)
        return self._map_to_origin_torch(points, dirs)

    @torch.no_grad()
    def _map_to_origin_torch(self, points, dirs=None):
        with torch.autocast(points.device.type if points.device.type != "cpu" else "cpu", enabled=False):
            self.map_data_conversion(points)
            has_dirs = dirs is not None
            mask = self.map_mask(points)
            if not mask.any():
                return points, dirs, mask
            inner = points[mask]
            n = inner.shape[0]
            hom = torch.vstack([inner.T, torch.ones([1, n], device=inner.device, dtype=inner.dtype)])
            moved = torch.matmul(self.map_data["transform"], hom).T[:, :3]
            origin = (moved - self.map_data["center"]) * self.map_data["scale"] + self.map_data["center"]
            points_copy = points.clone()
            dirs_copy = dirs.clone() if has_dirs else None
            if "map_source" in self.map_data:
                sb = self.map_data["empty_bound"]
                source_mask = torch.logical_and(sb[1] > points, points > sb[0]).all(1)
                points_copy[source_mask] = self.map_data["map_source"]
            points_copy[mask] = origin
            if has_dirs:
                dirs_copy[mask] = torch.matmul(self.map_data["rotation"], dirs[mask].T).T
            return points_copy, dirs_copy, mask


def get_seal_mapper(seal_config, config_path=None):
    """seal_utils.py:581-592 for the mapper type built here."""
    if seal_config.get("type") == "bbox":
        return SealBBoxMapper(seal_config, config_path)
    raise NotImplementedError(f"seal mapper type {seal_config.get('type')!r} (brush / anchor mappers need trimesh + pytorch3d mesh fitting)")


@torch.no_grad()
def fill_bitfield(density_bitfield, bounds, grid_size=128, bound=1.0):
    """Marks every occupancy cell (cascade 0) whose centre lies in one of the axis-aligned `bounds` [B,2,3] as occupied, in every
    time slice of `density_bitfield` [T, grid_size^3 / 8] (uint8, Morton-ordered bits, bit i%8 of byte i/8) -- what the reference's
    trainer does with `force_fill_bound` before rendering an edit (`hack_bitfield`): the marcher must sample inside the box the
    content is moved INTO, where the unedited scene is empty.  In place; returns the number of cells marked."""
    import raymarching
    dev = density_bitfield.device
    b = torch.as_tensor(bounds, dtype=torch.float32, device=dev).reshape(-1, 2, 3)
    H = int(grid_size)
    c = (torch.arange(H, dtype=torch.float32, device=dev) + 0.5) * (2.0 * bound / H) - bound
    inside = torch.zeros(H, H, H, dtype=torch.bool, device=dev)
    for lo, hi in b:
        mx, my, mz = (c > lo[0]) & (c < hi[0]), (c > lo[1]) & (c < hi[1]), (c > lo[2]) & (c < hi[2])
        inside |= mx[:, None, None] & my[None, :, None] & mz[None, None, :]
    coords = torch.nonzero(inside).to(torch.int32).contiguous()
    if coords.shape[0] == 0:
        return 0
    idx = raymarching.morton3D(coords).long()
    bits = torch.zeros(H * H * H, dtype=torch.uint8, device=dev)
    bits[idx] = 1
    packed = (bits.view(-1, 8) << torch.arange(8, dtype=torch.uint8, device=dev)).sum(1).to(torch.uint8)
    density_bitfield[:, : packed.shape[0]] |= packed[None]
    return int(coords.shape[0])
