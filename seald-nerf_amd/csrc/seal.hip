// SealD-NeRF bounding-box seal mapper on the sample stream (scope row "next" #1) for gfx950.
//
// Behavioural contract: SealNeRF/seal_utils.py of the reference --
//   map_mask            :132-153  (points.all(1) & strict AABB test of each bound, then points_in_mesh)
//   points_in_mesh      :675-693  (inside iff the ray along trimesh's test direction AND the opposite ray hit the mesh)
//   moller_trumbore     :638-672  (t, u, v exactly as written there, eps = 1e-8, t >= 0, u >= 0, v >= 0, u + v <= 1)
//   SealBBoxMapper.map_to_origin :245-286 (inverse transform, inverse scale about the source centre, inverse rotation of dirs)
//   modify_hsv :747-758 with color_utils.py:31-63 (rgb -> hsv, + modification, -> rgb)
// The reference evaluates this with boolean-mask gathers / scatters and O(points x triangles) temporaries in torch, inside the
// render loop; here it is one lane per sample slot, in place, between the marcher and the field kernel.  Dot products are
// accumulated x, y, z in fp32 (torch's einsum order is library-defined): masks agree with the torch restatement except for
// points within rounding of a face, mapped coordinates to ~1e-6 -- the tolerances its tests state.
#include "sdn_common.h"

namespace {

struct SealBoxArgs {
    float bounds[4][6];     // up to 4 AABBs {lo xyz, hi xyz}
    uint32_t n_bounds;
    const float *tris;      // [F][12]: v0, E1, E2, N (host-precomputed from the box triangles)
    uint32_t n_tris;
    float test_dir[3];
    float tinv[12];         // inverse transform, rows of [3 x 4]
    float rinv[9];          // inverse rotation
    float scale[3], center[3];
};

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) { return ax * bx + ay * by + az * bz; }

__device__ __forceinline__ bool any_hit(const float *__restrict__ tris, uint32_t F, float ox, float oy, float oz, float dx, float dy, float dz) {
    bool hit = false;
    for (uint32_t f = 0; f < F; f++) {
        const float *t = tris + 12 * f;
        const float a0x = ox - t[0], a0y = oy - t[1], a0z = oz - t[2];
        const float invdet = 1.0f / -(dot3(dx, dy, dz, t[9], t[10], t[11]) + 1e-8f);
        const float cx = a0y * dz - a0z * dy, cy = a0z * dx - a0x * dz, cz = a0x * dy - a0y * dx;   // A0 x d
        const float u = dot3(cx, cy, cz, t[6], t[7], t[8]) * invdet;
        const float v = -dot3(cx, cy, cz, t[3], t[4], t[5]) * invdet;
        const float tt = dot3(a0x, a0y, a0z, t[9], t[10], t[11]) * invdet;
        hit |= (tt >= 0.0f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= 1.0f);
    }
    return hit;
}

__global__ void __launch_bounds__(256) k_seal_bbox_map(float *__restrict__ xyzs, float *__restrict__ dirs, uint32_t M, SealBoxArgs A,
                                                       uint8_t *__restrict__ mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float x = xyzs[(size_t)i * 3], y = xyzs[(size_t)i * 3 + 1], z = xyzs[(size_t)i * 3 + 2];
    bool in = false;
    if (x != 0.0f && y != 0.0f && z != 0.0f) {   // `points.all(1)`: empty slots (and exact zeros) are never mapped
        for (uint32_t b = 0; b < A.n_bounds; b++)
            in |= (A.bounds[b][3] > x) & (x > A.bounds[b][0]) & (A.bounds[b][4] > y) & (y > A.bounds[b][1]) & (A.bounds[b][5] > z) & (z > A.bounds[b][2]);
    }
    if (in)
        in = any_hit(A.tris, A.n_tris, x, y, z, A.test_dir[0], A.test_dir[1], A.test_dir[2]) &&
             any_hit(A.tris, A.n_tris, x, y, z, -A.test_dir[0], -A.test_dir[1], -A.test_dir[2]);
    mask[i] = in ? 1 : 0;
    if (!in) return;
    float m[3];
    #pragma unroll
    for (int r = 0; r < 3; r++) {
        const float moved = A.tinv[4 * r] * x + A.tinv[4 * r + 1] * y + A.tinv[4 * r + 2] * z + A.tinv[4 * r + 3];
        m[r] = (moved - A.center[r]) * A.scale[r] + A.center[r];
    }
    const float dx = dirs[(size_t)i * 3], dy = dirs[(size_t)i * 3 + 1], dz = dirs[(size_t)i * 3 + 2];
    #pragma unroll
    for (int r = 0; r < 3; r++) {
        xyzs[(size_t)i * 3 + r] = m[r];
        dirs[(size_t)i * 3 + r] = A.rinv[3 * r] * dx + A.rinv[3 * r + 1] * dy + A.rinv[3 * r + 2] * dz;
    }
}

// color_utils.py:31-63 + seal_utils.py:747-758 on the masked samples, in place
__global__ void __launch_bounds__(256) k_seal_hsv(float *__restrict__ rgbs, const uint8_t *__restrict__ mask, uint32_t M, float mh, float ms, float mv) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M || !mask[i]) return;
    const float r = rgbs[(size_t)i * 3], g = rgbs[(size_t)i * 3 + 1], b = rgbs[(size_t)i * 3 + 2];
    const float cmax = fmaxf(r, fmaxf(g, b)), cmin = fminf(r, fminf(g, b));
    const float delta = cmax - cmin;
    float h;
    if (delta == 0.0f) h = 0.0f;
    else if (r >= g && r >= b) { h = (g - b) / delta; h = h - 6.0f * floorf(h / 6.0f); }   // torch `% 6` (result has the divisor's sign); first max wins ties
    else if (g >= b) h = (b - r) / delta + 2.0f;
    else h = (r - g) / delta + 4.0f;
    h = h / 6.0f + mh;
    const float s = (cmax == 0.0f ? 0.0f : delta / cmax) + ms;
    const float v = cmax + mv;
    const float c = v * s;
    const float h6 = h * 6.0f;
    const float xx = c * (-fabsf((h6 - 2.0f * floorf(h6 / 2.0f)) - 1.0f) + 1.0f);
    const float m = v - c;
    const uint32_t idx = ((uint32_t)(uint8_t)(int)h6) % 6u;   // `.type(torch.uint8)` truncation, then % 6
    float o0, o1, o2;
    switch (idx) {
        case 0: o0 = c; o1 = xx; o2 = 0; break;
        case 1: o0 = xx; o1 = c; o2 = 0; break;
        case 2: o0 = 0; o1 = c; o2 = xx; break;
        case 3: o0 = 0; o1 = xx; o2 = c; break;
        case 4: o0 = xx; o1 = 0; o2 = c; break;
        default: o0 = c; o1 = 0; o2 = xx; break;
    }
    rgbs[(size_t)i * 3] = o0 + m; rgbs[(size_t)i * 3 + 1] = o1 + m; rgbs[(size_t)i * 3 + 2] = o2 + m;
}

}  // namespace

extern "C" {

int sdn_seal_bbox_map(float *xyzs, float *dirs, uint32_t M, const float *bounds, uint32_t n_bounds, const float *tris, uint32_t n_tris,
                      const float *test_dir, const float *tinv, const float *rinv, const float *scale, const float *center, uint8_t *mask,
                      void *stream) {
    if (M == 0) return 0;
    if (!xyzs || !dirs || !bounds || !tris || !test_dir || !tinv || !rinv || !scale || !center || !mask) return SDN_E_BADARG;
    if (n_bounds == 0 || n_bounds > 4 || n_tris == 0) return SDN_E_UNSUPPORTED;
    SealBoxArgs a;
    for (uint32_t b = 0; b < n_bounds; b++)
        for (int k = 0; k < 6; k++) a.bounds[b][k] = bounds[6 * b + k];
    a.n_bounds = n_bounds; a.tris = tris; a.n_tris = n_tris;
    for (int k = 0; k < 3; k++) { a.test_dir[k] = test_dir[k]; a.scale[k] = scale[k]; a.center[k] = center[k]; }
    for (int k = 0; k < 12; k++) a.tinv[k] = tinv[k];
    for (int k = 0; k < 9; k++) a.rinv[k] = rinv[k];
    hipLaunchKernelGGL(k_seal_bbox_map, dim3(sdn_div_up(M, 256u)), dim3(256), 0, (hipStream_t)stream, xyzs, dirs, M, a, mask);
    return sdn_launch_status();
}

int sdn_seal_modify_hsv(float *rgbs, const uint8_t *mask, uint32_t M, float dh, float ds, float dv, void *stream) {
    if (M == 0) return 0;
    if (!rgbs || !mask) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_seal_hsv, dim3(sdn_div_up(M, 256u)), dim3(256), 0, (hipStream_t)stream, rgbs, mask, M, dh, ds, dv);
    return sdn_launch_status();
}

}  // extern "C"
