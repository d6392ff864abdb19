// SealD-NeRF bounding-box seal mapper on the sample stream (scope row "next" #1) for gfx950.
//
// Behavioural contract: SealNeRF/seal_utils.py of the reference --
//   map_mask            :132-153  (points.all(1) & strict AABB test of each bound, then points_in_mesh)
//   points_in_mesh      :675-693  (inside iff the ray along trimesh's test direction AND the opposite ray hit the mesh)
//   moller_trumbore     :638-672  (t, u, v exactly as written there, eps = 1e-8, t >= 0, u >= 0, v >= 0, u + v <= 1)
//   SealBBoxMapper.map_to_origin :245-286 (inverse transform, inverse scale about the source centre, inverse rotation of dirs)
//   modify_hsv :747-758 with color_utils.py:31-63 (rgb -> hsv, + modification, -> rgb)
//   modify_rgb :761-777 (hue / saturation of a target colour, brightness = its V + (V - mean V of the masked samples) + light offset)
//   the `mapSource` redirect of SealBBoxMapper.map_to_origin :269-273 (samples strictly inside the source box are sent to one point --
//   only in calls that map at least one sample: the early return of :251-252 comes first)
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

struct SealSourceArgs {
    float lo[3], hi[3];     // the source box (`empty_bound`)
    float to[3];            // `map_source`
    uint32_t tag;           // this call's tag: *flag == tag <=> some sample of this call was mapped
    uint32_t *flag;         // device word, only ever raised (atomicMax) -- tags increase from call to call, nothing is reset
};

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

// The samples of a call: all M slots, or -- in the device-driven loop, whose sample buffers may hold stale slots of earlier iterations
// beyond the live ones -- the slots of the iteration's live list (count read on the device: live_count[state[3]] with the loop record).
struct SealSlots {
    const uint32_t *live_idx;     // or nullptr = slots 0 .. M-1
    const uint32_t *live_count;
    const int32_t *state;
    uint32_t M;
    __device__ __forceinline__ uint32_t count() const { return live_idx ? (state ? live_count[state[3]] : live_count[0]) : M; }
    __device__ __forceinline__ uint32_t slot(uint32_t j) const { return live_idx ? live_idx[j] : j; }
};

// did this call map any of its samples?  (raises *flag to the call's tag)
__global__ void __launch_bounds__(256) k_seal_any(const uint8_t *__restrict__ mask, SealSlots L, uint32_t *__restrict__ flag, uint32_t tag) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = j < L.count() && mask[L.slot(j)] != 0;
    const unsigned long long vote = __ballot(in ? 1 : 0);
    if (vote != 0ull && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)vote) - 1u) atomicMax(flag, tag);
}

// seal_utils.py:269-273 behind the early return of :251-252: in a call that mapped at least one sample, every UNMAPPED sample strictly
// inside the source box moves to `map_source` (the mapped ones were overwritten after the redirect in the reference: they keep theirs)
__global__ void __launch_bounds__(256) k_seal_source_redirect(float *__restrict__ xyzs, const uint8_t *__restrict__ mask, uint32_t M, SealSourceArgs S) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M || *S.flag != S.tag || mask[i]) return;
    const float x = xyzs[(size_t)i * 3], y = xyzs[(size_t)i * 3 + 1], z = xyzs[(size_t)i * 3 + 2];
    if ((S.hi[0] > x) & (x > S.lo[0]) & (S.hi[1] > y) & (y > S.lo[1]) & (S.hi[2] > z) & (z > S.lo[2])) {
        xyzs[(size_t)i * 3] = S.to[0]; xyzs[(size_t)i * 3 + 1] = S.to[1]; xyzs[(size_t)i * 3 + 2] = S.to[2];
    }
}

// color_utils.py:31-46 for one colour: (h / 6, s, v)
__device__ __forceinline__ void rgb_to_hsv(float r, float g, float b, float &h, float &s, float &v) {
    const float cmax = fmaxf(r, fmaxf(g, b)), cmin = fminf(r, fminf(g, b));
    const float delta = cmax - cmin;
    if (delta == 0.0f) h = 0.0f;
    else if (r >= g && r >= b) { h = (g - b) / delta; h = h - 6.0f * floorf(h / 6.0f); }   // torch `% 6`; the first maximum wins ties
    else if (g >= b) h = (b - r) / delta + 2.0f;
    else h = (r - g) / delta + 4.0f;
    h = h / 6.0f;
    s = cmax == 0.0f ? 0.0f : delta / cmax;
    v = cmax;
}
// color_utils.py:49-63
__device__ __forceinline__ void hsv_to_rgb(float h, float s, float v, float &o0, float &o1, float &o2) {
    const float c = v * s;
    const float h6 = h * 6.0f;
    const float xx = c * (-fabsf((h6 - 2.0f * floorf(h6 / 2.0f)) - 1.0f) + 1.0f);
    const float m = v - c;
    const uint32_t idx = ((uint32_t)(uint8_t)(int)h6) % 6u;   // `.type(torch.uint8)` truncation, then % 6
    switch (idx) {
        case 0: o0 = c; o1 = xx; o2 = 0; break;
        case 1: o0 = xx; o1 = c; o2 = 0; break;
        case 2: o0 = 0; o1 = c; o2 = xx; break;
        case 3: o0 = 0; o1 = xx; o2 = c; break;
        case 4: o0 = xx; o1 = 0; o2 = c; break;
        default: o0 = c; o1 = 0; o2 = xx; break;
    }
    o0 += m; o1 += m; o2 += m;
}

// modify_rgb, pass 1: sum and count of V = max(r, g, b) over the masked samples.  The sum is taken in 2^-40 fixed point in a 64-bit
// integer: exact for the fp16- or fp32-valued colours of [0, 1] up to 2^-40 per sample and, above all, INDEPENDENT OF THE ORDER -- the
// host-stepped loop and the device loop hold the same samples in different slots and must tint them by the same mean, bit for bit
// (torch.mean's own fp32 pairwise order is library-defined; the two agree to ~1e-7, far inside the 1e-4 bar of the fixture test).
__global__ void __launch_bounds__(256) k_seal_rgb_sum(const float *__restrict__ rgbs, const uint8_t *__restrict__ mask, SealSlots L,
                                                      unsigned long long *__restrict__ acc) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0ull, c = 0ull;
    const uint32_t i = j < L.count() ? L.slot(j) : 0u;
    if (j < L.count() && mask[i]) {
        const float mx = fmaxf(rgbs[(size_t)i * 3], fmaxf(rgbs[(size_t)i * 3 + 1], rgbs[(size_t)i * 3 + 2]));
        v = (unsigned long long)((double)fminf(fmaxf(mx, 0.0f), 4.0f) * 1099511627776.0 + 0.5);
        c = 1ull;
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_down(v, off, 64);
        c += __shfl_down(c, off, 64);
    }
    if ((threadIdx.x & 63u) == 0u && c != 0ull) {
        atomicAdd(&acc[0], v);
        atomicAdd(&acc[1], c);
    }
}
// pass 2: hue and saturation of the target colour, brightness re-centred on it (seal_utils.py:769-775), -> rgb, in place
__global__ void __launch_bounds__(256) k_seal_rgb_apply(float *__restrict__ rgbs, const uint8_t *__restrict__ mask, uint32_t M, float tr, float tg, float tb,
                                                        float light_offset, const unsigned long long *__restrict__ acc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M || !mask[i] || acc[1] == 0ull) return;      // (a masked slot outside the call's samples while none of them is masked: left alone)
    const float mean = (float)(((double)acc[0] / 1099511627776.0) / (double)acc[1]);
    float h, s, v, mh, ms, mv;
    rgb_to_hsv(rgbs[(size_t)i * 3], rgbs[(size_t)i * 3 + 1], rgbs[(size_t)i * 3 + 2], h, s, v);
    rgb_to_hsv(tr, tg, tb, mh, ms, mv);
    const float nv = fminf(1.0f, fmaxf(0.0f, (mv + (v - mean)) + light_offset));
    float o0, o1, o2;
    hsv_to_rgb(mh, ms, nv, o0, o1, o2);
    rgbs[(size_t)i * 3] = o0; rgbs[(size_t)i * 3 + 1] = o1; rgbs[(size_t)i * 3 + 2] = o2;
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

// sdn_seal_bbox_map with the `mapSource` option: source_bound {lo xyz, hi xyz}, map_source [3]; flag: one device word owned by the caller
// (zeroed once when allocated; raised to this call's tag if one of the call's samples is mapped).  live_idx / live_count / state: the
// call's samples as a list (the device-driven loop; all NULL = the M slots).  Three launches, no host synchronisation.
int sdn_seal_bbox_map_source(float *xyzs, float *dirs, uint32_t M, const float *bounds, uint32_t n_bounds, const float *tris, uint32_t n_tris,
                             const float *test_dir, const float *tinv, const float *rinv, const float *scale, const float *center,
                             const float *source_bound, const float *map_source, uint32_t *flag, uint8_t *mask, const uint32_t *live_idx,
                             const uint32_t *live_count, const int32_t *state, void *stream) {
    if (M == 0) return 0;
    if (!source_bound || !map_source || !flag || (live_idx && !live_count)) return SDN_E_BADARG;
    int rc = sdn_seal_bbox_map(xyzs, dirs, M, bounds, n_bounds, tris, n_tris, test_dir, tinv, rinv, scale, center, mask, stream);
    if (rc) return rc;
    static uint32_t next_tag = 0;          // tags only grow (a wrap after 2^32 calls would need the flag cleared: not in this process's life)
    const uint32_t tag = ++next_tag;
    SealSourceArgs sa;
    for (int k = 0; k < 3; k++) { sa.lo[k] = source_bound[k]; sa.hi[k] = source_bound[3 + k]; sa.to[k] = map_source[k]; }
    sa.tag = tag; sa.flag = flag;
    const SealSlots L{live_idx, live_count, state, M};
    hipLaunchKernelGGL(k_seal_any, dim3(sdn_div_up(M, 256u)), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)mask, L, flag, tag);
    hipLaunchKernelGGL(k_seal_source_redirect, dim3(sdn_div_up(M, 256u)), dim3(256), 0, (hipStream_t)stream, xyzs, (const uint8_t *)mask, M, sa);
    return sdn_launch_status();
}

// modify_rgb (seal_utils.py:761-777) on the masked samples, in place: target colour rgb [3], light offset; scratch: 16 bytes of device
// memory (cleared here); live list as above (the MEAN is taken over the call's samples only).  Three stream operations, no host
// synchronisation; the mean is order-independent (see k_seal_rgb_sum).
int sdn_seal_modify_rgb(float *rgbs, const uint8_t *mask, uint32_t M, float r, float g, float b, float light_offset, void *scratch16,
                        const uint32_t *live_idx, const uint32_t *live_count, const int32_t *state, void *stream) {
    if (M == 0) return 0;
    if (!rgbs || !mask || !scratch16 || ((uintptr_t)scratch16 & 7u) != 0 || (live_idx && !live_count)) return SDN_E_BADARG;
    if (hipMemsetAsync(scratch16, 0, 16, (hipStream_t)stream) != hipSuccess) return sdn_launch_status();
    const SealSlots L{live_idx, live_count, state, M};
    hipLaunchKernelGGL(k_seal_rgb_sum, dim3(sdn_div_up(M, 256u)), dim3(256), 0, (hipStream_t)stream, (const float *)rgbs, mask, L, (unsigned long long *)scratch16);
    hipLaunchKernelGGL(k_seal_rgb_apply, dim3(sdn_div_up(M, 256u)), dim3(256), 0, (hipStream_t)stream, rgbs, mask, M, r, g, b, light_offset,
                       (const unsigned long long *)scratch16);
    return sdn_launch_status();
}

int sdn_seal_modify_hsv(float *rgbs, const uint8_t *mask, uint32_t M, float dh, float ds, float dv, void *stream) {
    if (M == 0) return 0;
    if (!rgbs || !mask) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_seal_hsv, dim3(sdn_div_up(M, 256u)), dim3(256), 0, (hipStream_t)stream, rgbs, mask, M, dh, ds, dv);
    return sdn_launch_status();
}

}  // extern "C"
