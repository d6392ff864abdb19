// Occupancy-grid ray marching + alpha compositing for gfx950 (MI355X).
//
// Behavioural contract: raymarching/src/raymarching.cu of the reference
// (line ranges cited per kernel).  Arithmetic contract: this file is built with
// -ffp-contract=off so that every float operation rounds on its own, in the
// order the reference source writes it -- which is what oracle/sdn_oracle.c
// evaluates on the CPU; ray/sample indices and counts are therefore bit-exact
// against the oracle, not merely close.
//
// Layout notes (MI355X): one lane per ray, 256-thread workgroups (4 waves, one
// per SIMD).  The density bitfield of one time slice is 256 KiB and lives in the
// XCD's L2 after first touch; the marcher is bound by dependent L2 byte gathers
// and by wave divergence, not by HBM.  Rays keep image order in rays_alive
// (stable compaction), so a wave's 64 rays are neighbouring pixels and take
// similar trip counts.
#include <stdlib.h>

#include "sdn_common.h"
#include "sdn_internal.h"

namespace {
using sdn_int::FrameSel;

constexpr float kSqrt3 = 1.7320508075688772f;
constexpr float kRPi = 0.3183098861837907f;

__device__ __forceinline__ float signf_(float x) { return copysignf(1.0f, x); }
__device__ __forceinline__ float clampf_(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// raymarching.cu:42-47
__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)exponent));
}
// raymarching.cu:49-54
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)exponent));
}
// raymarching.cu:56-81
__host__ __device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ __forceinline__ uint32_t morton3D_(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
// coordinates < 256: the first spreading step of expand_bits is the identity
__device__ __forceinline__ uint32_t expand_bits8(uint32_t v) {
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D_8bit(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits8(x) | (expand_bits8(y) << 1) | (expand_bits8(z) << 2);
}
__host__ __device__ __forceinline__ uint32_t morton3D_invert_(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// ---------------------------------------------------------------------------
// The DDA stepper shared by the three marching kernels
// (raymarching.cu:359-400 == 427-479 == 750-804).
// ---------------------------------------------------------------------------
constexpr uint32_t kCullRes = 32;  // cull grid resolution (see "Exact early-out" below)
constexpr uint32_t kCullWords = kCullRes * kCullRes * kCullRes / 32;  // 1024 words of marks, followed by 8 words of meta:
// meta = {x0, y0, z0, x1, y1, z1 (inclusive bounding box of the marked cells), -, -}
constexpr uint32_t kFineCacheCells = 4096;  // LDS budget for the fine-bit cache: 32 KiB
// meta[6] = 1: the cull-grid buffer also carries the PACKED fine-bit image of the marked bounding box (one 64-bit word per 4x4x4 block,
// x fastest) behind the meta record -- built once per occupancy slice, so a marching workgroup fills its LDS cache with a contiguous
// copy instead of a Morton gather with three integer divisions per word.
constexpr uint32_t kCullImageWord = kCullWords + 8;   // uint32 offset of the image (16-byte aligned)
__device__ __forceinline__ uint32_t m2(uint32_t v) { return (v & 1u) | ((v & 2u) << 2); }  // 2-bit Morton spread
__device__ __forceinline__ bool cull_marked(const uint32_t *cull_bits, int cx, int cy, int cz) {
    const uint32_t c = ((uint32_t)cz * kCullRes + (uint32_t)cy) * kCullRes + (uint32_t)cx;
    return (cull_bits[c >> 5] >> (c & 31u)) & 1u;
}

// FAST = (cascade == 1, bound == 1, H a power of two <= 256): the configuration dnerf runs (bound 1, grid 128).
// Then the mip level is always 0 and every scaling in the index / voxel-edge arithmetic is by a power of two, i.e.
// exact, so the float-only forms below produce the same bits as the reference's double / multi-step expressions.
template <bool FAST>
struct MarcherT {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
    float rH, H3, Hf, Cf, Hm1;
    float bound, dt_gamma, dt_min, dt_max;
    float halfH, twoRH, ex, ey, ez;  // FAST only
    float dt_const;                  // step when dt_gamma == 0 (clamp(t * 0, dt_min, dt_max) for every finite t)
    bool dt_is_const;
    double Hd;
    const uint8_t *__restrict__ grid;
    // FAST only: LDS copy of the fine bits of the cull grid's bounding box (one 64-bit word = one 4x4x4 voxel block)
    const unsigned long long *fine = nullptr;
    int fx0 = 0, fy0 = 0, fz0 = 0, fnx = 0, fny = 0, fnz = 0;

    __device__ __forceinline__ void init(const float *o, const float *d, float bound_, float dt_gamma_,
                                         uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid_) {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        rdx = 1 / dx; rdy = 1 / dy; rdz = 1 / dz;
        rH = 1 / (float)H;
        H3 = (float)(H * H * H);
        Hf = (float)H; Hd = (double)H; Cf = (float)C; Hm1 = (float)(H - 1);
        bound = bound_; dt_gamma = dt_gamma_;
        dt_min = 2 * kSqrt3 / (float)max_steps;
        dt_max = 2 * kSqrt3 * (float)(1 << (C - 1)) / (float)H;
        dt_is_const = (dt_gamma_ == 0.0f);
        dt_const = clampf_(0.0f, dt_min, dt_max);
        grid = grid_;
        halfH = 0.5f * Hf; twoRH = 2.0f * rH;
        // nx + 0.5f + 0.5f * signf(d) == nx + (signbit(d) ? 0 : 1), exactly
        ex = signbit(dx) ? 0.0f : 1.0f; ey = signbit(dy) ? 0.0f : 1.0f; ez = signbit(dz) ? 0.0f : 1.0f;
    }

    __device__ __forceinline__ float step_size(float t) const { return dt_is_const ? dt_const : clampf_(t * dt_gamma, dt_min, dt_max); }

    // ---- the pieces of one FAST loop-body evaluation (shared by the sequential chain below and the cooperative marcher) ----
    // position and voxel of the parameter t (raymarching.cu:752-768 with the exact float forms of the FAST configuration)
    __device__ __forceinline__ void locate(float t, float &x, float &y, float &z, int &nx, int &ny, int &nz) const {
        x = clampf_(ox + t * dx, -bound, bound);
        y = clampf_(oy + t * dy, -bound, bound);
        z = clampf_(oz + t * dz, -bound, bound);
        nx = (int)clampf_((x + 1) * halfH, 0.0f, Hm1);
        ny = (int)clampf_((y + 1) * halfH, 0.0f, Hm1);
        nz = (int)clampf_((z + 1) * halfH, 0.0f, Hm1);
    }
    // occupancy bit of a voxel (raymarching.cu:770-772)
    __device__ __forceinline__ bool occupied(int nx, int ny, int nz, const uint32_t *cull_bits) const {
        bool occ = false;
        if (fine) {
            // The LDS image holds the fine bits of every 4x4x4 block inside the bounding box of the marked cull cells, and
            // every occupied voxel lies in a marked cell: one LDS read answers the probe inside the box (a block of an unmarked
            // cell reads as zero), three register compares answer it outside.  bit = Morton code of the low two bits per axis.
            const int bx = (nx >> 2) - fx0, by = (ny >> 2) - fy0, bz = (nz >> 2) - fz0;
            if ((uint32_t)bx < (uint32_t)fnx && (uint32_t)by < (uint32_t)fny && (uint32_t)bz < (uint32_t)fnz) {
                const uint32_t b = m2((uint32_t)nx & 3u) | (m2((uint32_t)ny & 3u) << 1) | (m2((uint32_t)nz & 3u) << 2);
                // (24-bit multiplies: full-rate v_mad_u32_u24 instead of quarter-rate v_mul_lo_u32; all factors < 32)
                occ = (fine[__umul24(__umul24((uint32_t)bz, (uint32_t)fny) + (uint32_t)by, (uint32_t)fnx) + (uint32_t)bx] >> b) & 1ull;
            }
        } else if (!cull_bits || cull_marked(cull_bits, nx >> 2, ny >> 2, nz >> 2)) {
            // an unmarked cull cell holds no occupied voxel: the fine bit (a dependent L2 load) is only fetched near the object
            const uint32_t index = morton3D_8bit((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
            occ = grid[index >> 3] & (1u << (index & 7u));
        }
        return occ;
    }
    // parameter at which the ray leaves the voxel, evaluated at t (raymarching.cu:786-791)
    __device__ __forceinline__ float exit_t(float t, float x, float y, float z, int nx, int ny, int nz) const {
        const float tx = ((((float)nx + ex) * twoRH - 1) - x) * rdx;
        const float ty = ((((float)ny + ey) * twoRH - 1) - y) * rdy;
        const float tz = ((((float)nz + ez) * twoRH - 1) - z) * rdz;
        return t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }

    // One loop-body evaluation at parameter t.  Occupied: returns true with the sample in
    // (x,y,z,dt), t untouched.  Empty: returns false with t advanced past the voxel.
    __device__ __forceinline__ bool probe(float &t, float &x, float &y, float &z, float &dt, const uint32_t *cull_bits = nullptr) const {
        if constexpr (FAST) {
            int nx, ny, nz;
            locate(t, x, y, z, nx, ny, nz);
            dt = step_size(t);
            if (occupied(nx, ny, nz, cull_bits)) return true;
            const float tt = exit_t(t, x, y, z, nx, ny, nz);
            if (dt_is_const) {
                // `do t += dt while (t < tt)` with the same additions in the same order, without a divergent loop: one 128^3 voxel is
                // crossed in at most 8 steps of dt_min = 2 sqrt(3) / 1024; anything longer falls through to the loop
                const float d = dt_const;
                t += d;
                #pragma unroll
                for (int k = 0; k < 8; k++) t = (t < tt) ? t + d : t;
                while (t < tt) t += d;
            } else {
                do {
                    t += step_size(t);
                } while (t < tt);
            }
            return false;
        } else {
            x = clampf_(ox + t * dx, -bound, bound);
            y = clampf_(oy + t * dy, -bound, bound);
            z = clampf_(oz + t * dz, -bound, bound);
            dt = step_size(t);
            const int l0 = mip_from_pos(x, y, z, Cf), l1 = mip_from_dt(dt, Hf, Cf);
            const int level = l0 > l1 ? l0 : l1;
            const float mip_bound = fminf(scalbnf(1.0f, level), bound);
            const float mip_rbound = 1 / mip_bound;
            // `0.5 * (...) * H` is a double expression in the reference (0.5 is a double literal)
            const int nx = (int)clampf_((float)(0.5 * (double)(x * mip_rbound + 1) * Hd), 0.0f, Hm1);
            const int ny = (int)clampf_((float)(0.5 * (double)(y * mip_rbound + 1) * Hd), 0.0f, Hm1);
            const int nz = (int)clampf_((float)(0.5 * (double)(z * mip_rbound + 1) * Hd), 0.0f, Hm1);
            const uint32_t index = (uint32_t)((float)level * H3 + (float)morton3D_((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
            const bool occ = grid[index >> 3] & (1u << (index & 7u));
            if (occ) return true;
            const float tx = (((nx + 0.5f + 0.5f * signf_(dx)) * rH * 2 - 1) * mip_bound - x) * rdx;
            const float ty = (((ny + 0.5f + 0.5f * signf_(dy)) * rH * 2 - 1) * mip_bound - y) * rdy;
            const float tz = (((nz + 0.5f + 0.5f * signf_(dz)) * rH * 2 - 1) * mip_bound - z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do {
                t += step_size(t);
            } while (t < tt);
            return false;
        }
    }
};
using Marcher = MarcherT<false>;

static inline bool fast_config(float bound, uint32_t C, uint32_t H) {
    return C == 1 && bound == 1.0f && H >= 2 && H <= 256 && (H & (H - 1)) == 0;
}

// ---------------------------------------------------------------------------
// Exact early-out for rays that cannot produce a sample ("cull grid").
// cull[32^3] (x fastest) marks coarse cells (4^3 fine voxels of the 128^3 grid) whose 3x3x3 coarse neighbourhood
// holds any occupied voxel.  A ray whose remaining segment [t, far], tested every cell width, only sees unmarked
// cells stays >= half a coarse cell (2 fine voxels) away from every occupied voxel, so the reference's marcher --
// whose probe points lie on that segment up to float rounding -- emits nothing for it: returning "no samples"
// is exact, not an approximation.  Rays that may hit take the full reference chain from their own t.
// ---------------------------------------------------------------------------

__device__ __forceinline__ const uint8_t *frame_grid_uniform_y(const sdn_int::FrameSel &fs) { return fs.grid[blockIdx.y]; }

__global__ void __launch_bounds__(256) k_build_cull_grid(const uint8_t *__restrict__ bitfield, uint32_t *__restrict__ cull_bits,
                                                         sdn_int::FrameSel fs) {
    if (fs.n_frames > 1) {   // frame group: blockIdx.y = frame
        bitfield = frame_grid_uniform_y(fs);
        cull_bits += (size_t)blockIdx.y * fs.cull_stride;
    }
    const uint32_t c = threadIdx.x + blockIdx.x * blockDim.x;  // grid.x is exactly 32^3 threads
    const int cx = c & 31, cy = (c >> 5) & 31, cz = c >> 10;
    const unsigned long long *__restrict__ blocks = reinterpret_cast<const unsigned long long *>(bitfield);
    bool any = false;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                if (x < 0 || y < 0 || z < 0 || x > 31 || y > 31 || z > 31) continue;
                // a 4x4x4 block of the 128^3 grid is 64 consecutive Morton bits = one aligned 8-byte word
                any |= blocks[morton3D_((uint32_t)x, (uint32_t)y, (uint32_t)z)] != 0ull;
            }
    const unsigned long long m = __ballot(any);  // lane i <-> cell c0 + i
    if ((threadIdx.x & 63u) == 0) {
        cull_bits[c >> 5] = (uint32_t)m;
        cull_bits[(c >> 5) + 1] = (uint32_t)(m >> 32);
    }
    if (any) {
        int *meta = reinterpret_cast<int *>(cull_bits + kCullWords);
        atomicMin(meta + 0, cx); atomicMin(meta + 1, cy); atomicMin(meta + 2, cz);
        atomicMax(meta + 3, cx); atomicMax(meta + 4, cy); atomicMax(meta + 5, cz);
    }
}

__global__ void k_cull_meta_init(uint32_t *__restrict__ cull_bits, uint32_t n_frames, uint32_t cull_stride) {
    if (threadIdx.x < n_frames && blockIdx.x == 0) {
        int *meta = reinterpret_cast<int *>(cull_bits + (size_t)threadIdx.x * cull_stride + kCullWords);
        meta[0] = meta[1] = meta[2] = (int)kCullRes;
        meta[3] = meta[4] = meta[5] = -1;
        meta[6] = meta[7] = 0;
    }
}

__global__ void __launch_bounds__(256) k_build_fine_image(const uint8_t *__restrict__ bitfield, uint32_t *__restrict__ cull_bits, sdn_int::FrameSel fs) {
    if (fs.n_frames > 1) {   // frame group: blockIdx.y = frame
        bitfield = frame_grid_uniform_y(fs);
        cull_bits += (size_t)blockIdx.y * fs.cull_stride;
    }
    int *meta = reinterpret_cast<int *>(cull_bits + kCullWords);
    const int fx0 = meta[0], fy0 = meta[1], fz0 = meta[2];
    const int fnx = meta[3] - fx0 + 1, fny = meta[4] - fy0 + 1, fnz = meta[5] - fz0 + 1;
    const bool fits = fnx > 0 && fny > 0 && fnz > 0 && (uint32_t)(fnx * fny * fnz) <= kFineCacheCells;
    const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i == 0) meta[6] = fits ? 1 : 0;
    if (!fits || i >= (uint32_t)(fnx * fny * fnz)) return;
    const unsigned long long *__restrict__ blocks = reinterpret_cast<const unsigned long long *>(bitfield);
    unsigned long long *img = reinterpret_cast<unsigned long long *>(cull_bits + kCullImageWord);
    const int cx = fx0 + (int)i % fnx, cy = fy0 + ((int)i / fnx) % fny, cz = fz0 + (int)i / (fnx * fny);
    img[i] = blocks[morton3D_8bit((uint32_t)cx, (uint32_t)cy, (uint32_t)cz)];
}

// Scans the remaining segment [t, far] once per cull-cell width.  Returns false if no marked cell is met (the ray
// cannot produce a sample).  Otherwise t_end receives a parameter beyond which no marked cell is met any more: the
// marcher may stop there -- the reference would only step through empty voxels from there to `far`.
// t_safe (optional): a parameter up to which the ray certainly stays >= 2 fine voxels away from every occupied voxel -- the scan
// position one cell width before the first marked one (-FLT_MAX when unknown): where the cooperative marcher may start looking.
__device__ __forceinline__ bool ray_may_hit(const uint32_t *cull_bits, float ox, float oy, float oz, float dx, float dy, float dz,
                                            float t, float far, float &t_end, int bx0, int by0, int bz0, int bnx, int bny, int bnz,
                                            float *t_safe = nullptr) {
    if (t_safe) *t_safe = -__FLT_MAX__;
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float ds = (2.0f / kCullRes) / fmaxf(len, 1e-12f);  // parameter step = one cull cell along the ray
    t_end = far;
    if (bnx <= 0 || bny <= 0 || bnz <= 0) return false;       // nothing is marked: nothing is occupied
    // Marked cells only exist inside their bounding box: scan just the part of [t, far] that can be inside it (slab test,
    // widened by one scan step against rounding), not the whole chord of the unit cube.
    const float cw = 2.0f / kCullRes;
    float s0 = t, s1 = far;
    {
        const float lo[3] = {(float)bx0 * cw - 1.0f, (float)by0 * cw - 1.0f, (float)bz0 * cw - 1.0f};
        const float hi[3] = {(float)(bx0 + bnx) * cw - 1.0f, (float)(by0 + bny) * cw - 1.0f, (float)(bz0 + bnz) * cw - 1.0f};
        const float o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
        #pragma unroll
        for (int a = 0; a < 3; a++) {
            if (d[a] != 0.0f) {
                const float r = 1.0f / d[a];
                const float ta = (lo[a] - o[a]) * r, tb = (hi[a] - o[a]) * r;
                s0 = fmaxf(s0, fminf(ta, tb) - ds);
                s1 = fminf(s1, fmaxf(ta, tb) + ds);
            } else if (o[a] < lo[a] - cw || o[a] > hi[a] + cw) {
                return false;                                  // parallel to the slab and outside it
            }
        }
    }
    if (!(s0 <= s1)) return false;
    float s = s0;
    bool hit = false;
    // bounded: a unit-cube diagonal is 2*sqrt(3) / (2/32) = 56 cells; anything longer (degenerate direction, huge far)
    // falls through to "may hit, no early end" and takes the ordinary marcher
    for (int it = 0; it < 96; it++, s += ds) {
        const float ss = fminf(s, s1);
        const float x = clampf_(ox + ss * dx, -1.0f, 1.0f), y = clampf_(oy + ss * dy, -1.0f, 1.0f), z = clampf_(oz + ss * dz, -1.0f, 1.0f);
        const int cx = (int)fminf((x + 1) * (0.5f * kCullRes), (float)(kCullRes - 1));
        const int cy = (int)fminf((y + 1) * (0.5f * kCullRes), (float)(kCullRes - 1));
        const int cz = (int)fminf((z + 1) * (0.5f * kCullRes), (float)(kCullRes - 1));
        if (cull_marked(cull_bits, cx, cy, cz)) {
            if (!hit && t_safe) *t_safe = ss - ds;
            hit = true; t_end = ss + ds;
        }
        if (s >= s1) return hit;
    }
    t_end = far;
    if (t_safe) *t_safe = -__FLT_MAX__;
    return true;
}

// LDS-resident occupancy caches of the marching kernels (FAST configuration with a cull grid only)
struct OccCache {
    const uint32_t *s_cull = nullptr;            // 32^3 mark bits
    const unsigned long long *fine = nullptr;    // fine bits of the marked bounding box (one word = 4x4x4 voxels)
    int fx0 = 0, fy0 = 0, fz0 = 0, fnx = 0, fny = 0, fnz = 0;   // bounding box of the marked cull cells (origin, extent)
};

// Cooperative load by a 256-thread workgroup; contains a barrier, so every thread of the workgroup must call it.
template <bool FAST>
__device__ __forceinline__ void occ_cache_load(const uint32_t *__restrict__ cull, const uint8_t *__restrict__ grid, uint4 *s_cull4,
                                               unsigned long long *s_fine, OccCache &oc) {
    if constexpr (FAST) {
        if (cull) {  // kernel-uniform
            s_cull4[threadIdx.x] = reinterpret_cast<const uint4 *>(cull)[threadIdx.x];
            const int *meta = reinterpret_cast<const int *>(cull + kCullWords);
            oc.fx0 = meta[0]; oc.fy0 = meta[1]; oc.fz0 = meta[2];
            oc.fnx = meta[3] - oc.fx0 + 1; oc.fny = meta[4] - oc.fy0 + 1;
            const int fnz = meta[5] - oc.fz0 + 1;
            oc.fnz = fnz;
            if (oc.fnx > 0 && oc.fny > 0 && fnz > 0 && (uint32_t)(oc.fnx * oc.fny * fnz) <= kFineCacheCells && meta[6] == 1) {
                const int pairs = (oc.fnx * oc.fny * fnz + 1) / 2;        // contiguous copy, 16 bytes per lane
                const uint4 *__restrict__ img = reinterpret_cast<const uint4 *>(cull + kCullImageWord);
                for (int i = (int)threadIdx.x; i < pairs; i += 256) reinterpret_cast<uint4 *>(s_fine)[i] = img[i];
                oc.fine = s_fine;
            } else if (oc.fnx > 0 && oc.fny > 0 && fnz > 0 && (uint32_t)(oc.fnx * oc.fny * fnz) <= kFineCacheCells) {
                const unsigned long long *__restrict__ blocks = reinterpret_cast<const unsigned long long *>(grid);
                const int cells = oc.fnx * oc.fny * fnz;
                for (int i = (int)threadIdx.x; i < cells; i += 256) {
                    const int cx = oc.fx0 + i % oc.fnx, cy = oc.fy0 + (i / oc.fnx) % oc.fny, cz = oc.fz0 + i / (oc.fnx * oc.fny);
                    s_fine[i] = blocks[morton3D_8bit((uint32_t)cx, (uint32_t)cy, (uint32_t)cz)];
                }
                oc.fine = s_fine;
            }
            __syncthreads();
            oc.s_cull = reinterpret_cast<const uint32_t *>(s_cull4);
        }
    }
}

// The same in two steps for the cooperative marcher: the 4 KiB of marks (needed by the cull scan of phase A) first, the <= 32 KiB of
// fine bits only by workgroups that have a ray to march.  Both contain a barrier.
__device__ __forceinline__ void occ_cache_load_marks(const uint32_t *__restrict__ cull, uint4 *s_cull4, OccCache &oc) {
    if (cull) {  // kernel-uniform
        s_cull4[threadIdx.x] = reinterpret_cast<const uint4 *>(cull)[threadIdx.x];
        const int *meta = reinterpret_cast<const int *>(cull + kCullWords);
        oc.fx0 = meta[0]; oc.fy0 = meta[1]; oc.fz0 = meta[2];
        oc.fnx = meta[3] - oc.fx0 + 1; oc.fny = meta[4] - oc.fy0 + 1; oc.fnz = meta[5] - oc.fz0 + 1;
        oc.s_cull = reinterpret_cast<const uint32_t *>(s_cull4);
    }
    __syncthreads();
}
__device__ __forceinline__ void occ_cache_load_fine(const uint32_t *__restrict__ cull, const uint8_t *__restrict__ grid, unsigned long long *s_fine,
                                                    OccCache &oc) {
    if (cull && oc.fnx > 0 && oc.fny > 0 && oc.fnz > 0 && (uint32_t)(oc.fnx * oc.fny * oc.fnz) <= kFineCacheCells &&
        reinterpret_cast<const int *>(cull + kCullWords)[6] == 1) {
        const int pairs = (oc.fnx * oc.fny * oc.fnz + 1) / 2;
        const uint4 *__restrict__ img = reinterpret_cast<const uint4 *>(cull + kCullImageWord);
        for (int i = (int)threadIdx.x; i < pairs; i += 256) reinterpret_cast<uint4 *>(s_fine)[i] = img[i];
        oc.fine = s_fine;
    } else if (cull && oc.fnx > 0 && oc.fny > 0 && oc.fnz > 0 && (uint32_t)(oc.fnx * oc.fny * oc.fnz) <= kFineCacheCells) {
        const unsigned long long *__restrict__ blocks = reinterpret_cast<const unsigned long long *>(grid);
        const int cells = oc.fnx * oc.fny * oc.fnz;
        for (int i = (int)threadIdx.x; i < cells; i += 256) {
            const int cx = oc.fx0 + i % oc.fnx, cy = oc.fy0 + (i / oc.fnx) % oc.fny, cz = oc.fz0 + i / (oc.fnx * oc.fny);
            s_fine[i] = blocks[morton3D_8bit((uint32_t)cx, (uint32_t)cy, (uint32_t)cz)];
        }
        oc.fine = s_fine;
    }
    __syncthreads();
}


// ---------------------------------------------------------------------------
// utils
// ---------------------------------------------------------------------------
// raymarching.cu:92-145
__global__ void __launch_bounds__(256) k_near_far_from_aabb(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                            const float *__restrict__ aabb, uint32_t N, float min_near,
                                                            float *__restrict__ nears, float *__restrict__ fars) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float rdx = 1 / rays_d[n * 3], rdy = 1 / rays_d[n * 3 + 1], rdz = 1 / rays_d[n * 3 + 2];
    const float mx = __FLT_MAX__;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
    if (near > far) { float c = near; near = far; far = c; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }
    if (near > far_y || near_y > far) { nears[n] = mx; fars[n] = mx; return; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;
    float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
    if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
    if (near > far_z || near_z > far) { nears[n] = mx; fars[n] = mx; return; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;
    if (near < min_near) near = min_near;
    nears[n] = near;
    fars[n] = far;
}

// raymarching.cu:163-198
__global__ void __launch_bounds__(256) k_sph_from_ray(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                      float radius, uint32_t N, float *__restrict__ coords) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;
    const float C = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * C)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);
    const float phi = atan2f(z, x);
    coords[n * 2] = 2 * theta * kRPi - 1;
    coords[n * 2 + 1] = phi * kRPi;
}

// raymarching.cu:214-254
__global__ void __launch_bounds__(256) k_morton3D(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    indices[n] = (int32_t)morton3D_((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
__global__ void __launch_bounds__(256) k_morton3D_invert(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const int ind = indices[n];
    coords[n * 3] = (int32_t)morton3D_invert_((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int32_t)morton3D_invert_((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int32_t)morton3D_invert_((uint32_t)(ind >> 2));
}

// raymarching.cu:268-289.  One lane per output byte; the 8 floats are fetched as two float4
// (32 B per lane, 2 KiB per wave-instruction pair: fully coalesced).
__global__ void __launch_bounds__(256) k_packbits(const float4 *__restrict__ grid, uint32_t N, float thresh, uint8_t *__restrict__ bitfield) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float4 a = grid[(size_t)n * 2], b = grid[(size_t)n * 2 + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;
    bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;
    bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;
    bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;
    bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

// ---------------------------------------------------------------------------
// training march: count pass -> deterministic exclusive scan -> write pass
// (raymarching.cu:312-480; the reference allocates slots with two racing atomicAdds)
// ---------------------------------------------------------------------------
constexpr uint32_t kScanBlock = 1024;

template <bool FAST>
__global__ void __launch_bounds__(256) k_march_train_count(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                           const uint8_t *__restrict__ grid, float bound, float dt_gamma,
                                                           uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                           const float *__restrict__ nears, const float *__restrict__ fars,
                                                           const float *__restrict__ noises, uint32_t *__restrict__ num_steps_out,
                                                           const uint32_t *__restrict__ cull, float *__restrict__ sample_t) {
    // The one marching pass of a training step: counts every ray's samples AND records their parameters t (sample_t
    // [N, max_steps]), from which the emit pass rebuilds positions and deltas with the marcher's own expressions -- the reference
    // (and the first version of this file) marches every ray a second time to write them.  Same LDS occupancy caches and exact
    // cull-grid early-out as the inference marcher: most rays of a training batch miss the object and stop here at once.
    __shared__ uint4 s_cull4[FAST ? 256 : 1];
    __shared__ unsigned long long s_fine[FAST ? kFineCacheCells : 1];
    OccCache oc;
    occ_cache_load<FAST>(cull, grid, s_cull4, s_fine, oc);
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    MarcherT<FAST> m;
    m.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid);
    m.fine = oc.fine; m.fx0 = oc.fx0; m.fy0 = oc.fy0; m.fz0 = oc.fz0; m.fnx = oc.fnx; m.fny = oc.fny; m.fnz = oc.fnz;
    const float far = fars[n];
    float t = nears[n];
    t += m.step_size(t) * noises[n];
    uint32_t num_steps = 0;
    float x, y, z, dt;
    bool go = t < far;
    float t_end = far;
    if (FAST && oc.s_cull && go)
        go = ray_may_hit(oc.s_cull, m.ox, m.oy, m.oz, m.dx, m.dy, m.dz, t, far, t_end, oc.fx0, oc.fy0, oc.fz0, oc.fnx, oc.fny, oc.fnz);
    float *ts = sample_t + (size_t)n * max_steps;
    while (go && t < far && t < t_end && num_steps < max_steps) {
        if (m.probe(t, x, y, z, dt, oc.s_cull)) {
            ts[num_steps] = t;
            num_steps++;
            t += dt;
        }
    }
    num_steps_out[n] = num_steps;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Wave-per-ray form of the counting pass (FAST configuration with a constant step, i.e. dt_gamma == 0: what dnerf runs).
//
// With a constant step every parameter the marcher ever visits is a point of ONE lattice  L_0 = t_start, L_{k+1} = fl(L_k + dt):
// a sample advances by dt, and an empty voxel is left by `do t += dt while (t < tt)` -- the same additions.  Which lattice points
// are VISITED is a chain: next(k) = k + 1 if L_k lies in an occupied voxel, else the first m > k with L_m >= tt_k (tt_k: the voxel's
// exit parameter, from L_k alone).  So 64 lanes take 64 consecutive lattice points, evaluate occupancy / exit parameter in
// parallel with the sequential marcher's own expressions, and the wave walks the chain over ballots (scalar work, one readlane
// per empty voxel).  Visited set, samples, counts and the recorded parameters are the sequential kernel's bit for bit: nothing
// is approximated, the dependent chain of ~1000 cycles per voxel probe just becomes one probe round per 64 lattice points.
// One ray per wave instead of one per lane: 4096 rays fill the chip (4 waves per SIMD) where the lane-per-ray kernel kept 64
// waves busy for as long as its longest ray.  Occupancy bits and cull marks are read from global memory (L2-resident, 256 KiB +
// 4 KiB): a probe round is one parallel load, an LDS image per 4-ray workgroup would cost more than it saves.
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_lattice(float base, float dt, uint32_t lane, float &next_base, float &closed_step) {
    // L_{lane} of the lattice starting at base, by `lane` successive additions (float addition is not associative: the values
    // must come from the recurrence); also returns L_64.
    // Closed form for the common case.  While L stays in one binade every L_k is a multiple of the binade's ulp U, so
    // fl(L_k + dt) = L_k + c with the SAME c = round(dt / U) U for every k -- unless dt / U sits exactly between two integers (a tie
    // rounds to even, which depends on L_k).  Then L_k = base + k c exactly (k c and the sum are multiples of U below 2^24 U).  All
    // conditions are wave-uniform; anything else (binade crossing inside the window, a tie, a tiny base) takes the recurrence.
    {
        const float first = base + dt;
        const float c = first - base;                               // exact (Sterbenz-like: both multiples of U, |c| << base)
        const float err = dt - c;                                   // rounding error of base + dt (FastTwoSum, exact for base >= dt)
        const float last = base + 64.0f * c;
        const uint32_t eb = __float_as_uint(base) >> 23, el = __float_as_uint(last) >> 23;     // sign bit clear: t > 0
        const bool normal = base >= dt && dt > 0.0f && eb > 30u && eb < 255u;
        const float half_ulp = __uint_as_float((eb - 24u) << 23), c_cap = __uint_as_float((eb - 5u) << 23);   // U / 2, 2^18 U
        if (normal && eb == el && fabsf(err) != half_ulp && c < c_cap) {
            next_base = last;
            closed_step = c;                                        // L_k = base + k c holds for k = 0 .. 64
            return base + (float)lane * c;
        }
    }
    closed_step = 0.0f;
    float v = base, mine = base;
    #pragma unroll 8
    for (uint32_t i = 0; i < 64; i++) {
        if (i == lane) mine = v;
        v += dt;
    }
    next_base = v;
    return mine;
}

__global__ void __launch_bounds__(256) k_march_train_count_wave(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                                const uint8_t *__restrict__ grid, float bound, uint32_t max_steps, uint32_t N,
                                                                uint32_t H, const float *__restrict__ nears, const float *__restrict__ fars,
                                                                const float *__restrict__ noises, uint32_t *__restrict__ num_steps_out,
                                                                const uint32_t *__restrict__ cull, float *__restrict__ sample_t) {
    const uint32_t n = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (n >= N) return;   // wave-uniform
    MarcherT<true> m;
    m.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, 0.0f, max_steps, 1u, H, grid);
    const float far = fars[n], dt = m.dt_const;
    float t0 = nears[n];
    t0 += m.step_size(t0) * noises[n];
    bool go = t0 < far;
    float t_end = far;
    if (cull && go) {
        // ray_may_hit with the scan positions spread over the lanes (same positions: s_i by i successive additions of ds)
        const int *meta = reinterpret_cast<const int *>(cull + kCullWords);
        const int bx0 = meta[0], by0 = meta[1], bz0 = meta[2];
        const int bnx = meta[3] - bx0 + 1, bny = meta[4] - by0 + 1, bnz = meta[5] - bz0 + 1;
        const float len = sqrtf(m.dx * m.dx + m.dy * m.dy + m.dz * m.dz);
        const float ds = (2.0f / kCullRes) / fmaxf(len, 1e-12f);
        const float cw = 2.0f / kCullRes;
        float s0 = t0, s1 = far;
        bool may = !(bnx <= 0 || bny <= 0 || bnz <= 0);
        if (may) {
            const float lo[3] = {(float)bx0 * cw - 1.0f, (float)by0 * cw - 1.0f, (float)bz0 * cw - 1.0f};
            const float hi[3] = {(float)(bx0 + bnx) * cw - 1.0f, (float)(by0 + bny) * cw - 1.0f, (float)(bz0 + bnz) * cw - 1.0f};
            const float o[3] = {m.ox, m.oy, m.oz}, d[3] = {m.dx, m.dy, m.dz};
            #pragma unroll
            for (int a = 0; a < 3; a++) {
                if (d[a] != 0.0f) {
                    const float r = 1.0f / d[a];
                    const float ta = (lo[a] - o[a]) * r, tb = (hi[a] - o[a]) * r;
                    s0 = fmaxf(s0, fminf(ta, tb) - ds);
                    s1 = fminf(s1, fmaxf(ta, tb) + ds);
                } else if (o[a] < lo[a] - cw || o[a] > hi[a] + cw) {
                    may = false;
                }
            }
            if (!(s0 <= s1)) may = false;
        }
        if (!may) {
            go = false;
        } else {
            bool hit = false, ended = false;
            float base = s0;
            for (int round = 0; round < 2 && !ended; round++) {   // 96 scan positions: lanes 0..63, then 0..31
                float nb, unused_step;
                const float s = lane_lattice(base, ds, lane, nb, unused_step);
                base = nb;
                const bool in_range = round == 0 || lane < 32u;
                const float ss = fminf(s, s1);
                const float x = clampf_(m.ox + ss * m.dx, -1.0f, 1.0f), y = clampf_(m.oy + ss * m.dy, -1.0f, 1.0f), z = clampf_(m.oz + ss * m.dz, -1.0f, 1.0f);
                const int cx = (int)fminf((x + 1) * (0.5f * kCullRes), (float)(kCullRes - 1));
                const int cy = (int)fminf((y + 1) * (0.5f * kCullRes), (float)(kCullRes - 1));
                const int cz = (int)fminf((z + 1) * (0.5f * kCullRes), (float)(kCullRes - 1));
                const bool marked = in_range && cull_marked(cull, cx, cy, cz);
                // the sequential scan stops after the first position with s >= s1 (that position is still tested)
                const unsigned long long stop = __ballot(in_range && s >= s1);
                const uint32_t last = stop ? (uint32_t)__builtin_ctzll(stop) : (round == 0 ? 63u : 31u);
                const unsigned long long mk = __ballot(marked && lane <= last);
                if (mk) {
                    hit = true;
                    const uint32_t top = 63u - (uint32_t)__builtin_clzll(mk);
                    t_end = __shfl(ss + ds, (int)top, 64);
                }
                if (stop) ended = true;
            }
            if (!ended) { t_end = far; hit = true; }   // longer than 96 cells: "may hit, no early end"
            go = hit;
        }
    }
    uint32_t num_steps = 0;
    float *ts = sample_t + (size_t)n * max_steps;
    float base = t0;
    float carry_tt = -__FLT_MAX__;      // exit parameter of an empty voxel whose successor lies beyond the previous window
    // One window = 64 consecutive lattice points, one per lane.  A window's occupancy bytes are loaded while the window BEFORE it is
    // walked (the lattice does not depend on the walk), and for every lattice point, marked by the cull grid or not: the two
    // dependent loads per window of the first version (cull word, then occupancy byte: two L2 round trips with nothing to hide them
    // but three sibling waves doing the same) were the kernel's time.  An unmarked cull cell has no occupied voxel, so reading the
    // voxel's own bit gives the same answer.
    struct Window { float t, nb, cstep, x, y, z; int nx, ny, nz; uint32_t byte, bit; bool act; };
    auto fetch = [&](float from) __attribute__((always_inline)) {
        Window w;
        w.t = lane_lattice(from, dt, lane, w.nb, w.cstep);
        w.act = w.t < far && w.t < t_end;           // the loop condition of the sequential marcher at this lattice point
        // occupancy and (for an empty voxel) the exit parameter, with MarcherT<true>::probe's expressions
        w.x = clampf_(m.ox + w.t * m.dx, -bound, bound); w.y = clampf_(m.oy + w.t * m.dy, -bound, bound); w.z = clampf_(m.oz + w.t * m.dz, -bound, bound);
        w.nx = (int)clampf_((w.x + 1) * m.halfH, 0.0f, m.Hm1); w.ny = (int)clampf_((w.y + 1) * m.halfH, 0.0f, m.Hm1);
        w.nz = (int)clampf_((w.z + 1) * m.halfH, 0.0f, m.Hm1);
        const uint32_t index = morton3D_8bit((uint32_t)w.nx, (uint32_t)w.ny, (uint32_t)w.nz);
        w.byte = grid[index >> 3];                  // positions are clamped: a valid address for every lattice point
        w.bit = index & 7u;
        return w;
    };
    Window cur_w{};
    if (go) cur_w = fetch(base);
    while (go && num_steps < max_steps) {
        const Window nxt_w = fetch(cur_w.nb);
        const float t = cur_w.t, nb = cur_w.nb, cstep = cur_w.cstep, x = cur_w.x, y = cur_w.y, z = cur_w.z;
        const int nx = cur_w.nx, ny = cur_w.ny, nz = cur_w.nz;
        const bool act = cur_w.act;
        const bool occ = act && ((cur_w.byte >> cur_w.bit) & 1u);
        const float tx = ((((float)nx + m.ex) * m.twoRH - 1) - x) * m.rdx;
        const float ty = ((((float)ny + m.ey) * m.twoRH - 1) - y) * m.rdy;
        const float tz = ((((float)nz + m.ez) * m.twoRH - 1) - z) * m.rdz;
        const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        const unsigned long long occ_m = __ballot(act && occ), act_m = __ballot(act);
        // successor of an empty lane inside this window: the first later lattice point >= tt (at least one step), else 64
        // (a 128^3 voxel is crossed in at most 8 steps of dt_min; longer crossings walk on)
        uint32_t nxt = lane + 1u;
        const bool empty = act && !occ;
        if (cstep > 0.0f) {
            // closed-form window (wave-uniform): the lattice values are base + j cstep exactly, so the successor is an index
            // computation per lane -- an estimate by division, then exact comparisons against the lattice values themselves
            if (empty) {
                const float q = (tt - base) / cstep;
                int j = q < 64.0f ? (int)ceilf(q) : 64;
                if (j < (int)lane + 1) j = (int)lane + 1;
                while (j > (int)lane + 1 && base + (float)(j - 1) * cstep >= tt) j--;
                while (j < 64 && base + (float)j * cstep < tt) j++;
                nxt = (uint32_t)j;
            }
        } else {
            for (int k = 0; k < 64; k++) {
                const float lv = __shfl(t, (int)(nxt & 63u), 64);
                const bool more = empty && nxt < 64u && lv < tt;
                if (!__any(more)) break;
                if (more) nxt++;
            }
        }
        // entry point of the chain into this window
        uint32_t cur0 = 0;
        float new_carry = -__FLT_MAX__;
        if (carry_tt != -__FLT_MAX__) {
            const unsigned long long ge = __ballot(t >= carry_tt);
            cur0 = ge ? (uint32_t)__builtin_ctzll(ge) : 64u;
            if (!ge) new_carry = carry_tt;      // the voxel's exit lies beyond this window too: keep walking
        }
        // The walk itself is a SCALAR loop: position, masks, budget and the emitted set live in SGPRs (pinned with readfirstlane --
        // the compiler's uniformity analysis does not see through the loop-carried values and otherwise runs the loop as divergent
        // vector code under exec masks: measured 440 cycles per hop, 55 000 of a long ray's 81 000 cycles).
        uint32_t cur = __builtin_amdgcn_readfirstlane(cur0);
        uint32_t budget = __builtin_amdgcn_readfirstlane(max_steps - num_steps);
        uint32_t emit_lo = 0, emit_hi = 0;
        bool done = false;
        while (cur < 64u) {
            const unsigned long long am = act_m >> cur, om = occ_m >> cur;
            if (!(am & 1ull)) { done = true; break; }                      // t >= far or t >= t_end: the ray is finished
            if (om & 1ull) {
                // a run of occupied lattice points: every one is visited and sampled
                const unsigned long long rest = ~om;
                uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : 64u - cur;
                if (run > 64u - cur) run = 64u - cur;
                if (run >= budget) { run = budget; done = true; }
                const unsigned long long bits = ((run >= 64u) ? ~0ull : ((1ull << run) - 1ull)) << cur;
                emit_lo = __builtin_amdgcn_readfirstlane(emit_lo | (uint32_t)bits);
                emit_hi = __builtin_amdgcn_readfirstlane(emit_hi | (uint32_t)(bits >> 32));
                budget = __builtin_amdgcn_readfirstlane(budget - run);
                cur = __builtin_amdgcn_readfirstlane(cur + run);
                if (done) break;
            } else {
                // (v_readlane with a scalar lane index, not an LDS-latency ds_bpermute)
                const uint32_t to = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)cur);
                if (to >= 64u) new_carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tt), (int)cur));
                cur = to;
            }
        }
        const unsigned long long emit = ((unsigned long long)emit_hi << 32) | emit_lo;
        if ((emit >> lane) & 1ull) ts[num_steps + (uint32_t)__popcll(emit & ((1ull << lane) - 1ull))] = t;
        num_steps = __builtin_amdgcn_readfirstlane(num_steps + (uint32_t)__popcll(emit));
        if (done) break;
        carry_tt = new_carry;
        base = nb;
        cur_w = nxt_w;
    }
    if (lane == 0) num_steps_out[n] = num_steps;
}

// Block-level inclusive scan of one uint32 per thread (1024 threads = 16 waves): DPP-free,
// __shfl_up based wave scan + LDS carry.
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t *lds /*[16]*/, uint32_t &block_total) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    #pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off, 64);
        if (lane >= (uint32_t)off) v += u;
    }
    if (lane == 63) lds[wid] = v;
    __syncthreads();
    uint32_t carry = 0, total = 0;
    const uint32_t nw = blockDim.x >> 6;
    for (uint32_t w = 0; w < nw; w++) {
        const uint32_t s = lds[w];
        if (w < wid) carry += s;
        total += s;
    }
    block_total = total;
    __syncthreads();
    return v + carry;
}

// pass A: per-block totals
__global__ void __launch_bounds__(kScanBlock) k_scan_block_totals(const uint32_t *__restrict__ vals, uint32_t N, uint32_t *__restrict__ block_totals) {
    __shared__ uint32_t lds[16];
    const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
    uint32_t total;
    block_inclusive_scan(i < N ? vals[i] : 0u, lds, total);
    if (threadIdx.x == 0) block_totals[blockIdx.x] = total;
}

// pass B: offsets[i] = exclusive scan (every block re-sums the totals of the blocks before it:
// <= 625 L2-resident words for a full 800x800 frame); writes rays[] and bumps counter[] like
// the reference's atomics would, but in ray order.
__global__ void __launch_bounds__(kScanBlock) k_march_train_offsets(const uint32_t *__restrict__ num_steps, uint32_t N,
                                                                    const uint32_t *__restrict__ block_totals,
                                                                    int32_t *__restrict__ rays, int32_t *__restrict__ counter,
                                                                    uint32_t *__restrict__ base_out) {
    __shared__ uint32_t lds[16];
    __shared__ uint32_t s_prev;
    // sum of previous blocks' totals, cooperatively
    uint32_t part = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += kScanBlock) part += block_totals[b];
    uint32_t prev_total;
    block_inclusive_scan(part, lds, prev_total);
    if (threadIdx.x == 0) s_prev = prev_total;
    __syncthreads();
    const uint32_t base = (uint32_t)counter[0];  // reference semantics: slots start at the counter's current value
    const uint32_t ray_base = (uint32_t)counter[1];
    const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
    const uint32_t v = i < N ? num_steps[i] : 0u;
    uint32_t total;
    const uint32_t incl = block_inclusive_scan(v, lds, total);
    if (i < N) {
        const uint32_t ri = ray_base + i;
        rays[ri * 3] = (int32_t)i;
        rays[ri * 3 + 1] = (int32_t)(base + s_prev + incl - v);
        rays[ri * 3 + 2] = (int32_t)v;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        base_out[0] = base;  // consumed by the write pass; counter itself is bumped by k_march_train_finish
        base_out[1] = s_prev + total;
        base_out[2] = ray_base;
    }
}

// A training batch (N of a few thousand rays) does not need the two-pass scan: ONE workgroup walks the rays in chunks of its size with
// a running carry, writes the ray records and bumps the counters itself (the emit pass takes its bases from base_out) -- three
// launches (block totals, offsets, finish) become one.
__global__ void __launch_bounds__(kScanBlock) k_march_train_offsets_small(const uint32_t *__restrict__ num_steps, uint32_t N, int32_t *__restrict__ rays,
                                                                          int32_t *__restrict__ counter, uint32_t *__restrict__ base_out) {
    __shared__ uint32_t lds[16];
    const uint32_t base = (uint32_t)counter[0], ray_base = (uint32_t)counter[1];
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < N; c0 += kScanBlock) {
        const uint32_t i = c0 + threadIdx.x;
        const uint32_t v = i < N ? num_steps[i] : 0u;
        uint32_t total;
        const uint32_t incl = block_inclusive_scan(v, lds, total);
        if (i < N) {
            const uint32_t ri = ray_base + i;
            rays[ri * 3] = (int32_t)i;
            rays[ri * 3 + 1] = (int32_t)(base + carry + incl - v);
            rays[ri * 3 + 2] = (int32_t)v;
        }
        carry += total;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        base_out[0] = base; base_out[1] = carry; base_out[2] = ray_base;
        counter[0] = (int32_t)(base + carry);
        counter[1] = (int32_t)(ray_base + N);
    }
}

__global__ void k_march_train_finish(int32_t *__restrict__ counter, const uint32_t *__restrict__ base_out, uint32_t N) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        counter[0] += (int32_t)base_out[1];
        counter[1] += (int32_t)N;
    }
}

// Emit pass: one wave per ray, one lane per sample.  From the recorded t of a sample the marcher's own expressions give its
// position (clamp(o + t d)), its step dt = step_size(t) and the parameter after it, t + dt; delta[1] is the difference of two
// consecutive "after" parameters (the first one against the perturbed start) -- bit for bit what the sequential loop wrote
// (raymarching.cu:427-479), but coalesced and without marching again.
template <bool FAST>
__global__ void __launch_bounds__(256) k_march_train_emit(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float bound,
                                                          float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                          const float *__restrict__ nears, const float *__restrict__ noises,
                                                          const int32_t *__restrict__ rays, const uint32_t *__restrict__ bases,
                                                          const float *__restrict__ sample_t, float *__restrict__ xyzs,
                                                          float *__restrict__ dirs, float *__restrict__ deltas) {
    const uint32_t n = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (n >= N) return;
    const uint32_t ray_base = bases[2];              // the ray counter's value before this call (the offsets pass recorded it)
    const uint32_t point_index = (uint32_t)rays[(size_t)(ray_base + n) * 3 + 1];
    const uint32_t num_steps = (uint32_t)rays[(size_t)(ray_base + n) * 3 + 2];
    if (num_steps == 0 || point_index + num_steps > M) return;
    MarcherT<FAST> m;
    m.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, nullptr);
    float t0 = nears[n];
    t0 += m.step_size(t0) * noises[n];
    const float *ts = sample_t + (size_t)n * max_steps;
    for (uint32_t k = lane; k < num_steps; k += 64u) {
        const float t = ts[k];
        const float dt = m.step_size(t);
        const float after = t + dt;
        float last = t0;
        if (k > 0) { const float tp = ts[k - 1]; last = tp + m.step_size(tp); }
        const size_t p = (size_t)point_index + k;
        xyzs[p * 3] = clampf_(m.ox + t * m.dx, -bound, bound);
        xyzs[p * 3 + 1] = clampf_(m.oy + t * m.dy, -bound, bound);
        xyzs[p * 3 + 2] = clampf_(m.oz + t * m.dz, -bound, bound);
        dirs[p * 3] = m.dx; dirs[p * 3 + 1] = m.dy; dirs[p * 3 + 2] = m.dz;
        deltas[p * 2] = dt;
        deltas[p * 2 + 1] = after - last;
    }
}

// raymarching.cu:501-577
__global__ void __launch_bounds__(256) k_composite_train_fwd(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                             const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                                                             uint32_t M, uint32_t N, float T_thresh, float *__restrict__ weights_sum,
                                                             float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) {
        weights_sum[index] = 0; depth[index] = 0;
        image[(size_t)index * 3] = 0; image[(size_t)index * 3 + 1] = 0; image[(size_t)index * 3 + 2] = 0;
        return;
    }
    const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
    uint32_t step = 0;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    while (step < num_steps) {
        const float alpha = 1.0f - sdn_exp_cr(-s[0] * dl[0]);
        const float weight = alpha * T;
        r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
        t += dl[1];
        d += weight * t;
        ws += weight;
        T *= 1.0f - alpha;
        if (T < T_thresh) break;
        s++; c += 3; dl += 2; step++;
    }
    weights_sum[index] = ws; depth[index] = d;
    image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
}

// The INFERENCE compositing arithmetic (kernel_composite_rays, raymarching.cu:819-905: transmittance as 1 - weights_sum, the stop
// test on the transmittance IN FRONT of a sample, t running from the ray's near bound) over ALL samples of a ray at once, in the
// (offset, count) layout of march_rays_train.  With a ray's samples listed up front the iteration loop of the inference branch
// (dnerf/renderer.py:333-381) collapses into march -> field -> this kernel: per ray the same additions in the same order, so image and
// weights_sum are those of the loop bit for bit; the loop only exists to stop marching terminated rays early, which pays for a
// frame of 640 000 rays and costs a chain of ~30 dependent launches for a batch of 4 096.
__global__ void __launch_bounds__(256) k_composite_whole_rays(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                              const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                                                              const float *__restrict__ nears, uint32_t M, uint32_t N, float T_thresh,
                                                              float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    float weight_sum = 0, d = 0, r = 0, g = 0, b = 0;
    if (num_steps != 0 && offset + num_steps <= M) {
        const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
        float t = nears[index];
        for (uint32_t step = 0; step < num_steps; step++) {
            if (dl[0] == 0) break;
            const float alpha = 1.0f - sdn_exp_cr(-s[0] * dl[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t += dl[1];
            d += weight * t;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            if (T < T_thresh) break;
            s++; c += 3; dl += 2;
        }
    }
    weights_sum[index] = weight_sum; depth[index] = d;
    image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
}

// raymarching.cu:602-682
__global__ void __launch_bounds__(256) k_composite_train_bwd(const float *__restrict__ grad_weights_sum, const float *__restrict__ grad_image,
                                                             const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                             const float *__restrict__ deltas, const int32_t *__restrict__ rays,
                                                             const float *__restrict__ weights_sum, const float *__restrict__ image,
                                                             uint32_t M, uint32_t N, float T_thresh, float *__restrict__ grad_sigmas,
                                                             float *__restrict__ grad_rgbs) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float gws = grad_weights_sum[index];
    const float gi0 = grad_image[(size_t)index * 3], gi1 = grad_image[(size_t)index * 3 + 1], gi2 = grad_image[(size_t)index * 3 + 2];
    const float r_final = image[(size_t)index * 3], g_final = image[(size_t)index * 3 + 1], b_final = image[(size_t)index * 3 + 2];
    const float ws_final = weights_sum[index];
    const float *s = sigmas + offset, *c = rgbs + (size_t)offset * 3, *dl = deltas + (size_t)offset * 2;
    float *gs = grad_sigmas + offset, *gc = grad_rgbs + (size_t)offset * 3;
    uint32_t step = 0;
    float T = 1.0f, r = 0, g = 0, b = 0;
    while (step < num_steps) {
        const float alpha = 1.0f - sdn_exp_cr(-s[0] * dl[0]);
        const float weight = alpha * T;
        r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
        T *= 1.0f - alpha;
        gc[0] = gi0 * weight; gc[1] = gi1 * weight; gc[2] = gi2 * weight;
        gs[0] = dl[0] * (gi0 * (T * c[0] - (r_final - r)) + gi1 * (T * c[1] - (g_final - g)) +
                         gi2 * (T * c[2] - (b_final - b)) + gws * (1 - ws_final));
        if (T < T_thresh) break;
        s++; c += 3; dl += 2; gs++; gc += 3; step++;
    }
}

// ---------------------------------------------------------------------------
// inference
// ---------------------------------------------------------------------------
// raymarching.cu:701-805.  Extras over the reference kernel (all optional, none changes a sample):
//   * slots [step, n_step) of each alive ray are written as zeros here and the alignment tail [n_alive*n_step, M_pad)
//     is cleared by the same launch, so the caller never memsets the sample buffers;
//   * cull: exact early-out for rays that cannot produce a sample (see ray_may_hit);
//   * live_idx / live_count: the slots that received a sample are appended (one atomicAdd per wave, ballot-free prefix
//     via wave shuffles) to a compact list so that the field network is evaluated on samples, not on padded slots.
// Marches one ray from parameter t for up to n_step samples into its slots (the loop of raymarching.cu:750-804), zero-fills
// the unused slots and returns the number of samples written.
// tend (may be null): this ray's cached cull result.  The bound t_end found by ray_may_hit -- beyond it the ray meets no
// marked cell -- is a property of the ray, not of where the scan started, so the device loop computes it on a ray's first
// march and later iterations only compare (kTendUnset: not computed yet; kTendDead: the ray cannot produce a sample).
constexpr float kTendUnset = -2.0f, kTendDead = -1.0f;
// device loop: the cache's base pointer travels in the loop record (state[13], state[14]; 0 = no cache)
__device__ __forceinline__ float *state_tend(const int32_t *__restrict__ state) {
    return reinterpret_cast<float *>(((unsigned long long)(uint32_t)state[14] << 32) | (uint32_t)state[13]);
}
template <bool FAST>
__device__ __forceinline__ uint32_t march_ray(MarcherT<FAST> &m, const OccCache &oc, float t, float far, uint32_t n_step, float *px, float *pd,
                                              float *pl, float *tend = nullptr, uint8_t *psf = nullptr, uint32_t frame = 0,
                                              const float *t_begin = nullptr) {
    m.fine = oc.fine; m.fx0 = oc.fx0; m.fy0 = oc.fy0; m.fz0 = oc.fz0; m.fnx = oc.fnx; m.fny = oc.fny; m.fnz = oc.fnz;
    uint32_t step = 0;
    // t_begin: the walk resumes at t, a later point of the chain that started at *t_begin (k_cull_start's certified jump): the first
    // sample's second delta is measured from the chain's start, as the walk from there would have measured it
    float last_t = t_begin ? *t_begin : t, x, y, z, dt;
    bool go = t < far;
    float t_end = far;
    if (FAST && oc.s_cull && go) {
        const float cached = tend ? *tend : kTendUnset;
        if (cached != kTendUnset) {
            t_end = cached;                 // kTendDead (< every t) ends the ray here
            go = t < t_end;
        } else {
            go = ray_may_hit(oc.s_cull, m.ox, m.oy, m.oz, m.dx, m.dy, m.dz, t, far, t_end, oc.fx0, oc.fy0, oc.fz0, oc.fnx, oc.fny, oc.fnz);
            if (tend) *tend = go ? t_end : kTendDead;
        }
    }
    if (go) {
        while (t < far && t < t_end && step < n_step) {
            if (m.probe(t, x, y, z, dt, oc.s_cull)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
                t += dt;
                pl[0] = dt;
                pl[1] = t - last_t;
                last_t = t;
                px += 3; pd += 3; pl += 2;
                if (psf) psf[step] = (uint8_t)frame;   // frame group: which frame's time constants the field network applies
                step++;
            }
        }
    }
    for (uint32_t k = step; k < n_step; k++) {
        px[0] = 0; px[1] = 0; px[2] = 0;
        pd[0] = 0; pd[1] = 0; pd[2] = 0;
        pl[0] = 0; pl[1] = 0;
        px += 3; pd += 3; pl += 2;
    }
    return step;
}

// march_ray with a probe budget (FAST configuration with LDS caches; the caller has resolved the cull bound t_end): at most
// max_probes loop-body evaluations.  Returns the samples written; `more` = the budget ran out before the ray was finished -- t then is
// the lattice point the chain stands at, last_t the parameter behind its last sample, and the unused slots are NOT zero-filled (the
// cooperative marcher continues the ray).  A finished ray is exactly march_ray's result.
__device__ __forceinline__ uint32_t march_ray_some(MarcherT<true> &m, const OccCache &oc, float &t, float far, float t_end, uint32_t n_step,
                                                   float *px, float *pd, float *pl, uint8_t *psf, uint32_t frame, uint32_t max_probes, float &last_t,
                                                   bool &more) {
    m.fine = oc.fine; m.fx0 = oc.fx0; m.fy0 = oc.fy0; m.fz0 = oc.fz0; m.fnx = oc.fnx; m.fny = oc.fny; m.fnz = oc.fnz;
    uint32_t step = 0, probes = 0;
    float x, y, z, dt;
    last_t = t;
    more = false;
    while (t < far && t < t_end && step < n_step) {
        if (probes == max_probes) { more = true; break; }
        probes++;
        if (m.probe(t, x, y, z, dt, oc.s_cull)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
            t += dt;
            pl[0] = dt;
            pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2;
            if (psf) psf[step] = (uint8_t)frame;
            step++;
        }
    }
    if (!more) {
        for (uint32_t k = step; k < n_step; k++) {
            px[0] = 0; px[1] = 0; px[2] = 0;
            pd[0] = 0; pd[1] = 0; pd[2] = 0;
            pl[0] = 0; pl[1] = 0;
            px += 3; pd += 3; pl += 2;
        }
    }
    return step;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Cooperative marcher: kGL lanes per ray (FAST configuration, constant step, LDS occupancy caches).
//
// One lane per ray walks a dependent chain of ~1000 cycles per voxel probe, and a launch lasts as long as its longest chain
// (profiles/r03_marcher_pmc_summary.json: waves live 20 K cycles on average in a launch of 60 K, 61 % of their cycles waiting, 46 %
// of the VALU lanes active).  Here a ray is marched by a GROUP of kGL = 16 consecutive lanes: with a constant step every parameter
// the marcher visits is a point of one lattice L_0 = t, L_{k+1} = fl(L_k + dt) (see k_march_train_count_wave), so the group takes
// a window of 16 consecutive lattice points -- one per lane, by the closed form L_k = base + k c while the window stays inside one
// binade, by the recurrence otherwise --, evaluates position, voxel, occupancy bit and the voxel's exit parameter in parallel with
// the sequential marcher's own expressions (MarcherT<true>::locate / occupied / exit_t), derives every empty lane's successor
// (first later lattice point >= its exit parameter) and walks the chain  next(k) = occupied ? k + 1 : successor(k)  over the
// window with group-local shuffles.  Visited set, samples, step lengths and counts are the sequential chain's bit for bit; only
// the schedule changes: 16 lattice points (3-4 voxels) per round instead of one voxel per probe, 16x the waves to hide latency
// behind, no lane idling while its neighbour crosses empty space.
//
// Fresh rays (first march of a frame) do not walk the chain from the cube's entry point to the figure (~100 empty voxels): the
// group jumps along the lattice (closed form, binade by binade) to a window shortly before the parameter t_safe up to which the
// cull scan saw only unmarked cells (every point there is >= 2 fine voxels from any occupied voxel: nothing can be sampled), and
// RE-SYNCHRONISES with the chain exactly: from a visited point v of an empty run (= the lattice points of one voxel) the chain
// goes to the first lattice point >= exit(v), i.e. the start a of the next run -- unless a lattice point lies within rounding of
// the voxel boundary.  If EVERY lattice point i of a complete run R satisfies exit(L_i) <= L_a, the chain -- which enters R or
// lands on a directly from the run before, an overshoot being at most one lattice point -- is at a whichever points it visited
// before.  The group verifies that for the first complete run of its window (all of the run's lanes, not just the visited
// one) and starts the walk at a; if the check fails (a lattice point within ~1e-7 of a voxel face: ~3e-5 per run) it tries the
// next run, and gives up to the walk from the ray's own start after that.  Nothing is approximated.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kGL = 16;

#ifdef SDN_STAMPS
// Diagnostic build only (make diag; the product library has no stamp): shader-clock sums per phase of the cooperative kernels,
// accumulated by thread 0 of every workgroup.  [0..7] k_composite_march_g, [8..15] k_march_rays_g:
//   +0 workgroups with work  +1 phase A (entry -> task list complete)  +2 fine-bit cache  +3 phase B (all passes)  +4 live-list append
//   +5 tasks  +6 windows (rounds of group_march)  +7 slow-form windows
__device__ unsigned long long g_stamps[32];
#define SDN_STAMP_NOW() __builtin_amdgcn_s_memtime()
#define SDN_STAMP_ADD(i, v) atomicAdd(&g_stamps[i], (unsigned long long)(v))
#else
#define SDN_STAMP_NOW() 0ull
#define SDN_STAMP_ADD(i, v) ((void)(v))
#endif

// L_g of the lattice starting at `base` for the kGL lanes of a group, and L_kGL (the next window's base).  cstep > 0: closed form.
__device__ __forceinline__ float group_lattice(float base, float dt, uint32_t g, float &next_base, float &cstep) {
    {
        const float first = base + dt;
        const float c = first - base;                               // exact: both multiples of the binade's ulp U
        const float err = dt - c;                                   // rounding error of base + dt (exact for base >= dt)
        const float last = base + (float)kGL * c;
        const uint32_t eb = __float_as_uint(base) >> 23, el = __float_as_uint(last) >> 23;     // sign bit clear: t > 0
        const bool normal = base >= dt && dt > 0.0f && eb > 30u && eb < 255u;
        const float half_ulp = __uint_as_float((eb - 24u) << 23), c_cap = __uint_as_float((eb - 5u) << 23);   // U / 2, 2^18 U
        if (normal && eb == el && fabsf(err) != half_ulp && c < c_cap) {
            next_base = last;
            cstep = c;                                              // L_k = base + k c holds for k = 0 .. kGL
            return base + (float)g * c;
        }
    }
    cstep = 0.0f;
    float v = base, mine = base;
    for (uint32_t i = 0; i < kGL; i++) {
        if (i == g) mine = v;
        v += dt;
    }
    next_base = v;
    return mine;
}

// The largest lattice point <= target of the lattice t, fl(t + dt), ... (t itself if none): closed form inside a binade, one
// recurrence step across a binade boundary.  Gives up (returns the point reached) where the closed form does not hold.
__device__ __forceinline__ float lattice_floor(float t, float dt, float target) {
    for (int hop = 0; hop < 4; hop++) {
        const float first = t + dt;
        if (!(first <= target)) break;
        const float c = first - t;
        const float err = dt - c;
        const uint32_t eb = __float_as_uint(t) >> 23;
        const bool normal = t >= dt && dt > 0.0f && eb > 30u && eb < 254u;
        const float half_ulp = __uint_as_float((eb - 24u) << 23), c_cap = __uint_as_float((eb - 5u) << 23);
        if (!(normal && fabsf(err) != half_ulp && c < c_cap)) break;
        const float top = __uint_as_float(((eb + 1u) << 23) - 1u);       // the largest float of t's binade
        const float lim = fminf(target, top);
        float k = floorf((lim - t) / c);
        while (t + (k + 1.0f) * c <= lim) k += 1.0f;                       // (products and sums exact: multiples of U below 2^24 U)
        while (k > 0.0f && t + k * c > lim) k -= 1.0f;
        t = t + k * c;
        if (lim == target) break;
        if (!(t + dt <= target)) break;
        t = t + dt;                                                       // the one step that crosses into the next binade
    }
    return t;
}

// lane g - 1's value within the group (a group is one DPP row of 16 lanes): row_shr:1; lane 0 of the group gets 0
__device__ __forceinline__ uint32_t group_prev_lane(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
}

// Re-synchronisation with the sequential chain inside a window of kGL EMPTY lattice points starting at `cand` (all of them before
// the ray's end): looks for a complete run (the lattice points of one voxel) all of whose points i satisfy exit(L_i) <= L_a, a the
// start of the next run; the chain is then certainly at a, whatever it visited before (see the header comment).  Returns false if
// the window is not a closed-form window or no run qualifies.
__device__ __forceinline__ bool group_resync(const MarcherT<true> &m, const uint32_t *s_cull, float cand, uint32_t g, uint32_t gbase, float &found_t) {
    float nb, cstep;
    const float t = group_lattice(cand, m.dt_const, g, nb, cstep);
    if (!(cstep > 0.0f)) return false;                                                     // group-uniform
    float x, y, z;
    int nx, ny, nz;
    m.locate(t, x, y, z, nx, ny, nz);
    const bool occ = m.occupied(nx, ny, nz, s_cull);
    if ((uint32_t)(__ballot(occ) >> gbase) & 0xFFFFu) return false;                        // the caller saw none; be safe
    const float tt = m.exit_t(t, x, y, z, nx, ny, nz);
    const uint32_t vox = (uint32_t)nx | ((uint32_t)ny << 8) | ((uint32_t)nz << 16);
    const uint32_t vprev = group_prev_lane(vox);
    const uint32_t chg = (uint32_t)(__ballot(g > 0u && vox != vprev) >> gbase) & 0xFFFFu;  // bit k: lane k starts a run
    const uint32_t above = chg & ~((2u << g) - 1u);                                        // run starts after my lane
    const bool has_next = above != 0u;
    const uint32_t a = has_next ? (uint32_t)__builtin_ctz(above) : kGL;
    const bool bad = !has_next || !(tt <= cand + (float)a * cstep);
    const uint32_t badm = (uint32_t)(__ballot(bad) >> gbase) & 0xFFFFu;
    uint32_t rest = chg;                    // complete runs: [c_j, c_{j+1}) for consecutive set bits; the first whose lanes all pass
    while (rest) {
        const uint32_t cj = (uint32_t)__builtin_ctz(rest);
        rest &= rest - 1u;
        if (!rest) break;
        const uint32_t cn = (uint32_t)__builtin_ctz(rest);
        const uint32_t runm = ((1u << cn) - 1u) & ~((1u << cj) - 1u);
        if (!(badm & runm)) { found_t = cand + (float)cn * cstep; return true; }
    }
    return false;
}

// Marches one ray by the kGL lanes [gbase, gbase + kGL) of the wave; every lane of the group calls it with the same arguments
// (g = its index in the group).  Same contract as march_ray: writes up to n_step samples, zero-fills the unused slots, returns
// the number written.  t_safe: see ray_may_hit (-FLT_MAX: no jump).  The ray may have been started by march_ray_some: t0 is then
// the point its chain stopped at (a visited lattice point), step0 / last_t0 its sample count and the parameter behind its last sample.
//
// The state between rounds is ONE number: `exact_base`, a lattice point the sequential chain certainly visits.  Rounds:
// * SCAN: only the occupancy bits of a window are evaluated; while none is set the chain cannot sample, so the window is passed
//   without knowing which of its points the chain visits.  When a scan meets an occupied lattice point the group re-synchronises in
//   the (empty) window before it (group_resync) and goes on exactly from there; if that fails, exactly from the last known point.
// * EXACT, fast form (a closed-form window that starts at a visited point): the chain visits the start of every run and every
//   occupied point PROVIDED no empty point i of a complete run has exit(L_i) > L_a, a the start of the next run -- checked for all
//   of the run's lanes at once (one DPP move for the neighbour's voxel, bit operations for a, one compare).  Then the samples are
//   simply the occupied lattice points (up to the budget), and the next window starts at the first point after the window if its
//   last point is occupied, else at the start of its last (empty) run, which the chain visits.  No successor indices, no walk.
// * EXACT, slow form (a check failed: a lattice point within rounding of a voxel face, ~3e-5 per run; or a window across a binade
//   boundary): successor indices and the walk over the window, then the chain's next point by the marcher's own `t += dt` loop.
__device__ __forceinline__ uint32_t group_march(const MarcherT<true> &m, const uint32_t *s_cull, float t0, float far, float t_end, uint32_t n_step,
                                                float *px, float *pd, float *pl, uint8_t *psf, uint32_t frame, uint32_t g, uint32_t gbase,
                                                float t_safe, uint32_t step0, float last_t0, uint32_t *win_ctr = nullptr) {
    const float dt = m.dt_const;
    const uint32_t sh = gbase;
    uint32_t step = step0;                                   // samples the lane-per-ray phase already wrote (0: a fresh start)
    float last_t = last_t0;                                  // parameter behind the previous sample (the iteration's start: t0)
    float exact_base = t0;                                   // a lattice point the chain visits
    float scan_base = t0, prev_base = 0.0f, force_until = -__FLT_MAX__;
    bool scanning = true, prev_valid = false;
    const bool go = t0 < far && t0 < t_end;
    if (go && t_safe > t0 + 48.0f * dt)                      // fresh ray: jump along the lattice towards the first marked cull cell
        scan_base = lattice_floor(t0, dt, fminf(t_safe, fminf(far, t_end)) - 18.0f * dt);
    while (go) {
        const float wbase = scanning ? scan_base : exact_base;
        float nb, cstep;
        const float t = group_lattice(wbase, dt, g, nb, cstep);
        const bool act = t < far && t < t_end;                  // the loop condition of the sequential marcher at this lattice point
        float x, y, z;
        int nx, ny, nz;
        m.locate(t, x, y, z, nx, ny, nz);
        const bool occ = act && m.occupied(nx, ny, nz, s_cull);
        const uint32_t am = (uint32_t)(__ballot(act) >> sh) & 0xFFFFu, om = (uint32_t)(__ballot(occ) >> sh) & 0xFFFFu;
#ifdef SDN_STAMPS
        if (g == 0u && win_ctr) atomicAdd(win_ctr, 1u);
#endif
        if (scanning) {
            if (om == 0u) {
                if (am != 0xFFFFu) break;                        // the ray ends inside this window without another sample
                prev_base = wbase; prev_valid = cstep > 0.0f;
                scan_base = nb;
                continue;
            }
            scanning = false;
            if (wbase != exact_base) {                           // an occupied lattice point ahead and the chain's points are not known here
                float found;
                if (prev_valid && group_resync(m, s_cull, prev_base, g, gbase, found)) exact_base = found;
                else force_until = wbase;                        // walk exactly from the last known point up to this window
                continue;
            }
        }
        const float tt = m.exit_t(t, x, y, z, nx, ny, nz);
        const bool empty = act && !occ;
        const uint32_t budget = n_step - step;
        uint32_t emit = 0;
        bool done = false;
        float next_base = nb;
        // ---- fast form ---------------------------------------------------------------------------------------------------------
        bool fast = cstep > 0.0f;
        uint32_t chg = 0;
        if (fast) {
            const uint32_t vox = (uint32_t)nx | ((uint32_t)ny << 8) | ((uint32_t)nz << 16);
            const uint32_t vprev = group_prev_lane(vox);
            chg = (uint32_t)(__ballot(g > 0u && vox != vprev) >> sh) & 0xFFFFu;           // bit k: lane k starts a run
            const uint32_t above = chg & ~((2u << g) - 1u);
            const uint32_t a = above ? (uint32_t)__builtin_ctz(above) : kGL;
            const bool bad = empty && above && !(tt <= wbase + (float)a * cstep);
            if ((uint32_t)(__ballot(bad) >> sh) & 0xFFFFu) fast = false;
            // a window that ends inside an empty run continues at that run's start: it must not be the window's own start
            if (fast && am == 0xFFFFu && !((om >> (kGL - 1u)) & 1u) && chg == 0u) fast = false;
        }
        if (fast) {
            emit = om;
            const uint32_t cnt = (uint32_t)__builtin_popcount(om);
            if (cnt >= budget) {
                for (uint32_t k = cnt; k > budget; k--) emit &= ~(1u << (31u - (uint32_t)__builtin_clz(emit)));   // the first `budget` of them
                done = true;
            } else if (am != 0xFFFFu) {
                done = true;                                     // the chain runs into the ray's end inside this window
            } else if (!((om >> (kGL - 1u)) & 1u)) {
                next_base = wbase + (float)(31u - (uint32_t)__builtin_clz(chg)) * cstep;   // start of the trailing empty run
            }
        } else {
            // ---- slow form: successor indices + walk (k_march_train_count_wave's scheme on kGL lanes) --------------------------------
#ifdef SDN_STAMPS
            if (g == 0u && win_ctr) atomicAdd(win_ctr + 1, 1u);
#endif
            uint32_t nxt = g + 1u;
            if (cstep > 0.0f) {
                if (empty) {
                    const float q = (tt - wbase) / cstep;
                    int j = q < (float)kGL ? (int)ceilf(q) : (int)kGL;
                    if (j < (int)g + 1) j = (int)g + 1;
                    while (j > (int)g + 1 && wbase + (float)(j - 1) * cstep >= tt) j--;
                    while (j < (int)kGL && wbase + (float)j * cstep < tt) j++;
                    nxt = (uint32_t)j;
                }
            } else {
                for (uint32_t k = 0; k < kGL; k++) {
                    const float lv = __shfl(t, (int)(gbase + (nxt & (kGL - 1u))), 64);
                    if (empty && nxt < kGL && lv < tt) nxt++;
                }
            }
            uint32_t cur = 0, left = budget;
            float pending = -__FLT_MAX__;                        // exit parameter of an empty voxel whose successor lies beyond the window
            while (!done && cur < kGL) {
                if (!((am >> cur) & 1u)) { done = true; break; }                  // t >= far or t >= t_end: the ray is finished
                const uint32_t o = om >> cur;
                if (o & 1u) {
                    // a run of occupied lattice points: every one is visited and sampled
                    uint32_t run = (uint32_t)__builtin_ctz(~o);
                    if (run >= left) { run = left; done = true; }
                    emit |= ((1u << run) - 1u) << cur;
                    left -= run;
                    cur += run;
                } else {
                    const uint32_t to = (uint32_t)__shfl((int)nxt, (int)(gbase + cur), 64);
                    if (to >= kGL) pending = __shfl(tt, (int)(gbase + cur), 64);
                    cur = to;
                }
            }
            if (!done && pending != -__FLT_MAX__) {
                while (next_base < pending) next_base += dt;     // `do t += dt while (t < tt)`, continued beyond the window
            }
        }
        if (emit) {
            const bool mine = (emit >> g) & 1u;
            const uint32_t below = emit & ((1u << g) - 1u);
            const float t_after = t + dt;                                      // `t += dt` of the sequential marcher
            const uint32_t prev = below ? 31u - (uint32_t)__builtin_clz(below) : 0u, top = 31u - (uint32_t)__builtin_clz(emit);
            float prev_after, top_after;
            if (cstep > 0.0f) {                                                // the other lanes' values by the closed form: no shuffle
                prev_after = (wbase + (float)prev * cstep) + dt;
                top_after = (wbase + (float)top * cstep) + dt;
            } else {
                prev_after = __shfl(t_after, (int)(gbase + prev), 64);
                top_after = __shfl(t_after, (int)(gbase + top), 64);
            }
            if (mine) {
                const uint32_t slot = step + (uint32_t)__builtin_popcount(below);
                px[slot * 3] = x; px[slot * 3 + 1] = y; px[slot * 3 + 2] = z;
                pd[slot * 3] = m.dx; pd[slot * 3 + 1] = m.dy; pd[slot * 3 + 2] = m.dz;
                pl[slot * 2] = dt;
                pl[slot * 2 + 1] = t_after - (below ? prev_after : last_t);
                if (psf) psf[slot] = (uint8_t)frame;
            }
            last_t = top_after;
            step += (uint32_t)__builtin_popcount(emit);
        }
        if (done) break;
        exact_base = next_base;
        if (om == 0u && !(wbase < force_until)) {       // nothing to sample here: pass the following windows by their occupancy bits
            scanning = true;
            scan_base = next_base;
            prev_valid = false;
        }
    }
    for (uint32_t k = step + g; k < n_step; k += kGL) {
        px[k * 3] = 0; px[k * 3 + 1] = 0; px[k * 3 + 2] = 0;
        pd[k * 3] = 0; pd[k * 3 + 1] = 0; pd[k * 3 + 2] = 0;
        pl[k * 2] = 0; pl[k * 2 + 1] = 0;
    }
    return step;
}

// ---- frame groups (FrameSel::n_frames > 1) -------------------------------------------------------------------------------------
// The alive list is ordered by ray id (stable compaction; the steady mode's frozen list keeps that order) and rays are
// frame-major, so a 256-entry workgroup sees one frame, or two at a frame boundary.  A marching workgroup therefore loops over the
// frames present among its live entries (ascending; one round in all but the <= n_frames - 1 boundary workgroups of a launch):
// per round it loads the LDS occupancy caches of that frame and the lanes of that frame march.  Same code, same LDS address
// space, no per-lane indexing of the kernel-argument record (fs.grid[] is only ever indexed with a workgroup-uniform value).
__device__ __forceinline__ uint32_t block_min_256(uint32_t v, uint32_t *lds4) {
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return min(min(lds4[0], lds4[1]), min(lds4[2], lds4[3]));
}
constexpr uint32_t kNoFrame = 0xFFFFFFFFu;
__device__ __forceinline__ const uint8_t *frame_grid_uniform(const FrameSel &fs, uint32_t frame_uniform) {
    return fs.grid[__builtin_amdgcn_readfirstlane(frame_uniform)];
}

// Appends the slots n*n_step .. +step of every lane to the live list: one atomicAdd per wave, prefix by wave shuffles.
// Must be called by all 64 lanes of the wave.
__device__ __forceinline__ void live_append(uint32_t step, uint32_t n, uint32_t n_step, uint32_t *__restrict__ live_idx,
                                            uint32_t *__restrict__ live_count) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t incl = step;
    #pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += u;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(live_count, total);
    base = __shfl(base, 63, 64);
    const uint32_t dst = base + incl - step;
    for (uint32_t k = 0; k < step; k++) live_idx[dst + k] = n * n_step + k;
}

// Appends the slots n * n_step .. + steps of every list entry of the workgroup to the live list with ONE atomic per workgroup (a
// returning atomic on the one counter of an iteration costs ~11 ns of serialised time each: thousands of waves appending were a
// large part of a launch).  Entry i = thread i (slot base n0 + i).  Contains barriers: every thread of the workgroup calls it.
__device__ __forceinline__ void wg_live_append(const uint32_t *s_steps, uint32_t n0, uint32_t n_step, uint32_t *__restrict__ live_idx,
                                               uint32_t *__restrict__ live_count, uint32_t *s_scan) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t mine = s_steps[threadIdx.x];
    uint32_t incl = mine;
    #pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += u;
    }
    if (lane == 63u) s_scan[wid] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
        s_scan[4] = total ? atomicAdd(live_count, total) : 0u;
    }
    __syncthreads();
    uint32_t dst = s_scan[4] + incl - mine;
    for (uint32_t w = 0; w < wid; w++) dst += s_scan[w];
    if (mine) {
        const uint32_t first = (n0 + threadIdx.x) * n_step;
        for (uint32_t k = 0; k < mine; k++) live_idx[dst + k] = first + k;
    }
}

// Resumes the compositing of one ray over its n_step slots (raymarching.cu:845-904).  Returns true if the ray survives.
// The per-ray state and -- in the n_step == 8 case the loop spends most of its iterations in -- all 8 slots of the ray
// (8 sigmas, 24 colours, 16 deltas: twelve 16-byte loads) are fetched before the serial recurrence starts, so the ray pays one
// memory round trip instead of one per sample (a lane is a ray here and there is at most one wave per SIMD to hide latency
// behind); the arithmetic and its order, including the two early exits, are the reference's.
__device__ __forceinline__ bool composite_ray(int index, uint32_t n_step, float T_thresh, const float *s, const float *c, const float *dl,
                                              float *__restrict__ rays_t, float *__restrict__ weights_sum, float *__restrict__ depth,
                                              float *__restrict__ image, float *t_out = nullptr) {
    float t = rays_t[index];
    float weight_sum = weights_sum[index], d = depth[index];
    float r = image[(size_t)index * 3], g = image[(size_t)index * 3 + 1], b = image[(size_t)index * 3 + 2];
    uint32_t step = 0;
    if (n_step == 8) {
        float sv[8], cv[24], dv[16];
        #pragma unroll
        for (int q = 0; q < 2; q++) *reinterpret_cast<float4 *>(sv + 4 * q) = *reinterpret_cast<const float4 *>(s + 4 * q);
        #pragma unroll
        for (int q = 0; q < 6; q++) *reinterpret_cast<float4 *>(cv + 4 * q) = *reinterpret_cast<const float4 *>(c + 4 * q);
        #pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<float4 *>(dv + 4 * q) = *reinterpret_cast<const float4 *>(dl + 4 * q);
        bool open = true;   // false once the reference's loop would have left through a break
        #pragma unroll
        for (int k = 0; k < 8; k++) {
            if (open && dv[2 * k] == 0) open = false;
            if (open) {
                const float alpha = 1.0f - sdn_exp_cr(-sv[k] * dv[2 * k]);
                const float T = 1 - weight_sum;
                const float weight = alpha * T;
                weight_sum += weight;
                t += dv[2 * k + 1];
                d += weight * t;
                r += weight * cv[3 * k]; g += weight * cv[3 * k + 1]; b += weight * cv[3 * k + 2];
                if (T < T_thresh) open = false; else step++;
            }
        }
    } else {
        while (step < n_step) {
            if (dl[0] == 0) break;
            const float alpha = 1.0f - sdn_exp_cr(-s[0] * dl[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t += dl[1];
            d += weight * t;
            r += weight * c[0]; g += weight * c[1]; b += weight * c[2];
            if (T < T_thresh) break;
            s++; c += 3; dl += 2; step++;
        }
    }
    const bool survives = !(step < n_step);
    if (survives) rays_t[index] = t;
    if (t_out) *t_out = t;
    weights_sum[index] = weight_sum; depth[index] = d;
    image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
    return survives;
}

// raymarching.cu:701-805.  Extras over the reference kernel (all optional, none changes a sample):
//   * slots [step, n_step) of each alive ray are written as zeros here and the alignment tail [n_alive*n_step, M_pad)
//     is cleared by the same launch, so the caller never memsets the sample buffers;
//   * cull: exact early-out for rays that cannot produce a sample (see ray_may_hit), LDS occupancy caches;
//   * live_idx / live_count: the slots that received a sample are appended to a compact list so that the field network is
//     evaluated on samples, not on padded slots.
template <bool FAST>
__global__ void __launch_bounds__(256) k_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                                                    const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                                    const float *__restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                                    uint32_t C, uint32_t H, const uint8_t *__restrict__ grid,
                                                    const float *__restrict__ fars, float *__restrict__ xyzs, float *__restrict__ dirs,
                                                    float *__restrict__ deltas, const float *__restrict__ noises, uint32_t M_pad,
                                                    const uint32_t *__restrict__ cull, uint32_t *__restrict__ live_idx,
                                                    uint32_t *__restrict__ live_count, const int32_t *__restrict__ state,
                                                    const int32_t *__restrict__ rays_alive_b, FrameSel fs, const float *__restrict__ jump) {
    if (state) {  // device-driven loop: sizes, ping-pong side and the iteration's live counter come from the loop state
        n_alive = (uint32_t)state[0];
        n_step = (uint32_t)state[1];
        if (n_alive == 0) return;
        if (state[4]) rays_alive = rays_alive_b;
        live_count += state[3];
        const uint32_t m0 = n_alive * n_step;
        M_pad = m0 + (128u - m0 % 128u);
        if (!(state[15] && state[3] == 0)) jump = nullptr;   // the per-ray jump targets of k_cull_start hold for the first march only
        if (blockIdx.x * 256u >= n_alive + 128u) return;   // workgroup-uniform: beyond the list and its alignment tail (the host sizes the grid by a bound)
    }
    __shared__ uint4 s_cull4[FAST ? 256 : 1];  // 32^3 bits = 4 KiB
    __shared__ unsigned long long s_fine[FAST ? kFineCacheCells : 1];  // <= 32 KiB
    __shared__ uint32_t s_min4[4];
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    const int index = n < n_alive ? rays_alive[n] : -1;
    const bool grouped = fs.n_frames > 1;  // kernel-uniform
    const uint32_t frame = (grouped && index >= 0) ? (uint32_t)index / fs.rays_per_frame : (grouped ? kNoFrame : 0u);
    uint32_t step = 0;
    // the ray's inputs are fetched NOW: they land under the workgroup's barriers and its LDS fill instead of behind them
    float ro[3] = {0, 0, 0}, rdv[3] = {0, 0, 1}, t_ray = 0, far_ray = 0, t_jump = 0, noise = 0;
    if (index >= 0) {
        ro[0] = rays_o[(size_t)index * 3]; ro[1] = rays_o[(size_t)index * 3 + 1]; ro[2] = rays_o[(size_t)index * 3 + 2];
        rdv[0] = rays_d[(size_t)index * 3]; rdv[1] = rays_d[(size_t)index * 3 + 1]; rdv[2] = rays_d[(size_t)index * 3 + 2];
        t_ray = rays_t[index]; far_ray = fars[index];
        if (jump) t_jump = jump[index];
        if (noises) noise = noises[n];
    }
    uint32_t fcur = grouped ? block_min_256(frame, s_min4) : 0u;
    while (fcur != kNoFrame) {   // one round per frame present in this workgroup (exactly one without a frame group)
        const uint8_t *grid_f = grouped ? frame_grid_uniform(fs, fcur) : grid;
        const uint32_t *cull_f = (grouped && cull) ? cull + (size_t)fcur * fs.cull_stride : cull;
        OccCache oc;
        occ_cache_load<FAST>(cull_f, grid_f, s_cull4, s_fine, oc);
        if (index >= 0 && frame == fcur) {
            MarcherT<FAST> m;
            m.init(ro, rdv, bound, dt_gamma, max_steps, C, H, grid_f);
            float t = t_ray;
            t += m.step_size(t) * noise;
            float *tend = state ? state_tend(state) : nullptr;
            if (tend) tend += index;
            const float t_walk = jump ? t_jump : t;   // (== t where k_cull_start certified no jump)
            step = march_ray<FAST>(m, oc, t_walk, far_ray, n_step, xyzs + (size_t)n * n_step * 3, dirs + (size_t)n * n_step * 3,
                                   deltas + (size_t)n * n_step * 2, tend, grouped ? fs.slot_frame + (size_t)n * n_step : nullptr, frame,
                                   jump ? &t : nullptr);
        }
        if (!grouped) break;
        fcur = block_min_256((frame != kNoFrame && frame > fcur) ? frame : kNoFrame, s_min4);   // (the barriers inside also fence the LDS caches)
    }
    if (n >= n_alive) {
        const uint32_t slot = n_alive * n_step + (n - n_alive);  // spare lanes of the last blocks clear the alignment tail
        if (slot < M_pad) {
            xyzs[(size_t)slot * 3] = 0; xyzs[(size_t)slot * 3 + 1] = 0; xyzs[(size_t)slot * 3 + 2] = 0;
            dirs[(size_t)slot * 3] = 0; dirs[(size_t)slot * 3 + 1] = 0; dirs[(size_t)slot * 3 + 2] = 0;
            deltas[(size_t)slot * 2] = 0; deltas[(size_t)slot * 2 + 1] = 0;
        }
    }
    if (live_idx) live_append(step, n, n_step, live_idx, live_count);  // kernel-uniform condition
}

// Length of the alive list the loop kernels walk: the frozen list of the steady mode (state[8], dead entries = -1 included) when one
// exists -- the compositing + compaction pass then RE-compacts it (render.hip, FrameRun::enqueue) -- else the compacted list (state[0]).
__device__ __forceinline__ uint32_t loop_list_len(const int32_t *__restrict__ state) {
    return state[8] ? (uint32_t)state[8] : (uint32_t)state[0];
}

// raymarching.cu:819-905
__global__ void __launch_bounds__(256) k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *__restrict__ rays_alive,
                                                        float *__restrict__ rays_t, const float *__restrict__ sigmas,
                                                        const float *__restrict__ rgbs, const float *__restrict__ deltas,
                                                        float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image,
                                                        const int32_t *__restrict__ state, int32_t *__restrict__ rays_alive_b,
                                                        uint32_t *__restrict__ block_totals) {
    if (state) {
        n_alive = loop_list_len(state);
        n_step = (uint32_t)state[1];
        if (state[4]) rays_alive = rays_alive_b;
    }
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    bool survives = false;
    if (n < n_alive) {
        const int index = rays_alive[n];
        if (index >= 0) {    // (a frozen list holds dead entries)
            survives = composite_ray(index, n_step, T_thresh, sigmas + (size_t)n * n_step, rgbs + (size_t)n * n_step * 3,
                                     deltas + (size_t)n * n_step * 2, rays_t, weights_sum, depth, image);
            if (!survives) rays_alive[n] = -1;
        }
    }
    if (block_totals) {  // kernel-uniform: survivor count of this 256-ray block, for the fused compaction of the device loop
        __shared__ uint32_t s_cnt[4];
        const unsigned long long m = __ballot(survives);
        if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) block_totals[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
}

// ---------------------------------------------------------------------------
// stable compaction of rays_alive >= 0 (replaces the torch mask-select of dnerf/renderer.py:372)
// two launches: per-block survivor counts (wave ballot + popcount), then scatter.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kScanBlock) k_compact_count(const int32_t *__restrict__ in, uint32_t n, uint32_t *__restrict__ block_totals,
                                                              const int32_t *__restrict__ state, const int32_t *__restrict__ in_b) {
    if (state) {
        n = loop_list_len(state);
        if (state[4]) in = in_b;
        if (blockIdx.x * kScanBlock >= n) return;  // workgroup-uniform
    }
    __shared__ uint32_t lds[16];
    const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
    const bool keep = i < n && in[i] >= 0;
    const unsigned long long mask = __ballot(keep);
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < (kScanBlock >> 6); w++) t += lds[w];
        block_totals[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(kScanBlock) k_compact_scatter(const int32_t *__restrict__ in, uint32_t n, const uint32_t *__restrict__ block_totals,
                                                                int32_t *__restrict__ out, int32_t *__restrict__ n_out,
                                                                const int32_t *__restrict__ state, const int32_t *__restrict__ in_b,
                                                                int32_t *__restrict__ out_b, uint32_t totals_per_block) {
    // totals_per_block: block_totals holds one count per (kScanBlock / totals_per_block) entries -- 1: k_compact_count's, 4: the
    // per-256-ray survivor counts k_composite_rays wrote (the device loop skips the count launch)
    uint32_t last_block = gridDim.x - 1;
    if (state) {
        n = loop_list_len(state);
        if (state[4]) { in = in_b; out = out_b; }
        if (n == 0) {
            if (blockIdx.x == 0 && threadIdx.x == 0) n_out[0] = 0;
            return;
        }
        last_block = (n - 1) / kScanBlock;
        if (blockIdx.x > last_block) return;  // workgroup-uniform
    }
    __shared__ uint32_t lds[16];
    __shared__ uint32_t s_prev;
    uint32_t part = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x * totals_per_block; b += kScanBlock) part += block_totals[b];
    uint32_t prev_total;
    block_inclusive_scan(part, lds, prev_total);
    if (threadIdx.x == 0) s_prev = prev_total;
    __syncthreads();
    const uint32_t i = blockIdx.x * kScanBlock + threadIdx.x;
    const int32_t v = i < n ? in[i] : -1;
    const bool keep = v >= 0;
    const unsigned long long mask = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t below = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) lds[wid] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t carry = 0, total = 0;
    for (uint32_t w = 0; w < (kScanBlock >> 6); w++) {
        const uint32_t c = lds[w];
        if (w < wid) carry += c;
        total += c;
    }
    if (keep) out[s_prev + carry + below] = v;
    if (blockIdx.x == last_block && threadIdx.x == 0) n_out[0] = (int32_t)(s_prev + total);
}


// ---------------------------------------------------------------------------
// device-driven inference loop (dnerf/renderer.py:340-381 without a host round trip per iteration)
// state: [0] n_alive  [1] n_step  [2] steps done  [3] iteration  [4] ping-pong side  [5] N  [6] max_steps  [7] -
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_loop_init(uint32_t N, uint32_t max_steps, const float *__restrict__ nears, int32_t *__restrict__ alive_a,
                                                   float *__restrict__ rays_t, float *__restrict__ weights_sum, float *__restrict__ depth,
                                                   float *__restrict__ image, int32_t *__restrict__ state, int32_t *__restrict__ live_counts,
                                                   uint32_t n_counters, unsigned long long mailbox, uint32_t frame_tag,
                                                   float *__restrict__ rays_tend) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n < N) {
        if (rays_tend) rays_tend[n] = kTendUnset;
        alive_a[n] = (int32_t)n;
        rays_t[n] = nears[n];
        weights_sum[n] = 0; depth[n] = 0;
        image[(size_t)n * 3] = 0; image[(size_t)n * 3 + 1] = 0; image[(size_t)n * 3 + 2] = 0;
    }
    if (n < n_counters) live_counts[n] = 0;
    if (n == 0) {
        state[0] = (int32_t)N; state[1] = 1; state[2] = 0; state[3] = 0; state[4] = 0;
        state[5] = (int32_t)N; state[6] = (int32_t)max_steps; state[7] = 0;
        for (int k = 8; k < 16; k++) state[k] = 0;  // [8] frozen list length of the steady mode, [9] survivor accumulator
        state[10] = (int32_t)(uint32_t)mailbox;      // [10],[11] host mailbox (device-visible pointer, 0 = none), [12] frame tag
        state[11] = (int32_t)(uint32_t)(mailbox >> 32);
        state[12] = (int32_t)frame_tag;
        const unsigned long long tp = (unsigned long long)(uintptr_t)rays_tend;   // [13],[14] per-ray t_end cache (0 = none)
        state[13] = (int32_t)(uint32_t)tp;
        state[14] = (int32_t)(uint32_t)(tp >> 32);
    }
}


// Per-iteration snapshot {alive rays entering the next iteration, index of that iteration}: into the device ring `snap`
// (4 deep, immutable once written: the host may copy it out while the next iteration runs) and, when the frame driver
// registered a mailbox in coherent host memory, as ONE 64-bit system-scope store {tag : n_alive} the host polls for --
// no event, no copy, no stream wait on the critical path.  tag = frame_tag << 16 | (call + 1).
__device__ __forceinline__ void publish_snapshot(int32_t *__restrict__ state, int32_t *__restrict__ snap, int32_t call) {
    snap[(call & 3) * 2] = state[0];
    snap[(call & 3) * 2 + 1] = call + 1;
    const unsigned long long mb = ((unsigned long long)(uint32_t)state[11] << 32) | (uint32_t)state[10];
    if (mb) {
        const unsigned long long tag = ((uint32_t)state[12] << 16) | (uint32_t)(call + 1);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(mb) + (call & 3), (tag << 32) | (uint32_t)state[0], __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Culled start of the device loop (FAST configuration with a cull grid and the per-ray t_end cache).  The reference's iteration 0 marches
// all N rays by one step and the rays without a sample die in its compositing pass; here the exact cull test every ray would run on
// its first march (ray_may_hit: same arguments, same result, the marks read from global memory instead of an LDS copy) runs up front
// for all N rays -- filling the t_end cache -- and iteration 0 then works on the compacted list of the rays that MAY produce a sample:
// dense waves instead of N-ray launches in which most lanes stop at the cull test.  Nothing observable changes: a ray the test rejects
// produces no sample and dies in iteration 0 either way, the survivors of iteration 0 -- and therefore every later iteration, every
// sample and every count -- are the same, and the trace logs N for iteration 0 (state[15], k_loop_advance / k_scatter_advance).
// The certified jump.  With a constant step every parameter the marcher visits is a point of the lattice L_0 = t, L_(k+1) = fl(L_k + dt);
// the walk from t visits, of every voxel that holds lattice points, a first one, and leaves it for the first lattice point >= the
// voxel's exit parameter.  Everything before `t_safe` (ray_may_hit) is >= 2 fine voxels away from any occupied voxel, so the walk emits
// nothing there -- it only matters WHERE it stands when it gets there.  lattice_floor gives a lattice point L well before t_safe
// (bit-exact, or it stops early at one); L lies in some run of consecutive lattice points that `locate` puts into the same voxel.
// If every point q of that run has exit_t(q) <= L_a, the first lattice point behind the run, then the walk, wherever in the run it
// lands (it cannot jump the run: the exit parameter of the voxel before it is at most a rounding error beyond the run's first
// point), can never leave it for a point beyond L_a, and every hop advances: L_a is CERTAINLY visited.  The march may start there.
// Returns t when anything is not provable (binade edge, rounding tie, long run).
__device__ __forceinline__ float certified_jump(const MarcherT<true> &m, float t, float t_safe) {
    const float dt = m.dt_const;
    if (!(t_safe > t + 64.0f * dt)) return t;
    const float L = lattice_floor(t, dt, t_safe - 18.0f * dt);
    if (!(L > t)) return t;
    // closed form of the lattice around L (the conditions of lattice_floor, for the 14 steps either side that are looked at)
    const float c = (L + dt) - L, err = dt - c;
    const uint32_t eb = __float_as_uint(L) >> 23;
    if (!(L >= dt && dt > 0.0f && eb > 30u && eb < 254u)) return t;
    const float half_ulp = __uint_as_float((eb - 24u) << 23), c_cap = __uint_as_float((eb - 5u) << 23);
    if (!(fabsf(err) != half_ulp && c < c_cap && c > 0.0f)) return t;
    if ((__float_as_uint(L - 14.0f * c) >> 23) != eb || (__float_as_uint(L + 14.0f * c) >> 23) != eb) return t;
    float x, y, z;
    int vx, vy, vz, nx, ny, nz;
    m.locate(L, x, y, z, vx, vy, vz);
    // back to the first lattice point of the run
    float q = L;
    int k = 0;
    for (; k < 12; k++) {
        const float qm = q - c;
        if (!(qm > t) || (qm + dt) != q) return t;
        m.locate(qm, x, y, z, nx, ny, nz);
        if (nx != vx || ny != vy || nz != vz) break;
        q = qm;
    }
    if (k == 12) return t;
    // forward over the whole run: the largest exit parameter any of its points computes, and the first point behind it
    float max_tt = -__FLT_MAX__;
    for (k = 0; k < 26; k++) {
        m.locate(q, x, y, z, nx, ny, nz);
        if (nx != vx || ny != vy || nz != vz) break;
        max_tt = fmaxf(max_tt, m.exit_t(q, x, y, z, nx, ny, nz));
        q = q + dt;                                   // the lattice's own recurrence
    }
    if (k == 26 || !(max_tt <= q) || !(q < t_safe - 4.0f * dt)) return t;
    return q;
}

__device__ __forceinline__ bool cull_start_ray(uint32_t n, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                               const float *__restrict__ nears, const float *__restrict__ fars, const uint32_t *__restrict__ cull,
                                               const FrameSel &fs, int32_t *__restrict__ alive_a, float *__restrict__ rays_tend, float bound,
                                               float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, float *__restrict__ jump) {
    const uint32_t frame = fs.n_frames > 1 ? n / fs.rays_per_frame : 0u;
    const uint32_t *__restrict__ cull_f = fs.n_frames > 1 ? cull + (size_t)frame * fs.cull_stride : cull;
    const int *meta = reinterpret_cast<const int *>(cull_f + kCullWords);
    const int fx0 = meta[0], fy0 = meta[1], fz0 = meta[2];
    const int fnx = meta[3] - fx0 + 1, fny = meta[4] - fy0 + 1, fnz = meta[5] - fz0 + 1;
    const float t = nears[n], far = fars[n];
    bool go = t < far;
    float t_end = far, start = t;
    if (go) {
        MarcherT<true> m;
        m.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, nullptr);
        float t_safe;
        go = ray_may_hit(cull_f, m.ox, m.oy, m.oz, m.dx, m.dy, m.dz, t, far, t_end, fx0, fy0, fz0, fnx, fny, fnz, &t_safe);
        rays_tend[n] = go ? t_end : kTendDead;     // what march_ray would cache on the ray's first march
        if (go && jump && m.dt_is_const) start = certified_jump(m, t, fminf(t_safe, fminf(far, t_end)));
    }
    alive_a[n] = go ? (int32_t)n : -1;
    if (jump) jump[n] = start;
    return go;
}

__global__ void __launch_bounds__(256) k_cull_start(uint32_t N, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                    const float *__restrict__ nears, const float *__restrict__ fars,
                                                    const uint32_t *__restrict__ cull, FrameSel fs, int32_t *__restrict__ alive_a,
                                                    float *__restrict__ rays_tend, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                                                    uint32_t H, float *__restrict__ jump, uint32_t *__restrict__ block_totals) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    bool go = false;
    if (n < N) go = cull_start_ray(n, rays_o, rays_d, nears, fars, cull, fs, alive_a, rays_tend, bound, dt_gamma, max_steps, C, H, jump);
    // rays of this 256-ray block that go on: the compaction's scatter sums these (no separate count launch)
    __shared__ uint32_t s_cnt[4];
    const unsigned long long m = __ballot(go);
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_totals[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// after the compaction of the culled start: the list is in alive_b (side 1), n_out[0] rays long
__global__ void k_cull_advance(int32_t *__restrict__ state, const int32_t *__restrict__ n_out, int32_t *__restrict__ trace) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t n0 = n_out[0];
    state[4] = 1;
    state[15] = 1;            // iteration 0 runs on the culled list: the trace logs N for it
    if (n0 > 0) {
        state[0] = n0;
    } else {                  // no ray can produce a sample: the reference's iteration 0 (N rays, one step) finds nothing and the loop ends
        trace[0] = state[5];
        trace[1] = state[1];
        state[2] += state[1];
        state[3] = 1;
        state[0] = 0;
    }
}

// snap: 4 x {alive rays entering the next iteration, index of that iteration} -- an immutable per-iteration snapshot the
// host copies out on a side stream while the main stream already runs the next iteration.
__global__ void k_loop_advance(int32_t *__restrict__ state, const int32_t *__restrict__ n_out, int32_t *__restrict__ trace,
                               int32_t *__restrict__ snap, int freeze) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t it = state[3];
    const int32_t call = state[7];  // number of advance calls so far (no-op iterations included)
    state[7] = call + 1;
    if (state[0] > 0) {  // log only iterations that did work: (n_alive, n_step)
        trace[2 * it] = (it == 0 && state[15]) ? state[5] : state[0];   // culled start: iteration 0 is the reference's N-ray iteration
        trace[2 * it + 1] = state[1];
        state[2] += state[1];
        state[3] = it + 1;
        state[4] ^= 1;
        int32_t n_new = n_out[0];
        if (state[2] >= state[6]) n_new = 0;  // `while step < max_steps`
        state[0] = n_new;
        if (n_new > 0) {
            int32_t ns = state[5] / n_new;  // n_step = max(min(N // n_alive, 8), 1)
            state[1] = ns > 8 ? 8 : (ns < 1 ? 1 : ns);
        }
    }
    if (freeze) { state[8] = state[0]; state[9] = 0; }   // k_steady_begin folded in: the compacted list is the steady mode's new frozen list
    publish_snapshot(state, snap, call);
}

__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t *lds4) {
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();  // protects lds4 against the previous use
    if ((threadIdx.x & 63u) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// Stable compaction of the alive list (256-ray blocks, totals produced by k_composite_rays) fused with the loop advance:
// the workgroup that draws the last ticket -- every workgroup of the launch takes one after its last read of the loop
// record -- sums the totals and advances the record, so no workgroup can see a half-updated record.
__global__ void __launch_bounds__(256) k_scatter_advance(int32_t *__restrict__ alive_a, int32_t *__restrict__ alive_b,
                                                         const uint32_t *__restrict__ block_totals, int32_t *__restrict__ state,
                                                         int32_t *__restrict__ ticket, int32_t *__restrict__ trace, int32_t *__restrict__ snap,
                                                         int freeze) {
    __shared__ uint32_t lds4[4];
    __shared__ int s_last;
    const uint32_t n = loop_list_len(state);
    const int32_t *in = state[4] ? alive_b : alive_a;
    int32_t *out = state[4] ? alive_a : alive_b;
    const uint32_t nb = (n + 255u) / 256u;
    if (blockIdx.x < nb) {
        uint32_t part = 0;
        for (uint32_t b = threadIdx.x; b < blockIdx.x; b += 256) part += block_totals[b];
        const uint32_t prev = block_sum_256(part, lds4);
        const uint32_t i = blockIdx.x * 256u + threadIdx.x;
        const int32_t v = i < n ? in[i] : -1;
        const bool keep = v >= 0;
        const unsigned long long mask = __ballot(keep);
        const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
        const uint32_t below = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) lds4[wid] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t carry = 0;
        for (uint32_t w = 0; w < wid; w++) carry += lds4[w];
        if (keep) out[prev + carry + below] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    uint32_t part = 0;
    for (uint32_t b = threadIdx.x; b < nb; b += 256) part += block_totals[b];
    const uint32_t total = block_sum_256(part, lds4);
    if (threadIdx.x == 0) {
        *ticket = 0;
        const int32_t it = state[3];
        const int32_t call = state[7];
        state[7] = call + 1;
        if (state[0] > 0) {
            trace[2 * it] = (it == 0 && state[15]) ? state[5] : state[0];
            trace[2 * it + 1] = state[1];
            state[2] += state[1];
            state[3] = it + 1;
            state[4] ^= 1;
            int32_t n_new = (int32_t)total;
            if (state[2] >= state[6]) n_new = 0;  // `while step < max_steps`
            state[0] = n_new;
            if (n_new > 0) {
                const int32_t ns = state[5] / n_new;  // n_step = max(min(N // n_alive, 8), 1)
                state[1] = ns > 8 ? 8 : (ns < 1 ? 1 : ns);
            }
        }
        if (freeze) { state[8] = state[0]; state[9] = 0; }   // (k_steady_begin folded in)
        publish_snapshot(state, snap, call);
    }
}


// ---------------------------------------------------------------------------
// steady mode of the device loop: once n_alive <= N / 8 the schedule is n_step = 8 for good (n_alive only shrinks), so
// compositing iteration k and marching iteration k+1 of the same ray can be ONE kernel, and the alive list need not be
// compacted any more (dead entries are skipped; the field network only sees live samples through the live list).
// Two launches per iteration (fused field, composite+march) instead of five.  Samples, schedule and counts are unchanged.
// ---------------------------------------------------------------------------
__global__ void k_steady_begin(int32_t *__restrict__ state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { state[8] = state[0]; state[9] = 0; }
}

template <bool FAST>
__global__ void __launch_bounds__(256) k_composite_march(float T_thresh, int32_t *__restrict__ alive_a, int32_t *__restrict__ alive_b,
                                                         float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                                         const float *__restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                                         uint32_t C, uint32_t H, const uint8_t *__restrict__ grid, const float *__restrict__ fars,
                                                         const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                         float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                                                         float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image,
                                                         const uint32_t *__restrict__ cull, uint32_t *__restrict__ live_idx,
                                                         uint32_t *__restrict__ live_counts, int32_t *__restrict__ state,
                                                         int32_t *__restrict__ ticket, int32_t *__restrict__ trace, int32_t *__restrict__ snap,
                                                         FrameSel fs) {
    __shared__ uint4 s_cull4[FAST ? 256 : 1];
    __shared__ unsigned long long s_fine[FAST ? kFineCacheCells : 1];
    __shared__ uint32_t s_cnt[4], s_min4[4];
    __shared__ int s_last;
    const uint32_t n_alive = (uint32_t)state[0], n_step = (uint32_t)state[1], list_len = (uint32_t)state[8];
    const int32_t it = state[3];
    int32_t *__restrict__ alive = state[4] ? alive_b : alive_a;
    const bool march_next = (uint32_t)state[2] + n_step < (uint32_t)state[6];  // `while step < max_steps` admits another iteration
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    bool survives = false;
    uint32_t emitted = 0;
    if (n_alive > 0 && blockIdx.x * 256u < list_len) {  // workgroup-uniform
        const int index = n < list_len ? alive[n] : -1;
        // composite iteration `it` of this ray; survivors march iteration it + 1 below.  The marcher's inputs are fetched together with
        // the compositing's (they land under it, the barriers and the LDS fill), and the ray's new t travels in a register
        float ro[3] = {0, 0, 0}, rdv[3] = {0, 0, 1}, far_ray = 0, t_new = 0;
        if (index >= 0) {
            if (march_next) {
                ro[0] = rays_o[(size_t)index * 3]; ro[1] = rays_o[(size_t)index * 3 + 1]; ro[2] = rays_o[(size_t)index * 3 + 2];
                rdv[0] = rays_d[(size_t)index * 3]; rdv[1] = rays_d[(size_t)index * 3 + 1]; rdv[2] = rays_d[(size_t)index * 3 + 2];
                far_ray = fars[index];
            }
            survives = composite_ray(index, n_step, T_thresh, sigmas + (size_t)n * n_step, rgbs + (size_t)n * n_step * 3,
                                     deltas + (size_t)n * n_step * 2, rays_t, weights_sum, depth, image, &t_new);
            if (!survives) alive[n] = -1;
        }
        const bool marches = survives && march_next;
        const bool grouped = fs.n_frames > 1;  // kernel-uniform
        const uint32_t frame = (grouped && marches) ? (uint32_t)index / fs.rays_per_frame : (grouped ? kNoFrame : 0u);
        uint32_t fcur = grouped ? block_min_256(frame, s_min4) : 0u;
        while (fcur != kNoFrame) {   // one round per frame present among this workgroup's marching rays (see block_min_256)
            const uint8_t *grid_f = grouped ? frame_grid_uniform(fs, fcur) : grid;
            const uint32_t *cull_f = (grouped && cull) ? cull + (size_t)fcur * fs.cull_stride : cull;
            OccCache oc;
            occ_cache_load<FAST>(cull_f, grid_f, s_cull4, s_fine, oc);
            if (marches && frame == fcur) {
                float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3, *pl = deltas + (size_t)n * n_step * 2;
                MarcherT<FAST> m;
                m.init(ro, rdv, bound, dt_gamma, max_steps, C, H, grid_f);
                float *tend = state_tend(state);
                if (tend) tend += index;
                emitted = march_ray<FAST>(m, oc, t_new, far_ray, n_step, px, pd, pl, tend,
                                          grouped ? fs.slot_frame + (size_t)n * n_step : nullptr, frame);
            }
            if (!grouped) break;
            fcur = block_min_256((frame != kNoFrame && frame > fcur) ? frame : kNoFrame, s_min4);
        }
        live_append(emitted, n, n_step, live_idx, live_counts + it + 1);
    }
    // survivors of this workgroup -> global accumulator; the last workgroup of the launch advances the loop record
    const unsigned long long mask = __ballot(survives);
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (c) atomicAdd(state + 9, (int32_t)c);
        __threadfence();
        s_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    *ticket = 0;
    const int32_t call = state[7];
    state[7] = call + 1;
    if (state[0] > 0) {
        trace[2 * it] = state[0];
        trace[2 * it + 1] = state[1];
        state[2] += state[1];
        state[3] = it + 1;
        int32_t n_new = atomicAdd(state + 9, 0);
        state[9] = 0;
        if (state[2] >= state[6]) n_new = 0;
        state[0] = n_new;  // n_step stays 8: N / n_new >= 8 holds from here on
    }
    publish_snapshot(state, snap, call);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The loop kernels on the cooperative marcher (FAST configuration with a constant step): phase A runs one lane per ray -- alive
// test, cull scan or cached bound, compositing in the fused kernel -- and lists the rays that march in LDS; phase B marches the
// listed rays 16 at a time, kGL lanes each (group_march).  Outputs are those of k_march_rays<true> / k_composite_march<true> bit
// for bit (same slots, same zero fill, same counters); only the order of the live-sample list differs (it is unordered anyway).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kGW = 256;     // list entries (rays) per 256-thread workgroup: phase A one lane per entry, phase B 16 rays per pass
struct GroupTask { uint32_t n, tid, step0; float t, t_end, t_safe, far, last_t, o[3], d[3]; };   // 56 bytes: phase B reads no global memory to start a ray
constexpr uint32_t kProbeBudget = 12;   // loop-body evaluations a lane spends on its own ray (8 samples + a few empty voxels) before the ray goes to a group

// Phase B: the listed rays, 16 at a time, kGL lanes each.
__device__ __forceinline__ void run_group_tasks(const GroupTask *s_task, uint32_t *s_steps, uint32_t *s_scan, uint32_t ntask, const OccCache &oc,
                                                float bound, uint32_t max_steps, uint32_t H, const uint8_t *grid_f, uint32_t n_step,
                                                float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas, uint8_t *slot_frame,
                                                uint32_t frame, uint32_t stamp_base) {
    const uint32_t lane = threadIdx.x & 63u, g = lane & (kGL - 1u), gbase = lane & ~(kGL - 1u);
    const unsigned long long st0 = SDN_STAMP_NOW();
    for (uint32_t first = 0; first < ntask; first += 256u / kGL) {        // workgroup-uniform trip count
        const uint32_t gi = first + (threadIdx.x / kGL);
        if (gi < ntask) {                                                   // group-uniform
            const GroupTask tk = s_task[gi];
            MarcherT<true> m;
            m.init(tk.o, tk.d, bound, 0.0f, max_steps, 1u, H, grid_f);
            m.fine = oc.fine; m.fx0 = oc.fx0; m.fy0 = oc.fy0; m.fz0 = oc.fz0; m.fnx = oc.fnx; m.fny = oc.fny; m.fnz = oc.fnz;
            const uint32_t n = tk.n;
            const uint32_t step = group_march(m, oc.s_cull, tk.t, tk.far, tk.t_end, n_step, xyzs + (size_t)n * n_step * 3,
                                              dirs + (size_t)n * n_step * 3, deltas + (size_t)n * n_step * 2,
                                              slot_frame ? slot_frame + (size_t)n * n_step : nullptr, frame, g, gbase, tk.t_safe, tk.step0, tk.last_t,
#ifdef SDN_STAMPS
                                              s_scan + 6
#else
                                              nullptr
#endif
                                              );
            if (g == 0u) s_steps[tk.tid] = step;
        }
    }
    if (threadIdx.x == 0) {
        SDN_STAMP_ADD(stamp_base + 3, SDN_STAMP_NOW() - st0);
        SDN_STAMP_ADD(stamp_base + 5, ntask);
        SDN_STAMP_ADD(stamp_base + 0, 1);
    }
}

// Phase A of a marching ray (one lane): cull scan or its cached result; then the lane marches its own ray for at most kProbeBudget
// loop bodies (march_ray_some) -- a ray inside the figure is finished by then -- and a ray that is not finished (a long run of empty
// voxels ahead: a fresh ray on its way from the cube's face to the figure goes there at once) becomes a task for a group of phase B.
// Returns the samples written so far.
__device__ __forceinline__ uint32_t group_enlist(GroupTask *s_task, uint32_t *s_ntask, const OccCache &oc, MarcherT<true> &m, uint32_t n, int index,
                                                 float t, float far, float *tend, uint32_t n_step, float *__restrict__ xyzs,
                                                 float *__restrict__ dirs, float *__restrict__ deltas, uint8_t *slot_frame, uint32_t frame) {
    bool go = t < far;
    float t_end = far, t_safe = -__FLT_MAX__;
    if (oc.s_cull && go) {
        const float cached = tend ? *tend : kTendUnset;
        if (cached != kTendUnset) {
            t_end = cached;                 // kTendDead (< every t) ends the ray here
            go = t < t_end;
        } else {
            go = ray_may_hit(oc.s_cull, m.ox, m.oy, m.oz, m.dx, m.dy, m.dz, t, far, t_end, oc.fx0, oc.fy0, oc.fz0, oc.fnx, oc.fny, oc.fnz, &t_safe);
            if (tend) *tend = go ? t_end : kTendDead;
        }
    }
    float *px = xyzs + (size_t)n * n_step * 3, *pd = dirs + (size_t)n * n_step * 3, *pl = deltas + (size_t)n * n_step * 2;
    uint8_t *psf = slot_frame ? slot_frame + (size_t)n * n_step : nullptr;
    uint32_t step = 0;
    float last_t = t;
    bool more = false;
    const float t_start = t;
    if (go) {
        const bool jump = t_safe > t + 48.0f * m.dt_const;           // a fresh ray far from the first marked cell: straight to a group
        step = march_ray_some(m, oc, t, far, t_end, n_step, px, pd, pl, psf, frame, jump ? 0u : kProbeBudget, last_t, more);
    } else {
        for (uint32_t k = 0; k < n_step; k++) {
            px[0] = 0; px[1] = 0; px[2] = 0;
            pd[0] = 0; pd[1] = 0; pd[2] = 0;
            pl[0] = 0; pl[1] = 0;
            px += 3; pd += 3; pl += 2;
        }
    }
    if (more) {
        const uint32_t at = atomicAdd(s_ntask, 1u);
        s_task[at] = GroupTask{n, threadIdx.x, step, t, t_end, step == 0u && t == t_start ? t_safe : -__FLT_MAX__, far, last_t,
                               {m.ox, m.oy, m.oz}, {m.dx, m.dy, m.dz}};
    }
    return step;
}

__global__ void __launch_bounds__(256) k_march_rays_g(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive,
                                                      const float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                                      const float *__restrict__ rays_d, float bound, uint32_t max_steps, uint32_t H,
                                                      const uint8_t *__restrict__ grid, const float *__restrict__ fars, float *__restrict__ xyzs,
                                                      float *__restrict__ dirs, float *__restrict__ deltas, const float *__restrict__ noises,
                                                      uint32_t M_pad, const uint32_t *__restrict__ cull, uint32_t *__restrict__ live_idx,
                                                      uint32_t *__restrict__ live_count, const int32_t *__restrict__ state,
                                                      const int32_t *__restrict__ rays_alive_b, FrameSel fs) {
    if (state) {  // device-driven loop: sizes, ping-pong side and the iteration's live counter come from the loop state
        n_alive = (uint32_t)state[0];
        n_step = (uint32_t)state[1];
        if (n_alive == 0) return;
        if (state[4]) rays_alive = rays_alive_b;
        live_count += state[3];
        const uint32_t m0 = n_alive * n_step;
        M_pad = m0 + (128u - m0 % 128u);
    }
    __shared__ uint4 s_cull4[256];
    __shared__ unsigned long long s_fine[kFineCacheCells];
    __shared__ uint32_t s_min4[4];
    __shared__ GroupTask s_task[kGW];
    __shared__ uint32_t s_ntask, s_steps[kGW], s_scan[8];
    if (blockIdx.x * kGW >= n_alive + (M_pad - n_alive * n_step)) return;   // workgroup-uniform (grids are sized by an upper bound)
    const uint32_t n = blockIdx.x * kGW + threadIdx.x;
    const int index = n < n_alive ? rays_alive[n] : -1;
    const bool grouped = fs.n_frames > 1;  // kernel-uniform
    const uint32_t frame = (grouped && index >= 0) ? (uint32_t)index / fs.rays_per_frame : (grouped ? kNoFrame : 0u);
    s_steps[threadIdx.x] = 0;
    uint32_t fcur = grouped ? block_min_256(frame, s_min4) : 0u;
    while (fcur != kNoFrame) {   // one round per frame present in this workgroup (exactly one without a frame group)
        const uint8_t *grid_f = grouped ? frame_grid_uniform(fs, fcur) : grid;
        const uint32_t *cull_f = (grouped && cull) ? cull + (size_t)fcur * fs.cull_stride : cull;
        if (threadIdx.x == 0) { s_ntask = 0; s_scan[6] = 0; s_scan[7] = 0; }
        const unsigned long long sa0 = SDN_STAMP_NOW();
        OccCache oc;
        occ_cache_load<true>(cull_f, grid_f, s_cull4, s_fine, oc);
        if (!cull_f) __syncthreads();
        const unsigned long long sa1 = SDN_STAMP_NOW();
        if (index >= 0 && frame == fcur) {
            MarcherT<true> m;
            m.init(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, 0.0f, max_steps, 1u, H, grid_f);
            float t = rays_t[index];
            t += m.step_size(t) * (noises ? noises[n] : 0.0f);
            float *tend = state ? state_tend(state) : nullptr;
            if (tend) tend += index;
            s_steps[threadIdx.x] = group_enlist(s_task, &s_ntask, oc, m, n, index, t, fars[index], tend, n_step, xyzs, dirs, deltas,
                                                grouped ? fs.slot_frame : nullptr, fcur);
        }
        __syncthreads();
        const uint32_t ntask = s_ntask;
        if (threadIdx.x == 0) { SDN_STAMP_ADD(8 + 2, sa1 - sa0); SDN_STAMP_ADD(8 + 1, SDN_STAMP_NOW() - sa1); }
        if (ntask)                                                           // workgroup-uniform
            run_group_tasks(s_task, s_steps, s_scan, ntask, oc, bound, max_steps, H, grid_f, n_step, xyzs, dirs, deltas,
                            grouped ? fs.slot_frame : nullptr, fcur, 8u);
        if (!grouped) break;
        fcur = block_min_256((frame != kNoFrame && frame > fcur) ? frame : kNoFrame, s_min4);   // (the barriers inside also fence the LDS caches and the task list)
    }
    __syncthreads();
    const unsigned long long sl0 = SDN_STAMP_NOW();
    if (live_idx) wg_live_append(s_steps, blockIdx.x * kGW, n_step, live_idx, live_count, s_scan);   // kernel-uniform condition
    if (threadIdx.x == 0) {
        SDN_STAMP_ADD(8 + 4, SDN_STAMP_NOW() - sl0);
#ifdef SDN_STAMPS
        SDN_STAMP_ADD(8 + 6, s_scan[6]); SDN_STAMP_ADD(8 + 7, s_scan[7]);
#endif
    }
    if (n >= n_alive) {
        const uint32_t slot = n_alive * n_step + (n - n_alive);  // spare lanes of the last blocks clear the alignment tail
        if (slot < M_pad) {
            xyzs[(size_t)slot * 3] = 0; xyzs[(size_t)slot * 3 + 1] = 0; xyzs[(size_t)slot * 3 + 2] = 0;
            dirs[(size_t)slot * 3] = 0; dirs[(size_t)slot * 3 + 1] = 0; dirs[(size_t)slot * 3 + 2] = 0;
            deltas[(size_t)slot * 2] = 0; deltas[(size_t)slot * 2 + 1] = 0;
        }
    }
}

__global__ void __launch_bounds__(256) k_composite_march_g(float T_thresh, int32_t *__restrict__ alive_a, int32_t *__restrict__ alive_b,
                                                           float *__restrict__ rays_t, const float *__restrict__ rays_o,
                                                           const float *__restrict__ rays_d, float bound, uint32_t max_steps, uint32_t H,
                                                           const uint8_t *__restrict__ grid, const float *__restrict__ fars,
                                                           const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                                           float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas,
                                                           float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image,
                                                           const uint32_t *__restrict__ cull, uint32_t *__restrict__ live_idx,
                                                           uint32_t *__restrict__ live_counts, int32_t *__restrict__ state,
                                                           int32_t *__restrict__ ticket, int32_t *__restrict__ trace, int32_t *__restrict__ snap,
                                                           FrameSel fs) {
    __shared__ uint4 s_cull4[256];
    __shared__ unsigned long long s_fine[kFineCacheCells];
    __shared__ uint32_t s_cnt[4], s_min4[4];
    __shared__ GroupTask s_task[kGW];
    __shared__ uint32_t s_ntask, s_steps[kGW], s_scan[8];
    __shared__ int s_last;
    const uint32_t n_alive = (uint32_t)state[0], n_step = (uint32_t)state[1], list_len = (uint32_t)state[8];
    const int32_t it = state[3];
    int32_t *__restrict__ alive = state[4] ? alive_b : alive_a;
    const bool march_next = (uint32_t)state[2] + n_step < (uint32_t)state[6];  // `while step < max_steps` admits another iteration
    const uint32_t n = blockIdx.x * kGW + threadIdx.x;
    bool survives = false;
    if (n_alive > 0 && blockIdx.x * kGW < list_len) {  // workgroup-uniform
        const unsigned long long sa0 = SDN_STAMP_NOW();
        const int index = n < list_len ? alive[n] : -1;
        float t_now = 0.0f;
        s_steps[threadIdx.x] = 0;
        // composite iteration `it` of this ray; survivors march iteration it + 1 below
        if (index >= 0) {
            survives = composite_ray(index, n_step, T_thresh, sigmas + (size_t)n * n_step, rgbs + (size_t)n * n_step * 3,
                                     deltas + (size_t)n * n_step * 2, rays_t, weights_sum, depth, image, &t_now);
            if (!survives) alive[n] = -1;
        }
        const bool marches = survives && march_next;
        const bool grouped = fs.n_frames > 1;  // kernel-uniform
        const uint32_t frame = (grouped && marches) ? (uint32_t)index / fs.rays_per_frame : (grouped ? kNoFrame : 0u);
        uint32_t fcur = grouped ? block_min_256(frame, s_min4) : 0u;
        while (fcur != kNoFrame) {   // one round per frame present among this workgroup's marching rays (see block_min_256)
            const uint8_t *grid_f = grouped ? frame_grid_uniform(fs, fcur) : grid;
            const uint32_t *cull_f = (grouped && cull) ? cull + (size_t)fcur * fs.cull_stride : cull;
            if (threadIdx.x == 0) { s_ntask = 0; s_scan[6] = 0; s_scan[7] = 0; }
            const unsigned long long sa1 = SDN_STAMP_NOW();
            OccCache oc;
            occ_cache_load<true>(cull_f, grid_f, s_cull4, s_fine, oc);
            if (!cull_f) __syncthreads();
            const unsigned long long sa2 = SDN_STAMP_NOW();
            if (marches && frame == fcur) {
                MarcherT<true> m;
                m.init(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, 0.0f, max_steps, 1u, H, grid_f);
                float *tend = state_tend(state);
                if (tend) tend += index;
                s_steps[threadIdx.x] = group_enlist(s_task, &s_ntask, oc, m, n, index, t_now, fars[index], tend, n_step, xyzs, dirs, deltas,
                                                    grouped ? fs.slot_frame : nullptr, fcur);
            }
            __syncthreads();
            const uint32_t ntask = s_ntask;
            if (threadIdx.x == 0) { SDN_STAMP_ADD(2, sa2 - sa1); SDN_STAMP_ADD(1, (sa1 - sa0) + (SDN_STAMP_NOW() - sa2)); }
            if (ntask)
                run_group_tasks(s_task, s_steps, s_scan, ntask, oc, bound, max_steps, H, grid_f, n_step, xyzs, dirs, deltas,
                                grouped ? fs.slot_frame : nullptr, fcur, 0u);
            if (!grouped) break;
            fcur = block_min_256((frame != kNoFrame && frame > fcur) ? frame : kNoFrame, s_min4);
        }
        __syncthreads();
        const unsigned long long sl0 = SDN_STAMP_NOW();
        wg_live_append(s_steps, blockIdx.x * kGW, n_step, live_idx, live_counts + it + 1, s_scan);
        if (threadIdx.x == 0) {
            SDN_STAMP_ADD(4, SDN_STAMP_NOW() - sl0);
#ifdef SDN_STAMPS
            SDN_STAMP_ADD(6, s_scan[6]); SDN_STAMP_ADD(7, s_scan[7]);
#endif
        }
    }
    // survivors of this workgroup -> global accumulator; the last workgroup of the launch advances the loop record
    const unsigned long long mask = __ballot(survives);
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (c) atomicAdd(state + 9, (int32_t)c);
        __threadfence();
        s_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    *ticket = 0;
    const int32_t call = state[7];
    state[7] = call + 1;
    if (state[0] > 0) {
        trace[2 * it] = state[0];
        trace[2 * it + 1] = state[1];
        state[2] += state[1];
        state[3] = it + 1;
        int32_t n_new = atomicAdd(state + 9, 0);
        state[9] = 0;
        if (state[2] >= state[6]) n_new = 0;
        state[0] = n_new;  // n_step stays 8: N / n_new >= 8 holds from here on
    }
    publish_snapshot(state, snap, call);
}

// image = image + (1 - weights_sum) * bg ; depth = clamp(depth - nears, 0) / (fars - nears)   (dnerf/renderer.py:378-379)
__global__ void __launch_bounds__(256) k_loop_finish(uint32_t N, const float *__restrict__ nears, const float *__restrict__ fars,
                                                     const float *__restrict__ weights_sum, const float *__restrict__ depth,
                                                     const float *__restrict__ image, float bg, float *__restrict__ image_out,
                                                     float *__restrict__ depth_out) {
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n >= N) return;
    const float w = (1 - weights_sum[n]) * bg;
    image_out[(size_t)n * 3] = image[(size_t)n * 3] + w;
    image_out[(size_t)n * 3 + 1] = image[(size_t)n * 3 + 1] + w;
    image_out[(size_t)n * 3 + 2] = image[(size_t)n * 3 + 2] + w;
    depth_out[n] = fmaxf(depth[n] - nears[n], 0.0f) / (fars[n] - nears[n]);
}

}  // namespace

namespace {
// The cooperative (16 lanes per ray) marcher for the FAST configuration with a constant step is an OPT-IN (SDN_GROUP_MARCH=1 in the
// environment): it produces the lane-per-ray kernels' samples bit for bit and is neither faster nor slower on the headline frame
// (profiles/r03_cooperative_marcher.txt, DESIGN.md "The cooperative marcher").  Default: the lane-per-ray kernels.
bool use_group_march(float bound, float dt_gamma, uint32_t C, uint32_t H) {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("SDN_GROUP_MARCH");
        on = (e && e[0] == '1') ? 1 : 0;
    }
    return on && dt_gamma == 0.0f && fast_config(bound, C, H);
}
}  // namespace

namespace sdn_int {

int loop_begin(uint32_t N, uint32_t max_steps, const float *nears, int32_t *alive_a, float *rays_t, float *weights_sum, float *depth,
               float *image, int32_t *state, int32_t *live_counts, uint32_t n_counters, void *mailbox, uint32_t frame_tag, float *rays_tend,
               hipStream_t st) {
    const uint32_t threads = N > n_counters ? N : n_counters;
    hipLaunchKernelGGL(k_loop_init, dim3(sdn_div_up(threads, 256u)), dim3(256), 0, st, N, max_steps, nears, alive_a, rays_t, weights_sum, depth,
                       image, state, live_counts, n_counters, (unsigned long long)(uintptr_t)mailbox, frame_tag, rays_tend);
    return sdn_launch_status();
}

// Culled start (see k_cull_start): after loop_begin and after the context's cull grid(s) are in place.  Returns 0 without doing
// anything when the configuration has no exact cull test (not the FAST configuration, no cull grid, no t_end cache, cooperative marcher).
int loop_cull_start(uint32_t N, const float *rays_o, const float *rays_d, const float *nears, const float *fars, float bound, float dt_gamma,
                    uint32_t C, uint32_t H, const uint32_t *cull, const FrameSel &fs, int32_t *alive_a, int32_t *alive_b, float *rays_tend,
                    int32_t *state, uint32_t *block_totals, int32_t *n_out, int32_t *trace, uint32_t max_steps, float *jump, hipStream_t st) {
    static int off = -1;
    if (off < 0) { const char *e = getenv("SDN_CULL_START"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off || !cull || !rays_tend || H != 128 || !fast_config(bound, C, H) || use_group_march(bound, dt_gamma, C, H)) return 0;
    static int no_jump = -1;
    if (no_jump < 0) { const char *e = getenv("SDN_CULL_JUMP"); no_jump = (e && e[0] == '0') ? 1 : 0; }
    hipLaunchKernelGGL(k_cull_start, dim3(sdn_div_up(N, 256u)), dim3(256), 0, st, N, rays_o, rays_d, nears, fars, cull, fs, alive_a, rays_tend, bound,
                       dt_gamma, max_steps, C, H, no_jump ? (float *)nullptr : jump, block_totals);
    const uint32_t nb = sdn_div_up(N, kScanBlock);
    hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(kScanBlock), 0, st, (const int32_t *)alive_a, N, (const uint32_t *)block_totals, alive_b, n_out,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, 4u);
    hipLaunchKernelGGL(k_cull_advance, dim3(1), dim3(64), 0, st, state, (const int32_t *)n_out, trace);
    return sdn_launch_status();
}

int loop_march(uint32_t bound_alive, const int32_t *alive_a, const int32_t *alive_b, const float *rays_t, const float *rays_o,
               const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
               const float *fars, float *xyzs, float *dirs, float *deltas, const uint32_t *cull, uint32_t *live_idx,
               uint32_t *live_counts, const int32_t *state, const FrameSel &fs, hipStream_t st, const float *jump) {
    const dim3 g(sdn_div_up(bound_alive + 128u, 256u)), b(256);
    static int no_jump = -1;
    if (no_jump < 0) { const char *e = getenv("SDN_CULL_JUMP"); no_jump = (e && e[0] == '0') ? 1 : 0; }
    if (no_jump) jump = nullptr;
    if (use_group_march(bound, dt_gamma, C, H)) {
        if (cull && H != 128) cull = nullptr;
        hipLaunchKernelGGL(k_march_rays_g, dim3(sdn_div_up(bound_alive + 128u, kGW)), b, 0, st, 0u, 0u, alive_a, rays_t, rays_o, rays_d, bound, max_steps, H, grid, fars, xyzs, dirs, deltas,
                           (const float *)nullptr, 0u, cull, live_idx, live_counts, state, alive_b, fs);
    } else if (fast_config(bound, C, H)) {
        if (cull && H != 128) cull = nullptr;
        hipLaunchKernelGGL(k_march_rays<true>, g, b, 0, st, 0u, 0u, alive_a, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars,
                           xyzs, dirs, deltas, (const float *)nullptr, 0u, cull, live_idx, live_counts, state, alive_b, fs, cull ? jump : (const float *)nullptr);
    } else {
        hipLaunchKernelGGL(k_march_rays<false>, g, b, 0, st, 0u, 0u, alive_a, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars,
                           xyzs, dirs, deltas, (const float *)nullptr, 0u, (const uint32_t *)nullptr, live_idx, live_counts, state, alive_b, fs,
                           (const float *)nullptr);
    }
    return sdn_launch_status();
}

int loop_composite_compact(uint32_t bound_alive, float T_thresh, int32_t *alive_a, int32_t *alive_b, float *rays_t, const float *sigmas,
                           const float *rgbs, const float *deltas, float *weights_sum, float *depth, float *image, int32_t *state,
                           uint32_t *block_totals, int32_t *n_out, int32_t *trace, int32_t *snap, hipStream_t st, bool freeze) {
    const dim3 g(sdn_div_up(bound_alive, 256u)), b(256);
    if (bound_alive > 65536u) {
        // many rays (the first one or two iterations): every scatter workgroup would re-sum thousands of 256-ray totals;
        // use the 1024-ray count / scatter pair and a separate one-thread advance instead.  (Measured again in round 3 with the fused pair up
        // to 131 072 / 262 144 / 524 288 rays: 0.449 / 0.451 / 0.451 ms per frame against 0.441 -- every workgroup of the fused scatter pays a
        // device-scope fence for the last-workgroup election.)
        static_assert(kScanBlock == 1024, "the compositing kernel's 256-ray survivor counts are summed four to a scatter block");
        hipLaunchKernelGGL(k_composite_rays, g, b, 0, st, 0u, 0u, T_thresh, alive_a, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                           (const int32_t *)state, alive_b, block_totals);          // (+ survivors per 256 rays: no separate count launch)
        const uint32_t nb = sdn_div_up(bound_alive, kScanBlock);
        hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(kScanBlock), 0, st, (const int32_t *)alive_a, 0u, (const uint32_t *)block_totals,
                           alive_b, snap + 8, (const int32_t *)state, (const int32_t *)alive_b, alive_a, 4u);
        hipLaunchKernelGGL(k_loop_advance, dim3(1), dim3(64), 0, st, state, (const int32_t *)(snap + 8), trace, snap, freeze ? 1 : 0);
        return sdn_launch_status();
    }
    hipLaunchKernelGGL(k_composite_rays, g, b, 0, st, 0u, 0u, T_thresh, alive_a, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                       (const int32_t *)state, alive_b, block_totals);
    // n_out doubles as the ticket counter (zero between launches)
    hipLaunchKernelGGL(k_scatter_advance, g, b, 0, st, alive_a, alive_b, (const uint32_t *)block_totals, state, n_out, trace, snap, freeze ? 1 : 0);
    return sdn_launch_status();
}


int loop_steady_begin(uint32_t bound_alive, const int32_t *alive_a, const int32_t *alive_b, const float *rays_t, const float *rays_o,
                      const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
                      const float *fars, float *xyzs, float *dirs, float *deltas, const uint32_t *cull, uint32_t *live_idx,
                      uint32_t *live_counts, int32_t *state, const FrameSel &fs, hipStream_t st, bool frozen_already) {
    if (!frozen_already) hipLaunchKernelGGL(k_steady_begin, dim3(1), dim3(64), 0, st, state);
    return loop_march(bound_alive, alive_a, alive_b, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas,
                      cull, live_idx, live_counts, state, fs, st, nullptr);
}

int loop_composite_march(uint32_t bound_list, float T_thresh, int32_t *alive_a, int32_t *alive_b, float *rays_t, const float *rays_o,
                         const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t *grid,
                         const float *fars, const float *sigmas, const float *rgbs, float *xyzs, float *dirs, float *deltas,
                         float *weights_sum, float *depth, float *image, const uint32_t *cull, uint32_t *live_idx, uint32_t *live_counts,
                         int32_t *state, int32_t *ticket, int32_t *trace, int32_t *snap, const FrameSel &fs, hipStream_t st) {
    const dim3 g(sdn_div_up(bound_list, 256u)), b(256);
    if (use_group_march(bound, dt_gamma, C, H)) {
        if (cull && H != 128) cull = nullptr;
        hipLaunchKernelGGL(k_composite_march_g, dim3(sdn_div_up(bound_list, kGW)), b, 0, st, T_thresh, alive_a, alive_b, rays_t, rays_o, rays_d, bound, max_steps, H, grid, fars,
                           sigmas, rgbs, xyzs, dirs, deltas, weights_sum, depth, image, cull, live_idx, live_counts, state, ticket, trace, snap, fs);
    } else if (fast_config(bound, C, H)) {
        if (cull && H != 128) cull = nullptr;
        hipLaunchKernelGGL(k_composite_march<true>, g, b, 0, st, T_thresh, alive_a, alive_b, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                           grid, fars, sigmas, rgbs, xyzs, dirs, deltas, weights_sum, depth, image, cull, live_idx, live_counts, state, ticket,
                           trace, snap, fs);
    } else {
        hipLaunchKernelGGL(k_composite_march<false>, g, b, 0, st, T_thresh, alive_a, alive_b, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C,
                           H, grid, fars, sigmas, rgbs, xyzs, dirs, deltas, weights_sum, depth, image, (const uint32_t *)nullptr, live_idx,
                           live_counts, state, ticket, trace, snap, fs);
    }
    return sdn_launch_status();
}

int loop_finish(uint32_t N, const float *nears, const float *fars, const float *weights_sum, const float *depth, const float *image, float bg,
                float *image_out, float *depth_out, hipStream_t st) {
    hipLaunchKernelGGL(k_loop_finish, dim3(sdn_div_up(N, 256u)), dim3(256), 0, st, N, nears, fars, weights_sum, depth, image, bg, image_out,
                       depth_out);
    return sdn_launch_status();
}

// with_image: the buffer has sdn_cull_grid_bytes() behind it (marks + meta + packed fine bits); false: marks + meta only
int build_cull(const uint8_t *bitfield, uint32_t *cull_bits, hipStream_t st, bool with_image) {
    hipLaunchKernelGGL(k_cull_meta_init, dim3(1), dim3(64), 0, st, cull_bits, 1u, 0u);
    hipLaunchKernelGGL(k_build_cull_grid, dim3(kCullRes * kCullRes * kCullRes / 256), dim3(256), 0, st, bitfield, cull_bits, FrameSel());
    if (with_image) hipLaunchKernelGGL(k_build_fine_image, dim3(kFineCacheCells / 256), dim3(256), 0, st, bitfield, cull_bits, FrameSel());
    return sdn_launch_status();
}

// cull grids the caller kept per occupancy slice -> the context's contiguous copy (one launch, 4 KiB per frame)
struct CullCopy { const uint32_t *src[SDN_MAX_GROUP_FRAMES]; };
__global__ void __launch_bounds__(256) k_copy_cull_grids(CullCopy srcs, uint32_t *__restrict__ dst, uint32_t words, uint32_t stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) dst[(size_t)blockIdx.y * stride + i] = srcs.src[blockIdx.y][i];
}

int copy_cull(const void *const *prebuilt, uint32_t n_frames, uint32_t *cull_bits, hipStream_t st) {
    CullCopy c{};
    for (uint32_t f = 0; f < n_frames && f < (uint32_t)SDN_MAX_GROUP_FRAMES; f++) c.src[f] = (const uint32_t *)prebuilt[f];
    const uint32_t words = sdn_cull_grid_bytes() / 4u;
    hipLaunchKernelGGL(k_copy_cull_grids, dim3(sdn_div_up(words, 256u), n_frames), dim3(256), 0, st, c, cull_bits, words, n_frames > 1 ? words : 0u);
    return sdn_launch_status();
}

int build_cull_group(const FrameSel &fs, uint32_t *cull_bits, hipStream_t st) {
    hipLaunchKernelGGL(k_cull_meta_init, dim3(1), dim3(64), 0, st, cull_bits, fs.n_frames, fs.cull_stride);
    hipLaunchKernelGGL(k_build_cull_grid, dim3(kCullRes * kCullRes * kCullRes / 256, fs.n_frames), dim3(256), 0, st, (const uint8_t *)nullptr,
                       cull_bits, fs);
    hipLaunchKernelGGL(k_build_fine_image, dim3(kFineCacheCells / 256, fs.n_frames), dim3(256), 0, st, (const uint8_t *)nullptr, cull_bits, fs);
    return sdn_launch_status();
}

FrameSel frame_sel(const SdnRenderCtx *c) {
    FrameSel fs;
    if (c->n_group_frames > 1) {
        fs.n_frames = c->n_group_frames;
        fs.rays_per_frame = c->rays_per_frame;
        fs.cull_stride = sdn_cull_grid_bytes() / 4u;
        for (uint32_t f = 0; f < c->n_group_frames && f < SDN_MAX_GROUP_FRAMES; f++) fs.grid[f] = c->frame_bitfield[f];
        fs.slot_frame = c->slot_frame;
    }
    return fs;
}

}  // namespace sdn_int

namespace {
}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int sdn_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                           float *nears, float *fars, void *stream) {
    if (N == 0) return 0;
    if (!rays_o || !rays_d || !aabb || !nears || !fars) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_near_far_from_aabb, dim3(sdn_div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, aabb, N,
                       min_near, nears, fars);
    return sdn_launch_status();
}

int sdn_sph_from_ray(const float *rays_o, const float *rays_d, float radius, uint32_t N, float *coords, void *stream) {
    if (N == 0) return 0;
    if (!rays_o || !rays_d || !coords) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_sph_from_ray, dim3(sdn_div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, radius, N, coords);
    return sdn_launch_status();
}

int sdn_morton3D(const int32_t *coords, uint32_t N, int32_t *indices, void *stream) {
    if (N == 0) return 0;
    if (!coords || !indices) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_morton3D, dim3(sdn_div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, coords, N, indices);
    return sdn_launch_status();
}

int sdn_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords, void *stream) {
    if (N == 0) return 0;
    if (!coords || !indices) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_morton3D_invert, dim3(sdn_div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, indices, N, coords);
    return sdn_launch_status();
}

int sdn_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, void *stream) {
    if (N == 0) return 0;
    if (!grid || !bitfield) return SDN_E_BADARG;
    if (((uintptr_t)grid & 15u) != 0) return SDN_E_BADARG;  // float4 loads
    hipLaunchKernelGGL(k_packbits, dim3(sdn_div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, (const float4 *)grid, N, density_thresh,
                       bitfield);
    return sdn_launch_status();
}

static uint64_t train_scratch_words(uint32_t N) { return (uint64_t)N + sdn_div_up(N, kScanBlock) + 4u; }   // num_steps, block totals, bases

uint64_t sdn_march_rays_train_scratch_bytes(uint32_t N, uint32_t max_steps) {
    // scan workspace | cull grid of the time slice | t of every sample [N, max_steps]
    const uint64_t head = (train_scratch_words(N) * sizeof(uint32_t) + 15u) & ~(uint64_t)15u;
    const uint64_t cull = ((uint64_t)(kCullWords + 8u) * sizeof(uint32_t) + 15u) & ~(uint64_t)15u;
    return head + cull + (uint64_t)N * max_steps * sizeof(float);
}

int sdn_march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma,
                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *nears,
                         const float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                         const float *noises, void *scratch, void *stream) {
    return sdn_int::march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises,
                                     scratch, nullptr, (hipStream_t)stream);
}

}  // extern "C"

// prebuilt_cull: the cull grid of `grid` from sdn_build_cull_grid (a caller that marches the same occupancy slice step after step
// keeps it), or nullptr: built here into the scratch
int sdn_int::march_rays_train(const float *rays_o, const float *rays_d, const uint8_t *grid, float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                              uint32_t C, uint32_t H, uint32_t M, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                              int32_t *rays, int32_t *counter, const float *noises, void *scratch, const void *prebuilt_cull, hipStream_t st) {
    if (N == 0) return 0;
    if (!rays_o || !rays_d || !grid || !nears || !fars || !xyzs || !dirs || !deltas || !rays || !counter || !noises || !scratch)
        return SDN_E_BADARG;
    if (C == 0 || C > 16 || H == 0 || max_steps == 0 || ((uintptr_t)scratch & 15u) || ((uintptr_t)prebuilt_cull & 15u)) return SDN_E_BADARG;
    uint32_t *num_steps = (uint32_t *)scratch;
    const uint32_t nb = sdn_div_up(N, kScanBlock);
    uint32_t *block_totals = num_steps + N;
    uint32_t *base_out = block_totals + nb;
    const uint64_t head = (train_scratch_words(N) * sizeof(uint32_t) + 15u) & ~(uint64_t)15u;
    const uint64_t cull_bytes = ((uint64_t)(kCullWords + 8u) * sizeof(uint32_t) + 15u) & ~(uint64_t)15u;
    const uint32_t *cull = (const uint32_t *)((unsigned char *)scratch + head);
    float *sample_t = (float *)((unsigned char *)scratch + head + cull_bytes);
    const bool fast = fast_config(bound, C, H);
    const bool use_cull = fast && H == 128;
    if (use_cull && prebuilt_cull) {
        cull = (const uint32_t *)prebuilt_cull;
    } else if (use_cull) {
        int rc = sdn_int::build_cull(grid, (uint32_t *)((unsigned char *)scratch + head), st, false);   // (the scratch has no room for the image)
        if (rc) return rc;
    }
    if (fast && dt_gamma == 0.0f && H <= 256u)   // constant step: one WAVE per ray (k_march_train_count_wave), same samples bit for bit
        hipLaunchKernelGGL(k_march_train_count_wave, dim3(sdn_div_up(N, 4u)), dim3(256), 0, st, rays_o, rays_d, grid, bound, max_steps, N, H, nears,
                           fars, noises, num_steps, use_cull ? cull : nullptr, sample_t);
    else if (fast) hipLaunchKernelGGL(k_march_train_count<true>, dim3(sdn_div_up(N, 256u)), dim3(256), 0, st, rays_o, rays_d, grid, bound, dt_gamma,
                                 max_steps, N, C, H, nears, fars, noises, num_steps, use_cull ? cull : nullptr, sample_t);
    else hipLaunchKernelGGL(k_march_train_count<false>, dim3(sdn_div_up(N, 256u)), dim3(256), 0, st, rays_o, rays_d, grid, bound, dt_gamma,
                            max_steps, N, C, H, nears, fars, noises, num_steps, (const uint32_t *)nullptr, sample_t);
    // The reference writes ray records at rays[atomicAdd(counter+1, 1)] and points at atomicAdd(counter, n):
    // both bases are read on the device from the counter the caller hands in (zeroed by dnerf/renderer.py:291-292).
    const bool small = N <= 16u * kScanBlock;
    if (small) {
        hipLaunchKernelGGL(k_march_train_offsets_small, dim3(1), dim3(kScanBlock), 0, st, num_steps, N, rays, counter, base_out);
    } else {
        hipLaunchKernelGGL(k_scan_block_totals, dim3(nb), dim3(kScanBlock), 0, st, num_steps, N, block_totals);
        hipLaunchKernelGGL(k_march_train_offsets, dim3(nb), dim3(kScanBlock), 0, st, num_steps, N, block_totals, rays, counter, base_out);
    }
    if (fast) hipLaunchKernelGGL(k_march_train_emit<true>, dim3(sdn_div_up(N, 4u)), dim3(256), 0, st, rays_o, rays_d, bound, dt_gamma, max_steps, N, C, H, M,
                                 nears, noises, rays, base_out, sample_t, xyzs, dirs, deltas);
    else hipLaunchKernelGGL(k_march_train_emit<false>, dim3(sdn_div_up(N, 4u)), dim3(256), 0, st, rays_o, rays_d, bound, dt_gamma, max_steps, N, C, H, M,
                            nears, noises, rays, base_out, sample_t, xyzs, dirs, deltas);
    if (!small) hipLaunchKernelGGL(k_march_train_finish, dim3(1), dim3(64), 0, st, counter, base_out, N);
    return sdn_launch_status();
}

extern "C" {

int sdn_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, uint32_t M,
                                     uint32_t N, float T_thresh, float *weights_sum, float *depth, float *image, void *stream) {
    if (N == 0) return 0;
    // (M == 0: empty sample tensors have no address; every ray then takes the kernel's `offset + num_steps > M` exit and reads none)
    if ((M != 0 && (!sigmas || !rgbs || !deltas)) || !rays || !weights_sum || !depth || !image) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_composite_train_fwd, dim3(sdn_div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, sigmas, rgbs, deltas, rays, M, N,
                       T_thresh, weights_sum, depth, image);
    return sdn_launch_status();
}

int sdn_composite_whole_rays(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, const float *nears, uint32_t M,
                             uint32_t N, float T_thresh, float *weights_sum, float *depth, float *image, void *stream) {
    if (N == 0) return 0;
    if (!sigmas || !rgbs || !deltas || !rays || !nears || !weights_sum || !depth || !image) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_composite_whole_rays, dim3(sdn_div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, sigmas, rgbs, deltas, rays, nears, M, N,
                       T_thresh, weights_sum, depth, image);
    return sdn_launch_status();
}

int sdn_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image, const float *sigmas, const float *rgbs,
                                      const float *deltas, const int32_t *rays, const float *weights_sum, const float *image,
                                      uint32_t M, uint32_t N, float T_thresh, float *grad_sigmas, float *grad_rgbs, void *stream) {
    if (N == 0) return 0;
    if (M == 0) return 0;       // no sample, no gradient entry to write
    if (!grad_weights_sum || !grad_image || !sigmas || !rgbs || !deltas || !rays || !weights_sum || !image || !grad_sigmas || !grad_rgbs)
        return SDN_E_BADARG;
    hipLaunchKernelGGL(k_composite_train_bwd, dim3(sdn_div_up(N, 256u)), dim3(256), 0, (hipStream_t)stream, grad_weights_sum, grad_image,
                       sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs);
    return sdn_launch_status();
}

static int launch_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                             const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                             const uint8_t *grid, const float *fars, float *xyzs, float *dirs, float *deltas, const float *noises,
                             uint32_t M_pad, const uint32_t *cull, uint32_t *live_idx, uint32_t *live_count, hipStream_t st) {
    if (n_alive == 0 || n_step == 0) {
        // no sample slots: the M_pad padding slots are still the caller's to read -- the reference hands out zero-filled buffers
        // (raymarching.py:333-335), and no kernel runs here that would clear them
        if (M_pad && xyzs && dirs && deltas) {
            if (hipMemsetAsync(xyzs, 0, (size_t)M_pad * 12, st) != hipSuccess || hipMemsetAsync(dirs, 0, (size_t)M_pad * 12, st) != hipSuccess ||
                hipMemsetAsync(deltas, 0, (size_t)M_pad * 8, st) != hipSuccess)
                return sdn_launch_status();
        }
        return 0;
    }
    if (!rays_alive || !rays_t || !rays_o || !rays_d || !grid || !fars || !xyzs || !dirs || !deltas) return SDN_E_BADARG;
    if (C == 0 || C > 16 || H == 0 || max_steps == 0) return SDN_E_BADARG;
    if ((live_idx == nullptr) != (live_count == nullptr)) return SDN_E_BADARG;
    const uint32_t base = n_alive * n_step;
    if (M_pad < base) M_pad = base;
    const uint32_t threads = n_alive + (M_pad - base);  // one lane per alive ray + one per tail slot
    const dim3 g(sdn_div_up(threads, 256u)), b(256);
    if (use_group_march(bound, dt_gamma, C, H)) {
        if (cull && H != 128) cull = nullptr;  // the cull grid is built for the 128^3 grid only
        hipLaunchKernelGGL(k_march_rays_g, dim3(sdn_div_up(threads, kGW)), b, 0, st, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, max_steps, H, grid, fars, xyzs,
                           dirs, deltas, noises, M_pad, cull, live_idx, live_count, (const int32_t *)nullptr, (const int32_t *)nullptr, FrameSel());
    } else if (fast_config(bound, C, H)) {
        if (cull && H != 128) cull = nullptr;  // the cull grid is built for the 128^3 grid only
        hipLaunchKernelGGL(k_march_rays<true>, g, b, 0, st, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                           grid, fars, xyzs, dirs, deltas, noises, M_pad, cull, live_idx, live_count, (const int32_t *)nullptr,
                           (const int32_t *)nullptr, FrameSel(), (const float *)nullptr);
    } else {
        hipLaunchKernelGGL(k_march_rays<false>, g, b, 0, st, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                           grid, fars, xyzs, dirs, deltas, noises, M_pad, (const uint32_t *)nullptr, live_idx, live_count,
                           (const int32_t *)nullptr, (const int32_t *)nullptr, FrameSel(), (const float *)nullptr);
    }
    return sdn_launch_status();
}

int sdn_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                   const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                   const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas,
                   const float *noises, void *stream) {
    (void)nears;  // unused by the reference kernel as well (raymarching.cu:737)
    return launch_march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs,
                             deltas, noises, 0, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int sdn_march_rays_ex(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                      const float *rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                      const uint8_t *grid, const float *fars, float *xyzs, float *dirs, float *deltas, const float *noises,
                      uint32_t M_pad, const uint8_t *cull_grid, uint32_t *live_idx, uint32_t *live_count, void *stream) {
    return launch_march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs,
                             deltas, noises, M_pad, (const uint32_t *)cull_grid, live_idx, live_count, (hipStream_t)stream);
}

#ifdef SDN_STAMPS
// diagnostic build only: copies out and clears the stamp sums (see g_stamps)
int sdn_debug_stamps(unsigned long long *out32) {
    if (!out32) return SDN_E_BADARG;
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32) != hipSuccess) return sdn_launch_status();
    const unsigned long long zeros[32] = {};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros));
}
#endif

uint32_t sdn_cull_grid_bytes(void) { return kCullImageWord * 4 + kFineCacheCells * 8; }  // marks + bounding-box record + packed fine bits

int sdn_build_cull_grid(const uint8_t *bitfield, uint32_t H, uint8_t *cull_grid, void *stream) {
    if (!bitfield || !cull_grid) return SDN_E_BADARG;
    if (H != 128) return SDN_E_UNSUPPORTED;
    if (((uintptr_t)bitfield & 7u) != 0 || ((uintptr_t)cull_grid & 15u) != 0) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_cull_meta_init, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint32_t *)cull_grid, 1u, 0u);
    hipLaunchKernelGGL(k_build_cull_grid, dim3(kCullRes * kCullRes * kCullRes / 256), dim3(256), 0, (hipStream_t)stream, bitfield,
                       (uint32_t *)cull_grid, FrameSel());
    hipLaunchKernelGGL(k_build_fine_image, dim3(kFineCacheCells / 256), dim3(256), 0, (hipStream_t)stream, bitfield, (uint32_t *)cull_grid, FrameSel());
    return sdn_launch_status();
}

int sdn_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t, const float *sigmas,
                       const float *rgbs, const float *deltas, float *weights_sum, float *depth, float *image, void *stream) {
    if (n_alive == 0 || n_step == 0) return 0;
    if (!rays_alive || !rays_t || !sigmas || !rgbs || !deltas || !weights_sum || !depth || !image) return SDN_E_BADARG;
    hipLaunchKernelGGL(k_composite_rays, dim3(sdn_div_up(n_alive, 256u)), dim3(256), 0, (hipStream_t)stream, n_alive, n_step, T_thresh,
                       rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, (const int32_t *)nullptr, (int32_t *)nullptr,
                       (uint32_t *)nullptr);
    return sdn_launch_status();
}

uint64_t sdn_compact_alive_scratch_bytes(uint32_t n) { return (uint64_t)sdn_div_up(n, kScanBlock) * sizeof(uint32_t); }

int sdn_compact_alive(const int32_t *in, uint32_t n, int32_t *out, int32_t *n_out, void *scratch, void *stream) {
    if (!n_out) return SDN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return (int)hipMemsetAsync(n_out, 0, sizeof(int32_t), st);
    if (!in || !out || !scratch) return SDN_E_BADARG;
    const uint32_t nb = sdn_div_up(n, kScanBlock);
    hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(kScanBlock), 0, st, in, n, (uint32_t *)scratch, (const int32_t *)nullptr,
                       (const int32_t *)nullptr);
    hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(kScanBlock), 0, st, in, n, (const uint32_t *)scratch, out, n_out,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, 1u);
    return sdn_launch_status();
}

}  // extern "C"
