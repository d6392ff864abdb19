// Density-grid maintenance on the device (reference: NeRFRenderer.update_extra_state, dnerf/renderer.py:453-555).
//
// The reference evaluates the density network on every cell of every time slice (64 x 128^3 = 134 M points on the first 16
// calls, half of that afterwards) through the op-by-op network, fills a full-size tmp_grid, applies the EMA / maximum update with
// boolean-mask indexing, takes the mean with a host read-back and packs the bitfield slice by slice.  Here:
//   * sdn_density_query_cells_f16: one launch of the fused field kernel (CELLS variant, field.hip) per time slice -- the cell
//     centres are built and jittered in the kernel, nothing but sigma * density_scale is written;
//   * sdn_density_grid_ema: density = max(density * decay, tmp) where both are >= 0, plus the running sum of clamp(density, 0)
//     for the mean (fp64 accumulator on the device);
//   * sdn_density_grid_pack: threshold = min(mean, density_thresh) taken on the device, bitfield for all slices in one pass.
// No host synchronisation anywhere; one time slice of tmp_grid (8 MiB) is live at a time instead of 512 MiB.
#include "sdn_common.h"
#include "sdn_internal.h"

namespace {

constexpr int kEmaBlock = 256;

constexpr int kEmaPerThread = 8;   // float4s per thread: few, fat workgroups -- the fp64 atomics on one address serialise (2048 of them cost ~20 us)

// dnerf/renderer.py:536-538.  Block partial sums reduced in fp64, one atomic per block.
__global__ void __launch_bounds__(kEmaBlock) k_density_ema(float4 *__restrict__ grid, const float4 *__restrict__ tmp, uint32_t n4, float decay,
                                                           double *__restrict__ sum) {
    double v = 0.0;
    #pragma unroll
    for (int k = 0; k < kEmaPerThread; k++) {
        const uint32_t i = threadIdx.x + (blockIdx.x * kEmaPerThread + k) * kEmaBlock;
        if (i < n4) {
            float4 g = grid[i];
            const float4 t = tmp[i];
            // valid = (grid >= 0) & (tmp >= 0) -- false for NaN on either side, as in the reference's mask
            if (g.x >= 0 && t.x >= 0) g.x = fmaxf(g.x * decay, t.x);
            if (g.y >= 0 && t.y >= 0) g.y = fmaxf(g.y * decay, t.y);
            if (g.z >= 0 && t.z >= 0) g.z = fmaxf(g.z * decay, t.z);
            if (g.w >= 0 && t.w >= 0) g.w = fmaxf(g.w * decay, t.w);
            grid[i] = g;
            v += (double)((fmaxf(g.x, 0.0f) + fmaxf(g.y, 0.0f)) + (fmaxf(g.z, 0.0f) + fmaxf(g.w, 0.0f)));
        }
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __shared__ double s_part[kEmaBlock / 64];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = 0;
        #pragma unroll
        for (int w = 0; w < kEmaBlock / 64; w++) b += s_part[w];
        atomicAdd(sum, b);
    }
}

// dnerf/renderer.py:539-545 + raymarching.cu:268-289: mean -> threshold -> one byte per 8 cells (bit i%8 of byte i/8)
__global__ void __launch_bounds__(256) k_density_pack(const float4 *__restrict__ grid, uint32_t n8, const double *__restrict__ sum,
                                                      double cells, float density_thresh, float *__restrict__ mean_out,
                                                      uint8_t *__restrict__ bitfield) {
    const float mean = (float)(*sum / cells);
    const float thresh = fminf(mean, density_thresh);
    const uint32_t n = threadIdx.x + blockIdx.x * blockDim.x;
    if (n == 0 && mean_out) {
        mean_out[0] = mean;
        mean_out[1] = thresh;
    }
    if (n >= n8) return;
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

}  // namespace

extern "C" {

int sdn_density_query_cells_f16(const int32_t *cells, const uint32_t *cell_count, uint32_t n, const float *noise, uint32_t seed,
                                uint32_t grid_size, float cas_bound, const void *weights, const float *bias0, const void *table,
                                const int32_t *offsets_host, float S, uint32_t H, float bound, float density_scale, int zero_deform,
                                float *tmp_slice, void *stream) {
    if (n == 0) return 0;
    if (!weights || !bias0 || !table || !offsets_host || !tmp_slice) return SDN_E_BADARG;
    if ((cells == nullptr) != (cell_count == nullptr)) return SDN_E_BADARG;
    if (grid_size < 2 || grid_size > 1024 || !(cas_bound > 0)) return SDN_E_BADARG;
    // without a list, slot p IS the Morton index: n may not exceed the grid
    if (!cells && (uint64_t)n > (uint64_t)grid_size * grid_size * grid_size) return SDN_E_BADARG;
    if (((uintptr_t)weights & 15u) != 0 || ((uintptr_t)table & 3u) != 0) return SDN_E_BADARG;
    return sdn_int::field_cells_f16(cells, cell_count, n, noise, seed, grid_size, cas_bound, weights, bias0, table, offsets_host, S, H, bound,
                                    density_scale, zero_deform, tmp_slice, (hipStream_t)stream);
}

int sdn_density_grid_ema(float *density_grid, const float *tmp_grid, uint64_t n, float decay, double *sum, void *stream) {
    if (n == 0) return 0;
    if (!density_grid || !tmp_grid || !sum) return SDN_E_BADARG;
    if ((n & 3u) != 0 || (n >> 2) > 0xFFFFFFFFull || (((uintptr_t)density_grid | (uintptr_t)tmp_grid) & 15u) != 0) return SDN_E_BADARG;
    const uint32_t n4 = (uint32_t)(n >> 2);
    hipLaunchKernelGGL(k_density_ema, dim3(sdn_div_up(n4, (uint32_t)(kEmaBlock * kEmaPerThread))), dim3(kEmaBlock), 0, (hipStream_t)stream, (float4 *)density_grid,
                       (const float4 *)tmp_grid, n4, decay, sum);
    return sdn_launch_status();
}

int sdn_density_grid_pack(const float *density_grid, uint64_t n, const double *sum, float density_thresh, float *mean_out,
                          uint8_t *bitfield, void *stream) {
    if (n == 0) return 0;
    if (!density_grid || !sum || !bitfield) return SDN_E_BADARG;
    if ((n & 7u) != 0 || (n >> 3) > 0xFFFFFFFFull || ((uintptr_t)density_grid & 15u) != 0) return SDN_E_BADARG;
    const uint32_t n8 = (uint32_t)(n >> 3);
    hipLaunchKernelGGL(k_density_pack, dim3(sdn_div_up(n8, 256u)), dim3(256), 0, (hipStream_t)stream, (const float4 *)density_grid, n8, sum,
                       (double)n, density_thresh, mean_out, bitfield);
    return sdn_launch_status();
}

}  // extern "C"
