// Level table + index arithmetic of the multiresolution grid, shared by gridencoder.hip and field.hip.
// Reference: gridencoder/src/gridencoder.cu:39-84 (smoothstep, fast_hash, get_grid_index), :138-139 (scale / resolution).
#pragma once
#include <math.h>

#include "sdn_common.h"

namespace sdn_grid {

constexpr uint32_t kMaxLevels = 32;

struct LevelParams {
    uint32_t offset[kMaxLevels];        // row offset of the level (rows, not elements)
    uint32_t hashmap_size[kMaxLevels];  // rows in the level
    float scale[kMaxLevels];            // exp2f(level*S)*H - 1
    uint32_t resolution[kMaxLevels];    // ceil(scale) + 1
};

__device__ __forceinline__ float smoothstep_(float v) { return v * v * (3.0f - 2.0f * v); }
__device__ __forceinline__ float smoothstep_derivative_(float v) { return 6 * v * (1.0f - v); }

// gridencoder.cu:50-63
template <uint32_t D>
__device__ __forceinline__ uint32_t fast_hash(const uint32_t (&pos_grid)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t result = 0;
    #pragma unroll
    for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * primes[i];
    return result;
}

// gridencoder.cu:66-84 (returns the element index of channel 0 of the row)
template <uint32_t D, uint32_t C>
__device__ __forceinline__ uint32_t grid_index(uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution,
                                               const uint32_t (&pos_grid)[D]) {
    uint32_t stride = 1, index = 0;
    #pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pos_grid[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash<D>(pos_grid);
    // `index % hashmap_size` without the 30-instruction software division in the common cases (hashmap_size is uniform over the
    // workgroup): a power-of-two row count is an AND; a dense level holds every corner, so its index is already in range
    if ((hashmap_size & (hashmap_size - 1u)) == 0u) index &= hashmap_size - 1u;
    else if (index >= hashmap_size) index %= hashmap_size;
    return index * C;
}

inline int fill_levels(LevelParams &lp, const int32_t *offsets_host, uint32_t L, float S, uint32_t H) {
    if (L == 0 || L > kMaxLevels) return SDN_E_UNSUPPORTED;
    for (uint32_t l = 0; l < L; l++) {
        lp.offset[l] = (uint32_t)offsets_host[l];
        lp.hashmap_size[l] = (uint32_t)(offsets_host[l + 1] - offsets_host[l]);
        if (lp.hashmap_size[l] == 0) return SDN_E_BADARG;
        // gridencoder.cu:138-139, evaluated once on the host instead of per thread
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;
        lp.scale[l] = scale;
        lp.resolution[l] = (uint32_t)ceil((double)scale) + 1;
    }
    return 0;
}

}  // namespace sdn_grid
