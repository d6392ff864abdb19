// Occupancy-grid cell centres for the density-grid queries (dnerf/renderer.py:480-490 == :517-524): shared by the CELLS variants of the
// fused field kernels (field.hip: fp16, field_f32.hip: fp32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdn_cells {

// Morton code -> one coordinate (bits 0, 3, 6, ...): raymarching.cu:282-289
__device__ __forceinline__ uint32_t compact_bits3(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    return (x | (x >> 16)) & 0x0000ffffu;
}

// counter-based uniform [0,1): PCG output permutation of (seed, counter); 24 random mantissa bits
__device__ __forceinline__ float cell_uniform(uint32_t seed, uint32_t counter) {
    uint32_t v = (counter ^ seed) * 747796405u + 2891336453u;
    v = ((v >> ((v >> 28u) + 4u)) ^ v) * 277803737u;
    v = (v >> 22u) ^ v;
    v = (v ^ seed) * 747796405u + 2891336453u;
    v = ((v >> ((v >> 28u) + 4u)) ^ v) * 277803737u;
    v = (v >> 22u) ^ v;
    return (float)(v >> 8) * (1.0f / 16777216.0f);
}

// The jittered centre of Morton cell `cell` along dimension d, the reference's fp32 operations in the reference's order:
//   xyzs = 2 * coords / (grid_size - 1) - 1;  cas_xyzs = xyzs * (bound - half_grid);  cas_xyzs += (rand * 2 - 1) * half_grid
// (cell_inv = 1 / (grid_size - 1) in fp32: torch divides a tensor by a host scalar as a multiplication by its reciprocal)
__device__ __forceinline__ float cell_coord(uint32_t cell, int d, float r, float cell_inv, float cell_span, float cell_half) {
    const float c = (float)compact_bits3(cell >> d);
    return ((2.0f * c) * cell_inv - 1.0f) * cell_span + (r * 2.0f - 1.0f) * cell_half;
}

}  // namespace sdn_cells
