// hashgrid_dev.h -- device helpers shared by the hash-grid kernels (hashgrid.hip, hashgrid_binned.hip).
// Arithmetic of tiny-cuda-nn's grid.h (grid_scale / pos_fract / grid_index, coherent prime hash); see hashgrid.hip.
#pragma once
#include "us_common.h"
#include <math.h>

struct LevelTable {                 // passed by value: ~400 B of kernel arguments
    float    scale[US_MAX_LEVELS];
    uint32_t res[US_MAX_LEVELS];
    uint32_t off[US_MAX_LEVELS + 1];
};

static inline LevelTable make_table(const us_grid_desc* d) {
    LevelTable t;
    for (int l = 0; l < US_MAX_LEVELS; ++l) { t.scale[l] = d->scale[l]; t.res[l] = d->resolution[l]; }
    for (int l = 0; l <= US_MAX_LEVELS; ++l) t.off[l] = d->offset[l];
    return t;
}

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
struct LevelGeom {          // wave-uniform
    float scale; uint32_t res; uint32_t hs; bool hashed; uint32_t res2;
};

__device__ __forceinline__ LevelGeom level_geom(const LevelTable& t, uint32_t level) {
    LevelGeom g;
    g.scale = t.scale[level]; g.res = t.res[level]; g.hs = t.off[level + 1] - t.off[level];
    // tcnn grid_index: dims are accumulated while stride <= hashmap_size; the hash replaces the sum when the
    // final stride exceeds hashmap_size.  For a Hash grid that is exactly "res^3 (uint32) > hashmap_size", also
    // when the loop stopped early (res or res^2 already > hashmap_size).
    uint32_t stride = 1; bool early = false;
    for (int dim = 0; dim < 3; ++dim) { if (stride <= g.hs) stride *= g.res; else early = true; }
    g.hashed = early || (g.hs < stride);
    g.res2 = g.res * g.res;
    return g;
}

__device__ __forceinline__ void pos_fract(float x, float scale, float& pos, uint32_t& cell) {
    const float p = fmaf(scale, x, 0.5f);
    const float f = floorf(p);
    cell = (uint32_t)(int)f;
    pos = p - f;
}

// entry index of vertex (gx,gy,gz) inside the level
__device__ __forceinline__ uint32_t grid_index(const LevelGeom& g, uint32_t gx, uint32_t gy, uint32_t gz) {
    if (g.hashed) {
        const uint32_t h = gx ^ (gy * 2654435761u) ^ (gz * 805459861u);
        return h & (g.hs - 1u);                 // hashed levels always hold exactly 2^log2T entries
    }
    uint32_t idx = gx + gy * g.res + gz * g.res2;
    if (idx >= g.hs) idx %= g.hs;               // wrap-around of the +1 vertex at x == 1 (rare)
    return idx;
}

// decoders.py:101 clamps positions to [0,1] before the encoder; US_GRID_CLAMP01 folds that clamp into the load
__device__ __forceinline__ float load_x(const float* __restrict__ x, int64_t i, int k, int clamp) {
    const float v = x[i * 3 + k];
    return clamp ? fminf(fmaxf(v, 0.0f), 1.0f) : v;
}

// feature / gradient tensor layouts: row-major [N][L*F] (the torch module's view) or level-major [L][N][F]
// (US_GRID_LEVEL_MAJOR: what the fused MapStep path uses; a level's plane is contiguous, so a workgroup that owns
// one level streams N*F floats instead of touching every 128-byte row of the [N][32] matrix)
__device__ __forceinline__ int64_t feat_index(int lm, int64_t i, int64_t n, uint32_t level, uint32_t C, int F) {
    return lm ? ((int64_t)level * n + i) * F : i * C + (int64_t)level * F;
}

// weights in tcnn's multiplication order: w = ((1*a0)*a1)*a2
__device__ __forceinline__ float corner_weight(int c, const float pos[3]) {
    float w = (c & 1) ? pos[0] : 1.0f - pos[0];
    w *= (c & 2) ? pos[1] : 1.0f - pos[1];
    w *= (c & 4) ? pos[2] : 1.0f - pos[2];
    return w;
}

template <int F> struct Feat;
template <> struct Feat<1> { typedef float  T; };
template <> struct Feat<2> { typedef float2 T; };
template <> struct Feat<4> { typedef float4 T; };

template <int F> __device__ __forceinline__ void feat_to_array(const typename Feat<F>::T& v, float* a);
template <> __device__ __forceinline__ void feat_to_array<1>(const float& v, float* a) { a[0] = v; }
template <> __device__ __forceinline__ void feat_to_array<2>(const float2& v, float* a) { a[0] = v.x; a[1] = v.y; }
template <> __device__ __forceinline__ void feat_to_array<4>(const float4& v, float* a) { a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }


template <int F> __device__ __forceinline__ void array_to_feat(const float* a, typename Feat<F>::T& v);
template <> __device__ __forceinline__ void array_to_feat<1>(const float* a, float& v) { v = a[0]; }
template <> __device__ __forceinline__ void array_to_feat<2>(const float* a, float2& v) { v.x = a[0]; v.y = a[1]; }
template <> __device__ __forceinline__ void array_to_feat<4>(const float* a, float4& v) { v.x = a[0]; v.y = a[1]; v.z = a[2]; v.w = a[3]; }

// the 8 vertices of the cell: v[c], c = x + 2y + 4z
template <int F>
__device__ __forceinline__ void gather_corners(const LevelGeom& g, const typename Feat<F>::T* __restrict__ grid, const uint32_t cell[3],
                                               typename Feat<F>::T (&v)[8]) {
    if constexpr (F == 2) {
        // Divergent gathers cost one address cycle per lane, whatever their width: fetch the x and x+1 vertices of a
        // cell edge with ONE 16-byte load where they are neighbours in memory.  Dense levels: always (entries e, e+1).
        // Hashed levels: the coherent prime hash leaves x unmultiplied, so for even x the two vertices are the aligned
        // pair {k & ~1, k | 1}; odd x needs a second (8-byte) gather, issued for those lanes only.
        typedef float pair_t __attribute__((ext_vector_type(4), aligned(8)));
        const float2* g2 = reinterpret_cast<const float2*>(grid);
        if (g.hashed) {
            const uint32_t mask = g.hs - 1u;
            const uint32_t hy0 = cell[1] * 2654435761u, hz0 = cell[2] * 805459861u;
            const bool odd = cell[0] & 1u;
            pair_t pr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t h = ((q & 1) ? hy0 + 2654435761u : hy0) ^ ((q & 2) ? hz0 + 805459861u : hz0);
                const uint32_t k = (cell[0] ^ h) & mask;
                pr[q] = *reinterpret_cast<const pair_t*>(g2 + (k & ~1u));
            }
            float2 up[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) up[q] = make_float2(0.f, 0.f);
            if (odd) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t h = ((q & 1) ? hy0 + 2654435761u : hy0) ^ ((q & 2) ? hz0 + 805459861u : hz0);
                    up[q] = g2[((cell[0] + 1u) ^ h) & mask];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t h = ((q & 1) ? hy0 + 2654435761u : hy0) ^ ((q & 2) ? hz0 + 805459861u : hz0);
                const bool k_odd = ((cell[0] ^ h) & 1u) != 0u;                  // which half of the pair is vertex x
                const float2 lo = make_float2(pr[q].x, pr[q].y), hi = make_float2(pr[q].z, pr[q].w);
                v[2 * q] = k_odd ? hi : lo;
                v[2 * q + 1] = odd ? up[q] : (k_odd ? lo : hi);
            }
        } else {
            const uint32_t base = cell[0] + cell[1] * g.res + cell[2] * g.res2;
            uint32_t e[4]; bool wrap = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                e[q] = base + ((q & 1) ? g.res : 0u) + ((q & 2) ? g.res2 : 0u);
                wrap |= (e[q] >= g.hs - 1u);                   // also catches e = 0xFFFFFFFF (cells of coordinates < 0)
            }
            if (!wrap) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const pair_t pr = *reinterpret_cast<const pair_t*>(g2 + e[q]);
                    v[2 * q] = make_float2(pr.x, pr.y); v[2 * q + 1] = make_float2(pr.z, pr.w);
                }
            } else {                                                        // the +1 vertex wraps around the slab (x == 1)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[2 * q] = g2[e[q] % g.hs]; v[2 * q + 1] = g2[(e[q] + 1u) % g.hs];
                }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c)           // 8 independent gathers in flight
            v[c] = grid[grid_index(g, cell[0] + (c & 1), cell[1] + ((c >> 1) & 1), cell[2] + ((c >> 2) & 1))];
    }
}

