// hashgrid.hip -- multi-resolution hash-grid encoding for gfx950 (MI355X).
//
// Replaces tcnn.Encoding("HashGrid") as used by Uni-SLAM (reference src/UNISLAM.py:242-253 construction,
// src/networks/decoders.py:101-103 call, autograd backward at src/Mapper.py:444 / src/Tracker.py:241).
// The arithmetic follows the published tiny-cuda-nn algorithm (grid.h of the pinned commit 2ec562e8):
// grid_scale / grid_resolution / pos_fract / grid_index with the coherent prime hash; indices are bit-exact
// with oracle/hashgrid_ref.c, features use the same fmaf chain (corner order c = 0..7, bit d of c = +1 in dim d).
//
// Layout in HBM: `params` is ONE flat fp32 vector, level after level, entry-major ([entry][F]); a level's slab
// starts at offset[l]*F floats.  Outputs are [N][L*F] row-major (what the torch module returns).
//
// Kernels
//   k_fwd        one thread per (point, level); blockIdx.y = level, so the level's scale/resolution/offset are
//                wave-uniform kernel arguments living in SGPRs (no LDS staging needed for them) and all blocks in
//                flight work on the same level slab (<= 4 MiB for log2T = 19), which is what one XCD L2 holds.
//                8 independent 8-byte gathers per thread are issued before the first is consumed.
//   k_bwd_atomic tcnn-shaped scatter: 8*F global float atomics per (point, level).
//   k_bwd_sliced the fast path: a 1024-thread workgroup owns a SLICE (<= 32768 floats = 128 KiB of LDS) of one
//                level's table and a partition of the points; it re-derives the 8 corner indices of every point in
//                its partition (VALU is cheap, memory-side float atomics are not: ~20 G requests/s chip-wide),
//                accumulates hits in LDS with ds_add_f32 and finally flushes the slice with contiguous global
//                atomics (256 B per wave instruction = the full atomic rate).  No point sorting, no host sync.
//   k_bwd_input  dL/dx from the dy_dx the forward stored (tcnn kernel_grid_backward_input).
#include "us_common.h"
#include <math.h>
#include <string.h>

#include "hashgrid_dev.h"

// ---------------------------------------------------------------------------------------------------------------
// host: descriptor (tcnn GridEncodingTemplated constructor arithmetic, fp32 like the original)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int us_grid_desc_init(us_grid_desc* d, uint32_t n_levels, uint32_t n_features, uint32_t log2_hashmap_size,
                                 uint32_t base_resolution, float per_level_scale) {
    US_REQUIRE(d, US_ERR_NULL, "us_grid_desc_init: desc is NULL");
    US_REQUIRE(n_levels >= 1 && n_levels <= US_MAX_LEVELS, US_ERR_CONFIG, "us_grid_desc_init: n_levels %u not in 1..%d", n_levels, US_MAX_LEVELS);
    US_REQUIRE(n_features == 1 || n_features == 2 || n_features == 4, US_ERR_CONFIG, "us_grid_desc_init: n_features_per_level %u not in {1,2,4}", n_features);
    US_REQUIRE(log2_hashmap_size >= 3 && log2_hashmap_size <= 28, US_ERR_CONFIG, "us_grid_desc_init: log2_hashmap_size %u out of range", log2_hashmap_size);
    US_REQUIRE(base_resolution >= 1 && per_level_scale >= 1.0f, US_ERR_CONFIG, "us_grid_desc_init: bad base_resolution / per_level_scale");
    memset(d, 0, sizeof(*d));
    d->n_levels = n_levels; d->n_features = n_features; d->log2_hashmap_size = log2_hashmap_size;
    d->base_resolution = base_resolution; d->per_level_scale = per_level_scale;
    const float log2_pls = log2f(per_level_scale);
    uint64_t offset = 0;
    for (uint32_t l = 0; l < n_levels; ++l) {
        const float scale = exp2f((float)l * log2_pls) * (float)base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        const uint32_t max_params = 0xFFFFFFFFu / 2u;
        uint32_t n = (powf((float)res, 3.0f) > (float)max_params) ? max_params : res * res * res;
        n = ((n + 7u) / 8u) * 8u;
        const uint32_t cap = 1u << log2_hashmap_size;
        if (n > cap) n = cap;
        d->scale[l] = scale; d->resolution[l] = res; d->offset[l] = (uint32_t)offset;
        offset += n;
    }
    US_REQUIRE(offset * n_features < 0xFFFFFFFFull, US_ERR_CONFIG, "us_grid_desc_init: table too large");
    d->offset[n_levels] = (uint32_t)offset;
    d->n_params = (uint32_t)(offset * n_features);
    return US_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int F, bool DYDX>
__global__ __launch_bounds__(256) void k_fwd(LevelTable tab, uint32_t n_levels, const float* __restrict__ params,
                                             const float* __restrict__ x, int64_t n, float* __restrict__ out,
                                             float* __restrict__ dy_dx, int clamp, int lm,
                                             const int32_t* __restrict__ n_dev, int n_mul) {
    const uint32_t level = blockIdx.y;
    const LevelGeom g = level_geom(tab, level);
    const typename Feat<F>::T* grid = reinterpret_cast<const typename Feat<F>::T*>(params) + tab.off[level];
    const uint32_t C = n_levels * F;
    // n_dev: the number of points actually present is n_dev[0] * n_mul (device memory, <= n): the launch is sized for n, which stays the
    // plane stride of the level-major layout; threads beyond the count leave at once (us_zero_depth_resample: no row count on the host)
    int64_t n_act = n;
    if (n_dev) { const int64_t m = (int64_t)n_dev[0] * n_mul; n_act = m < n ? m : n; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_act; i += (int64_t)gridDim.x * blockDim.x) {
        float pos[3]; uint32_t cell[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pos_fract(load_x(x, i, k, clamp), g.scale, pos[k], cell[k]);
        typename Feat<F>::T v[8];
        gather_corners<F>(g, grid, cell, v);
        float res[F];
#pragma unroll
        for (int f = 0; f < F; ++f) res[f] = 0.0f;
        float va[8][F];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            feat_to_array<F>(v[c], va[c]);
            const float w = corner_weight(c, pos);
#pragma unroll
            for (int f = 0; f < F; ++f) res[f] = fmaf(w, va[c][f], res[f]);
        }
        float* o = out + feat_index(lm, i, n, level, C, F);
#pragma unroll
        for (int f = 0; f < F; ++f) o[f] = res[f];
        if (DYDX) {
            // tcnn: for grad_dim, idx over the two other dims (lower dim = bit 0 of idx), weight = scale*a*b,
            // grads += weight * (right - left)
            float* d = dy_dx + (i * C + level * F) * 3;
#pragma unroll
            for (int gd = 0; gd < 3; ++gd) {
                float acc[F];
#pragma unroll
                for (int f = 0; f < F; ++f) acc[f] = 0.0f;
                const int d0 = gd == 0 ? 1 : 0, d1 = gd == 2 ? 1 : 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float w = g.scale;
                    w *= (q & 1) ? pos[d0] : 1.0f - pos[d0];
                    w *= (q & 2) ? pos[d1] : 1.0f - pos[d1];
                    const int cl = ((q & 1) << d0) | (((q >> 1) & 1) << d1);
                    const int cr = cl | (1 << gd);
#pragma unroll
                    for (int f = 0; f < F; ++f) acc[f] += w * (va[cr][f] - va[cl][f]);
                }
                const float xin = x[i * 3 + gd];
                const bool pass = !clamp || (xin >= 0.0f && xin <= 1.0f);
#pragma unroll
                for (int f = 0; f < F; ++f) d[f * 3 + gd] = pass ? acc[f] : 0.0f;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_indices(LevelTable tab, uint32_t n_levels, const float* __restrict__ x,
                                                 int64_t n, uint32_t* __restrict__ idx, int clamp) {
    const uint32_t level = blockIdx.y;
    const LevelGeom g = level_geom(tab, level);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pos[3]; uint32_t cell[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pos_fract(load_x(x, i, k, clamp), g.scale, pos[k], cell[k]);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            idx[(i * n_levels + level) * 8 + c] =
                grid_index(g, cell[0] + (c & 1), cell[1] + ((c >> 1) & 1), cell[2] + ((c >> 2) & 1));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward wrt table: direct global atomics (levels selected by `level_mask`)
// ---------------------------------------------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(256) void k_bwd_atomic(LevelTable tab, uint32_t n_levels, uint32_t level_mask,
                                                    const float* __restrict__ x, const float* __restrict__ dL_dy,
                                                    int64_t n, float* __restrict__ grad, int clamp, int lm) {
    const uint32_t level = blockIdx.y;
    if (!((level_mask >> level) & 1u)) return;
    const LevelGeom g = level_geom(tab, level);
    float* gl = grad + (size_t)tab.off[level] * F;
    const uint32_t C = n_levels * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float dy[F]; bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { dy[f] = dL_dy[feat_index(lm, i, n, level, C, F) + f]; any |= (dy[f] != 0.0f); }
        if (!any) continue;                    // adding zeros changes nothing
        float pos[3]; uint32_t cell[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pos_fract(load_x(x, i, k, clamp), g.scale, pos[k], cell[k]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t e = grid_index(g, cell[0] + (c & 1), cell[1] + ((c >> 1) & 1), cell[2] + ((c >> 2) & 1));
            const float w = corner_weight(c, pos);
#pragma unroll
            for (int f = 0; f < F; ++f) atomicAdd(gl + (size_t)e * F + f, w * dy[f]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward wrt table: LDS-privatised slices
// ---------------------------------------------------------------------------------------------------------------
#define US_SLICE_THREADS 1024
#define US_SLICE_WAVES (US_SLICE_THREADS / 64)
#define US_QCAP 128                      // per-wave compaction queue entries (>= 2 x 64)

// LDS budget (160 KiB): accumulator slice + 16 per-wave queues of (index, F values)
template <int F> struct SliceCfg {
    static constexpr int QUEUE_FLOATS = US_SLICE_WAVES * US_QCAP * (1 + F);
    static constexpr int SLICE_FLOATS = (F == 4) ? 24576 : 32768;          // 96 / 128 KiB
    static_assert((SLICE_FLOATS + QUEUE_FLOATS) * 4 <= 163840, "LDS budget");
};
static uint32_t slice_entries_for(uint32_t F) { return (F == 4 ? 24576u : 32768u) / F; }

struct SliceMap {                        // prefix sum of slices per level (only levels in the mask have slices)
    uint32_t first[US_MAX_LEVELS + 1];
};

// A ds_add_f32 wave-instruction occupies the LDS for ~64 cycles however few lanes are active (measured: 16 sparse
// atomics per wave-iteration made the first version of this kernel 10x slower than its VALU work).  So hits are first
// COMPACTED: each wave appends its in-slice (index, values) pairs to a private LDS queue (ballot + mbcnt prefix, plain
// ds_write), and only full groups of 64 queue entries are turned into atomics -> 1/16 .. 1/32 of the instructions.
template <int F, bool COMPACT>
__global__ __launch_bounds__(US_SLICE_THREADS) void k_bwd_sliced(LevelTable tab, SliceMap smap, uint32_t n_levels,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ dL_dy, int64_t n,
                                                                 float* __restrict__ grad, int clamp, int lm, int exclusive) {
    typedef SliceCfg<F> SC;
    __shared__ __attribute__((aligned(16))) float acc[SC::SLICE_FLOATS];
    __shared__ uint32_t q_idx[US_SLICE_WAVES][US_QCAP];
    __shared__ float q_val[US_SLICE_WAVES][F][US_QCAP];
    constexpr uint32_t SLICE_ENTRIES = SC::SLICE_FLOATS / F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // which (level, slice) is blockIdx.y ?   (wave-uniform scalar search over <= 32 levels)
    uint32_t level = 0;
    while (level + 1 < n_levels && smap.first[level + 1] <= blockIdx.y) ++level;
    const uint32_t slice = blockIdx.y - smap.first[level];
    const LevelGeom g = level_geom(tab, level);
    const uint32_t lo = slice * SLICE_ENTRIES;
    const uint32_t cnt = min(SLICE_ENTRIES, g.hs - lo);
    {
        float4* a4 = reinterpret_cast<float4*>(acc);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t k = threadIdx.x; k < (cnt * F + 3) / 4; k += US_SLICE_THREADS) a4[k] = z4;
    }
    __syncthreads();

    const uint32_t C = n_levels * F;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;      // contiguous partition of the points for this blockIdx.x
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = min(n, i0 + per);
    uint32_t qn = 0;                                          // wave-uniform queue fill
    uint32_t* qi = q_idx[wave];
    // Every workgroup of a level streams the SAME x / dy addresses.  Started together they walk them in lockstep and all
    // 32 CUs of an XCD hammer one L2 channel at a time (measured: 10-30k cycles per step).  Rotating the start of each
    // workgroup's sweep by a slice-dependent amount spreads the requests over all channels.
    const int64_t n_steps = (i1 - i0 + US_SLICE_THREADS - 1) / US_SLICE_THREADS;
    const int64_t rot = n_steps > 0 ? (int64_t)((blockIdx.y * 2654435761u) >> 8) % n_steps : 0;
    for (int64_t step = 0; step < n_steps; ++step) {          // uniform trip count: ballots need whole waves
        int64_t sstep = step + rot; if (sstep >= n_steps) sstep -= n_steps;
        const int64_t i = i0 + sstep * US_SLICE_THREADS + threadIdx.x;
        bool live = i < i1;
        float dy[F];
#pragma unroll
        for (int f = 0; f < F; ++f) dy[f] = 0.0f;
        float pos[3] = {0.f, 0.f, 0.f}; uint32_t cell[3] = {0u, 0u, 0u};
        if (live) {
            bool any = false;
#pragma unroll
            for (int f = 0; f < F; ++f) { dy[f] = dL_dy[feat_index(lm, i, n, level, C, F) + f]; any |= (dy[f] != 0.0f); }
#pragma unroll
            for (int k = 0; k < 3; ++k) pos_fract(load_x(x, i, k, clamp), g.scale, pos[k], cell[k]);
            live = any;                                       // adding zeros changes nothing
        }
        if (!COMPACT && !live) continue;                      // whole waves of zero-gradient samples skip the hashing
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t e = grid_index(g, cell[0] + (c & 1), cell[1] + ((c >> 1) & 1), cell[2] + ((c >> 2) & 1)) - lo;
            const bool hit = live && (e < cnt);               // unsigned compare also rejects e < lo
            if (!COMPACT) {
                if (hit) {
                    const float w = corner_weight(c, pos);
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        atomicAdd(&acc[e * F + f], w * dy[f]);      // ds_add_f32: ~3 cycles per active lane
                    }
                }
                continue;
            }
            const unsigned long long mask = __ballot(hit);
            if (mask == 0ull) continue;                       // wave-uniform
            if (hit) {
                const uint32_t p = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                const float w = corner_weight(c, pos);
                qi[p] = e * F;
#pragma unroll
                for (int f = 0; f < F; ++f) q_val[wave][f][p] = w * dy[f];
            }
            qn += (uint32_t)__popcll(mask);
            if (qn >= 64u) {                                  // drain the LAST 64 entries with all 64 lanes active
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const uint32_t k = qn - 64u + lane;
                const uint32_t a = qi[k];
#pragma unroll
                for (int f = 0; f < F; ++f) atomicAdd(&acc[a + f], q_val[wave][f][k]);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                qn -= 64u;
            }
        }
    }
    if (qn) {                                                 // remainder (< 64 entries), once per wave
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if ((uint32_t)lane < qn) {
            const uint32_t a = qi[lane];
#pragma unroll
            for (int f = 0; f < F; ++f) atomicAdd(&acc[a + f], q_val[wave][f][lane]);
        }
    }
    __syncthreads();
    float* gl = grad + ((size_t)tab.off[level] + lo) * F;
    if (exclusive) {
        // this workgroup is the only one that touches this slice in this launch: plain read-modify-write
        for (uint32_t k = threadIdx.x; k < cnt * F; k += US_SLICE_THREADS) { const float v = acc[k]; if (v != 0.0f) gl[k] += v; }
    } else {
        for (uint32_t k = threadIdx.x; k < cnt * F; k += US_SLICE_THREADS) {
            const float v = acc[k];
            if (v != 0.0f) atomicAdd(gl + k, v);              // contiguous lanes -> 64-B atomic requests, full atomic rate
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward wrt positions
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bwd_input(const float* __restrict__ dL_dy, const float* __restrict__ dy_dx,
                                                   int64_t n, uint32_t C, float* __restrict__ dL_dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        for (uint32_t k = 0; k < C; ++k) {
            const float gk = dL_dy[i * C + k];
            const float* d = dy_dx + (i * C + k) * 3;
            r0 += gk * d[0]; r1 += gk * d[1]; r2 += gk * d[2];
        }
        dL_dx[i * 3 + 0] = r0; dL_dx[i * 3 + 1] = r1; dL_dx[i * 3 + 2] = r2;
    }
}

// The same without a stored dy_dx: every (point, level) gathers its 8 vertices again and forms dy/dx on the fly (what
// k_fwd<DYDX> writes, value for value).  Saves the [N][C][3] tensor (100 MB per grid at 4096 x 64) at the price of a second
// gather pass.
template <int F>
__device__ __forceinline__ void input_grad_level(const LevelTable& tab, uint32_t level, uint32_t n_levels, const float* __restrict__ params,
                                                 const float xv[3], const bool pass[3], const float* __restrict__ dL_dy, int64_t i, int64_t n,
                                                 int lm, float r[3]) {
    const uint32_t C = n_levels * F;
    const LevelGeom g = level_geom(tab, level);
    const typename Feat<F>::T* grid = reinterpret_cast<const typename Feat<F>::T*>(params) + tab.off[level];
    float pos[3]; uint32_t cell[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pos_fract(xv[k], g.scale, pos[k], cell[k]);
    typename Feat<F>::T v[8];
    gather_corners<F>(g, grid, cell, v);
    float va[8][F], dy[F];
#pragma unroll
    for (int c = 0; c < 8; ++c) feat_to_array<F>(v[c], va[c]);
#pragma unroll
    for (int f = 0; f < F; ++f) dy[f] = dL_dy[feat_index(lm, i, n, level, C, F) + f];
    float d[F][3];
#pragma unroll
    for (int gd = 0; gd < 3; ++gd) {
        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.0f;
        const int d0 = gd == 0 ? 1 : 0, d1 = gd == 2 ? 1 : 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float w = g.scale;
            w *= (q & 1) ? pos[d0] : 1.0f - pos[d0];
            w *= (q & 2) ? pos[d1] : 1.0f - pos[d1];
            const int cl = ((q & 1) << d0) | (((q >> 1) & 1) << d1);
            const int cr = cl | (1 << gd);
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] += w * (va[cr][f] - va[cl][f]);
        }
#pragma unroll
        for (int f = 0; f < F; ++f) d[f][gd] = pass[gd] ? acc[f] : 0.0f;
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) r[gd] += dy[f] * d[f][gd];
    }
}

// A wavefront = 16 points x 4 levels (lane = point + 16 * (level mod 4)): every lane walks the levels {r, r+4, r+8, ...} of its
// point (r = its row of 16 lanes), keeping its share of dL/dx in registers; two cross-row shuffles add the four shares.
// No LDS, no barrier: a wave stalled on the cache misses of a fine level holds up nobody else.  The 16 lanes of a row read 16
// consecutive points of one level plane (128 contiguous bytes of dL_dy in the level-major layout).
#define IG_POINTS 16
template <int F>
__global__ __launch_bounds__(256) void k_bwd_input_gather(LevelTable tab, uint32_t n_levels, const float* __restrict__ params,
                                                          const float* __restrict__ x, const float* __restrict__ dL_dy, int64_t n,
                                                          float* __restrict__ dL_dx, int clamp, int lm, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, pt = lane & 15;
    const int64_t i = ((int64_t)blockIdx.x * 4 + wave) * IG_POINTS + pt;
    const bool in = i < n;
    float xv[3] = {0.f, 0.f, 0.f}, r[3] = {0.f, 0.f, 0.f};
    bool pass[3] = {true, true, true};
    if (in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float xin = x[i * 3 + k];
            xv[k] = clamp ? fminf(fmaxf(xin, 0.0f), 1.0f) : xin;
            pass[k] = !clamp || (xin >= 0.0f && xin <= 1.0f);
        }
        for (uint32_t level = (uint32_t)row; level < n_levels; level += 4)
            input_grad_level<F>(tab, level, n_levels, params, xv, pass, dL_dy, i, n, lm, r);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {                                  // whole wave active: add the four rows' shares
        r[k] += __shfl_xor(r[k], 16, 64);
        r[k] += __shfl_xor(r[k], 32, 64);
    }
    if (row == 0 && in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) dL_dx[i * 3 + k] = accumulate ? dL_dx[i * 3 + k] + r[k] : r[k];
    }
}

// Both grids' input gradient AND its reduction to the ray in one launch: dL/d(rays_o), dL/d(rays_d) are what the camera poses receive
// (src/Tracker.py:170-174,241; joint pose optimisation src/Mapper.py:372-376,444), the per-point dL/dx is only a stop on the way.
// A workgroup = one ray; wave w = its samples 16w .. 16w+15 x 4 level rows as in k_bwd_input_gather, first through table A's levels,
// then through table B's (dL/dx = share_A + share_B, each share summed over its rows as the one-grid kernel does: the optional
// per-point output is bit-identical to two us_hashgrid_bwd_input_gather launches).  Then the adjoint of us_ray_points on the values
// the lanes hold: g = dL/dx / span, dL/do = sum_s g, dL/dd = sum_s g z_s -- wave sums, then the waves in order through LDS.
#define IR_MAX_WAVES 8
struct RaySpan { float span[3]; };
__global__ __launch_bounds__(64 * IR_MAX_WAVES) void k_bwd_input_rays(LevelTable tabA, LevelTable tabB, uint32_t n_levels,
                                                                      const float* __restrict__ pA, const float* __restrict__ pB,
                                                                      const float* __restrict__ x, const float* __restrict__ dyA,
                                                                      const float* __restrict__ dyB, int64_t n, int S,
                                                                      const float* __restrict__ z_vals, RaySpan bd, float* __restrict__ g_o,
                                                                      float* __restrict__ g_d, float* __restrict__ dL_dx, int clamp, int lm) {
    __shared__ float sh[IR_MAX_WAVES][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, pt = lane & 15;
    const int64_t ray = blockIdx.x;
    const int s = wave * IG_POINTS + pt;
    const bool in = s < S;
    const int64_t i = ray * S + s;
    float xv[3] = {0.f, 0.f, 0.f}, rA[3] = {0.f, 0.f, 0.f}, rB[3] = {0.f, 0.f, 0.f};
    bool pass[3] = {true, true, true};
    if (in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float xin = x[i * 3 + k];
            xv[k] = clamp ? fminf(fmaxf(xin, 0.0f), 1.0f) : xin;
            pass[k] = !clamp || (xin >= 0.0f && xin <= 1.0f);
        }
        for (uint32_t level = (uint32_t)row; level < n_levels; level += 4)
            input_grad_level<2>(tabA, level, n_levels, pA, xv, pass, dyA, i, n, lm, rA);
        for (uint32_t level = (uint32_t)row; level < n_levels; level += 4)
            input_grad_level<2>(tabB, level, n_levels, pB, xv, pass, dyB, i, n, lm, rB);
    }
    float so[3], sd[3];
    const float z = (in && row == 0) ? z_vals[i] : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rA[k] += __shfl_xor(rA[k], 16, 64); rA[k] += __shfl_xor(rA[k], 32, 64);
        rB[k] += __shfl_xor(rB[k], 16, 64); rB[k] += __shfl_xor(rB[k], 32, 64);
        const float r = rA[k] + rB[k];
        if (dL_dx && row == 0 && in) dL_dx[i * 3 + k] = r;
        const float gk = (row == 0 && in) ? r / bd.span[k] : 0.0f;
        so[k] = wave_sum(gk); sd[k] = wave_sum(gk * z);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { sh[wave][k] = so[k]; sh[wave][3 + k] = sd[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float a = 0.0f;
        const int nw = blockDim.x >> 6;
        for (int w = 0; w < nw; ++w) a += sh[w][threadIdx.x];
        (threadIdx.x < 3 ? g_o : g_d)[ray * 3 + (threadIdx.x % 3)] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
static int check_desc(const char* fn, const us_grid_desc* d) {
    US_REQUIRE(d, US_ERR_NULL, "%s: desc is NULL", fn);
    US_REQUIRE(d->n_levels >= 1 && d->n_levels <= US_MAX_LEVELS, US_ERR_CONFIG, "%s: n_levels %u", fn, d->n_levels);
    US_REQUIRE(d->n_features == 1 || d->n_features == 2 || d->n_features == 4, US_ERR_CONFIG, "%s: n_features %u", fn, d->n_features);
    US_REQUIRE(d->n_params == d->offset[d->n_levels] * d->n_features, US_ERR_CONFIG, "%s: descriptor not initialised by us_grid_desc_init", fn);
    return US_OK;
}

static unsigned point_blocks(int64_t n, int threads, int cap) {
    int64_t b = us_cdiv(n, threads);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// us_hashgrid_fwd with the point count optionally read on the device (n_dev[0] * n_mul <= n; NULL: n): shared with render.hip
int us_hashgrid_fwd_counted_rows(const us_grid_desc* d, const float* params, const float* x, int64_t n, float* out,
                                 float* dy_dx, int flags, const int32_t* n_dev, int n_mul, void* stream) {
    const int clamp = flags & US_GRID_CLAMP01, lm = (flags & US_GRID_LEVEL_MAJOR) ? 1 : 0; (void)lm;
    int rc = check_desc("us_hashgrid_fwd", d); if (rc) return rc;
    US_REQUIRE(n >= 0, US_ERR_SHAPE, "us_hashgrid_fwd: n < 0");
    if (n == 0) return US_OK;                       // empty batches carry NULL data pointers
    US_REQUIRE(params && x && out, US_ERR_NULL, "us_hashgrid_fwd: NULL pointer");
    US_REQUIRE(((uintptr_t)params & 15u) == 0, US_ERR_SHAPE, "us_hashgrid_fwd: params must be 16-byte aligned");
    const LevelTable t = make_table(d);
    dim3 grid(point_blocks(n, 256, 1 << 20), d->n_levels), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_FWD(F)                                                                                         \
    if (dy_dx) hipLaunchKernelGGL((k_fwd<F, true>), grid, block, 0, s, t, d->n_levels, params, x, n, out, dy_dx, clamp, lm, n_dev, n_mul); \
    else hipLaunchKernelGGL((k_fwd<F, false>), grid, block, 0, s, t, d->n_levels, params, x, n, out, dy_dx, clamp, lm, n_dev, n_mul);
    switch (d->n_features) { case 1: LAUNCH_FWD(1) break; case 2: LAUNCH_FWD(2) break; default: LAUNCH_FWD(4) break; }
#undef LAUNCH_FWD
    US_CHECK_LAUNCH("us_hashgrid_fwd");
    return US_OK;
}

extern "C" int us_hashgrid_fwd(const us_grid_desc* d, const float* params, const float* x, int64_t n, float* out,
                               float* dy_dx, int flags, void* stream) {
    return us_hashgrid_fwd_counted_rows(d, params, x, n, out, dy_dx, flags, nullptr, 0, stream);
}

extern "C" int us_hashgrid_indices(const us_grid_desc* d, const float* x, int64_t n, uint32_t* idx, int flags, void* stream) {
    const int clamp = flags & US_GRID_CLAMP01, lm = (flags & US_GRID_LEVEL_MAJOR) ? 1 : 0; (void)lm;
    int rc = check_desc("us_hashgrid_indices", d); if (rc) return rc;
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(x && idx, US_ERR_NULL, "us_hashgrid_indices: NULL pointer");
    const LevelTable t = make_table(d);
    dim3 grid(point_blocks(n, 256, 1 << 20), d->n_levels), block(256);
    hipLaunchKernelGGL(k_indices, grid, block, 0, (hipStream_t)stream, t, d->n_levels, x, n, idx, clamp);
    US_CHECK_LAUNCH("us_hashgrid_indices");
    return US_OK;
}

extern "C" int us_hashgrid_bwd_params(const us_grid_desc* d, const float* x, const float* dL_dy, int64_t n,
                                      float* grad_params, int mode, int flags, void* stream) {
    const int clamp = flags & US_GRID_CLAMP01, lm = (flags & US_GRID_LEVEL_MAJOR) ? 1 : 0; (void)lm;
    int rc = check_desc("us_hashgrid_bwd_params", d); if (rc) return rc;
    US_REQUIRE(mode >= -1 && mode <= 2, US_ERR_CONFIG, "us_hashgrid_bwd_params: mode %d", mode);
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(x && dL_dy && grad_params, US_ERR_NULL, "us_hashgrid_bwd_params: NULL pointer");
    const LevelTable t = make_table(d);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t F = d->n_features, L = d->n_levels;
    const uint32_t slice_entries = slice_entries_for(F);
    // per level: slices pay when the level receives many more updates (8 per point) than it has entries
    uint32_t sliced_mask = 0;
    for (uint32_t l = 0; l < L; ++l) {
        const uint64_t hs = d->offset[l + 1] - d->offset[l];
        bool sliced = mode >= 1 || (mode == -1 && (uint64_t)n * 8ull >= hs);
        if (sliced) sliced_mask |= 1u << l;
    }
    const uint32_t all = L == 32 ? 0xFFFFFFFFu : ((1u << L) - 1u);
    const uint32_t atomic_mask = all & ~sliced_mask;
    if (atomic_mask) {
        dim3 grid(point_blocks(n, 256, 1 << 20), L), block(256);
        switch (F) {
            case 1: hipLaunchKernelGGL((k_bwd_atomic<1>), grid, block, 0, s, t, L, atomic_mask, x, dL_dy, n, grad_params, clamp, lm); break;
            case 2: hipLaunchKernelGGL((k_bwd_atomic<2>), grid, block, 0, s, t, L, atomic_mask, x, dL_dy, n, grad_params, clamp, lm); break;
            default: hipLaunchKernelGGL((k_bwd_atomic<4>), grid, block, 0, s, t, L, atomic_mask, x, dL_dy, n, grad_params, clamp, lm); break;
        }
        US_CHECK_LAUNCH("us_hashgrid_bwd_params(atomic)");
    }
    if (sliced_mask) {
        SliceMap sm; uint32_t total = 0;
        for (uint32_t l = 0; l < L; ++l) {
            sm.first[l] = total;
            if ((sliced_mask >> l) & 1u) total += (uint32_t)us_cdiv(d->offset[l + 1] - d->offset[l], slice_entries);
        }
        for (uint32_t l = L; l <= US_MAX_LEVELS; ++l) sm.first[l] = total;
        // levels outside the mask get zero slices: first[l+1] == first[l], the in-kernel search skips them.
        // point partitions: ~4 workgroups per CU (one is resident at a time: the slice fills the LDS) keeps the tail of the
        // launch short; never fewer than 8192 points per workgroup so the <= 128 KiB slice flush stays a small part
        int64_t parts = us_cdiv(1024, total);
        const int64_t max_parts = us_cdiv(n, 8192);
        if (parts > max_parts) parts = max_parts;
        if (parts < 1) parts = 1;
        const int exclusive = parts == 1 ? 1 : 0;
        dim3 grid((unsigned)parts, total), block(US_SLICE_THREADS);
        switch (F) {
#define LAUNCH_SLICED(F)                                                                                                      \
    if (mode == 2) hipLaunchKernelGGL((k_bwd_sliced<F, true>), grid, block, 0, s, t, sm, L, x, dL_dy, n, grad_params, clamp, lm, exclusive); \
    else hipLaunchKernelGGL((k_bwd_sliced<F, false>), grid, block, 0, s, t, sm, L, x, dL_dy, n, grad_params, clamp, lm, exclusive);
            case 1: LAUNCH_SLICED(1) break;
            case 2: LAUNCH_SLICED(2) break;
            default: LAUNCH_SLICED(4) break;
#undef LAUNCH_SLICED
        }
        US_CHECK_LAUNCH("us_hashgrid_bwd_params(sliced)");
    }
    return US_OK;
}

extern "C" int us_hashgrid_bwd_input(const float* dL_dy, const float* dy_dx, int64_t n, uint32_t C, float* dL_dx,
                                     void* stream) {
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(dL_dy && dy_dx && dL_dx, US_ERR_NULL, "us_hashgrid_bwd_input: NULL pointer");
    hipLaunchKernelGGL(k_bwd_input, dim3(point_blocks(n, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, dL_dy, dy_dx, n, C, dL_dx);
    US_CHECK_LAUNCH("us_hashgrid_bwd_input");
    return US_OK;
}

extern "C" int us_hashgrid_bwd_input_gather(const us_grid_desc* d, const float* params, const float* x, const float* dL_dy, int64_t n,
                                            float* dL_dx, int flags, void* stream) {
    int rc = check_desc("us_hashgrid_bwd_input_gather", d); if (rc) return rc;
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(params && x && dL_dy && dL_dx, US_ERR_NULL, "us_hashgrid_bwd_input_gather: NULL pointer");
    US_REQUIRE(((uintptr_t)params & 15u) == 0, US_ERR_SHAPE, "us_hashgrid_bwd_input_gather: params must be 16-byte aligned");
    const LevelTable t = make_table(d);
    const int clamp = flags & US_GRID_CLAMP01, lm = (flags & US_GRID_LEVEL_MAJOR) ? 1 : 0, acc = (flags & US_GRID_ACCUMULATE) ? 1 : 0;
    dim3 grid((unsigned)us_cdiv(n, 4 * IG_POINTS)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_BIG(F) hipLaunchKernelGGL((k_bwd_input_gather<F>), grid, block, 0, s, t, d->n_levels, params, x, dL_dy, n, dL_dx, clamp, lm, acc);
    switch (d->n_features) { case 1: LAUNCH_BIG(1) break; case 2: LAUNCH_BIG(2) break; default: LAUNCH_BIG(4) break; }
#undef LAUNCH_BIG
    US_CHECK_LAUNCH("us_hashgrid_bwd_input_gather");
    return US_OK;
}

extern "C" int us_hashgrid_bwd_input_rays_supported(const us_grid_desc* a, const us_grid_desc* b, int n_samples) {
    if (!a || !b) return 0;
    if (a->n_features != 2 || b->n_features != 2 || a->n_levels != b->n_levels) return 0;
    if (a->n_levels < 1 || a->n_levels > US_MAX_LEVELS) return 0;
    return (n_samples >= 1 && n_samples <= IR_MAX_WAVES * IG_POINTS) ? 1 : 0;
}

extern "C" int us_hashgrid_bwd_input_rays(const us_grid_desc* a, const us_grid_desc* b, const float* paramsA, const float* paramsB,
                                          const float* x, const float* dL_dyA, const float* dL_dyB, int64_t n_rays, int n_samples,
                                          const float* z_vals, const float* bound_host, float* dL_do, float* dL_dd, float* dL_dx, int flags,
                                          void* stream) {
    int rc = check_desc("us_hashgrid_bwd_input_rays", a); if (rc) return rc;
    rc = check_desc("us_hashgrid_bwd_input_rays", b); if (rc) return rc;
    US_REQUIRE(us_hashgrid_bwd_input_rays_supported(a, b, n_samples), US_ERR_CONFIG,
               "us_hashgrid_bwd_input_rays: needs two F = 2 grids of equal depth and 1..%d samples per ray (got F %u / %u, L %u / %u, S %d)",
               IR_MAX_WAVES * IG_POINTS, a->n_features, b->n_features, a->n_levels, b->n_levels, n_samples);
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(paramsA && paramsB && x && dL_dyA && dL_dyB && z_vals && bound_host && dL_do && dL_dd, US_ERR_NULL, "us_hashgrid_bwd_input_rays: NULL pointer");
    US_REQUIRE((((uintptr_t)paramsA | (uintptr_t)paramsB) & 15u) == 0, US_ERR_SHAPE, "us_hashgrid_bwd_input_rays: params must be 16-byte aligned");
    const int clamp = flags & US_GRID_CLAMP01, lm = (flags & US_GRID_LEVEL_MAJOR) ? 1 : 0;
    RaySpan bd;
    for (int k = 0; k < 3; ++k) bd.span[k] = bound_host[3 + k] - bound_host[k];
    const int waves = (n_samples + IG_POINTS - 1) / IG_POINTS;
    hipLaunchKernelGGL(k_bwd_input_rays, dim3((unsigned)n_rays), dim3(64 * waves), 0, (hipStream_t)stream, make_table(a), make_table(b),
                       a->n_levels, paramsA, paramsB, x, dL_dyA, dL_dyB, n_rays * n_samples, n_samples, z_vals, bd, dL_do, dL_dd, dL_dx, clamp, lm);
    US_CHECK_LAUNCH("us_hashgrid_bwd_input_rays");
    return US_OK;
}
