// render.hip -- ray sampling, point generation, SDF->alpha compositing, masked losses and Adam for gfx950.
//
// Replaces the ATen op chains of reference src/utils/Renderer.py:81-101,132-158, src/common.py:152-166,
// src/Mapper.py:141-175,411-445 and src/Tracker.py:113-147,206-242.  Each per-ray quantity is produced by ONE
// 64-lane wavefront (lane = sample, DPP-lowered shuffles for the transmittance product scan and the five
// reductions), so a ray's samples never round-trip through HBM between the ~30 torch kernels they replace.
// The file is compiled with -ffp-contract=off: the reference evaluates these expressions as separate fp32 ops,
// and z_vals is required to match it bit for bit.
#include "us_common.h"
#include "dec_adam_dev.h"
#include "pose_step_dev.h"
#include "act_dev.h"
#include <math.h>
#include <string.h>

// ---------------------------------------------------------------------------------------------------------------
// K0a: depth-guided z sampling (Renderer.py:86-101) + jitter (Renderer.py:42-57)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_z(const float* __restrict__ gt_depth, int64_t n_rays,
                                                  const float* __restrict__ t_uni, int n_strat,
                                                  const float* __restrict__ t_surf, int n_imp, float c_free,
                                                  float surf_off, float surf_span, const float* __restrict__ t_rand,
                                                  float* __restrict__ z_vals, int rays_per_block) {
    extern __shared__ __attribute__((aligned(16))) float zs[];       // [rays_per_block][S] sorted samples
    const int S = n_strat + n_imp;
    const int rl = threadIdx.x / S, j = threadIdx.x - rl * S;
    const int64_t ray = (int64_t)blockIdx.x * rays_per_block + rl;
    const bool active = rl < rays_per_block && ray < n_rays;
    float v = 0.0f;
    if (active) {
        const float gt = gt_depth[ray];
        const float fg = c_free * gt, sb = gt - surf_off;
        int rank;
        if (j < n_strat) {
            v = fg * t_uni[j];
            rank = j;
            for (int k = 0; k < n_imp; ++k) rank += ((sb + surf_span * t_surf[k]) < v) ? 1 : 0;
        } else {
            const int k = j - n_strat;
            v = sb + surf_span * t_surf[k];
            rank = k;
            for (int i = 0; i < n_strat; ++i) rank += ((fg * t_uni[i]) <= v) ? 1 : 0;
        }
        zs[rl * S + rank] = v;        // the two ranks form a permutation of 0..S-1 (ties: free samples first)
    }
    __syncthreads();
    if (active) {
        const float* z = zs + rl * S;
        float out = z[j];
        if (t_rand) {
            const float lower = j > 0 ? 0.5f * (z[j] + z[j - 1]) : z[0];
            const float upper = j < S - 1 ? 0.5f * (z[j + 1] + z[j]) : z[S - 1];
            out = lower + (upper - lower) * t_rand[ray * S + j];
        }
        z_vals[ray * S + j] = out;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K0b: points along rays, normalised to the unit cube (Renderer.py:132-137) and its adjoint
// ---------------------------------------------------------------------------------------------------------------
struct Bound3 { float lo[3]; float span[3]; };

__global__ __launch_bounds__(256) void k_ray_points(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const float* __restrict__ z_vals, Bound3 bd, int64_t n_pts, int S,
                                                    float* __restrict__ pts) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pts; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ray = i / S;
        const float z = z_vals[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float p = rays_o[ray * 3 + k] + rays_d[ray * 3 + k] * z;
            pts[i * 3 + k] = (p - bd.lo[k]) / bd.span[k];
        }
    }
}

__global__ __launch_bounds__(256) void k_ray_points_bwd(const float* __restrict__ dpts, const float* __restrict__ dpts2, const float* __restrict__ z_vals,
                                                        Bound3 bd, int64_t n_rays, int S, float* __restrict__ d_o,
                                                        float* __restrict__ d_d) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;                        // wave-uniform
    float so[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
    for (int s = lane; s < S; s += 64) {
        const float z = z_vals[ray * S + s];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float g = dpts[(ray * S + s) * 3 + k];
            if (dpts2) g += dpts2[(ray * S + s) * 3 + k];       // the two grids' shares (us_mlp_bwd_pair_dydx)
            const float gk = g / bd.span[k];
            so[k] += gk; sd[k] += gk * z;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { so[k] = wave_sum(so[k]); sd[k] = wave_sum(sd[k]); }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { if (d_o) d_o[ray * 3 + k] = so[k]; if (d_d) d_d[ray * 3 + k] = sd[k]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K0c: mapping ray assembly, gather first then rotate (common.py:152-166 rotates the whole pool first)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_rays(const float* __restrict__ c2ws, const float* __restrict__ pool_depth,
                                                     const float* __restrict__ pool_color,
                                                     const float* __restrict__ pool_dirs, const int64_t* __restrict__ idx,
                                                     int64_t P, int64_t n_per, int64_t total, float* __restrict__ rays_o,
                                                     float* __restrict__ rays_d, float* __restrict__ depth,
                                                     float* __restrict__ color) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / n_per;
        const int64_t src = f * P + idx[i];
        const float* M = c2ws + f * 16;
        const float d0 = pool_dirs[src * 3 + 0], d1 = pool_dirs[src * 3 + 1], d2 = pool_dirs[src * 3 + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rays_d[i * 3 + k] = (d0 * M[k * 4 + 0] + d1 * M[k * 4 + 1]) + d2 * M[k * 4 + 2];
            rays_o[i * 3 + k] = M[k * 4 + 3];
            color[i * 3 + k] = pool_color[src * 3 + k];
        }
        depth[i] = pool_depth[src];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K0d: bounding-box pre-filter (Mapper.py:396-402 / Tracker.py:177-184): valid = far_bb >= gt_depth [&& gt_depth > 0]
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bbox_filter(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ gt_depth, Bound3 bd, int64_t n_rays,
                                                     int require_depth, uint8_t* __restrict__ valid, float* __restrict__ far_out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rays; i += (int64_t)gridDim.x * blockDim.x) {
        float far = INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float o = rays_o[i * 3 + k], d = rays_d[i * 3 + k];
            const float t0 = (bd.lo[k] - o) / d, t1 = (bd.span[k] - o) / d;     // bd.span carries the UPPER bound here
            far = fminf(far, fmaxf(t0, t1));
        }
        const float gt = gt_depth ? gt_depth[i] : 0.0f;
        if (valid) valid[i] = (far >= gt) && (!require_depth || gt > 0.0f) ? 1 : 0;
        if (far_out) far_out[i] = far;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K0e: K0d + K0a + K0b in one launch (the mapping / tracking iteration when no ray needs the zero-depth branch): validity
// flag, sorted + jittered z, unit-cube points.  Same arithmetic as the three kernels, value for value.
// The jitter draws come from t_rand[R][S] or, when that is NULL, from a counter-based generator (splitmix64 of
// seed + sample index, top 24 bits -> [0,1) like torch.rand): any iid uniform stream serves Renderer.py:42-57.
// ---------------------------------------------------------------------------------------------------------------
struct Bound3x { float lo[3]; float hi[3]; float span[3]; };

__device__ __forceinline__ float uniform24(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void k_sample_points(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ gt_depth, Bound3x bd, int64_t n_rays,
                                                       const float* __restrict__ t_uni, int n_strat,
                                                       const float* __restrict__ t_surf, int n_imp, float c_free,
                                                       float surf_off, float surf_span, const float* __restrict__ t_rand,
                                                       unsigned long long seed, const float* __restrict__ rng_counter,
                                                       int perturb, int require_depth,
                                                       uint8_t* __restrict__ valid, float* __restrict__ z_vals,
                                                       float* __restrict__ pts, int rays_per_block) {
    // a device-side counter (e.g. the optimiser's step count) varies the stream between replays of a captured hipGraph,
    // where the host-side seed is frozen into the launch
    if (rng_counter) seed += 0xD1B54A32D192ED03ull * (unsigned long long)__float_as_uint(rng_counter[0]);
    extern __shared__ __attribute__((aligned(16))) float zs[];       // [rays_per_block][S] sorted samples
    const int S = n_strat + n_imp;
    const int rl = threadIdx.x / S, j = threadIdx.x - rl * S;
    const int64_t ray = (int64_t)blockIdx.x * rays_per_block + rl;
    const bool active = rl < rays_per_block && ray < n_rays;
    float v = 0.0f, gt = 0.0f;
    if (active) {
        gt = gt_depth[ray];
        const float fg = c_free * gt, sb = gt - surf_off;
        int rank;
        if (j < n_strat) {
            v = fg * t_uni[j];
            rank = j;
            for (int k = 0; k < n_imp; ++k) rank += ((sb + surf_span * t_surf[k]) < v) ? 1 : 0;
        } else {
            const int k = j - n_strat;
            v = sb + surf_span * t_surf[k];
            rank = k;
            for (int i = 0; i < n_strat; ++i) rank += ((fg * t_uni[i]) <= v) ? 1 : 0;
        }
        zs[rl * S + rank] = v;
    }
    __syncthreads();
    if (active) {
        const float* z = zs + rl * S;
        float out = z[j];
        if (perturb) {
            const float lower = j > 0 ? 0.5f * (z[j] + z[j - 1]) : z[0];
            const float upper = j < S - 1 ? 0.5f * (z[j + 1] + z[j]) : z[S - 1];
            const float u = t_rand ? t_rand[ray * S + j] : uniform24(seed, (uint64_t)(ray * S + j));
            out = lower + (upper - lower) * u;
        }
        z_vals[ray * S + j] = out;
        float o3[3], d3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { o3[k] = rays_o[ray * 3 + k]; d3[k] = rays_d[ray * 3 + k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float p = o3[k] + d3[k] * out;
            pts[(ray * S + j) * 3 + k] = (p - bd.lo[k]) / bd.span[k];
        }
        if (j == 0 && valid) {
            float far = INFINITY;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float t0 = (bd.lo[k] - o3[k]) / d3[k], t1 = (bd.hi[k] - o3[k]) / d3[k];
                far = fminf(far, fmaxf(t0, t1));
            }
            valid[ray] = (far >= gt) && (!require_depth || gt > 0.0f) ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K5: compositing.  One wavefront per ray; lane l owns samples l*EPL .. l*EPL+EPL-1 (EPL = ceil(S/64) <= 2).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sdf_to_alpha(float sdf, float beta, float& sig, float& e) {
    // Renderer.py:154-158: 1 - exp(-beta * sigmoid(-sdf * beta))
    const float u = -sdf * beta;
    sig = 1.0f / (1.0f + expf(-u));
    e = expf(-beta * sig);
    return 1.0f - e;
}

// exclusive product scan over the 64 lanes
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(inc, o, 64);
        if (lane >= o) inc *= t;
    }
    const float ex = __shfl_up(inc, 1, 64);
    return lane == 0 ? 1.0f : ex;
}
// exclusive suffix sum: result(l) = sum_{k>l} v(k)
__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
    float inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_down(inc, o, 64);
        if (lane + o < 64) inc += t;
    }
    const float ex = __shfl_down(inc, 1, 64);
    return lane == 63 ? 0.0f : ex;
}

// ---- shared by the compositing and the loss kernels -------------------------------------------------------------------
#define LOSS_NSTAT 10
struct LossW { float w[5]; };
struct RayGate { bool sdf_depth; bool color; };

// median == nullptr in US_LOSS_TRK_ORIGINAL: the 10 x median test is applied later (k_track_gate_reduce), the gate is its other half
__device__ __forceinline__ RayGate ray_gate(int mode, float gt, float d, float unc, const float* median) {
    RayGate g; g.sdf_depth = true; g.color = true;
    const bool alpha_mask = (1.0f - unc) > 0.99f;                       // Mapper.py:414-415 / Tracker.py:210-211
    if (mode == US_LOSS_MAP_ORIGINAL) {
        g.sdf_depth = (gt > 0.0f) && alpha_mask;                        // Mapper.py:417-419; colour uses all rays (:427)
    } else if (mode == US_LOSS_TRK_ORIGINAL) {
        const float err = fabsf(gt - d);
        g.sdf_depth = (median == nullptr || err < 10.0f * median[0]) && alpha_mask;          // Tracker.py:214-218
        g.color = g.sdf_depth;                                          // Tracker.py:225
    }
    return g;
}

// loss terms riding on the compositing kernels (us_render_loss_fwd / us_render_loss_bwd): enabled == 0 -> plain compositing
struct LossFwd { int enabled, mode; const uint8_t* valid; const float* gt_depth; const float* gt_color; float tr, tr04; float* partials;
                 float* err; int act; };        // act (US_RENDER_ACT): 0, or US_RENDER_ACT_ON | rgb activation << 12 | sdf activation << 16 -- `raw` arrives as the decoders'
                                                // pre-activation outputs (US_MLP_OUT_PREACT): activated here and written back in place                 // err (US_LOSS_TRK_ORIGINAL): |gt - depth| per ray for the median gate, applied by k_track_gate_reduce
struct LossBwd { int enabled, mode; const uint8_t* valid; const float* gt_depth; const float* gt_color; const float* depth; const float* rgb;
                 const float* unc; float tr, tr04; LossW lw; const float* stats; float* loss_out; const float* median;
                 int act; };                    // act (US_RENDER_ACT): d_raw leaves as the gradient w.r.t. the decoders' PRE-activation outputs (US_MLP_DOUT_PREACT)

template <int EPL>
__global__ __launch_bounds__(256) void k_composite_fwd(const float* __restrict__ raw, const float* __restrict__ z_vals,
                                                       const float* __restrict__ beta_p, int64_t n_rays, int S,
                                                       float* __restrict__ term, float* __restrict__ unc,
                                                       float* __restrict__ depth, float* __restrict__ rgb,
                                                       float* __restrict__ dunc, float* __restrict__ weights, LossFwd lf) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float beta = beta_p[0];
    float a[EPL], z[EPL], c[EPL][3], tl[EPL], sdfv[EPL];
    float tprod = 1.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int s = lane * EPL + e;
        a[e] = 0.0f; z[e] = 0.0f; c[e][0] = c[e][1] = c[e][2] = 0.0f; tl[e] = 1.0f; sdfv[e] = 0.0f;
        if (s < S) {
            float4 r = *reinterpret_cast<const float4*>(raw + (ray * S + s) * 4);
            if (lf.act) {
                // the decoders left their pre-activation outputs (their launches are bound by VALU issue; this one waits on its loads):
                // the same act_fwd, value for value, and `raw` holds the activated sample from here on (the backward passes read it)
                const int ac = (lf.act >> 12) & 15, as = (lf.act >> 16) & 15;
                r.x = act_fwd(r.x, ac); r.y = act_fwd(r.y, ac); r.z = act_fwd(r.z, ac); r.w = act_fwd(r.w, as);
                *reinterpret_cast<float4*>(const_cast<float*>(raw) + (ray * S + s) * 4) = r;
            }
            float sig, ex;
            sdfv[e] = r.w;
            a[e] = sdf_to_alpha(r.w, beta, sig, ex);
            z[e] = z_vals[ray * S + s];
            c[e][0] = r.x; c[e][1] = r.y; c[e][2] = r.z;
            tl[e] = (1.0f - a[e]) + 1e-10f;
        }
        tprod *= tl[e];
    }
    float T = wave_excl_prod(tprod, lane);
    float w[EPL], s_w = 0.f, s_z = 0.f, s_c[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        w[e] = a[e] * T;
        T *= tl[e];
        s_w += w[e]; s_z += w[e] * z[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) s_c[k] += w[e] * c[e][k];
    }
    s_w = wave_sum(s_w); s_z = wave_sum(s_z);
#pragma unroll
    for (int k = 0; k < 3; ++k) s_c[k] = wave_sum(s_c[k]);
    float s_v = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { const float dz = s_z - z[e]; s_v += w[e] * (dz * dz); }
    s_v = wave_sum(s_v);
    if (weights) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) { const int s = lane * EPL + e; if (s < S) weights[ray * S + s] = w[e]; }
    }
    if (lane == 0) {
        term[ray] = s_w;
        unc[ray] = (1.0f - s_w) * (1.0f - s_w);
        depth[ray] = s_z;
        rgb[ray * 3 + 0] = s_c[0]; rgb[ray * 3 + 1] = s_c[1]; rgb[ray * 3 + 2] = s_c[2];
        dunc[ray] = sqrtf(s_v);
    }
    if (lf.enabled) {                                // k_loss_partials on the values this wave holds (mapping modes: no median)
        const float gt = lf.gt_depth[ray], d = s_z;
        RayGate gate = ray_gate(lf.mode, gt, d, (1.0f - s_w) * (1.0f - s_w), nullptr);
        if (lf.valid && !lf.valid[ray]) { gate.sdf_depth = false; gate.color = false; }
        if (lf.err && lane == 0) lf.err[ray] = fabsf(gt - d);
        float s3[3] = {0.f, 0.f, 0.f}, n3[3] = {0.f, 0.f, 0.f};
        if (gate.sdf_depth) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                if (lane * EPL + e < S) {
                    const float zz = z[e], sdf = sdfv[e];
                    const bool front = zz < (gt - lf.tr), back = zz > (gt + lf.tr);
                    const bool center = (zz > (gt - lf.tr04)) && (zz < (gt + lf.tr04));
                    if (front) { const float r = sdf - 1.0f; s3[0] += r * r; n3[0] += 1.f; }
                    else if (center) { const float r = (zz + sdf * lf.tr) - gt; s3[1] += r * r; n3[1] += 1.f; }
                    else if (!back) { const float r = (zz + sdf * lf.tr) - gt; s3[2] += r * r; n3[2] += 1.f; }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { s3[k] = wave_sum(s3[k]); n3[k] = wave_sum(n3[k]); }
        if (lane == 0) {
            float* p = lf.partials + ray * LOSS_NSTAT;
            p[0] = s3[0]; p[1] = s3[1]; p[2] = s3[2]; p[5] = n3[0]; p[6] = n3[1]; p[7] = n3[2];
            float cs = 0.f;
            if (gate.color) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { const float r = lf.gt_color[ray * 3 + k] - s_c[k]; cs += r * r; }
            }
            p[3] = cs; p[8] = gate.color ? 3.f : 0.f;
            const float r = gt - d;
            p[4] = gate.sdf_depth ? r * r : 0.f; p[9] = gate.sdf_depth ? 1.f : 0.f;
        }
    }
}

template <int EPL>
__global__ __launch_bounds__(256) void k_composite_bwd(const float* __restrict__ raw, const float* __restrict__ z_vals,
                                                       const float* __restrict__ beta_p, int64_t n_rays, int S,
                                                       const float* __restrict__ g_term, const float* __restrict__ g_unc,
                                                       const float* __restrict__ g_depth, const float* __restrict__ g_rgb,
                                                       const float* __restrict__ g_dunc, const float* __restrict__ g_sdf,
                                                       float* __restrict__ d_raw, float* __restrict__ d_beta,
                                                       float* __restrict__ beta_partials, LossBwd lb) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (lb.enabled && blockIdx.x == 0 && threadIdx.x == 0 && lb.loss_out) {
        float l = 0.f;
        for (int k = 0; k < 5; ++k) l += lb.lw.w[k] * (lb.stats[k] / lb.stats[5 + k]);      // 0/0 -> NaN like torch.mean([])
        lb.loss_out[0] = l;
    }
    if (ray >= n_rays) return;
    const float beta = beta_p[0];
    float a[EPL], z[EPL], c[EPL][3], tl[EPL], sg[EPL], ex[EPL], sdf[EPL];
    float tprod = 1.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int s = lane * EPL + e;
        a[e] = 0.f; z[e] = 0.f; c[e][0] = c[e][1] = c[e][2] = 0.f; tl[e] = 1.0f; sg[e] = 0.f; ex[e] = 0.f; sdf[e] = 0.f;
        if (s < S) {
            const float4 r = *reinterpret_cast<const float4*>(raw + (ray * S + s) * 4);
            sdf[e] = r.w;
            a[e] = sdf_to_alpha(r.w, beta, sg[e], ex[e]);
            z[e] = z_vals[ray * S + s];
            c[e][0] = r.x; c[e][1] = r.y; c[e][2] = r.z;
            tl[e] = (1.0f - a[e]) + 1e-10f;
        }
        tprod *= tl[e];
    }
    float T0 = wave_excl_prod(tprod, lane);
    float T[EPL], w[EPL], s_w = 0.f, s_z = 0.f;
    {
        float Tr = T0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { T[e] = Tr; w[e] = a[e] * Tr; Tr *= tl[e]; s_w += w[e]; s_z += w[e] * z[e]; }
    }
    s_w = wave_sum(s_w); s_z = wave_sum(s_z);
    const float gt_ = g_term ? g_term[ray] : 0.f, gu = g_unc ? g_unc[ray] : 0.f;
    const float gdu = g_dunc ? g_dunc[ray] : 0.f;
    float gd = g_depth ? g_depth[ray] : 0.f;
    float gc[3] = {0.f, 0.f, 0.f};
    if (g_rgb) { gc[0] = g_rgb[ray * 3]; gc[1] = g_rgb[ray * 3 + 1]; gc[2] = g_rgb[ray * 3 + 2]; }
    // k_loss_grad on the fly (us_render_loss_bwd): the loss gradients wrt depth / colour / sdf of this ray from the statistics
    float gt_l = 0.f, k_fs = 0.f, k_ce = 0.f, k_ta = 0.f;
    bool gate_sdf = false;
    if (lb.enabled) {
        gt_l = lb.gt_depth[ray];
        const float d = lb.depth[ray];
        RayGate gate = ray_gate(lb.mode, gt_l, d, lb.unc[ray], lb.median);
        if (lb.valid && !lb.valid[ray]) { gate.sdf_depth = false; gate.color = false; }
        gate_sdf = gate.sdf_depth;
        k_fs = 2.0f * lb.lw.w[0] / lb.stats[5]; k_ce = 2.0f * lb.lw.w[1] / lb.stats[6]; k_ta = 2.0f * lb.lw.w[2] / lb.stats[7];
        gd = gate.sdf_depth ? (2.0f * lb.lw.w[4] / lb.stats[9]) * (d - gt_l) : 0.f;
        const float kc = 2.0f * lb.lw.w[3] / lb.stats[8];
#pragma unroll
        for (int k = 0; k < 3; ++k) gc[k] = gate.color ? kc * (lb.rgb[ray * 3 + k] - lb.gt_color[ray * 3 + k]) : 0.f;
    }
    float kv = 0.f;                                  // g_dunc / (2 dunc): derivative of sqrt(V)
    if (gdu != 0.f) {
        float s_v = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { const float dz = s_z - z[e]; s_v += w[e] * (dz * dz); }
        s_v = wave_sum(s_v);
        kv = gdu / (2.0f * sqrtf(s_v));
        gd += kv * 2.0f * s_z * (s_w - 1.0f);        // dV/d depth = 2 depth (sum w - 1)
    }
    const float base = gt_ - 2.0f * (1.0f - s_w) * gu;
    float G[EPL], Gw_local = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const float dz = s_z - z[e];
        G[e] = base + gd * z[e] + (gc[0] * c[e][0] + gc[1] * c[e][1] + gc[2] * c[e][2]) + kv * (dz * dz);
        Gw_local += G[e] * w[e];
    }
    // suffix sums of G*w strictly after each element
    const float after_lane = wave_excl_suffix_sum(Gw_local, lane);
    float dbeta_local = 0.f;
    float run = after_lane;                          // sum over elements after the current one (filled backwards)
#pragma unroll
    for (int e = EPL - 1; e >= 0; --e) {
        const int s = lane * EPL + e;
        const float da = G[e] * T[e] - run / tl[e];
        run += G[e] * w[e];
        if (s < S) {
            // a = 1 - exp(-beta*sig), sig = sigmoid(-beta*sdf)
            const float dsig = sg[e] * (1.0f - sg[e]);
            const float da_dsdf = -(beta * beta) * ex[e] * dsig;
            const float da_dbeta = ex[e] * (sg[e] - beta * sdf[e] * dsig);
            float4 o;
            o.x = gc[0] * w[e]; o.y = gc[1] * w[e]; o.z = gc[2] * w[e];
            float gs = g_sdf ? g_sdf[ray * S + s] : 0.f;
            if (lb.enabled && gate_sdf) {
                const bool front = z[e] < (gt_l - lb.tr), back = z[e] > (gt_l + lb.tr);
                const bool center = (z[e] > (gt_l - lb.tr04)) && (z[e] < (gt_l + lb.tr04));
                if (front) gs = k_fs * (sdf[e] - 1.0f);
                else if (center) gs = k_ce * ((z[e] + sdf[e] * lb.tr) - gt_l) * lb.tr;
                else if (!back) gs = k_ta * ((z[e] + sdf[e] * lb.tr) - gt_l) * lb.tr;
            }
            o.w = da * da_dsdf + gs;
            if (lb.act) {                            // ... times the output activations' derivatives: what the decoders' backward pass would form first
                const int ac = (lb.act >> 12) & 15, as = (lb.act >> 16) & 15;
                o.x = o.x * act_bwd(c[e][0], ac); o.y = o.y * act_bwd(c[e][1], ac); o.z = o.z * act_bwd(c[e][2], ac); o.w = o.w * act_bwd(sdf[e], as);
            }
            *reinterpret_cast<float4*>(d_raw + (ray * S + s) * 4) = o;
            dbeta_local += da * da_dbeta;
        }
    }
    if (beta_partials) {                              // one value per ray; k_beta_reduce sums them in a fixed order
        dbeta_local = wave_sum(dbeta_local);
        if (lane == 0) beta_partials[ray] = dbeta_local;
    } else if (d_beta) {                              // fallback: R atomics on ONE address (contended: ~45 us at R = 4096)
        dbeta_local = wave_sum(dbeta_local);
        if (lane == 0 && dbeta_local != 0.f) atomicAdd(d_beta, dbeta_local);
    }
}

__global__ __launch_bounds__(1024) void k_beta_reduce(const float* __restrict__ partials, int64_t n, float* __restrict__ d_beta) {
    __shared__ double sh[1024];
    double acc = 0.0;
    for (int64_t r = threadIdx.x; r < n; r += 1024) acc += (double)partials[r];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) d_beta[0] += (float)sh[0];
}

// ---------------------------------------------------------------------------------------------------------------
// K6: masked losses.  Phase 1: per-ray partial sums/counts -> fixed-order reduction.  Phase 2: gradients.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loss_partials(int mode, const float* __restrict__ sdf_p, int64_t sdf_stride,
                                                       const uint8_t* __restrict__ valid,
                                                       const float* __restrict__ z_vals, const float* __restrict__ gt_depth,
                                                       const float* __restrict__ gt_color, const float* __restrict__ depth,
                                                       const float* __restrict__ rgb, const float* __restrict__ unc,
                                                       const float* __restrict__ median, int64_t n_rays, int S, float tr,
                                                       float tr04, float* __restrict__ partials) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float gt = gt_depth[ray], d = depth[ray];
    RayGate gate = ray_gate(mode, gt, d, unc[ray], median);
    if (valid && !valid[ray]) { gate.sdf_depth = false; gate.color = false; }
    float s[3] = {0.f, 0.f, 0.f}, n[3] = {0.f, 0.f, 0.f};
    if (gate.sdf_depth) {
        for (int k = lane; k < S; k += 64) {
            const float z = z_vals[ray * S + k], sdf = sdf_p[(ray * S + k) * sdf_stride];
            const bool front = z < (gt - tr), back = z > (gt + tr);
            const bool center = (z > (gt - tr04)) && (z < (gt + tr04));
            if (front) { const float r = sdf - 1.0f; s[0] += r * r; n[0] += 1.f; }
            else if (center) { const float r = (z + sdf * tr) - gt; s[1] += r * r; n[1] += 1.f; }   // front & center are disjoint
            else if (!back) { const float r = (z + sdf * tr) - gt; s[2] += r * r; n[2] += 1.f; }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { s[k] = wave_sum(s[k]); n[k] = wave_sum(n[k]); }
    if (lane == 0) {
        float* p = partials + ray * LOSS_NSTAT;
        p[0] = s[0]; p[1] = s[1]; p[2] = s[2]; p[5] = n[0]; p[6] = n[1]; p[7] = n[2];
        float cs = 0.f;
        if (gate.color) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float r = gt_color[ray * 3 + k] - rgb[ray * 3 + k]; cs += r * r; }
        }
        p[3] = cs; p[8] = gate.color ? 3.f : 0.f;
        const float r = gt - d;
        p[4] = gate.sdf_depth ? r * r : 0.f; p[9] = gate.sdf_depth ? 1.f : 0.f;
    }
}

// one workgroup per statistic, fixed summation order -> bitwise reproducible statistics
__global__ __launch_bounds__(1024) void k_loss_reduce(const float* __restrict__ partials, int64_t n_rays,
                                                      float* __restrict__ stats) {
    __shared__ double sh[16];
    const int k = blockIdx.x;
    double acc = 0.0;
    for (int64_t r0 = threadIdx.x; r0 < n_rays; r0 += 4096) {   // 4 independent loads in flight
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int64_t r = r0 + 1024 * u; v[u] = r < n_rays ? partials[r * LOSS_NSTAT + k] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += (double)v[u];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
        for (int w = 0; w < 16; ++w) v += sh[w];
        stats[k] = (float)v;
    }
}

__global__ __launch_bounds__(256) void k_loss_grad(int mode, const float* __restrict__ sdf_p, int64_t sdf_stride,
                                                   const uint8_t* __restrict__ valid, const float* __restrict__ z_vals,
                                                   const float* __restrict__ gt_depth, const float* __restrict__ gt_color,
                                                   const float* __restrict__ depth, const float* __restrict__ rgb,
                                                   const float* __restrict__ unc, const float* __restrict__ median,
                                                   int64_t n_rays, int S, float tr, float tr04, LossW lw,
                                                   const float* __restrict__ stats, float* __restrict__ g_sdf,
                                                   float* __restrict__ g_depth, float* __restrict__ g_rgb,
                                                   float* __restrict__ loss_out) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss_out) {
        float l = 0.f;
        for (int k = 0; k < 5; ++k) l += lw.w[k] * (stats[k] / stats[5 + k]);      // 0/0 -> NaN like torch.mean([])
        loss_out[0] = l;
    }
    if (ray >= n_rays) return;
    const float gt = gt_depth[ray], d = depth[ray];
    RayGate gate = ray_gate(mode, gt, d, unc[ray], median);
    if (valid && !valid[ray]) { gate.sdf_depth = false; gate.color = false; }
    const float k_fs = 2.0f * lw.w[0] / stats[5], k_ce = 2.0f * lw.w[1] / stats[6], k_ta = 2.0f * lw.w[2] / stats[7];
    for (int k = lane; k < S; k += 64) {
        float gval = 0.f;
        if (gate.sdf_depth) {
            const float z = z_vals[ray * S + k], sdf = sdf_p[(ray * S + k) * sdf_stride];
            const bool front = z < (gt - tr), back = z > (gt + tr);
            const bool center = (z > (gt - tr04)) && (z < (gt + tr04));
            if (front) gval = k_fs * (sdf - 1.0f);
            else if (center) gval = k_ce * ((z + sdf * tr) - gt) * tr;
            else if (!back) gval = k_ta * ((z + sdf * tr) - gt) * tr;
        }
        g_sdf[ray * S + k] = gval;
    }
    if (lane == 0) {
        g_depth[ray] = gate.sdf_depth ? (2.0f * lw.w[4] / stats[9]) * (d - gt) : 0.f;
        const float kc = 2.0f * lw.w[3] / stats[8];
#pragma unroll
        for (int k = 0; k < 3; ++k) g_rgb[ray * 3 + k] = gate.color ? kc * (rgb[ray * 3 + k] - gt_color[ray * 3 + k]) : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K0f: importance samples of the rays WITHOUT a depth measurement (Renderer.py:121-130 + common.sample_pdf :49-85): from the SDF
// of a coarse uniform pass to alpha, weights, the un-normalised cdf of the reference (pdf = weights[1:-1], common.py:55-56),
// inverse transform with the draws u, and the sort of the merged samples.  One wavefront per ray; S_u <= 128, n_imp <= 64.
// ---------------------------------------------------------------------------------------------------------------
#define IMP_MAX_U 128
// K0g: the rows of the rays without a depth measurement (the ~gt_mask of Renderer.py:104), in ascending order, and their number.
// One workgroup; the order makes the compacted pass reproducible against a boolean-mask indexing of the same rays.
__global__ __launch_bounds__(1024) void k_zero_depth_rows(const float* __restrict__ gt_depth, int64_t n_rays, int32_t* __restrict__ rows,
                                                          int32_t* __restrict__ count) {
    __shared__ uint32_t wave_n[16];
    __shared__ uint32_t base_sh;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_sh = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < n_rays; i0 += 1024) {
        const int64_t i = i0 + threadIdx.x;
        const bool z = i < n_rays && !(gt_depth[i] > 0.0f);
        const uint64_t m = __ballot(z);
        if (lane == 0) wave_n[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = base_sh;
        for (int w = 0; w < wv; ++w) before += wave_n[w];
        if (z) rows[before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < 16; ++w) t += wave_n[w]; base_sh += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) count[0] = (int32_t)base_sh;
}

// K0h: the coarse uniform pass of those rays (Renderer.py:106-114): far = far_bb + 0.01, z_j = 0*(1-t_j) + far*t_j, the jitter of
// Renderer.py:42-57, and the points normalised with common.normalize_3d_coordinate (to [-1,1], as the reference does THERE; the
// encoder's clamp to [0,1] follows).  One thread per (ray, sample); z_{j+-1} are recomputed instead of exchanged.
__global__ __launch_bounds__(256) void k_uniform_points(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                        const int32_t* __restrict__ rows, int64_t n_rows, Bound3x bd,
                                                        const float* __restrict__ t_uni, int Su, const float* __restrict__ t_rand,
                                                        unsigned long long seed, int perturb, float* __restrict__ z_uni,
                                                        float* __restrict__ pts, const int32_t* __restrict__ n_dev,
                                                        const float* __restrict__ rng_counter) {
    if (n_dev) { const int64_t m = n_dev[0]; n_rows = m < n_rows ? m : n_rows; }     // the row count on the device (us_zero_depth_resample)
    if (rng_counter) seed += 0xD1B54A32D192ED03ull * (unsigned long long)__float_as_uint(rng_counter[0]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * Su; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Su; const int j = (int)(i - r * Su);
        const int64_t ray = rows ? rows[r] : r;
        float o3[3], d3[3], far = INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o3[k] = rays_o[ray * 3 + k]; d3[k] = rays_d[ray * 3 + k];
            const float t0 = (bd.lo[k] - o3[k]) / d3[k], t1 = (bd.hi[k] - o3[k]) / d3[k];
            far = fminf(far, fmaxf(t0, t1));
        }
        far += 0.01f;
        const float zj = far * t_uni[j];
        float out = zj;
        if (perturb) {
            const float lower = j > 0 ? 0.5f * (zj + far * t_uni[j - 1]) : zj;
            const float upper = j < Su - 1 ? 0.5f * (far * t_uni[j + 1] + zj) : zj;
            const float u = t_rand ? t_rand[i] : uniform24(seed, (uint64_t)i);
            out = lower + (upper - lower) * u;
        }
        z_uni[i] = out;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float p = o3[k] + d3[k] * out;
            pts[i * 3 + k] = ((p - bd.lo[k]) / bd.span[k]) * 2.0f - 1.0f;
        }
    }
}

__global__ __launch_bounds__(256) void k_importance_z(const float* __restrict__ sdf, const float* __restrict__ z_uni,
                                                      const float* __restrict__ beta_p, const float* __restrict__ u,
                                                      unsigned long long seed, int64_t n_rays,
                                                      int Su, int n_imp, const int32_t* __restrict__ rows, float* __restrict__ z_out,
                                                      const float* __restrict__ rays_o, const float* __restrict__ rays_d, Bound3x bd,
                                                      float* __restrict__ pts_out, const int32_t* __restrict__ n_dev,
                                                      const float* __restrict__ rng_counter) {
    __shared__ float sh_cdf[4][IMP_MAX_U], sh_bin[4][IMP_MAX_U], sh_all[4][IMP_MAX_U + 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (n_dev) { const int64_t m = n_dev[0]; n_rays = m < n_rays ? m : n_rays; }     // the row count on the device (us_zero_depth_resample)
    if (rng_counter) seed += 0xD1B54A32D192ED03ull * (unsigned long long)__float_as_uint(rng_counter[0]);
    const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
    if (ray >= n_rays) return;                                   // wave-uniform
    float* cdf = sh_cdf[wv]; float* bin = sh_bin[wv]; float* all = sh_all[wv];
    const float beta = beta_p[0];
    // weights w_j = alpha_j * prod_{i<j} (1 - alpha_i + 1e-10)   (lane owns samples 2*lane, 2*lane+1)
    float a[2], z[2], tl[2];
    float tprod = 1.0f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = lane * 2 + e;
        a[e] = 0.f; z[e] = 0.f; tl[e] = 1.0f;
        if (j < Su) {
            float sg, ex;
            a[e] = sdf_to_alpha(sdf[ray * Su + j], beta, sg, ex);
            z[e] = z_uni[ray * Su + j];
            tl[e] = (1.0f - a[e]) + 1e-10f;
        }
        tprod *= tl[e];
    }
    float T = wave_excl_prod(tprod, lane);
    float w[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) { w[e] = a[e] * T; T *= tl[e]; }
    // pdf_k = w_{k+1}, k = 0 .. Su-3 ; cdf_0 = 0, cdf_k = sum_{i=1..k} w_i (k = 1 .. Su-2): inclusive scan of w with w_0 left out
    float pw[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) { const int j = lane * 2 + e; pw[e] = (j >= 1 && j <= Su - 2) ? w[e] : 0.0f; }
    float incl = pw[0] + pw[1];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    const float before = incl - (pw[0] + pw[1]);                  // sum over samples of lower lanes
    // cdf index k corresponds to sample j = k (cdf_j = sum_{i=1..j} w_i), valid for j = 0 .. Su-2
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int j = lane * 2 + e;
        if (j <= Su - 2) cdf[j] = before + (e == 0 ? pw[0] : pw[0] + pw[1]);
        if (j < Su) all[j] = z[e];                               // the uniform samples go first into the merge buffer
    }
    // bins = mids of consecutive uniform samples, k = 0 .. Su-2
    {
        const float z_next_lane = __shfl_down(z[0], 1, 64);
        const int j0 = lane * 2;
        if (j0 + 1 < Su) bin[j0] = 0.5f * (z[1] + z[0]);
        if (j0 + 2 < Su) bin[j0 + 1] = 0.5f * (z_next_lane + z[1]);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0): this wave's LDS writes are done
    const int nb = Su - 1;                                        // entries of cdf and of bins
    if (lane < n_imp) {
        const float uu = u ? u[ray * n_imp + lane] : uniform24(seed, (uint64_t)(ray * n_imp + lane));
        int lo = 0, hi = nb;                                      // searchsorted(right=True): first index with cdf > uu
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= uu) lo = mid + 1; else hi = mid; }
        const int below = lo - 1 > 0 ? lo - 1 : 0, above = lo < nb - 1 ? lo : nb - 1;
        const float cb = cdf[below], ca = cdf[above], bb = bin[below], ba = bin[above];
        float denom = ca - cb;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (uu - cb) / denom;
        all[Su + lane] = bb + t * (ba - bb);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // sort the Su + n_imp merged samples: rank by counting (ties broken by position)
    const int S = Su + n_imp;
    const int64_t dst = rows ? (int64_t)rows[ray] : ray;         // destination row in the full [R][S] sample matrix
    for (int i = lane; i < S; i += 64) {
        const float v = all[i];
        int rank = 0;
        for (int k = 0; k < S; ++k) { const float o = all[k]; rank += (o < v || (o == v && k < i)) ? 1 : 0; }
        z_out[dst * S + rank] = v;
        if (pts_out) {                                             // Renderer.py:132-137 for the row just written
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float p = rays_o[dst * 3 + k] + rays_d[dst * 3 + k] * v;
                pts_out[(dst * S + rank) * 3 + k] = (p - bd.lo[k]) / bd.span[k];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Lower median of |a - b| over the flagged elements (Tracker.py:212-214: depth_error.median() of the rays that passed the
// pre-filter; torch.median returns the lower of the two middle values).  One workgroup; no element flagged -> +inf.
// n <= MEDIAN_MAX.
// ---------------------------------------------------------------------------------------------------------------
#define MEDIAN_MAX 8192
// Radix select on the bit patterns (non-negative floats order like their bits): 4 passes of an 8-bit histogram in LDS.
// b == nullptr: a[] already holds the non-negative values (|gt - depth| left by the compositing kernel).  Every thread returns the median.
__device__ __forceinline__ float median_select(const float* __restrict__ a, const float* __restrict__ b, const uint8_t* __restrict__ valid, int n,
                                               uint32_t* key, uint32_t* hist, uint32_t* sel) {
    if (threadIdx.x == 0) sel[0] = 0;
    __syncthreads();
    uint32_t local = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float av = b ? fabsf(a[i] - b[i]) : a[i];          // unconditional: value and flag in ONE round trip
        const bool ok = !valid || valid[i];
        uint32_t k = 0xFFFFFFFFu;                                // flagged-out elements never enter a histogram
        if (ok) { k = __float_as_uint(av); ++local; }
        key[i] = k;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&sel[0], local);
    __syncthreads();
    const uint32_t cnt = sel[0];
    if (cnt == 0) return INFINITY;
    if (threadIdx.x == 0) { sel[1] = 0; sel[2] = (cnt - 1u) >> 1; }      // lower median: rank (cnt-1)/2, 0-based
    for (int pass = 3; pass >= 0; --pass) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t prefix = sel[1];
        const int sh = 8 * pass;
        if (pass == 3) {
            // sign + high exponent bits: a handful of distinct digits, nearly all elements in one or two of them -- plain atomics would
            // queue up on those addresses.  One atomic per distinct digit and wave: the first lane left names its digit, the lanes
            // that share it are counted by a ballot.  (All lanes of a wave run the same number of outer iterations: n is uniform.)
            for (int i0 = threadIdx.x & ~63; i0 < n; i0 += 1024) {
                const int i = i0 + (threadIdx.x & 63);
                const uint32_t k = i < n ? key[i] : 0xFFFFFFFFu;
                const bool act = k != 0xFFFFFFFFu;
                const uint32_t d = k >> 24;
                uint64_t rem = __ballot(act);
                while (rem) {
                    const int leader = __ffsll((unsigned long long)rem) - 1;
                    const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, leader);
                    const uint64_t same = __ballot(act && d == dl);
                    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[dl], (uint32_t)__popcll(same));
                    rem &= ~same;
                }
            }
        } else {
            for (int i = threadIdx.x; i < n; i += 1024) {
                const uint32_t k = key[i];
                if (k != 0xFFFFFFFFu && (k >> (sh + 8)) == prefix) atomicAdd(&hist[(k >> sh) & 255u], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {                                  // one wave: 4 bins per lane, wave scan, pick the bin of the rank
            const uint32_t rank = sel[2];
            uint32_t c[4], s4 = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c[q] = hist[4 * threadIdx.x + q]; s4 += c[q]; }
            uint32_t inc = s4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if ((int)threadIdx.x >= o) inc += t; }
            uint32_t run = inc - s4;                             // elements in bins before this lane's four
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (rank >= run && rank < run + c[q]) { sel[1] = (prefix << 8) | (4u * threadIdx.x + q); sel[2] = rank - run; }
                run += c[q];
            }
        }
        __syncthreads();
    }
    return __uint_as_float(sel[1]);
}

__global__ __launch_bounds__(1024) void k_masked_median(const float* __restrict__ a, const float* __restrict__ b,
                                                        const uint8_t* __restrict__ valid, int n, float* __restrict__ out) {
    __shared__ uint32_t key[MEDIAN_MAX];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel[3];                                  // count of flagged elements / prefix / remaining rank
    const float m = median_select(a, b, valid, n, key, hist, sel);
    if (threadIdx.x == 0) out[0] = m;
}

// mean of the flagged values (sum / max(count, 1)), one workgroup, fixed-order f64 sums: the tracker's mean rendered uncertainty of the
// rays that passed the pre-filter (src/Tracker.py:353, `rendered_weights.detach().mean()` on the compacted rays)
__global__ __launch_bounds__(1024) void k_masked_mean(const float* __restrict__ a, const uint8_t* __restrict__ valid, int64_t n, float* __restrict__ out) {
    __shared__ double shs[16], shc[16];
    double s = 0.0, c = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float v = a[i];
        const bool ok = !valid || valid[i];
        s += ok ? (double)v : 0.0; c += ok ? 1.0 : 0.0;
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { shs[threadIdx.x >> 6] = s; shc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0.0, C = 0.0;
        for (int w = 0; w < 16; ++w) { S += shs[w]; C += shc[w]; }
        out[0] = (float)(S / (C < 1.0 ? 1.0 : C));
    }
}

// The tracking loss's median gate and its statistics in ONE workgroup (src/Tracker.py:212-238): the compositing kernel left |gt - depth|
// per ray and the ten loss partials of every ray that passes the OTHER half of the gate (pre-filter, alpha mask); here the lower median
// over the pre-filtered rays, then the fixed-order sums of the partials of the rays with err < 10 x median.
__global__ __launch_bounds__(1024) void k_track_gate_reduce(const float* __restrict__ err, const uint8_t* __restrict__ valid,
                                                            const float* __restrict__ partials, int n, float* __restrict__ median_out,
                                                            float* __restrict__ stats) {
    __shared__ uint32_t key[MEDIAN_MAX];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel[3];
    __shared__ double sh[LOSS_NSTAT][16];
    // the ten partials of a ray are five unconditional 8-byte loads, two rays in flight per thread; those of the first 2048 rays (every
    // ray of a Replica / ScanNet tracking batch) are requested BEFORE the median's barriers and arrive under them
    static_assert(LOSS_NSTAT == 10, "five float2 per ray");
    float2 p[2][5];
    auto request = [&](int i0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 1024 * u;
            const float2* src = reinterpret_cast<const float2*>(partials + (int64_t)(i < n ? i : 0) * LOSS_NSTAT);
#pragma unroll
            for (int q = 0; q < 5; ++q) p[u][q] = src[q];
        }
    };
    request(threadIdx.x);
    const float med = median_select(err, nullptr, valid, n, key, hist, sel);
    if (threadIdx.x == 0) median_out[0] = med;
    const float thr = 10.0f * med;
    double acc[LOSS_NSTAT];
#pragma unroll
    for (int k = 0; k < LOSS_NSTAT; ++k) acc[k] = 0.0;
    // the gate comes from the keys in LDS (0xFFFFFFFF = not pre-filtered in)
    for (int i0 = threadIdx.x; i0 < n; i0 += 2048) {
        if (i0 != (int)threadIdx.x) request(i0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = i0 + 1024 * u;
            const uint32_t k = i < n ? key[i] : 0xFFFFFFFFu;
            const bool gate = k != 0xFFFFFFFFu && __uint_as_float(k) < thr;
#pragma unroll
            for (int q = 0; q < 5; ++q) { acc[2 * q] += gate ? (double)p[u][q].x : 0.0; acc[2 * q + 1] += gate ? (double)p[u][q].y : 0.0; }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < LOSS_NSTAT; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) sh[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < LOSS_NSTAT) {
        double v = 0.0;
        for (int w = 0; w < 16; ++w) v += sh[threadIdx.x][w];
        stats[threadIdx.x] = (float)v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam op order: lerp, mul+addcmul, sqrt/div/add, addcdiv)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float one_minus_b1, float b2,
                                              float one_minus_b2, float bc2_sqrt, float eps, float step_size) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + one_minus_b1 * (gi - m[i]);
        const float vi = v[i] * b2 + (one_minus_b2 * gi) * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] + (-step_size) * (mi / denom);
        m[i] = mi; v[i] = vi;
    }
}

// the same over up to ADAM_MAX_TENSORS SEPARATE tensors (each with its own p / g / m / v allocation and learning rate) in one launch:
// the whole optimizer.step() of a torch.optim.Adam whose param_groups hold the decoders' nn.Linear parameters and the two tables
// (src/Mapper.py:118-126,445) -- the drop-in optimiser unislam_amd.optim.Adam.  One-dimensional grid; tensor t owns workgroups
// [blk[t], blk[t+1]).  Tensors whose four pointers are 16-byte aligned take 16-byte accesses (+ a scalar tail), the others 4-byte ones.
#define ADAM_MAX_TENSORS 40
struct AdamTensors {
    float* p[ADAM_MAX_TENSORS]; const float* g[ADAM_MAX_TENSORS]; float* m[ADAM_MAX_TENSORS]; float* v[ADAM_MAX_TENSORS];
    int64_t n[ADAM_MAX_TENSORS]; float step_size[ADAM_MAX_TENSORS]; uint32_t blk[ADAM_MAX_TENSORS + 1]; int n_tensors;
};
__device__ __forceinline__ void adam_one(float gi, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, float one_minus_b1, float b2,
                                         float one_minus_b2, float bc2_sqrt, float eps, float step_size) {
    const float m0 = *m, v0 = *v;
    const float mi = m0 + one_minus_b1 * (gi - m0);
    const float vi = v0 * b2 + (one_minus_b2 * gi) * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    *p = *p + (-step_size) * (mi / denom);
    *m = mi; *v = vi;
}
__global__ __launch_bounds__(256) void k_adam_tensors(const AdamTensors at, float one_minus_b1, float b2, float one_minus_b2, float bc2_sqrt, float eps) {
    int t = 0;
    while (t + 1 < at.n_tensors && blockIdx.x >= at.blk[t + 1]) ++t;
    float* __restrict__ p = at.p[t]; const float* __restrict__ g = at.g[t]; float* __restrict__ m = at.m[t]; float* __restrict__ v = at.v[t];
    const int64_t n = at.n[t];
    const float step_size = at.step_size[t];
    const unsigned bx = blockIdx.x - at.blk[t], gx = at.blk[t + 1] - at.blk[t];
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0);
    typedef float vec_t __attribute__((ext_vector_type(4)));
    const int64_t nv = vec ? n / 4 : 0;
    for (int64_t k = (int64_t)bx * blockDim.x + threadIdx.x; k < nv; k += (int64_t)gx * blockDim.x) {
        const int64_t i = k * 4;
        const vec_t gv = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(g + i));
        const vec_t mv = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(m + i));
        const vec_t vv = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(v + i));
        vec_t pv = *reinterpret_cast<const vec_t*>(p + i), mo, vo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = gv[e], m0 = mv[e], v0 = vv[e];
            const float mi = m0 + one_minus_b1 * (gi - m0);
            const float vi = v0 * b2 + (one_minus_b2 * gi) * gi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            pv[e] = pv[e] + (-step_size) * (mi / denom);
            mo[e] = mi; vo[e] = vi;
        }
        *reinterpret_cast<vec_t*>(p + i) = pv;
        __builtin_nontemporal_store(mo, reinterpret_cast<vec_t*>(m + i)); __builtin_nontemporal_store(vo, reinterpret_cast<vec_t*>(v + i));
    }
    for (int64_t i = nv * 4 + (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += (int64_t)gx * blockDim.x)
        adam_one(g[i], p + i, m + i, v + i, one_minus_b1, b2, one_minus_b2, bc2_sqrt, eps, step_size);
}

// the same over up to ADAM_MAX_SEG segments of one flat parameter buffer, each with its own learning rate: one launch
#define ADAM_MAX_SEG 8
struct AdamSegs { int64_t off[ADAM_MAX_SEG]; int64_t n[ADAM_MAX_SEG]; float step_size[ADAM_MAX_SEG]; };
// step_dev != NULL: the (1-based) step count lives on the device -- k_step_inc advances step_dev[0] and leaves the bias corrections
// beside it, formed with the host path's double arithmetic; sg.step_size then carries the plain learning rates.
// A launch with its step count in the arguments cannot be replayed from a hipGraph; this one can.
// step_dev: float[8] = { count, -, bc1 (double), sqrt(bc2) (double), -, - }: one thread advances the count and forms the corrections
__global__ void k_step_inc(float* step_dev, double beta1, double beta2) {
    const float t = step_dev[0] + 1.0f;
    step_dev[0] = t;
    double* aux = reinterpret_cast<double*>(step_dev + 2);
    aux[0] = 1.0 - pow(beta1, (double)t);
    aux[1] = sqrt(1.0 - pow(beta2, (double)t));
}

// VEC = 4: segments whose offset and length are multiples of 4 floats (16-byte aligned base pointers): one 16-byte access per array and
// thread instead of four 4-byte ones.  Same arithmetic per element.
template <int VEC>
__device__ __forceinline__ void adam_segs_body(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, AdamSegs sg, float one_minus_b1, float b2,
                                                   float one_minus_b2, float bc2_sqrt, float eps, unsigned zero_mask,
                                                   const float* __restrict__ step_dev, const uint16_t* __restrict__ g16,
                                                   unsigned g16_mask, unsigned bx, unsigned by, unsigned gx) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    typedef uint16_t h_t __attribute__((ext_vector_type(VEC)));
    const int64_t n = sg.n[by] / VEC, o = sg.off[by];
    // g16 (us_adam_step_segments_bf16): a bfloat16 image of the gradient buffer (same indexing); the segments flagged in g16_mask read
    // their gradient THERE -- the payload of a data-parallel all-reduce as it came off the wire, no widening pass in between
    const bool narrow = g16 != nullptr && ((g16_mask >> by) & 1u);
    const bool zero = (zero_mask >> by) & 1u;                    // optimizer.zero_grad() of this segment, folded in
    float step_size = sg.step_size[by];
    if (step_dev) {
        const double* aux = reinterpret_cast<const double*>(step_dev + 2);
        step_size = (float)((double)step_size / aux[0]);
        bc2_sqrt = (float)aux[1];
    }
    for (int64_t k = (int64_t)bx * blockDim.x + threadIdx.x; k < n; k += (int64_t)gx * blockDim.x) {
        const int64_t i = o + k * VEC;
        // g, m, v are streamed once per step: non-temporal, so that the tables (p), which the next forward gathers from, stay cached
        vec_t gv;
        if (narrow) {
            const h_t hv = __builtin_nontemporal_load(reinterpret_cast<const h_t*>(g16 + i));
#pragma unroll
            for (int e = 0; e < VEC; ++e) gv[e] = __uint_as_float((uint32_t)hv[e] << 16);
        } else {
            gv = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(g + i));
        }
        const vec_t mv = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(m + i));
        const vec_t vv = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(v + i));
        vec_t pv = *reinterpret_cast<const vec_t*>(p + i), mo, vo;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float gi = gv[e], m0 = mv[e], v0 = vv[e];
            const float mi = m0 + one_minus_b1 * (gi - m0);
            const float vi = v0 * b2 + (one_minus_b2 * gi) * gi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            pv[e] = pv[e] + (-step_size) * (mi / denom);
            mo[e] = mi; vo[e] = vi;
        }
        *reinterpret_cast<vec_t*>(p + i) = pv;
        __builtin_nontemporal_store(mo, reinterpret_cast<vec_t*>(m + i)); __builtin_nontemporal_store(vo, reinterpret_cast<vec_t*>(v + i));
        if (zero) { vec_t z; for (int e = 0; e < VEC; ++e) z[e] = 0.0f; *reinterpret_cast<vec_t*>(g + i) = z; }
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_adam_segs(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, AdamSegs sg,
                                                   float one_minus_b1, float b2, float one_minus_b2, float bc2_sqrt, float eps, unsigned zero_mask,
                                                   const float* __restrict__ step_dev, const uint16_t* __restrict__ g16 = nullptr,
                                                   unsigned g16_mask = 0u) {
    adam_segs_body<VEC>(p, g, m, v, sg, one_minus_b1, b2, one_minus_b2, bc2_sqrt, eps, zero_mask, step_dev, g16, g16_mask, blockIdx.x, blockIdx.y, gridDim.x);
}
// ... with the decoder param group (and the pose group of a joint_opt window) riding along (us_adam_step_model).  A one-dimensional launch:
// the FIRST workgroups are the poses' (one per optimised frame: k_pose_window_step's work), then the decoder
// group's -- gd per decoder (fixed-order sums of their partial rows, then Adam), one for beta (f64 sum of the per-ray partials) -- so they
// are placed first and finish under the tables' stream; the rest are the table segments', each with workgroups in proportion to its size.  (As trailing slices of a
// two-dimensional grid they were placed last, behind 3 x gt workgroups that had nothing to do: 5-10 us SLOWER than two launches.)
// 256 threads walk through what k_mlp_reduce_pair_adam's 1024 do, in the same order: the results are the same bits.
struct SegBlocks { unsigned n[ADAM_MAX_SEG]; };
// the pose group of a joint_opt window riding along (us_adam_step_model with a us_pose_step_desc): n workgroups, the arguments of k_pose_window_step
struct PoseGroup { unsigned n; float* poses7; const float *g_o, *g_d, *dirs; float *m7, *v7, *g7_out, *step_dev; PoseStep ps; const int32_t* shape_dev; int64_t rows_a; };
template <int VEC>
__global__ __launch_bounds__(256) void k_adam_segs_model(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                         AdamSegs sg, unsigned gd, SegBlocks sb, float one_minus_b1, float b2, float one_minus_b2,
                                                         float bc2_sqrt, float eps, unsigned zero_mask, const float* __restrict__ step_dev, DecGroup dg,
                                                         PoseGroup pg) {
    if (blockIdx.x < pg.n) {                                     // the joint_opt window's poses: one workgroup per optimised frame, first of all
        pose_window_step_body<256>((int)blockIdx.x, pg.poses7, pg.g_o, pg.g_d, pg.dirs, pg.m7, pg.v7, pg.g7_out, pg.step_dev, pg.ps, nullptr, nullptr,
                                   nullptr, nullptr, pg.shape_dev, pg.rows_a);
        return;
    }
    const unsigned bid = blockIdx.x - pg.n;
    const unsigned nd = 2u * gd + 1u;
    if (bid >= nd) {                                             // a table segment's workgroup: segment k has blocks[k] of them (in proportion to its size)
        unsigned t = bid - nd, k = 0;
        while (k + 1u < (unsigned)ADAM_MAX_SEG && t >= sb.n[k]) { t -= sb.n[k]; ++k; }
        adam_segs_body<VEC>(p, g, m, v, sg, one_minus_b1, b2, one_minus_b2, bc2_sqrt, eps, zero_mask, step_dev, nullptr, 0u, t, k, sb.n[k]);
        return;
    }
    const int which = bid == nd - 1u ? 2 : (int)(bid / gd);
    const unsigned bx = bid % gd;
    if (which == 2) {                                            // beta: one workgroup
        if (!dg.beta_part) return;
        __shared__ double shd[1024];
        for (int vt = threadIdx.x; vt < 1024; vt += 256) {
            double acc = 0.0;
            for (long long r = vt; r < dg.n_rays; r += 1024) acc += (double)dg.beta_part[r];
            shd[vt] = acc;
        }
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            for (int vt = threadIdx.x; vt < o; vt += 256) shd[vt] += shd[vt + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) { const float gb = (float)shd[0]; *dg.g_beta = gb; dec_adam_apply(gb, dg.p_beta, dg.m_beta, dg.v_beta, dg.ad); }
        return;
    }
    __shared__ float sh[16][64];
    const float* partials = which ? dg.pb : dg.pa;
    const int np = which ? dg.npb : dg.npa;
    const int kl = threadIdx.x & 63, s4 = threadIdx.x >> 6;
    const int k = (int)bx * 64 + kl;
    if ((int)bx * 64 >= np) return;                              // (workgroup-uniform)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = s4 + 4 * q;
        float s = 0.0f;
        if (k < np)
            for (int r = sl; r < dg.n_rows; r += 16) s += partials[(size_t)r * np + k];
        sh[sl][kl] = s;
    }
    __syncthreads();
    if (s4 == 0 && k < np) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[q][kl];
        (which ? dg.gb : dg.ga)[k] = t;
        dec_adam_apply(t, (which ? dg.Pb : dg.Pa) + k, (which ? dg.mb : dg.ma) + k, (which ? dg.vb : dg.va) + k, dg.ad);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Tracking: camera pose -> rays and the adjoint (src/common.py:95-107,196-208 + pytorch3d quaternion_to_matrix).
// pose = (qr, qi, qj, qk, tx, ty, tz); R = I + s*M(q), s = 2/|q|^2; rays_d = R * dir_cam, rays_o = t.
// ---------------------------------------------------------------------------------------------------------------
struct Intr { float fx, fy, cx, cy; int W0, H0, wi; };

__device__ __forceinline__ void quat_rot(const float* q, float R[9]) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1.f - s * (j * j + k * k); R[1] = s * (i * j - k * r); R[2] = s * (i * k + j * r);
    R[3] = s * (i * j + k * r); R[4] = 1.f - s * (i * i + k * k); R[5] = s * (j * k - i * r);
    R[6] = s * (i * k - j * r); R[7] = s * (j * k + i * r); R[8] = 1.f - s * (i * i + j * j);
}

// pixel index inside the crop -> (u, v) -> camera direction -> world ray; also gathers gt depth / colour of the pixel
__global__ __launch_bounds__(256) void k_pose_rays(const float* __restrict__ pose, const int64_t* __restrict__ pix, int64_t n,
                                                   Intr in, const float* __restrict__ depth_img, const float* __restrict__ color_img,
                                                   int W, float* __restrict__ rays_o, float* __restrict__ rays_d,
                                                   float* __restrict__ dirs, float* __restrict__ gt_depth, float* __restrict__ gt_color) {
    float R[9]; quat_rot(pose, R);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = pix[t];
        const int u = in.W0 + (int)(p % in.wi), v = in.H0 + (int)(p / in.wi);
        const float d0 = ((float)u - in.cx) / in.fx, d1 = -((float)v - in.cy) / in.fy, d2 = -1.0f;
        dirs[t * 3] = d0; dirs[t * 3 + 1] = d1; dirs[t * 3 + 2] = d2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rays_d[t * 3 + k] = (d0 * R[k * 3] + d1 * R[k * 3 + 1]) + d2 * R[k * 3 + 2];
            rays_o[t * 3 + k] = pose[4 + k];
        }
        const int64_t px = (int64_t)v * W + u;
        gt_depth[t] = depth_img[px];
#pragma unroll
        for (int k = 0; k < 3; ++k) gt_color[t * 3 + k] = color_img[px * 3 + k];
    }
}

// k_pose_rays + k_sample_points in one launch for the tracker (src/Tracker.py:170-184 + src/utils/Renderer.py:81-101,132-137): thread =
// (ray, sample); every thread of a ray forms the ray from the pose and its pixel (30 flops, nothing to exchange), lane j == 0 leaves
// what the rest of the iteration reads: camera-frame direction (pose gradient), gt depth / colour, validity.  pix == nullptr: the
// pixels are drawn here -- a counter-based uniform draw per ray over the crop, like the reference's torch.randint (src/common.py:116).
// The world-frame rays themselves are not written unless asked for: nothing downstream reads them.
__global__ __launch_bounds__(256) void k_track_sample(const float* __restrict__ pose, const int64_t* __restrict__ pix, int64_t n_rays, Intr in,
                                                      int crop_h, const float* __restrict__ depth_img, const float* __restrict__ color_img, int W,
                                                      Bound3x bd, const float* __restrict__ t_uni, int n_strat, const float* __restrict__ t_surf,
                                                      int n_imp, float c_free, float surf_off, float surf_span, const float* __restrict__ t_rand,
                                                      unsigned long long seed, const float* __restrict__ rng_counter, int perturb,
                                                      float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ dirs,
                                                      float* __restrict__ gt_depth, float* __restrict__ gt_color, uint8_t* __restrict__ valid,
                                                      float* __restrict__ z_vals, float* __restrict__ pts, int rays_per_block) {
    if (rng_counter) seed += 0xD1B54A32D192ED03ull * (unsigned long long)__float_as_uint(rng_counter[0]);
    extern __shared__ __attribute__((aligned(16))) float zs[];       // [rays_per_block][S] sorted samples
    const int S = n_strat + n_imp;
    const int rl = threadIdx.x / S, j = threadIdx.x - rl * S;
    const int64_t ray = (int64_t)blockIdx.x * rays_per_block + rl;
    const bool active = rl < rays_per_block && ray < n_rays;
    float v = 0.0f, gt = 0.0f, o3[3] = {0.f, 0.f, 0.f}, d3[3] = {0.f, 0.f, 0.f}, dc[3] = {0.f, 0.f, 0.f};
    int64_t px = 0;
    if (active) {
        int64_t p;
        if (pix) p = pix[ray];
        else {                                                   // uniform over the crop's pixels: high 32 bits of a 64-bit hash, scaled
            unsigned long long zz = (seed ^ 0xA0761D6478BD642Full) + ((unsigned long long)ray + 1ull) * 0x9E3779B97F4A7C15ull;
            zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull; zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull; zz = zz ^ (zz >> 31);
            p = (int64_t)(((zz >> 32) * (unsigned long long)((int64_t)in.wi * crop_h)) >> 32);
        }
        const int u = in.W0 + (int)(p % in.wi), vv = in.H0 + (int)(p / in.wi);
        dc[0] = ((float)u - in.cx) / in.fx; dc[1] = -((float)vv - in.cy) / in.fy; dc[2] = -1.0f;
        float R[9]; quat_rot(pose, R);
#pragma unroll
        for (int k = 0; k < 3; ++k) { d3[k] = (dc[0] * R[k * 3] + dc[1] * R[k * 3 + 1]) + dc[2] * R[k * 3 + 2]; o3[k] = pose[4 + k]; }
        px = (int64_t)vv * W + u;
        gt = depth_img[px];
        const float fg = c_free * gt, sb = gt - surf_off;
        int rank;
        if (j < n_strat) {
            v = fg * t_uni[j];
            rank = j;
            for (int k = 0; k < n_imp; ++k) rank += ((sb + surf_span * t_surf[k]) < v) ? 1 : 0;
        } else {
            const int k = j - n_strat;
            v = sb + surf_span * t_surf[k];
            rank = k;
            for (int i = 0; i < n_strat; ++i) rank += ((fg * t_uni[i]) <= v) ? 1 : 0;
        }
        zs[rl * S + rank] = v;
    }
    __syncthreads();
    if (active) {
        const float* z = zs + rl * S;
        float out = z[j];
        if (perturb) {
            const float lower = j > 0 ? 0.5f * (z[j] + z[j - 1]) : z[0];
            const float upper = j < S - 1 ? 0.5f * (z[j + 1] + z[j]) : z[S - 1];
            const float u = t_rand ? t_rand[ray * S + j] : uniform24(seed, (uint64_t)(ray * S + j));
            out = lower + (upper - lower) * u;
        }
        z_vals[ray * S + j] = out;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float p = o3[k] + d3[k] * out;
            pts[(ray * S + j) * 3 + k] = (p - bd.lo[k]) / bd.span[k];
        }
        if (j == 0) {
            float far = INFINITY;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float t0 = (bd.lo[k] - o3[k]) / d3[k], t1 = (bd.hi[k] - o3[k]) / d3[k];
                far = fminf(far, fmaxf(t0, t1));
                dirs[ray * 3 + k] = dc[k];
                gt_color[ray * 3 + k] = color_img[px * 3 + k];
                if (rays_o) { rays_o[ray * 3 + k] = o3[k]; rays_d[ray * 3 + k] = d3[k]; }
            }
            gt_depth[ray] = gt;
            valid[ray] = (far >= gt) && (gt > 0.0f) ? 1 : 0;     // Tracker.py:177-184: inside the box AND a depth measurement
        }
    }
}

// The mapping counterpart: us_window_rays (both blocks) + us_sample_points in one launch (src/Mapper.py:372-406 + Renderer.py:81-101,
// 132-137): thread = (ray, sample); the ray's frame follows from its row (rows [0, b n_per): frame row / n_per; the extra block behind
// it: the newest n_extra_frames frames, n_extra pixels each, src/Mapper.py:385-393), its pool pixel from idx_a / idx_b or, when those
// are NULL, from a counter-based uniform draw over the pool (the reference's torch.randint, src/common.py:155).
struct WinShape { int b; long long n_per, P; int xf; long long xn; };
__global__ __launch_bounds__(256) void k_window_sample(const float* __restrict__ c2w_first, const float* __restrict__ poses7, WinShape ws,
                                                       const float* __restrict__ pool_depth, const float* __restrict__ pool_color,
                                                       const float* __restrict__ pool_dirs, const int64_t* __restrict__ idx_a,
                                                       const int64_t* __restrict__ idx_b, int64_t n_rays, Bound3x bd,
                                                       const float* __restrict__ t_uni, int n_strat, const float* __restrict__ t_surf, int n_imp,
                                                       float c_free, float surf_off, float surf_span, const float* __restrict__ t_rand,
                                                       unsigned long long seed, const float* __restrict__ rng_counter, int perturb,
                                                       float* __restrict__ rays_o, float* __restrict__ rays_d, float* __restrict__ dirs,
                                                       float* __restrict__ gt_depth, float* __restrict__ gt_color, uint8_t* __restrict__ valid,
                                                       float* __restrict__ z_vals, float* __restrict__ pts, int rays_per_block,
                                                       const int32_t* __restrict__ shape_dev, const int32_t* __restrict__ slots,
                                                       int64_t rows_a) {
    if (rng_counter) seed += 0xD1B54A32D192ED03ull * (unsigned long long)__float_as_uint(rng_counter[0]);
    // shape_dev (us_arena_window_sample): the window's shape lives on the device -- {frames b, pixels per frame, extra frames, extra
    // pixels, first-frame-fixed} -- and the launch has a FIXED row layout: rows [0, rows_a) hold the b * n_per rays of the first block,
    // rows [rows_a, n_rays) the extra block; rows beyond a block's real rays are padding: sampled like any ray (finite values all the way
    // down) and flagged invalid, i.e. dropped like a ray the pre-filter rejects.  slots: window frame -> row of the pool arena.  One
    // captured hipGraph then serves every window of a run, whatever its number of frames.
    bool first_fixed = c2w_first != nullptr;
    if (shape_dev) {
        ws.b = shape_dev[0]; ws.n_per = shape_dev[1]; ws.xf = shape_dev[2]; ws.xn = shape_dev[3] > 0 ? shape_dev[3] : 1;
        first_fixed = shape_dev[4] != 0;
        if (rng_counter) seed += 0xA24BAED4963EE407ull * (unsigned long long)__float_as_uint(rng_counter[1]);   // the window's epoch (float[8] step_dev)
    }
    extern __shared__ __attribute__((aligned(16))) float zs[];       // [rays_per_block][S] sorted samples
    const int S = n_strat + n_imp;
    const int rl = threadIdx.x / S, j = threadIdx.x - rl * S;
    const int64_t ray = (int64_t)blockIdx.x * rays_per_block + rl;
    const bool active = rl < rays_per_block && ray < n_rays;
    float v = 0.0f, gt = 0.0f, o3[3] = {0.f, 0.f, 0.f}, d3[3] = {0.f, 0.f, 0.f}, dc[3] = {0.f, 0.f, 0.f};
    int64_t src = 0;
    bool padding = false;
    if (active) {
        const int64_t ra = shape_dev ? rows_a : (int64_t)ws.b * ws.n_per;
        int64_t f, p;
        const int64_t* ix;
        int64_t k;
        if (ray < ra) {
            f = ray / ws.n_per; ix = idx_a; k = ray;
            if (f >= ws.b) { padding = true; f = ray % ws.b; }
        } else {
            k = ray - ra; f = (ws.b - ws.xf) + k / ws.xn; ix = idx_b;
            if (k >= (int64_t)ws.xf * ws.xn) { padding = true; f = k % ws.b; }
        }
        if (idx_a) p = ix[k];
        else {
            unsigned long long zz = (seed ^ 0x8CB92BA72F3D8DD7ull) + ((unsigned long long)ray + 1ull) * 0x9E3779B97F4A7C15ull;
            zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull; zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull; zz = zz ^ (zz >> 31);
            p = (int64_t)(((zz >> 32) * (unsigned long long)ws.P) >> 32);
        }
        src = (slots ? (int64_t)slots[f] : f) * ws.P + p;
        float R[9];
        if (f == 0 && first_fixed) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { R[a * 3] = c2w_first[a * 4]; R[a * 3 + 1] = c2w_first[a * 4 + 1]; R[a * 3 + 2] = c2w_first[a * 4 + 2]; o3[a] = c2w_first[a * 4 + 3]; }
        } else {                                                   // no fixed first frame: frame f reads poses7[f]
            const float* q = poses7 + (f - (first_fixed ? 1 : 0)) * 7;
            quat_rot(q, R);
            o3[0] = q[4]; o3[1] = q[5]; o3[2] = q[6];
        }
        dc[0] = pool_dirs[src * 3]; dc[1] = pool_dirs[src * 3 + 1]; dc[2] = pool_dirs[src * 3 + 2];
#pragma unroll
        for (int a = 0; a < 3; ++a) d3[a] = (dc[0] * R[a * 3] + dc[1] * R[a * 3 + 1]) + dc[2] * R[a * 3 + 2];
        gt = pool_depth[src];
        const float fg = c_free * gt, sb = gt - surf_off;
        int rank;
        if (j < n_strat) {
            v = fg * t_uni[j];
            rank = j;
            for (int a = 0; a < n_imp; ++a) rank += ((sb + surf_span * t_surf[a]) < v) ? 1 : 0;
        } else {
            const int a = j - n_strat;
            v = sb + surf_span * t_surf[a];
            rank = a;
            for (int i = 0; i < n_strat; ++i) rank += ((fg * t_uni[i]) <= v) ? 1 : 0;
        }
        zs[rl * S + rank] = v;
    }
    __syncthreads();
    if (active) {
        const float* z = zs + rl * S;
        float out = z[j];
        if (perturb) {
            const float lower = j > 0 ? 0.5f * (z[j] + z[j - 1]) : z[0];
            const float upper = j < S - 1 ? 0.5f * (z[j + 1] + z[j]) : z[S - 1];
            const float u = t_rand ? t_rand[ray * S + j] : uniform24(seed, (uint64_t)(ray * S + j));
            out = lower + (upper - lower) * u;
        }
        z_vals[ray * S + j] = out;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float p = o3[a] + d3[a] * out;
            pts[(ray * S + j) * 3 + a] = (p - bd.lo[a]) / bd.span[a];
        }
        if (j == 0) {
            float far = INFINITY;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float t0 = (bd.lo[a] - o3[a]) / d3[a], t1 = (bd.hi[a] - o3[a]) / d3[a];
                far = fminf(far, fmaxf(t0, t1));
                rays_o[ray * 3 + a] = o3[a]; rays_d[ray * 3 + a] = d3[a];
                if (dirs) dirs[ray * 3 + a] = dc[a];
                gt_color[ray * 3 + a] = pool_color[src * 3 + a];
            }
            gt_depth[ray] = gt;
            valid[ray] = (far >= gt) && !padding ? 1 : 0;        // Mapper.py:396-402 (rays without a depth pass: gt = 0)
        }
    }
}

// one workgroup: G = sum_rays g_d (x) dir, gt = sum_rays g_o, then the closed-form chain rule through R(q)
__global__ __launch_bounds__(1024) void k_pose_grad(const float* __restrict__ pose, const float* __restrict__ g_o,
                                                    const float* __restrict__ g_d, const float* __restrict__ dirs, int64_t n,
                                                    float* __restrict__ g_pose) {
    __shared__ double sh[12][16];
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0;
    for (int64_t t = threadIdx.x; t < n; t += 1024) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float ga = g_d[t * 3 + a];
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[a * 3 + b] += (double)(ga * dirs[t * 3 + b]);
            acc[9 + a] += (double)g_o[t * 3 + a];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) sh[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float G[9], gt[3];
        for (int k = 0; k < 12; ++k) { double v = 0.0; for (int w = 0; w < 16; ++w) v += sh[k][w]; if (k < 9) G[k] = (float)v; else gt[k - 9] = (float)v; }
        const float r = pose[0], i = pose[1], j = pose[2], k = pose[3];
        const float s = 2.0f / (r * r + i * i + j * j + k * k);
        const float M[9] = {-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r,
                            i * k - j * r, j * k + i * r, -(i * i + j * j)};
        const float dMr[9] = {0, -k, j, k, 0, -i, -j, i, 0};
        const float dMi[9] = {0, j, k, j, -2 * i, -r, k, r, -2 * i};
        const float dMj[9] = {-2 * j, i, r, i, 0, k, -r, k, -2 * j};
        const float dMk[9] = {-2 * k, -r, i, r, -2 * k, j, i, j, 0};
        float gm = 0, gr = 0, gi = 0, gj = 0, gk = 0;
        for (int e = 0; e < 9; ++e) { gm += G[e] * M[e]; gr += G[e] * dMr[e]; gi += G[e] * dMi[e]; gj += G[e] * dMj[e]; gk += G[e] * dMk[e]; }
        const float ds = -s * s;                                   // d s / d q_m = -s^2 q_m
        g_pose[0] = ds * r * gm + s * gr; g_pose[1] = ds * i * gm + s * gi;
        g_pose[2] = ds * j * gm + s * gj; g_pose[3] = ds * k * gm + s * gk;
        g_pose[4] = gt[0]; g_pose[5] = gt[1]; g_pose[6] = gt[2];
    }
}

// Adam with the step counter on the device (graph-capturable; torch.optim.Adam(capturable=True) arithmetic)
__global__ __launch_bounds__(256) void k_adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                  const float* __restrict__ step_dev) {
    const float step = step_dev[0];
    const float bc1 = 1.0f - powf(b1, step), bc2s = sqrtf(1.0f - powf(b2, step));
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (1.0f - b1) * (gi - m[i]);
        const float vi = v[i] * b2 + ((1.0f - b2) * gi) * gi;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] = p[i] + (-step_size) * (mi / denom);
        m[i] = mi; v[i] = vi;
    }
}

// The tracker's per-iteration optimiser step in one launch (Tracker.py:322-329,242): Adam with betas (0.5, 0.999) on the 7 pose
// numbers, lr_R for the quaternion and lr_T for the translation, step count on the device and incremented here.
__global__ __launch_bounds__(64) void k_pose_adam(float* __restrict__ pose, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, float lr_R, float lr_T, float b1, float b2, float eps,
                                                  float* __restrict__ step_dev) {
    const float step = step_dev[0] + 1.0f;
    __syncthreads();
    const int i = threadIdx.x;
    if (i == 0) step_dev[0] = step;
    if (i < 7) {
        const float bc1 = 1.0f - powf(b1, step), bc2s = sqrtf(1.0f - powf(b2, step));
        const float step_size = (i < 4 ? lr_R : lr_T) / bc1;
        const float gi = g[i];
        const float mi = m[i] + (1.0f - b1) * (gi - m[i]);
        const float vi = v[i] * b2 + ((1.0f - b2) * gi) * gi;
        const float denom = sqrtf(vi) / bc2s + eps;
        pose[i] = pose[i] + (-step_size) * (mi / denom);
        m[i] = mi; v[i] = vi;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
static unsigned grid_1d(int64_t n, int threads, int cap) {
    int64_t b = us_cdiv(n, threads);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

extern "C" int us_sample_z(const float* gt_depth, int64_t n_rays, const float* t_uni, int n_strat, const float* t_surf,
                           int n_imp, float c_free, float surf_off, float surf_span, const float* t_rand, float* z_vals,
                           void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(gt_depth && t_uni && t_surf && z_vals, US_ERR_NULL, "us_sample_z: NULL pointer");
    const int S = n_strat + n_imp;
    US_REQUIRE(n_strat >= 1 && n_imp >= 0 && S <= 256, US_ERR_SHAPE, "us_sample_z: n_strat %d n_imp %d (S must be <= 256)", n_strat, n_imp);
    const int rpb = 256 / S;
    hipLaunchKernelGGL(k_sample_z, dim3((unsigned)us_cdiv(n_rays, rpb)), dim3(256), (size_t)rpb * S * sizeof(float),
                       (hipStream_t)stream, gt_depth, n_rays, t_uni, n_strat, t_surf, n_imp, c_free, surf_off, surf_span,
                       t_rand, z_vals, rpb);
    US_CHECK_LAUNCH("us_sample_z");
    return US_OK;
}

extern "C" int us_sample_points(const float* rays_o, const float* rays_d, const float* gt_depth, const float* bound_host,
                                int64_t n_rays, const float* t_uni, int n_strat, const float* t_surf, int n_imp, float c_free,
                                float surf_off, float surf_span, const float* t_rand, uint64_t rng_seed, const float* rng_counter,
                                int perturb, int require_depth, uint8_t* valid, float* z_vals, float* pts, void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(rays_o && rays_d && gt_depth && bound_host && t_uni && t_surf && z_vals && pts, US_ERR_NULL, "us_sample_points: NULL pointer");
    const int S = n_strat + n_imp;
    US_REQUIRE(n_strat >= 1 && n_imp >= 0 && S <= 256, US_ERR_SHAPE, "us_sample_points: n_strat %d n_imp %d (S must be <= 256)", n_strat, n_imp);
    Bound3x bd;
    for (int k = 0; k < 3; ++k) { bd.lo[k] = bound_host[k]; bd.hi[k] = bound_host[3 + k]; bd.span[k] = bound_host[3 + k] - bound_host[k]; }
    const int rpb = 256 / S;
    hipLaunchKernelGGL(k_sample_points, dim3((unsigned)us_cdiv(n_rays, rpb)), dim3(256), (size_t)rpb * S * sizeof(float),
                       (hipStream_t)stream, rays_o, rays_d, gt_depth, bd, n_rays, t_uni, n_strat, t_surf, n_imp, c_free, surf_off,
                       surf_span, t_rand, (unsigned long long)rng_seed, rng_counter, perturb, require_depth, valid, z_vals, pts, rpb);
    US_CHECK_LAUNCH("us_sample_points");
    return US_OK;
}

static Bound3 make_bound(const float* b) {
    Bound3 bd;
    for (int k = 0; k < 3; ++k) { bd.lo[k] = b[k]; bd.span[k] = b[3 + k] - b[k]; }
    return bd;
}

extern "C" int us_bbox_filter(const float* rays_o, const float* rays_d, const float* gt_depth, const float* bound_host,
                              int64_t n_rays, int require_depth, uint8_t* valid, float* far_out, void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(rays_o && rays_d && bound_host && (valid || far_out), US_ERR_NULL, "us_bbox_filter: NULL pointer");
    US_REQUIRE(!valid || gt_depth, US_ERR_NULL, "us_bbox_filter: valid[] needs gt_depth");
    Bound3 bd;                                       // lo and HI (not span): the reference divides (bound - o) / d
    for (int k = 0; k < 3; ++k) { bd.lo[k] = bound_host[k]; bd.span[k] = bound_host[3 + k]; }
    hipLaunchKernelGGL(k_bbox_filter, dim3(grid_1d(n_rays, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, gt_depth,
                       bd, n_rays, require_depth, valid, far_out);
    US_CHECK_LAUNCH("us_bbox_filter");
    return US_OK;
}

extern "C" int us_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, const float* bound_host,
                             int64_t n_rays, int n_samples, float* pts, void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(rays_o && rays_d && z_vals && bound_host && pts, US_ERR_NULL, "us_ray_points: NULL pointer");
    US_REQUIRE(n_samples >= 1, US_ERR_SHAPE, "us_ray_points: n_samples %d", n_samples);
    const int64_t n = n_rays * n_samples;
    hipLaunchKernelGGL(k_ray_points, dim3(grid_1d(n, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, z_vals,
                       make_bound(bound_host), n, n_samples, pts);
    US_CHECK_LAUNCH("us_ray_points");
    return US_OK;
}

extern "C" int us_ray_points_bwd(const float* dL_dpts, const float* z_vals, const float* bound_host, int64_t n_rays,
                                 int n_samples, float* dL_do, float* dL_dd, void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(dL_dpts && z_vals && bound_host, US_ERR_NULL, "us_ray_points_bwd: NULL pointer");
    hipLaunchKernelGGL(k_ray_points_bwd, dim3((unsigned)us_cdiv(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, dL_dpts, (const float*)nullptr, z_vals,
                       make_bound(bound_host), n_rays, n_samples, dL_do, dL_dd);
    US_CHECK_LAUNCH("us_ray_points_bwd");
    return US_OK;
}

extern "C" int us_ray_points_bwd2(const float* dL_dpts_a, const float* dL_dpts_b, const float* z_vals, const float* bound_host, int64_t n_rays,
                                  int n_samples, float* dL_do, float* dL_dd, void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(dL_dpts_a && dL_dpts_b && z_vals && bound_host, US_ERR_NULL, "us_ray_points_bwd2: NULL pointer");
    hipLaunchKernelGGL(k_ray_points_bwd, dim3((unsigned)us_cdiv(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, dL_dpts_a, dL_dpts_b, z_vals,
                       make_bound(bound_host), n_rays, n_samples, dL_do, dL_dd);
    US_CHECK_LAUNCH("us_ray_points_bwd2");
    return US_OK;
}

extern "C" int us_gather_rays(const float* c2ws, const float* pool_depth, const float* pool_color, const float* pool_dirs,
                              const int64_t* idx, int b, int64_t pool_size, int64_t n_per_frame, float* rays_o,
                              float* rays_d, float* depth, float* color, void* stream) {
    US_REQUIRE(c2ws && pool_depth && pool_color && pool_dirs && idx && rays_o && rays_d && depth && color, US_ERR_NULL,
               "us_gather_rays: NULL pointer");
    US_REQUIRE(b >= 1 && pool_size >= 1 && n_per_frame >= 0, US_ERR_SHAPE, "us_gather_rays: bad shape");
    const int64_t total = (int64_t)b * n_per_frame;
    if (total == 0) return US_OK;
    hipLaunchKernelGGL(k_gather_rays, dim3(grid_1d(total, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, c2ws, pool_depth,
                       pool_color, pool_dirs, idx, pool_size, n_per_frame, total, rays_o, rays_d, depth, color);
    US_CHECK_LAUNCH("us_gather_rays");
    return US_OK;
}

extern "C" int us_composite_fwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples,
                                float* term, float* pixel_unc, float* depth, float* rgb, float* depth_unc, float* weights,
                                void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(raw && z_vals && beta && term && pixel_unc && depth && rgb && depth_unc, US_ERR_NULL, "us_composite_fwd: NULL pointer");
    US_REQUIRE(n_samples >= 1 && n_samples <= 128, US_ERR_SHAPE, "us_composite_fwd: n_samples %d not in 1..128", n_samples);
    dim3 grid((unsigned)us_cdiv(n_rays, 4)), block(256);
    if (n_samples <= 64)
        hipLaunchKernelGGL((k_composite_fwd<1>), grid, block, 0, (hipStream_t)stream, raw, z_vals, beta, n_rays, n_samples, term, pixel_unc, depth, rgb, depth_unc, weights, LossFwd{});
    else
        hipLaunchKernelGGL((k_composite_fwd<2>), grid, block, 0, (hipStream_t)stream, raw, z_vals, beta, n_rays, n_samples, term, pixel_unc, depth, rgb, depth_unc, weights, LossFwd{});
    US_CHECK_LAUNCH("us_composite_fwd");
    return US_OK;
}

extern "C" int us_composite_bwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples,
                                const float* g_term, const float* g_unc, const float* g_depth, const float* g_rgb,
                                const float* g_dunc, const float* g_sdf, float* d_raw, float* d_beta, float* beta_partials,
                                void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(raw && z_vals && beta && d_raw, US_ERR_NULL, "us_composite_bwd: NULL pointer");
    US_REQUIRE(n_samples >= 1 && n_samples <= 128, US_ERR_SHAPE, "us_composite_bwd: n_samples %d not in 1..128", n_samples);
    dim3 grid((unsigned)us_cdiv(n_rays, 4)), block(256);
    if (n_samples <= 64)
        hipLaunchKernelGGL((k_composite_bwd<1>), grid, block, 0, (hipStream_t)stream, raw, z_vals, beta, n_rays, n_samples, g_term, g_unc, g_depth, g_rgb, g_dunc, g_sdf, d_raw, d_beta, d_beta ? beta_partials : nullptr, LossBwd{});
    else
        hipLaunchKernelGGL((k_composite_bwd<2>), grid, block, 0, (hipStream_t)stream, raw, z_vals, beta, n_rays, n_samples, g_term, g_unc, g_depth, g_rgb, g_dunc, g_sdf, d_raw, d_beta, d_beta ? beta_partials : nullptr, LossBwd{});
    US_CHECK_LAUNCH("us_composite_bwd");
    if (d_beta && beta_partials) {
        hipLaunchKernelGGL(k_beta_reduce, dim3(1), dim3(1024), 0, (hipStream_t)stream, beta_partials, n_rays, d_beta);
        US_CHECK_LAUNCH("us_composite_bwd(beta)");
    }
    return US_OK;
}

extern "C" size_t us_loss_partials_size(int64_t n_rays) { return n_rays > 0 ? (size_t)n_rays * LOSS_NSTAT : 0; }

static int check_loss_mode(const char* fn, int mode, const float* median) {
    US_REQUIRE(mode >= US_LOSS_MAP_ORIGINAL && mode <= US_LOSS_TRK_NOMASK, US_ERR_CONFIG, "%s: mode %d", fn, mode);
    US_REQUIRE(mode != US_LOSS_TRK_ORIGINAL || median, US_ERR_NULL, "%s: tracking 'original' mask needs the median pointer", fn);
    return US_OK;
}

extern "C" int us_loss_stats(int mode, const float* sdf, int64_t sdf_stride, const uint8_t* valid, const float* z_vals,
                             const float* gt_depth, const float* gt_color,
                             const float* depth, const float* rgb, const float* pixel_unc, const float* median,
                             int64_t n_rays, int n_samples, double truncation, float* partials, float* stats, void* stream) {
    int rc = check_loss_mode("us_loss_stats", mode, median); if (rc) return rc;
    US_REQUIRE(sdf && z_vals && gt_depth && gt_color && depth && rgb && pixel_unc && partials && stats, US_ERR_NULL, "us_loss_stats: NULL pointer");
    US_REQUIRE(sdf_stride >= 1, US_ERR_SHAPE, "us_loss_stats: sdf_stride");
    US_REQUIRE(n_rays >= 1 && n_samples >= 1, US_ERR_SHAPE, "us_loss_stats: empty batch");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_partials, dim3((unsigned)us_cdiv(n_rays, 4)), dim3(256), 0, s, mode, sdf, sdf_stride, valid, z_vals, gt_depth, gt_color,
                       depth, rgb, pixel_unc, median, n_rays, n_samples, (float)truncation, (float)(0.4 * truncation), partials);
    US_CHECK_LAUNCH("us_loss_stats(partials)");
    hipLaunchKernelGGL(k_loss_reduce, dim3(LOSS_NSTAT), dim3(1024), 0, s, partials, n_rays, stats);
    US_CHECK_LAUNCH("us_loss_stats(reduce)");
    return US_OK;
}

extern "C" int us_loss_grad(int mode, const float* sdf, int64_t sdf_stride, const uint8_t* valid, const float* z_vals,
                            const float* gt_depth, const float* gt_color,
                            const float* depth, const float* rgb, const float* pixel_unc, const float* median,
                            int64_t n_rays, int n_samples, double truncation, const float* w_host5, const float* stats,
                            float* g_sdf, float* g_depth, float* g_rgb, float* loss_out, void* stream) {
    int rc = check_loss_mode("us_loss_grad", mode, median); if (rc) return rc;
    US_REQUIRE(sdf && z_vals && gt_depth && gt_color && depth && rgb && pixel_unc && w_host5 && stats && g_sdf && g_depth && g_rgb,
               US_ERR_NULL, "us_loss_grad: NULL pointer");
    US_REQUIRE(n_rays >= 1 && n_samples >= 1, US_ERR_SHAPE, "us_loss_grad: empty batch");
    LossW lw; for (int k = 0; k < 5; ++k) lw.w[k] = w_host5[k];
    hipLaunchKernelGGL(k_loss_grad, dim3((unsigned)us_cdiv(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, mode, sdf, sdf_stride, valid, z_vals,
                       gt_depth, gt_color, depth, rgb, pixel_unc, median, n_rays, n_samples, (float)truncation,
                       (float)(0.4 * truncation), lw, stats, g_sdf, g_depth, g_rgb, loss_out);
    US_CHECK_LAUNCH("us_loss_grad");
    return US_OK;
}

// compositing + loss statistics in one launch, loss gradients + compositing backward in one launch (mapping modes: no median)
extern "C" int us_render_loss_fwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples, int mode,
                                  const uint8_t* valid, const float* gt_depth, const float* gt_color, double truncation, float* term,
                                  float* pixel_unc, float* depth, float* rgb, float* depth_unc, float* partials, float* stats, void* stream) {
    const int act = (mode & US_RENDER_ACT_ON) ? (mode & 0x1FF000) : 0;     // US_RENDER_ACT(rgb, sdf): raw holds pre-activation outputs
    mode &= 0xFF;
    US_REQUIRE(mode == US_LOSS_MAP_ORIGINAL || mode == US_LOSS_MAP_NOMASK || mode == US_LOSS_TRK_NOMASK, US_ERR_CONFIG,
               "us_render_loss_fwd: mode %d needs the median of the rendered depth error (use us_composite_fwd + us_loss_stats)", mode);
    US_REQUIRE(n_rays >= 1, US_ERR_SHAPE, "us_render_loss_fwd: empty batch");
    US_REQUIRE(raw && z_vals && beta && gt_depth && gt_color && term && pixel_unc && depth && rgb && depth_unc && partials && stats, US_ERR_NULL,
               "us_render_loss_fwd: NULL pointer");
    US_REQUIRE(n_samples >= 1 && n_samples <= 128, US_ERR_SHAPE, "us_render_loss_fwd: n_samples %d not in 1..128", n_samples);
    LossFwd lf{1, mode, valid, gt_depth, gt_color, (float)truncation, (float)(0.4 * truncation), partials, nullptr, act};
    dim3 grid((unsigned)us_cdiv(n_rays, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (n_samples <= 64)
        hipLaunchKernelGGL((k_composite_fwd<1>), grid, block, 0, s, raw, z_vals, beta, n_rays, n_samples, term, pixel_unc, depth, rgb, depth_unc, (float*)nullptr, lf);
    else
        hipLaunchKernelGGL((k_composite_fwd<2>), grid, block, 0, s, raw, z_vals, beta, n_rays, n_samples, term, pixel_unc, depth, rgb, depth_unc, (float*)nullptr, lf);
    US_CHECK_LAUNCH("us_render_loss_fwd");
    hipLaunchKernelGGL(k_loss_reduce, dim3(LOSS_NSTAT), dim3(1024), 0, s, partials, n_rays, stats);
    US_CHECK_LAUNCH("us_render_loss_fwd(reduce)");
    return US_OK;
}

extern "C" int us_render_loss_bwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples, int mode,
                                  const uint8_t* valid, const float* gt_depth, const float* gt_color, const float* depth, const float* rgb,
                                  const float* pixel_unc, double truncation, const float* w5_host, const float* stats, float* d_raw,
                                  float* d_beta, float* beta_partials, float* loss_out, void* stream) {
    const bool defer_beta = (mode & US_LOSS_DEFER_BETA) != 0;    // the caller sums the per-ray d(beta) partials itself (us_beta_reduce)
    const int act = (mode & US_RENDER_ACT_ON) ? (mode & 0x1FF000) : 0;     // US_RENDER_ACT(rgb, sdf): d_raw w.r.t. pre-activation outputs
    mode &= 0xFF;
    US_REQUIRE(mode == US_LOSS_MAP_ORIGINAL || mode == US_LOSS_MAP_NOMASK || mode == US_LOSS_TRK_NOMASK, US_ERR_CONFIG,
               "us_render_loss_bwd: mode %d needs the median of the rendered depth error (use us_loss_grad + us_composite_bwd)", mode);
    US_REQUIRE(n_rays >= 1, US_ERR_SHAPE, "us_render_loss_bwd: empty batch");
    US_REQUIRE(raw && z_vals && beta && gt_depth && gt_color && depth && rgb && pixel_unc && w5_host && stats && d_raw, US_ERR_NULL,
               "us_render_loss_bwd: NULL pointer");
    US_REQUIRE(n_samples >= 1 && n_samples <= 128, US_ERR_SHAPE, "us_render_loss_bwd: n_samples %d not in 1..128", n_samples);
    LossBwd lb{};
    lb.enabled = 1; lb.mode = mode; lb.valid = valid; lb.gt_depth = gt_depth; lb.gt_color = gt_color; lb.depth = depth; lb.rgb = rgb;
    lb.unc = pixel_unc; lb.tr = (float)truncation; lb.tr04 = (float)(0.4 * truncation); lb.stats = stats; lb.loss_out = loss_out;
    lb.act = act;
    for (int k = 0; k < 5; ++k) lb.lw.w[k] = w5_host[k];
    dim3 grid((unsigned)us_cdiv(n_rays, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const float* nul = nullptr;
    if (n_samples <= 64)
        hipLaunchKernelGGL((k_composite_bwd<1>), grid, block, 0, s, raw, z_vals, beta, n_rays, n_samples, nul, nul, nul, nul, nul, nul, d_raw, d_beta,
                           d_beta ? beta_partials : nullptr, lb);
    else
        hipLaunchKernelGGL((k_composite_bwd<2>), grid, block, 0, s, raw, z_vals, beta, n_rays, n_samples, nul, nul, nul, nul, nul, nul, d_raw, d_beta,
                           d_beta ? beta_partials : nullptr, lb);
    US_CHECK_LAUNCH("us_render_loss_bwd");
    if (d_beta && beta_partials && !defer_beta) {
        hipLaunchKernelGGL(k_beta_reduce, dim3(1), dim3(1024), 0, s, beta_partials, n_rays, d_beta);
        US_CHECK_LAUNCH("us_render_loss_bwd(beta)");
    }
    return US_OK;
}

extern "C" int us_beta_reduce(const float* beta_partials, int64_t n_rays, float* d_beta, void* stream) {
    US_REQUIRE(beta_partials && d_beta, US_ERR_NULL, "us_beta_reduce: NULL pointer");
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    hipLaunchKernelGGL(k_beta_reduce, dim3(1), dim3(1024), 0, (hipStream_t)stream, beta_partials, n_rays, d_beta);
    US_CHECK_LAUNCH("us_beta_reduce");
    return US_OK;
}

static Bound3x make_bound3x(const float* bound_host) {
    Bound3x bd;
    for (int k = 0; k < 3; ++k) { bd.lo[k] = bound_host[k]; bd.hi[k] = bound_host[3 + k]; bd.span[k] = bound_host[3 + k] - bound_host[k]; }
    return bd;
}

extern "C" int us_importance_z_rows(const float* sdf_uni, const float* z_uni, const float* beta, const float* u, uint64_t rng_seed,
                                    int64_t n_rows, int n_uniform, int n_importance, const int32_t* rows, float* z_out,
                                    const float* rays_o, const float* rays_d, const float* bound_host, float* pts_out, void* stream) {
    if (n_rows <= 0) return n_rows == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(sdf_uni && z_uni && beta && z_out, US_ERR_NULL, "us_importance_z: NULL pointer");
    US_REQUIRE(!pts_out || (rays_o && rays_d && bound_host), US_ERR_NULL, "us_importance_z: pts_out needs rays_o, rays_d and the bound");
    US_REQUIRE(n_uniform >= 3 && n_uniform <= IMP_MAX_U && n_importance >= 1 && n_importance <= 64, US_ERR_SHAPE,
               "us_importance_z: n_uniform %d not in 3..%d or n_importance %d not in 1..64", n_uniform, IMP_MAX_U, n_importance);
    Bound3x bd = {};
    if (pts_out) bd = make_bound3x(bound_host);
    hipLaunchKernelGGL(k_importance_z, dim3((unsigned)us_cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, sdf_uni, z_uni, beta, u,
                       (unsigned long long)rng_seed, n_rows, n_uniform, n_importance, rows, z_out, rays_o, rays_d, bd, pts_out,
                       (const int32_t*)nullptr, (const float*)nullptr);
    US_CHECK_LAUNCH("us_importance_z");
    return US_OK;
}

extern "C" int us_importance_z(const float* sdf_uni, const float* z_uni, const float* beta, const float* u, int64_t n_rays, int n_uniform,
                               int n_importance, float* z_out, void* stream) {
    US_REQUIRE(u || n_rays <= 0, US_ERR_NULL, "us_importance_z: NULL pointer");
    return us_importance_z_rows(sdf_uni, z_uni, beta, u, 0, n_rays, n_uniform, n_importance, nullptr, z_out, nullptr, nullptr, nullptr,
                                nullptr, stream);
}

extern "C" int us_zero_depth_rows(const float* gt_depth, int64_t n_rays, int32_t* rows, int32_t* count, void* stream) {
    US_REQUIRE(gt_depth && rows && count, US_ERR_NULL, "us_zero_depth_rows: NULL pointer");
    US_REQUIRE(n_rays >= 0 && n_rays <= 0x7fffffff, US_ERR_SHAPE, "us_zero_depth_rows: n_rays %lld", (long long)n_rays);
    hipLaunchKernelGGL(k_zero_depth_rows, dim3(1), dim3(1024), 0, (hipStream_t)stream, gt_depth, n_rays, rows, count);
    US_CHECK_LAUNCH("us_zero_depth_rows");
    return US_OK;
}

extern "C" int us_uniform_points(const float* rays_o, const float* rays_d, const int32_t* rows, int64_t n_rows, const float* bound_host,
                                 const float* t_uni, int n_uniform, const float* t_rand, uint64_t rng_seed, int perturb, float* z_uni,
                                 float* pts, void* stream) {
    if (n_rows <= 0) return n_rows == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(rays_o && rays_d && bound_host && t_uni && z_uni && pts, US_ERR_NULL, "us_uniform_points: NULL pointer");
    US_REQUIRE(n_uniform >= 1, US_ERR_SHAPE, "us_uniform_points: n_uniform %d", n_uniform);
    hipLaunchKernelGGL(k_uniform_points, dim3(grid_1d(n_rows * n_uniform, 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d,
                       rows, n_rows, make_bound3x(bound_host), t_uni, n_uniform, t_rand, (unsigned long long)rng_seed, perturb, z_uni, pts,
                       (const int32_t*)nullptr, (const float*)nullptr);
    US_CHECK_LAUNCH("us_uniform_points");
    return US_OK;
}

// Renderer.py:104-130 for the rays WITHOUT a depth measurement with no row count on the host: compaction (rows, count[0] on the device),
// then the coarse uniform pass, the sdf grid, its decoder and the importance resampling, each launched for all n_rays rows and reading
// count[0] on the device -- workgroups beyond it leave at once.  The whole branch can be captured into a hipGraph.
int us_hashgrid_fwd_counted_rows(const us_grid_desc* d, const float* params, const float* x, int64_t n, float* out, float* dy_dx, int flags,
                                 const int32_t* n_dev, int n_mul, void* stream);                        // hashgrid.hip
int us_mlp_fwd_counted_rows(const us_mlp_desc* d, const float* params, const float* in, int64_t n, float* out, int64_t out_stride, int flags,
                            const int32_t* n_dev, int n_mul, void* stream);                             // mlp.hip

extern "C" int us_zero_depth_resample(const us_grid_desc* grid, const float* table, const us_mlp_desc* mlp, const float* mlp_params,
                                      const float* beta, const float* rays_o, const float* rays_d, const float* gt_depth, int64_t n_rays,
                                      const float* bound_host, const float* t_uni, int n_uniform, int n_importance, const float* t_rand,
                                      const float* u, uint64_t seed_uniform, uint64_t seed_importance, const float* rng_counter,
                                      int perturb, int32_t* rows, int32_t* count, float* z_uni, float* pts_uni, float* feat, float* sdf_uni, float* z_out, float* pts_out,
                                      void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(grid && table && mlp && mlp_params && beta && rays_o && rays_d && gt_depth && bound_host && t_uni && rows && count && z_uni &&
               pts_uni && feat && sdf_uni && z_out && pts_out, US_ERR_NULL, "us_zero_depth_resample: NULL pointer");
    US_REQUIRE(n_rays * (int64_t)n_uniform <= 0x7fffffff, US_ERR_SHAPE, "us_zero_depth_resample: n_rays %lld x n_uniform %d", (long long)n_rays, n_uniform);
    US_REQUIRE(n_uniform >= 3 && n_uniform <= IMP_MAX_U && n_importance >= 1 && n_importance <= 64, US_ERR_SHAPE,
               "us_zero_depth_resample: n_uniform %d not in 3..%d or n_importance %d not in 1..64", n_uniform, IMP_MAX_U, n_importance);
    US_REQUIRE(mlp->n_in == grid->n_levels * grid->n_features && mlp->n_out == 1, US_ERR_CONFIG,
               "us_zero_depth_resample: the decoder maps %u -> %u, the grid leaves %u features", mlp->n_in, mlp->n_out, grid->n_levels * grid->n_features);
    int rc = us_zero_depth_rows(gt_depth, n_rays, rows, count, stream); if (rc) return rc;
    const Bound3x bd = make_bound3x(bound_host);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_uniform_points, dim3(grid_1d(n_rays * n_uniform, 256, 1 << 20)), dim3(256), 0, s, rays_o, rays_d, (const int32_t*)rows,
                       n_rays, bd, t_uni, n_uniform, perturb ? t_rand : nullptr, (unsigned long long)seed_uniform, perturb, z_uni, pts_uni,
                       (const int32_t*)count, rng_counter);
    US_CHECK_LAUNCH("us_zero_depth_resample (uniform points)");
    const int64_t n = n_rays * n_uniform;
    rc = us_hashgrid_fwd_counted_rows(grid, table, pts_uni, n, feat, nullptr, US_GRID_CLAMP01 | US_GRID_LEVEL_MAJOR, count, n_uniform, stream);
    if (rc) return rc;
    rc = us_mlp_fwd_counted_rows(mlp, mlp_params, feat, n, sdf_uni, 1, US_MLP_LEVEL_MAJOR, count, n_uniform, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_importance_z, dim3((unsigned)us_cdiv(n_rays, 4)), dim3(256), 0, s, (const float*)sdf_uni, (const float*)z_uni, beta, u,
                       (unsigned long long)seed_importance, n_rays, n_uniform, n_importance, (const int32_t*)rows, z_out, rays_o, rays_d, bd, pts_out,
                       (const int32_t*)count, rng_counter);
    US_CHECK_LAUNCH("us_zero_depth_resample (importance samples)");
    return US_OK;
}

extern "C" int us_pose_adam_step(float* pose, const float* g, float* m, float* v, double lr_R, double lr_T, double beta1, double beta2,
                                 double eps, float* step_dev, void* stream) {
    US_REQUIRE(pose && g && m && v && step_dev, US_ERR_NULL, "us_pose_adam_step: NULL pointer");
    hipLaunchKernelGGL(k_pose_adam, dim3(1), dim3(64), 0, (hipStream_t)stream, pose, g, m, v, (float)lr_R, (float)lr_T, (float)beta1,
                       (float)beta2, (float)eps, step_dev);
    US_CHECK_LAUNCH("us_pose_adam_step");
    return US_OK;
}

extern "C" int us_masked_median(const float* a, const float* b, const uint8_t* valid, int64_t n, float* out, void* stream) {
    US_REQUIRE(a && b && out, US_ERR_NULL, "us_masked_median: NULL pointer");
    US_REQUIRE(n >= 0 && n <= MEDIAN_MAX, US_ERR_SHAPE, "us_masked_median: n = %lld not in 0..%d", (long long)n, MEDIAN_MAX);
    hipLaunchKernelGGL(k_masked_median, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, valid, (int)n, out);
    US_CHECK_LAUNCH("us_masked_median");
    return US_OK;
}

extern "C" int us_masked_mean(const float* a, const uint8_t* valid, int64_t n, float* out, void* stream) {
    US_REQUIRE(a && out, US_ERR_NULL, "us_masked_mean: NULL pointer");
    US_REQUIRE(n >= 0, US_ERR_SHAPE, "us_masked_mean: n = %lld", (long long)n);
    hipLaunchKernelGGL(k_masked_mean, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, valid, n, out);
    US_CHECK_LAUNCH("us_masked_mean");
    return US_OK;
}

extern "C" int us_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                            double eps, int step, void* stream) {
    US_REQUIRE(p && g && m && v, US_ERR_NULL, "us_adam_step: NULL pointer");
    US_REQUIRE(step >= 1, US_ERR_SHAPE, "us_adam_step: step %d (1-based)", step);
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(k_adam, dim3(grid_1d(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)(lr / bc1));
    US_CHECK_LAUNCH("us_adam_step");
    return US_OK;
}

#ifndef ADAM_VEC_BLOCKS
#define ADAM_VEC_BLOCKS 8192
#endif
static int adam_segments(float* p, float* g, float* m, float* v, int n_seg, const int64_t* seg_off, const int64_t* seg_n,
                         const double* seg_lr, double beta1, double beta2, double eps, int step, float* step_dev,
                         unsigned zero_grad_mask, void* stream, const uint16_t* g16 = nullptr, unsigned g16_mask = 0u) {
    US_REQUIRE(p && g && m && v && seg_off && seg_n && seg_lr, US_ERR_NULL, "us_adam_step_segments: NULL pointer");
    US_REQUIRE(n_seg >= 1 && n_seg <= ADAM_MAX_SEG, US_ERR_SHAPE, "us_adam_step_segments: n_seg %d not in 1..%d", n_seg, ADAM_MAX_SEG);
    US_REQUIRE(step_dev || step >= 1, US_ERR_SHAPE, "us_adam_step_segments: step %d (1-based)", step);
    const double bc1 = step_dev ? 1.0 : 1.0 - pow(beta1, (double)step), bc2 = step_dev ? 1.0 : 1.0 - pow(beta2, (double)step);
    AdamSegs sg;
    memset(&sg, 0, sizeof(sg));
    int64_t n_max = 0;
    for (int k = 0; k < n_seg; ++k) {
        US_REQUIRE(seg_off[k] >= 0 && seg_n[k] >= 0, US_ERR_SHAPE, "us_adam_step_segments: segment %d: offset %lld n %lld", k,
                   (long long)seg_off[k], (long long)seg_n[k]);
        sg.off[k] = seg_off[k]; sg.n[k] = seg_n[k]; sg.step_size[k] = (float)(seg_lr[k] / bc1);
        if (seg_n[k] > n_max) n_max = seg_n[k];
    }
    const bool advanced = (zero_grad_mask & US_ADAM_STEP_ADVANCED) != 0;      // us_adam_step_inc has run for this step already
    zero_grad_mask &= ~US_ADAM_STEP_ADVANCED;
    if (step_dev && !advanced) hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, beta1, beta2);   // also when no element is owned
    if (n_max == 0) return US_OK;
    bool vec4 = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0;
    for (int k = 0; k < n_seg; ++k) vec4 = vec4 && (seg_off[k] % 4 == 0) && (seg_n[k] % 4 == 0);
    if (vec4)
        hipLaunchKernelGGL(k_adam_segs<4>, dim3(grid_1d(n_max / 4, 256, ADAM_VEC_BLOCKS), n_seg), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sg,
                           (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, zero_grad_mask,
                           (const float*)step_dev, g16, g16_mask);
    else
        hipLaunchKernelGGL(k_adam_segs<1>, dim3(grid_1d(n_max, 256, 4096), n_seg), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sg,
                           (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, zero_grad_mask,
                           (const float*)step_dev, g16, g16_mask);
    US_CHECK_LAUNCH("us_adam_step_segments");
    return US_OK;
}

extern "C" int us_adam_step_tensors(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                                    const double* lr, double beta1, double beta2, double eps, int step, void* stream) {
    US_REQUIRE(p && g && m && v && n && lr, US_ERR_NULL, "us_adam_step_tensors: NULL pointer");
    US_REQUIRE(n_tensors >= 0 && n_tensors <= ADAM_MAX_TENSORS, US_ERR_SHAPE, "us_adam_step_tensors: n_tensors %d not in 0..%d", n_tensors, ADAM_MAX_TENSORS);
    US_REQUIRE(step >= 1, US_ERR_SHAPE, "us_adam_step_tensors: step %d (1-based)", step);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamTensors at;
    memset(&at, 0, sizeof(at));
    uint32_t blocks = 0;
    int k = 0;
    for (int t = 0; t < n_tensors; ++t) {
        US_REQUIRE(n[t] >= 0, US_ERR_SHAPE, "us_adam_step_tensors: tensor %d: n %lld", t, (long long)n[t]);
        if (n[t] == 0) continue;
        US_REQUIRE(p[t] && g[t] && m[t] && v[t], US_ERR_NULL, "us_adam_step_tensors: tensor %d: NULL pointer", t);
        at.p[k] = p[t]; at.g[k] = g[t]; at.m[k] = m[t]; at.v[k] = v[t]; at.n[k] = n[t]; at.step_size[k] = (float)(lr[t] / bc1);
        at.blk[k] = blocks;
        blocks += (uint32_t)grid_1d((n[t] + 3) / 4, 256, ADAM_VEC_BLOCKS);
        ++k;
    }
    at.blk[k] = blocks; at.n_tensors = k;
    if (k == 0) return US_OK;
    hipLaunchKernelGGL(k_adam_tensors, dim3(blocks), dim3(256), 0, (hipStream_t)stream, at, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)sqrt(bc2), (float)eps);
    US_CHECK_LAUNCH("us_adam_step_tensors");
    return US_OK;
}

int us_adam_segments_model(float* p, float* g, float* m, float* v, int n_seg, const int64_t* seg_off, const int64_t* seg_n, const double* seg_lr,
                           double beta1, double beta2, double eps, float* step_dev, unsigned zero_grad_mask, void* stream, const DecGroup& dg,
                           const us_pose_step_desc* poses) {
    US_REQUIRE(p && g && m && v && seg_off && seg_n && seg_lr && step_dev, US_ERR_NULL, "us_adam_step_model: NULL pointer");
    PoseGroup pg;
    memset(&pg, 0, sizeof(pg));
    if (poses) {
        US_REQUIRE(poses->poses7 && poses->g_rays_o && poses->g_rays_d && poses->dirs && poses->m7 && poses->v7, US_ERR_NULL, "us_adam_step_model: pose step: NULL pointer");
        US_REQUIRE(poses->n_poses >= 1 && poses->n_poses <= 4096, US_ERR_SHAPE, "us_adam_step_model: pose step: n_poses %d", poses->n_poses);
        US_REQUIRE(poses->shape_dev || (poses->n_a >= 0 && poses->n_b >= 0 && poses->row_a >= 0 && poses->row_b >= 0 && poses->first_pose_b >= 0), US_ERR_SHAPE,
                   "us_adam_step_model: pose step: bad shape");
        pg.n = (unsigned)poses->n_poses; pg.poses7 = poses->poses7; pg.g_o = poses->g_rays_o; pg.g_d = poses->g_rays_d; pg.dirs = poses->dirs;
        pg.m7 = poses->m7; pg.v7 = poses->v7; pg.g7_out = poses->g7_out; pg.step_dev = step_dev;
        pg.ps.nA = poses->n_a; pg.ps.rowA = poses->row_a; pg.ps.nB = poses->n_b; pg.ps.rowB = poses->row_b; pg.ps.jB = poses->first_pose_b;
        pg.ps.lr_q = (float)poses->lr_q; pg.ps.lr_t = (float)poses->lr_t; pg.ps.b1 = (float)beta1; pg.ps.b2 = (float)beta2; pg.ps.eps = (float)eps;
        pg.ps.own_step = 0; pg.ps.apply = 1;
        pg.shape_dev = poses->shape_dev; pg.rows_a = poses->rows_a;
    }
    US_REQUIRE(n_seg >= 1 && n_seg <= ADAM_MAX_SEG, US_ERR_SHAPE, "us_adam_step_model: n_seg %d not in 1..%d", n_seg, ADAM_MAX_SEG);
    AdamSegs sg;
    memset(&sg, 0, sizeof(sg));
    int64_t n_max = 0;
    bool vec4 = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0;
    for (int k = 0; k < n_seg; ++k) {
        US_REQUIRE(seg_off[k] >= 0 && seg_n[k] >= 0, US_ERR_SHAPE, "us_adam_step_model: segment %d: offset %lld n %lld", k, (long long)seg_off[k], (long long)seg_n[k]);
        sg.off[k] = seg_off[k]; sg.n[k] = seg_n[k]; sg.step_size[k] = (float)seg_lr[k];
        if (seg_n[k] > n_max) n_max = seg_n[k];
        vec4 = vec4 && (seg_off[k] % 4 == 0) && (seg_n[k] % 4 == 0);
    }
    const bool advanced = (zero_grad_mask & US_ADAM_STEP_ADVANCED) != 0;
    zero_grad_mask &= ~US_ADAM_STEP_ADVANCED;
    if (!advanced) hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, beta1, beta2);
    const int np = dg.npa > dg.npb ? dg.npa : dg.npb;
    const unsigned gd = (unsigned)us_cdiv(np, 64);               // workgroups per decoder
    const int vec = vec4 ? 4 : 1;
    const unsigned cap = vec4 ? ADAM_VEC_BLOCKS : 4096;
    int64_t n_sum = 0;
    for (int k = 0; k < n_seg; ++k) n_sum += seg_n[k];
    SegBlocks sb;
    memset(&sb, 0, sizeof(sb));
    unsigned total = 0;
    for (int k = 0; k < n_seg; ++k) {                            // <= cap workgroups in all, none with less than one turn of 256 elements if avoidable
        const int64_t want = us_cdiv(seg_n[k] / vec, 256), share = n_sum > 0 ? (int64_t)((double)cap * (double)seg_n[k] / (double)n_sum) : 1;
        int64_t nb = want < share ? want : share;
        if (nb < 1) nb = 1;
        sb.n[k] = (unsigned)nb; total += (unsigned)nb;
    }
    if (vec4)
        hipLaunchKernelGGL(k_adam_segs_model<4>, dim3(pg.n + 2u * gd + 1u + total), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sg, gd, sb,
                           (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 1.0f, (float)eps, zero_grad_mask, (const float*)step_dev, dg, pg);
    else
        hipLaunchKernelGGL(k_adam_segs_model<1>, dim3(pg.n + 2u * gd + 1u + total), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sg, gd, sb,
                           (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 1.0f, (float)eps, zero_grad_mask, (const float*)step_dev, dg, pg);
    US_CHECK_LAUNCH("us_adam_step_model");
    return US_OK;
}

extern "C" int us_adam_step_segments(float* p, float* g, float* m, float* v, int n_seg, const int64_t* seg_off,
                                     const int64_t* seg_n, const double* seg_lr, double beta1, double beta2, double eps, int step,
                                     unsigned zero_grad_mask, void* stream) {
    return adam_segments(p, g, m, v, n_seg, seg_off, seg_n, seg_lr, beta1, beta2, eps, step, nullptr, zero_grad_mask, stream);
}

extern "C" int us_adam_step_segments_dev(float* p, float* g, float* m, float* v, int n_seg, const int64_t* seg_off,
                                         const int64_t* seg_n, const double* seg_lr, double beta1, double beta2, double eps,
                                         float* step_dev, unsigned zero_grad_mask, void* stream) {
    US_REQUIRE(step_dev && ((uintptr_t)step_dev & 7u) == 0, US_ERR_NULL, "us_adam_step_segments_dev: step_dev is NULL or not 8-byte aligned");
    return adam_segments(p, g, m, v, n_seg, seg_off, seg_n, seg_lr, beta1, beta2, eps, 0, step_dev, zero_grad_mask, stream);
}

// us_adam_step_segments_dev where the segments flagged in bf16_mask (bit k = segment k) read their gradient from g_bf16, a bfloat16 image
// of the gradient buffer with the same indexing (the payload of the data-parallel all-reduce, dist.GradComm) instead of from g
extern "C" int us_adam_step_segments_bf16(float* p, float* g, const uint16_t* g_bf16, unsigned bf16_mask, float* m, float* v, int n_seg,
                                          const int64_t* seg_off, const int64_t* seg_n, const double* seg_lr, double beta1, double beta2,
                                          double eps, float* step_dev, unsigned zero_grad_mask, void* stream) {
    US_REQUIRE(step_dev && ((uintptr_t)step_dev & 7u) == 0, US_ERR_NULL, "us_adam_step_segments_bf16: step_dev is NULL or not 8-byte aligned");
    US_REQUIRE(g_bf16 || bf16_mask == 0u, US_ERR_NULL, "us_adam_step_segments_bf16: g_bf16 is NULL");
    US_REQUIRE(((uintptr_t)g_bf16 & 7u) == 0, US_ERR_SHAPE, "us_adam_step_segments_bf16: g_bf16 must be 8-byte aligned");
    return adam_segments(p, g, m, v, n_seg, seg_off, seg_n, seg_lr, beta1, beta2, eps, 0, step_dev, zero_grad_mask, stream, g_bf16, bf16_mask);
}

extern "C" int us_adam_step_inc(float* step_dev, double beta1, double beta2, void* stream) {
    US_REQUIRE(step_dev && ((uintptr_t)step_dev & 7u) == 0, US_ERR_NULL, "us_adam_step_inc: step_dev is NULL or not 8-byte aligned");
    hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, beta1, beta2);
    US_CHECK_LAUNCH("us_adam_step_inc");
    return US_OK;
}

extern "C" int us_pose_rays(const float* pose, const int64_t* pix, int64_t n, const float* intr_host4, int W0, int H0, int crop_w,
                            const float* depth_img, const float* color_img, int W, float* rays_o, float* rays_d, float* dirs,
                            float* gt_depth, float* gt_color, void* stream) {
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(pose && pix && intr_host4 && depth_img && color_img && rays_o && rays_d && dirs && gt_depth && gt_color, US_ERR_NULL,
               "us_pose_rays: NULL pointer");
    US_REQUIRE(crop_w >= 1 && W >= 1, US_ERR_SHAPE, "us_pose_rays: bad image shape");
    Intr in; in.fx = intr_host4[0]; in.fy = intr_host4[1]; in.cx = intr_host4[2]; in.cy = intr_host4[3]; in.W0 = W0; in.H0 = H0; in.wi = crop_w;
    hipLaunchKernelGGL(k_pose_rays, dim3(grid_1d(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, pose, pix, n, in, depth_img, color_img, W,
                       rays_o, rays_d, dirs, gt_depth, gt_color);
    US_CHECK_LAUNCH("us_pose_rays");
    return US_OK;
}

extern "C" int us_pose_grad(const float* pose, const float* g_rays_o, const float* g_rays_d, const float* dirs, int64_t n, float* g_pose,
                            void* stream) {
    US_REQUIRE(pose && g_rays_o && g_rays_d && dirs && g_pose, US_ERR_NULL, "us_pose_grad: NULL pointer");
    US_REQUIRE(n >= 1, US_ERR_SHAPE, "us_pose_grad: n < 1");
    hipLaunchKernelGGL(k_pose_grad, dim3(1), dim3(1024), 0, (hipStream_t)stream, pose, g_rays_o, g_rays_d, dirs, n, g_pose);
    US_CHECK_LAUNCH("us_pose_grad");
    return US_OK;
}

extern "C" int us_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                                double eps, const float* step_dev, void* stream) {
    US_REQUIRE(p && g && m && v && step_dev, US_ERR_NULL, "us_adam_step_dev: NULL pointer");
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    hipLaunchKernelGGL(k_adam_dev, dim3(grid_1d(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)lr, (float)beta1,
                       (float)beta2, (float)eps, step_dev);
    US_CHECK_LAUNCH("us_adam_step_dev");
    return US_OK;
}

extern "C" int us_track_sample(const float* pose, const int64_t* pix, int64_t n_rays, const float* intr_host4, int W0, int H0, int crop_w,
                               int crop_h, const float* depth_img, const float* color_img, int W, const float* bound_host, const float* t_uni,
                               int n_strat, const float* t_surf, int n_imp, float c_free, float surf_off, float surf_span, const float* t_rand,
                               uint64_t rng_seed, const float* rng_counter, int perturb, float* rays_o, float* rays_d, float* dirs,
                               float* gt_depth, float* gt_color, uint8_t* valid, float* z_vals, float* pts, void* stream) {
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(pose && intr_host4 && depth_img && color_img && bound_host && t_uni && t_surf && dirs && gt_depth && gt_color && valid && z_vals && pts,
               US_ERR_NULL, "us_track_sample: NULL pointer");
    US_REQUIRE((rays_o != nullptr) == (rays_d != nullptr), US_ERR_NULL, "us_track_sample: rays_o and rays_d together or neither");
    const int S = n_strat + n_imp;
    US_REQUIRE(n_strat >= 1 && n_imp >= 0 && S <= 256, US_ERR_SHAPE, "us_track_sample: n_strat %d n_imp %d (S must be <= 256)", n_strat, n_imp);
    US_REQUIRE(crop_w >= 1 && crop_h >= 1 && W >= 1 && (int64_t)crop_w * crop_h < (1ll << 31), US_ERR_SHAPE, "us_track_sample: bad image shape");
    Intr in; in.fx = intr_host4[0]; in.fy = intr_host4[1]; in.cx = intr_host4[2]; in.cy = intr_host4[3]; in.W0 = W0; in.H0 = H0; in.wi = crop_w;
    const int rpb = 256 / S;
    hipLaunchKernelGGL(k_track_sample, dim3((unsigned)us_cdiv(n_rays, rpb)), dim3(256), (size_t)rpb * S * sizeof(float), (hipStream_t)stream, pose,
                       pix, n_rays, in, crop_h, depth_img, color_img, W, make_bound3x(bound_host), t_uni, n_strat, t_surf, n_imp, c_free, surf_off,
                       surf_span, t_rand, (unsigned long long)rng_seed, rng_counter, perturb, rays_o, rays_d, dirs, gt_depth, gt_color, valid,
                       z_vals, pts, rpb);
    US_CHECK_LAUNCH("us_track_sample");
    return US_OK;
}

// the tracking loss (mode US_LOSS_TRK_ORIGINAL: the 10 x median gate) around the compositing in 2 + 1 launches
extern "C" int us_track_loss_fwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples, const uint8_t* valid,
                                 const float* gt_depth, const float* gt_color, double truncation, float* term, float* pixel_unc, float* depth,
                                 float* rgb, float* depth_unc, float* partials, float* err, float* median, float* stats, void* stream) {
    US_REQUIRE(n_rays >= 1 && n_rays <= MEDIAN_MAX, US_ERR_SHAPE, "us_track_loss_fwd: n_rays %lld not in 1..%d", (long long)n_rays, MEDIAN_MAX);
    US_REQUIRE(raw && z_vals && beta && gt_depth && gt_color && term && pixel_unc && depth && rgb && depth_unc && partials && err && median && stats,
               US_ERR_NULL, "us_track_loss_fwd: NULL pointer");
    US_REQUIRE(n_samples >= 1 && n_samples <= 128, US_ERR_SHAPE, "us_track_loss_fwd: n_samples %d not in 1..128", n_samples);
    LossFwd lf{1, US_LOSS_TRK_ORIGINAL, valid, gt_depth, gt_color, (float)truncation, (float)(0.4 * truncation), partials, err};
    dim3 grid((unsigned)us_cdiv(n_rays, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (n_samples <= 64)
        hipLaunchKernelGGL((k_composite_fwd<1>), grid, block, 0, s, raw, z_vals, beta, n_rays, n_samples, term, pixel_unc, depth, rgb, depth_unc, (float*)nullptr, lf);
    else
        hipLaunchKernelGGL((k_composite_fwd<2>), grid, block, 0, s, raw, z_vals, beta, n_rays, n_samples, term, pixel_unc, depth, rgb, depth_unc, (float*)nullptr, lf);
    US_CHECK_LAUNCH("us_track_loss_fwd");
    hipLaunchKernelGGL(k_track_gate_reduce, dim3(1), dim3(1024), 0, s, err, valid, partials, (int)n_rays, median, stats);
    US_CHECK_LAUNCH("us_track_loss_fwd(gate)");
    return US_OK;
}

extern "C" int us_track_loss_bwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples, const uint8_t* valid,
                                 const float* gt_depth, const float* gt_color, const float* depth, const float* rgb, const float* pixel_unc,
                                 const float* median, double truncation, const float* w5_host, const float* stats, float* d_raw, float* loss_out,
                                 void* stream) {
    US_REQUIRE(n_rays >= 1, US_ERR_SHAPE, "us_track_loss_bwd: empty batch");
    US_REQUIRE(raw && z_vals && beta && gt_depth && gt_color && depth && rgb && pixel_unc && median && w5_host && stats && d_raw, US_ERR_NULL,
               "us_track_loss_bwd: NULL pointer");
    US_REQUIRE(n_samples >= 1 && n_samples <= 128, US_ERR_SHAPE, "us_track_loss_bwd: n_samples %d not in 1..128", n_samples);
    LossBwd lb{};
    lb.enabled = 1; lb.mode = US_LOSS_TRK_ORIGINAL; lb.valid = valid; lb.gt_depth = gt_depth; lb.gt_color = gt_color; lb.depth = depth; lb.rgb = rgb;
    lb.unc = pixel_unc; lb.tr = (float)truncation; lb.tr04 = (float)(0.4 * truncation); lb.stats = stats; lb.loss_out = loss_out; lb.median = median;
    for (int k = 0; k < 5; ++k) lb.lw.w[k] = w5_host[k];
    dim3 grid((unsigned)us_cdiv(n_rays, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const float* nul = nullptr; float* nulw = nullptr;
    if (n_samples <= 64)
        hipLaunchKernelGGL((k_composite_bwd<1>), grid, block, 0, s, raw, z_vals, beta, n_rays, n_samples, nul, nul, nul, nul, nul, nul, d_raw, nulw, nulw, lb);
    else
        hipLaunchKernelGGL((k_composite_bwd<2>), grid, block, 0, s, raw, z_vals, beta, n_rays, n_samples, nul, nul, nul, nul, nul, nul, d_raw, nulw, nulw, lb);
    US_CHECK_LAUNCH("us_track_loss_bwd");
    return US_OK;
}

static int window_sample(const char* fn, const float* c2w_first, const float* poses7, int b, int64_t n_per_frame, int n_extra_frames, int64_t n_extra,
                         const float* pool_depth, const float* pool_color, const float* pool_dirs, int64_t pool_size,
                         const int64_t* idx_a, const int64_t* idx_b, const float* bound_host, const float* t_uni, int n_strat,
                         const float* t_surf, int n_imp, float c_free, float surf_off, float surf_span, const float* t_rand,
                         uint64_t rng_seed, const float* rng_counter, int perturb, float* rays_o, float* rays_d, float* dirs,
                         float* gt_depth, float* gt_color, uint8_t* valid, float* z_vals, float* pts, const int32_t* shape_dev,
                         const int32_t* slots, int64_t rows_a, int64_t rows_b, void* stream) {
    int64_t n_rays;
    if (shape_dev) {                                             // fixed row layout; the real shape is read on the device
        US_REQUIRE(rows_a >= 1 && rows_b >= 0 && pool_size >= 1, US_ERR_SHAPE, "%s: bad row layout", fn);
        US_REQUIRE(slots && c2w_first && poses7, US_ERR_NULL, "%s: slots, c2w_first and poses7 are required", fn);
        n_rays = rows_a + rows_b;
        b = 1; n_per_frame = rows_a; n_extra_frames = 0; n_extra = 1;
    } else {
        US_REQUIRE(b >= 1 && n_per_frame >= 0 && n_extra_frames >= 0 && n_extra_frames <= b && n_extra >= 0 && pool_size >= 1, US_ERR_SHAPE,
                   "%s: bad window shape", fn);
        if (n_extra_frames == 0 || n_extra == 0) { n_extra_frames = 0; n_extra = 1; }
        n_rays = (int64_t)b * n_per_frame + (int64_t)n_extra_frames * (n_extra_frames ? n_extra : 0);
        if (n_rays == 0) return US_OK;
        US_REQUIRE((c2w_first || poses7) && ((c2w_first && b == 1) || poses7), US_ERR_NULL, "%s: NULL pointer", fn);
        US_REQUIRE(!n_extra_frames || ((idx_a == nullptr) == (idx_b == nullptr)), US_ERR_NULL, "%s: idx_a and idx_b together, or neither (in-kernel draw)", fn);
    }
    US_REQUIRE(pool_depth && pool_color && pool_dirs && bound_host && t_uni && t_surf && rays_o && rays_d &&
               gt_depth && gt_color && valid && z_vals && pts, US_ERR_NULL, "%s: NULL pointer", fn);
    const int S = n_strat + n_imp;
    US_REQUIRE(n_strat >= 1 && n_imp >= 0 && S <= 256, US_ERR_SHAPE, "%s: n_strat %d n_imp %d (S must be <= 256)", fn, n_strat, n_imp);
    WinShape ws; ws.b = b; ws.n_per = n_per_frame; ws.P = pool_size; ws.xf = n_extra_frames; ws.xn = n_extra;
    const int rpb = 256 / S;
    hipLaunchKernelGGL(k_window_sample, dim3((unsigned)us_cdiv(n_rays, rpb)), dim3(256), (size_t)rpb * S * sizeof(float), (hipStream_t)stream,
                       c2w_first, poses7, ws, pool_depth, pool_color, pool_dirs, idx_a, idx_b, n_rays, make_bound3x(bound_host), t_uni, n_strat,
                       t_surf, n_imp, c_free, surf_off, surf_span, t_rand, (unsigned long long)rng_seed, rng_counter, perturb, rays_o, rays_d, dirs,
                       gt_depth, gt_color, valid, z_vals, pts, rpb, shape_dev, slots, rows_a);
    US_CHECK_LAUNCH(fn);
    return US_OK;
}

extern "C" int us_window_sample(const float* c2w_first, const float* poses7, int b, int64_t n_per_frame, int n_extra_frames, int64_t n_extra,
                                const float* pool_depth, const float* pool_color, const float* pool_dirs, int64_t pool_size,
                                const int64_t* idx_a, const int64_t* idx_b, const float* bound_host, const float* t_uni, int n_strat,
                                const float* t_surf, int n_imp, float c_free, float surf_off, float surf_span, const float* t_rand,
                                uint64_t rng_seed, const float* rng_counter, int perturb, float* rays_o, float* rays_d, float* dirs,
                                float* gt_depth, float* gt_color, uint8_t* valid, float* z_vals, float* pts, void* stream) {
    return window_sample("us_window_sample", c2w_first, poses7, b, n_per_frame, n_extra_frames, n_extra, pool_depth, pool_color, pool_dirs, pool_size,
                         idx_a, idx_b, bound_host, t_uni, n_strat, t_surf, n_imp, c_free, surf_off, surf_span, t_rand, rng_seed, rng_counter, perturb,
                         rays_o, rays_d, dirs, gt_depth, gt_color, valid, z_vals, pts, nullptr, nullptr, 0, 0, stream);
}

extern "C" int us_arena_window_sample(const float* c2w_first, const float* poses7, const int32_t* shape_dev, const int32_t* slots, int64_t rows_a,
                                      int64_t rows_b, const float* pool_depth, const float* pool_color, const float* pool_dirs, int64_t pool_size,
                                      const int64_t* idx_a, const int64_t* idx_b, const float* bound_host, const float* t_uni, int n_strat,
                                      const float* t_surf, int n_imp, float c_free, float surf_off, float surf_span, const float* t_rand,
                                      uint64_t rng_seed, const float* rng_counter, int perturb, float* rays_o, float* rays_d, float* dirs,
                                      float* gt_depth, float* gt_color, uint8_t* valid, float* z_vals, float* pts, void* stream) {
    US_REQUIRE(shape_dev, US_ERR_NULL, "us_arena_window_sample: shape_dev is NULL");
    return window_sample("us_arena_window_sample", c2w_first, poses7, 0, 0, 0, 0, pool_depth, pool_color, pool_dirs, pool_size, idx_a, idx_b, bound_host,
                         t_uni, n_strat, t_surf, n_imp, c_free, surf_off, surf_span, t_rand, rng_seed, rng_counter, perturb, rays_o, rays_d, dirs,
                         gt_depth, gt_color, valid, z_vals, pts, shape_dev, slots, rows_a, rows_b, stream);
}
