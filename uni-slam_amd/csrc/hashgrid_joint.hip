// hashgrid_joint.hip -- the two hash grids of Uni-SLAM (sdf + colour: src/networks/decoders.py:118,143 encode the SAME points with
// two tcnn.Encoding tables built from the same base resolution / per-level scale, src/UNISLAM.py:241-253) in ONE pass per direction.
//
// The grids differ only in log2_hashmap_size.  Per level the cell, the fractional position, the run structure of the samples of a
// ray and the coherent prime hash h of the 8 vertices are the same; the tables differ in how h (or the dense index) is folded:
//   JOINT levels  both dense with the same size (entry identical), or both hashed with hs_A <= hs_B (entry_A = entry_B & (hs_A-1)):
//                 with the same number of bins the two entries fall into the same BIN (bin = (entry/16) mod n_bins needs only bits
//                 both entries share), and the bin-local entry of A is the one of B masked.  One 20-byte record
//                 {local entry of B, dA0, dA1, dB0, dB1} serves both tables: one hash, one run scan bookkeeping, one cursor atomic,
//                 one stage slot, one global slot per record instead of two, and 20 instead of 24 bytes through HBM;
//   SPLIT levels  one grid dense, the other hashed (room0: levels 4-6, sdf 2^16 entries against 97 336 .. 474 552 dense colour
//                 entries): the entries are unrelated, each grid gets its own bins and its own 12-byte records, as in
//                 hashgrid_binned.hip; cells, runs and the scan of the run sums are still shared.
// Pipeline (same idea as hashgrid_binned.hip, "bin once, accumulate in f64"):
//   k_jfwd<gather,count>  encoder for both tables (one thread per point and level, blockIdx.y = level) that also leaves the binning
//                         counts: per workgroup (1024 points) and CORNER HALF (corners 0-3 / 4-7) one row of records-per-bin;
//   k_jcolscan, k_jscan   column scan over the rows, exclusive scan of the bin totals in records and in dwords (records of two sizes);
//   k_jwrite              every workgroup takes its 1024 points through all levels; run-combined corner sums of both grids (32 values
//                         per lane through the DPP scan), then per corner half: cursor atomics, records staged in LDS sorted by bin
//                         with their final address, flat copy-out.  A joint record does not fit a full-level stage (8192 x 24 B), so
//                         a level is emitted in two halves of 4 corners -- which is why the counts are kept per corner half.  The
//                         per-level cursors come from the count rows one level ahead (wave scans + one LDS fix-up), the gradients
//                         of the next level are prefetched and consumed before the level's first store;
//   k_jaccum              one workgroup per bin: f64 accumulators of both tables' slices in LDS (ds_add_f64), the bin's lines of
//                         both gradient tables written once.
// Results are those of us_hashgrid_bwd_binned / us_hashgrid_fwd on each grid (tests/test_gpu_joint.py).
#include "binned_dev.h"
#include <string.h>

#define J_THREADS 1024
#define J_MAX_LEVELS 16
#define J_MAX_BINS 8192                  // bins over all levels
#define J_LVL_BINS 512                   // bins of one level (split level: A bins + B bins)
#define J_STAGE 4096                     // stage entries: 1024 points x 4 corners
#define J_ACC_ENTRIES 2304               // table entries (A + B) per bin: 36 KiB of f64 accumulators
#define J_ACC_THREADS 512
#define J_ACC_UNROLL 4
#define J_TARGET_RECORDS 8192
#define J_WANT_MAX 8

enum { J_HASHED_A = 1, J_HASHED_B = 2, J_PACKABLE = 4, J_SPLIT = 8 };

struct JLevel {
    float    scale;
    uint32_t res, res2;
    uint32_t hsA, hsB;                   // entries of the level in table A / B
    uint32_t offA, offB;                 // first entry of the level in table A / B
    uint32_t flags;
    uint32_t lgA, lgB;                   // log2(bins); joint levels: equal
    uint32_t first;                      // first global bin of the level (split: the A bins, then the B bins)
    uint32_t maskA;                      // joint levels: local entry in A = local entry in B & maskA
};
struct JLevels { JLevel l[J_MAX_LEVELS]; };

__host__ __device__ __forceinline__ uint32_t j_level_bins(const JLevel& q) {
    return (q.flags & J_SPLIT) ? (1u << q.lgA) + (1u << q.lgB) : (1u << q.lgB);
}

static bool level_hashed(uint32_t res, uint32_t hs) {            // as level_geom() in hashgrid_dev.h
    uint32_t stride = 1; bool early = false;
    for (int dim = 0; dim < 3; ++dim) { if (stride <= hs) stride *= res; else early = true; }
    return early || hs < stride;
}

// host: level plan.  Returns the number of bins, or -1 when the pair of grids / the batch is outside this path.
static int make_jlevels(const us_grid_desc* a, const us_grid_desc* b, int64_t n, JLevels* out) {
    if (!a || !b || n <= 0) return -1;
    if (a->n_levels != b->n_levels || a->n_levels < 1 || a->n_levels > J_MAX_LEVELS) return -1;
    if (a->n_features != 2 || b->n_features != 2) return -1;
    if (a->n_params != a->offset[a->n_levels] * 2u || b->n_params != b->offset[b->n_levels] * 2u) return -1;
    uint32_t want = 0;
    while (((int64_t)J_TARGET_RECORDS << want) < n * 8 && want < J_WANT_MAX) ++want;
    if (want < 4) want = 4;
    memset(out, 0, sizeof(*out));
    uint32_t total = 0;
    for (uint32_t l = 0; l < a->n_levels; ++l) {
        if (memcmp(&a->scale[l], &b->scale[l], sizeof(float)) != 0 || a->resolution[l] != b->resolution[l]) return -1;
        JLevel& q = out->l[l];
        q.scale = a->scale[l]; q.res = a->resolution[l]; q.res2 = q.res * q.res;
        q.offA = a->offset[l]; q.offB = b->offset[l];
        q.hsA = a->offset[l + 1] - a->offset[l]; q.hsB = b->offset[l + 1] - b->offset[l];
        const bool hA = level_hashed(q.res, q.hsA), hB = level_hashed(q.res, q.hsB);
        q.flags = (hA ? J_HASHED_A : 0u) | (hB ? J_HASHED_B : 0u) | (q.res <= 1023u ? J_PACKABLE : 0u);
        const uint32_t linesA = (q.hsA + 15u) >> 4, linesB = (q.hsB + 15u) >> 4;
        bool joint = (hA && hB && q.hsA <= q.hsB) || (!hA && !hB && q.hsA == q.hsB);
        if (joint) {
            uint32_t lg = 0;
            while (bin_n_local(q.hsA, 0, lg) + bin_n_local(q.hsB, 0, lg) > J_ACC_ENTRIES) ++lg;   // capacity of the f64 slices
            if (lg < want) lg = want;
            while (lg > 0 && (1u << lg) > linesA) --lg;                                           // bins <= lines of the smaller table
            if (bin_n_local(q.hsA, 0, lg) + bin_n_local(q.hsB, 0, lg) > J_ACC_ENTRIES) joint = false;
            else {
                q.lgA = q.lgB = lg;
                q.maskA = (hA && q.hsA < q.hsB) ? (q.hsA >> lg) - 1u : 0xFFFFFFFFu;
            }
        }
        if (!joint) {
            q.flags |= J_SPLIT;
            uint32_t* lgs[2] = {&q.lgA, &q.lgB};
            const uint32_t hs[2] = {q.hsA, q.hsB}, lines[2] = {linesA, linesB};
            for (int s = 0; s < 2; ++s) {
                uint32_t lg = 0;
                while (bin_n_local(hs[s], 0, lg) > J_ACC_ENTRIES) ++lg;
                if (lg < want) lg = want;
                while (lg > 0 && (1u << lg) > lines[s]) --lg;
                if (bin_n_local(hs[s], 0, lg) > J_ACC_ENTRIES) return -1;
                *lgs[s] = lg;
            }
            q.maskA = 0xFFFFFFFFu;
        }
        q.first = total;
        const uint32_t nlb = j_level_bins(q);
        if (nlb > J_LVL_BINS) return -1;
        total += nlb;
    }
    if (total > J_MAX_BINS) return -1;
    return (int)total;
}

// the 8 entry indices of a cell in one table: hashed -> coherent prime hash & (hs-1); dense -> x + y*res + z*res^2 wrapped once at hs
__device__ __forceinline__ void j_entries(bool hashed, uint32_t hs, uint32_t res, uint32_t res2, const uint32_t cell[3], uint32_t (&e)[8]) {
    if (hashed) {                                                // wave-uniform
        const uint32_t hx[2] = {cell[0], cell[0] + 1u};
        const uint32_t hy0 = cell[1] * 2654435761u, hz0 = cell[2] * 805459861u;
        const uint32_t hy[2] = {hy0, hy0 + 2654435761u}, hz[2] = {hz0, hz0 + 805459861u};
        const uint32_t mask = hs - 1u;
#pragma unroll
        for (int c = 0; c < 8; ++c) e[c] = (hx[c & 1] ^ hy[(c >> 1) & 1] ^ hz[c >> 2]) & mask;
    } else {
        const uint32_t base = cell[0] + __umul24(cell[1], res) + __umul24(cell[2], res2);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t v = base + ((c & 1) ? 1u : 0u) + ((c & 2) ? res : 0u) + ((c & 4) ? res2 : 0u);
            e[c] = min(v, v - hs);                               // v < 2*hs: one conditional subtraction == v % hs
        }
    }
}

__device__ __forceinline__ uint32_t j_run_key(const JLevel& q, const uint32_t cell[3], bool live, int lane) {
    uint32_t key = (cell[0] & 1023u) | ((cell[1] & 1023u) << 10) | ((cell[2] & 1023u) << 20);
    if (!live || !(q.flags & J_PACKABLE)) key = 0xC0000000u | (uint32_t)lane;       // never equals a packed cell, unique per lane
    return key;
}

// ---------------------------------------------------------------------------------------------------------------
// forward (both tables) and / or the counts of the binning: one workgroup = 1024 points x one level
// ---------------------------------------------------------------------------------------------------------------
template <bool GATHER, bool COUNT>
__global__ __launch_bounds__(J_THREADS) void k_jfwd(JLevels lv, uint32_t n_levels, const float* __restrict__ pA, const float* __restrict__ pB,
                                                    const float* __restrict__ x, int64_t n, float* __restrict__ outA, float* __restrict__ outB,
                                                    int clamp, int lm, uint32_t* __restrict__ counts, uint32_t row_stride) {
    __shared__ uint32_t lcnt[2][J_LVL_BINS];
    __shared__ uint32_t done;
    const uint32_t level = blockIdx.y;
    const JLevel q = lv.l[level];
    const uint32_t nlb = j_level_bins(q);
    if (COUNT) {
        for (uint32_t t = threadIdx.x; t < 2 * J_LVL_BINS; t += J_THREADS) (&lcnt[0][0])[t] = 0u;
        if (threadIdx.x == 0) done = 0u;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, l8 = lane & 7;
    const int64_t i = (int64_t)blockIdx.x * J_THREADS + threadIdx.x;
    const bool in = i < n;
    float pos[3]; uint32_t cell[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pos_fract(in ? load_x(x, i, k, clamp) : 0.0f, q.scale, pos[k], cell[k]);
    if (GATHER && in) {
        const uint32_t C = n_levels * 2u;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            LevelGeom g;
            g.scale = q.scale; g.res = q.res; g.res2 = q.res2;
            g.hs = s ? q.hsB : q.hsA; g.hashed = (q.flags & (s ? J_HASHED_B : J_HASHED_A)) != 0u;
            const float2* grid = reinterpret_cast<const float2*>(s ? pB : pA) + (s ? q.offB : q.offA);
            float2 v[8];
            gather_corners<2>(g, grid, cell, v);
            float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = corner_weight(c, pos);
                r0 = fmaf(w, v[c].x, r0); r1 = fmaf(w, v[c].y, r1);
            }
            float* o = (s ? outB : outA) + feat_index(lm, i, n, level, C, 2);
            o[0] = r0; o[1] = r1;
        }
    }
    if (!COUNT) return;
    // ---- the records k_jwrite will emit for these points, per bin and corner half (every point counts: no gradient exists yet)
    const uint32_t key = j_run_key(q, cell, in, lane);
    const uint32_t knext = dpp_u32<DPP_ROW_SHL1>(key);               // whole wave active here
    const bool tail = in & ((l8 == 7) | (knext != key));
    if (tail) {
        uint32_t e[8];
        if (q.flags & J_SPLIT) {
            const uint32_t mA = (1u << q.lgA) - 1u, mB = (1u << q.lgB) - 1u, nbA = 1u << q.lgA;
            j_entries((q.flags & J_HASHED_A) != 0u, q.hsA, q.res, q.res2, cell, e);
#pragma unroll
            for (int c = 0; c < 8; ++c) atomicAdd(&lcnt[c >> 2][(e[c] >> BIN_LINE_LOG2) & mA], 1u);
            j_entries((q.flags & J_HASHED_B) != 0u, q.hsB, q.res, q.res2, cell, e);
#pragma unroll
            for (int c = 0; c < 8; ++c) atomicAdd(&lcnt[c >> 2][nbA + ((e[c] >> BIN_LINE_LOG2) & mB)], 1u);
        } else {
            const uint32_t m = (1u << q.lgB) - 1u;
            j_entries((q.flags & J_HASHED_B) != 0u, q.hsB, q.res, q.res2, cell, e);
#pragma unroll
            for (int c = 0; c < 8; ++c) atomicAdd(&lcnt[c >> 2][(e[c] >> BIN_LINE_LOG2) & m], 1u);
        }
    }
    // No closing barrier: the LAST wave to arrive (LDS ticket, acq_rel at workgroup scope: its reads of lcnt[] happen after every
    // counting wave's atomics) writes the two row segments.
    uint32_t ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(&done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket == J_THREADS / 64 - 1) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            uint32_t* row = counts + (size_t)(2u * blockIdx.x + ph) * row_stride + q.first;
            for (uint32_t t = (uint32_t)lane; t < nlb; t += 64) row[t] = lcnt[ph][t];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// scans
// ---------------------------------------------------------------------------------------------------------------
struct JBin { uint32_t level, side, bl, lg; };                   // side: 0 joint, 1 A only, 2 B only
__device__ __forceinline__ JBin j_bin_of(const JLevels& lv, uint32_t n_levels, uint32_t b) {
    JBin r;
    r.level = 0;
    for (uint32_t l = 1; l < n_levels; ++l) r.level += (lv.l[l].first <= b) ? 1u : 0u;
    const JLevel& q = lv.l[r.level];
    const uint32_t rel = b - q.first;
    if (q.flags & J_SPLIT) {
        const uint32_t nbA = 1u << q.lgA;
        r.side = rel < nbA ? 1u : 2u;
        r.bl = rel < nbA ? rel : rel - nbA;
        r.lg = rel < nbA ? q.lgA : q.lgB;
    } else {
        r.side = 0u; r.bl = rel; r.lg = q.lgB;
    }
    return r;
}

__device__ __forceinline__ void j_clear_bin(const JLevels& lv, const JBin jb, float* gradA, float* gradB, uint32_t tid, uint32_t nthreads) {
    const JLevel& q = lv.l[jb.level];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if ((s == 0 && jb.side == 2u) || (s == 1 && jb.side == 1u)) continue;
        const uint32_t hs = s ? q.hsB : q.hsA, lg = s ? q.lgB : q.lgA;
        float* gl = (s ? gradB : gradA) + (size_t)(s ? q.offB : q.offA) * 2u;
        const uint32_t n_local = bin_n_local(hs, jb.bl, lg);
        for (uint32_t loc = tid; loc < n_local; loc += nthreads) {
            const uint32_t e = entry_of(loc, jb.bl, lg);
            if (e < hs) *reinterpret_cast<float2*>(gl + (size_t)e * 2u) = make_float2(0.0f, 0.0f);
        }
    }
}

// column scan over the count rows: prefix[r][b] = sum of counts[r'][b] over the rows ordered before r; totals[b] = column sum.
// Rows of one XCD class (row % 8) are neighbours inside a bin.  8 lanes per bin.  In OVERWRITE mode the entries of bins that will be
// split over several accumulate workgroups (added with float atomics) are cleared here, two kernels ahead of the first add.
#define JCS_THREADS 64
#define JCS_BINS (JCS_THREADS / 8)
__global__ __launch_bounds__(JCS_THREADS) void k_jcolscan(JLevels lv, uint32_t n_levels, const uint32_t* __restrict__ counts,
                                                          uint32_t* __restrict__ prefix, uint32_t n_rows, uint32_t row_stride, uint32_t TB,
                                                          uint32_t* __restrict__ totals, float* __restrict__ gradA, float* __restrict__ gradB,
                                                          int overwrite) {
    const uint32_t xcd = threadIdx.x & 7u;
    const uint32_t b = blockIdx.x * JCS_BINS + (threadIdx.x >> 3);
    const bool ok = b < TB;
    uint32_t sum = 0;
    if (ok) {
        uint32_t w = xcd;
        for (; w + 56 < n_rows; w += 64) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = counts[(size_t)(w + 8 * k) * row_stride + b];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; w < n_rows; w += 8) sum += counts[(size_t)w * row_stride + b];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 8); if ((int)xcd >= o) incl += t; }
    const uint32_t total = __shfl(incl, 7, 8);
    if (ok) {
        uint32_t run = incl - sum;
        uint32_t w = xcd;
        for (; w + 56 < n_rows; w += 64) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = counts[(size_t)(w + 8 * k) * row_stride + b];
#pragma unroll
            for (int k = 0; k < 8; ++k) { prefix[(size_t)(w + 8 * k) * row_stride + b] = run; run += v[k]; }
        }
        for (; w < n_rows; w += 8) { const uint32_t v = counts[(size_t)w * row_stride + b]; prefix[(size_t)w * row_stride + b] = run; run += v; }
        if (xcd == 0) totals[b] = total;
    }
    if (!overwrite) return;
    unsigned long long hot = __ballot(ok && xcd == 0 && total > ACC_CHUNK);
    while (hot) {                                                // wave-uniform loop (one wave per workgroup)
        const int src = __ffsll((long long)hot) - 1;
        hot &= hot - 1ull;
        const uint32_t hb = blockIdx.x * JCS_BINS + ((uint32_t)src >> 3);
        j_clear_bin(lv, j_bin_of(lv, n_levels, hb), gradA, gradB, threadIdx.x, JCS_THREADS);
    }
}

// exclusive scans of totals[0..TB): rec_off (records) and dw_off (dwords: joint records are 5 dwords, single-grid records 3), and the
// list of extra chunks of hot bins (as k_bin_scan).  8 elements per thread.
__global__ __launch_bounds__(1024) void k_jscan(JLevels lv, uint32_t n_levels, const uint32_t* __restrict__ totals, uint32_t TB,
                                                uint32_t* __restrict__ rec_off, uint32_t* __restrict__ dw_off,
                                                uint32_t* __restrict__ extra, uint32_t* __restrict__ hdr) {
    __shared__ uint32_t sh[1024], sd[1024], sx[1024];
    const uint32_t t = threadIdx.x;
    constexpr int E = J_MAX_BINS / 1024;
    uint32_t c[E], dwn[E], s = 0, d = 0, xs = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t b = E * t + k;
        c[k] = b < TB ? totals[b] : 0u;
        dwn[k] = b < TB ? (j_bin_of(lv, n_levels, b).side == 0u ? 5u : 3u) : 0u;
        s += c[k]; d += c[k] * dwn[k]; xs += c[k] > ACC_CHUNK ? (c[k] - 1u) / ACC_CHUNK : 0u;
    }
    sh[t] = s; sd[t] = d; sx[t] = xs;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t a = (t >= o) ? sh[t - o] : 0u, bq = (t >= o) ? sd[t - o] : 0u, cq = (t >= o) ? sx[t - o] : 0u;
        __syncthreads();
        sh[t] += a; sd[t] += bq; sx[t] += cq;
        __syncthreads();
    }
    uint32_t chunk = ACC_CHUNK;
    const uint32_t x_all = sx[1023];
    if (x_all > ACC_EXTRA_MAX) {                                 // wave-uniform, rare: coarser chunks, scanned again
        chunk = ACC_CHUNK * ((x_all + ACC_EXTRA_MAX - 1u) / ACC_EXTRA_MAX);
        xs = 0;
#pragma unroll
        for (int k = 0; k < E; ++k) xs += c[k] > chunk ? (c[k] - 1u) / chunk : 0u;
        __syncthreads();
        sx[t] = xs;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            const uint32_t w = (t >= o) ? sx[t - o] : 0u;
            __syncthreads();
            sx[t] += w;
            __syncthreads();
        }
    }
    uint32_t run = sh[t] - s, drun = sd[t] - d, xrun = sx[t] - xs;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t b = E * t + k;
        if (b < TB) { rec_off[b] = run; dw_off[b] = drun; }
        run += c[k]; drun += c[k] * dwn[k];
        const uint32_t nx = c[k] > chunk ? (c[k] - 1u) / chunk : 0u;
        for (uint32_t j = 0; j < nx; ++j) extra[xrun + j] = b | ((j + 1u) << 16);
        xrun += nx;
    }
    if (t == 1023) { rec_off[TB] = sh[1023]; dw_off[TB] = sd[1023]; hdr[0] = sx[1023]; hdr[1] = chunk; }
}

// ---------------------------------------------------------------------------------------------------------------
// record pass
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(J_THREADS) void k_jwrite(JLevels lv, uint32_t n_levels, const float* __restrict__ x,
                                                      const float* __restrict__ dyA, const float* __restrict__ dyB, int64_t n, int clamp,
                                                      const uint32_t* __restrict__ counts, const uint32_t* __restrict__ prefix,
                                                      const uint32_t* __restrict__ dw_off, uint32_t row_stride,
                                                      uint32_t* __restrict__ rec, uint32_t rec_cap_dw) {
    __shared__ uint4 st4[J_STAGE];                               // joint: {loc, dA0, dA1, dB0}; single grid: {loc, d0, d1, address}
    __shared__ uint2 st2[J_STAGE];                               // joint: {dB1, address}
    __shared__ uint32_t cur[2][2][J_LVL_BINS];                   // [level parity][corner half][bin of the level]: stage cursor
    __shared__ uint32_t gdl[2][2][J_LVL_BINS];                   // record address (dwords) = cursor * DW + gdl
    __shared__ uint32_t wtot[2][J_THREADS / 64];
    __shared__ uint32_t atot[2][2], ttot[2][2];                  // [parity][corner half]: records of the A bins (split) / of all bins
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, l8 = lane & 7;
    const uint32_t wave = tid >> 6;
    const uint32_t sph = tid >> 9, sj = tid & 511u;              // role in the cursor set-up: (corner half, bin of the level)
    const size_t srow = (size_t)(2u * blockIdx.x + sph) * row_stride;
    const int64_t i = (int64_t)blockIdx.x * J_THREADS + tid;
    const bool in = i < n;
    float xv[3] = {0.f, 0.f, 0.f};
    if (in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) xv[k] = load_x(x, i, k, clamp);
    }
    // level-major gradient planes [L][N][2]
    auto load_dy = [&](uint32_t level, float (&d)[4]) {
        d[0] = d[1] = d[2] = d[3] = 0.0f;
        if (in) {
            const float2 a = *reinterpret_cast<const float2*>(dyA + ((int64_t)level * n + i) * 2);
            const float2 b = *reinterpret_cast<const float2*>(dyB + ((int64_t)level * n + i) * 2);
            d[0] = a.x; d[1] = a.y; d[2] = b.x; d[3] = b.y;
        }
    };
    // cursor set-up of a level in three steps (each thread: one bin of one corner half): loads; wave scan of the counts; fix-up over
    // the waves.  Step 3 needs step 2's wave totals of all waves: a barrier lies between them.
    auto setup_load = [&](uint32_t level, uint32_t& c, uint32_t& p, uint32_t& o) {
        const uint32_t nlb = j_level_bins(lv.l[level]), b = lv.l[level].first + sj;
        c = 0u; p = 0u; o = 0u;
        if (sj < nlb) { c = counts[srow + b]; p = prefix[srow + b]; o = dw_off[b]; }
    };
    auto setup_scan = [&](int par, uint32_t c, uint32_t& incl) {
        incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wtot[par][wave] = incl;
    };
    auto setup_fin = [&](int par, uint32_t level, uint32_t c, uint32_t p, uint32_t o, uint32_t incl) {
        const JLevel& q = lv.l[level];
        const uint32_t nlb = j_level_bins(q), split = q.flags & J_SPLIT;
        uint32_t before = 0;
        for (uint32_t w = 8u * sph; w < wave; ++w) before += wtot[par][w];
        const uint32_t excl = before + incl - c, dw = split ? 3u : 5u;
        if (sj < nlb) { cur[par][sph][sj] = excl; gdl[par][sph][sj] = o + (p - excl) * dw; }
        if (split && sj == (1u << q.lgA)) atot[par][sph] = excl;
        if (sj == nlb - 1u) ttot[par][sph] = excl + c;
    };

    uint32_t c1, p1, o1, incl1;
    float dn[4];
    setup_load(0, c1, p1, o1);
    load_dy(0, dn);
    setup_scan(0, c1, incl1);
    __syncthreads();
    setup_fin(0, 0, c1, p1, o1, incl1);

    for (uint32_t level = 0; level < n_levels; ++level) {
        const int par = level & 1;
        const JLevel q = lv.l[level];
        const bool split = (q.flags & J_SPLIT) != 0u, has_next = level + 1 < n_levels;
        const float dy[4] = {dn[0], dn[1], dn[2], dn[3]};
        // ---- cell, position, runs
        float pos[3]; uint32_t cell[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pos_fract(xv[k], q.scale, pos[k], cell[k]);
        const uint32_t key = j_run_key(q, cell, in, lane);
        const uint32_t kprev = dpp_u32<DPP_ROW_SHR(1)>(key), knext = dpp_u32<DPP_ROW_SHL1>(key);
        bool flag = (l8 == 0) | (kprev != key);                                  // head of a run
        const bool tail = in & ((l8 == 7) | (knext != key));
        // ---- entries: e = table B's (joint levels: also gives A's); eA = table A's on split levels
        uint32_t e[8], eA[8];
        j_entries((q.flags & J_HASHED_B) != 0u, q.hsB, q.res, q.res2, cell, e);
        if (split) j_entries((q.flags & J_HASHED_A) != 0u, q.hsA, q.res, q.res2, cell, eA);
        else {
#pragma unroll
            for (int c = 0; c < 8; ++c) eA[c] = 0u;
        }
        // ---- corner products of both grids and their segmented scan over the run
        float val[8][4];
        {
            const float a0[2] = {1.0f - pos[0], pos[0]}, a1[2] = {1.0f - pos[1], pos[1]}, a2[2] = {1.0f - pos[2], pos[2]};
            float wxy[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) wxy[c] = a0[c & 1] * a1[c >> 1];              // tcnn's order: ((1*a0)*a1)*a2
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float w = wxy[c & 3] * a2[c >> 2];
#pragma unroll
                for (int f = 0; f < 4; ++f) val[c][f] = w * dy[f];
            }
        }
#define J_SCAN_STEP(O)                                                                                               \
        if (__ballot(!flag && (l8 >= (O))) != 0ull) {            /* wave-uniform: nothing left to merge -> skip the step */ \
            const float takef = (!flag && (l8 >= (O))) ? 1.0f : 0.0f;                                                \
            const bool fprev = dpp_u32<DPP_ROW_SHR(O)>(flag ? 1u : 0u) != 0u;                                        \
            _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                           \
                _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                       \
                    val[c][f] = fmaf(dpp_f32<DPP_ROW_SHR(O)>(val[c][f]), takef, val[c][f]);                          \
            flag = flag | ((l8 >= (O)) & fprev);                                                                     \
        }
        J_SCAN_STEP(1)
        J_SCAN_STEP(2)
        J_SCAN_STEP(4)
#undef J_SCAN_STEP
        // ---- requests for the next level: its cursor inputs and gradients.  They are consumed below, BEFORE this level's first
        //      record store: a wait for a load also waits for every store issued before it (one vmcnt counter).
        if (has_next) { setup_load(level + 1, c1, p1, o1); load_dy(level + 1, dn); }

        auto consume_next = [&]() {
            if (has_next) {
                asm volatile("" : "+v"(dn[0]), "+v"(dn[1]), "+v"(dn[2]), "+v"(dn[3]), "+v"(c1), "+v"(p1), "+v"(o1));
                setup_scan(par ^ 1, c1, incl1);
            }
        };
        auto copy_out5 = [&](uint32_t cnt) {
            for (uint32_t k = tid; k < cnt; k += J_THREADS) {
                const uint4 a = st4[k]; const uint2 b = st2[k];
                if (b.y + 5u <= rec_cap_dw) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
                    u32x4 w; w.x = a.x; w.y = a.y; w.z = a.z; w.w = a.w;
                    *reinterpret_cast<u32x4*>(rec + b.y) = w;
                    rec[b.y + 4u] = b.x;
                }
            }
        };
        auto copy_out3 = [&](uint32_t cnt) {
            for (uint32_t k = tid; k < cnt; k += J_THREADS) {
                const uint4 a = st4[k];
                if (a.w + 3u <= rec_cap_dw) {
                    typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
                    u32x3 w; w.x = a.x; w.y = a.y; w.z = a.z;
                    *reinterpret_cast<u32x3*>(rec + a.w) = w;
                }
            }
        };

        lds_barrier();                                           // the previous level's copy-out has left the stage; cursors are in place
        if (!split) {
            const uint32_t nbm = (1u << q.lgB) - 1u, lg = q.lgB;
#define J_EMIT_JOINT(PH)                                                                                             \
            if (tail) {                                                                                              \
                uint32_t k4[4], g4[4];                                                                               \
                _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                                  \
                    const uint32_t b = (e[4 * (PH) + cc] >> BIN_LINE_LOG2) & nbm;                                    \
                    k4[cc] = atomicAdd(&cur[par][PH][b], 1u);                                                        \
                    g4[cc] = gdl[par][PH][b];                                                                        \
                }                                                                                                    \
                _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                                  \
                    const int c = 4 * (PH) + cc;                                                                     \
                    if (k4[cc] < J_STAGE) {                                                                          \
                        st4[k4[cc]] = make_uint4(local_of(e[c], lg), __float_as_uint(val[c][0]), __float_as_uint(val[c][1]), __float_as_uint(val[c][2])); \
                        st2[k4[cc]] = make_uint2(__float_as_uint(val[c][3]), g4[cc] + k4[cc] * 5u);                  \
                    }                                                                                                \
                }                                                                                                    \
            }
            J_EMIT_JOINT(0)
            consume_next();
            lds_barrier();
            copy_out5(min(ttot[par][0], (uint32_t)J_STAGE));
            if (has_next) setup_fin(par ^ 1, level + 1, c1, p1, o1, incl1);
            lds_barrier();
            J_EMIT_JOINT(1)
            lds_barrier();
            copy_out5(min(ttot[par][1], (uint32_t)J_STAGE));
#undef J_EMIT_JOINT
        } else {
            const uint32_t nbA = 1u << q.lgA;
            // SIDE 0: table A (bins [0, nbA) of the level, stage index = cursor); SIDE 1: table B (bins behind, cursor - atot)
#define J_EMIT_SINGLE(PH, SIDE)                                                                                      \
            if (tail) {                                                                                              \
                const uint32_t lg = (SIDE) ? q.lgB : q.lgA, nbm = (1u << lg) - 1u, sb = (SIDE) ? atot[par][PH] : 0u;  \
                uint32_t k4[4], g4[4];                                                                               \
                _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                                  \
                    const uint32_t ee = (SIDE) ? e[4 * (PH) + cc] : eA[4 * (PH) + cc];                               \
                    const uint32_t b = ((SIDE) ? nbA : 0u) + ((ee >> BIN_LINE_LOG2) & nbm);                          \
                    k4[cc] = atomicAdd(&cur[par][PH][b], 1u);                                                        \
                    g4[cc] = gdl[par][PH][b];                                                                        \
                }                                                                                                    \
                _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                                  \
                    const int c = 4 * (PH) + cc;                                                                     \
                    const uint32_t ee = (SIDE) ? e[c] : eA[c];                                                       \
                    const uint32_t si = k4[cc] - sb;                                                                 \
                    if (si < J_STAGE)                                                                                \
                        st4[si] = make_uint4(local_of(ee, lg), __float_as_uint(val[c][(SIDE) ? 2 : 0]), __float_as_uint(val[c][(SIDE) ? 3 : 1]), g4[cc] + k4[cc] * 3u); \
                }                                                                                                    \
            }
            J_EMIT_SINGLE(0, 0)
            consume_next();
            lds_barrier();
            copy_out3(min(atot[par][0], (uint32_t)J_STAGE));
            if (has_next) setup_fin(par ^ 1, level + 1, c1, p1, o1, incl1);
            lds_barrier();
            J_EMIT_SINGLE(0, 1)
            lds_barrier();
            copy_out3(min(ttot[par][0] - atot[par][0], (uint32_t)J_STAGE));
            lds_barrier();
            J_EMIT_SINGLE(1, 0)
            lds_barrier();
            copy_out3(min(atot[par][1], (uint32_t)J_STAGE));
            lds_barrier();
            J_EMIT_SINGLE(1, 1)
            lds_barrier();
            copy_out3(min(ttot[par][1] - atot[par][1], (uint32_t)J_STAGE));
#undef J_EMIT_SINGLE
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// accumulate: one workgroup per bin (+ one per extra chunk of a hot bin; those come first in the grid)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(J_ACC_THREADS) void k_jaccum(JLevels lv, uint32_t n_levels, uint32_t e_max,
                                                          const uint32_t* __restrict__ rec_off, const uint32_t* __restrict__ dw_off,
                                                          const uint32_t* __restrict__ extra, const uint32_t* __restrict__ hdr,
                                                          const uint32_t* __restrict__ rec, float* __restrict__ gradA,
                                                          float* __restrict__ gradB, int overwrite) {
    __shared__ double acc[2 * J_ACC_ENTRIES];                    // [B comp 0][B comp 1][A comp 0][A comp 1], by component (bank spread)
    uint32_t b, chunk = 0;
    const uint32_t CH = hdr[1];                                  // records per workgroup (k_jscan)
    if (blockIdx.x < e_max) {
        if (blockIdx.x >= hdr[0]) return;
        const uint32_t pk = extra[blockIdx.x];
        b = pk & 0xFFFFu; chunk = pk >> 16;
    } else {
        b = blockIdx.x - e_max;
    }
    const JBin jb = j_bin_of(lv, n_levels, b);
    const JLevel q = lv.l[jb.level];
    const uint32_t nA = jb.side != 2u ? bin_n_local(q.hsA, jb.bl, q.lgA) : 0u;
    const uint32_t nB = jb.side != 1u ? bin_n_local(q.hsB, jb.bl, q.lgB) : 0u;
    double* accB = acc;
    double* accA = acc + 2u * nB;
    const uint32_t b0 = rec_off[b], b1 = rec_off[b + 1];
    const bool hot = (b1 - b0) > CH;                             // several workgroups add into this bin's entries
    auto sweep = [&](bool zero_only) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t nl = s ? nB : nA, hs = s ? q.hsB : q.hsA, lg = s ? q.lgB : q.lgA;
            const double* a = s ? accB : accA;
            float* gl = (s ? gradB : gradA) + (size_t)(s ? q.offB : q.offA) * 2u;
            for (uint32_t loc = threadIdx.x; loc < nl; loc += J_ACC_THREADS) {
                const uint32_t e = entry_of(loc, jb.bl, lg);
                if (e >= hs) continue;
                float* p = gl + (size_t)e * 2u;
                if (zero_only) { *reinterpret_cast<float2*>(p) = make_float2(0.0f, 0.0f); continue; }
                const float v0 = (float)a[loc], v1 = (float)a[nl + loc];
                if (hot) {
                    if (v0 != 0.0f) atomicAdd(p, v0);
                    if (v1 != 0.0f) atomicAdd(p + 1, v1);
                } else if (overwrite) {
                    *reinterpret_cast<float2*>(p) = make_float2(v0, v1);
                } else if (v0 != 0.0f || v1 != 0.0f) {               // this workgroup is the only writer of its entries
                    float2 o = *reinterpret_cast<const float2*>(p);
                    o.x += v0; o.y += v1;
                    *reinterpret_cast<float2*>(p) = o;
                }
            }
        }
    };
    if (b0 == b1) {                                              // nothing landed in this bin (wave-uniform)
        if (overwrite) sweep(true);
        return;
    }
    const uint32_t r0 = b0 + chunk * CH;
    const uint32_t r1 = (b1 - r0 > CH) ? r0 + CH : b1;
    const uint32_t r_last = r1 - 1u, base_dw = dw_off[b];
    constexpr uint32_t STEP = J_ACC_THREADS * J_ACC_UNROLL;
    auto clear = [&]() { for (uint32_t k = threadIdx.x; k < 2u * (nA + nB); k += J_ACC_THREADS) acc[k] = 0.0; };
    // Software-pipelined record stream, two register buffers, every load unconditional (past the end it re-reads the last record):
    // a load behind a branch of its own is waited for at the end of that branch (see hashgrid_binned.hip).
    if (jb.side == 0u) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
        u32x4 w4[2][J_ACC_UNROLL]; uint32_t w1[2][J_ACC_UNROLL];
        auto fetch = [&](int buf, uint32_t base) {
#pragma unroll
            for (int u = 0; u < J_ACC_UNROLL; ++u) {
                const uint32_t r = min(base + u * J_ACC_THREADS + threadIdx.x, r_last);
                const uint32_t* src = rec + (size_t)base_dw + (size_t)(r - b0) * 5u;
                w4[buf][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
                w1[buf][u] = __builtin_nontemporal_load(src + 4);
            }
        };
        auto add = [&](int buf, uint32_t base) {
#pragma unroll
            for (int u = 0; u < J_ACC_UNROLL; ++u) {
                if (base + u * J_ACC_THREADS + threadIdx.x <= r_last) {
                    const uint32_t lb = w4[buf][u].x, la = lb & q.maskA;
                    if (lb < nB) {
                        atomicAdd(&accB[lb], (double)__uint_as_float(w4[buf][u].w));          // ds_add_f64
                        atomicAdd(&accB[nB + lb], (double)__uint_as_float(w1[buf][u]));
                        atomicAdd(&accA[la], (double)__uint_as_float(w4[buf][u].y));
                        atomicAdd(&accA[nA + la], (double)__uint_as_float(w4[buf][u].z));
                    }
                }
            }
        };
        fetch(0, r0);                                            // in flight while the accumulators are cleared
        clear();
        __syncthreads();
        for (uint32_t base = r0;;) {
            fetch(1, base + STEP); add(0, base); base += STEP;
            if (base >= r1) break;
            fetch(0, base + STEP); add(1, base); base += STEP;
            if (base >= r1) break;
        }
    } else {
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
        u32x3 w3[2][J_ACC_UNROLL];
        double* a = jb.side == 1u ? accA : accB;
        const uint32_t nl = jb.side == 1u ? nA : nB;
        auto fetch = [&](int buf, uint32_t base) {
#pragma unroll
            for (int u = 0; u < J_ACC_UNROLL; ++u) {
                const uint32_t r = min(base + u * J_ACC_THREADS + threadIdx.x, r_last);
                w3[buf][u] = __builtin_nontemporal_load(reinterpret_cast<const u32x3*>(rec + (size_t)base_dw + (size_t)(r - b0) * 3u));
            }
        };
        auto add = [&](int buf, uint32_t base) {
#pragma unroll
            for (int u = 0; u < J_ACC_UNROLL; ++u) {
                if (base + u * J_ACC_THREADS + threadIdx.x <= r_last) {
                    const uint32_t lc = w3[buf][u].x;
                    if (lc < nl) {
                        atomicAdd(&a[lc], (double)__uint_as_float(w3[buf][u].y));
                        atomicAdd(&a[nl + lc], (double)__uint_as_float(w3[buf][u].z));
                    }
                }
            }
        };
        fetch(0, r0);                                            // in flight while the accumulators are cleared
        clear();
        __syncthreads();
        for (uint32_t base = r0;;) {
            fetch(1, base + STEP); add(0, base); base += STEP;
            if (base >= r1) break;
            fetch(0, base + STEP); add(1, base); base += STEP;
            if (base >= r1) break;
        }
    }
    __syncthreads();
    sweep(false);
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
// workspace: totals | rec_off | dw_off (3 x (J_MAX_BINS + 64) u32) | hdr (16 u32) | extra (ACC_EXTRA_MAX u32) | count rows
// [2 * workgroups][row_stride] | their column prefixes | records
static uint32_t j_row_stride(int TB) { return ((uint32_t)TB + 63u) & ~63u; }
static size_t j_header_bytes(int TB, int64_t n) {
    const size_t rows = 2 * (size_t)us_cdiv(n, J_THREADS);
    return (size_t)(3 * (J_MAX_BINS + 64) + 16 + ACC_EXTRA_MAX) * sizeof(uint32_t) + 2 * rows * j_row_stride(TB) * sizeof(uint32_t);
}
static uint64_t j_record_dwords(const JLevels& lv, uint32_t n_levels, int64_t n) {
    uint64_t per_point = 0;
    for (uint32_t l = 0; l < n_levels; ++l) per_point += (lv.l[l].flags & J_SPLIT) ? 8u * 6u : 8u * 5u;
    return per_point * (uint64_t)n;
}

extern "C" int us_hashgrid_joint_supported(const us_grid_desc* a, const us_grid_desc* b, int64_t n) {
    JLevels lv;
    const int TB = make_jlevels(a, b, n, &lv);
    if (TB <= 0) return 0;
    if (j_record_dwords(lv, a->n_levels, n) > 0xFFFFFFF0ull) return 0;      // 32-bit record addresses (in dwords)
    if ((uint64_t)n * 8ull * a->n_levels >= 0xFFFFFFFFull) return 0;         // 32-bit record counts
    if (2 * us_cdiv(n, J_THREADS) >= 0x7FFFFFFF) return 0;
    return 1;
}

extern "C" size_t us_hashgrid_joint_workspace_bytes(const us_grid_desc* a, const us_grid_desc* b, int64_t n) {
    JLevels lv;
    const int TB = make_jlevels(a, b, n, &lv);
    if (TB <= 0) return 0;
    return j_header_bytes(TB, n) + (size_t)j_record_dwords(lv, a->n_levels, n) * sizeof(uint32_t);
}

struct JWorkspace { uint32_t *totals, *rec_off, *dw_off, *hdr, *extra, *counts, *prefix, *rec; uint32_t n_wg, n_rows, stride, rec_cap_dw; };
static JWorkspace j_carve(void* workspace, const JLevels& lv, uint32_t n_levels, int TB, int64_t n) {
    JWorkspace w;
    w.totals = (uint32_t*)workspace;
    w.rec_off = w.totals + (J_MAX_BINS + 64);
    w.dw_off = w.rec_off + (J_MAX_BINS + 64);
    w.hdr = w.dw_off + (J_MAX_BINS + 64);
    w.extra = w.hdr + 16;
    w.counts = w.extra + ACC_EXTRA_MAX;
    w.n_wg = (uint32_t)us_cdiv(n, J_THREADS); w.n_rows = 2 * w.n_wg; w.stride = j_row_stride(TB);
    w.prefix = w.counts + (size_t)w.n_rows * w.stride;
    w.rec = (uint32_t*)((char*)workspace + j_header_bytes(TB, n));
    w.rec_cap_dw = (uint32_t)j_record_dwords(lv, n_levels, n);
    return w;
}

#define J_CHECK_PAIR(name)                                                                                                   \
    US_REQUIRE(a && b, US_ERR_NULL, name ": desc is NULL");                                                                  \
    JLevels lv;                                                                                                              \
    const int TB = make_jlevels(a, b, n > 0 ? n : 1, &lv);                                                                   \
    US_REQUIRE(TB > 0 && us_hashgrid_joint_supported(a, b, n > 0 ? n : 1), US_ERR_CONFIG,                                    \
               name ": the two grids do not share a geometry this path takes (F = 2, <= %d levels, equal base resolution and " \
               "per-level scale, <= %d bins) or the batch is too large", J_MAX_LEVELS, J_MAX_BINS)

extern "C" int us_hashgrid_fwd_joint(const us_grid_desc* a, const us_grid_desc* b, const float* paramsA, const float* paramsB,
                                     const float* x, int64_t n, float* outA, float* outB, int flags, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (n < 0) return US_ERR_SHAPE;
    J_CHECK_PAIR("us_hashgrid_fwd_joint");
    if (n == 0) return US_OK;
    US_REQUIRE(paramsA && paramsB && x && outA && outB, US_ERR_NULL, "us_hashgrid_fwd_joint: NULL pointer");
    US_REQUIRE(((uintptr_t)paramsA & 15u) == 0 && ((uintptr_t)paramsB & 15u) == 0, US_ERR_SHAPE, "us_hashgrid_fwd_joint: params must be 16-byte aligned");
    const int clamp = flags & US_GRID_CLAMP01, lm = (flags & US_GRID_LEVEL_MAJOR) ? 1 : 0;
    dim3 grid((unsigned)us_cdiv(n, J_THREADS), a->n_levels), block(J_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (workspace) {
        US_REQUIRE(((uintptr_t)workspace & 15u) == 0, US_ERR_SHAPE, "us_hashgrid_fwd_joint: workspace must be 16-byte aligned");
        US_REQUIRE(workspace_bytes >= us_hashgrid_joint_workspace_bytes(a, b, n), US_ERR_WORKSPACE,
                   "us_hashgrid_fwd_joint: workspace %zu B < %zu B", workspace_bytes, us_hashgrid_joint_workspace_bytes(a, b, n));
        const JWorkspace w = j_carve(workspace, lv, a->n_levels, TB, n);
        hipLaunchKernelGGL((k_jfwd<true, true>), grid, block, 0, s, lv, a->n_levels, paramsA, paramsB, x, n, outA, outB, clamp, lm, w.counts, w.stride);
    } else {
        hipLaunchKernelGGL((k_jfwd<true, false>), grid, block, 0, s, lv, a->n_levels, paramsA, paramsB, x, n, outA, outB, clamp, lm, (uint32_t*)nullptr, 0u);
    }
    US_CHECK_LAUNCH("us_hashgrid_fwd_joint");
    return US_OK;
}

extern "C" int us_hashgrid_bwd_joint(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA,
                                     const float* dL_dyB, int64_t n, float* gradA, float* gradB, int flags, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    if (n < 0) return US_ERR_SHAPE;
    J_CHECK_PAIR("us_hashgrid_bwd_joint");
    hipStream_t s = (hipStream_t)stream;
    const int overwrite = (flags & US_GRID_BWD_OVERWRITE) ? 1 : 0, counted = (flags & US_GRID_BWD_COUNTED) ? 1 : 0;
    if (n == 0) {                                                // no samples: the gradients are zero
        if (overwrite) {
            US_REQUIRE(gradA && gradB, US_ERR_NULL, "us_hashgrid_bwd_joint: NULL pointer");
            hipError_t e = hipMemsetAsync(gradA, 0, (size_t)a->n_params * sizeof(float), s);
            if (e == hipSuccess) e = hipMemsetAsync(gradB, 0, (size_t)b->n_params * sizeof(float), s);
            if (e != hipSuccess) { us_set_error("us_hashgrid_bwd_joint: memset: %s", hipGetErrorString(e)); return (int)e; }
        }
        return US_OK;
    }
    US_REQUIRE(x && dL_dyA && dL_dyB && gradA && gradB && workspace, US_ERR_NULL, "us_hashgrid_bwd_joint: NULL pointer");
    US_REQUIRE(flags & US_GRID_LEVEL_MAJOR, US_ERR_CONFIG, "us_hashgrid_bwd_joint: the gradients must be level-major planes (US_GRID_LEVEL_MAJOR)");
    US_REQUIRE(((uintptr_t)gradA & 15u) == 0 && ((uintptr_t)gradB & 15u) == 0 && ((uintptr_t)workspace & 15u) == 0 &&
               ((uintptr_t)dL_dyA & 7u) == 0 && ((uintptr_t)dL_dyB & 7u) == 0, US_ERR_SHAPE,
               "us_hashgrid_bwd_joint: gradient tables and workspace must be 16-byte aligned, dL_dy 8-byte aligned");
    US_REQUIRE(workspace_bytes >= us_hashgrid_joint_workspace_bytes(a, b, n), US_ERR_WORKSPACE,
               "us_hashgrid_bwd_joint: workspace %zu B < %zu B", workspace_bytes, us_hashgrid_joint_workspace_bytes(a, b, n));
    const JWorkspace w = j_carve(workspace, lv, a->n_levels, TB, n);
    const int clamp = flags & US_GRID_CLAMP01;
    const uint32_t L = a->n_levels;
    if (!counted)
        hipLaunchKernelGGL((k_jfwd<false, true>), dim3(w.n_wg, L), dim3(J_THREADS), 0, s, lv, L, (const float*)nullptr, (const float*)nullptr, x, n,
                           (float*)nullptr, (float*)nullptr, clamp, 1, w.counts, w.stride);
    hipLaunchKernelGGL(k_jcolscan, dim3((unsigned)us_cdiv(TB, JCS_BINS)), dim3(JCS_THREADS), 0, s, lv, L, w.counts, w.prefix, w.n_rows, w.stride,
                       (uint32_t)TB, w.totals, gradA, gradB, overwrite);
    hipLaunchKernelGGL(k_jscan, dim3(1), dim3(1024), 0, s, lv, L, w.totals, (uint32_t)TB, w.rec_off, w.dw_off, w.extra, w.hdr);
    hipLaunchKernelGGL(k_jwrite, dim3(w.n_wg), dim3(J_THREADS), 0, s, lv, L, x, dL_dyA, dL_dyB, n, clamp, w.counts, w.prefix, w.dw_off, w.stride,
                       w.rec, w.rec_cap_dw);
    hipLaunchKernelGGL(k_jaccum, dim3(ACC_EXTRA_MAX + TB), dim3(J_ACC_THREADS), 0, s, lv, L, (uint32_t)ACC_EXTRA_MAX, w.rec_off, w.dw_off, w.extra,
                       w.hdr, w.rec, gradA, gradB, overwrite);
    US_CHECK_LAUNCH("us_hashgrid_bwd_joint");
    return US_OK;
}
