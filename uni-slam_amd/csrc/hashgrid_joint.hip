// hashgrid_joint.hip -- the two hash grids of Uni-SLAM (sdf + colour: src/networks/decoders.py:118,143 encode the SAME points with
// two tcnn.Encoding tables built from the same base resolution / per-level scale, src/UNISLAM.py:241-253) in ONE pass per direction.
//
// The grids differ only in log2_hashmap_size.  Per level the cell, the fractional position, the vertex runs of a ray's samples
// (binned_dev.h), the trilinear weights and the coherent prime hash of the 8 vertices are the same; what differs is how the hash
// (or the dense index) is folded into the table:
//   JOINT levels  both dense with the same size (entries identical), or both hashed with hs_A <= hs_B (entry_A = entry_B & (hs_A-1)):
//                 with the same number of bins a vertex falls into the same BIN of both tables (bin = (entry / 16) mod n_bins uses
//                 only bits both entries share) and its bin-local entry in A is the one in B masked.  One count, one cursor atomic
//                 and one stage position per record pair; the bin's record region holds its A records, then its B records.
//   SPLIT levels  one grid dense, the other hashed (room0: levels 4-6): unrelated entries, each grid has its own bins.
// Everything per (point, level) that does not depend on the table -- cell, keys, run masks, weights, the run scan's bookkeeping -- is
// computed once for both grids; records are 10 bytes (16-bit local entry, d0, d1) in two planes per (bin, grid) region (below).
//   k_jfwd<gather,count>  encoder for both tables (one thread per point and level, blockIdx.y = level) that also leaves the binning
//                         counts, one row of records-per-bin per 512 points;
//   k_jcolscan, k_jscan   column scan over the rows, exclusive scan of the bin totals;
//   k_jwrite              every 512-thread workgroup takes 512 points through all levels (two workgroups per CU: one computes while
//                         the other moves records): 4 products per slot through one DPP scan, then per grid: records staged in LDS
//                         sorted by bin with their final address, flat copy-out.  The per-level cursors come from the count rows one
//                         level ahead (wave scans + one LDS fix-up) instead of whole-kernel counter arrays, the gradients of the next
//                         level are prefetched and consumed before the level's first store;
//   k_jaccum_p            persistent workgroups, each one record pipeline over its (bin, grid) items: f64 accumulators in LDS
//                         (ds_add_f64), the bin's lines of the gradient table written once.
// Results are those of us_hashgrid_fwd / us_hashgrid_bwd_binned on each grid (tests/test_gpu_joint.py).
#include "binned_dev.h"
#include <string.h>

#ifndef J_FWD_THREADS
#define J_FWD_THREADS 1024               // encoder workgroup: 1024 points x one level (two count rows)
#endif
#ifndef J_PLAN_SLOPE
#define J_PLAN_SLOPE 0.29                 // make_jplan: estimated work of a workgroup of level l = 1 + slope * max(0, l - 3)
#endif
#ifndef J_FWD_FINE_FIRST
#define J_FWD_FINE_FIRST 0               // two-dimensional launch with blockIdx.y = 0 the FINEST level: render-only call 168 -> 159 us, but the dy/dx
                                         // encoder +4 us and BASELINE configs[2] +3 %: off (the XCD plan below has the order built in)
#endif
#ifndef J_FWD_XCD
#define J_FWD_XCD 1                     // the counting encoder's workgroups placed by XCD (make_jplan); 0: blockIdx.y = level
#endif
#ifndef J_FWD_WAVES_PER_EU
#define J_FWD_WAVES_PER_EU 4               // register budget of the encoder: 4 waves per SIMD = ONE 1024-thread workgroup per CU (it takes 70 registers);
                                         // 8 = two per CU (64 registers, no spill in the counting form): encoder 132 -> 140 us, same box (r4) -- the
                                         // gathers are served at the memory system's rate, more waves only spread each level's slab thinner
#endif
#ifndef J_DYDX_NT
#define J_DYDX_NT 1                      // dy/dx is written once and read once: non-temporal stores.  Measured (MI355X, 4096 x 64): with the
#endif                                   // planes [L][3][N][2] a store instruction covers whole 128-byte lines and the encoder takes 169 us (plain
                                         // stores: 177); as [L][N][3][2] -- 8 bytes of every 24 per instruction -- 198 plain and 261 non-temporal
#ifndef J_SMALL_BATCH
#define J_SMALL_BATCH (1 << 18)          // below this many points the count-free encoder runs 256-thread workgroups
#endif
#define J_ROW_POINTS 512                 // points per count row = per k_jwrite workgroup
#define J_MAX_LEVELS 16
#define J_MAX_BINS 8192                  // bins over all levels
#define J_LVL_BINS 512                   // bins of one level (split level: A bins + B bins)
#define J_STAGE (J_ROW_POINTS * 8)       // stage entries: every slot of every point of one level
#define J_ACC_DOUBLES 4096               // 32 KiB of f64 accumulators per (bin, grid)
#ifndef J_ACC_THREADS
#define J_ACC_THREADS 512
#endif
#ifndef J_ACC_UNROLL
#define J_ACC_UNROLL 2                   // pairs of records per thread and batch (r3, 8192 records per bin: 2 / 3 / 4: 86.6 / 85.8 / 87.9 us;
                                         // r4, 16384 per bin and 2048 workgroups: 2 against 3: 219 against 222 us for the whole table gradient)
#endif
#ifndef J_TARGET_RECORDS
#define J_TARGET_RECORDS 16384           // records per bin aimed at where capacity leaves a choice (the coarse / split levels).  Same box, both grids,
#endif                                   // scans included (us): 8192: 233.7-238.6, 16384: 222.2-227.4, 32768: as 16384; 4096: 233.6-234.3
#define J_WANT_MAX 8

enum { J_HASHED_A = 1, J_HASHED_B = 2, J_PACKABLE = 4, J_SPLIT = 8 };

// RECORDS: 10 bytes each, in TWO PLANES over one record index R -- the local entries as 16-bit numbers (< 2^11: J_ACC_DOUBLES / 2 entries
// per bin) in E[R], the value pairs in V[R].  Every (bin, grid) region starts at a multiple of 8 records, so the accumulate pass reads a
// PAIR of records as one aligned 4-byte load (two entries) + one aligned 16-byte load (four values), consecutive lanes consecutive pairs,
// and the record pass derives both addresses of a record from ONE per-bin number.  The accumulate pass is bound by its record stream
// (timing builds, MI355X, one contiguous aligned load per record: 8 / 12 / 16 bytes per record 79.9 / 103.1 / 121.3 us): 12-byte
// {entry, d0, d1} records 102.5 us, two planes 88.7 (same box); the same 10 bytes as 5-dword units of two records -- one unaligned 16-byte
// load + a 4-byte load over the same lines -- took 104: the shape of the loads counts, not only their bytes.
// A region of c records takes j_region(c) record indices; a joint bin holds its A region, then its B region -- and only the B region's
// entries are written: table A's local entry is table B's masked, so the A items of the accumulate pass read E of the B region (the
// 2-byte stores are the expensive part of the record pass's copy-out: without the A pass's, 126 -> 114 us and 132 -> 118 on two boxes;
// pairing the remaining ones into 4-byte stores costs more than it saves, with row shifts as well as with two positions per lane).
__host__ __device__ __forceinline__ uint32_t j_region(uint32_t c) { return (c + 7u) & ~7u; }
static_assert(J_ACC_DOUBLES / 2 <= 65536 && J_LVL_BINS <= 65536, "local entry and bin of the level in 16 bits each");

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
    const uint32_t cap = J_ACC_DOUBLES / 2;                      // entries of one grid per bin
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
            while (bin_n_local(q.hsB, 0, lg) > cap) ++lg;                                         // capacity of the larger f64 slice
            if (lg < want) lg = want;
            while (lg > 0 && (1u << lg) > linesA) --lg;                                           // bins <= lines of the smaller table
            if (bin_n_local(q.hsB, 0, lg) > cap) joint = false;
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
                while (bin_n_local(hs[s], 0, lg) > cap) ++lg;
                if (lg < want) lg = want;
                while (lg > 0 && (1u << lg) > lines[s]) --lg;
                if (bin_n_local(hs[s], 0, lg) > cap) return -1;
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

// ---------------------------------------------------------------------------------------------------------------
// forward (both tables) and / or the counts of the binning: one workgroup = 1024 points x one level
// ---------------------------------------------------------------------------------------------------------------
// DYDX: also leave d(features)/d(position) of both grids, planes [L][3][N][2] (= tcnn's dy_dx, which the input gradient
// contracts with dL/dy: us_hashgrid_dydx_rays) -- the 8 vertices are in registers here, and 24 contiguous bytes per thread and grid
// stream out coalesced, where a second gather pass over the tables (us_hashgrid_bwd_input_rays) costs as much as the encoder itself.
#define J_PLAN_SEGS 6
struct JPlan { uint32_t n; uint16_t level[8][J_PLAN_SEGS], first[8][J_PLAN_SEGS], count[8][J_PLAN_SEGS]; };
// host: who takes which (level, block) units.  Units in the order finest level first; XCD x takes a run from the front (heavy) and a run
// from the back (light) of what is left, 1/8 of the units in all, the split chosen so that its estimated work is 1/8 of the total.
// Work per unit of level l: 1 on the first four levels, + 0.29 per level beyond (the encoder's workgroup clocks: 4.4 us on levels 0-3,
// 19.5 us on level 15 -- the finer the level, the fewer lines the lanes of a wave share).
// the placement pays where a level's slab is a large part of an L2 (4 MiB): a table of 2^18 entries and more
static bool j_big_slab(const us_grid_desc* a, const us_grid_desc* b) {
    uint32_t m = 0;
    for (uint32_t l = 0; l < a->n_levels; ++l) { const uint32_t ha = a->offset[l + 1] - a->offset[l], hb = b->offset[l + 1] - b->offset[l]; m = ha > m ? ha : m; m = hb > m ? hb : m; }
    return m >= (1u << 18);
}
static bool make_jplan_uncached(uint32_t n_levels, uint32_t nb, double slope, JPlan* out);
static bool make_jplan(uint32_t n_levels, uint32_t nb, double slope, JPlan* out) {      // (the last plans are kept: an iteration asks for the same ones every time)
    struct Slot { uint32_t levels = 0, nb = 0; double slope = 0.0; bool ok = false; JPlan plan; };
    static thread_local Slot slots[4];
    static thread_local int next = 0;
    for (const Slot& c : slots) if (c.levels == n_levels && c.nb == nb && c.slope == slope) { *out = c.plan; return c.ok; }
    Slot& c = slots[next]; next = (next + 1) & 3;
    c.ok = make_jplan_uncached(n_levels, nb, slope, &c.plan); c.levels = n_levels; c.nb = nb; c.slope = slope;
    *out = c.plan;
    return c.ok;
}
static bool make_jplan_uncached(uint32_t n_levels, uint32_t nb, double slope, JPlan* out) {
    memset(out, 0, sizeof(*out));
    if (n_levels < 8u || nb == 0u || nb > 65535u) return false;
    const uint32_t total = n_levels * nb, per = (total + 7u) / 8u;
    auto w = [&](uint32_t l) { return 1.0 + slope * (l > 3u ? (double)(l - 3u) : 0.0); };
    double wsum = 0.0;
    for (uint32_t l = 0; l < n_levels; ++l) wsum += w(l) * nb;
    // front: unit index f counts from the finest level's block 0; back: from level 0's block 0 upwards
    uint32_t fl = n_levels - 1u, fo = 0u, bl = 0u, bo = 0u, left = total;      // (level, offset) of the next front / back unit; units left
    for (uint32_t x = 0; x < 8u; ++x) {
        const uint32_t cnt = left < per ? left : per;
        const double target = wsum / 8.0;
        // take nf units from the front so that the estimate comes closest to the target: the estimate grows with nf (a front unit is at
        // least as heavy as the back unit it displaces), so the crossing is found by bisection
        auto estimate = [&](uint32_t nf) {
            double acc = 0.0; uint32_t l = fl, o = fo, k = nf;
            while (k) { const uint32_t t = (nb - o) < k ? (nb - o) : k; acc += t * w(l); k -= t; o += t; if (o == nb) { o = 0; if (l == 0u) break; --l; } }
            l = bl; o = bo; k = cnt - nf;
            while (k) { const uint32_t t = (nb - o) < k ? (nb - o) : k; acc += t * w(l); k -= t; o += t; if (o == nb) { o = 0; ++l; } }
            return acc;
        };
        uint32_t lo_n = 0, hi_n = cnt;                             // estimate(lo_n) <= target < estimate(hi_n), where such a crossing exists
        while (hi_n - lo_n > 1u) { const uint32_t mid = (lo_n + hi_n) / 2u; if (estimate(mid) <= target) lo_n = mid; else hi_n = mid; }
        const double e_lo = estimate(lo_n), e_hi = estimate(hi_n);
        const uint32_t best_nf = ((e_lo > target ? e_lo - target : target - e_lo) <= (e_hi > target ? e_hi - target : target - e_hi)) ? lo_n : hi_n;
        int q = 0;
        uint32_t k = best_nf;
        while (k) {
            const uint32_t t = (nb - fo) < k ? (nb - fo) : k;
            if (q >= J_PLAN_SEGS) return false;
            out->level[x][q] = (uint16_t)fl; out->first[x][q] = (uint16_t)fo; out->count[x][q] = (uint16_t)t; ++q;
            k -= t; fo += t; if (fo == nb) { fo = 0; if (fl > 0u) --fl; }
        }
        k = cnt - best_nf;
        while (k) {
            const uint32_t t = (nb - bo) < k ? (nb - bo) : k;
            if (q >= J_PLAN_SEGS) return false;
            out->level[x][q] = (uint16_t)bl; out->first[x][q] = (uint16_t)(nb - bo - t); out->count[x][q] = (uint16_t)t; ++q;     // (from the level's end: where
                                                                                                                      //  the two runs meet in one level they do not overlap)
            k -= t; bo += t; if (bo == nb) { bo = 0; ++bl; }
        }
        left -= cnt;
    }
    out->n = per;                                                // workgroups per XCD: the launch has 8 * per
    return true;
}
template <bool GATHER, bool COUNT, bool DYDX = false>
__global__ __launch_bounds__(J_FWD_THREADS, DYDX ? 4 : J_FWD_WAVES_PER_EU) void k_jfwd(JLevels lv, JPlan plan, uint32_t n_levels, const float* __restrict__ pA, const float* __restrict__ pB,
                                                        const float* __restrict__ x, int64_t n, float* __restrict__ outA, float* __restrict__ outB,
                                                        int clamp, int lm, uint32_t* __restrict__ counts, uint32_t row_stride, uint32_t n_rows,
                                                        us_half_t* __restrict__ dydxA = nullptr, us_half_t* __restrict__ dydxB = nullptr
#ifdef J_FWD_TIMING
                                                        , unsigned long long* __restrict__ dbg = nullptr      // timing build (tools/fwd_clocks.py)
#endif
                                                        ) {
#ifdef J_FWD_TIMING
    const unsigned long long dbg_t0 = wall_clock64();
    unsigned long long dbg_t1 = 0;
#endif
    constexpr int HALVES = J_FWD_THREADS / J_ROW_POINTS;
    __shared__ uint32_t lcnt[HALVES][J_LVL_BINS];
    __shared__ uint32_t done;
    // XCD-AWARE PLACEMENT (plan.n != 0: a one-dimensional launch).  Workgroup i runs on XCD i & 7 (round-robin placement; checked with
    // HW_REG_XCC_ID in the timing build), and an XCD's workgroups take the plan's segments (level, first block, blocks) in order: the finest
    // levels whole, the coarse ones cut where the estimated work of the eight XCDs evens out -- a level's slab is then pulled into ONE L2
    // (two for a cut level) instead of all eight.
    uint32_t level = J_FWD_FINE_FIRST ? n_levels - 1u - blockIdx.y : blockIdx.y, bx_ = blockIdx.x;
    if (plan.n != 0u) {
        const uint32_t xcd = blockIdx.x & 7u;
        uint32_t j = blockIdx.x >> 3;
        bool found = false;
#pragma unroll
        for (int q = 0; q < J_PLAN_SEGS; ++q) {
            const uint32_t c = plan.count[xcd][q];
            if (!found && j < c) { found = true; level = plan.level[xcd][q]; bx_ = plan.first[xcd][q] + j; }
            else if (!found) j -= c;
        }
        if (!found) return;                                      // (padding of the launch to a multiple of eight)
    }
#define J_BX bx_
    const JLevel q = lv.l[level];
    const uint32_t nlb = j_level_bins(q);
    if (COUNT) {
        for (uint32_t t = threadIdx.x; t < HALVES * J_LVL_BINS; t += J_FWD_THREADS) (&lcnt[0][0])[t] = 0u;
        if (threadIdx.x == 0) done = 0u;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, lg16 = lane & (RUN_GROUP - 1);
    const int64_t i = (int64_t)J_BX * (COUNT ? J_FWD_THREADS : (int)blockDim.x) + threadIdx.x;   // (the counting side needs its 1024-point rows)
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
            float* o = (s ? outB : outA) + feat_index(lm & 1, i, n, level, C, 2);
            if (lm & 2) {
                // US_GRID_FEAT_SPLIT_BF16: the pair as the split-bf16 decoders' operands -- {hi(f0) | hi(f1) << 16, lo(f0) | lo(f1) << 16} with
                // hi = bf16(f), lo = bf16(f - hi), value for value what mlp_bf16.inc's bf_pack_split forms from the float pair: the same 8
                // bytes, and the decoders (forward and backward) load their B operands instead of spending ~45 VALU instructions per 32
                // points on the split.  This kernel waits on its gathers (14 % of its wave cycles issue anything): the conversions are free here.
                const __bf16 h0 = (__bf16)r0, h1 = (__bf16)r1;
                const __bf16 l0 = (__bf16)(r0 - (float)h0), l1 = (__bf16)(r1 - (float)h1);
                uint2 wv;
                wv.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
                wv.y = (uint32_t)__builtin_bit_cast(unsigned short, l0) | ((uint32_t)__builtin_bit_cast(unsigned short, l1) << 16);
                *reinterpret_cast<uint2*>(o) = wv;
            } else {
                o[0] = r0; o[1] = r1;
            }
            if (DYDX) {                                          // k_fwd<F, DYDX>'s arithmetic, value for value
                // planes [L][3][N][2] of IEEE half values: a store instruction covers whole lines.  (dy/dx = scale * feature differences: |.| < 65504
                // for any table this path trains, relative error 2^-11 -- against 5e-4 relative per element the pose gradient sums ~10^6 of them)
                _Float16* dd_base = reinterpret_cast<_Float16*>(s ? dydxB : dydxA) + (int64_t)level * 3 * n * 2 + i * 2;
#pragma unroll
                for (int gd = 0; gd < 3; ++gd) {
                    float a0 = 0.0f, a1 = 0.0f;
                    const int d0 = gd == 0 ? 1 : 0, d1 = gd == 2 ? 1 : 2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float w = q.scale;
                        w *= (e & 1) ? pos[d0] : 1.0f - pos[d0];
                        w *= (e & 2) ? pos[d1] : 1.0f - pos[d1];
                        const int cl = ((e & 1) << d0) | (((e >> 1) & 1) << d1);
                        const int cr = cl | (1 << gd);
                        a0 += w * (v[cr].x - v[cl].x); a1 += w * (v[cr].y - v[cl].y);
                    }
                    const float xin = x[i * 3 + gd];
                    const bool pass = !clamp || (xin >= 0.0f && xin <= 1.0f);
                    typedef _Float16 f2_t __attribute__((ext_vector_type(2)));
                    f2_t o2; o2.x = (_Float16)(pass ? a0 : 0.0f); o2.y = (_Float16)(pass ? a1 : 0.0f);
                    f2_t* dst = reinterpret_cast<f2_t*>(dd_base + (int64_t)gd * n * 2);
#if J_DYDX_NT
                    __builtin_nontemporal_store(o2, dst);
#else
                    *dst = o2;
#endif
                }
            }
        }
    }
#ifdef J_FWD_TIMING
    dbg_t1 = wall_clock64();                                     // (the features of this thread are on their way out)
#endif
    if (!COUNT) return;
    // ---- the records k_jwrite will emit for these points, per bin (every point counts: no gradient exists yet)
    uint32_t key[8];
    slot_keys(cell, in && (q.flags & J_PACKABLE), lane, key);
    const uint2 ht = slot_run_masks(key, lg16);                      // outside any condition on `in`: its row shifts read the neighbour lanes
    const uint32_t tail = in ? ht.y : 0u;
    const uint32_t half = threadIdx.x / J_ROW_POINTS;
    if (tail != 0u) {
        uint32_t e[8];
        slot_entries((q.flags & J_HASHED_B) != 0u, q.hsB, q.res, q.res2, cell, e);
        const uint32_t mB = (1u << q.lgB) - 1u, offB = (q.flags & J_SPLIT) ? (1u << q.lgA) : 0u;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if ((tail >> c) & 1u) atomicAdd(&lcnt[half][offB + ((e[c] >> BIN_LINE_LOG2) & mB)], 1u);
        if (q.flags & J_SPLIT) {
            const uint32_t mA = (1u << q.lgA) - 1u;
            slot_entries((q.flags & J_HASHED_A) != 0u, q.hsA, q.res, q.res2, cell, e);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if ((tail >> c) & 1u) atomicAdd(&lcnt[half][(e[c] >> BIN_LINE_LOG2) & mA], 1u);
        }
    }
    // No closing barrier: the LAST wave to arrive (LDS ticket, acq_rel at workgroup scope: its reads of lcnt[] happen after every
    // counting wave's atomics) writes the row segments.
    uint32_t ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(&done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket == J_FWD_THREADS / 64 - 1) {
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            const uint32_t r = HALVES * J_BX + h;
            if (r >= n_rows) break;
            uint32_t* row = counts + (size_t)r * row_stride + q.first;
            for (uint32_t t = (uint32_t)lane; t < nlb; t += 64) row[t] = lcnt[h][t];
        }
#ifdef J_FWD_TIMING
        if (dbg && lane == 0) {
            unsigned long long* d = dbg + 4 * ((size_t)level * n_rows / HALVES + J_BX);
            d[0] = dbg_t0; d[1] = dbg_t1; d[2] = wall_clock64();
            unsigned int xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned int hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            d[3] = ((unsigned long long)xcc << 32) | hw;
        }
#endif
    }
}

// ---------------------------------------------------------------------------------------------------------------
// scans
// ---------------------------------------------------------------------------------------------------------------
struct JBin { uint32_t level, kind, bl; };                       // kind: 0 joint (both grids), 1 grid A only, 2 grid B only
// (every level is visited with a compile-time index: the level table is a kernel argument, and indexing it with a run-time value
//  costs dependent scalar loads -- in k_jscan 128 of them per thread)
__device__ __forceinline__ JBin j_bin_of(const JLevels& lv, uint32_t n_levels, uint32_t b) {
    JBin r;
    r.level = 0; r.kind = 0; r.bl = b;
#pragma unroll
    for (uint32_t l = 0; l < J_MAX_LEVELS; ++l) {
        const JLevel& q = lv.l[l];
        if (l < n_levels && q.first <= b) {                      // the last level that starts at or before b
            const uint32_t rel = b - q.first, nbA = 1u << q.lgA;
            const bool split = (q.flags & J_SPLIT) != 0u;
            r.level = l;
            r.kind = split ? (rel < nbA ? 1u : 2u) : 0u;
            r.bl = (split && rel >= nbA) ? rel - nbA : rel;
        }
    }
    return r;
}
// the level's constants for a WAVE-UNIFORM run-time level: one scalar load from the kernel-argument segment
__device__ __forceinline__ const JLevel& j_level(const JLevels& lv, uint32_t level) {
    return lv.l[__builtin_amdgcn_readfirstlane(level)];
}

__device__ __forceinline__ void j_clear_bin(const JLevels& lv, const JBin jb, float* gradA, float* gradB, uint32_t tid, uint32_t nthreads) {
    const JLevel& q = j_level(lv, jb.level);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if ((s == 0 && jb.kind == 2u) || (s == 1 && jb.kind == 1u)) continue;
        const uint32_t hs = s ? q.hsB : q.hsA, lg = s ? q.lgB : q.lgA;
        float* gl = (s ? gradB : gradA) + (size_t)(s ? q.offB : q.offA) * 2u;
        const uint32_t n_local = bin_n_local(hs, jb.bl, lg);
        for (uint32_t loc = tid; loc < n_local; loc += nthreads) {
            const uint32_t e = entry_of(loc, jb.bl, lg);
            if (e < hs) *reinterpret_cast<float2*>(gl + (size_t)e * 2u) = make_float2(0.0f, 0.0f);
        }
    }
}

// column scan over the count rows: prefix[r][b] = sum of counts[r'][b] over the rows r' < r; totals[b] = column sum.
// One 1024-thread workgroup per tile of 16 bins: a wave reads 4 rows x 16 bins (four 64-byte segments) per load, wave w owns a
// contiguous block of rows and keeps its counts in registers; the blocks' sums are scanned through LDS, then every lane turns its
// counts into prefixes (scan over the 4 rows of a load with two shuffles, running sum over the loads) and stores them the way it
// loaded them.  One read and one write of the matrix.  In OVERWRITE mode the entries of bins that will be split over several
// accumulate workgroups (added with float atomics) are cleared here, two kernels ahead of the first add.
#define JCS_THREADS 1024
#define JCS_BINS 16
#define JCS_ITERS 8                      // loads per lane: a workgroup covers 16 waves x 8 loads x 4 rows = 512 rows per pass
__global__ __launch_bounds__(JCS_THREADS) void k_jcolscan(JLevels lv, uint32_t n_levels, const uint32_t* __restrict__ counts,
                                                          uint32_t* __restrict__ prefix, uint32_t n_rows, uint32_t row_stride, uint32_t TB,
                                                          uint32_t* __restrict__ totals, float* __restrict__ gradA, float* __restrict__ gradB,
                                                          int overwrite, uint32_t chunk0) {
    __shared__ uint32_t part[JCS_THREADS / 64][JCS_BINS];
    __shared__ uint32_t carry[JCS_BINS];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, rsub = lane >> 4, bcol = lane & 15u;
    const uint32_t b = blockIdx.x * JCS_BINS + bcol;
    const bool okb = b < TB;
    if (threadIdx.x < JCS_BINS) carry[threadIdx.x] = 0u;
    constexpr uint32_t PASS_ROWS = (JCS_THREADS / 64) * JCS_ITERS * 4;
    const uint32_t n_rows8 = (n_rows + 7u) / 8u, n_order = n_rows8 * 8u;
    for (uint32_t row0 = 0; row0 < n_order; row0 += PASS_ROWS) {     // one pass for up to 512 rows (262 144 points)
        // position o in the bin's ORDER <-> row: the rows of one XCD class (row % 8: workgroup w of k_jwrite runs on XCD w % 8) are
        // neighbours inside a bin, so the ~200-byte segments written through one XCD's L2 share their lines with that XCD only
        // (row order instead: k_jwrite 131 -> 180 us)
        const uint32_t obase = row0 + wave * (JCS_ITERS * 4) + rsub;
        auto row_of = [&](uint32_t o) { return (o % n_rows8) * 8u + (o / n_rows8); };
        uint32_t v[JCS_ITERS], sum = 0;
#pragma unroll
        for (int k = 0; k < JCS_ITERS; ++k) {
            const uint32_t o = obase + 4u * k, r = row_of(o);
            v[k] = (okb && o < n_order && r < n_rows) ? counts[(size_t)r * row_stride + b] : 0u;
        }
#pragma unroll
        for (int k = 0; k < JCS_ITERS; ++k) sum += v[k];
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);         // over the 4 rows of a load: the wave's block sum per bin
        __syncthreads();                                                        // carry[] of the previous pass is in place / consumed
        if (rsub == 0) part[wave][bcol] = sum;
        __syncthreads();
        uint32_t before = carry[bcol];
        for (uint32_t w = 0; w < wave; ++w) before += part[w][bcol];
        // prefixes: rows of one load are r, r+1, r+2, r+3 in the lanes rsub = 0..3
        uint32_t run = before;
#pragma unroll
        for (int k = 0; k < JCS_ITERS; ++k) {
            const uint32_t up1 = __shfl_up(v[k], 16, 64), up2 = __shfl_up(v[k], 32, 64), up3 = __shfl_up(v[k], 48, 64);
            const uint32_t excl = (rsub >= 1 ? up1 : 0u) + (rsub >= 2 ? up2 : 0u) + (rsub >= 3 ? up3 : 0u);
            const uint32_t o = obase + 4u * k, r = row_of(o);
            if (okb && o < n_order && r < n_rows) prefix[(size_t)r * row_stride + b] = run + excl;
            const uint32_t all4 = __shfl(excl + v[k], (int)(48u + bcol), 64);      // the 4 rows' sum sits in the lane rsub = 3
            run += all4;
        }
        __syncthreads();
        if (wave == JCS_THREADS / 64 - 1 && rsub == 0) carry[bcol] = run;        // the last wave's running sum = everything so far
    }
    __syncthreads();
    if (threadIdx.x < JCS_BINS && blockIdx.x * JCS_BINS + threadIdx.x < TB) totals[blockIdx.x * JCS_BINS + threadIdx.x] = carry[threadIdx.x];
    if (!overwrite) return;
    for (uint32_t k = 0; k < JCS_BINS; ++k) {                                    // workgroup-uniform
        const uint32_t hb = blockIdx.x * JCS_BINS + k;
        if (hb < TB && carry[k] > chunk0) j_clear_bin(lv, j_bin_of(lv, n_levels, hb), gradA, gradB, threadIdx.x, JCS_THREADS);
    }
}

// exclusive scans of totals[0..TB): rec_off (records per grid) and dw_off (first record index of the bin in the two planes: a joint bin
// holds its A region, then its B region, j_region(count) indices each; a single-grid bin one), and the list of extra chunks of hot bins.  8 elements per
// thread; wave scans + one fix-up over the 16 wave totals (two barriers).
struct JSingle { uint32_t lo[J_MAX_LEVELS], len[J_MAX_LEVELS]; };     // bin ranges of the split levels (bins that serve one grid only)
__global__ __launch_bounds__(1024) void k_jscan(JSingle sg, const uint32_t* __restrict__ totals, uint32_t TB,
                                                uint32_t* __restrict__ rec_off, uint32_t* __restrict__ dw_off,
                                                uint32_t* __restrict__ extra, uint32_t* __restrict__ hdr, uint32_t chunk0) {
    // chunk0 = ACC_CHUNK, or 0xFFFFFFFF (US_GRID_BWD_DETERMINISTIC): no bin is split, every sum is formed by one workgroup in f64
    __shared__ uint32_t ws[3][16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    constexpr int E = J_MAX_BINS / 1024;
    uint32_t c[E], dwn[E], s = 0, d = 0, xs = 0;                 // dwn: record indices of the bin's region(s)
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t b = E * t + k;
        c[k] = b < TB ? totals[b] : 0u;
        bool single = false;
#pragma unroll
        for (int l = 0; l < J_MAX_LEVELS; ++l) single |= (b - sg.lo[l]) < sg.len[l];
        dwn[k] = b < TB ? (single ? 1u : 2u) * j_region(c[k]) : 0u;
        s += c[k]; d += dwn[k]; xs += c[k] > chunk0 ? (c[k] - 1u) / chunk0 : 0u;
    }
    auto block_scan = [&](uint32_t v, int slot, uint32_t& total) {    // inclusive scan over the 1024 threads
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t q = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += q; }
        if (lane == 63) ws[slot][wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t q = ws[slot][w]; before += (uint32_t)w < wave ? q : 0u; all += q; }
        total = all;
        return incl + before;
    };
    uint32_t s_all, d_all, x_all;
    const uint32_t s_in = block_scan(s, 0, s_all), d_in = block_scan(d, 1, d_all);
    uint32_t x_in = block_scan(xs, 2, x_all);
    uint32_t chunk = chunk0;
    if (x_all > ACC_EXTRA_MAX) {                                 // wave-uniform, rare: coarser chunks, scanned again
        chunk = chunk0 * ((x_all + ACC_EXTRA_MAX - 1u) / ACC_EXTRA_MAX);
        xs = 0;
#pragma unroll
        for (int k = 0; k < E; ++k) xs += c[k] > chunk ? (c[k] - 1u) / chunk : 0u;
        __syncthreads();
        x_in = block_scan(xs, 2, x_all);
    }
    uint32_t run = s_in - s, drun = d_in - d, xrun = x_in - xs;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const uint32_t b = E * t + k;
        if (b < TB) { rec_off[b] = run; dw_off[b] = drun; }
        run += c[k]; drun += dwn[k];
        const uint32_t nx = c[k] > chunk ? (c[k] - 1u) / chunk : 0u;
        for (uint32_t j = 0; j < nx; ++j) extra[xrun + j] = b | ((j + 1u) << 16);
        xrun += nx;
    }
    if (t == 1023) { rec_off[TB] = s_all; dw_off[TB] = d_all; hdr[0] = x_all; hdr[1] = chunk; }
}

template <int BUF> struct JBufTag { static constexpr int value = BUF; };

// ---------------------------------------------------------------------------------------------------------------
// record pass: one workgroup = 512 points through all levels, two workgroups per CU
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(J_ROW_POINTS, 4) void k_jwrite(JLevels lv, uint32_t n_levels, const float* __restrict__ x,
                                                            const float* __restrict__ dyA, const float* __restrict__ dyB, int64_t n, int clamp,
                                                            const uint32_t* __restrict__ counts, const uint32_t* __restrict__ prefix,
                                                            const uint32_t* __restrict__ totals, const uint32_t* __restrict__ dw_off,
                                                            uint32_t row_stride, uint16_t* __restrict__ rec_e, uint2* __restrict__ rec_v,
                                                            uint32_t rec_cap, int64_t plane_stride, uint32_t level_lo
#ifdef J_WR_TIMING
                                                            , unsigned long long* __restrict__ dbg     // timing build (tools/wr_levels.py): [workgroup][24] clocks
#endif
                                                            ) {
#ifdef J_WR_TIMING
    if (threadIdx.x == 0) dbg[24 * blockIdx.x] = wall_clock64();
#endif
    __shared__ uint2 stxy[J_STAGE];                              // stage: {local entry | bin of the level << 16, d0}
    __shared__ uint32_t stz[J_STAGE];                            //        d1                                       (48 KiB together)
    __shared__ uint32_t cur[2][J_LVL_BINS];                      // [level parity][bin of the level]: stage cursor
    __shared__ uint32_t gra[2][J_LVL_BINS];                      // record index in the planes = cursor + gra
    __shared__ uint32_t grb[2][J_LVL_BINS];                      // joint bins: the same for the B record
    __shared__ uint32_t wtot[2][J_ROW_POINTS / 64];
    __shared__ uint32_t atot[2], ttot[2];                        // [parity]: records of the A bins (split level) / of all bins
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, lg16 = lane & (RUN_GROUP - 1);
    const uint32_t wave = tid >> 6;
    const size_t srow = (size_t)blockIdx.x * row_stride;
    const int64_t i = (int64_t)blockIdx.x * J_ROW_POINTS + tid;
    const bool in = i < n;
    float xv[3] = {0.f, 0.f, 0.f};
    if (in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) xv[k] = load_x(x, i, k, clamp);
    }
    auto load_dy = [&](uint32_t level, float (&d)[4]) {          // level-major gradient planes [L][N][2]
        d[0] = d[1] = d[2] = d[3] = 0.0f;
        if (in) {
            const float2 a = *reinterpret_cast<const float2*>(dyA + ((int64_t)level * plane_stride + i) * 2);
            const float2 b = *reinterpret_cast<const float2*>(dyB + ((int64_t)level * plane_stride + i) * 2);
            d[0] = a.x; d[1] = a.y; d[2] = b.x; d[3] = b.y;
        }
    };
    // cursor set-up of a level in three steps (thread t: bin t of the level): loads; wave scan of the counts; fix-up over the waves.
    // Step 3 needs step 2's wave totals of all waves: a barrier lies between them.
    auto setup_load = [&](uint32_t level, uint32_t& c, uint32_t& p, uint32_t& o, uint32_t& tt) {
        const uint32_t nlb = j_level_bins(lv.l[level]), b = lv.l[level].first + tid;
        c = 0u; p = 0u; o = 0u; tt = 0u;
        if (tid < nlb) { c = counts[srow + b]; p = prefix[srow + b]; o = dw_off[b]; tt = totals[b]; }
    };
    auto setup_scan = [&](int par, uint32_t c, uint32_t& incl) {
        incl = wave_incl_scan_u32(c);
        if (lane == 63) wtot[par][wave] = incl;
    };
    auto setup_fin = [&](int par, uint32_t level, uint32_t c, uint32_t p, uint32_t o, uint32_t tt, uint32_t incl) {
        const JLevel& q = lv.l[level];
        const uint32_t nlb = j_level_bins(q), split = q.flags & J_SPLIT;
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += wtot[par][w];
        const uint32_t excl = before + incl - c;
        if (tid < nlb) {
            cur[par][tid] = excl;
            gra[par][tid] = o + p - excl;                        // record index = o + p + (cursor - excl)
            grb[par][tid] = o + j_region(tt) + p - excl;         // joint bin: the B region follows the region of the bin's tt A records
        }
        if (split && tid == (1u << q.lgA)) atot[par] = excl;
        if (tid == nlb - 1u) ttot[par] = excl + c;
    };

    // levels [level_lo, n_levels): the whole pass, or one part of a pass cut by levels (us_hashgrid_bwd_joint_part)
    uint32_t c1, p1, o1, t1, incl1;
    float dn[4];
    setup_load(level_lo, c1, p1, o1, t1);
    load_dy(level_lo, dn);
    setup_scan((int)(level_lo & 1u), c1, incl1);
    __syncthreads();
    setup_fin((int)(level_lo & 1u), level_lo, c1, p1, o1, t1, incl1);

    for (uint32_t level = level_lo; level < n_levels; ++level) {
#ifdef J_WR_TIMING
        if (threadIdx.x == 0) dbg[24 * blockIdx.x + 1 + level] = wall_clock64();
#endif
        const int par = level & 1;
        const JLevel q = lv.l[level];
        const bool split = (q.flags & J_SPLIT) != 0u, has_next = level + 1 < n_levels;
        const float dy[4] = {dn[0], dn[1], dn[2], dn[3]};
        // ---- cell, position, vertex runs (binned_dev.h), entries
        float pos[3]; uint32_t cell[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pos_fract(xv[k], q.scale, pos[k], cell[k]);
        uint32_t key[8];
        slot_keys(cell, in && (q.flags & J_PACKABLE), lane, key);
        const uint2 ht = slot_run_masks(key, lg16);
        const uint32_t tail = in ? ht.y : 0u;
        uint32_t take_all, steps;                                // the run scan's bookkeeping, done once for both grids
        SLOT_SCAN_PRE(ht.x, take_all, steps)
        // ---- slot products of ONE grid (value pair g of the 4 gradients) and their segmented scan over the runs.  Weights and
        //      entries are derived again for the second grid instead of being held in 16 registers across the record phases.
        float val[8][2];
        auto products = [&](int g) {
            float w[8];
            slot_weights(pos, cell, w);
#pragma unroll
            for (int p = 0; p < 8; ++p) { val[p][0] = w[p] * dy[2 * g]; val[p][1] = w[p] * dy[2 * g + 1]; }
        };
        products(0);
        slot_scan_apply_pairs(val, take_all, steps);
        // ---- requests for the next level: its cursor inputs and gradients.  They are consumed below, BEFORE this level's first
        //      record store: a wait for a load also waits for every store issued before it (one vmcnt counter).
        if (has_next) { setup_load(level + 1, c1, p1, o1, t1); load_dy(level + 1, dn); }
        auto consume_next = [&]() {
            if (has_next) {
                asm volatile("" : "+v"(dn[0]), "+v"(dn[1]), "+v"(dn[2]), "+v"(dn[3]), "+v"(c1), "+v"(p1), "+v"(o1), "+v"(t1));
                setup_scan(par ^ 1, c1, incl1);
            }
        };
        // stage entry: {local entry | bin of the level << 16, d0, d1}; its position k in the stage is its cursor value (minus `sb`); at
        // copy-out it becomes record R = gr[bin] + k + sb: a 2-byte store into the entry plane and an 8-byte store into the value plane
        auto stage_put = [&](uint32_t k, uint32_t loc, float d0, float d1, uint32_t bin) {
            stxy[k] = make_uint2(loc | (bin << 16), __float_as_uint(d0)); stz[k] = __float_as_uint(d1);
        };
        auto copy_out = [&](uint32_t cnt, const uint32_t* gr, uint32_t sb, auto with_entries) {
            auto put = [&](const uint2 a, const uint32_t z, uint32_t k) {
#if defined(US_EXPERIMENTS) && (defined(J_WR_X_NOATOMIC) || defined(J_WR_X_LINEAR_STAGE))      // (timing experiments: the stage is in no order; keep the copy-out's stores contiguous)
                const uint32_t R = (uint32_t)(((uint64_t)(blockIdx.x * n_levels + level) * J_STAGE + k + sb + (gr[(a.x >> 16) & (J_LVL_BINS - 1)] & 0u)) % rec_cap);
#else
                const uint32_t R = gr[a.x >> 16] + k + sb;
#endif
                if (R < rec_cap) {
                    if (decltype(with_entries)::value) rec_e[R] = (uint16_t)a.x;
                    rec_v[R] = make_uint2(a.y, z);
                }
            };
            uint32_t k = tid;
            for (; k + J_ROW_POINTS < cnt; k += 2 * J_ROW_POINTS) {
                const uint2 a0 = stxy[k], a1 = stxy[k + J_ROW_POINTS];
                const uint32_t z0 = stz[k], z1 = stz[k + J_ROW_POINTS];
                put(a0, z0, k); put(a1, z1, k + J_ROW_POINTS);
            }
            if (k < cnt) put(stxy[k], stz[k], k);
        };
        lds_barrier();                                           // the previous level's copy-out has left the stage; cursors are in place
        if (!split) {
            // joint level: the stage entry carries table B's local entry; table A's is that masked (at copy-out), so the second half
            // only replaces the two values at the SAME positions -- no entries, bins or cursors again
            const uint32_t nbm = (1u << q.lgB) - 1u, lg = q.lgB;
            uint32_t k8[8];                                      // stage positions of this lane's records (0xFFFF: none)
            {
                uint32_t e[8];
                slot_entries((q.flags & J_HASHED_B) != 0u, q.hsB, q.res, q.res2, cell, e);
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    k8[c] = 0xFFFFu;
#if defined(US_EXPERIMENTS) && defined(J_WR_X_NOATOMIC)         // timing experiment (garbage results): no cursor atomics, conflict-free stage positions
                    if ((tail >> c) & 1u) k8[c] = (uint32_t)c * J_ROW_POINTS + tid;
#elif defined(US_EXPERIMENTS) && defined(J_WR_X_LINEAR_STAGE)   // timing experiment (garbage results): the atomics run, the stage positions are conflict-free
                    if ((tail >> c) & 1u) { const uint32_t dummy = atomicAdd(&cur[par][(e[c] >> BIN_LINE_LOG2) & nbm], 1u); k8[c] = dummy < 0xFFFFFFFFu ? (uint32_t)c * J_ROW_POINTS + tid : 0u; }
#else
                    if ((tail >> c) & 1u) k8[c] = atomicAdd(&cur[par][(e[c] >> BIN_LINE_LOG2) & nbm], 1u);
#endif
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (k8[c] < J_STAGE)
                        stage_put(k8[c], local_of(e[c], lg), val[c][0], val[c][1], (e[c] >> BIN_LINE_LOG2) & nbm);
                }
            }
            consume_next();
            lds_barrier();
            const uint32_t cnt = min(ttot[par], (uint32_t)J_STAGE);
            copy_out(cnt, gra[par], 0u, JBufTag<0>{});           // values only: table A reads the B region's entries, masked
            if (has_next) setup_fin(par ^ 1, level + 1, c1, p1, o1, t1, incl1);
            products(1);
            slot_scan_apply_pairs(val, take_all, steps);
            lds_barrier();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (k8[c] < J_STAGE) { stxy[k8[c]].y = __float_as_uint(val[c][0]); stz[k8[c]] = __float_as_uint(val[c][1]); }
            }
            lds_barrier();
            copy_out(cnt, grb[par], 0u, JBufTag<1>{});
        } else {
            const uint32_t nbA = 1u << q.lgA, mA = nbA - 1u, mB = (1u << q.lgB) - 1u;
            {                                                    // table A: bins [0, nbA) of the level, stage index = cursor
                uint32_t e[8];
                slot_entries((q.flags & J_HASHED_A) != 0u, q.hsA, q.res, q.res2, cell, e);
                uint32_t k8[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    k8[c] = 0xFFFFu;
                    if ((tail >> c) & 1u) k8[c] = atomicAdd(&cur[par][(e[c] >> BIN_LINE_LOG2) & mA], 1u);
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (((tail >> c) & 1u) && k8[c] < J_STAGE)
                        stage_put(k8[c], local_of(e[c], q.lgA), val[c][0], val[c][1], (e[c] >> BIN_LINE_LOG2) & mA);
                }
            }
            consume_next();
            lds_barrier();
            const uint32_t na = min(atot[par], (uint32_t)J_STAGE);
            copy_out(na, gra[par], 0u, JBufTag<1>{});
            if (has_next) setup_fin(par ^ 1, level + 1, c1, p1, o1, t1, incl1);
            products(1);
            slot_scan_apply_pairs(val, take_all, steps);
            lds_barrier();
            const uint32_t sb = atot[par];
            {                                                    // table B: bins behind the A bins, stage index = cursor - records of the A bins
                uint32_t e[8];
                slot_entries((q.flags & J_HASHED_B) != 0u, q.hsB, q.res, q.res2, cell, e);
                uint32_t k8[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    k8[c] = 0xFFFFFFFFu;
                    if ((tail >> c) & 1u) k8[c] = atomicAdd(&cur[par][nbA + ((e[c] >> BIN_LINE_LOG2) & mB)], 1u);
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (((tail >> c) & 1u) && k8[c] - sb < J_STAGE)
                        stage_put(k8[c] - sb, local_of(e[c], q.lgB), val[c][0], val[c][1], nbA + ((e[c] >> BIN_LINE_LOG2) & mB));
                }
            }
            lds_barrier();
            copy_out(min(ttot[par] - sb, (uint32_t)J_STAGE), gra[par], sb, JBufTag<1>{});
        }
    }
#ifdef J_WR_TIMING
    if (threadIdx.x == 0) dbg[24 * blockIdx.x + 17] = wall_clock64();
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// accumulate, persistent form: J_ACCP_GROUPS workgroups (four per CU, all resident), each with a list of items (item = (bin or extra
// chunk of a hot bin, grid)).  One workgroup per item spends most of its life outside the record stream: arguments, offsets, first
// records, clear, two barriers, sweep -- ~12 us for ~50 KB of records.  Here wave 0 reads the descriptors of ALL items of the workgroup
// up front (k_jitems' table: one round trip), and the record stream runs as ONE software pipeline over the items: the first records of
// item k + 1 are requested before the last batch of item k is added and are in flight across its barriers and its sweep (the
// barriers order LDS only).  The sweep returns the accumulators it read to zero, so no clear pass and no third barrier per item.
// r4, with tools/acc_balance.py (per-workgroup and per-item clocks of a timing build, bench batch):
//  * the kernel used to decode its items itself -- header, extra list, a walk over the levels, offsets: 22 KiB of straight-line code that
//    one wave runs once -- and every workgroup spent its first 9-10 us on instruction fetch; with the table the list stands after 1.7 us;
//  * with the numbering 2 x (extra slot or bin) | grid, 18 % of the indices were empty (unused extra slots, the other grid's side of
//    single-grid bins) and the workgroups drew 2 to 4 items: the table numbers the items densely;
//  * 2048 workgroups in two rounds paid the start-up twice: 1024, one round;
//  * what is left: the four workgroups of a CU do not run at one rate (the wave scheduler prefers the older waves: first-placed ones end
//    at 45 us, last-placed at 70-80 us) -- but the CU's rate is what the memory system gives it, 341 MB of records + 45 MB of
//    gradients in ~75 us; longer lists for the early workgroups moved their ends, not the kernel's, and drawing items from a counter
//    evened the ends out at the price of a counter and a descriptor round trip per item (or a look-ahead as long as the lists): slower;
//  * runs of records on ONE entry (every ray of a keyframe starts at its camera: the bins that own the vertices of those cells) serialise
//    a wave's ds_add_f64; summing such runs in registers first (segmented DPP scan) took the items of levels 2-5 from 12-28 us to
//    8-19 us -- and left the kernel where it was: the time went to their neighbours on the CU;
//  * s_setprio handed round the four workgroups of a CU item by item: ends 49-74 us by placement instead of 45-69, kernel 87 us: slower.
//    (The same in the record pass, whose two workgroups per CU end at 99 and 124 us -- tools/wr_levels.py: priority changing hands
//    level by level moved the first to 113 us and the second nowhere: that kernel is bound by what a CU issues, not by who issues it.)
// ---------------------------------------------------------------------------------------------------------------
#ifndef J_ACCP_GROUPS
#define J_ACCP_GROUPS 1024
#endif
#define J_ACCP_MAXI 32                   // items per workgroup: 2 * (ACC_EXTRA_MAX + J_MAX_BINS) / J_ACCP_GROUPS = 16.5
static_assert(2 * (ACC_EXTRA_MAX + J_MAX_BINS) <= J_ACCP_GROUPS * J_ACCP_MAXI, "k_jaccum_p: items per workgroup");
static_assert(J_MAX_LEVELS <= 16, "JI_MISC holds the level in four bits");
enum { JI_BASE, JI_C0, JI_C1, JI_NL, JI_MISC, JI_HS, JI_GOFF, JI_EBASE, JI_EMASK, JI_FIELDS };
// JI_MISC: bin of the level | lg << 16 | side << 24 | hot << 25 | empty bin << 26 | nothing behind this index << 27 | level << 28
#define JT_CAP (2 * (ACC_EXTRA_MAX + J_MAX_BINS))        // item descriptors: [JI_FIELDS][JT_CAP] in the workspace header

// ---------------------------------------------------------------------------------------------------------------
// item descriptors of the accumulate pass (one thread per item, behind k_jscan on its stream).  Items are numbered densely, grid A's
// first: [extra chunks of hot bins | bins with a side in A, level by level] then the same for grid B -- every index stands for work
// except the extra chunks of single-grid bins in the other grid's list.  (With the numbering 2 x (extra slot or bin) | grid that the
// accumulate kernel used to decode itself, 18 % of the indices of the bench batch were empty and its workgroups drew 2 to 4 items.)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_jitems(JLevels lv, uint32_t n_levels, uint32_t e_max, uint32_t SA, uint32_t SB,
                                               const uint32_t* __restrict__ rec_off, const uint32_t* __restrict__ dw_off,
                                               const uint32_t* __restrict__ extra, const uint32_t* __restrict__ hdr,
                                               uint32_t* __restrict__ items_tab) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_x = min(hdr[0], e_max), CH = hdr[1];
    const uint32_t nA = n_x + SA, nB = n_x + SB;
    if (u >= nA + nB) return;
    const uint32_t side = u < nA ? 0u : 1u, v = side ? u - nA : u;
    const bool is_extra = v < n_x;
    uint32_t b = 0, chunk = 0;
    if (is_extra) { const uint32_t pk = extra[v]; b = pk & 0xFFFFu; chunk = pk >> 16; }
    // ONE walk over the levels finds the bin (by its number: an extra chunk; by its place in this grid's list otherwise) and what the
    // accumulate pass needs to know about it
    JBin jb; jb.level = 0; jb.kind = 0; jb.bl = 0;
    uint32_t hs = 1u, lg = 0u, goff = 0u, emask = 0xFFFFu, r = v - n_x;
    bool found = false;
#pragma unroll
    for (uint32_t l = 0; l < J_MAX_LEVELS; ++l) {
        const JLevel& q = lv.l[l];
        if (l < n_levels && !found) {
            const bool split = (q.flags & J_SPLIT) != 0u;
            const uint32_t nbA = 1u << q.lgA, nbB = 1u << q.lgB, slots = (split && !side) ? nbA : nbB, nlb = split ? nbA + nbB : nbB;
            const bool hit = is_extra ? (b - q.first < nlb) : (r < slots);
            if (hit) {
                found = true;
                if (!is_extra) b = q.first + ((split && side) ? nbA : 0u) + r;
                const uint32_t rel = b - q.first;
                jb.level = l; jb.kind = split ? (rel < nbA ? 1u : 2u) : 0u; jb.bl = (split && rel >= nbA) ? rel - nbA : rel;
                hs = side ? q.hsB : q.hsA; lg = side ? q.lgB : q.lgA; goff = side ? q.offB : q.offA;
                if (jb.kind == 0u && side == 0u) emask = q.maskA & 0xFFFFu;                      // joint bin, table A: the B region's entries, masked
            } else if (!is_extra) r -= slots;
        }
    }
    bool ok = found;
    if ((jb.kind == 1u && side == 1u) || (jb.kind == 2u && side == 0u)) ok = false;            // a single-grid bin has no records of the other grid
    uint32_t cnt = 0, dwo = 0;
    if (ok) { const uint32_t b0 = rec_off[b]; cnt = rec_off[b + 1] - b0; dwo = dw_off[b]; }
    const uint32_t c0 = chunk * CH, c1 = (cnt > c0 && cnt - c0 > CH) ? c0 + CH : cnt;
    const uint32_t f[JI_FIELDS] = {dwo + ((jb.kind == 0u && side == 1u) ? j_region(cnt) : 0u), ok ? c0 : 0u, ok ? c1 : 0u, bin_n_local(hs, jb.bl, lg),
                                   jb.bl | (lg << 16) | (side << 24) | ((cnt > CH ? 1u : 0u) << 25) | ((ok && cnt == 0u && chunk == 0u ? 1u : 0u) << 26) |
                                       ((ok ? 0u : 1u) << 27) | (jb.level << 28),
                                   hs, goff, dwo + (jb.kind == 0u ? j_region(cnt) : 0u), emask};
#pragma unroll
    for (int k = 0; k < JI_FIELDS; ++k) items_tab[(size_t)k * JT_CAP + u] = f[k];
}


// torch.optim.Adam of the two tables' param groups (src/Mapper.py:118-126,445) applied INSIDE the sweep (us_hashgrid_bwd_joint_adam): with
// US_GRID_BWD_OVERWRITE | US_GRID_BWD_DETERMINISTIC every entry of both tables is written exactly once, by the workgroup that owns its
// bin, at the moment its final sum stands in LDS -- so that workgroup reads p, m, v of the entry and writes them back: no gradient table
// written (write_grad 0), none read back by an optimiser pass, one kernel boundary less.  The arithmetic per element is adam_segs_body's
// (render.hip): the same bits as the separate pass.  p == nullptr: off.
// (the tables' optimiser step inside the sweep is an EXPERIMENTS entry point -- us_hashgrid_bwd_joint_adam, measured slower than the separate
//  streaming pass: DESIGN.md 9 -- so the shipped build compiles its branches out of k_jaccum_p)
#ifdef US_EXPERIMENTS
#define J_ADAM_IN_SWEEP(ta) ((ta).pA != nullptr)
#else
#define J_ADAM_IN_SWEEP(ta) false
#endif
struct JTableAdam {
    float *pA, *mA, *vA, *pB, *mB, *vB;
    float lrA, lrB, one_minus_b1, b2, one_minus_b2, eps;
    const float* step_dev;               // us_adam_step_inc's float[8]: ALREADY advanced for this step
    int write_grad;
};
__device__ __forceinline__ void j_adam2(float g0, float g1, float* __restrict__ P, float* __restrict__ M, float* __restrict__ V, float step_size,
                                        float bc2_sqrt, const JTableAdam& ta) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 mv = __builtin_nontemporal_load(reinterpret_cast<const f2*>(M)), vv = __builtin_nontemporal_load(reinterpret_cast<const f2*>(V));
    f2 pv = *reinterpret_cast<const f2*>(P), mo, vo;
    {
        const float mi = mv.x + ta.one_minus_b1 * (g0 - mv.x);
        const float vi = vv.x * ta.b2 + (ta.one_minus_b2 * g0) * g0;
        const float denom = sqrtf(vi) / bc2_sqrt + ta.eps;
        pv.x = pv.x + (-step_size) * (mi / denom);
        mo.x = mi; vo.x = vi;
    }
    {
        const float mi = mv.y + ta.one_minus_b1 * (g1 - mv.y);
        const float vi = vv.y * ta.b2 + (ta.one_minus_b2 * g1) * g1;
        const float denom = sqrtf(vi) / bc2_sqrt + ta.eps;
        pv.y = pv.y + (-step_size) * (mi / denom);
        mo.y = mi; vo.y = vi;
    }
    *reinterpret_cast<f2*>(P) = pv;
    __builtin_nontemporal_store(mo, reinterpret_cast<f2*>(M)); __builtin_nontemporal_store(vo, reinterpret_cast<f2*>(V));
}

__global__ __launch_bounds__(J_ACC_THREADS, 8) void k_jaccum_p(uint32_t e_max, uint32_t SA, uint32_t SB,
                                                            const uint32_t* __restrict__ items_tab, const uint32_t* __restrict__ hdr,
                                                            const uint16_t* __restrict__ rec_e, const uint2* __restrict__ rec_v,
                                                            float* __restrict__ gradA,
                                                            float* __restrict__ gradB, uint16_t* __restrict__ gradB16, int overwrite, int side_sel,
                                                            const JTableAdam ta, uint32_t binA0, uint32_t binA1, uint32_t binB0, uint32_t binB1
#ifdef J_ACC_TIMING
                                                            , unsigned long long* __restrict__ dbg     // timing build (tools/acc_balance.py): per workgroup
#endif                                                                                                 // start, end (100 MHz ticks), items, records
                                                            ) {
#ifdef J_ACC_TIMING
    const unsigned long long dbg_t0 = wall_clock64();
    unsigned long long dbg_rec = 0;
#endif
    // side_sel: -1 both grids' items; 0 / 1: the items of grid A / B only (the data-parallel step finishes the colour table first, so
    // that its all-reduce travels while the sdf table is summed)
    __shared__ double acc[J_ACC_DOUBLES + 4 * 8];
    __shared__ uint32_t items[JI_FIELDS][J_ACCP_MAXI];           // items with records, in the order they are taken
    __shared__ uint32_t zitems[JI_FIELDS][J_ACCP_MAXI];          // OVERWRITE: bins nothing landed in (their entries get a zero gradient)
    __shared__ uint32_t n_items, n_zitems;
    const uint32_t tid = threadIdx.x;
    for (uint32_t k = tid; k < J_ACC_DOUBLES + 4 * 8; k += J_ACC_THREADS) acc[k] = 0.0;
    float ssA = 0.0f, ssB = 0.0f, bc2s = 1.0f;                   // Adam in the sweep: step sizes lr / (1 - b1^t), sqrt(1 - b2^t) as k_adam_segs forms them
    if (J_ADAM_IN_SWEEP(ta)) {
        const double* aux = reinterpret_cast<const double*>(ta.step_dev + 2);
        ssA = (float)((double)ta.lrA / aux[0]); ssB = (float)((double)ta.lrB / aux[0]); bc2s = (float)aux[1];
    }
    // ---- which items: workgroup b takes the table's indices i G + b on even turns i and i G + (G - 1 - b) on odd ones -- within a turn
    // the items grow with the index (finer levels: more records per bin), and the alternation cancels that trend over a list
    const uint32_t G = gridDim.x;
    const uint32_t n_x = min(hdr[0], e_max);
    const uint32_t nA = n_x + SA, nB = n_x + SB;                 // items of grid A / B: extra chunks + bins with a side in that grid
    // the items taken: grid A's indices [a0, a1) and grid B's [b0, b1) of its list -- everything (or one grid: side_sel), or, for a pass cut by
    // levels (binA1 > binA0 or binB1 > binB0; unsplit bins: no extra chunks), the bins [binX0, binX1) of each grid's level-by-level list
    const bool part = binA1 > binA0 || binB1 > binB0;
    const uint32_t a0 = part ? n_x + binA0 : 0u, a1 = part ? n_x + binA1 : (side_sel == 1 ? 0u : nA);
    const uint32_t b0 = part ? n_x + binB0 : 0u, b1 = part ? n_x + binB1 : (side_sel == 0 ? 0u : nB);
    const uint32_t cntA = a1 - a0, cnt_u = cntA + (b1 - b0);
    if (tid < 64) {                                              // wave 0, lane i: the workgroup's i-th turn
        const uint32_t turn = tid;
        const uint32_t v = turn * G + ((turn & 1u) ? G - 1u - blockIdx.x : blockIdx.x);
        const bool in = turn < (uint32_t)J_ACCP_MAXI && v < cnt_u;
        const uint32_t u = in ? (v < cntA ? a0 + v : nA + b0 + (v - cntA)) : 0u;
        uint32_t fields[JI_FIELDS];
#pragma unroll
        for (int f = 0; f < JI_FIELDS; ++f) fields[f] = items_tab[(size_t)f * JT_CAP + u];
        const bool valid = in && !((fields[JI_MISC] >> 27) & 1u);
        const bool has = valid && fields[JI_C1] > fields[JI_C0], zero = valid && ((fields[JI_MISC] >> 26) & 1u) && overwrite;
        const uint64_t mh = __ballot(has), mz = __ballot(zero), below = (1ull << tid) - 1ull;
        if (has) { const uint32_t p = (uint32_t)__popcll(mh & below);
#pragma unroll
            for (int f = 0; f < JI_FIELDS; ++f) items[f][p] = fields[f]; }
        if (zero) { const uint32_t p = (uint32_t)__popcll(mz & below);
#pragma unroll
            for (int f = 0; f < JI_FIELDS; ++f) zitems[f][p] = fields[f]; }
        if (tid == 0) { n_items = (uint32_t)__popcll(mh); n_zitems = (uint32_t)__popcll(mz); }
    }
    __syncthreads();
    const uint32_t n = n_items, nz = n_zitems;
    auto field = [&](int f, uint32_t k) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)items[f][k]); };
    constexpr uint32_t STEP = J_ACC_THREADS * J_ACC_UNROLL * 2;  // records per batch: every thread takes J_ACC_UNROLL pairs
    constexpr int EPT = J_ACC_DOUBLES / 2 / J_ACC_THREADS;       // entries per thread in the sweep
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(16)));
    const uint32_t* rec_e2 = reinterpret_cast<const uint32_t*>(rec_e);
    const u32x4* rec_v2 = reinterpret_cast<const u32x4*>(rec_v);
    uint32_t we[2][J_ACC_UNROLL];                                // a pair's two local entries
    u32x4 wv[2][J_ACC_UNROLL];                                   // ... and its four values
    // ---- the request side of the pipeline: item fk, records from fa on (fa is even: chunks of hot bins start at multiples of hdr[1])
    uint32_t fk = 0, fa = 0, f_c1 = 0, f_base = 0, f_ebase = 0;
    auto f_load = [&]() { if (fk < n) { fa = field(JI_C0, fk); f_c1 = field(JI_C1, fk); f_base = field(JI_BASE, fk); f_ebase = field(JI_EBASE, fk); } };
    auto fetch = [&](auto tag) {                                 // every load unconditional (the compiler closes a conditional block with
        constexpr int buf = decltype(tag)::value;                // s_waitcnt vmcnt(0): ONE record in flight per thread): past the end of an
                                                                 // item, and past the last item of the workgroup, it re-reads the last pair
        const size_t pair0 = (size_t)(f_base >> 1), epair0 = (size_t)(f_ebase >> 1);     // (regions start at multiples of 8 records)
#pragma unroll
        for (int u = 0; u < J_ACC_UNROLL; ++u) {
            const uint32_t q = min((fa >> 1) + u * J_ACC_THREADS + tid, (f_c1 - 1u) >> 1);
            we[buf][u] = __builtin_nontemporal_load(rec_e2 + epair0 + q);                                        // two 16-bit entries
            wv[buf][u] = __builtin_nontemporal_load(rec_v2 + pair0 + q);                                         // their four values
        }
        fa += STEP;
        if (fa >= f_c1) { ++fk; f_load(); }
    };
    // ---- the adding side: item k, records from a on
    uint32_t k = 0, a = 0, c1 = 0, nl = 0, misc = 0, hs = 0, goff = 0, REP = 1, rstride = 0, emask = 0xFFFFu;
    double* my = acc;
    auto a_load = [&]() {
        a = field(JI_C0, k); c1 = field(JI_C1, k); nl = field(JI_NL, k); misc = field(JI_MISC, k); hs = field(JI_HS, k); goff = field(JI_GOFF, k);
        emask = field(JI_EMASK, k);
        // slices of few entries (the sdf table: 256 per bin, 25 records per entry) are kept in 4 copies, which thins out same-address
        // collisions of the LDS atomics; the copies sit 2 nl + 8 doubles apart (bank spread)
        REP = (8u * nl <= J_ACC_DOUBLES) ? 4u : 1u; rstride = 2u * nl + 8u;
        my = acc + (tid & (REP - 1u)) * rstride;
#ifdef J_ACC_TIMING
        dbg_rec += c1 - a;
#endif
    };
    auto sweep = [&]() {                                         // sums of the item -> gradient table; the accumulators return to zero
        const uint32_t bl = misc & 0xFFFFu, lg = (misc >> 16) & 0xFFu, side = (misc >> 24) & 1u;
        const bool hot = ((misc >> 25) & 1u) != 0u;
        float* gl = (side ? gradB : gradA) + (size_t)goff * 2u;
#pragma unroll
        for (int e4 = 0; e4 < EPT; ++e4) {
            const uint32_t loc = tid + e4 * J_ACC_THREADS;
            if (loc >= nl) continue;
            double s0 = acc[loc], s1 = acc[nl + loc];
            acc[loc] = 0.0; acc[nl + loc] = 0.0;
            for (uint32_t r = 1; r < REP; ++r) {                 // fixed order
                s0 += acc[r * rstride + loc]; s1 += acc[r * rstride + nl + loc];
                acc[r * rstride + loc] = 0.0; acc[r * rstride + nl + loc] = 0.0;
            }
            const uint32_t e = entry_of(loc, bl, lg);
            if (e >= hs) continue;
            float* p = gl + (size_t)e * 2u;
            const float v0 = (float)s0, v1 = (float)s1;
            if (J_ADAM_IN_SWEEP(ta) && !hot) {                                 // (deterministic mode: no bin is hot)
                const size_t o = ((size_t)goff + e) * 2u;
                j_adam2(v0, v1, (side ? ta.pB : ta.pA) + o, (side ? ta.mB : ta.mA) + o, (side ? ta.vB : ta.vA) + o, side ? ssB : ssA, bc2s, ta);
                if (ta.write_grad) *reinterpret_cast<float2*>(p) = make_float2(v0, v1);
            } else if (hot) {
                if (v0 != 0.0f) atomicAdd(p, v0);
                if (v1 != 0.0f) atomicAdd(p + 1, v1);
            } else if (overwrite) {
                *reinterpret_cast<float2*>(p) = make_float2(v0, v1);
                if (side && gradB16) {                           // grid B's gradient also as a bfloat16 image (round to nearest even, as a
                    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));      // tensor copy rounds): the payload of a data-parallel all-reduce
                    bf2 o; o.x = (__bf16)v0; o.y = (__bf16)v1;
                    *reinterpret_cast<bf2*>(gradB16 + ((size_t)goff + e) * 2u) = o;
                }
            } else if (v0 != 0.0f || v1 != 0.0f) {               // this workgroup is the only writer of its entries
                float2 o = *reinterpret_cast<const float2*>(p);
                o.x += v0; o.y += v1;
                *reinterpret_cast<float2*>(p) = o;
            }
        }
    };
    auto step = [&](auto tag, auto other) -> bool {              // request the next batch, add this one; true when the workgroup is done
        constexpr int buf = decltype(tag)::value;
        fetch(other);
#pragma unroll
        for (int u = 0; u < J_ACC_UNROLL; ++u) {
            const uint32_t r0 = a + 2u * (u * J_ACC_THREADS + tid);
            const uint32_t l0 = we[buf][u] & emask, l1 = (we[buf][u] >> 16) & emask;
            if (r0 < c1 && l0 < nl) {
                atomicAdd(&my[l0], (double)__uint_as_float(wv[buf][u].x));                        // ds_add_f64
                atomicAdd(&my[nl + l0], (double)__uint_as_float(wv[buf][u].y));
            }
            if (r0 + 1u < c1 && l1 < nl) {
                atomicAdd(&my[l1], (double)__uint_as_float(wv[buf][u].z));
                atomicAdd(&my[nl + l1], (double)__uint_as_float(wv[buf][u].w));
            }
        }
        a += STEP;
        if (a >= c1) {                                           // the item's last batch (workgroup-uniform)
            lds_barrier();                                       // LDS only: the next item's records stay in flight
            sweep();
            if (++k == n) return true;
            a_load();
            lds_barrier();
        }
        return false;
    };
    if (n > 0) {
        f_load(); a_load();
        fetch(JBufTag<0>{});
        for (;;) {
            if (step(JBufTag<0>{}, JBufTag<1>{})) break;
            if (step(JBufTag<1>{}, JBufTag<0>{})) break;
        }
    }
    for (uint32_t z = 0; z < nz; ++z) {                          // empty bins: zero gradient
        const uint32_t znl = zitems[JI_NL][z], zm = zitems[JI_MISC][z], zhs = zitems[JI_HS][z];
        float* gl = (((zm >> 24) & 1u) ? gradB : gradA) + (size_t)zitems[JI_GOFF][z] * 2u;
        for (uint32_t loc = tid; loc < znl; loc += J_ACC_THREADS) {
            const uint32_t e = entry_of(loc, zm & 0xFFFFu, (zm >> 16) & 0xFFu);
            if (e < zhs && J_ADAM_IN_SWEEP(ta)) {                              // a zero gradient still moves the entry: m and v decay, p follows m
                const uint32_t zs = (zm >> 24) & 1u;
                const size_t o = ((size_t)zitems[JI_GOFF][z] + e) * 2u;
                j_adam2(0.0f, 0.0f, (zs ? ta.pB : ta.pA) + o, (zs ? ta.mB : ta.mA) + o, (zs ? ta.vB : ta.vA) + o, zs ? ssB : ssA, bc2s, ta);
                if (ta.write_grad) *reinterpret_cast<float2*>(gl + (size_t)e * 2u) = make_float2(0.0f, 0.0f);
            } else if (e < zhs) {
                *reinterpret_cast<float2*>(gl + (size_t)e * 2u) = make_float2(0.0f, 0.0f);
                if (((zm >> 24) & 1u) && gradB16) *reinterpret_cast<uint32_t*>(gradB16 + ((size_t)zitems[JI_GOFF][z] + e) * 2u) = 0u;
            }
        }
    }
#ifdef J_ACC_TIMING
    if (tid == 0) { dbg[4 * blockIdx.x] = dbg_t0; dbg[4 * blockIdx.x + 1] = wall_clock64(); dbg[4 * blockIdx.x + 2] = n; dbg[4 * blockIdx.x + 3] = dbg_rec << 32; }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
// workspace: totals | rec_off | dw_off (3 x (J_MAX_BINS + 64) u32) | hdr (16 u32) | extra (ACC_EXTRA_MAX u32) | item descriptors | count rows
// [rows][row_stride] | their column prefixes | records
static uint32_t j_row_stride(int TB) { return ((uint32_t)TB + 63u) & ~63u; }
static size_t j_header_bytes(int TB, int64_t n) {
    const size_t rows = (size_t)us_cdiv(n, J_ROW_POINTS);
    return (size_t)(3 * (J_MAX_BINS + 64) + 16 + ACC_EXTRA_MAX + JI_FIELDS * JT_CAP) * sizeof(uint32_t) + 2 * rows * j_row_stride(TB) * sizeof(uint32_t);
}
// record indices of both grids (every (bin, grid) region padded to 8) and the bytes of the two planes over them
static uint64_t j_record_cap(uint32_t n_levels, int64_t n) { return (uint64_t)n * 8ull * n_levels * 2ull + 16ull * J_MAX_BINS; }
static size_t j_record_bytes(uint32_t n_levels, int64_t n) { return (size_t)j_record_cap(n_levels, n) * 10u; }    // (cap is a multiple of 8: both planes 16-byte aligned)

extern "C" int us_hashgrid_joint_supported(const us_grid_desc* a, const us_grid_desc* b, int64_t n) {
    JLevels lv;
    const int TB = make_jlevels(a, b, n, &lv);
    if (TB <= 0) return 0;
    if (j_record_cap(a->n_levels, n) > 0xFFFFFFF0ull) return 0;              // 32-bit record indices
    return 1;
}

extern "C" size_t us_hashgrid_joint_workspace_bytes(const us_grid_desc* a, const us_grid_desc* b, int64_t n) {
    if (!us_hashgrid_joint_supported(a, b, n)) return 0;
    JLevels lv;
    const int TB = make_jlevels(a, b, n, &lv);
    return j_header_bytes(TB, n) + j_record_bytes(a->n_levels, n);
}

struct JWorkspace { uint32_t *totals, *rec_off, *dw_off, *hdr, *extra, *items, *counts, *prefix; uint16_t* rec_e; uint2* rec_v; uint32_t n_rows, stride, rec_cap; };
static JWorkspace j_carve(void* workspace, uint32_t n_levels, int TB, int64_t n) {
    JWorkspace w;
    w.totals = (uint32_t*)workspace;
    w.rec_off = w.totals + (J_MAX_BINS + 64);
    w.dw_off = w.rec_off + (J_MAX_BINS + 64);
    w.hdr = w.dw_off + (J_MAX_BINS + 64);
    w.extra = w.hdr + 16;
    w.items = w.extra + ACC_EXTRA_MAX;
    w.counts = w.items + (size_t)JI_FIELDS * JT_CAP;
    w.n_rows = (uint32_t)us_cdiv(n, J_ROW_POINTS); w.stride = j_row_stride(TB);
    w.prefix = w.counts + (size_t)w.n_rows * w.stride;
    w.rec_cap = (uint32_t)j_record_cap(n_levels, n);
    w.rec_e = (uint16_t*)((char*)workspace + j_header_bytes(TB, n));
    w.rec_v = (uint2*)((char*)w.rec_e + (size_t)w.rec_cap * 2u);
    return w;
}

#define J_CHECK_PAIR(name)                                                                                                   \
    US_REQUIRE(a && b, US_ERR_NULL, name ": desc is NULL");                                                                  \
    JLevels lv;                                                                                                              \
    const int TB = make_jlevels(a, b, n > 0 ? n : 1, &lv);                                                                   \
    US_REQUIRE(TB > 0 && us_hashgrid_joint_supported(a, b, n > 0 ? n : 1), US_ERR_CONFIG,                                    \
               name ": the two grids do not share a geometry this path takes (F = 2, <= %d levels, equal base resolution and " \
               "per-level scale, <= %d bins) or the batch is too large", J_MAX_LEVELS, J_MAX_BINS)

static int fwd_joint(const char* fn, const us_grid_desc* a, const us_grid_desc* b, const float* paramsA, const float* paramsB, const float* x,
                     int64_t n, float* outA, float* outB, us_half_t* dydxA, us_half_t* dydxB, int flags, void* workspace, size_t workspace_bytes,
                     void* stream) {
    if (n < 0) return US_ERR_SHAPE;
    US_REQUIRE(a && b, US_ERR_NULL, "%s: desc is NULL", fn);
    JLevels lv;
    // without a workspace only the geometry matters (the 32-bit record addresses bound the table gradient's batch, not the encoder's)
    const int TB = make_jlevels(a, b, workspace ? (n > 0 ? n : 1) : 1, &lv);
    US_REQUIRE(TB > 0 && (!workspace || us_hashgrid_joint_supported(a, b, n > 0 ? n : 1)), US_ERR_CONFIG,
               "%s: the two grids do not share a geometry this path takes (F = 2, <= %d levels, equal base resolution and per-level "
               "scale, <= %d bins) or the batch is too large", fn, J_MAX_LEVELS, J_MAX_BINS);
    if (n == 0) return US_OK;
    US_REQUIRE(paramsA && paramsB && x && outA && outB, US_ERR_NULL, "%s: NULL pointer", fn);
    US_REQUIRE(((uintptr_t)paramsA & 15u) == 0 && ((uintptr_t)paramsB & 15u) == 0, US_ERR_SHAPE, "%s: params must be 16-byte aligned", fn);
    US_REQUIRE((dydxA != nullptr) == (dydxB != nullptr), US_ERR_NULL, "%s: dy_dx of both grids or of neither", fn);
    US_REQUIRE(!dydxA || ((((uintptr_t)dydxA | (uintptr_t)dydxB) & 3u) == 0), US_ERR_SHAPE, "%s: dy_dx must be 4-byte aligned", fn);
    US_REQUIRE(!(flags & US_GRID_FEAT_SPLIT_BF16) || (flags & US_GRID_LEVEL_MAJOR), US_ERR_CONFIG, "%s: US_GRID_FEAT_SPLIT_BF16 needs level-major planes", fn);
    const int clamp = flags & US_GRID_CLAMP01, lm = ((flags & US_GRID_LEVEL_MAJOR) ? 1 : 0) | ((flags & US_GRID_FEAT_SPLIT_BF16) ? 2 : 0);
    dim3 grid((unsigned)us_cdiv(n, J_FWD_THREADS), a->n_levels), block(J_FWD_THREADS);
    hipStream_t s = (hipStream_t)stream;
    if (workspace) {
        US_REQUIRE(((uintptr_t)workspace & 15u) == 0, US_ERR_SHAPE, "%s: workspace must be 16-byte aligned", fn);
        US_REQUIRE(workspace_bytes >= us_hashgrid_joint_workspace_bytes(a, b, n), US_ERR_WORKSPACE,
                   "%s: workspace %zu B < %zu B", fn, workspace_bytes, us_hashgrid_joint_workspace_bytes(a, b, n));
        const JWorkspace w = j_carve(workspace, a->n_levels, TB, n);
        JPlan plan;
        memset(&plan, 0, sizeof(plan));
#if J_FWD_XCD
        if (!dydxA && j_big_slab(a, b) && make_jplan(a->n_levels, grid.x, J_PLAN_SLOPE, &plan)) grid = dim3(8u * plan.n, 1u);
#endif
        if (dydxA) hipLaunchKernelGGL((k_jfwd<true, true, true>), grid, block, 0, s, lv, plan, a->n_levels, paramsA, paramsB, x, n, outA, outB, clamp, lm, w.counts, w.stride, w.n_rows, dydxA, dydxB);
        else hipLaunchKernelGGL((k_jfwd<true, true, false>), grid, block, 0, s, lv, plan, a->n_levels, paramsA, paramsB, x, n, outA, outB, clamp, lm, w.counts, w.stride, w.n_rows, dydxA, dydxB
#ifdef J_FWD_TIMING
                                , (unsigned long long*)((char*)workspace + ((us_hashgrid_joint_workspace_bytes(a, b, n) & ~(size_t)7) - (size_t)393216))
#endif
                                );
    } else {
        // small batches (a tracking iteration: 80 000 points = 79 workgroups of 1024 per level) leave a ragged last round over the chip's
        // 256 CUs: 256-thread workgroups there
        const int threads = n >= J_SMALL_BATCH ? J_FWD_THREADS : 256;
        grid = dim3((unsigned)us_cdiv(n, threads), a->n_levels); block = dim3(threads);
        JPlan plan;
        memset(&plan, 0, sizeof(plan));
#if J_FWD_XCD
        if (j_big_slab(a, b) && make_jplan(a->n_levels, grid.x, J_PLAN_SLOPE, &plan)) grid = dim3(8u * plan.n, 1u);   // (also with dy/dx: the tracking encoder, -2.5 us)
#endif
        if (dydxA) hipLaunchKernelGGL((k_jfwd<true, false, true>), grid, block, 0, s, lv, plan, a->n_levels, paramsA, paramsB, x, n, outA, outB, clamp, lm, (uint32_t*)nullptr, 0u, 0u, dydxA, dydxB);
        else hipLaunchKernelGGL((k_jfwd<true, false, false>), grid, block, 0, s, lv, plan, a->n_levels, paramsA, paramsB, x, n, outA, outB, clamp, lm, (uint32_t*)nullptr, 0u, 0u, dydxA, dydxB);
    }
    US_CHECK_LAUNCH(fn);
    return US_OK;
}

extern "C" int us_hashgrid_fwd_joint(const us_grid_desc* a, const us_grid_desc* b, const float* paramsA, const float* paramsB,
                                     const float* x, int64_t n, float* outA, float* outB, int flags, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    return fwd_joint("us_hashgrid_fwd_joint", a, b, paramsA, paramsB, x, n, outA, outB, nullptr, nullptr, flags, workspace, workspace_bytes, stream);
}

extern "C" int us_hashgrid_fwd_joint_dydx(const us_grid_desc* a, const us_grid_desc* b, const float* paramsA, const float* paramsB,
                                          const float* x, int64_t n, float* outA, float* outB, us_half_t* dy_dxA, us_half_t* dy_dxB, int flags,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    US_REQUIRE(n <= 0 || (dy_dxA && dy_dxB), US_ERR_NULL, "us_hashgrid_fwd_joint_dydx: NULL pointer");
    return fwd_joint("us_hashgrid_fwd_joint_dydx", a, b, paramsA, paramsB, x, n, outA, outB, dy_dxA, dy_dxB, flags, workspace, workspace_bytes, stream);
}

// The input gradient from the stored dy_dx, reduced to the rays: a workgroup = one ray, lanes = 16 samples x 4 level rows, each lane
// its levels {row, row+4, ...} of grid A, then of grid B -- the summation order of k_bwd_input_rays (hashgrid.hip), so the two paths
// agree to the rounding of the stored half values -- then the adjoint of us_ray_points.  Pure streaming: 20 B per (point, level, grid).
#define JR_MAX_WAVES 8
struct JRaySpan { float span[3]; };
__global__ __launch_bounds__(64 * JR_MAX_WAVES) void k_dydx_rays(uint32_t n_levels, const float* __restrict__ dyA, const float* __restrict__ dyB,
                                                                 const us_half_t* __restrict__ dydxA, const us_half_t* __restrict__ dydxB, int64_t n, int S,
                                                                 const float* __restrict__ z_vals, JRaySpan bd, float* __restrict__ g_o,
                                                                 float* __restrict__ g_d, float* __restrict__ dL_dx) {
    __shared__ float sh[JR_MAX_WAVES][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, pt = lane & 15;
    const int64_t ray = blockIdx.x;
    const int s = wave * 16 + pt;
    const bool in = s < S;
    const int64_t i = ray * S + s;
    float rA[3] = {0.f, 0.f, 0.f}, rB[3] = {0.f, 0.f, 0.f};
    if (in) {
#pragma unroll
        for (int gsel = 0; gsel < 2; ++gsel) {
            const float* dy = gsel ? dyB : dyA; const _Float16* dd = reinterpret_cast<const _Float16*>(gsel ? dydxB : dydxA);
            float* r = gsel ? rB : rA;
            for (uint32_t level = (uint32_t)row; level < n_levels; level += 4) {
                const int64_t e = (int64_t)level * n + i;
                typedef float f2_t __attribute__((ext_vector_type(2)));
                const f2_t y = __builtin_nontemporal_load(reinterpret_cast<const f2_t*>(dy + e * 2));
                typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                const h2_t* d = reinterpret_cast<const h2_t*>(dd + (int64_t)level * 3 * n * 2 + i * 2);
                const h2_t e0 = __builtin_nontemporal_load(d), e1 = __builtin_nontemporal_load(d + n), e2 = __builtin_nontemporal_load(d + 2 * n);
                f2_t d0, d1, d2;
                d0.x = (float)e0.x; d0.y = (float)e0.y; d1.x = (float)e1.x; d1.y = (float)e1.y; d2.x = (float)e2.x; d2.y = (float)e2.y;
                float t[3];
                t[0] = y.x * d0.x; t[1] = y.x * d1.x; t[2] = y.x * d2.x;      // input_grad_level: r[gd] += dy[0] * d[0][gd], then dy[1] * d[1][gd]
                r[0] += t[0]; r[1] += t[1]; r[2] += t[2];
                r[0] += y.y * d0.y; r[1] += y.y * d1.y; r[2] += y.y * d2.y;
            }
        }
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

extern "C" int us_hashgrid_dydx_rays(uint32_t n_levels, const float* dL_dyA, const float* dL_dyB, const us_half_t* dy_dxA, const us_half_t* dy_dxB,
                                     int64_t n_rays, int n_samples, const float* z_vals, const float* bound_host, float* dL_do, float* dL_dd,
                                     float* dL_dx, void* stream) {
    US_REQUIRE(n_levels >= 1 && n_levels <= US_MAX_LEVELS, US_ERR_CONFIG, "us_hashgrid_dydx_rays: n_levels %u", n_levels);
    US_REQUIRE(n_samples >= 1 && n_samples <= 16 * JR_MAX_WAVES, US_ERR_SHAPE, "us_hashgrid_dydx_rays: n_samples %d not in 1..%d", n_samples, 16 * JR_MAX_WAVES);
    if (n_rays <= 0) return n_rays == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(dL_dyA && dL_dyB && dy_dxA && dy_dxB && z_vals && bound_host && dL_do && dL_dd, US_ERR_NULL, "us_hashgrid_dydx_rays: NULL pointer");
    JRaySpan bd;
    for (int k = 0; k < 3; ++k) bd.span[k] = bound_host[3 + k] - bound_host[k];
    const int waves = (n_samples + 15) / 16;
    hipLaunchKernelGGL(k_dydx_rays, dim3((unsigned)n_rays), dim3(64 * waves), 0, (hipStream_t)stream, n_levels, dL_dyA, dL_dyB, dy_dxA, dy_dxB,
                       n_rays * n_samples, n_samples, z_vals, bd, dL_do, dL_dd, dL_dx);
    US_CHECK_LAUNCH("us_hashgrid_dydx_rays");
    return US_OK;
}

static int bwd_joint(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA, const float* dL_dyB, int64_t n,
                     float* gradA, float* gradB, int flags, void* workspace, size_t workspace_bytes, void* stream, bool scan_only,
                     int64_t plane_stride = 0, uint16_t* gradB16 = nullptr, const JTableAdam* adam = nullptr, int part_lo = 0, int part_hi = 0,
                     int part_what = 0) {
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
    const bool scanned = (flags & US_GRID_BWD_SCANNED) != 0;
    const uint32_t chunk0 = (flags & US_GRID_BWD_DETERMINISTIC) ? 0xFFFFFFFFu : (uint32_t)ACC_CHUNK;
    US_REQUIRE(!(scan_only && !counted) && !(scanned && !counted), US_ERR_CONFIG,
               "us_hashgrid_bwd_joint: the scan passes can only run ahead on counts left by us_hashgrid_fwd_joint (US_GRID_BWD_COUNTED)");
    US_REQUIRE((scan_only || (x && dL_dyA && dL_dyB)) && gradA && gradB && workspace, US_ERR_NULL, "us_hashgrid_bwd_joint: NULL pointer");
    US_REQUIRE(flags & US_GRID_LEVEL_MAJOR, US_ERR_CONFIG, "us_hashgrid_bwd_joint: the gradients must be level-major planes (US_GRID_LEVEL_MAJOR)");
    US_REQUIRE(((uintptr_t)gradA & 15u) == 0 && ((uintptr_t)gradB & 15u) == 0 && ((uintptr_t)workspace & 15u) == 0 &&
               (scan_only || (((uintptr_t)dL_dyA & 7u) == 0 && ((uintptr_t)dL_dyB & 7u) == 0)), US_ERR_SHAPE,
               "us_hashgrid_bwd_joint: gradient tables and workspace must be 16-byte aligned, dL_dy 8-byte aligned");
    US_REQUIRE(workspace_bytes >= us_hashgrid_joint_workspace_bytes(a, b, n), US_ERR_WORKSPACE,
               "us_hashgrid_bwd_joint: workspace %zu B < %zu B", workspace_bytes, us_hashgrid_joint_workspace_bytes(a, b, n));
    const JWorkspace w = j_carve(workspace, a->n_levels, TB, n);
    const int clamp = flags & US_GRID_CLAMP01;
    const uint32_t L = a->n_levels;
    uint32_t SA = 0, SB = 0;                                     // bins with a side in grid A / B (a joint bin: both)
    for (uint32_t l = 0; l < L; ++l) { SA += (lv.l[l].flags & J_SPLIT) ? 1u << lv.l[l].lgA : 1u << lv.l[l].lgB; SB += 1u << lv.l[l].lgB; }
    if (!counted)
        hipLaunchKernelGGL((k_jfwd<false, true, false>), dim3((unsigned)us_cdiv(n, J_FWD_THREADS), L), dim3(J_FWD_THREADS), 0, s, lv, JPlan{}, L, (const float*)nullptr,
                           (const float*)nullptr, x, n, (float*)nullptr, (float*)nullptr, clamp, 1, w.counts, w.stride, w.n_rows);
    if (!scanned) {
        hipLaunchKernelGGL(k_jcolscan, dim3((unsigned)us_cdiv(TB, JCS_BINS)), dim3(JCS_THREADS), 0, s, lv, L, w.counts, w.prefix, w.n_rows, w.stride,
                           (uint32_t)TB, w.totals, gradA, gradB, overwrite, chunk0);
        JSingle sg;
        memset(&sg, 0, sizeof(sg));
        for (uint32_t l = 0; l < L; ++l)
            if (lv.l[l].flags & J_SPLIT) { sg.lo[l] = lv.l[l].first; sg.len[l] = j_level_bins(lv.l[l]); }
        hipLaunchKernelGGL(k_jscan, dim3(1), dim3(1024), 0, s, sg, w.totals, (uint32_t)TB, w.rec_off, w.dw_off, w.extra, w.hdr, chunk0);
        hipLaunchKernelGGL(k_jitems, dim3((2u * ACC_EXTRA_MAX + SA + SB + 255u) / 256u), dim3(256), 0, s, lv, L, (uint32_t)ACC_EXTRA_MAX, SA, SB, w.rec_off,
                           w.dw_off, w.extra, w.hdr, w.items);
    }
    if (scan_only) { US_CHECK_LAUNCH("us_hashgrid_joint_scan"); return US_OK; }
    const int side_sel = (flags & US_GRID_BWD_ONLY_A) ? 0 : ((flags & US_GRID_BWD_ONLY_B) ? 1 : -1);
    US_REQUIRE(!((flags & US_GRID_BWD_ONLY_A) && (flags & US_GRID_BWD_ONLY_B)), US_ERR_CONFIG, "us_hashgrid_bwd_joint: ONLY_A and ONLY_B are exclusive");
    US_REQUIRE(!(flags & US_GRID_BWD_RECORDS_READY) || side_sel >= 0, US_ERR_CONFIG,
               "us_hashgrid_bwd_joint: US_GRID_BWD_RECORDS_READY continues a call that summed the other grid (US_GRID_BWD_ONLY_A / _B)");
    // a pass cut by levels (us_hashgrid_bwd_joint_part): levels [part_lo, part_hi), record pass and / or accumulate pass
    const bool part = part_what != 0;
    uint32_t binA0 = 0, binA1 = 0, binB0 = 0, binB1 = 0;
    if (part) {
        for (uint32_t l = 0; l < L; ++l) {
            const uint32_t sa = (lv.l[l].flags & J_SPLIT) ? 1u << lv.l[l].lgA : 1u << lv.l[l].lgB, sb_ = 1u << lv.l[l].lgB;
            if ((int)l < part_lo) { binA0 += sa; binB0 += sb_; }
            if ((int)l < part_hi) { binA1 += sa; binB1 += sb_; }
        }
    }
    if (!(flags & US_GRID_BWD_RECORDS_READY) && (!part || (part_what & 1)))
        hipLaunchKernelGGL(k_jwrite, dim3(w.n_rows), dim3(J_ROW_POINTS), 0, s, lv, part ? (uint32_t)part_hi : L, x, dL_dyA, dL_dyB, n, clamp, w.counts, w.prefix, w.totals, w.dw_off,
                           w.stride, w.rec_e, w.rec_v, w.rec_cap, plane_stride > 0 ? plane_stride : n, part ? (uint32_t)part_lo : 0u
#ifdef J_WR_TIMING
                           , (unsigned long long*)((char*)workspace + ((us_hashgrid_joint_workspace_bytes(a, b, n) & ~(size_t)7) - (size_t)393216))   // the planes' unused end
#endif
                           );
    if (part && !(part_what & 2)) { US_CHECK_LAUNCH("us_hashgrid_bwd_joint_part"); return US_OK; }
    const uint32_t n_acc_items = part ? (binA1 - binA0) + (binB1 - binB0) : (side_sel < 0 ? 2u * ACC_EXTRA_MAX + SA + SB : ACC_EXTRA_MAX + (side_sel ? SB : SA));
    JTableAdam ta;
    memset(&ta, 0, sizeof(ta));
    if (adam) ta = *adam;
    hipLaunchKernelGGL(k_jaccum_p, dim3(n_acc_items < J_ACCP_GROUPS ? n_acc_items : J_ACCP_GROUPS), dim3(J_ACC_THREADS), 0, s, (uint32_t)ACC_EXTRA_MAX,
                       SA, SB, w.items, w.hdr, w.rec_e, w.rec_v, gradA, gradB, gradB16, overwrite, side_sel, ta, binA0, binA1, binB0, binB1
#ifdef J_ACC_TIMING
                       , (unsigned long long*)((char*)workspace + us_hashgrid_joint_workspace_bytes(a, b, n) - (size_t)J_ACCP_GROUPS * 32u)   // the last 64 KiB of the
#endif                                                                                                                                          // record planes: never reached
                       );
    US_CHECK_LAUNCH("us_hashgrid_bwd_joint");
    return US_OK;
}

extern "C" int us_hashgrid_bwd_joint(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA,
                                     const float* dL_dyB, int64_t n, float* gradA, float* gradB, int flags, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    return bwd_joint(a, b, x, dL_dyA, dL_dyB, n, gradA, gradB, flags, workspace, workspace_bytes, stream, false);
}

#ifdef US_EXPERIMENTS                    // measured-slower variants, kept buildable: tools/build_experiments.sh (include/unislam_hip_experiments.h)
extern "C" int us_hashgrid_bwd_joint_adam(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA, const float* dL_dyB,
                                          int64_t n, float* gradA, float* gradB, const us_table_adam_desc* adam, int flags, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    US_REQUIRE(adam && adam->pA && adam->mA && adam->vA && adam->pB && adam->mB && adam->vB && adam->step_dev, US_ERR_NULL, "us_hashgrid_bwd_joint_adam: NULL pointer");
    US_REQUIRE((flags & US_GRID_BWD_OVERWRITE) && (flags & US_GRID_BWD_DETERMINISTIC), US_ERR_CONFIG,
               "us_hashgrid_bwd_joint_adam: the optimiser step is applied by the one workgroup that owns an entry, once: needs US_GRID_BWD_OVERWRITE and "
               "US_GRID_BWD_DETERMINISTIC (no bin split over several workgroups) -- in the scan call as well");
    US_REQUIRE(((((uintptr_t)adam->pA) | ((uintptr_t)adam->mA) | ((uintptr_t)adam->vA) | ((uintptr_t)adam->pB) | ((uintptr_t)adam->mB) | ((uintptr_t)adam->vB)) & 7u) == 0 &&
               (((uintptr_t)adam->step_dev) & 7u) == 0, US_ERR_SHAPE, "us_hashgrid_bwd_joint_adam: tables, moments and step_dev must be 8-byte aligned");
    US_REQUIRE(n > 0, US_ERR_SHAPE, "us_hashgrid_bwd_joint_adam: n %lld (an empty batch has no sweep to carry the step: use the optimiser's own launch)", (long long)n);
    JTableAdam ta;
    ta.pA = adam->pA; ta.mA = adam->mA; ta.vA = adam->vA; ta.pB = adam->pB; ta.mB = adam->mB; ta.vB = adam->vB;
    ta.lrA = (float)adam->lrA; ta.lrB = (float)adam->lrB; ta.one_minus_b1 = (float)(1.0 - adam->beta1); ta.b2 = (float)adam->beta2;
    ta.one_minus_b2 = (float)(1.0 - adam->beta2); ta.eps = (float)adam->eps; ta.step_dev = adam->step_dev; ta.write_grad = adam->write_grad;
    return bwd_joint(a, b, x, dL_dyA, dL_dyB, n, gradA, gradB, flags, workspace, workspace_bytes, stream, false, 0, nullptr, &ta);
}

extern "C" int us_hashgrid_bwd_joint_part(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA, const float* dL_dyB,
                                          int64_t n, float* gradA, float* gradB, int flags, void* workspace, size_t workspace_bytes, int level_lo,
                                          int level_hi, int what, void* stream) {
    US_REQUIRE(a && level_lo >= 0 && level_lo < level_hi && level_hi <= (int)a->n_levels, US_ERR_SHAPE, "us_hashgrid_bwd_joint_part: levels [%d, %d)", level_lo, level_hi);
    US_REQUIRE(what >= 1 && what <= 3, US_ERR_CONFIG, "us_hashgrid_bwd_joint_part: what = %d (1 record pass, 2 accumulate pass, 3 both)", what);
    US_REQUIRE((flags & US_GRID_BWD_COUNTED) && (flags & US_GRID_BWD_SCANNED) && (flags & US_GRID_BWD_DETERMINISTIC) && (flags & US_GRID_BWD_OVERWRITE) &&
               !(flags & (US_GRID_BWD_ONLY_A | US_GRID_BWD_ONLY_B | US_GRID_BWD_RECORDS_READY)), US_ERR_CONFIG,
               "us_hashgrid_bwd_joint_part: a part of a pass whose counts and scans are in place (COUNTED | SCANNED), with unsplit bins "
               "(DETERMINISTIC: a level's bins are then its only items) in OVERWRITE mode, both grids");
    US_REQUIRE(n > 0, US_ERR_SHAPE, "us_hashgrid_bwd_joint_part: n %lld", (long long)n);
    return bwd_joint(a, b, x, dL_dyA, dL_dyB, n, gradA, gradB, flags, workspace, workspace_bytes, stream, false, 0, nullptr, nullptr, level_lo, level_hi, what);
}
#endif

extern "C" int us_hashgrid_bwd_joint_img(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA,
                                         const float* dL_dyB, int64_t n, float* gradA, float* gradB, uint16_t* gradB_bf16, int flags,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    US_REQUIRE(gradB_bf16 && ((uintptr_t)gradB_bf16 & 3u) == 0, US_ERR_NULL, "us_hashgrid_bwd_joint_img: gradB_bf16 is NULL or not 4-byte aligned");
    US_REQUIRE((flags & US_GRID_BWD_OVERWRITE) && (flags & US_GRID_BWD_DETERMINISTIC), US_ERR_CONFIG,
               "us_hashgrid_bwd_joint_img: the image is written where the table is (US_GRID_BWD_OVERWRITE) by the one workgroup that owns an entry "
               "(US_GRID_BWD_DETERMINISTIC: no bin split over several workgroups' float atomics) -- in the scan call as well");
    US_REQUIRE(!(flags & US_GRID_BWD_ONLY_A), US_ERR_CONFIG, "us_hashgrid_bwd_joint_img: the image belongs to grid B's accumulate pass");
    return bwd_joint(a, b, x, dL_dyA, dL_dyB, n, gradA, gradB, flags, workspace, workspace_bytes, stream, false, 0, gradB_bf16);
}

extern "C" int us_hashgrid_joint_scan(const us_grid_desc* a, const us_grid_desc* b, int64_t n, float* gradA, float* gradB, int flags,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    return bwd_joint(a, b, nullptr, nullptr, nullptr, n, gradA, gradB, (flags | US_GRID_BWD_COUNTED | US_GRID_LEVEL_MAJOR) & ~US_GRID_BWD_SCANNED,
                     workspace, workspace_bytes, stream, true);
}

extern "C" int us_hashgrid_bwd_joint_range(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA,
                                           const float* dL_dyB, int64_t n, int64_t plane_stride, float* gradA, float* gradB, int flags,
                                           void* workspace, size_t workspace_bytes, void* stream) {
    US_REQUIRE(plane_stride >= n, US_ERR_SHAPE, "us_hashgrid_bwd_joint_range: plane_stride %lld < n %lld", (long long)plane_stride, (long long)n);
    US_REQUIRE(!(flags & (US_GRID_BWD_COUNTED | US_GRID_BWD_SCANNED)), US_ERR_CONFIG,
               "us_hashgrid_bwd_joint_range: the counts of a forward pass belong to the whole batch, not to a range of it");
    return bwd_joint(a, b, x, dL_dyA, dL_dyB, n, gradA, gradB, flags, workspace, workspace_bytes, stream, false, plane_stride);
}
