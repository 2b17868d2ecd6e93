// hashgrid_binned.hip -- table gradient of the hash grid by "bin once, accumulate in f64" (gfx950).
//
// Why (all measured on MI355X, tools/lds_atomic_bench.hip, tools/bwd_bench.hip):
//   * memory-side float atomics serve ~20 G requests/s chip-wide: 67 M scattered lane-atomics per grid = 4.8 ms;
//   * privatising table slices in LDS and re-hashing every point once per slice (hashgrid.hip k_bwd_sliced) is bound by
//     ds_add_f32: ~3 cycles per ACTIVE LANE (195 cycles per 64-lane instruction), ~18 per lane on equal addresses, and
//     consecutive samples of a ray do fall into the same cells; the 4 MiB colour levels also need 32 slices = 32x re-hashing;
//   * ds_add_f64 runs at 21 cycles per 64 lanes (9x faster than f32), ds_add_u32 at 7.
// So each (point, level) is hashed a constant number of times instead of once per slice:
//   A0 k_bin<COUNT>   every workgroup takes 1024 consecutive points through ALL levels.  Per level, lanes whose neighbours
//                     (inside aligned groups of 8 lanes = 8 consecutive samples of one ray) sit in the same grid cell are
//                     RUN-COMBINED with DPP row shifts (segmented inclusive scan of the 8 corner contributions), so a run
//                     emits one record per corner instead of one per sample.  Records are counted per BIN, where
//                     bin = entry index mod n_bins(level): interleaving makes bins equally loaded whatever the geometry.
//                     The counts are kept PER WORKGROUP: row w of a [workgroups][bins] matrix.
//   scan              column scan over the workgroups + exclusive scan of the bin totals -> every (workgroup, bin) pair owns
//                     an exact, gap-free record range (no over-allocation, no overflow, no cursor to reserve from).
//   A1 k_bin<WRITE>   same pass again with the LDS counters preloaded with the workgroup's range starts: an LDS integer
//                     atomic returns the record's final position, so the pass has no barrier, no global atomic and no
//                     dependence between levels; it writes {local entry, F values} records.
//   B  k_bin_accum    one 256-thread workgroup per bin: 32 KiB of f64 accumulators in LDS (thousands of bins, ~5 resident
//                     workgroups per CU, 8 record loads in flight per thread), ds_add_f64 over the bin's records, then a
//                     plain (non-atomic) add into the gradient table: a bin owns its entries exclusively.
// Sums are formed in double precision and in a fixed order of records per bin up to the order of LDS atomics; no global atomics.
#include "hashgrid_dev.h"
#include <string.h>

#define BIN_THREADS 1024                 // binning kernels: 1024 consecutive points per workgroup
#ifndef BIN_MAX_TOTAL
#define BIN_MAX_TOTAL 4096               // bins over all levels (LDS counters of the binning passes: 16 KiB)
#endif
#ifndef BIN_ACC_DOUBLES
#define BIN_ACC_DOUBLES 4096             // 32 KiB of f64 accumulators per bin -> ~5 accumulate workgroups per CU
#endif
// accumulate kernel shape (tools/stream_bench.hip: 512 threads x 4 records in flight stream 12-byte records through 32 KiB
// of f64 LDS atomics at 4.9 TB/s; 256 x 8 reaches 3.8, the LDS footprint caps the resident waves)
#define ACC_THREADS 512
#define ACC_UNROLL 4
#define ACC_CHUNK 16384                  // records per accumulate workgroup: hotter bins are split (see k_bin_accum)
#define BIN_LINE_LOG2 4                  // bins interleave LINES of 16 entries (128 B of F = 2 gradients)

struct BinMap {
    uint32_t first[US_MAX_LEVELS + 1];   // prefix sum of bins per level
    uint32_t log2nb[US_MAX_LEVELS];      // bins per level = 1 << log2nb
};

// bin = (entry / 16) mod n_bins: a bin owns every n_bins-th 128-byte line of the level's gradient slab.  Interleaving spreads
// the hot places of the geometry (dense levels) and of the hash (whose high bits depend on y, z only) over all bins; whole
// lines keep the final sweep of the accumulate kernel coalesced.
__device__ __forceinline__ uint32_t bin_of(uint32_t e, uint32_t lg) { return (e >> BIN_LINE_LOG2) & ((1u << lg) - 1u); }
__device__ __forceinline__ uint32_t local_of(uint32_t e, uint32_t lg) {
    return ((e >> (BIN_LINE_LOG2 + lg)) << BIN_LINE_LOG2) | (e & ((1u << BIN_LINE_LOG2) - 1u));
}
__device__ __forceinline__ uint32_t entry_of(uint32_t loc, uint32_t bl, uint32_t lg) {
    return ((loc >> BIN_LINE_LOG2) << (BIN_LINE_LOG2 + lg)) | (bl << BIN_LINE_LOG2) | (loc & ((1u << BIN_LINE_LOG2) - 1u));
}
// local entries (multiple of 16; the last line of the slab may be partial) owned by bin bl of a level with hs entries
__host__ __device__ __forceinline__ uint32_t bin_n_local(uint32_t hs, uint32_t bl, uint32_t lg) {
    const uint32_t lines = (hs + (1u << BIN_LINE_LOG2) - 1u) >> BIN_LINE_LOG2;
    return bl < lines ? ((((lines - 1u - bl) >> lg) + 1u) << BIN_LINE_LOG2) : 0u;
}

static inline uint32_t bin_entries(uint32_t F) { return BIN_ACC_DOUBLES / F; }

// records are {local entry, F values} = 1+F dwords.  Measured: the scatter pass's stores run at HBM write speed (padding the
// F = 2 record to 16 bytes for single vector stores made the whole pass slower: bytes matter, not store instructions)
template <int F> struct RecW { static constexpr int DW = 1 + F; };
static inline uint32_t rec_dwords(uint32_t F) { return 1u + F; }

// bins per level: enough for the f64 slice to fit the LDS budget (capacity) AND enough to keep every bin near
// BIN_TARGET_RECORDS records whatever the level's size (a 4096-entry level receives as many records as a 4 MiB one)
#ifndef BIN_TARGET_RECORDS
#define BIN_TARGET_RECORDS 8192
#endif
#ifndef BIN_WANT_MAX
#define BIN_WANT_MAX 8
#endif
static int make_binmap(const us_grid_desc* d, int64_t n, BinMap* bm) {
    uint32_t total = 0;
    const uint32_t be = bin_entries(d->n_features);
    uint32_t want = 0;
    while (((int64_t)BIN_TARGET_RECORDS << want) < n * 8 && want < BIN_WANT_MAX) ++want;
    for (uint32_t l = 0; l < US_MAX_LEVELS; ++l) {
        bm->first[l] = total; bm->log2nb[l] = 0;
        if (l >= d->n_levels) continue;
        const uint32_t hs = d->offset[l + 1] - d->offset[l];
        const uint32_t lines = (hs + (1u << BIN_LINE_LOG2) - 1u) >> BIN_LINE_LOG2;
        uint32_t lg = 0;
        while (bin_n_local(hs, 0, lg) > be) ++lg;                // capacity of the f64 slice
        if (lg < want) lg = want;                                // load
        while (lg > 0 && (1u << lg) > lines) --lg;               // never more bins than lines
        if (bin_n_local(hs, 0, lg) > be) return -1;
        bm->log2nb[l] = lg;
        total += 1u << lg;
    }
    bm->first[US_MAX_LEVELS] = total;
    for (uint32_t l = d->n_levels; l <= US_MAX_LEVELS; ++l) bm->first[l] = total;
    return (int)total;
}

// ---- DPP row shifts (within rows of 16 lanes): shr: lane i <- lane i-n ; shl: lane i <- lane i+n
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v)));
}
#define DPP_ROW_SHL1 0x101
#define DPP_ROW_SHR(n) (0x110 + (n))

// one level of one point: corner records after run-combining.  emit[c] says whether this lane owns a record for corner c.
template <int F>
struct LevelRecords {
    float val[8][F];
    uint32_t idx[8];
    bool tail;
};

// COUNT_ONLY: the counting pass needs only WHICH lanes own a record (run tails of live samples) and the 8 entry indices;
// it counts every corner of every live tail (exactly what the writing pass emits), so it skips the weights, products and scan.
template <int F, bool COUNT_ONLY>
__device__ __forceinline__ void level_records(const LevelGeom& g, const float xv[3], const float dy[F], bool live, int lane,
                                              LevelRecords<F>& r) {
    float pos[3]; uint32_t cell[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pos_fract(xv[k], g.scale, pos[k], cell[k]);
    // run key: the cell (resolutions <= 1023 per axis; larger grids get unique keys = no combining)
#ifdef US_EXP_NO_COMBINE
    const bool packable = false;
#else
    const bool packable = g.res <= 1023u;
#endif
    uint32_t key = (cell[0] & 1023u) | ((cell[1] & 1023u) << 10) | ((cell[2] & 1023u) << 20);
    if (!live || !packable) key = 0xC0000000u | (uint32_t)lane;      // bits 30..31 set: never equals a packed cell, unique per lane
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (!COUNT_ONLY) {
            const float w = corner_weight(c, pos);
#pragma unroll
            for (int f = 0; f < F; ++f) r.val[c][f] = live ? w * dy[f] : 0.0f;
        }
        r.idx[c] = grid_index(g, cell[0] + (c & 1), cell[1] + ((c >> 1) & 1), cell[2] + ((c >> 2) & 1));
    }
    // Segmented inclusive scan over aligned groups of 8 lanes (Hillis-Steele with head flags, steps 1, 2, 4).  A run is a
    // maximal stretch of ADJACENT lanes in the same cell; equal cells that are not adjacent (arbitrary point order) stay
    // separate runs, so the scan is exact for any input, and simply finds nothing to merge on unordered points.
    const int l8 = lane & 7;
    // NB: every DPP move must execute with the whole wave active (a lane disabled by EXEC reads as 0 to its neighbours):
    // hoist them out of any short-circuit / divergent expression.
    const uint32_t kprev = dpp_u32<DPP_ROW_SHR(1)>(key), knext = dpp_u32<DPP_ROW_SHL1>(key);
    bool flag = (l8 == 0) | (kprev != key);                                  // head of a run
    const bool next_is_head = (l8 == 7) | (knext != key);
#define US_SCAN_STEP(O)                                                                                              \
    if (!COUNT_ONLY && __ballot(!flag && (l8 >= (O))) != 0ull) {   /* wave-uniform: nothing left to merge -> skip the step */ \
        const bool take = !flag && (l8 >= (O));                                                                      \
        const bool fprev = dpp_u32<DPP_ROW_SHR(O)>(flag ? 1u : 0u) != 0u;                                            \
        _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                               \
            _Pragma("unroll") for (int f = 0; f < F; ++f) {                                                         \
                const float t = dpp_f32<DPP_ROW_SHR(O)>(r.val[c][f]);                                                \
                r.val[c][f] += take ? t : 0.0f;                                                                      \
            }                                                                                                        \
        flag = flag | ((l8 >= (O)) & fprev);                                                                         \
    }
    US_SCAN_STEP(1)
    US_SCAN_STEP(2)
    US_SCAN_STEP(4)
#undef US_SCAN_STEP
    r.tail = live & next_is_head;
}

// A0 (WRITE = false): wg_counts[workgroup][bin] = records this workgroup will emit (every corner of every live run tail).
// A1 (WRITE = true) : LDS counters start at the workgroup's range starts (bin offset + column prefix); the LDS atomic of
//                     each record returns its final slot, and the record is stored at once.
template <int F, bool WRITE>
__global__ __launch_bounds__(BIN_THREADS) void k_bin(LevelTable tab, BinMap bm, uint32_t n_levels, const float* __restrict__ x,
                                                     const float* __restrict__ dL_dy, int64_t n, int clamp, int lm,
                                                     uint32_t* __restrict__ wg_counts, const uint32_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ rec) {
    __shared__ uint32_t lcnt[BIN_MAX_TOTAL];
    const uint32_t TB = bm.first[n_levels];
    uint32_t* row = wg_counts + (size_t)blockIdx.x * BIN_MAX_TOTAL;
    for (uint32_t t = threadIdx.x; t < TB; t += BIN_THREADS) lcnt[t] = WRITE ? offsets[t] + row[t] : 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * BIN_THREADS + threadIdx.x;
    const bool in = i < n;
    const uint32_t C = n_levels * F;
    float xv[3] = {0.f, 0.f, 0.f};
    if (in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) xv[k] = load_x(x, i, k, clamp);
    }
    // all levels' gradients of this point are fetched up front (one memory round trip instead of one per level)
    constexpr int LCH = 16;
    float dyv[LCH][F];
    for (uint32_t level = 0; level < n_levels; ++level) {
        if ((level % LCH) == 0) {
#pragma unroll
            for (int q = 0; q < LCH; ++q)
#pragma unroll
                for (int f = 0; f < F; ++f)
                    dyv[q][f] = (in && level + q < n_levels) ? dL_dy[feat_index(lm, i, n, level + q, C, F) + f] : 0.0f;
        }
        const LevelGeom g = level_geom(tab, level);
        const uint32_t lg = bm.log2nb[level], first = bm.first[level];
        float dy[F]; bool live = false;
#pragma unroll
        for (int f = 0; f < F; ++f) dy[f] = 0.0f;
#pragma unroll
        for (int q = 0; q < LCH; ++q)                           // static register indexing (level % LCH is wave-uniform)
            if ((int)(level % LCH) == q) {
#pragma unroll
                for (int f = 0; f < F; ++f) { dy[f] = dyv[q][f]; live |= (dy[f] != 0.0f); }
            }
        if (__ballot(live) == 0ull) continue;                    // a wave whose samples all have zero gradient skips the level
        LevelRecords<F> r;
        level_records<F, !WRITE>(g, xv, dy, live, lane, r);
        // Both passes emit EVERY corner of every live run tail (a corner whose weight is exactly 0 becomes a zero record),
        // so the counted ranges are exact.
        const bool e = r.tail;
        const unsigned long long mask = __ballot(e);
        if (mask == 0ull) continue;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t b = first + bin_of(r.idx[c], lg);
            uint32_t slot = 0;
            // With few bins (small batches) every emitting lane of the wave tends to hit the same bin: one LDS atomic for the
            // wave instead of <= 64 serialised ones.  With hundreds of bins per level the lanes scatter and plain LDS integer
            // atomics (7 cycles per wave instruction) are cheapest.
            bool done = false;
            if (lg <= 1) {                                       // wave-uniform
                const int lead = __ffsll((long long)mask) - 1;
                const uint32_t b0 = __builtin_amdgcn_readlane(b, lead);
                if (__ballot(e && b == b0) == mask) {
                    uint32_t base = 0;
                    const uint32_t mb = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    if (e && mb == 0) base = atomicAdd(&lcnt[b0], (uint32_t)__popcll(mask));
                    base = __builtin_amdgcn_readlane(base, lead);
                    slot = base + mb;
                    done = true;
                }
            }
            if (!done && e) slot = atomicAdd(&lcnt[b], 1u);
            if (WRITE && e) {
                uint32_t* dst = rec + (size_t)slot * RecW<F>::DW;
                dst[0] = local_of(r.idx[c], lg);
#pragma unroll
                for (int f = 0; f < F; ++f) dst[1 + f] = __float_as_uint(r.val[c][f]);
            }
        }
    }
    if (!WRITE) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < TB; t += BIN_THREADS) row[t] = lcnt[t];
    }
}

// column scan: wg_counts[w][b] <- sum of wg_counts[w'][b] over w' < w ; totals[b] = column sum.
// In OVERWRITE mode the bins that will be split over several accumulate workgroups (total > ACC_CHUNK, added with float
// atomics) get their gradient entries cleared here, two kernels ahead of the first add.
#define COLSCAN_THREADS 64
template <int F>
__global__ __launch_bounds__(COLSCAN_THREADS) void k_bin_colscan(LevelTable tab, BinMap bm, uint32_t n_levels, uint32_t* __restrict__ wg_counts,
                                                                 uint32_t n_wg, uint32_t TB, uint32_t* __restrict__ totals,
                                                                 float* __restrict__ grad, int overwrite) {
    const uint32_t b = blockIdx.x * COLSCAN_THREADS + threadIdx.x;
    uint32_t run = 0;
    if (b < TB) {
        uint32_t w = 0;
        for (; w + 8 <= n_wg; w += 8) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = wg_counts[(size_t)(w + k) * BIN_MAX_TOTAL + b];
#pragma unroll
            for (int k = 0; k < 8; ++k) { wg_counts[(size_t)(w + k) * BIN_MAX_TOTAL + b] = run; run += v[k]; }
        }
        for (; w < n_wg; ++w) { const uint32_t v = wg_counts[(size_t)w * BIN_MAX_TOTAL + b]; wg_counts[(size_t)w * BIN_MAX_TOTAL + b] = run; run += v; }
        totals[b] = run;
    }
    if (!overwrite) return;
    unsigned long long hot = __ballot(b < TB && run > ACC_CHUNK);
    while (hot) {                                                // wave-uniform loop (one wave per workgroup)
        const int src = __ffsll((long long)hot) - 1;
        hot &= hot - 1ull;
        const uint32_t hb = blockIdx.x * COLSCAN_THREADS + (uint32_t)src;
        uint32_t level = 0;
        for (uint32_t l = 1; l < n_levels; ++l) level += (bm.first[l] <= hb) ? 1u : 0u;
        const uint32_t lg = bm.log2nb[level], bl = hb - bm.first[level], hs = tab.off[level + 1] - tab.off[level];
        const uint32_t n_local = bin_n_local(hs, bl, lg);
        float* gl = grad + (size_t)tab.off[level] * F;
        for (uint32_t loc = threadIdx.x; loc < n_local; loc += COLSCAN_THREADS) {
            const uint32_t e = entry_of(loc, bl, lg);
            if (e < hs) {
#pragma unroll
                for (int f = 0; f < F; ++f) gl[(size_t)e * F + f] = 0.0f;
            }
        }
    }
}

// exclusive scan of counts[0..TB) -> offsets[0..TB]   (TB <= 4096: 4 elements per thread), and the list of EXTRA chunks:
// a bin with c > ACC_CHUNK records is accumulated by ceil(c / ACC_CHUNK) workgroups; chunk 0 belongs to the bin's own
// workgroup, chunks 1.. are listed in extra[] as bin | chunk << 16.
__global__ __launch_bounds__(1024) void k_bin_scan(const uint32_t* __restrict__ counts, uint32_t TB, uint32_t* __restrict__ offsets,
                                                   uint32_t* __restrict__ extra, uint32_t* __restrict__ n_extra) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t sx[1024];
    const uint32_t t = threadIdx.x;
    uint32_t c[4], s4 = 0, x4 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = (4 * t + k < TB) ? counts[4 * t + k] : 0u; s4 += c[k]; x4 += c[k] > ACC_CHUNK ? (c[k] - 1u) / ACC_CHUNK : 0u; }
    sh[t] = s4; sx[t] = x4;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t v = (t >= o) ? sh[t - o] : 0u, w = (t >= o) ? sx[t - o] : 0u;
        __syncthreads();
        sh[t] += v; sx[t] += w;
        __syncthreads();
    }
    uint32_t run = sh[t] - s4, xrun = sx[t] - x4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (4 * t + k < TB) offsets[4 * t + k] = run;
        run += c[k];
        const uint32_t nx = c[k] > ACC_CHUNK ? (c[k] - 1u) / ACC_CHUNK : 0u;
        for (uint32_t j = 0; j < nx; ++j) extra[xrun + j] = (4 * t + k) | ((j + 1u) << 16);
        xrun += nx;
    }
    if (t == 1023) { offsets[TB] = sh[1023]; *n_extra = sx[1023]; }
}

// B: one workgroup per bin (+ one per extra chunk of a hot bin; those come FIRST in the grid: they are the longest jobs)
template <int F>
__global__ __launch_bounds__(ACC_THREADS) void k_bin_accum(LevelTable tab, BinMap bm, uint32_t n_levels, uint32_t e_max,
                                                           const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ extra,
                                                           const uint32_t* __restrict__ n_extra,
                                                           const uint32_t* __restrict__ rec, float* __restrict__ grad, int overwrite) {
    __shared__ double acc[BIN_ACC_DOUBLES];
    uint32_t b, chunk = 0;
    if (blockIdx.x < e_max) {
        if (blockIdx.x >= *n_extra) return;
        const uint32_t pk = extra[blockIdx.x];
        b = pk & 0xFFFFu; chunk = pk >> 16;
    } else {
        b = blockIdx.x - e_max;
    }
    uint32_t level = 0;
    for (uint32_t l = 1; l < n_levels; ++l) level += (bm.first[l] <= b) ? 1u : 0u;
    const uint32_t lg = bm.log2nb[level], bl = b - bm.first[level];
    const uint32_t hs = tab.off[level + 1] - tab.off[level];
    const uint32_t n_local = bin_n_local(hs, bl, lg);                 // entries owned by this bin
    const uint32_t b0 = offsets[b], b1 = offsets[b + 1];
    const bool split = (b1 - b0) > ACC_CHUNK;                         // several workgroups add into this bin's entries
    const uint32_t r0 = b0 + chunk * ACC_CHUNK;
    const uint32_t r1 = (b1 - r0 > ACC_CHUNK) ? r0 + ACC_CHUNK : b1;
    float* gl = grad + (size_t)tab.off[level] * F;
    if (b0 == b1) {                                                   // nothing landed in this bin (wave-uniform)
        if (overwrite)
            for (uint32_t loc = threadIdx.x; loc < n_local; loc += ACC_THREADS) {
                const uint32_t e = entry_of(loc, bl, lg);
                if (e < hs) {
#pragma unroll
                    for (int f = 0; f < F; ++f) gl[(size_t)e * F + f] = 0.0f;
                }
            }
        return;
    }
    for (uint32_t k = threadIdx.x; k < n_local * F; k += ACC_THREADS) acc[k] = 0.0;
    __syncthreads();
    for (uint32_t base = r0; base < r1; base += ACC_THREADS * ACC_UNROLL) {
        uint32_t loc[ACC_UNROLL]; float v[ACC_UNROLL][F];
#pragma unroll
        for (int u = 0; u < ACC_UNROLL; ++u) {                   // issue every load of the group before the first use
            const uint32_t r = base + u * ACC_THREADS + threadIdx.x;
            loc[u] = 0xFFFFFFFFu;
            if (r < r1) {
                const uint32_t* src = rec + (size_t)r * RecW<F>::DW;
                loc[u] = src[0];
#pragma unroll
                for (int f = 0; f < F; ++f) v[u][f] = __uint_as_float(src[1 + f]);
            }
        }
#pragma unroll
        for (int u = 0; u < ACC_UNROLL; ++u) {
            if (loc[u] != 0xFFFFFFFFu) {
#pragma unroll
                for (int f = 0; f < F; ++f) atomicAdd(&acc[loc[u] * F + f], (double)v[u][f]);            // ds_add_f64
            }
        }
    }
    __syncthreads();
    for (uint32_t loc = threadIdx.x; loc < n_local; loc += ACC_THREADS) {
        const uint32_t e = entry_of(loc, bl, lg);
        if (e >= hs) continue;
        float v[F]; bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { v[f] = (float)acc[loc * F + f]; any |= (v[f] != 0.0f); }
        float* p = gl + (size_t)e * F;
        if (split) {
#pragma unroll
            for (int f = 0; f < F; ++f) if (v[f] != 0.0f) atomicAdd(p + f, v[f]);
        } else if (overwrite) {
            typename Feat<F>::T o;
            array_to_feat<F>(v, o);
            *reinterpret_cast<typename Feat<F>::T*>(p) = o;
        } else if (any) {                                        // this workgroup is the only writer of its entries
            typename Feat<F>::T o = *reinterpret_cast<const typename Feat<F>::T*>(p);
            float cur[F];
            feat_to_array<F>(o, cur);
#pragma unroll
            for (int f = 0; f < F; ++f) cur[f] += v[f];
            array_to_feat<F>(cur, o);
            *reinterpret_cast<typename Feat<F>::T*>(p) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
// workspace: totals | offsets (2 x (BIN_MAX_TOTAL + 64) u32) | n_extra (16 u32) | extra[e_max] | per-workgroup count rows
// [n_wg][BIN_MAX_TOTAL] | records
static uint32_t extra_max(const us_grid_desc* d, int64_t n) { return (uint32_t)(((uint64_t)n * 8ull * d->n_levels) / ACC_CHUNK) + 1u; }
static size_t header_bytes(const us_grid_desc* d, int64_t n) {
    const size_t em = ((size_t)extra_max(d, n) + 15u) & ~(size_t)15u;
    return (size_t)(2 * (BIN_MAX_TOTAL + 64) + 16 + em) * sizeof(uint32_t) + (size_t)us_cdiv(n, BIN_THREADS) * BIN_MAX_TOTAL * sizeof(uint32_t);
}

extern "C" size_t us_hashgrid_bwd_workspace_bytes(const us_grid_desc* d, int64_t n) {
    if (!d || n <= 0) return 0;
    return header_bytes(d, n) + (size_t)n * 8u * d->n_levels * rec_dwords(d->n_features) * sizeof(uint32_t);
}

extern "C" int us_hashgrid_bwd_binned(const us_grid_desc* d, const float* x, const float* dL_dy, int64_t n, float* grad_params,
                                      int flags, void* workspace, size_t workspace_bytes, void* stream) {
    US_REQUIRE(d, US_ERR_NULL, "us_hashgrid_bwd_binned: desc is NULL");
    US_REQUIRE(d->n_levels >= 1 && d->n_levels <= US_MAX_LEVELS && (d->n_features == 1 || d->n_features == 2 || d->n_features == 4) &&
               d->n_params == d->offset[d->n_levels] * d->n_features, US_ERR_CONFIG, "us_hashgrid_bwd_binned: bad descriptor");
    if (n < 0) return US_ERR_SHAPE;
    if (n == 0) {                                                // no samples: the gradient is zero
        if ((flags & US_GRID_BWD_OVERWRITE) && grad_params) {
            hipError_t e = hipMemsetAsync(grad_params, 0, (size_t)d->n_params * sizeof(float), (hipStream_t)stream);
            if (e != hipSuccess) { us_set_error("us_hashgrid_bwd_binned: memset: %s", hipGetErrorString(e)); return (int)e; }
        }
        return US_OK;
    }
    US_REQUIRE(x && dL_dy && grad_params && workspace, US_ERR_NULL, "us_hashgrid_bwd_binned: NULL pointer");
    US_REQUIRE(((uintptr_t)grad_params & 15u) == 0 && ((uintptr_t)workspace & 15u) == 0, US_ERR_SHAPE,
               "us_hashgrid_bwd_binned: grad_params and workspace must be 16-byte aligned");
    US_REQUIRE((uint64_t)n * 8ull * d->n_levels < 0xFFFFFFFFull, US_ERR_SHAPE, "us_hashgrid_bwd_binned: n too large for 32-bit record ranks");
    US_REQUIRE(workspace_bytes >= us_hashgrid_bwd_workspace_bytes(d, n), US_ERR_WORKSPACE,
               "us_hashgrid_bwd_binned: workspace %zu B < %zu B", workspace_bytes, us_hashgrid_bwd_workspace_bytes(d, n));
    BinMap bm;
    const int TB = make_binmap(d, n, &bm);
    US_REQUIRE(TB > 0 && TB <= BIN_MAX_TOTAL, US_ERR_CONFIG, "us_hashgrid_bwd_binned: %d bins > %d (table too large for this path)", TB, BIN_MAX_TOTAL);
    const LevelTable t = make_table(d);
    hipStream_t s = (hipStream_t)stream;
    const int clamp = flags & US_GRID_CLAMP01, lm = (flags & US_GRID_LEVEL_MAJOR) ? 1 : 0;
    const uint32_t e_max = extra_max(d, n);
    uint32_t* totals = (uint32_t*)workspace;
    uint32_t* offsets = totals + (BIN_MAX_TOTAL + 64);
    uint32_t* n_extra = offsets + (BIN_MAX_TOTAL + 64);
    uint32_t* extra = n_extra + 16;
    uint32_t* wg_counts = extra + (((size_t)e_max + 15u) & ~(size_t)15u);
    uint32_t* rec = (uint32_t*)((char*)workspace + header_bytes(d, n));
    const uint32_t n_wg = (uint32_t)us_cdiv(n, BIN_THREADS);
    dim3 gridA(n_wg), block(BIN_THREADS);
    const uint32_t L = d->n_levels;
    const int overwrite = (flags & US_GRID_BWD_OVERWRITE) ? 1 : 0;
    // (Splitting the levels into groups of ~100 MB of records, so that the accumulate pass would read them from the Infinity
    //  Cache, was measured SLOWER: 0.46 vs 0.36 ms per grid -- the fixed costs of four more passes outweigh the cache hits.)
#define LAUNCH_BIN(F)                                                                                                          \
    hipLaunchKernelGGL((k_bin<F, false>), gridA, block, 0, s, t, bm, L, x, dL_dy, n, clamp, lm, wg_counts, offsets, rec);      \
    hipLaunchKernelGGL((k_bin_colscan<F>), dim3(us_cdiv(TB, COLSCAN_THREADS)), dim3(COLSCAN_THREADS), 0, s, t, bm, L, wg_counts, n_wg, \
                       (uint32_t)TB, totals, grad_params, overwrite);                                                          \
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, totals, (uint32_t)TB, offsets, extra, n_extra);                  \
    hipLaunchKernelGGL((k_bin<F, true>), gridA, block, 0, s, t, bm, L, x, dL_dy, n, clamp, lm, wg_counts, offsets, rec);       \
    hipLaunchKernelGGL((k_bin_accum<F>), dim3(e_max + TB), dim3(ACC_THREADS), 0, s, t, bm, L, e_max, offsets, extra, n_extra, rec, \
                       grad_params, overwrite);
    switch (d->n_features) { case 1: LAUNCH_BIN(1) break; case 2: LAUNCH_BIN(2) break; default: LAUNCH_BIN(4) break; }
#undef LAUNCH_BIN
    US_CHECK_LAUNCH("us_hashgrid_bwd_binned");
    return US_OK;
}
