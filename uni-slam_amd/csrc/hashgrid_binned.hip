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
//   scan              exclusive scan of the bin counts -> exact, gap-free record ranges (no over-allocation, no overflow).
//   A1 k_bin<WRITE>   same pass again, now writing {local entry, F values} records into their bin's range.
//   B  k_bin_accum    one 256-thread workgroup per bin: 32 KiB of f64 accumulators in LDS (thousands of bins, ~5 resident
//                     workgroups per CU, 8 record loads in flight per thread), ds_add_f64 over the bin's records, then a
//                     plain (non-atomic) add into the gradient table: a bin owns its entries exclusively.
// Sums are formed in double precision; the only global atomics left are the per-workgroup bin-cursor reservations.
#include "hashgrid_dev.h"
#include <string.h>

#define BIN_THREADS 1024                 // binning kernels: 1024 consecutive points per workgroup
#define BIN_MAX_TOTAL 4096               // bins over all levels (LDS counters: 2 x 16 KiB)
#define BIN_ACC_DOUBLES 4096             // 32 KiB of f64 accumulators per bin -> ~5 accumulate workgroups per CU
#define ACC_THREADS 256
#define BIN_GROUP_BYTES (1ull << 40)      // record bytes per level group: effectively ONE group (see us_hashgrid_bwd_binned)
#define ACC_UNROLL 8                     // record loads in flight per thread (the accumulate kernel is a latency-bound stream)

struct BinMap {
    uint32_t first[US_MAX_LEVELS + 1];   // prefix sum of bins per level
    uint8_t  log2nb[US_MAX_LEVELS];      // bins per level = 1 << log2nb
    uint8_t  shift[US_MAX_LEVELS];       // > 0: BLOCKED bins (bin = entry >> shift, local = entry & mask): hashed levels, whose
                                         //      entries are already uniformly loaded -> the slice is a contiguous table range
                                         // = 0: INTERLEAVED bins (bin = entry & (nb-1), local = entry >> log2nb): dense levels,
                                         //      where geometry concentrates the hits in a few places
};

__device__ __forceinline__ uint32_t bin_of(uint32_t e, uint32_t lg, uint32_t sh) { return sh ? (e >> sh) : (e & ((1u << lg) - 1u)); }
__device__ __forceinline__ uint32_t local_of(uint32_t e, uint32_t lg, uint32_t sh) { return sh ? (e & ((1u << sh) - 1u)) : (e >> lg); }

static inline uint32_t bin_entries(uint32_t F) { return BIN_ACC_DOUBLES / F; }

// records are {local entry, F values} = 1+F dwords.  Measured: the scatter pass's stores run at HBM write speed (padding the
// F = 2 record to 16 bytes for single vector stores made the whole pass slower: bytes matter, not store instructions)
template <int F> struct RecW { static constexpr int DW = 1 + F; };
static inline uint32_t rec_dwords(uint32_t F) { return 1u + F; }

// bins per level: enough for the f64 slice to fit the LDS budget (capacity) AND enough to keep every bin near
// BIN_TARGET_RECORDS records whatever the level's size (a 4096-entry level receives as many records as a 4 MiB one)
#define BIN_TARGET_RECORDS 8192
static int make_binmap(const us_grid_desc* d, int64_t n, BinMap* bm) {
    uint32_t total = 0;
    const uint32_t be = bin_entries(d->n_features);
    uint32_t want = 0;
    while (((int64_t)BIN_TARGET_RECORDS << want) < n * 8 && want < 8) ++want;
    for (uint32_t l = 0; l < US_MAX_LEVELS; ++l) {
        bm->first[l] = total; bm->log2nb[l] = 0; bm->shift[l] = 0;
        if (l >= d->n_levels) continue;
        const uint32_t hs = d->offset[l + 1] - d->offset[l];
        uint32_t lg = 0;
        while (((uint64_t)be << lg) < hs) ++lg;                  // capacity
        if (lg < want) lg = want;                                // load
        while (lg > 0 && (1u << lg) > hs) --lg;                  // never more bins than entries
        bm->log2nb[l] = (uint8_t)lg;
        // hashed level <=> the slab holds exactly 2^log2T entries and the dense grid would not fit (tcnn grid_index)
        const uint64_t res = d->resolution[l];
        const bool hashed = (hs == (1u << d->log2_hashmap_size)) && (res * res * res > hs);
        uint32_t sh = 0;
        if (hashed && lg > 0) { while ((1u << (sh + lg)) < hs) ++sh; }       // hs is a power of two here: hs >> lg entries per bin
        bm->shift[l] = (uint8_t)sh;
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
// it counts every corner of every live tail (an upper bound: the writing pass drops all-zero records, and the accumulate
// kernel reads exactly what was written, [offset, cursor) per bin), so it skips the weights, products and the scan.
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

// A0 (WRITE = false): counts[bin] += records of this workgroup (one flush at the end).
// A1 (WRITE = true) : per level: count in LDS -> reserve the workgroup's share of every bin from the global cursors
//                     (initialised to the scan) -> store the records, which stayed in registers meanwhile.
template <int F, bool WRITE>
__global__ __launch_bounds__(BIN_THREADS) void k_bin(LevelTable tab, BinMap bm, uint32_t n_levels, uint32_t l0, uint32_t l1,
                                                     const float* __restrict__ x,
                                                     const float* __restrict__ dL_dy, int64_t n, int clamp, int lm,
                                                     uint32_t* __restrict__ counts, uint32_t* __restrict__ cursors,
                                                     uint32_t* __restrict__ rec) {
    __shared__ uint32_t lcnt[BIN_MAX_TOTAL];
    const uint32_t TB0 = bm.first[l0], TB1 = bm.first[l1];       // this launch handles levels [l0, l1) = bins [TB0, TB1)
    for (uint32_t t = TB0 + threadIdx.x; t < TB1; t += BIN_THREADS) lcnt[t] = 0;
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
    for (uint32_t level = l0; level < l1; ++level) {
        if (level == l0 || (level % LCH) == 0) {
            const uint32_t lb = level - (level % LCH);
#pragma unroll
            for (int q = 0; q < LCH; ++q)
#pragma unroll
                for (int f = 0; f < F; ++f)
                    dyv[q][f] = (in && lb + q >= l0 && lb + q < l1) ? dL_dy[feat_index(lm, i, n, lb + q, C, F) + f] : 0.0f;
        }
        const LevelGeom g = level_geom(tab, level);
        const uint32_t nb = 1u << bm.log2nb[level], lg = bm.log2nb[level], first = bm.first[level], sh = bm.shift[level];
        float dy[F]; bool live = false;
#pragma unroll
        for (int f = 0; f < F; ++f) dy[f] = 0.0f;
#pragma unroll
        for (int q = 0; q < LCH; ++q)                           // static register indexing (level % LCH is wave-uniform)
            if ((int)(level % LCH) == q) {
#pragma unroll
                for (int f = 0; f < F; ++f) { dy[f] = dyv[q][f]; live |= (dy[f] != 0.0f); }
            }
        LevelRecords<F> r;
        uint32_t rank[8];
        bool emit[8];
        const bool wave_live = __ballot(live) != 0ull;           // a wave whose samples all have zero gradient skips the hashing
        if (wave_live) level_records<F, !WRITE>(g, xv, dy, live, lane, r);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            emit[c] = false; rank[c] = 0;
            if (!wave_live) continue;
            bool nz = !WRITE;                                     // counting pass: every corner of a live tail (upper bound)
            if (WRITE) {
#pragma unroll
                for (int f = 0; f < F; ++f) nz |= (r.val[c][f] != 0.0f);
            }
            const bool e = r.tail & nz;
            emit[c] = e;
            const unsigned long long mask = __ballot(e);
            if (mask == 0ull) continue;
            const uint32_t b = first + bin_of(r.idx[c], lg, sh);
            // rank inside the workgroup's share of the bin.  With few bins (small batches) every emitting lane of the wave
            // tends to hit the same bin: one LDS atomic for the wave instead of <= 64 serialised ones.  With hundreds of
            // bins per level the lanes scatter and plain LDS integer atomics (7 cycles per wave instruction) are cheapest.
            if (lg <= 1) {                                       // wave-uniform
                const int lead = __ffsll((long long)mask) - 1;
                const uint32_t b0 = __builtin_amdgcn_readlane(b, lead);
                if (__ballot(e && b == b0) == mask) {
                    uint32_t base = 0;
                    const uint32_t mb = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    if (e && mb == 0) base = atomicAdd(&lcnt[b0], (uint32_t)__popcll(mask));
                    base = __builtin_amdgcn_readlane(base, lead);
                    rank[c] = base + mb;
                } else if (e) {
                    rank[c] = atomicAdd(&lcnt[b], 1u);
                }
            } else if (e) {
                rank[c] = atomicAdd(&lcnt[b], 1u);
            }
        }
        if (WRITE) {
            __syncthreads();                                     // this level's local counts are complete
            for (uint32_t t = threadIdx.x; t < nb; t += BIN_THREADS) {
                const uint32_t c = lcnt[first + t];
                lcnt[first + t] = c ? atomicAdd(&cursors[first + t], c) : 0u;      // count -> global base of our share
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (emit[c]) {
                    uint32_t* dst = rec + (size_t)(lcnt[first + bin_of(r.idx[c], lg, sh)] + rank[c]) * RecW<F>::DW;
                    const uint32_t loc = local_of(r.idx[c], lg, sh);
#ifdef US_EXP_A_NOSTORE
                    if (loc == 0xFFFFFFF0u)
#endif
                    dst[0] = loc;
#pragma unroll
                    for (int f = 0; f < F; ++f) dst[1 + f] = __float_as_uint(r.val[c][f]);
                }
            }
        }
    }
    if (!WRITE) {
        __syncthreads();
        for (uint32_t t = TB0 + threadIdx.x; t < TB1; t += BIN_THREADS) { const uint32_t c = lcnt[t]; if (c) atomicAdd(&counts[t], c); }
    }
}

// exclusive scan of counts[0..TB) -> offsets[0..TB], cursors[t] = offsets[t]   (TB <= 4096: 4 elements per thread)
__global__ __launch_bounds__(1024) void k_bin_scan(const uint32_t* __restrict__ counts_all, uint32_t TB0, uint32_t TB, uint32_t* __restrict__ offsets_all,
                                                   uint32_t* __restrict__ cursors_all) {
    const uint32_t* counts = counts_all + TB0; uint32_t* offsets = offsets_all + TB0; uint32_t* cursors = cursors_all + TB0;
    __shared__ uint32_t sh[1024];
    const uint32_t t = threadIdx.x;
    uint32_t c[4], s4 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = (4 * t + k < TB) ? counts[4 * t + k] : 0u; s4 += c[k]; }
    sh[t] = s4;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t v = (t >= o) ? sh[t - o] : 0u;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    uint32_t run = sh[t] - s4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (4 * t + k < TB) { offsets[4 * t + k] = run; cursors[4 * t + k] = run; }
        run += c[k];
    }
    if (t == 1023) offsets[TB] = sh[1023];
}

// B: one workgroup per bin
template <int F>
__global__ __launch_bounds__(ACC_THREADS) void k_bin_accum(LevelTable tab, BinMap bm, uint32_t n_levels, uint32_t bin0,
                                                           const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursors,
                                                           const uint32_t* __restrict__ rec, float* __restrict__ grad) {
    __shared__ double acc[BIN_ACC_DOUBLES];
    const uint32_t b = bin0 + blockIdx.x;
    uint32_t level = 0;
    while (level + 1 < n_levels && bm.first[level + 1] <= b) ++level;
    const uint32_t lg = bm.log2nb[level], bl = b - bm.first[level];
    const uint32_t hs = tab.off[level + 1] - tab.off[level];
    const uint32_t r0 = offsets[b], r1 = cursors[b];             // what the writing pass really stored (<= the counted range)
    if (r0 == r1) return;                                        // nothing landed in this bin (wave-uniform)
    const uint32_t sh = bm.shift[level];
    const uint32_t n_local = sh ? (1u << sh) : (bl < hs ? ((hs - 1u - bl) >> lg) + 1u : 0u);   // entries owned by this bin
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
#if defined(US_EXP_B_NOATOMIC)
#pragma unroll
                for (int f = 0; f < F; ++f) acc[loc[u] * F + f] = (double)v[u][f];
#elif defined(US_EXP_B_SPREAD)
#pragma unroll
                for (int f = 0; f < F; ++f) atomicAdd(&acc[((loc[u] + threadIdx.x * 7u) & (BIN_ACC_DOUBLES / F - 1)) * F + f], (double)v[u][f]);
#else
#pragma unroll
                for (int f = 0; f < F; ++f) atomicAdd(&acc[loc[u] * F + f], (double)v[u][f]);            // ds_add_f64
#endif
            }
        }
    }
    __syncthreads();
    float* gl = grad + (size_t)tab.off[level] * F;
    for (uint32_t k = threadIdx.x; k < n_local * F; k += ACC_THREADS) {
        const double v = acc[k];
        if (v != 0.0) {
            const uint32_t e = sh ? ((bl << sh) | (k / F)) : (((k / F) << lg) | bl), f = k % F;
            gl[(size_t)e * F + f] += (float)v;                   // this bin is the only writer of its entries
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
static size_t header_bytes() { return (size_t)(3 * (BIN_MAX_TOTAL + 64)) * sizeof(uint32_t); }   // multiple of 16 bytes

extern "C" size_t us_hashgrid_bwd_workspace_bytes(const us_grid_desc* d, int64_t n) {
    if (!d || n <= 0) return 0;
    return header_bytes() + (size_t)n * 8u * d->n_levels * rec_dwords(d->n_features) * sizeof(uint32_t);
}

extern "C" int us_hashgrid_bwd_binned(const us_grid_desc* d, const float* x, const float* dL_dy, int64_t n, float* grad_params,
                                      int flags, void* workspace, size_t workspace_bytes, void* stream) {
    US_REQUIRE(d, US_ERR_NULL, "us_hashgrid_bwd_binned: desc is NULL");
    US_REQUIRE(d->n_levels >= 1 && d->n_levels <= US_MAX_LEVELS && (d->n_features == 1 || d->n_features == 2 || d->n_features == 4) &&
               d->n_params == d->offset[d->n_levels] * d->n_features, US_ERR_CONFIG, "us_hashgrid_bwd_binned: bad descriptor");
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(x && dL_dy && grad_params && workspace, US_ERR_NULL, "us_hashgrid_bwd_binned: NULL pointer");
    US_REQUIRE((uint64_t)n * 8ull * d->n_levels < 0xFFFFFFFFull, US_ERR_SHAPE, "us_hashgrid_bwd_binned: n too large for 32-bit record ranks");
    US_REQUIRE(workspace_bytes >= us_hashgrid_bwd_workspace_bytes(d, n), US_ERR_WORKSPACE,
               "us_hashgrid_bwd_binned: workspace %zu B < %zu B", workspace_bytes, us_hashgrid_bwd_workspace_bytes(d, n));
    BinMap bm;
    const int TB = make_binmap(d, n, &bm);
    US_REQUIRE(TB <= BIN_MAX_TOTAL, US_ERR_CONFIG, "us_hashgrid_bwd_binned: %d bins > %d (table too large for this path)", TB, BIN_MAX_TOTAL);
    const LevelTable t = make_table(d);
    hipStream_t s = (hipStream_t)stream;
    const int clamp = flags & US_GRID_CLAMP01, lm = (flags & US_GRID_LEVEL_MAJOR) ? 1 : 0;
    uint32_t* counts = (uint32_t*)workspace;
    uint32_t* offsets = counts + (BIN_MAX_TOTAL + 64);
    uint32_t* cursors = offsets + (BIN_MAX_TOTAL + 64);
    uint32_t* rec = (uint32_t*)((char*)workspace + header_bytes());
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)(BIN_MAX_TOTAL + 64) * sizeof(uint32_t), s);
    if (e != hipSuccess) { us_set_error("us_hashgrid_bwd_binned: memset: %s", hipGetErrorString(e)); return (int)e; }
    dim3 gridA((unsigned)us_cdiv(n, BIN_THREADS)), block(BIN_THREADS);
    const uint32_t L = d->n_levels;
    // Level groups (count -> scan -> scatter -> accumulate per group, record buffer reused from offset 0).  Tried with
    // ~100 MB groups so that the accumulate pass would read the records from the Infinity Cache: measured SLOWER (0.46 vs
    // 0.36 ms per grid; the per-launch fixed costs of four passes outweigh the cache hits), so one group is the default.
    const uint64_t per_level = (uint64_t)n * 8ull * rec_dwords(d->n_features) * sizeof(uint32_t);
    uint32_t lg_levels = (uint32_t)(BIN_GROUP_BYTES / (per_level ? per_level : 1));
    if (lg_levels < 1) lg_levels = 1;
    if (lg_levels > L) lg_levels = L;
#define LAUNCH_BIN(F)                                                                                                          \
    for (uint32_t l0 = 0; l0 < L; l0 += lg_levels) {                                                                           \
        const uint32_t l1 = (l0 + lg_levels < L) ? l0 + lg_levels : L;                                                         \
        const uint32_t b0 = bm.first[l0], nbins = bm.first[l1] - bm.first[l0];                                                 \
        hipLaunchKernelGGL((k_bin<F, false>), gridA, block, 0, s, t, bm, L, l0, l1, x, dL_dy, n, clamp, lm, counts, cursors, rec); \
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, counts, b0, nbins, offsets, cursors);                        \
        hipLaunchKernelGGL((k_bin<F, true>), gridA, block, 0, s, t, bm, L, l0, l1, x, dL_dy, n, clamp, lm, counts, cursors, rec);  \
        hipLaunchKernelGGL((k_bin_accum<F>), dim3(nbins), dim3(ACC_THREADS), 0, s, t, bm, L, b0, offsets, cursors, rec, grad_params); \
    }
    switch (d->n_features) { case 1: LAUNCH_BIN(1) break; case 2: LAUNCH_BIN(2) break; default: LAUNCH_BIN(4) break; }
#undef LAUNCH_BIN
    US_CHECK_LAUNCH("us_hashgrid_bwd_binned");
    return US_OK;
}
