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
//      (or the        (inside aligned groups of 8 lanes = 8 consecutive samples of one ray) sit in the same grid cell are
//      encoder,       RUN-COMBINED with DPP row shifts (segmented inclusive scan of the 8 corner contributions), so a run
//      k_fwd_count)   emits one record per corner instead of one per sample.  Records are counted per BIN; a bin owns every
//                     n_bins-th 128-byte LINE of the level's gradient slab (interleaving balances the load whatever the
//                     geometry or the hash does).  The counts are kept PER WORKGROUP: row w of a [workgroups][bins] matrix.
//   scans             k_bin_colscan: column scan over the workgroups (XCD-major order); k_bin_scan: exclusive scan of the bin
//                     totals -> every (workgroup, bin) pair owns an exact, gap-free record range (no over-allocation, no
//                     overflow, no cursor to reserve from) + the list of bins hot enough to be split.
//   A1 k_bin<WRITE>   same pass with the values.  The records of one level are collected in an LDS stage sorted by bin (the
//                     scan of the workgroup's own counts seeds the cursors: one ds_add_rtn + one ds_write_b128 per record,
//                     each already carrying its final global slot) and copied out by a flat loop, consecutive threads
//                     writing consecutive 12-byte records {local entry, F values}: HBM sees full-line writes.
//   B  k_bin_accum    one 512-thread workgroup per bin (hot bins: one per 16 384-record chunk): 32 KiB of f64 accumulators
//                     in LDS, ds_add_f64 over the bin's records (non-temporal loads, 2 x 4 in flight per thread), then
//                     the bin's lines of the gradient table are written (OVERWRITE), added to, or -- split bins -- added
//                     with float atomics.
// Sums are formed in double precision; the only global float atomics are those of split bins.
#include "binned_dev.h"
#include <string.h>

#ifndef BIN_THREADS
#define BIN_THREADS 1024                 // binning kernels: 1024 consecutive points per workgroup
#endif
#define BIN_SCAN_ELEMS (BIN_MAX_TOTAL / BIN_THREADS)
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

struct BinMap {
    uint32_t first[US_MAX_LEVELS + 1];   // prefix sum of bins per level
    uint32_t log2nb[US_MAX_LEVELS];      // bins per level = 1 << log2nb
};

static inline uint32_t bin_entries(uint32_t F) { return BIN_ACC_DOUBLES / F; }

// records are {local entry, F values} = 1+F dwords.  Measured: the scatter pass's stores run at HBM write speed (padding the
// F = 2 record to 16 bytes for single vector stores made the whole pass slower: bytes matter, not store instructions)
template <int F> struct RecW { static constexpr int DW = 1 + F; };
static inline uint32_t rec_dwords(uint32_t F) { return 1u + F; }

// US_GRID_BWD_PACKED (F = 2): 8-byte records { local entry : 11 bits | v0 : top 26 bits of the fp32 | v1 : top 27 bits },
// round-to-nearest on the dropped 6 / 5 mantissa bits (relative 2^-18 / 2^-19 per record; the sums stay f64).  One 8-byte
// load / store per record instead of three dwords, a third fewer bytes through the two passes.  NaN / inf survive.
__device__ __forceinline__ uint32_t round_drop(float v, uint32_t half) {
    const uint32_t u = __float_as_uint(v);
    return ((u & 0x7FFFFFFFu) > 0x7F800000u) ? u : u + half;       // NaN payloads are left alone (the add could wrap the sign)
}
__device__ __forceinline__ uint2 pack_rec(uint32_t loc, float v0, float v1) {
    const uint32_t t0 = round_drop(v0, 0x20u) >> 6, t1 = round_drop(v1, 0x10u) >> 5;
    return make_uint2(loc | (t0 << 11), (t0 >> 21) | (t1 << 5));
}
__device__ __forceinline__ void unpack_rec(uint2 r, uint32_t& loc, float& v0, float& v1) {
    loc = r.x & 0x7FFu;
    v0 = __uint_as_float(((r.y & 31u) << 27) | ((r.x >> 11) << 6));
    v1 = __uint_as_float(r.y & ~31u);
}

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
    if (want < 4) want = 4;                                      // >= 16 bins per level: keeps the LDS counters of a wave apart
    for (uint32_t l = 0; l < US_MAX_LEVELS; ++l) {
        bm->first[l] = total; bm->log2nb[l] = 0;
        if (l >= d->n_levels) continue;
        const uint32_t hs = d->offset[l + 1] - d->offset[l];
        const uint32_t lines = (hs + (1u << BIN_LINE_LOG2) - 1u) >> BIN_LINE_LOG2;
        uint32_t lg = 0;
        while (bin_n_local(hs, 0, lg) > be) ++lg;                // capacity of the f64 slice
        if (lg < want) lg = want;                                // load (fewer bins on the coarse levels: measured, no gain)
        while (lg > 0 && (1u << lg) > lines) --lg;               // never more bins than lines
        if (bin_n_local(hs, 0, lg) > be) return -1;
        bm->log2nb[l] = lg;
        total += 1u << lg;
    }
    bm->first[US_MAX_LEVELS] = total;
    for (uint32_t l = d->n_levels; l <= US_MAX_LEVELS; ++l) bm->first[l] = total;
    return (int)total;
}

// Per-level constants of the binning passes, precomputed on the host: one s_load_dwordx8 per level.
struct BinLevel {
    float    scale;
    uint32_t res, res2;      // dense levels: entry = x + y*res + z*res2 (wrapped once at hs)
    uint32_t hs;             // entries of the level
    uint32_t hashed;         // 1: coherent prime hash & (hs-1)
    uint32_t lg;             // log2(bins of this level)
    uint32_t first;          // first bin of this level
    uint32_t packable;       // cell coordinates fit 10 bits each -> run-combining possible
};
struct BinLevels { BinLevel l[US_MAX_LEVELS]; };

static BinLevels make_bin_levels(const us_grid_desc* d, const BinMap& bm) {
    BinLevels r;
    memset(&r, 0, sizeof(r));
    for (uint32_t l = 0; l < d->n_levels; ++l) {
        BinLevel& q = r.l[l];
        q.scale = d->scale[l]; q.res = d->resolution[l]; q.res2 = q.res * q.res; q.hs = d->offset[l + 1] - d->offset[l];
        uint32_t stride = 1; bool early = false;                 // as level_geom() in hashgrid_dev.h
        for (int dim = 0; dim < 3; ++dim) { if (stride <= q.hs) stride *= q.res; else early = true; }
        q.hashed = (early || q.hs < stride) ? 1u : 0u;
        q.lg = bm.log2nb[l]; q.first = bm.first[l];
        q.packable = q.res <= 1023u ? 1u : 0u;
    }
    return r;
}

// Binning passes.  Every workgroup takes 1024 consecutive points through all levels.  Per level and point:
//   cell, fractional position, the 8 entry indices (hash: two 32-bit multiplies, the +1 vertices by adding the primes;
//   dense: two 24-bit mads, the wrap-around at hs as one unsigned min);
//   RUN-COMBINING: lanes of an aligned group of 8 (= 8 consecutive samples of a ray) that sit in the same cell form runs;
//   a segmented inclusive scan (Hillis-Steele, DPP row_shr 1/2/4 fused into v_fmac) sums the 8 x F corner products of a
//   run into its last lane, which alone emits records.  Equal cells that are not adjacent stay separate runs: exact for
//   any point order.  NB every DPP move executes with the whole wave active (a disabled lane reads as 0).
//   A0 (WRITE = false): wg_counts[workgroup][bin] = records this workgroup will emit (every corner of every live run tail).
//   A1 (WRITE = true) : the LDS counters start at the workgroup's range starts (bin offset + column prefix), so the
//                       returning LDS atomic IS the record's final slot; the 12-byte record is stored at once.
#define BIN_STAGE_RECORDS (BIN_THREADS * 8)   // records one workgroup emits per level at most

template <int F, bool WRITE, bool PACKED = false>
__global__ __launch_bounds__(BIN_THREADS) void k_bin(BinLevels lv, uint32_t n_levels, uint32_t TB, const float* __restrict__ x,
                                                     const float* __restrict__ dL_dy, int64_t n, int clamp, int lm,
                                                     uint32_t* __restrict__ wg_counts, const uint32_t* __restrict__ wg_prefix,
                                                     const uint32_t* __restrict__ offsets, uint32_t* __restrict__ rec, int all_live,
                                                     int64_t plane_stride) {
    // plane_stride: points per level plane of a level-major dL_dy (= n, or the whole batch when x / dL_dy are a range of it)
    // all_live: every point counts as live whatever its gradient (the counts came from the forward pass, which has no
    // gradients to look at: us_hashgrid_fwd_counted); a dead sample then emits zero records.
    // lcnt: COUNT pass: records per bin.  WRITE pass: cursor into the workgroup's LDS stage, where the records of one level
    // are collected sorted by bin (counting sort: the exclusive scan of this workgroup's own counts gives every bin's place),
    // each with its final global slot; the copy-out is then one flat, fully occupied loop.
    // (F = 4 records do not fit a stage beside the counters: that instantiation stores every record straight from the
    //  lane that owns it, lcnt being the global cursor.)
    constexpr bool STAGED = WRITE && F <= 2;
    __shared__ uint32_t lcnt[BIN_MAX_TOTAL];
    __shared__ uint32_t gdelta[STAGED ? BIN_MAX_TOTAL : 1];                    // global record slot = stage cursor + gdelta[bin]
    __shared__ uint4    st[STAGED ? BIN_STAGE_RECORDS : 1];                    // {global slot, local entry, values}: 128 KiB
    uint32_t* row = wg_counts + (size_t)blockIdx.x * BIN_MAX_TOTAL;
    if (WRITE && !STAGED) {
        const uint32_t* pre = wg_prefix + (size_t)blockIdx.x * BIN_MAX_TOTAL;
        for (uint32_t t = threadIdx.x; t < TB; t += BIN_THREADS) lcnt[t] = offsets[t] + pre[t];
    } else if (WRITE) {
        uint32_t* sc = reinterpret_cast<uint32_t*>(st);          // scan scratch (the stage is not in use yet)
        uint32_t c4[BIN_SCAN_ELEMS], s4 = 0;
#pragma unroll
        for (int k = 0; k < BIN_SCAN_ELEMS; ++k) { c4[k] = (BIN_SCAN_ELEMS * threadIdx.x + k < TB) ? row[BIN_SCAN_ELEMS * threadIdx.x + k] : 0u; s4 += c4[k]; }
        sc[threadIdx.x] = s4;
        __syncthreads();
        for (uint32_t o = 1; o < BIN_THREADS; o <<= 1) {
            const uint32_t v = (threadIdx.x >= o) ? sc[threadIdx.x - o] : 0u;
            __syncthreads();
            sc[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t run = sc[threadIdx.x] - s4;
#pragma unroll
        for (int k = 0; k < BIN_SCAN_ELEMS; ++k) { if (BIN_SCAN_ELEMS * threadIdx.x + k < TB) lcnt[BIN_SCAN_ELEMS * threadIdx.x + k] = run; run += c4[k]; }
        __syncthreads();
        const uint32_t* pre = wg_prefix + (size_t)blockIdx.x * BIN_MAX_TOTAL;
        for (uint32_t t = threadIdx.x; t < TB; t += BIN_THREADS) gdelta[t] = offsets[t] + pre[t] - lcnt[t];
    } else {
        for (uint32_t t = threadIdx.x; t < TB; t += BIN_THREADS) lcnt[t] = 0u;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, lg16 = lane & (RUN_GROUP - 1);
    const int64_t i = (int64_t)blockIdx.x * BIN_THREADS + threadIdx.x;
    const bool in = i < n;
    float xv[3] = {0.f, 0.f, 0.f};
    if (in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) xv[k] = load_x(x, i, k, clamp);
    }
    typedef typename Feat<F>::T FT;
    // The gradient of level l+1 is requested while level l is worked on, and CONSUMED before level l's first record store: on gfx9 a
    // wait for a load also waits for every store issued before it (one vmcnt counter), so a load must never be waited for behind
    // this level's stores.  (Round 1 held all levels' gradients in 2 x 32 registers instead; the vertex runs need those registers.)
    const int64_t dy_base = lm ? i * F : i * (int64_t)(n_levels * F), dy_step = lm ? plane_stride * F : (int64_t)F;
    auto load_dy = [&](uint32_t level, float (&d)[F]) {
#pragma unroll
        for (int f = 0; f < F; ++f) d[f] = 0.0f;
        if (in) feat_to_array<F>(*reinterpret_cast<const FT*>(dL_dy + dy_base + (int64_t)level * dy_step), d);
    };
    float dn[F];
    load_dy(0, dn);
    for (uint32_t level = 0; level < n_levels; ++level) {
        const BinLevel q = lv.l[level];
        float dy[F];
#pragma unroll
        for (int f = 0; f < F; ++f) dy[f] = dn[f];
        bool live = all_live && in;
#pragma unroll
        for (int f = 0; f < F; ++f) live |= (dy[f] != 0.0f);
        const uint32_t lbase = STAGED ? lcnt[q.first] : 0u;      // stage slot 0 of this level, read before anyone adds to it
        // The level's arithmetic (cells, indices, products, run scan) runs BEFORE the barrier that ends the previous level's
        // copy-out: the VALU phase of one wave overlaps the store phase of the others.
        uint32_t idx[8];
        float val[8][F];
        uint32_t tail = 0u;                                      // bit p: the lane ends a run of slot p and emits its record
        if (__ballot(live) != 0ull) {                            // a wave whose samples all have zero gradient skips the level
        // ---- cell and position
        float pos[3]; uint32_t cell[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pos_fract(xv[k], q.scale, pos[k], cell[k]);
        // ---- vertex runs (binned_dev.h): contributions by parity slot, runs of equal vertices over adjacent lanes
        uint32_t key[8];
        slot_keys(cell, live && q.packable, lane, key);
        const uint2 ht = slot_run_masks(key, lg16);
        uint32_t head = ht.x;
        tail = live ? ht.y : 0u;
        slot_entries(q.hashed != 0u, q.hs, q.res, q.res2, cell, idx);
        // ---- slot products and their segmented scan (only the writing pass needs the values)
        if (WRITE) {
            float w[8];
            slot_weights(pos, cell, w);
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int f = 0; f < F; ++f) val[p][f] = w[p] * dy[f];                  // dy == 0 on dead lanes
            if constexpr (F == 2) {                              // the scan written out (binned_dev.h): 32 instead of 48 instructions per step
                uint32_t take_all, steps;
                SLOT_SCAN_PRE(head, take_all, steps)
                slot_scan_apply_pairs(val, take_all, steps);
            } else {
                SLOT_SCAN(F)
            }
        }
        }   // wave has live samples
        if (level + 1 < n_levels) load_dy(level + 1, dn);
        auto consume_next = [&]() {                              // forces the wait for dn here, ahead of the stores that follow
#pragma unroll
            for (int f = 0; f < F; ++f) asm volatile("" : "+v"(dn[f]));
        };
        if (!STAGED) consume_next();
        if (STAGED) lds_barrier();                               // the previous level's copy-out has left the stage
        // ---- emit: one record per run end (a vertex whose weight is exactly 0 becomes a zero record, so the counted ranges are exact)
        if (tail != 0u) {
            const uint32_t nbm = (1u << q.lg) - 1u;
            if (STAGED) {
                // all cursor atomics (and the bins' slot offsets) are in flight before the first record is staged: one LDS
                // round trip per level instead of eight
                uint32_t cur[8], gd[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if ((tail >> c) & 1u) {
                        const uint32_t b = q.first + ((idx[c] >> BIN_LINE_LOG2) & nbm);
                        cur[c] = atomicAdd(&lcnt[b], 1u);
                        gd[c] = gdelta[b];
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if ((tail >> c) & 1u) {
                        uint4 e;                                 // {local entry, values, global slot}: the record's dwords lead,
                        e.x = local_of(idx[c], q.lg);            // so the copy-out stores v[0:2] of the ds_read_b128 as they are
                        e.y = __float_as_uint(val[c][0]);
                        e.z = F > 1 ? __float_as_uint(val[c][F > 1 ? 1 : 0]) : 0u;
                        e.w = cur[c] + gd[c];
                        st[cur[c] - lbase] = e;
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (!((tail >> c) & 1u)) continue;
                    const uint32_t b = q.first + ((idx[c] >> BIN_LINE_LOG2) & nbm);
                    if (WRITE) {
                        const uint32_t slot = atomicAdd(&lcnt[b], 1u);
                        uint32_t* dst = rec + (size_t)slot * RecW<F>::DW;
                        dst[0] = local_of(idx[c], q.lg);
#pragma unroll
                        for (int f = 0; f < F; ++f) dst[1 + f] = __float_as_uint(val[c][f]);
                    } else {
                        atomicAdd(&lcnt[b], 1u);
                    }
                }
            }
        }
        if (STAGED) {
            // ---- copy the level's records out.  The stage holds them sorted by bin, and inside a bin the global slots are
            //      consecutive, so consecutive threads write consecutive 12-byte records: contiguous runs per bin.
            consume_next();
            lds_barrier();
            const uint32_t n_lvl = lcnt[q.first + (1u << q.lg) - 1u] - lbase;
            auto put = [&](const uint4 e) {
                if constexpr (PACKED) {
                    static_assert(!PACKED || F == 2, "packed records carry two values");
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(rec) + ((size_t)e.w << 3)) =
                        pack_rec(e.x, __uint_as_float(e.y), __uint_as_float(e.z));
                } else {
                    struct __attribute__((packed, aligned(4))) RecT { uint32_t w[RecW<F>::DW]; };
                    RecT r;
                    r.w[0] = e.x; r.w[1] = e.y;
                    if (F > 1) r.w[F > 1 ? 2 : 1] = e.z;
                    const uint32_t byte_off = RecW<F>::DW == 3 ? (e.w << 3) + (e.w << 2) : e.w * (uint32_t)(RecW<F>::DW * 4);   // < 2^32 (host check)
                    *reinterpret_cast<RecT*>(reinterpret_cast<char*>(rec) + byte_off) = r;     // (non-temporal stores: 127 vs 78 us)
                }
            };
            // two stage reads in flight per thread (one per iteration left every store waiting for its own LDS round trip)
            uint32_t k = threadIdx.x;
            for (; k + BIN_THREADS < n_lvl; k += 2 * BIN_THREADS) {
                const uint4 e0 = st[k], e1 = st[k + BIN_THREADS];
                put(e0); put(e1);
            }
            if (k < n_lvl) put(st[k]);
        }
    }
    if (!WRITE) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < TB; t += BIN_THREADS) row[t] = lcnt[t];
    }
}

// column scan: wg_prefix[w][b] = sum of wg_counts[w'][b] over the workgroups w' ordered before w ; totals[b] = column sum.
// Order of the workgroups' segments inside a bin: by XCD first (workgroup w runs on XCD w % 8), so that the ~300-byte
// segments written through one XCD's L2 are neighbours.  8 lanes per bin, one per XCD class: each sums its class (pass 1),
// the classes are scanned across the 8 lanes, and pass 2 writes the prefixes.
// In OVERWRITE mode the bins that will be split over several accumulate workgroups (total > ACC_CHUNK, added with float
// atomics) get their gradient entries cleared here, two kernels ahead of the first add.
#define COLSCAN_THREADS 64
#define COLSCAN_BINS (COLSCAN_THREADS / 8)
template <int F>
__global__ __launch_bounds__(COLSCAN_THREADS) void k_bin_colscan(LevelTable tab, BinMap bm, uint32_t n_levels, const uint32_t* __restrict__ wg_counts,
                                                                 uint32_t* __restrict__ wg_prefix, uint32_t n_wg, uint32_t TB,
                                                                 uint32_t* __restrict__ totals, float* __restrict__ grad, int overwrite, uint32_t chunk0) {
    const uint32_t xcd = threadIdx.x & 7u;
    const uint32_t b = blockIdx.x * COLSCAN_BINS + (threadIdx.x >> 3);
    const bool ok = b < TB;
    uint32_t sum = 0;
    if (ok) {
        uint32_t w = xcd;
        for (; w + 56 < n_wg; w += 64) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = wg_counts[(size_t)(w + 8 * k) * BIN_MAX_TOTAL + b];
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[k];
        }
        for (; w < n_wg; w += 8) sum += wg_counts[(size_t)w * BIN_MAX_TOTAL + b];
    }
    uint32_t incl = sum;                                         // inclusive scan over the 8 XCD classes of the bin
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 8); if ((int)xcd >= o) incl += t; }
    const uint32_t total = __shfl(incl, 7, 8);
    if (ok) {
        uint32_t run = incl - sum;
        uint32_t w = xcd;
        for (; w + 56 < n_wg; w += 64) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = wg_counts[(size_t)(w + 8 * k) * BIN_MAX_TOTAL + b];
#pragma unroll
            for (int k = 0; k < 8; ++k) { wg_prefix[(size_t)(w + 8 * k) * BIN_MAX_TOTAL + b] = run; run += v[k]; }
        }
        for (; w < n_wg; w += 8) { const uint32_t v = wg_counts[(size_t)w * BIN_MAX_TOTAL + b]; wg_prefix[(size_t)w * BIN_MAX_TOTAL + b] = run; run += v; }
        if (xcd == 0) totals[b] = total;
    }
    if (!overwrite) return;
    unsigned long long hot = __ballot(ok && xcd == 0 && total > chunk0);
    while (hot) {                                                // wave-uniform loop (one wave per workgroup)
        const int src = __ffsll((long long)hot) - 1;
        hot &= hot - 1ull;
        const uint32_t hb = blockIdx.x * COLSCAN_BINS + ((uint32_t)src >> 3);
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
// a bin with c > chunk records is accumulated by ceil(c / chunk) workgroups; chunk 0 belongs to the bin's own workgroup,
// chunks 1.. are listed in extra[] as bin | chunk << 16.  chunk = ACC_CHUNK, or the multiple of it that keeps the list within
// ACC_EXTRA_MAX entries (hdr[0] = number of extras, hdr[1] = chunk).
__global__ __launch_bounds__(1024) void k_bin_scan(const uint32_t* __restrict__ counts, uint32_t TB, uint32_t* __restrict__ offsets,
                                                   uint32_t* __restrict__ extra, uint32_t* __restrict__ hdr, uint32_t chunk0) {
    // chunk0 = ACC_CHUNK, or 0xFFFFFFFF (US_GRID_BWD_DETERMINISTIC): no bin is split, every sum is formed by one workgroup in f64
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t sx[1024];
    const uint32_t t = threadIdx.x;
    uint32_t c[4], s4 = 0, x4 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = (4 * t + k < TB) ? counts[4 * t + k] : 0u; s4 += c[k]; x4 += c[k] > chunk0 ? (c[k] - 1u) / chunk0 : 0u; }
    sh[t] = s4; sx[t] = x4;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        const uint32_t v = (t >= o) ? sh[t - o] : 0u, w = (t >= o) ? sx[t - o] : 0u;
        __syncthreads();
        sh[t] += v; sx[t] += w;
        __syncthreads();
    }
    uint32_t chunk = chunk0;
    const uint32_t x_all = sx[1023];
    if (x_all > ACC_EXTRA_MAX) {                                 // wave-uniform, rare: coarser chunks, scanned again
        chunk = chunk0 * ((x_all + ACC_EXTRA_MAX - 1u) / ACC_EXTRA_MAX);
        x4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) x4 += c[k] > chunk ? (c[k] - 1u) / chunk : 0u;
        __syncthreads();
        sx[t] = x4;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            const uint32_t w = (t >= o) ? sx[t - o] : 0u;
            __syncthreads();
            sx[t] += w;
            __syncthreads();
        }
    }
    uint32_t run = sh[t] - s4, xrun = sx[t] - x4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (4 * t + k < TB) offsets[4 * t + k] = run;
        run += c[k];
        const uint32_t nx = c[k] > chunk ? (c[k] - 1u) / chunk : 0u;
        for (uint32_t j = 0; j < nx; ++j) extra[xrun + j] = (4 * t + k) | ((j + 1u) << 16);
        xrun += nx;
    }
    if (t == 1023) { offsets[TB] = sh[1023]; hdr[0] = sx[1023]; hdr[1] = chunk; }
}

// B: one workgroup per bin (+ one per extra chunk of a hot bin; those come FIRST in the grid: they are the longest jobs)
template <int F, bool PACKED = false>
__global__ __launch_bounds__(ACC_THREADS) void k_bin_accum(LevelTable tab, BinMap bm, uint32_t n_levels, uint32_t e_max,
                                                           const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ extra,
                                                           const uint32_t* __restrict__ n_extra,
                                                           const uint32_t* __restrict__ rec, float* __restrict__ grad, int overwrite) {
    // accumulators by component: acc[f * NE + loc].  (Interleaved {f0, f1} pairs put every ds_add_f64 of a wave on a 16-byte
    // stride, i.e. on half of the banks: PMC SQ_LDS_BANK_CONFLICT was 39 % of the LDS-active cycles.)
    constexpr uint32_t NE = BIN_ACC_DOUBLES / F;
    __shared__ double acc[BIN_ACC_DOUBLES];
    uint32_t b, chunk = 0;
    const uint32_t CH = n_extra[1];                                    // records per workgroup (k_bin_scan)
    if (blockIdx.x < e_max) {
        if (blockIdx.x >= n_extra[0]) return;
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
    const bool split = (b1 - b0) > CH;                                // several workgroups add into this bin's entries
    const uint32_t r0 = b0 + chunk * CH;
    const uint32_t r1 = (b1 - r0 > CH) ? r0 + CH : b1;
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
    // Software-pipelined record stream: the loads of batch k+1 are in flight while batch k goes through the LDS atomics;
    // the first batch is requested before the accumulators are cleared.
    uint32_t loc[2][ACC_UNROLL]; float v[2][ACC_UNROLL][F];
    // The loads are UNCONDITIONAL (a rank beyond the range re-reads the range's last record and is marked dead afterwards): a load
    // inside `if (r < r1)` sits in its own exec-masked block, and the compiler ends every such block with s_waitcnt vmcnt(0) --
    // one record in flight per thread instead of 2 x ACC_UNROLL (seen in the ISA of round 1's kernel).
    const uint32_t r_last = r1 - 1u;                                  // r1 > r0 here
    auto fetch = [&](int buf, uint32_t base) {
#pragma unroll
        for (int u = 0; u < ACC_UNROLL; ++u) {
            const uint32_t r = base + u * ACC_THREADS + threadIdx.x;
            const uint32_t rc = min(r, r_last);
            if constexpr (PACKED) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(rec) + rc);
                unpack_rec(make_uint2(w.x, w.y), loc[buf][u], v[buf][u][0], v[buf][u][F > 1 ? 1 : 0]);
            } else {
                const uint32_t* src = rec + (size_t)rc * RecW<F>::DW;
                // every record is read exactly once: non-temporal loads keep the stream out of the caches (measured -7 us)
                loc[buf][u] = __builtin_nontemporal_load(src);
#pragma unroll
                for (int f = 0; f < F; ++f) v[buf][u][f] = __uint_as_float(__builtin_nontemporal_load(src + 1 + f));
            }
        }
    };
    auto add = [&](int buf, uint32_t base) {                          // the loaded registers are first touched HERE
#pragma unroll
        for (int u = 0; u < ACC_UNROLL; ++u) {
            if (base + u * ACC_THREADS + threadIdx.x <= r_last && loc[buf][u] < n_local) {   // (records are always in range: the
#pragma unroll                                                                                      //  compare keeps a bug from writing LDS out of bounds)
                for (int f = 0; f < F; ++f) atomicAdd(&acc[f * NE + loc[buf][u]], (double)v[buf][u][f]);   // ds_add_f64
            }
        }
    };
    fetch(0, r0);
#pragma unroll
    for (int f = 0; f < F; ++f)
        for (uint32_t k = threadIdx.x; k < n_local; k += ACC_THREADS) acc[f * NE + k] = 0.0;
    __syncthreads();
    constexpr uint32_t STEP = ACC_THREADS * ACC_UNROLL;
    // two buffers, the loop unrolled by two so that neither needs a register move; every fetch is unconditional (past the end it
    // re-reads the last record, one cached line) so that no load sits behind a branch of its own
    for (uint32_t base = r0;;) {
        fetch(1, base + STEP);
        add(0, base);
        base += STEP;
        if (base >= r1) break;
        fetch(0, base + STEP);
        add(1, base);
        base += STEP;
        if (base >= r1) break;
    }
    __syncthreads();
    for (uint32_t loc = threadIdx.x; loc < n_local; loc += ACC_THREADS) {
        const uint32_t e = entry_of(loc, bl, lg);
        if (e >= hs) continue;
        float v[F]; bool any = false;
#pragma unroll
        for (int f = 0; f < F; ++f) { v[f] = (float)acc[f * NE + loc]; any |= (v[f] != 0.0f); }
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
// [n_wg][BIN_MAX_TOTAL] | their column prefixes [n_wg][BIN_MAX_TOTAL] | records
static uint32_t extra_max(const us_grid_desc*, int64_t) { return ACC_EXTRA_MAX; }
static size_t header_bytes(const us_grid_desc* d, int64_t n) {
    const size_t em = ((size_t)extra_max(d, n) + 15u) & ~(size_t)15u;
    return (size_t)(2 * (BIN_MAX_TOTAL + 64) + 16 + em) * sizeof(uint32_t) + 2 * (size_t)us_cdiv(n, BIN_THREADS) * BIN_MAX_TOTAL * sizeof(uint32_t);
}

extern "C" size_t us_hashgrid_bwd_workspace_bytes(const us_grid_desc* d, int64_t n) {
    if (!d || n <= 0) return 0;
    return header_bytes(d, n) + (size_t)n * 8u * d->n_levels * rec_dwords(d->n_features) * sizeof(uint32_t);
}

extern "C" int us_hashgrid_bwd_binned_supported(const us_grid_desc* d, int64_t n) {
    if (!d || n <= 0 || d->n_levels < 1 || d->n_levels > US_MAX_LEVELS) return 0;
    if (!(d->n_features == 1 || d->n_features == 2 || d->n_features == 4)) return 0;
    if ((uint64_t)n * 8ull * d->n_levels * rec_dwords(d->n_features) * 4ull > 0xFFFFFFFFull) return 0;
    BinMap bm;
    const int TB = make_binmap(d, n, &bm);
    return (TB > 0 && TB <= BIN_MAX_TOTAL) ? 1 : 0;
}

static int bwd_binned(const us_grid_desc* d, const float* x, const float* dL_dy, int64_t n, float* grad_params,
                      int flags, void* workspace, size_t workspace_bytes, void* stream, bool scan_only, int64_t plane_stride = 0) {
    const int64_t ps = plane_stride > 0 ? plane_stride : n;
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
    US_REQUIRE((scan_only || (x && dL_dy)) && grad_params && workspace, US_ERR_NULL, "us_hashgrid_bwd_binned: NULL pointer");
    US_REQUIRE(((uintptr_t)grad_params & 15u) == 0 && ((uintptr_t)workspace & 15u) == 0, US_ERR_SHAPE,
               "us_hashgrid_bwd_binned: grad_params and workspace must be 16-byte aligned");
    US_REQUIRE((uint64_t)n * 8ull * d->n_levels < 0xFFFFFFFFull, US_ERR_SHAPE, "us_hashgrid_bwd_binned: n too large for 32-bit record ranks");
    US_REQUIRE(workspace_bytes >= us_hashgrid_bwd_workspace_bytes(d, n), US_ERR_WORKSPACE,
               "us_hashgrid_bwd_binned: workspace %zu B < %zu B", workspace_bytes, us_hashgrid_bwd_workspace_bytes(d, n));
    BinMap bm;
    const int TB = make_binmap(d, n, &bm);
    US_REQUIRE(TB > 0 && TB <= BIN_MAX_TOTAL, US_ERR_CONFIG, "us_hashgrid_bwd_binned: %d bins > %d (table too large for this path)", TB, BIN_MAX_TOTAL);
    US_REQUIRE((uint64_t)n * 8ull * d->n_levels * rec_dwords(d->n_features) * 4ull <= 0xFFFFFFFFull, US_ERR_SHAPE,
               "us_hashgrid_bwd_binned: n = %lld too large for 32-bit record offsets (split the batch)", (long long)n);
    const LevelTable t = make_table(d);
    const BinLevels lv = make_bin_levels(d, bm);
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
    uint32_t* wg_prefix = wg_counts + (size_t)n_wg * BIN_MAX_TOTAL;
    dim3 gridA(n_wg), block(BIN_THREADS);
    const uint32_t L = d->n_levels;
    const int overwrite = (flags & US_GRID_BWD_OVERWRITE) ? 1 : 0, counted = (flags & US_GRID_BWD_COUNTED) ? 1 : 0;
    const bool packed = (flags & US_GRID_BWD_PACKED) != 0, scanned = (flags & US_GRID_BWD_SCANNED) != 0;
    const uint32_t chunk0 = (flags & US_GRID_BWD_DETERMINISTIC) ? 0xFFFFFFFFu : (uint32_t)ACC_CHUNK;
    US_REQUIRE(!(scan_only && !counted) && !(scanned && !counted), US_ERR_CONFIG,
               "us_hashgrid_bwd_binned: the scan passes can only run ahead on counts left by us_hashgrid_fwd_counted (US_GRID_BWD_COUNTED)");
#ifdef US_EXPERIMENTS
    US_REQUIRE(!packed || (d->n_features == 2 && bin_entries(2) <= 2048u), US_ERR_CONFIG,
               "us_hashgrid_bwd_binned: US_GRID_BWD_PACKED needs n_features == 2 (got %u)", d->n_features);
#else
    US_REQUIRE(!packed, US_ERR_CONFIG, "us_hashgrid_bwd_binned: US_GRID_BWD_PACKED is part of the experiments build only (tools/build_experiments.sh)");
#endif
    // (Splitting the levels into groups of ~100 MB of records, so that the accumulate pass would read them from the Infinity
    //  Cache, was measured SLOWER: 0.46 vs 0.36 ms per grid -- the fixed costs of four more passes outweigh the cache hits.)
#define LAUNCH_BIN_P(F, P)                                                                                                     \
    if (!counted) hipLaunchKernelGGL((k_bin<F, false>), gridA, block, 0, s, lv, L, (uint32_t)TB, x, dL_dy, n, clamp, lm, wg_counts, wg_prefix, offsets, rec, 0, ps); \
    if (!scanned) {                                                                                                            \
        hipLaunchKernelGGL((k_bin_colscan<F>), dim3(us_cdiv(TB, COLSCAN_BINS)), dim3(COLSCAN_THREADS), 0, s, t, bm, L, wg_counts, wg_prefix, n_wg, \
                           (uint32_t)TB, totals, grad_params, overwrite, chunk0);                                              \
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, totals, (uint32_t)TB, offsets, extra, n_extra, chunk0);      \
    }                                                                                                                          \
    if (scan_only) break;                                                                                                      \
    hipLaunchKernelGGL((k_bin<F, true, P>), gridA, block, 0, s, lv, L, (uint32_t)TB, x, dL_dy, n, clamp, lm, wg_counts, wg_prefix, offsets, rec, counted, ps); \
    hipLaunchKernelGGL((k_bin_accum<F, P>), dim3(e_max + TB), dim3(ACC_THREADS), 0, s, t, bm, L, e_max, offsets, extra, n_extra, rec, \
                       grad_params, overwrite);
#define LAUNCH_BIN(F) LAUNCH_BIN_P(F, false)
    switch (d->n_features) {
        case 1: LAUNCH_BIN(1) break;
#ifdef US_EXPERIMENTS                    // packed 8-byte records: measured (-7 us per grid), not bit-equal; only in the experiments build
        case 2: if (packed) { LAUNCH_BIN_P(2, true) } else { LAUNCH_BIN(2) } break;
#else
        case 2: LAUNCH_BIN(2) break;
#endif
        default: LAUNCH_BIN(4) break;
    }
#undef LAUNCH_BIN
#undef LAUNCH_BIN_P
    US_CHECK_LAUNCH("us_hashgrid_bwd_binned");
    return US_OK;
}

extern "C" int us_hashgrid_bwd_binned(const us_grid_desc* d, const float* x, const float* dL_dy, int64_t n, float* grad_params,
                                      int flags, void* workspace, size_t workspace_bytes, void* stream) {
    return bwd_binned(d, x, dL_dy, n, grad_params, flags, workspace, workspace_bytes, stream, false);
}

extern "C" int us_hashgrid_bwd_binned_range(const us_grid_desc* d, const float* x, const float* dL_dy, int64_t n, int64_t plane_stride,
                                            float* grad_params, int flags, void* workspace, size_t workspace_bytes, void* stream) {
    US_REQUIRE(plane_stride >= n, US_ERR_SHAPE, "us_hashgrid_bwd_binned_range: plane_stride %lld < n %lld", (long long)plane_stride, (long long)n);
    US_REQUIRE(!(flags & (US_GRID_BWD_COUNTED | US_GRID_BWD_SCANNED)), US_ERR_CONFIG,
               "us_hashgrid_bwd_binned_range: the counts of a forward pass belong to the whole batch, not to a range of it");
    return bwd_binned(d, x, dL_dy, n, grad_params, flags, workspace, workspace_bytes, stream, false, plane_stride);
}

extern "C" int us_hashgrid_bwd_scan(const us_grid_desc* d, int64_t n, float* grad_params, int flags, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    return bwd_binned(d, nullptr, nullptr, n, grad_params, (flags | US_GRID_BWD_COUNTED) & ~US_GRID_BWD_SCANNED, workspace, workspace_bytes, stream, true);
}

// ---------------------------------------------------------------------------------------------------------------
// Forward pass that also produces the counts of the binning (us_hashgrid_fwd_counted): the encoder is bound by its gathers
// (L2 sector traffic), so the run flags and the 8 LDS counter increments per (point, level) ride along almost for free and
// the backward pass starts at the column scan.  One 1024-thread workgroup = the 1024 points of one k_bin workgroup x one
// level; it writes that level's segment of the workgroup's count row.  Every point counts as live (no gradient exists yet).
// ---------------------------------------------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(BIN_THREADS) void k_fwd_count(LevelTable tab, BinLevels lv, uint32_t n_levels, const float* __restrict__ params,
                                                           const float* __restrict__ x, int64_t n, float* __restrict__ out, int clamp,
                                                           int lm, uint32_t* __restrict__ wg_counts) {
    __shared__ uint32_t lcnt[BIN_MAX_TOTAL];
    __shared__ uint32_t done;
    const uint32_t level = blockIdx.y;
    const BinLevel q = lv.l[level];
    const uint32_t nb = 1u << q.lg;
    for (uint32_t t = threadIdx.x; t < nb; t += BIN_THREADS) lcnt[t] = 0u;
    if (threadIdx.x == 0) done = 0u;
    __syncthreads();
    const LevelGeom g = level_geom(tab, level);
    const typename Feat<F>::T* grid = reinterpret_cast<const typename Feat<F>::T*>(params) + tab.off[level];
    const uint32_t C = n_levels * F;
    const int lane = threadIdx.x & 63, lg16 = lane & (RUN_GROUP - 1);
    const int64_t i = (int64_t)blockIdx.x * BIN_THREADS + threadIdx.x;
    const bool in = i < n;
    float pos[3]; uint32_t cell[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pos_fract(in ? load_x(x, i, k, clamp) : 0.0f, g.scale, pos[k], cell[k]);
    if (in) {
        typename Feat<F>::T v[8];
        gather_corners<F>(g, grid, cell, v);
        float res[F];
#pragma unroll
        for (int f = 0; f < F; ++f) res[f] = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float va[F];
            feat_to_array<F>(v[c], va);
            const float w = corner_weight(c, pos);
#pragma unroll
            for (int f = 0; f < F; ++f) res[f] = fmaf(w, va[f], res[f]);
        }
        float* o = out + feat_index(lm, i, n, level, C, F);
#pragma unroll
        for (int f = 0; f < F; ++f) o[f] = res[f];
    }
    // ---- the counts of k_bin<COUNT> with every point live (same vertex runs, same bins)
    uint32_t key[8];
    slot_keys(cell, in && q.packable, lane, key);
    const uint2 ht = slot_run_masks(key, lg16);                      // outside any condition on `in`: its row shifts read the neighbour lanes
    const uint32_t tail = in ? ht.y : 0u;
    if (tail != 0u) {
        const uint32_t nbm = nb - 1u;
        uint32_t idx[8];
        slot_entries(q.hashed != 0u, q.hs, q.res, q.res2, cell, idx);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if ((tail >> c) & 1u) atomicAdd(&lcnt[(idx[c] >> BIN_LINE_LOG2) & nbm], 1u);
    }
    // No closing barrier: a wave that has finished its gathers leaves at once; the LAST wave to arrive (an LDS ticket taken after
    // the wave's own counter atomics, which execute in order) writes the workgroup's row segment.
    // acq_rel at workgroup scope: the wave's counter atomics happen-before its ticket (release), and the last wave's reads of
    // lcnt[] happen-after every ticket it observed (acquire).  At LDS scope this costs an s_waitcnt, no cache operation.
    uint32_t ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(&done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket == BIN_THREADS / 64 - 1) {
        uint32_t* row = wg_counts + (size_t)blockIdx.x * BIN_MAX_TOTAL + q.first;
        for (uint32_t t = (uint32_t)lane; t < nb; t += 64) row[t] = lcnt[t];
    }
}

extern "C" int us_hashgrid_fwd_counted(const us_grid_desc* d, const float* params, const float* x, int64_t n, float* out, int flags,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    US_REQUIRE(d, US_ERR_NULL, "us_hashgrid_fwd_counted: desc is NULL");
    US_REQUIRE(d->n_levels >= 1 && d->n_levels <= US_MAX_LEVELS && (d->n_features == 1 || d->n_features == 2 || d->n_features == 4) &&
               d->n_params == d->offset[d->n_levels] * d->n_features, US_ERR_CONFIG, "us_hashgrid_fwd_counted: bad descriptor");
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(params && x && out && workspace, US_ERR_NULL, "us_hashgrid_fwd_counted: NULL pointer");
    US_REQUIRE(((uintptr_t)params & 15u) == 0 && ((uintptr_t)workspace & 15u) == 0, US_ERR_SHAPE,
               "us_hashgrid_fwd_counted: params and workspace must be 16-byte aligned");
    US_REQUIRE(workspace_bytes >= us_hashgrid_bwd_workspace_bytes(d, n), US_ERR_WORKSPACE,
               "us_hashgrid_fwd_counted: workspace %zu B < %zu B", workspace_bytes, us_hashgrid_bwd_workspace_bytes(d, n));
    BinMap bm;
    const int TB = make_binmap(d, n, &bm);
    US_REQUIRE(TB > 0 && TB <= BIN_MAX_TOTAL, US_ERR_CONFIG, "us_hashgrid_fwd_counted: %d bins > %d", TB, BIN_MAX_TOTAL);
    const LevelTable t = make_table(d);
    const BinLevels lv = make_bin_levels(d, bm);
    const int clamp = flags & US_GRID_CLAMP01, lm = (flags & US_GRID_LEVEL_MAJOR) ? 1 : 0;
    uint32_t* wg_counts = (uint32_t*)workspace + 2 * (BIN_MAX_TOTAL + 64) + 16 + (((size_t)extra_max(d, n) + 15u) & ~(size_t)15u);
    dim3 grid((unsigned)us_cdiv(n, BIN_THREADS), d->n_levels), block(BIN_THREADS);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_FC(F) hipLaunchKernelGGL((k_fwd_count<F>), grid, block, 0, s, t, lv, d->n_levels, params, x, n, out, clamp, lm, wg_counts);
    switch (d->n_features) { case 1: LAUNCH_FC(1) break; case 2: LAUNCH_FC(2) break; default: LAUNCH_FC(4) break; }
#undef LAUNCH_FC
    US_CHECK_LAUNCH("us_hashgrid_fwd_counted");
    return US_OK;
}
