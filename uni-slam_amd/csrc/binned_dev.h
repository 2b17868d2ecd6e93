// binned_dev.h -- helpers shared by the binned table-gradient kernels (hashgrid_binned.hip: one grid; hashgrid_joint.hip: two grids
// of equal geometry in one pass): the bin <-> entry maps, DPP row shifts, the LDS-only workgroup barrier.
#pragma once
#include "hashgrid_dev.h"

#define BIN_LINE_LOG2 4                  // bins interleave LINES of 16 entries (128 B of F = 2 gradients)
#ifndef ACC_CHUNK
#define ACC_CHUNK 16384                  // records per accumulate workgroup: hotter bins are split over several
#endif
#define ACC_EXTRA_MAX 256                // extra chunks of hot bins listed per launch

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

// ---- DPP row shifts (within rows of 16 lanes): shr: lane i <- lane i-n ; shl: lane i <- lane i+n
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v)));
}
#define DPP_ROW_SHL1 0x101
#define DPP_ROW_SHR(n) (0x110 + (n))

// inclusive scan over the 64 lanes of a wave, DPP only (row scans, then row_bcast:15 / row_bcast:31 carry the row sums over)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += dpp_u32<DPP_ROW_SHR(1)>(v);
    v += dpp_u32<DPP_ROW_SHR(2)>(v);
    v += dpp_u32<DPP_ROW_SHR(4)>(v);
    v += dpp_u32<DPP_ROW_SHR(8)>(v);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);      // rows 1, 3 += last lane of rows 0, 2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);      // rows 2, 3 += lane 31
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store
// (s_waitcnt vmcnt(0)), which would drain the record stores of the previous level twice per level; the stage protocol
// below needs only the LDS reads/writes of all waves to have completed.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------
// VERTEX RUNS.  Consecutive samples of a ray (adjacent lanes) sit in the same cell, or in neighbouring cells that share 4, 2 or 1
// of their 8 vertices.  The 8 contributions of a lane are therefore kept by the PARITY CLASS of the vertex they go to,
// slot p = (vx & 1) | (vy & 1) << 1 | (vz & 1) << 2 of the absolute vertex coordinates: a cell's 8 vertices take all 8 classes, and a
// vertex shared by the cells of two lanes sits in the same slot of both.  Per slot, lanes of an aligned group of RUN_GROUP (= one DPP
// row) whose slot holds the same vertex form a run; a segmented scan sums the run into its last lane, which alone emits a record.
// Along a line the cells that touch a given vertex are consecutive, so this finds every repeat inside a group: on the bench workload
// 4 296 records per ray and grid instead of 5 580 with runs of equal CELLS in groups of 8 (8 192 without combining).
// Exact for any point order: only equal vertices in adjacent lanes are merged.
// ---------------------------------------------------------------------------------------------------------------
#ifndef RUN_GROUP
#define RUN_GROUP 16
#endif

struct SlotGeom {
    uint32_t idx[8];          // entry index of the slot's vertex inside the level
    uint32_t key[8];          // identity of the slot's vertex (packed coordinates), or a per-lane unique value
};

// cell -> per-slot vertex keys.  live == false or coordinates that do not fit 10 bits each: keys no other lane can equal.
__device__ __forceinline__ void slot_keys(const uint32_t cell[3], bool mergeable, int lane, uint32_t (&key)[8]) {
    uint32_t k[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t cb = cell[a] & 1u;
        k[a][0] = ((cell[a] + cb) & 1023u) << (10 * a);            // the even vertex coordinate of the cell's edge along axis a
        k[a][1] = ((cell[a] + 1u - cb) & 1023u) << (10 * a);       // the odd one
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) key[p] = mergeable ? (k[0][p & 1] | k[1][(p >> 1) & 1] | k[2][p >> 2]) : (0xC0000000u | (uint32_t)lane);
}

// entry indices by slot (the arithmetic of grid_index: coherent prime hash & (hs-1), or x + y*res + z*res^2 wrapped once at hs)
__device__ __forceinline__ void slot_entries(bool hashed, uint32_t hs, uint32_t res, uint32_t res2, const uint32_t cell[3], uint32_t (&idx)[8]) {
    const uint32_t cbx = cell[0] & 1u, cby = cell[1] & 1u, cbz = cell[2] & 1u;
    if (hashed) {                                                // wave-uniform
        const uint32_t hy0 = cell[1] * 2654435761u, hz0 = cell[2] * 805459861u;
        const uint32_t hx[2] = {cell[0] + cbx, cell[0] + 1u - cbx};
        const uint32_t hy[2] = {cby ? hy0 + 2654435761u : hy0, cby ? hy0 : hy0 + 2654435761u};
        const uint32_t hz[2] = {cbz ? hz0 + 805459861u : hz0, cbz ? hz0 : hz0 + 805459861u};
        const uint32_t mask = hs - 1u;
#pragma unroll
        for (int p = 0; p < 8; ++p) idx[p] = (hx[p & 1] ^ hy[(p >> 1) & 1] ^ hz[p >> 2]) & mask;
    } else {
        const uint32_t base = cell[0] + __umul24(cell[1], res) + __umul24(cell[2], res2);
        const uint32_t sx[2] = {cbx, 1u - cbx};
        const uint32_t sy[2] = {cby ? res : 0u, cby ? 0u : res};
        const uint32_t sz[2] = {cbz ? res2 : 0u, cbz ? 0u : res2};
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const uint32_t e = base + sx[p & 1] + sy[(p >> 1) & 1] + sz[p >> 2];
            idx[p] = min(e, e - hs);                             // e < 2*hs: one conditional subtraction == e % hs
        }
    }
}

// trilinear weights by slot, in tcnn's multiplication order ((1*ax)*ay)*az for that vertex
__device__ __forceinline__ void slot_weights(const float pos[3], const uint32_t cell[3], float (&w)[8]) {
    float a[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool cb = (cell[k] & 1u) != 0u;                    // slot bit 0 <-> offset cb, slot bit 1 <-> offset 1 - cb
        const float lo = 1.0f - pos[k], hi = pos[k];
        a[k][0] = cb ? hi : lo; a[k][1] = cb ? lo : hi;
    }
    float wxy[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) wxy[c] = a[0][c & 1] * a[1][c >> 1];
#pragma unroll
    for (int p = 0; p < 8; ++p) w[p] = wxy[p & 3] * a[2][p >> 2];
}

// head / tail masks (bit p: slot p starts / ends a run in this lane), returned as {head, tail}.  To be called with the WHOLE WAVE
// active (never as one arm of `in ? ... : ...`): the tail mask of a lane is the head mask of the lane above it.
__device__ __forceinline__ uint2 slot_run_masks(const uint32_t (&key)[8], int lg16) {
    uint32_t head = 0u;
#pragma unroll
    for (int p = 0; p < 8; ++p) head |= (dpp_u32<DPP_ROW_SHR(1)>(key[p]) != key[p]) ? (1u << p) : 0u;
    if (lg16 == 0) head = 0xFFu;
    uint32_t tail = dpp_u32<DPP_ROW_SHL1>(head);                 // a run ends where the next lane starts one
    if (lg16 == RUN_GROUP - 1) tail = 0xFFu;
    return make_uint2(head, tail);
}

// segmented inclusive scan of NV values per slot over the runs (Hillis-Steele, DPP row_shr 1/2/4/8 fused into v_fmac): afterwards the
// last lane of every run holds the run's sum
#define SLOT_SCAN_STEP(O, NV)                                                                                        \
    if (__ballot((head != 0xFFu) && (lg16 >= (O))) != 0ull) {    /* wave-uniform: nothing left to merge -> skip the step */ \
        const uint32_t take = (lg16 >= (O)) ? (~head & 0xFFu) : 0u;                                                  \
        const uint32_t hprev = dpp_u32<DPP_ROW_SHR(O)>(head);                                                        \
        _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                             \
            const float takef = (take >> p) & 1u ? 1.0f : 0.0f;                                                      \
            _Pragma("unroll") for (int f = 0; f < (NV); ++f)                                                        \
                val[p][f] = fmaf(dpp_f32<DPP_ROW_SHR(O)>(val[p][f]), takef, val[p][f]);   /* t*1+v == t+v */         \
        }                                                                                                            \
        head |= (lg16 >= (O)) ? hprev : 0u;                                                                          \
    }
#define SLOT_SCAN(NV) SLOT_SCAN_STEP(1, NV) SLOT_SCAN_STEP(2, NV) SLOT_SCAN_STEP(4, NV) SLOT_SCAN_STEP(8, NV)

// The same scan in two parts, for kernels that run it for several value sets over the SAME runs (hashgrid_joint.hip: two grids):
// SLOT_SCAN_PRE evolves the head masks once and records, per step s, which slots take from the lane O = 1 << s below (take_all, 8 bits
// per step) and whether any lane of the wave takes at all (steps, wave-uniform); SLOT_SCAN_APPLY replays that on NV values per slot.
#define SLOT_SCAN_PRE(head, take_all, steps)                                                                         \
    {                                                                                                                \
        uint32_t h_ = (head);                                                                                        \
        (take_all) = 0u; (steps) = 0u;                                                                               \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                          \
            const int O_ = 1 << s_;                                                                                  \
            const uint32_t take_ = (lg16 >= O_) ? (~h_ & 0xFFu) : 0u;                                                \
            uint32_t hp_;                                                                                            \
            if (s_ == 0) hp_ = dpp_u32<DPP_ROW_SHR(1)>(h_); else if (s_ == 1) hp_ = dpp_u32<DPP_ROW_SHR(2)>(h_);      \
            else if (s_ == 2) hp_ = dpp_u32<DPP_ROW_SHR(4)>(h_); else hp_ = dpp_u32<DPP_ROW_SHR(8)>(h_);              \
            (take_all) |= take_ << (8 * s_);                                                                         \
            if (__ballot(take_ != 0u) != 0ull) (steps) |= 1u << s_;                                                  \
            h_ |= (lg16 >= O_) ? hp_ : 0u;                                                                           \
        }                                                                                                            \
    }
#define SLOT_SCAN_APPLY_STEP(S, O, NV, take_all, steps)                                                              \
    if ((steps) & (1u << (S))) {                                 /* wave-uniform */                                  \
        _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                             \
            const float takef = ((take_all) >> (8 * (S) + p)) & 1u ? 1.0f : 0.0f;                                    \
            _Pragma("unroll") for (int f = 0; f < (NV); ++f)                                                        \
                val[p][f] = fmaf(dpp_f32<DPP_ROW_SHR(O)>(val[p][f]), takef, val[p][f]);                              \
        }                                                                                                            \
    }
#define SLOT_SCAN_APPLY(NV, take_all, steps)                                                                         \
    SLOT_SCAN_APPLY_STEP(0, 1, NV, take_all, steps) SLOT_SCAN_APPLY_STEP(1, 2, NV, take_all, steps)                  \
    SLOT_SCAN_APPLY_STEP(2, 4, NV, take_all, steps) SLOT_SCAN_APPLY_STEP(3, 8, NV, take_all, steps)

// SLOT_SCAN_APPLY for TWO values per slot, written out: per step and slot v_bfe_u32 + v_cvt_f32_ubyte0 (take bit -> 0.0 / 1.0) and
// one v_fmac_f32 per value with the row shift on its first operand (val += shifted(val) * take) -- 32 vector instructions per step.
// The compiler pairs the two values of a slot into a v_pk_fma_f32, which cannot carry DPP, and pays a v_mov_b32_dpp per value and
// three instructions per take bit for it: 48 per step.  A step no lane of the wave takes in is skipped (s_cbranch on its bit of
// `steps`); the leading s_nop covers the two wait states between a VALU write of a register and its DPP read, which the compiler does
// not see through an asm block.  Operands: %0..%15 = val[p][f], %16 %17 = temporaries, %18 = take_all, %19 = steps.
#define RS_FMAC(i, t, O) "v_fmac_f32_dpp %" #i ", %" #i ", %" #t " row_shr:" #O " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define RS_SLOT(i0, i1, t, bit, O) "v_bfe_u32 %" #t ", %18, " #bit ", 1\n\tv_cvt_f32_ubyte0_e32 %" #t ", %" #t "\n\t" RS_FMAC(i0, t, O) RS_FMAC(i1, t, O)
#define RS_STEP(S, O, b0, b1, b2, b3, b4, b5, b6, b7)                                                                \
    "s_bitcmp1_b32 %19, " #S "\n\ts_cbranch_scc0 .Lrs" #S "_%=\n\t"                                                  \
    RS_SLOT(0, 1, 16, b0, O) RS_SLOT(2, 3, 17, b1, O) RS_SLOT(4, 5, 16, b2, O) RS_SLOT(6, 7, 17, b3, O)              \
    RS_SLOT(8, 9, 16, b4, O) RS_SLOT(10, 11, 17, b5, O) RS_SLOT(12, 13, 16, b6, O) RS_SLOT(14, 15, 17, b7, O)        \
    ".Lrs" #S "_%=:\n\t"
__device__ __forceinline__ void slot_scan_apply_pairs(float (&val)[8][2], uint32_t take_all, uint32_t steps) {
    float t0, t1;
    steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)steps);                              // wave-uniform by construction: into an SGPR
    asm("s_nop 1\n\t"
        RS_STEP(0, 1, 0, 1, 2, 3, 4, 5, 6, 7) RS_STEP(1, 2, 8, 9, 10, 11, 12, 13, 14, 15)
        RS_STEP(2, 4, 16, 17, 18, 19, 20, 21, 22, 23) RS_STEP(3, 8, 24, 25, 26, 27, 28, 29, 30, 31)
        : "+v"(val[0][0]), "+v"(val[0][1]), "+v"(val[1][0]), "+v"(val[1][1]), "+v"(val[2][0]), "+v"(val[2][1]), "+v"(val[3][0]),
          "+v"(val[3][1]), "+v"(val[4][0]), "+v"(val[4][1]), "+v"(val[5][0]), "+v"(val[5][1]), "+v"(val[6][0]), "+v"(val[6][1]),
          "+v"(val[7][0]), "+v"(val[7][1]), "=&v"(t0), "=&v"(t1)
        : "v"(take_all), "s"(steps)
        : "scc");
}
