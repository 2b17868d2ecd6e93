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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global store
// (s_waitcnt vmcnt(0)), which would drain the record stores of the previous level twice per level; the stage protocol
// below needs only the LDS reads/writes of all waves to have completed.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
