/*
 * unislam_hip_experiments.h -- entry points and flags of the EXPERIMENTS build of libunislam_hip.so (tools/build_experiments.sh:
 * make EXTRA=-DUS_EXPERIMENTS).  Variants that were built, held to the same parity tests as the shipped kernels and MEASURED SLOWER on
 * MI355X (DESIGN.md 5d / 5e has the numbers); they are kept buildable so that the measurements can be repeated, and are not part of the
 * shipped library or of the drop-in surface.  tests/test_gpu_experiments.py runs their parity tests when the loaded library exports them.
 */
#ifndef UNISLAM_HIP_EXPERIMENTS_H
#define UNISLAM_HIP_EXPERIMENTS_H
#include "unislam_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* US_GRID_BWD_PACKED (us_hashgrid_bwd_binned, n_features == 2 only): the intermediate records of the binned pass are kept in 8 bytes
 * { local entry, the two contributions rounded to 26 / 27 significant fp32 bits } instead of 12; the sums are still formed in f64.
 * Relative rounding per contribution <= 2^-18: far inside the 1e-3 parity bound, but not bit-equal to the unpacked pass.
 * Measured on MI355X (4096 x 64 points, room0 tables): 186 -> 179 us per colour-table gradient -- the two passes are bound by LDS
 * atomics and per-workgroup latency, not by their bytes, so the default keeps the exact 12-byte records.  (Round 1's reading of the
 * one-grid kernels.  Round 3's timing builds of the two-grid accumulate pass show the opposite -- that pass IS its record stream -- and
 * the two-grid kernels now write exact 10-byte records in two aligned planes: DESIGN.md 5f.) */
#define US_GRID_BWD_PACKED 32

/* Render-only encode + decode in one launch: outA = decoder A(grid A(x)), outB = decoder B(grid B(x)) -- what Decoders.forward
 * (src/networks/decoders.py:158-188) computes for Renderer.render_img / Mesher.eval_points, calls that need no gradient.  The features
 * stay in LDS.  Needs two 16-level F = 2 grids of equal base resolution and per-level scale, and two bf16 decoders (US_PREC_BF16 or
 * US_PREC_BF16_PLAIN, the same for both) of equal width and depth: us_encode_decode_supported() says whether a pair qualifies; results
 * are bit-identical to us_hashgrid_fwd + us_mlp_fwd.  flags: US_GRID_CLAMP01. */
int us_encode_decode_supported(const us_grid_desc* a, const us_grid_desc* b, const us_mlp_desc* ma, const us_mlp_desc* mb);
int us_encode_decode_fwd(const us_grid_desc* a, const us_grid_desc* b, const float* paramsA, const float* paramsB,
                         const us_mlp_desc* ma, const us_mlp_desc* mb, const float* mlp_paramsA, const float* mlp_paramsB,
                         const float* x, int64_t n, float* outA, int64_t strideA, float* outB, int64_t strideB, int flags, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* UNISLAM_HIP_EXPERIMENTS_H */
