/*
 * unislam_hip.h -- C ABI of libunislam_hip.so: the MI355X (gfx950) implementation of Uni-SLAM's
 * per-iteration volumetric-rendering hot path.
 *
 * The reference (dfki-av/Uni-SLAM) is pure Python and has no FFI of its own for this path; its native
 * compute comes from tiny-cuda-nn (tcnn.Encoding / tcnn.Network torch modules) and ATen ops.  Each entry
 * point below replaces the reference call site(s) it cites (paths relative to the reference tree).  The
 * Python host side (package uni-slam_amd/, importable as `unislam_amd`) binds these through ctypes and
 * mirrors the reference's module/function API; INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative US_ERR_* for argument errors, or a positive hipError_t;
 *     no C++ exception crosses the boundary; us_last_error() gives a thread-local message.
 *   - all pointers are DEVICE pointers (tensor.data_ptr()) unless the name ends in _host; fp32, row-major,
 *     contiguous.  The caller (PyTorch) owns every buffer incl. workspaces and keeps them alive until the
 *     stream work completes; the library never allocates or frees device memory and holds no mutable state.
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); launches are asynchronous and
 *     never synchronise, so every call can be captured into a hipGraph.
 *   - N = number of points = rays * samples; C = n_levels * n_features (32 for Uni-SLAM).
 */
#ifndef UNISLAM_HIP_H
#define UNISLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define US_MAX_LEVELS 32
#define US_ABI_VERSION 2          /* 2 (r6): us_hashgrid_bwd_joint_adam / _part left the shipped ABI (unislam_hip_experiments.h); new flags
                                   * US_GRID_FEAT_SPLIT_BF16, US_MLP_IN_SPLIT_BF16 */

enum {
    US_OK = 0,
    US_ERR_NULL = -1,       /* required pointer is NULL */
    US_ERR_SHAPE = -2,      /* size / shape outside what the kernels support */
    US_ERR_CONFIG = -3,     /* unsupported descriptor (features per level, width, ...) */
    US_ERR_WORKSPACE = -4   /* workspace too small */
};

const char* us_last_error(void);
int us_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Multi-resolution hash grid  (replaces tcnn.Encoding "HashGrid": construction src/UNISLAM.py:242-253,
 * call src/networks/decoders.py:103, backward via autograd at src/Mapper.py:444 / src/Tracker.py:241)
 * ---------------------------------------------------------------------------------------------- */
typedef struct us_grid_desc {
    uint32_t n_levels;            /* L  (<= US_MAX_LEVELS) */
    uint32_t n_features;          /* F  per level: 1, 2 or 4 */
    uint32_t log2_hashmap_size;
    uint32_t base_resolution;
    float    per_level_scale;
    float    scale[US_MAX_LEVELS];        /* exp2f(l*log2f(pls))*base - 1 */
    uint32_t resolution[US_MAX_LEVELS];   /* ceilf(scale)+1 */
    uint32_t offset[US_MAX_LEVELS + 1];   /* level start, in table ENTRIES (F floats each) */
    uint32_t n_params;                    /* F * offset[L] : length of the flat fp32 `params` vector */
} us_grid_desc;

/* host only: fills scale/resolution/offset/n_params exactly as tcnn's GridEncoding constructor does */
int us_grid_desc_init(us_grid_desc* desc_host, uint32_t n_levels, uint32_t n_features,
                      uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale);

/* flags for the three grid entry points: US_GRID_CLAMP01 folds decoders.py:101's torch.clamp(p, 0, 1) into the
 * position load (and zeroes dy_dx where the clamp is active, like torch.clamp's backward) */
#define US_GRID_CLAMP01 1
/* US_GRID_LEVEL_MAJOR: `out` of us_hashgrid_fwd / `dL_dy` of us_hashgrid_bwd_params are laid out [L][N][F] (each level's
 * plane contiguous) instead of the torch view [N][L*F]; dy_dx is not affected */
#define US_GRID_LEVEL_MAJOR 2
/* US_GRID_BWD_OVERWRITE (us_hashgrid_bwd_binned only): every entry of grad_params is WRITTEN (zero where no sample
 * contributed) instead of added to, so the caller need not clear the table gradient beforehand (what optimizer.zero_grad()
 * does at src/Mapper.py:443 before loss.backward()). */
#define US_GRID_BWD_OVERWRITE 4
/* US_GRID_BWD_COUNTED (us_hashgrid_bwd_binned only): the workspace already holds the binning counts of these x, written
 * by us_hashgrid_fwd_counted with the same desc / n / clamp flag; the counting pass is skipped. */
#define US_GRID_BWD_COUNTED 16
/* US_GRID_BWD_SCANNED (with US_GRID_BWD_COUNTED): us_hashgrid_bwd_scan has already run on this workspace, the gradient call starts at
 * its record pass. */
#define US_GRID_BWD_SCANNED 64
/* (flag value 32, US_GRID_BWD_PACKED, belongs to the experiments build: include/unislam_hip_experiments.h) */
/* US_GRID_BWD_DETERMINISTIC (us_hashgrid_bwd_binned, us_hashgrid_bwd_joint, and their scan calls): no bin is split over several
 * accumulate workgroups, so no float atomic takes part: every table-gradient entry is ONE f64 sum of its fp32 contributions, rounded
 * to fp32 once.  The order in which LDS atomics add the terms can then only move the sum at the 2^-53 level, which the final rounding
 * does not see: results repeat bit for bit from run to run (tests/test_gpu_joint.py).  Slower when the batch concentrates on few
 * cells (one workgroup then walks a hot bin alone); meant for diffing runs. */
#define US_GRID_BWD_DETERMINISTIC 128
/* US_GRID_BWD_ONLY_A / US_GRID_BWD_ONLY_B (us_hashgrid_bwd_joint): the record pass for both grids, the accumulate pass -- and so the
 * finished gradient table -- for ONE of them; a second call with the other flag and US_GRID_BWD_RECORDS_READY (same workspace, nothing
 * in between) sums the other grid from the records the first call left.  The data-parallel step finishes the colour table first and
 * lets its all-reduce travel while the sdf table is summed (src/Mapper.py:444 on N ranks).  Same results as one call. */
#define US_GRID_BWD_ONLY_A 256
#define US_GRID_BWD_ONLY_B 512
#define US_GRID_BWD_RECORDS_READY 1024
/* US_GRID_FEAT_SPLIT_BF16 (us_hashgrid_fwd_joint / _dydx, with US_GRID_LEVEL_MAJOR): the feature planes hold, per point and level, the pair
 * {bf16 hi(f0) | bf16 hi(f1) << 16, bf16 lo(f0) | bf16 lo(f1) << 16} (hi = bf16(f), lo = bf16(f - hi)) instead of two floats -- the same 8
 * bytes, the operand images of the split-operand decoders (US_PREC_BF16), which then take them with US_MLP_IN_SPLIT_BF16 and skip the
 * split of their inputs: results bit-identical to float planes.  Only the decoders read such planes. */
#define US_GRID_FEAT_SPLIT_BF16 2048
/* US_GRID_ACCUMULATE (us_hashgrid_bwd_input_gather only): dL_dx += instead of = (the second grid adds to the first) */
#define US_GRID_ACCUMULATE 8

/* out[N][C] = encode(x[N][3]);  dy_dx[N][C][3] optional (NULL when positions need no gradient) */
int us_hashgrid_fwd(const us_grid_desc* desc_host, const float* params, const float* x, int64_t n,
                    float* out, float* dy_dx, int flags, void* stream);

/* 8 corner indices per (point, level): idx[N][L][8] uint32 (entry index inside the level) -- parity/debug */
int us_hashgrid_indices(const us_grid_desc* desc_host, const float* x, int64_t n, uint32_t* idx, int flags,
                        void* stream);

/* grad_params[n_params] += scatter(dL_dy[N][C]).  Caller zeroes grad_params when it wants a fresh gradient.
 * mode 0: direct global float atomics (tcnn's kernel_grid_backward shape);
 * mode 1: LDS-privatised table slices (one workgroup accumulates a slice of one level over all points
 *         in LDS, then flushes it with contiguous atomics);  mode 2: the same with per-wave compaction queues;
 * mode -1: pick per level by shape (slices where a level receives more updates than it has entries). */
int us_hashgrid_bwd_params(const us_grid_desc* desc_host, const float* x, const float* dL_dy, int64_t n,
                           float* grad_params, int mode, int flags, void* stream);

/* Same result as us_hashgrid_bwd_params, by "bin once, accumulate in f64" (csrc/hashgrid_binned.hip): every (point,
 * level) is hashed a constant number of times, contributions of neighbouring samples in the same cell are combined in
 * registers, records are binned by entry index mod n_bins and each bin is summed in LDS with ds_add_f64.  Needs a caller-
 * provided workspace of us_hashgrid_bwd_workspace_bytes(desc, n) bytes (scratch: contents undefined afterwards).
 * This is the path MapStep uses for the 4096 x 64 mapping batch. */
size_t us_hashgrid_bwd_workspace_bytes(const us_grid_desc* desc_host, int64_t n);
/* 1 if us_hashgrid_bwd_binned takes this table and batch (bin budget: 4096 bins of <= 2048 entries; 32-bit record offsets:
 * n * 8 * L * 12 B < 4 GiB), else 0 -- then us_hashgrid_bwd_params serves */
int us_hashgrid_bwd_binned_supported(const us_grid_desc* desc_host, int64_t n);
int us_hashgrid_bwd_binned(const us_grid_desc* desc_host, const float* x, const float* dL_dy, int64_t n,
                           float* grad_params, int flags, void* workspace, size_t workspace_bytes, void* stream);
/* The two scan passes of us_hashgrid_bwd_binned, run ahead of the gradient call on a workspace whose counts are in place
 * (us_hashgrid_fwd_counted of the same x): they depend on the counts only, so the mapping step issues them right after the encoder, off
 * the backward pass's critical path.  flags / grad_params as in the gradient call that follows (with US_GRID_BWD_OVERWRITE the entries
 * of bins that several workgroups will add into are cleared HERE: grad_params must not be read between this call and the gradient
 * call).  Then call us_hashgrid_bwd_binned with US_GRID_BWD_COUNTED | US_GRID_BWD_SCANNED. */
int us_hashgrid_bwd_scan(const us_grid_desc* d, int64_t n, float* grad_params, int flags, void* workspace, size_t workspace_bytes,
                         void* stream);

/* us_hashgrid_fwd (without dy_dx) that also leaves the counts of the binned backward in `workspace` (the buffer later given to
 * us_hashgrid_bwd_binned together with US_GRID_BWD_COUNTED; same size).  The encoder is bound by its gathers, so the counting
 * rides along and the backward pass of the same points starts one kernel later. */
int us_hashgrid_fwd_counted(const us_grid_desc* desc_host, const float* params, const float* x, int64_t n, float* out,
                            int flags, void* workspace, size_t workspace_bytes, void* stream);

/* ---- both grids of Uni-SLAM in one pass per direction (csrc/hashgrid_joint.hip).  src/networks/decoders.py:118,143 encode the SAME
 * points with the sdf and the colour table, which src/UNISLAM.py:241-253 builds from one base resolution and one per-level scale:
 * cells, fractional positions, the runs of a ray's samples and the vertex hashes coincide, only log2_hashmap_size differs.
 * `a` / `b`: the two descriptors (F = 2, <= 16 levels, equal scale[] / resolution[]); levels where both tables are dense with equal
 * size, or both hashed with a's size <= b's, share one bin, one cursor and one stage position per vertex contribution; other levels
 * (one dense, one hashed) keep bins of their own per grid.  The intermediate records are exact: 10 bytes each (16-bit bin-local entry,
 * two f32 values) in two aligned planes.  Results are those of us_hashgrid_fwd / us_hashgrid_bwd_binned on each grid. */
int us_hashgrid_joint_supported(const us_grid_desc* a, const us_grid_desc* b, int64_t n);
size_t us_hashgrid_joint_workspace_bytes(const us_grid_desc* a, const us_grid_desc* b, int64_t n);
/* outA / outB = encode(x) with table a / b.  flags: US_GRID_CLAMP01, US_GRID_LEVEL_MAJOR.  workspace (nullable): when given
 * (us_hashgrid_joint_workspace_bytes), the binning counts of these points are left in it for us_hashgrid_bwd_joint + US_GRID_BWD_COUNTED. */
int us_hashgrid_fwd_joint(const us_grid_desc* a, const us_grid_desc* b, const float* paramsA, const float* paramsB, const float* x,
                          int64_t n, float* outA, float* outB, int flags, void* workspace, size_t workspace_bytes, void* stream);
/* gradA / gradB (+)= scatter(dL_dyA / dL_dyB): the table gradients of both grids (autograd backward of the two encoders at
 * src/Mapper.py:444).  dL_dy*: level-major planes [L][N][2] (US_GRID_LEVEL_MAJOR is required).  flags: US_GRID_CLAMP01,
 * US_GRID_LEVEL_MAJOR, US_GRID_BWD_OVERWRITE, US_GRID_BWD_COUNTED.  Every point counts as carrying a gradient (zero gradients
 * produce zero records), so the counts depend on x only. */
int us_hashgrid_bwd_joint(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA, const float* dL_dyB,
                          int64_t n, float* gradA, float* gradB, int flags, void* workspace, size_t workspace_bytes, void* stream);
/* us_hashgrid_bwd_joint that ALSO leaves grid B's gradient as a bfloat16 image (gradB_bf16: uint16 [n_params of b], same indexing, rounded to
 * nearest even as a tensor copy rounds): the payload of a data-parallel all-reduce comes out of the accumulate pass's sweep instead of a
 * narrowing pass over the table (44.7 MB read + 22 MB written per step at room0's sizes).  Needs US_GRID_BWD_OVERWRITE and
 * US_GRID_BWD_DETERMINISTIC (in the scan call too): every entry is written once, by the workgroup that owns it. */
int us_hashgrid_bwd_joint_img(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA, const float* dL_dyB, int64_t n,
                              float* gradA, float* gradB, uint16_t* gradB_bf16, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* The table gradient of a RANGE of the batch: x / dL_dy* point at the range's first point, n = points in the range, plane_stride =
 * points of the whole batch (the distance between the level planes of a level-major dL_dy).  The gradient is additive over points,
 * so a batch whose scratch (us_hashgrid_*_workspace_bytes grows with n) would exceed a budget is walked in ranges -- the first with
 * US_GRID_BWD_OVERWRITE, the others adding -- on a workspace sized for one range.  No US_GRID_BWD_COUNTED here. */
int us_hashgrid_bwd_joint_range(const us_grid_desc* a, const us_grid_desc* b, const float* x, const float* dL_dyA, const float* dL_dyB,
                                int64_t n, int64_t plane_stride, float* gradA, float* gradB, int flags, void* workspace,
                                size_t workspace_bytes, void* stream);
int us_hashgrid_bwd_binned_range(const us_grid_desc* desc_host, const float* x, const float* dL_dy, int64_t n, int64_t plane_stride,
                                 float* grad_params, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* The two scan passes of us_hashgrid_bwd_joint, run ahead of the gradient call on a workspace whose counts us_hashgrid_fwd_joint left
 * (they depend on the counts only, so the mapping step issues them beside the decoders' forward pass).  flags / gradA / gradB as in the
 * gradient call that follows, which then carries US_GRID_BWD_COUNTED | US_GRID_BWD_SCANNED.  With US_GRID_BWD_OVERWRITE the entries of
 * bins that several workgroups will add into are cleared HERE: the gradient tables must not be read in between. */
int us_hashgrid_joint_scan(const us_grid_desc* a, const us_grid_desc* b, int64_t n, float* gradA, float* gradB, int flags,
                           void* workspace, size_t workspace_bytes, void* stream);

/* dL_dx[N][3] = sum_k dL_dy[N][k] * dy_dx[N][k][:]   (tcnn kernel_grid_backward_input) */
int us_hashgrid_bwd_input(const float* dL_dy, const float* dy_dx, int64_t n, uint32_t n_out_features,
                          float* dL_dx, void* stream);
/* the same value for value WITHOUT a stored dy_dx: the 8 vertices of every (point, level) are gathered again and dy/dx is
 * formed on the fly (saves the [N][C][3] tensor the forward would write).  Gradient of the rays / camera poses:
 * src/Tracker.py:170-174,241 and the joint pose optimisation of src/Mapper.py:358-374,442-459.
 * flags: US_GRID_CLAMP01, US_GRID_LEVEL_MAJOR (layout of dL_dy), US_GRID_ACCUMULATE. */
int us_hashgrid_bwd_input_gather(const us_grid_desc* desc_host, const float* params, const float* x, const float* dL_dy,
                                 int64_t n, float* dL_dx, int flags, void* stream);

/* us_hashgrid_fwd_joint that also leaves d(features)/d(position) of both grids: dy_dxA / dy_dxB = planes [L][3][N][2] of IEEE HALF
 * values (level l, dimension gd, point i: the two features' derivatives; tcnn's dy_dx with the clamp's zero gradient folded in), 12 B
 * per (point, level, grid), written in whole lines by the threads that hold the 8 vertices anyway.  us_hashgrid_dydx_rays contracts them
 * with dL/dy and reduces to the rays: dL_do, dL_dd (and the per-point dL_dx, nullable) equal us_hashgrid_bwd_input_rays to the rounding
 * of the stored halves (2^-11 per value; a scratch tensor between two kernels of one iteration, never a result), at the price of a stream
 * instead of a second gather pass over the tables (MI355X, 2000 x 40 points: 20 against 65 us).  For the iterations that differentiate with respect to
 * the camera: src/Tracker.py:170-174,241 and src/Mapper.py:372-376,444. */
typedef uint16_t us_half_t;              /* the bits of an IEEE binary16 value (torch.float16) */
int us_hashgrid_fwd_joint_dydx(const us_grid_desc* a, const us_grid_desc* b, const float* paramsA, const float* paramsB, const float* x,
                               int64_t n, float* outA, float* outB, us_half_t* dy_dxA, us_half_t* dy_dxB, int flags, void* workspace,
                               size_t workspace_bytes, void* stream);
int us_hashgrid_dydx_rays(uint32_t n_levels, const float* dL_dyA, const float* dL_dyB, const us_half_t* dy_dxA, const us_half_t* dy_dxB,
                          int64_t n_rays, int n_samples, const float* z_vals, const float* bound_host, float* dL_do, float* dL_dd,
                          float* dL_dx, void* stream);

/* Both grids' input gradient reduced to the rays in ONE launch: dL_do[R][3], dL_dd[R][3] = adjoint of us_ray_points applied to
 * dL/dx = us_hashgrid_bwd_input_gather(a) + us_hashgrid_bwd_input_gather(b) -- what the camera pose receives in
 * Tracker.optimize_tracking (src/Tracker.py:170-174,241) and in the joint pose optimisation of Mapper.optimize_mapping
 * (src/Mapper.py:372-376,444).  x[R*S][3] unit-cube points of R rays x S samples, z_vals[R][S]; dL_dx (nullable): the per-point
 * gradient as well (bit-identical to the two one-grid launches).  Two F = 2 grids of equal depth, S <= 128
 * (us_hashgrid_bwd_input_rays_supported).  flags: US_GRID_CLAMP01, US_GRID_LEVEL_MAJOR. */
int us_hashgrid_bwd_input_rays_supported(const us_grid_desc* a, const us_grid_desc* b, int n_samples);
int us_hashgrid_bwd_input_rays(const us_grid_desc* a, const us_grid_desc* b, const float* paramsA, const float* paramsB, const float* x,
                               const float* dL_dyA, const float* dL_dyB, int64_t n_rays, int n_samples, const float* z_vals,
                               const float* bound_host, float* dL_do, float* dL_dd, float* dL_dx, int flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused tiny MLP on MFMA  (replaces tcnn.Network "FullyFusedMLP" src/networks/decoders.py:50-70,123,148
 * and the nn.Linear stacks src/networks/decoders.py:74-84,125-128,150-153)
 * ---------------------------------------------------------------------------------------------- */
enum { US_ACT_NONE = 0, US_ACT_TANH = 1, US_ACT_SIGMOID = 2 };
/* MFMA operand type; accumulation, parameters and parameter gradients are always fp32.
 * US_PREC_BF16: bf16 MFMA (v_mfma_f32_16x16x32_bf16) with SPLIT operands in the forward products (x = hi + lo, W x = W_hi x_hi +
 * W_hi x_lo + W_lo x_hi: operands good to 2^-16), plain bf16 operands in the gradient products; US_PREC_BF16_PLAIN: one MFMA per
 * product everywhere (operands good to 2^-8: rendered colour deviates 1.4e-3 from the fp32 decoders); US_PREC_F16: f16 MFMA
 * (v_mfma_f32_16x16x32_f16), ONE product per layer, operands good to 2^-11 -- the arithmetic of the reference's tcnn FullyFusedMLP
 * (src/networks/decoders.py:50-70) --, inputs pre-scaled by a constant and the gradient chain by a per-chunk power of two so that
 * f16's exponent range does not reach the results (csrc/mlp_bf16.inc). */
enum { US_PREC_F32 = 0, US_PREC_BF16 = 1, US_PREC_BF16_PLAIN = 2, US_PREC_F16 = 3 };

typedef struct us_mlp_desc {
    uint32_t n_in;        /* 32 */
    uint32_t width;       /* 16, 32 or 64 hidden neurons */
    uint32_t n_hidden;    /* number of hidden layers: 1 or 2  (n_hidden+1 matrices) */
    uint32_t n_out;       /* 1..16 real outputs (rows n_out..15 of the last matrix are padding) */
    uint32_t out_act;     /* US_ACT_* */
    uint32_t has_bias;    /* 0: tcnn layout (weights only)   1: biases appended after the weights */
    uint32_t precision;   /* US_PREC_* */
} us_mlp_desc;
/* params (flat fp32): W0[width][n_in], (n_hidden-1) x W[width][width], Wlast[16][width]   (row-major [out][in]),
 * then if has_bias: b0[width], (n_hidden-1) x b[width], blast[16]. */
size_t us_mlp_n_params(const us_mlp_desc* d);

/* flags: US_MLP_LEVEL_MAJOR -> `in` (and `dL_din`) are the hash grid's level-major [16][N][2] planes (feature k lives in
 * plane k/2, component k%2) instead of row-major [N][32] */
#define US_MLP_LEVEL_MAJOR 1
/* US_MLP_DEFER_REDUCE (us_mlp_bwd with a workspace): leave the per-workgroup partial weight gradients in the workspace; the caller adds
 * them with us_mlp_reduce later -- e.g. on another stream, beside the table gradient, instead of ahead of it */
#define US_MLP_DEFER_REDUCE 2
/* US_MLP_IN_SPLIT_BF16 (US_PREC_BF16 decoders, with US_MLP_LEVEL_MAJOR): `in` holds the planes us_hashgrid_fwd_joint wrote with
 * US_GRID_FEAT_SPLIT_BF16 (the inputs already as hi / lo bf16 pairs).  Outputs and gradients as with float planes, bit for bit. */
#define US_MLP_IN_SPLIT_BF16 4
/* US_MLP_OUT_PREACT (us_mlp_fwd / us_mlp_fwd_pair, bf16-family precisions): `out` receives the outputs BEFORE out_act; the consumer applies it
 * (us_render_loss_fwd + US_RENDER_ACT).  US_MLP_DOUT_PREACT (us_mlp_bwd / _pair / _pair_dydx, bf16-family): dL_dout is the gradient w.r.t.
 * those pre-activation outputs (us_render_loss_bwd + US_RENDER_ACT); `out` is not read (may be NULL). */
#define US_MLP_OUT_PREACT 8
#define US_MLP_DOUT_PREACT 16
int us_mlp_reduce(const us_mlp_desc* d, const void* workspace, size_t workspace_bytes, int64_t n, float* grad_params, void* stream);

/* out[i*out_stride + o] = act(MLP(in[i][:]))[o], o < n_out   (out_stride lets two decoders write one raw[N][4]) */
int us_mlp_fwd(const us_mlp_desc* d, const float* params, const float* in, int64_t n,
               float* out, int64_t out_stride, int flags, void* stream);

/* dL_din[N][n_in] (nullable) and grad_params += (nullable), from dL_dout[i*dout_stride + o].
 * `out` is the forward result (same layout as in us_mlp_fwd) used for the activation derivative.
 * workspace (nullable, us_mlp_bwd_workspace_bytes(d) bytes): per-workgroup partial weight gradients are stored there and
 * summed in a fixed order by a second small kernel (reproducible, no contended atomics); without it every workgroup adds
 * its partial vector with one float atomic per parameter. */
size_t us_mlp_bwd_workspace_bytes(const us_mlp_desc* d);
int us_mlp_bwd(const us_mlp_desc* d, const float* params, const float* in, const float* out, int64_t out_stride,
               const float* dL_dout, int64_t dout_stride, int64_t n, float* dL_din, float* grad_params, int flags,
               void* workspace, size_t workspace_bytes, void* stream);

/* Two decoders of equal shape (32 inputs, the same width, depth and bf16 precision: us_mlp_pair_supported) in ONE launch each way --
 * the sdf and the colour decoder of Decoders.forward.  Arguments and results as two us_mlp_fwd / us_mlp_bwd calls (bit-identical);
 * the backward pass forms the parameter gradients of both decoders (then with one workspace, us_mlp_bwd_workspace_bytes, per decoder) or of
 * neither (grad_params and workspaces NULL: the input gradients only, as tracking needs them). */
int us_mlp_pair_supported(const us_mlp_desc* a, const us_mlp_desc* b);
/* us_mlp_bwd_pair(..., US_MLP_DEFER_REDUCE, ...) leaves the partial weight gradients of both decoders in their workspaces;
 * us_mlp_reduce_pair adds them to the gradients later (one launch, fixed order).  (The pair launch lays out its partial rows its own
 * way: its workspaces are reduced by this function, not by us_mlp_reduce.) */
int us_mlp_reduce_pair(const us_mlp_desc* da, const us_mlp_desc* db, const void* workspace_a, const void* workspace_b, size_t workspace_bytes,
                       int64_t n, float* grad_params_a, float* grad_params_b, void* stream);
/* us_mlp_bwd_pair that also contracts dL/d(features) with the encoder's dy/dx while it is in registers: dL_dpts_a / dL_dpts_b [N][3]
 * receive each decoder's share of dL/d(point) (dy_dx_*: the planes [L][3][N][2] of us_hashgrid_fwd_joint_dydx for decoder a's / b's
 * grid; level-major inputs).  us_ray_points_bwd2 adds the two shares and reduces them to the rays.  The input gradient of the pose
 * optimisations (src/Mapper.py:372-376,444; src/Tracker.py:170-174,241) without a second pass over dL/d(features); dL_din_* may be
 * NULL when no table gradient follows (tracking). */
int us_mlp_bwd_pair_dydx(const us_mlp_desc* da, const us_mlp_desc* db, const float* params_a, const float* params_b, const float* in_a,
                         const float* in_b, const float* out_a, int64_t out_stride_a, const float* out_b, int64_t out_stride_b,
                         const float* dL_dout_a, int64_t dout_stride_a, const float* dL_dout_b, int64_t dout_stride_b, int64_t n,
                         float* dL_din_a, float* dL_din_b, float* grad_params_a, float* grad_params_b, int flags, void* workspace_a,
                         void* workspace_b, size_t workspace_bytes, const us_half_t* dy_dx_a, const us_half_t* dy_dx_b, float* dL_dpts_a,
                         float* dL_dpts_b, void* stream);
/* us_mlp_reduce_pair with torch.optim.Adam of the decoder param group folded in (src/Mapper.py:118,443-445; single process): the
 * gradients are the fixed-order sums of the partial rows us_mlp_bwd_pair(US_MLP_DEFER_REDUCE) left (WRITTEN to grad_params_*, no cleared
 * buffer needed), beta's gradient the f64 sum of the per-ray partials of us_render_loss_bwd(US_LOSS_DEFER_BETA) (beta_partials NULL: no
 * beta), then Adam on exactly those parameters with the bias corrections us_adam_step_inc left in step_dev -- the arithmetic of
 * us_adam_step_segments_dev.  One launch on the main stream instead of a fill, two reductions and a fork / join around the table gradient. */
int us_mlp_reduce_pair_adam(const us_mlp_desc* da, const us_mlp_desc* db, const void* workspace_a, const void* workspace_b,
                            size_t workspace_bytes, int64_t n, float* params_a, float* params_b, float* grad_params_a, float* grad_params_b,
                            float* m_a, float* m_b, float* v_a, float* v_b, const float* beta_partials, int64_t n_rays, float* beta,
                            float* grad_beta, float* m_beta, float* v_beta, double lr, double beta1, double beta2, double eps,
                            const float* step_dev, void* stream);
/* A joint_opt window's pose group for us_adam_step_model: the arguments of us_pose_window_step (row_a .. n_b: the rows of the optimised
 * frames) or, shape_dev != NULL, of us_arena_pose_step (the window's shape on the device; n_poses = the arena's pose capacity). */
typedef struct us_pose_step_desc {
    float* poses7; int n_poses;
    const float *g_rays_o, *g_rays_d, *dirs;
    int64_t row_a, n_a; int first_pose_b; int64_t row_b, n_b;
    float *m7, *v7, *g7_out;
    double lr_q, lr_t;
    const int32_t* shape_dev; int64_t rows_a;
} us_pose_step_desc;
/* us_mlp_reduce_pair_adam and us_adam_step_segments_dev in ONE launch: the whole optimiser step of a model (src/Mapper.py:111-139,443-445: one
 * torch.optim.Adam over the decoder group and the two tables) -- the decoder group's reductions + Adam are the first workgroups of the tables'
 * launch instead of a launch of their own in front of it; with poses != NULL the camera poses' group of a joint_opt window (Mapper.py:359-364)
 * as well: the work of us_pose_window_step / us_arena_pose_step, one workgroup per frame, ahead of both.  Arguments: those of
 * us_mlp_reduce_pair_adam (lr_decoders: its lr), then those of us_adam_step_segments_dev for the table segments.  Same arithmetic, same bits
 * as the separate calls. */
int us_adam_step_model(const us_mlp_desc* da, const us_mlp_desc* db, const void* workspace_a, const void* workspace_b, size_t workspace_bytes,
                       int64_t n, float* params_a, float* params_b, float* grad_params_a, float* grad_params_b, float* m_a, float* m_b,
                       float* v_a, float* v_b, const float* beta_partials, int64_t n_rays, float* beta, float* grad_beta, float* m_beta,
                       float* v_beta, double lr_decoders, float* p, float* g, float* m, float* v, int n_seg, const int64_t* seg_off,
                       const int64_t* seg_n, const double* seg_lr, double beta1, double beta2, double eps, float* step_dev,
                       unsigned zero_grad_mask, const us_pose_step_desc* poses, void* stream);
int us_mlp_fwd_pair(const us_mlp_desc* da, const us_mlp_desc* db, const float* params_a, const float* params_b, const float* in_a,
                    const float* in_b, int64_t n, float* out_a, int64_t out_stride_a, float* out_b, int64_t out_stride_b, int flags, void* stream);
int us_mlp_bwd_pair(const us_mlp_desc* da, const us_mlp_desc* db, const float* params_a, const float* params_b, const float* in_a,
                    const float* in_b, const float* out_a, int64_t out_stride_a, const float* out_b, int64_t out_stride_b,
                    const float* dL_dout_a, int64_t dout_stride_a, const float* dL_dout_b, int64_t dout_stride_b, int64_t n,
                    float* dL_din_a, float* dL_din_b, float* grad_params_a, float* grad_params_b, int flags, void* workspace_a,
                    void* workspace_b, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ray sampling / points  (replaces src/utils/Renderer.py:81-101,132-137 and src/common.py:152-166 gather+rotate)
 * ---------------------------------------------------------------------------------------------- */
/* depth-guided z sampling for rays with gt_depth > 0 (Renderer.py:86-101):
 *   z_free = (c_free*gt)*t_uni[j];  z_surf = (gt - surf_off) + surf_span*t_surf[k];  merge-sort;  optional jitter
 *   (Renderer.py:42-57) with t_rand[R][S] (NULL: no perturbation).  S = n_strat + n_imp <= 256.          */
int us_sample_z(const float* gt_depth, int64_t n_rays, const float* t_uni, int n_strat, const float* t_surf, int n_imp,
                float c_free, float surf_off, float surf_span, const float* t_rand, float* z_vals, void* stream);
/* c_free = 1.2f, surf_off = (float)(1.5*truncation), surf_span = (float)(3*truncation): the python scalars the
 * reference multiplies into fp32 tensors */

/* pts[R][S][3] = ((o + d*z) - bound_lo) / (bound_hi - bound_lo)   (Renderer.py:132-137); bound_host = {lo[3], hi[3]} */
int us_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, const float* bound_host,
                  int64_t n_rays, int n_samples, float* pts, void* stream);
/* adjoint of us_ray_points wrt rays_o / rays_d (tracking: pose gradient, Tracker.py:170-174) */
int us_ray_points_bwd(const float* dL_dpts, const float* z_vals, const float* bound_host, int64_t n_rays,
                      int n_samples, float* dL_do, float* dL_dd, void* stream);

/* the same for dL_dpts = dL_dpts_a + dL_dpts_b (the two grids' shares us_mlp_bwd_pair_dydx leaves) */
int us_ray_points_bwd2(const float* dL_dpts_a, const float* dL_dpts_b, const float* z_vals, const float* bound_host, int64_t n_rays,
                       int n_samples, float* dL_do, float* dL_dd, void* stream);

/* importance samples of the rays WITHOUT a depth measurement (src/utils/Renderer.py:121-130 + common.sample_pdf src/common.py:49-85):
 * sdf_uni[R][n_uniform] of the coarse uniform pass at z_uni[R][n_uniform] -> alpha (beta: device float[1]) -> weights -> the reference's
 * un-normalised cdf over weights[1:-1] -> inverse transform with the draws u[R][n_importance] -> z_out[R][n_uniform + n_importance],
 * sorted.  n_uniform <= 128, n_importance <= 64. */
int us_importance_z(const float* sdf_uni, const float* z_uni, const float* beta, const float* u, int64_t n_rays, int n_uniform,
                    int n_importance, float* z_out, void* stream);

/* The same with the pieces a sync-light mapping iteration needs around it (src/utils/Renderer.py:104-137 for the rays with gt_depth == 0):
 *   us_zero_depth_rows   rows[k] = indices of the rays with !(gt_depth > 0), ascending, and count[0] (device int32; ONE workgroup)
 *   us_uniform_points    the coarse pass of those rows (rows NULL: all rays): far = far_bb + 0.01 (Renderer.py:108-111), z_uni[k][j] =
 *                        far*t_uni[j] with the jitter of :42-57 (t_rand[n_rows][n_uniform] or, NULL, the in-kernel generator), and
 *                        pts[k][j][3] normalised to [-1,1] (common.normalize_3d_coordinate, what Renderer.py:113 feeds the encoder)
 *   us_importance_z_rows us_importance_z on compacted inputs, written to row rows[k] of z_out[R][S] (rows NULL: row k); u NULL: draws
 *                        from the in-kernel generator; pts_out (nullable, with rays_o / rays_d / bound_host): the unit-cube points
 *                        of the written rows, as us_ray_points would produce them. */
int us_zero_depth_rows(const float* gt_depth, int64_t n_rays, int32_t* rows, int32_t* count, void* stream);
int us_uniform_points(const float* rays_o, const float* rays_d, const int32_t* rows, int64_t n_rows, const float* bound_host,
                      const float* t_uni, int n_uniform, const float* t_rand, uint64_t rng_seed, int perturb, float* z_uni,
                      float* pts, void* stream);
int us_importance_z_rows(const float* sdf_uni, const float* z_uni, const float* beta, const float* u, uint64_t rng_seed,
                         int64_t n_rows, int n_uniform, int n_importance, const int32_t* rows, float* z_out,
                         const float* rays_o, const float* rays_d, const float* bound_host, float* pts_out, void* stream);
/* The whole branch of src/utils/Renderer.py:104-130 with NO row count on the host (it can be captured into a hipGraph): us_zero_depth_rows,
 * then us_uniform_points -> us_hashgrid_fwd (grid / table: the sdf grid) -> us_mlp_fwd (mlp / mlp_params: the sdf decoder, n_out 1) ->
 * us_importance_z_rows, each launched for all n_rays rows and reading count[0] on the device; workgroups beyond it leave at once.  The
 * rows of z_out[R][S] / pts_out[R][S][3] that belong to rays with !(gt_depth > 0) are rewritten, the others are left alone.  Scratch
 * (caller-owned, contents undefined afterwards): rows[n_rays] int32, count[1] int32, z_uni / sdf_uni [n_rays * n_uniform],
 * pts_uni [n_rays * n_uniform * 3], feat [n_rays * n_uniform * n_levels * n_features].  t_rand[k][n_uniform] / u[k][n_importance]
 * (nullable: in-kernel generator with seed_uniform / seed_importance, rng_counter mixed in on the device as in us_sample_points) are
 * indexed by the COMPACTED row k. */
int us_zero_depth_resample(const us_grid_desc* grid, const float* table, const us_mlp_desc* mlp, const float* mlp_params,
                           const float* beta, const float* rays_o, const float* rays_d, const float* gt_depth, int64_t n_rays,
                           const float* bound_host, const float* t_uni, int n_uniform, int n_importance, const float* t_rand,
                           const float* u, uint64_t seed_uniform, uint64_t seed_importance, const float* rng_counter, int perturb,
                           int32_t* rows, int32_t* count, float* z_uni, float* pts_uni, float* feat, float* sdf_uni, float* z_out, float* pts_out,
                           void* stream);

/* us_bbox_filter + us_sample_z + us_ray_points in ONE launch, value for value (the iteration of src/Mapper.py:396-406 +
 * src/utils/Renderer.py:81-101,132-137 when no ray takes the zero-depth branch).  perturb != 0: jitter with t_rand[R][S], or,
 * when t_rand is NULL, with an in-kernel counter-based uniform generator seeded by rng_seed (the reference draws
 * torch.rand there, Renderer.py:54; any iid U[0,1) stream serves); rng_counter (device float[1], may be NULL) is mixed into
 * the seed on the device, so that replays of a captured hipGraph draw fresh numbers.  valid may be NULL. */
int us_sample_points(const float* rays_o, const float* rays_d, const float* gt_depth, const float* bound_host,
                     int64_t n_rays, const float* t_uni, int n_strat, const float* t_surf, int n_imp, float c_free,
                     float surf_off, float surf_span, const float* t_rand, uint64_t rng_seed, const float* rng_counter,
                     int perturb, int require_depth, uint8_t* valid, float* z_vals, float* pts, void* stream);

/* bounding-box pre-filter (Mapper.py:396-402, Tracker.py:177-184): far = min_dim max((lo-o)/d, (hi-o)/d);
 * valid[i] = far >= gt_depth[i] (&& gt_depth[i] > 0 when require_depth); far_out[R] optional (Renderer.py:108-113) */
int us_bbox_filter(const float* rays_o, const float* rays_d, const float* gt_depth, const float* bound_host,
                   int64_t n_rays, int require_depth, uint8_t* valid, float* far_out, void* stream);

/* mapping ray assembly: gather-then-rotate (common.py:152-166 rotates all P pool pixels first; same result):
 * idx[b][n] int64 into per-frame pools depths[b][P], colors[b][P][3], dirs[b][P][3]; c2ws[b][4][4]            */
int us_gather_rays(const float* c2ws, const float* pool_depth, const float* pool_color, const float* pool_dirs,
                   const int64_t* idx, int b, int64_t pool_size, int64_t n_per_frame,
                   float* rays_o, float* rays_d, float* depth, float* color, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SDF -> alpha compositing  (replaces src/utils/Renderer.py:140-158)
 * ---------------------------------------------------------------------------------------------- */
/* raw[R][S][4] = (r,g,b,sdf); beta: device scalar.  Outputs per ray; weights[R][S] optional (NULL). */
int us_composite_fwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples,
                     float* term, float* pixel_unc, float* depth, float* rgb, float* depth_unc, float* weights,
                     void* stream);
/* upstream per-ray grads (any may be NULL = zero) + direct grad on the returned sdf[R][S] (nullable);
 * d_raw[R][S][4] written, d_beta[1] accumulated (+=).  beta_partials (nullable, R floats of scratch): per-ray beta
 * gradients are stored there and summed in a fixed order instead of R float atomics on one address. */
int us_composite_bwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples,
                     const float* g_term, const float* g_unc, const float* g_depth, const float* g_rgb,
                     const float* g_dunc, const float* g_sdf, float* d_raw, float* d_beta, float* beta_partials,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * Uncertainty-gated losses  (replaces src/Mapper.py:141-175,411-440 and src/Tracker.py:113-147,206-238)
 * Two phases so that data-dependent mask COUNTS can be all-reduced across ranks before gradients are scaled.
 * ---------------------------------------------------------------------------------------------- */
enum { US_LOSS_MAP_ORIGINAL = 0, US_LOSS_MAP_NOMASK = 1, US_LOSS_TRK_ORIGINAL = 2, US_LOSS_TRK_NOMASK = 3 };
/* or-ed into `mode` of us_render_loss_bwd: the per-ray d(beta) partials are left in beta_partials and summed by the caller's
 * us_beta_reduce(beta_partials, n_rays, d_beta, stream) (d_beta += fixed-order sum), off the backward pass's critical path */
#define US_LOSS_DEFER_BETA 256
/* US_RENDER_ACT(act_rgb, act_sdf), or-ed into `mode` of us_render_loss_fwd AND us_render_loss_bwd (act_* = the decoders' us_mlp_desc.out_act):
 * `raw` arrives as the decoders' PRE-activation outputs (us_mlp_fwd / _pair with US_MLP_OUT_PREACT).  us_render_loss_fwd applies the
 * activations itself and REWRITES raw in place with the activated samples (what the decoders would have written, value for value: the
 * backward calls and any later reader see the usual raw); us_render_loss_bwd leaves d_raw as the gradient w.r.t. the pre-activation
 * outputs (the decoders' backward launch then takes US_MLP_DOUT_PREACT and does not read `out`).  The decoder launches are bound by VALU
 * issue, these two wait on their loads: the ~80 VALU instructions per 32 points move where they are free. */
#define US_RENDER_ACT_ON 0x100000
#define US_RENDER_ACT(act_rgb, act_sdf) (US_RENDER_ACT_ON | ((act_rgb) << 12) | ((act_sdf) << 16))
int us_beta_reduce(const float* beta_partials, int64_t n_rays, float* d_beta, void* stream);
/* us_adam_step_segments(_dev): bit 31 of zero_grad_mask = the device-side step count was already advanced for this step by
 * us_adam_step_inc(step_dev, beta1, beta2, stream) (one thread: count + 1 and the two bias corrections) */
#define US_ADAM_STEP_ADVANCED 0x80000000u
int us_adam_step_inc(float* step_dev, double beta1, double beta2, void* stream);
enum { US_LS_FS = 0, US_LS_CENTER = 1, US_LS_TAIL = 2, US_LS_COLOR = 3, US_LS_DEPTH = 4, US_LS_N = 5 };
/* stats[10]: sums[5] then counts[5] (fp32; overwritten).  sdf[(r*S+s)*sdf_stride] (stride 4 reads raw[R][S][4]'s
 * 4th channel in place); valid[R] (nullable) drops rays the bbox pre-filter rejected without compacting the
 * batch (no host sync); median: device scalar (10x-median depth-error gate, tracking only; NULL for mapping).
 * partials: workspace of us_loss_partials_size(n_rays) floats for a fixed-order (deterministic) reduction.   */
size_t us_loss_partials_size(int64_t n_rays);
int us_loss_stats(int mode, const float* sdf, int64_t sdf_stride, const uint8_t* valid, const float* z_vals,
                  const float* gt_depth, const float* gt_color,
                  const float* depth, const float* rgb, const float* pixel_unc, const float* median,
                  int64_t n_rays, int n_samples, double truncation, float* partials, float* stats, void* stream);
/* gradients of  loss = sum_k w[k] * sums[k] / counts[k]  wrt sdf[R][S], depth[R], rgb[R][3], using the (possibly
 * globally reduced) counts in stats[5..9];  loss_out[1] (nullable) receives the scalar from stats.            */
int us_loss_grad(int mode, const float* sdf, int64_t sdf_stride, const uint8_t* valid, const float* z_vals,
                 const float* gt_depth, const float* gt_color,
                 const float* depth, const float* rgb, const float* pixel_unc, const float* median,
                 int64_t n_rays, int n_samples, double truncation, const float* w_host5, const float* stats,
                 float* g_sdf, float* g_depth, float* g_rgb, float* loss_out, void* stream);

/* us_composite_fwd + us_loss_stats in one launch + the reduction, and us_loss_grad + us_composite_bwd in one launch (+ the beta
 * reduction): the mapping iteration's per-ray chain src/utils/Renderer.py:140-158 -> src/Mapper.py:411-440 and its adjoint.  Modes
 * without the median gate only (US_LOSS_MAP_ORIGINAL, US_LOSS_MAP_NOMASK, US_LOSS_TRK_NOMASK); same values as the separate calls
 * (bit for bit for n_samples <= 64). */
int us_render_loss_fwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples, int mode,
                       const uint8_t* valid, const float* gt_depth, const float* gt_color, double truncation, float* termination,
                       float* pixel_unc, float* depth, float* rgb, float* depth_unc, float* partials, float* stats, void* stream);
int us_render_loss_bwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples, int mode,
                       const uint8_t* valid, const float* gt_depth, const float* gt_color, const float* depth, const float* rgb,
                       const float* pixel_unc, double truncation, const float* w_host5, const float* stats, float* d_raw,
                       float* d_beta, float* beta_partials, float* loss_out, void* stream);

/* the tracker's optimiser step (src/Tracker.py:322-329,242) in one launch: Adam on pose7 = (quaternion[4] with lr_R,
 * translation[3] with lr_T), us_adam_step_dev arithmetic; step_dev[0] (float) is incremented by the kernel before use */
int us_pose_adam_step(float* pose7, const float* g7, float* m7, float* v7, double lr_R, double lr_T, double beta1, double beta2,
                      double eps, float* step_dev, void* stream);

/* out[0] = lower median (torch.median) of |a[i] - b[i]| over the elements with valid[i] != 0 (valid NULL: all); +inf if none.
 * n <= 8192.  The 10 x median gate of the tracking loss: src/Tracker.py:212-214. */
int us_masked_median(const float* a, const float* b, const uint8_t* valid, int64_t n, float* out, void* stream);
/* out[0] = sum of a[i] over the elements with valid[i] != 0 (valid NULL: all) / max(their count, 1); fixed-order f64 sums, one launch.
 * The tracker's mean rendered uncertainty of the pre-filtered rays, `rendered_weights.detach().mean()` src/Tracker.py:353. */
int us_masked_mean(const float* a, const uint8_t* valid, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser  (replaces torch.optim.Adam at src/Mapper.py:364,445 / src/Tracker.py:328,242)
 * ---------------------------------------------------------------------------------------------- */
/* torch.optim.Adam semantics (amsgrad off, weight_decay 0); step = 1-based step count after increment */
int us_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                 double eps, int step, void* stream);

/* torch.optim.Adam's optimizer.step() over up to 40 SEPARATE tensors in ONE launch (each with its own parameter / gradient / moment
 * allocations, length n[t] and learning rate lr[t]; the pointer and length arrays are HOST arrays of n_tensors entries): the param_groups of
 * src/Mapper.py:118-126 (the decoders' nn.Linear parameters, beta, the two tables) as src/Mapper.py:445 steps them -- what
 * unislam_amd.optim.Adam calls.  us_adam_step's arithmetic per element; step is the 1-based count shared by all tensors. */
int us_adam_step_tensors(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                         const double* lr, double beta1, double beta2, double eps, int step, void* stream);

/* the same over n_seg (<= 8) segments [seg_off[k], seg_off[k] + seg_n[k]) of one flat buffer in ONE launch, each segment with
 * its own learning rate -- the optimizer's param_groups (src/Mapper.py:118-126: decoders, sdf tables, colour tables);
 * seg_off / seg_n / seg_lr are HOST arrays.  Bit k of zero_grad_mask: clear segment k of g once it has been consumed
 * (the optimizer.zero_grad() of src/Mapper.py:443 for gradients that the next backward ADDS to). */
int us_adam_step_segments(float* p, float* g, float* m, float* v, int n_seg, const int64_t* seg_off,
                          const int64_t* seg_n, const double* seg_lr, double beta1, double beta2, double eps, int step,
                          unsigned zero_grad_mask, void* stream);
/* The same with the step count on the device: step_dev is float[8], 8-byte aligned, zeroed for a fresh optimiser; step_dev[0] holds the
 * count, which this call advances by one and then uses for the bias corrections (kept in the remaining words).  No argument changes from step to step, so the launch can sit in a captured hipGraph (MapStep.capture). */
int us_adam_step_segments_dev(float* p, float* g, float* m, float* v, int n_seg, const int64_t* seg_off, const int64_t* seg_n,
                              const double* seg_lr, double beta1, double beta2, double eps, float* step_dev,
                              unsigned zero_grad_mask, void* stream);
/* us_adam_step_segments_dev where the segments flagged in bf16_mask (bit k = segment k) read their gradient from g_bf16 -- a bfloat16
 * image of the gradient buffer, same indexing: the payload of the data-parallel all-reduce as it came off the wire (no widening pass) */
int us_adam_step_segments_bf16(float* p, float* g, const uint16_t* g_bf16, unsigned bf16_mask, float* m, float* v, int n_seg,
                               const int64_t* seg_off, const int64_t* seg_n, const double* seg_lr, double beta1, double beta2,
                               double eps, float* step_dev, unsigned zero_grad_mask, void* stream);

/* the same with the 1-based step count in device memory (float[1]): nothing step-dependent is baked into the launch, so the
 * call can sit inside a captured hipGraph (torch.optim.Adam(capturable=True) arithmetic: bias corrections in fp32) */
int us_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                     double eps, const float* step_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Tracking: camera pose -> rays  (replaces cam_pose_to_matrix src/common.py:196-208 + pytorch3d quaternion_to_matrix,
 * get_sample_uv / select_uv / get_rays_from_uv src/common.py:95-150 as called by Tracker.optimize_tracking
 * src/Tracker.py:170-174) and the adjoint that autograd provides there (src/Tracker.py:241)
 * ---------------------------------------------------------------------------------------------- */
/* pose[7] = (qr,qi,qj,qk, tx,ty,tz) device; pix[n] int64 = flat indices into the crop [H0:H1, W0:W1] of width crop_w;
 * depth_img[H][W], color_img[H][W][3]; intr_host4 = {fx, fy, cx, cy}.  Writes rays_o/rays_d/dirs [n][3] (dirs = camera-frame
 * directions, kept for the adjoint), gt_depth[n], gt_color[n][3]. */
int us_pose_rays(const float* pose, const int64_t* pix, int64_t n, const float* intr_host4, int W0, int H0, int crop_w,
                 const float* depth_img, const float* color_img, int W, float* rays_o, float* rays_d, float* dirs,
                 float* gt_depth, float* gt_color, void* stream);
/* us_pose_rays + us_sample_points (require_depth) in ONE launch: pose -> pixel rays -> pre-filter flag, sorted + jittered z, unit-cube
 * points (src/Tracker.py:170-184 + src/utils/Renderer.py:81-101,132-137).  pix NULL: the n pixels are drawn in the kernel (counter-based
 * uniform draw over the crop_w x crop_h crop, seeded like the jitter: rng_seed mixed with rng_counter), where the reference calls
 * torch.randint (src/common.py:116).  rays_o / rays_d nullable (nothing downstream of the fused tracking iteration reads them). */
int us_track_sample(const float* pose, const int64_t* pix, int64_t n_rays, const float* intr_host4, int W0, int H0, int crop_w, int crop_h,
                    const float* depth_img, const float* color_img, int W, const float* bound_host, const float* t_uni, int n_strat,
                    const float* t_surf, int n_imp, float c_free, float surf_off, float surf_span, const float* t_rand, uint64_t rng_seed,
                    const float* rng_counter, int perturb, float* rays_o, float* rays_d, float* dirs, float* gt_depth, float* gt_color,
                    uint8_t* valid, float* z_vals, float* pts, void* stream);
/* The tracking loss with the 10 x median gate (US_LOSS_TRK_ORIGINAL, src/Tracker.py:206-238) fused around the compositing:
 *   us_track_loss_fwd  us_composite_fwd that also leaves |gt - depth| per ray (err[R]) and every ray's ten loss partials under the
 *                      median-free half of the gate, then ONE workgroup: lower median over the pre-filtered rays -> median[1], sums of
 *                      the partials of the rays with err < 10 median -> stats[10]   (= us_composite_fwd + us_masked_median +
 *                      us_loss_stats; n_rays <= 8192)
 *   us_track_loss_bwd  us_loss_grad + us_composite_bwd in one launch, given that median and those statistics. */
int us_track_loss_fwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples, const uint8_t* valid,
                      const float* gt_depth, const float* gt_color, double truncation, float* termination, float* pixel_unc, float* depth,
                      float* rgb, float* depth_unc, float* partials, float* err, float* median, float* stats, void* stream);
int us_track_loss_bwd(const float* raw, const float* z_vals, const float* beta, int64_t n_rays, int n_samples, const uint8_t* valid,
                      const float* gt_depth, const float* gt_color, const float* depth, const float* rgb, const float* pixel_unc,
                      const float* median, double truncation, const float* w_host5, const float* stats, float* d_raw, float* loss_out,
                      void* stream);
/* g_pose[7] = dL/dpose from dL/d rays_o, dL/d rays_d (closed-form chain rule through R(q) = I + 2 M(q)/|q|^2) */
int us_pose_grad(const float* pose, const float* g_rays_o, const float* g_rays_d, const float* dirs, int64_t n, float* g_pose,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * Mapping window with joint pose optimisation  (replaces, per iteration of Mapper.optimize_mapping with joint_opt -- the default,
 * configs/UNISLAM.yaml:50, on from the fifth keyframe src/Mapper.py:519 -- cam_pose_to_matrix + get_samples_all src/Mapper.py:372-393,
 * src/common.py:152-166,196-208, and the pose param group of the optimiser src/Mapper.py:359-364,443-445)
 * ---------------------------------------------------------------------------------------------- */
/* Rays of window frames f_begin .. f_begin + f_count - 1, n_per_frame pixels each: idx[f_count][n_per_frame] int64 into the frames'
 * pools (pool_*[b][P]..., indexed by the WINDOW frame number).  Frame 0 keeps the matrix c2w_first[4][4] (the oldest pose is fixed,
 * src/Mapper.py:374); frame f >= 1 uses poses7[f-1] = (qr,qi,qj,qk, tx,ty,tz) through pytorch3d's quaternion_to_matrix.  Outputs as
 * us_gather_rays plus dirs[n][3] (nullable), the camera-frame directions us_pose_window_step needs.  The output pointers address
 * the first row of this call's block (a second call appends the extra rays of the newest frames, src/Mapper.py:385-393).
 * c2w_first NULL (us_window_rays and us_window_sample): none of these frames is the window's fixed one -- frame f uses poses7[f]
 * (a data-parallel rank whose share of the window's frames does not hold the oldest). */
int us_window_rays(const float* c2w_first, const float* poses7, const float* pool_depth, const float* pool_color, const float* pool_dirs,
                   const int64_t* idx, int64_t pool_size, int f_begin, int f_count, int64_t n_per_frame, float* rays_o, float* rays_d,
                   float* depth, float* color, float* dirs, void* stream);
/* us_window_rays (both blocks) + us_sample_points in ONE launch: window poses -> pool pixels -> rays -> pre-filter flag, sorted + jittered
 * z, unit-cube points (src/Mapper.py:372-406 + src/utils/Renderer.py:81-101,132-137 when no ray takes the zero-depth branch).  Rows
 * [0, b * n_per_frame) hold frame row / n_per_frame; behind them n_extra pixels from each of the newest n_extra_frames frames
 * (src/Mapper.py:385-393; 0: none).  idx_a[b][n_per_frame] / idx_b[n_extra_frames][n_extra]: pool pixels; both NULL: drawn in the
 * kernel (counter-based uniform draw over the pool, seeded like the jitter: rng_seed mixed with rng_counter), where the reference
 * calls torch.randint (src/common.py:155).  Outputs as us_window_rays + us_sample_points (dirs nullable). */
int us_window_sample(const float* c2w_first, const float* poses7, int b, int64_t n_per_frame, int n_extra_frames, int64_t n_extra,
                     const float* pool_depth, const float* pool_color, const float* pool_dirs, int64_t pool_size, const int64_t* idx_a,
                     const int64_t* idx_b, const float* bound_host, const float* t_uni, int n_strat, const float* t_surf, int n_imp,
                     float c_free, float surf_off, float surf_span, const float* t_rand, uint64_t rng_seed, const float* rng_counter,
                     int perturb, float* rays_o, float* rays_d, float* dirs, float* gt_depth, float* gt_color, uint8_t* valid,
                     float* z_vals, float* pts, void* stream);
/* Pose gradient and Adam step of n_poses poses in one launch (one workgroup per pose).  Pose j owns rows
 * [row_a + j*n_a, row_a + (j+1)*n_a) of g_rays_o / g_rays_d / dirs and, if n_b > 0 and j >= first_pose_b, rows
 * [row_b + (j-first_pose_b)*n_b, ... + n_b).  g7 = dL/dpose by the closed-form chain rule through R(q) = I + 2 M(q)/|q|^2 (fixed-order
 * f64 sums); then torch.optim.Adam on the 7 numbers (lr_q for the quaternion, lr_t for the translation).  g7_out (nullable) receives
 * the gradients.  flags:
 *   0                  step_dev = the float[8] of us_adam_step_inc, ALREADY advanced for this optimiser step (the poses are one more
 *                      param group of the mapping optimiser: same count, same bias corrections as us_adam_step_segments_dev)
 *   US_POSE_OWN_STEP   step_dev = float[1], advanced by this launch, fp32 bias corrections (us_pose_adam_step arithmetic; the
 *                      tracker's per-frame optimiser src/Tracker.py:322-329,242); n_poses must be 1
 *   US_POSE_GRAD_ONLY  no optimiser step (m7, v7, step_dev may be NULL) */
#define US_POSE_GRAD_ONLY 1
#define US_POSE_OWN_STEP 2
int us_pose_window_step(float* poses7, int n_poses, const float* g_rays_o, const float* g_rays_d, const float* dirs, int64_t row_a,
                        int64_t n_a, int first_pose_b, int64_t row_b, int64_t n_b, float* m7, float* v7, float* g7_out, double lr_q,
                        double lr_t, double beta1, double beta2, double eps, float* step_dev, int flags, void* stream);
/* The same two launches for a window whose SHAPE lives on the device, over a persistent arena of keyframe pools -- so that ONE captured
 * hipGraph serves every window of a run, whatever its number of frames (src/Mapper.py:276-364: the window grows with the keyframe list).
 *   shape_dev int32[8] = { frames b, pixels per frame, extra frames, extra pixels, 1 if frame 0 is the fixed one, -, -, - };
 *   slots int32[b]: window frame -> row of pool_*[K][P]...;  poses7 [cap][7] (frame f uses row f - fixed).
 * Fixed row layout: rows [0, rows_a) carry the b * n_per rays of the first block, rows [rows_a, rows_a + rows_b) the extra block; rows
 * beyond a block's real rays are padding -- sampled like any ray, flagged invalid (dropped by the loss like a ray the pre-filter
 * rejects).  idx_a [rows_a] / idx_b [rows_b] or NULL (in-kernel draw; rng_counter = the optimiser's float[8] step_dev, whose [1] is
 * mixed in as the window's epoch).  us_arena_pose_step launches n_poses_cap workgroups; those beyond b - fixed leave. */
int us_arena_window_sample(const float* c2w_first, const float* poses7, const int32_t* shape_dev, const int32_t* slots, int64_t rows_a,
                           int64_t rows_b, const float* pool_depth, const float* pool_color, const float* pool_dirs, int64_t pool_size,
                           const int64_t* idx_a, const int64_t* idx_b, const float* bound_host, const float* t_uni, int n_strat,
                           const float* t_surf, int n_imp, float c_free, float surf_off, float surf_span, const float* t_rand,
                           uint64_t rng_seed, const float* rng_counter, int perturb, float* rays_o, float* rays_d, float* dirs,
                           float* gt_depth, float* gt_color, uint8_t* valid, float* z_vals, float* pts, void* stream);
int us_arena_pose_step(float* poses7, int n_poses_cap, const int32_t* shape_dev, int64_t rows_a, const float* g_rays_o,
                       const float* g_rays_d, const float* dirs, float* m7, float* v7, float* g7_out, double lr_q, double lr_t,
                       double beta1, double beta2, double eps, float* step_dev, void* stream);
/* A keyframe's pixel pool (src/Mapper.py:329-337,516-523: torch.randperm(n_pixels)[:pool_size] + three gathers) in one launch: pool row i =
 * pixel pi(i) of the frame (color [n_pixels][3], depth [n_pixels], dirs [n_pixels][3]), pi a pseudo-random bijection of [0, n_pixels)
 * keyed by `seed` -- pool_size distinct pixels, no sort.  has_zero (nullable, device int32[1], cleared by the caller): |= 1 if a pool
 * pixel has no depth. */
int us_pool_cut(const float* color, const float* depth, const float* dirs, int64_t n_pixels, int64_t pool_size, uint64_t seed,
                float* pool_color, float* pool_depth, float* pool_dirs, int32_t* has_zero, void* stream);
/* Mapper.keyframe_selection_LC's overlap measure (src/Mapper.py:188-240) in one launch: pix[n_pix] (flat pixel numbers y * W + x of the
 * current frame, c2w [4][4], depth [H*W]) -> for each of n_keyframes keyframes (pose pose_list[kf_frames[k]], a rigid [4][4]) the share of
 * the n_pix * n_samples ray points (pixels without a depth left out; z from 0.8 d to d + 0.5) that project inside its image with an
 * `edge`-pixel margin and in front of the camera.  intr_host4 = {fx, fy, cx, cy} (host).  percent [n_keyframes]. */
int us_keyframe_overlap(const float* c2w, const float* depth, const int64_t* pix, int n_pix, int n_samples, const float* intr_host4,
                        int H, int W, int edge, const float* pose_list, const int64_t* kf_frames, int n_keyframes, float* percent,
                        void* stream);
/* The tracker's pose step with the loop's minimum-loss bookkeeping in the same launch (src/Tracker.py:240-242 and :346-348):
 * us_pose_window_step(pose7, 1, ..., rows [0, n_rays), US_POSE_OWN_STEP), and before the step: where loss[0] -- the loss of THIS iteration,
 * rendered at pose7 as it is on entry -- is below min_loss[0], min_loss[0] takes it and best7[7] that pose (`candidate_cam_pose`).  A NaN
 * loss never counts as better (torch's `<`).  g7_out nullable.  draw_counter (float[1], nullable) is advanced by one: the rng_counter of
 * us_track_sample's in-kernel pixel draw when it has to outlive the per-frame optimiser state (step_dev restarts with every frame). */
int us_pose_track_step(float* pose7, const float* g_rays_o, const float* g_rays_d, const float* dirs, int64_t n_rays, float* m7, float* v7,
                       float* g7_out, double lr_q, double lr_t, double beta1, double beta2, double eps, float* step_dev, const float* loss,
                       float* min_loss, float* best7, float* draw_counter, void* stream);

/* Pose conversions of the drivers' per-frame glue, one launch each (src/common.py:182-208; on torch ops: chains of ~30 / ~20 small launches).
 *   us_matrix_to_cam_pose  c2w[n][4][4] row-major -> pose7[n][7] = (pytorch3d matrix_to_quaternion of the rotation, real part first; the
 *                          translation column), matrix_to_cam_pose(RT=True).  extrapolate != 0 (n must be 1, c2w holds TWO matrices):
 *                          pose7[0] = 2 * pose(c2w[1]) - pose(c2w[0]), the constant-speed prediction of src/Tracker.py:317-320.
 *   us_cam_pose_to_matrix  pose7[n][7] -> c2w[n][4][4] through pytorch3d quaternion_to_matrix, last row (0, 0, 0, 1). */
int us_matrix_to_cam_pose(const float* c2w, int n, int extrapolate, float* pose7, void* stream);
int us_cam_pose_to_matrix(const float* pose7, int n, float* c2w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNISLAM_HIP_H */
