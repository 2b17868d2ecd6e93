// window.hip -- the camera poses of a mapping window (and of the tracked frame) on the device, for gfx950.
//
// With joint_opt (the reference's default, configs/UNISLAM.yaml:50, on from the fifth keyframe: src/Mapper.py:519) every mapping
// iteration runs poses -> rotation -> rays (src/Mapper.py:372-376 -> src/common.py:152-166,196-208), and after the backward pass the
// gradient flows rays -> rotation -> quaternion / translation into one more Adam param group (src/Mapper.py:359-364,443-445).  The
// reference leaves both directions to autograd over ~20 small torch ops per iteration; here they are two launches:
//   k_window_rays        pixel gather + quaternion -> rotation + rotate, per ray (the oldest frame keeps its given matrix:
//                        "we fix the oldest c2w to avoid drifting", src/Mapper.py:374)
//   k_pose_window_step   one workgroup per optimised frame: G = sum_rays g_d (x) dir, g_t = sum_rays g_o over the frame's rays
//                        (fixed order, f64), the closed-form chain rule through R(q) = I + 2 M(q) / |q|^2, and Adam on the 7 numbers
// The tracker's single frame (src/Tracker.py:170-174,240-242) is the same step with one workgroup and its own step count.
// Compiled with -ffp-contract=off like render.hip: the ray arithmetic repeats the reference's separate fp32 ops.
#include "us_common.h"
#include <math.h>

__device__ __forceinline__ void w_quat_rot(const float* __restrict__ q, float R[9]) {          // pytorch3d quaternion_to_matrix (real part first)
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    const float s = 2.0f / (r * r + i * i + j * j + k * k);
    R[0] = 1.f - s * (j * j + k * k); R[1] = s * (i * j - k * r); R[2] = s * (i * k + j * r);
    R[3] = s * (i * j + k * r); R[4] = 1.f - s * (i * i + k * k); R[5] = s * (j * k - i * r);
    R[6] = s * (i * k - j * r); R[7] = s * (j * k + i * r); R[8] = 1.f - s * (i * i + j * j);
}

// rays of `f_count` frames starting at window frame `f_begin`, n_per pixels each: idx[f_count][n_per] into the frames' pools
__global__ __launch_bounds__(256) void k_window_rays(const float* __restrict__ c2w_first, const float* __restrict__ poses7,
                                                     const float* __restrict__ pool_depth, const float* __restrict__ pool_color,
                                                     const float* __restrict__ pool_dirs, const int64_t* __restrict__ idx, int64_t P,
                                                     int f_begin, int64_t n_per, int64_t total, float* __restrict__ rays_o,
                                                     float* __restrict__ rays_d, float* __restrict__ depth, float* __restrict__ color,
                                                     float* __restrict__ dirs) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = f_begin + i / n_per;
        const int64_t src = f * P + idx[i];
        float R[9], t[3];
        if (f == 0 && c2w_first) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { R[k * 3] = c2w_first[k * 4]; R[k * 3 + 1] = c2w_first[k * 4 + 1]; R[k * 3 + 2] = c2w_first[k * 4 + 2]; t[k] = c2w_first[k * 4 + 3]; }
        } else {                                                   // c2w_first NULL: no frame of this block is fixed, frame f reads poses7[f]
            const float* q = poses7 + (f - (c2w_first ? 1 : 0)) * 7;
            w_quat_rot(q, R);
            t[0] = q[4]; t[1] = q[5]; t[2] = q[6];
        }
        const float d0 = pool_dirs[src * 3 + 0], d1 = pool_dirs[src * 3 + 1], d2 = pool_dirs[src * 3 + 2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            rays_d[i * 3 + k] = (d0 * R[k * 3 + 0] + d1 * R[k * 3 + 1]) + d2 * R[k * 3 + 2];
            rays_o[i * 3 + k] = t[k];
            color[i * 3 + k] = pool_color[src * 3 + k];
        }
        if (dirs) { dirs[i * 3] = d0; dirs[i * 3 + 1] = d1; dirs[i * 3 + 2] = d2; }
        depth[i] = pool_depth[src];
    }
}

#include "pose_step_dev.h"

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_pose_window_step(float* __restrict__ poses7, const float* __restrict__ g_o, const float* __restrict__ g_d,
                                                          const float* __restrict__ dirs, float* __restrict__ m7, float* __restrict__ v7,
                                                          float* __restrict__ g7_out, float* __restrict__ step_dev, PoseStep ps,
                                                          const float* __restrict__ loss = nullptr, float* __restrict__ min_loss = nullptr,
                                                          float* __restrict__ best7 = nullptr, float* __restrict__ draw_counter = nullptr,
                                                          const int32_t* __restrict__ shape_dev = nullptr, int64_t rows_a = 0) {
    pose_window_step_body<THREADS>((int)blockIdx.x, poses7, g_o, g_d, dirs, m7, v7, g7_out, step_dev, ps, loss, min_loss, best7, draw_counter, shape_dev, rows_a);
}

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" int us_window_rays(const float* c2w_first, const float* poses7, const float* pool_depth, const float* pool_color,
                              const float* pool_dirs, const int64_t* idx, int64_t pool_size, int f_begin, int f_count, int64_t n_per_frame,
                              float* rays_o, float* rays_d, float* depth, float* color, float* dirs, void* stream) {
    US_REQUIRE(pool_depth && pool_color && pool_dirs && idx && rays_o && rays_d && depth && color, US_ERR_NULL, "us_window_rays: NULL pointer");
    US_REQUIRE(f_begin >= 0 && f_count >= 1 && pool_size >= 1 && n_per_frame >= 0, US_ERR_SHAPE, "us_window_rays: bad shape");
    US_REQUIRE(c2w_first || poses7, US_ERR_NULL, "us_window_rays: frame 0 needs its matrix (or, c2w_first NULL, every frame its pose)");
    US_REQUIRE((c2w_first && f_begin + f_count <= 1) || poses7, US_ERR_NULL, "us_window_rays: frames 1.. need poses7");
    const int64_t total = (int64_t)f_count * n_per_frame;
    if (total == 0) return US_OK;
    int64_t blocks = us_cdiv(total, 256); if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_window_rays, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, c2w_first, poses7, pool_depth, pool_color,
                       pool_dirs, idx, pool_size, f_begin, n_per_frame, total, rays_o, rays_d, depth, color, dirs);
    US_CHECK_LAUNCH("us_window_rays");
    return US_OK;
}

extern "C" int us_pose_window_step(float* poses7, int n_poses, const float* g_rays_o, const float* g_rays_d, const float* dirs,
                                   int64_t row_a, int64_t n_a, int first_pose_b, int64_t row_b, int64_t n_b, float* m7, float* v7,
                                   float* g7_out, double lr_q, double lr_t, double beta1, double beta2, double eps, float* step_dev,
                                   int flags, void* stream) {
    US_REQUIRE(poses7 && g_rays_o && g_rays_d && dirs, US_ERR_NULL, "us_pose_window_step: NULL pointer");
    US_REQUIRE(n_poses >= 1 && n_a >= 0 && n_b >= 0 && row_a >= 0 && row_b >= 0 && first_pose_b >= 0, US_ERR_SHAPE, "us_pose_window_step: bad shape");
    const int apply = (flags & US_POSE_GRAD_ONLY) ? 0 : 1, own = (flags & US_POSE_OWN_STEP) ? 1 : 0;
    US_REQUIRE(!apply || (m7 && v7 && step_dev), US_ERR_NULL, "us_pose_window_step: the optimiser step needs m7, v7 and step_dev");
    US_REQUIRE(!apply || own || ((uintptr_t)step_dev & 7u) == 0, US_ERR_SHAPE, "us_pose_window_step: step_dev (float[8]) must be 8-byte aligned");
    US_REQUIRE(!(apply && own) || n_poses == 1, US_ERR_CONFIG, "us_pose_window_step: US_POSE_OWN_STEP advances the count in the launch: one pose only");
    US_REQUIRE(apply || g7_out, US_ERR_NULL, "us_pose_window_step: US_POSE_GRAD_ONLY needs g7_out");
    PoseStep ps;
    ps.nA = n_a; ps.rowA = row_a; ps.nB = n_b; ps.rowB = row_b; ps.jB = first_pose_b;
    ps.lr_q = (float)lr_q; ps.lr_t = (float)lr_t; ps.b1 = (float)beta1; ps.b2 = (float)beta2; ps.eps = (float)eps;
    ps.own_step = own; ps.apply = apply;
    // (one pose over thousands of rays -- the tracker -- as ONE 1024-thread workgroup instead of 256 threads: measured 12.3 against 9.0 us:
    //  the wider reduction tree costs more than the shorter loop saves)
    hipLaunchKernelGGL(k_pose_window_step<256>, dim3((unsigned)n_poses), dim3(256), 0, (hipStream_t)stream, poses7, g_rays_o, g_rays_d, dirs, m7, v7,
                       g7_out, step_dev, ps);
    US_CHECK_LAUNCH("us_pose_window_step");
    return US_OK;
}

// us_pose_window_step with the window's shape read on the device: n_poses_cap workgroups are launched, the window's b - first run
extern "C" int us_arena_pose_step(float* poses7, int n_poses_cap, const int32_t* shape_dev, int64_t rows_a, const float* g_rays_o,
                                  const float* g_rays_d, const float* dirs, float* m7, float* v7, float* g7_out, double lr_q, double lr_t,
                                  double beta1, double beta2, double eps, float* step_dev, void* stream) {
    US_REQUIRE(poses7 && shape_dev && g_rays_o && g_rays_d && dirs && m7 && v7 && step_dev, US_ERR_NULL, "us_arena_pose_step: NULL pointer");
    US_REQUIRE(n_poses_cap >= 1 && rows_a >= 1, US_ERR_SHAPE, "us_arena_pose_step: bad shape");
    US_REQUIRE(((uintptr_t)step_dev & 7u) == 0, US_ERR_SHAPE, "us_arena_pose_step: step_dev (float[8]) must be 8-byte aligned");
    PoseStep ps;
    ps.nA = 0; ps.rowA = 0; ps.nB = 0; ps.rowB = 0; ps.jB = 0;
    ps.lr_q = (float)lr_q; ps.lr_t = (float)lr_t; ps.b1 = (float)beta1; ps.b2 = (float)beta2; ps.eps = (float)eps;
    ps.own_step = 0; ps.apply = 1;
    const float* nulf = nullptr; float* nulw = nullptr;
    hipLaunchKernelGGL(k_pose_window_step<256>, dim3((unsigned)n_poses_cap), dim3(256), 0, (hipStream_t)stream, poses7, g_rays_o, g_rays_d, dirs, m7, v7,
                       g7_out, step_dev, ps, nulf, nulw, nulw, nulw, shape_dev, rows_a);
    US_CHECK_LAUNCH("us_arena_pose_step");
    return US_OK;
}

// the tracker's pose step (src/Tracker.py:240-242 inside the loop of :333-348): us_pose_window_step for ONE pose with its own step count,
// plus the minimum-loss bookkeeping of the loop in the same launch -- loss[0] is the loss of THIS iteration (rendered at the pose as it is
// on entry); where it is below min_loss[0], min_loss and best7 take it and that pose.  NaN never counts as better (torch's `<`).
// draw_counter (nullable) is advanced by one: the counter of us_track_sample's in-kernel pixel draw, which must outlive the per-frame
// optimiser state (step_dev restarts at 0 with every frame; a draw keyed to it would pick the same pixels in every frame).
extern "C" int us_pose_track_step(float* pose7, const float* g_rays_o, const float* g_rays_d, const float* dirs, int64_t n_rays, float* m7,
                                  float* v7, float* g7_out, double lr_q, double lr_t, double beta1, double beta2, double eps, float* step_dev,
                                  const float* loss, float* min_loss, float* best7, float* draw_counter, void* stream) {
    US_REQUIRE(pose7 && g_rays_o && g_rays_d && dirs && m7 && v7 && step_dev, US_ERR_NULL, "us_pose_track_step: NULL pointer");
    US_REQUIRE(loss && min_loss && best7, US_ERR_NULL, "us_pose_track_step: loss, min_loss and best7 are required (us_pose_window_step has none)");
    US_REQUIRE(n_rays >= 0, US_ERR_SHAPE, "us_pose_track_step: bad shape");
    PoseStep ps;
    ps.nA = n_rays; ps.rowA = 0; ps.nB = 0; ps.rowB = 0; ps.jB = 0;
    ps.lr_q = (float)lr_q; ps.lr_t = (float)lr_t; ps.b1 = (float)beta1; ps.b2 = (float)beta2; ps.eps = (float)eps;
    ps.own_step = 1; ps.apply = 1;
    hipLaunchKernelGGL(k_pose_window_step<256>, dim3(1), dim3(256), 0, (hipStream_t)stream, pose7, g_rays_o, g_rays_d, dirs, m7, v7, g7_out,
                       step_dev, ps, loss, min_loss, best7, draw_counter);
    US_CHECK_LAUNCH("us_pose_track_step");
    return US_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// A keyframe's pixel pool (src/Mapper.py:329-337,516-523: `torch.randperm(H * W)[:int(0.1 * H * W)]`, then three gathers): a uniformly
// random subset of the frame's pixels WITHOUT repetition.  The reference sorts 816 000 random keys for it; here thread i takes pixel
// pi(i), pi a keyed pseudo-random bijection of [0, n_pixels) -- a four-round Feistel network on the next even power of two with cycle
// walking (values that land beyond n_pixels are permuted again: 1.3 rounds on average for a 680 x 1200 frame) -- so the first pool_size
// images are distinct pixels and no sort, no index tensor and no second pass exist.  One launch; has_zero[0] |= 1 where the pool holds
// a pixel without a depth (it decides whether the window's iterations carry the branch of src/utils/Renderer.py:104-130).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pool_mix(uint32_t x, uint32_t k) {
    x ^= k; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void k_pool_cut(const float* __restrict__ color, const float* __restrict__ depth, const float* __restrict__ dirs,
                                                  uint32_t n_pixels, uint32_t pool_size, int half_bits, uint32_t k0, uint32_t k1, uint32_t k2,
                                                  uint32_t k3, float* __restrict__ pool_color, float* __restrict__ pool_depth,
                                                  float* __restrict__ pool_dirs, int32_t* __restrict__ has_zero) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    bool zero = false;
    if (i < pool_size) {
        const uint32_t mask = (1u << half_bits) - 1u;
        uint32_t x = i;
        do {                                                       // a bijection of [0, 4^half_bits); walked until it lands in [0, n_pixels)
            uint32_t l = x >> half_bits, r = x & mask;
            const uint32_t keys[4] = {k0, k1, k2, k3};
#pragma unroll
            for (int q = 0; q < 4; ++q) { const uint32_t t = l ^ (pool_mix(r, keys[q]) & mask); l = r; r = t; }
            x = (l << half_bits) | r;
        } while (x >= n_pixels);
        const float d = depth[x];
        pool_depth[i] = d;
        zero = !(d > 0.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) { pool_color[i * 3 + c] = color[(size_t)x * 3 + c]; pool_dirs[i * 3 + c] = dirs[(size_t)x * 3 + c]; }
    }
    if (has_zero && __any(zero) && (threadIdx.x & 63) == 0) atomicOr(has_zero, 1);
}

extern "C" int us_pool_cut(const float* color, const float* depth, const float* dirs, int64_t n_pixels, int64_t pool_size, uint64_t seed,
                           float* pool_color, float* pool_depth, float* pool_dirs, int32_t* has_zero, void* stream) {
    US_REQUIRE(color && depth && dirs && pool_color && pool_depth && pool_dirs, US_ERR_NULL, "us_pool_cut: NULL pointer");
    US_REQUIRE(n_pixels >= 1 && n_pixels <= (1ll << 30) && pool_size >= 0 && pool_size <= n_pixels, US_ERR_SHAPE,
               "us_pool_cut: n_pixels %lld, pool_size %lld", (long long)n_pixels, (long long)pool_size);
    if (pool_size == 0) return US_OK;
    int half = 1;
    while ((1ll << (2 * half)) < n_pixels) ++half;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;                    // four round keys: splitmix64 of the seed
    uint32_t k[4];
    for (int q = 0; q < 4; q += 2) {
        z += 0x9E3779B97F4A7C15ull;
        uint64_t t = z; t = (t ^ (t >> 30)) * 0xBF58476D1CE4E5B9ull; t = (t ^ (t >> 27)) * 0x94D049BB133111EBull; t ^= t >> 31;
        k[q] = (uint32_t)t; k[q + 1] = (uint32_t)(t >> 32);
    }
    hipLaunchKernelGGL(k_pool_cut, dim3((unsigned)us_cdiv(pool_size, 256)), dim3(256), 0, (hipStream_t)stream, color, depth, dirs, (uint32_t)n_pixels,
                       (uint32_t)pool_size, half, k[0], k[1], k[2], k[3], pool_color, pool_depth, pool_dirs, has_zero);
    US_CHECK_LAUNCH("us_pool_cut");
    return US_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Keyframe overlap (Mapper.keyframe_selection_LC, src/Mapper.py:188-240): n_pix random pixels of the current frame with a depth, n_samples
// points along each ray between 0.8 d and d + 0.5, and for every keyframe the share of those points that project inside its image
// (20-pixel margin, in front of the camera).  The reference runs ~25 torch ops over [K, M, 4, 1] temporaries and a batched matrix
// inverse; here one workgroup per keyframe forms the points on the fly.  kf_frames[k]: row of pose_list (the estimated trajectory) that
// keyframe k uses; its world -> camera transform is the rigid inverse.
// ---------------------------------------------------------------------------------------------------------------
struct OverlapCam { float fx, fy, cx, cy; int H, W, edge; };
__global__ __launch_bounds__(256) void k_keyframe_overlap(const float* __restrict__ c2w, const float* __restrict__ depth, const int64_t* __restrict__ pix,
                                                          int n_pix, int n_samples, OverlapCam cam, const float* __restrict__ pose_list,
                                                          const int64_t* __restrict__ kf_frames, float* __restrict__ percent) {
    __shared__ int sh_in[4], sh_all[4];
    const float* Mk = pose_list + kf_frames[blockIdx.x] * 16;
    int n_in = 0, n_all = 0;
    for (int m = threadIdx.x; m < n_pix * n_samples; m += 256) {
        const int r = m / n_samples, sidx = m - r * n_samples;
        const int64_t p = pix[r];
        const float gd = depth[p];
        if (!(gd > 0.0f)) continue;                                 // Mapper.py:196-199: rays without a depth are left out
        const float x = (float)(p % cam.W), y = (float)(p / cam.W);
        const float dc[3] = {(x - cam.cx) / cam.fx, -(y - cam.cy) / cam.fy, -1.0f};
        const float t = n_samples > 1 ? (float)sidx / (float)(n_samples - 1) : 0.0f;
        const float z = gd * 0.8f * (1.0f - t) + (gd + 0.5f) * t;
        float pw[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            pw[a] = c2w[a * 4 + 3] + ((dc[0] * c2w[a * 4] + dc[1] * c2w[a * 4 + 1]) + dc[2] * c2w[a * 4 + 2]) * z;
        float pc[3];                                                // R^T (p - t): the keyframe's camera frame
#pragma unroll
        for (int a = 0; a < 3; ++a)
            pc[a] = Mk[a] * (pw[0] - Mk[3]) + Mk[4 + a] * (pw[1] - Mk[7]) + Mk[8 + a] * (pw[2] - Mk[11]);
        pc[0] = -pc[0];
        const float zz = pc[2] + 1e-5f;
        const float u = (cam.fx * pc[0] + cam.cx * pc[2]) / zz, v = (cam.fy * pc[1] + cam.cy * pc[2]) / zz;
        const bool in = u < (float)(cam.W - cam.edge) && u > (float)cam.edge && v < (float)(cam.H - cam.edge) && v > (float)cam.edge && zz < 0.0f;
        ++n_all; n_in += in ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) { n_in += __shfl_xor(n_in, o, 64); n_all += __shfl_xor(n_all, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh_in[threadIdx.x >> 6] = n_in; sh_all[threadIdx.x >> 6] = n_all; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int a = sh_in[0] + sh_in[1] + sh_in[2] + sh_in[3], b = sh_all[0] + sh_all[1] + sh_all[2] + sh_all[3];
        percent[blockIdx.x] = b > 0 ? (float)a / (float)b : 0.0f;
    }
}

extern "C" int us_keyframe_overlap(const float* c2w, const float* depth, const int64_t* pix, int n_pix, int n_samples, const float* intr_host4,
                                   int H, int W, int edge, const float* pose_list, const int64_t* kf_frames, int n_keyframes, float* percent,
                                   void* stream) {
    if (n_keyframes <= 0) return n_keyframes == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(c2w && depth && pix && intr_host4 && pose_list && kf_frames && percent, US_ERR_NULL, "us_keyframe_overlap: NULL pointer");
    US_REQUIRE(n_pix >= 1 && n_samples >= 1 && H >= 1 && W >= 1, US_ERR_SHAPE, "us_keyframe_overlap: bad shape");
    OverlapCam cam;
    cam.fx = intr_host4[0]; cam.fy = intr_host4[1]; cam.cx = intr_host4[2]; cam.cy = intr_host4[3]; cam.H = H; cam.W = W; cam.edge = edge;
    hipLaunchKernelGGL(k_keyframe_overlap, dim3((unsigned)n_keyframes), dim3(256), 0, (hipStream_t)stream, c2w, depth, pix, n_pix, n_samples, cam,
                       pose_list, kf_frames, percent);
    US_CHECK_LAUNCH("us_keyframe_overlap");
    return US_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// pose <-> matrix (src/common.py:182-208): the per-frame glue of both drivers.  On torch ops one conversion is a chain of ~30 / ~20
// small launches (0.43 / 0.25 ms of host time each, MI355X box) -- a third of a tracked frame; here one launch each.
// ---------------------------------------------------------------------------------------------------------------
// pytorch3d.transforms.matrix_to_quaternion (real part first): four candidates, the one with the largest |q_m| is returned; then
// the translation column.  extrapolate: n == 1 and c2w holds TWO matrices -- the constant-speed prediction 2 * pose(c2w[1]) - pose(c2w[0])
// of src/Tracker.py:317-320 comes out (element-wise on the 7 numbers, as there).
__device__ __forceinline__ void w_matrix_pose(const float* __restrict__ M, float q[7]) {
    const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[4], m11 = M[5], m12 = M[6], m20 = M[8], m21 = M[9], m22 = M[10];
    const float a[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22, 1.0f - m00 - m11 + m22};
    float qa[4];
    int best = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        qa[c] = a[c] > 0.0f ? sqrtf(a[c]) : 0.0f;                 // _sqrt_positive_part
        if (qa[c] > qa[best]) best = c;                            // argmax: the first of equal maxima
    }
    const float cand[4][4] = {{qa[0] * qa[0], m21 - m12, m02 - m20, m10 - m01},
                              {m21 - m12, qa[1] * qa[1], m10 + m01, m02 + m20},
                              {m02 - m20, m10 + m01, qa[2] * qa[2], m12 + m21},
                              {m10 - m01, m20 + m02, m21 + m12, qa[3] * qa[3]}};
    const float den = 2.0f * fmaxf(qa[best], 0.1f);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = cand[0][c];
        if (best == 1) v = cand[1][c]; else if (best == 2) v = cand[2][c]; else if (best == 3) v = cand[3][c];
        q[c] = v / den;
    }
    q[4] = M[3]; q[5] = M[7]; q[6] = M[11];
}

__global__ __launch_bounds__(64) void k_matrix_to_pose(const float* __restrict__ c2w, int n, int extrapolate, float* __restrict__ pose7) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float q[7];
    w_matrix_pose(c2w + (int64_t)i * 16, q);
    if (extrapolate) {
        float q1[7];
        w_matrix_pose(c2w + 16, q1);
        // Tracker.py:317-320 extrapolates the two quaternions element by element.  q and -q are one rotation, and which of the two
        // matrix_to_quaternion returns depends on its branch (pytorch3d: the largest of the four candidates' norms): where the branch
        // switches between the two frames, or the real part is negative under one convention and positive under the other, 2 q1 - q0
        // would point anywhere.  Put q1 on q0's hemisphere first: equal to the reference wherever its prediction is sane (q0 . q1 > 0
        // already), and independent of the helper's sign convention -- the one unpinned detail that could change a result.
        if (q[0] * q1[0] + q[1] * q1[1] + q[2] * q1[2] + q[3] * q1[3] < 0.0f) {
#pragma unroll
            for (int c = 0; c < 4; ++c) q1[c] = -q1[c];
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) q[c] = 2.0f * q1[c] - q[c];
    }
#pragma unroll
    for (int c = 0; c < 7; ++c) pose7[(int64_t)i * 7 + c] = q[c];
}

__global__ __launch_bounds__(64) void k_pose_to_matrix(const float* __restrict__ pose7, int n, float* __restrict__ c2w) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float* q = pose7 + (int64_t)i * 7;
    float R[9];
    w_quat_rot(q, R);
    float* M = c2w + (int64_t)i * 16;
    M[0] = R[0]; M[1] = R[1]; M[2] = R[2]; M[3] = q[4];
    M[4] = R[3]; M[5] = R[4]; M[6] = R[5]; M[7] = q[5];
    M[8] = R[6]; M[9] = R[7]; M[10] = R[8]; M[11] = q[6];
    M[12] = 0.0f; M[13] = 0.0f; M[14] = 0.0f; M[15] = 1.0f;
}

extern "C" int us_matrix_to_cam_pose(const float* c2w, int n, int extrapolate, float* pose7, void* stream) {
    US_REQUIRE(c2w && pose7, US_ERR_NULL, "us_matrix_to_cam_pose: NULL pointer");
    US_REQUIRE(n >= 0 && (!extrapolate || n == 1), US_ERR_SHAPE, "us_matrix_to_cam_pose: n = %d (the extrapolation takes two matrices and returns one pose)", n);
    if (n == 0) return US_OK;
    hipLaunchKernelGGL(k_matrix_to_pose, dim3((unsigned)us_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, c2w, n, extrapolate, pose7);
    US_CHECK_LAUNCH("us_matrix_to_cam_pose");
    return US_OK;
}

extern "C" int us_cam_pose_to_matrix(const float* pose7, int n, float* c2w, void* stream) {
    US_REQUIRE(pose7 && c2w, US_ERR_NULL, "us_cam_pose_to_matrix: NULL pointer");
    US_REQUIRE(n >= 0, US_ERR_SHAPE, "us_cam_pose_to_matrix: n = %d", n);
    if (n == 0) return US_OK;
    hipLaunchKernelGGL(k_pose_to_matrix, dim3((unsigned)us_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, pose7, n, c2w);
    US_CHECK_LAUNCH("us_cam_pose_to_matrix");
    return US_OK;
}
