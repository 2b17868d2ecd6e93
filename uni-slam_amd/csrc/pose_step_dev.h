// pose_step_dev.h -- the pose group's gradient + Adam (one workgroup per optimised frame), shared by its own launches (window.hip:
// us_pose_window_step / us_arena_pose_step / us_pose_track_step) and by the model's optimiser launch that carries it as leading workgroups
// (render.hip: k_adam_segs_model, us_adam_step_model).
#pragma once
#include "us_common.h"

struct PoseStep {
    int64_t nA;                 // rows [j * nA, (j + 1) * nA) belong to optimised pose j ... shifted by first_row_A
    int64_t rowA;               // first row of pose 0's block in segment A
    int64_t nB, rowB;           // second segment (the extra rays of the newest frames, src/Mapper.py:385-393); nB == 0: none
    int     jB;                 // first pose that owns rows in segment B
    float   lr_q, lr_t, b1, b2, eps;
    int     own_step;           // 1: step_dev is float[1], advanced here, fp32 bias corrections (the tracker's Adam, one workgroup)
                                // 0: step_dev is the float[8] of us_adam_step_inc, already advanced for this step
    int     apply;              // 0: gradient only
};

template <int THREADS>
__device__ __forceinline__ void pose_window_step_body(const int j, float* __restrict__ poses7, const float* __restrict__ g_o, const float* __restrict__ g_d,
                                                          const float* __restrict__ dirs, float* __restrict__ m7, float* __restrict__ v7,
                                                          float* __restrict__ g7_out, float* __restrict__ step_dev, PoseStep ps,
                                                          const float* __restrict__ loss, float* __restrict__ min_loss,
                                                          float* __restrict__ best7, float* __restrict__ draw_counter,
                                                          const int32_t* __restrict__ shape_dev, int64_t rows_a) {
    __shared__ double sh[12][THREADS / 64];
    __shared__ float g7[7];
    if (shape_dev) {
        // the window's shape on the device (us_arena_pose_step; layout of us_arena_window_sample): the launch covers the arena's pose
        // capacity, workgroups beyond the window's optimised frames leave; rows as laid out there (first block from row 0, extra block
        // from row rows_a)
        const int b = shape_dev[0], n_per = shape_dev[1], xf = shape_dev[2], xn = shape_dev[3], first = shape_dev[4] != 0 ? 1 : 0;
        if (j >= b - first) return;                                // (uniform over the workgroup)
        const int fb = (b - xf) > first ? (b - xf) : first;        // the first optimised frame that owns rows of the extra block
        ps.nA = n_per; ps.rowA = (int64_t)first * n_per;
        ps.nB = xf > 0 ? xn : 0; ps.jB = fb - first; ps.rowB = rows_a + (int64_t)(fb - (b - xf)) * xn;
    }
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0;
    for (int seg = 0; seg < 2; ++seg) {
        int64_t n, r0;
        if (seg == 0) { n = ps.nA; r0 = ps.rowA + (int64_t)j * ps.nA; }
        else { if (ps.nB == 0 || j < ps.jB) break; n = ps.nB; r0 = ps.rowB + (int64_t)(j - ps.jB) * ps.nB; }
        for (int64_t t = threadIdx.x; t < n; t += THREADS) {
            const int64_t r = r0 + t;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float ga = g_d[r * 3 + a];
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[a * 3 + b] += (double)(ga * dirs[r * 3 + b]);
                acc[9 + a] += (double)g_o[r * 3 + a];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) sh[k][wave] = v;
    }
    __syncthreads();
    float* pose = poses7 + (int64_t)j * 7;
    if (threadIdx.x == 0) {
        float G[9], gt[3];
        for (int k = 0; k < 12; ++k) {
            double v = 0.0;
            for (int w = 0; w < THREADS / 64; ++w) v += sh[k][w];
            if (k < 9) G[k] = (float)v; else gt[k - 9] = (float)v;
        }
        const float r = pose[0], i = pose[1], jj = pose[2], k = pose[3];
        const float s = 2.0f / (r * r + i * i + jj * jj + k * k);
        const float M[9] = {-(jj * jj + k * k), i * jj - k * r, i * k + jj * r, i * jj + k * r, -(i * i + k * k), jj * k - i * r,
                            i * k - jj * r, jj * k + i * r, -(i * i + jj * jj)};
        const float dMr[9] = {0, -k, jj, k, 0, -i, -jj, i, 0};
        const float dMi[9] = {0, jj, k, jj, -2 * i, -r, k, r, -2 * i};
        const float dMj[9] = {-2 * jj, i, r, i, 0, k, -r, k, -2 * jj};
        const float dMk[9] = {-2 * k, -r, i, r, -2 * k, jj, i, jj, 0};
        float gm = 0, gr = 0, gi = 0, gj = 0, gk = 0;
        for (int e = 0; e < 9; ++e) { gm += G[e] * M[e]; gr += G[e] * dMr[e]; gi += G[e] * dMi[e]; gj += G[e] * dMj[e]; gk += G[e] * dMk[e]; }
        const float ds = -s * s;                                   // d s / d q_m = -s^2 q_m
        g7[0] = ds * r * gm + s * gr; g7[1] = ds * i * gm + s * gi;
        g7[2] = ds * jj * gm + s * gj; g7[3] = ds * k * gm + s * gk;
        g7[4] = gt[0]; g7[5] = gt[1]; g7[6] = gt[2];
    }
    float step = 0.0f;
    if (ps.apply && ps.own_step) step = step_dev[0] + 1.0f;       // read by every thread before thread 0 stores the new count
    __syncthreads();
    const int e = threadIdx.x;
    if (e < 7) {
        const float gi = g7[e];
        if (g7_out) g7_out[(int64_t)j * 7 + e] = gi;
        if (loss) {                                                // the tracker's candidate (src/Tracker.py:346-348): the pose the loss was
            const bool better = loss[0] < min_loss[0];             // rendered at is kept while it is the best so far; the seven lanes read
            if (better) {                                          // the old minimum before lane 0 replaces it (one wave, program order)
                best7[e] = pose[e];
                if (e == 0) min_loss[0] = loss[0];
            }
        }
        if (ps.apply) {
            float step_size, bc2s;
            const float lr = e < 4 ? ps.lr_q : ps.lr_t;
            if (ps.own_step) {                                     // torch.optim.Adam(capturable) arithmetic, as us_pose_adam_step
                const float bc1 = 1.0f - powf(ps.b1, step);
                bc2s = sqrtf(1.0f - powf(ps.b2, step));
                step_size = lr / bc1;
            } else {                                               // the corrections us_adam_step_inc left, as k_adam_segs reads them
                const double* aux = reinterpret_cast<const double*>(step_dev + 2);
                step_size = (float)((double)lr / aux[0]);
                bc2s = (float)aux[1];
            }
            float* m = m7 + (int64_t)j * 7; float* v = v7 + (int64_t)j * 7;
            const float m0 = m[e], v0 = v[e];
            const float mi = m0 + (1.0f - ps.b1) * (gi - m0);
            const float vi = v0 * ps.b2 + ((1.0f - ps.b2) * gi) * gi;
            const float denom = sqrtf(vi) / bc2s + ps.eps;
            pose[e] = pose[e] + (-step_size) * (mi / denom);
            m[e] = mi; v[e] = vi;
        }
    }
    if (ps.apply && ps.own_step && threadIdx.x == 0) step_dev[0] = step;
    if (draw_counter && threadIdx.x == 0) draw_counter[0] += 1.0f;
}

