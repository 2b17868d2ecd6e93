// dec_adam_dev.h -- the decoder param group's fixed-order reductions + Adam, shared by its own launch (mlp.hip: k_mlp_reduce_pair_adam)
// and by the tables' optimiser launch that carries them as extra slices (render.hip: k_adam_segs_model).
#pragma once
#include "us_common.h"

struct DecAdam { float lr, one_minus_b1, b2, one_minus_b2, eps; const float* step_dev; };
__device__ __forceinline__ void dec_adam_apply(float g, float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const DecAdam& ad) {
    const double* aux = reinterpret_cast<const double*>(ad.step_dev + 2);          // as k_adam_segs (render.hip) reads them
    const float step_size = (float)((double)ad.lr / aux[0]), bc2_sqrt = (float)aux[1];
    const float m0 = *m, v0 = *v;
    const float mi = m0 + ad.one_minus_b1 * (g - m0);
    const float vi = v0 * ad.b2 + (ad.one_minus_b2 * g) * g;
    const float denom = sqrtf(vi) / bc2_sqrt + ad.eps;
    *p = *p + (-step_size) * (mi / denom);
    *m = mi; *v = vi;
}

// everything the decoder group's slices need (pa / pb: the partial rows the decoders' backward launch left)
struct DecGroup {
    const float *pa, *pb; int n_rows, npa, npb;
    float *ga, *gb, *Pa, *Pb, *ma, *mb, *va, *vb;
    const float* beta_part; long long n_rays; float *p_beta, *g_beta, *m_beta, *v_beta;
    DecAdam ad;
};
// the tables' optimiser launch with the decoder group (and, poses != NULL, a joint_opt window's pose group) riding along (render.hip)
int us_adam_segments_model(float* p, float* g, float* m, float* v, int n_seg, const int64_t* seg_off, const int64_t* seg_n, const double* seg_lr,
                           double beta1, double beta2, double eps, float* step_dev, unsigned zero_grad_mask, void* stream, const DecGroup& dg,
                           const us_pose_step_desc* poses);
