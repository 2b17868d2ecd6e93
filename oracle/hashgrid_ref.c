/*
 * oracle/hashgrid_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, gcc) of the multi-resolution hash-grid encoding that
 * Uni-SLAM obtains from the un-vendored dependency
 *   tinycudann @ 2ec562e853e6f482b5d09168705205f46358fb39   (reference requirements.txt:90)
 * constructed at  reference src/UNISLAM.py:241-253  (otype HashGrid, n_levels 16,
 * n_features_per_level 2, base_resolution 16, per_level_scale, log2_hashmap_size) and
 * called at       reference src/networks/decoders.py:101-103.
 *
 * The dependency's source is absent from /root/reference, so this file restates the
 * PUBLISHED tiny-cuda-nn algorithm (include/tiny-cuda-nn/encodings/grid.h of that pin:
 * grid_scale, grid_resolution, pos_fract, grid_index, coherent prime hash, kernel_grid,
 * kernel_grid_backward, kernel_grid_backward_input and the offset-table constructor).
 * PARITY UNPINNED for this part: the reference holds no test, golden vector or fixture
 * at this boundary and tiny-cuda-nn cannot be built or run here (needs CUDA).  The
 * restatement is pinned only by spec-derived known-answer tests (tests/test_oracle_kat.py)
 * and by the level tables of SURVEY.md Appendix A.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define ORC_MAX_LEVELS 32

typedef struct {
    uint32_t n_levels;
    uint32_t n_features;          /* features per level (F) */
    uint32_t log2_hashmap_size;
    uint32_t base_resolution;
    float    per_level_scale;
    float    scale[ORC_MAX_LEVELS];       /* grid_scale(level) */
    uint32_t resolution[ORC_MAX_LEVELS];  /* grid_resolution(scale) */
    uint32_t offset[ORC_MAX_LEVELS + 1];  /* in table ENTRIES (F floats each) */
    uint32_t n_params;                    /* F * offset[n_levels] */
} orc_grid_desc;

/* tcnn grid.h: grid_scale() -- "exp2f(level * log2_per_level_scale) * base_resolution - 1.0f" */
static float orc_grid_scale(uint32_t level, float log2_pls, uint32_t base_resolution) {
    return exp2f((float)level * log2_pls) * (float)base_resolution - 1.0f;
}
/* tcnn grid.h: grid_resolution() -- "(uint32_t)ceilf(scale) + 1" */
static uint32_t orc_grid_resolution(float scale) { return (uint32_t)ceilf(scale) + 1u; }

static uint32_t orc_next_multiple(uint32_t v, uint32_t d) { return ((v + d - 1u) / d) * d; }

/* tcnn GridEncodingTemplated constructor: per-level parameter counts and offsets. */
int orc_grid_desc_init(orc_grid_desc* d, uint32_t n_levels, uint32_t n_features,
                       uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale) {
    if (!d || n_levels == 0 || n_levels > ORC_MAX_LEVELS || n_features == 0) return -1;
    memset(d, 0, sizeof(*d));
    d->n_levels = n_levels; d->n_features = n_features;
    d->log2_hashmap_size = log2_hashmap_size; d->base_resolution = base_resolution;
    d->per_level_scale = per_level_scale;
    const float log2_pls = log2f(per_level_scale);
    uint32_t offset = 0;
    for (uint32_t l = 0; l < n_levels; ++l) {
        const float scale = orc_grid_scale(l, log2_pls, base_resolution);
        const uint32_t res = orc_grid_resolution(scale);
        d->scale[l] = scale; d->resolution[l] = res;
        const uint32_t max_params = 0xFFFFFFFFu / 2u;
        uint32_t params_in_level;
        if (powf((float)res, 3.0f) > (float)max_params) params_in_level = max_params;
        else params_in_level = res * res * res;
        params_in_level = orc_next_multiple(params_in_level, 8u);
        const uint32_t cap = 1u << log2_hashmap_size;
        if (params_in_level > cap) params_in_level = cap;   /* GridType::Hash */
        d->offset[l] = offset;
        offset += params_in_level;
    }
    d->offset[n_levels] = offset;
    d->n_params = offset * n_features;
    return 0;
}

/* tcnn grid.h: grid_index<3, CoherentPrime>() */
static uint32_t orc_grid_index(uint32_t hashmap_size, uint32_t res, const uint32_t g[3]) {
    uint32_t stride = 1, index = 0;
    for (uint32_t dim = 0; dim < 3 && stride <= hashmap_size; ++dim) {
        index += g[dim] * stride;
        stride *= res;
    }
    if (hashmap_size < stride) {
        /* coherent_prime_hash: factors {1, 2654435761, 805459861}, uint32 wrap-around */
        index = (g[0] * 1u) ^ (g[1] * 2654435761u) ^ (g[2] * 805459861u);
    }
    return index % hashmap_size;
}

/* tcnn grid.h: pos_fract() with identity interpolation (InterpolationType::Linear) */
static void orc_pos_fract(float in, float scale, float* pos, uint32_t* pos_grid) {
    float p = fmaf(scale, in, 0.5f);
    float t = floorf(p);
    *pos_grid = (uint32_t)(int)t;
    *pos = p - t;
}

/* 8 corner indices per (point, level): idx_out[N][L][8], corner c: bit d of c selects +1 in dim d */
void orc_hashgrid_indices(const orc_grid_desc* d, const float* x, int64_t N, uint32_t* idx_out) {
    const uint32_t L = d->n_levels;
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        for (uint32_t l = 0; l < L; ++l) {
            const uint32_t hs = d->offset[l + 1] - d->offset[l];
            float pos[3]; uint32_t g[3];
            for (int k = 0; k < 3; ++k) orc_pos_fract(x[i * 3 + k], d->scale[l], &pos[k], &g[k]);
            for (uint32_t c = 0; c < 8; ++c) {
                uint32_t gl[3];
                for (int k = 0; k < 3; ++k) gl[k] = g[k] + ((c >> k) & 1u);
                idx_out[(i * L + l) * 8 + c] = orc_grid_index(hs, d->resolution[l], gl);
            }
        }
    }
}

/*
 * tcnn kernel_grid (forward).  out[N][L*F] row-major (the torch binding's view),
 * dy_dx (optional) [N][L*F][3].
 */
void orc_hashgrid_fwd(const orc_grid_desc* d, const float* params, const float* x, int64_t N,
                      float* out, float* dy_dx) {
    const uint32_t L = d->n_levels, F = d->n_features, C = L * F;
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        for (uint32_t l = 0; l < L; ++l) {
            const float* grid = params + (size_t)d->offset[l] * F;
            const uint32_t hs = d->offset[l + 1] - d->offset[l];
            const uint32_t res = d->resolution[l];
            const float scale = d->scale[l];
            float pos[3]; uint32_t g[3];
            for (int k = 0; k < 3; ++k) orc_pos_fract(x[i * 3 + k], scale, &pos[k], &g[k]);
            float result[8] = {0};
            for (uint32_t c = 0; c < 8; ++c) {
                float w = 1.0f; uint32_t gl[3];
                for (int k = 0; k < 3; ++k) {
                    if (((c >> k) & 1u) == 0) { w *= 1.0f - pos[k]; gl[k] = g[k]; }
                    else { w *= pos[k]; gl[k] = g[k] + 1u; }
                }
                const uint32_t idx = orc_grid_index(hs, res, gl) * F;
                for (uint32_t f = 0; f < F; ++f) result[f] = fmaf(w, grid[idx + f], result[f]);
            }
            for (uint32_t f = 0; f < F; ++f) out[i * C + l * F + f] = result[f];
            if (dy_dx) {
                for (uint32_t gd = 0; gd < 3; ++gd) {
                    float grads[8] = {0};
                    for (uint32_t c = 0; c < 4; ++c) {
                        float w = scale; uint32_t gl[3];
                        for (uint32_t ng = 0; ng < 2; ++ng) {
                            const uint32_t dim = ng >= gd ? ng + 1 : ng;
                            if (((c >> ng) & 1u) == 0) { w *= 1.0f - pos[dim]; gl[dim] = g[dim]; }
                            else { w *= pos[dim]; gl[dim] = g[dim] + 1u; }
                        }
                        gl[gd] = g[gd];
                        const uint32_t il = orc_grid_index(hs, res, gl) * F;
                        gl[gd] = g[gd] + 1u;
                        const uint32_t ir = orc_grid_index(hs, res, gl) * F;
                        for (uint32_t f = 0; f < F; ++f)
                            grads[f] += w * (grid[ir + f] - grid[il + f]) * 1.0f; /* pos_derivative == 1 */
                    }
                    for (uint32_t f = 0; f < F; ++f) dy_dx[((size_t)i * C + l * F + f) * 3 + gd] = grads[f];
                }
            }
        }
    }
}

/*
 * tcnn kernel_grid_backward: grad_params[idx*F+f] += w * dL_dy[i][l*F+f].
 * The GPU original uses atomicAdd (order undefined); here the sum runs in point order,
 * accumulated in double and rounded once, so it is a deterministic reference value.
 * grad_params (float, n_params) is OVERWRITTEN.
 */
void orc_hashgrid_bwd_params(const orc_grid_desc* d, const float* x, const float* dL_dy, int64_t N,
                             float* grad_params) {
    const uint32_t L = d->n_levels, F = d->n_features, C = L * F;
    double* acc = (double*)calloc((size_t)d->n_params, sizeof(double));
    if (!acc) return;
    #pragma omp parallel for schedule(static)
    for (uint32_t l = 0; l < L; ++l) {          /* levels own disjoint param ranges: no races */
        double* gacc = acc + (size_t)d->offset[l] * F;
        const uint32_t hs = d->offset[l + 1] - d->offset[l];
        const uint32_t res = d->resolution[l];
        for (int64_t i = 0; i < N; ++i) {
            float pos[3]; uint32_t g[3];
            for (int k = 0; k < 3; ++k) orc_pos_fract(x[i * 3 + k], d->scale[l], &pos[k], &g[k]);
            for (uint32_t c = 0; c < 8; ++c) {
                float w = 1.0f; uint32_t gl[3];
                for (int k = 0; k < 3; ++k) {
                    if (((c >> k) & 1u) == 0) { w *= 1.0f - pos[k]; gl[k] = g[k]; }
                    else { w *= pos[k]; gl[k] = g[k] + 1u; }
                }
                const uint32_t idx = orc_grid_index(hs, res, gl) * F;
                for (uint32_t f = 0; f < F; ++f)
                    gacc[idx + f] += (double)(w * dL_dy[i * C + l * F + f]);
            }
        }
    }
    for (size_t k = 0; k < d->n_params; ++k) grad_params[k] = (float)acc[k];
    free(acc);
}

/* tcnn kernel_grid_backward_input: dL_dx[i][dim] = sum_k dL_dy[i][k] * dy_dx[i][k][dim] */
void orc_hashgrid_bwd_input(const float* dL_dy, const float* dy_dx, int64_t N, uint32_t C, float* dL_dx) {
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        float r[3] = {0, 0, 0};
        for (uint32_t k = 0; k < C; ++k) {
            const float g = dL_dy[i * C + k];
            for (int dim = 0; dim < 3; ++dim) r[dim] += g * dy_dx[((size_t)i * C + k) * 3 + dim];
        }
        for (int dim = 0; dim < 3; ++dim) dL_dx[i * 3 + dim] = r[dim];
    }
}
