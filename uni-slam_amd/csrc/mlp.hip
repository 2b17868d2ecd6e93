// mlp.hip -- fully fused tiny MLP (forward and backward) on the gfx950 matrix cores.
//
// Replaces tcnn.Network("FullyFusedMLP") (reference src/networks/decoders.py:50-70, calls :123,:148) and the
// nn.Linear stacks of the torch path (src/networks/decoders.py:74-84, :125-128, :150-153).
//
// Orientation.  Everything is computed TRANSPOSED: H_l^T[neuron][point] = W_l[neuron][k] * H_{l-1}^T[k][point].
// With v_mfma_f32_16x16x4_f32 (A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15], D[row=4*(lane>>4)+reg][col=lane&15])
// a 16-neuron x 16-point accumulator tile has the point on the lane and 4 neurons in its 4 registers, which is
// exactly the B-operand shape of the next layer's MFMAs when K-step s takes register (s&3) of tile (s>>2):
//     k(s, g) = 16*(s>>2) + 4*g + (s&3),  g = lane>>4
// so activations never leave the register file between layers (no LDS, no cross-lane moves); only the weight
// operand is read from LDS (one ds_read_b128 feeds 4 K-steps x 4 point tiles = 16 MFMAs).  The input features
// use the same k(s,g) map, so a lane loads two float4 per point straight from the row-major [N][32] tensor.
// f32-input MFMA is bit-for-bit a k-ordered fmaf chain, so this path matches an fp32 reference to rounding;
// the work is ~30 kFLOP/point, <1% of the fp32 matrix peak at the target ray rate, so exact fp32 is affordable.
//
// Backward (one launch): recompute the hidden activations, then per layer
//     dW_l  += dH_l  * H_{l-1}^T      reduction over POINTS -> both operands are transposed through a per-wave LDS
//                                     scratch ([neuron][64 points], 272-byte rows), accumulated in MFMA registers
//                                     across the whole grid-stride loop and flushed once with float atomics;
//     dH_{l-1} = W_l^T * dH_l (.) relu'   same chained form as the forward, with W^T also resident in LDS.
#include "us_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));

#define MLP_WAVES 4                       // forward kernel
#ifndef MLP_FWD_MAX_WG
#define MLP_FWD_MAX_WG 512                // 2 workgroups per CU: each wave loops over a few chunks with the next one prefetched
#endif
#define MLP_THREADS (MLP_WAVES * 64)

__device__ __forceinline__ v4f mfma4(float a, float b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// NQ_ = 16-point tiles per wave iteration, WAVES_ = waves per workgroup.  Forward: 4 tiles (2 at width 64), 4 waves.
// Backward: 2 tiles and 8 waves (4 at width 64): the per-wave transposing scratch then fits 8 waves = 2 per SIMD, so
// one wave's LDS round trips overlap another wave's MFMAs.
template <int NIN, int WIDTH, int NHID, int NQ_ = ((WIDTH == 64) ? 2 : 4), int WAVES_ = MLP_WAVES>
struct MlpCfg {
    static constexpr int N_IN = NIN, W = WIDTH, NH = NHID;
    static constexpr int NQ = NQ_;
    static constexpr int WAVES = WAVES_;
    static constexpr int PTS = 16 * NQ;               // points per wave iteration
    static constexpr int SCR_STRIDE = PTS + 4;        // scratch row: PTS points + 4 pad floats (16-byte aligned rows)
    static constexpr int MT = WIDTH / 16;          // 16-row tiles of a hidden layer
    static constexpr int KB_IN = NIN / 16;         // 16-wide K blocks of the input layer
    static constexpr int KB_H = WIDTH / 16;
    static constexpr int S_IN = NIN + 4;           // LDS row strides (floats)
    static constexpr int S_W = WIDTH + 4;
    static constexpr int S_O = 16 + 4;
    // flat parameter offsets (floats)
    static constexpr int P_W0 = 0;
    static constexpr int P_WH = WIDTH * NIN;
    static constexpr int P_WL = P_WH + (NHID - 1) * WIDTH * WIDTH;
    static constexpr int N_W = P_WL + 16 * WIDTH;
    static constexpr int P_B0 = N_W;
    static constexpr int P_BH = P_B0 + WIDTH;
    static constexpr int P_BL = P_BH + (NHID - 1) * WIDTH;
    static constexpr int N_B = NHID * WIDTH + 16;
    // LDS offsets (floats): forward weights, biases, then transposed weights (backward only)
    static constexpr int L_W0 = 0;
    static constexpr int L_WH = L_W0 + WIDTH * S_IN;
    static constexpr int L_WL = L_WH + (NHID - 1) * WIDTH * S_W;
    static constexpr int L_B = L_WL + 16 * S_W;
    static constexpr int L_FWD_END = L_B + N_B;
    static constexpr int L_W0T = ((L_FWD_END + 3) / 4) * 4;
    static constexpr int L_WHT = L_W0T + NIN * S_W;
    static constexpr int L_WLT = L_WHT + (NHID - 1) * WIDTH * S_W;
    static constexpr int L_BWD_END = L_WLT + WIDTH * S_O;
    static constexpr int SCR_ROWS_A = (NIN > WIDTH ? NIN : WIDTH);
    static constexpr int SCR_ROWS_B = (WIDTH > 16 ? WIDTH : 16);
    static constexpr int L_SCR = ((L_BWD_END + 3) / 4) * 4;
    static constexpr int SCR_PER_WAVE = (SCR_ROWS_A + SCR_ROWS_B) * SCR_STRIDE;
    static constexpr int L_TOTAL_BWD = L_SCR + WAVES * SCR_PER_WAVE;
};

// out[q][m] = W(rows 16m..16m+15) * in + bias      (KB 16-wide K blocks; W in LDS with row stride SW)
template <int NQ, int KB, int MT, int SW>
__device__ __forceinline__ void dense(const float* w, const float* bias, const v4f (&in)[NQ][KB],
                                      v4f (&out)[NQ][MT], int row, int g) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        v4f b0 = {0.f, 0.f, 0.f, 0.f};
        if (bias) b0 = *reinterpret_cast<const v4f*>(bias + 16 * m + 4 * g);
#pragma unroll
        for (int q = 0; q < NQ; ++q) out[q][m] = b0;
#pragma unroll
        for (int b = 0; b < KB; ++b) {
            const v4f a = *reinterpret_cast<const v4f*>(w + (16 * m + row) * SW + 16 * b + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < NQ; ++q) out[q][m] = mfma4(a[c], in[q][b][c], out[q][m]);
        }
    }
}

template <int NQ, int MT>
__device__ __forceinline__ void relu_(v4f (&h)[NQ][MT]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // ONE instruction: v_max_i32(bits, 0) -- a negative float (and -0.0) is a negative integer, a positive one a positive integer,
                // so the signed-integer maximum with 0 IS ReLU for every non-NaN x.  fmaxf(x, 0) compiles to two (a canonicalising
                // v_max_f32 x, x in front; so does v_med3(x, 0, inf), which LLVM folds back to maxnum), and the decoder kernels are bound by
                // VALU issue (profiles/r06_mlp_pmc.txt: 91 % of the forward kernel's cycles issue a VALU instruction).  Not inline assembly: the
                // hazard recogniser does not see through it, and a VALU read of an MFMA result needs wait states (an asm v_max read stale
                // accumulators in the f32 kernel).
                const int b = __float_as_int(h[q][m][r]);
                h[q][m][r] = __int_as_float(b > 0 ? b : 0);
            }
}

// dh *= (h > 0)
template <int NQ, int MT>
__device__ __forceinline__ void relu_bwd_(v4f (&dh)[NQ][MT], const v4f (&h)[NQ][MT]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) dh[q][m][r] = h[q][m][r] > 0.0f ? dh[q][m][r] : 0.0f;
}

#include "act_dev.h"

// cooperative load of the flat parameter vector [W0 | WH | WL | biases] into the LDS images.  All of a thread's global loads
// are issued before the first LDS store (a load -> store -> load ... loop cost ~11 dependent round trips = most of the
// kernels' fixed 9 us).
template <typename C, bool BWD, int THREADS>
__device__ __forceinline__ void load_weights(float* lds, const float* __restrict__ params, bool has_bias) {
    constexpr int NIN = C::N_IN, WIDTH = C::W, NHID = C::NH;
    constexpr int NP = C::N_W + C::N_B, PER = (NP + THREADS - 1) / THREADS;
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = (int)threadIdx.x + j * THREADS;
        v[j] = (k < C::N_W || (has_bias && k < NP)) ? params[k] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = (int)threadIdx.x + j * THREADS;
        if (k < WIDTH * NIN) {
            const int r = k / NIN, c = k % NIN;
            lds[C::L_W0 + r * C::S_IN + c] = v[j];
            if (BWD) lds[C::L_W0T + c * C::S_W + r] = v[j];
        } else if (NHID == 2 && k < C::P_WL) {
            const int i = k - C::P_WH, r = i / WIDTH, c = i % WIDTH;
            lds[C::L_WH + r * C::S_W + c] = v[j];
            if (BWD) lds[C::L_WHT + c * C::S_W + r] = v[j];
        } else if (k < C::N_W) {
            const int i = k - C::P_WL, r = i / WIDTH, c = i % WIDTH;
            lds[C::L_WL + r * C::S_W + c] = v[j];
            if (BWD) lds[C::L_WLT + c * C::S_O + r] = v[j];
        } else if (k < NP) {
            lds[C::L_B + (k - C::N_W)] = v[j];
        }
    }
}

// input features of 64 points as B operands: xb[q][b][c] = in[p(q)][16b + 4g + c]
// LM: level-major planes [NIN/2][N][2]: features 16b+4g .. +3 are planes 8b+2g and 8b+2g+1 (two float2 loads)
// CLAMP (forward kernels: a row beyond n is computed and never stored): rows beyond n read row n - 1 instead of taking zeros -- no zero
// initialisation of the 16 input registers and no divergent branch around the loads of every chunk
template <int NQ, int NIN, bool CLAMP = false>
__device__ __forceinline__ void load_inputs(const float* __restrict__ in, int64_t base, int64_t n, int row, int g,
                                            v4f (&xb)[NQ][NIN / 16], int lm) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int64_t p = base + 16 * q + row;
        if (CLAMP) p = p < n ? p : n - 1;
#pragma unroll
        for (int b = 0; b < NIN / 16; ++b) {
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (CLAMP || p < n) {
                if (lm & 1) {
                    const float2 lo = *reinterpret_cast<const float2*>(in + ((int64_t)(8 * b + 2 * g) * n + p) * 2);
                    const float2 hi = *reinterpret_cast<const float2*>(in + ((int64_t)(8 * b + 2 * g + 1) * n + p) * 2);
                    v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
                } else {
                    v = *reinterpret_cast<const v4f*>(in + p * NIN + 16 * b + 4 * g);
                }
            }
            xb[q][b] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------------------
template <int NIN, int WIDTH, int NHID>
__global__ __launch_bounds__(MLP_THREADS) void k_mlp_fwd(const float* __restrict__ params, int has_bias, int n_out,
                                                         int act, const float* __restrict__ in, int64_t n,
                                                         float* __restrict__ out, int64_t out_stride, int lm,
                                                         const int32_t* __restrict__ n_dev, int n_mul) {
    typedef MlpCfg<NIN, WIDTH, NHID> C;
    constexpr int NQ = C::NQ, PTS = C::PTS;
    __shared__ __attribute__((aligned(16))) float lds[C::L_FWD_END];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, g = lane >> 4;
    const float* bias = lds + C::L_B;
    // n_dev: n_dev[0] * n_mul (<= n) points are present; n stays the plane stride and the bound of every access, the chunk count follows
    // the device-side number (the rows of a last, partial chunk beyond it are computed and never read)
    int64_t n_act = n;
    if (n_dev) { const int64_t m = (int64_t)n_dev[0] * n_mul; n_act = m < n ? m : n; }
    const int64_t n_chunks = (n_act + PTS - 1) / PTS;
    const int64_t stride = (int64_t)gridDim.x * MLP_WAVES;
    int64_t chunk = (int64_t)blockIdx.x * MLP_WAVES + wave;
    // the first chunk's features are requested before the weights: the two latencies overlap; inside the loop the next chunk's
    // loads are in flight while this one goes through the MFMAs (a workgroup runs a few chunks per wave: us_mlp_fwd caps the grid)
    v4f xn[NQ][C::KB_IN];
    if (chunk < n_chunks) load_inputs<NQ, NIN>(in, chunk * PTS, n, row, g, xn, lm);
    load_weights<C, false, MLP_THREADS>(lds, params, has_bias != 0);
    __syncthreads();
    for (; chunk < n_chunks; chunk += stride) {
        const int64_t base = chunk * PTS;
        v4f xb[NQ][C::KB_IN];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int b = 0; b < C::KB_IN; ++b) xb[q][b] = xn[q][b];
        if (chunk + stride < n_chunks) load_inputs<NQ, NIN>(in, (chunk + stride) * PTS, n, row, g, xn, lm);
        v4f h0[NQ][C::MT];
        dense<NQ, C::KB_IN, C::MT, C::S_IN>(lds + C::L_W0, bias, xb, h0, row, g);
        relu_<NQ, C::MT>(h0);
        v4f y[NQ][1];
        if (NHID == 2) {
            v4f h1[NQ][C::MT];
            dense<NQ, C::KB_H, C::MT, C::S_W>(lds + C::L_WH, bias + WIDTH, h0, h1, row, g);
            relu_<NQ, C::MT>(h1);
            dense<NQ, C::KB_H, 1, C::S_W>(lds + C::L_WL, bias + NHID * WIDTH, h1, y, row, g);
        } else {
            dense<NQ, C::KB_H, 1, C::S_W>(lds + C::L_WL, bias + NHID * WIDTH, h0, y, row, g);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int64_t p = base + 16 * q + row;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * g + r;
                if (p < n && o < n_out) out[p * out_stride + o] = act_fwd(y[q][0][r], act);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------------------------
template <int NQ, int MT>
__device__ __forceinline__ void scratch_store(float* buf, const v4f (&t)[NQ][MT], int row, int g) {
    constexpr int SCR_STRIDE = 16 * NQ + 4;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) buf[(16 * m + 4 * g + r) * SCR_STRIDE + 16 * q + row] = t[q][m][r];
}

// acc[mo][mi] += D(rows out) * H(rows in)^T over the 64 points held in the two scratch images
template <int NQ, int MO, int MI>
__device__ __forceinline__ void wgrad(const float* bufD, const float* bufH, v4f (&acc)[MO][MI], int row, int g) {
    constexpr int SCR_STRIDE = 16 * NQ + 4;
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
        v4f a[MO], b[MI];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) a[mo] = *reinterpret_cast<const v4f*>(bufD + (16 * mo + row) * SCR_STRIDE + 16 * qq + 4 * g);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) b[mi] = *reinterpret_cast<const v4f*>(bufH + (16 * mi + row) * SCR_STRIDE + 16 * qq + 4 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mo][mi] = mfma4(a[mo][c], b[mi][c], acc[mo][mi]);
    }
}

template <int NQ, int MT>
__device__ __forceinline__ void bias_acc(v4f (&db)[MT], const v4f (&dh)[NQ][MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q) db[m] += dh[q][m];
}

// Weight-gradient hand-off.  Every wave holds a full set of partial dW tiles.  Letting each wave add them to the global
// gradient made 2048 waves hammer the same ~3 K addresses with float atomics (measured: 190 of 240 us).  Instead the
// waves of a workgroup are summed in LDS (one after the other: no LDS atomics), and the workgroup's vector goes either
// to its row of a partials workspace (plain stores; k_mlp_reduce sums the rows in a fixed order) or, without a
// workspace, to the gradient with ONE atomic per parameter per workgroup.
template <int MO, int MI>
__device__ __forceinline__ void stage_wgrad(float* stage, int poff, int K, const v4f (&acc)[MO][MI], int row, int g, bool add = false) {
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* p = stage + poff + (16 * mo + 4 * g + r) * K + 16 * mi + row;
                *p = add ? *p + acc[mo][mi][r] : acc[mo][mi][r];
            }
}

template <int MT>
__device__ __forceinline__ void stage_bgrad(float* stage, int poff, v4f (&db)[MT], int row, int g, bool add = false) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = db[m][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (row == 0) stage[poff + 16 * m + 4 * g + r] = add ? stage[poff + 16 * m + 4 * g + r] + v : v;
        }
}

// End of the backward kernels: the WAVES waves of a workgroup hold one set of weight-gradient tiles each.  Every wave stores
// its set into its OWN region of LDS (plain, independent stores -- the serial "wave w adds into one image" form spent 24 us
// in dependent LDS read-modify-writes), then all threads sum the regions in wave order (fixed order: reproducible).
template <int WAVES, int NP>
__device__ __forceinline__ void sum_wave_regions(const float* stage, int np, int threads, float* __restrict__ grad_params,
                                                 float* __restrict__ partials) {
    constexpr int NPP = (NP + 3) / 4 * 4;
    for (int k = threadIdx.x; k < np; k += threads) {
        float v = stage[k];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) v += stage[w * NPP + k];
        if (partials) partials[(size_t)blockIdx.x * np + k] = v;
        else if (v != 0.0f) atomicAdd(grad_params + k, v);
    }
}

// sum of the per-workgroup partial rows, fixed order -> bitwise reproducible decoder gradients.
// 64 parameters x 16 row slices per workgroup: 16 independent partial sums per parameter keep the loads short and parallel.
__global__ __launch_bounds__(1024) void k_mlp_reduce(const float* __restrict__ partials, int n_rows, int np, float* __restrict__ grad) {
    __shared__ float sh[16][64];
    const int kl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    float s = 0.0f;
    if (k < np)
        for (int r = sl; r < n_rows; r += 16) s += partials[(size_t)r * np + k];
    sh[sl][kl] = s;
    __syncthreads();
    if (sl == 0 && k < np) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[q][kl];
        grad[k] += t;
    }
}

// ... for the two decoders of a pair launch at once (blockIdx.y = decoder)
__global__ __launch_bounds__(1024) void k_mlp_reduce_pair(const float* __restrict__ pa, const float* __restrict__ pb, int n_rows, int npa, int npb,
                                                          float* __restrict__ ga, float* __restrict__ gb) {
    __shared__ float sh[16][64];
    const float* partials = blockIdx.y ? pb : pa;
    float* grad = blockIdx.y ? gb : ga;
    const int np = blockIdx.y ? npb : npa;
    const int kl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    float s = 0.0f;
    if (k < np)
        for (int r = sl; r < n_rows; r += 16) s += partials[(size_t)r * np + k];
    sh[sl][kl] = s;
    __syncthreads();
    if (sl == 0 && k < np) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[q][kl];
        grad[k] += t;
    }
}

#ifndef MLP_BWD_W32
#define MLP_BWD_W32 8
#endif
#ifndef MLP_BWD_NQ
#define MLP_BWD_NQ 2
#endif
#define MLP_BWD_WAVES(WIDTH) ((WIDTH) == 64 ? 4 : MLP_BWD_W32)

template <int NIN, int WIDTH, int NHID>
__global__ __launch_bounds__(MLP_BWD_WAVES(WIDTH) * 64) void k_mlp_bwd(const float* __restrict__ params, int has_bias, int n_out,
                                                         int act, const float* __restrict__ in,
                                                         const float* __restrict__ out, int64_t out_stride,
                                                         const float* __restrict__ dL_dout, int64_t dout_stride,
                                                         int64_t n, float* __restrict__ dL_din,
                                                         float* __restrict__ grad_params, int lm, float* __restrict__ partials) {
    typedef MlpCfg<NIN, WIDTH, NHID, (WIDTH == 64 ? 2 : MLP_BWD_NQ), MLP_BWD_WAVES(WIDTH)> C;
    static_assert(C::L_TOTAL_BWD <= 40960, "MLP backward exceeds 160 KiB of LDS");
    constexpr int NQ = C::NQ, PTS = C::PTS, WAVES = C::WAVES, THREADS = WAVES * 64;
    __shared__ __attribute__((aligned(16))) float lds[C::L_TOTAL_BWD];
    load_weights<C, true, THREADS>(lds, params, has_bias != 0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, g = lane >> 4;
    const float* bias = lds + C::L_B;
    // per-WAVE scratch: written and read by the same wave only.  LDS operations of one wave execute in order, so no
    // workgroup barrier is needed between a wave's stores and its own loads (the waves run free of each other).
    float* bufA = lds + C::L_SCR + wave * C::SCR_PER_WAVE;         // H_{l-1}^T image  [neuron][PTS points]
    float* bufB = bufA + C::SCR_ROWS_A * C::SCR_STRIDE;            // dH_l^T image

    v4f gW0[C::MT][C::KB_IN], gWH[C::MT][C::MT], gWL[1][C::MT];
    v4f gB0[C::MT], gBH[C::MT], gBL[1];
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < C::MT; ++a) {
#pragma unroll
        for (int b = 0; b < C::KB_IN; ++b) gW0[a][b] = zero;
#pragma unroll
        for (int b = 0; b < C::MT; ++b) gWH[a][b] = zero;
        gWL[0][a] = zero; gB0[a] = zero; gBH[a] = zero;
    }
    gBL[0] = zero;

    const int64_t n_chunks = (n + PTS - 1) / PTS;
    // Two waves per SIMD cannot hide a global round trip: everything a chunk reads (its 32 input features per point, out and
    // dL/dout) is requested one chunk ahead and waited for only when the previous chunk's MFMA work has been issued.
    const int64_t chunk0 = (int64_t)blockIdx.x * WAVES + wave, cstep = (int64_t)gridDim.x * WAVES;
    v4f xn[NQ][C::KB_IN];
    float on[NQ][4], dn[NQ][4];
    auto fetch = [&](int64_t base) {
        load_inputs<NQ, NIN>(in, base, n, row, g, xn, lm);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int64_t p = base + 16 * q + row;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * g + r;
                const bool ok = p < n && o < n_out;
                on[q][r] = ok ? out[p * out_stride + o] : 0.0f;
                dn[q][r] = ok ? dL_dout[p * dout_stride + o] : 0.0f;
            }
        }
    };
    if (chunk0 < n_chunks) fetch(chunk0 * PTS);
    for (int64_t chunk = chunk0; chunk < n_chunks; chunk += cstep) {
        const int64_t base = chunk * PTS;
        v4f xb[NQ][C::KB_IN];
        // dL/d(pre-activation of the output layer), rows >= n_out and points >= n are zero
        v4f dO[NQ][1];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int b = 0; b < C::KB_IN; ++b) xb[q][b] = xn[q][b];
            const int64_t p = base + 16 * q + row;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * g + r;
                dO[q][0][r] = (p < n && o < n_out) ? dn[q][r] * act_bwd(on[q][r], act) : 0.0f;
            }
        }
        if (chunk + cstep < n_chunks) fetch((chunk + cstep) * PTS);
        v4f h0[NQ][C::MT], h1[NQ][C::MT];
        dense<NQ, C::KB_IN, C::MT, C::S_IN>(lds + C::L_W0, bias, xb, h0, row, g);
        relu_<NQ, C::MT>(h0);
        if (NHID == 2) {
            dense<NQ, C::KB_H, C::MT, C::S_W>(lds + C::L_WH, bias + WIDTH, h0, h1, row, g);
            relu_<NQ, C::MT>(h1);
        }
        // ---- output layer:  dWL += dO * Hlast^T ;  dHlast = WL^T dO (.) relu'
        v4f (&hl)[NQ][C::MT] = (NHID == 2) ? h1 : h0;
        if (grad_params) {
            scratch_store<NQ, C::MT>(bufA, hl, row, g);
            scratch_store<NQ, 1>(bufB, dO, row, g);
            __builtin_amdgcn_wave_barrier();
            wgrad<NQ, 1, C::MT>(bufB, bufA, gWL, row, g); bias_acc<NQ, 1>(gBL, dO);
            __builtin_amdgcn_wave_barrier();
        }
        v4f dh[NQ][C::MT];
        dense<NQ, 1, C::MT, C::S_O>(lds + C::L_WLT, nullptr, dO, dh, row, g);
        relu_bwd_<NQ, C::MT>(dh, hl);
        if (NHID == 2) {
            // ---- hidden layer: dWH += dH1 * H0^T ; dH0 = WH^T dH1 (.) relu'
            if (grad_params) {
                scratch_store<NQ, C::MT>(bufA, h0, row, g);
                scratch_store<NQ, C::MT>(bufB, dh, row, g);
                __builtin_amdgcn_wave_barrier();
                wgrad<NQ, C::MT, C::MT>(bufB, bufA, gWH, row, g); bias_acc<NQ, C::MT>(gBH, dh);
                __builtin_amdgcn_wave_barrier();
            }
            v4f dh0[NQ][C::MT];
            dense<NQ, C::KB_H, C::MT, C::S_W>(lds + C::L_WHT, nullptr, dh, dh0, row, g);
            relu_bwd_<NQ, C::MT>(dh0, h0);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int m = 0; m < C::MT; ++m) dh[q][m] = dh0[q][m];
        }
        // ---- input layer: dW0 += dH0 * X^T ; dX = W0^T dH0
        if (grad_params) {
            scratch_store<NQ, C::KB_IN>(bufA, xb, row, g);  // xb tile b, register c is feature 16b + 4g + c: same map
            scratch_store<NQ, C::MT>(bufB, dh, row, g);
            __builtin_amdgcn_wave_barrier();
            wgrad<NQ, C::MT, C::KB_IN>(bufB, bufA, gW0, row, g);
            bias_acc<NQ, C::MT>(gB0, dh);
            __builtin_amdgcn_wave_barrier();
        }
        if (dL_din) {
            v4f dx[NQ][C::KB_IN];
            dense<NQ, C::KB_H, C::KB_IN, C::S_W>(lds + C::L_W0T, nullptr, dh, dx, row, g);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int64_t p = base + 16 * q + row;
                if (p < n) {
#pragma unroll
                    for (int b = 0; b < C::KB_IN; ++b) {
                        if (lm & 1) {
                            *reinterpret_cast<float2*>(dL_din + ((int64_t)(8 * b + 2 * g) * n + p) * 2) = make_float2(dx[q][b][0], dx[q][b][1]);
                            *reinterpret_cast<float2*>(dL_din + ((int64_t)(8 * b + 2 * g + 1) * n + p) * 2) = make_float2(dx[q][b][2], dx[q][b][3]);
                        } else {
                            *reinterpret_cast<v4f*>(dL_din + p * NIN + 16 * b + 4 * g) = dx[q][b];
                        }
                    }
                }
            }
        }
    }
    if (grad_params) {
        constexpr int NP = C::N_W + C::N_B, NPP = (NP + 3) / 4 * 4;
        // at most 8 regions: with more waves, wave w and wave w + 8 share region w & 7 in successive rounds (store, then add)
        constexpr int REG = WAVES < 8 ? WAVES : 8;
        static_assert(REG * NPP <= C::L_TOTAL_BWD && WAVES % REG == 0, "staging area");
        __syncthreads();                                           // weights and scratch images are dead now: the whole LDS is stage
        float* stage = lds + (wave % REG) * NPP;
#pragma unroll
        for (int round = 0; round < WAVES / REG; ++round) {
            if (wave / REG == round) {
                const bool add = round > 0;
                stage_wgrad<C::MT, C::KB_IN>(stage, C::P_W0, NIN, gW0, row, g, add);
                if (NHID == 2) stage_wgrad<C::MT, C::MT>(stage, C::P_WH, WIDTH, gWH, row, g, add);
                stage_wgrad<1, C::MT>(stage, C::P_WL, WIDTH, gWL, row, g, add);
                stage_bgrad<C::MT>(stage, C::P_B0, gB0, row, g, add);
                if (NHID == 2) stage_bgrad<C::MT>(stage, C::P_BH, gBH, row, g, add);
                stage_bgrad<1>(stage, C::P_BL, gBL, row, g, add);
            }
            __syncthreads();
        }
        sum_wave_regions<REG, NP>(lds, has_bias ? NP : C::N_W, THREADS, grad_params, partials);
    }
}

#include "mlp_bf16.inc"
#include "dec_adam_dev.h"

// -------------------------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------------------------
static int check_mlp(const char* fn, const us_mlp_desc* d) {
    US_REQUIRE(d, US_ERR_NULL, "%s: desc is NULL", fn);
    US_REQUIRE(d->n_in == 32, US_ERR_CONFIG, "%s: n_in %u (only 32 = 16 levels x 2 features is built)", fn, d->n_in);
    US_REQUIRE(d->width == 16 || d->width == 32 || d->width == 64, US_ERR_CONFIG, "%s: width %u not in {16,32,64}", fn, d->width);
    US_REQUIRE(d->n_hidden == 1 || d->n_hidden == 2, US_ERR_CONFIG, "%s: n_hidden %u not in {1,2}", fn, d->n_hidden);
    US_REQUIRE(d->n_out >= 1 && d->n_out <= 16, US_ERR_CONFIG, "%s: n_out %u not in 1..16", fn, d->n_out);
    US_REQUIRE(d->out_act <= US_ACT_SIGMOID, US_ERR_CONFIG, "%s: out_act %u", fn, d->out_act);
    US_REQUIRE(d->precision <= US_PREC_F16, US_ERR_CONFIG, "%s: precision %u", fn, d->precision);
    return US_OK;
}

extern "C" size_t us_mlp_n_params(const us_mlp_desc* d) {
    if (!d) return 0;
    size_t nw = (size_t)d->width * d->n_in + (size_t)(d->n_hidden - 1) * d->width * d->width + 16u * d->width;
    if (d->has_bias) nw += (size_t)d->n_hidden * d->width + 16u;
    return nw;
}

#define MLP_DISPATCH(KERNEL, ...)                                                                                  \
    do {                                                                                                           \
        const int key = (int)d->width * 10 + (int)d->n_hidden;                                                     \
        switch (key) {                                                                                             \
            case 161: hipLaunchKernelGGL((KERNEL<32, 16, 1>), grid, block, 0, s, __VA_ARGS__); break;              \
            case 162: hipLaunchKernelGGL((KERNEL<32, 16, 2>), grid, block, 0, s, __VA_ARGS__); break;              \
            case 321: hipLaunchKernelGGL((KERNEL<32, 32, 1>), grid, block, 0, s, __VA_ARGS__); break;              \
            case 322: hipLaunchKernelGGL((KERNEL<32, 32, 2>), grid, block, 0, s, __VA_ARGS__); break;              \
            case 641: hipLaunchKernelGGL((KERNEL<32, 64, 1>), grid, block, 0, s, __VA_ARGS__); break;              \
            default:  hipLaunchKernelGGL((KERNEL<32, 64, 2>), grid, block, 0, s, __VA_ARGS__); break;              \
        }                                                                                                          \
    } while (0)

// the kernels' `lm` word: bit 0 level-major planes, bit 1 inputs already split into hi / lo bf16 pairs (US_MLP_IN_SPLIT_BF16), bit 2 outputs
// before the activation (US_MLP_OUT_PREACT), bit 3 dL_dout w.r.t. those (US_MLP_DOUT_PREACT)
static int mlp_lm_word(const char* fn, int flags, const us_mlp_desc* d, int* lm_out) {
    const int split = (flags & US_MLP_IN_SPLIT_BF16) ? 1 : 0;
    US_REQUIRE(!split || ((flags & US_MLP_LEVEL_MAJOR) && d && d->precision == US_PREC_BF16 && d->n_in == 32), US_ERR_CONFIG,
               "%s: US_MLP_IN_SPLIT_BF16 needs US_MLP_LEVEL_MAJOR and US_PREC_BF16 decoders of 32 inputs", fn);
    const int pre = flags & (US_MLP_OUT_PREACT | US_MLP_DOUT_PREACT);
    US_REQUIRE(!pre || (d && d->precision != US_PREC_F32), US_ERR_CONFIG, "%s: US_MLP_OUT_PREACT / US_MLP_DOUT_PREACT are served by the bf16-family kernels", fn);
    *lm_out = ((flags & US_MLP_LEVEL_MAJOR) ? 1 : 0) | (split ? 2 : 0) | ((flags & US_MLP_OUT_PREACT) ? 4 : 0) | ((flags & US_MLP_DOUT_PREACT) ? 8 : 0);
    return US_OK;
}

// us_mlp_fwd with the point count optionally read on the device (n_dev[0] * n_mul <= n; NULL: n): shared with render.hip
int us_mlp_fwd_counted_rows(const us_mlp_desc* d, const float* params, const float* in, int64_t n, float* out,
                            int64_t out_stride, int flags, const int32_t* n_dev, int n_mul, void* stream) {
    int rc = check_mlp("us_mlp_fwd", d); if (rc) return rc;
    int lm; rc = mlp_lm_word("us_mlp_fwd", flags, d, &lm); if (rc) return rc;
    US_REQUIRE(out_stride >= (int64_t)d->n_out, US_ERR_SHAPE, "us_mlp_fwd: out_stride %lld < n_out", (long long)out_stride);
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(params && in && out, US_ERR_NULL, "us_mlp_fwd: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool bf = d->precision != US_PREC_F32;
    const int pts = bf ? 16 * MLP_BF_FWD_NQ : (d->width == 64 ? 32 : 64);
    // fp32: 2 workgroups per CU, each wave loops with its next chunk prefetched (21.9 vs 23.8 us at 262144 points, MI355X); bf16: mlp_bf16.inc
    const int64_t cap = bf ? MLP_BF_FWD_CAP1 : MLP_FWD_MAX_WG;
    int64_t nb = us_cdiv(n, pts * MLP_WAVES); if (nb > cap) nb = cap;
    dim3 grid((unsigned)nb), block(MLP_THREADS);
    if (d->precision == US_PREC_BF16) MLP_DISPATCH(k_mlp_fwd_bf16x3, params, (int)d->has_bias, (int)d->n_out, (int)d->out_act, in, n, out, out_stride, lm, n_dev, n_mul);
    else if (d->precision == US_PREC_F16) MLP_DISPATCH(k_mlp_fwd_f16, params, (int)d->has_bias, (int)d->n_out, (int)d->out_act, in, n, out, out_stride, lm, n_dev, n_mul);
    else if (bf) MLP_DISPATCH(k_mlp_fwd_bf16, params, (int)d->has_bias, (int)d->n_out, (int)d->out_act, in, n, out, out_stride, lm, n_dev, n_mul);
    else MLP_DISPATCH(k_mlp_fwd, params, (int)d->has_bias, (int)d->n_out, (int)d->out_act, in, n, out, out_stride, lm, n_dev, n_mul);
    US_CHECK_LAUNCH("us_mlp_fwd");
    return US_OK;
}

extern "C" int us_mlp_fwd(const us_mlp_desc* d, const float* params, const float* in, int64_t n, float* out,
                          int64_t out_stride, int flags, void* stream) {
    return us_mlp_fwd_counted_rows(d, params, in, n, out, out_stride, flags, nullptr, 0, stream);
}

#ifndef MLP_BWD_MAX_WG
#define MLP_BWD_MAX_WG 256
#endif

// workgroups (= partial rows of the weight gradient in the workspace) of a one-decoder backward launch: ONE definition for the launch
// (us_mlp_bwd) and for the reduction that reads the rows later (us_mlp_reduce)
static int64_t mlp_bwd_rows(const us_mlp_desc* d, int64_t n, int* waves_out) {
    const bool bf = d->precision != US_PREC_F32;
    const int waves = bf ? MLP_BF_BWD_WAVES(d->width) : MLP_BWD_WAVES(d->width);
    int64_t nb = us_cdiv(n, (bf ? 16 * MLP_BF_BWD_NQ(d->width, d->n_hidden) : (d->width == 64 ? 32 : 16 * MLP_BWD_NQ)) * waves);
    if (nb > MLP_BWD_MAX_WG) nb = MLP_BWD_MAX_WG;               // one workgroup per CU
    if (waves_out) *waves_out = waves;
    return nb;
}

extern "C" size_t us_mlp_bwd_workspace_bytes(const us_mlp_desc* d) {
    return d ? (size_t)MLP_BWD_MAX_WG * us_mlp_n_params(d) * sizeof(float) : 0;
}

extern "C" int us_mlp_bwd(const us_mlp_desc* d, const float* params, const float* in, const float* out,
                          int64_t out_stride, const float* dL_dout, int64_t dout_stride, int64_t n, float* dL_din,
                          float* grad_params, int flags, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_mlp("us_mlp_bwd", d); if (rc) return rc;
    int lm; rc = mlp_lm_word("us_mlp_bwd", flags, d, &lm); if (rc) return rc;
    US_REQUIRE(out_stride >= (int64_t)d->n_out && dout_stride >= (int64_t)d->n_out, US_ERR_SHAPE, "us_mlp_bwd: stride < n_out");
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(params && in && out && dL_dout, US_ERR_NULL, "us_mlp_bwd: NULL pointer");
    if (!dL_din && !grad_params) return US_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool bf = d->precision != US_PREC_F32;
    int waves;
    const int64_t nb = mlp_bwd_rows(d, n, &waves);
    dim3 grid((unsigned)nb), block(waves * 64);
    float* partials = nullptr;
    if (grad_params && workspace) {
        US_REQUIRE(workspace_bytes >= us_mlp_bwd_workspace_bytes(d), US_ERR_WORKSPACE, "us_mlp_bwd: workspace %zu B < %zu B",
                   workspace_bytes, us_mlp_bwd_workspace_bytes(d));
        partials = (float*)workspace;
    }
    if (d->precision == US_PREC_BF16) MLP_DISPATCH(k_mlp_bwd_bf16x3, params, (int)d->has_bias, (int)d->n_out, (int)d->out_act, in, out, out_stride, dL_dout,
                                                   dout_stride, n, dL_din, grad_params, lm, partials);
    else if (d->precision == US_PREC_F16) MLP_DISPATCH(k_mlp_bwd_f16, params, (int)d->has_bias, (int)d->n_out, (int)d->out_act, in, out, out_stride, dL_dout,
                                                       dout_stride, n, dL_din, grad_params, lm, partials);
    else if (bf) MLP_DISPATCH(k_mlp_bwd_bf16, params, (int)d->has_bias, (int)d->n_out, (int)d->out_act, in, out, out_stride, dL_dout,
                              dout_stride, n, dL_din, grad_params, lm, partials);
    else MLP_DISPATCH(k_mlp_bwd, params, (int)d->has_bias, (int)d->n_out, (int)d->out_act, in, out, out_stride, dL_dout,
                      dout_stride, n, dL_din, grad_params, lm, partials);
    US_CHECK_LAUNCH("us_mlp_bwd");
    if (partials && !(flags & US_MLP_DEFER_REDUCE)) {
        const int np = (int)us_mlp_n_params(d);
        hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)us_cdiv(np, 64)), dim3(1024), 0, s, partials, (int)nb, np, grad_params);
        US_CHECK_LAUNCH("us_mlp_bwd(reduce)");
    }
    return US_OK;
}

// the reduction us_mlp_bwd(..., US_MLP_DEFER_REDUCE, workspace) left out: grad_params += sum of the workspace's partial rows (fixed order)
extern "C" int us_mlp_reduce(const us_mlp_desc* d, const void* workspace, size_t workspace_bytes, int64_t n, float* grad_params, void* stream) {
    int rc = check_mlp("us_mlp_reduce", d); if (rc) return rc;
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(workspace && grad_params, US_ERR_NULL, "us_mlp_reduce: NULL pointer");
    US_REQUIRE(workspace_bytes >= us_mlp_bwd_workspace_bytes(d), US_ERR_WORKSPACE, "us_mlp_reduce: workspace %zu B < %zu B", workspace_bytes,
               us_mlp_bwd_workspace_bytes(d));
    const int64_t nb = mlp_bwd_rows(d, n, nullptr);
    const int np = (int)us_mlp_n_params(d);
    hipLaunchKernelGGL(k_mlp_reduce, dim3((unsigned)us_cdiv(np, 64)), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, (int)nb, np, grad_params);
    US_CHECK_LAUNCH("us_mlp_reduce");
    return US_OK;
}

// ---- two decoders of equal shape in one launch each way (k_mlp_fwd_pair / k_mlp_bwd_pair in mlp_bf16.inc)
static bool mlp_pair_ok(const us_mlp_desc* a, const us_mlp_desc* b) {
    return a && b && a->n_in == 32 && b->n_in == 32 && a->width == b->width && a->n_hidden == b->n_hidden && a->precision == b->precision &&
           (a->precision == US_PREC_BF16 || a->precision == US_PREC_BF16_PLAIN || a->precision == US_PREC_F16) && (a->width == 16 || a->width == 32 || a->width == 64) &&
           (a->n_hidden == 1 || a->n_hidden == 2) && a->n_out >= 1 && a->n_out <= 16 && b->n_out >= 1 && b->n_out <= 16 &&
           a->out_act <= US_ACT_SIGMOID && b->out_act <= US_ACT_SIGMOID;
}
extern "C" int us_mlp_pair_supported(const us_mlp_desc* a, const us_mlp_desc* b) { return mlp_pair_ok(a, b) ? 1 : 0; }

// workgroups (= partial rows of the weight gradient) PER DECODER of the pair's backward launch: both decoders together put one
// workgroup on every CU (MI355X, 2 x 32 decoders, 262 144 points: 48.0 us; 2 x 192: 61.0; 2 x 256: 55.8; 2 x 384: 64.6)
#ifndef MLP_BWD_PAIR_WG
#define MLP_BWD_PAIR_WG 128
#endif
#ifndef MLP_BWD_IN_WG
#define MLP_BWD_IN_WG 128                // workgroups per decoder of the input-gradient-only launch (mlp_bf16.inc has the sweep)
#endif
static_assert(MLP_BWD_PAIR_WG <= MLP_BWD_MAX_WG, "the pair launch's partial rows must fit the workspace us_mlp_bwd_workspace_bytes sizes");
static int64_t mlp_pair_rows(const us_mlp_desc* d, int64_t n) {
    const int waves = MLP_BF_BWD_WAVES(d->width);
    int64_t nb = us_cdiv(n, 16 * MLP_BF_BWD_NQ(d->width, d->n_hidden) * waves);
    return nb > MLP_BWD_PAIR_WG ? MLP_BWD_PAIR_WG : nb;
}

#define MLP_PAIR_CASES(KERNEL, MODE, ...)                                                                          \
    switch (key) {                                                                                                 \
        case 161: hipLaunchKernelGGL((KERNEL<32, 16, 1, MODE>), grid, block, 0, s, __VA_ARGS__); break;            \
        case 162: hipLaunchKernelGGL((KERNEL<32, 16, 2, MODE>), grid, block, 0, s, __VA_ARGS__); break;            \
        case 321: hipLaunchKernelGGL((KERNEL<32, 32, 1, MODE>), grid, block, 0, s, __VA_ARGS__); break;            \
        case 322: hipLaunchKernelGGL((KERNEL<32, 32, 2, MODE>), grid, block, 0, s, __VA_ARGS__); break;            \
        case 641: hipLaunchKernelGGL((KERNEL<32, 64, 1, MODE>), grid, block, 0, s, __VA_ARGS__); break;            \
        default:  hipLaunchKernelGGL((KERNEL<32, 64, 2, MODE>), grid, block, 0, s, __VA_ARGS__); break;            \
    }
// MODE (mlp_bf16.inc): 1 = bf16 with split operands, 2 = f16, 0 = bf16 with one product
#define MLP_PAIR_DISPATCH(KERNEL, ...)                                                                             \
    do {                                                                                                           \
        const int key = (int)da->width * 10 + (int)da->n_hidden;                                                   \
        if (da->precision == US_PREC_BF16) { MLP_PAIR_CASES(KERNEL, 1, __VA_ARGS__) }                              \
        else if (da->precision == US_PREC_F16) { MLP_PAIR_CASES(KERNEL, 2, __VA_ARGS__) }                          \
        else { MLP_PAIR_CASES(KERNEL, 0, __VA_ARGS__) }                                                            \
    } while (0)

extern "C" int us_mlp_fwd_pair(const us_mlp_desc* da, const us_mlp_desc* db, const float* params_a, const float* params_b, const float* in_a,
                               const float* in_b, int64_t n, float* out_a, int64_t out_stride_a, float* out_b, int64_t out_stride_b, int flags,
                               void* stream) {
    US_REQUIRE(mlp_pair_ok(da, db), US_ERR_CONFIG, "us_mlp_fwd_pair: needs two bf16 decoders (32 inputs) of equal width, depth and precision");
    US_REQUIRE(out_stride_a >= (int64_t)da->n_out && out_stride_b >= (int64_t)db->n_out, US_ERR_SHAPE, "us_mlp_fwd_pair: out_stride < n_out");
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(params_a && params_b && in_a && in_b && out_a && out_b, US_ERR_NULL, "us_mlp_fwd_pair: NULL pointer");
    int lm; { const int rc = mlp_lm_word("us_mlp_fwd_pair", flags, da, &lm); if (rc) return rc; }
    hipStream_t s = (hipStream_t)stream;
    int64_t nb = us_cdiv(n, 16 * MLP_BF_FWD_NQ * MLP_WAVES); if (nb > MLP_BF_FWD_CAP) nb = MLP_BF_FWD_CAP;           // as us_mlp_fwd (bf16)
    dim3 grid((unsigned)nb, 2), block(MLP_THREADS);
    MlpFwdJob a = {params_a, (int)da->has_bias, (int)da->n_out, (int)da->out_act, in_a, out_a, (long long)out_stride_a};
    MlpFwdJob b = {params_b, (int)db->has_bias, (int)db->n_out, (int)db->out_act, in_b, out_b, (long long)out_stride_b};
    MLP_PAIR_DISPATCH(k_mlp_fwd_pair, a, b, n, lm);
    US_CHECK_LAUNCH("us_mlp_fwd_pair");
    return US_OK;
}

static int mlp_bwd_pair(const us_mlp_desc* da, const us_mlp_desc* db, const float* params_a, const float* params_b, const float* in_a,
                        const float* in_b, const float* out_a, int64_t out_stride_a, const float* out_b, int64_t out_stride_b,
                        const float* dL_dout_a, int64_t dout_stride_a, const float* dL_dout_b, int64_t dout_stride_b, int64_t n,
                        float* dL_din_a, float* dL_din_b, float* grad_params_a, float* grad_params_b, int flags, void* workspace_a,
                        void* workspace_b, size_t workspace_bytes, const us_half_t* dy_dx_a, const us_half_t* dy_dx_b, float* dpts_a, float* dpts_b,
                        void* stream) {
    US_REQUIRE(mlp_pair_ok(da, db), US_ERR_CONFIG, "us_mlp_bwd_pair: needs two bf16 decoders (32 inputs) of equal width, depth and precision");
    US_REQUIRE(out_stride_a >= (int64_t)da->n_out && dout_stride_a >= (int64_t)da->n_out && out_stride_b >= (int64_t)db->n_out &&
               dout_stride_b >= (int64_t)db->n_out, US_ERR_SHAPE, "us_mlp_bwd_pair: stride < n_out");
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(params_a && params_b && in_a && in_b && out_a && out_b && dL_dout_a && dL_dout_b, US_ERR_NULL, "us_mlp_bwd_pair: NULL pointer");
    const bool wgrad = grad_params_a || grad_params_b;           // tracking asks for the input gradients only
    if (wgrad) {
        US_REQUIRE(grad_params_a && grad_params_b && workspace_a && workspace_b && workspace_a != workspace_b, US_ERR_NULL,
                   "us_mlp_bwd_pair: parameter gradients are formed for both decoders or for neither, with one workspace per decoder");
        US_REQUIRE(workspace_bytes >= us_mlp_bwd_workspace_bytes(da) && workspace_bytes >= us_mlp_bwd_workspace_bytes(db), US_ERR_WORKSPACE,
                   "us_mlp_bwd_pair: workspace %zu B too small", workspace_bytes);
    } else {
        US_REQUIRE((dL_din_a && dL_din_b) || dy_dx_a, US_ERR_NULL, "us_mlp_bwd_pair: nothing to compute (no parameter gradients, no input gradients)");
        workspace_a = workspace_b = nullptr;
    }
    US_REQUIRE((dL_din_a != nullptr) == (dL_din_b != nullptr), US_ERR_NULL, "us_mlp_bwd_pair: input gradients of both decoders or of neither");
    int lm; { const int rc = mlp_lm_word("us_mlp_bwd_pair", flags, da, &lm); if (rc) return rc; }
    if (dy_dx_a || dy_dx_b || dpts_a || dpts_b) {
        US_REQUIRE(dy_dx_a && dy_dx_b && dpts_a && dpts_b && dpts_a != dpts_b, US_ERR_NULL, "us_mlp_bwd_pair_dydx: dy_dx and dL_dpts of both decoders");
        US_REQUIRE(lm && da->n_in == 32, US_ERR_CONFIG, "us_mlp_bwd_pair_dydx: level-major inputs of 16 levels x 2 features");
        US_REQUIRE(((((uintptr_t)dy_dx_a) | ((uintptr_t)dy_dx_b)) & 3u) == 0, US_ERR_SHAPE, "us_mlp_bwd_pair_dydx: dy_dx must be 4-byte aligned");
    }
    hipStream_t s = (hipStream_t)stream;
    const int waves = MLP_BF_BWD_WAVES(da->width);
    const int64_t nb = mlp_pair_rows(da, n);
    dim3 grid((unsigned)nb, 2), block(waves * 64);
    MlpBwdJob a = {params_a, (int)da->has_bias, (int)da->n_out, (int)da->out_act, in_a, out_a, (long long)out_stride_a, dL_dout_a,
                   (long long)dout_stride_a, dL_din_a, grad_params_a, (float*)workspace_a, dy_dx_a, dpts_a};
    MlpBwdJob b = {params_b, (int)db->has_bias, (int)db->n_out, (int)db->out_act, in_b, out_b, (long long)out_stride_b, dL_dout_b,
                   (long long)dout_stride_b, dL_din_b, grad_params_b, (float*)workspace_b, dy_dx_b, dpts_b};
    if (wgrad) {
        MLP_PAIR_DISPATCH(k_mlp_bwd_pair, a, b, n, lm);
    } else {                                                     // input gradients only: the lean kernel, one workgroup per 8 chunks or so
        int64_t rows = us_cdiv(n, 16 * MLP_BF_BWD_IN_NQ * MLP_BF_BWD_IN_WAVES);
        if (rows > MLP_BWD_IN_WG) rows = MLP_BWD_IN_WG;
        grid = dim3((unsigned)rows, 2); block = dim3(MLP_BF_BWD_IN_WAVES * 64);
        MLP_PAIR_DISPATCH(k_mlp_bwd_pair_in, a, b, n, lm);
    }
    US_CHECK_LAUNCH("us_mlp_bwd_pair");
    if (wgrad && !(flags & US_MLP_DEFER_REDUCE)) {
        const int npa = (int)us_mlp_n_params(da), npb = (int)us_mlp_n_params(db);
        hipLaunchKernelGGL(k_mlp_reduce_pair, dim3((unsigned)us_cdiv(npa > npb ? npa : npb, 64), 2), dim3(1024), 0, s, (const float*)workspace_a,
                           (const float*)workspace_b, (int)nb, npa, npb, grad_params_a, grad_params_b);
        US_CHECK_LAUNCH("us_mlp_bwd_pair(reduce)");
    }
    return US_OK;
}

extern "C" int us_mlp_bwd_pair(const us_mlp_desc* da, const us_mlp_desc* db, const float* params_a, const float* params_b, const float* in_a,
                               const float* in_b, const float* out_a, int64_t out_stride_a, const float* out_b, int64_t out_stride_b,
                               const float* dL_dout_a, int64_t dout_stride_a, const float* dL_dout_b, int64_t dout_stride_b, int64_t n,
                               float* dL_din_a, float* dL_din_b, float* grad_params_a, float* grad_params_b, int flags, void* workspace_a,
                               void* workspace_b, size_t workspace_bytes, void* stream) {
    return mlp_bwd_pair(da, db, params_a, params_b, in_a, in_b, out_a, out_stride_a, out_b, out_stride_b, dL_dout_a, dout_stride_a, dL_dout_b,
                        dout_stride_b, n, dL_din_a, dL_din_b, grad_params_a, grad_params_b, flags, workspace_a, workspace_b, workspace_bytes,
                        nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int us_mlp_bwd_pair_dydx(const us_mlp_desc* da, const us_mlp_desc* db, const float* params_a, const float* params_b, const float* in_a,
                                    const float* in_b, const float* out_a, int64_t out_stride_a, const float* out_b, int64_t out_stride_b,
                                    const float* dL_dout_a, int64_t dout_stride_a, const float* dL_dout_b, int64_t dout_stride_b, int64_t n,
                                    float* dL_din_a, float* dL_din_b, float* grad_params_a, float* grad_params_b, int flags, void* workspace_a,
                                    void* workspace_b, size_t workspace_bytes, const us_half_t* dy_dx_a, const us_half_t* dy_dx_b, float* dL_dpts_a,
                                    float* dL_dpts_b, void* stream) {
    US_REQUIRE(n <= 0 || (dy_dx_a && dy_dx_b && dL_dpts_a && dL_dpts_b), US_ERR_NULL, "us_mlp_bwd_pair_dydx: NULL pointer");
    return mlp_bwd_pair(da, db, params_a, params_b, in_a, in_b, out_a, out_stride_a, out_b, out_stride_b, dL_dout_a, dout_stride_a, dL_dout_b,
                        dout_stride_b, n, dL_din_a, dL_din_b, grad_params_a, grad_params_b, flags, workspace_a, workspace_b, workspace_bytes,
                        dy_dx_a, dy_dx_b, dL_dpts_a, dL_dpts_b, stream);
}

// the reductions us_mlp_bwd_pair(..., US_MLP_DEFER_REDUCE) left out, both decoders in one launch (fixed order)
extern "C" int us_mlp_reduce_pair(const us_mlp_desc* da, const us_mlp_desc* db, const void* workspace_a, const void* workspace_b, size_t workspace_bytes,
                                  int64_t n, float* grad_params_a, float* grad_params_b, void* stream) {
    US_REQUIRE(mlp_pair_ok(da, db), US_ERR_CONFIG, "us_mlp_reduce_pair: needs two bf16 decoders (32 inputs) of equal width, depth and precision");
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(workspace_a && workspace_b && grad_params_a && grad_params_b, US_ERR_NULL, "us_mlp_reduce_pair: NULL pointer");
    US_REQUIRE(workspace_bytes >= us_mlp_bwd_workspace_bytes(da) && workspace_bytes >= us_mlp_bwd_workspace_bytes(db), US_ERR_WORKSPACE,
               "us_mlp_reduce_pair: workspace %zu B too small", workspace_bytes);
    const int npa = (int)us_mlp_n_params(da), npb = (int)us_mlp_n_params(db);
    hipLaunchKernelGGL(k_mlp_reduce_pair, dim3((unsigned)us_cdiv(npa > npb ? npa : npb, 64), 2), dim3(1024), 0, (hipStream_t)stream,
                       (const float*)workspace_a, (const float*)workspace_b, (int)mlp_pair_rows(da, n), npa, npb, grad_params_a, grad_params_b);
    US_CHECK_LAUNCH("us_mlp_reduce_pair");
    return US_OK;
}

// The same reductions with the optimiser step of the decoder param group folded in (single process; the data-parallel step needs the
// gradients themselves for its all-reduce): gradient = fixed-order sum of the partial rows (written to grad_params: no cleared buffer
// needed), beta's gradient = f64 sum of the per-ray partials, then torch.optim.Adam on exactly those parameters -- one launch on the
// MAIN stream in front of the tables' Adam pass instead of a fill, two reductions and a fork / join around the table gradient
// (src/Mapper.py:443-445: zero_grad, backward, step for the param group of :118).
__global__ __launch_bounds__(1024) void k_mlp_reduce_pair_adam(const float* __restrict__ pa, const float* __restrict__ pb, int n_rows, int npa, int npb,
                                                               float* __restrict__ ga, float* __restrict__ gb, float* __restrict__ Pa,
                                                               float* __restrict__ Pb, float* __restrict__ ma, float* __restrict__ mb,
                                                               float* __restrict__ va, float* __restrict__ vb, const float* __restrict__ beta_part,
                                                               int64_t n_rays, float* __restrict__ p_beta,
                                                               float* __restrict__ g_beta, float* __restrict__ m_beta, float* __restrict__ v_beta,
                                                               DecAdam ad) {
    if (blockIdx.y == 2) {                                       // beta: one workgroup
        if (blockIdx.x != 0 || !beta_part) return;
        __shared__ double shd[1024];
        double acc = 0.0;
        for (int64_t r = threadIdx.x; r < n_rays; r += 1024) acc += (double)beta_part[r];
        shd[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) { const float g = (float)shd[0]; *g_beta = g; dec_adam_apply(g, p_beta, m_beta, v_beta, ad); }
        return;
    }
    __shared__ float sh[16][64];
    const float* partials = blockIdx.y ? pb : pa;
    const int np = blockIdx.y ? npb : npa;
    const int kl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kl;
    float s = 0.0f;
    if (k < np)
        for (int r = sl; r < n_rows; r += 16) s += partials[(size_t)r * np + k];
    sh[sl][kl] = s;
    __syncthreads();
    if (sl == 0 && k < np) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sh[q][kl];
        (blockIdx.y ? gb : ga)[k] = t;
        dec_adam_apply(t, (blockIdx.y ? Pb : Pa) + k, (blockIdx.y ? mb : ma) + k, (blockIdx.y ? vb : va) + k, ad);
    }
}

extern "C" int us_mlp_reduce_pair_adam(const us_mlp_desc* da, const us_mlp_desc* db, const void* workspace_a, const void* workspace_b,
                                       size_t workspace_bytes, int64_t n, float* params_a, float* params_b, float* grad_params_a,
                                       float* grad_params_b, float* m_a, float* m_b, float* v_a, float* v_b, const float* beta_partials,
                                       int64_t n_rays, float* beta, float* grad_beta, float* m_beta, float* v_beta, double lr, double beta1,
                                       double beta2, double eps, const float* step_dev, void* stream) {
    US_REQUIRE(mlp_pair_ok(da, db), US_ERR_CONFIG, "us_mlp_reduce_pair_adam: needs two bf16 decoders (32 inputs) of equal width, depth and precision");
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(workspace_a && workspace_b && params_a && params_b && grad_params_a && grad_params_b && m_a && m_b && v_a && v_b && step_dev, US_ERR_NULL,
               "us_mlp_reduce_pair_adam: NULL pointer");
    US_REQUIRE(!beta_partials || (beta && grad_beta && m_beta && v_beta && n_rays >= 1), US_ERR_NULL, "us_mlp_reduce_pair_adam: beta needs p, g, m, v and n_rays");
    US_REQUIRE(((uintptr_t)step_dev & 7u) == 0, US_ERR_SHAPE, "us_mlp_reduce_pair_adam: step_dev (float[8]) must be 8-byte aligned");
    US_REQUIRE(workspace_bytes >= us_mlp_bwd_workspace_bytes(da) && workspace_bytes >= us_mlp_bwd_workspace_bytes(db), US_ERR_WORKSPACE,
               "us_mlp_reduce_pair_adam: workspace %zu B too small", workspace_bytes);
    const int npa = (int)us_mlp_n_params(da), npb = (int)us_mlp_n_params(db);
    DecAdam ad; ad.lr = (float)lr; ad.one_minus_b1 = (float)(1.0 - beta1); ad.b2 = (float)beta2; ad.one_minus_b2 = (float)(1.0 - beta2); ad.eps = (float)eps;
    ad.step_dev = step_dev;
    hipLaunchKernelGGL(k_mlp_reduce_pair_adam, dim3((unsigned)us_cdiv(npa > npb ? npa : npb, 64), beta_partials ? 3 : 2), dim3(1024), 0,
                       (hipStream_t)stream, (const float*)workspace_a, (const float*)workspace_b, (int)mlp_pair_rows(da, n), npa, npb, grad_params_a,
                       grad_params_b, params_a, params_b, m_a, m_b, v_a, v_b, beta_partials, n_rays, beta, grad_beta, m_beta, v_beta, ad);
    US_CHECK_LAUNCH("us_mlp_reduce_pair_adam");
    return US_OK;
}

// The same, and the optimiser pass of the tables, in ONE launch: the decoder group's reductions are three extra slices of the tables' Adam
// launch (render.hip: k_adam_segs_model) -- 50 small workgroups beside thousands, instead of a 7.5 us launch in front of them.
extern "C" int us_adam_step_model(const us_mlp_desc* da, const us_mlp_desc* db, const void* workspace_a, const void* workspace_b,
                                  size_t workspace_bytes, int64_t n, float* params_a, float* params_b, float* grad_params_a,
                                  float* grad_params_b, float* m_a, float* m_b, float* v_a, float* v_b, const float* beta_partials,
                                  int64_t n_rays, float* beta, float* grad_beta, float* m_beta, float* v_beta, double lr_decoders,
                                  float* p, float* g, float* m, float* v, int n_seg, const int64_t* seg_off, const int64_t* seg_n,
                                  const double* seg_lr, double beta1, double beta2, double eps, float* step_dev, unsigned zero_grad_mask,
                                  const us_pose_step_desc* poses, void* stream) {
    US_REQUIRE(mlp_pair_ok(da, db), US_ERR_CONFIG, "us_adam_step_model: needs two bf16 decoders (32 inputs) of equal width, depth and precision");
    if (n <= 0) return n == 0 ? US_OK : US_ERR_SHAPE;
    US_REQUIRE(workspace_a && workspace_b && params_a && params_b && grad_params_a && grad_params_b && m_a && m_b && v_a && v_b && step_dev, US_ERR_NULL,
               "us_adam_step_model: NULL pointer");
    US_REQUIRE(!beta_partials || (beta && grad_beta && m_beta && v_beta && n_rays >= 1), US_ERR_NULL, "us_adam_step_model: beta needs p, g, m, v and n_rays");
    US_REQUIRE(((uintptr_t)step_dev & 7u) == 0, US_ERR_SHAPE, "us_adam_step_model: step_dev (float[8]) must be 8-byte aligned");
    US_REQUIRE(workspace_bytes >= us_mlp_bwd_workspace_bytes(da) && workspace_bytes >= us_mlp_bwd_workspace_bytes(db), US_ERR_WORKSPACE,
               "us_adam_step_model: workspace %zu B too small", workspace_bytes);
    DecGroup dg;
    dg.pa = (const float*)workspace_a; dg.pb = (const float*)workspace_b; dg.n_rows = (int)mlp_pair_rows(da, n);
    dg.npa = (int)us_mlp_n_params(da); dg.npb = (int)us_mlp_n_params(db);
    dg.ga = grad_params_a; dg.gb = grad_params_b; dg.Pa = params_a; dg.Pb = params_b; dg.ma = m_a; dg.mb = m_b; dg.va = v_a; dg.vb = v_b;
    dg.beta_part = beta_partials; dg.n_rays = n_rays; dg.p_beta = beta; dg.g_beta = grad_beta; dg.m_beta = m_beta; dg.v_beta = v_beta;
    dg.ad.lr = (float)lr_decoders; dg.ad.one_minus_b1 = (float)(1.0 - beta1); dg.ad.b2 = (float)beta2; dg.ad.one_minus_b2 = (float)(1.0 - beta2);
    dg.ad.eps = (float)eps; dg.ad.step_dev = step_dev;
    return us_adam_segments_model(p, g, m, v, n_seg, seg_off, seg_n, seg_lr, beta1, beta2, eps, step_dev, zero_grad_mask, stream, dg, poses);
}

#include <string.h>
#ifdef US_EXPERIMENTS                    // measured-slower variants, kept buildable: tools/build_experiments.sh (include/unislam_hip_experiments.h)
#include "encode_decode.inc"
#endif
