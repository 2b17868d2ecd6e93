// tools/lds_atomic_bench.hip -- microbenchmark: cost of LDS float atomics on gfx950 (development aid, not product code)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 2048
#define LDS_FLOATS 32768

__device__ __forceinline__ uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// MODE 0: ds_add_f32 all lanes random | 1: ds_add_f32, 2 of 64 lanes active | 2: ds_add_u32 all lanes | 3: plain ds_write
// 8: ds_add_u64 random | 9: ds_add_u64 same address | 10: ds_add_u32 same address | 11: ds_add_f64 random | 12: u64 run-of-8 conflicts | 13: f32 run-of-8
// 4: read+add+write (non atomic) | 5: ds_add_f32 all lanes SAME address | 6: ds_add_rtn_f32 all lanes random | 7: ds_add_f32 4 lanes
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int active_div) {
    __shared__ float lds[LDS_FLOATS];
    for (int i = threadIdx.x; i < LDS_FLOATS; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    uint32_t s = threadIdx.x * 9781u + blockIdx.x * 6271u + 1u;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int it = 0; it < ITERS; ++it) {
        const uint32_t idx = rng(s) & (LDS_FLOATS - 1);
        const float v = (float)(idx & 7);
        if (MODE == 0) atomicAdd(&lds[idx], v);
        if (MODE == 1) { if ((lane % 32) == 0) atomicAdd(&lds[idx], v); }
        if (MODE == 2) atomicAdd((unsigned*)&lds[idx], idx);
        if (MODE == 3) lds[idx] = v;
        if (MODE == 4) lds[idx] += v;
        if (MODE == 5) atomicAdd(&lds[it & 1023], v);
        if (MODE == 6) acc += atomicAdd(&lds[idx], v);
        if (MODE == 7) { if ((lane % 16) == 0) atomicAdd(&lds[idx], v); }
        if (MODE == 8) atomicAdd((unsigned long long*)&lds[(idx & (LDS_FLOATS / 2 - 1)) * 2], (unsigned long long)idx);
        if (MODE == 9) atomicAdd((unsigned long long*)&lds[(it & 511) * 2], (unsigned long long)idx);
        if (MODE == 10) atomicAdd((unsigned*)&lds[it & 1023], idx);
        if (MODE == 11) atomicAdd((double*)&lds[(idx & (LDS_FLOATS / 2 - 1)) * 2], (double)v);
        if (MODE == 12) { const uint32_t j = __shfl(idx, lane & ~7); atomicAdd((unsigned long long*)&lds[(j & (LDS_FLOATS / 2 - 1)) * 2], (unsigned long long)idx); }
        if (MODE == 13) { const uint32_t j = __shfl(idx, lane & ~7); atomicAdd(&lds[j], v); }
    }
    __syncthreads();
    float t = acc;
    for (int i = threadIdx.x; i < LDS_FLOATS; i += blockDim.x) t += lds[i];
    if (t == 123.456f) out[0] = t;
}

template <int MODE> void run(const char* name, int threads) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, threads>>>(out, 1); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k<MODE><<<256, threads>>>(out, 1);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double waves = threads / 64.0;
    const double instr_per_cu = waves * ITERS;
    printf("%-44s threads %4d: %8.3f ms  -> %7.1f ns per wave-instruction per CU (%.0f cycles @2.4GHz)\n", name, threads, ms,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
    hipFree(out);
}

int main() {
    for (int threads : {256, 1024}) {
        if (threads == 256) {
            run<0>("ds_add_f32 64 lanes random", 256); run<1>("ds_add_f32 2 lanes active", 256); run<7>("ds_add_f32 4 lanes active", 256);
            run<2>("ds_add_u32 64 lanes random", 256); run<3>("ds_write_b32 random", 256); run<4>("read+add+write non-atomic", 256);
            run<5>("ds_add_f32 64 lanes same address", 256); run<6>("ds_add_rtn_f32 random", 256);
        } else {
            run<0>("ds_add_f32 64 lanes random", 1024); run<1>("ds_add_f32 2 lanes active", 1024); run<7>("ds_add_f32 4 lanes active", 1024);
            run<2>("ds_add_u32 64 lanes random", 1024); run<3>("ds_write_b32 random", 1024); run<4>("read+add+write non-atomic", 1024);
            run<5>("ds_add_f32 64 lanes same address", 1024); run<6>("ds_add_rtn_f32 random", 1024);
            run<8>("ds_add_u64 64 lanes random", 1024); run<9>("ds_add_u64 64 lanes same address", 1024);
            run<10>("ds_add_u32 64 lanes same address", 1024); run<11>("ds_add_f64 64 lanes random", 1024);
            run<12>("ds_add_u64 runs of 8 equal addresses", 1024); run<13>("ds_add_f32 runs of 8 equal addresses", 1024);
        }
    }
    return 0;
}
