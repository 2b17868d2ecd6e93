// tools/stream_bench.hip -- how fast can 256-thread workgroups stream per-bin record ranges? (development microbenchmark)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_bench.hip -o /tmp/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: dwordx4 coalesced, MODE 1: 12-byte records as 3 dword loads, MODE 2: records as dwordx3
template <int MODE, int UNROLL, int LDS_KB, bool ATOMIC, int T>
__global__ __launch_bounds__(T) void k_stream(const uint32_t* __restrict__ rec, const uint32_t* __restrict__ offsets, float* __restrict__ out) {
    __shared__ double acc[LDS_KB * 128];
    const uint32_t b = blockIdx.x;
    const uint32_t r0 = offsets[b], r1 = offsets[b + 1];
    for (uint32_t k = threadIdx.x; k < LDS_KB * 128; k += T) acc[k] = 0.0;
    __syncthreads();
    float s = 0.f;
    if (MODE == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(rec);
        const uint32_t q0 = (r0 * 3u + 3u) / 4u, q1 = (r1 * 3u) / 4u;
        for (uint32_t base = q0; base < q1; base += T * UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { const uint32_t q = base + u * T + threadIdx.x; v[u] = q < q1 ? p[q] : make_uint4(0, 0, 0, 0); }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (ATOMIC) { atomicAdd(&acc[(v[u].x & (LDS_KB * 64 - 1)) * 2], (double)__uint_as_float(v[u].y)); atomicAdd(&acc[(v[u].x & (LDS_KB * 64 - 1)) * 2 + 1], (double)__uint_as_float(v[u].z)); }
                else s += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
            }
        }
    } else {
        for (uint32_t base = r0; base < r1; base += T * UNROLL) {
            uint32_t loc[UNROLL]; float a[UNROLL], c[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint32_t r = base + u * T + threadIdx.x;
                loc[u] = 0xFFFFFFFFu; a[u] = 0; c[u] = 0;
                if (r < r1) {
                    if (MODE == 1) { const uint32_t* src = rec + (size_t)r * 3; loc[u] = src[0]; a[u] = __uint_as_float(src[1]); c[u] = __uint_as_float(src[2]); }
                    else { struct __attribute__((packed, aligned(4))) R3 { uint32_t w[3]; }; const R3 t = *reinterpret_cast<const R3*>(rec + (size_t)r * 3); loc[u] = t.w[0]; a[u] = __uint_as_float(t.w[1]); c[u] = __uint_as_float(t.w[2]); }
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (loc[u] != 0xFFFFFFFFu) {
                    if (ATOMIC) { atomicAdd(&acc[(loc[u] & (LDS_KB * 64 - 1)) * 2], (double)a[u]); atomicAdd(&acc[(loc[u] & (LDS_KB * 64 - 1)) * 2 + 1], (double)c[u]); }
                    else s += a[u] + c[u] + __uint_as_float(loc[u]);
                }
            }
        }
    }
    __syncthreads();
    if (ATOMIC) { for (uint32_t k = threadIdx.x; k < LDS_KB * 128; k += T) s += (float)acc[k]; }
    if (s == 1.2345f) out[b] = s;
}

template <int MODE, int UNROLL, int LDS_KB, bool ATOMIC, int T = 256>
static int run(const char* name, const uint32_t* rec, const uint32_t* off, float* out, uint32_t nb, double bytes) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_stream<MODE, UNROLL, LDS_KB, ATOMIC, T>), dim3(nb), dim3(T), 0, 0, rec, off, out);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_stream<MODE, UNROLL, LDS_KB, ATOMIC, T>), dim3(nb), dim3(T), 0, 0, rec, off, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-44s nb %5u : %7.1f us  %6.2f TB/s\n", name, nb, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    const uint32_t total = 24664560;                    // records (12 B each) as in tools/bwd_bench.hip
    uint32_t* rec; float* out; uint32_t* off;
    CHECK(hipMalloc(&rec, (size_t)total * 12 + 64)); CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&off, (65536 + 1) * 4));
    std::vector<uint32_t> h((size_t)total * 3);
    uint32_t x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 8; }
    CHECK(hipMemcpy(rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double bytes = (double)total * 12;
    for (uint32_t nb : {2048u, 4096u, 8192u}) {
        std::vector<uint32_t> o(nb + 1);
        for (uint32_t i = 0; i <= nb; ++i) o[i] = (uint32_t)((uint64_t)total * i / nb);
        CHECK(hipMemcpy(off, o.data(), o.size() * 4, hipMemcpyHostToDevice));
        run<1, 8, 32, true, 256>("rec 3xdword, 32 KB, f64 atomics, 256 thr x8", rec, off, out, nb, bytes);
        run<1, 16, 32, true, 256>("rec 3xdword, 32 KB, f64 atomics, 256 thr x16", rec, off, out, nb, bytes);
        run<1, 4, 32, true, 512>("rec 3xdword, 32 KB, f64 atomics, 512 thr x4", rec, off, out, nb, bytes);
        run<1, 8, 32, true, 512>("rec 3xdword, 32 KB, f64 atomics, 512 thr x8", rec, off, out, nb, bytes);
        run<1, 4, 32, true, 1024>("rec 3xdword, 32 KB, f64 atomics, 1024 thr x4", rec, off, out, nb, bytes);
        run<1, 8, 32, false, 512>("rec 3xdword, 32 KB, no atomics, 512 thr x8", rec, off, out, nb, bytes);
        run<1, 8, 16, true, 256>("rec 3xdword, 16 KB, f64 atomics, 256 thr x8", rec, off, out, nb, bytes);
        run<1, 8, 64, true, 1024>("rec 3xdword, 64 KB, f64 atomics, 1024 thr x8", rec, off, out, nb, bytes);
        run<1, 4, 64, true, 1024>("rec 3xdword, 64 KB, f64 atomics, 1024 thr x4", rec, off, out, nb, bytes);
    }
    return 0;
}
