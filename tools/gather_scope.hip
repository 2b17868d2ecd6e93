// tools/gather_scope.hip -- random 8-byte gathers from a table that sits in the L2, as plain loads and as agent-scope atomic loads (which
// do not allocate in the vector cache): does a gather that bypasses the CU's cache cost the L2 -> CU path less than a 128-byte line?
// (development microbenchmark)   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_scope.hip -o tools/gather_scope.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SCOPED>
__global__ __launch_bounds__(1024) void k_random(const unsigned long long* __restrict__ table, uint32_t mask, float* __restrict__ out, int iters) {
    uint32_t h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long s = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned long long v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            h = h * 1664525u + 1013904223u;
            const unsigned long long* p = table + ((h >> 8) & mask);
            if (SCOPED) v[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v[k] = *p;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    if (s == 0x123456789abcdefull) out[threadIdx.x] = 1.0f;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256, iters = 200;                      // one 1024-thread workgroup per CU, like the encoder
    for (int lg : {16, 19, 22}) {                             // 0.5, 4, 32 MiB of 8-byte entries
        unsigned long long* table; const size_t entries = (size_t)1 << lg;
        CHECK(hipMalloc(&table, entries * 8)); CHECK(hipMemset(table, 0, entries * 8));
        for (int scoped = 0; scoped < 2; ++scoped) {
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (scoped) hipLaunchKernelGGL(k_random<1>, dim3(blocks), dim3(1024), 0, 0, table, (uint32_t)(entries - 1), out, iters);
                else hipLaunchKernelGGL(k_random<0>, dim3(blocks), dim3(1024), 0, 0, table, (uint32_t)(entries - 1), out, iters);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 2) {
                    const double lane_loads = (double)blocks * 1024 * iters * 8;
                    printf("table %6.1f MiB  %-22s %8.3f ms  %7.1f G lane-loads/s\n", entries * 8.0 / 1048576, scoped ? "agent-scope atomic load" : "plain load", ms,
                           lane_loads / ms * 1e-6);
                }
            }
        }
        CHECK(hipFree(table));
    }
    return 0;
}
