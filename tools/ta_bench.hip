// tools/ta_bench.hip -- what does a wave's 8-byte gather cost the address path, by how many lanes share an address / a line?
// (development microbenchmark; table small enough to stay in the CU's vector cache, so this is the issue cost, not the miss cost)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ta_bench.hip -o /tmp/ta_bench && /tmp/ta_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// every lane loads ITERS x 8 float2; lane's entry index = f(lane, pattern), advanced by a per-iteration stride inside a 16 KiB table
template <int PATTERN>
__global__ __launch_bounds__(1024) void k_gather(const float2* __restrict__ table, float* __restrict__ out, int iters) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t base;
    if (PATTERN == 0) base = 0;                          // all 64 lanes one address
    else if (PATTERN == 1) base = (lane >> 4) * 16;      // runs of 16 lanes share an address; 4 lines
    else if (PATTERN == 2) base = (lane >> 2) * 16;      // runs of 4 lanes; 16 lines
    else if (PATTERN == 3) base = lane;                  // 64 consecutive entries: 4 lines, coalesced
    else if (PATTERN == 4) base = lane * 16;             // 64 distinct lines (128-byte stride)
    else base = (lane * 16) ^ ((lane & 1) * 8);          // 64 distinct lines, addresses scrambled within the line
    float s = 0.f;
    uint32_t off = wave * 7u;
    for (int it = 0; it < iters; ++it) {
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = table[(base + off + 131u * k) & 2047u];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k].x + v[k].y;
        off += 17u;
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

// random 8-byte (or 16-byte pair) gathers from a table of `entries` float2 (power of two): hash of (lane, iteration) as the index
template <int WIDE>
__global__ __launch_bounds__(1024) void k_random(const float2* __restrict__ table, uint32_t mask, float* __restrict__ out, int iters) {
    uint32_t h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (WIDE) {
            typedef float pair_t __attribute__((ext_vector_type(4), aligned(8)));
            pair_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { h = h * 1664525u + 1013904223u; v[k] = *reinterpret_cast<const pair_t*>(table + ((h >> 8) & mask & ~1u)); }
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k].x + v[k].w;
        } else {
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { h = h * 1664525u + 1013904223u; v[k] = table[(h >> 8) & mask]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k].x + v[k].y;
        }
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    float2* table; float* out;
    CHECK(hipMalloc(&table, 2048 * sizeof(float2))); CHECK(hipMemset(table, 0, 2048 * sizeof(float2)));
    CHECK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 512;
    const char* names[] = {"one address", "runs of 16 (4 lines)", "runs of 4 (16 lines)", "64 consecutive (4 lines)", "64 lines", "64 lines scrambled"};
    for (int p = 0; p < 6; ++p) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            switch (p) {
                case 0: hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(1024), 0, 0, table, out, iters); break;
                case 1: hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(1024), 0, 0, table, out, iters); break;
                case 2: hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(1024), 0, 0, table, out, iters); break;
                case 3: hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(1024), 0, 0, table, out, iters); break;
                case 4: hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(1024), 0, 0, table, out, iters); break;
                default: hipLaunchKernelGGL(k_gather<5>, dim3(blocks), dim3(1024), 0, 0, table, out, iters); break;
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) {
                const double wave_loads = (double)blocks * 16 * iters * 8;          // wave-level load instructions
                // 256 CUs, 2 workgroups of 16 waves each resident per CU: cycles per wave-load per CU at 2.4 GHz
                printf("%-28s %8.3f ms   %6.1f cycles per wave load (per CU, 2.4 GHz)\n", names[p], ms, ms * 1e-3 * 2.4e9 / (wave_loads / 256));
            }
        }
    }
    for (int lg = 11; lg <= 26; lg += (lg >= 14 && lg < 21) ? 1 : 3) {   // 16 KiB ... 512 MiB tables (every power of two between 128 KiB and 16 MiB)
        float2* big; const size_t entries = (size_t)1 << lg;
        CHECK(hipMalloc(&big, entries * sizeof(float2))); CHECK(hipMemset(big, 0, entries * sizeof(float2)));
        for (int wide = 0; wide < 2; ++wide) {
            const int it2 = 200;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (wide) hipLaunchKernelGGL(k_random<1>, dim3(blocks), dim3(1024), 0, 0, big, (uint32_t)(entries - 1), out, it2);
                else hipLaunchKernelGGL(k_random<0>, dim3(blocks), dim3(1024), 0, 0, big, (uint32_t)(entries - 1), out, it2);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) {
                    const double lane_loads = (double)blocks * 1024 * it2 * 8;
                    printf("random %s gathers, table %8.2f MiB: %8.3f ms  %7.1f G lane-loads/s  %5.1f cycles per wave load per CU\n", wide ? "16-byte" : " 8-byte",
                           entries * 8.0 / 1048576, ms, lane_loads / ms * 1e-6, ms * 1e-3 * 2.4e9 / (lane_loads / 64 / 256));
                }
            }
        }
        CHECK(hipFree(big));
    }
    return 0;
}
