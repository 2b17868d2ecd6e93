// tools/bwd_bench.hip -- development microbenchmark of the table-gradient kernels (not product code).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics [-DUS_EXP_...] tools/bwd_bench.hip -o /tmp/bwd_bench
#include "../uni-slam_amd/csrc/api.cpp"
#include "../uni-slam_amd/csrc/hashgrid.hip"
#include "../uni-slam_amd/csrc/hashgrid_binned.hip"
#include <vector>
#include <random>

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 262144;
    us_grid_desc d;
    for (uint32_t log2T : {16u, 19u}) {
        us_grid_desc_init(&d, 16, 2, log2T, 16, 1.2996847159335432f);
        std::mt19937 rng(1);
        std::uniform_real_distribution<float> U(0.f, 1.f);
        std::vector<float> hx(n * 3), hdy(n * 32);
        // ray-like points: 64 consecutive samples along a line
        for (int64_t r = 0; r < n / 64; ++r) {
            float o[3] = {0.4f + 0.2f * U(rng), 0.4f + 0.2f * U(rng), 0.4f + 0.2f * U(rng)}, dd[3] = {U(rng) - 0.5f, U(rng) - 0.5f, U(rng) - 0.5f};
            for (int s = 0; s < 64; ++s) for (int k = 0; k < 3; ++k) hx[(r * 64 + s) * 3 + k] = fminf(fmaxf(o[k] + dd[k] * s / 64.0f, 0.f), 1.f);
        }
        for (auto& v : hdy) v = U(rng) - 0.5f;
        float *x, *dy, *g; hipMalloc(&x, n * 12); hipMalloc(&dy, n * 128); hipMalloc(&g, (size_t)d.n_params * 4);
        hipMemcpy(x, hx.data(), n * 12, hipMemcpyHostToDevice); hipMemcpy(dy, hdy.data(), n * 128, hipMemcpyHostToDevice);
        hipMemset(g, 0, (size_t)d.n_params * 4);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int mode : {1}) {
            for (int lm : {0, 2}) {
                us_hashgrid_bwd_params(&d, x, dy, n, g, mode, lm, 0); hipDeviceSynchronize();
                hipEventRecord(a);
                for (int r = 0; r < 3; ++r) us_hashgrid_bwd_params(&d, x, dy, n, g, mode, lm, 0);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("log2T %2u  n %ld  mode %d  layout %s : %8.3f ms\n", log2T, (long)n, mode, lm ? "level-major" : "row-major", ms / 3);
            }
        }
        {
            const size_t wsb = us_hashgrid_bwd_workspace_bytes(&d, n);
            void* ws; hipMalloc(&ws, wsb);
            for (int lm : {0, 2}) {
                int rc = us_hashgrid_bwd_binned(&d, x, dy, n, g, lm | US_GRID_BWD_OVERWRITE, ws, wsb, 0); hipDeviceSynchronize();
                if (rc) printf("binned rc=%d %s\n", rc, us_last_error());
                hipEventRecord(a);
                for (int r = 0; r < 3; ++r) us_hashgrid_bwd_binned(&d, x, dy, n, g, lm | US_GRID_BWD_OVERWRITE, ws, wsb, 0);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                std::vector<uint32_t> hdr(2 * (BIN_MAX_TOTAL + 64)); hipMemcpy(hdr.data(), ws, hdr.size() * 4, hipMemcpyDeviceToHost);
                uint32_t tot = 0, mx = 0, nz = 0; for (int q = 0; q < BIN_MAX_TOTAL; ++q) { tot += hdr[q]; if (hdr[q] > mx) mx = hdr[q]; nz += hdr[q] != 0; }   // bin totals
                printf("log2T %2u  n %ld  BINNED  layout %s : %8.3f ms   (workspace %.0f MB, records %u of %ld, bins used %u, largest bin %u)\n", log2T, (long)n, lm ? "level-major" : "row-major", ms / 3, wsb / 1e6, tot, (long)n * 128, nz, mx);
            }
            hipFree(ws);
        }
        hipFree(x); hipFree(dy); hipFree(g);
    }
    return 0;
}
