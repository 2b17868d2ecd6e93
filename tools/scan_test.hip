// tools/scan_test.hip -- unit check of level_records() run-combining against an uncombined host sum (development aid)
#include "../uni-slam_amd/csrc/api.cpp"
#include "../uni-slam_amd/csrc/hashgrid.hip"
#include "../uni-slam_amd/csrc/hashgrid_binned.hip"
#include <vector>
#include <map>
#include <random>

__global__ void k_test(LevelTable tab, const float* x, const float* dy, float* val, uint32_t* idx, int* tail) {
    const int lane = threadIdx.x & 63;
    const int i = threadIdx.x;
    const LevelGeom g = level_geom(tab, 0);
    float xv[3] = {x[i * 3], x[i * 3 + 1], x[i * 3 + 2]};
    float d[2] = {dy[i * 2], dy[i * 2 + 1]};
    LevelRecords<2> r;
    level_records<2, false>(g, xv, d, d[0] != 0.f || d[1] != 0.f, lane, r);
    for (int c = 0; c < 8; ++c) { val[(i * 8 + c) * 2] = r.val[c][0]; val[(i * 8 + c) * 2 + 1] = r.val[c][1]; idx[i * 8 + c] = r.idx[c]; }
    tail[i] = r.tail;
}

__global__ void k_dpp(uint32_t* out) {
    const uint32_t v = 1000u + threadIdx.x;
    out[threadIdx.x * 4 + 0] = dpp_u32<DPP_ROW_SHL1>(v);
    out[threadIdx.x * 4 + 1] = dpp_u32<DPP_ROW_SHR(1)>(v);
    out[threadIdx.x * 4 + 2] = dpp_u32<DPP_ROW_SHR(2)>(v);
    out[threadIdx.x * 4 + 3] = dpp_u32<DPP_ROW_SHR(4)>(v);
}

int main() {
    { uint32_t* o; hipMalloc(&o, 64 * 16); k_dpp<<<1, 64>>>(o); std::vector<uint32_t> h(256); hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
      for (int i = 0; i < 20; ++i) printf("lane %2d: shl1 %u shr1 %u shr2 %u shr4 %u\n", i, h[i*4], h[i*4+1], h[i*4+2], h[i*4+3]); }
    us_grid_desc d; us_grid_desc_init(&d, 16, 2, 16, 16, 1.2996847159335432f);
    LevelTable t = make_table(&d);
    const int n = 256;
    std::mt19937 rng(3); std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float> x(n * 3), dy(n * 2);
    for (int i = 0; i < n; ++i) {
        // runs: copy the previous point's cell with probability 0.6 (jitter inside the cell)
        if (i > 0 && U(rng) < 0.6f) for (int k = 0; k < 3; ++k) { float c = floorf(15.f * x[(i - 1) * 3 + k] + 0.5f); x[i * 3 + k] = fminf(fmaxf((c - 0.5f + 0.05f + 0.9f * U(rng)) / 15.f, 0.f), 1.f); }
        else for (int k = 0; k < 3; ++k) x[i * 3 + k] = U(rng);
        dy[i * 2] = (U(rng) < 0.1f) ? 0.f : U(rng) - 0.5f; dy[i * 2 + 1] = dy[i * 2] == 0.f ? 0.f : U(rng) - 0.5f;
    }
    float *dx, *ddy, *dval; uint32_t* didx; int* dtail;
    hipMalloc(&dx, n * 12); hipMalloc(&ddy, n * 8); hipMalloc(&dval, n * 64); hipMalloc(&didx, n * 32); hipMalloc(&dtail, n * 4);
    hipMemcpy(dx, x.data(), n * 12, hipMemcpyHostToDevice); hipMemcpy(ddy, dy.data(), n * 8, hipMemcpyHostToDevice);
    k_test<<<1, n>>>(t, dx, ddy, dval, didx, dtail);
    std::vector<float> val(n * 16); std::vector<uint32_t> idx(n * 8); std::vector<int> tail(n);
    hipMemcpy(val.data(), dval, n * 64, hipMemcpyDeviceToHost); hipMemcpy(idx.data(), didx, n * 32, hipMemcpyDeviceToHost); hipMemcpy(tail.data(), dtail, n * 4, hipMemcpyDeviceToHost);
    // host: uncombined sums per entry
    std::map<uint32_t, double> ref, got;
    for (int i = 0; i < n; ++i) {
        float pos[3]; uint32_t cell[3];
        for (int k = 0; k < 3; ++k) { float p = fmaf(15.f, x[i * 3 + k], 0.5f); float f = floorf(p); cell[k] = (uint32_t)f; pos[k] = p - f; }
        for (int c = 0; c < 8; ++c) {
            float w = ((c & 1) ? pos[0] : 1 - pos[0]); w *= ((c & 2) ? pos[1] : 1 - pos[1]); w *= ((c & 4) ? pos[2] : 1 - pos[2]);
            uint32_t e = (cell[0] + (c & 1)) + (cell[1] + ((c >> 1) & 1)) * 16 + (cell[2] + ((c >> 2) & 1)) * 256; if (e >= 4096) e %= 4096;
            ref[e * 2] += w * dy[i * 2]; ref[e * 2 + 1] += w * dy[i * 2 + 1];
        }
        if (tail[i] & 1) for (int c = 0; c < 8; ++c) { got[idx[i * 8 + c] * 2] += val[(i * 8 + c) * 2]; got[idx[i * 8 + c] * 2 + 1] += val[(i * 8 + c) * 2 + 1]; }
    }
    int bad = 0, tails = 0; for (int i = 0; i < n; ++i) tails += tail[i] & 1;
    for (auto& kv : ref) { double g = got.count(kv.first) ? got[kv.first] : 0.0; if (fabs(g - kv.second) > 1e-5 * (1 + fabs(kv.second))) { if (bad < 10) printf("entry %u f%u: got %g ref %g\n", kv.first / 2, kv.first % 2, g, kv.second); ++bad; } }
    printf("tails %d of %d points, mismatching values %d of %zu\n", tails, n, bad, ref.size());
    for (int i = 0; i < 24; ++i) { uint32_t c0 = (uint32_t)floorf(15.f * x[i * 3] + 0.5f), c1 = (uint32_t)floorf(15.f * x[i * 3 + 1] + 0.5f), c2 = (uint32_t)floorf(15.f * x[i * 3 + 2] + 0.5f); printf("i %2d cell (%2u,%2u,%2u) live %d tail %d devkey %x\n", i, c0, c1, c2, dy[i * 2] != 0.f, tail[i] & 1, (unsigned)tail[i] >> 8); }
    return 0;
}
