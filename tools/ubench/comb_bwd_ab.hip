// Kernel-level A/B of the two combination-stage adjoints on random data: k_comb_bwd_h against the software-pipelined
// k_comb_bwd_p2 (same buffers, same packed weight planes: random fp16 fragments; rev = a random permutation of the rows).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -I metatrain_amd/csrc -I include tools/ubench/comb_bwd_ab.hip -o tools/ubench/comb_bwd_ab.bin
#include "../../metatrain_amd/csrc/pet_comb.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include <algorithm>
#include <numeric>
using namespace pet;
int main(int argc, char** argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : 1194;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> dM(E * D), XF(E * D), LNS(E * 2), CA(E * 2 * D), lng(2 * D);
    for (auto& v : dM) v = 1e-3f * nd(rng);
    for (auto& v : XF) v = nd(rng);
    for (auto& v : CA) v = nd(rng);
    for (auto& v : lng) v = 1.f + 0.1f * nd(rng);
    for (int64_t r = 0; r < E; r++) { LNS[2 * r] = 0.05f * nd(rng); LNS[2 * r + 1] = 1.f + 0.1f * std::fabs(nd(rng)); }
    std::vector<int> rev(E); std::iota(rev.begin(), rev.end(), 0); std::shuffle(rev.begin(), rev.end(), rng);
    const size_t n2 = (size_t)(2 * D / 32) * (D / 16) * 64, n0 = (size_t)(2 * D / 32) * (2 * D / 16) * 64;  // fragments per plane
    std::vector<_Float16> w2v(2 * n2 * 8), w0v(2 * n0 * 8);
    for (auto& v : w2v) v = (_Float16)(0.08f * nd(rng));
    for (auto& v : w0v) v = (_Float16)(0.08f * nd(rng));
    float *d_dM, *d_XF, *d_LNS, *d_CA, *d_lng, *d_o0, *d_o1; int* d_rev; _Float16 *d_w2, *d_w0;
    (void)hipMalloc(&d_dM, E * D * 4); (void)hipMalloc(&d_XF, E * D * 4); (void)hipMalloc(&d_LNS, E * 2 * 4); (void)hipMalloc(&d_CA, E * 2 * D * 4);
    (void)hipMalloc(&d_lng, 2 * D * 4); (void)hipMalloc(&d_o0, E * 2 * D * 4); (void)hipMalloc(&d_o1, E * 2 * D * 4); (void)hipMalloc(&d_rev, E * 4);
    (void)hipMalloc(&d_w2, w2v.size() * 2); (void)hipMalloc(&d_w0, w0v.size() * 2);
    (void)hipMemcpy(d_dM, dM.data(), E * D * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_XF, XF.data(), E * D * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_LNS, LNS.data(), E * 2 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_CA, CA.data(), E * 2 * D * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_lng, lng.data(), 2 * D * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_rev, rev.data(), E * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_w2, w2v.data(), w2v.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(d_w0, w0v.data(), w0v.size() * 2, hipMemcpyHostToDevice);
    W2 w2b, w0b;
    w2b.h = reinterpret_cast<const f16x8*>(d_w2); w2b.l = w2b.h + n2;
    w0b.h = reinterpret_cast<const f16x8*>(d_w0); w0b.l = w0b.h + n0;
    const int grid = (int)((E + 127) / 128);
    const size_t lds0 = (size_t)4 * (2 * D / 32) * 4 * 64 * sizeof(float4), lds1 = (size_t)4 * 40960;
    allow_big_lds(k_comb_bwd_h<false>, lds0);
    allow_big_lds(k_comb_bwd_p2<false>, lds1);
    auto run0 = [&] { k_comb_bwd_h<false><<<grid, 256, lds0>>>(d_dM, d_XF, d_rev, d_LNS, d_CA, d_lng, w2b, w0b, d_o0, E, nullptr); };
    auto run1 = [&] { k_comb_bwd_p2<false><<<grid, 256, lds1>>>(d_dM, d_XF, d_rev, d_LNS, d_CA, d_lng, w2b, w0b, d_o1, E, nullptr); };
    run0(); run1();
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("error %s\n", hipGetErrorString(err)); return 1; }
    std::vector<float> o0(E * 2 * D), o1(E * 2 * D);
    (void)hipMemcpy(o0.data(), d_o0, o0.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, md = 0; size_t at = 0, nbad = 0;
    for (size_t i = 0; i < o0.size(); i++) {
        mx = std::max(mx, (double)std::fabs(o0[i]));
        const double d = std::fabs((double)o0[i] - o1[i]);
        if (d > md || d != d) { md = d; at = i; }
    }
    int hist[4] = {0, 0, 0, 0}, half[2] = {0, 0};
    for (int64_t r = 0; r < E; r++) {
        double d = 0; int cm = 0;
        for (int c = 0; c < 2 * D; c++) { const double e = std::fabs((double)o0[r * 2 * D + c] - o1[r * 2 * D + c]); if (e > d) { d = e; cm = c; } }
        if (d > 3e-6 * mx) { if (nbad < 6) printf("  row %lld (wave %lld) col %d diff %.3e\n", (long long)r, (long long)((r % 128) / 32), cm, d); nbad++; hist[(r % 128) / 32]++; half[cm / D]++; }
    }
    printf("E=%lld dcat: max|.|=%.3e max diff=%.3e (rel %.3e) at row %zu col %zu; rows off %zu (by wave %d %d %d %d; worst column in half %d %d)\n",
           (long long)E, mx, md, md / mx, at / (2 * D), at % (2 * D), nbad, hist[0], hist[1], hist[2], hist[3], half[0], half[1]);
    if (getenv("TIME")) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        auto timeit = [&](const char* what, auto fn) {
            for (int i = 0; i < 3; i++) fn();
            (void)hipEventRecord(e0);
            for (int i = 0; i < 10; i++) fn();
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms per launch\n", what, ms / 10);
        };
        timeit("k_comb_bwd_h", run0); timeit("k_comb_bwd_p2", run1);
    }
    return 0;
}
