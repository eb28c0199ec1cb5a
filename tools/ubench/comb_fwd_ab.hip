// Kernel-level A/B of the two combination-stage forward kernels on random data: k_comb_h against the software-pipelined
// k_comb_p2 (same buffers, same packed weight planes: random fp16 fragments; rev = a random permutation of the rows).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -I metatrain_amd/csrc -I include tools/ubench/comb_fwd_ab.hip -o tools/ubench/comb_fwd_ab.bin
#include "../../metatrain_amd/csrc/pet_comb.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include <algorithm>
#include <numeric>
using namespace pet;
static void cmp(const char* what, const std::vector<float>& a, const std::vector<float>& b, int ld) {
    double mx = 0, md = 0; size_t at = 0, nbad = 0;
    for (size_t i = 0; i < a.size(); i++) {
        mx = std::max(mx, (double)std::fabs(a[i]));
        const double d = std::fabs((double)a[i] - b[i]);
        if (d > md || d != d) { md = d; at = i; }
    }
    for (size_t r = 0; r < a.size() / ld; r++) {
        double d = 0;
        for (int c = 0; c < ld; c++) d = std::max(d, std::fabs((double)a[r * ld + c] - b[r * ld + c]));
        if (d > 3e-6 * mx) nbad++;
    }
    printf("%s: max|.|=%.3e max diff=%.3e (rel %.3e) at row %zu col %zu; rows off %zu\n", what, mx, md, md / mx, at / ld, at % ld, nbad);
}
int main(int argc, char** argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : 1194;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> XF(E * D), Min(E * D), lng(2 * D), lnb(2 * D), b0(2 * D), b2(D);
    for (auto& v : XF) v = nd(rng);
    for (auto& v : Min) v = nd(rng);
    for (auto& v : lng) v = 1.f + 0.1f * nd(rng);
    for (auto& v : lnb) v = 0.1f * nd(rng);
    for (auto& v : b0) v = 0.3f * nd(rng);
    for (auto& v : b2) v = 0.3f * nd(rng);
    std::vector<int> rev(E); std::iota(rev.begin(), rev.end(), 0); std::shuffle(rev.begin(), rev.end(), rng);
    const size_t n0 = (size_t)(2 * D / 32) * (2 * D / 16) * 64, n2 = (size_t)(D / 32) * (2 * D / 16) * 64;  // fragments per plane
    std::vector<_Float16> w0v(2 * n0 * 8), w2v(2 * n2 * 8);
    for (auto& v : w0v) v = (_Float16)(0.08f * nd(rng));
    for (auto& v : w2v) v = (_Float16)(0.08f * nd(rng));
    float *d_XF, *d_Min, *d_lng, *d_lnb, *d_b0, *d_b2, *d_ca0, *d_ca1, *d_ln0, *d_ln1, *d_m0, *d_m1; int* d_rev; _Float16 *d_w0, *d_w2;
    (void)hipMalloc(&d_XF, E * D * 4); (void)hipMalloc(&d_Min, E * D * 4); (void)hipMalloc(&d_lng, 2 * D * 4); (void)hipMalloc(&d_lnb, 2 * D * 4);
    (void)hipMalloc(&d_b0, 2 * D * 4); (void)hipMalloc(&d_b2, D * 4); (void)hipMalloc(&d_ca0, E * 2 * D * 4); (void)hipMalloc(&d_ca1, E * 2 * D * 4);
    (void)hipMalloc(&d_ln0, E * 2 * 4); (void)hipMalloc(&d_ln1, E * 2 * 4); (void)hipMalloc(&d_m0, E * D * 4); (void)hipMalloc(&d_m1, E * D * 4);
    (void)hipMalloc(&d_rev, E * 4); (void)hipMalloc(&d_w0, w0v.size() * 2); (void)hipMalloc(&d_w2, w2v.size() * 2);
    (void)hipMemcpy(d_XF, XF.data(), E * D * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_Min, Min.data(), E * D * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_lng, lng.data(), 2 * D * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_lnb, lnb.data(), 2 * D * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_b0, b0.data(), 2 * D * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_b2, b2.data(), D * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_rev, rev.data(), E * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_w0, w0v.data(), w0v.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(d_w2, w2v.data(), w2v.size() * 2, hipMemcpyHostToDevice);
    W2 w0, w2;
    w0.h = reinterpret_cast<const f16x8*>(d_w0); w0.l = w0.h + n0;
    w2.h = reinterpret_cast<const f16x8*>(d_w2); w2.l = w2.h + n2;
    const int grid = (int)((E + 127) / 128);
    const size_t lds0 = (size_t)4 * 8 * 2 * 64 * sizeof(f16x8), lds1 = (size_t)4 * 32768;
    allow_big_lds(k_comb_h<false>, lds0);
    allow_big_lds(k_comb_p2<false>, lds1);
    auto run0 = [&] { k_comb_h<false><<<grid, 256, lds0>>>(d_XF, d_rev, d_lng, d_lnb, w0, d_b0, w2, d_b2, d_Min, nullptr, nullptr, d_ca0, d_ln0, d_m0, E); };
    auto run1 = [&] { k_comb_p2<false><<<grid, 256, lds1>>>(d_XF, d_rev, d_lng, d_lnb, w0, d_b0, w2, d_b2, d_Min, nullptr, nullptr, d_ca1, d_ln1, d_m1, E); };
    run0(); run1();
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("error %s\n", hipGetErrorString(err)); return 1; }
    std::vector<float> m0(E * D), m1(E * D), c0(E * 2 * D), c1(E * 2 * D), l0(E * 2), l1(E * 2);
    (void)hipMemcpy(m0.data(), d_m0, m0.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(m1.data(), d_m1, m1.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c0.data(), d_ca0, c0.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c1.data(), d_ca1, c1.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(l0.data(), d_ln0, l0.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(l1.data(), d_ln1, l1.size() * 4, hipMemcpyDeviceToHost);
    printf("E=%lld\n", (long long)E);
    cmp("Mout", m0, m1, D); cmp("CA", c0, c1, 2 * D); cmp("LNS", l0, l1, 2);
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
        timeit("k_comb_h", run0); timeit("k_comb_p2", run1);
    }
    return 0;
}
