// Kernel-level A/B of the two edge-MLP adjoints on random data: k_emlp_bwd_h against k_emlp_bwd_p2, same buffers, same
// packed weight planes (random fp16 fragments: the layout only has to be the same for both). Prints the largest
// difference of dX1 relative to its largest entry; with CHUNK=c only the saved pre-activations of hidden chunk c are
// non-zero (g = -30 elsewhere: sigma = 0), which localises a wrong chunk.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -I metatrain_amd/csrc -I include tools/ubench/emlp_bwd_ab.hip -o /tmp/emlp_bwd_ab
#include "../../metatrain_amd/csrc/pet_trr.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
using namespace pet;
namespace pet {  // what pet_trr.hip expects from the other translation units
bool use_trr() { return true; }
}
int main(int argc, char** argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : 1194;
    const int chunk = argc > 2 ? atoi(argv[2]) : -1;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> dY(E * D), X1(E * D), VG(E * 2 * DFF), gamma(D);
    for (auto& v : dY) v = 1e-3f * nd(rng);
    for (auto& v : X1) v = nd(rng);
    for (auto& v : gamma) v = 1.f + 0.1f * nd(rng);
    for (int64_t r = 0; r < E; r++)
        for (int n = 0; n < DFF; n++) {
            const bool on = chunk < 0 || n / 32 == chunk;
            VG[r * 2 * DFF + n] = on ? nd(rng) : 0.f;
            VG[r * 2 * DFF + DFF + n] = on ? nd(rng) : -30.f;
        }
    const size_t nout = (size_t)(DFF / 32) * (D / 16) * 64, nin = (size_t)(D / 32) * (2 * DFF / 16) * 64;  // fragments per plane
    std::vector<_Float16> wo(2 * nout * 8), wi(2 * nin * 8);
    for (auto& v : wo) v = (_Float16)(0.1f * nd(rng));
    for (auto& v : wi) v = (_Float16)(0.1f * nd(rng));
    float *d_dY, *d_X1, *d_VG, *d_g, *d_o0, *d_o1;
    _Float16 *d_wo, *d_wi;
    hipMalloc(&d_dY, dY.size() * 4); hipMalloc(&d_X1, X1.size() * 4); hipMalloc(&d_VG, VG.size() * 4); hipMalloc(&d_g, D * 4);
    hipMalloc(&d_o0, E * D * 4); hipMalloc(&d_o1, E * D * 4); hipMalloc(&d_wo, wo.size() * 2); hipMalloc(&d_wi, wi.size() * 2);
    hipMemcpy(d_dY, dY.data(), dY.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_X1, X1.data(), X1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_VG, VG.data(), VG.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_g, gamma.data(), D * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_wo, wo.data(), wo.size() * 2, hipMemcpyHostToDevice); hipMemcpy(d_wi, wi.data(), wi.size() * 2, hipMemcpyHostToDevice);
    W2 woutb, winb;
    woutb.h = reinterpret_cast<const f16x8*>(d_wo); woutb.l = woutb.h + nout;
    winb.h = reinterpret_cast<const f16x8*>(d_wi); winb.l = winb.h + nin;
    const int grid = (int)((E + 127) / 128);
    k_emlp_bwd_h<false, false><<<grid, 256>>>(d_dY, d_X1, d_VG, d_g, woutb, winb, d_o0, E, nullptr);
    const size_t lds = (size_t)4 * 40960;
    allow_big_lds(k_emlp_bwd_p2<false, false>, lds);
    k_emlp_bwd_p2<false, false><<<grid, 256, lds>>>(d_dY, d_X1, d_VG, d_g, woutb, winb, d_o1, E, nullptr);
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
        timeit("k_emlp_bwd_h", [&] { k_emlp_bwd_h<false, false><<<grid, 256>>>(d_dY, d_X1, d_VG, d_g, woutb, winb, d_o0, E, nullptr); });
        timeit("k_emlp_bwd_p2", [&] { k_emlp_bwd_p2<false, false><<<grid, 256, lds>>>(d_dY, d_X1, d_VG, d_g, woutb, winb, d_o1, E, nullptr); });
    }
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("error %s\n", hipGetErrorString(err)); return 1; }
    std::vector<float> o0(E * D), o1(E * D);
    hipMemcpy(o0.data(), d_o0, o0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, md = 0; int64_t at = 0;
    for (size_t i = 0; i < o0.size(); i++) {
        mx = std::max(mx, (double)std::fabs(o0[i]));
        const double d = std::fabs((double)o0[i] - o1[i]);
        if (d > md) { md = d; at = i; }
    }
    // per 32-row tile and per 32-column group: where the differences sit
    printf("E=%lld chunk=%d max|dX1|=%.3e max diff=%.3e (rel %.3e) at row %lld col %lld\n", (long long)E, chunk, mx, md, md / mx,
           (long long)(at / D), (long long)(at % D));
    int bad_rows = 0;
    for (int64_t r = 0; r < E; r++) {
        double d = 0;
        for (int c = 0; c < D; c++) d = std::max(d, std::fabs((double)o0[r * D + c] - o1[r * D + c]));
        if (d > 1e-5 * mx) { if (bad_rows < 6) printf("  row %lld (lane row %lld of its tile) diff %.3e\n", (long long)r, (long long)(r % 32), d); bad_rows++; }
    }
    printf("rows off: %d of %lld\n", bad_rows, (long long)E);
    return 0;
}
