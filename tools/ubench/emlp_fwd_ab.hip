// Kernel-level A/B of the two edge-MLP forward kernels on random data: the persistent k_emlp_h against the software-
// pipelined k_emlp_p2 (same buffers, same packed weight planes: random fp16 fragments). Prints the largest difference of
// X2 and of the saved pre-activations relative to their largest entries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -I metatrain_amd/csrc -I include tools/ubench/emlp_fwd_ab.hip -o tools/ubench/emlp_fwd_ab.bin
#include "../../metatrain_amd/csrc/pet_trr.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
using namespace pet;
namespace pet {  // what pet_trr.hip expects from the other translation units
bool use_trr() { return true; }
}
static void cmp(const char* what, const std::vector<float>& a, const std::vector<float>& b, int ld) {
    double mx = 0, md = 0; size_t at = 0;
    for (size_t i = 0; i < a.size(); i++) {
        mx = std::max(mx, (double)std::fabs(a[i]));
        const double d = std::fabs((double)a[i] - b[i]);
        if (d > md || d != d) { md = d; at = i; }
    }
    printf("%s: max|.|=%.3e max diff=%.3e (rel %.3e) at row %zu col %zu\n", what, mx, md, md / mx, at / ld, at % ld);
    // rows whose difference is above 2e-6 of the largest entry, by position in the 128-row workgroup
    int hist[4] = {0, 0, 0, 0}, colhist[4] = {0, 0, 0, 0}; size_t nbad = 0;
    for (size_t r = 0; r < a.size() / ld; r++) {
        double d = 0; int cmax = 0;
        for (int c = 0; c < ld; c++) { const double e = std::fabs((double)a[r * ld + c] - b[r * ld + c]); if (e > d) { d = e; cmax = c; } }
        int ncol = 0; for (int c = 0; c < ld; c++) if (std::fabs((double)a[r * ld + c] - b[r * ld + c]) > 2e-6 * mx) ncol++;
        if (d > 2e-6 * mx && nbad < 8) printf("   (%d columns of this row are off)", ncol);
        if (d > 2e-6 * mx) { hist[(r % 128) / 32]++; colhist[(cmax % 128) / 32]++; if (nbad < 8) printf("   row %zu (wave %zu, lane row %zu) col %d diff %.3e\n", r, (r % 128) / 32, r % 32, cmax, d); nbad++; }
    }
    printf("   rows off: %zu; by wave of the workgroup: %d %d %d %d; by 32-column group of the worst column: %d %d %d %d\n", nbad, hist[0], hist[1], hist[2], hist[3], colhist[0], colhist[1], colhist[2], colhist[3]);
}
int main(int argc, char** argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : 1194;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> X1(E * D), gamma(D), bin(2 * DFF), bout(D);
    for (auto& v : X1) v = nd(rng);
    for (auto& v : gamma) v = 1.f + 0.1f * nd(rng);
    for (auto& v : bin) v = 0.3f * nd(rng);
    for (auto& v : bout) v = 0.3f * nd(rng);
    const size_t nin = (size_t)(2 * DFF / 32) * (D / 16) * 64, nout = (size_t)(D / 32) * (DFF / 16) * 64;  // fragments per plane
    std::vector<_Float16> wi(2 * nin * 8), wo(2 * nout * 8);
    for (auto& v : wi) v = (_Float16)(0.1f * nd(rng));
    for (auto& v : wo) v = (_Float16)(0.1f * nd(rng));
    float *d_X1, *d_g, *d_bi, *d_bo, *d_vg0, *d_vg1, *d_o0, *d_o1;
    _Float16 *d_wi, *d_wo;
    (void)hipMalloc(&d_X1, X1.size() * 4); (void)hipMalloc(&d_g, D * 4); (void)hipMalloc(&d_bi, bin.size() * 4); (void)hipMalloc(&d_bo, D * 4);
    (void)hipMalloc(&d_vg0, E * 2 * DFF * 4); (void)hipMalloc(&d_vg1, E * 2 * DFF * 4); (void)hipMalloc(&d_o0, E * D * 4); (void)hipMalloc(&d_o1, E * D * 4);
    (void)hipMalloc(&d_wi, wi.size() * 2); (void)hipMalloc(&d_wo, wo.size() * 2);
    (void)hipMemcpy(d_X1, X1.data(), X1.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_g, gamma.data(), D * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_bi, bin.data(), bin.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_bo, bout.data(), D * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_wi, wi.data(), wi.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(d_wo, wo.data(), wo.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemset(d_vg0, 0, E * 2 * DFF * 4); (void)hipMemset(d_vg1, 0, E * 2 * DFF * 4);
    W2 win, wout;
    win.h = reinterpret_cast<const f16x8*>(d_wi); win.l = win.h + nin;
    wout.h = reinterpret_cast<const f16x8*>(d_wo); wout.l = wout.h + nout;
    {
        const size_t lds = (size_t)4 * 2 * 16 * 64 * sizeof(float4) + (size_t)4 * 32 * TILE32_LD * sizeof(float);
        int ncu = 256;
        const int grid = std::min((int)((E + 127) / 128), ncu);
        allow_big_lds(k_emlp_h<true, false>, lds);
        k_emlp_h<true, false><<<grid, 256, lds>>>(d_X1, d_g, nullptr, win, d_bi, wout, d_bo, d_vg0, d_o0, E);
    }
    {
        const size_t lds = (size_t)4 * EP2_WAVE_LDS;
        allow_big_lds(k_emlp_p2<false>, lds);
        k_emlp_p2<false><<<(int)((E + 127) / 128), 256, lds>>>(d_X1, d_g, nullptr, win, d_bi, wout, d_bo, getenv("NOVG") ? nullptr : d_vg1, d_o1, E);
    }
    if (getenv("TIME")) {   // launch times: old kernel, new kernel, new kernel without the [v; g] stores
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        auto timeit = [&](const char* what, auto fn) {
            for (int i = 0; i < 3; i++) fn();
            (void)hipEventRecord(e0);
            for (int i = 0; i < 10; i++) fn();
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.3f ms per launch\n", what, ms / 10);
        };
        const size_t lds0 = (size_t)4 * 2 * 16 * 64 * sizeof(float4) + (size_t)4 * 32 * TILE32_LD * sizeof(float);
        const size_t lds1 = (size_t)4 * EP2_WAVE_LDS;
        const int g1 = (int)((E + 127) / 128), g0 = std::min(g1, 256);
        timeit("k_emlp_h", [&] { k_emlp_h<true, false><<<g0, 256, lds0>>>(d_X1, d_g, nullptr, win, d_bi, wout, d_bo, d_vg0, d_o0, E); });
        timeit("k_emlp_p2", [&] { k_emlp_p2<false><<<g1, 256, lds1>>>(d_X1, d_g, nullptr, win, d_bi, wout, d_bo, d_vg1, d_o1, E); });
        timeit("k_emlp_h, no VG", [&] { k_emlp_h<true, false><<<g0, 256, lds0>>>(d_X1, d_g, nullptr, win, d_bi, wout, d_bo, nullptr, d_o0, E); });
        timeit("k_emlp_p2, no VG", [&] { k_emlp_p2<false><<<g1, 256, lds1>>>(d_X1, d_g, nullptr, win, d_bi, wout, d_bo, nullptr, d_o1, E); });
    }
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) { printf("error %s\n", hipGetErrorString(err)); return 1; }
    std::vector<float> o0(E * D), o1(E * D), v0(E * 2 * DFF), v1(E * 2 * DFF);
    (void)hipMemcpy(o0.data(), d_o0, o0.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(v0.data(), d_vg0, v0.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(v1.data(), d_vg1, v1.size() * 4, hipMemcpyDeviceToHost);
    printf("E=%lld\n", (long long)E);
    {   // fp64 reference of X2 for the first rows, from the de-quantised planes W = h + l / 2048
        auto wfrag = [&](const std::vector<_Float16>& w, size_t nfr, int ktot16, int n, int k) {
            const int tile = n / 32, kb = k / 16, kk = k % 16, h = (kk % 8) / 4, j = (kk / 8) * 4 + kk % 4;
            const size_t fr = ((size_t)tile * ktot16 + kb) * 64 + (n % 32) + 32 * h;
            return (double)w[fr * 8 + j] + (double)w[(nfr + fr) * 8 + j] / 2048.0;
        };
        double worst0 = 0, worst1 = 0;
        const int64_t nref = std::min<int64_t>(E, 256);
        for (int64_t r = 0; r < nref; r++) {
            double xn[D], ss = 0;
            for (int k = 0; k < D; k++) ss += (double)X1[r * D + k] * X1[r * D + k];
            const double rstd = 1.0 / std::sqrt(ss / D + 1.1920928955078125e-07);
            for (int k = 0; k < D; k++) xn[k] = X1[r * D + k] * rstd * gamma[k];
            double u[DFF];
            for (int n = 0; n < DFF; n++) {
                double v = bin[n], g = bin[DFF + n];
                for (int k = 0; k < D; k++) { v += wfrag(wi, nin, D / 16, n, k) * xn[k]; g += wfrag(wi, nin, D / 16, DFF + n, k) * xn[k]; }
                u[n] = v / (1.0 + std::exp(-g));
            }
            for (int m = 0; m < D; m++) {
                double o = bout[m] + X1[r * D + m];
                for (int n = 0; n < DFF; n++) o += wfrag(wo, nout, DFF / 16, m, n) * u[n];
                worst0 = std::max(worst0, std::fabs(o - o0[r * D + m]));
                worst1 = std::max(worst1, std::fabs(o - o1[r * D + m]));
            }
        }
        printf("X2 against fp64 on the first %lld rows: k_emlp_h %.3e, k_emlp_p2 %.3e\n", (long long)nref, worst0, worst1);
    }
    cmp("X2", o0, o1, D);
    cmp("VG", v0, v1, 2 * DFF);
    return 0;
}
