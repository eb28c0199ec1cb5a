// ablations of the emlp stage: VG stores on/off, cheap gate on/off, NT=2 fused v|g gemm
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
constexpr int D = 128, DFF = 256;
template <bool STORE_VG, bool CHEAP, bool LOADX>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X1, const float* __restrict__ gamma, const float4* __restrict__ win,
                                         const float* __restrict__ bin, const float4* __restrict__ wout, const float* __restrict__ bout,
                                         float* __restrict__ VG, float* __restrict__ X2, int64_t E) {
    const RowLane L; const int64_t row0 = wave_row0(); if (row0 >= E) return;
    const bool valid = row0 + L.r < E; const int64_t row = valid ? row0 + L.r : E - 1;
    float4 x[16];
    if (LOADX) { load_rowfrag<16>(x, X1, row, D, L.h); rmsnorm_frag<16>(x, gamma, L.h); }
    else { for (int i = 0; i < 16; i++) x[i] = make_float4(0.1f * L.lane, 0.2f, 0.3f, 0.4f); }
    f32x16 out[4]; acc_zero<4>(out);
#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        f32x16 v[1], g[1];
        acc_zero<1>(v); acc_zero<1>(g);
        gemm_t<16, 1, 4>(win, 16, 0, hc, x, v, L.lane);
        gemm_t<16, 1, 4>(win, 16, 0, DFF / 32 + hc, x, g, L.lane);
        float4 u[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 vv = acc_q(v[0], q), gg = acc_q(g[0], q);
            if (STORE_VG && valid) {
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + 32 * hc + 8 * q + 4 * L.h) = vv;
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + DFF + 32 * hc + 8 * q + 4 * L.h) = gg;
            }
            if (CHEAP) u[q] = make_float4(vv.x * gg.x, vv.y * gg.y, vv.z * gg.z, vv.w * gg.w);
            else u[q] = make_float4(vv.x * sigm_(gg.x), vv.y * sigm_(gg.y), vv.z * sigm_(gg.z), vv.w * sigm_(gg.w));
        }
        gemm_t<4, 4>(wout, DFF / 8, 4 * hc, 0, u, out, L.lane);
    }
    if (valid) { float4 y[16]; acc_to_frag<4>(out, y); store_rowfrag<16>(y, X2, row, D, L.h); }
}
template <bool A, bool B, bool C> void run(int64_t E, float* X1, float* gamma, float4* win, float* bin, float4* wout, float* bout, float* VG, float* X2) {
    int grid = (E + 127) / 128;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<A, B, C><<<grid, 256>>>(X1, gamma, win, bin, wout, bout, VG, X2, E);
    hipEventRecord(e0); for (int i = 0; i < 3; i++) k<A, B, C><<<grid, 256>>>(X1, gamma, win, bin, wout, bout, VG, X2, E); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("storeVG=%d cheapGate=%d loadX=%d: %.1f us  %.1f TF/s\n", A, B, C, ms * 1e3, E * 2.0 * (128 * 512 + 256 * 128) / ms / 1e9);
}
int main() {
    int64_t E = 381910;
    float *X1, *X2, *VG, *gamma, *bin, *bout; float4 *win, *wout;
    hipMalloc(&X1, E * 512); hipMalloc(&X2, E * 512); hipMalloc(&VG, E * 2048); hipMalloc(&gamma, 512); hipMalloc(&bin, 2048); hipMalloc(&bout, 512);
    hipMalloc(&win, 512 * 128 * 4); hipMalloc(&wout, 128 * 256 * 4);
    hipMemset(X1, 0, E * 512); hipMemset(gamma, 0, 512); hipMemset(bin, 0, 2048); hipMemset(bout, 0, 512); hipMemset(win, 0, 512 * 128 * 4); hipMemset(wout, 0, 128 * 256 * 4);
    run<true, false, true>(E, X1, gamma, win, bin, wout, bout, VG, X2);
    run<false, false, true>(E, X1, gamma, win, bin, wout, bout, VG, X2);
    run<true, true, true>(E, X1, gamma, win, bin, wout, bout, VG, X2);
    run<false, true, true>(E, X1, gamma, win, bin, wout, bout, VG, X2);
    run<false, true, false>(E, X1, gamma, win, bin, wout, bout, VG, X2);
    return 0;
}
