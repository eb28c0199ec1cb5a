// emlp stage throughput vs forced occupancy (dynamic LDS padding limits workgroups per CU)
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
constexpr int D = 128, DFF = 256;
__global__ __launch_bounds__(256) void k(const float* __restrict__ X1, const float* __restrict__ gamma, const float4* __restrict__ win,
                                         const float* __restrict__ bin, const float4* __restrict__ wout, const float* __restrict__ bout,
                                         float* __restrict__ VG, float* __restrict__ X2, int64_t E) {
    extern __shared__ float pad[];
    const RowLane L; const int64_t row0 = wave_row0(); if (row0 >= E) return;
    const bool valid = row0 + L.r < E; const int64_t row = valid ? row0 + L.r : E - 1;
    float4 x[16];
    load_rowfrag<16>(x, X1, row, D, L.h);
    rmsnorm_frag<16>(x, gamma, L.h);
    f32x16 out[4]; acc_bias<4>(out, bout, 0, L.h);
#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        f32x16 v[1], g[1];
        acc_bias<1>(v, bin, 32 * hc, L.h); acc_bias<1>(g, bin, DFF + 32 * hc, L.h);
        gemm_t<16, 1, 4>(win, 16, 0, hc, x, v, L.lane);
        gemm_t<16, 1, 4>(win, 16, 0, DFF / 32 + hc, x, g, L.lane);
        float4 u[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 vv = acc_q(v[0], q), gg = acc_q(g[0], q);
            if (valid) {
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + 32 * hc + 8 * q + 4 * L.h) = vv;
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + DFF + 32 * hc + 8 * q + 4 * L.h) = gg;
            }
            u[q] = make_float4(vv.x * sigm_(gg.x), vv.y * sigm_(gg.y), vv.z * sigm_(gg.z), vv.w * sigm_(gg.w));
        }
        gemm_t<4, 4>(wout, DFF / 8, 4 * hc, 0, u, out, L.lane);
    }
    if (valid) {
        float4 y[16], xr[16]; acc_to_frag<4>(out, y); load_rowfrag<16>(xr, X1, row, D, L.h);
#pragma unroll
        for (int i = 0; i < 16; i++) { y[i].x += xr[i].x; y[i].y += xr[i].y; y[i].z += xr[i].z; y[i].w += xr[i].w; }
        store_rowfrag<16>(y, X2, row, D, L.h);
    }
    if (row0 < 0) pad[0] = 1.f;
}
int main() {
    int64_t E = 381910;
    float *X1, *X2, *VG, *gamma, *bin, *bout; float4 *win, *wout;
    int grid = (E + 127) / 128;
    hipMalloc(&X1, E * 512); hipMalloc(&X2, E * 512); hipMalloc(&VG, E * 2048); hipMalloc(&gamma, 512); hipMalloc(&bin, 2048); hipMalloc(&bout, 512);
    hipMalloc(&win, 512 * 128 * 4); hipMalloc(&wout, 128 * 256 * 4);
    hipMemset(X1, 0, E * 512); hipMemset(gamma, 0, 512); hipMemset(bin, 0, 2048); hipMemset(bout, 0, 512); hipMemset(win, 0, 512 * 128 * 4); hipMemset(wout, 0, 128 * 256 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t lds : {0ul, 70ul * 1024, 100ul * 1024}) {
        k<<<grid, 256, lds>>>(X1, gamma, win, bin, wout, bout, VG, X2, E);
        hipEventRecord(e0); for (int i = 0; i < 3; i++) k<<<grid, 256, lds>>>(X1, gamma, win, bin, wout, bout, VG, X2, E); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("dyn LDS %3zu KB (=> %s WG/CU): %.1f us  %.1f TF/s\n", lds / 1024, lds == 0 ? "reg-limited" : (lds < 80 * 1024 ? "2" : "1"), ms * 1e3, E * 2.0 * (128 * 512 + 256 * 128) / ms / 1e9);
    }
    return 0;
}
