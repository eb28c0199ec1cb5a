// in-kernel timeline of the emlp stage (same code as k_emlp_t) with s_memtime stamps per phase
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
#include <vector>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
constexpr int D = 128, DFF = 256;
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
template <bool SAVE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X1, const float* __restrict__ gamma, const float4* __restrict__ win,
                                         const float* __restrict__ bin, const float4* __restrict__ wout, const float* __restrict__ bout,
                                         float* __restrict__ VG, float* __restrict__ X2, int64_t E, unsigned long long* T) {
    const RowLane L; const int64_t row0 = wave_row0(); if (row0 >= E) return;
    const bool valid = row0 + L.r < E; const int64_t row = valid ? row0 + L.r : E - 1;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned long long ts[4] = {0, 0, 0, 0}, t0 = now(), tl;
    float4 x[16];
    load_rowfrag<16>(x, X1, row, D, L.h);
    rmsnorm_frag<16>(x, gamma, L.h);
    f32x16 out[4]; acc_bias<4>(out, bout, 0, L.h);
    tl = now();
#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        unsigned long long a = now();
        f32x16 v[1], g[1];
        acc_bias<1>(v, bin, 32 * hc, L.h); acc_bias<1>(g, bin, DFF + 32 * hc, L.h);
        gemm_t<16, 1, 4>(win, 16, 0, hc, x, v, L.lane);
        gemm_t<16, 1, 4>(win, 16, 0, DFF / 32 + hc, x, g, L.lane);
        asm volatile("" :: "v"(v[0][15]), "v"(g[0][15]));
        unsigned long long b = now();
        float4 u[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 vv = acc_q(v[0], q), gg = acc_q(g[0], q);
            if (SAVE && valid) {
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + 32 * hc + 8 * q + 4 * L.h) = vv;
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + DFF + 32 * hc + 8 * q + 4 * L.h) = gg;
            }
            u[q] = make_float4(vv.x * sigm_(gg.x), vv.y * sigm_(gg.y), vv.z * sigm_(gg.z), vv.w * sigm_(gg.w));
        }
        asm volatile("" :: "v"(u[3].w));
        unsigned long long c = now();
        gemm_t<4, 4>(wout, DFF / 8, 4 * hc, 0, u, out, L.lane);
        asm volatile("" :: "v"(out[3][15]));
        unsigned long long d = now();
        ts[0] += b - a; ts[1] += c - b; ts[2] += d - c;
    }
    unsigned long long te = now();
    if (valid) {
        float4 y[16], xr[16]; acc_to_frag<4>(out, y); load_rowfrag<16>(xr, X1, row, D, L.h);
#pragma unroll
        for (int i = 0; i < 16; i++) { y[i].x += xr[i].x; y[i].y += xr[i].y; y[i].z += xr[i].z; y[i].w += xr[i].w; }
        store_rowfrag<16>(y, X2, row, D, L.h);
    }
    unsigned long long tf = now();
    if (L.lane == 0) { T[gw * 6] = tl - t0; T[gw * 6 + 1] = ts[0]; T[gw * 6 + 2] = ts[1]; T[gw * 6 + 3] = ts[2]; T[gw * 6 + 4] = tf - te; T[gw * 6 + 5] = tf - t0; }
}
template <bool SAVE> void run(int64_t E) {
    float *X1, *X2, *VG, *gamma, *bin, *bout; float4 *win, *wout; unsigned long long* T;
    int grid = (E + 127) / 128, nw = grid * 4;
    hipMalloc(&X1, E * 512); hipMalloc(&X2, E * 512); hipMalloc(&VG, E * 2048); hipMalloc(&gamma, 512); hipMalloc(&bin, 2048); hipMalloc(&bout, 512);
    hipMalloc(&win, 512 * 128 * 4); hipMalloc(&wout, 128 * 256 * 4); hipMalloc(&T, nw * 48);
    hipMemset(X1, 0, E * 512); hipMemset(gamma, 0, 512); hipMemset(bin, 0, 2048); hipMemset(bout, 0, 512); hipMemset(win, 0, 512 * 128 * 4); hipMemset(wout, 0, 128 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SAVE><<<grid, 256>>>(X1, gamma, win, bin, wout, bout, VG, X2, E, T);
    hipEventRecord(e0); k<SAVE><<<grid, 256>>>(X1, gamma, win, bin, wout, bout, VG, X2, E, T); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nw * 6); hipMemcpy(h.data(), T, nw * 48, hipMemcpyDeviceToHost);
    double a[6] = {0}; for (int w = 0; w < nw; w++) for (int i = 0; i < 6; i++) a[i] += h[w * 6 + i];
    printf("save=%d: %.1f us %.1f TF/s | per wave cycles: prologue %.0f  gemm1 %.0f  gate+store %.0f  gemm2 %.0f  epilogue %.0f  total %.0f (own MFMA time 98304)\n",
           SAVE, ms * 1e3, E * 2.0 * (128 * 512 + 256 * 128) / ms / 1e9, a[0] / nw, a[1] / nw, a[2] / nw, a[3] / nw, a[4] / nw, a[5] / nw);
}
int main() { run<true>(381910); run<false>(381910); return 0; }
