// emlp stage with the weight slices staged through LDS (shared by the 8 waves of a 512-thread WG),
// activations register-resident (TRR). One barrier per hidden chunk; next chunk's weights prefetched.
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
constexpr int D = 128, DFF = 256;
constexpr int NW = 8;               // waves per workgroup
constexpr int SLICE = 3 * 1024;     // float4 per chunk: v tile (16 kg) + g tile (16 kg) + wout (4 tiles x 4 kg), 1 KiB = 64 float4 each
// gemm with A fragments read from LDS: Ws[(t * KGS + kg) * 64 + lane]
template <int KGS, int NT>
__device__ __forceinline__ void gemm_s(const float4* Ws, const float4* x, f32x16 (&acc)[NT], int lane) {
#pragma unroll
    for (int kg = 0; kg < KGS; kg++) {
        float4 w[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) w[t] = Ws[(t * KGS + kg) * 64 + lane];
        const float4 xv = x[kg];
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].x, xv.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].y, xv.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].z, xv.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t].w, xv.w, acc[t], 0, 0, 0);
    }
}
// source float4 index (in the packed global arrays) of slice element i of chunk hc
__device__ __forceinline__ const float4* slice_src(const float4* win, const float4* wout, int hc, int i) {
    if (i < 1024) return win + (size_t)hc * 1024 + i;                      // v tile hc: 16 kg x 64
    if (i < 2048) return win + (size_t)(8 + hc) * 1024 + (i - 1024);       // g tile
    const int j = i - 2048, t = j >> 8, r = j & 255;                       // wout: tile t, kgs 4hc..4hc+3
    return wout + ((size_t)t * 32 + 4 * hc) * 64 + r;
}
__global__ __launch_bounds__(512) void k(const float* __restrict__ X1, const float* __restrict__ gamma, const float4* __restrict__ win,
                                         const float* __restrict__ bin, const float4* __restrict__ wout, const float* __restrict__ bout,
                                         float* __restrict__ VG, float* __restrict__ X2, int64_t E) {
    extern __shared__ float4 Ws[];   // [2][SLICE]
    const RowLane L;
    const int64_t row0 = ((int64_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 32;
    const bool valid = row0 + L.r < E; const int64_t row = valid ? row0 + L.r : E - 1;
    float4 stage[6];
#pragma unroll
    for (int s = 0; s < 6; s++) stage[s] = *slice_src(win, wout, 0, threadIdx.x + 512 * s);
    float4 x[16];
    load_rowfrag<16>(x, X1, row, D, L.h);
    rmsnorm_frag<16>(x, gamma, L.h);
    f32x16 out[4]; acc_bias<4>(out, bout, 0, L.h);
#pragma unroll
    for (int s = 0; s < 6; s++) Ws[threadIdx.x + 512 * s] = stage[s];
    __syncthreads();
#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        const float4* W0 = Ws + (hc & 1) * SLICE;
        if (hc + 1 < DFF / 32) {
#pragma unroll
            for (int s = 0; s < 6; s++) stage[s] = *slice_src(win, wout, hc + 1, threadIdx.x + 512 * s);
        }
        f32x16 v[1], g[1];
        acc_bias<1>(v, bin, 32 * hc, L.h); acc_bias<1>(g, bin, DFF + 32 * hc, L.h);
        gemm_s<16, 1>(W0, x, v, L.lane);
        gemm_s<16, 1>(W0 + 1024, x, g, L.lane);
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
        gemm_s<4, 4>(W0 + 2048, u, out, L.lane);
        if (hc + 1 < DFF / 32) {
            float4* W1 = Ws + ((hc + 1) & 1) * SLICE;
#pragma unroll
            for (int s = 0; s < 6; s++) W1[threadIdx.x + 512 * s] = stage[s];
        }
        __syncthreads();
    }
    if (valid) {
        float4 y[16], xr[16]; acc_to_frag<4>(out, y); load_rowfrag<16>(xr, X1, row, D, L.h);
#pragma unroll
        for (int i = 0; i < 16; i++) { y[i].x += xr[i].x; y[i].y += xr[i].y; y[i].z += xr[i].z; y[i].w += xr[i].w; }
        store_rowfrag<16>(y, X2, row, D, L.h);
    }
}
int main() {
    int64_t E = 381910;
    float *X1, *X2, *VG, *gamma, *bin, *bout; float4 *win, *wout;
    int grid = (E + 32 * NW - 1) / (32 * NW);
    hipMalloc(&X1, E * 512 + 4096); hipMalloc(&X2, E * 512 + 4096); hipMalloc(&VG, E * 2048 + 4096); hipMalloc(&gamma, 512); hipMalloc(&bin, 2048); hipMalloc(&bout, 512);
    hipMalloc(&win, 512 * 128 * 4); hipMalloc(&wout, 128 * 256 * 4);
    hipMemset(X1, 0, E * 512); hipMemset(gamma, 0, 512); hipMemset(bin, 0, 2048); hipMemset(bout, 0, 512); hipMemset(win, 0, 512 * 128 * 4); hipMemset(wout, 0, 128 * 256 * 4);
    size_t lds = 2 * SLICE * 16;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<grid, 512, lds>>>(X1, gamma, win, bin, wout, bout, VG, X2, E);
    hipEventRecord(e0); for (int i = 0; i < 3; i++) k<<<grid, 512, lds>>>(X1, gamma, win, bin, wout, bout, VG, X2, E); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("emlp LDS-staged weights, 8 waves/WG, lds %zu KB: %.1f us  %.1f TF/s  (%s)\n", lds / 1024, ms * 1e3, E * 2.0 * (128 * 512 + 256 * 128) / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
