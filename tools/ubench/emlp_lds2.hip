// emlp stage, weights staged through a double-buffered 16 KiB LDS ring shared by the 4 waves of a WG
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
constexpr int D = 128, DFF = 256;
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
// 16 KiB slice = 1024 float4; thread tid handles elements tid + 256 s, s = 0..3
struct Ring {
    float4* lds; int cur; float4 st[4];
    __device__ void fetch_tile(const float4* W, int kg_total, int tile, int kg0) {   // NT=1, KGS=16: contiguous
        const float4* p = W + ((size_t)tile * kg_total + kg0) * 64 + threadIdx.x;
#pragma unroll
        for (int s = 0; s < 4; s++) st[s] = p[256 * s];
    }
    __device__ void fetch_4x4(const float4* W, int kg_total, int kg0) {              // NT=4 (tiles 0..3), KGS=4
#pragma unroll
        for (int s = 0; s < 4; s++) st[s] = W[((size_t)s * kg_total + kg0) * 64 + threadIdx.x];   // piece t = s, 256 float4 each
    }
    __device__ void commit() {
        float4* dst = lds + (cur ^ 1) * 1024;
#pragma unroll
        for (int s = 0; s < 4; s++) dst[threadIdx.x + 256 * s] = st[s];
        __syncthreads();
        cur ^= 1;
    }
    __device__ const float4* buf() const { return lds + cur * 1024; }
};
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ X1, const float* __restrict__ gamma, const float4* __restrict__ win,
                                            const float* __restrict__ bin, const float4* __restrict__ wout, const float* __restrict__ bout,
                                            float* __restrict__ VG, float* __restrict__ X2, int64_t E) {
    __shared__ float4 Ws[2048];
    Ring ring{Ws, 1, {}};
    const RowLane L; const int64_t row0 = wave_row0();
    const bool valid = row0 + L.r < E; const int64_t row = valid ? row0 + L.r : E - 1;
    ring.fetch_tile(win, 16, 0, 0);
    float4 x[16];
    load_rowfrag<16>(x, X1, row, D, L.h);
    rmsnorm_frag<16>(x, gamma, L.h);
    f32x16 out[4]; acc_bias<4>(out, bout, 0, L.h);
    ring.commit();                                   // v tile of chunk 0 ready
#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        f32x16 v[1], g[1];
        acc_bias<1>(v, bin, 32 * hc, L.h); acc_bias<1>(g, bin, DFF + 32 * hc, L.h);
        ring.fetch_tile(win, 16, DFF / 32 + hc, 0);  // g tile
        gemm_s<16, 1>(ring.buf(), x, v, L.lane);
        ring.commit();
        ring.fetch_4x4(wout, DFF / 8, 4 * hc);       // wout slice
        gemm_s<16, 1>(ring.buf(), x, g, L.lane);
        ring.commit();
        if (hc + 1 < DFF / 32) ring.fetch_tile(win, 16, hc + 1, 0);
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
        gemm_s<4, 4>(ring.buf(), u, out, L.lane);
        if (hc + 1 < DFF / 32) ring.commit(); 
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
    int grid = (E + 127) / 128;
    hipMalloc(&X1, E * 512 + 65536); hipMalloc(&X2, E * 512 + 65536); hipMalloc(&VG, E * 2048 + 65536); hipMalloc(&gamma, 512); hipMalloc(&bin, 2048); hipMalloc(&bout, 512);
    hipMalloc(&win, 512 * 128 * 4); hipMalloc(&wout, 128 * 256 * 4);
    hipMemset(X1, 0, E * 512); hipMemset(gamma, 0, 512); hipMemset(bin, 0, 2048); hipMemset(bout, 0, 512); hipMemset(win, 0, 512 * 128 * 4); hipMemset(wout, 0, 128 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<grid, 256>>>(X1, gamma, win, bin, wout, bout, VG, X2, E);
    hipEventRecord(e0); for (int i = 0; i < 3; i++) k<<<grid, 256>>>(X1, gamma, win, bin, wout, bout, VG, X2, E); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("emlp, LDS weight ring (4 waves/WG): %.1f us  %.1f TF/s (%s)\n", ms * 1e3, E * 2.0 * (128 * 512 + 256 * 128) / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
