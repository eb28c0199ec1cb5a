// persistent variants of the TRR 128->128 stage: weights streamed from L2 vs held in LDS
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
template <int KGS, int NT>
__device__ __forceinline__ void gemm_lds(const float4* Ws, int kg_total, int tile0, const float4* x, f32x16 (&acc)[NT], int lane) {
#pragma unroll
    for (int kg = 0; kg < KGS; kg++) {
        float4 w[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) w[t] = Ws[((tile0 + t) * kg_total + kg) * 64 + lane];
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
template <bool LDSW, int NTT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, const float4* __restrict__ W, float* __restrict__ Y, int64_t R) {
    extern __shared__ float4 Ws[];
    if (LDSW) { for (int i = threadIdx.x; i < 4096; i += 256) Ws[i] = W[i]; __syncthreads(); }
    const RowLane L;
    const int64_t ntiles = (R + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row0 = tile * 32;
        const int64_t row = row0 + L.r < R ? row0 + L.r : R - 1;
        float4 x[16];
        load_rowfrag<16>(x, X, row, 128, L.h);
#pragma unroll 1
        for (int c = 0; c < 4 / NTT; c++) {
            f32x16 acc[NTT]; acc_zero<NTT>(acc);
            if (LDSW) gemm_lds<16, NTT>(Ws, 16, NTT * c, x, acc, L.lane);
            else gemm_t<16, NTT, 2>(W, 16, 0, NTT * c, x, acc, L.lane);
            float4 y[4 * NTT]; acc_to_frag<NTT>(acc, y); store_rowfrag<4 * NTT>(y, Y + 32 * NTT * c, row, 128, L.h);
        }
    }
}
template <bool LDSW, int NTT> void run(const float* X, const float4* W, float* Y, int64_t R, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    size_t lds = LDSW ? 65536 : 0;
    hipFuncSetAttribute((const void*)k<LDSW, NTT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    k<LDSW, NTT><<<grid, 256, lds>>>(X, W, Y, R);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) k<LDSW, NTT><<<grid, 256, lds>>>(X, W, Y, R);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("weights=%s NT=%d grid=%d: %.1f us  %.1f TFLOP/s\n", LDSW ? "LDS" : "L2 ", NTT, grid, ms * 1e3, 2.0 * R * 128 * 128 / ms / 1e9);
}
int main() {
    int64_t R = 401910; float *X, *Y; float4* W;
    hipMalloc(&X, R * 512); hipMalloc(&Y, R * 512); hipMalloc(&W, 65536); hipMemset(X, 0, R * 512); hipMemset(W, 0, 65536);
    for (int grid : {256, 512, 1024, 3140}) { run<false, 2>(X, W, Y, R, grid); run<true, 2>(X, W, Y, R, grid); run<false, 4>(X, W, Y, R, grid); run<true, 4>(X, W, Y, R, grid); }
    return 0;
}
