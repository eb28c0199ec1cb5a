// hypothesis test: are the streamed weights an L2 hot spot? give groups of waves their own weight copy
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
template <int NTT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, const float4* __restrict__ W, float* __restrict__ Y, int64_t R, int ncopies, int reps) {
    const RowLane L; const int64_t row0 = wave_row0(); if (row0 >= R) return;
    const int64_t row = row0 + L.r < R ? row0 + L.r : R - 1;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float4* Wc = W + (size_t)(gw % ncopies) * 4096;
    float4 x[16];
    load_rowfrag<16>(x, X, row, 128, L.h);
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll 1
    for (int c = 0; c < 4 / NTT; c++) {
        f32x16 acc[NTT]; acc_zero<NTT>(acc);
        gemm_t<16, NTT, 2>(Wc, 16, 0, NTT * c, x, acc, L.lane);
        float4 y[4 * NTT]; acc_to_frag<NTT>(acc, y); store_rowfrag<4 * NTT>(y, Y + 32 * NTT * c, row, 128, L.h);
    }
    }
}
template <int NTT> void run(const float* X, const float4* W, float* Y, int64_t R, int ncopies, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = (R + 127) / 128;
    k<NTT><<<grid, 256>>>(X, W, Y, R, ncopies, reps);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) k<NTT><<<grid, 256>>>(X, W, Y, R, ncopies, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("NT=%d copies=%4d reps=%d: %.1f us  %.1f TFLOP/s\n", NTT, ncopies, reps, ms * 1e3, 2.0 * R * 128 * 128 * reps / ms / 1e9);
}
int main() {
    int64_t R = 401910; float *X, *Y; float4* W;
    hipMalloc(&X, R * 512); hipMalloc(&Y, R * 512); hipMalloc(&W, 65536 * 1024); hipMemset(X, 0, R * 512); hipMemset(W, 0, 65536 * 1024);
    for (int reps : {1, 4}) for (int nc : {1, 8, 64, 1024}) { run<2>(X, W, Y, R, nc, reps); }
    return 0;
}
