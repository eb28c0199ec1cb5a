// hypothesis test: fragment-shaped (32 B/row) vs tile-major (1 KiB contiguous per instruction) activations
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
template <bool TILEMAJOR, int NTT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, const float4* __restrict__ W, float* __restrict__ Y, int64_t R) {
    const RowLane L; const int64_t row0 = wave_row0(); if (row0 >= R) return;
    const int64_t row = row0 + L.r < R ? row0 + L.r : R - 1;
    float4 x[16];
    if (TILEMAJOR) { const float4* p = (const float4*)X + (row0 / 32) * 16 * 64 + L.lane;
#pragma unroll
        for (int kg = 0; kg < 16; kg++) x[kg] = p[kg * 64]; }
    else load_rowfrag<16>(x, X, row, 128, L.h);
#pragma unroll 1
    for (int c = 0; c < 4 / NTT; c++) {
        f32x16 acc[NTT]; acc_zero<NTT>(acc);
        gemm_t<16, NTT, 2>(W, 16, 0, NTT * c, x, acc, L.lane);
        float4 y[4 * NTT]; acc_to_frag<NTT>(acc, y);
        if (TILEMAJOR) { float4* q = (float4*)Y + ((row0 / 32) * 16 + 4 * NTT * c) * 64 + L.lane;
#pragma unroll
            for (int kg = 0; kg < 4 * NTT; kg++) q[kg * 64] = y[kg]; }
        else store_rowfrag<4 * NTT>(y, Y + 32 * NTT * c, row, 128, L.h);
    }
}
template <bool TM, int NTT> void run(const float* X, const float4* W, float* Y, int64_t R) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = (R + 127) / 128;
    k<TM, NTT><<<grid, 256>>>(X, W, Y, R);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) k<TM, NTT><<<grid, 256>>>(X, W, Y, R);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%s NT=%d: %.1f us  %.1f TFLOP/s\n", TM ? "tile-major" : "row-major ", NTT, ms * 1e3, 2.0 * R * 128 * 128 / ms / 1e9);
}
int main() {
    int64_t R = 401920; float *X, *Y; float4* W;
    hipMalloc(&X, R * 512); hipMalloc(&Y, R * 512); hipMalloc(&W, 65536); hipMemset(X, 0, R * 512); hipMemset(W, 0, 65536);
    run<false, 2>(X, W, Y, R); run<true, 2>(X, W, Y, R); run<false, 4>(X, W, Y, R); run<true, 4>(X, W, Y, R);
    run<false, 1>(X, W, Y, R); run<true, 1>(X, W, Y, R);
    return 0;
}
