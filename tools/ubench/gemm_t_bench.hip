// microbenchmark of the TRR 128->128 stage: what limits it? (weights stream, row loads, stores)
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
template <int MODE, int NTT, int PFF>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, const float4* __restrict__ W, float* __restrict__ Y, int64_t R, int reps) {
    const RowLane L; const int64_t row0 = wave_row0(); if (row0 >= R) return;
    const int64_t row = row0 + L.r < R ? row0 + L.r : R - 1;
    float4 x[16];
    if (MODE & 1) load_rowfrag<16>(x, X, row, 128, L.h);
    else { for (int k = 0; k < 16; k++) x[k] = make_float4(1.f, 2.f, 3.f, 4.f); }
    for (int rep = 0; rep < reps; rep++) {
#pragma unroll 1
        for (int c = 0; c < 4 / NTT; c++) {
            f32x16 acc[NTT]; acc_zero<NTT>(acc);
            gemm_t<16, NTT, PFF>(W, 16, 0, NTT * c, x, acc, L.lane);
            if (MODE & 2) { float4 y[4 * NTT]; acc_to_frag<NTT>(acc, y); store_rowfrag<4 * NTT>(y, Y + 32 * NTT * c, row, 128, L.h); }
            else if (acc[0][0] == 123.456f) Y[0] = 1.f;
        }
    }
}
template <int MODE, int NTT, int PFF> void run(const float* X, const float4* W, float* Y, int64_t R, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = (R + 127) / 128;
    k<MODE, NTT, PFF><<<grid, 256>>>(X, W, Y, R, reps);
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) k<MODE, NTT, PFF><<<grid, 256>>>(X, W, Y, R, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("mode(load=%d,store=%d) NT=%d PF=%d reps=%d: %.1f us  %.1f TFLOP/s\n", MODE & 1, (MODE >> 1) & 1, NTT, PFF, reps, ms * 1e3, 2.0 * R * 128 * 128 * reps / ms / 1e9);
}
int main() {
    int64_t R = 401910; float *X, *Y; float4* W;
    hipMalloc(&X, R * 512); hipMalloc(&Y, R * 512); hipMalloc(&W, 65536); hipMemset(X, 0, R * 512); hipMemset(W, 0, 65536);
    run<0, 2, 2>(X, W, Y, R, 1); run<0, 2, 2>(X, W, Y, R, 8);
    run<1, 2, 2>(X, W, Y, R, 1); run<3, 2, 2>(X, W, Y, R, 1); run<3, 2, 4>(X, W, Y, R, 1);
    run<3, 4, 2>(X, W, Y, R, 1); run<3, 4, 3>(X, W, Y, R, 1); run<0, 4, 2>(X, W, Y, R, 8); run<3, 1, 6>(X, W, Y, R, 1);
    run<3, 2, 2>(X, W, Y, R, 4); run<3, 4, 2>(X, W, Y, R, 4);
    return 0;
}
