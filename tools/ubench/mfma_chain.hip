// microbenchmark: fp32 MFMA 32x32x2 throughput vs number of independent accumulators and waves/SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu, float* d) {
    int iters = 2000;
    int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(d, 10, 1.f, 1.f);
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(d, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d wg/cu=%d : %.1f TFLOP/s (%.3f ms)\n", NACC, wg_per_cu, flops / ms / 1e9, ms);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 2; w++) { run<1>(w, d); run<2>(w, d); run<4>(w, d); }
    return 0;
}
