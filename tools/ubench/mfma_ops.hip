// MFMA 32x32x2 f32 throughput with rotating source registers (like a real GEMM) vs constant operands,
// and with a concurrent trickle of global loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ W, float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float4 a[16], b[16];
    for (int i = 0; i < 16; i++) { a[i] = W[threadIdx.x + 256 * i]; b[i] = W[threadIdx.x + 256 * (i + 16)]; }
    const float4* wp = W + (threadIdx.x & 63);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int kg = 0; kg < 16; kg++) {
            float4 av = a[kg];
            if (MODE == 2) av = wp[(it * 16 + kg) % 1024 * 64];   // streamed weights (L2-resident 1 MiB)
            const float4 bv = (MODE == 0) ? b[0] : b[kg];
            if (MODE == 0) av = a[0];
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int NACC> void run(const float4* W, float* d, int wg_per_cu) {
    int iters = 200, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NACC><<<grid, 256>>>(W, d, 10);
    hipEventRecord(e0); k<MODE, NACC><<<grid, 256>>>(W, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 64 * NACC * 4096.0;
    printf("mode=%d (0 const ops, 1 rotating regs, 2 streamed A) NACC=%d wg/cu=%d : %.1f TFLOP/s\n", MODE, NACC, wg_per_cu, flops / ms / 1e9);
}
int main() {
    float* d; float4* W; hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&W, 1 << 21); hipMemset(W, 0, 1 << 21);
    for (int w = 1; w <= 2; w++) { run<0, 2>(W, d, w); run<1, 2>(W, d, w); run<2, 2>(W, d, w); run<1, 1>(W, d, w); run<2, 1>(W, d, w); run<2, 4>(W, d, w); }
    return 0;
}
