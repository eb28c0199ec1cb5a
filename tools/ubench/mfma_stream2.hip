// why is (dependent MFMA chain + streamed A operand) slow?  variants:
//  0: loads issued+waited each k-group, MFMA reads CONSTANT regs (loaded value only kept alive)
//  1: MFMA reads the loaded regs directly
//  2: loaded regs copied with v_mov before use
//  3: like 1 but weights read from LDS (ds_read_b128) instead of global
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NT, int PF>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ W, float* out, int iters) {
    __shared__ float4 Ws[16 * 64 * 2];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64 * 2; i += 256) Ws[i] = W[i];
    __syncthreads();
    float4 x[16];
    for (int i = 0; i < 16; i++) x[i] = W[threadIdx.x + 256 * i];
    f32x16 acc[NT];
    for (int t = 0; t < NT; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    float4 c = W[lane];
    for (int it = 0; it < iters; it++) {
        const float4* wp = W + (size_t)((it * NT) % 64) * 16 * 64 + lane;
        float4 wb[PF][NT];
#pragma unroll
        for (int s = 0; s < PF; s++)
#pragma unroll
            for (int t = 0; t < NT; t++) wb[s][t] = (MODE == 3) ? Ws[(t * 16 + s) * 64 % 2048 + lane] : wp[(t * 16 + s) * 64];
#pragma unroll
        for (int kg = 0; kg < 16; kg++) {
            const int cur = kg % PF;
            float4 a[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                a[t] = wb[cur][t];
                if (MODE == 0) { asm volatile("" :: "v"(a[t].x), "v"(a[t].y), "v"(a[t].z), "v"(a[t].w)); a[t] = c; }
                if (MODE == 2) { asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %1, %1\n v_mov_b32 %2, %2\n v_mov_b32 %3, %3" : "+v"(a[t].x), "+v"(a[t].y), "+v"(a[t].z), "+v"(a[t].w)); }
            }
            const float4 xv = x[kg];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, xv.x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, xv.y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, xv.z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, xv.w, acc[t], 0, 0, 0);
            if (kg + PF < 16)
#pragma unroll
                for (int t = 0; t < NT; t++) wb[cur][t] = (MODE == 3) ? Ws[((t * 16 + kg + PF) * 64) % 2048 + lane] : wp[(t * 16 + kg + PF) * 64];
        }
    }
    float s = 0.f;
    for (int t = 0; t < NT; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int NT, int PF> void run(const float4* W, float* d, int wg_per_cu) {
    int iters = 200, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NT, PF><<<grid, 256>>>(W, d, 10);
    hipEventRecord(e0); k<MODE, NT, PF><<<grid, 256>>>(W, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode=%d NT=%d PF=%d wg/cu=%d : %.1f TFLOP/s\n", MODE, NT, PF, wg_per_cu, (double)grid * 4 * iters * 64 * NT * 4096.0 / ms / 1e9);
}
int main() {
    float* d; float4* W; hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&W, 2 << 20); hipMemset(W, 0, 2 << 20);
    for (int w = 1; w <= 2; w++) { run<0, 1, 4>(W, d, w); run<1, 1, 4>(W, d, w); run<2, 1, 4>(W, d, w); run<3, 1, 4>(W, d, w); run<3, 2, 2>(W, d, w); run<1, 2, 4>(W, d, w); }
    return 0;
}
