// microbenchmark: bandwidth of row-fragment (32 B per row per instruction) vs full-line access
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_frag(const float* __restrict__ X, float* __restrict__ Y, long R) {
    int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (row0 >= R) return;
    long row = row0 + r < R ? row0 + r : R - 1;
    float4 x[16];
    const float4* p = (const float4*)(X + row * 128 + 4 * h);
#pragma unroll
    for (int kg = 0; kg < 16; kg++) x[kg] = p[2 * kg];
    float4* q = (float4*)(Y + row * 128 + 4 * h);
#pragma unroll
    for (int kg = 0; kg < 16; kg++) { x[kg].x += 1.f; q[2 * kg] = x[kg]; }
}
__global__ __launch_bounds__(256) void k_line(const float* __restrict__ X, float* __restrict__ Y, long R) {
    int lane = threadIdx.x & 63;
    long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (row0 >= R) return;
    float4 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {  // 2 rows (1 KB) per instruction
        long row = row0 + 2 * i + (lane >> 5); if (row >= R) row = R - 1;
        x[i] = ((const float4*)(X + row * 128))[lane & 31];
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        long row = row0 + 2 * i + (lane >> 5); if (row >= R) row = R - 1;
        x[i].x += 1.f; ((float4*)(Y + row * 128))[lane & 31] = x[i];
    }
}
int main() {
    long R = 400000; float *X, *Y;
    hipMalloc(&X, R * 512); hipMalloc(&Y, R * 512); hipMemset(X, 0, R * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = (R + 127) / 128;
    for (int v = 0; v < 2; v++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) { if (v == 0) k_frag<<<grid, 256>>>(X, Y, R); else k_line<<<grid, 256>>>(X, Y, R); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us/launch, %.2f TB/s (r+w)\n", v == 0 ? "fragment" : "full-line", ms * 100, 2.0 * R * 512 * 10 / ms / 1e9);
        }
    }
    return 0;
}
