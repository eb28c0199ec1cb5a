// access-pattern test for attention operands: row-major [row][384] (64-B slices per head at 1.5 KB stride)
// vs head-major [head][row][48] (contiguous per (atom, head)); T = 20 tokens per atom, 8 heads.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <bool HEADMAJOR>
__global__ __launch_bounds__(256) void k(const float* __restrict__ QKV, float* __restrict__ out, long R, int natoms) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int atom = gw >> 3, head = gw & 7;
    if (atom >= natoms) return;
    const int c16 = lane & 15, g4 = lane >> 4;
    const long start = (long)atom * 20;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; t++) {
        long row = start + 16 * t + c16; if (16 * t + c16 >= 20) row = start;
#pragma unroll
        for (int part = 0; part < 3; part++) {   // Q, K, V slices
            const float4 v = HEADMAJOR ? *(const float4*)(QKV + ((long)head * R + row) * 48 + 16 * part + 4 * g4)
                                       : *(const float4*)(QKV + row * 384 + 128 * part + 16 * head + 4 * g4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    // write a 64-B slice per token (like dQ) 
#pragma unroll
    for (int t = 0; t < 2; t++) {
        long row = start + 16 * t + c16;
        if (16 * t + c16 < 20) {
            if (HEADMAJOR) *(float4*)(out + ((long)head * R + row) * 48 + 4 * g4) = acc;
            else *(float4*)(out + row * 384 + 16 * head + 4 * g4) = acc;
        }
    }
}
int main() {
    int natoms = 20000; long R = (long)natoms * 20;
    float *a, *b; hipMalloc(&a, R * 1536); hipMalloc(&b, R * 1536); hipMemset(a, 0, R * 1536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = natoms * 8 / 4;
    for (int v = 0; v < 2; v++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) { if (v) k<true><<<grid, 256>>>(a, b, R, natoms); else k<false><<<grid, 256>>>(a, b, R, natoms); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%s: %.1f us  read %.2f TB/s (useful bytes)\n", v ? "head-major" : "row-major ", ms * 1e3, R * 1536.0 / ms / 1e9);
    }
    return 0;
}
