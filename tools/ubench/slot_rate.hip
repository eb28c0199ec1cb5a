// What one "slot" of the software-pipelined edge-MLP kernels costs with nothing but its skeleton: three dependent-free
// or dependent v_mfma_f32_32x32x16_f16 followed by NV VALU instructions, sched_barrier after every slot, one wave per
// SIMD (160 KB of dynamic LDS keep a second workgroup off the CU). Prints shader-clock cycles per slot (s_memtime).
//   PAT 0: four accumulator pairs in turn (the dn / out GEMM pattern)   PAT 1: one pair, every slot (the du pattern)
//   PAT 2: two pairs alternating (the [v; g] pattern)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/slot_rate.hip -o tools/ubench/slot_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)
template <int PAT, int NV, bool TRANS>
__global__ __launch_bounds__(256) void k_slots(float* out, long long* cyc, int iters) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * (lane - j)); }
    f32x16 acc[4], acl[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) { acc[t][r] = 0.f; acl[t][r] = 0.f; }
    float v[8];
    for (int j = 0; j < 8; j++) v[j] = 1.0f + 0.01f * (lane + j);
    const long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 24; s++) {
            const int t = PAT == 0 ? (s & 3) : PAT == 1 ? 0 : (s & 1);
            acl[t] = MFMA(a, b, acl[t]);
            acc[t] = MFMA(b, a, acc[t]);
            acl[t] = MFMA(b, b, acl[t]);
#pragma unroll
            for (int j = 0; j < NV; j++) {
                if (TRANS && (j & 7) == 7) v[j & 7] = __builtin_amdgcn_rcpf(v[j & 7]);
                else v[j & 7] = v[j & 7] * 1.0001f + 0.5f;
            }
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float sum = 0.f;
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) sum += acc[t][r] + acl[t][r];
    for (int j = 0; j < 8; j++) sum += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = sum + lds[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int PAT, int NV, bool TRANS>
static void run(const char* what, float* out, long long* cyc) {
    const int iters = 2000;
    const size_t lds = 160 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_slots<PAT, NV, TRANS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_slots<PAT, NV, TRANS><<<256, 256, lds>>>(out, cyc, 10);
    (void)hipEventRecord(e0);
    k_slots<PAT, NV, TRANS><<<256, 256, lds>>>(out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c = 0; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %7.1f clock64 ticks per slot, %7.1f ns per slot (96 matrix-pipe cycles of work)\n", what, (double)c / (iters * 24.0),
           ms * 1e6 / (iters * 24.0));
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    run<0, 0, false>("4 pairs in turn, no VALU", out, cyc);
    run<0, 8, false>("4 pairs in turn, 8 VALU", out, cyc);
    run<0, 16, false>("4 pairs in turn, 16 VALU", out, cyc);
    run<0, 24, false>("4 pairs in turn, 24 VALU", out, cyc);
    run<0, 16, true>("4 pairs in turn, 16 VALU (2 rcp)", out, cyc);
    run<2, 0, false>("2 pairs alternating, no VALU", out, cyc);
    run<2, 16, false>("2 pairs alternating, 16 VALU", out, cyc);
    run<1, 0, false>("1 pair, no VALU", out, cyc);
    run<1, 16, false>("1 pair, 16 VALU", out, cyc);
    return 0;
}
