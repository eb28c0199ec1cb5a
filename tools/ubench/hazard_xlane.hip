// Which producer -> cross-lane-consumer adjacencies need wait states on gfx950 that hipcc (ROCm 7.2) does not insert?
// Round-2 finding: k_compress_bwd_h returned row sums with the LAST addend missing for whole waves (only in workgroups
// dispatched after the first resident batch: warm caches, no stalls to hide the window). Every sequence below is
// written in inline asm on fixed registers, so nothing is padded or reordered by the compiler.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/hazard_xlane.hip -o hazard_xlane && ./hazard_xlane
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)

// producer: v_pk_add_f32 v[10:11] = v[10:11] + v[12:13]; then NOPS wait states; then the consumer reads v10 / v11
#define PROLOGUE "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\tv_mov_b32 v16, %6\n\ts_nop 7\n\t"
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17"

template <int MODE, int NOPS>
__device__ __forceinline__ void seq(float a0, float a1, float b0, float b1, int addr, float& r0, float& r1) {
    if (MODE == 0) {  // packed add -> ds_bpermute
        asm volatile(PROLOGUE "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n\t"
                     ".rept %7\n\ts_nop 0\n\t.endr\n\t"
                     "ds_bpermute_b32 v14, v16, v10\n\tds_bpermute_b32 v15, v16, v11\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_mov_b32 %0, v14\n\tv_mov_b32 %1, v15\n\t"
                     : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(addr), "n"(NOPS) : CLOB);
    } else if (MODE == 1) {  // plain adds -> ds_bpermute
        asm volatile(PROLOGUE "v_add_f32 v10, v10, v12\n\tv_add_f32 v11, v11, v13\n\t"
                     ".rept %7\n\ts_nop 0\n\t.endr\n\t"
                     "ds_bpermute_b32 v14, v16, v10\n\tds_bpermute_b32 v15, v16, v11\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_mov_b32 %0, v14\n\tv_mov_b32 %1, v15\n\t"
                     : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(addr), "n"(NOPS) : CLOB);
    } else if (MODE == 2) {  // packed add (2 wait states, the documented pre-swap pad) -> permlane32_swap -> NOPS -> consumer
        asm volatile(PROLOGUE "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n\tv_mov_b32 v14, v10\n\ts_nop 1\n\t"
                     "v_permlane32_swap_b32 v10, v14\n\t"
                     ".rept %7\n\ts_nop 0\n\t.endr\n\t"
                     "v_add_f32 %0, v10, v14\n\tv_mov_b32 %1, v11\n\t"  // r0 = lo + hi of (a0 + b0)
                     : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(addr), "n"(NOPS) : CLOB);
    } else if (MODE == 3) {  // packed add -> NOPS -> permlane32_swap (pre-swap window), consumer after 4 states
        asm volatile(PROLOGUE "v_pk_add_f32 v[10:11], v[10:11], v[12:13]\n\t"
                     ".rept %7\n\ts_nop 0\n\t.endr\n\t"
                     "v_mov_b32 v14, v10\n\t"
                     ".rept %7\n\ts_nop 0\n\t.endr\n\t"
                     "v_permlane32_swap_b32 v10, v14\n\ts_nop 3\n\t"
                     "v_add_f32 %0, v10, v14\n\tv_mov_b32 %1, v11\n\t"
                     : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(addr), "n"(NOPS) : CLOB);
    } else if (MODE == 4) {  // packed fma chain end -> ds_bpermute of the SECOND half only
        asm volatile(PROLOGUE "v_pk_fma_f32 v[10:11], v[10:11], v[12:13], v[12:13]\n\t"
                     ".rept %7\n\ts_nop 0\n\t.endr\n\t"
                     "ds_bpermute_b32 v15, v16, v11\n\tds_bpermute_b32 v14, v16, v10\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_mov_b32 %0, v14\n\tv_mov_b32 %1, v15\n\t"
                     : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(addr), "n"(NOPS) : CLOB);
    }
}

template <int MODE, int NOPS>
__global__ void k_test(const float* __restrict__ in, unsigned long long* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    const int addr = ((lane ^ 32) << 2);
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nbad = 0;
    float a0 = in[gid & 65535], a1 = in[(gid + 77) & 65535];
    for (int it = 0; it < iters; it++) {
        const float b0 = in[(gid + 131 * it + 1) & 65535], b1 = in[(gid + 257 * it + 5) & 65535];
        float r0, r1;
        seq<MODE, NOPS>(a0, a1, b0, b1, addr, r0, r1);
        float e0, e1;  // expected, through the compiler's own (padded) code
        if (MODE == 0 || MODE == 1) {
            e0 = __shfl_xor(a0 + b0, 32); e1 = __shfl_xor(a1 + b1, 32);
        } else if (MODE == 4) {
            e0 = __shfl_xor(fmaf(a0, b0, b0), 32); e1 = __shfl_xor(fmaf(a1, b1, b1), 32);
        } else {
            const float s = a0 + b0;
            e0 = s + __shfl_xor(s, 32);  // lo + hi (commutative: same bits in both halves)
            e1 = a1 + b1;
        }
        nbad += (__float_as_uint(r0) != __float_as_uint(e0)) + (__float_as_uint(r1) != __float_as_uint(e1));
        a0 = b0 * 0.5f + 0.25f; a1 = b1 * 0.5f - 0.125f;
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int MODE, int NOPS>
static void run(const float* d_in, unsigned long long* d_bad, const char* what) {
    hipMemset(d_bad, 0, 8);
    k_test<MODE, NOPS><<<8192, 256>>>(d_in, d_bad, 64);
    unsigned long long h = 0;
    hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
    printf("%-58s wait states %d : %llu mismatches of %llu\n", what, NOPS, h, 2ull * 8192 * 256 * 64);
}

int main() {
    std::vector<float> h(65536);
    for (int i = 0; i < 65536; i++) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.0f - 0.5f;
    float* d_in; unsigned long long* d_bad;
    hipMalloc(&d_in, 65536 * 4); hipMalloc(&d_bad, 8);
    hipMemcpy(d_in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
#define RUN4(M, W) run<M, 0>(d_in, d_bad, W); run<M, 1>(d_in, d_bad, W); run<M, 2>(d_in, d_bad, W); run<M, 3>(d_in, d_bad, W); run<M, 4>(d_in, d_bad, W);
    RUN4(0, "v_pk_add_f32 -> ds_bpermute_b32 (data operand)")
    RUN4(1, "v_add_f32 x2 -> ds_bpermute_b32 (data operand)")
    RUN4(4, "v_pk_fma_f32 -> ds_bpermute_b32 (second half first)")
    RUN4(3, "v_pk_add_f32 -> v_mov -> v_permlane32_swap (pre-swap pad)")
    RUN4(2, "v_permlane32_swap -> v_add_f32 reading both operands")
    return 0;
}
