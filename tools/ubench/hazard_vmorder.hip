// Do global loads return IN ORDER when an L1-hitting load is issued behind an L1-missing one (gfx950)?
// s_waitcnt vmcnt(N > 0) is only meaningful if they do. (round-2 hunt for the k_compress_bwd_h fault.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/hazard_vmorder.hip -o hazard_vmorder && ./hazard_vmorder
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int NHOT, int TWO_ADDR>
__global__ void k_test(const float* __restrict__ cold, size_t cold_f4, const float* __restrict__ hot,
                       unsigned long long* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; it++) {
        // one cold float4 per lane (a fresh 1 KiB per wave and iteration), then NHOT loads of a 2 KB table
        const size_t ci = ((wave * iters + it) * 64 + lane) % cold_f4;
        const float4* cp = reinterpret_cast<const float4*>(cold) + ci;
        const int hoff = TWO_ADDR ? (lane >> 5) * 16 : (lane & 15) * 16;
        float4 c, h;
        asm volatile(
            "s_mov_b64 s[20:21], %[hot]\n\t"
            "s_nop 4\n\t"
            "global_load_dwordx4 v[24:27], %[cp], off\n\t"
            ".rept %[nhot]\n\t"
            "global_load_dwordx4 v[28:31], %[hoff], s[20:21] offset:256\n\t"
            ".endr\n\t"
            "s_waitcnt vmcnt(%[nhot])\n\t"      // in-order return => the cold load has landed
            "v_mov_b32 %[c0], v24\n\tv_mov_b32 %[c1], v25\n\tv_mov_b32 %[c2], v26\n\tv_mov_b32 %[c3], v27\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            "v_mov_b32 %[h0], v28\n\t"
            : [c0] "=&v"(c.x), [c1] "=&v"(c.y), [c2] "=&v"(c.z), [c3] "=&v"(c.w), [h0] "=&v"(h.x)
            : [cp] "v"(cp), [hot] "s"(hot), [hoff] "v"(hoff), [nhot] "n"(NHOT)
            : "s20", "s21", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "memory");
        const float base = (float)(ci & 0xfffff);
        nbad += (c.x != base) + (c.y != base + 0.25f) + (c.z != base + 0.5f) + (c.w != base + 0.75f);
        if (h.x == -12345.f) nbad += 1000;
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int NHOT, int TWO_ADDR>
static void run(const float* cold, size_t cold_f4, const float* hot, unsigned long long* d_bad) {
    (void)hipMemset(d_bad, 0, 8);
    k_test<NHOT, TWO_ADDR><<<2048, 256>>>(cold, cold_f4, hot, d_bad, 64);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
    printf("cold load + %2d hot loads (%s), s_waitcnt vmcnt(%d): %llu stale components of %llu\n", NHOT,
           TWO_ADDR ? "2 addresses per load" : "16 addresses per load", NHOT, h, 4ull * 2048 * 256 * 64);
}

int main() {
    const size_t cold_f4 = (size_t)1 << 26;  // 1 GiB
    std::vector<float> hc(cold_f4 * 4);
    for (size_t i = 0; i < cold_f4; i++) {
        const float b = (float)(i & 0xfffff);
        hc[4 * i] = b; hc[4 * i + 1] = b + 0.25f; hc[4 * i + 2] = b + 0.5f; hc[4 * i + 3] = b + 0.75f;
    }
    float *cold, *hot; unsigned long long* d_bad;
    (void)hipMalloc(&cold, cold_f4 * 16); (void)hipMalloc(&hot, 4096); (void)hipMalloc(&d_bad, 8);
    (void)hipMemcpy(cold, hc.data(), cold_f4 * 16, hipMemcpyHostToDevice);
    (void)hipMemset(hot, 0, 4096);
    run<1, 1>(cold, cold_f4, hot, d_bad); run<4, 1>(cold, cold_f4, hot, d_bad); run<16, 1>(cold, cold_f4, hot, d_bad);
    run<1, 0>(cold, cold_f4, hot, d_bad); run<4, 0>(cold, cold_f4, hot, d_bad); run<16, 0>(cold, cold_f4, hot, d_bad);
    return 0;
}
