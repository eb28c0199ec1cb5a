// streamed-weight MFMA throughput vs explicit prefetch depth PF (k-groups in flight) and accumulators NT
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
template <int NT, int PF>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ W, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float4 x[16];
    for (int i = 0; i < 16; i++) x[i] = W[threadIdx.x + 256 * i];
    f32x16 acc[NT]; acc_zero<NT>(acc);
    for (int it = 0; it < iters; it++) gemm_t<16, NT, PF>(W, 16, 0, (it * NT) % 64, x, acc, lane);   // 64 tiles x 16 kg x 1 KiB = 1 MiB of weights
    float s = 0.f;
    for (int i = 0; i < NT; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NT, int PF> void run(const float4* W, float* d, int wg_per_cu) {
    int iters = 200, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NT, PF><<<grid, 256>>>(W, d, 10);
    hipEventRecord(e0); k<NT, PF><<<grid, 256>>>(W, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("NT=%d PF=%2d wg/cu=%d : %.1f TFLOP/s\n", NT, PF, wg_per_cu, (double)grid * 4 * iters * 64 * NT * 4096.0 / ms / 1e9);
}
int main() {
    float* d; float4* W; hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&W, 2 << 20); hipMemset(W, 0, 2 << 20);
    for (int w = 1; w <= 2; w++) {
        run<1, 2>(W, d, w); run<1, 4>(W, d, w); run<1, 8>(W, d, w); run<1, 12>(W, d, w);
        run<2, 2>(W, d, w); run<2, 4>(W, d, w); run<2, 8>(W, d, w);
        run<4, 2>(W, d, w); run<4, 4>(W, d, w);
    }
    return 0;
}
