// in-kernel timeline (s_memtime) of the TRR 128->128 stage for a few waves
#include "../../metatrain_amd/csrc/trr.h"
#include <stdio.h>
#include <vector>
#include <algorithm>
namespace pet { void set_error(const std::string&) {} }
using namespace pet;
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, const float4* __restrict__ W, float* __restrict__ Y, int64_t R,
                                         unsigned long long* T) {
    const RowLane L; const int64_t row0 = wave_row0(); if (row0 >= R) return;
    const int64_t row = row0 + L.r < R ? row0 + L.r : R - 1;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned long long t0 = now();
    float4 x[16];
    load_rowfrag<16>(x, X, row, 128, L.h);
    float s = 0; for (int i = 0; i < 16; i++) s += x[i].x;   // force the wait
    asm volatile("" :: "v"(s));
    unsigned long long t1 = now();
    unsigned long long tc[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        f32x16 acc[2]; acc_zero<2>(acc);
        gemm_t<16, 2, 2>(W, 16, 0, 2 * c, x, acc, L.lane);
        float4 y[8]; acc_to_frag<2>(acc, y);
        asm volatile("" :: "v"(y[7].w));
        tc[c] = now();
        store_rowfrag<8>(y, Y + 64 * c, row, 128, L.h);
    }
    unsigned long long t3 = now();
    if (L.lane == 0) { T[gw * 5 + 0] = t0; T[gw * 5 + 1] = t1; T[gw * 5 + 2] = tc[0]; T[gw * 5 + 3] = tc[1]; T[gw * 5 + 4] = t3; }
}
int main() {
    int64_t R = 401910; float *X, *Y; float4* W; unsigned long long* T;
    int grid = (R + 127) / 128; int nw = grid * 4;
    hipMalloc(&X, R * 512); hipMalloc(&Y, R * 512); hipMalloc(&W, 65536); hipMalloc(&T, nw * 40);
    hipMemset(X, 0, R * 512); hipMemset(W, 0, 65536); hipMemset(T, 0, nw * 40);
    k<<<grid, 256>>>(X, W, Y, R, T); k<<<grid, 256>>>(X, W, Y, R, T); hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw * 5); hipMemcpy(h.data(), T, nw * 40, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull, tmax = 0; for (int w = 0; w < nw; w++) { if (!h[w*5]) continue; tmin = std::min(tmin, h[w * 5]); tmax = std::max(tmax, h[w * 5 + 4]); }
    printf("kernel span %.1f us (memtime ticks @100MHz?) raw %llu\n", (tmax - tmin) / 100.0, tmax - tmin);
    double a[4] = {0, 0, 0, 0}; int n = 0;
    for (int w = 0; w < nw; w++) { if (!h[w*5]) continue; for (int i = 0; i < 4; i++) a[i] += h[w * 5 + i + 1] - h[w * 5 + i]; n++; }
    printf("avg ticks: load %.1f  chunk0 %.1f  chunk1 %.1f  store-issue %.1f  (n=%d)\n", a[0] / n, a[1] / n, a[2] / n, a[3] / n, n);
    for (int w : {0, 1, 2, 3, 4000, 4001, 9000, 12000}) printf("wave %5d start %8llu: load %5llu c0 %5llu c1 %5llu st %5llu\n", w, h[w*5]-tmin, h[w*5+1]-h[w*5], h[w*5+2]-h[w*5+1], h[w*5+3]-h[w*5+2], h[w*5+4]-h[w*5+3]);
    return 0;
}
