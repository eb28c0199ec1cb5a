// fp32 GEMM on the bf16 matrix cores by 3-way splitting (x = hi + mid + lo, 8 + 8 + 8 mantissa bits):
// accuracy against fp64 and throughput against the fp32 MFMA path, same TRR operand layout.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}
// W [N][K] fp32 -> three fragment-ordered bf16x8 arrays; index (tile * (K/16) + kb) * 64 + lane
__global__ void k_pack3(const float* W, int N, int K, bf16x8* Wh, bf16x8* Wm, bf16x8* Wl) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int kbn = K / 16;
    if (idx >= (N / 32) * kbn * 64) return;
    const int lane = idx & 63, kb = (idx >> 6) % kbn, tile = (idx >> 6) / kbn;
    const int n = 32 * tile + (lane & 31), g = lane >> 5;
    bf16x8 h, m, l;
    for (int j = 0; j < 8; j++) {
        const int k = 16 * kb + (j < 4 ? 4 * g + j : 8 + 4 * g + (j - 4));
        __bf16 a, b, c;
        split3(W[(size_t)n * K + k], a, b, c);
        h[j] = a; m[j] = b; l[j] = c;
    }
    Wh[idx] = h; Wm[idx] = m; Wl[idx] = l;
}
__global__ void k_pack_f32(const float* W, int N, int K, float4* Wp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int kgn = K / 8;
    if (idx >= (N / 32) * kgn * 64) return;
    const int lane = idx & 63, kg = (idx >> 6) % kgn, tile = (idx >> 6) / kgn;
    const float* p = W + (size_t)(32 * tile + (lane & 31)) * K + 8 * kg + 4 * (lane >> 5);
    Wp[idx] = make_float4(p[0], p[1], p[2], p[3]);
}

template <int K, int NT, int TERMS>
__global__ __launch_bounds__(256) void k_bf16(const float* __restrict__ X, const bf16x8* __restrict__ Wh,
                                              const bf16x8* __restrict__ Wm, const bf16x8* __restrict__ Wl,
                                              float* __restrict__ Y, int64_t R, int N, int reps) {
    const bool store = reps == 1;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (row0 >= R) return;
    const int64_t row = row0 + (lane & 31) < R ? row0 + (lane & 31) : R - 1;
    const int g = lane >> 5;
    bf16x8 xh[K / 16], xm[K / 16], xl[K / 16];
#pragma unroll
    for (int kb = 0; kb < K / 16; kb++) {
        const float4 a = *reinterpret_cast<const float4*>(X + row * K + 16 * kb + 4 * g);
        const float4 b = *reinterpret_cast<const float4*>(X + row * K + 16 * kb + 8 + 4 * g);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; j++) { __bf16 h, m, l; split3(v[j], h, m, l); xh[kb][j] = h; xm[kb][j] = m; xl[kb][j] = l; }
    }
    for (int rep = 0; rep < reps; rep++)
    for (int t0 = 0; t0 < N / 32; t0 += NT) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < K / 16; kb++) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const size_t wi = ((size_t)(t0 + t) * (K / 16) + kb) * 64 + lane;
                const bf16x8 wh = Wh[wi], wm = Wm[wi], wl = Wl[wi];
                if (TERMS >= 6) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[kb], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[kb], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xm[kb], acc[t], 0, 0, 0);
                }
                if (TERMS >= 3) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh[kb], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xm[kb], acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[kb], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = 32 * (t0 + t) + (r & 3) + 8 * (r >> 2) + 4 * g;   // C row = output feature
                if (store ? row0 + (lane & 31) < R : acc[t][r] == 123.456f) Y[(row0 + (lane & 31)) * N + n] = acc[t][r];
            }
    }
}
template <int K, int NT>
__global__ __launch_bounds__(256) void k_f32(const float* __restrict__ X, const float4* __restrict__ Wp,
                                             float* __restrict__ Y, int64_t R, int N, int reps) {
    const bool store = reps == 1;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (row0 >= R) return;
    const int64_t row = row0 + (lane & 31) < R ? row0 + (lane & 31) : R - 1;
    const int g = lane >> 5;
    float4 x[K / 8];
#pragma unroll
    for (int kg = 0; kg < K / 8; kg++) x[kg] = *reinterpret_cast<const float4*>(X + row * K + 8 * kg + 4 * g);
    for (int rep = 0; rep < reps; rep++)
    for (int t0 = 0; t0 < N / 32; t0 += NT) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
#pragma unroll
        for (int kg = 0; kg < K / 8; kg++) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float4 w = Wp[((size_t)(t0 + t) * (K / 8) + kg) * 64 + lane];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, x[kg].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, x[kg].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, x[kg].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, x[kg].w, acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = 32 * (t0 + t) + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (store ? row0 + (lane & 31) < R : acc[t][r] == 123.456f) Y[(row0 + (lane & 31)) * N + n] = acc[t][r];
            }
    }
}
// bf16x6 with the weight fragments of NT tiles staged in LDS once per workgroup (4 waves share them),
// double buffered: global -> LDS for step i+1 is issued before the MFMAs of step i.
template <int K, int NT>
__global__ __launch_bounds__(256) void k_bf16_lds(const float* __restrict__ X, const bf16x8* __restrict__ Wh,
                                                  const bf16x8* __restrict__ Wm, const bf16x8* __restrict__ Wl,
                                                  float* __restrict__ Y, int64_t R, int N, int reps) {
    constexpr int KB = K / 16, FR = NT * KB * 64;  // fragments (bf16x8) per split per step
    extern __shared__ __attribute__((aligned(16))) bf16x8 sm[];  // [2][3][FR]
    const bool store = reps == 1;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    const int64_t row = row0 + (lane & 31) < R ? row0 + (lane & 31) : R - 1;
    const int g = lane >> 5;
    bf16x8 xh[KB], xm[KB], xl[KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        const float4 a = *reinterpret_cast<const float4*>(X + row * K + 16 * kb + 4 * g);
        const float4 b = *reinterpret_cast<const float4*>(X + row * K + 16 * kb + 8 + 4 * g);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; j++) { __bf16 h, m, l; split3(v[j], h, m, l); xh[kb][j] = h; xm[kb][j] = m; xl[kb][j] = l; }
    }
    const int steps = reps * (N / 32 / NT);
    auto stage = [&](int step, int buf) {
        const int t0 = (step % (N / 32 / NT)) * NT;
        for (int i = threadIdx.x; i < FR; i += 256) {
            const size_t src = (size_t)t0 * KB * 64 + i;
            sm[(buf * 3 + 0) * FR + i] = Wh[src];
            sm[(buf * 3 + 1) * FR + i] = Wm[src];
            sm[(buf * 3 + 2) * FR + i] = Wl[src];
        }
    };
    stage(0, 0);
    __syncthreads();
    for (int step = 0; step < steps; step++) {
        const int buf = step & 1;
        if (step + 1 < steps) stage(step + 1, buf ^ 1);
        const bf16x8* wh = sm + (buf * 3 + 0) * FR;
        const bf16x8* wm = sm + (buf * 3 + 1) * FR;
        const bf16x8* wl = sm + (buf * 3 + 2) * FR;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int wi = (t * KB + kb) * 64 + lane;
                const bf16x8 a = wh[wi], b = wm[wi], c = wl[wi];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, xh[kb], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xl[kb], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, xm[kb], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, xh[kb], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xm[kb], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xh[kb], acc[t], 0, 0, 0);
            }
        }
        const int t0 = (step % (N / 32 / NT)) * NT;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = 32 * (t0 + t) + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (store ? row0 + (lane & 31) < R : acc[t][r] == 123.456f) Y[(row0 + (lane & 31)) * N + n] = acc[t][r];
            }
        __syncthreads();
    }
}

static double maxrel(const std::vector<float>& y, const std::vector<double>& ref) {
    double e = 0, s = 0;
    for (size_t i = 0; i < ref.size(); i++) { e = fmax(e, fabs(y[i] - ref[i])); s = fmax(s, fabs(ref[i])); }
    return e / s;
}
int main() {
    constexpr int K = 128, N = 512;
    const int64_t R = 401910, RC = 4096;  // RC rows are checked against fp64
    std::vector<float> hX(R * K), hW((size_t)N * K);
    srand(1);
    for (auto& v : hX) v = (float)((rand() / (double)RAND_MAX) * 4 - 2);
    for (auto& v : hW) v = (float)(((rand() / (double)RAND_MAX) * 2 - 1) / sqrt((double)K));
    std::vector<double> ref(RC * N);
    for (int64_t r = 0; r < RC; r++)
        for (int n = 0; n < N; n++) { double s = 0; for (int k = 0; k < K; k++) s += (double)hX[r * K + k] * hW[(size_t)n * K + k]; ref[r * N + n] = s; }
    float *X, *W, *Y; float4* Wp; bf16x8 *Wh, *Wm, *Wl;
    hipMalloc(&X, R * K * 4); hipMalloc(&W, N * K * 4); hipMalloc(&Y, R * N * 4); hipMalloc(&Wp, N * K * 4);
    hipMalloc(&Wh, N * K * 2); hipMalloc(&Wm, N * K * 2); hipMalloc(&Wl, N * K * 2);
    hipMemcpy(X, hX.data(), R * K * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), N * K * 4, hipMemcpyHostToDevice);
    k_pack3<<<(N / 32 * K / 16 * 64 + 255) / 256, 256>>>(W, N, K, Wh, Wm, Wl);
    k_pack_f32<<<(N / 32 * K / 8 * 64 + 255) / 256, 256>>>(W, N, K, Wp);
    const int grid = (int)((R + 127) / 128);
    std::vector<float> hY(RC * N);
    auto check = [&](const char* name) { hipMemcpy(hY.data(), Y, RC * N * 4, hipMemcpyDeviceToHost); printf("%-18s max|err|/max|ref| = %.3e\n", name, maxrel(hY, ref)); };
    k_f32<K, 2><<<grid, 256>>>(X, Wp, Y, R, N, 1); check("fp32 mfma");
    k_bf16<K, 2, 1><<<grid, 256>>>(X, Wh, Wm, Wl, Y, R, N, 1); check("bf16 x1");
    k_bf16<K, 2, 3><<<grid, 256>>>(X, Wh, Wm, Wl, Y, R, N, 1); check("bf16 x3 terms");
    k_bf16<K, 2, 6><<<grid, 256>>>(X, Wh, Wm, Wl, Y, R, N, 1); check("bf16 x6 terms");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        launch(); hipEventRecord(e0); for (int i = 0; i < 5; i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-26s %8.1f us  %7.1f TFLOP/s (fp32-equivalent)\n", name, ms * 1e3, 2.0 * R * K * N * 4 / ms / 1e9);
    };
    timeit("fp32 mfma NT=2 reps=4", [&] { k_f32<K, 2><<<grid, 256>>>(X, Wp, Y, R, N, 4); });
    timeit("fp32 mfma NT=4 reps=4", [&] { k_f32<K, 4><<<grid, 256>>>(X, Wp, Y, R, N, 4); });
    timeit("bf16x6 NT=2 reps=4", [&] { k_bf16<K, 2, 6><<<grid, 256>>>(X, Wh, Wm, Wl, Y, R, N, 4); });
    timeit("bf16x6 NT=4 reps=4", [&] { k_bf16<K, 4, 6><<<grid, 256>>>(X, Wh, Wm, Wl, Y, R, N, 4); });
    k_bf16_lds<K, 2><<<grid, 256, 2 * 3 * 2 * (K / 16) * 64 * 16>>>(X, Wh, Wm, Wl, Y, R, N, 1); check("bf16 x6 (LDS weights)");
    timeit("bf16x6 LDS NT=2 reps=4", [&] { k_bf16_lds<K, 2><<<grid, 256, 2 * 3 * 2 * (K / 16) * 64 * 16>>>(X, Wh, Wm, Wl, Y, R, N, 4); });
    timeit("bf16x6 LDS NT=4 reps=4", [&] { k_bf16_lds<K, 4><<<grid, 256, 2 * 3 * 4 * (K / 16) * 64 * 16>>>(X, Wh, Wm, Wl, Y, R, N, 4); });
    timeit("bf16x3 NT=4 reps=4", [&] { k_bf16<K, 4, 3><<<grid, 256>>>(X, Wh, Wm, Wl, Y, R, N, 4); });
    return 0;
}
