// fp32 MFMA row-tile toolkit for gfx950 (wave64).
//
// Every dense stage of PET is "rows x small weight": M = #edges (+ #atoms) rows,
// K, N in {128..1024}. A workgroup (256 threads = 4 waves) owns BM = 64 rows:
//   * the A tile lives in LDS as [64][K+4] fp32 (the +4 pad makes the per-lane
//     ds_read_b128 of 4 consecutive k conflict-free: (K+4)/4 is odd mod 16);
//   * the weight is streamed from L2 straight into registers in a pre-packed
//     "fragment order" so each wave's load is one fully coalesced 1 KiB read;
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles) accumulates 32x32 tiles.
// Waves are arranged 2 (row blocks of 32) x 2 (column halves).
//
// k ordering trick: one float4 per lane covers 4 MFMA k-steps. Lane group g = lane>>5
// supplies k = 8*kg + 4*g + j for step j, for both A and B, so the sum over k is
// complete after the 4 steps; the summation order differs from a serial dot product
// only by fp32 reassociation.
#pragma once
#include "common.h"

namespace pet {

constexpr int BM = 64;
constexpr int NTHREADS = 256;

__host__ __device__ constexpr int lds_ld(int K) { return K + 4; }

struct WaveId {
    int lane, wave, rb, ch;  // rb: row block (0/1), ch: column half (0/1)
    __device__ WaveId() {
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        rb = wave & 1;
        ch = wave >> 1;
    }
};

// C/D layout of the 32x32 tile (cdna_hip_programming.md §3):
//   col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), r in [0, 16)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// acc[t] += A[32 x KS] (LDS, this wave's row block, columns 0..KS-1)
//           * W[k-groups kg0 .. kg0+KS/8) of tiles tile0..tile0+NT) (global, packed with
//             kg_total = K/8 k-groups per 32-column tile)
template <int KS, int NT>
__device__ __forceinline__ void gemm_acc(const float* As, int lda, const float4* __restrict__ Wp,
                                         int kg_total, int kg0, int tile0, f32x16 (&acc)[NT], int lane) {
    constexpr int KG = KS / 8;
    const float* arow = As + (lane & 31) * lda + (lane >> 5) * 4;
    const float4* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) bp[t] = Wp + ((size_t)(tile0 + t) * kg_total + kg0) * 64 + lane;
    float4 bc[NT], bn[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) bc[t] = bp[t][0];
#pragma unroll 4
    for (int kg = 0; kg < KG; kg++) {
        if (kg + 1 < KG) {
#pragma unroll
            for (int t = 0; t < NT; t++) bn[t] = bp[t][(kg + 1) * 64];
        }
        const float4 a = *reinterpret_cast<const float4*>(arow + kg * 8);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bc[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bc[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bc[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bc[t].w, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) bc[t] = bn[t];
    }
}

template <int NT>
__device__ __forceinline__ void acc_fill_bias(f32x16 (&acc)[NT], const float* __restrict__ bias, int col0,
                                              int lane) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
        float b = bias ? bias[col0 + 32 * t + (lane & 31)] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = b;
    }
}

// for each element of this wave's accumulators: f(row_in_tile [0,64), col, value)
template <int NT, class F>
__device__ __forceinline__ void acc_foreach(f32x16 (&acc)[NT], int rb, int col0, int lane, F f) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            f(rb * 32 + acc_row(r, lane), col0 + 32 * t + (lane & 31), acc[t][r]);
        }
    }
}

// Cooperative load of a [64][K] fp32 row tile from global (row stride ld_g) into LDS
// [64][K+4]; rows >= n_rows are zero-filled. 32 consecutive lanes read one row's
// float4s (512 B contiguous).
template <int K>
__device__ __forceinline__ void load_rows_to_lds(float* As, const float* __restrict__ G, int64_t row0,
                                                 int64_t n_rows, int ld_g) {
    constexpr int C4 = K / 4;
    constexpr int LDA = lds_ld(K);
    for (int idx = threadIdx.x; idx < BM * C4; idx += NTHREADS) {
        int r = idx / C4, c = idx % C4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < n_rows) v = *reinterpret_cast<const float4*>(G + (row0 + r) * ld_g + 4 * c);
        *reinterpret_cast<float4*>(As + r * LDA + 4 * c) = v;
    }
}

// ---- row-wise normalisations on an LDS tile [64][K+4] ------------------------------
// Four threads per row (256 threads / 64 rows), strided columns, xor-shuffle reduce.
// RMSNorm: torch.nn.RMSNorm(d), eps = finfo(float32).eps.
template <int K>
__device__ __forceinline__ void rmsnorm_rows_inplace(float* As, const float* __restrict__ gamma,
                                                     float* rstd_out /* LDS [64] or nullptr */) {
    constexpr int LDA = lds_ld(K);
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    float* row = As + r * LDA;
    float ss = 0.f;
    for (int c = q * 4; c < K; c += 16) {
        float4 v = *reinterpret_cast<float4*>(row + c);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    const float rstd = rsqrtf(ss * (1.0f / K) + 1.1920928955078125e-07f);
    if (rstd_out && q == 0) rstd_out[r] = rstd;
    for (int c = q * 4; c < K; c += 16) {
        float4 v = *reinterpret_cast<float4*>(row + c);
        float4 g = *reinterpret_cast<const float4*>(gamma + c);
        v.x *= rstd * g.x; v.y *= rstd * g.y; v.z *= rstd * g.z; v.w *= rstd * g.w;
        *reinterpret_cast<float4*>(row + c) = v;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// d silu / dx = s (1 + x (1 - s))
__device__ __forceinline__ float silu_grad_(float x) {
    float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}

}  // namespace pet
