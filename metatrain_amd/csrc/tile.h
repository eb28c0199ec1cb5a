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

// RB row blocks of 32 rows per workgroup: 2 -> (row block, column half) as above; 1 -> one row block, four column quarters
template <int RB>
struct WaveIdT {
    int lane, wave, rb, ch;
    __device__ WaveIdT() {
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        rb = RB == 2 ? (wave & 1) : 0;
        ch = RB == 2 ? (wave >> 1) : wave;
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

// The same product with DEPTH weight blocks in flight (gemm_acc: one): a small graph runs one workgroup per CU and every
// block is a dependent round trip to L2 / MALL -- K = 256 is 32 of them (k_node_bwd2<1>'s expansion adjoint, k_center_bwd).
// Same MFMA order, same bits.
template <int KS, int NT, int DEPTH>
__device__ __forceinline__ void gemm_acc_deep(const float* As, int lda, const float4* __restrict__ Wp, int kg_total, int kg0,
                                              int tile0, f32x16 (&acc)[NT], int lane) {
    constexpr int KG = KS / 8;
    static_assert(KG % DEPTH == 0, "the ring index must be static");
    const float* arow = As + (lane & 31) * lda + (lane >> 5) * 4;
    const float4* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) bp[t] = Wp + ((size_t)(tile0 + t) * kg_total + kg0) * 64 + lane;
    float4 b[DEPTH][NT];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int t = 0; t < NT; t++) b[d][t] = bp[t][d * 64];
#pragma unroll 1
    for (int kg0i = 0; kg0i < KG; kg0i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const int kg = kg0i + d;
            const float4 a = *reinterpret_cast<const float4*>(arow + kg * 8);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[d][t].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[d][t].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[d][t].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[d][t].w, acc[t], 0, 0, 0);
            if (kg + DEPTH < KG)
#pragma unroll
                for (int t = 0; t < NT; t++) b[d][t] = bp[t][(kg + DEPTH) * 64];
        }
    }
}

// ---- the same GEMM on the fp16 matrix cores (f16x3, see trr.h): the A tile stays fp32 in LDS and is split on
// the fly as its fragments are read (two ds_read_b128 and ~30 VALU per K block of 16, against three 32-cycle MFMAs
// per tile instead of eight 64-cycle ones); the weights come as two fp16 planes (abi.hip k_pack2h).
// WX carries both operand forms; h == nullptr selects the fp32 MFMA path above.
// rscale (optional, LDS [32][2] for this wave's row block: power-of-two scale and its inverse per row): for adjoint
// tiles whose rows can be small as a whole; the contribution is accumulated locally and added as (hi + lo/2048)/scale.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
struct WX {
    const float4* f = nullptr;
    const f16x8_t* h = nullptr;
    const f16x8_t* l = nullptr;
};
// DEPTH weight blocks (K = 16 each) in flight: 2 where many workgroups share a CU; the 32-row kernels of small graphs, where
// one wave per SIMD has to cover the L2 round trip by itself, take 8 (a whole K = 128 operand requested at once).
template <int KS, int NT, int DEPTH = 2>
__device__ __forceinline__ void gemm_acc_x(const float* As, int lda, const WX& w, int kg_total, int kg0, int tile0,
                                           f32x16 (&acc)[NT], int lane, const float* rscale = nullptr) {
    if (w.h == nullptr) {
        gemm_acc<KS, NT>(As, lda, w.f, kg_total, kg0, tile0, acc, lane);
        return;
    }
    constexpr int KB = KS / 16;
    static_assert(KB % DEPTH == 0, "the ring index must be static");
    const int kb_total = kg_total / 2, kb0 = kg0 / 2;
    const float* arow = As + (lane & 31) * lda + (lane >> 5) * 4;
    const float sc = rscale ? rscale[2 * (lane & 31)] : 1.0f;
    size_t base[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) base[t] = ((size_t)(tile0 + t) * kb_total + kb0) * 64 + lane;
    f16x8_t wh[DEPTH][NT], wl[DEPTH][NT];
#pragma unroll
    for (int s = 0; s < DEPTH; s++)
#pragma unroll
        for (int t = 0; t < NT; t++) { wh[s][t] = w.h[base[t] + s * 64]; wl[s][t] = w.l[base[t] + s * 64]; }
    f32x16 ah[NT], al[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) { ah[t][r] = 0.f; al[t][r] = 0.f; }
#pragma unroll 1
    for (int kb0i = 0; kb0i < KB; kb0i += DEPTH)
#pragma unroll
    for (int cur = 0; cur < DEPTH; cur++) {
        const int kb = kb0i + cur;
        const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * kb);
        const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * kb + 8);
        const float v[8] = {a0.x * sc, a0.y * sc, a0.z * sc, a0.w * sc, a1.x * sc, a1.y * sc, a1.z * sc, a1.w * sc};
        f16x8_t xh, xl;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const _Float16 hj = (_Float16)v[j];
            xh[j] = hj;
            xl[j] = (_Float16)((v[j] - (float)hj) * 2048.0f);
        }
#pragma unroll
        for (int t = 0; t < NT; t++) al[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl[cur][t], al[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) ah[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh[cur][t], ah[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) al[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh[cur][t], al[t], 0, 0, 0);
        if (kb + DEPTH < KB)
#pragma unroll
            for (int t = 0; t < NT; t++) {
                wh[cur][t] = w.h[base[t] + (kb + DEPTH) * 64];
                wl[cur][t] = w.l[base[t] + (kb + DEPTH) * 64];
            }
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = ah[t][r] + al[t][r] * (1.0f / 2048.0f);
            acc[t][r] += rscale ? v * rscale[2 * acc_row(r, lane) + 1] : v;
        }
}
// The same product with the WHOLE weight operand (K = 16 DEPTH) requested ahead of time: a 32-row workgroup of a small graph
// runs one wave per SIMD, every product starts with an exposed L2 / MALL round trip, and the operand of the NEXT product can
// be requested before the current one starts (k_node2<1>, k_node_bwd2<1>). Same MFMA order as gemm_acc_x: same bits.
template <int NT, int DEPTH>
struct XRing {
    f16x8_t wh[DEPTH][NT], wl[DEPTH][NT];
};
template <int NT, int DEPTH>
__device__ __forceinline__ void xring_request(XRing<NT, DEPTH>& R, const WX& w, int kg_total, int kg0, int tile0, int lane) {
    const int kb_total = kg_total / 2, kb0 = kg0 / 2;
#pragma unroll
    for (int s = 0; s < DEPTH; s++)
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const size_t i = ((size_t)(tile0 + t) * kb_total + kb0 + s) * 64 + lane;
            R.wh[s][t] = w.h[i];
            R.wl[s][t] = w.l[i];
        }
}
template <int NT, int DEPTH>
__device__ __forceinline__ void gemm_acc_x_ring(const float* As, int lda, const XRing<NT, DEPTH>& R, f32x16 (&acc)[NT],
                                                int lane, const float* rscale = nullptr) {
    const float* arow = As + (lane & 31) * lda + (lane >> 5) * 4;
    const float sc = rscale ? rscale[2 * (lane & 31)] : 1.0f;
    f32x16 ah[NT], al[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) { ah[t][r] = 0.f; al[t][r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < DEPTH; kb++) {
        const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * kb);
        const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * kb + 8);
        const float v[8] = {a0.x * sc, a0.y * sc, a0.z * sc, a0.w * sc, a1.x * sc, a1.y * sc, a1.z * sc, a1.w * sc};
        f16x8_t xh, xl;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const _Float16 hj = (_Float16)v[j];
            xh[j] = hj;
            xl[j] = (_Float16)((v[j] - (float)hj) * 2048.0f);
        }
#pragma unroll
        for (int t = 0; t < NT; t++) al[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, R.wl[kb][t], al[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) ah[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, R.wh[kb][t], ah[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) al[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, R.wh[kb][t], al[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = ah[t][r] + al[t][r] * (1.0f / 2048.0f);
            acc[t][r] += rscale ? v * rscale[2 * acc_row(r, lane) + 1] : v;
        }
}
// ---- the same product with the A tile ALREADY split: two fp16 planes [64][K + 8] in LDS, written once by
// split_tile_planes (or directly by the stage that produces the tile). A stage that runs several GEMMs on one tile
// (the node MLP: 16 column-chunk GEMMs on the same [64 x 256] rows) then splits each element once instead of once per
// GEMM and per wave; the K loop is two ds_read_b128 and the MFMAs. Within every block of 16 k the planes hold k in the
// order a lane consumes it (lane group g of gemm_acc_x takes k = 4 g + j and 8 + 4 g + j): plane_pos().
__host__ __device__ constexpr int plane_ld(int K) { return K + 8; }
__device__ __forceinline__ int plane_pos(int k) {
    const int b = k & 15;
    return (k & ~15) + (b < 4 ? b : (b < 8 ? b + 4 : (b < 12 ? b - 4 : b)));
}
// DEPTH weight blocks (K = 16 each) are in flight: at one wave per SIMD the L2 round trip (~1.5k cycles) has to be
// covered by this wave's own MFMAs (96 cycles per block and tile).
template <int KS, int NT, int DEPTH = 2>
__device__ __forceinline__ void gemm_acc_hs(const _Float16* Ah, const _Float16* Al, int ldh, const WX& w, int kg_total,
                                            int kg0, int tile0, f32x16 (&acc)[NT], int lane, int tstride = 1) {
    constexpr int KB = KS / 16;
    static_assert(KB % DEPTH == 0, "the ring index must be static");
    const int kb_total = kg_total / 2, kb0 = kg0 / 2;
    const int roff = (lane & 31) * ldh + (lane >> 5) * 8;
    size_t base[NT];  // output tile t = tile0 + t tstride (tstride != 1: e.g. the value and the gate columns of one hidden tile)
#pragma unroll
    for (int t = 0; t < NT; t++) base[t] = ((size_t)(tile0 + t * tstride) * kb_total + kb0) * 64 + lane;
    f16x8_t wh[DEPTH][NT], wl[DEPTH][NT];
#pragma unroll
    for (int s = 0; s < DEPTH; s++)
#pragma unroll
        for (int t = 0; t < NT; t++) { wh[s][t] = w.h[base[t] + s * 64]; wl[s][t] = w.l[base[t] + s * 64]; }
    f32x16 ah[NT], al[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) { ah[t][r] = 0.f; al[t][r] = 0.f; }
#pragma unroll 1
    for (int kb0i = 0; kb0i < KB; kb0i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const int kb = kb0i + d;
            const f16x8_t xh = *reinterpret_cast<const f16x8_t*>(Ah + roff + 16 * kb);
            const f16x8_t xl = *reinterpret_cast<const f16x8_t*>(Al + roff + 16 * kb);
#pragma unroll
            for (int t = 0; t < NT; t++) al[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl[d][t], al[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) ah[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh[d][t], ah[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) al[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh[d][t], al[t], 0, 0, 0);
            if (kb + DEPTH < KB)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    wh[d][t] = w.h[base[t] + (kb + DEPTH) * 64];
                    wl[d][t] = w.l[base[t] + (kb + DEPTH) * 64];
                }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] += ah[t][r] + al[t][r] * (1.0f / 2048.0f);
}
// Two row blocks per wave: every weight block is requested ONCE and multiplies the A fragments of both 32-row blocks
// (acc[rb * NT + t]; row block rb at + rb * rb_stride elements). The 64-row node kernel ran (row block, column half) waves,
// so two waves of a workgroup streamed the same weights (k_node2w, pet_fwd.hip: - 13 % of the stage at 80 000 atoms).
// Per output element the MFMA order is that of gemm_acc_hs / _x_ring.
template <int KS, int NT, int DEPTH>
__device__ __forceinline__ void gemm_acc_hs_rb2(const _Float16* Ah, const _Float16* Al, int ldh, const WX& w, int kg_total,
                                                int kg0, int tile0, f32x16 (&acc)[2 * NT], int lane, int tstride,
                                                int rb_stride) {
    constexpr int KB = KS / 16;
    static_assert(KB % DEPTH == 0, "the ring index must be static");
    const int kb_total = kg_total / 2, kb0 = kg0 / 2;
    const int roff = (lane & 31) * ldh + (lane >> 5) * 8;
    size_t base[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) base[t] = ((size_t)(tile0 + t * tstride) * kb_total + kb0) * 64 + lane;
    f16x8_t wh[DEPTH][NT], wl[DEPTH][NT];
#pragma unroll
    for (int s = 0; s < DEPTH; s++)
#pragma unroll
        for (int t = 0; t < NT; t++) { wh[s][t] = w.h[base[t] + s * 64]; wl[s][t] = w.l[base[t] + s * 64]; }
    f32x16 ah[2 * NT], al[2 * NT];
#pragma unroll
    for (int t = 0; t < 2 * NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) { ah[t][r] = 0.f; al[t][r] = 0.f; }
#pragma unroll 1
    for (int kb0i = 0; kb0i < KB; kb0i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const int kb = kb0i + d;
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const f16x8_t xh = *reinterpret_cast<const f16x8_t*>(Ah + rb * rb_stride + roff + 16 * kb);
                const f16x8_t xl = *reinterpret_cast<const f16x8_t*>(Al + rb * rb_stride + roff + 16 * kb);
#pragma unroll
                for (int t = 0; t < NT; t++) al[rb * NT + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl[d][t], al[rb * NT + t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; t++) ah[rb * NT + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh[d][t], ah[rb * NT + t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; t++) al[rb * NT + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh[d][t], al[rb * NT + t], 0, 0, 0);
            }
            if (kb + DEPTH < KB)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    wh[d][t] = w.h[base[t] + (kb + DEPTH) * 64];
                    wl[d][t] = w.l[base[t] + (kb + DEPTH) * 64];
                }
        }
    }
#pragma unroll
    for (int t = 0; t < 2 * NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] += ah[t][r] + al[t][r] * (1.0f / 2048.0f);
}
template <int NT, int DEPTH>
__device__ __forceinline__ void gemm_acc_x_ring_rb2(const float* As, int lda, const XRing<NT, DEPTH>& R, f32x16 (&acc)[2 * NT],
                                                    int lane, int rb_stride, const float* rscale = nullptr) {
    f32x16 ah[2 * NT], al[2 * NT];
#pragma unroll
    for (int t = 0; t < 2 * NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) { ah[t][r] = 0.f; al[t][r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < DEPTH; kb++) {
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            const float* arow = As + rb * rb_stride + (lane & 31) * lda + (lane >> 5) * 4;
            const float sc = rscale ? rscale[64 * rb + 2 * (lane & 31)] : 1.0f;
            const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * kb);
            const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * kb + 8);
            const float v[8] = {a0.x * sc, a0.y * sc, a0.z * sc, a0.w * sc, a1.x * sc, a1.y * sc, a1.z * sc, a1.w * sc};
            f16x8_t xh, xl;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const _Float16 hj = (_Float16)v[j];
                xh[j] = hj;
                xl[j] = (_Float16)((v[j] - (float)hj) * 2048.0f);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) al[rb * NT + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, R.wl[kb][t], al[rb * NT + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) ah[rb * NT + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, R.wh[kb][t], ah[rb * NT + t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) al[rb * NT + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, R.wh[kb][t], al[rb * NT + t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float v = ah[rb * NT + t][r] + al[rb * NT + t][r] * (1.0f / 2048.0f);
                acc[rb * NT + t][r] += rscale ? v * rscale[64 * rb + 2 * acc_row(r, lane) + 1] : v;
            }
}
// split one value into its two fp16 pieces (high piece, and the remainder scaled by 2^11: trr.h split2)
__device__ __forceinline__ void split_hl(float v, _Float16& h, _Float16& l) {
    h = (_Float16)v;
    l = (_Float16)((v - (float)h) * 2048.0f);
}
// [64][K] fp32 tile (leading dimension lda) -> the two planes; 256 threads, four per row
template <int K, int ROWS = BM>
__device__ __forceinline__ void split_tile_planes(const float* As, int lda, _Float16* Ah, _Float16* Al) {
    constexpr int LDH = plane_ld(K), TPR = NTHREADS / ROWS;
    const int r = threadIdx.x / TPR, q = threadIdx.x % TPR;
    for (int c = 4 * q; c < K; c += 4 * TPR) {  // this thread: columns c .. c + 3, a run of four inside one block of 16
        const float4 v = *reinterpret_cast<const float4*>(As + r * lda + c);
        const int p = r * LDH + plane_pos(c);
        _Float16 h0, l0, h1, l1, h2, l2, h3, l3;
        split_hl(v.x, h0, l0); split_hl(v.y, h1, l1); split_hl(v.z, h2, l2); split_hl(v.w, h3, l3);
        Ah[p] = h0; Ah[p + 1] = h1; Ah[p + 2] = h2; Ah[p + 3] = h3;
        Al[p] = l0; Al[p + 1] = l1; Al[p + 2] = l2; Al[p + 3] = l3;
    }
}

// per-row power-of-two scales of a staged [64][K] tile: rs[2 r] = scale (row maximum into [1, 2)), rs[2 r + 1] = inverse
template <int K, int ROWS = BM>
__device__ __forceinline__ void tile_row_scales(const float* As, int lda, float* rs) {
    constexpr int TPR = NTHREADS / ROWS;
    const int r = threadIdx.x / TPR, q = threadIdx.x % TPR;
    float m = 0.f;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        const float4 v = *reinterpret_cast<const float4*>(As + r * lda + c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    if (TPR == 8) m = fmaxf(m, __shfl_xor(m, 4));
    int e = (__float_as_int(m) >> 23) & 0xff;
    e = e > 253 ? 253 : e;
    if (q == 0) {
        rs[2 * r] = __int_as_float((254 - e) << 23);
        rs[2 * r + 1] = __int_as_float(e << 23);
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
template <int K, int ROWS = BM>
__device__ __forceinline__ void load_rows_to_lds(float* As, const float* __restrict__ G, int64_t row0,
                                                 int64_t n_rows, int ld_g) {
    constexpr int C4 = K / 4;
    constexpr int LDA = lds_ld(K);
    for (int idx = threadIdx.x; idx < ROWS * C4; idx += NTHREADS) {
        int r = idx / C4, c = idx % C4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < n_rows) v = *reinterpret_cast<const float4*>(G + (row0 + r) * ld_g + 4 * c);
        *reinterpret_cast<float4*>(As + r * LDA + 4 * c) = v;
    }
}

// Cooperative store of a [64][K] fp32 LDS tile to global rows (the inverse of load_rows_to_lds): 512-B row segments.
template <int K, int ROWS = BM>
__device__ __forceinline__ void store_rows_from_lds(const float* As, float* __restrict__ G, int64_t row0, int64_t n_rows,
                                                    int ld_g) {
    constexpr int C4 = K / 4;
    constexpr int LDA = lds_ld(K);
    for (int idx = threadIdx.x; idx < ROWS * C4; idx += NTHREADS) {
        const int r = idx / C4, c = idx % C4;
        if (row0 + r < n_rows)
            *reinterpret_cast<float4*>(G + (row0 + r) * ld_g + 4 * c) = *reinterpret_cast<const float4*>(As + r * LDA + 4 * c);
    }
}

// Device-coherent accesses (relaxed atomics at agent scope: they pass the per-XCD L2 without a cache write-back / invalidate
// fence) for data one workgroup hands to another inside a kernel (k_node2 / k_node_bwd2, SPLIT)
__device__ __forceinline__ void st4_agent(float* p, float4 v) {
    __hip_atomic_store(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld4_agent(const float* cp) {
    float* p = const_cast<float*>(cp);
    return make_float4(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
template <int K, int ROWS = BM>
__device__ __forceinline__ void store_rows_from_lds_agent(const float* As, float* __restrict__ G, int64_t row0, int64_t n_rows,
                                                          int ld_g) {
    constexpr int C4 = K / 4;
    constexpr int LDA = lds_ld(K);
    for (int idx = threadIdx.x; idx < ROWS * C4; idx += NTHREADS) {
        const int r = idx / C4, c = idx % C4;
        if (row0 + r < n_rows) st4_agent(G + (row0 + r) * ld_g + 4 * c, *reinterpret_cast<const float4*>(As + r * LDA + 4 * c));
    }
}

// A wave's [32 rows x 64 columns] accumulator pair leaves through a wave-private [32][64] fp32 staging tile so that
// the global stores are float4 rows (16 lanes per row, 4 rows per instruction) instead of one float per lane; f(row in
// the wave's block, column offset 0..60 step 4, float4) does the store (and may add a residual it loads the same way).
// No workgroup barrier: the LDS serves one wave's requests in order.
template <class F>
__device__ __forceinline__ void wave_rows64(const f32x16 (&acc)[2], float* stage, int lane, F f) {
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) stage[acc_row(r, lane) * 64 + 32 * t + (lane & 31)] = acc[t][r];
    __builtin_amdgcn_wave_barrier();
    const int rr = lane >> 4, cc = 4 * (lane & 15);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int r = 4 * j + rr;
        f(r, cc, *reinterpret_cast<const float4*>(stage + r * 64 + cc));
    }
    __builtin_amdgcn_wave_barrier();
}

// The inverse: f(row in the wave's block, column offset, float4&) LOADS whole float4 rows of a [32 x 64] block, which
// then sit in the staging tile in row-major order; read element (acc_row(r, lane), 32 t + (lane & 31)) afterwards and
// call __builtin_amdgcn_wave_barrier() before the tile is reused.
template <class F>
__device__ __forceinline__ void wave_load_rows64(float* stage, int lane, F f) {
    const int rr = lane >> 4, cc = 4 * (lane & 15);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int r = 4 * j + rr;
        float4 v;
        f(r, cc, v);
        *reinterpret_cast<float4*>(stage + r * 64 + cc) = v;
    }
    __builtin_amdgcn_wave_barrier();
}
// One accumulator ([32 rows x 32 columns]) through a wave-private [32][32] staging tile: float4 rows of 128 B, eight lanes
// per row (the 32-row workgroups of the node kernels, whose waves own a quarter of the columns).
template <class F>
__device__ __forceinline__ void wave_rows32(const f32x16& acc, float* stage, int lane, F f) {
#pragma unroll
    for (int r = 0; r < 16; r++) stage[acc_row(r, lane) * 32 + (lane & 31)] = acc[r];
    __builtin_amdgcn_wave_barrier();
    const int rr = lane >> 3, cc = 4 * (lane & 7);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = 8 * j + rr;
        f(r, cc, *reinterpret_cast<const float4*>(stage + r * 32 + cc));
    }
    __builtin_amdgcn_wave_barrier();
}
template <class F>
__device__ __forceinline__ void wave_load_rows32(float* stage, int lane, F f) {
    const int rr = lane >> 3, cc = 4 * (lane & 7);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = 8 * j + rr;
        float4 v;
        f(r, cc, v);
        *reinterpret_cast<float4*>(stage + r * 32 + cc) = v;
    }
    __builtin_amdgcn_wave_barrier();
}
// [64][K] fp32 tile -> planes, every row multiplied by its power-of-two scale rs[2 r] first (adjoint rows)
template <int K, int ROWS = BM>
__device__ __forceinline__ void split_tile_planes_scaled(const float* As, int lda, const float* rs, _Float16* Ah,
                                                         _Float16* Al) {
    constexpr int LDH = plane_ld(K), TPR = NTHREADS / ROWS;
    const int r = threadIdx.x / TPR, q = threadIdx.x % TPR;
    const float sc = rs[2 * r];
    for (int c = 4 * q; c < K; c += 4 * TPR) {
        const float4 v = *reinterpret_cast<const float4*>(As + r * lda + c);
        const int p = r * LDH + plane_pos(c);
        _Float16 h0, l0, h1, l1, h2, l2, h3, l3;
        split_hl(v.x * sc, h0, l0); split_hl(v.y * sc, h1, l1); split_hl(v.z * sc, h2, l2); split_hl(v.w * sc, h3, l3);
        Ah[p] = h0; Ah[p + 1] = h1; Ah[p + 2] = h2; Ah[p + 3] = h3;
        Al[p] = l0; Al[p + 1] = l1; Al[p + 2] = l2; Al[p + 3] = l3;
    }
}

// ---- row-wise normalisations on an LDS tile [64][K+4] ------------------------------
// Four threads per row (256 threads / 64 rows), strided columns, xor-shuffle reduce.
// RMSNorm: torch.nn.RMSNorm(d), eps = finfo(float32).eps.
template <int K, int ROWS = BM>
__device__ __forceinline__ void rmsnorm_rows_inplace(float* As, const float* __restrict__ gamma,
                                                     float* rstd_out /* LDS [64] or nullptr */) {
    constexpr int LDA = lds_ld(K), TPR = NTHREADS / ROWS;
    const int r = threadIdx.x / TPR, q = threadIdx.x % TPR;
    float* row = As + r * LDA;
    float ss = 0.f;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 v = *reinterpret_cast<float4*>(row + c);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    if (TPR == 8) ss += __shfl_xor(ss, 4);
    const float rstd = rsqrtf(ss * (1.0f / K) + 1.1920928955078125e-07f);
    if (rstd_out && q == 0) rstd_out[r] = rstd;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 v = *reinterpret_cast<float4*>(row + c);
        float4 g = *reinterpret_cast<const float4*>(gamma + c);
        v.x *= rstd * g.x; v.y *= rstd * g.y; v.z *= rstd * g.z; v.w *= rstd * g.w;
        *reinterpret_cast<float4*>(row + c) = v;
    }
}

// RMSNorm (beta == nullptr) or torch.nn.LayerNorm(K) (eps 1e-5, biased variance, weight + bias; transformer.py:170-176)
template <int K, int ROWS = BM>
__device__ __forceinline__ void norm_rows_inplace(float* As, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta) {
    if (beta == nullptr) {
        rmsnorm_rows_inplace<K, ROWS>(As, gamma, nullptr);
        return;
    }
    constexpr int LDA = lds_ld(K), TPR = NTHREADS / ROWS;
    const int r = threadIdx.x / TPR, q = threadIdx.x % TPR;
    float* row = As + r * LDA;
    float s = 0.f;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 v = *reinterpret_cast<float4*>(row + c);
        s += (v.x + v.y) + (v.z + v.w);
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (TPR == 8) s += __shfl_xor(s, 4);
    const float mean = s * (1.0f / K);
    float ss = 0.f;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 v = *reinterpret_cast<float4*>(row + c);
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    if (TPR == 8) ss += __shfl_xor(ss, 4);
    const float rstd = rsqrtf(ss * (1.0f / K) + 1e-5f);
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 v = *reinterpret_cast<float4*>(row + c);
        float4 g = *reinterpret_cast<const float4*>(gamma + c);
        float4 b = *reinterpret_cast<const float4*>(beta + c);
        v.x = (v.x - mean) * rstd * g.x + b.x; v.y = (v.y - mean) * rstd * g.y + b.y;
        v.z = (v.z - mean) * rstd * g.z + b.z; v.w = (v.w - mean) * rstd * g.w + b.w;
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
