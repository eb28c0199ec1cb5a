// The edge head (backend.py:651-777: y = w_l . silu(W2 silu(W0 x + b0) + b2) + b_l per edge, times the cutoff factor, summed
// per centre atom by k_atom_sum) and its adjoint in the form of k_emlp_s (pet_emlp_s.hip): one-accumulator split-operand
// products, two desynchronised four-wave workgroups per CU, the four waves of a workgroup sharing ONE stream of weight
// fragments through a four-slot LDS ring requested three stages ahead. Round 5.
//
// k_head_h / k_head_bwd_h (pet_trr.hip) stream their weights per wave straight from L2 (64 KB per 128 x 128 product and wave):
// with every weight block replaced by block 0 they run 25 % faster (tools/experiments/README.md), and the adjoint needs 492
// registers in its two-accumulator arithmetic (one wave per SIMD). Here a product is 16 stages of four fragments
// (tile pair tp, K block kb: tiles 2 tp, 2 tp + 1 x (h, l)), six MFMAs per wave and stage; nothing but ring requests sits in
// the vmcnt queue between a tile's first stage and its last.
//   forward   x -> (power-of-two row scale) planes of 64 x' -> a1 = W0 x + b0 -> s1 = silu(a1) -> scale, planes ->
//             a2 = W2 s1 + b2 -> y = w_l . silu(a2) + b_l
//   adjoint   recomputes a1 (parked as fp32 fragments over the consumed row tile), s1, a2;  da2 = gy w_l silu'(a2);
//             ds1 = W2^T da2;  da1 = ds1 silu'(a1);  dx = W0^T da1  (each operand scaled per row by a power of two first)
// Inference only (the training adjoint exports s1, da2, da1, s2 y for the weight gradients: k_head_bwd_h<true> keeps doing that).
#include "rows_s.h"

namespace pet {

// stage g of a kernel's weight stream: matrix g / 16, tile pair (g % 16) / 8, K block g % 8; wave w brings tile 2 tp + (w >> 1),
// plane w & 1. NM matrices in all; past the end: the last stage again (identical bytes; keeps vmcnt uniform)
// (the matrices of a kernel's stream, in order -- forward: W0, W2; adjoint: W0, W2, W2^T, W0^T -- are passed as references to
// the kernel's own arguments: a local aggregate of them ends up in scratch memory, and scratch loads are vmcnt traffic)
#define HS_W const W2 &wa, const W2 &wb, const W2 &wc, const W2 &wd
#define HS_WARGS wa, wb, wc, wd
template <int NM>
__device__ __forceinline__ void hs_request(int g, HS_W, unsigned ring_u, int wave, unsigned lane16) {
    g = g < 16 * NM ? g : 16 * NM - 1;
    const unsigned dst = ring_u + (unsigned)(g & (HS_NSLOT - 1)) * HS_SLOT + wave * 1024;
    const int mi = g >> 4, tp = (g >> 3) & 1, kb = g & 7;
    const int idx = (2 * tp + (wave >> 1)) * 8 + kb;
    // (g is a compile-time constant at every call)
    if (mi == 0) ab_dma_piece((wave & 1) ? wa.l : wa.h, idx, lane16, dst);
    else if (mi == 1) ab_dma_piece((wave & 1) ? wb.l : wb.h, idx, lane16, dst);
    else if (mi == 2) ab_dma_piece((wave & 1) ? wc.l : wc.h, idx, lane16, dst);
    else ab_dma_piece((wave & 1) ? wd.l : wd.h, idx, lane16, dst);
}
template <int NM>
__device__ __forceinline__ void hs_gemm(f32x16 (&acc)[4], const f16x8 (&xh)[8], const f16x8 (&xl)[8], int g0, HS_W,
                                        const char* ring, unsigned ring_u, int wave, unsigned lane16) {
    hs_gemm_r(acc, xh, xl, g0, [&](int g) { hs_request<NM>(g, HS_WARGS, ring_u, wave, lane16); }, ring, lane16);
}

__global__ __launch_bounds__(256, 2) void k_head_s(const float* __restrict__ Xin, W2 w0, const float* __restrict__ b0, W2 w2,
                                                  const float* __restrict__ b2, const float* __restrict__ wl, float bl,
                                                  const float* __restrict__ fc, float* __restrict__ ypred,
                                                  float* __restrict__ yout, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) char hs_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * HS_NW + wave) * WROWS;
    const bool live = row0 < R;
    if (!live) row0 = ((R - 1) / WROWS) * WROWS;  // run along on the last tile (same barriers), store nothing
    const int64_t row = row0 + L.r < R ? row0 + L.r : R - 1;
    const bool valid = live && row0 + L.r < R;
    char* tile = hs_smem + wave * 16384;
    const char* ring = hs_smem + HS_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    dma_tile128(Xin, row0, R, tile_u, L);
    const float fcr = fc ? fc[row] : 1.0f;
    asm volatile("" ::"v"(fcr));
    hs_request<2>(0, w0, w2, w2, w2, ring_u, wave, lane16);
    hs_request<2>(1, w0, w2, w2, w2, ring_u, wave, lane16);
    hs_request<2>(2, w0, w2, w2, w2, ring_u, wave, lane16);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // the rows and the cutoff factor (the three ring requests may be in flight)
    f16x8 xh[8], xl[8];
    float inv;
    {   // backbone features are un-normalised rows: power-of-two row scale (largest entry in [1, 2)), exact
        float4 x[16];
        tile128_to_frag(x, tile, L);
        float sc;
        inv = row_scale_pow2<16>(x, sc);
        hs_planes(x, xh, xl);
    }
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = ab_zero();
    hs_gemm<2>(acc, xh, xl, 0, w0, w2, w2, w2, ring, ring_u, wave, lane16);
    {
        float4 s1[16];
        const float f = inv * ABQ_INV;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                s1[4 * t + j] = make_float4(silu_(fmaf(acc[t][4 * j], f, hs_vec(b0, t, j, 0, L.h))),
                                            silu_(fmaf(acc[t][4 * j + 1], f, hs_vec(b0, t, j, 1, L.h))),
                                            silu_(fmaf(acc[t][4 * j + 2], f, hs_vec(b0, t, j, 2, L.h))),
                                            silu_(fmaf(acc[t][4 * j + 3], f, hs_vec(b0, t, j, 3, L.h))));
        float sc;
        inv = row_scale_pow2<16>(s1, sc);
        hs_planes(s1, xh, xl);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = ab_zero();
    hs_gemm<2>(acc, xh, xl, 16, w0, w2, w2, w2, ring, ring_u, wave, lane16);
    float part = 0.f;
    {
        const float f = inv * ABQ_INV;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    part = fmaf(silu_(fmaf(acc[t][4 * j + i], f, hs_vec(b2, t, j, i, L.h))), hs_vec(wl, t, j, i, L.h), part);
    }
    const float y = row_sum(part) + bl;
    if (valid && L.h == 0) {
        if (ypred) ypred[row] = y;
        yout[row] = y * fcr;
    }
}

__global__ __launch_bounds__(256, 2) void k_head_bwd_s(const float* __restrict__ Xin, W2 w0f, const float* __restrict__ b0, W2 w2f,
                                                      const float* __restrict__ b2, W2 w2b, W2 w0b, const float* __restrict__ wl,
                                                      const float* __restrict__ gA, const int* __restrict__ ctr,
                                                      const float* __restrict__ fc, const float* __restrict__ ypred,
                                                      float* __restrict__ dfc, float* __restrict__ dXout, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) char hs_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * HS_NW + wave) * WROWS;
    const bool live = row0 < R;
    if (!live) row0 = ((R - 1) / WROWS) * WROWS;
    const int64_t row = row0 + L.r < R ? row0 + L.r : R - 1;
    const bool valid = live && row0 + L.r < R;
    char* tile = hs_smem + wave * 16384;
    const char* ring = hs_smem + HS_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    dma_tile128(Xin, row0, R, tile_u, L);
    float gy;  // dL/dy of this edge: the centre atom's seed times the cutoff factor
    {
        const float ga = gA[ctr[row]];
        gy = ga * fc[row];
        if (valid && L.h == 0) dfc[row] = ga * ypred[row];  // d(y fc)/dfc
    }
    asm volatile("" : "+v"(gy));  // (formed here: a later compiler wait for its operands would sit behind the ring requests)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows, seeds and the dfc store: nothing but fragments in the queue from here on
    hs_request<4>(0, w0f, w2f, w2b, w0b, ring_u, wave, lane16);
    hs_request<4>(1, w0f, w2f, w2b, w0b, ring_u, wave, lane16);
    hs_request<4>(2, w0f, w2f, w2b, w0b, ring_u, wave, lane16);
    f16x8 xh[8], xl[8];
    float inv;
    {
        float4 x[16];
        tile128_to_frag(x, tile, L);
        float sc;
        inv = row_scale_pow2<16>(x, sc);
        hs_planes(x, xh, xl);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float4* const a1p = reinterpret_cast<float4*>(tile);  // a1 as row fragments [kg][lane] over the consumed rows
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = ab_zero();
    hs_gemm<4>(acc, xh, xl, 0, w0f, w2f, w2b, w0b, ring, ring_u, wave, lane16);
    {   // a1 = W0 x + b0 (parked), s1 = silu(a1) -> planes
        float4 s1[16];
        const float f = inv * ABQ_INV;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 a = make_float4(fmaf(acc[t][4 * j], f, hs_vec(b0, t, j, 0, L.h)), fmaf(acc[t][4 * j + 1], f, hs_vec(b0, t, j, 1, L.h)),
                                             fmaf(acc[t][4 * j + 2], f, hs_vec(b0, t, j, 2, L.h)), fmaf(acc[t][4 * j + 3], f, hs_vec(b0, t, j, 3, L.h)));
                a1p[(4 * t + j) * 64 + L.lane] = a;
                s1[4 * t + j] = make_float4(silu_(a.x), silu_(a.y), silu_(a.z), silu_(a.w));
            }
        float sc;
        inv = row_scale_pow2<16>(s1, sc);
        hs_planes(s1, xh, xl);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = ab_zero();
    hs_gemm<4>(acc, xh, xl, 16, w0f, w2f, w2b, w0b, ring, ring_u, wave, lane16);
    {   // a2 = W2 s1 + b2  ->  da2 = gy w_l silu'(a2)
        float4 d[16];
        const float f = inv * ABQ_INV;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                d[4 * t + j] = make_float4(gy * hs_vec(wl, t, j, 0, L.h) * silu_g_(fmaf(acc[t][4 * j], f, hs_vec(b2, t, j, 0, L.h))),
                                           gy * hs_vec(wl, t, j, 1, L.h) * silu_g_(fmaf(acc[t][4 * j + 1], f, hs_vec(b2, t, j, 1, L.h))),
                                           gy * hs_vec(wl, t, j, 2, L.h) * silu_g_(fmaf(acc[t][4 * j + 2], f, hs_vec(b2, t, j, 2, L.h))),
                                           gy * hs_vec(wl, t, j, 3, L.h) * silu_g_(fmaf(acc[t][4 * j + 3], f, hs_vec(b2, t, j, 3, L.h))));
        float sc;
        inv = row_scale_pow2<16>(d, sc);
        hs_planes(d, xh, xl);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = ab_zero();
    hs_gemm<4>(acc, xh, xl, 32, w0f, w2f, w2b, w0b, ring, ring_u, wave, lane16);
    {   // ds1 = W2^T da2  ->  da1 = ds1 silu'(a1)
        float4 d[16];
        const float f = inv * ABQ_INV;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 a = a1p[(4 * t + j) * 64 + L.lane];
                d[4 * t + j] = make_float4(acc[t][4 * j] * f * silu_g_(a.x), acc[t][4 * j + 1] * f * silu_g_(a.y),
                                           acc[t][4 * j + 2] * f * silu_g_(a.z), acc[t][4 * j + 3] * f * silu_g_(a.w));
            }
        float sc;
        inv = row_scale_pow2<16>(d, sc);
        hs_planes(d, xh, xl);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = ab_zero();
    hs_gemm<4>(acc, xh, xl, 48, w0f, w2f, w2b, w0b, ring, ring_u, wave, lane16);
    // dx = W0^T da1: whole lines through the wave's tile (a1 is dead)
    float4 dx[16];
    {
        const float f = inv * ABQ_INV;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                dx[4 * t + j] = make_float4(acc[t][4 * j] * f, acc[t][4 * j + 1] * f, acc[t][4 * j + 2] * f, acc[t][4 * j + 3] * f);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    store_rows_lines<16>(dx, reinterpret_cast<float*>(tile), L, [&](int r) { return live && row0 + r < R ? dXout + (row0 + r) * D : nullptr; });
}

static inline W2 hs_w2(const void* base, int n_out, int k_in) {
    const size_t n8 = (size_t)(n_out / 32) * (k_in / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}

// false = not served (weights not packed for it, small graph, or the edge MLP's switch is off: pet_config_set("emlp_s"))
bool head_edge_s(const Model& m, const float* Xin, const float* fc, float* ypred, float* yout, int64_t E, hipStream_t st) {
    if (!emlp_s_serves(E) || !m.eh0.fwd2s || !m.eh2.fwd2s) return false;
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    allow_big_lds(k_head_s, lds);
    k_head_s<<<(int)cdiv(E, HS_NW * WROWS), 256, lds, st>>>(Xin, hs_w2(m.eh0.fwd2s, DH, D), m.eh0.b, hs_w2(m.eh2.fwd2s, DH, DH), m.eh2.b,
                                                             m.ell_w, m.ell_b, fc, ypred, yout, E);
    return true;
}
bool head_edge_bwd_s(const Model& m, const float* Xin, const float* gA, const int* ctr, const float* fc, const float* ypred,
                     float* dfc, float* dXout, int64_t E, hipStream_t st) {
    if (!emlp_s_serves(E) || !m.eh0.fwd2s || !m.eh2.fwd2s || !m.eh0.bwd2s || !m.eh2.bwd2s) return false;
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    allow_big_lds(k_head_bwd_s, lds);
    k_head_bwd_s<<<(int)cdiv(E, HS_NW * WROWS), 256, lds, st>>>(Xin, hs_w2(m.eh0.fwd2s, DH, D), m.eh0.b, hs_w2(m.eh2.fwd2s, DH, DH),
                                                                 m.eh2.b, hs_w2(m.eh2.bwd2s, DH, DH), hs_w2(m.eh0.bwd2s, DH, D), m.ell_w, gA,
                                                                 ctr, fc, ypred, dfc, dXout, E);
    return true;
}

}  // namespace pet
