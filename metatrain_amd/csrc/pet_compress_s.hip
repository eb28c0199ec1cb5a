// The inference adjoint of the compress stage (transformer.py:499-521) in the form of k_emlp_s / k_head_s (rows_s.h): round 5.
//   forward (k_compress_h, pet_trr.hip):  a0 = [v, d] Wc^T + Tbl[species] (+ M W0c^T);  e = SiLU(a0) W2^T + b2
//   adjoint:  da0 = (dE W2) . silu'(a0);  dgeo += da0 Wc;  dM += da0 W0c
// Same arithmetic as k_compress_bwd_h (pet_trr.hip), which keeps serving small graphs and the training passes; that kernel
// streams 64 KB of weight fragments per product and WAVE from L2 (served from L1 it ran 25 % faster: tools/experiments/README.md):
// 0.87 -> 0.62 ms per launch (later GNN layers), 0.47 -> 0.34 (first layer). Every vector load and store of a tile happens before
// its first ring stage or after its last. The FORWARD was built in this form too and stayed with k_compress_h: 0.75 against 0.71
// and 0.56 against 0.54 ms -- it is bound by its two stores per row, and the geometry / species terms' loads (64 float4 of Wc per
// lane), which k_compress_h issues behind its product, must be consumed before the ring starts here.
#include "rows_s.h"

namespace pet {

// the values are formed HERE (a compiler wait for their operands further down would sit behind the ring requests)
__device__ __forceinline__ void cs_pin(float4 (&x)[16]) {
#pragma unroll
    for (int k = 0; k < 16; k++) asm volatile("" : "+v"(x[k].x), "+v"(x[k].y), "+v"(x[k].z), "+v"(x[k].w));
}
// one (tile pair, K block) stage of a 128 x 128 matrix: wave w brings tile 2 tp + (w >> 1), plane w & 1
__device__ __forceinline__ void cs_piece(const W2& m, int r, unsigned dst, int wave, unsigned lane16) {
    const int tp = (r >> 3) & 1, kb = r & 7;
    ab_dma_piece((wave & 1) ? m.l : m.h, (2 * tp + (wave >> 1)) * 8 + kb, lane16, dst);
}

// stream of the adjoint: W2^T (16 stages), Wc^T padded to one tile (4 stages of two K blocks x (h, l)), then (not FIRST) W0c^T (16)
template <bool FIRST>
__global__ __launch_bounds__(256, 2) void k_compress_bwd_s(const float* __restrict__ dXe, const float* __restrict__ a0, W2 w2b,
                                                          W2 wcp /* Wc^T padded to [32][D], planes of 64 w */, W2 w0cb,
                                                          float* __restrict__ dgeo, float* __restrict__ dM, int64_t E) {
    extern __shared__ __attribute__((aligned(16))) char cs_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * HS_NW + wave) * WROWS;
    const bool live = row0 < E;
    if (!live) row0 = ((E - 1) / WROWS) * WROWS;
    const int64_t row = row0 + L.r < E ? row0 + L.r : E - 1;
    const bool valid = live && row0 + L.r < E;
    char* tile = cs_smem + wave * 16384;
    const char* ring = cs_smem + HS_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    constexpr int NST = FIRST ? 20 : 36;
    auto req = [&](int g) {
        g = g < NST ? g : NST - 1;
        const unsigned dst = ring_u + (unsigned)(g & (HS_NSLOT - 1)) * HS_SLOT + wave * 1024;
        if (g < 16) cs_piece(w2b, g, dst, wave, lane16);
        else if (g < 20) ab_dma_piece((wave & 1) ? wcp.l : wcp.h, 2 * (g - 16) + (wave >> 1), lane16, dst);  // K blocks 2 r, 2 r + 1
        else cs_piece(w0cb, g - 20, dst, wave, lane16);
    };
    dma_tile128(dXe, row0, E, tile_u, L);
    float4 d[16];  // a0, then da0
    load_rowfrag<16>(d, a0, row, D, L.h);
    cs_pin(d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    req(0);
    req(1);
    req(2);
    f16x8 xh[8], xl[8];
    f32x16 acc[4];
    float inv;
    {
        float4 x[16];
        tile128_to_frag(x, tile, L);
        float sc;
        inv = row_scale_pow2<16>(x, sc);
        hs_planes(x, xh, xl);
    }
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = ab_zero();
    hs_gemm_r(acc, xh, xl, 0, req, ring, lane16);
    {
        const float f = inv * ABQ_INV;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float4& a = d[4 * t + j];
                a = make_float4(acc[t][4 * j] * f * silu_g_(a.x), acc[t][4 * j + 1] * f * silu_g_(a.y), acc[t][4 * j + 2] * f * silu_g_(a.z),
                                acc[t][4 * j + 3] * f * silu_g_(a.w));
            }
        float sc;
        inv = row_scale_pow2<16>(d, sc);
        hs_planes(d, xh, xl);
    }
    // dgeo[row][q] += sum_c da0[c] Wc[c][q] as one MFMA tile: Wc^T padded to 32 rows is the A operand, the tile's first four rows --
    // registers 0 .. 3 of the lanes with h = 0 -- are dgeo[row][0 .. 3] (k_compress_bwd_h)
    f32x16 gt = ab_zero();
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int g = 16 + r;
        HS_STAGE_SYNC();
        req(g + 3);
        const char* slot = ring + (g & (HS_NSLOT - 1)) * HS_SLOT + lane16;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * j) * 1024);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * j + 1) * 1024);
            AB_MFMA3(gt, wh, wl, xh[2 * r + j], xl[2 * r + j]);
        }
    }
    if (!FIRST) {
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = ab_zero();
        hs_gemm_r(acc, xh, xl, 20, req, ring, lane16);
    }
    // ---- after the last stage: the read-modify-writes
    const float f = inv * ABQ_INV;
    if (valid && L.h == 0) {
        float4* dg = reinterpret_cast<float4*>(dgeo + row * 4);
        const float4 old = *dg;
        *dg = make_float4(fmaf(gt[0], f, old.x), fmaf(gt[1], f, old.y), fmaf(gt[2], f, old.z), fmaf(gt[3], f, old.w));
    }
    if (!FIRST) {
        auto rows = [&](int r) { return dM + (row0 + r < E ? row0 + r : E - 1) * D; };
        float4 old[16], y[16];
        request_rows_addend<16>(old, L, rows);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                y[4 * t + j] = make_float4(acc[t][4 * j] * f, acc[t][4 * j + 1] * f, acc[t][4 * j + 2] * f, acc[t][4 * j + 3] * f);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        store_rows_lines_add<16>(y, old, reinterpret_cast<float*>(tile), L,
                                 [&](int r) { return live && row0 + r < E ? dM + (row0 + r) * D : nullptr; });
    }
}

static inline W2 cs_w2(const void* base, int tiles, int kbs) {
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + (size_t)tiles * kbs * 64;
    return w;
}

// false = not served (weights not packed for it, small graph, or the edge MLP's switch is off: pet_config_set("emlp_s"))
bool compress_bwd_s(bool first, const float* dXe, const float* a0, const GnnLayerW& G, float* dgeo, float* dM, int64_t E,
                    hipStream_t st) {
    if (!emlp_s_serves(E) || !G.compress2.bwd2s || !G.wc2s || !(first || G.compress0_msg.bwd2s)) return false;
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    const int grid = (int)cdiv(E, HS_NW * WROWS);
    const W2 w2b = cs_w2(G.compress2.bwd2s, D / 32, D / 16), wcp = cs_w2(G.wc2s, 1, D / 16);
    if (first) {
        allow_big_lds(k_compress_bwd_s<true>, lds);
        k_compress_bwd_s<true><<<grid, 256, lds, st>>>(dXe, a0, w2b, wcp, W2(), dgeo, nullptr, E);
    } else {
        allow_big_lds(k_compress_bwd_s<false>, lds);
        k_compress_bwd_s<false><<<grid, 256, lds, st>>>(dXe, a0, w2b, wcp, cs_w2(G.compress0_msg.bwd2s, D / 32, D / 16), dgeo, dM, E);
    }
    return true;
}

}  // namespace pet
