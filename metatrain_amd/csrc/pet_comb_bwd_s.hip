// The inference adjoint of the message-passing combination stage (backend.py:559-575; forward: pet_comb_s.hip) in the form of
// k_emlp_s (rows_s.h): one-accumulator split-operand products, two desynchronised four-wave workgroups per CU, one weight stream per
// workgroup through a four-slot LDS ring requested three stages ahead. Round 6. k_comb_bwd_p2 (pet_comb_bwd.hip: 440 + 184
// registers, one wave per SIMD, 40 KB of LDS per wave, its weights streamed per WAVE from L2) keeps serving small graphs and the
// training pass (which exports da for the weight gradients).
//   forward   a = W0g xhat + b0g (xhat = LayerNorm-hat([e ; e[rev]]), the affine part folded into W0g / b0g);  M' = M + e + W2 silu(a) + b2
//   adjoint   t1 = W2^T dM;  da = t1 . silu'(a);  dxhat = W0g^T da;  dcat = rstd (dxhat - mean(dxhat) - xhat mean(dxhat . xhat))
// in chunks of 32 hidden units, as the forward: per chunk 4 stages of W2^T (one 32-unit tile, K = 128) and 8 stages of W0g^T (eight
// 32-column output tiles, K = the chunk's two K blocks): the stream has the forward's 96 stages. Two things keep the kernel inside
// 256 registers and 16 KB of LDS per wave:
//   * sum_j dxhat[j] xhat[j] = sum_h da[h] (a[h] - b0g[h]) -- the LayerNorm adjoint's second sum is taken chunk by chunk from values
//     the loop holds anyway, so xhat is needed only where the result is formed, one 128-column half at a time;
//   * the wave's 16-KB row tile is split once dM has been turned into planes: the LOW plane of dM is parked in its first half (each
//     lane reads back the fragments it wrote: 32 registers less, which is what the loop was short of), and the saved pre-activations
//     arrive by LDS-DMA in the second half, a slab of two chunks (64 of the 256 columns) at a time: the first before the ring starts,
//     the next requested when the odd chunk has read its values -- 8 requests that may stay in flight over the next three stage
//     waits (vmcnt(2 + 8)) and are drained by the fourth, nine stages before they are read.
// dM is an adjoint: one power-of-two scale per row, carried by t1, da, dxhat and the sums until the result is formed.
#include "rows_s.h"

namespace pet {

constexpr int CBS_SPC = 12, CBS_NC = 2 * D / 32;

// stage 12 hc + s; wave w brings fragment w
//   s < 4:  W2^T tile hc (hidden units 32 hc ..), K blocks 2 s + (w >> 1) of the 128 message features, plane w & 1
//   s >= 4: W0g^T output tile s - 4 (columns 32 (s - 4) .. of the 256), K blocks 2 hc + (w >> 1) of the hidden units, plane w & 1
__device__ __forceinline__ void cbs_request(int hc, int s, const W2& w2b, const W2& w0b, unsigned ring_u, int wave, unsigned lane16) {
    if (s >= CBS_SPC) { s -= CBS_SPC; hc += 1; }
    if (hc >= CBS_NC) { hc = CBS_NC - 1; s = CBS_SPC - 1; }  // past the end: the last stage again (keeps vmcnt uniform)
    const unsigned dst = ring_u + (unsigned)((CBS_SPC * hc + s) & (HS_NSLOT - 1)) * HS_SLOT + wave * 1024;
    const int pl = wave & 1, j = wave >> 1;
    if (s < 4) ab_dma_piece(pl ? w2b.l : w2b.h, hc * (D / 16) + 2 * s + j, lane16, dst);
    else ab_dma_piece(pl ? w0b.l : w0b.h, (s - 4) * (2 * D / 16) + 2 * hc + j, lane16, dst);
}
#define CBS_STAGE_SYNC(LEAD)                                                        \
    do {                                                                            \
        if (LEAD) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");      \
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");            \
        __syncthreads();                                                            \
    } while (0)

// whole rows of a [32 x 128] fp32 block (leading dimension ld) into the wave's tile: trr.h dma_tile128 with one wave-uniform base and
// 32-bit lane offsets (so_rows_s.hip rs_dma_tile: the 64-bit row addresses of three call sites do not stay live across the loop)
__device__ __forceinline__ void cbs_dma_rows(const float* __restrict__ X, int64_t r0, int64_t R, int ld, unsigned lds_base,
                                             const RowLane& L) {
    const float* base = X + r0 * ld;
    const int rmax = (int)(R - 1 - r0 < 31 ? R - 1 - r0 : 31);
    const int hi = L.lane >> 5, c = L.lane & 31;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int r = 2 * j + hi, rr = r < rmax ? r : rmax;
        const unsigned off = ((unsigned)rr * (unsigned)ld + 4u * (unsigned)(c ^ (r & 15))) * 4u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_base + j * 1024) : "memory");
    }
}

// a [32 rows x 64 columns] slab of the [E, 256] pre-activations into 8 KB of LDS: instruction j brings rows 4 j .. 4 j + 3 (16 lanes x
// 16 B per row); row r, 16-B piece c lands at byte 256 r + 16 (c ^ (r & 15))
__device__ __forceinline__ void cbs_dma_slab(const float* __restrict__ X, int64_t r0, int64_t R, unsigned lds_base, const RowLane& L) {
    const float* base = X + r0 * (2 * D);
    const int rmax = (int)(R - 1 - r0 < 31 ? R - 1 - r0 : 31);
    const int q = L.lane >> 4, c = L.lane & 15;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int r = 4 * j + q, rr = r < rmax ? r : rmax;
        const unsigned off = ((unsigned)rr * (unsigned)(2 * D) + 4u * (unsigned)(c ^ (r & 15))) * 4u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_base + j * 1024) : "memory");
    }
}

// an opaque copy of the lane indices / of the tile's first row: what is derived from it is formed where it is used (and not kept
// across the chunk loop in spilled registers: spills are scratch loads, i.e. vmcnt traffic inside the ring's window)
#define CBS_LANE(Lx)                           \
    RowLane Lx = L;                            \
    asm volatile("" : "+v"(Lx.lane));          \
    Lx.r = Lx.lane & 31;                       \
    Lx.h = Lx.lane >> 5
#define CBS_ROW0(rx) int64_t rx = row0; asm volatile("" : "+s"(rx))

template <bool ADD_DM>
__global__ __launch_bounds__(256, 2) void k_comb_bwd_s(const float* __restrict__ dM, const float* __restrict__ XF,
                                                      const int* __restrict__ rev, const float* __restrict__ LNS,
                                                      const float* __restrict__ CA, const float* __restrict__ b0g, W2 w2b, W2 w0b,
                                                      float* __restrict__ dcat, int64_t E) {
    extern __shared__ __attribute__((aligned(16))) char cbs_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * HS_NW + wave) * WROWS;
    const bool live = row0 < E;
    if (!live) row0 = ((E - 1) / WROWS) * WROWS;  // run along on the last tile (same barriers), store nothing
    char* tile = cbs_smem + wave * 16384;
    const char* ring = cbs_smem + HS_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    // ---- rows: dM -> planes (the low plane parked in the first half of the consumed tile: every lane reads back what it wrote);
    // then the first slab (chunks 0, 1: 64 columns) of the saved pre-activations into the second half
    cbs_dma_rows(dM, row0, E, D, tile_u, L);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f16x8 mh[8];
    float inv;
    {
        float4 d[16];
        tile128_to_frag(d, tile, L);
        float sc;
        inv = row_scale_pow2<16>(d, sc);
        f16x8 ml[8];
        hs_planes(d, mh, ml);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have RETURNED before the tile is written again (pet_comb_s.hip)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kb = 0; kb < 8; kb++) *reinterpret_cast<f16x8*>(tile + kb * 1024 + lane16) = ml[kb];
    }
    {
        CBS_LANE(L1);
        CBS_ROW0(r1);
        cbs_dma_slab(CA, r1, E, tile_u + 8192, L1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cbs_request(0, 0, w2b, w0b, ring_u, wave, lane16);
    cbs_request(0, 1, w2b, w0b, ring_u, wave, lane16);
    cbs_request(0, 2, w2b, w0b, ring_u, wave, lane16);
    f32x16 dl[8];
#pragma unroll
    for (int t = 0; t < 8; t++) dl[t] = ab_zero();
    float s2 = 0.f;  // sum_h da[h] (a[h] - b0g[h]) of this lane's hidden units (scaled like dM)

#pragma unroll 1
    for (int hc = 0; hc < CBS_NC; hc++) {
        CBS_LANE(Lp);
        const unsigned lane16p = (unsigned)Lp.lane * 16u;
        f32x16 t1 = ab_zero();
#pragma unroll
        for (int s = 0; s < 4; s++) {
            CBS_STAGE_SYNC(false);
            cbs_request(hc, s + 3, w2b, w0b, ring_u, wave, lane16p);
            const char* slot = ring + ((CBS_SPC * hc + s) & (HS_NSLOT - 1)) * HS_SLOT + lane16p;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * j) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * j + 1) * 1024);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(tile + (2 * s + j) * 1024 + lane16p);
                AB_MFMA3(t1, wh, wl, mh[2 * s + j], xl);
            }
        }
        // da = t1 . silu'(a): the chunk's 32 pre-activations of this lane's row from the slab (its columns 32 (hc & 1) + 8 j + 4 h ..)
        {
            const char* rowp = tile + 8192 + 256 * Lp.r;
            const int sw = Lp.r & 15;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 a4 = *reinterpret_cast<const float4*>(rowp + 16 * ((8 * (hc & 1) + 2 * j + Lp.h) ^ sw));
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float d = t1[4 * j + i] * ABQ_INV * silu_g_(av[i]);
                    s2 = fmaf(d, av[i] - hs_vec(b0g + 32 * hc, 0, j, i, Lp.h), s2);
                    t1[4 * j + i] = d;
                }
            }
        }
        const bool refill = (hc & 1) && hc + 1 < CBS_NC;
        if (refill) {  // the next slab (chunks hc + 1, hc + 2); this one's two chunks have read their values
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            CBS_LANE(L2);
            CBS_ROW0(r2);
            cbs_dma_slab(CA + 32 * (hc + 1), r2, E, tile_u + 8192, L2);
        }
        f16x8 uh[2], ul[2];
        ab_tile_planes(t1, uh, ul);
#pragma unroll
        for (int s = 4; s < 12; s++) {
            CBS_STAGE_SYNC(refill && s < 7);  // (the fragments of stages 4 .. 6 were requested before the refill)
            cbs_request(hc, s + 3, w2b, w0b, ring_u, wave, lane16p);
            const char* slot = ring + ((CBS_SPC * hc + s) & (HS_NSLOT - 1)) * HS_SLOT + lane16p;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * j) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * j + 1) * 1024);
                AB_MFMA3(dl[s - 4], wh, wl, uh[j], ul[j]);
            }
        }
    }
    // ---- the LayerNorm adjoint. dl = 64 dxhat (row-scaled); the two sums, then one 128-column half of the result at a time
    // (the row's statistics and its reverse edge are fetched here, not held across the loop: the loop has no register to spare)
    float mean, rstd;
    int rv;
    {
        CBS_LANE(Ls);
        CBS_ROW0(rs);
        const int64_t row = rs + Ls.r < E ? rs + Ls.r : E - 1;
        mean = LNS[row * 2];
        rstd = LNS[row * 2 + 1];
        const int64_t rl = rs + (Ls.lane & 31);
        rv = rev[rl < E ? rl : E - 1];  // row r of the e[rev] tile takes XF[rev[row0 + r]]
    }
    const float f = ABS_INV * inv;
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            dl[t][i] *= f;
            s1 += dl[t][i];
        }
    const float m1 = row_sum(s1) * (1.0f / 256.0f);
    const float m2 = row_sum(s2) * inv * (1.0f / 256.0f);
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {  // the e[rev] half first: the other one carries the dM addend, which then finds half of dl dead
        const int half = 1 - pass;
        CBS_LANE(Le);
        CBS_ROW0(re);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (half == 0) cbs_dma_rows(XF, re, E, D, tile_u, Le);
        else {
#pragma unroll
            for (int j = 0; j < 16; j++) {  // the e[rev] rows (pet_comb_s.hip)
                const int r = 2 * j + (Le.lane >> 5);
                const int64_t rr = __shfl(rv, r);
                glds16_trr(XF + rr * D + 4 * ((Le.lane & 31) ^ (r & 15)), tile_u + j * 1024);
            }
        }
        float4 dm[ADD_DM ? 16 : 1];
        if (ADD_DM && half == 0)  // dcat[p][:D] leaves as dM[p] + dcat[p][:D], the first two terms of dXF (k_dxf, pet_bwd.hip) in its order
            request_rows_addend<16>(reinterpret_cast<float4(&)[16]>(dm), Le, [&](int r) { return dM + (re + r < E ? re + r : E - 1) * D; });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float4 x[16];  // the rows, then the result in their place
        tile128_to_frag(x, tile, Le);
        const float rm = rstd * m2;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x16& a = dl[4 * half + t];
                float4& xx = x[4 * t + j];
                xx = make_float4(rstd * (a[4 * j] - m1 - (xx.x - mean) * rm), rstd * (a[4 * j + 1] - m1 - (xx.y - mean) * rm),
                                 rstd * (a[4 * j + 2] - m1 - (xx.z - mean) * rm), rstd * (a[4 * j + 3] - m1 - (xx.w - mean) * rm));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        auto out = [&](int r) { return live && re + r < E ? dcat + (re + r) * (2 * D) + D * half : nullptr; };
        if (ADD_DM && half == 0) store_rows_lines_add<16>(x, reinterpret_cast<float4(&)[16]>(dm), reinterpret_cast<float*>(tile), Le, out);
        else store_rows_lines<16>(x, reinterpret_cast<float*>(tile), Le, out);
    }
}

static inline W2 cbs_w2(const void* base, int n_tiles_dim, int k_dim) {
    const size_t n8 = (size_t)(n_tiles_dim / 32) * (k_dim / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}

// false = not served (small graphs, planes missing, or pet_config_set("emlp_s", 0)); c0g = comb0 with the LayerNorm folded in
bool comb_bwd_s(const float* dM, const float* XF, const int* rev, const float* LNS, const float* CA, const Lin& c0g, const Lin& c2,
                float* dcat, int64_t E, bool add_dm, hipStream_t st) {
    if (!emlp_s_serves(E) || !c0g.bwd2s || !c2.bwd2s || !c0g.b) return false;
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    // bwd2s operands: tiles over k_in, K = n_out
    const W2 w2b = cbs_w2(c2.bwd2s, c2.k_in, c2.n_out), w0b = cbs_w2(c0g.bwd2s, c0g.k_in, c0g.n_out);
    const int grid = (int)cdiv(E, HS_NW * WROWS);
    if (add_dm) {
        allow_big_lds(k_comb_bwd_s<true>, lds);
        k_comb_bwd_s<true><<<grid, 256, lds, st>>>(dM, XF, rev, LNS, CA, c0g.b, w2b, w0b, dcat, E);
    } else {
        allow_big_lds(k_comb_bwd_s<false>, lds);
        k_comb_bwd_s<false><<<grid, 256, lds, st>>>(dM, XF, rev, LNS, CA, c0g.b, w2b, w0b, dcat, E);
    }
    return true;
}

}  // namespace pet
