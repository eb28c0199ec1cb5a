// The message-passing combination stage (backend.py:559-575: M' = M + e + W2 silu(W0 LayerNorm([e ; e[rev]]) + b0) + b2), forward,
// in the form of k_emlp_s (pet_emlp_s.hip, rows_s.h): one-accumulator split-operand products, two desynchronised four-wave
// workgroups per CU, one weight stream per workgroup through a four-slot LDS ring requested three stages ahead. Round 5, second
// attempt: the first one (tools/experiments/pet_comb_s.hip, 1.61 against k_comb_p2's 1.67 ms) loaded the LayerNorm's weight and
// bias behind its first ring requests, gathered M and e[rev] as 32-byte pieces and stored the pre-activations as 32-byte pieces.
// Here the LayerNorm's affine part is folded into W0 (Model comb0_g: W0 diag(gamma), b0 + W0 beta; abi.hip fold_norm_s), the rows
// arrive through the wave's tile by LDS-DMA before the ring starts, the planes of all 256 columns live in registers, and the
// pre-activations leave as whole 128-byte lines through the tile with the exact vmcnt behind them.
//
//   rows      M (or the edge embedding of the neighbour species, first layer): row fragments; e[p] then e[rev[p]]: LDS-DMA through
//             the tile, one after the other; LayerNorm statistics over the 256 columns (they leave to LNS for the adjoint);
//             planes of 64 xhat in registers (16 K blocks); 64 (M + e + b2) = initial value of the out accumulators
//   chunk hc  (32 hidden units, 8 of them): a = W0g[chunk] xhat + b0g (8 stages of 2 K blocks: 48 MFMAs), saved for the adjoint
//             (k_comb_bwd_p2 reads it); u = silu(a) as planes at scale 1; out += W2[:, chunk] u (4 stages: 24 MFMAs)
//   stores    M' as whole lines through the wave's tile
#include "rows_s.h"

namespace pet {

constexpr int CB_SPC = 12, CB_NC = 2 * D / 32;

// stage 12 hc + s; wave w brings fragment w
//   s < 8:  W0g tile hc, K blocks 2 s + j (j = w >> 1), plane w & 1
//   s >= 8: W2 K block 2 hc + (s - 8) / 2, output tiles 2 th + (w >> 1) (th = (s - 8) % 2), plane w & 1
__device__ __forceinline__ void cb_request(int hc, int s, const W2& w0, const W2& w2, unsigned ring_u, int wave, unsigned lane16) {
    if (s >= CB_SPC) { s -= CB_SPC; hc += 1; }
    if (hc >= CB_NC) { hc = CB_NC - 1; s = CB_SPC - 1; }  // past the end: the last stage again (identical bytes; keeps vmcnt uniform)
    const unsigned dst = ring_u + (unsigned)((CB_SPC * hc + s) & (HS_NSLOT - 1)) * HS_SLOT + wave * 1024;
    const int pl = wave & 1, j = wave >> 1;
    if (s < 8) ab_dma_piece(pl ? w0.l : w0.h, hc * (2 * D / 16) + 2 * s + j, lane16, dst);
    else ab_dma_piece(pl ? w2.l : w2.h, (2 * ((s - 8) & 1) + j) * (2 * D / 16) + 2 * hc + ((s - 8) >> 1), lane16, dst);
}
// (vmcnt retires in order, stores included: the chunk's four pre-activation store instructions are issued between the requests of
// stages 12 hc + 10 and + 11, so the count is 2 + 4 for the stages 12 hc + 8 .. + 10; a partial tile skips store instructions)
#define CB_STAGE_SYNC(AFTER_STORES)                                                          \
    do {                                                                                     \
        if ((AFTER_STORES) && !full) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
        else if (AFTER_STORES) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");              \
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                                \
        __syncthreads();                                                                     \
    } while (0)

template <bool FIRST>
__global__ __launch_bounds__(256, 2) void k_comb_s(const float* __restrict__ XF, const int* __restrict__ rev, W2 w0,
                                                  const float* __restrict__ b0, W2 w2, const float* __restrict__ b2,
                                                  const float* __restrict__ Min, const float* __restrict__ edge_emb,
                                                  const int* __restrict__ sp_nbr, float* __restrict__ CA,
                                                  float* __restrict__ LNS, float* __restrict__ Mout, int64_t E) {
    extern __shared__ __attribute__((aligned(16))) char cb_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * HS_NW + wave) * WROWS;
    const bool live = row0 < E;
    if (!live) row0 = ((E - 1) / WROWS) * WROWS;
    const int64_t row = row0 + L.r < E ? row0 + L.r : E - 1;
    const bool valid = live && row0 + L.r < E;
    char* tile = cb_smem + wave * 16384;
    const char* ring = cb_smem + HS_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    // ---- the three row sets, all consumed before the ring starts
    dma_tile128(XF, row0, E, tile_u, L);
    int rv;
    {
        const int64_t rl = row0 + (L.lane & 31);
        rv = rev[rl < E ? rl : E - 1];  // row r of the second tile takes XF[rev[row0 + r]]
    }
    float4 oi[16];  // M + b2 (then + e): the initial value of the out accumulators
    {
        if (FIRST) load_rowfrag<16>(oi, edge_emb, (int64_t)sp_nbr[row], D, L.h);
        else load_rowfrag<16>(oi, Min, row, D, L.h);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                oi[4 * t + j].x += hs_vec(b2, t, j, 0, L.h); oi[4 * t + j].y += hs_vec(b2, t, j, 1, L.h);
                oi[4 * t + j].z += hs_vec(b2, t, j, 2, L.h); oi[4 * t + j].w += hs_vec(b2, t, j, 3, L.h);
            }
    }
    float4 x[32];  // [e ; e[rev]] as one 256-wide row fragment
    {
        float4 xa[16];
        tile128_to_frag(xa, tile, L);
        // the reads must have RETURNED before the tile is requested again: an L2-warm LDS-DMA lands after 250-400 cycles, sooner than
        // sixteen queued ds_read_b128 of a busy CU are served (the first version of this kernel waited only for their issue: its
        // results changed from run to run)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 16; j++) {  // the e[rev] rows into the (consumed) tile
            const int r = 2 * j + (L.lane >> 5);
            const int64_t rr = __shfl(rv, r);
            glds16_trr(XF + rr * 128 + 4 * ((L.lane & 31) ^ (r & 15)), tile_u + j * 1024);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {  // M + e + b2: the residuals ride in the accumulator
            oi[k].x += xa[k].x; oi[k].y += xa[k].y; oi[k].z += xa[k].z; oi[k].w += xa[k].w;
            x[k] = xa[k];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile128_to_frag(xa, tile, L);
#pragma unroll
        for (int k = 0; k < 16; k++) x[16 + k] = xa[k];
        // the initial value waits in the (consumed) tile while the statistics and the planes need the registers
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; k++) reinterpret_cast<float4*>(tile)[k * 64 + L.lane] = oi[k];
    }
    float mean, rstd;
    {
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k++) s1 += (x[k].x + x[k].y) + (x[k].z + x[k].w);
        mean = row_sum(s1) * (1.0f / 256.0f);
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            x[k].x -= mean; x[k].y -= mean; x[k].z -= mean; x[k].w -= mean;
            s2 += x[k].x * x[k].x + x[k].y * x[k].y + x[k].z * x[k].z + x[k].w * x[k].w;
        }
        rstd = rsqrtf(row_sum(s2) * (1.0f / 256.0f) + 1e-5f);  // LayerNorm eps (backend.py:95-97)
    }
    if (LNS && valid && L.h == 0) {
        LNS[row * 2] = mean;
        LNS[row * 2 + 1] = rstd;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the statistics' store: nothing but fragments in the queue from here on)
    cb_request(0, 0, w0, w2, ring_u, wave, lane16);
    cb_request(0, 1, w0, w2, ring_u, wave, lane16);
    cb_request(0, 2, w0, w2, ring_u, wave, lane16);
    f16x8 xh[16], xl[16];  // planes of 64 xhat (the LayerNorm's weight and bias are in W0g / b0g)
    {
        const float f = rstd * ABS;
#pragma unroll
        for (int kb = 0; kb < 16; kb++) {
            const float v8[8] = {x[2 * kb].x * f, x[2 * kb].y * f, x[2 * kb].z * f, x[2 * kb].w * f,
                                 x[2 * kb + 1].x * f, x[2 * kb + 1].y * f, x[2 * kb + 1].z * f, x[2 * kb + 1].w * f};
            ab_split8(v8, xh[kb], xl[kb]);
        }
    }
    f32x16 out[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4 o4 = reinterpret_cast<const float4*>(tile)[(4 * t + j) * 64 + L.lane];
            out[t][4 * j] = o4.x * ABS; out[t][4 * j + 1] = o4.y * ABS; out[t][4 * j + 2] = o4.z * ABS; out[t][4 * j + 3] = o4.w * ABS;
        }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float* const otile = reinterpret_cast<float*>(tile);  // [32][TILE32_LD] staging of the pre-activation lines
    const bool stores = CA != nullptr && live;  // wave-uniform: the chunk's four store instructions are issued
    const bool full = row0 + WROWS <= E;

#pragma unroll 1
    for (int hc = 0; hc < CB_NC; hc++) {
        f32x16 aa;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) aa[4 * j + i] = hs_vec(b0 + 32 * hc, 0, j, i, L.h) * ABQ;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            CB_STAGE_SYNC(false);
            cb_request(hc, s + 3, w0, w2, ring_u, wave, lane16);
            const char* slot = ring + ((CB_SPC * hc + s) & (HS_NSLOT - 1)) * HS_SLOT + lane16;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * j) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * j + 1) * 1024);
                AB_MFMA3(aa, wh, wl, xh[2 * s + j], xl[2 * s + j]);
            }
        }
        f32x16 u;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            aa[i] *= ABQ_INV;
            u[i] = silu_(aa[i]);
        }
        if (stores) {  // whole 128-B lines through the staging tile (8 lanes per row)
            float4 t4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) t4[q] = make_float4(aa[4 * q], aa[4 * q + 1], aa[4 * q + 2], aa[4 * q + 3]);
            store_tile32_lines(t4, otile, CA + 32 * hc, row0, E, 2 * D, L);
        }
        f16x8 uh[2], ul[2];
        ab_tile_planes(u, uh, ul);
#pragma unroll
        for (int s = 8; s < 12; s++) {
            CB_STAGE_SYNC(s < 11 && stores);  // (the fragments of stages 8 .. 10 were requested before the stores)
            cb_request(hc, s + 3, w0, w2, ring_u, wave, lane16);
            const char* slot = ring + ((CB_SPC * hc + s) & (HS_NSLOT - 1)) * HS_SLOT + lane16;
            const int kb2 = (s - 8) >> 1, th = (s - 8) & 1;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * t) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * t + 1) * 1024);
                AB_MFMA3(out[2 * th + t], wh, wl, uh[kb2], ul[kb2]);
            }
        }
    }
    // ---- M' = out / 64: whole lines through the wave's own tile
    float4 y[16];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            y[4 * t + j] = make_float4(out[t][4 * j] * ABS_INV, out[t][4 * j + 1] * ABS_INV, out[t][4 * j + 2] * ABS_INV, out[t][4 * j + 3] * ABS_INV);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    store_rows_lines<16>(y, reinterpret_cast<float*>(tile), L, [&](int r) { return live && row0 + r < E ? Mout + (row0 + r) * D : nullptr; });
}

static inline W2 cb_w2(const void* base, int n_out, int k_in) {
    const size_t n8 = (size_t)(n_out / 32) * (k_in / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}

// false = not served (small graphs, weights not packed for it, or pet_config_set("emlp_s", 0)); c0g = comb0 with the LayerNorm folded in
bool comb_s(bool first, const float* XF, const int* rev, const Lin& c0g, const Lin& c2, const float* Min, const float* edge_emb,
            const int* sp_nbr, float* CA, float* LNS, float* Mout, int64_t E, hipStream_t st) {
    if (!emlp_s_serves(E) || !c0g.fwd2s || !c2.fwd2s) return false;
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    const W2 w0 = cb_w2(c0g.fwd2s, c0g.n_out, c0g.k_in), w2 = cb_w2(c2.fwd2s, c2.n_out, c2.k_in);
    const int grid = (int)cdiv(E, HS_NW * WROWS);
    if (first) {
        allow_big_lds(k_comb_s<true>, lds);
        k_comb_s<true><<<grid, 256, lds, st>>>(XF, rev, w0, c0g.b, w2, c2.b, nullptr, edge_emb, sp_nbr, CA, LNS, Mout, E);
    } else {
        allow_big_lds(k_comb_s<false>, lds);
        k_comb_s<false><<<grid, 256, lds, st>>>(XF, rev, w0, c0g.b, w2, c2.b, Min, edge_emb, sp_nbr, CA, LNS, Mout, E);
    }
    return true;
}

}  // namespace pet
