// The message-passing combination stage (backend.py:559-575: M' = M + e + W2 silu(W0 LayerNorm([e ; e[rev]]) + b0) + b2) in the
// form of k_emlp_s (pet_emlp_s.hip): one-accumulator split-operand products, two desynchronised 4-wave workgroups per CU,
// one weight stream per workgroup through a four-slot LDS ring requested three stages ahead. Round 5.
//
//   rows      e[p]: LDS-DMA (whole rows); e[rev[p]] and M (or the edge embedding of the neighbour species, first layer): row
//             fragments straight into registers; LayerNorm over the 256 columns (its statistics leave to LNS for the
//             adjoint); planes of 64 x: the e[p] half in registers, the e[rev[p]] half parked over the e tile;
//             M + e + b2 = initial value of the out accumulators
//   chunk hc  (32 hidden units, 8 of them): a = W0[chunk] cat + b0 (8 stages of 2 K blocks: 48 MFMAs), saved for the adjoint;
//             u = silu(a) as planes at scale 1; out += W2[:, chunk] u (4 stages: 24 MFMAs)
//   stores    M' as whole lines through the wave's (dead) plane tile; the pre-activations leave as row fragments (the tile
//             is the parked half's home while the chunks run: no staging tile for them)
#include "ablk.h"

namespace pet {

constexpr int CS_NW = 4, CS_SLOT = 4096, CS_NSLOT = 4, CS_SPC = 12, CS_NC = 2 * D / 32;

// stage 12 hc + s; wave w brings fragment w
//   s < 8:  W0 tile hc, K blocks 2 s + j (j = w >> 1), plane w & 1
//   s >= 8: W2 K block 2 hc + (s - 8) / 2, output tiles 2 th + (w >> 1) (th = (s - 8) % 2), plane w & 1
__device__ __forceinline__ void cs_request(int hc, int s, const W2& w0, const W2& w2, unsigned ring_u, int wave, unsigned lane16) {
    if (s >= CS_SPC) { s -= CS_SPC; hc += 1; }
    if (hc >= CS_NC) { hc = CS_NC - 1; s = CS_SPC - 1; }  // past the end: the last stage again (identical bytes; keeps vmcnt uniform)
    const unsigned dst = ring_u + (unsigned)((CS_SPC * hc + s) & (CS_NSLOT - 1)) * CS_SLOT + wave * 1024;
    const int pl = wave & 1, j = wave >> 1;
    if (s < 8) ab_dma_piece(pl ? w0.l : w0.h, hc * (2 * D / 16) + 2 * s + j, lane16, dst);
    else ab_dma_piece(pl ? w2.l : w2.h, (2 * ((s - 8) & 1) + j) * (2 * D / 16) + 2 * hc + ((s - 8) >> 1), lane16, dst);
}
// (vmcnt retires in order, stores included: the chunk's four pre-activation stores are issued between the requests of
// stages 12 hc + 10 and + 11, so the count is 2 + 4 for the stages 12 hc + 8 .. + 10)
#define CS_STAGE_SYNC(AFTER_STORES)                                                     \
    do {                                                                                \
        if (AFTER_STORES) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");              \
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                           \
        __syncthreads();                                                                \
    } while (0)

__device__ __forceinline__ void cs_bias_tile(f32x16& acc, const float* __restrict__ b, int h) {  // scalar loads (pet_emlp_s.hip)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float lo = b[8 * j + i], hi = b[8 * j + 4 + i];
            acc[4 * j + i] = (h ? hi : lo) * ABQ;
        }
}

template <bool FIRST>
__global__ __launch_bounds__(256, 2) void k_comb_s(const float* __restrict__ XF, const int* __restrict__ rev,
                                                   const float* __restrict__ ln_g, const float* __restrict__ ln_b, W2 w0,
                                                   const float* __restrict__ b0, W2 w2, const float* __restrict__ b2,
                                                   const float* __restrict__ Min, const float* __restrict__ edge_emb,
                                                   const int* __restrict__ sp_nbr, float* __restrict__ CA,
                                                   float* __restrict__ LNS, float* __restrict__ Mout, int64_t E) {
    extern __shared__ __attribute__((aligned(16))) char cs_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * CS_NW + wave) * WROWS;
    const bool live = row0 < E;
    if (!live) row0 = ((E - 1) / WROWS) * WROWS;
    const int64_t row = row0 + L.r < E ? row0 + L.r : E - 1;
    const bool valid = live && row0 + L.r < E;
    char* tile = cs_smem + wave * 16384;
    const char* ring = cs_smem + CS_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    dma_tile128(XF, row0, E, tile_u, L);
    f32x16 out[4];
    f16x8 xph[8], xpl[8];  // planes of the e[p] half
    {
        {
            float4 mi[16];
            if (FIRST) load_rowfrag<16>(mi, edge_emb, (int64_t)sp_nbr[row], D, L.h);
            else load_rowfrag<16>(mi, Min, row, D, L.h);
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float4 b4 = *reinterpret_cast<const float4*>(b2 + 32 * t + 8 * j + 4 * L.h);
                    out[t][4 * j] = mi[4 * t + j].x + b4.x; out[t][4 * j + 1] = mi[4 * t + j].y + b4.y;
                    out[t][4 * j + 2] = mi[4 * t + j].z + b4.z; out[t][4 * j + 3] = mi[4 * t + j].w + b4.w;
                }
        }
        // (M is consumed before the e[rev] rows are requested: both row sets in flight at once, next to the accumulators and the
        // e tile's fragments, do not fit 256 registers)
        __builtin_amdgcn_sched_barrier(0);
        float4 xr[16];
        load_rowfrag<16>(xr, XF, (int64_t)rev[row], D, L.h);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cs_request(0, 0, w0, w2, ring_u, wave, lane16);
        cs_request(0, 1, w0, w2, ring_u, wave, lane16);
        cs_request(0, 2, w0, w2, ring_u, wave, lane16);
        float4 xo[16];
        tile128_to_frag(xo, tile, L);
        float s1 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {  // 64 (M + e + b2): the residuals ride in the accumulator
                const float4 e4 = xo[4 * t + j];
                out[t][4 * j] = (out[t][4 * j] + e4.x) * ABS; out[t][4 * j + 1] = (out[t][4 * j + 1] + e4.y) * ABS;
                out[t][4 * j + 2] = (out[t][4 * j + 2] + e4.z) * ABS; out[t][4 * j + 3] = (out[t][4 * j + 3] + e4.w) * ABS;
            }
#pragma unroll
        for (int k = 0; k < 16; k++) s1 += xo[k].x + xo[k].y + xo[k].z + xo[k].w + xr[k].x + xr[k].y + xr[k].z + xr[k].w;
        const float mean = row_sum(s1) * (1.0f / 256.0f);
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            float d;
            d = xo[k].x - mean; s2 += d * d; d = xo[k].y - mean; s2 += d * d;
            d = xo[k].z - mean; s2 += d * d; d = xo[k].w - mean; s2 += d * d;
            d = xr[k].x - mean; s2 += d * d; d = xr[k].y - mean; s2 += d * d;
            d = xr[k].z - mean; s2 += d * d; d = xr[k].w - mean; s2 += d * d;
        }
        const float rstd = rsqrtf(row_sum(s2) * (1.0f / 256.0f) + 1e-5f);  // LayerNorm eps (backend.py:95-97)
        if (LNS && valid && L.h == 0) {
            LNS[row * 2] = mean;
            LNS[row * 2 + 1] = rstd;
        }
        // affine LayerNorm and planes, one K block (two float4) at a time: all 64 gamma / beta loads up front would not fit
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            float v8[8];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int k = 2 * kb + q;
                const float4 ga = *reinterpret_cast<const float4*>(ln_g + 8 * k + 4 * L.h);
                const float4 ba = *reinterpret_cast<const float4*>(ln_b + 8 * k + 4 * L.h);
                v8[4 * q] = ((xo[k].x - mean) * rstd * ga.x + ba.x) * ABS; v8[4 * q + 1] = ((xo[k].y - mean) * rstd * ga.y + ba.y) * ABS;
                v8[4 * q + 2] = ((xo[k].z - mean) * rstd * ga.z + ba.z) * ABS; v8[4 * q + 3] = ((xo[k].w - mean) * rstd * ga.w + ba.w) * ABS;
            }
            ab_split8(v8, xph[kb], xpl[kb]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        // planes of 64 x of the e[rev] half over the (consumed) e tile: [kb][plane H, L][lane] f16x8 (ablk.h ab_park_planes)
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            float v8[8];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int k = 2 * kb + q;
                const float4 gb = *reinterpret_cast<const float4*>(ln_g + D + 8 * k + 4 * L.h);
                const float4 bb = *reinterpret_cast<const float4*>(ln_b + D + 8 * k + 4 * L.h);
                v8[4 * q] = ((xr[k].x - mean) * rstd * gb.x + bb.x) * ABS; v8[4 * q + 1] = ((xr[k].y - mean) * rstd * gb.y + bb.y) * ABS;
                v8[4 * q + 2] = ((xr[k].z - mean) * rstd * gb.z + bb.z) * ABS; v8[4 * q + 3] = ((xr[k].w - mean) * rstd * gb.w + bb.w) * ABS;
            }
            f16x8 h, l;
            ab_split8(v8, h, l);
            *reinterpret_cast<f16x8*>(tile + ((kb * 2 + 0) * 64 + L.lane) * 16) = h;
            *reinterpret_cast<f16x8*>(tile + ((kb * 2 + 1) * 64 + L.lane) * 16) = l;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const bool stores = CA != nullptr && live;  // wave-uniform: the chunk's four store instructions are issued
    const bool full = row0 + WROWS <= E;
    float* ca_row = CA + row * (2 * D) + 4 * L.h;

#pragma unroll 1
    for (int hc = 0; hc < CS_NC; hc++) {
        f32x16 aa;
        cs_bias_tile(aa, b0 + 32 * hc, L.h);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            CS_STAGE_SYNC(false);
            cs_request(hc, s + 3, w0, w2, ring_u, wave, lane16);
            const char* slot = ring + ((CS_SPC * hc + s) & (CS_NSLOT - 1)) * CS_SLOT + lane16;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int kb = 2 * s + j;
                const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * j) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * j + 1) * 1024);
                if (kb < 8) {
                    AB_MFMA3(aa, wh, wl, xph[kb], xpl[kb]);
                } else {
                    const f16x8 rh = *reinterpret_cast<const f16x8*>(tile + (((kb - 8) * 2 + 0) * 64 + L.lane) * 16);
                    const f16x8 rl = *reinterpret_cast<const f16x8*>(tile + (((kb - 8) * 2 + 1) * 64 + L.lane) * 16);
                    AB_MFMA3(aa, wh, wl, rh, rl);
                }
            }
        }
        f32x16 u;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            aa[i] *= ABQ_INV;
            u[i] = silu_(aa[i]);
        }
        if (stores && valid) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4*>(ca_row + 32 * hc + 8 * q) = make_float4(aa[4 * q], aa[4 * q + 1], aa[4 * q + 2], aa[4 * q + 3]);
        }
        f16x8 uh[2], ul[2];
        ab_tile_planes(u, uh, ul);
#pragma unroll
        for (int s = 8; s < 12; s++) {
            if ((s < 11 && stores) && !full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CS_STAGE_SYNC(s < 11 && stores && full);
            cs_request(hc, s + 3, w0, w2, ring_u, wave, lane16);
            const char* slot = ring + ((CS_SPC * hc + s) & (CS_NSLOT - 1)) * CS_SLOT + lane16;
            const int kb2 = (s - 8) >> 1, th = (s - 8) & 1;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * t) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * t + 1) * 1024);
                AB_MFMA3(out[2 * th + t], wh, wl, uh[kb2], ul[kb2]);
            }
        }
    }
    // ---- M' = out / 64: whole lines through the wave's own tile (the parked planes are dead)
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float* stg = reinterpret_cast<float*>(tile);  // [32][TILE_LD]
    const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
#pragma unroll
    for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                *reinterpret_cast<float4*>(stg + L.r * TILE_LD + 8 * (4 * t + j) + 4 * L.h) =
                    make_float4(out[2 * c + t][4 * j] * ABS_INV, out[2 * c + t][4 * j + 1] * ABS_INV,
                                out[2 * c + t][4 * j + 2] * ABS_INV, out[2 * c + t][4 * j + 3] * ABS_INV);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = 4 * j + rr;
            if (live && row0 + r < E)
                *reinterpret_cast<float4*>(Mout + (row0 + r) * D + 64 * c + cc) = *reinterpret_cast<const float4*>(stg + r * TILE_LD + cc);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static int g_comb_s = 1;  // pet_config_set("comb_s", 0): the one-wave-per-SIMD pipelined kernel k_comb_p2
void set_comb_s(int v) { g_comb_s = v ? 1 : 0; }

static inline W2 cs_w2(const void* base, int n_out, int k_in) {
    const size_t n8 = (size_t)(n_out / 32) * (k_in / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}

// false = not served (small graphs, weights not packed for it, or switched off)
bool comb_s(bool first, const float* XF, const int* rev, const float* ln_g, const float* ln_b, const Lin& c0, const Lin& c2,
            const float* Min, const float* edge_emb, const int* sp_nbr, float* CA, float* LNS, float* Mout, int64_t E,
            hipStream_t st) {
    if (!g_comb_s || !c0.fwd2s || !c2.fwd2s || E < 16384) return false;
    const size_t lds = CS_NW * 16384 + CS_NSLOT * CS_SLOT;
    const W2 w0 = cs_w2(c0.fwd2s, c0.n_out, c0.k_in), w2 = cs_w2(c2.fwd2s, c2.n_out, c2.k_in);
    const int grid = (int)cdiv(E, CS_NW * WROWS);
    if (first) {
        allow_big_lds(k_comb_s<true>, lds);
        k_comb_s<true><<<grid, 256, lds, st>>>(XF, rev, ln_g, ln_b, w0, c0.b, w2, c2.b, nullptr, edge_emb, sp_nbr, CA, LNS, Mout, E);
    } else {
        allow_big_lds(k_comb_s<false>, lds);
        k_comb_s<false><<<grid, 256, lds, st>>>(XF, rev, ln_g, ln_b, w0, c0.b, w2, c2.b, Min, edge_emb, sp_nbr, CA, LNS, Mout, E);
    }
    return true;
}

}  // namespace pet
