// k_ablk_bwd1: the fused attention adjoint of the 32-slot tiles, round 6. The arithmetic is k_ablk_bwd<1, LN>'s (ablk_bwd.h: read
// that header for the algebra and the layout conventions); what changed is where the operands live and how the weight stream
// is fed, after measuring what the GEMM phases of that kernel (dAO, the Q, K, V recomputation, dXn: 53 % of its time) wait for:
// a stage of 24 MFMAs (768 matrix-pipe cycles) took ~1 750 cycles -- ~400 waiting for the stage's LDS-DMA (requested ONE stage
// ahead into a two-slot ring: an L2 round trip under load is longer than a stage), ~230 in the workgroup barrier, ~250 for the
// LDS reads of the fragments behind it.
//   * The planes of the normalised rows xhat live in REGISTERS (64 of them; the kernel runs one wave per SIMD for its LDS anyway,
//     so registers up to 512 are free), not in the wave's 16-KB LDS tile: the Q, K, V recomputation reads no LDS for them, the
//     norm adjoint at the end takes xhat from the same registers,
//   * and the four freed 16-KB tiles join the ring: SIX slots of 16 KB, every stage requested THREE stages ahead, the wait at a
//     stage boundary is the exact count of what may still be in flight (vmcnt retires in order; stages are 3 or 4 pieces per wave).
//   * Biases come through the scalar cache (a vector load between two stages would sit in the in-order queue).
#pragma once
#include "ablk_bwd.h"

namespace pet {

constexpr int AB1_NSLOT = 6, AB1_AHEAD = 3, AB1_NSTAGE = 32;
// ring slot k: 0, 1 = the ring region behind the tiles; 2 .. 5 = the X-row tiles of waves 0 .. 3 (dead once every wave holds its
// xhat planes in registers)
__device__ __forceinline__ unsigned ab1_slot(unsigned smem_u, int k) {
    return k < 2 ? smem_u + 4u * 32768u + (unsigned)k * AB_SLOT_B : smem_u + (unsigned)(k - 2) * 32768u;
}
// stage g of ab_bwd_request's stream into ring slot g % 6; past the end: a stage of the same kind as the stream would continue
// with (the Q, K, V stages of a head pair: 3 pieces per wave), into a slot nobody reads -- keeps the wait counts uniform
__device__ __forceinline__ void ab1_request(int g, const W2& wqkv, const W2& wot, const W2& wqkvt, unsigned smem_u, int wave,
                                            unsigned lane16) {
    const unsigned dst = ab1_slot(smem_u, g % AB1_NSLOT);
    const int ge = g < AB1_NSTAGE ? g : 4 + (g - AB1_NSTAGE);
    const int r = ge < 4 ? -1 : (ge - 4) % 7, hp = ge < 4 ? 0 : (ge - 4) / 7;
    if (r >= 0 && r < 4) {  // QKV: 12 pieces, 3 per wave
        const int kb0 = 2 * r;
#pragma unroll
        for (int p0 = 0; p0 < 12; p0 += 4) {
            const int p = p0 + wave;
            const int j = p / 6, f = p % 6;
            ab_dma_piece((f & 1) ? wqkv.l : wqkv.h, 32 * (f >> 1) + hp * 8 + kb0 + j, lane16, dst + p * 1024);
        }
    } else {  // 16 pieces, 4 per wave: j * 8 + 2 t + plane
#pragma unroll
        for (int p0 = 0; p0 < 16; p0 += 4) {
            const int p = p0 + wave;
            const int j = p >> 3, t = (p >> 1) & 3, pl = p & 1;
            if (r < 0) {
                ab_dma_piece(pl ? wot.l : wot.h, t * 8 + 2 * ge + j, lane16, dst + p * 1024);
            } else {
                const int st = 2 * (r - 4) + j;
                ab_dma_piece(pl ? wqkvt.l : wqkvt.h, t * 24 + 8 * (st >> 1) + 2 * hp + (st & 1), lane16, dst + p * 1024);
            }
        }
    }
}
// the stage's fragments have landed: everything but the two stages requested after it (N = their pieces of this wave) may
// still be in flight; then the workgroup barrier
#define AB1_STAGE_SYNC(N)                                                       \
    do {                                                                        \
        asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory");        \
        __syncthreads();                                                        \
    } while (0)

// accumulators of a token-form tile initialised with 4096 x bias, the bias read through the SCALAR cache (wave-uniform
// addresses, both halves of a column group, selected by lane half)
__device__ __forceinline__ void ab1_bias_tile(f32x16& acc, const float* __restrict__ b, int h) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float lo = b[8 * j + i], hi = b[8 * j + 4 + i];
            acc[4 * j + i] = (h ? hi : lo) * ABQ;
        }
}

// xhat = (x - mean) rstd (RMSNorm: mean = 0) as planes of 64 xhat in registers; returns rstd (ab_park_xhat's arithmetic)
template <bool LN>
__device__ __forceinline__ float ab1_xhat_planes(float4 (&x)[16], f16x8 (&xh)[8], f16x8 (&xl)[8]) {
    if (LN) {
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) sm += (x[k].x + x[k].y) + (x[k].z + x[k].w);
        const float mean = row_sum(sm) * (1.0f / 128.0f);
#pragma unroll
        for (int k = 0; k < 16; k++) { x[k].x -= mean; x[k].y -= mean; x[k].z -= mean; x[k].w -= mean; }
    }
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) ss += x[k].x * x[k].x + x[k].y * x[k].y + x[k].z * x[k].z + x[k].w * x[k].w;
    const float rstd = rsqrtf(row_sum(ss) * (1.0f / 128.0f) + (LN ? 1e-5f : 1.1920928955078125e-07f));
#pragma unroll
    for (int k = 0; k < 16; k++) { x[k].x *= rstd; x[k].y *= rstd; x[k].z *= rstd; x[k].w *= rstd; }
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
        const float v[8] = {x[2 * kb].x * ABS, x[2 * kb].y * ABS, x[2 * kb].z * ABS, x[2 * kb].w * ABS,
                            x[2 * kb + 1].x * ABS, x[2 * kb + 1].y * ABS, x[2 * kb + 1].z * ABS, x[2 * kb + 1].w * ABS};
        ab_split8(v, xh[kb], xl[kb]);
    }
    return rstd;
}
// ab_norm_adjoint_planes with xhat from the register planes
template <bool LN>
__device__ __forceinline__ void ab1_norm_adjoint(float4 (&w)[16], const f16x8 (&xph)[8], const f16x8 (&xpl)[8], float rstd) {
    float4 xh[16];
    float dot = 0.f;
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
        const f16x8 h = xph[kb], l = xpl[kb];
        xh[2 * kb] = make_float4(((float)h[0] + (float)l[0]) * ABS_INV, ((float)h[1] + (float)l[1]) * ABS_INV,
                                 ((float)h[2] + (float)l[2]) * ABS_INV, ((float)h[3] + (float)l[3]) * ABS_INV);
        xh[2 * kb + 1] = make_float4(((float)h[4] + (float)l[4]) * ABS_INV, ((float)h[5] + (float)l[5]) * ABS_INV,
                                     ((float)h[6] + (float)l[6]) * ABS_INV, ((float)h[7] + (float)l[7]) * ABS_INV);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) dot += xh[k].x * w[k].x + xh[k].y * w[k].y + xh[k].z * w[k].z + xh[k].w * w[k].w;
    const float md = row_sum(dot) * (1.0f / 128.0f);
    float sw = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        w[k].x = rstd * (w[k].x - xh[k].x * md); w[k].y = rstd * (w[k].y - xh[k].y * md);
        w[k].z = rstd * (w[k].z - xh[k].z * md); w[k].w = rstd * (w[k].w - xh[k].w * md);
        sw += (w[k].x + w[k].y) + (w[k].z + w[k].w);
    }
    if (LN) {
        const float mw = row_sum(sw) * (1.0f / 128.0f);
#pragma unroll
        for (int k = 0; k < 16; k++) { w[k].x -= mw; w[k].y -= mw; w[k].z -= mw; w[k].w -= mw; }
    }
}

template <bool LN>
__global__ __launch_bounds__(256) void k_ablk_bwd1(
    const float* __restrict__ X, const float* __restrict__ dX1, const float* __restrict__ dOC, W2 wqkv,
    const float* __restrict__ bqkv, W2 wot, W2 wqkvt, const float* __restrict__ fc, const int4* __restrict__ desc, int n_list,
    int64_t E, float qscale, float scale, float* __restrict__ dXin, float* __restrict__ dbias) {
    extern __shared__ __attribute__((aligned(16))) char ab_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = 4;
    int li = blockIdx.x * NW + wave;
    const bool live = li < n_list;
    li = live ? li : n_list - 1;
    const AbAtom a(desc + 2 * (size_t)li, E);
    // per wave: 16 KB for the incoming X rows (a ring slot afterwards) | 16 KB incoming adjoint rows, then dAO (row fragments)
    char* tile = ab_smem + wave * 32768;
    char* tileB = tile + 16384;
    const unsigned smem_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ab_smem);
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    float bias[1][16];
    ab_key_bias<1>(bias, a, fc, L);  // (its loads are consumed before the first request: nothing but row loads and fragments queue up)
    asm volatile("" ::"v"(bias[0][0]), "v"(bias[0][15]));
    ab_dma_rows<1>(X, a, tile_u, L);
    // incoming adjoint: dX1 rows of the neighbours, dOC row of the centre token
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int r = 2 * j + (L.lane >> 5);
        const int p = (L.lane & 31) ^ (r & 15);
        int s = r;
        s = s < a.T ? s : a.T - 1;
        const float* src = a.centre(s) ? dOC + (int64_t)a.atom(s) * D : dX1 + a.edge(s) * D;
        glds16_trr(src + 4 * p, tile_u + 16384 + j * 1024);
    }
    ab1_request(0, wqkv, wot, wqkvt, smem_u, wave, lane16);   // slots 0, 1: the ring region
    ab1_request(1, wqkv, wot, wqkvt, smem_u, wave, lane16);
    const AbSel sel1 = ab_selectors(L, 1.0f);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the rows (stages 0 and 1: 4 pieces per wave each, behind them)
    f16x8 xph[8], xpl[8];
    float rstd;
    {
        float4 x[16];
        tile128_to_frag(x, tile, L);
        rstd = ab1_xhat_planes<LN>(x, xph, xpl);
    }
    // ---- dAO = dY Wo (token form), parked as row fragments [kg][lane] over the rows it came from
    float inv_sc;
    {
        float4 d[16];
        float m = 0.f;
        tile128_to_frag(d, tileB, L);
        const bool lv = L.r < a.T;
#pragma unroll
        for (int kg = 0; kg < 16; kg++) {
            if (!lv) d[kg] = make_float4(0.f, 0.f, 0.f, 0.f);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(d[kg].x), fabsf(d[kg].y))), fmaxf(fabsf(d[kg].z), fabsf(d[kg].w)));
        }
        const bool gb = L.r >= a.TA;
        float ma = gb ? 0.f : m, mb = gb ? m : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            ma = fmaxf(ma, __shfl_xor(ma, o));
            mb = fmaxf(mb, __shfl_xor(mb, o));
        }
        m = gb ? mb : ma;
        int e = ((__float_as_int(m) >> 23) & 0xff) + 2;
        e = e > 253 ? 253 : e;
        e = e < 16 ? 16 : e;
        const float sc = __int_as_float((254 - e) << 23) * ABS;
        inv_sc = __int_as_float(e << 23);
        __syncthreads();  // every wave holds its X rows in registers: their tiles are ring slots 2 .. 5 from here on
        ab1_request(2, wqkv, wot, wqkvt, smem_u, wave, lane16);
        f32x16 da[4];
#pragma unroll
        for (int t = 0; t < 4; t++) da[t] = ab_zero();
#define AB1_DAO_STAGE(G, NWAIT)                                                                        \
    {                                                                                                  \
        AB1_STAGE_SYNC(NWAIT);                                                                         \
        ab1_request(G + AB1_AHEAD, wqkv, wot, wqkvt, smem_u, wave, lane16);                            \
        const char* slot = ab_smem + (ab1_slot(smem_u, G % AB1_NSLOT) - smem_u) + lane16;              \
        _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                \
            const int kb = 2 * G + j;                                                                  \
            f16x8 wh[4], wl[4];                                                                        \
            _Pragma("unroll") for (int t = 0; t < 4; t++) {                                            \
                wh[t] = *reinterpret_cast<const f16x8*>(slot + (8 * j + 2 * t) * 1024);                \
                wl[t] = *reinterpret_cast<const f16x8*>(slot + (8 * j + 2 * t + 1) * 1024);            \
            }                                                                                          \
            const float v8[8] = {d[2 * kb].x * sc, d[2 * kb].y * sc, d[2 * kb].z * sc, d[2 * kb].w * sc,   \
                                 d[2 * kb + 1].x * sc, d[2 * kb + 1].y * sc, d[2 * kb + 1].z * sc, d[2 * kb + 1].w * sc}; \
            f16x8 dh, dl;                                                                              \
            ab_split8(v8, dh, dl);                                                                     \
            _Pragma("unroll") for (int t = 0; t < 4; t++) AB_MFMA3(da[t], wh[t], wl[t], dh, dl);       \
        }                                                                                              \
    }
        // in flight behind stage g when it is waited for: stages g + 1, g + 2 (4 pieces per wave for the Wo^T and Wqkv^T
        // stages, 3 for the Q, K, V stages): g = 0: 4 + 4, 1: 4 + 4, 2: 4 + 3, 3: 3 + 3
        AB1_DAO_STAGE(0, 8) AB1_DAO_STAGE(1, 8) AB1_DAO_STAGE(2, 7) AB1_DAO_STAGE(3, 6)
#undef AB1_DAO_STAGE
        float m2 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) m2 = fmaxf(m2, fabsf(da[t][i]));
        float m2a = gb ? 0.f : m2, m2b = gb ? m2 : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            m2a = fmaxf(m2a, __shfl_xor(m2a, o));
            m2b = fmaxf(m2b, __shfl_xor(m2b, o));
        }
        m2 = (gb ? m2b : m2a) * ABQ_INV;
        int e2 = ((__float_as_int(m2) >> 23) & 0xff) + 2;
        e2 = e2 > 253 ? 253 : e2;
        e2 = e2 < 16 ? 16 : e2;
        const float s2 = __int_as_float((254 - e2) << 23) * ABS_INV;
        inv_sc *= __int_as_float(e2 << 23);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                *reinterpret_cast<float4*>(tileB + ((4 * t + j) * 64 + L.lane) * 16) =
                    make_float4(da[t][4 * j] * s2, da[t][4 * j + 1] * s2, da[t][4 * j + 2] * s2, da[t][4 * j + 3] * s2);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");

    f32x16 dxn[4];
#pragma unroll
    for (int t = 0; t < 4; t++) dxn[t] = ab_zero();
    float db = 0.f;
    constexpr float LN2 = 0.6931471805599453f;

#pragma unroll 1
    for (int hp = 0; hp < 4; hp++) {
        const int gbase = 4 + 7 * hp;
        // ---- Q^T, K^T, V^T of the head pair (token form), as in the forward
        f32x16 q, k, v;
        ab1_bias_tile(q, bqkv + 32 * hp, L.h);
        ab1_bias_tile(k, bqkv + D + 32 * hp, L.h);
        ab1_bias_tile(v, bqkv + 2 * D + 32 * hp, L.h);
#define AB1_QKV_STAGE(SG, NWAIT)                                                                       \
    {                                                                                                  \
        const int g = gbase + SG;                                                                      \
        AB1_STAGE_SYNC(NWAIT);                                                                         \
        ab1_request(g + AB1_AHEAD, wqkv, wot, wqkvt, smem_u, wave, lane16);                            \
        const char* slot = ab_smem + (ab1_slot(smem_u, g % AB1_NSLOT) - smem_u) + lane16;              \
        _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                \
            const int kb = 2 * SG + j;                                                                 \
            const f16x8 wqh = *reinterpret_cast<const f16x8*>(slot + (6 * j + 0) * 1024);              \
            const f16x8 wql = *reinterpret_cast<const f16x8*>(slot + (6 * j + 1) * 1024);              \
            const f16x8 wkh = *reinterpret_cast<const f16x8*>(slot + (6 * j + 2) * 1024);              \
            const f16x8 wkl = *reinterpret_cast<const f16x8*>(slot + (6 * j + 3) * 1024);              \
            const f16x8 wvh = *reinterpret_cast<const f16x8*>(slot + (6 * j + 4) * 1024);              \
            const f16x8 wvl = *reinterpret_cast<const f16x8*>(slot + (6 * j + 5) * 1024);              \
            AB_MFMA3(q, wqh, wql, xph[kb], xpl[kb]);                                                   \
            AB_MFMA3(k, wkh, wkl, xph[kb], xpl[kb]);                                                   \
            AB_MFMA3(v, wvh, wvl, xph[kb], xpl[kb]);                                                   \
        }                                                                                              \
    }
        // (r = 0: stages r = 1, 2 behind it: 3 + 3; r = 1: 3 + 3; r = 2: 3 + 4; r = 3: 4 + 4)
        AB1_QKV_STAGE(0, 6) AB1_QKV_STAGE(1, 6) AB1_QKV_STAGE(2, 7) AB1_QKV_STAGE(3, 8)
#undef AB1_QKV_STAGE
        // ---- operand planes: token form (index = head of the pair) and feature form (index = token K block)
        f16x8 qh[2], ql[2], kH[2], kL[2], vH[2], vL[2], dah[2], dal[2];
        f16x8 qfH[2], qfL[2], kfH[2], kfL[2], dfH[2], dfL[2];
        ab_tile_planes(q, qscale * ABS_INV, qh, ql);
        ab_tile_planes(k, ABS_INV, kH, kL);
        ab_tile_planes(v, ABS_INV, vH, vL);
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const float4 d0 = *reinterpret_cast<const float4*>(tileB + ((4 * hp + 2 * b) * 64 + L.lane) * 16);
            const float4 d1 = *reinterpret_cast<const float4*>(tileB + ((4 * hp + 2 * b + 1) * 64 + L.lane) * 16);
            const float v8[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            ab_split8(v8, dah[b], dal[b]);
        }
        ab_transpose(qh, ql, sel1, qfH, qfL);
        ab_transpose(kH, kL, sel1, kfH, kfL);
        ab_transpose(dah, dal, sel1, dfH, dfL);
        f32x16 dq, dk, dv;  // token-form tiles of the pair: registers 8 hd .. 8 hd + 7 from head hd
        f32x16 dkh[2], dvh[2];
#pragma unroll
        for (int hd = 0; hd < 2; hd++) { dkh[hd] = ab_zero(); dvh[hd] = ab_zero(); }
        {
            // the two heads of the pair side by side
            f32x16 s[2], dp[2];
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                s[hd] = ab_zero();
                dp[hd] = ab_zero();
                AB_MFMA3(s[hd], kH[hd], kL[hd], qh[hd], ql[hd]);
                AB_MFMA3(dp[hd], vH[hd], vL[hd], dah[hd], dal[hd]);
            }
            float mx[2], sum[2], dl[2];
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                mx[hd] = -INFINITY;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    s[hd][i] = fmaf(s[hd][i], ABQ_INV, bias[0][i]);
                    mx[hd] = fmaxf(mx[hd], s[hd][i]);
                }
            }
#pragma unroll
            for (int hd = 0; hd < 2; hd++) mx[hd] = fmaxf(mx[hd], __shfl_xor(mx[hd], 32));
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                sum[hd] = 0.f;
                dl[hd] = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float p = __builtin_amdgcn_exp2f(s[hd][i] - mx[hd]);
                    s[hd][i] = p;
                    sum[hd] += p;
                    dl[hd] = fmaf(p, dp[hd][i], dl[hd]);
                }
            }
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                sum[hd] += __shfl_xor(sum[hd], 32);
                dl[hd] += __shfl_xor(dl[hd], 32);
            }
            f32x16 dqh[2];
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                const float inv = __builtin_amdgcn_rcpf(sum[hd]);
                const float delta = dl[hd] * inv * ABQ_INV;
                const float inv64 = inv * ABS;
                dqh[hd] = ab_zero();
                f32x16 ds;  // 64 dS^T
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float pn = s[hd][i] * inv64;  // 64 P^T
                    s[hd][i] = pn;
                    ds[i] = pn * fmaf(dp[hd][i], ABQ_INV, -delta);
                }
                f16x8 pth[2], ptl[2], sth[2], stl[2];
                ab_tile_planes(s[hd], pth, ptl);
                ab_tile_planes(ds, sth, stl);
                // dQ^T += K^T dS^T
#pragma unroll
                for (int b = 0; b < 2; b++) AB_MFMA3(dqh[hd], kfH[b], kfL[b], sth[b], stl[b]);
                // the (query, key) forms: P and dS with lane = key, registers = queries
                f16x8 ph[2], pl[2], sh[2], sl[2];
                ab_transpose(pth, ptl, sel1, ph, pl);
                db += ab_transpose_sum(sth, stl, sel1, sh, sl);
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    AB_MFMA3(dkh[hd], qfH[b], qfL[b], sh[b], sl[b]);
                    AB_MFMA3(dvh[hd], dfH[b], dfL[b], ph[b], pl[b]);
                }
            }
#pragma unroll
            for (int hd = 0; hd < 2; hd++)
#pragma unroll
                for (int j = 0; j < 8; j++) dq[8 * hd + j] = dqh[hd][8 * hd + j];
        }
#pragma unroll
        for (int hd = 0; hd < 2; hd++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                dk[8 * hd + j] = dkh[hd][8 * hd + j];
                dv[8 * hd + j] = dvh[hd][8 * hd + j];
            }
        // ---- dXn^T += Wqkv^T [dQ; dK; dV]^T: K blocks 2 hp, 2 hp + 1 of each of the three parts
        f16x8 gh[3][2], gl[3][2];
        ab_tile_planes(dq, scale * ABS_INV, gh[0], gl[0]);
        ab_tile_planes(dk, LN2 * ABS_INV, gh[1], gl[1]);
        ab_tile_planes(dv, ABS_INV, gh[2], gl[2]);
#define AB1_DXN_STAGE(XX, NWAIT)                                                                       \
    {                                                                                                  \
        const int g = gbase + 4 + XX;                                                                  \
        AB1_STAGE_SYNC(NWAIT);                                                                         \
        ab1_request(g + AB1_AHEAD, wqkv, wot, wqkvt, smem_u, wave, lane16);                            \
        const char* slot = ab_smem + (ab1_slot(smem_u, g % AB1_NSLOT) - smem_u) + lane16;              \
        _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                \
            const int st = 2 * XX + j, part = st >> 1, b = st & 1;                                     \
            f16x8 wh[4], wl[4];                                                                        \
            _Pragma("unroll") for (int t = 0; t < 4; t++) {                                            \
                wh[t] = *reinterpret_cast<const f16x8*>(slot + (8 * j + 2 * t) * 1024);                \
                wl[t] = *reinterpret_cast<const f16x8*>(slot + (8 * j + 2 * t + 1) * 1024);            \
            }                                                                                          \
            _Pragma("unroll") for (int t = 0; t < 4; t++) AB_MFMA3(dxn[t], wh[t], wl[t], gh[part][b], gl[part][b]); \
        }                                                                                              \
    }
        // (r = 4: stages r = 5, 6 behind it: 4 + 4; r = 5: 4 + 3 (the next pair's first Q, K, V stage); r = 6: 3 + 3)
        AB1_DXN_STAGE(0, 8) AB1_DXN_STAGE(1, 7) AB1_DXN_STAGE(2, 6)
#undef AB1_DXN_STAGE
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stand-in requests behind the last stage
    // ---- key-bias gradient (summed over the heads; one writer per edge and layer)
    {
        const float vb = (db + __shfl_xor(db, 32)) * (inv_sc * ABS_INV);  // the transposed planes held 64 dS
        const int key = L.r;
        if (live && L.h == 0 && key < a.T && !a.centre(key)) dbias[a.edge(key)] = vb;
    }
    // ---- norm adjoint, residual, whole-line stores
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float* stg = reinterpret_cast<float*>(tileB);  // staging for whole-line stores: the dAO rows are dead
    {
        float4 w[16];
        const float f = ABQ_INV * inv_sc;  // (W_qkv^T carries the norm's weight: dxn is the adjoint w.r.t. xhat)
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                w[4 * t + j] = make_float4(dxn[t][4 * j] * f, dxn[t][4 * j + 1] * f, dxn[t][4 * j + 2] * f, dxn[t][4 * j + 3] * f);
        const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
        float4 dr[2][8];  // the residual (dX1 rows) in the store's shape, requested before the norm adjoint's arithmetic
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int s = 4 * j + rr;
                dr[c][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s < a.T && !a.centre(s)) dr[c][j] = *reinterpret_cast<const float4*>(dX1 + a.edge(s) * D + 64 * c + cc);
            }
        ab1_norm_adjoint<LN>(w, xph, xpl, rstd);
#pragma unroll
        for (int c = 0; c < 2; c++) {
#pragma unroll
            for (int kg = 0; kg < 8; kg++)
                *reinterpret_cast<float4*>(stg + L.r * TILE_LD + 8 * kg + 4 * L.h) = w[8 * c + kg];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = 4 * j + rr, s = r;
                if (live && s < a.T) {
                    float4 o4 = *reinterpret_cast<const float4*>(stg + r * TILE_LD + cc);
                    o4.x += dr[c][j].x; o4.y += dr[c][j].y; o4.z += dr[c][j].z; o4.w += dr[c][j].w;
                    float* dst = dXin + (a.centre(s) ? E + a.atom(s) : a.edge(s)) * D;
                    *reinterpret_cast<float4*>(dst + 64 * c + cc) = o4;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace pet
