// The fused attention adjoint k_ablk_bwd (see pet_ablk.hip for the forward, the layout conventions and the weight ring).
// A header because the kernel is compiled in two translation units: the 64-slot tiles (NQ = 2) in pet_ablk.hip with the
// library's flags, the 32-slot tiles (NQ = 1) in pet_ablk_bwd1.hip with the matrix products in VGPR form (build.py).
#pragma once
#include "ablk.h"

namespace pet {

// adjoint stages (slots of 16 KB): 0 .. 3 Wo^T blocks (kb = 2 g + j: 8 fragments each); then per head pair seven stages:
// four QKV stages as in the forward, three Wqkv^T stages (steps st = 2 x + j: K block 8 (st / 2) + 2 hp + st % 2, 8 fragments)
constexpr int AB_SLOT_B = 16384;
template <int NW>
__device__ __forceinline__ void ab_bwd_request(int g, const W2& wqkv, const W2& wot, const W2& wqkvt, unsigned ring_u,
                                               int wave, unsigned lane16) {
#ifdef AB_ABL_NODMA
    if (g > 0) return;
#endif
    const unsigned dst = ring_u + (unsigned)(g & 1) * AB_SLOT_B;
    const int r = g < 4 ? -1 : (g - 4) % 7, hp = g < 4 ? 0 : (g - 4) / 7;
    if (r >= 0 && r < 4) {  // QKV: 12 pieces
        const int kb0 = 2 * r;
#pragma unroll
        for (int p0 = 0; p0 < 12; p0 += NW) {
            const int p = p0 + wave;
            if (p < 12) {
                const int j = p / 6, f = p % 6;
                ab_dma_piece((f & 1) ? wqkv.l : wqkv.h, 32 * (f >> 1) + hp * 8 + kb0 + j, lane16, dst + p * 1024);
            }
        }
    } else {  // 16 pieces: j * 8 + 2 t + plane
#pragma unroll
        for (int p0 = 0; p0 < 16; p0 += NW) {
            const int p = p0 + wave;
            const int j = p >> 3, t = (p >> 1) & 3, pl = p & 1;
            if (r < 0) {
                ab_dma_piece(pl ? wot.l : wot.h, t * 8 + 2 * g + j, lane16, dst + p * 1024);
            } else {
                const int st = 2 * (r - 4) + j;
                ab_dma_piece(pl ? wqkvt.l : wqkvt.h, t * 24 + 8 * (st >> 1) + 2 * hp + (st & 1), lane16, dst + p * 1024);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// adjoint: (dX1 | dOC) -> dXin, key-bias gradient. Q, K, V are recomputed from the layer input X.
//
//   dAO  = dY Wo                                  token form; its feature form by TRANSPOSITION ON THE MATRIX CORE:
//   a token-form tile (lane = token, regs = features) fed as the A operand against a selection matrix comes out as
//   C[token][feature] = lane = feature, regs = tokens; the planes are fp16 numbers, so two MFMAs per plane move them
//   exactly (ab_transpose).
//   S^T  = K Q^T, P^T = soft-max over the keys    (as in the forward)
//   dP^T = V dAO^T          A = V (token form),  B = dAO (token form)
//   dS^T = P^T (dP^T - delta),  delta = sum_keys P^T dP^T
//   dQ^T = K^T dS^T         A = K (feature form),  B = dS^T (the C registers)      -> token form
//   dK^T = Q^T dS           A = Q (feature form),  B = dS (transposed tile)        -> token form
//   dV^T = dAO^T P          A = dAO (feature form), B = P (transposed tile)        -> token form
//   dXn^T += Wqkv^T [dQ; dK; dV]^T of the head pair, then the norm adjoint and the residual.
// The incoming rows are scaled by ONE power of two per atom (their largest entry in [0.25, 0.5)): the sums over queries mix
// rows, so a per-row scale as in the row kernels would not factor out. Slots past the atom's last token get a zero
// adjoint row, which removes them from every sum over queries.
// ---------------------------------------------------------------------------------------------
template <int NQ, bool LN>
__global__ __launch_bounds__(256) void k_ablk_bwd(
    const float* __restrict__ X, const float* __restrict__ dX1, const float* __restrict__ dOC,
    const float* __restrict__ gamma, const float* __restrict__ beta, W2 wqkv, const float* __restrict__ bqkv, W2 wot,
    W2 wqkvt, const float* __restrict__ fc, const int4* __restrict__ desc, int n_list, int64_t E, float qscale, float scale,
    float* __restrict__ dXin, float* __restrict__ dbias) {
    extern __shared__ __attribute__((aligned(16))) char ab_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NW = 4 / NQ;  // waves per workgroup: NW x NQ x 32 KB of tiles + the 32 KB ring = all of the CU's LDS
    int li = blockIdx.x * NW + wave;
    const bool live = li < n_list;
    li = live ? li : n_list - 1;
    const AbAtom a(desc + 2 * (size_t)li, E);
    // per wave: NQ x 16 KB planes of the normalised rows | NQ x 16 KB incoming adjoint rows, then dAO (row fragments)
    char* tile = ab_smem + wave * (NQ * 32768);
    char* tileB = tile + NQ * 16384;
    const char* ring = ab_smem + NW * NQ * 32768;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    AB_T0();
    ab_dma_rows<NQ>(X, a, tile_u, L);
    // incoming adjoint: dX1 rows of the neighbours, dOC row of the centre token
#pragma unroll
    for (int tq = 0; tq < NQ; tq++)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int r = 2 * j + (L.lane >> 5);
            const int p = (L.lane & 31) ^ (r & 15);
            int s = 32 * tq + r;
            s = s < a.T ? s : a.T - 1;
            const float* src = a.centre(s) ? dOC + (int64_t)a.atom(s) * D : dX1 + a.edge(s) * D;
            glds16_trr(src + 4 * p, tile_u + NQ * 16384 + tq * 16384 + j * 1024);
        }
    ab_bwd_request<NW>(0, wqkv, wot, wqkvt, ring_u, wave, lane16);
    float bias[NQ][16];
    ab_key_bias<NQ>(bias, a, fc, L);
    const AbSel sel1 = ab_selectors(L, 1.0f);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AB_T(8);
    // (the norm's weight and bias are folded into W_qkv -- Model::qkv_g, abi.hip fold_norm_s --: the parked planes are xhat,
    // the operand of the Q, K, V recomputation and, at the end, the xhat of the norm adjoint: the layer input is read ONCE)
    float rstd[NQ];
#pragma unroll
    for (int tq = 0; tq < NQ; tq++) {
        float4 x[16];
        tile128_to_frag(x, tile + tq * 16384, L);
        rstd[tq] = ab_park_xhat<LN>(x, tile + tq * 16384, L);
    }
    AB_T(9);
    // ---- dAO = dY Wo (token form), parked as row fragments [kg][lane] over the rows it came from
    float inv_sc;  // inverse of the power-of-two scale of this lane's atom (the lane is a token in every place it is used)
    {
        float4 d[NQ][16];
        float m = 0.f;
#pragma unroll
        for (int tq = 0; tq < NQ; tq++) {
            tile128_to_frag(d[tq], tileB + tq * 16384, L);
            const bool live = 32 * tq + L.r < a.T;
#pragma unroll
            for (int kg = 0; kg < 16; kg++) {
                if (!live) d[tq][kg] = make_float4(0.f, 0.f, 0.f, 0.f);
                m = fmaxf(fmaxf(m, fmaxf(fabsf(d[tq][kg].x), fabsf(d[tq][kg].y))),
                          fmaxf(fabsf(d[tq][kg].z), fabsf(d[tq][kg].w)));
            }
        }
        // one scale per ATOM: the sums over queries stay inside an atom, so the two atoms of a paired tile keep their own
        // scale and an atom's arithmetic does not depend on what it was paired with
        const bool gb = NQ == 1 && L.r >= a.TA;
        float ma = gb ? 0.f : m, mb = gb ? m : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            ma = fmaxf(ma, __shfl_xor(ma, o));
            mb = fmaxf(mb, __shfl_xor(mb, o));
        }
        m = gb ? mb : ma;
        int e = ((__float_as_int(m) >> 23) & 0xff) + 2;  // largest entry of the atom's scaled rows in [0.25, 0.5)
        e = e > 253 ? 253 : e;
        e = e < 16 ? 16 : e;  // all-zero rows: keep the scale finite
        const float sc = __int_as_float((254 - e) << 23) * ABS;  // ... and the planes hold 64 x that
        inv_sc = __int_as_float(e << 23);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        f32x16 da[NQ][4];
#pragma unroll
        for (int tq = 0; tq < NQ; tq++)
#pragma unroll
            for (int t = 0; t < 4; t++) da[tq][t] = ab_zero();
#pragma unroll
        for (int g = 0; g < 4; g++) {
            AB_STAGE_SYNC();
            ab_bwd_request<NW>(g + 1, wqkv, wot, wqkvt, ring_u, wave, lane16);
            const char* slot = ring + (g & 1) * AB_SLOT_B + lane16;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int kb = 2 * g + j;
                f16x8 wh[4], wl[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    wh[t] = *reinterpret_cast<const f16x8*>(slot + (8 * j + 2 * t) * 1024);
                    wl[t] = *reinterpret_cast<const f16x8*>(slot + (8 * j + 2 * t + 1) * 1024);
                }
#pragma unroll
                for (int tq = 0; tq < NQ; tq++) {
                    const float v8[8] = {d[tq][2 * kb].x * sc, d[tq][2 * kb].y * sc, d[tq][2 * kb].z * sc, d[tq][2 * kb].w * sc,
                                         d[tq][2 * kb + 1].x * sc, d[tq][2 * kb + 1].y * sc, d[tq][2 * kb + 1].z * sc,
                                         d[tq][2 * kb + 1].w * sc};
                    f16x8 dh, dl;
                    ab_split8(v8, dh, dl);
#pragma unroll
                    for (int t = 0; t < 4; t++) AB_MFMA3(da[tq][t], wh[t], wl[t], dh, dl);
                }
            }
        }
        // dAO = dY Wo can be far from dY's magnitude (a checkpoint with a large output projection): a second power of two
        // per atom puts ITS largest entry in [0.25, 0.5) before it is split into fp16 planes and multiplied on
        float m2 = 0.f;
#pragma unroll
        for (int tq = 0; tq < NQ; tq++)
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int i = 0; i < 16; i++) m2 = fmaxf(m2, fabsf(da[tq][t][i]));
        float m2a = gb ? 0.f : m2, m2b = gb ? m2 : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            m2a = fmaxf(m2a, __shfl_xor(m2a, o));
            m2b = fmaxf(m2b, __shfl_xor(m2b, o));
        }
        m2 = (gb ? m2b : m2a) * ABQ_INV;  // largest |dAO| of this lane's atom (in the first scale)
        int e2 = ((__float_as_int(m2) >> 23) & 0xff) + 2;
        e2 = e2 > 253 ? 253 : e2;
        e2 = e2 < 16 ? 16 : e2;
        const float s2 = __int_as_float((254 - e2) << 23) * ABS_INV;  // accumulator (4096 x) -> planes' source (64 x), rescaled
        inv_sc *= __int_as_float(e2 << 23);
#pragma unroll
        for (int tq = 0; tq < NQ; tq++)
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<float4*>(tileB + tq * 16384 + ((4 * t + j) * 64 + L.lane) * 16) =
                        make_float4(da[tq][t][4 * j] * s2, da[tq][t][4 * j + 1] * s2, da[tq][t][4 * j + 2] * s2,
                                    da[tq][t][4 * j + 3] * s2);  // 64 dAO: what the planes are split from
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");

    AB_T(10);
    f32x16 dxn[NQ][4];
#pragma unroll
    for (int tq = 0; tq < NQ; tq++)
#pragma unroll
        for (int t = 0; t < 4; t++) dxn[tq][t] = ab_zero();
    float db[NQ];
#pragma unroll
    for (int tk = 0; tk < NQ; tk++) db[tk] = 0.f;
    constexpr float LN2 = 0.6931471805599453f;

#pragma unroll 1
    for (int hp = 0; hp < 4; hp++) {
        const int gbase = 4 + 7 * hp;
        // ---- Q^T, K^T, V^T of the head pair (token form), as in the forward
        f32x16 q[NQ], k[NQ], v[NQ];
#pragma unroll
        for (int tq = 0; tq < NQ; tq++) {
            ab_bias_tile(q[tq], bqkv + 32 * hp, L.h);
            ab_bias_tile(k[tq], bqkv + D + 32 * hp, L.h);
            ab_bias_tile(v[tq], bqkv + 2 * D + 32 * hp, L.h);
        }
#pragma unroll
        for (int sg = 0; sg < 4; sg++) {
            const int g = gbase + sg;
            AB_STAGE_SYNC();
            ab_bwd_request<NW>(g + 1, wqkv, wot, wqkvt, ring_u, wave, lane16);
            const char* slot = ring + (g & 1) * AB_SLOT_B + lane16;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int kb = 2 * sg + j;
                const f16x8 wqh = *reinterpret_cast<const f16x8*>(slot + (6 * j + 0) * 1024);
                const f16x8 wql = *reinterpret_cast<const f16x8*>(slot + (6 * j + 1) * 1024);
                const f16x8 wkh = *reinterpret_cast<const f16x8*>(slot + (6 * j + 2) * 1024);
                const f16x8 wkl = *reinterpret_cast<const f16x8*>(slot + (6 * j + 3) * 1024);
                const f16x8 wvh = *reinterpret_cast<const f16x8*>(slot + (6 * j + 4) * 1024);
                const f16x8 wvl = *reinterpret_cast<const f16x8*>(slot + (6 * j + 5) * 1024);
#pragma unroll
                for (int tq = 0; tq < NQ; tq++) {
                    const char* tp = tile + tq * 16384;
                    const f16x8 xh = *reinterpret_cast<const f16x8*>(tp + ((kb * 2 + 0) * 64 + L.lane) * 16);
                    const f16x8 xl = *reinterpret_cast<const f16x8*>(tp + ((kb * 2 + 1) * 64 + L.lane) * 16);
                    AB_MFMA3(q[tq], wqh, wql, xh, xl);
                    AB_MFMA3(k[tq], wkh, wkl, xh, xl);
                    AB_MFMA3(v[tq], wvh, wvl, xh, xl);
                }
            }
        }
        AB_T(11);
        // ---- operand planes: token form (index = head of the pair) and feature form (index = token K block)
        f16x8 qh[NQ][2], ql[NQ][2], kH[NQ][2], kL[NQ][2], vH[NQ][2], vL[NQ][2], dah[NQ][2], dal[NQ][2];
        f16x8 qfH[NQ][2], qfL[NQ][2], kfH[NQ][2], kfL[NQ][2], dfH[NQ][2], dfL[NQ][2];
#pragma unroll
        for (int tq = 0; tq < NQ; tq++) {
            ab_tile_planes(q[tq], qscale * ABS_INV, qh[tq], ql[tq]);
            ab_tile_planes(k[tq], ABS_INV, kH[tq], kL[tq]);
            ab_tile_planes(v[tq], ABS_INV, vH[tq], vL[tq]);
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const float4 d0 = *reinterpret_cast<const float4*>(tileB + tq * 16384 + ((4 * hp + 2 * b) * 64 + L.lane) * 16);
                const float4 d1 = *reinterpret_cast<const float4*>(tileB + tq * 16384 + ((4 * hp + 2 * b + 1) * 64 + L.lane) * 16);
                const float v8[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                ab_split8(v8, dah[tq][b], dal[tq][b]);
            }
            ab_transpose(qh[tq], ql[tq], sel1, qfH[tq], qfL[tq]);
            ab_transpose(kH[tq], kL[tq], sel1, kfH[tq], kfL[tq]);
            ab_transpose(dah[tq], dal[tq], sel1, dfH[tq], dfL[tq]);
        }
        AB_T(12);
        f32x16 dq[NQ], dk[NQ], dv[NQ];  // token-form tiles of the pair: registers 8 hd .. 8 hd + 7 from head hd
        f32x16 dkh[2][NQ], dvh[2][NQ];
#pragma unroll
        for (int hd = 0; hd < 2; hd++)
#pragma unroll
            for (int tk = 0; tk < NQ; tk++) { dkh[hd][tk] = ab_zero(); dvh[hd][tk] = ab_zero(); }
#pragma unroll
        for (int tq = 0; tq < NQ; tq++) {
            // the two heads of the pair side by side
            f32x16 s[2][NQ], dp[2][NQ];
#pragma unroll
            for (int hd = 0; hd < 2; hd++)
#pragma unroll
                for (int tk = 0; tk < NQ; tk++) {
                    s[hd][tk] = ab_zero();
                    dp[hd][tk] = ab_zero();
                    AB_MFMA3(s[hd][tk], kH[tk][hd], kL[tk][hd], qh[tq][hd], ql[tq][hd]);
                    AB_MFMA3(dp[hd][tk], vH[tk][hd], vL[tk][hd], dah[tq][hd], dal[tq][hd]);
                }
            float mx[2], sum[2], dl[2];
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                mx[hd] = -INFINITY;
#pragma unroll
                for (int tk = 0; tk < NQ; tk++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        s[hd][tk][i] = fmaf(s[hd][tk][i], ABQ_INV, bias[tk][i]);
                        mx[hd] = fmaxf(mx[hd], s[hd][tk][i]);
                    }
            }
#pragma unroll
            for (int hd = 0; hd < 2; hd++) mx[hd] = fmaxf(mx[hd], __shfl_xor(mx[hd], 32));
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                sum[hd] = 0.f;
                dl[hd] = 0.f;
#pragma unroll
                for (int tk = 0; tk < NQ; tk++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float p = __builtin_amdgcn_exp2f(s[hd][tk][i] - mx[hd]);
                        s[hd][tk][i] = p;
                        sum[hd] += p;
                        dl[hd] = fmaf(p, dp[hd][tk][i], dl[hd]);
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
#pragma unroll
                for (int tk = 0; tk < NQ; tk++) {
                    f32x16 ds;  // 64 dS^T
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float pn = s[hd][tk][i] * inv64;  // 64 P^T
                        s[hd][tk][i] = pn;
                        ds[i] = pn * fmaf(dp[hd][tk][i], ABQ_INV, -delta);
                    }
                    f16x8 pth[2], ptl[2], sth[2], stl[2];
                    ab_tile_planes(s[hd][tk], pth, ptl);
                    ab_tile_planes(ds, sth, stl);
                    // dQ^T += K^T dS^T (keys of tile tk)
#pragma unroll
                    for (int b = 0; b < 2; b++) AB_MFMA3(dqh[hd], kfH[tk][b], kfL[tk][b], sth[b], stl[b]);
                    // the (query, key) forms: P and dS with lane = key, registers = queries of tile tq
                    f16x8 ph[2], pl[2], sh[2], sl[2];
                    ab_transpose(pth, ptl, sel1, ph, pl);
                    db[tk] += ab_transpose_sum(sth, stl, sel1, sh, sl);
#pragma unroll
                    for (int b = 0; b < 2; b++) {
                        AB_MFMA3(dkh[hd][tk], qfH[tq][b], qfL[tq][b], sh[b], sl[b]);
                        AB_MFMA3(dvh[hd][tk], dfH[tq][b], dfL[tq][b], ph[b], pl[b]);
                    }
                }
            }
#pragma unroll
            for (int hd = 0; hd < 2; hd++)
#pragma unroll
                for (int j = 0; j < 8; j++) dq[tq][8 * hd + j] = dqh[hd][8 * hd + j];
        }
#pragma unroll
        for (int hd = 0; hd < 2; hd++)
#pragma unroll
            for (int tk = 0; tk < NQ; tk++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    dk[tk][8 * hd + j] = dkh[hd][tk][8 * hd + j];
                    dv[tk][8 * hd + j] = dvh[hd][tk][8 * hd + j];
                }
        AB_T(13);
        // ---- dXn^T += Wqkv^T [dQ; dK; dV]^T: K blocks 2 hp, 2 hp + 1 of each of the three parts
        f16x8 gh[NQ][3][2], gl[NQ][3][2];
#pragma unroll
        for (int tq = 0; tq < NQ; tq++) {
            ab_tile_planes(dq[tq], scale * ABS_INV, gh[tq][0], gl[tq][0]);
            ab_tile_planes(dk[tq], LN2 * ABS_INV, gh[tq][1], gl[tq][1]);
            ab_tile_planes(dv[tq], ABS_INV, gh[tq][2], gl[tq][2]);
        }
#pragma unroll
        for (int x = 0; x < 3; x++) {
            const int g = gbase + 4 + x;
            AB_STAGE_SYNC();
            if (g + 1 < 32) ab_bwd_request<NW>(g + 1, wqkv, wot, wqkvt, ring_u, wave, lane16);
            const char* slot = ring + (g & 1) * AB_SLOT_B + lane16;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int st = 2 * x + j, part = st >> 1, b = st & 1;
                f16x8 wh[4], wl[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    wh[t] = *reinterpret_cast<const f16x8*>(slot + (8 * j + 2 * t) * 1024);
                    wl[t] = *reinterpret_cast<const f16x8*>(slot + (8 * j + 2 * t + 1) * 1024);
                }
#pragma unroll
                for (int tq = 0; tq < NQ; tq++)
#pragma unroll
                    for (int t = 0; t < 4; t++) AB_MFMA3(dxn[tq][t], wh[t], wl[t], gh[tq][part][b], gl[tq][part][b]);
            }
        }
        AB_T(14);
    }
    // ---- key-bias gradient (summed over the heads; one writer per edge and layer)
    AB_T(15);
#pragma unroll
    for (int tk = 0; tk < NQ; tk++) {
        const float v = (db[tk] + __shfl_xor(db[tk], 32)) * (inv_sc * ABS_INV);  // the transposed planes held 64 dS
        const int key = 32 * tk + L.r;
        if (live && L.h == 0 && key < a.T && !a.centre(key)) dbias[a.edge(key)] = v;
    }
    // ---- norm adjoint, residual, whole-line stores
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float* stg = reinterpret_cast<float*>(tileB);  // staging for whole-line stores: the dAO rows are dead
#pragma unroll
    for (int tq = 0; tq < NQ; tq++) {
        if (32 * tq >= a.T) continue;
        float4 w[16];
        const float f = ABQ_INV * inv_sc;  // (W_qkv^T carries the norm's weight: dxn is the adjoint w.r.t. xhat)
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                w[4 * t + j] = make_float4(dxn[tq][t][4 * j] * f, dxn[tq][t][4 * j + 1] * f, dxn[tq][t][4 * j + 2] * f,
                                           dxn[tq][t][4 * j + 3] * f);
        const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
        float4 dr[2][8];  // the residual (dX1 rows) in the store's shape, requested before the norm adjoint's arithmetic
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int s = 32 * tq + 4 * j + rr;
                dr[c][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s < a.T && !a.centre(s)) dr[c][j] = *reinterpret_cast<const float4*>(dX1 + a.edge(s) * D + 64 * c + cc);
            }
        ab_norm_adjoint_planes<LN>(w, tile + tq * 16384, rstd[tq], L);
#pragma unroll
        for (int c = 0; c < 2; c++) {
#pragma unroll
            for (int kg = 0; kg < 8; kg++)
                *reinterpret_cast<float4*>(stg + L.r * TILE_LD + 8 * kg + 4 * L.h) = w[8 * c + kg];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = 4 * j + rr, s = 32 * tq + r;
                if (live && s < a.T) {
                    float4 o4 = *reinterpret_cast<const float4*>(stg + r * TILE_LD + cc);
                    o4.x += dr[c][j].x; o4.y += dr[c][j].y; o4.z += dr[c][j].z; o4.w += dr[c][j].w;
                    float* dst = dXin + (a.centre(s) ? E + a.atom(s) : a.edge(s)) * D;
                    *reinterpret_cast<float4*>(dst + 64 * c + cc) = o4;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        AB_T(16);
    }
}

}  // namespace pet
