// The adjoint of the per-atom fused attention block as TWO kernels, split at dQ | dK | dV (round 5).
//
// The one-kernel adjoint (pet_ablk.hip k_ablk_bwd) holds, per wave, the 64 accumulators of dXn next to everything the
// attention core needs: 446 registers and 32 KB of LDS per wave, i.e. ONE wave per SIMD, whose matrix and vector work
// cannot overlap (timing ablations of round 5, 8 x 10 000 atoms, ms per launch: 3.30 as built, 2.52 with every MFMA
// replaced by one multiply-add, 1.76 with, on top of that, no weight stream and no stage barriers -- the parts add up).
// Split:
//
//   k_ablk_bwd_core   TWO waves per 32-slot tile, one per half of the heads (head pairs 2 p, 2 p + 1), eight waves and four
//                     tiles per workgroup = two waves per SIMD. Per tile 32 KB of LDS shared by its two waves: the planes
//                     of the normalised rows (wave 0 makes them) and the planes of the incoming adjoint rows (wave 1), later
//                     dAO = dY Wo^T (each wave its own 64 columns). Per wave: dAO of its heads, then per head pair Q, K, V
//                     (recomputed), and per head S, P, dP, dS, dQ, dK, dV as in the one-kernel form -- with the feature
//                     forms made PER HEAD (the transposition against a selection matrix of one K block leaves the other
//                     head's lanes zero), so that both heads accumulate into the pair's three token-form tiles. Leaves
//                     dQ | dK | dV as fp16 planes (64 x, in the atom's power-of-two scale) in MFMA operand order, dense:
//                     tile k owns tokens tok0 .. tok0 + T - 1 (Graph::tile_desc), piece c = 2 (6 hp + 2 part + b) + plane
//                     at byte (tok0 * 96 + c * T * 2) * 16, inside it [half h][slot r] 16-B entries: 1 536 B per token,
//                     the size of the fp32 rows the three-kernel form writes there. Also: the scale per token, the
//                     key-bias gradient (the two waves' sums over their heads meet in LDS).
//   k_ablk_bwd_x      one wave per tile: dXn^T = Wqkv^T [dQ; dK; dV]^T with the planes as B operands straight from global
//                     memory (no split arithmetic, requests three stages ahead), norm adjoint, residual, whole-line stores.
//
// The accumulation order of dXn (head pair, part, K block) and every scale are those of k_ablk_bwd. Reference:
// pet/modules/transformer.py:86-152, 203-234 (autograd of the PreLN attention block).
#include "ablk.h"

namespace pet {

constexpr int AB2_MISC = 256;    // per workgroup (= tile): { first-scale exponents (2), second-scale maxima [half][atom] (4), key-bias partial sums (32) }

// The core kernel's weights do NOT go through a workgroup-shared LDS ring: with two waves per tile a ring stage is 6 .. 9 MFMAs
// per wave, far shorter than the ~1 us an LDS-DMA request takes to land, and the LDS left beside the tiles (31 KB) cannot hold
// enough stages in flight (first version of this kernel, 24 two-slot stages: 35 us per wave, 2.27 ms per launch). Each wave
// reads its own fragments from L2 into registers, three K blocks ahead -- the registers are free during the GEMM phases, the
// kernel's register peak is in the attention core -- and the two halves of a tile meet at three barriers only.
struct AbW6 {  // the six QKV fragments of one K block of one head pair: Qh Ql Kh Kl Vh Vl
    f16x8 f[6];
};
__device__ __forceinline__ void ab2_ld_qkv(AbW6& w, const W2& wqkv, int hp, int kb, int lane) {
#pragma unroll
    for (int part = 0; part < 3; part++) {
        const size_t i = (size_t)(32 * part + hp * 8 + kb) * 64 + lane;
        w.f[2 * part] = wqkv.h[i];
        w.f[2 * part + 1] = wqkv.l[i];
    }
}
struct AbW4 {  // Wo^T fragments of one K block, this half's two tiles: t0 h, t0 l, t1 h, t1 l
    f16x8 f[4];
};
__device__ __forceinline__ void ab2_ld_wot(AbW4& w, const W2& wot, int p, int kb, int lane) {
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const size_t i = (size_t)((2 * p + t) * 8 + kb) * 64 + lane;
        w.f[2 * t] = wot.h[i];
        w.f[2 * t + 1] = wot.l[i];
    }
}

// planes of ONE K block of a tile (index = head of the pair) -> planes of its transpose with the other head's lanes zero
__device__ __forceinline__ void ab_transpose_head(const f16x8& h, const f16x8& l, const f16x8& sel, f16x8 (&th)[2],
                                                  f16x8 (&tl)[2]) {
    f32x16 ch = ab_zero(), cl = ab_zero();
    ch = PET_MFMA_H(h, sel, ch);
    cl = PET_MFMA_H(l, sel, cl);
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            th[b][j] = (_Float16)ch[8 * b + j];
            tl[b][j] = (_Float16)cl[8 * b + j];
        }
}

template <bool LN>
__global__ __launch_bounds__(128, 2) void k_ablk_bwd_core(
    const float* __restrict__ X, const float* __restrict__ dX1, const float* __restrict__ dOC,
    const float* __restrict__ gamma, const float* __restrict__ beta, W2 wqkv, const float* __restrict__ bqkv, W2 wot,
    const float* __restrict__ fc, const int4* __restrict__ desc, int n_list, int64_t E, float qscale, float scale,
    f16x8* __restrict__ G, float* __restrict__ scl, float* __restrict__ dbias, int abl) {
    extern __shared__ __attribute__((aligned(16))) char ab_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one tile per workgroup, wave = half of the heads: four workgroups per CU (32.25 KB of LDS, <= 256 registers) = two waves
    // per SIMD that are NOT in step with each other -- eight-wave workgroups put the two waves of a SIMD behind the same
    // barriers, where they waited for memory at the same time
    const int tl = 0, p = wave;
    int li = blockIdx.x;
    const bool live = li < n_list;
    li = live ? li : n_list - 1;
    const AbAtom a(desc + 2 * (size_t)li, E);
    const int tok0 = __builtin_amdgcn_readfirstlane(desc[2 * (size_t)li + 1].z);
    char* tile = ab_smem + tl * 32768;   // planes of the normalised rows
    char* tileB = tile + 16384;          // planes of the incoming adjoint rows, then 64 dAO (fp32 row fragments)
    float* misc = reinterpret_cast<float*>(ab_smem + 32768);
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    if (abl & 4) {
    } else if (p == 0) {
        ab_dma_rows<1>(X, a, tile_u, L);
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int r = 2 * j + (L.lane >> 5);
            const int pc = (L.lane & 31) ^ (r & 15);
            const int s = r < a.T ? r : a.T - 1;
            const float* src = a.centre(s) ? dOC + (int64_t)a.atom(s) * D : dX1 + a.edge(s) * D;
            glds16_trr(src + 4 * pc, tile_u + 16384 + j * 1024);
        }
    }
    constexpr int PFW = 3;  // K blocks of weights in flight
    float bias[1][16];
    ab_key_bias<1>(bias, a, fc, L);
    const AbSel sel1 = ab_selectors(L, 1.0f);
    const bool gb = L.r >= a.TA;  // this lane's token belongs to the tile's second atom
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p == 0) {
        float4 x[16];
        tile128_to_frag(x, tile, L);
        norm_frag<16, LN>(x, gamma, beta, L.h);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        ab_park_planes(x, tile, L);
    } else {
        // one power of two per ATOM (largest entry of its scaled rows in [0.25, 0.5)); the planes hold 64 x that
        float4 d[16];
        tile128_to_frag(d, tileB, L);
        float m = 0.f;
        const bool lv = L.r < a.T;
#pragma unroll
        for (int kg = 0; kg < 16; kg++) {
            if (!lv) d[kg] = make_float4(0.f, 0.f, 0.f, 0.f);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(d[kg].x), fabsf(d[kg].y))), fmaxf(fabsf(d[kg].z), fabsf(d[kg].w)));
        }
        float ma = gb ? 0.f : m, mb = gb ? m : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            ma = fmaxf(ma, __shfl_xor(ma, o));
            mb = fmaxf(mb, __shfl_xor(mb, o));
        }
        int ea = ((__float_as_int(ma) >> 23) & 0xff) + 2, eb = ((__float_as_int(mb) >> 23) & 0xff) + 2;
        ea = ea > 253 ? 253 : (ea < 16 ? 16 : ea);
        eb = eb > 253 ? 253 : (eb < 16 ? 16 : eb);
        if (L.lane == 0) {
            misc[0] = __int_as_float(ea << 23);
            misc[1] = __int_as_float(eb << 23);
        }
        const float sc = __int_as_float((254 - (gb ? eb : ea)) << 23);
#pragma unroll
        for (int kg = 0; kg < 16; kg++) { d[kg].x *= sc; d[kg].y *= sc; d[kg].z *= sc; d[kg].w *= sc; }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        ab_park_planes(d, tileB, L);  // (times 64)
    }
    AbW4 wo4[PFW];
#pragma unroll
    for (int kb = 0; kb < PFW; kb++) ab2_ld_wot(wo4[kb], wot, p, kb, L.lane);
    __syncthreads();  // both row-plane tiles are complete

    // ---- dAO = dY Wo^T, this half's 64 columns (tiles 2 p, 2 p + 1)
    float inv_sc;
    {
        f32x16 da[2];
        da[0] = ab_zero();
        da[1] = ab_zero();
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const f16x8 dh = *reinterpret_cast<const f16x8*>(tileB + ((g * 2 + 0) * 64 + L.lane) * 16);
            const f16x8 dl = *reinterpret_cast<const f16x8*>(tileB + ((g * 2 + 1) * 64 + L.lane) * 16);
            const AbW4 wc = wo4[g % PFW];
            if (g + PFW < 8 && !(abl & 2)) ab2_ld_wot(wo4[g % PFW], wot, p, g + PFW, L.lane);
#pragma unroll
            for (int t = 0; t < 2; t++) AB_MFMA3(da[t], wc.f[2 * t], wc.f[2 * t + 1], dh, dl);
        }
        // second power of two per atom from the largest |dAO| over ALL 128 columns: the halves meet in LDS
        float m2 = 0.f;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int i = 0; i < 16; i++) m2 = fmaxf(m2, fabsf(da[t][i]));
        float m2a = gb ? 0.f : m2, m2b = gb ? m2 : 0.f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            m2a = fmaxf(m2a, __shfl_xor(m2a, o));
            m2b = fmaxf(m2b, __shfl_xor(m2b, o));
        }
        if (L.lane == 0) {
            misc[2 + 2 * p] = m2a;
            misc[3 + 2 * p] = m2b;
        }
        __syncthreads();  // every wave is done with the planes of the adjoint rows; the maxima and the first scale are visible
        m2 = (gb ? fmaxf(misc[3], misc[5]) : fmaxf(misc[2], misc[4])) * ABQ_INV;
        int e2 = ((__float_as_int(m2) >> 23) & 0xff) + 2;
        e2 = e2 > 253 ? 253 : e2;
        e2 = e2 < 16 ? 16 : e2;
        const float s2 = __int_as_float((254 - e2) << 23) * ABS_INV;
        inv_sc = misc[gb ? 1 : 0] * __int_as_float(e2 << 23);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                *reinterpret_cast<float4*>(tileB + ((4 * (2 * p + t) + j) * 64 + L.lane) * 16) =
                    make_float4(da[t][4 * j] * s2, da[t][4 * j + 1] * s2, da[t][4 * j + 2] * s2, da[t][4 * j + 3] * s2);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    if (p == 0 && live && L.h == 0 && L.r < a.T) scl[tok0 + L.r] = inv_sc;

    float db = 0.f;
    constexpr float LN2 = 0.6931471805599453f;
    const char* gbase = reinterpret_cast<const char*>(G) + ((size_t)tok0 * 96 + (size_t)(L.h * a.T + L.r)) * 16;
    const size_t pstride = (size_t)a.T * 32;  // bytes of one piece

    // The planes of a round are stored AFTER the next round's Q, K, V products: vmcnt retires in order, loads and stores alike,
    // so weight fragments requested behind the stores would wait for the stores' acknowledgements (0.35 ms per launch).
    f16x8 oh[3][2], ol[3][2];  // [part][K block] planes of dQ | dK | dV of the round before
#pragma unroll
    for (int x = 0; x < 3; x++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int j = 0; j < 8; j++) { oh[x][b][j] = (_Float16)0.f; ol[x][b][j] = (_Float16)0.f; }
    auto store_planes = [&](int hp_) {
        if (live && L.r < a.T && !(abl & 1)) {
            char* dst = const_cast<char*>(gbase) + (size_t)(12 * hp_) * pstride;
#pragma unroll
            for (int x = 0; x < 3; x++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    *reinterpret_cast<f16x8*>(dst + (size_t)(4 * x + 2 * b) * pstride) = oh[x][b];
                    *reinterpret_cast<f16x8*>(dst + (size_t)(4 * x + 2 * b + 1) * pstride) = ol[x][b];
                }
        }
    };
#pragma unroll 1
    for (int r = 0; r < 2; r++) {
        const int hp = 2 * p + r;
        // ---- Q^T, K^T, V^T of the head pair (token form)
        f32x16 q, k, v;
        ab_bias_tile(q, bqkv + 32 * hp, L.h);
        ab_bias_tile(k, bqkv + D + 32 * hp, L.h);
        ab_bias_tile(v, bqkv + 2 * D + 32 * hp, L.h);
        {
            AbW6 w6[PFW];
#pragma unroll
            for (int kb = 0; kb < PFW; kb++) ab2_ld_qkv(w6[kb], wqkv, hp, kb, L.lane);
#pragma unroll
            for (int kb = 0; kb < 8; kb++) {
                const f16x8 xh = *reinterpret_cast<const f16x8*>(tile + ((kb * 2 + 0) * 64 + L.lane) * 16);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(tile + ((kb * 2 + 1) * 64 + L.lane) * 16);
                const AbW6 wc = w6[kb % PFW];
                if (kb + PFW < 8 && !(abl & 2)) ab2_ld_qkv(w6[kb % PFW], wqkv, hp, kb + PFW, L.lane);
                AB_MFMA3(q, wc.f[0], wc.f[1], xh, xl);
                AB_MFMA3(k, wc.f[2], wc.f[3], xh, xl);
                AB_MFMA3(v, wc.f[4], wc.f[5], xh, xl);
            }
        }
        if (r > 0) store_planes(hp - 1);
        // ---- token-form planes of the pair (index = head)
        f16x8 qh[2], ql[2], kH[2], kL[2], vH[2], vL[2], dah[2], dal[2];
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
        f32x16 dq = ab_zero(), dk = ab_zero(), dv = ab_zero();  // token-form tiles of the pair; a head fills its own 8 registers
#pragma unroll
        for (int hd = 0; hd < 2; hd++) {
            if (abl & 8) break;
            const f16x8 selh = hd ? sel1.i1 : sel1.i0;
            f32x16 s = ab_zero(), dp = ab_zero();
            AB_MFMA3(s, kH[hd], kL[hd], qh[hd], ql[hd]);
            AB_MFMA3(dp, vH[hd], vL[hd], dah[hd], dal[hd]);
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                s[i] = fmaf(s[i], ABQ_INV, bias[0][i]);
                mx = fmaxf(mx, s[i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f, dl = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float pe = __builtin_amdgcn_exp2f(s[i] - mx);
                s[i] = pe;
                sum += pe;
                dl = fmaf(pe, dp[i], dl);
            }
            sum += __shfl_xor(sum, 32);
            dl += __shfl_xor(dl, 32);
            const float inv = __builtin_amdgcn_rcpf(sum);
            const float delta = dl * inv * ABQ_INV;
            const float inv64 = inv * ABS;
            f32x16 ds;  // 64 dS^T
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float pn = s[i] * inv64;  // 64 P^T
                s[i] = pn;
                ds[i] = pn * fmaf(dp[i], ABQ_INV, -delta);
            }
            f16x8 pth[2], ptl[2], sth[2], stl[2];
            ab_tile_planes(s, pth, ptl);
            ab_tile_planes(ds, sth, stl);
            {   // dQ^T += K^T dS^T: K of this head in feature form, the other head's rows zero
                f16x8 kfH[2], kfL[2];
                ab_transpose_head(kH[hd], kL[hd], selh, kfH, kfL);
#pragma unroll
                for (int b = 0; b < 2; b++) AB_MFMA3(dq, kfH[b], kfL[b], sth[b], stl[b]);
            }
            // the (query, key) forms: P and dS with lane = key, registers = queries
            f16x8 ph[2], pl[2], sh[2], sl[2];
            ab_transpose(pth, ptl, sel1, ph, pl);
            db += ab_transpose_sum(sth, stl, sel1, sh, sl);
            {
                f16x8 qfH[2], qfL[2];
                ab_transpose_head(qh[hd], ql[hd], selh, qfH, qfL);
#pragma unroll
                for (int b = 0; b < 2; b++) AB_MFMA3(dk, qfH[b], qfL[b], sh[b], sl[b]);
            }
            {
                f16x8 dfH[2], dfL[2];
                ab_transpose_head(dah[hd], dal[hd], selh, dfH, dfL);
#pragma unroll
                for (int b = 0; b < 2; b++) AB_MFMA3(dv, dfH[b], dfL[b], ph[b], pl[b]);
            }
        }
        // ---- dQ | dK | dV of the pair as planes: piece 2 (6 hp + 2 part + b) + plane
        ab_tile_planes(dq, scale * ABS_INV, oh[0], ol[0]);
        ab_tile_planes(dk, LN2 * ABS_INV, oh[1], ol[1]);
        ab_tile_planes(dv, ABS_INV, oh[2], ol[2]);
    }
    store_planes(2 * p + 1);
    // ---- key-bias gradient: this half's four heads; half 1 hands its sums over in LDS, half 0 writes (one writer per edge)
    {
        const float vsum = (db + __shfl_xor(db, 32)) * (inv_sc * ABS_INV);  // the transposed planes held 64 dS
        if (p == 1 && L.h == 0) misc[8 + L.r] = vsum;
        __syncthreads();
        if (p == 0 && live && L.h == 0 && L.r < a.T && !a.centre(L.r)) dbias[a.edge(L.r)] = vsum + misc[8 + L.r];
    }
}

// ---------------------------------------------------------------------------------------------
// k_ablk_bwd_x: dXn^T += Wqkv^T [dQ; dK; dV]^T from the planes, norm adjoint, residual
// stage g = 0 .. 11: head pair g / 3, part g % 3, K blocks 0, 1: the 16 fragments t * 24 + 8 part + 2 hp + b (t = 0 .. 3) x (H, L)
// ---------------------------------------------------------------------------------------------
constexpr int AB2_SLOT_X = 16384;
__device__ __forceinline__ void ab2x_request(int g, const W2& wqkvt, unsigned ring_u, int wave, unsigned lane16) {
#ifdef AB_ABL_NODMA
    if (g > 0) return;
#endif
    const unsigned dst = ring_u + (unsigned)(g & 1) * AB2_SLOT_X;
    const int hp = g / 3, part = g % 3;
#pragma unroll
    for (int p0 = 0; p0 < 16; p0 += 8) {
        const int pc = p0 + wave;
        const int b = pc >> 3, t = (pc >> 1) & 3, pl = pc & 1;
        ab_dma_piece(pl ? wqkvt.l : wqkvt.h, t * 24 + 8 * part + 2 * hp + b, lane16, dst + pc * 1024);
    }
}

// stage boundary of k_ablk_bwd_x: this wave's fragments of the stage have landed (vmcnt retires in order: everything but the
// N requests issued after them -- the plane loads of a later stage -- is complete), then the workgroup barrier
template <int N>
__device__ __forceinline__ void ab2x_sync() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#ifndef AB_ABL_NOBAR
    __syncthreads();
#endif
}

template <bool LN>
__global__ __launch_bounds__(512) void k_ablk_bwd_x(
    const float* __restrict__ X, const float* __restrict__ dX1, const f16x8* __restrict__ G, const float* __restrict__ scl,
    const float* __restrict__ gamma, W2 wqkvt, const int4* __restrict__ desc, int n_list, int64_t E,
    float* __restrict__ dXin) {
    extern __shared__ __attribute__((aligned(16))) char ab_smem[];
    constexpr int NW = 8, PF = 3;
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int li = blockIdx.x * NW + wave;
    const bool live = li < n_list;
    li = live ? li : n_list - 1;
    const AbAtom a(desc + 2 * (size_t)li, E);
    const int tok0 = __builtin_amdgcn_readfirstlane(desc[2 * (size_t)li + 1].z);
    char* tile = ab_smem + wave * 16384;  // the layer input rows (for the norm adjoint), then the staging tile of the stores
    const char* ring = ab_smem + NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    const int rc = L.r < a.T ? L.r : a.T - 1;  // slots past the last token repeat it (never stored)
    const char* gbase = reinterpret_cast<const char*>(G) + ((size_t)tok0 * 96 + (size_t)(L.h * a.T + rc)) * 16;
    const size_t pstride = (size_t)a.T * 32;
    const float inv_sc = scl[tok0 + rc];
    ab_dma_rows<1>(X, a, tile_u, L);
    ab2x_request(0, wqkvt, ring_u, wave, lane16);
    f16x8 bq[PF][4];  // [stage in flight][2 b + plane]
    auto ldB = [&](int g, f16x8 (&o)[4]) {
#pragma unroll
        for (int c = 0; c < 4; c++) o[c] = *reinterpret_cast<const f16x8*>(gbase + (size_t)(4 * g + c) * pstride);
    };
#pragma unroll
    for (int g = 0; g < PF; g++) ldB(g, bq[g]);
    f32x16 dxn[4];
#pragma unroll
    for (int t = 0; t < 4; t++) dxn[t] = ab_zero();
#pragma unroll
    for (int g = 0; g < 12; g++) {
        // behind this stage's fragments in the request order: the plane loads of stage g - 1 + PF (g = 0: of all PF stages)
        if (g == 0) ab2x_sync<4 * PF>();
        else if (g - 1 + PF < 12) ab2x_sync<4>();
        else ab2x_sync<0>();
        if (g + 1 < 12) ab2x_request(g + 1, wqkvt, ring_u, wave, lane16);
        const char* slot = ring + (g & 1) * AB2_SLOT_X + lane16;
        f16x8 cur[4];
#pragma unroll
        for (int c = 0; c < 4; c++) cur[c] = bq[g % PF][c];
        if (g + PF < 12) ldB(g + PF, bq[g % PF]);
#pragma unroll
        for (int b = 0; b < 2; b++) {
            f16x8 wh[4], wl[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                wh[t] = *reinterpret_cast<const f16x8*>(slot + (8 * b + 2 * t) * 1024);
                wl[t] = *reinterpret_cast<const f16x8*>(slot + (8 * b + 2 * t + 1) * 1024);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) AB_MFMA3(dxn[t], wh[t], wl[t], cur[2 * b], cur[2 * b + 1]);
        }
    }
    // ---- norm adjoint, residual, whole-line stores (k_ablk_bwd's epilogue)
    __syncthreads();  // (nobody reads the ring any more; the tiles are wave-private)
    float4 w[16], x[16];
    tile128_to_frag(x, tile, L);
    const float f = ABQ_INV * inv_sc;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4 g4 = *reinterpret_cast<const float4*>(gamma + 32 * t + 8 * j + 4 * L.h);
            w[4 * t + j] = make_float4(dxn[t][4 * j] * f * g4.x, dxn[t][4 * j + 1] * f * g4.y,
                                       dxn[t][4 * j + 2] * f * g4.z, dxn[t][4 * j + 3] * f * g4.w);
        }
    const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
    auto residual = [&](int c, float4 (&dr)[8]) {  // the dX1 rows in the store's shape
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int s = 4 * j + rr;
            dr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s < a.T && !a.centre(s)) dr[j] = *reinterpret_cast<const float4*>(dX1 + a.edge(s) * D + 64 * c + cc);
        }
    };
    float4 dr0[8], dr1[8];
    residual(0, dr0);  // requested before the norm adjoint's arithmetic
    norm_bwd_frag<16, LN>(w, x);
    residual(1, dr1);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float* stg = reinterpret_cast<float*>(tile);
#pragma unroll
    for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int kg = 0; kg < 8; kg++) *reinterpret_cast<float4*>(stg + L.r * TILE_LD + 8 * kg + 4 * L.h) = w[8 * c + kg];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = 4 * j + rr;
            if (live && r < a.T) {
                float4 o4 = *reinterpret_cast<const float4*>(stg + r * TILE_LD + cc);
                const float4 d4 = c ? dr1[j] : dr0[j];
                o4.x += d4.x; o4.y += d4.y; o4.z += d4.z; o4.w += d4.w;
                float* dst = dXin + (a.centre(r) ? E + a.atom(r) : a.edge(r)) * D;
                *reinterpret_cast<float4*>(dst + 64 * c + cc) = o4;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int g_ablk_abl = 0;    // debugging: timing ablations of the core kernel (bits: 1 no stores, 2 stale weights, 4 no row loads, 8 no attention core)
void set_ablk_abl(int v) { g_ablk_abl = v; }
static int g_ablk_split = 1;  // pet_config_set("attn_bwd_split", 0): the one-kernel adjoint for the 32-slot tiles too
void set_ablk_split(int v) { g_ablk_split = v ? 1 : 0; }
int ablk_split() { return g_ablk_split; }

// the 32-slot tiles of the graph (list, n1 of them): planes -> G (R x 384 floats), scales -> scl (R floats)
void ablk_bwd_split_launch(bool ln, const float* X, const float* dX1, const float* dOC, const float* g_attn, const float* b_attn,
                           W2 wq, const float* bq, W2 wot, W2 wqt, const float* fc, const int4* list, int n1, int64_t E,
                           float qscale, float scale, float* G, float* scl, float* dXin, float* dbias, hipStream_t st) {
    const size_t lds_a = 32768 + AB2_MISC, lds_b = 8 * 16384 + 2 * AB2_SLOT_X;
    f16x8* Gp = reinterpret_cast<f16x8*>(G);
    if (ln) {
        allow_big_lds(k_ablk_bwd_core<true>, lds_a);
        allow_big_lds(k_ablk_bwd_x<true>, lds_b);
        k_ablk_bwd_core<true><<<n1, 128, lds_a, st>>>(X, dX1, dOC, g_attn, b_attn, wq, bq, wot, fc, list, n1, E, qscale,
                                                               scale, Gp, scl, dbias, g_ablk_abl);
        k_ablk_bwd_x<true><<<cdiv(n1, 8), 512, lds_b, st>>>(X, dX1, Gp, scl, g_attn, wqt, list, n1, E, dXin);
    } else {
        allow_big_lds(k_ablk_bwd_core<false>, lds_a);
        allow_big_lds(k_ablk_bwd_x<false>, lds_b);
        k_ablk_bwd_core<false><<<n1, 128, lds_a, st>>>(X, dX1, dOC, g_attn, nullptr, wq, bq, wot, fc, list, n1, E,
                                                                qscale, scale, Gp, scl, dbias, g_ablk_abl);
        k_ablk_bwd_x<false><<<cdiv(n1, 8), 512, lds_b, st>>>(X, dX1, Gp, scl, g_attn, wqt, list, n1, E, dXin);
    }
}

}  // namespace pet
