// Message-passing combination stage (backend.py:559-575) and its adjoint as TRR kernels on the 16-bit matrix cores
// (f16x3, trr.h): one wave = 32 edges, one wave per SIMD, weight fragments through rings that run across the
// hidden chunks.
//   forward:  cat = [e ; e[rev]] (256) -> LayerNorm -> W0 (256 -> 256) -> SiLU -> W2 (256 -> 128);  M' = M + e + out
//   adjoint:  dM -> dcat [E, 256] (the ji scatter of its second half is k_dxf's gather)
// Same arithmetic as k_comb / k_comb_bwd (pet_fwd.hip / pet_bwd.hip), which stay selectable (trr = 0).
#include <type_traits>
#include "common.h"
#include "model.h"
#include "trr.h"

namespace pet {

#define TRR_PROLOGUE(NROWS)                                   \
    const RowLane L;                                          \
    const int64_t row0 = wave_row0();                         \
    if (row0 >= (NROWS)) return;                              \
    const bool valid = row0 + L.r < (NROWS);                  \
    const int64_t row = valid ? row0 + L.r : (NROWS) - 1

// ---------------------------------------------------------------------------------
// k_comb_p2: the same stage as a software-pipelined kernel (default since round 3; pet_config_set("comb_pipe", 0) restores
// k_comb_h), built like k_emlp_p2 (pet_trr.hip): the three stages of a hidden chunk -- a = W0 LayerNorm([e ; e[rev]]) (one
// 32-unit tile, K = 256: 48 MFMAs), SiLU + operand split (VALU), out += W2 silu(a) (24 MFMAs) -- depend on each other and
// a wave issues in order, so iteration hc is 24 slots of one f16x3 MFMA triple followed by VALU work that does not depend
// on it:
//   slots  0..7   out GEMM of chunk hc - 1 (2 K blocks x 4 tiles)  | slot s: pre-activations 2s, 2s + 1 of chunk hc folded
//   slots  8..23  a GEMM of chunk hc + 1 (16 K blocks, one tile)   | slot a: silu of element a; odd a: split of a pair;
//                                                                   |         last slot: the saved pre-activations leave
// e[p] and e[rev[p]] arrive by LDS-DMA as whole rows (the second through the row index rev[p] of each row); the split
// e[p] half stays in registers, the split e[rev] half is parked over its tile; M + e + b2 is the initial value of the out
// accumulators, so nothing is loaded after the loop.
// LDS per wave: [e tile, then u operand 4 KB | staging tile 4.5 KB | bias 1 KB] 16 KB, [e[rev] tile / split planes] 16 KB.
// ---------------------------------------------------------------------------------
template <bool FIRST>
__global__ __launch_bounds__(256) void k_comb_p2(const float* __restrict__ XF, const int* __restrict__ rev,
                                                  const float* __restrict__ ln_g, const float* __restrict__ ln_b, W2 w0,
                                                  const float* __restrict__ b0, W2 w2, const float* __restrict__ b2,
                                                  const float* __restrict__ Min, const float* __restrict__ edge_emb,
                                                  const int* __restrict__ sp_nbr, float* __restrict__ CA,
                                                  float* __restrict__ LNS, float* __restrict__ Mout, int64_t E) {
    extern __shared__ __attribute__((aligned(16))) char cp_lds[];
    TRR_PROLOGUE(E);
    constexpr int NC = 2 * D / 32;  // hidden chunks of 32
    char* const my = cp_lds + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 32768;
    f16x8* const usp = reinterpret_cast<f16x8*>(my);                  // [2 K blocks x (h, l)][64] (over the consumed e tile)
    float* const otile = reinterpret_cast<float*>(my + 4096);         // [32][TILE32_LD]
    float* const bias = reinterpret_cast<float*>(my + 4096 + 32 * TILE32_LD * 4);  // b0 [256]
    char* const rt = my + 16384;
    f16x8* const xrp = reinterpret_cast<f16x8*>(rt);                  // after the split: [8 K blocks x (h, l)][64]
    // the two row tiles
    dma_tile128(XF, row0, E, __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)my), L);
    {
        const int64_t rl = row0 + (L.lane & 31);
        const int rv = rev[rl < E ? rl : E - 1];  // row r of the tile takes XF[rev[row0 + r]]
        const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)rt);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int r = 2 * j + (L.lane >> 5);
            const int64_t rr = __shfl(rv, r);
            glds16_trr(XF + rr * 128 + 4 * ((L.lane & 31) ^ (r & 15)), base + j * 1024);
        }
    }
    float4 mi[16];
    if (FIRST) load_rowfrag<16>(mi, edge_emb, (int64_t)sp_nbr[row], D, L.h);
    else load_rowfrag<16>(mi, Min, row, D, L.h);
    const float4 b0v = reinterpret_cast<const float4*>(b0)[L.lane];
#ifdef C_ABL_W0  // timing ablation (results are wrong): every weight block is one of two: the stream comes from the CU's L1
    auto aidx = [&](int b) { return (size_t)(b & 1) * 64 + L.lane; };
#else
    auto aidx = [&](int b) { return (size_t)b * 64 + L.lane; };
#endif  // W0 tile hc, K block kb (of 16) at b = 16 hc + kb
    constexpr int RD = 4;  // W0 ring: K blocks kb .. kb + 3 (eight spill into the store regions of the loop)
    WBlk2<1> ra[RD];
#pragma unroll
    for (int b = 0; b < RD; b++) ld_blk2<1>(ra[b], w0, aidx(b), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA is not visible to the compiler's own bookkeeping
    f32x16 out[4], outl[4];
    Split2<8> xs;  // the e[p] half (K blocks 0..7) in registers; the e[rev[p]] half (blocks 8..15) parked in LDS
    {
        float4 xo[16], xr[16];
        tile128_to_frag(xo, my, L);
        tile128_to_frag(xr, rt, L);
        acc_bias<4>(out, b2, 0, L.h);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {  // M + e ride in the accumulator
                const float4 e = xo[4 * t + q], m4 = mi[4 * t + q];
                out[t][4 * q] += e.x + m4.x; out[t][4 * q + 1] += e.y + m4.y;
                out[t][4 * q + 2] += e.z + m4.z; out[t][4 * q + 3] += e.w + m4.w;
            }
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++)
            s1 += xo[k].x + xo[k].y + xo[k].z + xo[k].w + xr[k].x + xr[k].y + xr[k].z + xr[k].w;
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
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float4 ga = *reinterpret_cast<const float4*>(ln_g + 8 * k + 4 * L.h);
            const float4 ba = *reinterpret_cast<const float4*>(ln_b + 8 * k + 4 * L.h);
            const float4 gb = *reinterpret_cast<const float4*>(ln_g + D + 8 * k + 4 * L.h);
            const float4 bb = *reinterpret_cast<const float4*>(ln_b + D + 8 * k + 4 * L.h);
            xo[k].x = (xo[k].x - mean) * rstd * ga.x + ba.x; xo[k].y = (xo[k].y - mean) * rstd * ga.y + ba.y;
            xo[k].z = (xo[k].z - mean) * rstd * ga.z + ba.z; xo[k].w = (xo[k].w - mean) * rstd * ga.w + ba.w;
            xr[k].x = (xr[k].x - mean) * rstd * gb.x + bb.x; xr[k].y = (xr[k].y - mean) * rstd * gb.y + bb.y;
            xr[k].z = (xr[k].z - mean) * rstd * gb.z + bb.z; xr[k].w = (xr[k].w - mean) * rstd * gb.w + bb.w;
        }
        split_frag2<8>(xo, xs);
        Split2<8> t;
        split_frag2<8>(xr, t);
#pragma unroll
        for (int k = 0; k < 8; k++) { xrp[(2 * k) * 64 + L.lane] = t.h[k]; xrp[(2 * k + 1) * 64 + L.lane] = t.l[k]; }
        reinterpret_cast<float4*>(bias)[L.lane] = b0v;  // (the e tile under it is consumed)
    }
    acc_zero<4>(outl);
    __builtin_amdgcn_sched_barrier(0);
    // W2 fragments of one chunk, one (K block, tile) pair per slot of the out GEMM: ring entry s = 4 kb2 + tile
    f16x8 roh[8], rol[8];
    auto ld_w2 = [&](int c, int s) {
#ifdef C_ABL_W0
        const size_t i = (size_t)(s & 1) * 64 + L.lane;
#else
        const size_t i = ((size_t)(s & 3) * 16 + 2 * c + (s >> 2)) * 64 + L.lane;
#endif
        roh[s] = w2.h[i];
        rol[s] = w2.l[i];
    };
    f32x16 a1, a1l;  // pre-activations of the chunk the element slices work on; rebuilt for the next chunk in slots 8..23
    f16x8 ph, pl;    // K block 8.. of the parked e[rev] planes, requested one slot ahead
    f16x8 uh, ul;    // K block of the split u operand
    auto rd_p = [&](int kb) {  // kb = 8 .. 15
        unsigned o = L.lane;  // opaque offset: read here, not hoisted into registers
        asm volatile("" : "+v"(o));
        ph = xrp[o + (2 * (kb - 8)) * 64];
        pl = xrp[o + (2 * (kb - 8) + 1) * 64];
    };
    auto rd_u = [&](int kb2) {
        unsigned o = L.lane;
        asm volatile("" : "+v"(o));
        uh = usp[o + (2 * kb2) * 64];
        ul = usp[o + (2 * kb2 + 1) * 64];
    };
    // slot kb (0..15) of the a GEMM of chunk c
    auto a_slot = [&](int c, int kb) {
        WBlk2<1>& wb = ra[kb % RD];
        const f16x8 xh = kb < 8 ? xs.h[kb & 7] : ph, xl = kb < 8 ? xs.l[kb & 7] : pl;
        if (kb == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; r++) z[r] = 0.f;
            a1l = PET_MFMA_H(wb.l[0], xh, z);
            a1 = PET_MFMA_H(wb.h[0], xh, z);
        } else {
            a1l = PET_MFMA_H(wb.l[0], xh, a1l);
            a1 = PET_MFMA_H(wb.h[0], xh, a1);
        }
        a1l = PET_MFMA_H(wb.h[0], xl, a1l);
        if (kb >= 7 && kb < 15) rd_p(kb + 1);
        int nb = 16 * c + kb + RD;
        nb = nb < 16 * NC ? nb : 16 * NC - 1;  // past the end: the last block again (no branch in the loop body)
        ld_blk2<1>(wb, w0, aidx(nb), 0);
    };
    // slot s (0..7) of the out GEMM: K block s >> 2 of the u operand, output tile s & 3
    auto out_slot = [&](int s) {
        const int t = s & 3;
        outl[t] = PET_MFMA_H(rol[s], uh, outl[t]);
        out[t] = PET_MFMA_H(roh[s], uh, out[t]);
        outl[t] = PET_MFMA_H(roh[s], ul, outl[t]);
        if (s == 3) rd_u(1);
    };
    // MODE 1: first iteration (no out GEMM of a previous chunk), 2: last (no a GEMM of a next chunk)
    auto iteration = [&](auto mode, int hc) {
        constexpr int MODE = decltype(mode)::value;
        float aa[16];  // pre-activations of the chunk: element e = 4 q + c is hidden unit 8 q + 4 h + c
        float4 bq;
#pragma unroll
        for (int sl = 0; sl < 8; sl++) {
            if (MODE != 1) out_slot(sl);
#pragma unroll
            for (int e = 2 * sl; e < 2 * sl + 2; e++) {
                if ((e & 3) == 0) bq = *reinterpret_cast<const float4*>(bias + 32 * hc + 8 * (e >> 2) + 4 * L.h);
                aa[e] = a1[e] + a1l[e] * (1.0f / 2048.0f) + f4c(bq, e & 3);
                asm volatile("" : "+v"(aa[e]));  // computed in this slot
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        h16x2 sh[4], sl4[4];  // the fragment pair under construction
        float u0 = 0.f;
#pragma unroll
        for (int a = 0; a < 16; a++) {
            if (MODE != 2) a_slot(hc + 1, a);
            if (a < 8) ld_w2(hc, a);  // this chunk's fragments: its out GEMM runs in slots 0..7 of the next iteration
            const float u = silu_(aa[a]);
            if (a & 1) {
                h16x2 hp, lp;
                split_pair_pinned(u0, u, hp, lp);
                sh[(a & 7) >> 1] = hp;
                sl4[(a & 7) >> 1] = lp;
            } else {
                u0 = u;
                asm volatile("" : "+v"(u0));
            }
            if ((a & 7) == 7) {
                f16x8 fh, fl;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    fh[2 * w] = sh[w][0]; fh[2 * w + 1] = sh[w][1];
                    fl[2 * w] = sl4[w][0]; fl[2 * w + 1] = sl4[w][1];
                }
                usp[(2 * (a >> 3)) * 64 + L.lane] = fh;
                usp[(2 * (a >> 3) + 1) * 64 + L.lane] = fl;
            }
            if (CA && a == 11) {  // the saved pre-activations: whole 128-B lines through the staging tile
                float4 t4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) t4[q] = make_float4(aa[4 * q], aa[4 * q + 1], aa[4 * q + 2], aa[4 * q + 3]);
                store_tile32_lines(t4, otile, CA + 32 * hc, row0, E, 2 * D, L);
            }
            if (a == 15) rd_u(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int a = 0; a < 16; a++) {  // a of chunk 0
        a_slot(0, a);
        __builtin_amdgcn_sched_barrier(0);
    }
    iteration(std::integral_constant<int, 1>{}, 0);
#pragma unroll 1
    for (int hc = 1; hc + 1 < NC; hc++) iteration(std::integral_constant<int, 0>{}, hc);
    iteration(std::integral_constant<int, 2>{}, NC - 1);
#pragma unroll
    for (int sl = 0; sl < 8; sl++) out_slot(sl);
    fold_low<4>(out, outl);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float4 t4[4] = {acc_q(out[t], 0), acc_q(out[t], 1), acc_q(out[t], 2), acc_q(out[t], 3)};
        store_tile32_lines(t4, otile, Mout + 32 * t, row0, E, D, L);
    }
}

static inline W2 w2_of(const void* base, int n_tiles_dim, int k_dim) {
    const size_t n8 = (size_t)(n_tiles_dim / 32) * (k_dim / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}

bool trr_comb(bool first, const float* XF, const Graph& g, const GnnLayerW& G, const float* Min,
              const float* edge_emb, float* CA, float* LNS, float* Mout, int64_t E, hipStream_t st) {
    if (!G.comb0.fwd2 || !G.comb2.fwd2 || E <= 0) return false;
    if (comb_s(first, XF, g.rev, G.comb0_g, G.comb2, Min, edge_emb, g.sp_nbr, CA, LNS, Mout, E, st)) return true;  // large graphs (pet_comb_s.hip)
    const W2 w0 = w2_of(G.comb0.fwd2, G.comb0.n_out, G.comb0.k_in), w2 = w2_of(G.comb2.fwd2, G.comb2.n_out, G.comb2.k_in);
    const int grid = cdiv(E, WG_ROWS);
    const size_t lds = (size_t)4 * 32768;  // per wave: e tile (then operands / staging / bias), e[rev] tile (then its planes)
    if (first) {
        allow_big_lds(k_comb_p2<true>, lds);
        k_comb_p2<true><<<grid, 256, lds, st>>>(XF, g.rev, G.ln_g, G.ln_b, w0, G.comb0.b, w2, G.comb2.b, nullptr, edge_emb,
                                                g.sp_nbr, CA, LNS, Mout, E);
    } else {
        allow_big_lds(k_comb_p2<false>, lds);
        k_comb_p2<false><<<grid, 256, lds, st>>>(XF, g.rev, G.ln_g, G.ln_b, w0, G.comb0.b, w2, G.comb2.b, Min, edge_emb,
                                                 g.sp_nbr, CA, LNS, Mout, E);
    }
    return true;
}

}  // namespace pet
