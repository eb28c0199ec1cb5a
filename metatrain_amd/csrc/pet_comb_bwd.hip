// Message-passing combination stage (backend.py:559-575) and its adjoint as TRR kernels on the 16-bit matrix cores
// (f16x3, trr.h): one wave = 32 edges, one wave per SIMD, weight fragments through rings that run across the
// hidden chunks.
//   forward:  cat = [e ; e[rev]] (256) -> LayerNorm -> W0 (256 -> 256) -> SiLU -> W2 (256 -> 128);  M' = M + e + out
//   adjoint:  dM -> dcat [E, 256] (the ji scatter of its second half is k_dxf's gather)
// Same arithmetic as k_comb / k_comb_bwd (pet_fwd.hip / pet_bwd.hip), which stay selectable (trr = 0).
// This file: the ADJOINT, a translation unit of its own because it is compiled with the matrix products in VGPR form
// (build.py VGPR_FORM: k_comb_bwd_p2 1.89 -> 1.81 ms per launch with it, k_comb_p2 1.67 -> 1.76 -- hence not the forward).
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
// k_comb_bwd_p2: the adjoint as a software-pipelined kernel (pet_config_set("comb_bwd_pipe", 0) restores k_comb_bwd_h),
// built like k_emlp_bwd_p2 (pet_trr.hip). Phase 1, iteration hc = 16 slots of one f16x3 MFMA triple + a VALU slice:
//   slots  0..7   first half of dln += da(hc - 1) W0 (2 K blocks x 4 tiles) | slot s: elements 2s, 2s + 1 of da(hc) = t1 . silu'(CA)
//   slots  8..15  t1(hc + 1) = (dM W2)[chunk hc + 1], 8 K blocks             | slot k: one pair of da(hc) split; its planes parked
// The da chunks are parked SPLIT (fp16 planes, 4 KB per chunk) -- they are the operand of the slots 0..7 above and of
// phase 2 (second half of dln: 64 slots of MFMAs over the parked planes, no VALU re-split). dM arrives by LDS-DMA as whole
// rows into the (still empty) park region; the saved pre-activations arrive by LDS-DMA as whole 128-B lines two chunks
// ahead. LDS per wave: [parked planes 32 KB | CA chunks 2 x 4 KB] = all 160 KB of the CU.
// ---------------------------------------------------------------------------------
template <bool TRAIN, bool ADD_DM = false>
__global__ __launch_bounds__(256) void k_comb_bwd_p2(const float* __restrict__ dM, const float* __restrict__ XF,
                                                      const int* __restrict__ rev, const float* __restrict__ LNS,
                                                      const float* __restrict__ CA, const float* __restrict__ ln_g,
                                                      W2 w2b, W2 w0b, float* __restrict__ dcat, int64_t E,
                                                      float* __restrict__ t_da) {
    extern __shared__ __attribute__((aligned(16))) char cb_lds[];
    TRR_PROLOGUE(E);
    constexpr int NC = 2 * D / 32;
    char* const my = cb_lds + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 40960;
    f16x8* const pp = reinterpret_cast<f16x8*>(my);  // parked planes: [(2 hc + kb2) x (h, l)][64]
    float4* const park = reinterpret_cast<float4*>(my);  // (the epilogue's staging tile)
    const char* const cat = my + 32768;
    dma_tile128(dM, row0, E, __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)my), L);
    const unsigned cabase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)cat);
    const float* casrc[4];
#pragma unroll
    for (int i4 = 0; i4 < 4; i4++) {
        const int r = 8 * i4 + (L.lane >> 3);
        int64_t rr = row0 + r;
        rr = rr < E ? rr : E - 1;
        casrc[i4] = CA + rr * (2 * D) + 4 * ((L.lane & 7) ^ ((r >> 1) & 7));
    }
    auto dma_ca = [&](int hc, int buf) {
        hc = hc < NC ? hc : NC - 1;
#pragma unroll
        for (int i4 = 0; i4 < 4; i4++) glds16_trr(casrc[i4] + 32 * hc, cabase + buf * 4096 + i4 * 1024);
    };
    dma_ca(0, 0);
    dma_ca(1, 1);
#ifdef C_ABL_W0  // timing ablation (results are wrong): every weight block is one of two: the stream comes from the CU's L1
    auto aidx = [&](int b) { return (size_t)(b & 1) * 64 + L.lane; };
#else
    auto aidx = [&](int b) { return (size_t)b * 64 + L.lane; };
#endif  // W2^T tile hc, K block kb: b = 8 hc + kb
    // W0^T stream position p = 0 .. 31: K block p & 15 of output half p >> 4 (tile t at + t * 16 * 64)
#ifdef C_ABL_W0
    auto bidx = [&](int p) { return (size_t)(p & 1) * 64 + L.lane; };
#else
    auto bidx = [&](int p) { return ((size_t)(4 * (p >> 4)) * 16 + (p & 15)) * 64 + L.lane; };
#endif
    constexpr int RA = 4;
    WBlk2<1> ra[RA];
    WBlk2<4> rb[2];
#pragma unroll
    for (int b = 0; b < RA; b++) ld_blk2<1>(ra[b], w2b, aidx(b), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Split2<8> ms;
    float inv;  // dM is an adjoint: one power-of-two scale per row; da and dl carry it until they leave the kernel
    {
        float4 d[16];
        tile128_to_frag(d, my, L);
        float sc;
        inv = row_scale_pow2<16>(d, sc);
        split_frag2<8>(d, ms);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < 2; b++) ld_blk2<4>(rb[b], w0b, bidx(b), 16 * 64);
    f32x16 dl[4], dll[4];
    acc_zero<4>(dl);
    acc_zero<4>(dll);
    f32x16 t1, t1l;
    f16x8 dh, dlo;  // K block of the parked da planes, requested one slot ahead
    float4 cq;      // saved pre-activations of four elements
    auto rd_d = [&](int kblk) {  // kblk = 2 hc + kb2
        unsigned o = L.lane;  // opaque offset: read here, not hoisted or forwarded through registers
        asm volatile("" : "+v"(o));
        dh = pp[o + (2 * kblk) * 64];
        dlo = pp[o + (2 * kblk + 1) * 64];
    };
    auto rd_ca = [&](int buf, int q) {
        unsigned o = 128 * L.r + 16 * ((2 * q + L.h) ^ ((L.r >> 1) & 7));
        asm volatile("" : "+v"(o));
        cq = *reinterpret_cast<const float4*>(cat + buf * 4096 + o);
    };
    auto t1_slot = [&](int c, int kb) {
        WBlk2<1>& wb = ra[kb % RA];
        if (kb == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; r++) z[r] = 0.f;
            t1l = PET_MFMA_H(wb.l[0], ms.h[0], z);
            t1 = PET_MFMA_H(wb.h[0], ms.h[0], z);
        } else {
            t1l = PET_MFMA_H(wb.l[0], ms.h[kb], t1l);
            t1 = PET_MFMA_H(wb.h[0], ms.h[kb], t1);
        }
        t1l = PET_MFMA_H(wb.h[0], ms.l[kb], t1l);
        int nb = 8 * c + kb + RA;
        nb = nb < 8 * NC ? nb : 8 * NC - 1;  // past the end: the last block again (no branch in the loop body)
        ld_blk2<1>(wb, w2b, aidx(nb), 0);
    };
    // slot s (0..7) of the W0^T stream position pair (p0, p0 + 1): K block s >> 2, tile s & 3; `nxt` = K block to request
    // for the slot after this step (-1: none)
    auto dl_slot = [&](int p0, int s, int nxt) {
        const int st = s >> 2, t = s & 3;
        WBlk2<4>& wb = rb[st];
        dll[t] = PET_MFMA_H(wb.l[t], dh, dll[t]);
        dl[t] = PET_MFMA_H(wb.h[t], dh, dl[t]);
        dll[t] = PET_MFMA_H(wb.h[t], dlo, dll[t]);
        if (t == 3) {
            if (nxt >= 0) rd_d(nxt);
            int np = p0 + st + 2;
            np = np < 4 * NC ? np : 4 * NC - 1;
            ld_blk2<4>(wb, w0b, bidx(np), 16 * 64);
        }
    };
    // MODE 1: first iteration (no dl of a previous chunk), 2: last (no t1 of a next chunk, no further CA request)
    auto iteration = [&](auto mode, int hc) {
        constexpr int MODE = decltype(mode)::value;
        float da[16];  // element e = 4 q + c: hidden unit 8 q + 4 h + c of the chunk
        h16x2 sh[4], sl4[4];
        const int buf = hc & 1;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            // dl(hc - 1): K blocks 2 (hc - 1), 2 (hc - 1) + 1; the operand of the second is requested in slot 3
            if (MODE != 1) dl_slot(2 * (hc - 1), s, s == 3 ? 2 * (hc - 1) + 1 : -1);
#pragma unroll
            for (int e = 2 * s; e < 2 * s + 2; e++) {
                const float v = t1[e] + t1l[e] * (1.0f / 2048.0f);
                da[e] = v * silu_g_(f4c(cq, e & 3));
                asm volatile("" : "+v"(da[e]));  // computed in this slot
                if ((e & 3) == 3 && e < 15) rd_ca(buf, (e >> 2) + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TRAIN && valid) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4*>(t_da + row * (2 * D) + 32 * hc + 8 * q + 4 * L.h) =
                    make_float4(da[4 * q] * inv, da[4 * q + 1] * inv, da[4 * q + 2] * inv, da[4 * q + 3] * inv);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE != 2) t1_slot(hc + 1, k);
            // this buffer's next chunk (a clamped repeat past the end), requested BEHIND the ring requests that are waited
            // for within this iteration (K blocks 4..7, made in slots 8..11): the first wait that covers it is then the
            // one for a block requested in slots 12..15, eight slots into the next iteration
            if (k == 3 && MODE != 2) dma_ca(hc + 2, buf);
            {   // pair k of da(hc): values 2k, 2k + 1; K block 2 hc + (k >> 2) is complete after four pairs
                h16x2 hp, lp;
                split_pair_pinned(da[2 * k], da[2 * k + 1], hp, lp);
                sh[k & 3] = hp;
                sl4[k & 3] = lp;
                if ((k & 3) == 3) {
                    f16x8 fh, fl;
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        fh[2 * w] = sh[w][0]; fh[2 * w + 1] = sh[w][1];
                        fl[2 * w] = sl4[w][0]; fl[2 * w + 1] = sl4[w][1];
                    }
                    const int kblk = 2 * hc + (k >> 2);
                    pp[(2 * kblk) * 64 + L.lane] = fh;
                    pp[(2 * kblk + 1) * 64 + L.lane] = fl;
                }
            }
            if (k == 7) {
                rd_d(2 * hc);                          // operand of the next iteration's first dl step
                if (MODE != 2) rd_ca(buf ^ 1, 0);      // first four pre-activations of the next chunk
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    rd_ca(0, 0);
#pragma unroll
    for (int k = 0; k < 8; k++) {  // t1 of chunk 0
        t1_slot(0, k);
        __builtin_amdgcn_sched_barrier(0);
    }
    iteration(std::integral_constant<int, 1>{}, 0);
#pragma unroll 1
    for (int hc = 1; hc + 1 < NC; hc++) iteration(std::integral_constant<int, 0>{}, hc);
    iteration(std::integral_constant<int, 2>{}, NC - 1);
#pragma unroll
    for (int s = 0; s < 8; s++) dl_slot(2 * (NC - 1), s, s == 3 ? 2 * (NC - 1) + 1 : (s == 7 ? 0 : -1));
    fold_low<4>(dl, dll);
    acc_scale<4>(dl, inv);
    float4 wlo[16];
    acc_to_frag<4>(dl, wlo);
    acc_zero<4>(dl);
    acc_zero<4>(dll);
    // phase 2: the second half of the output over the parked planes (W0^T stream positions 16 .. 31)
#pragma unroll 1
    for (int hc = 0; hc < NC; hc++) {
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int nk = s == 3 ? 2 * hc + 1 : (s == 7 && hc + 1 < NC ? 2 * hc + 2 : -1);
            dl_slot(16 + 2 * hc, s, nk);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    fold_low<4>(dl, dll);
    acc_scale<4>(dl, inv);
    float4 whi[16], xo[16], xr[16];
    acc_to_frag<4>(dl, whi);
    load_rowfrag<16>(xo, XF, row, D, L.h);
    load_rowfrag<16>(xr, XF, (int64_t)rev[row], D, L.h);
    float4 dm[16];
    if (ADD_DM) request_rows_addend<16>(dm, L, [&](int r) { return dM + (row0 + r < E ? row0 + r : E - 1) * D; });
    const float mean = LNS[row * 2], rstd = LNS[row * 2 + 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {  // dyhat = dln * gamma; LayerNorm adjoint sums
        const float4 ga = *reinterpret_cast<const float4*>(ln_g + 8 * k + 4 * L.h);
        const float4 gb = *reinterpret_cast<const float4*>(ln_g + D + 8 * k + 4 * L.h);
        wlo[k].x *= ga.x; wlo[k].y *= ga.y; wlo[k].z *= ga.z; wlo[k].w *= ga.w;
        whi[k].x *= gb.x; whi[k].y *= gb.y; whi[k].z *= gb.z; whi[k].w *= gb.w;
        s1 += wlo[k].x + wlo[k].y + wlo[k].z + wlo[k].w + whi[k].x + whi[k].y + whi[k].z + whi[k].w;
        s2 += wlo[k].x * (xo[k].x - mean) + wlo[k].y * (xo[k].y - mean) + wlo[k].z * (xo[k].z - mean) +
              wlo[k].w * (xo[k].w - mean) + whi[k].x * (xr[k].x - mean) + whi[k].y * (xr[k].y - mean) +
              whi[k].z * (xr[k].z - mean) + whi[k].w * (xr[k].w - mean);
    }
    const float m1 = row_sum(s1) * (1.0f / 256.0f);
    const float m2 = row_sum(s2) * rstd * rstd * (1.0f / 256.0f);
    {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            wlo[k] = make_float4(rstd * (wlo[k].x - m1 - (xo[k].x - mean) * m2), rstd * (wlo[k].y - m1 - (xo[k].y - mean) * m2),
                                 rstd * (wlo[k].z - m1 - (xo[k].z - mean) * m2), rstd * (wlo[k].w - m1 - (xo[k].w - mean) * m2));
            whi[k] = make_float4(rstd * (whi[k].x - m1 - (xr[k].x - mean) * m2), rstd * (whi[k].y - m1 - (xr[k].y - mean) * m2),
                                 rstd * (whi[k].z - m1 - (xr[k].z - mean) * m2), rstd * (whi[k].w - m1 - (xr[k].w - mean) * m2));
        }
        // the parked chunks are consumed: the wave's park region is its staging tile for full-line stores (trr.h)
        float* otile = reinterpret_cast<float*>(park);
        // ADD_DM: dcat[p][:D] leaves as dM[p] + dcat[p][:D], the first two terms of dXF (k_dxf, pet_bwd.hip) in its order
        if (ADD_DM) store_rows_lines_add<16>(wlo, dm, otile, L, [&](int r) { return row0 + r < E ? dcat + (row0 + r) * (2 * D) : nullptr; });
        else store_rows_lines<16>(wlo, otile, L, [&](int r) { return row0 + r < E ? dcat + (row0 + r) * (2 * D) : nullptr; });
        store_rows_lines<16>(whi, otile, L, [&](int r) { return row0 + r < E ? dcat + (row0 + r) * (2 * D) + D : nullptr; });
    }
}

static inline W2 w2_of(const void* base, int n_tiles_dim, int k_dim) {
    const size_t n8 = (size_t)(n_tiles_dim / 32) * (k_dim / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}

bool trr_comb_bwd(const float* dM, const float* XF, const Graph& g, const GnnLayerW& G, const float* LNS,
                  const float* CA, float* dcat, int64_t E, float* t_da, hipStream_t st, bool add_dm) {
    if (!G.comb0.bwd2 || !G.comb2.bwd2 || E <= 0) return false;
    // bwd2 operands: tiles over k_in, K = n_out
    const W2 w2b = w2_of(G.comb2.bwd2, G.comb2.k_in, G.comb2.n_out), w0b = w2_of(G.comb0.bwd2, G.comb0.k_in, G.comb0.n_out);
    const int grid = cdiv(E, WG_ROWS);
    const size_t lds = (size_t)4 * 40960;  // per wave: parked da planes 32 KB, pre-activation chunks 2 x 4 KB
    if (t_da) {
        if (add_dm) return false;
        allow_big_lds(k_comb_bwd_p2<true>, lds);
        k_comb_bwd_p2<true><<<grid, 256, lds, st>>>(dM, XF, g.rev, LNS, CA, G.ln_g, w2b, w0b, dcat, E, t_da);
    } else if (add_dm) {
        allow_big_lds(k_comb_bwd_p2<false, true>, lds);
        k_comb_bwd_p2<false, true><<<grid, 256, lds, st>>>(dM, XF, g.rev, LNS, CA, G.ln_g, w2b, w0b, dcat, E, nullptr);
    } else {
        allow_big_lds(k_comb_bwd_p2<false>, lds);
        k_comb_bwd_p2<false><<<grid, 256, lds, st>>>(dM, XF, g.rev, LNS, CA, G.ln_g, w2b, w0b, dcat, E, nullptr);
    }
    return true;
}

}  // namespace pet
