// Edge/token-stream stages of PET in the transposed register-resident form (trr.h): one wave =
// 32 rows through a whole stage, no LDS, no barriers, weights streamed from L2 in fragment order.
// Same math and same saved tensors as the LDS-tile kernels in pet_fwd.hip / pet_bwd.hip (kept for
// A/B comparison, PET_HIP_TRR=0); reference line map in pet_fwd.hip.
#include <type_traits>
#include <stdlib.h>

#include "common.h"
#include "model.h"
#include "pet_ws.h"
#include "trr.h"

namespace pet {

#define TRR_PROLOGUE(NROWS)                                   \
    const RowLane L;                                          \
    const int64_t row0 = wave_row0();                         \
    if (row0 >= (NROWS)) return;                              \
    const bool valid = row0 + L.r < (NROWS);                  \
    const int64_t row = valid ? row0 + L.r : (NROWS) - 1

// ---------------------------------------------------------------------------------
// f16x3 versions (trr.h) of the 128-wide row GEMM and the stages built on it. `oscale` multiplies the finished
// accumulators (the inverse of a row_scale_pow2 applied to an adjoint input; 1 for forward activations).
// ---------------------------------------------------------------------------------
// UNROLLED: the column-group loop is fully unrolled, for epilogues that index register arrays with the group number
// (a runtime index would send those arrays to scratch memory)
template <int NC2, bool UNROLLED = false, int RING = 4, class Epilogue>
__device__ __forceinline__ void row_gemm128_h(const W2& w, const float* __restrict__ bias, const Split2<8>& xs,
                                              const RowLane& L, float oscale, Epilogue epi) {
#ifdef H_ABL_W0  // timing ablation (results are wrong): every weight block is block 0 -- the stream comes from the CU's L1
    auto widx = [&](int b) { return (size_t)(b & 1) * 64 + L.lane; };
#else
    auto widx = [&](int b) { return ((size_t)(2 * (b >> 3)) * 8 + (b & 7)) * 64 + L.lane; };
#endif
    WBlk2<2> ring[RING];  // RING (a power of two) weight blocks in flight
#pragma unroll
    for (int b = 0; b < RING; b++) ld_blk2<2>(ring[b], w, widx(b), 8 * 64);
    constexpr bool BIAS_AHEAD = RING >= 4;  // the two-waves-per-SIMD callers (RING 2) have no registers for it
    // a run-time `oscale` undoes a power-of-two scale of the INPUT rows: the bias must then be added after it
    const bool unit_scale = __builtin_constant_p(oscale) && oscale == 1.0f;
    float4 bnext[8];
    if (bias && BIAS_AHEAD && unit_scale) ld_bias<2>(bnext, bias, 0, L.h);
#pragma unroll UNROLLED ? NC2 : 1
    for (int c = 0; c < NC2; c++) {
        f32x16 acc[2], acl[2];
        acc_zero<2>(acl);
        if (bias && unit_scale) {
            if (!BIAS_AHEAD) ld_bias<2>(bnext, bias, 64 * c, L.h);
            acc_from<2>(acc, bnext);
            if (BIAS_AHEAD && c + 1 < NC2) ld_bias<2>(bnext, bias, 64 * (c + 1), L.h);
        } else {
            if (bias) ld_bias<2>(bnext, bias, 64 * c, L.h);
            acc_zero<2>(acc);
        }
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            WBlk2<2>& wb = ring[kb & (RING - 1)];
            mfma3<2>(acc, acl, wb, xs.h[kb], xs.l[kb]);
            const int nb = 8 * c + kb + RING;
            if (nb < 8 * NC2) ld_blk2<2>(wb, w, widx(nb), 8 * 64);
        }
        fold_low<2>(acc, acl);
        // forward callers pass the literal 1 and the multiply folds away; adjoint callers pass a per-row value and
        // always multiply: a test on it would be a divergent branch around spill code ("exec hazards")
        if (!unit_scale) {
            acc_scale<2>(acc, oscale);
            if (bias) {
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        acc[t][4 * q] += bnext[4 * t + q].x; acc[t][4 * q + 1] += bnext[4 * t + q].y;
                        acc[t][4 * q + 2] += bnext[4 * t + q].z; acc[t][4 * q + 3] += bnext[4 * t + q].w;
                    }
            }
        }
        epi(c, acc);
    }
}

// ---------------------------------------------------------------------------------
// "Shared weight stream" kernels (round 3; pet_config_set("lds_w", bits)). The TRR kernels above give every 32-row WAVE
// its own copy of the weight stream from L2: 196 KB of fragments against 64 KB of rows per tile in the QKV stage, i.e.
// three quarters of what passes the CU's vector-memory path (64 B / clk) is the same weights over and over, and that
// path is what the stage spends ~40 % of its time on. Here ONE workgroup of eight waves
// (256 rows, two waves per SIMD) streams each weight chunk ONCE, by LDS-DMA (global_load_lds_dwordx4: no registers, the
// lane-linear fragment order of the packed weights is exactly the LDS image), into a double buffer; the waves read
// their A fragments with ds_read_b128 (conflict-free: 16 B per lane, consecutive lanes) and meet at one barrier per
// chunk. Row fragments, split operands and accumulators stay in registers as in the TRR kernels: same arithmetic,
// bit-identical results.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void glds16_w(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
constexpr int SW_WAVES = 8;                       // waves per workgroup = 256 rows
constexpr int SW_CHUNK = 2 * 8 * 64;              // f16x8 entries of one plane of a 64-column x 128-K weight chunk (16 KB)
constexpr size_t SW_WBYTES = (size_t)2 * 2 * SW_CHUNK * 16;   // two buffers x two planes
template <bool LN>
__global__ __launch_bounds__(512) void k_qkv_s(const float* __restrict__ X, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, W2 win, const float* __restrict__ bin,
                                               float* __restrict__ QKV, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) char smem_sw[];
    const f16x8* wb = reinterpret_cast<const f16x8*>(smem_sw);  // [buffer][plane][tile 0..1][kb 0..7][lane]
    const RowLane L;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds = reinterpret_cast<float*>(smem_sw + SW_WBYTES) + wave * 32 * TILE_LD;
    const int64_t row0 = ((int64_t)blockIdx.x * SW_WAVES + wave) * WROWS;
    const int64_t rowc = row0 + L.r < R ? row0 + L.r : R - 1;   // waves past the end run along and store nothing
    const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem_sw);
    auto dma = [&](int c, int buf) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const f16x8* src = (p ? win.l : win.h) + (size_t)c * SW_CHUNK + threadIdx.x;
#pragma unroll
            for (int r = 0; r < 2; r++)
                glds16_w(src + r * 512, wbase + (unsigned)(((buf * 2 + p) * SW_CHUNK + r * 512 + wave * 64) * 16));
        }
    };
    dma(0, 0);
    Split2<8> xs;
    {
        float4 x[16];
        load_rowfrag<16>(x, X, rowc, D, L.h);
        norm_frag<16, LN>(x, gamma, beta, L.h);
        split_frag2<8>(x, xs);
    }
    float4 yprev[8];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < 6; c++) {
        if (c + 1 < 6) dma(c + 1, (c + 1) & 1);
        // the previous chunk's 64 columns leave while this chunk's MFMAs run
        if (c > 0) store_tile64_lines(yprev, lds, QKV + 64 * (c - 1), row0, R, 3 * D, L);
        const f16x8* bh = wb + (size_t)((c & 1) * 2) * SW_CHUNK + L.lane;
        const f16x8* bl = bh + SW_CHUNK;
        f32x16 acc[2], acl[2];
        acc_bias<2>(acc, bin, 64 * c, L.h);
        acc_zero<2>(acl);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            WBlk2<2> w;
#pragma unroll
            for (int t = 0; t < 2; t++) { w.h[t] = bh[(t * 8 + kb) * 64]; w.l[t] = bl[(t * 8 + kb) * 64]; }
            mfma3<2>(acc, acl, w, xs.h[kb], xs.l[kb]);
        }
        fold_low<2>(acc, acl);
        acc_to_frag<2>(acc, yprev);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next chunk has landed (and this wave's stores are out)
        __syncthreads();                                    // ... for every wave; and every wave is done with this buffer
    }
    store_tile64_lines(yprev, lds, QKV + 64 * 5, row0, R, 3 * D, L);
}

__global__ __launch_bounds__(256, 2) void k_oproj_h(const float* __restrict__ AO, const float* __restrict__ X, W2 wo,
                                                  const float* __restrict__ bo, float* __restrict__ X1,
                                                  float* __restrict__ OC, int64_t E, int64_t R) {
    PET_TRR_ROWS_LDS();
    TRR_PROLOGUE(R);
    Split2<8> xs;
    float inv;  // attention outputs are not normalised rows: scale them like an adjoint (the bias is added after)
    {
        float4 a[16];
        load_rows_lines128(a, lds_tile, AO, row0, R, L);
        float sc;
        inv = row_scale_pow2<16>(a, sc);
        split_frag2<8>(a, xs);
    }
    row_gemm128_h<2>(wo, nullptr, xs, L, inv, [&](int c, f32x16 (&acc)[2]) {
        {   // + bias, which must not be scaled with the row
            float4 bb[8];
            ld_bias<2>(bb, bo, 64 * c, L.h);
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    acc[t][4 * q] += bb[4 * t + q].x; acc[t][4 * q + 1] += bb[4 * t + q].y;
                    acc[t][4 * q + 2] += bb[4 * t + q].z; acc[t][4 * q + 3] += bb[4 * t + q].w;
                }
        }
        float4 y[8];
        acc_to_frag<2>(acc, y);
        if (valid && row < E) {
            float4 xr[8];
            load_rowfrag<8>(xr, X + 64 * c, row, D, L.h);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                y[k].x += xr[k].x; y[k].y += xr[k].y; y[k].z += xr[k].z; y[k].w += xr[k].w;
            }
        }
        store_rows_lines<8>(y, lds_tile, L, [&](int r) -> float* {
            const int64_t rw = row0 + r;
            return rw < E ? X1 + rw * D + 64 * c : (rw < R ? OC + (rw - E) * D + 64 * c : nullptr);
        });
    });
}

__global__ __launch_bounds__(256, 2) void k_oproj_bwd_h(const float* __restrict__ dX1, const float* __restrict__ dOC,
                                                      W2 wob, float* __restrict__ dAO, int64_t E, int64_t R) {
    PET_TRR_ROWS_LDS();
    TRR_PROLOGUE(R);
    Split2<8> xs;
    float inv;
    {
        float4 d[16];
        load_rows_lines(d, lds_tile, L, [&](int r) {
            const int64_t rw = row0 + r < R ? row0 + r : R - 1;
            return rw < E ? dX1 + rw * D : dOC + (rw - E) * D;
        });
        float sc;
        inv = row_scale_pow2<16>(d, sc);
        split_frag2<8>(d, xs);
    }
    row_gemm128_h<2>(wob, nullptr, xs, L, inv, [&](int c, f32x16 (&acc)[2]) {
        float4 y[8];
        acc_to_frag<2>(acc, y);
        store_tile64_lines(y, lds_tile, dAO + 64 * c, row0, R, D, L);
    });
}

// dXin = (row < E ? dX1 : 0) + RMSNorm^T(dQKV Win): K = 384 in three 128-wide slices. One power-of-two scale per
// row: it only ever shrinks (a later slice with larger entries rescales the accumulators, exactly).
template <bool LN>
__global__ __launch_bounds__(256) void k_qkv_bwd_h(const float* __restrict__ dQKV, const float* __restrict__ X,
                                                    const float* __restrict__ gamma, W2 winb,
                                                    const float* __restrict__ dX1, float* __restrict__ dXin, int64_t E,
                                                    int64_t R) {
    PET_TRR_ROWS_LDS();
    TRR_PROLOGUE(R);
    auto widx = [&](int b) { return (size_t)b * 64 + L.lane; };  // tile 0; tile t at + t * 24 * 64
    WBlk2<4> ring[4];
#pragma unroll
    for (int b = 0; b < 4; b++) ld_blk2<4>(ring[b], winb, widx(b), 24 * 64);
    f32x16 dn[4], dnl[4];
    acc_zero<4>(dn);
    acc_zero<4>(dnl);
    float4 d[16];
    load_rowfrag<16>(d, dQKV, row, 3 * D, L.h);
    float scale = 0.f, inv = 0.f;  // scale applied to what the accumulators hold, and its inverse
#pragma unroll 1
    for (int ks = 0; ks < 3; ks++) {
        Split2<8> xs;
        {   // branch-free per row (a divergent branch around accumulator code costs more than the multiplies)
            float sc;
            const float iv = row_pow2<16>(d, sc);
            const bool shrink = ks == 0 || sc < scale;
            const float sc_eff = shrink ? sc : scale;
            if (ks > 0) {  // larger entries than before: bring the running sums to the new scale (factor 1 otherwise)
                const float f = sc_eff * inv;
                acc_scale<4>(dn, f);
                acc_scale<4>(dnl, f);
            }
            scale = sc_eff;
            inv = shrink ? iv : inv;
#pragma unroll
            for (int kg = 0; kg < 16; kg++) { d[kg].x *= sc_eff; d[kg].y *= sc_eff; d[kg].z *= sc_eff; d[kg].w *= sc_eff; }
            split_frag2<8>(d, xs);
        }
        if (ks + 1 < 3) load_rowfrag<16>(d, dQKV + 128 * (ks + 1), row, 3 * D, L.h);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            WBlk2<4>& wb = ring[kb & 3];
            mfma3<4>(dn, dnl, wb, xs.h[kb], xs.l[kb]);
            const int nb = 8 * ks + kb + 4;
            if (nb < 24) ld_blk2<4>(wb, winb, widx(nb), 24 * 64);
        }
    }
    fold_low<4>(dn, dnl);
    acc_scale<4>(dn, inv);
    float4 w[16], x[16];
    acc_to_frag<4>(dn, w);
    load_rowfrag<16>(x, X, row, D, L.h);
#pragma unroll
    for (int kg = 0; kg < 16; kg++) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + 8 * kg + 4 * L.h);
        w[kg].x *= g.x; w[kg].y *= g.y; w[kg].z *= g.z; w[kg].w *= g.w;
    }
    norm_bwd_frag<16, LN>(w, x);
    if (valid && row < E) {
        load_rowfrag<16>(x, dX1, row, D, L.h);
#pragma unroll
        for (int kg = 0; kg < 16; kg++) {
            w[kg].x += x[kg].x; w[kg].y += x[kg].y; w[kg].z += x[kg].z; w[kg].w += x[kg].w;
        }
    }
    store_rows_lines<16>(w, lds_tile, L, [&](int r) { return row0 + r < R ? dXin + (row0 + r) * D : nullptr; });
}

// f16x2 planes of a Lin (abi.hip k_pack2h): two consecutive arrays of n8 fragments
static inline W2 w2_fwd(const Lin& L) {
    const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(L.fwd2);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}
static inline W2 w2_bwd(const Lin& L) {
    const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(L.bwd2);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}
static inline W2 w2_wc(const GnnLayerW& G) {  // one 32-row tile, K = D
    const f16x8* b = reinterpret_cast<const f16x8*>(G.wc2);
    W2 w; w.h = b; w.l = b + (size_t)(D / 16) * 64;
    return w;
}
// ---------------------------------------------------------------------------------
// k_emlp_p2: the edge MLP as a software-pipelined kernel (default since round 3; pet_config_set("emlp_pipe", 0) restores
// the persistent k_emlp_h). Same scheme as k_emlp_bwd_p2 below: the three stages of a hidden chunk -- [v; g] = Win xn
// (48 MFMAs), u = v sigma(g) + operand split (VALU), out += Wout u (24 MFMAs) -- depend on each other and a wave issues
// in order, so iteration hc is 24 slots of one f16x3 MFMA triple followed by VALU work that does not depend on it:
//   slots  0..7   out GEMM of chunk hc - 1 (2 K blocks x 4 tiles)  | slot s: pre-activations 2s, 2s + 1 of chunk hc folded
//   slots  8..23  [v; g] GEMM of chunk hc + 1 (8 K blocks x 2)      | slot a: u of element a; odd a: split of a pair;
//                                                                   |         the saved [v; g] leave as whole 128-B lines
// x arrives by LDS-DMA (whole rows) and is parked, normalised and split, over its own tile; the residual and the output
// bias are the INITIAL VALUE of the out accumulators; the split u operand (2 K blocks) and a private copy of the
// input bias live in LDS. Rings: Win blocks four K blocks (8 slots) ahead, the Wout fragments of a chunk requested in
// slots 8..15 of the iteration in which its u is made (they are used in slots 0..7 of the next one).
// LDS per wave: [x tile / split planes 16 KB | u operand 4 KB | store staging tile 4.5 KB | input bias 2 KB].
// ---------------------------------------------------------------------------------
constexpr int EP2_WAVE_LDS = 16384 + 4096 + 32 * TILE32_LD * 4 + 2 * DFF * 4;
template <bool LN>
__global__ __launch_bounds__(256) void k_emlp_p2(const float* __restrict__ X1, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, W2 win, const float* __restrict__ bin,
                                                  W2 wout, const float* __restrict__ bout, float* __restrict__ VG,
                                                  float* __restrict__ X2, int64_t E) {
    extern __shared__ __attribute__((aligned(16))) char ep_lds[];
    TRR_PROLOGUE(E);
    constexpr int NC = DFF / 32;
    static_assert(NC >= 3, "first and last iteration peeled");
    char* const my = ep_lds + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * EP2_WAVE_LDS;
    f16x8* const xsp = reinterpret_cast<f16x8*>(my);             // after the split: [8 K blocks x (h, l)][64]
    f16x8* const usp = reinterpret_cast<f16x8*>(my + 16384);     // [2 K blocks x (h, l)][64]
    float* const otile = reinterpret_cast<float*>(my + 16384 + 4096);
    float* const bias = reinterpret_cast<float*>(my + 16384 + 4096 + 32 * TILE32_LD * 4);
#ifndef P2_ABL_NOPROLOG
    dma_tile128(X1, row0, E, __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)my), L);
#endif
    {   // private copy of the input bias [v 256 | g 256]
        const float4* b4 = reinterpret_cast<const float4*>(bin);
        reinterpret_cast<float4*>(bias)[L.lane] = b4[L.lane];
        reinterpret_cast<float4*>(bias)[64 + L.lane] = b4[64 + L.lane];
    }
    auto widx = [&](int b) { return (size_t)b * 64 + L.lane; };  // Win: v tile hc, K block kb at b = 8 hc + kb; g tile at + TS
    constexpr size_t TS = (size_t)NC * 8 * 64;
    WBlk2<2> rw[4];  // Win ring: K blocks kb .. kb + 3
#pragma unroll
    for (int b = 0; b < 4; b++) ld_blk2<2>(rw[b], win, widx(b), TS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA is not visible to the compiler's own bookkeeping
    f32x16 out[4], outl[4];
    {
        float4 x[16];
        tile128_to_frag(x, my, L);
        acc_bias<4>(out, bout, 0, L.h);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {  // the residual rides in the accumulator
                out[t][4 * q] += x[4 * t + q].x; out[t][4 * q + 1] += x[4 * t + q].y;
                out[t][4 * q + 2] += x[4 * t + q].z; out[t][4 * q + 3] += x[4 * t + q].w;
            }
        norm_frag<16, LN>(x, gamma, beta, L.h);
        Split2<8> t;
        split_frag2<8>(x, t);
#pragma unroll
        for (int k = 0; k < 8; k++) { xsp[(2 * k) * 64 + L.lane] = t.h[k]; xsp[(2 * k + 1) * 64 + L.lane] = t.l[k]; }
    }
    acc_zero<4>(outl);
    __builtin_amdgcn_sched_barrier(0);
    // Wout fragments of one chunk, one (K block, tile) pair per slot of the out GEMM: ring entry s = 4 kb2 + tile
    f16x8 roh[8], rol[8];
    auto ld_wout = [&](int c, int s) {
        const size_t i = ((size_t)(s & 3) * (DFF / 16) + 2 * c + (s >> 2)) * 64 + L.lane;
        roh[s] = wout.h[i];
        rol[s] = wout.l[i];
    };
    f32x16 vg[2], vgl[2];  // [v; g] of the chunk the element slices work on; rebuilt for the next chunk in slots 8..23
    // LDS operands are requested one slot ahead of the MFMAs that use them
    f16x8 xh, xl;  // K block of the parked xn planes
    f16x8 uh, ul;  // K block of the split u operand
    auto rd_x = [&](int kb) {
        unsigned o = L.lane;  // opaque offset: read here, not hoisted into registers
        asm volatile("" : "+v"(o));
        xh = xsp[o + (2 * kb) * 64];
        xl = xsp[o + (2 * kb + 1) * 64];
    };
    auto rd_u = [&](int kb2) {
        unsigned o = L.lane;
        asm volatile("" : "+v"(o));
        uh = usp[o + (2 * kb2) * 64];
        ul = usp[o + (2 * kb2 + 1) * 64];
    };
    // slot a (0..15) of the [v; g] GEMM of chunk c: K block a >> 1, tile a & 1 (0: v, 1: g)
    auto vg_slot = [&](int c, int a) {
        const int kb = a >> 1, t = a & 1;
        WBlk2<2>& wb = rw[kb & 3];
        if (kb == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; r++) z[r] = 0.f;
            vgl[t] = PET_MFMA_H(wb.l[t], xh, z);
            vg[t] = PET_MFMA_H(wb.h[t], xh, z);
        } else {
            vgl[t] = PET_MFMA_H(wb.l[t], xh, vgl[t]);
            vg[t] = PET_MFMA_H(wb.h[t], xh, vg[t]);
        }
        vgl[t] = PET_MFMA_H(wb.h[t], xl, vgl[t]);
        if (t == 1) {
            if (kb < 7) rd_x(kb + 1);
            int nb = 8 * c + kb + 4;
            nb = nb < 8 * NC ? nb : 8 * NC - 1;
            ld_blk2<2>(wb, win, widx(nb), TS);
        }
    };
    // slot s (0..7) of the out GEMM: K block s >> 2 of the u operand, output tile s & 3
    auto out_slot = [&](int s) {
        const int t = s & 3;
        outl[t] = PET_MFMA_H(rol[s], uh, outl[t]);
        out[t] = PET_MFMA_H(roh[s], uh, out[t]);
        outl[t] = PET_MFMA_H(roh[s], ul, outl[t]);
        if (s == 3) rd_u(1);
    };
    // MODE 1: first iteration (no out GEMM of a previous chunk), 2: last (no [v; g] GEMM of a next chunk)
    auto iteration = [&](auto mode, int hc) {
        constexpr int MODE = decltype(mode)::value;
        float vv[16], gg[16];  // pre-activations of the chunk: element e = 4 q + c is hidden unit 8 q + 4 h + c
        float4 bv, bg;
#pragma unroll
        for (int sl = 0; sl < 8; sl++) {
            if (MODE != 1) out_slot(sl);
#pragma unroll
            for (int e = 2 * sl; e < 2 * sl + 2; e++) {
                if ((e & 3) == 0) {
                    bv = *reinterpret_cast<const float4*>(bias + 32 * hc + 8 * (e >> 2) + 4 * L.h);
                    bg = *reinterpret_cast<const float4*>(bias + DFF + 32 * hc + 8 * (e >> 2) + 4 * L.h);
                }
                vv[e] = vg[0][e] + vgl[0][e] * (1.0f / 2048.0f) + f4c(bv, e & 3);
                gg[e] = vg[1][e] + vgl[1][e] * (1.0f / 2048.0f) + f4c(bg, e & 3);
                asm volatile("" : "+v"(vv[e]), "+v"(gg[e]));  // computed in this slot
            }
            if (sl == 7 && MODE != 2) rd_x(0);
            __builtin_amdgcn_sched_barrier(0);
        }
        h16x2 sh[4], sl4[4];  // the fragment pair under construction
        float u0 = 0.f;
#pragma unroll
        for (int a = 0; a < 16; a++) {
            if (MODE != 2) vg_slot(hc + 1, a);
            if (a < 8) ld_wout(hc, a);  // this chunk's fragments: its out GEMM runs in slots 0..7 of the next iteration
            const float u = vv[a] * sigm_(gg[a]);
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
            if (VG && (a == 3 || a == 11)) {  // the saved pre-activations: whole 128-B lines through the staging tile
                float4 t4[4];
                const float* src = a == 3 ? vv : gg;
#pragma unroll
                for (int q = 0; q < 4; q++) t4[q] = make_float4(src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]);
                store_tile32_lines(t4, otile, VG + (a == 3 ? 0 : DFF) + 32 * hc, row0, E, 2 * DFF, L);
            }
            if (a == 15) rd_u(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    rd_x(0);
#pragma unroll
    for (int a = 0; a < 16; a++) {  // [v; g] of chunk 0
        vg_slot(0, a);
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
        store_tile32_lines(t4, otile, X2 + 32 * t, row0, E, D, L);
    }
}

// ---------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------
// k_emlp_bwd_p2: the edge-MLP adjoint as a software-pipelined kernel (default since round 3;
// pet_config_set("emlp_bwd_pipe", 0) restores k_emlp_bwd_h). Two things k_emlp_bwd_h paid for at one wave per SIMD:
//  * vmcnt retires IN ORDER (loads and stores alike on gfx950), so the first wait for a weight block that was requested
//    AFTER an HBM load also waits for that HBM load: with weight rings that reach ~24 MFMAs (0.3 us) ahead, the HBM
//    latency of the VG chunk was exposed once per hidden chunk and that of X1 / the dY re-read once more each per tile
//    (timing ablations, 8 x 10k atoms, ms per step: 7.36 full, 6.92 without the two epilogue loads, 6.01 without the VG
//    loads, 5.41 without both);
//  * the three stages of a hidden chunk depend on each other and a wave issues in order, so the matrix pipe idled
//    during the VALU stage and the VALU during both GEMMs.
// ---------------------------------------------------------------------------------
// dY arrives by LDS-DMA as whole 512-B rows and is parked, split, over its own fp32 tile (the residual is rebuilt from
// the planes: h + 2^-11 l', 22 bits of the scaled row); the VG chunks arrive by LDS-DMA as whole 128-B lines into two
// 8 KB buffers, two chunks ahead; the X1 tile follows into the same buffers in the last two iterations, so the
// epilogue makes no memory round trip. The chunk loop: the three stages of a hidden chunk are du = Wout^T dY (24
// MFMAs), the SwiGLU adjoint + operand split (VALU), dn += [dv | dg] Win (48 MFMAs). Iteration hc is 24 "slots" of one
// f16x3 MFMA triple (96 matrix-pipe cycles) FOLLOWED in the instruction stream by a slice of VALU work that does not
// depend on it (sched_barrier after every slot: the order below is the issue order):
//   slots  0..15  dn(hc - 1), four steps of four output tiles      | slot t: element t of the SwiGLU adjoint of chunk hc
//   slots  8..15                                                    |         + two values of the split of its first half
//   slots 16..23  du(hc + 1), eight K blocks                        |         two values of the split of the second half
// The split [dv | dg] operand lives in LDS (one 8 KB buffer per wave: K blocks 0 / 2 are rewritten in slots 11 / 15,
// K blocks 1 / 3 in 19 / 23, each after the dn step that reads the old one), du's accumulators are read directly by the
// element slices (the next du starts from a literal zero at slot 16). The VG request for chunk hc + 2 sits behind slot 19:
// behind this iteration's last Win^T request and behind the Wout^T requests that are waited for in this iteration, so
// the first wait that covers it is the one for the Win^T block of slot 8 of the next iteration.
template <bool TRAIN, bool LN, bool GATHER = false>
__global__ __launch_bounds__(256) void k_emlp_bwd_p2(const float* __restrict__ dY, const float* __restrict__ X1,
                                                      const float* __restrict__ VG, const float* __restrict__ gamma,
                                                      W2 woutb, W2 winb, float* __restrict__ dX1, int64_t E,
                                                      float* __restrict__ t_dvg, int ldy, const float* __restrict__ dY2,
                                                      const int* __restrict__ rev2) {
    extern __shared__ __attribute__((aligned(16))) char ebp_lds[];  // [4 waves][dY / split planes 16 KB | VG chunks 2 x 8 KB, X1 at the end | [dv | dg] 8 KB]
    TRR_PROLOGUE(E);
    constexpr int NC = DFF / 32;
    static_assert(NC % 2 == 0 && NC >= 4, "two VG buffers, first and last iteration peeled");
    char* const my = ebp_lds + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 40960;
    f16x8* const ysp = reinterpret_cast<f16x8*>(my);
    const char* const vgt = my + 16384;  // two chunk buffers [v 4 KB | g 4 KB]; the X1 tile in the last two iterations
    f16x8* const dsp = reinterpret_cast<f16x8*>(my + 32768);  // [4 K blocks x (h, l)][64]
    float4 y2[16];  // dY2 != nullptr: dY = dY[row] + dY2[rev2[row]] (rows of ldy floats) -- the ji gather of the combination
                    // adjoint (k_dxf, pet_bwd.hip) made while the tile is read
    int myrev = 0;  // lane l: rev2 of tile row l & 31
    if (GATHER) {
        const int64_t r2 = row0 + (L.lane & 31);
        myrev = rev2[r2 < E ? r2 : E - 1];
    }
    {
        const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)my);
#ifndef P2_ABL_NOPROLOG
        dma_tile128(dY, row0, E, base, L, ldy);
#endif
    }
    if (GATHER)  // whole 512-B rows, two per instruction (the line mapping of trr.h request_rows_lines)
        request_rows_lines(y2, L, [&](int r) { return dY2 + (int64_t)__shfl(myrev, r) * ldy; });
    // VG chunk hc -> buffer: whole 128-B lines (8 lanes per row and half, 8 rows per instruction); row r, 16-B piece p
    // of a half lands at byte 128 r + 16 (p ^ ((r >> 1) & 7)): conflict-free for the per-row reads below
    const unsigned vgbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)vgt);
    const float* vgsrc[4];
#pragma unroll
    for (int i4 = 0; i4 < 4; i4++) {
        const int r = 8 * i4 + (L.lane >> 3);
        int64_t rr = row0 + r;
        rr = rr < E ? rr : E - 1;
        vgsrc[i4] = VG + rr * (2 * DFF) + 4 * ((L.lane & 7) ^ ((r >> 1) & 7));
    }
    auto dma_vg = [&](int hc, int buf) {
        hc = hc < NC ? hc : NC - 1;  // past the end: the last chunk again (no branch)
#pragma unroll
        for (int i4 = 0; i4 < 4; i4++) {
            glds16_trr(vgsrc[i4] + 32 * hc, vgbase + buf * 8192 + i4 * 1024);
            glds16_trr(vgsrc[i4] + DFF + 32 * hc, vgbase + buf * 8192 + 4096 + i4 * 1024);
        }
    };
    // rows 16 half .. of the X1 tile into chunk buffer `half` (the two buffers together are a tile128 layout)
    auto dma_x1_half = [&](int half) {
#pragma unroll
        for (int j8 = 0; j8 < 8; j8++) {
            const int r = 16 * half + 2 * j8 + (L.lane >> 5);
            int64_t rr = row0 + r;
            rr = rr < E ? rr : E - 1;
            glds16_trr(X1 + rr * 128 + 4 * ((L.lane & 31) ^ (r & 15)), vgbase + half * 8192 + j8 * 1024);
        }
    };
    dma_vg(0, 0);
    dma_vg(1, 1);
    auto aidx = [&](int b) { return (size_t)b * 64 + L.lane; };
    // Win^T stream position b = 4 hc + s -> K block (s = 0: dv first half, 1: dg first half, 2: dv second, 3: dg second)
    auto bkb = [&](int b) { const int hc = b >> 2, st = b & 3; return (st & 1 ? 16 : 0) + 2 * hc + (st >> 1); };
    auto bidx = [&](int b) { return (size_t)bkb(b) * 64 + L.lane; };
    constexpr int RA = 4;  // Wout^T ring: blocks kb .. kb + 3 (eight blocks do not fit the register file next to the rest)
    WBlk2<1> ra[RA];
    WBlk2<4> rb[2];
#pragma unroll
    for (int b = 0; b < RA; b++) ld_blk2<1>(ra[b], woutb, aidx(b), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float inv;
    {
        float4 dy[16];
        tile128_to_frag(dy, my, L);
        if (GATHER) {  // the gathered rows take the tile's place (same swizzle) and are added as fragments
            __builtin_amdgcn_wave_barrier();
            const int rr = L.lane >> 5, pc = L.lane & 31;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int r = 2 * j + rr;
                *reinterpret_cast<float4*>(my + 512 * r + 16 * (pc ^ (r & 15))) = y2[j];
            }
            __builtin_amdgcn_wave_barrier();
            const char* rowp = my + 512 * L.r;
            const int sw = L.r & 15;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float4 g = *reinterpret_cast<const float4*>(rowp + 16 * ((2 * k + L.h) ^ sw));
                dy[k] = make_float4(dy[k].x + g.x, dy[k].y + g.y, dy[k].z + g.z, dy[k].w + g.w);
            }
            __builtin_amdgcn_wave_barrier();
        }
        float sc;
        inv = row_scale_pow2<16>(dy, sc);
        Split2<8> t;
        split_frag2<8>(dy, t);
#pragma unroll
        for (int k = 0; k < 8; k++) { ysp[(2 * k) * 64 + L.lane] = t.h[k]; ysp[(2 * k + 1) * 64 + L.lane] = t.l[k]; }
    }
    __builtin_amdgcn_sched_barrier(0);  // the Win^T ring is requested only now: the split above needs the registers
#pragma unroll
    for (int b = 0; b < 2; b++) ld_blk2<4>(rb[b], winb, bidx(b), 32 * 64);
    f32x16 dn[4], dnl[4];
    acc_zero<4>(dn);
    acc_zero<4>(dnl);
    f32x16 du, dul;  // du of the chunk the element slices work on; rebuilt for the next chunk in slots 16..23
    // LDS operands are requested one slot ahead of the MFMAs / elements that use them (the order inside a slot is
    // fixed, so a read issued in its own slot would expose the LDS latency 24 times per iteration)
    f16x8 yh, yl;   // K block of the parked dY planes
    f16x8 dh, dl;   // K block of the split [dv | dg] operand
    float4 vq, gq;  // saved pre-activations of four elements
    auto rd_y = [&](int kb) {
        unsigned yo = L.lane;  // opaque offset: read here, not hoisted into registers
        asm volatile("" : "+v"(yo));
        yh = ysp[yo + (2 * kb) * 64];
        yl = ysp[yo + (2 * kb + 1) * 64];
    };
    auto rd_d = [&](int st) {  // step st of dn takes K block (0, 2, 1, 3)[st]
        const int jj = ((st & 1) << 1) | (st >> 1);
        unsigned o = L.lane;
        asm volatile("" : "+v"(o));
        dh = dsp[o + (2 * jj) * 64];
        dl = dsp[o + (2 * jj + 1) * 64];
    };
    auto rd_vg = [&](int buf, int q) {
        unsigned o = 128 * L.r + 16 * ((2 * q + L.h) ^ ((L.r >> 1) & 7));
        asm volatile("" : "+v"(o));
        const char* pv = vgt + buf * 8192 + o;
        vq = *reinterpret_cast<const float4*>(pv);
        gq = *reinterpret_cast<const float4*>(pv + 4096);
    };
    // one K block of du(c): block 8 c + kb of the Wout^T stream, ring slot kb % RA, refilled RA blocks ahead
    auto du_slot = [&](int c, int kb) {
        WBlk2<1>& wb = ra[kb % RA];
        if (kb == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; r++) z[r] = 0.f;
            dul = PET_MFMA_H(wb.l[0], yh, z);
            du = PET_MFMA_H(wb.h[0], yh, z);
        } else {
            dul = PET_MFMA_H(wb.l[0], yh, dul);
            du = PET_MFMA_H(wb.h[0], yh, du);
        }
        dul = PET_MFMA_H(wb.h[0], yl, dul);
        if (kb < 7) rd_y(kb + 1);
        int nb = 8 * c + kb + RA;
        nb = nb < 8 * NC ? nb : 8 * NC - 1;  // past the end: the last block again
        ld_blk2<1>(wb, woutb, aidx(nb), 0);
    };
    // output tile t of step st of dn(c)
    auto dn_slot = [&](int c, int st, int t) {
        WBlk2<4>& wb = rb[st & 1];
        dnl[t] = PET_MFMA_H(wb.l[t], dh, dnl[t]);
        dn[t] = PET_MFMA_H(wb.h[t], dh, dn[t]);
        dnl[t] = PET_MFMA_H(wb.h[t], dl, dnl[t]);
        if (t == 3) {
            if (st < 3) rd_d(st + 1);
            int nb = 4 * c + st + 2;
            nb = nb < 4 * NC ? nb : 4 * NC - 1;
            ld_blk2<4>(wb, winb, bidx(nb), 32 * 64);
        }
    };
    // MODE 1: first iteration (no dn of a previous chunk), 2: last (no du of a next chunk, no further VG request).
    // On entry: (vq, gq) = elements 0..3 of this chunk, (dh, dl) = operand of dn step 0.
    auto iteration = [&](auto mode, int hc, int buf) {
        constexpr int MODE = decltype(mode)::value;
        float dv[16], dg[16];  // element e = 4 q + c: hidden unit 8 q + 4 h + c of the chunk
        h16x2 sh[4], sl[4];    // the fragment pair under construction
        // values n, n + 1 (of 0..15) of half hf: n < 8: dv of elements 8 hf + n (K block hf), else dg (K block 2 + hf)
        auto split_pair = [&](int hf, int n) {
            const float x0 = n < 8 ? dv[8 * hf + n] : dg[8 * hf + n - 8];
            const float x1 = n < 8 ? dv[8 * hf + n + 1] : dg[8 * hf + n - 7];
            h16x2 hp, lp;
            split_pair_pinned(x0, x1, hp, lp);
            sh[(n & 7) >> 1] = hp;
            sl[(n & 7) >> 1] = lp;
            if ((n & 7) == 6) {
                f16x8 fh, fl;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    fh[2 * u] = sh[u][0]; fh[2 * u + 1] = sh[u][1];
                    fl[2 * u] = sl[u][0]; fl[2 * u + 1] = sl[u][1];
                }
                const int kbk = n < 8 ? hf : 2 + hf;
                dsp[(2 * kbk) * 64 + L.lane] = fh;
                dsp[(2 * kbk + 1) * 64 + L.lane] = fl;
            }
        };
#pragma unroll
        for (int t = 0; t < 16; t++) {
            if (MODE != 1) dn_slot(hc - 1, t >> 2, t & 3);
            if (t == 15 && MODE != 2) rd_y(0);
            {   // element t
                const int q = t >> 2, c = t & 3;
                const float d = du[t] + dul[t] * (1.0f / 2048.0f);
                const float sg = sigm_(f4c(gq, c));
                dv[t] = d * sg;
                dg[t] = d * f4c(vq, c) * sg * (1.f - sg);
                asm volatile("" : "+v"(dv[t]), "+v"(dg[t]));  // computed in this slot
                if (c == 3 && q < 3) rd_vg(buf, q + 1);
            }
            if (t >= 8) split_pair(0, 2 * (t - 8));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TRAIN && valid) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                *reinterpret_cast<float4*>(t_dvg + row * (2 * DFF) + 32 * hc + 8 * q + 4 * L.h) =
                    make_float4(dv[4 * q] * inv, dv[4 * q + 1] * inv, dv[4 * q + 2] * inv, dv[4 * q + 3] * inv);
                *reinterpret_cast<float4*>(t_dvg + row * (2 * DFF) + DFF + 32 * hc + 8 * q + 4 * L.h) =
                    make_float4(dg[4 * q] * inv, dg[4 * q + 1] * inv, dg[4 * q + 2] * inv, dg[4 * q + 3] * inv);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            if (MODE != 2) du_slot(hc + 1, t);
            if (t == 3) {
                // this buffer's next chunk; in the last two iterations the halves of the X1 tile for the epilogue instead.
                // Requested BEHIND the Wout^T requests that are waited for within this iteration (K blocks 4..7, made in
                // slots 16..19): the first wait that covers it is the one for the Win^T block of slot 8 of the next
                // iteration, twelve slots away (right after slot 15 it was the Wout^T wait of slot 20)
                if (MODE == 2 || hc + 2 == NC) dma_x1_half(buf);  // (wave-uniform branch)
                else dma_vg(hc + 2, buf);
            }
            if (t == 7) {
                rd_d(0);
                if (MODE != 2) rd_vg(buf ^ 1, 0);
            }
            split_pair(1, 2 * t);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    rd_y(0);
    rd_vg(0, 0);
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {  // du(0)
        du_slot(0, kb);
        __builtin_amdgcn_sched_barrier(0);
    }
    iteration(std::integral_constant<int, 1>{}, 0, 0);
#pragma unroll 1
    for (int hc = 1; hc + 1 < NC; hc += 2) {
        iteration(std::integral_constant<int, 0>{}, hc, 1);
        iteration(std::integral_constant<int, 0>{}, hc + 1, 0);
    }
    iteration(std::integral_constant<int, 2>{}, NC - 1, 1);
#pragma unroll
    for (int t = 0; t < 16; t++) dn_slot(NC - 1, t >> 2, t & 3);
    fold_low<4>(dn, dnl);
    acc_scale<4>(dn, inv);
    float4 w[16], x[16];
    acc_to_frag<4>(dn, w);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the second half of the X1 tile
    tile128_to_frag(x, vgt, L);
#pragma unroll
    for (int kg = 0; kg < 16; kg++) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + 8 * kg + 4 * L.h);
        w[kg].x *= g.x; w[kg].y *= g.y; w[kg].z *= g.z; w[kg].w *= g.w;
    }
    norm_bwd_frag<16, LN>(w, x);
    if (valid) {
        const float f = inv * (1.0f / 2048.0f);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {  // the residual: dY = inv (h + 2^-11 l')
            const f16x8 yh = ysp[(2 * kb) * 64 + L.lane], yl = ysp[(2 * kb + 1) * 64 + L.lane];
            w[2 * kb].x += (float)yh[0] * inv + (float)yl[0] * f; w[2 * kb].y += (float)yh[1] * inv + (float)yl[1] * f;
            w[2 * kb].z += (float)yh[2] * inv + (float)yl[2] * f; w[2 * kb].w += (float)yh[3] * inv + (float)yl[3] * f;
            w[2 * kb + 1].x += (float)yh[4] * inv + (float)yl[4] * f; w[2 * kb + 1].y += (float)yh[5] * inv + (float)yl[5] * f;
            w[2 * kb + 1].z += (float)yh[6] * inv + (float)yl[6] * f; w[2 * kb + 1].w += (float)yh[7] * inv + (float)yl[7] * f;
        }
        store_rowfrag<16>(w, dX1, row, D, L.h);
    }
}

// ---------------------------------------------------------------------------------
// compress stage (transformer.py:499-521) and its adjoint as TRR kernels on f16x3. One wave = 32 edges:
//   forward:  a0 = [v,d] Wc^T + Tbl[species] (+ M W0c^T);  e = SiLU(a0) W2^T + b2
//   adjoint:  da0 = (dE W2) . silu'(a0);  dgeo += da0 Wc;  dM += da0 W0c
// Same arithmetic as k_compress / k_compress_bwd (pet_fwd.hip / pet_bwd.hip), which stay selectable.
// ---------------------------------------------------------------------------------
template <bool FIRST>
__global__ __launch_bounds__(256, 2) void k_compress_h(const float4* __restrict__ geo, const int* __restrict__ sp_nbr,
                                                     const float* __restrict__ wc /*[D][4]*/,
                                                     const float* __restrict__ tbl /*[ns][D]*/,
                                                     const float* __restrict__ Min, W2 w0c, W2 w2,
                                                     const float* __restrict__ b2, float* __restrict__ a0_out,
                                                     float* __restrict__ Xout, int64_t E) {
    PET_TRR_ROWS_LDS();
    TRR_PROLOGUE(E);
    float4 a0[16];  // row fragment of the pre-activation: entries a0[kg] = features 8 kg + 4 h .. + 3
    if (!FIRST) {   // message term first: the geometry terms below then need no registers during the GEMM
        Split2<8> ms;
        float minv;
        {   // messages are an UN-normalised residual stream: a checkpoint may carry them at 1e-4 or 1e+4, outside the
            // range in which fp16 pieces keep fp32 accuracy, so the row goes in scaled by a power of two (exact)
            float4 mrow[16];
            load_rowfrag<16>(mrow, Min, row, D, L.h);
            float sc;
            minv = row_scale_pow2<16>(mrow, sc);
            split_frag2<8>(mrow, ms);
        }
        row_gemm128_h<2, true, 2>(w0c, nullptr, ms, L, minv, [&](int c, f32x16 (&acc)[2]) {
            acc_to_frag<2>(acc, &a0[8 * c]);
        });
    } else {
#pragma unroll
        for (int kg = 0; kg < 16; kg++) a0[kg] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        const float4 g = geo[row];
        const float* trow = tbl + (size_t)sp_nbr[row] * D;
#pragma unroll
        for (int kg = 0; kg < 16; kg++) {
            const int c = 8 * kg + 4 * L.h;
            const float4 t = *reinterpret_cast<const float4*>(trow + c);
            const float4 w0 = *reinterpret_cast<const float4*>(wc + 4 * c), w1 = *reinterpret_cast<const float4*>(wc + 4 * c + 4),
                         w2v = *reinterpret_cast<const float4*>(wc + 4 * c + 8), w3 = *reinterpret_cast<const float4*>(wc + 4 * c + 12);
            a0[kg].x += fmaf(g.w, w0.w, fmaf(g.z, w0.z, fmaf(g.y, w0.y, g.x * w0.x))) + t.x;
            a0[kg].y += fmaf(g.w, w1.w, fmaf(g.z, w1.z, fmaf(g.y, w1.y, g.x * w1.x))) + t.y;
            a0[kg].z += fmaf(g.w, w2v.w, fmaf(g.z, w2v.z, fmaf(g.y, w2v.y, g.x * w2v.x))) + t.z;
            a0[kg].w += fmaf(g.w, w3.w, fmaf(g.z, w3.z, fmaf(g.y, w3.y, g.x * w3.x))) + t.w;
            if ((kg & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // 20 float4 loads in flight, not 80
        }
    }
    if (a0_out) store_rows_lines<16>(a0, lds_tile, L, [&](int r) { return row0 + r < E ? a0_out + (row0 + r) * D : nullptr; });
    Split2<8> ss;
    float sinv;
    {
#pragma unroll
        for (int kg = 0; kg < 16; kg++)
            a0[kg] = make_float4(silu_(a0[kg].x), silu_(a0[kg].y), silu_(a0[kg].z), silu_(a0[kg].w));
        float sc;
        sinv = row_scale_pow2<16>(a0, sc);  // silu(a0) is as large as a0
        split_frag2<8>(a0, ss);
    }
    row_gemm128_h<2, true, 2>(w2, b2, ss, L, sinv, [&](int c, f32x16 (&acc)[2]) {
        float4 y[8];
        acc_to_frag<2>(acc, y);
        store_tile64_lines(y, lds_tile, Xout + 64 * c, row0, E, D, L);
    });
}

template <bool FIRST, bool TRAIN>
__global__ __launch_bounds__(256, 2) void k_compress_bwd_h(const float* __restrict__ dXe, const float* __restrict__ a0,
                                                         W2 w2b, W2 wcp /* Wc^T padded to [32][D] */, W2 w0cb,
                                                         float* __restrict__ dgeo, float* __restrict__ dM, int64_t E,
                                                         float* __restrict__ t_da0) {
    PET_TRR_ROWS_LDS();
    TRR_PROLOGUE(E);
    Split2<8> ys;
    float inv;
    {
        float4 d[16];
        load_rowfrag<16>(d, dXe, row, D, L.h);
        float sc;
        inv = row_scale_pow2<16>(d, sc);
        split_frag2<8>(d, ys);
    }
    float4 da0[16];
    load_rowfrag<16>(da0, a0, row, D, L.h);  // holds a0 until the chunk's product arrives
    row_gemm128_h<2, true>(w2b, nullptr, ys, L, inv, [&](int c, f32x16 (&acc)[2]) {
        float4 y[8];
        acc_to_frag<2>(acc, y);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float4& a = da0[8 * c + k];
            a = make_float4(y[k].x * silu_g_(a.x), y[k].y * silu_g_(a.y), y[k].z * silu_g_(a.z), y[k].w * silu_g_(a.w));
        }
    });
    if (TRAIN && valid) store_rowfrag<16>(da0, t_da0, row, D, L.h);
    // dgeo[row][q] += sum_c da0[c] Wc[c][q] as ONE more MFMA tile: Wc^T padded to 32 rows is the A operand, the split
    // da0 row fragment the B operand, and the tile's first four rows -- registers 0..3 of the lanes with h = 0 -- are
    // dgeo[row][0..3]. (Until round 2 this was 64 float4 loads of Wc^T and 256 packed FMAs per lane plus a cross-lane
    // sum; that block returned sums with ONE product missing -- upper half-wave, second element of a packed pair --
    // for a few waves of every launch of more than 512 workgroups, differently from run to run:
    // tools/debug/dbg10k*.py, DESIGN.md section 7. The matrix-core form has no weight loads in VALU code at all.)
    Split2<8> ds;
    float inv2;
    {
        float sc;
        inv2 = row_scale_pow2<16>(da0, sc);
        split_frag2<8>(da0, ds);
    }
    {
        f32x16 g[1], gl[1];
        acc_zero<1>(g);
        acc_zero<1>(gl);
        WBlk2<1> wb[2];
        ld_blk2<1>(wb[0], wcp, L.lane, 0);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            if (kb + 1 < 8) ld_blk2<1>(wb[(kb + 1) & 1], wcp, (size_t)(kb + 1) * 64 + L.lane, 0);
            mfma3<1>(g, gl, wb[kb & 1], ds.h[kb], ds.l[kb]);
        }
        fold_low<1>(g, gl);
        if (valid && L.h == 0) {
            float4* dg = reinterpret_cast<float4*>(dgeo + row * 4);
            const float4 old = *dg;
            *dg = make_float4(fmaf(g[0][0], inv2, old.x), fmaf(g[0][1], inv2, old.y), fmaf(g[0][2], inv2, old.z),
                              fmaf(g[0][3], inv2, old.w));
        }
    }
    if (!FIRST) {
        row_gemm128_h<2, true>(w0cb, nullptr, ds, L, inv2, [&](int c, f32x16 (&acc)[2]) {
            float4 y[8], old[8];
            acc_to_frag<2>(acc, y);
            load_rowfrag<8>(old, dM + 64 * c, row, D, L.h);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                y[k].x += old[k].x; y[k].y += old[k].y; y[k].z += old[k].z; y[k].w += old[k].w;
            }
            store_tile64_lines(y, lds_tile, dM + 64 * c, row0, E, D, L);
        });
    }
}

// ---------------------------------------------------------------------------------
// host launchers (declared in model.h)
// ---------------------------------------------------------------------------------
static inline int grid_rows(int64_t rows) { return cdiv(rows, WG_ROWS); }
// The GEMMs of these kernels are split-operand products on the 16-bit matrix cores (f16x3: 2-way fp16 split, three
// MFMAs per K block; 1.7e-7 product error against fp64, tools/ubench). The fp32-MFMA and 3-way bf16 (bf16x6)
// generations of round 1 were removed in round 2; pet_config_set("trr", 0) selects the LDS-tile kernels of
// pet_fwd.hip / pet_bwd.hip, which are the one fallback (and the path of the LayerNorm / PostLN / residual variants).
// pet_config_set("trr_compress", bits): 1 compress (+adjoint), 2 edge head (+adjoint); 0 = the LDS-tile kernels
static int g_trr_tilek = 3;
void set_trr_compress(int v) { g_trr_tilek = v; }

// beta: the LayerNorm bias, nullptr = RMSNorm (here and below)
void trr_qkv(const float* X, const float* gamma, const float* beta, const Lin& qkv, float* QKV, int64_t R,
             hipStream_t st) {
    if (R <= 0) return;
    const size_t lds = SW_WBYTES + (size_t)SW_WAVES * 32 * TILE_LD * 4;
    const int g8 = (int)cdiv(R, SW_WAVES * WROWS);
    if (beta) { allow_big_lds(k_qkv_s<true>, lds); k_qkv_s<true><<<g8, 512, lds, st>>>(X, gamma, beta, w2_fwd(qkv), qkv.b, QKV, R); }
    else { allow_big_lds(k_qkv_s<false>, lds); k_qkv_s<false><<<g8, 512, lds, st>>>(X, gamma, beta, w2_fwd(qkv), qkv.b, QKV, R); }
}
void trr_qkv_bwd(const float* dQKV, const float* X, const float* gamma, bool layer_norm, const Lin& qkv,
                 const float* dX1, float* dXin, int64_t E, int64_t R, hipStream_t st) {
    if (R <= 0) return;
    if (layer_norm) k_qkv_bwd_h<true><<<grid_rows(R), 256, 0, st>>>(dQKV, X, gamma, w2_bwd(qkv), dX1, dXin, E, R);
    else k_qkv_bwd_h<false><<<grid_rows(R), 256, 0, st>>>(dQKV, X, gamma, w2_bwd(qkv), dX1, dXin, E, R);
}
void trr_oproj(const float* AO, const float* X, const Lin& out, float* X1, float* OC, int64_t E, int64_t R,
               hipStream_t st) {
    if (R <= 0) return;
    k_oproj_h<<<grid_rows(R), 256, 0, st>>>(AO, X, w2_fwd(out), out.b, X1, OC, E, R);
}
void trr_oproj_bwd(const float* dX1, const float* dOC, const Lin& out, float* dAO, int64_t E, int64_t R,
                   hipStream_t st) {
    if (R <= 0) return;
    k_oproj_bwd_h<<<grid_rows(R), 256, 0, st>>>(dX1, dOC, w2_bwd(out), dAO, E, R);
}
void trr_emlp(const float* X1, const float* gamma, const float* beta, const Lin& win, const Lin& wout, float* VG,
              float* X2, int64_t E, hipStream_t st) {
    if (E <= 0) return;
    if (emlp_s(X1, gamma, beta, win, wout, VG, X2, E, st)) return;  // large graphs: two waves per SIMD (pet_emlp_s.hip)
    const size_t lds = (size_t)4 * EP2_WAVE_LDS;
    if (beta) {
        allow_big_lds(k_emlp_p2<true>, lds);
        k_emlp_p2<true><<<grid_rows(E), 256, lds, st>>>(X1, gamma, beta, w2_fwd(win), win.b, w2_fwd(wout), wout.b, VG, X2, E);
    } else {
        allow_big_lds(k_emlp_p2<false>, lds);
        k_emlp_p2<false><<<grid_rows(E), 256, lds, st>>>(X1, gamma, beta, w2_fwd(win), win.b, w2_fwd(wout), wout.b, VG, X2, E);
    }
}
template <bool LN>
static void launch_emlp_bwd(const float* dY, const float* X1, const float* VG, const float* gamma, const float* beta,
                            const Lin& win, const Lin& wout, float* dX1, int64_t E, hipStream_t st, float* t_dvg, int ldy,
                            const float* dY2, const int* rev2) {
    const int grid = grid_rows(E);
    const size_t lds = (size_t)4 * 40960;  // per wave: dY tile / split planes 16 KB, VG chunks / X1 tile 16 KB, [dv | dg] 8 KB (all 160 KB of the CU)
    if (dY2 && !t_dvg) {
        allow_big_lds(k_emlp_bwd_p2<false, LN, true>, lds);
        k_emlp_bwd_p2<false, LN, true><<<grid, 256, lds, st>>>(dY, X1, VG, gamma, w2_bwd(wout), w2_bwd(win), dX1, E, nullptr, ldy, dY2, rev2);
    } else if (t_dvg) {
        allow_big_lds(k_emlp_bwd_p2<true, LN>, lds);
        k_emlp_bwd_p2<true, LN><<<grid, 256, lds, st>>>(dY, X1, VG, gamma, w2_bwd(wout), w2_bwd(win), dX1, E, t_dvg, ldy, dY2, rev2);
    } else {
        allow_big_lds(k_emlp_bwd_p2<false, LN>, lds);
        k_emlp_bwd_p2<false, LN><<<grid, 256, lds, st>>>(dY, X1, VG, gamma, w2_bwd(wout), w2_bwd(win), dX1, E, nullptr, ldy, dY2, rev2);
    }
}
void trr_emlp_bwd(const float* dY, const float* X1, const float* VG, const float* gamma, const float* beta,
                  const Lin& win, const Lin& wout, float* dX1, int64_t E, hipStream_t st, float* t_dvg, int ldy,
                  const float* dY2, const int* rev2) {
    if (E <= 0) return;
    if (beta) launch_emlp_bwd<true>(dY, X1, VG, gamma, beta, win, wout, dX1, E, st, t_dvg, ldy, dY2, rev2);
    else launch_emlp_bwd<false>(dY, X1, VG, gamma, beta, win, wout, dX1, E, st, t_dvg, ldy, dY2, rev2);
}

// ---------------------------------------------------------------------------------
// edge head (backend.py:171-217, 726-777) and its adjoint as TRR kernels on f16x3:
//   y = wl . SiLU(W2 SiLU(W0 x + b0) + b2) + bl;   yout = y * fc
// The adjoint recomputes a1 / a2 (as k_head_bwd does) and keeps them in registers.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_head_h(const float* __restrict__ Xin, W2 w0, const float* __restrict__ b0,
                                                    W2 w2, const float* __restrict__ b2, const float* __restrict__ wl,
                                                    float bl, const float* __restrict__ fc, float* __restrict__ ypred,
                                                    float* __restrict__ yout, int64_t R) {
    PET_TRR_ROWS_LDS();
    TRR_PROLOGUE(R);
    Split2<8> xs;
    float xinv;
    {   // backbone features are un-normalised rows: power-of-two row scale (see k_compress_h)
        float4 x[16];
        load_rows_lines128(x, lds_tile, Xin, row0, R, L);
        float sc;
        xinv = row_scale_pow2<16>(x, sc);
        split_frag2<8>(x, xs);
    }
    float4 s1[16];
    row_gemm128_h<2, true, 2>(w0, b0, xs, L, xinv, [&](int c, f32x16 (&acc)[2]) {
        float4 y[8];
        acc_to_frag<2>(acc, y);
#pragma unroll
        for (int k = 0; k < 8; k++) s1[8 * c + k] = make_float4(silu_(y[k].x), silu_(y[k].y), silu_(y[k].z), silu_(y[k].w));
    });
    {
        float sc;
        xinv = row_scale_pow2<16>(s1, sc);
        split_frag2<8>(s1, xs);
    }
    float part = 0.f;
    row_gemm128_h<2, true, 2>(w2, b2, xs, L, xinv, [&](int c, f32x16 (&acc)[2]) {
        float4 y[8];
        acc_to_frag<2>(acc, y);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 ww = *reinterpret_cast<const float4*>(wl + 64 * c + 8 * k + 4 * L.h);
            part += silu_(y[k].x) * ww.x + silu_(y[k].y) * ww.y + silu_(y[k].z) * ww.z + silu_(y[k].w) * ww.w;
        }
    });
    const float y = row_sum(part) + bl;
    if (valid && L.h == 0) {
        if (ypred) ypred[row] = y;
        yout[row] = fc ? y * fc[row] : y;
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(256) void k_head_bwd_h(const float* __restrict__ Xin, W2 w0f, const float* __restrict__ b0,
                                                     W2 w2f, const float* __restrict__ b2, W2 w0b, W2 w2b,
                                                     const float* __restrict__ wl, const float* __restrict__ gA,
                                                     const int* __restrict__ ctr, const float* __restrict__ fc,
                                                     const float* __restrict__ ypred, float* __restrict__ dfc,
                                                     float* __restrict__ dXout, int64_t R, float* __restrict__ t_s1,
                                                     float* __restrict__ t_da2, float* __restrict__ t_da1,
                                                     float* __restrict__ t_s2y) {
    PET_TRR_ROWS_LDS();
    TRR_PROLOGUE(R);
    float gy;  // dL/dy of this edge: the centre atom's seed times the cutoff factor
    {
        const float ga = gA[ctr[row]];
        gy = ga * fc[row];
        if (valid && L.h == 0) dfc[row] = ga * ypred[row];  // d(y fc)/dfc
    }
    Split2<8> xs;
    float xinv;
    {   // the same power-of-two row scales as the forward kernel (k_head_h)
        float4 x[16];
        load_rowfrag<16>(x, Xin, row, D, L.h);
        float sc;
        xinv = row_scale_pow2<16>(x, sc);
        split_frag2<8>(x, xs);
    }
    float4 a1[16], t[16];
    row_gemm128_h<2, true>(w0f, b0, xs, L, xinv, [&](int c, f32x16 (&acc)[2]) {
        float4 y[8];
        acc_to_frag<2>(acc, y);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a1[8 * c + k] = y[k];
            t[8 * c + k] = make_float4(silu_(y[k].x), silu_(y[k].y), silu_(y[k].z), silu_(y[k].w));
        }
    });
    if (TRAIN && valid) store_rowfrag<16>(t, t_s1, row, DH, L.h);
    {
        float sc;
        xinv = row_scale_pow2<16>(t, sc);
        split_frag2<8>(t, xs);
    }
    // a2 = W2 s1 + b2  ->  da2 = gy wl silu'(a2)   (t is reused for da2)
    row_gemm128_h<2, true>(w2f, b2, xs, L, xinv, [&](int c, f32x16 (&acc)[2]) {
        float4 y[8];
        acc_to_frag<2>(acc, y);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 ww = *reinterpret_cast<const float4*>(wl + 64 * c + 8 * k + 4 * L.h);
            if (TRAIN && valid)
                *reinterpret_cast<float4*>(t_s2y + row * DH + 64 * c + 8 * k + 4 * L.h) =
                    make_float4(gy * silu_(y[k].x), gy * silu_(y[k].y), gy * silu_(y[k].z), gy * silu_(y[k].w));
            t[8 * c + k] = make_float4(gy * ww.x * silu_g_(y[k].x), gy * ww.y * silu_g_(y[k].y),
                                       gy * ww.z * silu_g_(y[k].z), gy * ww.w * silu_g_(y[k].w));
        }
    });
    if (TRAIN && valid) store_rowfrag<16>(t, t_da2, row, DH, L.h);
    float inv;
    {
        float sc;
        inv = row_scale_pow2<16>(t, sc);
        split_frag2<8>(t, xs);
    }
    // ds1 = da2 W2  ->  da1 = ds1 silu'(a1)
    row_gemm128_h<2, true>(w2b, nullptr, xs, L, inv, [&](int c, f32x16 (&acc)[2]) {
        float4 y[8];
        acc_to_frag<2>(acc, y);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 a = a1[8 * c + k];
            t[8 * c + k] = make_float4(y[k].x * silu_g_(a.x), y[k].y * silu_g_(a.y), y[k].z * silu_g_(a.z), y[k].w * silu_g_(a.w));
        }
    });
    if (TRAIN && valid) store_rowfrag<16>(t, t_da1, row, DH, L.h);
    {
        float sc;
        inv = row_scale_pow2<16>(t, sc);
        split_frag2<8>(t, xs);
    }
    row_gemm128_h<2, true>(w0b, nullptr, xs, L, inv, [&](int c, f32x16 (&acc)[2]) {  // dx = da1 W0
        float4 y[8];
        acc_to_frag<2>(acc, y);
        store_tile64_lines(y, lds_tile, dXout + 64 * c, row0, R, D, L);
    });
}

bool trr_head_edge(const Model& m, const float* Xin, const float* fc, float* ypred, float* yout, int64_t E,
                   hipStream_t st) {
    if (!((g_trr_tilek & 2) && m.eh0.fwd2 && m.eh2.fwd2)) return false;
    if (head_edge_s(m, Xin, fc, ypred, yout, E, st)) return true;  // large graphs: two workgroups per CU, shared weight ring
    k_head_h<<<grid_rows(E), 256, 0, st>>>(Xin, w2_fwd(m.eh0), m.eh0.b, w2_fwd(m.eh2), m.eh2.b, m.ell_w, m.ell_b, fc, ypred,
                                           yout, E);
    return true;
}
bool trr_head_edge_bwd(const Model& m, const float* Xin, const float* gA, const int* ctr, const float* fc,
                       const float* ypred, float* dfc, float* dXout, int64_t E, float* t_s1, float* t_da2, float* t_da1,
                       float* t_s2y, hipStream_t st) {
    if (!((g_trr_tilek & 2) && m.eh0.fwd2 && m.eh2.fwd2 && m.eh0.bwd2 && m.eh2.bwd2)) return false;
    if (!t_s1 && head_edge_bwd_s(m, Xin, gA, ctr, fc, ypred, dfc, dXout, E, st)) return true;  // (inference; pet_head_s.hip)
    const int grid = grid_rows(E);
    if (t_s1)
        k_head_bwd_h<true><<<grid, 256, 0, st>>>(Xin, w2_fwd(m.eh0), m.eh0.b, w2_fwd(m.eh2), m.eh2.b, w2_bwd(m.eh0),
                                                 w2_bwd(m.eh2), m.ell_w, gA, ctr, fc, ypred, dfc, dXout, E, t_s1, t_da2,
                                                 t_da1, t_s2y);
    else
        k_head_bwd_h<false><<<grid, 256, 0, st>>>(Xin, w2_fwd(m.eh0), m.eh0.b, w2_fwd(m.eh2), m.eh2.b, w2_bwd(m.eh0),
                                                  w2_bwd(m.eh2), m.ell_w, gA, ctr, fc, ypred, dfc, dXout, E, nullptr, nullptr,
                                                  nullptr, nullptr);
    return true;
}

bool trr_compress(bool first, const Graph& g, const GnnLayerW& G, const float* Min, float* a0_out, float* Xout,
                  int64_t E, hipStream_t st) {
    if (!((g_trr_tilek & 1) && G.compress2.fwd2 && (first || G.compress0_msg.fwd2)) || E <= 0) return false;
    if (first)
        k_compress_h<true><<<grid_rows(E), 256, 0, st>>>(g.geo, g.sp_nbr, G.wc, G.tbl, nullptr, W2(), w2_fwd(G.compress2),
                                                       G.compress2.b, a0_out, Xout, E);
    else
        k_compress_h<false><<<grid_rows(E), 256, 0, st>>>(g.geo, g.sp_nbr, G.wc, G.tbl, Min, w2_fwd(G.compress0_msg),
                                                        w2_fwd(G.compress2), G.compress2.b, a0_out, Xout, E);
    return true;
}
bool trr_compress_bwd(bool first, const float* dXe, const float* a0, const GnnLayerW& G, float* dgeo, float* dM,
                      int64_t E, float* t_da0, hipStream_t st) {
    if (!((g_trr_tilek & 1) && G.compress2.bwd2 && G.wc2 && (first || G.compress0_msg.bwd2)) || E <= 0) return false;
    if (!t_da0 && compress_bwd_s(first, dXe, a0, G, dgeo, dM, E, st)) return true;  // (inference; pet_compress_s.hip)
    const int grid = grid_rows(E);
    if (first) {
        if (t_da0) k_compress_bwd_h<true, true><<<grid, 256, 0, st>>>(dXe, a0, w2_bwd(G.compress2), w2_wc(G), W2(), dgeo, nullptr, E, t_da0);
        else k_compress_bwd_h<true, false><<<grid, 256, 0, st>>>(dXe, a0, w2_bwd(G.compress2), w2_wc(G), W2(), dgeo, nullptr, E, nullptr);
    } else {
        if (t_da0) k_compress_bwd_h<false, true><<<grid, 256, 0, st>>>(dXe, a0, w2_bwd(G.compress2), w2_wc(G), w2_bwd(G.compress0_msg), dgeo, dM, E, t_da0);
        else k_compress_bwd_h<false, false><<<grid, 256, 0, st>>>(dXe, a0, w2_bwd(G.compress2), w2_wc(G), w2_bwd(G.compress0_msg), dgeo, dM, E, nullptr);
    }
    return true;
}

}  // namespace pet
