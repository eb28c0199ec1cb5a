// Shared device helpers of the per-atom fused attention kernels (pet_ablk.hip: forward and fused adjoint; pet_ablk2.hip: the
// two-kernel adjoint). See pet_ablk.hip for the layout conventions (token form / feature form, planes of 64 x).
#pragma once
#include "common.h"
#include "model.h"
#include "pet_ws.h"
#include "trr.h"

namespace pet {

#ifdef AB_ABL_NOMFMA  // timing ablation (results are wrong): the matrix products of this file become one multiply-add
__device__ __forceinline__ f32x16 ab_fake_mfma(const f16x8& a, const f16x8& b, f32x16 c) {
    c[0] += (float)a[0] * (float)b[0];
    return c;
}
#undef PET_MFMA_H
#define PET_MFMA_H(A, B, C) ab_fake_mfma((A), (B), (C))
#endif

constexpr float ABS = 64.0f;             // plane scale
constexpr float ABS_INV = 1.0f / 64.0f;
constexpr float ABQ = 4096.0f;           // accumulator scale = ABS^2
constexpr float ABQ_INV = 1.0f / 4096.0f;
constexpr float AB_LOG2E = 1.4426950408889634f;

union H8 {
    f16x8 v;
    h16x2 p[4];
};
// the two planes of eight values v = 64 x: hi = fp16(v), lo = fp16(v - hi); the low piece is derived from the PINNED high pair
// (trr.h split_pair_pinned: otherwise the compiler may convert twice with instructions that round differently)
__device__ __forceinline__ void ab_split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
    H8 a, b;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        h16x2 hp, lp;
        hp[0] = (_Float16)x[2 * j]; hp[1] = (_Float16)x[2 * j + 1];
        asm volatile("" : "+v"(hp));
        // fp16(x - hi) in ONE instruction per value: the mixed-precision fma reads the fp16 half directly (no v_cvt_f32_f16 +
        // v_sub) and writes its result as the low / high fp16 half of the destination (no v_cvt_pk_f16_f32 behind it: three
        // instructions per pair of values instead of four). x - hi is exact in fp32 (hi is x rounded to 11 bits), so the
        // single rounding to fp16 gives the bits the separate conversion gave.
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(x[2 * j]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(x[2 * j + 1]));
        a.p[j] = hp; b.p[j] = lp;
    }
    hi = a.v; lo = b.v;
}
// acc += 4096 (a b) from the planes of a (A operand) and b (B operand)
#define AB_MFMA3(acc, aH, aL, bH, bL)          \
    do {                                       \
        acc = PET_MFMA_H((aH), (bH), (acc));   \
        acc = PET_MFMA_H((aL), (bH), (acc));   \
        acc = PET_MFMA_H((aH), (bL), (acc));   \
    } while (0)

__device__ __forceinline__ f32x16 ab_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.f;
    return z;
}
// the eight registers 8 kb .. 8 kb + 7 of a C tile (K block kb of the NEXT product) scaled by f
__device__ __forceinline__ void ab_regs8(const f32x16& a, int kb, float f, float (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = a[8 * kb + j] * f;
}
__device__ __forceinline__ void ab_regs8(const f32x16& a, int kb, float (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = a[8 * kb + j];
}
// planes of the two K blocks of a C tile (times f)
__device__ __forceinline__ void ab_tile_planes(const f32x16& a, float f, f16x8 (&hi)[2], f16x8 (&lo)[2]) {
#pragma unroll
    for (int b = 0; b < 2; b++) {
        float t8[8];
        ab_regs8(a, b, f, t8);
        ab_split8(t8, hi[b], lo[b]);
    }
}
__device__ __forceinline__ void ab_tile_planes(const f32x16& a, f16x8 (&hi)[2], f16x8 (&lo)[2]) {
#pragma unroll
    for (int b = 0; b < 2; b++) {
        float t8[8];
        ab_regs8(a, b, t8);
        ab_split8(t8, hi[b], lo[b]);
    }
}

#ifdef AB_PROFILE
// debugging aid (build with -DAB_PROFILE): shader cycles per phase, summed over the waves of every launch
extern __device__ unsigned long long ab_prof[32];  // defined in pet_ablk.hip
#define AB_T(i)                                                                              \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                          \
        if (threadIdx.x % 64 == 0) atomicAdd(&ab_prof[(i)], t_ - ab_t0);                     \
        ab_t0 = __builtin_amdgcn_s_memtime();                                                \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#define AB_T0() unsigned long long ab_t0 = __builtin_amdgcn_s_memtime()
#else
#define AB_T(i)
#define AB_T0()
#endif

// A tile: 32 NQ token slots holding atom A (slot 0 = its centre token = row E + atom of the token stream, slots 1 .. TA - 1
// its neighbours = CSR rows startA ..) and, in a 32-slot tile, possibly a second atom B behind it (Graph::tile_desc; the
// graph build pairs small atoms with partners that fit). Tokens attend within their own atom only. Slots past the last
// token repeat it (their results are never stored and, as keys, are masked).
struct AbAtom {
    int atomA, startA, TA, atomB, startB, TB, T;
    int64_t E;
    __device__ __forceinline__ AbAtom(const int4* __restrict__ d, int64_t e) : E(e) {
        const int4 d0 = d[0], d1 = d[1];
        atomA = __builtin_amdgcn_readfirstlane(d0.x); startA = __builtin_amdgcn_readfirstlane(d0.y);
        TA = __builtin_amdgcn_readfirstlane(d0.z); atomB = __builtin_amdgcn_readfirstlane(d0.w);
        startB = __builtin_amdgcn_readfirstlane(d1.x); TB = __builtin_amdgcn_readfirstlane(d1.y);
        T = TA + TB;
    }
    __device__ __forceinline__ bool centre(int s) const { return s == 0 || s == TA; }
    __device__ __forceinline__ int atom(int s) const { return s < TA ? atomA : atomB; }
    __device__ __forceinline__ int64_t edge(int s) const {  // CSR row of a neighbour slot
        return s < TA ? (int64_t)startA + s - 1 : (int64_t)startB + (s - TA) - 1;
    }
    __device__ __forceinline__ int64_t row(int s) const {  // row of the token stream [edges | centre tokens]
        s = s < T ? s : T - 1;
        return centre(s) ? E + atom(s) : edge(s);
    }
};

// whole-row LDS-DMA of the atom's token rows into the wave's tile(s): tile tq holds slots 32 tq .. 32 tq + 31 in the
// layout of trr.h dma_tile128 (row r, 16-B piece p at byte 512 r + 16 (p ^ (r & 15)))
template <int NQ>
__device__ __forceinline__ void ab_dma_rows(const float* __restrict__ X, const AbAtom& a, unsigned lds_base,
                                            const RowLane& L) {
#pragma unroll
    for (int tq = 0; tq < NQ; tq++)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int r = 2 * j + (L.lane >> 5);
            const int p = (L.lane & 31) ^ (r & 15);
            glds16_trr(X + a.row(32 * tq + r) * D + 4 * p, lds_base + tq * 16384 + j * 1024);
        }
}

// planes of a normalised row tile in the wave's LDS: [kb 0..7][plane H, L][lane] f16x8
__device__ __forceinline__ void ab_park_planes(const float4 (&x)[16], char* tile, const RowLane& L) {
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
        const float v[8] = {x[2 * kb].x * ABS, x[2 * kb].y * ABS, x[2 * kb].z * ABS, x[2 * kb].w * ABS,
                            x[2 * kb + 1].x * ABS, x[2 * kb + 1].y * ABS, x[2 * kb + 1].z * ABS, x[2 * kb + 1].w * ABS};
        f16x8 h, l;
        ab_split8(v, h, l);
        *reinterpret_cast<f16x8*>(tile + ((kb * 2 + 0) * 64 + L.lane) * 16) = h;
        *reinterpret_cast<f16x8*>(tile + ((kb * 2 + 1) * 64 + L.lane) * 16) = l;
    }
}

// The normalised row WITHOUT the norm's weight and bias, xhat = (x - mean) rstd (RMSNorm: mean = 0), as planes of 64 xhat
// parked like ab_park_planes; returns rstd. For kernels whose Linear behind the norm has the norm's affine part folded in
// (abi.hip fold_norm_s): the planes are the operand of the product AND the xhat of the norm adjoint (ab_norm_adjoint_planes).
template <bool LN>
__device__ __forceinline__ float ab_park_xhat(float4 (&x)[16], char* tile, const RowLane& L) {
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
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    ab_park_planes(x, tile, L);
    return rstd;
}
// w (the adjoint w.r.t. xhat, a row fragment) -> the adjoint w.r.t. the norm's input, xhat read back from its parked planes:
//   rstd (w - xhat mean(xhat w))   (LayerNorm: minus its mean)
template <bool LN>
__device__ __forceinline__ void ab_norm_adjoint_planes(float4 (&w)[16], const char* tile, float rstd, const RowLane& L) {
    float4 xh[16];
    float dot = 0.f;
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(tile + ((kb * 2 + 0) * 64 + L.lane) * 16);
        const f16x8 l = *reinterpret_cast<const f16x8*>(tile + ((kb * 2 + 1) * 64 + L.lane) * 16);
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

// key bias (log2 of the cutoff factor, transformer.py:109-110) of the keys this lane's S^T registers hold; -inf masks
// the slots past the last token and the other atom of a paired tile (the lane is a QUERY: its atom decides)
template <int NQ>
__device__ __forceinline__ void ab_key_bias(float (&bias)[NQ][16], const AbAtom& a, const float* __restrict__ fc,
                                            const RowLane& L) {
    const bool qb = (L.r < a.T ? L.r : a.T - 1) >= a.TA;
#pragma unroll
    for (int tk = 0; tk < NQ; tk++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int key = 32 * tk + 8 * (i >> 2) + 4 * L.h + (i & 3);
            float b = -INFINITY;
            if (key < a.T && (key >= a.TA) == qb)
                b = a.centre(key) ? 0.f : __builtin_amdgcn_logf(fmaxf(fc[a.edge(key)], 1e-15f));
            bias[tk][i] = b;
        }
}

// accumulators of a token-form tile initialised with 4096 x bias (features 8 j + 4 h .. + 3 of the tile at b)
__device__ __forceinline__ void ab_bias_tile(f32x16& acc, const float* __restrict__ b, int h) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float4 v = *reinterpret_cast<const float4*>(b + 8 * j + 4 * h);
        acc[4 * j] = v.x * ABQ; acc[4 * j + 1] = v.y * ABQ; acc[4 * j + 2] = v.z * ABQ; acc[4 * j + 3] = v.w * ABQ;
    }
}

// one 1-KB fragment (64 lanes x 16 B) of a packed weight plane -> LDS by LDS-DMA
__device__ __forceinline__ void ab_dma_piece(const f16x8* plane, int idx, unsigned lane16, unsigned lds_dst) {
    const char* base = reinterpret_cast<const char*>(plane + (size_t)idx * 64);
    glds16_trr(reinterpret_cast<const float*>(base + lane16), lds_dst);
}

#ifdef AB_ABL_NOBAR  // timing ablation (results are wrong): no workgroup barrier at the stage boundaries
#define AB_STAGE_SYNC() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define AB_STAGE_SYNC()                                   \
    do {                                                  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  \
        __syncthreads();                                  \
    } while (0)
#endif

struct AbSel {
    f16x8 i0, i1;  // selection matrices of the two K blocks of a 32-wide tile, B-operand form
};
__device__ __forceinline__ AbSel ab_selectors(const RowLane& L, float one) {
    AbSel s;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int f = 8 * (j >> 2) + 4 * L.h + (j & 3);
        s.i0[j] = (f == L.r) ? (_Float16)one : (_Float16)0.0f;
        s.i1[j] = (16 + f == L.r) ? (_Float16)one : (_Float16)0.0f;
    }
    return s;
}
// planes (index = K block of the 32-wide tile) of a tile -> planes of its transpose (times the selector's entry), exactly
__device__ __forceinline__ void ab_transpose(const f16x8 (&h)[2], const f16x8 (&l)[2], const AbSel& sel,
                                             f16x8 (&th)[2], f16x8 (&tl)[2]) {
    f32x16 ch = ab_zero(), cl = ab_zero();
    ch = PET_MFMA_H(h[0], sel.i0, ch); cl = PET_MFMA_H(l[0], sel.i0, cl);
    ch = PET_MFMA_H(h[1], sel.i1, ch); cl = PET_MFMA_H(l[1], sel.i1, cl);
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            th[b][j] = (_Float16)ch[8 * b + j];
            tl[b][j] = (_Float16)cl[8 * b + j];
        }
}
// the same, also returning the sum over the registers of the transposed VALUES (h + l): column sums of the tile
__device__ __forceinline__ float ab_transpose_sum(const f16x8 (&h)[2], const f16x8 (&l)[2], const AbSel& sel,
                                                  f16x8 (&th)[2], f16x8 (&tl)[2]) {
    f32x16 ch = ab_zero(), cl = ab_zero();
    ch = PET_MFMA_H(h[0], sel.i0, ch); cl = PET_MFMA_H(l[0], sel.i0, cl);
    ch = PET_MFMA_H(h[1], sel.i1, ch); cl = PET_MFMA_H(l[1], sel.i1, cl);
    float sh = 0.f, sl = 0.f;
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            th[b][j] = (_Float16)ch[8 * b + j];
            tl[b][j] = (_Float16)cl[8 * b + j];
            sh += ch[8 * b + j];
            sl += cl[8 * b + j];
        }
    return sh + sl;
}

// k_ablk_bwd<1, LN> (pet_ablk_bwd1.hip)
void ablk_bwd1_launch(bool ln, const float* X, const float* dX1, const float* dOC, const float* gamma, const float* beta, W2 wqkv,
                      const float* bqkv, W2 wot, W2 wqkvt, const float* fc, const int4* desc, int n_list, int64_t E, float qscale,
                      float scale, float* dXin, float* dbias, hipStream_t st);

}  // namespace pet
