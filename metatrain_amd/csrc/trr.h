// Transposed, register-resident row-tile toolkit (gfx950, wave64, v_mfma_f32_32x32x16_f16 on split operands).
//
// One WAVE owns 32 rows and runs a whole chain of dense layers on them without LDS and
// without barriers. The product is computed transposed, Y^T = W X^T:
//   A operand = weights  W[n0 + (lane&31)][k]     (packed fragment order, streamed from L2)
//   B operand = rows     X[row = lane&31][k]      (registers)
//   C/D       = Y^T tile: column = lane&31 = ROW, row-in-tile = feature
// so lane (r = lane&31, h = lane>>5) holds, of row r, the features
//   n = 32 t + 8 q + 4 h + j      (t tile, q = reg>>2, j = reg&3)
// which is exactly the k-set  k = 8 kg + 4 h + j  (kg = 4 t + q)  that the same lane must
// supply as B operand of the NEXT layer. A "row fragment" of a [32 x K] activation is
// therefore K/8 float4 per lane, and the f32x16 accumulators of one GEMM *are* the row
// fragment of its output: activations never leave the register file inside a chain.
// Row statistics (RMSNorm, LayerNorm, dot-product heads) are a lane-local sum plus one
// xor-32 shuffle.
#pragma once
#include "common.h"

namespace pet {

struct RowLane {
    int lane, r, h;
    __device__ RowLane() {
        lane = threadIdx.x & 63;
        r = lane & 31;
        h = lane >> 5;
    }
};

// rows handled per wave / per 256-thread workgroup
constexpr int WROWS = 32;
constexpr int WG_ROWS = 128;

__device__ __forceinline__ int64_t wave_row0() {
    return ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * WROWS;  // wave-uniform (SGPR)
}

// x[kg] = X[row][8 kg + 4 h .. + 3]
template <int KG>
__device__ __forceinline__ void load_rowfrag(float4 (&x)[KG], const float* __restrict__ X, int64_t row, int ld,
                                             int h) {
    const float4* p = reinterpret_cast<const float4*>(X + row * ld + 4 * h);
#pragma unroll
    for (int kg = 0; kg < KG; kg++) x[kg] = p[2 * kg];
}

template <int KG>
__device__ __forceinline__ void store_rowfrag(const float4 (&y)[KG], float* __restrict__ Y, int64_t row, int ld,
                                              int h) {
    float4* p = reinterpret_cast<float4*>(Y + row * ld + 4 * h);
#pragma unroll
    for (int kg = 0; kg < KG; kg++) p[2 * kg] = y[kg];
}

// accumulator tile t <-> row-fragment entries 4t .. 4t+3
__device__ __forceinline__ float4 acc_q(const f32x16& a, int q) {
    return make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

template <int NT>
__device__ __forceinline__ void acc_to_frag(const f32x16 (&acc)[NT], float4* y) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) y[4 * t + q] = acc_q(acc[t], q);
}

template <int NT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
}

// acc[t] initialised with bias[n] at this lane's features; all loads are issued before the first
// use so they cost one L2 round trip, not one per float4
template <int NT>
__device__ __forceinline__ void acc_bias(f32x16 (&acc)[NT], const float* __restrict__ bias, int col0, int h) {
    float4 b[NT][4];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) b[t][q] = *reinterpret_cast<const float4*>(bias + col0 + 32 * t + 8 * q + 4 * h);
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc[t][4 * q] = b[t][q].x; acc[t][4 * q + 1] = b[t][q].y;
            acc[t][4 * q + 2] = b[t][q].z; acc[t][4 * q + 3] = b[t][q].w;
        }
}

// accumulator tiles initialised from a prefetched bias fragment (float4 per (tile, q))
template <int NT>
__device__ __forceinline__ void acc_from(f32x16 (&acc)[NT], const float4 (&b)[4 * NT]) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc[t][4 * q] = b[4 * t + q].x; acc[t][4 * q + 1] = b[4 * t + q].y;
            acc[t][4 * q + 2] = b[4 * t + q].z; acc[t][4 * q + 3] = b[4 * t + q].w;
        }
}
template <int NT>
__device__ __forceinline__ void ld_bias(float4 (&b)[4 * NT], const float* __restrict__ bias, int col0, int h) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) b[4 * t + q] = *reinterpret_cast<const float4*>(bias + col0 + 32 * t + 8 * q + 4 * h);
}

// ---------------------------------------------------------------------------------------------
// fp32 GEMM on the fp16 matrix cores ("f16x3"): a TWO-way split whose low piece is pre-scaled,
//   x = h + l,  h = fp16(x) (11 significant bits),  l' = fp16((x - h) * 2^11)   (the next 11 bits, stored at x's
//   own magnitude so that it never falls into fp16's subnormal range),
//   x w = h_x h_w + 2^-11 (h_x l'_w + l'_x h_w) + 2^-22 l'_x l'_w   (last term dropped: 2^-22 relative),
// i.e. THREE v_mfma_f32_32x32x16_f16 per K block on two accumulators (the high-high sum and the cross sum, combined
// as acc + 2^-11 acl at the end) and two weight planes (round 1's 3-way bf16 split, "bf16x6", needed six MFMAs on
// three planes and was removed in round 2). Measured against fp64 (tools/ubench/f16x3.hip): 1.7e-7 (bf16x6 3.7e-7,
// fp32 MFMA 4.5e-7). fp16's range: |x| up to 65504; elements below 6e-5 keep 6e-8 ABSOLUTE accuracy,
// which is what fp32 gives relative to an O(1) row; rows that are small as a whole (adjoints) are scaled by a
// power of two first (row_scale_pow2), exactly, and the result scaled back.
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct W2 {  // weight fragments, high and (scaled) low plane, [(tile * kb_total + kb) * 64 + lane]
    const f16x8 *h = nullptr, *l = nullptr;
};
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
// x - h with h read as the fp16 (low / high) half of a register: ONE mixed-precision fma instead of v_cvt_f32_f16 + v_sub
// (the difference is exact in fp32 either way, so the result is the same bit for bit)
__device__ __forceinline__ float sub_half_lo(float x, h16x2 hp) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x));
    return r;
}
__device__ __forceinline__ float sub_half_hi(float x, h16x2 hp) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x));
    return r;
}
__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * 2048.0f);
}
// Full-line stores of a [32 rows x 8 KG columns] output fragment. A row fragment gives each row 32 contiguous bytes
// per store instruction (its two lanes), i.e. four separate partial writes per 128-B line on the way to L2. Staged
// through a wave-private LDS tile instead, every store instruction writes four rows x 256 B (16 lanes per row): whole
// lines. lds: this wave's [32][TILE_LD] floats; rowptr(r) = where columns 0.. of tile row r go, or nullptr to skip it.
constexpr int TILE_LD = 68;
template <int KG, class RowPtr>
__device__ __forceinline__ void store_rows_lines(const float4 (&y)[KG], float* lds, const RowLane& L, RowPtr rowptr) {
    static_assert(KG % 8 == 0, "whole 64-column groups");
    float* wr = lds + L.r * TILE_LD + 4 * L.h;
    const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
#pragma unroll
    for (int g = 0; g < KG / 8; g++) {
#pragma unroll
        for (int kg = 0; kg < 8; kg++) *reinterpret_cast<float4*>(wr + 8 * kg) = y[8 * g + kg];
        __builtin_amdgcn_wave_barrier();  // same wave: the LDS serves its requests in order
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = 4 * j + rr;
            const float4 v = *reinterpret_cast<const float4*>(lds + r * TILE_LD + cc);
            float* p = rowptr(r);
            if (p) *reinterpret_cast<float4*>(p + 64 * g + cc) = v;
        }
        __builtin_amdgcn_wave_barrier();  // the tile is overwritten by the next group
    }
}
// The same with an addend that was requested in the line mapping of the stores (request_rows_addend): y + a leaves.
template <int KG, class RowPtr>
__device__ __forceinline__ void request_rows_addend(float4 (&a)[KG], const RowLane& L, RowPtr rowptr) {
    const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
#pragma unroll
    for (int g = 0; g < KG / 8; g++)
#pragma unroll
        for (int j = 0; j < 8; j++) a[8 * g + j] = *reinterpret_cast<const float4*>(rowptr(4 * j + rr) + 64 * g + cc);
}
template <int KG, class RowPtr>
__device__ __forceinline__ void store_rows_lines_add(const float4 (&y)[KG], const float4 (&a)[KG], float* lds, const RowLane& L,
                                                     RowPtr rowptr) {
    static_assert(KG % 8 == 0, "whole 64-column groups");
    float* wr = lds + L.r * TILE_LD + 4 * L.h;
    const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
#pragma unroll
    for (int g = 0; g < KG / 8; g++) {
#pragma unroll
        for (int kg = 0; kg < 8; kg++) *reinterpret_cast<float4*>(wr + 8 * kg) = y[8 * g + kg];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = 4 * j + rr;
            const float4 v = *reinterpret_cast<const float4*>(lds + r * TILE_LD + cc);
            const float4 b = a[8 * g + j];
            float* p = rowptr(r);
            if (p) *reinterpret_cast<float4*>(p + 64 * g + cc) = make_float4(b.x + v.x, b.y + v.y, b.z + v.z, b.w + v.w);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// rows row0 .. row0 + 31 of a row-major matrix Y (leading dimension ld), rows >= n_rows not written
__device__ __forceinline__ void store_tile64_lines(const float4 (&y)[8], float* lds, float* __restrict__ Y, int64_t row0,
                                                   int64_t n_rows, int ld, const RowLane& L) {
    store_rows_lines<8>(y, lds, L, [&](int r) { return row0 + r < n_rows ? Y + (row0 + r) * ld : nullptr; });
}
// Full-line loads of a [32 rows x 128 columns] row tile: 32 lanes read one row's 512 B (two rows per instruction), the
// fragments are then picked up from the wave-private LDS tile [32][ROWS_LD]. load_rowfrag touches 32 different lines
// per instruction (32 B of each); this touches 8. rowptr(r) = first of the 128 floats of tile row r (never null: clamp).
constexpr int ROWS_LD = 132;
// Two halves so that the global loads can be in flight across other work: request (coalesced mapping, 16 registers) ...
template <class RowPtr>
__device__ __forceinline__ void request_rows_lines(float4 (&t)[16], const RowLane& L, RowPtr rowptr) {
    const int rr = L.lane >> 5, cc = 4 * (L.lane & 31);
#pragma unroll
    for (int j = 0; j < 16; j++) t[j] = *reinterpret_cast<const float4*>(rowptr(2 * j + rr) + cc);
}
// ... and turn into row fragments through the LDS tile
__device__ __forceinline__ void rows_lines_to_frag(float4 (&x)[16], const float4 (&t)[16], float* lds, const RowLane& L) {
    const int rr = L.lane >> 5, cc = 4 * (L.lane & 31);
#pragma unroll
    for (int j = 0; j < 16; j++) *reinterpret_cast<float4*>(lds + (2 * j + rr) * ROWS_LD + cc) = t[j];
    __builtin_amdgcn_wave_barrier();
    const float* rd = lds + L.r * ROWS_LD + 4 * L.h;
#pragma unroll
    for (int kg = 0; kg < 16; kg++) x[kg] = *reinterpret_cast<const float4*>(rd + 8 * kg);
    __builtin_amdgcn_wave_barrier();
}
template <class RowPtr>
__device__ __forceinline__ void load_rows_lines(float4 (&x)[16], float* lds, const RowLane& L, RowPtr rowptr) {
    float4 t[16];
    request_rows_lines(t, L, rowptr);
    rows_lines_to_frag(x, t, lds, L);
}
// rows row0 .. row0 + 31 of a row-major [n_rows][128] matrix (rows past the end read the last row)
__device__ __forceinline__ void load_rows_lines128(float4 (&x)[16], float* lds, const float* __restrict__ X, int64_t row0,
                                                   int64_t n_rows, const RowLane& L) {
    load_rows_lines(x, lds, L, [&](int r) { return X + (row0 + r < n_rows ? row0 + r : n_rows - 1) * 128; });
}
// one LDS tile per wave for both directions: [32][ROWS_LD] floats (the store helpers use its first [32][TILE_LD])
#define PET_TRR_ROWS_LDS()                                                          \
    __shared__ __attribute__((aligned(16))) float trr_tiles_[4][32 * ROWS_LD];      \
    float* const lds_tile = trr_tiles_[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)]

// [32 rows x 32 columns] chunk as whole lines (8 lanes per row, 8 rows per instruction), in two halves like above;
// lds: [32][TILE32_LD] floats. rowptr(r) = first of the 32 floats of tile row r.
template <class RowPtr>
__device__ __forceinline__ void request_tile32_lines(float4 (&t)[4], const RowLane& L, RowPtr rowptr) {
    const int rr = L.lane >> 3, cc = 4 * (L.lane & 7);
#pragma unroll
    for (int j = 0; j < 4; j++) t[j] = *reinterpret_cast<const float4*>(rowptr(8 * j + rr) + cc);
}
__device__ __forceinline__ void tile32_lines_to_frag(float4 (&x)[4], const float4 (&t)[4], float* lds, const RowLane& L) {
    const int rr = L.lane >> 3, cc = 4 * (L.lane & 7);
#pragma unroll
    for (int j = 0; j < 4; j++) *reinterpret_cast<float4*>(lds + (8 * j + rr) * 36 + cc) = t[j];
    __builtin_amdgcn_wave_barrier();
    const float* rd = lds + L.r * 36 + 4 * L.h;
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = *reinterpret_cast<const float4*>(rd + 8 * q);
    __builtin_amdgcn_wave_barrier();
}

// The same for a [32 rows x 32 columns] tile (y[q] = columns 8 q + 4 h ..): 8 lanes per row, one full 128-B line each.
// lds: this wave's [32][TILE32_LD] floats.
constexpr int TILE32_LD = 36;
__device__ __forceinline__ void store_tile32_lines(const float4 (&y)[4], float* lds, float* __restrict__ Y, int64_t row0,
                                                   int64_t n_rows, int ld, const RowLane& L) {
    float* wr = lds + L.r * TILE32_LD + 4 * L.h;
#pragma unroll
    for (int q = 0; q < 4; q++) *reinterpret_cast<float4*>(wr + 8 * q) = y[q];
    __builtin_amdgcn_wave_barrier();
    const int rr = L.lane >> 3, cc = 4 * (L.lane & 7);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int r = 8 * j + rr;
        const float4 v = *reinterpret_cast<const float4*>(lds + r * TILE32_LD + cc);
        if (row0 + r < n_rows) *reinterpret_cast<float4*>(Y + (row0 + r) * ld + cc) = v;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int KB>
struct Split2 {
    f16x8 h[KB], l[KB];
};
template <int KB>
__device__ __forceinline__ void split_frag2(const float4* x, Split2<KB>& s) {
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        const float v[8] = {x[2 * kb].x, x[2 * kb].y, x[2 * kb].z, x[2 * kb].w,
                            x[2 * kb + 1].x, x[2 * kb + 1].y, x[2 * kb + 1].z, x[2 * kb + 1].w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            _Float16 a, b;
            split2(v[j], a, b);
            s.h[kb][j] = a; s.l[kb][j] = b;
        }
    }
}
#define PET_MFMA_H(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A), (B), (C), 0, 0, 0)
template <int NT>
struct WBlk2 {
    f16x8 h[NT], l[NT];
};
template <int NT>
__device__ __forceinline__ void ld_blk2(WBlk2<NT>& b, const W2& w, size_t i0, size_t tile_stride) {
#pragma unroll
    for (int t = 0; t < NT; t++) { b.h[t] = w.h[i0 + t * tile_stride]; b.l[t] = w.l[i0 + t * tile_stride]; }
}
// acc += W_h x_h ; acl += W_l' x_h + W_h x_l'
template <int NT>
__device__ __forceinline__ void mfma3(f32x16 (&acc)[NT], f32x16 (&acl)[NT], const WBlk2<NT>& b, const f16x8& xh,
                                      const f16x8& xl) {
#pragma unroll
    for (int t = 0; t < NT; t++) acl[t] = PET_MFMA_H(b.l[t], xh, acl[t]);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = PET_MFMA_H(b.h[t], xh, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; t++) acl[t] = PET_MFMA_H(b.h[t], xl, acl[t]);
}
// single-term training mode (pet_config_set("train_bf16", 1)): the high planes only
template <int NT>
struct WBlk1 {
    f16x8 h[NT];
};
template <int NT>
__device__ __forceinline__ void ld_blk1(WBlk1<NT>& b, const W2& w, size_t i0, size_t tile_stride) {
#pragma unroll
    for (int t = 0; t < NT; t++) b.h[t] = w.h[i0 + t * tile_stride];
}
template <int NT>
__device__ __forceinline__ void mfma1(f32x16 (&acc)[NT], const WBlk1<NT>& b, const f16x8& xh) {
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = PET_MFMA_H(b.h[t], xh, acc[t]);
}
template <int KB>
__device__ __forceinline__ void high_frag(const float4* x, f16x8 (&h)[KB]) {
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        const float v[8] = {x[2 * kb].x, x[2 * kb].y, x[2 * kb].z, x[2 * kb].w,
                            x[2 * kb + 1].x, x[2 * kb + 1].y, x[2 * kb + 1].z, x[2 * kb + 1].w};
#pragma unroll
        for (int j = 0; j < 8; j++) h[kb][j] = (_Float16)v[j];
    }
}
template <int NT>
__device__ __forceinline__ void fold_low(f32x16 (&acc)[NT], const f32x16 (&acl)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] += acl[t][r] * (1.0f / 2048.0f);
}

// Adjoint rows can be small as a whole (loss-dependent seeds): scale the row by the power of two that puts its
// largest element in [1, 2) before an f16x3 split, and the result back afterwards (both exact).
// Returns the inverse scale; `sc` receives the scale that was applied.
template <int KG>
__device__ __forceinline__ float row_pow2(const float4 (&x)[KG], float& sc) {  // the scale and its inverse, not applied
    float m = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; kg++)
        m = fmaxf(fmaxf(m, fmaxf(fabsf(x[kg].x), fabsf(x[kg].y))), fmaxf(fabsf(x[kg].z), fabsf(x[kg].w)));
    m = fmaxf(m, __shfl_xor(m, 32));
    int e = (__float_as_int(m) >> 23) & 0xff;
    e = e > 253 ? 253 : e;
    sc = __int_as_float((254 - e) << 23);  // 2^(127 - e)
    return __int_as_float(e << 23);        // 2^(e - 127); 0 for an all-zero row, whose outputs are 0 anyway
}
template <int KG>
__device__ __forceinline__ float row_scale_pow2(float4 (&x)[KG], float& sc) {
    const float inv = row_pow2<KG>(x, sc);
#pragma unroll
    for (int kg = 0; kg < KG; kg++) { x[kg].x *= sc; x[kg].y *= sc; x[kg].z *= sc; x[kg].w *= sc; }
    return inv;
}
template <int NT>
__device__ __forceinline__ void acc_scale(f32x16 (&acc)[NT], float f) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] *= f;
}

// sum over the row: lane-local + the partner lane holding the other half of the features
__device__ __forceinline__ float row_sum(float v) { return v + __shfl_xor(v, 32); }

// RMSNorm in place on a row fragment (torch.nn.RMSNorm, eps = finfo(float32).eps)
template <int KG>
__device__ __forceinline__ float rmsnorm_frag(float4 (&x)[KG], const float* __restrict__ gamma, int h) {
    float ss = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; kg++) ss += x[kg].x * x[kg].x + x[kg].y * x[kg].y + x[kg].z * x[kg].z + x[kg].w * x[kg].w;
    ss = row_sum(ss);
    const float rstd = rsqrtf(ss * (1.0f / (8 * KG)) + 1.1920928955078125e-07f);
#pragma unroll
    for (int kg = 0; kg < KG; kg++) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + 8 * kg + 4 * h);
        x[kg].x *= rstd * g.x; x[kg].y *= rstd * g.y; x[kg].z *= rstd * g.z; x[kg].w *= rstd * g.w;
    }
    return rstd;
}

// RMSNorm adjoint: given w = gamma * dn (fragment) and the un-normalised input x,
// dx = rstd * w - x * rstd^3 * mean(w x)
template <int KG>
__device__ __forceinline__ void rmsnorm_bwd_frag(float4 (&w)[KG], const float4 (&x)[KG]) {
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; kg++) {
        ss += x[kg].x * x[kg].x + x[kg].y * x[kg].y + x[kg].z * x[kg].z + x[kg].w * x[kg].w;
        dot += x[kg].x * w[kg].x + x[kg].y * w[kg].y + x[kg].z * w[kg].z + x[kg].w * w[kg].w;
    }
    ss = row_sum(ss);
    dot = row_sum(dot);
    const float rstd = rsqrtf(ss * (1.0f / (8 * KG)) + 1.1920928955078125e-07f);
    const float coef = dot * rstd * rstd * rstd * (1.0f / (8 * KG));
#pragma unroll
    for (int kg = 0; kg < KG; kg++) {
        w[kg].x = rstd * w[kg].x - x[kg].x * coef; w[kg].y = rstd * w[kg].y - x[kg].y * coef;
        w[kg].z = rstd * w[kg].z - x[kg].z * coef; w[kg].w = rstd * w[kg].w - x[kg].w * coef;
    }
}

// The norm of a transformer layer as a template switch: LN = false is torch.nn.RMSNorm (above), LN = true is
// torch.nn.LayerNorm (eps 1e-5, weight + bias; transformer.py:170-176): the RMS formulas on the centred row.
template <int KG, bool LN>
__device__ __forceinline__ void norm_frag(float4 (&x)[KG], const float* __restrict__ gamma, const float* __restrict__ beta,
                                          int h) {
    if (!LN) {
        rmsnorm_frag<KG>(x, gamma, h);
        return;
    }
    float sm = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; kg++) sm += (x[kg].x + x[kg].y) + (x[kg].z + x[kg].w);
    const float mean = row_sum(sm) * (1.0f / (8 * KG));
    float ss = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; kg++) {
        x[kg].x -= mean; x[kg].y -= mean; x[kg].z -= mean; x[kg].w -= mean;
        ss += x[kg].x * x[kg].x + x[kg].y * x[kg].y + x[kg].z * x[kg].z + x[kg].w * x[kg].w;
    }
    ss = row_sum(ss);
    const float rstd = rsqrtf(ss * (1.0f / (8 * KG)) + 1e-5f);
#pragma unroll
    for (int kg = 0; kg < KG; kg++) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + 8 * kg + 4 * h);
        const float4 b = *reinterpret_cast<const float4*>(beta + 8 * kg + 4 * h);
        x[kg].x = x[kg].x * (rstd * g.x) + b.x; x[kg].y = x[kg].y * (rstd * g.y) + b.y;
        x[kg].z = x[kg].z * (rstd * g.z) + b.z; x[kg].w = x[kg].w * (rstd * g.w) + b.w;
    }
}
// adjoint: w = gamma * dn on entry, dx on exit; x (the un-normalised input) is centred in place when LN
template <int KG, bool LN>
__device__ __forceinline__ void norm_bwd_frag(float4 (&w)[KG], float4 (&x)[KG]) {
    if (!LN) {
        rmsnorm_bwd_frag<KG>(w, x);
        return;
    }
    float sm = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; kg++) sm += (x[kg].x + x[kg].y) + (x[kg].z + x[kg].w);
    const float mean = row_sum(sm) * (1.0f / (8 * KG));
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; kg++) {
        x[kg].x -= mean; x[kg].y -= mean; x[kg].z -= mean; x[kg].w -= mean;
        ss += x[kg].x * x[kg].x + x[kg].y * x[kg].y + x[kg].z * x[kg].z + x[kg].w * x[kg].w;
        dot += x[kg].x * w[kg].x + x[kg].y * w[kg].y + x[kg].z * w[kg].z + x[kg].w * w[kg].w;
    }
    ss = row_sum(ss);
    dot = row_sum(dot);
    const float rstd = rsqrtf(ss * (1.0f / (8 * KG)) + 1e-5f);
    const float coef = dot * rstd * rstd * rstd * (1.0f / (8 * KG));
    float sw = 0.f;
#pragma unroll
    for (int kg = 0; kg < KG; kg++) {
        w[kg].x = rstd * w[kg].x - x[kg].x * coef; w[kg].y = rstd * w[kg].y - x[kg].y * coef;
        w[kg].z = rstd * w[kg].z - x[kg].z * coef; w[kg].w = rstd * w[kg].w - x[kg].w * coef;
        sw += (w[kg].x + w[kg].y) + (w[kg].z + w[kg].w);
    }
    const float mw = row_sum(sw) * (1.0f / (8 * KG));
#pragma unroll
    for (int kg = 0; kg < KG; kg++) { w[kg].x -= mw; w[kg].y -= mw; w[kg].z -= mw; w[kg].w -= mw; }
}

// sigmoid on the hardware transcendentals: v_exp_f32 (2^x, ~1 ulp) and v_rcp_f32 (~1 ulp). The
// argument scaling costs a relative error of ~1e-7 |x|, far inside the 1e-5 parity budget, and
// saves ~15 VALU instructions per element over expf() + IEEE division in the gate-heavy stages.
__device__ __forceinline__ float sigm_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float silu_(float x) { return x * sigm_(x); }
__device__ __forceinline__ float silu_g_(float x) {
    float s = sigm_(x);
    return s * (1.0f + x * (1.0f - s));
}

// ---------------------------------------------------------------------------------------------
// Pieces of the software-pipelined row kernels (k_emlp_p2 / k_emlp_bwd_p2 in pet_trr.hip, k_comb_p2 in pet_comb.hip)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float f4c(const float4& v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
// f16x3 split of two values into packed (high, pre-scaled low) pairs, computed where it stands in the instruction
// stream (the pipelined kernels place it in a particular slot). The low piece is derived from the PINNED high pair:
// pinning the finished pair alone let the compiler convert x -> fp16 twice, once for the pair (v_cvt_pk_f16_f32) and once
// for the remainder (v_cvt_f16_f32), and the two instructions do not round every input alike (a handful of values in
// 3e5 came out one fp16 ulp apart: rows off by 1e-4; tools/ubench/emlp_fwd_ab.hip).
__device__ __forceinline__ void split_pair_pinned(float x0, float x1, h16x2& hp, h16x2& lp) {
    hp[0] = (_Float16)x0; hp[1] = (_Float16)x1;
    asm volatile("" : "+v"(hp));
    lp[0] = (_Float16)(sub_half_lo(x0, hp) * 2048.0f);
    lp[1] = (_Float16)(sub_half_hi(x1, hp) * 2048.0f);
    asm volatile("" : "+v"(lp));
}
// LDS-DMA: 16 B per lane from global memory straight into LDS at lds_dst + 16 lane (no registers; counted by vmcnt,
// invisible to the compiler's own wait bookkeeping)
__device__ __forceinline__ void glds16_trr(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// whole-row LDS-DMA of a [32 x 128] fp32 tile: instruction j brings rows 2j, 2j + 1; row r, 16-B piece p lands at
// byte 512 r + 16 (p ^ (r & 15)) of the tile
__device__ __forceinline__ void dma_tile128(const float* __restrict__ X, int64_t row0, int64_t n_rows, unsigned lds_base,
                                            const RowLane& L, int ld = 128) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int r = 2 * j + (L.lane >> 5);
        int64_t rr = row0 + r;
        rr = rr < n_rows ? rr : n_rows - 1;
        const int p = (L.lane & 31) ^ (r & 15);
        glds16_trr(X + rr * ld + 4 * p, lds_base + j * 1024);
    }
}
// row fragment (trr.h) out of such a tile
__device__ __forceinline__ void tile128_to_frag(float4 (&x)[16], const char* tile, const RowLane& L) {
    const char* rowp = tile + 512 * L.r;
    const int sw = L.r & 15;
#pragma unroll
    for (int kg = 0; kg < 16; kg++) x[kg] = *reinterpret_cast<const float4*>(rowp + 16 * ((2 * kg + L.h) ^ sw));
}

}  // namespace pet
