// Weight-gradient GEMM for the training row: for y = x W^T + b,
//   dW[n][k] = sum_rows dY[row][n] * X[row][k],   db[n] = sum_rows dY[row][n].
// M = n_out, N = k_in are small (<= 1024 x 512), the reduction runs over 10^5..10^6 rows, so the
// rows are split over workgroups (split-K); every workgroup accumulates a [128 x KB] block in fp32
// MFMA registers over its row range and writes ONE partial; a second kernel sums the partials in a
// fixed order (deterministic, no float atomics).
//
// Operand layout: v_mfma_f32_32x32x2_f32 with the row index as the MFMA k dimension,
//   A[i = n][kk = row] = dY tile, B[kk = row][j = k] = X tile, both staged in LDS as [32 rows][cols]
// so that a lane reads 32 consecutive floats of one row (conflict-free ds_read_b32).
// X rows are produced by a prologue functor (identity / RMSNorm-hat / SwiGLU / SiLU / LayerNorm-hat),
// i.e. the forward activation is rebuilt from what the forward pass saved; dY rows come from the
// reverse pass' own buffers.
#pragma once
#include "common.h"
#include "tile.h"

namespace pet {

constexpr int WG_RB = 32;  // rows per staging block

// ---- row sources ----------------------------------------------------------------------------
// Each provides: static constexpr int COLS; fill(float* dst /*[32][COLS+4]*/, row0, n_rows): cooperative
struct SrcPlain {  // rows of a row-major buffer, columns [col0, col0 + COLS)
    const float* p; int ld; int col0;
};
struct SrcSplit {  // rows < e from a, rows >= e from b (token streams: edges then centres)
    const float* a; const float* b; int64_t e; int ld;
};

template <int COLS>
__device__ __forceinline__ void fill_plain(float* dst, const float* __restrict__ p, int ld, int col0, int64_t row0,
                                           int64_t n_rows) {
    constexpr int C4 = COLS / 4, LDS = COLS + 4;
    for (int idx = threadIdx.x; idx < WG_RB * C4; idx += NTHREADS) {
        const int r = idx / C4, c = idx % C4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < n_rows) v = *reinterpret_cast<const float4*>(p + (row0 + r) * ld + col0 + 4 * c);
        *reinterpret_cast<float4*>(dst + r * LDS + 4 * c) = v;
    }
}

// in-place transforms on a staged [32][COLS+4] tile; 8 threads per row
template <int COLS>
__device__ __forceinline__ void tile_rms_hat(float* t) {  // x -> x * rsqrt(mean x^2 + eps)
    constexpr int LDS = COLS + 4;
    const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
    float ss = 0.f;
    for (int c = q * 4; c < COLS; c += 32) {
        const float4 v = *reinterpret_cast<float4*>(t + r * LDS + c);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
    const float rstd = rsqrtf(ss * (1.0f / COLS) + 1.1920928955078125e-07f);
    for (int c = q * 4; c < COLS; c += 32) {
        float4 v = *reinterpret_cast<float4*>(t + r * LDS + c);
        v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
        *reinterpret_cast<float4*>(t + r * LDS + c) = v;
    }
}

template <int COLS>
__device__ __forceinline__ void tile_ln_hat(float* t) {  // x -> (x - mean) * rsqrt(var + 1e-5)
    constexpr int LDS = COLS + 4;
    const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
    float sm = 0.f;
    for (int c = q * 4; c < COLS; c += 32) {
        const float4 v = *reinterpret_cast<float4*>(t + r * LDS + c);
        sm += (v.x + v.y) + (v.z + v.w);
    }
    sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 4);
    const float mean = sm * (1.0f / COLS);
    float ss = 0.f;
    for (int c = q * 4; c < COLS; c += 32) {
        const float4 v = *reinterpret_cast<float4*>(t + r * LDS + c);
        const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
        ss += a * a + b * b + cc * cc + d * d;
    }
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
    const float rstd = rsqrtf(ss * (1.0f / COLS) + 1e-5f);
    for (int c = q * 4; c < COLS; c += 32) {
        float4 v = *reinterpret_cast<float4*>(t + r * LDS + c);
        v.x = (v.x - mean) * rstd; v.y = (v.y - mean) * rstd; v.z = (v.z - mean) * rstd; v.w = (v.w - mean) * rstd;
        *reinterpret_cast<float4*>(t + r * LDS + c) = v;
    }
}

// ---- the kernel -----------------------------------------------------------------------------
// grid = (n_blocks of 128 output rows, splits). XMODE: 0 plain, 1 rms-hat, 2 swiglu(VG: v|g), 3 silu,
// 4 layernorm-hat of [x ; x[rev]] (COLS = 256), 5 layernorm-hat of the row itself (statistics recomputed).
// YMODE: 0 plain, 1 split (edges | centres).
struct WgradArgs {
    const float* y0; const float* y1; int64_t y_split; int y_ld; int y_col0;
    const float* x0; int x_ld; int x_col0; int x_hid;  // x_hid: SwiGLU hidden size (gate at +x_hid)
    const int* rev; const float* lns;                  // XMODE 4
    int64_t n_rows;
    float* partial;  // [splits][n_out][KB]  (this launch's k-block only)
    float* partial_b; // [splits][n_out] or null
    int n_out;
};

template <int KB, int XMODE>
__global__ __launch_bounds__(NTHREADS) void k_wgrad(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDY = 128 + 4, LDX = KB + 4, KT = KB / 32;
    float* Ys = smem;                 // [32][132]
    float* Xs = smem + WG_RB * LDY;   // [32][KB+4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int64_t rows_per = ((a.n_rows + nsplit - 1) / nsplit + WG_RB - 1) / WG_RB * WG_RB;
    const int64_t r_begin = (int64_t)split * rows_per;
    const int64_t r_end = min(a.n_rows, r_begin + rows_per);
    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    float bsum = 0.f;  // thread c < 128: column sum of dY (bias gradient)
    // Register prefetch: the global loads of block i + 1 are issued before the MFMAs of block i (the kernel is
    // otherwise load -> barrier -> MFMA with nothing in flight during the MFMAs). Per thread: 4 float4 of dY and
    // KB / 32 float4 of X (twice that for the SwiGLU source, which reads v and g). XMODE 4 gathers through `rev`
    // and keeps the direct path.
    constexpr int YQ = WG_RB * 32 / NTHREADS;             // 4
    constexpr int XQ = WG_RB * (KB / 4) / NTHREADS;       // KB / 32
    float4 ypre[YQ], xpre[XMODE == 4 ? 1 : XQ], gpre[XMODE == 2 ? XQ : 1];
    auto fetch = [&](int64_t row0) {
#pragma unroll
        for (int q = 0; q < YQ; q++) {
            const int idx = threadIdx.x + q * NTHREADS, r = idx >> 5, c = idx & 31;
            const int64_t row = row0 + r;
            ypre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < r_end) {
                const float* src = (a.y1 && row >= a.y_split) ? a.y1 + (row - a.y_split) * a.y_ld
                                                              : a.y0 + row * a.y_ld;
                ypre[q] = *reinterpret_cast<const float4*>(src + a.y_col0 + 128 * nb + 4 * c);
            }
        }
        if (XMODE != 4) {
#pragma unroll
            for (int q = 0; q < XQ; q++) {
                const int idx = threadIdx.x + q * NTHREADS, r = idx / (KB / 4), c = idx % (KB / 4);
                xpre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (XMODE == 2) gpre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row0 + r < r_end) {
                    const float* base = a.x0 + (row0 + r) * a.x_ld + a.x_col0 + 4 * c;
                    xpre[q] = *reinterpret_cast<const float4*>(base);
                    if (XMODE == 2) gpre[q] = *reinterpret_cast<const float4*>(base + a.x_hid);
                }
            }
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (int64_t row0 = r_begin; row0 < r_end; row0 += WG_RB) {
        __syncthreads();
        // ---- stage dY block: columns [128 nb, 128 nb + 128)
#pragma unroll
        for (int q = 0; q < YQ; q++) {
            const int idx = threadIdx.x + q * NTHREADS, r = idx >> 5, c = idx & 31;
            *reinterpret_cast<float4*>(Ys + r * LDY + 4 * c) = ypre[q];
        }
        // ---- stage X block
        if (XMODE == 4) {  // [x ; x[rev]] then LayerNorm-hat with the saved (mean, rstd)
            for (int idx = threadIdx.x; idx < WG_RB * 64; idx += NTHREADS) {
                const int r = idx >> 6, c = idx & 63;
                const int64_t row = row0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < r_end) {
                    const int64_t src = c < 32 ? row : (int64_t)a.rev[row];
                    v = *reinterpret_cast<const float4*>(a.x0 + src * a.x_ld + 4 * (c & 31));
                    const float mean = a.lns[2 * row], rstd = a.lns[2 * row + 1];
                    v.x = (v.x - mean) * rstd; v.y = (v.y - mean) * rstd; v.z = (v.z - mean) * rstd; v.w = (v.w - mean) * rstd;
                }
                *reinterpret_cast<float4*>(Xs + r * LDX + 4 * c) = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < XQ; q++) {
                const int idx = threadIdx.x + q * NTHREADS, r = idx / (KB / 4), c = idx % (KB / 4);
                float4 v = xpre[q];
                if (XMODE == 2) {
                    const float4 g = gpre[q];
                    v = make_float4(v.x * sigmoidf_(g.x), v.y * sigmoidf_(g.y), v.z * sigmoidf_(g.z), v.w * sigmoidf_(g.w));
                }
                if (XMODE == 3) { v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w); }
                *reinterpret_cast<float4*>(Xs + r * LDX + 4 * c) = v;
            }
        }
        if (row0 + WG_RB < r_end) fetch(row0 + WG_RB);
        __syncthreads();
        if (XMODE == 1) {
            tile_rms_hat<KB>(Xs);
            __syncthreads();
        }
        if (XMODE == 5) {
            tile_ln_hat<KB>(Xs);
            __syncthreads();
        }
        if (a.partial_b && threadIdx.x < 128) {
#pragma unroll 8
            for (int r = 0; r < WG_RB; r++) bsum += Ys[r * LDY + threadIdx.x];
        }
        // ---- MFMA: wave w owns output rows 32 w .. 32 w + 31 of this 128-row block, all KB columns
#pragma unroll 4
        for (int s = 0; s < WG_RB / 2; s++) {
            const int rr = 2 * s + (lane >> 5);
            const float av = Ys[rr * LDY + 32 * wave + (lane & 31)];
#pragma unroll
            for (int t = 0; t < KT; t++) {
                const float bv = Xs[rr * LDX + 32 * t + (lane & 31)];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
    // ---- write this split's partial: out[n = 128 nb + 32 wave + acc_row][k = 32 t + lane&31]
    float* P = a.partial + ((size_t)split * a.n_out + 128 * nb + 32 * wave) * KB;
#pragma unroll
    for (int t = 0; t < KT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) P[(size_t)acc_row(r, lane) * KB + 32 * t + (lane & 31)] = acc[t][r];
    if (a.partial_b && threadIdx.x < 128) a.partial_b[(size_t)split * a.n_out + 128 * nb + threadIdx.x] = bsum;
}

// ---- the same GEMM on the 16-bit matrix cores (bf16x3) ------------------------------------------
// 54 % of the fp32-MFMA bound was all k_wgrad reached (77 TFLOP/s; its operands move at 1.6 TB/s), and the bound itself
// sits below what HBM delivers, so the reduction runs as a split-operand product: every fp32 value is split three ways,
//   x = h + m + l   (8 + 8 + 8 significand bits, each piece a bf16: fp32's exponent range, so the adjoint rows dY need no
//   scaling whatever their magnitude -- which is why this is bf16 and not the f16x3 of the forward stages),
// and a product keeps the six terms hh + hm + mh + hl + lh + mm (the dropped ones are below 2^-24 relative; measured
// against fp64: 3.7e-7). v_mfma_f32_32x32x16_bf16 contracts 16 ROWS per instruction: six of them per 16 rows and 32 x 32
// tile cost 192 cycles against 512 for the eight fp32 MFMAs they replace.
// LDS layout: three planes per operand, column-major -- [column][LDR = 40] bf16, the 32 rows of the block contiguous
// (+ 8 pad) -- because the MFMA wants 8 consecutive ROWS of one column per lane (one ds_read_b128; 80 B between
// consecutive columns = 5 x 16 B: the 16 lanes of a read phase cover all 64 banks). A thread stages TWO consecutive rows
// of four columns, so each (column, plane) is one packed 32-bit store; the row-pair slot is rotated by 4 x ((column >> 4)
// & 3) inside its column, which keeps the 16-byte read groups intact and turns the stores' 8-way bank conflict (column
// stride 4 x 80 B) into a 2-way one (free for ds_write_b32).
// Workgroups are numbered so that the n_out / 128 blocks that read the same X rows sit on the same XCD (ids 8 apart),
// i.e. the re-read of X comes from that XCD's L2.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int WB_LDW = 20;  // 32-bit words per staged column (40 bf16)

__device__ __forceinline__ unsigned bf16_bits(float x) {
    const __bf16 v = (__bf16)x;
    return (unsigned)__builtin_bit_cast(unsigned short, v);
}
// pieces of a (row 2p) and b (row 2p + 1) of one column, packed per plane: low half = the even row
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned ha = bf16_bits(a), hb = bf16_bits(b);
    const float ra = a - __uint_as_float(ha << 16), rb = b - __uint_as_float(hb << 16);
    const unsigned ma = bf16_bits(ra), mb = bf16_bits(rb);
    const float sa = ra - __uint_as_float(ma << 16), sb = rb - __uint_as_float(mb << 16);
    h = ha | (hb << 16);
    m = ma | (mb << 16);
    l = bf16_bits(sa) | (bf16_bits(sb) << 16);
}
__device__ __forceinline__ int wb_slot(int c, int pp) { return c * WB_LDW + ((pp + 4 * ((c >> 4) & 3)) & 15); }
__device__ __forceinline__ void wb_store4(unsigned* plane_h, unsigned* plane_m, unsigned* plane_l, int c0, int pp,
                                          const float4& a, const float4& b) {
    const float va[4] = {a.x, a.y, a.z, a.w}, vb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        unsigned h, m, l;
        split3_pair(va[e], vb[e], h, m, l);
        const int w = wb_slot(c0 + e, pp);
        plane_h[w] = h; plane_m[w] = m; plane_l[w] = l;
    }
}
__device__ __forceinline__ void wb_store4_high(unsigned* plane_h, int c0, int pp, const float4& a, const float4& b) {
    const float va[4] = {a.x, a.y, a.z, a.w}, vb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 4; e++) plane_h[wb_slot(c0 + e, pp)] = bf16_bits(va[e]) | (bf16_bits(vb[e]) << 16);
}
// eight rows of one column: the words are read as what they were stored as (no type punning across the barrier)
__device__ __forceinline__ bf16x8 wb_load8(const unsigned* p) {
    const uint4 w = *reinterpret_cast<const uint4*>(p);
    return __builtin_bit_cast(bf16x8, w);
}
#define PET_MFMA_BF(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16((A), (B), (C), 0, 0, 0)

// grid = n_blocks * splits workgroups (1-D, XCD-aware numbering when splits % 8 == 0). 128 x-columns per launch
// (x_col0 selects them). XMODE as in k_wgrad; XMODE 1 needs the whole row in the launch (k_in == 128).
// ONE (pet_config_set("train_bf16", 1)): the high bf16 pieces only -- one plane per operand, one MFMA per tile and 16 rows
template <int XMODE, bool ONE = false>
__global__ __launch_bounds__(NTHREADS, 2) void k_wgrad_b(WgradArgs a, int nb_total, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned wsm[];
    constexpr int KB = 128, KT = 4, PLANE = 128 * WB_LDW;  // words per plane
    constexpr int NP = ONE ? 1 : 3;
    unsigned* Yp = wsm;               // [NP][128 columns][20 words]
    unsigned* Xp = wsm + NP * PLANE;  // [NP][128 columns][20 words]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int nb, split;
    if ((nsplit & 7) == 0) {  // ids L, L + 8, ... share an XCD: the nb_total readers of one row range are neighbours there
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        nb = slot % nb_total;
        split = (slot / nb_total) * 8 + xcd;
    } else {
        nb = blockIdx.x % nb_total;
        split = blockIdx.x / nb_total;
    }
    const int64_t rows_per = ((a.n_rows + nsplit - 1) / nsplit + WG_RB - 1) / WG_RB * WG_RB;
    const int64_t r_begin = (int64_t)split * rows_per;
    const int64_t r_end = min(a.n_rows, r_begin + rows_per);
    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    // staging map: thread = (column quad c4 = tid & 31, row-pair group rg = tid >> 5); load q covers row 2 pp + (q & 1)
    // of pair pp = rg + 8 (q >> 1): the 32 lanes of a half wave read one 512-byte row
    const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};  // bias gradient: this thread's four dY columns over the rows it stages
    float4 ypre[4], xpre[4], gpre[XMODE == 2 ? 4 : 1];
    float2 lpre[XMODE == 4 ? 4 : 1];  // XMODE 4: the saved LayerNorm (mean, rstd) of the row
    auto fetch = [&](int64_t row0) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t row = row0 + 2 * (rg + 8 * (q >> 1)) + (q & 1);
            ypre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            xpre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (XMODE == 2) gpre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (XMODE == 4) lpre[q] = make_float2(0.f, 0.f);
            if (row < r_end) {
                const float* ys = (a.y1 && row >= a.y_split) ? a.y1 + (row - a.y_split) * a.y_ld : a.y0 + row * a.y_ld;
                ypre[q] = *reinterpret_cast<const float4*>(ys + a.y_col0 + 128 * nb + 4 * c4);
                if (XMODE == 4) {  // [x ; x[rev]]: columns below 128 from the row itself, the rest from its reverse edge
                    const int64_t src = a.x_col0 < 128 ? row : (int64_t)a.rev[row];
                    xpre[q] = *reinterpret_cast<const float4*>(a.x0 + src * a.x_ld + 4 * c4);
                    lpre[q] = *reinterpret_cast<const float2*>(a.lns + 2 * row);
                } else {
                    const float* base = a.x0 + row * a.x_ld + a.x_col0 + 4 * c4;
                    xpre[q] = *reinterpret_cast<const float4*>(base);
                    if (XMODE == 2) gpre[q] = *reinterpret_cast<const float4*>(base + a.x_hid);
                }
            }
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (int64_t row0 = r_begin; row0 < r_end; row0 += WG_RB) {
        __syncthreads();  // the previous block's MFMAs have read the planes
        float4 xv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 v = xpre[q];
            if (XMODE == 1) {  // RMSNorm-hat: the row's 32 float4 sit in the 32 lanes of this half wave
                float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
                ss += __shfl_xor(ss, 8); ss += __shfl_xor(ss, 16);
                const float rstd = rsqrtf(ss * (1.0f / KB) + 1.1920928955078125e-07f);
                v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
            }
            if (XMODE == 5) {  // LayerNorm-hat of the row itself, same half-wave layout
                float sm = (v.x + v.y) + (v.z + v.w);
                sm += __shfl_xor(sm, 1); sm += __shfl_xor(sm, 2); sm += __shfl_xor(sm, 4);
                sm += __shfl_xor(sm, 8); sm += __shfl_xor(sm, 16);
                const float mean = sm * (1.0f / KB);
                v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
                float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
                ss += __shfl_xor(ss, 8); ss += __shfl_xor(ss, 16);
                const float rstd = rsqrtf(ss * (1.0f / KB) + 1e-5f);
                v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
            }
            if (XMODE == 2) {
                const float4 g = gpre[q];
                v = make_float4(v.x * sigmoidf_(g.x), v.y * sigmoidf_(g.y), v.z * sigmoidf_(g.z), v.w * sigmoidf_(g.w));
            }
            if (XMODE == 3) { v.x = siluf_(v.x); v.y = siluf_(v.y); v.z = siluf_(v.z); v.w = siluf_(v.w); }
            if (XMODE == 4) {
                const float mean = lpre[q].x, rstd = lpre[q].y;  // (0, 0) past the end: the zero row stays zero
                v.x = (v.x - mean) * rstd; v.y = (v.y - mean) * rstd; v.z = (v.z - mean) * rstd; v.w = (v.w - mean) * rstd;
            }
            xv[q] = v;
            bs[0] += ypre[q].x; bs[1] += ypre[q].y; bs[2] += ypre[q].z; bs[3] += ypre[q].w;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int pp = rg + 8 * h;
            if constexpr (ONE) {
                wb_store4_high(Yp, 4 * c4, pp, ypre[2 * h], ypre[2 * h + 1]);
                wb_store4_high(Xp, 4 * c4, pp, xv[2 * h], xv[2 * h + 1]);
            } else {
                wb_store4(Yp, Yp + PLANE, Yp + 2 * PLANE, 4 * c4, pp, ypre[2 * h], ypre[2 * h + 1]);
                wb_store4(Xp, Xp + PLANE, Xp + 2 * PLANE, 4 * c4, pp, xv[2 * h], xv[2 * h + 1]);
            }
        }
        if (row0 + WG_RB < r_end) fetch(row0 + WG_RB);
        __syncthreads();
        // ---- MFMA: wave w owns output rows (dY columns) 32 w .. 32 w + 31 of this 128-row block, all 128 x columns
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int pp0 = 8 * ks + 4 * (lane >> 5);
            const int wy = wb_slot(32 * wave + (lane & 31), pp0);
            const bf16x8 ah = wb_load8(Yp + wy);
            if constexpr (ONE) {
#pragma unroll
                for (int t = 0; t < KT; t++)
                    acc[t] = PET_MFMA_BF(ah, wb_load8(Xp + wb_slot(32 * t + (lane & 31), pp0)), acc[t]);
                continue;
            }
            const bf16x8 am = wb_load8(Yp + (NP - 2) * PLANE + wy);
            const bf16x8 al = wb_load8(Yp + (NP - 1) * PLANE + wy);
#pragma unroll
            for (int t = 0; t < KT; t++) {
                const int wx = wb_slot(32 * t + (lane & 31), pp0);
                const bf16x8 bh = wb_load8(Xp + wx);
                const bf16x8 bm = wb_load8(Xp + (NP - 2) * PLANE + wx);
                const bf16x8 bl = wb_load8(Xp + (NP - 1) * PLANE + wx);
                acc[t] = PET_MFMA_BF(al, bh, acc[t]);
                acc[t] = PET_MFMA_BF(ah, bl, acc[t]);
                acc[t] = PET_MFMA_BF(am, bm, acc[t]);
                acc[t] = PET_MFMA_BF(am, bh, acc[t]);
                acc[t] = PET_MFMA_BF(ah, bm, acc[t]);
                acc[t] = PET_MFMA_BF(ah, bh, acc[t]);
            }
        }
    }
    // ---- write this split's partial: out[n = 128 nb + 32 wave + acc_row][k = 32 t + lane&31]
    float* P = a.partial + ((size_t)split * a.n_out + 128 * nb + 32 * wave) * KB;
#pragma unroll
    for (int t = 0; t < KT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) P[(size_t)acc_row(r, lane) * KB + 32 * t + (lane & 31)] = acc[t][r];
    if (a.partial_b) {  // column sums of dY: the eight row groups of a column quad, in a fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(wsm);  // [8][128]
#pragma unroll
        for (int e = 0; e < 4; e++) red[rg * 128 + 4 * c4 + e] = bs[e];
        __syncthreads();
        if (threadIdx.x < 128) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; g++) s += red[g * 128 + threadIdx.x];
            a.partial_b[(size_t)split * a.n_out + 128 * nb + threadIdx.x] = s;
        }
    }
}

// out[i] (+)= scale * sum_s partial[s][i]   (i < n); fixed summation order
__global__ void k_reduce_partials(const float* __restrict__ partial, int nsplit, int64_t n, float* __restrict__ out,
                                  int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; k++) s += partial[(size_t)k * n + i];
    out[i] = accumulate ? out[i] + s : s;
}

}  // namespace pet
