// Second-order reverse pass of the PET training step (force-loss term):
//   L = L_E(E) + L_F(dE/dR)   ->   dL/dtheta = (dE/dtheta)^T dL_E/dE + d/dtheta <u, dE/dR>,  u = dL_F/d(dE/dR).
// The reference gets the second term from autograd's double backward (evaluate_model(is_training=True)
// builds dE/dR with create_graph, pet/trainer.py:417-462). Here it is forward-over-reverse, stage by stage:
//   1. tangent sweep:  x' = J x' along dR = u  (every Linear once more on the tangent rows, every
//      nonlinearity's derivative), tangents of all Linear inputs are kept;
//   2. joint reverse sweep carrying two adjoints per activation:
//        lambda = dE_tot/dx   (the force pass' adjoint, recomputed here unfused), and
//        nu     = d/dx [ L_E + <u, dE/dR> ]:   nu_x = J^T nu_y + d/dx <lambda_y, J(x) x'>;
//      every Linear y = W x + b gets  dW += nu_y x^T + lambda_y x'^T,  db += sum nu_y.
// This first version is deliberately UNFUSED (one generic LDS-tile MFMA GEMM + small row kernels):
// it is the correctness baseline for the training row; the inference path keeps its fused kernels.
// The identities behind each row kernel were checked against torch.autograd in fp64.
#include <vector>

#include "common.h"
#include "cutoff.h"
#include "model.h"
#include "pet_ws.h"
#include "tile.h"
#include "train.h"
#include "trr.h"

namespace pet {

constexpr int LD128 = lds_ld(128);
constexpr float RMS_EPS = 1.1920928955078125e-07f;

// ---------------------------------------------------------------------------------------------
// generic GEMM:  Y[r][n] (+)= sum_k X[r][k] * cs[k] * W[n][k] + bias[n],  W packed in fragment order
// K, n_out multiples of 128; 64 rows per workgroup, K staged through LDS in chunks of 128.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_gemm(const float* __restrict__ X, int ldx, int K,
                                                   const float* __restrict__ cs, const float4* __restrict__ Wp,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                   int n_out, int64_t R, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int nk = K / 128;
    for (int nblk = 0; nblk < n_out / 128; nblk++) {
        f32x16 acc[2];
        acc_fill_bias<2>(acc, bias, 128 * nblk + 64 * w.ch, w.lane);
        for (int kc = 0; kc < nk; kc++) {
            if (nk > 1 || nblk == 0) {
                __syncthreads();
                for (int idx = threadIdx.x; idx < BM * 32; idx += NTHREADS) {
                    const int r = idx >> 5, c = idx & 31;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row0 + r < R) v = *reinterpret_cast<const float4*>(X + (row0 + r) * ldx + 128 * kc + 4 * c);
                    if (cs) {
                        const float4 s = *reinterpret_cast<const float4*>(cs + 128 * kc + 4 * c);
                        v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
                    }
                    *reinterpret_cast<float4*>(As + r * LD128 + 4 * c) = v;
                }
                __syncthreads();
            }
            gemm_acc<128, 2>(As + w.rb * 32 * LD128, LD128, Wp, K / 8, 16 * kc, 4 * nblk + 2 * w.ch, acc, w.lane);
        }
        acc_foreach<2>(acc, w.rb, 128 * nblk + 64 * w.ch, w.lane, [&](int r, int c, float v) {
            if (row0 + r < R) {
                float* y = Y + (row0 + r) * ldy + c;
                *y = accumulate ? *y + v : v;
            }
        });
    }
}

// 16-bit planes are staged in fragment order (slot 8 g + j  <->  k = 4 g + j | 8 + 4 g + j - 4), so that a lane's
// A operand is one ds_read_b128 per plane.
constexpr int LDB16 = 128 + 8;  // 16-bit elements per LDS row (272 B: rows shift by 4 banks)

// The same GEMM as f16x3 (trr.h): two fp16 planes, three MFMAs per K block on a high and a cross accumulator.
// The operands here are tangents and adjoints of arbitrary magnitude, so every staged 64 x 128 chunk is scaled row by
// row with a power of two (row maximum into [1, 2); the 32 threads that stage a row are consecutive lanes, so the
// maximum is five shuffles), its products are accumulated in chunk-local accumulators and added to the running sum
// with the inverse scale.
__global__ __launch_bounds__(NTHREADS) void k_gemm_h(const float* __restrict__ X, int ldx, int K,
                                                     const float* __restrict__ cs, W2 w, const float* __restrict__ bias,
                                                     float* __restrict__ Y, int ldy, int n_out, int64_t R,
                                                     int accumulate) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ph[];  // [2][64][LDB16], then float rinv[64]
    float* rinv = reinterpret_cast<float*>(ph + 2 * BM * LDB16);
    const WaveId wv;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int nk = K / 128, kbt = K / 16;
    const int g = wv.lane >> 5;
    for (int nblk = 0; nblk < n_out / 128; nblk++) {
        f32x16 acc[2];
        acc_fill_bias<2>(acc, bias, 128 * nblk + 64 * wv.ch, wv.lane);
        for (int kc = 0; kc < nk; kc++) {
            if (nk > 1 || nblk == 0) {
                __syncthreads();
                float4 v[BM * 32 / NTHREADS];
                float mx[BM * 32 / NTHREADS];
#pragma unroll
                for (int q = 0; q < BM * 32 / NTHREADS; q++) {
                    const int idx = threadIdx.x + q * NTHREADS, r = idx >> 5, c = idx & 31;
                    v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row0 + r < R) v[q] = *reinterpret_cast<const float4*>(X + (row0 + r) * ldx + 128 * kc + 4 * c);
                    if (cs) {
                        const float4 sc = *reinterpret_cast<const float4*>(cs + 128 * kc + 4 * c);
                        v[q].x *= sc.x; v[q].y *= sc.y; v[q].z *= sc.z; v[q].w *= sc.w;
                    }
                    float m = fmaxf(fmaxf(fabsf(v[q].x), fabsf(v[q].y)), fmaxf(fabsf(v[q].z), fabsf(v[q].w)));
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                    mx[q] = m;
                }
#pragma unroll
                for (int q = 0; q < BM * 32 / NTHREADS; q++) {
                    const int idx = threadIdx.x + q * NTHREADS, r = idx >> 5, c = idx & 31;
                    int e = (__float_as_int(mx[q]) >> 23) & 0xff;
                    e = e > 253 ? 253 : e;
                    const float sc = __int_as_float((254 - e) << 23);
                    if (c == 0) rinv[r] = __int_as_float(e << 23);
                    const int qq = c & 3, slot = 16 * (c >> 2) + (qq & 1) * 8 + (qq >> 1) * 4;
                    const float f[4] = {v[q].x * sc, v[q].y * sc, v[q].z * sc, v[q].w * sc};
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    f16x4 h, l;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        _Float16 a, b;
                        split2(f[j], a, b);
                        h[j] = a; l[j] = b;
                    }
                    _Float16* d = ph + r * LDB16 + slot;
                    *reinterpret_cast<f16x4*>(d) = h;
                    *reinterpret_cast<f16x4*>(d + BM * LDB16) = l;
                }
                __syncthreads();
            }
            const _Float16* arow = ph + (wv.rb * 32 + (wv.lane & 31)) * LDB16 + 8 * g;
            size_t base[2];
#pragma unroll
            for (int t = 0; t < 2; t++) base[t] = ((size_t)(4 * nblk + 2 * wv.ch + t) * kbt + 8 * kc) * 64 + wv.lane;
            f16x8 wh[2][2], wl[2][2];
#pragma unroll
            for (int sidx = 0; sidx < 2; sidx++)
#pragma unroll
                for (int t = 0; t < 2; t++) { wh[sidx][t] = w.h[base[t] + sidx * 64]; wl[sidx][t] = w.l[base[t] + sidx * 64]; }
            f32x16 ah[2], al[2];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) { ah[t][r] = 0.f; al[t][r] = 0.f; }
#pragma unroll
            for (int kb = 0; kb < 8; kb++) {
                const int cur = kb & 1;
                const f16x8 xh = *reinterpret_cast<const f16x8*>(arow + 16 * kb);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(arow + BM * LDB16 + 16 * kb);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    al[t] = PET_MFMA_H(xh, wl[cur][t], al[t]);
                    ah[t] = PET_MFMA_H(xh, wh[cur][t], ah[t]);
                    al[t] = PET_MFMA_H(xl, wh[cur][t], al[t]);
                }
                if (kb + 2 < 8)
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        wh[cur][t] = w.h[base[t] + (kb + 2) * 64];
                        wl[cur][t] = w.l[base[t] + (kb + 2) * 64];
                    }
            }
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[t][r] += (ah[t][r] + al[t][r] * (1.0f / 2048.0f)) * rinv[wv.rb * 32 + acc_row(r, wv.lane)];
        }
        acc_foreach<2>(acc, wv.rb, 128 * nblk + 64 * wv.ch, wv.lane, [&](int r, int c, float v) {
            if (row0 + r < R) {
                float* y = Y + (row0 + r) * ldy + c;
                *y = accumulate ? *y + v : v;
            }
        });
    }
}

// ---------------------------------------------------------------------------------------------
// TRR forms (trr.h: one wave = 32 rows, no LDS, no barrier) of the generic f16x3 GEMM for the two shape families that
// make up most of the second-order pass: K = 128 with any n_out (a multiple of 64), and 128 output columns with any K (a
// multiple of 128; wider outputs take one launch per 128 columns). Same products and the same power-of-two row scaling as k_gemm_h (one scale per row and per 128-wide
// K slice; with several slices a running scale that only shrinks, applied with selects, as in k_qkv_bwd_h).
// ---------------------------------------------------------------------------------------------
template <bool ONE>  // ONE: single fp16 term per product (train_bf16 mode), else f16x3
__global__ __launch_bounds__(256, 2) void k_rowgemm_k128(const float* __restrict__ X, int ldx,
                                                         const float* __restrict__ cs, W2 w,
                                                         const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                         int n_out, int64_t R, int accumulate) {
    __shared__ __attribute__((aligned(16))) float tiles[4][32 * ROWS_LD];  // wave-private: whole-line loads / stores (trr.h)
    const RowLane L;
    const int64_t row0 = wave_row0();
    if (row0 >= R) return;
    const bool valid = row0 + L.r < R;
    const int64_t row = valid ? row0 + L.r : R - 1;
    float* lds = tiles[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    Split2<ONE ? 1 : 8> xs;
    f16x8 xh1[ONE ? 8 : 1];
    float inv;
    {
        float4 x[16];
        load_rows_lines(x, lds, L, [&](int r) { return X + (row0 + r < R ? row0 + r : R - 1) * ldx; });
        if (cs) {
#pragma unroll
            for (int kg = 0; kg < 16; kg++) {
                const float4 c4 = *reinterpret_cast<const float4*>(cs + 8 * kg + 4 * L.h);
                x[kg].x *= c4.x; x[kg].y *= c4.y; x[kg].z *= c4.z; x[kg].w *= c4.w;
            }
        }
        float sc;
        inv = row_scale_pow2<16>(x, sc);
        if constexpr (ONE) high_frag<8>(x, xh1);
        else split_frag2<8>(x, xs);
    }
    const int nc = n_out / 64;  // 64-wide column groups = pairs of 32-wide weight tiles
    auto widx = [&](int b) { return ((size_t)(2 * (b >> 3)) * 8 + (b & 7)) * 64 + L.lane; };
    WBlk2<2> ring[ONE ? 1 : 2];
    WBlk1<2> ring1[ONE ? 4 : 1];  // the single-term form has registers to spare: four blocks in flight
    if constexpr (ONE) {
#pragma unroll
        for (int b = 0; b < 4; b++) ld_blk1<2>(ring1[b], w, widx(b < 8 * nc ? b : 8 * nc - 1), 8 * 64);
    } else {
#pragma unroll
        for (int b = 0; b < 2; b++) ld_blk2<2>(ring[b], w, widx(b), 8 * 64);
    }
#pragma unroll 1
    for (int c = 0; c < nc; c++) {
        f32x16 acc[2], acl[2];
        acc_zero<2>(acc);
        acc_zero<2>(acl);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            if constexpr (ONE) {
                WBlk1<2>& wb = ring1[kb & 3];
                mfma1<2>(acc, wb, xh1[kb]);
                int nb = 8 * c + kb + 4;
                nb = nb < 8 * nc ? nb : 8 * nc - 1;
                ld_blk1<2>(wb, w, widx(nb), 8 * 64);
            } else {
                WBlk2<2>& wb = ring[kb & 1];
                mfma3<2>(acc, acl, wb, xs.h[kb], xs.l[kb]);
                int nb = 8 * c + kb + 2;
                nb = nb < 8 * nc ? nb : 8 * nc - 1;  // past the end: a harmless reload of the last block
                ld_blk2<2>(wb, w, widx(nb), 8 * 64);
            }
        }
        if constexpr (!ONE) fold_low<2>(acc, acl);
        acc_scale<2>(acc, inv);
        float4 y[8];
        acc_to_frag<2>(acc, y);
        if (bias) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias + 64 * c + 8 * k + 4 * L.h);
                y[k].x += b4.x; y[k].y += b4.y; y[k].z += b4.z; y[k].w += b4.w;
            }
        }
        if (accumulate) {
            float4 old[8];
            load_rowfrag<8>(old, Y + 64 * c, row, ldy, L.h);
#pragma unroll
            for (int k = 0; k < 8; k++) { y[k].x += old[k].x; y[k].y += old[k].y; y[k].z += old[k].z; y[k].w += old[k].w; }
        }
        store_tile64_lines(y, lds, Y + 64 * c, row0, R, ldy, L);
    }
}

template <bool ONE>
__global__ __launch_bounds__(256) void k_rowgemm_n128(const float* __restrict__ X, int ldx, int K,
                                                      const float* __restrict__ cs, W2 w,
                                                      const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                      int64_t R, int accumulate) {
    __shared__ __attribute__((aligned(16))) float tiles[4][32 * TILE_LD];  // wave-private: whole-line stores (trr.h)
    const RowLane L;
    const int64_t row0 = wave_row0();
    if (row0 >= R) return;
    const bool valid = row0 + L.r < R;
    const int64_t row = valid ? row0 + L.r : R - 1;
    float* lds = tiles[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
    const int kbt = K / 16, nks = K / 128;
    const size_t ts = (size_t)kbt * 64;  // tile stride: 32 output columns
    auto widx = [&](int b) { return (size_t)b * 64 + L.lane; };
    WBlk2<4> ring[ONE ? 1 : 4];
    WBlk1<4> ring1[ONE ? 4 : 1];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        if constexpr (ONE) ld_blk1<4>(ring1[b], w, widx(b), ts);
        else ld_blk2<4>(ring[b], w, widx(b), ts);
    }
    f32x16 dn[4], dnl[4];
    acc_zero<4>(dn);
    acc_zero<4>(dnl);
    float4 d[16];
    load_rowfrag<16>(d, X, row, ldx, L.h);
    float scale = 0.f, inv = 0.f;  // scale applied to what the accumulators hold, and its inverse
#pragma unroll 1
    for (int ks = 0; ks < nks; ks++) {
        Split2<ONE ? 1 : 8> xs;
        f16x8 xh1[ONE ? 8 : 1];
        {
            if (cs) {
#pragma unroll
                for (int kg = 0; kg < 16; kg++) {
                    const float4 c4 = *reinterpret_cast<const float4*>(cs + 128 * ks + 8 * kg + 4 * L.h);
                    d[kg].x *= c4.x; d[kg].y *= c4.y; d[kg].z *= c4.z; d[kg].w *= c4.w;
                }
            }
            float sc;
            const float iv = row_pow2<16>(d, sc);
            const bool shrink = ks == 0 || sc < scale;
            const float sc_eff = shrink ? sc : scale;
            if (ks > 0) {
                const float f = sc_eff * inv;  // 1 unless this slice is larger than everything before it
                acc_scale<4>(dn, f);
                acc_scale<4>(dnl, f);
            }
            scale = sc_eff;
            inv = shrink ? iv : inv;
#pragma unroll
            for (int kg = 0; kg < 16; kg++) { d[kg].x *= sc_eff; d[kg].y *= sc_eff; d[kg].z *= sc_eff; d[kg].w *= sc_eff; }
            if constexpr (ONE) high_frag<8>(d, xh1);
            else split_frag2<8>(d, xs);
        }
        if (ks + 1 < nks) load_rowfrag<16>(d, X + 128 * (ks + 1), row, ldx, L.h);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            int nb = 8 * ks + kb + 4;
            nb = nb < kbt ? nb : kbt - 1;
            if constexpr (ONE) {
                WBlk1<4>& wb = ring1[kb & 3];
                mfma1<4>(dn, wb, xh1[kb]);
                ld_blk1<4>(wb, w, widx(nb), ts);
            } else {
                WBlk2<4>& wb = ring[kb & 3];
                mfma3<4>(dn, dnl, wb, xs.h[kb], xs.l[kb]);
                ld_blk2<4>(wb, w, widx(nb), ts);
            }
        }
    }
    if constexpr (!ONE) fold_low<4>(dn, dnl);
    acc_scale<4>(dn, inv);
    float4 y[16];
    acc_to_frag<4>(dn, y);
    if (bias) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + 8 * k + 4 * L.h);
            y[k].x += b4.x; y[k].y += b4.y; y[k].z += b4.z; y[k].w += b4.w;
        }
    }
    if (accumulate) {
        float4 old[16];
        load_rowfrag<16>(old, Y, row, ldy, L.h);
#pragma unroll
        for (int k = 0; k < 16; k++) { y[k].x += old[k].x; y[k].y += old[k].y; y[k].z += old[k].z; y[k].w += old[k].w; }
    }
    store_rows_lines<16>(y, lds, L, [&](int r) { return row0 + r < R ? Y + (row0 + r) * ldy : nullptr; });
}

static int g_so_trr = 1;  // pet_config_set("so_trr", 0): every generic GEMM through the LDS-tile k_gemm_h
void set_so_trr(int v) { g_so_trr = v ? 1 : 0; }
// Y[R, n_out] (=|+=) (X[R, K] * cs) W^T + bias on the TRR kernels when the shape allows; false otherwise
static int g_train_bf16 = 0;  // pet_config_set("train_bf16", 1): ONE 16-bit MFMA term per product in the training GEMMs
void set_train_bf16(int v) { g_train_bf16 = v ? 1 : 0; }
int train_bf16() { return g_train_bf16; }
static bool rowgemm_trr(hipStream_t st, const float* X, int K, const float* cs, W2 w, const float* bias, float* Y,
                        int n_out, int64_t R, bool acc) {
    if (!g_so_trr) return false;
    const int grid = (int)cdiv(R, WG_ROWS);
    if (K == 128 && n_out % 64 == 0) {
        if (g_train_bf16) k_rowgemm_k128<true><<<grid, 256, 0, st>>>(X, K, cs, w, bias, Y, n_out, n_out, R, acc ? 1 : 0);
        else k_rowgemm_k128<false><<<grid, 256, 0, st>>>(X, K, cs, w, bias, Y, n_out, n_out, R, acc ? 1 : 0);
        return true;
    }
    if (n_out % 128 == 0 && K % 128 == 0) {  // 128 output columns per launch (k_gemm_h re-stages X per column block too)
        const size_t ts4 = (size_t)4 * (K / 16) * 64;  // four 32-column weight tiles
        for (int nb = 0; nb < n_out / 128; nb++) {
            W2 wn; wn.h = w.h + nb * ts4; wn.l = w.l + nb * ts4;
            if (g_train_bf16)
                k_rowgemm_n128<true><<<grid, 256, 0, st>>>(X, K, K, cs, wn, bias ? bias + 128 * nb : nullptr, Y + 128 * nb,
                                                           n_out, R, acc ? 1 : 0);
            else
                k_rowgemm_n128<false><<<grid, 256, 0, st>>>(X, K, K, cs, wn, bias ? bias + 128 * nb : nullptr, Y + 128 * nb,
                                                            n_out, R, acc ? 1 : 0);
        }
        return true;
    }
    return false;
}

static int g_so_f16x3 = 1;  // pet_config_set("so_f16x3", 0): generic training GEMMs on the fp32 MFMA
void set_so_f16x3(int v) { g_so_f16x3 = v ? 1 : 0; }
static inline W2 w2_at(const void* base, int n_out, int k_in) {
    const size_t n8 = (size_t)(n_out / 32) * (k_in / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}
struct Ctx {
    const Model& m;
    const Graph& g;
    hipStream_t st;
};

// y = x W^T (+ b): forward orientation
static void mm_fwd(const Ctx& c, const Lin& L, const float* X, float* Y, int64_t R, bool bias, const float* cs = nullptr,
                   bool acc = false) {
    if (R <= 0) return;
    ProfScope ps("so_gemm", c.st, 2.0 * (double)R * L.k_in * L.n_out, 4.0 * (double)R * (L.k_in + L.n_out));
    if (g_so_f16x3 && !g_train_bf16 && g_so_trr &&
        rowgemm_s(c.st, X, L.k_in, cs, L.fwd2s, bias ? L.b : nullptr, Y, L.n_out, R, acc)) {  // large row counts (so_rows_s.hip)
    } else if (g_so_f16x3 && L.fwd2 &&
        rowgemm_trr(c.st, X, L.k_in, cs, w2_at(L.fwd2, L.n_out, L.k_in), bias ? L.b : nullptr, Y, L.n_out, R, acc)) {
    } else if (g_so_f16x3 && L.fwd2)
        k_gemm_h<<<cdiv(R, BM), NTHREADS, 2 * BM * LDB16 * 2 + BM * 4, c.st>>>(X, L.k_in, L.k_in, cs,
                                                                               w2_at(L.fwd2, L.n_out, L.k_in),
                                                                               bias ? L.b : nullptr, Y, L.n_out, L.n_out, R,
                                                                               acc ? 1 : 0);
    else
        k_gemm<<<cdiv(R, BM), NTHREADS, BM * LD128 * 4, c.st>>>(X, L.k_in, L.k_in, cs, L.fwd, bias ? L.b : nullptr, Y,
                                                               L.n_out, L.n_out, R, acc ? 1 : 0);
}
// x_adj = y_adj W: transposed orientation
static void mm_bwd(const Ctx& c, const Lin& L, const float* Yadj, float* Xadj, int64_t R, bool acc = false) {
    if (R <= 0) return;
    ProfScope ps("so_gemm", c.st, 2.0 * (double)R * L.k_in * L.n_out, 4.0 * (double)R * (L.k_in + L.n_out));
    if (g_so_f16x3 && !g_train_bf16 && g_so_trr && rowgemm_s(c.st, Yadj, L.n_out, nullptr, L.bwd2s, nullptr, Xadj, L.k_in, R, acc)) {
    } else if (g_so_f16x3 && L.bwd2 &&
        rowgemm_trr(c.st, Yadj, L.n_out, nullptr, w2_at(L.bwd2, L.k_in, L.n_out), nullptr, Xadj, L.k_in, R, acc)) {
    } else if (g_so_f16x3 && L.bwd2)  // the transposed operand: tiles over k_in, K = n_out
        k_gemm_h<<<cdiv(R, BM), NTHREADS, 2 * BM * LDB16 * 2 + BM * 4, c.st>>>(Yadj, L.n_out, L.n_out, nullptr,
                                                                               w2_at(L.bwd2, L.k_in, L.n_out), nullptr, Xadj,
                                                                               L.k_in, L.k_in, R, acc ? 1 : 0);
    else
        k_gemm<<<cdiv(R, BM), NTHREADS, BM * LD128 * 4, c.st>>>(Yadj, L.n_out, L.n_out, nullptr, L.bwd, nullptr, Xadj,
                                                               L.k_in, L.k_in, R, acc ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__global__ void k_add3(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                       float* __restrict__ out, int64_t n4) {  // out = a + b (+ c), float4 granularity
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 x = reinterpret_cast<const float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    if (c) {
        const float4 z = reinterpret_cast<const float4*>(c)[i];
        x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w;
    }
    reinterpret_cast<float4*>(out)[i] = x;
}
static void add3(const Ctx& c, const float* a, const float* b, const float* cc, float* out, int64_t n) {
    if (n > 0) k_add3<<<cdiv(n / 4, 256), 256, 0, c.st>>>(a, b, cc, out, n / 4);
}

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// ---------------------------------------------------------------------------------------------
// geometry tangent along dR = u:  v' = u[nbr] - u[ctr];  (v, dist)' , fc' , (log fc)'
// ---------------------------------------------------------------------------------------------
// adaptive cutoff: tangent of the per-atom cutoff through the implicit-function step,
//   r_i' = -(1 / dn_root_i) sum_{q in all-edge row i} (d bump / d d)(d_q; r_i) d_q',   d_q' = v_q . (u_j - u_i) / |v_q|
__global__ void k_adapt_rdot(const float* __restrict__ u, const int* __restrict__ rowptr0,
                             const int* __restrict__ perm0, const int* __restrict__ nbr0,
                             const float4* __restrict__ vin, const float* __restrict__ r_newton,
                             const float* __restrict__ inv_dn, float* __restrict__ rdot, int N, float w,
                             const float* __restrict__ ucell, const int* __restrict__ shift0,
                             const int* __restrict__ sys) {
    const int gid = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    const int a = gid < N ? gid : N - 1;
    const float r = r_newton[a];
    float s = 0.f;
    for (int q = rowptr0[a] + l; q < rowptr0[a + 1]; q += 16) {
        const float4 v = vin[perm0[q]];
        const int j = nbr0[q];
        const float nrm = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
        float tx = u[3 * j] - u[3 * a], ty = u[3 * j + 1] - u[3 * a + 1], tz = u[3 * j + 2] - u[3 * a + 2];
        if (ucell) {  // the cell moves too: v' += S . cell'
            const float* c = ucell + 9 * sys[a];
            const float sa = (float)shift0[3 * q], sb = (float)shift0[3 * q + 1], sc = (float)shift0[3 * q + 2];
            tx += sa * c[0] + sb * c[3] + sc * c[6];
            ty += sa * c[1] + sb * c[4] + sc * c[7];
            tz += sa * c[2] + sb * c[5] + sc * c[8];
        }
        const float dd = nrm > 0.f ? (v.x * tx + v.y * ty + v.z * tz) / nrm : 0.f;
        s += cutoff_deriv_dev(v.w, r, w, PET_CUTOFF_BUMP) * dd;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (l == 0 && gid < N) rdot[gid] = -inv_dn[gid] * s;
}

// "grid" method (adaptive_cutoff.py legacy path): r_i = sum_k p_k w_k(n_i1 .. n_iK), n_ik = sum_q bump(d_q; p_k, w), so
//   r_i' = sum_q [ sum_k (d r_i / d n_ik) (d bump / d d)(d_q; p_k, w) ] d_q'   with the per-atom row drdn the graph build kept
// (the transpose of pet_bwd.hip k_adapt_dv_grid)
__global__ void k_adapt_rdot_grid(const float* __restrict__ u, const int* __restrict__ rowptr0, const int* __restrict__ perm0,
                                  const int* __restrict__ nbr0, const float4* __restrict__ vin,
                                  const float* __restrict__ drdn, float* __restrict__ rdot, int N, float w, int K, float pmin,
                                  float dp, const float* __restrict__ ucell, const int* __restrict__ shift0,
                                  const int* __restrict__ sys) {
    __shared__ float s_c[16][GRID_MAX_PROBES];
    const int grp = threadIdx.x >> 4;
    const int gid = blockIdx.x * (blockDim.x / 16) + grp;
    const int l = threadIdx.x & 15;
    const int a = gid < N ? gid : N - 1;
    for (int k = l; k < K; k += 16) s_c[grp][k] = drdn[(int64_t)a * GRID_MAX_PROBES + k];
    __syncthreads();
    float s = 0.f;
    for (int q = rowptr0[a] + l; q < rowptr0[a + 1]; q += 16) {
        const float4 v = vin[perm0[q]];
        const int j = nbr0[q];
        const float nrm = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
        float tx = u[3 * j] - u[3 * a], ty = u[3 * j + 1] - u[3 * a + 1], tz = u[3 * j + 2] - u[3 * a + 2];
        if (ucell) {
            const float* c = ucell + 9 * sys[a];
            const float sa = (float)shift0[3 * q], sb = (float)shift0[3 * q + 1], sc = (float)shift0[3 * q + 2];
            tx += sa * c[0] + sb * c[3] + sc * c[6];
            ty += sa * c[1] + sb * c[4] + sc * c[7];
            tz += sa * c[2] + sb * c[5] + sc * c[8];
        }
        const float dd = nrm > 0.f ? (v.x * tx + v.y * ty + v.z * tz) / nrm : 0.f;
        float coef = 0.f;
        for (int k = 0; k < K; k++) coef += s_c[grp][k] * cutoff_deriv_dev(v.w, pmin + (float)k * dp, w, PET_CUTOFF_BUMP);
        s += coef * dd;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (l == 0 && gid < N) rdot[gid] = s;
}
// tangent of the atomic cutoffs into g.ad_gr (either method)
static void adaptive_rdot(const Model& m, const Graph& g, const float* u, const float* ucell, hipStream_t st) {
    const int N = (int)g.n_nodes;
    if (g.grid_probes > 0)
        k_adapt_rdot_grid<<<cdiv(N, 16), 256, 0, st>>>(u, g.rowptr0, g.perm0, g.nbr0, g.vin, g.grid_drdn, g.ad_gr, N,
                                                       m.h.cutoff_width_adaptive, g.grid_probes, 0.5f,
                                                       m.h.cutoff_width_adaptive / 4.0f, ucell, g.shift0, g.sys);
    else
        k_adapt_rdot<<<cdiv(N, 16), 256, 0, st>>>(u, g.rowptr0, g.perm0, g.nbr0, g.vin, g.r_newton, g.inv_dn, g.ad_gr, N,
                                                  m.h.cutoff_width_adaptive, ucell, g.shift0, g.sys);
}

__global__ void k_geom_jvp(const float* __restrict__ u, const int* __restrict__ ctr, const int* __restrict__ nbr,
                           const float4* __restrict__ geo, const float* __restrict__ d0, const float* __restrict__ fc,
                           float4* __restrict__ Tgeo, float* __restrict__ Tfc, float* __restrict__ Tkb, int64_t E,
                           float cutoff, float width, int fn, const float* __restrict__ pc,
                           const float* __restrict__ rdot, const float* __restrict__ ucell,
                           const int* __restrict__ shift, const int* __restrict__ sys) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    const int i = ctr[p], j = nbr[p];
    float vx = u[3 * j] - u[3 * i], vy = u[3 * j + 1] - u[3 * i + 1], vz = u[3 * j + 2] - u[3 * i + 2];
    if (ucell) {  // tangent of the cell (a stress term in the loss): v = r_j - r_i + S . cell, so v' += S . cell'
        const float* c = ucell + 9 * sys[i];
        const float sa = (float)shift[3 * p], sb = (float)shift[3 * p + 1], sc = (float)shift[3 * p + 2];
        vx += sa * c[0] + sb * c[3] + sc * c[6];
        vy += sa * c[1] + sb * c[4] + sc * c[7];
        vz += sa * c[2] + sb * c[5] + sc * c[8];
    }
    const float4 g = geo[p];
    const float vd = g.x * vx + g.y * vy + g.z * vz;
    const float nrm = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
    Tgeo[p] = make_float4(vx, vy, vz, vd / g.w);
    const float dd0 = nrm > 0.f ? vd / nrm : 0.f;
    // fc = f(d - c): with the adaptive pair cutoff c = (r_i + r_j) / 2 the tangent is f_d (d' - c')
    const float cd = pc ? 0.5f * (rdot[i] + rdot[j]) : 0.f;
    const float dfc = cutoff_deriv_dev(d0[p], pc ? pc[p] : cutoff, width, fn) * (dd0 - cd);
    Tfc[p] = dfc;
    const float f = fc[p];
    Tkb[p] = f >= 1e-15f ? dfc / f : 0.f;
}

// the two kernels above for any caller (gen_train.hip: the size-generic pass with the adaptive cutoff)
int geometry_tangent(const Model& m, const Graph& g, const float* u, const float* ucell, float* Tgeo, float* Tfc, float* Tkb,
                     hipStream_t st) {
    const int64_t E = g.n_edges;
    if (g.adaptive) adaptive_rdot(m, g, u, ucell, st);  // g.ad_gr doubles as the tangent of the atomic cutoffs
    k_geom_jvp<<<cdiv(E, 256), 256, 0, st>>>(u, g.ctr, g.nbr, g.geo, g.d0, g.fc, reinterpret_cast<float4*>(Tgeo), Tfc, Tkb, E,
                                            m.h.cutoff, m.h.cutoff_width, m.h.cutoff_function, g.adaptive ? g.pc : nullptr,
                                            g.adaptive ? g.ad_gr : nullptr, ucell, g.shift, g.sys);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// Ta0[p][c] = Tgeo[p] . Wc[c]   (4 -> D)
__global__ void k_geo_lin(const float4* __restrict__ Tgeo, const float* __restrict__ wc, float* __restrict__ out,
                          int64_t E) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * D) return;
    const int64_t p = idx / D;
    const int c = (int)(idx % D);
    const float4 g = Tgeo[p];
    const float4 wv = reinterpret_cast<const float4*>(wc)[c];
    out[idx] = g.x * wv.x + g.y * wv.y + g.z * wv.z + g.w * wv.w;
}

// ---------------------------------------------------------------------------------------------
// elementwise nonlinearities: tangent and joint reverse
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void silu_d(float a, float& s0, float& d1, float& d2) {
    const float s = sigmoidf_(a);
    s0 = a * s;
    d1 = s * (1.0f + a * (1.0f - s));
    d2 = s * (1.0f - s) * (2.0f + a * (1.0f - 2.0f * s));
}
// s = silu(a) (optional), sd = silu'(a) ad
__global__ void k_silu_jvp(const float* __restrict__ a, const float* __restrict__ ad, float* __restrict__ s,
                           float* __restrict__ sd, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0, d1, d2;
    silu_d(a[i], s0, d1, d2);
    if (s) s[i] = s0;
    sd[i] = d1 * ad[i];
}
// in place on (ls, ns): lambda_a = s' lambda_s,  nu_a = s' nu_s + s'' a' lambda_s
__global__ void k_silu_rev(const float* __restrict__ a, const float* __restrict__ ad, float* __restrict__ ls,
                           float* __restrict__ ns, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0, d1, d2;
    silu_d(a[i], s0, d1, d2);
    const float l = ls[i];
    ls[i] = d1 * l;
    ns[i] = d1 * ns[i] + d2 * ad[i] * l;
}
// u = v sigmoid(g):  ud = vd s + v s' gd        VG = [v | g] with hidden size H
__global__ void k_swiglu_jvp(const float* __restrict__ VG, const float* __restrict__ VGd, float* __restrict__ Ud,
                             int64_t rows, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * H) return;
    const int64_t r = i / H;
    const int c = (int)(i % H);
    const float v = VG[r * 2 * H + c], g = VG[r * 2 * H + H + c];
    const float vd = VGd[r * 2 * H + c], gd = VGd[r * 2 * H + H + c];
    const float s = sigmoidf_(g), s1 = s * (1.0f - s);
    Ud[i] = vd * s + v * s1 * gd;
}
// (lu, nu) [rows,H] -> (lVG, nVG) [rows,2H]
__global__ void k_swiglu_rev(const float* __restrict__ VG, const float* __restrict__ VGd, const float* __restrict__ lu,
                             const float* __restrict__ nu, float* __restrict__ lVG, float* __restrict__ nVG,
                             int64_t rows, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * H) return;
    const int64_t r = i / H;
    const int c = (int)(i % H);
    const float v = VG[r * 2 * H + c], g = VG[r * 2 * H + H + c];
    const float vd = VGd[r * 2 * H + c], gd = VGd[r * 2 * H + H + c];
    const float s = sigmoidf_(g), s1 = s * (1.0f - s), s2 = s1 * (1.0f - 2.0f * s);
    const float l = lu[i], n = nu[i];
    lVG[r * 2 * H + c] = l * s;
    lVG[r * 2 * H + H + c] = l * v * s1;
    nVG[r * 2 * H + c] = n * s + l * s1 * gd;
    nVG[r * 2 * H + H + c] = n * v * s1 + l * (vd * s1 + v * s2 * gd);
}

// ---------------------------------------------------------------------------------------------
// RMSNorm (gamma folded into the following Linear): xhat = x r, r = rsqrt(mean x^2 + eps)
//   xhat' = r (x' - xhat A),  A = mean(xhat x')
// one float4 per lane, K/4 lanes per row
// ---------------------------------------------------------------------------------------------
// LN = true: LayerNorm-hat. (x - mean) / sqrt(var + eps) is the RMS-hat of the CENTRED row, and centring is a linear,
// symmetric projection P = I - 11^T / K: the tangent is the RMS tangent of (P x, P x'), the reverse sweep is the RMS
// reverse sweep on the centred row followed by P on both adjoints (a linear map adds no second-order term).
constexpr float LN_EPS = 1e-5f;
__device__ __forceinline__ float sum4(const float4& v) { return (v.x + v.y) + (v.z + v.w); }
template <int LPR, int K>
__device__ __forceinline__ void centre_row(float4& v) {
    const float mu = group_sum<LPR>(sum4(v)) * (1.0f / K);
    v.x -= mu; v.y -= mu; v.z -= mu; v.w -= mu;
}
template <int K, bool LN>
__global__ void k_rms_jvp(const float* __restrict__ X, const float* __restrict__ Xd, float* __restrict__ XHd,
                          int64_t R) {
    constexpr int LPR = K / 4;
    const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const int c = threadIdx.x % LPR;
    const bool valid = row < R;
    float4 x = make_float4(0, 0, 0, 0), xd = x;
    if (valid) {
        x = *reinterpret_cast<const float4*>(X + row * K + 4 * c);
        xd = *reinterpret_cast<const float4*>(Xd + row * K + 4 * c);
    }
    if (LN) { centre_row<LPR, K>(x); centre_row<LPR, K>(xd); }
    const float ss = group_sum<LPR>(dot4(x, x));
    const float xxd = group_sum<LPR>(dot4(x, xd));
    const float r = rsqrtf(ss * (1.0f / K) + (LN ? LN_EPS : RMS_EPS));
    const float rA = r * r * r * xxd * (1.0f / K);  // r * A / (x -> xhat scale): xhat A r = x r^2 A, A = r mean(x x')
    if (valid)
        *reinterpret_cast<float4*>(XHd + row * K + 4 * c) =
            make_float4(r * xd.x - rA * x.x, r * xd.y - rA * x.y, r * xd.z - rA * x.z, r * xd.w - rA * x.w);
}
// (lin_l, lin_n): adjoints w.r.t. the Linear input xn = gamma * xhat.  Accumulates into (LX, NX):
//   lambda_x += r (l - xhat B)
//   nu_x     += r (n - xhat m(xhat, n)) - r^2 [ (C - 3AB) xhat + B x' + A l ],   l = gamma*lin_l, n = gamma*lin_n,
//   A = m(xhat, x'), B = m(xhat, l), C = m(l, x')
template <int K, bool LN>
__global__ void k_rms_rev(const float* __restrict__ X, const float* __restrict__ Xd, const float* __restrict__ lin_l,
                          const float* __restrict__ lin_n, const float* __restrict__ gamma, float* __restrict__ LX,
                          float* __restrict__ NX, int64_t R) {
    constexpr int LPR = K / 4;
    const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    const int c = threadIdx.x % LPR;
    const bool valid = row < R;
    float4 x = make_float4(0, 0, 0, 0), xd = x, l = x, n = x;
    if (valid) {
        const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * c);
        x = *reinterpret_cast<const float4*>(X + row * K + 4 * c);
        xd = *reinterpret_cast<const float4*>(Xd + row * K + 4 * c);
        l = *reinterpret_cast<const float4*>(lin_l + row * K + 4 * c);
        n = *reinterpret_cast<const float4*>(lin_n + row * K + 4 * c);
        l.x *= gm.x; l.y *= gm.y; l.z *= gm.z; l.w *= gm.w;
        n.x *= gm.x; n.y *= gm.y; n.z *= gm.z; n.w *= gm.w;
    }
    if (LN) { centre_row<LPR, K>(x); centre_row<LPR, K>(xd); }
    const float ss = group_sum<LPR>(dot4(x, x));
    const float r = rsqrtf(ss * (1.0f / K) + (LN ? LN_EPS : RMS_EPS));
    const float4 xh = make_float4(x.x * r, x.y * r, x.z * r, x.w * r);
    const float A = group_sum<LPR>(dot4(xh, xd)) * (1.0f / K);
    const float B = group_sum<LPR>(dot4(xh, l)) * (1.0f / K);
    const float C = group_sum<LPR>(dot4(l, xd)) * (1.0f / K);
    const float Mn = group_sum<LPR>(dot4(xh, n)) * (1.0f / K);
    const float r2 = r * r, k3 = C - 3.0f * A * B;
    float4 da = make_float4(r * (l.x - xh.x * B), r * (l.y - xh.y * B), r * (l.z - xh.z * B), r * (l.w - xh.w * B));
    float4 db = make_float4(r * (n.x - xh.x * Mn) - r2 * (k3 * xh.x + B * xd.x + A * l.x),
                            r * (n.y - xh.y * Mn) - r2 * (k3 * xh.y + B * xd.y + A * l.y),
                            r * (n.z - xh.z * Mn) - r2 * (k3 * xh.z + B * xd.z + A * l.z),
                            r * (n.w - xh.w * Mn) - r2 * (k3 * xh.w + B * xd.w + A * l.w));
    if (LN) { centre_row<LPR, K>(da); centre_row<LPR, K>(db); }  // P^T = P on both adjoints
    if (!valid) return;
    float4* lx = reinterpret_cast<float4*>(LX + row * K + 4 * c);
    float4* nx = reinterpret_cast<float4*>(NX + row * K + 4 * c);
    float4 a = *lx, b = *nx;
    a.x += da.x; a.y += da.y; a.z += da.z; a.w += da.w;
    b.x += db.x; b.y += db.y; b.z += db.z; b.w += db.w;
    *lx = a;
    *nx = b;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm of cat = [x ; x[rev]] (2D columns; affine folded into the following Linear).
// One wave per row, lane c holds float4 c of the 256 columns: c < 32 -> x[p], else x[rev[p]].
// LayerNorm = RMSNorm of the centred row, so the formulas above apply to xc = cat - mean, followed
// by the same centring of the result.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4_sub(float4 a, float s) { return make_float4(a.x - s, a.y - s, a.z - s, a.w - s); }
__device__ __forceinline__ float f4_sum(float4 a) { return a.x + a.y + a.z + a.w; }

__global__ void k_lncat_jvp(const float* __restrict__ XF, const float* __restrict__ XFd, const int* __restrict__ rev,
                            const float* __restrict__ LNS, float* __restrict__ CHd, int64_t E) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    const bool valid = p < E;
    float4 x = make_float4(0, 0, 0, 0), xd = x;
    float mean = 0.f, r = 0.f;
    if (valid) {
        const int64_t src = c < 32 ? p : (int64_t)rev[p];
        x = *reinterpret_cast<const float4*>(XF + src * D + 4 * (c & 31));
        xd = *reinterpret_cast<const float4*>(XFd + src * D + 4 * (c & 31));
        mean = LNS[2 * p];
        r = LNS[2 * p + 1];
    }
    const float md = group_sum<64>(f4_sum(xd)) * (1.0f / (2 * D));
    const float4 xh = make_float4((x.x - mean) * r, (x.y - mean) * r, (x.z - mean) * r, (x.w - mean) * r);
    const float4 xcd = f4_sub(xd, md);
    const float A = group_sum<64>(dot4(xh, xcd)) * (1.0f / (2 * D));
    if (valid)
        *reinterpret_cast<float4*>(CHd + p * 2 * D + 4 * c) = make_float4(
            r * (xcd.x - xh.x * A), r * (xcd.y - xh.y * A), r * (xcd.z - xh.z * A), r * (xcd.w - xh.w * A));
}
// (lin_l, lin_n) [E,2D] adjoints w.r.t. the Linear input y = gamma xhat + beta  ->  (Lc, Nc) [E,2D] adjoints of cat
__global__ void k_lncat_rev(const float* __restrict__ XF, const float* __restrict__ XFd, const int* __restrict__ rev,
                            const float* __restrict__ LNS, const float* __restrict__ lin_l,
                            const float* __restrict__ lin_n, const float* __restrict__ gamma, float* __restrict__ Lc,
                            float* __restrict__ Nc, int64_t E) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    const bool valid = p < E;
    float4 x = make_float4(0, 0, 0, 0), xd = x, l = x, n = x;
    float mean = 0.f, r = 0.f;
    if (valid) {
        const int64_t src = c < 32 ? p : (int64_t)rev[p];
        x = *reinterpret_cast<const float4*>(XF + src * D + 4 * (c & 31));
        xd = *reinterpret_cast<const float4*>(XFd + src * D + 4 * (c & 31));
        mean = LNS[2 * p];
        r = LNS[2 * p + 1];
        const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * c);
        l = *reinterpret_cast<const float4*>(lin_l + p * 2 * D + 4 * c);
        n = *reinterpret_cast<const float4*>(lin_n + p * 2 * D + 4 * c);
        l.x *= gm.x; l.y *= gm.y; l.z *= gm.z; l.w *= gm.w;
        n.x *= gm.x; n.y *= gm.y; n.z *= gm.z; n.w *= gm.w;
    }
    constexpr float IK = 1.0f / (2 * D);
    const float md = group_sum<64>(f4_sum(xd)) * IK;
    const float4 xh = make_float4((x.x - mean) * r, (x.y - mean) * r, (x.z - mean) * r, (x.w - mean) * r);
    const float4 xcd = f4_sub(xd, md);
    const float A = group_sum<64>(dot4(xh, xcd)) * IK;
    const float B = group_sum<64>(dot4(xh, l)) * IK;
    const float C = group_sum<64>(dot4(l, xcd)) * IK;
    const float Mn = group_sum<64>(dot4(xh, n)) * IK;
    const float r2 = r * r, k3 = C - 3.0f * A * B;
    float4 a = make_float4(r * (l.x - xh.x * B), r * (l.y - xh.y * B), r * (l.z - xh.z * B), r * (l.w - xh.w * B));
    float4 b = make_float4(r * (n.x - xh.x * Mn) - r2 * (k3 * xh.x + B * xcd.x + A * l.x),
                           r * (n.y - xh.y * Mn) - r2 * (k3 * xh.y + B * xcd.y + A * l.y),
                           r * (n.z - xh.z * Mn) - r2 * (k3 * xh.z + B * xcd.z + A * l.z),
                           r * (n.w - xh.w * Mn) - r2 * (k3 * xh.w + B * xcd.w + A * l.w));
    const float ma = group_sum<64>(f4_sum(a)) * IK, mb = group_sum<64>(f4_sum(b)) * IK;
    if (!valid) return;
    *reinterpret_cast<float4*>(Lc + p * 2 * D + 4 * c) = f4_sub(a, ma);
    *reinterpret_cast<float4*>(Nc + p * 2 * D + 4 * c) = f4_sub(b, mb);
}
// out[p][c] = base[p][c] + cat[p][c] + cat[rev[p]][D + c]   (adjoint of cat = [x ; x[rev]] plus a pass-through)
__global__ void k_cat_gather(const float* __restrict__ base, const float* __restrict__ cat, const int* __restrict__ rev,
                             float* __restrict__ out, int64_t E) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index over [E, D/4]
    if (idx >= E * (D / 4)) return;
    const int64_t p = idx / (D / 4);
    const int c = (int)(idx % (D / 4));
    float4 a = reinterpret_cast<const float4*>(base)[idx];
    const float4 b = *reinterpret_cast<const float4*>(cat + p * 2 * D + 4 * c);
    const float4 d = *reinterpret_cast<const float4*>(cat + (int64_t)rev[p] * 2 * D + D + 4 * c);
    a.x += b.x + d.x; a.y += b.y + d.y; a.z += b.z + d.z; a.w += b.w + d.w;
    reinterpret_cast<float4*>(out)[idx] = a;
}

// ---------------------------------------------------------------------------------------------
// attention of one (atom, head): tokens t = 0 (centre) and 1..n (edges), T <= 128.
//   s_ij = scale q_i.k_j + b_j,  P = softmax_j,  o_i = sum_j P_ij v_j,   b_j = log max(fc_j, 1e-15), b_0 = 0
// Plain VALU implementation, one wave per (atom, head); per-token rows of K, V, ... live in LDS.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t tok_row(int t, int64_t E, int atom, int start) {
    return t == 0 ? E + atom : (int64_t)start + t - 1;
}
__device__ __forceinline__ void ld16(float (&dst)[HD], const float* __restrict__ src) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 v = reinterpret_cast<const float4*>(src)[q];
        dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
    }
}
__device__ __forceinline__ void st16(float* __restrict__ dst, const float (&src)[HD], float mul) {
#pragma unroll
    for (int q = 0; q < 4; q++)
        reinterpret_cast<float4*>(dst)[q] =
            make_float4(src[4 * q] * mul, src[4 * q + 1] * mul, src[4 * q + 2] * mul, src[4 * q + 3] * mul);
}
__device__ __forceinline__ float dot16(const float (&a)[HD], const float* b) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < HD; q++) s += a[q] * b[q];
    return s;
}

// LDS staging of `what` (column offset in the QKV row: 0 q, D k, 2D v) for all tokens: dst[t][16]
__device__ __forceinline__ void stage_tokens(float* dst, const float* __restrict__ src, int col, int T, int64_t E,
                                             int atom, int start, int head, float mul) {
    for (int idx = threadIdx.x; idx < T * 4; idx += 64) {
        const int t = idx >> 2, q = idx & 3;
        float4 v = *reinterpret_cast<const float4*>(src + tok_row(t, E, atom, start) * (3 * D) + col + HD * head + 4 * q);
        v.x *= mul; v.y *= mul; v.z *= mul; v.w *= mul;
        *reinterpret_cast<float4*>(dst + t * HD + 4 * q) = v;
    }
}

__global__ __launch_bounds__(64) void k_attn_jvp(const float* __restrict__ QKV, const float* __restrict__ QKVd,
                                                 const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                 const float* __restrict__ Tkb, float* __restrict__ AOd, int64_t E,
                                                 int N, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int atom = blockIdx.x / NHEAD, head = blockIdx.x % NHEAD;
    const int start = rowptr[atom];
    const int T = rowptr[atom + 1] - start + 1;
    float* Ks = smem;
    float* Vs = Ks + T * HD;
    float* Kd = Vs + T * HD;
    float* Vd = Kd + T * HD;
    float* bs = Vd + T * HD;  // [T] bias, [T] bias tangent
    float* bd = bs + T;
    stage_tokens(Ks, QKV, D, T, E, atom, start, head, 1.f);
    stage_tokens(Vs, QKV, 2 * D, T, E, atom, start, head, 1.f);
    stage_tokens(Kd, QKVd, D, T, E, atom, start, head, 1.f);
    stage_tokens(Vd, QKVd, 2 * D, T, E, atom, start, head, 1.f);
    for (int t = threadIdx.x; t < T; t += 64) {
        bs[t] = t == 0 ? 0.f : logf(fmaxf(fc[start + t - 1], 1e-15f));
        bd[t] = t == 0 ? 0.f : Tkb[start + t - 1];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += 64) {
        const int64_t row = tok_row(i, E, atom, start);
        float q[HD], qd[HD];
        ld16(q, QKV + row * (3 * D) + HD * head);
        ld16(qd, QKVd + row * (3 * D) + HD * head);
#pragma unroll
        for (int c = 0; c < HD; c++) { q[c] *= scale; qd[c] *= scale; }
        float mx = -INFINITY;
        for (int j = 0; j < T; j++) mx = fmaxf(mx, dot16(q, Ks + j * HD) + bs[j]);
        float l = 0.f, an = 0.f, o1[HD], o2[HD], o3[HD];
#pragma unroll
        for (int c = 0; c < HD; c++) o1[c] = o2[c] = o3[c] = 0.f;
        for (int j = 0; j < T; j++) {
            const float e = expf(dot16(q, Ks + j * HD) + bs[j] - mx);
            const float sd = dot16(qd, Ks + j * HD) + dot16(q, Kd + j * HD) + bd[j];
            l += e;
            an += e * sd;
#pragma unroll
            for (int c = 0; c < HD; c++) {
                const float v = Vs[j * HD + c];
                o1[c] += e * v;
                o2[c] += e * sd * v;
                o3[c] += e * Vd[j * HD + c];
            }
        }
        const float il = 1.0f / l, a = an * il;
        float od[HD];
#pragma unroll
        for (int c = 0; c < HD; c++) od[c] = (o2[c] - a * o1[c] + o3[c]) * il;
        st16(AOd + row * D + HD * head, od, 1.f);
    }
}

// joint reverse: (lo, no) = (lambda, nu) at the attention output [R,D] -> (lQKV, nQKV) [R,3D]
__global__ __launch_bounds__(64) void k_attn_rev(const float* __restrict__ QKV, const float* __restrict__ QKVd,
                                                 const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                 const float* __restrict__ Tkb, const float* __restrict__ LO,
                                                 const float* __restrict__ NO, float* __restrict__ lQKV,
                                                 float* __restrict__ nQKV, int64_t E, int N, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int atom = blockIdx.x / NHEAD, head = blockIdx.x % NHEAD;
    const int start = rowptr[atom];
    const int T = rowptr[atom + 1] - start + 1;
    float* Ks = smem;            // K, V, K', V'
    float* Vs = Ks + T * HD;
    float* Kd = Vs + T * HD;
    float* Vd = Kd + T * HD;
    float* Qs = Vd + T * HD;     // scale*Q, scale*Q', lambda_o, nu_o (rows staged for pass B)
    float* Qd = Qs + T * HD;
    float* Ls = Qd + T * HD;
    float* Ns = Ls + T * HD;
    float* bs = Ns + T * HD;     // [T] bias, [T] bias tangent
    float* bd = bs + T;
    float* st = bd + T;          // [T][6] row statistics: max, 1/l, a, delta, c, e
    stage_tokens(Ks, QKV, D, T, E, atom, start, head, 1.f);
    stage_tokens(Vs, QKV, 2 * D, T, E, atom, start, head, 1.f);
    stage_tokens(Kd, QKVd, D, T, E, atom, start, head, 1.f);
    stage_tokens(Vd, QKVd, 2 * D, T, E, atom, start, head, 1.f);
    stage_tokens(Qs, QKV, 0, T, E, atom, start, head, scale);
    stage_tokens(Qd, QKVd, 0, T, E, atom, start, head, scale);
    for (int idx = threadIdx.x; idx < T * 4; idx += 64) {
        const int t = idx >> 2, q = idx & 3;
        const int64_t row = tok_row(t, E, atom, start);
        *reinterpret_cast<float4*>(Ls + t * HD + 4 * q) = *reinterpret_cast<const float4*>(LO + row * D + HD * head + 4 * q);
        *reinterpret_cast<float4*>(Ns + t * HD + 4 * q) = *reinterpret_cast<const float4*>(NO + row * D + HD * head + 4 * q);
    }
    for (int t = threadIdx.x; t < T; t += 64) {
        bs[t] = t == 0 ? 0.f : logf(fmaxf(fc[start + t - 1], 1e-15f));
        bd[t] = t == 0 ? 0.f : Tkb[start + t - 1];
    }
    __syncthreads();
    // ---- pass A: one thread per query row i -> row statistics, lambda_q, nu_q
    for (int i = threadIdx.x; i < T; i += 64) {
        float q[HD], qd[HD], lo[HD], no[HD];
#pragma unroll
        for (int c = 0; c < HD; c++) {
            q[c] = Qs[i * HD + c]; qd[c] = Qd[i * HD + c]; lo[c] = Ls[i * HD + c]; no[c] = Ns[i * HD + c];
        }
        float mx = -INFINITY;
        for (int j = 0; j < T; j++) mx = fmaxf(mx, dot16(q, Ks + j * HD) + bs[j]);
        float l = 0.f, an = 0.f, dn = 0.f, cn = 0.f, en = 0.f;
        for (int j = 0; j < T; j++) {
            const float e = expf(dot16(q, Ks + j * HD) + bs[j] - mx);
            const float sd = dot16(qd, Ks + j * HD) + dot16(q, Kd + j * HD) + bd[j];
            const float lp = dot16(lo, Vs + j * HD);
            const float np = dot16(no, Vs + j * HD) + dot16(lo, Vd + j * HD);
            l += e; an += e * sd; dn += e * lp; cn += e * lp * sd; en += e * np;
        }
        const float il = 1.0f / l, a = an * il, de = dn * il, cc = cn * il, ee = en * il;
        st[i * 6] = mx; st[i * 6 + 1] = il; st[i * 6 + 2] = a; st[i * 6 + 3] = de; st[i * 6 + 4] = cc; st[i * 6 + 5] = ee;
        float lq[HD], nq[HD];
#pragma unroll
        for (int c = 0; c < HD; c++) lq[c] = nq[c] = 0.f;
        for (int j = 0; j < T; j++) {
            const float P = expf(dot16(q, Ks + j * HD) + bs[j] - mx) * il;
            const float sd = dot16(qd, Ks + j * HD) + dot16(q, Kd + j * HD) + bd[j];
            const float lp = dot16(lo, Vs + j * HD);
            const float np = dot16(no, Vs + j * HD) + dot16(lo, Vd + j * HD);
            const float ls = P * (lp - de);
            const float ns = P * (np - ee) + P * ((lp - de) * (sd - a) - (cc - a * de));
#pragma unroll
            for (int c = 0; c < HD; c++) {
                lq[c] += ls * Ks[j * HD + c];
                nq[c] += ns * Ks[j * HD + c] + ls * Kd[j * HD + c];
            }
        }
        const int64_t row = tok_row(i, E, atom, start);
        st16(lQKV + row * (3 * D) + HD * head, lq, scale);
        st16(nQKV + row * (3 * D) + HD * head, nq, scale);
    }
    __syncthreads();
    // ---- pass B: one thread per key row j -> lambda_k, nu_k, lambda_v, nu_v
    for (int j = threadIdx.x; j < T; j += 64) {
        float k[HD], kd[HD], v[HD], vd[HD];
#pragma unroll
        for (int c = 0; c < HD; c++) {
            k[c] = Ks[j * HD + c]; kd[c] = Kd[j * HD + c]; v[c] = Vs[j * HD + c]; vd[c] = Vd[j * HD + c];
        }
        const float bj = bs[j], bdj = bd[j];
        float lk[HD], nk[HD], lv[HD], nv[HD];
#pragma unroll
        for (int c = 0; c < HD; c++) lk[c] = nk[c] = lv[c] = nv[c] = 0.f;
        for (int i = 0; i < T; i++) {
            const float mx = st[i * 6], il = st[i * 6 + 1], a = st[i * 6 + 2], de = st[i * 6 + 3], cc = st[i * 6 + 4],
                        ee = st[i * 6 + 5];
            const float P = expf(dot16(k, Qs + i * HD) + bj - mx) * il;
            const float sd = dot16(k, Qd + i * HD) + dot16(kd, Qs + i * HD) + bdj;
            const float lp = dot16(v, Ls + i * HD);
            const float np = dot16(v, Ns + i * HD) + dot16(vd, Ls + i * HD);
            const float ls = P * (lp - de);
            const float ns = P * (np - ee) + P * ((lp - de) * (sd - a) - (cc - a * de));
            const float Pd = P * (sd - a);
#pragma unroll
            for (int c = 0; c < HD; c++) {
                const float qs = Qs[i * HD + c], lo = Ls[i * HD + c];
                lk[c] += ls * qs;                        // Qs already carries `scale`
                nk[c] += ns * qs + ls * Qd[i * HD + c];
                lv[c] += P * lo;
                nv[c] += P * Ns[i * HD + c] + Pd * lo;
            }
        }
        const int64_t row = tok_row(j, E, atom, start);
        st16(lQKV + row * (3 * D) + D + HD * head, lk, 1.f);
        st16(nQKV + row * (3 * D) + D + HD * head, nk, 1.f);
        st16(lQKV + row * (3 * D) + 2 * D + HD * head, lv, 1.f);
        st16(nQKV + row * (3 * D) + 2 * D + HD * head, nv, 1.f);
    }
}

// ---------------------------------------------------------------------------------------------
// last layers: pred = wl . s2 + bl (edges: times fc, summed per atom).  32 lanes per row.
//   tangent:  T_pred = wl . s2'  ;  atomic' contributions  node: T_pred,  edge: fc' pred + fc T_pred
//   reverse:  (l_ep, n_ep) = (lA fc, nA fc + lA fc')  [node: (lA, nA)]
//             l_s2 = l_ep wl,  n_s2 = n_ep wl,   G = n_ep s2 + l_ep s2'  (column sums -> d wl),  n_ep -> d bl
// ---------------------------------------------------------------------------------------------
__global__ void k_last_layer(const float* __restrict__ S2, const float* __restrict__ S2d, const float* __restrict__ wl,
                             float bl, const float* __restrict__ lA, const float* __restrict__ nA,
                             const int* __restrict__ ctr, const float* __restrict__ fc, const float* __restrict__ Tfc,
                             float* __restrict__ Ls2, float* __restrict__ Ns2, float* __restrict__ G,
                             float* __restrict__ nep_out, float* __restrict__ tan_out, int64_t R) {
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int c = threadIdx.x & 31;
    const bool valid = row < R;
    float4 s = make_float4(0, 0, 0, 0), sd = s;
    const float4 w4 = *reinterpret_cast<const float4*>(wl + 4 * c);
    if (valid) {
        s = *reinterpret_cast<const float4*>(S2 + row * DH + 4 * c);
        sd = *reinterpret_cast<const float4*>(S2d + row * DH + 4 * c);
    }
    const float pred = group_sum<32>(dot4(s, w4)) + bl;
    const float tpred = group_sum<32>(dot4(sd, w4));
    if (!valid) return;
    float lep, nep, tan;
    if (ctr) {
        const int i = ctr[row];
        const float f = fc[row], fd = Tfc[row];
        lep = lA[i] * f;
        nep = (nA ? nA[i] * f : 0.f) + lA[i] * fd;
        tan = fd * pred + f * tpred;
    } else {
        lep = lA[row];
        nep = nA ? nA[row] : 0.f;
        tan = tpred;
    }
    *reinterpret_cast<float4*>(Ls2 + row * DH + 4 * c) = make_float4(lep * w4.x, lep * w4.y, lep * w4.z, lep * w4.w);
    *reinterpret_cast<float4*>(Ns2 + row * DH + 4 * c) = make_float4(nep * w4.x, nep * w4.y, nep * w4.z, nep * w4.w);
    *reinterpret_cast<float4*>(G + row * DH + 4 * c) =
        make_float4(nep * s.x + lep * sd.x, nep * s.y + lep * sd.y, nep * s.z + lep * sd.z, nep * s.w + lep * sd.w);
    if (c == 0) {
        nep_out[row] = nep;
        tan_out[row] = tan;
    }
}
// atomic'[i] = node'[i] + sum_{p in row i} edge'[p]
__global__ void k_tangent_atom_sum(const float* __restrict__ tnode, const float* __restrict__ tedge,
                                   const int* __restrict__ rowptr, float* __restrict__ out, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float s = tnode[i];
    for (int p = rowptr[i]; p < rowptr[i + 1]; p++) s += tedge[p];
    out[i] = s;
}

// ---------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------
struct SoAttn {
    float *TX, *Txh, *TQKV, *TAO, *TX1, *Tx1h, *TVG, *TU, *TH, *TOC, *TH1, *Th1h, *TVGn, *TUn, *THn;
};
struct SoGnn {
    std::vector<SoAttn> attn;
    float *Ta0, *Ts0, *TXF, *Tcath, *TCA, *TS, *TMout;
};
struct SoHead {
    float *a1, *s1, *a2, *s2, *Ta1, *Ts1, *Ta2, *Ts2;  // [rows, DH]
};
struct SoWs {
    std::vector<SoGnn> gnn;
    float *Tgeo, *Tfc, *Tkb, *TH0, *TM0;
    SoHead he, hn;
    float *tan_e, *tan_n, *nep;  // [E], [N], [max(E,N)]
    float *LM, *NM, *LX, *NX, *LH, *NH;
    float* tmp[6];
    size_t bytes = 0;
};

static void carve_so(const Model& m, int64_t N, int64_t E, void* base, SoWs& s) {
    Carver c(base);
    const int64_t R = E + N;
    const int64_t Ea = E > 0 ? E : 1, Na = N > 0 ? N : 1, Ra = R;
    s.Tgeo = c.take<float>(Ea * 4);
    s.Tfc = c.take<float>(Ea);
    s.Tkb = c.take<float>(Ea);
    s.TH0 = c.take<float>(Na * DN);
    s.TM0 = c.take<float>(Ea * D);
    s.gnn.resize(m.h.num_gnn_layers);
    float* prevH = s.TH0;
    for (auto& G : s.gnn) {
        G.attn.resize(m.h.num_attention_layers);
        G.Ta0 = c.take<float>(Ea * D);
        G.Ts0 = c.take<float>(Ea * D);
        for (auto& A : G.attn) {
            A.TX = c.take<float>(Ra * D);
            A.Txh = c.take<float>(Ra * D);
            A.TQKV = c.take<float>(Ra * 3 * D);
            A.TAO = c.take<float>(Ra * D);
            A.TX1 = c.take<float>(Ea * D);
            A.Tx1h = c.take<float>(Ea * D);
            A.TVG = c.take<float>(Ea * 2 * DFF);
            A.TU = c.take<float>(Ea * DFF);
            A.TH = prevH;
            A.TOC = c.take<float>(Na * D);
            A.TH1 = c.take<float>(Na * DN);
            A.Th1h = c.take<float>(Na * DN);
            A.TVGn = c.take<float>(Na * 2 * DNF);
            A.TUn = c.take<float>(Na * DNF);
            A.THn = c.take<float>(Na * DN);
            prevH = A.THn;
        }
        G.TXF = c.take<float>(Ea * D);
        G.Tcath = c.take<float>(Ea * 2 * D);
        G.TCA = c.take<float>(Ea * 2 * D);
        G.TS = c.take<float>(Ea * 2 * D);
        G.TMout = c.take<float>(Ea * D);
    }
    auto head = [&](SoHead& h, int64_t rows) {
        h.a1 = c.take<float>(rows * DH); h.s1 = c.take<float>(rows * DH);
        h.a2 = c.take<float>(rows * DH); h.s2 = c.take<float>(rows * DH);
        h.Ta1 = c.take<float>(rows * DH); h.Ts1 = c.take<float>(rows * DH);
        h.Ta2 = c.take<float>(rows * DH); h.Ts2 = c.take<float>(rows * DH);
    };
    head(s.he, Ea);
    head(s.hn, Na);
    s.tan_e = c.take<float>(Ea);
    s.tan_n = c.take<float>(Na);
    s.nep = c.take<float>(Ea > Na ? Ea : Na);
    s.LM = c.take<float>(Ea * D); s.NM = c.take<float>(Ea * D);
    s.LX = c.take<float>(Ra * D); s.NX = c.take<float>(Ra * D);
    s.LH = c.take<float>(Na * DN); s.NH = c.take<float>(Na * DN);
    int64_t big = Ea * 2 * DFF;
    if (Na * 2 * DNF > big) big = Na * 2 * DNF;
    if (Ra * 3 * D > big) big = Ra * 3 * D;
    for (auto& t : s.tmp) t = c.take<float>(big);
    s.bytes = c.off;
}

int64_t so_workspace_bytes(const Model& m, int64_t N, int64_t E) {
    if (train_generic(m)) return gen_train_workspace_bytes(m, N, E);
    SoWs s;
    carve_so(m, N, E, nullptr, s);
    return (int64_t)s.bytes;
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
static inline int grid1(int64_t n) { return cdiv(n, 256); }

static void rms_jvp(const Ctx& c, int K, const float* X, const float* Xd, float* XHd, int64_t R) {
    if (R <= 0) return;
    const bool ln = c.m.layer_norm();
    if (K == 128 && ln) k_rms_jvp<128, true><<<cdiv(R, 8), 256, 0, c.st>>>(X, Xd, XHd, R);
    else if (K == 128) k_rms_jvp<128, false><<<cdiv(R, 8), 256, 0, c.st>>>(X, Xd, XHd, R);
    else if (ln) k_rms_jvp<256, true><<<cdiv(R, 4), 256, 0, c.st>>>(X, Xd, XHd, R);
    else k_rms_jvp<256, false><<<cdiv(R, 4), 256, 0, c.st>>>(X, Xd, XHd, R);
}
static void rms_rev(const Ctx& c, int K, const float* X, const float* Xd, const float* ll, const float* ln,
                    const float* gamma, float* LX, float* NX, int64_t R) {
    if (R <= 0) return;
    const bool lnorm = c.m.layer_norm();
    if (K == 128 && lnorm) k_rms_rev<128, true><<<cdiv(R, 8), 256, 0, c.st>>>(X, Xd, ll, ln, gamma, LX, NX, R);
    else if (K == 128) k_rms_rev<128, false><<<cdiv(R, 8), 256, 0, c.st>>>(X, Xd, ll, ln, gamma, LX, NX, R);
    else if (lnorm) k_rms_rev<256, true><<<cdiv(R, 4), 256, 0, c.st>>>(X, Xd, ll, ln, gamma, LX, NX, R);
    else k_rms_rev<256, false><<<cdiv(R, 4), 256, 0, c.st>>>(X, Xd, ll, ln, gamma, LX, NX, R);
}

// heads: primal recompute + tangent (forward part)
static void head_tangent(const Ctx& c, const Lin& h0, const Lin& h2, const float* Xin, const float* TXin, SoHead& h,
                         int64_t rows) {
    if (rows <= 0) return;
    mm_fwd(c, h0, Xin, h.a1, rows, true);
    mm_fwd(c, h0, TXin, h.Ta1, rows, false);
    k_silu_jvp<<<grid1(rows * DH), 256, 0, c.st>>>(h.a1, h.Ta1, h.s1, h.Ts1, rows * DH);
    mm_fwd(c, h2, h.s1, h.a2, rows, true);
    mm_fwd(c, h2, h.Ts1, h.Ta2, rows, false);
    k_silu_jvp<<<grid1(rows * DH), 256, 0, c.st>>>(h.a2, h.Ta2, h.s2, h.Ts2, rows * DH);
}

// heads: joint reverse. On return (Lout, Nout) [rows, k_in] hold the adjoints of the head input.
static void head_reverse(const Ctx& c, Trainer& tr, SoWs& s, bool edge, const Lin& h0, const Lin& h2, const float* wl,
                         float bl, const float* Xin, const float* TXin, int k_in, SoHead& h, const float* lA,
                         const float* nA, float* tan_out, float* Lout, float* Nout, int64_t rows) {
    if (rows <= 0) return;
    const std::string hk = edge ? "edge_heads.@.0" : "node_heads.@.0";
    const std::string lk = edge ? "edge_last_layers.@.0.@" : "node_last_layers.@.0.@";
    float *l2 = s.tmp[0], *n2 = s.tmp[1], *G = s.tmp[2], *l1 = s.tmp[3], *n1 = s.tmp[4];
    k_last_layer<<<cdiv(rows, 8), 256, 0, c.st>>>(h.s2, h.Ts2, wl, bl, lA, nA, edge ? c.g.ctr : nullptr, c.g.fc, s.Tfc,
                                                  l2, n2, G, s.nep, tan_out, rows);
    tr.colsum(G, rows, DH, tr.gp(lk + ".weight"));
    tr.vecsum(s.nep, rows, tr.gp(lk + ".bias"));
    k_silu_rev<<<grid1(rows * DH), 256, 0, c.st>>>(h.a2, h.Ta2, l2, n2, rows * DH);  // -> (l_a2, n_a2)
    tr.linear(hk + ".2", DH, DH, {n2, nullptr, 0, DH}, {h.s1, DH, 0, nullptr, nullptr}, 0, rows);
    tr.linear(hk + ".2", DH, DH, {l2, nullptr, 0, DH}, {h.Ts1, DH, 0, nullptr, nullptr}, 0, rows, false);
    mm_bwd(c, h2, l2, l1, rows);
    mm_bwd(c, h2, n2, n1, rows);
    k_silu_rev<<<grid1(rows * DH), 256, 0, c.st>>>(h.a1, h.Ta1, l1, n1, rows * DH);  // -> (l_a1, n_a1)
    tr.linear(hk + ".0", DH, k_in, {n1, nullptr, 0, DH}, {Xin, k_in, 0, nullptr, nullptr}, 0, rows);
    tr.linear(hk + ".0", DH, k_in, {l1, nullptr, 0, DH}, {TXin, k_in, 0, nullptr, nullptr}, 0, rows, false);
    mm_bwd(c, h0, l1, Lout, rows);
    mm_bwd(c, h0, n1, Nout, rows);
}

// ---------------------------------------------------------------------------------------------
// driver
// ---------------------------------------------------------------------------------------------
int backward_train2(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, void* ws2, int64_t ws2_bytes,
                    const float* lA, const float* nA, const float* u, float* tangent_atomic, hipStream_t st,
                    const float* ucell) {
    // other sizes, PostLN, residual, more than 127 neighbours per atom: gen_train.hip (it recomputes what it needs in ws2)
    if (train_generic_for(m, g)) return gen_train2(m, g, ws2, ws2_bytes, lA, nA, u, ucell, tangent_atomic, st);
    PET_REQUIRE(m.grad_flat, PET_ERR_ARGUMENT, "pet_model_zero_grad has not been called");
    PET_REQUIRE(m.trainable(), PET_ERR_UNSUPPORTED,
                "training is built for transformer_type=PreLN, featurizer_type=feedforward only");
    Workspace w;
    carve_workspace(m, g.n_nodes, g.n_edges, ws, w, true);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "workspace too small for training");
    SoWs s;
    carve_so(m, g.n_nodes, g.n_edges, ws2, s);
    PET_REQUIRE((int64_t)s.bytes <= ws2_bytes, PET_ERR_ARGUMENT, "second-order workspace too small");
    const int64_t N = g.n_nodes, E = g.n_edges, R = E + N;
    if (N == 0) return PET_OK;
    PET_REQUIRE(E > 0, PET_ERR_UNSUPPORTED, "training on a batch without any edge is not supported");
    PET_REQUIRE(g.max_nbr + 1 <= 128, PET_ERR_UNSUPPORTED, "more than 127 neighbours per atom is not supported yet");

    const float scale = 1.0f / (sqrtf((float)HD) * m.h.attention_temperature);
    const int T_max = g.max_nbr + 1;
    const int nt_attn = (T_max + 15) / 16;
    const size_t lds_jvp = (size_t)T_max * (4 * HD + 2) * sizeof(float);
    const size_t lds_rev = (size_t)T_max * (8 * HD + 2 + 6) * sizeof(float);
    allow_big_lds(k_attn_rev, lds_rev);
    const Ctx c{m, g, st};
    Trainer tr{m, g, w, m.grad_flat, st};
    const int nG = m.h.num_gnn_layers, nA_ = m.h.num_attention_layers;

    // =========================== tangent sweep ===========================
    if (g.adaptive) adaptive_rdot(m, g, u, ucell, st);  // g.ad_gr doubles as the tangent of the atomic cutoffs
    k_geom_jvp<<<grid1(E), 256, 0, st>>>(u, g.ctr, g.nbr, g.geo, g.d0, g.fc, reinterpret_cast<float4*>(s.Tgeo), s.Tfc,
                                        s.Tkb, E, m.h.cutoff, m.h.cutoff_width, m.h.cutoff_function,
                                        g.adaptive ? g.pc : nullptr, g.adaptive ? g.ad_gr : nullptr, ucell, g.shift, g.sys);
    PET_HIP_CHECK(hipMemsetAsync(s.TH0, 0, N * DN * sizeof(float), st));  // embeddings do not move with R
    PET_HIP_CHECK(hipMemsetAsync(s.TM0, 0, E * D * sizeof(float), st));
    for (int gi = 0; gi < nG; gi++) {
        const GnnLayerW& G = m.gnn[gi];
        const GnnBufs& B = w.gnn[gi];
        SoGnn& S = s.gnn[gi];
        const float* TMin = gi == 0 ? s.TM0 : s.gnn[gi - 1].TMout;
        k_geo_lin<<<grid1(E * D), 256, 0, st>>>(reinterpret_cast<const float4*>(s.Tgeo), G.wc, S.Ta0, E);
        if (gi > 0) mm_fwd(c, G.compress0_msg, TMin, S.Ta0, E, false, nullptr, true);
        k_silu_jvp<<<grid1(E * D), 256, 0, st>>>(B.a0, S.Ta0, nullptr, S.Ts0, E * D);
        mm_fwd(c, G.compress2, S.Ts0, S.attn[0].TX, E, false);
        for (int a = 0; a < nA_; a++) {
            const AttnLayerW& A = G.attn[a];
            const AttnBufs& Ab = B.attn[a];
            SoAttn& Sa = S.attn[a];
            float* TXnext = a + 1 < nA_ ? S.attn[a + 1].TX : S.TXF;
            mm_fwd(c, A.cc, Sa.TH, Sa.TX + E * D, N, false);                 // centre tokens
            rms_jvp(c, D, Ab.X, Sa.TX, Sa.Txh, R);
            mm_fwd(c, A.qkv, Sa.Txh, Sa.TQKV, R, false, A.g_attn);
            {
                ProfScope psj("so_attn_jvp", st, 0.0);
                if (!attn_jvp_mfma(nt_attn, Ab.QKV, Sa.TQKV, g, s.Tkb, Sa.TAO, scale, st))
                    k_attn_jvp<<<(int)N * NHEAD, 64, lds_jvp, st>>>(Ab.QKV, Sa.TQKV, g.rowptr, g.fc, s.Tkb, Sa.TAO, E,
                                                                   (int)N, scale);
            }
            float* TO = s.tmp[0];                                            // [R,D] output_linear tangent
            mm_fwd(c, A.out, Sa.TAO, TO, R, false);
            add3(c, Sa.TX, TO, nullptr, Sa.TX1, E * D);                      // edge residual
            PET_HIP_CHECK(hipMemcpyAsync(Sa.TOC, TO + E * D, N * D * sizeof(float), hipMemcpyDeviceToDevice, st));
            // node chain
            PET_HIP_CHECK(hipMemcpyAsync(Sa.TH1, Sa.TH, N * DN * sizeof(float), hipMemcpyDeviceToDevice, st));
            mm_fwd(c, A.ce, Sa.TOC, Sa.TH1, N, false, nullptr, true);
            rms_jvp(c, DN, Ab.H1, Sa.TH1, Sa.Th1h, N);
            mm_fwd(c, A.cmlp_in, Sa.Th1h, Sa.TVGn, N, false, A.g_center);
            k_swiglu_jvp<<<grid1(N * DNF), 256, 0, st>>>(Ab.VGn, Sa.TVGn, Sa.TUn, N, DNF);
            PET_HIP_CHECK(hipMemcpyAsync(Sa.THn, Sa.TH1, N * DN * sizeof(float), hipMemcpyDeviceToDevice, st));
            mm_fwd(c, A.cmlp_out, Sa.TUn, Sa.THn, N, false, nullptr, true);
            // edge MLP
            rms_jvp(c, D, Ab.X1, Sa.TX1, Sa.Tx1h, E);
            mm_fwd(c, A.mlp_in, Sa.Tx1h, Sa.TVG, E, false, A.g_mlp);
            k_swiglu_jvp<<<grid1(E * DFF), 256, 0, st>>>(Ab.VG, Sa.TVG, Sa.TU, E, DFF);
            PET_HIP_CHECK(hipMemcpyAsync(TXnext, Sa.TX1, E * D * sizeof(float), hipMemcpyDeviceToDevice, st));
            mm_fwd(c, A.mlp_out, Sa.TU, TXnext, E, false, nullptr, true);
        }
        k_lncat_jvp<<<cdiv(E, 4), 256, 0, st>>>(B.XF, S.TXF, g.rev, B.LNS, S.Tcath, E);
        mm_fwd(c, G.comb0, S.Tcath, S.TCA, E, false, G.ln_g);
        k_silu_jvp<<<grid1(E * 2 * D), 256, 0, st>>>(B.CA, S.TCA, nullptr, S.TS, E * 2 * D);
        add3(c, TMin, S.TXF, nullptr, S.TMout, E * D);
        mm_fwd(c, G.comb2, S.TS, S.TMout, E, false, nullptr, true);
    }
    const GnnBufs& last = w.gnn.back();
    const float* THlast = s.gnn.back().attn.back().THn;
    const float* TMlast = s.gnn.back().TMout;
    head_tangent(c, m.eh0, m.eh2, last.Mout, TMlast, s.he, E);
    head_tangent(c, m.nh0, m.nh2, last.Hout, THlast, s.hn, N);

    // =========================== joint reverse sweep ===========================
    head_reverse(c, tr, s, true, m.eh0, m.eh2, m.ell_w, m.ell_b, last.Mout, TMlast, D, s.he, lA, nA, s.tan_e, s.LM, s.NM,
                 E);
    head_reverse(c, tr, s, false, m.nh0, m.nh2, m.nll_w, m.nll_b, last.Hout, THlast, DN, s.hn, lA, nA, s.tan_n, s.LH,
                 s.NH, N);
    if (tangent_atomic) k_tangent_atom_sum<<<grid1(N), 256, 0, st>>>(s.tan_n, s.tan_e, g.rowptr, tangent_atomic, (int)N);
    float *t0 = s.tmp[0], *t1 = s.tmp[1], *t2 = s.tmp[2], *t3 = s.tmp[3], *t4 = s.tmp[4], *t5 = s.tmp[5];
    const bool lnm = m.layer_norm();
    const int nx = lnm ? 5 : 1;  // weight-gradient row source: LayerNorm-hat / RMSNorm-hat of the saved input
    for (int gi = nG - 1; gi >= 0; gi--) {
        const GnnLayerW& G = m.gnn[gi];
        const GnnBufs& B = w.gnn[gi];
        SoGnn& S = s.gnn[gi];
        const std::string gs = std::to_string(gi);
        const std::string pre = "gnn_layers." + gs;
        const float* Min = gi == 0 ? nullptr : w.gnn[gi - 1].Mout;
        const float* TMin = gi == 0 ? s.TM0 : s.gnn[gi - 1].TMout;
        // system conditioning: the embedding enters additively where the node features leave the layer and has no
        // tangent, so its parameters see the second-order adjoint nu of those features only
        tr.cond_accumulate(s.NH, gi == nG - 1);
        // ---- M_out = M_in + XF + comb2(silu(comb0(LN(cat))))
        tr.linear("combination_mlps." + gs + ".2", D, 2 * D, {s.NM, nullptr, 0, D}, {B.CA, 2 * D, 0, nullptr, nullptr}, 3, E);
        tr.linear("combination_mlps." + gs + ".2", D, 2 * D, {s.LM, nullptr, 0, D}, {S.TS, 2 * D, 0, nullptr, nullptr}, 0, E,
                  false);
        mm_bwd(c, G.comb2, s.LM, t0, E);  // l_S, n_S [E,2D]
        mm_bwd(c, G.comb2, s.NM, t1, E);
        k_silu_rev<<<grid1(E * 2 * D), 256, 0, st>>>(B.CA, S.TCA, t0, t1, E * 2 * D);  // -> (l_CA, n_CA)
        tr.linear_after_norm("combination_mlps." + gs + ".0", G.comb0.w, 2 * D, 2 * D, {t1, nullptr, 0, 2 * D},
                             {B.XF, D, 0, g.rev, B.LNS}, 4, E, "combination_norms." + gs + ".weight", G.ln_g,
                             "combination_norms." + gs + ".bias", G.ln_b);
        tr.linear_after_norm("combination_mlps." + gs + ".0", G.comb0.w, 2 * D, 2 * D, {t0, nullptr, 0, 2 * D},
                             {S.Tcath, 2 * D, 0, nullptr, nullptr}, 0, E, "combination_norms." + gs + ".weight", G.ln_g,
                             "combination_norms." + gs + ".bias", G.ln_b, true);
        mm_bwd(c, G.comb0, t0, t2, E);  // adjoints w.r.t. the LayerNorm output [E,2D]
        mm_bwd(c, G.comb0, t1, t3, E);
        k_lncat_rev<<<cdiv(E, 4), 256, 0, st>>>(B.XF, S.TXF, g.rev, B.LNS, t2, t3, G.ln_g, t4, t5, E);
        k_cat_gather<<<grid1(E * (D / 4)), 256, 0, st>>>(s.LM, t4, g.rev, s.LX, E);
        k_cat_gather<<<grid1(E * (D / 4)), 256, 0, st>>>(s.NM, t5, g.rev, s.NX, E);
        for (int a = nA_ - 1; a >= 0; a--) {
            const AttnLayerW& A = G.attn[a];
            const AttnBufs& Ab = B.attn[a];
            SoAttn& Sa = S.attn[a];
            const std::string lp = pre + ".trans.layers." + std::to_string(a);
            // ---- edge MLP: Xnext = X1 + w_out(swiglu(w_in(rms(X1))))
            tr.linear(lp + ".mlp.w_out", D, DFF, {s.NX, nullptr, 0, D}, {Ab.VG, 2 * DFF, DFF, nullptr, nullptr}, 2, E);
            tr.linear(lp + ".mlp.w_out", D, DFF, {s.LX, nullptr, 0, D}, {Sa.TU, DFF, 0, nullptr, nullptr}, 0, E, false);
            mm_bwd(c, A.mlp_out, s.LX, t0, E);  // l_U, n_U [E,DFF]
            mm_bwd(c, A.mlp_out, s.NX, t1, E);
            k_swiglu_rev<<<grid1(E * DFF), 256, 0, st>>>(Ab.VG, Sa.TVG, t0, t1, t2, t3, E, DFF);  // (l_VG, n_VG)
            tr.linear_after_norm(lp + ".mlp.w_in", A.mlp_in.w, 2 * DFF, D, {t3, nullptr, 0, 2 * DFF},
                                 {Ab.X1, D, 0, nullptr, nullptr}, nx, E, lp + ".norm_mlp.weight", A.g_mlp,
                                 lnm ? lp + ".norm_mlp.bias" : std::string(), lnm ? A.b_mlp : nullptr);
            tr.linear_after_norm(lp + ".mlp.w_in", A.mlp_in.w, 2 * DFF, D, {t2, nullptr, 0, 2 * DFF},
                                 {Sa.Tx1h, D, 0, nullptr, nullptr}, 0, E, lp + ".norm_mlp.weight", A.g_mlp, "", nullptr,
                                 true);
            mm_bwd(c, A.mlp_in, t2, t0, E);
            mm_bwd(c, A.mlp_in, t3, t1, E);
            rms_rev(c, D, Ab.X1, Sa.TX1, t0, t1, A.g_mlp, s.LX, s.NX, E);  // LX/NX[edge] = adjoints of X1
            // ---- node chain: Hn = H1 + w_out(swiglu(w_in(rms(H1))))
            tr.linear(lp + ".center_mlp.w_out", DN, DNF, {s.NH, nullptr, 0, DN}, {Ab.VGn, 2 * DNF, DNF, nullptr, nullptr}, 2,
                      N);
            tr.linear(lp + ".center_mlp.w_out", DN, DNF, {s.LH, nullptr, 0, DN}, {Sa.TUn, DNF, 0, nullptr, nullptr}, 0, N,
                      false);
            mm_bwd(c, A.cmlp_out, s.LH, t0, N);
            mm_bwd(c, A.cmlp_out, s.NH, t1, N);
            k_swiglu_rev<<<grid1(N * DNF), 256, 0, st>>>(Ab.VGn, Sa.TVGn, t0, t1, t2, t3, N, DNF);
            tr.linear_after_norm(lp + ".center_mlp.w_in", A.cmlp_in.w, 2 * DNF, DN, {t3, nullptr, 0, 2 * DNF},
                                 {Ab.H1, DN, 0, nullptr, nullptr}, nx, N, lp + ".norm_center_features.weight", A.g_center,
                                 lnm ? lp + ".norm_center_features.bias" : std::string(), lnm ? A.b_center : nullptr);
            tr.linear_after_norm(lp + ".center_mlp.w_in", A.cmlp_in.w, 2 * DNF, DN, {t2, nullptr, 0, 2 * DNF},
                                 {Sa.Th1h, DN, 0, nullptr, nullptr}, 0, N, lp + ".norm_center_features.weight",
                                 A.g_center, "", nullptr, true);
            mm_bwd(c, A.cmlp_in, t2, t0, N);
            mm_bwd(c, A.cmlp_in, t3, t1, N);
            rms_rev(c, DN, Ab.H1, Sa.TH1, t0, t1, A.g_center, s.LH, s.NH, N);  // LH/NH = adjoints of H1
            // ---- H1 = H + center_expansion(OC)
            tr.linear(lp + ".center_expansion", DN, D, {s.NH, nullptr, 0, DN}, {Ab.OC, D, 0, nullptr, nullptr}, 0, N);
            tr.linear(lp + ".center_expansion", DN, D, {s.LH, nullptr, 0, DN}, {Sa.TOC, D, 0, nullptr, nullptr}, 0, N, false);
            mm_bwd(c, A.ce, s.LH, s.LX + E * D, N);  // centre rows of the output_linear adjoint
            mm_bwd(c, A.ce, s.NH, s.NX + E * D, N);
            // ---- O = output_linear(AO) over all R tokens; (LX, NX) are its adjoints
            tr.linear(lp + ".attention.output_linear", D, D, {s.NX, nullptr, 0, D}, {Ab.AO, D, 0, nullptr, nullptr}, 0, R);
            tr.linear(lp + ".attention.output_linear", D, D, {s.LX, nullptr, 0, D}, {Sa.TAO, D, 0, nullptr, nullptr}, 0, R,
                      false);
            mm_bwd(c, A.out, s.LX, t0, R);  // l_AO, n_AO [R,D]
            mm_bwd(c, A.out, s.NX, t1, R);
            PET_HIP_CHECK(hipMemsetAsync(s.LX + E * D, 0, N * D * sizeof(float), st));  // centre tokens: norm path only
            PET_HIP_CHECK(hipMemsetAsync(s.NX + E * D, 0, N * D * sizeof(float), st));
            {   // (l_QKV, n_QKV) [R,3D]
                ProfScope psr("so_attn_rev", st, 0.0);
                if (!attn_rev_mfma(nt_attn, Ab.QKV, Sa.TQKV, g, s.Tkb, t0, t1, t2, t3, scale, st))
                    k_attn_rev<<<(int)N * NHEAD, 64, lds_rev, st>>>(Ab.QKV, Sa.TQKV, g.rowptr, g.fc, s.Tkb, t0, t1, t2,
                                                                   t3, E, (int)N, scale);
            }
            tr.linear_after_norm(lp + ".attention.input_linear", A.qkv.w, 3 * D, D, {t3, nullptr, 0, 3 * D},
                                 {Ab.X, D, 0, nullptr, nullptr}, nx, R, lp + ".norm_attention.weight", A.g_attn,
                                 lnm ? lp + ".norm_attention.bias" : std::string(), lnm ? A.b_attn : nullptr);
            tr.linear_after_norm(lp + ".attention.input_linear", A.qkv.w, 3 * D, D, {t2, nullptr, 0, 3 * D},
                                 {Sa.Txh, D, 0, nullptr, nullptr}, 0, R, lp + ".norm_attention.weight", A.g_attn, "",
                                 nullptr, true);
            mm_bwd(c, A.qkv, t2, t0, R);
            mm_bwd(c, A.qkv, t3, t1, R);
            rms_rev(c, D, Ab.X, Sa.TX, t0, t1, A.g_attn, s.LX, s.NX, R);  // adjoints of the tokens X
            // ---- centre token = center_contraction(H)
            tr.linear(lp + ".center_contraction", D, DN, {s.NX + E * D, nullptr, 0, D}, {Ab.H, DN, 0, nullptr, nullptr}, 0, N);
            tr.linear(lp + ".center_contraction", D, DN, {s.LX + E * D, nullptr, 0, D}, {Sa.TH, DN, 0, nullptr, nullptr}, 0,
                      N, false);
            mm_bwd(c, A.cc, s.LX + E * D, s.LH, N, true);
            mm_bwd(c, A.cc, s.NX + E * D, s.NH, N, true);
        }
        // ---- e = compress.2(silu(a0)),  a0 = geo Wc^T + Tbl[species] (+ M_in W0c^T)
        tr.linear(pre + ".compress.2", D, D, {s.NX, nullptr, 0, D}, {B.a0, D, 0, nullptr, nullptr}, 3, E);
        tr.linear(pre + ".compress.2", D, D, {s.LX, nullptr, 0, D}, {S.Ts0, D, 0, nullptr, nullptr}, 0, E, false);
        mm_bwd(c, G.compress2, s.LX, t0, E);
        mm_bwd(c, G.compress2, s.NX, t1, E);
        k_silu_rev<<<grid1(E * D), 256, 0, st>>>(B.a0, S.Ta0, t0, t1, E * D);  // (l_a0, n_a0)
        tr.compress0(gi, t1, Min, t0, reinterpret_cast<const float4*>(s.Tgeo), TMin);
        if (gi > 0) {
            mm_bwd(c, G.compress0_msg, t0, s.LM, E, true);
            mm_bwd(c, G.compress0_msg, t1, s.NM, E, true);
        }
        if (tr.err) return tr.err;
    }
    // embeddings: H0 = node_emb[species], M0 = edge_emb[neighbour species] (their tangents are zero)
    tr.species_rows(s.NH, g.sp, N, DN, tr.gp("node_embedders.0.weight"));
    tr.species_rows(s.NM, g.sp_nbr, E, D, tr.gp("edge_embedder.weight"));
    tr.cond_finish();
    PET_HIP_CHECK(hipGetLastError());
    return tr.err;
}

}  // namespace pet
