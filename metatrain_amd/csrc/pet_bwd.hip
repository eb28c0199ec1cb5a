// PET reverse pass (dL/dR) on gfx950. Hand-written adjoint of every stage in pet_fwd.hip;
// replaces torch.autograd.grad(E, positions) of utils/output_gradient.py:34-40 for the
// inference / force path (activation gradients only, no weight gradients).
//
// Every dense adjoint is a row-tile GEMM against the TRANSPOSED weight (Lin::bwd packs
// W^T in fragment order), so the same MFMA toolkit applies. Cross-atom terms are
// gathers, never float atomics:
//   * ji message gather   e_rev[p] = e[rev[p]]  ->  de[p] += dcat_hi[rev[p]]   (rev is an involution)
//   * dE/dR_a = sum_{p in row a} (dv[rev[p]] - dv[p])                          (v_p = r_j - r_i + S.cell)
// so the result is deterministic run to run.
#include "common.h"
#include "cutoff.h"
#include "model.h"
#include "pet_ws.h"
#include "train.h"
#include "tile.h"

namespace pet {

constexpr int LD128 = lds_ld(128);
constexpr int LD256 = lds_ld(256);

int attn_tiles(const Graph& g);
int exchange_backward(const Graph& g, float* dXF, int layer, hipStream_t st);  // pet_fwd.hip
double g_sum_t2(const Graph& g);

__device__ __forceinline__ float sigmoid_grad_from(float s) { return s * (1.0f - s); }

// RMSNorm adjoint on a tile: W holds w = gamma * dn (LDS [64][K+4]); x rows come from
// global. dx = rstd * (w - xhat * mean(w * xhat)), xhat = x * rstd.
// calls f(row_in_tile, col, dx) for the 4-thread-per-row decomposition.
template <int K, int ROWS = BM, class F>
__device__ __forceinline__ void rmsnorm_bwd_rows(const float* Wt, const float* __restrict__ Xg, int64_t row0,
                                                 int64_t n_rows, int ldx, F f) {
    constexpr int LDW = lds_ld(K), TPR = NTHREADS / ROWS;
    const int r = threadIdx.x / TPR, q = threadIdx.x % TPR;
    const bool valid = row0 + r < n_rows;
    const float* xrow = Xg + (row0 + r) * ldx;
    float ss = 0.f, dot = 0.f;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 x = valid ? *reinterpret_cast<const float4*>(xrow + c) : make_float4(0, 0, 0, 0);
        float4 wv = *reinterpret_cast<const float4*>(Wt + r * LDW + c);
        ss += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        dot += x.x * wv.x + x.y * wv.y + x.z * wv.z + x.w * wv.w;
    }
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); if (TPR == 8) ss += __shfl_xor(ss, 4);
    dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); if (TPR == 8) dot += __shfl_xor(dot, 4);
    const float rstd = rsqrtf(ss * (1.0f / K) + 1.1920928955078125e-07f);
    const float coef = dot * rstd * rstd * rstd * (1.0f / K);  // mean(w*xhat) * rstd / x-scale
    if (!valid) return;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 x = *reinterpret_cast<const float4*>(xrow + c);
        float4 wv = *reinterpret_cast<const float4*>(Wt + r * LDW + c);
        f(r, c, make_float4(rstd * wv.x - x.x * coef, rstd * wv.y - x.y * coef, rstd * wv.z - x.z * coef,
                            rstd * wv.w - x.w * coef));
    }
}

// The same with LayerNorm (ln): xhat = (x - mean) rstd, dx = rstd (w - mean(w) - xhat mean(w xhat)).
template <int K, int ROWS = BM, class F>
__device__ __forceinline__ void norm_bwd_rows(const float* Wt, const float* __restrict__ Xg, int64_t row0,
                                              int64_t n_rows, int ldx, bool ln, F f) {
    if (!ln) {
        rmsnorm_bwd_rows<K, ROWS>(Wt, Xg, row0, n_rows, ldx, f);
        return;
    }
    constexpr int LDW = lds_ld(K), TPR = NTHREADS / ROWS;
    const int r = threadIdx.x / TPR, q = threadIdx.x % TPR;
    const bool valid = row0 + r < n_rows;
    const float* xrow = Xg + (row0 + r) * ldx;
    float s = 0.f, sw = 0.f;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 x = valid ? *reinterpret_cast<const float4*>(xrow + c) : make_float4(0, 0, 0, 0);
        float4 wv = *reinterpret_cast<const float4*>(Wt + r * LDW + c);
        s += (x.x + x.y) + (x.z + x.w);
        sw += (wv.x + wv.y) + (wv.z + wv.w);
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); if (TPR == 8) s += __shfl_xor(s, 4);
    sw += __shfl_xor(sw, 1); sw += __shfl_xor(sw, 2); if (TPR == 8) sw += __shfl_xor(sw, 4);
    const float mean = s * (1.0f / K), mw = sw * (1.0f / K);
    float ss = 0.f, dot = 0.f;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 x = valid ? *reinterpret_cast<const float4*>(xrow + c) : make_float4(0, 0, 0, 0);
        float4 wv = *reinterpret_cast<const float4*>(Wt + r * LDW + c);
        x.x -= mean; x.y -= mean; x.z -= mean; x.w -= mean;
        ss += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        dot += x.x * wv.x + x.y * wv.y + x.z * wv.z + x.w * wv.w;
    }
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); if (TPR == 8) ss += __shfl_xor(ss, 4);
    dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); if (TPR == 8) dot += __shfl_xor(dot, 4);
    const float rstd = rsqrtf(ss * (1.0f / K) + 1e-5f);
    const float coef = dot * rstd * rstd * rstd * (1.0f / K);
    if (!valid) return;
    for (int c = q * 4; c < K; c += 4 * TPR) {
        float4 x = *reinterpret_cast<const float4*>(xrow + c);
        float4 wv = *reinterpret_cast<const float4*>(Wt + r * LDW + c);
        f(r, c, make_float4(rstd * (wv.x - mw) - (x.x - mean) * coef, rstd * (wv.y - mw) - (x.y - mean) * coef,
                            rstd * (wv.z - mw) - (x.z - mean) * coef, rstd * (wv.w - mw) - (x.w - mean) * coef));
    }
}

// ---------------------------------------------------------------------------------
// heads
// ---------------------------------------------------------------------------------
// GEN (pet_predict_backward: any number of properties, features given by the caller): the gradient w.r.t. the head's
// hidden row arrives as a per-atom row Gd[atom][DH] = sum_p gA[atom][p] Wl[p][:] (instead of gA[atom] * wl), and the
// cutoff-factor gradient is dfc = Gd[atom] . hidden + gb[atom] (gb = sum_p gA[atom][p] bl[p]) from the recomputed hidden row.
template <int K, bool EDGE, bool TRAIN, bool GEN = false>
__global__ __launch_bounds__(NTHREADS) void k_head_bwd(const float* __restrict__ Xin, WX w0f,
                                                        const float* __restrict__ b0, WX w2f,
                                                        const float* __restrict__ b2, WX w0b,
                                                        WX w2b, const float* __restrict__ wl,
                                                        const float* __restrict__ gA, const int* __restrict__ ctr,
                                                        const float* __restrict__ fc, const float* __restrict__ ypred,
                                                        float* __restrict__ dfc, float* __restrict__ dXout, int64_t R,
                                                        float* __restrict__ t_s1, float* __restrict__ t_da2,
                                                        float* __restrict__ t_da1, float* __restrict__ t_s2y,
                                                        const float* __restrict__ Gd = nullptr,
                                                        const float* __restrict__ gb = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDK = lds_ld(K);
    float* A = smem;
    float* S = smem + BM * LDK;
    float* gy = S + BM * LD128;  // [64]
    float* rs = gy + BM;         // [64][2] power-of-two row scales of the adjoint tiles (f16x3 GEMMs)
    int* at = reinterpret_cast<int*>(rs + 2 * BM);  // [64] GEN: atom of the row (Gd / gb row index)
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<K>(A, Xin, row0, R, K);
    if (threadIdx.x < BM) {
        const int64_t row = row0 + threadIdx.x;
        float g = 0.f;
        if (row < R) {
            if (GEN) {
                g = EDGE ? fc[row] : 1.0f;
                at[threadIdx.x] = EDGE ? ctr[row] : (int)row;
            } else if (EDGE) {
                const float ga = gA[ctr[row]];
                g = ga * fc[row];
                dfc[row] = ga * ypred[row];  // d(y fc)/dfc
            } else {
                g = gA[row];
            }
        } else if (GEN) {
            at[threadIdx.x] = 0;
        }
        gy[threadIdx.x] = g;
    }
    __syncthreads();
    if (w0f.h) {  // the forward kernel's row scales (k_head): un-normalised backbone features and hidden rows
        tile_row_scales<K>(A, LDK, rs);
        __syncthreads();
    }
    f32x16 a1[2], acc[2];
    acc_fill_bias<2>(a1, b0, 64 * w.ch, w.lane);
    gemm_acc_x<K, 2>(A + w.rb * 32 * LDK, LDK, w0f, K / 8, 0, 2 * w.ch, a1, w.lane, w0f.h ? rs + 64 * w.rb : nullptr);
    acc_foreach<2>(a1, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
        const float s1 = siluf_(v);
        S[r * LD128 + c] = s1;
        if (TRAIN && row0 + r < R) t_s1[(row0 + r) * DH + c] = s1;
    });
    __syncthreads();
    if (w2f.h) {
        tile_row_scales<128>(S, LD128, rs);
        __syncthreads();
    }
    acc_fill_bias<2>(acc, b2, 64 * w.ch, w.lane);
    gemm_acc_x<128, 2>(S + w.rb * 32 * LD128, LD128, w2f, 16, 0, 2 * w.ch, acc, w.lane, w2f.h ? rs + 64 * w.rb : nullptr);
    __syncthreads();
    if (GEN && EDGE) {  // dfc[row] = Gd[atom] . silu(a2) + gb[atom]
        acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
            S[r * LD128 + c] = siluf_(v) * Gd[(int64_t)at[r] * DH + c];
        });
        __syncthreads();
        {
            const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
            float sum = 0.f;
            for (int c = q * 4; c < 128; c += 16) {
                const float4 v = *reinterpret_cast<float4*>(S + r * LD128 + c);
                sum += v.x + v.y + v.z + v.w;
            }
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            if (q == 0 && row0 + r < R) dfc[row0 + r] = sum + gb[at[r]];
        }
        __syncthreads();
    }
    // da2 = gy * wl * silu'(a2)
    acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
        const float d2 = gy[r] * (GEN ? Gd[(int64_t)at[r] * DH + c] : wl[c]) * silu_grad_(v);
        S[r * LD128 + c] = d2;
        if (TRAIN && row0 + r < R) {
            t_da2[(row0 + r) * DH + c] = d2;
            t_s2y[(row0 + r) * DH + c] = gy[r] * siluf_(v);
        }
    });
    __syncthreads();
    acc_fill_bias<2>(acc, nullptr, 0, w.lane);
    if (w2b.h) {
        tile_row_scales<128>(S, LD128, rs);
        __syncthreads();
    }
    gemm_acc_x<128, 2>(S + w.rb * 32 * LD128, LD128, w2b, 16, 0, 2 * w.ch, acc, w.lane,
                       w2b.h ? rs + 64 * w.rb : nullptr);  // ds1 = da2 W2
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rr = w.rb * 32 + acc_row(r, w.lane), cc = 64 * w.ch + 32 * t + (w.lane & 31);
            const float d1 = acc[t][r] * silu_grad_(a1[t][r]);
            S[rr * LD128 + cc] = d1;  // da1
            if (TRAIN && row0 + rr < R) t_da1[(row0 + rr) * DH + cc] = d1;
        }
    __syncthreads();
    constexpr int NTO = K / 64;  // output columns K split over the two column halves
    f32x16 dx[NTO];
    acc_fill_bias<NTO>(dx, nullptr, 0, w.lane);
    if (w0b.h) {
        tile_row_scales<128>(S, LD128, rs);
        __syncthreads();
    }
    gemm_acc_x<128, NTO>(S + w.rb * 32 * LD128, LD128, w0b, 16, 0, NTO * w.ch, dx, w.lane,
                         w0b.h ? rs + 64 * w.rb : nullptr);  // dx = da1 W0
    acc_foreach<NTO>(dx, w.rb, (K / 2) * w.ch, w.lane, [&](int r, int c, float v) {
        if (row0 + r < R) dXout[(row0 + r) * K + c] = v;
    });
}

static int g_dxf_fused = 1;
void set_dxf_fused(int v) { g_dxf_fused = v ? 1 : 0; }
static inline bool dxf_fused_on() { return g_dxf_fused != 0; }

// dXF[p] = dM[p] + dcat[p][:D] + dcat[rev[p]][D:]   (inference graphs on one rank: formed inside k_comb_bwd_p2 / k_emlp_bwd_p2
// instead -- `dxf_fused` in backward())
__global__ void k_dxf(const float* __restrict__ dM, const float* __restrict__ dcat, const int* __restrict__ rev,
                      float* __restrict__ dX, int64_t E) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * (D / 4)) return;
    const int64_t p = idx / (D / 4);
    const int c = (int)(idx % (D / 4));
    float4 a = reinterpret_cast<const float4*>(dM)[idx];
    float4 b = *reinterpret_cast<const float4*>(dcat + p * (2 * D) + 4 * c);
    float4 d = *reinterpret_cast<const float4*>(dcat + (int64_t)rev[p] * (2 * D) + D + 4 * c);
    reinterpret_cast<float4*>(dX)[idx] = make_float4(a.x + b.x + d.x, a.y + b.y + d.y, a.z + b.z + d.z, a.w + b.w + d.w);
}

// ---------------------------------------------------------------------------------
// SwiGLU MLP adjoint (edge rows: K = 128, hidden 256; node rows: K = 256, hidden 512)
//   y = x + Wout (v * sig(g)),  [v; g] = Win RMSNorm(x)
//   dx = dy + RMSNorm^T( Win^T [du sig(g) ; du v sig'(g)] ),  du = Wout^T dy
// ---------------------------------------------------------------------------------
// NORM = false (PostLN, transformer.py:246-247: the MLP reads already-normalised tokens): dx = dy + Win^T [..]; ln: LayerNorm
template <int K, int HID, bool TRAIN, bool NORM = true>
__global__ __launch_bounds__(NTHREADS) void k_swiglu_bwd(const float* __restrict__ dY, const float* __restrict__ Xin,
                                                          const float* __restrict__ VG, const float* __restrict__ gamma,
                                                          const float4* __restrict__ woutb, const float4* __restrict__ winb,
                                                          float* __restrict__ dXout, int64_t R, float* __restrict__ t_dvg,
                                                          bool ln = false) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDK = lds_ld(K);
    constexpr int NTO = K / 64;
    float* A = smem;              // [64][K+4]: dY tile, later w = gamma * dn
    float* U = smem + BM * LDK;   // [64][132]
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<K>(A, dY, row0, R, K);
    __syncthreads();
    f32x16 dn[NTO];
    acc_fill_bias<NTO>(dn, nullptr, 0, w.lane);
#pragma unroll 1
    for (int hc = 0; hc < HID / 128; hc++) {
        f32x16 du[2];
        acc_fill_bias<2>(du, nullptr, 0, w.lane);
        // du chunk = dY Wout[:, chunk]  (woutb: rows = hidden, k = K)
        gemm_acc<K, 2>(A + w.rb * 32 * LDK, LDK, woutb, K / 8, 0, 4 * hc + 2 * w.ch, du, w.lane);
        float sg[2][16], vv[2][16];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t row = row0 + w.rb * 32 + acc_row(r, w.lane);
                const int cc = 128 * hc + 64 * w.ch + 32 * t + (w.lane & 31);
                float v = 0.f, g = 0.f;
                if (row < R) {
                    v = VG[row * (2 * HID) + cc];
                    g = VG[row * (2 * HID) + HID + cc];
                }
                sg[t][r] = sigmoidf_(g);
                vv[t][r] = v;
            }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = w.rb * 32 + acc_row(r, w.lane), cc = 64 * w.ch + 32 * t + (w.lane & 31);
                const float dvv = du[t][r] * sg[t][r];
                U[rr * LD128 + cc] = dvv;  // dv
                if (TRAIN && row0 + rr < R) t_dvg[(row0 + rr) * (2 * HID) + 128 * hc + cc] = dvv;
            }
        __syncthreads();
        gemm_acc<128, NTO>(U + w.rb * 32 * LD128, LD128, winb, 2 * HID / 8, 16 * hc, NTO * w.ch, dn, w.lane);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = w.rb * 32 + acc_row(r, w.lane), cc = 64 * w.ch + 32 * t + (w.lane & 31);
                const float dgv = du[t][r] * vv[t][r] * sigmoid_grad_from(sg[t][r]);
                U[rr * LD128 + cc] = dgv;  // dg
                if (TRAIN && row0 + rr < R) t_dvg[(row0 + rr) * (2 * HID) + HID + 128 * hc + cc] = dgv;
            }
        __syncthreads();
        gemm_acc<128, NTO>(U + w.rb * 32 * LD128, LD128, winb, 2 * HID / 8, HID / 8 + 16 * hc, NTO * w.ch, dn, w.lane);
    }
    if (!NORM) {
        acc_foreach<NTO>(dn, w.rb, (K / 2) * w.ch, w.lane, [&](int r, int c, float v) {
            const int64_t row = row0 + r;
            if (row < R) dXout[row * K + c] = dY[row * K + c] + v;
        });
        return;
    }
    __syncthreads();  // everyone is done with the dY tile in A
    acc_foreach<NTO>(dn, w.rb, (K / 2) * w.ch, w.lane, [&](int r, int c, float v) { A[r * LDK + c] = v * gamma[c]; });
    __syncthreads();
    norm_bwd_rows<K>(A, Xin, row0, R, K, ln, [&](int r, int c, float4 dx) {
        const int64_t o = (row0 + r) * K + c;
        float4 dy = *reinterpret_cast<const float4*>(dY + o);
        *reinterpret_cast<float4*>(dXout + o) = make_float4(dy.x + dx.x, dy.y + dx.y, dy.z + dx.z, dy.w + dx.w);
    });
}

// The node-row instance (K = 256, hidden 512) rebuilt like k_node2 (pet_fwd.hip): the dY tile is scaled per row by a
// power of two (adjoint rows have any magnitude) and split ONCE into fp16 planes for the four du chunk GEMMs, all products
// on the 16-bit matrix cores (the fp32-MFMA form above spends 46 % of its time in the matrix pipe: 64 cycles per 2 k),
// everything between the two scalings stays in scaled units; the saved pre-activations arrive as float4 rows through
// wave-private staging tiles. Inference only (no weight-gradient exports).
// SPLIT (RB = 1): the hidden chunks of a row tile on four workgroups as in k_node2 -- partial dn tiles through Pp (device-
// coherent stores / loads), the workgroup that arrives last at the tile's counter sums them in chunk order (deterministic;
// not the bits of the unsplit accumulation, whose chunks interleave their two products) and runs the epilogue.
template <int RB, bool SPLIT = false>  // 32 RB rows per workgroup (see k_node2)
__global__ __launch_bounds__(NTHREADS) void k_node_bwd2(const float* __restrict__ dY, const float* __restrict__ Xin,
                                                         const float* __restrict__ VG, const float* __restrict__ gamma,
                                                         WX woutb, WX winb, float* __restrict__ dXout, int64_t R, bool ln,
                                                         const float4* __restrict__ wceb, float* __restrict__ dOC,
                                                         float* __restrict__ Pp, int* __restrict__ cnt) {
    static_assert(!SPLIT || RB == 1, "the hidden-chunk split is built for the 32-row workgroups");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = 256, HID = DNF, LDK = lds_ld(K), LDH = plane_ld(K);
    constexpr int ROWS = 32 * RB, NCH = 4 / RB, WC = 256 / NCH, HC = 128 / NCH, NTH = HC / 32, NTO = WC / 32;
    constexpr int XD = RB == 1 ? 8 : 2;  // weight blocks in flight (tile.h gemm_acc_x)
    float* A = smem;                                                 // [ROWS][260] dY tile; at the end w = gamma * dn
    float* U = smem;                                                 // [ROWS][132] dv / dg chunk (aliases A while A is dead)
    float* stage = smem + ROWS * LD128;                              // 4 staging tiles (behind U, inside A)
    _Float16* Ph = reinterpret_cast<_Float16*>(smem + ROWS * LDK);   // [ROWS][264] planes of the scaled dY rows
    _Float16* Pl = Ph + ROWS * LDH;
    float* rs = smem + ROWS * LDK + ROWS * LDH;                      // [ROWS][2] scale, inverse
    const WaveIdT<RB> w;
    constexpr int SW = RB == 2 ? 64 : 32;                            // staging tile width
    float* my_stage = stage + w.wave * (32 * SW);
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int64_t wrow0 = row0 + 32 * w.rb;
    load_rows_to_lds<K, ROWS>(A, dY, row0, R, K);
    __syncthreads();
    tile_row_scales<K, ROWS>(A, LDK, rs);
    __syncthreads();
    split_tile_planes_scaled<K, ROWS>(A, LDK, rs, Ph, Pl);
    __syncthreads();  // A is dead until the epilogue
    f32x16 dn[NTO];
    acc_fill_bias<NTO>(dn, nullptr, 0, w.lane);
    const int hc_lo = SPLIT ? (int)blockIdx.y : 0, hc_hi = SPLIT ? (int)blockIdx.y + 1 : HID / 128;
#pragma unroll 1
    for (int hc = hc_lo; hc < hc_hi; hc++) {
        f32x16 du[NTH];
        acc_fill_bias<NTH>(du, nullptr, 0, w.lane);
        const int hcol0 = 128 * hc + HC * w.ch;
        // RB == 1 (small graphs, one wave per SIMD, every product starts with an exposed round trip): the K = 128 operands of
        // this chunk's two dn products and its saved [v | g] rows are requested around the du product instead of after it
        XRing<NTO, 8> r1, r2;
        const bool ring = RB == 1 && winb.h != nullptr;
        float4 vpre[4], gpre[4];
        if constexpr (RB == 1) {
            if (ring) xring_request(r1, winb, 2 * HID / 8, 16 * hc, NTO * w.ch, w.lane);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t rr = wrow0 + 8 * j + (w.lane >> 3);
                const float* p = VG + (rr < R ? rr : R - 1) * (2 * HID) + hcol0 + 4 * (w.lane & 7);
                vpre[j] = *reinterpret_cast<const float4*>(p);
                gpre[j] = *reinterpret_cast<const float4*>(p + HID);
                if (rr >= R) vpre[j] = gpre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        gemm_acc_hs<K, NTH, (RB == 1 ? 4 : 8)>(Ph + w.rb * 32 * LDH, Pl + w.rb * 32 * LDH, LDH, woutb, K / 8, 0, hcol0 / 32, du, w.lane);
        if (ring) xring_request(r2, winb, 2 * HID / 8, HID / 8 + 16 * hc, NTO * w.ch, w.lane);
        float sg[NTH][16], vv[NTH][16];
        auto vg_rows = [&](int off) {
            return [&, off](int r, int cc, float4& v) {
                if constexpr (RB == 1) {
                    v = off ? gpre[r >> 3] : vpre[r >> 3];  // r = 8 j + (lane >> 3), cc = 4 (lane & 7): the mapping of the request
                } else
                v = wrow0 + r < R ? *reinterpret_cast<const float4*>(VG + (wrow0 + r) * (2 * HID) + off + hcol0 + cc)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            };
        };
        if constexpr (RB == 2) wave_load_rows64(my_stage, w.lane, vg_rows(0));
        else wave_load_rows32(my_stage, w.lane, vg_rows(0));
#pragma unroll
        for (int t = 0; t < NTH; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) vv[t][r] = my_stage[acc_row(r, w.lane) * SW + 32 * t + (w.lane & 31)];
        __builtin_amdgcn_wave_barrier();
        if constexpr (RB == 2) wave_load_rows64(my_stage, w.lane, vg_rows(HID));
        else wave_load_rows32(my_stage, w.lane, vg_rows(HID));
#pragma unroll
        for (int t = 0; t < NTH; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) sg[t][r] = sigmoidf_(my_stage[acc_row(r, w.lane) * SW + 32 * t + (w.lane & 31)]);
        __builtin_amdgcn_wave_barrier();
        __syncthreads();  // the previous chunk's readers of U are done
#pragma unroll
        for (int t = 0; t < NTH; t++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                U[(w.rb * 32 + acc_row(r, w.lane)) * LD128 + HC * w.ch + 32 * t + (w.lane & 31)] = du[t][r] * sg[t][r];  // dv
        __syncthreads();
        if (ring) gemm_acc_x_ring<NTO, 8>(U + w.rb * 32 * LD128, LD128, r1, dn, w.lane);
        else gemm_acc_x<128, NTO, XD>(U + w.rb * 32 * LD128, LD128, winb, 2 * HID / 8, 16 * hc, NTO * w.ch, dn, w.lane);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NTH; t++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                U[(w.rb * 32 + acc_row(r, w.lane)) * LD128 + HC * w.ch + 32 * t + (w.lane & 31)] =
                    du[t][r] * vv[t][r] * sigmoid_grad_from(sg[t][r]);  // dg
        __syncthreads();
        if (ring) gemm_acc_x_ring<NTO, 8>(U + w.rb * 32 * LD128, LD128, r2, dn, w.lane);
        else gemm_acc_x<128, NTO, XD>(U + w.rb * 32 * LD128, LD128, winb, 2 * HID / 8, HID / 8 + 16 * hc, NTO * w.ch, dn, w.lane);
    }
    __syncthreads();  // everyone is done with U: A takes w = gamma * dn (back in true units)
    if constexpr (SPLIT) {
        __shared__ int last_arrival;
        constexpr int NCHK = HID / 128, IT = ROWS * (K / 4) / NTHREADS;
        const size_t prows = (size_t)gridDim.x * ROWS;
#pragma unroll
        for (int t = 0; t < NTO; t++)
            wave_rows32(dn[t], my_stage, w.lane, [&](int r, int cc, float4 v) {
                st4_agent(Pp + ((size_t)blockIdx.y * prows + (wrow0 + r)) * K + WC * w.ch + 32 * t + cc, v);
            });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the coherent stores are acknowledged (k_node2)
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = atomicAdd(&cnt[blockIdx.x], 1);
            last_arrival = old == NCHK - 1;
            if (last_arrival) cnt[blockIdx.x] = 0;
        }
        __syncthreads();
        if (!last_arrival) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // consumer side of the hand-over (see k_node2 in pet_fwd.hip)
        float4 pk[IT][NCHK];
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int idx = threadIdx.x + it * NTHREADS, r = idx / (K / 4), c = 4 * (idx % (K / 4));
#pragma unroll
            for (int k = 0; k < NCHK; k++) pk[it][k] = ld4_agent(Pp + ((size_t)k * prows + row0 + r) * K + c);
        }
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int idx = threadIdx.x + it * NTHREADS, r = idx / (K / 4), c = 4 * (idx % (K / 4));
            float4 sum = pk[it][0];
#pragma unroll
            for (int k = 1; k < NCHK; k++)
                sum = make_float4(sum.x + pk[it][k].x, sum.y + pk[it][k].y, sum.z + pk[it][k].z, sum.w + pk[it][k].w);
            const float sc = rs[2 * r + 1];
            const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
            *reinterpret_cast<float4*>(A + r * LDK + c) = make_float4(sum.x * sc * ga.x, sum.y * sc * ga.y, sum.z * sc * ga.z, sum.w * sc * ga.w);
        }
    } else
    acc_foreach<NTO>(dn, w.rb, WC * w.ch, w.lane, [&](int r, int c, float v) { A[r * LDK + c] = v * rs[2 * r + 1] * gamma[c]; });
    __syncthreads();
    // dOC = dH1 Wce (k_expand_bwd's GEMM) from the dH1 tile while it is on chip, when the caller passes dOC: one launch less
    // on the critical path of a small box. The tile goes over the planes, which nobody reads any more.
    float* T = smem + ROWS * LDK;  // [ROWS][260]
    if (dOC) {
        for (int idx = threadIdx.x; idx < ROWS * (K / 4); idx += NTHREADS)  // rows past the end stay zero
            *reinterpret_cast<float4*>(T + (idx / (K / 4)) * LDK + 4 * (idx % (K / 4))) = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    norm_bwd_rows<K, ROWS>(A, Xin, row0, R, K, ln, [&](int r, int c, float4 dx) {
        const int64_t o = (row0 + r) * K + c;
        const float4 dy = *reinterpret_cast<const float4*>(dY + o);
        const float4 v = make_float4(dy.x + dx.x, dy.y + dx.y, dy.z + dx.z, dy.w + dx.w);
        *reinterpret_cast<float4*>(dXout + o) = v;
        if (dOC) *reinterpret_cast<float4*>(T + r * LDK + c) = v;
    });
    if (dOC) {
        constexpr int NTC = RB;  // 128 columns over the NCH column groups
        __syncthreads();
        f32x16 acc[NTC];
        acc_fill_bias<NTC>(acc, nullptr, 0, w.lane);
        if constexpr (RB == 1) gemm_acc_deep<256, NTC, 8>(T + w.rb * 32 * LDK, LDK, wceb, 32, 0, NTC * w.ch, acc, w.lane);
        else gemm_acc<256, NTC>(T + w.rb * 32 * LDK, LDK, wceb, 32, 0, NTC * w.ch, acc, w.lane);
        acc_foreach<NTC>(acc, w.rb, 32 * NTC * w.ch, w.lane, [&](int r, int c, float v) {
            if (row0 + r < R) dOC[(row0 + r) * D + c] = v;
        });
    }
}

// ---------------------------------------------------------------------------------
// node chain adjoint, second half: dOC = dH1 Wce ; (dH_in = dH1 is finished by k_center_bwd)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_expand_bwd(const float* __restrict__ dH1, const float4* __restrict__ wceb,
                                                          float* __restrict__ dOC, int64_t N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<256>(smem, dH1, row0, N, DN);
    __syncthreads();
    f32x16 acc[2];
    acc_fill_bias<2>(acc, nullptr, 0, w.lane);
    gemm_acc<256, 2>(smem + w.rb * 32 * LD256, LD256, wceb, 32, 0, 2 * w.ch, acc, w.lane);
    __syncthreads();  // the dH1 tile is consumed: its memory stages the float4 row stores (tile.h wave_rows64)
    const int64_t wrow0 = row0 + 32 * w.rb;
    wave_rows64(acc, smem + w.wave * (32 * 64), w.lane, [&](int r, int cc, float4 v) {
        if (wrow0 + r < N) *reinterpret_cast<float4*>(dOC + (wrow0 + r) * D + 64 * w.ch + cc) = v;
    });
}

// dH_in = dH1 + dC Wcc   (centre contraction adjoint); DEEP (small graphs): eight weight blocks in flight
template <bool DEEP>
__global__ __launch_bounds__(NTHREADS) void k_center_bwd(const float* __restrict__ dC, const float* __restrict__ dH1,
                                                          const float4* __restrict__ wccb, float* __restrict__ dHin,
                                                          int64_t N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<128>(smem, dC, row0, N, D);
    __syncthreads();
    f32x16 acc[4];
    acc_fill_bias<4>(acc, nullptr, 0, w.lane);
    if constexpr (DEEP) gemm_acc_deep<128, 4, 8>(smem + w.rb * 32 * LD128, LD128, wccb, 16, 0, 4 * w.ch, acc, w.lane);
    else gemm_acc<128, 4>(smem + w.rb * 32 * LD128, LD128, wccb, 16, 0, 4 * w.ch, acc, w.lane);
    __syncthreads();  // the dC tile is consumed: its memory stages float4 row traffic (tile.h wave_rows64)
    const int64_t wrow0 = row0 + 32 * w.rb;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const f32x16 pair[2] = {acc[2 * half], acc[2 * half + 1]};
        wave_rows64(pair, smem + w.wave * (32 * 64), w.lane, [&](int r, int cc, float4 v) {
            const int64_t row = wrow0 + r;
            if (row >= N) return;
            const int64_t o = row * DN + 128 * w.ch + 64 * half + cc;
            const float4 d1 = *reinterpret_cast<const float4*>(dH1 + o);
            *reinterpret_cast<float4*>(dHin + o) = make_float4(d1.x + v.x, d1.y + v.y, d1.z + v.z, d1.w + v.w);
        });
    }
}

// ---------------------------------------------------------------------------------
// output_linear adjoint: dAO = dOut Wo, rows < E from dX1, rows >= E from dOC
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_oproj_bwd(const float* __restrict__ dX1, const float* __restrict__ dOC,
                                                         const float4* __restrict__ wob, float* __restrict__ dAO,
                                                         int64_t E, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    for (int idx = threadIdx.x; idx < BM * 32; idx += NTHREADS) {
        const int r = idx >> 5, c = idx & 31;
        const int64_t row = row0 + r;
        float4 v = make_float4(0, 0, 0, 0);
        if (row < E) v = *reinterpret_cast<const float4*>(dX1 + row * D + 4 * c);
        else if (row < R) v = *reinterpret_cast<const float4*>(dOC + (row - E) * D + 4 * c);
        *reinterpret_cast<float4*>(smem + r * LD128 + 4 * c) = v;
    }
    __syncthreads();
    f32x16 acc[2];
    acc_fill_bias<2>(acc, nullptr, 0, w.lane);
    gemm_acc<128, 2>(smem + w.rb * 32 * LD128, LD128, wob, 16, 0, 2 * w.ch, acc, w.lane);
    acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
        if (row0 + r < R) dAO[(row0 + r) * D + c] = v;
    });
}

// ---------------------------------------------------------------------------------
// attention adjoint: one wave per (atom, head). Pass A (query tiles, transposed scores)
// gives dQ, delta, lse and the key-bias gradient; pass B (key tiles, plain scores)
// gives dK and dV. 16x16x4 fp32 MFMA, no LDS, no atomics.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int64_t token_row_b(int t, int T, int64_t E, int atom, int start) {
    return (t == 0 || t >= T) ? E + atom : (int64_t)start + t - 1;
}

template <int NT>
__global__ __launch_bounds__(256) void k_attn_bwd(const float* __restrict__ QKV, const float* __restrict__ dAO,
                                                   const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                   float* __restrict__ dQKV, float* __restrict__ dbias_h,
                                                   int64_t E, int N, float scale) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int atom = gw / NHEAD, head = gw % NHEAD;
    if (atom >= N) return;
    const int start = rowptr[atom];
    const int T = rowptr[atom + 1] - start + 1;
    const int nt = (T + 15) >> 4;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int qo = HD * head, ko = D + HD * head, vo = 2 * D + HD * head;
    float4 kf[NT], vf[NT];
    float bias_r[NT][4];  // bias of key 16kt + 4g4 + r (rows of the transposed tile)
    float bias_c[NT];     // bias of key 16kt + c16   (columns of the plain tile)
    float db[NT][4];
    float lse[NT], delta[NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        lse[kt] = 0.f;
        delta[kt] = 0.f;
        if (kt < nt) {
            const int64_t row = token_row_b(16 * kt + c16, T, E, atom, start);
            kf[kt] = *reinterpret_cast<const float4*>(QKV + row * (3 * D) + ko + 4 * g4);
            vf[kt] = *reinterpret_cast<const float4*>(QKV + row * (3 * D) + vo + 4 * g4);
            const int kc = 16 * kt + c16;
            bias_c[kt] = kc >= T ? -INFINITY : (kc == 0 ? 0.f : logf(fmaxf(fc[start + kc - 1], 1e-15f)));
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = 16 * kt + 4 * g4 + r;
                bias_r[kt][r] = key >= T ? -INFINITY : (key == 0 ? 0.f : logf(fmaxf(fc[start + key - 1], 1e-15f)));
                db[kt][r] = 0.f;
            }
        }
    }
    // ------------------------------ pass A ------------------------------
#pragma unroll
    for (int qt = 0; qt < NT; qt++) {
        if (qt < nt) {
            const int q = 16 * qt + c16;
            const int64_t qrow = token_row_b(q, T, E, atom, start);
            float4 qf = *reinterpret_cast<const float4*>(QKV + qrow * (3 * D) + qo + 4 * g4);
            qf.x *= scale; qf.y *= scale; qf.z *= scale; qf.w *= scale;
            float4 dof = make_float4(0, 0, 0, 0);
            if (q < T) dof = *reinterpret_cast<const float4*>(dAO + qrow * D + qo + 4 * g4);
            f32x4 s[NT], dp[NT];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; kt++) {
                if (kt < nt) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].x, qf.x, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].y, qf.y, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].z, qf.z, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].w, qf.w, a, 0, 0, 0);
                    f32x4 b = {0.f, 0.f, 0.f, 0.f};
                    b = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt].x, dof.x, b, 0, 0, 0);
                    b = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt].y, dof.y, b, 0, 0, 0);
                    b = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt].z, dof.z, b, 0, 0, 0);
                    b = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt].w, dof.w, b, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        a[r] += bias_r[kt][r];
                        mx = fmaxf(mx, a[r]);
                    }
                    s[kt] = a;
                    dp[kt] = b;
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float p = expf(s[kt][r] - mx);
                        s[kt][r] = p;
                        sum += p;
                    }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float inv = 1.0f / sum;
            float dl = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        s[kt][r] *= inv;
                        dl += s[kt][r] * dp[kt][r];
                    }
            dl += __shfl_xor(dl, 16);
            dl += __shfl_xor(dl, 32);
            lse[qt] = mx + logf(sum);
            delta[qt] = dl;
            f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float ds = s[kt][r] * (dp[kt][r] - dl);  // dS^T[key][q]
                        db[kt][r] += ds;  // padded queries have dO = 0 => dp = dl = 0 => ds = 0
                        const int64_t krow = token_row_b(16 * kt + 4 * g4 + r, T, E, atom, start);
                        const float kk = QKV[krow * (3 * D) + ko + c16];
                        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kk, ds, dq, 0, 0, 0);
                    }
            if (q < T)
                *reinterpret_cast<float4*>(dQKV + qrow * (3 * D) + qo + 4 * g4) =
                    make_float4(dq[0] * scale, dq[1] * scale, dq[2] * scale, dq[3] * scale);
        }
    }
    // key-bias gradient: sum over the 16 query columns; lanes c16 < 4 write one key each into the attention
    // layer's head-major slice [NHEAD, E] (one writer per (layer, head, edge): plain stores, k_dfc_attn sums)
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
        if (kt < nt) {
            float v4[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = db[kt][r];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                v4[r] = v;
            }
            const float v = c16 == 0 ? v4[0] : c16 == 1 ? v4[1] : c16 == 2 ? v4[2] : v4[3];
            const int key = 16 * kt + 4 * g4 + c16;
            if (c16 < 4 && key >= 1 && key < T) dbias_h[(int64_t)head * E + start + key - 1] = v;
        }
    // ------------------------------ pass B ------------------------------
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        if (kt < nt) {
            f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dvv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qt = 0; qt < NT; qt++) {
                if (qt < nt) {
                    const int64_t qrow_c = token_row_b(16 * qt + c16, T, E, atom, start);
                    float4 qf = *reinterpret_cast<const float4*>(QKV + qrow_c * (3 * D) + qo + 4 * g4);
                    qf.x *= scale; qf.y *= scale; qf.z *= scale; qf.w *= scale;
                    float4 dof = make_float4(0, 0, 0, 0);
                    if (16 * qt + c16 < T) dof = *reinterpret_cast<const float4*>(dAO + qrow_c * D + qo + 4 * g4);
                    // S[q][key]: A = Q (rows q), B = K (cols key)
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.x, kf[kt].x, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.y, kf[kt].y, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.z, kf[kt].z, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.w, kf[kt].w, a, 0, 0, 0);
                    f32x4 b = {0.f, 0.f, 0.f, 0.f};  // dP[q][key] = dO V^T
                    b = __builtin_amdgcn_mfma_f32_16x16x4f32(dof.x, vf[kt].x, b, 0, 0, 0);
                    b = __builtin_amdgcn_mfma_f32_16x16x4f32(dof.y, vf[kt].y, b, 0, 0, 0);
                    b = __builtin_amdgcn_mfma_f32_16x16x4f32(dof.z, vf[kt].z, b, 0, 0, 0);
                    b = __builtin_amdgcn_mfma_f32_16x16x4f32(dof.w, vf[kt].w, b, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int qq = 16 * qt + 4 * g4 + r;  // row of the plain tile
                        const float l = __shfl(lse[qt], 4 * g4 + r);
                        const float dl = __shfl(delta[qt], 4 * g4 + r);
                        float p = expf(a[r] + bias_c[kt] - l);
                        if (qq >= T) p = 0.f;
                        const float ds = p * (b[r] - dl);
                        const int64_t qrow_r = token_row_b(qq, T, E, atom, start);
                        float dov = 0.f;
                        if (qq < T) dov = dAO[qrow_r * D + qo + c16];
                        const float qv = QKV[qrow_r * (3 * D) + qo + c16];
                        dvv = __builtin_amdgcn_mfma_f32_16x16x4f32(dov, p, dvv, 0, 0, 0);
                        dk = __builtin_amdgcn_mfma_f32_16x16x4f32(qv, ds, dk, 0, 0, 0);
                    }
                }
            }
            const int key = 16 * kt + c16;
            if (key < T) {
                const int64_t krow = token_row_b(key, T, E, atom, start);
                *reinterpret_cast<float4*>(dQKV + krow * (3 * D) + ko + 4 * g4) =
                    make_float4(dk[0] * scale, dk[1] * scale, dk[2] * scale, dk[3] * scale);
                *reinterpret_cast<float4*>(dQKV + krow * (3 * D) + vo + 4 * g4) =
                    make_float4(dvv[0], dvv[1], dvv[2], dvv[3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// input_linear + RMSNorm adjoint: dXin = (rows<E ? dX1 : 0) + RMSNorm^T(dQKV Win)
// ---------------------------------------------------------------------------------
// POST (transformer.py:243-245): no norm in front of input_linear and every token row carries the residual gradient dX1
template <bool POST>
__global__ __launch_bounds__(NTHREADS) void k_qkv_bwd(const float* __restrict__ dQKV, const float* __restrict__ X,
                                                       const float* __restrict__ gamma, const float4* __restrict__ winb,
                                                       const float* __restrict__ dX1, float* __restrict__ dXin,
                                                       int64_t E, int64_t R, bool ln) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    f32x16 dn[2];
    acc_fill_bias<2>(dn, nullptr, 0, w.lane);
#pragma unroll 1
    for (int ks = 0; ks < 3; ks++) {
        __syncthreads();
        load_rows_to_lds<128>(smem, dQKV + 128 * ks, row0, R, 3 * D);
        __syncthreads();
        gemm_acc<128, 2>(smem + w.rb * 32 * LD128, LD128, winb, 48, 16 * ks, 2 * w.ch, dn, w.lane);
    }
    if (POST) {
        acc_foreach<2>(dn, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
            const int64_t row = row0 + r;
            if (row < R) dXin[row * D + c] = dX1[row * D + c] + v;
        });
        return;
    }
    __syncthreads();
    acc_foreach<2>(dn, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) { smem[r * LD128 + c] = v * gamma[c]; });
    __syncthreads();
    norm_bwd_rows<128>(smem, X, row0, R, D, ln, [&](int r, int c, float4 dx) {
        const int64_t row = row0 + r;
        if (row < E) {
            float4 d1 = *reinterpret_cast<const float4*>(dX1 + row * D + c);
            dx.x += d1.x; dx.y += d1.y; dx.z += d1.z; dx.w += d1.w;
        }
        *reinterpret_cast<float4*>(dXin + row * D + c) = dx;
    });
}

// ---------------------------------------------------------------------------------
// PostLN: adjoint of Y = Norm(S) (k_rownorm): rows < E of dY from dYe, rows >= E from dYc; dS on all E+N rows
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_rownorm_bwd(const float* __restrict__ dYe, const float* __restrict__ dYc,
                                                           const float* __restrict__ S, const float* __restrict__ gamma,
                                                           bool ln, float* __restrict__ dS, int64_t E, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    for (int idx = threadIdx.x; idx < BM * 32; idx += NTHREADS) {
        const int r = idx >> 5, c = idx & 31;
        const int64_t row = row0 + r;
        float4 v = make_float4(0, 0, 0, 0);
        if (row < E) v = *reinterpret_cast<const float4*>(dYe + row * D + 4 * c);
        else if (row < R) v = *reinterpret_cast<const float4*>(dYc + (row - E) * D + 4 * c);
        const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * c);
        *reinterpret_cast<float4*>(smem + r * LD128 + 4 * c) = make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
    }
    __syncthreads();
    norm_bwd_rows<128>(smem, S, row0, R, D, ln, [&](int r, int c, float4 dx) {
        *reinterpret_cast<float4*>(dS + (row0 + r) * D + c) = dx;
    });
}

// residual featuriser adjoint (k_resmix): Mout[p] = 0.5 (Min[p] + e[rev[p]]) with rev an involution, so
//   de[p] = ge[p] + 0.5 dMout[rev[p]],  dMin[p] = 0.5 dMout[p];  dMout == nullptr (last layer): de = ge, dMin = 0
__global__ void k_resmix_bwd(const float* __restrict__ ge, const float* __restrict__ dMout, const int* __restrict__ rev,
                             float* __restrict__ de, float* __restrict__ dMin, int64_t E) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * (D / 4)) return;
    const int64_t p = idx / (D / 4);
    const int c = (int)(idx % (D / 4));
    float4 a = ge ? reinterpret_cast<const float4*>(ge)[idx] : make_float4(0, 0, 0, 0);
    float4 mp = make_float4(0, 0, 0, 0);
    if (dMout) {
        const float4 b = *reinterpret_cast<const float4*>(dMout + (int64_t)rev[p] * D + 4 * c);
        a.x += 0.5f * b.x; a.y += 0.5f * b.y; a.z += 0.5f * b.z; a.w += 0.5f * b.w;
        const float4 q = reinterpret_cast<const float4*>(dMout)[idx];
        mp = make_float4(0.5f * q.x, 0.5f * q.y, 0.5f * q.z, 0.5f * q.w);
    }
    reinterpret_cast<float4*>(de)[idx] = a;
    reinterpret_cast<float4*>(dMin)[idx] = mp;
}

// ---------------------------------------------------------------------------------
// compress adjoint: da0 = (dE W2) * silu'(a0); dgeo += da0 Wc; dM = dMpass + da0 W0c
// ---------------------------------------------------------------------------------
template <bool FIRST, bool TRAIN>
__global__ __launch_bounds__(NTHREADS) void k_compress_bwd(const float* __restrict__ dXe, const float* __restrict__ a0,
                                                            const float4* __restrict__ w2b, const float* __restrict__ wct,
                                                            const float4* __restrict__ w0cb, float* __restrict__ dgeo,
                                                            float* __restrict__ dM /* in/out, !FIRST */, int64_t E,
                                                            float* __restrict__ t_da0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<128>(smem, dXe, row0, E, D);
    __syncthreads();
    f32x16 acc[2];
    acc_fill_bias<2>(acc, nullptr, 0, w.lane);
    gemm_acc<128, 2>(smem + w.rb * 32 * LD128, LD128, w2b, 16, 0, 2 * w.ch, acc, w.lane);
    __syncthreads();
    acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
        const int64_t row = row0 + r;
        const float d0v = row < E ? v * silu_grad_(a0[row * D + c]) : 0.f;
        smem[r * LD128 + c] = d0v;
        if (TRAIN && row < E) t_da0[row * D + c] = d0v;
    });
    __syncthreads();
    {   // dgeo[row][k] += sum_c da0[row][c] * Wc[c][k]; 4 threads per row, thread q -> component q
        const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
        const float* wrow = wct + q * D;
        float s = 0.f;
        for (int c = 0; c < 128; c += 4) {
            float4 a = *reinterpret_cast<float4*>(smem + r * LD128 + c);
            float4 ww = *reinterpret_cast<const float4*>(wrow + c);
            s += a.x * ww.x + a.y * ww.y + a.z * ww.z + a.w * ww.w;
        }
        if (row0 + r < E) dgeo[(row0 + r) * 4 + q] += s;
    }
    if (!FIRST) {
        acc_fill_bias<2>(acc, nullptr, 0, w.lane);
        gemm_acc<128, 2>(smem + w.rb * 32 * LD128, LD128, w0cb, 16, 0, 2 * w.ch, acc, w.lane);
        acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
            const int64_t row = row0 + r;
            if (row < E) dM[row * D + c] += v;
        });
    }
}

// ---------------------------------------------------------------------------------
// geometry adjoint
// ---------------------------------------------------------------------------------
// key-bias adjoint -> cutoff-factor gradient: bias = log(clamp(fc, 1e-15)), so the gradient
// passes only where fc >= 1e-15 (transformer.py:109-110); heads and layers are summed here.
// dbias_l: one head-major slice [NHEAD, E] per attention layer, each written once by that layer's adjoint
__global__ void k_dfc_attn(const float* __restrict__ fc, const float* __restrict__ dbias_l, int slices,
                           float* __restrict__ dfc_attn, int64_t E, int hstep) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    float dbias = 0.f;
    for (int h = 0; h < slices * NHEAD; h += hstep) dbias += dbias_l[(int64_t)h * E + p];
    const float f = fc[p];
    dfc_attn[p] = f >= 1e-15f ? dbias / f : 0.f;
}

// dgeo = d/d(vx,vy,vz,dist), dfc_a + dfc_b = d/d(cutoff factor)  ->  d/d(edge vector)
__global__ void k_geom_bwd(const float4* __restrict__ geo, const float* __restrict__ d0,
                           const float* __restrict__ dgeo, const float* __restrict__ dfc_a,
                           const float* __restrict__ dfc_b, float4* __restrict__ dv, int64_t E, float cutoff,
                           float width, int fn, const float* __restrict__ pc, float* __restrict__ gc) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    const float4 g = geo[p];
    const float4 dg = reinterpret_cast<const float4*>(dgeo)[p];
    const float dfc_t = (dfc_a ? dfc_a[p] : 0.f) + (dfc_b ? dfc_b[p] : 0.f);
    // adaptive cutoff: fc = f(d - c) with the pair cutoff c, so df/dc = -df/dd
    const float dd0 = dfc_t * cutoff_deriv_dev(d0[p], pc ? pc[p] : cutoff, width, fn);
    if (gc) gc[p] = -dd0;
    const float nrm = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
    const float c_dist = dg.w / g.w;                    // d sqrt(v.v + 1e-15) / dv = v / dist
    const float c_d0 = nrm > 0.f ? dd0 / nrm : 0.f;     // d |v| / dv = v / |v|
    const float c = c_dist + c_d0;
    dv[p] = make_float4(dg.x + c * g.x, dg.y + c * g.y, dg.z + c * g.z, 0.f);
}

// dE/dR_a = sum_{p in row a} (dv[rev[p]] - dv[p]); one 16-lane group per atom
__global__ void k_pos_grad(const float4* __restrict__ dv, const int* __restrict__ rowptr, const int* __restrict__ rev,
                           float* __restrict__ gpos, int N) {
    const int gid = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    const int a = gid < N ? gid : N - 1;
    float x = 0.f, y = 0.f, z = 0.f;
    for (int p = rowptr[a] + l; p < rowptr[a + 1]; p += 16) {
        const float4 m = dv[p], q = dv[rev[p]];
        x += q.x - m.x; y += q.y - m.y; z += q.z - m.z;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        x += __shfl_xor(x, o); y += __shfl_xor(y, o); z += __shfl_xor(z, o);
    }
    if (l == 0 && gid < N) {
        gpos[3 * gid] = x; gpos[3 * gid + 1] = y; gpos[3 * gid + 2] = z;
    }
}

// ---- adaptive cutoff adjoint (adaptive_cutoff.py:196-229: the implicit-function step carries the gradient)
// g_r[i] = dL/d r_i = sum over kept edges touching i of dL/dc / 2 = 1/2 sum_{p in row i} (gc[p] + gc[rev p])
__global__ void k_adapt_gr(const float* __restrict__ gc, const int* __restrict__ rowptr, const int* __restrict__ rev,
                           float* __restrict__ gr, int N) {
    const int gid = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    const int a = gid < N ? gid : N - 1;
    float s = 0.f;
    for (int p = rowptr[a] + l; p < rowptr[a + 1]; p += 16) s += gc[p] + gc[rev[p]];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (l == 0 && gid < N) gr[gid] = 0.5f * s;
}
// every input edge q of atom i (all-edge CSR): d r_i / d d_q = -(d bump(d_q; r, w)/d d_q) / dn_root,
// d bump / d d = -(d bump / d r); the edge vector gets (v / |v|) times that
__global__ void k_adapt_dv(const int* __restrict__ rowptr0, const int* __restrict__ perm0,
                           const float4* __restrict__ vin, const float* __restrict__ gr,
                           const float* __restrict__ r_newton, const float* __restrict__ inv_dn,
                           float4* __restrict__ dvA, int N, float w) {
    const int gid = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    if (gid >= N) return;
    const float coef = gr[gid] * inv_dn[gid], r = r_newton[gid];
    for (int q = rowptr0[gid] + l; q < rowptr0[gid + 1]; q += 16) {
        const float4 v = vin[perm0[q]];
        // d r_adapt / d d_q = -(d bump/d d)(d_q; r) * inv_dn  (bump derivative w.r.t. d, bump taper = BUMP)
        const float dd = -coef * cutoff_deriv_dev(v.w, r, w, PET_CUTOFF_BUMP);
        const float nrm = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
        const float c = nrm > 0.f ? dd / nrm : 0.f;
        dvA[q] = make_float4(c * v.x, c * v.y, c * v.z, 0.f);
    }
}
// "grid" method: r_i = sum_k p_k w_k(n_i1 .. n_iK), n_ik = sum_q bump(d_q; p_k, w): every input edge q of atom i gets
//   d r_i / d d_q = sum_k (d r_i / d n_ik) (d bump / d d)(d_q; p_k, w)   with the per-atom row drdn of k_adaptive_grid
__global__ void k_adapt_dv_grid(const int* __restrict__ rowptr0, const int* __restrict__ perm0,
                                const float4* __restrict__ vin, const float* __restrict__ gr,
                                const float* __restrict__ drdn, float4* __restrict__ dvA, int N, float w, int K,
                                float pmin, float dp) {
    __shared__ float s_c[16][GRID_MAX_PROBES];
    const int grp = threadIdx.x >> 4;
    const int gid = blockIdx.x * (blockDim.x / 16) + grp;
    const int l = threadIdx.x & 15;
    const int a = gid < N ? gid : N - 1;
    for (int k = l; k < K; k += 16) s_c[grp][k] = gr[a] * drdn[(int64_t)a * GRID_MAX_PROBES + k];
    __syncthreads();
    if (gid >= N) return;
    for (int q = rowptr0[gid] + l; q < rowptr0[gid + 1]; q += 16) {
        const float4 v = vin[perm0[q]];
        float dd = 0.f;
        for (int k = 0; k < K; k++) dd += s_c[grp][k] * cutoff_deriv_dev(v.w, pmin + (float)k * dp, w, PET_CUTOFF_BUMP);
        const float nrm = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
        const float c = nrm > 0.f ? dd / nrm : 0.f;
        dvA[q] = make_float4(c * v.x, c * v.y, c * v.z, 0.f);
    }
}
// gpos[a] += sum_{q in row0 a} (dvA[rev0 q] - dvA[q])
__global__ void k_pos_grad_acc(const float4* __restrict__ dv, const int* __restrict__ rowptr,
                               const int* __restrict__ rev, float* __restrict__ gpos, int N) {
    const int gid = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    const int a = gid < N ? gid : N - 1;
    float x = 0.f, y = 0.f, z = 0.f;
    for (int p = rowptr[a] + l; p < rowptr[a + 1]; p += 16) {
        const float4 m = dv[p];
        const int rp = rev[p];
        const float4 q = rp >= 0 ? dv[rp] : make_float4(0.f, 0.f, 0.f, 0.f);
        x += q.x - m.x; y += q.y - m.y; z += q.z - m.z;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        x += __shfl_xor(x, o); y += __shfl_xor(y, o); z += __shfl_xor(z, o);
    }
    if (l == 0 && gid < N) {
        gpos[3 * gid] += x; gpos[3 * gid + 1] += y; gpos[3 * gid + 2] += z;
    }
}

// dE/dcell[s][a][k] = sum_{edges of system s} S_a dv_k (structures.py:212-219); one block per system
__global__ void k_cell_grad(const float4* __restrict__ dv, const int* __restrict__ shift, const int* __restrict__ ctr,
                            const int* __restrict__ sys, const int* __restrict__ rowptr, float* __restrict__ gcell,
                            int N, int64_t E, int accumulate) {
    // systems are contiguous atom ranges; find this system's atom range by scanning sys[] boundaries
    const int s = blockIdx.x;
    __shared__ double red[9][256];  // fp64 sums: a system's edges with a shift are a signed sum with heavy cancellation
    __shared__ int range[2];
    if (threadIdx.x == 0) {
        int lo = 0, hi = N;
        // lower bound of s and of s+1 in the sorted sys[] array
        int a = 0, b = N;
        while (a < b) { int m = (a + b) >> 1; if (sys[m] < s) a = m + 1; else b = m; }
        lo = a; b = N;
        while (a < b) { int m = (a + b) >> 1; if (sys[m] < s + 1) a = m + 1; else b = m; }
        hi = a;
        range[0] = rowptr[lo];
        range[1] = rowptr[hi];
    }
    __syncthreads();
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = 0.0;
    for (int p = range[0] + threadIdx.x; p < range[1]; p += blockDim.x) {
        const int ia = shift[3 * p], ib = shift[3 * p + 1], ic = shift[3 * p + 2];
        if ((ia | ib | ic) == 0) continue;  // most edges stay inside the cell
        const float4 d = dv[p];
        const double sa = ia, sb = ib, sc = ic, dx = d.x, dy = d.y, dz = d.z;
        acc[0] += sa * dx; acc[1] += sa * dy; acc[2] += sa * dz;
        acc[3] += sb * dx; acc[4] += sb * dy; acc[5] += sb * dz;
        acc[6] += sc * dx; acc[7] += sc * dy; acc[8] += sc * dz;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
#pragma unroll
            for (int k = 0; k < 9; k++) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 9)
        gcell[9 * s + threadIdx.x] = (accumulate ? gcell[9 * s + threadIdx.x] : 0.f) + (float)red[threadIdx.x][0];
}

// ---------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------
// The adjoint kernels are templated on TRAIN (export of the operands the weight-gradient GEMMs need):
// launch KERN<..., true> when the training hooks are active, KERN<..., false> otherwise.
#define PET_TA(...) __VA_ARGS__
#define PET_LAUNCH_TR(tr, KERN, TARGS, GRID, LDS, STREAM, ...)                         \
    do {                                                                               \
        if (tr) KERN<TARGS, true><<<GRID, NTHREADS, LDS, STREAM>>>(__VA_ARGS__);       \
        else KERN<TARGS, false><<<GRID, NTHREADS, LDS, STREAM>>>(__VA_ARGS__);         \
    } while (0)
template <int NT>
static void launch_attn_bwd(const float* QKV, const float* dAO, const Graph& g, float* dQKV, float* dbias_h,
                            float scale, hipStream_t st) {
    int waves = (int)g.n_nodes * NHEAD;
    k_attn_bwd<NT><<<cdiv(waves, 4), 256, 0, st>>>(QKV, dAO, g.rowptr, g.fc, dQKV, dbias_h, g.n_edges,
                                                   (int)g.n_nodes, scale);
}

// both operand forms of a Linear for the LDS-tile kernels (tile.h gemm_acc_x); fwd: x W^T, bwd: dy W
static inline WX wx_f(const Lin& L) {
    WX w;
    w.f = L.fwd;
    if (L.fwd2) {
        const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
        w.h = reinterpret_cast<const f16x8_t*>(L.fwd2);
        w.l = w.h + n8;
    }
    return w;
}
static inline WX wx_b(const Lin& L) {
    WX w;
    w.f = L.bwd;
    if (L.bwd2) {
        const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
        w.h = reinterpret_cast<const f16x8_t*>(L.bwd2);
        w.l = w.h + n8;
    }
    return w;
}

// (pet_graph_build refuses a list with an edge that has no (j, i, -S) partner, so the reverse pass needs no check
//  and no host synchronisation of its own.)
// Stage P: adjoint of PETBackend.predict. seeds gA [N] -> w.dH [N,DN], w.dM [E,D], w.dfc [E]
int backward_predict(const Model& m, const Graph& g, Workspace& w, const float* gA, hipStream_t st,
                     Trainer* tr = nullptr) {
    const int64_t N = g.n_nodes, E = g.n_edges;
    const int gE = cdiv(E, BM), gN = cdiv(N, BM);
    const size_t lds2 = 2 * BM * LD128 * 4;
    const double fE = (double)E, fN = (double)N;
    allow_big_lds(k_head_bwd<256, false, false>, (BM * LD256 + BM * LD128) * 4 + 768);
    allow_big_lds(k_head_bwd<256, false, true>, (BM * LD256 + BM * LD128) * 4 + 768);
    allow_big_lds(k_head_bwd<128, true, false>, lds2 + 768);
    allow_big_lds(k_head_bwd<128, true, true>, lds2 + 768);
    const GnnBufs& last = w.gnn.back();
    if (E > 0) {
        ProfScope ps("head_edge_bwd", st, fE * 2.0 * (D * DH + DH * DH + DH));
        if (!(use_trr() && trr_head_edge_bwd(m, last.Mout, gA, g.ctr, g.fc, w.ypred_e, w.dfc, w.dM, E, tr ? w.hs1 : nullptr,
                                              tr ? w.hda2 : nullptr, tr ? w.hda1 : nullptr, tr ? w.hs2y : nullptr, st)))
        PET_LAUNCH_TR(tr, k_head_bwd, PET_TA(128, true), gE, lds2 + 768, st, last.Mout, wx_f(m.eh0), m.eh0.b, wx_f(m.eh2),
            m.eh2.b, wx_b(m.eh0), wx_b(m.eh2), m.ell_w, gA, g.ctr, g.fc, w.ypred_e, w.dfc, w.dM, E, tr ? w.hs1 : nullptr,
            tr ? w.hda2 : nullptr, tr ? w.hda1 : nullptr, tr ? w.hs2y : nullptr);
        if (tr) tr->heads(true, last.Mout, D, E, gA);
    }
    {
        SideStream ss = side_stream();
        if (tr) ss.enabled = false;  // the weight-gradient scratch is shared: one stream
        const hipStream_t s2 = ss.stream(st);
        ss.fork(st);
        {
            ProfScope ps("head_node_bwd", s2, fN * 2.0 * (DN * DH + DH * DH + DH));
            PET_LAUNCH_TR(tr, k_head_bwd, PET_TA(256, false), gN, (BM * LD256 + BM * LD128) * 4 + 768, s2,  last.Hout,
                wx_f(m.nh0), m.nh0.b, wx_f(m.nh2), m.nh2.b, wx_b(m.nh0), wx_b(m.nh2), m.nll_w, gA, nullptr, nullptr, nullptr,
                nullptr, w.dH, N, tr ? w.hs1 : nullptr, tr ? w.hda2 : nullptr, tr ? w.hda1 : nullptr,
                tr ? w.hs2y : nullptr);
            if (tr) tr->heads(false, last.Hout, DN, N, gA);
        }
        ss.join(st);
    }
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// Stage F: adjoint of PETBackend.calculate_features. (w.dH, w.dM) -> w.dgeo [E,4], w.dbias [E]
// Residual featuriser: one (node, edge) gradient pair per readout layer in g_node / g_edge (null entries = zero) instead
// of the seeds in w.dH / w.dM.
int backward_features(const Model& m, const Graph& g, Workspace& w, hipStream_t st, Trainer* tr = nullptr,
                      const float* const* g_node = nullptr, const float* const* g_edge = nullptr) {
    const int64_t N = g.n_nodes, E = g.n_edges, R = E + N;
    if (E == 0) return PET_OK;
    const int nt = attn_tiles(g);
    PET_REQUIRE(nt <= 8, PET_ERR_UNSUPPORTED, "more than 127 neighbours per atom is not supported yet");
    const bool post = m.post_ln(), res = m.residual(), ln = m.layer_norm();
    PET_REQUIRE(!tr || m.trainable(), PET_ERR_UNSUPPORTED,
                "training is built for transformer_type=PreLN, featurizer_type=feedforward only");
    const int nxm = ln ? 5 : 1;  // weight-gradient row source: LayerNorm-hat / RMSNorm-hat of the saved input
    PET_REQUIRE(!res || (g_node && g_edge), PET_ERR_ARGUMENT,
                "residual featuriser: the reverse pass starts from one gradient pair per readout layer "
                "(pet_backward_features_layers)");
    const float scale = 1.0f / (sqrtf((float)HD) * m.h.attention_temperature);
    const int gE = cdiv(E, BM), gN = cdiv(N, BM), gR = cdiv(R, BM);
    const size_t lds1 = BM * LD128 * 4, lds2 = 2 * BM * LD128 * 4;
    const double fE = (double)E, fN = (double)N, fR = (double)R;
    const bool trr = use_trr();
    const bool trr_l = trr && m.plain_layers();  // the TRR transformer-layer kernels are PreLN (RMSNorm or LayerNorm)
    const bool fused_attn = trr_l && !tr && ablk_bwd_on(g) && m.gnn[0].attn[0].qkv.bwd2s;
    // the forward that filled this workspace decided by itself whether Q, K, V were written: if it ran the fused block, the
    // three-kernel adjoint would read buffers nobody wrote (a switch flipped between the two calls) -- refuse
    const Graph::FwdRecord* fwd_rec = w.base ? g.fwd_record(w.base) : nullptr;
    const bool fwd_unsaved = fwd_rec && fwd_rec->attn_unsaved;
    // a forward that kept nothing (save_for_backward = 0) wrote neither [v; g] nor the compress pre-activations: no adjoint
    // can follow it, whatever the switches say
    PET_REQUIRE(!fwd_rec || fwd_rec->save != 0, PET_ERR_ARGUMENT,
                "the last forward into this workspace ran with save_for_backward = 0: nothing was kept for an adjoint");
    PET_REQUIRE(!fwd_unsaved || fused_attn, PET_ERR_ARGUMENT,
                "the forward of this workspace ran the fused attention block (Q, K, V not saved) but the adjoint is "
                "configured for the three-kernel form: pet_config_set changed between forward and backward");
    // k_dxf folded into its producer and its consumer (pet_config_set("dxf_fused", 0): the separate kernel)
    bool node_cnt_zeroed = false;  // k_node_bwd2 SPLIT: arrival counters zeroed once per adjoint, then they reset themselves
    const bool dxf_fused = trr_l && !tr && !res && !g.x_fn && m.h.num_attention_layers >= 1 && dxf_fused_on();
    PET_HIP_CHECK(hipMemsetAsync(w.dgeo, 0, E * 4 * sizeof(float), st));
    allow_big_lds(k_swiglu_bwd<256, DNF, false, true>, (BM * LD256 + BM * LD128) * 4);
    allow_big_lds(k_swiglu_bwd<256, DNF, true, true>, (BM * LD256 + BM * LD128) * 4);
    allow_big_lds(k_expand_bwd, BM * LD256 * 4);
    float* dH = w.dH;
    float* dH_alt = w.dH2;
    float* dX = w.dX;
    float* dX_alt = w.dX2;
    float* dM = w.dM;      // gradient w.r.t. the messages leaving the layer below the one being processed
    float* dM_alt = w.dM2;
    bool have_dM = !res;   // residual: no message gradient enters the last layer
    // node-feature adjoint chain on the side stream: it only meets the edge chain at output_linear^T
    // (needs dOC) and at the centre rows of the token gradient (k_center_bwd)
    SideStream ss = side_stream();
    if (tr || post || res) ss.enabled = false;
    const hipStream_t s2 = ss.stream(st);
    ss.fork(st);  // the seeds in w.dH / w.dM were produced on the main stream
    for (int gi = m.h.num_gnn_layers - 1; gi >= 0; gi--) {
        const GnnLayerW& G = m.gnn[gi];
        const GnnBufs& B = w.gnn[gi];
        // dH is the adjoint of the node features LEAVING this layer, where the system embedding was added (training:
        // the side stream is off, so dH is complete on this stream)
        if (tr) tr->cond_accumulate(dH, gi == m.h.num_gnn_layers - 1);
        if (res) {
            // backend.py:621-647: this layer's features were read out (seeds g_node / g_edge), its edge features were
            // averaged into the next layer's messages, and its node features started from a fresh embedding
            if (g_node[gi]) PET_HIP_CHECK(hipMemcpyAsync(dH, g_node[gi], N * DN * sizeof(float), hipMemcpyDeviceToDevice, st));
            else PET_HIP_CHECK(hipMemsetAsync(dH, 0, N * DN * sizeof(float), st));
            ProfScope ps("comb_bwd", st, 0.0, fE * 4.0 * 4 * D);
            k_resmix_bwd<<<cdiv(E * (D / 4), 256), 256, 0, st>>>(g_edge[gi], have_dM ? dM : nullptr, g.rev, dX, dM_alt, E);
            std::swap(dM, dM_alt);
            have_dM = true;
        } else {
            {
                ProfScope ps("comb_bwd", st, fE * 2.0 * (2 * D * 2 * D + 2 * D * D), fE * 4.0 * (3 * D + 2 * D + 2 * D));  // dM, e, e[rev], CA in; dcat out
                if (!tr && comb_bwd_s(dM, B.XF, g.rev, B.LNS, B.CA, G.comb0_g, G.comb2, w.dcat, E, dxf_fused, st)) {  // large graphs (pet_comb_bwd_s.hip)
                } else
                PET_REQUIRE(trr_comb_bwd(dM, B.XF, g, G, B.LNS, B.CA, w.dcat, E, tr ? w.dCA : nullptr, st, dxf_fused), PET_ERR_ARGUMENT,
                            "combination adjoint: the split weight planes are missing (pet_model_finalize)");
                if (tr) {
                    const std::string gs = std::to_string(gi);
                    tr->linear("combination_mlps." + gs + ".2", D, 2 * D, {dM, nullptr, 0, D},
                               {B.CA, 2 * D, 0, nullptr, nullptr}, 3, E);
                    tr->linear_after_norm("combination_mlps." + gs + ".0", G.comb0.w, 2 * D, 2 * D,
                                          {w.dCA, nullptr, 0, 2 * D}, {B.XF, D, 0, g.rev, B.LNS}, 4, E,
                                          "combination_norms." + gs + ".weight", G.ln_g,
                                          "combination_norms." + gs + ".bias", G.ln_b);
                }
            }
            if (!dxf_fused) {
                ProfScope ps("dxf", st, 0.0);
                k_dxf<<<cdiv(E * (D / 4), 256), 256, 0, st>>>(dM, w.dcat, g.rev, dX, E);
            }
            if (g.x_fn) {   // one box over several ranks: the adjoints on ghost rows go home to their owners
                int rc = exchange_backward(g, dX, gi, st);
                if (rc) return rc;
            }
        }
        for (int a = m.h.num_attention_layers - 1; a >= 0; a--) {
            const AttnLayerW& A = G.attn[a];
            const AttnBufs& Ab = B.attn[a];
            const std::string lp = "gnn_layers." + std::to_string(gi) + ".trans.layers." + std::to_string(a);
            // dX (edge rows) = grad wrt the edge MLP output; dH = grad wrt Hn
            {
                ProfScope ps("node_bwd", s2, fN * 2.0 * (D * DN + DN * 2 * DNF + DNF * DN));
                const WX wob = wx_b(A.cmlp_out), wib = wx_b(A.cmlp_in);
                bool expand_done = false;
                // large graphs: two shared-ring GEMMs and two row-wise kernels that can run BESIDE the edge kernels (pet_node_s.hip);
                // scratch: the attention-output temporary of the forward pass, which no adjoint kernel touches
                if (!tr && node_planes() && (size_t)N * 3 * DNF <= (size_t)R * D &&
                    node_bwd_s(A, dH, Ab.H1, Ab.VGn, dH_alt, w.AO, N, ln, s2)) {
                } else
                if (!tr && node_planes() && wob.h && wib.h) {
                    const int nr = node_rows(N);
                    const size_t lds_nb = (size_t)nr * LD256 * 4 + (size_t)2 * nr * plane_ld(256) * 2 + nr * 8;
                    // small graphs: four workgroups per row tile (k_node_bwd2, SPLIT); partials and arrival counters in the
                    // attention-output temporary of the forward pass, which no adjoint kernel touches
                    const int nt32 = cdiv(N, 32);
                    const size_t p_floats = (size_t)(DNF / 128) * nt32 * 32 * DN;
                    const bool split = nr == 32 && node_split_on() && nt32 <= 128 && p_floats + nt32 <= (size_t)R * D;
                    if (split) {
                        int* cnt = reinterpret_cast<int*>(w.AO + p_floats);
                        if (!node_cnt_zeroed) PET_HIP_CHECK(hipMemsetAsync(cnt, 0, nt32 * sizeof(int), s2));
                        node_cnt_zeroed = true;
                        allow_big_lds(k_node_bwd2<1, true>, lds_nb);
                        k_node_bwd2<1, true><<<dim3(nt32, DNF / 128), NTHREADS, lds_nb, s2>>>(
                            dH, Ab.H1, Ab.VGn, A.g_center, wob, wib, dH_alt, N, ln, A.ce.bwd, w.dOC, w.AO, cnt);
                        expand_done = true;
                    } else if (nr == 32) {  // the expansion adjoint in the same launch
                        allow_big_lds(k_node_bwd2<1>, lds_nb);
                        k_node_bwd2<1><<<cdiv(N, 32), NTHREADS, lds_nb, s2>>>(dH, Ab.H1, Ab.VGn, A.g_center, wob, wib, dH_alt, N, ln,
                                                                             A.ce.bwd, w.dOC, nullptr, nullptr);
                        expand_done = true;
                    } else {
                        allow_big_lds(k_node_bwd2<2>, lds_nb);
                        k_node_bwd2<2><<<gN, NTHREADS, lds_nb, s2>>>(dH, Ab.H1, Ab.VGn, A.g_center, wob, wib, dH_alt, N, ln,
                                                                     nullptr, nullptr, nullptr, nullptr);
                    }
                } else
                PET_LAUNCH_TR(tr, k_swiglu_bwd, PET_TA(256, DNF), gN, (BM * LD256 + BM * LD128) * 4, s2,  dH, Ab.H1,
                    Ab.VGn, A.g_center, A.cmlp_out.bwd, A.cmlp_in.bwd, dH_alt, N, tr ? w.dVGn : nullptr, ln);
                if (!expand_done && !(!tr && expand_bwd_s(A.ce, dH_alt, w.dOC, N, s2)))
                    k_expand_bwd<<<gN, NTHREADS, BM * LD256 * 4, s2>>>(dH_alt, A.ce.bwd, w.dOC, N);
                if (tr) {
                    tr->linear(lp + ".center_mlp.w_out", DN, DNF, {dH, nullptr, 0, DN},
                               {Ab.VGn, 2 * DNF, DNF, nullptr, nullptr}, 2, N);
                    tr->linear_after_norm(lp + ".center_mlp.w_in", A.cmlp_in.w, 2 * DNF, DN,
                                          {w.dVGn, nullptr, 0, 2 * DNF}, {Ab.H1, DN, 0, nullptr, nullptr}, nxm, N,
                                          lp + ".norm_center_features.weight", A.g_center,
                                          ln ? lp + ".norm_center_features.bias" : std::string(), ln ? A.b_center : nullptr);
                    tr->linear(lp + ".center_expansion", DN, D, {dH_alt, nullptr, 0, DN},
                               {Ab.OC, D, 0, nullptr, nullptr}, 0, N);
                }
            }
            if (post) {
                // transformer.py:245-247 backwards on every token: norm_mlp^T of [dX edge rows ; dOC], the MLP residual
                // block on normalised tokens, norm_attention^T; dX_alt = gradient of (tokens + attention output)
                ProfScope ps("emlp_bwd", st, fR * 2.0 * (D * 2 * DFF + DFF * D));
                k_rownorm_bwd<<<gR, NTHREADS, lds1, st>>>(dX, w.dOC, Ab.S2, A.g_mlp, ln, dX_alt, E, R);
                k_swiglu_bwd<128, DFF, false, false><<<gR, NTHREADS, lds2, st>>>(dX_alt, nullptr, Ab.VG, nullptr, A.mlp_out.bwd,
                                                                                 A.mlp_in.bwd, dX, R, nullptr, false);
                k_rownorm_bwd<<<gR, NTHREADS, lds1, st>>>(dX, dX + E * D, Ab.X1, A.g_attn, ln, dX_alt, E, R);
            } else {
                const bool recompute = fwd_rec ? fwd_rec->emlp_unsaved : (!tr && trr_l && emlp_recompute_on(A.mlp_in, A.mlp_out, E));
                ProfScope ps("emlp_bwd", st, fE * 2.0 * (D * 2 * DFF + DFF * D), fE * 4.0 * (3 * D + (recompute ? 0 : 2 * DFF)));  // dY, X1 (and the saved VG) in; dX1 out
                // the forward of this workspace did not save [v; g] (its record says so; without a record -- a graph handle
                // made anew for the adjoint call -- the forward followed the same switches as this call does)
                if (recompute) {
                    const bool gat = dxf_fused && a == m.h.num_attention_layers - 1;
                    PET_REQUIRE(!tr && trr_l && emlp_bwd_s(gat ? w.dcat : dX, Ab.X1, ln, A.mlp_in_g, A.mlp_out, dX_alt, E, st,
                                                          gat ? 2 * D : D, gat ? w.dcat + D : nullptr, gat ? g.rev : nullptr),
                                PET_ERR_ARGUMENT, "the forward of this workspace did not save the edge MLP's pre-activations "
                                "and the recomputing adjoint is switched off: pet_config_set changed between forward and backward");
                } else if (dxf_fused && a == m.h.num_attention_layers - 1) {
                    // dXF[p] = (dM[p] + dcat[p][:D]) + dcat[rev[p]][D:]: the bracket left k_comb_bwd_p2 in dcat's first
                    // half, the gather is made while this kernel reads its tile
                    trr_emlp_bwd(w.dcat, Ab.X1, Ab.VG, A.g_mlp, ln ? A.b_mlp : nullptr, A.mlp_in, A.mlp_out, dX_alt, E, st,
                                 nullptr, 2 * D, w.dcat + D, g.rev);
                } else if (trr_l) {
                    trr_emlp_bwd(dX, Ab.X1, Ab.VG, A.g_mlp, ln ? A.b_mlp : nullptr, A.mlp_in, A.mlp_out, dX_alt, E, st,
                                 tr ? w.dVG : nullptr);
                }
                else PET_LAUNCH_TR(tr, k_swiglu_bwd, PET_TA(128, DFF), gE, lds2, st, dX, Ab.X1, Ab.VG, A.g_mlp,
                    A.mlp_out.bwd, A.mlp_in.bwd, dX_alt, E, tr ? w.dVG : nullptr, ln);
                if (tr) {
                    tr->linear(lp + ".mlp.w_out", D, DFF, {dX, nullptr, 0, D}, {Ab.VG, 2 * DFF, DFF, nullptr, nullptr},
                               2, E);
                    tr->linear_after_norm(lp + ".mlp.w_in", A.mlp_in.w, 2 * DFF, D, {w.dVG, nullptr, 0, 2 * DFF},
                                          {Ab.X1, D, 0, nullptr, nullptr}, nxm, E, lp + ".norm_mlp.weight", A.g_mlp,
                                          ln ? lp + ".norm_mlp.bias" : std::string(), ln ? A.b_mlp : nullptr);
                }
            }
            ss.join(st);  // dOC ready
            // dX_alt (edge rows) = dX1, dH_alt = dH1; PostLN: dX_alt holds all E+N rows of d(tokens + attention output)
            const float* dOCr = post ? dX_alt + E * D : w.dOC;
            // the per-atom fused adjoint (pet_ablk.hip): Q, K, V recomputed from X, only dX and the key-bias gradient leave
            bool fusedb = false;
            if (fused_attn) {
                ProfScope ps("attn_blk_bwd", st, fR * 2.0 * D * 4 * D + 2.0 * 4.0 * D * g_sum_t2(g), fR * 4.0 * 3 * D);  // algorithmic: the adjoint's own products (the Q, K, V recomputation is this design's choice, not counted); X, dX1 in; dX out
                fusedb = ablk_bwd(m, g, A, Ab.X, dX_alt, w.dOC, dX,
                                  w.dbias_l + ((int64_t)gi * m.h.num_attention_layers + a) * NHEAD * E, scale, st);
                // (the key-bias reduction below assumes every layer took the same form)
                PET_REQUIRE(fusedb, PET_ERR_ARGUMENT, "the fused attention adjoint refused a layer (weights not packed for it)");
            }
            if (!fusedb) {
            {
                ProfScope ps("oproj_bwd", st, fR * 2.0 * D * D, fR * 4.0 * 2 * D);  // dX1 (| dOC) in; dAO out
                if (trr_l) trr_oproj_bwd(dX_alt, w.dOC, A.out, w.dAO, E, R, st);
                else k_oproj_bwd<<<gR, NTHREADS, lds1, st>>>(dX_alt, dOCr, A.out.bwd, w.dAO, E, R);
                if (tr)
                    tr->linear(lp + ".attention.output_linear", D, D, {dX_alt, w.dOC, E, D},
                               {Ab.AO, D, 0, nullptr, nullptr}, 0, R);
            }
            {
                ProfScope ps("attn_bwd", st, 2.0 * 4.0 * D * g_sum_t2(g), fR * 4.0 * (3 * D + D + 3 * D));
                float* dbias_h = w.dbias_l + ((int64_t)gi * m.h.num_attention_layers + a) * NHEAD * E;
                if (!(trr && attn_bwd_preload(nt, Ab.QKV, w.dAO, g, w.dQKV, dbias_h, scale, st))) switch (nt) {
                    case 1: launch_attn_bwd<1>(Ab.QKV, w.dAO, g, w.dQKV, dbias_h, scale, st); break;
                    case 2: launch_attn_bwd<2>(Ab.QKV, w.dAO, g, w.dQKV, dbias_h, scale, st); break;
                    case 3: launch_attn_bwd<3>(Ab.QKV, w.dAO, g, w.dQKV, dbias_h, scale, st); break;
                    case 4: launch_attn_bwd<4>(Ab.QKV, w.dAO, g, w.dQKV, dbias_h, scale, st); break;
                    case 5: case 6: launch_attn_bwd<6>(Ab.QKV, w.dAO, g, w.dQKV, dbias_h, scale, st); break;
                    default: launch_attn_bwd<8>(Ab.QKV, w.dAO, g, w.dQKV, dbias_h, scale, st); break;
                }
            }
            if (tr)
                tr->linear_after_norm(lp + ".attention.input_linear", A.qkv.w, 3 * D, D, {w.dQKV, nullptr, 0, 3 * D},
                                      {Ab.X, D, 0, nullptr, nullptr}, nxm, R, lp + ".norm_attention.weight", A.g_attn,
                                      ln ? lp + ".norm_attention.bias" : std::string(), ln ? A.b_attn : nullptr);
            {
                ProfScope ps("qkv_bwd", st, fR * 2.0 * D * 3 * D, fR * 4.0 * (3 * D + 3 * D));  // dQKV, X, dX1 in; dX out
                if (trr_l) trr_qkv_bwd(w.dQKV, Ab.X, A.g_attn, ln, A.qkv, dX_alt, dX, E, R, st);
                else if (post) k_qkv_bwd<true><<<gR, NTHREADS, lds1, st>>>(w.dQKV, nullptr, nullptr, A.qkv.bwd, dX_alt, dX, E, R, false);
                else k_qkv_bwd<false><<<gR, NTHREADS, lds1, st>>>(w.dQKV, Ab.X, A.g_attn, A.qkv.bwd, dX_alt, dX, E, R, ln);
            }
            }
            ss.fork(st);  // centre rows of dX ready
            {
                ProfScope ps("center_bwd", s2, fN * 2.0 * DN * D);
                if (!tr && center_bwd_s(A.cc, dX + E * D, dH_alt, dH, N, s2)) {
                } else if (N <= 4096) k_center_bwd<true><<<gN, NTHREADS, lds1, s2>>>(dX + E * D, dH_alt, A.cc.bwd, dH, N);
                else k_center_bwd<false><<<gN, NTHREADS, lds1, s2>>>(dX + E * D, dH_alt, A.cc.bwd, dH, N);
                if (tr)
                    tr->linear(lp + ".center_contraction", D, DN, {dX + E * D, nullptr, 0, D},
                               {Ab.H, DN, 0, nullptr, nullptr}, 0, N);
            }
            // now dX (edge rows) = grad wrt this layer's input edge tokens, dH = grad wrt its input h
        }
        {
            ProfScope ps("compress_bwd", st, fE * 2.0 * (D * D * (gi == 0 ? 3 : 4) + 4 * D));
            if (trr && trr_compress_bwd(gi == 0, dX, B.a0, G, w.dgeo, dM, E, tr ? w.da0 : nullptr, st)) {
                // TRR kernel on f16x3 (pet_trr.hip)
            } else if (gi == 0)
                PET_LAUNCH_TR(tr, k_compress_bwd, PET_TA(true), gE, lds1, st, dX, B.a0, G.compress2.bwd, G.wct,
                    nullptr, w.dgeo, nullptr, E, tr ? w.da0 : nullptr);
            else
                PET_LAUNCH_TR(tr, k_compress_bwd, PET_TA(false), gE, lds1, st, dX, B.a0, G.compress2.bwd, G.wct,
                    G.compress0_msg.bwd, w.dgeo, dM, E, tr ? w.da0 : nullptr);
            if (tr) {
                const std::string pre = "gnn_layers." + std::to_string(gi);
                tr->linear(pre + ".compress.2", D, D, {dX, nullptr, 0, D}, {B.a0, D, 0, nullptr, nullptr}, 3, E);
                tr->compress0(gi, w.da0, gi > 0 ? w.gnn[gi - 1].Mout : nullptr);
            }
        }
        // dM now holds d/dMout of layer gi-1 (pass-through + compress adjoint)
    }
    ss.join(st);
    if (tr) {
        tr->embeddings(dH, dM);
        tr->cond_finish();
        if (tr->err) return tr->err;
    }
    // fused adjoint: one slice per attention layer (the head sum), at the place of the layer's first head slice
    k_dfc_attn<<<cdiv(E, 256), 256, 0, st>>>(g.fc, w.dbias_l, m.h.num_gnn_layers * m.h.num_attention_layers, w.dbias, E,
                                             fused_attn ? NHEAD : 1);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// Stage G: adjoint of PETBackend.preprocess. (dgeo [E,4], dfc_a + dfc_b [E]) -> dL/dR, dL/dcell
int backward_geometry(const Model& m, const Graph& g, Workspace& w, const float* dgeo, const float* dfc_a,
                      const float* dfc_b, float* gpos, float* gcell, hipStream_t st) {
    const int64_t N = g.n_nodes, E = g.n_edges;
    if (E == 0) {  // isolated atoms: no position dependence at all
        PET_HIP_CHECK(hipMemsetAsync(gpos, 0, N * 3 * sizeof(float), st));
        if (gcell) PET_HIP_CHECK(hipMemsetAsync(gcell, 0, g.n_systems * 9 * sizeof(float), st));
        return PET_OK;
    }
    k_geom_bwd<<<cdiv(E, 256), 256, 0, st>>>(g.geo, g.d0, dgeo, dfc_a, dfc_b, reinterpret_cast<float4*>(w.dv), E,
                                             m.h.cutoff, m.h.cutoff_width, m.h.cutoff_function,
                                             g.adaptive ? g.pc : nullptr, g.adaptive ? g.ad_gc : nullptr);
    k_pos_grad<<<cdiv(N, 16), 256, 0, st>>>(reinterpret_cast<const float4*>(w.dv), g.rowptr, g.rev, gpos, (int)N);
    if (gcell)
        k_cell_grad<<<(int)g.n_systems, 256, 0, st>>>(reinterpret_cast<const float4*>(w.dv), g.shift, g.ctr, g.sys,
                                                      g.rowptr, gcell, (int)N, E, 0);
    if (g.adaptive) {
        k_adapt_gr<<<cdiv(N, 16), 256, 0, st>>>(g.ad_gc, g.rowptr, g.rev, g.ad_gr, (int)N);
        if (g.grid_probes > 0)
            k_adapt_dv_grid<<<cdiv(N, 16), 256, 0, st>>>(g.rowptr0, g.perm0, g.vin, g.ad_gr, g.grid_drdn, g.ad_dv, (int)N,
                                                         m.h.cutoff_width_adaptive, g.grid_probes, 0.5f,
                                                         m.h.cutoff_width_adaptive / 4.0f);
        else
        k_adapt_dv<<<cdiv(N, 16), 256, 0, st>>>(g.rowptr0, g.perm0, g.vin, g.ad_gr, g.r_newton, g.inv_dn, g.ad_dv,
                                                (int)N, m.h.cutoff_width_adaptive);
        k_pos_grad_acc<<<cdiv(N, 16), 256, 0, st>>>(g.ad_dv, g.rowptr0, g.rev0, gpos, (int)N);
        if (gcell)  // the adaptive term reaches the cell through the shift vectors of ALL input edges
            k_cell_grad<<<(int)g.n_systems, 256, 0, st>>>(g.ad_dv, g.shift0, g.ctr, g.sys, g.rowptr0, gcell, (int)N,
                                                          g.n_edges_in, 1);
    }
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// Gn[i][c] = sum_p gA[i][p] Wn[p][c], Ge likewise with We, gb[i] = sum_p gA[i][p] be[p]
__global__ void k_last_bwd(const float* __restrict__ gA, const float* __restrict__ nw, const float* __restrict__ ew,
                           const float* __restrict__ eb, int P, float* __restrict__ Gn, float* __restrict__ Ge,
                           float* __restrict__ gbv, int n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * DH) return;
    const int i = (int)(idx / DH), c = (int)(idx % DH);
    float a = 0.f, b = 0.f, s = 0.f;
    for (int p = 0; p < P; p++) {
        const float gv = gA[(int64_t)i * P + p];
        a = fmaf(gv, nw[(int64_t)p * DH + c], a);
        b = fmaf(gv, ew[(int64_t)p * DH + c], b);
        s = fmaf(gv, eb[p], s);
    }
    Gn[idx] = a;
    Ge[idx] = b;
    if (c == 0) gbv[i] = s;
}

// adjoint of predict() (pet_fwd.hip) for the features the caller passes in
int predict_backward(const Model& m, const Graph& g, const HeadW& H, const LastW& Lw, const float* node_feat,
                     const float* edge_feat, const float* fc, const float* gA, float* g_node, float* g_edge, float* g_fc,
                     float* scratch, hipStream_t st) {
    if (m.generic()) return gen_predict_backward(m, g, H, Lw, node_feat, edge_feat, fc, gA, g_node, g_edge, g_fc, st);
    const int64_t N = g.n_nodes, E = g.n_edges;
    if (N == 0) return PET_OK;
    float* Gn = scratch;
    float* Ge = scratch + N * DH;
    float* gbv = Ge + N * DH;
    if (!fc) fc = g.fc;
    const int gE = cdiv(E, BM), gN = cdiv(N, BM);
    const size_t ldsn = (BM * LD256 + BM * LD128) * 4 + 768 + BM * 4, ldse = 2 * BM * LD128 * 4 + 768 + BM * 4;
    allow_big_lds(k_head_bwd<256, false, false, true>, ldsn);
    allow_big_lds(k_head_bwd<128, true, false, true>, ldse);
    k_last_bwd<<<cdiv(N * DH, 256), 256, 0, st>>>(gA, Lw.nw, Lw.ew, Lw.eb, Lw.P, Gn, Ge, gbv, (int)N);
    k_head_bwd<256, false, false, true><<<gN, NTHREADS, ldsn, st>>>(
        node_feat, wx_f(H.nh0), H.nh0.b, wx_f(H.nh2), H.nh2.b, wx_b(H.nh0), wx_b(H.nh2), nullptr, nullptr, nullptr, nullptr,
        nullptr, nullptr, g_node, N, nullptr, nullptr, nullptr, nullptr, Gn, gbv);
    if (E > 0)
        k_head_bwd<128, true, false, true><<<gE, NTHREADS, ldse, st>>>(
            edge_feat, wx_f(H.eh0), H.eh0.b, wx_f(H.eh2), H.eh2.b, wx_b(H.eh0), wx_b(H.eh2), nullptr, nullptr, g.ctr, fc,
            nullptr, g_fc, g_edge, E, nullptr, nullptr, nullptr, nullptr, Ge, gbv);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

#define PET_CARVE(w)                                                                      \
    Workspace w;                                                                          \
    carve_workspace(m, g.n_nodes, g.n_edges, ws, w);                                      \
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "workspace too small")

static int d2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes) PET_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    return PET_OK;
}

int backward_predict_abi(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* gA,
                         float* g_node, float* g_edge, float* g_fc, hipStream_t st) {
    if (use_generic(m, g) || generic_workspace(g, ws)) return gen_backward_predict(m, g, ws, ws_bytes, gA, g_node, g_edge, g_fc, st);
    PET_CARVE(w);
    if (g.n_nodes == 0) return PET_OK;
    int rc;
    if ((rc = backward_predict(m, g, w, gA, st))) return rc;
    if (g_node && (rc = d2d(g_node, w.dH, g.n_nodes * DN * sizeof(float), st))) return rc;
    if (g_edge && (rc = d2d(g_edge, w.dM, g.n_edges * D * sizeof(float), st))) return rc;
    if (g_fc && (rc = d2d(g_fc, w.dfc, g.n_edges * sizeof(float), st))) return rc;
    return PET_OK;
}

int backward_features_abi(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* g_node,
                          const float* g_edge, float* g_geo, float* g_fc, hipStream_t st) {
    if (use_generic(m, g) || generic_workspace(g, ws)) return gen_backward_features(m, g, ws, ws_bytes, &g_node, &g_edge, 1, g_geo, g_fc, st);
    PET_CARVE(w);
    if (g.n_nodes == 0) return PET_OK;
    int rc;
    if ((rc = d2d(w.dH, g_node, g.n_nodes * DN * sizeof(float), st))) return rc;
    if ((rc = d2d(w.dM, g_edge, g.n_edges * D * sizeof(float), st))) return rc;
    if ((rc = backward_features(m, g, w, st))) return rc;
    if ((rc = d2d(g_geo, w.dgeo, g.n_edges * 4 * sizeof(float), st))) return rc;
    if ((rc = d2d(g_fc, w.dbias, g.n_edges * sizeof(float), st))) return rc;
    return PET_OK;
}

int backward_features_layers_abi(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* const* g_node,
                                 const float* const* g_edge, int n_layers, float* g_geo, float* g_fc, hipStream_t st) {
    if (use_generic(m, g) || generic_workspace(g, ws)) return gen_backward_features(m, g, ws, ws_bytes, g_node, g_edge, n_layers, g_geo, g_fc, st);
    PET_CARVE(w);
    PET_REQUIRE(n_layers == m.num_readout_layers(), PET_ERR_ARGUMENT,
                "expected one gradient pair per readout layer (" + std::to_string(m.num_readout_layers()) + ")");
    if (g.n_nodes == 0) return PET_OK;
    int rc;
    if (!m.residual()) {
        if (g_node[0]) { if ((rc = d2d(w.dH, g_node[0], g.n_nodes * DN * sizeof(float), st))) return rc; }
        else PET_HIP_CHECK(hipMemsetAsync(w.dH, 0, g.n_nodes * DN * sizeof(float), st));
        if (g_edge[0]) { if ((rc = d2d(w.dM, g_edge[0], g.n_edges * D * sizeof(float), st))) return rc; }
        else if (g.n_edges) PET_HIP_CHECK(hipMemsetAsync(w.dM, 0, g.n_edges * D * sizeof(float), st));
        if ((rc = backward_features(m, g, w, st))) return rc;
    } else if ((rc = backward_features(m, g, w, st, nullptr, g_node, g_edge))) return rc;
    if ((rc = d2d(g_geo, w.dgeo, g.n_edges * 4 * sizeof(float), st))) return rc;
    if ((rc = d2d(g_fc, w.dbias, g.n_edges * sizeof(float), st))) return rc;
    return PET_OK;
}

int backward_geometry_abi(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* g_geo,
                          const float* g_fc, float* gpos, float* gcell, hipStream_t st) {
    if (use_generic(m, g) || generic_workspace(g, ws)) return gen_backward_geometry(m, g, ws, ws_bytes, g_geo, g_fc, gpos, gcell, st);
    PET_CARVE(w);
    if (g.n_nodes == 0) return PET_OK;
    return backward_geometry(m, g, w, g_geo, g_fc, nullptr, gpos, gcell, st);
}

// preprocess^T on its own: only the d/d(edge vector) buffer is needed, not a forward workspace
int geometry_backward(const Model& m, const Graph& g, const float* g_geo, const float* g_fc, float* gpos, float* gcell,
                      float* scratch, hipStream_t st) {
    Workspace w;
    w.dv = scratch;
    if (g.n_nodes == 0) return PET_OK;
    return backward_geometry(m, g, w, g_geo, g_fc, nullptr, gpos, gcell, st);
}

// shared by the size-generic path (gen.hip): only the d/d(edge vector) buffer is needed
int backward_geometry_generic(const Model& m, const Graph& g, float* dv_scratch, const float* dgeo, const float* dfc_a,
                              const float* dfc_b, float* gpos, float* gcell, hipStream_t st) {
    Workspace w;
    w.dv = dv_scratch;
    return backward_geometry(m, g, w, dgeo, dfc_a, dfc_b, gpos, gcell, st);
}

int backward(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* gA, float* gpos,
             float* gcell, hipStream_t st) {
    if (use_generic(m, g) || generic_workspace(g, ws)) return gen_backward(m, g, ws, ws_bytes, gA, gpos, gcell, st);
    Workspace w;
    carve_workspace(m, g.n_nodes, g.n_edges, ws, w);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "workspace too small");
    if (g.n_nodes == 0) return PET_OK;
    int rc;
    if (g.n_edges > 0) {
        if ((rc = backward_predict(m, g, w, gA, st))) return rc;
        if ((rc = backward_features(m, g, w, st))) return rc;
    }
    return backward_geometry(m, g, w, w.dgeo, w.dfc, w.dbias, gpos, gcell, st);
}

// Training reverse pass (energy term): dL/dtheta accumulated into the model's flat gradient
// buffer for the seeds gA = dL/d(atomic prediction), plus dL/dR (and dL/dcell) when requested.
// Needs a forward run with save_for_backward = 2 on a training workspace.
int backward_train(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* gA, float* gpos,
                   float* gcell, hipStream_t st) {
    // other sizes, PostLN, residual, and any graph with an atom of more than 127 neighbours: the energy term alone = the
    // size-generic second-order pass without a tangent
    if (train_generic_for(m, g)) {
        PET_REQUIRE(generic_workspace(g, ws), PET_ERR_ARGUMENT, "pet_forward with save_for_backward = 2 has not run on this workspace");
        // (the energy-only step has no second-order workspace of its own in the ABI: the dual activations come from the
        // stream's pool and go back to it on every exit)
        const int64_t n2 = gen_train_workspace_bytes(m, g.n_nodes, g.n_edges);
        PoolBuf ws2_pool;
        PET_HIP_CHECK(ws2_pool.alloc((size_t)n2, st));
        void* ws2 = ws2_pool.p;
        int rc = gen_train2(m, g, ws2, n2, nullptr, gA, nullptr, nullptr, nullptr, st);
        if (!rc && gpos) rc = gen_backward(m, g, ws, ws_bytes, gA, gpos, gcell, st);
        return rc;
    }
    PET_REQUIRE(m.grad_flat, PET_ERR_ARGUMENT, "pet_model_zero_grad has not been called");
    Workspace w;
    carve_workspace(m, g.n_nodes, g.n_edges, ws, w, true);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "workspace too small for training");
    if (g.n_nodes == 0) return PET_OK;
    PET_REQUIRE(g.n_edges > 0, PET_ERR_UNSUPPORTED, "training on a batch without any edge is not supported");
    Trainer tr{m, g, w, m.grad_flat, st};
    int rc;
    if ((rc = backward_predict(m, g, w, gA, st, &tr))) return rc;
    if ((rc = backward_features(m, g, w, st, &tr))) return rc;
    if (tr.err) return tr.err;
    if (gpos) return backward_geometry(m, g, w, w.dgeo, w.dfc, w.dbias, gpos, gcell, st);
    return PET_OK;
}

}  // namespace pet
