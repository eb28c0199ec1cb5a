// Shared building blocks of the size-generic PET path (gen.hip: inference + dE/dR; gen_train.hip: training): fp32 FMA
// GEMM over raw torch weights, run-time-width row kernels, wave-per-(atom, head) attention, workspace carve. Everything
// here lives in an anonymous namespace: each translation unit gets its own copy of the kernels.
#pragma once
#include <type_traits>

#include "common.h"
#include "model.h"

namespace pet {

namespace {

struct GD {
    int D, DN, DFF, DNF, DH, NH, HD;
    bool expanded;  // d_node != d_pet (transformer.py:189-201)
};
static GD dims_of(const Model& m) {
    GD d;
    d.D = m.h.d_pet; d.DN = m.h.d_node; d.DFF = m.h.d_feedforward; d.DNF = 2 * d.DN; d.DH = m.h.d_head;
    d.NH = m.h.num_heads; d.HD = d.D / d.NH; d.expanded = d.DN != d.D;
    return d;
}

// ---------------------------------------------------------------------------------------------
// Y[r][o] (+)= b[o] + sum_i X[r * ldx + i] * W[o * so + i * si]
// ---------------------------------------------------------------------------------------------
// 64 x 64 output tile per workgroup, one 32 x 32 quadrant per wave on the fp32 matrix core (v_mfma_f32_32x32x2_f32: exact fp32
// products, fp32 accumulation -- no operand splitting, so adjoint rows of any magnitude need no scaling); K in chunks of 32
// through LDS ([64][33] floats per operand: a lane reads row (lane & 31), column 2 s + (lane >> 5) -- stride 33, no bank
// conflicts). Any R, NO, KI (tails are zero-filled) and either orientation of the raw torch weight (so, si).
// V4: 16-byte global loads (K, the strides and NO multiples of 4, 16-byte aligned bases: every size but the odd ones)
template <bool ACC, bool V4>
__global__ __launch_bounds__(256) void k_gen_lin(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                                                 int64_t so, int64_t si, const float* __restrict__ b,
                                                 float* __restrict__ Y, int64_t ldy, int64_t R, int NO, int KI) {
    constexpr int KC = 32, LD = KC + 1;
    __shared__ float Xs[64 * LD], Ws[64 * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = wave & 1, cb = wave >> 1;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int o0 = blockIdx.y * 64;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    const float* xa = Xs + (rb * 32 + (lane & 31)) * LD + (lane >> 5);
    const float* wa = Ws + (cb * 32 + (lane & 31)) * LD + (lane >> 5);
    // register prefetch: the global loads of chunk k0 + KC are in flight during the MFMAs of chunk k0
    float xr[8], wr[8];
    auto fetch = [&](int k0) {
        if constexpr (V4) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int idx = threadIdx.x + 256 * q;
                {
                    const int rr = idx >> 3, k4 = 4 * (idx & 7);
                    const int64_t r = r0 + rr;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (r < R && k0 + k4 < KI) v = *reinterpret_cast<const float4*>(X + r * ldx + k0 + k4);
                    xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
                }
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (si == 1) {  // W[o][k]: four k of one output row
                    const int o = o0 + (idx >> 3), k4 = 4 * (idx & 7);
                    if (o < NO && k0 + k4 < KI) v = *reinterpret_cast<const float4*>(W + (int64_t)o * so + k0 + k4);
                } else {        // transposed: four output rows of one k
                    const int o = o0 + 4 * (idx & 15), kk = idx >> 4;
                    if (o < NO && k0 + kk < KI) v = *reinterpret_cast<const float4*>(W + (int64_t)(k0 + kk) * si + o);
                }
                wr[4 * q] = v.x; wr[4 * q + 1] = v.y; wr[4 * q + 2] = v.z; wr[4 * q + 3] = v.w;
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = threadIdx.x + 256 * q;
            {   // X rows: k fastest (a row's 128 B are contiguous)
                const int rr = idx >> 5, kk = idx & 31;
                const int64_t r = r0 + rr;
                xr[q] = (r < R && k0 + kk < KI) ? X[r * ldx + k0 + kk] : 0.f;
            }
            // W: forward orientation W[o][k] (si == 1): k fastest; transposed (so == 1): o fastest
            const int rr = si == 1 ? idx >> 5 : idx & 63, kk = si == 1 ? idx & 31 : idx >> 6;
            const int o = o0 + rr;
            wr[q] = (o < NO && k0 + kk < KI) ? W[(int64_t)o * so + (int64_t)(k0 + kk) * si] : 0.f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < KI; k0 += KC) {
        __syncthreads();  // the previous chunk's MFMAs have read the tiles
        if constexpr (V4) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int idx = threadIdx.x + 256 * q;
                float* xd = Xs + (idx >> 3) * LD + 4 * (idx & 7);
#pragma unroll
                for (int j = 0; j < 4; j++) xd[j] = xr[4 * q + j];
                if (si == 1) {
                    float* wd = Ws + (idx >> 3) * LD + 4 * (idx & 7);
#pragma unroll
                    for (int j = 0; j < 4; j++) wd[j] = wr[4 * q + j];
                } else {
                    float* wd = Ws + 4 * (idx & 15) * LD + (idx >> 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) wd[j * LD] = wr[4 * q + j];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int idx = threadIdx.x + 256 * q;
                Xs[(idx >> 5) * LD + (idx & 31)] = xr[q];
                const int rr = si == 1 ? idx >> 5 : idx & 63, kk = si == 1 ? idx & 31 : idx >> 6;
                Ws[rr * LD + kk] = wr[q];
            }
        }
        __syncthreads();
        if (k0 + KC < KI) fetch(k0 + KC);
#pragma unroll
        for (int s2 = 0; s2 < KC / 2; s2++)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * s2], wa[2 * s2], acc, 0, 0, 0);
    }
    const int o = o0 + cb * 32 + (lane & 31);
    if (o >= NO) return;
    const float bo = b ? b[o] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int64_t r = r0 + rb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        if (r >= R) continue;
        const float v = acc[i] + bo;
        if (ACC) Y[r * ldy + o] += v;
        else Y[r * ldy + o] = v;
    }
}

struct Lins {
    hipStream_t st;
    void launch(dim3 grid, bool acc, const float* X, int64_t ldx, const float* W, int64_t so, int64_t si, const float* b,
                float* Y, int64_t ldy, int64_t R, int NO, int KI) const {
        const bool v4 = KI % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
                        ((si == 1 && so % 4 == 0) || (so == 1 && si % 4 == 0 && NO % 4 == 0));
        if (acc) {
            if (v4) k_gen_lin<true, true><<<grid, 256, 0, st>>>(X, ldx, W, so, si, b, Y, ldy, R, NO, KI);
            else k_gen_lin<true, false><<<grid, 256, 0, st>>>(X, ldx, W, so, si, b, Y, ldy, R, NO, KI);
        } else {
            if (v4) k_gen_lin<false, true><<<grid, 256, 0, st>>>(X, ldx, W, so, si, b, Y, ldy, R, NO, KI);
            else k_gen_lin<false, false><<<grid, 256, 0, st>>>(X, ldx, W, so, si, b, Y, ldy, R, NO, KI);
        }
    }
    // y = x W^T + b
    void fwd(const float* X, int64_t ldx, const Lin& L, float* Y, int64_t ldy, int64_t R, bool acc = false,
             int col0 = 0, int kin = -1) const {
        if (R <= 0) return;
        const int K = kin < 0 ? L.k_in : kin;  // a column block [col0, col0 + K) of the weight (compress.0)
        dim3 grid((unsigned)cdiv(R, 64), (unsigned)cdiv(L.n_out, 64));
        launch(grid, acc, X, ldx, L.w + col0, L.k_in, 1, L.b, Y, ldy, R, L.n_out, K);
    }
    // dx (+)= dy W
    void bwd(const float* dY, int64_t ldy, const Lin& L, float* dX, int64_t ldx, int64_t R, bool acc = false,
             int col0 = 0, int kin = -1) const {
        if (R <= 0) return;
        const int K = kin < 0 ? L.k_in : kin;
        dim3 grid((unsigned)cdiv(R, 64), (unsigned)cdiv(K, 64));
        launch(grid, acc, dY, ldy, L.w + col0, 1, L.k_in, nullptr, dX, ldx, R, K, L.n_out);
    }
};

// ---------------------------------------------------------------------------------------------
// row kernels, one wave per row, run-time width
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float gsig(float x) { return 1.0f / (1.0f + expf(-x)); }

// torch.nn.RMSNorm (eps = finfo(float32).eps, weight) or torch.nn.LayerNorm (eps 1e-5, weight + bias)
__global__ void k_gen_norm(const float* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta,
                           int ln, float eps, float* __restrict__ Y, int64_t R, int W) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* x = X + r * W;
    float mean = 0.f;
    if (ln) {
        float s = 0.f;
        for (int k = lane; k < W; k += 64) s += x[k];
        mean = wave_sum(s) / W;
    }
    float s2 = 0.f;
    for (int k = lane; k < W; k += 64) { const float c = x[k] - mean; s2 += c * c; }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / W + eps);
    for (int k = lane; k < W; k += 64) Y[r * W + k] = (x[k] - mean) * rstd * gamma[k] + (beta ? beta[k] : 0.f);
}

// dX (+)= adjoint of k_gen_norm at X for the incoming dY
__global__ void k_gen_norm_bwd(const float* __restrict__ X, const float* __restrict__ gamma, int ln, float eps,
                               const float* __restrict__ dY, float* __restrict__ dX, int acc, int64_t R, int W) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const float* x = X + r * W;
    const float* dy = dY + r * W;
    float mean = 0.f;
    if (ln) {
        float s = 0.f;
        for (int k = lane; k < W; k += 64) s += x[k];
        mean = wave_sum(s) / W;
    }
    float s2 = 0.f;
    for (int k = lane; k < W; k += 64) { const float c = x[k] - mean; s2 += c * c; }
    const float rstd = 1.0f / sqrtf(wave_sum(s2) / W + eps);
    float m1 = 0.f, m2 = 0.f;
    for (int k = lane; k < W; k += 64) {
        const float gk = dy[k] * gamma[k];
        m1 += gk;
        m2 += gk * (x[k] - mean) * rstd;
    }
    m1 = ln ? wave_sum(m1) / W : 0.f;
    m2 = wave_sum(m2) / W;
    for (int k = lane; k < W; k += 64) {
        const float v = rstd * (dy[k] * gamma[k] - m1 - (x[k] - mean) * rstd * m2);
        if (acc) dX[r * W + k] += v;
        else dX[r * W + k] = v;
    }
}

// FeedForward (transformer.py:39-50): S = v * sigmoid(g), [v | g] = VG [R, 2F]
__global__ void k_gen_swiglu(const float* __restrict__ VG, float* __restrict__ S, int64_t R, int F) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * F) return;
    const int64_t r = idx / F;
    const int k = (int)(idx % F);
    S[idx] = VG[r * 2 * F + k] * gsig(VG[r * 2 * F + F + k]);
}
__global__ void k_gen_swiglu_bwd(const float* __restrict__ VG, const float* __restrict__ dS, float* __restrict__ dVG,
                                 int64_t R, int F) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * F) return;
    const int64_t r = idx / F;
    const int k = (int)(idx % F);
    const float v = VG[r * 2 * F + k], s = gsig(VG[r * 2 * F + F + k]), d = dS[idx];
    dVG[r * 2 * F + k] = d * s;
    dVG[r * 2 * F + F + k] = d * v * s * (1.f - s);
}
__global__ void k_gen_silu(const float* __restrict__ A, float* __restrict__ S, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) S[idx] = A[idx] * gsig(A[idx]);
}
// dA = dS * silu'(A)   (in place on dS allowed)
__global__ void k_gen_silu_bwd(const float* __restrict__ A, const float* __restrict__ dS, float* __restrict__ dA, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float a = A[idx], s = gsig(a);
    dA[idx] = dS[idx] * s * (1.f + a * (1.f - s));
}
// Y[r][0..W) (+)= a * A[r * lda + ..] + b * B[rowB(r) * ldb + ..]   (B, index optional)
__global__ void k_gen_axpby(float a, const float* __restrict__ A, int64_t lda, float b, const float* __restrict__ B,
                            int64_t ldb, const int* __restrict__ index, float* __restrict__ Y, int64_t ldy, int acc,
                            int64_t R, int W) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * W) return;
    const int64_t r = idx / W;
    const int k = (int)(idx % W);
    float v = A ? a * A[r * lda + k] : 0.f;
    if (B) v += b * B[(index ? (int64_t)index[r] : r) * ldb + k];
    if (acc) Y[r * ldy + k] += v;
    else Y[r * ldy + k] = v;
}
// Y[r][..] = table[index[r]][..]
__global__ void k_gen_embed(const int* __restrict__ index, const float* __restrict__ table, float* __restrict__ Y,
                            int64_t ldy, int64_t R, int W) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * W) return;
    const int64_t r = idx / W;
    const int k = (int)(idx % W);
    Y[r * ldy + k] = table[(int64_t)index[r] * W + k];
}
// out[a][..] += cond[system of atom a][..]
__global__ void k_gen_add_cond(float* __restrict__ H, const float* __restrict__ cond, const int* __restrict__ sys32,
                               const int64_t* __restrict__ sys64, int64_t N, int W) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * W) return;
    const int64_t a = idx / W;
    const int64_t s = sys64 ? sys64[a] : (int64_t)sys32[a];
    H[idx] += cond[s * W + (idx % W)];
}
// conditioning.py:82-100 for one system per block: silu(project.0 [emb_q ; emb_s]) -> project.2
__global__ void k_gen_system_cond(const int64_t* __restrict__ charge, const int64_t* __restrict__ spin,
                                  const float* __restrict__ qe, const float* __restrict__ se, const float* __restrict__ w0,
                                  const float* __restrict__ b0, const float* __restrict__ w2, const float* __restrict__ b2,
                                  float* __restrict__ out, int max_charge, int DN) {
    extern __shared__ float sm[];  // [2 DN] input, [DN] hidden
    float* x = sm;
    float* hdn = sm + 2 * DN;
    const int s = blockIdx.x;
    const int64_t q = charge[s] + max_charge, mult = spin[s] - 1;
    for (int k = threadIdx.x; k < DN; k += blockDim.x) { x[k] = qe[q * DN + k]; x[DN + k] = se[mult * DN + k]; }
    __syncthreads();
    for (int o = threadIdx.x; o < DN; o += blockDim.x) {
        float a = b0[o];
        for (int k = 0; k < 2 * DN; k++) a = fmaf(w0[(int64_t)o * 2 * DN + k], x[k], a);
        hdn[o] = a * gsig(a);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < DN; o += blockDim.x) {
        float a = b2[o];
        for (int k = 0; k < DN; k++) a = fmaf(w2[(int64_t)o * DN + k], hdn[k], a);
        out[(int64_t)s * DN + o] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// attention (transformer.py:86-152, 565-589): tokens of atom i = [centre row E + i ; its CSR edge rows], key bias
// log(max(fc, 1e-15)) on edge keys (0 for the centre), scale 1 / (sqrt(head_dim) temperature)
// ---------------------------------------------------------------------------------------------
// One wave per (atom, head). A lane is (token, feature slice): the head's features are cut into SL slices of DS <= 16, so a
// lane holds DS values of every row it touches (registers stay small at any head dimension) and a pass serves TPW = 64 / SL
// tokens (head dimension 64: four slices, 16 tokens per pass -- with ~20 tokens per atom most lanes work, where one lane per
// token with the whole row in registers left two thirds of them idle and ran at one wave per SIMD). Dot products are the
// slices' partial sums added across the SL lanes of a token by xor shuffles.
template <int HDM>
struct GAttnLanes {
    static constexpr int DS = HDM < 16 ? HDM : 16, SL = HDM / DS, TPW = 64 / SL;
};
// DS values of a row slice: float4 pieces when the layout allows it (v4), guarded scalars otherwise; zeros past nvalid
template <int DS>
__device__ __forceinline__ void gen_ld_slice(float (&o)[DS], const float* __restrict__ p, int nvalid, bool v4) {
    if (v4) {
#pragma unroll
        for (int g = 0; g < DS / 4; g++) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * g < nvalid) t = *reinterpret_cast<const float4*>(p + 4 * g);
            o[4 * g] = t.x; o[4 * g + 1] = t.y; o[4 * g + 2] = t.z; o[4 * g + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < DS; j++) o[j] = j < nvalid ? p[j] : 0.f;
    }
}
template <int SL, int TPW>
__device__ __forceinline__ float gen_slice_sum(float v) {
#pragma unroll
    for (int o = TPW; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

template <int HDM>
__global__ __launch_bounds__(64) void k_gen_attn_fwd(const float* __restrict__ QKV, const int* __restrict__ rowptr,
                                                     const float* __restrict__ fc, float* __restrict__ AO,
                                                     float* __restrict__ LSE, int64_t E, int D, int NH, int HD, float scale) {
    using GL = GAttnLanes<HDM>;
    constexpr int DS = GL::DS, SL = GL::SL, TPW = GL::TPW;
    const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int tok = lane % TPW, d0 = DS * (lane / TPW);
    const int nv = HD - d0 < 0 ? 0 : (HD - d0 < DS ? HD - d0 : DS);
    const bool v4 = DS % 4 == 0 && (HD & 3) == 0 && (D & 3) == 0;
    const int p0 = rowptr[i], T = rowptr[i + 1] - p0 + 1;
    const int64_t ld = 3 * (int64_t)D;
    for (int t0 = 0; t0 < T; t0 += TPW) {
        const int tq = t0 + tok;
        const bool live = tq < T;
        const int64_t rq = !live ? E + i : (tq == 0 ? E + i : (int64_t)p0 + tq - 1);
        float q[DS], acc[DS];
        gen_ld_slice<DS>(q, QKV + rq * ld + h * HD + d0, nv, v4);
#pragma unroll
        for (int j = 0; j < DS; j++) { q[j] *= scale; acc[j] = 0.f; }
        float mx = -INFINITY, l = 0.f;
        for (int tk = 0; tk < T; tk++) {
            const int64_t rk = tk == 0 ? E + i : (int64_t)p0 + tk - 1;
            float kk[DS], vv[DS];
            gen_ld_slice<DS>(kk, QKV + rk * ld + D + h * HD + d0, nv, v4);
            gen_ld_slice<DS>(vv, QKV + rk * ld + 2 * D + h * HD + d0, nv, v4);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < DS; j++) s = fmaf(q[j], kk[j], s);
            s = gen_slice_sum<SL, TPW>(s) + (tk == 0 ? 0.f : logf(fmaxf(fc[p0 + tk - 1], 1e-15f)));
            const float mn = fmaxf(mx, s), c = expf(mx - mn), p = expf(s - mn);
            l = l * c + p;
#pragma unroll
            for (int j = 0; j < DS; j++) acc[j] = acc[j] * c + p * vv[j];
            mx = mn;
        }
        if (live) {
            const float il = 1.0f / l;
#pragma unroll
            for (int j = 0; j < DS; j++)
                if (j < nv) AO[rq * D + h * HD + d0 + j] = acc[j] * il;
            if (d0 == 0) LSE[rq * NH + h] = mx + logf(l);
        }
    }
}

// pass A, lanes = (query, slice): delta = <dO, O>, dQ
template <int HDM>
__global__ __launch_bounds__(64) void k_gen_attn_bwd_q(const float* __restrict__ QKV, const float* __restrict__ AO,
                                                       const float* __restrict__ dAO, const float* __restrict__ LSE,
                                                       const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                       float* __restrict__ dQKV, float* __restrict__ DELTA, int64_t E,
                                                       int D, int NH, int HD, float scale) {
    using GL = GAttnLanes<HDM>;
    constexpr int DS = GL::DS, SL = GL::SL, TPW = GL::TPW;
    const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int tok = lane % TPW, d0 = DS * (lane / TPW);
    const int nv = HD - d0 < 0 ? 0 : (HD - d0 < DS ? HD - d0 : DS);
    const bool v4 = DS % 4 == 0 && (HD & 3) == 0 && (D & 3) == 0;
    const int p0 = rowptr[i], T = rowptr[i + 1] - p0 + 1;
    const int64_t ld = 3 * (int64_t)D;
    for (int t0 = 0; t0 < T; t0 += TPW) {
        const int tq = t0 + tok;
        const bool live = tq < T;
        const int64_t rq = !live ? E + i : (tq == 0 ? E + i : (int64_t)p0 + tq - 1);
        float q[DS], dO[DS], dq[DS], ao[DS];
        gen_ld_slice<DS>(q, QKV + rq * ld + h * HD + d0, nv, v4);
        gen_ld_slice<DS>(dO, dAO + rq * D + h * HD + d0, nv, v4);
        gen_ld_slice<DS>(ao, AO + rq * D + h * HD + d0, nv, v4);
        float delta = 0.f;
#pragma unroll
        for (int j = 0; j < DS; j++) { q[j] *= scale; dq[j] = 0.f; delta = fmaf(dO[j], ao[j], delta); }
        delta = gen_slice_sum<SL, TPW>(delta);
        const float lse = LSE[rq * NH + h];
        for (int tk = 0; tk < T; tk++) {
            const int64_t rk = tk == 0 ? E + i : (int64_t)p0 + tk - 1;
            float kk[DS], vv[DS];
            gen_ld_slice<DS>(kk, QKV + rk * ld + D + h * HD + d0, nv, v4);
            gen_ld_slice<DS>(vv, QKV + rk * ld + 2 * D + h * HD + d0, nv, v4);
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int j = 0; j < DS; j++) { s = fmaf(q[j], kk[j], s); dp = fmaf(dO[j], vv[j], dp); }
            s = gen_slice_sum<SL, TPW>(s) + (tk == 0 ? 0.f : logf(fmaxf(fc[p0 + tk - 1], 1e-15f)));
            dp = gen_slice_sum<SL, TPW>(dp);
            const float ds = expf(s - lse) * (dp - delta);
#pragma unroll
            for (int j = 0; j < DS; j++) dq[j] = fmaf(ds, kk[j], dq[j]);
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < DS; j++)
                if (j < nv) dQKV[rq * ld + h * HD + d0 + j] = dq[j] * scale;
            if (d0 == 0) DELTA[rq * NH + h] = delta;
        }
    }
}

// pass B, lanes = (key, slice): dK, dV and the key-bias gradient (edge keys; head-major [NH][E])
template <int HDM>
__global__ __launch_bounds__(64) void k_gen_attn_bwd_k(const float* __restrict__ QKV, const float* __restrict__ dAO,
                                                       const float* __restrict__ LSE, const float* __restrict__ DELTA,
                                                       const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                       float* __restrict__ dQKV, float* __restrict__ dbias_h, int64_t E,
                                                       int D, int NH, int HD, float scale) {
    using GL = GAttnLanes<HDM>;
    constexpr int DS = GL::DS, SL = GL::SL, TPW = GL::TPW;
    const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int tok = lane % TPW, d0 = DS * (lane / TPW);
    const int nv = HD - d0 < 0 ? 0 : (HD - d0 < DS ? HD - d0 : DS);
    const bool v4 = DS % 4 == 0 && (HD & 3) == 0 && (D & 3) == 0;
    const int p0 = rowptr[i], T = rowptr[i + 1] - p0 + 1;
    const int64_t ld = 3 * (int64_t)D;
    for (int t0 = 0; t0 < T; t0 += TPW) {
        const int tk = t0 + tok;
        const bool live = tk < T;
        const int64_t rk = !live ? E + i : (tk == 0 ? E + i : (int64_t)p0 + tk - 1);
        float k[DS], v[DS], dk[DS], dv[DS];
        gen_ld_slice<DS>(k, QKV + rk * ld + D + h * HD + d0, nv, v4);
        gen_ld_slice<DS>(v, QKV + rk * ld + 2 * D + h * HD + d0, nv, v4);
#pragma unroll
        for (int j = 0; j < DS; j++) { dk[j] = 0.f; dv[j] = 0.f; }
        const float bias = (!live || tk == 0) ? 0.f : logf(fmaxf(fc[p0 + tk - 1], 1e-15f));
        float db = 0.f;
        for (int tq = 0; tq < T; tq++) {
            const int64_t rq = tq == 0 ? E + i : (int64_t)p0 + tq - 1;
            float qq[DS], dO[DS];
            gen_ld_slice<DS>(qq, QKV + rq * ld + h * HD + d0, nv, v4);
            gen_ld_slice<DS>(dO, dAO + rq * D + h * HD + d0, nv, v4);
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int j = 0; j < DS; j++) { s = fmaf(qq[j], k[j], s); dp = fmaf(dO[j], v[j], dp); }
            s = gen_slice_sum<SL, TPW>(s);
            dp = gen_slice_sum<SL, TPW>(dp);
            const float p = expf(s * scale + bias - LSE[rq * NH + h]);
            const float ds = p * (dp - DELTA[rq * NH + h]);
            db += ds;
#pragma unroll
            for (int j = 0; j < DS; j++) { dv[j] = fmaf(p, dO[j], dv[j]); dk[j] = fmaf(ds * scale, qq[j], dk[j]); }
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < DS; j++)
                if (j < nv) { dQKV[rk * ld + D + h * HD + d0 + j] = dk[j]; dQKV[rk * ld + 2 * D + h * HD + d0 + j] = dv[j]; }
            if (tk > 0 && d0 == 0) dbias_h[(int64_t)h * E + p0 + tk - 1] = db;
        }
    }
}
// dfc[p] += (sum_h dbias_h[h][p]) / fc[p]   (d log(max(fc, 1e-15)) / dfc; 0 below the clamp)
__global__ void k_gen_dfc(const float* __restrict__ dbias_h, const float* __restrict__ fc, float* __restrict__ dfc,
                          int64_t E, int NH) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    float s = 0.f;
    for (int h = 0; h < NH; h++) s += dbias_h[(int64_t)h * E + p];
    const float f = fc[p];
    dfc[p] += f > 1e-15f ? s / f : 0.f;
}

template <class F>
static void attn_dispatch(int HD, F f) {
    if (HD <= 4) f(std::integral_constant<int, 4>());
    else if (HD <= 16) f(std::integral_constant<int, 16>());
    else if (HD <= 32) f(std::integral_constant<int, 32>());
    else if (HD <= 64) f(std::integral_constant<int, 64>());
    else f(std::integral_constant<int, 128>());
}

// ---------------------------------------------------------------------------------------------
// heads (backend.py:651-777): pred[i][p] = node_pred[i][p] + sum_{e in row i} fc_e edge_pred[e][p]
// ---------------------------------------------------------------------------------------------
__global__ void k_gen_atom_sum(const float* __restrict__ npred, const float* __restrict__ epred, const float* __restrict__ fc,
                               const int* __restrict__ rowptr, float* __restrict__ out, int64_t N, int P) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * P) return;
    const int64_t i = idx / P;
    const int p = (int)(idx % P);
    float s = npred[idx];
    for (int e = rowptr[i]; e < rowptr[i + 1]; e++) s += fc[e] * epred[(int64_t)e * P + p];
    out[idx] = s;
}
// edge_sum[i][k] = sum_{e in row i} fc_e X[e][k]
__global__ void k_gen_edge_sum(const float* __restrict__ X, const float* __restrict__ fc, const int* __restrict__ rowptr,
                               float* __restrict__ out, int64_t ldo, int64_t N, int W) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * W) return;
    const int64_t i = idx / W;
    const int k = (int)(idx % W);
    float s = 0.f;
    for (int e = rowptr[i]; e < rowptr[i + 1]; e++) s += fc[e] * X[(int64_t)e * W + k];
    out[i * ldo + k] = s;
}
// seeds of the edge head: dEpred[e][p] = fc_e gA[ctr e][p];  dfc[e] (+)= sum_p gA[ctr e][p] epred[e][p]
__global__ void k_gen_edge_seed(const float* __restrict__ gA, const int* __restrict__ ctr, const float* __restrict__ fc,
                                const float* __restrict__ epred, float* __restrict__ dEp, float* __restrict__ dfc, int acc,
                                int64_t E, int P) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t i = ctr[e];
    float s = 0.f;
    for (int p = 0; p < P; p++) {
        const float gv = gA[i * P + p];
        dEp[e * P + p] = fc[e] * gv;
        s = fmaf(gv, epred[e * P + p], s);
    }
    if (acc) dfc[e] += s;
    else dfc[e] = s;
}

// ---------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------
struct GAttn {
    float *X, *QKV, *AO, *LSE, *X1, *VG, *T1, *S2, *H, *H1, *VGn, *Hn, *TOKo;
};
struct GGnn {
    std::vector<GAttn> attn;
    float *a0, *XF, *CA, *Mout, *Hin, *Hout;
};
struct GWs {
    std::vector<GGnn> gnn;
    float *H0, *M0, *cond;
    // temporaries
    float *tE1, *tE2, *tE3, *tE4;   // [R][wmax]
    float *tN1, *tN2, *tN3;   // [N][nmax]
    float *dX, *dX2, *dM, *dH, *dQKV, *DELTA, *dbias_h, *dgeo, *dfc, *dv;
    size_t bytes;
};
static int imax(int a, int b) { return a > b ? a : b; }
static void gen_carve(const Model& m, int64_t N, int64_t E, void* base, GWs& w) {
    const GD d = dims_of(m);
    Carver c(base);
    const int64_t R = E + N, Ra = R > 0 ? R : 1, Na = N > 0 ? N : 1, Ea = E > 0 ? E : 1;
    const bool post = m.post_ln();
    w.gnn.resize(m.h.num_gnn_layers);
    w.H0 = c.take<float>(Na * d.DN);
    w.M0 = c.take<float>(Ea * d.D);
    w.cond = m.h.system_conditioning ? c.take<float>(Na * d.DN) : nullptr;
    float* prev = w.H0;
    for (size_t gi = 0; gi < w.gnn.size(); gi++) {
        GGnn& G = w.gnn[gi];
        G.attn.resize(m.h.num_attention_layers);
        G.a0 = c.take<float>(Ea * d.D);
        G.XF = c.take<float>(Ea * d.D);
        G.CA = c.take<float>(Ea * 2 * d.D);
        G.Mout = c.take<float>(Ea * d.D);
        if (m.residual() && gi > 0) prev = c.take<float>(Na * d.DN);
        G.Hin = prev;
        for (auto& A : G.attn) {
            A.X = c.take<float>(Ra * d.D);
            A.QKV = c.take<float>(Ra * 3 * d.D);
            A.AO = c.take<float>(Ra * d.D);
            A.LSE = c.take<float>(Ra * d.NH);
            A.X1 = c.take<float>(Ra * d.D);
            A.VG = c.take<float>(Ra * 2 * d.DFF);
            A.T1 = post ? c.take<float>(Ra * d.D) : nullptr;
            A.S2 = post ? c.take<float>(Ra * d.D) : nullptr;
            A.TOKo = c.take<float>(Na * d.D);   // the centre token leaving the layer (attention output / PostLN norm_mlp row)
            A.H = prev;
            A.H1 = d.expanded ? c.take<float>(Na * d.DN) : nullptr;
            A.VGn = d.expanded ? c.take<float>(Na * 2 * d.DNF) : nullptr;
            A.Hn = c.take<float>(Na * d.DN);
            prev = A.Hn;
        }
        G.Hout = prev;
    }
    const int wmax = imax(imax(3 * d.D, 2 * d.DFF), imax(2 * d.D, d.DH));
    const int nmax = imax(imax(2 * d.DNF, d.DN), imax(d.DH, d.D));
    w.tE1 = c.take<float>(Ra * wmax); w.tE2 = c.take<float>(Ra * wmax); w.tE3 = c.take<float>(Ra * wmax);
    w.tE4 = c.take<float>(Ra * wmax);
    w.tN1 = c.take<float>(Na * nmax); w.tN2 = c.take<float>(Na * nmax); w.tN3 = c.take<float>(Na * nmax);
    w.dX = c.take<float>(Ra * d.D); w.dX2 = c.take<float>(Ra * d.D);
    w.dM = c.take<float>(Ea * d.D); w.dH = c.take<float>(Na * d.DN);
    w.dQKV = c.take<float>(Ra * 3 * d.D);
    w.DELTA = c.take<float>(Ra * d.NH);
    w.dbias_h = c.take<float>(Ea * d.NH);
    w.dgeo = c.take<float>(Ea * 4); w.dfc = c.take<float>(Ea); w.dv = c.take<float>(Ea * 4);
    w.bytes = c.off;
}

static inline int g1(int64_t n) { return (int)cdiv(n > 0 ? n : 1, 256); }

struct Ops {   // launch helpers of one pass
    const Model& m;
    const Graph& g;
    GD d;
    hipStream_t st;
    Lins lin;
    int64_t N, E, R;
    Ops(const Model& m_, const Graph& g_, hipStream_t s) : m(m_), g(g_), d(dims_of(m_)), st(s), lin{s}, N(g_.n_nodes),
        E(g_.n_edges), R(g_.n_nodes + g_.n_edges) {}
    float eps() const { return m.layer_norm() ? 1e-5f : 1.1920929e-07f; }
    void norm(const float* X, const float* gamma, const float* beta, float* Y, int64_t rows, int W) const {
        if (rows > 0) k_gen_norm<<<(int)cdiv(rows, 4), 256, 0, st>>>(X, gamma, m.layer_norm() ? beta : nullptr, m.layer_norm(), eps(), Y, rows, W);
    }
    void norm_bwd(const float* X, const float* gamma, const float* dY, float* dX, bool acc, int64_t rows, int W) const {
        if (rows > 0) k_gen_norm_bwd<<<(int)cdiv(rows, 4), 256, 0, st>>>(X, gamma, m.layer_norm(), eps(), dY, dX, acc, rows, W);
    }
    void axpby(float a, const float* A, int64_t lda, float b, const float* B, int64_t ldb, const int* index, float* Y,
               int64_t ldy, bool acc, int64_t rows, int W) const {
        if (rows > 0) k_gen_axpby<<<g1(rows * W), 256, 0, st>>>(a, A, lda, b, B, ldb, index, Y, ldy, acc, rows, W);
    }
    void copy(const float* A, float* Y, int64_t rows, int W) const { axpby(1.f, A, W, 0.f, nullptr, 0, nullptr, Y, W, false, rows, W); }
    void add(const float* A, float* Y, int64_t rows, int W) const { axpby(1.f, A, W, 0.f, nullptr, 0, nullptr, Y, W, true, rows, W); }
    // y = x + w_out(swiglu(w_in(norm(x))))  -> VG saved; out may alias nothing of the inputs
    // FFN block on `rows` rows of width W: N = (normed ? norm(X) : X); VG = w_in N; S = swiglu(VG); Y = base + w_out S
    void ffn(const float* Xin, bool normed, const float* gamma, const float* beta, const Lin& w_in, const Lin& w_out,
             float* VG, const float* base, float* Y, float* tA, float* tB, int64_t rows, int W, int F) const {
        const float* Nn = Xin;
        if (normed) { norm(Xin, gamma, beta, tA, rows, W); Nn = tA; }
        lin.fwd(Nn, W, w_in, VG, 2 * F, rows);
        if (rows > 0) k_gen_swiglu<<<g1(rows * F), 256, 0, st>>>(VG, tB, rows, F);
        if (base != Y) copy(base, Y, rows, W);
        lin.fwd(tB, F, w_out, Y, W, rows, true);   // += w_out S + bias
    }
    // adjoint of the FFN branch: dIn (+)= d/dXin [w_out(swiglu(w_in(norm(Xin))))] for dY; the residual path is the caller's
    void ffn_bwd(const float* Xin, bool normed, const float* gamma, const Lin& w_in, const Lin& w_out, const float* VG,
                 const float* dY, float* dIn, bool acc, float* tA, float* tB, int64_t rows, int W, int F) const {
        lin.bwd(dY, W, w_out, tA, F, rows);                                       // dS
        if (rows > 0) k_gen_swiglu_bwd<<<g1(rows * F), 256, 0, st>>>(VG, tA, tB, rows, F);  // dVG
        if (normed) {
            lin.bwd(tB, 2 * F, w_in, tA, W, rows);                                // dN
            norm_bwd(Xin, gamma, tA, dIn, acc, rows, W);
        } else
            lin.bwd(tB, 2 * F, w_in, dIn, W, rows, acc);
    }
};

}  // namespace

}  // namespace pet
