// Size-generic PET path: forward + hand-written reverse pass (dE/dR) for ANY (d_pet, d_node, d_feedforward, d_head,
// num_heads) -- the reference is size-generic (pet/documentation.py:196-213; its own architecture suites run at
// d_pet = 1, pet/tests/test_basic.py:22-32) while the tuned kernels of pet_fwd / pet_trr / pet_attn / pet_comb are ONE
// compiled instantiation (128 / 256 / 256 / 128 / 8). A model of any other size runs here, behind the same C ABI:
//
//   * every Linear is one fp32 FMA GEMM kernel over the RAW torch weights [n_out, k_in] (k_gen_lin: 64 x 64 output tiles,
//     K chunks of 16 through LDS, run-time bounds everywhere), the adjoint dX = dY W is the same kernel with swapped strides;
//   * norms / SwiGLU / SiLU are row kernels with a run-time width (one wave per row);
//   * attention is one wave per (atom, head): lanes own queries (forward, dQ) or keys (dK, dV, key-bias gradient) and walk
//     the other index with an online soft-max, so any head dimension up to 128 and ANY number of neighbours is served
//     (no 16-token tiles) -- no cross-lane reduction, fixed summation order, bit-reproducible;
//   * d_node == d_pet follows transformer.py:189-201: no centre contraction / expansion / centre MLP, the node features
//     leaving a layer ARE the centre token;
//   * all architecture switches of the tuned path: RMSNorm / LayerNorm, PreLN / PostLN, feedforward / residual featuriser,
//     SwiGLU / SiLU (tied halves), system conditioning, bump / cosine / adaptive cutoffs (graph side, shared).
// Correctness-first (the matrix cores are not used): a few percent of the tuned path's rate, documented in DESIGN.md.
// Training of generic sizes is not built (refused in abi.hip).
#include <type_traits>

#include "common.h"
#include "model.h"

namespace pet {

int attn_tiles(const Graph& g);
int backward_geometry_generic(const Model& m, const Graph& g, float* dv_scratch, const float* dgeo, const float* dfc_a,
                              const float* dfc_b, float* gpos, float* gcell, hipStream_t st);  // pet_bwd.hip

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
template <bool ACC>
__global__ __launch_bounds__(256) void k_gen_lin(const float* __restrict__ X, int64_t ldx, const float* __restrict__ W,
                                                 int64_t so, int64_t si, const float* __restrict__ b,
                                                 float* __restrict__ Y, int64_t ldy, int64_t R, int NO, int KI) {
    __shared__ float Xs[64][17], Ws[64][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int o0 = blockIdx.y * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < KI; k0 += 16) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
            const int rr = idx >> 4, kk = idx & 15;
            const int64_t r = r0 + rr;
            Xs[rr][kk] = (r < R && k0 + kk < KI) ? X[r * ldx + k0 + kk] : 0.f;
            const int o = o0 + rr;
            Ws[rr][kk] = (o < NO && k0 + kk < KI) ? W[(int64_t)o * so + (int64_t)(k0 + kk) * si] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            float a[4], c[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { a[i] = Xs[ty + 16 * i][kk]; c[i] = Ws[tx + 16 * i][kk]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], c[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int64_t r = r0 + ty + 16 * i;
        if (r >= R) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int o = o0 + tx + 16 * j;
            if (o >= NO) continue;
            const float v = acc[i][j] + (b ? b[o] : 0.f);
            if (ACC) Y[r * ldy + o] += v;
            else Y[r * ldy + o] = v;
        }
    }
}

struct Lins {
    hipStream_t st;
    // y = x W^T + b
    void fwd(const float* X, int64_t ldx, const Lin& L, float* Y, int64_t ldy, int64_t R, bool acc = false,
             int col0 = 0, int kin = -1) const {
        if (R <= 0) return;
        const int K = kin < 0 ? L.k_in : kin;  // a column block [col0, col0 + K) of the weight (compress.0)
        dim3 grid((unsigned)cdiv(R, 64), (unsigned)cdiv(L.n_out, 64));
        if (acc) k_gen_lin<true><<<grid, 256, 0, st>>>(X, ldx, L.w + col0, L.k_in, 1, L.b, Y, ldy, R, L.n_out, K);
        else k_gen_lin<false><<<grid, 256, 0, st>>>(X, ldx, L.w + col0, L.k_in, 1, L.b, Y, ldy, R, L.n_out, K);
    }
    // dx (+)= dy W
    void bwd(const float* dY, int64_t ldy, const Lin& L, float* dX, int64_t ldx, int64_t R, bool acc = false,
             int col0 = 0, int kin = -1) const {
        if (R <= 0) return;
        const int K = kin < 0 ? L.k_in : kin;
        dim3 grid((unsigned)cdiv(R, 64), (unsigned)cdiv(K, 64));
        if (acc) k_gen_lin<true><<<grid, 256, 0, st>>>(dY, ldy, L.w + col0, 1, L.k_in, nullptr, dX, ldx, R, K, L.n_out);
        else k_gen_lin<false><<<grid, 256, 0, st>>>(dY, ldy, L.w + col0, 1, L.k_in, nullptr, dX, ldx, R, K, L.n_out);
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
template <int HDM>
__global__ __launch_bounds__(64) void k_gen_attn_fwd(const float* __restrict__ QKV, const int* __restrict__ rowptr,
                                                     const float* __restrict__ fc, float* __restrict__ AO,
                                                     float* __restrict__ LSE, int64_t E, int D, int NH, int HD, float scale) {
    const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int p0 = rowptr[i], T = rowptr[i + 1] - p0 + 1;
    const int64_t ld = 3 * (int64_t)D;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int tq = t0 + lane;
        const bool live = tq < T;
        const int64_t rq = !live ? E + i : (tq == 0 ? E + i : (int64_t)p0 + tq - 1);
        float q[HDM], acc[HDM];
#pragma unroll
        for (int d = 0; d < HDM; d++) {
            q[d] = d < HD ? QKV[rq * ld + h * HD + d] * scale : 0.f;
            acc[d] = 0.f;
        }
        float mx = -INFINITY, l = 0.f;
        for (int tk = 0; tk < T; tk++) {
            const int64_t rk = tk == 0 ? E + i : (int64_t)p0 + tk - 1;
            const float* kp = QKV + rk * ld + D + h * HD;
            const float* vp = QKV + rk * ld + 2 * D + h * HD;
            float s = tk == 0 ? 0.f : logf(fmaxf(fc[p0 + tk - 1], 1e-15f));
#pragma unroll
            for (int d = 0; d < HDM; d++)
                if (d < HD) s = fmaf(q[d], kp[d], s);
            const float mn = fmaxf(mx, s), c = expf(mx - mn), p = expf(s - mn);
            l = l * c + p;
#pragma unroll
            for (int d = 0; d < HDM; d++)
                if (d < HD) acc[d] = acc[d] * c + p * vp[d];
            mx = mn;
        }
        if (live) {
            const float il = 1.0f / l;
#pragma unroll
            for (int d = 0; d < HDM; d++)
                if (d < HD) AO[rq * D + h * HD + d] = acc[d] * il;
            LSE[rq * NH + h] = mx + logf(l);
        }
    }
}

// pass A, lanes = queries: delta = <dO, O>, dQ
template <int HDM>
__global__ __launch_bounds__(64) void k_gen_attn_bwd_q(const float* __restrict__ QKV, const float* __restrict__ AO,
                                                       const float* __restrict__ dAO, const float* __restrict__ LSE,
                                                       const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                       float* __restrict__ dQKV, float* __restrict__ DELTA, int64_t E,
                                                       int D, int NH, int HD, float scale) {
    const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int p0 = rowptr[i], T = rowptr[i + 1] - p0 + 1;
    const int64_t ld = 3 * (int64_t)D;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int tq = t0 + lane;
        if (tq >= T) continue;
        const int64_t rq = tq == 0 ? E + i : (int64_t)p0 + tq - 1;
        float q[HDM], dO[HDM], dq[HDM];
        float delta = 0.f;
#pragma unroll
        for (int d = 0; d < HDM; d++) {
            q[d] = d < HD ? QKV[rq * ld + h * HD + d] * scale : 0.f;
            dO[d] = d < HD ? dAO[rq * D + h * HD + d] : 0.f;
            dq[d] = 0.f;
            if (d < HD) delta = fmaf(dO[d], AO[rq * D + h * HD + d], delta);
        }
        const float lse = LSE[rq * NH + h];
        for (int tk = 0; tk < T; tk++) {
            const int64_t rk = tk == 0 ? E + i : (int64_t)p0 + tk - 1;
            const float* kp = QKV + rk * ld + D + h * HD;
            const float* vp = QKV + rk * ld + 2 * D + h * HD;
            float s = tk == 0 ? 0.f : logf(fmaxf(fc[p0 + tk - 1], 1e-15f));
            float dp = 0.f;
#pragma unroll
            for (int d = 0; d < HDM; d++)
                if (d < HD) { s = fmaf(q[d], kp[d], s); dp = fmaf(dO[d], vp[d], dp); }
            const float ds = expf(s - lse) * (dp - delta);
#pragma unroll
            for (int d = 0; d < HDM; d++)
                if (d < HD) dq[d] = fmaf(ds, kp[d], dq[d]);
        }
#pragma unroll
        for (int d = 0; d < HDM; d++)
            if (d < HD) dQKV[rq * ld + h * HD + d] = dq[d] * scale;
        DELTA[rq * NH + h] = delta;
    }
}

// pass B, lanes = keys: dK, dV and the key-bias gradient (edge keys; head-major [NH][E])
template <int HDM>
__global__ __launch_bounds__(64) void k_gen_attn_bwd_k(const float* __restrict__ QKV, const float* __restrict__ dAO,
                                                       const float* __restrict__ LSE, const float* __restrict__ DELTA,
                                                       const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                       float* __restrict__ dQKV, float* __restrict__ dbias_h, int64_t E,
                                                       int D, int NH, int HD, float scale) {
    const int i = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int p0 = rowptr[i], T = rowptr[i + 1] - p0 + 1;
    const int64_t ld = 3 * (int64_t)D;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int tk = t0 + lane;
        if (tk >= T) continue;
        const int64_t rk = tk == 0 ? E + i : (int64_t)p0 + tk - 1;
        float k[HDM], v[HDM], dk[HDM], dv[HDM];
#pragma unroll
        for (int d = 0; d < HDM; d++) {
            k[d] = d < HD ? QKV[rk * ld + D + h * HD + d] : 0.f;
            v[d] = d < HD ? QKV[rk * ld + 2 * D + h * HD + d] : 0.f;
            dk[d] = 0.f; dv[d] = 0.f;
        }
        const float bias = tk == 0 ? 0.f : logf(fmaxf(fc[p0 + tk - 1], 1e-15f));
        float db = 0.f;
        for (int tq = 0; tq < T; tq++) {
            const int64_t rq = tq == 0 ? E + i : (int64_t)p0 + tq - 1;
            const float* qp = QKV + rq * ld + h * HD;
            const float* dop = dAO + rq * D + h * HD;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HDM; d++)
                if (d < HD) { s = fmaf(qp[d], k[d], s); dp = fmaf(dop[d], v[d], dp); }
            const float p = expf(s * scale + bias - LSE[rq * NH + h]);
            const float ds = p * (dp - DELTA[rq * NH + h]);
            db += ds;
#pragma unroll
            for (int d = 0; d < HDM; d++)
                if (d < HD) { dv[d] = fmaf(p, dop[d], dv[d]); dk[d] = fmaf(ds * scale, qp[d], dk[d]); }
        }
#pragma unroll
        for (int d = 0; d < HDM; d++)
            if (d < HD) { dQKV[rk * ld + D + h * HD + d] = dk[d]; dQKV[rk * ld + 2 * D + h * HD + d] = dv[d]; }
        if (tk > 0) dbias_h[(int64_t)h * E + p0 + tk - 1] = db;
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

int64_t gen_workspace_bytes(const Model& m, int64_t N, int64_t E) {
    GWs w;
    gen_carve(m, N, E, nullptr, w);
    return (int64_t)w.bytes;
}

// ---------------------------------------------------------------------------------------------
// predict (a function of the features it is given) and its adjoint
// ---------------------------------------------------------------------------------------------
static int gen_head_fwd(const Ops& o, const Lin& h0, const Lin& h2, const float* X, int W, int64_t rows, float* a1, float* s1,
                        float* a2, float* s2) {
    o.lin.fwd(X, W, h0, a1, o.d.DH, rows);
    if (rows > 0) k_gen_silu<<<g1(rows * o.d.DH), 256, 0, o.st>>>(a1, s1, rows * o.d.DH);
    o.lin.fwd(s1, o.d.DH, h2, a2, o.d.DH, rows);
    if (rows > 0) k_gen_silu<<<g1(rows * o.d.DH), 256, 0, o.st>>>(a2, s2, rows * o.d.DH);
    return PET_OK;
}

int gen_predict(const Model& m, const Graph& g, const HeadW& H, const LastW& Lw, const float* node_feat,
                const float* edge_feat, const float* fc, float* atomic, float* node_hidden, float* edge_hidden,
                hipStream_t st) {
    Ops o(m, g, st);
    const int64_t N = o.N, E = o.E;
    if (N == 0) return PET_OK;
    const int DH = o.d.DH, P = Lw.P;
    if (!fc) fc = g.fc;
    // (the caller's scratch is sized for the compiled instantiation: this path takes its own from the stream's pool)
    const int64_t need = 4 * (N + E + 2) * (int64_t)DH + (N + E + 2) * (int64_t)P;
    float* scratch = nullptr;
    PET_HIP_CHECK(hipMallocAsync((void**)&scratch, (size_t)need * sizeof(float), st));
    float* a1n = scratch; float* s1n = a1n + N * DH; float* a2n = s1n + N * DH; float* s2n = a2n + N * DH;
    float* a1e = s2n + N * DH; float* s1e = a1e + E * DH; float* a2e = s1e + E * DH; float* s2e = a2e + E * DH;
    float* np = s2e + E * DH; float* ep = np + N * P;
    gen_head_fwd(o, H.nh0, H.nh2, node_feat, o.d.DN, N, a1n, s1n, a2n, s2n);
    Lin ln; ln.w = Lw.nw; ln.b = Lw.nb; ln.n_out = P; ln.k_in = DH;
    o.lin.fwd(s2n, DH, ln, np, P, N);
    if (E > 0) {
        gen_head_fwd(o, H.eh0, H.eh2, edge_feat, o.d.D, E, a1e, s1e, a2e, s2e);
        Lin le; le.w = Lw.ew; le.b = Lw.eb; le.n_out = P; le.k_in = DH;
        o.lin.fwd(s2e, DH, le, ep, P, E);
    }
    k_gen_atom_sum<<<g1(N * P), 256, 0, st>>>(np, ep, fc, g.rowptr, atomic, N, P);
    if (node_hidden) o.copy(s2n, node_hidden, N, DH);
    if (edge_hidden && E > 0) o.copy(s2e, edge_hidden, E, DH);
    PET_HIP_CHECK(hipFreeAsync(scratch, st));
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

int gen_predict_backward(const Model& m, const Graph& g, const HeadW& H, const LastW& Lw, const float* node_feat,
                         const float* edge_feat, const float* fc, const float* gA, float* g_node, float* g_edge, float* g_fc,
                         hipStream_t st) {
    Ops o(m, g, st);
    const int64_t N = o.N, E = o.E;
    if (N == 0) return PET_OK;
    const int DH = o.d.DH, P = Lw.P;
    if (!fc) fc = g.fc;
    const int64_t M = N > E ? N : E;
    float* scratch = nullptr;
    PET_HIP_CHECK(hipMallocAsync((void**)&scratch, (size_t)(4 * M * DH + 2 * M * P) * sizeof(float), st));
    float* a1 = scratch; float* s1 = a1 + M * DH; float* a2 = s1 + M * DH; float* s2 = a2 + M * DH;
    float* pr = s2 + M * DH; float* dpr = pr + M * P;
    // node branch (recomputed from the features: nothing is read from a forward workspace)
    gen_head_fwd(o, H.nh0, H.nh2, node_feat, o.d.DN, N, a1, s1, a2, s2);
    Lin ln; ln.w = Lw.nw; ln.b = Lw.nb; ln.n_out = P; ln.k_in = DH;
    o.lin.bwd(gA, P, ln, s2, DH, N);                                                   // d s2
    k_gen_silu_bwd<<<g1(N * DH), 256, 0, st>>>(a2, s2, s2, N * DH);                    // d a2
    o.lin.bwd(s2, DH, H.nh2, s1, DH, N);                                               // d s1
    k_gen_silu_bwd<<<g1(N * DH), 256, 0, st>>>(a1, s1, s1, N * DH);                    // d a1
    o.lin.bwd(s1, DH, H.nh0, g_node, o.d.DN, N);
    if (E > 0) {
        gen_head_fwd(o, H.eh0, H.eh2, edge_feat, o.d.D, E, a1, s1, a2, s2);
        Lin le; le.w = Lw.ew; le.b = Lw.eb; le.n_out = P; le.k_in = DH;
        o.lin.fwd(s2, DH, le, pr, P, E);
        k_gen_edge_seed<<<g1(E), 256, 0, st>>>(gA, g.ctr, fc, pr, dpr, g_fc, 0, E, P);
        o.lin.bwd(dpr, P, le, s2, DH, E);
        k_gen_silu_bwd<<<g1(E * DH), 256, 0, st>>>(a2, s2, s2, E * DH);
        o.lin.bwd(s2, DH, H.eh2, s1, DH, E);
        k_gen_silu_bwd<<<g1(E * DH), 256, 0, st>>>(a1, s1, s1, E * DH);
        o.lin.bwd(s1, DH, H.eh0, g_edge, o.d.D, E);
    }
    PET_HIP_CHECK(hipFreeAsync(scratch, st));
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

int gen_aux_outputs(const Model& m, const Graph& g, const float* node_feat, const float* edge_feat, float* feature,
                    float* last_layer, float* scratch, hipStream_t st) {
    Ops o(m, g, st);
    const int64_t N = o.N, E = o.E;
    const GD& d = o.d;
    if (N == 0) return PET_OK;
    if (feature) {
        o.axpby(1.f, node_feat, d.DN, 0.f, nullptr, 0, nullptr, feature, d.DN + d.D, false, N, d.DN);
        k_gen_edge_sum<<<g1(N * d.D), 256, 0, st>>>(edge_feat, g.fc, g.rowptr, feature + d.DN, d.DN + d.D, N, d.D);
    }
    if (last_layer) {
        PET_REQUIRE(m.has_fused_head, PET_ERR_ARGUMENT, "last-layer features need the fused target's heads");
        const int64_t M = N > E ? N : E;
        (void)scratch;  // sized for the compiled instantiation: this path takes its temporaries from the stream's pool
        float* tmp = nullptr;
        PET_HIP_CHECK(hipMallocAsync((void**)&tmp, (size_t)4 * M * d.DH * sizeof(float), st));
        float* b1 = tmp; float* b2 = b1 + M * d.DH; float* b3 = b2 + M * d.DH; float* b4 = b3 + M * d.DH;
        gen_head_fwd(o, m.nh0, m.nh2, node_feat, d.DN, N, b1, b2, b3, b4);
        o.axpby(1.f, b4, d.DH, 0.f, nullptr, 0, nullptr, last_layer, 2 * d.DH, false, N, d.DH);
        if (E > 0) {
            gen_head_fwd(o, m.eh0, m.eh2, edge_feat, d.D, E, b1, b2, b3, b4);
            k_gen_edge_sum<<<g1(N * d.DH), 256, 0, st>>>(b4, g.fc, g.rowptr, last_layer + d.DH, 2 * d.DH, N, d.DH);
        } else
            o.axpby(0.f, b4, d.DH, 0.f, nullptr, 0, nullptr, last_layer + d.DH, 2 * d.DH, false, N, d.DH);
        PET_HIP_CHECK(hipFreeAsync(tmp, st));
    }
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// ---------------------------------------------------------------------------------------------
// forward (backend.py:496-649)
// ---------------------------------------------------------------------------------------------
int gen_forward_layers(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, int save, float* atomic,
                       float* const* node_feats, float* const* edge_feats, int n_layers, hipStream_t st) {
    GWs w;
    gen_carve(m, g.n_nodes, g.n_edges, ws, w);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT,
                "forward workspace too small for the size-generic path (other model sizes, or an atom with more than 127 "
                "neighbours): size it with pet_forward_workspace_bytes_for(model, graph)");
    PET_REQUIRE(save != 2, PET_ERR_UNSUPPORTED,
                "training is built for the compiled model size (d_pet=128, d_node=256, d_feedforward=256, d_head=128, "
                "num_heads=8) and at most 127 neighbours per atom");
    const bool post = m.post_ln(), res = m.residual();
    PET_REQUIRE(res ? (n_layers == m.h.num_gnn_layers || (n_layers == 1 && !node_feats[0] && !edge_feats[0])) : n_layers == 1,
                PET_ERR_ARGUMENT, "expected one feature pair per readout layer");
    PET_REQUIRE(!(atomic && res), PET_ERR_UNSUPPORTED,
                "the fused head reads one readout layer; with the residual featuriser use pet_forward_layers and pet_predict");
    Ops o(m, g, st);
    const GD& d = o.d;
    const int64_t N = o.N, E = o.E, R = o.R;
    if (N == 0) return PET_OK;
    const int D = d.D, DN = d.DN;
    const float scale = 1.0f / (sqrtf((float)d.HD) * m.h.attention_temperature);
    const int L = m.h.num_gnn_layers, AL = m.h.num_attention_layers;
    k_gen_embed<<<g1(N * DN), 256, 0, st>>>(g.sp, m.node_emb, w.H0, DN, N, DN);
    if (E > 0) k_gen_embed<<<g1(E * D), 256, 0, st>>>(g.sp_nbr, m.edge_emb, w.M0, D, E, D);
    if (m.h.system_conditioning) {
        PET_REQUIRE(g.cond_charge && g.n_cond_systems >= 1 && g.n_cond_systems <= N, PET_ERR_ARGUMENT,
                    "system_conditioning: call pet_graph_set_conditioning (charge, spin multiplicity, system indices) first");
        k_gen_system_cond<<<(int)g.n_cond_systems, 128, 3 * DN * sizeof(float), st>>>(
            g.cond_charge, g.cond_spin, m.cond_qe, m.cond_se, m.cond_w0, m.cond_b0, m.cond_w2, m.cond_b2, w.cond,
            m.h.max_charge, DN);
    }
    const float* Min = w.M0;
    for (int gi = 0; gi < L; gi++) {
        const GnnLayerW& G = m.gnn[gi];
        GGnn& B = w.gnn[gi];
        if (res && gi > 0) k_gen_embed<<<g1(N * DN), 256, 0, st>>>(g.sp, m.node_embs[gi], B.Hin, DN, N, DN);
        if (E > 0) {
            // tokens = [edge_embedder([v, d]) ; (gi > 0: neighbor_embedder[species]) ; message] -> compress (transformer.py:499-521)
            const int kin = (gi == 0 ? 2 : 3) * D;
            float* TOK = w.tE1;
            o.lin.fwd(reinterpret_cast<const float*>(g.geo), 4, G.eemb, TOK, kin, E);
            if (gi > 0) k_gen_embed<<<g1(E * D), 256, 0, st>>>(g.sp_nbr, G.nbr_emb, TOK + D, kin, E, D);
            o.axpby(1.f, Min, D, 0.f, nullptr, 0, nullptr, TOK + (gi == 0 ? D : 2 * D), kin, false, E, D);
            o.lin.fwd(TOK, kin, G.c0, B.a0, D, E);
            k_gen_silu<<<g1(E * D), 256, 0, st>>>(B.a0, w.tE2, E * D);
            o.lin.fwd(w.tE2, D, G.compress2, B.attn[0].X, D, E);
        }
        for (int a = 0; a < AL; a++) {
            const AttnLayerW& A = G.attn[a];
            GAttn& Ab = B.attn[a];
            float* Xnext = (a + 1 < AL) ? B.attn[a + 1].X : B.XF;
            // centre token (transformer.py:210-214)
            if (d.expanded) o.lin.fwd(Ab.H, DN, A.cc, Ab.X + E * D, D, N);
            else o.copy(Ab.H, Ab.X + E * D, N, D);
            const float* Xatt = Ab.X;
            if (!post) { o.norm(Ab.X, A.g_attn, A.b_attn, w.tE1, R, D); Xatt = w.tE1; }
            o.lin.fwd(Xatt, D, A.qkv, Ab.QKV, 3 * D, R);
            attn_dispatch(d.HD, [&](auto hdm) {
                k_gen_attn_fwd<decltype(hdm)::value><<<dim3((unsigned)N, (unsigned)d.NH), 64, 0, st>>>(
                    Ab.QKV, g.rowptr, g.fc, Ab.AO, Ab.LSE, E, D, d.NH, d.HD, scale);
            });
            float* OUT = w.tE2;  // output_linear of every token
            o.lin.fwd(Ab.AO, D, A.out, OUT, D, R);
            if (!post) {
                o.copy(OUT + E * D, Ab.TOKo, N, D);
                // edges: residual + MLP (transformer.py:229-232)
                if (E > 0) {
                    o.axpby(1.f, Ab.X, D, 1.f, OUT, D, nullptr, Ab.X1, D, false, E, D);
                    o.ffn(Ab.X1, true, A.g_mlp, A.b_mlp, A.mlp_in, A.mlp_out, Ab.VG, Ab.X1, Xnext, w.tE1, w.tE3, E, D, d.DFF);
                }
            } else {
                // transformer.py:245-247 on every token: S1 = tokens + attention; T1 = norm(S1); S2 = T1 + MLP(T1); T2 = norm(S2)
                o.axpby(1.f, Ab.X, D, 1.f, OUT, D, nullptr, Ab.X1, D, false, R, D);
                o.norm(Ab.X1, A.g_attn, A.b_attn, Ab.T1, R, D);
                o.ffn(Ab.T1, false, nullptr, nullptr, A.mlp_in, A.mlp_out, Ab.VG, Ab.T1, Ab.S2, w.tE1, w.tE3, R, D, d.DFF);
                o.norm(Ab.S2, A.g_mlp, A.b_mlp, w.tE1, R, D);
                if (E > 0) o.copy(w.tE1, Xnext, E, D);
                o.copy(w.tE1 + E * D, Ab.TOKo, N, D);
            }
            // node update (transformer.py:221-227)
            if (d.expanded) {
                o.copy(Ab.H, Ab.H1, N, DN);
                o.lin.fwd(Ab.TOKo, D, A.ce, Ab.H1, DN, N, true);
                o.ffn(Ab.H1, true, A.g_center, A.b_center, A.cmlp_in, A.cmlp_out, Ab.VGn, Ab.H1, Ab.Hn, w.tN1, w.tN2, N, DN, d.DNF);
            } else
                o.copy(Ab.TOKo, Ab.Hn, N, DN);
            if (a + 1 == AL && m.h.system_conditioning)
                k_gen_add_cond<<<g1(N * DN), 256, 0, st>>>(Ab.Hn, w.cond, g.sys, g.cond_sys, N, DN);
        }
        if (E > 0 && res) {
            if (gi + 1 < L) o.axpby(0.5f, Min, D, 0.5f, B.XF, D, g.rev, B.Mout, D, false, E, D);   // backend.py:640-647
        } else if (E > 0) {
            // backend.py:559-575: m = m + e + MLP(LayerNorm([e ; e[rev]]))
            float* CAT = w.tE1;
            o.axpby(1.f, B.XF, D, 0.f, nullptr, 0, nullptr, CAT, 2 * D, false, E, D);
            o.axpby(0.f, nullptr, 0, 1.f, B.XF, D, g.rev, CAT + D, 2 * D, false, E, D);
            k_gen_norm<<<(int)cdiv(E, 4), 256, 0, st>>>(CAT, G.ln_g, G.ln_b, 1, 1e-5f, w.tE2, E, 2 * D);
            o.lin.fwd(w.tE2, 2 * D, G.comb0, B.CA, 2 * D, E);
            k_gen_silu<<<g1(E * 2 * D), 256, 0, st>>>(B.CA, w.tE3, E * 2 * D);
            o.axpby(1.f, Min, D, 1.f, B.XF, D, nullptr, B.Mout, D, false, E, D);
            o.lin.fwd(w.tE3, 2 * D, G.comb2, B.Mout, D, E, true);
        }
        Min = B.Mout;
    }
    const GGnn& last = w.gnn.back();
    if (atomic) {
        PET_REQUIRE(m.has_fused_head, PET_ERR_ARGUMENT,
                    "pet_forward with d_atomic needs the fused single-property target; use pet_predict for other heads");
        HeadW H; H.nh0 = m.nh0; H.nh2 = m.nh2; H.eh0 = m.eh0; H.eh2 = m.eh2;
        const LastW& Lw = m.lasts.at("@|0|@");
        int rc = gen_predict(m, g, H, Lw, last.Hout, last.Mout, g.fc, atomic, nullptr, nullptr, st);
        if (rc) return rc;
    }
    for (int l = 0; l < n_layers; l++) {
        const GGnn& Bl = res ? w.gnn[l] : last;
        if (node_feats[l])
            PET_HIP_CHECK(hipMemcpyAsync(node_feats[l], Bl.Hout, N * DN * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (edge_feats[l] && E > 0)
            PET_HIP_CHECK(hipMemcpyAsync(edge_feats[l], res ? Bl.XF : Bl.Mout, E * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// ---------------------------------------------------------------------------------------------
// reverse pass of the features: (dL/d node features, dL/d edge features) of every readout layer -> dL/d geometry [E,4],
// dL/d cutoff factors [E] (the attention key biases)
// ---------------------------------------------------------------------------------------------
int gen_backward_features(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* const* g_node,
                          const float* const* g_edge, int n_layers, float* g_geo, float* g_fc, hipStream_t st) {
    GWs w;
    gen_carve(m, g.n_nodes, g.n_edges, ws, w);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "workspace too small");
    PET_REQUIRE(n_layers == m.num_readout_layers(), PET_ERR_ARGUMENT, "expected one gradient pair per readout layer");
    Ops o(m, g, st);
    const GD& d = o.d;
    const int64_t N = o.N, E = o.E, R = o.R;
    if (N == 0) return PET_OK;
    const int D = d.D, DN = d.DN;
    const bool post = m.post_ln(), res = m.residual();
    const float scale = 1.0f / (sqrtf((float)d.HD) * m.h.attention_temperature);
    const int L = m.h.num_gnn_layers, AL = m.h.num_attention_layers;
    if (E > 0) {
        PET_HIP_CHECK(hipMemsetAsync(g_geo, 0, E * 4 * sizeof(float), st));
        PET_HIP_CHECK(hipMemsetAsync(g_fc, 0, E * sizeof(float), st));
    }
    auto seed = [&](const float* src, float* dst, int64_t rows, int W) -> int {
        if (rows <= 0) return PET_OK;
        if (src) o.copy(src, dst, rows, W);
        else PET_HIP_CHECK(hipMemsetAsync(dst, 0, rows * W * sizeof(float), st));
        return PET_OK;
    };
    int rc;
    // dH: adjoint of the node features entering the next stage; dM: adjoint of the messages leaving layer gi
    if (!res) {
        if ((rc = seed(g_node[0], w.dH, N, DN))) return rc;
        if ((rc = seed(g_edge[0], w.dM, E, D))) return rc;
    } else {
        if ((rc = seed(nullptr, w.dM, E, D))) return rc;   // the last layer's messages are never read
    }
    for (int gi = L - 1; gi >= 0; gi--) {
        const GnnLayerW& G = m.gnn[gi];
        GGnn& B = w.gnn[gi];
        const float* Min = gi == 0 ? w.M0 : w.gnn[gi - 1].Mout;
        (void)Min;
        float* dXF = w.dX;   // adjoint of the edge tokens leaving the transformer, [E][D] (rows E.. are scratch)
        float* dMin = w.dX2; // adjoint of the incoming messages
        if (res) {
            // readout of this layer + (gi + 1 < L) the averaged messages: Mout = 0.5 (Min + XF[rev])
            if ((rc = seed(g_node[gi], w.dH, N, DN))) return rc;
            if ((rc = seed(g_edge[gi], dXF, E, D))) return rc;
            if (E > 0) {
                if (gi + 1 < L) {
                    o.axpby(0.f, nullptr, 0, 0.5f, w.dM, D, g.rev, dXF, D, true, E, D);  // rev is an involution
                    o.axpby(0.5f, w.dM, D, 0.f, nullptr, 0, nullptr, dMin, D, false, E, D);
                } else
                    PET_HIP_CHECK(hipMemsetAsync(dMin, 0, E * D * sizeof(float), st));
            }
        } else if (E > 0) {
            // Mout = Min + XF + comb2(silu(comb0(LN([XF ; XF[rev]]))))
            float* dS = w.tE1;                        // [E][2D]
            o.lin.bwd(w.dM, D, G.comb2, dS, 2 * D, E);
            k_gen_silu_bwd<<<g1(E * 2 * D), 256, 0, st>>>(B.CA, dS, dS, E * 2 * D);
            float* dCN = w.tE2;
            o.lin.bwd(dS, 2 * D, G.comb0, dCN, 2 * D, E);
            float* CAT = w.tE3;
            o.axpby(1.f, B.XF, D, 0.f, nullptr, 0, nullptr, CAT, 2 * D, false, E, D);
            o.axpby(0.f, nullptr, 0, 1.f, B.XF, D, g.rev, CAT + D, 2 * D, false, E, D);
            float* dCAT = w.tE1;
            k_gen_norm_bwd<<<(int)cdiv(E, 4), 256, 0, st>>>(CAT, G.ln_g, 1, 1e-5f, dCN, dCAT, 0, E, 2 * D);
            // dXF = dM + dCAT[:, :D] + dCAT[rev][:, D:]
            o.axpby(1.f, w.dM, D, 0.f, nullptr, 0, nullptr, dXF, D, false, E, D);
            o.axpby(1.f, dCAT, 2 * D, 1.f, dCAT + D, 2 * D, g.rev, dXF, D, true, E, D);
            o.copy(w.dM, dMin, E, D);
        }
        // transformer layers, last to first. dTok: adjoint of the tokens LEAVING layer a = [dXF ; centre part via dH]
        for (int a = AL - 1; a >= 0; a--) {
            const AttnLayerW& A = G.attn[a];
            GAttn& Ab = B.attn[a];
            // ---- node update adjoint: dH (of Hn) -> dTOKo [N][D] (tN3) and dH (of H entering the layer)
            float* dTOKo = w.tN3;
            if (d.expanded) {
                // Hn = H1 + cmlp(norm(H1)); H1 = H + ce(TOKo)
                float* dH1 = w.tN1;
                o.copy(w.dH, dH1, N, DN);
                {   // ffn_bwd needs two temporaries of width max(2 DNF, DN): tN2 and tN3 (dTOKo is produced after)
                    o.ffn_bwd(Ab.H1, true, A.g_center, A.cmlp_in, A.cmlp_out, Ab.VGn, w.dH, dH1, true, w.tN2, w.tN3, N, DN, d.DNF);
                }
                o.lin.bwd(dH1, DN, A.ce, dTOKo, D, N);
                o.copy(dH1, w.dH, N, DN);             // through the residual H1 = H + ...
            } else {
                o.copy(w.dH, dTOKo, N, D);
                PET_HIP_CHECK(hipMemsetAsync(w.dH, 0, N * DN * sizeof(float), st));
            }
            // ---- token adjoint entering output_linear: dOUT [R][D] in tE2, and dX (adjoint of the tokens ENTERING the layer)
            float* dOUT = w.tE2;
            float* dXin = w.tE3;   // [R][D]
            if (!post) {
                // edges: X2 = X1 + mlp(norm(X1)); X1 = X + OUT_e
                if (E > 0) {
                    float* dX1 = dXin;   // reuse: rows 0..E
                    o.copy(dXF, dX1, E, D);
                    o.ffn_bwd(Ab.X1, true, A.g_mlp, A.mlp_in, A.mlp_out, Ab.VG, dXF, dX1, true, w.tE1, w.tE2, E, D, d.DFF);
                    o.copy(dX1, dOUT, E, D);          // dOUT_e = dX1 ; dX_e (residual) = dX1 (already in dXin rows 0..E)
                }
                o.copy(dTOKo, dOUT + E * D, N, D);
                PET_HIP_CHECK(hipMemsetAsync(dXin + E * D, 0, N * D * sizeof(float), st));  // centre token has no residual
            } else {
                // T2 = norm_mlp(S2) [edges -> next tokens, centre -> TOKo]; S2 = T1 + mlp(T1); T1 = norm_attn(S1); S1 = X + OUT
                float* dT2 = w.tE1;
                if (E > 0) o.copy(dXF, dT2, E, D);
                o.copy(dTOKo, dT2 + E * D, N, D);
                float* dS2 = w.tE2;
                o.norm_bwd(Ab.S2, A.g_mlp, dT2, dS2, false, R, D);
                float* dT1 = dXin;
                o.copy(dS2, dT1, R, D);
                o.ffn_bwd(Ab.T1, false, nullptr, A.mlp_in, A.mlp_out, Ab.VG, dS2, dT1, true, w.tE1, w.tE4, R, D, d.DFF);
                float* dS1 = w.tE1;
                o.norm_bwd(Ab.X1, A.g_attn, dT1, dS1, false, R, D);
                o.copy(dS1, dOUT, R, D);
                o.copy(dS1, dXin, R, D);
            }
            // ---- output_linear, attention, input_linear
            float* dAO = w.tE1;
            o.lin.bwd(dOUT, D, A.out, dAO, D, R);
            attn_dispatch(d.HD, [&](auto hdm) {
                constexpr int HDM = decltype(hdm)::value;
                k_gen_attn_bwd_q<HDM><<<dim3((unsigned)N, (unsigned)d.NH), 64, 0, st>>>(
                    Ab.QKV, Ab.AO, dAO, Ab.LSE, g.rowptr, g.fc, w.dQKV, w.DELTA, E, D, d.NH, d.HD, scale);
                k_gen_attn_bwd_k<HDM><<<dim3((unsigned)N, (unsigned)d.NH), 64, 0, st>>>(
                    Ab.QKV, dAO, Ab.LSE, w.DELTA, g.rowptr, g.fc, w.dQKV, w.dbias_h, E, D, d.NH, d.HD, scale);
            });
            if (E > 0) k_gen_dfc<<<g1(E), 256, 0, st>>>(w.dbias_h, g.fc, g_fc, E, d.NH);
            if (!post) {
                float* dXN = w.tE2;
                o.lin.bwd(w.dQKV, 3 * D, A.qkv, dXN, D, R);
                o.norm_bwd(Ab.X, A.g_attn, dXN, dXin, true, R, D);
            } else
                o.lin.bwd(w.dQKV, 3 * D, A.qkv, dXin, D, R, true);
            // ---- split the token adjoint: edges -> dXF of the previous layer, centre -> dH through center_contraction
            if (E > 0) o.copy(dXin, dXF, E, D);
            if (d.expanded) o.lin.bwd(dXin + E * D, D, A.cc, w.dH, DN, N, true);
            else o.add(dXin + E * D, w.dH, N, DN);
        }
        // ---- compress adjoint: X0 = c2(silu(a0)), a0 = c0 [EE ; (nbr emb) ; Min]
        if (E > 0) {
            const int kin = (gi == 0 ? 2 : 3) * D;
            float* dS = w.tE1;
            o.lin.bwd(dXF, D, G.compress2, dS, D, E);
            k_gen_silu_bwd<<<g1(E * D), 256, 0, st>>>(B.a0, dS, dS, E * D);
            float* dTOK = w.tE2;
            o.lin.bwd(dS, D, G.c0, dTOK, kin, E);
            o.lin.bwd(dTOK, kin, G.eemb, g_geo, 4, E, true);                       // through edge_embedder([v, d])
            o.axpby(1.f, dTOK + (gi == 0 ? D : 2 * D), kin, 0.f, nullptr, 0, nullptr, dMin, D, true, E, D);
            o.copy(dMin, w.dM, E, D);                                               // adjoint of the previous layer's messages
        }
        if (res) PET_HIP_CHECK(hipMemsetAsync(w.dH, 0, N * DN * sizeof(float), st));  // each layer starts from an embedding
    }
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// staged pieces the C ABI exposes (features and their adjoints live in the workspace between the calls)
int gen_backward(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* gA, float* gpos, float* gcell,
                 hipStream_t st) {
    GWs w;
    gen_carve(m, g.n_nodes, g.n_edges, ws, w);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "workspace too small");
    PET_REQUIRE(!m.residual(), PET_ERR_UNSUPPORTED, "residual featuriser: use the staged calls (one head per readout layer)");
    PET_REQUIRE(m.has_fused_head, PET_ERR_ARGUMENT, "pet_backward needs the fused single-property target");
    const int64_t N = g.n_nodes, E = g.n_edges;
    if (N == 0) return PET_OK;
    const GD d = dims_of(m);
    HeadW H; H.nh0 = m.nh0; H.nh2 = m.nh2; H.eh0 = m.eh0; H.eh2 = m.eh2;
    const LastW& Lw = m.lasts.at("@|0|@");
    const GGnn& last = w.gnn.back();
    float *gn = nullptr, *ge = nullptr, *gfh = nullptr, *ggeo = nullptr, *gfc = nullptr;
    const int64_t Ea = E > 0 ? E : 1;
    PET_HIP_CHECK(hipMallocAsync((void**)&gn, (size_t)(N * d.DN + Ea * d.D + Ea + Ea * 4 + Ea) * sizeof(float), st));
    ge = gn + N * d.DN; gfh = ge + Ea * d.D; ggeo = gfh + Ea; gfc = ggeo + Ea * 4;
    int rc = gen_predict_backward(m, g, H, Lw, last.Hout, last.Mout, g.fc, gA, gn, ge, gfh, st);
    const float* gnp[1] = {gn};
    const float* gep[1] = {ge};
    if (!rc) rc = gen_backward_features(m, g, ws, ws_bytes, gnp, gep, 1, ggeo, gfc, st);
    // the two cutoff-factor gradients (heads, attention key biases) are added by the geometry kernel
    if (!rc) rc = backward_geometry_generic(m, g, w.dv, ggeo, E > 0 ? gfh : nullptr, E > 0 ? gfc : nullptr, gpos, gcell, st);
    PET_HIP_CHECK(hipFreeAsync(gn, st));
    return rc;
}

// staged adjoint of the fused head on the features the forward left in the workspace (pet_backward_predict)
int gen_backward_predict(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* gA, float* g_node,
                         float* g_edge, float* g_fc, hipStream_t st) {
    GWs w;
    gen_carve(m, g.n_nodes, g.n_edges, ws, w);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "workspace too small");
    PET_REQUIRE(!m.residual() && m.has_fused_head, PET_ERR_ARGUMENT, "pet_backward_predict needs the fused single-property target");
    if (g.n_nodes == 0) return PET_OK;
    const GD d = dims_of(m);
    HeadW H; H.nh0 = m.nh0; H.nh2 = m.nh2; H.eh0 = m.eh0; H.eh2 = m.eh2;
    const GGnn& last = w.gnn.back();
    const int64_t Ea = g.n_edges > 0 ? g.n_edges : 1;
    float* tmp = nullptr;   // outputs the caller did not ask for
    PET_HIP_CHECK(hipMallocAsync((void**)&tmp, (size_t)(g.n_nodes * d.DN + Ea * d.D + Ea) * sizeof(float), st));
    int rc = gen_predict_backward(m, g, H, m.lasts.at("@|0|@"), last.Hout, last.Mout, g.fc, gA, g_node ? g_node : tmp,
                                  g_edge ? g_edge : tmp + g.n_nodes * d.DN, g_fc ? g_fc : tmp + g.n_nodes * d.DN + Ea * d.D, st);
    PET_HIP_CHECK(hipFreeAsync(tmp, st));
    return rc;
}

int gen_backward_geometry(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* g_geo, const float* g_fc,
                          float* gpos, float* gcell, hipStream_t st) {
    GWs w;
    gen_carve(m, g.n_nodes, g.n_edges, ws, w);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "workspace too small");
    if (g.n_nodes == 0) return PET_OK;
    return backward_geometry_generic(m, g, w.dv, g_geo, g_fc, nullptr, gpos, gcell, st);
}

}  // namespace pet
