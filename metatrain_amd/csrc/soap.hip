// SOAP-BPNN hot path for gfx950: spherical expansion, power spectrum,
// LayerNorm + MLP tail, and the hand-written reverse pass for dE/dR. See include/soap_hip.h.
//
//   c[i][l][m][n][a] = sum_{p in row i} Y_lm(r_p / |r_p|) R_nl(|r_p|) fc(|r_p|) w_a(species of neighbour)   (a17, spex)
//   ps[i][l][(n a), (n' a')] = sum_m c[i][l][m][n a] c[i][l][m][n' a']        power_spectrum.py:125-136
//   e_i = w3 . silu(W2 silu(W1 LN(ps_i * enc[s_i])))                           model.py:553-595, 1204-1219
//
// Everything is HBM / latency bound (a few hundred FLOP per byte only in the tail's first Linear);
// one workgroup per atom, per-pair quantities staged in LDS, CSR rows give coalesced per-atom reads,
// reductions are fixed-order (no float atomics): bit-reproducible.
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/soap_hip.h"
#include "common.h"
#include "model.h"
#include "tile.h"

namespace pet {

// pet_bwd.hip (shared geometry adjoint: dE/dR_a = sum_{p in row a} (dv[rev p] - dv[p]), dE/dcell)
__global__ void k_pos_grad(const float4* __restrict__ dv, const int* __restrict__ rowptr, const int* __restrict__ rev,
                           float* __restrict__ gpos, int N);
__global__ void k_cell_grad(const float4* __restrict__ dv, const int* __restrict__ shift, const int* __restrict__ ctr,
                            const int* __restrict__ sys, const int* __restrict__ rowptr, float* __restrict__ gcell,
                            int N, int64_t E, int accumulate);

// abi.hip
__global__ void k_pack(const float* __restrict__ W, int64_t s_n, int64_t s_k, int n_out, int k_in,
                       float4* __restrict__ out);
static int g_soap_mfma = 1;  // pet_config_set("soap_mfma", 0) selects the per-atom tail kernels
void set_soap_mfma(int v) { g_soap_mfma = v ? 1 : 0; }

constexpr int MAXL = SOAP_MAX_L;
constexpr int PC = 32;  // pairs per LDS chunk

struct SoapDims {
    int L, C, F, NLM, NCOEF, ITEMS, S, H, NH, ns, legacy, layernorm, n_grid;
    int n_per_l[MAXL + 1], rad_off[MAXL + 1], coef_off[MAXL + 2], feat_off[MAXL + 2];
    int ncmax;             // largest n_per_l[l] * C
    // packed power spectrum (round 6): p_l[a][b] = p_l[b][a], so the inference path stores the upper triangle a <= b of every
    // l block only -- Sp = sum_l nc (nc + 1) / 2 floats per atom instead of S = sum_l nc^2 (2 360 against 4 544 for the default
    // basis) -- at pfeat_off[l] + a nc - a (a - 1) / 2 + (b - a); the first Linear and the LayerNorm are folded accordingly
    // (k_soap_prep_wallp). Halves the feature traffic of the four kernels that stream [N][S].
    int Sp, pfeat_off[MAXL + 2];
    float rc, width, inv_h;
};
__host__ __device__ __forceinline__ int soap_tri(int nc, int lo, int hi) { return lo * nc - lo * (lo - 1) / 2 + (hi - lo); }

constexpr int MAXNH = 8;  // hidden layers of the tail (soap_bpnn/documentation.py: num_hidden_layers, default 2)
struct SoapSet {  // weights of one (centre-species) set
    const float *ln_w = nullptr, *ln_b = nullptr, *W1 = nullptr, *W2 = nullptr, *w3 = nullptr;
    const float* Wh[MAXNH - 1] = {};  // hidden Linear k + 2 of the stack ("bpnn.<s>.<2 (k + 1)>.weight"); Wh[0] == W2
};

struct SoapModel {
    soap_hypers_t h;
    SoapDims d;
    std::map<std::string, std::pair<float*, int64_t>> raw;
    // training (soap_train.h): gradient slot and Adam moments of every trainable parameter, keyed like `raw`
    std::map<std::string, std::pair<float*, int64_t>> grad;
    std::map<std::string, float*> adam_m, adam_v;
    std::vector<void*> owned;
    float* table = nullptr;      // [n_grid][F][2]
    float* shnorm = nullptr;     // [(L+1)*(L+1)] normalisation (sqrt 2 folded in for m > 0)
    int* coef_lut = nullptr;     // [NCOEF] lm | rn << 8 | a << 16
    int* item_lut = nullptr;     // [ITEMS] lm | rn << 8 | base << 16
    int2* feat_lut = nullptr;    // [S]     coefficient offsets of c[.][a], c[.][b] | M << 16 | nc << 24
    int2* out_lut = nullptr;     // [NCOEF] offset of G[0][a] in the feature row, offset of c[m][0] | nc << 16
    float* species_w = nullptr;  // [ns, C]
    const float* enc = nullptr;  // [ns, S] or null
    SoapSet* sets = nullptr;     // device array [n_sets]
    int n_sets = 0;
    // MFMA tail: all sets' first Linear stacked (LayerNorm weight folded in), zero padded to [NOUTP][Kp]
    int NT = 0, NOUTP = 0, Kp = 0;
    float* wall = nullptr;       // [NOUTP][Kp]
    float4 *wall_fwd = nullptr, *wall_bwd = nullptr;
    float *wall_rs = nullptr, *wall_b = nullptr;  // [NOUTP] row sums, W1 beta
    float4 *wall_fwd_set = nullptr, *wall_bwd_set = nullptr;  // per network: [n_sets][1][Kp/8][64], [n_sets][Kp/32][4][64]
    // packed power spectrum (SoapDims::Sp): the first Linear of every network over the upper-triangle layout,
    // W'[j][(a, b)] = gamma W1[j][(a, b)] + gamma W1[j][(b, a)] (a < b), for the adjoint with the diagonal doubled
    int Kpp = 0;
    float *wallp = nullptr, *wallpb = nullptr;
    float4 *wallp_fwd_set = nullptr, *wallp_bwd_set = nullptr;
    // which layout the last forward into a workspace left in its feature buffer (true = packed); soap_bwd and the training
    // pass follow it (the training pass rebuilds the full layout from the stored expansion coefficients)
    mutable std::map<const void*, bool> ws_packed;
    bool finalized = false;
};

static int salloc(SoapModel& m, void** p, size_t bytes) {
    PET_HIP_CHECK(hipMalloc(p, bytes > 0 ? bytes : 4));
    m.owned.push_back(*p);
    return PET_OK;
}

// ---------------------------------------------------------------------------------------------
// per-pair building blocks
// ---------------------------------------------------------------------------------------------
// Orthonormal real spherical harmonics of a unit vector (x, y, z), Y[l*l + l + m], by the standard recurrences
//   c_m + i s_m = (x + i y)^m,   Q_m^m = (-1)^m (2m-1)!!,   Q_{m+1}^m = (2m+1) z Q_m^m,
//   Q_l^m = ((2l-1) z Q_{l-1}^m - (l+m-1) Q_{l-2}^m) / (l-m),     Y_l^{+-m} = F_lm Q_l^m {c_m | s_m}
// (shn[l*(L+1)+m] = F_lm, with sqrt(2) folded in for m > 0), and the gradient of their polynomial extension
// by differentiating the recurrences.
// One m-chain (all l >= m for one pair): the per-pair work is split over (pair, m) threads.
__device__ __forceinline__ void sh_chain(float x, float y, float z, int m, int L, const float* __restrict__ shn,
                                         float* Y, float* Gx, float* Gy, float* Gz, float ir) {
    float cm = 1.f, sm = 0.f, cp = 1.f, sp_ = 0.f;  // (c_m, s_m) and (c_{m-1}, s_{m-1})
    float qmm = 1.f;
    for (int k = 1; k <= m; k++) {
        cp = cm; sp_ = sm;
        cm = x * cp - y * sp_;
        sm = x * sp_ + y * cp;
        qmm *= -(2 * k - 1);
    }
    float q2 = 0.f, dq2 = 0.f, q1 = qmm, dq1 = 0.f;
    for (int l = m; l <= L; l++) {
        float q, dq;
        if (l == m) { q = qmm; dq = 0.f; }
        else if (l == m + 1) { q = (2 * m + 1) * z * q1; dq = (2 * m + 1) * q1; }
        else {
            const float inv = 1.0f / (l - m);
            q = ((2 * l - 1) * z * q1 - (l + m - 1) * q2) * inv;
            dq = ((2 * l - 1) * (q1 + z * dq1) - (l + m - 1) * dq2) * inv;
        }
        const float f = shn[l * (L + 1) + m];
        const int ip = l * l + l + m, im = l * l + l - m;
        // value and gradient of the polynomial extension, then the chain through u = v / r: (I - u u^T) / r
        float yv[2], gx[2], gy[2], gz[2];
        if (m == 0) {
            yv[0] = f * q; gx[0] = 0.f; gy[0] = 0.f; gz[0] = f * dq;
        } else {
            const float fm = f * q * m;
            yv[0] = f * q * cm; gx[0] = fm * cp;  gy[0] = -fm * sp_; gz[0] = f * dq * cm;
            yv[1] = f * q * sm; gx[1] = fm * sp_; gy[1] = fm * cp;   gz[1] = f * dq * sm;
        }
        for (int k = 0; k < (m == 0 ? 1 : 2); k++) {
            const int idx = k == 0 ? ip : im;
            Y[idx] = yv[k];
            if (Gx) {
                const float dot = x * gx[k] + y * gy[k] + z * gz[k];
                Gx[idx] = (gx[k] - x * dot) * ir; Gy[idx] = (gy[k] - y * dot) * ir; Gz[idx] = (gz[k] - z * dot) * ir;
            }
        }
        if (l > m) { q2 = q1; dq2 = dq1; }
        q1 = q; dq1 = dq;
        if (l == m) { q2 = 0.f; dq2 = 0.f; }
    }
}

// Hermite spline of radial function f at distance r: R fc and (optionally) d(R fc)/dr
__device__ __forceinline__ void radial_one(const SoapDims& d, const float* __restrict__ table, int f, float r, float fc,
                                           float dfc, float* R, float* dR) {
    float t = r * d.inv_h;
    int k = (int)t;
    if (k > d.n_grid - 2) k = d.n_grid - 2;
    const float s = t - k, h = 1.0f / d.inv_h;
    const float s2 = s * s, om = 1.f - s;
    const float h00 = (1.f + 2.f * s) * om * om, h10 = s * om * om, h01 = s2 * (3.f - 2.f * s), h11 = s2 * (s - 1.f);
    // table entry: (R, dR/dr, chord slope (R[k+1] - R[k]) / h, 0). The derivative of the Hermite cubic needs the chord
    // slope; taken from the fp32 node values it is a difference of two numbers 2e-3 relative apart times 1 / h = 410:
    // 4e-5 |R| of rounding noise in dR/dr, which showed as 1.1e-5 .. 1.5e-5 in dE/dR of dilute systems (round-2 sweep,
    // tests/debug/fuzz_soap.py). The host computes it in fp64.
    const float4 a = reinterpret_cast<const float4*>(table)[(size_t)k * d.F + f];
    const float4 b = reinterpret_cast<const float4*>(table)[(size_t)(k + 1) * d.F + f];
    const float v = h00 * a.x + h10 * h * a.y + h01 * b.x + h11 * h * b.y;
    *R = v * fc;
    if (dR) {
        const float g00 = 6.f * s2 - 6.f * s, g10 = 3.f * s2 - 4.f * s + 1.f, g11 = 3.f * s2 - 2.f * s;
        const float dv = g10 * a.y + g11 * b.y - g00 * a.z;
        *dR = dv * fc + v * dfc;
    }
}

__device__ __forceinline__ float shifted_cosine(float r, float rc, float w, float* dfc) {
    float s = (r - (rc - w)) / w;
    if (r >= rc) { if (dfc) *dfc = 0.f; return 0.f; }
    if (s <= 0.f) { if (dfc) *dfc = 0.f; return 1.f; }
    if (dfc) *dfc = -0.5f * 3.14159274f * sinf(3.14159274f * s) / w;
    return 0.5f * (1.0f + cosf(3.14159274f * s));
}

// ---------------------------------------------------------------------------------------------
// a17: spherical expansion, one workgroup per atom
// ---------------------------------------------------------------------------------------------
constexpr int MAXK = 12;  // coefficients per thread (NCOEF <= 256 * MAXK)

__global__ __launch_bounds__(256) void k_soap_expand(SoapDims d, const float4* __restrict__ geo,
                                                     const int* __restrict__ rowptr, const int* __restrict__ sp_nbr,
                                                     const float* __restrict__ table, const float* __restrict__ shn,
                                                     const int* __restrict__ lut, const float* __restrict__ spw,
                                                     float* __restrict__ Cf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ys = smem;                     // [PC][NLM]
    float* Rs = Ys + PC * d.NLM;          // [PC][F]
    float* us = Rs + PC * d.F;            // [PC][8] unit vector, r, 1/r, fc, dfc
    int* sps = reinterpret_cast<int*>(us + PC * 8);
    const int i = blockIdx.x, tid = threadIdx.x;
    const int p0 = rowptr[i], p1 = rowptr[i + 1];
    float acc[MAXK];
    int code[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
        acc[k] = 0.f;
        const int idx = tid + 256 * k;
        code[k] = idx < d.NCOEF ? lut[idx] : -1;
    }
    for (int base = p0; base < p1; base += PC) {
        const int npc = min(PC, p1 - base);
        __syncthreads();
        if (tid < npc) {
            const float4 g = geo[base + tid];
            const float r = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
            const float ir = r > 0.f ? 1.0f / r : 0.f;
            float* u = us + tid * 8;
            u[0] = g.x * ir; u[1] = g.y * ir; u[2] = g.z * ir; u[3] = r; u[4] = ir;
            u[5] = shifted_cosine(r, d.rc, d.width, nullptr);
            sps[tid] = sp_nbr[base + tid];
        }
        __syncthreads();
        // per-pair quantities, spread over the workgroup: (pair, m) chains of Y_lm and (pair, f) spline values
        for (int idx = tid; idx < npc * (d.L + 1); idx += 256) {
            const int pp = idx / (d.L + 1), mm = idx % (d.L + 1);
            const float* u = us + pp * 8;
            sh_chain(u[0], u[1], u[2], mm, d.L, shn, Ys + pp * d.NLM, nullptr, nullptr, nullptr, 0.f);
        }
        for (int idx = tid; idx < npc * d.F; idx += 256) {
            const int pp = idx / d.F, f = idx % d.F;
            radial_one(d, table, f, us[pp * 8 + 3], us[pp * 8 + 5], 0.f, Rs + pp * d.F + f, nullptr);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXK; k++) {
            if (code[k] < 0) continue;
            const int lm = code[k] & 255, rn = (code[k] >> 8) & 255, a = code[k] >> 16;
            float s = acc[k];
            for (int pp = 0; pp < npc; pp++) s += Ys[pp * d.NLM + lm] * Rs[pp * d.F + rn] * spw[sps[pp] * d.C + a];
            acc[k] = s;
        }
    }
#pragma unroll
    for (int k = 0; k < MAXK; k++)
        if (code[k] >= 0) Cf[(size_t)i * d.NCOEF + tid + 256 * k] = acc[k];
}

// power spectrum (+ centre encoding): feats[i][feat_off[l] + p1 * nc + p2] = sum_m c[l][m][p1] c[l][m][p2]
__device__ __forceinline__ float block_sum(float v, float* red /*[8]*/);

__global__ __launch_bounds__(256) void k_soap_ps(SoapDims d, const float* __restrict__ Cf, const int* __restrict__ sp,
                                                 const float* __restrict__ enc, float* __restrict__ feats,
                                                 float* __restrict__ tail) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[8];
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < d.NCOEF; k += 256) smem[k] = Cf[(size_t)i * d.NCOEF + k];
    __syncthreads();
    float sum = 0.f;
    const float* e = enc ? enc + (size_t)sp[i] * d.S : nullptr;
    for (int idx = threadIdx.x; idx < d.S; idx += 256) {
        int l = 0;
        while (idx >= d.feat_off[l + 1]) l++;
        const int nc = d.n_per_l[l] * d.C, q = idx - d.feat_off[l];
        const int a = q / nc, b = q % nc;
        const float* base = smem + d.coef_off[l];
        float s = 0.f;
        for (int m = 0; m < 2 * l + 1; m++) s += base[m * nc + a] * base[m * nc + b];
        const float v = e ? s * e[idx] : s;
        feats[(size_t)i * d.S + idx] = v;
        sum += v;
    }
    if (!d.layernorm) return;
    // LayerNorm statistics (two passes: mean, then centred variance) for the tail
    const float mean = block_sum(sum, red) / d.S;
    float var = 0.f;
    for (int idx = threadIdx.x; idx < d.S; idx += 256) {
        const float c = feats[(size_t)i * d.S + idx] - mean;  // this thread's own writes
        var += c * c;
    }
    const float rstd = rsqrtf(block_sum(var, red) / d.S + 1e-5f);
    if (threadIdx.x == 0) {
        tail[(size_t)i * (2 + 2 * d.H)] = mean;
        tail[(size_t)i * (2 + 2 * d.H) + 1] = rstd;
    }
}

__device__ __forceinline__ float block_sum(float v, float* red /*[8]*/) {
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu(float x) { return x * sigm(x); }
__device__ __forceinline__ float dsilu(float x) { const float s = sigm(x); return s * (1.0f + x * (1.0f - s)); }

// ---------------------------------------------------------------------------------------------
// a18: LayerNorm + MLP + last layer, one workgroup per atom. tail[i] = {mean, rstd, a1[H], a2[H]}
// ---------------------------------------------------------------------------------------------
constexpr int MAXH = 64;  // neurons per hidden layer; the MFMA tails serve 32 (the default), other widths the per-atom kernels

__global__ __launch_bounds__(256) void k_soap_tail(SoapDims d, const float* __restrict__ feats,
                                                   const int* __restrict__ sp, const SoapSet* __restrict__ sets,
                                                   float* __restrict__ tail, float* __restrict__ atomic) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                 // [S] normalised input
    float* part = xs + d.S;           // [256][H+1]
    float* red = part + 256 * (d.H + 1);  // [8]
    float* hs = red + 8;              // [2*H]
    const int i = blockIdx.x, tid = threadIdx.x;
    const SoapSet W = sets[d.legacy ? sp[i] : 0];
    const float* x = feats + (size_t)i * d.S;
    float mean = 0.f, rstd = 1.f;
    if (d.layernorm) {
        float s = 0.f;
        for (int k = tid; k < d.S; k += 256) s += x[k];
        mean = block_sum(s, red) / d.S;
        float v = 0.f;
        for (int k = tid; k < d.S; k += 256) { const float c = x[k] - mean; v += c * c; }
        rstd = rsqrtf(block_sum(v, red) / d.S + 1e-5f);
        for (int k = tid; k < d.S; k += 256) xs[k] = (x[k] - mean) * rstd * W.ln_w[k] + W.ln_b[k];
    } else {
        for (int k = tid; k < d.S; k += 256) xs[k] = x[k];
    }
    __syncthreads();
    float acc[MAXH];
#pragma unroll
    for (int j = 0; j < MAXH; j++) acc[j] = 0.f;
    for (int k = tid; k < d.S; k += 256) {
        const float xv = xs[k];
#pragma unroll
        for (int j = 0; j < MAXH; j++)
            if (j < d.H) acc[j] += W.W1[(size_t)j * d.S + k] * xv;
    }
#pragma unroll
    for (int j = 0; j < MAXH; j++)
        if (j < d.H) part[tid * (d.H + 1) + j] = acc[j];
    __syncthreads();
    if (tid < d.H) {
        float s = 0.f;
        for (int t = 0; t < 256; t++) s += part[t * (d.H + 1) + tid];
        hs[tid] = s;  // a1
    }
    __syncthreads();
    float a2 = 0.f;
    if (tid < d.H && d.NH > 1) {
        for (int q = 0; q < d.H; q++) a2 += W.W2[tid * d.H + q] * silu(hs[q]);
        hs[d.H + tid] = a2;
    }
    __syncthreads();
    if (tid < 64) {
        float e = 0.f;
        if (tid < d.H) e = W.w3[tid] * silu(d.NH > 1 ? hs[d.H + tid] : hs[tid]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
        if (tid == 0) atomic[i] = e;
    }
    float* tl = tail + (size_t)i * (2 + 2 * d.H);
    if (tid == 0) { tl[0] = mean; tl[1] = rstd; }
    if (tid < 2 * d.H) tl[2 + tid] = hs[tid];
}

// ---- MFMA tail: 64 atoms per workgroup, first Linear of ALL sets as one [64 x Kp] x [Kp x NOUTP] GEMM ----
__global__ void k_soap_prep_wall(SoapDims d, const SoapSet* __restrict__ sets, int n_sets, int NOUTP, int Kp,
                                 float* __restrict__ wall) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)NOUTP * Kp) return;
    const int o = (int)(idx / Kp), k = (int)(idx % Kp);
    float v = 0.f;
    if (o < n_sets * d.H && k < d.S) {
        const SoapSet W = sets[o / d.H];
        v = W.W1[(size_t)(o % d.H) * d.S + k];
        if (d.layernorm) v *= W.ln_w[k];
    }
    wall[idx] = v;
}
__global__ void k_soap_prep_rows(SoapDims d, const SoapSet* __restrict__ sets, int n_sets, int NOUTP, int Kp,
                                 const float* __restrict__ wall, float* __restrict__ rs, float* __restrict__ bs) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NOUTP) return;
    double s = 0.0, b = 0.0;
    if (o < n_sets * d.H) {
        const SoapSet W = sets[o / d.H];
        for (int k = 0; k < d.S; k++) {
            s += wall[(size_t)o * Kp + k];
            if (d.layernorm) b += (double)W.W1[(size_t)(o % d.H) * d.S + k] * W.ln_b[k];
        }
    }
    rs[o] = (float)s;
    bs[o] = (float)b;
}

// The first Linear over the packed (upper-triangle) feature layout, one network after the other: [n_sets * H][Kpp].
// diag2: the adjoint's copy -- its GEMM output is G[a][b] = dx[a][b] + dx[b][a], which on the diagonal is TWICE dx[a][a].
__global__ void k_soap_prep_wallp(SoapDims d, const SoapSet* __restrict__ sets, int n_sets, int Kpp, int diag2,
                                  float* __restrict__ wallp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_sets * d.H * Kpp) return;
    const int o = (int)(idx / Kpp), t = (int)(idx % Kpp);
    float v = 0.f;
    if (t < d.Sp) {
        int l = 0;
        while (t >= d.pfeat_off[l + 1]) l++;
        const int nc = d.n_per_l[l] * d.C;
        int q = t - d.pfeat_off[l], a = 0;
        while (q >= nc - a) { q -= nc - a; a++; }
        const int b = a + q;
        const SoapSet W = sets[o / d.H];
        const float* w1 = W.W1 + (size_t)(o % d.H) * d.S + d.feat_off[l];
        const float* lw = W.ln_w + d.feat_off[l];
        const int k1 = a * nc + b, k2 = b * nc + a;
        v = w1[k1] * (d.layernorm ? lw[k1] : 1.f);
        if (a != b) v += w1[k2] * (d.layernorm ? lw[k2] : 1.f);
        else if (diag2) v *= 2.f;
    }
    wallp[idx] = v;
}

template <int NT>
__global__ __launch_bounds__(NTHREADS) void k_soap_tail_fwd_mfma(SoapDims d, const float* __restrict__ feats,
                                                                 const int* __restrict__ sp,
                                                                 const SoapSet* __restrict__ sets,
                                                                 const float4* __restrict__ Wp, int Kp,
                                                                 const float* __restrict__ rs,
                                                                 const float* __restrict__ bs, float* __restrict__ tail,
                                                                 float* __restrict__ atomic, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NOUTP = 64 * NT, LDO = NOUTP + 1, LDA = lds_ld(128);
    float* As = smem;                    // [64][132], later out [64][NOUTP + 1]
    float* out = smem;
    float* a1s = smem + BM * (LDO > LDA ? LDO : LDA);  // [64][32] silu(a1)
    const WaveId w;
    const int row0 = blockIdx.x * BM;
    f32x16 acc[NT];
    acc_fill_bias<NT>(acc, nullptr, 0, w.lane);
    for (int kc = 0; kc < Kp / 128; kc++) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < BM * 32; idx += NTHREADS) {
            const int r = idx >> 5, c = idx & 31, col = 128 * kc + 4 * c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < N && col < d.S) v = *reinterpret_cast<const float4*>(feats + (size_t)(row0 + r) * d.S + col);
            *reinterpret_cast<float4*>(As + r * LDA + 4 * c) = v;
        }
        __syncthreads();
        gemm_acc<128, NT>(As + w.rb * 32 * LDA, LDA, Wp, Kp / 8, 16 * kc, NT * w.ch, acc, w.lane);
    }
    __syncthreads();
    acc_foreach<NT>(acc, w.rb, 32 * NT * w.ch, w.lane, [&](int r, int c, float v) { out[r * LDO + c] = v; });
    __syncthreads();
    const int H = 32, TS = 2 + 2 * H;
    for (int item = threadIdx.x; item < BM * H; item += NTHREADS) {
        const int r = item >> 5, j = item & 31, atom = row0 + r;
        float a1 = 0.f;
        if (atom < N) {
            const int s = d.legacy ? sp[atom] : 0;
            const float mu = d.layernorm ? tail[(size_t)atom * TS] : 0.f;
            const float rstd = d.layernorm ? tail[(size_t)atom * TS + 1] : 1.f;
            a1 = rstd * (out[r * LDO + s * H + j] - mu * rs[s * H + j]) + bs[s * H + j];
            tail[(size_t)atom * TS + 2 + j] = a1;
        }
        a1s[r * H + j] = silu(a1);
    }
    __syncthreads();
    for (int item = threadIdx.x; item < BM * H; item += NTHREADS) {
        const int r = item >> 5, j = item & 31, atom = row0 + r;
        float e = 0.f;
        if (atom < N) {
            const SoapSet W = sets[d.legacy ? sp[atom] : 0];
            if (d.NH > 1) {
                float a2 = 0.f;
                for (int q = 0; q < H; q++) a2 += W.W2[j * H + q] * a1s[r * H + q];
                tail[(size_t)atom * TS + 2 + H + j] = a2;
                e = W.w3[j] * silu(a2);
            } else {
                e = W.w3[j] * a1s[r * H + j];
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) e += __shfl_xor(e, o);
        if (j == 0 && atom < N) atomic[atom] = e;
    }
}

template <int NT>
__global__ __launch_bounds__(NTHREADS) void k_soap_tail_bwd_mfma(SoapDims d, const float* __restrict__ feats,
                                                                 const int* __restrict__ sp,
                                                                 const SoapSet* __restrict__ sets,
                                                                 const float4* __restrict__ Wpb, int Kp,
                                                                 const float* __restrict__ rs,
                                                                 const float* __restrict__ bs,
                                                                 const float* __restrict__ enc,
                                                                 const float* __restrict__ tail,
                                                                 const float* __restrict__ gA, float* __restrict__ dF,
                                                                 int N, const float* __restrict__ da2x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NOUTP = 64 * NT, LDD = lds_ld(NOUTP);
    float* Ds = smem;                 // [64][NOUTP + 4] d a1, zero outside the atom's own set
    float* d2 = Ds + BM * LDD;        // [64][32] d a2
    float* st = d2 + BM * 32;         // [64][4] mean, rstd, m1, m2
    float* ot = st + BM * 4;          // [64][132] output tile staging: full-row (512 B) reads / writes below
    const WaveId w;
    const int row0 = blockIdx.x * BM;
    const int H = 32, TS = 2 + 2 * H;
    for (int idx = threadIdx.x; idx < BM * LDD; idx += NTHREADS) Ds[idx] = 0.f;
    for (int item = threadIdx.x; item < BM * H; item += NTHREADS) {
        const int r = item >> 5, j = item & 31, atom = row0 + r;
        float v = 0.f;
        if (atom < N && d.NH > 1) {
            const SoapSet W = sets[d.legacy ? sp[atom] : 0];
            v = da2x ? da2x[(size_t)atom * H + j] : gA[atom] * W.w3[j] * dsilu(tail[(size_t)atom * TS + 2 + H + j]);
        }
        d2[r * H + j] = v;
    }
    __syncthreads();
    for (int item = threadIdx.x; item < BM * H; item += NTHREADS) {
        const int r = item >> 5, j = item & 31, atom = row0 + r;
        float da1 = 0.f, t1 = 0.f, t2 = 0.f;
        float mu = 0.f, rstd = 1.f;
        if (atom < N) {
            const int s = d.legacy ? sp[atom] : 0;
            const SoapSet W = sets[s];
            const float a1 = tail[(size_t)atom * TS + 2 + j];
            if (d.NH > 1) {
                float acc = 0.f;
                for (int q = 0; q < H; q++) acc += W.W2[q * H + j] * d2[r * H + q];
                da1 = acc * dsilu(a1);
            } else {
                da1 = gA[atom] * W.w3[j] * dsilu(a1);
            }
            Ds[r * LDD + s * H + j] = da1;
            if (d.layernorm) {
                mu = tail[(size_t)atom * TS];
                rstd = tail[(size_t)atom * TS + 1];
                const float rsj = rs[s * H + j];
                const float raw = (a1 - bs[s * H + j]) / rstd + mu * rsj;  // Wall_s[j] . x
                t1 = da1 * rsj;
                t2 = da1 * raw;
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
        if (j == 0) {
            st[r * 4] = mu; st[r * 4 + 1] = rstd;
            st[r * 4 + 2] = t1 / d.S;                        // m1 = mean(dxn gamma)
            st[r * 4 + 3] = rstd * (t2 - mu * t1) / d.S;     // m2 = mean(dxn gamma xhat)
        }
    }
    __syncthreads();
    for (int nblk = 0; nblk < Kp / 128; nblk++) {
        f32x16 acc[2];
        acc_fill_bias<2>(acc, nullptr, 0, w.lane);
        gemm_acc<NOUTP, 2>(Ds + w.rb * 32 * LDD, LDD, Wpb, NOUTP / 8, 0, 4 * nblk + 2 * w.ch, acc, w.lane);
        __syncthreads();  // the previous block's staging tile has been consumed
        acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) { ot[r * lds_ld(128) + c] = v; });
        __syncthreads();
        for (int idx = threadIdx.x; idx < BM * 32; idx += NTHREADS) {
            const int r = idx >> 5, c4 = idx & 31, k = 128 * nblk + 4 * c4, atom = row0 + r;
            if (atom >= N || k >= d.S) continue;
            float4 v = *reinterpret_cast<const float4*>(ot + r * lds_ld(128) + 4 * c4);
            if (d.layernorm) {
                const float4 x = *reinterpret_cast<const float4*>(feats + (size_t)atom * d.S + k);
                const float mu = st[r * 4], rs_ = st[r * 4 + 1], m1 = st[r * 4 + 2], m2 = st[r * 4 + 3];
                v.x = rs_ * (v.x - m1 - (x.x - mu) * rs_ * m2);
                v.y = rs_ * (v.y - m1 - (x.y - mu) * rs_ * m2);
                v.z = rs_ * (v.z - m1 - (x.z - mu) * rs_ * m2);
                v.w = rs_ * (v.w - m1 - (x.w - mu) * rs_ * m2);
            }
            if (enc) {
                const float4 e = *reinterpret_cast<const float4*>(enc + (size_t)sp[atom] * d.S + k);
                v.x *= e.x; v.y *= e.y; v.z *= e.z; v.w *= e.w;
            }
            *reinterpret_cast<float4*>(dF + (size_t)atom * d.S + k) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Power spectrum and its adjoint, second generation (pet_config_set("soap_pair", 1), with the expansion kernels):
//   k_soap_ps_w     one wave per atom, features decoded by a table (no division per feature), LayerNorm statistics
//                   as fp64 sum / sum of squares in the same pass (the first version re-read its own output);
//   k_soap_ps_bwd_s one workgroup per atom: the atom's dF row is staged in LDS and symmetrised in place
//                   (G[a][b] = dF[a][b] + dF[b][a]), after which dC[m][a] = sum_b G[b][a] c[m][b] reads consecutive
//                   addresses across lanes (the first version re-read dF from global 2 (2l+1) times).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_soap_ps_w(SoapDims d, const float* __restrict__ Cf, const int* __restrict__ sp,
                                                   const float* __restrict__ enc, const int2* __restrict__ flut,
                                                   float* __restrict__ feats, float* __restrict__ tail, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    float* cs = smem + (size_t)wave * d.NCOEF;
    for (int k = lane; k < d.NCOEF; k += 64) cs[k] = Cf[(size_t)i * d.NCOEF + k];
    __builtin_amdgcn_wave_barrier();
    const float* e = enc ? enc + (size_t)sp[i] * d.S : nullptr;
    double s1 = 0.0, s2 = 0.0;
    for (int idx = lane; idx < d.S; idx += 64) {
        const int2 code = flut[idx];
        const int pa = code.x, pb = code.y & 0xffff, M = (code.y >> 16) & 0xff, nc = code.y >> 24;
        float v = 0.f;
        for (int m = 0; m < M; m++) v += cs[pa + m * nc] * cs[pb + m * nc];
        if (e) v *= e[idx];
        feats[(size_t)i * d.S + idx] = v;
        s1 += (double)v;
        s2 += (double)v * (double)v;
    }
    if (!d.layernorm) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) {
        const double mean = s1 / d.S, var = s2 / d.S - mean * mean;
        tail[(size_t)i * (2 + 2 * d.H)] = (float)mean;
        tail[(size_t)i * (2 + 2 * d.H) + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
    }
}

// The same on the fp32 matrix core (ncmax <= 32): p_l = C_l^T C_l is a [nc x (2l+1)] x [(2l+1) x nc] product and both MFMA
// operands are the SAME register -- lane (a = lane & 31, m = 2 s + (lane >> 5)) holds c[l][m][a] as A[a][m] and as B[m][a] --
// so an l block costs ceil((2l+1) / 2) v_mfma_f32_32x32x2_f32 and as many ds_read_b32 (exact fp32 products, fp32 sums, as
// the scalar loop). One wave per atom; the 32 x 32 tile leaves as rows of nc consecutive floats; LayerNorm statistics in fp64.
// PACKED: only the upper triangle a <= b leaves (SoapDims::Sp floats per atom); the LayerNorm statistics count an
// off-diagonal entry twice, so mean and rstd are those of the full S-vector.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_soap_ps_m(SoapDims d, const float* __restrict__ Cf, const int* __restrict__ sp,
                                                   const float* __restrict__ enc, float* __restrict__ feats,
                                                   float* __restrict__ tail, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    // PACKED: behind the coefficients a staging tile for one l block's triangle (528 floats for nc = 32): the triangle's rows are
    // nc - p1 floats long, so storing them from the accumulator layout is 16 partial-line store instructions per block; through
    // the tile the block leaves as whole 8-byte-per-lane stores (every block starts at an even float: nc is a multiple of C = 4)
    const int stg_len = PACKED ? (d.ncmax * (d.ncmax + 1) / 2 + 3) & ~3 : 0;
    float* cs = smem + (size_t)wave * (d.NCOEF + stg_len);
    float* stg = cs + d.NCOEF;
    for (int k = lane; k < d.NCOEF; k += 64) cs[k] = Cf[(size_t)i * d.NCOEF + k];
    __builtin_amdgcn_wave_barrier();
    const float* e = enc ? enc + (size_t)sp[i] * d.S : nullptr;
    float* out = feats + (size_t)i * (PACKED ? d.Sp : d.S);
    const int a = lane & 31, mh = lane >> 5;
    double s1 = 0.0, s2 = 0.0;
    for (int l = 0; l <= d.L; l++) {
        const int nc = d.n_per_l[l] * d.C, M = 2 * l + 1;
        const float* c = cs + d.coef_off[l];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        {   // (at most MAXL + 1 K steps: unrolled, the LDS reads of a block in flight together)
            float cv[MAXL + 1];
#pragma unroll
            for (int s9 = 0; s9 <= MAXL; s9++) {
                const int m = 2 * s9 + mh;
                const bool ok = m < M && a < nc;
                const float t = c[ok ? m * nc + a : 0];
                cv[s9] = ok ? t : 0.f;
            }
#pragma unroll
            for (int s9 = 0; s9 <= MAXL; s9++)
                if (2 * s9 < M) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cv[s9], cv[s9], acc, 0, 0, 0);
        }
        if (a < nc) {  // this lane: column b = a of rows p1(r)
            const int f0 = d.feat_off[l];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int p1 = (r & 3) + 8 * (r >> 2) + 4 * mh;
                if (PACKED) {
                    if (p1 <= a) {  // row p1 of the triangle: nc - p1 consecutive floats across the lanes
                        const float v = acc[r];
                        stg[soap_tri(nc, p1, a)] = v;
                        const double wv = p1 == a ? (double)v : 2.0 * (double)v;
                        s1 += wv;
                        s2 += wv * (double)v;
                    }
                } else if (p1 < nc) {
                    const int idx = f0 + p1 * nc + a;
                    float v = acc[r];
                    if (e) v *= e[idx];
                    out[idx] = v;
                    s1 += (double)v;
                    s2 += (double)v * (double)v;
                }
            }
        }
        if (PACKED) {
            __builtin_amdgcn_wave_barrier();
            const int n2 = nc * (nc + 1) / 4;  // float2 per block (nc (nc + 1) / 2 is even)
            float2* dst = reinterpret_cast<float2*>(out + d.pfeat_off[l]);
            for (int k = lane; k < n2; k += 64) dst[k] = reinterpret_cast<const float2*>(stg)[k];
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!d.layernorm) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) {
        const double mean = s1 / d.S, var = s2 / d.S - mean * mean;
        tail[(size_t)i * (2 + 2 * d.H)] = (float)mean;
        tail[(size_t)i * (2 + 2 * d.H) + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
    }
}

__global__ __launch_bounds__(256) void k_soap_ps_bwd_s(SoapDims d, const float* __restrict__ Cf,
                                                       const float* __restrict__ dF, const int2* __restrict__ olut,
                                                       float* __restrict__ dCf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cs = smem;             // [NCOEF]
    float* G = smem + d.NCOEF;    // [S]
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < d.NCOEF; k += 256) cs[k] = Cf[(size_t)i * d.NCOEF + k];
    for (int k = threadIdx.x; k < d.S / 4; k += 256)
        reinterpret_cast<float4*>(G)[k] = reinterpret_cast<const float4*>(dF + (size_t)i * d.S)[k];
    __syncthreads();
    for (int l = 0; l <= d.L; l++) {  // symmetrise each block in place: pairs a < b, the diagonal doubles
        const int nc = d.n_per_l[l] * d.C;
        float* g = G + d.feat_off[l];
        for (int q = threadIdx.x; q < nc * nc; q += 256) {
            const int a = q / nc, b = q - a * nc;
            if (a > b) continue;
            const float t = g[a * nc + b] + g[b * nc + a];
            g[a * nc + b] = t;
            g[b * nc + a] = t;
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < d.NCOEF; o += 256) {
        const int2 code = olut[o];
        const int g0 = code.x, c0 = code.y & 0xffff, nc = code.y >> 16;
        double v = 0.0;
        for (int b = 0; b < nc; b++) v += (double)(G[g0 + b * nc] * cs[c0 + b]);
        dCf[(size_t)i * d.NCOEF + o] = (float)v;
    }
}

// The adjoint on the fp32 matrix core: dC_l [(2l+1) x nc] = C_l [(2l+1) x nc] (dF_l + dF_l^T) [nc x nc] as 16 x 16 x 4 tiles (m
// is at most 2 L + 1 = 13 .. 17 rows: the 32-row tile would waste more than half of itself), two MFMAs per K step -- one
// with dF[b][a], one with dF[a][b] -- instead of a symmetrisation pass over the staged row; the workgroup's four waves take
// the l blocks in turn. fp32 products and sums (the scalar kernel summed its 28 terms in fp64).
// PACKED: dF arrives as the symmetrised upper triangle G[a][b] = dx[a][b] + dx[b][a] (k_soap_tail_bwd_set with the packed
// weights): one MFMA per K step, half the staged row.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <bool PACKED>
__global__ __launch_bounds__(256) void k_soap_ps_bwd_m(SoapDims d, const float* __restrict__ Cf,
                                                       const float* __restrict__ dF, float* __restrict__ dCf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cs = smem;             // [NCOEF]
    float* G = smem + d.NCOEF;    // [S] (PACKED: [Sp])
    const int i = blockIdx.x;
    const int SF = PACKED ? d.Sp : d.S;
    for (int k = threadIdx.x; k < d.NCOEF; k += 256) cs[k] = Cf[(size_t)i * d.NCOEF + k];
    for (int k = threadIdx.x; k < SF / 4; k += 256)
        reinterpret_cast<float4*>(G)[k] = reinterpret_cast<const float4*>(dF + (size_t)i * SF)[k];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    // (collecting the result in LDS and storing it as whole lines behind a workgroup barrier was measured in round 6: 0.63 against
    // 0.56 ms -- the four waves' own 64-byte row pieces overlap with the slower waves' products)
    float* out = dCf + (size_t)i * d.NCOEF;
    for (int l = wave; l <= d.L; l += 4) {
        const int nc = d.n_per_l[l] * d.C, M = 2 * l + 1;
        const float* c = cs + d.coef_off[l];
        const float* g = G + (PACKED ? d.pfeat_off[l] : d.feat_off[l]);
        for (int m0 = 0; m0 < M; m0 += 16)
            for (int a0 = 0; a0 < nc; a0 += 16) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                const int mi = m0 + li, aj = a0 + li;
                if (PACKED) {
                    // (ncmax <= 32: eight K steps, unrolled with clamped addresses and zeroed operands past the block, so that the
                    // sixteen LDS reads are in flight together; the rolled loop waited for two dependent reads per MFMA)
                    float av[8], bv[8];
#pragma unroll
                    for (int s8 = 0; s8 < 8; s8++) {
                        const int b = 4 * s8 + kq;
                        const bool oka = mi < M && b < nc, okb = aj < nc && b < nc;
                        const float ta = c[oka ? mi * nc + b : 0];
                        const float tb = g[okb ? soap_tri(nc, b < aj ? b : aj, b < aj ? aj : b) : 0];  // G[a][b] = G[b][a]
                        av[s8] = oka ? ta : 0.f;
                        bv[s8] = okb ? tb : 0.f;
                    }
#pragma unroll
                    for (int s8 = 0; s8 < 8; s8++)
                        if (4 * s8 < nc) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s8], bv[s8], acc, 0, 0, 0);
                } else
                for (int b0 = 0; b0 < nc; b0 += 4) {
                    const int b = b0 + kq;
                    const float av = (mi < M && b < nc) ? c[mi * nc + b] : 0.f;
                    const bool ok = aj < nc && b < nc;
                    if (PACKED) {
                        const float bv = ok ? g[soap_tri(nc, b < aj ? b : aj, b < aj ? aj : b)] : 0.f;  // G[a][b] = G[b][a]
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                    } else {
                        const float bv = ok ? g[b * nc + aj] : 0.f, bt = ok ? g[aj * nc + b] : 0.f;  // dF[b][a], dF[a][b]
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bt, acc, 0, 0, 0);
                    }
                }
                if (aj < nc) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int mm = m0 + 4 * kq + r;
                        if (mm < M) out[d.coef_off[l] + mm * nc + aj] = acc[r];
                    }
                }
            }
    }
}


// (The first-generation fused power-spectrum + tail kernels -- k_soap_ps_tail_fwd / _bwd behind pet_config_set("soap_fused", 1):
// features never stored, but 4.3 + 7.0 ms against 0.5 + 0.6 + 0.9 + 0.7 for the separate kernels -- were removed in round 6:
// the packed power spectrum halves the feature traffic at no cost in time.)
static int g_soap_ps_mfma = 1;  // pet_config_set("soap_ps_mfma", 0): the power spectrum (and its adjoint) on the VALU kernels
void set_soap_ps_mfma(int v) { g_soap_ps_mfma = v ? 1 : 0; }

// ---------------------------------------------------------------------------------------------
// a18, species-sorted tiles (pet_config_set("soap_sorted", 1), default): in legacy mode every atom uses ONE of the
// per-species networks, so the stacked GEMM above does n_species times the needed MFMA work. Here the atoms are
// bucketed by network once per forward (counting sort, k_sp_*), a 64-atom tile holds atoms of one network only and
// multiplies with that network's [32 x Kp] weight: one 32-column tile, the two column-half waves split K instead
// and their accumulators are summed through LDS. Per-atom results do not depend on tile mates, so the (atomic)
// order inside a bucket does not show in the output.
// ---------------------------------------------------------------------------------------------
constexpr int SP_MAXSETS = 8;
struct SpInfo {  // device-side bucket table
    int count[SP_MAXSETS], cursor[SP_MAXSETS], offs[SP_MAXSETS + 1], tstart[SP_MAXSETS + 1];
};
__global__ void k_sp_count(const int* __restrict__ sp, int legacy, int N, SpInfo* __restrict__ info) {
    __shared__ int hist[SP_MAXSETS];
    if (threadIdx.x < SP_MAXSETS) hist[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) atomicAdd(&hist[legacy ? sp[i] : 0], 1);
    __syncthreads();
    if (threadIdx.x < SP_MAXSETS && hist[threadIdx.x]) atomicAdd(&info->count[threadIdx.x], hist[threadIdx.x]);
}
__global__ void k_sp_scan(int n_sets, SpInfo* __restrict__ info) {
    info->offs[0] = 0;
    info->tstart[0] = 0;
    for (int s = 0; s < n_sets; s++) {
        info->offs[s + 1] = info->offs[s] + info->count[s];
        info->tstart[s + 1] = info->tstart[s] + (info->count[s] + BM - 1) / BM;
        info->cursor[s] = 0;
    }
}
__global__ __launch_bounds__(1024) void k_sp_fill(const int* __restrict__ sp, int legacy, int N, SpInfo* __restrict__ info,
                                                   int* __restrict__ perm) {
    // one atomic per (workgroup of 1 024 atoms, network) instead of one per atom: 100 k atoms on 4 cursors serialised in
    // L2 (0.63 ms per atom, 0.08 ms per wave)
    __shared__ int wcount[SP_MAXSETS][16], wbase[SP_MAXSETS][16];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = i < N ? (legacy ? sp[i] : 0) : -1;
    int rank = 0;
    for (int t = 0; t < SP_MAXSETS; t++) {
        const unsigned long long m = __ballot(s == t);
        if (lane == 0) wcount[t][wave] = __popcll(m);
        if (s == t) rank = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < SP_MAXSETS) {
        const int t = threadIdx.x;
        int tot = 0;
        for (int w = 0; w < 16; w++) { wbase[t][w] = tot; tot += wcount[t][w]; }
        const int base = tot ? atomicAdd(&info->cursor[t], tot) : 0;
        for (int w = 0; w < 16; w++) wbase[t][w] += base;
    }
    __syncthreads();
    if (s >= 0) perm[info->offs[s] + wbase[s][wave] + rank] = i;
}
// tile -> (network, first slot in perm, number of atoms); false if the tile index is past the last bucket
__device__ __forceinline__ bool sp_tile(const SpInfo* __restrict__ info, int n_sets, int t, int& s, int& base, int& cnt) {
    if (t >= info->tstart[n_sets]) return false;
    s = 0;
    while (t >= info->tstart[s + 1]) s++;
    base = info->offs[s] + BM * (t - info->tstart[s]);
    cnt = min(BM, info->offs[s + 1] - base);
    return true;
}

__global__ __launch_bounds__(NTHREADS) void k_soap_tail_fwd_set(SoapDims d, const float* __restrict__ feats,
                                                                const int* __restrict__ perm,
                                                                const SpInfo* __restrict__ info, int n_sets,
                                                                const SoapSet* __restrict__ sets,
                                                                const float4* __restrict__ Wps, int Kp,
                                                                const float* __restrict__ rs,
                                                                const float* __restrict__ bs, float* __restrict__ tail,
                                                                float* __restrict__ atomic) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 32, LDO = H + 1, LDA = lds_ld(128), TS = 2 + 2 * H;
    float* As = smem;                    // [64][132]; later the K-split partial sums [64][33] + out [64][33]
    float* a1s = smem + BM * LDA;        // [64][32] silu(a1)
    int* rows = reinterpret_cast<int*>(a1s + BM * H);  // [64] atom of each row, -1 beyond the bucket
    int sidx, base, cnt;
    if (!sp_tile(info, n_sets, blockIdx.x, sidx, base, cnt)) return;
    const WaveId w;
    if (threadIdx.x < BM) rows[threadIdx.x] = (int)threadIdx.x < cnt ? perm[base + threadIdx.x] : -1;
    const float4* Wp = Wps + (size_t)sidx * (Kp / 8) * 64;
    f32x16 acc[1];
    acc_fill_bias<1>(acc, nullptr, 0, w.lane);
    // this thread's 8 float4 of a chunk: rows r0 + 8 q (q = 0..7), column group c; the next chunk is requested
    // before the MFMAs of the current one, so its HBM latency hides behind them and the barriers
    __syncthreads();
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    // the GEMM runs on CENTRED features (x - mean): a1 = rstd W~(x - mu) + W beta without the large, cancelling
    // terms W~x and mu rowsum(W~) of the stacked kernel
    int64_t rowoff[8];
    float rowmu[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int at = rows[r0 + 8 * q];
        rowoff[q] = at >= 0 ? (int64_t)at * d.S : -1;
        rowmu[q] = at >= 0 && d.layernorm ? tail[(size_t)at * TS] : 0.f;
    }
    float4 pre[8];
    auto fetch = [&](int kc) {
        const int col = 128 * kc + 4 * c;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            // raw values: the centring waits until the tile is written (a subtraction here makes the load's latency part of
            // this trip instead of hiding it behind the MFMAs)
            pre[q] = make_float4(rowmu[q], rowmu[q], rowmu[q], rowmu[q]);  // (x - mu = 0 in the padding)
            if (rowoff[q] >= 0 && col < d.S) pre[q] = *reinterpret_cast<const float4*>(feats + rowoff[q] + col);
        }
    };
    fetch(0);
    // The 4 544-term dot products are summed in fp32 only WITHIN a chunk (32 MFMA steps); the 36 chunk results are
    // added in fp64. A single fp32 accumulator over all 1 152 steps leaves a1 with ~3e-6 of rounding noise, which the
    // reverse pass turns into the same relative error of dE/dR (silu'(a1) and the LayerNorm adjoint's mean(dx xhat),
    // which is rebuilt from the saved a1): the worst force component of a 10 000-atom box then sat at 1.05e-5.
    double tot[16];
#pragma unroll
    for (int r = 0; r < 16; r++) tot[r] = 0.0;
    for (int kc = 0; kc < Kp / 128; kc++) {
        // this wave's eight weight fragments of the chunk (the column-half index splits K: 64 of the chunk's 128 k) are
        // requested together, ahead of the barriers: gemm_acc asks for one fragment per four MFMAs, i.e. eight dependent L2
        // round trips per chunk, which is what this kernel waited for (20 k cycles per chunk against 2 k of MFMA)
        float4 bw[8];
        {
            const float4* bp = Wp + ((size_t)16 * kc + 8 * w.ch) * 64 + w.lane;
#pragma unroll
            for (int kg = 0; kg < 8; kg++) bw[kg] = bp[kg * 64];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++)
            *reinterpret_cast<float4*>(As + (r0 + 8 * q) * LDA + 4 * c) =
                make_float4(pre[q].x - rowmu[q], pre[q].y - rowmu[q], pre[q].z - rowmu[q], pre[q].w - rowmu[q]);
        __syncthreads();
        if (kc + 1 < Kp / 128) fetch(kc + 1);
        acc_fill_bias<1>(acc, nullptr, 0, w.lane);
        {
            const float* arow = As + (w.rb * 32 + (w.lane & 31)) * LDA + 64 * w.ch + (w.lane >> 5) * 4;
#pragma unroll
            for (int kg = 0; kg < 8; kg++) {  // the same products in the same order as gemm_acc<64, 1>
                const float4 av = *reinterpret_cast<const float4*>(arow + kg * 8);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bw[kg].x, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bw[kg].y, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bw[kg].z, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bw[kg].w, acc[0], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) tot[r] += (double)acc[0][r];
    }
    __syncthreads();
    double* part = reinterpret_cast<double*>(smem);                  // [64][33] partial sums of the second K half
    float* out = smem + 2 * BM * LDO;                                 // [64][33]
    if (w.ch == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) part[(w.rb * 32 + acc_row(r, w.lane)) * LDO + (w.lane & 31)] = tot[r];
    }
    __syncthreads();
    if (w.ch == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int idx = (w.rb * 32 + acc_row(r, w.lane)) * LDO + (w.lane & 31);
            out[idx] = (float)(tot[r] + part[idx]);
        }
    }
    __syncthreads();
    const SoapSet W = sets[sidx];
    for (int item = threadIdx.x; item < BM * H; item += NTHREADS) {
        const int r = item >> 5, j = item & 31, at = rows[r];
        float a1 = 0.f;
        if (at >= 0) {
            const float rstd = d.layernorm ? tail[(size_t)at * TS + 1] : 1.f;
            a1 = rstd * out[r * LDO + j] + bs[sidx * H + j];
            tail[(size_t)at * TS + 2 + j] = a1;
        }
        a1s[r * H + j] = silu(a1);
    }
    __syncthreads();
    for (int item = threadIdx.x; item < BM * H; item += NTHREADS) {
        const int r = item >> 5, j = item & 31, at = rows[r];
        float e = 0.f;
        if (at >= 0) {
            if (d.NH > 1) {
                float a2 = 0.f;
                for (int q = 0; q < H; q++) a2 += W.W2[j * H + q] * a1s[r * H + q];
                tail[(size_t)at * TS + 2 + H + j] = a2;
                e = W.w3[j] * silu(a2);
            } else {
                e = W.w3[j] * a1s[r * H + j];
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) e += __shfl_xor(e, o);
        if (j == 0 && at >= 0) atomic[at] = e;
    }
}

__global__ __launch_bounds__(NTHREADS) void k_soap_tail_bwd_set(SoapDims d, const float* __restrict__ feats,
                                                                const int* __restrict__ perm,
                                                                const SpInfo* __restrict__ info, int n_sets,
                                                                const int* __restrict__ sp,
                                                                const SoapSet* __restrict__ sets,
                                                                const float4* __restrict__ Wpbs, int Kp,
                                                                const float* __restrict__ rs,
                                                                const float* __restrict__ bs,
                                                                const float* __restrict__ enc,
                                                                const float* __restrict__ tail,
                                                                const float* __restrict__ gA, float* __restrict__ dF,
                                                                const float* __restrict__ da2x, int packed) {
    // packed != 0: feats / dF rows are the upper-triangle layout (SoapDims::Sp floats), Wpbs the packed weights with the
    // diagonal doubled; the output is G[a][b] = dx[a][b] + dx[b][a] = rstd (g'_t - 2 (m1 + xhat_t m2)): the two LayerNorm means
    // (over the FULL S-vector, as before) enter twice
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = 32, LDD = lds_ld(H), TS = 2 + 2 * H;
    const int SR = packed ? d.Sp : d.S;   // row pitch and length of feats / dF
    const float pm = packed ? 2.f : 1.f;
    float* Ds = smem;                 // [64][36] d a1
    float* d2 = Ds + BM * LDD;        // [64][32] d a2
    float* st = d2 + BM * H;          // [64][4] mean, rstd, m1, m2
    float* ot = st + BM * 4;          // [64][132] output tile staging
    int* rows = reinterpret_cast<int*>(ot + BM * lds_ld(128));  // [64]
    int sidx, base, cnt;
    if (!sp_tile(info, n_sets, blockIdx.x, sidx, base, cnt)) return;
    const WaveId w;
    if (threadIdx.x < BM) rows[threadIdx.x] = (int)threadIdx.x < cnt ? perm[base + threadIdx.x] : -1;
    const SoapSet W = sets[sidx];
    const float4* Wpb = Wpbs + (size_t)sidx * (Kp / 32) * 4 * 64;
    __syncthreads();
    for (int item = threadIdx.x; item < BM * H; item += NTHREADS) {
        const int r = item >> 5, j = item & 31, at = rows[r];
        float v = 0.f;
        if (at >= 0 && d.NH > 1)
            v = da2x ? da2x[(size_t)at * H + j] : gA[at] * W.w3[j] * dsilu(tail[(size_t)at * TS + 2 + H + j]);
        d2[r * H + j] = v;
    }
    __syncthreads();
    for (int item = threadIdx.x; item < BM * H; item += NTHREADS) {
        const int r = item >> 5, j = item & 31, at = rows[r];
        float da1 = 0.f, t1 = 0.f, t2 = 0.f;
        float mu = 0.f, rstd = 1.f;
        if (at >= 0) {
            const float a1 = tail[(size_t)at * TS + 2 + j];
            if (d.NH > 1) {
                float acc = 0.f;
                for (int q = 0; q < H; q++) acc += W.W2[q * H + j] * d2[r * H + q];
                da1 = acc * dsilu(a1);
            } else {
                da1 = gA[at] * W.w3[j] * dsilu(a1);
            }
            if (d.layernorm) {
                mu = tail[(size_t)at * TS];
                rstd = tail[(size_t)at * TS + 1];
                t1 = da1 * rs[sidx * H + j];
                t2 = da1 * (a1 - bs[sidx * H + j]) / rstd;  // da1 * Wall_s[j] . (x - mu)
            }
        }
        Ds[r * LDD + j] = da1;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
        if (j == 0) {
            st[r * 4] = mu; st[r * 4 + 1] = rstd;
            st[r * 4 + 2] = pm * t1 / d.S;          // m1 = mean(dxn gamma)
            st[r * 4 + 3] = pm * rstd * t2 / d.S;   // m2 = mean(dxn gamma xhat)
        }
    }
    __syncthreads();
    // the feature values the LayerNorm adjoint needs are requested one block ahead (8 float4 per thread)
    const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    int64_t rowoff[8];
#pragma unroll
    for (int q = 0; q < 8; q++) rowoff[q] = rows[r0 + 8 * q] >= 0 ? (int64_t)rows[r0 + 8 * q] * SR : -1;
    float4 xpre[8];
    auto fetch = [&](int nblk) {
        const int k = 128 * nblk + 4 * c4;
#pragma unroll
        for (int q = 0; q < 8; q++)
            xpre[q] = d.layernorm && rowoff[q] >= 0 && k < SR ? *reinterpret_cast<const float4*>(feats + rowoff[q] + k)
                                                             : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    fetch(0);
    for (int nblk = 0; nblk < Kp / 128; nblk++) {
        f32x16 acc[2];
        acc_fill_bias<2>(acc, nullptr, 0, w.lane);
        gemm_acc<H, 2>(Ds + w.rb * 32 * LDD, LDD, Wpb, H / 8, 0, 4 * nblk + 2 * w.ch, acc, w.lane);
        __syncthreads();  // the previous block's staging tile has been consumed
        acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) { ot[r * lds_ld(128) + c] = v; });
        __syncthreads();
        float4 xcur[8];
#pragma unroll
        for (int q = 0; q < 8; q++) xcur[q] = xpre[q];
        if (nblk + 1 < Kp / 128) fetch(nblk + 1);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int r = r0 + 8 * q, k = 128 * nblk + 4 * c4;
            const int at = rows[r];
            if (at < 0 || k >= SR) continue;
            float4 v = *reinterpret_cast<const float4*>(ot + r * lds_ld(128) + 4 * c4);
            if (d.layernorm) {
                const float4 x = xcur[q];
                const float mu = st[r * 4], rs_ = st[r * 4 + 1], m1 = st[r * 4 + 2], m2 = st[r * 4 + 3];
                v.x = rs_ * (v.x - m1 - (x.x - mu) * rs_ * m2);
                v.y = rs_ * (v.y - m1 - (x.y - mu) * rs_ * m2);
                v.z = rs_ * (v.z - m1 - (x.z - mu) * rs_ * m2);
                v.w = rs_ * (v.w - m1 - (x.w - mu) * rs_ * m2);
            }
            if (enc) {
                const float4 e = *reinterpret_cast<const float4*>(enc + (size_t)sp[at] * d.S + k);
                v.x *= e.x; v.y *= e.y; v.z *= e.z; v.w *= e.w;
            }
            *reinterpret_cast<float4*>(dF + (size_t)at * SR + k) = v;
        }
    }
}

// reverse of the tail: dF[i][k] = d e_i / d ps_i[k] * gA[i]
__global__ __launch_bounds__(256) void k_soap_tail_bwd(SoapDims d, const float* __restrict__ feats,
                                                       const int* __restrict__ sp, const SoapSet* __restrict__ sets,
                                                       const float* __restrict__ enc, const float* __restrict__ tail,
                                                       const float* __restrict__ gA, float* __restrict__ dF,
                                                       const float* __restrict__ da2x) {
    __shared__ float da1[MAXH], da2[MAXH], red[8];
    const int i = blockIdx.x, tid = threadIdx.x;
    const SoapSet W = sets[d.legacy ? sp[i] : 0];
    const float* tl = tail + (size_t)i * (2 + 2 * d.H);
    const float mean = tl[0], rstd = tl[1], g = gA[i];
    if (tid < d.H) {
        if (d.NH > 1) da2[tid] = da2x ? da2x[(size_t)i * d.H + tid] : g * W.w3[tid] * dsilu(tl[2 + d.H + tid]);
        else da1[tid] = g * W.w3[tid] * dsilu(tl[2 + tid]);
    }
    __syncthreads();
    if (tid < d.H && d.NH > 1) {
        float s = 0.f;
        for (int j = 0; j < d.H; j++) s += W.W2[j * d.H + tid] * da2[j];
        da1[tid] = s * dsilu(tl[2 + tid]);
    }
    __syncthreads();
    const float* x = feats + (size_t)i * d.S;
    // pass 1: dxn gamma and the two LayerNorm means
    float s1 = 0.f, s2 = 0.f;
    for (int k = tid; k < d.S; k += 256) {
        float dx = 0.f;
        for (int q = 0; q < d.H; q++) dx += W.W1[(size_t)q * d.S + k] * da1[q];
        if (d.layernorm) {
            dx *= W.ln_w[k];
            s1 += dx;
            s2 += dx * (x[k] - mean) * rstd;
        }
        dF[(size_t)i * d.S + k] = dx;
    }
    if (d.layernorm) {
        const float m1 = block_sum(s1, red) / d.S;
        const float m2 = block_sum(s2, red) / d.S;
        for (int k = tid; k < d.S; k += 256) {
            const float xh = (x[k] - mean) * rstd;
            dF[(size_t)i * d.S + k] = rstd * (dF[(size_t)i * d.S + k] - m1 - xh * m2);
        }
    }
    if (enc) {
        const float* e = enc + (size_t)sp[i] * d.S;
        for (int k = tid; k < d.S; k += 256) dF[(size_t)i * d.S + k] *= e[k];
    }
}

// Hidden layers beyond the second (num_hidden_layers > 2; the reference's MLPMap stacks Linear + SiLU per layer,
// soap_bpnn/model.py). Every tail kernel above keeps a1, a2 and the 4 544-wide first Linear, which is all of the work; one
// wave per atom continues from a2: a_{k+3} = Wh[k+1] silu(a_{k+2}) -> ext[atom][k][H], and replaces the atom's energy.
__global__ __launch_bounds__(256) void k_soap_tail_extra_fwd(SoapDims d, const int* __restrict__ sp,
                                                             const SoapSet* __restrict__ sets,
                                                             const float* __restrict__ tail, float* __restrict__ ext,
                                                             float* __restrict__ atomic, int N) {
    __shared__ float hs[4][MAXH];
    const int wave = threadIdx.x >> 6, j = threadIdx.x & 63;
    const int at = blockIdx.x * 4 + wave;
    if (at >= N) return;  // whole waves leave; no workgroup barrier below
    const int H = d.H, TS = 2 + 2 * H, NX = d.NH - 2;
    const SoapSet* W = sets + (d.legacy ? sp[at] : 0);
    float a = j < H ? tail[(size_t)at * TS + 2 + H + j] : 0.f;
    for (int k = 0; k < NX; k++) {
        if (j < H) hs[wave][j] = silu(a);
        __builtin_amdgcn_wave_barrier();
        float s = 0.f;
        if (j < H) {
            const float* w = W->Wh[k + 1] + (size_t)j * H;
            for (int q = 0; q < H; q++) s += w[q] * hs[wave][q];
            ext[((size_t)at * NX + k) * H + j] = s;
        }
        a = s;
        __builtin_amdgcn_wave_barrier();
    }
    float e = j < H ? W->w3[j] * silu(a) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    if (j == 0) atomic[at] = e;
}

// ... and its reverse down to d a2 [N, H], which the tail adjoints then start from instead of gA w3 silu'(a2)
__global__ __launch_bounds__(256) void k_soap_tail_extra_bwd(SoapDims d, const int* __restrict__ sp,
                                                             const SoapSet* __restrict__ sets,
                                                             const float* __restrict__ tail,
                                                             const float* __restrict__ ext, const float* __restrict__ gA,
                                                             float* __restrict__ da2, int N) {
    __shared__ float ds[4][MAXH];
    const int wave = threadIdx.x >> 6, j = threadIdx.x & 63;
    const int at = blockIdx.x * 4 + wave;
    if (at >= N) return;
    const int H = d.H, TS = 2 + 2 * H, NX = d.NH - 2;
    const SoapSet* W = sets + (d.legacy ? sp[at] : 0);
    float da = j < H ? gA[at] * W->w3[j] * dsilu(ext[((size_t)at * NX + NX - 1) * H + j]) : 0.f;
    for (int k = NX - 1; k >= 0; k--) {  // through a_{k+3} = Wh[k+1] silu(a_{k+2})
        if (j < H) ds[wave][j] = da;
        __builtin_amdgcn_wave_barrier();
        float s = 0.f;
        if (j < H) {
            const float* w = W->Wh[k + 1];
            for (int q = 0; q < H; q++) s += w[(size_t)q * H + j] * ds[wave][q];
            const float aprev = k > 0 ? ext[((size_t)at * NX + k - 1) * H + j] : tail[(size_t)at * TS + 2 + H + j];
            s *= dsilu(aprev);
        }
        da = s;
        __builtin_amdgcn_wave_barrier();
    }
    if (j < H) da2[(size_t)at * H + j] = da;
}

// dC[l][m][a] = sum_b (dF[l][a][b] + dF[l][b][a]) c[l][m][b]
__global__ __launch_bounds__(256) void k_soap_ps_bwd(SoapDims d, const float* __restrict__ Cf,
                                                     const float* __restrict__ dF, const int* __restrict__ lut,
                                                     float* __restrict__ dCf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < d.NCOEF; k += 256) smem[k] = Cf[(size_t)i * d.NCOEF + k];
    __syncthreads();
    const float* dFi = dF + (size_t)i * d.S;
    for (int idx = threadIdx.x; idx < d.NCOEF; idx += 256) {
        int l = 0;
        while (idx >= d.coef_off[l + 1]) l++;
        const int nc = d.n_per_l[l] * d.C, q = idx - d.coef_off[l];
        const int m = q / nc, a = q % nc;
        const float* crow = smem + d.coef_off[l] + m * nc;
        const float* fb = dFi + d.feat_off[l];
        float s = 0.f;
        for (int b = 0; b < nc; b++) s += (fb[a * nc + b] + fb[b * nc + a]) * crow[b];
        dCf[(size_t)i * d.NCOEF + idx] = s;
    }
}

// reverse of the expansion: dv[p] = d/d(edge vector) for every pair of the row
__global__ __launch_bounds__(256) void k_soap_expand_bwd(SoapDims d, const float4* __restrict__ geo,
                                                         const int* __restrict__ rowptr,
                                                         const int* __restrict__ sp_nbr,
                                                         const float* __restrict__ table,
                                                         const float* __restrict__ shn, const int* __restrict__ ilut,
                                                         const float* __restrict__ spw,
                                                         const float* __restrict__ dCf, float4* __restrict__ dv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dCs = smem;                       // [NCOEF]
    float* Ys = dCs + d.NCOEF;               // [PC][NLM] and three gradient components w.r.t. the edge vector
    float* Gx = Ys + PC * d.NLM;
    float* Gy = Gx + PC * d.NLM;
    float* Gz = Gy + PC * d.NLM;
    float* Rs = Gz + PC * d.NLM;             // [PC][F] R fc, d(R fc)/dr
    float* dRs = Rs + PC * d.F;
    float* us = dRs + PC * d.F;              // [PC][8] unit vector, r, 1/r, fc, dfc
    int* sps = reinterpret_cast<int*>(us + PC * 8);
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < d.NCOEF; k += 256) dCs[k] = dCf[(size_t)i * d.NCOEF + k];
    const int p0 = rowptr[i], p1 = rowptr[i + 1];
    for (int base = p0; base < p1; base += PC) {
        const int npc = min(PC, p1 - base);
        __syncthreads();
        if (tid < npc) {
            const float4 g = geo[base + tid];
            const float r = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
            const float ir = r > 0.f ? 1.0f / r : 0.f;
            float* u = us + tid * 8;
            u[0] = g.x * ir; u[1] = g.y * ir; u[2] = g.z * ir; u[3] = r; u[4] = ir;
            float dfc;
            u[5] = shifted_cosine(r, d.rc, d.width, &dfc);
            u[6] = dfc;
            sps[tid] = sp_nbr[base + tid];
        }
        __syncthreads();
        for (int idx = tid; idx < npc * (d.L + 1); idx += 256) {
            const int pp = idx / (d.L + 1), mm = idx % (d.L + 1);
            const float* u = us + pp * 8;
            sh_chain(u[0], u[1], u[2], mm, d.L, shn, Ys + pp * d.NLM, Gx + pp * d.NLM, Gy + pp * d.NLM, Gz + pp * d.NLM,
                     u[4]);
        }
        for (int idx = tid; idx < npc * d.F; idx += 256) {
            const int pp = idx / d.F, f = idx % d.F;
            const float* u = us + pp * 8;
            radial_one(d, table, f, u[3], u[5], u[6], Rs + pp * d.F + f, dRs + pp * d.F + f);
        }
        __syncthreads();
        for (int pp = wave; pp < npc; pp += 4) {
            const float* w = spw + sps[pp] * d.C;
            const float ux = us[pp * 8], uy = us[pp * 8 + 1], uz = us[pp * 8 + 2];
            float ax = 0.f, ay = 0.f, az = 0.f;
            for (int it = lane; it < d.ITEMS; it += 64) {
                const int code = ilut[it];
                const int lm = code & 255, rn = (code >> 8) & 255, cb = code >> 16;
                float A = 0.f;
                for (int a = 0; a < d.C; a++) A += dCs[cb + a] * w[a];
                const float y = Ys[pp * d.NLM + lm], rf = Rs[pp * d.F + rn], drf = dRs[pp * d.F + rn];
                const float radial = A * drf * y;   // along the unit vector
                const float ang = A * rf;
                ax += radial * ux + ang * Gx[pp * d.NLM + lm];
                ay += radial * uy + ang * Gy[pp * d.NLM + lm];
                az += radial * uz + ang * Gz[pp * d.NLM + lm];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ax += __shfl_xor(ax, o); ay += __shfl_xor(ay, o); az += __shfl_xor(az, o);
            }
            if (lane == 0) dv[base + pp] = make_float4(ax, ay, az, 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// a17, second generation: no workgroup barriers, every lane busy.
//   k_soap_expand_w      one WAVE per atom: lanes 0..31 run the spherical-harmonic levels of one pair each, lanes
//                        32..63 that pair's radial splines, into wave-private LDS; then every lane owns up to
//                        MAXI (lm, n) items and sums them over the pairs with the C = 4 species channels as one
//                        float4 accumulator (the first generation looped 1120 scalar coefficients x pairs behind
//                        three __syncthreads, one workgroup per atom, and re-read the species weight from global).
//   k_soap_expand_bwd_p  one LANE per pair, flat over the CSR: the pair's Y_lm levels (with gradients) and splines
//                        stay in registers, dC rows of the centre atom come through L1 (the ~26 pairs of an atom
//                        sit in the same one or two waves), no LDS, no reduction, no barrier.
// Both need C == 4 (float4 channels; Alchemical default and 4-species legacy) and at most 64 * MAXI items;
// otherwise soap_fwd / soap_bwd fall back to the first-generation kernels (pet_config_set("soap_pair", 0) too).
// ---------------------------------------------------------------------------------------------
constexpr int MAXI = 8;
static int g_soap_pair = 1;
void set_soap_pair(int v) { g_soap_pair = v ? 1 : 0; }

template <int LMAX>
struct ShLevels {  // the recurrences of sh_chain, all m-chains advanced one l at a time
    float x, y, z, ir;
    float cm[LMAX + 1], sm[LMAX + 1], q1[LMAX + 1], q2[LMAX + 1], dq1[LMAX + 1], dq2[LMAX + 1];
    __device__ __forceinline__ void init(float x_, float y_, float z_, float ir_) {
        x = x_; y = y_; z = z_; ir = ir_;
        cm[0] = 1.f; sm[0] = 0.f;
        float qmm = 1.f;
#pragma unroll
        for (int m = 0; m <= LMAX; m++) {
            if (m > 0) {
                cm[m] = x * cm[m - 1] - y * sm[m - 1];
                sm[m] = x * sm[m - 1] + y * cm[m - 1];
                qmm *= -(2 * m - 1);
            }
            q1[m] = qmm; q2[m] = 0.f; dq1[m] = 0.f; dq2[m] = 0.f;
        }
    }
    // level l (compile-time after unrolling): Y / G[mi], mi = l + m (cos) and l - m (sin)
    template <bool GRAD>
    __device__ __forceinline__ void level(int l, const float* __restrict__ shn, int L, float* Y, float* Gx, float* Gy,
                                          float* Gz) {
#pragma unroll
        for (int m = 0; m <= LMAX; m++) {
            if (m > l) continue;
            float q, dq;
            if (l == m) { q = q1[m]; dq = 0.f; }
            else if (l == m + 1) { q = (2 * m + 1) * z * q1[m]; dq = (2 * m + 1) * q1[m]; }
            else {
                const float inv = 1.0f / (l - m);
                q = ((2 * l - 1) * z * q1[m] - (l + m - 1) * q2[m]) * inv;
                dq = ((2 * l - 1) * (q1[m] + z * dq1[m]) - (l + m - 1) * dq2[m]) * inv;
            }
            const float f = shn[l * (L + 1) + m];
            if (m == 0) {
                Y[l] = f * q;
                if (GRAD) {
                    const float gz = f * dq, dot = z * gz;
                    Gx[l] = (-x * dot) * ir; Gy[l] = (-y * dot) * ir; Gz[l] = (gz - z * dot) * ir;
                }
            } else {
                const float fq = f * q, fm = fq * m, cp = cm[m - 1], sp = sm[m - 1];
                Y[l + m] = fq * cm[m];
                Y[l - m] = fq * sm[m];
                if (GRAD) {
                    float gx = fm * cp, gy = -fm * sp, gz = f * dq * cm[m];
                    float dot = x * gx + y * gy + z * gz;
                    Gx[l + m] = (gx - x * dot) * ir; Gy[l + m] = (gy - y * dot) * ir; Gz[l + m] = (gz - z * dot) * ir;
                    gx = fm * sp; gy = fm * cp; gz = f * dq * sm[m];
                    dot = x * gx + y * gy + z * gz;
                    Gx[l - m] = (gx - x * dot) * ir; Gy[l - m] = (gy - y * dot) * ir; Gz[l - m] = (gz - z * dot) * ir;
                }
            }
            if (l > m) { q2[m] = q1[m]; dq2[m] = dq1[m]; }
            q1[m] = q; dq1[m] = dq;
            if (l == m) { q2[m] = 0.f; dq2[m] = 0.f; }
        }
    }
};

template <int LMAX>
__global__ __launch_bounds__(256) void k_soap_expand_w(SoapDims d, const float4* __restrict__ geo,
                                                       const int* __restrict__ rowptr, const int* __restrict__ sp_nbr,
                                                       const float* __restrict__ table, const float* __restrict__ shn,
                                                       const int* __restrict__ ilut, const float* __restrict__ spw,
                                                       float* __restrict__ Cf, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const int ld = ((d.NLM + d.F + 4 + 3) & ~3) + 4;   // row: Y[NLM] | R fc [F] | species weights [4] | r, fc (16 B aligned)
    float* rows = smem + (size_t)wave * 32 * ld;
    const int wo = ld - 8;
    const int p0 = rowptr[i], p1 = rowptr[i + 1];
    float4 acc[MAXI];
    int code[MAXI];
#pragma unroll
    for (int k = 0; k < MAXI; k++) {
        acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int it = lane + 64 * k;
        code[k] = it < d.ITEMS ? ilut[it] : -1;
    }
    for (int base = p0; base < p1; base += 32) {
        const int npc = min(32, p1 - base);
        const int pp = lane & 31;
        if (pp < npc) {
            const float4 g = geo[base + pp];
            const float r = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
            const float ir = r > 0.f ? 1.0f / r : 0.f;
            float* row = rows + pp * ld;
            if (lane < 32) {
                ShLevels<LMAX> sh;
                sh.init(g.x * ir, g.y * ir, g.z * ir, ir);
#pragma unroll
                for (int l = 0; l <= LMAX; l++) {
                    if (l > d.L) break;
                    float Y[2 * LMAX + 1];
                    sh.template level<false>(l, shn, d.L, Y, nullptr, nullptr, nullptr);
#pragma unroll
                    for (int mi = 0; mi < 2 * LMAX + 1; mi++)
                        if (mi <= 2 * l) row[l * l + mi] = Y[mi];
                }
                *reinterpret_cast<float4*>(row + wo) = *reinterpret_cast<const float4*>(spw + sp_nbr[base + pp] * 4);
            } else {
                row[wo + 4] = r;
                row[wo + 5] = shifted_cosine(r, d.rc, d.width, nullptr);
            }
        }
        __builtin_amdgcn_wave_barrier();  // same wave wrote the rows; LDS serves a wave's requests in order
        // the radial functions: (pair, function) items over ALL 64 lanes -- the two spline nodes of an item are independent
        // loads, where 32 lanes walking the functions of their own pair waited for each pair of them in turn (and the
        // spherical-harmonic half of the wave sat idle behind the same branch)
        for (int it = lane; it < npc * d.F; it += 64) {
            const int q = it / d.F, f = it - q * d.F;
            float* rq = rows + q * ld;
            radial_one(d, table, f, rq[wo + 4], rq[wo + 5], 0.f, rq + d.NLM + f, nullptr);
        }
        __builtin_amdgcn_wave_barrier();
        for (int q = 0; q < npc; q++) {
            const float* row = rows + q * ld;
            const float4 w4 = *reinterpret_cast<const float4*>(row + wo);
#pragma unroll
            for (int k = 0; k < MAXI; k++) {
                if (code[k] < 0) continue;
                const float yr = row[code[k] & 255] * row[d.NLM + ((code[k] >> 8) & 255)];
                acc[k].x += yr * w4.x; acc[k].y += yr * w4.y; acc[k].z += yr * w4.z; acc[k].w += yr * w4.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < MAXI; k++)
        if (code[k] >= 0) *reinterpret_cast<float4*>(Cf + (size_t)i * d.NCOEF + (code[k] >> 16)) = acc[k];
}

template <int LMAX>
__global__ __launch_bounds__(256) void k_soap_expand_bwd_p(SoapDims d, const float4* __restrict__ geo,
                                                           const int* __restrict__ ctr, const int* __restrict__ sp_nbr,
                                                           const float* __restrict__ table,
                                                           const float* __restrict__ shn, const float* __restrict__ spw,
                                                           const float* __restrict__ dCf, float4* __restrict__ dv,
                                                           int64_t E, int lds_floats) {
    // The 256 pairs of a workgroup are consecutive in CSR order, i.e. they belong to ~10 consecutive centres: those
    // centres' adjoint coefficient rows (NCOEF floats each) are staged in LDS once, and the 343 float4 a pair reads come
    // from there (lanes of one centre read the same address: a broadcast) instead of as many dependent L1 round trips.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t p0 = (int64_t)blockIdx.x * 256;
    const int64_t plast = (p0 + 255 < E ? p0 + 255 : E - 1);
    const int c0 = ctr[p0], c1 = ctr[plast];
    const bool staged = (int64_t)(c1 - c0 + 1) * d.NCOEF <= lds_floats;
    if (staged) {
        const float4* src = reinterpret_cast<const float4*>(dCf + (size_t)c0 * d.NCOEF);
        const int n4 = (c1 - c0 + 1) * (d.NCOEF / 4);
        for (int k = threadIdx.x; k < n4; k += 256) reinterpret_cast<float4*>(smem)[k] = src[k];
        __syncthreads();
    }
    const int64_t p = p0 + threadIdx.x;
    if (p >= E) return;
    const float4 g = geo[p];
    const float r = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
    const float ir = r > 0.f ? 1.0f / r : 0.f;
    const float ux = g.x * ir, uy = g.y * ir, uz = g.z * ir;
    float dfc;
    const float fc = shifted_cosine(r, d.rc, d.width, &dfc);
    const float4 w4 = *reinterpret_cast<const float4*>(spw + sp_nbr[p] * 4);
    const float* dC = staged ? smem + (size_t)(ctr[p] - c0) * d.NCOEF : dCf + (size_t)ctr[p] * d.NCOEF;
    ShLevels<LMAX> sh;
    sh.init(ux, uy, uz, ir);
    double ax = 0.0, ay = 0.0, az = 0.0, along = 0.0;  // `along` multiplies the unit vector; fp64 sums (280 items)
#pragma unroll
    for (int l = 0; l <= LMAX; l++) {
        if (l > d.L) break;
        float Y[2 * LMAX + 1], Gx[2 * LMAX + 1], Gy[2 * LMAX + 1], Gz[2 * LMAX + 1], T[2 * LMAX + 1];
        sh.template level<true>(l, shn, d.L, Y, Gx, Gy, Gz);
#pragma unroll
        for (int mi = 0; mi < 2 * LMAX + 1; mi++) T[mi] = 0.f;
        const int nl = d.n_per_l[l];
        const float* dCl = dC + d.coef_off[l];
        for (int n = 0; n < nl; n++) {
            float R, dR;
            // (requesting the spline nodes of function n + 1 before evaluating function n was measured in round 6: 1.04 - 1.08
            // against 0.97 - 0.98 ms; three waves per SIMD at 168 registers and 52 KB of staging: 1.07 ms)
            radial_one(d, table, d.rad_off[l] + n, r, fc, dfc, &R, &dR);
            float ay_ = 0.f;  // sum over m of A Y in fp32 (at most 2 l + 1 terms), then one fp64 add per radial function
#pragma unroll
            for (int mi = 0; mi < 2 * LMAX + 1; mi++) {
                if (mi > 2 * l) continue;
                const float4 c = *reinterpret_cast<const float4*>(dCl + (size_t)(mi * nl + n) * 4);
                const float A = c.x * w4.x + c.y * w4.y + c.z * w4.z + c.w * w4.w;
                ay_ = fmaf(A, Y[mi], ay_);
                T[mi] += A * R;
            }
            along += (double)(ay_ * dR);
        }
#pragma unroll
        for (int mi = 0; mi < 2 * LMAX + 1; mi++) {
            if (mi > 2 * l) continue;
            ax += (double)(T[mi] * Gx[mi]); ay += (double)(T[mi] * Gy[mi]); az += (double)(T[mi] * Gz[mi]);
        }
    }
    dv[p] = make_float4((float)(ax + along * ux), (float)(ay + along * uy), (float)(az + along * uz), 0.f);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct SoapWs {
    float *Cf, *feats, *tail, *dF, *dCf, *dv;
    float *tail_ext, *da2;  // num_hidden_layers > 2: a3 .. a_NH [N][NH - 2][H] and the adjoint of a2 [N][H]
    int* perm;     // [N] atoms bucketed by network (species-sorted tail tiles)
    SpInfo* info;
    size_t bytes;
};
static void carve_soap(const SoapDims& d, int64_t N, int64_t E, void* base, SoapWs& w) {
    Carver c(base);
    const int64_t Na = N > 0 ? N : 1, Ea = E > 0 ? E : 1;
    w.Cf = c.take<float>(Na * d.NCOEF);
    w.feats = c.take<float>(Na * d.S);
    w.tail = c.take<float>(Na * (2 + 2 * d.H));
    w.dF = c.take<float>(Na * d.S);
    w.dCf = c.take<float>(Na * d.NCOEF);
    w.dv = c.take<float>(Ea * 4);
    w.perm = c.take<int>(Na);
    w.info = reinterpret_cast<SpInfo*>(c.take<int>((sizeof(SpInfo) + 3) / 4));
    w.tail_ext = c.take<float>(d.NH > 2 ? Na * (d.NH - 2) * d.H : 1);
    w.da2 = c.take<float>(d.NH > 2 ? Na * d.H : 1);
    w.bytes = c.off;
}

static int soap_get(const SoapModel& m, const std::string& key, int64_t numel, const float** out) {
    auto it = m.raw.find(key);
    PET_REQUIRE(it != m.raw.end(), PET_ERR_ARGUMENT, "missing parameter '" + key + "'");
    PET_REQUIRE(it->second.second == numel, PET_ERR_ARGUMENT, "parameter '" + key + "' has the wrong size");
    *out = it->second.first;
    return PET_OK;
}

static int soap_finalize(SoapModel& m, hipStream_t st) {
    const SoapDims& d = m.d;
    PET_REQUIRE(m.table != nullptr, PET_ERR_ARGUMENT, "soap_model_set_radial_table has not been called");
    int rc;
    // normalisation of the real spherical harmonics: sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!) [* sqrt 2]
    std::vector<float> shn((d.L + 1) * (d.L + 1), 0.f);
    for (int l = 0; l <= d.L; l++)
        for (int mm = 0; mm <= l; mm++) {
            double f = (2 * l + 1) / (4.0 * M_PI);
            for (int k = l - mm + 1; k <= l + mm; k++) f /= k;
            shn[l * (d.L + 1) + mm] = (float)(sqrt(f) * (mm > 0 ? sqrt(2.0) : 1.0));
        }
    std::vector<int> clut(d.NCOEF), ilut(d.ITEMS);
    int idx = 0, it = 0;
    for (int l = 0; l <= d.L; l++)
        for (int mi = 0; mi < 2 * l + 1; mi++)
            for (int n = 0; n < d.n_per_l[l]; n++) {
                const int lm = l * l + mi, rn = d.rad_off[l] + n;
                ilut[it++] = lm | (rn << 8) | (idx << 16);
                for (int a = 0; a < d.C; a++) clut[idx++] = lm | (rn << 8) | (a << 16);
            }
    std::vector<int2> flut(d.S), olut(d.NCOEF);
    for (int l = 0; l <= d.L; l++) {
        const int nc = d.n_per_l[l] * d.C, M = 2 * l + 1;
        for (int a = 0; a < nc; a++)
            for (int b = 0; b < nc; b++)
                flut[d.feat_off[l] + a * nc + b] = make_int2(d.coef_off[l] + a, (d.coef_off[l] + b) | (M << 16) | (nc << 24));
        for (int mi = 0; mi < M; mi++)
            for (int a = 0; a < nc; a++)
                olut[d.coef_off[l] + mi * nc + a] = make_int2(d.feat_off[l] + a, (d.coef_off[l] + mi * nc) | (nc << 16));
    }
    if (!m.shnorm) {
        if ((rc = salloc(m, (void**)&m.feat_lut, flut.size() * sizeof(int2)))) return rc;
        if ((rc = salloc(m, (void**)&m.out_lut, olut.size() * sizeof(int2)))) return rc;
        if ((rc = salloc(m, (void**)&m.shnorm, shn.size() * 4))) return rc;
        if ((rc = salloc(m, (void**)&m.coef_lut, clut.size() * 4))) return rc;
        if ((rc = salloc(m, (void**)&m.item_lut, ilut.size() * 4))) return rc;
        if ((rc = salloc(m, (void**)&m.species_w, (size_t)d.ns * d.C * 4))) return rc;
        if ((rc = salloc(m, (void**)&m.sets, sizeof(SoapSet) * m.n_sets))) return rc;
    }
    PET_HIP_CHECK(hipMemcpyAsync(m.shnorm, shn.data(), shn.size() * 4, hipMemcpyHostToDevice, st));
    PET_HIP_CHECK(hipMemcpyAsync(m.coef_lut, clut.data(), clut.size() * 4, hipMemcpyHostToDevice, st));
    PET_HIP_CHECK(hipMemcpyAsync(m.item_lut, ilut.data(), ilut.size() * 4, hipMemcpyHostToDevice, st));
    PET_HIP_CHECK(hipMemcpyAsync(m.feat_lut, flut.data(), flut.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    PET_HIP_CHECK(hipMemcpyAsync(m.out_lut, olut.data(), olut.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    if (d.legacy) {
        std::vector<float> eye((size_t)d.ns * d.C, 0.f);
        for (int s = 0; s < d.ns; s++) eye[s * d.C + s] = 1.f;
        PET_HIP_CHECK(hipMemcpyAsync(m.species_w, eye.data(), eye.size() * 4, hipMemcpyHostToDevice, st));
        m.enc = nullptr;
    } else {
        const float* emb;
        if ((rc = soap_get(m, "species_embedding.weight", (int64_t)d.ns * d.C, &emb))) return rc;
        PET_HIP_CHECK(hipMemcpyAsync(m.species_w, emb, (size_t)d.ns * d.C * 4, hipMemcpyDeviceToDevice, st));
        if ((rc = soap_get(m, "center_encoding.weight", (int64_t)d.ns * d.S, &m.enc))) return rc;
    }
    std::vector<SoapSet> sets(m.n_sets);
    for (int s = 0; s < m.n_sets; s++) {
        const std::string ss = std::to_string(s);
        if (d.layernorm) {
            if ((rc = soap_get(m, "layernorm." + ss + ".weight", d.S, &sets[s].ln_w))) return rc;
            if ((rc = soap_get(m, "layernorm." + ss + ".bias", d.S, &sets[s].ln_b))) return rc;
        }
        if ((rc = soap_get(m, "bpnn." + ss + ".0.weight", (int64_t)d.H * d.S, &sets[s].W1))) return rc;
        for (int k = 1; k < d.NH; k++)
            if ((rc = soap_get(m, "bpnn." + ss + "." + std::to_string(2 * k) + ".weight", (int64_t)d.H * d.H,
                               &sets[s].Wh[k - 1])))
                return rc;
        sets[s].W2 = sets[s].Wh[0];
        if ((rc = soap_get(m, "last_layers.energy." + ss + ".weight", d.H, &sets[s].w3))) return rc;
    }
    PET_HIP_CHECK(hipMemcpyAsync(m.sets, sets.data(), sizeof(SoapSet) * m.n_sets, hipMemcpyHostToDevice, st));
    // stacked first Linear for the MFMA tail
    m.NT = 0;
    if (d.H == 32 && m.n_sets <= 8 && d.S % 4 == 0) {
        const int NT = (m.n_sets + 1) / 2, NOUTP = 64 * NT, Kp = (d.S + 127) / 128 * 128;
        if (!m.wall || m.NOUTP != NOUTP || m.Kp != Kp) {
            const size_t n4 = (size_t)NOUTP * Kp / 4;
            if ((rc = salloc(m, (void**)&m.wall, (size_t)NOUTP * Kp * 4))) return rc;
            if ((rc = salloc(m, (void**)&m.wall_fwd, n4 * sizeof(float4)))) return rc;
            if ((rc = salloc(m, (void**)&m.wall_bwd, n4 * sizeof(float4)))) return rc;
            if ((rc = salloc(m, (void**)&m.wall_rs, NOUTP * 4))) return rc;
            if ((rc = salloc(m, (void**)&m.wall_b, NOUTP * 4))) return rc;
        }
        m.NT = NT; m.NOUTP = NOUTP; m.Kp = Kp;
        const size_t n4 = (size_t)NOUTP * Kp / 4;
        k_soap_prep_wall<<<cdiv((int64_t)NOUTP * Kp, 256), 256, 0, st>>>(d, m.sets, m.n_sets, NOUTP, Kp, m.wall);
        k_soap_prep_rows<<<cdiv(NOUTP, 64), 64, 0, st>>>(d, m.sets, m.n_sets, NOUTP, Kp, m.wall, m.wall_rs, m.wall_b);
        k_pack<<<cdiv(n4, 256), 256, 0, st>>>(m.wall, Kp, 1, NOUTP, Kp, m.wall_fwd);   // x W^T: tiles over NOUTP
        k_pack<<<cdiv(n4, 256), 256, 0, st>>>(m.wall, 1, Kp, Kp, NOUTP, m.wall_bwd);   // dy W: tiles over Kp
        {   // the same weights network by network, for the species-sorted tiles
            const size_t per = (size_t)(Kp / 8) * 64;  // float4 per network, both orientations
            if (!m.wall_fwd_set) {
                if ((rc = salloc(m, (void**)&m.wall_fwd_set, m.n_sets * per * sizeof(float4)))) return rc;
                if ((rc = salloc(m, (void**)&m.wall_bwd_set, m.n_sets * per * sizeof(float4)))) return rc;
            }
            for (int sset = 0; sset < m.n_sets; sset++) {
                const float* ws = m.wall + (size_t)sset * d.H * Kp;
                k_pack<<<cdiv(per, 256), 256, 0, st>>>(ws, Kp, 1, d.H, Kp, m.wall_fwd_set + sset * per);
                k_pack<<<cdiv(per, 256), 256, 0, st>>>(ws, 1, Kp, Kp, d.H, m.wall_bwd_set + sset * per);
            }
        }
        if (d.ncmax <= 32 && d.Sp % 4 == 0) {  // packed power spectrum: the networks' first Linear over the upper-triangle layout
            const int Kpp = (d.Sp + 127) / 128 * 128;
            const size_t per = (size_t)(Kpp / 8) * 64, nw = (size_t)m.n_sets * d.H * Kpp;
            if (!m.wallp || m.Kpp != Kpp) {
                if ((rc = salloc(m, (void**)&m.wallp, nw * 4))) return rc;
                if ((rc = salloc(m, (void**)&m.wallpb, nw * 4))) return rc;
                if ((rc = salloc(m, (void**)&m.wallp_fwd_set, m.n_sets * per * sizeof(float4)))) return rc;
                if ((rc = salloc(m, (void**)&m.wallp_bwd_set, m.n_sets * per * sizeof(float4)))) return rc;
            }
            m.Kpp = Kpp;
            k_soap_prep_wallp<<<cdiv((int64_t)nw, 256), 256, 0, st>>>(d, m.sets, m.n_sets, Kpp, 0, m.wallp);
            k_soap_prep_wallp<<<cdiv((int64_t)nw, 256), 256, 0, st>>>(d, m.sets, m.n_sets, Kpp, 1, m.wallpb);
            for (int sset = 0; sset < m.n_sets; sset++) {
                k_pack<<<cdiv(per, 256), 256, 0, st>>>(m.wallp + (size_t)sset * d.H * Kpp, Kpp, 1, d.H, Kpp,
                                                        m.wallp_fwd_set + sset * per);
                k_pack<<<cdiv(per, 256), 256, 0, st>>>(m.wallpb + (size_t)sset * d.H * Kpp, 1, Kpp, Kpp, d.H,
                                                        m.wallp_bwd_set + sset * per);
            }
        }
        PET_HIP_CHECK(hipGetLastError());
    }
    PET_HIP_CHECK(hipStreamSynchronize(st));  // host vectors go out of scope
    m.finalized = true;
    return PET_OK;
}

static bool soap_pair_ok(const SoapDims& d) {
    return g_soap_pair && d.C == 4 && d.ITEMS <= 64 * MAXI && d.NLM <= 255 && d.F <= 255 && d.NCOEF < 32768;
}
static int g_soap_sorted = 1;
void set_soap_sorted(int v) { g_soap_sorted = v ? 1 : 0; }
static bool soap_sorted_ok(const SoapModel& m) {
    return g_soap_sorted && g_soap_mfma && m.NT > 0 && m.wall_fwd_set != nullptr && m.n_sets <= SP_MAXSETS;
}
// pet_config_set("soap_packed", 0): the full [N][S] feature layout in inference too (the layout the training pass, the
// feature output and the Alchemical centre encoding use)
static int g_soap_packed = 1;
void set_soap_packed(int v) { g_soap_packed = v ? 1 : 0; }
static bool soap_packed_ok(const SoapModel& m) {
    for (int l = 0; l <= m.d.L; l++) {  // every block's triangle an even number of floats (k_soap_ps_m<true> stores 8 bytes per lane)
        const int nc = m.d.n_per_l[l] * m.d.C;
        if ((nc * (nc + 1) / 2) % 2) return false;
    }
    return g_soap_packed && g_soap_pair && g_soap_ps_mfma && m.enc == nullptr && m.wallp_fwd_set != nullptr &&
           m.d.ncmax <= 32 && (size_t)(m.d.NCOEF + m.d.Sp) * 4 <= 64 * 1024;
}
static void soap_note_layout(const SoapModel& m, const void* ws, bool packed) {
    if (m.ws_packed.size() > 64) m.ws_packed.clear();
    m.ws_packed[ws] = packed;
}
static bool soap_ws_packed(const SoapModel& m, const void* ws) {
    auto it = m.ws_packed.find(ws);
    return it != m.ws_packed.end() && it->second;
}
static size_t lds_expand(const SoapDims& d) { return (size_t)PC * (d.NLM + d.F + 8 + 1) * 4; }
static size_t lds_expand_bwd(const SoapDims& d) {
    return ((size_t)d.NCOEF + (size_t)PC * (4 * d.NLM + 2 * d.F + 8 + 1)) * 4;
}
static size_t lds_tail(const SoapDims& d) { return ((size_t)d.S + 256 * (d.H + 1) + 8 + 2 * d.H) * 4; }

static int soap_fwd(const SoapModel& m, const Graph& g, void* ws, int64_t ws_bytes, float* atomic, float* features,
                    hipStream_t st) {
    const SoapDims& d = m.d;
    SoapWs w;
    carve_soap(d, g.n_nodes, g.n_edges, ws, w);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "soap workspace too small");
    const int N = (int)g.n_nodes;
    if (N == 0) return PET_OK;
    allow_big_lds(k_soap_tail, lds_tail(d));
    {
        ProfScope ps("soap_expand", st, 2.0 * (double)g.n_edges * d.NCOEF, (double)g.n_edges * 20 + (double)N * d.NCOEF * 4);
        if (soap_pair_ok(d)) {
            const size_t lds = (size_t)4 * 32 * (((d.NLM + d.F + 4 + 3) & ~3) + 4) * 4;
            allow_big_lds(k_soap_expand_w<6>, lds);
            allow_big_lds(k_soap_expand_w<MAXL>, lds);
            if (d.L <= 6)
                k_soap_expand_w<6><<<cdiv(N, 4), 256, lds, st>>>(d, g.geo, g.rowptr, g.sp_nbr, m.table, m.shnorm,
                                                                  m.item_lut, m.species_w, w.Cf, N);
            else
                k_soap_expand_w<MAXL><<<cdiv(N, 4), 256, lds, st>>>(d, g.geo, g.rowptr, g.sp_nbr, m.table, m.shnorm,
                                                                     m.item_lut, m.species_w, w.Cf, N);
        } else {
            k_soap_expand<<<N, 256, lds_expand(d), st>>>(d, g.geo, g.rowptr, g.sp_nbr, m.table, m.shnorm, m.coef_lut,
                                                         m.species_w, w.Cf);
        }
    }
    const bool packed = soap_sorted_ok(m) && soap_packed_ok(m) && features == nullptr;
    soap_note_layout(m, ws, packed);
    {
    {
            ProfScope ps("soap_ps", st, 2.0 * (double)N * d.S * (d.L + 1), (double)N * (d.NCOEF + (packed ? d.Sp : d.S)) * 4);
            if (packed) {
                const size_t lds_ps = (size_t)4 * (d.NCOEF + ((d.ncmax * (d.ncmax + 1) / 2 + 3) & ~3)) * 4;
                allow_big_lds(k_soap_ps_m<true>, lds_ps);
                k_soap_ps_m<true><<<cdiv(N, 4), 256, lds_ps, st>>>(d, w.Cf, g.sp, nullptr, w.feats, w.tail, N);
            } else if (g_soap_pair && g_soap_ps_mfma && d.ncmax <= 32) {
                allow_big_lds(k_soap_ps_m<false>, (size_t)4 * d.NCOEF * 4);
                k_soap_ps_m<false><<<cdiv(N, 4), 256, (size_t)4 * d.NCOEF * 4, st>>>(d, w.Cf, g.sp, m.enc, w.feats, w.tail, N);
            } else if (g_soap_pair && d.ncmax < 128 && d.NCOEF < 65536) {
                allow_big_lds(k_soap_ps_w, (size_t)4 * d.NCOEF * 4);
                k_soap_ps_w<<<cdiv(N, 4), 256, (size_t)4 * d.NCOEF * 4, st>>>(d, w.Cf, g.sp, m.enc, m.feat_lut, w.feats,
                                                                            w.tail, N);
            } else {
                k_soap_ps<<<N, 256, (size_t)d.NCOEF * 4, st>>>(d, w.Cf, g.sp, m.enc, w.feats, w.tail);
            }
        }
        {
            ProfScope ps("soap_tail", st, 2.0 * (double)N * d.S * d.H, (double)N * (packed ? d.Sp : d.S) * 4);
            if (soap_sorted_ok(m)) {
                PET_HIP_CHECK(hipMemsetAsync(w.info, 0, sizeof(SpInfo), st));
                k_sp_count<<<cdiv(N, 256), 256, 0, st>>>(g.sp, d.legacy, N, w.info);
                k_sp_scan<<<1, 1, 0, st>>>(m.n_sets, w.info);
                k_sp_fill<<<cdiv(N, 1024), 1024, 0, st>>>(g.sp, d.legacy, N, w.info, w.perm);
                const size_t lds = ((size_t)BM * lds_ld(128) + BM * 32 + BM) * 4;
                SoapDims dt = d;
                if (packed) dt.S = d.Sp;  // (the kernel uses S as the row pitch / length of the feature rows only)
                k_soap_tail_fwd_set<<<cdiv(N, BM) + m.n_sets, NTHREADS, lds, st>>>(
                    dt, w.feats, w.perm, w.info, m.n_sets, m.sets, packed ? m.wallp_fwd_set : m.wall_fwd_set,
                    packed ? m.Kpp : m.Kp, m.wall_rs, m.wall_b, w.tail, atomic);
            } else if (m.NT > 0 && g_soap_mfma) {
                const int NOUTP = m.NOUTP, lda = lds_ld(128);
                const size_t lds = ((size_t)BM * (NOUTP + 1 > lda ? NOUTP + 1 : lda) + BM * 32) * 4;
                const int grid = cdiv(N, BM);
#define SOAP_TAIL_FWD(NTV)                                                                                        \
        case NTV:                                                                                                     \
            allow_big_lds(k_soap_tail_fwd_mfma<NTV>, lds);                                                            \
            k_soap_tail_fwd_mfma<NTV><<<grid, NTHREADS, lds, st>>>(d, w.feats, g.sp, m.sets, m.wall_fwd, m.Kp,        \
                                                                    m.wall_rs, m.wall_b, w.tail, atomic, N);           \
            break;
                switch (m.NT) { SOAP_TAIL_FWD(1) SOAP_TAIL_FWD(2) SOAP_TAIL_FWD(3) SOAP_TAIL_FWD(4) }
#undef SOAP_TAIL_FWD
            } else {
                k_soap_tail<<<N, 256, lds_tail(d), st>>>(d, w.feats, g.sp, m.sets, w.tail, atomic);
            }
        }
    }
    if (d.NH > 2) k_soap_tail_extra_fwd<<<cdiv(N, 4), 256, 0, st>>>(d, g.sp, m.sets, w.tail, w.tail_ext, atomic, N);
    if (features)
        PET_HIP_CHECK(hipMemcpyAsync(features, w.feats, (size_t)N * d.S * 4, hipMemcpyDeviceToDevice, st));
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

static int soap_bwd(const SoapModel& m, const Graph& g, void* ws, int64_t ws_bytes, const float* gA, float* gpos,
                    float* gcell, hipStream_t st) {
    const SoapDims& d = m.d;
    SoapWs w;
    carve_soap(d, g.n_nodes, g.n_edges, ws, w);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "soap workspace too small");
    const int N = (int)g.n_nodes;
    if (N == 0) return PET_OK;
    if (g.n_edges == 0) {
        PET_HIP_CHECK(hipMemsetAsync(gpos, 0, (size_t)N * 3 * 4, st));
        if (gcell) PET_HIP_CHECK(hipMemsetAsync(gcell, 0, g.n_systems * 9 * 4, st));
        return PET_OK;
    }
    allow_big_lds(k_soap_expand_bwd, lds_expand_bwd(d));
    const float* da2x = nullptr;
    if (d.NH > 2) {
        k_soap_tail_extra_bwd<<<cdiv(N, 4), 256, 0, st>>>(d, g.sp, m.sets, w.tail, w.tail_ext, gA, w.da2, N);
        da2x = w.da2;
    }
    const bool packed = soap_ws_packed(m, ws);  // what the forward into this workspace stored
    PET_REQUIRE(!packed || (soap_sorted_ok(m) && soap_packed_ok(m)), PET_ERR_ARGUMENT,
                "the forward of this workspace stored the packed power spectrum but the adjoint is configured for the full "
                "layout: pet_config_set changed between soap_forward and soap_backward");
    {
    {
        ProfScope ps("soap_tail_bwd", st, 2.0 * (double)N * d.S * d.H, (double)N * (packed ? d.Sp : d.S) * 8);
        if (soap_sorted_ok(m)) {  // perm / info were filled by the forward pass on this workspace
            const size_t lds = ((size_t)BM * lds_ld(32) + BM * 32 + BM * 4 + BM * lds_ld(128) + BM) * 4;
            k_soap_tail_bwd_set<<<cdiv(N, BM) + m.n_sets, NTHREADS, lds, st>>>(
                d, w.feats, w.perm, w.info, m.n_sets, g.sp, m.sets, packed ? m.wallp_bwd_set : m.wall_bwd_set,
                packed ? m.Kpp : m.Kp, m.wall_rs, m.wall_b, m.enc, w.tail, gA, w.dF, da2x, packed ? 1 : 0);
        } else if (m.NT > 0 && g_soap_mfma) {
            const size_t lds = ((size_t)BM * lds_ld(m.NOUTP) + BM * 32 + BM * 4 + BM * lds_ld(128)) * 4;
            const int grid = cdiv(N, BM);
#define SOAP_TAIL_BWD(NTV)                                                                                        \
    case NTV:                                                                                                     \
        allow_big_lds(k_soap_tail_bwd_mfma<NTV>, lds);                                                            \
        k_soap_tail_bwd_mfma<NTV><<<grid, NTHREADS, lds, st>>>(d, w.feats, g.sp, m.sets, m.wall_bwd, m.Kp,        \
                                                                m.wall_rs, m.wall_b, m.enc, w.tail, gA, w.dF, N,   \
                                                                da2x);                                             \
        break;
            switch (m.NT) { SOAP_TAIL_BWD(1) SOAP_TAIL_BWD(2) SOAP_TAIL_BWD(3) SOAP_TAIL_BWD(4) }
#undef SOAP_TAIL_BWD
        } else {
            k_soap_tail_bwd<<<N, 256, 0, st>>>(d, w.feats, g.sp, m.sets, m.enc, w.tail, gA, w.dF, da2x);
        }
    }
    {
        ProfScope ps("soap_ps_bwd", st, 4.0 * (double)N * d.S * (d.L + 1), (double)N * (2 * d.NCOEF + (packed ? d.Sp : d.S)) * 4);
        if (packed) {
            k_soap_ps_bwd_m<true><<<N, 256, (size_t)(d.NCOEF + d.Sp) * 4, st>>>(d, w.Cf, w.dF, w.dCf);
        } else if (g_soap_pair && g_soap_ps_mfma && d.S % 4 == 0 && (size_t)(d.NCOEF + d.S) * 4 <= 64 * 1024) {
            k_soap_ps_bwd_m<false><<<N, 256, (size_t)(d.NCOEF + d.S) * 4, st>>>(d, w.Cf, w.dF, w.dCf);
        } else if (g_soap_pair && d.S % 4 == 0 && d.NCOEF < 65536 && (size_t)(d.NCOEF + d.S) * 4 <= 64 * 1024) {
            k_soap_ps_bwd_s<<<N, 256, (size_t)(d.NCOEF + d.S) * 4, st>>>(d, w.Cf, w.dF, m.out_lut, w.dCf);
        } else {
            k_soap_ps_bwd<<<N, 256, (size_t)d.NCOEF * 4, st>>>(d, w.Cf, w.dF, m.coef_lut, w.dCf);
        }
    }
    }
    {
        ProfScope ps("soap_expand_bwd", st, 2.0 * (double)g.n_edges * (d.NCOEF + 8.0 * d.ITEMS),
                     (double)g.n_edges * 36 + (double)N * d.NCOEF * 4);
        if (soap_pair_ok(d)) {
            const int grid = (int)cdiv(g.n_edges, 256);
            const int lds_floats = (d.NCOEF % 4 == 0) ? 16 * 1024 : 0;  // 64 KB: the rows of up to ~14 centres of the default basis
            if (d.L <= 6) {
                allow_big_lds(k_soap_expand_bwd_p<6>, (size_t)lds_floats * 4);
                k_soap_expand_bwd_p<6><<<grid, 256, (size_t)lds_floats * 4, st>>>(d, g.geo, g.ctr, g.sp_nbr, m.table, m.shnorm,
                                                                                 m.species_w, w.dCf,
                                                                                 reinterpret_cast<float4*>(w.dv), g.n_edges,
                                                                                 lds_floats);
            } else {
                allow_big_lds(k_soap_expand_bwd_p<MAXL>, (size_t)lds_floats * 4);
                k_soap_expand_bwd_p<MAXL><<<grid, 256, (size_t)lds_floats * 4, st>>>(d, g.geo, g.ctr, g.sp_nbr, m.table,
                                                                                    m.shnorm, m.species_w, w.dCf,
                                                                                    reinterpret_cast<float4*>(w.dv),
                                                                                    g.n_edges, lds_floats);
            }
        } else {
            k_soap_expand_bwd<<<N, 256, lds_expand_bwd(d), st>>>(d, g.geo, g.rowptr, g.sp_nbr, m.table, m.shnorm,
                                                                 m.item_lut, m.species_w, w.dCf,
                                                                 reinterpret_cast<float4*>(w.dv));
        }
    }
    k_pos_grad<<<cdiv(N, 16), 256, 0, st>>>(reinterpret_cast<const float4*>(w.dv), g.rowptr, g.rev, gpos, N);
    if (gcell)
        k_cell_grad<<<(int)g.n_systems, 256, 0, st>>>(reinterpret_cast<const float4*>(w.dv), g.shift, g.ctr, g.sys,
                                                      g.rowptr, gcell, N, g.n_edges, 0);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

#include "soap_train.h"

}  // namespace pet

using namespace pet;

struct soap_model {
    SoapModel m;
};

extern "C" {

int soap_model_create(const soap_hypers_t* h, soap_model_t** out) {
    PET_REQUIRE(h && out, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(h->max_angular >= 0 && h->max_angular <= SOAP_MAX_L, PET_ERR_UNSUPPORTED, "max_angular out of range");
    PET_REQUIRE(h->num_neurons_per_layer >= 1 && h->num_neurons_per_layer <= MAXH, PET_ERR_UNSUPPORTED,
                "num_neurons_per_layer > 64 is not built");
    PET_REQUIRE(h->num_hidden_layers >= 1 && h->num_hidden_layers <= MAXNH, PET_ERR_UNSUPPORTED,
                "num_hidden_layers must be 1 .. 8");
    PET_REQUIRE(h->n_species >= 1 && h->n_channels >= 1 && h->n_channels <= 255, PET_ERR_ARGUMENT, "bad species counts");
    PET_REQUIRE(!h->legacy || h->n_channels == h->n_species, PET_ERR_ARGUMENT,
                "legacy (Orthogonal species): n_channels must equal n_species");
    soap_model_t* sm = new soap_model_t();
    SoapModel& m = sm->m;
    m.h = *h;
    SoapDims& d = m.d;
    memset(&d, 0, sizeof(d));
    d.L = h->max_angular; d.C = h->n_channels; d.ns = h->n_species; d.legacy = h->legacy; d.layernorm = h->layernorm;
    d.H = h->num_neurons_per_layer; d.NH = h->num_hidden_layers; d.rc = h->cutoff; d.width = h->cutoff_width;
    d.NLM = (d.L + 1) * (d.L + 1);
    int f = 0, co = 0, fo = 0, items = 0, pfo = 0;
    for (int l = 0; l <= d.L; l++) {
        const int n = h->n_per_l[l];
        if (n < 0 || n > 64) { delete sm; set_error("bad n_per_l"); return PET_ERR_ARGUMENT; }
        d.n_per_l[l] = n; d.rad_off[l] = f; d.coef_off[l] = co; d.feat_off[l] = fo;
        if (n * d.C > d.ncmax) d.ncmax = n * d.C;
        d.pfeat_off[l] = pfo;
        f += n; co += (2 * l + 1) * n * d.C; fo += (n * d.C) * (n * d.C); items += (2 * l + 1) * n;
        pfo += (n * d.C) * (n * d.C + 1) / 2;
    }
    d.coef_off[d.L + 1] = co; d.feat_off[d.L + 1] = fo; d.pfeat_off[d.L + 1] = pfo; d.Sp = pfo;
    d.F = f; d.NCOEF = co; d.S = fo; d.ITEMS = items;
    m.n_sets = h->legacy ? h->n_species : 1;
    if (d.F > 255 || d.NLM > 255 || d.NCOEF > 256 * MAXK || d.NCOEF >= 65536) {
        delete sm;
        set_error("SOAP basis too large for the compiled limits");
        return PET_ERR_UNSUPPORTED;
    }
    *out = sm;
    return PET_OK;
}

void soap_model_destroy(soap_model_t* sm) {
    if (!sm) return;
    for (void* p : sm->m.owned) (void)hipFree(p);
    delete sm;
}

int64_t soap_model_feature_size(const soap_model_t* sm) { return sm ? sm->m.d.S : -1; }

int soap_model_set_radial_table(soap_model_t* sm, const float* d_table, int32_t n_grid, void* stream) {
    PET_REQUIRE(sm && d_table && n_grid >= 4, PET_ERR_ARGUMENT, "bad argument");
    SoapModel& m = sm->m;
    const size_t bytes = (size_t)n_grid * m.d.F * 4 * sizeof(float);
    int rc = salloc(m, (void**)&m.table, bytes);
    if (rc) return rc;
    PET_HIP_CHECK(hipMemcpyAsync(m.table, d_table, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    m.d.n_grid = n_grid;
    m.d.inv_h = (float)(n_grid - 1) / m.d.rc;
    m.finalized = false;
    return PET_OK;
}

int soap_model_set_param(soap_model_t* sm, const char* key, const float* d_data, int64_t numel, void* stream) {
    PET_REQUIRE(sm && key && d_data && numel > 0, PET_ERR_ARGUMENT, "bad argument");
    SoapModel& m = sm->m;
    float* p;
    auto it = m.raw.find(key);
    if (it != m.raw.end() && it->second.second == numel) {
        p = it->second.first;
    } else {
        int rc = salloc(m, (void**)&p, numel * sizeof(float));
        if (rc) return rc;
        m.raw[key] = {p, numel};
    }
    PET_HIP_CHECK(hipMemcpyAsync(p, d_data, numel * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    m.finalized = false;
    return PET_OK;
}

int soap_model_finalize(soap_model_t* sm, void* stream) {
    PET_REQUIRE(sm, PET_ERR_ARGUMENT, "null model");
    return soap_finalize(sm->m, (hipStream_t)stream);
}

int64_t soap_workspace_bytes(const soap_model_t* sm, int64_t n_nodes, int64_t n_edges) {
    if (!sm) return -1;
    SoapWs w;
    carve_soap(sm->m.d, n_nodes, n_edges, nullptr, w);
    return (int64_t)w.bytes;
}

int soap_forward(const soap_model_t* sm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                 float* d_atomic, float* d_features, void* stream) {
    PET_REQUIRE(sm && pg, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(sm->m.finalized, PET_ERR_ARGUMENT, "soap_model_finalize has not been called");
    if (pg->g.n_nodes == 0) return PET_OK;   // an empty system: nothing to write (zero-sized buffers may be null)
    PET_REQUIRE(d_workspace && d_atomic, PET_ERR_ARGUMENT, "null argument");
    return soap_fwd(sm->m, pg->g, d_workspace, workspace_bytes, d_atomic, d_features, (hipStream_t)stream);
}

int soap_backward(const soap_model_t* sm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                  const float* d_grad_atomic, float* d_grad_positions, float* d_grad_cells, void* stream) {
    PET_REQUIRE(sm && pg, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(sm->m.finalized, PET_ERR_ARGUMENT, "soap_model_finalize has not been called");
    if (pg->g.n_nodes == 0) return PET_OK;
    PET_REQUIRE(d_workspace && d_grad_atomic && d_grad_positions, PET_ERR_ARGUMENT, "null argument");
    return soap_bwd(sm->m, pg->g, d_workspace, workspace_bytes, d_grad_atomic, d_grad_positions, d_grad_cells,
                    (hipStream_t)stream);
}

int soap_model_zero_grad(soap_model_t* sm, void* stream) {
    PET_REQUIRE(sm, PET_ERR_ARGUMENT, "null model");
    return soap_zero_grad(sm->m, (hipStream_t)stream);
}

int64_t soap_train_workspace_bytes(const soap_model_t* sm, int64_t n_nodes, int64_t n_edges) {
    if (!sm) return -1;
    SoapTrainWs w;
    carve_soap_train(sm->m, n_nodes, n_edges, nullptr, w);
    return (int64_t)w.bytes;
}

int soap_train_gradients(soap_model_t* sm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                         void* d_train_workspace, int64_t train_workspace_bytes, const float* d_grad_atomic,
                         const float* d_u, float* d_tangent_atomic, void* stream) {
    PET_REQUIRE(sm && pg, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(sm->m.finalized, PET_ERR_ARGUMENT, "soap_model_finalize has not been called");
    if (pg->g.n_nodes == 0) return PET_OK;
    PET_REQUIRE(d_workspace && d_train_workspace && d_grad_atomic && d_tangent_atomic, PET_ERR_ARGUMENT, "null argument");
    return soap_train_grads(sm->m, pg->g, d_workspace, workspace_bytes, d_train_workspace, train_workspace_bytes,
                            d_grad_atomic, d_u, d_tangent_atomic, (hipStream_t)stream);
}

static int soap_copy_out(const std::map<std::string, std::pair<float*, int64_t>>& table, const char* key, float* d_out,
                         int64_t numel, void* stream) {
    auto it = table.find(key);
    PET_REQUIRE(it != table.end(), PET_ERR_ARGUMENT, std::string("no entry '") + key + "'");
    PET_REQUIRE(it->second.second == numel, PET_ERR_ARGUMENT, std::string("'") + key + "' has another size");
    PET_HIP_CHECK(hipMemcpyAsync(d_out, it->second.first, numel * sizeof(float), hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
    return PET_OK;
}

int soap_model_get_grad(const soap_model_t* sm, const char* key, float* d_out, int64_t numel, void* stream) {
    PET_REQUIRE(sm && key && d_out, PET_ERR_ARGUMENT, "null argument");
    return soap_copy_out(sm->m.grad, key, d_out, numel, stream);
}

int soap_model_get_param(const soap_model_t* sm, const char* key, float* d_out, int64_t numel, void* stream) {
    PET_REQUIRE(sm && key && d_out, PET_ERR_ARGUMENT, "null argument");
    return soap_copy_out(sm->m.raw, key, d_out, numel, stream);
}

int soap_adam_step(soap_model_t* sm, float lr, float beta1, float beta2, float eps, int64_t step, void* stream) {
    PET_REQUIRE(sm, PET_ERR_ARGUMENT, "null model");
    return soap_adam(sm->m, lr, beta1, beta2, eps, step, (hipStream_t)stream);
}

}  // extern "C"
