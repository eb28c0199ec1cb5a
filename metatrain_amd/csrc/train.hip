// First-order parameter gradients of the PET backend (training row a16, energy term):
// weight-gradient GEMMs and the small reductions around them. See train.h / wgrad.h.
#include "train.h"

#include <vector>

#include "wgrad.h"

namespace pet {

float* Trainer::gp(const std::string& key) const {
    auto it = m.grad_off.find(key);
    return it == m.grad_off.end() ? nullptr : grads + it->second;
}

#define TR_CHECK(expr)                                         \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess && err == PET_OK) {               \
            err = PET_ERR_HIP;                                 \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                      \
    } while (0)

// out[n * ldo + col0 + k] (+)= sum_s partial[s][n][k]
// 64 outputs per block, the splits dealt over 4 thread groups (fixed order -> deterministic), LDS combine
__global__ __launch_bounds__(256) void k_reduce_2d(const float* __restrict__ partial, int nsplit, int n_out, int kb,
                                                   float* __restrict__ out, int ldo, int col0, int accumulate) {
    __shared__ float red[4][64];
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + o;
    const int64_t tot = (int64_t)n_out * kb;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < tot) {
        const float* p = partial + i;
        int k = sg;
        for (; k + 12 < nsplit; k += 16) {
            s0 += p[(size_t)k * tot];
            s1 += p[(size_t)(k + 4) * tot];
            s2 += p[(size_t)(k + 8) * tot];
            s3 += p[(size_t)(k + 12) * tot];
        }
        for (; k < nsplit; k += 4) s0 += p[(size_t)k * tot];
    }
    red[sg][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sg == 0 && i < tot) {
        const float s = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
        float* dst = out + (i / kb) * ldo + col0 + (i % kb);
        *dst = accumulate ? *dst + s : s;
    }
}
static void reduce_2d(const float* partial, int nsplit, int n_out, int kb, float* out, int ldo, int col0, int accumulate,
                      hipStream_t st) {
    k_reduce_2d<<<cdiv((int64_t)n_out * kb, 64), 256, 0, st>>>(partial, nsplit, n_out, kb, out, ldo, col0, accumulate);
}

// norm feeding a Linear (y = xhat gamma + beta):  dW += G gamma + db (x) beta,
//   dgamma[k] = sum_n W[n][k] G[n][k],   LayerNorm only: dbeta[k] = sum_n W[n][k] db[n].
// Elementwise part; G is overwritten with W.G and G2 (LayerNorm) with W[n][k] db[n] for the column sums.
__global__ void k_norm_fixup(float* __restrict__ G, float* __restrict__ G2, const float* __restrict__ W,
                             const float* __restrict__ gamma, const float* __restrict__ beta,
                             const float* __restrict__ db, int n_out, int k_in, float* __restrict__ dW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_out * k_in) return;
    const int n = (int)(i / k_in), k = (int)(i % k_in);
    const float gv = G[i], wv = W[i];
    float upd = gv * gamma[k];
    if (G2) {
        const float dbn = db[n];
        upd += dbn * beta[k];
        G2[i] = wv * dbn;
    }
    dW[i] += upd;
    G[i] = wv * gv;
}

// column sums of a row-major [R, C] buffer, two stages; also sums with a per-row species index
__global__ void k_colsum_partial(const float* __restrict__ buf, int64_t n_rows, int C, float* __restrict__ partial) {
    const int c = threadIdx.x;
    const int nsplit = gridDim.x;
    const int64_t per = (n_rows + nsplit - 1) / nsplit;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(n_rows, r0 + per);
    float s = 0.f;
    if (c < C)
        for (int64_t r = r0; r < r1; r++) s += buf[r * C + c];
    if (c < C) partial[(size_t)blockIdx.x * C + c] = s;
}

// partial[split][species][c] = sum over the split's rows with idx[row] == species
// (species s0 .. s0 + ns - 1 only: the host walks the species axis in chunks that fit 64 KB of LDS)
__global__ void k_species_sum_partial(const float* __restrict__ buf, const int* __restrict__ idx, int64_t n_rows,
                                      int C, int ns, int s0, float* __restrict__ partial) {
    extern __shared__ float acc[];  // [ns][C]
    const int c = threadIdx.x;
    for (int i = threadIdx.x; i < ns * C; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int nsplit = gridDim.x;
    const int64_t per = (n_rows + nsplit - 1) / nsplit;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(n_rows, r0 + per);
    if (c < C)
        for (int64_t r = r0; r < r1; r++) {
            const int sp = idx[r] - s0;
            if (sp >= 0 && sp < ns) acc[sp * C + c] += buf[r * C + c];  // column c is private to this thread
        }
    __syncthreads();
    for (int i = threadIdx.x; i < ns * C; i += blockDim.x) partial[(size_t)blockIdx.x * ns * C + i] = acc[i];
}

// dWc[n][c] = sum_rows da0[row][n] * geo[row][c]   (c < 4), partial per split
__global__ void k_geo_wgrad_partial(const float* __restrict__ da0, const float4* __restrict__ geo, int64_t n_rows,
                                    float* __restrict__ partial) {
    const int n = threadIdx.x;  // 128 threads
    const int nsplit = gridDim.x;
    const int64_t per = (n_rows + nsplit - 1) / nsplit;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(n_rows, r0 + per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = r0; r < r1; r++) {
        const float d = da0[r * D + n];
        const float4 gv = geo[r];
        s.x += d * gv.x; s.y += d * gv.y; s.z += d * gv.z; s.w += d * gv.w;
    }
    reinterpret_cast<float4*>(partial)[(size_t)blockIdx.x * D + n] = s;
}

__global__ void k_edge_gy_sum_partial(const float* __restrict__ gA, const int* __restrict__ ctr,
                                      const float* __restrict__ fc, int64_t n_rows, float* __restrict__ partial) {
    // sum_p gA[ctr[p]] * fc[p]  (edge last-layer bias gradient); one block = one split, LDS tree
    __shared__ float red[256];
    const int nsplit = gridDim.x;
    const int64_t per = (n_rows + nsplit - 1) / nsplit;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(n_rows, r0 + per);
    float s = 0.f;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) s += ctr ? gA[ctr[r]] * fc[r] : gA[r];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

template <int KB, int XMODE>
static void launch_k_wgrad(const WgradArgs& a, int nb, int nsplit, hipStream_t st) {
    const size_t lds = (size_t)WG_RB * (132 + KB + 4) * sizeof(float);
    k_wgrad<KB, XMODE><<<dim3(nb, nsplit), NTHREADS, lds, st>>>(a);
}
template <int XMODE>
static void launch_k_wgrad_b(const WgradArgs& a, int nb, int nsplit, hipStream_t st) {
    if (train_bf16()) {  // single-term mode: one plane per operand (20 KB)
        k_wgrad_b<XMODE, true><<<nb * nsplit, NTHREADS, (size_t)2 * 128 * WB_LDW * sizeof(unsigned), st>>>(a, nb, nsplit);
        return;
    }
    const size_t lds = (size_t)6 * 128 * WB_LDW * sizeof(unsigned);  // 60 KB: two workgroups per CU
    k_wgrad_b<XMODE><<<nb * nsplit, NTHREADS, lds, st>>>(a, nb, nsplit);
}

// pet_config_set("wgrad_bf16", 0): weight gradients on the fp32 MFMA (k_wgrad) instead of bf16x3 (k_wgrad_b, default)
static int g_wgrad_bf16 = 1;
void set_wgrad_bf16(int v) { g_wgrad_bf16 = v ? 1 : 0; }

// dW block [n_out, k_in] into dst (row stride ldw) and, if db_dst, the bias gradient
static void wgrad_core(Trainer& t, int n_out, int k_in, Trainer::Y y, Trainer::X x, int xmode, int64_t n_rows,
                       float* dst, int ldw, float* db_dst, bool accumulate) {
    if (n_rows <= 0 || t.err) return;
    ProfScope ps("wgrad", t.st, 2.0 * (double)n_rows * n_out * k_in, 4.0 * (double)n_rows * (n_out + k_in));
    const int nb = n_out / 128;
    // bf16x3 kernel: 128 x columns per launch; the RMSNorm-hat source needs its whole row in one launch
    const bool whole_row = xmode == 1 || xmode == 5;  // the norm-hat sources need the whole row in one launch
    const bool b16 = g_wgrad_bf16 && !(whole_row && k_in != 128);
    const int KB = b16 ? 128 : ((whole_row || xmode == 4) ? k_in : 128);
    int nsplit = (b16 ? 512 : 768) / nb;  // two (bf16x3: 60 KB of LDS) / three (fp32: 34 - 50 KB) workgroups per CU
    const int64_t max_by_rows = (n_rows + WG_RB - 1) / WG_RB;
    if (nsplit > max_by_rows) nsplit = (int)max_by_rows;
    while ((size_t)nsplit * n_out * (KB + 1) > t.w.partial_floats && nsplit > 1) nsplit /= 2;
    if (b16 && nsplit > 8) nsplit &= ~7;  // XCD-aware workgroup numbering wants a multiple of 8
    float* pb = t.w.partial + (size_t)nsplit * n_out * KB;
    for (int k0 = 0; k0 < k_in; k0 += KB) {
        WgradArgs a;
        a.y0 = y.p0; a.y1 = y.p1; a.y_split = y.split; a.y_ld = y.ld; a.y_col0 = 0;
        a.x0 = x.p; a.x_ld = x.ld; a.x_col0 = k0; a.x_hid = x.hid; a.rev = x.rev; a.lns = x.lns;
        a.n_rows = n_rows; a.partial = t.w.partial; a.partial_b = (k0 == 0 && db_dst) ? pb : nullptr; a.n_out = n_out;
        if (b16) {
            switch (xmode) {
                case 0: launch_k_wgrad_b<0>(a, nb, nsplit, t.st); break;
                case 1: launch_k_wgrad_b<1>(a, nb, nsplit, t.st); break;
                case 2: launch_k_wgrad_b<2>(a, nb, nsplit, t.st); break;
                case 3: launch_k_wgrad_b<3>(a, nb, nsplit, t.st); break;
                case 4: launch_k_wgrad_b<4>(a, nb, nsplit, t.st); break;
                case 5: launch_k_wgrad_b<5>(a, nb, nsplit, t.st); break;
                default: t.err = PET_ERR_ARGUMENT; return;
            }
        } else if (KB == 128) {
            switch (xmode) {
                case 0: launch_k_wgrad<128, 0>(a, nb, nsplit, t.st); break;
                case 1: launch_k_wgrad<128, 1>(a, nb, nsplit, t.st); break;
                case 2: launch_k_wgrad<128, 2>(a, nb, nsplit, t.st); break;
                case 3: launch_k_wgrad<128, 3>(a, nb, nsplit, t.st); break;
                case 5: launch_k_wgrad<128, 5>(a, nb, nsplit, t.st); break;
                default: t.err = PET_ERR_ARGUMENT; return;
            }
        } else {
            if (xmode == 1) launch_k_wgrad<256, 1>(a, nb, nsplit, t.st);
            else if (xmode == 4) launch_k_wgrad<256, 4>(a, nb, nsplit, t.st);
            else if (xmode == 5) launch_k_wgrad<256, 5>(a, nb, nsplit, t.st);
            else { t.err = PET_ERR_ARGUMENT; return; }
        }
        reduce_2d(t.w.partial, nsplit, n_out, KB, dst, ldw, k0, accumulate ? 1 : 0, t.st);
        if (k0 == 0 && db_dst)
            reduce_2d(pb, nsplit, n_out, 1, db_dst, 1, 0, accumulate ? 1 : 0, t.st);
    }
}

void Trainer::linear(const std::string& key, int n_out, int k_in, Y y, X x, int xmode, int64_t n_rows,
                     bool with_bias) {
    float* dW = gp(key + ".weight");
    float* db = with_bias ? gp(key + ".bias") : nullptr;
    if (!dW) { err = PET_ERR_ARGUMENT; set_error("no gradient slot for " + key); return; }
    wgrad_core(*this, n_out, k_in, y, x, xmode, n_rows, dW, k_in, db, true);
}

void Trainer::linear_after_norm(const std::string& key, const float* W, int n_out, int k_in, Y y, X x, int xmode,
                                int64_t n_rows, const std::string& gamma_key, const float* gamma,
                                const std::string& beta_key, const float* beta, bool tangent_pair) {
    if (n_rows <= 0 || err) return;
    float* dW = gp(key + ".weight");
    float* db = gp(key + ".bias");
    float* dgamma = gp(gamma_key);
    float* dbeta = (beta_key.empty() || tangent_pair) ? nullptr : gp(beta_key);
    if (!dW || !db || !dgamma) { err = PET_ERR_ARGUMENT; set_error("no gradient slot for " + key); return; }
    // G = dY^T xhat into scratch, bias gradient of THIS call into gvec (needed un-accumulated for dbeta)
    // a tangent pair (lambda_y, d xhat) has no bias / beta term: d(y) = W (gamma * d xhat)
    wgrad_core(*this, n_out, k_in, y, x, xmode, n_rows, w.gmat, k_in, tangent_pair ? nullptr : w.gvec, false);
    float* G2 = dbeta ? w.gmat + 1024 * 256 : nullptr;
    k_norm_fixup<<<cdiv((int64_t)n_out * k_in, 256), 256, 0, st>>>(w.gmat, G2, W, gamma, beta, w.gvec, n_out, k_in, dW);
    colsum(w.gmat, n_out, k_in, dgamma);
    if (dbeta) colsum(G2, n_out, k_in, dbeta);
    if (!tangent_pair) reduce_2d(w.gvec, 1, n_out, 1, db, 1, 0, 1, st);
}

// ---------------------------------------------------------------------------------------------
// system conditioning: parameter gradients of  cond_s = W2 silu(W0 [emb_q(charge_s) ; emb_m(spin_s)] + b0) + b2
// ---------------------------------------------------------------------------------------------
// dcond[s][c] (=|+=) sum over the atoms of system s of dH[i][c]: one block per system, the system's atoms are a
// contiguous run of the (non-decreasing) system indices, fixed summation order
__global__ __launch_bounds__(DN) void k_cond_accum(const float* __restrict__ dH, const int* __restrict__ sys32,
                                                   const int64_t* __restrict__ sys64, int N, float* __restrict__ dcond,
                                                   int accumulate) {
    const int s = blockIdx.x, c = threadIdx.x;
    __shared__ int range[2];
    if (c == 0) {
        auto at = [&](int i) { return sys64 ? sys64[i] : (int64_t)sys32[i]; };
        int a = 0, b = N;
        while (a < b) { const int mid = (a + b) >> 1; if (at(mid) < s) a = mid + 1; else b = mid; }
        range[0] = a; b = N;
        while (a < b) { const int mid = (a + b) >> 1; if (at(mid) < s + 1) a = mid + 1; else b = mid; }
        range[1] = a;
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = range[0]; i < range[1]; i++) acc += dH[(size_t)i * DN + c];
    dcond[(size_t)s * DN + c] = accumulate ? dcond[(size_t)s * DN + c] + acc : acc;
}
// per system: the projection's hidden row and the adjoints that the weight gradients need; scr [n_sys][5 DN] =
// (x [2 DN] | hid [DN] | da [DN] | -- ) and dx [n_sys][2 DN]
__global__ __launch_bounds__(DN) void k_cond_bwd_rows(const int64_t* __restrict__ charge, const int64_t* __restrict__ spin,
                                                      const float* __restrict__ emb_q, const float* __restrict__ emb_m,
                                                      const float* __restrict__ w0, const float* __restrict__ b0,
                                                      const float* __restrict__ w2, const float* __restrict__ dcond,
                                                      float* __restrict__ scr, float* __restrict__ dx, int max_charge,
                                                      int max_spin) {
    __shared__ float x[2 * DN], da[DN], dc[DN];
    const int s = blockIdx.x, t = threadIdx.x;
    int q = (int)charge[s] + max_charge, mi = (int)spin[s] - 1;
    q = q < 0 ? 0 : (q > 2 * max_charge ? 2 * max_charge : q);
    mi = mi < 0 ? 0 : (mi > max_spin - 1 ? max_spin - 1 : mi);
    x[t] = emb_q[(size_t)q * DN + t];
    x[DN + t] = emb_m[(size_t)mi * DN + t];
    dc[t] = dcond[(size_t)s * DN + t];
    __syncthreads();
    float a = b0[t];
    for (int k = 0; k < 2 * DN; k++) a = fmaf(w0[(size_t)t * 2 * DN + k], x[k], a);
    const float sg = 1.0f / (1.0f + expf(-a));
    float dh = 0.f;  // d hidden[t] = sum_o W2[o][t] dcond[o]
    for (int o = 0; o < DN; o++) dh = fmaf(w2[(size_t)o * DN + t], dc[o], dh);
    const float dat = dh * sg * (1.0f + a * (1.0f - sg));  // silu'(a)
    da[t] = dat;
    float* row = scr + (size_t)s * 5 * DN;
    row[t] = x[t]; row[DN + t] = x[DN + t];
    row[2 * DN + t] = a * sg;
    row[3 * DN + t] = dat;
    __syncthreads();
    float d0 = 0.f, d1 = 0.f;  // dx[k] = sum_o W0[o][k] da[o]
    for (int o = 0; o < DN; o++) {
        d0 = fmaf(w0[(size_t)o * 2 * DN + t], da[o], d0);
        d1 = fmaf(w0[(size_t)o * 2 * DN + DN + t], da[o], d1);
    }
    dx[(size_t)s * 2 * DN + t] = d0;
    dx[(size_t)s * 2 * DN + DN + t] = d1;
}
// one block per output row t of the two weight matrices: sums over the systems in order (deterministic)
__global__ __launch_bounds__(2 * DN) void k_cond_bwd_weights(const float* __restrict__ scr, const float* __restrict__ dcond,
                                                             int n_sys, float* __restrict__ gw0, float* __restrict__ gb0,
                                                             float* __restrict__ gw2, float* __restrict__ gb2) {
    const int t = blockIdx.x, k = threadIdx.x;  // k < 2 DN
    float a0 = 0.f, a2 = 0.f, s0 = 0.f, s2 = 0.f;
    for (int s = 0; s < n_sys; s++) {
        const float* row = scr + (size_t)s * 5 * DN;
        const float dat = row[3 * DN + t], dct = dcond[(size_t)s * DN + t];
        a0 = fmaf(dat, row[k], a0);
        if (k < DN) a2 = fmaf(dct, row[2 * DN + k], a2);
        s0 += dat; s2 += dct;
    }
    gw0[(size_t)t * 2 * DN + k] += a0;
    if (k < DN) gw2[(size_t)t * DN + k] += a2;
    if (k == 0) { gb0[t] += s0; gb2[t] += s2; }
}
// embedding rows: one block, the systems in order (several systems may share a charge or a multiplicity)
__global__ __launch_bounds__(DN) void k_cond_bwd_emb(const int64_t* __restrict__ charge, const int64_t* __restrict__ spin,
                                                     const float* __restrict__ dx, int n_sys, int max_charge, int max_spin,
                                                     float* __restrict__ gq, float* __restrict__ gm) {
    const int c = threadIdx.x;
    for (int s = 0; s < n_sys; s++) {
        int q = (int)charge[s] + max_charge, mi = (int)spin[s] - 1;
        q = q < 0 ? 0 : (q > 2 * max_charge ? 2 * max_charge : q);
        mi = mi < 0 ? 0 : (mi > max_spin - 1 ? max_spin - 1 : mi);
        gq[(size_t)q * DN + c] += dx[(size_t)s * 2 * DN + c];
        gm[(size_t)mi * DN + c] += dx[(size_t)s * 2 * DN + DN + c];
    }
}

void Trainer::cond_accumulate(const float* dHout, bool first) {
    if (!m.h.system_conditioning || err) return;
    if (!g.cond_charge || g.n_cond_systems < 1 || !w.dcond) { err = PET_ERR_ARGUMENT; set_error("system conditioning: no charges / multiplicities on this graph"); return; }
    k_cond_accum<<<(int)g.n_cond_systems, DN, 0, st>>>(dHout, g.sys, g.cond_sys, (int)g.n_nodes, w.dcond, first ? 0 : 1);
}
void Trainer::cond_finish() {
    if (!m.h.system_conditioning || err) return;
    const std::string sc = "system_conditioning.";
    float *gq = gp(sc + "charge_embedding.weight"), *gm = gp(sc + "spin_multiplicity_embedding.weight");
    float *gw0 = gp(sc + "project.0.weight"), *gb0 = gp(sc + "project.0.bias");
    float *gw2 = gp(sc + "project.2.weight"), *gb2 = gp(sc + "project.2.bias");
    if (!gq || !gm || !gw0 || !gb0 || !gw2 || !gb2) { err = PET_ERR_ARGUMENT; set_error("no gradient slot for the conditioning parameters"); return; }
    const int ns = (int)g.n_cond_systems;
    float* scr = w.partial;                    // [ns][5 DN]
    float* dx = w.partial + (size_t)ns * 5 * DN;  // [ns][2 DN]
    if ((size_t)ns * 7 * DN > w.partial_floats) { err = PET_ERR_UNSUPPORTED; set_error("too many systems for the conditioning scratch"); return; }
    k_cond_bwd_rows<<<ns, DN, 0, st>>>(g.cond_charge, g.cond_spin, m.cond_qe, m.cond_se, m.cond_w0, m.cond_b0, m.cond_w2,
                                       w.dcond, scr, dx, m.h.max_charge, m.h.max_spin_multiplicity);
    k_cond_bwd_weights<<<DN, 2 * DN, 0, st>>>(scr, w.dcond, ns, gw0, gb0, gw2, gb2);
    k_cond_bwd_emb<<<1, DN, 0, st>>>(g.cond_charge, g.cond_spin, dx, ns, m.h.max_charge, m.h.max_spin_multiplicity, gq, gm);
}

void Trainer::heads(bool edge, const float* Xin, int k_in, int64_t n_rows, const float* gA) {
    if (n_rows <= 0 || err) return;
    const std::string h = edge ? "edge_heads.@.0" : "node_heads.@.0";
    const std::string l = edge ? "edge_last_layers.@.0.@" : "node_last_layers.@.0.@";
    linear(h + ".0", DH, k_in, {w.hda1, nullptr, 0, DH}, {Xin, k_in, 0, nullptr, nullptr}, 0, n_rows);
    linear(h + ".2", DH, DH, {w.hda2, nullptr, 0, DH}, {w.hs1, DH, 0, nullptr, nullptr}, 0, n_rows);
    // last layer: d w = colsum(gy * s2), d b = sum gy
    const int nsplit = 256;
    k_colsum_partial<<<nsplit, 128, 0, st>>>(w.hs2y, n_rows, DH, w.partial);
    reduce_2d(w.partial, nsplit, DH, 1, gp(l + ".weight"), 1, 0, 1, st);
    k_edge_gy_sum_partial<<<nsplit, 256, 0, st>>>(gA, edge ? g.ctr : nullptr, g.fc, n_rows, w.partial);
    reduce_2d(w.partial, nsplit, 1, 1, gp(l + ".bias"), 1, 0, 1, st);
}

void Trainer::colsum(const float* buf, int64_t n_rows, int C, float* dst) {
    if (n_rows <= 0 || err) return;
    const int nsplit = 256;
    k_colsum_partial<<<nsplit, 256, 0, st>>>(buf, n_rows, C, w.partial);
    reduce_2d(w.partial, nsplit, C, 1, dst, 1, 0, 1, st);
}

void Trainer::vecsum(const float* vec, int64_t n_rows, float* dst) {
    if (n_rows <= 0 || err) return;
    const int nsplit = 256;
    k_edge_gy_sum_partial<<<nsplit, 256, 0, st>>>(vec, nullptr, nullptr, n_rows, w.partial);
    reduce_2d(w.partial, nsplit, 1, 1, dst, 1, 0, 1, st);
}

static void species_sum(Trainer& t, const float* buf, const int* idx, int64_t n_rows, int C, float* dst /*[ns,C]*/) {
    if (n_rows <= 0 || t.err) return;
    const int ns = t.m.h.n_species;
    const int nsplit = 2048;
    // the species axis in chunks of 64 KB / (4 C): 64 species for the node embedding (C = 256), 128 for the edge ones
    const int chunk = (64 * 1024) / (C * (int)sizeof(float));
    for (int s0 = 0; s0 < ns; s0 += chunk) {
        const int nc = ns - s0 < chunk ? ns - s0 : chunk;
        k_species_sum_partial<<<nsplit, 256, (size_t)nc * C * sizeof(float), t.st>>>(buf, idx, n_rows, C, nc, s0, t.w.partial);
        reduce_2d(t.w.partial, nsplit, nc * C, 1, dst + (size_t)s0 * C, 1, 0, 1, t.st);
    }
}

void Trainer::species_rows(const float* buf, const int* idx, int64_t n_rows, int C, float* dst) {
    species_sum(*this, buf, idx, n_rows, C, dst);
}

void Trainer::embeddings(const float* dH0, const float* dM0) {
    species_sum(*this, dH0, g.sp, g.n_nodes, DN, gp("node_embedders.0.weight"));
    // layer-0 messages are edge_embedder[species of the neighbour] (backend.py:516): residual path
    species_sum(*this, dM0, g.sp_nbr, g.n_edges, D, gp("edge_embedder.weight"));
}

// compress.0 was folded at load time (abi.hip finalize):
//   a0 = geo Wc^T + Tbl[species] (+ M W0c^T),  Wc = W0a Wee,  Tbl[s] = W0a bee + b0 + W0b emb[s]
// un-fold the gradients on the host in fp64 (a few hundred KB once per step).
void Trainer::compress0(int gi, const float* da0, const float* Min, const float* la0, const float4* Tgeo,
                        const float* TMin) {
    if (g.n_edges <= 0 || err) return;
    const int64_t E = g.n_edges;
    const int ns = m.h.n_species;
    const int kin = (gi == 0 ? 2 : 3) * D;
    const std::string pre = "gnn_layers." + std::to_string(gi);
    const int nsplit = 2048;
    // dWc [D,4]
    k_geo_wgrad_partial<<<nsplit, 128, 0, st>>>(da0, g.geo, E, w.partial);
    reduce_2d(w.partial, nsplit, D * 4, 1, w.gvec, 1, 0, 0, st);
    if (la0) {  // second-order pair: lambda_a0^T (d geo)
        k_geo_wgrad_partial<<<nsplit, 128, 0, st>>>(la0, Tgeo, E, w.partial);
        reduce_2d(w.partial, nsplit, D * 4, 1, w.gvec, 1, 0, 1, st);
    }
    // dTbl [ns, D] into gvec + 512
    float* dTbl_d = w.gvec + 512;
    TR_CHECK(hipMemsetAsync(dTbl_d, 0, (size_t)ns * D * sizeof(float), st));
    species_sum(*this, da0, g.sp_nbr, E, D, dTbl_d);
    if (gi > 0) {  // message block of compress.0: columns 2D..3D
        float* dW0 = gp(pre + ".compress.0.weight");
        wgrad_core(*this, D, D, {da0, nullptr, 0, D}, {Min, D, 0, nullptr, nullptr}, 0, E, dW0 + 2 * D, kin, nullptr, true);
        if (la0)
            wgrad_core(*this, D, D, {la0, nullptr, 0, D}, {TMin, D, 0, nullptr, nullptr}, 0, E, dW0 + 2 * D, kin, nullptr,
                       true);
    }
    std::vector<float> hWc(D * 4), hTbl((size_t)ns * D), hW0((size_t)D * kin), hWee(D * 4), hBee(D), hEmb((size_t)ns * D);
    const std::string emb_key = gi == 0 ? "edge_embedder.weight" : pre + ".neighbor_embedder.weight";
    auto raw = [&](const std::string& k) { return m.raw.at(k).first; };
    TR_CHECK(hipMemcpyAsync(hWc.data(), w.gvec, hWc.size() * 4, hipMemcpyDeviceToHost, st));
    TR_CHECK(hipMemcpyAsync(hTbl.data(), dTbl_d, hTbl.size() * 4, hipMemcpyDeviceToHost, st));
    TR_CHECK(hipMemcpyAsync(hW0.data(), raw(pre + ".compress.0.weight"), hW0.size() * 4, hipMemcpyDeviceToHost, st));
    TR_CHECK(hipMemcpyAsync(hWee.data(), raw(pre + ".edge_embedder.weight"), hWee.size() * 4, hipMemcpyDeviceToHost, st));
    TR_CHECK(hipMemcpyAsync(hBee.data(), raw(pre + ".edge_embedder.bias"), hBee.size() * 4, hipMemcpyDeviceToHost, st));
    TR_CHECK(hipMemcpyAsync(hEmb.data(), raw(emb_key), hEmb.size() * 4, hipMemcpyDeviceToHost, st));
    TR_CHECK(hipStreamSynchronize(st));
    if (err) return;
    std::vector<double> dbc(D, 0.0);
    for (int s = 0; s < ns; s++)
        for (int o = 0; o < D; o++) dbc[o] += hTbl[(size_t)s * D + o];
    std::vector<float> gW0a((size_t)D * D), gW0b((size_t)D * D), gWee(D * 4), gBee(D), gB0(D), gEmb((size_t)ns * D);
    for (int o = 0; o < D; o++) {
        gB0[o] = (float)dbc[o];
        for (int k = 0; k < D; k++) {
            double a = dbc[o] * hBee[k];  // d(W0a bee)/dW0a
            for (int c = 0; c < 4; c++) a += (double)hWc[o * 4 + c] * hWee[k * 4 + c];  // dWc Wee^T
            gW0a[(size_t)o * D + k] = (float)a;
            double b = 0.0;
            for (int s = 0; s < ns; s++) b += (double)hTbl[(size_t)s * D + o] * hEmb[(size_t)s * D + k];
            gW0b[(size_t)o * D + k] = (float)b;
        }
    }
    for (int k = 0; k < D; k++) {
        double bb = 0.0;
        for (int o = 0; o < D; o++) bb += (double)hW0[(size_t)o * kin + k] * dbc[o];
        gBee[k] = (float)bb;
        for (int c = 0; c < 4; c++) {
            double a = 0.0;
            for (int o = 0; o < D; o++) a += (double)hW0[(size_t)o * kin + k] * hWc[o * 4 + c];
            gWee[k * 4 + c] = (float)a;
        }
    }
    for (int s = 0; s < ns; s++)
        for (int k = 0; k < D; k++) {
            double a = 0.0;
            for (int o = 0; o < D; o++) a += (double)hW0[(size_t)o * kin + D + k] * hTbl[(size_t)s * D + o];
            gEmb[(size_t)s * D + k] = (float)a;
        }
    // upload into scratch and add into the flat gradient (device-side add keeps stream order)
    float* scratch = w.gmat;
    auto add_block = [&](const std::vector<float>& h, float* dst, int rows, int cols, int ld) {
        if (!dst) { err = PET_ERR_ARGUMENT; return; }
        TR_CHECK(hipMemcpyAsync(scratch, h.data(), h.size() * 4, hipMemcpyHostToDevice, st));
        reduce_2d(scratch, 1, rows, cols, dst, ld, 0, 1, st);
        TR_CHECK(hipStreamSynchronize(st));  // scratch / host vector reuse
    };
    float* dW0 = gp(pre + ".compress.0.weight");
    add_block(gW0a, dW0, D, D, kin);
    add_block(gW0b, dW0 ? dW0 + D : nullptr, D, D, kin);
    add_block(gB0, gp(pre + ".compress.0.bias"), D, 1, 1);
    add_block(gWee, gp(pre + ".edge_embedder.weight"), D, 4, 4);
    add_block(gBee, gp(pre + ".edge_embedder.bias"), D, 1, 1);
    add_block(gEmb, gp(emb_key), ns, D, D);
}

}  // namespace pet
