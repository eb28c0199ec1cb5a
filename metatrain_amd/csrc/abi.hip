// extern "C" entry points of libpet_hip.so (see include/pet_hip.h) + model packing.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"
#include "model.h"

namespace pet {

static thread_local std::string g_error;

static int g_trr = -1;
bool use_trr() {
    if (g_trr < 0) {
        const char* e = getenv("PET_HIP_TRR");
        g_trr = (e && e[0] == '0') ? 0 : 1;
    }
    return g_trr == 1;
}
void set_use_trr(int v) { g_trr = v ? 1 : 0; }
static int g_side_override = -1;  // -1: environment default
void set_side_stream(int v) { g_side_override = v ? 1 : 0; }

void set_error(const std::string& msg) { g_error = msg; }

const SideStream& side_stream() {
    static SideStream ss;
    static SideStream off;  // enabled == false: everything on the caller's stream
    static bool init = false;
    if (g_side_override == 0) return off;
    if (!init) {
        init = true;
        const char* e = getenv("PET_HIP_SIDE");
        if (!(e && e[0] == '0')) {
            // The node chain runs BELOW the caller's stream: its result is needed one edge-MLP later, and at equal priority
            // its 133 KB workgroups take whole CUs from the edge kernels they overlap with (measured: 51.0 -> 49.7 ms per
            // step). PET_HIP_SIDE_PRIO = same | high overrides.
            const char* pr = getenv("PET_HIP_SIDE_PRIO");
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            const int prio = pr && pr[0] == 's' ? 0 : (pr && pr[0] == 'h' ? greatest : least);
            if (hipStreamCreateWithPriority(&ss.s, hipStreamNonBlocking, prio) == hipSuccess &&
                hipEventCreateWithFlags(&ss.to_side, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&ss.to_main, hipEventDisableTiming) == hipSuccess)
                ss.enabled = true;
        }
    }
    return ss;
}
void SideStream::fork(hipStream_t main) const {
    if (!enabled) return;
    (void)hipEventRecord(to_side, main);
    (void)hipStreamWaitEvent(s, to_side, 0);
}
void SideStream::join(hipStream_t main) const {
    if (!enabled) return;
    (void)hipEventRecord(to_main, s);
    (void)hipStreamWaitEvent(main, to_main, 0);
}

// ---------------------------------------------------------------------------------
// profiling: HIP events on the launch stream around every stage
// ---------------------------------------------------------------------------------
struct ProfRec {
    std::string name;
    hipEvent_t e0, e1;
    double flops, bytes;
};
static bool g_prof_on = false;
static std::string g_prof_filter;  // empty = every stage
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;

ProfScope::ProfScope(const char* n, hipStream_t s, double f, double b) : name(n), st(s), flops(f), bytes(b) {
    if (!g_prof_on || (!g_prof_filter.empty() && g_prof_filter != n)) return;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // The start stamp must not be taken while the previous kernel of the stream is still draining (a marker is stamped
    // when the command processor reaches it: measured 2.22 ms for a kernel whose rocprofv3 duration is 1.82 ms): make the
    // stream wait for everything before the scope first, then stamp.
    hipEvent_t fence = nullptr;
    if (hipEventCreateWithFlags(&fence, hipEventDisableTiming) == hipSuccess) {
        (void)hipEventRecord(fence, st);
        (void)hipStreamWaitEvent(st, fence, 0);
        (void)hipEventDestroy(fence);
    }
    (void)hipEventRecord(e0, st);
}
ProfScope::~ProfScope() {
    if (!e0) return;
    (void)hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({name, e0, e1, flops, bytes});
}

// two-way fp16 split of the same fragments (trr.h, f16x3 GEMMs): out[plane][(t * kbn + kb) * 64 + l][8]
__global__ void k_pack2h(const float* __restrict__ W, int64_t s_n, int64_t s_k, int n_out, int k_in,
                         _Float16* __restrict__ out, float lscale = 2048.0f, float hscale = 1.0f) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int kbn = k_in / 16;
    int64_t total = (int64_t)(n_out / 32) * kbn * 64;
    if (idx >= total) return;
    int l = idx & 63;
    int kb = (idx >> 6) % kbn;
    int t = (int)((idx >> 6) / kbn);
    int64_t n = t * 32 + (l & 31);
    for (int j = 0; j < 8; j++) {
        const int64_t k = kb * 16 + (j < 4 ? 4 * (l >> 5) + j : 8 + 4 * (l >> 5) + (j - 4));
        const float x = W[n * s_n + k * s_k];
        const _Float16 h = (_Float16)x;
        out[(0 * total + idx) * 8 + j] = (_Float16)((float)h * hscale);
        out[(1 * total + idx) * 8 + j] = (_Float16)((x - (float)h) * lscale);
    }
}

// ---------------------------------------------------------------------------------
// weight packing into MFMA fragment order (tile.h)
// ---------------------------------------------------------------------------------
__global__ void k_pack(const float* __restrict__ W, int64_t s_n, int64_t s_k, int n_out, int k_in,
                       float4* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int kgn = k_in / 8;
    int64_t total = (int64_t)(n_out / 32) * kgn * 64;
    if (idx >= total) return;
    int l = idx & 63;
    int kg = (idx >> 6) % kgn;
    int t = (int)((idx >> 6) / kgn);
    int64_t n = t * 32 + (l & 31), k = kg * 8 + (l >> 5) * 4;
    const float* p = W + n * s_n + k * s_k;
    out[idx] = make_float4(p[0], p[s_k], p[2 * s_k], p[3 * s_k]);
}

int dev_alloc(Model& m, void** p, size_t bytes) {
    PET_HIP_CHECK(hipMalloc(p, bytes > 0 ? bytes : 4));
    m.owned.push_back(*p);
    return PET_OK;
}

// derived buffers (packed weights, folded tables) are allocated once per name, so that
// pet_model_finalize can run after every optimizer step without growing the footprint
static int named_alloc(Model& m, const std::string& name, void** p, size_t bytes) {
    auto it = m.named.find(name);
    if (it != m.named.end() && it->second.second == bytes) {
        *p = it->second.first;
        return PET_OK;
    }
    int rc = dev_alloc(m, p, bytes);
    if (rc) return rc;
    m.named[name] = {*p, bytes};
    return PET_OK;
}

// pack the [n_out, k_in] sub-matrix starting at column col0 of a row-major matrix with
// leading dimension ld
static int pack_lin(Model& m, const std::string& name, Lin& L, const float* w, const float* b, int n_out,
                    int k_in, int ld, int col0, hipStream_t st) {
    L.w = w + col0;
    L.b = b;
    L.n_out = n_out;
    L.k_in = k_in;
    if (m.generic()) return PET_OK;  // gen.hip reads the raw torch layout; no MFMA fragment forms
    PET_REQUIRE(n_out % 32 == 0 && k_in % 32 == 0, PET_ERR_UNSUPPORTED, "linear shape not tileable");
    size_t n4 = (size_t)(n_out / 32) * (k_in / 8) * 64;
    int rc;
    if ((rc = named_alloc(m, name + ":fwd", (void**)&L.fwd, n4 * sizeof(float4))) != PET_OK) return rc;
    if ((rc = named_alloc(m, name + ":bwd", (void**)&L.bwd, n4 * sizeof(float4))) != PET_OK) return rc;
    k_pack<<<cdiv(n4, 256), 256, 0, st>>>(w + col0, ld, 1, n_out, k_in, L.fwd);
    // transposed operand: rows = original columns, k = original rows
    k_pack<<<cdiv(n4, 256), 256, 0, st>>>(w + col0, 1, ld, k_in, n_out, L.bwd);
    {  // f16x3 operand planes: fwd tiles over n_out (K = k_in), bwd over k_in (K = n_out)
        const size_t n8 = (size_t)(n_out / 32) * (k_in / 16) * 64;  // == (k_in / 32) * (n_out / 16) * 64
        if ((rc = named_alloc(m, name + ":fwd2", &L.fwd2, 2 * n8 * 16)) != PET_OK) return rc;
        if ((rc = named_alloc(m, name + ":bwd2", &L.bwd2, 2 * n8 * 16)) != PET_OK) return rc;
        k_pack2h<<<cdiv(n8, 256), 256, 0, st>>>(w + col0, ld, 1, n_out, k_in, (_Float16*)L.fwd2);
        k_pack2h<<<cdiv(n8, 256), 256, 0, st>>>(w + col0, 1, ld, k_in, n_out, (_Float16*)L.bwd2);
    }
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// W diag(gamma) and b + W beta of a Linear behind a norm (beta == nullptr: RMSNorm), fp64 row sums
__global__ void k_fold_norm(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ gamma,
                            const float* __restrict__ beta, int n_out, int k_in, float* __restrict__ Wg, float* __restrict__ bg) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_out) return;
    double acc = b[n];
    for (int k = 0; k < k_in; k++) {
        const float w = W[(size_t)n * k_in + k];
        Wg[(size_t)n * k_in + k] = w * gamma[k];
        if (beta) acc += (double)w * (double)beta[k];
    }
    bg[n] = (float)acc;
}
static int pack_lin_s(Model& m, const std::string& name, Lin& L, hipStream_t st, int ld = 0);
static int fold_norm_s(Model& m, const std::string& name, const Lin& src, const float* gamma, const float* beta, Lin& out,
                       hipStream_t st) {
    if (m.generic()) return PET_OK;
    float *wg = nullptr, *bg = nullptr;
    int rc;
    if ((rc = named_alloc(m, name + ":wg", (void**)&wg, (size_t)src.n_out * src.k_in * sizeof(float))) != PET_OK) return rc;
    if ((rc = named_alloc(m, name + ":bg", (void**)&bg, (size_t)src.n_out * sizeof(float))) != PET_OK) return rc;
    k_fold_norm<<<cdiv(src.n_out, 128), 128, 0, st>>>(src.w, src.b, gamma, beta, src.n_out, src.k_in, wg, bg);
    out = Lin();
    out.w = wg; out.b = bg; out.n_out = src.n_out; out.k_in = src.k_in;
    return pack_lin_s(m, name + ":g", out, st);
}

// "scaled" planes for the single-accumulator products of pet_ablk.hip: H = fp16(64 w), L = fp16(64 w - H)
// (ld: leading dimension of L.w when it is a column block of a wider matrix; 0 = k_in)
static int pack_lin_s(Model& m, const std::string& name, Lin& L, hipStream_t st, int ld) {
    if (m.generic()) return PET_OK;
    if (ld == 0) ld = L.k_in;
    const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
    int rc;
    if ((rc = named_alloc(m, name + ":fwd2s", &L.fwd2s, 2 * n8 * 16)) != PET_OK) return rc;
    if ((rc = named_alloc(m, name + ":bwd2s", &L.bwd2s, 2 * n8 * 16)) != PET_OK) return rc;
    k_pack2h<<<cdiv(n8, 256), 256, 0, st>>>(L.w, ld, 1, L.n_out, L.k_in, (_Float16*)L.fwd2s, 64.0f, 64.0f);
    k_pack2h<<<cdiv(n8, 256), 256, 0, st>>>(L.w, 1, ld, L.k_in, L.n_out, (_Float16*)L.bwd2s, 64.0f, 64.0f);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

static int get(const Model& m, const std::string& key, int64_t numel, const float** out) {
    auto it = m.raw.find(key);
    PET_REQUIRE(it != m.raw.end(), PET_ERR_ARGUMENT, "missing parameter '" + key + "'");
    PET_REQUIRE(it->second.second == numel, PET_ERR_ARGUMENT,
                "parameter '" + key + "' has " + std::to_string(it->second.second) + " elements, expected " +
                    std::to_string(numel));
    *out = it->second.first;
    return PET_OK;
}

static int get_lin(Model& m, const std::string& key, int n_out, int k_in, Lin& L, hipStream_t st) {
    const float *w, *b;
    int rc;
    if ((rc = get(m, key + ".weight", (int64_t)n_out * k_in, &w)) != PET_OK) return rc;
    if ((rc = get(m, key + ".bias", n_out, &b)) != PET_OK) return rc;
    return pack_lin(m, key, L, w, b, n_out, k_in, k_in, 0, st);
}

// compress.0 folded with the 4 -> D edge embedder and the species embeddings, accumulated in fp64:
//   Wc[o][c] = sum_k W0[o][k] Wee[k][c],  Tbl[s][o] = b0[o] + sum_k W0[o][k] bee[k] + sum_k W0[o][D+k] emb[s][k]
__global__ void k_fold_compress0(const float* __restrict__ W0, int kin, const float* __restrict__ b0,
                                 const float* __restrict__ Wee, const float* __restrict__ bee,
                                 const float* __restrict__ emb, int ns, float* __restrict__ Wc,
                                 float* __restrict__ Wct, float* __restrict__ Tbl) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < D * 4) {
        const int o = idx >> 2, c = idx & 3;
        double s = 0.0;
        for (int k = 0; k < D; k++) s += (double)W0[(size_t)o * kin + k] * (double)Wee[k * 4 + c];
        Wc[o * 4 + c] = (float)s;
        Wct[c * D + o] = (float)s;
    } else if (idx < D * 4 + ns * D) {
        const int t = idx - D * 4, sidx = t / D, o = t % D;
        double s = (double)b0[o];
        for (int k = 0; k < D; k++) s += (double)W0[(size_t)o * kin + k] * (double)bee[k];
        for (int k = 0; k < D; k++) s += (double)W0[(size_t)o * kin + D + k] * (double)emb[(size_t)sidx * D + k];
        Tbl[(size_t)sidx * D + o] = (float)s;
    }
}

int finalize(Model& m, hipStream_t st) {
    const pet_hypers_t& h = m.h;
    const int ns = h.n_species;
    int rc;
    // the model's own sizes (they shadow the compiled instantiation's constants, which only the tuned kernels use)
    const int D = h.d_pet, DN = h.d_node, DFF = h.d_feedforward, DNF = 2 * h.d_node, DH = h.d_head;
    const bool generic = m.generic(), expanded = DN != D;
    m.gnn.clear();
    m.gnn.resize(h.num_gnn_layers);
    for (int g = 0; g < h.num_gnn_layers; g++) {
        GnnLayerW& G = m.gnn[g];
        const std::string pre = "gnn_layers." + std::to_string(g);
        G.attn.resize(h.num_attention_layers);
        for (int a = 0; a < h.num_attention_layers; a++) {
            AttnLayerW& A = G.attn[a];
            const std::string lp = pre + ".trans.layers." + std::to_string(a);
            if ((rc = get_lin(m, lp + ".attention.input_linear", 3 * D, D, A.qkv, st))) return rc;
            if ((rc = get_lin(m, lp + ".attention.output_linear", D, D, A.out, st))) return rc;
            if ((rc = pack_lin_s(m, lp + ".attention.input_linear", A.qkv, st))) return rc;
            if ((rc = pack_lin_s(m, lp + ".attention.output_linear", A.out, st))) return rc;
            if ((rc = get(m, lp + ".norm_attention.weight", D, &A.g_attn))) return rc;
            if ((rc = get(m, lp + ".norm_mlp.weight", D, &A.g_mlp))) return rc;
            if ((rc = get_lin(m, lp + ".mlp.w_in", 2 * DFF, D, A.mlp_in, st))) return rc;
            if ((rc = get_lin(m, lp + ".mlp.w_out", D, DFF, A.mlp_out, st))) return rc;
            if ((rc = pack_lin_s(m, lp + ".mlp.w_in", A.mlp_in, st))) return rc;   // k_emlp_s (pet_emlp_s.hip)
            if ((rc = pack_lin_s(m, lp + ".mlp.w_out", A.mlp_out, st))) return rc;
            if (m.layer_norm()) {  // torch.nn.LayerNorm: weight + bias (transformer.py:170-176)
                if ((rc = get(m, lp + ".norm_attention.bias", D, &A.b_attn))) return rc;
                if ((rc = get(m, lp + ".norm_mlp.bias", D, &A.b_mlp))) return rc;
            }
            if ((rc = fold_norm_s(m, lp + ".mlp.w_in", A.mlp_in, A.g_mlp, m.layer_norm() ? A.b_mlp : nullptr, A.mlp_in_g, st))) return rc;
            if ((rc = fold_norm_s(m, lp + ".attention.input_linear", A.qkv, A.g_attn, m.layer_norm() ? A.b_attn : nullptr, A.qkv_g, st))) return rc;
            if (!expanded) continue;  // d_node == d_pet: Identity modules, no parameters (transformer.py:196-201)
            if ((rc = get_lin(m, lp + ".center_contraction", D, DN, A.cc, st))) return rc;
            if ((rc = get_lin(m, lp + ".center_expansion", DN, D, A.ce, st))) return rc;
            if ((rc = pack_lin_s(m, lp + ".center_contraction", A.cc, st))) return rc;  // k_rowlin_s (pet_center_s.hip)
            if ((rc = pack_lin_s(m, lp + ".center_expansion", A.ce, st))) return rc;
            if ((rc = get(m, lp + ".norm_center_features.weight", DN, &A.g_center))) return rc;
            if (m.layer_norm() && (rc = get(m, lp + ".norm_center_features.bias", DN, &A.b_center))) return rc;
            if ((rc = get_lin(m, lp + ".center_mlp.w_in", 2 * DNF, DN, A.cmlp_in, st))) return rc;
            if ((rc = get_lin(m, lp + ".center_mlp.w_out", DN, DNF, A.cmlp_out, st))) return rc;
            if ((rc = pack_lin_s(m, lp + ".center_mlp.w_in", A.cmlp_in, st))) return rc;  // k_rowgemm_s (so_rows_s.hip)
            if ((rc = pack_lin_s(m, lp + ".center_mlp.w_out", A.cmlp_out, st))) return rc;
        }
        // ---- compress.0 decomposition (transformer.py:499-521) -------------------
        // tokens = [edge_embedder([v, d]) ; (g>0: neighbor_embedder[species]) ; message]
        // compress.0 is linear in each block, so
        //   a0 = [v,d] (W0a Wee)^T + (W0a bee + b0) + W0b emb[species] (+ W0c message, g>0)
        // where for g == 0 the "message" block IS an embedding lookup (backend.py:516).
        const int kin = (g == 0 ? 2 : 3) * D;
        const float *w0, *b0, *wee, *bee, *emb;
        if ((rc = get(m, pre + ".compress.0.weight", (int64_t)D * kin, &w0))) return rc;
        if ((rc = get(m, pre + ".compress.0.bias", D, &b0))) return rc;
        if ((rc = get(m, pre + ".edge_embedder.weight", D * 4, &wee))) return rc;
        if ((rc = get(m, pre + ".edge_embedder.bias", D, &bee))) return rc;
        if (g == 0) {
            if ((rc = get(m, "edge_embedder.weight", (int64_t)ns * D, &emb))) return rc;
        } else {
            if ((rc = get(m, pre + ".neighbor_embedder.weight", (int64_t)ns * D, &emb))) return rc;
        }
        G.eemb.w = wee; G.eemb.b = bee; G.eemb.n_out = D; G.eemb.k_in = 4;
        G.c0.w = w0; G.c0.b = b0; G.c0.n_out = D; G.c0.k_in = kin;
        G.nbr_emb = g == 0 ? nullptr : emb;
        if (generic) {   // no folded / packed forms: gen.hip evaluates edge_embedder and compress.0 as uploaded
            if ((rc = get_lin(m, pre + ".compress.2", D, D, G.compress2, st))) return rc;
            if (m.residual()) continue;
            const std::string gs_ = std::to_string(g);
            if ((rc = get(m, "combination_norms." + gs_ + ".weight", 2 * D, &G.ln_g))) return rc;
            if ((rc = get(m, "combination_norms." + gs_ + ".bias", 2 * D, &G.ln_b))) return rc;
            if ((rc = get_lin(m, "combination_mlps." + gs_ + ".0", 2 * D, 2 * D, G.comb0, st))) return rc;
            if ((rc = get_lin(m, "combination_mlps." + gs_ + ".2", D, 2 * D, G.comb2, st))) return rc;
            continue;
        }
        if ((rc = named_alloc(m, pre + ":wc", (void**)&G.wc, D * 4 * sizeof(float)))) return rc;
        if ((rc = named_alloc(m, pre + ":wct", (void**)&G.wct, 4 * D * sizeof(float)))) return rc;
        if ((rc = named_alloc(m, pre + ":tbl", (void**)&G.tbl, (size_t)ns * D * sizeof(float)))) return rc;
        k_fold_compress0<<<cdiv(D * 4 + ns * D, 128), 128, 0, st>>>(w0, kin, b0, wee, bee, emb, ns, G.wc, G.wct, G.tbl);
        {   // the 4 -> D composite once more as one 32-row MFMA tile (rows 4..31 zero), f16x3 planes, for the TRR adjoint
            const size_t n8 = (size_t)(D / 16) * 64;
            if ((rc = named_alloc(m, pre + ":wcp", (void**)&G.wcp, 32 * D * sizeof(float)))) return rc;
            if ((rc = named_alloc(m, pre + ":wc2", &G.wc2, 2 * n8 * 16))) return rc;
            PET_HIP_CHECK(hipMemsetAsync(G.wcp, 0, 32 * D * sizeof(float), st));
            PET_HIP_CHECK(hipMemcpyAsync(G.wcp, G.wct, 4 * D * sizeof(float), hipMemcpyDeviceToDevice, st));
            k_pack2h<<<cdiv(n8, 256), 256, 0, st>>>(G.wcp, D, 1, 32, D, (_Float16*)G.wc2);
            if ((rc = named_alloc(m, pre + ":wc2s", &G.wc2s, 2 * n8 * 16))) return rc;  // k_compress_bwd_s (pet_compress_s.hip)
            k_pack2h<<<cdiv(n8, 256), 256, 0, st>>>(G.wcp, D, 1, 32, D, (_Float16*)G.wc2s, 64.0f, 64.0f);
        }
        if (g > 0) {
            if ((rc = pack_lin(m, pre + ".compress.0:msg", G.compress0_msg, w0, nullptr, D, D, kin, 2 * D, st))) return rc;
            if ((rc = pack_lin_s(m, pre + ".compress.0:msg", G.compress0_msg, st, kin))) return rc;  // k_compress_s / _bwd_s
        }
        if ((rc = get_lin(m, pre + ".compress.2", D, D, G.compress2, st))) return rc;
        if ((rc = pack_lin_s(m, pre + ".compress.2", G.compress2, st))) return rc;
        const std::string gs = std::to_string(g);
        if (m.residual()) continue;  // backend.py:589-649: no combination modules, messages are averaged
        if ((rc = get(m, "combination_norms." + gs + ".weight", 2 * D, &G.ln_g))) return rc;
        if ((rc = get(m, "combination_norms." + gs + ".bias", 2 * D, &G.ln_b))) return rc;
        if ((rc = get_lin(m, "combination_mlps." + gs + ".0", 2 * D, 2 * D, G.comb0, st))) return rc;
        if ((rc = get_lin(m, "combination_mlps." + gs + ".2", D, 2 * D, G.comb2, st))) return rc;
        if ((rc = fold_norm_s(m, "combination_mlps." + gs + ".0", G.comb0, G.ln_g, G.ln_b, G.comb0_g, st))) return rc;  // k_comb_s
        if ((rc = pack_lin_s(m, "combination_mlps." + gs + ".2", G.comb2, st))) return rc;
        if ((rc = pack_lin_s(m, "combination_mlps." + gs + ".0", G.comb0, st))) return rc;  // k_rowgemm_s (so_rows_s.hip)
    }
    m.node_embs.assign(m.residual() ? h.num_gnn_layers : 1, nullptr);  // backend.py:93-119: one per readout layer
    for (size_t l = 0; l < m.node_embs.size(); l++)
        if ((rc = get(m, "node_embedders." + std::to_string(l) + ".weight", (int64_t)ns * DN, &m.node_embs[l]))) return rc;
    m.node_emb = m.node_embs[0];
    if ((rc = get(m, "edge_embedder.weight", (int64_t)ns * D, &m.edge_emb))) return rc;
    if (h.system_conditioning) {
        const std::string sc = "system_conditioning.";
        if ((rc = get(m, sc + "charge_embedding.weight", (int64_t)(2 * h.max_charge + 1) * DN, &m.cond_qe))) return rc;
        if ((rc = get(m, sc + "spin_multiplicity_embedding.weight", (int64_t)h.max_spin_multiplicity * DN, &m.cond_se))) return rc;
        if ((rc = get(m, sc + "project.0.weight", (int64_t)DN * 2 * DN, &m.cond_w0))) return rc;
        if ((rc = get(m, sc + "project.0.bias", DN, &m.cond_b0))) return rc;
        if ((rc = get(m, sc + "project.2.weight", (int64_t)DN * DN, &m.cond_w2))) return rc;
        if ((rc = get(m, sc + "project.2.bias", DN, &m.cond_b2))) return rc;
    }
    // every head that was uploaded: "node_heads.<t>.<l>.0.weight" names a (target, readout layer); "node_last_layers.
    // <t>.<l>.<block>.weight" a block of P = numel / DH properties (backend.py:171-217). "@" is the fused target.
    m.heads.clear();
    m.lasts.clear();
    std::vector<std::pair<std::string, std::string>> head_ids, last_ids;
    for (const auto& kv : m.raw) {
        const std::string& k = kv.first;
        auto field = [&](int i) {  // i-th dot-separated field
            size_t a = 0;
            for (int n = 0; n < i; n++) a = k.find('.', a) + 1;
            return k.substr(a, k.find('.', a) - a);
        };
        if (k.rfind("node_heads.", 0) == 0 && k.size() > 9 && k.compare(k.size() - 9, 9, ".0.weight") == 0)
            head_ids.push_back({field(1), field(2)});
        if (k.rfind("node_last_layers.", 0) == 0 && k.compare(k.size() - 7, 7, ".weight") == 0) {
            // the block name may itself contain dots: everything between the layer field and ".weight"
            const size_t a = std::string("node_last_layers.").size() + field(1).size() + 1 + field(2).size() + 1;
            last_ids.push_back({field(1) + "." + field(2), k.substr(a, k.size() - 7 - a)});
        }
    }
    for (const auto& id : head_ids) {
        HeadW& H = m.heads[id.first + "|" + id.second];
        const std::string tl = id.first + "." + id.second;
        if ((rc = get_lin(m, "node_heads." + tl + ".0", DH, DN, H.nh0, st))) return rc;
        if ((rc = get_lin(m, "node_heads." + tl + ".2", DH, DH, H.nh2, st))) return rc;
        if ((rc = get_lin(m, "edge_heads." + tl + ".0", DH, D, H.eh0, st))) return rc;
        if ((rc = get_lin(m, "edge_heads." + tl + ".2", DH, DH, H.eh2, st))) return rc;
        if ((rc = pack_lin_s(m, "edge_heads." + tl + ".0", H.eh0, st))) return rc;  // k_head_s / k_head_bwd_s (pet_head_s.hip)
        if ((rc = pack_lin_s(m, "edge_heads." + tl + ".2", H.eh2, st))) return rc;
    }
    for (const auto& id : last_ids) {
        const std::string key = id.first + "." + id.second;  // <t>.<l>.<block>
        LastW Lw;
        const auto it = m.raw.find("node_last_layers." + key + ".weight");
        Lw.P = (int)(it->second.second / DH);
        PET_REQUIRE(Lw.P >= 1 && (int64_t)Lw.P * DH == it->second.second, PET_ERR_ARGUMENT, "bad last-layer shape: " + key);
        if ((rc = get(m, "node_last_layers." + key + ".weight", (int64_t)Lw.P * DH, &Lw.nw))) return rc;
        if ((rc = get(m, "node_last_layers." + key + ".bias", Lw.P, &Lw.nb))) return rc;
        if ((rc = get(m, "edge_last_layers." + key + ".weight", (int64_t)Lw.P * DH, &Lw.ew))) return rc;
        if ((rc = get(m, "edge_last_layers." + key + ".bias", Lw.P, &Lw.eb))) return rc;
        std::string tl = id.first;
        tl[tl.rfind('.')] = '|';
        m.lasts[tl + "|" + id.second] = Lw;
    }
    m.has_fused_head = m.heads.count("@|0") && m.lasts.count("@|0|@") && m.lasts["@|0|@"].P == 1;
    if (m.has_fused_head) {
        const HeadW& H = m.heads["@|0"];
        const LastW& Lw = m.lasts["@|0|@"];
        m.nh0 = H.nh0; m.nh2 = H.nh2; m.eh0 = H.eh0; m.eh2 = H.eh2;
        m.nll_w = Lw.nw; m.ell_w = Lw.ew;
        PET_HIP_CHECK(hipMemcpyAsync(&m.nll_b, Lw.nb, sizeof(float), hipMemcpyDeviceToHost, st));
        PET_HIP_CHECK(hipMemcpyAsync(&m.ell_b, Lw.eb, sizeof(float), hipMemcpyDeviceToHost, st));
        PET_HIP_CHECK(hipStreamSynchronize(st));
    }
    PET_REQUIRE(m.species_table != nullptr, PET_ERR_ARGUMENT, "missing species_to_species_index");
    m.finalized = true;
    return PET_OK;
}

__global__ void k_i64_to_i32(const int64_t* __restrict__ in, int* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)in[i];
}

}  // namespace pet

using namespace pet;

struct pet_model {
    Model m;
};

extern "C" {

const char* pet_last_error(void) { return g_error.c_str(); }
const char* pet_version(void) { return "pet_hip 0.1 (gfx950, fp32 MFMA)"; }

// any size is served: the compiled instantiation (d_pet=128, d_node=256, d_feedforward=256, d_head=128, num_heads=8) by the
// tuned kernels, everything else by the size-generic path (gen.hip; head dimension up to 128)
int pet_hypers_supported(const pet_hypers_t* h) {
    return h && h->d_pet >= 1 && h->d_node >= 1 && h->d_feedforward >= 1 && h->d_head >= 1 && h->num_heads >= 1 &&
           h->d_pet % h->num_heads == 0 && h->d_pet / h->num_heads <= 128 && h->num_gnn_layers >= 1 &&
           h->num_attention_layers >= 1 && h->n_species >= 1 && h->n_species <= MAX_SPECIES;
}

int pet_model_create(const pet_hypers_t* h, pet_model_t** out) {
    PET_REQUIRE(h && out, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pet_hypers_supported(h), PET_ERR_UNSUPPORTED,
                "unsupported sizes: d_pet must be a multiple of num_heads with a head dimension of at most 128, at most "
                "128 species");
    PET_REQUIRE(h->cutoff_function == PET_CUTOFF_BUMP || h->cutoff_function == PET_CUTOFF_COSINE,
                PET_ERR_UNSUPPORTED, "unknown cutoff function");
    PET_REQUIRE((h->normalization == PET_NORM_RMS || h->normalization == PET_NORM_LAYER) &&
                    (h->transformer_type == PET_PRE_LN || h->transformer_type == PET_POST_LN) &&
                    (h->featurizer_type == PET_FEATURIZER_FEEDFORWARD || h->featurizer_type == PET_FEATURIZER_RESIDUAL),
                PET_ERR_UNSUPPORTED, "unknown normalization / transformer_type / featurizer_type");
    PET_REQUIRE(h->adaptive_cutoff_method == PET_ADAPTIVE_SOLVER || h->adaptive_cutoff_method == PET_ADAPTIVE_GRID,
                PET_ERR_UNSUPPORTED, "unknown adaptive_cutoff_method");
    PET_REQUIRE(!h->system_conditioning || (h->max_charge >= 0 && h->max_spin_multiplicity >= 1), PET_ERR_ARGUMENT,
                "system_conditioning needs max_charge >= 0 and max_spin_multiplicity >= 1");
    pet_model_t* pm = new pet_model_t();
    pm->m.h = *h;
    *out = pm;
    return PET_OK;
}

void pet_model_destroy(pet_model_t* pm) {
    if (!pm) return;
    for (void* p : pm->m.owned) (void)hipFree(p);
    delete pm;
}

int pet_model_set_param(pet_model_t* pm, const char* key, const void* d_data, int64_t numel, void* stream) {
    PET_REQUIRE(pm && key && d_data && numel > 0, PET_ERR_ARGUMENT, "bad argument");
    Model& m = pm->m;
    hipStream_t st = (hipStream_t)stream;
    std::string k(key);
    if (k == "species_to_species_index") {
        int* p = m.species_table;
        if (!p || m.species_table_len != (int)numel) {  // re-uploads (one per optimizer step through the mirror) reuse it
            int rc = dev_alloc(m, (void**)&p, numel * sizeof(int));
            if (rc) return rc;
        }
        k_i64_to_i32<<<cdiv(numel, 256), 256, 0, st>>>((const int64_t*)d_data, p, (int)numel);
        PET_HIP_CHECK(hipGetLastError());
        m.species_table = p;
        m.species_table_len = (int)numel;
        return PET_OK;
    }
    float* p;
    auto it = m.raw.find(k);
    if (it != m.raw.end() && it->second.second == numel) {
        p = it->second.first;  // overwrite in place (weights updated by an optimizer step)
    } else {
        // the flat gradient / Adam buffers and their segment table are laid out by upload order and size: a parameter
        // cannot change its size once it has a slot
        PET_REQUIRE(it == m.raw.end(), PET_ERR_ARGUMENT,
                    "parameter '" + k + "' was uploaded with " + std::to_string(it->second.second) +
                        " elements before and cannot be re-set with " + std::to_string(numel) + ": create a new model");
        int rc = dev_alloc(m, (void**)&p, numel * sizeof(float));
        if (rc) return rc;
        m.grad_off[k] = m.n_params;
        m.n_params += numel;
        m.raw[k] = {p, numel};
    }
    PET_HIP_CHECK(hipMemcpyAsync(p, d_data, numel * sizeof(float), hipMemcpyDeviceToDevice, st));
    m.finalized = false;
    return PET_OK;
}

int pet_model_tie_halves(pet_model_t* pm, const char* key) {
    PET_REQUIRE(pm && key, PET_ERR_ARGUMENT, "null argument");
    return tie_halves(pm->m, key);
}

int pet_optimizer_state(pet_model_t* pm, float* d_m, float* d_v, int64_t numel, int direction, void* stream) {
    PET_REQUIRE(pm && d_m && d_v, PET_ERR_ARGUMENT, "null argument");
    return optimizer_state(pm->m, d_m, d_v, numel, direction, (hipStream_t)stream);
}

int pet_model_finalize(pet_model_t* pm, void* stream) {
    PET_REQUIRE(pm, PET_ERR_ARGUMENT, "null model");
    return finalize(pm->m, (hipStream_t)stream);
}

int64_t pet_model_num_params(const pet_model_t* pm) { return pm ? pm->m.n_params : 0; }

int pet_model_zero_grad(pet_model_t* pm, void* stream) {
    PET_REQUIRE(pm, PET_ERR_ARGUMENT, "null model");
    Model& m = pm->m;
    PET_REQUIRE(m.n_params > 0, PET_ERR_ARGUMENT, "model has no parameters");
    if (!m.grad_flat) {
        int rc = dev_alloc(m, (void**)&m.grad_flat, m.n_params * sizeof(float));
        if (rc) return rc;
    }
    PET_HIP_CHECK(hipMemsetAsync(m.grad_flat, 0, m.n_params * sizeof(float), (hipStream_t)stream));
    return PET_OK;
}

int pet_model_get_grad(const pet_model_t* pm, const char* key, float* d_dst, int64_t numel, void* stream) {
    PET_REQUIRE(pm && key && d_dst, PET_ERR_ARGUMENT, "null argument");
    const Model& m = pm->m;
    PET_REQUIRE(m.grad_flat, PET_ERR_ARGUMENT, "pet_model_zero_grad has not been called");
    auto it = m.raw.find(key);
    PET_REQUIRE(it != m.raw.end() && it->second.second == numel, PET_ERR_ARGUMENT,
                std::string("unknown parameter or size mismatch: ") + key);
    PET_HIP_CHECK(hipMemcpyAsync(d_dst, m.grad_flat + m.grad_off.at(key), numel * sizeof(float),
                                 hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PET_OK;
}

int pet_model_get_param(const pet_model_t* pm, const char* key, float* d_dst, int64_t numel, void* stream) {
    PET_REQUIRE(pm && key && d_dst, PET_ERR_ARGUMENT, "null argument");
    auto it = pm->m.raw.find(key);
    PET_REQUIRE(it != pm->m.raw.end() && it->second.second == numel, PET_ERR_ARGUMENT,
                std::string("unknown parameter or size mismatch: ") + key);
    PET_HIP_CHECK(hipMemcpyAsync(d_dst, it->second.first, numel * sizeof(float), hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
    return PET_OK;
}

int pet_model_flat_grad(pet_model_t* pm, float* d_flat, int64_t numel, int direction, void* stream) {
    PET_REQUIRE(pm && d_flat, PET_ERR_ARGUMENT, "null argument");
    Model& m = pm->m;
    PET_REQUIRE(m.grad_flat, PET_ERR_ARGUMENT, "pet_model_zero_grad has not been called");
    PET_REQUIRE(numel == m.n_params, PET_ERR_ARGUMENT, "flat gradient has pet_model_num_params elements");
    if (direction == 0)
        PET_HIP_CHECK(hipMemcpyAsync(d_flat, m.grad_flat, numel * sizeof(float), hipMemcpyDeviceToDevice,
                                     (hipStream_t)stream));
    else
        PET_HIP_CHECK(hipMemcpyAsync(m.grad_flat, d_flat, numel * sizeof(float), hipMemcpyDeviceToDevice,
                                     (hipStream_t)stream));
    return PET_OK;
}

int pet_adam_step(pet_model_t* pm, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float max_grad_norm, int64_t step, float* d_grad_norm, void* stream) {
    PET_REQUIRE(pm, PET_ERR_ARGUMENT, "null model");
    return adam_step(pm->m, lr, beta1, beta2, eps, weight_decay, max_grad_norm, step, d_grad_norm,
                     (hipStream_t)stream);
}

int64_t pet_train_workspace_bytes(const pet_model_t* pm, int64_t n_nodes, int64_t n_edges) {
    if (!pm) return -1;
    return forward_workspace_bytes(pm->m, n_nodes, n_edges, true);
}

int64_t pet_train_workspace_bytes_for(const pet_model_t* pm, const pet_graph_t* pg) {
    if (!pm || !pg) return -1;
    if (train_generic_for(pm->m, pg->g)) return gen_workspace_bytes(pm->m, pg->g.n_nodes, pg->g.n_edges);
    return forward_workspace_bytes(pm->m, pg->g.n_nodes, pg->g.n_edges, true);
}

int64_t pet_train2_workspace_bytes_for(const pet_model_t* pm, const pet_graph_t* pg) {
    if (!pm || !pg) return -1;
    if (train_generic_for(pm->m, pg->g)) return gen_train_workspace_bytes(pm->m, pg->g.n_nodes, pg->g.n_edges);
    return so_workspace_bytes(pm->m, pg->g.n_nodes, pg->g.n_edges);
}

int pet_backward_train(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                       const float* d_grad_atomic, float* d_grad_positions, float* d_grad_cells, void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace && d_grad_atomic, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pm->m.finalized, PET_ERR_ARGUMENT, "pet_model_finalize has not been called");
    return backward_train(pm->m, pg->g, d_workspace, workspace_bytes, d_grad_atomic, d_grad_positions, d_grad_cells,
                          (hipStream_t)stream);
}

int64_t pet_train2_workspace_bytes(const pet_model_t* pm, int64_t n_nodes, int64_t n_edges) {
    if (!pm) return -1;
    return so_workspace_bytes(pm->m, n_nodes, n_edges);
}

int pet_backward_train2(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                        void* d_workspace2, int64_t workspace2_bytes, const float* d_lambda_atomic,
                        const float* d_nu_atomic, const float* d_u, float* d_tangent_atomic, void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace && d_workspace2 && d_lambda_atomic && d_u, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pm->m.finalized, PET_ERR_ARGUMENT, "pet_model_finalize has not been called");
    return backward_train2(pm->m, pg->g, d_workspace, workspace_bytes, d_workspace2, workspace2_bytes,
                           d_lambda_atomic, d_nu_atomic, d_u, d_tangent_atomic, (hipStream_t)stream);
}

int pet_backward_train2_cell(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                             void* d_workspace2, int64_t workspace2_bytes, const float* d_lambda_atomic,
                             const float* d_nu_atomic, const float* d_u, const float* d_u_cell, float* d_tangent_atomic,
                             void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace && d_workspace2 && d_lambda_atomic && d_u, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pm->m.finalized, PET_ERR_ARGUMENT, "pet_model_finalize has not been called");
    PET_REQUIRE(!d_u_cell || pg->g.shift, PET_ERR_ARGUMENT, "a cell tangent needs a pet_graph_build handle (cell shifts)");
    return backward_train2(pm->m, pg->g, d_workspace, workspace_bytes, d_workspace2, workspace2_bytes,
                           d_lambda_atomic, d_nu_atomic, d_u, d_tangent_atomic, (hipStream_t)stream, d_u_cell);
}

int64_t pet_nl_workspace_bytes(int64_t n_atoms) { return nl_workspace_bytes(n_atoms); }

int pet_nl_build(const float* d_positions, const float* h_cell, const int32_t* h_pbc, int64_t n_atoms,
                 float cutoff, void* d_workspace, int32_t* d_pairs, float* d_vectors, int64_t capacity,
                 int64_t* n_pairs, void* stream) {
    PET_REQUIRE(h_cell && h_pbc && n_pairs, PET_ERR_ARGUMENT, "null argument");
    if (n_atoms == 0) { *n_pairs = 0; return PET_OK; }
    PET_REQUIRE(d_workspace, PET_ERR_ARGUMENT, "null argument");
    return nl_build(d_positions, h_cell, h_pbc, n_atoms, cutoff, d_workspace, d_pairs, d_vectors, capacity,
                    n_pairs, (hipStream_t)stream);
}

int64_t pet_nl_batch_workspace_bytes(int64_t n_atoms, int64_t n_systems) { return nl_batch_workspace_bytes(n_atoms, n_systems); }

int pet_nl_build_batch(const float* d_positions, const float* h_cells, const int32_t* h_pbc, const int64_t* h_first_atom,
                       int64_t n_systems, float cutoff, void* d_workspace, int32_t* d_pairs, float* d_vectors,
                       int64_t capacity, int64_t* n_pairs, void* stream) {
    PET_REQUIRE(h_cells && h_pbc && h_first_atom && n_pairs, PET_ERR_ARGUMENT, "null argument");
    if (n_systems <= 0 || h_first_atom[n_systems] == 0) { *n_pairs = 0; return PET_OK; }   // no atom at all
    PET_REQUIRE(d_positions && d_workspace, PET_ERR_ARGUMENT, "null argument");
    return nl_build_batch(d_positions, h_cells, h_pbc, h_first_atom, n_systems, cutoff, d_workspace, d_pairs, d_vectors,
                          capacity, n_pairs, (hipStream_t)stream);
}

int64_t pet_graph_workspace_bytes(int64_t n_nodes, int64_t n_edges_in) {
    return graph_workspace_bytes(n_nodes, n_edges_in);
}

int pet_graph_build(const pet_model_t* pm, const float* d_positions, const float* d_cells,
                    const int32_t* d_centers, const int32_t* d_neighbors, const int32_t* d_cell_shifts,
                    const int32_t* d_species, const int32_t* d_system_indices, int64_t n_nodes,
                    int64_t n_edges_in, int64_t n_systems, void* d_workspace, int64_t workspace_bytes,
                    pet_graph_t** out, void* stream) {
    PET_REQUIRE(pm && out && d_workspace, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pm->m.species_table != nullptr, PET_ERR_ARGUMENT, "species_to_species_index not set");
    PET_REQUIRE(n_nodes < (int64_t(1) << 30) && n_edges_in < (int64_t(1) << 30), PET_ERR_ARGUMENT,
                "graph too large for int32 indices");
    pet_graph_t* pg = new pet_graph_t();
    pg->cutoff = pm->m.h.cutoff;
    int rc = graph_build(pm->m, d_positions, d_cells, d_centers, d_neighbors, d_cell_shifts, d_species,
                         d_system_indices, n_nodes, n_edges_in, n_systems, d_workspace, workspace_bytes,
                         pg->g, (hipStream_t)stream);
    if (rc != PET_OK) {
        delete pg;
        return rc;
    }
    *out = pg;
    return PET_OK;
}

void pet_graph_destroy(pet_graph_t* g) { delete g; }
int64_t pet_graph_num_edges(const pet_graph_t* g) { return g ? g->g.n_edges : 0; }
int32_t pet_graph_max_neighbors(const pet_graph_t* g) { return g ? g->g.max_nbr : 0; }

int pet_graph_export_batch(const pet_graph_t* pg, int64_t* el_nodes, int64_t* el_nbr, float* ev, float* ed,
                           uint8_t* mask, int64_t* rni, float* cf, float* stats, int64_t* centers,
                           int64_t* neighbors, int64_t* slot, int64_t* shifts, void* stream) {
    PET_REQUIRE(pg, PET_ERR_ARGUMENT, "null graph");
    return graph_export(pg->g, pg->cutoff, el_nodes, el_nbr, ev, ed, mask, rni, cf, stats, centers, neighbors,
                        slot, shifts, (hipStream_t)stream);
}

int pet_graph_csr(const pet_graph_t* pg, const int32_t** rowptr, const int32_t** ctr, const int32_t** nbr,
                  const int32_t** rev) {
    PET_REQUIRE(pg, PET_ERR_ARGUMENT, "null graph");
    if (rowptr) *rowptr = pg->g.rowptr;
    if (ctr) *ctr = pg->g.ctr;
    if (nbr) *nbr = pg->g.nbr;
    if (rev) *rev = pg->g.rev;
    return PET_OK;
}

int pet_graph_set_exchange(pet_graph_t* pg, const int32_t* d_export_rows, int64_t n_export, const int32_t* d_ghost_rows,
                           int64_t n_ghost, float* d_export_buf, float* d_ghost_buf, pet_exchange_fn fn, void* user) {
    PET_REQUIRE(pg, PET_ERR_ARGUMENT, "null graph");
    PET_REQUIRE(n_export >= 0 && n_ghost >= 0 && (n_export == 0 || (d_export_rows && d_export_buf)) &&
                    (n_ghost == 0 || (d_ghost_rows && d_ghost_buf)), PET_ERR_ARGUMENT, "bad exchange lists");
    Graph& g = pg->g;
    g.x_export = d_export_rows; g.n_export = n_export; g.x_export_buf = d_export_buf;
    g.x_ghost = d_ghost_rows; g.n_ghost = n_ghost; g.x_ghost_buf = d_ghost_buf;
    g.x_fn = fn; g.x_user = user;
    return PET_OK;
}

int pet_graph_set_conditioning(pet_graph_t* pg, const int64_t* d_charge, const int64_t* d_spin_multiplicity,
                               const int64_t* d_system_indices, int64_t n_systems) {
    PET_REQUIRE(pg && d_charge && d_spin_multiplicity && n_systems >= 1, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(d_system_indices || pg->g.sys, PET_ERR_ARGUMENT,
                "this graph handle has no system indices (pet_graph_from_batch): pass d_system_indices");
    pg->g.cond_charge = d_charge;
    pg->g.cond_spin = d_spin_multiplicity;
    pg->g.cond_sys = d_system_indices;
    pg->g.n_cond_systems = n_systems;
    return PET_OK;
}

int64_t pet_forward_workspace_bytes(const pet_model_t* pm, int64_t n_nodes, int64_t n_edges) {
    if (!pm) return -1;
    return forward_workspace_bytes(pm->m, n_nodes, n_edges);
}

int64_t pet_forward_workspace_bytes_for(const pet_model_t* pm, const pet_graph_t* pg) {
    if (!pm || !pg) return -1;
    if (use_generic(pm->m, pg->g)) return gen_workspace_bytes(pm->m, pg->g.n_nodes, pg->g.n_edges);
    return forward_workspace_bytes(pm->m, pg->g.n_nodes, pg->g.n_edges, false);
}

int pet_forward(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                int save_for_backward, float* d_atomic, float* d_node_features, float* d_edge_features,
                void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(d_atomic || d_node_features || save_for_backward, PET_ERR_ARGUMENT,
                "nothing to compute: d_atomic and d_node_features are both NULL");
    PET_REQUIRE(pm->m.finalized, PET_ERR_ARGUMENT, "pet_model_finalize has not been called");
    return forward(pm->m, pg->g, d_workspace, workspace_bytes, save_for_backward, d_atomic, d_node_features,
                   d_edge_features, (hipStream_t)stream);
}

int32_t pet_model_num_readout_layers(const pet_model_t* pm) { return pm ? pm->m.num_readout_layers() : -1; }

int pet_forward_layers(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                       int save_for_backward, float* const* h_node_features, float* const* h_edge_features, int32_t n_layers,
                       void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace && h_node_features && h_edge_features, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pm->m.finalized, PET_ERR_ARGUMENT, "pet_model_finalize has not been called");
    PET_REQUIRE(n_layers == pm->m.num_readout_layers(), PET_ERR_ARGUMENT,
                "expected one feature pair per readout layer (" + std::to_string(pm->m.num_readout_layers()) + ")");
    return forward_layers(pm->m, pg->g, d_workspace, workspace_bytes, save_for_backward, nullptr, h_node_features,
                          h_edge_features, n_layers, (hipStream_t)stream);
}

int pet_backward_features_layers(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                                 const float* const* h_grad_node_features, const float* const* h_grad_edge_features,
                                 int32_t n_layers, float* d_grad_geometry, float* d_grad_cutoff, void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace && h_grad_node_features && h_grad_edge_features && d_grad_geometry && d_grad_cutoff,
                PET_ERR_ARGUMENT, "null argument");
    return backward_features_layers_abi(pm->m, pg->g, d_workspace, workspace_bytes, h_grad_node_features,
                                        h_grad_edge_features, n_layers, d_grad_geometry, d_grad_cutoff, (hipStream_t)stream);
}

// ---- predict as a function of its arguments ---------------------------------------------------------------------
static int find_head(const Model& m, const char* target, int32_t layer, const char* block, const HeadW** H, const LastW** Lw) {
    PET_REQUIRE(target && block, PET_ERR_ARGUMENT, "null target / block name");
    const std::string hk = std::string(target) + "|" + std::to_string(layer);
    const auto hi = m.heads.find(hk);
    PET_REQUIRE(hi != m.heads.end(), PET_ERR_ARGUMENT, "no heads were uploaded for target '" + std::string(target) +
                                                           "', readout layer " + std::to_string(layer));
    const auto li = m.lasts.find(hk + "|" + block);
    PET_REQUIRE(li != m.lasts.end(), PET_ERR_ARGUMENT, "no last layer was uploaded for block '" + std::string(block) + "' of target '" +
                                                           std::string(target) + "'");
    *H = &hi->second;
    *Lw = &li->second;
    return PET_OK;
}

int32_t pet_model_block_properties(const pet_model_t* pm, const char* target, int32_t readout_layer, const char* block) {
    if (!pm || !pm->m.finalized) return -1;
    const HeadW* H;
    const LastW* Lw;
    if (find_head(pm->m, target, readout_layer, block, &H, &Lw) != PET_OK) return -1;
    return Lw->P;
}

int64_t pet_predict_scratch_floats(int64_t n_nodes, int64_t n_edges) { return predict_scratch_floats(n_nodes, n_edges); }

int pet_predict(const pet_model_t* pm, const pet_graph_t* pg, const char* target, int32_t readout_layer, const char* block,
                const float* d_node_features, const float* d_edge_features, const float* d_cutoff_factors, float* d_atomic,
                float* d_node_hidden, float* d_edge_hidden, float* d_scratch, void* stream) {
    PET_REQUIRE(pm && pg, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pm->m.finalized, PET_ERR_ARGUMENT, "pet_model_finalize has not been called");
    const HeadW* H;
    const LastW* Lw;
    int rc = find_head(pm->m, target, readout_layer, block, &H, &Lw);
    if (rc) return rc;
    if (pg->g.n_nodes == 0) return PET_OK;  // an empty system (pet/tests/test_functionality.py:79-103): nothing to write
    PET_REQUIRE(d_node_features && d_atomic && d_scratch, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(d_edge_features || pg->g.n_edges == 0, PET_ERR_ARGUMENT, "null edge features");
    return predict(pm->m, pg->g, *H, *Lw, d_node_features, d_edge_features, d_cutoff_factors, d_atomic, d_node_hidden,
                   d_edge_hidden, d_scratch, (hipStream_t)stream);
}

int pet_predict_backward(const pet_model_t* pm, const pet_graph_t* pg, const char* target, int32_t readout_layer,
                         const char* block, const float* d_node_features, const float* d_edge_features,
                         const float* d_cutoff_factors, const float* d_grad_atomic, float* d_grad_node_features,
                         float* d_grad_edge_features, float* d_grad_cutoff, float* d_scratch, void* stream) {
    PET_REQUIRE(pm && pg, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pm->m.finalized, PET_ERR_ARGUMENT, "pet_model_finalize has not been called");
    const HeadW* H;
    const LastW* Lw;
    int rc = find_head(pm->m, target, readout_layer, block, &H, &Lw);
    if (rc) return rc;
    if (pg->g.n_nodes == 0) return PET_OK;
    PET_REQUIRE(d_node_features && d_grad_atomic && d_grad_node_features && d_scratch, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE((d_edge_features && d_grad_edge_features && d_grad_cutoff) || pg->g.n_edges == 0, PET_ERR_ARGUMENT,
                "null edge argument");
    return predict_backward(pm->m, pg->g, *H, *Lw, d_node_features, d_edge_features, d_cutoff_factors, d_grad_atomic,
                            d_grad_node_features, d_grad_edge_features, d_grad_cutoff, d_scratch, (hipStream_t)stream);
}

int pet_geometry_backward(const pet_model_t* pm, const pet_graph_t* pg, const float* d_grad_geometry,
                          const float* d_grad_cutoff, float* d_grad_positions, float* d_grad_cells, float* d_scratch,
                          void* stream) {
    PET_REQUIRE(pm && pg && d_grad_positions, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pg->g.n_edges == 0 || (d_grad_geometry && d_grad_cutoff && d_scratch), PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pg->g.d0 != nullptr, PET_ERR_ARGUMENT,
                "this graph was made from batch_data and has no positions: use the pet_graph_build handle");
    return geometry_backward(pm->m, pg->g, d_grad_geometry, d_grad_cutoff, d_grad_positions, d_grad_cells, d_scratch,
                             (hipStream_t)stream);
}

int64_t pet_graph_from_batch_workspace_bytes(int64_t n_nodes, int64_t max_neighbors) {
    return graph_from_batch_workspace_bytes(n_nodes, max_neighbors);
}

int pet_graph_from_batch(const int64_t* d_element_indices_nodes, const int64_t* d_element_indices_neighbors,
                         const float* d_edge_vectors, const float* d_edge_distances, const uint8_t* d_padding_mask,
                         const int64_t* d_reverse_neighbor_index, const float* d_cutoff_factors, int64_t n_nodes,
                         int64_t max_neighbors, void* d_workspace, int64_t workspace_bytes, pet_graph_t** out, void* stream) {
    PET_REQUIRE(out && d_workspace && n_nodes >= 0 && max_neighbors >= 0, PET_ERR_ARGUMENT, "bad argument");
    PET_REQUIRE(n_nodes * max_neighbors == 0 || (d_padding_mask && d_cutoff_factors), PET_ERR_ARGUMENT,
                "padding_mask and cutoff_factors are required");
    PET_REQUIRE((d_edge_vectors == nullptr) == (d_edge_distances == nullptr), PET_ERR_ARGUMENT,
                "edge_vectors and edge_distances come together");
    pet_graph_t* pg = new pet_graph_t();
    pg->cutoff = 0.f;
    int rc = graph_from_batch(d_element_indices_nodes, d_element_indices_neighbors, d_edge_vectors, d_edge_distances,
                              d_padding_mask, d_reverse_neighbor_index, d_cutoff_factors, n_nodes, max_neighbors,
                              d_workspace, workspace_bytes, pg->g, (hipStream_t)stream);
    if (rc != PET_OK) {
        delete pg;
        return rc;
    }
    *out = pg;
    return PET_OK;
}

int pet_aux_outputs(const pet_model_t* pm, const pet_graph_t* pg, const float* d_node_features,
                    const float* d_edge_features, float* d_feature, float* d_last_layer_features, float* d_scratch,
                    void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_node_features, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(d_edge_features || pg->g.n_edges == 0, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(pm->m.finalized, PET_ERR_ARGUMENT, "pet_model_finalize has not been called");
    return aux_outputs(pm->m, pg->g, d_node_features, d_edge_features, d_feature, d_last_layer_features, d_scratch,
                       (hipStream_t)stream);
}

int pet_backward(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                 const float* d_grad_atomic, float* d_grad_positions, float* d_grad_cells, void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace && d_grad_atomic && d_grad_positions, PET_ERR_ARGUMENT,
                "null argument");
    PET_REQUIRE(pm->m.finalized, PET_ERR_ARGUMENT, "pet_model_finalize has not been called");
    return backward(pm->m, pg->g, d_workspace, workspace_bytes, d_grad_atomic, d_grad_positions, d_grad_cells,
                    (hipStream_t)stream);
}

int pet_backward_predict(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                         const float* d_grad_atomic, float* d_grad_node_features, float* d_grad_edge_features,
                         float* d_grad_cutoff, void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace && d_grad_atomic, PET_ERR_ARGUMENT, "null argument");
    return backward_predict_abi(pm->m, pg->g, d_workspace, workspace_bytes, d_grad_atomic, d_grad_node_features,
                                d_grad_edge_features, d_grad_cutoff, (hipStream_t)stream);
}

int pet_backward_features(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                          const float* d_grad_node_features, const float* d_grad_edge_features,
                          float* d_grad_geometry, float* d_grad_cutoff, void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace && d_grad_node_features && d_grad_geometry && d_grad_cutoff,
                PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE(d_grad_edge_features || pg->g.n_edges == 0, PET_ERR_ARGUMENT, "null argument");
    return backward_features_abi(pm->m, pg->g, d_workspace, workspace_bytes, d_grad_node_features,
                                 d_grad_edge_features, d_grad_geometry, d_grad_cutoff, (hipStream_t)stream);
}

int pet_backward_geometry(const pet_model_t* pm, const pet_graph_t* pg, void* d_workspace, int64_t workspace_bytes,
                          const float* d_grad_geometry, const float* d_grad_cutoff, float* d_grad_positions,
                          float* d_grad_cells, void* stream) {
    if (pg && pg->g.n_nodes == 0) return PET_OK;  // an empty system: nothing to compute, zero-sized buffers may be null
    PET_REQUIRE(pm && pg && d_workspace && d_grad_positions, PET_ERR_ARGUMENT, "null argument");
    PET_REQUIRE((d_grad_geometry && d_grad_cutoff) || pg->g.n_edges == 0, PET_ERR_ARGUMENT, "null argument");
    return backward_geometry_abi(pm->m, pg->g, d_workspace, workspace_bytes, d_grad_geometry, d_grad_cutoff,
                                 d_grad_positions, d_grad_cells, (hipStream_t)stream);
}

int pet_sum_over_atoms(const pet_graph_t* pg, const float* d_atomic, float* d_out, void* stream) {
    PET_REQUIRE(pg && d_out, PET_ERR_ARGUMENT, "null argument");
    if (pg->g.n_nodes == 0) {
        PET_HIP_CHECK(hipMemsetAsync(d_out, 0, (size_t)(pg->g.n_systems > 0 ? pg->g.n_systems : 0) * sizeof(float),
                                     (hipStream_t)stream));
        return PET_OK;
    }
    PET_REQUIRE(d_atomic, PET_ERR_ARGUMENT, "null argument");
    return sum_over_atoms(pg->g, d_atomic, d_out, (hipStream_t)stream);
}

int pet_profile_enable(int on) {
    g_prof_on = on != 0;
    return PET_OK;
}

int pet_profile_select(const char* stage) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_filter = stage ? stage : "";
    return PET_OK;
}

int pet_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    return PET_OK;
}

int pet_config_set(const char* key, int value) {
    PET_REQUIRE(key, PET_ERR_ARGUMENT, "null key");
    const std::string k(key);
    if (k == "side_stream") set_side_stream(value);
    else if (k == "trr") set_use_trr(value);
    else if (k == "soap_mfma") set_soap_mfma(value);
    else if (k == "soap_pair") set_soap_pair(value);
    else if (k == "soap_packed") set_soap_packed(value);
    else if (k == "soap_ps_mfma") set_soap_ps_mfma(value);
    else if (k == "soap_sorted") set_soap_sorted(value);
    else if (k == "attn_fused") set_attn_fused(value);
    else if (k == "emlp_s") set_emlp_s(value);
    else if (k == "attn_fused_prof") ablk_prof_dump();
    else if (k == "trr_compress") set_trr_compress(value);
    else if (k == "node_planes") set_node_planes(value);
    else if (k == "center_fused") set_center_fused(value);
    else if (k == "dxf_fused") set_dxf_fused(value);
    else if (k == "node_split") set_node_split(value);
    else if (k == "sorted_shortcut") set_sorted_shortcut(value);
    else if (k == "so_trr") set_so_trr(value);
    else if (k == "wgrad_bf16") set_wgrad_bf16(value);
    else if (k == "train_bf16") set_train_bf16(value);
    else if (k == "so_f16x3") set_so_f16x3(value);
    else PET_REQUIRE(false, PET_ERR_ARGUMENT, "unknown config key '" + k + "'");
    return PET_OK;
}

int pet_profile_report(int max_entries, char (*names)[64], double* total_ms, int64_t* calls, double* flops,
                       double* bytes, int* n_entries) {
    PET_REQUIRE(names && total_ms && calls && flops && bytes && n_entries, PET_ERR_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::vector<std::string> order;
    std::map<std::string, int> index;
    for (auto& r : g_prof) {
        PET_HIP_CHECK(hipEventSynchronize(r.e1));
        float ms = 0.f;
        PET_HIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
        auto it = index.find(r.name);
        int i;
        if (it == index.end()) {
            if ((int)order.size() >= max_entries) continue;
            i = (int)order.size();
            index[r.name] = i;
            order.push_back(r.name);
            strncpy(names[i], r.name.c_str(), 63);
            names[i][63] = 0;
            total_ms[i] = 0;
            calls[i] = 0;
            flops[i] = 0;
            bytes[i] = 0;
        } else {
            i = it->second;
        }
        total_ms[i] += ms;
        calls[i] += 1;
        flops[i] += r.flops;
        bytes[i] += r.bytes;
    }
    *n_entries = (int)order.size();
    return PET_OK;
}

}  // extern "C"
