// Device-resident PET weights: raw copies keyed by the reference state-dict names
//  plus the packed / derived forms the kernels consume.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace pet {

// compiled instantiation (pet/documentation.py:159-259 defaults)
constexpr int D = 128;        // d_pet
constexpr int DN = 256;       // d_node
constexpr int DFF = 256;      // d_feedforward (edge SwiGLU hidden; w_in has 2*DFF outputs)
constexpr int DNF = 2 * DN;   // node SwiGLU hidden (transformer.py:188-190: 2 * dim_node_features)
constexpr int DH = 128;       // d_head
constexpr int NHEAD = 8;
constexpr int HD = D / NHEAD; // 16
constexpr int MAX_SPECIES = 128;

// y = x W^T + b with W [n_out, k_in] (torch Linear). fwd: packed for x W^T;
// bwd: packed W^T for dx = dy W.
struct Lin {
    const float* w = nullptr;  // raw [n_out, k_in]
    const float* b = nullptr;  // raw [n_out]
    float4* fwd = nullptr;
    float4* bwd = nullptr;
    void* fwd2 = nullptr;  // f16x3 operands (trr.h): two fp16 planes, fragment order
    void* bwd2 = nullptr;
    void* fwd2s = nullptr;  // the same planes with the low piece scaled by 64 (single-accumulator products, pet_ablk.hip)
    void* bwd2s = nullptr;
    int n_out = 0, k_in = 0;
};

struct AttnLayerW {
    Lin qkv, out, mlp_in, mlp_out, cc, ce, cmlp_in, cmlp_out;
    Lin qkv_g;     // qkv with norm_attention folded in (k_ablk_bwd)
    Lin mlp_in_g;  // mlp_in with norm_mlp folded in: W diag(gamma), b + W beta (k_emlp_bwd_s works on the un-scaled normalised rows)
    const float *g_attn = nullptr, *g_mlp = nullptr, *g_center = nullptr;
    const float *b_attn = nullptr, *b_mlp = nullptr, *b_center = nullptr;  // LayerNorm biases (nullptr: RMSNorm)
};

struct GnnLayerW {
    std::vector<AttnLayerW> attn;
    Lin compress2;      // [D, D]
    Lin compress0_msg;  // columns of compress.0 that multiply the incoming message (g >= 1)
    float* wc = nullptr;   // [D, 4]  compress.0[:, :D] @ edge_embedder.weight  (4 -> D composite)
    float* wct = nullptr;  // [4, D]  transpose, for the backward dot products
    float* wcp = nullptr;  // [32, D] wct padded with zero rows to one MFMA tile
    void* wc2 = nullptr;   // f16x3 planes of wcp in fragment order (dgeo = da0 Wc as a 32-wide GEMM tile, pet_trr.hip)
    void* wc2s = nullptr;  // the same as planes of 64 w (single-accumulator products, pet_compress_s.hip)
    float* tbl = nullptr;  // [n_species, D] species part of compress.0 (+ all biases)
    Lin comb0, comb2;
    Lin comb0_g;  // comb0 with the combination LayerNorm's weight and bias folded in (k_comb_s, pet_comb_s.hip)
    const float *ln_g = nullptr, *ln_b = nullptr;
    // raw forms for the size-generic path (gen.hip): edge_embedder (4 -> d_pet), compress.0 as uploaded, neighbor_embedder
    Lin eemb, c0;
    const float* nbr_emb = nullptr;
};

// Heads of one (target, readout layer): node_heads.<t>.<l>.{0,2}, edge_heads.<t>.<l>.{0,2} (backend.py:171-217);
// last layers of one (target, readout layer, block): [P, DH] weights + [P] biases, node and edge (P = properties).
struct HeadW {
    Lin nh0, nh2, eh0, eh2;
};
struct LastW {
    const float *nw = nullptr, *nb = nullptr, *ew = nullptr, *eb = nullptr;
    int P = 0;
};

// the tuned kernels are ONE instantiation; every other size runs on the generic path (gen.hip)
inline bool compiled_size(const pet_hypers_t& h) {
    return h.d_pet == D && h.d_node == DN && h.d_feedforward == DFF && h.d_head == DH && h.num_heads == NHEAD;
}

struct Model {
    pet_hypers_t h;
    bool generic() const { return !compiled_size(h); }
    std::map<std::string, std::pair<float*, int64_t>> raw;  // device copies
    int* species_table = nullptr;
    int species_table_len = 0;
    std::vector<GnnLayerW> gnn;
    const float* node_emb = nullptr;  // [ns, DN]  node_embedders.0
    std::vector<const float*> node_embs;  // node_embedders.<l>: one per GNN layer for the residual featuriser, else [0]
    // architecture variants (pet_hypers_t): the TRR kernels serve RMSNorm + PreLN transformer layers only
    bool layer_norm() const { return h.normalization == PET_NORM_LAYER; }
    bool post_ln() const { return h.transformer_type == PET_POST_LN; }
    bool residual() const { return h.featurizer_type == PET_FEATURIZER_RESIDUAL; }
    bool plain_layers() const { return !post_ln(); }  // PreLN (either norm): what the TRR transformer-layer kernels serve
    bool plain() const { return plain_layers() && !residual(); }
    bool trainable() const { return !post_ln() && !residual(); }  // RMSNorm or LayerNorm, PreLN, feedforward featuriser
    int num_readout_layers() const { return residual() ? h.num_gnn_layers : 1; }
    const float* edge_emb = nullptr;  // [ns, D]
    // system conditioning (conditioning.py:38-52): embeddings [2 max_charge + 1, DN], [max_spin, DN]; project.0 [DN, 2 DN],
    // project.2 [DN, DN] as raw torch Linear weights (a per-SYSTEM MLP: a few rows, evaluated by one small kernel)
    const float *cond_qe = nullptr, *cond_se = nullptr, *cond_w0 = nullptr, *cond_b0 = nullptr, *cond_w2 = nullptr,
                *cond_b2 = nullptr;
    // the FUSED target (keys with "@" for target and block: pet_forward / the native training step): one property
    bool has_fused_head = false;
    Lin nh0, nh2, eh0, eh2;
    const float *nll_w = nullptr, *ell_w = nullptr;  // [DH]
    float nll_b = 0.f, ell_b = 0.f;
    // every head uploaded (the fused one included, under "@"), for pet_predict: key "<target>|<layer>" / "...|<block>"
    std::map<std::string, HeadW> heads;
    std::map<std::string, LastW> lasts;
    std::vector<void*> owned;  // device allocations to free
    std::map<std::string, std::pair<void*, size_t>> named;  // derived buffers, reused across finalize calls
    bool finalized = false;
    int64_t n_params = 0;
    // training (row a16): one flat gradient buffer, parameter `key` at grad_off[key] (upload order)
    std::map<std::string, int64_t> grad_off;
    float* grad_flat = nullptr;
    float* adam_m = nullptr;   // first / second moments, same layout as grad_flat
    float* adam_v = nullptr;
    float** seg_ptr = nullptr; // device table: parameter storage of segment i
    int64_t* seg_off = nullptr;  // device table: first flat index of segment i (n_seg + 1 entries)
    int n_seg = 0;
    float* opt_scalars = nullptr;  // [4] sum of squares, clip coefficient
    // tied parameters (activation = "SiLU": w_in held as [W; W]): (flat offset, half length) pairs
    std::vector<int64_t> ties;
    bool ties_dirty = false;
    int64_t* d_ties = nullptr;
    uint8_t* d_dup = nullptr;  // [n_params] 1 on the second copy of a tied parameter
    int64_t max_tie_half = 0;
};

// abi.hip
int dev_alloc(Model& m, void** p, size_t bytes);
int finalize(Model& m, hipStream_t st);

// optim.hip: fused clip_grad_norm_ + Adam/AdamW over the flat gradient buffer, then re-pack
int adam_step(Model& m, float lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
              int64_t step, float* d_grad_norm, hipStream_t st);

int tie_halves(Model& m, const std::string& key);
int optimizer_state(Model& m, float* d_m, float* d_v, int64_t numel, int direction, hipStream_t st);

// graph.hip
// na + nb <= MAILBOX_INTS - 1 device integers (two sources, either may be empty) to `out` on the host, ordered after
// everything issued on `st` so far: a one-wave kernel into the calling thread's pinned mailbox, which the host polls
constexpr int MAILBOX_INTS = 64;
int read_back(const int* d_a, int na, const int* d_b, int nb, int* out, hipStream_t st);
int64_t graph_workspace_bytes(int64_t n_nodes, int64_t e0);
int graph_attention_lists(const Graph& g, hipStream_t st);  // graph.hip: lazily, for a graph built before pet_model_finalize
int graph_build(const Model& m, const float* pos, const float* cells, const int* centers,
                const int* neighbors, const int* shifts, const int* species, const int* sys,
                int64_t n_nodes, int64_t e0, int64_t n_systems, void* ws, int64_t ws_bytes,
                Graph& g, hipStream_t st);
int graph_check_reverse(Graph& g, hipStream_t st);
int64_t graph_from_batch_workspace_bytes(int64_t n_nodes, int64_t max_nbr);
int graph_from_batch(const int64_t* el_nodes, const int64_t* el_nbr, const float* ev, const float* ed, const uint8_t* mask,
                     const int64_t* rni, const float* cf, int64_t n_nodes, int64_t max_nbr, void* ws, int64_t ws_bytes,
                     Graph& g, hipStream_t st);
int graph_export(const Graph& g, float cutoff, int64_t* el_nodes, int64_t* el_nbr, float* ev,
                 float* ed, uint8_t* mask, int64_t* rni, float* cf, float* stats, int64_t* centers,
                 int64_t* neighbors, int64_t* slot, int64_t* shifts, hipStream_t st);
int sum_over_atoms(const Graph& g, const float* atomic, float* out, hipStream_t st);

// nl.hip
int64_t nl_workspace_bytes(int64_t n_atoms);
int64_t nl_batch_workspace_bytes(int64_t n_atoms, int64_t n_systems);
int nl_build_batch(const float* d_pos, const float* h_cells, const int* h_pbc, const int64_t* h_first_atom, int64_t n_sys,
                   float cutoff, void* ws, int* d_pairs, float* d_vectors, int64_t capacity, int64_t* n_pairs,
                   hipStream_t st);
int nl_build(const float* d_pos, const float* h_cell, const int* h_pbc, int64_t n, float cutoff,
             void* ws, int* d_pairs, float* d_vectors, int64_t capacity, int64_t* n_pairs,
             hipStream_t st);

// A graph with an atom of more than 127 neighbours (attention tiles of 16 tokens, at most 8 per atom in the tuned kernels)
// runs on the size-generic path too, whatever the model size: its attention walks keys one by one, without a tile limit.
bool use_generic(const Model& m, const Graph& g);  // pet_fwd.hip

// gen.hip: the size-generic path (any d_pet / d_node / d_feedforward / d_head / num_heads; inference + dE/dR)
int64_t gen_workspace_bytes(const Model& m, int64_t n_nodes, int64_t n_edges);
int gen_forward_layers(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, int save, float* atomic,
                       float* const* node_feats, float* const* edge_feats, int n_layers, hipStream_t st);
int gen_backward_features(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* const* g_node,
                          const float* const* g_edge, int n_layers, float* g_geo, float* g_fc, hipStream_t st);
int gen_backward(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* gA, float* gpos, float* gcell,
                 hipStream_t st);
int gen_predict(const Model& m, const Graph& g, const HeadW& H, const LastW& Lw, const float* node_feat,
                const float* edge_feat, const float* fc, float* atomic, float* node_hidden, float* edge_hidden,
                hipStream_t st);
int gen_predict_backward(const Model& m, const Graph& g, const HeadW& H, const LastW& Lw, const float* node_feat,
                         const float* edge_feat, const float* fc, const float* gA, float* g_node, float* g_edge, float* g_fc,
                         hipStream_t st);
int gen_aux_outputs(const Model& m, const Graph& g, const float* node_feat, const float* edge_feat, float* feature,
                    float* last_layer, float* scratch, hipStream_t st);
int gen_backward_predict(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* gA, float* g_node,
                         float* g_edge, float* g_fc, hipStream_t st);
int gen_backward_geometry(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* g_geo, const float* g_fc,
                          float* gpos, float* gcell, hipStream_t st);
// gen_train.hip: the size-generic TRAINING pass (forward-over-reverse on dual activations, any size / PostLN / residual)
int64_t gen_train_workspace_bytes(const Model& m, int64_t n_nodes, int64_t n_edges);
int norm_rev_rows(const float* Xp, const float* Xt, const float* gamma, int ln, float eps, const float* NYp, const float* NYt,
                  float* NXp, float* NXt, int64_t R, int W, hipStream_t st);
int gen_train2(const Model& m, const Graph& g, void* ws2, int64_t ws2_bytes, const float* lA, const float* nA, const float* u,
               const float* ucell, float* tangent_atomic, hipStream_t st);
// a model whose TRAINING runs on the size-generic path: other sizes, PostLN layers, the residual featuriser
inline bool train_generic(const Model& m) { return m.generic() || !m.trainable(); }
// ... and for a built graph: an atom of more than 127 neighbours, or NO edge at all (a batch of isolated atoms -- reference
// structures of a dataset -- trains the node path alone; the tuned passes launch over E rows)
inline bool train_generic_for(const Model& m, const Graph& g) { return train_generic(m) || use_generic(m, g) || g.n_edges == 0; }
// the workspace `ws` was last filled by a size-generic forward (pet_fwd.hip keeps the record): its adjoints must follow
bool generic_workspace(const Graph& g, const void* ws);
int backward_geometry_generic(const Model& m, const Graph& g, float* dv_scratch, const float* dgeo, const float* dfc_a,
                              const float* dfc_b, float* gpos, float* gcell, hipStream_t st);

// pet_fwd.hip / pet_bwd.hip
int64_t forward_workspace_bytes(const Model& m, int64_t n_nodes, int64_t n_edges, bool train = false);
int forward(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, int save, float* atomic,
            float* node_feat, float* edge_feat, hipStream_t st);
int forward_layers(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, int save, float* atomic,
                   float* const* node_feats, float* const* edge_feats, int n_layers, hipStream_t st);
int backward_features_layers_abi(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* const* g_node,
                                 const float* const* g_edge, int n_layers, float* g_geo, float* g_fc, hipStream_t st);
int backward(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* grad_atomic,
             float* grad_pos, float* grad_cells, hipStream_t st);
int64_t predict_scratch_floats(int64_t n_nodes, int64_t n_edges);
int predict(const Model& m, const Graph& g, const HeadW& H, const LastW& Lw, const float* node_feat, const float* edge_feat,
            const float* fc, float* atomic, float* node_hidden, float* edge_hidden, float* scratch, hipStream_t st);
int predict_backward(const Model& m, const Graph& g, const HeadW& H, const LastW& Lw, const float* node_feat,
                     const float* edge_feat, const float* fc, const float* grad_atomic, float* g_node, float* g_edge,
                     float* g_fc, float* scratch, hipStream_t st);
int aux_outputs(const Model& m, const Graph& g, const float* node_feat, const float* edge_feat, float* feature,
                float* last_layer, float* scratch, hipStream_t st);
int backward_train(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* grad_atomic,
                   float* grad_pos, float* grad_cells, hipStream_t st);
// so.hip: second-order (force-loss) reverse pass
int64_t so_workspace_bytes(const Model& m, int64_t n_nodes, int64_t n_edges);
// tangents of (edge vector, distance), cutoff factor and key bias along (u, ucell), adaptive cutoffs ('solver') included
int geometry_tangent(const Model& m, const Graph& g, const float* u, const float* ucell, float* Tgeo, float* Tfc, float* Tkb,
                     hipStream_t st);
int backward_train2(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, void* ws2, int64_t ws2_bytes,
                    const float* lambda_atomic, const float* nu_atomic, const float* u, float* tangent_atomic,
                    hipStream_t st, const float* u_cell = nullptr);
int backward_predict_abi(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* grad_atomic,
                         float* g_node, float* g_edge, float* g_fc, hipStream_t st);
int backward_features_abi(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* g_node,
                          const float* g_edge, float* g_geo, float* g_fc, hipStream_t st);
int geometry_backward(const Model& m, const Graph& g, const float* g_geo, const float* g_fc, float* gpos, float* gcell,
                      float* scratch, hipStream_t st);
int backward_geometry_abi(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, const float* g_geo,
                          const float* g_fc, float* grad_pos, float* grad_cells, hipStream_t st);

// pet_trr.hip: transposed register-resident stages (default; PET_HIP_TRR=0 selects the LDS-tile kernels)
bool use_trr();
void set_use_trr(int v);
void set_side_stream(int v);
void set_so_f16x3(int v);   // so.hip: 1 = generic training GEMMs as f16x3 on the 16-bit matrix cores (default), 0 = fp32 MFMA
void set_train_bf16(int v);  // so.hip / train.hip: 1 = ONE 16-bit MFMA term per product in the training GEMMs (default 0: f16x3 / bf16x3)
int train_bf16();
void set_wgrad_bf16(int v);  // train.hip: 1 = weight gradients as bf16x3 products on the 16-bit matrix cores (default)
void set_so_trr(int v);     // so.hip: 1 = K = 128 / n_out = 128 generic GEMMs as TRR kernels (default)
void set_soap_mfma(int v);  // soap.hip: 1 = MFMA tail (default), 0 = per-atom tail kernels
void set_soap_ps_mfma(int v);  // soap.hip: 1 = power spectrum and its adjoint on the fp32 matrix core (default)
void set_soap_packed(int v); // soap.hip: 1 = inference stores the upper triangle of every power-spectrum block only (default)
void set_trr_compress(int v);
void set_node_planes(int v);  // pet_fwd.hip / pet_bwd.hip: node-row kernels on pre-split fp16 planes (default 1)
bool node_planes();
void set_dxf_fused(int v);     // pet_bwd.hip: 1 = dXF formed inside k_comb_bwd_p2 / k_emlp_bwd_p2 instead of by k_dxf (default)
void set_node_split(int v);    // pet_fwd.hip: 1 = graphs of <= 4 096 atoms: four workgroups per 32-row tile of the node update (default)
bool node_split_on();
void set_center_fused(int v);  // pet_fwd.hip: 1 = k_node2 also writes the next layer's centre tokens (default)
int node_rows(int64_t N);   // rows per workgroup of the node-row kernels (32: two workgroups per CU; 64)
struct Graph;
bool trr_compress(bool first, const Graph& g, const GnnLayerW& G, const float* Min, float* a0_out, float* Xout, int64_t E,
                  hipStream_t st);
bool trr_compress_bwd(bool first, const float* dXe, const float* a0, const GnnLayerW& G, float* dgeo, float* dM, int64_t E,
                      float* t_da0, hipStream_t st);
struct Model;
bool trr_head_edge(const Model& m, const float* Xin, const float* fc, float* ypred, float* yout, int64_t E, hipStream_t st);
bool trr_head_edge_bwd(const Model& m, const float* Xin, const float* gA, const int* ctr, const float* fc,
                       const float* ypred, float* dfc, float* dXout, int64_t E, float* t_s1, float* t_da2, float* t_da1,
                       float* t_s2y, hipStream_t st);
void set_soap_sorted(int v);  // soap.hip: 1 = tail GEMM on species-sorted tiles, one network per tile (default)
void set_soap_pair(int v);  // soap.hip: 1 = wave-per-atom expansion / lane-per-pair adjoint (default), 0 = first generation
// beta: the LayerNorm bias of the layer's norm, nullptr = RMSNorm
void trr_qkv(const float* X, const float* gamma, const float* beta, const Lin& qkv, float* QKV, int64_t R,
             hipStream_t st);
void trr_qkv_bwd(const float* dQKV, const float* X, const float* gamma, bool layer_norm, const Lin& qkv,
                 const float* dX1, float* dXin, int64_t E, int64_t R, hipStream_t st);
void trr_oproj(const float* AO, const float* X, const Lin& out, float* X1, float* OC, int64_t E, int64_t R,
               hipStream_t st);
void trr_oproj_bwd(const float* dX1, const float* dOC, const Lin& out, float* dAO, int64_t E, int64_t R,
                   hipStream_t st);
void trr_emlp(const float* X1, const float* gamma, const float* beta, const Lin& win, const Lin& wout, float* VG,
              float* X2, int64_t E, hipStream_t st);
void trr_emlp_bwd(const float* dY, const float* X1, const float* VG, const float* gamma, const float* beta,
                  const Lin& win, const Lin& wout, float* dX1, int64_t E, hipStream_t st, float* t_dvg = nullptr,
                  int ldy = 128, const float* dY2 = nullptr, const int* rev2 = nullptr);  // dY2: dY = dY[p] + dY2[rev2[p]], rows of ldy floats

// pet_comb.hip: combination stage and adjoint as TRR kernels (f16x3); false if the split operands are missing
bool trr_comb(bool first, const float* XF, const Graph& g, const GnnLayerW& G, const float* Min,
              const float* edge_emb, float* CA, float* LNS, float* Mout, int64_t E, hipStream_t st);
bool trr_comb_bwd(const float* dM, const float* XF, const Graph& g, const GnnLayerW& G, const float* LNS,
                  const float* CA, float* dcat, int64_t E, float* t_da, hipStream_t st, bool add_dm = false);  // add_dm: dcat[p][:D] += dM[p]

// pet_ablk.hip: the per-atom fused attention block (norm -> QKV -> attention -> output projection in one kernel, the
// adjoint recomputing Q, K, V); false = not served (an atom of more than 64 tokens, planes missing, switched off)
void set_sorted_shortcut(int v);  // graph.hip: 1 = a neighbour list that is ordered by centre skips the radix sort (default)
void set_attn_fused(int v);
void set_emlp_s(int v);
bool emlp_recompute_on(const Lin& win, const Lin& wout, int64_t E);
bool emlp_s_serves(int64_t E);
// pet_comb_bwd_s.hip: the inference adjoint of the combination stage with a workgroup-shared weight ring; false = not served
bool comb_bwd_s(const float* dM, const float* XF, const int* rev, const float* LNS, const float* CA, const Lin& c0g, const Lin& c2,
                float* dcat, int64_t E, bool add_dm, hipStream_t st);
// pet_node_s.hip: the node update of large graphs as three shared-ring row GEMMs (every kernel fits beside an edge kernel's workgroup)
bool node_fwd_s(const AttnLayerW& A, const float* H, const float* OC, float* H1, float* VGn, float* Hn, float* tmp, int64_t N,
                hipStream_t st);
bool node_bwd_s(const AttnLayerW& A, const float* dHn, const float* H1, const float* VGn, float* dH1, float* tmp, int64_t N, bool ln,
                hipStream_t st);
// so_rows_s.hip: the generic row GEMM of the training passes with a workgroup-shared weight ring; false = not served
bool rowgemm_s(hipStream_t st, const float* X, int K, const float* cs, const void* planes, const float* bias, float* Y, int n_out,
               int64_t R, bool acc);
// the same with an addend A (may be Y) and, for K == 256, a RMSNorm (norm 1) / LayerNorm (2) of the rows in front (weight cs, bias cb)
bool rowgemm_s_ex(hipStream_t st, const float* X, int K, const float* cs, const void* planes, const float* bias, const float* A,
                  float* Y, int n_out, int64_t R, int norm, const float* cb);
bool rowgemm_s_swiglu_bwd(hipStream_t st, const float* X, const void* planes, const float* VG, float* dVG, int hid, int64_t R);
bool rowgemm_s_norm_bwd(hipStream_t st, const float* X, int K, const void* planes, const float* xn, const float* gamma, int ln,
                        const float* dres, float* out, int64_t R);
bool emlp_s_forced();
struct Model;
struct GnnLayerW;
struct Graph;
bool compress_bwd_s(bool first, const float* dXe, const float* a0, const GnnLayerW& G, float* dgeo, float* dM, int64_t E, hipStream_t st);
bool center_s(const Lin& cc, const float* H, float* Xc, int64_t N, hipStream_t st);
bool expand_bwd_s(const Lin& ce, const float* dH1, float* dOC, int64_t N, hipStream_t st);
bool center_bwd_s(const Lin& cc, const float* dC, const float* dH1, float* dHin, int64_t N, hipStream_t st);
bool comb_s(bool first, const float* XF, const int* rev, const Lin& c0g, const Lin& c2, const float* Min, const float* edge_emb,
            const int* sp_nbr, float* CA, float* LNS, float* Mout, int64_t E, hipStream_t st);
bool head_edge_s(const Model& m, const float* Xin, const float* fc, float* ypred, float* yout, int64_t E, hipStream_t st);
bool head_edge_bwd_s(const Model& m, const float* Xin, const float* gA, const int* ctr, const float* fc, const float* ypred,
                     float* dfc, float* dXout, int64_t E, hipStream_t st);
bool emlp_bwd_s(const float* dY, const float* X1, bool ln, const Lin& win_g, const Lin& wout, float* dX1, int64_t E,
                hipStream_t st, int ldy, const float* dY2, const int* rev2);
bool emlp_s(const float* X1, const float* gamma, const float* beta, const Lin& win, const Lin& wout, float* VG, float* X2,
            int64_t E, hipStream_t st);
int attn_fused();
// graphs of at least this many 32-slot attention tiles (about 4 700 atoms at 19 neighbours) take the fused per-atom block:
// measured crossover of one box, graph + forward + dE/dR, fused against three-kernel form -- 3 000 atoms 3.29 / 2.96 ms,
// 5 000: 4.08 / 4.16, 7 000: 5.21 / 5.51, 10 000: 6.72 / 7.30 (round 5, k_ablk_fwd4 and the VGPR-form adjoint)
constexpr int ABLK_MIN_TILES = 3840;
void ablk_prof_dump();  // debugging aid: per-phase cycle sums of the fused kernels (library built with -DAB_PROFILE)
bool ablk_fwd(const Model& m, const Graph& g, const AttnLayerW& A, const float* X, float* X1, float* OC, float scale,
              hipStream_t st);
bool ablk_bwd_on(const Graph& g);
bool ablk_bwd(const Model& m, const Graph& g, const AttnLayerW& A, const float* X, const float* dX1, const float* dOC,
              float* dXin, float* dbias, float scale, hipStream_t st);

// pet_attn.hip: preload variants of the attention kernels (NT <= 4); return false if not handled
bool attn_fwd_preload(int nt, const float* QKV, const Graph& g, float* AO, float scale, hipStream_t st);
bool attn_bwd_preload(int nt, const float* QKV, const float* dAO, const Graph& g, float* dQKV, float* dbias_h,
                      float scale, hipStream_t st);

// second-order attention on MFMA (training pass); false if the tile count is not served
bool attn_jvp_mfma(int nt, const float* QKV, const float* QKVd, const Graph& g, const float* Tkb, float* AOd,
                   float scale, hipStream_t st);
bool attn_rev_mfma(int nt, const float* QKV, const float* QKVd, const Graph& g, const float* Tkb, const float* LO,
                   const float* NO, float* lQKV, float* nQKV, float scale, hipStream_t st);

// abi.hip: a second HIP stream for the node-feature chain, which is independent of the edge chain
// between output_linear and the next attention layer (PET_HIP_SIDE=0 runs everything on one stream)
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t to_side = nullptr, to_main = nullptr;
    bool enabled = false;
    hipStream_t stream(hipStream_t main) const { return enabled ? s : main; }
    void fork(hipStream_t main) const;  // side waits for everything issued on main so far
    void join(hipStream_t main) const;  // main waits for everything issued on side so far
};
const SideStream& side_stream();

// profiling (abi.hip)
struct ProfScope {
    ProfScope(const char* name, hipStream_t st, double flops, double bytes = 0.0);
    ~ProfScope();
    const char* name;
    hipStream_t st;
    double flops;
    double bytes = 0.0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

}  // namespace pet

// opaque handle behind pet_graph_t (shared by abi.hip and soap.hip)
struct pet_graph {
    pet::Graph g;
    float cutoff;
};
