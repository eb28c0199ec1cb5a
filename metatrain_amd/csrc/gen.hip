// Size-generic PET path: forward + hand-written reverse pass (dE/dR) for ANY (d_pet, d_node, d_feedforward, d_head,
// num_heads) -- the reference is size-generic (pet/documentation.py:196-213; its own architecture suites run at
// d_pet = 1, pet/tests/test_basic.py:22-32) while the tuned kernels of pet_fwd / pet_trr / pet_attn / pet_comb are ONE
// compiled instantiation (128 / 256 / 256 / 128 / 8). A model of any other size runs here, behind the same C ABI:
//
//   * every Linear is one GEMM kernel over the RAW torch weights [n_out, k_in] on the fp32 matrix core (k_gen_lin:
//     v_mfma_f32_32x32x2_f32, 64 x 64 output tiles, K chunks of 32 through LDS, run-time bounds everywhere), the adjoint
//     dX = dY W is the same kernel with swapped strides;
//   * norms / SwiGLU / SiLU are row kernels with a run-time width (one wave per row);
//   * attention is one wave per (atom, head): a lane is (query | key, 16-feature slice) -- queries for the forward and dQ, keys
//     for dK, dV and the key-bias gradient -- and walks the other index with an online soft-max, so any head dimension up
//     to 128 and ANY number of neighbours is served (no 16-token tiles); dot products are xor-shuffle sums over the slices:
//     fixed summation order, bit-reproducible;
//   * d_node == d_pet follows transformer.py:189-201: no centre contraction / expansion / centre MLP, the node features
//     leaving a layer ARE the centre token;
//   * all architecture switches of the tuned path: RMSNorm / LayerNorm, PreLN / PostLN, feedforward / residual featuriser,
//     SwiGLU / SiLU (tied halves), system conditioning, bump / cosine / adaptive cutoffs (graph side, shared).
// Correctness-first: fp32 products throughout (no split operands, no tuned tiles); rates in DESIGN.md section 1.
// Training on this path: gen_train.hip.
#include <string>
#include <vector>

#include "gen_common.h"

namespace pet {

int attn_tiles(const Graph& g);
int backward_geometry_generic(const Model& m, const Graph& g, float* dv_scratch, const float* dgeo, const float* dfc_a,
                              const float* dfc_b, float* gpos, float* gcell, hipStream_t st);  // pet_bwd.hip


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
    PoolBuf scratch_pool;
    PET_HIP_CHECK(scratch_pool.alloc((size_t)need * sizeof(float), st));
    scratch = scratch_pool.as<float>();
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
    PoolBuf scratch_pool;
    PET_HIP_CHECK(scratch_pool.alloc((size_t)(4 * M * DH + 2 * M * P) * sizeof(float), st));
    scratch = scratch_pool.as<float>();
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
        PoolBuf tmp_pool;
        PET_HIP_CHECK(tmp_pool.alloc((size_t)4 * M * d.DH * sizeof(float), st));
        float* tmp = tmp_pool.as<float>();
        float* b1 = tmp; float* b2 = b1 + M * d.DH; float* b3 = b2 + M * d.DH; float* b4 = b3 + M * d.DH;
        gen_head_fwd(o, m.nh0, m.nh2, node_feat, d.DN, N, b1, b2, b3, b4);
        o.axpby(1.f, b4, d.DH, 0.f, nullptr, 0, nullptr, last_layer, 2 * d.DH, false, N, d.DH);
        if (E > 0) {
            gen_head_fwd(o, m.eh0, m.eh2, edge_feat, d.D, E, b1, b2, b3, b4);
            k_gen_edge_sum<<<g1(N * d.DH), 256, 0, st>>>(b4, g.fc, g.rowptr, last_layer + d.DH, 2 * d.DH, N, d.DH);
        } else
            o.axpby(0.f, b4, d.DH, 0.f, nullptr, 0, nullptr, last_layer + d.DH, 2 * d.DH, false, N, d.DH);
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
    (void)save;  // everything the reverse passes need is kept anyway; the training pass (gen_train.hip) recomputes
    const bool post = m.post_ln(), res = m.residual();
    PET_REQUIRE(res ? (n_layers == m.h.num_gnn_layers || (n_layers == 1 && !node_feats[0] && !edge_feats[0])) : n_layers == 1,
                PET_ERR_ARGUMENT, "expected one feature pair per readout layer");
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
        // the fused single-property target; with the residual featuriser the sum over the readout layers (backend.py:468-481)
        const int NR = m.num_readout_layers();
        PoolBuf tmp_pool;
        if (NR > 1) PET_HIP_CHECK(tmp_pool.alloc((size_t)N * sizeof(float), st));
        float* tmp = tmp_pool.as<float>();
        for (int l = 0; l < NR; l++) {
            auto hi = m.heads.find("@|" + std::to_string(l));
            auto li = m.lasts.find("@|" + std::to_string(l) + "|@");
            PET_REQUIRE(hi != m.heads.end() && li != m.lasts.end() && li->second.P == 1, PET_ERR_ARGUMENT,
                        "pet_forward with d_atomic needs the fused single-property target (of every readout layer); use "
                        "pet_predict for other heads");
            const GGnn& Bl = res ? w.gnn[l] : last;
            int rc = gen_predict(m, g, hi->second, li->second, Bl.Hout, res ? Bl.XF : Bl.Mout, g.fc, l == 0 ? atomic : tmp,
                                 nullptr, nullptr, st);
            if (rc) return rc;
            if (l > 0) o.add(tmp, atomic, N, 1);
        }
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
    const int64_t N = g.n_nodes, E = g.n_edges;
    if (N == 0) return PET_OK;
    const GD d = dims_of(m);
    const int NR = m.num_readout_layers();
    const bool res = m.residual();
    const int64_t Ea = E > 0 ? E : 1;
    float* buf = nullptr;   // per readout layer: d node features, d edge features; then d fc (heads, summed), scratch d fc, d geo, d fc (attention)
    const size_t per = (size_t)N * d.DN + (size_t)Ea * d.D;
    PoolBuf buf_pool;
    PET_HIP_CHECK(buf_pool.alloc((per * NR + (size_t)Ea * 7) * sizeof(float), st));
    buf = buf_pool.as<float>();
    float* gfh = buf + per * NR;
    float* gft = gfh + Ea;
    float* ggeo = gft + Ea;
    float* gfc = ggeo + Ea * 4;
    std::vector<const float*> gnp(NR), gep(NR);
    int rc = PET_OK;
    for (int l = 0; l < NR && !rc; l++) {
        auto hi = m.heads.find("@|" + std::to_string(l));
        auto li = m.lasts.find("@|" + std::to_string(l) + "|@");
        PET_REQUIRE(hi != m.heads.end() && li != m.lasts.end() && li->second.P == 1, PET_ERR_ARGUMENT,
                    "pet_backward needs the fused single-property target (of every readout layer)");
        const GGnn& Bl = res ? w.gnn[l] : w.gnn.back();
        float* gn = buf + per * l;
        float* ge = gn + (size_t)N * d.DN;
        gnp[l] = gn; gep[l] = ge;
        rc = gen_predict_backward(m, g, hi->second, li->second, Bl.Hout, res ? Bl.XF : Bl.Mout, g.fc, gA, gn, ge, l == 0 ? gfh : gft, st);
        if (!rc && l > 0 && E > 0) { Ops o(m, g, st); o.add(gft, gfh, E, 1); }
    }
    if (!rc) rc = gen_backward_features(m, g, ws, ws_bytes, gnp.data(), gep.data(), NR, ggeo, gfc, st);
    // the two cutoff-factor gradients (heads, attention key biases) are added by the geometry kernel
    if (!rc) rc = backward_geometry_generic(m, g, w.dv, ggeo, E > 0 ? gfh : nullptr, E > 0 ? gfc : nullptr, gpos, gcell, st);
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
    PoolBuf tmp_pool;
    PET_HIP_CHECK(tmp_pool.alloc((size_t)(g.n_nodes * d.DN + Ea * d.D + Ea) * sizeof(float), st));
    tmp = tmp_pool.as<float>();
    int rc = gen_predict_backward(m, g, H, m.lasts.at("@|0|@"), last.Hout, last.Mout, g.fc, gA, g_node ? g_node : tmp,
                                  g_edge ? g_edge : tmp + g.n_nodes * d.DN, g_fc ? g_fc : tmp + g.n_nodes * d.DN + Ea * d.D, st);
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
