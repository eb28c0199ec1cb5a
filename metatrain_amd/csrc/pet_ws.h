// Activation workspace shared by the forward and backward passes. Everything the
// reverse pass needs is kept (288 GB of HBM3E per GPU: ~24 KB per edge is cheap),
// so the backward never recomputes a GEMM.
#pragma once
#include <vector>

#include "common.h"
#include "model.h"

namespace pet {

struct AttnBufs {
    float* X = nullptr;    // [(E+N), D] tokens entering the layer (edges rows 0..E-1, centres E..)
    float* QKV = nullptr;  // [(E+N), 3D]
    float* X1 = nullptr;   // [E, D]  edges after attention residual (input of the edge MLP)
                           // PostLN: [(E+N), D] tokens + attention output BEFORE norm_attention (transformer.py:245)
    float* S2 = nullptr;   // PostLN only: [(E+N), D] tokens + MLP output before norm_mlp (transformer.py:247)
    float* VG = nullptr;   // [E, 2*DFF] SwiGLU pre-activations (value | gate); PostLN: (E+N) rows, the MLP sees every token
    float* H = nullptr;    // [N, DN] node features entering the layer (alias of the producer)
    float* Hn = nullptr;   // [N, DN] node features leaving the layer
    float* H1 = nullptr;   // [N, DN] after centre expansion residual
    float* VGn = nullptr;  // [N, 2*DNF]
    float* AO = nullptr;   // [(E+N), D] attention output before output_linear (shared temp unless training)
    float* OC = nullptr;   // [N, D] centre rows of output_linear (shared temp unless training)
};

struct GnnBufs {
    std::vector<AttnBufs> attn;
    float* a0 = nullptr;    // [E, D] compress.0 pre-activation
    float* XF = nullptr;    // [E, D] edge tokens leaving the transformer
    float* CA = nullptr;    // [E, 2D] combination MLP pre-activation
    float* LNS = nullptr;   // [E, 2] LayerNorm (mean, rstd) of [e ; e_rev]
    float* Mout = nullptr;  // [E, D] messages leaving the layer
    float* Hout = nullptr;  // [N, DN]
    float* Hin = nullptr;   // [N, DN] node features entering the layer (residual featuriser: its own embedding)
};

struct Workspace {
    std::vector<GnnBufs> gnn;
    float* H0 = nullptr;     // [N, DN] node embedding
    float* AO = nullptr;     // [(E+N), D] attention output before output_linear (temp)
    float* OC = nullptr;     // [N, D] centre rows of output_linear (temp)
    float* T1 = nullptr;     // PostLN only: [(E+N), D] norm_attention output (temp)
    float* cond = nullptr;   // system conditioning only: [n_systems <= N, DN] per-system embedding
    float* ypred_e = nullptr;  // [E] edge last-layer prediction before the cutoff weight
    float* ye = nullptr;       // [E] fc * ypred_e
    float* ynode = nullptr;    // [N]
    // backward temporaries
    float* dM = nullptr;      // [E, D]
    float* dM2 = nullptr;     // [E, D]
    float* dX = nullptr;      // [(E+N), D]
    float* dX2 = nullptr;     // [(E+N), D]
    float* dQKV = nullptr;    // [(E+N), 3D]
    float* dAO = nullptr;     // [(E+N), D]
    float* dcat = nullptr;    // [E, 2D]
    float* dH = nullptr;      // [N, DN]
    float* dH2 = nullptr;     // [N, DN]
    float* dOC = nullptr;     // [N, D]
    float* dgeo = nullptr;    // [E, 4] accumulated d/d(vx,vy,vz,dist)
    float* dfc = nullptr;     // [E]
    float* dbias = nullptr;   // [E] attention key-bias gradient (summed over heads/layers)
    float* dbias_l = nullptr; // [layers, NHEAD, E] key-bias gradient per attention layer, head-major
    float* delta = nullptr;   // [(E+N), NHEAD]
    float* lse = nullptr;     // [(E+N), NHEAD]
    float* dv = nullptr;      // [E, 4] d/d(edge vector)
    // training only (row a16): adjoint operands the weight-gradient GEMMs need
    float* dVG = nullptr;     // [E, 2*DFF]  d(value | gate) of the edge MLP
    float* dVGn = nullptr;    // [N, 2*DNF]
    float* dCA = nullptr;     // [E, 2D]     d(combination MLP pre-activation)
    float* da0 = nullptr;     // [E, D]      d(compress.0 pre-activation)
    float* hs1 = nullptr;     // [max(E,N), DH] head temporaries: silu(a1), d a2, d a1, gy * silu(a2)
    float* hda2 = nullptr;
    float* hda1 = nullptr;
    float* hs2y = nullptr;
    float* dcond = nullptr;   // system conditioning only: [n_systems <= N, DN] adjoint of the per-system embedding
    float* gmat = nullptr;    // [2][1024 x 256] reduced weight-gradient block (+ LayerNorm beta scratch)
    float* gvec = nullptr;    // [32768]
    float* partial = nullptr; // split-K partials
    size_t partial_floats = 0;
    size_t bytes = 0;
    const void* base = nullptr;  // what the caller handed over (key of Graph::fwd_record)
};

inline void carve_workspace(const Model& m, int64_t N, int64_t E, void* base, Workspace& w, bool train = false) {
    Carver c(base);
    w.base = base;
    const int64_t R = E + N;
    const int64_t Ea = E > 0 ? E : 1, Na = N > 0 ? N : 1, Ra = R > 0 ? R : 1;
    w.gnn.resize(m.h.num_gnn_layers);
    for (auto& G : w.gnn) {
        G.attn.resize(m.h.num_attention_layers);
        for (auto& A : G.attn) {
            A.X = c.take<float>(Ra * D);
            A.QKV = c.take<float>(Ra * 3 * D);
            A.X1 = c.take<float>((m.post_ln() ? Ra : Ea) * D);
            A.VG = c.take<float>((m.post_ln() ? Ra : Ea) * 2 * DFF);
            if (m.post_ln()) A.S2 = c.take<float>(Ra * D);
            A.H1 = c.take<float>(Na * DN);
            A.VGn = c.take<float>(Na * 2 * DNF);
        }
        G.a0 = c.take<float>(Ea * D);
        G.XF = c.take<float>(Ea * D);
        G.CA = c.take<float>(Ea * 2 * D);
        G.LNS = c.take<float>(Ea * 2);
        G.Mout = c.take<float>(Ea * D);
        G.Hout = nullptr;
    }
    // node feature chain: H0 plus one buffer per attention layer
    w.H0 = c.take<float>(Na * DN);
    float* prev = w.H0;
    for (size_t gi = 0; gi < w.gnn.size(); gi++) {
        GnnBufs& G = w.gnn[gi];
        if (m.residual() && gi > 0) prev = c.take<float>(Na * DN);  // backend.py:617: a fresh embedding per layer
        G.Hin = prev;
        for (auto& A : G.attn) {
            A.H = prev;
            A.Hn = c.take<float>(Na * DN);
            prev = A.Hn;
        }
        G.Hout = prev;
    }
    w.AO = c.take<float>(Ra * D);
    w.OC = c.take<float>(Na * D);
    if (m.post_ln()) w.T1 = c.take<float>(Ra * D);
    if (m.h.system_conditioning) w.cond = c.take<float>(Na * DN);
    w.ypred_e = c.take<float>(Ea);
    w.ye = c.take<float>(Ea);
    w.ynode = c.take<float>(Na);
    w.dM = c.take<float>(Ea * D);
    w.dM2 = c.take<float>(Ea * D);
    w.dX = c.take<float>(Ra * D);
    w.dX2 = c.take<float>(Ra * D);
    w.dQKV = c.take<float>(Ra * 3 * D);
    w.dAO = c.take<float>(Ra * D);
    w.dcat = c.take<float>(Ea * 2 * D);
    w.dH = c.take<float>(Na * DN);
    w.dH2 = c.take<float>(Na * DN);
    w.dOC = c.take<float>(Na * D);
    w.dgeo = c.take<float>(Ea * 4);
    w.dfc = c.take<float>(Ea);
    w.dbias = c.take<float>(Ea);
    w.delta = c.take<float>(Ra * NHEAD);
    w.lse = c.take<float>(Ra * NHEAD);
    w.dv = c.take<float>(Ea * 4);
    w.dbias_l = c.take<float>((int64_t)m.h.num_gnn_layers * m.h.num_attention_layers * NHEAD * Ea);
    // everything above is identical with and without `train`, so an inference backward can run on
    // a workspace carved for training
    for (auto& G : w.gnn)
        for (auto& A : G.attn) {
            A.AO = train ? c.take<float>(Ra * D) : w.AO;
            A.OC = train ? c.take<float>(Na * D) : w.OC;
        }
    if (train) {
        const int64_t Ma = Ea > Na ? Ea : Na;
        w.dVG = c.take<float>(Ea * 2 * DFF);
        w.dVGn = c.take<float>(Na * 2 * DNF);
        w.dCA = c.take<float>(Ea * 2 * D);
        w.da0 = c.take<float>(Ea * D);
        w.hs1 = c.take<float>(Ma * DH);
        w.hda2 = c.take<float>(Ma * DH);
        w.hda1 = c.take<float>(Ma * DH);
        w.hs2y = c.take<float>(Ma * DH);
        if (m.h.system_conditioning) w.dcond = c.take<float>(Na * DN);
        w.gmat = c.take<float>(2 * 1024 * 256);
        w.gvec = c.take<float>(32768);
        w.partial_floats = (size_t)48 << 20;  // 192 MB of split-K partials
        w.partial = c.take<float>(w.partial_floats);
    }
    w.bytes = c.off;
}

}  // namespace pet
