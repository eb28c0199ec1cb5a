// Parameter gradients (first order) for the training row: the reverse pass of
// pet_bwd.hip calls these hooks while the adjoint operands of each stage are still in its
// temporaries; every hook is one or a few split-K weight-gradient GEMMs (wgrad.h).
#pragma once
#include <string>

#include "common.h"
#include "model.h"
#include "pet_ws.h"

namespace pet {

struct Trainer {
    const Model& m;
    const Graph& g;
    Workspace& w;
    float* grads;  // flat, laid out like Model::grad_off
    hipStream_t st;
    int err = PET_OK;

    float* gp(const std::string& key) const;  // destination of one parameter's gradient

    // generic y = x W^T + b: dW [n_out, k_in] (+ db) from dY rows and rebuilt X rows
    struct Y { const float* p0; const float* p1; int64_t split; int ld; };
    struct X { const float* p; int ld; int hid; const int* rev; const float* lns; };
    // xmode: 0 plain, 1 rms-hat, 2 swiglu(v|g), 3 silu, 4 layernorm-hat([x; x[rev]]), 5 layernorm-hat(x)
    void linear(const std::string& key, int n_out, int k_in, Y y, X x, int xmode, int64_t n_rows,
                bool with_bias = true);
    // same, X normalised by a norm with weight `gamma_key` (and bias `beta_key` for LayerNorm)
    void linear_after_norm(const std::string& key, const float* W, int n_out, int k_in, Y y, X x, int xmode,
                           int64_t n_rows, const std::string& gamma_key, const float* gamma,
                           const std::string& beta_key = "", const float* beta = nullptr,
                           bool tangent_pair = false);
    void colsum(const float* buf, int64_t n_rows, int C, float* dst);  // dst[c] += sum_rows buf[row][c]
    void vecsum(const float* vec, int64_t n_rows, float* dst);         // dst[0] += sum_rows vec[row]
    void species_rows(const float* buf, const int* idx, int64_t n_rows, int C, float* dst);

    void heads(bool edge, const float* Xin, int k_in, int64_t n_rows, const float* gA);
    void embeddings(const float* dH0, const float* dM0);
    // system conditioning (conditioning.py:82-100; backend.py:543-545 adds the per-system embedding to the node features
    // LEAVING every GNN layer): cond_accumulate sums that layer's node-feature adjoint over the atoms of each system
    // into w.dcond, cond_finish back-propagates the sum through the two-layer projection and the two embeddings
    void cond_accumulate(const float* dHout, bool first);
    void cond_finish();
    // (la0, Tgeo, TMin): second-order pair lambda_a0 with the tangents of the compress.0 inputs
    void compress0(int gi, const float* da0, const float* Min, const float* la0 = nullptr,
                   const float4* Tgeo = nullptr, const float* TMin = nullptr);
};

}  // namespace pet
