/* C ABI of the MI355X (gfx950) SOAP-BPNN hot path: libpet_hip.so, soap_* entry points.
 *
 * Replaces, behind plain device pointers, what the reference computes in
 *   soap_bpnn/modules/power_spectrum.py:66-175  SoapPowerSpectrum.forward  (-> torch-spex SphericalExpansion)
 *   soap_bpnn/model.py:553-595, 1204-1219       centre encoding, LayerNorm, MLP, bias-free last layer
 *   utils/output_gradient.py:7-63               dE/dR by autograd
 * Edge geometry, CSR order, cutoff factors and the ij->ji map come from the same pet_graph_t the PET
 * path uses (pet_graph_build with the SOAP cutoff radius / width and the ShiftedCosine = "Cosine" taper).
 *
 * PARITY UNPINNED: the spherical-expansion arithmetic of the reference lives in
 * torch-spex / sphericart, which are not under /root/reference; this path is checked against the CPU
 * restatement in oracle/soap.py only.
 *
 * All pointers named d_* are device pointers; every call is asynchronous on `stream` (a hipStream_t).
 * Return value: 0 on success, negative PET_ERR_* otherwise; message via pet_last_error().
 */
#ifndef SOAP_HIP_H
#define SOAP_HIP_H

#include <stdint.h>

#include "pet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct soap_model soap_model_t;

#define SOAP_MAX_L 8

typedef struct {
    float cutoff;             /* soap.cutoff.radius */
    float cutoff_width;       /* soap.cutoff.width (ShiftedCosine) */
    int32_t max_angular;      /* <= SOAP_MAX_L */
    int32_t n_per_l[SOAP_MAX_L + 1]; /* radial functions kept per l (LaplacianEigenstates trimming, host side) */
    int32_t n_species;        /* number of atomic types */
    int32_t n_channels;       /* species channels: n_species (Orthogonal, legacy) or 4 (Alchemical) */
    int32_t legacy;           /* 1: per-centre-species LayerNorm / MLP / last layer, identity species weights */
    int32_t layernorm;        /* bpnn.layernorm */
    int32_t num_hidden_layers;      /* bpnn.num_hidden_layers (1 .. 8) */
    int32_t num_neurons_per_layer;  /* bpnn.num_neurons_per_layer (1 .. 64; 32 = the MFMA tails) */
} soap_hypers_t;

int soap_model_create(const soap_hypers_t* hypers, soap_model_t** out);
void soap_model_destroy(soap_model_t* m);
/* size of the power spectrum = sum_l (n_per_l[l] * n_channels)^2 */
int64_t soap_model_feature_size(const soap_model_t* m);
/* Radial basis as a Hermite spline table on a uniform grid of n_grid points over [0, cutoff]:
 * d_table [n_grid][F][4] = (R(r_k), dR/dr(r_k), (R(r_k+1) - R(r_k)) / h, 0), F = sum_l n_per_l[l], functions ordered
 * l-major (what spex's spliner holds, plus the chord slope of the interval that starts at the node, which the
 * derivative of the Hermite cubic needs and which must not be formed from the fp32 node values; 0 at the last node;
 * built on the host in fp64, see metatrain_amd/soap_bpnn/radial.py). */
int soap_model_set_radial_table(soap_model_t* m, const float* d_table, int32_t n_grid, void* stream);
/* Parameters (fp32, row-major), keys:
 *   "species_embedding.weight" [n_species, n_channels]   (Alchemical only)
 *   "center_encoding.weight"   [n_species, S]            (non-legacy only)
 *   "layernorm.<s>.weight" / ".bias" [S],  "bpnn.<s>.0.weight" [H, S],  "bpnn.<s>.2.weight" [H, H],
 *   "last_layers.energy.<s>.weight" [1, H]   with s = centre species index (legacy) or 0. */
int soap_model_set_param(soap_model_t* m, const char* key, const float* d_data, int64_t numel, void* stream);
int soap_model_finalize(soap_model_t* m, void* stream);

int64_t soap_workspace_bytes(const soap_model_t* m, int64_t n_nodes, int64_t n_edges);
/* Per-atom energies d_atomic [N]; d_features [N, S] optional copy of the (centre-encoded) power spectrum.
 * Keeps what soap_backward needs in the workspace. */
int soap_forward(const soap_model_t* m, const pet_graph_t* g, void* d_workspace, int64_t workspace_bytes,
                 float* d_atomic, float* d_features, void* stream);
/* dL/dR [N,3] (and dL/dcell [S,3,3], may be NULL) for dL/d(atomic energy) = d_grad_atomic [N]. */
int soap_backward(const soap_model_t* m, const pet_graph_t* g, void* d_workspace, int64_t workspace_bytes,
                  const float* d_grad_atomic, float* d_grad_positions, float* d_grad_cells, void* stream);

/* ---- training step (reference loop body soap_bpnn/trainer.py:344-391: zero_grad, evaluate_model(is_training=True),
 * loss.backward(), Adam lr 1e-3 without clipping, soap_bpnn/documentation.py TrainerHypers) -------------------------------
 * Trainable: layernorm.<s>.{weight,bias}, bpnn.<s>.<2k>.weight, last_layers.energy.<s>.weight (every parameter of a
 * legacy = True model) and, for legacy = False models, species_embedding.weight and center_encoding.weight. */
/* Allocates (first call) and zeroes the gradient slot of every trainable parameter (optimizer.zero_grad()). */
int soap_model_zero_grad(soap_model_t* m, void* stream);
int64_t soap_train_workspace_bytes(const soap_model_t* m, int64_t n_nodes, int64_t n_edges);
/* What loss.backward() leaves in parameter.grad for  L = L_E(E) + L_F(dE/dR):  ADDS  d/d theta [ sum_i gA_i e_i + <u, dE/dR> ]
 * to the slots, with d_grad_atomic [N] = dL_E/d e_i and d_u [N,3] = dL_F/d(dE/dR) (NULL: energy-only loss). The double
 * backward of utils/output_gradient.py:34-40 (create_graph=True) is done as forward-over-reverse: d_tangent_atomic [N]
 * receives the tangent of every atomic energy along u (its sum equals <u, dE/dR>, a self-check). soap_forward must have
 * run on d_workspace for this graph. */
int soap_train_gradients(soap_model_t* m, const pet_graph_t* g, void* d_workspace, int64_t workspace_bytes,
                         void* d_train_workspace, int64_t train_workspace_bytes, const float* d_grad_atomic,
                         const float* d_u, float* d_tangent_atomic, void* stream);
int soap_model_get_grad(const soap_model_t* m, const char* key, float* d_out, int64_t numel, void* stream);
int soap_model_get_param(const soap_model_t* m, const char* key, float* d_out, int64_t numel, void* stream);
/* torch.optim.Adam step (no weight decay; `step` counted from 1) on every trainable parameter, then soap_model_finalize. */
int soap_adam_step(soap_model_t* m, float lr, float beta1, float beta2, float eps, int64_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif
