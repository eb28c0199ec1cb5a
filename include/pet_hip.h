/*
 * pet_hip.h -- C ABI of libpet_hip.so: the MI355X (gfx950) hot path of metatrain's PET.
 *
 * Drop-in boundary: the plain-tensor L1 API of the reference,
 *   PETBackend.preprocess          src/metatrain/pet/modules/backend.py:238-342
 *   PETBackend.calculate_features  src/metatrain/pet/modules/backend.py:344-418
 *   PETBackend.predict             src/metatrain/pet/modules/backend.py:420-494
 *   compute_gradient (dE/dR)       src/metatrain/utils/output_gradient.py:7-63
 *   vesin.ase_neighbor_list        src/metatrain/utils/neighbor_lists.py:131-135
 * The reference is pure Python on torch; a maintainer binds this library with the
 * ctypes stub shown in INTEGRATION.md (or torch.ops via a thin TORCH_LIBRARY shim).
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory owned by the caller (e.g. a torch
 *     tensor's data_ptr()); the library never frees or retains it beyond the call,
 *     except for workspaces which the caller keeps alive while handles refer to them;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream);
 *   - all calls return PET_OK (0) or a negative error code; pet_last_error() returns
 *     a thread-local human-readable message for the last failure;
 *   - float data is fp32, indices are int32 unless stated otherwise;
 *   - no torch types, no host-side allocation of device memory on the hot path.
 */
#ifndef PET_HIP_H
#define PET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PET_OK 0
#define PET_ERR_HIP -1          /* a HIP runtime call failed                        */
#define PET_ERR_UNSUPPORTED -2  /* hypers outside the compiled kernel instantiation */
#define PET_ERR_ARGUMENT -3     /* bad argument / missing parameter / short buffer  */
#define PET_ERR_GRAPH -4        /* neighbour list is not a full list (no ji for ij) */

#define PET_CUTOFF_COSINE 0
#define PET_CUTOFF_BUMP 1

#define PET_NORM_RMS 0
#define PET_NORM_LAYER 1
#define PET_PRE_LN 0
#define PET_POST_LN 1
#define PET_FEATURIZER_FEEDFORWARD 0
#define PET_FEATURIZER_RESIDUAL 1
#define PET_ADAPTIVE_SOLVER 0
#define PET_ADAPTIVE_GRID 1

/* Mirrors the subset of ModelHypers that shapes the hot path
 * (src/metatrain/pet/documentation.py:159-259). */
typedef struct pet_hypers {
    float cutoff;            /* 4.5   */
    float cutoff_width;      /* 0.5   */
    int32_t cutoff_function; /* PET_CUTOFF_BUMP */
    int32_t d_pet;           /* 128   */
    int32_t d_head;          /* 128   */
    int32_t d_node;          /* 256   */
    int32_t d_feedforward;   /* 256   */
    int32_t num_heads;       /* 8     */
    int32_t num_attention_layers; /* 2 */
    int32_t num_gnn_layers;       /* 2 */
    float attention_temperature;  /* 1.0 */
    int32_t nl_is_strict;    /* = long_range.enable (backend.py:38); 0 => filter d <= cutoff */
    int32_t n_species;       /* len(atomic_types) */
    int32_t max_atomic_number; /* species_to_species_index has max+1 entries */
    float num_neighbors_adaptive; /* target neighbour count of the adaptive cutoff ("solver" method,
                                     pet/modules/adaptive_cutoff.py:110-229); <= 0: fixed cutoff */
    float cutoff_width_adaptive;  /* taper width of the adaptive-cutoff probe */
    /* the architecture variants older checkpoints use (pet/checkpoints.py:190-205 upgrades them to
     * LayerNorm + PostLN + residual); all 0 = the current defaults */
    int32_t normalization;    /* PET_NORM_RMS | PET_NORM_LAYER (transformer.py:170-176; LayerNorm: eps 1e-5, weight + bias) */
    int32_t transformer_type; /* PET_PRE_LN (transformer.py:203-234) | PET_POST_LN (transformer.py:236-262) */
    int32_t featurizer_type;  /* PET_FEATURIZER_FEEDFORWARD (backend.py:496-587) | PET_FEATURIZER_RESIDUAL (backend.py:589-649) */
    int32_t adaptive_cutoff_method; /* PET_ADAPTIVE_SOLVER (adaptive_cutoff.py:110-229) | PET_ADAPTIVE_GRID (:232-395, legacy) */
    int32_t system_conditioning;    /* 1: charge / spin-multiplicity embedding added to the node features leaving every
                                       GNN layer (conditioning.py, backend.py:121-130,517-545); needs pet_graph_set_conditioning */
    int32_t max_charge;             /* charges in [-max_charge, max_charge]   (10) */
    int32_t max_spin_multiplicity;  /* multiplicities in [1, max]             (10) */
} pet_hypers_t;

typedef struct pet_model pet_model_t; /* packed weights on the device */
typedef struct pet_graph pet_graph_t; /* CSR edge graph living in a caller workspace */

/* ---- library ------------------------------------------------------------------ */
const char* pet_last_error(void);
const char* pet_version(void);
/* 1 if the default hypers instantiation (d_pet=128, d_node=256, d_ff=256, d_head=128,
 * heads=8) matches `h`, else 0. */
int pet_hypers_supported(const pet_hypers_t* h);

/* ---- model: owns device copies of the weights in MFMA fragment order ----------- */
int pet_model_create(const pet_hypers_t* h, pet_model_t** out);
void pet_model_destroy(pet_model_t* m);
/* Upload one tensor of the reference state dict (SURVEY §8(b) key schema, e.g.
 * "gnn_layers.0.trans.layers.1.mlp.w_in.weight"). `d_data` is a device pointer to
 * `numel` contiguous fp32 values (int64 values for "species_to_species_index").
 * The FUSED target (what pet_forward / the native training step evaluate; one property) is
 * uploaded with its target and block names replaced by "@" ("node_heads.@.0.0.weight" ...,
 * "node_last_layers.@.0.@.weight"); any further heads / last layers (other targets, blocks with
 * P > 1 properties, other readout layers) are uploaded under their own names and served by
 * pet_predict.
 * activation = "SiLU" (transformer.py:32-49): upload every "...w_in.weight" / "...w_in.bias" as the tensor stacked on
 * itself ([W; W], [b; b]): with equal value and gate halves the SwiGLU stage computes silu(W x + b) exactly; the
 * gradient of W is the sum of the two halves' gradients (metatrain_amd/runtime.py does both). */
int pet_model_set_param(pet_model_t* m, const char* key, const void* d_data,
                        int64_t numel, void* stream);
/* Pack / precompute derived tables once all parameters are set. */
int pet_model_finalize(pet_model_t* m, void* stream);
int64_t pet_model_num_params(const pet_model_t* m);

/* ---- neighbour list (replaces vesin at utils/neighbor_lists.py:131-135) --------- */
/* Full periodic neighbour list of ONE system within `cutoff` (strict: |D| < cutoff).
 * Two-call protocol: pass d_pairs = NULL to count; then allocate n_pairs rows.
 *   d_positions [n,3] fp32, h_cell[9] row-major lattice (host), h_pbc[3] (host)
 *   d_pairs     [n_pairs,5] int32 rows (i, j, Sa, Sb, Sc), grouped by i (CSR order)
 *   d_vectors   [n_pairs,3] fp32 D = r_j - r_i + S.cell   (may be NULL)
 *   d_workspace: pet_nl_workspace_bytes(n) bytes. */
int64_t pet_nl_workspace_bytes(int64_t n_atoms);
int pet_nl_build(const float* d_positions, const float* h_cell, const int32_t* h_pbc,
                 int64_t n_atoms, float cutoff, void* d_workspace, int32_t* d_pairs,
                 float* d_vectors, int64_t capacity, int64_t* n_pairs, void* stream);

/* All systems of a batch in one set of launches (what the reference does per system inside DataLoader workers,
 * utils/neighbor_lists.py:13-46): positions concatenated, system s owning atoms h_first_atom[s] .. h_first_atom[s+1]-1,
 * h_cells [S,9] / h_pbc [S,3] on the host. Pairs carry GLOBAL atom indices (the offsets concatenate_structures adds,
 * structures.py:81-85), grouped by centre. d_pairs = NULL counts only. With d_pairs: one call does everything when
 * `capacity` suffices; otherwise PET_ERR_ARGUMENT and *n_pairs tells the size to retry with. One device->host
 * read-back per call (two when a system has an open boundary: its bounding box). */
int64_t pet_nl_batch_workspace_bytes(int64_t n_atoms, int64_t n_systems);
int pet_nl_build_batch(const float* d_positions, const float* h_cells, const int32_t* h_pbc, const int64_t* h_first_atom,
                       int64_t n_systems, float cutoff, void* d_workspace, int32_t* d_pairs, float* d_vectors,
                       int64_t capacity, int64_t* n_pairs, void* stream);

/* ---- preprocess (replaces PETBackend.preprocess, backend.py:238) --------------- */
int64_t pet_graph_workspace_bytes(int64_t n_nodes, int64_t n_edges_in);
/* Builds edge vectors, the non-strict filter, cutoff factors, the CSR (= NEF slot)
 * order and the ij->ji map.  Inputs as in compute_batch_tensors
 * (pet/modules/structures.py:115-131): d_positions [N,3], d_cells [S,3,3],
 * d_centers/d_neighbors [E] int32 (global atom indices), d_cell_shifts [E,3] int32,
 * d_species [N] int32 atomic numbers, d_system_indices [N] int32.
 * ONE device->host read-back (kept edges, max neighbours, validation counters) happens
 * here, like the reference's int(torch.max(num_neighbors)) (structures.py:292); nothing
 * downstream (forward, reverse passes) synchronises with the host. The read-back is a one-wave
 * kernel into a pinned mailbox that the calling thread polls (under 10 us against 27 us for
 * hipMemcpyAsync + hipStreamSynchronize: it sits on the critical path of a small box's MD step).
 * A list that arrives ordered by centre with nothing to drop skips the radix sort: the first build
 * of a process asks the device, later builds assume what the build before them found and verify
 * it with the read-back above (a wrong guess costs a second build, never a wrong graph).
 * Errors: PET_ERR_GRAPH if a kept edge (i, j, S) has no partner (j, i, -S) -- every consumer
 * gathers through the ij->ji map, the reference's get_corresponding_edges (nef.py:88-166)
 * has the same precondition; PET_ERR_ARGUMENT for indices outside [0, N), atomic numbers
 * outside the model's atomic_types, or system_indices that are not non-decreasing runs
 * inside [0, n_systems). */
int pet_graph_build(const pet_model_t* m, const float* d_positions, const float* d_cells,
                    const int32_t* d_centers, const int32_t* d_neighbors,
                    const int32_t* d_cell_shifts, const int32_t* d_species,
                    const int32_t* d_system_indices, int64_t n_nodes, int64_t n_edges_in,
                    int64_t n_systems, void* d_workspace, int64_t workspace_bytes,
                    pet_graph_t** out, void* stream);
void pet_graph_destroy(pet_graph_t* g);
int64_t pet_graph_num_edges(const pet_graph_t* g);     /* kept edges E            */
int32_t pet_graph_max_neighbors(const pet_graph_t* g); /* M of the NEF grid       */
/* The 12 batch_data tensors of backend.py:328-341 in the reference's padded NEF
 * layout (pads replicate kept edge 0). int64 where the reference
 * has int64, uint8 for the bool mask. Any output pointer may be NULL. */
int pet_graph_export_batch(const pet_graph_t* g,
                           int64_t* d_element_indices_nodes,     /* [N]     */
                           int64_t* d_element_indices_neighbors, /* [N,M]   */
                           float* d_edge_vectors,                /* [N,M,3] */
                           float* d_edge_distances,              /* [N,M]   */
                           uint8_t* d_padding_mask,              /* [N,M]   */
                           int64_t* d_reverse_neighbor_index,    /* [N,M]   */
                           float* d_cutoff_factors,              /* [N,M]   */
                           float* d_atomic_cutoffs_stats,        /* [N]     */
                           int64_t* d_centers,                   /* [E]     */
                           int64_t* d_neighbors,                 /* [E]     */
                           int64_t* d_nef_to_edges_neighbor,     /* [E]     */
                           int64_t* d_cell_shifts,               /* [E,3]   */
                           void* stream);
/* CSR views for callers that want the unpadded layout (device pointers into the
 * graph workspace): rowptr [N+1], ctr/nbr/rev [E] int32. */
int pet_graph_csr(const pet_graph_t* g, const int32_t** d_rowptr, const int32_t** d_ctr,
                  const int32_t** d_nbr, const int32_t** d_rev);

/* ONE box over several GPUs with a per-layer exchange (not in the reference, which leaves large systems to the MD engine's
 * domain decomposition, pet/model.py:1004-1017). A rank's graph holds its OWNED centres with all
 * their edges plus, for every foreign neighbour j of an owned atom i, the reverse edge (j -> i) as a "ghost" row. After the
 * transformer layers of each GNN layer the library gathers the "export" rows (owned centre, foreign neighbour) of the edge
 * tokens into d_export_buf, calls fn(user, 0, layer) -- the caller's all-to-all, which must fill d_ghost_buf with the
 * owners' rows, stream-ordered on the call's stream -- and scatters d_ghost_buf into the ghost rows before the combination
 * stage reads e[reversed] (backend.py:559-575). In the reverse pass the adjoints that land on ghost rows are gathered into
 * d_ghost_buf, zeroed locally, fn(user, 1, layer) carries them home and d_export_buf is ADDED to the export rows. Halo
 * atoms need no complete neighbourhood: one cutoff of halo instead of (layers + 1). Buffers [n, d_pet] fp32; row lists are
 * CSR rows of this graph (pet_graph_csr). Default model size, PreLN + feedforward, inference + forces. fn == NULL: off. */
typedef int (*pet_exchange_fn)(void* user, int direction, int layer);
int pet_graph_set_exchange(pet_graph_t* g, const int32_t* d_export_rows, int64_t n_export, const int32_t* d_ghost_rows,
                           int64_t n_ghost, float* d_export_buf, float* d_ghost_buf, pet_exchange_fn fn, void* user);
/* system_conditioning (backend.py:375-378: batch_data["charge"], ["spin_multiplicity"], ["system_indices"]): per-system
 * total charge and spin multiplicity (2S + 1) for the forward passes on this graph handle. d_system_indices [N] may be
 * NULL for a pet_graph_build handle (it has them); the three device arrays must stay alive while the handle is used.
 * Values outside [-max_charge, max_charge] / [1, max_spin_multiplicity] are the caller's to reject (conditioning.py:54-80
 * does it on the host); the kernels clamp them.
 * Training (pet_backward_train / pet_backward_train2) sums the node-feature adjoints per system for the conditioning
 * parameters: the system indices must then be non-decreasing (what concatenate_structures produces). */
int pet_graph_set_conditioning(pet_graph_t* g, const int64_t* d_charge, const int64_t* d_spin_multiplicity,
                               const int64_t* d_system_indices, int64_t n_systems);

/* ---- features + predict + gradient -------------------------------------------- */
/* Activation workspace for one forward (+ saved tensors for the backward). */
int64_t pet_forward_workspace_bytes(const pet_model_t* m, int64_t n_nodes, int64_t n_edges);
/* The same for a built graph. The tuned kernels are one compiled model size (d_pet=128, d_node=256, d_feedforward=256,
 * d_head=128, num_heads=8) with at most 127 neighbours per atom; every other model size (the reference is size-generic,
 * pet/documentation.py:196-213; d_node == d_pet follows transformer.py:189-201) AND any graph with a denser atom (the
 * reference pads to any max(num_neighbors), pet/modules/structures.py:292-294) runs on a size-generic path with its own,
 * larger workspace layout. pet_forward_workspace_bytes answers for the model alone; a caller that may meet dense graphs
 * sizes the workspace with this function. Training of other sizes / PostLN / residual models runs on that path too
 * (pet_train_workspace_bytes answers for the model, pet_train_workspace_bytes_for / pet_train2_workspace_bytes_for for a
 * built graph: a model of the compiled size trains on a graph with a denser atom through that path as well). */
int64_t pet_forward_workspace_bytes_for(const pet_model_t* m, const pet_graph_t* g);
/* calculate_features + predict for the fused target (the heads uploaded under the name "@", one property):
 *   d_atomic [N]       per-atom prediction (node + sum of cutoff-weighted edge terms); NULL = features only
 *   d_node_features [N,d_node] / d_edge_features [E,d_pet] (CSR rows): optional copies
 *   of the backbone features (backend.py:585-586) -- may be NULL.
 * save_for_backward != 0 keeps what pet_backward needs in the workspace. */
int pet_forward(const pet_model_t* m, const pet_graph_t* g, void* d_workspace,
                int64_t workspace_bytes, int save_for_backward, float* d_atomic,
                float* d_node_features, float* d_edge_features, void* stream);

/* ---- the three PETBackend calls as functions of their arguments ---------------------------------
 * pet_graph_build + pet_graph_export_batch is preprocess (backend.py:238). The two calls below take a
 * batch_data dictionary as the reference's calculate_features / predict do (backend.py:344, :420): a graph handle
 * made FROM the padded NEF tensors, whoever produced them -- this library's preprocess, the reference's, or a caller
 * that edited them in between (pet/tests/test_backend.py:122-150 is the executable spec).
 *   d_element_indices_nodes [N] i64, d_element_indices_neighbors [N,M] i64, d_edge_vectors [N,M,3] f32,
 *   d_edge_distances [N,M] f32, d_padding_mask [N,M] u8, d_reverse_neighbor_index [N,M] i64, d_cutoff_factors [N,M] f32.
 * Real slots are a prefix of each row (nef.py:63-85); CSR row of (i, slot) = rowptr[i] + slot. The handle holds
 * what features / heads and their adjoints read (rowptr, ctr, nbr, rev, species, geometry, cutoff factors); it has
 * no positions, so the geometry adjoint needs the pet_graph_build handle. One device->host read-back.
 * predict only reads the row structure and the cutoff factors: every pointer except d_padding_mask and
 * d_cutoff_factors may be NULL for a handle that is only passed to pet_predict / pet_predict_backward. */
int64_t pet_graph_from_batch_workspace_bytes(int64_t n_nodes, int64_t max_neighbors);
int pet_graph_from_batch(const int64_t* d_element_indices_nodes, const int64_t* d_element_indices_neighbors,
                         const float* d_edge_vectors, const float* d_edge_distances, const uint8_t* d_padding_mask,
                         const int64_t* d_reverse_neighbor_index, const float* d_cutoff_factors, int64_t n_nodes,
                         int64_t max_neighbors, void* d_workspace, int64_t workspace_bytes, pet_graph_t** out,
                         void* stream);
/* calculate_features alone: pet_forward with d_atomic = NULL (no heads), d_node_features / d_edge_features out.
 *
 * predict (backend.py:420-494) for ONE (target, readout layer, block) on the features the caller passes:
 *   heads node_heads.<target>.<layer> / edge_heads.<target>.<layer> (backend.py:651-687), last layers
 *   node_last_layers.<target>.<layer>.<block> / edge_... with P properties (:689-777), edge predictions weighted by the
 *   cutoff factor and summed per atom (:762-772), node + edge (:468-476). Summing over readout layers and blocks is
 *   the caller's loop, as in the reference.
 *   target / block: the names the heads were uploaded with ("@" for the fused target of pet_forward);
 *   d_node_features [N, d_node], d_edge_features [E, d_pet] (CSR rows), d_cutoff_factors [E] (NULL = the graph's),
 *   d_atomic [N, P] out; d_node_hidden [N, d_head] / d_edge_hidden [E, d_head] optional outs: the last-layer features
 *   predict returns as its 2nd / 3rd value; d_scratch: pet_predict_scratch_floats(N, E) floats. */
int32_t pet_model_block_properties(const pet_model_t* m, const char* target, int32_t readout_layer, const char* block);
int64_t pet_predict_scratch_floats(int64_t n_nodes, int64_t n_edges);
int pet_predict(const pet_model_t* m, const pet_graph_t* g, const char* target, int32_t readout_layer, const char* block,
                const float* d_node_features, const float* d_edge_features, const float* d_cutoff_factors,
                float* d_atomic, float* d_node_hidden, float* d_edge_hidden, float* d_scratch, void* stream);
/* Its adjoint, from the same inputs (the head MLPs are recomputed; nothing is read from a forward workspace):
 *   d_grad_atomic [N, P] -> d_grad_node_features [N, d_node], d_grad_edge_features [E, d_pet], d_grad_cutoff [E]. */
int pet_predict_backward(const pet_model_t* m, const pet_graph_t* g, const char* target, int32_t readout_layer,
                         const char* block, const float* d_node_features, const float* d_edge_features,
                         const float* d_cutoff_factors, const float* d_grad_atomic, float* d_grad_node_features,
                         float* d_grad_edge_features, float* d_grad_cutoff, float* d_scratch, void* stream);

/* Auxiliary per-atom outputs of pet/model.py:730-875 ("feature" and "mtt::aux::<target>_last_layer_features"),
 * from the backbone features pet_forward returned (same graph):
 *   d_feature [N, d_node + d_pet]          = [ node features | sum over edges of cutoff_factor * edge features ]
 *                                            (model.py:750-755; feedforward featuriser: the last GNN layer)
 *   d_last_layer_features [N, 2 * d_head]  = [ node-head hidden | sum over edges of cutoff_factor * edge-head hidden ]
 *                                            (model.py:795-812): the inputs of the target's last Linear layers
 * Either may be NULL. d_scratch: n_edges * d_head + max(n_edges, n_nodes) floats, needed for d_last_layer_features. */
int pet_aux_outputs(const pet_model_t* m, const pet_graph_t* g, const float* d_node_features,
                    const float* d_edge_features, float* d_feature, float* d_last_layer_features, float* d_scratch,
                    void* stream);
/* Reverse pass of the last pet_forward on (m, g, workspace):
 *   d_grad_atomic [N] = dL/d(atomic prediction) (ones => L = total energy),
 *   d_grad_positions [N,3] = dL/dR  (what compute_gradient returns; force = -grad),
 *   d_grad_cells [S,3,3] = dL/dcell through the S.cell term (may be NULL). */
int pet_backward(const pet_model_t* m, const pet_graph_t* g, void* d_workspace,
                 int64_t workspace_bytes, const float* d_grad_atomic,
                 float* d_grad_positions, float* d_grad_cells, void* stream);
/* The same reverse pass split at the three PETBackend calls, so that torch autograd can chain
 * preprocess -> calculate_features -> predict as three nodes (pet/tests/test_backend.py takes
 * autograd.grad of the summed prediction w.r.t. positions and a strain tensor). Edge-shaped
 * gradients are CSR rows (row p = slot p - rowptr[ctr[p]] of atom ctr[p]).
 *   predict^T   : d_grad_atomic [N] -> d_grad_node_features [N,d_node], d_grad_edge_features
 *                 [E,d_pet], d_grad_cutoff [E] (through the cutoff weight of backend.py:768-772)
 *   features^T  : (d_grad_node_features, d_grad_edge_features) -> d_grad_geometry [E,4] =
 *                 d/d(vx,vy,vz,dist) and d_grad_cutoff [E] (through the attention key bias)
 *   preprocess^T: (d_grad_geometry, d_grad_cutoff) -> d_grad_positions [N,3], d_grad_cells */
int pet_backward_predict(const pet_model_t* m, const pet_graph_t* g, void* d_workspace,
                         int64_t workspace_bytes, const float* d_grad_atomic,
                         float* d_grad_node_features, float* d_grad_edge_features,
                         float* d_grad_cutoff, void* stream);
int pet_backward_features(const pet_model_t* m, const pet_graph_t* g, void* d_workspace,
                          int64_t workspace_bytes, const float* d_grad_node_features,
                          const float* d_grad_edge_features, float* d_grad_geometry,
                          float* d_grad_cutoff, void* stream);
int pet_backward_geometry(const pet_model_t* m, const pet_graph_t* g, void* d_workspace,
                          int64_t workspace_bytes, const float* d_grad_geometry,
                          const float* d_grad_cutoff, float* d_grad_positions, float* d_grad_cells,
                          void* stream);

/* Every readout layer at once (backend.py:93-119: the residual featuriser reads out the features of EVERY GNN layer,
 * the feedforward one only the last: pet_model_num_readout_layers = num_gnn_layers or 1).
 *   pet_forward_layers: h_node_features[l] -> device [N,d_node], h_edge_features[l] -> device [E,d_pet] (CSR rows), two
 *     HOST arrays of n_layers device pointers (calculate_features' two lists, backend.py:344-418); no heads.
 *   pet_backward_features_layers: its adjoint for one gradient per returned tensor (NULL entries = zero). */
int32_t pet_model_num_readout_layers(const pet_model_t* m);
int pet_forward_layers(const pet_model_t* m, const pet_graph_t* g, void* d_workspace, int64_t workspace_bytes,
                       int save_for_backward, float* const* h_node_features, float* const* h_edge_features,
                       int32_t n_layers, void* stream);
int pet_backward_features_layers(const pet_model_t* m, const pet_graph_t* g, void* d_workspace, int64_t workspace_bytes,
                                 const float* const* h_grad_node_features, const float* const* h_grad_edge_features,
                                 int32_t n_layers, float* d_grad_geometry, float* d_grad_cutoff, void* stream);

/* preprocess^T without a forward workspace (the autograd node of preprocess on its own):
 * d_scratch: 4 * n_edges floats. Needs the pet_graph_build handle (positions, shifts). */
int pet_geometry_backward(const pet_model_t* m, const pet_graph_t* g, const float* d_grad_geometry,
                          const float* d_grad_cutoff, float* d_grad_positions, float* d_grad_cells, float* d_scratch,
                          void* stream);

/* ---- training step (trainer.py:391-480) ---------------- */
/* Served architectures: transformer_type = PreLN with the feed-forward featuriser; normalization RMSNorm or LayerNorm,
 * activation SwiGLU or SiLU, system conditioning on or off. PostLN / residual models: PET_ERR_UNSUPPORTED from
 * pet_forward(save_for_backward = 2) and from the two training reverse passes. */
/* The model owns one gradient slot per uploaded parameter (same numel, fp32).
 * pet_model_zero_grad allocates (first call) and clears them -- optimizer.zero_grad(). */
int pet_model_zero_grad(pet_model_t* m, void* stream);
/* Copy the accumulated gradient of parameter `key` (same key as pet_model_set_param) into d_dst. */
int pet_model_get_grad(const pet_model_t* m, const char* key, float* d_dst, int64_t numel,
                       void* stream);
/* Current value of parameter `key` (after optimizer steps). */
int pet_model_get_param(const pet_model_t* m, const char* key, float* d_dst, int64_t numel,
                        void* stream);
/* The gradient slots as ONE flat fp32 buffer of pet_model_num_params elements (upload order):
 * direction 0 copies it out to d_flat, 1 copies d_flat back in. This is the bucket the RCCL
 * gradient all-reduce runs on (torch DDP's role, pet/trainer.py:344-345): 11.6 MB, one collective. */
int pet_model_flat_grad(pet_model_t* m, float* d_flat, int64_t numel, int direction, void* stream);
/* clip_grad_norm_(max_grad_norm; <= 0: off) + torch.optim.Adam (weight_decay < 0) or AdamW
 * (weight_decay >= 0) on every parameter, `step` counted from 1 (pet/trainer.py:367-376,463-467),
 * then re-packs the weights. d_grad_norm (device, may be NULL) receives the pre-clip total norm. */
int pet_adam_step(pet_model_t* m, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float max_grad_norm, int64_t step, float* d_grad_norm, void* stream);
/* Adam's first / second moments as two flat fp32 buffers of pet_model_num_params elements in upload order (the layout
 * of pet_model_flat_grad): direction 0 copies them out, 1 copies them in. With the step counter this is the
 * optimizer_state_dict a checkpoint keeps (pet/trainer.py:697-717, trainer checkpoint v15). */
int pet_optimizer_state(pet_model_t* m, float* d_m, float* d_v, int64_t numel, int direction, void* stream);
/* Declare parameter `key` (uploaded as a tensor stacked on itself, see pet_model_set_param) TIED: pet_adam_step sums
 * the gradient slots of its two halves, counts the parameter once in the clipping norm and gives both halves the same
 * update, so that they stay equal. */
int pet_model_tie_halves(pet_model_t* m, const char* key);
/* Workspace for pet_forward(save_for_backward = 2) + pet_backward_train. */
int64_t pet_train_workspace_bytes(const pet_model_t* m, int64_t n_nodes, int64_t n_edges);
int64_t pet_train_workspace_bytes_for(const pet_model_t* m, const pet_graph_t* g);   /* graph-aware (dense atoms) */
/* Reverse pass of loss.backward() for L with dL/d(atomic prediction) = d_grad_atomic [N]:
 * accumulates dL/dtheta into the model's gradient slots; d_grad_positions [N,3] (and d_grad_cells)
 * may be NULL. Needs pet_forward(save_for_backward = 2) on a training workspace. */
int pet_backward_train(const pet_model_t* m, const pet_graph_t* g, void* d_workspace,
                       int64_t workspace_bytes, const float* d_grad_atomic,
                       float* d_grad_positions, float* d_grad_cells, void* stream);
/* Second-order reverse pass for losses on dE/dR (forces), i.e. what loss.backward() does through the
 * create_graph=True gradient of evaluate_model(is_training=True) (pet/trainer.py:417-462):
 *   d_lambda_atomic [N]  seeds the force pass used (grad_outputs of autograd.grad: ones),
 *   d_nu_atomic [N]      dL/d(atomic prediction) of the energy term (may be NULL = 0),
 *   d_u [N,3]            dL/d(dE/dR),
 *   d_tangent_atomic [N] optional out: directional derivative of every atomic prediction along dR = u.
 * Accumulates dL/dtheta into the gradient slots. Needs pet_forward(save_for_backward = 2) on a
 * training workspace plus a second workspace of pet_train2_workspace_bytes. */
int64_t pet_train2_workspace_bytes(const pet_model_t* m, int64_t n_nodes, int64_t n_edges);
int64_t pet_train2_workspace_bytes_for(const pet_model_t* m, const pet_graph_t* g);  /* graph-aware (dense atoms) */
int pet_backward_train2(const pet_model_t* m, const pet_graph_t* g, void* d_workspace,
                        int64_t workspace_bytes, void* d_workspace2, int64_t workspace2_bytes,
                        const float* d_lambda_atomic, const float* d_nu_atomic, const float* d_u,
                        float* d_tangent_atomic, void* stream);
/* The same with a tangent of the CELLS as well (a stress / strain-gradient term in the loss, utils/evaluate_model.py:
 * 305-321: positions @ strain, cell @ strain under create_graph): the sweep runs along (dR, dcell) = (d_u [N,3],
 * d_u_cell [S,3,3]) with u = dL/d(dE/dR), u_cell = dL/d(dE/dcell); d_u_cell = NULL is pet_backward_train2. */
int pet_backward_train2_cell(const pet_model_t* m, const pet_graph_t* g, void* d_workspace,
                             int64_t workspace_bytes, void* d_workspace2, int64_t workspace2_bytes,
                             const float* d_lambda_atomic, const float* d_nu_atomic, const float* d_u,
                             const float* d_u_cell, float* d_tangent_atomic, void* stream);
/* Per-system sum (utils/sum_over_atoms.py:10-48): d_out[S] = sum_{atoms of s} d_atomic. */
int pet_sum_over_atoms(const pet_graph_t* g, const float* d_atomic, float* d_out, void* stream);

/* ---- profiling hooks used by bench.py ------------------------------------------ */
/* When enabled, every kernel launch of pet_forward/pet_backward is bracketed with HIP
 * events on the launch stream; pet_profile_report fills name / total ms / calls / algorithmic
 * FLOPs / algorithmic HBM bytes per stage. */
int pet_profile_enable(int on);
/* Restrict the event bracketing to one stage name (e.g. "emlp_bwd"); NULL or "" = all. */
int pet_profile_select(const char* stage);
int pet_profile_reset(void);
int pet_profile_report(int max_entries, char (*names)[64], double* total_ms, int64_t* calls,
                       double* flops, double* bytes, int* n_entries);
/* Runtime switches used by tests / the benchmark (every setting meets the same parity bar):
 *   "side_stream" 1 = node-feature chain on a second HIP stream (default); that stream is created one priority level below the
 *                 caller's (environment PET_HIP_SIDE_PRIO = same | high overrides; PET_HIP_SIDE = 0 disables the stream)
 *   "trr"         1 = transposed register-resident row kernels on f16x3 split-operand products (default); 0 = the LDS-tile
 *                 kernels for the transformer layers (the one fallback generation, also the transformer-layer path of
 *                 PostLN models). The combination stage has one implementation (the software-pipelined TRR kernel).
 *   "attn_fused"  bits: 1 = the per-atom fused attention block in the forward (norm -> QKV -> soft-max attention -> output
 *                 projection in one kernel; Q, K, V and the attention output never reach HBM; csrc/pet_ablk.hip), 2 = its
 *                 adjoint (recomputes Q, K, V from the layer input), 4 = whatever the graph's size; default 3: graphs of
 *                 fewer than 3 840 attention tiles (fewer than about 4 700 atoms: latency-bound there) and graphs in which more
 *                 than 5 % of the atoms have more than 32 tokens keep the three-kernel form, as do training forwards,
 *                 graphs with an atom of more than 64 tokens and PostLN models. 0 = the three-kernel form everywhere.
 *   "emlp_s"      the edge MLP and its adjoint -- and, in inference, the edge head and its adjoint (csrc/pet_head_s.hip), the compress
 *                 adjoint (pet_compress_s.hip), the combination stage and its adjoint (pet_comb_s.hip, pet_comb_bwd_s.hip), from
 *                 16 384 atoms on the node-row Linear layers around the attention block (pet_center_s.hip), and in training the
 *                 generic GEMMs of the second-order pass (so_rows_s.hip) -- as two desynchronised four-wave workgroups per CU on one-accumulator products
 *                 with a workgroup-shared weight ring (csrc/pet_emlp_s.hip, rows_s.h; the adjoint RECOMPUTES the SwiGLU pre-activations, so an inference forward does not
 *                 store them): 1 = for graphs of at least 28 672 edge rows (default), v > 1 = from v rows on, 0 = never
 *                 (the one-wave-per-SIMD pipelined kernels everywhere). A forward that ran without saving can only be followed
 *                 by the recomputing adjoint: flipping the switch in between makes pet_backward fail (PET_ERR_ARGUMENT).
 *   "trr_compress" bit mask of f16x3 TRR kernels replacing LDS-tile ones: 1 compress (+adjoint), 2 edge head (+adjoint);
 *                  default 3; 0 = LDS-tile kernels
 *   "node_planes" 1 = node-row kernels k_node2 / k_node2w / k_node_bwd2 (coalesced tiles, fp16 planes; default), 0 = k_node /
 *                 k_swiglu_bwd; 2 forces 32 rows per workgroup (by default up to 16 384 atoms)
 *   "so_trr"      1 = generic training GEMMs with K = 128 or n_out = 128 as TRR kernels (default); 0 = LDS-tile k_gemm_h
 *   "soap_ps_mfma" 1 = SOAP-BPNN power spectrum and its adjoint on the fp32 matrix core (default); 0 = the VALU kernels
 *   "node_split"  1 = graphs of at most 4 096 atoms: the node update and its adjoint run the four hidden chunks of a 32-row
 *                 tile on four workgroups (partials through a temporary, the last arrival finishes the tile) -- default;
 *                 0 = one workgroup per tile
 *   "center_fused" 1 = the node-update kernel also writes the next attention layer's centre tokens (default); 0 = k_center
 *   "sorted_shortcut" 1 = pet_graph_build skips the radix sort of the edges when the neighbour list is already ordered by centre
 *                 with no edge to drop (the first build asks the device, later ones assume the previous build's answer and verify
 *                 it with the build's one read-back; default); 0 = always sort
 *   "dxf_fused"   1 = inference adjoint on one rank: dXF[p] = dM[p] + dcat[p][:D] + dcat[rev[p]][D:] formed inside the
 *                 combination adjoint (first two terms) and the edge-MLP adjoint (the gather) -- default; 0 = k_dxf launch
 *   "train_bf16"  1 = the GEMMs of the second-order pass and the weight-gradient GEMMs keep ONE 16-bit MFMA term per product
 *                 (fp16 / bf16 high planes, fp32 accumulation: BASELINE configs[2]'s "bf16 MFMA MLPs"; gradients within
 *                 ~1e-3, NOT the 1e-5 parity mode; 20-step loss curve in tests/test_gpu_train.py); default 0
 *   "wgrad_bf16"  1 = weight-gradient GEMMs of the training passes as bf16x3 split-operand products (default); 0 = fp32 MFMA
 *   "so_f16x3"    1 = generic GEMMs of the second-order (training) pass as f16x3 (default); 0 = fp32 MFMA
 *   (removed in round 4 with the kernels they selected: "emlp_pipe", "emlp_bwd_pipe", "comb_pipe", "comb_bwd_pipe",
 *   "emlp_recompute", "line_stores", "lds_w"; in round 6: "attn_lds" with the two staged-adjoint instantiations only it reached,
 *   "tile_f16x3" (the fp32 fallback of the LDS-tile kernels remains for weights without fp16 planes), "soap_fused" with the
 *   first-generation fused SOAP kernels, the A/B value 3 of "node_planes" with its kernel)
 *   "soap_mfma"   1 = SOAP-BPNN LayerNorm + MLP tail on MFMA (default)
 *   "soap_packed" 1 = SOAP-BPNN inference (legacy / per-species networks) stores the upper triangle of every power-spectrum block
 *                 only (p_l[a][b] = p_l[b][a]: 2 360 instead of 4 544 floats per atom for the default basis), LayerNorm statistics
 *                 weighted and the first Linear folded accordingly -- default; 0 = the full [N][S] layout (what the feature output,
 *                 the Alchemical centre encoding and the training pass use in any case)
 *   "soap_sorted" 1 = SOAP-BPNN tail GEMM on species-sorted atom tiles, one network per tile (default)
 *   "soap_pair"   1 = SOAP expansion one wave per atom, its adjoint one lane per pair (default); 0 = first generation
 * Unknown keys return PET_ERR_ARGUMENT. */
int pet_config_set(const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* PET_HIP_H */
