// TorchScript-visible wrapper around the C ABI of include/pet_hip.h:
//   torch.classes.pet_hip.PetHipModule  -- owns the packed weights, picklable (so torch.jit.save works),
//   .atomic_energies(positions, cells, centers, neighbors, cell_shifts, species, system_indices) -> [N, 1]
// differentiable w.r.t. positions and cells through a C++ autograd node that calls pet_forward / pet_backward,
// which is what an exported AtomisticModel needs (forces via torch.autograd.grad inside TorchScript,
// metatomic's evaluate path; reference: pet/model.py:416-537, utils/output_gradient.py:34-40).
// Host-side glue only: every FLOP still runs in the HIP kernels behind the C ABI. Plain C++ (no device code).
#include <c10/hip/HIPStream.h>
#include <torch/custom_class.h>
#include <torch/script.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/pet_hip.h"

namespace {

void check(int rc, const char* what) {
    TORCH_CHECK(rc == PET_OK, "libpet_hip: ", what, " failed (", rc, "): ", pet_last_error());
}
void* stream_of(const at::Tensor& t) {
    return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}
at::Tensor as_i32(const at::Tensor& t) { return t.to(at::kInt).contiguous(); }
at::Tensor as_f32(const at::Tensor& t) { return t.detach().to(at::kFloat).contiguous(); }

struct GraphHolder : torch::CustomClassHolder {  // one preprocess() result kept alive by the autograd node
    pet_graph_t* g = nullptr;
    at::Tensor graph_ws, fwd_ws, keep_alive[7];
    int64_t n_nodes = 0, n_systems = 0;
    ~GraphHolder() override {
        if (g) pet_graph_destroy(g);
    }
};

struct PetHipModule : torch::CustomClassHolder {
    std::vector<double> hypers;  // pet_hypers_t, field by field (see hypers_struct())
    std::vector<int64_t> atomic_types;
    std::vector<std::string> keys;  // reference state-dict keys with the target replaced by "@"
    std::vector<at::Tensor> tensors;
    pet_model_t* model = nullptr;
    int64_t model_device = -1;

    PetHipModule(std::vector<double> hypers_, std::vector<int64_t> atomic_types_, std::vector<std::string> keys_,
                 std::vector<at::Tensor> tensors_)
        : hypers(std::move(hypers_)), atomic_types(std::move(atomic_types_)), keys(std::move(keys_)),
          tensors(std::move(tensors_)) {
        TORCH_CHECK(hypers.size() == 16 || hypers.size() == 19 || hypers.size() == 20 || hypers.size() == 23,
                    "pet_hip: expected the first 16, 19, 20 or all 23 fields of pet_hypers_t");
        TORCH_CHECK(keys.size() == tensors.size(), "pet_hip: keys / tensors length mismatch");
    }
    ~PetHipModule() override {
        if (model) pet_model_destroy(model);
    }

    pet_hypers_t hypers_struct() const {
        pet_hypers_t h{};
        h.cutoff = (float)hypers[0]; h.cutoff_width = (float)hypers[1]; h.cutoff_function = (int32_t)hypers[2];
        h.d_pet = (int32_t)hypers[3]; h.d_head = (int32_t)hypers[4]; h.d_node = (int32_t)hypers[5];
        h.d_feedforward = (int32_t)hypers[6]; h.num_heads = (int32_t)hypers[7];
        h.num_attention_layers = (int32_t)hypers[8]; h.num_gnn_layers = (int32_t)hypers[9];
        h.attention_temperature = (float)hypers[10]; h.nl_is_strict = (int32_t)hypers[11];
        h.n_species = (int32_t)hypers[12]; h.max_atomic_number = (int32_t)hypers[13];
        h.num_neighbors_adaptive = (float)hypers[14]; h.cutoff_width_adaptive = (float)hypers[15];
        if (hypers.size() >= 19) {
            h.normalization = (int32_t)hypers[16]; h.transformer_type = (int32_t)hypers[17];
            h.featurizer_type = (int32_t)hypers[18];
        }
        if (hypers.size() >= 20) h.adaptive_cutoff_method = (int32_t)hypers[19];
        TORCH_CHECK(hypers.size() < 23 || hypers[20] == 0.0,
                    "pet_hip: system conditioning needs the three PETBackend calls (batch_data carries charge / spin)");
        return h;
    }

    void ensure_model(const at::Tensor& like) {
        TORCH_CHECK(like.is_cuda(), "pet_hip runs on MI355X only: got a tensor on ", like.device(), " (no CPU path)");
        if (model && model_device == like.device().index()) return;
        if (model) pet_model_destroy(model);
        model = nullptr;
        pet_hypers_t h = hypers_struct();
        check(pet_model_create(&h, &model), "pet_model_create");
        void* st = stream_of(like);
        for (size_t i = 0; i < keys.size(); i++) {
            at::Tensor t = keys[i] == "species_to_species_index"
                               ? tensors[i].to(like.device(), at::kLong).contiguous()
                               : tensors[i].detach().to(like.device(), at::kFloat).contiguous();
            check(pet_model_set_param(model, keys[i].c_str(), t.data_ptr(), t.numel(), st), keys[i].c_str());
            c10::hip::getCurrentHIPStream(like.device().index()).synchronize();  // `t` may be a temporary
        }
        check(pet_model_finalize(model, st), "pet_model_finalize");
        model_device = like.device().index();
    }

    at::Tensor atomic_energies(const at::Tensor& positions, const at::Tensor& cells, const at::Tensor& centers,
                               const at::Tensor& neighbors, const at::Tensor& cell_shifts, const at::Tensor& species,
                               const at::Tensor& system_indices);
};

struct EnergyFn : torch::autograd::Function<EnergyFn> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& positions,
                              const at::Tensor& cells, c10::intrusive_ptr<PetHipModule> mod, const at::Tensor& centers,
                              const at::Tensor& neighbors, const at::Tensor& cell_shifts, const at::Tensor& species,
                              const at::Tensor& system_indices) {
        mod->ensure_model(positions);
        auto gh = c10::make_intrusive<GraphHolder>();
        void* st = stream_of(positions);
        gh->n_nodes = positions.size(0);
        gh->n_systems = cells.size(0);
        const int64_t e0 = centers.size(0);
        gh->keep_alive[0] = as_f32(positions); gh->keep_alive[1] = as_f32(cells);
        gh->keep_alive[2] = as_i32(centers); gh->keep_alive[3] = as_i32(neighbors);
        gh->keep_alive[4] = as_i32(cell_shifts); gh->keep_alive[5] = as_i32(species);
        gh->keep_alive[6] = as_i32(system_indices);
        auto bytes = at::TensorOptions().dtype(at::kByte).device(positions.device());
        gh->graph_ws = at::empty({pet_graph_workspace_bytes(gh->n_nodes, e0)}, bytes);
        check(pet_graph_build(mod->model, gh->keep_alive[0].data_ptr<float>(), gh->keep_alive[1].data_ptr<float>(),
                              gh->keep_alive[2].data_ptr<int32_t>(), gh->keep_alive[3].data_ptr<int32_t>(),
                              gh->keep_alive[4].data_ptr<int32_t>(), gh->keep_alive[5].data_ptr<int32_t>(),
                              gh->keep_alive[6].data_ptr<int32_t>(), gh->n_nodes, e0, gh->n_systems,
                              gh->graph_ws.data_ptr(), gh->graph_ws.numel(), &gh->g, st),
              "pet_graph_build");
        gh->fwd_ws = at::empty({pet_forward_workspace_bytes_for(mod->model, gh->g)}, bytes);
        at::Tensor atomic = at::empty({gh->n_nodes}, positions.options().dtype(at::kFloat));
        check(pet_forward(mod->model, gh->g, gh->fwd_ws.data_ptr(), gh->fwd_ws.numel(), 1, atomic.data_ptr<float>(),
                          nullptr, nullptr, st),
              "pet_forward");
        ctx->saved_data["graph"] = gh;
        ctx->saved_data["module"] = mod;
        ctx->saved_data["pos_dtype"] = (int64_t)positions.scalar_type();
        ctx->saved_data["cell_dtype"] = (int64_t)cells.scalar_type();
        return atomic.unsqueeze(1).to(positions.scalar_type());
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                   torch::autograd::variable_list grad_out) {
        auto gh = ctx->saved_data["graph"].toCustomClass<GraphHolder>();
        auto mod = ctx->saved_data["module"].toCustomClass<PetHipModule>();
        at::Tensor ga = grad_out[0];
        TORCH_CHECK(!ga.requires_grad(), "pet_hip: double backward (create_graph=True) is not available through the "
                                         "TorchScript op; train through metatrain_amd.pet (PETBackend / TrainStep)");
        ga = as_f32(ga.reshape({-1}));
        at::Tensor gpos = at::empty({gh->n_nodes, 3}, ga.options());
        at::Tensor gcell = at::empty({gh->n_systems, 3, 3}, ga.options());
        check(pet_backward(mod->model, gh->g, gh->fwd_ws.data_ptr(), gh->fwd_ws.numel(), ga.data_ptr<float>(),
                           gpos.data_ptr<float>(), gcell.data_ptr<float>(), stream_of(ga)),
              "pet_backward");
        const auto pd = (at::ScalarType)ctx->saved_data["pos_dtype"].toInt();
        const auto cd = (at::ScalarType)ctx->saved_data["cell_dtype"].toInt();
        return {gpos.to(pd), gcell.to(cd), at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor PetHipModule::atomic_energies(const at::Tensor& positions, const at::Tensor& cells,
                                         const at::Tensor& centers, const at::Tensor& neighbors,
                                         const at::Tensor& cell_shifts, const at::Tensor& species,
                                         const at::Tensor& system_indices) {
    return EnergyFn::apply(positions, cells, c10::intrusive_ptr<PetHipModule>::reclaim_copy(this), centers, neighbors,
                           cell_shifts, species, system_indices);
}


// =====================================================================================================================
// PetHipBackend: the three calls of PETBackend (backend.py:238 / :344 / :420) as TorchScript-visible methods whose
// results are functions of their ARGUMENTS only, each with its own C++ autograd node, so that a torch.nn.Module that
// mirrors the reference's PETBackend can be torch.jit.script-ed (utils/testing/torchscript.py:39-75) and exported.
// The parameters are passed in on every call (the module owns them as ordinary nn.Parameters under the reference's
// state-dict names); the packed device model is rebuilt when any of them changed (data pointer / version counter).
// =====================================================================================================================
struct PetHipBackend : torch::CustomClassHolder {
    std::vector<double> hypers;      // the 16 fields of pet_hypers_t + [16] activation == "SiLU"
    std::vector<int64_t> atomic_types;
    std::vector<std::string> keys;   // state-dict keys of the float parameters, in the order `params` arrive
    pet_model_t* model = nullptr;
    int64_t model_device = -1;
    std::vector<std::pair<void*, int64_t>> stamp;
    int64_t generation = 0;  // bumped by every re-upload: a backward checks that its forward saw the same weights

    PetHipBackend(std::vector<double> hypers_, std::vector<int64_t> atomic_types_, std::vector<std::string> keys_)
        : hypers(std::move(hypers_)), atomic_types(std::move(atomic_types_)), keys(std::move(keys_)) {
        TORCH_CHECK(hypers.size() == 17 || hypers.size() == 20 || hypers.size() == 21 || hypers.size() == 24,
                    "pet_hip: expected the first 16 fields of pet_hypers_t, the SiLU flag and (optionally) normalization, "
                    "transformer_type, featurizer_type, adaptive_cutoff_method");
    }
    ~PetHipBackend() override {
        if (model) pet_model_destroy(model);
    }

    pet_hypers_t hypers_struct() const {
        pet_hypers_t h{};
        h.cutoff = (float)hypers[0]; h.cutoff_width = (float)hypers[1]; h.cutoff_function = (int32_t)hypers[2];
        h.d_pet = (int32_t)hypers[3]; h.d_head = (int32_t)hypers[4]; h.d_node = (int32_t)hypers[5];
        h.d_feedforward = (int32_t)hypers[6]; h.num_heads = (int32_t)hypers[7];
        h.num_attention_layers = (int32_t)hypers[8]; h.num_gnn_layers = (int32_t)hypers[9];
        h.attention_temperature = (float)hypers[10]; h.nl_is_strict = (int32_t)hypers[11];
        h.n_species = (int32_t)hypers[12]; h.max_atomic_number = (int32_t)hypers[13];
        h.num_neighbors_adaptive = (float)hypers[14]; h.cutoff_width_adaptive = (float)hypers[15];
        if (hypers.size() >= 20) {
            h.normalization = (int32_t)hypers[17]; h.transformer_type = (int32_t)hypers[18];
            h.featurizer_type = (int32_t)hypers[19];
        }
        if (hypers.size() >= 21) h.adaptive_cutoff_method = (int32_t)hypers[20];
        if (hypers.size() >= 24) {
            h.system_conditioning = (int32_t)hypers[21]; h.max_charge = (int32_t)hypers[22];
            h.max_spin_multiplicity = (int32_t)hypers[23];
        }
        return h;
    }

    void ensure_model(const std::vector<at::Tensor>& params, const at::Tensor& like) {
        TORCH_CHECK(like.is_cuda(), "pet_hip runs on MI355X only: got a tensor on ", like.device(), " (no CPU path)");
        TORCH_CHECK(params.size() == keys.size(), "pet_hip: ", params.size(), " parameters for ", keys.size(), " keys");
        bool fresh = model && model_device == like.device().index() && stamp.size() == params.size();
        for (size_t i = 0; fresh && i < params.size(); i++)
            fresh = stamp[i].first == params[i].data_ptr() && stamp[i].second == (int64_t)params[i]._version();
        if (fresh) return;
        const bool rebuild = !model || model_device != like.device().index();
        if (rebuild) {
            if (model) pet_model_destroy(model);
            model = nullptr;
            pet_hypers_t h = hypers_struct();
            check(pet_model_create(&h, &model), "pet_model_create");
            // species_to_species_index from atomic_types (backend.py:63-71)
            at::Tensor table = at::full({(int64_t)h.max_atomic_number + 1}, -1, at::TensorOptions().dtype(at::kLong));
            for (size_t i = 0; i < atomic_types.size(); i++) table[atomic_types[i]] = (int64_t)i;
            table = table.to(like.device());
            check(pet_model_set_param(model, "species_to_species_index", table.data_ptr(), table.numel(), stream_of(like)),
                  "species table");
            c10::hip::getCurrentHIPStream(like.device().index()).synchronize();
        }
        void* st = stream_of(like);
        const bool silu = hypers[16] != 0.0;
        std::vector<at::Tensor> keep;
        for (size_t i = 0; i < keys.size(); i++) {
            at::Tensor t = params[i].detach().to(like.device(), at::kFloat).contiguous();
            if (silu && keys[i].find(".w_in.") != std::string::npos) t = at::cat({t, t}, 0).contiguous();
            check(pet_model_set_param(model, keys[i].c_str(), t.data_ptr(), t.numel(), st), keys[i].c_str());
            keep.push_back(t);
        }
        c10::hip::getCurrentHIPStream(like.device().index()).synchronize();  // the uploads read temporaries
        check(pet_model_finalize(model, st), "pet_model_finalize");
        model_device = like.device().index();
        stamp.clear();
        for (const auto& p : params) stamp.push_back({p.data_ptr(), (int64_t)p._version()});
        generation++;
    }
    // a backward that runs after a parameter update / another forward would pair the new weights with the
    // activations its forward saved: refuse instead of returning a silently inconsistent gradient
    void check_generation(int64_t seen) const {
        TORCH_CHECK(seen == generation, "pet_hip: the parameters were re-uploaded (optimizer step or device change) between "
                    "this node's forward and its backward; run the forward again before differentiating");
    }

    std::vector<at::Tensor> preprocess(std::vector<at::Tensor> params, const at::Tensor& positions, const at::Tensor& centers,
                                       const at::Tensor& neighbors, const at::Tensor& species, const at::Tensor& cells,
                                       const at::Tensor& cell_shifts, const at::Tensor& system_indices);
    std::vector<at::Tensor> calculate_features(std::vector<at::Tensor> params, const at::Tensor& element_indices_nodes,
                                               const at::Tensor& element_indices_neighbors, const at::Tensor& edge_vectors,
                                               const at::Tensor& edge_distances, const at::Tensor& padding_mask,
                                               const at::Tensor& reverse_neighbor_index, const at::Tensor& cutoff_factors,
                                               std::vector<at::Tensor> conditioning);
    std::vector<at::Tensor> predict(std::vector<at::Tensor> params, std::string target, int64_t readout_layer,
                                    std::string block, const at::Tensor& node_features, const at::Tensor& edge_features,
                                    const at::Tensor& padding_mask, const at::Tensor& cutoff_factors);
};

// CSR bookkeeping of a graph handle as torch index tensors (plumbing between the NEF grid and CSR rows)
struct CsrIndex {
    at::Tensor ctr, slot;  // [E] int64
};
static CsrIndex csr_index(pet_graph_t* g, int64_t n_nodes, const at::Tensor& like) {
    const int32_t *rowptr, *ctr, *nbr, *rev;
    check(pet_graph_csr(g, &rowptr, &ctr, &nbr, &rev), "pet_graph_csr");
    const int64_t e = pet_graph_num_edges(g);
    auto i32 = at::TensorOptions().dtype(at::kInt).device(like.device());
    at::Tensor rp = at::from_blob((void*)rowptr, {n_nodes + 1}, i32).to(at::kLong);
    at::Tensor c = at::from_blob((void*)ctr, {e}, i32).to(at::kLong);
    CsrIndex ix;
    ix.ctr = c;
    ix.slot = at::arange(e, c.options()) - rp.index_select(0, c);
    return ix;
}
static at::Tensor to_csr(const at::Tensor& nef, const CsrIndex& ix) {
    return nef.index({ix.ctr, ix.slot}).contiguous();
}
static at::Tensor to_nef(const at::Tensor& csr, const CsrIndex& ix, int64_t n, int64_t m) {
    std::vector<int64_t> shape = {n, m};
    for (int64_t d = 1; d < csr.dim(); d++) shape.push_back(csr.size(d));
    at::Tensor out = at::zeros(shape, csr.options());
    out.index_put_({ix.ctr, ix.slot}, csr);
    return out;
}

struct BatchGraph : torch::CustomClassHolder {  // a pet_graph_t + everything it points into
    pet_graph_t* g = nullptr;
    at::Tensor ws, fwd_ws;
    std::vector<at::Tensor> keep;
    CsrIndex ix;
    int64_t n_nodes = 0, n_systems = 0, m = 0;
    ~BatchGraph() override {
        if (g) pet_graph_destroy(g);
    }
};

// ---- preprocess: (positions, cells) -> edge_vectors, edge_distances, cutoff_factors (+ the nine index tensors) ----------
struct PreprocessFn : torch::autograd::Function<PreprocessFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& positions,
                                                  const at::Tensor& cells, c10::intrusive_ptr<PetHipBackend> be,
                                                  const at::Tensor& centers, const at::Tensor& neighbors,
                                                  const at::Tensor& species, const at::Tensor& cell_shifts,
                                                  const at::Tensor& system_indices) {
        auto bg = c10::make_intrusive<BatchGraph>();
        void* st = stream_of(positions);
        bg->n_nodes = positions.size(0);
        bg->n_systems = cells.size(0);
        const int64_t e0 = centers.size(0);
        bg->keep = {as_f32(positions), as_f32(cells), as_i32(centers), as_i32(neighbors), as_i32(cell_shifts),
                    as_i32(species), as_i32(system_indices)};
        auto bytes = at::TensorOptions().dtype(at::kByte).device(positions.device());
        bg->ws = at::empty({pet_graph_workspace_bytes(bg->n_nodes, e0)}, bytes);
        check(pet_graph_build(be->model, bg->keep[0].data_ptr<float>(), bg->keep[1].data_ptr<float>(),
                              bg->keep[2].data_ptr<int32_t>(), bg->keep[3].data_ptr<int32_t>(),
                              bg->keep[4].data_ptr<int32_t>(), bg->keep[5].data_ptr<int32_t>(),
                              bg->keep[6].data_ptr<int32_t>(), bg->n_nodes, e0, bg->n_systems, bg->ws.data_ptr(),
                              bg->ws.numel(), &bg->g, st),
              "pet_graph_build");
        const int64_t n = bg->n_nodes, m = pet_graph_max_neighbors(bg->g), e = pet_graph_num_edges(bg->g);
        bg->m = m;
        auto dev = positions.device();
        auto i64 = at::TensorOptions().dtype(at::kLong).device(dev), f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        at::Tensor el_nodes = at::empty({n}, i64), el_nbr = at::empty({n, m}, i64), ev = at::empty({n, m, 3}, f32),
                   ed = at::empty({n, m}, f32), mask = at::empty({n, m}, bytes), rni = at::empty({n, m}, i64),
                   cf = at::empty({n, m}, f32), stats = at::empty({n}, f32), ctr = at::empty({e}, i64),
                   nbr = at::empty({e}, i64), slot = at::empty({e}, i64), shifts = at::empty({e, 3}, i64);
        check(pet_graph_export_batch(bg->g, el_nodes.data_ptr<int64_t>(), el_nbr.data_ptr<int64_t>(), ev.data_ptr<float>(),
                                     ed.data_ptr<float>(), mask.data_ptr<uint8_t>(), rni.data_ptr<int64_t>(),
                                     cf.data_ptr<float>(), stats.data_ptr<float>(), ctr.data_ptr<int64_t>(),
                                     nbr.data_ptr<int64_t>(), slot.data_ptr<int64_t>(), shifts.data_ptr<int64_t>(), st),
              "pet_graph_export_batch");
        if (e > 0) bg->ix = csr_index(bg->g, n, positions);
        ctx->saved_data["graph"] = bg;
        ctx->saved_data["backend"] = be;
        ctx->saved_data["generation"] = be->generation;
        ctx->saved_data["pos_dtype"] = (int64_t)positions.scalar_type();
        ctx->saved_data["cell_dtype"] = (int64_t)cells.scalar_type();
        const auto dt = positions.scalar_type();
        std::vector<at::Tensor> out = {el_nodes, el_nbr, ev.to(dt), ed.to(dt), mask.to(at::kBool), rni, cf.to(dt),
                                       stats.to(dt), ctr.to(centers.scalar_type()), nbr.to(centers.scalar_type()), slot,
                                       shifts.to(cell_shifts.scalar_type())};
        ctx->mark_non_differentiable({out[0], out[1], out[4], out[5], out[7], out[8], out[9], out[10], out[11]});
        return out;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                   torch::autograd::variable_list go) {
        auto bg = ctx->saved_data["graph"].toCustomClass<BatchGraph>();
        auto be = ctx->saved_data["backend"].toCustomClass<PetHipBackend>();
        be->check_generation(ctx->saved_data["generation"].toInt());
        const auto pd = (at::ScalarType)ctx->saved_data["pos_dtype"].toInt();
        const auto cd = (at::ScalarType)ctx->saved_data["cell_dtype"].toInt();
        const int64_t n = bg->n_nodes, e = pet_graph_num_edges(bg->g);
        auto f32 = at::TensorOptions().dtype(at::kFloat).device(bg->ws.device());
        at::Tensor gpos = at::zeros({n, 3}, f32), gcell = at::zeros({bg->n_systems, 3, 3}, f32);
        if (e > 0) {
            for (int k : {2, 3, 6})
                TORCH_CHECK(!go[k].defined() || !go[k].requires_grad(), "pet_hip: double backward through preprocess is not built");
            at::Tensor g_ev = go[2].defined() ? as_f32(go[2]) : at::zeros({n, bg->m, 3}, f32);
            at::Tensor g_ed = go[3].defined() ? as_f32(go[3]) : at::zeros({n, bg->m}, f32);
            at::Tensor g_cf = go[6].defined() ? as_f32(go[6]) : at::zeros({n, bg->m}, f32);
            at::Tensor geo = at::cat({to_csr(g_ev, bg->ix), to_csr(g_ed, bg->ix).unsqueeze(1)}, 1).contiguous();
            at::Tensor gfc = to_csr(g_cf, bg->ix);
            at::Tensor scratch = at::empty({4 * e}, f32);
            check(pet_geometry_backward(be->model, bg->g, geo.data_ptr<float>(), gfc.data_ptr<float>(), gpos.data_ptr<float>(),
                                        gcell.data_ptr<float>(), scratch.data_ptr<float>(), stream_of(gpos)),
                  "pet_geometry_backward");
        }
        return {gpos.to(pd), gcell.to(cd), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

static c10::intrusive_ptr<BatchGraph> graph_from_batch(const at::Tensor& el_nodes, const at::Tensor& el_nbr, const at::Tensor& ev,
                                                       const at::Tensor& ed, const at::Tensor& mask, const at::Tensor& rni,
                                                       const at::Tensor& cf, bool light) {
    auto bg = c10::make_intrusive<BatchGraph>();
    const int64_t n = mask.size(0), m = mask.size(1);
    bg->n_nodes = n;
    bg->m = m;
    auto as_i64 = [](const at::Tensor& t) { return t.to(at::kLong).contiguous(); };
    at::Tensor mk = mask.to(at::kByte).contiguous(), cfc = as_f32(cf);
    if (light) bg->keep = {mk, cfc};
    else bg->keep = {mk, cfc, as_i64(el_nodes), as_i64(el_nbr), as_f32(ev), as_f32(ed), as_i64(rni)};
    auto bytes = at::TensorOptions().dtype(at::kByte).device(mask.device());
    bg->ws = at::empty({pet_graph_from_batch_workspace_bytes(n, m)}, bytes);
    check(pet_graph_from_batch(light ? nullptr : bg->keep[2].data_ptr<int64_t>(), light ? nullptr : bg->keep[3].data_ptr<int64_t>(),
                               light ? nullptr : bg->keep[4].data_ptr<float>(), light ? nullptr : bg->keep[5].data_ptr<float>(),
                               mk.data_ptr<uint8_t>(), light ? nullptr : bg->keep[6].data_ptr<int64_t>(), cfc.data_ptr<float>(),
                               n, m, bg->ws.data_ptr(), bg->ws.numel(), &bg->g, stream_of(mask)),
          "pet_graph_from_batch");
    if (pet_graph_num_edges(bg->g) > 0) bg->ix = csr_index(bg->g, n, mask);
    return bg;
}

// ---- calculate_features: (edge_vectors, edge_distances, cutoff_factors) -> node features [N, d_node], edge features [N, M, d_pet]
struct FeaturesFn : torch::autograd::Function<FeaturesFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& ev,
                                                  const at::Tensor& ed, const at::Tensor& cf,
                                                  c10::intrusive_ptr<PetHipBackend> be, const at::Tensor& el_nodes,
                                                  const at::Tensor& el_nbr, const at::Tensor& mask, const at::Tensor& rni,
                                                  const at::Tensor& charge, const at::Tensor& spin,
                                                  const at::Tensor& system_indices) {
        auto bg = graph_from_batch(el_nodes, el_nbr, ev, ed, mask, rni, cf, false);
        if (be->hypers_struct().system_conditioning) {
            // batch_data["charge"], ["spin_multiplicity"], ["system_indices"] (backend.py:375-378)
            TORCH_CHECK(charge.numel() > 0 && spin.numel() == charge.numel() && system_indices.numel() == mask.size(0),
                        "pet_hip: system_conditioning needs batch_data['charge'], ['spin_multiplicity'] and ['system_indices']");
            auto i64 = [&](const at::Tensor& t) { return t.to(mask.device(), at::kLong).contiguous(); };
            bg->keep.push_back(i64(charge)); bg->keep.push_back(i64(spin)); bg->keep.push_back(i64(system_indices));
            const size_t k = bg->keep.size();
            check(pet_graph_set_conditioning(bg->g, bg->keep[k - 3].data_ptr<int64_t>(), bg->keep[k - 2].data_ptr<int64_t>(),
                                             bg->keep[k - 1].data_ptr<int64_t>(), bg->keep[k - 3].numel()),
                  "pet_graph_set_conditioning");
        }
        const int64_t n = bg->n_nodes, e = pet_graph_num_edges(bg->g);
        const pet_hypers_t h = be->hypers_struct();
        auto dev = mask.device();
        auto bytes = at::TensorOptions().dtype(at::kByte).device(dev), f32 = at::TensorOptions().dtype(at::kFloat).device(dev);
        bg->fwd_ws = at::empty({pet_forward_workspace_bytes_for(be->model, bg->g)}, bytes);
        // backend.py:344-418 returns two lists, one entry per readout layer: here [node_0 .. node_{L-1}, edge_0 .. edge_{L-1}]
        const int64_t L = pet_model_num_readout_layers(be->model);
        std::vector<at::Tensor> nf, ef;
        std::vector<float*> pn, pe;
        for (int64_t l = 0; l < L; l++) {
            nf.push_back(at::empty({n, h.d_node}, f32));
            ef.push_back(at::empty({e, h.d_pet}, f32));
            pn.push_back(nf[l].data_ptr<float>());
            pe.push_back(ef[l].data_ptr<float>());
        }
        check(pet_forward_layers(be->model, bg->g, bg->fwd_ws.data_ptr(), bg->fwd_ws.numel(), 1, pn.data(), pe.data(), (int32_t)L,
                                 stream_of(mask)),
              "pet_forward_layers");
        ctx->saved_data["graph"] = bg;
        ctx->saved_data["backend"] = be;
        ctx->saved_data["generation"] = be->generation;
        ctx->saved_data["dtype"] = (int64_t)ev.scalar_type();
        const auto dt = ev.scalar_type();
        torch::autograd::variable_list out;
        for (int64_t l = 0; l < L; l++) out.push_back(nf[l].to(dt));
        for (int64_t l = 0; l < L; l++)
            out.push_back((e > 0 ? to_nef(ef[l], bg->ix, n, bg->m) : at::zeros({n, bg->m, h.d_pet}, f32)).to(dt));
        return out;
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                   torch::autograd::variable_list go) {
        auto bg = ctx->saved_data["graph"].toCustomClass<BatchGraph>();
        auto be = ctx->saved_data["backend"].toCustomClass<PetHipBackend>();
        be->check_generation(ctx->saved_data["generation"].toInt());
        const auto dt = (at::ScalarType)ctx->saved_data["dtype"].toInt();
        const int64_t n = bg->n_nodes, m = bg->m, e = pet_graph_num_edges(bg->g);
        const pet_hypers_t h = be->hypers_struct();
        auto f32 = at::TensorOptions().dtype(at::kFloat).device(bg->ws.device());
        at::Tensor g_ev = at::zeros({n, m, 3}, f32), g_ed = at::zeros({n, m}, f32), g_cf = at::zeros({n, m}, f32);
        if (e > 0) {
            const int64_t L = (int64_t)go.size() / 2;
            std::vector<at::Tensor> keep;
            std::vector<const float*> pn(L, nullptr), pe(L, nullptr);
            for (int64_t k = 0; k < 2 * L; k++) {
                TORCH_CHECK(!go[k].defined() || !go[k].requires_grad(),
                            "pet_hip: double backward through calculate_features is not built (train through "
                            "metatrain_amd.pet.PETBackend in train() mode or TrainStep)");
                if (!go[k].defined()) continue;
                keep.push_back(k < L ? as_f32(go[k]) : to_csr(as_f32(go[k]), bg->ix));
                (k < L ? pn[k] : pe[k - L]) = keep.back().data_ptr<float>();
            }
            at::Tensor geo = at::empty({e, 4}, f32), gfc = at::empty({e}, f32);
            check(pet_backward_features_layers(be->model, bg->g, bg->fwd_ws.data_ptr(), bg->fwd_ws.numel(), pn.data(), pe.data(),
                                               (int32_t)L, geo.data_ptr<float>(), gfc.data_ptr<float>(), stream_of(geo)),
                  "pet_backward_features_layers");
            g_ev = to_nef(geo.narrow(1, 0, 3).contiguous(), bg->ix, n, m);
            g_ed = to_nef(geo.select(1, 3).contiguous(), bg->ix, n, m);
            g_cf = to_nef(gfc, bg->ix, n, m);
        }
        return {g_ev.to(dt), g_ed.to(dt), g_cf.to(dt), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// ---- predict: (node features, edge features, cutoff_factors) -> per-atom predictions [N, P] of one block ---------------
struct PredictFn : torch::autograd::Function<PredictFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& nf,
                                                  const at::Tensor& ef_nef, const at::Tensor& cf,
                                                  c10::intrusive_ptr<PetHipBackend> be, const at::Tensor& mask,
                                                  std::string target, int64_t layer, std::string block) {
        auto bg = graph_from_batch(at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), mask, at::Tensor(), cf, true);
        const int64_t n = bg->n_nodes, e = pet_graph_num_edges(bg->g);
        const pet_hypers_t h = be->hypers_struct();
        const int32_t P = pet_model_block_properties(be->model, target.c_str(), (int32_t)layer, block.c_str());
        TORCH_CHECK(P >= 1, "pet_hip: no head / last layer for target '", target, "', readout layer ", layer, ", block '", block, "'");
        auto f32 = at::TensorOptions().dtype(at::kFloat).device(mask.device());
        at::Tensor nfc = as_f32(nf);
        at::Tensor efc = e > 0 ? to_csr(as_f32(ef_nef), bg->ix) : at::zeros({0, h.d_pet}, f32);
        at::Tensor atomic = at::empty({n, (int64_t)P}, f32), hn = at::empty({n, h.d_head}, f32), he = at::empty({e, h.d_head}, f32);
        at::Tensor scratch = at::empty({pet_predict_scratch_floats(n, e)}, f32);
        check(pet_predict(be->model, bg->g, target.c_str(), (int32_t)layer, block.c_str(), nfc.data_ptr<float>(),
                          efc.data_ptr<float>(), nullptr, atomic.data_ptr<float>(), hn.data_ptr<float>(), he.data_ptr<float>(),
                          scratch.data_ptr<float>(), stream_of(mask)),
              "pet_predict");
        ctx->saved_data["graph"] = bg;
        ctx->saved_data["backend"] = be;
        ctx->saved_data["generation"] = be->generation;
        ctx->saved_data["nf"] = nfc;
        ctx->saved_data["ef"] = efc;
        ctx->saved_data["target"] = target;
        ctx->saved_data["block"] = block;
        ctx->saved_data["layer"] = layer;
        ctx->saved_data["P"] = (int64_t)P;
        ctx->saved_data["dtype"] = (int64_t)nf.scalar_type();
        const auto dt = nf.scalar_type();
        at::Tensor he_nef = e > 0 ? to_nef(he, bg->ix, n, bg->m) : at::zeros({n, bg->m, h.d_head}, f32);
        std::vector<at::Tensor> out = {atomic.to(dt), hn.to(dt), he_nef.to(dt)};
        ctx->mark_non_differentiable({out[1], out[2]});
        return out;
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                   torch::autograd::variable_list go) {
        auto bg = ctx->saved_data["graph"].toCustomClass<BatchGraph>();
        auto be = ctx->saved_data["backend"].toCustomClass<PetHipBackend>();
        be->check_generation(ctx->saved_data["generation"].toInt());
        const auto dt = (at::ScalarType)ctx->saved_data["dtype"].toInt();
        const std::string target = ctx->saved_data["target"].toStringRef(), block = ctx->saved_data["block"].toStringRef();
        const int64_t layer = ctx->saved_data["layer"].toInt();
        at::Tensor nf = ctx->saved_data["nf"].toTensor(), ef = ctx->saved_data["ef"].toTensor();
        const int64_t n = bg->n_nodes, m = bg->m, e = pet_graph_num_edges(bg->g);
        const pet_hypers_t h = be->hypers_struct();
        TORCH_CHECK(!go[0].requires_grad(), "pet_hip: double backward through predict is not built");
        auto f32 = at::TensorOptions().dtype(at::kFloat).device(bg->ws.device());
        at::Tensor ga = as_f32(go[0]).reshape({n, ctx->saved_data["P"].toInt()}).contiguous();  // (n may be 0: empty system)
        at::Tensor g_nf = at::empty({n, h.d_node}, f32), g_ef = at::zeros({e, h.d_pet}, f32), g_fc = at::zeros({e}, f32);
        at::Tensor scratch = at::empty({pet_predict_scratch_floats(n, e)}, f32);
        check(pet_predict_backward(be->model, bg->g, target.c_str(), (int32_t)layer, block.c_str(), nf.data_ptr<float>(),
                                   ef.data_ptr<float>(), nullptr, ga.data_ptr<float>(), g_nf.data_ptr<float>(),
                                   g_ef.data_ptr<float>(), g_fc.data_ptr<float>(), scratch.data_ptr<float>(), stream_of(ga)),
              "pet_predict_backward");
        at::Tensor g_ef_nef = e > 0 ? to_nef(g_ef, bg->ix, n, m) : at::zeros({n, m, h.d_pet}, f32);
        at::Tensor g_cf = e > 0 ? to_nef(g_fc, bg->ix, n, m) : at::zeros({n, m}, f32);
        return {g_nf.to(dt), g_ef_nef.to(dt), g_cf.to(dt), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

std::vector<at::Tensor> PetHipBackend::preprocess(std::vector<at::Tensor> params, const at::Tensor& positions,
                                                  const at::Tensor& centers, const at::Tensor& neighbors,
                                                  const at::Tensor& species, const at::Tensor& cells,
                                                  const at::Tensor& cell_shifts, const at::Tensor& system_indices) {
    ensure_model(params, positions);
    return PreprocessFn::apply(positions, cells, c10::intrusive_ptr<PetHipBackend>::reclaim_copy(this), centers, neighbors,
                               species, cell_shifts, system_indices);
}
std::vector<at::Tensor> PetHipBackend::calculate_features(std::vector<at::Tensor> params, const at::Tensor& el_nodes,
                                                          const at::Tensor& el_nbr, const at::Tensor& ev, const at::Tensor& ed,
                                                          const at::Tensor& mask, const at::Tensor& rni, const at::Tensor& cf,
                                                          std::vector<at::Tensor> conditioning) {
    ensure_model(params, ev);
    TORCH_CHECK(conditioning.empty() || conditioning.size() == 3,
                "pet_hip: conditioning = [] or [charge, spin_multiplicity, system_indices]");
    const at::Tensor none = at::empty({0}, mask.options().dtype(at::kLong));  // (autograd wants defined tensors)
    return FeaturesFn::apply(ev, ed, cf, c10::intrusive_ptr<PetHipBackend>::reclaim_copy(this), el_nodes, el_nbr, mask, rni,
                             conditioning.empty() ? none : conditioning[0], conditioning.empty() ? none : conditioning[1],
                             conditioning.empty() ? none : conditioning[2]);
}
std::vector<at::Tensor> PetHipBackend::predict(std::vector<at::Tensor> params, std::string target, int64_t readout_layer,
                                               std::string block, const at::Tensor& nf, const at::Tensor& ef,
                                               const at::Tensor& mask, const at::Tensor& cf) {
    ensure_model(params, nf);
    return PredictFn::apply(nf, ef, cf, c10::intrusive_ptr<PetHipBackend>::reclaim_copy(this), mask, target, readout_layer,
                            block);
}

using BackendState = std::tuple<std::vector<double>, std::vector<int64_t>, std::vector<std::string>>;

using State = std::tuple<std::vector<double>, std::vector<int64_t>, std::vector<std::string>, std::vector<at::Tensor>>;

}  // namespace

TORCH_LIBRARY(pet_hip, m) {
    m.class_<GraphHolder>("GraphHolder");
    m.class_<BatchGraph>("BatchGraph");
    m.class_<PetHipBackend>("PetHipBackend")
        .def(torch::init<std::vector<double>, std::vector<int64_t>, std::vector<std::string>>())
        .def("preprocess", &PetHipBackend::preprocess)
        .def("calculate_features", &PetHipBackend::calculate_features)
        .def("predict", &PetHipBackend::predict)
        .def_pickle(
            [](const c10::intrusive_ptr<PetHipBackend>& self) -> BackendState {
                return BackendState(self->hypers, self->atomic_types, self->keys);
            },
            [](BackendState s) {
                return c10::make_intrusive<PetHipBackend>(std::get<0>(s), std::get<1>(s), std::get<2>(s));
            });
    m.class_<PetHipModule>("PetHipModule")
        .def(torch::init<std::vector<double>, std::vector<int64_t>, std::vector<std::string>, std::vector<at::Tensor>>())
        .def("atomic_energies", &PetHipModule::atomic_energies)
        .def("num_tensors", [](const c10::intrusive_ptr<PetHipModule>& self) { return (int64_t)self->tensors.size(); })
        .def_pickle(
            [](const c10::intrusive_ptr<PetHipModule>& self) -> State {
                std::vector<at::Tensor> cpu;
                for (const auto& t : self->tensors) cpu.push_back(t.detach().cpu());
                return State(self->hypers, self->atomic_types, self->keys, cpu);
            },
            [](State s) {
                return c10::make_intrusive<PetHipModule>(std::get<0>(s), std::get<1>(s), std::get<2>(s), std::get<3>(s));
            });
}
