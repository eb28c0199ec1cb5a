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
        TORCH_CHECK(hypers.size() == 16, "pet_hip: expected the 16 fields of pet_hypers_t");
        TORCH_CHECK(keys.size() == tensors.size(), "pet_hip: keys / tensors length mismatch");
    }
    ~PetHipModule() override {
        if (model) pet_model_destroy(model);
    }

    pet_hypers_t hypers_struct() const {
        pet_hypers_t h;
        h.cutoff = (float)hypers[0]; h.cutoff_width = (float)hypers[1]; h.cutoff_function = (int32_t)hypers[2];
        h.d_pet = (int32_t)hypers[3]; h.d_head = (int32_t)hypers[4]; h.d_node = (int32_t)hypers[5];
        h.d_feedforward = (int32_t)hypers[6]; h.num_heads = (int32_t)hypers[7];
        h.num_attention_layers = (int32_t)hypers[8]; h.num_gnn_layers = (int32_t)hypers[9];
        h.attention_temperature = (float)hypers[10]; h.nl_is_strict = (int32_t)hypers[11];
        h.n_species = (int32_t)hypers[12]; h.max_atomic_number = (int32_t)hypers[13];
        h.num_neighbors_adaptive = (float)hypers[14]; h.cutoff_width_adaptive = (float)hypers[15];
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
        gh->fwd_ws = at::empty({pet_forward_workspace_bytes(mod->model, gh->n_nodes, pet_graph_num_edges(gh->g))}, bytes);
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

using State = std::tuple<std::vector<double>, std::vector<int64_t>, std::vector<std::string>, std::vector<at::Tensor>>;

}  // namespace

TORCH_LIBRARY(pet_hip, m) {
    m.class_<GraphHolder>("GraphHolder");
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
