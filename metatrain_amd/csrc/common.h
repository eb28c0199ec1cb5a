// Shared declarations for the MI355X (gfx950) PET hot-path library.
// One process per GPU; every entry point launches on the caller's HIP stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/pet_hip.h"

namespace pet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

void set_error(const std::string& msg);

#define PET_HIP_CHECK(expr)                                                         \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            char _b[512];                                                           \
            snprintf(_b, sizeof(_b), "%s:%d: %s failed: %s", __FILE__, __LINE__,    \
                     #expr, hipGetErrorString(_e));                                 \
            pet::set_error(_b);                                                     \
            return PET_ERR_HIP;                                                     \
        }                                                                           \
    } while (0)

#define PET_REQUIRE(cond, code, msg)                                                \
    do {                                                                            \
        if (!(cond)) {                                                              \
            pet::set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + \
                           ": " + (msg));                                           \
            return (code);                                                          \
        }                                                                           \
    } while (0)

// bump allocator over a caller-provided device buffer (256-byte aligned carves);
// with base == nullptr it only measures.
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    template <class T>
    T* take(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

inline int cdiv(int64_t a, int64_t b) { return int((a + b - 1) / b); }

// gfx950 has 160 KiB of LDS per CU; kernels asking for more than 64 KiB of dynamic LDS
// must opt in once per function.
template <class Kern>
inline void allow_big_lds(Kern kern, size_t bytes) {
    if (bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}

// --------------------------------------------------------------------------------
// Edge graph in CSR order (edges stably sorted by centre; pet/modules/nef.py:63-70
// defines exactly this order as the NEF slot order).
// --------------------------------------------------------------------------------
constexpr int GRID_MAX_PROBES = 64;  // adaptive_cutoff_method = "grid": probe cutoffs 0.5, 0.5 + w/4, .. < cutoff

struct Graph {
    int64_t n_nodes = 0, n_edges_in = 0, n_systems = 0;
    int64_t n_edges = 0;  // kept edges (host copy, valid after build)
    int max_nbr = 0;      // host copy
    // per input edge
    float4* vin = nullptr;     // [E0] (vx,vy,vz,d0 = |v| + 1e-15)
    int* keep = nullptr;       // [E0] 0/1
    int* kidx = nullptr;       // [E0] exclusive scan of keep: index among kept edges
    int* sort_keys_in = nullptr;
    int* sort_keys_out = nullptr;
    int* sort_vals_in = nullptr;
    int* perm = nullptr;       // [E0] sorted position -> input edge
    // CSR
    int* rowptr = nullptr;     // [N+1]
    int* ctr = nullptr;        // [E]
    int* nbr = nullptr;        // [E]
    int* shift = nullptr;      // [E,3]
    int* rev = nullptr;        // [E] CSR index of (j,i,-S)
    int* sp = nullptr;         // [N] species index of the atom
    int* sp_nbr = nullptr;     // [E] species index of the neighbour
    float4* geo = nullptr;     // [E] (vx,vy,vz,dist = sqrt(v.v + 1e-15))
    float* d0 = nullptr;       // [E] |v| + 1e-15
    float* fc = nullptr;       // [E] cutoff factor
    int* sys = nullptr;        // [N] system index
    int* scalars = nullptr;    // [128] device ([24 + t] atoms of t <= 32 tokens, [64 + t] fill cursors of that sort): n_kept, max_nbr, n_bad_reverse, pad source, input checks; [8..12] atoms per
                               // attention tile count (1, 2, 3, 4, more), [13..17] fill cursors
    // atoms listed by attention tile count nt = ceil((neighbours + 1) / 16): the attention kernels are launched per
    // tile count over exactly their own atoms (pet_attn.hip); bucket_start[k] = first entry of tile count k + 1
    int* atom_order = nullptr; // [N]
    // attention tiles of the per-atom fused block (pet_ablk.hip): a tile is 32 token slots holding ONE atom of at most 32
    // tokens or TWO whose tokens add up to at most 32 (the plan pairs small atoms with the largest partners that fit);
    // atoms of 33 .. 64 tokens get a 64-slot tile each. Two int4 per tile: (atom A, first CSR row A, tokens A, atom B),
    // (first CSR row B, tokens B or 0, -, -).
    int* atoms_by_t = nullptr; // [N] atoms of at most 32 tokens, grouped by token count, ascending atom index inside a group
    int* tsort_tmp = nullptr;  // [ceil(N / 256)][33] per-block counts / first positions of that sort
    int4* tile_desc = nullptr; // [2 N]: the 32-slot tiles first (n_tiles1), then the 64-slot ones (n_tiles2)
    int n_tiles1 = 0, n_tiles2 = 0;  // host copies
    // What this graph's last forward into a given workspace left there (pet_fwd.hip note_workspace): whether it ran the
    // size-generic path, and whether its attention layers ran the fused per-atom block WITHOUT saving Q, K, V (the adjoint
    // must then be the fused one). One record per workspace (the last four), so that forwards of the same graph into
    // different workspaces do not overwrite each other's record.
    struct FwdRecord {
        const void* ws = nullptr;
        bool generic = false;
        bool attn_unsaved = false;  // some attention layer ran the fused block: its QKV / AO buffers were not written
        bool emlp_unsaved = false;  // the edge MLPs did not write [v; g]: the adjoint recomputes them (pet_emlp_s.hip)
        int save = -1;              // the forward's save level (0 = nothing kept for an adjoint: pet_backward must refuse)
    };
    mutable FwdRecord fwd_rec[4];
    mutable int fwd_rec_next = 0;
    const FwdRecord* fwd_record(const void* ws) const {
        for (const FwdRecord& r : fwd_rec)
            if (r.ws == ws && ws) return &r;
        return nullptr;
    }
    FwdRecord& fwd_record_new(const void* ws) const {
        for (FwdRecord& r : fwd_rec)
            if (r.ws == ws) return r = FwdRecord{ws, false, false, false, -1};
        FwdRecord& r = fwd_rec[fwd_rec_next];
        fwd_rec_next = (fwd_rec_next + 1) % 4;
        return r = FwdRecord{ws, false, false, false, -1};
    }
    bool attn_lists = false;         // atom_order / bucket_start (and the tile plan) exist (graph.hip graph_attention_lists)
    bool tiles_planned = false;      // the graph build made tile_desc (large graphs, or the fused block forced)
    int bucket_start[6] = {0, 0, 0, 0, 0, 0};  // host copy
    // adaptive cutoff (structures.py:225-263): CSR over ALL input edges (the root finder and its
    // implicit-function gradient see every edge within the maximum cutoff, kept or not)
    bool adaptive = false;
    int* rowptr0 = nullptr;    // [N+1]
    int* perm0 = nullptr;      // [E0] sorted-by-centre position -> input edge
    int* nbr0 = nullptr;       // [E0]
    int* shift0 = nullptr;     // [E0,3]
    int* rev0 = nullptr;       // [E0] position of the reverse edge in the all-edge CSR
    float* r_atom = nullptr;   // [N] adapted cutoff (after the IFT step and the clamp)
    float* r_newton = nullptr; // [N] root of the Newton-bisection loop
    float* inv_dn = nullptr;   // [N] 1 / max(dn_total/dr, 1e-6) at the root; 0 if the clamp is active
    float* grid_drdn = nullptr; // [N, GRID_MAX_PROBES] "grid" method: d(atomic cutoff) / d(smoothed count at probe k)
    int grid_probes = 0;        // number of probe cutoffs of the "grid" method (0: "solver")
    // system conditioning (pet_graph_set_conditioning): caller-owned device arrays
    const int64_t* cond_charge = nullptr;  // [n_cond_systems]
    const int64_t* cond_spin = nullptr;    // [n_cond_systems]
    const int64_t* cond_sys = nullptr;     // [N] or nullptr: use `sys`
    int64_t n_cond_systems = 0;
    // per-layer exchange of edge tokens for ONE box over several ranks (pet_graph_set_exchange; pet/partition.py): rows whose
    // centre another rank owns ("ghost") receive their transformer output from that rank before the combination stage, rows
    // with an owned centre and a foreign neighbour ("export") are sent; the adjoints travel the other way
    const int* x_export = nullptr;   // [n_export] CSR rows, grouped by destination rank
    const int* x_ghost = nullptr;    // [n_ghost] CSR rows, grouped by source rank
    int64_t n_export = 0, n_ghost = 0;
    float* x_export_buf = nullptr;   // [n_export, d_pet] caller-owned staging (what the collective reads / writes)
    float* x_ghost_buf = nullptr;    // [n_ghost, d_pet]
    int (*x_fn)(void* user, int direction, int layer) = nullptr;  // the collective: 0 = export -> ghost, 1 = ghost -> export
    void* x_user = nullptr;
    float* pc = nullptr;       // [E] pair cutoff of every kept edge
    float* ad_gc = nullptr;    // [E] scratch: dL/d(pair cutoff)
    float* ad_gr = nullptr;    // [N] scratch: dL/d(atomic cutoff)
    float4* ad_dv = nullptr;   // [E0] scratch: adaptive part of dL/d(edge vector), all-edge CSR order
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    void* scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
};

// A temporary from the stream's memory pool, returned on EVERY exit of the scope (PET_REQUIRE / PET_HIP_CHECK return early).
struct PoolBuf {
    void* p = nullptr;
    hipStream_t st = nullptr;
    PoolBuf() = default;
    PoolBuf(const PoolBuf&) = delete;
    PoolBuf& operator=(const PoolBuf&) = delete;
    hipError_t alloc(size_t bytes, hipStream_t s) {
        st = s;
        return hipMallocAsync(&p, bytes, s);
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
    ~PoolBuf() { if (p) (void)hipFreeAsync(p, st); }
};

}  // namespace pet
