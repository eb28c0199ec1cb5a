// Periodic cell-list neighbour list on the GPU (replaces the vesin call at
// src/metatrain/utils/neighbor_lists.py:131-135; contract in oracle/nl.py).
//
// Atoms are wrapped into the cell and binned along the lattice directions with bin
// width >= cutoff where the cell allows it; a thin cell gets a single bin and the
// search walks over ceil(cutoff / height) periodic images instead. Bins are filled by
// a stable radix sort (atom order inside a bin = atom index), so the output order is
// deterministic. Two passes (count, exclusive scan, fill) of a wave-per-bin kernel emit
// pairs grouped by centre, i.e. already in the CSR order pet_graph_build wants; all
// systems of a batch go through the same launches (pet_nl_build_batch).
#include "common.h"
#include "model.h"

#include <algorithm>
#include <math.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <string>
#include <vector>

namespace pet {

struct NlParams {
    float cell[9];     // effective lattice (rows), unit vectors on non-periodic axes
    float inv[9];      // inverse (columns map cartesian -> fractional)
    float origin[3];   // fractional origin for non-periodic axes
    float extent[3];   // fractional extent for non-periodic axes (bins span [origin, origin+extent))
    int nb[3];         // bins per axis
    int reach[3];      // bins to search on each side
    int pbc[3];
    float cutoff2;
};

// order-preserving float <-> int map so that integer atomicMin/Max implement float min/max
__host__ __device__ inline int ord_i(float f) {
    int i;
    memcpy(&i, &f, 4);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ inline float unord_f(int i) {
    i = i >= 0 ? i : i ^ 0x7fffffff;
    float f;
    memcpy(&f, &i, 4);
    return f;
}

__device__ __forceinline__ void frac_of(const float* pos, int i, const NlParams& prm, float fr[3]) {
    float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
    for (int a = 0; a < 3; a++) fr[a] = x * prm.inv[a] + y * prm.inv[3 + a] + z * prm.inv[6 + a];
}


// ---------------------------------------------------------------------------------------------------------------
// Batched build: every system of a batch in ONE set of launches, one wave per occupied bin.
//   k_nlb_bin      thread per atom: its system (binary search in first_atom), wrap into the cell, bin key
//                  (system's bin base + bin)
//   radix sort     atoms by key -> bins are contiguous runs of the sorted array, systems are contiguous blocks of bins
//   k_nlb_post_sort  bin starts, the bins' systems, and the sorted SoA tile source: spos[q] = (wrapped x, y, z, atom id) so that
//                  a wave reads a bin's atoms as ONE coalesced 16 B-per-lane load
//   k_nlb_pairs    one WAVE per bin. It stages the atoms of the (2 reach + 1)^3 surrounding bins -- periodic images
//                  included -- as one dense candidate tile in wave-private LDS (position, atom id, image number), then
//                  for every centre of its bin all 64 lanes test 64 candidates at a time; hits are compacted with a
//                  ballot / prefix-popcount and written at the centre's running offset (PASS 1) or counted (PASS 0).
//                  No thread ever walks a neighbour list serially, and lane utilisation is that of the dense tile
//                  (~125 candidates for 4.6 centres at rho = 0.05 / A^3) instead of one lane per atom.
// Pairs come out grouped by centre in atom order (CSR order), neighbours in (bin, sorted-atom) order: deterministic.
// The distance is evaluated as (w_j - w_i) + offset(S) exactly like the per-atom kernel above did, which makes it
// bitwise antisymmetric under (i, j, S) <-> (j, i, -S): the list is a full list even at the cutoff's last ulp.
// ---------------------------------------------------------------------------------------------------------------
constexpr int NL_TILE = 512;        // candidates staged per wave and round
constexpr int NL_MAX_IMAGES = 125;  // (2 reach + 1)^3 <= 125: reach <= 2 per axis on the tile path (thin cells: fallback rounds)

__global__ void k_nlb_bin(const float* __restrict__ pos, int n, const NlParams* __restrict__ prms,
                          const int* __restrict__ first_atom, const int* __restrict__ bin_base, int n_sys,
                          int* __restrict__ bin_key, int* __restrict__ atom_id, int* __restrict__ wrap,
                          float* __restrict__ wpos, int* __restrict__ sys_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = n_sys;  // last s with first_atom[s] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first_atom[mid] <= i) lo = mid; else hi = mid;
    }
    const NlParams& prm = prms[lo];
    float fr[3];
    frac_of(pos, i, prm, fr);
    int b[3], wr[3];
    for (int a = 0; a < 3; a++) {
        if (prm.pbc[a]) {
            const float fl = floorf(fr[a]);
            wr[a] = (int)fl;
            b[a] = min((int)((fr[a] - fl) * prm.nb[a]), prm.nb[a] - 1);
        } else {
            wr[a] = 0;
            b[a] = max(0, min((int)((fr[a] - prm.origin[a]) / prm.extent[a] * prm.nb[a]), prm.nb[a] - 1));
        }
    }
    bin_key[i] = bin_base[lo] + (b[0] * prm.nb[1] + b[1]) * prm.nb[2] + b[2];
    atom_id[i] = i;
    sys_of[i] = lo;
    wrap[3 * i] = wr[0]; wrap[3 * i + 1] = wr[1]; wrap[3 * i + 2] = wr[2];
    for (int k = 0; k < 3; k++)
        wpos[3 * i + k] = pos[3 * i + k] - (wr[0] * prm.cell[k] + wr[1] * prm.cell[3 + k] + wr[2] * prm.cell[6 + k]);
}


// bounding boxes of the open systems (fractional coordinates along the completed lattice)
__global__ void k_nlb_bbox(const float* __restrict__ pos, int n, const NlParams* __restrict__ prms,
                           const int* __restrict__ first_atom, int n_sys, int* __restrict__ bbox /*[S][6]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = n_sys;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first_atom[mid] <= i) lo = mid; else hi = mid;
    }
    float fr[3];
    frac_of(pos, i, prms[lo], fr);
    for (int a = 0; a < 3; a++) {
        atomicMin(bbox + 6 * lo + a, ord_i(fr[a]));
        atomicMax(bbox + 6 * lo + 3 + a, ord_i(fr[a]));
    }
}

template <int PASS>
__global__ __launch_bounds__(256) void k_nlb_pairs(const float* __restrict__ pos, const float4* __restrict__ spos,
                                                    const int* __restrict__ wrap, const int* __restrict__ bin_start,
                                                    const NlParams* __restrict__ prms, const int* __restrict__ bin_base,
                                                    const int* __restrict__ bin_sys, int total_bins,
                                                    int* __restrict__ counts, const int* __restrict__ offsets,
                                                    int* __restrict__ pairs, float* __restrict__ vectors) {
    __shared__ float4 t_pos[4][NL_TILE];   // candidate: wrapped position, atom id in .w
    __shared__ uint8_t t_img[4][NL_TILE];  // candidate: index of its periodic image in t_off / t_sh
    __shared__ float4 t_off[4][NL_MAX_IMAGES + 3];
    __shared__ int t_sh[4][NL_MAX_IMAGES + 3];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bin = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wv);
    if (bin >= total_bins) return;
    const int c0 = bin_start[bin], nc = bin_start[bin + 1] - c0;
    if (nc == 0) return;
    const int s = bin_sys[bin];
    const NlParams& prm = prms[s];
    const int lb = bin - bin_base[s];
    const int b0 = lb / (prm.nb[1] * prm.nb[2]), b1 = (lb / prm.nb[2]) % prm.nb[1], b2 = lb % prm.nb[2];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    float4* tp = t_pos[wv];
    uint8_t* ti = t_img[wv];
    // centres of this bin, 64 at a time: lane k holds centre k
    for (int cb = 0; cb < nc; cb += 64) {
        const int ncc = min(64, nc - cb);
        float4 ctr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < ncc) ctr = spos[c0 + cb + lane];
        const int ci_atom = __float_as_int(ctr.w);
        int run = (PASS == 1 && lane < ncc) ? offsets[ci_atom] : 0;  // PASS 0: the count; PASS 1: the next output row
        // the surrounding bins in rounds of at most NL_TILE candidates and NL_MAX_IMAGES distinct images
        int n_tile = 0, n_img = 0;
        auto flush = [&]() {
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < ncc; k++) {  // wave-uniform loop over the centres
                const float xi = __shfl(ctr.x, k), yi = __shfl(ctr.y, k), zi = __shfl(ctr.z, k);
                const int i = __shfl(ci_atom, k);
                int base = __shfl(run, k);
                for (int t0 = 0; t0 < n_tile; t0 += 64) {
                    const int t = t0 + lane;
                    bool hit = false;
                    float4 cj = make_float4(0.f, 0.f, 0.f, 0.f);
                    int img = 0;
                    if (t < n_tile) {
                        cj = tp[t];
                        img = ti[t];
                        const float4 o = t_off[wv][img];
                        const float dx = (cj.x - xi) + o.x, dy = (cj.y - yi) + o.y, dz = (cj.z - zi) + o.z;
                        const int j = __float_as_int(cj.w);
                        hit = (dx * dx + dy * dy + dz * dz < prm.cutoff2) && !(j == i && t_sh[wv][img] == 0x00808080);
                    }
                    const unsigned long long m = __ballot(hit);
                    if (PASS == 1 && hit) {
                        const int64_t out = base + __popcll(m & lt_mask);
                        const int j = __float_as_int(cj.w);
                        const int code = t_sh[wv][img];
                        const int Sa = ((code >> 16) & 0xff) - 128 + wrap[3 * i] - wrap[3 * j];
                        const int Sb = ((code >> 8) & 0xff) - 128 + wrap[3 * i + 1] - wrap[3 * j + 1];
                        const int Sc = (code & 0xff) - 128 + wrap[3 * i + 2] - wrap[3 * j + 2];
                        int* row = pairs + 5 * out;
                        row[0] = i; row[1] = j; row[2] = Sa; row[3] = Sb; row[4] = Sc;
                        if (vectors) {  // D = r_j - r_i + S.cell evaluated like structures.py:212-220
                            vectors[3 * out] = (pos[3 * j] - pos[3 * i]) + (Sa * prm.cell[0] + Sb * prm.cell[3] + Sc * prm.cell[6]);
                            vectors[3 * out + 1] = (pos[3 * j + 1] - pos[3 * i + 1]) + (Sa * prm.cell[1] + Sb * prm.cell[4] + Sc * prm.cell[7]);
                            vectors[3 * out + 2] = (pos[3 * j + 2] - pos[3 * i + 2]) + (Sa * prm.cell[2] + Sb * prm.cell[5] + Sc * prm.cell[8]);
                        }
                    }
                    base += __popcll(m);
                }
                if (lane == k) run = base;
            }
            __builtin_amdgcn_wave_barrier();
            n_tile = 0;
            n_img = 0;
        };
        for (int da = -prm.reach[0]; da <= prm.reach[0]; da++) {
            int ba = b0 + da, sa = 0;
            if (prm.pbc[0]) { sa = (ba >= 0) ? ba / prm.nb[0] : -((-ba + prm.nb[0] - 1) / prm.nb[0]); ba -= sa * prm.nb[0]; }
            else if (ba < 0 || ba >= prm.nb[0]) continue;
            for (int db = -prm.reach[1]; db <= prm.reach[1]; db++) {
                int bb = b1 + db, sb = 0;
                if (prm.pbc[1]) { sb = (bb >= 0) ? bb / prm.nb[1] : -((-bb + prm.nb[1] - 1) / prm.nb[1]); bb -= sb * prm.nb[1]; }
                else if (bb < 0 || bb >= prm.nb[1]) continue;
                for (int dc = -prm.reach[2]; dc <= prm.reach[2]; dc++) {
                    int bc = b2 + dc, sc = 0;
                    if (prm.pbc[2]) { sc = (bc >= 0) ? bc / prm.nb[2] : -((-bc + prm.nb[2] - 1) / prm.nb[2]); bc -= sc * prm.nb[2]; }
                    else if (bc < 0 || bc >= prm.nb[2]) continue;
                    const int nbin = bin_base[s] + (ba * prm.nb[1] + bb) * prm.nb[2] + bc;
                    const int q0 = bin_start[nbin];
                    int nq = bin_start[nbin + 1] - q0;
                    if (nq == 0) continue;
                    if (n_img == NL_MAX_IMAGES) flush();
                    const int img = n_img++;
                    if (lane == 0) {  // image offset, evaluated exactly as the per-atom kernel did (antisymmetric in S)
                        t_off[wv][img] = make_float4(sa * prm.cell[0] + sb * prm.cell[3] + sc * prm.cell[6],
                                                     sa * prm.cell[1] + sb * prm.cell[4] + sc * prm.cell[7],
                                                     sa * prm.cell[2] + sb * prm.cell[5] + sc * prm.cell[8], 0.f);
                        t_sh[wv][img] = ((sa + 128) << 16) | ((sb + 128) << 8) | (sc + 128);
                    }
                    for (int q = 0; q < nq; q += 64) {  // one coalesced 16 B-per-lane load per 64 atoms of the bin
                        const int m = min(64, nq - q);
                        if (n_tile + m > NL_TILE) {
                            flush();
                            // the image slot was reset with the tile: register it again
                            if (lane == 0) {
                                t_off[wv][0] = make_float4(sa * prm.cell[0] + sb * prm.cell[3] + sc * prm.cell[6],
                                                           sa * prm.cell[1] + sb * prm.cell[4] + sc * prm.cell[7],
                                                           sa * prm.cell[2] + sb * prm.cell[5] + sc * prm.cell[8], 0.f);
                                t_sh[wv][0] = ((sa + 128) << 16) | ((sb + 128) << 8) | (sc + 128);
                            }
                            n_img = 1;
                        }
                        if (lane < m) {
                            tp[n_tile + lane] = spos[q0 + q + lane];
                            ti[n_tile + lane] = (uint8_t)(n_img - 1);
                        }
                        n_tile += m;
                    }
                }
            }
        }
        if (n_tile > 0) flush();
        if (PASS == 0 && lane < ncc) counts[ci_atom] = run;
    }
}

struct NlWs {
    int *bin_key, *bin_key_sorted, *atom_id, *sorted_atoms, *wrap, *bin_start, *counts, *offsets, *sys_of, *bin_sys;
    int *first_atom, *bin_base, *bbox;
    float* wpos;
    float4* spos;
    NlParams* prms;
    void* tmp;
    size_t tmp_bytes;
    size_t total;
};

static int carve_nl(NlWs& w, void* base, int64_t n, int64_t n_sys) {
    Carver c(base);
    const int64_t na = n > 0 ? n : 1, ns = n_sys > 0 ? n_sys : 1;
    w.bin_key = c.take<int>(na);
    w.bin_key_sorted = c.take<int>(na);
    w.atom_id = c.take<int>(na);
    w.sorted_atoms = c.take<int>(na);
    w.wrap = c.take<int>(3 * na);
    w.bin_start = c.take<int>(na + 64 * ns + 2);   // at most n + 32 bins per system
    w.bin_sys = c.take<int>(na + 64 * ns + 2);
    w.counts = c.take<int>(na + 1);
    w.offsets = c.take<int>(na + 1);
    w.sys_of = c.take<int>(na);
    w.wpos = c.take<float>(3 * na);
    w.spos = c.take<float4>(na);
    w.first_atom = c.take<int>(ns + 1);
    w.bin_base = c.take<int>(ns + 1);
    w.bbox = c.take<int>(6 * ns);
    w.prms = c.take<NlParams>(ns);
    size_t s1 = 0, s2 = 0;
    int* ni = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, s1, ni, ni, ni, ni, (size_t)na, 0, 32) != hipSuccess) return PET_ERR_HIP;
    if (rocprim::exclusive_scan(nullptr, s2, ni, ni, 0, (size_t)na + 1, rocprim::plus<int>()) != hipSuccess)
        return PET_ERR_HIP;
    w.tmp_bytes = s1 > s2 ? s1 : s2;
    w.tmp = c.take<char>(w.tmp_bytes + 256);
    w.total = c.off;
    return PET_OK;
}

int64_t nl_batch_workspace_bytes(int64_t n_atoms, int64_t n_systems) {
    NlWs w;
    if (carve_nl(w, nullptr, n_atoms, n_systems) != PET_OK) return -1;
    return (int64_t)w.total;
}
int64_t nl_workspace_bytes(int64_t n_atoms) { return nl_batch_workspace_bytes(n_atoms, 1); }

static void cross(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static double norm3(const double* a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// lattice part of the parameters of one system (effective cell, inverse, heights); false on a singular cell
static int lattice_params(const float* h_cell, const int* h_pbc, float cutoff, NlParams& prm, double height[3]) {
    double c[3][3];
    for (int a = 0; a < 3; a++)
        for (int k = 0; k < 3; k++) c[a][k] = h_pbc[a] ? (double)h_cell[3 * a + k] : 0.0;
    {   // Gram-Schmidt completion: unit vectors orthogonal to everything already present
        double basis[3][3];
        int nbasis = 0;
        auto add_basis = [&](const double* v) {
            double r[3] = {v[0], v[1], v[2]};
            for (int q = 0; q < nbasis; q++) {
                double d = r[0] * basis[q][0] + r[1] * basis[q][1] + r[2] * basis[q][2];
                for (int k = 0; k < 3; k++) r[k] -= d * basis[q][k];
            }
            double nn = norm3(r);
            if (nn > 1e-9) {
                for (int k = 0; k < 3; k++) basis[nbasis][k] = r[k] / nn;
                nbasis++;
            }
            return nn;
        };
        for (int a = 0; a < 3; a++)
            if (h_pbc[a]) add_basis(c[a]);
        for (int a = 0; a < 3; a++) {
            if (h_pbc[a]) continue;
            double best[3] = {0, 0, 0}, bestn = 0;
            for (int e = 0; e < 3; e++) {
                double r[3] = {e == 0 ? 1.0 : 0.0, e == 1 ? 1.0 : 0.0, e == 2 ? 1.0 : 0.0};
                for (int q = 0; q < nbasis; q++) {
                    double d = r[0] * basis[q][0] + r[1] * basis[q][1] + r[2] * basis[q][2];
                    for (int k = 0; k < 3; k++) r[k] -= d * basis[q][k];
                }
                double nn = norm3(r);
                if (nn > bestn + 1e-9) { bestn = nn; for (int k = 0; k < 3; k++) best[k] = r[k] / nn; }
            }
            PET_REQUIRE(bestn > 1e-6, PET_ERR_ARGUMENT, "cannot complete the lattice for non-periodic axes");
            for (int k = 0; k < 3; k++) c[a][k] = best[k];
            add_basis(c[a]);
        }
    }
    double cr[3];
    cross(c[1], c[2], cr);
    const double det = c[0][0] * cr[0] + c[0][1] * cr[1] + c[0][2] * cr[2];
    PET_REQUIRE(fabs(det) > 1e-12, PET_ERR_ARGUMENT, "singular cell");
    double inv[3][3];  // inv[k][a]: cartesian k -> fractional a
    {
        double r0[3], r1[3], r2[3];
        cross(c[1], c[2], r0);
        cross(c[2], c[0], r1);
        cross(c[0], c[1], r2);
        for (int k = 0; k < 3; k++) { inv[k][0] = r0[k] / det; inv[k][1] = r1[k] / det; inv[k][2] = r2[k] / det; }
    }
    for (int a = 0; a < 3; a++)
        for (int k = 0; k < 3; k++) { prm.cell[3 * a + k] = (float)c[a][k]; prm.inv[3 * k + a] = (float)inv[k][a]; }
    for (int a = 0; a < 3; a++) prm.pbc[a] = h_pbc[a] ? 1 : 0;
    prm.cutoff2 = cutoff * cutoff;
    for (int a = 0; a < 3; a++) {
        double x[3];
        cross(c[(a + 1) % 3], c[(a + 2) % 3], x);
        height[a] = fabs(det) / norm3(x);
        prm.origin[a] = 0.f;
        prm.extent[a] = 1.f;
    }
    return PET_OK;
}

// bins of one system from its heights (and, for open axes, its bounding box); returns the number of bins
static int64_t bin_params(NlParams& prm, const double height[3], const float* h_bbox, float cutoff, int64_t n_sys_atoms) {
    int64_t total_bins = 1;
    for (int a = 0; a < 3; a++) {
        double span = height[a];  // length covered by the bins along this axis
        if (!prm.pbc[a]) {
            // fractional coordinate of a unit vector axis is a length already
            const float lo = h_bbox[a], hi = h_bbox[3 + a];
            const float ext = fmaxf(hi - lo, 1e-3f) * 1.0001f + 1e-4f;
            prm.origin[a] = lo - 0.5e-4f;
            prm.extent[a] = ext;
            span = ext * height[a];
        }
        int nb = (int)floor(span / cutoff);
        nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
        prm.nb[a] = nb;
        total_bins *= nb;
    }
    while (total_bins > n_sys_atoms + 32) {  // keep the bin table inside the workspace carve
        int a = 0;
        for (int k = 1; k < 3; k++) if (prm.nb[k] > prm.nb[a]) a = k;
        if (prm.nb[a] <= 1) break;
        total_bins = total_bins / prm.nb[a];
        prm.nb[a] = (prm.nb[a] + 1) / 2;
        total_bins *= prm.nb[a];
    }
    for (int a = 0; a < 3; a++) {
        const double span = prm.pbc[a] ? height[a] : prm.extent[a] * height[a];
        const double width = span / prm.nb[a];
        prm.reach[a] = (int)ceil(cutoff / width);
        if (!prm.pbc[a] && prm.reach[a] > prm.nb[a]) prm.reach[a] = prm.nb[a];
    }
    return total_bins;
}


// What follows the sort of the bin keys, in one launch (three tiny launches and a memset before: a small box's neighbour list is
// launch-bound): thread t < nbins + 1: first sorted position of bin t; t < nbins: the bin's system;
// t < n: the wrapped position of the t-th atom in bin order (spos, below); thread 0: the pair counts' terminator.
__global__ void k_nlb_post_sort(const int* __restrict__ sorted_keys, const int* __restrict__ sorted_atoms,
                                const float* __restrict__ wpos, const int* __restrict__ bin_base, int n, int n_sys, int nbins,
                                int* __restrict__ bin_start, int* __restrict__ bin_sys, float4* __restrict__ spos,
                                int* __restrict__ counts_end) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) *counts_end = 0;
    if (t <= nbins) {
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sorted_keys[mid] < t) lo = mid + 1; else hi = mid;
        }
        bin_start[t] = lo;
    }
    if (t < nbins) {
        int lo = 0, hi = n_sys;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (bin_base[mid] <= t) lo = mid; else hi = mid;
        }
        bin_sys[t] = lo;
    }
    if (t < n) {
        const int j = sorted_atoms[t];
        spos[t] = make_float4(wpos[3 * j], wpos[3 * j + 1], wpos[3 * j + 2], __int_as_float(j));
    }
}

int nl_build_batch(const float* d_pos, const float* h_cells, const int* h_pbc, const int64_t* h_first_atom, int64_t n_sys,
                   float cutoff, void* ws, int* d_pairs, float* d_vectors, int64_t capacity, int64_t* n_pairs,
                   hipStream_t st) {
    *n_pairs = 0;
    PET_REQUIRE(n_sys >= 1 && h_first_atom && h_cells && h_pbc, PET_ERR_ARGUMENT, "bad argument");
    const int64_t n = h_first_atom[n_sys];
    if (n == 0) return PET_OK;
    PET_REQUIRE(cutoff > 0, PET_ERR_ARGUMENT, "cutoff must be positive");
    PET_REQUIRE(n < (int64_t)1 << 30, PET_ERR_UNSUPPORTED, "more than 2^30 atoms in one neighbour-list batch");
    NlWs w;
    int rc = carve_nl(w, ws, n, n_sys);
    if (rc) return rc;
    std::vector<NlParams> prm(n_sys);
    std::vector<double> height(3 * n_sys);
    std::vector<int> first(n_sys + 1), base(n_sys + 1);
    bool any_open = false;
    for (int64_t s = 0; s < n_sys; s++) {
        first[s] = (int)h_first_atom[s];
        PET_REQUIRE(h_first_atom[s + 1] >= h_first_atom[s], PET_ERR_ARGUMENT, "first_atom must be non-decreasing");
        if ((rc = lattice_params(h_cells + 9 * s, h_pbc + 3 * s, cutoff, prm[s], &height[3 * s]))) return rc;
        any_open = any_open || !(h_pbc[3 * s] && h_pbc[3 * s + 1] && h_pbc[3 * s + 2]);
    }
    first[n_sys] = (int)n;
    const int T = 256;
    // first_atom, bin_base, (bbox), prms sit one behind the other in the workspace (carve_nl): when no system is open nothing on
    // the device is needed to fill them, and they travel in ONE copy below (three before: each a launch on a small box's critical path)
    if (any_open)
        PET_HIP_CHECK(hipMemcpyAsync(w.first_atom, first.data(), (n_sys + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    std::vector<float> h_bbox(6 * n_sys);
    for (int64_t s = 0; s < n_sys; s++)
        for (int k = 0; k < 6; k++) h_bbox[6 * s + k] = k < 3 ? 0.f : 1.f;
    if (any_open) {  // one read-back for all open systems of the batch
        std::vector<int> init(6 * n_sys);
        for (int64_t s = 0; s < n_sys; s++)
            for (int k = 0; k < 3; k++) { init[6 * s + k] = ord_i(INFINITY); init[6 * s + 3 + k] = ord_i(-INFINITY); }
        PET_HIP_CHECK(hipMemcpyAsync(w.bbox, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice, st));
        PET_HIP_CHECK(hipMemcpyAsync(w.prms, prm.data(), n_sys * sizeof(NlParams), hipMemcpyHostToDevice, st));
        k_nlb_bbox<<<cdiv(n, T), T, 0, st>>>(d_pos, (int)n, w.prms, w.first_atom, (int)n_sys, w.bbox);
        std::vector<int> enc(6 * n_sys);
        if ((int)enc.size() < MAILBOX_INTS) {
            if (int rcr = read_back(w.bbox, (int)enc.size(), nullptr, 0, enc.data(), st)) return rcr;
        } else {
            PET_HIP_CHECK(hipMemcpyAsync(enc.data(), w.bbox, enc.size() * sizeof(int), hipMemcpyDeviceToHost, st));
            PET_HIP_CHECK(hipStreamSynchronize(st));
        }
        for (size_t k = 0; k < enc.size(); k++) h_bbox[k] = unord_f(enc[k]);
    }
    int64_t total_bins = 0;
    bool tile_ok = true;
    for (int64_t s = 0; s < n_sys; s++) {
        base[s] = (int)total_bins;
        if (h_first_atom[s + 1] == h_first_atom[s]) { prm[s].nb[0] = prm[s].nb[1] = prm[s].nb[2] = 1; prm[s].reach[0] = prm[s].reach[1] = prm[s].reach[2] = 0; total_bins += 1; continue; }
        total_bins += bin_params(prm[s], &height[3 * s], &h_bbox[6 * s], cutoff, h_first_atom[s + 1] - h_first_atom[s]);
        for (int a = 0; a < 3; a++) tile_ok = tile_ok && prm[s].reach[a] <= 100;  // image numbers are stored +128 in one byte
    }
    base[n_sys] = (int)total_bins;
    PET_REQUIRE(tile_ok, PET_ERR_UNSUPPORTED, "a cell is more than 100 times thinner than the cutoff");
    if (any_open) {
        PET_HIP_CHECK(hipMemcpyAsync(w.prms, prm.data(), n_sys * sizeof(NlParams), hipMemcpyHostToDevice, st));
        PET_HIP_CHECK(hipMemcpyAsync(w.bin_base, base.data(), (n_sys + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    } else {
        char* const d0 = reinterpret_cast<char*>(w.first_atom);
        const size_t span = (size_t)(reinterpret_cast<char*>(w.prms) - d0) + n_sys * sizeof(NlParams);
        static thread_local std::vector<char> block;  // (the copy is asynchronous: its source outlives the call)
        block.assign(span, 0);
        std::memcpy(block.data(), first.data(), (n_sys + 1) * sizeof(int));
        std::memcpy(block.data() + (reinterpret_cast<char*>(w.bin_base) - d0), base.data(), (n_sys + 1) * sizeof(int));
        std::memcpy(block.data() + (reinterpret_cast<char*>(w.prms) - d0), prm.data(), n_sys * sizeof(NlParams));
        PET_HIP_CHECK(hipMemcpyAsync(d0, block.data(), span, hipMemcpyHostToDevice, st));
    }
    k_nlb_bin<<<cdiv(n, T), T, 0, st>>>(d_pos, (int)n, w.prms, w.first_atom, w.bin_base, (int)n_sys, w.bin_key, w.atom_id,
                                        w.wrap, w.wpos, w.sys_of);
    int bits = 1;
    while (((int64_t)1 << bits) <= total_bins) bits++;
    size_t tb = w.tmp_bytes;
    PET_HIP_CHECK(rocprim::radix_sort_pairs(w.tmp, tb, w.bin_key, w.bin_key_sorted, w.atom_id, w.sorted_atoms, (size_t)n, 0,
                                            bits, st));
    k_nlb_post_sort<<<cdiv(std::max<int64_t>(total_bins + 1, n), T), T, 0, st>>>(w.bin_key_sorted, w.sorted_atoms, w.wpos, w.bin_base,
                                                                               (int)n, (int)n_sys, (int)total_bins, w.bin_start,
                                                                               w.bin_sys, w.spos, w.counts + n);
    k_nlb_pairs<0><<<cdiv(total_bins, 4), 256, 0, st>>>(d_pos, w.spos, w.wrap, w.bin_start, w.prms, w.bin_base, w.bin_sys,
                                                        (int)total_bins, w.counts, nullptr, nullptr, nullptr);
    tb = w.tmp_bytes;
    PET_HIP_CHECK(rocprim::exclusive_scan(w.tmp, tb, w.counts, w.offsets, 0, (size_t)n + 1, rocprim::plus<int>(), st));
    int total = 0;
    if (int rcr = read_back(w.offsets + n, 1, nullptr, 0, &total, st)) return rcr;
    *n_pairs = total;
    if (!d_pairs) return PET_OK;
    PET_REQUIRE(capacity >= total, PET_ERR_ARGUMENT,
                "pair buffer too small: " + std::to_string(total) + " pairs for a capacity of " + std::to_string(capacity));
    k_nlb_pairs<1><<<cdiv(total_bins, 4), 256, 0, st>>>(d_pos, w.spos, w.wrap, w.bin_start, w.prms, w.bin_base, w.bin_sys,
                                                        (int)total_bins, w.counts, w.offsets, d_pairs, d_vectors);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// one system: the batch path with a single entry (pairs then carry that system's own atom indices)
int nl_build(const float* d_pos, const float* h_cell, const int* h_pbc, int64_t n, float cutoff, void* ws, int* d_pairs,
             float* d_vectors, int64_t capacity, int64_t* n_pairs, hipStream_t st) {
    const int64_t first[2] = {0, n};
    return nl_build_batch(d_pos, h_cell, h_pbc, first, 1, cutoff, ws, d_pairs, d_vectors, capacity, n_pairs, st);
}

}  // namespace pet
