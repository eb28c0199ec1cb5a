// Periodic cell-list neighbour list on the GPU (replaces the vesin call at
// src/metatrain/utils/neighbor_lists.py:131-135; contract in oracle/nl.py).
//
// Atoms are wrapped into the cell and binned along the lattice directions with bin
// width >= cutoff where the cell allows it; a thin cell gets a single bin and the
// search walks over ceil(cutoff / height) periodic images instead. Bins are filled by
// a stable radix sort (atom order inside a bin = atom index), so the output order is
// deterministic. Two passes (count, exclusive scan, fill) emit pairs grouped by
// centre, i.e. already in the CSR order pet_graph_build wants.
#include "common.h"
#include "model.h"

#include <math.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

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

__global__ void k_nl_bbox(const float* __restrict__ pos, int n, NlParams prm, int* __restrict__ bbox) {
    // bbox[0..2] = min fractional coordinate, bbox[3..5] = max (ordered-int encoded)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float f[3] = {INFINITY, INFINITY, INFINITY}, F[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n) {
        float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        for (int a = 0; a < 3; a++) {
            float fr = x * prm.inv[a] + y * prm.inv[3 + a] + z * prm.inv[6 + a];
            f[a] = fr;
            F[a] = fr;
        }
    }
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) {
            f[a] = fminf(f[a], __shfl_xor(f[a], o));
            F[a] = fmaxf(F[a], __shfl_xor(F[a], o));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; a++) {
            atomicMin(bbox + a, ord_i(f[a]));
            atomicMax(bbox + 3 + a, ord_i(F[a]));
        }
    }
}

__device__ __forceinline__ void frac_of(const float* pos, int i, const NlParams& prm, float fr[3]) {
    float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
    for (int a = 0; a < 3; a++) fr[a] = x * prm.inv[a] + y * prm.inv[3 + a] + z * prm.inv[6 + a];
}

__global__ void k_nl_bin(const float* __restrict__ pos, int n, NlParams prm, int* __restrict__ bin_key,
                         int* __restrict__ atom_id, int* __restrict__ wrap, float* __restrict__ wpos) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float fr[3];
    frac_of(pos, i, prm, fr);
    int b[3], wr[3];
    for (int a = 0; a < 3; a++) {
        if (prm.pbc[a]) {
            float fl = floorf(fr[a]);
            wr[a] = (int)fl;
            float f = fr[a] - fl;
            b[a] = min((int)(f * prm.nb[a]), prm.nb[a] - 1);
        } else {
            wr[a] = 0;
            float f = (fr[a] - prm.origin[a]) / prm.extent[a];
            b[a] = max(0, min((int)(f * prm.nb[a]), prm.nb[a] - 1));
        }
    }
    bin_key[i] = (b[0] * prm.nb[1] + b[1]) * prm.nb[2] + b[2];
    atom_id[i] = i;
    wrap[3 * i] = wr[0]; wrap[3 * i + 1] = wr[1]; wrap[3 * i + 2] = wr[2];
    // wrapped cartesian position
    for (int k = 0; k < 3; k++)
        wpos[3 * i + k] = pos[3 * i + k] - (wr[0] * prm.cell[k] + wr[1] * prm.cell[3 + k] + wr[2] * prm.cell[6 + k]);
}

__global__ void k_nl_bin_start(const int* __restrict__ sorted_keys, int n, int nbins, int* __restrict__ bin_start) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nbins) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (sorted_keys[mid] < b) lo = mid + 1; else hi = mid;
    }
    bin_start[b] = lo;
}

// PASS 0: count, PASS 1: fill.  One thread per centre atom.
template <int PASS>
__global__ void k_nl_pairs(const float* __restrict__ pos, const float* __restrict__ wpos, const int* __restrict__ wrap,
                           const int* __restrict__ sorted_atoms, const int* __restrict__ bin_start, int n,
                           NlParams prm, int* __restrict__ counts, const int* __restrict__ offsets,
                           int* __restrict__ pairs, float* __restrict__ vectors) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float fr[3];
    frac_of(pos, i, prm, fr);
    int b[3];
    for (int a = 0; a < 3; a++) {
        if (prm.pbc[a]) {
            float f = fr[a] - floorf(fr[a]);
            b[a] = min((int)(f * prm.nb[a]), prm.nb[a] - 1);
        } else {
            float f = (fr[a] - prm.origin[a]) / prm.extent[a];
            b[a] = max(0, min((int)(f * prm.nb[a]), prm.nb[a] - 1));
        }
    }
    const float xi = wpos[3 * i], yi = wpos[3 * i + 1], zi = wpos[3 * i + 2];
    const int wi0 = wrap[3 * i], wi1 = wrap[3 * i + 1], wi2 = wrap[3 * i + 2];
    int count = 0;
    int64_t out = PASS == 1 ? offsets[i] : 0;
    for (int da = -prm.reach[0]; da <= prm.reach[0]; da++) {
        int ba = b[0] + da, sa = 0;
        if (prm.pbc[0]) { sa = (ba >= 0) ? ba / prm.nb[0] : -((-ba + prm.nb[0] - 1) / prm.nb[0]); ba -= sa * prm.nb[0]; }
        else if (ba < 0 || ba >= prm.nb[0]) continue;
        for (int db = -prm.reach[1]; db <= prm.reach[1]; db++) {
            int bb = b[1] + db, sb = 0;
            if (prm.pbc[1]) { sb = (bb >= 0) ? bb / prm.nb[1] : -((-bb + prm.nb[1] - 1) / prm.nb[1]); bb -= sb * prm.nb[1]; }
            else if (bb < 0 || bb >= prm.nb[1]) continue;
            for (int dc = -prm.reach[2]; dc <= prm.reach[2]; dc++) {
                int bc = b[2] + dc, sc = 0;
                if (prm.pbc[2]) { sc = (bc >= 0) ? bc / prm.nb[2] : -((-bc + prm.nb[2] - 1) / prm.nb[2]); bc -= sc * prm.nb[2]; }
                else if (bc < 0 || bc >= prm.nb[2]) continue;
                const float ox = sa * prm.cell[0] + sb * prm.cell[3] + sc * prm.cell[6];
                const float oy = sa * prm.cell[1] + sb * prm.cell[4] + sc * prm.cell[7];
                const float oz = sa * prm.cell[2] + sb * prm.cell[5] + sc * prm.cell[8];
                const int bin = (ba * prm.nb[1] + bb) * prm.nb[2] + bc;
                for (int q = bin_start[bin]; q < bin_start[bin + 1]; q++) {
                    const int j = sorted_atoms[q];
                    if (j == i && sa == 0 && sb == 0 && sc == 0) continue;
                    // (w_j - w_i) + offset: bitwise antisymmetric under (i,j,S) <-> (j,i,-S), so the list is
                    // always a full list even for pairs within an ulp of the cutoff
                    const float dx = (wpos[3 * j] - xi) + ox, dy = (wpos[3 * j + 1] - yi) + oy, dz = (wpos[3 * j + 2] - zi) + oz;
                    const float d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 < prm.cutoff2) {
                        if (PASS == 1) {
                            // shift relative to the caller's (unwrapped) positions
                            const int Sa = sa + wi0 - wrap[3 * j], Sb = sb + wi1 - wrap[3 * j + 1], Sc = sc + wi2 - wrap[3 * j + 2];
                            pairs[5 * out] = i; pairs[5 * out + 1] = j;
                            pairs[5 * out + 2] = Sa; pairs[5 * out + 3] = Sb; pairs[5 * out + 4] = Sc;
                            if (vectors) {
                                // D = r_j - r_i + S.cell evaluated like structures.py:212-220
                                vectors[3 * out] = (pos[3 * j] - pos[3 * i]) + (Sa * prm.cell[0] + Sb * prm.cell[3] + Sc * prm.cell[6]);
                                vectors[3 * out + 1] = (pos[3 * j + 1] - pos[3 * i + 1]) + (Sa * prm.cell[1] + Sb * prm.cell[4] + Sc * prm.cell[7]);
                                vectors[3 * out + 2] = (pos[3 * j + 2] - pos[3 * i + 2]) + (Sa * prm.cell[2] + Sb * prm.cell[5] + Sc * prm.cell[8]);
                            }
                            out++;
                        }
                        count++;
                    }
                }
            }
        }
    }
    if (PASS == 0) counts[i] = count;
}

struct NlWs {
    int *bin_key, *bin_key_sorted, *atom_id, *sorted_atoms, *wrap, *bin_start, *counts, *offsets;
    float *wpos, *bbox;
    void* tmp;
    size_t tmp_bytes;
    size_t total;
};

static int carve_nl(NlWs& w, void* base, int64_t n) {
    Carver c(base);
    const int64_t na = n > 0 ? n : 1;
    w.bin_key = c.take<int>(na);
    w.bin_key_sorted = c.take<int>(na);
    w.atom_id = c.take<int>(na);
    w.sorted_atoms = c.take<int>(na);
    w.wrap = c.take<int>(3 * na);
    w.bin_start = c.take<int>(na + 2 + 64);
    w.counts = c.take<int>(na + 1);
    w.offsets = c.take<int>(na + 1);
    w.wpos = c.take<float>(3 * na);
    w.bbox = c.take<float>(8);
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

int64_t nl_workspace_bytes(int64_t n_atoms) {
    NlWs w;
    if (carve_nl(w, nullptr, n_atoms) != PET_OK) return -1;
    return (int64_t)w.total;
}

static void cross(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static double norm3(const double* a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

int nl_build(const float* d_pos, const float* h_cell, const int* h_pbc, int64_t n, float cutoff, void* ws,
             int* d_pairs, float* d_vectors, int64_t capacity, int64_t* n_pairs, hipStream_t st) {
    *n_pairs = 0;
    if (n == 0) return PET_OK;
    PET_REQUIRE(cutoff > 0, PET_ERR_ARGUMENT, "cutoff must be positive");
    NlWs w;
    int rc = carve_nl(w, ws, n);
    if (rc) return rc;
    // effective lattice: non-periodic axes get a unit vector completing the basis
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
    NlParams prm;
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
    // heights of the effective cell along each axis
    double height[3];
    for (int a = 0; a < 3; a++) {
        double x[3];
        cross(c[(a + 1) % 3], c[(a + 2) % 3], x);
        height[a] = fabs(det) / norm3(x);
    }
    float h_bbox[6] = {0, 0, 0, 1, 1, 1};
    const bool any_open = !(h_pbc[0] && h_pbc[1] && h_pbc[2]);
    const int T = 256;
    if (any_open) {
        int init[6];
        for (int k = 0; k < 3; k++) { init[k] = ord_i(INFINITY); init[3 + k] = ord_i(-INFINITY); }
        int* bb = reinterpret_cast<int*>(w.bbox);
        PET_HIP_CHECK(hipMemcpyAsync(bb, init, sizeof(init), hipMemcpyHostToDevice, st));
        k_nl_bbox<<<cdiv(n, T), T, 0, st>>>(d_pos, (int)n, prm, bb);
        int enc[6];
        PET_HIP_CHECK(hipMemcpyAsync(enc, bb, sizeof(enc), hipMemcpyDeviceToHost, st));
        PET_HIP_CHECK(hipStreamSynchronize(st));
        for (int k = 0; k < 6; k++) h_bbox[k] = unord_f(enc[k]);
    }
    int64_t total_bins = 1;
    for (int a = 0; a < 3; a++) {
        double span = height[a];  // length covered by the bins along this axis
        if (h_pbc[a]) {
            prm.origin[a] = 0.f;
            prm.extent[a] = 1.f;
        } else {
            // fractional coordinate of a unit vector axis is a length already
            float lo = h_bbox[a], hi = h_bbox[3 + a];
            float ext = fmaxf(hi - lo, 1e-3f) * 1.0001f + 1e-4f;
            prm.origin[a] = lo - 0.5e-4f;
            prm.extent[a] = ext;
            span = ext * height[a];
        }
        int nb = (int)floor(span / cutoff);
        nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
        prm.nb[a] = nb;
        total_bins *= nb;
    }
    // keep the bin table inside the workspace carve (n + 64 entries)
    while (total_bins > n + 32) {
        int a = 0;
        for (int k = 1; k < 3; k++) if (prm.nb[k] > prm.nb[a]) a = k;
        if (prm.nb[a] <= 1) break;
        total_bins = total_bins / prm.nb[a];
        prm.nb[a] = (prm.nb[a] + 1) / 2;
        total_bins *= prm.nb[a];
    }
    for (int a = 0; a < 3; a++) {
        double span = h_pbc[a] ? height[a] : prm.extent[a] * height[a];
        double width = span / prm.nb[a];
        prm.reach[a] = (int)ceil(cutoff / width);
        if (!h_pbc[a] && prm.reach[a] > prm.nb[a]) prm.reach[a] = prm.nb[a];
    }
    k_nl_bin<<<cdiv(n, T), T, 0, st>>>(d_pos, (int)n, prm, w.bin_key, w.atom_id, w.wrap, w.wpos);
    size_t tb = w.tmp_bytes;
    PET_HIP_CHECK(rocprim::radix_sort_pairs(w.tmp, tb, w.bin_key, w.bin_key_sorted, w.atom_id, w.sorted_atoms,
                                            (size_t)n, 0, 32, st));
    k_nl_bin_start<<<cdiv(total_bins + 1, T), T, 0, st>>>(w.bin_key_sorted, (int)n, (int)total_bins, w.bin_start);
    PET_HIP_CHECK(hipMemsetAsync(w.counts + n, 0, sizeof(int), st));
    k_nl_pairs<0><<<cdiv(n, 128), 128, 0, st>>>(d_pos, w.wpos, w.wrap, w.sorted_atoms, w.bin_start, (int)n, prm,
                                                 w.counts, nullptr, nullptr, nullptr);
    tb = w.tmp_bytes;
    PET_HIP_CHECK(rocprim::exclusive_scan(w.tmp, tb, w.counts, w.offsets, 0, (size_t)n + 1, rocprim::plus<int>(), st));
    int total = 0;
    PET_HIP_CHECK(hipMemcpyAsync(&total, w.offsets + n, sizeof(int), hipMemcpyDeviceToHost, st));
    PET_HIP_CHECK(hipStreamSynchronize(st));
    *n_pairs = total;
    if (!d_pairs) return PET_OK;
    PET_REQUIRE(capacity >= total, PET_ERR_ARGUMENT, "pair buffer too small");
    k_nl_pairs<1><<<cdiv(n, 128), 128, 0, st>>>(d_pos, w.wpos, w.wrap, w.sorted_atoms, w.bin_start, (int)n, prm,
                                                 w.counts, w.offsets, d_pairs, d_vectors);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

}  // namespace pet
