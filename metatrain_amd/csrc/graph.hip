// Preprocessing kernels: edge geometry, non-strict filter, CSR (= NEF slot) order,
// ij -> ji map, NEF export.  HBM-bound integer/gather work: one thread per edge or
// per atom, coalesced SoA reads, no atomics on floating point.
//
// Reference semantics: pet/modules/structures.py:115-378, pet/modules/nef.py:34-251,
// pet/modules/utilities.py:4-39 (see include/pet_hip.h for the boundary).
#include "common.h"
#include "model.h"
#include "cutoff.h"

#include <atomic>
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace pet {

// ----------------------------------------------------------------------------------
// A few device integers to the host, on the critical path of a small box's step (the graph build cannot size its launches
// before it knows the kept edge count; the neighbour list cannot return before it knows the pair count). hipMemcpyAsync to
// a stack variable + hipStreamSynchronize costs 27 us per read-back on this stack (pageable destination; 16 us into pinned
// memory); a one-wave kernel that writes the values and then a sequence word into coherent pinned host memory that the host
// polls costs under 10 us, launch included (tools/ubench/readback_latency.hip). One mailbox per host thread.
// ----------------------------------------------------------------------------------
__global__ void k_publish(const int* __restrict__ a, int na, const int* __restrict__ b, int nb, volatile int* host, int seq) {
    const int t = threadIdx.x;
    if (t < na) host[t] = a[t];
    else if (t < na + nb) host[t] = b[t - na];
    __threadfence_system();
    __syncthreads();
    if (t == 0) host[MAILBOX_INTS - 1] = seq;
}

namespace {
struct Mailbox {
    int* p = nullptr;
    int seq = 0;
    ~Mailbox() { if (p) (void)hipHostFree(p); }
};
thread_local Mailbox t_mailbox;
}  // namespace

int read_back(const int* d_a, int na, const int* d_b, int nb, int* out, hipStream_t st) {
    PET_REQUIRE(na >= 0 && nb >= 0 && na + nb <= MAILBOX_INTS - 1, PET_ERR_ARGUMENT, "read_back: too many integers");
    Mailbox& mb = t_mailbox;
    if (!mb.p) {
        PET_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&mb.p), MAILBOX_INTS * sizeof(int),
                                    hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable));
        std::memset(mb.p, 0, MAILBOX_INTS * sizeof(int));
    }
    const int seq = ++mb.seq == 0 ? ++mb.seq : mb.seq;  // 0 is the word's initial value
    k_publish<<<1, MAILBOX_INTS, 0, st>>>(d_a, na, d_b, nb, mb.p, seq);
    PET_HIP_CHECK(hipGetLastError());
    for (uint64_t spins = 1;; spins++) {
        if (__atomic_load_n(&mb.p[MAILBOX_INTS - 1], __ATOMIC_ACQUIRE) == seq) break;
        __builtin_ia32_pause();
        if ((spins & 0x3FFFF) == 0) {  // every few ms: a stream that failed, or drained without the word (not seen so far), ends the wait
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) {
                PET_HIP_CHECK(hipStreamSynchronize(st));
                PET_REQUIRE(__atomic_load_n(&mb.p[MAILBOX_INTS - 1], __ATOMIC_ACQUIRE) == seq, PET_ERR_HIP,
                            "read_back: the stream drained without the mailbox word arriving");
                break;
            }
            if (q != hipErrorNotReady) PET_HIP_CHECK(q);
        }
    }
    for (int k = 0; k < na + nb; k++) out[k] = mb.p[k];
    return PET_OK;
}

// ----------------------------------------------------------------------------------
// adaptive cutoff (pet/modules/adaptive_cutoff.py:46-229, "solver" method)
// ----------------------------------------------------------------------------------
// bump(d; r, w) and d bump / d r in closed form (adaptive_cutoff.py:74-93)
__device__ __forceinline__ void adaptive_term(float d, float r, float w, float& f, float& df) {
    const float scaled = (d - (r - w)) / w;
    const bool active = scaled > 0.0f && scaled < 1.0f;
    const float safe = fminf(fmaxf(scaled, 1e-6f), 1.0f - 1e-6f);
    const float s = 3.14159274f * safe;
    const float sn = sinf(s);
    const float t = tanhf(cosf(s) / sn);
    f = active ? 0.5f * (1.0f + t) : (scaled <= 0.0f ? 1.0f : 0.0f);
    df = active ? (0.5f * 3.14159274f / w) * (1.0f - t * t) / (sn * sn) : 0.0f;
}

// one 16-lane group per atom over its row of the all-edge CSR: ten Newton-bisection steps, then the
// implicit-function-theorem step and the clamp to [rc/16, rc]
__global__ void k_adaptive_solve(const int* __restrict__ rowptr0, const int* __restrict__ perm0,
                                 const float4* __restrict__ vin, float* __restrict__ r_atom,
                                 float* __restrict__ r_newton, float* __restrict__ inv_dn, int N, float rc, float w,
                                 float target) {
    const int gid = blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    const int a = gid < N ? gid : N - 1;
    const int p0 = rowptr0[a], p1 = rowptr0[a + 1];
    const float inv_rc = 1.0f / rc;
    float r_lo = 0.f, r_hi = rc, r = 0.5f * rc;
    float n = 0.f, dn = 0.f;
    for (int it = 0; it <= 10; it++) {
        float fs = 0.f, dfs = 0.f;
        for (int p = p0 + l; p < p1; p += 16) {
            float f, df;
            adaptive_term(vin[perm0[p]].w, r, w, f, df);
            fs += f;
            dfs += df;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { fs += __shfl_xor(fs, o); dfs += __shfl_xor(dfs, o); }
        const float x = r * inv_rc;
        n = fs + target * x * x * x;
        dn = dfs + 3.0f * target * x * x * inv_rc;
        if (it == 10) break;  // the eleventh evaluation is at the root: dn_root and the residual
        const float f = n - target;
        const bool below = f <= 0.f;
        r_lo = below ? r : r_lo;
        r_hi = below ? r_hi : r;
        const float r_nt = r - f / fmaxf(dn, 1e-6f);
        r = (r_nt >= r_lo && r_nt <= r_hi) ? r_nt : 0.5f * (r_lo + r_hi);
    }
    const float idn = 1.0f / fmaxf(dn, 1e-6f);
    const float r_ift = r - (n - target) * idn;
    const float lo = rc * (1.0f / 16.0f);
    const bool clamped = r_ift < lo || r_ift > rc;
    if (l == 0 && gid < N) {
        r_atom[gid] = fminf(fmaxf(r_ift, lo), rc);
        r_newton[gid] = r;
        inv_dn[gid] = clamped ? 0.f : idn;
    }
}

// The legacy "grid" method (adaptive_cutoff.py:232-395), one 16-lane group per atom over its row of the all-edge CSR:
//   n_k    = sum over the atom's edges of bump(d; p_k, w) on the probe grid p_k = 0.5 + k w / 4 < rc      (:297-327)
//   diff_k = n_k - target + target x_k^3, x = linspace(0, 1, K)                                          (:349-365)
//   wd_k   = max(|torch.gradient(diff)_k|, 1e-12); logw_k = -(diff_k / wd_k)^2 / 2; w = softmax(logw)    (:366-393)
//   r      = sum_k p_k w_k                                                                               (:291-293)
// and, for the reverse pass, dr / dn_k through all of that (written out by hand below). The reference subtracts the
// GLOBAL maximum of logw before the exponential; the row maximum used here gives the same weights wherever the
// reference's do not underflow.
__global__ void k_adaptive_grid(const int* __restrict__ rowptr0, const int* __restrict__ perm0,
                                const float4* __restrict__ vin, float* __restrict__ r_atom, float* __restrict__ drdn,
                                int N, float w, float target, int K, float pmin, float dp) {
    __shared__ float s_diff[16][GRID_MAX_PROBES], s_w[16][GRID_MAX_PROBES], s_e[16][GRID_MAX_PROBES];
    const int grp = threadIdx.x >> 4;
    const int gid = blockIdx.x * (blockDim.x / 16) + grp;
    const int l = threadIdx.x & 15;
    const int a = gid < N ? gid : N - 1;
    const int p0 = rowptr0[a], p1 = rowptr0[a + 1];
    for (int k = 0; k < K; k++) {
        const float pk = pmin + (float)k * dp;
        float fs = 0.f;
        for (int p = p0 + l; p < p1; p += 16) {
            float f, df;
            adaptive_term(vin[perm0[p]].w, pk, w, f, df);
            fs += f;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) fs += __shfl_xor(fs, o);
        const float x = (float)k / (float)(K - 1);
        if (l == 0) s_diff[grp][k] = fs - target + target * x * x * x;
    }
    if (l != 0 || gid >= N) return;
    float* diff = s_diff[grp];
    float* wt = s_w[grp];
    float* ev = s_e[grp];
    // logw (kept in wt), u and the slope sign (packed into ev for now)
    float mx = -INFINITY;
    for (int k = 0; k < K; k++) {
        const float gk = k == 0 ? diff[1] - diff[0] : (k == K - 1 ? diff[K - 1] - diff[K - 2] : 0.5f * (diff[k + 1] - diff[k - 1]));
        const float wd = fmaxf(fabsf(gk), 1e-12f);
        const float u = diff[k] / wd;
        wt[k] = -0.5f * u * u;
        mx = fmaxf(mx, wt[k]);
    }
    float sum = 0.f;
    for (int k = 0; k < K; k++) {
        wt[k] = expf(wt[k] - mx);
        sum += wt[k];
    }
    const float isum = 1.0f / sum;
    float r = 0.f;
    for (int k = 0; k < K; k++) {
        wt[k] *= isum;
        r += (pmin + (float)k * dp) * wt[k];
    }
    r_atom[gid] = r;
    // reverse: c_k = dr/du_k = -w_k (p_k - r) u_k; dr/ddiff_j = c_j / wd_j + sum_k e_k G_kj with
    // e_k = dr/dg_k = -c_k u_k sign(g_k) / wd_k where |g_k| > 1e-12, and G the finite-difference stencil of torch.gradient
    for (int k = 0; k < K; k++) {
        const float gk = k == 0 ? diff[1] - diff[0] : (k == K - 1 ? diff[K - 1] - diff[K - 2] : 0.5f * (diff[k + 1] - diff[k - 1]));
        const float ag = fabsf(gk);
        const float wd = fmaxf(ag, 1e-12f);
        const float u = diff[k] / wd;
        const float c = -wt[k] * ((pmin + (float)k * dp) - r) * u;
        ev[k] = ag > 1e-12f ? -c * u * (gk > 0.f ? 1.0f : -1.0f) / wd : 0.f;
        wt[k] = c / wd;  // the direct term (the weights themselves are not needed any more)
    }
    for (int j = 0; j < K; j++) {
        float v = wt[j];
        // column j of the stencil: rows j - 1 and j + 1 (interior rows: +-1/2), the two one-sided boundary rows
        if (j >= 1) v += ev[j - 1] * (j - 1 == 0 ? 1.0f : 0.5f);           // G[j-1][j]
        if (j + 1 <= K - 1) v -= ev[j + 1] * (j + 1 == K - 1 ? 1.0f : 0.5f);  // G[j+1][j]
        if (j == 0) v -= ev[0];                                             // G[0][0] = -1
        if (j == K - 1) v += ev[K - 1];                                     // G[K-1][K-1] = +1
        drdn[(int64_t)gid * GRID_MAX_PROBES + j] = v;
    }
}

// pair cutoffs (r_i + r_j) / 2 and the mask d <= pair cutoff (structures.py:248-252)
__global__ void k_adaptive_keep(const int* __restrict__ centers, const int* __restrict__ neighbors,
                                const float4* __restrict__ vin, const float* __restrict__ r_atom,
                                int* __restrict__ keep, int* __restrict__ sort_keys, int* __restrict__ sort_vals,
                                int n_edges, int n_nodes) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int i = centers[e], j = neighbors[e];
    const bool ok = i >= 0 && i < n_nodes && j >= 0 && j < n_nodes;  // bad indices were counted by k_edge_geometry
    const float pc = ok ? (r_atom[i] + r_atom[j]) / 2.0f : -1.0f;
    const int kp = (ok && vin[e].w <= pc) ? 1 : 0;
    keep[e] = kp;
    sort_keys[e] = kp ? i : n_nodes;
    sort_vals[e] = e;
}

__global__ void k_gather_all(const int* __restrict__ perm0, const int* __restrict__ neighbors,
                             const int* __restrict__ shifts, int* __restrict__ nbr0, int* __restrict__ shift0,
                             int n) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int e = perm0[q];
    nbr0[q] = neighbors[e];
    shift0[3 * q] = shifts[3 * e];
    shift0[3 * q + 1] = shifts[3 * e + 1];
    shift0[3 * q + 2] = shifts[3 * e + 2];
}

// ----------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------
__global__ void k_species_index(const int* __restrict__ species, const int* __restrict__ table,
                                int table_len, int* __restrict__ sp, int n, int* __restrict__ n_unknown,
                                const int* __restrict__ sys_in, int* __restrict__ sys_out, int n_systems,
                                int* __restrict__ n_bad_sys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    {   // system_indices must be non-decreasing runs inside [0, n_systems) (concatenate_structures, structures.py:87-91):
        // the per-system sums and the cell lookup rely on it. Violations are reported by the host; the stored index
        // is clamped so that the kernels stay in bounds.
        const int sy = sys_in[i];
        if (sy < 0 || sy >= n_systems || (i > 0 && sys_in[i - 1] > sy)) atomicAdd(n_bad_sys, 1);
        sys_out[i] = min(max(sy, 0), n_systems - 1);
    }
    int z = species[i];
    int s = (z >= 0 && z < table_len) ? table[z] : -1;
    if (s < 0) {  // not one of the model's atomic_types: reported by the host, index 0 keeps the kernels in bounds
        atomicAdd(n_unknown, 1);
        s = 0;
    }
    sp[i] = s;
}

static int g_sorted_shortcut = 1;  // pet_config_set("sorted_shortcut", 0): always run the radix sort of the edges
void set_sorted_shortcut(int v) { g_sorted_shortcut = v ? 1 : 0; }

// structures.py:206-221 and the non-strict mask of :265-267
__global__ void k_edge_geometry(const float* __restrict__ pos, const float* __restrict__ cells,
                                const int* __restrict__ centers, const int* __restrict__ neighbors,
                                const int* __restrict__ shifts, const int* __restrict__ sys,
                                float4* __restrict__ vin, int* __restrict__ keep,
                                int* __restrict__ sort_keys, int* __restrict__ sort_vals,
                                int n_edges, int n_nodes, float cutoff, int strict, int* __restrict__ n_bad,
                                int* __restrict__ unsorted, int* __restrict__ kidx_iota = nullptr) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    if (kidx_iota) kidx_iota[e] = e;  // a list assumed sorted with nothing to drop: edge e is kept edge e (graph_build)
    int i = centers[e], j = neighbors[e];
    // *unsorted stays 0 only if every edge is kept and the centres come in non-decreasing order: the sort keys are then
    // sorted as they stand and graph_build skips the radix sort (a list from pet_nl_build / vesin is ordered like that)
    if (e > 0 && centers[e - 1] > i) *unsorted = 1;
    if (i < 0 || i >= n_nodes || j < 0 || j >= n_nodes) {  // reported by the host; the edge is dropped
        atomicAdd(n_bad, 1);
        *unsorted = 1;
        vin[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        keep[e] = 0;
        sort_keys[e] = n_nodes;
        sort_vals[e] = e;
        return;
    }
    const float* c = cells + 9 * sys[i];
    float sa = (float)shifts[3 * e], sb = (float)shifts[3 * e + 1], sc = (float)shifts[3 * e + 2];
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float contrib = fmaf(sc, c[6 + k], fmaf(sb, c[3 + k], sa * c[k]));
        v[k] = (pos[3 * j + k] - pos[3 * i + k]) + contrib;
    }
    float d0 = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + 1e-15f;
    int kp = strict ? 1 : (d0 <= cutoff ? 1 : 0);  // strict == 2: adaptive first pass, every edge
    vin[e] = make_float4(v[0], v[1], v[2], d0);
    keep[e] = kp;
    if (!kp) *unsorted = 1;
    sort_keys[e] = kp ? i : n_nodes;  // dropped edges sort behind every real centre
    sort_vals[e] = e;
}

// rowptr[i] = first sorted position whose key >= i
// veto (graph_build, a list ASSUMED to be sorted): if the word is set the keys are not sorted after all -- every row is left
// empty, so that the kernels behind this one (which take their counts from rowptr / scalars[0]) touch nothing before the host
// sees the word and builds again with the sort.
// max_out (the kept-edge CSR): the largest row length as well -- rowptr[i + 1] comes from the neighbouring thread through LDS (the
// block's last thread searches for it), one atomicMax per block; this was a launch of its own (k_max_nbr: 5 us of a small box's
// 70-us graph build, 20 us at 100 000 atoms, where its one atomic per wave serialised). bucket_counts: k_bucket_count's histogram too.
__global__ __launch_bounds__(256) void k_rowptr(const int* __restrict__ sorted_keys, int n_edges, int* __restrict__ rowptr,
                                                int n_nodes, int* __restrict__ scalars, const int* __restrict__ veto = nullptr,
                                                int* __restrict__ max_out = nullptr, int* __restrict__ bucket_counts = nullptr) {
    __shared__ int lo_s[257];
    __shared__ int wmax[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_hi = (veto && *veto) ? 0 : n_edges;
    auto lower = [&](int key) {
        int lo = 0, hi = n_hi;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sorted_keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    int lo = 0;
    if (i <= n_nodes) {
        lo = lower(i);
        rowptr[i] = lo;
        if (i == n_nodes) scalars[0] = lo;
    }
    if (!max_out) return;
    lo_s[threadIdx.x] = lo;
    if (threadIdx.x == blockDim.x - 1) lo_s[blockDim.x] = i + 1 <= n_nodes ? lower(i + 1) : lo;
    __syncthreads();
    int v = i < n_nodes ? lo_s[threadIdx.x + 1] - lo : 0;
    if (bucket_counts) {  // k_bucket_count's histogram of the atoms by attention tile count, from the row length at hand
        const int b = i < n_nodes ? min((v + 1 + 15) >> 4, 5) - 1 : -1;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const unsigned long long m = __ballot(b == k);
            if ((threadIdx.x & 63) == 0 && m) atomicAdd(&bucket_counts[k], __popcll(m));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (v > 0) atomicMax(max_out, v);
    }
}

// Atoms by attention tile count (bucket b = min(ceil((deg + 1) / 16), 5) - 1): counts, then a fill whose order inside a
// bucket is whatever the atomics give -- every atom's attention is independent, so results do not depend on it.
__global__ void k_bucket_count(const int* __restrict__ rowptr, int n_nodes, int* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = i < n_nodes ? min((rowptr[i + 1] - rowptr[i] + 1 + 15) >> 4, 5) - 1 : -1;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const unsigned long long m = __ballot(b == k);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counts[k], __popcll(m));
    }
}
__global__ void k_bucket_fill(const int* __restrict__ rowptr, int n_nodes, const int* __restrict__ counts,
                              int* __restrict__ cursors, int* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = i < n_nodes ? rowptr[i] : 0, r1 = i < n_nodes ? rowptr[i + 1] : 0;
    const int b = i < n_nodes ? min((r1 - r0 + 1 + 15) >> 4, 5) - 1 : -1;
    const int lane = threadIdx.x & 63;
    int start = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const unsigned long long m = __ballot(b == k);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&cursors[k], __popcll(m));
        base = __shfl(base, 0);
        if (b == k) order[start + base + __popcll(m & ((1ull << lane) - 1ull))] = i;
        start += counts[k];
    }
}
// ---- attention tiles of the fused block (pet_ablk.hip; Graph::tile_desc) -------------------------------------------
// atoms of at most 32 tokens counted and listed by token count t = neighbours + 1 (counts: hist[t], t = 1 .. 32)
// Deterministic (every build of the same graph pairs the same atoms: an atom's position inside its tile decides the
// summation order of its attention products, so the pairing must not depend on the order atomics happen to land in):
// per-block counts, one scan over the blocks per bin, then ranks from ballots in thread order.
__global__ void k_thist(const int* __restrict__ rowptr, int n_nodes, int* __restrict__ hist, int* __restrict__ blockhist) {
    __shared__ int lh[33];
    if (threadIdx.x < 33) lh[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = i < n_nodes ? rowptr[i + 1] - rowptr[i] + 1 : 0;
    if (t >= 1 && t <= 32) atomicAdd(&lh[t], 1);  // a count: the same whatever the order
    __syncthreads();
    if (threadIdx.x < 33) {
        blockhist[blockIdx.x * 33 + threadIdx.x] = lh[threadIdx.x];
        if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
    }
}
// blockhist[b][t] <- first position of block b's atoms of t tokens (bins laid out one after the other)
__global__ void k_tscan(int* __restrict__ blockhist, int n_blocks, const int* __restrict__ hist) {
    const int t = blockIdx.x + 1, lane = threadIdx.x;  // one wave per bin: 64 blocks per wave scan
    int run = 0;
    for (int u = 1; u < t; u++) run += hist[u];
    for (int b0 = 0; b0 < n_blocks; b0 += 64) {
        const int b = b0 + lane;
        const int c = b < n_blocks ? blockhist[b * 33 + t] : 0;
        int incl = c;
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (b < n_blocks) blockhist[b * 33 + t] = run + incl - c;
        run += __shfl(incl, 63);
    }
}
__global__ void k_tsort(const int* __restrict__ rowptr, int n_nodes, const int* __restrict__ blockbase,
                        int* __restrict__ out) {
    __shared__ int wcount[4][33];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = i < n_nodes ? rowptr[i + 1] - rowptr[i] + 1 : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rank = 0;
    for (int u = 1; u <= 32; u++) {
        const unsigned long long m = __ballot(t == u);
        if (t == u) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcount[wave][u] = __popcll(m);
    }
    __syncthreads();
    if (t >= 1 && t <= 32) {
        int base = blockbase[blockIdx.x * 33 + t];
        for (int w = 0; w < wave; w++) base += wcount[w][t];
        out[base + rank] = i;
    }
}
static int bucket_atoms_by_tile_count(Graph& g, hipStream_t st, bool counted = false) {  // scalars[8..17] were zeroed with the rest
    if (g.n_nodes <= 0) return PET_OK;
    const int T = 256;
    if (!counted)  // (graph_build: k_rowptr has filled the histogram)
        k_bucket_count<<<cdiv(g.n_nodes, T), T, 0, st>>>(g.rowptr, (int)g.n_nodes, g.scalars + 8);
    k_bucket_fill<<<cdiv(g.n_nodes, T), T, 0, st>>>(g.rowptr, (int)g.n_nodes, g.scalars + 8, g.scalars + 13, g.atom_order);
    // the per-atom attention tiles serve the fused block only (pet_ablk.hip: graphs of at least ABLK_MIN_TILES tiles, or forced; tiles <= atoms)
    g.tiles_planned = g.n_nodes >= ABLK_MIN_TILES || (attn_fused() & 4);
    if (!g.tiles_planned) return PET_OK;
    const int nb = cdiv(g.n_nodes, T);
    k_thist<<<nb, T, 0, st>>>(g.rowptr, (int)g.n_nodes, g.scalars + 24, g.tsort_tmp);
    k_tscan<<<32, 64, 0, st>>>(g.tsort_tmp, nb, g.scalars + 24);
    k_tsort<<<nb, T, 0, st>>>(g.rowptr, (int)g.n_nodes, g.tsort_tmp, g.atoms_by_t);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}
static void set_bucket_starts(Graph& g, const int* counts) {
    g.bucket_start[0] = 0;
    for (int k = 0; k < 5; k++) g.bucket_start[k + 1] = g.bucket_start[k] + counts[k];
}

// the plan: segments of tiles; a segment takes `n` tiles, atom A of tile k from bin ta at offset offa + k, atom B (tb > 0)
// from bin tb at offset offb + k
struct TilePlan {
    int n_seg;
    int first[72];  // first tile of the segment (first[n_seg] = number of tiles)
    short ta[72], tb[72];
    int offa[72], offb[72];
    int binstart[34];
    int tok0[72];  // tokens of all tiles before the segment (dense per-tile buffers of pet_ablk.hip: tile k starts at tok0 + r * (ta + tb))
};
__global__ void k_tile_fill(TilePlan plan, const int* __restrict__ by_t, const int* __restrict__ rowptr,
                            int4* __restrict__ desc) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= plan.first[plan.n_seg]) return;
    int sgm = 0;
    while (sgm + 1 < plan.n_seg && plan.first[sgm + 1] <= k) sgm++;
    const int r = k - plan.first[sgm];
    const int ta = plan.ta[sgm], tb = plan.tb[sgm];
    const int A = by_t[plan.binstart[ta] + plan.offa[sgm] + r];
    const int B = tb > 0 ? by_t[plan.binstart[tb] + plan.offb[sgm] + r] : 0;
    desc[2 * k] = make_int4(A, rowptr[A], ta, B);
    desc[2 * k + 1] = make_int4(tb > 0 ? rowptr[B] : 0, tb, plan.tok0[sgm] + r * (ta + tb), 0);
}
__global__ void k_tile_fill_big(const int* __restrict__ atoms, int n, const int* __restrict__ rowptr, int4* __restrict__ desc) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int A = atoms[k];
    desc[2 * k] = make_int4(A, rowptr[A], rowptr[A + 1] - rowptr[A] + 1, 0);
    desc[2 * k + 1] = make_int4(0, 0, 0, 0);
}
// host: pair the smallest atoms with the largest partners that still fit 32 slots (hist[t]: atoms of t tokens)
static int plan_attention_tiles(Graph& g, const int* hist, hipStream_t st) {
    g.n_tiles1 = g.n_tiles2 = 0;
    if (g.n_nodes <= 0 || !g.tiles_planned) return PET_OK;
    static const bool pairing = !(getenv("PET_HIP_TILE_PAIRS") && getenv("PET_HIP_TILE_PAIRS")[0] == '0');
    TilePlan plan;
    int rem[33], off[33];
    plan.binstart[0] = plan.binstart[1] = 0;
    for (int t = 1; t <= 32; t++) {
        rem[t] = hist[t];
        off[t] = 0;
        plan.binstart[t + 1] = plan.binstart[t] + hist[t];
    }
    int nseg = 0, ntile = 0, ntok = 0;
    auto add = [&](int ta, int oa, int tb, int ob, int n) {
        plan.first[nseg] = ntile;
        plan.tok0[nseg] = ntok;
        ntok += n * (ta + tb);
        plan.ta[nseg] = (short)ta; plan.tb[nseg] = (short)tb;
        plan.offa[nseg] = oa; plan.offb[nseg] = ob;
        nseg++;
        ntile += n;
    };
    for (int lo = 1; lo <= 32; lo++) {
        while (pairing && rem[lo] > 0 && lo <= 16) {
            int h = 32 - lo;
            while (h >= lo && !(rem[h] > 0 && (h != lo || rem[lo] >= 2))) h--;
            if (h < lo) break;
            const int n = h == lo ? rem[lo] / 2 : (rem[lo] < rem[h] ? rem[lo] : rem[h]);
            if (h == lo) {
                add(lo, off[lo], lo, off[lo] + n, n);
                off[lo] += 2 * n; rem[lo] -= 2 * n;
            } else {
                add(lo, off[lo], h, off[h], n);
                off[lo] += n; rem[lo] -= n;
                off[h] += n; rem[h] -= n;
            }
        }
        if (rem[lo] > 0) {
            add(lo, off[lo], 0, 0, rem[lo]);
            off[lo] += rem[lo]; rem[lo] = 0;
        }
    }
    plan.first[nseg] = ntile;
    plan.n_seg = nseg;
    g.n_tiles1 = ntile;
    g.n_tiles2 = g.bucket_start[4] - g.bucket_start[2];
    if (ntile > 0) k_tile_fill<<<cdiv(ntile, 256), 256, 0, st>>>(plan, g.atoms_by_t, g.rowptr, g.tile_desc);
    if (g.n_tiles2 > 0)
        k_tile_fill_big<<<cdiv(g.n_tiles2, 256), 256, 0, st>>>(g.atom_order + g.bucket_start[2], g.n_tiles2, g.rowptr,
                                                               g.tile_desc + 2 * (size_t)ntile);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

__global__ void k_csr_fill(const int* __restrict__ perm, const float4* __restrict__ vin,
                           const int* __restrict__ centers, const int* __restrict__ neighbors,
                           const int* __restrict__ shifts, const int* __restrict__ sp,
                           int* __restrict__ ctr, int* __restrict__ nbr, int* __restrict__ shift,
                           int* __restrict__ sp_nbr, float4* __restrict__ geo,
                           float* __restrict__ d0, float* __restrict__ fc, const int* __restrict__ n_kept_dev,
                           float cutoff, float width, int fn, const float* __restrict__ r_atom,
                           float* __restrict__ pc, int* __restrict__ rev) {
    // launched over all input edges: the kept count stays on the device until the single read-back of graph_build
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= *n_kept_dev) return;
    rev[p] = -1;  // "no partner yet" (k_reverse, k_find_pad_src)
    int e = perm[p];
    float4 v = vin[e];
    int j = neighbors[e];
    ctr[p] = centers[e];
    nbr[p] = j;
    shift[3 * p] = shifts[3 * e];
    shift[3 * p + 1] = shifts[3 * e + 1];
    shift[3 * p + 2] = shifts[3 * e + 2];
    sp_nbr[p] = sp[j];
    // structures.py:330: the network sees sqrt(sum v^2 + 1e-15), not |v| + 1e-15
    geo[p] = make_float4(v.x, v.y, v.z, sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + 1e-15f));
    d0[p] = v.w;
    float c = cutoff;
    if (r_atom) {
        c = (r_atom[centers[e]] + r_atom[j]) / 2.0f;
        pc[p] = c;
    }
    fc[p] = cutoff_value(v.w, c, width, fn);
}

// nef.py:88-166 restated as a search in row j (rows hold <= a few dozen edges)
// Four lanes per edge walk row j in strides of four (a row holds ~20 - 40 edges: the one-lane scan was that many dependent
// loads). Only the edge of a pair (p: i -> j, S; q: j -> i, -S) with i < j searches and writes both rev[p] = q and rev[q] = p
// (i == j, an atom's own image: both search); rev arrives filled with -1 (k_csr_fill), and k_find_pad_src turns what is
// still -1 -- an edge whose partner is missing -- into the error count and a self reference.
__global__ void k_reverse(const int* __restrict__ rowptr, const int* __restrict__ ctr,
                          const int* __restrict__ nbr, const int* __restrict__ shift,
                          int* __restrict__ rev, int* __restrict__ scalars) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = t >> 2, sub = t & 3;
    bool live = p < scalars[0];
    int found = -1;
    if (live) {
        const int i = ctr[p], j = nbr[p];
        live = i <= j;
        if (live) {
            const int sa = -shift[3 * p], sb = -shift[3 * p + 1], sc = -shift[3 * p + 2];
            // four candidates per lane in flight (clamped to the row: a repeat of its last entry), then the comparisons:
            // a row of 40 edges is 3 dependent rounds instead of 10
            const int b = rowptr[j], e = rowptr[j + 1];
            for (int q0 = b + sub; q0 < e && found < 0; q0 += 16) {
                int c[4];
#pragma unroll
                for (int u = 0; u < 4; u++) c[u] = nbr[min(q0 + 4 * u, e - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + 4 * u;
                    if (q < e && c[u] == i && shift[3 * q] == sa && shift[3 * q + 1] == sb && shift[3 * q + 2] == sc) found = q;
                }
            }
        }
    }
    found = max(found, __shfl_xor(found, 1));  // at most one lane of the four finds the partner (edges are unique)
    found = max(found, __shfl_xor(found, 2));
    if (!live || sub != 0 || found < 0) return;
    rev[p] = found;
    rev[found] = p;
}
// reverse edges inside the all-edge CSR (ctr0 = the sorted keys of the first pass)
__global__ void k_reverse_all(const int* __restrict__ rowptr0, const int* __restrict__ ctr0,
                              const int* __restrict__ nbr0, const int* __restrict__ shift0, int* __restrict__ rev0,
                              int n) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int i = ctr0[q], j = nbr0[q];
    const int sa = -shift0[3 * q], sb = -shift0[3 * q + 1], sc = -shift0[3 * q + 2];
    int found = -1;
    for (int t = rowptr0[j]; t < rowptr0[j + 1]; t++) {
        if (nbr0[t] == i && shift0[3 * t] == sa && shift0[3 * t + 1] == sb && shift0[3 * t + 2] == sc) {
            found = t;
            break;
        }
    }
    rev0[q] = found;
}

// ---- NEF export (backend.py:328-341) ------------------------------------------------
__global__ void k_export_nodes(const int* __restrict__ sp, int64_t* __restrict__ el_nodes,
                               float* __restrict__ cut_stats, int n, float cutoff,
                               const float* __restrict__ r_atom) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (el_nodes) el_nodes[i] = sp[i];
    if (cut_stats) cut_stats[i] = r_atom ? r_atom[i] : cutoff;
}

__global__ void k_export_nef(const int* __restrict__ rowptr, const int* __restrict__ nbr,
                             const int* __restrict__ rev, const int* __restrict__ sp_nbr,
                             const float4* __restrict__ geo, const float* __restrict__ fc,
                             const int* __restrict__ perm, const int* __restrict__ kidx,
                             int64_t* __restrict__ el_nbr, float* __restrict__ ev,
                             float* __restrict__ ed, uint8_t* __restrict__ mask,
                             int64_t* __restrict__ rni, float* __restrict__ cf, int n_nodes,
                             int m, int pad_src /* CSR position of kept edge 0 */) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n_nodes * m) return;
    int i = (int)(idx / m), k = (int)(idx % m);
    int start = rowptr[i], n = rowptr[i + 1] - start;
    bool real = k < n;
    // nef.py:75-80: nef_indices is zero-initialised, so pads alias kept edge 0
    int p = real ? start + k : pad_src;
    if (el_nbr) el_nbr[idx] = (pad_src < 0 && !real) ? 0 : sp_nbr[p];
    float4 g = (pad_src < 0 && !real) ? make_float4(0, 0, 0, 0) : geo[p];
    if (ev) { ev[3 * idx] = g.x; ev[3 * idx + 1] = g.y; ev[3 * idx + 2] = g.z; }
    if (ed) ed[idx] = g.w;
    if (mask) mask[idx] = real ? 1 : 0;
    if (cf) cf[idx] = real ? fc[p] : 0.0f;
    if (rni) {
        if (real) {
            int q = rev[p];
            int j = nbr[p];
            rni[idx] = (int64_t)j * m + (q - rowptr[j]);
        } else {
            // structures.py:359-362: cumsum(~mask) - 1 in row-major order, closed form
            rni[idx] = (int64_t)i * m - start + (k - n);
        }
    }
}

__global__ void k_export_edges(const int* __restrict__ perm, const int* __restrict__ kidx,
                               const int* __restrict__ rowptr, const int* __restrict__ ctr,
                               const int* __restrict__ nbr, const int* __restrict__ shift,
                               int64_t* __restrict__ centers, int64_t* __restrict__ neighbors,
                               int64_t* __restrict__ slot, int64_t* __restrict__ shifts_out,
                               int n_kept) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_kept) return;
    int k = kidx[perm[p]];  // position among kept edges in the caller's order
    int i = ctr[p];
    if (centers) centers[k] = i;
    if (neighbors) neighbors[k] = nbr[p];
    if (slot) slot[k] = p - rowptr[i];
    if (shifts_out) {
        shifts_out[3 * k] = shift[3 * p];
        shifts_out[3 * k + 1] = shift[3 * p + 1];
        shifts_out[3 * k + 2] = shift[3 * p + 2];
    }
}

// CSR position of kept edge 0 (the edge every NEF pad aliases)
__global__ void k_find_pad_src(const int* __restrict__ perm, const int* __restrict__ kidx,
                               const int* __restrict__ keep, const int* __restrict__ n_kept_dev,
                               int* __restrict__ out, int* __restrict__ rev, int* __restrict__ n_unpaired) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= *n_kept_dev) return;
    int e = perm[p];
    if (keep[e] && kidx[e] == 0) *out = p;
    // an edge without its (j, i, -S) partner (k_reverse left -1) makes graph_build fail (PET_ERR_GRAPH); pointing it at
    // itself keeps every later gather in bounds whatever the caller does with the error
    if (rev[p] < 0) {
        rev[p] = p;
        atomicAdd(n_unpaired, 1);
    }
}

__global__ __launch_bounds__(256) void k_sum_over_atoms(const float* __restrict__ atomic, const int* __restrict__ sys,
                                                        float* __restrict__ out, int n) {
    // One workgroup per system. system_indices is non-decreasing (validated by graph_build), so the atoms of system s
    // are the run [lower_bound(s), lower_bound(s + 1)); strided fp64 partial sums and a fixed-order LDS tree make the
    // result deterministic and independent of the run length (a 10 k-atom box is 40 coalesced loads per lane, not
    // 10 k dependent ones).
    __shared__ double red[256];
    const int s = blockIdx.x;
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (sys[mid] < s) lo = mid + 1; else hi = mid; }
    const int begin = lo;
    hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (sys[mid] < s + 1) lo = mid + 1; else hi = mid; }
    const int end = lo;
    double acc = 0.0;
    for (int k = begin + threadIdx.x; k < end; k += 256) acc += (double)atomic[k];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[s] = (float)red[0];
}

// ----------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------
static int sort_bits(int64_t n_nodes) {
    int bits = 1;
    while ((int64_t(1) << bits) <= n_nodes) bits++;
    return bits;
}

static int carve_graph(Graph& g, void* ws, int64_t n_nodes, int64_t e0, size_t* total) {
    Carver c(ws);
    g.vin = c.take<float4>(e0);
    g.keep = c.take<int>(e0);
    g.kidx = c.take<int>(e0);
    g.sort_keys_in = c.take<int>(e0);
    g.sort_keys_out = c.take<int>(e0);
    g.sort_vals_in = c.take<int>(e0);
    g.perm = c.take<int>(e0);
    g.rowptr = c.take<int>(n_nodes + 1);
    g.ctr = c.take<int>(e0);
    g.nbr = c.take<int>(e0);
    g.shift = c.take<int>(3 * e0);
    g.rev = c.take<int>(e0);
    g.sp = c.take<int>(n_nodes);
    g.sp_nbr = c.take<int>(e0);
    g.geo = c.take<float4>(e0);
    g.d0 = c.take<float>(e0);
    g.fc = c.take<float>(e0);
    g.sys = c.take<int>(n_nodes);
    g.scalars = c.take<int>(128);
    g.atom_order = c.take<int>(n_nodes > 0 ? n_nodes : 1);
    g.atoms_by_t = c.take<int>(n_nodes > 0 ? n_nodes : 1);
    g.tsort_tmp = c.take<int>((size_t)(n_nodes / 256 + 2) * 33);
    g.tile_desc = c.take<int4>(2 * (n_nodes > 0 ? n_nodes : 1));
    g.rowptr0 = c.take<int>(n_nodes + 1);
    g.perm0 = c.take<int>(e0);
    g.nbr0 = c.take<int>(e0);
    g.shift0 = c.take<int>(3 * e0);
    g.rev0 = c.take<int>(e0);
    g.r_atom = c.take<float>(n_nodes);
    g.r_newton = c.take<float>(n_nodes);
    g.inv_dn = c.take<float>(n_nodes);
    g.grid_drdn = c.take<float>(n_nodes * GRID_MAX_PROBES);
    g.pc = c.take<float>(e0);
    g.ad_gc = c.take<float>(e0);
    g.ad_gr = c.take<float>(n_nodes);
    g.ad_dv = c.take<float4>(e0);
    size_t sort_bytes = 0, scan_bytes = 0;
    int* ni = nullptr;
    if (rocprim::radix_sort_pairs(nullptr, sort_bytes, ni, ni, ni, ni, (size_t)e0, 0,
                                  sort_bits(n_nodes)) != hipSuccess)
        return PET_ERR_HIP;
    if (rocprim::exclusive_scan(nullptr, scan_bytes, ni, ni, 0, (size_t)e0, rocprim::plus<int>()) !=
        hipSuccess)
        return PET_ERR_HIP;
    g.sort_tmp_bytes = sort_bytes;
    g.scan_tmp_bytes = scan_bytes;
    g.sort_tmp = c.take<char>(sort_bytes + 256);
    g.scan_tmp = c.take<char>(scan_bytes + 256);
    *total = c.off;
    return PET_OK;
}

int64_t graph_workspace_bytes(int64_t n_nodes, int64_t e0) {
    Graph g;
    size_t total = 0;
    if (carve_graph(g, nullptr, n_nodes, e0 > 0 ? e0 : 1, &total) != PET_OK) return -1;
    return (int64_t)total;
}

// The attention lists of a graph that was built before its model had weights (see graph_build): made on the first tuned
// forward. The bucket / histogram slots of `scalars` are still zero: nothing has written them.
int graph_attention_lists(const Graph& gc, hipStream_t st) {
    Graph& g = const_cast<Graph&>(gc);
    if (g.attn_lists || g.n_nodes <= 0) return PET_OK;
    if (int rcb = bucket_atoms_by_tile_count(g, st)) return rcb;
    int host_scalars[57] = {0};
    if (int rcr = read_back(g.scalars, 57, nullptr, 0, host_scalars, st)) return rcr;
    set_bucket_starts(g, host_scalars + 8);
    if (int rcp = plan_attention_tiles(g, host_scalars + 24, st)) return rcp;
    g.attn_lists = true;
    return PET_OK;
}

// The sorted shortcut (below) needs to know whether the list is ordered by centre before it can skip the sort: asked of the
// device that is a read-back in the middle of the build. A molecular-dynamics driver hands over a list of the same kind
// every step, so the build ASSUMES what the previous build found: "sorted" skips the sort unasked, and the build's one
// read-back at the end carries the word that says whether that was right (if not: k_rowptr has left the graph empty, and
// the build runs again with the sort); "unsorted" sorts unasked, which is right for either kind of list.
enum SortMode { SORT_ASK, SORT_ALWAYS, SORT_ASSUME_SORTED };
static std::atomic<int> g_list_was_sorted{-1};  // the previous build's list (-1: no build yet)

static int graph_build_once(const Model& m, const float* pos, const float* cells, const int* centers,
                            const int* neighbors, const int* shifts, const int* species, const int* sys,
                            int64_t n_nodes, int64_t e0, int64_t n_systems, void* ws, int64_t ws_bytes,
                            Graph& g, hipStream_t st, SortMode mode, bool* wrong_guess) {
    size_t need = 0;
    int rc = carve_graph(g, ws, n_nodes, e0 > 0 ? e0 : 1, &need);
    if (rc != PET_OK) return rc;
    PET_REQUIRE((int64_t)need <= ws_bytes, PET_ERR_ARGUMENT, "graph workspace too small");
    g.n_nodes = n_nodes;
    g.n_edges_in = e0;
    g.n_systems = n_systems;
    const int T = 256;
    PET_HIP_CHECK(hipMemsetAsync(g.scalars, 0, 128 * sizeof(int), st));
    if (n_nodes > 0) {
        k_species_index<<<cdiv(n_nodes, T), T, 0, st>>>(species, m.species_table, m.species_table_len,
                                                        g.sp, (int)n_nodes, g.scalars + 5, sys, g.sys,
                                                        (int)(n_systems > 0 ? n_systems : 1), g.scalars + 7);
    }
    g.adaptive = m.h.num_neighbors_adaptive > 0.f;
    const int* sorted_keys = g.sort_keys_out;
    if (g.adaptive || !g_sorted_shortcut) mode = SORT_ALWAYS;
    if (e0 > 0) {
        k_edge_geometry<<<cdiv(e0, T), T, 0, st>>>(pos, cells, centers, neighbors, shifts, g.sys, g.vin,
                                                   g.keep, g.sort_keys_in, g.sort_vals_in, (int)e0,
                                                   (int)n_nodes, m.h.cutoff, g.adaptive ? 2 : m.h.nl_is_strict,
                                                   g.scalars + 6, g.scalars + 60, mode == SORT_ASSUME_SORTED ? g.kidx : nullptr);
        size_t sb = g.sort_tmp_bytes, cb = g.scan_tmp_bytes;
        if (g.adaptive) {
            // all-edge CSR -> per-atom cutoffs -> pair mask; then the usual kept-edge CSR below
            PET_HIP_CHECK(rocprim::radix_sort_pairs(g.sort_tmp, sb, g.sort_keys_in, g.sort_keys_out,
                                                    g.sort_vals_in, g.perm0, (size_t)e0, 0,
                                                    sort_bits(n_nodes), st));
            k_rowptr<<<cdiv(n_nodes + 1, T), T, 0, st>>>(g.sort_keys_out, (int)e0, g.rowptr0, (int)n_nodes,
                                                         g.scalars + 4);
            k_gather_all<<<cdiv(e0, T), T, 0, st>>>(g.perm0, neighbors, shifts, g.nbr0, g.shift0, (int)e0);
            if (m.h.adaptive_cutoff_method == PET_ADAPTIVE_GRID) {
                // probe grid = torch.arange(0.5, cutoff, width / 4) (adaptive_cutoff.py:266-277)
                const double dp = (double)m.h.cutoff_width_adaptive / 4.0;
                const int K = (int)std::ceil(((double)m.h.cutoff - 0.5) / dp);
                PET_REQUIRE(K >= 2 && K <= GRID_MAX_PROBES, PET_ERR_UNSUPPORTED,
                            "adaptive_cutoff_method = 'grid': " + std::to_string(K) + " probe cutoffs (2 .. " +
                                std::to_string(GRID_MAX_PROBES) + " are built)");
                g.grid_probes = K;
                k_adaptive_grid<<<cdiv(n_nodes, 16), 256, 0, st>>>(g.rowptr0, g.perm0, g.vin, g.r_atom, g.grid_drdn,
                                                                   (int)n_nodes, m.h.cutoff_width_adaptive,
                                                                   m.h.num_neighbors_adaptive, K, 0.5f, (float)dp);
            } else
            k_adaptive_solve<<<cdiv(n_nodes, 16), 256, 0, st>>>(g.rowptr0, g.perm0, g.vin, g.r_atom, g.r_newton,
                                                                g.inv_dn, (int)n_nodes, m.h.cutoff,
                                                                m.h.cutoff_width_adaptive,
                                                                m.h.num_neighbors_adaptive);
            k_reverse_all<<<cdiv(e0, T), T, 0, st>>>(g.rowptr0, g.sort_keys_out, g.nbr0, g.shift0, g.rev0, (int)e0);
            k_adaptive_keep<<<cdiv(e0, T), T, 0, st>>>(centers, neighbors, g.vin, g.r_atom, g.keep,
                                                       g.sort_keys_in, g.sort_vals_in, (int)e0, (int)n_nodes);
            sb = g.sort_tmp_bytes;
        }
        // a list that is already ordered by centre with no edge to drop (k_edge_geometry) needs no sort: one 4-byte
        // read-back (the build ends with one anyway) against three radix passes over the edges
        int unsorted = 1;
        if (mode == SORT_ASK) {
            if (int rcr = read_back(g.scalars + 60, 1, nullptr, 0, &unsorted, st)) return rcr;
        } else if (mode == SORT_ASSUME_SORTED)
            unsorted = 0;
        if (unsorted) {
            PET_HIP_CHECK(rocprim::radix_sort_pairs(g.sort_tmp, sb, g.sort_keys_in, g.sort_keys_out,
                                                    g.sort_vals_in, g.perm, (size_t)e0, 0,
                                                    sort_bits(n_nodes), st));
        } else {
            sorted_keys = g.sort_keys_in;
            g.perm = g.sort_vals_in;  // the identity
        }
        // kidx = position among the kept edges; a list assumed sorted keeps every edge (k_edge_geometry has written the identity)
        if (mode != SORT_ASSUME_SORTED)
            PET_HIP_CHECK(rocprim::exclusive_scan(g.scan_tmp, cb, g.keep, g.kidx, 0, (size_t)e0,
                                                  rocprim::plus<int>(), st));
    }
    g.attn_lists = m.finalized;
    k_rowptr<<<cdiv(n_nodes + 1, T), T, 0, st>>>(sorted_keys, (int)e0, g.rowptr, (int)n_nodes, g.scalars,
                                                 e0 > 0 && mode == SORT_ASSUME_SORTED ? g.scalars + 60 : nullptr, g.scalars + 1,
                                                 g.attn_lists && n_nodes > 0 ? g.scalars + 8 : nullptr);
    if (e0 > 0) {
        // sized by the input edge count; the kernels read the kept count from the device (scalars[0]), so the whole
        // build needs ONE device -> host read-back (below), like the reference's int(torch.max(num_neighbors))
        k_csr_fill<<<cdiv(e0, T), T, 0, st>>>(g.perm, g.vin, centers, neighbors, shifts, g.sp, g.ctr,
                                              g.nbr, g.shift, g.sp_nbr, g.geo, g.d0, g.fc, g.scalars,
                                              m.h.cutoff, m.h.cutoff_width, m.h.cutoff_function,
                                              g.adaptive ? g.r_atom : nullptr, g.pc, g.rev);
        k_reverse<<<cdiv(4 * (int64_t)e0, T), T, 0, st>>>(g.rowptr, g.ctr, g.nbr, g.shift, g.rev, g.scalars);
        k_find_pad_src<<<cdiv(e0, T), T, 0, st>>>(g.perm, g.kidx, g.keep, g.scalars, g.scalars + 3, g.rev, g.scalars + 2);
    }
    // the per-tile-count atom lists and the attention tile plan serve the PET layers only: a model handle without weights
    // (what the SOAP-BPNN path builds its graphs with; a mirror that runs preprocess before its weights are uploaded)
    // skips them here, and the first tuned forward on the graph makes them (graph_attention_lists)
    if (g.attn_lists)
        if (int rcb = bucket_atoms_by_tile_count(g, st, true)) return rcb;
    int host_scalars[61] = {0};
    if (int rcr = read_back(g.scalars, 61, nullptr, 0, host_scalars, st)) return rcr;
    if (e0 > 0 && !g.adaptive && g_sorted_shortcut) {
        g_list_was_sorted.store(host_scalars[60] ? 0 : 1, std::memory_order_relaxed);
        if (mode == SORT_ASSUME_SORTED && host_scalars[60]) {
            *wrong_guess = true;
            return PET_OK;
        }
    }
    g.n_edges = host_scalars[0];
    g.max_nbr = host_scalars[1];
    if (g.attn_lists) {
        set_bucket_starts(g, host_scalars + 8);
        if (int rcp = plan_attention_tiles(g, host_scalars + 24, st)) return rcp;
    }
    PET_REQUIRE(host_scalars[6] == 0, PET_ERR_ARGUMENT,
                std::to_string(host_scalars[6]) + " neighbour-list entries index atoms outside [0, n_nodes)");
    PET_REQUIRE(host_scalars[5] == 0, PET_ERR_ARGUMENT,
                std::to_string(host_scalars[5]) + " atom(s) have an atomic number that is not in the model's atomic_types");
    PET_REQUIRE(host_scalars[7] == 0, PET_ERR_ARGUMENT,
                std::to_string(host_scalars[7]) + " entries of system_indices are outside [0, n_systems) or decreasing");
    // every consumer (the ji gather of the forward pass included) needs the (j, i, -S) partner of every kept edge: the
    // reference's get_corresponding_edges (nef.py:88-166) has the same precondition
    PET_REQUIRE(host_scalars[2] == 0, PET_ERR_GRAPH,
                "neighbour list is not a full list: " + std::to_string(host_scalars[2]) +
                    " kept edges have no reverse edge (j, i, -S)");
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

int graph_build(const Model& m, const float* pos, const float* cells, const int* centers,
                const int* neighbors, const int* shifts, const int* species, const int* sys,
                int64_t n_nodes, int64_t e0, int64_t n_systems, void* ws, int64_t ws_bytes,
                Graph& g, hipStream_t st) {
    const int last = g_list_was_sorted.load(std::memory_order_relaxed);
    const SortMode mode = last < 0 ? SORT_ASK : last ? SORT_ASSUME_SORTED : SORT_ALWAYS;
    bool wrong_guess = false;
    int rc = graph_build_once(m, pos, cells, centers, neighbors, shifts, species, sys, n_nodes, e0, n_systems, ws, ws_bytes,
                              g, st, mode, &wrong_guess);
    if (rc == PET_OK && wrong_guess)
        rc = graph_build_once(m, pos, cells, centers, neighbors, shifts, species, sys, n_nodes, e0, n_systems, ws, ws_bytes,
                              g, st, SORT_ALWAYS, &wrong_guess);
    return rc;
}

int graph_check_reverse(Graph& g, hipStream_t st) {
    int bad = 0;
    if (int rcr = read_back(g.scalars + 2, 1, nullptr, 0, &bad, st)) return rcr;
    PET_REQUIRE(bad == 0, PET_ERR_GRAPH,
                "neighbour list is not a full list: " + std::to_string(bad) +
                    " edges have no reverse edge (j, i, -S)");
    return PET_OK;
}

int graph_export(const Graph& g, float cutoff, int64_t* el_nodes, int64_t* el_nbr, float* ev,
                 float* ed, uint8_t* mask, int64_t* rni, float* cf, float* stats, int64_t* centers,
                 int64_t* neighbors, int64_t* slot, int64_t* shifts, hipStream_t st) {
    const int T = 256;
    if (g.n_nodes > 0)
        k_export_nodes<<<cdiv(g.n_nodes, T), T, 0, st>>>(g.sp, el_nodes, stats, (int)g.n_nodes, cutoff,
                                                         g.adaptive ? g.r_atom : nullptr);
    int64_t cells = g.n_nodes * (int64_t)g.max_nbr;
    if (cells > 0) {
        int pad_src = -1;
        if (int rcr = read_back(g.scalars + 3, 1, nullptr, 0, &pad_src, st)) return rcr;
        k_export_nef<<<cdiv(cells, T), T, 0, st>>>(g.rowptr, g.nbr, g.rev, g.sp_nbr, g.geo, g.fc, g.perm,
                                                   g.kidx, el_nbr, ev, ed, mask, rni, cf, (int)g.n_nodes,
                                                   g.max_nbr, pad_src);
    }
    if (g.n_edges > 0)
        k_export_edges<<<cdiv(g.n_edges, T), T, 0, st>>>(g.perm, g.kidx, g.rowptr, g.ctr, g.nbr, g.shift,
                                                         centers, neighbors, slot, shifts, (int)g.n_edges);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// ----------------------------------------------------------------------------------
// CSR graph FROM the reference's padded batch_data tensors (backend.py:328-341): what calculate_features / predict
// need of a batch_data dictionary they are handed -- whoever built it (this library's preprocess, the reference's,
// or a caller that edited it). Real slots are a prefix of every row (nef.py:63-85), so
//   rowptr = exclusive scan of the row counts of padding_mask, CSR row p = rowptr[i] + slot,
//   reverse_neighbor_index[i][slot] = j * M + slot_j  ->  rev[p] = rowptr[j] + slot_j, nbr[p] = j.
// Only what the feature / head kernels and their adjoints read is filled: rowptr, ctr, nbr, rev, sp, sp_nbr, geo, fc.
// ----------------------------------------------------------------------------------
__global__ void k_mask_counts(const uint8_t* __restrict__ mask, int n, int M, int* __restrict__ counts,
                              int* __restrict__ scalars) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0;
    if (i < n) {
        for (int k = 0; k < M; k++) c += mask[(int64_t)i * M + k] ? 1 : 0;
        // a real slot after a pad would break the prefix layout: counted, reported by the host
        int prefix = 0;
        while (prefix < M && mask[(int64_t)i * M + prefix]) prefix++;
        if (prefix != c) atomicAdd(&scalars[2], 1);
        counts[i] = c;
    }
    int v = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(&scalars[1], v);
}
__global__ void k_from_batch_fill(const int64_t* __restrict__ el_nodes, const int64_t* __restrict__ el_nbr,
                                  const float* __restrict__ ev, const float* __restrict__ ed,
                                  const int64_t* __restrict__ rni, const float* __restrict__ cf,
                                  const int* __restrict__ rowptr, int n, int M, int* __restrict__ ctr,
                                  int* __restrict__ nbr, int* __restrict__ rev, int* __restrict__ sp,
                                  int* __restrict__ sp_nbr, float4* __restrict__ geo, float* __restrict__ fc,
                                  int* __restrict__ scalars) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * M) return;
    const int i = (int)(idx / M), k = (int)(idx % M);
    // any of el_nodes / el_nbr / ev + ed / rni may be NULL: predict only needs the row structure and the cutoff factors
    if (k == 0) sp[i] = el_nodes ? (int)el_nodes[i] : 0;
    const int cnt = rowptr[i + 1] - rowptr[i];
    if (k >= cnt) return;
    const int p = rowptr[i] + k;
    ctr[p] = i;
    if (rni) {
        const int64_t r = rni[idx];
        const int j = (int)(r / M), kj = (int)(r % M);
        const bool ok = r >= 0 && j < n && kj < rowptr[j + 1] - rowptr[j];
        if (!ok) atomicAdd(&scalars[2], 1);
        nbr[p] = ok ? j : i;
        rev[p] = ok ? rowptr[j] + kj : p;
    } else {
        nbr[p] = i;
        rev[p] = p;
    }
    sp_nbr[p] = el_nbr ? (int)el_nbr[idx] : 0;
    geo[p] = ev ? make_float4(ev[3 * idx], ev[3 * idx + 1], ev[3 * idx + 2], ed[idx]) : make_float4(0.f, 0.f, 0.f, 0.f);
    fc[p] = cf[idx];
}

static int carve_from_batch(Graph& g, void* ws, int64_t n_nodes, int64_t M, size_t* total) {
    Carver c(ws);
    const int64_t cap = n_nodes * M > 0 ? n_nodes * M : 1;
    g.rowptr = c.take<int>(n_nodes + 1);
    g.kidx = c.take<int>(n_nodes + 1);  // row counts (scan input)
    g.ctr = c.take<int>(cap);
    g.nbr = c.take<int>(cap);
    g.rev = c.take<int>(cap);
    g.sp = c.take<int>(n_nodes > 0 ? n_nodes : 1);
    g.sp_nbr = c.take<int>(cap);
    g.geo = c.take<float4>(cap);
    g.fc = c.take<float>(cap);
    g.scalars = c.take<int>(128);
    g.atom_order = c.take<int>(n_nodes > 0 ? n_nodes : 1);
    g.atoms_by_t = c.take<int>(n_nodes > 0 ? n_nodes : 1);
    g.tsort_tmp = c.take<int>((size_t)(n_nodes / 256 + 2) * 33);
    g.tile_desc = c.take<int4>(2 * (n_nodes > 0 ? n_nodes : 1));
    size_t scan_bytes = 0;
    int* ni = nullptr;
    if (rocprim::exclusive_scan(nullptr, scan_bytes, ni, ni, 0, (size_t)(n_nodes + 1), rocprim::plus<int>()) != hipSuccess)
        return PET_ERR_HIP;
    g.scan_tmp_bytes = scan_bytes;
    g.scan_tmp = c.take<char>(scan_bytes + 256);
    *total = c.off;
    return PET_OK;
}

int64_t graph_from_batch_workspace_bytes(int64_t n_nodes, int64_t max_nbr) {
    Graph g;
    size_t total = 0;
    if (carve_from_batch(g, nullptr, n_nodes, max_nbr, &total) != PET_OK) return -1;
    return (int64_t)total;
}

int graph_from_batch(const int64_t* el_nodes, const int64_t* el_nbr, const float* ev, const float* ed, const uint8_t* mask,
                     const int64_t* rni, const float* cf, int64_t n_nodes, int64_t M, void* ws, int64_t ws_bytes, Graph& g,
                     hipStream_t st) {
    size_t need = 0;
    int rc = carve_from_batch(g, ws, n_nodes, M, &need);
    if (rc != PET_OK) return rc;
    PET_REQUIRE((int64_t)need <= ws_bytes, PET_ERR_ARGUMENT, "graph-from-batch workspace too small");
    g.n_nodes = n_nodes;
    g.n_edges_in = n_nodes * M;
    g.n_systems = 0;
    g.adaptive = false;
    const int T = 256;
    PET_HIP_CHECK(hipMemsetAsync(g.scalars, 0, 128 * sizeof(int), st));
    PET_HIP_CHECK(hipMemsetAsync(g.kidx, 0, (n_nodes + 1) * sizeof(int), st));
    if (n_nodes > 0 && M > 0)
        k_mask_counts<<<cdiv(n_nodes, T), T, 0, st>>>(mask, (int)n_nodes, (int)M, g.kidx, g.scalars);
    size_t cb = g.scan_tmp_bytes;
    PET_HIP_CHECK(rocprim::exclusive_scan(g.scan_tmp, cb, g.kidx, g.rowptr, 0, (size_t)(n_nodes + 1), rocprim::plus<int>(), st));
    if (n_nodes > 0 && M > 0)
        k_from_batch_fill<<<cdiv(n_nodes * M, T), T, 0, st>>>(el_nodes, el_nbr, ev, ed, rni, cf, g.rowptr, (int)n_nodes, (int)M,
                                                              g.ctr, g.nbr, g.rev, g.sp, g.sp_nbr, g.geo, g.fc, g.scalars);
    else if (n_nodes > 0)
        k_from_batch_fill<<<cdiv(n_nodes, T), T, 0, st>>>(el_nodes, el_nbr, ev, ed, rni, cf, g.rowptr, (int)n_nodes, 1, g.ctr,
                                                          g.nbr, g.rev, g.sp, g.sp_nbr, g.geo, g.fc, g.scalars);
    g.attn_lists = true;
    if (int rcb = bucket_atoms_by_tile_count(g, st)) return rcb;
    int host_scalars[58] = {0};
    if (int rcr = read_back(g.scalars, 57, g.rowptr + n_nodes, 1, host_scalars, st)) return rcr;
    g.n_edges = host_scalars[57];
    g.max_nbr = host_scalars[1];
    set_bucket_starts(g, host_scalars + 8);
    if (int rcp = plan_attention_tiles(g, host_scalars + 24, st)) return rcp;
    PET_REQUIRE(host_scalars[2] == 0, PET_ERR_GRAPH,
                "batch_data is not a NEF batch: " + std::to_string(host_scalars[2]) +
                    " rows with a real slot behind a pad, or reverse_neighbor_index entries that do not point at a real slot");
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

int sum_over_atoms(const Graph& g, const float* atomic, float* out, hipStream_t st) {
    if (g.n_systems > 0)  // systems without atoms get 0
        k_sum_over_atoms<<<(int)g.n_systems, 256, 0, st>>>(atomic, g.sys, out, (int)g.n_nodes);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

}  // namespace pet
