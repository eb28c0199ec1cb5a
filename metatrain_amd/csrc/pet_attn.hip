// Per-atom attention, forward and adjoint, "preload" variants for up to 63 neighbours (NT <= 4).
//
// Same math and same operand-layout trick as k_attn_fwd / k_attn_bwd in pet_fwd.hip / pet_bwd.hip
// (one wave per (atom, head), transposed score tiles on v_mfma_f32_16x16x4_f32, no LDS), but every
// global operand -- Q, K, V, dO fragments and the scalar re-reads used as A operands -- is loaded
// into registers up front in ONE burst, so the wave pays one memory round trip instead of a chain
// of dependent ones (the first version was latency-bound: MFMA busy 19 %, 58 % of wave time in
// s_waitcnt). Reference: pet/modules/transformer.py:86-152, 565-589.
#include "common.h"
#include "model.h"

namespace pet {

__device__ __forceinline__ int64_t tok_row(int t, int T, int64_t E, int atom, int start) {
    return (t == 0 || t >= T) ? E + atom : (int64_t)start + t - 1;
}
__device__ __forceinline__ float key_bias(int key, int T, const float* __restrict__ fc, int start) {
    if (key >= T) return -INFINITY;
    if (key == 0) return 0.f;
    return logf(fmaxf(fc[start + key - 1], 1e-15f));  // transformer.py:109-110
}
// log2-domain softmax: scores carry a factor log2(e) (folded into the query scale and the key bias), so the
// exponential is the bare v_exp_f32 (exp2) instead of the 13-instruction expf expansion; the kernels below are
// VALU-issue bound, not MFMA bound, so instruction count is what matters.
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float key_bias2(int key, int T, const float* __restrict__ fc, int start) {
    if (key >= T) return -INFINITY;
    if (key == 0) return 0.f;
    return __builtin_amdgcn_logf(fmaxf(fc[start + key - 1], 1e-15f));  // v_log_f32 = log2
}
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Reductions across the four 16-lane groups (lanes l, l^16, l^32, l^48) with the gfx950 row swaps
// (VALU speed; the first version used ds_bpermute and spent most of its time in lgkmcnt waits).
__device__ __forceinline__ float g4_sum(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float g4_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// sum over the 16 lanes of a row with DPP (quad_perm, quad_perm, row_half_mirror, row_mirror)
__device__ __forceinline__ float row16_sum(float v) {
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xF, 0xF, true));
    return v;
}

// Key-bias gradient of one 16-key tile. After the row sums every lane of a 16-lane row holds the four key values of
// its g4 group; lanes c16 < 4 store one key each, so a wave writes the tile's 16 keys as one 64-byte run of the
// attention layer's head-major slice [NHEAD, E]. Plain stores: each (layer, head, edge) has exactly one writer,
// and k_dfc_attn sums the slices.
__device__ __forceinline__ void store_key_bias(float* __restrict__ dbh, int64_t E, int head, int start, int T,
                                               int kt, int g4, int c16, float v0, float v1, float v2, float v3) {
    const float v = c16 == 0 ? v0 : c16 == 1 ? v1 : c16 == 2 ? v2 : v3;
    const int key = 16 * kt + 4 * g4 + c16;
    if (c16 < 4 && key >= 1 && key < T) dbh[(int64_t)head * E + start + key - 1] = v;
}

template <int NT>
__global__ __launch_bounds__(256) void k_attn_fwd_p(const float* __restrict__ QKV, const int* __restrict__ rowptr,
                                                     const float* __restrict__ fc, float* __restrict__ AO,
                                                     int64_t E, int N, float scale, const int* __restrict__ atoms,
                                                     int n_list) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, and the compiler knows it
    const int li = gw / NHEAD, head = gw % NHEAD;
    if (li >= n_list) return;
    const int atom = atoms[li];  // bucketed launch: the graph's list of the atoms with this tile count (graph.hip)
    const int start = rowptr[atom];
    const int T = rowptr[atom + 1] - start + 1;
    const int nt = (T + 15) >> 4;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int qo = HD * head, ko = D + HD * head, vo = 2 * D + HD * head;
    float4 kf[NT], qf[NT];
    float vs[NT][4], bias[NT][4];
    int64_t qrow[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (t < nt) {
            const int64_t rc = tok_row(16 * t + c16, T, E, atom, start);
            qrow[t] = rc;
            kf[t] = *reinterpret_cast<const float4*>(QKV + rc * (3 * D) + ko + 4 * g4);
            qf[t] = *reinterpret_cast<const float4*>(QKV + rc * (3 * D) + qo + 4 * g4);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = 16 * t + 4 * g4 + r;
                vs[t][r] = QKV[tok_row(key, T, E, atom, start) * (3 * D) + vo + c16];
                bias[t][r] = key_bias2(key, T, fc, start);
            }
        }
    }
#pragma unroll
    for (int qt = 0; qt < NT; qt++) {
        if (qt < nt) {
            const float s2 = scale * LOG2E;
            const float4 q = make_float4(qf[qt].x * s2, qf[qt].y * s2, qf[qt].z * s2, qf[qt].w * s2);
            f32x4 s[NT];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; kt++) {
                if (kt < nt) {
                    f32x4 a = {bias[kt][0], bias[kt][1], bias[kt][2], bias[kt][3]};  // the bias rides in the accumulator
                    a = MFMA16(kf[kt].x, q.x, a); a = MFMA16(kf[kt].y, q.y, a);
                    a = MFMA16(kf[kt].z, q.z, a); a = MFMA16(kf[kt].w, q.w, a);
#pragma unroll
                    for (int r = 0; r < 4; r++) mx = fmaxf(mx, a[r]);
                    s[kt] = a;
                }
            }
            mx = g4_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) { const float p = __builtin_amdgcn_exp2f(s[kt][r] - mx); s[kt][r] = p; sum += p; }
            sum = g4_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) o = MFMA16(vs[kt][r], s[kt][r], o);
            if (16 * qt + c16 < T)
                *reinterpret_cast<float4*>(AO + qrow[qt] * D + qo + 4 * g4) =
                    make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Second-order attention for the training pass (so.hip), same tile conventions as k_attn_*_p.
//   tangent:  o' = sum_j [ P_ij v'_j + P_ij (s'_ij - a_i) v_j ],  s' = scale (q'.k + q.k') + b',  a_i = <P_i, s'_i>
//   joint reverse for (lambda_o, nu_o) -> (lambda, nu) of q, k, v:
//     lambda_P = lambda_o.v,  nu_P = nu_o.v + lambda_o.v',  delta = <P, lambda_P>,  c = <P, lambda_P s'>,  e = <P, nu_P>
//     lambda_s = P (lambda_P - delta),   nu_s = P (nu_P - e) + P [ (lambda_P - delta)(s' - a) - (c - a delta) ]
//     lambda_q = scale lambda_s K          nu_q = scale (nu_s K + lambda_s K')
//     lambda_k = scale lambda_s^T Q        nu_k = scale (nu_s^T Q + lambda_s^T Q')
//     lambda_v = P^T lambda_o              nu_v = P^T nu_o + P'^T lambda_o,   P' = P (s' - a)
// (identities checked against torch.autograd in fp64)
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_attn_jvp_p(const float* __restrict__ QKV, const float* __restrict__ QKVd,
                                                     const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                     const float* __restrict__ Tkb, float* __restrict__ AOd,
                                                     int64_t E, int N, float scale, const int* __restrict__ atoms,
                                                     int n_list) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, and the compiler knows it
    const int li = gw / NHEAD, head = gw % NHEAD;
    if (li >= n_list) return;
    const int atom = atoms[li];
    const int start = rowptr[atom];
    const int T = rowptr[atom + 1] - start + 1;
    const int nt = (T + 15) >> 4;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int qo = HD * head, ko = D + HD * head, vo = 2 * D + HD * head;
    float4 kf[NT], kdf[NT], qf[NT], qdf[NT];
    float vs[NT][4], vds[NT][4], bias[NT][4], biasd[NT][4];
    int64_t qrow[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (t < nt) {
            const int64_t rc = tok_row(16 * t + c16, T, E, atom, start);
            qrow[t] = rc;
            kf[t] = *reinterpret_cast<const float4*>(QKV + rc * (3 * D) + ko + 4 * g4);
            kdf[t] = *reinterpret_cast<const float4*>(QKVd + rc * (3 * D) + ko + 4 * g4);
            const float4 q = *reinterpret_cast<const float4*>(QKV + rc * (3 * D) + qo + 4 * g4);
            const float4 qd = *reinterpret_cast<const float4*>(QKVd + rc * (3 * D) + qo + 4 * g4);
            qf[t] = make_float4(q.x * scale, q.y * scale, q.z * scale, q.w * scale);
            qdf[t] = make_float4(qd.x * scale, qd.y * scale, qd.z * scale, qd.w * scale);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = 16 * t + 4 * g4 + r;
                const int64_t rr = tok_row(key, T, E, atom, start);
                vs[t][r] = QKV[rr * (3 * D) + vo + c16];
                vds[t][r] = QKVd[rr * (3 * D) + vo + c16];
                bias[t][r] = key_bias(key, T, fc, start);
                biasd[t][r] = (key >= 1 && key < T) ? Tkb[start + key - 1] : 0.f;
            }
        }
    }
#pragma unroll
    for (int qt = 0; qt < NT; qt++) {
        if (qt < nt) {
            f32x4 s[NT], sd[NT];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; kt++) {
                if (kt < nt) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
                    a = MFMA16(kf[kt].x, qf[qt].x, a); a = MFMA16(kf[kt].y, qf[qt].y, a);
                    a = MFMA16(kf[kt].z, qf[qt].z, a); a = MFMA16(kf[kt].w, qf[qt].w, a);
                    b = MFMA16(kf[kt].x, qdf[qt].x, b); b = MFMA16(kf[kt].y, qdf[qt].y, b);
                    b = MFMA16(kf[kt].z, qdf[qt].z, b); b = MFMA16(kf[kt].w, qdf[qt].w, b);
                    b = MFMA16(kdf[kt].x, qf[qt].x, b); b = MFMA16(kdf[kt].y, qf[qt].y, b);
                    b = MFMA16(kdf[kt].z, qf[qt].z, b); b = MFMA16(kdf[kt].w, qf[qt].w, b);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        a[r] += bias[kt][r];
                        b[r] += biasd[kt][r];
                        mx = fmaxf(mx, a[r]);
                    }
                    s[kt] = a;
                    sd[kt] = b;
                }
            }
            mx = g4_max(mx);
            float sum = 0.f, an = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float p = __builtin_amdgcn_exp2f(LOG2E * (s[kt][r] - mx));
                        s[kt][r] = p;
                        sum += p;
                        an += p * sd[kt][r];
                    }
            sum = g4_sum(sum);
            an = g4_sum(an);
            const float inv = 1.0f / sum, av = an * inv;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float p = s[kt][r];               // un-normalised; 1/sum applied at the end
                        o = MFMA16(vds[kt][r], p, o);
                        o = MFMA16(vs[kt][r], p * (sd[kt][r] - av), o);
                    }
            if (16 * qt + c16 < T)
                *reinterpret_cast<float4*>(AOd + qrow[qt] * D + qo + 4 * g4) =
                    make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void k_attn_rev_p(const float* __restrict__ QKV, const float* __restrict__ QKVd,
                                                     const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                     const float* __restrict__ Tkb, const float* __restrict__ LO,
                                                     const float* __restrict__ NO, float* __restrict__ lQKV,
                                                     float* __restrict__ nQKV, int64_t E, int N, float scale,
                                                     const int* __restrict__ atoms,
                                                     int n_list) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, and the compiler knows it
    const int li = gw / NHEAD, head = gw % NHEAD;
    if (li >= n_list) return;
    const int atom = atoms[li];
    const int start = rowptr[atom];
    const int T = rowptr[atom + 1] - start + 1;
    const int nt = (T + 15) >> 4;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int qo = HD * head, ko = D + HD * head, vo = 2 * D + HD * head;
    // fragments: token on the 16-lane axis (rows 16t + c16), 4 features per lane group
    float4 kf[NT], vf[NT], qf[NT], lof[NT], kdf[NT], vdf[NT], qdf[NT], nof[NT];
    // scalars: token on the (group, register) axis (rows 16t + 4 g4 + r), feature c16
    float ks[NT][4], qs[NT][4], los[NT][4], kds[NT][4], qds[NT][4], nos[NT][4];
    float bias_r[NT][4], biasd_r[NT][4], bias_c[NT], biasd_c[NT];
    int64_t rowc[NT];
    __shared__ __attribute__((aligned(16))) float stat_all[4][5][NT * 16];
    float (*stat)[NT * 16] = stat_all[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (t < nt) {
            const int tc = 16 * t + c16;
            const int64_t rc = tok_row(tc, T, E, atom, start);
            rowc[t] = rc;
            const bool real = tc < T;
            kf[t] = *reinterpret_cast<const float4*>(QKV + rc * (3 * D) + ko + 4 * g4);
            vf[t] = *reinterpret_cast<const float4*>(QKV + rc * (3 * D) + vo + 4 * g4);
            kdf[t] = *reinterpret_cast<const float4*>(QKVd + rc * (3 * D) + ko + 4 * g4);
            vdf[t] = *reinterpret_cast<const float4*>(QKVd + rc * (3 * D) + vo + 4 * g4);
            const float4 q = *reinterpret_cast<const float4*>(QKV + rc * (3 * D) + qo + 4 * g4);
            const float4 qd = *reinterpret_cast<const float4*>(QKVd + rc * (3 * D) + qo + 4 * g4);
            qf[t] = make_float4(q.x * scale, q.y * scale, q.z * scale, q.w * scale);
            qdf[t] = make_float4(qd.x * scale, qd.y * scale, qd.z * scale, qd.w * scale);
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            lof[t] = real ? *reinterpret_cast<const float4*>(LO + rc * D + qo + 4 * g4) : z4;
            nof[t] = real ? *reinterpret_cast<const float4*>(NO + rc * D + qo + 4 * g4) : z4;
            bias_c[t] = key_bias(tc, T, fc, start);
            biasd_c[t] = (tc >= 1 && real) ? Tkb[start + tc - 1] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int tr = 16 * t + 4 * g4 + r;
                const int64_t rr = tok_row(tr, T, E, atom, start);
                const bool rl = tr < T;
                ks[t][r] = QKV[rr * (3 * D) + ko + c16];
                qs[t][r] = QKV[rr * (3 * D) + qo + c16];
                kds[t][r] = QKVd[rr * (3 * D) + ko + c16];
                qds[t][r] = QKVd[rr * (3 * D) + qo + c16];
                los[t][r] = rl ? LO[rr * D + qo + c16] : 0.f;
                nos[t][r] = rl ? NO[rr * D + qo + c16] : 0.f;
                bias_r[t][r] = key_bias(tr, T, fc, start);
                biasd_r[t][r] = (tr >= 1 && rl) ? Tkb[start + tr - 1] : 0.f;
            }
        }
    }
    // ---- pass A: per query tile, transposed tiles (keys x queries): lambda_q, nu_q and the row statistics
#pragma unroll
    for (int qt = 0; qt < NT; qt++) {
        if (qt < nt) {
            f32x4 s[NT], lp[NT], sd[NT], np[NT];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; kt++) {
                if (kt < nt) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f},
                          d = {0.f, 0.f, 0.f, 0.f};
                    a = MFMA16(kf[kt].x, qf[qt].x, a); a = MFMA16(kf[kt].y, qf[qt].y, a);
                    a = MFMA16(kf[kt].z, qf[qt].z, a); a = MFMA16(kf[kt].w, qf[qt].w, a);
                    b = MFMA16(vf[kt].x, lof[qt].x, b); b = MFMA16(vf[kt].y, lof[qt].y, b);
                    b = MFMA16(vf[kt].z, lof[qt].z, b); b = MFMA16(vf[kt].w, lof[qt].w, b);
                    c = MFMA16(kf[kt].x, qdf[qt].x, c); c = MFMA16(kf[kt].y, qdf[qt].y, c);
                    c = MFMA16(kf[kt].z, qdf[qt].z, c); c = MFMA16(kf[kt].w, qdf[qt].w, c);
                    c = MFMA16(kdf[kt].x, qf[qt].x, c); c = MFMA16(kdf[kt].y, qf[qt].y, c);
                    c = MFMA16(kdf[kt].z, qf[qt].z, c); c = MFMA16(kdf[kt].w, qf[qt].w, c);
                    d = MFMA16(vf[kt].x, nof[qt].x, d); d = MFMA16(vf[kt].y, nof[qt].y, d);
                    d = MFMA16(vf[kt].z, nof[qt].z, d); d = MFMA16(vf[kt].w, nof[qt].w, d);
                    d = MFMA16(vdf[kt].x, lof[qt].x, d); d = MFMA16(vdf[kt].y, lof[qt].y, d);
                    d = MFMA16(vdf[kt].z, lof[qt].z, d); d = MFMA16(vdf[kt].w, lof[qt].w, d);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        a[r] += bias_r[kt][r];
                        c[r] += biasd_r[kt][r];
                        mx = fmaxf(mx, a[r]);
                    }
                    s[kt] = a; lp[kt] = b; sd[kt] = c; np[kt] = d;
                }
            }
            mx = g4_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) { const float p = __builtin_amdgcn_exp2f(LOG2E * (s[kt][r] - mx)); s[kt][r] = p; sum += p; }
            sum = g4_sum(sum);
            const float inv = 1.0f / sum;
            float de = 0.f, av = 0.f, cv = 0.f, ev = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float p = s[kt][r] * inv;
                        s[kt][r] = p;
                        de += p * lp[kt][r];
                        av += p * sd[kt][r];
                        cv += p * lp[kt][r] * sd[kt][r];
                        ev += p * np[kt][r];
                    }
            de = g4_sum(de); av = g4_sum(av); cv = g4_sum(cv); ev = g4_sum(ev);
            if (g4 == 0) {
                stat[0][16 * qt + c16] = mx + logf(sum);
                stat[1][16 * qt + c16] = de;
                stat[2][16 * qt + c16] = av;
                stat[3][16 * qt + c16] = cv - av * de;
                stat[4][16 * qt + c16] = ev;
            }
            f32x4 lq = {0.f, 0.f, 0.f, 0.f}, nq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float p = s[kt][r];
                        const float ls = p * (lp[kt][r] - de);
                        const float ns = p * (np[kt][r] - ev) + p * ((lp[kt][r] - de) * (sd[kt][r] - av) - (cv - av * de));
                        lq = MFMA16(ks[kt][r], ls, lq);
                        nq = MFMA16(ks[kt][r], ns, nq);
                        nq = MFMA16(kds[kt][r], ls, nq);
                    }
            if (16 * qt + c16 < T) {
                *reinterpret_cast<float4*>(lQKV + rowc[qt] * (3 * D) + qo + 4 * g4) =
                    make_float4(lq[0] * scale, lq[1] * scale, lq[2] * scale, lq[3] * scale);
                *reinterpret_cast<float4*>(nQKV + rowc[qt] * (3 * D) + qo + 4 * g4) =
                    make_float4(nq[0] * scale, nq[1] * scale, nq[2] * scale, nq[3] * scale);
            }
        }
    }
    // ---- pass B: per key tile, plain tiles (queries x keys): lambda / nu of k and v
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        if (kt < nt) {
            f32x4 lk = {0.f, 0.f, 0.f, 0.f}, nk = {0.f, 0.f, 0.f, 0.f}, lv = {0.f, 0.f, 0.f, 0.f},
                  nv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qt = 0; qt < NT; qt++) {
                if (qt < nt) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f},
                          d = {0.f, 0.f, 0.f, 0.f};
                    a = MFMA16(qf[qt].x, kf[kt].x, a); a = MFMA16(qf[qt].y, kf[kt].y, a);
                    a = MFMA16(qf[qt].z, kf[kt].z, a); a = MFMA16(qf[qt].w, kf[kt].w, a);
                    b = MFMA16(lof[qt].x, vf[kt].x, b); b = MFMA16(lof[qt].y, vf[kt].y, b);
                    b = MFMA16(lof[qt].z, vf[kt].z, b); b = MFMA16(lof[qt].w, vf[kt].w, b);
                    c = MFMA16(qdf[qt].x, kf[kt].x, c); c = MFMA16(qdf[qt].y, kf[kt].y, c);
                    c = MFMA16(qdf[qt].z, kf[kt].z, c); c = MFMA16(qdf[qt].w, kf[kt].w, c);
                    c = MFMA16(qf[qt].x, kdf[kt].x, c); c = MFMA16(qf[qt].y, kdf[kt].y, c);
                    c = MFMA16(qf[qt].z, kdf[kt].z, c); c = MFMA16(qf[qt].w, kdf[kt].w, c);
                    d = MFMA16(nof[qt].x, vf[kt].x, d); d = MFMA16(nof[qt].y, vf[kt].y, d);
                    d = MFMA16(nof[qt].z, vf[kt].z, d); d = MFMA16(nof[qt].w, vf[kt].w, d);
                    d = MFMA16(lof[qt].x, vdf[kt].x, d); d = MFMA16(lof[qt].y, vdf[kt].y, d);
                    d = MFMA16(lof[qt].z, vdf[kt].z, d); d = MFMA16(lof[qt].w, vdf[kt].w, d);
                    const float4 s0 = *reinterpret_cast<const float4*>(&stat[0][16 * qt + 4 * g4]);
                    const float4 s1 = *reinterpret_cast<const float4*>(&stat[1][16 * qt + 4 * g4]);
                    const float4 s2 = *reinterpret_cast<const float4*>(&stat[2][16 * qt + 4 * g4]);
                    const float4 s3 = *reinterpret_cast<const float4*>(&stat[3][16 * qt + 4 * g4]);
                    const float4 s4 = *reinterpret_cast<const float4*>(&stat[4][16 * qt + 4 * g4]);
                    const float lse[4] = {s0.x, s0.y, s0.z, s0.w}, der[4] = {s1.x, s1.y, s1.z, s1.w},
                                avr[4] = {s2.x, s2.y, s2.z, s2.w}, ccr[4] = {s3.x, s3.y, s3.z, s3.w},
                                evr[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int qq = 16 * qt + 4 * g4 + r;
                        float p = __builtin_amdgcn_exp2f(LOG2E * (a[r] + bias_c[kt] - lse[r]));
                        if (qq >= T) p = 0.f;
                        const float sdv = c[r] + biasd_c[kt];
                        const float ls = p * (b[r] - der[r]);
                        const float pd = p * (sdv - avr[r]);
                        const float ns = p * (d[r] - evr[r]) + p * ((b[r] - der[r]) * (sdv - avr[r]) - ccr[r]);
                        lk = MFMA16(qs[qt][r], ls, lk);
                        nk = MFMA16(qs[qt][r], ns, nk);
                        nk = MFMA16(qds[qt][r], ls, nk);
                        lv = MFMA16(los[qt][r], p, lv);
                        nv = MFMA16(nos[qt][r], p, nv);
                        nv = MFMA16(los[qt][r], pd, nv);
                    }
                }
            }
            if (16 * kt + c16 < T) {
                float* lrow = lQKV + rowc[kt] * (3 * D);
                float* nrow = nQKV + rowc[kt] * (3 * D);
                *reinterpret_cast<float4*>(lrow + ko + 4 * g4) = make_float4(lk[0] * scale, lk[1] * scale, lk[2] * scale, lk[3] * scale);
                *reinterpret_cast<float4*>(nrow + ko + 4 * g4) = make_float4(nk[0] * scale, nk[1] * scale, nk[2] * scale, nk[3] * scale);
                *reinterpret_cast<float4*>(lrow + vo + 4 * g4) = make_float4(lv[0], lv[1], lv[2], lv[3]);
                *reinterpret_cast<float4*>(nrow + vo + 4 * g4) = make_float4(nv[0], nv[1], nv[2], nv[3]);
            }
        }
    }
}

// Bucketed launches: the graph build lists the atoms by tile count (Graph::atom_order, bucket_start), so every
// instantiation is launched over exactly its own atoms (no early-exit workgroups).
static inline int bucket_count(const Graph& g, int K) { return g.bucket_start[K] - g.bucket_start[K - 1]; }
static inline const int* bucket_atoms(const Graph& g, int K) { return g.atom_order + g.bucket_start[K - 1]; }

bool attn_jvp_mfma(int nt, const float* QKV, const float* QKVd, const Graph& g, const float* Tkb, float* AOd,
                   float scale, hipStream_t st) {
    if (nt > 4) return false;
    const int N = (int)g.n_nodes;
#define PET_ATTN_JVP(K)                                                                                            \
    if (nt >= K && bucket_count(g, K) > 0)                                                                         \
        k_attn_jvp_p<K><<<cdiv((int64_t)bucket_count(g, K) * NHEAD, 4), 256, 0, st>>>(                             \
            QKV, QKVd, g.rowptr, g.fc, Tkb, AOd, g.n_edges, N, scale, bucket_atoms(g, K), bucket_count(g, K));
    PET_ATTN_JVP(1) PET_ATTN_JVP(2) PET_ATTN_JVP(3) PET_ATTN_JVP(4)
#undef PET_ATTN_JVP
    return true;
}
bool attn_rev_mfma(int nt, const float* QKV, const float* QKVd, const Graph& g, const float* Tkb, const float* LO,
                   const float* NO, float* lQKV, float* nQKV, float scale, hipStream_t st) {
    if (nt > 3) return false;  // NT = 4 exceeds the register file; the VALU kernel in so.hip serves those batches
    const int N = (int)g.n_nodes;
#define PET_ATTN_REV(K)                                                                                            \
    if (nt >= K && bucket_count(g, K) > 0)                                                                         \
        k_attn_rev_p<K><<<cdiv((int64_t)bucket_count(g, K) * NHEAD, 4), 256, 0, st>>>(                             \
            QKV, QKVd, g.rowptr, g.fc, Tkb, LO, NO, lQKV, nQKV, g.n_edges, N, scale, bucket_atoms(g, K),           \
            bucket_count(g, K));
    PET_ATTN_REV(1) PET_ATTN_REV(2) PET_ATTN_REV(3)
#undef PET_ATTN_REV
    return true;
}

// ---------------------------------------------------------------------------------------------
// LDS-staged adjoint: ONE workgroup (8 waves = 8 heads) per atom. The atom's token rows (QKV and dO) are fetched
// once as full coalesced rows into LDS, every wave builds its fragments from LDS, results overwrite the head's own
// Q / K / V slots and leave as full rows again (64 B slices per (row, head) -> 1.5 KB rows). A forward of the same
// form was slower than k_attn_fwd_p (1.2 against 1.0 ms per launch) and was removed.
// ---------------------------------------------------------------------------------------------
constexpr int LDB = 4 * D + 4;   // adjoint staging row: q | k | v | dO

// Adjoint, single pass over the query tiles. The score tile is formed once, transposed (keys x queries), which is
// the layout the softmax reductions and dQ want; P and dS are then turned into the (queries x keys) operand
// layout dK / dV need through a wave-private 16x16 LDS scratch (4 ds_write_b32 + 1 ds_read_b128 per matrix)
// instead of recomputing scores and exponentials in a second pass: 20 MFMA and 4 exp2 per tile pair instead of
// 28 and 8. The scratch lives in columns of the staged rows that the wave has already consumed into registers
// (P in its dO columns, dS in its K columns of rows 0..15; row pitch 516 floats = 4 mod 64 banks, so both the
// scalar writes and the float4 reads are conflict-free); dK lands in the K columns only after the last tile.
template <int NT>
__global__ __launch_bounds__(512) void k_attn_bwd_l(const float* __restrict__ QKV, const float* __restrict__ dAO,
                                                     const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                     float* __restrict__ dQKV, float* __restrict__ dbias_h,
                                                     int64_t E, int N, float scale, const int* __restrict__ atoms) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int atom = atoms[blockIdx.x];  // one workgroup per atom of this tile count's list
    const int start = rowptr[atom];
    const int T = rowptr[atom + 1] - start + 1;
    const int nt = (T + 15) >> 4;
    const int TP = 16 * nt;
    for (int idx = threadIdx.x; idx < TP * D; idx += 512) {  // D float4 per row: 96 of QKV + 32 of dO
        const int t = idx / D, c = idx % D;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) {
            const int64_t row = tok_row(t, T, E, atom, start);
            v = c < 96 ? *reinterpret_cast<const float4*>(QKV + row * (3 * D) + 4 * c)
                       : *reinterpret_cast<const float4*>(dAO + row * D + 4 * (c - 96));
        }
        *reinterpret_cast<float4*>(sm + t * LDB + 4 * c) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;
    const int qo = HD * head, ko = D + HD * head, vo = 2 * D + HD * head, doo = 3 * D + HD * head;
    const float s2 = scale * LOG2E;
    float4 kf[NT], vf[NT], qf[NT], dof[NT];
    float ks[NT][4], qs[NT][4], dos[NT][4];
    float bias_r[NT][4], db[NT][4];
    f32x4 dk[NT], dv[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if (t < nt) {
            const int tc = 16 * t + c16;
            const float* row = sm + tc * LDB;
            kf[t] = *reinterpret_cast<const float4*>(row + ko + 4 * g4);
            vf[t] = *reinterpret_cast<const float4*>(row + vo + 4 * g4);
            const float4 q = *reinterpret_cast<const float4*>(row + qo + 4 * g4);
            qf[t] = make_float4(q.x * s2, q.y * s2, q.z * s2, q.w * s2);
            dof[t] = *reinterpret_cast<const float4*>(row + doo + 4 * g4);  // zero rows beyond T
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int tr = 16 * t + 4 * g4 + r;
                const float* rr = sm + tr * LDB;
                ks[t][r] = rr[ko + c16];
                qs[t][r] = rr[qo + c16];
                dos[t][r] = rr[doo + c16];
                bias_r[t][r] = key_bias2(tr, T, fc, start);
                db[t][r] = 0.f;
                dk[t][r] = 0.f;
                dv[t][r] = 0.f;
            }
        }
    }
    float* sc_p = sm + doo;  // [16 keys][16 queries] scratch, row pitch LDB
    float* sc_s = sm + ko;
#pragma unroll
    for (int qt = 0; qt < NT; qt++) {
        if (qt < nt) {
            f32x4 s[NT], dp[NT];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; kt++) {
                if (kt < nt) {
                    f32x4 a = {bias_r[kt][0], bias_r[kt][1], bias_r[kt][2], bias_r[kt][3]};
                    f32x4 b = {0.f, 0.f, 0.f, 0.f};
                    a = MFMA16(kf[kt].x, qf[qt].x, a); b = MFMA16(vf[kt].x, dof[qt].x, b);
                    a = MFMA16(kf[kt].y, qf[qt].y, a); b = MFMA16(vf[kt].y, dof[qt].y, b);
                    a = MFMA16(kf[kt].z, qf[qt].z, a); b = MFMA16(vf[kt].z, dof[qt].z, b);
                    a = MFMA16(kf[kt].w, qf[qt].w, a); b = MFMA16(vf[kt].w, dof[qt].w, b);
#pragma unroll
                    for (int r = 0; r < 4; r++) mx = fmaxf(mx, a[r]);
                    s[kt] = a;
                    dp[kt] = b;
                }
            }
            mx = g4_max(mx);
            float sum = 0.f, dl = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float p = __builtin_amdgcn_exp2f(s[kt][r] - mx);
                        s[kt][r] = p;
                        sum += p;
                        dl += p * dp[kt][r];
                    }
            sum = g4_sum(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            dl = g4_sum(dl) * inv;
            f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NT; kt++)
                if (kt < nt) {
                    float ds[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float pn = s[kt][r] * inv;
                        ds[r] = pn * (dp[kt][r] - dl);
                        db[kt][r] += ds[r];
                        sc_p[(4 * g4 + r) * LDB + c16] = pn;
                        sc_s[(4 * g4 + r) * LDB + c16] = ds[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) dq = MFMA16(ks[kt][r], ds[r], dq);
                    // same wave wrote it; LDS serves one wave's requests in order
                    const float4 pb = *reinterpret_cast<const float4*>(sc_p + c16 * LDB + 4 * g4);
                    const float4 sb = *reinterpret_cast<const float4*>(sc_s + c16 * LDB + 4 * g4);
                    dv[kt] = MFMA16(dos[qt][0], pb.x, dv[kt]); dk[kt] = MFMA16(qs[qt][0], sb.x, dk[kt]);
                    dv[kt] = MFMA16(dos[qt][1], pb.y, dv[kt]); dk[kt] = MFMA16(qs[qt][1], sb.y, dk[kt]);
                    dv[kt] = MFMA16(dos[qt][2], pb.z, dv[kt]); dk[kt] = MFMA16(qs[qt][2], sb.z, dk[kt]);
                    dv[kt] = MFMA16(dos[qt][3], pb.w, dv[kt]); dk[kt] = MFMA16(qs[qt][3], sb.w, dk[kt]);
                }
            *reinterpret_cast<float4*>(sm + (16 * qt + c16) * LDB + qo + 4 * g4) =
                make_float4(dq[0] * scale, dq[1] * scale, dq[2] * scale, dq[3] * scale);
        }
    }
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
        if (kt < nt) {
            store_key_bias(dbias_h, E, head, start, T, kt, g4, c16, row16_sum(db[kt][0]), row16_sum(db[kt][1]),
                           row16_sum(db[kt][2]), row16_sum(db[kt][3]));
            float* row = sm + (16 * kt + c16) * LDB;
            *reinterpret_cast<float4*>(row + ko + 4 * g4) =
                make_float4(dk[kt][0] * scale, dk[kt][1] * scale, dk[kt][2] * scale, dk[kt][3] * scale);
            *reinterpret_cast<float4*>(row + vo + 4 * g4) = make_float4(dv[kt][0], dv[kt][1], dv[kt][2], dv[kt][3]);
        }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * (3 * D / 4); idx += 512) {
        const int t = idx / (3 * D / 4), c = idx % (3 * D / 4);
        *reinterpret_cast<float4*>(dQKV + tok_row(t, T, E, atom, start) * (3 * D) + 4 * c) =
            *reinterpret_cast<const float4*>(sm + t * LDB + 4 * c);
    }
}

// ---- persistent adjoint with LDS-DMA prefetch ----------------------------------------------------------------
// k_attn_bwd_l is bound by bytes in flight, not by HBM or issue slots: each workgroup loads (~40 KB), computes,
// stores, and only ~1/4 of the resident workgroups are in their load phase at any time (measured 2.3 TB/s with
// FETCH+WRITE = algorithmic bytes). Here a workgroup walks a strided list of atoms and, as soon as the current
// atom's operand fragments are in registers, refills the SAME staging rows with the next atom's Q/K/V/dO rows
// by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave-instruction, no registers, not counted by the compiler's
// s_waitcnt bookkeeping), so the next load is in flight for the whole compute phase. dQ/dK/dV leave straight
// from the accumulator registers (64 B segments per lane quad). Serves atoms with at most `cap` tokens
// (cap = 32: two 32-row staging buffers + the transpose scratch = 149 KB, one workgroup of 8 waves per CU; the
// compiler may then use up to 256 registers, which the preloaded fragments need); the rest fall to k_attn_bwd_l.
constexpr int SCP = 20;  // scratch row pitch: (4 g4 + r) * 20 + c16 hits 64 distinct banks
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void attn_issue_rows(const float* __restrict__ QKV, const float* __restrict__ dAO,
                                                const float* sm, int atom, int start, int T, int64_t E) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned base = (unsigned)(uintptr_t)sm;
    for (int h = wave; h < 2 * T; h += 8) {  // one wave-instruction per half row (64 float4)
        const int t = h >> 1;
        const int64_t row = tok_row(t, T, E, atom, start);
        const float* src = (h & 1) == 0 ? QKV + row * (3 * D) + 4 * lane
                           : lane < 32  ? QKV + row * (3 * D) + 256 + 4 * lane
                                        : dAO + row * D + 4 * (lane - 32);
        glds16(src, __builtin_amdgcn_readfirstlane(base + (unsigned)(t * LDB + (h & 1) * 256) * 4u));
    }
}

template <int NT>
__global__ __launch_bounds__(512) void k_attn_bwd_a(const float* __restrict__ QKV, const float* __restrict__ dAO,
                                                     const int* __restrict__ rowptr, const float* __restrict__ fc,
                                                     float* __restrict__ dQKV, float* __restrict__ dbias_h,
                                                     int64_t E, int N, float scale, int cap,
                                                     const int* __restrict__ atoms, int n_list) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // 2 x [cap][LDB] rows, then [8 waves][2][16][SCP]
    __shared__ float sbias_all[2][16 * NT];                     // log2 of the cutoff factor per key, per buffer
    const int head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qo = HD * head, ko = D + HD * head, vo = 2 * D + HD * head, doo = 3 * D + HD * head;
    const float s2 = scale * LOG2E;
    const int step = gridDim.x;
    int li = blockIdx.x;  // walks the graph's list of the atoms with at most `cap` tokens
    int atom = li < n_list ? atoms[li] : N;
    float fc_pre = 1.f;  // thread t holds the cutoff factor of key t of the atom in flight
    int it = 0;
    if (atom < N) {
        const int st0 = rowptr[atom], T0 = rowptr[atom + 1] - st0 + 1;
        attn_issue_rows(QKV, dAO, sm, atom, st0, T0, E);
        if (threadIdx.x >= 1 && (int)threadIdx.x < T0) fc_pre = fc[st0 + threadIdx.x - 1];
    }
    while (atom < N) {
        const int start = rowptr[atom];
        const int T = rowptr[atom + 1] - start + 1;
        const int nt = (T + 15) >> 4;
        const int lane = threadIdx.x & 63;
        const int c16 = lane & 15, g4 = lane >> 4;
        const float* buf = sm + (it & 1) * cap * LDB;  // this atom's rows; the other buffer takes the next atom's
        float* sc_p = sm + 2 * cap * LDB + head * (2 * 16 * SCP);
        float* sc_s = sc_p + 16 * SCP;
        float* sbias = sbias_all[it & 1];
        if (threadIdx.x < 16 * NT)
            sbias[threadIdx.x] = (int)threadIdx.x >= T ? -INFINITY
                                 : threadIdx.x == 0    ? 0.f
                                                       : __builtin_amdgcn_logf(fmaxf(fc_pre, 1e-15f));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces (and last atom's stores) landed
        __syncthreads();                                  // ... and everybody else's
        // everyone is also done with the other buffer (read in the previous iteration's prologue): refill it now,
        // so the next atom's rows are in flight during this atom's whole compute phase
        li += step;
        const int nxt = li < n_list ? atoms[li] : N;
        fc_pre = 1.f;
        if (nxt < N) {
            const int st1 = rowptr[nxt], T1 = rowptr[nxt + 1] - st1 + 1;
            attn_issue_rows(QKV, dAO, sm + ((it + 1) & 1) * cap * LDB, nxt, st1, T1, E);
            if (threadIdx.x >= 1 && (int)threadIdx.x < T1) fc_pre = fc[st1 + threadIdx.x - 1];
        }
        float4 kf[NT], vf[NT], qf[NT], dof[NT];
        float ks[NT][4], qs[NT][4], dos[NT][4];
        float bias_r[NT][4], db[NT][4];
        f32x4 dk[NT], dv[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (t < nt) {
                const int tc = 16 * t + c16;
                const float* row = buf + tc * LDB;
                const bool in = tc < T;  // rows beyond T hold stale data: read them anyway, select afterwards
                const float4 k4 = *reinterpret_cast<const float4*>(row + ko + 4 * g4);
                const float4 v4 = *reinterpret_cast<const float4*>(row + vo + 4 * g4);
                const float4 q4 = *reinterpret_cast<const float4*>(row + qo + 4 * g4);
                const float4 d4 = *reinterpret_cast<const float4*>(row + doo + 4 * g4);
                kf[t] = make_float4(in ? k4.x : 0.f, in ? k4.y : 0.f, in ? k4.z : 0.f, in ? k4.w : 0.f);
                vf[t] = make_float4(in ? v4.x : 0.f, in ? v4.y : 0.f, in ? v4.z : 0.f, in ? v4.w : 0.f);
                qf[t] = make_float4(in ? q4.x * s2 : 0.f, in ? q4.y * s2 : 0.f, in ? q4.z * s2 : 0.f,
                                    in ? q4.w * s2 : 0.f);
                dof[t] = make_float4(in ? d4.x : 0.f, in ? d4.y : 0.f, in ? d4.z : 0.f, in ? d4.w : 0.f);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int tr = 16 * t + 4 * g4 + r;
                    const float* rr = buf + tr * LDB;
                    const bool inr = tr < T;
                    const float k1 = rr[ko + c16], q1 = rr[qo + c16], d1 = rr[doo + c16];
                    ks[t][r] = inr ? k1 : 0.f;
                    qs[t][r] = inr ? q1 : 0.f;
                    dos[t][r] = inr ? d1 : 0.f;
                    bias_r[t][r] = sbias[tr];
                    db[t][r] = 0.f;
                    dk[t][r] = 0.f;
                    dv[t][r] = 0.f;
                }
            }
        }
#pragma unroll
        for (int qt = 0; qt < NT; qt++) {
            if (qt < nt) {
                f32x4 s[NT], dp[NT];
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT; kt++) {
                    if (kt < nt) {
                        f32x4 a = {bias_r[kt][0], bias_r[kt][1], bias_r[kt][2], bias_r[kt][3]};
                        f32x4 b = {0.f, 0.f, 0.f, 0.f};
                        a = MFMA16(kf[kt].x, qf[qt].x, a); b = MFMA16(vf[kt].x, dof[qt].x, b);
                        a = MFMA16(kf[kt].y, qf[qt].y, a); b = MFMA16(vf[kt].y, dof[qt].y, b);
                        a = MFMA16(kf[kt].z, qf[qt].z, a); b = MFMA16(vf[kt].z, dof[qt].z, b);
                        a = MFMA16(kf[kt].w, qf[qt].w, a); b = MFMA16(vf[kt].w, dof[qt].w, b);
#pragma unroll
                        for (int r = 0; r < 4; r++) mx = fmaxf(mx, a[r]);
                        s[kt] = a;
                        dp[kt] = b;
                    }
                }
                mx = g4_max(mx);
                float sum = 0.f, dl = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; kt++)
                    if (kt < nt)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float p = __builtin_amdgcn_exp2f(s[kt][r] - mx);
                            s[kt][r] = p;
                            sum += p;
                            dl += p * dp[kt][r];
                        }
                sum = g4_sum(sum);
                const float inv = __builtin_amdgcn_rcpf(sum);
                dl = g4_sum(dl) * inv;
                f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NT; kt++)
                    if (kt < nt) {
                        float ds[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float pn = s[kt][r] * inv;
                            ds[r] = pn * (dp[kt][r] - dl);
                            db[kt][r] += ds[r];
                            sc_p[(4 * g4 + r) * SCP + c16] = pn;
                            sc_s[(4 * g4 + r) * SCP + c16] = ds[r];
                        }
#pragma unroll
                        for (int r = 0; r < 4; r++) dq = MFMA16(ks[kt][r], ds[r], dq);
                        const float4 pb = *reinterpret_cast<const float4*>(sc_p + c16 * SCP + 4 * g4);
                        const float4 sb = *reinterpret_cast<const float4*>(sc_s + c16 * SCP + 4 * g4);
                        dv[kt] = MFMA16(dos[qt][0], pb.x, dv[kt]); dk[kt] = MFMA16(qs[qt][0], sb.x, dk[kt]);
                        dv[kt] = MFMA16(dos[qt][1], pb.y, dv[kt]); dk[kt] = MFMA16(qs[qt][1], sb.y, dk[kt]);
                        dv[kt] = MFMA16(dos[qt][2], pb.z, dv[kt]); dk[kt] = MFMA16(qs[qt][2], sb.z, dk[kt]);
                        dv[kt] = MFMA16(dos[qt][3], pb.w, dv[kt]); dk[kt] = MFMA16(qs[qt][3], sb.w, dk[kt]);
                    }
                const int tq = 16 * qt + c16;
                if (tq < T)
                    *reinterpret_cast<float4*>(dQKV + tok_row(tq, T, E, atom, start) * (3 * D) + qo + 4 * g4) =
                        make_float4(dq[0] * scale, dq[1] * scale, dq[2] * scale, dq[3] * scale);
            }
        }
#pragma unroll
        for (int kt = 0; kt < NT; kt++)
            if (kt < nt) {
                store_key_bias(dbias_h, E, head, start, T, kt, g4, c16, row16_sum(db[kt][0]), row16_sum(db[kt][1]),
                               row16_sum(db[kt][2]), row16_sum(db[kt][3]));
                const int tk = 16 * kt + c16;
                if (tk < T) {
                    float* out = dQKV + tok_row(tk, T, E, atom, start) * (3 * D);
                    *reinterpret_cast<float4*>(out + ko + 4 * g4) =
                        make_float4(dk[kt][0] * scale, dk[kt][1] * scale, dk[kt][2] * scale, dk[kt][3] * scale);
                    *reinterpret_cast<float4*>(out + vo + 4 * g4) = make_float4(dv[kt][0], dv[kt][1], dv[kt][2], dv[kt][3]);
                }
            }
        atom = nxt;
        it++;
    }
}

// The attention adjoint of the three-kernel form (the forward is always k_attn_fwd_p, one wave per (atom, head) straight from global
// memory): a persistent kernel with LDS-DMA prefetch (k_attn_bwd_a) for atoms of at most 32 tokens, the per-atom LDS-staged k_attn_bwd_l
// for the rest. (Rounds 1 - 5 kept a switch, "attn_lds", that sent every atom through k_attn_bwd_l; measured per launch on 8 x 10k-atom
// boxes: 2.5 ms against 1.8 ms. Removed in round 6 with the two instantiations only it reached.)
// The generic wave-per-head kernels of pet_fwd.hip / pet_bwd.hip serve more than 64 tokens per atom and trr = 0.

// Atoms are served by the instantiation that matches their own tile count (registers / LDS, hence waves in
// flight, scale with NT): one launch per tile count, over the graph's list of the atoms that have it.
bool attn_fwd_preload(int nt, const float* QKV, const Graph& g, float* AO, float scale, hipStream_t st) {
    if (nt > 4) return false;
    const int N = (int)g.n_nodes;
    if (N <= 4096) {
        // small graphs: ONE launch of the largest instantiation over the atoms of every tile count (the lists are consecutive
        // runs of atom_order; the kernel walks an atom's own tile count, so the arithmetic and its order are those of the
        // bucketed launches) -- each launch on the critical path of a 1 000-atom box costs 5 - 14 us
        int km = 0;
        for (int K = 1; K <= 4 && K <= nt; K++)
            if (bucket_count(g, K) > 0) km = K;
        const int n_all = km ? g.bucket_start[km] - g.bucket_start[0] : 0;
        if (n_all <= 0) return true;
        const int grid = cdiv((int64_t)n_all * NHEAD, 4);
        const int* atoms = g.atom_order + g.bucket_start[0];
        switch (km) {
            case 1: k_attn_fwd_p<1><<<grid, 256, 0, st>>>(QKV, g.rowptr, g.fc, AO, g.n_edges, N, scale, atoms, n_all); break;
            case 2: k_attn_fwd_p<2><<<grid, 256, 0, st>>>(QKV, g.rowptr, g.fc, AO, g.n_edges, N, scale, atoms, n_all); break;
            case 3: k_attn_fwd_p<3><<<grid, 256, 0, st>>>(QKV, g.rowptr, g.fc, AO, g.n_edges, N, scale, atoms, n_all); break;
            default: k_attn_fwd_p<4><<<grid, 256, 0, st>>>(QKV, g.rowptr, g.fc, AO, g.n_edges, N, scale, atoms, n_all); break;
        }
        return true;
    }
#define PET_ATTN_FWD(K)                                                                                            \
    if (nt >= K && bucket_count(g, K) > 0)                                                                         \
        k_attn_fwd_p<K><<<cdiv((int64_t)bucket_count(g, K) * NHEAD, 4), 256, 0, st>>>(                             \
            QKV, g.rowptr, g.fc, AO, g.n_edges, N, scale, bucket_atoms(g, K), bucket_count(g, K));
    PET_ATTN_FWD(1) PET_ATTN_FWD(2) PET_ATTN_FWD(3) PET_ATTN_FWD(4)
#undef PET_ATTN_FWD
    return true;
}
bool attn_bwd_preload(int nt, const float* QKV, const float* dAO, const Graph& g, float* dQKV, float* dbias_h,
                      float scale, hipStream_t st) {
    if (nt > 4) return false;
    const int N = (int)g.n_nodes;
    constexpr int first = 3;
    {  // persistent LDS-DMA kernel for every atom with at most `cap` tokens (tile counts 1 and 2)
        constexpr int cap = 32;
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        }
        const int n_list = g.bucket_start[2];
        if (n_list > 0) {
            const size_t lds = ((size_t)2 * cap * LDB + 8 * 2 * 16 * SCP) * sizeof(float);  // 149 KB: one per CU
            allow_big_lds(k_attn_bwd_a<2>, lds);
            const int grid = n_list < n_cu ? n_list : n_cu;
            k_attn_bwd_a<2><<<grid, 512, lds, st>>>(QKV, dAO, g.rowptr, g.fc, dQKV, dbias_h, g.n_edges, N, scale, cap,
                                                    g.atom_order, n_list);
        }
    }
#define PET_ATTN_BWD_L(K)                                                                                      \
    if (nt >= K && K >= first && bucket_count(g, K) > 0) {                                                     \
        const size_t lds = (size_t)16 * K * LDB * sizeof(float);                                               \
        allow_big_lds(k_attn_bwd_l<K>, lds);                                                                   \
        k_attn_bwd_l<K><<<bucket_count(g, K), 512, lds, st>>>(QKV, dAO, g.rowptr, g.fc, dQKV, dbias_h,         \
                                                              g.n_edges, N, scale, bucket_atoms(g, K));        \
    }
    PET_ATTN_BWD_L(3) PET_ATTN_BWD_L(4)
#undef PET_ATTN_BWD_L
    return true;
}

}  // namespace pet
