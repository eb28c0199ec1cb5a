// PET forward on gfx950: fused row-tile kernels (fp32 MFMA) + per-atom attention.
//
// Token layout: one buffer X[(E+N), D]; rows 0..E-1 are the edges in CSR order
// (edge p belongs to atom ctr[p], rows rowptr[i]..rowptr[i+1]-1), rows E..E+N-1
// are the centre tokens. The reference pads every atom to max_neighbors and runs
// [N, M+1, D] batches (pet/modules/transformer.py:463-562); here there are no pads.
//
// Reference map:
//   k_compress   transformer.py:499-521 (edge_embedder, neighbor_embedder, compress MLP)
//   k_center     transformer.py:211-214 (center_contraction)
//   k_qkv        transformer.py:216-218 + 103-107 (norm_attention, input_linear)
//   k_attn_fwd   transformer.py:108-151 / 565-589 (softmax(QK^T/sqrt(hd)/tau + log fc) V)
//   k_oproj      transformer.py:151 + 229 (output_linear, edge residual)
//   k_node       transformer.py:222-227 (center_expansion, center_mlp)
//   k_emlp       transformer.py:230-232 (edge SwiGLU MLP)
//   (combination stage backend.py:559-575: k_comb_p2 in pet_comb.hip)
//   k_head_*     backend.py:651-687, 726-777 (heads, last layers, cutoff-weighted sum)
#include <mutex>
#include <set>

#include "common.h"
#include "model.h"
#include "pet_ws.h"
#include "tile.h"

namespace pet {

constexpr int LD128 = lds_ld(128);
constexpr int LD256 = lds_ld(256);

// store this wave's NT accumulator tiles to a row-major global array
template <int NT>
__device__ __forceinline__ void store_acc(f32x16 (&acc)[NT], float* __restrict__ G, int64_t row0,
                                          int64_t n_rows, int ld, int rb, int col0, int lane) {
    acc_foreach<NT>(acc, rb, col0, lane, [&](int r, int c, float v) {
        if (row0 + r < n_rows) G[(row0 + r) * ld + c] = v;
    });
}

// ---------------------------------------------------------------------------------
// node embedding lookup
// ---------------------------------------------------------------------------------
__global__ void k_node_embed(const int* __restrict__ sp, const float* __restrict__ emb,
                             float* __restrict__ H, int n) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
    if (idx >= (int64_t)n * (DN / 4)) return;
    int i = (int)(idx / (DN / 4)), c = (int)(idx % (DN / 4));
    reinterpret_cast<float4*>(H)[idx] = reinterpret_cast<const float4*>(emb + (size_t)sp[i] * DN)[c];
}

// ---------------------------------------------------------------------------------
// system conditioning (conditioning.py:82-100): cond[s] = W2 silu(W0 [emb_q[charge_s + max_charge] ; emb_m[mult_s - 1]] + b0) + b2,
// one 256-thread block per SYSTEM (a handful of rows: plain fp32 dot products), then h_i += cond[system of i] after every
// GNN layer (backend.py:543-545, :628-629)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(DN) void k_system_cond(const int64_t* __restrict__ charge, const int64_t* __restrict__ spin,
                                                   const float* __restrict__ emb_q, const float* __restrict__ emb_m,
                                                   const float* __restrict__ w0, const float* __restrict__ b0,
                                                   const float* __restrict__ w2, const float* __restrict__ b2,
                                                   float* __restrict__ cond, int max_charge, int max_spin) {
    __shared__ float x[2 * DN], hid[DN];
    const int s = blockIdx.x, t = threadIdx.x;
    int q = (int)charge[s] + max_charge, mi = (int)spin[s] - 1;
    q = q < 0 ? 0 : (q > 2 * max_charge ? 2 * max_charge : q);  // the caller validates (conditioning.py:54-80); stay in bounds
    mi = mi < 0 ? 0 : (mi > max_spin - 1 ? max_spin - 1 : mi);
    x[t] = emb_q[(size_t)q * DN + t];
    x[DN + t] = emb_m[(size_t)mi * DN + t];
    __syncthreads();
    float a = b0[t];
    for (int k = 0; k < 2 * DN; k++) a = fmaf(w0[(size_t)t * 2 * DN + k], x[k], a);
    hid[t] = siluf_(a);
    __syncthreads();
    float o = b2[t];
    for (int k = 0; k < DN; k++) o = fmaf(w2[(size_t)t * DN + k], hid[k], o);
    cond[(size_t)s * DN + t] = o;
}
__global__ void k_add_cond(float* __restrict__ H, const float* __restrict__ cond, const int* __restrict__ sys32,
                           const int64_t* __restrict__ sys64, int n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
    if (idx >= (int64_t)n * (DN / 4)) return;
    const int i = (int)(idx / (DN / 4)), c = (int)(idx % (DN / 4));
    const int64_t s = sys64 ? sys64[i] : (int64_t)sys32[i];
    float4 h = reinterpret_cast<float4*>(H)[idx];
    const float4 a = reinterpret_cast<const float4*>(cond + s * DN)[c];
    h.x += a.x; h.y += a.y; h.z += a.z; h.w += a.w;
    reinterpret_cast<float4*>(H)[idx] = h;
}

// ---------------------------------------------------------------------------------
// compress: a0 = [v,d] Wc^T + Tbl[species] (+ M W0c^T);  e = SiLU(a0) W2^T + b2
// ---------------------------------------------------------------------------------
template <bool FIRST>
__global__ __launch_bounds__(NTHREADS) void k_compress(const float4* __restrict__ geo,
                                                        const int* __restrict__ sp_nbr,
                                                        const float* __restrict__ wc,   // [D,4]
                                                        const float* __restrict__ tbl,  // [ns,D]
                                                        const float* __restrict__ Min,  // [E,D] (!FIRST)
                                                        WX w0c, WX w2,
                                                        const float* __restrict__ b2,
                                                        float* __restrict__ a0_out,  // [E,D] or null
                                                        float* __restrict__ Xout,    // [E,D]
                                                        int64_t E) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* S = smem;                 // [64][132]
    float* A = smem + BM * LD128;    // [64][132] (!FIRST)
    float4* geo_s = reinterpret_cast<float4*>(smem + (FIRST ? 1 : 2) * BM * LD128);  // [64]
    int* sp_s = reinterpret_cast<int*>(geo_s + BM);                                   // [64]
    float* rs = reinterpret_cast<float*>(sp_s + BM);  // [64][2] power-of-two row scales for the f16x3 GEMMs
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    if (threadIdx.x < BM) {
        int64_t p = row0 + threadIdx.x;
        geo_s[threadIdx.x] = p < E ? geo[p] : make_float4(0, 0, 0, 0);
        sp_s[threadIdx.x] = p < E ? sp_nbr[p] : 0;
    }
    if (!FIRST) load_rows_to_lds<128>(A, Min, row0, E, D);
    __syncthreads();
    if (!FIRST && w0c.h) {  // messages: un-normalised rows
        tile_row_scales<128>(A, LD128, rs);
        __syncthreads();
    }
    if (FIRST) {
        const int c = threadIdx.x & 127;
        const float4 wv = reinterpret_cast<const float4*>(wc)[c];
        for (int r = threadIdx.x >> 7; r < BM; r += 2) {
            float4 g = geo_s[r];
            float a = fmaf(g.w, wv.w, fmaf(g.z, wv.z, fmaf(g.y, wv.y, g.x * wv.x))) + tbl[sp_s[r] * D + c];
            if (a0_out && row0 + r < E) a0_out[(row0 + r) * D + c] = a;
            S[r * LD128 + c] = siluf_(a);
        }
    } else {
        f32x16 acc[2];
        const int col0 = 64 * w.ch;
        acc_fill_bias<2>(acc, nullptr, 0, w.lane);
        gemm_acc_x<128, 2>(A + w.rb * 32 * LD128, LD128, w0c, 16, 0, 2 * w.ch, acc, w.lane, w0c.h ? rs + 64 * w.rb : nullptr);
        // The geometry / species terms are added AFTER the GEMM. With them pre-filled into `acc` and live across
        // gemm_acc_x, a few (row, 16-column) groups of the result came out different from launch to launch
        // (same inputs, same weights; lanes 48..63 of one wave): not understood, avoided by this order.
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int c = col0 + 32 * t + (w.lane & 31);
            const float4 wv = reinterpret_cast<const float4*>(wc)[c];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = w.rb * 32 + acc_row(r, w.lane);
                float4 g = geo_s[row];
                acc[t][r] += fmaf(g.w, wv.w, fmaf(g.z, wv.z, fmaf(g.y, wv.y, g.x * wv.x))) + tbl[sp_s[row] * D + c];
            }
        }
        acc_foreach<2>(acc, w.rb, col0, w.lane, [&](int r, int c, float v) {
            if (a0_out && row0 + r < E) a0_out[(row0 + r) * D + c] = v;
            S[r * LD128 + c] = siluf_(v);
        });
    }
    __syncthreads();
    if (w2.h) {
        tile_row_scales<128>(S, LD128, rs);
        __syncthreads();
    }
    f32x16 acc2[2];
    acc_fill_bias<2>(acc2, b2, 64 * w.ch, w.lane);
    gemm_acc_x<128, 2>(S + w.rb * 32 * LD128, LD128, w2, 16, 0, 2 * w.ch, acc2, w.lane, w2.h ? rs + 64 * w.rb : nullptr);
    store_acc<2>(acc2, Xout, row0, E, D, w.rb, 64 * w.ch, w.lane);
}

// ---------------------------------------------------------------------------------
// centre contraction: X[E + i] = H[i] Wcc^T + b   (DN -> D)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_center(const float* __restrict__ H, WX wcc,
                                                      const float* __restrict__ bcc, float* __restrict__ Xc,
                                                      int64_t N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    float* rs = smem + BM * LD256;  // [64][2] power-of-two row scales: node features are un-normalised rows
    load_rows_to_lds<256>(smem, H, row0, N, DN);
    __syncthreads();
    if (wcc.h) {
        tile_row_scales<256>(smem, LD256, rs);
        __syncthreads();
    }
    f32x16 acc[2];
    acc_fill_bias<2>(acc, bcc, 64 * w.ch, w.lane);
    gemm_acc_x<256, 2>(smem + w.rb * 32 * LD256, LD256, wcc, 32, 0, 2 * w.ch, acc, w.lane, wcc.h ? rs + 64 * w.rb : nullptr);
    store_acc<2>(acc, Xc, row0, N, D, w.rb, 64 * w.ch, w.lane);
}

// ---------------------------------------------------------------------------------
// QKV = Norm(X) Win^T + b   (D -> 3D) over all E+N token rows; Norm = RMSNorm, LayerNorm (beta) or, for PostLN
// (transformer.py:243: attention on the raw tokens), nothing
// ---------------------------------------------------------------------------------
template <bool NORM>
__global__ __launch_bounds__(NTHREADS) void k_qkv(const float* __restrict__ X, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta,
                                                   const float4* __restrict__ win, const float* __restrict__ bin,
                                                   float* __restrict__ QKV, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<128>(smem, X, row0, R, D);
    __syncthreads();
    if (NORM) {
        norm_rows_inplace<128>(smem, gamma, beta);
        __syncthreads();
    }
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        f32x16 acc[2];
        const int col0 = 128 * c + 64 * w.ch;
        acc_fill_bias<2>(acc, bin, col0, w.lane);
        gemm_acc<128, 2>(smem + w.rb * 32 * LD128, LD128, win, 16, 0, 4 * c + 2 * w.ch, acc, w.lane);
        store_acc<2>(acc, QKV, row0, R, 3 * D, w.rb, col0, w.lane);
    }
}

// ---------------------------------------------------------------------------------
// attention: one wave per (atom, head); 16x16x4 fp32 MFMA, no LDS.
// Scores are produced transposed (keys x queries) so that the soft-maxed tile is
// already in the B-operand layout of the PV product.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int64_t token_row(int t, int T, int64_t E, int atom, int start) {
    // token 0 = centre, token t >= 1 = edge start + t - 1; out-of-range tokens alias token 0
    return (t == 0 || t >= T) ? E + atom : (int64_t)start + t - 1;
}

template <int NT>
__global__ __launch_bounds__(256) void k_attn_fwd(const float* __restrict__ QKV, const int* __restrict__ rowptr,
                                                   const float* __restrict__ fc, float* __restrict__ AO,
                                                   int64_t E, int N, float scale) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int atom = gw / NHEAD, head = gw % NHEAD;
    if (atom >= N) return;
    const int start = rowptr[atom];
    const int T = rowptr[atom + 1] - start + 1;
    const int nt = (T + 15) >> 4;
    const int c16 = lane & 15, g4 = lane >> 4;
    // key fragments (A operand of S^T = K Q^T): K[key = 16 kt + c16][4 g4 .. 4 g4 + 3]
    float4 kf[NT];
    float bias[NT][4];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
        if (kt < nt) {
            const int64_t row = token_row(16 * kt + c16, T, E, atom, start);
            kf[kt] = *reinterpret_cast<const float4*>(QKV + row * (3 * D) + D + HD * head + 4 * g4);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = 16 * kt + 4 * g4 + r;  // C-layout row of the S^T tile
                float b = 0.0f;
                if (key >= T) b = -INFINITY;
                else if (key > 0) b = logf(fmaxf(fc[start + key - 1], 1e-15f));  // transformer.py:109-110
                bias[kt][r] = b;
            }
        }
    }
#pragma unroll 1
    for (int qt = 0; qt < nt; qt++) {
        const int q = 16 * qt + c16;
        const int64_t qrow = token_row(q, T, E, atom, start);
        float4 qf = *reinterpret_cast<const float4*>(QKV + qrow * (3 * D) + HD * head + 4 * g4);
        qf.x *= scale; qf.y *= scale; qf.z *= scale; qf.w *= scale;
        f32x4 s[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
            if (kt < nt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].x, qf.x, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].y, qf.y, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].z, qf.z, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].w, qf.w, a, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    a[r] += bias[kt][r];
                    mx = fmaxf(mx, a[r]);
                }
                s[kt] = a;
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
            if (kt < nt) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float p = expf(s[kt][r] - mx);
                    s[kt][r] = p;
                    sum += p;
                }
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        // O^T[d][q] = sum_key V^T[d][key] P^T[key][q]
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; kt++) {
            if (kt < nt) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t vrow = token_row(16 * kt + 4 * g4 + r, T, E, atom, start);
                    const float vv = QKV[vrow * (3 * D) + 2 * D + HD * head + c16];
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, s[kt][r], o, 0, 0, 0);
                }
            }
        }
        if (q < T) {
            float4 ov = make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
            *reinterpret_cast<float4*>(AO + qrow * D + HD * head + 4 * g4) = ov;
        }
    }
}

// ---------------------------------------------------------------------------------
// output_linear + edge residual:  rows < E: X1 = X + AO Wo^T + b ; rows >= E: OC = AO Wo^T + b
// POST (transformer.py:243-245): every token keeps its residual, X1 = X + AO Wo^T + b on all E+N rows
// ---------------------------------------------------------------------------------
template <bool POST>
__global__ __launch_bounds__(NTHREADS) void k_oproj(const float* __restrict__ AO, const float* __restrict__ X,
                                                     const float4* __restrict__ wo, const float* __restrict__ bo,
                                                     float* __restrict__ X1, float* __restrict__ OC, int64_t E,
                                                     int64_t R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<128>(smem, AO, row0, R, D);
    __syncthreads();
    f32x16 acc[2];
    acc_fill_bias<2>(acc, bo, 64 * w.ch, w.lane);
    gemm_acc<128, 2>(smem + w.rb * 32 * LD128, LD128, wo, 16, 0, 2 * w.ch, acc, w.lane);
    acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
        const int64_t row = row0 + r;
        if (row < E || (POST && row < R)) X1[row * D + c] = X[row * D + c] + v;
        else if (row < R) OC[(row - E) * D + c] = v;
    });
}

// ---------------------------------------------------------------------------------
// node update: h1 = h + OC Wce^T + b ; h2 = h1 + SwiGLU_MLP(RMSNorm(h1))
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_node(const float* __restrict__ H, const float* __restrict__ OC,
                                                    WX wce, const float* __restrict__ bce,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, WX win,
                                                    const float* __restrict__ bin, WX wout,
                                                    const float* __restrict__ bout,
                                                    float* __restrict__ H1, float* __restrict__ VGn,
                                                    float* __restrict__ Hn, int64_t N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;               // [64][260]
    float* U = smem + BM * LD256;   // [64][132] (OC tile, then SwiGLU hidden chunk)
    float* rs = U + BM * LD128;     // [64][2] row scales of the OC tile (un-normalised attention output)
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<128>(U, OC, row0, N, D);
    __syncthreads();
    if (wce.h) {
        tile_row_scales<128>(U, LD128, rs);
        __syncthreads();
    }
#pragma unroll 1
    for (int c = 0; c < 2; c++) {  // 256 output columns in two chunks of 128
        f32x16 acc[2];
        const int col0 = 128 * c + 64 * w.ch;
        acc_fill_bias<2>(acc, bce, col0, w.lane);
        gemm_acc_x<128, 2>(U + w.rb * 32 * LD128, LD128, wce, 16, 0, 4 * c + 2 * w.ch, acc, w.lane,
                           wce.h ? rs + 64 * w.rb : nullptr);
        acc_foreach<2>(acc, w.rb, col0, w.lane, [&](int r, int cc, float v) {
            const int64_t row = row0 + r;
            float h1 = v + (row < N ? H[row * DN + cc] : 0.f);
            Hs[r * LD256 + cc] = h1;
            if (row < N) H1[row * DN + cc] = h1;
        });
    }
    __syncthreads();
    norm_rows_inplace<256>(Hs, gamma, beta);
    __syncthreads();
    f32x16 out[4];  // this wave: 32 rows x 128 columns (128 * ch ..)
    acc_fill_bias<4>(out, bout, 128 * w.ch, w.lane);
#pragma unroll 1
    for (int hc = 0; hc < DNF / 128; hc++) {
        f32x16 av[2], ag[2];
        const int hcol0 = 128 * hc + 64 * w.ch;  // hidden columns of this wave
        acc_fill_bias<2>(av, bin, hcol0, w.lane);
        acc_fill_bias<2>(ag, bin, DNF + hcol0, w.lane);
        gemm_acc_x<256, 2>(Hs + w.rb * 32 * LD256, LD256, win, 32, 0, hcol0 / 32, av, w.lane);
        gemm_acc_x<256, 2>(Hs + w.rb * 32 * LD256, LD256, win, 32, 0, (DNF + hcol0) / 32, ag, w.lane);
        __syncthreads();  // previous chunk's readers of U are done
#pragma unroll
        for (int t = 0; t < 2; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = w.rb * 32 + acc_row(r, w.lane);
                const int cc = 64 * w.ch + 32 * t + (w.lane & 31);
                const float v = av[t][r], gte = ag[t][r];
                if (VGn && row0 + rr < N) {
                    VGn[(row0 + rr) * (2 * DNF) + 128 * hc + cc] = v;
                    VGn[(row0 + rr) * (2 * DNF) + DNF + 128 * hc + cc] = gte;
                }
                U[rr * LD128 + cc] = v * sigmoidf_(gte);  // transformer.py:42-43: value * sigmoid(gate)
            }
        }
        __syncthreads();
        gemm_acc_x<128, 4>(U + w.rb * 32 * LD128, LD128, wout, DNF / 8, 16 * hc, 4 * w.ch, out, w.lane);
    }
    acc_foreach<4>(out, w.rb, 128 * w.ch, w.lane, [&](int r, int c, float v) {
        const int64_t row = row0 + r;
        if (row < N) Hn[row * DN + c] = H1[row * DN + c] + v;
    });
}

// The same stage with the memory side rebuilt (k_node spends more than half of its time outside the GEMMs: one float
// per lane to and from global memory at one workgroup per CU): the H rows arrive as a coalesced tile, h1 leaves as one,
// the SwiGLU pre-activations and the result leave through wave-private staging tiles as float4 rows (tile.h
// wave_rows64); and the normalised rows are split ONCE into fp16 planes (gemm_acc_hs) for the 16 column-chunk GEMMs of
// the centre MLP's first Linear. LDS: [ h tile fp32 -> (SwiGLU chunk | 4 staging tiles) ][ OC tile -> planes ][ scales ].
// RB = 2: 64 rows per workgroup, waves = 2 row blocks x 2 column halves (1 workgroup per CU: 133 KB of LDS). RB = 1: 32 rows,
// the four waves own a quarter of the columns each and two workgroups share a CU (67 KB): half the dependent chain per
// wave and a second workgroup to fill its stalls -- the node chain sits on the critical path of a small box (one
// k_node2 per attention layer between the output projection and the next layer's centre tokens).
// SPLIT (RB = 1, graphs of at most 128 row tiles): the four 128-unit chunks of the hidden layer go to four workgroups
// (blockIdx.y) -- a workgroup of a small graph is alone on its CU and its time is the round trips of the 1.8 MB of weights it
// streams (bytes in flight / latency = 25 GB/s per CU), so four CUs per row tile stream a quarter each. Every workgroup
// forms h1 and its planes, runs its chunk and leaves a partial [32 x 256] output in Pp; the one that arrives last at the
// tile's counter (fence + atomic, the counter resets itself) adds bias and partials in chunk order -- the order of the
// unsplit accumulation, so the same bits whoever is last -- and finishes (Hn, the next layer's centre tokens).
template <int RB, bool SPLIT = false>
__global__ __launch_bounds__(NTHREADS) void k_node2(const float* __restrict__ H, const float* __restrict__ OC,
                                                     WX wce, const float* __restrict__ bce,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, WX win,
                                                     const float* __restrict__ bin, WX wout,
                                                     const float* __restrict__ bout,
                                                     float* __restrict__ H1, float* __restrict__ VGn,
                                                     float* __restrict__ Hn, int64_t N, WX wcn,
                                                     const float* __restrict__ bcn, float* __restrict__ Xcn,
                                                     float* __restrict__ Pp, int* __restrict__ cnt) {
    static_assert(!SPLIT || RB == 1, "the hidden-chunk split is built for the 32-row workgroups");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 32 * RB, NCH = 4 / RB;      // rows per workgroup, column groups
    constexpr int WC = 256 / NCH, HC = 128 / NCH;    // output columns / hidden columns (per 128-chunk) of one wave
    constexpr int NTH = HC / 32, NTO = WC / 32;
    constexpr int XD = RB == 1 ? 8 : 2;  // weight blocks in flight in the products that split their A operand on the fly
    constexpr int LDH = plane_ld(256);
    float* Hs = smem;                                                  // [ROWS][260] h, then h1, then dead
    float* U = smem;                                                   // [ROWS][132] SwiGLU chunk (aliases Hs)
    float* stage = smem + ROWS * LD128;                                // 4 staging tiles (behind U, inside Hs)
    _Float16* Ph = reinterpret_cast<_Float16*>(smem + ROWS * LD256);   // [ROWS][264] high pieces of Norm(h1)
    _Float16* Pl = Ph + ROWS * LDH;                                    // [ROWS][264] low pieces
    float* OCs = smem + ROWS * LD256;                                  // [ROWS][132] OC tile (before the planes exist)
    float* rs = smem + ROWS * LD256 + ROWS * LDH;                      // [ROWS][2] row scales of the OC tile
    const WaveIdT<RB> w;
    float* my_stage = stage + w.wave * (RB == 2 ? 32 * 64 : 32 * 32);
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    load_rows_to_lds<128, ROWS>(OCs, OC, row0, N, D);
    load_rows_to_lds<256, ROWS>(Hs, H, row0, N, DN);
    __syncthreads();
    tile_row_scales<128, ROWS>(OCs, LD128, rs);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < RB; c++) {  // this wave's WC output columns in groups of 64
        f32x16 acc[2];
        const int col0 = 64 * (NCH * c + w.ch);
        acc_fill_bias<2>(acc, bce, col0, w.lane);
        gemm_acc_x<128, 2, XD>(OCs + w.rb * 32 * LD128, LD128, wce, 16, 0, col0 / 32, acc, w.lane, rs + 64 * w.rb);
        acc_foreach<2>(acc, w.rb, col0, w.lane, [&](int r, int cc, float v) { Hs[r * LD256 + cc] += v; });  // h1 = h + ...
    }
    __syncthreads();
    if constexpr (SPLIT) {  // (read again by whichever workgroup finishes the tile)
        if (blockIdx.y == 0) store_rows_from_lds_agent<256, ROWS>(Hs, H1, row0, N, DN);
    } else store_rows_from_lds<256, ROWS>(Hs, H1, row0, N, DN);
    __syncthreads();
    norm_rows_inplace<256, ROWS>(Hs, gamma, beta);
    __syncthreads();
    split_tile_planes<256, ROWS>(Hs, LD256, Ph, Pl);
    __syncthreads();  // Hs is dead from here on: U and the staging tiles reuse its memory
    f32x16 out[NTO];  // this wave: 32 rows x WC columns (WC * ch ..)
    acc_fill_bias<NTO>(out, SPLIT ? nullptr : bout, WC * w.ch, w.lane);
    const int64_t wrow0 = row0 + 32 * w.rb;  // first row of this wave's block
    const int hc_lo = SPLIT ? (int)blockIdx.y : 0, hc_hi = SPLIT ? (int)blockIdx.y + 1 : DNF / 128;
#pragma unroll 1
    for (int hc = hc_lo; hc < hc_hi; hc++) {
        f32x16 av[NTH], ag[NTH];
        const int hcol0 = 128 * hc + HC * w.ch;  // hidden columns of this wave
        acc_fill_bias<NTH>(av, bin, hcol0, w.lane);
        acc_fill_bias<NTH>(ag, bin, DNF + hcol0, w.lane);
        XRing<NTO, 8> ro;  // RB == 1: this chunk's K = 128 operand of the output product, requested before the hidden products
        const bool ring = RB == 1 && wout.h != nullptr;
        if (ring) xring_request(ro, wout, DNF / 8, 16 * hc, NTO * w.ch, w.lane);
        if constexpr (RB == 1) {  // value and gate tile in ONE product (one exposed round trip instead of two); same MFMA order per tile
            f32x16 vg[2] = {av[0], ag[0]};
            gemm_acc_hs<256, 2, 4>(Ph + w.rb * 32 * LDH, Pl + w.rb * 32 * LDH, LDH, win, 32, 0, hcol0 / 32, vg, w.lane, DNF / 32);
            av[0] = vg[0];
            ag[0] = vg[1];
        } else {
        gemm_acc_hs<256, NTH, 8>(Ph + w.rb * 32 * LDH, Pl + w.rb * 32 * LDH, LDH, win, 32, 0, hcol0 / 32, av, w.lane);
        gemm_acc_hs<256, NTH, 8>(Ph + w.rb * 32 * LDH, Pl + w.rb * 32 * LDH, LDH, win, 32, 0, (DNF + hcol0) / 32, ag, w.lane);
        }
        if (VGn) {  // saved for the adjoint: [value | gate] pre-activations, whole float4 rows
            if constexpr (RB == 2) {
                wave_rows64(av, my_stage, w.lane, [&](int r, int cc, float4 v) {
                    if (wrow0 + r < N) *reinterpret_cast<float4*>(VGn + (wrow0 + r) * (2 * DNF) + hcol0 + cc) = v;
                });
                wave_rows64(ag, my_stage, w.lane, [&](int r, int cc, float4 v) {
                    if (wrow0 + r < N) *reinterpret_cast<float4*>(VGn + (wrow0 + r) * (2 * DNF) + DNF + hcol0 + cc) = v;
                });
            } else {
                wave_rows32(av[0], my_stage, w.lane, [&](int r, int cc, float4 v) {
                    if (wrow0 + r < N) *reinterpret_cast<float4*>(VGn + (wrow0 + r) * (2 * DNF) + hcol0 + cc) = v;
                });
                wave_rows32(ag[0], my_stage, w.lane, [&](int r, int cc, float4 v) {
                    if (wrow0 + r < N) *reinterpret_cast<float4*>(VGn + (wrow0 + r) * (2 * DNF) + DNF + hcol0 + cc) = v;
                });
            }
        }
        __syncthreads();  // previous chunk's readers of U are done
#pragma unroll
        for (int t = 0; t < NTH; t++)
#pragma unroll
            for (int r = 0; r < 16; r++)  // transformer.py:42-43: value * sigmoid(gate)
                U[(w.rb * 32 + acc_row(r, w.lane)) * LD128 + HC * w.ch + 32 * t + (w.lane & 31)] = av[t][r] * sigmoidf_(ag[t][r]);
        __syncthreads();
        if (ring) gemm_acc_x_ring<NTO, 8>(U + w.rb * 32 * LD128, LD128, ro, out, w.lane);
        else gemm_acc_x<128, NTO, XD>(U + w.rb * 32 * LD128, LD128, wout, DNF / 8, 16 * hc, NTO * w.ch, out, w.lane);
    }
    // Xcn: the NEXT attention layer's centre tokens = center_contraction(Hn) (transformer.py:211-214) from the Hn tile while
    // it is on chip (k_center's arithmetic: power-of-two row scales, the same GEMM) -- one launch and one stream hand-over
    // less per layer on the critical path of a small box
    float* Hn_s = smem + ROWS * LD256;  // [ROWS][260]: over the planes, which nobody reads after the last hidden chunk
    if constexpr (SPLIT) {
        __shared__ int last_arrival;
        const size_t prows = (size_t)gridDim.x * ROWS;  // rows of one partial (whole tiles)
#pragma unroll
        for (int t = 0; t < NTO; t++)
            wave_rows32(out[t], my_stage, w.lane, [&](int r, int cc, float4 v) {
                st4_agent(Pp + ((size_t)blockIdx.y * prows + (wrow0 + r)) * DN + WC * w.ch + 32 * t + cc, v);
            });
        // the partial (and h1) went out as device-coherent stores: once they are acknowledged (the workgroup-scope release
        // waits for that) the workgroup may be counted -- no L2 write-back fence, which costs 10+ us with a step's dirty lines
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = atomicAdd(&cnt[blockIdx.x], 1);
            last_arrival = old == (int)gridDim.y - 1;
            if (last_arrival) cnt[blockIdx.x] = 0;  // ready for the next launch on this stream
        }
        __syncthreads();
        if (!last_arrival) return;
        // Consumer side of the hand-over: an agent-scope ACQUIRE (buffer_inv, no write-back) orders the loads below after the
        // counter update this workgroup saw. Producer side: the partials are relaxed agent-scope atomic stores (write-through
        // to the device-coherent level) that the workgroup-scope release above has waited for (s_waitcnt vmcnt(0)); an
        // agent-scope RELEASE would add the L2 write-back of every dirty line of the step (10+ us) for
        // data that already went through. That part relies on gfx950's write-through behaviour of sc1 stores rather than on
        // the formal memory model; tests/test_gpu_stress.py (node_split_stress) runs the hand-over 200 x and demands
        // bit-identical results.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // every load of the reduction in flight before the first sum (8 positions x (4 partials + h1) per thread)
        constexpr int IT = ROWS * (DN / 4) / NTHREADS, NCHK = DNF / 128;
        float4 pk[IT][NCHK], h1v[IT];
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int idx = threadIdx.x + it * NTHREADS, r = idx / (DN / 4), c = 4 * (idx % (DN / 4));
#pragma unroll
            for (int k = 0; k < NCHK; k++) pk[it][k] = ld4_agent(Pp + ((size_t)k * prows + row0 + r) * DN + c);
            h1v[it] = ld4_agent(H1 + (row0 + r < N ? row0 + r : N - 1) * DN + c);
        }
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const int idx = threadIdx.x + it * NTHREADS, r = idx / (DN / 4), c = 4 * (idx % (DN / 4));
            float4 sum = *reinterpret_cast<const float4*>(bout + c);
#pragma unroll
            for (int k = 0; k < NCHK; k++)  // ((((b + chunk 0) + chunk 1) + chunk 2) + chunk 3): the unsplit order
                sum = make_float4(sum.x + pk[it][k].x, sum.y + pk[it][k].y, sum.z + pk[it][k].z, sum.w + pk[it][k].w);
            const int64_t row = row0 + r;
            float4 hn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < N) {
                hn = make_float4(h1v[it].x + sum.x, h1v[it].y + sum.y, h1v[it].z + sum.z, h1v[it].w + sum.w);
                *reinterpret_cast<float4*>(Hn + row * DN + c) = hn;
            }
            if (Xcn) *reinterpret_cast<float4*>(Hn_s + r * LD256 + c) = hn;
        }
    } else {
    if (Xcn) __syncthreads();
    auto add_h1 = [&](int col) {  // Hn = h1 + MLP
        return [&, col](int r, int cc, float4 v) {
            const int64_t row = wrow0 + r;
            float4 hn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < N) {
                const int64_t o = row * DN + col + cc;
                const float4 h1 = *reinterpret_cast<const float4*>(H1 + o);
                hn = make_float4(h1.x + v.x, h1.y + v.y, h1.z + v.z, h1.w + v.w);
                *reinterpret_cast<float4*>(Hn + o) = hn;
            }
            if (Xcn) *reinterpret_cast<float4*>(Hn_s + (w.rb * 32 + r) * LD256 + col + cc) = hn;
        };
    };
    if constexpr (RB == 2) {
#pragma unroll
        for (int half = 0; half < 2; half++) {  // this wave's 128 columns in two groups of 64
            const f32x16 pair[2] = {out[2 * half], out[2 * half + 1]};
            wave_rows64(pair, my_stage, w.lane, add_h1(WC * w.ch + 64 * half));
        }
    } else {
#pragma unroll
        for (int t = 0; t < NTO; t++) wave_rows32(out[t], my_stage, w.lane, add_h1(WC * w.ch + 32 * t));
    }
    }
    if (Xcn) {
        constexpr int NTC = RB;  // 128 centre-token columns over the NCH column groups
        __syncthreads();
        tile_row_scales<256, ROWS>(Hn_s, LD256, rs);
        __syncthreads();
        f32x16 cacc[NTC];
        acc_fill_bias<NTC>(cacc, bcn, 32 * NTC * w.ch, w.lane);
        gemm_acc_x<256, NTC, XD>(Hn_s + w.rb * 32 * LD256, LD256, wcn, 32, 0, NTC * w.ch, cacc, w.lane, rs + 64 * w.rb);
        store_acc<NTC>(cacc, Xcn, row0, N, D, w.rb, 32 * NTC * w.ch, w.lane);
    }
}

// k_node2w: the 64-row node update of large graphs with every weight block requested ONCE per workgroup. k_node2<2> runs
// (row block, column half) waves, so two waves stream the same weights (phase timings at 80 000 atoms, us per launch: tile
// loads 60, expansion + norm + planes 73, hidden products 234 for 50 us of MFMA work, U exchange + output product 167,
// epilogue 57). Here a wave owns a column quarter for BOTH 32-row blocks (tile.h gemm_acc_hs_rb2 / gemm_acc_x_ring_rb2): the
// value and gate tiles of a hidden chunk in one product, the K = 128 operand of the output product requested behind the
// barriers. 0.52 -> 0.455 ms per launch: the rest is the exposed latency of each phase at one workgroup per CU (133 KB of
// LDS). Same LDS layout and the same MFMA order per output element as k_node2<2> (the A/B kernel of rounds 3 - 5; its forward instantiation is no longer launched).
__global__ __launch_bounds__(NTHREADS) void k_node2w(const float* __restrict__ H, const float* __restrict__ OC,
                                                      WX wce, const float* __restrict__ bce,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, WX win,
                                                      const float* __restrict__ bin, WX wout,
                                                      const float* __restrict__ bout,
                                                      float* __restrict__ H1, float* __restrict__ VGn,
                                                      float* __restrict__ Hn, int64_t N, WX wcn,
                                                      const float* __restrict__ bcn, float* __restrict__ Xcn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 64, LDH = plane_ld(256);
    float* Hs = smem;                                                  // [64][260] h, then h1, then dead
    float* U = smem;                                                   // [64][132] SwiGLU chunk (aliases Hs)
    float* stage = smem + ROWS * LD128;                                // 4 staging tiles [32][32] (behind U, inside Hs)
    _Float16* Ph = reinterpret_cast<_Float16*>(smem + ROWS * LD256);   // [64][264] high pieces of Norm(h1)
    _Float16* Pl = Ph + ROWS * LDH;
    float* OCs = smem + ROWS * LD256;                                  // [64][132] OC tile (before the planes exist)
    float* rs = smem + ROWS * LD256 + ROWS * LDH;                      // [64][2] row scales
    const int lane = threadIdx.x & 63, ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // column quarter
    float* my_stage = stage + ch * (32 * 32);
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    load_rows_to_lds<128, ROWS>(OCs, OC, row0, N, D);
    load_rows_to_lds<256, ROWS>(Hs, H, row0, N, DN);
    __syncthreads();
    tile_row_scales<128, ROWS>(OCs, LD128, rs);
    __syncthreads();
    {   // h1 = h + center_expansion(OC): this wave's 64 columns, both row blocks
        XRing<2, 8> rc;
        xring_request(rc, wce, 16, 0, 2 * ch, lane);
        f32x16 acc[4];
        f32x16 b2[2];
        acc_fill_bias<2>(b2, bce, 64 * ch, lane);
        acc[0] = acc[2] = b2[0];
        acc[1] = acc[3] = b2[1];
        gemm_acc_x_ring_rb2<2, 8>(OCs, LD128, rc, acc, lane, 32 * LD128, rs);
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    Hs[(rb * 32 + acc_row(r, lane)) * LD256 + 64 * ch + 32 * t + (lane & 31)] += acc[rb * 2 + t][r];
    }
    __syncthreads();
    store_rows_from_lds<256, ROWS>(Hs, H1, row0, N, DN);
    __syncthreads();
    norm_rows_inplace<256, ROWS>(Hs, gamma, beta);
    __syncthreads();
    split_tile_planes<256, ROWS>(Hs, LD256, Ph, Pl);
    __syncthreads();  // Hs is dead from here on: U and the staging tiles reuse its memory
    f32x16 out[4];  // [row block][2 tiles]: 64 rows x this wave's 64 columns
    {
        f32x16 b2[2];
        acc_fill_bias<2>(b2, bout, 64 * ch, lane);
        out[0] = out[2] = b2[0];
        out[1] = out[3] = b2[1];
    }
#pragma unroll 1
    for (int hc = 0; hc < DNF / 128; hc++) {
        const int hcol0 = 128 * hc + 32 * ch;  // this wave's 32 hidden units of the chunk
        f32x16 vg[4];  // [row block][value, gate]
        {
            f32x16 bv[1], bg[1];
            acc_fill_bias<1>(bv, bin, hcol0, lane);
            acc_fill_bias<1>(bg, bin, DNF + hcol0, lane);
            vg[0] = vg[2] = bv[0];
            vg[1] = vg[3] = bg[0];
        }
        gemm_acc_hs_rb2<256, 2, 4>(Ph, Pl, LDH, win, 32, 0, hcol0 / 32, vg, lane, DNF / 32, 32 * LDH);
        if (VGn) {  // saved for the adjoint: [value | gate] pre-activations, whole float4 rows
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int q = 0; q < 2; q++)
                    wave_rows32(vg[rb * 2 + q], my_stage, lane, [&](int r, int cc, float4 v) {
                        const int64_t row = row0 + 32 * rb + r;
                        if (row < N) *reinterpret_cast<float4*>(VGn + row * (2 * DNF) + q * DNF + hcol0 + cc) = v;
                    });
        }
        XRing<2, 8> ro;  // the K = 128 operand of the output product: requested here, it arrives behind the barriers
        xring_request(ro, wout, DNF / 8, 16 * hc, 2 * ch, lane);
        __syncthreads();  // previous chunk's readers of U are done
#pragma unroll
        for (int rb = 0; rb < 2; rb++)
#pragma unroll
            for (int r = 0; r < 16; r++)  // transformer.py:42-43: value * sigmoid(gate)
                U[(rb * 32 + acc_row(r, lane)) * LD128 + 32 * ch + (lane & 31)] = vg[rb * 2][r] * sigmoidf_(vg[rb * 2 + 1][r]);
        __syncthreads();
        gemm_acc_x_ring_rb2<2, 8>(U, LD128, ro, out, lane, 32 * LD128);
    }
    float* Hn_s = smem + ROWS * LD256;  // [64][260]: over the planes, which nobody reads after the last hidden chunk
    if (Xcn) __syncthreads();
#pragma unroll
    for (int rb = 0; rb < 2; rb++)
#pragma unroll
        for (int t = 0; t < 2; t++)
            wave_rows32(out[rb * 2 + t], my_stage, lane, [&](int r, int cc, float4 v) {  // Hn = h1 + MLP
                const int64_t row = row0 + 32 * rb + r;
                const int col = 64 * ch + 32 * t;
                float4 hn = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < N) {
                    const int64_t o = row * DN + col + cc;
                    const float4 h1 = *reinterpret_cast<const float4*>(H1 + o);
                    hn = make_float4(h1.x + v.x, h1.y + v.y, h1.z + v.z, h1.w + v.w);
                    *reinterpret_cast<float4*>(Hn + o) = hn;
                }
                if (Xcn) *reinterpret_cast<float4*>(Hn_s + (rb * 32 + r) * LD256 + col + cc) = hn;
            });
    if (Xcn) {  // the next attention layer's centre tokens (k_node2): this wave's 32 of the 128 columns, both row blocks
        __syncthreads();
        tile_row_scales<256, ROWS>(Hn_s, LD256, rs);
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
            f32x16 cacc[1];
            acc_fill_bias<1>(cacc, bcn, 32 * ch, lane);
            gemm_acc_x<256, 1, 8>(Hn_s + rb * 32 * LD256, LD256, wcn, 32, 0, ch, cacc, lane, rs + 64 * rb);
            store_acc<1>(cacc, Xcn, row0, N, D, rb, 32 * ch, lane);
        }
    }
}

// ---------------------------------------------------------------------------------
// edge MLP: X2 = X1 + SwiGLU_MLP(Norm(X1)); PostLN (NORM = false, transformer.py:246-247): X2 = X1 + SwiGLU_MLP(X1) on
// already-normalised tokens, centre rows included (E = the row count)
// ---------------------------------------------------------------------------------
template <bool NORM>
__global__ __launch_bounds__(NTHREADS) void k_emlp(const float* __restrict__ X1, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta,
                                                    const float4* __restrict__ win, const float* __restrict__ bin,
                                                    const float4* __restrict__ wout, const float* __restrict__ bout,
                                                    float* __restrict__ VG, float* __restrict__ X2, int64_t E) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* A = smem;               // [64][132]
    float* U = smem + BM * LD128;  // [64][132]
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<128>(A, X1, row0, E, D);
    __syncthreads();
    if (NORM) {
        norm_rows_inplace<128>(A, gamma, beta);
        __syncthreads();
    }
    f32x16 out[2];
    acc_fill_bias<2>(out, bout, 64 * w.ch, w.lane);
#pragma unroll 1
    for (int hc = 0; hc < DFF / 128; hc++) {
        f32x16 av[2], ag[2];
        const int hcol0 = 128 * hc + 64 * w.ch;
        acc_fill_bias<2>(av, bin, hcol0, w.lane);
        acc_fill_bias<2>(ag, bin, DFF + hcol0, w.lane);
        gemm_acc<128, 2>(A + w.rb * 32 * LD128, LD128, win, 16, 0, hcol0 / 32, av, w.lane);
        gemm_acc<128, 2>(A + w.rb * 32 * LD128, LD128, win, 16, 0, (DFF + hcol0) / 32, ag, w.lane);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = w.rb * 32 + acc_row(r, w.lane);
                const int cc = 64 * w.ch + 32 * t + (w.lane & 31);
                const float v = av[t][r], gte = ag[t][r];
                if (VG && row0 + rr < E) {
                    VG[(row0 + rr) * (2 * DFF) + 128 * hc + cc] = v;
                    VG[(row0 + rr) * (2 * DFF) + DFF + 128 * hc + cc] = gte;
                }
                U[rr * LD128 + cc] = v * sigmoidf_(gte);
            }
        }
        __syncthreads();
        gemm_acc<128, 2>(U + w.rb * 32 * LD128, LD128, wout, DFF / 8, 16 * hc, 2 * w.ch, out, w.lane);
    }
    acc_foreach<2>(out, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
        const int64_t row = row0 + r;
        if (row < E) X2[row * D + c] = X1[row * D + c] + v;
    });
}

// ---------------------------------------------------------------------------------
// PostLN: Y = Norm(S) row by row (transformer.py:245,247); rows < E go to Ye, rows >= E (the centre tokens) to Yc
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void k_rownorm(const float* __restrict__ S, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ Ye,
                                                       float* __restrict__ Yc, int64_t E, int64_t R) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<128>(smem, S, row0, R, D);
    __syncthreads();
    norm_rows_inplace<128>(smem, gamma, beta);
    __syncthreads();
    for (int idx = threadIdx.x; idx < BM * 32; idx += NTHREADS) {
        const int r = idx >> 5, c = idx & 31;
        const int64_t row = row0 + r;
        if (row >= R) continue;
        const float4 v = *reinterpret_cast<const float4*>(smem + r * LD128 + 4 * c);
        if (row < E) *reinterpret_cast<float4*>(Ye + row * D + 4 * c) = v;
        else *reinterpret_cast<float4*>(Yc + (row - E) * D + 4 * c) = v;
    }
}

// ---------------------------------------------------------------------------------
// residual featuriser (backend.py:621-647): Mout[p] = 0.5 (Min[p] + e[rev[p]]); layer 0: Min = edge_embedder[species]
// ---------------------------------------------------------------------------------
__global__ void k_resmix(const float* __restrict__ Min, const float* __restrict__ emb, const int* __restrict__ sp_nbr,
                         const float* __restrict__ XF, const int* __restrict__ rev, float* __restrict__ Mout, int64_t E) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E * (D / 4)) return;
    const int64_t p = idx / (D / 4);
    const int c = (int)(idx % (D / 4));
    const float4 a = Min ? reinterpret_cast<const float4*>(Min)[idx]
                         : *reinterpret_cast<const float4*>(emb + (int64_t)sp_nbr[p] * D + 4 * c);
    const float4 b = *reinterpret_cast<const float4*>(XF + (int64_t)rev[p] * D + 4 * c);
    reinterpret_cast<float4*>(Mout)[idx] = make_float4(0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z),
                                                       0.5f * (a.w + b.w));
}

// ---------------------------------------------------------------------------------
// heads: y = w . SiLU(W2 SiLU(W0 x + b0) + b2) + b     (backend.py:171-217, 726-777)
// ---------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(NTHREADS) void k_head(const float* __restrict__ Xin, WX w0,
                                                    const float* __restrict__ b0, WX w2,
                                                    const float* __restrict__ b2, const float* __restrict__ wl,
                                                    float bl, const float* __restrict__ fc /* or null */,
                                                    float* __restrict__ ypred, float* __restrict__ yout,
                                                    int64_t R, float* __restrict__ hidden = nullptr, int ldh = 0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDK = lds_ld(K);
    float* A = smem;              // [64][K+4]
    float* S = smem + BM * LDK;   // [64][132]
    float* rs = S + BM * LD128;   // [64][2] row scales: backbone features and the hidden rows are un-normalised
    const WaveId w;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    load_rows_to_lds<K>(A, Xin, row0, R, K);
    __syncthreads();
    if (w0.h) {
        tile_row_scales<K>(A, LDK, rs);
        __syncthreads();
    }
    f32x16 acc[2];
    acc_fill_bias<2>(acc, b0, 64 * w.ch, w.lane);
    gemm_acc_x<K, 2>(A + w.rb * 32 * LDK, LDK, w0, K / 8, 0, 2 * w.ch, acc, w.lane, w0.h ? rs + 64 * w.rb : nullptr);
    acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) { S[r * LD128 + c] = siluf_(v); });
    __syncthreads();
    if (w2.h) {
        tile_row_scales<128>(S, LD128, rs);
        __syncthreads();
    }
    acc_fill_bias<2>(acc, b2, 64 * w.ch, w.lane);
    gemm_acc_x<128, 2>(S + w.rb * 32 * LD128, LD128, w2, 16, 0, 2 * w.ch, acc, w.lane, w2.h ? rs + 64 * w.rb : nullptr);
    __syncthreads();
    acc_foreach<2>(acc, w.rb, 64 * w.ch, w.lane, [&](int r, int c, float v) {
        const float a = siluf_(v);
        if (hidden && row0 + r < R) hidden[(row0 + r) * ldh + c] = a;  // last-layer features (pet_aux_outputs)
        S[r * LD128 + c] = a * wl[c];
    });
    __syncthreads();
    {
        const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
        float s = 0.f;
        for (int c = q * 4; c < 128; c += 16) {
            float4 v = *reinterpret_cast<float4*>(S + r * LD128 + c);
            s += v.x + v.y + v.z + v.w;
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q == 0 && row0 + r < R) {
            const float y = s + bl;
            if (ypred) ypred[row0 + r] = y;
            yout[row0 + r] = fc ? y * fc[row0 + r] : y;
        }
    }
}

// atomic[i] = ynode[i] + sum_{edges of i} ye   (backend.py:768-772, 476)
__global__ void k_atom_sum(const float* __restrict__ ynode, const float* __restrict__ ye,
                           const int* __restrict__ rowptr, float* __restrict__ atomic, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int p = rowptr[i]; p < rowptr[i + 1]; p++) s += ye[p];
    atomic[i] = ynode[i] + s;
}

// out[i][c] = sum_{edges p of i} fc[p] X[p][c], c < 128: one wave per atom, two columns per lane
// (the cutoff-weighted edge sums of pet/model.py:753 and :797-799)
__global__ void k_edge_sum_fc(const float* __restrict__ X, const float* __restrict__ fc, const int* __restrict__ rowptr,
                              float* __restrict__ out, int ldo, int n) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    float2 s = make_float2(0.f, 0.f);
    for (int p = rowptr[i]; p < rowptr[i + 1]; p++) {
        const float f = fc[p];
        const float2 x = *reinterpret_cast<const float2*>(X + (int64_t)p * D + 2 * lane);
        s.x = fmaf(f, x.x, s.x);
        s.y = fmaf(f, x.y, s.y);
    }
    *reinterpret_cast<float2*>(out + (int64_t)i * ldo + 2 * lane) = s;
}
__global__ void k_copy_rows(const float* __restrict__ X, int w, float* __restrict__ out, int ldo, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * w) return;
    out[(idx / w) * ldo + idx % w] = X[idx];
}

// ---------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------
// ---- per-layer exchange of edge tokens (one box over several ranks): staging kernels around the caller's collective
__global__ void k_rows_gather(const float* __restrict__ X, const int* __restrict__ rows, int64_t n, float* __restrict__ buf) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 per thread, D / 4 per row
    if (idx >= n * (D / 4)) return;
    const int64_t r = idx / (D / 4);
    const int c = (int)(idx % (D / 4));
    reinterpret_cast<float4*>(buf)[idx] = reinterpret_cast<const float4*>(X + (int64_t)rows[r] * D)[c];
}
// mode 0: X[rows] = buf; 1: X[rows] += buf; 2: X[rows] = 0
__global__ void k_rows_scatter(const float* __restrict__ buf, const int* __restrict__ rows, int64_t n, float* __restrict__ X,
                               int mode) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * (D / 4)) return;
    const int64_t r = idx / (D / 4);
    const int c = (int)(idx % (D / 4));
    float4* dst = reinterpret_cast<float4*>(X + (int64_t)rows[r] * D) + c;
    if (mode == 2) { *dst = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const float4 v = reinterpret_cast<const float4*>(buf)[idx];
    if (mode == 0) *dst = v;
    else { float4 o = *dst; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; *dst = o; }
}
// forward: XF[ghost rows] <- the owners' XF[export rows]   (before the combination stage reads e[rev])
int exchange_forward(const Graph& g, float* XF, int layer, hipStream_t st) {
    if (!g.x_fn) return PET_OK;
    if (g.n_export > 0)
        k_rows_gather<<<cdiv(g.n_export * (D / 4), 256), 256, 0, st>>>(XF, g.x_export, g.n_export, g.x_export_buf);
    PET_REQUIRE(g.x_fn(g.x_user, 0, layer) == 0, PET_ERR_ARGUMENT, "the exchange callback failed (forward)");
    if (g.n_ghost > 0)
        k_rows_scatter<<<cdiv(g.n_ghost * (D / 4), 256), 256, 0, st>>>(g.x_ghost_buf, g.x_ghost, g.n_ghost, XF, 0);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}
// reverse: the adjoint that landed on ghost rows belongs to their owners: send it, zero it here, add what arrives to the
// export rows
int exchange_backward(const Graph& g, float* dXF, int layer, hipStream_t st) {
    if (!g.x_fn) return PET_OK;
    if (g.n_ghost > 0) {
        k_rows_gather<<<cdiv(g.n_ghost * (D / 4), 256), 256, 0, st>>>(dXF, g.x_ghost, g.n_ghost, g.x_ghost_buf);
        k_rows_scatter<<<cdiv(g.n_ghost * (D / 4), 256), 256, 0, st>>>(nullptr, g.x_ghost, g.n_ghost, dXF, 2);
    }
    PET_REQUIRE(g.x_fn(g.x_user, 1, layer) == 0, PET_ERR_ARGUMENT, "the exchange callback failed (reverse)");
    if (g.n_export > 0)
        k_rows_scatter<<<cdiv(g.n_export * (D / 4), 256), 256, 0, st>>>(g.x_export_buf, g.x_export, g.n_export, dXF, 1);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

int64_t forward_workspace_bytes(const Model& m, int64_t n_nodes, int64_t n_edges, bool train) {
    if (m.generic() || (train && train_generic(m))) return gen_workspace_bytes(m, n_nodes, n_edges);  // gen.hip
    Workspace w;
    carve_workspace(m, n_nodes, n_edges, nullptr, w, train);
    return (int64_t)w.bytes;
}

template <int NT>
static void launch_attn_fwd(const float* QKV, const Graph& g, float* AO, float scale, hipStream_t st) {
    int waves = (int)g.n_nodes * NHEAD;
    k_attn_fwd<NT><<<cdiv(waves, 4), 256, 0, st>>>(QKV, g.rowptr, g.fc, AO, g.n_edges, (int)g.n_nodes, scale);
}

int attn_tiles(const Graph& g) { return (g.max_nbr + 1 + 15) / 16; }
bool use_generic(const Model& m, const Graph& g) { return m.generic() || attn_tiles(g) > 8; }
// Which layout the last forward of THIS graph left in a workspace: a training forward of a PostLN / residual model of the
// compiled size runs on the size-generic path while its inference forward runs on the tuned kernels, and the adjoint calls
// must follow. Recorded on the graph handle (host side, the caller owns its lifetime): no process-global table, and an
// adjoint call on a workspace this graph's forward has not written falls back to what (model, graph) say.
bool generic_workspace(const Graph& g, const void* ws) {
    const Graph::FwdRecord* r = g.fwd_record(ws);
    return r && r->generic;
}
static Graph::FwdRecord& note_workspace(const Graph& g, const void* ws, bool generic) {
    Graph::FwdRecord& r = g.fwd_record_new(ws);
    r.generic = generic;
    return r;
}

static int g_node_split = 1;  // pet_config_set("node_split", 0): one workgroup per 32-row tile in k_node2<1> / k_node_bwd2<1>
void set_node_split(int v) { g_node_split = v ? 1 : 0; }
bool node_split_on() { return g_node_split != 0; }
static int g_center_fused = 1;  // pet_config_set("center_fused", 0): the next layer's centre tokens by their own k_center launch
void set_center_fused(int v) { g_center_fused = v ? 1 : 0; }
static int g_node_planes = 1;  // k_node2 / k_node_bwd2: A tiles pre-split into fp16 planes (pet_config_set("node_planes", 0): k_node)
void set_node_planes(int v) { g_node_planes = v > 2 ? 2 : v; }
bool node_planes() { return g_node_planes != 0; }
// rows per workgroup of k_node2 / k_node_bwd2: node_planes = 2 forces 32 (the tests' route to the 32-row kernels), 1 chooses by the number of atoms
static int g_node_rows_threshold = 16384;  // measured: 1 000 / 3 000 / 10 000 atoms gain 14 / 8 / 2 %, 80 000 lose 8 % of the stage
int node_rows(int64_t N) {
    if (g_node_planes == 2) return 32;
    return N <= g_node_rows_threshold ? 32 : 64;
}

// sum_i (n_i + 1)^2 estimated from the mean neighbour count (exact value is not needed on the hot path)
double g_sum_t2(const Graph& g) {
    if (g.n_nodes == 0) return 0.0;
    const double t = (double)g.n_edges / (double)g.n_nodes + 1.0;
    return (double)g.n_nodes * t * t;
}

// both operand forms of a Linear for the LDS-tile kernels: fp16 planes when f16x3 is on and the weight was packed
static inline WX wx_fwd(const Lin& L) {
    WX w;
    w.f = L.fwd;
    if (L.fwd2) {
        const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
        w.h = reinterpret_cast<const f16x8_t*>(L.fwd2);
        w.l = w.h + n8;
    }
    return w;
}

int forward(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, int save, float* atomic,
            float* node_feat, float* edge_feat, hipStream_t st) {
    PET_REQUIRE(!m.residual() || (!node_feat && !edge_feat), PET_ERR_ARGUMENT,
                "residual featuriser: one feature pair per GNN layer, use pet_forward_layers");
    return forward_layers(m, g, ws, ws_bytes, save, atomic, &node_feat, &edge_feat, 1, st);
}

// node_feats / edge_feats: n_layers = num_readout_layers() output pointers each (entries may be null)
int forward_layers(const Model& m, const Graph& g, void* ws, int64_t ws_bytes, int save, float* atomic,
                   float* const* node_feats, float* const* edge_feats, int n_layers, hipStream_t st) {
    const bool gen = use_generic(m, g) || (save == 2 && train_generic_for(m, g));
    PET_REQUIRE(!(gen && g.x_fn), PET_ERR_UNSUPPORTED, "the per-layer exchange is built for the tuned path (default model size)");
    Graph::FwdRecord& fwd_rec = note_workspace(g, ws, gen);
    fwd_rec.save = save;
    if (gen) return gen_forward_layers(m, g, ws, ws_bytes, save, atomic, node_feats, edge_feats, n_layers, st);
    if (int rcl = graph_attention_lists(g, st)) return rcl;
    Workspace w;
    carve_workspace(m, g.n_nodes, g.n_edges, ws, w, save == 2);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "forward workspace too small");
    const bool post = m.post_ln(), res = m.residual();
    PET_REQUIRE(res ? (n_layers == m.h.num_gnn_layers || (n_layers == 1 && !node_feats[0] && !edge_feats[0])) : n_layers == 1,
                PET_ERR_ARGUMENT, "expected one feature pair per readout layer (" + std::to_string(m.num_readout_layers()) + ")");
    PET_REQUIRE(!(atomic && res), PET_ERR_UNSUPPORTED,
                "the fused head reads one readout layer; with the residual featuriser use pet_forward_layers and "
                "pet_predict per readout layer");
    PET_REQUIRE(save != 2 || m.trainable(), PET_ERR_UNSUPPORTED,
                "training is built for transformer_type=PreLN, featurizer_type=feedforward only");
    const int64_t N = g.n_nodes, E = g.n_edges, R = E + N;
    if (N == 0) return PET_OK;
    const int nt = attn_tiles(g);
    const float scale = 1.0f / (sqrtf((float)HD) * m.h.attention_temperature);
    const int gE = cdiv(E, BM), gN = cdiv(N, BM), gR = cdiv(R, BM);
    const size_t lds1 = BM * LD128 * 4, lds2 = 2 * BM * LD128 * 4;
    const size_t lds_c = lds2 + BM * 20 + BM * 8;
    const double fE = (double)E, fN = (double)N, fR = (double)R;
    const bool trr = use_trr();
    const bool trr_l = trr && m.plain_layers();  // the TRR transformer-layer kernels are PreLN (RMSNorm or LayerNorm)
    // attention: 4 T^2 d FLOPs per atom per layer; T^2 summed on the host side of the graph
    const double attn_flops = 4.0 * D * g_sum_t2(g);
    const int L = m.h.num_gnn_layers, AL = m.h.num_attention_layers;

    allow_big_lds(k_center, BM * LD256 * 4 + BM * 8);
    allow_big_lds(k_node, (BM * LD256 + BM * LD128) * 4 + BM * 8);
    allow_big_lds(k_head<256>, (BM * LD256 + BM * LD128) * 4 + BM * 8);
    SideStream ss = side_stream();
    if (post) ss.enabled = false;  // PostLN: the node update needs the MLP output of the centre token, one chain
    const hipStream_t s2 = ss.stream(st);  // node-feature chain
    bool side_busy = false;
    auto launch_center = [&](int gi, int a) {
        const AttnLayerW& A = m.gnn[gi].attn[a];
        AttnBufs& Ab = w.gnn[gi].attn[a];
        if (a == 0 && res && gi > 0)  // backend.py:617: every GNN layer starts from its own node embedding
            k_node_embed<<<cdiv(N * (DN / 4), 256), 256, 0, s2>>>(g.sp, m.node_embs[gi], w.gnn[gi].Hin, (int)N);
        ProfScope ps("center", s2, fN * 2.0 * DN * D);
        if (save != 2 && center_s(A.cc, Ab.H, Ab.X + E * D, N, s2)) return;  // large graphs (pet_center_s.hip)
        k_center<<<gN, NTHREADS, BM * LD256 * 4 + BM * 8, s2>>>(Ab.H, wx_fwd(A.cc), A.cc.b, Ab.X + E * D, N);
    };
    k_node_embed<<<cdiv(N * (DN / 4), 256), 256, 0, st>>>(g.sp, m.node_emb, w.H0, (int)N);
    const bool conditioned = m.h.system_conditioning != 0;
    if (conditioned) {
        PET_REQUIRE(g.cond_charge && g.n_cond_systems >= 1 && g.n_cond_systems <= N, PET_ERR_ARGUMENT,
                    "system_conditioning: call pet_graph_set_conditioning (charge, spin multiplicity, system indices) first");
        k_system_cond<<<(int)g.n_cond_systems, DN, 0, st>>>(g.cond_charge, g.cond_spin, m.cond_qe, m.cond_se, m.cond_w0,
                                                             m.cond_b0, m.cond_w2, m.cond_b2, w.cond, m.h.max_charge,
                                                             m.h.max_spin_multiplicity);
    }
    ss.fork(st);
    launch_center(0, 0);
    side_busy = true;
    bool node_cnt_zeroed = false;  // k_node2 SPLIT: the arrival counters are zeroed once per forward, then reset themselves
    for (int gi = 0; gi < L; gi++) {
        const GnnLayerW& G = m.gnn[gi];
        GnnBufs& B = w.gnn[gi];
        const float* Min = gi == 0 ? nullptr : w.gnn[gi - 1].Mout;
        if (E > 0) {
            ProfScope ps("compress", st, fE * 2.0 * (D * D * (gi == 0 ? 3 : 4) + 4 * D));
            if (trr && trr_compress(gi == 0, g, G, Min, save == 0 ? nullptr : B.a0, B.attn[0].X, E, st)) {  // (no adjoint follows: the pre-activation is not stored)
                // TRR kernel on f16x3 (pet_trr.hip)
            } else if (gi == 0)
                k_compress<true><<<gE, NTHREADS, lds1 + BM * 20 + BM * 8, st>>>(g.geo, g.sp_nbr, G.wc, G.tbl, nullptr, WX(),
                                                                        wx_fwd(G.compress2), G.compress2.b, B.a0,
                                                                        B.attn[0].X, E);
            else
                k_compress<false><<<gE, NTHREADS, lds_c, st>>>(g.geo, g.sp_nbr, G.wc, G.tbl, Min,
                                                               wx_fwd(G.compress0_msg), wx_fwd(G.compress2), G.compress2.b,
                                                               B.a0, B.attn[0].X, E);
        }
        for (int a = 0; a < AL; a++) {
            const AttnLayerW& A = G.attn[a];
            AttnBufs& Ab = B.attn[a];
            float* Xnext = (a + 1 < AL) ? B.attn[a + 1].X : B.XF;
            if (side_busy) {  // the centre rows of this layer's tokens come from the node chain
                ss.join(st);
                side_busy = false;
            }
            // the per-atom fused block (pet_ablk.hip): QKV, attention output and nothing else of this stage reach HBM.
            // Training keeps the three-kernel form (its second-order pass reads the saved QKV and AO).
            bool fused = false;
            if (trr_l && save != 2 && E > 0 && (save == 0 || ablk_bwd_on(g))) {
                ProfScope ps("attn_blk", st, fR * 2.0 * D * 4 * D + attn_flops, fR * 4.0 * 2 * D);  // X in; X1 | OC out
                fused = ablk_fwd(m, g, A, Ab.X, Ab.X1, Ab.OC, scale, st);
                if (fused && save) fwd_rec.attn_unsaved = true;  // the adjoint of this workspace must be the fused one
            }
            if (!fused) {
            {
                ProfScope ps("qkv", st, fR * 2.0 * D * 3 * D, fR * 4.0 * (D + 3 * D));  // X in, QKV out
                if (trr_l) trr_qkv(Ab.X, A.g_attn, m.layer_norm() ? A.b_attn : nullptr, A.qkv, Ab.QKV, R, st);
                else if (post) k_qkv<false><<<gR, NTHREADS, lds1, st>>>(Ab.X, nullptr, nullptr, A.qkv.fwd, A.qkv.b, Ab.QKV, R);
                else k_qkv<true><<<gR, NTHREADS, lds1, st>>>(Ab.X, A.g_attn, A.b_attn, A.qkv.fwd, A.qkv.b, Ab.QKV, R);
            }
            {
                ProfScope ps("attn_fwd", st, attn_flops, fR * 4.0 * (3 * D + D));
                if (!(trr && attn_fwd_preload(nt, Ab.QKV, g, Ab.AO, scale, st))) switch (nt) {
                    case 1: launch_attn_fwd<1>(Ab.QKV, g, Ab.AO, scale, st); break;
                    case 2: launch_attn_fwd<2>(Ab.QKV, g, Ab.AO, scale, st); break;
                    case 3: launch_attn_fwd<3>(Ab.QKV, g, Ab.AO, scale, st); break;
                    case 4: launch_attn_fwd<4>(Ab.QKV, g, Ab.AO, scale, st); break;
                    case 5: case 6: launch_attn_fwd<6>(Ab.QKV, g, Ab.AO, scale, st); break;
                    default: launch_attn_fwd<8>(Ab.QKV, g, Ab.AO, scale, st); break;
                }
            }
            {
                ProfScope ps("oproj", st, fR * 2.0 * D * D, fR * 4.0 * 3 * D);  // AO, X in; X1 (| OC) out
                if (trr_l) trr_oproj(Ab.AO, Ab.X, A.out, Ab.X1, Ab.OC, E, R, st);
                else if (post) k_oproj<true><<<gR, NTHREADS, lds1, st>>>(Ab.AO, Ab.X, A.out.fwd, A.out.b, Ab.X1, nullptr, E, R);
                else k_oproj<false><<<gR, NTHREADS, lds1, st>>>(Ab.AO, Ab.X, A.out.fwd, A.out.b, Ab.X1, Ab.OC, E, R);
            }
            }
            if (post) {
                // transformer.py:245-247 on every token (edges and centre): norm_attention, + MLP, norm_mlp; the edge rows
                // of the result are the next layer's tokens, the centre row feeds center_expansion
                ProfScope ps("emlp", st, fR * 2.0 * (D * 2 * DFF + DFF * D));
                k_rownorm<<<gR, NTHREADS, lds1, st>>>(Ab.X1, A.g_attn, A.b_attn, w.T1, w.T1 + E * D, E, R);
                k_emlp<false><<<gR, NTHREADS, lds2, st>>>(w.T1, nullptr, nullptr, A.mlp_in.fwd, A.mlp_in.b, A.mlp_out.fwd,
                                                           A.mlp_out.b, Ab.VG, Ab.S2, R);
                k_rownorm<<<gR, NTHREADS, lds1, st>>>(Ab.S2, A.g_mlp, A.b_mlp, Xnext, Ab.OC, E, R);
            }
            // node chain (side stream): node update of this layer, then the centre token of the next
            ss.fork(st);
            bool center_done = false;
            {
                ProfScope ps("node", s2, fN * 2.0 * (D * DN + DN * 2 * DNF + DNF * DN));
                WX wcn;
                const float* bcn = nullptr;
                float* xcn = nullptr;
                const WX wci = wx_fwd(A.cmlp_in), wce_ = wx_fwd(A.ce), wco = wx_fwd(A.cmlp_out);
                // large graphs: three shared-ring GEMMs that can run BESIDE the edge MLP (pet_node_s.hip); its scratch lives in
                // dQKV, which only the adjoint uses
                if (node_planes() && (size_t)N * DNF <= (size_t)R * 3 * D &&
                    node_fwd_s(A, Ab.H, Ab.OC, Ab.H1, Ab.VGn, Ab.Hn, w.dQKV, N, s2)) {
                } else
                if (node_planes() && wci.h && wce_.h && wco.h) {
                    const int nr = node_rows(N);
                    const size_t lds_n2 = (size_t)nr * LD256 * 4 + (size_t)2 * nr * plane_ld(256) * 2 + nr * 8;
                    // the next layer's centre tokens in the same launch when they are center_contraction(Hn) as it leaves this
                    // kernel: not behind the conditioning add, not into a residual GNN layer (its own embedding), f16x3 weights
                    const bool has_next = a + 1 < AL || gi + 1 < L;
                    if (has_next && g_center_fused && nr == 32 && !(a + 1 == AL && (conditioned || res))) {  // (large graphs: k_center is quicker)
                        const AttnLayerW& An = a + 1 < AL ? G.attn[a + 1] : m.gnn[gi + 1].attn[0];
                        AttnBufs& Abn = a + 1 < AL ? B.attn[a + 1] : w.gnn[gi + 1].attn[0];
                        wcn = wx_fwd(An.cc);
                        if (wcn.h && Abn.H == Ab.Hn) { bcn = An.cc.b; xcn = Abn.X + E * D; center_done = true; }
                    }
                    // small graphs: the hidden chunks of a row tile on four workgroups; partial outputs and the tiles'
                    // arrival counters live in dQKV, which only the adjoint uses (k_node2, SPLIT)
                    const int nt32 = cdiv(N, 32);
                    const size_t p_floats = (size_t)(DNF / 128) * nt32 * 32 * DN;
                    const bool split = nr == 32 && node_split_on() && nt32 <= 128 && p_floats + nt32 <= (size_t)R * 3 * D;
                    if (split) {
                        int* cnt = reinterpret_cast<int*>(w.dQKV + p_floats);
                        if (!node_cnt_zeroed) PET_HIP_CHECK(hipMemsetAsync(cnt, 0, nt32 * sizeof(int), s2));
                        node_cnt_zeroed = true;
                        allow_big_lds(k_node2<1, true>, lds_n2);
                        k_node2<1, true><<<dim3(nt32, DNF / 128), NTHREADS, lds_n2, s2>>>(
                            Ab.H, Ab.OC, wce_, A.ce.b, A.g_center, A.b_center, wci, A.cmlp_in.b, wco, A.cmlp_out.b, Ab.H1, Ab.VGn,
                            Ab.Hn, N, wcn, bcn, xcn, w.dQKV, cnt);
                    } else if (nr == 32) {
                        allow_big_lds(k_node2<1>, lds_n2);
                        k_node2<1><<<cdiv(N, 32), NTHREADS, lds_n2, s2>>>(Ab.H, Ab.OC, wce_, A.ce.b, A.g_center, A.b_center, wci,
                                                                         A.cmlp_in.b, wco, A.cmlp_out.b, Ab.H1, Ab.VGn, Ab.Hn, N,
                                                                         wcn, bcn, xcn, nullptr, nullptr);
                    } else {
                        allow_big_lds(k_node2w, lds_n2);
                        k_node2w<<<gN, NTHREADS, lds_n2, s2>>>(Ab.H, Ab.OC, wce_, A.ce.b, A.g_center, A.b_center, wci,
                                                               A.cmlp_in.b, wco, A.cmlp_out.b, Ab.H1, Ab.VGn, Ab.Hn, N, wcn, bcn, xcn);
                    }
                } else
                k_node<<<gN, NTHREADS, (BM * LD256 + BM * LD128) * 4 + BM * 8, s2>>>(
                    Ab.H, Ab.OC, wx_fwd(A.ce), A.ce.b, A.g_center, A.b_center, wx_fwd(A.cmlp_in), A.cmlp_in.b,
                    wx_fwd(A.cmlp_out), A.cmlp_out.b, Ab.H1, Ab.VGn, Ab.Hn, N);
            }
            if (a + 1 == AL && conditioned)  // backend.py:543-545: the node features LEAVING the GNN layer
                k_add_cond<<<cdiv(N * (DN / 4), 256), 256, 0, s2>>>(Ab.Hn, w.cond, g.sys, g.cond_sys, (int)N);
            if (center_done) {
            } else if (a + 1 < AL) launch_center(gi, a + 1);
            else if (gi + 1 < L) launch_center(gi + 1, 0);
            side_busy = true;
            if (E > 0 && !post) {
                const bool vg_out = save != 0 && !(trr_l && save == 1 && emlp_recompute_on(A.mlp_in, A.mlp_out, E));
                ProfScope ps("emlp", st, fE * 2.0 * (D * 2 * DFF + DFF * D), fE * 4.0 * (2 * D + (vg_out ? 2 * DFF : 0)));  // X1 in; X2 (and VG, when it is saved) out
                if (trr_l) {
                    float* vg = save == 0 ? nullptr : Ab.VG;  // [v; g] is stored for the adjoint unless none follows ...
                    if (save == 1 && emlp_recompute_on(A.mlp_in, A.mlp_out, E)) {  // ... or the adjoint recomputes it (noted on the graph)
                        vg = nullptr;
                        fwd_rec.emlp_unsaved = true;
                    }
                    trr_emlp(Ab.X1, A.g_mlp, m.layer_norm() ? A.b_mlp : nullptr, A.mlp_in, A.mlp_out, vg, Xnext, E, st);
                }
                else k_emlp<true><<<gE, NTHREADS, lds2, st>>>(Ab.X1, A.g_mlp, A.b_mlp, A.mlp_in.fwd, A.mlp_in.b, A.mlp_out.fwd,
                                                              A.mlp_out.b, Ab.VG, Xnext, E);
            }
        }
        if (g.x_fn) {   // one box over several ranks: the transformer outputs of foreign centres arrive from their owners
            PET_REQUIRE(m.plain() && save != 2, PET_ERR_UNSUPPORTED,
                        "the per-layer exchange is built for PreLN + feedforward models, inference and forces");
            int rc = exchange_forward(g, B.XF, gi, st);
            if (rc) return rc;
        }
        if (E > 0 && res) {
            if (gi + 1 < L) {  // the messages of the next layer (backend.py:640-647); the last layer's are never read
                ProfScope ps("comb", st, 0.0, fE * 4.0 * 3 * D);
                k_resmix<<<cdiv(E * (D / 4), 256), 256, 0, st>>>(Min, m.edge_emb, g.sp_nbr, B.XF, g.rev, B.Mout, E);
            }
        } else if (E > 0) {
            ProfScope ps("comb", st, fE * 2.0 * (2 * D * 2 * D + 2 * D * D), fE * 4.0 * (3 * D + 2 * D + D));  // e, e[rev], M in; CA, M out
            // the software-pipelined TRR kernel (pet_comb.hip) is the one implementation of this stage
            PET_REQUIRE(trr_comb(gi == 0, B.XF, g, G, Min, m.edge_emb, B.CA, B.LNS, B.Mout, E, st), PET_ERR_ARGUMENT,
                        "combination stage: the split weight planes are missing (pet_model_finalize)");
        }
    }
    const GnnBufs& last = w.gnn.back();
    if (atomic) {
        PET_REQUIRE(m.has_fused_head, PET_ERR_ARGUMENT,
                    "pet_forward with d_atomic needs the fused single-property target (keys 'node_heads.@...'); use "
                    "pet_predict for other heads");
    }
    if (atomic) {
        ProfScope ps("head_node", s2, fN * 2.0 * (DN * DH + DH * DH + DH));
        k_head<256><<<gN, NTHREADS, (BM * LD256 + BM * LD128) * 4 + BM * 8, s2>>>(
            last.Hout, wx_fwd(m.nh0), m.nh0.b, wx_fwd(m.nh2), m.nh2.b, m.nll_w, m.nll_b, nullptr, nullptr, w.ynode, N);
    }
    if (atomic && E > 0) {
        ProfScope ps("head_edge", st, fE * 2.0 * (D * DH + DH * DH + DH));
        if (!(trr && trr_head_edge(m, last.Mout, g.fc, w.ypred_e, w.ye, E, st)))
        k_head<128><<<gE, NTHREADS, lds2 + BM * 8, st>>>(last.Mout, wx_fwd(m.eh0), m.eh0.b, wx_fwd(m.eh2), m.eh2.b, m.ell_w,
                                                m.ell_b, g.fc, w.ypred_e, w.ye, E);
    }
    ss.join(st);
    if (atomic) k_atom_sum<<<cdiv(N, 256), 256, 0, st>>>(w.ynode, w.ye, g.rowptr, atomic, (int)N);
    for (int l = 0; l < n_layers; l++) {  // backend.py:585-586 / :621-625: (node, edge) features of every readout layer
        const GnnBufs& Bl = res ? w.gnn[l] : last;
        if (node_feats[l])
            PET_HIP_CHECK(hipMemcpyAsync(node_feats[l], Bl.Hout, N * DN * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (edge_feats[l] && E > 0)
            PET_HIP_CHECK(hipMemcpyAsync(edge_feats[l], res ? Bl.XF : Bl.Mout, E * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// Auxiliary per-atom outputs (pet/model.py:730-875) from the backbone features pet_forward returned:
//   feature    [N, DN + D]  = [ node features | sum_e fc_e edge features ]
//   last_layer [N, 2 DH]    = [ node-head hidden | sum_e fc_e edge-head hidden ]   (the inputs of the last Linear)
// scratch: E DH + max(E, N) floats, needed for last_layer
int aux_outputs(const Model& m, const Graph& g, const float* node_feat, const float* edge_feat, float* feature,
                float* last_layer, float* scratch, hipStream_t st) {
    if (m.generic()) return gen_aux_outputs(m, g, node_feat, edge_feat, feature, last_layer, scratch, st);
    const int64_t N = g.n_nodes, E = g.n_edges;
    if (N == 0) return PET_OK;
    const int gN = (int)cdiv(N, BM), gE = (int)cdiv(E, BM);
    if (feature) {
        k_copy_rows<<<cdiv(N * DN, 256), 256, 0, st>>>(node_feat, DN, feature, DN + D, N);
        k_edge_sum_fc<<<cdiv(N, 4), 256, 0, st>>>(edge_feat, g.fc, g.rowptr, feature + DN, DN + D, (int)N);
    }
    if (last_layer) {
        allow_big_lds(k_head<256>, (BM * LD256 + BM * LD128) * 4 + BM * 8);
        const size_t lds2 = (size_t)(BM * LD128 * 2) * 4 + BM * 8;
        PET_REQUIRE(scratch, PET_ERR_ARGUMENT, "last-layer features need the scratch buffer");
        float* hid_e = scratch;             // [E, DH] edge-head hidden rows
        float* ytmp = scratch + E * DH;     // [max(E, N)] the heads' scalar predictions, not wanted here
        k_head<256><<<gN, NTHREADS, (BM * LD256 + BM * LD128) * 4 + BM * 8, st>>>(
            node_feat, wx_fwd(m.nh0), m.nh0.b, wx_fwd(m.nh2), m.nh2.b, m.nll_w, m.nll_b, nullptr, nullptr,
            ytmp, N, last_layer, 2 * DH);
        if (E > 0)
            k_head<128><<<gE, NTHREADS, lds2, st>>>(edge_feat, wx_fwd(m.eh0), m.eh0.b, wx_fwd(m.eh2), m.eh2.b,
                                                    m.ell_w, m.ell_b, nullptr, nullptr, ytmp, E, hid_e, DH);
        k_edge_sum_fc<<<cdiv(N, 4), 256, 0, st>>>(hid_e, g.fc, g.rowptr, last_layer + DH, 2 * DH, (int)N);
    }
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

// ---------------------------------------------------------------------------------
// PETBackend.predict as a function of its arguments (backend.py:420-494): heads of ONE (target, readout layer) on the
// GIVEN node / edge features, the last layers of ONE block with P properties, cutoff-weighted edge sum:
//   atomic[i][p] = Wn[p] . hn_i + bn[p] + We[p] . (sum_e fc_e he_e) + be[p] sum_e fc_e        (backend.py:726-777)
// (the edge sum commutes with the last Linear, so it is taken on the 128 hidden columns once, not on P outputs).
// ---------------------------------------------------------------------------------
__global__ void k_fc_sum(const float* __restrict__ fc, const int* __restrict__ rowptr, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int p = rowptr[i]; p < rowptr[i + 1]; p++) s += fc[p];
    out[i] = s;
}
// one wave per atom: lane l owns hidden columns 2 l, 2 l + 1 of both halves; P outputs by wave reductions
__global__ void k_last_layers(const float* __restrict__ hn, const float* __restrict__ hes, const float* __restrict__ csum,
                              const float* __restrict__ nw, const float* __restrict__ nb, const float* __restrict__ ew,
                              const float* __restrict__ eb, int P, float* __restrict__ atomic, int n) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const float2 a = *reinterpret_cast<const float2*>(hn + (int64_t)i * DH + 2 * lane);
    const float2 b = *reinterpret_cast<const float2*>(hes + (int64_t)i * DH + 2 * lane);
    const float cs = csum[i];
    for (int p = 0; p < P; p++) {
        const float2 wn = *reinterpret_cast<const float2*>(nw + (int64_t)p * DH + 2 * lane);
        const float2 we = *reinterpret_cast<const float2*>(ew + (int64_t)p * DH + 2 * lane);
        float v = a.x * wn.x + a.y * wn.y + b.x * we.x + b.y * we.y;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) atomic[(int64_t)i * P + p] = v + nb[p] + eb[p] * cs;
    }
}

// floats: node hidden [N, DH] | edge hidden [E, DH] | edge sums [N, DH] | fc sums [N] | scalar scratch [max(E, N)]
// (+ the adjoint's per-atom rows Gn, Ge [N, DH] each and gb [N])
int64_t predict_scratch_floats(int64_t N, int64_t E) {
    const int64_t Na = N > 0 ? N : 1, Ea = E > 0 ? E : 1;
    return 4 * Na * DH + Ea * DH + 2 * Na + (Ea > Na ? Ea : Na) + 1024;
}

int predict(const Model& m, const Graph& g, const HeadW& H, const LastW& Lw, const float* node_feat, const float* edge_feat,
            const float* fc, float* atomic, float* node_hidden, float* edge_hidden, float* scratch, hipStream_t st) {
    if (m.generic()) return gen_predict(m, g, H, Lw, node_feat, edge_feat, fc, atomic, node_hidden, edge_hidden, st);
    const int64_t N = g.n_nodes, E = g.n_edges;
    if (N == 0) return PET_OK;
    const int64_t Ea = E > 0 ? E : 1;
    float* hid_n = node_hidden ? node_hidden : scratch;
    float* hid_e = edge_hidden ? edge_hidden : scratch + N * DH;
    float* sums = scratch + N * DH + Ea * DH;
    float* csum = sums + N * DH;
    float* ytmp = csum + N;
    if (!fc) fc = g.fc;
    const int gN = (int)cdiv(N, BM), gE = (int)cdiv(E, BM);
    allow_big_lds(k_head<256>, (BM * LD256 + BM * LD128) * 4 + BM * 8);
    const size_t lds2 = (size_t)(BM * LD128 * 2) * 4 + BM * 8;
    // the heads' own dot-product output is not wanted here (P properties follow): any DH-vector serves as `wl`
    k_head<256><<<gN, NTHREADS, (BM * LD256 + BM * LD128) * 4 + BM * 8, st>>>(
        node_feat, wx_fwd(H.nh0), H.nh0.b, wx_fwd(H.nh2), H.nh2.b, Lw.nw, 0.f, nullptr, nullptr, ytmp, N, hid_n, DH);
    if (E > 0)
        k_head<128><<<gE, NTHREADS, lds2, st>>>(edge_feat, wx_fwd(H.eh0), H.eh0.b, wx_fwd(H.eh2), H.eh2.b, Lw.ew,
                                                0.f, nullptr, nullptr, ytmp, E, hid_e, DH);
    k_edge_sum_fc<<<cdiv(N, 4), 256, 0, st>>>(hid_e, fc, g.rowptr, sums, DH, (int)N);
    k_fc_sum<<<cdiv(N, 256), 256, 0, st>>>(fc, g.rowptr, csum, (int)N);
    k_last_layers<<<cdiv(N, 4), 256, 0, st>>>(hid_n, sums, csum, Lw.nw, Lw.nb, Lw.ew, Lw.eb, Lw.P, atomic, (int)N);
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

}  // namespace pet
