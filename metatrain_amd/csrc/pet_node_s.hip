// The node update of an attention layer (transformer.py:203-234: h1 = h + center_expansion(o_centre); h' = h1 + MLP(norm(h1))) for
// LARGE graphs as three row GEMMs on the shared weight ring (so_rows_s.hip) and one row-wise kernel. Round 6.
//   h1  = h + o Wce^T + b                      k_rowgemm_s        (K = 128 -> 256, addend h)
//   vg  = norm(h1) Win^T + b                   k_rowgemm_s_k2<1|2> (the norm in the GEMM's prologue; 256 -> 2 x 512, saved for the adjoint)
//   u   = value . sigmoid(gate)                k_node_swiglu
//   h'  = h1 + u Wout^T + b                    k_rowgemm_s_n2     (512 -> 256, addend h1)
// The fused kernels that do this in one launch (k_node2w, k_node2 in pet_fwd.hip) need 420 - 512 registers and 67 - 133 KB of LDS
// per workgroup: on the second stream they cannot share a CU with the edge kernels they are meant to overlap with (two 80-KB,
// 256-register workgroups fill a CU; a node workgroup needs it EMPTY), so in the step of a large batch they run behind the edge MLP
// instead of beside it and the next attention block waits for them. Every kernel of this chain fits beside one workgroup of an
// edge kernel. Same saved tensors (H1, [value | gate] pre-activations, Hn) as the fused kernels: the adjoint is unchanged.
#include "common.h"
#include "model.h"
#include "tile.h"

namespace pet {

// U = value * sigmoid(gate) (transformer.py:42-43) of [N, 2 DNF] = [value | gate]
__global__ __launch_bounds__(256) void k_node_swiglu(const float* __restrict__ VG, float* __restrict__ U, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int64_t row = i / (DNF / 4);
    const int c = (int)(i % (DNF / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(VG + row * (2 * DNF) + c);
    const float4 g = *reinterpret_cast<const float4*>(VG + row * (2 * DNF) + DNF + c);
    *reinterpret_cast<float4*>(U + row * DNF + c) =
        make_float4(v.x * sigmoidf_(g.x), v.y * sigmoidf_(g.y), v.z * sigmoidf_(g.z), v.w * sigmoidf_(g.w));
}

// (dv, dg) = (du sigmoid(g), du v sigmoid'(g)) of u = v sigmoid(g): [N, DNF] and the saved [value | gate] -> [N, 2 DNF]
__global__ __launch_bounds__(256) void k_node_swiglu_bwd(const float* __restrict__ VG, const float* __restrict__ dU,
                                                         float* __restrict__ dVG, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int64_t row = i / (DNF / 4);
    const int c = (int)(i % (DNF / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(VG + row * (2 * DNF) + c);
    const float4 g = *reinterpret_cast<const float4*>(VG + row * (2 * DNF) + DNF + c);
    const float4 d = *reinterpret_cast<const float4*>(dU + row * DNF + c);
    const float sx = sigmoidf_(g.x), sy = sigmoidf_(g.y), sz = sigmoidf_(g.z), sw = sigmoidf_(g.w);
    *reinterpret_cast<float4*>(dVG + row * (2 * DNF) + c) = make_float4(d.x * sx, d.y * sy, d.z * sz, d.w * sw);
    *reinterpret_cast<float4*>(dVG + row * (2 * DNF) + DNF + c) =
        make_float4(d.x * v.x * sx * (1.f - sx), d.y * v.y * sy * (1.f - sy), d.z * v.z * sz * (1.f - sz), d.w * v.w * sw * (1.f - sw));
}

// out = dres + (adjoint of y = norm(x) gamma (+ beta) applied to dy): RMSNorm (ln == 0; eps 2^-23) or LayerNorm (eps 1e-5) of rows of
// DN = 256; one wave per row. dx = rstd (dyh - [mean(dyh)] - xh mean(dyh xh)), dyh = dy gamma, xh = the normalised row
__global__ __launch_bounds__(256) void k_node_norm_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ gamma, int ln, const float* __restrict__ dres,
                                                       float* __restrict__ out, int64_t N) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int l = threadIdx.x & 63;
    auto wave_sum = [](float s) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        return s;
    };
    float4 v = *reinterpret_cast<const float4*>(x + row * DN + 4 * l);
    if (ln) {
        const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / DN);
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
    }
    const float rstd = rsqrtf(wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / DN) + (ln ? 1e-5f : 1.1920928955078125e-07f));
    v.x *= rstd; v.y *= rstd; v.z *= rstd; v.w *= rstd;
    const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * l);
    float4 d = *reinterpret_cast<const float4*>(dy + row * DN + 4 * l);
    d.x *= g.x; d.y *= g.y; d.z *= g.z; d.w *= g.w;
    const float m1 = ln ? wave_sum((d.x + d.y) + (d.z + d.w)) * (1.0f / DN) : 0.f;
    const float m2 = wave_sum(d.x * v.x + d.y * v.y + d.z * v.z + d.w * v.w) * (1.0f / DN);
    const float4 r = *reinterpret_cast<const float4*>(dres + row * DN + 4 * l);
    *reinterpret_cast<float4*>(out + row * DN + 4 * l) =
        make_float4(r.x + rstd * (d.x - m1 - v.x * m2), r.y + rstd * (d.y - m1 - v.y * m2), r.z + rstd * (d.z - m1 - v.z * m2),
                    r.w + rstd * (d.w - m1 - v.w * m2));
}

// false = not served (fewer atoms than the row kernels' threshold, planes missing, or pet_config_set("emlp_s", 0)); nothing has been
// launched in that case. tmp: [N, DNF] floats of scratch (the SwiGLU output)
bool node_fwd_s(const AttnLayerW& A, const float* H, const float* OC, float* H1, float* VGn, float* Hn, float* tmp, int64_t N,
                hipStream_t st) {
    if (!emlp_s_serves(N) || !A.ce.fwd2s || !A.cmlp_in.fwd2s || !A.cmlp_out.fwd2s || !VGn || N <= 0) return false;
    float* U = tmp;
    if (!rowgemm_s_ex(st, OC, D, nullptr, A.ce.fwd2s, A.ce.b, H, H1, DN, N, 0, nullptr)) return false;
    rowgemm_s_ex(st, H1, DN, A.g_center, A.cmlp_in.fwd2s, A.cmlp_in.b, nullptr, VGn, 2 * DNF, N, A.b_center ? 2 : 1, A.b_center);
    k_node_swiglu<<<(int)cdiv(N * (DNF / 4), 256), 256, 0, st>>>(VGn, U, N * (DNF / 4));
    rowgemm_s_ex(st, U, DNF, nullptr, A.cmlp_out.fwd2s, A.cmlp_out.b, H1, Hn, DN, N, 0, nullptr);
    return true;
}

// The adjoint of the same update (inference): dH1 = dHn + norm^T(W_in^T swiglu'(W_out^T dHn)). tmp: [N, 3 DNF] floats of scratch.
//   du  = dHn Wout                             k_rowgemm_s_k2<0>  (256 -> 512)
//   dvg = swiglu'(saved vg) du                 k_node_swiglu_bwd
//   dy  = dvg Win                              k_rowgemm_s_n2     (1024 -> 256)
//   dH1 = dHn + norm^T(dy; h1)                 k_node_norm_bwd
bool node_bwd_s(const AttnLayerW& A, const float* dHn, const float* H1, const float* VGn, float* dH1, float* tmp, int64_t N, bool ln,
                hipStream_t st) {
    if (!emlp_s_serves(N) || !A.cmlp_in.bwd2s || !A.cmlp_out.bwd2s || N <= 0) return false;
    float* dU = tmp;            // [N, DNF]; dead after the SwiGLU adjoint: dy reuses it
    float* dVG = tmp + N * DNF;  // [N, 2 DNF]
    float* dY = tmp;
    if (!rowgemm_s_ex(st, dHn, DN, nullptr, A.cmlp_out.bwd2s, nullptr, nullptr, dU, DNF, N, 0, nullptr)) return false;
    k_node_swiglu_bwd<<<(int)cdiv(N * (DNF / 4), 256), 256, 0, st>>>(VGn, dU, dVG, N * (DNF / 4));
    rowgemm_s_ex(st, dVG, 2 * DNF, nullptr, A.cmlp_in.bwd2s, nullptr, nullptr, dY, DN, N, 0, nullptr);
    k_node_norm_bwd<<<(int)cdiv(N, 4), 256, 0, st>>>(dY, H1, A.g_center, ln ? 1 : 0, dHn, dH1, N);
    return true;
}

}  // namespace pet
