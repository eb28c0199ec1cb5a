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
//   dvg = swiglu'(saved vg) (dHn Wout)         k_rowgemm_s_k2<0, 1>  (256 -> 512, the SwiGLU adjoint in the epilogue)
//   dH1 = dHn + norm^T(dvg Win; h1)            k_rowgemm_s_n2<1>     (1024 -> 256, the norm adjoint and the residual in the epilogue)
// (the first form of this chain ran the two row-wise steps as kernels of their own: four links instead of two, 42.98 against 42.82 ms)
bool node_bwd_s(const AttnLayerW& A, const float* dHn, const float* H1, const float* VGn, float* dH1, float* tmp, int64_t N, bool ln,
                hipStream_t st) {
    if (!emlp_s_serves(N) || !A.cmlp_in.bwd2s || !A.cmlp_out.bwd2s || N <= 0) return false;
    float* dVG = tmp;  // [N, 2 DNF]
    // the SwiGLU adjoint is the first GEMM's epilogue, the norm adjoint and the residual the second one's (so_rows_s.hip): two
    // launches in the dependent chain beside the edge kernels instead of four
    if (!rowgemm_s_swiglu_bwd(st, dHn, A.cmlp_out.bwd2s, VGn, dVG, DNF, N)) return false;
    rowgemm_s_norm_bwd(st, dVG, 2 * DNF, A.cmlp_in.bwd2s, H1, A.g_center, ln ? 1 : 0, dHn, dH1, N);
    return true;
}

}  // namespace pet
