// The node update of an attention layer (transformer.py:203-234: h1 = h + center_expansion(o_centre); h' = h1 + MLP(norm(h1))) for
// LARGE graphs as a chain of generic row GEMMs on the shared weight ring (so_rows_s.hip k_rowgemm_s) with two row-wise kernels in
// between. Round 6. The fused kernels that do this in one launch (k_node2w, k_node2 in pet_fwd.hip) need 420 - 512 registers and
// 67 - 133 KB of LDS per workgroup: on the side stream they cannot share a CU with the edge kernels they are meant to overlap with
// (two 80-KB, 256-register workgroups fill a CU; a node workgroup needs it EMPTY), so in the step of a large batch they run behind
// the edge MLP instead of beside it and the next attention block waits for them (trace: 0.45 - 0.6 ms per layer). Every kernel of
// this chain fits beside one workgroup of an edge kernel. Same saved tensors (H1, [value | gate] pre-activations, Hn) as the fused
// kernels, so the adjoint is unchanged.
#include "common.h"
#include "model.h"
#include "tile.h"

namespace pet {

// y = norm(x) gamma (+ beta): RMSNorm (beta == nullptr; eps 2^-23) or torch.nn.LayerNorm (eps 1e-5) of rows of DN = 256; one wave per row
__global__ __launch_bounds__(256) void k_node_norm(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ y, int64_t N) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int l = threadIdx.x & 63;
    float4 v = *reinterpret_cast<const float4*>(x + row * DN + 4 * l);
    auto wave_sum = [](float s) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        return s;
    };
    float mean = 0.f;
    if (beta) {
        mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / DN);
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
    }
    const float ss = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    const float rstd = rsqrtf(ss * (1.0f / DN) + (beta ? 1e-5f : 1.1920928955078125e-07f));
    const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * l);
    float4 o = make_float4(v.x * rstd * g.x, v.y * rstd * g.y, v.z * rstd * g.z, v.w * rstd * g.w);
    if (beta) {
        const float4 b = *reinterpret_cast<const float4*>(beta + 4 * l);
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
    }
    *reinterpret_cast<float4*>(y + row * DN + 4 * l) = o;
}

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

// false = not served (fewer atoms than the row kernels' threshold, planes missing, or pet_config_set("emlp_s", 0)); nothing is
// launched in that case. tmp: [N, DN + DNF] floats of scratch (the normalised rows, the SwiGLU output)
bool node_fwd_s(const AttnLayerW& A, const float* H, const float* OC, float* H1, float* VGn, float* Hn, float* tmp, int64_t N,
                hipStream_t st) {
    if (!emlp_s_serves(N) || !A.ce.fwd2s || !A.cmlp_in.fwd2s || !A.cmlp_out.fwd2s || !VGn || N <= 0) return false;
    float* Y = tmp;
    float* U = tmp + N * DN;
    // h1 = h + center_expansion(o): the residual rides in the output rows
    if (hipMemcpyAsync(H1, H, N * DN * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
    if (!rowgemm_s(st, OC, D, nullptr, A.ce.fwd2s, A.ce.b, H1, DN, N, true)) return false;
    k_node_norm<<<(int)cdiv(N, 4), 256, 0, st>>>(H1, A.g_center, A.b_center, Y, N);
    rowgemm_s(st, Y, DN, nullptr, A.cmlp_in.fwd2s, A.cmlp_in.b, VGn, 2 * DNF, N, false);
    k_node_swiglu<<<(int)cdiv(N * (DNF / 4), 256), 256, 0, st>>>(VGn, U, N * (DNF / 4));
    (void)hipMemcpyAsync(Hn, H1, N * DN * sizeof(float), hipMemcpyDeviceToDevice, st);
    rowgemm_s(st, U, DNF, nullptr, A.cmlp_out.fwd2s, A.cmlp_out.b, Hn, DN, N, true);
    return true;
}

}  // namespace pet
