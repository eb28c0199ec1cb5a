// Optimizer step of the training row (pet/trainer.py:463-467):
//   clip_grad_norm_(parameters, max_norm)  ->  Adam / AdamW update  ->  re-pack the weights.
// HBM-bound elementwise work over 2.9 M fp32 parameters (grad + m + v + p = 46 MB of traffic per
// step): one launch over the flat gradient index space, parameters addressed through a segment table.
#include <math.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "model.h"

namespace pet {

// ---- global L2 norm of the flat gradient (deterministic two-stage sum, fp64 accumulate) -------
constexpr int NORM_BLOCKS = 256;

// `dup` (optional) marks the second copy of a tied parameter (activation = "SiLU": w_in is held as [W; W]): the norm is
// that of the reference's parameter list, which has W once
__global__ void k_sumsq_partial(const float* __restrict__ g, int64_t n, double* __restrict__ partial,
                                const uint8_t* __restrict__ dup) {
    __shared__ double red[256];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = min(n, i0 + per);
    double s = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256)
        if (!dup || !dup[i]) s += (double)g[i] * (double)g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// scalars[0] = total norm, scalars[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
// (torch.nn.utils.clip_grad_norm_); max_norm <= 0 disables clipping
__global__ void k_clip_coef(const double* __restrict__ partial, int nb, float max_norm, float* __restrict__ scalars) {
    if (threadIdx.x || blockIdx.x) return;
    double s = 0.0;
    for (int i = 0; i < nb; i++) s += partial[i];
    const float norm = (float)sqrt(s);
    float coef = 1.0f;
    if (max_norm > 0.f) {
        coef = max_norm / (norm + 1e-6f);
        if (coef > 1.0f) coef = 1.0f;
    }
    scalars[0] = norm;
    scalars[1] = coef;
}

// torch.optim.Adam (weight_decay < 0: none) / AdamW (decoupled decay), fp32, bias-corrected
__global__ void k_adam(float* const* __restrict__ seg_ptr, const int64_t* __restrict__ seg_off, int n_seg,
                       float* __restrict__ g, float* __restrict__ mom, float* __restrict__ var, int64_t n,
                       const float* __restrict__ scalars, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float bc1, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = n_seg;  // last segment with seg_off[s] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    float* p = seg_ptr[lo] + (i - seg_off[lo]);
    const float grad = g[i] * scalars[1];
    g[i] = grad;  // clip_grad_norm_ rescales .grad in place
    float w = *p;
    if (weight_decay >= 0.f) w *= 1.0f - lr * weight_decay;
    const float m1 = beta1 * mom[i] + (1.0f - beta1) * grad;
    const float v1 = beta2 * var[i] + (1.0f - beta2) * grad * grad;
    mom[i] = m1;
    var[i] = v1;
    const float denom = sqrtf(v1) / bc2_sqrt + eps;
    *p = w - (lr / bc1) * (m1 / denom);
}

// tied halves: the gradient of W used as both the value and the gate projection is the sum of the two halves' slots;
// both slots get that sum, so that the two copies receive the same Adam update and stay equal
__global__ void k_tie_grads(float* __restrict__ g, const int64_t* __restrict__ ties /*[n_ties][2] offset, half*/, int n_ties) {
    const int t = blockIdx.y;
    if (t >= n_ties) return;
    const int64_t off = ties[2 * t], half = ties[2 * t + 1];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const float s = g[off + i] + g[off + half + i];
    g[off + i] = s;
    g[off + half + i] = s;
}

int tie_halves(Model& m, const std::string& key) {
    const auto it = m.grad_off.find(key);
    PET_REQUIRE(it != m.grad_off.end(), PET_ERR_ARGUMENT, "unknown parameter '" + key + "'");
    const int64_t numel = m.raw.at(key).second;
    PET_REQUIRE(numel % 2 == 0, PET_ERR_ARGUMENT, "a tied parameter has an even number of elements");
    for (size_t k = 0; k + 1 < m.ties.size(); k += 2)
        if (m.ties[k] == it->second) return PET_OK;  // already registered
    m.ties.push_back(it->second);
    m.ties.push_back(numel / 2);
    m.ties_dirty = true;
    return PET_OK;
}

static int ensure_ties(Model& m, hipStream_t st) {
    if (!m.ties_dirty) return PET_OK;
    int rc;
    if ((rc = dev_alloc(m, (void**)&m.d_ties, m.ties.size() * sizeof(int64_t)))) return rc;
    if (!m.d_dup && (rc = dev_alloc(m, (void**)&m.d_dup, m.n_params))) return rc;
    std::vector<uint8_t> dup(m.n_params, 0);
    int64_t max_half = 0;
    for (size_t k = 0; k + 1 < m.ties.size(); k += 2) {
        for (int64_t i = 0; i < m.ties[k + 1]; i++) dup[m.ties[k] + m.ties[k + 1] + i] = 1;
        max_half = std::max(max_half, m.ties[k + 1]);
    }
    m.max_tie_half = max_half;
    PET_HIP_CHECK(hipMemcpyAsync(m.d_ties, m.ties.data(), m.ties.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    PET_HIP_CHECK(hipMemcpyAsync(m.d_dup, dup.data(), dup.size(), hipMemcpyHostToDevice, st));
    PET_HIP_CHECK(hipStreamSynchronize(st));
    m.ties_dirty = false;
    return PET_OK;
}

// Adam's first / second moments as flat buffers in upload order (the layout of pet_model_flat_grad): direction 0
// copies them out, 1 copies them in (checkpoint / resume of the native step; pet/trainer.py:697-717 saves
// optimizer_state_dict)
int optimizer_state(Model& m, float* d_m, float* d_v, int64_t numel, int direction, hipStream_t st);

static int ensure_state(Model& m, hipStream_t st) {
    if (m.adam_m) return PET_OK;
    int rc;
    const size_t bytes = m.n_params * sizeof(float);
    if ((rc = dev_alloc(m, (void**)&m.adam_m, bytes))) return rc;
    if ((rc = dev_alloc(m, (void**)&m.adam_v, bytes))) return rc;
    PET_HIP_CHECK(hipMemsetAsync(m.adam_m, 0, bytes, st));
    PET_HIP_CHECK(hipMemsetAsync(m.adam_v, 0, bytes, st));
    // segment table sorted by flat offset
    std::vector<std::pair<int64_t, float*>> segs;
    for (const auto& kv : m.grad_off) segs.push_back({kv.second, m.raw.at(kv.first).first});
    std::sort(segs.begin(), segs.end());
    m.n_seg = (int)segs.size();
    std::vector<float*> ptrs(m.n_seg);
    std::vector<int64_t> offs(m.n_seg + 1);
    for (int i = 0; i < m.n_seg; i++) {
        ptrs[i] = segs[i].second;
        offs[i] = segs[i].first;
    }
    offs[m.n_seg] = m.n_params;
    if ((rc = dev_alloc(m, (void**)&m.seg_ptr, ptrs.size() * sizeof(float*)))) return rc;
    if ((rc = dev_alloc(m, (void**)&m.seg_off, offs.size() * sizeof(int64_t)))) return rc;
    if ((rc = dev_alloc(m, (void**)&m.opt_scalars, 4 * sizeof(float) + NORM_BLOCKS * sizeof(double)))) return rc;
    PET_HIP_CHECK(hipMemcpyAsync(m.seg_ptr, ptrs.data(), ptrs.size() * sizeof(float*), hipMemcpyHostToDevice, st));
    PET_HIP_CHECK(hipMemcpyAsync(m.seg_off, offs.data(), offs.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    PET_HIP_CHECK(hipStreamSynchronize(st));
    return PET_OK;
}

int adam_step(Model& m, float lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
              int64_t step, float* d_grad_norm, hipStream_t st) {
    PET_REQUIRE(m.grad_flat, PET_ERR_ARGUMENT, "pet_model_zero_grad has not been called");
    PET_REQUIRE(step >= 1, PET_ERR_ARGUMENT, "optimizer steps are counted from 1");
    int rc;
    if ((rc = ensure_state(m, st))) return rc;
    if ((rc = ensure_ties(m, st))) return rc;
    const int n_ties = (int)(m.ties.size() / 2);
    if (n_ties > 0)
        k_tie_grads<<<dim3(cdiv(m.max_tie_half, 256), n_ties), 256, 0, st>>>(m.grad_flat, m.d_ties, n_ties);
    double* partial = reinterpret_cast<double*>(m.opt_scalars + 4);
    k_sumsq_partial<<<NORM_BLOCKS, 256, 0, st>>>(m.grad_flat, m.n_params, partial, n_ties > 0 ? m.d_dup : nullptr);
    k_clip_coef<<<1, 1, 0, st>>>(partial, NORM_BLOCKS, max_grad_norm, m.opt_scalars);
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    k_adam<<<cdiv(m.n_params, 256), 256, 0, st>>>(m.seg_ptr, m.seg_off, m.n_seg, m.grad_flat, m.adam_m, m.adam_v,
                                                  m.n_params, m.opt_scalars, lr, beta1, beta2, eps, weight_decay, bc1,
                                                  bc2_sqrt);
    PET_HIP_CHECK(hipGetLastError());
    if (d_grad_norm)
        PET_HIP_CHECK(hipMemcpyAsync(d_grad_norm, m.opt_scalars, sizeof(float), hipMemcpyDeviceToDevice, st));
    return finalize(m, st);  // the packed / folded forms follow the updated raw weights
}

int optimizer_state(Model& m, float* d_m, float* d_v, int64_t numel, int direction, hipStream_t st) {
    PET_REQUIRE(numel == m.n_params, PET_ERR_ARGUMENT, "optimizer state has pet_model_num_params elements per moment");
    int rc;
    if ((rc = ensure_state(m, st))) return rc;
    const size_t bytes = m.n_params * sizeof(float);
    if (direction == 0) {
        PET_HIP_CHECK(hipMemcpyAsync(d_m, m.adam_m, bytes, hipMemcpyDeviceToDevice, st));
        PET_HIP_CHECK(hipMemcpyAsync(d_v, m.adam_v, bytes, hipMemcpyDeviceToDevice, st));
    } else {
        PET_HIP_CHECK(hipMemcpyAsync(m.adam_m, d_m, bytes, hipMemcpyDeviceToDevice, st));
        PET_HIP_CHECK(hipMemcpyAsync(m.adam_v, d_v, bytes, hipMemcpyDeviceToDevice, st));
    }
    return PET_OK;
}

}  // namespace pet
