// The fused attention adjoint of the 32-slot tiles, k_ablk_bwd<1, LN> (ablk_bwd.h), in a translation unit of its own: it is
// compiled WITHOUT -fgpu-rdc and with -mllvm -amdgpu-mfma-vgpr-form (build.py VGPR_FORM). At one wave per SIMD the compiler
// otherwise selects the AGPR form of every MFMA, and every product whose result feeds vector arithmetic (Q, K, V, S, dP, dS,
// the transposes ...) is copied out with 16 v_accvgpr_read: 814 of the kernel's 7 958 instructions, 571 of the 1 721 vector
// instructions of a head pair's round. In VGPR form the accumulators are ordinary registers and the AGPRs are only spill
// space: 141 copies, 317 registers instead of 395. (The 64-slot form crashes this compiler's AGPR-copy rewrite pass with the
// option on, so it stays in pet_ablk.hip.)
#include "ablk_bwd.h"

namespace pet {

void ablk_bwd1_launch(bool ln, const float* X, const float* dX1, const float* dOC, const float* gamma, const float* beta, W2 wqkv,
                      const float* bqkv, W2 wot, W2 wqkvt, const float* fc, const int4* desc, int n_list, int64_t E, float qscale,
                      float scale, float* dXin, float* dbias, hipStream_t st) {
    constexpr int WPB = 4;
    const size_t lds = (size_t)WPB * 32768 + 2 * AB_SLOT_B;
    if (ln) {
        allow_big_lds(k_ablk_bwd<1, true>, lds);
        k_ablk_bwd<1, true><<<cdiv(n_list, WPB), 64 * WPB, lds, st>>>(X, dX1, dOC, gamma, beta, wqkv, bqkv, wot, wqkvt, fc, desc, n_list,
                                                                     E, qscale, scale, dXin, dbias);
    } else {
        allow_big_lds(k_ablk_bwd<1, false>, lds);
        k_ablk_bwd<1, false><<<cdiv(n_list, WPB), 64 * WPB, lds, st>>>(X, dX1, dOC, gamma, beta, wqkv, bqkv, wot, wqkvt, fc, desc, n_list,
                                                                      E, qscale, scale, dXin, dbias);
    }
}

}  // namespace pet
