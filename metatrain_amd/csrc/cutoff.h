// The two cutoff functions and the derivative of the taper, shared by the preprocessing, adjoint and training kernels (a
// header: no translation unit depends on device code of another one).
#pragma once
#include "common.h"

namespace pet {

// ----------------------------------------------------------------------------------
// cutoff functions (pet/modules/utilities.py:4-39)
// ----------------------------------------------------------------------------------
__device__ __forceinline__ float cutoff_value(float d, float rc, float width, int fn) {
    float s = (d - (rc - width)) / width;
    if (fn == PET_CUTOFF_BUMP) {
        s = fminf(fmaxf(s, 1e-6f), 1.0f - 1e-6f);
        return 0.5f * (1.0f + tanhf(1.0f / tanf(3.14159274f * s)));
    }
    s = fminf(fmaxf(s, 0.0f), 1.0f);
    return 0.5f * (1.0f + cosf(3.14159274f * s));
}

// d fc / d d0 (zero outside the taper because of the clamp)
__device__ __forceinline__ float cutoff_deriv(float d, float rc, float width, int fn) {
    float s = (d - (rc - width)) / width;
    if (fn == PET_CUTOFF_BUMP) {
        if (!(s >= 1e-6f && s <= 1.0f - 1e-6f)) return 0.0f;
        float x = 3.14159274f * s;
        float sn = sinf(x), cs = cosf(x);
        float t = tanhf(cs / sn);
        // d/ds [0.5 (1 + tanh(cot x))] = 0.5 (1 - t^2) * (-pi / sin^2 x)
        return 0.5f * (1.0f - t * t) * (-3.14159274f / (sn * sn)) / width;
    }
    if (!(s >= 0.0f && s <= 1.0f)) return 0.0f;
    return -0.5f * 3.14159274f * sinf(3.14159274f * s) / width;
}

__device__ __forceinline__ float cutoff_deriv_dev(float d, float rc, float width, int fn) { return cutoff_deriv(d, rc, width, fn); }

}  // namespace pet
