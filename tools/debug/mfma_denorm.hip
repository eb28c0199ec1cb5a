// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs? (unscaled low planes in pet_ablk.hip would rely on it)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val, float b_val) {
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
    f32x16 c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)((_Float16)(a_val * 0.5f)); }
}
int main() {
    float* d;
    (void)hipMalloc(&d, 64);
    float h[4];
    const float vals[] = {1.0f, 6.103515625e-05f, 3.0517578125e-05f, 9.5367431640625e-07f, 5.9604644775390625e-08f};
    for (float v : vals) {
        k<<<1, 64>>>(d, v, 1.0f);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("a = %.10e (as fp16 %.10e): mfma sum over K=16 -> %.10e (expected %.10e)  cvt(a/2)=%.10e\n", v, h[1], h[0], 16.0 * h[1], h[2]);
        k<<<1, 64>>>(d, 1.0f, v);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("   as B operand: %.10e\n", h[0]);
    }
    return 0;
}
