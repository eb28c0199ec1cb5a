// Shared pieces of the row kernels built like k_emlp_s (pet_emlp_s.hip) on 128 x 128 products: pet_head_s.hip (edge head),
// pet_compress_s.hip (compress stage). One-accumulator split-operand products (ablk.h), two desynchronised four-wave workgroups
// per CU (80 KB of LDS each: four 16-KB row tiles + the ring), the four waves of a workgroup sharing ONE stream of weight
// fragments through a four-slot LDS ring requested three stages ahead. A product is 16 stages of four fragments (tile pair tp,
// K block kb: tiles 2 tp, 2 tp + 1 x (h, l)), six MFMAs per wave and stage; nothing but ring requests may sit in the vmcnt
// queue between a tile's first stage and its last (vmcnt retires in order).
#pragma once
#include "ablk.h"

namespace pet {

constexpr int HS_NW = 4, HS_SLOT = 4096, HS_NSLOT = 4;

// this wave's fragment of the stage has landed (everything but its fragments of the two stages requested after it), then the
// workgroup barrier: the stage is complete for everybody, and everybody is done with the stage before it
#define HS_STAGE_SYNC()                                   \
    do {                                                  \
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");  \
        __syncthreads();                                  \
    } while (0)

// one 128 x 128 product: acc[t] (4 tiles, zero on entry) += W[tile t] planes . x planes; stages g0 .. g0 + 15 of the stream
// (req(g): this wave's request of stage g, clamped to the stream's last stage)
template <class Req>
__device__ __forceinline__ void hs_gemm_r(f32x16 (&acc)[4], const f16x8 (&xh)[8], const f16x8 (&xl)[8], int g0, Req req,
                                          const char* ring, unsigned lane16) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int g = g0 + r, tp = r >> 3, kb = r & 7;
        HS_STAGE_SYNC();
        req(g + 3);
        const char* slot = ring + (g & (HS_NSLOT - 1)) * HS_SLOT + lane16;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * t) * 1024);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * t + 1) * 1024);
            AB_MFMA3(acc[2 * tp + t], wh, wl, xh[kb], xl[kb]);
        }
    }
}
// planes of 64 x of a row fragment (already scaled by its power of two)
__device__ __forceinline__ void hs_planes(const float4 (&x)[16], f16x8 (&xh)[8], f16x8 (&xl)[8]) {
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
        const float v8[8] = {x[2 * kb].x * ABS, x[2 * kb].y * ABS, x[2 * kb].z * ABS, x[2 * kb].w * ABS,
                             x[2 * kb + 1].x * ABS, x[2 * kb + 1].y * ABS, x[2 * kb + 1].z * ABS, x[2 * kb + 1].w * ABS};
        ab_split8(v8, xh[kb], xl[kb]);
    }
}
// the value of a 128-vector at the feature of accumulator register 4 j + i of tile t, read through the SCALAR cache (wave-uniform
// addresses, both halves of the column group, selected by lane half: a vector load would queue behind the ring requests)
__device__ __forceinline__ float hs_vec(const float* __restrict__ v, int t, int j, int i, int h) {
    const float lo = v[32 * t + 8 * j + i], hi = v[32 * t + 8 * j + 4 + i];
    return h ? hi : lo;
}

}  // namespace pet
