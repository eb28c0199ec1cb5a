// Edge/token-stream stages of PET in the transposed register-resident form (trr.h): one wave =
// 32 rows through a whole stage, no LDS, no barriers, weights streamed from L2 in fragment order.
// Same math and same saved tensors as the LDS-tile kernels in pet_fwd.hip / pet_bwd.hip (kept for
// A/B comparison, PET_HIP_TRR=0); reference line map in pet_fwd.hip.
#include "common.h"
#include "model.h"
#include "pet_ws.h"
#include "trr.h"

namespace pet {

#define TRR_PROLOGUE(NROWS)                                   \
    const RowLane L;                                          \
    const int64_t row0 = wave_row0();                         \
    if (row0 >= (NROWS)) return;                              \
    const bool valid = row0 + L.r < (NROWS);                  \
    const int64_t row = valid ? row0 + L.r : (NROWS) - 1

// ---------------------------------------------------------------------------------
// QKV = RMSNorm(X) Win^T + b
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void k_qkv_t(const float* __restrict__ X, const float* __restrict__ gamma,
                                                const float4* __restrict__ win, const float* __restrict__ bin,
                                                float* __restrict__ QKV, int64_t R) {
    TRR_PROLOGUE(R);
    float4 x[16];
    load_rowfrag<16>(x, X, row, D, L.h);
    rmsnorm_frag<16>(x, gamma, L.h);
#pragma unroll 1
    for (int c = 0; c < 6; c++) {
        f32x16 acc[2];
        acc_bias<2>(acc, bin, 64 * c, L.h);
        gemm_t<16, 2>(win, 16, 0, 2 * c, x, acc, L.lane);
        if (valid) {
            float4 y[8];
            acc_to_frag<2>(acc, y);
            store_rowfrag<8>(y, QKV + 64 * c, row, 3 * D, L.h);
        }
    }
}

// dXin = (row < E ? dX1 : 0) + RMSNorm^T(dQKV Win)
__global__ __launch_bounds__(256, 2) void k_qkv_bwd_t(const float* __restrict__ dQKV, const float* __restrict__ X,
                                                    const float* __restrict__ gamma, const float4* __restrict__ winb,
                                                    const float* __restrict__ dX1, float* __restrict__ dXin,
                                                    int64_t E, int64_t R) {
    TRR_PROLOGUE(R);
    f32x16 dn[4];
    acc_zero<4>(dn);
#pragma unroll 1
    for (int ks = 0; ks < 3; ks++) {
        float4 d[16];
        load_rowfrag<16>(d, dQKV + 128 * ks, row, 3 * D, L.h);
        gemm_t<16, 4>(winb, 48, 16 * ks, 0, d, dn, L.lane);
    }
    float4 w[16], x[16];
    acc_to_frag<4>(dn, w);
    load_rowfrag<16>(x, X, row, D, L.h);
#pragma unroll
    for (int kg = 0; kg < 16; kg++) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + 8 * kg + 4 * L.h);
        w[kg].x *= g.x; w[kg].y *= g.y; w[kg].z *= g.z; w[kg].w *= g.w;
    }
    rmsnorm_bwd_frag<16>(w, x);
    if (valid) {
        if (row < E) {
            load_rowfrag<16>(x, dX1, row, D, L.h);
#pragma unroll
            for (int kg = 0; kg < 16; kg++) {
                w[kg].x += x[kg].x; w[kg].y += x[kg].y; w[kg].z += x[kg].z; w[kg].w += x[kg].w;
            }
        }
        store_rowfrag<16>(w, dXin, row, D, L.h);
    }
}

// ---------------------------------------------------------------------------------
// output_linear (+ edge residual) and its adjoint
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void k_oproj_t(const float* __restrict__ AO, const float* __restrict__ X,
                                                  const float4* __restrict__ wo, const float* __restrict__ bo,
                                                  float* __restrict__ X1, float* __restrict__ OC, int64_t E,
                                                  int64_t R) {
    TRR_PROLOGUE(R);
    float4 a[16];
    load_rowfrag<16>(a, AO, row, D, L.h);
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
        f32x16 acc[2];
        acc_bias<2>(acc, bo, 64 * c, L.h);
        gemm_t<16, 2>(wo, 16, 0, 2 * c, a, acc, L.lane);
        if (valid) {
            float4 y[8];
            acc_to_frag<2>(acc, y);
            if (row < E) {
                float4 xr[8];
                load_rowfrag<8>(xr, X + 64 * c, row, D, L.h);
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    y[k].x += xr[k].x; y[k].y += xr[k].y; y[k].z += xr[k].z; y[k].w += xr[k].w;
                }
                store_rowfrag<8>(y, X1 + 64 * c, row, D, L.h);
            } else {
                store_rowfrag<8>(y, OC + 64 * c, row - E, D, L.h);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_oproj_bwd_t(const float* __restrict__ dX1, const float* __restrict__ dOC,
                                                      const float4* __restrict__ wob, float* __restrict__ dAO,
                                                      int64_t E, int64_t R) {
    TRR_PROLOGUE(R);
    float4 d[16];
    if (row < E) load_rowfrag<16>(d, dX1, row, D, L.h);
    else load_rowfrag<16>(d, dOC, row - E, D, L.h);
#pragma unroll 1
    for (int c = 0; c < 2; c++) {
        f32x16 acc[2];
        acc_zero<2>(acc);
        gemm_t<16, 2>(wob, 16, 0, 2 * c, d, acc, L.lane);
        if (valid) {
            float4 y[8];
            acc_to_frag<2>(acc, y);
            store_rowfrag<8>(y, dAO + 64 * c, row, D, L.h);
        }
    }
}

// ---------------------------------------------------------------------------------
// edge SwiGLU MLP: X2 = X1 + Wout (v * sig(g)) + b,  [v; g] = Win RMSNorm(X1) + b
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emlp_t(const float* __restrict__ X1, const float* __restrict__ gamma,
                                                 const float4* __restrict__ win, const float* __restrict__ bin,
                                                 const float4* __restrict__ wout, const float* __restrict__ bout,
                                                 float* __restrict__ VG, float* __restrict__ X2, int64_t E) {
    TRR_PROLOGUE(E);
    float4 x[16];
    load_rowfrag<16>(x, X1, row, D, L.h);
    rmsnorm_frag<16>(x, gamma, L.h);
    f32x16 out[4];
    acc_bias<4>(out, bout, 0, L.h);
#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        // value and gate tiles as two interleaved MFMA chains (tiles hc and DFF/32 + hc of w_in)
        f32x16 vg[2];
        {
            f32x16 v[1], g[1];
            acc_bias<1>(v, bin, 32 * hc, L.h);
            acc_bias<1>(g, bin, DFF + 32 * hc, L.h);
            vg[0] = v[0];
            vg[1] = g[0];
        }
        gemm_t<16, 2, 2>(win, 16, 0, hc, x, vg, L.lane, DFF / 32);
        float4 u[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 vv = acc_q(vg[0], q), gg = acc_q(vg[1], q);
            if (VG && valid) {
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + 32 * hc + 8 * q + 4 * L.h) = vv;
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + DFF + 32 * hc + 8 * q + 4 * L.h) = gg;
            }
            u[q] = make_float4(vv.x * sigm_(gg.x), vv.y * sigm_(gg.y), vv.z * sigm_(gg.z), vv.w * sigm_(gg.w));
        }
        gemm_t<4, 4>(wout, DFF / 8, 4 * hc, 0, u, out, L.lane);
    }
    if (valid) {
        float4 y[16], xr[16];
        acc_to_frag<4>(out, y);
        load_rowfrag<16>(xr, X1, row, D, L.h);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            y[k].x += xr[k].x; y[k].y += xr[k].y; y[k].z += xr[k].z; y[k].w += xr[k].w;
        }
        store_rowfrag<16>(y, X2, row, D, L.h);
    }
}

// bf16x6 operands of a Lin (abi.hip k_pack3): three consecutive arrays of n8 fragments
static inline W3 w3_fwd(const Lin& L) {
    const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
    const bf16x8* b = reinterpret_cast<const bf16x8*>(L.fwd3);
    W3 w; w.h = b; w.m = b + n8; w.l = b + 2 * n8;
    return w;
}
static inline W3 w3_bwd(const Lin& L) {
    const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
    const bf16x8* b = reinterpret_cast<const bf16x8*>(L.bwd3);
    W3 w; w.h = b; w.m = b + n8; w.l = b + 2 * n8;
    return w;
}

// same stage with the GEMMs on the bf16 matrix cores (bf16x6, trr.h)
__global__ __launch_bounds__(256) void k_emlp_b(const float* __restrict__ X1, const float* __restrict__ gamma, W3 win,
                                                 const float* __restrict__ bin, W3 wout,
                                                 const float* __restrict__ bout, float* __restrict__ VG,
                                                 float* __restrict__ X2, int64_t E) {
    TRR_PROLOGUE(E);
    Split3<8> xs;
    {
        float4 x[16];
        load_rowfrag<16>(x, X1, row, D, L.h);
        rmsnorm_frag<16>(x, gamma, L.h);
        split_frag<8>(x, xs);
    }
    f32x16 out[4];
    acc_bias<4>(out, bout, 0, L.h);
#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        f32x16 vg[2];
        {
            f32x16 v[1], g[1];
            acc_bias<1>(v, bin, 32 * hc, L.h);
            acc_bias<1>(g, bin, DFF + 32 * hc, L.h);
            vg[0] = v[0];
            vg[1] = g[0];
        }
        gemm_b<8, 2, 2>(win, 8, 0, hc, xs, 0, vg, L.lane, DFF / 32);
        float4 u[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 vv = acc_q(vg[0], q), gg = acc_q(vg[1], q);
            if (VG && valid) {
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + 32 * hc + 8 * q + 4 * L.h) = vv;
                *reinterpret_cast<float4*>(VG + row * (2 * DFF) + DFF + 32 * hc + 8 * q + 4 * L.h) = gg;
            }
            u[q] = make_float4(vv.x * sigm_(gg.x), vv.y * sigm_(gg.y), vv.z * sigm_(gg.z), vv.w * sigm_(gg.w));
        }
        Split3<2> us;
        split_frag<2>(u, us);
        gemm_b<2, 4, 2>(wout, DFF / 16, 2 * hc, 0, us, 0, out, L.lane);
    }
    if (valid) {
        float4 y[16], xr[16];
        acc_to_frag<4>(out, y);
        load_rowfrag<16>(xr, X1, row, D, L.h);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            y[k].x += xr[k].x; y[k].y += xr[k].y; y[k].z += xr[k].z; y[k].w += xr[k].w;
        }
        store_rowfrag<16>(y, X2, row, D, L.h);
    }
}

// dX1 = dY + RMSNorm^T( Win^T [du sig(g) ; du v sig'(g)] ),  du = Wout^T dY
template <bool TRAIN>
__global__ __launch_bounds__(256, 2) void k_emlp_bwd_t(const float* __restrict__ dY, const float* __restrict__ X1,
                                                     const float* __restrict__ VG, const float* __restrict__ gamma,
                                                     const float4* __restrict__ woutb, const float4* __restrict__ winb,
                                                     float* __restrict__ dX1, int64_t E, float* __restrict__ t_dvg) {
    TRR_PROLOGUE(E);
    float4 dy[16];
    load_rowfrag<16>(dy, dY, row, D, L.h);
    f32x16 dn[4];
    acc_zero<4>(dn);
#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        f32x16 du[1];
        acc_zero<1>(du);
        gemm_t<16, 1, 4>(woutb, 16, 0, hc, dy, du, L.lane);  // du[:, chunk] = dY Wout[:, chunk]
        float4 dv[4], dg[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 vv = *reinterpret_cast<const float4*>(VG + row * (2 * DFF) + 32 * hc + 8 * q + 4 * L.h);
            const float4 gg = *reinterpret_cast<const float4*>(VG + row * (2 * DFF) + DFF + 32 * hc + 8 * q + 4 * L.h);
            const float4 d = acc_q(du[0], q);
            const float sx = sigm_(gg.x), sy = sigm_(gg.y), sz = sigm_(gg.z), sw = sigm_(gg.w);
            dv[q] = make_float4(d.x * sx, d.y * sy, d.z * sz, d.w * sw);
            dg[q] = make_float4(d.x * vv.x * sx * (1.f - sx), d.y * vv.y * sy * (1.f - sy),
                                d.z * vv.z * sz * (1.f - sz), d.w * vv.w * sw * (1.f - sw));
            if (TRAIN && valid) {
                *reinterpret_cast<float4*>(t_dvg + row * (2 * DFF) + 32 * hc + 8 * q + 4 * L.h) = dv[q];
                *reinterpret_cast<float4*>(t_dvg + row * (2 * DFF) + DFF + 32 * hc + 8 * q + 4 * L.h) = dg[q];
            }
        }
        gemm_t<4, 4>(winb, 2 * DFF / 8, 4 * hc, 0, dv, dn, L.lane);
        gemm_t<4, 4>(winb, 2 * DFF / 8, DFF / 8 + 4 * hc, 0, dg, dn, L.lane);
    }
    float4 w[16], x[16];
    acc_to_frag<4>(dn, w);
    load_rowfrag<16>(x, X1, row, D, L.h);
#pragma unroll
    for (int kg = 0; kg < 16; kg++) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + 8 * kg + 4 * L.h);
        w[kg].x *= g.x; w[kg].y *= g.y; w[kg].z *= g.z; w[kg].w *= g.w;
    }
    rmsnorm_bwd_frag<16>(w, x);
    if (valid) {
        load_rowfrag<16>(x, dY, row, D, L.h);  // residual: re-read dY (L2 hit) instead of keeping 64 registers live
#pragma unroll
        for (int kg = 0; kg < 16; kg++) {
            w[kg].x += x[kg].x; w[kg].y += x[kg].y; w[kg].z += x[kg].z; w[kg].w += x[kg].w;
        }
        store_rowfrag<16>(w, dX1, row, D, L.h);
    }
}

// ---------------------------------------------------------------------------------
// host launchers (declared in model.h)
// ---------------------------------------------------------------------------------
static inline int grid_rows(int64_t rows) { return cdiv(rows, WG_ROWS); }

void trr_qkv(const float* X, const float* gamma, const Lin& qkv, float* QKV, int64_t R, hipStream_t st) {
    k_qkv_t<<<grid_rows(R), 256, 0, st>>>(X, gamma, qkv.fwd, qkv.b, QKV, R);
}
void trr_qkv_bwd(const float* dQKV, const float* X, const float* gamma, const Lin& qkv, const float* dX1,
                 float* dXin, int64_t E, int64_t R, hipStream_t st) {
    k_qkv_bwd_t<<<grid_rows(R), 256, 0, st>>>(dQKV, X, gamma, qkv.bwd, dX1, dXin, E, R);
}
void trr_oproj(const float* AO, const float* X, const Lin& out, float* X1, float* OC, int64_t E, int64_t R,
               hipStream_t st) {
    k_oproj_t<<<grid_rows(R), 256, 0, st>>>(AO, X, out.fwd, out.b, X1, OC, E, R);
}
void trr_oproj_bwd(const float* dX1, const float* dOC, const Lin& out, float* dAO, int64_t E, int64_t R,
                   hipStream_t st) {
    k_oproj_bwd_t<<<grid_rows(R), 256, 0, st>>>(dX1, dOC, out.bwd, dAO, E, R);
}
// pet_config_set("bf16x6", 1): edge MLP GEMMs on the bf16 matrix cores (3-way split operands). Off by default:
// as accurate as the fp32 MFMA path (tools/ubench/bf16x3.hip: 3.7e-7 vs 4.5e-7 against fp64) and 1.85x faster as
// a plain GEMM, but in this fused stage the split fragments cost 96 VGPRs, occupancy drops to one wave per SIMD
// and the exposed weight-load latency at each small GEMM outweighs the shorter MFMA time (4.3 vs 3.2 ms / step).
static int g_bf16x6 = 0;
void set_bf16x6(int v) { g_bf16x6 = v ? 1 : 0; }

void trr_emlp(const float* X1, const float* gamma, const Lin& win, const Lin& wout, float* VG, float* X2,
              int64_t E, hipStream_t st) {
    if (g_bf16x6 && win.fwd3 && wout.fwd3)
        k_emlp_b<<<grid_rows(E), 256, 0, st>>>(X1, gamma, w3_fwd(win), win.b, w3_fwd(wout), wout.b, VG, X2, E);
    else
        k_emlp_t<<<grid_rows(E), 256, 0, st>>>(X1, gamma, win.fwd, win.b, wout.fwd, wout.b, VG, X2, E);
}
void trr_emlp_bwd(const float* dY, const float* X1, const float* VG, const float* gamma, const Lin& win,
                  const Lin& wout, float* dX1, int64_t E, hipStream_t st, float* t_dvg) {
    if (t_dvg) k_emlp_bwd_t<true><<<grid_rows(E), 256, 0, st>>>(dY, X1, VG, gamma, wout.bwd, win.bwd, dX1, E, t_dvg);
    else k_emlp_bwd_t<false><<<grid_rows(E), 256, 0, st>>>(dY, X1, VG, gamma, wout.bwd, win.bwd, dX1, E, nullptr);
}

}  // namespace pet
