// The generic row GEMM of the training passes, Y[R, 128 nn] (=|+=) (X[R, 128 nk] * cs) W^T + bias, in the form of k_emlp_s
// (rows_s.h): one-accumulator split-operand products, two desynchronised four-wave workgroups per CU, the four waves of a
// workgroup sharing ONE stream of weight fragments through a four-slot LDS ring requested three stages ahead. Round 6.
//
// k_rowgemm_k128 / k_rowgemm_n128 (so.hip) stream 64 KB of weight fragments per 128 x 128 product and WAVE from L2 -- 2 to 4 KB
// per row against the 1.5 to 2.5 KB per row that the row operands move through HBM -- and the wide shapes take one launch per
// 128 output columns, each re-reading X. Here a workgroup's 128 rows share each fragment (16 KB per product and wave), and one
// launch covers every shape of the second-order pass: product p = (column block nb, K slice ks), nb outermost; stage g of
// the stream = product g / 16, tile pair (g % 16) / 8, K block g % 8.
//
// The ring's stage waits count vmcnt (rows_s.h), which retires in order, so whatever else a wave sends to memory between a
// product's stages is waited for with the ring request behind it. Two things have to travel there:
//   * the next K slice of the rows (nk > 1): requested by LDS-DMA into the wave's row tile as soon as the current slice
//     has been turned into planes, 16 requests that may stay in flight over the product's first three stages (their waits
//     allow 2 + 16) and are drained at the fourth -- by the other workgroup's MFMAs, the kernel is bound by HBM;
//   * a finished column block (nn > 1): stored after its last stage through the row tile as whole lines; the 16 stores may
//     stay in flight over the next product's first three stages in the same way. When both happen (nk > 1 and nn > 1: the
//     256 x 256 matrices of the combination MLP) the row tile is busy staging the stores, so the first slice of the next
//     column block is requested behind them and waited for in full -- one exposed round trip per column block.
// Row scaling as in k_rowgemm_n128: one power of two per row and K slice, a running scale that only shrinks (exact).
#include "rows_s.h"
#include "tile.h"

namespace pet {

#define RS_STAGE_SYNC_LEAD()                                          \
    do {                                                              \
        asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");  \
        __syncthreads();                                              \
    } while (0)

// dma_tile128 (trr.h) with the addresses as one wave-uniform base (the slice's first row) + a 32-bit lane offset: the 64-bit
// row addresses of trr.h's form, three sets of sixteen, were kept across the product loop in spilled registers
__device__ __forceinline__ void rs_dma_tile(const float* __restrict__ Xs, int64_t r0, int64_t R, int ldx, unsigned lds_base,
                                            const RowLane& L) {
    const float* base = Xs + r0 * ldx;
    const int rmax = (int)(R - 1 - r0 < 31 ? R - 1 - r0 : 31);  // rows past the end: the last one again
    const int hi = L.lane >> 5, c = L.lane & 31;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int r = 2 * j + hi, rr = r < rmax ? r : rmax;
        const unsigned off = ((unsigned)rr * (unsigned)ldx + 4u * (unsigned)(c ^ (r & 15))) * 4u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_base + j * 1024) : "memory");
    }
}

// v[i] for the lanes of the low half, v[i + 4] for the others (the compiler folds the select into the address: one divergent
// vector load per value, consumed before the product's first stage; pinning the two values to scalar registers does not
// compile -- behind the ring's asm statements the loads are not provably unclobbered, so they are not scalar loads)
__device__ __forceinline__ float rs_pick(const float* __restrict__ v, int i, int h) { return h ? v[i + 4] : v[i]; }

// hs_gemm_r (rows_s.h) with `lead` (wave-uniform): 16 vector-memory operations of this wave that are younger than the ring
// requests of the first three stages may stay in flight over those stages. req(s, slot): request stage s of THIS product
// (s = 3 .. 18; 16 .. 18 are the first stages of the next one) into ring slot `slot`
template <class Req>
__device__ __forceinline__ void rs_gemm(f32x16 (&acc)[4], const f16x8 (&xh)[8], const f16x8 (&xl)[8], bool lead, Req req,
                                        const char* ring, unsigned lane16) {
#pragma unroll
    for (int r = 0; r < 16; r++) {  // (a product starts at a stage that is a multiple of 16: ring slot = r & 3)
        const int tp = r >> 3, kb = r & 7;
        if (r < 3 && lead) RS_STAGE_SYNC_LEAD();
        else HS_STAGE_SYNC();
        req(r + 3, (r + 3) & (HS_NSLOT - 1));
        const char* slot = ring + (r & (HS_NSLOT - 1)) * HS_SLOT + lane16;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * t) * 1024);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * t + 1) * 1024);
            AB_MFMA3(acc[2 * tp + t], wh, wl, xh[kb], xl[kb]);
        }
    }
}

// w: planes of 64 W (k_pack2h with both scales 64: Lin::fwd2s / bwd2s), fragments [(tile * (K / 16) + kb) * 64 + lane]
__global__ __launch_bounds__(256, 2) void k_rowgemm_s(const float* __restrict__ X, int ldx, int nk, const float* __restrict__ cs,
                                                     W2 w, const float* __restrict__ bias, float* __restrict__ Y, int ldy, int nn,
                                                     int64_t R, const float* __restrict__ A /* addend rows (ld = ldy), may be Y; or null */) {
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * HS_NW + wave) * WROWS;
    const bool live = row0 < R;
    if (!live) row0 = ((R - 1) / WROWS) * WROWS;  // run along on the last tile (same barriers), store nothing
    char* tile = rs_smem + wave * 16384;
    const char* ring = rs_smem + HS_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    const int kbt = 8 * nk, np = nk * nn;
    // stage s (0 .. 15) of product (nb, ks): tile pair s / 8, K block s % 8; wave w brings tile 2 tp + (w >> 1), plane w & 1
    const f16x8* plane = (wave & 1) ? w.l : w.h;
    auto piece = [=](int nb, int ks, int s, int slot) {
        const unsigned dst = ring_u + (unsigned)slot * HS_SLOT + wave * 1024;
        ab_dma_piece(plane, (4 * nb + 2 * (s >> 3) + (wave >> 1)) * kbt + 8 * ks + (s & 7), lane16, dst);
    };
    rs_dma_tile(X, row0, R, ldx, tile_u, L);
    piece(0, 0, 0, 0);
    piece(0, 0, 1, 1);
    piece(0, 0, 2, 2);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // the rows (the three ring requests may be in flight)
    f16x8 xh[8], xl[8];
    f32x16 acc[4];
    float scale = 0.f, inv = 0.f;  // the scale applied to what the accumulators hold, and its inverse
#pragma unroll 1
    for (int p = 0; p < np; p++) {
        const int nb = __builtin_amdgcn_readfirstlane(p / nk), ks = __builtin_amdgcn_readfirstlane(p - nb * nk);
        bool lead = nk == 1 && nb > 0;  // the previous column block's stores
        RowLane Lp = L;  // (the same for everything that is derived from the lane number: 16 swizzled LDS addresses, 16 DMA offsets ...)
        asm volatile("" : "+v"(Lp.lane), "+v"(Lp.r), "+v"(Lp.h));
        int64_t r0 = row0;  // opaque per product: the 16 + 16 + 16 row addresses of the requests below are formed where they are used
        asm volatile("" : "+s"(r0));  // (hoisted out of the loop they cost 150 registers of spills)
        if (ks == 0) {
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = ab_zero();
        }
        if (nk > 1 || p == 0) {  // this slice of the rows: tile -> fragments -> (column scales) -> power-of-two row scale -> planes
            float4 x[16];
            tile128_to_frag(x, tile, Lp);
            if (cs) {
                const float* v = cs + 128 * ks;
#pragma unroll
                for (int kg = 0; kg < 16; kg++) {
                    x[kg].x *= rs_pick(v, 8 * kg + 0, Lp.h);
                    x[kg].y *= rs_pick(v, 8 * kg + 1, Lp.h);
                    x[kg].z *= rs_pick(v, 8 * kg + 2, Lp.h);
                    x[kg].w *= rs_pick(v, 8 * kg + 3, Lp.h);
                }
            }
            float sc;
            const float iv = row_pow2<16>(x, sc);
            const bool shrink = ks == 0 || sc < scale;
            const float sc_eff = shrink ? sc : scale;
            if (ks > 0) {
                const float f = sc_eff * inv;  // 1 unless this slice is larger than everything before it
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[t][i] *= f;
            }
            scale = sc_eff;
            inv = shrink ? iv : inv;
#pragma unroll
            for (int kg = 0; kg < 16; kg++) { x[kg].x *= sc_eff; x[kg].y *= sc_eff; x[kg].z *= sc_eff; x[kg].w *= sc_eff; }
            hs_planes(x, xh, xl);
        }
        if (ks + 1 < nk) {  // the next slice into the tile while this product runs (the fragments above have been read)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            rs_dma_tile(X + 128 * (ks + 1), r0, R, ldx, tile_u, Lp);
            lead = true;
        }
        {   // the product after this one (past the end: this product's last stage again -- keeps vmcnt uniform)
            const bool last = p + 1 == np;
            const int ksn = ks + 1 < nk ? ks + 1 : 0, nbn = ks + 1 < nk ? nb : nb + 1;
            rs_gemm(acc, xh, xl, lead, [=](int s, int slot) {
                if (s < 16) piece(nb, ks, s, slot);
                else if (last) piece(nb, ks, 15, slot);
                else piece(nbn, ksn, s - 16, slot);
            }, ring, lane16);
        }
        if (ks + 1 < nk) continue;
        // ---- the column block is complete: scale back, bias, (addend), whole-line stores through the row tile
        const float f = inv * ABQ_INV;
        const float* bv = bias ? bias + 128 * nb : nullptr;
#pragma unroll
        for (int g = 0; g < 2; g++) {  // 64 columns at a time (registers: the planes stay live for the next column block)
            float4 y[8];
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const f32x16& a = acc[2 * g + t];
                    y[4 * t + j] = make_float4(a[4 * j] * f, a[4 * j + 1] * f, a[4 * j + 2] * f, a[4 * j + 3] * f);
                }
            if (bv) {
                const float* v = bv + 64 * g;
#pragma unroll
                for (int kg = 0; kg < 8; kg++) {
                    y[kg].x += rs_pick(v, 8 * kg + 0, Lp.h);
                    y[kg].y += rs_pick(v, 8 * kg + 1, Lp.h);
                    y[kg].z += rs_pick(v, 8 * kg + 2, Lp.h);
                    y[kg].w += rs_pick(v, 8 * kg + 3, Lp.h);
                }
            }
            const int col = 128 * nb + 64 * g;
            int64_t r0 = row0;
            asm volatile("" : "+s"(r0));
            auto out = [&](int r) { return live && r0 + r < R ? Y + (r0 + r) * ldy + col : nullptr; };
            __builtin_amdgcn_wave_barrier();
            if (A) {
                float4 old[8];
                request_rows_addend<8>(old, Lp, [&](int r) { return A + (r0 + r < R ? r0 + r : R - 1) * ldy + col; });
                store_rows_lines_add<8>(y, old, reinterpret_cast<float*>(tile), Lp, out);
            } else
                store_rows_lines<8>(y, reinterpret_cast<float*>(tile), Lp, out);
        }
        if (nk > 1 && nb + 1 < nn) {  // the first slice again for the next column block, behind the stores: waited for in full
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            int64_t r1 = row0;
            asm volatile("" : "+s"(r1));
            rs_dma_tile(X, r1, R, ldx, tile_u, Lp);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Two shapes get kernels of their own, because the generic one re-reads the rows of the input once per column block when
// BOTH K and the output are wider than 128 (the slice it needs next lands with its full latency exposed, once per block):
//   k_rowgemm_s_k2   K = 256, any width of output: the planes of both slices stay in registers across the column blocks (128
//                    registers), the column blocks leave 32 columns at a time; optionally the rows are normalised first
//                    (NORM 1: RMSNorm, 2: LayerNorm, with weight cs and bias cb -- the node update's norm_center_features)
//   k_rowgemm_s_n2   256 output columns, any K: the K slices are the OUTER loop and both column blocks accumulate at once
//                    (128 accumulator registers), every slice is read once
// Both take an addend (Y = A + ...; A == Y accumulates in place). Same stream order convention as above: product p of the
// stream = (column block, K slice) in the order the kernel consumes them.
// ---------------------------------------------------------------------------------------------------------------------------
#define RS_SETUP()                                                                                       \
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];                                       \
    const RowLane L;                                                                                     \
    const unsigned lane16 = (unsigned)L.lane * 16u;                                                      \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                   \
    int64_t row0 = ((int64_t)blockIdx.x * HS_NW + wave) * WROWS;                                         \
    const bool live = row0 < R;                                                                          \
    if (!live) row0 = ((R - 1) / WROWS) * WROWS;                                                         \
    const bool full = live && row0 + WROWS <= R; /* every store instruction of the tile is issued */     \
    char* tile = rs_smem + wave * 16384;                                                                 \
    const char* ring = rs_smem + HS_NW * 16384;                                                          \
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);                   \
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);                   \
    const f16x8* plane = (wave & 1) ? w.l : w.h
#define RS_LANE(Lx)                            \
    RowLane Lx = L;                            \
    asm volatile("" : "+v"(Lx.lane));          \
    Lx.r = Lx.lane & 31;                       \
    Lx.h = Lx.lane >> 5
#define RS_ROW0(rx) int64_t rx = row0; asm volatile("" : "+s"(rx))

// one 128-column block of results: scale back, bias, addend, 32 columns at a time as whole 128-B lines through the tile
// (4 x 4 store instructions; the addend's row-fragment loads are consumed before the stores are issued)
__device__ __forceinline__ void rs_store_block(const f32x16 (&acc)[4], float f, const float* __restrict__ bv, const float* __restrict__ A,
                                               float* __restrict__ Y, int ldy, int col, int64_t r0, int64_t R, bool live, char* tile,
                                               const RowLane& Lq) {
    const int64_t row = r0 + Lq.r < R ? r0 + Lq.r : R - 1;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        float4 y[4];
#pragma unroll
        for (int j = 0; j < 4; j++) y[j] = make_float4(acc[t][4 * j] * f, acc[t][4 * j + 1] * f, acc[t][4 * j + 2] * f, acc[t][4 * j + 3] * f);
        if (bv) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 b4 = *reinterpret_cast<const float4*>(bv + 32 * t + 8 * j + 4 * Lq.h);
                y[j].x += b4.x; y[j].y += b4.y; y[j].z += b4.z; y[j].w += b4.w;
            }
        }
        if (A) {
            float4 a4[4];
            load_rowfrag<4>(a4, A + col + 32 * t, row, ldy, Lq.h);
#pragma unroll
            for (int j = 0; j < 4; j++) { y[j].x += a4[j].x; y[j].y += a4[j].y; y[j].z += a4[j].z; y[j].w += a4[j].w; }
        }
        __builtin_amdgcn_wave_barrier();
        store_tile32_lines(y, reinterpret_cast<float*>(tile), Y + col + 32 * t, r0, live ? R : 0, ldy, Lq);
    }
}

// The node update's adjoint, two epilogues (pet_node_s.hip): each removes a row-wise kernel -- and a link of the dependent chain that
// runs beside the edge kernels -- and the round trip of its input through HBM.
// (a) du (this 128-column block of the 512) -> (dv, dg) = (du sigmoid(g), du v sigmoid'(g)) with the saved [value | gate] rows:
//     two 32-column tiles leave per accumulator tile (8 x 4 store instructions per block)
__device__ __forceinline__ void rs_store_block_swiglu_bwd(const f32x16 (&acc)[4], float f, const float* __restrict__ VG,
                                                          float* __restrict__ dVG, int hid, int col, int64_t r0, int64_t R, bool live,
                                                          char* tile, const RowLane& Lq) {
    const int64_t row = r0 + Lq.r < R ? r0 + Lq.r : R - 1;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        float4 v4[4], g4[4], dv[4], dg[4];
        load_rowfrag<4>(v4, VG + col + 32 * t, row, 2 * hid, Lq.h);
        load_rowfrag<4>(g4, VG + hid + col + 32 * t, row, 2 * hid, Lq.h);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float du[4] = {acc[t][4 * j] * f, acc[t][4 * j + 1] * f, acc[t][4 * j + 2] * f, acc[t][4 * j + 3] * f};
            const float sx = sigmoidf_(g4[j].x), sy = sigmoidf_(g4[j].y), sz = sigmoidf_(g4[j].z), sw = sigmoidf_(g4[j].w);
            dv[j] = make_float4(du[0] * sx, du[1] * sy, du[2] * sz, du[3] * sw);
            dg[j] = make_float4(du[0] * v4[j].x * sx * (1.f - sx), du[1] * v4[j].y * sy * (1.f - sy), du[2] * v4[j].z * sz * (1.f - sz),
                                du[3] * v4[j].w * sw * (1.f - sw));
        }
        __builtin_amdgcn_wave_barrier();
        store_tile32_lines(dv, reinterpret_cast<float*>(tile), dVG + col + 32 * t, r0, live ? R : 0, 2 * hid, Lq);
        store_tile32_lines(dg, reinterpret_cast<float*>(tile), dVG + hid + col + 32 * t, r0, live ? R : 0, 2 * hid, Lq);
    }
}
// (b) dy (both 128-column blocks of the 256-wide row, in the accumulators) -> out = dres + norm^T(dy; x): RMSNorm (ln == 0; eps
//     2^-23) or LayerNorm (eps 1e-5) with weight gamma: dx = rstd (dyh - [mean(dyh)] - xh mean(dyh xh)), dyh = dy gamma. The row x
//     is read twice (statistics, then the result), 32 columns at a time: the second read comes from L2.
__device__ __forceinline__ void rs_store_norm_bwd(f32x16 (&a0)[4], f32x16 (&a1)[4], float f, const float* __restrict__ x,
                                                  const float* __restrict__ gamma, int ln, const float* __restrict__ dres,
                                                  float* __restrict__ out, int64_t r0, int64_t R, bool live, char* tile,
                                                  const RowLane& Lq) {
    const int64_t row = r0 + Lq.r < R ? r0 + Lq.r : R - 1;
    float s1 = 0.f, s2 = 0.f, sx = 0.f, sxx = 0.f;  // sum dyh, sum dyh x, sum x, sum x^2 (LayerNorm: centred below)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            f32x16& a = b ? a1[t] : a0[t];
            float4 x4[4];
            load_rowfrag<4>(x4, x + 128 * b + 32 * t, row, 256, Lq.h);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + 128 * b + 32 * t + 8 * j + 4 * Lq.h);
                const float xv[4] = {x4[j].x, x4[j].y, x4[j].z, x4[j].w}, gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float d = a[4 * j + i] * f * gv[i];
                    a[4 * j + i] = d;  // dyh from here on
                    s1 += d; s2 = fmaf(d, xv[i], s2); sx += xv[i]; sxx = fmaf(xv[i], xv[i], sxx);
                }
            }
        }
    s1 = row_sum(s1); s2 = row_sum(s2); sx = row_sum(sx); sxx = row_sum(sxx);
    const float mean = ln ? sx * (1.0f / 256.0f) : 0.f;
    const float var = sxx * (1.0f / 256.0f) - mean * mean;
    const float rstd = rsqrtf(var + (ln ? 1e-5f : 1.1920928955078125e-07f));
    const float m1 = ln ? s1 * (1.0f / 256.0f) : 0.f;
    const float m2 = (s2 - mean * s1) * rstd * (1.0f / 256.0f);  // mean(dyh xh), xh = (x - mean) rstd
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const f32x16& a = b ? a1[t] : a0[t];
            float4 x4[4], r4[4], y[4];
            load_rowfrag<4>(x4, x + 128 * b + 32 * t, row, 256, Lq.h);
            load_rowfrag<4>(r4, dres + 128 * b + 32 * t, row, 256, Lq.h);
#pragma unroll
            for (int j = 0; j < 4; j++)
                y[j] = make_float4(r4[j].x + rstd * (a[4 * j] - m1 - (x4[j].x - mean) * rstd * m2),
                                   r4[j].y + rstd * (a[4 * j + 1] - m1 - (x4[j].y - mean) * rstd * m2),
                                   r4[j].z + rstd * (a[4 * j + 2] - m1 - (x4[j].z - mean) * rstd * m2),
                                   r4[j].w + rstd * (a[4 * j + 3] - m1 - (x4[j].w - mean) * rstd * m2));
            __builtin_amdgcn_wave_barrier();
            store_tile32_lines(y, reinterpret_cast<float*>(tile), out + 128 * b + 32 * t, r0, live ? R : 0, 256, Lq);
        }
}

template <int NORM, int EPI = 0>  // EPI 1: the SwiGLU adjoint as the epilogue (A = the saved [value | gate] rows, Y = [N, 2 x 128 nn])
__global__ __launch_bounds__(256, 2) void k_rowgemm_s_k2(const float* __restrict__ X, int ldx, const float* __restrict__ cs,
                                                        const float* __restrict__ cb, W2 w, const float* __restrict__ bias,
                                                        const float* __restrict__ A, float* __restrict__ Y, int ldy, int nn, int64_t R) {
    RS_SETUP();
    constexpr int nk = 2, kbt = 16;
    auto piece = [=](int nb, int ks, int s, int slot) {
        const unsigned dst = ring_u + (unsigned)slot * HS_SLOT + wave * 1024;
        ab_dma_piece(plane, (4 * nb + 2 * (s >> 3) + (wave >> 1)) * kbt + 8 * ks + (s & 7), lane16, dst);
    };
    rs_dma_tile(X, row0, R, ldx, tile_u, L);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f16x8 xh[2][8], xl[2][8];
    float inv;
    {
        float4 x0[16], x1[16];
        tile128_to_frag(x0, tile, L);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads have returned before the tile is requested again
        __builtin_amdgcn_wave_barrier();
        rs_dma_tile(X + 128, row0, R, ldx, tile_u, L);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile128_to_frag(x1, tile, L);
        if (NORM) {  // RMSNorm (eps 2^-23) or torch.nn.LayerNorm (eps 1e-5) over the 256 columns, then weight (and bias)
            float mean = 0.f;
            if (NORM == 2) {
                float s1 = 0.f;
#pragma unroll
                for (int k = 0; k < 16; k++) s1 += (x0[k].x + x0[k].y) + (x0[k].z + x0[k].w) + (x1[k].x + x1[k].y) + (x1[k].z + x1[k].w);
                mean = row_sum(s1) * (1.0f / 256.0f);
            }
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (NORM == 2) {
                    x0[k].x -= mean; x0[k].y -= mean; x0[k].z -= mean; x0[k].w -= mean;
                    x1[k].x -= mean; x1[k].y -= mean; x1[k].z -= mean; x1[k].w -= mean;
                }
                s2 += x0[k].x * x0[k].x + x0[k].y * x0[k].y + x0[k].z * x0[k].z + x0[k].w * x0[k].w;
                s2 += x1[k].x * x1[k].x + x1[k].y * x1[k].y + x1[k].z * x1[k].z + x1[k].w * x1[k].w;
            }
            const float rstd = rsqrtf(row_sum(s2) * (1.0f / 256.0f) + (NORM == 2 ? 1e-5f : 1.1920928955078125e-07f));
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float4 g0 = *reinterpret_cast<const float4*>(cs + 8 * k + 4 * L.h);
                const float4 g1 = *reinterpret_cast<const float4*>(cs + 128 + 8 * k + 4 * L.h);
                x0[k] = make_float4(x0[k].x * rstd * g0.x, x0[k].y * rstd * g0.y, x0[k].z * rstd * g0.z, x0[k].w * rstd * g0.w);
                x1[k] = make_float4(x1[k].x * rstd * g1.x, x1[k].y * rstd * g1.y, x1[k].z * rstd * g1.z, x1[k].w * rstd * g1.w);
                if (NORM == 2) {
                    const float4 b0 = *reinterpret_cast<const float4*>(cb + 8 * k + 4 * L.h);
                    const float4 b1 = *reinterpret_cast<const float4*>(cb + 128 + 8 * k + 4 * L.h);
                    x0[k].x += b0.x; x0[k].y += b0.y; x0[k].z += b0.z; x0[k].w += b0.w;
                    x1[k].x += b1.x; x1[k].y += b1.y; x1[k].z += b1.z; x1[k].w += b1.w;
                }
            }
        } else if (cs) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float4 g0 = *reinterpret_cast<const float4*>(cs + 8 * k + 4 * L.h);
                const float4 g1 = *reinterpret_cast<const float4*>(cs + 128 + 8 * k + 4 * L.h);
                x0[k].x *= g0.x; x0[k].y *= g0.y; x0[k].z *= g0.z; x0[k].w *= g0.w;
                x1[k].x *= g1.x; x1[k].y *= g1.y; x1[k].z *= g1.z; x1[k].w *= g1.w;
            }
        }
        float sc0, sc1;
        const float iv0 = row_pow2<16>(x0, sc0), iv1 = row_pow2<16>(x1, sc1);
        const float sc = sc0 < sc1 ? sc0 : sc1;  // one power of two for the row: the larger slice decides
        inv = sc0 < sc1 ? iv0 : iv1;
#pragma unroll
        for (int k = 0; k < 16; k++) { x0[k].x *= sc; x0[k].y *= sc; x0[k].z *= sc; x0[k].w *= sc; }
        hs_planes(x0, xh[0], xl[0]);
#pragma unroll
        for (int k = 0; k < 16; k++) { x1[k].x *= sc; x1[k].y *= sc; x1[k].z *= sc; x1[k].w *= sc; }
        hs_planes(x1, xh[1], xl[1]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (the norm's weight loads: nothing but fragments in the queue from here on)
    piece(0, 0, 0, 0);
    piece(0, 0, 1, 1);
    piece(0, 0, 2, 2);
    const float f = inv * ABQ_INV;
#pragma unroll 1
    for (int nb = 0; nb < nn; nb++) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = ab_zero();
        const bool last = nb + 1 == nn;
        // (16 store instructions of the previous block may be in flight over the first three stages -- if all of them were issued)
        if (nb > 0 && !full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a partial tile skips store instructions: drained instead of counted)
        if (EPI == 1 && nb > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (32 stores per block there: drained, not budgeted)
        rs_gemm(acc, xh[0], xl[0], EPI == 0 && nb > 0 && full, [=](int s, int slot) {
            if (s < 16) piece(nb, 0, s, slot); else piece(nb, 1, s - 16, slot);
        }, ring, lane16);
        rs_gemm(acc, xh[1], xl[1], false, [=](int s, int slot) {
            if (s < 16) piece(nb, 1, s, slot);
            else if (last) piece(nb, 1, 15, slot);
            else piece(nb + 1, 0, s - 16, slot);
        }, ring, lane16);
        RS_LANE(Lq);
        RS_ROW0(r0);
        if (EPI == 1) rs_store_block_swiglu_bwd(acc, f, A, Y, 128 * nn, 128 * nb, r0, R, live, tile, Lq);
        else rs_store_block(acc, f, bias ? bias + 128 * nb : nullptr, A, Y, ldy, 128 * nb, r0, R, live, tile, Lq);
    }
}

template <int EPI = 0>  // EPI 1: the norm adjoint as the epilogue (bias = the norm's weight, A = the residual's adjoint, xn = the normalised rows' source)
__global__ __launch_bounds__(256, 2) void k_rowgemm_s_n2(const float* __restrict__ X, int ldx, int nk, W2 w,
                                                        const float* __restrict__ bias, const float* __restrict__ A,
                                                        float* __restrict__ Y, int ldy, int64_t R,
                                                        const float* __restrict__ xn = nullptr, int ln = 0) {
    RS_SETUP();
    (void)full;
    const int kbt = 8 * nk;
    auto piece = [=](int nb, int ks, int s, int slot) {
        const unsigned dst = ring_u + (unsigned)slot * HS_SLOT + wave * 1024;
        ab_dma_piece(plane, (4 * nb + 2 * (s >> 3) + (wave >> 1)) * kbt + 8 * ks + (s & 7), lane16, dst);
    };
    rs_dma_tile(X, row0, R, ldx, tile_u, L);
    piece(0, 0, 0, 0);
    piece(0, 0, 1, 1);
    piece(0, 0, 2, 2);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    f32x16 acc0[4], acc1[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { acc0[t] = ab_zero(); acc1[t] = ab_zero(); }
    f16x8 xh[8], xl[8];
    float scale = 0.f, inv = 0.f;
#pragma unroll 1
    for (int ks = 0; ks < nk; ks++) {
        RS_LANE(Lp);
        RS_ROW0(r0);
        {   // the slice: its largest entry first (one pass over the tile), then planes eight fragments at a time (registers)
            const char* rowp = tile + 512 * Lp.r;
            const int sw = Lp.r & 15;
            float m = 0.f;
#pragma unroll
            for (int kg = 0; kg < 16; kg++) {
                const float4 v = *reinterpret_cast<const float4*>(rowp + 16 * ((2 * kg + Lp.h) ^ sw));
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
            m = fmaxf(m, __shfl_xor(m, 32));
            int e = (__float_as_int(m) >> 23) & 0xff;
            e = e > 253 ? 253 : e;
            const float sc = __int_as_float((254 - e) << 23), iv = __int_as_float(e << 23);  // trr.h row_pow2
            const bool shrink = ks == 0 || sc < scale;
            const float sc_eff = shrink ? sc : scale;
            if (ks > 0) {
                const float g = sc_eff * inv;  // 1 unless this slice is larger than everything before it
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int i = 0; i < 16; i++) { acc0[t][i] *= g; acc1[t][i] *= g; }
            }
            scale = sc_eff;
            inv = shrink ? iv : inv;
            const float fs = sc_eff;  // (times ABS in a second step: an all-zero row's scale is 2^127, and 0 x (2^127 x 64) is not 0)
#pragma unroll
            for (int kb = 0; kb < 8; kb++) {
                const float4 a = *reinterpret_cast<const float4*>(rowp + 16 * ((4 * kb + Lp.h) ^ sw));
                const float4 b = *reinterpret_cast<const float4*>(rowp + 16 * ((4 * kb + 2 + Lp.h) ^ sw));
                const float v8[8] = {a.x * fs * ABS, a.y * fs * ABS, a.z * fs * ABS, a.w * fs * ABS,
                                     b.x * fs * ABS, b.y * fs * ABS, b.z * fs * ABS, b.w * fs * ABS};
                ab_split8(v8, xh[kb], xl[kb]);
            }
        }
        const bool more = ks + 1 < nk;
        if (more) {  // the next slice into the tile while this slice's two products run
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            rs_dma_tile(X + 128 * (ks + 1), r0, R, ldx, tile_u, Lp);
        }
        rs_gemm(acc0, xh, xl, more, [=](int s, int slot) {
            if (s < 16) piece(0, ks, s, slot); else piece(1, ks, s - 16, slot);
        }, ring, lane16);
        rs_gemm(acc1, xh, xl, false, [=](int s, int slot) {
            if (s < 16) piece(1, ks, s, slot);
            else if (!more) piece(1, ks, 15, slot);
            else piece(0, ks + 1, s - 16, slot);
        }, ring, lane16);
    }
    const float f = inv * ABQ_INV;
    RS_LANE(Lq);
    RS_ROW0(r0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (EPI == 1) {
        rs_store_norm_bwd(acc0, acc1, f, xn, bias, ln, A, Y, r0, R, live, tile, Lq);
        return;
    }
    rs_store_block(acc0, f, bias, A, Y, ldy, 0, r0, R, live, tile, Lq);
    rs_store_block(acc1, f, bias ? bias + 128 : nullptr, A, Y, ldy, 128, r0, R, live, tile, Lq);
}

// false = not served: planes missing, shape not a multiple of 128 both ways, a small matrix of rows, or the row kernels'
// switch is off (pet_config_set("emlp_s", 0); "emlp_s" = 2 serves the tests' small graphs too).
// Y = [A +] (norm(X) | X * cs) W^T [+ bias]; A may be Y (accumulate in place); norm 1 / 2: RMSNorm / LayerNorm of the 256-wide rows
// with weight cs (and bias cb) before the product (K == 256 only)
bool rowgemm_s_ex(hipStream_t st, const float* X, int K, const float* cs, const void* planes, const float* bias, const float* A,
                  float* Y, int n_out, int64_t R, int norm, const float* cb) {
    if (!planes || K % 128 || n_out % 128 || K > 1024 || n_out > 1024 || !emlp_s_serves(R)) return false;
    if (norm && (K != 256 || !cs || (norm == 2 && !cb))) return false;
    const size_t n8 = (size_t)(n_out / 32) * (K / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(planes);
    W2 w; w.h = b; w.l = b + n8;
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    const int grid = (int)cdiv(R, HS_NW * WROWS);
    if (norm == 1) {
        allow_big_lds(k_rowgemm_s_k2<1>, lds);
        k_rowgemm_s_k2<1><<<grid, 256, lds, st>>>(X, K, cs, cb, w, bias, A, Y, n_out, n_out / 128, R);
    } else if (norm == 2) {
        allow_big_lds(k_rowgemm_s_k2<2>, lds);
        k_rowgemm_s_k2<2><<<grid, 256, lds, st>>>(X, K, cs, cb, w, bias, A, Y, n_out, n_out / 128, R);
    } else if (K == 256 && n_out >= 256) {
        allow_big_lds(k_rowgemm_s_k2<0>, lds);
        k_rowgemm_s_k2<0><<<grid, 256, lds, st>>>(X, K, cs, nullptr, w, bias, A, Y, n_out, n_out / 128, R);
    } else if (n_out == 256 && K >= 256 && !cs) {
        allow_big_lds(k_rowgemm_s_n2<0>, lds);
        k_rowgemm_s_n2<0><<<grid, 256, lds, st>>>(X, K, K / 128, w, bias, A, Y, n_out, R);
    } else {
        allow_big_lds(k_rowgemm_s, lds);
        k_rowgemm_s<<<grid, 256, lds, st>>>(X, K, K / 128, cs, w, bias, Y, n_out, n_out / 128, R, A);
    }
    return true;
}
// dVG[R, 2 hid] = swiglu'(VG) ((X[R, 256]) W^T): the node update's adjoint, first half (hid a multiple of 128)
bool rowgemm_s_swiglu_bwd(hipStream_t st, const float* X, const void* planes, const float* VG, float* dVG, int hid, int64_t R) {
    if (!planes || hid % 128 || hid > 1024 || !emlp_s_serves(R)) return false;
    const f16x8* b = reinterpret_cast<const f16x8*>(planes);
    W2 w; w.h = b; w.l = b + (size_t)(hid / 32) * (256 / 16) * 64;
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    allow_big_lds(k_rowgemm_s_k2<0, 1>, lds);
    k_rowgemm_s_k2<0, 1><<<(int)cdiv(R, HS_NW * WROWS), 256, lds, st>>>(X, 256, nullptr, nullptr, w, nullptr, VG, dVG, 2 * hid, hid / 128, R);
    return true;
}
// out[R, 256] = dres + norm^T((X[R, K]) W^T; xn): the node update's adjoint, second half (gamma: the norm's weight; ln: LayerNorm)
bool rowgemm_s_norm_bwd(hipStream_t st, const float* X, int K, const void* planes, const float* xn, const float* gamma, int ln,
                        const float* dres, float* out, int64_t R) {
    if (!planes || K % 128 || K < 256 || K > 1024 || !emlp_s_serves(R)) return false;
    const f16x8* b = reinterpret_cast<const f16x8*>(planes);
    W2 w; w.h = b; w.l = b + (size_t)(256 / 32) * (K / 16) * 64;
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    allow_big_lds(k_rowgemm_s_n2<1>, lds);
    k_rowgemm_s_n2<1><<<(int)cdiv(R, HS_NW * WROWS), 256, lds, st>>>(X, K, K / 128, w, gamma, dres, out, 256, R, xn, ln);
    return true;
}
bool rowgemm_s(hipStream_t st, const float* X, int K, const float* cs, const void* planes, const float* bias, float* Y, int n_out,
               int64_t R, bool acc) {
    return rowgemm_s_ex(st, X, K, cs, planes, bias, acc ? Y : nullptr, Y, n_out, R, 0, nullptr);
}

}  // namespace pet
