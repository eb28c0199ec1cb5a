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
                                                     int64_t R, int accumulate) {
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
            if (accumulate) {
                float4 old[8];
                request_rows_addend<8>(old, Lp, [&](int r) { return Y + (r0 + r < R ? r0 + r : R - 1) * ldy + col; });
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

// false = not served: planes missing, shape not a multiple of 128 both ways, a small matrix of rows, or the row kernels'
// switch is off (pet_config_set("emlp_s", 0); "emlp_s" = 2 serves the tests' small graphs too)
bool rowgemm_s(hipStream_t st, const float* X, int K, const float* cs, const void* planes, const float* bias, float* Y, int n_out,
               int64_t R, bool acc) {
    if (!planes || K % 128 || n_out % 128 || K > 1024 || n_out > 1024 || !emlp_s_serves(R)) return false;
    const size_t n8 = (size_t)(n_out / 32) * (K / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(planes);
    W2 w; w.h = b; w.l = b + n8;
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    allow_big_lds(k_rowgemm_s, lds);
    k_rowgemm_s<<<(int)cdiv(R, HS_NW * WROWS), 256, lds, st>>>(X, K, K / 128, cs, w, bias, Y, n_out, n_out / 128, R, acc ? 1 : 0);
    return true;
}

}  // namespace pet
