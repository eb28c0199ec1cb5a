// The three node-row Linear layers around the attention block of large graphs in the form of k_emlp_s / k_head_s (rows_s.h):
//   k_center       X[E + i] = H[i] Wcc^T + b            (DN -> D;  transformer.py:211-214, centre contraction)
//   k_expand_bwd   dOC = dH1 Wce                        (DN -> D;  adjoint of the centre expansion, transformer.py:222)
//   k_center_bwd   dHin = dH1 + dC Wcc                  (D -> DN;  adjoint of the contraction)
// The LDS-tile kernels they replace (pet_fwd.hip / pet_bwd.hip; they keep serving small graphs and the training passes) run 64-row
// workgroups whose waves each stream their own weight blocks from L2, two in flight: 63 / 76 / 83 us per launch at 80 000 atoms for
// 25 us worth of HBM traffic, four launches of each per step on a node chain that the edge kernels do not overlap.
// Here: one-accumulator split-operand products, two desynchronised four-wave workgroups per CU, a four-slot weight ring shared
// by the workgroup; a 256-wide row arrives as two 128-column halves through the wave's 16-KB tile; every row is scaled by its
// power of two first (node features and adjoints are un-normalised). Round 5.
#include "rows_s.h"

namespace pet {

// Y[N, NOUT] = X[N, KIN] W^T (+ bias) (+ addend rows), W as planes of 64 w in fragment order [tile][K block]
template <int KIN, int NOUT, bool ADD>
__global__ __launch_bounds__(256, 2) void k_rowlin_s(const float* __restrict__ X, W2 w, const float* __restrict__ bias,
                                                    const float* __restrict__ addend, float* __restrict__ Y, int64_t N) {
    constexpr int KB = KIN / 16, NT = NOUT / 32, NST = (NT / 2) * KB;
    extern __shared__ __attribute__((aligned(16))) char cs_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * HS_NW + wave) * WROWS;
    const bool live = row0 < N;
    if (!live) row0 = ((N - 1) / WROWS) * WROWS;  // run along on the last tile (same barriers), store nothing
    char* tile = cs_smem + wave * 16384;
    const char* ring = cs_smem + HS_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    auto req = [&](int g) {  // stage g: tile pair g / KB, K block g % KB; wave w brings tile 2 tp + (w >> 1), plane w & 1
        g = g < NST ? g : NST - 1;
        const unsigned dst = ring_u + (unsigned)(g & (HS_NSLOT - 1)) * HS_SLOT + wave * 1024;
        const int tp = g / KB, kb = g % KB;
        ab_dma_piece((wave & 1) ? w.l : w.h, (2 * tp + (wave >> 1)) * KB + kb, lane16, dst);
    };
    // the rows: 128 columns at a time through the tile; the power-of-two scale is the whole row's
    float4 x[KIN / 8];
    {
        float4 xa[16];
        dma_tile128(X, row0, N, tile_u, L, KIN);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile128_to_frag(xa, tile, L);
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = xa[k];
        if (KIN == 256) {
            // the reads must have RETURNED before the tile is requested again (an L2-warm LDS-DMA lands after 250-400 cycles, sooner
            // than sixteen queued ds_read_b128 of a busy CU are served; the copies above are no instructions, so nothing else waits)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            dma_tile128(X + 128, row0, N, tile_u, L, KIN);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tile128_to_frag(xa, tile, L);
#pragma unroll
            for (int k = 0; k < 16; k++) x[(KIN == 256 ? 16 : 0) + k] = xa[k];
        }
    }
    req(0);
    req(1);
    req(2);
    float inv;
    f16x8 xh[KB], xl[KB];
    {
        float sc;
        inv = row_scale_pow2<KIN / 8>(x, sc);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            const float v8[8] = {x[2 * kb].x * ABS, x[2 * kb].y * ABS, x[2 * kb].z * ABS, x[2 * kb].w * ABS,
                                 x[2 * kb + 1].x * ABS, x[2 * kb + 1].y * ABS, x[2 * kb + 1].z * ABS, x[2 * kb + 1].w * ABS};
            ab_split8(v8, xh[kb], xl[kb]);
        }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = ab_zero();
#pragma unroll
    for (int g = 0; g < NST; g++) {
        const int tp = g / KB, kb = g % KB;
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
    // ---- after the last stage: whole lines through the wave's tile, 128 columns at a time
    const float f = inv * ABQ_INV;
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int hf = 0; hf < NOUT / 128; hf++) {
        float4 y[16];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f32x16& a = acc[4 * hf + t];
                y[4 * t + j] = make_float4(a[4 * j] * f, a[4 * j + 1] * f, a[4 * j + 2] * f, a[4 * j + 3] * f);
                if (bias) {
                    y[4 * t + j].x += hs_vec(bias + 128 * hf, t, j, 0, L.h); y[4 * t + j].y += hs_vec(bias + 128 * hf, t, j, 1, L.h);
                    y[4 * t + j].z += hs_vec(bias + 128 * hf, t, j, 2, L.h); y[4 * t + j].w += hs_vec(bias + 128 * hf, t, j, 3, L.h);
                }
            }
        auto out = [&](int r) { return live && row0 + r < N ? Y + (row0 + r) * NOUT + 128 * hf : nullptr; };
        if (ADD) {
            float4 old[16];
            request_rows_addend<16>(old, L, [&](int r) { return addend + (row0 + r < N ? row0 + r : N - 1) * NOUT + 128 * hf; });
            store_rows_lines_add<16>(y, old, reinterpret_cast<float*>(tile), L, out);
        } else {
            store_rows_lines<16>(y, reinterpret_cast<float*>(tile), L, out);
        }
    }
}

static inline W2 ns_w2(const void* base, int tiles, int kbs) {
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + (size_t)tiles * kbs * 64;
    return w;
}
// the node chain takes this form from 16 384 atoms on (below: the 32-row node kernels fuse these layers, pet_fwd.hip node_rows)
static bool center_s_serves(int64_t N) { return emlp_s_serves((int64_t)1 << 40) && (N >= 16384 || emlp_s_forced()); }

template <int KIN, int NOUT, bool ADD>
static void rowlin_s_launch(const float* X, W2 w, const float* bias, const float* addend, float* Y, int64_t N, hipStream_t st) {
    const size_t lds = HS_NW * 16384 + HS_NSLOT * HS_SLOT;
    allow_big_lds(k_rowlin_s<KIN, NOUT, ADD>, lds);
    k_rowlin_s<KIN, NOUT, ADD><<<(int)cdiv(N, HS_NW * WROWS), 256, lds, st>>>(X, w, bias, addend, Y, N);
}
// false = not served (weights not packed for it, small graph, or pet_config_set("emlp_s", 0))
bool center_s(const Lin& cc, const float* H, float* Xc, int64_t N, hipStream_t st) {  // X[E + i] = H[i] Wcc^T + b
    if (!center_s_serves(N) || !cc.fwd2s) return false;
    rowlin_s_launch<DN, D, false>(H, ns_w2(cc.fwd2s, D / 32, DN / 16), cc.b, nullptr, Xc, N, st);
    return true;
}
bool expand_bwd_s(const Lin& ce, const float* dH1, float* dOC, int64_t N, hipStream_t st) {  // dOC = dH1 Wce
    if (!center_s_serves(N) || !ce.bwd2s) return false;
    rowlin_s_launch<DN, D, false>(dH1, ns_w2(ce.bwd2s, D / 32, DN / 16), nullptr, nullptr, dOC, N, st);
    return true;
}
bool center_bwd_s(const Lin& cc, const float* dC, const float* dH1, float* dHin, int64_t N, hipStream_t st) {  // dHin = dH1 + dC Wcc
    if (!center_s_serves(N) || !cc.bwd2s) return false;
    rowlin_s_launch<D, DN, true>(dC, ns_w2(cc.bwd2s, DN / 32, D / 16), nullptr, dH1, dHin, N, st);
    return true;
}

}  // namespace pet
