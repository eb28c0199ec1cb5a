// The edge MLP (transformer.py:39-50, 230-232: X2 = X1 + W_out swiglu(W_in RMSNorm(X1))) built like the fused attention
// forward (pet_ablk.hip) instead of like the one-wave-per-SIMD pipelined row kernel k_emlp_p2 (pet_trr.hip): round 5.
//
// k_emlp_p2 keeps two accumulators per product (acc + 2^-11 acl), which costs it the whole register file: one wave per
// SIMD, and measured (tools/experiments/README.md) that wave spends 20 % of its life waiting for its own 16 KB of rows with
// nothing else resident, and its matrix and vector work add up instead of overlapping. Here every product is the
// ONE-accumulator split-operand product of ablk.h (planes H = fp16(s x), L = fp16(s x - H); three MFMAs on one accumulator),
// so a wave needs ~200 registers and 16 KB of LDS: two workgroups of four waves per CU = two waves per SIMD that are NOT in
// step with each other (an eight-wave workgroup was tried first: its waves meet at every stage barrier, so the two waves of a
// SIMD were in their matrix phase or in their vector phase at the same time -- SQ_VALU_MFMA_COEXEC_CYCLES 875 per wave against
// 5 600 in k_emlp_p2 -- and it ran no faster than k_emlp_p2). The four waves of a workgroup share ONE stream of weight fragments
// through a four-slot LDS ring (each weight byte is fetched once per four row tiles instead of once per tile), requested
// three stages ahead (an LDS-DMA request takes ~1 us to land; a stage is 6 MFMAs per wave).
//
//   rows      32 per wave, LDS-DMA; residual + output bias = initial value of the out accumulators; RMSNorm / LayerNorm;
//             planes of 64 xn parked over the rows
//   chunk hc  (32 hidden units, 8 of them): [v; g] = W_in xn + b (4 stages of 2 K blocks: 48 MFMAs), saved if asked for;
//             u = v sigmoid(g), planes of u at scale 1 (fp16 range 65504: u is not bounded like a normalised row);
//             out += W_out[:, chunk] u (2 stages of one K block x 4 tiles: 24 MFMAs)
//   stores    whole lines through the wave's (dead) plane tile, as in k_ablk_fwd
// Scales: W planes 64 x, xn planes 64 x -> [v; g] accumulators hold 4096 x; u planes 1 x -> out accumulators hold 64 x.
#include "ablk.h"

namespace pet {

constexpr int ES_NW = 4;       // waves per workgroup
constexpr int ES_SLOT = 4096;  // ring slot: 4 fragments, one per wave
constexpr int ES_NSLOT = 4;
constexpr int ES_SPC = 12;     // stages per hidden chunk: 8 of W_in, 4 of W_out
constexpr int ES_NSTAGE = ES_SPC * (DFF / 32);

// stage g = 12 hc + s of the weight stream; wave w brings fragment w of the stage
//   s < 8:  W_in K block s of tiles v (hc) and g (DFF / 32 + hc): fragments {vh, vl, gh, gl}
//   s >= 8: W_out K block 2 hc + (s - 8) / 2 of output tiles 2 th, 2 th + 1 (th = (s - 8) % 2): fragments 2 t + plane
__device__ __forceinline__ void es_request(int hc, int s, const W2& win, const W2& wout, unsigned ring_u, int wave,
                                           unsigned lane16) {  // s may run past the chunk (s < 2 ES_SPC): the next chunk's stage
    if (s >= ES_SPC) { s -= ES_SPC; hc += 1; }
    if (hc >= DFF / 32) { hc = DFF / 32 - 1; s = ES_SPC - 1; }  // past the end: the last stage again (identical bytes; keeps vmcnt uniform)
#ifdef AB_ABL_NODMA
    if (hc > 0 || s > 2) return;
#endif
    const unsigned dst = ring_u + (unsigned)((ES_SPC * hc + s) & (ES_NSLOT - 1)) * ES_SLOT + wave * 1024;
    if (s < 8) {
        const int tile = (wave >> 1) * (DFF / 32) + hc;
        ab_dma_piece((wave & 1) ? win.l : win.h, tile * (D / 16) + s, lane16, dst);
    } else {
        const int kb2 = (s - 8) >> 1, t = 2 * ((s - 8) & 1) + (wave >> 1);
        ab_dma_piece((wave & 1) ? wout.l : wout.h, t * (DFF / 16) + 2 * hc + kb2, lane16, dst);
    }
}
// this wave's fragment of the stage has landed (everything but its fragments of the two stages requested after it), then
// the workgroup barrier: the stage is complete for everybody, and everybody is done with the stage before it
#ifdef AB_ABL_NOBAR
#define ES_BARRIER()
#else
#define ES_BARRIER() __syncthreads()
#endif
// (vmcnt retires in order, stores included: behind the chunk's eight [v; g] store instructions -- issued between the requests of stages 12 hc + 10 and 12 hc + 11 -- the
// count is 2 + 8 for the three stages whose fragments were requested before them and are waited for after them: 12 hc + 8 .. + 10)
#define ES_STAGE_SYNC(AFTER_STORES)                                           \
    do {                                                                      \
        if ((AFTER_STORES) && !full) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   /* a partial tile skips store instructions */ \
        else if (AFTER_STORES) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");        \
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");                 \
        ES_BARRIER();                                                         \
    } while (0)
// accumulator tile initialised with 4096 x bias, the bias read through the SCALAR cache (wave-uniform addresses, both halves
// of a column group, selected by lane half): a vector load here would sit in the in-order vmcnt queue behind the ring requests
__device__ __forceinline__ void es_bias_tile(f32x16& acc, const float* __restrict__ b, int h) {
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float lo = b[8 * j + i], hi = b[8 * j + 4 + i];
            acc[4 * j + i] = (h ? hi : lo) * ABQ;
        }
}

template <bool LN>
__global__ __launch_bounds__(256, 2) void k_emlp_s(const float* __restrict__ X1, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, W2 win, const float* __restrict__ bin, W2 wout,
                                                const float* __restrict__ bout, float* __restrict__ VG, float* __restrict__ X2,
                                                int64_t E) {
    extern __shared__ __attribute__((aligned(16))) char es_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * ES_NW + wave) * WROWS;
    const bool live = row0 < E;
    if (!live) row0 = ((E - 1) / WROWS) * WROWS;  // run along on the last tile (same barriers), store nothing
    char* tile = es_smem + wave * 16384;
    const char* ring = es_smem + ES_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    dma_tile128(X1, row0, E, tile_u, L);
    es_request(0, 0, win, wout, ring_u, wave, lane16);
    es_request(0, 1, win, wout, ring_u, wave, lane16);
    es_request(0, 2, win, wout, ring_u, wave, lane16);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // the rows (the three ring requests may still be in flight)
    f32x16 out[4];
    f16x8 xph[8], xpl[8];
    {
        float4 x[16];
        tile128_to_frag(x, tile, L);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {  // 64 (x + b_out): the residual rides in the accumulator
                const float4 b4 = *reinterpret_cast<const float4*>(bout + 32 * t + 8 * j + 4 * L.h);
                out[t][4 * j] = (x[4 * t + j].x + b4.x) * ABS; out[t][4 * j + 1] = (x[4 * t + j].y + b4.y) * ABS;
                out[t][4 * j + 2] = (x[4 * t + j].z + b4.z) * ABS; out[t][4 * j + 3] = (x[4 * t + j].w + b4.w) * ABS;
            }
        norm_frag<16, LN>(x, gamma, beta, L.h);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {  // planes of 64 xn, kept in REGISTERS: every hidden chunk reads all of them (an eighth
            // of the kernel's LDS reads), and the row tile becomes the staging tile of the whole-line [v; g] stores
            const float v8[8] = {x[2 * kb].x * ABS, x[2 * kb].y * ABS, x[2 * kb].z * ABS, x[2 * kb].w * ABS,
                                 x[2 * kb + 1].x * ABS, x[2 * kb + 1].y * ABS, x[2 * kb + 1].z * ABS, x[2 * kb + 1].w * ABS};
            ab_split8(v8, xph[kb], xpl[kb]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float* const otile = reinterpret_cast<float*>(tile);  // [32][TILE32_LD] staging (trr.h store_tile32_lines)
    const bool full = row0 + WROWS <= E;        // every row of the tile exists: each store instruction has active lanes
    const bool stores = VG != nullptr && live;  // wave-uniform: the chunk's eight store instructions are issued

#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        f32x16 va, ga;
        es_bias_tile(va, bin + 32 * hc, L.h);
        es_bias_tile(ga, bin + DFF + 32 * hc, L.h);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int g = ES_SPC * hc + s;
            ES_STAGE_SYNC(false);
            es_request(hc, s + 3, win, wout, ring_u, wave, lane16);
            const char* slot = ring + (g & (ES_NSLOT - 1)) * ES_SLOT + lane16;
            const f16x8 wvh = *reinterpret_cast<const f16x8*>(slot + 0 * 1024);
            const f16x8 wvl = *reinterpret_cast<const f16x8*>(slot + 1 * 1024);
            const f16x8 wgh = *reinterpret_cast<const f16x8*>(slot + 2 * 1024);
            const f16x8 wgl = *reinterpret_cast<const f16x8*>(slot + 3 * 1024);
            AB_MFMA3(va, wvh, wvl, xph[s], xpl[s]);
            AB_MFMA3(ga, wgh, wgl, xph[s], xpl[s]);
        }
        // pre-activations, saved for the adjoint; u = v sigmoid(g) (transformer.py:42-43) as planes at scale 1
        f32x16 u;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            va[i] *= ABQ_INV;
            ga[i] *= ABQ_INV;
            u[i] = va[i] * sigm_(ga[i]);
        }
        if (stores) {  // whole 128-B lines through the staging tile (8 lanes per row)
            float4 t4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) t4[q] = make_float4(va[4 * q], va[4 * q + 1], va[4 * q + 2], va[4 * q + 3]);
            store_tile32_lines(t4, otile, VG + 32 * hc, row0, E, 2 * DFF, L);
#pragma unroll
            for (int q = 0; q < 4; q++) t4[q] = make_float4(ga[4 * q], ga[4 * q + 1], ga[4 * q + 2], ga[4 * q + 3]);
            store_tile32_lines(t4, otile, VG + DFF + 32 * hc, row0, E, 2 * DFF, L);
        }
        f16x8 uh[2], ul[2];
        ab_tile_planes(u, uh, ul);
#pragma unroll
        for (int s = 8; s < 12; s++) {
            const int g = ES_SPC * hc + s;
            ES_STAGE_SYNC(s < 11 && stores);  // (the fragments of stages 8 .. 10 were requested before the stores)
            es_request(hc, s + 3, win, wout, ring_u, wave, lane16);
            const char* slot = ring + (g & (ES_NSLOT - 1)) * ES_SLOT + lane16;
            const int kb2 = (s - 8) >> 1, th = (s - 8) & 1;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * t) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * t + 1) * 1024);
                AB_MFMA3(out[2 * th + t], wh, wl, uh[kb2], ul[kb2]);
            }
        }
    }
    // ---- X2 = out / 64: whole lines through the wave's own tile (the planes are dead)
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float* stg = reinterpret_cast<float*>(tile);  // [32][TILE_LD]
    const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
#pragma unroll
    for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                *reinterpret_cast<float4*>(stg + L.r * TILE_LD + 8 * (4 * t + j) + 4 * L.h) =
                    make_float4(out[2 * c + t][4 * j] * ABS_INV, out[2 * c + t][4 * j + 1] * ABS_INV,
                                out[2 * c + t][4 * j + 2] * ABS_INV, out[2 * c + t][4 * j + 3] * ABS_INV);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = 4 * j + rr;
            if (live && row0 + r < E)
                *reinterpret_cast<float4*>(X2 + (row0 + r) * D + 64 * c + cc) = *reinterpret_cast<const float4*>(stg + r * TILE_LD + cc);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// The adjoint in the same form, WITHOUT the saved pre-activations: [v; g] of a hidden chunk is recomputed from the layer
// input (48 MFMAs per chunk on top of the adjoint's 72), so the forward does not write 2 KB per row and this kernel does not
// read them -- and nothing but weight fragments is requested between a tile's first and last instruction (an HBM load in
// the middle of the stream would sit in front of every later fragment wait: vmcnt retires in order).
//   dX1 = dY + NormAdj(gamma . dxn, x1),  dxn = W_in^T [dv; dg],  dv = du s(g),  dg = du v s(g) (1 - s(g)),  du = W_out^T dY
// (transformer.py:39-50, 230-232 under autograd; k_emlp_bwd_p2's arithmetic with the recomputation of k_emlp_s in front). The
// norm's weight and bias are folded into W_in (abi.hip fold_norm_s: W diag(gamma), b + W beta), so xn = W-side and the kernel
// works on xhat: the recomputation's operand is also the xhat of the norm adjoint, and the layer input is read ONCE.
// Per wave: planes of 64 xn in registers, planes of 64 dY' (dY' = the row times the power of two that puts its largest entry
// in [1, 2)) in the wave's LDS tile, [dv | dg] as planes at scale 1. Stages of a chunk (four fragments each, one per wave):
//   0 .. 7    W_in K block s, tiles v / g                     va, ga += W xn           (6 MFMAs)
//   8 .. 11   W_out^T tile hc, K blocks 2 (s - 8), + 1        du += W^T dY'            (6)
//   12 .. 19  W_in^T K block kk = (s - 12) / 2 of the chunk (dv 0, 1; dg 0, 1), tiles 2 th, 2 th + 1: dn += W^T [dv | dg]  (6)
// GATHER: dY = dY[row] + dY2[rev2[row]] (rows of ldy floats), the ji gather of the combination adjoint (pet_bwd.hip).
// ---------------------------------------------------------------------------------------------
constexpr int EB_SPC = 20;
__device__ __forceinline__ void eb_request(int hc, int s, const W2& win, const W2& woutT, const W2& winT, unsigned ring_u, int wave,
                                           unsigned lane16) {
    if (s >= EB_SPC) { s -= EB_SPC; hc += 1; }
    if (hc >= DFF / 32) { hc = DFF / 32 - 1; s = EB_SPC - 1; }
#ifdef AB_ABL_NODMA
    if (hc > 0 || s > 2) return;
#endif
    const unsigned dst = ring_u + (unsigned)((EB_SPC * hc + s) & (ES_NSLOT - 1)) * ES_SLOT + wave * 1024;
    const int pl = wave & 1, j = wave >> 1;
    if (s < 8) {
        ab_dma_piece(pl ? win.l : win.h, (j * (DFF / 32) + hc) * (D / 16) + s, lane16, dst);
    } else if (s < 12) {  // W_out^T: tiles over the hidden units (DFF / 32), K = D: fragment (hc, kb)
        ab_dma_piece(pl ? woutT.l : woutT.h, hc * (D / 16) + 2 * (s - 8) + j, lane16, dst);
    } else {              // W_in^T: tiles over D (4), K = 2 DFF: K block of the chunk: dv -> 2 hc + b, dg -> DFF / 16 + 2 hc + b
        const int kk = (s - 12) >> 1, t = 2 * ((s - 12) & 1) + j;
        const int kb = (kk >> 1) * (DFF / 16) + 2 * hc + (kk & 1);
        ab_dma_piece(pl ? winT.l : winT.h, t * (2 * DFF / 16) + kb, lane16, dst);
    }
}

template <bool LN, bool GATHER>
__global__ __launch_bounds__(256, 2) void k_emlp_bwd_s(const float* __restrict__ dY, const float* __restrict__ X1, W2 win,
                                                       const float* __restrict__ bin, W2 woutT, W2 winT,
                                                       float* __restrict__ dX1, int64_t E, int ldy,
                                                       const float* __restrict__ dY2, const int* __restrict__ rev2) {
    extern __shared__ __attribute__((aligned(16))) char es_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t row0 = ((int64_t)blockIdx.x * ES_NW + wave) * WROWS;
    const bool live = row0 < E;
    if (!live) row0 = ((E - 1) / WROWS) * WROWS;
    const int64_t row = row0 + L.r < E ? row0 + L.r : E - 1;
    constexpr bool full = true;  // (ES_STAGE_SYNC: this kernel has no stores between its stages)
    char* tile = es_smem + wave * 16384;
    const char* ring = es_smem + ES_NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    // the layer input by LDS-DMA, the adjoint rows as row fragments straight into registers, the first ring stages
    dma_tile128(X1, row0, E, tile_u, L);
    float4 dy[16];
    load_rowfrag<16>(dy, dY, row, ldy, L.h);
    if (GATHER) {
        float4 d2[16];
        load_rowfrag<16>(d2, dY2, (int64_t)rev2[row], ldy, L.h);
#pragma unroll
        for (int k = 0; k < 16; k++) { dy[k].x += d2[k].x; dy[k].y += d2[k].y; dy[k].z += d2[k].z; dy[k].w += d2[k].w; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    eb_request(0, 0, win, woutT, winT, ring_u, wave, lane16);
    eb_request(0, 1, win, woutT, winT, ring_u, wave, lane16);
    eb_request(0, 2, win, woutT, winT, ring_u, wave, lane16);
    // xhat = the normalised row WITHOUT the norm's weight and bias (they are folded into W_in: Model::mlp_in_g), as planes of
    // 64 xhat in registers: the operand of the recomputation, and -- (H + L) / 64 -- the xhat of the norm adjoint at the end,
    // which therefore needs neither the layer input again nor gamma
    f16x8 xph[8], xpl[8];
    float rstd;
    {
        float4 x[16];
        tile128_to_frag(x, tile, L);
        if (LN) {
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) sm += (x[k].x + x[k].y) + (x[k].z + x[k].w);
            const float mean = row_sum(sm) * (1.0f / 128.0f);
#pragma unroll
            for (int k = 0; k < 16; k++) { x[k].x -= mean; x[k].y -= mean; x[k].z -= mean; x[k].w -= mean; }
        }
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) ss += x[k].x * x[k].x + x[k].y * x[k].y + x[k].z * x[k].z + x[k].w * x[k].w;
        rstd = rsqrtf(row_sum(ss) * (1.0f / 128.0f) + (LN ? 1e-5f : 1.1920928955078125e-07f));
        const float f = rstd * ABS;
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            const float v8[8] = {x[2 * kb].x * f, x[2 * kb].y * f, x[2 * kb].z * f, x[2 * kb].w * f,
                                 x[2 * kb + 1].x * f, x[2 * kb + 1].y * f, x[2 * kb + 1].z * f, x[2 * kb + 1].w * f};
            ab_split8(v8, xph[kb], xpl[kb]);
        }
    }
    float inv;
    {
        float sc;
        inv = row_scale_pow2<16>(dy, sc);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        ab_park_planes(dy, tile, L);  // planes of 64 dY' over the (consumed) input rows
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    f32x16 dn[4];
#pragma unroll
    for (int t = 0; t < 4; t++) dn[t] = ab_zero();

#pragma unroll 1
    for (int hc = 0; hc < DFF / 32; hc++) {
        f32x16 va, ga;
        es_bias_tile(va, bin + 32 * hc, L.h);
        es_bias_tile(ga, bin + DFF + 32 * hc, L.h);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            ES_STAGE_SYNC(false);
            eb_request(hc, s + 3, win, woutT, winT, ring_u, wave, lane16);
            const char* slot = ring + ((EB_SPC * hc + s) & (ES_NSLOT - 1)) * ES_SLOT + lane16;
            const f16x8 wvh = *reinterpret_cast<const f16x8*>(slot + 0 * 1024);
            const f16x8 wvl = *reinterpret_cast<const f16x8*>(slot + 1 * 1024);
            const f16x8 wgh = *reinterpret_cast<const f16x8*>(slot + 2 * 1024);
            const f16x8 wgl = *reinterpret_cast<const f16x8*>(slot + 3 * 1024);
            AB_MFMA3(va, wvh, wvl, xph[s], xpl[s]);
            AB_MFMA3(ga, wgh, wgl, xph[s], xpl[s]);
        }
        f32x16 du = ab_zero();
#pragma unroll
        for (int s = 8; s < 12; s++) {
            ES_STAGE_SYNC(false);
            eb_request(hc, s + 3, win, woutT, winT, ring_u, wave, lane16);
            const char* slot = ring + ((EB_SPC * hc + s) & (ES_NSLOT - 1)) * ES_SLOT + lane16;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int kb = 2 * (s - 8) + j;
                const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * j) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * j + 1) * 1024);
                const f16x8 yh = *reinterpret_cast<const f16x8*>(tile + ((kb * 2 + 0) * 64 + L.lane) * 16);
                const f16x8 yl = *reinterpret_cast<const f16x8*>(tile + ((kb * 2 + 1) * 64 + L.lane) * 16);
                AB_MFMA3(du, wh, wl, yh, yl);
            }
        }
        // SwiGLU adjoint; [dv | dg] as planes at scale 1 (K blocks dv 0, dv 1, dg 0, dg 1)
        f16x8 dh[4], dl[4];
        {
            f32x16 dv, dg;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float d = du[i] * ABQ_INV, v = va[i] * ABQ_INV, sg = sigm_(ga[i] * ABQ_INV);
                dv[i] = d * sg;
                dg[i] = d * v * sg * (1.f - sg);
            }
            f16x8 h2[2], l2[2];
            ab_tile_planes(dv, h2, l2);
            dh[0] = h2[0]; dh[1] = h2[1]; dl[0] = l2[0]; dl[1] = l2[1];
            ab_tile_planes(dg, h2, l2);
            dh[2] = h2[0]; dh[3] = h2[1]; dl[2] = l2[0]; dl[3] = l2[1];
        }
#pragma unroll
        for (int s = 12; s < 20; s++) {
            ES_STAGE_SYNC(false);
            eb_request(hc, s + 3, win, woutT, winT, ring_u, wave, lane16);
            const char* slot = ring + ((EB_SPC * hc + s) & (ES_NSLOT - 1)) * ES_SLOT + lane16;
            const int kk = (s - 12) >> 1, th = (s - 12) & 1;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * t) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * t + 1) * 1024);
                AB_MFMA3(dn[2 * th + t], wh, wl, dh[kk], dl[kk]);
            }
        }
    }
    // ---- epilogue: the norm adjoint on (w = d xhat-space adjoint, xhat from the planes), the residual dY' from ITS planes
    // (hi + lo keeps 22 bits of the pass-through adjoint: 2.4e-7 of the row's largest entry per layer, linear in the depth;
    // the fp32 rows would cost 64 registers the kernel does not have. tests/test_gpu_emlp_s.py runs a 3 x 3-layer model.)
    float4 w[16];
    {
        const float f = inv * ABS_INV;  // dn holds 64 x (W_in^T planes) of the scaled row
        float dot = 0.f, sw = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                w[4 * t + j] = make_float4(dn[t][4 * j] * f, dn[t][4 * j + 1] * f, dn[t][4 * j + 2] * f, dn[t][4 * j + 3] * f);
        float4 xh4[16];
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
            xh4[2 * kb] = make_float4(((float)xph[kb][0] + (float)xpl[kb][0]) * ABS_INV, ((float)xph[kb][1] + (float)xpl[kb][1]) * ABS_INV,
                                      ((float)xph[kb][2] + (float)xpl[kb][2]) * ABS_INV, ((float)xph[kb][3] + (float)xpl[kb][3]) * ABS_INV);
            xh4[2 * kb + 1] = make_float4(((float)xph[kb][4] + (float)xpl[kb][4]) * ABS_INV, ((float)xph[kb][5] + (float)xpl[kb][5]) * ABS_INV,
                                          ((float)xph[kb][6] + (float)xpl[kb][6]) * ABS_INV, ((float)xph[kb][7] + (float)xpl[kb][7]) * ABS_INV);
        }
#pragma unroll
        for (int k = 0; k < 16; k++)
            dot += xh4[k].x * w[k].x + xh4[k].y * w[k].y + xh4[k].z * w[k].z + xh4[k].w * w[k].w;
        const float md = row_sum(dot) * (1.0f / 128.0f);
#pragma unroll
        for (int k = 0; k < 16; k++) {  // rstd (w - xhat mean(xhat w)); LayerNorm: minus its mean
            w[k].x = rstd * (w[k].x - xh4[k].x * md); w[k].y = rstd * (w[k].y - xh4[k].y * md);
            w[k].z = rstd * (w[k].z - xh4[k].z * md); w[k].w = rstd * (w[k].w - xh4[k].w * md);
            sw += (w[k].x + w[k].y) + (w[k].z + w[k].w);
        }
        if (LN) {
            const float mw = row_sum(sw) * (1.0f / 128.0f);
#pragma unroll
            for (int k = 0; k < 16; k++) { w[k].x -= mw; w[k].y -= mw; w[k].z -= mw; w[k].w -= mw; }
        }
    }
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
        const f16x8 yh = *reinterpret_cast<const f16x8*>(tile + ((kb * 2 + 0) * 64 + L.lane) * 16);
        const f16x8 yl = *reinterpret_cast<const f16x8*>(tile + ((kb * 2 + 1) * 64 + L.lane) * 16);
        const float f = inv * ABS_INV;
        w[2 * kb].x += ((float)yh[0] + (float)yl[0]) * f; w[2 * kb].y += ((float)yh[1] + (float)yl[1]) * f;
        w[2 * kb].z += ((float)yh[2] + (float)yl[2]) * f; w[2 * kb].w += ((float)yh[3] + (float)yl[3]) * f;
        w[2 * kb + 1].x += ((float)yh[4] + (float)yl[4]) * f; w[2 * kb + 1].y += ((float)yh[5] + (float)yl[5]) * f;
        w[2 * kb + 1].z += ((float)yh[6] + (float)yl[6]) * f; w[2 * kb + 1].w += ((float)yh[7] + (float)yl[7]) * f;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    store_rows_lines<16>(w, reinterpret_cast<float*>(tile), L, [&](int r) { return live && row0 + r < E ? dX1 + (row0 + r) * D : nullptr; });
}

// pet_config_set("emlp_s", v): 0 = the one-wave-per-SIMD pipelined kernels k_emlp_p2 / k_emlp_bwd_p2 everywhere; 1 = these
// kernels for graphs of at least 28 672 edge rows (default: the measured crossover -- 1 000 atoms, 19 k rows: the pipelined
// kernels 2 % ahead; 2 000 atoms, 38 k rows: these 1 % ahead); v > 1 = from v rows on (the tests force small graphs through)
static int g_emlp_s = 1;
static int64_t g_es_min_rows = 28672;
void set_emlp_s(int v) {
    g_emlp_s = v ? 1 : 0;
    g_es_min_rows = v > 1 ? v : 28672;
}
bool emlp_s_serves(int64_t E) { return g_emlp_s && E >= g_es_min_rows; }
bool emlp_s_forced() { return g_emlp_s && g_es_min_rows < 28672; }  // pet_config_set("emlp_s", v > 1): the tests' small graphs  // (the edge head's kernels follow the same policy: pet_head_s.hip)
bool emlp_recompute_on(const Lin& win, const Lin& wout, int64_t E) {
    return g_emlp_s && E >= g_es_min_rows && win.fwd2s && wout.fwd2s && wout.bwd2s;
}

static inline W2 es_w2(const void* base, int n_out, int k_in) {
    const size_t n8 = (size_t)(n_out / 32) * (k_in / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(base);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}

// false = not served (weights not packed for it, or switched off)
bool emlp_s(const float* X1, const float* gamma, const float* beta, const Lin& win, const Lin& wout, float* VG, float* X2,
            int64_t E, hipStream_t st) {
    if (!g_emlp_s || !win.fwd2s || !wout.fwd2s || E < g_es_min_rows) return false;
    const size_t lds = ES_NW * 16384 + ES_NSLOT * ES_SLOT;
    const W2 wi = es_w2(win.fwd2s, win.n_out, win.k_in), wo = es_w2(wout.fwd2s, wout.n_out, wout.k_in);
    const int grid = (int)cdiv(E, ES_NW * WROWS);
    if (beta) {
        allow_big_lds(k_emlp_s<true>, lds);
        k_emlp_s<true><<<grid, 256, lds, st>>>(X1, gamma, beta, wi, win.b, wo, wout.b, VG, X2, E);
    } else {
        allow_big_lds(k_emlp_s<false>, lds);
        k_emlp_s<false><<<grid, 256, lds, st>>>(X1, gamma, beta, wi, win.b, wo, wout.b, VG, X2, E);
    }
    return true;
}

// the adjoint with recomputed pre-activations; false = not served
bool emlp_bwd_s(const float* dY, const float* X1, bool ln, const Lin& win_g, const Lin& wout, float* dX1, int64_t E,
                hipStream_t st, int ldy, const float* dY2, const int* rev2) {
    if (!g_emlp_s || !win_g.fwd2s || !win_g.bwd2s || !wout.bwd2s) return false;
    if (E <= 0) return true;
    const size_t lds = ES_NW * 16384 + ES_NSLOT * ES_SLOT;
    const W2 wi = es_w2(win_g.fwd2s, win_g.n_out, win_g.k_in), wot = es_w2(wout.bwd2s, wout.n_out, wout.k_in),
             wit = es_w2(win_g.bwd2s, win_g.n_out, win_g.k_in);
    const int grid = (int)cdiv(E, ES_NW * WROWS);
#define PET_EB(LNF, GF)                                                                                        \
    {                                                                                                          \
        allow_big_lds(k_emlp_bwd_s<LNF, GF>, lds);                                                             \
        k_emlp_bwd_s<LNF, GF><<<grid, 256, lds, st>>>(dY, X1, wi, win_g.b, wot, wit, dX1, E, ldy, dY2, rev2);  \
    }
    if (ln) { if (dY2) PET_EB(true, true) else PET_EB(true, false) }
    else { if (dY2) PET_EB(false, true) else PET_EB(false, false) }
#undef PET_EB
    return true;
}

}  // namespace pet
