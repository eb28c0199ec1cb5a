// Per-atom fused attention block (round 4): one WAVE owns the tokens of one atom (centre token + its neighbours, at most
// 32 NQ) and runs  norm -> QKV -> soft-max attention -> output projection (+ residual)  on them without ever writing
// Q, K, V or the attention output to HBM; the adjoint recomputes Q, K, V from the saved layer input in the same way and
// emits only the gradient of the layer input. Reference: pet/modules/transformer.py:86-152 (AttentionBlock.forward),
// :203-234 (the PreLN layer around it), :565-589 (manual_attention).
//
// Everything is v_mfma_f32_32x32x16_f16 on split operands and the C/D layout of one product IS the operand layout of
// the next, so the chain needs no LDS exchange and no cross-lane traffic except the soft-max row statistics:
//
//   token form   (lane = token,   regs = features)  Q^T, K^T   = W x^T        A = weight fragment, B = row planes
//   feature form (lane = feature, regs = tokens)    V          = x W^T        A = row planes,      B = weight fragment
//   S^T [key, query]   = K Q^T   A = K (token form, one head = 8 regs),  B = Q (token form)     -> lane = query
//   soft-max over the keys = over the 16 regs of a lane and its partner lane (xor 32)
//   O^T [feature, query] = V^T P^T   A = V (feature form, K blocks = 8 regs of tokens),  B = P^T (the S^T registers)
//                        -> lane = query = token, regs = features: the row fragment of the attention output,
//                           i.e. directly the B operand of the output projection X1^T = Wo AO^T.
//
// Split operands, ONE accumulator per product ("f16x3"): every tensor -- weights, activations, adjoints -- is held as
// two fp16 planes of 64 x,
//   H = fp16(64 x),    L = fp16(64 x - H)      (L is a normal fp16 number for |x| > 4e-3; below that the pair still
//                                               carries x to 5e-10 absolute: the matrix cores honour fp16 subnormals,
//                                               tools/debug/mfma_denorm.hip)
// and a product is three MFMAs on one accumulator that then holds 4096 a b:
//   4096 a b = a_H b_H + a_L b_H + a_H b_L          (+ a_L b_L: 2^-22 relative, dropped).
// |64 x| must stay inside fp16: |x| < 1023, which holds for normalised rows, weights, Q / K / V (sums of normalised rows
// times weights), soft-max weights and attention outputs; adjoint rows are scaled per atom by a power of two first.
// The weight planes are packed by abi.hip (Lin::fwd2s / bwd2s).
#include "ablk_bwd.h"

namespace pet {

// ---------------------------------------------------------------------------------------------
// The weight stream, shared by the waves of a workgroup. Every wave needs the same weight fragments at (about) the
// same time -- a tile is 32 token slots whatever the atom's size -- and a wave of its own streams 260 KB of them from
// L2 per tile: measured, the forward's QKV phase took exactly as long with its MFMAs removed as with them (the L2 ->
// CU path was the bound, 25 TB/s). So the workgroup fetches each "stage" (a few KB of fragments) ONCE, by LDS-DMA into
// a two-slot ring; fragments are lane-linear 1-KB pieces, which is exactly the LDS image a ds_read_b128 per lane wants.
//   stage g:  wait for the own pieces of stage g  ->  barrier (stage g landed for everybody, everybody is done with
//             stage g - 1's slot)  ->  request stage g + 1 into that slot  ->  the products of stage g.
// One barrier per 18 .. 24 MFMAs of a wave; waves whose tile index is past the list run along on the last tile (same
// barrier count) and store nothing.
// ---------------------------------------------------------------------------------------------
constexpr int AB_SLOT = 12288;  // bytes of a ring slot: two QKV K blocks (2 x 6 fragments) or two Wo steps (2 x 4)
// forward stages 0 .. 15: QKV blocks (hp = g / 4, kb = 2 (g % 4) + j), pieces j * 6 + {Qh, Ql, Kh, Kl, Vh, Vl};
// stages 16 .. 23: Wo steps n = 2 (g - 16) + j (c = n / 8, kb = n % 8), pieces j * 4 + {tile 2c h, l, tile 2c + 1 h, l}
template <int NW>
__device__ __forceinline__ void ab_fwd_request(int g, const W2& wqkv, const W2& wo, unsigned ring_u, int wave,
                                               unsigned lane16) {
#ifdef AB_ABL_NODMA  // timing ablation (results are wrong): the weight stream stops after its first stage
    if (g > 0) return;
#endif
    const unsigned dst = ring_u + (unsigned)(g & 1) * AB_SLOT;
    if (g < 16) {
        const int hp = g >> 2, kb0 = 2 * (g & 3);
#pragma unroll
        for (int p0 = 0; p0 < 12; p0 += NW) {
            const int p = p0 + wave;
            if (p < 12) {
                const int j = p / 6, f = p % 6;
                ab_dma_piece((f & 1) ? wqkv.l : wqkv.h, 32 * (f >> 1) + hp * 8 + kb0 + j, lane16, dst + p * 1024);
            }
        }
    } else {
        const int n0 = 2 * (g - 16);
#pragma unroll
        for (int p = wave; p < 8; p += NW) {  // 8 pieces
            const int n = n0 + (p >> 2), f = p & 3;
            ab_dma_piece((f & 1) ? wo.l : wo.h, (2 * (n >> 3) + (f >> 1)) * 8 + (n & 7), lane16, dst + p * 1024);
        }
    }
}
// (the adjoint's ring stages and the adjoint kernel itself: ablk_bwd.h)

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int NQ, bool LN>
__global__ __launch_bounds__(256) void k_ablk_fwd(
    const float* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta, W2 wqkv,
    const float* __restrict__ bqkv, W2 wo, const float* __restrict__ bo, const float* __restrict__ fc,
    const int4* __restrict__ desc, int n_list, int64_t E, float qscale, float* __restrict__ X1, float* __restrict__ OC) {
    constexpr int NW = 4;  // waves per workgroup (instantiated for the 64-slot tiles, NQ = 2; the 32-slot tiles: k_ablk_fwd4):
                           // NW x NQ x 16 KB of row planes + the 24 KB ring
    extern __shared__ __attribute__((aligned(16))) char ab_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int li = blockIdx.x * NW + wave;
    const bool live = li < n_list;
    li = live ? li : n_list - 1;
    const AbAtom a(desc + 2 * (size_t)li, E);
    char* tile = ab_smem + wave * (NQ * 16384);
    const char* ring = ab_smem + NW * NQ * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    AB_T0();
    ab_dma_rows<NQ>(X, a, tile_u, L);
    ab_fwd_request<NW>(0, wqkv, wo, ring_u, wave, lane16);
    float bias[NQ][16];
    ab_key_bias<NQ>(bias, a, fc, L);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AB_T(0);
    // rows -> normalised -> planes, parked over the fp32 tile they came from
#pragma unroll
    for (int tq = 0; tq < NQ; tq++) {
        float4 x[16];
        tile128_to_frag(x, tile + tq * 16384, L);
        norm_frag<16, LN>(x, gamma, beta, L.h);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        ab_park_planes(x, tile + tq * 16384, L);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");

    AB_T(1);
    f16x8 aoh[NQ][8], aol[NQ][8];  // attention output: planes of the row fragment, K block = head
#pragma unroll
    for (int hp = 0; hp < 4; hp++) {  // unrolled: the planes are indexed with hp (a run-time index would go to scratch)
        // ---- Q^T, K^T (token form) and V (feature form) of the head pair: 32 features each
        f32x16 q[NQ], k[NQ], v[NQ];
        {
            const float bv = bqkv[2 * D + 32 * hp + L.r] * ABQ;
#pragma unroll
            for (int tq = 0; tq < NQ; tq++) {
                ab_bias_tile(q[tq], bqkv + 32 * hp, L.h);
                ab_bias_tile(k[tq], bqkv + D + 32 * hp, L.h);
#pragma unroll
                for (int i = 0; i < 16; i++) v[tq][i] = bv;
            }
        }
#pragma unroll
        for (int sg = 0; sg < 4; sg++) {
            const int g = 4 * hp + sg;
            AB_STAGE_SYNC();
            ab_fwd_request<NW>(g + 1, wqkv, wo, ring_u, wave, lane16);
            const char* slot = ring + (g & 1) * AB_SLOT + lane16;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int kb = 2 * sg + j;
                const f16x8 wqh = *reinterpret_cast<const f16x8*>(slot + (6 * j + 0) * 1024);
                const f16x8 wql = *reinterpret_cast<const f16x8*>(slot + (6 * j + 1) * 1024);
                const f16x8 wkh = *reinterpret_cast<const f16x8*>(slot + (6 * j + 2) * 1024);
                const f16x8 wkl = *reinterpret_cast<const f16x8*>(slot + (6 * j + 3) * 1024);
                const f16x8 wvh = *reinterpret_cast<const f16x8*>(slot + (6 * j + 4) * 1024);
                const f16x8 wvl = *reinterpret_cast<const f16x8*>(slot + (6 * j + 5) * 1024);
#pragma unroll
                for (int tq = 0; tq < NQ; tq++) {
                    const char* tp = tile + tq * 16384;
                    const f16x8 xh = *reinterpret_cast<const f16x8*>(tp + ((kb * 2 + 0) * 64 + L.lane) * 16);
                    const f16x8 xl = *reinterpret_cast<const f16x8*>(tp + ((kb * 2 + 1) * 64 + L.lane) * 16);
                    AB_MFMA3(q[tq], wqh, wql, xh, xl);
                    AB_MFMA3(k[tq], wkh, wkl, xh, xl);
                    AB_MFMA3(v[tq], xh, xl, wvh, wvl);
                }
            }
        }
        AB_T(2);
        // ---- operand planes of the attention products (the accumulators hold 4096 x the value, the planes 64 x)
        f16x8 qh[NQ][2], ql[NQ][2], kH[NQ][2], kL[NQ][2], vH[NQ][2], vL[NQ][2];
#pragma unroll
        for (int tq = 0; tq < NQ; tq++) {
            ab_tile_planes(q[tq], qscale * ABS_INV, qh[tq], ql[tq]);
            ab_tile_planes(k[tq], ABS_INV, kH[tq], kL[tq]);
            ab_tile_planes(v[tq], ABS_INV, vH[tq], vL[tq]);
        }
        AB_T(3);
        // ---- the two heads of the pair, side by side (independent chains for the scheduler to interleave)
#pragma unroll
        for (int tq = 0; tq < NQ; tq++) {
            f32x16 s[2][NQ];
#pragma unroll
            for (int hd = 0; hd < 2; hd++)
#pragma unroll
                for (int tk = 0; tk < NQ; tk++) {
                    s[hd][tk] = ab_zero();
                    AB_MFMA3(s[hd][tk], kH[tk][hd], kL[tk][hd], qh[tq][hd], ql[tq][hd]);
                }
            float mx[2], sum[2];
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                mx[hd] = -INFINITY;
#pragma unroll
                for (int tk = 0; tk < NQ; tk++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        s[hd][tk][i] = fmaf(s[hd][tk][i], ABQ_INV, bias[tk][i]);
                        mx[hd] = fmaxf(mx[hd], s[hd][tk][i]);
                    }
            }
#pragma unroll
            for (int hd = 0; hd < 2; hd++) mx[hd] = fmaxf(mx[hd], __shfl_xor(mx[hd], 32)) - 6.0f;  // p comes out as 64 p
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                sum[hd] = 0.f;
#pragma unroll
                for (int tk = 0; tk < NQ; tk++)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float p = __builtin_amdgcn_exp2f(s[hd][tk][i] - mx[hd]);
                        s[hd][tk][i] = p;
                        sum[hd] += p;
                    }
            }
#pragma unroll
            for (int hd = 0; hd < 2; hd++) sum[hd] += __shfl_xor(sum[hd], 32);  // 64 x the soft-max denominator
            f32x16 o[2];
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                o[hd] = ab_zero();
#pragma unroll
                for (int tk = 0; tk < NQ; tk++) {
                    f16x8 ph[2], pl[2];
                    ab_tile_planes(s[hd][tk], ph, pl);
#pragma unroll
                    for (int b = 0; b < 2; b++) AB_MFMA3(o[hd], vH[tk][b], vL[tk][b], ph[b], pl[b]);
                }
            }
#pragma unroll
            for (int hd = 0; hd < 2; hd++) {
                const float inv = __builtin_amdgcn_rcpf(sum[hd]);  // o = 4096 sum(V 64 p): o / (64 denominator) = 64 AO
                float t8[8];
                ab_regs8(o[hd], hd, inv, t8);
                ab_split8(t8, aoh[tq][2 * hp + hd], aol[tq][2 * hp + hd]);
            }
        }
        AB_T(4);
    }
    // ---- output projection, bias, residual: X1 = X + Wo AO + bo (edge rows); OC = Wo AO + bo (the centre token).
    // Both token tiles share a stage's fragments; the staging tile for whole-line stores is the wave's own (dead) planes.
    float* stg = reinterpret_cast<float*>(tile);  // [32][TILE_LD]
    const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
#pragma unroll
    for (int c = 0; c < 2; c++) {  // 64 output features at a time
        float4 xr[NQ][8];  // the residual rows of this half in the store's shape, requested before the products
#pragma unroll
        for (int tq = 0; tq < NQ; tq++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int s = 32 * tq + 4 * j + rr;
                xr[tq][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (s < a.T && !a.centre(s)) xr[tq][j] = *reinterpret_cast<const float4*>(X + a.edge(s) * D + 64 * c + cc);
            }
        f32x16 y[NQ][2];
#pragma unroll
        for (int tq = 0; tq < NQ; tq++) {
            ab_bias_tile(y[tq][0], bo + 64 * c, L.h);
            ab_bias_tile(y[tq][1], bo + 64 * c + 32, L.h);
        }
#pragma unroll
        for (int sg = 0; sg < 4; sg++) {
            const int g = 16 + 4 * c + sg;
            AB_STAGE_SYNC();
            if (g + 1 < 24) ab_fwd_request<NW>(g + 1, wqkv, wo, ring_u, wave, lane16);
            const char* slot = ring + (g & 1) * AB_SLOT + lane16;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int kb = 2 * sg + j;
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    wh[t] = *reinterpret_cast<const f16x8*>(slot + (4 * j + 2 * t) * 1024);
                    wl[t] = *reinterpret_cast<const f16x8*>(slot + (4 * j + 2 * t + 1) * 1024);
                }
#pragma unroll
                for (int tq = 0; tq < NQ; tq++)
#pragma unroll
                    for (int t = 0; t < 2; t++) AB_MFMA3(y[tq][t], wh[t], wl[t], aoh[tq][kb], aol[tq][kb]);
            }
        }
        AB_T(5);
#pragma unroll
        for (int tq = 0; tq < NQ; tq++) {
            if (32 * tq >= a.T) continue;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *reinterpret_cast<float4*>(stg + L.r * TILE_LD + 8 * (4 * t + j) + 4 * L.h) =
                        make_float4(y[tq][t][4 * j] * ABQ_INV, y[tq][t][4 * j + 1] * ABQ_INV, y[tq][t][4 * j + 2] * ABQ_INV,
                                    y[tq][t][4 * j + 3] * ABQ_INV);
            __builtin_amdgcn_wave_barrier();
            // whole lines out: 16 lanes per row, four rows per instruction
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = 4 * j + rr, s = 32 * tq + r;
                if (live && s < a.T) {
                    float4 o4 = *reinterpret_cast<const float4*>(stg + r * TILE_LD + cc);
                    o4.x += xr[tq][j].x; o4.y += xr[tq][j].y; o4.z += xr[tq][j].z; o4.w += xr[tq][j].w;
                    float* dst = a.centre(s) ? OC + (int64_t)a.atom(s) * D : X1 + a.edge(s) * D;
                    *reinterpret_cast<float4*>(dst + 64 * c + cc) = o4;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        AB_T(6);
    }
}

// ---------------------------------------------------------------------------------------------
// k_ablk_fwd4 (round 5): the same forward for the 32-slot tiles as TWO desynchronised four-wave workgroups per CU instead of
// one eight-wave workgroup. In k_ablk_fwd the eight waves meet at every stage barrier, so the two waves of a SIMD run their
// matrix phases and their vector phases at the same time; two independent workgroups drift apart and fill each other's
// stalls (what made k_emlp_s faster than k_emlp_p2, pet_emlp_s.hip). Each workgroup has its own weight ring: four slots
// of four fragments (one per wave and stage), requested three stages ahead:
//   stage g = 12 hp + r (r = 0 .. 11): fragments 4 r .. 4 r + 3 of the head pair's 48 = [K block kb][Qh Ql Kh Kl Vh Vl]
//                                       (r % 3 = 0: Q, K of kb = 2 (r / 3); 1: V of that kb, Q of the next; 2: K, V of the next)
//   stage g = 48 + n (n = 0 .. 15):     W_o step n (c = n / 8, kb = n % 8): tiles 2 c, 2 c + 1 x (h, l)
// Every stage is six MFMAs per wave. Both halves of the output projection are accumulated before anything is stored, so no
// load or store sits between two stages (vmcnt retires in order: the count at a stage boundary is always the two fragments
// requested after the one waited for).
// ---------------------------------------------------------------------------------------------
constexpr int AB4_SLOT = 4096, AB4_NSLOT = 4, AB4_NSTAGE = 64;
__device__ __forceinline__ void ab4_request(int g, const W2& wqkv, const W2& wo, unsigned ring_u, int wave, unsigned lane16) {
    g = g < AB4_NSTAGE ? g : AB4_NSTAGE - 1;  // past the end: the last stage again (identical bytes; keeps vmcnt uniform)
    const unsigned dst = ring_u + (unsigned)(g & (AB4_NSLOT - 1)) * AB4_SLOT + wave * 1024;
    if (g < 48) {
        const int hp = g / 12, pc = 4 * (g % 12) + wave, kb = pc / 6, f = pc % 6;
        ab_dma_piece((f & 1) ? wqkv.l : wqkv.h, 32 * (f >> 1) + hp * 8 + kb, lane16, dst);
    } else {
        const int n = g - 48;
        ab_dma_piece((wave & 1) ? wo.l : wo.h, (2 * (n >> 3) + (wave >> 1)) * 8 + (n & 7), lane16, dst);
    }
}
#define AB4_STAGE_SYNC()                                  \
    do {                                                  \
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");  \
        __syncthreads();                                  \
    } while (0)

template <bool LN>
__global__ __launch_bounds__(256, 2) void k_ablk_fwd4(
    const float* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta, W2 wqkv,
    const float* __restrict__ bqkv, W2 wo, const float* __restrict__ bo, const float* __restrict__ fc,
    const int4* __restrict__ desc, int n_list, int64_t E, float qscale, float* __restrict__ X1, float* __restrict__ OC) {
    constexpr int NW = 4;
    extern __shared__ __attribute__((aligned(16))) char ab_smem[];
    const RowLane L;
    const unsigned lane16 = (unsigned)L.lane * 16u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int li = blockIdx.x * NW + wave;
    const bool live = li < n_list;
    li = live ? li : n_list - 1;
    const AbAtom a(desc + 2 * (size_t)li, E);
    char* tile = ab_smem + wave * 16384;
    const char* ring = ab_smem + NW * 16384;
    const unsigned tile_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile);
    const unsigned ring_u = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
    float bias[1][16];
    ab_key_bias<1>(bias, a, fc, L);  // (its loads are consumed before the requests below: nothing but fragments in the queue)
    asm volatile("" ::"v"(bias[0][0]), "v"(bias[0][15]));
    ab_dma_rows<1>(X, a, tile_u, L);
    ab4_request(0, wqkv, wo, ring_u, wave, lane16);
    ab4_request(1, wqkv, wo, ring_u, wave, lane16);
    ab4_request(2, wqkv, wo, ring_u, wave, lane16);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // the rows
    {
        float4 x[16];
        tile128_to_frag(x, tile, L);
        norm_frag<16, LN>(x, gamma, beta, L.h);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        ab_park_planes(x, tile, L);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");

    f16x8 aoh[8], aol[8];  // attention output: planes of the row fragment, K block = head
#pragma unroll
    for (int hp = 0; hp < 4; hp++) {
        f32x16 q, k, v;
        {
            const float bv = bqkv[2 * D + 32 * hp + L.r] * ABQ;
            ab_bias_tile(q, bqkv + 32 * hp, L.h);
            ab_bias_tile(k, bqkv + D + 32 * hp, L.h);
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = bv;
        }
        asm volatile("" ::"v"(q[0]), "v"(k[0]), "v"(v[0]));  // the bias loads are consumed before the stage waits
        f16x8 xh, xl;
#pragma unroll
        for (int r = 0; r < 12; r++) {
            const int g = 12 * hp + r;
            AB4_STAGE_SYNC();
            ab4_request(g + 3, wqkv, wo, ring_u, wave, lane16);
            const char* slot = ring + (g & (AB4_NSLOT - 1)) * AB4_SLOT + lane16;
            const f16x8 f0 = *reinterpret_cast<const f16x8*>(slot + 0 * 1024);
            const f16x8 f1 = *reinterpret_cast<const f16x8*>(slot + 1 * 1024);
            const f16x8 f2 = *reinterpret_cast<const f16x8*>(slot + 2 * 1024);
            const f16x8 f3 = *reinterpret_cast<const f16x8*>(slot + 3 * 1024);
            const int m = r / 3;
            if (r % 3 == 0) {  // Q, K of K block 2 m
                xh = *reinterpret_cast<const f16x8*>(tile + (((2 * m) * 2 + 0) * 64 + L.lane) * 16);
                xl = *reinterpret_cast<const f16x8*>(tile + (((2 * m) * 2 + 1) * 64 + L.lane) * 16);
                AB_MFMA3(q, f0, f1, xh, xl);
                AB_MFMA3(k, f2, f3, xh, xl);
            } else if (r % 3 == 1) {  // V of K block 2 m, Q of 2 m + 1
                AB_MFMA3(v, xh, xl, f0, f1);
                xh = *reinterpret_cast<const f16x8*>(tile + (((2 * m + 1) * 2 + 0) * 64 + L.lane) * 16);
                xl = *reinterpret_cast<const f16x8*>(tile + (((2 * m + 1) * 2 + 1) * 64 + L.lane) * 16);
                AB_MFMA3(q, f2, f3, xh, xl);
            } else {  // K, V of K block 2 m + 1
                AB_MFMA3(k, f0, f1, xh, xl);
                AB_MFMA3(v, xh, xl, f2, f3);
            }
        }
        // ---- the attention core of the head pair: exactly k_ablk_fwd's
        f16x8 qh[2], ql[2], kH[2], kL[2], vH[2], vL[2];
        ab_tile_planes(q, qscale * ABS_INV, qh, ql);
        ab_tile_planes(k, ABS_INV, kH, kL);
        ab_tile_planes(v, ABS_INV, vH, vL);
        f32x16 s[2];
#pragma unroll
        for (int hd = 0; hd < 2; hd++) {
            s[hd] = ab_zero();
            AB_MFMA3(s[hd], kH[hd], kL[hd], qh[hd], ql[hd]);
        }
        float mx[2], sum[2];
#pragma unroll
        for (int hd = 0; hd < 2; hd++) {
            mx[hd] = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                s[hd][i] = fmaf(s[hd][i], ABQ_INV, bias[0][i]);
                mx[hd] = fmaxf(mx[hd], s[hd][i]);
            }
        }
#pragma unroll
        for (int hd = 0; hd < 2; hd++) mx[hd] = fmaxf(mx[hd], __shfl_xor(mx[hd], 32)) - 6.0f;  // p comes out as 64 p
#pragma unroll
        for (int hd = 0; hd < 2; hd++) {
            sum[hd] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float pe = __builtin_amdgcn_exp2f(s[hd][i] - mx[hd]);
                s[hd][i] = pe;
                sum[hd] += pe;
            }
        }
#pragma unroll
        for (int hd = 0; hd < 2; hd++) sum[hd] += __shfl_xor(sum[hd], 32);
        f32x16 o[2];
#pragma unroll
        for (int hd = 0; hd < 2; hd++) {
            o[hd] = ab_zero();
            f16x8 ph[2], pl[2];
            ab_tile_planes(s[hd], ph, pl);
#pragma unroll
            for (int b = 0; b < 2; b++) AB_MFMA3(o[hd], vH[b], vL[b], ph[b], pl[b]);
        }
#pragma unroll
        for (int hd = 0; hd < 2; hd++) {
            const float inv = __builtin_amdgcn_rcpf(sum[hd]);
            float t8[8];
            ab_regs8(o[hd], hd, inv, t8);
            ab_split8(t8, aoh[2 * hp + hd], aol[2 * hp + hd]);
        }
    }
    // ---- output projection: both 64-column halves accumulated before the stores
    f32x16 y[2][2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        ab_bias_tile(y[c][0], bo + 64 * c, L.h);
        ab_bias_tile(y[c][1], bo + 64 * c + 32, L.h);
    }
    asm volatile("" ::"v"(y[0][0][0]), "v"(y[0][1][0]), "v"(y[1][0][0]), "v"(y[1][1][0]));
#pragma unroll
    for (int n = 0; n < 16; n++) {
        const int g = 48 + n, c = n >> 3, kb = n & 7;
        AB4_STAGE_SYNC();
        ab4_request(g + 3, wqkv, wo, ring_u, wave, lane16);
        const char* slot = ring + (g & (AB4_NSLOT - 1)) * AB4_SLOT + lane16;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(slot + (2 * t) * 1024);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(slot + (2 * t + 1) * 1024);
            AB_MFMA3(y[c][t], wh, wl, aoh[kb], aol[kb]);
        }
    }
    // ---- bias is in, residual, whole-line stores: X1 = X + Wo AO + bo (edge rows); OC = Wo AO + bo (the centre token)
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    float* stg = reinterpret_cast<float*>(tile);  // [32][TILE_LD]
    const int rr = L.lane >> 4, cc = 4 * (L.lane & 15);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        float4 xr[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int sl = 4 * j + rr;
            xr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sl < a.T && !a.centre(sl)) xr[j] = *reinterpret_cast<const float4*>(X + a.edge(sl) * D + 64 * c + cc);
        }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                *reinterpret_cast<float4*>(stg + L.r * TILE_LD + 8 * (4 * t + j) + 4 * L.h) =
                    make_float4(y[c][t][4 * j] * ABQ_INV, y[c][t][4 * j + 1] * ABQ_INV, y[c][t][4 * j + 2] * ABQ_INV,
                                y[c][t][4 * j + 3] * ABQ_INV);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = 4 * j + rr;
            if (live && r < a.T) {
                float4 o4 = *reinterpret_cast<const float4*>(stg + r * TILE_LD + cc);
                o4.x += xr[j].x; o4.y += xr[j].y; o4.z += xr[j].z; o4.w += xr[j].w;
                float* dst = a.centre(r) ? OC + (int64_t)a.atom(r) * D : X1 + a.edge(r) * D;
                *reinterpret_cast<float4*>(dst + 64 * c + cc) = o4;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// pet_config_set("attn_fused", bits): 1 = fused forward, 2 = fused adjoint (the forward of a call sequence whose adjoint
// follows is fused only together with it: the three-kernel adjoint reads the saved Q, K, V), 4 = whatever the graph
// (by default graphs of fewer than ABLK_MIN_TILES = 3 840 tiles, and graphs in which more than 5 % of the atoms have more than 32 tokens,
// take the three-kernel form: ablk_serves); 0 = the three-kernel form (QKV / attention / projection) everywhere
static int g_attn_fused = 3;
void set_attn_fused(int v) { g_attn_fused = v; }
int attn_fused() { return g_attn_fused; }

#ifdef AB_PROFILE
__device__ unsigned long long ab_prof[32];
void ablk_prof_dump() {
    unsigned long long h[32];
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(ab_prof), sizeof(h));
    for (int i = 0; i < 32; i++)
        if (h[i]) fprintf(stderr, "ab_prof[%d] = %.3f Mcycles\n", i, h[i] * 1e-6);
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ab_prof), z, sizeof(z));
}
#else
void ablk_prof_dump() {}
#endif

static inline W2 w2s_fwd(const Lin& L) {
    const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(L.fwd2s);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}
static inline W2 w2s_bwd(const Lin& L) {
    const size_t n8 = (size_t)(L.n_out / 32) * (L.k_in / 16) * 64;
    const f16x8* b = reinterpret_cast<const f16x8*>(L.bwd2s);
    W2 w; w.h = b; w.l = b + n8;
    return w;
}

// The few atoms of 33 .. 64 tokens run the NQ = 2 instantiation: a grid of a few hundred waves whose run time is one
// wave's latency (70 / 200 us). On the caller's stream it would stand in front of the NQ = 1 launch; on a stream of its
// own it runs beside it (both only read the layer input and write disjoint rows).
struct TailStream {
    hipStream_t s = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    bool ok = false;
};
static TailStream& tail_stream() {
    static TailStream t;
    static bool init = false;
    if (!init) {
        init = true;
        const char* pr = getenv("PET_HIP_TAIL_PRIO");  // low | high | (default) the caller's level
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const int prio = pr && pr[0] == 'l' ? least : (pr && pr[0] == 'h' ? greatest : 0);
        t.ok = hipStreamCreateWithPriority(&t.s, hipStreamNonBlocking, prio) == hipSuccess &&
               hipEventCreateWithFlags(&t.fork_ev, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&t.join_ev, hipEventDisableTiming) == hipSuccess;
    }
    return t;
}
// stream for the NQ = 2 launch (forked from st), or st itself
static hipStream_t tail_fork(hipStream_t st, bool want) {
    if (!want || !side_stream().enabled) return st;
    TailStream& t = tail_stream();
    if (!t.ok) return st;
    (void)hipEventRecord(t.fork_ev, st);
    (void)hipStreamWaitEvent(t.s, t.fork_ev, 0);
    return t.s;
}
static void tail_join(hipStream_t st, hipStream_t ts) {
    if (ts == st) return;
    TailStream& t = tail_stream();
    (void)hipEventRecord(t.join_ev, ts);
    (void)hipStreamWaitEvent(st, t.join_ev, 0);
}

// whether the adjoint of this graph's attention layers runs fused (the forward of the same call sequence did, then)
static bool ablk_serves(const Graph& g) {
    if (g.bucket_start[5] > g.bucket_start[4]) return false;  // an atom of more than 64 tokens
    if (!g.tiles_planned) return false;                       // a small graph built before the block was forced
    if (g_attn_fused & 4) return true;
    // A tile is one wave's serial chain (40 us forward, 100 us adjoint): below a few waves per SIMD the launch costs that
    // latency whatever its size, and the three row-parallel kernels are quicker (one box, three-kernel / fused ms per step:
    // 1 000 atoms 1.69 / 2.18, 3 000: 2.96 / 3.29, 5 000: 4.16 / 4.08, 10 000: 7.30 / 6.72; model.h ABLK_MIN_TILES). Many
    // 64-slot tiles (the adjoint's instantiation for them spills): likewise.
    return g.n_tiles1 >= ABLK_MIN_TILES && (int64_t)g.n_tiles2 * 20 <= g.n_nodes;
}
bool ablk_bwd_on(const Graph& g) { return (g_attn_fused & 2) && ablk_serves(g); }

// atoms of at most 64 tokens (attention tile counts 1 .. 4 of the graph's bucket lists); false = not served
bool ablk_fwd(const Model& m, const Graph& g, const AttnLayerW& A, const float* X, float* X1, float* OC, float scale,
              hipStream_t st) {
    if (!(g_attn_fused & 1) || !A.qkv.fwd2s || !A.out.fwd2s || !ablk_serves(g)) return false;
    const bool ln = m.layer_norm();
    const float qscale = scale * AB_LOG2E;
    const W2 wq = w2s_fwd(A.qkv), wo = w2s_fwd(A.out);
    const float* beta = ln ? A.b_attn : nullptr;
    const int n1 = g.n_tiles1, n2 = g.n_tiles2;
#define PET_ABLK_FWD(NQ, LNF, LIST, CNT)                                                                            \
    {                                                                                                               \
        constexpr int NW = 4;                                                                                       \
        const size_t lds = (size_t)NW * NQ * 16384 + 2 * AB_SLOT;                                                   \
        allow_big_lds(k_ablk_fwd<NQ, LNF>, lds);                                                                    \
        k_ablk_fwd<NQ, LNF><<<cdiv(CNT, NW), 64 * NW, lds, st>>>(X, A.g_attn, beta, wq, A.qkv.b, wo, A.out.b, g.fc,  \
                                                                 LIST, CNT, g.n_edges, qscale, X1, OC);             \
    }
    const hipStream_t st1 = st;
    const hipStream_t ts = tail_fork(st1, n1 > 0 && n2 > 0);
    if (n2 > 0) {
        st = ts;
        if (ln) PET_ABLK_FWD(2, true, g.tile_desc + 2 * (size_t)n1, n2) else PET_ABLK_FWD(2, false, g.tile_desc + 2 * (size_t)n1, n2)
        st = st1;
    }
    if (n1 > 0) {  // the 32-slot tiles: two four-wave workgroups per CU
        const size_t lds = (size_t)4 * 16384 + AB4_NSLOT * AB4_SLOT;
        if (ln) {
            allow_big_lds(k_ablk_fwd4<true>, lds);
            k_ablk_fwd4<true><<<cdiv(n1, 4), 256, lds, st>>>(X, A.g_attn, beta, wq, A.qkv.b, wo, A.out.b, g.fc, g.tile_desc, n1,
                                                             g.n_edges, qscale, X1, OC);
        } else {
            allow_big_lds(k_ablk_fwd4<false>, lds);
            k_ablk_fwd4<false><<<cdiv(n1, 4), 256, lds, st>>>(X, A.g_attn, beta, wq, A.qkv.b, wo, A.out.b, g.fc, g.tile_desc, n1,
                                                              g.n_edges, qscale, X1, OC);
        }
    }
    tail_join(st1, ts);
#undef PET_ABLK_FWD
    return true;
}

// dXin [E + N, D] = adjoint of the layer input; dbias [E] = key-bias gradient of this layer summed over the heads
bool ablk_bwd(const Model& m, const Graph& g, const AttnLayerW& A, const float* X, const float* dX1, const float* dOC,
              float* dXin, float* dbias, float scale, hipStream_t st) {
    if (!ablk_bwd_on(g) || !A.qkv_g.fwd2s || !A.qkv_g.bwd2s || !A.out.bwd2s) return false;
    const bool ln = m.layer_norm();
    const float qscale = scale * AB_LOG2E;
    const W2 wq = w2s_fwd(A.qkv_g), wqt = w2s_bwd(A.qkv_g), wot = w2s_bwd(A.out);  // (norm_attention folded into W_qkv)
    const float* beta = ln ? A.b_attn : nullptr;
    const int n1 = g.n_tiles1, n2 = g.n_tiles2;
#define PET_ABLK_BWD(NQ, LNF, LIST, CNT)                                                                             \
    {                                                                                                                \
        constexpr int WPB = 4 / NQ;                                                                                  \
        const size_t lds = (size_t)WPB * NQ * 32768 + 2 * AB_SLOT_B;                                                 \
        allow_big_lds(k_ablk_bwd<NQ, LNF>, lds);                                                                     \
        k_ablk_bwd<NQ, LNF><<<cdiv(CNT, WPB), 64 * WPB, lds, st>>>(X, dX1, dOC, A.g_attn, beta, wq, A.qkv_g.b, wot,  \
                                                                   wqt, g.fc, LIST, CNT, g.n_edges, qscale, scale,   \
                                                                   dXin, dbias);                                     \
    }
    const hipStream_t st1 = st;
    const hipStream_t ts = tail_fork(st1, n1 > 0 && n2 > 0);
    if (n2 > 0) {
        st = ts;
        if (ln) PET_ABLK_BWD(2, true, g.tile_desc + 2 * (size_t)n1, n2) else PET_ABLK_BWD(2, false, g.tile_desc + 2 * (size_t)n1, n2)
        st = st1;
    }
    if (n1 > 0)  // the 32-slot tiles: pet_ablk_bwd1.hip (its own translation unit: matrix products in VGPR form, build.py)
        ablk_bwd1_launch(ln, X, dX1, dOC, A.g_attn, beta, wq, A.qkv_g.b, wot, wqt, g.fc, g.tile_desc, n1, g.n_edges, qscale, scale,
                         dXin, dbias, st);
    tail_join(st1, ts);
#undef PET_ABLK_BWD
    return true;
}

}  // namespace pet
