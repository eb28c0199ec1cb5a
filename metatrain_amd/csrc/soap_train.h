// SOAP-BPNN training step (reference loop body soap_bpnn/trainer.py:344-391:
// zero_grad -> evaluate_model(is_training=True) -> MSE(E/atom) + MSE(dE/dR) -> loss.backward() -> Adam).
// Included by soap.hip inside namespace pet (same translation unit: the kernels below reuse its per-pair building blocks).
//
// What the reference obtains from autograd's double backward is computed here as FORWARD-OVER-REVERSE, like the PET
// training pass (so.hip): with u = dL_F/d(dE/dR) and v'_p = u_j - u_i the tangent of every edge vector,
//     <u, dE/dR> = d/d eps sum_i e_i(R + eps u) = sum_i e'_i,
// so  dL/d theta = d/d theta  J,   J = sum_i (gA_i e_i + e'_i),   gA_i = dL_E/d e_i,
// and J is a function of the per-atom pairs (x_i, x'_i) = (power spectrum, its tangent along u), which do not depend on
// the tail's parameters. One tangent sweep through the descriptor (k_soap_expand_jvp, k_soap_ps_jvp), then per atom the
// tail's primal + tangent forward and the joint reverse (k_soap_tail_train), then deterministic reductions over the
// atoms of each network for the weight gradients (k_soap_wgrad1 / k_soap_wgrad2: fixed summation order, no float
// atomics), then Adam (torch.optim.Adam semantics, soap_bpnn/trainer.py:279-291: lr 1e-3, no clipping).
//
// Trainable here: layernorm.<s>.{weight,bias}, bpnn.<s>.<2k>.weight, last_layers.energy.<s>.weight -- every
// parameter of the default (legacy = True) model -- and, for legacy = False models, the centre encoding and the
// Alchemical species embedding, which sit in FRONT of the tail: J is carried back through LayerNorm, the encoding, the
// power spectrum and the expansion (k_soap_ybar ... k_soap_embed_reduce below).
#pragma once

// ---------------------------------------------------------------------------------------------
// tangent sweep through the descriptor
// ---------------------------------------------------------------------------------------------
__global__ void k_edge_tangent(const float* __restrict__ u, const int* __restrict__ ctr, const int* __restrict__ nbr,
                               float4* __restrict__ vd, int64_t E) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E) return;
    const int i = ctr[p], j = nbr[p];
    vd[p] = make_float4(u[3 * j] - u[3 * i], u[3 * j + 1] - u[3 * i + 1], u[3 * j + 2] - u[3 * i + 2], 0.f);
}

// C'_i[lm n a] = sum_p (Y'_lm R_n + Y_lm R'_n)(p) w_a(species of the neighbour), Y' = grad Y . v', R' = d(R fc)/dr (u . v').
// One workgroup per atom, any number of species channels (the first-generation expansion kernel with tangents).
__global__ __launch_bounds__(256) void k_soap_expand_jvp(SoapDims d, const float4* __restrict__ geo,
                                                         const float4* __restrict__ vdot,
                                                         const int* __restrict__ rowptr, const int* __restrict__ sp_nbr,
                                                         const float* __restrict__ table, const float* __restrict__ shn,
                                                         const int* __restrict__ lut, const float* __restrict__ spw,
                                                         float* __restrict__ Cd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ys = smem;                     // [PC][NLM]
    float* Yd = Ys + PC * d.NLM;          // [PC][NLM]
    float* Gs = Yd + PC * d.NLM;          // [PC][3][NLM]
    float* Rs = Gs + PC * 3 * d.NLM;      // [PC][F]
    float* Rd = Rs + PC * d.F;            // [PC][F]
    float* us = Rd + PC * d.F;            // [PC][12] unit vector, r, 1/r, fc, dfc, v'(3), u . v'
    int* sps = reinterpret_cast<int*>(us + PC * 12);
    const int i = blockIdx.x, tid = threadIdx.x;
    const int p0 = rowptr[i], p1 = rowptr[i + 1];
    float acc[MAXK];
    int code[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
        acc[k] = 0.f;
        const int idx = tid + 256 * k;
        code[k] = idx < d.NCOEF ? lut[idx] : -1;
    }
    for (int base = p0; base < p1; base += PC) {
        const int npc = min(PC, p1 - base);
        __syncthreads();
        if (tid < npc) {
            const float4 g = geo[base + tid], t = vdot[base + tid];
            const float r = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
            const float ir = r > 0.f ? 1.0f / r : 0.f;
            float* u = us + tid * 12;
            u[0] = g.x * ir; u[1] = g.y * ir; u[2] = g.z * ir; u[3] = r; u[4] = ir;
            float dfc;
            u[5] = shifted_cosine(r, d.rc, d.width, &dfc);
            u[6] = dfc;
            u[7] = t.x; u[8] = t.y; u[9] = t.z;
            u[10] = u[0] * t.x + u[1] * t.y + u[2] * t.z;
            sps[tid] = sp_nbr[base + tid];
        }
        __syncthreads();
        for (int idx = tid; idx < npc * (d.L + 1); idx += 256) {
            const int pp = idx / (d.L + 1), mm = idx % (d.L + 1);
            const float* u = us + pp * 12;
            float* G = Gs + pp * 3 * d.NLM;
            sh_chain(u[0], u[1], u[2], mm, d.L, shn, Ys + pp * d.NLM, G, G + d.NLM, G + 2 * d.NLM, u[4]);
        }
        for (int idx = tid; idx < npc * d.F; idx += 256) {
            const int pp = idx / d.F, f = idx % d.F;
            const float* u = us + pp * 12;
            float dR;
            radial_one(d, table, f, u[3], u[5], u[6], Rs + pp * d.F + f, &dR);
            Rd[pp * d.F + f] = dR * u[10];
        }
        __syncthreads();
        for (int idx = tid; idx < npc * d.NLM; idx += 256) {
            const int pp = idx / d.NLM, lm = idx % d.NLM;
            const float* u = us + pp * 12;
            const float* G = Gs + pp * 3 * d.NLM;
            Yd[idx] = G[lm] * u[7] + G[d.NLM + lm] * u[8] + G[2 * d.NLM + lm] * u[9];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXK; k++) {
            if (code[k] < 0) continue;
            const int lm = code[k] & 255, rn = (code[k] >> 8) & 255, a = code[k] >> 16;
            float s = acc[k];
            for (int pp = 0; pp < npc; pp++)
                s += (Yd[pp * d.NLM + lm] * Rs[pp * d.F + rn] + Ys[pp * d.NLM + lm] * Rd[pp * d.F + rn]) *
                     spw[sps[pp] * d.C + a];
            acc[k] = s;
        }
    }
#pragma unroll
    for (int k = 0; k < MAXK; k++)
        if (code[k] >= 0) Cd[(size_t)i * d.NCOEF + tid + 256 * k] = acc[k];
}

// x'_i[l (a b)] = enc * sum_m (c'[m][a] c[m][b] + c[m][a] c'[m][b])   (tangent of power_spectrum.py:125-136)
__global__ __launch_bounds__(256) void k_soap_ps_jvp(SoapDims d, const float* __restrict__ Cf, const float* __restrict__ Cd,
                                                     const int* __restrict__ sp, const float* __restrict__ enc,
                                                     const int2* __restrict__ flut, float* __restrict__ xd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cs = smem;
    float* cd = smem + d.NCOEF;
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < d.NCOEF; k += 256) {
        cs[k] = Cf[(size_t)i * d.NCOEF + k];
        cd[k] = Cd[(size_t)i * d.NCOEF + k];
    }
    __syncthreads();
    const float* e = enc ? enc + (size_t)sp[i] * d.S : nullptr;
    for (int idx = threadIdx.x; idx < d.S; idx += 256) {
        const int2 code = flut[idx];
        const int pa = code.x, pb = code.y & 0xffff, M = (code.y >> 16) & 0xff, nc = code.y >> 24;
        float v = 0.f;
        for (int m = 0; m < M; m++) v += cd[pa + m * nc] * cs[pb + m * nc] + cs[pa + m * nc] * cd[pb + m * nc];
        xd[(size_t)i * d.S + idx] = e ? v * e[idx] : v;
    }
}

// ---------------------------------------------------------------------------------------------
// per atom: primal + tangent forward of the tail and the joint reverse down to the first Linear
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float d2silu(float x) {
    const float s = sigm(x);
    return s * (1.0f - s) * (2.0f + x * (1.0f - 2.0f * s));
}

__device__ __forceinline__ double block_sum_d(double v, double* red /*[4]*/) {
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// pack[i] = {mu, rstd, mu', m = mean(xhat c'),  then per hidden layer k: abar_k[H], adbar_k[H], h_k[H], h'_k[H]}
__host__ __device__ inline int soap_pack_size(int H, int NH) { return 4 + 4 * H * NH; }

__global__ __launch_bounds__(256) void k_soap_tail_train(SoapDims d, const float* __restrict__ feats,
                                                         const float* __restrict__ xd /* may be null */,
                                                         const int* __restrict__ sp, const SoapSet* __restrict__ sets,
                                                         const float* __restrict__ gA, float seed_tangent,
                                                         float* __restrict__ pack, float* __restrict__ edot) {
    __shared__ double red[4];
    __shared__ float sa[2 * MAXH], sb[2 * MAXH], hbar[2 * MAXH];
    __shared__ float part[4][2 * MAXH];
    __shared__ float sak[MAXNH][2 * MAXH];  // [a_k | a'_k] of every hidden layer (lane j writes and reads its own entries)
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, NH = d.NH, PK = soap_pack_size(H, NH);
    const SoapSet* Wp = sets + (d.legacy ? sp[i] : 0);
    const SoapSet W = *Wp;
    const float* x = feats + (size_t)i * d.S;
    const float* xt = xd ? xd + (size_t)i * d.S : nullptr;
    float mu = 0.f, rstd = 1.f, mud = 0.f, mm = 0.f;
    if (d.layernorm) {
        double s1 = 0.0, s1d = 0.0;
        for (int k = tid; k < d.S; k += 256) { s1 += (double)x[k]; if (xt) s1d += (double)xt[k]; }
        const double mean = block_sum_d(s1, red) / d.S;
        const double meand = block_sum_d(s1d, red) / d.S;
        double s2 = 0.0, s3 = 0.0;
        for (int k = tid; k < d.S; k += 256) {
            const double c = (double)x[k] - mean;
            s2 += c * c;
            if (xt) s3 += c * ((double)xt[k] - meand);
        }
        const double var = block_sum_d(s2, red) / d.S;
        const double r = 1.0 / sqrt(var + 1e-5);
        const double cc = block_sum_d(s3, red) / d.S;   // mean(c c')
        mu = (float)mean; rstd = (float)r; mud = (float)meand; mm = (float)(r * cc);  // m = mean(xhat c')
    }
    // a1 = W1 y, a1' = W1 y' with y = gamma xhat + beta, y' = gamma xhat', xhat' = rstd (c' - xhat m)
    float acc[2 * MAXH];
#pragma unroll
    for (int j = 0; j < 2 * MAXH; j++) acc[j] = 0.f;
    for (int k = tid; k < d.S; k += 256) {
        float y = x[k], yd = xt ? xt[k] : 0.f;
        if (d.layernorm) {
            const float xh = (y - mu) * rstd, g = W.ln_w[k];
            yd = g * rstd * ((yd - mud) - xh * mm);
            y = g * xh + W.ln_b[k];
        }
#pragma unroll
        for (int j = 0; j < MAXH; j++) {
            if (j >= H) break;
            const float w = W.W1[(size_t)j * d.S + k];
            acc[j] += w * y;
            acc[MAXH + j] += w * yd;
        }
    }
#pragma unroll
    for (int j = 0; j < 2 * MAXH; j++) {
        float v = acc[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) part[wave][j] = v;
    }
    __syncthreads();
    float* pk = pack + (size_t)i * PK;
    if (tid < 2 * MAXH) sa[tid] = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];  // sa = [a_k | a'_k]
    __syncthreads();
    if (wave != 0) return;
    // one wave from here: lane j < H owns row j of every hidden layer (LDS serves a wave's requests in order)
    const int j = lane;
    for (int k = 0; k < NH; k++) {
        float a = 0.f, ad = 0.f;
        if (j < H) {
            if (k == 0) { a = sa[j]; ad = sa[MAXH + j]; }
            else {
                const float* wk = Wp->Wh[k - 1] + (size_t)j * H;
                for (int q = 0; q < H; q++) { a += wk[q] * sb[q]; ad += wk[q] * sb[MAXH + q]; }
            }
            sak[k][j] = a; sak[k][MAXH + j] = ad;
        }
        __builtin_amdgcn_wave_barrier();
        if (j < H) {
            const float h = silu(a), hd = dsilu(a) * ad;
            sb[j] = h; sb[MAXH + j] = hd;
            pk[4 + 4 * H * k + 2 * H + j] = h;
            pk[4 + 4 * H * k + 3 * H + j] = hd;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // last layer e = w3 . h_NH, e' = w3 . h'_NH; seeds ebar = gA_i, e'bar = seed_tangent
    float e1 = j < H ? W.w3[j] * sb[MAXH + j] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e1 += __shfl_xor(e1, o);
    if (lane == 0) {
        edot[i] = e1;
        pk[0] = mu; pk[1] = rstd; pk[2] = mud; pk[3] = mm;
    }
    const float eb = gA[i], edb = seed_tangent;
    if (j < H) { hbar[j] = eb * W.w3[j]; hbar[MAXH + j] = edb * W.w3[j]; }
    __builtin_amdgcn_wave_barrier();
    for (int k = NH - 1; k >= 0; k--) {
        float ab = 0.f, adb = 0.f;
        if (j < H) {
            const float akk = sak[k][j], adkk = sak[k][MAXH + j];
            const float s1 = dsilu(akk);
            ab = hbar[j] * s1 + hbar[MAXH + j] * d2silu(akk) * adkk;
            adb = hbar[MAXH + j] * s1;
            pk[4 + 4 * H * k + j] = ab;
            pk[4 + 4 * H * k + H + j] = adb;
            sa[j] = ab; sa[MAXH + j] = adb;
        }
        __builtin_amdgcn_wave_barrier();
        if (k > 0 && j < H) {   // through a_k = W_k h_{k-1}
            const float* wk = Wp->Wh[k - 1];
            float hb = 0.f, hdb = 0.f;
            for (int q = 0; q < H; q++) { hb += wk[q * H + j] * sa[q]; hdb += wk[q * H + j] * sa[MAXH + q]; }
            hbar[j] = hb; hbar[MAXH + j] = hdb;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradients: sums over the atoms of one network in a fixed order
// ---------------------------------------------------------------------------------------------
// first Linear + LayerNorm: thread = feature k, block = 64 features x (network, atom chunk).
//   dW1[j][k] = sum_i (abar1[i][j] y[i][k] + adbar1[i][j] y'[i][k]),  ybar = W1^T abar1, y'bar = W1^T adbar1,
//   dgamma[k] = sum_i (ybar xhat + y'bar xhat'),  dbeta[k] = sum_i ybar.
// out[(chunk, network)][H + 2][S]
constexpr int WG_ATOMS = 32;
__global__ __launch_bounds__(64) void k_soap_wgrad1(SoapDims d, const float* __restrict__ feats, const float* __restrict__ xd,
                                                    const int* __restrict__ perm, const SpInfo* __restrict__ info,
                                                    const SoapSet* __restrict__ sets, const float* __restrict__ pack,
                                                    int n_chunks, float* __restrict__ out) {
    __shared__ float sv[WG_ATOMS][2 * MAXH + 4];
    __shared__ int sat[WG_ATOMS];
    const int s = blockIdx.y, chunk = blockIdx.z, k = blockIdx.x * 64 + threadIdx.x;
    const int H = d.H, PK = soap_pack_size(H, d.NH);
    const int a0 = info->offs[s], a1 = info->offs[s + 1];
    const int per = (a1 - a0 + n_chunks - 1) / n_chunks;
    const int lo = a0 + chunk * per, hi = min(a1, lo + per);
    const SoapSet W = sets[s];
    const bool live = k < d.S;
    float w1[MAXH], acc[MAXH], dg = 0.f, db = 0.f;
#pragma unroll
    for (int j = 0; j < MAXH; j++) {
        acc[j] = 0.f;
        w1[j] = live && j < H ? W.W1[(size_t)j * d.S + k] : 0.f;
    }
    const float g = live && d.layernorm ? W.ln_w[k] : 1.f, be = live && d.layernorm ? W.ln_b[k] : 0.f;
    for (int b0 = lo; b0 < hi; b0 += WG_ATOMS) {
        const int nb = min(WG_ATOMS, hi - b0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < nb * (2 * H + 4); idx += 64) {
            const int a = idx / (2 * H + 4), c = idx % (2 * H + 4);
            const int at = perm[b0 + a];
            if (c == 0) sat[a] = at;
            const float* pk = pack + (size_t)at * PK;
            sv[a][c] = c < 4 ? pk[c] : pk[4 + (c - 4)];   // stats, then abar1[H] | adbar1[H]
        }
        __syncthreads();
        if (!live) continue;
        for (int a = 0; a < nb; a++) {
            const int at = sat[a];
            float y = feats[(size_t)at * d.S + k], yd = xd ? xd[(size_t)at * d.S + k] : 0.f;
            float xh = 0.f, xhd = 0.f;
            if (d.layernorm) {
                const float mu = sv[a][0], r = sv[a][1], mud = sv[a][2], mm = sv[a][3];
                xh = (y - mu) * r;
                xhd = r * ((yd - mud) - xh * mm);
                y = g * xh + be;
                yd = g * xhd;
            }
            float yb = 0.f, ydb = 0.f;
#pragma unroll
            for (int j = 0; j < MAXH; j++) {
                if (j >= H) break;
                const float ab = sv[a][4 + j], adb = sv[a][4 + H + j];
                acc[j] += ab * y + adb * yd;
                yb += w1[j] * ab;
                ydb += w1[j] * adb;
            }
            dg += yb * xh + ydb * xhd;
            db += yb;
        }
    }
    if (!live) return;
    float* o = out + ((size_t)chunk * gridDim.y + s) * (H + 2) * d.S;
#pragma unroll
    for (int j = 0; j < MAXH; j++)
        if (j < H) o[(size_t)j * d.S + k] = acc[j];
    o[(size_t)H * d.S + k] = dg;
    o[(size_t)(H + 1) * d.S + k] = db;
}

// sum the chunk partials and ADD to the gradient slots of network s
__global__ void k_soap_wgrad1_reduce(SoapDims d, const float* __restrict__ part, int n_chunks, int n_sets, int s,
                                     float* __restrict__ gW1, float* __restrict__ gG, float* __restrict__ gB) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)(d.H + 2) * d.S;
    if (idx >= n) return;
    float v = 0.f;
    for (int c = 0; c < n_chunks; c++) v += part[((size_t)c * n_sets + s) * n + idx];
    const int row = (int)(idx / d.S), k = (int)(idx % d.S);
    if (row < d.H) gW1[idx] += v;
    else if (row == d.H) { if (gG) gG[k] += v; }
    else if (gB) gB[k] += v;
}

// hidden layer kk (1 <= kk < NH) or the last layer (kk == NH): one workgroup per (network, layer), elements strided over the
// threads, atoms in order
//   dW_kk[j][q] = sum_i (abar_kk[i][j] h_{kk-1}[i][q] + adbar_kk[i][j] h'_{kk-1}[i][q]),  dw3[j] = sum_i (gA_i h_NH[i][j] + st h'_NH[i][j])
__global__ __launch_bounds__(256) void k_soap_wgrad2(SoapDims d, const int* __restrict__ perm, const SpInfo* __restrict__ info,
                                                     const float* __restrict__ pack, const float* __restrict__ gA,
                                                     float seed_tangent, int s, int kk, float* __restrict__ gout) {
    extern __shared__ float sp_[];   // [WG_ATOMS][4 H + 1]: abar_kk | adbar_kk | h_{kk-1} | h'_{kk-1} | gA
    const int H = d.H, NH = d.NH, PK = soap_pack_size(H, NH), LD = 4 * H + 1;
    const bool last = kk == NH;
    const int a0 = info->offs[s], a1 = info->offs[s + 1];
    constexpr int MAXE = (MAXH * MAXH + 255) / 256;
    double acc[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; e++) acc[e] = 0.0;
    const int n_el = last ? H : H * H;
    for (int b0 = a0; b0 < a1; b0 += WG_ATOMS) {
        const int nb = min(WG_ATOMS, a1 - b0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < nb * LD; idx += 256) {
            const int a = idx / LD, c = idx % LD;
            const int at = perm[b0 + a];
            const float* pk = pack + (size_t)at * PK + 4;
            float v;
            if (c == 4 * H) v = gA[at];
            else if (c < 2 * H) v = last ? 0.f : pk[4 * H * kk + c];
            else v = pk[4 * H * (kk - 1) + c];
            sp_[idx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            const int el = threadIdx.x + 256 * e;
            if (el >= n_el) continue;
            float v = 0.f;
            if (!last) {
                const int j = el / H, q = el % H;
                for (int a = 0; a < nb; a++) {
                    const float* p = sp_ + a * LD;
                    v += p[j] * p[2 * H + q] + p[H + j] * p[3 * H + q];
                }
            } else {
                for (int a = 0; a < nb; a++) {
                    const float* p = sp_ + a * LD;
                    v += p[4 * H] * p[2 * H + el] + seed_tangent * p[3 * H + el];
                }
            }
            acc[e] += (double)v;
        }
    }
#pragma unroll
    for (int e = 0; e < MAXE; e++) {
        const int el = threadIdx.x + 256 * e;
        if (el < n_el) gout[el] += (float)acc[e];
    }
}

// ---------------------------------------------------------------------------------------------
// legacy = False: the centre encoding and the Alchemical species embedding sit in FRONT of the tail, so J has to be carried
// back through LayerNorm, the encoding, the power spectrum and the expansion:
//   (nu_y, lambda_y) = W1^T (abar_1, adbar_1)          adjoints of the first Linear's input y and of its tangent y'
//   (nu_x, lambda_x) = LayerNorm^T (second order)       gen_train.hip norm_rev_rows
//   x = p enc[species], x' = p' enc:  d enc = sum_i (nu_x p + lambda_x p'),  (nu_p, lambda_p) = (nu_x, lambda_x) enc
//   p = sum_m c c, p' = sum_m (c' c + c c'):  nu_c = P(c, nu_p) + P(c', lambda_p),  lambda_c = P(c, lambda_p),
//       P(c, f)[m][a] = sum_b (f[a][b] + f[b][a]) c[m][b]
//   c[lm n a] = sum_pairs B_{lm n} w[species of the neighbour][a]:  d w[s][a] = sum_{pairs with neighbour species s}
//       sum_{lm n} (nu_c B + lambda_c B'),  B = Y_lm R_n, B' = Y'_lm R_n + Y_lm R'_n
// All sums over atoms / pairs run in a fixed order (chunk partials, then the chunks in order).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_soap_ybar(SoapDims d, const SoapSet* __restrict__ sets, const float* __restrict__ pack,
                                                   float* __restrict__ NY, float* __restrict__ LY) {
    __shared__ float ab[2 * MAXH];
    const int i = blockIdx.x, H = d.H, PK = soap_pack_size(H, d.NH);
    const float* pk = pack + (size_t)i * PK + 4;
    if ((int)threadIdx.x < H) { ab[threadIdx.x] = pk[threadIdx.x]; ab[MAXH + threadIdx.x] = pk[H + threadIdx.x]; }
    __syncthreads();
    const float* W1 = sets[0].W1;
    for (int k = threadIdx.x; k < d.S; k += 256) {
        float a = 0.f, b = 0.f;
        for (int j = 0; j < H; j++) {
            const float w = W1[(size_t)j * d.S + k];
            a += w * ab[j];
            b += w * ab[MAXH + j];
        }
        NY[(size_t)i * d.S + k] = a;
        LY[(size_t)i * d.S + k] = b;
    }
}

// T = nu_x p + lambda_x p' (its sum over the atoms of a species is d enc); (nu_x, lambda_x) *= enc -> (nu_p, lambda_p)
__global__ __launch_bounds__(256) void k_soap_enc_rev(SoapDims d, const float* __restrict__ Cf, const float* __restrict__ Cd,
                                                      const int* __restrict__ sp, const float* __restrict__ enc,
                                                      const int2* __restrict__ flut, float* __restrict__ NX,
                                                      float* __restrict__ LX, float* __restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cs = smem;
    float* cd = smem + d.NCOEF;
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < d.NCOEF; k += 256) {
        cs[k] = Cf[(size_t)i * d.NCOEF + k];
        cd[k] = Cd[(size_t)i * d.NCOEF + k];
    }
    __syncthreads();
    const float* e = enc + (size_t)sp[i] * d.S;
    for (int idx = threadIdx.x; idx < d.S; idx += 256) {
        const int2 code = flut[idx];
        const int pa = code.x, pb = code.y & 0xffff, M = (code.y >> 16) & 0xff, nc = code.y >> 24;
        float p = 0.f, pd = 0.f;
        for (int m = 0; m < M; m++) {
            p += cs[pa + m * nc] * cs[pb + m * nc];
            pd += cd[pa + m * nc] * cs[pb + m * nc] + cs[pa + m * nc] * cd[pb + m * nc];
        }
        const size_t o = (size_t)i * d.S + idx;
        const float nx = NX[o], lx = LX[o];
        T[o] = nx * p + lx * pd;
        NX[o] = nx * e[idx];
        LX[o] = lx * e[idx];
    }
}

// partial[(chunk, s)][k] = sum over the atoms i of the chunk with species s of T[i][k]
__global__ __launch_bounds__(256) void k_soap_colsum_species(const float* __restrict__ T, const int* __restrict__ sp, int N,
                                                             int S, int n_chunks, float* __restrict__ partial) {
    const int k = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y, s = blockIdx.z;
    if (k >= S) return;
    const int per = (N + n_chunks - 1) / n_chunks, lo = chunk * per, hi = min(N, lo + per);
    float acc = 0.f;
    for (int i = lo; i < hi; i++)
        if (sp[i] == s) acc += T[(size_t)i * S + k];
    partial[((size_t)chunk * gridDim.z + s) * S + k] = acc;
}
__global__ void k_soap_reduce_add(const float* __restrict__ partial, int n_chunks, int64_t n, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float v = 0.f;
    for (int c = 0; c < n_chunks; c++) v += partial[(size_t)c * n + idx];
    out[idx] += v;
}

__global__ __launch_bounds__(256) void k_soap_ps_rev2(SoapDims d, const float* __restrict__ Cf, const float* __restrict__ Cd,
                                                      const float* __restrict__ NP, const float* __restrict__ LP,
                                                      float* __restrict__ NC, float* __restrict__ LC) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cs = smem;
    float* cd = smem + d.NCOEF;
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < d.NCOEF; k += 256) {
        cs[k] = Cf[(size_t)i * d.NCOEF + k];
        cd[k] = Cd[(size_t)i * d.NCOEF + k];
    }
    __syncthreads();
    const float* np = NP + (size_t)i * d.S;
    const float* lp = LP + (size_t)i * d.S;
    for (int idx = threadIdx.x; idx < d.NCOEF; idx += 256) {
        int l = 0;
        while (idx >= d.coef_off[l + 1]) l++;
        const int nc = d.n_per_l[l] * d.C, q = idx - d.coef_off[l];
        const int m = q / nc, a = q % nc;
        const float* crow = cs + d.coef_off[l] + m * nc;
        const float* drow = cd + d.coef_off[l] + m * nc;
        const float* fn = np + d.feat_off[l];
        const float* fl = lp + d.feat_off[l];
        float vn = 0.f, vl = 0.f;
        for (int b = 0; b < nc; b++) {
            const float sn = fn[a * nc + b] + fn[b * nc + a], sl = fl[a * nc + b] + fl[b * nc + a];
            vn += sn * crow[b] + sl * drow[b];
            vl += sl * crow[b];
        }
        NC[(size_t)i * d.NCOEF + idx] = vn;
        LC[(size_t)i * d.NCOEF + idx] = vl;
    }
}

// q[p][a] = sum_{lm n} (nu_c[i][lm n a] B_{lm n}(p) + lambda_c[i][lm n a] B'_{lm n}(p)) for every pair p of atom i
__global__ __launch_bounds__(256) void k_soap_embed_pairs(SoapDims d, const float4* __restrict__ geo,
                                                          const float4* __restrict__ vdot, const int* __restrict__ rowptr,
                                                          const float* __restrict__ table, const float* __restrict__ shn,
                                                          const int* __restrict__ lut, const float* __restrict__ NC,
                                                          const float* __restrict__ LC, float* __restrict__ q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ys = smem;                     // [PC][NLM]
    float* Yd = Ys + PC * d.NLM;          // [PC][NLM]
    float* Gs = Yd + PC * d.NLM;          // [PC][3][NLM]
    float* Rs = Gs + PC * 3 * d.NLM;      // [PC][F]
    float* Rd = Rs + PC * d.F;            // [PC][F]
    float* us = Rd + PC * d.F;            // [PC][12]
    float* ncs = us + PC * 12;            // [NCOEF] nu_c of this atom
    float* lcs = ncs + d.NCOEF;           // [NCOEF] lambda_c
    const int i = blockIdx.x, tid = threadIdx.x;
    const int p0 = rowptr[i], p1 = rowptr[i + 1];
    for (int k = tid; k < d.NCOEF; k += 256) {
        ncs[k] = NC[(size_t)i * d.NCOEF + k];
        lcs[k] = LC[(size_t)i * d.NCOEF + k];
    }
    for (int base = p0; base < p1; base += PC) {
        const int npc = min(PC, p1 - base);
        __syncthreads();
        if (tid < npc) {
            const float4 g = geo[base + tid], t = vdot[base + tid];
            const float r = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
            const float ir = r > 0.f ? 1.0f / r : 0.f;
            float* u = us + tid * 12;
            u[0] = g.x * ir; u[1] = g.y * ir; u[2] = g.z * ir; u[3] = r; u[4] = ir;
            float dfc;
            u[5] = shifted_cosine(r, d.rc, d.width, &dfc);
            u[6] = dfc;
            u[7] = t.x; u[8] = t.y; u[9] = t.z;
            u[10] = u[0] * t.x + u[1] * t.y + u[2] * t.z;
        }
        __syncthreads();
        for (int idx = tid; idx < npc * (d.L + 1); idx += 256) {
            const int pp = idx / (d.L + 1), mm = idx % (d.L + 1);
            const float* u = us + pp * 12;
            float* G = Gs + pp * 3 * d.NLM;
            sh_chain(u[0], u[1], u[2], mm, d.L, shn, Ys + pp * d.NLM, G, G + d.NLM, G + 2 * d.NLM, u[4]);
        }
        for (int idx = tid; idx < npc * d.F; idx += 256) {
            const int pp = idx / d.F, f = idx % d.F;
            const float* u = us + pp * 12;
            float dR;
            radial_one(d, table, f, u[3], u[5], u[6], Rs + pp * d.F + f, &dR);
            Rd[pp * d.F + f] = dR * u[10];
        }
        __syncthreads();
        for (int idx = tid; idx < npc * d.NLM; idx += 256) {
            const int pp = idx / d.NLM, lm = idx % d.NLM;
            const float* u = us + pp * 12;
            const float* G = Gs + pp * 3 * d.NLM;
            Yd[idx] = G[lm] * u[7] + G[d.NLM + lm] * u[8] + G[2 * d.NLM + lm] * u[9];
        }
        __syncthreads();
        for (int out = tid; out < npc * d.C; out += 256) {
            const int pp = out / d.C, a = out % d.C;
            float s = 0.f;
            for (int idx = 0; idx < d.NCOEF; idx++) {
                const int code = lut[idx];
                if ((code >> 16) != a) continue;
                const int lm = code & 255, rn = (code >> 8) & 255;
                const float y = Ys[pp * d.NLM + lm], yd = Yd[pp * d.NLM + lm], r = Rs[pp * d.F + rn], rd = Rd[pp * d.F + rn];
                s += ncs[idx] * (y * r) + lcs[idx] * (yd * r + y * rd);
            }
            q[(size_t)(base + pp) * d.C + a] = s;
        }
    }
}

// partial[chunk][s][a] = sum over the pairs p of the chunk whose neighbour has species s of q[p][a]
__global__ __launch_bounds__(256) void k_soap_embed_reduce(const float* __restrict__ q, const int* __restrict__ sp_nbr,
                                                           int64_t E, int C, int ns, int n_chunks, float* __restrict__ partial) {
    const int t = threadIdx.x, chunk = blockIdx.x;
    if (t >= ns * C) return;
    const int s = t / C, a = t % C;
    const int64_t per = (E + n_chunks - 1) / n_chunks, lo = chunk * per, hi = min(E, lo + per);
    float acc = 0.f;
    for (int64_t p = lo; p < hi; p++)
        if (sp_nbr[p] == s) acc += q[p * C + a];
    partial[(size_t)chunk * ns * C + t] = acc;
}

// torch.optim.Adam (no weight decay, no amsgrad): bias-corrected moments, step counted from 1
__global__ void k_soap_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float c1, float c2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi, vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= lr * (mi / c1) / (sqrtf(vi / c2) + eps);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct SoapTrainWs {
    float4* vd;
    float *Cd, *xd, *pack, *part;
    float *NY, *LY, *NX, *LX, *NC, *LC, *q, *part2;   // legacy = False only (see k_soap_ybar)
    int* perm;
    SpInfo* info;
    int n_chunks;
    size_t bytes;
};
constexpr int SOAP_EMBED_CHUNKS = 64;   // chunk partials of the centre-encoding / species-embedding reductions
static int soap_train_chunks(int64_t N) { return (int)std::min<int64_t>(32, std::max<int64_t>(1, (N + 1023) / 1024)); }
static void carve_soap_train(const SoapModel& m, int64_t N, int64_t E, void* base, SoapTrainWs& w) {
    const SoapDims& d = m.d;
    Carver c(base);
    const int64_t Na = N > 0 ? N : 1, Ea = E > 0 ? E : 1;
    w.n_chunks = soap_train_chunks(N);
    w.vd = c.take<float4>(Ea);
    w.Cd = c.take<float>(Na * d.NCOEF);
    w.xd = c.take<float>(Na * d.S);
    w.pack = c.take<float>(Na * soap_pack_size(d.H, d.NH));
    w.part = c.take<float>((size_t)w.n_chunks * m.n_sets * (d.H + 2) * d.S);
    w.perm = c.take<int>(Na);
    w.info = reinterpret_cast<SpInfo*>(c.take<int>((sizeof(SpInfo) + 3) / 4));
    const bool front = !d.legacy;
    const int64_t ns_s = front ? Na * d.S : 1, nc_s = front ? Na * d.NCOEF : 1;
    w.NY = c.take<float>(ns_s); w.LY = c.take<float>(ns_s);
    w.NX = c.take<float>(ns_s); w.LX = c.take<float>(ns_s);
    w.NC = c.take<float>(nc_s); w.LC = c.take<float>(nc_s);
    w.q = c.take<float>(front ? Ea * d.C : 1);
    w.part2 = c.take<float>(front ? (size_t)SOAP_EMBED_CHUNKS * d.ns * (d.S > d.C ? d.S : d.C) : 1);
    w.bytes = c.off;
}

static bool soap_trainable_key(const SoapModel& m, const std::string& key) {
    if (!m.d.legacy && (key == "species_embedding.weight" || key == "center_encoding.weight")) return true;
    return key.rfind("layernorm.", 0) == 0 || key.rfind("bpnn.", 0) == 0 || key.rfind("last_layers.", 0) == 0;
}

static int soap_zero_grad(SoapModel& m, hipStream_t st) {
    for (auto& kv : m.raw) {
        if (!soap_trainable_key(m, kv.first)) continue;
        auto it = m.grad.find(kv.first);
        if (it == m.grad.end() || it->second.second != kv.second.second) {
            float *g, *am, *av;
            int rc;
            if ((rc = salloc(m, (void**)&g, kv.second.second * 4))) return rc;
            if ((rc = salloc(m, (void**)&am, kv.second.second * 4))) return rc;
            if ((rc = salloc(m, (void**)&av, kv.second.second * 4))) return rc;
            PET_HIP_CHECK(hipMemsetAsync(am, 0, kv.second.second * 4, st));
            PET_HIP_CHECK(hipMemsetAsync(av, 0, kv.second.second * 4, st));
            m.grad[kv.first] = {g, kv.second.second};
            m.adam_m[kv.first] = am;
            m.adam_v[kv.first] = av;
        }
        PET_HIP_CHECK(hipMemsetAsync(m.grad[kv.first].first, 0, kv.second.second * 4, st));
    }
    return PET_OK;
}

// Parameter gradients of  J = sum_i gA_i e_i + <u, dE/dR>  ADDED to the slots; tangent_atomic [N] = e'_i (sum = <u, dE/dR>).
// `ws` is the workspace soap_forward ran on (spherical expansion and power spectrum are read from it).
static int soap_train_grads(SoapModel& m, const Graph& g, void* ws, int64_t ws_bytes, void* tws, int64_t tws_bytes,
                            const float* gA, const float* u, float* tangent_atomic, hipStream_t st) {
    const SoapDims& d = m.d;
    PET_REQUIRE(d.legacy || d.ns * d.C <= 256, PET_ERR_UNSUPPORTED, "more than 64 species with the Alchemical embedding");
    PET_REQUIRE(d.H <= MAXH && d.NH <= MAXNH, PET_ERR_UNSUPPORTED, "tail size outside the compiled limits");
    PET_REQUIRE(!m.grad.empty(), PET_ERR_ARGUMENT, "soap_model_zero_grad has not been called");
    SoapWs w;
    carve_soap(d, g.n_nodes, g.n_edges, ws, w);
    PET_REQUIRE((int64_t)w.bytes <= ws_bytes, PET_ERR_ARGUMENT, "soap workspace too small");
    SoapTrainWs t;
    carve_soap_train(m, g.n_nodes, g.n_edges, tws, t);
    PET_REQUIRE((int64_t)t.bytes <= tws_bytes, PET_ERR_ARGUMENT, "soap training workspace too small");
    const int N = (int)g.n_nodes;
    if (N == 0) return PET_OK;
    if (soap_ws_packed(m, ws)) {  // the inference forward stored the packed power spectrum: this pass reads the full layout,
        // rebuilt from the expansion coefficients the same forward left in the workspace (the LayerNorm statistics stay)
        allow_big_lds(k_soap_ps_m<false>, (size_t)4 * d.NCOEF * 4);
        k_soap_ps_m<false><<<cdiv(N, 4), 256, (size_t)4 * d.NCOEF * 4, st>>>(d, w.Cf, g.sp, m.enc, w.feats, w.tail, N);
        soap_note_layout(m, ws, false);
    }
    const bool tangent = u != nullptr && g.n_edges > 0;
    if (!tangent && !d.legacy) {   // the front-of-the-tail pass reads the tangents: zeros for an energy-only loss
        PET_HIP_CHECK(hipMemsetAsync(t.vd, 0, (size_t)(g.n_edges > 0 ? g.n_edges : 1) * sizeof(float4), st));
        PET_HIP_CHECK(hipMemsetAsync(t.Cd, 0, (size_t)N * d.NCOEF * 4, st));
        PET_HIP_CHECK(hipMemsetAsync(t.xd, 0, (size_t)N * d.S * 4, st));
    }
    if (tangent) {
        k_edge_tangent<<<cdiv(g.n_edges, 256), 256, 0, st>>>(u, g.ctr, g.nbr, t.vd, g.n_edges);
        const size_t lds = (size_t)PC * (5 * d.NLM + 2 * d.F + 12 + 1) * 4;
        allow_big_lds(k_soap_expand_jvp, lds);
        k_soap_expand_jvp<<<N, 256, lds, st>>>(d, g.geo, t.vd, g.rowptr, g.sp_nbr, m.table, m.shnorm, m.coef_lut,
                                               m.species_w, t.Cd);
        allow_big_lds(k_soap_ps_jvp, (size_t)2 * d.NCOEF * 4);
        k_soap_ps_jvp<<<N, 256, (size_t)2 * d.NCOEF * 4, st>>>(d, w.Cf, t.Cd, g.sp, m.enc, m.feat_lut, t.xd);
    }
    const float* xd = tangent ? t.xd : nullptr;
    const float seed_t = tangent ? 1.f : 0.f;
    k_soap_tail_train<<<N, 256, 0, st>>>(d, w.feats, xd, g.sp, m.sets, gA, seed_t, t.pack, tangent_atomic);
    // atoms bucketed by network (the forward's own bucketing exists only on its MFMA path)
    PET_HIP_CHECK(hipMemsetAsync(t.info, 0, sizeof(SpInfo), st));
    k_sp_count<<<cdiv(N, 256), 256, 0, st>>>(g.sp, d.legacy, N, t.info);
    k_sp_scan<<<1, 1, 0, st>>>(m.n_sets, t.info);
    k_sp_fill<<<cdiv(N, 1024), 1024, 0, st>>>(g.sp, d.legacy, N, t.info, t.perm);
    k_soap_wgrad1<<<dim3(cdiv(d.S, 64), m.n_sets, t.n_chunks), 64, 0, st>>>(d, w.feats, xd, t.perm, t.info, m.sets, t.pack,
                                                                          t.n_chunks, t.part);
    for (int s = 0; s < m.n_sets; s++) {
        const std::string ss = std::to_string(s);
        float* gW1 = m.grad.at("bpnn." + ss + ".0.weight").first;
        float* gG = d.layernorm ? m.grad.at("layernorm." + ss + ".weight").first : nullptr;
        float* gB = d.layernorm ? m.grad.at("layernorm." + ss + ".bias").first : nullptr;
        k_soap_wgrad1_reduce<<<cdiv((int64_t)(d.H + 2) * d.S, 256), 256, 0, st>>>(d, t.part, t.n_chunks, m.n_sets, s, gW1,
                                                                               gG, gB);
        const size_t lds2 = (size_t)WG_ATOMS * (4 * d.H + 1) * 4;
        for (int kk = 1; kk < d.NH; kk++)
            k_soap_wgrad2<<<1, 256, lds2, st>>>(d, t.perm, t.info, t.pack, gA, seed_t, s, kk,
                                                m.grad.at("bpnn." + ss + "." + std::to_string(2 * kk) + ".weight").first);
        k_soap_wgrad2<<<1, 256, lds2, st>>>(d, t.perm, t.info, t.pack, gA, seed_t, s, d.NH,
                                            m.grad.at("last_layers.energy." + ss + ".weight").first);
    }
    if (!d.legacy) {   // centre encoding and species embedding (k_soap_ybar ...)
        const float* ln_w = d.layernorm ? m.raw.at("layernorm.0.weight").first : nullptr;
        k_soap_ybar<<<N, 256, 0, st>>>(d, m.sets, t.pack, t.NY, t.LY);
        float *NX = t.NY, *LX = t.LY, *T = t.NX;
        if (d.layernorm) {
            int rc = norm_rev_rows(w.feats, t.xd, ln_w, 1, 1e-5f, t.NY, t.LY, t.NX, t.LX, N, d.S, st);
            if (rc) return rc;
            NX = t.NX; LX = t.LX; T = t.NY;
        }
        allow_big_lds(k_soap_enc_rev, (size_t)2 * d.NCOEF * 4);
        k_soap_enc_rev<<<N, 256, (size_t)2 * d.NCOEF * 4, st>>>(d, w.Cf, t.Cd, g.sp, m.enc, m.feat_lut, NX, LX, T);
        const int nch = (int)std::min<int64_t>(SOAP_EMBED_CHUNKS, N);
        k_soap_colsum_species<<<dim3(cdiv(d.S, 256), nch, d.ns), 256, 0, st>>>(T, g.sp, N, d.S, nch, t.part2);
        k_soap_reduce_add<<<cdiv((int64_t)d.ns * d.S, 256), 256, 0, st>>>(t.part2, nch, (int64_t)d.ns * d.S,
                                                                        m.grad.at("center_encoding.weight").first);
        allow_big_lds(k_soap_ps_rev2, (size_t)2 * d.NCOEF * 4);
        k_soap_ps_rev2<<<N, 256, (size_t)2 * d.NCOEF * 4, st>>>(d, w.Cf, t.Cd, NX, LX, t.NC, t.LC);
        if (g.n_edges > 0) {
            const size_t lds = ((size_t)PC * (5 * d.NLM + 2 * d.F + 12) + 2 * d.NCOEF) * 4;
            allow_big_lds(k_soap_embed_pairs, lds);
            k_soap_embed_pairs<<<N, 256, lds, st>>>(d, g.geo, t.vd, g.rowptr, m.table, m.shnorm, m.coef_lut, t.NC, t.LC, t.q);
            const int nce = (int)std::min<int64_t>(SOAP_EMBED_CHUNKS, g.n_edges);
            k_soap_embed_reduce<<<nce, 256, 0, st>>>(t.q, g.sp_nbr, g.n_edges, d.C, d.ns, nce, t.part2);
            k_soap_reduce_add<<<cdiv((int64_t)d.ns * d.C, 256), 256, 0, st>>>(t.part2, nce, (int64_t)d.ns * d.C,
                                                                            m.grad.at("species_embedding.weight").first);
        }
    }
    PET_HIP_CHECK(hipGetLastError());
    return PET_OK;
}

static int soap_adam(SoapModel& m, float lr, float b1, float b2, float eps, int64_t step, hipStream_t st) {
    PET_REQUIRE(!m.grad.empty(), PET_ERR_ARGUMENT, "soap_model_zero_grad has not been called");
    PET_REQUIRE(step >= 1, PET_ERR_ARGUMENT, "Adam steps are counted from 1");
    const float c1 = 1.f - powf(b1, (float)step), c2 = 1.f - powf(b2, (float)step);
    for (auto& kv : m.grad) {
        float* p = m.raw.at(kv.first).first;
        const int64_t n = kv.second.second;
        k_soap_adam<<<cdiv(n, 256), 256, 0, st>>>(p, kv.second.first, m.adam_m.at(kv.first), m.adam_v.at(kv.first), n, lr, b1,
                                                  b2, eps, c1, c2);
    }
    PET_HIP_CHECK(hipGetLastError());
    return soap_finalize(m, st);   // re-derive the packed / folded forms the forward kernels read
}
