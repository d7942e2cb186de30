// fp32-grade row GEMMs of the sampler on the fp16 matrix pipe (included by sd_kernels.hip after the fused fp32 kernels).
//
// Every GEMM operand x is pre-scaled by a power of two s and stored as TWO fp16 numbers, hi = fp16(s x) and
// lo = fp16(s x - hi): 22 mantissa bits.  a.b is then three v_mfma_f32_32x32x16_f16 (lo.hi, hi.lo, hi.hi; lo.lo is below
// fp32's own rounding) accumulated in fp32 and un-scaled in the epilogue.  Measured (tools/exp/gemm_f16x3.hip): relative
// L2 error against fp64 1.8e-7 for K = 256 - lower than the fp32 FMA chain's 2.8e-7 - at 2.4x the rate of
// v_mfma_f32_32x32x2_f32, which is 16x slower per flop than the fp16 instruction and, sharing the VALU's fp32 lanes,
// also stalls the co-resident wave's LayerNorm/GELU work (tools/stamps.py).
//
// Weights (static over a rollout) are split once per sd_ddim_sample call into "fragment-major" planes: the 16 bytes
// lane l needs for (k-step, column tile, plane) sit at [..][lane][8], so one wave-load reads 1 KiB contiguous (8 cache
// lines instead of 64 - with 3 MFMAs per product the row-major layout is bound by the L1 tag rate).  Activations
// are split by whoever writes the LDS panel: an LDS row holds either D floats or {hi[D], lo[D]} halfs, same bytes.
// The folded cross-attention operands (G = K_h Wq_h, V' = V_h Wo_h^T) get the same treatment once per rollout; the step
// token's row, identical for every trajectory, is read from one shared block per (layer, step) instead of being
// copied into every trajectory.
//
// Used by sd_ddim_sample for hidden_dim 256 when the folded path applies; everything else stays on the fp32 kernels.
#pragma once

constexpr float F16_ACT_SCALE = 8.0f;     // LayerNorm outputs, attention outputs, GELU outputs (|x| < 8190)
constexpr float F16_P_SCALE = 1024.0f;    // softmax probabilities (<= 1)
#ifndef SD_F16_GRING
#define SD_F16_GRING 4
#endif
constexpr int F16_GRING = SD_F16_GRING;   // k-steps of G in flight

// ---------------------------------------------------------------------------------------------------
// once-per-rollout preparation
// ---------------------------------------------------------------------------------------------------
__global__ void f16_absmax_kernel(const float *__restrict__ x, long n, unsigned *out) {
    float m = 0.f;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {   // 16-byte loads (every weight matrix: 4 x fewer round trips)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n >> 2); i += (long)gridDim.x * blockDim.x) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + 4 * i);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // read first: a thousand waves' atomics on one word serialise (the launch took 17 us for a 256 x 256 matrix, 18 launches per rollout)
    const unsigned b = __builtin_bit_cast(unsigned, m);
    if ((threadIdx.x & 63) == 0 && b > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, b);
}

// rows of [G (D) | V' (D)]: separate maxima
__global__ void f16_absmax_gv_kernel(const float *__restrict__ gv, long rows, int D, unsigned *outG, unsigned *outV) {
    float mg = 0.f, mv = 0.f;
    const long n = rows * 2 * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float a = fabsf(gv[i]);
        if ((i % (2 * D)) < D) mg = fmaxf(mg, a);
        else mv = fmaxf(mv, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mg = fmaxf(mg, __shfl_xor(mg, o, 64));
        mv = fmaxf(mv, __shfl_xor(mv, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(outG, __builtin_bit_cast(unsigned, mg));
        atomicMax(outV, __builtin_bit_cast(unsigned, mv));
    }
}

// W (N x D, row-major fp32, N a multiple of D) -> fragment-major split planes
//   dst[pass = n / D][wn][ks][tn][plane][lane][8],  n = pass*D + wn*WN + tn*32 + (lane & 31),  k = ks*16 + 8*(lane >> 5) + e
template <int D>
__global__ void f16_pack_weight_kernel(const float *__restrict__ W, int N, const unsigned *maxbits, f16 *__restrict__ dst,
                                       float *scale_out) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16;
    const float scale = f16_scale_from_bits(*maxbits);
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const long total = (long)N * (D / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / (D / 8)), k8 = (int)(i % (D / 8));
        const f32x4 a = *reinterpret_cast<const f32x4 *>(W + (long)n * D + k8 * 8);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(W + (long)n * D + k8 * 8 + 4);
        f16x4 h0, l0, h1, l1;
        f16_split4(a, scale, h0, l0);
        f16_split4(b, scale, h1, l1);
        const int pass = n / D, nn = n % D, wn = nn / C::WN, tn = (nn % C::WN) / 32, l31 = nn & 31;
        const int ks = k8 >> 1, lane = (k8 & 1) * 32 + l31;
        f16 *o = dst + ((((long)(pass * C::WAVES_N + wn) * NK + ks) * C::TN + tn) * 2) * 512 + lane * 8;
        *reinterpret_cast<f16x4 *>(o) = h0;
        *reinterpret_cast<f16x4 *>(o + 4) = h1;
        *reinterpret_cast<f16x4 *>(o + 512) = l0;
        *reinterpret_cast<f16x4 *>(o + 516) = l1;
    }
}

// Training: every D x D weight block of a step in ONE launch, fixed scale F16_W_SCALE (the value panel_gemm16_kernel applies
// when it splits in registers).  src_off[b] = float offset of block b (D rows of D floats, row-major) in `src`; block b's planes
// go to dst + b * 2 * D * D halfs in the pass layout of f16_pack_weight_kernel.  grid = (chunks, n_blocks).
template <int D>
__global__ void f16_pack_blocks_kernel(const float *__restrict__ src, const long *__restrict__ src_off, f16 *__restrict__ dst, float scale) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16;
    const float *W = src + src_off[blockIdx.y];
    f16 *out = dst + (long)blockIdx.y * 2 * D * D;
    constexpr int total = D * (D / 8);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i / (D / 8), k8 = i % (D / 8);
        const f32x4 a = *reinterpret_cast<const f32x4 *>(W + (long)n * D + k8 * 8);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(W + (long)n * D + k8 * 8 + 4);
        f16x4 h0, l0, h1, l1;
        f16_split4(a, scale, h0, l0);
        f16_split4(b, scale, h1, l1);
        const int wn = n / C::WN, tn = (n % C::WN) / 32, l31 = n & 31;
        const int ks = k8 >> 1, lane = (k8 & 1) * 32 + l31;
        f16 *o = out + ((((long)wn * NK + ks) * C::TN + tn) * 2) * 512 + lane * 8;
        *reinterpret_cast<f16x4 *>(o) = h0;
        *reinterpret_cast<f16x4 *>(o + 4) = h1;
        *reinterpret_cast<f16x4 *>(o + 512) = l0;
        *reinterpret_cast<f16x4 *>(o + 516) = l1;
    }
}

// The same planes of the TRANSPOSED block (element (n, k) = W[k][n]: the B operand of a dX GEMM) straight from the row-major
// weights: a workgroup moves one 64 x 64 tile through LDS (coalesced reads along a weight row, conflict-free column reads),
// instead of a gather of every W^T into an fp32 copy first (49 us per step for the 2.6 M weights of the C2 decoder).
// grid = ((D / 64)^2, n_blocks), block = 256.
template <int D>
__global__ __launch_bounds__(256) void f16_pack_blocks_t_kernel(const float *__restrict__ src, const long *__restrict__ src_off, f16 *__restrict__ dst,
                                                                 float scale) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16, TPR = D / 64;
    __shared__ float tile[64][65];
    const float *W = src + src_off[blockIdx.y];
    f16 *out = dst + (long)blockIdx.y * 2 * D * D;
    const int n0 = (blockIdx.x / TPR) * 64, k0 = (blockIdx.x % TPR) * 64;   // tile of W^T: rows n0.., columns k0..
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {                      // W rows k0 + r, columns n0 + 4 c4 .. +3
        const int r = i >> 4, c4 = (i & 15) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(W + (long)(k0 + r) * D + n0 + c4);
        tile[r][c4] = v[0]; tile[r][c4 + 1] = v[1]; tile[r][c4 + 2] = v[2]; tile[r][c4 + 3] = v[3];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {                       // (n, group of 8 k)
        const int nl = i & 63, k8l = i >> 6, n = n0 + nl, k8 = (k0 >> 3) + k8l;
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = tile[k8l * 8 + e][nl];
            b[e] = tile[k8l * 8 + 4 + e][nl];
        }
        f16x4 h0, l0, h1, l1;
        f16_split4(a, scale, h0, l0);
        f16_split4(b, scale, h1, l1);
        const int wn = n / C::WN, tn = (n % C::WN) / 32, l31 = n & 31;
        const int ks = k8 >> 1, lane = (k8 & 1) * 32 + l31;
        f16 *o = out + ((((long)wn * NK + ks) * C::TN + tn) * 2) * 512 + lane * 8;
        *reinterpret_cast<f16x4 *>(o) = h0;
        *reinterpret_cast<f16x4 *>(o + 4) = h1;
        *reinterpret_cast<f16x4 *>(o + 512) = l0;
        *reinterpret_cast<f16x4 *>(o + 516) = l1;
    }
}

// G rows -> per (item, head) blocks [ks][plane][half][16 slots][8]: the A operand of S^T = G LN2(h)^T.
// src row (item*4 + h)*src_slots + s holds slot slot0 + s; slots without a source row are written as zeros.
template <int D>
__global__ void f16_pack_g_kernel(const float *__restrict__ src, long items, int src_slots, int slot0, const unsigned *maxbits,
                                  f16 *__restrict__ dst, float *scale_out) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) *scale_out = scale;
    const long total = items * 4 * 16 * (D / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k8 = (int)(i % (D / 8));
        const int slot = (int)((i / (D / 8)) % 16);
        const long ih = i / (D / 8) / 16;   // item*4 + head
        f16x4 h0 = {0, 0, 0, 0}, l0 = h0, h1 = h0, l1 = h0;
        const int s = slot - slot0;
        if (s >= 0 && s < src_slots) {
            const float *row = src + (ih * src_slots + s) * 2 * D + k8 * 8;
            f16_split4(*reinterpret_cast<const f32x4 *>(row), scale, h0, l0);
            f16_split4(*reinterpret_cast<const f32x4 *>(row + 4), scale, h1, l1);
        }
        f16 *o = dst + ih * (32 * D) + (((k8 >> 1) * 2) * 2 + (k8 & 1)) * 128 + slot * 8;
        *reinterpret_cast<f16x4 *>(o) = h0;
        *reinterpret_cast<f16x4 *>(o + 4) = h1;
        *reinterpret_cast<f16x4 *>(o + 256) = l0;       // plane stride: 2 halves x 128
        *reinterpret_cast<f16x4 *>(o + 260) = l1;
    }
}

// V' rows -> per (trajectory, head) blocks [wn][tn][plane][lane][8]: the B operand of H += P V' for the k-step
// (trajectory, head); lane = 32*half + (col & 31) holds slots 8*half .. 8*half + 7 of column col.
template <int D>
__global__ void f16_pack_v_kernel(const float *__restrict__ gv, long items, const unsigned *maxbits, f16 *__restrict__ dst,
                                  float *scale_out) {
    using C = PanelCfg<D>;
    const float scale = f16_scale_from_bits(*maxbits);
    if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) *scale_out = scale;
    const long total = items * 4 * 2 * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % D), half = (int)((i / D) & 1);
        const long ih = i / D / 2;
        f16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = gv[(ih * 16 + 8 * half + e) * 2 * D + D + col] * scale;
            hh[e] = (f16)v;
            ll[e] = (f16)(v - (float)hh[e]);
        }
        const int wn = col / C::WN, tn = (col % C::WN) / 32, lane = half * 32 + (col & 31);
        f16 *o = dst + ih * (32 * D) + ((wn * C::TN + tn) * 2) * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = hh[e];
            o[512 + e] = ll[e];
        }
    }
}

// step token: one block per (layer, step) with k = head (k >= 4 zero): the extra k-step of H += P V'
template <int D>
__global__ void f16_pack_vstep_kernel(const float *__restrict__ gvstep, long items, const unsigned *maxbits,
                                      f16 *__restrict__ dst) {
    using C = PanelCfg<D>;
    const float scale = f16_scale_from_bits(*maxbits);
    const long total = items * 2 * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % D), half = (int)((i / D) & 1);
        const long item = i / D / 2;
        const int wn = col / C::WN, tn = (col % C::WN) / 32, lane = half * 32 + (col & 31);
        f16 *o = dst + item * (32 * D) + ((wn * C::TN + tn) * 2) * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = 0.f;
            if (half == 0 && e < 4) v = gvstep[(item * 4 + e) * 2 * D + D + col] * scale;
            const f16 h = (f16)v;
            o[e] = h;
            o[512 + e] = (f16)(v - (float)h);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// device pieces of the fused kernels
// ---------------------------------------------------------------------------------------------------
// H += U*c + bias   /   U = U*c + bias.  Written on whole accumulator vectors so that the backend emits packed
// v_pk_fma_f32 (two fp32 FMAs per instruction)
template <int D, bool INTO_H>
__device__ __forceinline__ void f16_unscale(f32x16 (&H)[PanelCfg<D>::TM][PanelCfg<D>::TN], f32x16 (&U)[PanelCfg<D>::TM][PanelCfg<D>::TN],
                                            float c, const float *bias, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const float bv = bias[p.col(tn)];
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) {
            if constexpr (INTO_H) H[tm][tn] = H[tm][tn] + (U[tm][tn] * c + bv);
            else U[tm][tn] = U[tm][tn] * c + bv;
        }
    }
}

// gelu(U*c + b1) -> split planes of the panel
template <int D>
__device__ __forceinline__ void f16_gelu_to_planes(float *sA, const f32x16 (&U)[PanelCfg<D>::TM][PanelCfg<D>::TN], float c,
                                                   const float *bias, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    constexpr int ROWP = 2 * C::LDA;
    f16 *sH = reinterpret_cast<f16 *>(sA);
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const float bv = bias[p.col(tn)];
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 pre = {U[tm][tn][r] * c + bv, U[tm][tn][r + 1] * c + bv};
                const f32x2 v2 = gelu_erf_fast2(pre) * F16_ACT_SCALE;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float v = v2[e];
                    const f16 h = (f16)v;
                    f16 *o = sH + p.row(tm, r + e) * ROWP + p.col(tn);
                    o[0] = h;
                    o[D] = (f16)(v - (float)h);
                }
            }
    }
}

// LayerNorm of the fp32 panel rows, written back IN PLACE as split planes {hi[D], lo[D]} (same bytes per row).
// All 16 lanes of a row have their values in registers before the first of them stores (one wave, program order).
template <int D>
__device__ __forceinline__ void f16_layer_norm_to_planes(float *sA, const float *ln_w, const float *ln_b, int lane, int wave) {
    using C = PanelCfg<D>;
    constexpr int V4 = D / 64;
#ifndef SD_LN_ROWS
#define SD_LN_ROWS 2   // rows per 16-lane group in flight: the DPP reduction chains of independent rows interleave
#endif
    constexpr int NR = SD_LN_ROWS;
    const int sub = lane & 15, grp = lane >> 4;
    for (int row0 = wave * 4 + grp; row0 < C::BM; row0 += 16 * NR) {
        f32x4 v[NR][V4];
        float s[NR], q[NR], mean[NR], rstd[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            s[n] = 0.f;
#pragma unroll
            for (int j = 0; j < V4; ++j) {
                v[n][j] = *reinterpret_cast<const f32x4 *>(sA + (row0 + 16 * n) * C::LDA + 4 * (sub + 16 * j));
                s[n] += (v[n][j][0] + v[n][j][1]) + (v[n][j][2] + v[n][j][3]);
            }
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) mean[n] = row16_sum(s[n]) * (1.0f / D);
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            q[n] = 0.f;
#pragma unroll
            for (int j = 0; j < V4; ++j) {   // whole-vector expressions: packed fp32 instructions
                v[n][j] = v[n][j] - mean[n];
                const f32x4 sq = v[n][j] * v[n][j];
                q[n] += (sq[0] + sq[1]) + (sq[2] + sq[3]);
            }
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) rstd[n] = 1.0f / sqrtf(row16_sum(q[n]) * (1.0f / D) + SD_LN_EPS);
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (sub + 16 * j);
            const f32x4 gw = *reinterpret_cast<const f32x4 *>(ln_w + c);
            const f32x4 gb = *reinterpret_cast<const f32x4 *>(ln_b + c);
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                f16 *rowp = reinterpret_cast<f16 *>(sA + (row0 + 16 * n) * C::LDA);
                const f32x4 y = (v[n][j] * rstd[n]) * gw + gb;
                f16x4 h, l;
                f16_split4(y, F16_ACT_SCALE, h, l);
                *reinterpret_cast<f16x4 *>(rowp + c) = h;
                *reinterpret_cast<f16x4 *>(rowp + D + c) = l;
            }
        }
    }
}

// attention output rows (fp32, HBM) -> split planes of the panel
template <int D>
__device__ __forceinline__ void f16_load_panel(float *sA, const float *src, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    constexpr int VEC_PER_ROW = D / 4;
    constexpr int ITERS = C::BM * VEC_PER_ROW / 256, BATCH = ITERS < 16 ? ITERS : 16;
    const float *base = src + p.r0 * D;
#pragma unroll
    for (int b0 = 0; b0 < ITERS; b0 += BATCH) {
        f32x4 v[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int i = threadIdx.x + (b0 + b) * 256;
            const int row = i / VEC_PER_ROW, c4 = i - row * VEC_PER_ROW;
            v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < p.R_left) v[b] = *reinterpret_cast<const f32x4 *>(base + (unsigned)(row * D + c4 * 4));
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int i = threadIdx.x + (b0 + b) * 256;
            const int row = i / VEC_PER_ROW, c4 = i - row * VEC_PER_ROW;
            f16x4 h, l;
            f16_split4(v[b], F16_ACT_SCALE, h, l);
            f16 *rowp = reinterpret_cast<f16 *>(sA + row * C::LDA);
            *reinterpret_cast<f16x4 *>(rowp + c4 * 4) = h;
            *reinterpret_cast<f16x4 *>(rowp + D + c4 * 4) = l;
        }
    }
}

// accumulator tile -> global rows.  Measured (tools/stamps.py): 64 one-register stores per pass (no lane shuffles) take
// 25-30 k cycles, the 16 quad-transposed 16-byte stores of chain_store_acc 6 k: the store path is bound by
// wave-instructions, not by bytes or by the ~4 VALU instructions per value of the transpose.
template <int D>
__device__ __forceinline__ void f16_store_acc(float *dst, int ld, int col0, const f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN],
                                              const ChainPos<D> &p) {
    chain_store_acc<D>(dst, ld, col0, acc, p);
}

// The residual stream h is private to the sampler's chain kernels, so it lives in HBM in ACCUMULATOR order:
// [panel][wave][j = (tn, tm, g)][lane][4 floats] - every load / store is a contiguous 1-KiB wave access of registers
// 4g .. 4g+3, no quad transposes (4 VALU instructions per value) and 16 instead of 64 load instructions per wave.
template <int D>
__device__ __forceinline__ float *f16_h_frag(float *h, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    return h + ((p.r0 / C::BM) * 4 + p.wave) * (long)(C::TM * C::TN * 16 * 64) + p.lane * 4;
}
template <int D>
__device__ __forceinline__ void f16_load_h(f32x16 (&H)[PanelCfg<D>::TM][PanelCfg<D>::TN], const float *h, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    const float *base = f16_h_frag<D>(const_cast<float *>(h), p);
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (unsigned)(((tn * C::TM + tm) * 4 + g) * 256));
                H[tm][tn][4 * g] = v[0]; H[tm][tn][4 * g + 1] = v[1]; H[tm][tn][4 * g + 2] = v[2]; H[tm][tn][4 * g + 3] = v[3];
            }
}
template <int D>
__device__ __forceinline__ void f16_store_h(float *h, const f32x16 (&H)[PanelCfg<D>::TM][PanelCfg<D>::TN], const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    float *base = f16_h_frag<D>(h, p);
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {H[tm][tn][4 * g], H[tm][tn][4 * g + 1], H[tm][tn][4 * g + 2], H[tm][tn][4 * g + 3]};
                SD_NT_STORE(v, reinterpret_cast<f32x4 *>(base + (unsigned)(((tn * C::TM + tm) * 4 + g) * 256)));
            }
}

// q | k | v tile of one pass -> the head-major buffer [sample][head][q|k|v][token][64]: wave wn's 64 columns are head wn
// (D = 4 x 64), so its 64 x 64 tile is one contiguous 16-KB run per sample (the row-major [R][3D] layout scatters it in
// 256-byte pieces at a 3-KB stride; the attention kernel then reads each head's Q, K and V as contiguous 25-KB blocks).
template <int D>
__device__ __forceinline__ void f16_store_qkv(float *qkv, int which, int T, const f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN],
                                              const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    static_assert(C::WN == 64 && C::WAVES_M == 1, "head-major stores assume 4 heads of 64 features");
    const long b0 = p.r0 / T;
    const int t0 = (int)(p.r0 - b0 * T);
    float *base = qkv + ((b0 * 4 + p.wn) * 3 + which) * (long)T * 64;     // this wave's head, first sample of the panel
    const unsigned next_sample = (unsigned)(4 * 3 * T * 64);               // same head, next sample
    const int i4 = p.lane & 3;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float x0 = acc[tm][tn][4 * g], x1 = acc[tm][tn][4 * g + 1], x2 = acc[tm][tn][4 * g + 2], x3 = acc[tm][tn][4 * g + 3];
                quad_transpose(x0, x1, x2, x3, p.lane);
                const int row = tm * 32 + 8 * g + 4 * p.half + i4;
                if (row < p.R_left) {
                    int t = t0 + row;
                    unsigned off = 0;
                    if (t >= T) { t -= T; off = next_sample; }      // T >= 64: at most one sample boundary inside a panel
                    const f32x4 v = {x0, x1, x2, x3};
                    SD_NT_STORE(v, reinterpret_cast<f32x4 *>(base + off + (unsigned)(t * 64 + tn * 32 + (p.l31 & ~3))));
                }
            }
}

// head of a step: h = x Wemb^T + b + pe on the fp32 MFMA (K = J), then qkv = LN1(h) Wqkv^T + b of layer 0 on the fp16 pipe
struct F16HeadArgs {
    DecoderHeadArgs g;
    const f16 *wf_qkv;
    const float *sc;     // sc[3] = scale of layer 0's in_proj
    int qkv_head_major;
    int h_frag;
};

struct F16LayerArgs {
    DecoderLayerArgs g;                       // fp32 pointers (h, a, qkv, biases, LN parameters, cb, tail); g.gv unused
    const f16 *wf_o, *wf_1, *wf_2, *wf_qkv;   // split fragment-major weights (wf_qkv: next layer's in_proj, 3 passes)
    const float *sc_own, *sc_next;            // scales: sc_own[0..2] = Wo, W1, W2, sc_own[4] = G, sc_own[5] = V'; sc_next[3] = in_proj
    const f16 *g16, *v16;                     // this layer: per (trajectory, head) blocks of 32*D halfs
    const f16 *gstep, *vstep;                 // this layer and step: 4 head blocks / one block
    const float *cstep;                       // 4 score biases of the step token
    int qkv_head_major;                       // layout of g.b.qkv (f16_store_qkv) - what attention_f16_head_kernel reads
    int h_frag;                               // g.a.h is in accumulator order (f16_load_h / f16_store_h)
    int next_head;                            // last layer only: go on with the NEXT step's head on the updated x (head.g.x unused)
    F16HeadArgs head;
};

struct F16Scores {
    f16x8 g[F16_GRING][2];
    f32x4 c[4];
    const f16 *gp;     // this lane's G row: block base + half*128 + slot*8
    long b0;
    int n_traj;
};

template <int D>
__device__ __forceinline__ void f16_scores_prime(F16Scores &f, const F16LayerArgs &fa, const ChainPos<D> &p) {
    const DecoderLayerArgs &g = fa.g;
    const int h = __builtin_amdgcn_readfirstlane(p.wave);
    const int Mc = g.Mk - 1;
    f.b0 = p.r0 / g.T;
    f.n_traj = (int)((p.r0 + p.R_left - 1) / g.T - f.b0) + 1;
    const int slot = p.l31 & 15, tl = (f.n_traj > 1) ? (p.l31 >> 4) : 0;
    const f16 *blk = (slot == Mc) ? fa.gstep + h * (32 * D) : fa.g16 + ((f.b0 + tl) * 4 + h) * (32 * D);
    f.gp = blk + p.half * 128 + slot * 8;
#pragma unroll
    for (int s = 0; s < F16_GRING - 1; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) f.g[s][pl] = *reinterpret_cast<const f16x8 *>(f.gp + (s * 2 + pl) * 256);
    const float *cbase = g.cb + f.b0 * 64 + h * 16 + 4 * p.half;
    const float cs = fa.cstep[h];
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // accumulator rows 4q..4q+3 are key slots 8*(q&1) + 4*half + i of trajectory q>>1
        f.c[q] = *reinterpret_cast<const f32x4 *>(cbase + ((f.n_traj > 1) ? (q >> 1) * 64 : 0) + (q & 1) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((q & 1) * 8 + 4 * p.half + i == Mc) f.c[q][i] = cs;
    }
}

// S^T of head `wave` on the fp16 pipe, softmax, P -> split planes (columns k = tl*64 + head*16 + slot, and the step
// token's probability of head h at column 128 + h; columns 132..143 zero)
template <int D>
__device__ __forceinline__ void f16_scores(float *sA, F16Scores &f, const F16LayerArgs &fa, const ChainPos<D> &p, float cg) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16, ROWP = 2 * C::LDA;
    const DecoderLayerArgs &g = fa.g;
    const int h = __builtin_amdgcn_readfirstlane(p.wave);
    const int Mc = g.Mk - 1;
    f16 *sH = reinterpret_cast<f16 *>(sA);
    const f16 *hB = sH + p.l31 * ROWP + 8 * p.half;
    f32x16 sc[2];
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[tq][r] = 0.f;
    f16x8 hf[2][2][2];
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) hf[0][tq][pl] = *reinterpret_cast<const f16x8 *>(hB + tq * 32 * ROWP + pl * D);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int cur = ks % F16_GRING, fill = (ks + F16_GRING - 1) % F16_GRING;
        if (ks + F16_GRING - 1 < NK) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) f.g[fill][pl] = *reinterpret_cast<const f16x8 *>(f.gp + ((ks + F16_GRING - 1) * 2 + pl) * 256);
        }
        if (ks + 1 < NK) {
#pragma unroll
            for (int tq = 0; tq < 2; ++tq)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    hf[(ks + 1) & 1][tq][pl] = *reinterpret_cast<const f16x8 *>(hB + tq * 32 * ROWP + pl * D + (ks + 1) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            sc[tq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.g[cur][1], hf[ks & 1][tq][0], sc[tq], 0, 0, 0);
            sc[tq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.g[cur][0], hf[ks & 1][tq][1], sc[tq], 0, 0, 0);
        }
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) sc[tq] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.g[cur][0], hf[ks & 1][tq][0], sc[tq], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    SD_STAMP(g.slot, 4);
    __syncthreads();   // every wave has read LN2(h): the panel now receives P
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
        const int qrow = tq * 32 + p.l31;
        const bool q_ok = qrow < p.R_left;
        const int key_lo = (int)((p.r0 + qrow) / g.T - f.b0) * 16;
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = (r & 3) + 8 * (r >> 2) + 4 * p.half;
            const bool ok = q_ok && kk >= key_lo && kk < key_lo + g.Mk;
            const float v = ok ? sc[tq][r] * cg + f.c[r >> 2][r & 3] : -INFINITY;
            sc[tq][r] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx == -INFINITY) mx = 0.f;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = exp2f((sc[tq][r] - mx) * g.scale_log2e);
            sc[tq][r] = pv;
            psum += pv;
        }
        psum += __shfl_xor(psum, 32, 64);
        const float inv = (psum > 0.f ? 1.0f / psum : 0.f) * F16_P_SCALE;
        f16 *rowp = sH + qrow * ROWP;
        float pstep = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = {sc[tq][4 * q] * inv, sc[tq][4 * q + 1] * inv, sc[tq][4 * q + 2] * inv, sc[tq][4 * q + 3] * inv};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((q & 1) * 8 + 4 * p.half + i == Mc && (q >> 1) * 16 == key_lo) pstep = t[i];
            f16x4 ph, pl;
            f16_split4(t, 1.0f, ph, pl);
            const int kcol = (q >> 1) * 64 + h * 16 + (q & 1) * 8 + 4 * p.half;
            *reinterpret_cast<f16x4 *>(rowp + kcol) = ph;
            *reinterpret_cast<f16x4 *>(rowp + D + kcol) = pl;
        }
        // step-token column of this head (held by the half that owns slot Mc) and this wave's share of the zero padding
        if (((Mc >> 2) & 1) == p.half) {
            const f16 sh = (f16)pstep;
            rowp[128 + h] = sh;
            rowp[D + 128 + h] = (f16)(pstep - (float)sh);
        } else {
#pragma unroll
            for (int z = 0; z < 3; ++z) {
                rowp[132 + 3 * h + z] = (f16)0.f;
                rowp[D + 132 + 3 * h + z] = (f16)0.f;
            }
        }
    }
}

// U = P V' (K = 64 per trajectory of the panel + the step-token k-step)
template <int D, int NT>
__device__ __forceinline__ void f16_pv(f32x16 (&U)[PanelCfg<D>::TM][PanelCfg<D>::TN], const f16 *aH, const F16LayerArgs &fa,
                                       const ChainPos<D> &p, long b0) {
    using C = PanelCfg<D>;
    constexpr int ROWP = 2 * C::LDA, NS = 4 * NT + 1;
    const unsigned loff = (unsigned)(p.wn * C::TN * 2 * 512 + p.lane * 8);
    const f16 *vb = fa.v16 + b0 * 4 * (32 * D);   // wave-uniform; blocks of the NT trajectories are consecutive
    auto bsrc = [&](int s, int tn, int pl) -> const f16 * {
        return (s < 4 * NT ? vb + s * (32 * D) : fa.vstep) + loff + (tn * 2 + pl) * 512;
    };
    f16x8 bq[3][C::TN][2], af[2][C::TM][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bq[s][tn][pl] = *reinterpret_cast<const f16x8 *>(bsrc(s, tn, pl));
    // P columns: k-step s < 4*NT covers k = s*16 .. (tl = s / 4, head = s % 4); the step-token step reads k = 128..143
    auto kcol = [&](int s) { return s < 4 * NT ? s * 16 : 128; };
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[0][tm][pl] = *reinterpret_cast<const f16x8 *>(aH + tm * 32 * ROWP + pl * D + kcol(0));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int cur = s % 3, fill = (s + 2) % 3;
        if (s + 2 < NS) {
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bq[fill][tn][pl] = *reinterpret_cast<const f16x8 *>(bsrc(s + 2, tn, pl));
        }
        if (s + 1 < NS) {
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    af[(s + 1) & 1][tm][pl] = *reinterpret_cast<const f16x8 *>(aH + tm * 32 * ROWP + pl * D + kcol(s + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
                    U[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1][tm][TA[t]], bq[cur][tn][TB[t]], U[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------------------
// decoder layer (folded cross-attention) with every row GEMM on the fp16 pipe
// ---------------------------------------------------------------------------------------------------
template <int D, bool TAIL>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void decoder_layer_f16_kernel(F16LayerArgs fa) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16, ROWP = 2 * C::LDA;
    constexpr long WSTREAM = (long)NK * C::TN * 2 * 512;   // halfs per (pass, wave) fragment stream
    const DecoderLayerArgs &g = fa.g;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const ChainPos<D> p(g.a.R);
    const f16 *aH = reinterpret_cast<const f16 *>(sA) + (p.wm * C::WM + p.l31) * ROWP + 8 * p.half;
    const long wOff = (long)__builtin_amdgcn_readfirstlane(p.wn) * WSTREAM;   // wave-uniform: scalar base + 32-bit lane offset
    const unsigned loff = (unsigned)p.lane * 8;
    const float c_o = 1.0f / (F16_ACT_SCALE * fa.sc_own[0]), c_1 = 1.0f / (F16_ACT_SCALE * fa.sc_own[1]);
    const float c_2 = 1.0f / (F16_ACT_SCALE * fa.sc_own[2]), c_g = 1.0f / (F16_ACT_SCALE * fa.sc_own[4]);
    const float c_v = 1.0f / (F16_P_SCALE * fa.sc_own[5]);
    f32x16 H[C::TM][C::TN], U[C::TM][C::TN];
    F16Ring<D> ring;
    SD_STAMP(g.slot, 0);
    f16_prime<D>(ring, fa.wf_o + wOff, loff);
    if (fa.h_frag) f16_load_h<D>(H, g.a.h, p);
    else chain_load_acc<D>(H, g.a.h, p);
    f16_load_panel<D>(sA, g.a.a, p);
    __syncthreads();
    SD_STAMP(g.slot, 1);
    f16_gemm<D, true>(U, aH, fa.wf_o + wOff, loff, ring);                 // h += a Wo^T + bo   (self-attention out)
    SD_STAMP(g.slot, 2);
    F16Scores fs;
    f16_scores_prime<D>(fs, fa, p);
    f16_unscale<D, true>(H, U, c_o, g.a.bo, p);
    __syncthreads();
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    f16_layer_norm_to_planes<D>(sA, g.a.ln_w, g.a.ln_b, p.lane, p.wave);
    __syncthreads();
    SD_STAMP(g.slot, 3);
    f16_scores<D>(sA, fs, fa, p, c_g);
    __syncthreads();
    SD_STAMP(g.slot, 5);
    chain_zero<D>(U);
    if (fs.n_traj > 1) f16_pv<D, 2>(U, aH, fa, p, fs.b0);     // h += P V' + boc
    else f16_pv<D, 1>(U, aH, fa, p, fs.b0);
    SD_STAMP(g.slot, 6);
    f16_prime<D>(ring, fa.wf_1 + wOff, loff);
    f16_unscale<D, true>(H, U, c_v, g.b.bo, p);
    __syncthreads();
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    f16_layer_norm_to_planes<D>(sA, g.b.ln_w, g.b.ln_b, p.lane, p.wave);
    __syncthreads();
    SD_STAMP(g.slot, 7);
    f16_gemm<D, true>(U, aH, fa.wf_1 + wOff, loff, ring);                 // u = gelu(LN3(h) W1^T + b1)
    SD_STAMP(g.slot, 8);
    f16_prime<D>(ring, fa.wf_2 + wOff, loff);
    __syncthreads();
    f16_gelu_to_planes<D>(sA, U, c_1, g.b.b1, p);
    __syncthreads();
    SD_STAMP(g.slot, 9);
    f16_gemm<D, true>(U, aH, fa.wf_2 + wOff, loff, ring);                 // h += u W2^T + b2
    SD_STAMP(g.slot, 10);
    if constexpr (TAIL) {
        f16_unscale<D, true>(H, U, c_2, g.b.b2, p);
        __syncthreads();
        chain_acc_to_lds<D>(sA, H, p);                        // fp32 rows: fc_out (+ DDIM) stays on the fp32 path
        __syncthreads();
        SD_STAMP(g.slot, 11);
        if (!fa.next_head) {
            panel_fc_out<D>(sA, g, p);
            SD_STAMP(g.slot, 12);
            return;
        }
        // the updated x rows stay in LDS (behind the K-half exchange area, 68 floats per row) and the next step's head
        // follows at once: no launch, no x round trip, no second ramp-up
        constexpr int XP = 68;
        float *xs = sA + 4096;
        f16_prime<D>(ring, fa.head.wf_qkv + wOff, loff);
        panel_fc_out<D>(sA, g, p, xs, XP);
        SD_STAMP(g.slot, 12);
        f16_head_body<D, true>(fa.head, sA, xs, XP, p, ring, g.slot, 12);
        return;
    } else {
        const float c_q = 1.0f / (F16_ACT_SCALE * fa.sc_next[3]);
        f16_prime<D>(ring, fa.wf_qkv + wOff, loff);
        f16_unscale<D, true>(H, U, c_2, g.b.b2, p);
        if (fa.h_frag) f16_store_h<D>(g.a.h, H, p);
        else f16_store_acc<D>(g.a.h, D, 0, H, p);
        SD_STAMP(g.slot, 11);
        __syncthreads();
        chain_acc_to_lds<D>(sA, H, p);
        __syncthreads();
        f16_layer_norm_to_planes<D>(sA, g.b.nln_w, g.b.nln_b, p.lane, p.wave);
        __syncthreads();
        SD_STAMP(g.slot, 12);
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {                // next layer's q | k | v
            f16_gemm<D, true>(U, aH, fa.wf_qkv + (long)pass * C::WAVES_N * WSTREAM + wOff, loff, ring);
            SD_STAMP(g.slot, 13 + 2 * pass);
            if (pass < 2) f16_prime<D>(ring, fa.wf_qkv + (long)(pass + 1) * C::WAVES_N * WSTREAM + wOff, loff);
            f16_unscale<D, false>(H, U, c_q, g.b.bqkv + pass * D, p);
            if (fa.qkv_head_major) f16_store_qkv<D>(g.b.qkv, pass, g.T, U, p);
            else f16_store_acc<D>(g.b.qkv, 3 * D, pass * D, U, p);
            SD_STAMP(g.slot, 14 + 2 * pass);
        }
    }
}

// ONE_WRAP: T >= 64, so the positional index of a row wraps at most once inside a 64-row panel.
// Everything of the head after its x rows sit in LDS (fp32, xs[row * xpitch + j], zero-padded to a multiple of 8 columns);
// ring: primed with the first fragments of wf_qkv.
template <int D, bool ONE_WRAP>
__device__ __forceinline__ void f16_head_body(const F16HeadArgs &fa, float *sA, const float *xs, int xpitch, const ChainPos<D> &p,
                                              F16Ring<D> &ring, int st_slot = SD_STAMP_HEAD_SLOT, int st_base = 0) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16, ROWP = 2 * C::LDA;
    constexpr long WSTREAM = (long)NK * C::TN * 2 * 512;
    const DecoderHeadArgs &g = fa.g;
    const float *aBase = xs + (p.wm * C::WM + p.l31) * xpitch + 4 * p.half;
    const f16 *aH = reinterpret_cast<const f16 *>(sA) + (p.wm * C::WM + p.l31) * ROWP + 8 * p.half;
    const long wOff = (long)__builtin_amdgcn_readfirstlane(p.wn) * WSTREAM;
    const unsigned loff = (unsigned)p.lane * 8;
    const float c_q = 1.0f / (F16_ACT_SCALE * fa.sc[3]);
    const int J = g.J, Jp = (J + 7) & ~7;   // <= 64
    // embedding weights of all k-steps (J <= 64: at most 8) requested up front
    f32x4 ew[8][C::TN];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int kk = ks * 8 + 4 * p.half;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (ks * 8 < Jp && kk < J) t = *reinterpret_cast<const f32x4 *>(g.emb_w + (long)(p.wn * C::WN + tn * 32 + p.l31) * J + kk);
            ew[ks][tn] = t;
        }
    __syncthreads();
    SD_STAMP(st_slot, st_base + 1);
    f32x16 H[C::TM][C::TN];
    chain_zero<D>(H);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks * 8 >= Jp) break;
        f32x4 af[C::TM];
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) af[tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * xpitch + ks * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
                    H[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][j], ew[ks][tn][j], H[tm][tn], 0, 0, 0);
    }
    SD_STAMP(st_slot, st_base + 2);
    {   // + bias + positional row.  Position = row index inside its trajectory: ONE modulo per lane, then offsets (< 64 <= T)
        const int pos0 = (int)((p.r0 + p.wm * C::WM + 4 * p.half) % g.T);
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int col = p.col(tn);
            const float bv = g.emb_b[col];
            const float *pec = g.pe + col;
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int pos = pos0 + tm * 32 + (r & 3) + 8 * (r >> 2);
                    if constexpr (ONE_WRAP) pos = pos >= g.T ? pos - g.T : pos;
                    else pos %= g.T;
                    H[tm][tn][r] += bv + pec[(unsigned)(pos * D)];
                }
        }
    }
    SD_STAMP(st_slot, st_base + 3);
    if (fa.h_frag) f16_store_h<D>(g.h, H, p);
    else f16_store_acc<D>(g.h, D, 0, H, p);
    __syncthreads();
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    f16_layer_norm_to_planes<D>(sA, g.ln_w, g.ln_b, p.lane, p.wave);
    __syncthreads();
    SD_STAMP(st_slot, st_base + 4);
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        f16_gemm<D, true>(H, aH, fa.wf_qkv + (long)pass * C::WAVES_N * WSTREAM + wOff, loff, ring);
        SD_STAMP(st_slot, st_base + 5 + 2 * pass);
        if (pass < 2) f16_prime<D>(ring, fa.wf_qkv + (long)(pass + 1) * C::WAVES_N * WSTREAM + wOff, loff);
        f16_unscale<D, false>(H, H, c_q, g.bqkv + pass * D, p);
        if constexpr (C::WN == 64) {
            if (fa.qkv_head_major) f16_store_qkv<D>(g.qkv, pass, g.T, H, p);
            else f16_store_acc<D>(g.qkv, 3 * D, pass * D, H, p);
        } else {
            f16_store_acc<D>(g.qkv, 3 * D, pass * D, H, p);
        }
        SD_STAMP(st_slot, st_base + 6 + 2 * pass);
    }
}

template <int D, bool ONE_WRAP>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void decoder_head_f16_kernel(F16HeadArgs fa) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16;
    constexpr long WSTREAM = (long)NK * C::TN * 2 * 512;
    const DecoderHeadArgs &g = fa.g;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const ChainPos<D> p(g.R);
    F16Ring<D> ring;
    f16_prime<D>(ring, fa.wf_qkv + (long)__builtin_amdgcn_readfirstlane(p.wn) * WSTREAM, (unsigned)p.lane * 8);
    const int J = g.J, Jp = (J + 7) & ~7;   // <= 64
    SD_STAMP(SD_STAMP_HEAD_SLOT, 0);
    // x rows -> panel (fp32, K = J zero-padded), 4 loads in flight per thread and round trip
    for (int i0 = threadIdx.x; i0 < C::BM * Jp; i0 += 4 * 256) {
        float xv[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = i0 + b * 256;
            const int row = i / Jp, j = i - row * Jp;
            xv[b] = (i < C::BM * Jp && row < p.R_left && j < J) ? g.x[(p.r0 + row) * J + j] : 0.f;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = i0 + b * 256;
            const int row = i / Jp, j = i - row * Jp;
            if (i < C::BM * Jp) sA[row * C::LDA + j] = xv[b];
        }
    }
    f16_head_body<D, ONE_WRAP>(fa, sA, sA, C::LDA, p, ring);
}

// ---------------------------------------------------------------------------------------------------
// Self-attention of the sampler on the fp16 pipe (head dim 64, T <= 128).  Same structure as attention_pipe_kernel:
// one workgroup per sample streams (head, 64-key chunk) units, wave w owns queries 32w..32w+31, S^T = K Q^T so that a
// query is a lane column and P^T is the B operand of O^T = V^T P^T straight from the accumulator.  K and V are split
// into fp16 pairs when a chunk is staged (K rows as they are, V transposed: the A operand of O^T needs 8 keys of one
// feature per lane), Q when a head's fragments are fetched, P after the exponentials.  With the k-slot order of a
// 16-key group defined as {4*half + 0..3, 8 + 4*half + 0..3}, a lane's 8 consecutive accumulator registers ARE its
// B fragment, and the matching V^T fragment is two 8-byte LDS reads.
// ---------------------------------------------------------------------------------------------------
constexpr float F16_QKV_SCALE = 8.0f;
#ifndef SD_ATT16_WGS
#define SD_ATT16_WGS 2
#endif
constexpr int ATT16_PITCH = 136;   // halfs per LDS row: {hi[64], lo[64]} + 8 (272 B: 16 rows hit 16 distinct 4-bank groups)

__global__ __launch_bounds__(256, SD_ATT16_WGS) void attention_f16_kernel(const float *__restrict__ qkv, int ld, float *__restrict__ out, int ldo,
                                                              int T, int heads, float scale_log2e) {
    constexpr int HD = 64, KC = 64;
    __shared__ __attribute__((aligned(16))) f16 sK[KC * ATT16_PITCH];
    __shared__ __attribute__((aligned(16))) f16 sV[HD * ATT16_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x, D = heads * HD;
    const int nchunks = (T + KC - 1) / KC, nunits = heads * nchunks;
    const int qi = wave * 32 + l31;
    const bool wave_active = wave * 32 < T, q_ok = qi < T;
    const float *base = qkv + (long)b * T * ld;   // workgroup-uniform: scalar base, 32-bit offsets below
    const float c_s = scale_log2e / (F16_QKV_SCALE * F16_QKV_SCALE);          // raw S^T accumulator -> log2-domain score
    const float c_o = 1.0f / F16_QKV_SCALE;                                   // P carries its 2^10 into the row sum as well

    // staging: K pieces (key = idx / 16, 4 features) keep rows contiguous; V pieces put the 64 keys on the lanes so that
    // the transposed 2-byte LDS writes of one instruction are contiguous.  Inside a 16-key group V^T stores key k at
    // position (k&3) + 4*((k>>3)&1) + 8*((k>>2)&1): the 8 k-slots of a lane half are then one 16-byte read.
    unsigned koff[4], voff[4];   // global offsets (floats) of this thread's pieces relative to (chunk, head)
    int klds[4], vlds[4], krow[4], vrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        krow[i] = idx >> 4;
        const int kc4 = idx & 15;
        vrow[i] = idx & 63;
        const int vc4 = idx >> 6;
        koff[i] = (unsigned)(krow[i] * ld + kc4 * 4);
        voff[i] = (unsigned)(vrow[i] * ld + vc4 * 4);
        klds[i] = krow[i] * ATT16_PITCH + kc4 * 4;
        const int k16 = vrow[i] & 15;
        vlds[i] = (vc4 * 4) * ATT16_PITCH + (vrow[i] & ~15) + (k16 & 3) + 4 * ((k16 >> 3) & 1) + 8 * ((k16 >> 2) & 1);
    }
    f32x4 kreg[4], vreg[4];
    auto fetch = [&](int u) {
        const int h = u / nchunks, kc0 = (u - h * nchunks) * KC;
        const float *kb = base + (long)kc0 * ld + D + h * HD;   // uniform
        const int left = T - kc0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = a;
            if (krow[i] < left) a = *reinterpret_cast<const f32x4 *>(kb + koff[i]);
            if (vrow[i] < left) d = *reinterpret_cast<const f32x4 *>(kb + D + voff[i]);
            kreg[i] = a;
            vreg[i] = d;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f16x4 hh, ll;
            f16_split4(kreg[i], F16_QKV_SCALE, hh, ll);
            *reinterpret_cast<f16x4 *>(sK + klds[i]) = hh;
            *reinterpret_cast<f16x4 *>(sK + klds[i] + HD) = ll;
            f16_split4(vreg[i], F16_QKV_SCALE, hh, ll);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sV[vlds[i] + e * ATT16_PITCH] = hh[e];
                sV[vlds[i] + e * ATT16_PITCH + KC] = ll[e];
            }
        }
    };
    f32x4 qraw[8];
    f16x8 qf[4][2];
    const unsigned qoff = (unsigned)((q_ok ? qi : 0) * ld + 8 * half);
    auto fetch_q = [&](int h) {
        const float *qb = base + h * HD;   // uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) qraw[i] = *reinterpret_cast<const f32x4 *>(qb + qoff + (unsigned)((i >> 1) * 16 + (i & 1) * 4));
    };
    fetch(0);
    fetch_q(0);
    f32x16 o[2];
    float m_run = -INFINITY, l_part = 0.f;

    for (int u = 0; u < nunits; ++u) {
        const int h = u / nchunks, c = u - h * nchunks, kc0 = c * KC;
        if (u == 2) SD_STAMP(SD_STAMP_ATT_SLOT, 0);
        __syncthreads();   // every wave is done reading the previous unit's K/V
        if (u == 2) SD_STAMP(SD_STAMP_ATT_SLOT, 1);
        stage();
        if (u == 2) SD_STAMP(SD_STAMP_ATT_SLOT, 2);
        if (u + 1 < nunits) fetch(u + 1);
        if (c == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f16x4 h0, l0, h1, l1;
                f16_split4(qraw[2 * ks], F16_QKV_SCALE, h0, l0);
                f16_split4(qraw[2 * ks + 1], F16_QKV_SCALE, h1, l1);
                qf[ks][0] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                qf[ks][1] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            }
            m_run = -INFINITY;
            l_part = 0.f;
        }
        if (c == nchunks - 1 && h + 1 < heads) fetch_q(h + 1);
        if (u == 2) SD_STAMP(SD_STAMP_ATT_SLOT, 3);
        __syncthreads();
        if (u == 2) SD_STAMP(SD_STAMP_ATT_SLOT, 4);
        if (!wave_active) continue;
        const int n_valid = min(T - kc0, KC);             // keys of this chunk
        const int kt_valid = (n_valid + 31) / 32;
        f32x16 sc[2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt < kt_valid) {
                const f16 *kp = sK + (kt * 32 + l31) * ATT16_PITCH + 8 * half;
                f16x8 kf[2][2];
                kf[0][0] = *reinterpret_cast<const f16x8 *>(kp);
                kf[0][1] = *reinterpret_cast<const f16x8 *>(kp + HD);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks + 1 < 4) {
                        kf[(ks + 1) & 1][0] = *reinterpret_cast<const f16x8 *>(kp + (ks + 1) * 16);
                        kf[(ks + 1) & 1][1] = *reinterpret_cast<const f16x8 *>(kp + HD + (ks + 1) * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0) sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][1], qf[0][0], zero16, 0, 0, 0);
                    else sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks & 1][1], qf[ks][0], sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks & 1][0], qf[ks][1], sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks & 1][0], qf[ks][0], sc[kt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                sc[kt] = zero16;
            }
        }
        if (u == 2) SD_STAMP(SD_STAMP_ATT_SLOT, 5);
        if (n_valid < KC) {   // wave-uniform: only the last chunk of a head has keys to mask
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (key >= n_valid) sc[kt][r] = -INFINITY;
                }
        }
        float m_c = sc[0][0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {   // v_max3_f32: hipcc keeps two v_max_f32 for nested fmaxf
            if (r == 0) asm("v_max_f32 %0, %1, %2" : "=v"(m_c) : "v"(sc[0][0]), "v"(sc[1][0]));
            else asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m_c) : "v"(m_c), "v"(sc[0][r]), "v"(sc[1][r]));
        }
        m_c = fmaxf(m_c, __shfl_xor(m_c, 32, 64));
        const float m_new = fmaxf(m_run, m_c);
        const float mb = m_new * c_s - 10.0f;             // the 2^10 of F16_P_SCALE rides in the exponent
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(sc[kt][r] * c_s - mb);
                sc[kt][r] = pv;
                psum += pv;
            }
        if (c == 0) {   // wave-uniform: first chunk of a head, nothing to rescale
            l_part = psum;
        } else {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_s);
            l_part = l_part * alpha + psum;
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ft][r] *= alpha;
        }
        m_run = m_new;
        if (u == 2) SD_STAMP(SD_STAMP_ATT_SLOT, 6);
        // O^T += V^T P^T, 16 keys per step: registers 8*j2 .. 8*j2+7 of tile kt are this lane's B fragment
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const int kt = gg >> 1, j2 = gg & 1;
            if (kt * 32 + j2 * 16 >= n_valid) break;   // wave-uniform: no live key in this and the later groups
            f16x8 ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = sc[kt][8 * j2 + e];
                ph[e] = (f16)t;
                pl[e] = (f16)(t - (float)ph[e]);
            }
            const f16 *vp = sV + l31 * ATT16_PITCH + kt * 32 + j2 * 16 + 8 * half;
#pragma unroll
            for (int ft = 0; ft < 2; ++ft) {
                const f16x8 vh = *reinterpret_cast<const f16x8 *>(vp + ft * 32 * ATT16_PITCH);
                const f16x8 vl = *reinterpret_cast<const f16x8 *>(vp + ft * 32 * ATT16_PITCH + KC);
                if (c == 0 && gg == 0) o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, zero16, 0, 0, 0);
                else o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[ft], 0, 0, 0);
                o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[ft], 0, 0, 0);
                o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[ft], 0, 0, 0);
            }
        }
        if (u == 2) SD_STAMP(SD_STAMP_ATT_SLOT, 7);
        if (u == 3) SD_STAMP(SD_STAMP_ATT_SLOT, 8);
        if (c == nchunks - 1) {
            const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
            const float inv = c_o / l_tot;
            if (q_ok) {
                float *op = out + ((long)b * T + qi) * ldo + h * HD;
#pragma unroll
                for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 t = {o[ft][4 * g4] * inv, o[ft][4 * g4 + 1] * inv, o[ft][4 * g4 + 2] * inv, o[ft][4 * g4 + 3] * inv};
                        *reinterpret_cast<f32x4 *>(op + ft * 32 + 8 * g4 + 4 * half) = t;
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Same computation, one workgroup per (sample, head): all T <= 128 keys of the head are staged at once, so a
// workgroup pays one HBM round trip and two barriers in total (the streaming kernel above pays them per 64-key
// chunk: its (head, chunk) unit took 14 k cycles against 1.5 k of MFMA and 3 k of VALU work - tools/stamps.py),
// and the softmax is a single pass over 4 score tiles.
// ---------------------------------------------------------------------------------------------------
constexpr int ATT16H_KP = 136;   // K rows: {hi[64], lo[64]} + 8 halfs
constexpr int ATT16H_VP = 264;   // V^T rows: {hi[128 keys], lo[128 keys]} + 8 halfs (528 B = 132 dwords = 4 mod 64)
constexpr size_t ATT16H_LDS = (size_t)(128 * ATT16H_KP + 64 * ATT16H_VP) * sizeof(f16);

// HM: qkv is the head-major buffer written by f16_store_qkv ([sample][head][q|k|v][token][64]); else [token][3D] rows.
template <bool HM>
__global__ __launch_bounds__(256, 2) void attention_f16_head_kernel(const float *__restrict__ qkv, int ld_rm, float *__restrict__ out, int ldo,
                                                                   int T, int heads, float scale_log2e) {
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) f16 smem16[];
    f16 *sK = smem16, *sV = smem16 + 128 * ATT16H_KP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads, D = heads * HD;
    const int qi = wave * 32 + l31;
    const bool q_ok = qi < T;
    // workgroup-uniform bases of this head's Q, K, V rows and their row stride
    const int ld = HM ? HD : ld_rm;
    const float *base = HM ? qkv + (long)blockIdx.x * 3 * T * HD : qkv + (long)b * T * ld_rm + h * HD;
    const long k_at = HM ? (long)T * HD : D, v_at = 2 * k_at;
    const float c_s = scale_log2e / (F16_QKV_SCALE * F16_QKV_SCALE);
    const float c_o = 1.0f / F16_QKV_SCALE;

    SD_STAMP(SD_STAMP_ATT_SLOT, 0);
    // every load of the workgroup is issued before anything is converted
    f32x4 kreg[8], vreg[8], qraw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i;
        const int krow = idx >> 4, kc4 = idx & 15;   // K: 16 pieces per key row
        const int vrow = idx & 127, vc4 = idx >> 7;  // V: the keys on the lanes (transposed 2-byte LDS writes stay contiguous)
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = a;
        if (krow < T) a = *reinterpret_cast<const f32x4 *>(base + k_at + (unsigned)(krow * ld + kc4 * 4));
        if (vrow < T) d = *reinterpret_cast<const f32x4 *>(base + v_at + (unsigned)(vrow * ld + vc4 * 4));
        kreg[i] = a;
        vreg[i] = d;
    }
    {
        const unsigned qoff = (unsigned)((q_ok ? qi : 0) * ld + 8 * half);
#pragma unroll
        for (int i = 0; i < 8; ++i) qraw[i] = *reinterpret_cast<const f32x4 *>(base + qoff + (unsigned)((i >> 1) * 16 + (i & 1) * 4));
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i;
        const int krow = idx >> 4, kc4 = idx & 15;
        const int vrow = idx & 127, vc4 = idx >> 7;
        f16x4 hh, ll;
        f16_split4(kreg[i], F16_QKV_SCALE, hh, ll);
        *reinterpret_cast<f16x4 *>(sK + krow * ATT16H_KP + kc4 * 4) = hh;
        *reinterpret_cast<f16x4 *>(sK + krow * ATT16H_KP + HD + kc4 * 4) = ll;
        f16_split4(vreg[i], F16_QKV_SCALE, hh, ll);
        const int k16 = vrow & 15;   // key position inside its 16-key group: the 8 k-slots of a lane half contiguous
        const int vpos = (vrow & ~15) + (k16 & 3) + 4 * ((k16 >> 3) & 1) + 8 * ((k16 >> 2) & 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sV[(vc4 * 4 + e) * ATT16H_VP + vpos] = hh[e];
            sV[(vc4 * 4 + e) * ATT16H_VP + 128 + vpos] = ll[e];
        }
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 2);
    f16x8 qf[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        f16x4 h0, l0, h1, l1;
        f16_split4(qraw[2 * ks], F16_QKV_SCALE, h0, l0);
        f16_split4(qraw[2 * ks + 1], F16_QKV_SCALE, h1, l1);
        qf[ks][0] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        qf[ks][1] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 3);
    __syncthreads();
    SD_STAMP(SD_STAMP_ATT_SLOT, 4);
    if (wave * 32 >= T) return;   // no barrier follows
    const int kt_valid = (T + 31) / 32;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 sc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (kt < kt_valid) {
            const f16 *kp = sK + (kt * 32 + l31) * ATT16H_KP + 8 * half;
            f16x8 kf[4][2];   // the tile's 8 fragments in one go: one LDS latency per tile
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kf[ks][0] = *reinterpret_cast<const f16x8 *>(kp + ks * 16);
                kf[ks][1] = *reinterpret_cast<const f16x8 *>(kp + HD + ks * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks == 0) sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][1], qf[0][0], zero16, 0, 0, 0);
                else sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][1], qf[ks][0], sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][0], qf[ks][1], sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][0], qf[ks][0], sc[kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = -INFINITY;
        }
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 5);
    {   // keys past T in the last live tile
        const int kt = kt_valid - 1;   // wave-uniform
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
            if (t4 == kt && (T & 31)) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t4 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= T) sc[t4][r] = -INFINITY;
            }
    }
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float m01, m23;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m01) : "v"(m), "v"(sc[0][r]), "v"(sc[1][r]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m23) : "v"(m01), "v"(sc[2][r]), "v"(sc[3][r]));
        m = m23;
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mb = m * c_s - 10.0f;   // the 2^10 of F16_P_SCALE rides in the exponent
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(sc[kt][r] * c_s - mb);
            sc[kt][r] = pv;
            psum += pv;
        }
    SD_STAMP(SD_STAMP_ATT_SLOT, 6);
    f32x16 o[2];
    const int n_groups = (T + 15) / 16;   // live 16-key groups
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
        if (gg >= n_groups) break;   // wave-uniform
        const int kt = gg >> 1, j2 = gg & 1;
        f16x8 ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = sc[kt][8 * j2 + e];
            ph[e] = (f16)t;
            pl[e] = (f16)(t - (float)ph[e]);
        }
        const f16 *vp = sV + l31 * ATT16H_VP + gg * 16 + 8 * half;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const f16x8 vh = *reinterpret_cast<const f16x8 *>(vp + ft * 32 * ATT16H_VP);
            const f16x8 vl = *reinterpret_cast<const f16x8 *>(vp + ft * 32 * ATT16H_VP + 128);
            if (gg == 0) o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, zero16, 0, 0, 0);
            else o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[ft], 0, 0, 0);
            o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[ft], 0, 0, 0);
            o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[ft], 0, 0, 0);
        }
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 7);
    const float l_tot = psum + __shfl_xor(psum, 32, 64);
    const float inv = c_o / l_tot;
    if (q_ok) {
        float *op = out + ((long)b * T + qi) * ldo + h * HD;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 t = {o[ft][4 * g4] * inv, o[ft][4 * g4 + 1] * inv, o[ft][4 * g4 + 2] * inv, o[ft][4 * g4 + 3] * inv};
                *reinterpret_cast<f32x4 *>(op + ft * 32 + 8 * g4 + 4 * half) = t;
            }
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 8);
}

// The same kernel with V^T staged into K's LDS after the scores (one more barrier pair): 34 KB of LDS instead of 68, so
// 3-4 workgroups per CU instead of 2 keep the HBM queues fed while others compute.
#ifndef SD_ATT_LV_OCC
#define SD_ATT_LV_OCC 3
#endif
constexpr size_t ATT16LV_LDS = (size_t)(128 * ATT16H_KP > 64 * ATT16H_VP ? 128 * ATT16H_KP : 64 * ATT16H_VP) * sizeof(f16);
// DROP (training forward): dropout on the probabilities, see attention_kernel; instantiated for HM = false only, so the
// sampler's kernel (register-tight at 3 workgroups per CU) is untouched.
template <bool HM, bool DROP = false>
__global__ __launch_bounds__(256, SD_ATT_LV_OCC) void attention_f16_head_lv_kernel(const float *__restrict__ qkv, int ld_rm, float *__restrict__ out, int ldo,
                                                                   int T, int heads, float scale_log2e, float *__restrict__ lse2,
                                                                   DropoutArgs da = DropoutArgs{}) {
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) f16 smem16[];
    f16 *sK = smem16, *sV = smem16;   // V^T takes K's place once the scores are done
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads, D = heads * HD;
    const int qi = wave * 32 + l31;
    const bool q_ok = qi < T;
    // workgroup-uniform bases of this head's Q, K, V rows and their row stride
    const int ld = HM ? HD : ld_rm;
    const float *base = HM ? qkv + (long)blockIdx.x * 3 * T * HD : qkv + (long)b * T * ld_rm + h * HD;
    const long k_at = HM ? (long)T * HD : D, v_at = 2 * k_at;
    const float c_s = scale_log2e / (F16_QKV_SCALE * F16_QKV_SCALE);
    const float c_o = 1.0f / F16_QKV_SCALE;

    SD_STAMP(SD_STAMP_ATT_SLOT, 0);
    // every load of the workgroup is issued before anything is converted
    f32x4 kreg[8], vreg[8], qraw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i;
        const int krow = idx >> 4, kc4 = idx & 15;   // K: 16 pieces per key row
        const int vrow = idx & 127, vc4 = idx >> 7;  // V: the keys on the lanes (transposed 2-byte LDS writes stay contiguous)
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = a;
        if (krow < T) a = *reinterpret_cast<const f32x4 *>(base + k_at + (unsigned)(krow * ld + kc4 * 4));
        if (vrow < T) d = *reinterpret_cast<const f32x4 *>(base + v_at + (unsigned)(vrow * ld + vc4 * 4));
        kreg[i] = a;
        vreg[i] = d;
    }
    {
        const unsigned qoff = (unsigned)((q_ok ? qi : 0) * ld + 8 * half);
#pragma unroll
        for (int i = 0; i < 8; ++i) qraw[i] = *reinterpret_cast<const f32x4 *>(base + qoff + (unsigned)((i >> 1) * 16 + (i & 1) * 4));
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i;
        const int krow = idx >> 4, kc4 = idx & 15;
        f16x4 hh, ll;
        f16_split4(kreg[i], F16_QKV_SCALE, hh, ll);
        *reinterpret_cast<f16x4 *>(sK + krow * ATT16H_KP + kc4 * 4) = hh;
        *reinterpret_cast<f16x4 *>(sK + krow * ATT16H_KP + HD + kc4 * 4) = ll;
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 2);
    f16x8 qf[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        f16x4 h0, l0, h1, l1;
        f16_split4(qraw[2 * ks], F16_QKV_SCALE, h0, l0);
        f16_split4(qraw[2 * ks + 1], F16_QKV_SCALE, h1, l1);
        qf[ks][0] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        qf[ks][1] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 3);
    __syncthreads();
    SD_STAMP(SD_STAMP_ATT_SLOT, 4);
    const bool live = wave * 32 < T;   // a wave without queries still stages V and meets the barriers
    const int kt_valid = (T + 31) / 32;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 sc[4];
    if (live) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        if (kt < kt_valid) {
            const f16 *kp = sK + (kt * 32 + l31) * ATT16H_KP + 8 * half;
            f16x8 kf[4][2];   // the tile's 8 fragments in one go: one LDS latency per tile
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kf[ks][0] = *reinterpret_cast<const f16x8 *>(kp + ks * 16);
                kf[ks][1] = *reinterpret_cast<const f16x8 *>(kp + HD + ks * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks == 0) sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0][1], qf[0][0], zero16, 0, 0, 0);
                else sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][1], qf[ks][0], sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][0], qf[ks][1], sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][0], qf[ks][0], sc[kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = -INFINITY;
        }
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 5);
    {   // keys past T in the last live tile
        const int kt = kt_valid - 1;   // wave-uniform
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
            if (t4 == kt && (T & 31)) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t4 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= T) sc[t4][r] = -INFINITY;
            }
    }
    }
    __syncthreads();   // every wave has its scores: K's LDS is free
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i;
        const int vrow = idx & 127, vc4 = idx >> 7;
        f16x4 hh, ll;
        f16_split4(vreg[i], F16_QKV_SCALE, hh, ll);
        const int k16 = vrow & 15;   // key position inside its 16-key group: the 8 k-slots of a lane half contiguous
        const int vpos = (vrow & ~15) + (k16 & 3) + 4 * ((k16 >> 3) & 1) + 8 * ((k16 >> 2) & 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sV[(vc4 * 4 + e) * ATT16H_VP + vpos] = hh[e];
            sV[(vc4 * 4 + e) * ATT16H_VP + 128 + vpos] = ll[e];
        }
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 6);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float m01, m23;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m01) : "v"(m), "v"(sc[0][r]), "v"(sc[1][r]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m23) : "v"(m01), "v"(sc[2][r]), "v"(sc[3][r]));
        m = m23;
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float mb = m * c_s - 10.0f;   // the 2^10 of F16_P_SCALE rides in the exponent
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(sc[kt][r] * c_s - mb);
            sc[kt][r] = pv;
            psum += pv;
        }
    if constexpr (DROP) {
        if (live) {
            const unsigned long mrow = ((unsigned long)b * heads + h) * T + (q_ok ? qi : 0);
            const unsigned long wq = (unsigned long)((T + 3) >> 2);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
                if (kt < kt_valid) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 m4 = dropout_quad(da, mrow * wq + (unsigned long)((kt * 32 + 8 * g + 4 * half) >> 2));
#pragma unroll
                        for (int e = 0; e < 4; ++e) sc[kt][4 * g + e] *= m4[e];
                    }
                }
        }
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 7);
    __syncthreads();   // V^T staged by everyone
    if (!live) return;
    f32x16 o[2];
    const int n_groups = (T + 15) / 16;   // live 16-key groups
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
        if (gg >= n_groups) break;   // wave-uniform
        const int kt = gg >> 1, j2 = gg & 1;
        f16x8 ph, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = sc[kt][8 * j2 + e];
            ph[e] = (f16)t;
            pl[e] = (f16)(t - (float)ph[e]);
        }
        const f16 *vp = sV + l31 * ATT16H_VP + gg * 16 + 8 * half;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const f16x8 vh = *reinterpret_cast<const f16x8 *>(vp + ft * 32 * ATT16H_VP);
            const f16x8 vl = *reinterpret_cast<const f16x8 *>(vp + ft * 32 * ATT16H_VP + 128);
            if (gg == 0) o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, zero16, 0, 0, 0);
            else o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[ft], 0, 0, 0);
            o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[ft], 0, 0, 0);
            o[ft] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[ft], 0, 0, 0);
        }
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 8);
    const float l_tot = psum + __shfl_xor(psum, 32, 64);
    const float inv = c_o / l_tot;
    // log2-sum-exp of the scaled scores (the training backward recomputes the probabilities from it): p carries 2^10
    if (lse2 && q_ok && half == 0) lse2[((long)b * heads + h) * T + qi] = mb + log2f(l_tot);
    if (q_ok) {
        float *op = out + ((long)b * T + qi) * ldo + h * HD;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 t = {o[ft][4 * g4] * inv, o[ft][4 * g4 + 1] * inv, o[ft][4 * g4 + 2] * inv, o[ft][4 * g4 + 3] * inv};
                *reinterpret_cast<f32x4 *>(op + ft * 32 + 8 * g4 + 4 * half) = t;
            }
    }
    SD_STAMP(SD_STAMP_ATT_SLOT, 9);
}

// ---------------------------------------------------------------------------------------------------
// The generic row-panel linear layer (panel_gemm_kernel) on the fp16 pipe:
//     out[R,N] = act(LN?(A)[R,D] W[N,D]^T + bias) (+ res)
// for callers that hand over plain fp32 operands of unknown magnitude (training forward and dX, encoders, the memory
// K/V projection).  The LDS panel is split in place with a power-of-two scale PER ROW (row abs-max -> [8192, 16384)), so
// gradients of any size keep 22 bits; W is split in registers as its fragments arrive, with a fixed scale 2^8 (the
// fp16 subnormal range bounds lo's absolute error by 2^-25/256 = 1.2e-10 per weight; |w| >= 256 would overflow, loudly).
// Per 16-deep k-step and wave: 12 fp16 MFMAs (384 cycles) + ~80 VALU instructions for the two splits, against
// 32 fp32 MFMAs (2048 cycles).
// ---------------------------------------------------------------------------------------------------
// DROP (training): out = res + dropout(A W^T + bias) - torch's x + dropout1(sa_block(x)) / x + dropout2(ff_block(x)); the mask
// is applied to the quad-transposed values (4 consecutive columns of one row = one Philox call, sd_common.h) and the
// residual is then added from a 16-byte load.
// PRE (training): W is not the fp32 matrix but its split fragment-major planes (f16_pack_blocks_kernel, scale 2^8, one block
// of D rows per pass) - the sampler's K loop: one contiguous 1-KiB wave load per fragment and no split in registers.
template <int D, bool HAS_LN, int ACT, bool HAS_RES, bool DROP = false, bool PRE = false>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void panel_gemm16_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                            const float *__restrict__ bias, const float *__restrict__ ln_w,
                                                            const float *__restrict__ ln_b, const float *res, float *out, int R,
                                                            int N, int lda, DropoutArgs da = DropoutArgs{}) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16, ROWP = 2 * C::LDA;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    float *sInv = sA + C::BM * C::LDA;   // 64 row un-scales behind the panel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r0 = (long)blockIdx.x * C::BM;
    {   // panel load, every load of a batch in flight before the first LDS write
        constexpr int VEC_PER_ROW = D / 4, ITERS = C::BM * VEC_PER_ROW / 256, BATCH = ITERS < 16 ? ITERS : 16;
#pragma unroll
        for (int b0 = 0; b0 < ITERS; b0 += BATCH) {
            f32x4 v[BATCH];
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const int i = tid + (b0 + b) * 256, row = i / VEC_PER_ROW, c4 = i - row * VEC_PER_ROW;
                v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r0 + row < R) v[b] = *reinterpret_cast<const f32x4 *>(A + (r0 + row) * lda + c4 * 4);
            }
#pragma unroll
            for (int b = 0; b < BATCH; ++b) {
                const int i = tid + (b0 + b) * 256, row = i / VEC_PER_ROW, c4 = i - row * VEC_PER_ROW;
                *reinterpret_cast<f32x4 *>(sA + row * C::LDA + c4 * 4) = v[b];
            }
        }
    }
    __syncthreads();
    f16_rows_to_planes<D, HAS_LN>(sA, sInv, ln_w, ln_b, lane, wave);
    __syncthreads();

    const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
    const int l31 = lane & 31, half = lane >> 5;
    const f16 *aH = reinterpret_cast<const f16 *>(sA) + (wm * C::WM + l31) * ROWP + 8 * half;
    const bool full_panel = r0 + C::BM <= R;  // workgroup-uniform
    // un-scale of this lane's accumulator rows: rows tm*32 + 8g + 4*half + 0..3 are registers 4g .. 4g+3
    f32x4 inv[C::TM][4];
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            inv[tm][g] = *reinterpret_cast<const f32x4 *>(sInv + wm * C::WM + tm * 32 + 8 * g + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) inv[tm][g][e] *= 1.0f / F16_W_SCALE;
        }

    for (int n0 = 0; n0 < N; n0 += D) {
        f32x16 acc[C::TM][C::TN];
        if constexpr (PRE) {
            constexpr long WSTREAM = (long)NK * C::TN * 2 * 512;   // halfs per (pass, wave) fragment stream
            const f16 *wf = reinterpret_cast<const f16 *>(W) + ((long)(n0 / D) * C::WAVES_N + __builtin_amdgcn_readfirstlane(wn)) * WSTREAM;
            F16Ring<D> ring;
            f16_prime<D>(ring, wf, (unsigned)lane * 8);
            f16_gemm<D, true>(acc, aH, wf, (unsigned)lane * 8, ring);
        } else {
            const float *wBase = W + (long)(n0 + wn * C::WN + l31) * D + 8 * half;
            f32x4 braw[2][C::TN][2];
            f16x8 af[2][C::TM][2];
    #pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) {
                braw[0][tn][0] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D);
                braw[0][tn][1] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D + 4);
            }
    #pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
    #pragma unroll
                for (int pl = 0; pl < 2; ++pl) af[0][tm][pl] = *reinterpret_cast<const f16x8 *>(aH + tm * 32 * ROWP + pl * D);
    #pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < NK) {
    #pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn) {
                        braw[nxt][tn][0] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D + (ks + 1) * 16);
                        braw[nxt][tn][1] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D + (ks + 1) * 16 + 4);
                    }
    #pragma unroll
                    for (int tm = 0; tm < C::TM; ++tm)
    #pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
                            af[nxt][tm][pl] = *reinterpret_cast<const f16x8 *>(aH + tm * 32 * ROWP + pl * D + (ks + 1) * 16);
                }
                f16x8 bh[C::TN], bl[C::TN];
    #pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) {
                    f16x4 h0, l0, h1, l1;
                    f16_split4(braw[cur][tn][0], F16_W_SCALE, h0, l0);
                    f16_split4(braw[cur][tn][1], F16_W_SCALE, h1, l1);
                    bh[tn] = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
                    bl[tn] = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
                }
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
    #pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn) {
                        if (ks == 0) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][tm][1], bh[tn], z, 0, 0, 0);
                        } else {
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][tm][1], bh[tn], acc[tm][tn], 0, 0, 0);
                        }
                    }
    #pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
    #pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][tm][0], bl[tn], acc[tm][tn], 0, 0, 0);
    #pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
    #pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][tm][0], bh[tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // epilogue: un-scale (per row), bias, residual, activation; quad transpose -> 16-byte stores
        const int i4 = lane & 3;
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int col = n0 + wn * C::WN + tn * 32 + l31;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) {
                const long orow = r0 + wm * C::WM + tm * 32 + 4 * half + i4;
                float *op = out + orow * N + n0 + wn * C::WN + tn * 32 + (l31 & ~3);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[tm][tn][4 * g + e] * inv[tm][g][e] + bv;
                        if constexpr (HAS_RES && ACT != 0) {   // the residual enters before the activation: element by element
                            const long row = r0 + wm * C::WM + tm * 32 + 8 * g + 4 * half + e;
                            v += (full_panel || row < R) ? res[row * N + col] : 0.f;
                        }
                        if constexpr (ACT == 1) v = gelu_erf(v);
                        x[e] = v;
                    }
                    quad_transpose(x[0], x[1], x[2], x[3], lane);
                    if (full_panel || orow + 8 * g < R) {
                        f32x4 v4 = {x[0], x[1], x[2], x[3]};
                        const long row = orow + 8 * g;
                        const int c0 = n0 + wn * C::WN + tn * 32 + (l31 & ~3);
                        if constexpr (DROP) v4 = v4 * dropout_quad(da, ((unsigned long)row * (unsigned long)((N + 3) >> 2)) + (unsigned long)(c0 >> 2));
                        // residual from ONE 16-byte load of the transposed position (64 dword loads per lane and pass before:
                        // the residual variant took 41 us against 25 us for the plain one)
                        if constexpr (HAS_RES && ACT == 0) v4 = v4 + *reinterpret_cast<const f32x4 *>(res + row * N + c0);
                        *reinterpret_cast<f32x4 *>(op + (long)(8 * g) * N) = v4;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Row chains of the unfused decoder layer (the reference's shipped shapes: horizon 10, hundreds of memory rows, where
// the cross-attention cannot be folded) on the fp16 pipe, hidden_dim 128 / 256:
//   KIND 0 (chain A): h += a Wo^T + bo ;  q = LN2(h) Wq^T + bq
//   KIND 1 (chain B): h += a Woc^T + boc ;  h += W2 gelu(W1 LN3(h) + b1) + b2 ;  qkv' = LN1'(h) Wqkv'^T + b'  (if a next layer)
// A single workgroup's GEMM unit is 192 fp16 MFMAs (6 k cycles) instead of 512 fp32 ones (33 k): at the robot's B = 1 the
// rollout is one serial chain of such units.
// ---------------------------------------------------------------------------------------------------
struct F16ChainArgs {
    ChainAArgs a;                       // KIND 0
    ChainBArgs b;                       // KIND 1
    const f16 *w0, *w1, *w2, *wqkv;     // fragment-major split weights: (Wo, Wq) or (Woc, W1, W2, next in_proj)
    const float *sc;                    // scales of w0, w1, w2
    const float *sc_next;               // scale of wqkv
};

template <int D, int KIND>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void chain_f16_kernel(F16ChainArgs fa) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16, ROWP = 2 * C::LDA;
    constexpr long WSTREAM = (long)NK * C::TN * 2 * 512;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const long R = KIND == 0 ? fa.a.R : fa.b.R;
    const ChainPos<D> p(R);
    const f16 *aH = reinterpret_cast<const f16 *>(sA) + (p.wm * C::WM + p.l31) * ROWP + 8 * p.half;
    const long wOff = (long)__builtin_amdgcn_readfirstlane(p.wn) * WSTREAM;
    const unsigned loff = (unsigned)p.lane * 8;
    const float c0 = 1.0f / (F16_ACT_SCALE * fa.sc[0]), c1 = 1.0f / (F16_ACT_SCALE * fa.sc[1]);
    f32x16 H[C::TM][C::TN], U[C::TM][C::TN];
    F16Ring<D> ring;
    f16_prime<D>(ring, fa.w0 + wOff, loff);
    float *h = KIND == 0 ? fa.a.h : fa.b.h;
    chain_load_acc<D>(H, h, p);
    f16_load_panel<D>(sA, KIND == 0 ? fa.a.a : fa.b.a, p);
    __syncthreads();
    f16_gemm<D, true>(U, aH, fa.w0 + wOff, loff, ring);
    f16_prime<D>(ring, fa.w1 + wOff, loff);
    f16_unscale<D, true>(H, U, c0, KIND == 0 ? fa.a.bo : fa.b.bo, p);
    if constexpr (KIND == 0) f16_store_acc<D>(h, D, 0, H, p);
    __syncthreads();
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    f16_layer_norm_to_planes<D>(sA, KIND == 0 ? fa.a.ln_w : fa.b.ln_w, KIND == 0 ? fa.a.ln_b : fa.b.ln_b, p.lane, p.wave);
    __syncthreads();
    f16_gemm<D, true>(U, aH, fa.w1 + wOff, loff, ring);
    if constexpr (KIND == 0) {
        f16_unscale<D, false>(H, U, c1, fa.a.bq, p);
        f16_store_acc<D>(fa.a.q, D, 0, U, p);
        return;
    } else {
        const float c2 = 1.0f / (F16_ACT_SCALE * fa.sc[2]);
        f16_prime<D>(ring, fa.w2 + wOff, loff);
        __syncthreads();
        f16_gelu_to_planes<D>(sA, U, c1, fa.b.b1, p);
        __syncthreads();
        f16_gemm<D, true>(U, aH, fa.w2 + wOff, loff, ring);
        const bool has_next = fa.b.nln_w != nullptr;   // workgroup-uniform
        if (has_next) f16_prime<D>(ring, fa.wqkv + wOff, loff);
        f16_unscale<D, true>(H, U, c2, fa.b.b2, p);
        f16_store_acc<D>(h, D, 0, H, p);
        if (!has_next) return;
        const float cq = 1.0f / (F16_ACT_SCALE * fa.sc_next[0]);
        __syncthreads();
        chain_acc_to_lds<D>(sA, H, p);
        __syncthreads();
        f16_layer_norm_to_planes<D>(sA, fa.b.nln_w, fa.b.nln_b, p.lane, p.wave);
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            f16_gemm<D, true>(U, aH, fa.wqkv + (long)pass * C::WAVES_N * WSTREAM + wOff, loff, ring);
            if (pass < 2) f16_prime<D>(ring, fa.wqkv + (long)(pass + 1) * C::WAVES_N * WSTREAM + wOff, loff);
            f16_unscale<D, false>(H, U, cq, fa.b.bqkv + pass * D, p);
            f16_store_acc<D>(fa.b.qkv, 3 * D, pass * D, U, p);
        }
    }
}
