// Shared device pieces of the row-panel kernels: the panel geometry (PanelCfg, ChainPos) and the split-fp16 GEMM core
// (weight ring, K loop, fp32 rows -> fp16 hi | lo planes with one power-of-two scale per row).  Included by
// sd_kernels.hip (sampler + single ops) and sd_train_chain.hip (fused training chains).
#ifndef SD_PANEL_H
#define SD_PANEL_H
#include "sd_common.h"

template <int D>
struct PanelCfg {
    static constexpr int BM = 64;
    static constexpr int WAVES_N = (D / 32 >= 4) ? 4 : D / 32;
    static constexpr int WAVES_M = 4 / WAVES_N;
    static constexpr int WM = BM / WAVES_M;  // rows per wave
    static constexpr int WN = D / WAVES_N;   // output columns per wave per pass
    static constexpr int TM = WM / 32;
    static constexpr int TN = WN / 32;
    static constexpr int LDA = D + 4;
    static constexpr size_t LDS_BYTES = (size_t)BM * LDA * sizeof(float);
};

// per-thread geometry of the accumulator tile map
template <int D>
struct ChainPos {
    using C = PanelCfg<D>;
    int lane, wave, l31, half, wm, wn;
    long r0;
    int R_left;  // valid rows in this panel (<= 64)
    __device__ ChainPos(long R) {
        lane = threadIdx.x & 63;
        wave = threadIdx.x >> 6;
        l31 = lane & 31;
        half = lane >> 5;
        wm = wave / C::WAVES_N;
        wn = wave % C::WAVES_N;
        // Workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8): relabel them so that consecutive panels -
        // which share a trajectory's folded cross-attention blocks - run on the same XCD and meet in its L2.
        const unsigned nb = gridDim.x, q8 = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7;
        const unsigned panel = xcd * q8 + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
        r0 = (long)panel * C::BM;
        const long left = R - r0;
        R_left = left < C::BM ? (int)left : C::BM;
    }
    __device__ __forceinline__ int row(int tm, int r) const { return wm * C::WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half; }
    __device__ __forceinline__ int col(int tn) const { return wn * C::WN + tn * 32 + l31; }
};

#ifndef SD_F16_WRING
#define SD_F16_WRING 2   // slots of the run-ahead weight ring (2: +0.2 % over 3, 4: -1.7 %)
#endif
template <int D>
struct F16Ring {
    f16x8 b[SD_F16_WRING][PanelCfg<D>::TN][2];   // [slot][column tile][plane]
};

// wf: this wave's fragment stream of one pass (wave-uniform pointer: scalar base); loff = lane*8 halfs
template <int D>
__device__ __forceinline__ void f16_prime(F16Ring<D> &ring, const f16 *wf, unsigned loff) {
    using C = PanelCfg<D>;
#pragma unroll
    for (int s = 0; s < SD_F16_WRING - 1; ++s)
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) ring.b[s][tn][pl] = *reinterpret_cast<const f16x8 *>(wf + loff + (unsigned)(((s * C::TN + tn) * 2 + pl) * 512));
}

// acc (+)= A(panel planes) W^T over K = D;  aH: (f16*)panel + row*(2*LDA) + 8*half of this lane's first row.
// ZERO: the accumulator starts at 0 (passed to the first MFMA as the inline constant, no register clearing)
template <int D, bool ZERO>
__device__ __forceinline__ void f16_gemm(f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN], const f16 *aH, const f16 *wf, unsigned loff,
                                         F16Ring<D> &ring) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16, ROWP = 2 * C::LDA;
    f16x8 af[2][C::TM][2];
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[0][tm][pl] = *reinterpret_cast<const f16x8 *>(aH + tm * 32 * ROWP + pl * D);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        constexpr int RS = SD_F16_WRING;
        const int cur = ks % RS, fill = (ks + RS - 1) % RS;
#ifndef SD_ABL_NO_WLOAD   // ablation builds (tools/ab_build.sh): results are wrong, timings tell what bounds the kernel
        if (ks + RS - 1 < NK) {
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    ring.b[fill][tn][pl] = *reinterpret_cast<const f16x8 *>(wf + loff + (unsigned)((((ks + RS - 1) * C::TN + tn) * 2 + pl) * 512));
        }
#endif
#ifndef SD_ABL_NO_ALOAD
        if (ks + 1 < NK) {
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    af[(ks + 1) & 1][tm][pl] = *reinterpret_cast<const f16x8 *>(aH + tm * 32 * ROWP + pl * D + (ks + 1) * 16);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};   // small terms first: lo.hi, hi.lo, hi.hi
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) {
                    if (ZERO && ks == 0 && t == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][tm][TA[0]], ring.b[cur][tn][TB[0]], z, 0, 0, 0);
                    } else {
#ifdef SD_ABL_ONE_MFMA
                        if (t == 2)
#endif
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][tm][TA[t]], ring.b[cur][tn][TB[t]], acc[tm][tn], 0, 0, 0);
                    }
                }
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr float F16_W_SCALE = 256.0f;

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false)));
    return v;
}

// amax (training): the bits of the largest |value| of the panel are atomically max-ed into one of the SD_AMAX_WORDS words at
// amax (sd_gemm_tn_grouped's scale = the maximum over the words).  One word for all of a launch's 1 600 waves serialises
// them in one L2 atomic unit: +6 us per row pass measured; spread by workgroup, a word sees ~25.
__device__ __forceinline__ void f16_emit_amax(unsigned *amax, float wmax, int lane) {
    wmax = fmaxf(wmax, __shfl_xor(wmax, 16, 64));
    wmax = fmaxf(wmax, __shfl_xor(wmax, 32, 64));
    if (lane == 0) atomicMax(amax + (blockIdx.x & (SD_AMAX_WORDS - 1)), __builtin_bit_cast(unsigned, wmax));
}

// fp32 panel rows (optionally LayerNorm-ed on the way) -> split planes in place + 1/scale per row.
// n_out (training): the first n_rows LayerNorm-ed rows are also written to n_out (row pitch D; the X operand of dW).
template <int D, bool HAS_LN>
__device__ __forceinline__ void f16_rows_to_planes(float *sA, float *sInv, const float *ln_w, const float *ln_b, int lane, int wave,
                                                   float *n_out = nullptr, int n_rows = 0, unsigned *amax = nullptr) {
    using C = PanelCfg<D>;
    constexpr int V4 = D / 64;
    // Two rows per 16-lane group in flight: with one wave per SIMD the reduction chains of a row (sum -> mean -> squares ->
    // rstd -> abs-max -> scale, ~10 dependent DPP steps each) are latency, and two independent rows interleave.  gamma / beta
    // are fetched once, not per row (tools/exp/chain_stamps.py: the LayerNorm pass 10.3 k cycles against 3.8 k without it).
    constexpr int NR = 2;
    const int sub = lane & 15, grp = lane >> 4;
    float wmax = 0.f;
    f32x4 gw[HAS_LN ? V4 : 1], gb[HAS_LN ? V4 : 1];
    if constexpr (HAS_LN) {
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            gw[j] = *reinterpret_cast<const f32x4 *>(ln_w + 4 * (sub + 16 * j));
            gb[j] = *reinterpret_cast<const f32x4 *>(ln_b + 4 * (sub + 16 * j));
        }
    }
    for (int row0 = wave * 4 + grp; row0 < C::BM; row0 += 16 * NR) {
        f32x4 v[NR][V4];
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int j = 0; j < V4; ++j) v[n][j] = *reinterpret_cast<const f32x4 *>(sA + (row0 + 16 * n) * C::LDA + 4 * (sub + 16 * j));
        if constexpr (HAS_LN) {
            float s[NR], q[NR], mean[NR], rstd[NR];
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                s[n] = 0.f;
#pragma unroll
                for (int j = 0; j < V4; ++j) s[n] += (v[n][j][0] + v[n][j][1]) + (v[n][j][2] + v[n][j][3]);
            }
#pragma unroll
            for (int n = 0; n < NR; ++n) mean[n] = row16_sum(s[n]) * (1.0f / D);
#pragma unroll
            for (int n = 0; n < NR; ++n) {
                q[n] = 0.f;
#pragma unroll
                for (int j = 0; j < V4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[n][j][e] -= mean[n];
                        q[n] += v[n][j][e] * v[n][j][e];
                    }
            }
#pragma unroll
            for (int n = 0; n < NR; ++n) rstd[n] = 1.0f / sqrtf(row16_sum(q[n]) * (1.0f / D) + SD_LN_EPS);
#pragma unroll
            for (int n = 0; n < NR; ++n)
#pragma unroll
                for (int j = 0; j < V4; ++j) {
                    const int c = 4 * (sub + 16 * j), row = row0 + 16 * n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[n][j][e] = v[n][j][e] * rstd[n] * gw[j][e] + gb[j][e];
                    if (n_out && row < n_rows) *reinterpret_cast<f32x4 *>(n_out + (unsigned)(row * D + c)) = v[n][j];
                }
        }
        float m[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            m[n] = 0.f;
#pragma unroll
            for (int j = 0; j < V4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) m[n] = fmaxf(m[n], fabsf(v[n][j][e]));
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) m[n] = row16_max(m[n]);
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            const int row = row0 + 16 * n;
            wmax = fmaxf(wmax, m[n]);
            const float scale = f16_scale_from_bits(__builtin_bit_cast(unsigned, m[n]));
            if (sub == 0) sInv[row] = 1.0f / scale;
            f16 *rowp = reinterpret_cast<f16 *>(sA + row * C::LDA);
#pragma unroll
            for (int j = 0; j < V4; ++j) {
                const int c = 4 * (sub + 16 * j);
                f16x4 h, l;
                f16_split4(v[n][j], scale, h, l);
                *reinterpret_cast<f16x4 *>(rowp + c) = h;
                *reinterpret_cast<f16x4 *>(rowp + D + c) = l;
            }
        }
    }
    if (amax) f16_emit_amax(amax, wmax, lane);
}

#endif
