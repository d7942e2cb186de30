// Fused row chains of one transformer layer for TRAINING (sd_train_fwd_chain / sd_train_bwd_chain, soccerdiffusion_hip.h).
//
// The unfused training step runs every row-local operation as its own launch: a 64-row workgroup of the panel GEMM lives
// ~46 k cycles for 6 k cycles of MFMA (panel load, split, epilogue store every time), the elementwise kernels between them
// (dropout of dy, GELU backward, LayerNorm forward recomputation and backward) each read and write 26 MB at B = 256, and the
// GPU runs ~110 launches of 15 - 45 us per step.  Here one workgroup keeps its 64 rows in LDS / registers through a whole
// chain: the next GEMM's input is produced in the panel by the previous GEMM's epilogue, and only what the backward or a
// weight-gradient GEMM needs leaves the chip.
//
// Numerics are those of panel_gemm16_kernel: every GEMM input row is split into fp16 hi | lo with its own power-of-two scale
// (gradients of any magnitude keep 22 bits), weights are the per-step split planes (scale 2^8), three fp16 MFMAs per product,
// fp32 accumulation.
//
// Geometry (PanelCfg / ChainPos, sd_panel.h): 4 waves, wave wn owns output columns [wn*WN, (wn+1)*WN); the accumulator
// register 4g + e of tile (tm, tn) is row tm*32 + 8g + 4*half + e, column tn*32 + l31.  Epilogues quad-transpose so that a
// lane holds 4 consecutive columns of one row: 16-byte stores, one Philox call per lane and quad, 16-byte LDS writes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/soccerdiffusion_hip.h"
#include "sd_common.h"
#include "sd_panel.h"

// Phase stamps of workgroup 0, wave 0 (developer builds: AB_TU=sd_train_chain tools/ab_build.sh stamps -DSD_TC_STAMPS; read with
// tools/exp/chain_stamps.py)
#ifdef SD_TC_STAMPS
__device__ unsigned long long tc_stamps[64];
#define TC_STAMP(i)                                                                                \
    do {                                                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0) tc_stamps[i] = __builtin_readcyclecounter();      \
    } while (0)
extern "C" int sd_tc_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tc_stamps), sizeof(unsigned long long) * 64);
}
#else
#define TC_STAMP(i)
#endif

namespace {

struct FwdArgs {
    long R;
    int n_next;
    const float *a; const f16 *wo; const float *bo; const float *h_in; float *h_out;
    const float *ln_w, *ln_b; float *n_out; const f16 *w1; const float *b1; float *pre; float *u;
    const f16 *w2; const float *b2; float *h2_out;
    const float *nln_w, *nln_b; float *nn_out; const f16 *wn; const float *bn; float *y_out;
    DropoutArgs d_out, d_act, d_ffn;
    unsigned *amax_a, *amax_n, *amax_u, *amax_nn, *amax_h2;
};

struct BwdArgs {
    long R;
    int passes, ldy;
    const float *dy; float *dym; const f16 *wt;
    const float *pre; float *dpre; const f16 *wt1;
    const float *x; const float *ln_w; const float *dres; float *dg; float *db;
    float *dx;
    DropoutArgs d_in, d_act;
    unsigned *amax_dy, *amax_dpre, *amax_dx;
};

// 64 rows of `src` (row pitch ld) -> fp32 LDS panel, rows past the end as zeros
template <int D>
__device__ __forceinline__ void tc_load_rows(float *sA, const float *src, int ld, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    constexpr int VEC_PER_ROW = D / 4, ITERS = C::BM * VEC_PER_ROW / 256, BATCH = ITERS < 16 ? ITERS : 16;
    const float *base = src + p.r0 * ld;
#pragma unroll
    for (int b0 = 0; b0 < ITERS; b0 += BATCH) {
        f32x4 v[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int i = threadIdx.x + (b0 + b) * 256, row = i / VEC_PER_ROW, c4 = i - row * VEC_PER_ROW;
            v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < p.R_left) v[b] = *reinterpret_cast<const f32x4 *>(base + (long)row * ld + c4 * 4);
        }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int i = threadIdx.x + (b0 + b) * 256, row = i / VEC_PER_ROW, c4 = i - row * VEC_PER_ROW;
            *reinterpret_cast<f32x4 *>(sA + row * C::LDA + c4 * 4) = v[b];
        }
    }
}

// fp32 panel rows -> f(row, first column, quad) -> split planes in place + 1/scale per row.  The row loop is a run-time
// loop (4 rows per wave and iteration): a Philox call per quad inside f stays at D / 64 independent chains per lane - applied
// in the unrolled global-load loop instead, 16 of them were interleaved and spilled.
template <int D, class F>
__device__ __forceinline__ void tc_rows_to_planes(float *sA, float *sInv, const ChainPos<D> &p, unsigned *amax, F &&f) {
    using C = PanelCfg<D>;
    constexpr int V4 = D / 64;
    const int sub = p.lane & 15, grp = p.lane >> 4;
    float wmax = 0.f;
    for (int row = p.wave * 4 + grp; row < C::BM; row += 16) {
        f32x4 v[V4];
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (sub + 16 * j);
            v[j] = *reinterpret_cast<const f32x4 *>(sA + row * C::LDA + c);
            f(row, c, v[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[j][e]));
        }
        m = row16_max(m);
        wmax = fmaxf(wmax, m);
        const float scale = f16_scale_from_bits(__builtin_bit_cast(unsigned, m));
        if (sub == 0) sInv[row] = 1.0f / scale;
        f16 *rowp = reinterpret_cast<f16 *>(sA + row * C::LDA);
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (sub + 16 * j);
            f16x4 h, l;
            f16_split4(v[j], scale, h, l);
            *reinterpret_cast<f16x4 *>(rowp + c) = h;
            *reinterpret_cast<f16x4 *>(rowp + D + c) = l;
        }
    }
    if (amax) f16_emit_amax(amax, wmax, p.lane);
}

// un-scale of this lane's accumulator rows (1 / (row scale * weight scale)), registers 4g .. 4g+3 <-> inv[tm][g]
template <int D>
__device__ __forceinline__ void tc_load_inv(f32x4 (&inv)[PanelCfg<D>::TM][4], const float *sInv, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            inv[tm][g] = *reinterpret_cast<const f32x4 *>(sInv + p.wm * C::WM + tm * 32 + 8 * g + 4 * p.half) * (1.0f / F16_W_SCALE);
}

// Visits the tile as quads: f(tm, tn, g, row in panel, first column, value) with value = acc * inv + bias, 4 consecutive
// columns of one row per lane.  f may change the value; it is not written back.
template <int D, class F>
__device__ __forceinline__ void tc_for_quads(const f32x16 (&U)[PanelCfg<D>::TM][PanelCfg<D>::TN], const f32x4 (&inv)[PanelCfg<D>::TM][4],
                                             const float *bias, const ChainPos<D> &p, F &&f) {
    using C = PanelCfg<D>;
    const int i4 = p.lane & 3;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const float bv = bias ? bias[p.col(tn)] : 0.f;
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float x0 = U[tm][tn][4 * g] * inv[tm][g][0] + bv, x1 = U[tm][tn][4 * g + 1] * inv[tm][g][1] + bv;
                float x2 = U[tm][tn][4 * g + 2] * inv[tm][g][2] + bv, x3 = U[tm][tn][4 * g + 3] * inv[tm][g][3] + bv;
                quad_transpose(x0, x1, x2, x3, p.lane);
                const int row = p.wm * C::WM + tm * 32 + 8 * g + 4 * p.half + i4;
                const int c0 = p.wn * C::WN + tn * 32 + (p.l31 & ~3);
                f(tm, tn, g, row, c0, f32x4{x0, x1, x2, x3});
            }
    }
}

__device__ __forceinline__ f32x4 gelu4(const f32x4 &v) {
    const f32x2 a = gelu_erf_fast2(f32x2{v[0], v[1]}), b = gelu_erf_fast2(f32x2{v[2], v[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}

// ======================================================================================
// forward
// ======================================================================================
template <int D, bool HAS_OUT, bool HAS_FFN, bool DROP>
__global__ __launch_bounds__(256, 2) void train_fwd_chain_kernel(FwdArgs fa) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16, ROWP = 2 * C::LDA, QPR = D / 4;
    constexpr long WSTREAM = (long)NK * C::TN * 2 * 512;   // halfs per (pass, wave) fragment stream
    extern __shared__ __attribute__((aligned(16))) float sA[];
    float *sInv = sA + C::BM * C::LDA;
    const ChainPos<D> p(fa.R);
    const f16 *aH = reinterpret_cast<const f16 *>(sA) + (p.wm * C::WM + p.l31) * ROWP + 8 * p.half;
    const long wOff = (long)__builtin_amdgcn_readfirstlane(p.wn) * WSTREAM;
    const unsigned loff = (unsigned)p.lane * 8;
    f32x16 U[C::TM][C::TN];
    f32x4 Hq[C::TM][C::TN][4];   // the residual stream of this lane's quads (HAS_OUT && HAS_FFN)
    f32x4 inv[C::TM][4];
    F16Ring<D> ring;

    TC_STAMP(0);
    if constexpr (HAS_OUT) {
        f16_prime<D>(ring, fa.wo + wOff, loff);
        tc_load_rows<D>(sA, fa.a, D, p);
        __syncthreads();
        TC_STAMP(1);
        f16_rows_to_planes<D, false>(sA, sInv, nullptr, nullptr, p.lane, p.wave, nullptr, 0, fa.amax_a);
        __syncthreads();
        TC_STAMP(2);
        f16_gemm<D, true>(U, aH, fa.wo + wOff, loff, ring);
        TC_STAMP(3);
        if (HAS_FFN) f16_prime<D>(ring, fa.w1 + wOff, loff);
        else if (fa.n_next) f16_prime<D>(ring, fa.wn + wOff, loff);
        tc_load_inv<D>(inv, sInv, p);
        __syncthreads();   // every wave is done with the planes: the panel becomes h1
        tc_for_quads<D>(U, inv, fa.bo, p, [&](int tm, int tn, int g, int row, int c0, f32x4 v) {
            if (row < p.R_left) {
                const long at = (p.r0 + row) * D + c0;
                if constexpr (DROP) v = v * dropout_quad(fa.d_out, (unsigned long)(p.r0 + row) * QPR + (unsigned long)(c0 >> 2));
                v = v + *reinterpret_cast<const f32x4 *>(fa.h_in + at);
#ifndef SD_TC_ABL_NOSTORE   // ablation builds (tools/ab_build.sh): wrong results, timings tell what bounds the kernel
                *reinterpret_cast<f32x4 *>(fa.h_out + at) = v;
#endif
            } else {
                v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (HAS_FFN) Hq[tm][tn][g] = v;
            *reinterpret_cast<f32x4 *>(sA + row * C::LDA + c0) = v;
        });
        __syncthreads();
        TC_STAMP(4);
    } else {
        if (fa.n_next) f16_prime<D>(ring, fa.wn + wOff, loff);
        tc_load_rows<D>(sA, fa.h_in, D, p);
        __syncthreads();
    }

    if constexpr (HAS_FFN) {
        f16_rows_to_planes<D, true>(sA, sInv, fa.ln_w, fa.ln_b, p.lane, p.wave, fa.n_out + p.r0 * D, p.R_left, fa.amax_n);
        __syncthreads();
        TC_STAMP(5);
        f16_gemm<D, true>(U, aH, fa.w1 + wOff, loff, ring);
        TC_STAMP(6);
        f16_prime<D>(ring, fa.w2 + wOff, loff);
        tc_load_inv<D>(inv, sInv, p);
        __syncthreads();
        tc_for_quads<D>(U, inv, fa.b1, p, [&](int, int, int, int row, int c0, f32x4 v) {
            if (row < p.R_left) {
                const long at = (p.r0 + row) * D + c0;
#ifndef SD_TC_ABL_NOSTORE
                *reinterpret_cast<f32x4 *>(fa.pre + at) = v;
#endif
                v = gelu4(v);
                if constexpr (DROP) v = v * dropout_quad(fa.d_act, (unsigned long)(p.r0 + row) * QPR + (unsigned long)(c0 >> 2));
#ifndef SD_TC_ABL_NOSTORE
                *reinterpret_cast<f32x4 *>(fa.u + at) = v;
#endif
            } else {
                v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            *reinterpret_cast<f32x4 *>(sA + row * C::LDA + c0) = v;
        });
        __syncthreads();
        TC_STAMP(7);
        f16_rows_to_planes<D, false>(sA, sInv, nullptr, nullptr, p.lane, p.wave, nullptr, 0, fa.amax_u);
        __syncthreads();
        TC_STAMP(8);
        f16_gemm<D, true>(U, aH, fa.w2 + wOff, loff, ring);
        TC_STAMP(9);
        if (fa.n_next) f16_prime<D>(ring, fa.wn + wOff, loff);
        tc_load_inv<D>(inv, sInv, p);
        __syncthreads();
        float h2max = 0.f;
        tc_for_quads<D>(U, inv, fa.b2, p, [&](int tm, int tn, int g, int row, int c0, f32x4 v) {
            if (row < p.R_left) {
                const long at = (p.r0 + row) * D + c0;
                if constexpr (DROP) v = v * dropout_quad(fa.d_ffn, (unsigned long)(p.r0 + row) * QPR + (unsigned long)(c0 >> 2));
                v = v + Hq[tm][tn][g];
                h2max = fmaxf(fmaxf(h2max, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
#ifndef SD_TC_ABL_NOSTORE
                *reinterpret_cast<f32x4 *>(fa.h2_out + at) = v;
#endif
            } else {
                v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            *reinterpret_cast<f32x4 *>(sA + row * C::LDA + c0) = v;
        });
        if (fa.amax_h2) {   // workgroup-uniform; only the last layer asks for it
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) h2max = fmaxf(h2max, __shfl_xor(h2max, o, 64));
            f16_emit_amax(fa.amax_h2, h2max, p.lane);
        }
        __syncthreads();
    }

    TC_STAMP(10);
    if (fa.n_next == 0) return;   // workgroup-uniform
    f16_rows_to_planes<D, true>(sA, sInv, fa.nln_w, fa.nln_b, p.lane, p.wave, fa.nn_out + p.r0 * D, p.R_left, fa.amax_nn);
    __syncthreads();
    TC_STAMP(11);
    tc_load_inv<D>(inv, sInv, p);
    const int ldo = fa.n_next * D;
    for (int pass = 0; pass < fa.n_next; ++pass) {
        const f16 *w = fa.wn + (long)pass * C::WAVES_N * WSTREAM + wOff;
        f16_gemm<D, true>(U, aH, w, loff, ring);
        if (pass + 1 < fa.n_next) f16_prime<D>(ring, w + C::WAVES_N * WSTREAM, loff);
        TC_STAMP(12 + 2 * pass);
        tc_for_quads<D>(U, inv, fa.bn + pass * D, p, [&](int, int, int, int row, int c0, f32x4 v) {
#ifndef SD_TC_ABL_NOSTORE
            if (row < p.R_left) SD_NT_STORE(v, reinterpret_cast<f32x4 *>(fa.y_out + (p.r0 + row) * ldo + pass * D + c0));
#else
            if (v[0] == 1234.5f) fa.y_out[0] = v[1];
#endif
        });
        TC_STAMP(13 + 2 * pass);
    }
}

// ======================================================================================
// backward
// ======================================================================================
template <int D>
__device__ __forceinline__ void tc_acc_to_lds(float *sA, const f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN], const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    const int i4 = p.lane & 3;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float x0 = acc[tm][tn][4 * g], x1 = acc[tm][tn][4 * g + 1], x2 = acc[tm][tn][4 * g + 2], x3 = acc[tm][tn][4 * g + 3];
                quad_transpose(x0, x1, x2, x3, p.lane);
                const int row = p.wm * C::WM + tm * 32 + 8 * g + 4 * p.half + i4;
                const int c0 = p.wn * C::WN + tn * 32 + (p.l31 & ~3);
                *reinterpret_cast<f32x4 *>(sA + row * C::LDA + c0) = f32x4{x0, x1, x2, x3};
            }
}

// panel rows dn (gradient of the LayerNorm output) -> dx = LN-backward(dn; x, gamma) + dres (stored); dgamma / dbeta of the
// 64 rows are reduced in the workgroup and added to dg / db with 2 D atomics
template <int D>
__device__ __forceinline__ void tc_ln_bwd_rows(float *sA, const BwdArgs &fa, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    constexpr int V4 = D / 64;
    const int sub = p.lane & 15, grp = p.lane >> 4;
    f32x4 gam[V4], dgam[V4], dbet[V4];
    float dxmax = 0.f;
#pragma unroll
    for (int j = 0; j < V4; ++j) {
        gam[j] = *reinterpret_cast<const f32x4 *>(fa.ln_w + 4 * (sub + 16 * j));
        dgam[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        dbet[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = p.wave * 4 + grp; row < C::BM; row += 16) {
        const bool valid = row < p.R_left;
        const long at0 = (p.r0 + row) * D;
        f32x4 dn[V4], xv[V4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (sub + 16 * j);
            dn[j] = *reinterpret_cast<const f32x4 *>(sA + row * C::LDA + c);
            xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (valid) xv[j] = *reinterpret_cast<const f32x4 *>(fa.x + at0 + c);
            s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
        }
        const float mean = row16_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            xv[j] = xv[j] - mean;
            const f32x4 sq = xv[j] * xv[j];
            q += (sq[0] + sq[1]) + (sq[2] + sq[3]);
        }
        const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / D) + SD_LN_EPS);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            xv[j] = xv[j] * rstd;                       // xhat
            dgam[j] = dgam[j] + dn[j] * xv[j];
            dbet[j] = dbet[j] + dn[j];
            dn[j] = dn[j] * gam[j];                     // g = dn o gamma
            const f32x4 gx = dn[j] * xv[j];
            s1 += (dn[j][0] + dn[j][1]) + (dn[j][2] + dn[j][3]);
            s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
        }
        const float c1 = row16_sum(s1) * (1.0f / D), c2 = row16_sum(s2) * (1.0f / D);
        if (valid) {
#pragma unroll
            for (int j = 0; j < V4; ++j) {
                const int c = 4 * (sub + 16 * j);
                f32x4 dx = (dn[j] - c1 - xv[j] * c2) * rstd;
                if (fa.dres) dx = dx + *reinterpret_cast<const f32x4 *>(fa.dres + at0 + c);
                *reinterpret_cast<f32x4 *>(fa.dx + at0 + c) = dx;
                dxmax = fmaxf(fmaxf(dxmax, fmaxf(fabsf(dx[0]), fabsf(dx[1]))), fmaxf(fabsf(dx[2]), fabsf(dx[3])));
            }
        }
    }
    if (fa.amax_dx) {   // workgroup-uniform
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) dxmax = fmaxf(dxmax, __shfl_xor(dxmax, o, 64));
        f16_emit_amax(fa.amax_dx, dxmax, p.lane);
    }
    // the 4 row groups of a wave hold the same columns: fold them, then the 4 waves through LDS
#pragma unroll
    for (int j = 0; j < V4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = dgam[j][e], b = dbet[j][e];
            a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
            b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
            dgam[j][e] = a; dbet[j][e] = b;
        }
    __syncthreads();   // the panel is free
    if (grp == 0) {
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (sub + 16 * j);
            *reinterpret_cast<f32x4 *>(sA + (p.wave * 2 + 0) * D + c) = dgam[j];
            *reinterpret_cast<f32x4 *>(sA + (p.wave * 2 + 1) * D + c) = dbet[j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int which = c / D, col = c - which * D;
        const float t = (sA[(0 * 2 + which) * D + col] + sA[(1 * 2 + which) * D + col]) + (sA[(2 * 2 + which) * D + col] + sA[(3 * 2 + which) * D + col]);
        atomicAdd((which ? fa.db : fa.dg) + col, t);
    }
}

template <int D, bool HAS_GELU, bool HAS_LN, bool DROP>
__global__ __launch_bounds__(256, 2) void train_bwd_chain_kernel(BwdArgs fa) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 16, ROWP = 2 * C::LDA;
    constexpr long WSTREAM = (long)NK * C::TN * 2 * 512;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    float *sInv = sA + C::BM * C::LDA;
    const ChainPos<D> p(fa.R);
    const f16 *aH = reinterpret_cast<const f16 *>(sA) + (p.wm * C::WM + p.l31) * ROWP + 8 * p.half;
    const long wOff = (long)__builtin_amdgcn_readfirstlane(p.wn) * WSTREAM;
    const unsigned loff = (unsigned)p.lane * 8;
    f32x16 ACC[C::TM][C::TN], U[C::TM][C::TN];
    f32x4 inv[C::TM][4];
    F16Ring<D> ring;

    for (int pass = 0; pass < fa.passes; ++pass) {
        const f16 *w = fa.wt + (long)pass * C::WAVES_N * WSTREAM + wOff;
        f16_prime<D>(ring, w, loff);
        if (pass) __syncthreads();   // the previous pass's planes are no longer read
        tc_load_rows<D>(sA, fa.dy + pass * D, fa.ldy, p);
        __syncthreads();
        if constexpr (DROP) {   // g = dy o mask (one pass, row pitch D: checked on the host), kept for the weight gradient
            tc_rows_to_planes<D>(sA, sInv, p, fa.amax_dy, [&](int row, int c, f32x4 &v) {
                if (row < p.R_left) {
                    v = v * dropout_quad(fa.d_in, (unsigned long)(p.r0 + row) * (D / 4) + (unsigned long)(c >> 2));
                    if (fa.dym) *reinterpret_cast<f32x4 *>(fa.dym + (p.r0 + row) * D + c) = v;
                }
            });
        } else {
            f16_rows_to_planes<D, false>(sA, sInv, nullptr, nullptr, p.lane, p.wave, nullptr, 0, fa.amax_dy);
        }
        __syncthreads();
        f16_gemm<D, true>(U, aH, w, loff, ring);
        tc_load_inv<D>(inv, sInv, p);
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = U[tm][tn][r] * inv[tm][r >> 2][r & 3];
                    ACC[tm][tn][r] = pass ? ACC[tm][tn][r] + t : t;
                }
    }

    if constexpr (HAS_GELU) {
        f16_prime<D>(ring, fa.wt1 + wOff, loff);
        __syncthreads();
        tc_acc_to_lds<D>(sA, ACC, p);
        __syncthreads();
        tc_rows_to_planes<D>(sA, sInv, p, fa.amax_dpre, [&](int row, int c, f32x4 &v) {   // dpre = t o gelu'(pre) [o mask], kept for dW1
            if (row < p.R_left) {
                const long at = (p.r0 + row) * D + c;
                const f32x4 u = *reinterpret_cast<const f32x4 *>(fa.pre + at);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_fast(u[e]);
                if constexpr (DROP) v = v * dropout_quad(fa.d_act, (unsigned long)(p.r0 + row) * (D / 4) + (unsigned long)(c >> 2));
                *reinterpret_cast<f32x4 *>(fa.dpre + at) = v;
            }
        });
        __syncthreads();
        f16_gemm<D, true>(U, aH, fa.wt1 + wOff, loff, ring);
        tc_load_inv<D>(inv, sInv, p);
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) ACC[tm][tn][r] = U[tm][tn][r] * inv[tm][r >> 2][r & 3];
    }

    if constexpr (HAS_LN) {
        __syncthreads();
        tc_acc_to_lds<D>(sA, ACC, p);
        __syncthreads();
        tc_ln_bwd_rows<D>(sA, fa, p);
    } else {
        const int i4 = p.lane & 3;
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float x0 = ACC[tm][tn][4 * g], x1 = ACC[tm][tn][4 * g + 1], x2 = ACC[tm][tn][4 * g + 2], x3 = ACC[tm][tn][4 * g + 3];
                    quad_transpose(x0, x1, x2, x3, p.lane);
                    const int row = p.wm * C::WM + tm * 32 + 8 * g + 4 * p.half + i4;
                    const int c0 = p.wn * C::WN + tn * 32 + (p.l31 & ~3);
                    if (row < p.R_left) SD_NT_STORE((f32x4{x0, x1, x2, x3}), reinterpret_cast<f32x4 *>(fa.dx + (p.r0 + row) * D + c0));
                }
    }
}

// ======================================================================================
// host
// ======================================================================================
template <int D>
size_t chain_lds() {
    return PanelCfg<D>::LDS_BYTES + PanelCfg<D>::BM * sizeof(float);
}

#define SD_CHAIN_LAUNCH(KFN, D_, ARGS)                                                                             \
    do {                                                                                                           \
        auto kfn = KFN;                                                                                            \
        const size_t lds = chain_lds<D_>();                                                                        \
        static DevFlag attr_set;                                                                                     \
        if (lds > 64 * 1024 && !attr_set) {                                                                        \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
            attr_set = true;                                                                                       \
        }                                                                                                          \
        dim3 grid((unsigned)((ARGS.R + PanelCfg<D_>::BM - 1) / PanelCfg<D_>::BM)), block(256);                     \
        SD_LAUNCH(kfn, grid, block, lds, s, ARGS);                                                                 \
    } while (0)

template <int D>
int launch_fwd(const FwdArgs &fa, bool has_out, bool has_ffn, bool drop, hipStream_t s) {
    if (has_out && has_ffn) {
        if (drop) SD_CHAIN_LAUNCH((train_fwd_chain_kernel<D, true, true, true>), D, fa);
        else SD_CHAIN_LAUNCH((train_fwd_chain_kernel<D, true, true, false>), D, fa);
    } else if (has_out) {
        if (drop) SD_CHAIN_LAUNCH((train_fwd_chain_kernel<D, true, false, true>), D, fa);
        else SD_CHAIN_LAUNCH((train_fwd_chain_kernel<D, true, false, false>), D, fa);
    } else {
        SD_CHAIN_LAUNCH((train_fwd_chain_kernel<D, false, false, false>), D, fa);
    }
    SD_CHECK_LAUNCH("train_fwd_chain_kernel");
    return 0;
}

template <int D>
int launch_bwd(const BwdArgs &fa, bool has_gelu, bool has_ln, bool drop, hipStream_t s) {
    if (has_gelu) {
        if (drop) SD_CHAIN_LAUNCH((train_bwd_chain_kernel<D, true, true, true>), D, fa);
        else SD_CHAIN_LAUNCH((train_bwd_chain_kernel<D, true, true, false>), D, fa);
    } else if (has_ln) {
        if (drop) SD_CHAIN_LAUNCH((train_bwd_chain_kernel<D, false, true, true>), D, fa);
        else SD_CHAIN_LAUNCH((train_bwd_chain_kernel<D, false, true, false>), D, fa);
    } else {
        if (drop) SD_CHAIN_LAUNCH((train_bwd_chain_kernel<D, false, false, true>), D, fa);
        else SD_CHAIN_LAUNCH((train_bwd_chain_kernel<D, false, false, false>), D, fa);
    }
    SD_CHECK_LAUNCH("train_bwd_chain_kernel");
    return 0;
}

}  // namespace

extern "C" int sd_train_fwd_chain(const sd_train_fwd_chain_args *a, void *stream) {
    if (!a || a->R <= 0 || !a->h_in) return fail(SD_E_BADARG, "sd_train_fwd_chain: bad argument");
    const bool has_out = a->a != nullptr, has_ffn = a->w1 != nullptr, has_next = a->n_next > 0;
    if (!has_out && has_ffn) return fail(SD_E_BADARG, "sd_train_fwd_chain: the feed-forward stage needs the out-projection stage");
    if (!has_out && !has_next) return fail(SD_E_BADARG, "sd_train_fwd_chain: nothing to do");
    if (has_out && (!a->wo || !a->bo || !a->h_out)) return fail(SD_E_BADARG, "sd_train_fwd_chain: out-projection stage needs wo, bo, h_out");
    if (has_ffn && (!a->ln_w || !a->ln_b || !a->n_out || !a->b1 || !a->pre || !a->u || !a->w2 || !a->b2 || !a->h2_out))
        return fail(SD_E_BADARG, "sd_train_fwd_chain: feed-forward stage needs ln_w, ln_b, n_out, b1, pre, u, w2, b2, h2_out");
    if (a->n_next < 0 || a->n_next > 3 || (has_next && (!a->nln_w || !a->nln_b || !a->nn_out || !a->wn || !a->bn || !a->y_out)))
        return fail(SD_E_BADARG, "sd_train_fwd_chain: next-projection stage needs nln_w, nln_b, nn_out, wn, bn, y_out and n_next in 1..3");
    if (!(a->p >= 0.f) || !(a->p < 1.f)) return fail(SD_E_BADARG, "sd_train_fwd_chain: p must be in [0, 1)");
    FwdArgs fa;
    fa.R = a->R; fa.n_next = a->n_next;
    fa.a = a->a; fa.wo = (const f16 *)a->wo; fa.bo = a->bo; fa.h_in = a->h_in; fa.h_out = a->h_out;
    fa.ln_w = a->ln_w; fa.ln_b = a->ln_b; fa.n_out = a->n_out; fa.w1 = (const f16 *)a->w1; fa.b1 = a->b1; fa.pre = a->pre; fa.u = a->u;
    fa.w2 = (const f16 *)a->w2; fa.b2 = a->b2; fa.h2_out = a->h2_out;
    fa.nln_w = a->nln_w; fa.nln_b = a->nln_b; fa.nn_out = a->nn_out; fa.wn = (const f16 *)a->wn; fa.bn = a->bn; fa.y_out = a->y_out;
    fa.d_out = make_dropout(a->p, a->seed, a->site_out);
    fa.d_act = make_dropout(a->p, a->seed, a->site_act);
    fa.d_ffn = make_dropout(a->p, a->seed, a->site_ffn);
    fa.amax_a = a->amax_a; fa.amax_n = a->amax_n; fa.amax_u = a->amax_u; fa.amax_nn = a->amax_nn; fa.amax_h2 = a->amax_h2;
    const bool drop = a->p > 0.f && has_out;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(SD_KCLASS_LAYER_CHAIN, s);
    switch (a->d) {
        case 64: return launch_fwd<64>(fa, has_out, has_ffn, drop, s);
        case 128: return launch_fwd<128>(fa, has_out, has_ffn, drop, s);
        case 256: return launch_fwd<256>(fa, has_out, has_ffn, drop, s);
    }
    return fail(SD_E_BADDIM, "sd_train_fwd_chain: hidden_dim must be one of 64, 128, 256");
}

extern "C" int sd_train_bwd_chain(const sd_train_bwd_chain_args *a, void *stream) {
    if (!a || a->R <= 0 || !a->dy || !a->wt || !a->dx || a->passes < 1 || a->passes > 3) return fail(SD_E_BADARG, "sd_train_bwd_chain: bad argument");
    const bool has_gelu = a->pre != nullptr, has_ln = a->x != nullptr;
    if (a->ldy < a->passes * a->d || a->ldy % 4 != 0) return fail(SD_E_BADARG, "sd_train_bwd_chain: ldy must be >= passes * d and a multiple of 4");
    if (has_gelu && (!a->dpre || !a->wt1 || !has_ln)) return fail(SD_E_BADARG, "sd_train_bwd_chain: the GELU stage needs dpre, wt1 and the LayerNorm stage");
    if (has_ln && (!a->ln_w || !a->dg || !a->db)) return fail(SD_E_BADARG, "sd_train_bwd_chain: the LayerNorm stage needs ln_w, dg, db");
    if (!(a->p >= 0.f) || !(a->p < 1.f)) return fail(SD_E_BADARG, "sd_train_bwd_chain: p must be in [0, 1)");
    const bool drop = a->p > 0.f && (has_gelu || !has_ln);   // the LayerNorm + projection chain's dY comes from an attention core: never masked
    if (drop && (a->passes != 1 || a->ldy != a->d)) return fail(SD_E_BADARG, "sd_train_bwd_chain: a masked dy has one pass and row stride d");
    BwdArgs fa;
    fa.R = a->R; fa.passes = a->passes; fa.ldy = a->ldy;
    fa.dy = a->dy; fa.dym = a->dym; fa.wt = (const f16 *)a->wt;
    fa.pre = a->pre; fa.dpre = a->dpre; fa.wt1 = (const f16 *)a->wt1;
    fa.x = a->x; fa.ln_w = a->ln_w; fa.dres = a->dres; fa.dg = a->dg; fa.db = a->db; fa.dx = a->dx;
    fa.d_in = make_dropout(a->p, a->seed, a->site_in);
    fa.d_act = make_dropout(a->p, a->seed, a->site_act);
    fa.amax_dy = a->amax_dy; fa.amax_dpre = a->amax_dpre; fa.amax_dx = a->amax_dx;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(SD_KCLASS_LAYER_CHAIN, s);
    switch (a->d) {
        case 64: return launch_bwd<64>(fa, has_gelu, has_ln, drop, s);
        case 128: return launch_bwd<128>(fa, has_gelu, has_ln, drop, s);
        case 256: return launch_bwd<256>(fa, has_gelu, has_ln, drop, s);
    }
    return fail(SD_E_BADDIM, "sd_train_bwd_chain: hidden_dim must be one of 64, 128, 256");
}
