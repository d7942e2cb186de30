// Trajectory-owning decoder kernels of the sampler (hidden_dim 256, 4 heads, horizon <= 100, <= 64 memory rows): ONE
// workgroup of 8 waves carries ONE trajectory through a whole denoiser step - embedding, every decoder layer with its
// self-attention (reference: nn.TransformerDecoderLayer as built by soccer_diffusion/ml/model/decoder.py:26-35, norm_first;
// forward of decoder.py:38-54), fc_out and the DDIM update - so the residual stream, q | k | v and the attention output
// never leave the CU.  Same numerics as sd_f16x3.h: every product is three fp16 MFMAs on hi / lo operand pairs with fp32
// accumulation; the cross-attention is the folded form of sd_kernels.hip (xattn_fold_kernel).
//
// Geometry.  Every row GEMM is computed TRANSPOSED, out^T[n][token] = W[n][:] . X[token][:], on v_mfma_f32_16x16x32_f16:
// T = 100 pads to 7 token tiles of 16 (12 %; 32-row tiles would pad 28 %).  A = a 16-feature tile of W (fragment-major planes in
// HBM / L2, 1 KiB per wave load), B = X^T from the LDS panel.  Wave w owns output features 32w .. 32w+31 (two n-tiles) of
// all 7 token tiles: a weight fragment is read by exactly one wave and reused 7 times from registers.  The accumulator
// has the token on the lane and 4 consecutive features in its registers, so
//   * LayerNorm statistics come from the accumulators (mean and centred sum of squares of a wave's 32 features, combined
//     over the 8 waves by Chan's formula through 6.4 KB of LDS) - no fp32 round trip of the panel;
//   * the residual stream lives in registers for the whole step, pre-multiplied by the (power of two) scale of the
//     GEMM that accumulates into it.
// Self-attention, per head: [Q_h | K_h | V_h]^T = 12 n-tiles; wave w computes Q tile w (w < 4) or K tile w-4 for all tokens
// and a half (by tokens) of V tile w>>1, from ONE fp16 plane of LayerNorm 1's output (2 MFMAs per product: the only site
// where the measured rollout error allows it, NOTEBOOK.md 5.11).  Q, K, V, O have an LDS buffer each (split planes), so a head is
// two barrier-delimited phases: W = write Q | K | V of head h (+ out-projection of head h-1 into the residual registers),
// X = S^T = K Q^T (keys x queries) per query tile on waves 0..6, softmax in registers (a query is a lane column), O^T = V^T P^T
// with P^T straight from the score accumulators (B operand) and V^T through ds_read_b64_tr_b16, O -> LDS; the projection GEMM
// of head h+1 is issued BEFORE the attention by waves 0..3 and AFTER it by waves 4..7, so that the two waves of a SIMD are in
// complementary (matrix / vector) jobs most of the time.
// Folded cross-attention: S^T_h = G_h LN2(h)^T is ONE 16 x 16 tile per (head, token tile) (16 key slots); softmax over the
// accumulator rows; P -> LDS [token][head*16 + slot | step columns]; H += V'^T P^T as a column-split GEMM with K = 96.
//
// LDS (163 200 B of 163 840, map below): X panel 100 x 1 KiB (hi | lo, 16-byte chunks XOR-swizzled by token & 15: every
// ds_read_b128 of a fragment is conflict-free), K 100 x 256 B, V 100 x 288 B (together: the 100 x 512 B probabilities of the
// cross-attention), Q / O 100 x 256 B each inside the panel's second half during the self-attention block, LayerNorm partials.  Rows >= T are never stored; reads of padded tokens clamp to row T-1 (finite values;
// padded keys are masked, padded queries never leave the workgroup).
#pragma once
#include "sd_common.h"
#include "soccerdiffusion_hip.h"   // SD_STATUS_SHARP_LOGITS, SD_SHARP_LOGIT_LIMIT
#include <type_traits>

namespace tj {

constexpr int D = 256, HD = 64, NH = 4, TMAX = 100;
constexpr int NTT_A = 7;                    // token tiles of the Stage-A experiment kernel (T = 97 .. 100)
constexpr int NTHREADS = 512;
constexpr int MAX_L = 8;
constexpr float ACT = 8.0f;                 // scale of LayerNorm outputs, q, k, v, attention / GELU outputs (as sd_f16x3.h)
constexpr float PSC = 1024.0f;              // scale of the cross-attention probabilities
constexpr float XSC = 1.0f;                 // scale of the trajectory values x at the embedding: x has no a-priori bound (an untrained
                                            // denoiser drives |x| into the hundreds); |x| < 65 504, and below 0.125 the lo part's
                                            // absolute error is 3e-8 - under fp32's own rounding of an O(1) sum
constexpr int XROW = 1024, X1ROW = 512, QROW = 256, VROW = 288, PROW = 512;
// LDS map.  Outside the self-attention block: [X panel 100 x 1 KiB (hi | lo)] [cross-attention probabilities 100 x 512 B] [stats].
// Inside it the panel holds LayerNorm 1's output as ONE plane (100 x 512 B: the Q | K | V projection reads the hi part only,
// see sa_block) and the freed half takes Q and O, so that Q, K, V and O of a head each have a buffer of their own.
constexpr int LDS_X = 0;
constexpr int LDS_SQ = LDS_X + TMAX * X1ROW;   // Q of the current head          (inside the X panel's second half)
constexpr int LDS_SO = LDS_SQ + TMAX * QROW;   // attention output of the head   (likewise)
constexpr int LDS_Q = LDS_X + TMAX * XROW;     // K of the current head; the x rows at the embedding
constexpr int LDS_K = LDS_Q + TMAX * QROW;     // V of the current head
constexpr int LDS_P = LDS_Q;                   // cross-attention probabilities: 100 x 512 B over both
constexpr int LDS_STAT = LDS_K + TMAX * VROW;
constexpr int LDS_BYTES = LDS_STAT + TMAX * 8 * 8;
static_assert(LDS_SO + TMAX * QROW <= LDS_Q, "Q and O fit the freed half of the panel");
static_assert(LDS_BYTES <= 163840 && LDS_P + TMAX * PROW <= LDS_STAT, "LDS budget");
constexpr long HFRAG_FLOATS = 8L * 2 * NTT_A * 256;   // residual stream of one trajectory in fragment order (Stage-A kernel)

typedef short s16x4 __attribute__((ext_vector_type(4)));

// diagnostic build (-DTJ_STAMPS): every wave of the first TJ_STAMP_WGS workgroups records the shader clock at phase boundaries
#ifdef TJ_STAMPS
constexpr int TJ_NSTAMP = 96, TJ_STAMP_WGS = 512;
__device__ unsigned long long *g_tj_stamps;
#define TJ_STAMP(i)                                                                                                            \
    do {                                                                                                                       \
        if (blockIdx.x < TJ_STAMP_WGS && (threadIdx.x & 63) == 0)                                                              \
            g_tj_stamps[((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * TJ_NSTAMP + (i)] = __builtin_amdgcn_s_memtime();        \
    } while (0)
// TJ_SYNC(site): a workgroup barrier that adds the cycles this wave spent in it to slot 64 + site (sites: 0 / 1 LayerNorm
// exchange / panel complete, 2 / 3 self-attention phase W / X, 4 / 5 cross-attention, 6 / 7 feed-forward, 8 .. 10 embedding, tail)
template <int SITE>
__device__ __forceinline__ void tj_sync() {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x < TJ_STAMP_WGS && (threadIdx.x & 63) == 0) g_tj_stamps[((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * TJ_NSTAMP + 64 + SITE] += t1 - t0;
}
#define TJ_SYNC(site) tj_sync<site>()
#else
#define TJ_STAMP(i) do {} while (0)
#define TJ_SYNC(site) __syncthreads()
#endif

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// c += (ah + al) (bh + bl) without lo.lo, small terms first.
// Precision experiment (tools/exp/precision_sites.sh, NOTEBOOK.md 5.11): -DTJ_DROP_ALO=<mask> / -DTJ_DROP_BLO=<mask> drop the
// A-lo x B-hi / A-hi x B-lo product at the GEMM sites whose bit is set (A = weights, K, V^T, G, V'^T; B = activations, Q, P).
#ifndef TJ_DROP_ALO
#define TJ_DROP_ALO 0
#endif
#ifndef TJ_DROP_BLO
#define TJ_DROP_BLO 0
#endif
enum Site { S_QKV = 1, S_SCORES = 2, S_PV = 4, S_OUT = 8, S_XSC = 16, S_XPV = 32, S_W1 = 64, S_W2 = 128, S_EMB = 256, S_FC = 512 };
template <int SITE = 0>
__device__ __forceinline__ void mma3(f32x4 &c, f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl) {
    if constexpr (!(TJ_DROP_ALO & SITE)) c = mfma16(al, bh, c);
    if constexpr (!(TJ_DROP_BLO & SITE)) c = mfma16(ah, bl, c);
    c = mfma16(ah, bh, c);
}

// all-reduce over the four 16-lane rows of a wave (lanes t, t+16, t+32, t+48): two v_permlane*_swap, no LDS round trip.
// (__builtin_amdgcn_permlane32_swap(v, v) folds its two results into one on ROCm 7.2: inline assembly, checked on gfx950
// by tools/exp/perm_test.hip.)
__device__ __forceinline__ float rows4_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = a + b;
    a = v;
    b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float rows4_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    v = fmaxf(a, b);
    a = v;
    b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// x (already scaled) as hi = fp16(x), lo = fp16(x - hi): v_cvt_pk_f16_f32 both ways, 3 VALU instructions per element
__device__ __forceinline__ void split4(const f32x4 &x, f16x4 &h, f16x4 &l) {
    h = __builtin_convertvector(x, f16x4);
    l = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), f16x4);
}
__device__ __forceinline__ void split_store(char *hi_at, char *lo_at, const f32x4 &v) {
    f16x4 h, l;
    split4(v, h, l);
    *reinterpret_cast<f16x4 *>(hi_at) = h;
    *reinterpret_cast<f16x4 *>(lo_at) = l;
}

// ---------------------------------------------------------------------------------------------------
// once-per-call packing (host side: sd_kernels.hip, sampler mode 3)
// ---------------------------------------------------------------------------------------------------
// W (N x K row-major fp32, zero-padded to Np x Kp with Np % 16 == 0, Kp % 32 == 0) -> [n-tile][k-step][plane][lane][8]:
// lane = 16 g + i holds W[16 nt + i][32 ks + 8 g + 0..7] * scale as hi / lo.  maxbits: abs-max word (scale derived on the
// device) or NULL with a fixed scale.
// The order of the contraction index inside the activation panels.  A lane of an accumulator tile holds four consecutive
// features of its token for EACH of the wave's two n-tiles (32 w + 4 g + r and 32 w + 16 + 4 g + r): stored side by side they are one
// 16-byte slot, i.e. one ds_write_b128 per plane and token tile, conflict-free (8-lane groups over 32 banks), where two
// ds_write_b64 - 16-lane groups sharing the same half of their slots - always collide two-way (- 6 % of the step with conflict-free
// stores in a timing experiment).  So position 8 g + e of a 32-feature k-step holds feature 16 (e >> 2) + 4 g + (e & 3); every weight
// matrix whose contraction runs over such a panel (K = 256: in_proj, out_proj, linear1, linear2, fc_out, the folded keys) is packed in
// that order.
__device__ __forceinline__ int kperm(int k8, int e) { return (k8 >> 2) * 32 + 16 * (e >> 2) + 4 * (k8 & 3) + (e & 3); }

static __global__ void pack_w16_kernel(const float *__restrict__ W, int N, int K, int Np, int Kp, const unsigned *maxbits, float fixed_scale,
                                f16 *__restrict__ dst, float *scale_out) {
    const float scale = maxbits ? f16_scale_from_bits(*maxbits) : fixed_scale;
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const int nks = Kp / 32;
    const long total = (long)Np * (Kp / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / (Kp / 8)), k8 = (int)(i % (Kp / 8));
        f16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = Kp == D ? kperm(k8, e) : k8 * 8 + e;
            const float v = (n < N && k < K) ? W[(long)n * K + k] * scale : 0.f;
            hh[e] = (f16)v;
            ll[e] = (f16)(v - (float)hh[e]);
        }
        const int nt = n >> 4, ks = k8 >> 2, lane = (k8 & 3) * 16 + (n & 15);
        f16 *o = dst + (((long)nt * nks + ks) * 2) * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = hh[e];
            o[512 + e] = ll[e];
        }
    }
}

// Folded keys of the context rows: gv rows [(item * 4 + head) * 16 + slot][2 D] (G in the first D columns) ->
// per (item, head) blocks [ks 8][plane][lane = 16 g + slot][8], k = 32 ks + 8 g + e.  Slots >= n_slots are zero.
// items = trajectories x key tiles (tile kt of a trajectory = item % nkt; its slot s is memory row 16 kt + s, valid below n_slots)
static __global__ void pack_g16_kernel(const float *__restrict__ gv, long items, int n_slots, const unsigned *maxbits, f16 *__restrict__ dst,
                                float *scale_out, int nkt = 1) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const long total = items * 4 * 16 * (D / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k8 = (int)(i % (D / 8)), slot = (int)((i / (D / 8)) % 16);
        const long ih = i / (D / 8) / 16;
        f16x4 h0 = {0, 0, 0, 0}, l0 = h0, h1 = h0, l1 = h0;
        if (16 * (int)((ih >> 2) % nkt) + slot < n_slots) {
            const float *row = gv + (ih * 16 + slot) * 2 * D + kperm(k8, 0);   // features .. + 3 and + 16 .. + 19 (kperm)
            f16_split4(*reinterpret_cast<const f32x4 *>(row), scale, h0, l0);
            f16_split4(*reinterpret_cast<const f32x4 *>(row + 16), scale, h1, l1);
        }
        f16 *o = dst + ih * (8 * 2 * 512) + ((k8 >> 2) * 2) * 512 + ((k8 & 3) * 16 + slot) * 8;
        *reinterpret_cast<f16x4 *>(o) = h0;
        *reinterpret_cast<f16x4 *>(o + 4) = h1;
        *reinterpret_cast<f16x4 *>(o + 512) = l0;
        *reinterpret_cast<f16x4 *>(o + 516) = l1;
    }
}
// Folded keys of the step tokens: gvstep rows [item * 4 + head][2 D] -> [item][head][ks][plane][g][8]
static __global__ void pack_gstep16_kernel(const float *__restrict__ gvstep, long items, const unsigned *maxbits, f16 *__restrict__ dst,
                                    float *scale_out) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const long total = items * 4 * (D / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k8 = (int)(i % (D / 8));
        const long ih = i / (D / 8);
        f16x4 h0, l0, h1, l1;
        const float *row = gvstep + ih * 2 * D + kperm(k8, 0);
        f16_split4(*reinterpret_cast<const f32x4 *>(row), scale, h0, l0);
        f16_split4(*reinterpret_cast<const f32x4 *>(row + 16), scale, h1, l1);
        f16 *o = dst + ih * (8 * 2 * 32) + ((k8 >> 2) * 2) * 32 + (k8 & 3) * 8;
        *reinterpret_cast<f16x4 *>(o) = h0;
        *reinterpret_cast<f16x4 *>(o + 4) = h1;
        *reinterpret_cast<f16x4 *>(o + 32) = l0;
        *reinterpret_cast<f16x4 *>(o + 36) = l1;
    }
}
// Folded values of the context rows -> per item [n-tile 16][kk 2][plane][lane = 16 g + i][8]: V'^T[n = 16 nt + i][k = 32 kk + 8 g + e],
// k = head * 16 + slot
static __global__ void pack_v16_kernel(const float *__restrict__ gv, long items, int n_slots, const unsigned *maxbits, f16 *__restrict__ dst,
                                float *scale_out, int nkt = 1) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const long total = items * 16 * 2 * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), kk = (int)((i >> 6) & 1), nt = (int)((i >> 7) & 15);
        const long item = i >> 11;
        const int n = 16 * nt + (lane & 15), g = lane >> 4;
        f16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * kk + 8 * g + e, head = k >> 4, slot = k & 15;
            const float v = 16 * (int)(item % nkt) + slot < n_slots ? gv[((item * 4 + head) * 16 + slot) * 2 * D + D + n] * scale : 0.f;
            hh[e] = (f16)v;
            ll[e] = (f16)(v - (float)hh[e]);
        }
        f16 *o = dst + (((item * 16 + nt) * 2 + kk) * 2) * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = hh[e];
            o[512 + e] = ll[e];
        }
    }
}
// Folded values of the step tokens -> [item][plane][head][n]
static __global__ void pack_vstep16_kernel(const float *__restrict__ gvstep, long items, const unsigned *maxbits, f16 *__restrict__ dst,
                                           float *scale_out = nullptr) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const long total = items * 4 * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % D), head = (int)((i / D) & 3);
        const long item = i / D / 4;
        const float v = gvstep[(item * 4 + head) * 2 * D + D + n] * scale;
        const f16 h = (f16)v;
        dst[(item * 2 + 0) * 4 * D + head * D + n] = h;
        dst[(item * 2 + 1) * 4 * D + head * D + n] = (f16)(v - (float)h);
    }
}

// row-major [B][T][256] <-> fragment order [B][wave][a][tt][lane][4] (Stage-A kernel and tests): element r of lane 16 g + t
// is feature 32 w + 16 a + 4 g + r of token 16 tt + t (tokens >= T: zero)
static __global__ void to_hfrag_kernel(const float *__restrict__ rows, float *__restrict__ frag, int B, int T) {
    const long total = (long)B * 8 * 2 * NTT_A * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long j = i >> 6;
        const int tt = (int)(j % NTT_A); j /= NTT_A;
        const int a = (int)(j & 1); j >>= 1;
        const int w = (int)(j & 7);
        const long b = j >> 3;
        const int tok = 16 * tt + (lane & 15), n = 32 * w + 16 * a + 4 * (lane >> 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tok < T) v = *reinterpret_cast<const f32x4 *>(rows + ((long)b * T + tok) * D + n);
        *reinterpret_cast<f32x4 *>(frag + i * 4) = v;
    }
}
static __global__ void from_hfrag_kernel(const float *__restrict__ frag, float *__restrict__ rows, int B, int T) {
    const long total = (long)B * 8 * 2 * NTT_A * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        long j = i >> 6;
        const int tt = (int)(j % NTT_A); j /= NTT_A;
        const int a = (int)(j & 1); j >>= 1;
        const int w = (int)(j & 7);
        const long b = j >> 3;
        const int tok = 16 * tt + (lane & 15), n = 32 * w + 16 * a + 4 * (lane >> 4);
        if (tok < T) *reinterpret_cast<f32x4 *>(rows + ((long)b * T + tok) * D + n) = *reinterpret_cast<const f32x4 *>(frag + i * 4);
    }
}

// ---------------------------------------------------------------------------------------------------
// device pieces
// ---------------------------------------------------------------------------------------------------
// Requires 96 < T <= 100: token tiles 0..5 are full, tile 6 holds tokens 96 .. T-1 and its other lanes clamp to T-1.
// A 16-byte chunk of a row is addressed as  row base + ((g ^ key[1:0]) << 4) + ((m2 ^ key[3:2]) << 6) + constant,  key =
// token & 15, m2 = two compile-time chunk bits: four address registers per row pitch cover every fragment of tiles 0..5
// (plus a multiple of 16 rows as an immediate), four more the clamped tile.
struct Ctx {
    char *smem;
    int lane, w, g, t, T;
    int tok6;            // this lane's token of tile 6, clamped to T - 1
    bool ok6;            // ... and whether it exists
    unsigned xa[4], xa6[4];   // X panel: m2 = plane | (ks & 1) << 1   (+ (ks >> 1) * 256 + tt * 16 KiB)
    // Q-layout rows (256 B; m2 = plane | kk << 1) and P rows (512 B; m2 = plane | (kk & 1) << 1, + (kk >> 1) * 256) hold the same
    // in-row part as the X panel at a smaller row pitch: derived from xa where they are used (one VALU instruction per address)
};

struct LnAffine { f32x4 w[2], b[2]; };
__device__ __forceinline__ void ln_affine_load(const Ctx &c, const float *ln_w, const float *ln_b, LnAffine &p) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        p.w[a] = *reinterpret_cast<const f32x4 *>(ln_w + 32 * c.w + 16 * a + 4 * c.g);
        p.b[a] = *reinterpret_cast<const f32x4 *>(ln_b + 32 * c.w + 16 * a + 4 * c.g);
    }
    __builtin_amdgcn_sched_barrier(0);
}


// Short GEMM loops (K = 64 / 96 / 112: the attention and the out-projections) as a software pipeline: load(s) fills ring slot
// s % DEPTH with the LDS operands of step s and is issued DEPTH - 1 steps before mma(s) consumes them (hipcc places every
// ds_read right before its first use and waits for it: ~100 cycles of LDS round trip per 48 - 96 cycles of MFMAs).
template <int NSTEP, int DEPTH, class Load, class Mma>
__device__ __forceinline__ void ring_pipe(Load load, Mma mma) {
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s)
        if (s < NSTEP) load(s);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        if (s + DEPTH - 1 < NSTEP) load(s + DEPTH - 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(s);
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct AK64 { f16x8 a[2][2][2]; };
__device__ __forceinline__ void load_k64(const Ctx &c, AK64 &f, const f16 *pa0, const f16 *pa1) {
    const unsigned lo = (unsigned)c.lane * 8;
    __builtin_amdgcn_sched_barrier(0);   // not earlier than here (32 registers)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            f.a[0][kk][pl] = *reinterpret_cast<const f16x8 *>(pa0 + lo + (kk * 2 + pl) * 512);
            f.a[1][kk][pl] = *reinterpret_cast<const f16x8 *>(pa1 + lo + (kk * 2 + pl) * 512);
        }
    __builtin_amdgcn_sched_barrier(0);
}

struct SaW {
    const f16 *w_in;      // in_proj (768 x 256) planes
    const float *b_in;    // 768
    const f16 *w_o;       // out_proj (256 x 256) planes
    float s_in;           // power-of-two scale of w_in
    float scale_log2e;    // log2(e) / sqrt(head dim)
    int *status;          // range-guard word of sd_ddim_sample_ex or NULL: SD_STATUS_SHARP_LOGITS
};

struct Bias2 { f32x4 v[2]; };
__device__ __forceinline__ Bias2 bias_load(const Ctx &c, const float *bias) {
    Bias2 b;
#pragma unroll
    for (int a = 0; a < 2; ++a) b.v[a] = *reinterpret_cast<const f32x4 *>(bias + 32 * c.w + 16 * a + 4 * c.g);
    __builtin_amdgcn_sched_barrier(0);
    return b;
}

// ---------------------------------------------------------------------------------------------------
// one decoder layer on the residual registers (X holds LN1(h) on entry; on exit LN1 of the next layer if nln_w)
// ---------------------------------------------------------------------------------------------------
struct LayerW {
    const float *n2_w, *n2_b, *n3_w, *n3_b;
    const f16 *w_in, *w_o, *w_1, *w_2;            // fragment-major planes
    const float *b_in, *b_o, *b_1, *b_2, *b_oc;
    const float *sc;                              // [0] Wo, [1] W1, [2] W2, [3] in_proj, [4] G, [5] V', [6] G of the step token, [7] its V'
    const f16 *g16, *v16;                         // folded context blocks of this layer, all trajectories
    const float *cb;                              // [B][64] score biases
    const f16 *gstep, *vstep;                     // this layer and step (trajectory 0's block when the step tokens are per trajectory)
    const float *cstep;                           // 4 score biases of the step token
    const float *nln_w, *nln_b;                   // LayerNorm that follows (next layer's norm1), or NULL
};

// halfs of one step token's folded blocks (per layer): G [head 4][ks 8][plane 2][g 4][8], V' [plane 2][head 4][D]
constexpr long STEP_G_HALFS = 4 * 8 * 2 * 32, STEP_V_HALFS = 2 * 4 * D;

// The step token's folded blocks carry scales of their own (sc[6], sc[7]): the context blocks are packed once per context, before the
// step tokens of later calls are known (sd_sampler_prepare / sd_sampler_eps).  Scores: the accumulator row of the step token's slot is
// multiplied by c_gs instead of 1 / (ACT sc[4]).  Values: H accumulates at the scale PSC s_v with s_v = min(sc[5], sc[7]); the
// probabilities of the context slots are multiplied by m_c = s_v / sc[5], the step token's by m_s = s_v / sc[7] - powers of two <= 1, one of
// them 1: the kind with the larger values is exact, the other loses only what is below the larger kind's resolution.
struct StepScale { float c_gs, s_v, m_c, m_s; };
__device__ __forceinline__ StepScale step_scale(const LayerW &L) {
    const float s5 = L.sc[5], s7 = L.sc[7];
    const float sv = fminf(s5, s7);
    return StepScale{1.0f / (ACT * L.sc[6]), sv, sv / s5, sv / s7};
}

// ---------------------------------------------------------------------------------------------------
// Stage-A experiment kernel (tools/exp/traj_layer.hip): h' = h + SelfAttention(LN1(h)), h in fragment order
// ---------------------------------------------------------------------------------------------------
struct SaArgs {
    const float *h_in;
    float *h_out;
    const float *ln_w, *ln_b;
    const f16 *w_in;
    const float *b_in;
    const f16 *w_o;
    const float *b_o;
    float s_in, s_o;
    float scale_log2e;
    int T, B;
};

// ---------------------------------------------------------------------------------------------------
// One whole denoiser step of the sampler per launch: x -> embedding + positional rows -> L decoder layers -> fc_out -> DDIM update
// of x in place (reference loop: soccer_diffusion/ml/inference/plot.py:122-131 around model.py:159-179)
// ---------------------------------------------------------------------------------------------------
struct StepArgs {
    float *x;                      // [B][T][J] in / out
    float *eps_out;                // [B][T][J] or NULL (noise prediction, for tests)
    const f16 *w_emb;              // [16 n-tiles][1][2][64][8] (K = J padded to 32), scale s_emb
    const float *b_emb, *pe;       // bias [256], positional table [>= T][256]
    const float *n1_w, *n1_b;      // layer 0's norm1
    const f16 *w_out;              // [2 n-tiles][8][2][64][8] (rows >= J zero), scale s_out
    const float *b_out;            // [J]
    const float *sc_io;            // [0] s_emb, [1] s_out
    float c0, c1, c2, c3;          // DDIM coefficients of this step (sqrt a_t, sqrt(1 - a_t), sqrt a_prev, sqrt(1 - a_prev))
    float scale_log2e;
    int T, B, J, L, Mk, update_x;
    int nkt;                       // key tiles of 16 memory slots (1; 2 .. 4 only for the WIDE instantiations: 17 .. 64 memory rows)
    const int *step_map;           // step_per_traj: trajectory b reads step block step_map[b] (duplicates of token 0 are folded once), or NULL: block b
    int step_per_traj;             // 0: one step token for the whole batch (the sampler's loop); 1: trajectory b reads block b of gstep / vstep /
                                   // cstep (forward_with_context with a step per sample: sd_sampler_eps)
    int *status;                   // range-guard word (SD_STATUS_SHARP_LOGITS) or NULL
    LayerW layer[MAX_L];
};

// ---------------------------------------------------------------------------------------------------
// The kernel family.  NTT = ceil(T / 16) token tiles (T <= 100: 1 .. 7; tiles 0 .. NTT-2 are full, the last holds tokens
// 16 (NTT-1) .. T-1 and its other lanes clamp to T-1).  PRECISE: the Q | K | V projection reads both planes of LayerNorm 1 (three
// products, as every other site) - see sa_block_precise.
// ---------------------------------------------------------------------------------------------------
template <int NTT, bool PRECISE>
struct TJ {
static constexpr int LAST0 = 16 * (NTT - 1);   // first token of the last tile
static constexpr int NH0 = (NTT + 1) / 2;       // token tiles of the first half (even waves); the odd waves take NTT - NH0
static constexpr int NKP = (NTT + 1) / 2;       // key-tile pairs (32 keys) of the P V product


static __device__ __forceinline__ void ctx_init(Ctx &c, char *smem, int T) {
    c.smem = smem;
    c.lane = threadIdx.x & 63;
    c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.g = c.lane >> 4;
    c.t = c.lane & 15;
    c.T = T;
    c.ok6 = LAST0 + c.t < T;
    c.tok6 = c.ok6 ? LAST0 + c.t : T - 1;
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        const unsigned in_row = (unsigned)(((c.g ^ (c.t & 3)) << 4) + ((m2 ^ (c.t >> 2)) << 6));
        const unsigned in_row6 = (unsigned)(((c.g ^ (c.tok6 & 3)) << 4) + ((m2 ^ ((c.tok6 >> 2) & 3)) << 6));
        c.xa[m2] = (unsigned)(c.t * XROW) + in_row;
        c.xa6[m2] = (unsigned)(c.tok6 * XROW) + in_row6;
    }
}

// A copy of the context whose per-lane values the compiler must treat as new: every address derived from them is then
// computed inside the loop iteration / phase that uses it.  Without this hipcc hoists ~100 loop-invariant LDS addresses out
// of the head and layer loops, keeps them live across everything and spills as many registers.
static __device__ __forceinline__ Ctx ctx_local(const Ctx &c) {
    // recomputed from the lane number (one laundered register) rather than copied: the 12 per-lane values of the caller's
    // context then need not stay in registers across the phases (~25 VALU instructions per phase entry)
    Ctx d;
    d.smem = c.smem;
    d.lane = c.lane;
    asm volatile("" : "+v"(d.lane));
    d.w = c.w;
    d.g = d.lane >> 4;
    d.t = d.lane & 15;
    d.T = c.T;
    d.ok6 = LAST0 + d.t < d.T;
    d.tok6 = d.ok6 ? LAST0 + d.t : d.T - 1;
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        const unsigned in_row = (unsigned)(((d.g ^ (d.t & 3)) << 4) + ((m2 ^ (d.t >> 2)) << 6));
        const unsigned in_row6 = (unsigned)(((d.g ^ (d.tok6 & 3)) << 4) + ((m2 ^ ((d.tok6 >> 2) & 3)) << 6));
        d.xa[m2] = (unsigned)(d.t * XROW) + in_row;
        d.xa6[m2] = (unsigned)(d.tok6 * XROW) + in_row6;
    }
    return d;
}

static __device__ __forceinline__ bool tok_ok(const Ctx &c, int tt) { return tt < NTT - 1 || c.ok6; }
static __device__ __forceinline__ int tok_of(const Ctx &c, int tt) { return tt < NTT - 1 ? 16 * tt + c.t : c.tok6; }
// reader addresses: chunk (g, plane, k-step) of this lane's token in tile tt
static __device__ __forceinline__ unsigned x_at(const Ctx &c, int tt, int pl, int ks) {
    const int m2 = pl | ((ks & 1) << 1);
    return (tt < NTT - 1 ? c.xa[m2] + (unsigned)(tt * 16 * XROW) : c.xa6[m2]) + (unsigned)((ks >> 1) * 256);
}
static __device__ __forceinline__ unsigned q_at(const Ctx &c, int tt, int pl, int kk) {
    const int m2 = pl | (kk << 1);   // xa = row * 1024 + in_row: the row part shrinks to row * 256
    return tt < NTT - 1 ? c.xa[m2] - (unsigned)(c.t * (XROW - QROW)) + (unsigned)(tt * 16 * QROW) : c.xa6[m2] - (unsigned)(c.tok6 * (XROW - QROW));
}
static __device__ __forceinline__ unsigned p_at(const Ctx &c, int tt, int pl, int kk) {
    const int m2 = pl | ((kk & 1) << 1);
    return (tt < NTT - 1 ? c.xa[m2] - (unsigned)(c.t * (XROW - PROW)) + (unsigned)(tt * 16 * PROW) : c.xa6[m2] - (unsigned)(c.tok6 * (XROW - PROW))) +
           (unsigned)((kk >> 1) * 256);
}
// one-plane panel of the self-attention block (512-byte rows, chunk = g | ks << 2): the same in-row swizzle as the full panel
static __device__ __forceinline__ unsigned x1_at(const Ctx &c, int tt, int ks) {
    return (tt < NTT - 1 ? c.xa[ks & 3] - (unsigned)(c.t * (XROW - X1ROW)) + (unsigned)(tt * 16 * X1ROW) : c.xa6[ks & 3] - (unsigned)(c.tok6 * (XROW - X1ROW))) +
           (unsigned)((ks >> 2) * 256);
}
static __device__ __forceinline__ unsigned x1_off(int tok, int chunk) { return (unsigned)(tok * X1ROW + ((chunk ^ (tok & 15)) << 4)); }
// generic forms (writers: the chunk's low two bits are not the lane's g).  chunk = gk | plane << 2 | kstep << 3
static __device__ __forceinline__ unsigned x_off(int tok, int chunk) { return (unsigned)(tok * XROW + ((chunk ^ (tok & 15)) << 4)); }
static __device__ __forceinline__ unsigned q_off(int tok, int chunk) { return (unsigned)(tok * QROW + ((chunk ^ (tok & 15)) << 4)); }
static __device__ __forceinline__ unsigned p_off(int tok, int chunk) { return (unsigned)(tok * PROW + ((chunk ^ (tok & 15)) << 4)); }

static __device__ __forceinline__ f16x8 lds16(const char *p) { return *reinterpret_cast<const f16x8 *>(p); }

// H[0][tt], H[1][tt] (features 32 w + 4 g + r and 32 w + 16 + 4 g + r of this lane's token in tile tt) -> X panel, as split planes
// of value * ACT: slot g of k-step w (kperm), one 16-byte store per plane
static __device__ __forceinline__ void split_store8(char *hi_at, char *lo_at, const f32x4 &v0, const f32x4 &v1) {
    f16x4 h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    *reinterpret_cast<f16x8 *>(hi_at) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<f16x8 *>(lo_at) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}
static __device__ __forceinline__ void store_x(const Ctx &c, int tt, const f32x4 &v0, const f32x4 &v1) {
    if (!tok_ok(c, tt)) return;
    const int chunk = c.g | (c.w << 3), tok = tok_of(c, tt);
    char *X = c.smem + LDS_X;
    split_store8(X + x_off(tok, chunk), X + x_off(tok, chunk | 4), v0, v1);
}

// the same values as ONE fp16 plane of the one-plane panel (k-step w -> chunk bits 2..4)
static __device__ __forceinline__ void store_x1(const Ctx &c, int tt, const f32x4 &v0, const f32x4 &v1) {
    if (!tok_ok(c, tt)) return;
    const int chunk = c.g | (c.w << 2), tok = tok_of(c, tt);
    *reinterpret_cast<f16x8 *>(c.smem + LDS_X + x1_off(tok, chunk)) =
        __builtin_shufflevector(__builtin_convertvector(v0, f16x4), __builtin_convertvector(v1, f16x4), 0, 1, 2, 3, 4, 5, 6, 7);
}

// LayerNorm over the 256 features of H -> split planes of the X panel (scaled by ACT).  Per wave: mean and centred sum of
// squares of its 32 features (two-pass, in registers + two row all-reduces), then Chan's combination of the 8 waves' pairs:
// one exchange, two barriers (the first also fences the X panel's previous readers).
// HI_ONLY: the one-plane panel of the self-attention block (LayerNorm 1).
// a LayerNorm's affine parameters of this lane's features: requested first - their L2 round trip passes under the statistics (a
// workgroup is alone on its CU) - and, where a request to HBM is also due (the folded keys before LayerNorm 2), BEFORE it: loads
// return in order, a parameter requested after 16 KB of HBM reads would arrive behind them
// rows_out (training): the normalised rows also go to HBM (this trajectory's [T][256] block, unscaled); amax: running max |.| of them
template <bool HI_ONLY = false>
static __device__ __forceinline__ void layer_norm_to_x(const Ctx &c0, const f32x4 (&H)[2][NTT], const LnAffine &aff, float *rows_out = nullptr,
                                                       float *amax = nullptr) {
    const Ctx c = ctx_local(c0);
    float *stat = reinterpret_cast<float *>(c.smem + LDS_STAT);
    const f32x4 (&gwv)[2] = aff.w, (&gbv)[2] = aff.b;
    float mw[NTT], qw[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const f32x4 s4 = H[0][tt] + H[1][tt];
        mw[tt] = rows4_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / 32);
    }
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const f32x4 d0 = H[0][tt] - mw[tt], d1 = H[1][tt] - mw[tt];
        const f32x4 q4 = d0 * d0 + d1 * d1;
        qw[tt] = rows4_sum((q4[0] + q4[1]) + (q4[2] + q4[3]));
    }
    if (c.g == 0) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
            if (tok_ok(c, tt)) *reinterpret_cast<f32x2 *>(stat + (tok_of(c, tt) * 8 + c.w) * 2) = f32x2{mw[tt], qw[tt]};
    }
    TJ_STAMP(49);
    TJ_SYNC(0);
    TJ_STAMP(50);
    // Chan's combination of the 8 waves' pairs, once per (token, wave) instead of once per lane: lane (t, g) combines token tiles
    // g and g + 4 (tile 7 does not exist: those lanes repeat tile 6), publishes (mean, rstd) in a scratch row of its own wave -
    // LDS operations of one wave execute in order, no barrier - and every lane reads back the seven pairs of its token.
    // The scratch (8 waves x 16 tokens x 80 B in the K / V / P region) is free between this LayerNorm's two barriers: every reader
    // of that region has passed the first one.
    char *scr = c.smem + LDS_P + c.w * 1280 + c.t * 80;
#pragma unroll
    for (int j = 0; j < (NTT > 4 ? 2 : 1); ++j) {
        const int tl = c.g + 4 * j;
        const float *sp = stat + (tl < NTT - 1 ? 16 * tl + c.t : c.tok6) * 16;
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(sp), p1 = *reinterpret_cast<const f32x4 *>(sp + 4);
        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(sp + 8), p3 = *reinterpret_cast<const f32x4 *>(sp + 12);
        const float m = (((p0[0] + p0[2]) + (p1[0] + p1[2])) + ((p2[0] + p2[2]) + (p3[0] + p3[2]))) * 0.125f;
        const f32x4 e0 = f32x4{p0[0], p0[2], p1[0], p1[2]} - m, e1 = f32x4{p2[0], p2[2], p3[0], p3[2]} - m;
        const f32x4 ee = e0 * e0 + e1 * e1;
        const float m2 = (((p0[1] + p0[3]) + (p1[1] + p1[3])) + ((p2[1] + p2[3]) + (p3[1] + p3[3]))) + 32.0f * ((ee[0] + ee[1]) + (ee[2] + ee[3]));
        *reinterpret_cast<f32x2 *>(scr + 8 * tl) = f32x2{m, __builtin_amdgcn_rsqf(m2 * (1.0f / D) + SD_LN_EPS)};
    }
    __builtin_amdgcn_wave_barrier();
    float mean[NTT], rstd[NTT];
#pragma unroll
    for (int i = 0; i < (NTT + 1) / 2; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(scr + 16 * i);
        mean[2 * i] = v[0];
        rstd[2 * i] = v[1];
        if (2 * i + 1 < NTT) {
            mean[2 * i + 1] = v[2];
            rstd[2 * i + 1] = v[3];
        }
    }
    TJ_STAMP(51);
    const f32x4 gw0 = gwv[0] * ACT, gb0 = gbv[0] * ACT, gw1 = gwv[1] * ACT, gb1 = gbv[1] * ACT;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
        const f32x4 y0 = ((H[0][tt] - mean[tt]) * rstd[tt]) * gw0 + gb0, y1 = ((H[1][tt] - mean[tt]) * rstd[tt]) * gw1 + gb1;
        if constexpr (HI_ONLY) store_x1(c, tt, y0, y1);
        else store_x(c, tt, y0, y1);
        if (rows_out && tok_ok(c, tt)) {
            const f32x4 u0 = y0 * (1.0f / ACT), u1 = y1 * (1.0f / ACT);
            float *at = rows_out + (long)tok_of(c, tt) * D + 32 * c.w + 4 * c.g;
            SD_NT_STORE(u0, reinterpret_cast<f32x4 *>(at));
            SD_NT_STORE(u1, reinterpret_cast<f32x4 *>(at + 16));
            if (amax) {
                const f32x4 m4 = f32x4{fmaxf(fabsf(u0[0]), fabsf(u1[0])), fmaxf(fabsf(u0[1]), fabsf(u1[1])), fmaxf(fabsf(u0[2]), fabsf(u1[2])), fmaxf(fabsf(u0[3]), fabsf(u1[3]))};
                *amax = fmaxf(*amax, fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // one tile at a time: hipcc otherwise interleaves all 7 and spills
    }
    TJ_STAMP(52);
    TJ_SYNC(1);   // X complete
}
template <bool HI_ONLY = false>
static __device__ __forceinline__ void layer_norm_to_x(const Ctx &c0, const f32x4 (&H)[2][NTT], const float *ln_w, const float *ln_b, float *rows_out = nullptr,
                                                       float *amax = nullptr) {
    LnAffine aff;
    ln_affine_load(c0, ln_w, ln_b, aff);
    layer_norm_to_x<HI_ONLY>(c0, H, aff, rows_out, amax);
}

// K = 256 GEMM against the X panel as ONE software pipeline over 8 k-steps x 7 token tiles: the B fragment of the next tile
// (two planes, two ds_read_b128) is requested before the MFMAs of the current tile, the A fragments of the next k-step at the
// first tile of the current one; the order "issue next loads -> MFMAs" is pinned with sched_barrier (hipcc otherwise issues
// every load right before its use and waits for it).  A0 / A1: the two n-tiles of this wave (fragment streams
// [ks][plane][lane][8]); body(tt, a0h, a0l, a1h, a1l, bh, bl) issues the MFMAs of token tile tt.
template <class Body>
static __device__ __forceinline__ void gemm_pipe(const Ctx &c, const f16 *pa0, const f16 *pa1, Body body) {
    const char *X = c.smem + LDS_X;
    const unsigned lo = (unsigned)c.lane * 8;
    f16x8 a0[2][2], a1[2][2], b[2][2];   // A: [k-step parity][plane]; B: [stage parity][plane]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        a0[0][pl] = *reinterpret_cast<const f16x8 *>(pa0 + lo + pl * 512);
        a1[0][pl] = *reinterpret_cast<const f16x8 *>(pa1 + lo + pl * 512);
        b[0][pl] = lds16(X + x_at(c, 0, pl, 0));
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int cur = (ks * NTT + tt) & 1, nxt = cur ^ 1;
            if (tt == 0 && ks + 1 < 8) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    a0[(ks + 1) & 1][pl] = *reinterpret_cast<const f16x8 *>(pa0 + lo + ((ks + 1) * 2 + pl) * 512);
                    a1[(ks + 1) & 1][pl] = *reinterpret_cast<const f16x8 *>(pa1 + lo + ((ks + 1) * 2 + pl) * 512);
                }
            }
            const int nks = tt + 1 < NTT ? ks : ks + 1, ntt = tt + 1 < NTT ? tt + 1 : 0;
            if (nks < 8) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) b[nxt][pl] = lds16(X + x_at(c, ntt, pl, nks));
            }
            __builtin_amdgcn_sched_barrier(0);
            body(tt, a0[ks & 1][0], a0[ks & 1][1], a1[ks & 1][0], a1[ks & 1][1], b[cur][0], b[cur][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Q | K | V projection of one head.  acc0[tt] += A0 . X^T for all token tiles, acc1[i] += A1 . X^T for token tiles tt1 + i
// (tt1 = 4 for odd waves, else 0), K = 256, against the ONE-plane panel: two MFMAs per product (W_lo X_hi + W_hi X_hi).  The
// lo part of LayerNorm 1's output is dropped here and only here: measured over the 50-step rollout against the fp64 oracle
// 5.1e-6 (three products: 4.2e-7; bar 1e-4; dropping the WEIGHTS' lo part instead: 2.8e-5 - tools/exp/precision_sites.sh,
// NOTEBOOK.md 5.11).  Same software pipeline as gemm_pipe; the token half of A1 is a wave-uniform run-time predicate so that
// the code exists once per parity.
struct HeadAcc { f32x4 a0[NTT], a1[NH0]; };
template <bool odd>
static __device__ __forceinline__ void gemm_head(const Ctx &c, HeadAcc &acc, const f16 *pa0, const f16 *pa1) {
    const char *X = c.smem + LDS_X;
    const unsigned lo = (unsigned)c.lane * 8;
    f16x8 a0[2][2], a1[2][2], b[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        a0[0][pl] = *reinterpret_cast<const f16x8 *>(pa0 + lo + pl * 512);
        a1[0][pl] = *reinterpret_cast<const f16x8 *>(pa1 + lo + pl * 512);
    }
    b[0] = lds16(X + x1_at(c, 0, 0));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const int cur = (ks * NTT + tt) & 1, nxt = cur ^ 1;
            if (tt == 0 && ks + 1 < 8) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    a0[(ks + 1) & 1][pl] = *reinterpret_cast<const f16x8 *>(pa0 + lo + ((ks + 1) * 2 + pl) * 512);
                    a1[(ks + 1) & 1][pl] = *reinterpret_cast<const f16x8 *>(pa1 + lo + ((ks + 1) * 2 + pl) * 512);
                }
            }
            const int nks = tt + 1 < NTT ? ks : ks + 1, ntt = tt + 1 < NTT ? tt + 1 : 0;
            if (nks < 8) b[nxt] = lds16(X + x1_at(c, ntt, nks));
            __builtin_amdgcn_sched_barrier(0);
            acc.a0[tt] = mfma16(a0[ks & 1][1], b[cur], acc.a0[tt]);
            acc.a0[tt] = mfma16(a0[ks & 1][0], b[cur], acc.a0[tt]);
            if constexpr (!odd) {
                if (tt < NH0) {
                    acc.a1[tt] = mfma16(a1[ks & 1][1], b[cur], acc.a1[tt]);
                    acc.a1[tt] = mfma16(a1[ks & 1][0], b[cur], acc.a1[tt]);
                }
            } else if (tt >= NH0) {
                acc.a1[tt - NH0] = mfma16(a1[ks & 1][1], b[cur], acc.a1[tt - NH0]);
                acc.a1[tt - NH0] = mfma16(a1[ks & 1][0], b[cur], acc.a1[tt - NH0]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// acc[a][tt] += A_a . X^T, n-tiles 2 w + a of a 256 x 256 matrix in fragment-major planes
template <int SITE>
static __device__ __forceinline__ void gemm_x2(const Ctx &c, f32x4 (&acc)[2][NTT], const f16 *wmat) {
    const f16 *pa = wmat + (long)(2 * c.w) * (8 * 2 * 512);
    gemm_pipe(c, pa, pa + 8 * 2 * 512, [&](int tt, f16x8 a0h, f16x8 a0l, f16x8 a1h, f16x8 a1l, f16x8 bh, f16x8 bl) __attribute__((always_inline)) {
        mma3<SITE>(acc[0][tt], a0h, a0l, bh, bl);
        mma3<SITE>(acc[1][tt], a1h, a1l, bh, bl);
    });
}

// the A fragments of a K = 64 GEMM: n-tiles at pa0 / pa1 (streams at the first k-step), [n-tile][k-step][plane]
// acc[a][tt] += A_a . B^T with B rows of 64 features in a Q-layout buffer (K = 64)
static __device__ __forceinline__ void gemm_k64(const Ctx &c, f32x4 (&acc)[2][NTT], const AK64 &f, const char *Bbuf) {
    const f16x8 (&a)[2][2][2] = f.a;
    f16x8 b[3][2];
    ring_pipe<2 * NTT, 3>(
        [&](int s) __attribute__((always_inline)) {
            b[s % 3][0] = lds16(Bbuf + q_at(c, s >> 1, 0, s & 1));
            b[s % 3][1] = lds16(Bbuf + q_at(c, s >> 1, 1, s & 1));
        },
        [&](int s) __attribute__((always_inline)) {
            const int tt = s >> 1, kk = s & 1;
            mma3<S_OUT>(acc[0][tt], a[0][kk][0], a[0][kk][1], b[s % 3][0], b[s % 3][1]);
            mma3<S_OUT>(acc[1][tt], a[1][kk][0], a[1][kk][1], b[s % 3][0], b[s % 3][1]);
        });
}


// ---------------------------------------------------------------------------------------------------
// The self-attention block: H (residual accumulators, pre-scaled by ACT * s_o) += sum over heads of Wo[:, head] . O_head^T.
// Q, K, V and O of a head each have an LDS buffer of their own, so a head needs two barriers: after its Q | K | V are written and
// after its attention output is.  Between them two INDEPENDENT jobs run, in opposite order on the two waves of a SIMD (waves
// w and w + 4), so that one wave's MFMA stream overlaps the other's VALU work:
//   phase W:  [out-projection of head h-1 (MFMA)]      ||  [Q | K | V of head h: accumulators -> split planes in LDS (VALU)]
//   phase X:  [Q | K | V projection of head h+1 (MFMA)] ||  [attention of head h: scores, softmax, P V, O -> LDS (MFMA + VALU)]
// ---------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void head_zero(HeadAcc &acc) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) acc.a0[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NH0; ++i) acc.a1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
static __device__ __forceinline__ void head_gemm(const Ctx &c0, const SaW &a, int h, HeadAcc &acc) {
    const Ctx c = ctx_local(c0);
    const int w = c.w;
    const int nt0 = (w < 4 ? 0 : 16) + 4 * h + (w & 3);     // Q tile (waves 0..3) or K tile (waves 4..7)
    const int nt1 = 32 + 4 * h + (w >> 1);                   // V tile, token half w & 1
    head_zero(acc);
    // the token half of the V tile is a wave-uniform predicate: as a run-time condition inside the pipeline it costs a branch per step
    // (2 - 4 MFMAs), so the loop exists once per parity
    if (w & 1) gemm_head<true>(c, acc, a.w_in + (long)nt0 * (8 * 2 * 512), a.w_in + (long)nt1 * (8 * 2 * 512));
    else gemm_head<false>(c, acc, a.w_in + (long)nt0 * (8 * 2 * 512), a.w_in + (long)nt1 * (8 * 2 * 512));
}
// accumulators of head h -> Q or K tile and V piece as split planes
static __device__ __forceinline__ void head_write_qkv(const Ctx &c0, const SaW &a, int h, const HeadAcc &acc) {
    const Ctx c = ctx_local(c0);
    const int w = c.w, g = c.g, t = c.t, w3 = w & 3;
    const float c_in = 1.0f / a.s_in;   // accumulator -> ACT * value
    {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.b_in + (w < 4 ? 0 : D) + HD * h + 16 * w3 + 4 * g) * ACT;
        char *dst = c.smem + (w < 4 ? LDS_SQ : LDS_Q);
        const int chunk = (2 * (w3 & 1) + (g >> 1)) | ((w3 >> 1) << 3);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            if (!tok_ok(c, tt)) continue;
            const int tok = tok_of(c, tt);
            split_store(dst + q_off(tok, chunk) + 8 * (g & 1), dst + q_off(tok, chunk | 4) + 8 * (g & 1), acc.a0[tt] * c_in + bv);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    {   // V piece -> rows [token][hi 64 | lo 64] (features 16 (w >> 1) + 4 g + r)
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.b_in + 2 * D + HD * h + 16 * (w >> 1) + 4 * g) * ACT;
        const int tt1 = (w & 1) ? NH0 : 0, n1 = (w & 1) ? NTT - NH0 : NH0;
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            if (i >= n1) continue;
            const int tok = 16 * (tt1 + i) + t;
            if (tok >= c.T) continue;
            char *at = c.smem + LDS_K + tok * VROW + 2 * (16 * (w >> 1) + 4 * g);
            split_store(at, at + 128, acc.a1[i] * c_in + bv);
        }
    }
}
// attention of query tile w (waves 0 .. NTT-1) of the head whose Q, K (att_scores) and V (att_pv) are in LDS: O -> LDS.  Two
// pieces, so that the precise variant can put a barrier (K's buffer becomes V's) between them.
// att_scores: S^T = K Q^T, softmax numerators in S (times 2^10), their sum in psum
// lse2 (training): log2-sum-exp of the scaled scores of this lane's query, the value attention_bwd16_kernel recomputes the probabilities from
static __device__ __forceinline__ void att_scores(const Ctx &c, const SaW &a, const char *Qb, const char *Kb, f32x4 (&S)[NTT], float &psum,
                                                  float *lse2 = nullptr) {
    const int w = c.w, g = c.g, t = c.t;
    // ---- scores S^T[key][query] = K Q^T
    {
        f16x8 qf[2][2];
        const int qtok = min(16 * w + t, c.T - 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) qf[kk][pl] = lds16(Qb + q_off(qtok, g | (pl << 2) | (kk << 3)));
#pragma unroll
        for (int kt = 0; kt < NTT; ++kt) S[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16x8 kf[4][2];
        ring_pipe<2 * NTT, 4>(
            [&](int s) __attribute__((always_inline)) {
                kf[s & 3][0] = lds16(Kb + q_at(c, s >> 1, 0, s & 1));
                kf[s & 3][1] = lds16(Kb + q_at(c, s >> 1, 1, s & 1));
            },
            [&](int s) __attribute__((always_inline)) { mma3<S_SCORES>(S[s >> 1], kf[s & 3][0], kf[s & 3][1], qf[s & 1][0], qf[s & 1][1]); });
    }
    // ---- softmax over the keys of this lane's query: registers r of tile kt are keys 16 kt + 4 g + r (only tile 6 has masked keys)
    const float c_s = a.scale_log2e / (ACT * ACT);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (LAST0 + 4 * g + r >= c.T) S[NTT - 1][r] = -INFINITY;
    f32x4 m4 = S[0];
#pragma unroll
    for (int kt = 1; kt < NTT; ++kt) m4 = f32x4{fmaxf(m4[0], S[kt][0]), fmaxf(m4[1], S[kt][1]), fmaxf(m4[2], S[kt][2]), fmaxf(m4[3], S[kt][3])};
    const float m = rows4_max(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));
    // The Q | K | V projection read ONE fp16 plane of LayerNorm 1's output: the error that leaves in a logit grows with the logit.  The
    // path is validated up to |q.k| / sqrt(hd) = SD_SHARP_LOGIT_LIMIT; a larger top logit sets a status bit (never taken on the
    // validated range: one compare and a scalar branch per query tile) and the host repeats the rollout with three products.
    if constexpr (!PRECISE) {
        if (a.status && __builtin_amdgcn_ballot_w64(fabsf(m) * c_s > SD_SHARP_LOGIT_LIMIT * 1.44269504088896340736f) != 0) {
            if (c.lane == 0) atomicOr(a.status, SD_STATUS_SHARP_LOGITS);
        }
    }
    const float mb = m * c_s - 10.0f;   // probabilities carry 2^10 (fp16 lo parts stay normal)
    f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NTT; ++kt) {
        const f32x4 e = S[kt] * c_s - mb;
        S[kt] = f32x4{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1]), __builtin_amdgcn_exp2f(e[2]), __builtin_amdgcn_exp2f(e[3])};
        ps = ps + S[kt];
    }
    psum = rows4_sum((ps[0] + ps[1]) + (ps[2] + ps[3]));
    if (lse2) *lse2 = mb + __builtin_amdgcn_logf(psum);   // v_log_f32 is log2
}
// att_pv: O^T = V^T P^T with P^T straight from the score accumulators, V^T through transposing LDS reads; O / psum -> LDS planes
// rows_out (training): the attention output of this head also goes to HBM, [token][ld] at the head's 64 columns (unscaled); amax: running max |.| of it
static __device__ __forceinline__ void att_pv(const Ctx &c, const f32x4 (&S)[NTT], float psum, const char *Vb, char *Ob, float *rows_out = nullptr,
                                              int ld = 0, float *amax = nullptr) {
    const int w = c.w, g = c.g, t = c.t;
    f32x4 O[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) O[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q4 = t >> 2, p4 = t & 3;
    // 16 steps (key pair kp, feature tile ft) of three MFMAs; the V^T fragments (four transposing reads) come three steps ahead
    f16x8 ph[NKP], pl[NKP];
#pragma unroll
    for (int kp = 0; kp < NKP; ++kp) {
        // P fragment of keys 32 kp ..: elements 0..3 = tile 2 kp, 4..7 = tile 2 kp + 1 (beyond the last tile: zero)
        const f32x4 pa4 = S[2 * kp], pb4 = 2 * kp + 1 < NTT ? S[2 * kp + 1 < NTT ? 2 * kp + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
        f16x4 pah, pal, pbh, pbl;
        split4(pa4, pah, pal);
        split4(pb4, pbh, pbl);
        ph[kp] = __builtin_shufflevector(pah, pbh, 0, 1, 2, 3, 4, 5, 6, 7);
        pl[kp] = __builtin_shufflevector(pal, pbl, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    f16x8 vf[4][2];
    ring_pipe<4 * NKP, 4>(
        [&](int s) __attribute__((always_inline)) {
            const int kp = s >> 2, ft = s & 3;
            const int r0 = min(32 * kp + 4 * g + q4, c.T - 1), r1 = min(32 * kp + 16 + 4 * g + q4, c.T - 1);
            const char *v0 = Vb + r0 * VROW + 8 * p4, *v1 = Vb + r1 * VROW + 8 * p4;
#pragma unroll
            for (int pn = 0; pn < 2; ++pn) {
                const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(v0 + pn * 128 + ft * 32));
                const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(v1 + pn * 128 + ft * 32));
                vf[s & 3][pn] = __builtin_shufflevector(__builtin_bit_cast(f16x4, x0), __builtin_bit_cast(f16x4, x1), 0, 1, 2, 3, 4, 5, 6, 7);
            }
        },
        [&](int s) __attribute__((always_inline)) { mma3<S_PV>(O[s & 3], vf[s & 3][0], vf[s & 3][1], ph[s >> 2], pl[s >> 2]); });
    // O^T tile ft: features 16 ft + 4 g + r of query 16 w + t, times ACT / sum -> LDS planes
    const float inv = 1.0f / psum;
    const int tok = 16 * w + t;
    if (tok < c.T) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {   // features 32 kk + 4 g + r and 32 kk + 16 + 4 g + r: slot g of k-step kk (kperm)
            const int chunk = g | (kk << 3);
            split_store8(Ob + q_off(tok, chunk), Ob + q_off(tok, chunk | 4), O[2 * kk] * inv, O[2 * kk + 1] * inv);
        }
        if (rows_out) {
            const float inv1 = inv * (1.0f / ACT);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const f32x4 v = O[ft] * inv1;
                SD_NT_STORE(v, reinterpret_cast<f32x4 *>(rows_out + (long)tok * ld + 16 * ft + 4 * g));
                if (amax) *amax = fmaxf(*amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            }
        }
    }
}
static __device__ __forceinline__ void head_attention(const Ctx &c0, const SaW &a) {
    const Ctx c = ctx_local(c0);
    f32x4 S[NTT];
    float psum;
    att_scores(c, a, c.smem + LDS_SQ, c.smem + LDS_Q, S, psum);
    att_pv(c, S, psum, c.smem + LDS_K, c.smem + LDS_SO);
}
// out-projection of head h: its weight fragments (32 registers) are requested at the end of the head's phase X - their L2 round
// trip passes under the barrier and the other wave's job - and consumed in the next phase W
static __device__ __forceinline__ void head_out_load(const Ctx &c, const SaW &a, int h, AK64 &wo) {
    load_k64(c, wo, a.w_o + ((long)(2 * c.w) * 8 + 2 * h) * (2 * 512), a.w_o + ((long)(2 * c.w + 1) * 8 + 2 * h) * (2 * 512));
}
static __device__ __forceinline__ void head_out_proj(const Ctx &c0, const AK64 &wo, f32x4 (&H)[2][NTT]) {
    const Ctx c = ctx_local(c0);
    gemm_k64(c, H, wo, c.smem + LDS_SO);
}

// H = H * f + bias[feature]; the bias is requested (bias_load) before the GEMM whose result it completes
static __device__ __forceinline__ void unscale_h(f32x4 (&H)[2][NTT], float f, const Bias2 &b) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) H[a][tt] = H[a][tt] * f + b.v[a];
}

// ---------------------------------------------------------------------------------------------------
// PRECISE self-attention block: three fp16 products at the Q | K | V projection too (22-bit operands at every site: the result does
// not depend on how sharp the attention is - tests/test_gpu_denoiser.py::test_mode3_noise_prediction_*).  LayerNorm 1's output keeps
// both planes, i.e. the whole X panel, which leaves TWO buffers for the four images of a head: Q, later O, in LDS_Q; K, later V, in
// LDS_K.  Five barriers per head, every wave in the same phase (the fast variant's two-barrier, complementary-job structure needs
// the 51 KB that the second plane occupies).
// ---------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void sa_head_precise(const Ctx &c0, const SaW &a, int h, f32x4 (&H)[2][NTT]) {
    const Ctx c = ctx_local(c0);
    char *Qb = c.smem + LDS_Q, *Kb = c.smem + LDS_K;
    const int w = c.w, g = c.g, t = c.t;
    HeadAcc acc;
    head_zero(acc);
    if (h == 1) TJ_STAMP(17);   // (diagnostic build: the phases of head 1)
    {
        const int nt0 = (w < 4 ? 0 : 16) + 4 * h + (w & 3);     // Q tile (waves 0..3) or K tile (waves 4..7)
        const int nt1 = 32 + 4 * h + (w >> 1);                   // V tile, token half w & 1
        const f16 *pa0 = a.w_in + (long)nt0 * (8 * 2 * 512), *pa1 = a.w_in + (long)nt1 * (8 * 2 * 512);
        auto run = [&](auto odd_c) __attribute__((always_inline)) {
            constexpr bool OD = decltype(odd_c)::value;
            gemm_pipe(c, pa0, pa1, [&](int tt, f16x8 a0h, f16x8 a0l, f16x8 a1h, f16x8 a1l, f16x8 bh, f16x8 bl) __attribute__((always_inline)) {
                mma3<S_QKV>(acc.a0[tt], a0h, a0l, bh, bl);
                if constexpr (!OD) {
                    if (tt < NH0) mma3<S_QKV>(acc.a1[tt < NH0 ? tt : 0], a1h, a1l, bh, bl);
                } else {
                    if (tt >= NH0) mma3<S_QKV>(acc.a1[tt >= NH0 ? tt - NH0 : 0], a1h, a1l, bh, bl);
                }
            });
        };
        if (w & 1) run(std::true_type{});
        else run(std::false_type{});
    }
    const float c_in = 1.0f / a.s_in;   // accumulator -> ACT * value
    if (h == 1) TJ_STAMP(18);
    TJ_SYNC(2);            // B1: the previous head's readers of Q / O and K / V are done
    {   // Q or K tile -> LDS planes (features 16 (w & 3) + 4 g + r of the head, natural order on both operands of the scores)
        const int w3 = w & 3;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.b_in + (w < 4 ? 0 : D) + HD * h + 16 * w3 + 4 * g) * ACT;
        char *dst = w < 4 ? Qb : Kb;
        const int chunk = (2 * (w3 & 1) + (g >> 1)) | ((w3 >> 1) << 3);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            if (!tok_ok(c, tt)) continue;
            const int tok = tok_of(c, tt);
            split_store(dst + q_off(tok, chunk) + 8 * (g & 1), dst + q_off(tok, chunk | 4) + 8 * (g & 1), acc.a0[tt] * c_in + bv);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (h == 1) TJ_STAMP(19);
    TJ_SYNC(2);            // B2: Q, K complete
    f32x4 S[NTT];
    float psum = 1.f;
    if (w < NTT) att_scores(ctx_local(c0), a, Qb, Kb, S, psum);
    if (h == 1) TJ_STAMP(20);
    TJ_SYNC(3);            // B3: K is dead
    {   // V piece -> rows [token][hi 64 | lo 64] (features 16 (w >> 1) + 4 g + r) over K
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.b_in + 2 * D + HD * h + 16 * (w >> 1) + 4 * g) * ACT;
        const int tt1 = (w & 1) ? NH0 : 0, n1 = (w & 1) ? NTT - NH0 : NH0;
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            if (i >= n1) continue;
            const int tok = 16 * (tt1 + i) + t;
            if (tok >= c.T) continue;
            char *at = Kb + tok * VROW + 2 * (16 * (w >> 1) + 4 * g);
            split_store(at, at + 128, acc.a1[i] * c_in + bv);
        }
    }
    if (h == 1) TJ_STAMP(21);
    TJ_SYNC(3);            // B4: V complete (every wave has read its Q fragments: O may overwrite Q)
    AK64 wo;
    if (w < NTT) att_pv(ctx_local(c0), S, psum, Kb, Qb);
    // the out-projection's weight fragments: their L2 round trip passes under the barrier
    head_out_load(c, a, h, wo);
    if (h == 1) TJ_STAMP(22);
    TJ_SYNC(3);            // B5: O complete
    gemm_k64(ctx_local(c0), H, wo, Qb);
    if (h == 1) TJ_STAMP(23);
}
static __device__ __forceinline__ void sa_block_precise(const Ctx &c, const SaW &a, f32x4 (&H)[2][NTT], const float *b_o, Bias2 &bo) {
    bo = bias_load(c, b_o);
#pragma unroll 1
    for (int h = 0; h < NH; ++h) sa_head_precise(c, a, h, H);
}

// X (one plane) holds LayerNorm 1's output on entry; b_o: the out-projection bias, requested before the last head's projection
static __device__ __forceinline__ void sa_block(const Ctx &c, const SaW &a, f32x4 (&H)[2][NTT], const float *b_o, Bias2 &bo) {
    const bool first = c.w < 4;   // the quartet that runs the MFMA job of a phase first
    HeadAcc acc;
    AK64 wo;
    TJ_STAMP(3);
    head_gemm(c, a, 0, acc);
    TJ_STAMP(4);
#pragma unroll 1
    for (int h = 0; h < NH; ++h) {
        // phase W (the MFMA job exists twice in the code, before and after the VALU job: each quartet runs one copy)
        if (first && h > 0) head_out_proj(c, wo, H);
        TJ_STAMP(25 + h);
        head_write_qkv(c, a, h, acc);
        TJ_STAMP(41 + h);
        if (!first && h > 0) head_out_proj(c, wo, H);
        TJ_STAMP(45 + h);
        TJ_SYNC(2);                                // Q, K, V of head h complete (and every reader of head h-1's O is done)
        TJ_STAMP(5 + 3 * h);
        // phase X
        if (first && h + 1 < NH) head_gemm(c, a, h + 1, acc);
        TJ_STAMP(17 + h);
        if (c.w < NTT) head_attention(c, a);
        TJ_STAMP(21 + h);
        if (!first && h + 1 < NH) head_gemm(c, a, h + 1, acc);
        head_out_load(c, a, h, wo);
        TJ_STAMP(6 + 3 * h);
        TJ_SYNC(3);                                // O of head h complete; Q, K, V free
        TJ_STAMP(7 + 3 * h);
    }
    bo = bias_load(c, b_o);
    head_out_proj(c, wo, H);
}

static __device__ __forceinline__ void scale_h(f32x4 (&H)[2][NTT], float f) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) H[a][tt] = H[a][tt] * f;
}



// ---- folded cross-attention over 17 .. 64 memory rows (WIDE): the folded blocks of a trajectory are nkt = ceil(Mk / 16) key tiles
// laid out as nkt consecutive 16-slot blocks (block traj * nkt + kt of g16 / v16 / cb: the same fragment layouts as one tile); the
// step token is slot (Mk - 1) & 15 of the LAST tile.  Scores of all tiles stay in registers (4 x 4 accumulators), the softmax runs
// over them, then one round per tile: write P (the same 96-column panel), barrier, h += V'^T P^T.  Written for the robot's shapes
// (reference sim_scratch.yaml: 51 memory rows, B = 1): nothing is prefetched across phases, two barriers per extra tile.
static __device__ __forceinline__ void cross_wide(const Ctx &c0, const LayerW &L, f32x4 (&H)[2][NTT], long traj, int Mk, int nkt, float scale_log2e, long sblk) {
    const Ctx c = ctx_local(c0);
    const f16 *gstep = L.gstep + sblk * STEP_G_HALFS, *vstep = L.vstep + sblk * STEP_V_HALFS;
    constexpr int KT = 4;
    const int hh = c.w >> 1, Mc = Mk - 1, klast = nkt - 1, mcs = Mc & 15;
    const bool odd = c.w & 1;
    const int tt1 = odd ? NH0 : 0, n1 = odd ? NTT - NH0 : NH0;
    char *Pb = c.smem + LDS_P;
    const char *X = c.smem + LDS_X;
    f32x4 S[KT][NH0];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < NH0; ++i) S[kt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        if (kt >= nkt) break;
        // folded keys of (head, key tile): lanes of the step token's slot read the shared step block instead
        f16x8 gfr[8][2];
        const bool stepl = kt == klast && c.t == mcs;
        const f16 *gp = stepl ? gstep + (long)hh * (8 * 2 * 32) + c.g * 8 : L.g16 + ((traj * nkt + kt) * 4 + hh) * (8 * 2 * 512) + c.lane * 8;
        const int gstride = stepl ? 32 : 512;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) gfr[ks][pl] = *reinterpret_cast<const f16x8 *>(gp + (ks * 2 + pl) * gstride);
        auto scores = [&](auto odd_c) __attribute__((always_inline)) {
            constexpr bool OD = decltype(odd_c)::value;
            constexpr int N1 = OD ? NTT - NH0 : NH0, T1 = OD ? NH0 : 0;
            constexpr int N1D = N1 > 0 ? N1 : 1;
            f16x8 xb[3][2];
            ring_pipe<8 * N1, 3>(
                [&](int s) __attribute__((always_inline)) {
                    xb[s % 3][0] = lds16(X + x_at(c, T1 + s % N1D, 0, s / N1D));
                    xb[s % 3][1] = lds16(X + x_at(c, T1 + s % N1D, 1, s / N1D));
                },
                [&](int s) __attribute__((always_inline)) { mma3<S_XSC>(S[kt][s % N1D], gfr[s / N1D][0], gfr[s / N1D][1], xb[s % 3][0], xb[s % 3][1]); });
        };
        if (odd) scores(std::true_type{});
        else scores(std::false_type{});
    }
    // ---- softmax over all key slots of all tiles (accumulator rows 4 g + r of tile kt = memory row 16 kt + 4 g + r)
    const float c_g = 1.0f / (ACT * L.sc[4]);
    const float cs = L.cstep[sblk * 4 + hh];
    const StepScale ss = step_scale(L);
    f32x4 cbv[KT], cgl = f32x4{c_g, c_g, c_g, c_g};   // cgl: the last tile's multipliers (the step token's slot carries its own scale)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        cbv[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < nkt) cbv[kt] = *reinterpret_cast<const f32x4 *>(L.cb + (traj * nkt + kt) * 64 + hh * 16 + 4 * c.g);
        if (kt == klast) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * c.g + r == mcs) {
                    cbv[kt][r] = cs;
                    cgl[r] = ss.c_gs;
                }
        }
    }
#pragma unroll
    for (int i = 0; i < NH0; ++i) {
        if (i >= n1) continue;
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 v = S[kt][i] * (kt == klast ? cgl : f32x4{c_g, c_g, c_g, c_g}) + cbv[kt];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * kt + 4 * c.g + r >= Mk) v[r] = -INFINITY;
            S[kt][i] = v;
            m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        }
        m = rows4_max(m);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const f32x4 e = (S[kt][i] - m) * scale_log2e;
            const f32x4 p = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1]), __builtin_amdgcn_exp2f(e[2]), __builtin_amdgcn_exp2f(e[3])};
            S[kt][i] = p;
            sum += (p[0] + p[1]) + (p[2] + p[3]);
        }
        sum = rows4_sum(sum);
        const float f = PSC / sum;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) S[kt][i] = S[kt][i] * f;
    }
    // ---- h += sum over tiles of V'_kt^T P_kt^T + boc
    const float up = PSC * ss.s_v;
    const Bias2 boc = bias_load(c, L.b_oc);
    scale_h(H, up);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        if (kt >= nkt) break;
        TJ_SYNC(4);   // the previous readers of this region (out-projection / the previous tile's P V') are done: P may be written
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            if (i >= n1) continue;
            const f32x4 p = S[kt][i];
            const int tt = tt1 + i, tok = tt < NTT - 1 ? 16 * tt + c.t : c.tok6;
            if (tt == NTT - 1 && !c.ok6) continue;
            const int chunk = (2 * (hh & 1) + (c.g >> 1)) | ((hh >> 1) << 3);
            split_store(Pb + p_off(tok, chunk) + 8 * (c.g & 1), Pb + p_off(tok, chunk | 4) + 8 * (c.g & 1), p * ss.m_c);
            if (kt == klast && c.g == (mcs >> 2)) {   // the step token's probability again at k = 64 + 8 hh (its V' comes from the step block)
                const float pv = p[mcs & 3] * ss.m_s;
                const f16 ph = (f16)pv, pl = (f16)(pv - (float)ph);
                const f16x8 z8h = {ph, 0, 0, 0, 0, 0, 0, 0}, z8l = {pl, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<f16x8 *>(Pb + p_off(tok, hh | (2 << 3))) = z8h;
                *reinterpret_cast<f16x8 *>(Pb + p_off(tok, hh | 4 | (2 << 3))) = z8l;
            }
        }
        // folded values of this tile (and, for the last one, the step token's value columns)
        f16x8 av[2][3][2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f16 *vp = L.v16 + (((traj * nkt + kt) * 16 + 2 * c.w + n) * 2) * (2 * 512) + c.lane * 8;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) av[n][kk][pl] = *reinterpret_cast<const f16x8 *>(vp + (kk * 2 + pl) * 512);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const f16 sv = vstep[(pl * 4 + c.g) * D + 16 * (2 * c.w + n) + c.t];
                av[n][2][pl] = f16x8{sv, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        TJ_SYNC(5);   // P complete
        f16x8 pb[3][2];
        if (kt == klast) {
            ring_pipe<3 * NTT, 3>(
                [&](int s) __attribute__((always_inline)) {
                    pb[s % 3][0] = lds16(Pb + p_at(c, s / 3, 0, s % 3));
                    pb[s % 3][1] = lds16(Pb + p_at(c, s / 3, 1, s % 3));
                },
                [&](int s) __attribute__((always_inline)) {
                    const int tt = s / 3, kk = s % 3;
                    mma3<S_XPV>(H[0][tt], av[0][kk][0], av[0][kk][1], pb[s % 3][0], pb[s % 3][1]);
                    mma3<S_XPV>(H[1][tt], av[1][kk][0], av[1][kk][1], pb[s % 3][0], pb[s % 3][1]);
                });
        } else {
            ring_pipe<2 * NTT, 3>(
                [&](int s) __attribute__((always_inline)) {
                    pb[s % 3][0] = lds16(Pb + p_at(c, s / 2, 0, s % 2));
                    pb[s % 3][1] = lds16(Pb + p_at(c, s / 2, 1, s % 2));
                },
                [&](int s) __attribute__((always_inline)) {
                    const int tt = s / 2, kk = s % 2;
                    mma3<S_XPV>(H[0][tt], av[0][kk][0], av[0][kk][1], pb[s % 3][0], pb[s % 3][1]);
                    mma3<S_XPV>(H[1][tt], av[1][kk][0], av[1][kk][1], pb[s % 3][0], pb[s % 3][1]);
                });
        }
    }
    unscale_h(H, 1.0f / up, boc);
}

template <bool WIDE = false>
static __device__ __forceinline__ void decoder_layer(const Ctx &c0, const LayerW &L, f32x4 (&H)[2][NTT], long traj, int Mk, float scale_log2e, int *status,
                                                     int nkt = 1, long sblk = 0) {
    const f16 *gstep = L.gstep + sblk * STEP_G_HALFS, *vstep = L.vstep + sblk * STEP_V_HALFS;
    // ---- self-attention block: h += Wo . SA(LN1(h)) + bo
    {
        const Ctx &c = c0;
        const float s_o = L.sc[0], up = ACT * s_o;
        scale_h(H, up);
        Bias2 bo;
        const SaW sw{L.w_in, L.b_in, L.w_o, L.sc[3], scale_log2e, status};
        if constexpr (PRECISE) sa_block_precise(c, sw, H, L.b_o, bo);
        else sa_block(c, sw, H, L.b_o, bo);
        unscale_h(H, 1.0f / up, bo);
    }
    TJ_STAMP(31);
    // The folded keys of this wave's head (16 fragments = 64 registers) are requested BEFORE LayerNorm 2 and land under it: a
    // workgroup is alone on its CU, nothing else hides their HBM / MALL round trips (the score phase took 18 k cycles for 96 MFMAs).
    // Lanes of slot Mc read the step token's shared row instead of their trajectory's block.
    LnAffine aff2;
    ln_affine_load(c0, L.n2_w, L.n2_b, aff2);
    f16x8 gfr[8][2];
    if constexpr (!WIDE) {
        const int hh = c0.w >> 1, Mc = Mk - 1;
        const f16 *gp = c0.t == Mc ? gstep + (long)hh * (8 * 2 * 32) + c0.g * 8 : L.g16 + (traj * 4 + hh) * (8 * 2 * 512) + c0.lane * 8;
        const int gstride = c0.t == Mc ? 32 : 512;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) gfr[ks][pl] = *reinterpret_cast<const f16x8 *>(gp + (ks * 2 + pl) * gstride);
        __builtin_amdgcn_sched_barrier(0);
    }
    layer_norm_to_x(c0, H, aff2);
    TJ_STAMP(32);
    // ---- folded cross-attention: wave w scores head w >> 1 for token tiles tt1 .. (half w & 1)
    if constexpr (WIDE) {
        cross_wide(c0, L, H, traj, Mk, nkt, scale_log2e, sblk);
    } else {
        const Ctx c = ctx_local(c0);
        const int hh = c.w >> 1, Mc = Mk - 1;
        const bool odd = c.w & 1;
        const int tt1 = odd ? NH0 : 0, n1 = odd ? NTT - NH0 : NH0;
        char *Pb = c.smem + LDS_P;
        const char *X = c.smem + LDS_X;
        f32x4 S[NH0];
#pragma unroll
        for (int i = 0; i < NH0; ++i) S[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The score biases, the step token's value columns and the folded values of this trajectory (48 registers, HBM) are requested before
        // the score GEMM: an HBM round trip under load is longer than the softmax, and a load consumed before an older one has returned
        // waits for that one too (loads return in order).
        f32x4 cbv = *reinterpret_cast<const f32x4 *>(L.cb + traj * 64 + hh * 16 + 4 * c.g);
        const float cs = L.cstep[sblk * 4 + hh];
        __builtin_amdgcn_sched_barrier(0);
        f16x8 av[2][3][2];
        f16 svv[2][2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f16 *vp = L.v16 + ((traj * 16 + 2 * c.w + n) * 2) * (2 * 512) + c.lane * 8;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) av[n][kk][pl] = *reinterpret_cast<const f16x8 *>(vp + (kk * 2 + pl) * 512);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) svv[n][pl] = vstep[(pl * 4 + c.g) * D + 16 * (2 * c.w + n) + c.t];   // k = 64 + 8 g: head g
        }
        __builtin_amdgcn_sched_barrier(0);
        // (8 k-steps x 4 or 3 token tiles) steps of three MFMAs, the panel fragments two steps ahead; once per parity: the tile
        // numbers are compile-time
        auto scores = [&](auto odd_c) __attribute__((always_inline)) {
            constexpr bool OD = decltype(odd_c)::value;
            constexpr int N1 = OD ? NTT - NH0 : NH0, T1 = OD ? NH0 : 0;
            constexpr int N1D = N1 > 0 ? N1 : 1;   // (one token tile: the odd waves have none)
            f16x8 xb[3][2];
            ring_pipe<8 * N1, 3>(
                [&](int s) __attribute__((always_inline)) {
                    xb[s % 3][0] = lds16(X + x_at(c, T1 + s % N1D, 0, s / N1D));
                    xb[s % 3][1] = lds16(X + x_at(c, T1 + s % N1D, 1, s / N1D));
                },
                [&](int s) __attribute__((always_inline)) { mma3<S_XSC>(S[s % N1D], gfr[s / N1D][0], gfr[s / N1D][1], xb[s % 3][0], xb[s % 3][1]); });
        };
        if (odd) scores(std::true_type{});
        else scores(std::false_type{});
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) av[n][2][pl] = f16x8{svv[n][pl], 0, 0, 0, 0, 0, 0, 0};
        // softmax over the Mk key slots (accumulator rows 4 g + r) of each token (lane column)
        // the step token's slot carries the scales of the step blocks (sc[6], sc[7]): a score multiplier of its own, and both kinds of
        // probability are brought to the common value scale s_v = min(sc[5], sc[7]) (step_scale)
        const float c_g = 1.0f / (ACT * L.sc[4]);
        const StepScale ss = step_scale(L);
        f32x4 cg4 = {c_g, c_g, c_g, c_g};
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * c.g + r == Mc) {
                cbv[r] = cs;
                cg4[r] = ss.c_gs;
            }
        TJ_SYNC(4);   // the previous readers of Q / K (out-projection, PV) are done: P may be written
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            if (i >= n1) continue;
            f32x4 v = S[i] * cg4 + cbv;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * c.g + r >= Mk) v[r] = -INFINITY;
            const float m = rows4_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            const f32x4 e = (v - m) * scale_log2e;
            f32x4 p = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1]), __builtin_amdgcn_exp2f(e[2]), __builtin_amdgcn_exp2f(e[3])};
            const float sum = rows4_sum((p[0] + p[1]) + (p[2] + p[3]));
            const float f = PSC / sum;
            const float pstep = p[Mc & 3] * (f * ss.m_s);
            p = p * (f * ss.m_c);
            const int tt = tt1 + i, tok = tt < NTT - 1 ? 16 * tt + c.t : c.tok6;
            if (tt == NTT - 1 && !c.ok6) continue;
            // k = 16 hh + 4 g + r: k-step hh >> 1, 8-group 2 (hh & 1) + g / 2
            const int chunk = (2 * (hh & 1) + (c.g >> 1)) | ((hh >> 1) << 3);
            split_store(Pb + p_off(tok, chunk) + 8 * (c.g & 1), Pb + p_off(tok, chunk | 4) + 8 * (c.g & 1), p);
            // the step token's probability again at k = 64 + 8 hh (its V' comes from the shared step block): a whole chunk
            if (c.g == (Mc >> 2)) {
                const float pv = pstep;
                const f16 ph = (f16)pv, pl = (f16)(pv - (float)ph);
                const f16x8 z8h = {ph, 0, 0, 0, 0, 0, 0, 0}, z8l = {pl, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<f16x8 *>(Pb + p_off(tok, hh | (2 << 3))) = z8h;
                *reinterpret_cast<f16x8 *>(Pb + p_off(tok, hh | 4 | (2 << 3))) = z8l;
            }
        }
        TJ_SYNC(5);   // P complete
        TJ_STAMP(33);
        // h += V'^T P^T + boc: K = 64 (context slots of 4 heads) + 32 (step columns)
        const float up = PSC * ss.s_v;
        const Bias2 boc = bias_load(c, L.b_oc);
        scale_h(H, up);
        f16x8 pb[3][2];
        ring_pipe<3 * NTT, 3>(
            [&](int s) __attribute__((always_inline)) {
                pb[s % 3][0] = lds16(Pb + p_at(c, s / 3, 0, s % 3));
                pb[s % 3][1] = lds16(Pb + p_at(c, s / 3, 1, s % 3));
            },
            [&](int s) __attribute__((always_inline)) {
                const int tt = s / 3, kk = s % 3;
                mma3<S_XPV>(H[0][tt], av[0][kk][0], av[0][kk][1], pb[s % 3][0], pb[s % 3][1]);
                mma3<S_XPV>(H[1][tt], av[1][kk][0], av[1][kk][1], pb[s % 3][0], pb[s % 3][1]);
            });
        unscale_h(H, 1.0f / up, boc);
    }
    TJ_STAMP(34);
    layer_norm_to_x(c0, H, L.n3_w, L.n3_b);
    TJ_STAMP(35);
    // ---- feed-forward: h += W2 gelu(W1 LN3(h) + b1) + b2
    {
        const Ctx c = ctx_local(c0);
        f32x4 U[2][NTT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) U[a][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const Bias2 b1 = bias_load(c, L.b_1);
        gemm_x2<S_W1>(c, U, L.w_1);
        TJ_STAMP(36);
        const float c1 = 1.0f / (ACT * L.sc[1]);
        const Bias2 b2 = bias_load(c, L.b_2);   // lands under the GELU
        TJ_SYNC(6);   // every wave has read LN3(h): the panel receives gelu(u)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            f32x4 gl[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x4 pre = U[a][tt] * c1 + b1.v[a];
                const f32x2 g0 = gelu_erf_as2(f32x2{pre[0], pre[1]}) * ACT, g1 = gelu_erf_as2(f32x2{pre[2], pre[3]}) * ACT;
                gl[a] = f32x4{g0[0], g0[1], g1[0], g1[1]};
            }
            store_x(c, tt, gl[0], gl[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        TJ_SYNC(7);
        TJ_STAMP(37);
        const float up = ACT * L.sc[2];
        scale_h(H, up);
        gemm_x2<S_W2>(c, H, L.w_2);
        unscale_h(H, 1.0f / up, b2);
    }
    TJ_STAMP(38);
    if (L.nln_w) layer_norm_to_x<!PRECISE>(c0, H, L.nln_w, L.nln_b);
    TJ_STAMP(39);
}


static __device__ __forceinline__ void sa_body(const SaArgs &a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx c;
    ctx_init(c, smem, a.T);
    const long traj = blockIdx.x;
    TJ_STAMP(0);
    const float *hin = a.h_in + traj * HFRAG_FLOATS + (long)c.w * (2 * NTT * 256) + c.lane * 4;
    f32x4 H[2][NTT];
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) H[aa][tt] = *reinterpret_cast<const f32x4 *>(hin + (aa * NTT + tt) * 256);
    TJ_STAMP(1);
    layer_norm_to_x<true>(c, H, a.ln_w, a.ln_b);
    const float up = ACT * a.s_o;
    scale_h(H, up);
    TJ_STAMP(2);
    const SaW sw{a.w_in, a.b_in, a.w_o, a.s_in, a.scale_log2e, nullptr};
    Bias2 bo;
    sa_block(c, sw, H, a.b_o, bo);
    TJ_STAMP(31);
    unscale_h(H, 1.0f / up, bo);
    float *hout = a.h_out + traj * HFRAG_FLOATS + (long)c.w * (2 * NTT * 256) + c.lane * 4;
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) __builtin_nontemporal_store(H[aa][tt], reinterpret_cast<f32x4 *>(hout + (aa * NTT + tt) * 256));
    TJ_STAMP(32);
}

template <bool WIDE = false>
static __device__ __forceinline__ void step_body(const StepArgs &a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx c;
    ctx_init(c, smem, a.T);
#ifdef TJ_PRIO_YOUNG   // A/B: static priority for the second-dispatched quartet (it loses every MFMA arbitration to its SIMD partner)
    if (c.w >= 4) __builtin_amdgcn_s_setprio(TJ_PRIO_YOUNG);
#endif
#ifdef TJ_PRIO_OLD
    if (c.w < 4) __builtin_amdgcn_s_setprio(TJ_PRIO_OLD);
#endif
    const long traj = blockIdx.x;
    const int J = a.J;
    TJ_STAMP(0);
    f32x4 H[2][NTT];
    // ---- embedding: h^T = Wemb . x^T + b + pe^T.  x rows -> Q region as split planes (k = joint, zero-padded to 32)
    {
        char *Qb = c.smem + LDS_Q;
        // weights, bias and the positional rows of this wave's tokens are requested first (they land while x is staged)
        f16x8 we[2][2];
        f32x4 be[2], pe4[2][NTT];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f16 *wp = a.w_emb + (long)(2 * c.w + n) * (2 * 512) + c.lane * 8;
            we[n][0] = *reinterpret_cast<const f16x8 *>(wp);
            we[n][1] = *reinterpret_cast<const f16x8 *>(wp + 512);
            const int n0 = 32 * c.w + 16 * n + 4 * c.g;
            be[n] = *reinterpret_cast<const f32x4 *>(a.b_emb + n0);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) pe4[n][tt] = *reinterpret_cast<const f32x4 *>(a.pe + (long)tok_of(c, tt) * D + n0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float *xr = a.x + traj * (long)a.T * J;
        if (J & 3) {
            // any joint count <= 32 (the reference's database has 22: soccer_diffusion/dataset/models.py:222-247): one thread per (token,
            // 8-joint chunk), scalar loads (rows are not 16-byte aligned), joints >= J zero
            for (int i = threadIdx.x; i < a.T * 4; i += NTHREADS) {
                const int tok = i >> 2, chunk = i & 3;
                f16x8 h8, l8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = 8 * chunk + e;
                    const float v = j < J ? xr[tok * J + j] * XSC : 0.f;
                    h8[e] = (f16)v;
                    l8[e] = (f16)(v - (float)h8[e]);
                }
                *reinterpret_cast<f16x8 *>(Qb + q_off(tok, chunk)) = h8;
                *reinterpret_cast<f16x8 *>(Qb + q_off(tok, chunk | 4)) = l8;
            }
        }
        const int nvec = (J & 3) ? 0 : a.T * J / 4;
        for (int i = threadIdx.x; i < nvec; i += NTHREADS) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + 4 * i) * XSC;
            const int tok = (4 * i) / J, j0 = 4 * i - tok * J;     // 4 consecutive joints of one token
            const int chunk = j0 >> 3;
            split_store(Qb + q_off(tok, chunk) + 2 * (j0 & 7), Qb + q_off(tok, chunk | 4) + 2 * (j0 & 7), v);
            if (j0 + 4 >= J) {   // this thread also zeroes k = J .. 31 of its token (the weights there are zero; LDS garbage might be NaN)
                const f16x4 z4 = {0, 0, 0, 0};
                for (int k = J; k < 32; k += 4) {
                    *reinterpret_cast<f16x4 *>(Qb + q_off(tok, k >> 3) + 2 * (k & 7)) = z4;
                    *reinterpret_cast<f16x4 *>(Qb + q_off(tok, (k >> 3) | 4) + 2 * (k & 7)) = z4;
                }
            }
        }
        TJ_SYNC(8);
        const float c_e = 1.0f / (XSC * a.sc_io[0]);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f16x8 ah = we[n][0], al = we[n][1];
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                mma3<S_EMB>(acc, ah, al, lds16(Qb + q_at(c, tt, 0, 0)), lds16(Qb + q_at(c, tt, 1, 0)));
                H[n][tt] = acc * c_e + (be[n] + pe4[n][tt]);
            }
        }
    }
    TJ_STAMP(1);
    layer_norm_to_x<!PRECISE>(c, H, a.n1_w, a.n1_b);
    TJ_STAMP(2);
#pragma unroll 1
    for (int l = 0; l < a.L; ++l) decoder_layer<WIDE>(c, a.layer[l], H, traj, a.Mk, a.scale_log2e, a.status, a.nkt, a.step_per_traj ? (a.step_map ? (long)a.step_map[traj] : traj) : 0L);
    // ---- fc_out + DDIM: eps^T = Wout . h^T + b.  h has no a-priori bound: one power-of-two scale per token
    {
        float *stat = reinterpret_cast<float *>(c.smem + LDS_STAT);
        float am[NTT];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const f32x4 a0 = H[0][tt], a1 = H[1][tt];
            float m = fmaxf(fmaxf(fabsf(a0[0]), fabsf(a0[1])), fmaxf(fabsf(a0[2]), fabsf(a0[3])));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a1[0]), fabsf(a1[1])), fmaxf(fabsf(a1[2]), fabsf(a1[3]))));
            am[tt] = rows4_max(m);
        }
        if (c.g == 0) {
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt)
                if (tok_ok(c, tt)) stat[tok_of(c, tt) * 8 + c.w] = am[tt];
        }
        TJ_SYNC(9);   // also: the X panel's readers (last W2 GEMM) are done
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const float *sp = stat + tok_of(c, tt) * 8;
            const f32x4 p0 = *reinterpret_cast<const f32x4 *>(sp), p1 = *reinterpret_cast<const f32x4 *>(sp + 4);
            const float m = fmaxf(fmaxf(fmaxf(p0[0], p0[1]), fmaxf(p0[2], p0[3])), fmaxf(fmaxf(p1[0], p1[1]), fmaxf(p1[2], p1[3])));
            const float s = f16_scale_from_bits(__builtin_bit_cast(unsigned, m));
            store_x(c, tt, H[0][tt] * s, H[1][tt] * s);
        }
        // all 32 weight fragments of fc_out (the residual registers are free now) are requested before the barrier
        f16x8 wf[2][8][2];
        if (c.w < NTT) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wf[n][ks][pl] = *reinterpret_cast<const f16x8 *>(a.w_out + ((long)(n * 8 + ks) * 2 + pl) * 512 + c.lane * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        TJ_SYNC(10);
        if (c.w < NTT) {
            const char *X = c.smem + LDS_X;
            f32x4 E[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const int tok = c.w < NTT - 1 ? 16 * c.w + c.t : c.tok6;
            const bool ok = c.w < NTT - 1 || c.ok6;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int m2lo = (ks & 1) << 1;
                const unsigned b0 = (c.w < NTT - 1 ? c.xa[m2lo] + (unsigned)(c.w * 16 * XROW) : c.xa6[m2lo]) + (unsigned)((ks >> 1) * 256);
                const unsigned b1 = (c.w < NTT - 1 ? c.xa[m2lo | 1] + (unsigned)(c.w * 16 * XROW) : c.xa6[m2lo | 1]) + (unsigned)((ks >> 1) * 256);
                const f16x8 bh = lds16(X + b0), bl = lds16(X + b1);
#pragma unroll
                for (int n = 0; n < 2; ++n) mma3<S_FC>(E[n], wf[n][ks][0], wf[n][ks][1], bh, bl);
            }
            // this lane's token scale again (the statistics are still in LDS)
            const float *sp = stat + tok * 8;
            const f32x4 p0 = *reinterpret_cast<const f32x4 *>(sp), p1 = *reinterpret_cast<const f32x4 *>(sp + 4);
            const float m = fmaxf(fmaxf(fmaxf(p0[0], p0[1]), fmaxf(p0[2], p0[3])), fmaxf(fmaxf(p1[0], p1[1]), fmaxf(p1[2], p1[3])));
            const float c_o = 1.0f / (f16_scale_from_bits(__builtin_bit_cast(unsigned, m)) * a.sc_io[1]);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int j0 = 16 * n + 4 * c.g;
                if (!ok || j0 >= J) continue;
                const long at = (traj * a.T + tok) * J + j0;
                if (J & 3) {   // rows of J floats are not 16-byte aligned, the last group of four is ragged: element by element
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (j0 + r >= J) continue;
                        const float e = E[n][r] * c_o + a.b_out[j0 + r];
                        if (a.eps_out) a.eps_out[at + r] = e;
                        if (a.update_x) {
                            const float x0 = (a.x[at + r] - a.c1 * e) / a.c0;
                            a.x[at + r] = a.c2 * x0 + a.c3 * e;
                        }
                    }
                    continue;
                }
                const f32x4 e = E[n] * c_o + *reinterpret_cast<const f32x4 *>(a.b_out + j0);
                if (a.eps_out) *reinterpret_cast<f32x4 *>(a.eps_out + at) = e;
                if (a.update_x) {
                    const f32x4 xv = *reinterpret_cast<const f32x4 *>(a.x + at);
                    f32x4 xn;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {   // the oracle's fp32 op order (oracle/ddim_ref.py, as fc_out_kernel)
                        const float x0 = (xv[r] - a.c1 * e[r]) / a.c0;
                        xn[r] = a.c2 * x0 + a.c3 * e[r];
                    }
                    *reinterpret_cast<f32x4 *>(a.x + at) = xn;
                }
            }
        }
    }
    TJ_STAMP(40);
}
};   // struct TJ

static __global__ __launch_bounds__(NTHREADS, 2) void traj_sa_kernel(SaArgs a) { TJ<NTT_A, false>::sa_body(a); }

// sampler mode 3: NTT token tiles; PRECISE = three fp16 products at the Q | K | V site too
template <int NTT, bool PRECISE>
__global__ __launch_bounds__(NTHREADS, 2) void traj_step_kernel(StepArgs a) { TJ<NTT, PRECISE>::step_body(a); }
// ... with 17 .. 64 memory rows (2 .. 4 key tiles in the folded cross-attention: cross_wide); three products everywhere
template <int NTT>
__global__ __launch_bounds__(NTHREADS, 2) void traj_step_wide_kernel(StepArgs a) { TJ<NTT, true>::template step_body<true>(a); }


}   // namespace tj
