// The GENERIC trajectory-owning step kernels of the sampler: hidden_dim 128 / 256 / 512, any number of memory rows.
//
// Same idea as sd_traj.h (ONE workgroup of 8 waves carries ONE trajectory through a whole denoiser step: embedding, every decoder
// layer with its self-attention, cross-attention and feed-forward - reference: nn.TransformerDecoderLayer as built by
// soccer_diffusion/ml/model/decoder.py:26-35, norm_first; forward of decoder.py:38-54 - fc_out and the DDIM update; the residual
// stream lives in registers, q | k | v and the attention outputs never leave the CU), same numerics (every product is three fp16
// MFMAs on hi / lo operand pairs with fp32 accumulation: sampler mode 3), but written for the shapes sd_traj.h's hand-scheduled
// hidden_dim-256 / <= 64-memory-rows kernels do not take - the reference's own configs (ml/training/config/default.yaml: hidden_dim
// 128, 312 memory rows; larger_model.yaml: hidden_dim 512, 8 layers, 312 memory rows) and any hidden_dim-256 model with more than 64
// memory rows.  Differences:
//   * geometry is a template parameter: wave w owns output features [16 NA w, 16 NA (w + 1)), NA = D / 128 n-tiles, of all NTT token
//     tiles; the contraction index keeps its natural order in every panel (no lane-pair permutation: 8-byte LDS stores);
//   * the cross-attention is NOT folded: Q_c = Wq LN2(h) is a full row GEMM whose result replaces LN2(h) in the panel, then every
//     (head, query tile) unit streams the trajectory's projected memory K / V (split planes in HBM, packed once per context in MFMA
//     fragment order) in pairs of 32 keys with an online softmax, writes its normalised output over its own Q block, and the
//     out-projection is a second full row GEMM.  Memory length is a run-time number (the step token is one more "pair" with a single key);
//   * every wave is in the same phase (five barriers per self-attention head, as sd_traj.h's PRECISE variant).
// hidden_dim 512: the panel of a 100-token trajectory (200 KB as split planes) does not fit the CU's 160 KB of LDS: T <= 48 there
// (every shipped config has trajectory_prediction_length 10); longer horizons at 512 stay on the row-panel kernels (sampler mode 2).
#include "sd_common.h"
#include "sd_trajg.h"
#include "../../include/soccerdiffusion_hip.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace tg {

constexpr int NTHREADS = 512, MAX_L = 8;
constexpr float ACT = 8.0f;      // scale of LayerNorm outputs, q, k, v, attention / GELU outputs (as sd_traj.h)
constexpr float PSC = 1024.0f;   // scale of probabilities
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// c += (ah + al) (bh + bl) without lo.lo, small terms first
__device__ __forceinline__ void mma3(f32x4 &c, f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl) {
    c = mfma16(al, bh, c);
    c = mfma16(ah, bl, c);
    c = mfma16(ah, bh, c);
}
__device__ __forceinline__ float rows4_sum(float v) {   // all-reduce over lanes t, t + 16, t + 32, t + 48 (sd_traj.h)
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
__device__ __forceinline__ f16x8 lds16(const char *p) { return *reinterpret_cast<const f16x8 *>(p); }
__device__ __forceinline__ f16x8 glb16(const f16 *p) { return *reinterpret_cast<const f16x8 *>(p); }
__device__ __forceinline__ f16x8 zero8() { return f16x8{0, 0, 0, 0, 0, 0, 0, 0}; }

// ---------------------------------------------------------------------------------------------------
// once-per-call packing
// ---------------------------------------------------------------------------------------------------
// scale slots of a layer's 16-float row
enum { SC_IN = 0, SC_O = 1, SC_Q = 2, SC_OC = 3, SC_1 = 4, SC_2 = 5, SC_K = 6, SC_V = 7, SC_KS = 8, SC_VS = 9, SC_EMB = 10, SC_OUT = 11 };

__global__ void absmax_kernel(const float *__restrict__ x, long n, unsigned *word) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f && __builtin_bit_cast(unsigned, m) > *word) atomicMax(word, __builtin_bit_cast(unsigned, m));
}
// W (N x K row-major fp32, zero-padded to Np x Kp, Np % 16 == 0, Kp % 32 == 0) -> [n-tile][k-step][plane][lane][8]: lane = 16 g + i holds
// W[16 nt + i][32 ks + 8 g + 0..7] * scale as hi / lo (natural k order)
__global__ void pack_w_kernel(const float *__restrict__ W, int N, int K, int Np, int Kp, const unsigned *maxbits, f16 *__restrict__ dst,
                              float *scale_out) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const int nks = Kp / 32;
    const long total = (long)Np * (Kp / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / (Kp / 8)), k8 = (int)(i % (Kp / 8));
        f16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k8 * 8 + e;
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
// Projected memory rows kv [item][rows_per_item][2 D] (K in the first D columns, V in the last) -> the K planes of the cross-attention
// scores: per item [head 4][pair nkp][tile 2][kk D/128][plane 2][lane 64][8], lane = 16 g + i: key 32 pair + 16 tile + i, feature
// head D/4 + 32 kk + 8 g + e.  Keys >= rows_per_item are zero.
__global__ void pack_k_kernel(const float *__restrict__ kv, long items, int rows_per_item, int nkp, int D, const unsigned *maxbits,
                              f16 *__restrict__ dst, float *scale_out) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const int KH = D / 128;
    const long per_item = 4L * nkp * 2 * KH * 64, total = items * per_item;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long item = i / per_item;
        long j = i - item * per_item;
        const int lane = (int)(j & 63); j >>= 6;
        const int kk = (int)(j % KH); j /= KH;
        const int tile = (int)(j & 1); j >>= 1;
        const int pair = (int)(j % nkp);
        const int head = (int)(j / nkp);
        const int key = 32 * pair + 16 * tile + (lane & 15), f0 = head * (D / 4) + 32 * kk + 8 * (lane >> 4);
        f16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = key < rows_per_item ? kv[((long)item * rows_per_item + key) * 2 * D + f0 + e] * scale : 0.f;
            hh[e] = (f16)v;
            ll[e] = (f16)(v - (float)hh[e]);
        }
        f16 *o = dst + ((i >> 6) * 2) * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = hh[e];
            o[512 + e] = ll[e];
        }
    }
}
// ... -> the V^T planes of P V: per item [head 4][pair nkp][ft D/64][plane 2][lane 64][8], lane = 16 g + i: feature head D/4 + 16 ft + i,
// element e = key 32 pair + 16 (e >> 2) + 4 g + (e & 3) (the order in which a score accumulator pair holds its keys)
__global__ void pack_vt_kernel(const float *__restrict__ kv, long items, int rows_per_item, int nkp, int D, const unsigned *maxbits,
                               f16 *__restrict__ dst, float *scale_out) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const int NQ = D / 64;
    const long per_item = 4L * nkp * NQ * 64, total = items * per_item;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long item = i / per_item;
        long j = i - item * per_item;
        const int lane = (int)(j & 63); j >>= 6;
        const int ft = (int)(j % NQ); j /= NQ;
        const int pair = (int)(j % nkp);
        const int head = (int)(j / nkp);
        const int g = lane >> 4, f = head * (D / 4) + 16 * ft + (lane & 15);
        f16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = 32 * pair + 16 * (e >> 2) + 4 * g + (e & 3);
            const float v = key < rows_per_item ? kv[((long)item * rows_per_item + key) * 2 * D + D + f] * scale : 0.f;
            hh[e] = (f16)v;
            ll[e] = (f16)(v - (float)hh[e]);
        }
        f16 *o = dst + ((i >> 6) * 2) * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = hh[e];
            o[512 + e] = ll[e];
        }
    }
}
// The step tokens, all layers in one launch (a forward_with_context call of the reference's loop pays this once per call).  grid (n_tok, L,
// 2 D / 64), 256 threads: rows 64 z .. 64 z + 63 of K | V = Wkv tok + bkv -> kvstep [l][tok][2 D] (fp32), abs-max of the K and of the V columns
// -> words SC_KS / SC_VS of row l.  A wave has its 16 rows' weight loads in flight together (one workgroup per (token, layer) walking the rows one
// by one was a chain of 2 D / 4 dependent round trips: ~ 180 us at hidden_dim 512 for ONE token).
struct StepKvArgs { const float *wkv[MAX_L], *bkv[MAX_L]; };
template <int D>
__global__ __launch_bounds__(256) void step_kv_all_kernel(StepKvArgs a, const float *__restrict__ tokens, float *__restrict__ kvstep, long layer_stride,
                                                          unsigned *maxbits, const int *__restrict__ map) {
    constexpr int NK = (D + 255) / 256;   // float4 pieces of a row per lane
    const int l = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long tok = blockIdx.x;
    if (map && map[tok] != (int)tok) return;   // a duplicate of token 0 (step_map_kernel of sd_kernels.hip): nobody reads its rows
    f32x4 t4[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) t4[k] = 4 * lane + 256 * k < D ? *reinterpret_cast<const f32x4 *>(tokens + tok * D + 4 * lane + 256 * k) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float *wkv = a.wkv[l], *bkv = a.bkv[l];
    float *out = kvstep + l * layer_stride + tok * 2 * D;
    const int o0 = blockIdx.z * 64 + wv * 16;
    f32x4 w4[16][NK];
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int k = 0; k < NK; ++k)
            w4[u][k] = 4 * lane + 256 * k < D ? *reinterpret_cast<const f32x4 *>(wkv + (long)(o0 + u) * D + 4 * lane + 256 * k) : f32x4{0.f, 0.f, 0.f, 0.f};
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        float p = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) p += (w4[u][k][0] * t4[k][0] + w4[u][k][1] * t4[k][1]) + (w4[u][k][2] * t4[k][2] + w4[u][k][3] * t4[k][3]);
        const float v = wave_sum(p) + bkv[o0 + u];
        if (lane == 0) out[o0 + u] = v;
        m = fmaxf(m, fabsf(v));
    }
    if (lane == 0) {   // (a workgroup's 64 rows lie on one side of D: 64 divides 128, 256 and 512)
        unsigned *word = maxbits + l * 16 + (o0 < D ? SC_KS : SC_VS);
        const unsigned b = __builtin_bit_cast(unsigned, m);
        if (b > __atomic_load_n(word, __ATOMIC_RELAXED)) atomicMax(word, b);
    }
}
// grid (blocks, L): kvstep [l][n_tok][2 D] -> compact split rows [l][n_tok][4][D]: K hi, K lo, V hi, V lo; scales from words SC_KS / SC_VS ->
// the layer's scale row (with no_ctx also into SC_K / SC_V: no context rows - their unused planes take the step rows' scales, a scale of 1
// would drag the common value scale down)
__global__ void pack_step_all_kernel(const float *__restrict__ kvstep, long src_layer_stride, long n_tok, int D, const unsigned *maxbits,
                                     f16 *__restrict__ dst, long dst_layer_stride, float *scales, int no_ctx, const int *__restrict__ map) {
    const int l = blockIdx.y;
    const float sk = f16_scale_from_bits(maxbits[l * 16 + SC_KS]), sv = f16_scale_from_bits(maxbits[l * 16 + SC_VS]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scales[l * 16 + SC_KS] = sk;
        scales[l * 16 + SC_VS] = sv;
        if (no_ctx) {
            scales[l * 16 + SC_K] = sk;
            scales[l * 16 + SC_V] = sv;
        }
    }
    const float *src = kvstep + l * src_layer_stride;
    f16 *d = dst + l * dst_layer_stride;
    const long total = n_tok * 2 * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long tok = i / (2 * D);
        if (map && map[tok] != (int)tok) continue;
        const int c = (int)(i - tok * 2 * D), isv = c >= D, f = isv ? c - D : c;
        const float v = src[i] * (isv ? sv : sk);
        const f16 h = (f16)v;
        d[(tok * 4 + 2 * isv) * D + f] = h;
        d[(tok * 4 + 2 * isv + 1) * D + f] = (f16)(v - (float)h);
    }
}
// abs-max of the K columns and of the V columns of kv [rows][2 D], separately
__global__ void absmax_kv_kernel(const float *__restrict__ kv, long rows, int D, unsigned *wordK, unsigned *wordV) {
    float mk = 0.f, mv = 0.f;
    const long total = rows * 2 * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float a = fabsf(kv[i]);
        if ((int)(i % (2 * D)) < D) mk = fmaxf(mk, a);
        else mv = fmaxf(mv, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mk = fmaxf(mk, __shfl_xor(mk, o, 64));
        mv = fmaxf(mv, __shfl_xor(mv, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (mk > 0.f && __builtin_bit_cast(unsigned, mk) > *wordK) atomicMax(wordK, __builtin_bit_cast(unsigned, mk));
        if (mv > 0.f && __builtin_bit_cast(unsigned, mv) > *wordV) atomicMax(wordV, __builtin_bit_cast(unsigned, mv));
    }
}
__global__ void zero_words_kernel(unsigned *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
// zeroes words [col0, col0 + ncols) of every 16-word row
__global__ void zero_word_cols16_kernel(unsigned *mb, int rows, int col0, int ncols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * ncols) mb[(i / ncols) * 16 + col0 + i % ncols] = 0u;
}
// ---------------------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------------------
struct LayerW {
    const float *n1_w, *n1_b, *n2_w, *n2_b, *n3_w, *n3_b;
    const f16 *w_in, *w_o, *w_q, *w_oc, *w_1, *w_2;      // fragment-major planes
    const float *b_in, *b_o, *b_q, *b_oc, *b_1, *b_2;
    const float *sc;                                     // this layer's scale row
    const f16 *kp, *vp;                                  // memory K / V^T planes of this layer, all trajectories
    const f16 *kvs;                                      // step token rows [n_tok][4][D] of this layer (block 0 / this step)
};
struct StepArgs {
    float *x;                      // [B][T][J] in / out
    float *eps_out;                // [B][T][J] or NULL
    const f16 *w_emb;              // [D / 16 n-tiles][1][2][64][8] (K = J padded to 32)
    const float *b_emb, *pe;
    const f16 *w_out;              // [2 n-tiles][D / 32][2][64][8] (rows >= J zero)
    const float *b_out;
    const float *sc_io;            // scale row L: SC_EMB, SC_OUT
    float c0, c1, c2, c3;
    float scale_log2e;
    const int *step_map;           // step_per_traj: trajectory b reads step rows step_map[b], or NULL: rows b
    int T, B, J, L, Mc, nkp, update_x, step_per_traj;   // nkp: pairs of 32 context rows, ceil(Mc / 32)
    long kv_traj_halfs;            // halfs of one trajectory's K (= V^T) planes in a layer: nkp * 32 keys * D features * 2 planes
    LayerW layer[MAX_L];
};

// ---------------------------------------------------------------------------------------------------
// device code
// ---------------------------------------------------------------------------------------------------
template <int D, int NTT>
struct TG {
    static constexpr int HD = D / 4, NA = D / 128, KS = D / 32, NQ = HD / 16, KH = HD / 32;
    static constexpr int XROW = 4 * D, QROW = 4 * HD, VROW = 4 * HD + 32, EROW = 128;
    static constexpr int XCH = D / 8;                     // 16-byte chunks per plane of a panel row
    static constexpr int QCH = HD / 8;                    // ... of a Q / K / O row
    static constexpr int QMASK = 2 * QCH - 1 < 15 ? 2 * QCH - 1 : 15;
    static constexpr int LAST0 = 16 * (NTT - 1);
    static constexpr int NJOB = (3 * NQ + 7) / 8;         // Q | K | V n-tiles of a head per wave
    static constexpr int NUNIT = (4 * NTT + 7) / 8;       // (head, query tile) units of the cross-attention per wave

    struct Ctx {
        char *smem;
        int lane, w, g, t, T;
        int tokl;     // this lane's token of the last tile, clamped to T - 1
        bool okl;     // ... and whether it exists
        unsigned oQ, oK, oS;   // LDS offsets of the Q / O buffer, the K / V buffer, the statistics
    };
    static __device__ __forceinline__ void ctx_init(Ctx &c, char *smem, int T) {
        c.smem = smem;
        c.lane = threadIdx.x & 63;
        c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        c.g = c.lane >> 4;
        c.t = c.lane & 15;
        c.T = T;
        c.okl = LAST0 + c.t < T;
        c.tokl = c.okl ? LAST0 + c.t : T - 1;
        c.oQ = (unsigned)(T * XROW);
        c.oK = c.oQ + (unsigned)(T * QROW);
        c.oS = c.oK + (unsigned)(T * VROW);
    }
    // A copy of the context whose per-lane values the compiler must treat as new (sd_traj.h: ctx_local): every LDS address derived from
    // them is then computed inside the phase that uses it.  Without this hipcc hoists the loop-invariant addresses of ALL phases out of the
    // head and layer loops, keeps hundreds of them live and spills as many registers.
    static __device__ __forceinline__ Ctx ctx_local(const Ctx &c) {
        Ctx d = c;
        asm volatile("" : "+v"(d.lane));
        d.g = d.lane >> 4;
        d.t = d.lane & 15;
        d.okl = LAST0 + d.t < d.T;
        d.tokl = d.okl ? LAST0 + d.t : d.T - 1;
        return d;
    }
    static __device__ __forceinline__ bool tok_ok(const Ctx &c, int tt) { return tt < NTT - 1 || c.okl; }
    static __device__ __forceinline__ int tok_of(const Ctx &c, int tt) { return tt < NTT - 1 ? 16 * tt + c.t : c.tokl; }
    // panel: row tok, 16-byte chunk (plane * XCH + k / 8), XOR-swizzled by the token (reads of 16 lanes = 16 tokens hit 16 different chunks)
    static __device__ __forceinline__ unsigned x_off(int tok, int chunk) { return (unsigned)(tok * XROW + ((chunk ^ (tok & 15)) << 4)); }
    static __device__ __forceinline__ unsigned q_off(int tok, int chunk) { return (unsigned)(tok * QROW + ((chunk ^ (tok & QMASK)) << 4)); }
    static __device__ __forceinline__ unsigned e_off(int tok, int chunk) { return (unsigned)(tok * EROW + ((chunk ^ (tok & 7)) << 4)); }

    // 16 features of n-tile nt of this lane's token -> panel planes (value already scaled)
    static __device__ __forceinline__ void store_x(const Ctx &c, int tt, int nt, const f32x4 &v) {
        if (!tok_ok(c, tt)) return;
        const int tok = tok_of(c, tt), ch = 2 * nt + (c.g >> 1);
        char *X = c.smem;
        split_store(X + x_off(tok, ch) + 8 * (c.g & 1), X + x_off(tok, XCH + ch) + 8 * (c.g & 1), v);
    }

    // ---- LayerNorm over the D features of H -> panel (scaled by ACT).  Per wave: mean and centred sum of squares of its 16 NA
    // features per token, exchanged through LDS, combined by Chan's formula in every lane (two barriers: the first also fences the
    // panel's previous readers)
    static __device__ __forceinline__ void layer_norm_to_x(const Ctx &c0, const f32x4 (&H)[NA][NTT], const float *ln_w, const float *ln_b) {
        const Ctx c = ctx_local(c0);
        constexpr float FW = 16.0f * NA;
        float *stat = reinterpret_cast<float *>(c.smem + c.oS);
        f32x4 gw[NA], gb[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            gw[a] = *reinterpret_cast<const f32x4 *>(ln_w + 16 * (NA * c.w + a) + 4 * c.g) * ACT;
            gb[a] = *reinterpret_cast<const f32x4 *>(ln_b + 16 * (NA * c.w + a) + 4 * c.g) * ACT;
        }
        float mw[NTT], qw[NTT];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            f32x4 s4 = H[0][tt];
#pragma unroll
            for (int a = 1; a < NA; ++a) s4 = s4 + H[a][tt];
            mw[tt] = rows4_sum((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.0f / FW);
            f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const f32x4 dd = H[a][tt] - mw[tt];
                q4 = q4 + dd * dd;
            }
            qw[tt] = rows4_sum((q4[0] + q4[1]) + (q4[2] + q4[3]));
        }
        if (c.g == 0) {
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt)
                if (tok_ok(c, tt)) *reinterpret_cast<f32x2 *>(stat + (tok_of(c, tt) * 8 + c.w) * 2) = f32x2{mw[tt], qw[tt]};
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            const float *sp = stat + tok_of(c, tt) * 16;
            const f32x4 p0 = *reinterpret_cast<const f32x4 *>(sp), p1 = *reinterpret_cast<const f32x4 *>(sp + 4);
            const f32x4 p2 = *reinterpret_cast<const f32x4 *>(sp + 8), p3 = *reinterpret_cast<const f32x4 *>(sp + 12);
            const float m = (((p0[0] + p0[2]) + (p1[0] + p1[2])) + ((p2[0] + p2[2]) + (p3[0] + p3[2]))) * 0.125f;
            const f32x4 e0 = f32x4{p0[0], p0[2], p1[0], p1[2]} - m, e1 = f32x4{p2[0], p2[2], p3[0], p3[2]} - m;
            const f32x4 ee = e0 * e0 + e1 * e1;
            const float m2 = (((p0[1] + p0[3]) + (p1[1] + p1[3])) + ((p2[1] + p2[3]) + (p3[1] + p3[3]))) + FW * ((ee[0] + ee[1]) + (ee[2] + ee[3]));
            const float rstd = __builtin_amdgcn_rsqf(m2 * (1.0f / D) + SD_LN_EPS);
#pragma unroll
            for (int a = 0; a < NA; ++a) store_x(c, tt, NA * c.w + a, ((H[a][tt] - m) * rstd) * gw[a] + gb[a]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- K = D GEMM against the panel for NJ n-tiles (fragment streams pa[j]: [ks][plane][lane][8]) and all token tiles: the B fragment
    // of the next step and the A fragments of the next k-step are requested before the MFMAs of the current step
    template <int NJ, class F>
    static __device__ __forceinline__ void gemm_panel(const Ctx &c, const f16 *const (&pa)[NJ], F body) {
        const char *X = c.smem;
        const unsigned lo = (unsigned)c.lane * 8;
        f16x8 a[2][NJ][2], b[2][2];
        auto b_at = [&](int tt, int pl, int ks) __attribute__((always_inline)) { return x_off(tok_of(c, tt), pl * XCH + 4 * ks + c.g); };
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[0][j][pl] = glb16(pa[j] + lo + pl * 512);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[0][pl] = lds16(X + b_at(0, pl, 0));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                const int cur = (ks * NTT + tt) & 1, nxt = cur ^ 1;
                if (tt == 0 && ks + 1 < KS) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) a[(ks + 1) & 1][j][pl] = glb16(pa[j] + lo + ((ks + 1) * 2 + pl) * 512);
                }
                const int nks = tt + 1 < NTT ? ks : ks + 1, ntt = tt + 1 < NTT ? tt + 1 : 0;
                if (nks < KS) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) b[nxt][pl] = lds16(X + b_at(ntt, pl, nks));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) body(j, tt, a[ks & 1][j][0], a[ks & 1][j][1], b[cur][0], b[cur][1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // acc[a][tt] += W[n-tiles NA w + a] . X^T
    static __device__ __forceinline__ void gemm_own(const Ctx &c0, f32x4 (&acc)[NA][NTT], const f16 *wmat) {
        const Ctx c = ctx_local(c0);
        const f16 *pa[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) pa[a] = wmat + (long)(NA * c.w + a) * (KS * 2 * 512);
        gemm_panel<NA>(c, pa, [&](int j, int tt, f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl) __attribute__((always_inline)) { mma3(acc[j][tt], ah, al, bh, bl); });
    }

    static __device__ __forceinline__ void scale_h(f32x4 (&H)[NA][NTT], float f) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) H[a][tt] = H[a][tt] * f;
    }
    // this lane's slice of a bias vector (features 16 (NA w + a) + 4 g + r).  Requested BEFORE the GEMM whose result it completes: loaded at its
    // point of use, behind the pipeline's scheduling fences, every bias cost an exposed L2 round trip - at the robot's B = 1 (one workgroup, nothing
    // else to run) a dozen of them per layer
    struct BiasV { f32x4 v[NA]; };
    static __device__ __forceinline__ BiasV bias_load(const Ctx &c, const float *bias) {
        BiasV b;
#pragma unroll
        for (int a = 0; a < NA; ++a) b.v[a] = *reinterpret_cast<const f32x4 *>(bias + 16 * (NA * c.w + a) + 4 * c.g);
        return b;
    }
    // H = H * f + bias[feature]
    static __device__ __forceinline__ void unscale_h(f32x4 (&H)[NA][NTT], float f, const BiasV &b) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) H[a][tt] = H[a][tt] * f + b.v[a];
    }

    // ---- self-attention of head h on the residual accumulators (pre-scaled by ACT * s_o): H += Wo[:, head] . O_head^T.
    // Q (later O) in the Q buffer, K (later V) in the K buffer; five barriers
    static __device__ __forceinline__ void sa_head(const Ctx &c0, const LayerW &L, int h, f32x4 (&H)[NA][NTT], float scale_log2e) {
        Ctx c = ctx_local(c0);
        char *Qb = c.smem + c.oQ, *Kb = c.smem + c.oK;
        const int w = c.w;
        int g = c.g, t = c.t;
        // jobs of this wave: n-tiles w, w + 8, ... of the head's [Q | K | V] block (3 NQ n-tiles)
        f32x4 acc[NJOB][NTT];
#pragma unroll
        for (int j = 0; j < NJOB; ++j)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) acc[j][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f16 *pa[NJOB];
#pragma unroll
        for (int j = 0; j < NJOB; ++j) {
            const int job = min(w + 8 * j, 3 * NQ - 1), which = job / NQ, tile = job % NQ;
            pa[j] = L.w_in + (long)(which * (D / 16) + h * NQ + tile) * (KS * 2 * 512);
        }
        // the jobs' bias slices and the input scale, requested before the GEMM (see bias_load)
        f32x4 bjob[NJOB];
#pragma unroll
        for (int j = 0; j < NJOB; ++j) {
            const int job = min(w + 8 * j, 3 * NQ - 1), which = job / NQ, tile = job % NQ;
            bjob[j] = *reinterpret_cast<const f32x4 *>(L.b_in + which * D + HD * h + 16 * tile + 4 * g) * ACT;
        }
        const float c_in = 1.0f / L.sc[SC_IN];   // accumulator -> ACT * value
        // (a wave has NJOB or NJOB - 1 jobs: the choice is made once, outside the pipeline - a wave-uniform branch inside every one of its
        // unrolled steps made hipcc keep the whole prefetch state live across all of them: 277 spilled registers at hidden_dim 128, T = 100)
        auto body = [&](int j, int tt, f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl) __attribute__((always_inline)) { mma3(acc[j][tt], ah, al, bh, bl); };
        if (w + 8 * (NJOB - 1) < 3 * NQ) {
            gemm_panel<NJOB>(c, pa, body);
        } else if constexpr (NJOB > 1) {
            const f16 *pb[NJOB - 1];
#pragma unroll
            for (int j = 0; j < NJOB - 1; ++j) pb[j] = pa[j];
            gemm_panel<NJOB - 1>(c, pb, body);
        }
        __syncthreads();   // B1: the previous head's readers of Q / O and K / V are done
        c = ctx_local(c0); g = c.g; t = c.t;
#pragma unroll
        for (int j = 0; j < NJOB; ++j) {
            const int job = w + 8 * j, which = job / NQ, tile = job % NQ;
            if (job >= 2 * NQ) continue;   // (V tiles wait for K to die)
            const f32x4 bv = bjob[j];
            char *dst = which == 0 ? Qb : Kb;
            const int ch = 2 * tile + (g >> 1);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                if (!tok_ok(c, tt)) continue;
                const int tok = tok_of(c, tt);
                split_store(dst + q_off(tok, ch) + 8 * (g & 1), dst + q_off(tok, QCH + ch) + 8 * (g & 1), acc[j][tt] * c_in + bv);
            }
        }
        __syncthreads();   // B2: Q, K complete
        c = ctx_local(c0); g = c.g; t = c.t;
        f32x4 S[NTT];
        float psum = 1.f;
#if defined(TG_ABL) && (TG_ABL & 8)
        for (int kt = 0; kt < NTT; ++kt) S[kt] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (false) {
#else
        if (w < NTT) {
#endif
            // scores S^T[key][query] = K Q^T of query tile w
            f16x8 qf[KH][2];
            const int qtok = min(16 * w + t, c.T - 1);
#pragma unroll
            for (int kk = 0; kk < KH; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) qf[kk][pl] = lds16(Qb + q_off(qtok, pl * QCH + 4 * kk + g));
#pragma unroll
            for (int kt = 0; kt < NTT; ++kt) {
                S[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int ktok = tok_of(c, kt);
#pragma unroll
                for (int kk = 0; kk < KH; ++kk)
                    mma3(S[kt], lds16(Kb + q_off(ktok, 4 * kk + g)), lds16(Kb + q_off(ktok, QCH + 4 * kk + g)), qf[kk][0], qf[kk][1]);
            }
            const float c_s = scale_log2e / (ACT * ACT);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (LAST0 + 4 * g + r >= c.T) S[NTT - 1][r] = -INFINITY;
            f32x4 m4 = S[0];
#pragma unroll
            for (int kt = 1; kt < NTT; ++kt) m4 = f32x4{fmaxf(m4[0], S[kt][0]), fmaxf(m4[1], S[kt][1]), fmaxf(m4[2], S[kt][2]), fmaxf(m4[3], S[kt][3])};
            const float m = rows4_max(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])));
            const float mb = m * c_s - 10.0f;   // probabilities carry 2^10
            f32x4 ps = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NTT; ++kt) {
                const f32x4 e = S[kt] * c_s - mb;
                S[kt] = f32x4{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1]), __builtin_amdgcn_exp2f(e[2]), __builtin_amdgcn_exp2f(e[3])};
                ps = ps + S[kt];
            }
            psum = rows4_sum((ps[0] + ps[1]) + (ps[2] + ps[3]));
        }
        __syncthreads();   // B3: K is dead
        c = ctx_local(c0); g = c.g; t = c.t;
#pragma unroll
        for (int j = 0; j < NJOB; ++j) {
            const int job = w + 8 * j, tile = job % NQ;
            if (job < 2 * NQ || job >= 3 * NQ) continue;
            const f32x4 bv = bjob[j];
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                if (!tok_ok(c, tt)) continue;
                char *at = Kb + tok_of(c, tt) * VROW + 2 * (16 * tile + 4 * g);
                split_store(at, at + 2 * HD, acc[j][tt] * c_in + bv);
            }
        }
        __syncthreads();   // B4: V complete (every wave has read its Q fragments: O may overwrite Q)
        c = ctx_local(c0); g = c.g; t = c.t;
#if defined(TG_ABL) && (TG_ABL & 16)
        if (false) {
#else
        if (w < NTT) {
#endif
            // O^T = V^T P^T: P^T straight from the score accumulators (key order 16 (e >> 2) + 4 g + (e & 3) inside a pair of key tiles),
            // V^T through transposing LDS reads
            constexpr int NKP = (NTT + 1) / 2;
            f32x4 O[NQ];
#pragma unroll
            for (int ft = 0; ft < NQ; ++ft) O[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int q4 = t >> 2, p4 = t & 3;
#pragma unroll
            for (int kp = 0; kp < NKP; ++kp) {
                const f32x4 pa4 = S[2 * kp], pb4 = 2 * kp + 1 < NTT ? S[2 * kp + 1 < NTT ? 2 * kp + 1 : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
                f16x4 pah, pal, pbh, pbl;
                split4(pa4, pah, pal);
                split4(pb4, pbh, pbl);
                const f16x8 ph = __builtin_shufflevector(pah, pbh, 0, 1, 2, 3, 4, 5, 6, 7), pl = __builtin_shufflevector(pal, pbl, 0, 1, 2, 3, 4, 5, 6, 7);
                const int r0 = min(32 * kp + 4 * g + q4, c.T - 1), r1 = min(32 * kp + 16 + 4 * g + q4, c.T - 1);
                const char *v0 = Kb + r0 * VROW + 8 * p4, *v1 = Kb + r1 * VROW + 8 * p4;
#pragma unroll
                for (int ft = 0; ft < NQ; ++ft) {
                    f16x8 vf[2];
#pragma unroll
                    for (int pn = 0; pn < 2; ++pn) {
                        const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(v0 + pn * 2 * HD + ft * 32));
                        const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(v1 + pn * 2 * HD + ft * 32));
                        vf[pn] = __builtin_shufflevector(__builtin_bit_cast(f16x4, x0), __builtin_bit_cast(f16x4, x1), 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                    mma3(O[ft], vf[0], vf[1], ph, pl);
                }
            }
            const float inv = 1.0f / psum;
            const int tok = 16 * w + t;
            if (tok < c.T) {
#pragma unroll
                for (int ft = 0; ft < NQ; ++ft) {
                    const int ch = 2 * ft + (g >> 1);
                    split_store(Qb + q_off(tok, ch) + 8 * (g & 1), Qb + q_off(tok, QCH + ch) + 8 * (g & 1), O[ft] * inv);
                }
            }
        }
        __syncthreads();   // B5: O complete
        c = ctx_local(c0); g = c.g; t = c.t;
        // out-projection of this head: K = HD
#if defined(TG_ABL) && (TG_ABL & 64)
        if (false)
#endif
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            f16x8 wo[NA][2];
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    wo[a][pl] = glb16(L.w_o + ((long)(NA * w + a) * KS + h * KH + kk) * (2 * 512) + pl * 512 + c.lane * 8);
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                const int tok = tok_of(c, tt);
                const f16x8 bh = lds16(Qb + q_off(tok, 4 * kk + g)), bl = lds16(Qb + q_off(tok, QCH + 4 * kk + g));
#pragma unroll
                for (int a = 0; a < NA; ++a) mma3(H[a][tt], wo[a][0], wo[a][1], bh, bl);
            }
        }
    }

    // ---- cross-attention over the projected memory (Mc context rows streamed from HBM + the step token); on entry the panel holds
    // LN2(h), on exit H has the block's output added
    static __device__ __forceinline__ void cross_block(const Ctx &c0, const LayerW &L, f32x4 (&H)[NA][NTT], long traj, const StepArgs &a, long sblk) {
        Ctx c = ctx_local(c0);
        char *X = c.smem;
        const int w = c.w;
        int g = c.g, t = c.t;
        // Q_c = Wq LN2(h) + bq -> the panel (ACT * q as split planes), in place
        {
            f32x4 U[NA][NTT];
#pragma unroll
            for (int aa = 0; aa < NA; ++aa)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) U[aa][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const BiasV bq = bias_load(c, L.b_q);
            const float cq = 1.0f / L.sc[SC_Q];
            gemm_own(c, U, L.w_q);
            __syncthreads();   // every wave has read LN2(h)
#pragma unroll
            for (int aa = 0; aa < NA; ++aa) {
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) store_x(c, tt, NA * w + aa, U[aa][tt] * cq + bq.v[aa] * ACT);
            }
            __syncthreads();
        }
        // units (head, query tile): u = w, w + 8, ...
        const float sK = L.sc[SC_K], sV = L.sc[SC_V], sKs = L.sc[SC_KS], sVs = L.sc[SC_VS];
        const float sv = fminf(sV, sVs), m_c = sv / sV, m_s = sv / sVs;   // common value scale (sd_traj.h: step_scale)
        const float cl_c = a.scale_log2e / (ACT * sK), cl_s = a.scale_log2e / (ACT * sKs);
        const f16 *kvs = L.kvs + sblk * 4 * D;
#pragma unroll 1
        for (int ui = 0; ui < NUNIT; ++ui) {
            const int u = w + 8 * ui;
            if (u >= 4 * NTT) break;
            c = ctx_local(c0); g = c.g; t = c.t;
            const int hh = u / NTT, qt = u - hh * NTT;
            const int qtok = qt < NTT - 1 ? 16 * qt + t : c.tokl;
            f16x8 qf[KH][2];
#pragma unroll
            for (int kk = 0; kk < KH; ++kk)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) qf[kk][pl] = lds16(X + x_off(qtok, pl * XCH + 4 * (hh * KH + kk) + g));
            f32x4 O[NQ];
#pragma unroll
            for (int ft = 0; ft < NQ; ++ft) O[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
            float m_run = -1e30f, l_run = 0.f;
            const f16 *kp = L.kp + traj * a.kv_traj_halfs + (long)hh * a.nkp * (2 * KH * 2 * 512) + c.lane * 8;
            const f16 *vp = L.vp + traj * a.kv_traj_halfs + (long)hh * a.nkp * (NQ * 2 * 512) + c.lane * 8;
            // one pair of 32 keys in two pieces: scores (two 16-key tiles) - after which the K fragments are dead and the next pair's may be
            // requested into the same registers - and online softmax + O += V^T P^T
            auto scores = [&](const f16x8 (&kf)[2][KH][2], f32x4 &S0, f32x4 &S1) __attribute__((always_inline)) {
                S0 = f32x4{0.f, 0.f, 0.f, 0.f};
                S1 = S0;
#pragma unroll
                for (int kk = 0; kk < KH; ++kk) {
                    mma3(S0, kf[0][kk][0], kf[0][kk][1], qf[kk][0], qf[kk][1]);
                    mma3(S1, kf[1][kk][0], kf[1][kk][1], qf[kk][0], qf[kk][1]);
                }
            };
            auto finish = [&](f32x4 S0, f32x4 S1, const f16x8 (&vf)[NQ][2], float cl, float pm, int nvalid) __attribute__((always_inline)) {
                S0 = S0 * cl;
                S1 = S1 * cl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (4 * g + r >= nvalid) S0[r] = -INFINITY;
                    if (16 + 4 * g + r >= nvalid) S1[r] = -INFINITY;
                }
                const float mx = rows4_max(fmaxf(fmaxf(fmaxf(S0[0], S0[1]), fmaxf(S0[2], S0[3])), fmaxf(fmaxf(S1[0], S1[1]), fmaxf(S1[2], S1[3]))));
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                f32x4 p0, p1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p0[r] = __builtin_amdgcn_exp2f(S0[r] - m_new);
                    p1[r] = __builtin_amdgcn_exp2f(S1[r] - m_new);
                }
                const f32x4 ps = p0 + p1;
                l_run = l_run * alpha + rows4_sum((ps[0] + ps[1]) + (ps[2] + ps[3]));
                f16x4 ah, al, bh, bl;
                split4(p0 * (PSC * pm), ah, al);
                split4(p1 * (PSC * pm), bh, bl);
                const f16x8 ph = __builtin_shufflevector(ah, bh, 0, 1, 2, 3, 4, 5, 6, 7), pl = __builtin_shufflevector(al, bl, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int ft = 0; ft < NQ; ++ft) {
                    O[ft] = O[ft] * alpha;
                    mma3(O[ft], vf[ft][0], vf[ft][1], ph, pl);
                }
            };
            const int ncp = a.nkp;
            f16x8 kf[2][KH][2], vf[NQ][2];
            f32x4 S0, S1;
            if (ncp > 0) {
#pragma unroll
                for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                    for (int kk = 0; kk < KH; ++kk)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) kf[tl][kk][pl] = glb16(kp + ((tl * KH + kk) * 2 + pl) * 512);
            }
#pragma unroll 1
            for (int p = 0; p < ncp; ++p) {
                // V^T of this pair is requested before the scores, K of the next pair right after them: each stream is one phase ahead
#pragma unroll
                for (int ft = 0; ft < NQ; ++ft)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) vf[ft][pl] = glb16(vp + ((long)p * NQ + ft) * (2 * 512) + pl * 512);
                __builtin_amdgcn_sched_barrier(0);
                scores(kf, S0, S1);
                __builtin_amdgcn_sched_barrier(0);
                if (p + 1 < ncp) {
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                        for (int kk = 0; kk < KH; ++kk)
#pragma unroll
                            for (int pl = 0; pl < 2; ++pl) kf[tl][kk][pl] = glb16(kp + (((long)(p + 1) * 2 + tl) * KH + kk) * (2 * 512) + pl * 512);
                }
                __builtin_amdgcn_sched_barrier(0);
                finish(S0, S1, vf, cl_c, m_c, a.Mc - 32 * p);
            }
            {   // the step token: one key (row 0 of a pair whose other rows are zero), from the compact split rows
                const bool krow = t == 0;
#pragma unroll
                for (int kk = 0; kk < KH; ++kk)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        kf[0][kk][pl] = krow ? glb16(kvs + pl * D + hh * HD + 32 * kk + 8 * g) : zero8();
                        kf[1][kk][pl] = zero8();
                    }
#pragma unroll
                for (int ft = 0; ft < NQ; ++ft)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const f16 sv1 = g == 0 ? kvs[(2 + pl) * D + hh * HD + 16 * ft + t] : (f16)0;
                        vf[ft][pl] = f16x8{sv1, 0, 0, 0, 0, 0, 0, 0};
                    }
                scores(kf, S0, S1);
                finish(S0, S1, vf, cl_s, m_s, 1);
            }
            // o = O / (PSC sv l) -> ACT * o over this unit's own Q block of the panel
            const float inv = ACT / (PSC * sv * l_run);
            if (qt < NTT - 1 || c.okl) {
#pragma unroll
                for (int ft = 0; ft < NQ; ++ft) {
                    const int ch = 2 * (hh * NQ + ft) + (g >> 1);
                    split_store(X + x_off(qtok, ch) + 8 * (g & 1), X + x_off(qtok, XCH + ch) + 8 * (g & 1), O[ft] * inv);
                }
            }
        }
        __syncthreads();   // the attention output is complete in the panel
        const float up = ACT * L.sc[SC_OC];
        const BiasV boc = bias_load(c, L.b_oc);
        scale_h(H, up);
        gemm_own(c, H, L.w_oc);
        unscale_h(H, 1.0f / up, boc);
    }

    static __device__ __forceinline__ void decoder_layer(const Ctx &c0, const LayerW &L, f32x4 (&H)[NA][NTT], long traj, const StepArgs &a, long sblk) {
        const Ctx &c = c0;
        // ---- self-attention block: h += Wo . SA(LN1(h)) + bo   (the panel holds LN1(h))
        {
            const float up = ACT * L.sc[SC_O];
            const BiasV bo = bias_load(c, L.b_o);
            scale_h(H, up);
#if !(defined(TG_ABL) && (TG_ABL & 1))
#pragma unroll 1
            for (int h = 0; h < 4; ++h) sa_head(c, L, h, H, a.scale_log2e);
#endif
            unscale_h(H, 1.0f / up, bo);
        }
        layer_norm_to_x(c, H, L.n2_w, L.n2_b);
#if !(defined(TG_ABL) && (TG_ABL & 2))
        cross_block(c, L, H, traj, a, sblk);
#endif
        layer_norm_to_x(c, H, L.n3_w, L.n3_b);
#if !(defined(TG_ABL) && (TG_ABL & 4))
        // ---- feed-forward: h += W2 gelu(W1 LN3(h) + b1) + b2
        {
            f32x4 U[NA][NTT];
#pragma unroll
            for (int aa = 0; aa < NA; ++aa)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) U[aa][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const BiasV bb1 = bias_load(c0, L.b_1), bb2 = bias_load(c0, L.b_2);
            const float c1 = 1.0f / (ACT * L.sc[SC_1]), up = ACT * L.sc[SC_2];
            gemm_own(c0, U, L.w_1);
            __syncthreads();   // every wave has read LN3(h): the panel receives gelu(u)
            const Ctx c = ctx_local(c0);
#pragma unroll
            for (int aa = 0; aa < NA; ++aa) {
                const f32x4 b1 = bb1.v[aa];
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    const f32x4 pre = U[aa][tt] * c1 + b1;
                    const f32x2 g0 = gelu_erf_as2(f32x2{pre[0], pre[1]}) * ACT, g1 = gelu_erf_as2(f32x2{pre[2], pre[3]}) * ACT;
                    store_x(c, tt, NA * c.w + aa, f32x4{g0[0], g0[1], g1[0], g1[1]});
                }
            }
            __syncthreads();
            scale_h(H, up);
            gemm_own(c0, H, L.w_2);
            unscale_h(H, 1.0f / up, bb2);
        }
#endif
    }

    static __device__ __forceinline__ void step_body(const StepArgs &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        Ctx c;
        ctx_init(c, smem, a.T);
        const long traj = blockIdx.x;
        const int J = a.J;
        f32x4 H[NA][NTT];
        // ---- embedding: h^T = Wemb . x^T + b + pe^T.  x rows -> the K buffer as split planes (k = joint, zero-padded to 32)
        {
            char *Eb = c.smem + c.oK;
            const float *xr = a.x + traj * (long)a.T * J;
            for (int i = threadIdx.x; i < a.T * 4; i += NTHREADS) {
                const int tok = i >> 2, chunk = i & 3;
                f16x8 h8, l8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = 8 * chunk + e;
                    const float v = j < J ? xr[tok * J + j] : 0.f;
                    h8[e] = (f16)v;
                    l8[e] = (f16)(v - (float)h8[e]);
                }
                *reinterpret_cast<f16x8 *>(Eb + e_off(tok, chunk)) = h8;
                *reinterpret_cast<f16x8 *>(Eb + e_off(tok, chunk | 4)) = l8;
            }
            __syncthreads();
            const float c_e = 1.0f / a.sc_io[SC_EMB];
#pragma unroll
            for (int aa = 0; aa < NA; ++aa) {
                const int nt = NA * c.w + aa;
                const f16 *wp = a.w_emb + (long)nt * (2 * 512) + c.lane * 8;
                const f16x8 ah = glb16(wp), al = glb16(wp + 512);
                const f32x4 be = *reinterpret_cast<const f32x4 *>(a.b_emb + 16 * nt + 4 * c.g);
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    const int tok = tok_of(c, tt);
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    mma3(acc, ah, al, lds16(Eb + e_off(tok, c.g)), lds16(Eb + e_off(tok, 4 | c.g)));
                    H[aa][tt] = acc * c_e + (be + *reinterpret_cast<const f32x4 *>(a.pe + (long)tok * D + 16 * nt + 4 * c.g));
                }
            }
        }
        const long sblk = a.step_per_traj ? (a.step_map ? (long)a.step_map[traj] : traj) : 0L;
        layer_norm_to_x(c, H, a.layer[0].n1_w, a.layer[0].n1_b);
#pragma unroll 1
        for (int l = 0; l < a.L; ++l) {
            decoder_layer(c, a.layer[l], H, traj, a, sblk);
            if (l + 1 < a.L) layer_norm_to_x(c, H, a.layer[l + 1].n1_w, a.layer[l + 1].n1_b);
        }
        // ---- fc_out + DDIM: eps^T = Wout . h^T + b.  h has no a-priori bound: one power-of-two scale per token
        {
            float *stat = reinterpret_cast<float *>(c.smem + c.oS);
            float am[NTT];
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                float m = 0.f;
#pragma unroll
                for (int aa = 0; aa < NA; ++aa) {
                    const f32x4 v = H[aa][tt];
                    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                }
                am[tt] = rows4_max(m);
            }
            if (c.g == 0) {
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt)
                    if (tok_ok(c, tt)) stat[tok_of(c, tt) * 8 + c.w] = am[tt];
            }
            __syncthreads();   // also: the panel's readers (last W2 GEMM) are done
            auto tok_scale = [&](int tok) __attribute__((always_inline)) {
                const float *sp = stat + tok * 8;
                const f32x4 p0 = *reinterpret_cast<const f32x4 *>(sp), p1 = *reinterpret_cast<const f32x4 *>(sp + 4);
                const float m = fmaxf(fmaxf(fmaxf(p0[0], p0[1]), fmaxf(p0[2], p0[3])), fmaxf(fmaxf(p1[0], p1[1]), fmaxf(p1[2], p1[3])));
                return f16_scale_from_bits(__builtin_bit_cast(unsigned, m));
            };
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                const float s = tok_scale(tok_of(c, tt));
#pragma unroll
                for (int aa = 0; aa < NA; ++aa) store_x(c, tt, NA * c.w + aa, H[aa][tt] * s);
            }
            __syncthreads();
            if (c.w < NTT) {
                const char *X = c.smem;
                f32x4 E[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                const int tok = c.w < NTT - 1 ? 16 * c.w + c.t : c.tokl;
                const bool ok = c.w < NTT - 1 || c.okl;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const f16x8 bh = lds16(X + x_off(tok, 4 * ks + c.g)), bl = lds16(X + x_off(tok, XCH + 4 * ks + c.g));
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f16 *wp = a.w_out + ((long)(n * KS + ks) * 2) * 512 + c.lane * 8;
                        mma3(E[n], glb16(wp), glb16(wp + 512), bh, bl);
                    }
                }
                const float c_o = 1.0f / (tok_scale(tok) * a.sc_io[SC_OUT]);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int j0 = 16 * n + 4 * c.g;
                    if (!ok || j0 >= J) continue;
                    const long at = (traj * a.T + tok) * J + j0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (j0 + r >= J) continue;
                        const float e = E[n][r] * c_o + a.b_out[j0 + r];
                        if (a.eps_out) a.eps_out[at + r] = e;
                        if (a.update_x) {   // the oracle's fp32 op order (oracle/ddim_ref.py)
                            const float x0 = (a.x[at + r] - a.c1 * e) / a.c0;
                            a.x[at + r] = a.c2 * x0 + a.c3 * e;
                        }
                    }
                }
            }
        }
    }
};   // struct TG

template <int D, int NTT>
__global__ __launch_bounds__(NTHREADS) void traj_step_generic_kernel(StepArgs a) { TG<D, NTT>::step_body(a); }

}   // namespace tg

#ifndef TG_NO_HOST   // (register-pressure experiments compile single instantiations of the kernel without the dispatch below)
// ======================================================================================
// host side
// ======================================================================================
namespace {

size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

struct GScratch {
    f16 *wpl;        // per layer: in_proj (3 d^2) | Wo | Wq | Woc | W1 | W2, each as [n-tile][ks][plane][lane][8]
    f16 *wio;        // embedding (d x 32) | fc_out (32 x d) planes
    f16 *kp, *vp;    // per layer, per trajectory: nkp * 64 * d halfs each
    f16 *kvs;        // per layer: n_tok x 4 x d halfs
    float *scales;   // (L + 1) rows of 16
    unsigned *maxbits;
    int nkp;
    size_t kv_layer_halfs, kvs_layer_halfs;
};

GScratch gcarve(float *ws, int B, int Mc, int d, int L, int n_tok) {
    GScratch s;
    size_t off = 0;
    s.nkp = (Mc + 31) / 32;
    s.kv_layer_halfs = (size_t)B * s.nkp * 64 * d;
    s.kvs_layer_halfs = (size_t)n_tok * 4 * d;
    s.wpl = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * 8 * d * d);
    s.wio = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)64 * d);
    s.kp = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * s.kv_layer_halfs / 2 + 1);
    s.vp = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * s.kv_layer_halfs / 2 + 1);
    s.kvs = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * s.kvs_layer_halfs / 2 + 1);
    s.scales = ws + off; off += align64((size_t)(L + 1) * 16);
    s.maxbits = reinterpret_cast<unsigned *>(ws + off); off += align64((size_t)(L + 1) * 16);
    return s;
}

// weight planes of layer l: which = 0 in_proj, 1 Wo, 2 Wq, 3 Woc, 4 W1, 5 W2
f16 *wplane(const GScratch &s, int l, int d, int which) {
    const size_t off = which == 0 ? 0 : (size_t)(2 + which) * 2 * d * d;   // in_proj takes 3 blocks of 2 d^2 halfs
    return s.wpl + (size_t)l * 16 * d * d + off;
}

int zero_words(unsigned *p, int n, hipStream_t st) {
    SD_LAUNCH(tg::zero_words_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, p, n);
    SD_CHECK_LAUNCH("zero_words_kernel");
    return 0;
}

typedef void (*StepFn)(tg::StepArgs);
template <int D>
StepFn step_fn(int ntt) {
    switch (ntt) {
        case 1: return tg::traj_step_generic_kernel<D, 1>;
        case 2: return tg::traj_step_generic_kernel<D, 2>;
        case 3: return tg::traj_step_generic_kernel<D, 3>;
        default: break;
    }
    if constexpr (D <= 256) {
        switch (ntt) {
            case 4: return tg::traj_step_generic_kernel<D, 4>;
            case 5: return tg::traj_step_generic_kernel<D, 5>;
            case 6: return tg::traj_step_generic_kernel<D, 6>;
            case 7: return tg::traj_step_generic_kernel<D, 7>;
            default: break;
        }
    }
    return nullptr;
}

}   // namespace

bool trajg_ok(int d, int heads, int T, int Mk, int J, int L) {
    static const char *e1 = getenv("SD_SAMPLER_TRAJ"), *e2 = getenv("SD_SAMPLER_GEMM"), *e3 = getenv("SD_SAMPLER_TRAJG");
    if ((e1 && strcmp(e1, "0") == 0) || (e2 && strcmp(e2, "f32") == 0) || (e3 && strcmp(e3, "0") == 0)) return false;
    if (!(d == 128 || d == 256 || d == 512) || heads != 4) return false;
    return T >= 1 && T <= (d == 512 ? 48 : 100) && Mk >= 1 && J >= 1 && J <= 32 && L >= 1 && L <= tg::MAX_L;
}

size_t trajg_workspace_floats(int B, int Mc, int d, int L, int n_tok) {
    const size_t nkp = (size_t)(Mc + 31) / 32;
    return align64((size_t)L * 8 * d * d) + align64((size_t)64 * d) + 2 * align64((size_t)L * B * nkp * 32 * d + 1) +
           align64((size_t)L * n_tok * 2 * d + 1) + 2 * align64((size_t)(L + 1) * 16) + 64;
}

int trajg_prepare_weights(const sd_denoiser_weights *w, float *gws, int B, int Mc, int n_tok, hipStream_t st) {
    const int d = w->d, L = w->L;
    const GScratch s = gcarve(gws, B, Mc, d, L, n_tok);
    int rc;
    for (int l = 0; l <= L; ++l)   // words SC_IN .. SC_2 of the layers' rows, SC_EMB / SC_OUT of row L
        if ((rc = zero_words(s.maxbits + l * 16 + (l < L ? tg::SC_IN : tg::SC_EMB), l < L ? 6 : 2, st))) return rc;
    for (int pass = 0; pass < 2; ++pass) {
        for (int l = 0; l < L; ++l) {
            const sd_layer_weights &lw = w->layers[l];
            const float *mats[6] = {lw.sa_in_w, lw.sa_out_w, lw.ca_in_w, lw.ca_out_w, lw.lin1_w, lw.lin2_w};   // (ca_in_w: its first d rows = Wq)
            const int rows[6] = {3 * d, d, d, d, d, d};
            for (int m = 0; m < 6; ++m) {
                unsigned *mb = s.maxbits + l * 16 + m;
                if (pass == 0) {
                    SD_LAUNCH(tg::absmax_kernel, dim3(grid_for((long)rows[m] * d)), dim3(256), 0, st, mats[m], (long)rows[m] * d, mb);
                    SD_CHECK_LAUNCH("absmax_kernel");
                } else {
                    SD_LAUNCH(tg::pack_w_kernel, dim3(grid_for((long)rows[m] * d / 8)), dim3(256), 0, st, mats[m], rows[m], d, rows[m], d, mb,
                              wplane(s, l, d, m), s.scales + l * 16 + m);
                    SD_CHECK_LAUNCH("pack_w_kernel");
                }
            }
        }
        unsigned *mbe = s.maxbits + L * 16 + tg::SC_EMB, *mbo = s.maxbits + L * 16 + tg::SC_OUT;
        if (pass == 0) {
            SD_LAUNCH(tg::absmax_kernel, dim3(grid_for((long)d * w->J)), dim3(256), 0, st, w->emb_w, (long)d * w->J, mbe);
            SD_CHECK_LAUNCH("absmax_kernel");
            SD_LAUNCH(tg::absmax_kernel, dim3(grid_for((long)d * w->J)), dim3(256), 0, st, w->out_w, (long)d * w->J, mbo);
            SD_CHECK_LAUNCH("absmax_kernel");
        } else {
            SD_LAUNCH(tg::pack_w_kernel, dim3(grid_for((long)d * 4)), dim3(256), 0, st, w->emb_w, d, w->J, d, 32, mbe, s.wio, s.scales + L * 16 + tg::SC_EMB);
            SD_CHECK_LAUNCH("pack_w_kernel");
            SD_LAUNCH(tg::pack_w_kernel, dim3(grid_for((long)32 * d / 8)), dim3(256), 0, st, w->out_w, w->J, d, 32, d, mbo, s.wio + (size_t)2 * 32 * d,
                      s.scales + L * 16 + tg::SC_OUT);
            SD_CHECK_LAUNCH("pack_w_kernel");
        }
    }
    return 0;
}

int trajg_prepare_ctx(const sd_denoiser_weights *w, float *gws, const float *ctx, float *kvtmp, int B, int Mc, int n_tok, hipStream_t st) {
    const int d = w->d, L = w->L;
    const GScratch s = gcarve(gws, B, Mc, d, L, n_tok);
    if (Mc == 0) return 0;   // no planes; the context scales are set from the step tokens' (trajg_prepare_steps)
    int rc;
    SD_LAUNCH(tg::zero_word_cols16_kernel, dim3(1), dim3(64), 0, st, s.maxbits, L, (int)tg::SC_K, 2);
    SD_CHECK_LAUNCH("zero_word_cols16_kernel");
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        float *kvl = kvtmp + (size_t)l * B * Mc * 2 * d;
        // the memory is NOT layer-normed: K = mem Wk^T + bk, V = mem Wv^T + bv (rows [d, 3 d) of in_proj), fp32 results of the split-fp16 row GEMM
        if ((rc = linear(ctx, lw.ca_in_w + (size_t)d * d, lw.ca_in_b + d, nullptr, nullptr, nullptr, kvl, B * Mc, 2 * d, d, 0, st, 0))) return rc;
        unsigned *mb = s.maxbits + l * 16;
        SD_LAUNCH(tg::absmax_kv_kernel, dim3(grid_for((long)B * Mc * 2 * d)), dim3(256), 0, st, kvl, (long)B * Mc, d, mb + tg::SC_K, mb + tg::SC_V);
        SD_CHECK_LAUNCH("absmax_kv_kernel");
        SD_LAUNCH(tg::pack_k_kernel, dim3(grid_for((long)B * s.nkp * 8 * (d / 128) * 64)), dim3(256), 0, st, kvl, (long)B, Mc, s.nkp, d, mb + tg::SC_K,
                  s.kp + (size_t)l * s.kv_layer_halfs, s.scales + l * 16 + tg::SC_K);
        SD_CHECK_LAUNCH("pack_k_kernel");
        SD_LAUNCH(tg::pack_vt_kernel, dim3(grid_for((long)B * s.nkp * 4 * (d / 64) * 64)), dim3(256), 0, st, kvl, (long)B, Mc, s.nkp, d, mb + tg::SC_V,
                  s.vp + (size_t)l * s.kv_layer_halfs, s.scales + l * 16 + tg::SC_V);
        SD_CHECK_LAUNCH("pack_vt_kernel");
    }
    return 0;
}

int trajg_prepare_steps(const sd_denoiser_weights *w, float *gws, const float *tokens, float *kvstep, int B, int Mc, int n_tok, hipStream_t st,
                        const int *map) {
    const int d = w->d, L = w->L;
    const GScratch s = gcarve(gws, B, Mc, d, L, n_tok);
    SD_LAUNCH(tg::zero_word_cols16_kernel, dim3(1), dim3(64), 0, st, s.maxbits, L, (int)tg::SC_KS, 2);
    SD_CHECK_LAUNCH("zero_word_cols16_kernel");
    tg::StepKvArgs ka{};
    for (int l = 0; l < L; ++l) {
        ka.wkv[l] = w->layers[l].ca_in_w + (size_t)d * d;   // the memory is NOT layer-normed: rows [d, 3 d) of in_proj
        ka.bkv[l] = w->layers[l].ca_in_b + d;
    }
    const dim3 kvgrid((unsigned)n_tok, (unsigned)L, (unsigned)(2 * d / 64));
    if (d == 128) SD_LAUNCH(tg::step_kv_all_kernel<128>, kvgrid, dim3(256), 0, st, ka, tokens, kvstep, (long)n_tok * 2 * d, s.maxbits, map);
    else if (d == 256) SD_LAUNCH(tg::step_kv_all_kernel<256>, kvgrid, dim3(256), 0, st, ka, tokens, kvstep, (long)n_tok * 2 * d, s.maxbits, map);
    else SD_LAUNCH(tg::step_kv_all_kernel<512>, kvgrid, dim3(256), 0, st, ka, tokens, kvstep, (long)n_tok * 2 * d, s.maxbits, map);
    SD_CHECK_LAUNCH("step_kv_all_kernel");
    SD_LAUNCH(tg::pack_step_all_kernel, dim3(grid_for((long)n_tok * 2 * d), (unsigned)L), dim3(256), 0, st, kvstep, (long)n_tok * 2 * d, (long)n_tok, d,
              s.maxbits, s.kvs, (long)s.kvs_layer_halfs, s.scales, Mc == 0 ? 1 : 0, map);
    SD_CHECK_LAUNCH("pack_step_all_kernel");
    return 0;
}

int trajg_step(const sd_denoiser_weights *w, float *gws, float *x, float *eps, int B, int T, int Mc, int i, int n_tok, const float *coef,
               bool per_traj, hipStream_t st, const int *map) {
    const int d = w->d, L = w->L;
    const GScratch s = gcarve(gws, B, Mc, d, L, n_tok);
    tg::StepArgs a{};
    a.x = x;
    a.eps_out = eps;
    a.w_emb = s.wio;
    a.b_emb = w->emb_b;
    a.pe = w->pe;
    a.w_out = s.wio + (size_t)2 * 32 * d;
    a.b_out = w->out_b;
    a.sc_io = s.scales + L * 16;
    if (coef) { a.c0 = coef[0]; a.c1 = coef[1]; a.c2 = coef[2]; a.c3 = coef[3]; }
    a.scale_log2e = (1.0f / sqrtf((float)(d / w->heads))) * 1.44269504088896340736f;
    a.T = T; a.B = B; a.J = w->J; a.L = L; a.Mc = Mc; a.nkp = s.nkp; a.update_x = coef ? 1 : 0;
    a.step_per_traj = per_traj ? 1 : 0;
    a.step_map = per_traj ? map : nullptr;
    a.kv_traj_halfs = (long)s.nkp * 64 * d;
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        tg::LayerW &q = a.layer[l];
        q.n1_w = lw.n1_w; q.n1_b = lw.n1_b; q.n2_w = lw.n2_w; q.n2_b = lw.n2_b; q.n3_w = lw.n3_w; q.n3_b = lw.n3_b;
        q.w_in = wplane(s, l, d, 0); q.w_o = wplane(s, l, d, 1); q.w_q = wplane(s, l, d, 2); q.w_oc = wplane(s, l, d, 3);
        q.w_1 = wplane(s, l, d, 4); q.w_2 = wplane(s, l, d, 5);
        q.b_in = lw.sa_in_b; q.b_o = lw.sa_out_b; q.b_q = lw.ca_in_b; q.b_oc = lw.ca_out_b; q.b_1 = lw.lin1_b; q.b_2 = lw.lin2_b;
        q.sc = s.scales + l * 16;
        q.kp = s.kp + (size_t)l * s.kv_layer_halfs;
        q.vp = s.vp + (size_t)l * s.kv_layer_halfs;
        q.kvs = s.kvs + (size_t)l * s.kvs_layer_halfs + (size_t)(per_traj ? 0 : i) * 4 * d;
    }
    const int ntt = (T + 15) / 16;
    const StepFn fn = d == 128 ? step_fn<128>(ntt) : d == 256 ? step_fn<256>(ntt) : step_fn<512>(ntt);
    if (!fn) return fail(SD_E_BADARG, "traj_step_generic_kernel: horizon out of range");
    const int hd = d / 4;
    const size_t lds = (size_t)T * (4 * d + 4 * hd + (4 * hd + 32) + 64);
    ProfScope prof(SD_KCLASS_TRAJ_STEP, st);
    static DevFlag attr_set[3][8];
    const int di = d == 128 ? 0 : d == 256 ? 1 : 2;
    if (!attr_set[di][ntt]) {
        const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) return fail((int)e, "traj_step_generic_kernel: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        attr_set[di][ntt] = true;
    }
    SD_LAUNCH(fn, dim3((unsigned)B), dim3(tg::NTHREADS), lds, st, a);
    SD_CHECK_LAUNCH("traj_step_generic_kernel");
    return 0;
}
#endif   // TG_NO_HOST
