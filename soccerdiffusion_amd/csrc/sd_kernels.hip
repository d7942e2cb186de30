// MI355X (gfx950 / CDNA4) kernels + C ABI for the SoccerDiffusion denoiser hot path.
// Interface and reference citations: include/soccerdiffusion_hip.h.  Design: DESIGN.md.
//
// Data and accumulation are fp32.  Contractions in this file run on v_mfma_f32_32x32x2_f32 (exact fp32 fma
// chain, 64 FLOP/clk/SIMD); sd_f16x3.h (included below) holds the split-fp16 versions of the row GEMMs, the
// folded decoder layer and the self-attention that linear() and sd_ddim_sample use where they apply.
// Fragment maps of the fp32 MFMA used throughout (wave = 64 lanes):
//   A operand: lane l holds A[i = l & 31][k = l >> 5]
//   B operand: lane l holds B[k = l >> 5][j = l & 31]
//   C/D:       lane l, reg r holds D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31]
// The k order inside a K-loop is free as long as A and B agree, so each lane fetches 4
// consecutive k (one 16-byte load) at k0 + 4 * (l >> 5) and feeds them to 4 MFMAs:
// together the two lane halves cover k0 .. k0 + 7.  Weights are torch-layout W[N][K]
// (K contiguous), which is exactly that B-fragment shape.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "../../include/soccerdiffusion_hip.h"

#include "sd_common.h"
#include "sd_panel.h"

static thread_local const char *g_last_error = "ok";
int fail(int code, const char *msg) {
    g_last_error = msg;
    return code;
}

// Device word that every dropout kernel adds to the high half of its Philox key at RUN time (NULL: none).  A training
// step captured into a hipGraph freezes its kernel arguments, so the per-step part of the mask key must come from
// memory: the host bumps the word before each replay (soccerdiffusion_amd.training.GraphedTrainStep).
const unsigned *g_dropout_epoch = nullptr;
extern "C" int sd_set_dropout_epoch(const uint32_t *device_word) {
    g_dropout_epoch = device_word;
    return 0;
}

static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
ProfScope::ProfScope(int cls, hipStream_t s) : on(g_prof_on), st(s) {
    if (on) {
        rec.cls = cls;
        rec.a = prof_event();
        rec.b = prof_event();
        (void)hipEventRecord(rec.a, st);
    }
}
ProfScope::~ProfScope() {
    if (on) {
        (void)hipEventRecord(rec.b, st);
        g_prof_recs.push_back(rec);
    }
}

// ======================================================================================
// Row-panel GEMM:  out[R,N] = act(LN?(A)[R,D] @ W[N,D]^T + bias) (+ res)
//
// One workgroup (4 waves) owns a panel of BM = 64 rows.  The panel (64 x D fp32) is read
// from HBM once, kept in LDS for all N/D output passes, and LayerNorm is applied to it in
// place (the whole row is resident, so the statistics need no extra pass over memory).
// Waves split the D output columns of a pass, so B fragments (weights) are private to a
// wave and go global -> registers directly (L2/L1 resident, 16 B per lane); the K loop
// has no barrier at all.  LDS rows are padded by 4 floats: 16 consecutive rows then hit
// 16 distinct 4-bank groups for ds_read_b128.
// ======================================================================================

template <int D>
__device__ __forceinline__ void panel_layer_norm(float *sA, int lda, const float *ln_w, const float *ln_b, int lane, int wave,
                                                 int rows);

template <int D, bool HAS_LN, int ACT, bool HAS_RES>
__global__ __launch_bounds__(256) void panel_gemm_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                          const float *__restrict__ bias,
                                                          const float *__restrict__ ln_w,
                                                          const float *__restrict__ ln_b, const float *res,
                                                          float *out, int R, int N, int lda) {
    using C = PanelCfg<D>;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const long r0 = (long)blockIdx.x * C::BM;

    // ---- panel load: rows are consecutive in memory, 16 B per lane, zero-fill the tail --
    constexpr int VEC_PER_ROW = D / 4;
    for (int i = tid; i < C::BM * VEC_PER_ROW; i += 256) {
        const int row = i / VEC_PER_ROW;
        const int c4 = i - row * VEC_PER_ROW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < R) v = *reinterpret_cast<const f32x4 *>(A + (r0 + row) * lda + c4 * 4);
        *reinterpret_cast<f32x4 *>(sA + row * C::LDA + c4 * 4) = v;
    }
    __syncthreads();

    if constexpr (HAS_LN) {
        panel_layer_norm<D>(sA, C::LDA, ln_w, ln_b, lane, wave, C::BM);
        __syncthreads();
    }

    const int wm = wave / C::WAVES_N;
    const int wn = wave % C::WAVES_N;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const float *aBase = sA + (wm * C::WM + l31) * C::LDA + 4 * half;
    const bool full_panel = r0 + C::BM <= R;  // workgroup-uniform

    for (int n0 = 0; n0 < N; n0 += D) {
        f32x16 acc[C::TM][C::TN];
        if constexpr (HAS_RES) {
            // the residual is the accumulator's initial value: its loads are in flight while
            // the first weight fragments arrive, and the add costs nothing
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) {
                const int col = n0 + wn * C::WN + tn * 32 + l31;
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long row = r0 + wm * C::WM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        acc[tm][tn][r] = (full_panel || row < R) ? res[row * N + col] : 0.f;
                    }
            }
        } else {
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
        }

        // ---- K loop: weight fragments prefetched one k-step (1024 MFMA cycles) ahead and A
        // fragments double-buffered; sched_barriers pin "issue next loads, then 16 MFMAs" so
        // the compiler cannot sink a load next to its use (it does otherwise: -40 %).
        const float *wBase = W + (long)(n0 + wn * C::WN + l31) * D + 4 * half;
        constexpr int NK = D / 8;
        f32x4 bf[2][C::TN], af[2][C::TM];
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) bf[0][tn] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D);
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) af[0][tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * C::LDA);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < NK) {
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
                    bf[nxt][tn] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D + (ks + 1) * 8);
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
                    af[nxt][tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * C::LDA + (ks + 1) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][tm][j], bf[cur][tn][j], acc[tm][tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: bias, activation, then a quad transpose so that every lane stores 16
        // bytes (4 consecutive columns of one row): 4x fewer store instructions -----------------
        const int i4 = lane & 3;
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int col = n0 + wn * C::WN + tn * 32 + l31;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) {
                float *op = out + (r0 + wm * C::WM + tm * 32 + 4 * half + i4) * N + n0 + wn * C::WN + tn * 32 + (l31 & ~3);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[tm][tn][4 * g + e] + bv;
                        if constexpr (ACT == 1) v = gelu_erf(v);
                        x[e] = v;
                    }
                    quad_transpose(x[0], x[1], x[2], x[3], lane);
                    if (full_panel || r0 + wm * C::WM + tm * 32 + 4 * half + i4 + 8 * g < R) {
                        const f32x4 v4 = {x[0], x[1], x[2], x[3]};
                        *reinterpret_cast<f32x4 *>(op + (long)(8 * g) * N) = v4;
                    }
                }
            }
        }
    }
}

template <int D>
static int launch_panel(const float *A, int lda, const float *W, const float *bias, const float *ln_w, const float *ln_b,
                        const float *res, float *out, int R, int N, int act, hipStream_t s) {
    using C = PanelCfg<D>;
    ProfScope prof(SD_KCLASS_PANEL_GEMM, s);
    dim3 grid((R + C::BM - 1) / C::BM), block(256);
    const size_t lds = C::LDS_BYTES;
#define SD_PANEL(LN_, ACT_, RES_)                                                                              \
    do {                                                                                                       \
        auto kfn = panel_gemm_kernel<D, LN_, ACT_, RES_>;                                                      \
        static DevFlag attr_set;                                                                                 \
        if (lds > 64 * 1024 && !attr_set) {                                                                    \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                   \
        }                                                                                                      \
        SD_LAUNCH(kfn, grid, block, lds, s, A, W, bias, ln_w, ln_b, res, out, R, N, lda);                  \
    } while (0)
    const bool ln = ln_w != nullptr;
    const bool rs = res != nullptr;
    if (ln && act == 0 && !rs) SD_PANEL(true, 0, false);
    else if (ln && act == 1 && !rs) SD_PANEL(true, 1, false);
    else if (!ln && act == 0 && !rs) SD_PANEL(false, 0, false);
    else if (!ln && act == 0 && rs) SD_PANEL(false, 0, true);
    else if (!ln && act == 1 && !rs) SD_PANEL(false, 1, false);
    else if (ln && act == 0 && rs) SD_PANEL(true, 0, true);
    else return fail(SD_E_BADARG, "sd_op_linear: unsupported LN/act/res combination");
#undef SD_PANEL
    SD_CHECK_LAUNCH("panel_gemm_kernel");
    return 0;
}

int linear16(const float *A, const float *W, const float *bias, const float *ln_w, const float *ln_b, const float *res,
             float *out, int R, int N, int d, int act, hipStream_t s, int lda);
int linear32(const float *A, const float *W, const float *bias, const float *ln_w, const float *ln_b, const float *res,
             float *out, int R, int N, int d, int act, hipStream_t s, int lda);

int linear(const float *A, const float *W, const float *bias, const float *ln_w, const float *ln_b, const float *res,
           float *out, int R, int N, int d, int act, hipStream_t s, int lda) {
    if (lda == 0) lda = d;
    if (lda < d || lda % 4 != 0) return fail(SD_E_BADARG, "linear: row stride must be >= d and a multiple of 4");
    if (!A || !W || !out || R <= 0 || N <= 0) return fail(SD_E_BADARG, "linear: null pointer or empty shape");
    if (N % d != 0) return fail(SD_E_BADDIM, "linear: N must be a multiple of d");
    return linear16(A, W, bias, ln_w, ln_b, res, out, R, N, d, act, s, lda);   // split-fp16 kernel (sd_f16x3.h)
}

// the same layer on v_mfma_f32_32x32x2_f32 only (exact fp32 fma chain, no operand range limits): the sampler's
// range-guard fallback (sd_ddim_sample_ex, max_mode <= 1) projects the memory keys / values with it
int linear32(const float *A, const float *W, const float *bias, const float *ln_w, const float *ln_b, const float *res,
             float *out, int R, int N, int d, int act, hipStream_t s, int lda) {
    if (lda == 0) lda = d;
    if (lda < d || lda % 4 != 0) return fail(SD_E_BADARG, "linear: row stride must be >= d and a multiple of 4");
    if (!A || !W || !out || R <= 0 || N <= 0) return fail(SD_E_BADARG, "linear: null pointer or empty shape");
    if (N % d != 0) return fail(SD_E_BADDIM, "linear: N must be a multiple of d");
    switch (d) {
        case 64: return launch_panel<64>(A, lda, W, bias, ln_w, ln_b, res, out, R, N, act, s);
        case 128: return launch_panel<128>(A, lda, W, bias, ln_w, ln_b, res, out, R, N, act, s);
        case 256: return launch_panel<256>(A, lda, W, bias, ln_w, ln_b, res, out, R, N, act, s);
        case 512: return launch_panel<512>(A, lda, W, bias, ln_w, ln_b, res, out, R, N, act, s);
    }
    return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
}

// ======================================================================================
// Row-chain kernels: the row-wise part of a transformer layer fused over one 64-row panel.
//
// Between two attention cores everything is row-local: out-projection + residual,
// LayerNorm, projections, GELU feed-forward, the next layer's LayerNorm + QKV.  A
// workgroup keeps the residual stream of its 64 rows in the MFMA accumulator registers
// (wave w owns columns [w*D/4, (w+1)*D/4) of every row: the same tile map for every GEMM,
// so "h += ..." is simply C-in = h) and the current GEMM input in the LDS panel.  Between
// GEMMs the accumulator is written back into the panel, LayerNorm runs in place, and the
// next GEMM starts: no HBM round trip, no launch, one HBM read (attention output + h) and
// one write (h, next q/qkv) per chain.
//   chain A (decoder):        h += a Wo^T + bo ;            q  = LN2(h) Wq^T + bq
//   chain B (decoder/encoder): h += a Wo^T + bo ; h += W2 gelu(W1 LN(h) + b1) + b2 ;
//                             qkv' = LN1'(h) Wqkv'^T + b'   (next layer, optional)
// ======================================================================================
template <int D>
struct ChainCfg : PanelCfg<D> {};

template <int D>
__device__ __forceinline__ void chain_gemm(f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN], const float *aBase,
                                           const float *wBase) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 8;
    f32x4 bf[2][C::TN], af[2][C::TM];
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) bf[0][tn] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D);
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) af[0][tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * C::LDA);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < NK) {
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
                bf[nxt][tn] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D + (ks + 1) * 8);
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
                af[nxt][tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * C::LDA + (ks + 1) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][tm][j], bf[cur][tn][j], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Weight-fragment ring that runs AHEAD of the GEMM it feeds: chain_prime() issues the loads of
// k-steps 0 and 1 of the NEXT GEMM right after the current K loop, i.e. before the current
// epilogue / stores / LayerNorm.  Their L2 latency hides behind that work, and — vmcnt being
// in-order — they are older than the epilogue's stores, so the next K loop does not wait for
// stores to drain before its first MFMA.  chain_gemm_primed() then keeps the ring 2 k-steps deep.
template <int D>
struct WeightRing {
    f32x4 b[3][PanelCfg<D>::TN];
};

template <int D>
__device__ __forceinline__ void chain_prime(WeightRing<D> &ring, const float *wBase) {
    using C = PanelCfg<D>;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn)
            ring.b[s][tn] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D + s * 8);
}

template <int D>
__device__ __forceinline__ void chain_gemm_primed(f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN], const float *aBase,
                                                  const float *wBase, WeightRing<D> &ring) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 8;
    f32x4 af[2][C::TM];
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) af[0][tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * C::LDA);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int cur = ks % 3, fill = (ks + 2) % 3;
        if (ks + 2 < NK) {
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
                ring.b[fill][tn] = *reinterpret_cast<const f32x4 *>(wBase + (long)tn * 32 * D + (ks + 2) * 8);
        }
        if (ks + 1 < NK) {
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
                af[(ks + 1) & 1][tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * C::LDA + (ks + 1) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][tm][j], ring.b[cur][tn][j], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}


template <int D>
__device__ __forceinline__ void chain_load_panel(float *sA, const float *src, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    constexpr int VEC_PER_ROW = D / 4;
    constexpr int ITERS = C::BM * VEC_PER_ROW / 256, BATCH = ITERS < 16 ? ITERS : 16;
    // every load of a batch is in flight before the first LDS write (a plain copy loop is compiled into one
    // HBM round trip per iteration: 16 x ~2 us per panel)
    const float *base = src + p.r0 * D;   // wave-uniform; 32-bit lane offsets
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
            *reinterpret_cast<f32x4 *>(sA + row * C::LDA + c4 * 4) = v[b];
        }
    }
}

template <int D>
__device__ __forceinline__ void chain_load_acc(f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN], const float *src,
                                               const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    // one wave-uniform base + 32-bit lane offsets: the compiler then keeps a single scalar
    // pointer instead of 64 per-element 64-bit addresses (which it spilled)
    const float *base = src + p.r0 * D;
    const unsigned lane_off = (unsigned)((p.wm * C::WM + 4 * p.half) * D + p.wn * C::WN + p.l31);
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // rows past the end of the last panel re-read row 0 (their results are never stored): no per-element
                // branch, which is what a predicated load compiles to
                const int row = p.row(tm, r);
                const unsigned off = lane_off + (unsigned)((tm * 32 + (r & 3) + 8 * (r >> 2)) * D + tn * 32);
                acc[tm][tn][r] = base[row < p.R_left ? off : (unsigned)(p.wn * C::WN + p.l31 + tn * 32)];
            }
}

template <int D, int ACT>
__device__ __forceinline__ void chain_bias_act(f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN], const float *bias,
                                               const ChainPos<D> &p) {
    using C = PanelCfg<D>;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const float bv = bias[p.col(tn)];
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[tm][tn][r] + bv;
                if constexpr (ACT == 1) v = gelu_erf(v);
                acc[tm][tn][r] = v;
            }
    }
}

template <int D>
__device__ __forceinline__ void chain_zero(f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN]) {
    using C = PanelCfg<D>;
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
}

template <int D>
__device__ __forceinline__ void chain_acc_to_lds(float *sA, const f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN],
                                                 const ChainPos<D> &p) {
    using C = PanelCfg<D>;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) sA[p.row(tm, r) * C::LDA + p.col(tn)] = acc[tm][tn][r];
}

// bias + GELU applied on the way into the panel, one element at a time: the activated
// tile is never live in registers as a whole (64 erf temporaries would spill)
template <int D>
__device__ __forceinline__ void chain_gelu_to_lds(float *sA, const f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN],
                                                  const float *bias, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const float bv = bias[p.col(tn)];
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) sA[p.row(tm, r) * C::LDA + p.col(tn)] = gelu_erf(acc[tm][tn][r] + bv);
    }
}

template <int D>
__device__ __forceinline__ void chain_store_acc(float *dst, int ld, int col0,
                                                const f32x16 (&acc)[PanelCfg<D>::TM][PanelCfg<D>::TN],
                                                const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    // quad transpose -> each lane owns 4 consecutive columns of one row -> 16-byte non-temporal stores
    float *base = dst + p.r0 * ld + col0;  // wave-uniform; 32-bit lane offsets below
    const int i4 = p.lane & 3;
    const unsigned lane_off = (unsigned)((p.wm * C::WM + 4 * p.half + i4) * ld + p.wn * C::WN + (p.l31 & ~3));
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float x0 = acc[tm][tn][4 * g], x1 = acc[tm][tn][4 * g + 1], x2 = acc[tm][tn][4 * g + 2], x3 = acc[tm][tn][4 * g + 3];
                quad_transpose(x0, x1, x2, x3, p.lane);
                const int row = p.wm * C::WM + tm * 32 + 8 * g + 4 * p.half + i4;
                if (row < p.R_left) {
                    const f32x4 v = {x0, x1, x2, x3};
                    SD_NT_STORE(v, reinterpret_cast<f32x4 *>(base + lane_off + (unsigned)((tm * 32 + 8 * g) * ld + tn * 32)));
                }
            }
}

// LayerNorm of the LDS panel in place (two-pass, fp32).  A wave normalises 4 rows at a time:
// 16 lanes per row (one DPP row), each lane D/16 values in 16-byte pieces, so the row sums
// are in-lane adds + 4 DPP rotations, and 4 independent rows are in flight per wave.
template <int D>
__device__ __forceinline__ void panel_layer_norm(float *sA, int lda, const float *ln_w, const float *ln_b, int lane, int wave,
                                                 int rows) {
    constexpr int V4 = D / 64;  // 16-byte pieces per lane: lane i of a row owns columns 4*(i + 16*j) .. +3
    const int sub = lane & 15, grp = lane >> 4;
    for (int row = wave * 4 + grp; row < rows; row += 16) {
        f32x4 v[V4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            v[j] = *reinterpret_cast<const f32x4 *>(sA + row * lda + 4 * (sub + 16 * j));
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mean = row16_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < V4; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[j][e] -= mean;
                q += v[j][e] * v[j][e];
            }
        }
        const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / D) + SD_LN_EPS);
#pragma unroll
        for (int j = 0; j < V4; ++j) {
            const int c = 4 * (sub + 16 * j);
            const f32x4 gw = *reinterpret_cast<const f32x4 *>(ln_w + c);
            const f32x4 gb = *reinterpret_cast<const f32x4 *>(ln_b + c);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = v[j][e] * rstd * gw[e] + gb[e];
            *reinterpret_cast<f32x4 *>(sA + row * lda + c) = y;
        }
    }
}

template <int D>
__device__ __forceinline__ void chain_layer_norm(float *sA, const float *ln_w, const float *ln_b, int lane, int wave) {
    panel_layer_norm<D>(sA, PanelCfg<D>::LDA, ln_w, ln_b, lane, wave, PanelCfg<D>::BM);
}

struct ChainAArgs {
    const float *a;              // attention output rows [R, D]
    float *h;                    // residual stream [R, D] in/out
    const float *wo, *bo;        // out projection
    const float *ln_w, *ln_b;    // norm2
    const float *wq, *bq;        // cross-attention Q projection (rows [0:D) of in_proj)
    float *q;                    // [R, D]
    long R;
};

template <int D>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void chain_a_kernel(ChainAArgs g) {
    using C = PanelCfg<D>;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const ChainPos<D> p(g.R);
    const float *aBase = sA + (p.wm * C::WM + p.l31) * C::LDA + 4 * p.half;
    f32x16 H[C::TM][C::TN];
    chain_load_acc<D>(H, g.h, p);   // issued first: in flight while the panel lands
    chain_load_panel<D>(sA, g.a, p);
    __syncthreads();
    chain_gemm<D>(H, aBase, g.wo + (long)(p.wn * C::WN + p.l31) * D + 4 * p.half);
    chain_bias_act<D, 0>(H, g.bo, p);
    chain_store_acc<D>(g.h, D, 0, H, p);
    __syncthreads();  // every wave is done reading the panel
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    chain_layer_norm<D>(sA, g.ln_w, g.ln_b, p.lane, p.wave);
    __syncthreads();
    chain_zero<D>(H);
    chain_gemm<D>(H, aBase, g.wq + (long)(p.wn * C::WN + p.l31) * D + 4 * p.half);
    chain_bias_act<D, 0>(H, g.bq, p);
    chain_store_acc<D>(g.q, D, 0, H, p);
}

struct ChainBArgs {
    const float *a;              // attention output rows [R, D]
    float *h;                    // residual stream [R, D] in/out
    const float *wo, *bo;        // out projection of that attention
    const float *ln_w, *ln_b;    // FFN norm (norm3 decoder / norm2 encoder)
    const float *w1, *b1, *w2, *b2;
    const float *nln_w, *nln_b;  // next layer's norm1 (NULL: no next layer)
    const float *wqkv, *bqkv;    // next layer's in_proj (3D, D)
    float *qkv;                  // [R, 3D]
    long R;
};

template <int D>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void chain_b_kernel(ChainBArgs g) {
    using C = PanelCfg<D>;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const ChainPos<D> p(g.R);
    const float *aBase = sA + (p.wm * C::WM + p.l31) * C::LDA + 4 * p.half;
    const long wOff = (long)(p.wn * C::WN + p.l31) * D + 4 * p.half;
    f32x16 H[C::TM][C::TN], U[C::TM][C::TN];
    chain_load_acc<D>(H, g.h, p);
    chain_load_panel<D>(sA, g.a, p);
    __syncthreads();
    chain_gemm<D>(H, aBase, g.wo + wOff);          // h += a Wo^T
    chain_bias_act<D, 0>(H, g.bo, p);
    __syncthreads();
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    chain_layer_norm<D>(sA, g.ln_w, g.ln_b, p.lane, p.wave);
    __syncthreads();
    chain_zero<D>(U);
    chain_gemm<D>(U, aBase, g.w1 + wOff);          // u = gelu(LN(h) W1^T + b1)
    __syncthreads();
    chain_gelu_to_lds<D>(sA, U, g.b1, p);
    __syncthreads();
    chain_gemm<D>(H, aBase, g.w2 + wOff);          // h += u W2^T + b2
    chain_bias_act<D, 0>(H, g.b2, p);
    chain_store_acc<D>(g.h, D, 0, H, p);
    if (g.nln_w == nullptr) return;
    __syncthreads();
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    chain_layer_norm<D>(sA, g.nln_w, g.nln_b, p.lane, p.wave);
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {          // next layer's q | k | v
        chain_zero<D>(U);
        chain_gemm<D>(U, aBase, g.wqkv + (long)pass * D * D + wOff);
        chain_bias_act<D, 0>(U, g.bqkv + pass * D, p);
        chain_store_acc<D>(g.qkv, 3 * D, pass * D, U, p);
    }
}

// --------------------------------------------------------------------------------------
// Whole decoder layer tail in ONE kernel: chain A, the cross-attention core and chain B.
//
// The cross-attention of a row only needs that row's query and the (few) projected memory
// rows of its own trajectory, so it is row-local too.  With 4 heads and D/4 = head dim the
// head split coincides with the wave split of the panel GEMMs: wave w owns head w.  Q goes
// through the LDS panel (the accumulator has features on lanes; S^T = K Q^T needs queries on
// lanes), K/V fragments of the <= 64 keys of the panel's trajectories come straight from
// L2, keys of other trajectories are masked to -inf, and the head's output overwrites its
// own Q columns in the panel, which is then the A operand of the out-projection.
// Used when (trajectories per panel) x (memory rows) <= 64; otherwise the host falls back
// to chain A + attention_kernel + chain B.
// --------------------------------------------------------------------------------------
// Diagnostic build only (-DSD_STAMPS, tools/stamps.py): every wave records the shader clock at phase
// boundaries of decoder_layer_kernel into a host-provided buffer [layer][workgroup][wave][32].
#ifdef SD_STAMPS
__device__ unsigned long long *g_stamp_buf = nullptr;
__device__ long g_stamp_wgs = 0;
#define SD_STAMP(slot, i)                                                                                              \
    do {                                                                                                               \
        if ((threadIdx.x & 63) == 0 && g_stamp_buf && blockIdx.x < g_stamp_wgs)                                        \
            g_stamp_buf[(((long)(slot) * g_stamp_wgs + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 32 + (i)] =             \
                __builtin_amdgcn_s_memtime();                                                                          \
    } while (0)
extern "C" int sd_debug_set_stamps(void *buf, long wgs) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_wgs), &wgs, sizeof(wgs));
    return (int)e;
}
#else
#define SD_STAMP(slot, i)
#endif
#define SD_STAMP_ATT_SLOT 5    /* attention_f16_kernel, unit 2 of the last launch (stamp 8: same point one unit later) */
#define SD_STAMP_HEAD_SLOT 4   /* diagnostic builds, L = 4: the head kernel's stamps go behind the layers' */

struct DecoderLayerArgs {
    ChainAArgs a;      // a.q unused
    ChainBArgs b;      // b.a / b.h unused (same panel)
    const float *kv;   // [B*Mk, 2D] projected memory keys | values of this layer, Mk rows per trajectory
    int T, Mk, B;
    float scale_log2e;
    // last layer only (b.nln_w == NULL): fc_out (+ DDIM update) fused behind the chain; fo_w NULL = store h instead
    const float *fo_w, *fo_b;
    float *eps, *x_io;
    float c0, c1, c2, c3;
    int J;
    // folded cross-attention (see panel_folded_scores): per trajectory 64 rows [head][16 key slots] of
    // (K_h Wq_h | V_h Wo_h^T), 2D floats each, and the score bias bq_h . K_h[key]; NULL = unfolded path
    const float *gv, *cb;
    int slot;   // layer index (stamp builds)
};

template <int D, int NKT>
__device__ __forceinline__ void panel_cross_attention(float *sA, const DecoderLayerArgs &g, const ChainPos<D> &p) {
    using C = PanelCfg<D>;
    constexpr int HD = D / 4, KS = HD / 8, FT = (HD + 31) / 32;
    const int h = __builtin_amdgcn_readfirstlane(p.wave);  // one head per wave (made provably wave-uniform)
    const int Mk = g.Mk;
    const long b0 = p.r0 / g.T;
    const int n_traj = (int)((p.r0 + p.R_left - 1) / g.T - b0) + 1;
    const int n_keys = n_traj * Mk;  // <= 32 * NKT
    // Memory rows of consecutive trajectories are consecutive in kv, so panel key number
    // `key` (= local trajectory * Mk + m) is simply row b0*Mk + key: one base + 32-bit offsets.
    // kvh is wave-uniform (scalar base); every access below is kvh[32-bit lane offset]
    const float *kvh = g.kv + b0 * Mk * 2 * D + h * HD;
#pragma unroll 1
    for (int tq = 0; tq < 2; ++tq) {
        const int qrow = tq * 32 + p.l31;
        const bool q_ok = qrow < p.R_left;
        const int key_lo = (int)((p.r0 + qrow) / g.T - b0) * Mk;  // this query's own trajectory: keys [key_lo, key_lo + Mk)
        float *qbase = sA + qrow * C::LDA + h * HD;
        const float *qfrag = qbase + 4 * p.half;  // Q fragments are re-read from the panel per k-step (LDS is cheap,
                                                  // 32 registers are not: the K fragments of a tile are held instead)
        f32x16 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
            const int key = kt * 32 + p.l31;  // A-fragment row of this lane
            const int koff = (key < n_keys ? key : 0) * 2 * D + 4 * p.half;
            const float km = key < n_keys ? 1.f : 0.f;
            // all K fragments of the tile are requested before the first MFMA (one L2 round trip per
            // tile instead of one per k-step: hipcc otherwise waits vmcnt(0) after every load)
            f32x4 kf[KS];
#pragma unroll
            for (int st = 0; st < KS; ++st) kf[st] = *reinterpret_cast<const f32x4 *>(kvh + koff + st * 8);
            f32x4 qv[2];
            qv[0] = *reinterpret_cast<const f32x4 *>(qfrag);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                if (st + 1 < KS) qv[(st + 1) & 1] = *reinterpret_cast<const f32x4 *>(qfrag + (st + 1) * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[st][j] * km, qv[st & 1][j], sc[kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the next tile's loads from being hoisted (registers)
        }
        // keys of other trajectories (and padding) -> -inf; softmax per query (lane column)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * p.half;
                const bool ok = q_ok && key >= key_lo && key < key_lo + Mk;
                const float v = ok ? sc[kt][r] : -INFINITY;
                sc[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx == -INFINITY) mx = 0.f;  // padded query row: every p becomes 0
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f((sc[kt][r] - mx) * g.scale_log2e);
                sc[kt][r] = pv;
                psum += pv;
            }
        psum += __shfl_xor(psum, 32, 64);
        const float inv = psum > 0.f ? 1.0f / psum : 0.f;
        // O^T = V^T P^T : A = V[key][h*HD + ft*32 + l31] from L2, B = P^T from the accumulator
        f32x16 o[FT];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ft][r] = 0.f;
        const int vlane = D + p.l31;  // V half of the row, this lane's feature
        const float fm = 1.f;          // HD >= 32 here (D >= 128): every feature lane is live
        // V operands: the 4 x FT values of the NEXT 8-key group are requested before the current
        // group's MFMAs (pinned), so one L2 latency is exposed per tile pass instead of one per MFMA
        auto load_v = [&](int gg, float (&dst)[4][FT]) {
            const int kt = gg >> 2, gq = gg & 3;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int key = kt * 32 + ri + 8 * gq + 4 * p.half;
                const int voff = (key < n_keys ? key : 0) * 2 * D + vlane;
                const float vm = key < n_keys ? fm : 0.f;
#pragma unroll
                for (int ft = 0; ft < FT; ++ft) dst[ri][ft] = kvh[voff + ft * 32] * vm;
            }
        };
        float av[2][4][FT];
        load_v(0, av[0]);
#pragma unroll
        for (int gg = 0; gg < NKT * 4; ++gg) {
            const int kt = gg >> 2, gq = gg & 3;
            const bool live = kt * 32 + 8 * gq < n_keys;           // wave-uniform
            const bool next_live = gg + 1 < NKT * 4 && ((gg + 1) >> 2) * 32 + 8 * ((gg + 1) & 3) < n_keys;
            if (next_live) load_v(gg + 1, av[(gg + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (live) {
#pragma unroll
                for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                    for (int ft = 0; ft < FT; ++ft)
                        o[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gg & 1][ri][ft], sc[kt][4 * gq + ri], o[ft], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // this head's output replaces its own Q columns of the panel (only this wave reads them)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int f = ft * 32 + 8 * gq + 4 * p.half;
                if (f < HD) {
                    f32x4 t = {o[ft][4 * gq] * inv, o[ft][4 * gq + 1] * inv, o[ft][4 * gq + 2] * inv, o[ft][4 * gq + 3] * inv};
                    *reinterpret_cast<f32x4 *>(qbase + f) = t;
                }
            }
    }
}

// --------------------------------------------------------------------------------------
// Folded cross-attention (sampling: the memory is fixed over the rollout, so everything that
// depends only on memory and weights is computed once, by xattn_fold_kernel):
//     scores_h = (LN2(h) Wq_h^T + bq_h) K_h^T  = LN2(h) (K_h Wq_h)^T + bq_h . K_h      =: LN2(h) G_h^T + c_h
//     out      = sum_h P_h V_h Woc_h^T + boc    = [P_0 .. P_3] [V_0 Woc_0^T ; .. ]      =: P V'
// so the Q projection (2 T d^2) and the out projection (2 T d^2) of the layer disappear; what is
// left is a (64 x D) x (D x 128) and a (64 x 128) x (128 x D) product per panel, 128 = 2 trajectories
// x 4 heads x 16 key slots.  Row k = tl*64 + head*16 + key of the per-trajectory gv block is both
// the G row (first D floats) and the V' row (last D floats) of that key, so the 128 rows seen by
// a panel are consecutive in memory from its first trajectory on.
// --------------------------------------------------------------------------------------
constexpr int FOLD_RING = 8;   // k-steps of G in flight (first touch comes from HBM, not L2)
constexpr int FOLD_VRING = 4;

template <int D>
struct FoldState {
    f32x4 g[FOLD_RING];
    f32x4 c[4];
    const float *gbase;   // wave-uniform: first gv row of the panel's first trajectory
    unsigned goff;        // this lane's G row (A operand row = key slot l31 of head `wave`)
    int n_traj;
};

template <int D>
__device__ __forceinline__ void fold_prime(FoldState<D> &f, const DecoderLayerArgs &g, const ChainPos<D> &p) {
    const int h = __builtin_amdgcn_readfirstlane(p.wave);
    const long b0 = p.r0 / g.T;
    f.n_traj = (int)((p.r0 + p.R_left - 1) / g.T - b0) + 1;
    f.gbase = g.gv + b0 * 64 * 2 * D;
    const int tl = (f.n_traj > 1) ? (p.l31 >> 4) : 0;   // a lone trajectory: the other half tile re-reads it (masked)
    f.goff = (unsigned)((tl * 64 + h * 16 + (p.l31 & 15)) * 2 * D + 4 * p.half);
#pragma unroll
    for (int s = 0; s < FOLD_RING - 1; ++s) f.g[s] = *reinterpret_cast<const f32x4 *>(f.gbase + f.goff + s * 8);
    const float *cbase = g.cb + b0 * 64 + h * 16 + 4 * p.half;
#pragma unroll
    for (int q = 0; q < 4; ++q)   // score-bias of accumulator rows 4q..4q+3: key slot 8q + 4*half + i
        f.c[q] = *reinterpret_cast<const f32x4 *>(cbase + ((f.n_traj > 1) ? (q >> 1) * 64 : 0) + (q & 1) * 8);
}

// S^T (32 key slots x 64 queries) of head `wave`; then softmax per query and P -> panel columns
template <int D>
__device__ __forceinline__ void panel_folded_scores(float *sA, FoldState<D> &f, const DecoderLayerArgs &g,
                                                    const ChainPos<D> &p, float (&vr)[FOLD_VRING][4][PanelCfg<D>::TN],
                                                    unsigned voff) {
    using C = PanelCfg<D>;
    constexpr int NK = D / 8;
    const int h = __builtin_amdgcn_readfirstlane(p.wave);
    const float *hB = sA + p.l31 * C::LDA + 4 * p.half;
    f32x16 sc[2];
#pragma unroll
    for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[tq][r] = 0.f;
    f32x4 hf[2][2];
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) hf[0][tq] = *reinterpret_cast<const f32x4 *>(hB + tq * 32 * C::LDA);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int cur = ks % FOLD_RING, fill = (ks + FOLD_RING - 1) % FOLD_RING;
        if (ks + FOLD_RING - 1 < NK)
            f.g[fill] = *reinterpret_cast<const f32x4 *>(f.gbase + f.goff + (ks + FOLD_RING - 1) * 8);
        if (ks + 1 < NK) {
#pragma unroll
            for (int tq = 0; tq < 2; ++tq)
                hf[(ks + 1) & 1][tq] = *reinterpret_cast<const f32x4 *>(hB + tq * 32 * C::LDA + (ks + 1) * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tq = 0; tq < 2; ++tq)
                sc[tq] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.g[cur][j], hf[ks & 1][tq][j], sc[tq], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    SD_STAMP(g.slot, 4);
    // first V' rows in flight while the softmax runs (they come from HBM on first touch)
#pragma unroll
    for (int s = 0; s < FOLD_VRING - 1; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) vr[s][j][tn] = f.gbase[voff + (unsigned)((s * 8 + j) * 2 * D + tn * 32)];
    __syncthreads();   // every wave has read LN2(h): the panel now receives P
    const long b0 = p.r0 / g.T;
#pragma unroll
    for (int tq = 0; tq < 2; ++tq) {
        const int qrow = tq * 32 + p.l31;
        const bool q_ok = qrow < p.R_left;
        const int key_lo = (int)((p.r0 + qrow) / g.T - b0) * 16;   // own trajectory: slots [key_lo, key_lo + Mk)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = (r & 3) + 8 * (r >> 2) + 4 * p.half;
            const bool ok = q_ok && kk >= key_lo && kk < key_lo + g.Mk;
            const float v = ok ? sc[tq][r] + f.c[r >> 2][r & 3] : -INFINITY;
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
        const float inv = psum > 0.f ? 1.0f / psum : 0.f;
        // P[query][k], k = tl*64 + head*16 + key: accumulator rows 4q..4q+3 are slots 8q + 4*half + 0..3
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = {sc[tq][4 * q] * inv, sc[tq][4 * q + 1] * inv, sc[tq][4 * q + 2] * inv, sc[tq][4 * q + 3] * inv};
            *reinterpret_cast<f32x4 *>(sA + qrow * C::LDA + (q >> 1) * 64 + h * 16 + (q & 1) * 8 + 4 * p.half) = t;
        }
    }
}

// H += P V'   (K = 64 per trajectory touched by the panel)
template <int D>
__device__ __forceinline__ void panel_folded_pv(f32x16 (&H)[PanelCfg<D>::TM][PanelCfg<D>::TN], const float *aBase,
                                                const float *gbase, unsigned voff, int n_traj,
                                                float (&vr)[FOLD_VRING][4][PanelCfg<D>::TN]) {
    using C = PanelCfg<D>;
    f32x4 af[2][C::TM];
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) af[0][tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * C::LDA);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        if (part == 1 && n_traj < 2) break;   // workgroup-uniform
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const int ks = part * 8 + k8;
            const int cur = ks % FOLD_VRING, fill = (ks + FOLD_VRING - 1) % FOLD_VRING;
            const bool more = (ks + FOLD_VRING - 1 < 8) || (n_traj > 1 && ks + FOLD_VRING - 1 < 16);
            if (more) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn)
                        vr[fill][j][tn] = gbase[voff + (unsigned)(((ks + FOLD_VRING - 1) * 8 + j) * 2 * D + tn * 32)];
            }
            if (ks + 1 < 16) {
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
                    af[(ks + 1) & 1][tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * C::LDA + (ks + 1) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn)
                        H[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][tm][j], vr[cur][j][tn], H[tm][tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// fc_out (d -> J <= 64) + DDIM update on the panel that holds the final h (see fc_out_kernel)
// x_lds (optional): the updated x rows are also left in LDS at x_lds[row * x_pitch + j] (zero elsewhere up to column 64),
// for a kernel that goes on with the next step's embedding
template <int D>
__device__ __forceinline__ void panel_fc_out(float *sA, const DecoderLayerArgs &g, const ChainPos<D> &p, float *x_lds = nullptr,
                                             int x_pitch = 0) {
    using C = PanelCfg<D>;
    constexpr int KH = D / 2;
    const int tm = p.wave & 1, kh = p.wave >> 1;
    const float *aB = sA + (tm * 32 + p.l31) * C::LDA + kh * KH + 4 * p.half;
    const int J = g.J, n_tiles = (J + 31) / 32;
    f32x16 acc[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
    for (int tn = 0; tn < n_tiles; ++tn) {
        const int j = tn * 32 + p.l31;
        const float *wp = g.fo_w + (long)(j < J ? j : 0) * D + kh * KH + 4 * p.half;
        const float wmask = j < J ? 1.f : 0.f;
        // the weight fragments are requested 16 k-steps at a time (one L2 round trip per batch, not per k-step)
        constexpr int KB = 16;
#pragma unroll
        for (int kb = 0; kb < KH / 8; kb += KB) {
            f32x4 bf[KB];
#pragma unroll
            for (int s = 0; s < KB; ++s)
                if (kb + s < KH / 8) bf[s] = *reinterpret_cast<const f32x4 *>(wp + (kb + s) * 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KB; ++s) {
                if (kb + s >= KH / 8) break;
                const f32x4 af = *reinterpret_cast<const f32x4 *>(aB + (kb + s) * 8);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (tn == 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj], bf[s][jj] * wmask, acc[0], 0, 0, 0);
                    else acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj], bf[s][jj] * wmask, acc[1], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();  // panel consumed: reuse it for the K-half exchange
    if (x_lds)
        for (int i = threadIdx.x; i < C::BM * x_pitch; i += 256) x_lds[i] = 0.f;
    if (kh == 1) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) sA[((tm * 2 + tn) * 16 + r) * 64 + p.lane] = acc[tn][r];
    }
    __syncthreads();
    if (kh == 0) {
        for (int tn = 0; tn < n_tiles; ++tn) {
            const int j = tn * 32 + p.l31;
            if (j >= J) continue;
            const float bv = g.fo_b[j];
            // all 16 x values are requested before the first is used (a load-update-store loop costs an HBM
            // round trip per element)
            float xv[16];
            float *xb = g.x_io ? g.x_io + p.r0 * J : nullptr;
            if (xb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * p.half;
                    xv[r] = row < p.R_left ? xb[(unsigned)(row * J + j)] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * p.half;
                if (row >= p.R_left) continue;
                const float e = (tn == 0 ? acc[0][r] : acc[1][r]) + sA[((tm * 2 + tn) * 16 + r) * 64 + p.lane] + bv;
                if (g.eps) g.eps[(p.r0 + row) * J + j] = e;
                if (xb) {
                    const float x0 = (xv[r] - g.c1 * e) / g.c0;
                    const float xn = g.c2 * x0 + g.c3 * e;
                    xb[(unsigned)(row * J + j)] = xn;
                    if (x_lds) x_lds[row * x_pitch + j] = xn;
                }
            }
        }
    }
}

// TAIL = true is the last layer's instantiation (no next-layer QKV; fc_out + DDIM fused behind
// the chain).  Separate instantiations keep the tail's registers out of the common kernel.
template <int D, bool TAIL, bool FOLD>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void decoder_layer_kernel(DecoderLayerArgs g) {
    using C = PanelCfg<D>;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const ChainPos<D> p(g.a.R);
    const float *aBase = sA + (p.wm * C::WM + p.l31) * C::LDA + 4 * p.half;
    const long wOff = (long)(p.wn * C::WN + p.l31) * D + 4 * p.half;
    f32x16 H[C::TM][C::TN], U[C::TM][C::TN];
    WeightRing<D> ring;
    SD_STAMP(g.slot, 0);
#ifdef SD_STAMPS
    if ((threadIdx.x & 63) == 0 && g_stamp_buf) {   // where this workgroup runs: HW_ID (cu/sh/se) and XCC_ID
        unsigned long long *sb = g_stamp_buf + (((long)g.slot * g_stamp_wgs + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 32;
        sb[30] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        sb[31] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    chain_prime<D>(ring, g.a.wo + wOff);           // first weights in flight while the panel lands
    chain_load_acc<D>(H, g.a.h, p);
    chain_load_panel<D>(sA, g.a.a, p);
    __syncthreads();
    SD_STAMP(g.slot, 1);
    chain_gemm_primed<D>(H, aBase, g.a.wo + wOff, ring);   // h += a Wo^T + bo   (self-attention out)
    SD_STAMP(g.slot, 2);
    if constexpr (FOLD) {
        FoldState<D> fs;
        fold_prime<D>(fs, g, p);          // G rows + score bias in flight behind the epilogue and LN2
        chain_bias_act<D, 0>(H, g.a.bo, p);
        __syncthreads();
        chain_acc_to_lds<D>(sA, H, p);
        __syncthreads();
        chain_layer_norm<D>(sA, g.a.ln_w, g.a.ln_b, p.lane, p.wave);
        __syncthreads();
        SD_STAMP(g.slot, 3);
        float vr[FOLD_VRING][4][C::TN];
        const unsigned voff = (unsigned)(4 * p.half * 2 * D + D + p.wn * C::WN + p.l31);
        panel_folded_scores<D>(sA, fs, g, p, vr, voff);
        __syncthreads();
        SD_STAMP(g.slot, 5);
        panel_folded_pv<D>(H, aBase, fs.gbase, voff, fs.n_traj, vr);   // h += P V' + boc
        SD_STAMP(g.slot, 6);
    } else {
        chain_prime<D>(ring, g.a.wq + wOff);
        chain_bias_act<D, 0>(H, g.a.bo, p);
        __syncthreads();
        chain_acc_to_lds<D>(sA, H, p);
        __syncthreads();
        chain_layer_norm<D>(sA, g.a.ln_w, g.a.ln_b, p.lane, p.wave);
        __syncthreads();
        chain_zero<D>(U);
        chain_gemm_primed<D>(U, aBase, g.a.wq + wOff, ring);   // q = LN2(h) Wq^T + bq
        chain_bias_act<D, 0>(U, g.a.bq, p);
        __syncthreads();
        chain_acc_to_lds<D>(sA, U, p);
        __syncthreads();
        {   // a_c over the memory of each row's trajectory (wave-uniform choice of 1 or 2 key tiles)
            const int n_traj = (int)((p.r0 + p.R_left - 1) / g.T - p.r0 / g.T) + 1;
            if (n_traj * g.Mk <= 32) panel_cross_attention<D, 1>(sA, g, p);
            else panel_cross_attention<D, 2>(sA, g, p);
        }
        chain_prime<D>(ring, g.b.wo + wOff);   // primed after the attention: its 24 registers are needed there
        __syncthreads();
        chain_gemm_primed<D>(H, aBase, g.b.wo + wOff, ring);   // h += a_c Woc^T + boc
    }
    chain_prime<D>(ring, g.b.w1 + wOff);
    chain_bias_act<D, 0>(H, g.b.bo, p);
    __syncthreads();
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    chain_layer_norm<D>(sA, g.b.ln_w, g.b.ln_b, p.lane, p.wave);
    __syncthreads();
    SD_STAMP(g.slot, 7);
    chain_zero<D>(U);
    chain_gemm_primed<D>(U, aBase, g.b.w1 + wOff, ring);   // u = gelu(LN3(h) W1^T + b1)
    SD_STAMP(g.slot, 8);
    chain_prime<D>(ring, g.b.w2 + wOff);
    __syncthreads();
    chain_gelu_to_lds<D>(sA, U, g.b.b1, p);
    __syncthreads();
    SD_STAMP(g.slot, 9);
    chain_gemm_primed<D>(H, aBase, g.b.w2 + wOff, ring);   // h += u W2^T + b2
    SD_STAMP(g.slot, 10);
    if constexpr (TAIL) {
        chain_bias_act<D, 0>(H, g.b.b2, p);
        __syncthreads();                 // last layer: eps = h Wout^T + b (+ DDIM update of x) right here
        chain_acc_to_lds<D>(sA, H, p);
        __syncthreads();
        SD_STAMP(g.slot, 11);
        panel_fc_out<D>(sA, g, p);
        SD_STAMP(g.slot, 12);
        return;
    }
    const bool has_next = g.b.nln_w != nullptr;            // workgroup-uniform
    if (has_next) chain_prime<D>(ring, g.b.wqkv + wOff);
    chain_bias_act<D, 0>(H, g.b.b2, p);
    chain_store_acc<D>(g.a.h, D, 0, H, p);
    if (!has_next) return;
    SD_STAMP(g.slot, 11);
    __syncthreads();
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    chain_layer_norm<D>(sA, g.b.nln_w, g.b.nln_b, p.lane, p.wave);
    __syncthreads();
    SD_STAMP(g.slot, 12);
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {          // next layer's q | k | v
        chain_zero<D>(U);
        chain_gemm_primed<D>(U, aBase, g.b.wqkv + (long)pass * D * D + wOff, ring);
        SD_STAMP(g.slot, 13 + 2 * pass);
        if (pass < 2) chain_prime<D>(ring, g.b.wqkv + (long)(pass + 1) * D * D + wOff);  // older than this pass's stores
        chain_bias_act<D, 0>(U, g.b.bqkv + pass * D, p);
        chain_store_acc<D>(g.b.qkv, 3 * D, pass * D, U, p);
        SD_STAMP(g.slot, 14 + 2 * pass);
    }
}

// --------------------------------------------------------------------------------------
// Head of a denoiser step in one launch: h = x Wemb^T + b + pe (nn.Linear(J -> d) on the MFMA,
// K = J zero-padded to a multiple of 8), h stored, then qkv = LN1(h) Wqkv^T + b of layer 0.
// Needs J % 4 == 0 (16-byte weight fragments); otherwise patch_embed + panel_gemm are used.
// --------------------------------------------------------------------------------------
struct DecoderHeadArgs {
    const float *x;               // [R, J]
    const float *emb_w, *emb_b;   // (D, J), (D)
    const float *pe;              // [T_max, D]
    const float *ln_w, *ln_b;     // layer 0 norm1
    const float *wqkv, *bqkv;     // layer 0 in_proj (3D, D)
    float *h, *qkv;               // [R, D], [R, 3D]
    long R;
    int T, J;
};

template <int D>
__global__ __launch_bounds__(256, (D <= 256 ? 2 : 1)) void decoder_head_kernel(DecoderHeadArgs g) {
    using C = PanelCfg<D>;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const ChainPos<D> p(g.R);
    const float *aBase = sA + (p.wm * C::WM + p.l31) * C::LDA + 4 * p.half;
    const long wOff = (long)(p.wn * C::WN + p.l31) * D + 4 * p.half;
    WeightRing<D> ring;
    chain_prime<D>(ring, g.wqkv + wOff);
    const int J = g.J, Jp = (J + 7) & ~7;  // <= 64 < LDA
    for (int i = threadIdx.x; i < C::BM * Jp; i += 256) {
        const int row = i / Jp, j = i - row * Jp;
        sA[row * C::LDA + j] = (row < p.R_left && j < J) ? g.x[(p.r0 + row) * J + j] : 0.f;
    }
    __syncthreads();
    f32x16 H[C::TM][C::TN];
    chain_zero<D>(H);
    for (int k0 = 0; k0 < Jp; k0 += 8) {
        const int kk = k0 + 4 * p.half;
        f32x4 bf[C::TN], af[C::TM];
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (kk < J) t = *reinterpret_cast<const f32x4 *>(g.emb_w + (long)(p.wn * C::WN + tn * 32 + p.l31) * J + kk);
            bf[tn] = t;
        }
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) af[tm] = *reinterpret_cast<const f32x4 *>(aBase + tm * 32 * C::LDA + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
                    H[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm][j], bf[tn][j], H[tm][tn], 0, 0, 0);
    }
    // + bias + positional row (position = row index inside its trajectory)
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const int col = p.col(tn);
        const float bv = g.emb_b[col];
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = p.row(tm, r);
                const int pos = (int)((p.r0 + (row < p.R_left ? row : 0)) % g.T);
                H[tm][tn][r] += bv + g.pe[(long)pos * D + col];
            }
    }
    chain_store_acc<D>(g.h, D, 0, H, p);
    __syncthreads();  // the x staging area is part of the panel
    chain_acc_to_lds<D>(sA, H, p);
    __syncthreads();
    chain_layer_norm<D>(sA, g.ln_w, g.ln_b, p.lane, p.wave);
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        chain_zero<D>(H);
        chain_gemm_primed<D>(H, aBase, g.wqkv + (long)pass * D * D + wOff, ring);
        if (pass < 2) chain_prime<D>(ring, g.wqkv + (long)(pass + 1) * D * D + wOff);
        chain_bias_act<D, 0>(H, g.bqkv + pass * D, p);
        chain_store_acc<D>(g.qkv, 3 * D, pass * D, H, p);
    }
}

template <typename Args, typename KA, typename KB, typename KC, typename KD>
static int launch_chain(const Args &g, int d, KA k64, KB k128, KC k256, KD k512, const char *name, hipStream_t s,
                        int kclass = SD_KCLASS_LAYER_CHAIN) {
    if (g.R <= 0) return fail(SD_E_BADARG, "chain: empty shape");
    ProfScope prof(kclass, s);
    dim3 grid((unsigned)((g.R + 63) / 64)), block(256);
#define SD_CHAIN(D_, K_)                                                                                        \
    do {                                                                                                        \
        const size_t lds = PanelCfg<D_>::LDS_BYTES;                                                             \
        static DevFlag attr_set;                                                                                  \
        if (lds > 64 * 1024 && !attr_set) {                                                                     \
            (void)hipFuncSetAttribute((const void *)K_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        SD_LAUNCH(K_, grid, block, lds, s, g);                                                                  \
    } while (0)
    switch (d) {
        case 64: SD_CHAIN(64, k64); break;
        case 128: SD_CHAIN(128, k128); break;
        case 256: SD_CHAIN(256, k256); break;
        case 512: SD_CHAIN(512, k512); break;
        default: return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
    }
#undef SD_CHAIN
    SD_CHECK_LAUNCH(name);
    return 0;
}

static int chain_a(const ChainAArgs &g, int d, hipStream_t s) {
    return launch_chain(g, d, chain_a_kernel<64>, chain_a_kernel<128>, chain_a_kernel<256>, chain_a_kernel<512>,
                        "chain_a_kernel", s);
}
static int chain_b(const ChainBArgs &g, int d, hipStream_t s) {
    return launch_chain(g, d, chain_b_kernel<64>, chain_b_kernel<128>, chain_b_kernel<256>, chain_b_kernel<512>,
                        "chain_b_kernel", s);
}

#include "sd_f16x3.h"
#include "sd_traj.h"
#include "sd_trajg.h"

template <int D>
static int launch_panel16(const float *A, int lda, const float *W, const float *bias, const float *ln_w, const float *ln_b,
                          const float *res, float *out, int R, int N, int act, hipStream_t s, const DropoutArgs &da = DropoutArgs{}) {
    using C = PanelCfg<D>;
    ProfScope prof(SD_KCLASS_PANEL_GEMM, s);
    dim3 grid((R + C::BM - 1) / C::BM), block(256);
    const size_t lds = C::LDS_BYTES + C::BM * sizeof(float);
    if (da.thresh) {   // training: out = res + dropout(A W^T + bias)
        if (ln_w || act != 0 || !res || N % 4 != 0) return fail(SD_E_BADARG, "sd_op_linear_dropout: needs res, no LayerNorm, no activation, N % 4 == 0");
        auto kfn = panel_gemm16_kernel<D, false, 0, true, true>;
        static DevFlag attr_set;
        if (lds > 64 * 1024 && !attr_set) {
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        SD_LAUNCH(kfn, grid, block, lds, s, A, W, bias, ln_w, ln_b, res, out, R, N, lda, da);
        SD_CHECK_LAUNCH("panel_gemm16_kernel");
        return 0;
    }
#define SD_PANEL16(LN_, ACT_, RES_)                                                                            \
    do {                                                                                                       \
        auto kfn = panel_gemm16_kernel<D, LN_, ACT_, RES_>;                                                    \
        static DevFlag attr_set;                                                                                 \
        if (lds > 64 * 1024 && !attr_set) {                                                                    \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                   \
        }                                                                                                      \
        SD_LAUNCH(kfn, grid, block, lds, s, A, W, bias, ln_w, ln_b, res, out, R, N, lda, DropoutArgs{});       \
    } while (0)
    const bool ln = ln_w != nullptr;
    const bool rs = res != nullptr;
    if (ln && act == 0 && !rs) SD_PANEL16(true, 0, false);
    else if (ln && act == 1 && !rs) SD_PANEL16(true, 1, false);
    else if (!ln && act == 0 && !rs) SD_PANEL16(false, 0, false);
    else if (!ln && act == 0 && rs) SD_PANEL16(false, 0, true);
    else if (!ln && act == 1 && !rs) SD_PANEL16(false, 1, false);
    else if (ln && act == 0 && rs) SD_PANEL16(true, 0, true);
    else return fail(SD_E_BADARG, "sd_op_linear: unsupported LN/act/res combination");
#undef SD_PANEL16
    SD_CHECK_LAUNCH("panel_gemm16_kernel");
    return 0;
}

int linear16(const float *A, const float *W, const float *bias, const float *ln_w, const float *ln_b, const float *res,
             float *out, int R, int N, int d, int act, hipStream_t s, int lda) {
    switch (d) {
        case 64: return launch_panel16<64>(A, lda, W, bias, ln_w, ln_b, res, out, R, N, act, s);
        case 128: return launch_panel16<128>(A, lda, W, bias, ln_w, ln_b, res, out, R, N, act, s);
        case 256: return launch_panel16<256>(A, lda, W, bias, ln_w, ln_b, res, out, R, N, act, s);
        case 512: return launch_panel16<512>(A, lda, W, bias, ln_w, ln_b, res, out, R, N, act, s);
    }
    return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
}

// training: the row GEMM on pre-split weight planes (sd_pack_weight_blocks); same variants as launch_panel16
template <int D>
static int launch_panel16_packed(const float *A, int lda, const void *wpk, const float *bias, const float *ln_w, const float *ln_b,
                                 const float *res, float *out, int R, int N, int act, hipStream_t s, const DropoutArgs &da) {
    using C = PanelCfg<D>;
    ProfScope prof(SD_KCLASS_PANEL_GEMM, s);
    dim3 grid((R + C::BM - 1) / C::BM), block(256);
    const size_t lds = C::LDS_BYTES + C::BM * sizeof(float);
    const float *W = reinterpret_cast<const float *>(wpk);
#define SD_PANEL16P(LN_, ACT_, RES_, DROP_)                                                                    \
    do {                                                                                                       \
        auto kfn = panel_gemm16_kernel<D, LN_, ACT_, RES_, DROP_, true>;                                       \
        static DevFlag attr_set;                                                                                 \
        if (lds > 64 * 1024 && !attr_set) {                                                                    \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                   \
        }                                                                                                      \
        SD_LAUNCH(kfn, grid, block, lds, s, A, W, bias, ln_w, ln_b, res, out, R, N, lda, da);                  \
    } while (0)
    const bool ln = ln_w != nullptr, rs = res != nullptr;
    if (da.thresh) {
        if (ln || act != 0 || !rs || N % 4 != 0) return fail(SD_E_BADARG, "sd_op_linear_packed: dropout needs res, no LayerNorm, no activation");
        SD_PANEL16P(false, 0, true, true);
    } else if (ln && act == 0 && !rs) SD_PANEL16P(true, 0, false, false);
    else if (!ln && act == 0 && !rs) SD_PANEL16P(false, 0, false, false);
    else if (!ln && act == 0 && rs) SD_PANEL16P(false, 0, true, false);
    else return fail(SD_E_BADARG, "sd_op_linear_packed: unsupported LN/act/res combination");
#undef SD_PANEL16P
    SD_CHECK_LAUNCH("panel_gemm16_kernel");
    return 0;
}

extern "C" int sd_op_linear_packed(const float *A, int lda, const void *wpk, const float *bias, const float *ln_w, const float *ln_b,
                                   const float *res, float *out, int R, int N, int d, float p, uint64_t seed, uint64_t site,
                                   void *stream) {
    if (!A || !wpk || !out || R <= 0 || N <= 0 || N % d != 0) return fail(SD_E_BADARG, "sd_op_linear_packed: bad argument");
    if ((ln_w == nullptr) != (ln_b == nullptr)) return fail(SD_E_BADARG, "sd_op_linear_packed: ln_w/ln_b mismatch");
    if (!(p >= 0.f) || !(p < 1.f)) return fail(SD_E_BADARG, "sd_op_linear_packed: p must be in [0, 1)");
    if (lda == 0) lda = d;
    if (lda < d || lda % 4 != 0) return fail(SD_E_BADARG, "sd_op_linear_packed: row stride must be >= d and a multiple of 4");
    hipStream_t s = (hipStream_t)stream;
    const DropoutArgs da = p > 0.f ? make_dropout(p, seed, site) : DropoutArgs{};
    switch (d) {
        case 64: return launch_panel16_packed<64>(A, lda, wpk, bias, ln_w, ln_b, res, out, R, N, 0, s, da);
        case 128: return launch_panel16_packed<128>(A, lda, wpk, bias, ln_w, ln_b, res, out, R, N, 0, s, da);
        case 256: return launch_panel16_packed<256>(A, lda, wpk, bias, ln_w, ln_b, res, out, R, N, 0, s, da);
        case 512: return launch_panel16_packed<512>(A, lda, wpk, bias, ln_w, ln_b, res, out, R, N, 0, s, da);
    }
    return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
}

extern "C" int sd_pack_weight_blocks(const float *src, const int64_t *src_off_dev, int n_blocks, void *dst, int d, int transposed,
                                     void *stream) {
    if (!src || !src_off_dev || !dst || n_blocks <= 0) return fail(SD_E_BADARG, "sd_pack_weight_blocks: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long *off = reinterpret_cast<const long *>(src_off_dev);
    f16 *out = reinterpret_cast<f16 *>(dst);
    const int chunks = d * (d / 8) / 256 > 0 ? d * (d / 8) / 256 : 1, tiles = (d / 64) * (d / 64);
#define SD_PACK(D_)                                                                                                              \
    do {                                                                                                                         \
        if (transposed) SD_LAUNCH((f16_pack_blocks_t_kernel<D_>), dim3(tiles, n_blocks), dim3(256), 0, s, src, off, out, F16_W_SCALE); \
        else SD_LAUNCH((f16_pack_blocks_kernel<D_>), dim3(chunks, n_blocks), dim3(256), 0, s, src, off, out, F16_W_SCALE);         \
    } while (0)
    switch (d) {
        case 64: SD_PACK(64); break;
        case 128: SD_PACK(128); break;
        case 256: SD_PACK(256); break;
        case 512: SD_PACK(512); break;
        default: return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
    }
#undef SD_PACK
    SD_CHECK_LAUNCH("f16_pack_blocks_kernel");
    return 0;
}

extern "C" int sd_op_linear_dropout(const float *A, int lda, const float *W, const float *bias, const float *res, float *out, int R,
                                    int N, int d, float p, uint64_t seed, uint64_t site, void *stream) {
    if (!A || !W || !out || !res || R <= 0 || N <= 0 || N % d != 0) return fail(SD_E_BADARG, "sd_op_linear_dropout: bad argument");
    if (!(p >= 0.f) || !(p < 1.f)) return fail(SD_E_BADARG, "sd_op_linear_dropout: p must be in [0, 1)");
    if (lda == 0) lda = d;
    if (lda < d || lda % 4 != 0) return fail(SD_E_BADARG, "sd_op_linear_dropout: row stride must be >= d and a multiple of 4");
    hipStream_t s = (hipStream_t)stream;
    if (p == 0.f) return linear(A, W, bias, nullptr, nullptr, res, out, R, N, d, 0, s, lda);
    const DropoutArgs da = make_dropout(p, seed, site);
    switch (d) {
        case 64: return launch_panel16<64>(A, lda, W, bias, nullptr, nullptr, res, out, R, N, 0, s, da);
        case 128: return launch_panel16<128>(A, lda, W, bias, nullptr, nullptr, res, out, R, N, 0, s, da);
        case 256: return launch_panel16<256>(A, lda, W, bias, nullptr, nullptr, res, out, R, N, 0, s, da);
        case 512: return launch_panel16<512>(A, lda, W, bias, nullptr, nullptr, res, out, R, N, 0, s, da);
    }
    return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
}

static int decoder_layer_f16(const F16LayerArgs &fa, hipStream_t s) {
    if (fa.g.a.R <= 0) return fail(SD_E_BADARG, "decoder_layer_f16: empty shape");
    ProfScope prof(SD_KCLASS_LAYER_CHAIN, s);
    dim3 grid((unsigned)((fa.g.a.R + 63) / 64)), block(256);
    const size_t lds = PanelCfg<256>::LDS_BYTES;
    static DevFlag attr_set;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)decoder_layer_f16_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)decoder_layer_f16_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (fa.g.fo_w) SD_LAUNCH((decoder_layer_f16_kernel<256, true>), grid, block, lds, s, fa);
    else SD_LAUNCH((decoder_layer_f16_kernel<256, false>), grid, block, lds, s, fa);
    SD_CHECK_LAUNCH("decoder_layer_f16_kernel");
    return 0;
}

// self-attention over qkv [B*T][3d] on the fp16 pipe (head dim 64, T <= 128)
static int attention_f16(const float *qkv, float *out, int B, int T, int d, int heads, hipStream_t s, bool head_major,
                         float *lse2 = nullptr, const DropoutArgs &da = DropoutArgs{}) {
    ProfScope prof(SD_KCLASS_ATTENTION, s);
    const float sl2e = (1.0f / sqrtf(64.0f)) * 1.44269504088896340736f;
    if (da.thresh) {   // training forward with dropout on the probabilities: row-major q|k|v only
        if (head_major) return fail(SD_E_BADARG, "attention_f16: dropout needs the row-major q|k|v layout");
        SD_LAUNCH((attention_f16_head_lv_kernel<false, true>), dim3(B * heads), dim3(256), ATT16LV_LDS, s, qkv, 3 * d, out, d, T, heads, sl2e, lse2, da);
        SD_CHECK_LAUNCH("attention_f16_head_lv_kernel");
        return 0;
    }
    // default: one workgroup per (sample, head), V^T staged late into K's LDS (35 KB of LDS, 3 workgroups per CU);
    // A/B runs: "stage2" = K and V^T staged together (69 KB, 2 per CU), "stream" = the per-sample streaming kernel
    static const char *env = getenv("SD_ATT16");
    if (!head_major && !lse2 && env && strcmp(env, "stream") == 0) {
        SD_LAUNCH(attention_f16_kernel, dim3(B), dim3(256), 0, s, qkv, 3 * d, out, d, T, heads, sl2e);
        SD_CHECK_LAUNCH("attention_f16_kernel");
        return 0;
    }
    if (lse2 || !(env && strcmp(env, "stage2") == 0)) {
        if (head_major) SD_LAUNCH((attention_f16_head_lv_kernel<true>), dim3(B * heads), dim3(256), ATT16LV_LDS, s, qkv, 3 * d, out, d, T, heads, sl2e, lse2, DropoutArgs{});
        else SD_LAUNCH((attention_f16_head_lv_kernel<false>), dim3(B * heads), dim3(256), ATT16LV_LDS, s, qkv, 3 * d, out, d, T, heads, sl2e, lse2, DropoutArgs{});
        SD_CHECK_LAUNCH("attention_f16_head_lv_kernel");
        return 0;
    }
    static DevFlag attr_set;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)attention_f16_head_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT16H_LDS);
        (void)hipFuncSetAttribute((const void *)attention_f16_head_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT16H_LDS);
        attr_set = true;
    }
    if (head_major) SD_LAUNCH((attention_f16_head_kernel<true>), dim3(B * heads), dim3(256), ATT16H_LDS, s, qkv, 3 * d, out, d, T, heads, sl2e);
    else SD_LAUNCH((attention_f16_head_kernel<false>), dim3(B * heads), dim3(256), ATT16H_LDS, s, qkv, 3 * d, out, d, T, heads, sl2e);
    SD_CHECK_LAUNCH("attention_f16_head_kernel");
    return 0;
}

template <int D>
static int chain_f16_launch(const F16ChainArgs &fa, int kind, hipStream_t s) {
    const long R = kind == 0 ? fa.a.R : fa.b.R;
    if (R <= 0) return fail(SD_E_BADARG, "chain_f16: empty shape");
    ProfScope prof(SD_KCLASS_LAYER_CHAIN, s);
    dim3 grid((unsigned)((R + 63) / 64)), block(256);
    const size_t lds = PanelCfg<D>::LDS_BYTES;
    static DevFlag attr_set;
    if (lds > 64 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute((const void *)chain_f16_kernel<D, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)chain_f16_kernel<D, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (kind == 0) SD_LAUNCH((chain_f16_kernel<D, 0>), grid, block, lds, s, fa);
    else SD_LAUNCH((chain_f16_kernel<D, 1>), grid, block, lds, s, fa);
    SD_CHECK_LAUNCH("chain_f16_kernel");
    return 0;
}
static int chain_f16(const F16ChainArgs &fa, int d, int kind, hipStream_t s) {
    return d == 128 ? chain_f16_launch<128>(fa, kind, s) : d == 256 ? chain_f16_launch<256>(fa, kind, s) : chain_f16_launch<512>(fa, kind, s);
}

static int decoder_head_f16(const F16HeadArgs &fa, hipStream_t s, int d = 256) {
    if (fa.g.R <= 0) return fail(SD_E_BADARG, "decoder_head_f16: empty shape");
    ProfScope prof(SD_KCLASS_HEAD, s);
    dim3 grid((unsigned)((fa.g.R + 63) / 64)), block(256);
    const bool one_wrap = fa.g.T >= 64;
    if (d == 128) {
        if (one_wrap) SD_LAUNCH((decoder_head_f16_kernel<128, true>), grid, block, PanelCfg<128>::LDS_BYTES, s, fa);
        else SD_LAUNCH((decoder_head_f16_kernel<128, false>), grid, block, PanelCfg<128>::LDS_BYTES, s, fa);
    } else if (d == 512) {
        const size_t lds = PanelCfg<512>::LDS_BYTES;
        static DevFlag attr_set512;
        if (!attr_set512) {
            (void)hipFuncSetAttribute((const void *)decoder_head_f16_kernel<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void *)decoder_head_f16_kernel<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set512 = true;
        }
        if (one_wrap) SD_LAUNCH((decoder_head_f16_kernel<512, true>), grid, block, lds, s, fa);
        else SD_LAUNCH((decoder_head_f16_kernel<512, false>), grid, block, lds, s, fa);
    } else {
        const size_t lds = PanelCfg<256>::LDS_BYTES;
        static DevFlag attr_set;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)decoder_head_f16_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void *)decoder_head_f16_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        if (one_wrap) SD_LAUNCH((decoder_head_f16_kernel<256, true>), grid, block, lds, s, fa);
        else SD_LAUNCH((decoder_head_f16_kernel<256, false>), grid, block, lds, s, fa);
    }
    SD_CHECK_LAUNCH("decoder_head_f16_kernel");
    return 0;
}

static int decoder_head(const DecoderHeadArgs &g, int d, hipStream_t s) {
    return launch_chain(g, d, decoder_head_kernel<64>, decoder_head_kernel<128>, decoder_head_kernel<256>,
                        decoder_head_kernel<512>, "decoder_head_kernel", s, SD_KCLASS_HEAD);
}

// true when the fused decoder-layer kernel applies: 4 heads == 4 waves each owning one head's
// columns (D >= 128), and at most 64 memory keys among the trajectories touching a 64-row panel
static bool fused_layer_ok(int d, int heads, int T, int Mk) {
    if (heads != 4 || d < 128) return false;
    const int n_traj = (63 + T - 1) / T + 1;
    return (long)n_traj * Mk <= 64;
}

static int decoder_layer(const DecoderLayerArgs &g, int d, hipStream_t s) {
    if (g.a.R <= 0) return fail(SD_E_BADARG, "decoder_layer: empty shape");
    ProfScope prof(SD_KCLASS_LAYER_CHAIN, s);
    dim3 grid((unsigned)((g.a.R + 63) / 64)), block(256);
#define SD_DL(D_)                                                                                                \
    do {                                                                                                         \
        auto kfn = g.gv ? (g.fo_w ? decoder_layer_kernel<D_, true, true> : decoder_layer_kernel<D_, false, true>)  \
                        : (g.fo_w ? decoder_layer_kernel<D_, true, false> : decoder_layer_kernel<D_, false, false>); \
        const size_t lds = PanelCfg<D_>::LDS_BYTES;                                                              \
        static DevFlag attr_set;                                                                                   \
        if (lds > 64 * 1024 && !attr_set) {                                                                      \
            (void)hipFuncSetAttribute((const void *)decoder_layer_kernel<D_, true, true>,                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
            (void)hipFuncSetAttribute((const void *)decoder_layer_kernel<D_, false, true>,                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
            (void)hipFuncSetAttribute((const void *)decoder_layer_kernel<D_, true, false>,                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
            (void)hipFuncSetAttribute((const void *)decoder_layer_kernel<D_, false, false>,                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
            attr_set = true;                                                                                     \
        }                                                                                                        \
        SD_LAUNCH(kfn, grid, block, lds, s, g);                                                                  \
    } while (0)
    switch (d) {
        case 128: SD_DL(128); break;
        case 256: SD_DL(256); break;
        case 512: SD_DL(512); break;
        default: return fail(SD_E_BADDIM, "decoder_layer: hidden_dim must be 128, 256 or 512");
    }
#undef SD_DL
    SD_CHECK_LAUNCH("decoder_layer_kernel");
    return 0;
}

// ======================================================================================
// Attention core (self- and cross-attention), unmasked, one workgroup per (sample, head).
//
// Computed transposed: S^T = K Q^T (keys x queries) so that a query is a lane column.
// The softmax statistics of a query are then in-lane reductions plus one lane^32 exchange,
// and the probabilities, still sitting in the accumulator registers, are directly the
// B operand of O^T = V^T P^T (the contraction runs over the accumulator's row index), so
// P never touches LDS.  Keys are streamed through LDS in chunks of KC with an online
// softmax, so any S is supported; each wave owns 32 queries (128 per workgroup pass).
// ======================================================================================
template <int HD>
struct AttnCfg {
    static constexpr int KC = 64;                          // keys per LDS chunk (2 tiles: 32 score registers, 4 workgroups/CU)
    static constexpr int FT = (HD + 31) / 32;              // 32-feature tiles of O^T
    static constexpr int LDK = HD + 4;                     // ds_read_b128 conflict-free
    static constexpr int LDV = FT * 32;                    // zero-padded feature columns
    static constexpr int KSTEPS = HD / 8;
    static constexpr size_t LDS_BYTES = (size_t)KC * (LDK + LDV) * sizeof(float);
};

// DROP (training): dropout on the attention probabilities (torch nn.MultiheadAttention dropout): the normaliser uses
// the un-dropped probabilities, O = (P o m) V; mask row = (sample, head, query), column = key (sd_common.h).
template <int HD, bool DROP = false>
__global__ __launch_bounds__(256) void attention_kernel(const float *__restrict__ q, int ldq,
                                                         const float *__restrict__ k, const float *__restrict__ v,
                                                         int ldkv, const float *__restrict__ k_extra,
                                                         const float *__restrict__ v_extra, float *__restrict__ out,
                                                         int ldo, int Tq, int S, int heads, float scale_log2e,
                                                         float *__restrict__ lse2, DropoutArgs da = DropoutArgs{}) {
    using C = AttnCfg<HD>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S_all = S + (k_extra ? 1 : 0);
    // LDS is sized by the launcher for min(KC, S rounded up to a key tile) rows: short
    // memories (cross-attention, M = 11) take 17 KB instead of 68 KB and 8 workgroups fit a CU
    const int rows_cap = min(C::KC, ((S_all + 31) / 32) * 32);
    float *sK = smem;
    float *sV = smem + rows_cap * C::LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const float *qb = q + (long)b * Tq * ldq + h * HD;
    const float *kb = k + (long)b * S * ldkv + h * HD;
    const float *vb = v + (long)b * S * ldkv + h * HD;

    for (int qpass = 0; qpass < Tq; qpass += 128) {
        const int q0 = qpass + wave * 32;
        const bool wave_active = q0 < Tq;  // wave-uniform
        // Q fragments stay in registers for the whole pass: B[k = feature][j = query]
        f32x4 qf[C::KSTEPS];
        {
            const int qi = q0 + l31;
            const bool ok = qi < Tq;
#pragma unroll
            for (int st = 0; st < C::KSTEPS; ++st) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if (ok) t = *reinterpret_cast<const f32x4 *>(qb + (long)qi * ldq + st * 8 + 4 * half);
                qf[st] = t;
            }
        }
        f32x16 o[C::FT];
#pragma unroll
        for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ft][r] = 0.f;
        float m_run = -INFINITY, l_part = 0.f;

        for (int kc0 = 0; kc0 < S_all; kc0 += C::KC) {
            __syncthreads();  // previous chunk fully consumed
            // ---- stage K and V chunk (zero-filled past S_all; V feature pad zeroed) -----
            constexpr int KV4 = HD / 4;
            const int rows_staged = ((min(S_all - kc0, C::KC) + 31) / 32) * 32;  // only the tiles that are read
            for (int i = tid; i < rows_staged * KV4; i += 256) {
                const int row = i / KV4, c4 = i - row * KV4;
                const int key = kc0 + row;
                f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
                if (key < S) {
                    kv = *reinterpret_cast<const f32x4 *>(kb + (long)key * ldkv + c4 * 4);
                    vv = *reinterpret_cast<const f32x4 *>(vb + (long)key * ldkv + c4 * 4);
                } else if (key < S_all) {
                    kv = *reinterpret_cast<const f32x4 *>(k_extra + h * HD + c4 * 4);
                    vv = *reinterpret_cast<const f32x4 *>(v_extra + h * HD + c4 * 4);
                }
                *reinterpret_cast<f32x4 *>(sK + row * C::LDK + c4 * 4) = kv;
                *reinterpret_cast<f32x4 *>(sV + row * C::LDV + c4 * 4) = vv;
            }
            if constexpr (C::LDV > HD) {
                constexpr int PADW = C::LDV - HD;
                for (int i = tid; i < rows_staged * PADW; i += 256) sV[(i / PADW) * C::LDV + HD + (i % PADW)] = 0.f;
            }
            __syncthreads();
            if (!wave_active) continue;

            // ---- S^T chunk = K Q^T : KC/32 tiles of (32 keys x 32 queries) ---------------
            constexpr int KT = C::KC / 32;
            f32x16 sc[KT];
            const int kt_valid = (min(S_all - kc0, C::KC) + 31) / 32;  // wave-uniform
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
                if (kt < kt_valid) {
                    const float *kp = sK + (kt * 32 + l31) * C::LDK + 4 * half;
#pragma unroll
                    for (int st = 0; st < C::KSTEPS; ++st) {
                        const f32x4 kf = *reinterpret_cast<const f32x4 *>(kp + st * 8);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[st][j], sc[kt], 0, 0, 0);
                    }
                }
            }
            // ---- online softmax for this lane's query (column l31) -----------------------
            float m_c = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kc0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float sv = (key < S_all) ? sc[kt][r] : -INFINITY;
                    sc[kt][r] = sv;
                    m_c = fmaxf(m_c, sv);
                }
            m_c = fmaxf(m_c, __shfl_xor(m_c, 32, 64));
            const float m_new = fmaxf(m_run, m_c);  // finite: every chunk holds >= 1 valid key
            const float alpha = exp2f((m_run - m_new) * scale_log2e);
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = exp2f((sc[kt][r] - m_new) * scale_log2e);
                    sc[kt][r] = p;
                    psum += p;
                }
            if constexpr (DROP) {
                const int qi = q0 + l31;
                const unsigned long mrow = ((unsigned long)b * heads + h) * Tq + (qi < Tq ? qi : 0);
                const unsigned long wq = (unsigned long)((S_all + 3) >> 2);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    if (kt < kt_valid) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 m4 = dropout_quad(da, mrow * wq + (unsigned long)((kc0 + kt * 32 + 8 * g + 4 * half) >> 2));
#pragma unroll
                            for (int e = 0; e < 4; ++e) sc[kt][4 * g + e] *= m4[e];
                        }
                    }
            }
            l_part = l_part * alpha + psum;
#pragma unroll
            for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ft][r] *= alpha;
            // ---- O^T += V^T P^T : A = V^T[feature l31][key], B = P^T straight from sc ------
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if (kt < kt_valid) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // registers 4g..4g+3 contract keys 8g..8g+7 of the tile: skip the group when
                        // all of them are past the end (their p is 0); one wave-uniform test per 8 keys
                        if (kc0 + kt * 32 + 8 * g < S_all) {
#pragma unroll
                            for (int ri = 0; ri < 4; ++ri) {
                                const int r = 4 * g + ri;
                                const int krow = kt * 32 + ri + 8 * g + 4 * half;
#pragma unroll
                                for (int ft = 0; ft < C::FT; ++ft) {
                                    const float a = sV[krow * C::LDV + ft * 32 + l31];
                                    o[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sc[kt][r], o[ft], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (wave_active) {
            const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
            const float inv = 1.0f / l_tot;
            const int qi = q0 + l31;
            // log2-domain log-sum-exp of the scaled scores, for the backward's P recompute
            if (lse2 && qi < Tq && half == 0) lse2[((long)b * heads + h) * Tq + qi] = m_run * scale_log2e + log2f(l_tot);
            if (qi < Tq) {
                float *op = out + ((long)b * Tq + qi) * ldo + h * HD;
#pragma unroll
                for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int f = ft * 32 + 8 * g + 4 * half;  // 4 consecutive features
                        if (f < HD) {
                            f32x4 t = {o[ft][4 * g] * inv, o[ft][4 * g + 1] * inv, o[ft][4 * g + 2] * inv,
                                       o[ft][4 * g + 3] * inv};
                            *reinterpret_cast<f32x4 *>(op + f) = t;
                        }
                    }
            }
        }
    }
}

// --------------------------------------------------------------------------------------
// Pipelined variant for Tq <= 128 (one query pass): ONE workgroup per sample walks all heads
// and key chunks as a single stream of (head, chunk) units.  The K/V rows of unit u+1 are
// loaded global -> registers while unit u computes and are written to LDS after it, and the
// next head's Q fragments are fetched during the current head's last chunk, so after the
// first unit no HBM/L2 latency is exposed (attention_kernel exposes it once per chunk).
// --------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256, (HD <= 64 ? 2 : 1)) void attention_pipe_kernel(const float *__restrict__ q, int ldq,
                                                              const float *__restrict__ k, const float *__restrict__ v,
                                                              int ldkv, float *__restrict__ out, int ldo, int Tq, int S,
                                                              int heads, float scale_log2e, float *__restrict__ lse2) {
    using C = AttnCfg<HD>;
    constexpr int KV4 = HD / 4;
    constexpr int NV = (C::KC * KV4 + 255) / 256;  // 16-byte pieces of K (and of V) per thread per chunk
    constexpr int KT = C::KC / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sK = smem;
    float *sV = smem + C::KC * C::LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x;
    const int nchunks = (S + C::KC - 1) / C::KC;
    const int nunits = heads * nchunks;
    const int q0 = wave * 32;
    const bool wave_active = q0 < Tq;  // wave-uniform
    const int qi = q0 + l31;
    const bool q_ok = qi < Tq;
    const float *qrow = q + ((long)b * Tq + (q_ok ? qi : 0)) * ldq + 4 * half;
    const float *kb = k + (long)b * S * ldkv;
    const float *vb = v + (long)b * S * ldkv;

    if constexpr (C::LDV > HD) {  // zero the feature padding of V once; staging never touches it
        constexpr int PADW = C::LDV - HD;
        for (int i = tid; i < C::KC * PADW; i += 256) sV[(i / PADW) * C::LDV + HD + (i % PADW)] = 0.f;
    }
    f32x4 kreg[NV], vreg[NV];
    auto fetch = [&](int u) {  // unit u = (head, chunk): this thread's pieces of its 64 K/V rows
        const int h = u / nchunks, kc0 = (u - h * nchunks) * C::KC;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / KV4, c4 = idx - row * KV4;
            const int key = kc0 + row;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
            if (idx < C::KC * KV4 && key < S) {
                a = *reinterpret_cast<const f32x4 *>(kb + (long)key * ldkv + h * HD + c4 * 4);
                d = *reinterpret_cast<const f32x4 *>(vb + (long)key * ldkv + h * HD + c4 * 4);
            }
            kreg[i] = a;
            vreg[i] = d;
        }
    };
    f32x4 qf[C::KSTEPS], qn[C::KSTEPS];
    auto fetch_q = [&](int h, f32x4 (&dst)[C::KSTEPS]) {
#pragma unroll
        for (int st = 0; st < C::KSTEPS; ++st) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (q_ok) t = *reinterpret_cast<const f32x4 *>(qrow + h * HD + st * 8);
            dst[st] = t;
        }
    };
    fetch(0);
    fetch_q(0, qn);
    f32x16 o[C::FT];
    float m_run = -INFINITY, l_part = 0.f;

    for (int u = 0; u < nunits; ++u) {
        const int h = u / nchunks, c = u - h * nchunks, kc0 = c * C::KC;
        __syncthreads();  // every wave is done reading the previous unit's K/V
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / KV4, c4 = idx - row * KV4;
            if (idx < C::KC * KV4) {
                *reinterpret_cast<f32x4 *>(sK + row * C::LDK + c4 * 4) = kreg[i];
                *reinterpret_cast<f32x4 *>(sV + row * C::LDV + c4 * 4) = vreg[i];
            }
        }
        if (u + 1 < nunits) fetch(u + 1);  // in flight during this unit's MFMAs
        if (c == 0) {
#pragma unroll
            for (int st = 0; st < C::KSTEPS; ++st) qf[st] = qn[st];
#pragma unroll
            for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ft][r] = 0.f;
            m_run = -INFINITY;
            l_part = 0.f;
        }
        if (c == nchunks - 1 && h + 1 < heads) fetch_q(h + 1, qn);
        __syncthreads();
        if (wave_active) {
            f32x16 sc[KT];
            const int kt_valid = (min(S - kc0, C::KC) + 31) / 32;  // wave-uniform
            // K fragments double-buffered and the order "read next, then 4 MFMAs" pinned: hipcc otherwise
            // issues every ds_read right before the MFMAs that need it and exposes the LDS latency each time
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
                if (kt < kt_valid) {
                    const float *kp = sK + (kt * 32 + l31) * C::LDK + 4 * half;
                    f32x4 kf[2];
                    kf[0] = *reinterpret_cast<const f32x4 *>(kp);
#pragma unroll
                    for (int st = 0; st < C::KSTEPS; ++st) {
                        if (st + 1 < C::KSTEPS) kf[(st + 1) & 1] = *reinterpret_cast<const f32x4 *>(kp + (st + 1) * 8);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[st & 1][j], qf[st][j], sc[kt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            float m_c = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kc0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float sv = (key < S) ? sc[kt][r] : -INFINITY;
                    sc[kt][r] = sv;
                    m_c = fmaxf(m_c, sv);
                }
            m_c = fmaxf(m_c, __shfl_xor(m_c, 32, 64));
            const float m_new = fmaxf(m_run, m_c);
            const float alpha = exp2f((m_run - m_new) * scale_log2e);
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = exp2f((sc[kt][r] - m_new) * scale_log2e);
                    sc[kt][r] = pv;
                    psum += pv;
                }
            l_part = l_part * alpha + psum;
#pragma unroll
            for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ft][r] *= alpha;
            // P V: groups of 4 accumulator registers (8 keys); the V operands of the next group are read
            // while the current group's MFMAs run (pinned); dead groups (keys past S) are skipped
            {
                const float *vp = sV + (4 * half) * C::LDV + l31;
                float av[2][4][C::FT];
#pragma unroll
                for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                    for (int ft = 0; ft < C::FT; ++ft) av[0][ri][ft] = vp[ri * C::LDV + ft * 32];
#pragma unroll
                for (int gg = 0; gg < KT * 4; ++gg) {
                    const int kt = gg >> 2, g = gg & 3;
                    if (gg + 1 < KT * 4) {
                        const int kt1 = (gg + 1) >> 2, g1 = (gg + 1) & 3;
#pragma unroll
                        for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                            for (int ft = 0; ft < C::FT; ++ft)
                                av[(gg + 1) & 1][ri][ft] = vp[(kt1 * 32 + 8 * g1 + ri) * C::LDV + ft * 32];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (kc0 + kt * 32 + 8 * g < S) {  // wave-uniform
#pragma unroll
                        for (int ri = 0; ri < 4; ++ri)
#pragma unroll
                            for (int ft = 0; ft < C::FT; ++ft)
                                o[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gg & 1][ri][ft], sc[kt][4 * g + ri], o[ft], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (c == nchunks - 1) {
                const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
                const float inv = 1.0f / l_tot;
                if (lse2 && q_ok && half == 0) lse2[((long)b * heads + h) * Tq + qi] = m_run * scale_log2e + log2f(l_tot);
                if (q_ok) {
                    float *op = out + ((long)b * Tq + qi) * ldo + h * HD;
#pragma unroll
                    for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int f = ft * 32 + 8 * g + 4 * half;
                            if (f < HD) {
                                f32x4 t = {o[ft][4 * g] * inv, o[ft][4 * g + 1] * inv, o[ft][4 * g + 2] * inv,
                                           o[ft][4 * g + 3] * inv};
                                *reinterpret_cast<f32x4 *>(op + f) = t;
                            }
                        }
                }
            }
        }
    }
}

static int attention(const float *q, int ldq, const float *k, const float *v, int ldkv, const float *k_extra,
                     const float *v_extra, float *out, int ldo, int B, int Tq, int S, int d, int heads,
                     hipStream_t s, float *lse2 = nullptr, const DropoutArgs &da = DropoutArgs{}) {
    if (!q || !k || !v || !out || B <= 0 || Tq <= 0 || S < 0 || heads <= 0)
        return fail(SD_E_BADARG, "attention: null pointer or empty shape");
    if ((k_extra == nullptr) != (v_extra == nullptr)) return fail(SD_E_BADARG, "attention: k_extra/v_extra mismatch");
    if (S + (k_extra ? 1 : 0) <= 0) return fail(SD_E_BADARG, "attention: no keys");
    if (d % heads != 0) return fail(SD_E_BADDIM, "attention: d not divisible by heads");
    const int hd = d / heads;
    ProfScope prof(SD_KCLASS_ATTENTION, s);
    const float sl2e = (1.0f / sqrtf((float)hd)) * 1.44269504088896340736f;
    dim3 grid(B * heads), block(256);
    // small batches (the robot: B = 1): one workgroup per (sample, head) instead of one per sample walking its heads
    // in sequence - the rollout there is a chain of kernel latencies (robot shape: 11.6 -> 9.4 ms per rollout)
    constexpr int pipe_min_b = 64;
    if (Tq <= 128 && !k_extra && S > 0 && B >= pipe_min_b && !da.thresh) {
        dim3 gridp(B);
#define SD_ATTNP(HD_)                                                                                            \
    do {                                                                                                         \
        auto kfn = attention_pipe_kernel<HD_>;                                                                   \
        const size_t lds = AttnCfg<HD_>::LDS_BYTES;                                                              \
        static DevFlag attr_set;                                                                                   \
        if (lds > 64 * 1024 && !attr_set) {                                                                      \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            attr_set = true;                                                                                     \
        }                                                                                                        \
        SD_LAUNCH(kfn, gridp, block, lds, s, q, ldq, k, v, ldkv, out, ldo, Tq, S, heads, sl2e, lse2);            \
    } while (0)
        switch (hd) {
            case 16: SD_ATTNP(16); break;
            case 32: SD_ATTNP(32); break;
            case 64: SD_ATTNP(64); break;
            case 128: SD_ATTNP(128); break;
            default: return fail(SD_E_BADDIM, "attention: head dim must be 16, 32, 64 or 128");
        }
#undef SD_ATTNP
        SD_CHECK_LAUNCH("attention_pipe_kernel");
        return 0;
    }
#define SD_ATTN(HD_)                                                                                             \
    do {                                                                                                         \
        auto kfn = da.thresh ? attention_kernel<HD_, true> : attention_kernel<HD_, false>;                       \
        const int s_all = S + (k_extra ? 1 : 0);                                                                 \
        const int rows_cap = std::min((int)AttnCfg<HD_>::KC, ((s_all + 31) / 32) * 32);                           \
        const size_t lds = (size_t)rows_cap * (AttnCfg<HD_>::LDK + AttnCfg<HD_>::LDV) * sizeof(float);           \
        static DevFlag attr_set[2];                                                                                \
        if (!attr_set[da.thresh ? 1 : 0]) {                                                                      \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize,             \
                                      (int)AttnCfg<HD_>::LDS_BYTES);                                             \
            attr_set[da.thresh ? 1 : 0] = true;                                                                  \
        }                                                                                                        \
        SD_LAUNCH(kfn, grid, block, lds, s, q, ldq, k, v, ldkv, k_extra, v_extra, out, ldo, Tq, S, heads, sl2e, lse2, da); \
    } while (0)
    switch (hd) {
        case 16: SD_ATTN(16); break;
        case 32: SD_ATTN(32); break;
        case 64: SD_ATTN(64); break;
        case 128: SD_ATTN(128); break;
        default: return fail(SD_E_BADDIM, "attention: head dim must be 16, 32, 64 or 128");
    }
#undef SD_ATTN
    SD_CHECK_LAUNCH("attention_kernel");
    return 0;
}

// ======================================================================================
// Patch embedding (+ bias + positional table):  Conv1d(kernel = stride = p) == a GEMM with
// K = C*p <= a few hundred; 0.4 % of the step's FLOPs, done on the VALU.  p = 1 is the
// decoder's nn.Linear(J -> d).  One workgroup = 64 output rows; W^T and the 64 input
// patches are staged in LDS ([k][c] and [row][k+1]: conflict-free reads).
// ======================================================================================
__global__ __launch_bounds__(256) void patch_embed_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ bias,
                                                           const float *__restrict__ pe, float *__restrict__ out,
                                                           long rows, int n_per_sample, int S, int C, int p, int d,
                                                           int RB, int DC) {
    extern __shared__ float sm[];
    const int K = C * p;
    float *sX = sm;                          // [RB][K+1]
    float *sW = sm + (size_t)RB * (K + 1);   // [K][DC]  (column chunk of W^T)
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * RB;
    for (int i = tid; i < RB * K; i += 256) {
        const int row = i / K, kk = i - row * K;
        const long r = r0 + row;
        float v = 0.f;
        if (r < rows) {
            const long bidx = r / n_per_sample;
            const int n = (int)(r - bidx * n_per_sample);
            const int c = kk / p, t = kk - c * p;  // w is (d, C, p): kk = c*p + t
            v = x[((bidx * S) + (long)n * p + t) * C + c];
        }
        sX[row * (K + 1) + kk] = v;
    }
    for (int c0 = 0; c0 < d; c0 += DC) {
        __syncthreads();
        for (int i = tid; i < DC * K; i += 256) {
            const int cc = i / K, kk = i - cc * K;
            sW[kk * DC + cc] = (c0 + cc < d) ? w[(long)(c0 + cc) * K + kk] : 0.f;
        }
        __syncthreads();
        if (RB == 64 && DC % 256 == 0) {
            // a thread owns one output column for all 64 rows: the weight goes through a register and the patch value is ONE
            // address for the whole workgroup (a broadcast read) - half the LDS reads of the generic loop below, none of
            // them conflicting (80 -> 25 us for the decoder's Linear(20 -> 256) at 25 600 rows, a training step's embedding)
            for (int cc = tid; cc < DC; cc += 256) {
                float acc[64];
#pragma unroll
                for (int row = 0; row < 64; ++row) acc[row] = 0.f;
                for (int kk = 0; kk < K; ++kk) {
                    const float wv = sW[kk * DC + cc];
#pragma unroll
                    for (int row = 0; row < 64; ++row) acc[row] = fmaf(sX[row * (K + 1) + kk], wv, acc[row]);
                }
                const int c = c0 + cc;
                if (c < d) {
                    const float bv = bias[c];
#pragma unroll
                    for (int row = 0; row < 64; ++row) {
                        const long r = r0 + row;
                        if (r < rows) out[r * d + c] = acc[row] + bv + pe[(long)(r % n_per_sample) * d + c];
                    }
                }
            }
            continue;
        }
        for (int i = tid; i < RB * DC; i += 256) {
            const int row = i / DC, cc = i - row * DC;
            const long r = r0 + row;
            const int c = c0 + cc;
            if (r >= rows || c >= d) continue;
            float acc = 0.f;
            for (int kk = 0; kk < K; ++kk) acc = fmaf(sX[row * (K + 1) + kk], sW[kk * DC + cc], acc);
            const int n = (int)(r % n_per_sample);
            out[r * d + c] = acc + bias[c] + pe[(long)n * d + c];
        }
    }
}

static int patch_embed(const float *x, const float *w, const float *b, const float *pe, float *out, int B, int S,
                       int C, int p, int d, hipStream_t s) {
    if (!x || !w || !b || !pe || !out || B <= 0 || S <= 0 || C <= 0 || p <= 0 || d <= 0)
        return fail(SD_E_BADARG, "patch_embed: null pointer or empty shape");
    const int n = S / p;
    if (n <= 0) return fail(SD_E_BADARG, "patch_embed: sequence shorter than one patch");
    const long rows = (long)B * n;
    const int K = C * p;
    const int RB = (K <= 64) ? 64 : 16;
    const long budget = 15360 - (long)RB * (K + 1);  // floats (60 KB total)
    long DC = budget > 0 ? (budget / K) / 32 * 32 : 0;
    if (DC > d) DC = (d + 31) / 32 * 32;
    if (DC < 32) return fail(SD_E_TOOBIG, "patch_embed: C*p too large for LDS staging");
    const size_t lds = ((size_t)RB * (K + 1) + (size_t)K * DC) * sizeof(float);
    ProfScope prof(SD_KCLASS_PATCH_EMBED, s);
    SD_LAUNCH(patch_embed_kernel, dim3((unsigned)((rows + RB - 1) / RB)), dim3(256), lds, s, x, w, b, pe, out,
                       rows, n, S, C, p, d, RB, (int)DC);
    SD_CHECK_LAUNCH("patch_embed_kernel");
    return 0;
}

// ======================================================================================
// fc_out (d -> J) fused with the DDIM update of x:  eps = h W^T + b;  x <- ddim(x, eps).
// J = 20 pads to one 32-wide MFMA tile.  Same row-panel skeleton as panel_gemm (64 rows of
// h in LDS); the 4 waves take (row tile, K half) and the two K halves are summed through
// LDS.  Weight rows >= J read as zero.  The kernel is HBM-bound (one read of h).
// coef = {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)}.
// ======================================================================================
#define SD_MAX_J 64
template <int D>
__global__ __launch_bounds__(256) void fc_out_kernel(const float *__restrict__ h, const float *__restrict__ W,
                                                      const float *__restrict__ b, float *__restrict__ eps,
                                                      float *x_io, float c0, float c1, float c2, float c3, long R,
                                                      int J) {
    constexpr int BM = 64, LDA = D + 4, KH = D / 2;
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const long r0 = (long)blockIdx.x * BM;
    constexpr int VEC_PER_ROW = D / 4;
    for (int i = tid; i < BM * VEC_PER_ROW; i += 256) {
        const int row = i / VEC_PER_ROW, c4 = i - row * VEC_PER_ROW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < R) v = *reinterpret_cast<const f32x4 *>(h + (r0 + row) * D + c4 * 4);
        *reinterpret_cast<f32x4 *>(sA + row * LDA + c4 * 4) = v;
    }
    __syncthreads();
    const int tm = wave & 1, kh = wave >> 1;
    const float *aBase = sA + (tm * 32 + l31) * LDA + kh * KH + 4 * half;
    const int n_tiles = (J + 31) / 32;  // 1 or 2
    f32x16 acc[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
    for (int tn = 0; tn < n_tiles; ++tn) {
        const int j = tn * 32 + l31;
        const float *wp = W + (long)(j < J ? j : 0) * D + kh * KH + 4 * half;
        const float wmask = j < J ? 1.f : 0.f;
#pragma unroll 4
        for (int k0 = 0; k0 < KH; k0 += 8) {
            f32x4 bf = *reinterpret_cast<const f32x4 *>(wp + k0);
            const f32x4 af = *reinterpret_cast<const f32x4 *>(aBase + k0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (tn == 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj], bf[jj] * wmask, acc[0], 0, 0, 0);
                else acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj], bf[jj] * wmask, acc[1], 0, 0, 0);
            }
        }
    }
    __syncthreads();  // panel no longer needed: reuse it for the K-half exchange
    float *sX = sA;   // [2 row tiles][2 n tiles][16 regs][64 lanes]
    if (kh == 1) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) sX[((tm * 2 + tn) * 16 + r) * 64 + lane] = acc[tn][r];
    }
    __syncthreads();
    if (kh == 0) {
        for (int tn = 0; tn < n_tiles; ++tn) {
            const int j = tn * 32 + l31;
            if (j >= J) continue;
            const float bv = b[j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = r0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= R) continue;
                const float e = (tn == 0 ? acc[0][r] : acc[1][r]) + sX[((tm * 2 + tn) * 16 + r) * 64 + lane] + bv;
                if (eps) eps[row * J + j] = e;
                if (x_io) {
                    const float xv = x_io[row * J + j];
                    const float x0 = (xv - c1 * e) / c0;
                    x_io[row * J + j] = c2 * x0 + c3 * e;
                }
            }
        }
    }
}

static int fc_out(const float *h, const float *W, const float *b, float *eps, float *x_io, const float *coef, long R,
                  int d, int J, hipStream_t s) {
    if (!h || !W || !b || (!eps && !x_io) || R <= 0) return fail(SD_E_BADARG, "fc_out: null pointer or empty shape");
    if (x_io && !coef) return fail(SD_E_BADARG, "fc_out: DDIM update needs coefficients");
    if (J <= 0 || J > SD_MAX_J) return fail(SD_E_TOOBIG, "fc_out: joints must be in 1..64");
    const float c0 = coef ? coef[0] : 1.f, c1 = coef ? coef[1] : 0.f, c2 = coef ? coef[2] : 1.f, c3 = coef ? coef[3] : 0.f;
    ProfScope prof(SD_KCLASS_FC_OUT, s);
    dim3 grid((unsigned)((R + 63) / 64)), block(256);
#define SD_FCOUT(D_)                                                                                            \
    do {                                                                                                        \
        auto kfn = fc_out_kernel<D_>;                                                                           \
        const size_t lds = (size_t)64 * (D_ + 4) * sizeof(float);                                               \
        static DevFlag attr_set;                                                                                  \
        if (lds > 64 * 1024 && !attr_set) {                                                                     \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            attr_set = true;                                                                                    \
        }                                                                                                       \
        SD_LAUNCH(kfn, grid, block, lds, s, h, W, b, eps, x_io, c0, c1, c2, c3, R, J);                          \
    } while (0)
    switch (d) {
        case 64: SD_FCOUT(64); break;
        case 128: SD_FCOUT(128); break;
        case 256: SD_FCOUT(256); break;
        case 512: SD_FCOUT(512); break;
        default: return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
    }
#undef SD_FCOUT
    SD_CHECK_LAUNCH("fc_out_kernel");
    return 0;
}

// ======================================================================================
// small elementwise kernels
// ======================================================================================
__global__ void step_token_kernel(const void *steps, int is_i64, const float *__restrict__ freq,
                                  const float *__restrict__ token, float *out, long stride, int B, int d) {
    const int b = blockIdx.x;
    const int n = d / 4;
    const float t = is_i64 ? (float)reinterpret_cast<const int64_t *>(steps)[b] : reinterpret_cast<const float *>(steps)[b];
    float *o = out + (long)b * stride;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        float v;
        if (i < n) v = sinf(t * freq[i]);
        else if (i < 2 * n) v = cosf(t * freq[i - n]);
        else v = token[i - 2 * n];
        o[i] = v;
    }
}

__global__ void gather_rows_kernel(const int64_t *__restrict__ idx, const float *__restrict__ table, float *out,
                                   long stride, int B, int d, int n_states) {
    const int b = blockIdx.x;
    int64_t i = idx[b];
    if (i < 0) i = 0;
    if (i >= n_states) i = n_states - 1;
    for (int c = threadIdx.x; c < d; c += blockDim.x) out[(long)b * stride + c] = table[i * d + c];
}

__global__ void add_noise_kernel(const float *__restrict__ x0, const float *__restrict__ noise,
                                 const int64_t *__restrict__ t, const float *__restrict__ acp, float *out, int B,
                                 int per_sample) {
    const long n = (long)B * per_sample;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float a = acp[t[i / per_sample]];
        out[i] = sqrtf(a) * x0[i] + sqrtf(1.0f - a) * noise[i];
    }
}

__global__ void ddim_step_kernel(const float *__restrict__ eps, const float *x, float *x_prev, float c0, float c1,
                                 float c2, float c3, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float e = eps[i];
        const float x0 = (x[i] - c1 * e) / c0;
        x_prev[i] = c2 * x0 + c3 * e;
    }
}

// Memory keys/values are kept per layer as [B][Mk][2D] (Mk = context rows + the step row) so
// that a trajectory's rows are contiguous.  mode 0: copy the projected context rows
// [B*Mc][2D] into rows 0..Mc-1 of each trajectory; mode 1: write the current step token's
// projected row into row Mc of every trajectory.  grid = (blocks, layers).
__global__ void kv_place_kernel(const float *__restrict__ src, long src_layer_stride, float *dst, long dst_layer_stride,
                                int B, int Mc, int Mk, int w2, int mode) {
    const float *sl = src + (long)blockIdx.y * src_layer_stride;
    float *dl = dst + (long)blockIdx.y * dst_layer_stride;
    const long n = (mode == 0) ? (long)B * Mc * w2 : (long)B * w2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % w2);
        const long row = i / w2;
        if (mode == 0) {
            const long b = row / Mc;
            const int m = (int)(row - b * Mc);
            dl[((b * Mk) + m) * w2 + c] = sl[i];
        } else {
            dl[((row * Mk) + Mc) * w2 + c] = sl[c];
        }
    }
}

// Folding of the cross-attention projections into the projected memory (once per rollout, see
// panel_folded_scores).  For memory row r = item*keys_per_item + m and head h (= blockIdx.y):
//     G  = K_h[r] Wq_h          (D floats)   K_h = kv[r][h*HD .. +HD),  Wq_h = rows h*HD.. of Wq (D x D)
//     V' = V_h[r] Wo[:, h]^T    (D floats)   V_h = kv[r][D + h*HD ..),  Wo[:, h] = columns h*HD.. of Wo
//     c  = bq_h . K_h[r]
// written to row item*item_rows + ((key0 + m) / head_rows) * 4 head_rows + h*head_rows + (key0 + m) % head_rows of gv ([rows][2D]) /
// cb ([rows]): an item's keys in tiles of head_rows slots, [tile][head][slot] (one tile when keys_per_item <= head_rows).
// VALU kernel (0.5 % of a rollout's flops): a block owns FOLD_RB memory rows in LDS, thread = column.
constexpr int FOLD_RB = 16;

__global__ __launch_bounds__(256) void xattn_fold_kernel(const float *__restrict__ kv, long n_rows, int keys_per_item,
                                                         const float *__restrict__ wq, const float *__restrict__ bq,
                                                         const float *__restrict__ wo, float *__restrict__ gv,
                                                         float *__restrict__ cb, long item_rows, int head_rows, int key0,
                                                         int D, int HD, unsigned *maxG, unsigned *maxV) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Ks = sm, *Vs = sm + FOLD_RB * HD;
    float mg = 0.f, mv = 0.f;   // abs-max of what this thread writes (the fp16x3 path derives its scales from them)
    const int h = blockIdx.y;
    const long r0 = (long)blockIdx.x * FOLD_RB;
    for (int i = threadIdx.x; i < FOLD_RB * HD; i += 256) {
        const int r = i / HD, j = i - r * HD;
        const bool ok = r0 + r < n_rows;
        Ks[i] = ok ? kv[(r0 + r) * 2 * D + h * HD + j] : 0.f;
        Vs[i] = ok ? kv[(r0 + r) * 2 * D + D + h * HD + j] : 0.f;
    }
    __syncthreads();
    long dst[FOLD_RB];
#pragma unroll
    for (int r = 0; r < FOLD_RB; ++r) {
        const long row = r0 + r, item = row / keys_per_item;
        const int key = key0 + (int)(row - item * keys_per_item);   // key tile key / head_rows, slot key % head_rows
        dst[r] = item * item_rows + (long)(key / head_rows) * (4 * head_rows) + (long)h * head_rows + key % head_rows;
    }
    if (threadIdx.x < FOLD_RB && r0 + threadIdx.x < n_rows) {
        float c = 0.f;
        for (int j = 0; j < HD; ++j) c += bq[h * HD + j] * Ks[threadIdx.x * HD + j];
        const long row = r0 + threadIdx.x, item = row / keys_per_item;
        const int key = key0 + (int)(row - item * keys_per_item);
        cb[item * item_rows + (long)(key / head_rows) * (4 * head_rows) + (long)h * head_rows + key % head_rows] = c;
    }
    for (int n = threadIdx.x; n < D; n += 256) {
        float acc[FOLD_RB];
#pragma unroll
        for (int r = 0; r < FOLD_RB; ++r) acc[r] = 0.f;
        for (int j = 0; j < HD; j += 4) {
            float w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = wq[(long)(h * HD + j + e) * D + n];
#pragma unroll
            for (int r = 0; r < FOLD_RB; ++r) {
                const f32x4 k4 = *reinterpret_cast<const f32x4 *>(Ks + r * HD + j);
                acc[r] += k4[0] * w[0] + k4[1] * w[1] + k4[2] * w[2] + k4[3] * w[3];
            }
        }
#pragma unroll
        for (int r = 0; r < FOLD_RB; ++r)
            if (r0 + r < n_rows) {
                gv[dst[r] * 2 * D + n] = acc[r];
                mg = fmaxf(mg, fabsf(acc[r]));
            }
#pragma unroll
        for (int r = 0; r < FOLD_RB; ++r) acc[r] = 0.f;
        for (int j = 0; j < HD; j += 4) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(wo + (long)n * D + h * HD + j);
#pragma unroll
            for (int r = 0; r < FOLD_RB; ++r) {
                const f32x4 v4 = *reinterpret_cast<const f32x4 *>(Vs + r * HD + j);
                acc[r] += v4[0] * w[0] + v4[1] * w[1] + v4[2] * w[2] + v4[3] * w[3];
            }
        }
#pragma unroll
        for (int r = 0; r < FOLD_RB; ++r)
            if (r0 + r < n_rows) {
                gv[dst[r] * 2 * D + D + n] = acc[r];
                mv = fmaxf(mv, fabsf(acc[r]));
            }
    }
    if (maxG) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mg = fmaxf(mg, __shfl_xor(mg, o, 64));
            mv = fmaxf(mv, __shfl_xor(mv, o, 64));
        }
        // most waves find the word already at or above their value: read first, the atomics of 40 000 waves on two
        // addresses would otherwise serialise (the kernel took 2.3x as long)
        if ((threadIdx.x & 63) == 0) {
            const unsigned bg = __builtin_bit_cast(unsigned, mg), bv = __builtin_bit_cast(unsigned, mv);
            if (bg > __atomic_load_n(maxG, __ATOMIC_RELAXED)) atomicMax(maxG, bg);
            if (bv > __atomic_load_n(maxV, __ATOMIC_RELAXED)) atomicMax(maxV, bv);
        }
    }
}

// The current step token's folded rows ([4 heads][2D] + 4 score biases per layer) -> key slot Mc of
// every trajectory.  grid = (blocks, layers).
__global__ void fold_place_kernel(const float *__restrict__ src, const float *__restrict__ csrc, long src_layer_stride,
                                  long csrc_layer_stride, float *gv, float *cb, long gv_layer_stride,
                                  long cb_layer_stride, int B, int Mc, int w2) {
    const float *sl = src + (long)blockIdx.y * src_layer_stride;
    const float *cl = csrc + (long)blockIdx.y * csrc_layer_stride;
    float *gl = gv + (long)blockIdx.y * gv_layer_stride;
    float *bl = cb + (long)blockIdx.y * cb_layer_stride;
    const long n = (long)B * 4 * w2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % w2);
        const long bh = i / w2;           // trajectory * 4 + head
        const int h = (int)(bh & 3);
        const long row = (bh >> 2) * 64 + h * 16 + Mc;
        gl[row * w2 + c] = sl[h * w2 + c];
        if (c == 0) bl[row] = cl[h];
    }
}

// Normalizer.normalize / denormalize (reference dataset/pytorch.py:410-414): per-joint affine
__global__ void normalize_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                 const float *__restrict__ stdv, float *__restrict__ out, long n, int J, int inverse) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % J);
        out[i] = inverse ? x[i] * stdv[j] + mean[j] : (x[i] - mean[j]) / stdv[j];
    }
}

__global__ void copy_rows_kernel(const float *__restrict__ src, long src_stride, float *dst, long dst_stride,
                                 long rows, int width) {
    const long n = rows * width;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / width;
        const int c = (int)(i - r * width);
        dst[r * dst_stride + c] = src[r * src_stride + c];
    }
}

// Zero-fill of workspace regions as a KERNEL, not hipMemsetAsync: memset nodes captured into a hipGraph were observed
// (ROCm 7.2, gfx950) to replay with a garbage fill byte - the range-guard word read 0x58585858 / 0x80808080 after a
// replay, and the abs-max words behind the fp16 scales were hit the same way (a replay then differed from the eager
// rollout in the last bits).  A kernel node has no such problem.  n16 = number of 16-byte pieces.
__global__ void zero_fill_kernel(f32x4 *__restrict__ p, long n16) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) p[i] = z;
}
__global__ void zero_words_kernel(unsigned *__restrict__ p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}
// bytes: a multiple of 4; 16-byte aligned regions go through the wide kernel
static int zero_async(void *ptr, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && bytes % 16 == 0) {
        SD_LAUNCH(zero_fill_kernel, dim3(grid_for((long)(bytes / 16))), dim3(256), 0, st, reinterpret_cast<f32x4 *>(ptr), (long)(bytes / 16));
    } else {
        SD_LAUNCH(zero_words_kernel, dim3(grid_for((long)(bytes / 4))), dim3(256), 0, st, reinterpret_cast<unsigned *>(ptr), (long)(bytes / 4));
    }
    SD_CHECK_LAUNCH("zero_fill_kernel");
    return 0;
}

// Range guard of the sampler: ORs `bit` into *status when any value is not finite.  The split-fp16 kernels use fixed
// activation scales (sd_f16x3.h): |8 x| >= 65520 becomes an fp16 infinity, its lo part -inf, their products NaN, and a
// NaN stays in its trajectory's rows down to x - so one pass over the sample is a complete detector (33 MB at B = 4096).
__global__ void finite_check_kernel(const float *__restrict__ x, long n, int *status, int bit) {
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const unsigned u = __builtin_bit_cast(unsigned, x[i]);
        bad |= (u & 0x7f800000u) == 0x7f800000u;   // exponent all ones: inf or NaN
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, bit);
}


// ======================================================================================
// Layer drivers
// ======================================================================================
struct Scratch {  // carve-up of the caller's workspace (floats)
    float *h, *qkv, *a, *u, *kv, *kvstep, *kvtmp;
    float *gv, *cb, *gvstep, *cstep;   // folded cross-attention (sampler only)
    // fp16x3 operands of the sampler (sd_f16x3.h): split weights, folded blocks, scales
    f16 *wf, *g16, *v16, *gstep16, *vstep16;
    f16 *wio;   // sampler mode 3 (sd_traj.h): embedding (256 x 32) and fc_out (32 x 256) planes
    float *scales;
    unsigned *maxbits;
    // split weights of the unfused row chains (chain_f16_kernel, hidden_dim 128 / 256 / 512): per layer
    // [Wo | Wq | Woc | W1 | W2 | in_proj x 3], their scales and abs-max words
    f16 *wfc;
    float *scc;
    unsigned *mbc;
    int *stepmap;   // trajectory kernels, per-trajectory step tokens: block index each trajectory reads (step_map_kernel)
    float *gws;   // region of the generic trajectory kernels (sd_trajg.hip: hidden_dim 128 / 256 / 512, any memory length), or NULL
};

// the fp16x3 kernels are instantiated for hidden_dim 256 (the folded fp32 kernels serve the other sizes);
// SD_SAMPLER_GEMM=f32 keeps the fp32-MFMA kernels (A/B runs)
static bool f16_env_ok() {
    static const char *env = getenv("SD_SAMPLER_GEMM");
    return !(env && strcmp(env, "f32") == 0);
}
static bool f16_ok(int d, int J) { return f16_env_ok() && d == 256 && J % 4 == 0; }

// folded cross-attention applies: fused layer kernel, <= 16 key slots per head, <= 2 trajectories per panel
static bool fold_ok(int d, int heads, int T, int Mk) { return heads == 4 && d >= 128 && Mk <= 16 && T >= 64; }

static size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

static Scratch carve(float *ws, long R, long RM, int d, int L, int n_steps, long B = 0) {
    Scratch s;
    size_t off = 0;
    const size_t kt = B > 0 ? (size_t)((RM / B + 15) / 16) : 1;   // key tiles of the folded blocks (memory rows per trajectory / 16)
    s.h = ws + off; off += align64((size_t)((R + 63) / 64 * 64) * d);   // whole panels (accumulator-order layout)
    s.qkv = ws + off; off += align64((size_t)R * 3 * d);
    s.a = ws + off; off += align64((size_t)R * d);
    s.u = ws + off; off += align64((size_t)R * d);
    s.kv = ws + off; off += align64((size_t)L * RM * 2 * d);
    s.kvstep = ws + off; off += align64((size_t)L * (n_steps > 0 ? n_steps : 1) * 2 * d);
    s.kvtmp = ws + off; off += align64((size_t)L * RM * 2 * d);
    s.gv = s.cb = s.gvstep = s.cstep = nullptr;
    if (n_steps > 0) {
        s.gv = ws + off; off += align64((size_t)L * B * kt * 64 * 2 * d);
        s.cb = ws + off; off += align64((size_t)L * B * kt * 64);
        s.gvstep = ws + off; off += align64((size_t)L * n_steps * 4 * 2 * d);
        s.cstep = ws + off; off += align64((size_t)L * n_steps * 4);
    }
    s.wf = s.g16 = s.v16 = s.gstep16 = s.vstep16 = s.wio = nullptr;
    s.scales = nullptr;
    s.maxbits = nullptr;
    if (n_steps > 0 && d == 256) {   // sizes in floats (2 halfs each)
        s.wf = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * 6 * d * d);
        s.g16 = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * B * kt * 4 * 16 * d);
        s.v16 = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * B * kt * 4 * 16 * d);
        s.gstep16 = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * n_steps * 4 * 16 * d);
        s.vstep16 = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * n_steps * 16 * d);
        s.scales = ws + off; off += align64((size_t)(L + 1) * 8);
        s.maxbits = reinterpret_cast<unsigned *>(ws + off); off += align64((size_t)(L + 1) * 8);
        s.wio = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)2 * 32 * d);
    }
    s.wfc = nullptr;
    s.scc = nullptr;
    s.mbc = nullptr;
    if (n_steps > 0 && (d == 128 || d == 256 || d == 512)) {
        s.wfc = reinterpret_cast<f16 *>(ws + off); off += align64((size_t)L * 8 * d * d);
        s.scc = ws + off; off += align64((size_t)L * 8);
        s.mbc = reinterpret_cast<unsigned *>(ws + off); off += align64((size_t)L * 8);
    }
    s.stepmap = nullptr;
    if (n_steps > 0) { s.stepmap = reinterpret_cast<int *>(ws + off); off += align64((size_t)n_steps); }
    s.gws = nullptr;
    if (n_steps > 0 && B > 0 && (d == 128 || d == 256 || d == 512)) s.gws = ws + off;   // trajg_workspace_floats(B, Mc, d, L, n_steps) floats
    return s;
}

extern "C" size_t sd_workspace_floats(int B, int T, int M, int d, int L, int n_steps) {
    // M memory rows per trajectory (+1: the sampler adds the step row to the context rows)
    const size_t R = (size_t)B * T, RM = (size_t)B * ((M > 0 ? M : 0) + 1);
    const size_t kt = (size_t)(((M > 0 ? M : 0) + 1 + 15) / 16);   // as carve()
    size_t n = align64(R * d) * 2 + align64((R + 63) / 64 * 64 * d) + align64(R * 3 * d) + 2 * align64((size_t)L * RM * 2 * d) +
               align64((size_t)L * (n_steps > 0 ? n_steps : 1) * 2 * d) + 1024;
    if (n_steps > 0)   // folded cross-attention blocks of the sampler: 64 rows of 2d (+ 1 bias) per trajectory and layer
        n += align64((size_t)L * B * kt * 64 * 2 * d) + align64((size_t)L * B * kt * 64) + align64((size_t)L * n_steps * 4 * 2 * d) +
             align64((size_t)L * n_steps * 4);
    if (n_steps > 0 && d == 256)   // fp16x3 operands (sd_f16x3.h)
        n += align64((size_t)L * 6 * d * d) + 2 * align64((size_t)L * B * kt * 4 * 16 * d) + align64((size_t)L * n_steps * 4 * 16 * d) +
             align64((size_t)L * n_steps * 16 * d) + 2 * align64((size_t)(L + 1) * 8) + align64((size_t)2 * 32 * d);
    if (n_steps > 0 && (d == 128 || d == 256 || d == 512)) n += align64((size_t)L * 8 * d * d) + 2 * align64((size_t)L * 8);
    if (n_steps > 0) n += align64((size_t)n_steps);   // step map
    if (n_steps > 0 && (d == 128 || d == 256 || d == 512)) n += trajg_workspace_floats(B, M > 0 ? M : 0, d, L, n_steps);
    return n;
}

// Decoder stack on the fused row chains.  On entry s.h holds the embedded trajectory rows.
//   head:      qkv = LN1_0(h) Wqkv_0^T + b
//   per layer: a = self-attention(qkv);  chain A: h += a Wo^T + bo, q = LN2(h) Wq^T + bq;
//              a = cross-attention(q, memory K/V);  chain B: h += a Woc^T + boc,
//              h += FFN(LN3(h)), qkv = LN1_{l+1}(h) Wqkv_{l+1}^T + b (if any)
// kv(l): projected memory keys | values of layer l, [B][Mk][2d] (a trajectory's rows contiguous).
struct TailArgs {  // fc_out (+ DDIM) after the last layer
    float *eps, *x_io;
    const float *coef;  // host, 4 floats, or NULL
};

struct FoldArgs {  // folded cross-attention blocks per layer (gv NULL: unfolded)
    const float *gv, *cb;
    size_t gv_stride, cb_stride;
};

// split weights of the unfused chains of layer l (which: 0 Wo, 1 Wq, 2 Woc, 3 W1, 4 W2, 5 in_proj)
static f16 *f16_wfc(const Scratch &s, int l, int d, int which) { return s.wfc + ((size_t)l * 8 + which) * 2 * d * d; }

template <int D>
static int f16_prepare_chain_d(const sd_denoiser_weights *w, const Scratch &s, hipStream_t st) {
    const int d = w->d, L = w->L;
    if (int rz = zero_async(s.mbc, (size_t)L * 8 * sizeof(unsigned), st)) return rz;
    for (int pass = 0; pass < 2; ++pass)
        for (int l = 0; l < L; ++l) {
            const sd_layer_weights &lw = w->layers[l];
            const float *mats[6] = {lw.sa_out_w, lw.ca_in_w, lw.ca_out_w, lw.lin1_w, lw.lin2_w, lw.sa_in_w};
            const int rows[6] = {d, d, d, d, d, 3 * d};
            for (int m = 0; m < 6; ++m) {
                if (pass == 0) {
                    SD_LAUNCH(f16_absmax_kernel, dim3(grid_for((long)rows[m] * d)), dim3(256), 0, st, mats[m], (long)rows[m] * d, s.mbc + l * 8 + m);
                    SD_CHECK_LAUNCH("f16_absmax_kernel");
                } else {
                    SD_LAUNCH((f16_pack_weight_kernel<D>), dim3(grid_for((long)rows[m] * d / 8)), dim3(256), 0, st, mats[m], rows[m],
                              s.mbc + l * 8 + m, f16_wfc(s, l, d, m), s.scc + l * 8 + m);
                    SD_CHECK_LAUNCH("f16_pack_weight_kernel");
                }
            }
        }
    return 0;
}
static int f16_prepare_chain(const sd_denoiser_weights *w, const Scratch &s, hipStream_t st) {
    return w->d == 128 ? f16_prepare_chain_d<128>(w, s, st) : w->d == 256 ? f16_prepare_chain_d<256>(w, s, st) : f16_prepare_chain_d<512>(w, s, st);
}

// the unfused row chains run on the fp16 pipe when their split weights were prepared (sampler, hidden_dim 128 / 256 / 512;
// SD_SAMPLER_GEMM=f32 keeps the fp32 kernels)
static bool chain16_ok(int d, int J) {
    static const char *env = getenv("SD_SAMPLER_GEMM");
    if (env && strcmp(env, "f32") == 0) return false;
    return (d == 128 || d == 256 || d == 512) && J % 4 == 0;
}

template <typename KV>
static int decoder_stack(const sd_denoiser_weights *w, const float *x, const Scratch &s, int B, int T, int Mk, KV kv,
                         const TailArgs &tail, hipStream_t st, const FoldArgs &fold = FoldArgs{nullptr, nullptr, 0, 0},
                         bool chain16 = false) {
    const int d = w->d, heads = w->heads;
    const long R = (long)B * T;
    const sd_layer_weights &l0 = w->layers[0];
    int rc;
    const bool fused = fused_layer_ok(d, heads, T, Mk) && (long)B * Mk * 2 * d < (1L << 30);
    chain16 = chain16 && !fused && s.wfc != nullptr;
    if (chain16) {
        F16HeadArgs fh{DecoderHeadArgs{x, w->emb_w, w->emb_b, w->pe, l0.n1_w, l0.n1_b, l0.sa_in_w, l0.sa_in_b, s.h, s.qkv, R, T, w->J},
                       f16_wfc(s, 0, d, 5), s.scc + 2, 0, 0};   // the head reads sc[3]: the scale of layer 0's in_proj = scc[5]
        rc = decoder_head_f16(fh, st, d);
    } else if (w->J % 4 == 0) {  // embed + LN1 + QKV of layer 0 in one launch
        DecoderHeadArgs gh{x, w->emb_w, w->emb_b, w->pe, l0.n1_w, l0.n1_b, l0.sa_in_w, l0.sa_in_b, s.h, s.qkv, R, T, w->J};
        rc = decoder_head(gh, d, st);
    } else {
        rc = patch_embed(x, w->emb_w, w->emb_b, w->pe, s.h, B, T, w->J, 1, d, st);
        if (rc) return rc;
        rc = linear(s.h, l0.sa_in_w, l0.sa_in_b, l0.n1_w, l0.n1_b, nullptr, s.qkv, (int)R, 3 * d, d, 0, st);
    }
    if (rc) return rc;
    for (int l = 0; l < w->L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        rc = attention(s.qkv, 3 * d, s.qkv + d, s.qkv + 2 * d, 3 * d, nullptr, nullptr, s.a, d, B, T, T, d, heads, st);
        if (rc) return rc;
        ChainAArgs ga{s.a, s.h, lw.sa_out_w, lw.sa_out_b, lw.n2_w, lw.n2_b, lw.ca_in_w, lw.ca_in_b, s.u, R};
        const float *kvl = kv(l);
        const bool last = l + 1 == w->L;
        const sd_layer_weights *nx = last ? nullptr : &w->layers[l + 1];
        ChainBArgs gb{s.a, s.h, lw.ca_out_w, lw.ca_out_b, lw.n3_w, lw.n3_b, lw.lin1_w, lw.lin1_b, lw.lin2_w, lw.lin2_b,
                      nx ? nx->n1_w : nullptr, nx ? nx->n1_b : nullptr, nx ? nx->sa_in_w : nullptr,
                      nx ? nx->sa_in_b : nullptr, s.qkv, R};
        if (fused) {
            DecoderLayerArgs gl{ga, gb, kvl, T, Mk, B, (1.0f / sqrtf((float)(d / heads))) * 1.44269504088896340736f,
                                nullptr, nullptr, nullptr, nullptr, 1.f, 0.f, 1.f, 0.f, w->J,
                                fold.gv ? fold.gv + l * fold.gv_stride : nullptr, fold.gv ? fold.cb + l * fold.cb_stride : nullptr, l};
            if (last) {
                gl.fo_w = w->out_w;
                gl.fo_b = w->out_b;
                gl.eps = tail.eps;
                gl.x_io = tail.x_io;
                if (tail.coef) { gl.c0 = tail.coef[0]; gl.c1 = tail.coef[1]; gl.c2 = tail.coef[2]; gl.c3 = tail.coef[3]; }
            }
            if ((rc = decoder_layer(gl, d, st))) return rc;
            continue;
        }
        if (chain16) {
            F16ChainArgs fa{ga, gb, f16_wfc(s, l, d, 0), f16_wfc(s, l, d, 1), nullptr, nullptr, s.scc + l * 8, nullptr};
            if ((rc = chain_f16(fa, d, 0, st))) return rc;
            rc = attention(s.u, d, kvl, kvl + d, 2 * d, nullptr, nullptr, s.a, d, B, T, Mk, d, heads, st);
            if (rc) return rc;
            // chain B: (Woc, W1, W2) of this layer, in_proj of the next
            F16ChainArgs fb{ga, gb, f16_wfc(s, l, d, 2), f16_wfc(s, l, d, 3), f16_wfc(s, l, d, 4),
                            last ? nullptr : f16_wfc(s, l + 1, d, 5), s.scc + l * 8 + 2, s.scc + (last ? l : l + 1) * 8 + 5};
            if ((rc = chain_f16(fb, d, 1, st))) return rc;
            continue;
        }
        if ((rc = chain_a(ga, d, st))) return rc;
        rc = attention(s.u, d, kvl, kvl + d, 2 * d, nullptr, nullptr, s.a, d, B, T, Mk, d, heads, st);
        if (rc) return rc;
        if ((rc = chain_b(gb, d, st))) return rc;
    }
    if (fused) return 0;  // the last fused layer already produced eps / updated x
    return fc_out(s.h, w->out_w, w->out_b, tail.eps, tail.x_io, tail.coef, R, d, w->J, st);
}

// ---- fp16x3 sampler path (sd_f16x3.h) --------------------------------------------------------------
// split weights of layer l: [Wo | W1 | W2 | in_proj (3 passes)] in fragment-major planes, 12 d^2 halfs
static f16 *f16_wf(const Scratch &s, int l, int d, int which) {   // which: 0 Wo, 1 W1, 2 W2, 3 in_proj
    return s.wf + ((size_t)l * 6 + (which < 3 ? which : 3)) * 2 * d * d;
}

// once per rollout, after the fp32 fold (gv, gvstep): scales, split weights, split folded blocks
static int f16_prepare(const sd_denoiser_weights *w, const Scratch &s, int B, int Mc, int n_steps, hipStream_t st) {
    const int d = w->d, L = w->L;
    const size_t gvstride = (size_t)B * 64 * 2 * d, gvsstride = (size_t)n_steps * 4 * 2 * d;
    // maxbits were zeroed before the fold kernels, which left the abs-max of G and V' there
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        const float *mats[4] = {lw.sa_out_w, lw.lin1_w, lw.lin2_w, lw.sa_in_w};
        const int rows[4] = {d, d, d, 3 * d};
        unsigned *mb = s.maxbits + l * 8;
        for (int m = 0; m < 4; ++m) {
            SD_LAUNCH(f16_absmax_kernel, dim3(grid_for((long)rows[m] * d)), dim3(256), 0, st, mats[m], (long)rows[m] * d, mb + m);
            SD_CHECK_LAUNCH("f16_absmax_kernel");
        }
    }
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        const float *mats[4] = {lw.sa_out_w, lw.lin1_w, lw.lin2_w, lw.sa_in_w};
        const int rows[4] = {d, d, d, 3 * d};
        unsigned *mb = s.maxbits + l * 8;
        float *sc = s.scales + l * 8;
        for (int m = 0; m < 4; ++m) {
            SD_LAUNCH((f16_pack_weight_kernel<256>), dim3(grid_for((long)rows[m] * d / 8)), dim3(256), 0, st, mats[m], rows[m], mb + m,
                      f16_wf(s, l, d, m), sc + m);
            SD_CHECK_LAUNCH("f16_pack_weight_kernel");
        }
        const size_t blk = (size_t)32 * d;
        if (Mc > 0) {
            SD_LAUNCH((f16_pack_g_kernel<256>), dim3(grid_for((long)B * 4 * 16 * d / 8)), dim3(256), 0, st, s.gv + l * gvstride, (long)B, 16, 0,
                      mb + 4, s.g16 + (size_t)l * B * 4 * blk, sc + 4);
            SD_CHECK_LAUNCH("f16_pack_g_kernel");
            SD_LAUNCH((f16_pack_v_kernel<256>), dim3(grid_for((long)B * 4 * 2 * d)), dim3(256), 0, st, s.gv + l * gvstride, (long)B, mb + 5,
                      s.v16 + (size_t)l * B * 4 * blk, sc + 5);
            SD_CHECK_LAUNCH("f16_pack_v_kernel");
        } else {   // no context rows: the per-trajectory blocks are all zero
            int rz = zero_async(s.g16 + (size_t)l * B * 4 * blk, (size_t)B * 4 * blk * sizeof(f16), st);
            if (!rz) rz = zero_async(s.v16 + (size_t)l * B * 4 * blk, (size_t)B * 4 * blk * sizeof(f16), st);
            if (rz) return rz;
        }
        SD_LAUNCH((f16_pack_g_kernel<256>), dim3(grid_for((long)n_steps * 4 * 16 * d / 8)), dim3(256), 0, st, s.gvstep + l * gvsstride,
                  (long)n_steps, 1, Mc, mb + 4, s.gstep16 + (size_t)l * n_steps * 4 * blk, Mc > 0 ? (float *)nullptr : sc + 4);
        SD_CHECK_LAUNCH("f16_pack_g_kernel");
        if (Mc == 0) {   // scale of V' was not written by a per-trajectory pack
            SD_LAUNCH((f16_pack_v_kernel<256>), dim3(1), dim3(64), 0, st, s.gvstep, 0L, mb + 5, s.v16, sc + 5);
            SD_CHECK_LAUNCH("f16_pack_v_kernel");
        }
        SD_LAUNCH((f16_pack_vstep_kernel<256>), dim3(grid_for((long)n_steps * 2 * d)), dim3(256), 0, st, s.gvstep + l * gvsstride,
                  (long)n_steps, mb + 5, s.vstep16 + (size_t)l * n_steps * blk);
        SD_CHECK_LAUNCH("f16_pack_vstep_kernel");
    }
    return 0;
}

// one denoiser step + DDIM update on the fp16x3 kernels (step index i selects the step-token blocks)
static int decoder_stack_f16(const sd_denoiser_weights *w, float *x, const Scratch &s, int B, int T, int Mc, int i, int n_steps,
                             const float *coef, hipStream_t st, float *eps = nullptr) {
    const int d = w->d, heads = w->heads, L = w->L, Mk = Mc + 1;
    const long R = (long)B * T;
    const size_t blk = (size_t)32 * d, cbstride = (size_t)B * 64;
    const sd_layer_weights &l0 = w->layers[0];
    // q | k | v go to the attention kernel head-major when it is the fp16 one (head dim 64, T <= 128; SD_QKV=rows: A/B runs)
    static const char *qenv = getenv("SD_QKV");
    const bool att16 = d / heads == 64 && T <= 128, hm = att16 && heads == 4 && !(qenv && strcmp(qenv, "rows") == 0);
    static const char *henv = getenv("SD_H");   // "rows": h stays row-major (A/B runs)
    const int hfrag = !(henv && strcmp(henv, "rows") == 0);
    F16HeadArgs fh{DecoderHeadArgs{x, w->emb_w, w->emb_b, w->pe, l0.n1_w, l0.n1_b, l0.sa_in_w, l0.sa_in_b, s.h, s.qkv, R, T, w->J},
                   f16_wf(s, 0, d, 3), s.scales, hm ? 1 : 0, hfrag};
    // the head of steps 1.. runs inside the previous step's last layer kernel (SD_MERGE_HEAD=0: always its own launch; A/B runs)
    static const char *menv = getenv("SD_MERGE_HEAD");
    const bool merge = !(menv && strcmp(menv, "0") == 0) && T >= 64 && w->J <= 64;
    int rc = 0;
    if (i == 0 || !merge) rc = decoder_head_f16(fh, st);
    if (rc) return rc;
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        if (att16) rc = attention_f16(s.qkv, s.a, B, T, d, heads, st, hm);
        else rc = attention(s.qkv, 3 * d, s.qkv + d, s.qkv + 2 * d, 3 * d, nullptr, nullptr, s.a, d, B, T, T, d, heads, st);
        if (rc) return rc;
        const bool last = l + 1 == L;
        const sd_layer_weights *nx = last ? nullptr : &w->layers[l + 1];
        ChainAArgs ga{s.a, s.h, lw.sa_out_w, lw.sa_out_b, lw.n2_w, lw.n2_b, lw.ca_in_w, lw.ca_in_b, s.u, R};
        ChainBArgs gb{s.a, s.h, lw.ca_out_w, lw.ca_out_b, lw.n3_w, lw.n3_b, lw.lin1_w, lw.lin1_b, lw.lin2_w, lw.lin2_b,
                      nx ? nx->n1_w : nullptr, nx ? nx->n1_b : nullptr, nx ? nx->sa_in_w : nullptr, nx ? nx->sa_in_b : nullptr,
                      s.qkv, R};
        F16LayerArgs fa{};
        fa.g = DecoderLayerArgs{ga, gb, nullptr, T, Mk, B, (1.0f / sqrtf((float)(d / heads))) * 1.44269504088896340736f,
                                nullptr, nullptr, nullptr, nullptr, 1.f, 0.f, 1.f, 0.f, w->J, nullptr, s.cb + l * cbstride, l};
        if (last) {
            fa.g.fo_w = w->out_w;
            fa.g.fo_b = w->out_b;
            fa.g.x_io = x;
            fa.g.eps = eps;
            fa.g.c0 = coef[0]; fa.g.c1 = coef[1]; fa.g.c2 = coef[2]; fa.g.c3 = coef[3];
            fa.next_head = merge && i + 1 < n_steps;
            fa.head = fh;
#ifdef SD_STAMPS
            if (merge && !fa.next_head && n_steps > 1) fa.g.slot = L + 2;   // keep the stamps of the merged launch before it
#endif
        }
        fa.wf_o = f16_wf(s, l, d, 0);
        fa.wf_1 = f16_wf(s, l, d, 1);
        fa.wf_2 = f16_wf(s, l, d, 2);
        fa.wf_qkv = last ? nullptr : f16_wf(s, l + 1, d, 3);
        fa.sc_own = s.scales + l * 8;
        fa.sc_next = s.scales + (l + 1) * 8;
        fa.g16 = s.g16 + (size_t)l * B * 4 * blk;
        fa.v16 = s.v16 + (size_t)l * B * 4 * blk;
        fa.gstep = s.gstep16 + ((size_t)l * n_steps + i) * 4 * blk;
        fa.vstep = s.vstep16 + ((size_t)l * n_steps + i) * blk;
        fa.cstep = s.cstep + ((size_t)l * n_steps + i) * 4;
        fa.qkv_head_major = hm ? 1 : 0;
        fa.h_frag = hfrag;
        if ((rc = decoder_layer_f16(fa, st))) return rc;
    }
    return 0;
}


// ---- sampler mode 3: the trajectory-owning step kernel (sd_traj.h) ---------------------------------------
// One launch per DDIM step: embedding, all layers (self-attention inside), fc_out and the DDIM update for one trajectory per
// workgroup.  Same folded cross-attention blocks (gv, cb) and abs-max words as mode 2; the split planes are written in the
// 16x16x32 fragment order of sd_traj.h into the same workspace regions.  SD_SAMPLER_TRAJ=0 in the environment keeps mode 2.
static bool traj_ok(int d, int heads, int T, int Mk, int J, int L) {
    static const char *env = getenv("SD_SAMPLER_TRAJ");
    if (env && strcmp(env, "0") == 0) return false;
    // any joint count up to 32 (the embedding's K and fc_out's N are zero-padded to 32 in the packed planes; the reference's database
    // has 22 joints: soccer_diffusion/dataset/models.py:222-247)
    static const char *mr = getenv("SD_TRAJ_MAXROWS");   // A/B runs: memory rows beyond this go to the generic kernels (sd_trajg.hip) instead of the wide instantiation
    const int max_rows = mr ? atoi(mr) : 64;
    return f16_env_ok() && d == 256 && heads == 4 && T >= 1 && T <= tj::TMAX && Mk >= 1 && Mk <= (max_rows < 64 ? max_rows : 64) && J >= 1 && J <= 32 && L >= 1 &&
           L <= tj::MAX_L;
}
// key tiles of 16 memory slots in the folded blocks: 1 for the trajectory kernels proper, 2 .. 4 for traj_step_wide_kernel (17 .. 64 rows)
static int key_tiles(int Mk) { return Mk <= 16 ? 1 : (Mk + 15) / 16; }

// the instantiation for ceil(T / 16) token tiles; precise = three fp16 products at the Q | K | V site too (sampler mode 3), else two (mode 4)
typedef void (*TrajStepFn)(tj::StepArgs);
static TrajStepFn traj_step_wide_fn(int ntt) {
    switch (ntt) {
        case 1: return tj::traj_step_wide_kernel<1>;
        case 2: return tj::traj_step_wide_kernel<2>;
        case 3: return tj::traj_step_wide_kernel<3>;
        case 4: return tj::traj_step_wide_kernel<4>;
        case 5: return tj::traj_step_wide_kernel<5>;
        case 6: return tj::traj_step_wide_kernel<6>;
        case 7: return tj::traj_step_wide_kernel<7>;
        default: return nullptr;
    }
}
template <bool PRECISE>
static TrajStepFn traj_step_fn(int ntt) {
    switch (ntt) {
        case 1: return tj::traj_step_kernel<1, PRECISE>;
        case 2: return tj::traj_step_kernel<2, PRECISE>;
        case 3: return tj::traj_step_kernel<3, PRECISE>;
        case 4: return tj::traj_step_kernel<4, PRECISE>;
        case 5: return tj::traj_step_kernel<5, PRECISE>;
        case 6: return tj::traj_step_kernel<6, PRECISE>;
        case 7: return tj::traj_step_kernel<7, PRECISE>;
        default: return nullptr;
    }
}

// zeroes words [col0, col0 + ncols) of every 8-word row of the abs-max table
__global__ void zero_word_cols_kernel(unsigned *mb, int rows, int col0, int ncols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * ncols) mb[(i / ncols) * 8 + col0 + i % ncols] = 0u;
}
static int zero_word_cols(unsigned *mb, int rows, int col0, int ncols, hipStream_t st) {
    SD_LAUNCH(zero_word_cols_kernel, dim3((unsigned)((rows * ncols + 63) / 64)), dim3(64), 0, st, mb, rows, col0, ncols);
    SD_CHECK_LAUNCH("zero_word_cols_kernel");
    return 0;
}

// The trajectory path prepares its operands in three independent stages (sd_ddim_sample_eps runs all three per call;
// sd_sampler_prepare / sd_sampler_eps let a caller that evaluates the denoiser step by step - the reference's own loop,
// soccer_diffusion/ml/inference/plot.py:122-131 - keep the first two across calls):
//   weights: abs-max + split planes of every matrix (words 0 .. 3 of a layer's row of the abs-max table, 6 / 7 of row L)
//   context: K / V of the context rows, the fold, its split blocks g16 / v16 (words 4 / 5, scales sc[4] / sc[5])
//   steps:   K / V of the n_tok step tokens, their fold, the split step blocks (words 6 / 7, scales sc[6] / sc[7])
static int traj_prepare_weights(const sd_denoiser_weights *w, const Scratch &s, hipStream_t st) {
    const int d = w->d, L = w->L;
    int rc = zero_word_cols(s.maxbits, L, 0, 4, st);
    if (!rc) rc = zero_word_cols(s.maxbits + L * 8, 1, 6, 2, st);
    if (rc) return rc;
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        const float *mats[4] = {lw.sa_out_w, lw.lin1_w, lw.lin2_w, lw.sa_in_w};
        const int rows[4] = {d, d, d, 3 * d};
        for (int m = 0; m < 4; ++m) {
            SD_LAUNCH(f16_absmax_kernel, dim3(grid_for((long)rows[m] * d)), dim3(256), 0, st, mats[m], (long)rows[m] * d, s.maxbits + l * 8 + m);
            SD_CHECK_LAUNCH("f16_absmax_kernel");
        }
    }
    SD_LAUNCH(f16_absmax_kernel, dim3(grid_for((long)d * w->J)), dim3(256), 0, st, w->emb_w, (long)d * w->J, s.maxbits + L * 8 + 6);
    SD_CHECK_LAUNCH("f16_absmax_kernel");
    SD_LAUNCH(f16_absmax_kernel, dim3(grid_for((long)d * w->J)), dim3(256), 0, st, w->out_w, (long)d * w->J, s.maxbits + L * 8 + 7);
    SD_CHECK_LAUNCH("f16_absmax_kernel");
    SD_LAUNCH(tj::pack_w16_kernel, dim3(grid_for((long)d * 4)), dim3(256), 0, st, w->emb_w, d, w->J, d, 32, s.maxbits + L * 8 + 6, 0.f, s.wio,
              s.scales + L * 8 + 6);
    SD_CHECK_LAUNCH("pack_w16_kernel");
    SD_LAUNCH(tj::pack_w16_kernel, dim3(grid_for((long)32 * d / 8)), dim3(256), 0, st, w->out_w, w->J, d, 32, d, s.maxbits + L * 8 + 7, 0.f,
              s.wio + (size_t)2 * 32 * d, s.scales + L * 8 + 7);
    SD_CHECK_LAUNCH("pack_w16_kernel");
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        const float *mats[4] = {lw.sa_out_w, lw.lin1_w, lw.lin2_w, lw.sa_in_w};
        const int rows[4] = {d, d, d, 3 * d};
        for (int m = 0; m < 4; ++m) {
            SD_LAUNCH(tj::pack_w16_kernel, dim3(grid_for((long)rows[m] * d / 8)), dim3(256), 0, st, mats[m], rows[m], d, rows[m], d, s.maxbits + l * 8 + m, 0.f,
                      f16_wf(s, l, d, m), s.scales + l * 8 + m);
            SD_CHECK_LAUNCH("pack_w16_kernel");
        }
    }
    return 0;
}

static int traj_prepare_ctx(const sd_denoiser_weights *w, const Scratch &s, const float *ctx, int B, int Mc, hipStream_t st) {
    const int d = w->d, L = w->L, hd = d / 4, nkt = key_tiles(Mc + 1);
    const size_t gvstride = (size_t)B * nkt * 64 * 2 * d, cbstride = (size_t)B * nkt * 64;
    const size_t lds = 2 * (size_t)FOLD_RB * hd * sizeof(float);
    int rc = zero_async(s.gv, L * gvstride * sizeof(float), st);   // unused key slots must be finite
    if (!rc) rc = zero_async(s.cb, L * cbstride * sizeof(float), st);
    if (!rc) rc = zero_word_cols(s.maxbits, L, 4, 2, st);
    if (rc) return rc;
    const size_t blk = (size_t)32 * d;   // halfs per (trajectory, head)
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        unsigned *mb = s.maxbits + l * 8;
        if (Mc > 0) {
            const long rows = (long)B * Mc;
            float *kvl = s.kvtmp + (size_t)l * B * Mc * 2 * d;
            rc = linear(ctx, lw.ca_in_w + (size_t)d * d, lw.ca_in_b + d, nullptr, nullptr, nullptr, kvl, B * Mc, 2 * d, d, 0, st, 0);
            if (rc) return rc;
            SD_LAUNCH(xattn_fold_kernel, dim3((unsigned)((rows + FOLD_RB - 1) / FOLD_RB), 4), dim3(256), lds, st, kvl, rows, Mc, lw.ca_in_w, lw.ca_in_b,
                      lw.ca_out_w, s.gv + l * gvstride, s.cb + l * cbstride, 64L * nkt, 16, 0, d, hd, mb + 4, mb + 5);
            SD_CHECK_LAUNCH("xattn_fold_kernel");
        }
        SD_LAUNCH(tj::pack_g16_kernel, dim3(grid_for((long)B * nkt * 4 * 16 * d / 8)), dim3(256), 0, st, s.gv + l * gvstride, (long)B * nkt, Mc, mb + 4,
                  s.g16 + (size_t)l * B * nkt * 4 * blk, s.scales + l * 8 + 4, nkt);
        SD_CHECK_LAUNCH("pack_g16_kernel");
        SD_LAUNCH(tj::pack_v16_kernel, dim3(grid_for((long)B * nkt * 16 * 2 * 64)), dim3(256), 0, st, s.gv + l * gvstride, (long)B * nkt, Mc, mb + 5,
                  s.v16 + (size_t)l * B * nkt * 4 * blk, s.scales + l * 8 + 5, nkt);
        SD_CHECK_LAUNCH("pack_v16_kernel");
    }
    return 0;
}

// ---- the step tokens' part of the preparation, all layers in one launch each (a forward_with_context call of the reference's loop pays
// it once per call: four launches instead of four per layer)
// map[b] = 0 where step token b equals token 0 bit for bit (the usual call of the reference's loop: one timestep for the whole batch, passed
// as a (B,) tensor), else b: trajectory b reads the folded blocks of token map[b], and only tokens with map[b] == b are folded and packed
__global__ void step_map_kernel(const float *__restrict__ tokens, int n_tok, int d, int *map) {
    const int b = blockIdx.x;
    bool same = true;
    for (int k = threadIdx.x; k < d; k += blockDim.x)
        same &= __builtin_bit_cast(unsigned, tokens[(long)b * d + k]) == __builtin_bit_cast(unsigned, tokens[k]);
    same = __syncthreads_and(same);
    if (threadIdx.x == 0) map[b] = same ? 0 : b;
}
struct StepFoldArgs { const float *wkv[tj::MAX_L], *bkv[tj::MAX_L], *wq[tj::MAX_L], *bq[tj::MAX_L], *woc[tj::MAX_L]; };
// hidden_dim 256.  step_kv_kernel, grid (n_tok, L, 8), 256 threads: rows 64 z .. 64 z + 63 of K | V = Wkv tok + bkv (memory rows are not
// layer-normed) -> kvstep[l][tok][2 D].  step_fold_all_kernel, grid (n_tok, L, 4 heads): the fold of xattn_fold_kernel for this one row and head:
// G_h = Wq_h^T K_h, V'_h = Woc_h V_h, c_h = bq_h . K_h -> gvstep row [tok * 4 + h][2 D], cstep [tok * 4 + h]; abs-max of G / V' -> words 6 / 7 of
// the layer's row.  (One workgroup per (token, layer) did all of it as a chain of dependent weight loads: 130 us for ONE token - what every
// forward_with_context call of the reference's loop pays, 27 % of a B = 256 rollout; 16 rows in flight per wave: 100 us; the work of a token
// and layer spread over 8 + 4 workgroups: see NOTEBOOK round 5.)
__global__ __launch_bounds__(256) void step_kv_kernel(StepFoldArgs a, const float *__restrict__ tokens, float *__restrict__ kvstep, long n_tok,
                                                      const int *__restrict__ map) {
    constexpr int D = tj::D;
    const int l = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long tok = blockIdx.x;
    if (map && map[tok] != (int)tok) return;   // a duplicate of token 0: nobody reads its blocks
    const f32x4 t4 = *reinterpret_cast<const f32x4 *>(tokens + tok * D + 4 * lane);
    const float *wkv = a.wkv[l], *bkv = a.bkv[l];
    const int o0 = blockIdx.z * 64 + wv * 16;
    f32x4 w4[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w4[u] = *reinterpret_cast<const f32x4 *>(wkv + (long)(o0 + u) * D + 4 * lane);
    float *out = kvstep + ((long)l * n_tok + tok) * 2 * D;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const float s = wave_sum((w4[u][0] * t4[0] + w4[u][1] * t4[1]) + (w4[u][2] * t4[2] + w4[u][3] * t4[3]));
        if (lane == 0) out[o0 + u] = s + bkv[o0 + u];
    }
}
__global__ __launch_bounds__(256) void step_fold_all_kernel(StepFoldArgs a, const float *__restrict__ kvstep, long n_tok, float *__restrict__ gvstep,
                                                            long gv_layer_stride, float *__restrict__ cstep, long c_layer_stride,
                                                            unsigned *maxbits, const int *__restrict__ map) {
    constexpr int D = tj::D, HD = tj::HD;
    __shared__ __attribute__((aligned(16))) float sk[HD], sv[HD];
    const int l = blockIdx.y, h = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long tok = blockIdx.x;
    if (map && map[tok] != (int)tok) return;
    const int n = threadIdx.x;
    const float *wq = a.wq[l], *woc = a.woc[l] + (long)n * D;
    // the head's weights first (64 + 16 loads per thread in flight), then its K / V slice
    float wqv[HD];
    f32x4 wov[HD / 4];
#pragma unroll
    for (int j = 0; j < HD; ++j) wqv[j] = wq[(long)(h * HD + j) * D + n];
#pragma unroll
    for (int j = 0; j < HD / 4; ++j) wov[j] = *reinterpret_cast<const f32x4 *>(woc + h * HD + 4 * j);
    const float *kv = kvstep + ((long)l * n_tok + tok) * 2 * D;
    if (threadIdx.x < HD) sk[threadIdx.x] = kv[h * HD + threadIdx.x];
    else if (threadIdx.x < 2 * HD) sv[threadIdx.x - HD] = kv[D + h * HD + threadIdx.x - HD];
    __syncthreads();
    if (wv == 0) {   // the score bias of head h
        const float c = wave_sum(a.bq[l][h * HD + lane] * sk[lane]);
        if (lane == 0) cstep[l * c_layer_stride + tok * 4 + h] = c;
    }
    float g = 0.f, v = 0.f;
#pragma unroll
    for (int j = 0; j < HD; ++j) g += wqv[j] * sk[j];
#pragma unroll
    for (int j = 0; j < HD / 4; ++j) {
        const f32x4 v4 = *reinterpret_cast<const f32x4 *>(sv + 4 * j);
        v += (wov[j][0] * v4[0] + wov[j][1] * v4[1]) + (wov[j][2] * v4[2] + wov[j][3] * v4[3]);
    }
    float *out = gvstep + l * gv_layer_stride + tok * 4 * 2 * D;
    out[(long)h * 2 * D + n] = g;
    out[(long)h * 2 * D + D + n] = v;
    float mg = fabsf(g), mv = fabsf(v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mg = fmaxf(mg, __shfl_xor(mg, o, 64));
        mv = fmaxf(mv, __shfl_xor(mv, o, 64));
    }
    if (lane == 0) {
        const unsigned bg = __builtin_bit_cast(unsigned, mg), bv = __builtin_bit_cast(unsigned, mv);
        if (bg > __atomic_load_n(maxbits + l * 8 + 6, __ATOMIC_RELAXED)) atomicMax(maxbits + l * 8 + 6, bg);
        if (bv > __atomic_load_n(maxbits + l * 8 + 7, __ATOMIC_RELAXED)) atomicMax(maxbits + l * 8 + 7, bv);
    }
}
// grid (blocks, L): tj::pack_gstep16_kernel + tj::pack_vstep16_kernel of every layer (scales from words 6 / 7 -> sc[6], sc[7]); with
// no_ctx also sc[4] = sc[6], sc[5] = sc[7] (no context rows: the all-zero context blocks carry no scale of their own - a scale of 1 from an
// abs-max of 0 would drag the common value scale of tj::step_scale down to 1)
__global__ void pack_step16_all_kernel(const float *__restrict__ gvstep, long gv_layer_stride, long n_tok, const unsigned *maxbits,
                                       f16 *__restrict__ gdst, long g_layer_stride, f16 *__restrict__ vdst, long v_layer_stride, float *scales,
                                       int no_ctx, const int *__restrict__ map) {
    constexpr int D = tj::D;
    const int l = blockIdx.y;
    const float sg = f16_scale_from_bits(maxbits[l * 8 + 6]), sv = f16_scale_from_bits(maxbits[l * 8 + 7]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scales[l * 8 + 6] = sg;
        scales[l * 8 + 7] = sv;
        if (no_ctx) {
            scales[l * 8 + 4] = sg;
            scales[l * 8 + 5] = sv;
        }
    }
    const float *src = gvstep + l * gv_layer_stride;
    f16 *gd = gdst + l * g_layer_stride, *vd = vdst + l * v_layer_stride;
    const long ng = n_tok * 4 * (D / 8), nv = n_tok * 4 * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ng + nv; i += (long)gridDim.x * blockDim.x) {
        if (i < ng) {
            const int k8 = (int)(i % (D / 8));
            const long ih = i / (D / 8);
            if (map && map[ih >> 2] != (int)(ih >> 2)) continue;
            f16x4 h0, l0, h1, l1;
            const float *row = src + ih * 2 * D + tj::kperm(k8, 0);
            f16_split4(*reinterpret_cast<const f32x4 *>(row), sg, h0, l0);
            f16_split4(*reinterpret_cast<const f32x4 *>(row + 16), sg, h1, l1);
            f16 *o = gd + ih * (8 * 2 * 32) + ((k8 >> 2) * 2) * 32 + (k8 & 3) * 8;
            *reinterpret_cast<f16x4 *>(o) = h0;
            *reinterpret_cast<f16x4 *>(o + 4) = h1;
            *reinterpret_cast<f16x4 *>(o + 32) = l0;
            *reinterpret_cast<f16x4 *>(o + 36) = l1;
        } else {
            const long j = i - ng;
            const int n = (int)(j % D), head = (int)((j / D) & 3);
            const long item = j / D / 4;
            if (map && map[item] != (int)item) continue;
            const float v = src[(item * 4 + head) * 2 * D + D + n] * sv;
            const f16 h = (f16)v;
            vd[(item * 2 + 0) * 4 * D + head * D + n] = h;
            vd[(item * 2 + 1) * 4 * D + head * D + n] = (f16)(v - (float)h);
        }
    }
}

// n_tok step tokens (rows of `tokens`): one per DDIM step of a rollout, or one per trajectory of a single evaluation
static int traj_prepare_steps(const sd_denoiser_weights *w, const Scratch &s, const float *tokens, int n_tok, int Mc, hipStream_t st,
                              bool per_traj = false) {
    const int d = w->d, L = w->L;
    const size_t gvsstride = (size_t)n_tok * 4 * 2 * d, cssstride = (size_t)n_tok * 4;
    const size_t blk = (size_t)32 * d;
    int rc = zero_word_cols(s.maxbits, L, 6, 2, st);
    if (rc) return rc;
    const int *map = nullptr;
    if (per_traj) {   // one token per trajectory: fold the distinct ones only (see step_map_kernel)
        SD_LAUNCH(step_map_kernel, dim3((unsigned)n_tok), dim3(64), 0, st, tokens, n_tok, d, s.stepmap);
        SD_CHECK_LAUNCH("step_map_kernel");
        map = s.stepmap;
    }
    StepFoldArgs fa{};
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        fa.wkv[l] = lw.ca_in_w + (size_t)d * d; fa.bkv[l] = lw.ca_in_b + d;
        fa.wq[l] = lw.ca_in_w; fa.bq[l] = lw.ca_in_b; fa.woc[l] = lw.ca_out_w;
    }
    SD_LAUNCH(step_kv_kernel, dim3((unsigned)n_tok, (unsigned)L, 8), dim3(256), 0, st, fa, tokens, s.kvstep, (long)n_tok, map);
    SD_CHECK_LAUNCH("step_kv_kernel");
    SD_LAUNCH(step_fold_all_kernel, dim3((unsigned)n_tok, (unsigned)L, 4), dim3(256), 0, st, fa, s.kvstep, (long)n_tok, s.gvstep, (long)gvsstride, s.cstep,
              (long)cssstride, s.maxbits, map);
    SD_CHECK_LAUNCH("step_fold_all_kernel");
    // per-layer regions as carved for mode 2 (n_tok * 4 * blk / n_tok * blk halfs), the step blocks packed densely inside
    unsigned gx = grid_for((long)n_tok * 4 * (d / 8 + d));
    SD_LAUNCH(pack_step16_all_kernel, dim3(gx, (unsigned)L), dim3(256), 0, st, s.gvstep, (long)gvsstride, (long)n_tok, s.maxbits, s.gstep16,
              (long)((size_t)n_tok * 4 * blk), s.vstep16, (long)((size_t)n_tok * blk), s.scales, Mc == 0 ? 1 : 0, map);
    SD_CHECK_LAUNCH("pack_step16_all_kernel");
    return 0;
}

// one denoiser step + DDIM update in ONE launch (step index i selects the step-token blocks)
// coef NULL: no DDIM update (x is only read); per_traj: trajectory b reads step block b of the n_steps prepared ones (i = 0)
static int decoder_step_traj(const sd_denoiser_weights *w, float *x, const Scratch &s, int B, int T, int Mc, int i, int n_steps,
                             const float *coef, hipStream_t st, bool precise, float *eps = nullptr, int32_t *status = nullptr,
                             bool per_traj = false) {
    const int d = w->d, L = w->L, nkt = key_tiles(Mc + 1);
    const size_t blk = (size_t)32 * d, cbstride = (size_t)B * nkt * 64;
    tj::StepArgs a{};
    a.nkt = nkt;
    a.status = status;
    a.x = x;
    a.eps_out = eps;
    a.w_emb = s.wio;
    a.b_emb = w->emb_b;
    a.pe = w->pe;
    a.n1_w = w->layers[0].n1_w;
    a.n1_b = w->layers[0].n1_b;
    a.w_out = s.wio + (size_t)2 * 32 * d;
    a.b_out = w->out_b;
    a.sc_io = s.scales + L * 8 + 6;
    if (coef) { a.c0 = coef[0]; a.c1 = coef[1]; a.c2 = coef[2]; a.c3 = coef[3]; }
    a.scale_log2e = (1.0f / sqrtf((float)(d / w->heads))) * 1.44269504088896340736f;
    a.T = T; a.B = B; a.J = w->J; a.L = L; a.Mk = Mc + 1; a.update_x = coef ? 1 : 0;
    a.step_per_traj = per_traj ? 1 : 0;
    a.step_map = per_traj ? s.stepmap : nullptr;
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        tj::LayerW &q = a.layer[l];
        q.n2_w = lw.n2_w; q.n2_b = lw.n2_b; q.n3_w = lw.n3_w; q.n3_b = lw.n3_b;
        q.w_o = f16_wf(s, l, d, 0); q.w_1 = f16_wf(s, l, d, 1); q.w_2 = f16_wf(s, l, d, 2); q.w_in = f16_wf(s, l, d, 3);
        q.b_in = lw.sa_in_b; q.b_o = lw.sa_out_b; q.b_1 = lw.lin1_b; q.b_2 = lw.lin2_b; q.b_oc = lw.ca_out_b;
        q.sc = s.scales + l * 8;
        q.g16 = s.g16 + (size_t)l * B * nkt * 4 * blk;
        q.v16 = s.v16 + (size_t)l * B * nkt * 4 * blk;
        q.cb = s.cb + l * cbstride;
        // per-layer regions as carved for mode 2 (n_steps * 4 * blk / n_steps * blk halfs), the step blocks packed densely inside
        q.gstep = s.gstep16 + (size_t)l * n_steps * 4 * blk + (size_t)i * (4 * 8 * 2 * 32);
        q.vstep = s.vstep16 + (size_t)l * n_steps * blk + (size_t)i * (2 * 4 * d);
        q.cstep = s.cstep + ((size_t)l * n_steps + i) * 4;
        q.nln_w = l + 1 < L ? w->layers[l + 1].n1_w : nullptr;
        q.nln_b = l + 1 < L ? w->layers[l + 1].n1_b : nullptr;
    }
    const int ntt = (T + 15) / 16;
    // more than 16 memory rows: the wide instantiation (three products everywhere, whatever the mode asked for)
    const int variant = nkt > 1 ? 2 : (precise ? 1 : 0);
    const TrajStepFn fn = variant == 2 ? traj_step_wide_fn(ntt) : (precise ? traj_step_fn<true>(ntt) : traj_step_fn<false>(ntt));
    if (!fn) return fail(SD_E_BADARG, "traj_step_kernel: horizon out of range");
    ProfScope prof(SD_KCLASS_TRAJ_STEP, st);
    static DevFlag attr_set[3][8];
    if (!attr_set[variant][ntt]) {
        const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, tj::LDS_BYTES);
        if (e != hipSuccess) return fail((int)e, "traj_step_kernel: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        attr_set[variant][ntt] = true;
    }
    SD_LAUNCH(fn, dim3((unsigned)B), dim3(tj::NTHREADS), (size_t)tj::LDS_BYTES, st, a);
    SD_CHECK_LAUNCH("traj_step_kernel");
    return 0;
}

// Encoder stack (self-attention + FFN layers): chain B with norm2 as the FFN norm.
static int encoder_stack(const sd_layer_weights *layers, int L, const Scratch &s, int B, int n, int d, int heads,
                         hipStream_t st) {
    const long R = (long)B * n;
    int rc = linear(s.h, layers[0].sa_in_w, layers[0].sa_in_b, layers[0].n1_w, layers[0].n1_b, nullptr, s.qkv, (int)R,
                    3 * d, d, 0, st);
    if (rc) return rc;
    for (int l = 0; l < L; ++l) {
        const sd_layer_weights &lw = layers[l];
        rc = attention(s.qkv, 3 * d, s.qkv + d, s.qkv + 2 * d, 3 * d, nullptr, nullptr, s.a, d, B, n, n, d, heads, st);
        if (rc) return rc;
        const sd_layer_weights *nx = (l + 1 == L) ? nullptr : &layers[l + 1];
        ChainBArgs gb{s.a, s.h, lw.sa_out_w, lw.sa_out_b, lw.n2_w, lw.n2_b, lw.lin1_w, lw.lin1_b, lw.lin2_w, lw.lin2_b,
                      nx ? nx->n1_w : nullptr, nx ? nx->n1_b : nullptr, nx ? nx->sa_in_w : nullptr,
                      nx ? nx->sa_in_b : nullptr, s.qkv, R};
        if ((rc = chain_b(gb, d, st))) return rc;
    }
    return 0;
}

static int check_denoiser(const sd_denoiser_weights *w) {
    if (!w || !w->layers || !w->emb_w || !w->emb_b || !w->out_w || !w->out_b || !w->pe)
        return fail(SD_E_BADARG, "denoiser weights: null pointer");
    if (w->d != 64 && w->d != 128 && w->d != 256 && w->d != 512)
        return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
    if (w->heads <= 0 || w->d % w->heads != 0) return fail(SD_E_BADDIM, "hidden_dim not divisible by heads");
    if (w->L <= 0 || w->J <= 0 || w->J > SD_MAX_J) return fail(SD_E_BADARG, "denoiser weights: bad L or J");
    return 0;
}

extern "C" int sd_denoiser_forward(const sd_denoiser_weights *w, const float *x, const float *memory,
                                   float *eps_out, float *workspace, int B, int T, int M, void *stream) {
    int rc = check_denoiser(w);
    if (rc) return rc;
    if (!x || !memory || !eps_out || !workspace || B <= 0 || T <= 0 || M <= 0)
        return fail(SD_E_BADARG, "sd_denoiser_forward: null pointer or empty shape");
    if (T > w->T_max) return fail(SD_E_TOOBIG, "sd_denoiser_forward: horizon exceeds positional table");
    hipStream_t st = (hipStream_t)stream;
    const int d = w->d, R = B * T;
    Scratch s = carve(workspace, R, (long)B * M, d, w->L, 0);
    for (int l = 0; l < w->L; ++l) {
        // memory is NOT layer-normed: K = mem Wk^T + bk, V = mem Wv^T + bv (rows [d:3d) of in_proj)
        const sd_layer_weights &lw = w->layers[l];
        rc = linear(memory, lw.ca_in_w + (size_t)d * d, lw.ca_in_b + d, nullptr, nullptr, nullptr,
                    s.kv + (size_t)l * B * M * 2 * d, B * M, 2 * d, d, 0, st);
        if (rc) return rc;
    }
    const float *kvbase = s.kv;
    const size_t kvstride = (size_t)B * M * 2 * d;
    return decoder_stack(w, x, s, B, T, M, [=](int l) { return kvbase + l * kvstride; }, TailArgs{eps_out, nullptr, nullptr}, st);
}

extern "C" int sd_encoder_forward(const sd_encoder_weights *w, const float *x, float *out, float *workspace, int B,
                                  int S, void *stream) {
    if (!w || !w->layers || !w->emb_w || !w->emb_b || !w->pe || !x || !out || !workspace || B <= 0 || S <= 0)
        return fail(SD_E_BADARG, "sd_encoder_forward: null pointer or empty shape");
    if (w->d != 64 && w->d != 128 && w->d != 256 && w->d != 512)
        return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
    if (w->p <= 0 || S / w->p <= 0 || S / w->p > w->S_max)
        return fail(SD_E_TOOBIG, "sd_encoder_forward: sequence exceeds positional table");
    hipStream_t st = (hipStream_t)stream;
    const int d = w->d, n = S / w->p, R = B * n;
    Scratch s = carve(workspace, R, 1, d, 1, 0);
    int rc = patch_embed(x, w->emb_w, w->emb_b, w->pe, s.h, B, S, w->C, w->p, d, st);
    if (rc) return rc;
    if ((rc = encoder_stack(w->layers, w->L, s, B, n, d, w->heads, st))) return rc;
    SD_LAUNCH(copy_rows_kernel, dim3(grid_for((long)R * d)), dim3(256), 0, st, s.h, (long)d, out, (long)d,
                       (long)R, d);
    SD_CHECK_LAUNCH("copy_rows_kernel");
    return 0;
}

extern "C" int sd_sampler_mode(int d, int heads, int T, int Mc, int J) {
    const int Mk = Mc + 1;
    if (traj_ok(d, heads, T, Mk, J, 1) || trajg_ok(d, heads, T, Mk, J, 1)) return 3;   // (the layer count is checked at the call: <= 8; mode 4 is opt-in)
    if (!(fold_ok(d, heads, T, Mk) && fused_layer_ok(d, heads, T, Mk))) return 0;
    if (!f16_ok(d, J)) return 1;
    return 2;
}

extern "C" int sd_ddim_sample(const sd_denoiser_weights *w, const float *ctx, const float *step_tokens,
                              const float *coef, float *x, float *trace, float *workspace, int B, int T, int Mc,
                              int n_steps, void *stream) {
    return sd_ddim_sample_eps(w, ctx, step_tokens, coef, x, trace, nullptr, workspace, B, T, Mc, n_steps, nullptr, -1, stream);
}

extern "C" int sd_ddim_sample_ex(const sd_denoiser_weights *w, const float *ctx, const float *step_tokens,
                                 const float *coef, float *x, float *trace, float *workspace, int B, int T, int Mc,
                                 int n_steps, int32_t *status, int max_mode, void *stream) {
    return sd_ddim_sample_eps(w, ctx, step_tokens, coef, x, trace, nullptr, workspace, B, T, Mc, n_steps, status, max_mode, stream);
}

extern "C" int sd_ddim_sample_eps(const sd_denoiser_weights *w, const float *ctx, const float *step_tokens,
                                  const float *coef, float *x, float *trace, float *eps_trace, float *workspace, int B, int T, int Mc,
                                  int n_steps, int32_t *status, int max_mode, void *stream) {
    int rc = check_denoiser(w);
    if (rc) return rc;
    if (!step_tokens || !coef || !x || !workspace || B <= 0 || T <= 0 || Mc < 0 || n_steps <= 0 || (Mc > 0 && !ctx))
        return fail(SD_E_BADARG, "sd_ddim_sample: null pointer or empty shape");
    if (max_mode < -1 || max_mode > 4) return fail(SD_E_BADARG, "sd_ddim_sample_ex: max_mode must be -1, 0, 1, 2, 3 or 4");
    // automatic = the highest mode that is valid for ANY weights: 3.  Mode 4's two-product Q | K | V site is opt-in (max_mode = 4):
    // it is validated up to SD_SHARP_LOGIT_LIMIT and reports through `status` when a logit leaves that range
    if (max_mode < 0) max_mode = 3;
    if (max_mode == 4 && !status) return fail(SD_E_BADARG, "sd_ddim_sample_ex: max_mode 4 needs a status word (SD_STATUS_SHARP_LOGITS)");
    if (T > w->T_max) return fail(SD_E_TOOBIG, "sd_ddim_sample: horizon exceeds positional table");
    hipStream_t st = (hipStream_t)stream;
    const int d = w->d, R = B * T, L = w->L;
    Scratch s = carve(workspace, R, (long)B * (Mc + 1), d, L, n_steps, B);
    // once per rollout: K/V of the context rows (placed as rows 0..Mc-1 of each trajectory's
    // [Mk][2d] block, Mk = Mc + 1) and of all n_steps step tokens, per layer
    const int Mk = Mc + 1;
    const size_t kvstride = (size_t)B * Mk * 2 * d, kvsstride = (size_t)n_steps * 2 * d;
    const bool small = (long)B * Mk * 2 * d < (1L << 30);
    // the trajectory kernel (modes 3 / 4) takes any horizon <= 100: it needs the folded blocks, not the row-panel kernels' T >= 64
    const bool traj = max_mode >= 3 && small && traj_ok(d, w->heads, T, Mk, w->J, L) && s.wf != nullptr && s.wio != nullptr;
    // ... and the generic trajectory kernels (sd_trajg.hip) every other hidden_dim 128 / 256 / 512 shape, whatever its memory length
    const bool trajg = !traj && max_mode >= 3 && trajg_ok(d, w->heads, T, Mk, w->J, L) && s.gws != nullptr;
    if (trajg) {
        if (status) {
            if (int rz = zero_async(status, sizeof(int32_t), st)) return rz;
        }
        if ((rc = trajg_prepare_weights(w, s.gws, B, Mc, n_steps, st))) return rc;
        if ((rc = trajg_prepare_ctx(w, s.gws, ctx, s.kvtmp, B, Mc, n_steps, st))) return rc;
        if ((rc = trajg_prepare_steps(w, s.gws, step_tokens, s.kvstep, B, Mc, n_steps, st))) return rc;
        for (int i = 0; i < n_steps; ++i) {
            float *eps_i = eps_trace ? eps_trace + (size_t)i * R * w->J : nullptr;
            if ((rc = trajg_step(w, s.gws, x, eps_i, B, T, Mc, i, n_steps, coef + 4 * i, false, st))) return rc;
            if (trace) {
                const long n = (long)R * w->J;
                SD_LAUNCH(copy_rows_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, trace + (size_t)i * n, n, 1L, (int)n);
                SD_CHECK_LAUNCH("copy_rows_kernel");
            }
        }
        if (status) {
            const long n = (long)R * w->J;
            SD_LAUNCH(finite_check_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, status, SD_STATUS_NONFINITE);
            SD_CHECK_LAUNCH("finite_check_kernel");
        }
        return 0;
    }
    const bool fold = traj || (max_mode >= 1 && fold_ok(d, w->heads, T, Mk) && fused_layer_ok(d, w->heads, T, Mk) && small);
    const int nkt = traj ? key_tiles(Mk) : 1;   // (the row-panel fold: Mk <= 16)
    const size_t gvstride = (size_t)B * nkt * 64 * 2 * d, cbstride = (size_t)B * nkt * 64;
    const size_t gvsstride = (size_t)n_steps * 4 * 2 * d, cssstride = (size_t)n_steps * 4;
    const bool f16 = traj || (max_mode >= 2 && fold && f16_ok(d, w->J) && s.wf != nullptr);
    const bool chain16 = max_mode >= 2 && !fold && chain16_ok(d, w->J) && s.wfc != nullptr && !fused_layer_ok(d, w->heads, T, Mk);
    if (status) {
        if (int rz = zero_async(status, sizeof(int32_t), st)) return rz;
    }
    if (traj) {
        if ((rc = traj_prepare_weights(w, s, st))) return rc;
        if ((rc = traj_prepare_ctx(w, s, ctx, B, Mc, st))) return rc;
        if ((rc = traj_prepare_steps(w, s, step_tokens, n_steps, Mc, st))) return rc;
    }
    for (int l = 0; l < L && !traj; ++l) {
        const sd_layer_weights &lw = w->layers[l];
        const float *wkv = lw.ca_in_w + (size_t)d * d, *bkv = lw.ca_in_b + d;
        auto lin = max_mode >= 2 ? linear : linear32;
        if (Mc > 0) {
            rc = lin(ctx, wkv, bkv, nullptr, nullptr, nullptr, s.kvtmp + (size_t)l * B * Mc * 2 * d, B * Mc, 2 * d, d, 0, st, 0);
            if (rc) return rc;
        }
        rc = lin(step_tokens, wkv, bkv, nullptr, nullptr, nullptr, s.kvstep + (size_t)l * kvsstride, n_steps, 2 * d, d, 0, st, 0);
        if (rc) return rc;
    }
    if (chain16 && (rc = f16_prepare_chain(w, s, st))) return rc;
    if (traj) {
    } else if (fold) {
        // the memory is fixed over the rollout: fold Wq into its keys and Woc into its values once
        const int hd = d / 4;
        const size_t lds = 2 * (size_t)FOLD_RB * hd * sizeof(float);
        int rz = zero_async(s.gv, L * gvstride * sizeof(float), st);   // unused key slots must be finite
        if (!rz && f16) rz = zero_async(s.maxbits, (size_t)(L + 1) * 8 * sizeof(unsigned), st);   // abs-max words
        if (!rz) rz = zero_async(s.cb, L * cbstride * sizeof(float), st);
        if (rz) return rz;
        for (int l = 0; l < L; ++l) {
            const sd_layer_weights &lw = w->layers[l];
            if (Mc > 0) {
                const long rows = (long)B * Mc;
                SD_LAUNCH(xattn_fold_kernel, dim3((unsigned)((rows + FOLD_RB - 1) / FOLD_RB), 4), dim3(256), lds, st,
                          s.kvtmp + (size_t)l * B * Mc * 2 * d, rows, Mc, lw.ca_in_w, lw.ca_in_b, lw.ca_out_w,
                          s.gv + l * gvstride, s.cb + l * cbstride, 64L * nkt, 16, 0, d, hd, f16 ? s.maxbits + l * 8 + 4 : (unsigned *)nullptr,
                          f16 ? s.maxbits + l * 8 + 5 : (unsigned *)nullptr);
                SD_CHECK_LAUNCH("xattn_fold_kernel");
            }
            SD_LAUNCH(xattn_fold_kernel, dim3((unsigned)((n_steps + FOLD_RB - 1) / FOLD_RB), 4), dim3(256), lds, st,
                      s.kvstep + (size_t)l * kvsstride, (long)n_steps, 1, lw.ca_in_w, lw.ca_in_b, lw.ca_out_w,
                      s.gvstep + l * gvsstride, s.cstep + l * cssstride, 4L, 1, 0, d, hd, f16 ? s.maxbits + l * 8 + 4 : (unsigned *)nullptr,
                      f16 ? s.maxbits + l * 8 + 5 : (unsigned *)nullptr);
            SD_CHECK_LAUNCH("xattn_fold_kernel");
        }
        if (f16 && (rc = f16_prepare(w, s, B, Mc, n_steps, st))) return rc;
    } else if (Mc > 0) {
        SD_LAUNCH(kv_place_kernel, dim3(grid_for((long)B * Mc * 2 * d), L), dim3(256), 0, st, s.kvtmp, (long)B * Mc * 2 * d, s.kv,
                  (long)kvstride, B, Mc, Mk, 2 * d, 0);
        SD_CHECK_LAUNCH("kv_place_kernel");
    }
    for (int i = 0; i < n_steps; ++i) {
        // the noise prediction of this step (the very values the DDIM update consumes), when the caller asked for them
        float *eps_i = eps_trace ? eps_trace + (size_t)i * R * w->J : nullptr;
        // this step's token row -> row Mc of every trajectory, all layers in one launch
        if (traj) {
            if ((rc = decoder_step_traj(w, x, s, B, T, Mc, i, n_steps, coef + 4 * i, st, max_mode == 3, eps_i, max_mode == 3 ? nullptr : status))) return rc;
        } else if (f16) {
            if ((rc = decoder_stack_f16(w, x, s, B, T, Mc, i, n_steps, coef + 4 * i, st, eps_i))) return rc;
        } else if (fold) {
            SD_LAUNCH(fold_place_kernel, dim3(grid_for((long)B * 4 * 2 * d), L), dim3(256), 0, st, s.gvstep + (size_t)i * 4 * 2 * d,
                      s.cstep + (size_t)i * 4, (long)gvsstride, (long)cssstride, s.gv, s.cb, (long)gvstride, (long)cbstride, B, Mc,
                      2 * d);
            SD_CHECK_LAUNCH("fold_place_kernel");
            rc = decoder_stack(w, x, s, B, T, Mk, [=](int) { return (const float *)nullptr; }, TailArgs{eps_i, x, coef + 4 * i}, st,
                               FoldArgs{s.gv, s.cb, gvstride, cbstride});
            if (rc) return rc;
        } else {
            SD_LAUNCH(kv_place_kernel, dim3(grid_for((long)B * 2 * d), L), dim3(256), 0, st, s.kvstep + (size_t)i * 2 * d, (long)kvsstride,
                      s.kv, (long)kvstride, B, Mc, Mk, 2 * d, 1);
            SD_CHECK_LAUNCH("kv_place_kernel");
            const float *kvbase = s.kv;
            rc = decoder_stack(w, x, s, B, T, Mk, [=](int l) { return kvbase + l * kvstride; }, TailArgs{eps_i, x, coef + 4 * i}, st,
                               FoldArgs{nullptr, nullptr, 0, 0}, chain16);
            if (rc) return rc;
        }
        if (trace) {
            const long n = (long)R * w->J;
            SD_LAUNCH(copy_rows_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, trace + (size_t)i * n, n,
                               1L, (int)n);
            SD_CHECK_LAUNCH("copy_rows_kernel");
        }
    }
    if (status) {
        const long n = (long)R * w->J;
        SD_LAUNCH(finite_check_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, status, SD_STATUS_NONFINITE);
        SD_CHECK_LAUNCH("finite_check_kernel");
    }
    return 0;
}

// ---- the denoiser evaluated step by step on the trajectory kernels (the reference's own loop form) --------------------------
// *generic: the shape runs on the generic trajectory kernels (sd_trajg.hip) instead of sd_traj.h's
static int sampler_eval_args(const sd_denoiser_weights *w, float *workspace, int B, int T, int Mc, int n_tok, int max_mode, const char *who,
                             Scratch *out, bool *precise, bool *generic) {
    int rc = check_denoiser(w);
    if (rc) return rc;
    if (!workspace || B <= 0 || T <= 0 || Mc < 0 || (n_tok != 1 && n_tok != B)) return fail(SD_E_BADARG, who);
    if (max_mode < -1 || max_mode > 4) return fail(SD_E_BADARG, who);
    if (T > w->T_max) return fail(SD_E_TOOBIG, who);
    if (max_mode < 0) max_mode = 3;
    const int d = w->d, Mk = Mc + 1;
    *out = carve(workspace, (long)B * T, (long)B * Mk, d, w->L, n_tok, B);
    const bool small = (long)B * Mk * 2 * d < (1L << 30);
    *generic = false;
    if (!(max_mode >= 3 && small && traj_ok(d, w->heads, T, Mk, w->J, w->L) && out->wf && out->wio)) {
        if (!(max_mode >= 3 && trajg_ok(d, w->heads, T, Mk, w->J, w->L) && out->gws)) return SD_E_UNSUPPORTED;
        *generic = true;
    }
    *precise = max_mode == 3 || *generic;
    return 0;
}

extern "C" int sd_sampler_prepare(const sd_denoiser_weights *w, const float *ctx, float *workspace, int B, int T, int Mc, int n_tok,
                                  int what, int max_mode, void *stream) {
    Scratch s;
    bool precise, generic;
    int rc = sampler_eval_args(w, workspace, B, T, Mc, n_tok, max_mode, "sd_sampler_prepare: bad argument", &s, &precise, &generic);
    if (rc) return rc;
    if ((what & ~(SD_PREPARE_WEIGHTS | SD_PREPARE_CONTEXT)) || (Mc > 0 && (what & SD_PREPARE_CONTEXT) && !ctx))
        return fail(SD_E_BADARG, "sd_sampler_prepare: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (generic) {
        if ((what & SD_PREPARE_WEIGHTS) && (rc = trajg_prepare_weights(w, s.gws, B, Mc, n_tok, st))) return rc;
        if ((what & SD_PREPARE_CONTEXT) && (rc = trajg_prepare_ctx(w, s.gws, ctx, s.kvtmp, B, Mc, n_tok, st))) return rc;
        return 0;
    }
    if ((what & SD_PREPARE_WEIGHTS) && (rc = traj_prepare_weights(w, s, st))) return rc;
    if ((what & SD_PREPARE_CONTEXT) && (rc = traj_prepare_ctx(w, s, ctx, B, Mc, st))) return rc;
    return 0;
}

extern "C" int sd_sampler_eps(const sd_denoiser_weights *w, const float *step_tokens, const float *x, float *eps, float *workspace,
                              int B, int T, int Mc, int n_tok, int32_t *status, int max_mode, void *stream) {
    Scratch s;
    bool precise, generic;
    int rc = sampler_eval_args(w, workspace, B, T, Mc, n_tok, max_mode, "sd_sampler_eps: bad argument", &s, &precise, &generic);
    if (rc) return rc;
    if (!step_tokens || !x || !eps) return fail(SD_E_BADARG, "sd_sampler_eps: null pointer");
    if (!precise && !status) return fail(SD_E_BADARG, "sd_sampler_eps: max_mode 4 needs a status word (SD_STATUS_SHARP_LOGITS)");
    hipStream_t st = (hipStream_t)stream;
    if (status) {
        if (int rz = zero_async(status, sizeof(int32_t), st)) return rz;
    }
    if (generic) {
        if (n_tok > 1) {
            SD_LAUNCH(step_map_kernel, dim3((unsigned)n_tok), dim3(64), 0, st, step_tokens, n_tok, w->d, s.stepmap);
            SD_CHECK_LAUNCH("step_map_kernel");
        }
        if ((rc = trajg_prepare_steps(w, s.gws, step_tokens, s.kvstep, B, Mc, n_tok, st, n_tok > 1 ? s.stepmap : nullptr))) return rc;
        return trajg_step(w, s.gws, const_cast<float *>(x), eps, B, T, Mc, 0, n_tok, nullptr, n_tok > 1, st, n_tok > 1 ? s.stepmap : nullptr);
    }
    if ((rc = traj_prepare_steps(w, s, step_tokens, n_tok, Mc, st, n_tok > 1))) return rc;
    // x is only read (no DDIM coefficients: no update)
    return decoder_step_traj(w, const_cast<float *>(x), s, B, T, Mc, 0, n_tok, nullptr, st, precise, eps, precise ? nullptr : status, n_tok > 1);
}

// ======================================================================================
// thin C-ABI wrappers
// ======================================================================================
extern "C" int sd_abi_version(void) { return SD_ABI_VERSION; }

extern "C" int sd_profile_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}

extern "C" int sd_profile_collect(double *ms_by_class, long *launches_by_class, int n_classes) {
    if (!ms_by_class || !launches_by_class || n_classes < SD_KCLASS_COUNT) return fail(SD_E_BADARG, "sd_profile_collect: bad argument");
    for (int i = 0; i < n_classes; ++i) {
        ms_by_class[i] = 0.0;
        launches_by_class[i] = 0;
    }
    for (const ProfRec &r : g_prof_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return fail((int)e, "sd_profile_collect: hipEventSynchronize");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return fail((int)e, "sd_profile_collect: hipEventElapsedTime");
        ms_by_class[r.cls] += ms;
        launches_by_class[r.cls] += 1;
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    return 0;
}
extern "C" const char *sd_last_error(void) { return g_last_error; }

extern "C" int sd_step_token(const void *steps, int steps_is_i64, const float *freq, const float *token, float *out,
                             long out_row_stride, int B, int d, void *stream) {
    if (!steps || !freq || !token || !out || B <= 0 || d < 8 || d % 4) return fail(SD_E_BADARG, "sd_step_token: bad argument");
    SD_LAUNCH(step_token_kernel, dim3(B), dim3(d < 256 ? 64 : 256), 0, (hipStream_t)stream, steps,
                       steps_is_i64, freq, token, out, out_row_stride, B, d);
    SD_CHECK_LAUNCH("step_token_kernel");
    return 0;
}

extern "C" int sd_game_state_embed(const int64_t *idx, const float *table, float *out, long out_row_stride, int B,
                                   int d, int n_states, void *stream) {
    if (!idx || !table || !out || B <= 0 || d <= 0 || n_states <= 0) return fail(SD_E_BADARG, "sd_game_state_embed: bad argument");
    SD_LAUNCH(gather_rows_kernel, dim3(B), dim3(d < 256 ? 64 : 256), 0, (hipStream_t)stream, idx, table, out,
                       out_row_stride, B, d, n_states);
    SD_CHECK_LAUNCH("gather_rows_kernel");
    return 0;
}

extern "C" int sd_ddim_add_noise(const float *x0, const float *noise, const int64_t *t, const float *acp, float *out,
                                 int B, int per_sample, void *stream) {
    if (!x0 || !noise || !t || !acp || !out || B <= 0 || per_sample <= 0) return fail(SD_E_BADARG, "sd_ddim_add_noise: bad argument");
    SD_LAUNCH(add_noise_kernel, dim3(grid_for((long)B * per_sample)), dim3(256), 0, (hipStream_t)stream, x0,
                       noise, t, acp, out, B, per_sample);
    SD_CHECK_LAUNCH("add_noise_kernel");
    return 0;
}

extern "C" int sd_ddim_step(const float *eps, const float *x, float *x_prev, float sqrt_a_t, float sqrt_1m_a_t,
                            float sqrt_a_prev, float sqrt_1m_a_prev, long n, void *stream) {
    if (!eps || !x || !x_prev || n <= 0) return fail(SD_E_BADARG, "sd_ddim_step: bad argument");
    SD_LAUNCH(ddim_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, eps, x, x_prev, sqrt_a_t,
                       sqrt_1m_a_t, sqrt_a_prev, sqrt_1m_a_prev, n);
    SD_CHECK_LAUNCH("ddim_step_kernel");
    return 0;
}

extern "C" int sd_normalize(const float *x, const float *mean, const float *stdv, float *out, long n, int J, int inverse,
                            void *stream) {
    if (!x || !mean || !stdv || !out || n <= 0 || J <= 0) return fail(SD_E_BADARG, "sd_normalize: bad argument");
    SD_LAUNCH(normalize_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, mean, stdv, out, n, J, inverse);
    SD_CHECK_LAUNCH("normalize_kernel");
    return 0;
}

extern "C" int sd_op_linear(const float *A, const float *W, const float *bias, const float *ln_w, const float *ln_b,
                            const float *res, float *out, int R, int N, int d, int act, void *stream) {
    if ((ln_w == nullptr) != (ln_b == nullptr)) return fail(SD_E_BADARG, "sd_op_linear: ln_w/ln_b mismatch");
    return linear(A, W, bias, ln_w, ln_b, res, out, R, N, d, act, (hipStream_t)stream);
}

extern "C" int sd_op_linear_strided(const float *A, int lda, const float *W, const float *bias, const float *ln_w,
                                    const float *ln_b, const float *res, float *out, int R, int N, int d, int act,
                                    void *stream) {
    if ((ln_w == nullptr) != (ln_b == nullptr)) return fail(SD_E_BADARG, "sd_op_linear_strided: ln_w/ln_b mismatch");
    return linear(A, W, bias, ln_w, ln_b, res, out, R, N, d, act, (hipStream_t)stream, lda);
}

// Self-attention over a packed q|k|v row buffer ([B*T][3d], head dim 64, T <= 128): the fp16x3 kernel of the sampler
static bool att_op_f16(const float *q, int ldq, const float *k, const float *v, int ldkv, int ldo, int B, int Tq, int S, int d,
                       int heads, const float *out) {
    return q && out && B > 0 && heads > 0 && d == heads * 64 && Tq == S && Tq >= 1 && Tq <= 128 && ldq == 3 * d && ldkv == 3 * d &&
           ldo == d && k == q + d && v == q + 2 * d;
}

extern "C" int sd_op_attention_lse(const float *q, int ldq, const float *k, const float *v, int ldkv, float *out,
                                   int ldo, float *lse2, int B, int Tq, int S, int d, int heads, void *stream) {
    return sd_op_attention_lse_dropout(q, ldq, k, v, ldkv, out, ldo, lse2, B, Tq, S, d, heads, 0.f, 0, 0, stream);
}

extern "C" int sd_op_attention_lse_dropout(const float *q, int ldq, const float *k, const float *v, int ldkv, float *out,
                                           int ldo, float *lse2, int B, int Tq, int S, int d, int heads, float p, uint64_t seed,
                                           uint64_t site, void *stream) {
    if (!lse2) return fail(SD_E_BADARG, "sd_op_attention_lse: lse2 is required");
    if (!(p >= 0.f) || !(p < 1.f)) return fail(SD_E_BADARG, "sd_op_attention_lse_dropout: p must be in [0, 1)");
    const DropoutArgs da = make_dropout(p, seed, site);
    if (att_op_f16(q, ldq, k, v, ldkv, ldo, B, Tq, S, d, heads, out))
        return attention_f16(q, out, B, Tq, d, heads, (hipStream_t)stream, false, lse2, da);
    return attention(q, ldq, k, v, ldkv, nullptr, nullptr, out, ldo, B, Tq, S, d, heads, (hipStream_t)stream, lse2, da);
}

extern "C" int sd_op_attention(const float *q, int ldq, const float *k, const float *v, int ldkv, const float *k_extra,
                               const float *v_extra, float *out, int ldo, int B, int Tq, int S, int d, int heads,
                               void *stream) {
    if (!k_extra && !v_extra && att_op_f16(q, ldq, k, v, ldkv, ldo, B, Tq, S, d, heads, out))
        return attention_f16(q, out, B, Tq, d, heads, (hipStream_t)stream, false);
    return attention(q, ldq, k, v, ldkv, k_extra, v_extra, out, ldo, B, Tq, S, d, heads, (hipStream_t)stream);
}

extern "C" int sd_op_patch_embed(const float *x, const float *w, const float *b, const float *pe, float *out, int B,
                                 int S, int C, int p, int d, void *stream) {
    return patch_embed(x, w, b, pe, out, B, S, C, p, d, (hipStream_t)stream);
}

extern "C" int sd_op_fc_out(const float *h, const float *W, const float *b, float *eps, float *x_io,
                            const float *coef4_host, int R, int d, int J, void *stream) {
    return fc_out(h, W, b, eps, x_io, coef4_host, R, d, J, (hipStream_t)stream);
}
