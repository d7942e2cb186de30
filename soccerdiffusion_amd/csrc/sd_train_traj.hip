// Training forward of ONE decoder layer with a workgroup per trajectory (sd_train_layer_fwd, soccerdiffusion_hip.h): what the fused path
// of rounds 2 - 3 runs as four launches - self-attention core, row chain A (out-projection, LayerNorm 2, cross-attention query), the
// cross-attention core, row chain B (out-projection, feed-forward, the next layer's LayerNorm 1 + Q | K | V projection) - in one, built
// from the phases of the sampler's trajectory kernel (sd_traj.h, TJ<NTT, true>): q | k | v, both attention outputs and the residual
// stream stay on the CU between the blocks, and B = 256 trajectories are ONE round of 256 workgroups instead of partial rounds of 400
// row panels / 1 024 attention workgroups.  Reference semantics: nn.TransformerDecoderLayer (norm_first, torch's dropout sites) as
// built by soccer_diffusion/ml/model/decoder.py:26-33 and run by ml/training/train.py:204-240.
//
// It STORES exactly the tensors the (unchanged) backward reads, in their row-major layouts: a_sa, lse_sa, h1, n2, q, a_ca, lse_ca, h2,
// nf, pre, u, h3 and the next layer's n1', q | k | v', plus the abs-max words of the grouped weight-gradient GEMM's operands; the dropout
// masks are those of sd_common.h's Philox function at the same (site, row, column) indices, so the backward regenerates them.
// Numerics: split fp16 operands, three MFMAs per product at every site, fp32 accumulation; weights in the 16 x 16 x 32 fragment order
// of sd_traj.h (pack_w16_kernel, `kperm`) with the fixed scale 2^8 of the training planes (|w| < 256); activations x ACT = 8.
#include "../../include/soccerdiffusion_hip.h"
#include "sd_common.h"
#include "sd_traj.h"

namespace tjt {
using namespace tj;

constexpr float WSC = 256.0f;   // scale of the packed weights
constexpr int VT_PITCH = 144;   // bytes of one feature row of the transposed memory values: hi 64 | lo 64 | 16 (bank spread)

struct Args {
    const float *h, *qkv;
    float *a_sa, *lse_sa, *h1, *n2, *q;
    const float *kv;
    float *a_ca, *lse_ca, *h2, *nf, *pre, *u, *h3, *nn1, *qkv2;
    const f16 *w_o, *w_q, *w_oc, *w_1, *w_2, *w_n;
    const float *b_o, *b_q, *b_oc, *b_1, *b_2, *b_n;
    const float *n2_w, *n2_b, *n3_w, *n3_b, *nn_w, *nn_b;
    DropoutArgs d_sap, d_sao, d_cap, d_cao, d_act, d_ffn;
    unsigned *ax_asa, *ax_n2, *ax_aca, *ax_nf, *ax_u, *ax_nn, *ax_out;
    int T, M, B;
    float scale_log2e;
    // head of the stack (train_head_fwd_kernel): h0 = x Wemb^T + b + pe -> h3; then the "next" LayerNorm + projection block
    const float *x_in, *b_emb, *pe;
    const f16 *w_emb;
    int J;
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ void emit_amax(unsigned *words, float m, int lane) {
    if (!words) return;
    m = wave_max(m);
    if (lane == 0 && m > 0.f) atomicMax(words + (blockIdx.x & (SD_AMAX_WORDS - 1)), __builtin_bit_cast(unsigned, m));
}
__device__ __forceinline__ float max4(const f32x4 &v) { return fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))); }

template <int NTT>
struct L {
    using K = TJ<NTT, true>;

    // y = acc * cy + bias -> dropout (site d, logical width 256) -> base + y; rows of this trajectory start at row0.  Stores to `out`.
    static __device__ __forceinline__ void residual_epilogue(const Ctx &c, f32x4 (&H)[2][NTT], const f32x4 (&Y)[2][NTT], const Bias2 &b, float cy,
                                                             const DropoutArgs &d, long row0, const float *base, float *out, float *amax) {
        if (base) {   // every load of the residual rows first (one round trip, not one per tile)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2)
                    H[a2][tt] = *reinterpret_cast<const f32x4 *>(base + (row0 + K::tok_of(c, tt)) * D + 32 * c.w + 16 * a2 + 4 * c.g);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
            if (!K::tok_ok(c, tt)) continue;
            const long row = row0 + K::tok_of(c, tt);
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) {
                const int col = 32 * c.w + 16 * a2 + 4 * c.g;
                f32x4 y = Y[a2][tt] * cy + b.v[a2];
                if (d.thresh) y = y * dropout_quad(d, (unsigned long)row * (D / 4) + (unsigned long)(col >> 2));
                const f32x4 v = H[a2][tt] + y;
                H[a2][tt] = v;
                SD_NT_STORE((f32x4)(v), reinterpret_cast<f32x4 *>(out + row * D + col));
                if (amax) *amax = fmaxf(*amax, max4(v));
            }
        }
    }

    static __device__ __forceinline__ void zero(f32x4 (&Y)[2][NTT]) {
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) Y[a2][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // q | k | v of head h of this trajectory (rows of [T][768]) -> Q, K (natural feature order) and V planes in LDS, x ACT.  Two
    // halves: every load of a head is REQUESTED first (a load - convert - store loop costs one HBM round trip per iteration), for head
    // h + 1 while head h's attention runs, and written to LDS once that head's buffers are free.
    static constexpr int QKV_VECS = (TMAX * 16 * 3 + NTHREADS - 1) / NTHREADS;   // 10
    struct QkvRegs { f32x4 v[QKV_VECS]; };
    static __device__ __forceinline__ void load_qkv(QkvRegs &r, const float *qkv, int h, int T) {
#pragma unroll
        for (int n = 0; n < QKV_VECS; ++n) {
            const int i = threadIdx.x + n * NTHREADS;
            const int which = i / (T * 16), rr = i - which * (T * 16);
            const int tok = rr >> 4, j = rr & 15;
            r.v[n] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < T * 16 * 3) r.v[n] = *reinterpret_cast<const f32x4 *>(qkv + (long)tok * (3 * D) + which * D + HD * h + 4 * j);
        }
    }
    static __device__ __forceinline__ void store_qkv(const QkvRegs &r, char *smem, int T) {
        char *Qb = smem + LDS_SQ, *Kb = smem + LDS_Q, *Vb = smem + LDS_K;
#pragma unroll
        for (int n = 0; n < QKV_VECS; ++n) {
            const int i = threadIdx.x + n * NTHREADS;
            if (i >= T * 16 * 3) continue;
            const int which = i / (T * 16), rr = i - which * (T * 16);
            const int tok = rr >> 4, j = rr & 15;
            f16x4 hi, lo;
            split4(r.v[n] * ACT, hi, lo);
            if (which < 2) {
                char *dst = which == 0 ? Qb : Kb;
                const int chunk = ((j & 7) >> 1) | ((j >> 3) << 3);
                *reinterpret_cast<f16x4 *>(dst + K::q_off(tok, chunk) + 8 * (j & 1)) = hi;
                *reinterpret_cast<f16x4 *>(dst + K::q_off(tok, chunk | 4) + 8 * (j & 1)) = lo;
            } else {
                char *at = Vb + tok * VROW + 8 * j;
                *reinterpret_cast<f16x4 *>(at) = hi;
                *reinterpret_cast<f16x4 *>(at + 128) = lo;
            }
        }
    }

    // HEAD: the stack's entry instead of a layer - embedding (nn.Linear(J -> d) + positional rows, decoder.py:48-50) -> h3, then the
    // LayerNorm 1 + Q | K | V projection of layer 0 (the block every layer runs for its successor)
    template <bool HEAD>
    static __device__ __forceinline__ void body(const Args &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        Ctx c;
        K::ctx_init(c, smem, a.T);
        const long traj = blockIdx.x, row0 = traj * a.T;
        const float cy = 1.0f / (ACT * WSC);
        f32x4 H[2][NTT], Y[2][NTT];
        float am = 0.f;
        if constexpr (HEAD) {
            // x rows -> LDS planes (k = joint, zero-padded to 32), as the sampler's step kernel stages them
            char *Qb = smem + LDS_Q;
            const int J = a.J;
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
                for (int tt = 0; tt < NTT; ++tt) pe4[n][tt] = *reinterpret_cast<const f32x4 *>(a.pe + (long)K::tok_of(c, tt) * D + n0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float *xr = a.x_in + row0 * J;
            const int nvec = a.T * J / 4;
            for (int i = threadIdx.x; i < nvec; i += NTHREADS) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xr + 4 * i);
                const int tok = (4 * i) / J, j0 = 4 * i - tok * J;
                const int chunk = j0 >> 3;
                split_store(Qb + K::q_off(tok, chunk) + 2 * (j0 & 7), Qb + K::q_off(tok, chunk | 4) + 2 * (j0 & 7), v);
                if (j0 + 4 >= J) {
                    const f16 z = (f16)0.f;
                    const f16x4 z4 = {z, z, z, z};
                    for (int k = J; k < 32; k += 4) {
                        *reinterpret_cast<f16x4 *>(Qb + K::q_off(tok, k >> 3) + 2 * (k & 7)) = z4;
                        *reinterpret_cast<f16x4 *>(Qb + K::q_off(tok, (k >> 3) | 4) + 2 * (k & 7)) = z4;
                    }
                }
            }
            __syncthreads();
            const float c_e = 1.0f / WSC;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    mma3<S_EMB>(acc, we[n][0], we[n][1], K::lds16(Qb + K::q_at(c, tt, 0, 0)), K::lds16(Qb + K::q_at(c, tt, 1, 0)));
                    H[n][tt] = acc * c_e + (be[n] + pe4[n][tt]);
                    if (K::tok_ok(c, tt))
                        SD_NT_STORE(H[n][tt], reinterpret_cast<f32x4 *>(a.h3 + (row0 + K::tok_of(c, tt)) * D + 32 * c.w + 16 * n + 4 * c.g));
                }
        } else {
        // =========================== self-attention block ===========================
        zero(Y);
        float am_a = 0.f;
        QkvRegs qr;
        load_qkv(qr, a.qkv + row0 * (3 * D), 0, a.T);
#pragma unroll 1
        for (int h = 0; h < NH; ++h) {
            store_qkv(qr, smem, a.T);
            __syncthreads();   // Q, K, V of head h staged (and every reader of head h-1's O is past its out-projection)
            if (h + 1 < NH) load_qkv(qr, a.qkv + row0 * (3 * D), h + 1, a.T);   // lands while this head's attention runs
            __builtin_amdgcn_sched_barrier(0);
            if (c.w < NTT) {
                const Ctx cl = K::ctx_local(c);
                const SaW sw{nullptr, nullptr, nullptr, 1.f, a.scale_log2e, nullptr};
                f32x4 S[NTT];
                float psum, lse;
                K::att_scores(cl, sw, smem + LDS_SQ, smem + LDS_Q, S, psum, &lse);
                const int qtok = 16 * cl.w + cl.t;
                if (qtok < a.T) {
                    const long mrow = (traj * NH + h) * a.T + qtok;
                    if (cl.g == 0) a.lse_sa[mrow] = lse;
                    if (a.d_sap.thresh) {   // dropout on the probabilities: keys 16 kt + 4 g + r are one quad
                        const unsigned long wq = (unsigned long)((a.T + 3) >> 2);
#pragma unroll
                        for (int kt = 0; kt < NTT; ++kt)
                            if (16 * kt + 4 * cl.g < a.T) S[kt] = S[kt] * dropout_quad(a.d_sap, (unsigned long)mrow * wq + (unsigned long)(4 * kt + cl.g));
                    }
                }
                K::att_pv(cl, S, psum, smem + LDS_K, smem + LDS_SO, a.a_sa + row0 * D + HD * h, D, &am_a);
            }
            AK64 wo;
            load_k64(c, wo, a.w_o + ((long)(2 * c.w) * 8 + 2 * h) * (2 * 512), a.w_o + ((long)(2 * c.w + 1) * 8 + 2 * h) * (2 * 512));
            __syncthreads();   // O of head h complete; Q, K, V free
            K::gemm_k64(K::ctx_local(c), Y, wo, smem + LDS_SO);
        }
        emit_amax(a.ax_asa, am_a, c.lane);
        {   // h1 = h + dropout(a Wo^T + bo)
            const Bias2 bo = bias_load(c, a.b_o);
            residual_epilogue(c, H, Y, bo, cy, a.d_sao, row0, a.h, a.h1, nullptr);
        }
        // =========================== LayerNorm 2, cross-attention query ===========================
        am = 0.f;
        K::template layer_norm_to_x<false>(c, H, a.n2_w, a.n2_b, a.n2 + row0 * D, &am);
        emit_amax(a.ax_n2, am, c.lane);
        zero(Y);
        {
            const Bias2 bq = bias_load(c, a.b_q);
            K::template gemm_x2<S_W1>(K::ctx_local(c), Y, a.w_q);
            __syncthreads();   // every wave has read LN2(h1): the panel receives q; the Q / K / V region the memory's keys and values
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                f32x4 qv[2];
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2) {
                    qv[a2] = Y[a2][tt] * cy + bq.v[a2];
                    if (K::tok_ok(c, tt)) SD_NT_STORE((f32x4)(qv[a2]), reinterpret_cast<f32x4 *>(a.q + (row0 + K::tok_of(c, tt)) * D + 32 * c.w + 16 * a2 + 4 * c.g));
                }
                K::store_x(c, tt, qv[0] * ACT, qv[1] * ACT);
            }
        }
        // =========================== cross-attention over the M memory rows ===========================
        char *Kp = smem + LDS_Q;                 // keys: 16 rows in the panel's own row layout (x_off), kperm feature order like q
        char *VT = smem + LDS_Q + 16 * XROW;     // values transposed: [feature 256][plane][32 key positions], position 8 g + e = key 4 g + e (e < 4)
        {
            const float *kvp = a.kv + traj * (long)a.M * (2 * D);
            {   // keys
                const int m = threadIdx.x >> 5, ks = (threadIdx.x >> 2) & 7, g4 = threadIdx.x & 3;
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
                if (m < a.M) {
                    v0 = *reinterpret_cast<const f32x4 *>(kvp + (long)m * (2 * D) + 32 * ks + 4 * g4) * ACT;
                    v1 = *reinterpret_cast<const f32x4 *>(kvp + (long)m * (2 * D) + 32 * ks + 16 + 4 * g4) * ACT;
                }
                K::split_store8(Kp + K::x_off(m, g4 | (ks << 3)), Kp + K::x_off(m, g4 | 4 | (ks << 3)), v0, v1);
            }
            const f16 z0 = (f16)0.f;
            const f16x8 z8 = {z0, z0, z0, z0, z0, z0, z0, z0};
            for (int i = threadIdx.x; i < D * VT_PITCH / 16; i += NTHREADS) *reinterpret_cast<f16x8 *>(VT + 16 * i) = z8;
            __syncthreads();
            for (int i = threadIdx.x; i < a.M * (D / 4); i += NTHREADS) {
                const int m = i / (D / 4), j = i - m * (D / 4);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(kvp + (long)m * (2 * D) + D + 4 * j) * ACT;
                f16x4 hi, lo;
                split4(v, hi, lo);
                const int pos = (m >> 2) * 8 + (m & 3);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f16 *row = reinterpret_cast<f16 *>(VT + (4 * j + e) * VT_PITCH);
                    row[pos] = hi[e];
                    row[32 + pos] = lo[e];
                }
            }
            __syncthreads();   // q planes, keys, values staged
        }
        float am_c = 0.f;
        if (c.w < NTT) {
            const Ctx cl = K::ctx_local(c);
            const char *X = smem + LDS_X;
            f32x4 S[NH];
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                S[hh] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int ks = 2 * hh + kk;
                    const f16x8 kh = K::lds16(Kp + K::x_off(cl.t, cl.g | (ks << 3))), kl = K::lds16(Kp + K::x_off(cl.t, cl.g | 4 | (ks << 3)));
                    mma3<S_XSC>(S[hh], kh, kl, K::lds16(X + K::x_at(cl, cl.w, 0, ks)), K::lds16(X + K::x_at(cl, cl.w, 1, ks)));
                }
            }
            // every q fragment this wave needs is in registers: its tile's rows of the panel may now receive the attention output
            const float c_l = a.scale_log2e / (ACT * ACT);
            const int qtok = 16 * cl.w + cl.t;
            const bool q_ok = qtok < a.T;
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                f32x4 v = S[hh] * c_l;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * cl.g + r >= a.M) v[r] = -INFINITY;
                const float m = rows4_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                const float mb = m - 10.0f;
                f32x4 p = {__builtin_amdgcn_exp2f(v[0] - mb), __builtin_amdgcn_exp2f(v[1] - mb), __builtin_amdgcn_exp2f(v[2] - mb), __builtin_amdgcn_exp2f(v[3] - mb)};
                const float psum = rows4_sum((p[0] + p[1]) + (p[2] + p[3]));
                const long mrow = (traj * NH + hh) * a.T + (q_ok ? qtok : 0);
                if (q_ok && cl.g == 0) a.lse_ca[mrow] = mb + __builtin_amdgcn_logf(psum);
                if (a.d_cap.thresh && 4 * cl.g < a.M) p = p * dropout_quad(a.d_cap, (unsigned long)mrow * (unsigned long)((a.M + 3) >> 2) + (unsigned long)cl.g);
                f16x4 ph4, pl4;
                split4(p, ph4, pl4);
                const f16 z = (f16)0.f;
                const f16x8 ph = {ph4[0], ph4[1], ph4[2], ph4[3], z, z, z, z}, pl = {pl4[0], pl4[1], pl4[2], pl4[3], z, z, z, z};
                f32x4 O[4];
#pragma unroll
                for (int ft = 0; ft < 4; ++ft) {
                    const char *vrow = VT + (HD * hh + 16 * ft + cl.t) * VT_PITCH + 16 * cl.g;
                    O[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma3<S_XPV>(O[ft], K::lds16(vrow), K::lds16(vrow + 64), ph, pl);
                }
                if (q_ok) {
                    const float inv = 1.0f / psum, inv1 = inv * (1.0f / ACT);
                    char *Xw = smem + LDS_X;
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int chunk = cl.g | ((2 * hh + jj) << 3);
                        K::split_store8(Xw + K::x_off(qtok, chunk), Xw + K::x_off(qtok, chunk | 4), O[2 * jj] * inv, O[2 * jj + 1] * inv);
                    }
#pragma unroll
                    for (int ft = 0; ft < 4; ++ft) {
                        const f32x4 o = O[ft] * inv1;
                        SD_NT_STORE((f32x4)(o), reinterpret_cast<f32x4 *>(a.a_ca + (row0 + qtok) * D + HD * hh + 16 * ft + 4 * cl.g));
                        am_c = fmaxf(am_c, max4(o));
                    }
                }
            }
        }
        emit_amax(a.ax_aca, am_c, c.lane);
        __syncthreads();   // the panel holds the cross-attention output of every token
        zero(Y);
        {   // h2 = h1 + dropout(a_ca Woc^T + boc)
            const Bias2 boc = bias_load(c, a.b_oc);
            K::template gemm_x2<S_OUT>(K::ctx_local(c), Y, a.w_oc);
            residual_epilogue(c, H, Y, boc, cy, a.d_cao, row0, nullptr, a.h2, nullptr);
        }
        // =========================== feed-forward ===========================
        am = 0.f;
        K::template layer_norm_to_x<false>(c, H, a.n3_w, a.n3_b, a.nf + row0 * D, &am);
        emit_amax(a.ax_nf, am, c.lane);
        zero(Y);
        {
            const Bias2 b1 = bias_load(c, a.b_1);
            K::template gemm_x2<S_W1>(K::ctx_local(c), Y, a.w_1);
            __syncthreads();   // every wave has read LN3(h2): the panel receives dropout(gelu(pre))
            float am_u = 0.f;
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                f32x4 uv[2];
                const bool ok = K::tok_ok(c, tt);
                const long row = row0 + K::tok_of(c, tt);
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2) {
                    const int col = 32 * c.w + 16 * a2 + 4 * c.g;
                    const f32x4 pre = Y[a2][tt] * cy + b1.v[a2];
                    const f32x2 g0 = gelu_erf_as2(f32x2{pre[0], pre[1]}), g1 = gelu_erf_as2(f32x2{pre[2], pre[3]});
                    f32x4 u = {g0[0], g0[1], g1[0], g1[1]};
                    if (a.d_act.thresh) u = u * dropout_quad(a.d_act, (unsigned long)row * (D / 4) + (unsigned long)(col >> 2));
                    if (ok) {
                        SD_NT_STORE((f32x4)(pre), reinterpret_cast<f32x4 *>(a.pre + row * D + col));
                        SD_NT_STORE((f32x4)(u), reinterpret_cast<f32x4 *>(a.u + row * D + col));
                        am_u = fmaxf(am_u, max4(u));
                    }
                    uv[a2] = u * ACT;
                }
                K::store_x(c, tt, uv[0], uv[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            emit_amax(a.ax_u, am_u, c.lane);
            __syncthreads();
            zero(Y);
            const Bias2 b2 = bias_load(c, a.b_2);
            K::template gemm_x2<S_W2>(K::ctx_local(c), Y, a.w_2);
            float am_o = 0.f;
            residual_epilogue(c, H, Y, b2, cy, a.d_ffn, row0, nullptr, a.h3, a.ax_out ? &am_o : nullptr);
            emit_amax(a.ax_out, am_o, c.lane);
        }
        }   // !HEAD
        // =========================== the next layer's LayerNorm 1 and Q | K | V projection ===========================
        if (a.w_n) {
            am = 0.f;
            K::template layer_norm_to_x<false>(c, H, a.nn_w, a.nn_b, a.nn1 + row0 * D, &am);
            emit_amax(a.ax_nn, am, c.lane);
#pragma unroll 1
            for (int p = 0; p < 3; ++p) {
                zero(Y);
                const Bias2 bn = bias_load(c, a.b_n + p * D);
                K::template gemm_x2<S_QKV>(K::ctx_local(c), Y, a.w_n + (long)p * 16 * (8 * 2 * 512));
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    if (!K::tok_ok(c, tt)) continue;
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2)
                        SD_NT_STORE((f32x4)(Y[a2][tt] * cy + bn.v[a2]), reinterpret_cast<f32x4 *>(a.qkv2 + (row0 + K::tok_of(c, tt)) * (3 * D) + p * D + 32 * c.w + 16 * a2 + 4 * c.g));
                }
            }
        }
    }
};

template <int NTT>
__global__ __launch_bounds__(NTHREADS, 2) void train_layer_fwd_kernel(Args a) { L<NTT>::template body<false>(a); }
template <int NTT>
__global__ __launch_bounds__(NTHREADS, 2) void train_head_fwd_kernel(Args a) { L<NTT>::template body<true>(a); }

// Every weight matrix of a model in ONE launch (after each optimizer step): matrix blockIdx.y = rows[y] x 256 floats at base + src[y]
// -> planes at dst + dst_off[y] (halfs), the layout of pack_w16_kernel with the fixed scale WSC.
__global__ void pack_w16_multi_kernel(const float *__restrict__ base, const long *__restrict__ src, const int *__restrict__ rows,
                                      const long *__restrict__ dst_off, f16 *__restrict__ dst) {
    const int y = blockIdx.y, N = rows[y];
    const float *W = base + src[y];
    f16 *out = dst + dst_off[y];
    const long total = (long)N * (D / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / (D / 8)), k8 = (int)(i % (D / 8));
        f16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = W[(long)n * D + kperm(k8, e)] * WSC;
            hh[e] = (f16)v;
            ll[e] = (f16)(v - (float)hh[e]);
        }
        const int nt = n >> 4, ks = k8 >> 2, lane = (k8 & 3) * 16 + (n & 15);
        f16 *o = out + (((long)nt * 8 + ks) * 2) * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = hh[e];
            o[512 + e] = ll[e];
        }
    }
}

typedef void (*Fn)(Args);
static Fn head_kernel_for(int ntt) {
    switch (ntt) {
        case 1: return train_head_fwd_kernel<1>;
        case 2: return train_head_fwd_kernel<2>;
        case 3: return train_head_fwd_kernel<3>;
        case 4: return train_head_fwd_kernel<4>;
        case 5: return train_head_fwd_kernel<5>;
        case 6: return train_head_fwd_kernel<6>;
        case 7: return train_head_fwd_kernel<7>;
        default: return nullptr;
    }
}
static Fn kernel_for(int ntt) {
    switch (ntt) {
        case 1: return train_layer_fwd_kernel<1>;
        case 2: return train_layer_fwd_kernel<2>;
        case 3: return train_layer_fwd_kernel<3>;
        case 4: return train_layer_fwd_kernel<4>;
        case 5: return train_layer_fwd_kernel<5>;
        case 6: return train_layer_fwd_kernel<6>;
        case 7: return train_layer_fwd_kernel<7>;
        default: return nullptr;
    }
}

}   // namespace tjt

extern "C" size_t sd_pack_weight_traj_halfs(int N, int K) { return (size_t)((N + 15) / 16 * 16) * ((K + 31) / 32 * 32) * 2; }

extern "C" int sd_pack_weight_traj(const float *w, int N, int K, void *planes, void *stream) {
    if (!w || !planes || N <= 0 || !(K == tj::D || (K >= 1 && K <= 32))) return fail(SD_E_BADARG, "sd_pack_weight_traj: weights must be (N, 256) or (N, <= 32)");
    const int Np = (N + 15) / 16 * 16, Kp = K == tj::D ? K : 32;
    SD_LAUNCH(tj::pack_w16_kernel, dim3(grid_for((long)Np * (Kp / 8))), dim3(256), 0, (hipStream_t)stream, w, N, K, Np, Kp, (const unsigned *)nullptr, tjt::WSC,
              (f16 *)planes, (float *)nullptr);
    SD_CHECK_LAUNCH("pack_w16_kernel");
    return 0;
}

extern "C" int sd_pack_weight_traj_multi(const float *base, const int64_t *src_offsets, const int32_t *rows, const int64_t *dst_offsets, int n,
                                         int max_rows, void *planes, void *stream) {
    if (!base || !src_offsets || !rows || !dst_offsets || !planes || n <= 0 || max_rows <= 0) return fail(SD_E_BADARG, "sd_pack_weight_traj_multi: bad argument");
    unsigned gx = (unsigned)(((long)max_rows * (tj::D / 8) + 255) / 256);
    SD_LAUNCH(tjt::pack_w16_multi_kernel, dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, base, (const long *)src_offsets, (const int *)rows,
              (const long *)dst_offsets, (f16 *)planes);
    SD_CHECK_LAUNCH("pack_w16_multi_kernel");
    return 0;
}

extern "C" int sd_train_layer_fwd_ok(int d, int heads, int T, int M) { return d == tj::D && heads == tj::NH && T >= 1 && T <= tj::TMAX && M >= 1 && M <= 16; }

extern "C" int sd_train_layer_fwd(const sd_train_layer_fwd_args *p, void *stream) {
    if (!p || p->B <= 0) return fail(SD_E_BADARG, "sd_train_layer_fwd: bad argument");
    if (!sd_train_layer_fwd_ok(p->d, p->heads, p->T, p->M)) return fail(SD_E_BADDIM, "sd_train_layer_fwd: hidden_dim 256, 4 heads, T <= 100, M <= 16");
    const void *need[] = {p->h, p->qkv, p->a_sa, p->lse_sa, p->h1, p->n2, p->q, p->kv, p->a_ca, p->lse_ca, p->h2, p->nf, p->pre, p->u, p->h3, p->w_o, p->w_q,
                          p->w_oc, p->w_1, p->w_2, p->b_o, p->b_q, p->b_oc, p->b_1, p->b_2, p->n2_w, p->n2_b, p->n3_w, p->n3_b};
    for (const void *q : need)
        if (!q) return fail(SD_E_BADARG, "sd_train_layer_fwd: null pointer");
    if (p->w_n && (!p->b_n || !p->nn_w || !p->nn_b || !p->nn1 || !p->qkv2)) return fail(SD_E_BADARG, "sd_train_layer_fwd: the next-projection stage needs b_n, nn_w, nn_b, nn1, qkv2");
    if (!(p->p >= 0.f) || !(p->p < 1.f)) return fail(SD_E_BADARG, "sd_train_layer_fwd: p must be in [0, 1)");
    tjt::Args a;
    a.h = p->h; a.qkv = p->qkv; a.a_sa = p->a_sa; a.lse_sa = p->lse_sa; a.h1 = p->h1; a.n2 = p->n2; a.q = p->q; a.kv = p->kv;
    a.a_ca = p->a_ca; a.lse_ca = p->lse_ca; a.h2 = p->h2; a.nf = p->nf; a.pre = p->pre; a.u = p->u; a.h3 = p->h3; a.nn1 = p->nn1; a.qkv2 = p->qkv2;
    a.w_o = (const f16 *)p->w_o; a.w_q = (const f16 *)p->w_q; a.w_oc = (const f16 *)p->w_oc; a.w_1 = (const f16 *)p->w_1; a.w_2 = (const f16 *)p->w_2;
    a.w_n = (const f16 *)p->w_n;
    a.b_o = p->b_o; a.b_q = p->b_q; a.b_oc = p->b_oc; a.b_1 = p->b_1; a.b_2 = p->b_2; a.b_n = p->b_n;
    a.n2_w = p->n2_w; a.n2_b = p->n2_b; a.n3_w = p->n3_w; a.n3_b = p->n3_b; a.nn_w = p->nn_w; a.nn_b = p->nn_b;
    a.d_sap = make_dropout(p->p, p->seed, p->site_sa_probs);
    a.d_sao = make_dropout(p->p, p->seed, p->site_sa_out);
    a.d_cap = make_dropout(p->p, p->seed, p->site_ca_probs);
    a.d_cao = make_dropout(p->p, p->seed, p->site_ca_out);
    a.d_act = make_dropout(p->p, p->seed, p->site_act);
    a.d_ffn = make_dropout(p->p, p->seed, p->site_ffn);
    a.ax_asa = p->amax_a_sa; a.ax_n2 = p->amax_n2; a.ax_aca = p->amax_a_ca; a.ax_nf = p->amax_nf; a.ax_u = p->amax_u; a.ax_nn = p->amax_nn; a.ax_out = p->amax_out;
    a.T = p->T; a.M = p->M; a.B = p->B;
    a.scale_log2e = (1.0f / sqrtf((float)tj::HD)) * 1.44269504088896340736f;
    const int ntt = (p->T + 15) / 16;
    const tjt::Fn fn = tjt::kernel_for(ntt);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(SD_KCLASS_LAYER_CHAIN, s);
    static DevFlag attr_set[8];
    if (!attr_set[ntt]) {
        const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, tj::LDS_BYTES);
        if (e != hipSuccess) return fail((int)e, "train_layer_fwd_kernel: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        attr_set[ntt] = true;
    }
    SD_LAUNCH(fn, dim3((unsigned)p->B), dim3(tj::NTHREADS), (size_t)tj::LDS_BYTES, s, a);
    SD_CHECK_LAUNCH("train_layer_fwd_kernel");
    return 0;
}

extern "C" int sd_train_head_fwd(const float *x, const void *w_emb, const float *b_emb, const float *pe, float *h0, const float *ln_w, const float *ln_b,
                                 float *n1, const void *w_qkv, const float *b_qkv, float *qkv, uint32_t *amax_n1, int B, int T, int J, void *stream) {
    if (!x || !w_emb || !b_emb || !pe || !h0 || !ln_w || !ln_b || !n1 || !w_qkv || !b_qkv || !qkv || B <= 0) return fail(SD_E_BADARG, "sd_train_head_fwd: null pointer or empty shape");
    if (T < 1 || T > tj::TMAX || J < 4 || J > 32 || J % 4) return fail(SD_E_BADDIM, "sd_train_head_fwd: T <= 100, J a multiple of 4 up to 32 (hidden_dim 256)");
    tjt::Args a{};
    a.x_in = x; a.w_emb = (const f16 *)w_emb; a.b_emb = b_emb; a.pe = pe; a.J = J; a.h3 = h0;
    a.nn_w = ln_w; a.nn_b = ln_b; a.nn1 = n1; a.w_n = (const f16 *)w_qkv; a.b_n = b_qkv; a.qkv2 = qkv; a.ax_nn = amax_n1;
    a.T = T; a.B = B; a.M = 1;
    const int ntt = (T + 15) / 16;
    const tjt::Fn fn = tjt::head_kernel_for(ntt);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(SD_KCLASS_LAYER_CHAIN, s);
    static DevFlag attr_set[8];
    if (!attr_set[ntt]) {
        const hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, tj::LDS_BYTES);
        if (e != hipSuccess) return fail((int)e, "train_head_fwd_kernel: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        attr_set[ntt] = true;
    }
    SD_LAUNCH(fn, dim3((unsigned)B), dim3(tj::NTHREADS), (size_t)tj::LDS_BYTES, s, a);
    SD_CHECK_LAUNCH("train_head_fwd_kernel");
    return 0;
}
