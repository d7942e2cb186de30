// SoccerDiffusion image path, TRAINING: what a ResNet basic block needs beyond the inference convolutions of sd_conv.hip
// when the reference trains the backbone with every step (soccer_diffusion/ml/training/train.py:226-240 -> ml/model/encoder/image.py:38-52
// -> torchvision BasicBlock under autograd, BatchNorm2d in training mode):
//   * training-mode BatchNorm on NHWC tensors: batch statistics (per-channel shifted sums, double accumulators), normalise + affine
//     (+ residual) + ReLU with the abs-max word the next convolution scales its fp16 planes with, running statistics as
//     torch.nn.BatchNorm2d updates them (momentum, unbiased variance);
//   * its backward: the two per-channel reductions (sum g, sum g x_hat; g = the incoming gradient behind the ReLU mask), then
//     dy = gamma rstd (g - mean g - x_hat mean(g x_hat)), with the abs-max word of dy for the data-gradient convolution;
//   * the convolution's weight gradient dW[co][ci][tap] = sum over output pixels of dY[pixel][co] X[pixel * stride + tap - pad][ci]: a
//     "TN" product whose contraction runs over pixels, on the fp16 matrix pipe with block floating point per 32 pixels (as
//     gemm_tn16_kernel of sd_train.hip: gradients have no a-priori magnitude), fp32 atomics into torch's (Cout, Cin, k, k) layout.
// The data gradient needs no kernel of its own: it is the forward kernel (sd_conv3x3_bn_act / sd_conv1x1_bn_act with an identity
// epilogue) on the flipped, transposed weights - for the stride-2 stage entries on dY dilated with zeros (host side:
// soccerdiffusion_amd/conv_training.py).
#include "../../include/soccerdiffusion_hip.h"
#include "sd_common.h"
#include <stdlib.h>

namespace cvt {

// Workgroup ids go round-robin over the 8 XCDs (id % 8), each with its own L2.  A kernel whose neighbouring workgroups read overlapping rows (the
// 3 x 3 pooling windows) wants neighbours on ONE XCD: logical id = (id % 8) * (n / 8) + id / 8 gives every XCD a contiguous band (n a multiple of 8).
__device__ __forceinline__ unsigned xcd_band_id() {
    const unsigned b = blockIdx.x, n = gridDim.x;
    return (n & 7u) ? b : (b & 7u) * (n >> 3) + (b >> 3);
}

// ---- per-channel reductions over an NHWC tensor: thread -> 4 consecutive channels (one 16-byte load per pixel), a block walks its
// pixel range in steps of 256 * 4 / C pixels; lanes with the same channel group are combined through LDS, workgroups through partial sums in HBM
// A pixel's contribution is split into LOAD (global reads only) and ADD: four pixels' loads are issued before the first add - with one
// 16-byte load in flight per thread the reductions ran at 1.8 TB/s (statistics) / 2.9 TB/s (backward sums) against the 5+ TB/s of the
// element-wise kernels beside them.
template <class V, int UNROLL = 4, class L, class A>   // L(item, channel group base) -> V;  A(V, channel group base, s, q) accumulates into two f32x4
__device__ __forceinline__ void channel_reduce(long npix, int C, float *part /* [gridDim.x][C][2] */, L load, A add) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    // C in {64 .. 1024}: one pass, cg = C / 4 channel groups divide the 256 threads; wider tensors (ResNet-50's 2048): 1024 channels per pass
    for (int cb = 0; cb < C; cb += 1024) {
        const int cc = C - cb < 1024 ? C - cb : 1024, cg = cc / 4;
        const int c4 = tid % cg, prow = tid / cg, pstep = 256 / cg, c0 = cb + 4 * c4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
        long p = p0 + prow;
        if constexpr (UNROLL == 4) {
            for (; p + 3 * pstep < p1; p += 4 * pstep) {
                const V v0 = load(p, c0), v1 = load(p + pstep, c0), v2 = load(p + 2 * pstep, c0), v3 = load(p + 3 * pstep, c0);
                add(v0, c0, s, q);
                add(v1, c0, s, q);
                add(v2, c0, s, q);
                add(v3, c0, s, q);
            }
        }
        for (; p < p1; p += pstep) add(load(p, c0), c0, s, q);
        if (cb > 0) __syncthreads();   // the previous pass has read `red`
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[tid * 8 + e] = s[e];
            red[tid * 8 + 4 + e] = q[e];
        }
        __syncthreads();
        if (tid < cg) {
            for (int r = 1; r < pstep; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[e] += red[(r * cg + tid) * 8 + e];
                    q[e] += red[(r * cg + tid) * 8 + 4 + e];
                }
            // the workgroup's partial sums, plain stores: partial_sum_kernel adds the workgroups up in double.  (Double atomics on the 2 C
            // sums themselves ran at ~ 5 G atomics/s: 2048 workgroups x 128 sums = 50 us per launch, 512 x 1024 at layer 4 = 100 us - for
            // 20 us of data.)
            float *o = part + ((long)blockIdx.x * C + cb + 4 * tid) * 2;
            *reinterpret_cast<f32x4 *>(o) = f32x4{s[0], q[0], s[1], q[1]};
            *reinterpret_cast<f32x4 *>(o + 4) = f32x4{s[2], q[2], s[3], q[3]};
        }
    }
}
// acc[e] = sum over the partials of part[p][e], e < n (a multiple of 32): a workgroup takes 32 entries, its 8 thread rows every 8th partial
__global__ __launch_bounds__(256) void partial_sum_kernel(const float *__restrict__ part, int nparts, int n, double *acc) {
    __shared__ double red[8][32];
    const int e = blockIdx.x * 32 + (threadIdx.x & 31), row = threadIdx.x >> 5;
    double s = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int p = row;
    for (; p + 24 < nparts; p += 32) {
        const float a = part[(long)p * n + e], b = part[(long)(p + 8) * n + e], c = part[(long)(p + 16) * n + e], d = part[(long)(p + 24) * n + e];
        s += (double)a;
        s1 += (double)b;
        s2 += (double)c;
        s3 += (double)d;
    }
    for (; p < nparts; p += 8) s += (double)part[(long)p * n + e];
    s += s1 + s2 + s3;
    red[row][threadIdx.x & 31] = s;
    __syncthreads();
    if (row == 0) {
#pragma unroll
        for (int r = 1; r < 8; ++r) s += red[r][threadIdx.x & 31];
        acc[e] = s;
    }
}

// sums of (y - pivot) and (y - pivot)^2 per channel; pivot = the first pixel's value of the channel (keeps the variance formula
// q / n - (s / n)^2 free of the cancellation a large mean would cause)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ y, long npix, int C, float *part) {
    channel_reduce<f32x4>(
        npix, C, part, [&](long p, int c0) { return *reinterpret_cast<const f32x4 *>(y + p * C + c0); },
        [&](const f32x4 &yv, int c0, f32x4 &s, f32x4 &q) {
            const f32x4 v = yv - *reinterpret_cast<const f32x4 *>(y + c0);
            s = s + v;
            q = q + v * v;
        });
}
// mean, rstd of the batch; running statistics as torch.nn.BatchNorm2d (momentum m: r = (1 - m) r + m batch, unbiased variance)
__global__ void bn_finalize_kernel(const double *acc, const float *__restrict__ y, long npix, int C, float eps, float momentum, float *mean,
                                   float *rstd, float *running_mean, float *running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double n = (double)npix, s = acc[2 * c] / n, var = fmax(acc[2 * c + 1] / n - s * s, 0.0), m = (double)y[c] + s;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (npix > 1 ? n / (n - 1.0) : 1.0));
    }
}
__device__ __forceinline__ f32x4 bn_affine(const f32x4 &y, const float *mean, const float *rstd, const float *gamma, const float *beta, int c0) {
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(rstd + c0) * *reinterpret_cast<const f32x4 *>(gamma + c0);
    return (y - *reinterpret_cast<const f32x4 *>(mean + c0)) * sc + *reinterpret_cast<const f32x4 *>(beta + c0);
}
// z = relu?((y - mean) rstd gamma + beta (+ res)); abs-max of z -> word
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ y, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ res,
                                                       float *__restrict__ z, long n4, int C, int relu, unsigned *amax) {
    float mx = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)((4 * i) % C);
        f32x4 v = bn_affine(*reinterpret_cast<const f32x4 *>(y + 4 * i), mean, rstd, gamma, beta, c0);
        if (res) v = v + *reinterpret_cast<const f32x4 *>(res + 4 * i);
        if (relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        *reinterpret_cast<f32x4 *>(z + 4 * i) = v;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const unsigned b = __builtin_bit_cast(unsigned, mx);
        if ((threadIdx.x & 63) == 0 && b > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, b);
    }
}
// backward reductions: acc[c] = (sum g, sum g x_hat), g = dz behind the ReLU mask, x_hat = (y - mean) rstd.  The mask is z > 0, or - z NULL, a
// unit without residual operand - recomputed from y with the forward's own expression (one tensor less to read, here and in the apply kernel)
struct BnBwdLoad { f32x4 g, z, y; };
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ dz, const float *__restrict__ z, const float *__restrict__ y,
                                                            const float *__restrict__ mean, const float *__restrict__ rstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, long npix, int C, int relu,
                                                            float *part) {
    channel_reduce<BnBwdLoad>(
        npix, C, part,
        [&](long p, int c0) {
            BnBwdLoad v;
            v.g = *reinterpret_cast<const f32x4 *>(dz + p * C + c0);
            v.y = *reinterpret_cast<const f32x4 *>(y + p * C + c0);
            v.z = (relu && z) ? *reinterpret_cast<const f32x4 *>(z + p * C + c0) : f32x4{1.f, 1.f, 1.f, 1.f};
            return v;
        },
        [&](const BnBwdLoad &v, int c0, f32x4 &s, f32x4 &q) {
            f32x4 g = v.g;
            const f32x4 zz = (relu && !z) ? bn_affine(v.y, mean, rstd, gamma, beta, c0) : v.z;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
            const f32x4 xh = (v.y - *reinterpret_cast<const f32x4 *>(mean + c0)) * *reinterpret_cast<const f32x4 *>(rstd + c0);
            s = s + g;
            q = q + g * xh;
        });
}
// dy = gamma rstd (g - s1 / n - x_hat s2 / n); dgamma = s2, dbeta = s1 (block 0 writes them); g -> dres (the residual branch's gradient) if asked
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ dz, const float *__restrict__ z, const float *__restrict__ y,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const double *__restrict__ acc, float *__restrict__ dy, float *__restrict__ dres,
                                                           float *dgamma, float *dbeta, long npix, int C, int relu, unsigned *amax) {
    const double inv_n = 1.0 / (double)npix;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            dgamma[c] = (float)acc[2 * c + 1];
            dbeta[c] = (float)acc[2 * c];
        }
    float mx = 0.f;
    const long n4 = npix * C / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)((4 * i) % C);
        f32x4 g = *reinterpret_cast<const f32x4 *>(dz + 4 * i);
        const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + 4 * i);
        if (relu) {
            const f32x4 zz = z ? *reinterpret_cast<const f32x4 *>(z + 4 * i) : bn_affine(yv, mean, rstd, gamma, beta, c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        }
        if (dres) *reinterpret_cast<f32x4 *>(dres + 4 * i) = g;
        const f32x4 rs = *reinterpret_cast<const f32x4 *>(rstd + c0);
        const f32x4 xh = (yv - *reinterpret_cast<const f32x4 *>(mean + c0)) * rs;
        f32x4 m1, m2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m1[e] = (float)(acc[2 * (c0 + e)] * inv_n);
            m2[e] = (float)(acc[2 * (c0 + e) + 1] * inv_n);
        }
        const f32x4 v = (g - m1 - xh * m2) * (rs * *reinterpret_cast<const f32x4 *>(gamma + c0));
        *reinterpret_cast<f32x4 *>(dy + 4 * i) = v;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const unsigned b = __builtin_bit_cast(unsigned, mx);
        if ((threadIdx.x & 63) == 0 && b > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, b);
    }
}

// ---- the stem's BatchNorm + ReLU + max-pool (3 x 3, stride 2, padding 1) as ONE forward launch and two backward launches, without the
// 3-GB tensor z = relu(BN(y)) in between (160 frames of 480 x 640: y and z are 3.15 GB each; torch's route writes z, reads it for the pool,
// writes the pool's gradient dz and reads it twice in the BatchNorm backward).  Forward: a thread owns 4 channels of a pooled pixel, takes
// the maximum of relu(BN(y)) over its window and remembers WHERE it was (one byte per element: 0 .. 8 in scan order, first maximum wins like
// ATen's kernel; 9 = the window holds no positive value, its gradient dies in the ReLU).  Backward: the gradient of convolution pixel (r, c)
// is g = sum of dp over the <= 4 windows whose winner it is - gathered from dp and the index bytes, no dz tensor - and feeds the same two
// BatchNorm launches as bn_bwd_reduce / bn_bwd_apply.
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float *__restrict__ y, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ p,
                                                               unsigned *__restrict__ idx, int N, int Hc, int Wc, int Hp, int Wp, int C, unsigned *amax) {
    // (32-bit index arithmetic: the host refuses tensors of 2^31 float4 or more; three 64-bit divisions per element cost more than its nine loads)
    const unsigned cg = C / 4;
    const unsigned n4 = (unsigned)N * Hp * Wp * cg;
    float mx = 0.f;
    // (bands of consecutive pooled rows per XCD: pooled rows i and i + 1 share convolution row 2 i + 1 - with round-robin ids the two reads went to
    // different L2s and the counters showed 1.57 x y's bytes fetched from HBM)
    for (unsigned e = xcd_band_id() * blockDim.x + threadIdx.x; e < n4; e += gridDim.x * blockDim.x) {
        const int c0 = (int)(e % cg) * 4;
        unsigned t = e / cg;
        const int j = (int)(t % (unsigned)Wp); t /= (unsigned)Wp;
        const int i = (int)(t % (unsigned)Hp);
        const int n = (int)(t / (unsigned)Hp);
        f32x4 m = {0.f, 0.f, 0.f, 0.f};
        unsigned k[4] = {9u, 9u, 9u, 9u};
        // all nine loads first, from clamped coordinates (a window position outside the map re-reads a border pixel and is masked below): with a
        // branch per position the loads went out one at a time and the launch ran at the rate of ONE 16-byte load in flight per thread
        f32x4 yv[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int r = 2 * i - 1 + ky, rc = r < 0 ? 0 : (r >= Hc ? Hc - 1 : r);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int c = 2 * j - 1 + kx, cc = c < 0 ? 0 : (c >= Wc ? Wc - 1 : c);
                yv[ky * 3 + kx] = *reinterpret_cast<const f32x4 *>(y + (((long)n * Hc + rc) * Wc + cc) * C + c0);
            }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int r = 2 * i - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int c = 2 * j - 1 + kx;
                const bool in = r >= 0 && r < Hc && c >= 0 && c < Wc;
                const f32x4 v = bn_affine(yv[ky * 3 + kx], mean, rstd, gamma, beta, c0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (in && v[q] > m[q]) {
                        m[q] = v[q];
                        k[q] = (unsigned)(ky * 3 + kx);
                    }
            }
        }
        *reinterpret_cast<f32x4 *>(p + 4l * e) = m;
        idx[e] = k[0] | (k[1] << 8) | (k[2] << 16) | (k[3] << 24);
        mx = fmaxf(mx, fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])));
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const unsigned b = __builtin_bit_cast(unsigned, mx);
        if ((threadIdx.x & 63) == 0 && b > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, b);
    }
}
// The backward works on 2 x 2 BLOCKS of convolution pixels (2 i' + a, 2 j' + b): together they lie in exactly the four windows (i', j'), (i', j' + 1),
// (i' + 1, j'), (i' + 1, j' + 1) - pixel (even, even) in the first only, at its centre (position 4); (even, odd) in the first two (5, 3); (odd, even)
// in the first and third (7, 1); (odd, odd) in all four (8, 6, 2, 0) - so one thread loads four index words and four gradients for four pixels
// (a thread per PIXEL loaded 2.25 windows on average, behind branches that depend on the pixel's parity).
struct PoolGeom { int Hc, Wc, Hp, Wp, C, Hb, Wb; };   // Hb x Wb blocks per image
struct PoolBlock { f32x4 g[4], y[4]; bool in[4]; long at[4]; };
__device__ __forceinline__ PoolBlock pool_block(const float *__restrict__ dp, const unsigned *__restrict__ idx, const float *__restrict__ y, const PoolGeom &G,
                                                unsigned item, int c0) {
    const int jb = (int)(item % (unsigned)G.Wb);
    const unsigned t = item / (unsigned)G.Wb;
    const int ib = (int)(t % (unsigned)G.Hb), n = (int)(t / (unsigned)G.Hb);
    PoolBlock B;
    unsigned k4[4];
    f32x4 d[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = 2 * ib + a, c = 2 * jb + b, rc = r < G.Hc ? r : G.Hc - 1, cc = c < G.Wc ? c : G.Wc - 1;
            B.in[2 * a + b] = r < G.Hc && c < G.Wc;
            B.at[2 * a + b] = (((long)n * G.Hc + rc) * G.Wc + cc) * G.C + c0;
            B.y[2 * a + b] = *reinterpret_cast<const f32x4 *>(y + B.at[2 * a + b]);
            const int i = ib + a, j = jb + b, ic = i < G.Hp ? i : G.Hp - 1, jc = j < G.Wp ? j : G.Wp - 1;
            const long w = (((long)n * G.Hp + ic) * G.Wp + jc) * G.C + c0;
            const bool wok = i < G.Hp && j < G.Wp;
            k4[2 * a + b] = wok ? idx[w >> 2] : 0x09090909u;   // (9 matches no position)
            d[2 * a + b] = *reinterpret_cast<const f32x4 *>(dp + w);
        }
    auto pick = [&](int win, unsigned pos, int q) __attribute__((always_inline)) { return ((k4[win] >> (8 * q)) & 255u) == pos ? d[win][q] : 0.f; };
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        B.g[0][q] = pick(0, 4u, q);
        B.g[1][q] = pick(0, 5u, q) + pick(1, 3u, q);
        B.g[2][q] = pick(0, 7u, q) + pick(2, 1u, q);
        B.g[3][q] = (pick(0, 8u, q) + pick(1, 6u, q)) + (pick(2, 2u, q) + pick(3, 0u, q));
    }
    return B;
}
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const float *__restrict__ dp, const unsigned *__restrict__ idx, const float *__restrict__ y,
                                                                 const float *__restrict__ mean, const float *__restrict__ rstd, long nitems, PoolGeom G,
                                                                 float *part) {
    channel_reduce<PoolBlock, 1>(
        nitems, G.C, part, [&](long p, int c0) { return pool_block(dp, idx, y, G, (unsigned)p, c0); },
        [&](const PoolBlock &v, int c0, f32x4 &s, f32x4 &q) {
            const f32x4 mu = *reinterpret_cast<const f32x4 *>(mean + c0), rs = *reinterpret_cast<const f32x4 *>(rstd + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!v.in[k]) continue;
                s = s + v.g[k];
                q = q + v.g[k] * ((v.y[k] - mu) * rs);
            }
        });
}
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float *__restrict__ dp, const unsigned *__restrict__ idx, const float *__restrict__ y,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                const float *__restrict__ gamma, const double *__restrict__ acc, float *__restrict__ dy,
                                                                float *dgamma, float *dbeta, long npix, long nitems, PoolGeom G, unsigned *amax) {
    const double inv_n = 1.0 / (double)npix;
    const int C = G.C, cg = C / 4;
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            dgamma[c] = (float)acc[2 * c + 1];
            dbeta[c] = (float)acc[2 * c];
        }
    float mx = 0.f;
    const unsigned n4 = (unsigned)(nitems * cg);
    for (unsigned i = xcd_band_id() * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        const int c0 = (int)(i % (unsigned)cg) * 4;
        const PoolBlock B = pool_block(dp, idx, y, G, i / (unsigned)cg, c0);
        const f32x4 rs = *reinterpret_cast<const f32x4 *>(rstd + c0), mu = *reinterpret_cast<const f32x4 *>(mean + c0);
        const f32x4 sc = rs * *reinterpret_cast<const f32x4 *>(gamma + c0);
        f32x4 m1, m2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m1[e] = (float)(acc[2 * (c0 + e)] * inv_n);
            m2[e] = (float)(acc[2 * (c0 + e) + 1] * inv_n);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!B.in[k]) continue;
            const f32x4 v = (B.g[k] - m1 - ((B.y[k] - mu) * rs) * m2) * sc;
            *reinterpret_cast<f32x4 *>(dy + B.at[k]) = v;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    if (amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        const unsigned b = __builtin_bit_cast(unsigned, mx);
        if ((threadIdx.x & 63) == 0 && b > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, b);
    }
}

// ---- convolution weight gradient.  One WAVE owns a 64 (co) x 64 (ci) tile of dW for ONE tap and a group of `rows_per_item` output image
// rows (n, oy); it walks their pixels 32 at a time: A[i = co][k = pixel] = dY[pixel][co0 + lane & 31 (+ 32)], B[k = pixel][j = ci] =
// X[pixel * stride + tap - pad][ci0 + lane & 31 (+ 32)] (zero outside the image), both read as they lie in memory (NHWC: a pixel's channels
// are one 128-byte segment per half-wave), split into fp16 hi / lo with one power-of-two scale per operand and 32 pixels (block floating
// point), 24 MFMAs into a sub-accumulator that is added un-scaled to the wave's fp32 accumulators; at the end fp32 atomics into
// dW (Cout, Cin, k, k).
struct WgradArgs {
    const float *dy;   // [N][Ho][Wo][Cout]
    const float *x;    // [N][H][W][Cin]
    const unsigned *dy_amax, *x_amax;   // abs-max words of the two tensors, or NULL: block floating point per 32 pixels
    float *dw;         // [Cout][Cin][ks][ks], zeroed by the caller
    int N, H, W, Ho, Wo, Cin, Cout, ks, stride, pad;
    int rows_per_item;   // output image rows per work item
    int n_row_items;     // ceil(N * Ho / rows_per_item)
};
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <bool GLOBAL>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int taps = a.ks * a.ks, ct = a.Cout / 64, it = a.Cin / 64;
    const long total = (long)ct * it * taps * a.n_row_items;
    if (item >= total) return;
    // consecutive waves share the row group (the operands' pixels: L2 locality), then tap, then the channel tiles
    const int tap = (int)(item % taps); item /= taps;
    const int cit = (int)(item % it); item /= it;
    const int cot = (int)(item % ct); item /= ct;
    const int ky = tap / a.ks, kx = tap - ky * a.ks;
    const long row0 = item * a.rows_per_item, row1 = row0 + a.rows_per_item < (long)a.N * a.Ho ? row0 + a.rows_per_item : (long)a.N * a.Ho;
    const float *dyc = a.dy + cot * 64 + l31, *xc = a.x + cit * 64 + l31;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Both tensors' abs-max words known (the producing launches left them behind): ONE power-of-two scale per operand for the whole
    // launch, as the forward convolution does - no per-step reductions, no sub-accumulator (the step is bound by its vector instructions)
    constexpr bool global_scale = GLOBAL;
    const float gsy = global_scale ? f16_scale_from_bits(*a.dy_amax) : 1.f, gsx = global_scale ? f16_scale_from_bits(*a.x_amax) : 1.f;
    for (long row = row0; row < row1; ++row) {
        const int n = (int)(row / a.Ho), oy = (int)(row - (long)n * a.Ho);
        const int iy = oy * a.stride + ky - a.pad;
        if (iy < 0 || iy >= a.H) continue;   // wave-uniform: the whole image row reads padding
        const float *dyr = dyc + row * (long)a.Wo * a.Cout;
        const float *xr = xc + ((long)n * a.H + iy) * a.W * a.Cin;
        for (int x0 = 0; x0 < a.Wo; x0 += 32) {
            float av[2][2][8], bv[2][2][8];
            float my = 0.f, mx = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ox = x0 + 16 * st + 8 * half + e, ix = ox * a.stride + kx - a.pad;
                    const bool oky = ox < a.Wo, okx = oky && ix >= 0 && ix < a.W;
                    const float *pa = dyr + (long)(oky ? ox : 0) * a.Cout, *pb = xr + (long)(okx ? ix : 0) * a.Cin;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const float va = oky ? pa[32 * t] : 0.f, vb = okx ? pb[32 * t] : 0.f;
                        av[st][t][e] = va;
                        bv[st][t][e] = vb;
                        my = fmaxf(my, fabsf(va));
                        mx = fmaxf(mx, fabsf(vb));
                    }
                }
            if constexpr (global_scale) {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float va = av[st][t][e] * gsy, vb = bv[st][t][e] * gsx;
                            ah[t][e] = (f16)va;
                            al[t][e] = (f16)(va - (float)ah[t][e]);
                            bh[t][e] = (f16)vb;
                            bl[t][e] = (f16)(vb - (float)bh[t][e]);
                        }
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) {
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
                        }
                }
                continue;
            }
            const float sy = f16_scale_from_bits(__builtin_bit_cast(unsigned, wave_max(my)));
            const float sx = f16_scale_from_bits(__builtin_bit_cast(unsigned, wave_max(mx)));
            f32x16 sub[2][2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float va = av[st][t][e] * sy, vb = bv[st][t][e] * sx;
                        ah[t][e] = (f16)va;
                        al[t][e] = (f16)(va - (float)ah[t][e]);
                        bh[t][e] = (f16)vb;
                        bl[t][e] = (f16)(vb - (float)bh[t][e]);
                    }
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        if (st == 0) sub[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], zero16, 0, 0, 0);
                        else sub[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], sub[tm][tn], 0, 0, 0);
                        sub[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], sub[tm][tn], 0, 0, 0);
                        sub[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], sub[tm][tn], 0, 0, 0);
                    }
            }
            const float un = 1.0f / (sy * sx);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = acc[tm][tn] + sub[tm][tn] * un;
        }
    }
    // accumulator element r of lane (l31, half): row (co) = (r & 3) + 8 (r >> 2) + 4 half of the 32-row tile, column (ci) = l31
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int ci = cit * 64 + tn * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cot * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float v = acc[tm][tn][r] * (global_scale ? 1.0f / (gsy * gsx) : 1.0f);
                if (v != 0.f) atomicAdd(a.dw + ((long)co * a.Cin + ci) * taps + tap, v);
            }
        }
}

// ---- 3 x 3, stride 1 (13 of ResNet-18's 16 block convolutions): all nine taps from ONE staging of the operands.  The per-wave kernel above
// fetches 16 KB per 262 kFLOP - every tap re-reads dY and a shifted X - and sits on the L2's bandwidth (~ 90 TFLOP/s).  Here a workgroup
// (4 waves) owns a 64 (co) x 64 (ci) tile of dW for ALL nine taps and walks 32-pixel COLUMN STRIPS of the images downwards: a step = the 32
// output pixels (n, oy, x0 .. x0 + 31); its operands are dY's row (32 x 64) and the three halo rows oy - 1 .. oy + 1 of X (34 x 64), of which
// only row oy + 1 is new - the rows live in a ring of four LDS slots as fp16 hi | lo planes (16-byte global loads; one power-of-two scale per
// tensor from its abs-max word), dY in two alternating buffers, so ONE barrier per step orders everything (a wave stores step m + 1's
// operands only after the barrier that every wave reaches after its MFMAs of step m - 1, and those stores touch neither the slots nor the dY
// buffer step m reads).  Wave (co half, ci half) runs 9 taps x 6 MFMAs per step on fragments from transposing LDS reads (ds_read_b64_tr_b16:
// 8 consecutive pixels of one channel per lane) - the dY fragments are shared by the nine taps, a kernel row's three taps are windows of
// one 12-pixel fragment.  (The first version staged dY and the whole 3 x 34 halo per 32 pixels of an image ROW: 134 staged pixels per step
// instead of 66, X read 3.2 times - more than an XCD's L2 holds across its workgroups - and 1.05 ms per layer-1 launch; NOTEBOOK round 5.)
// A strip's first step is preceded by two staging-only steps (rows oy - 1, oy).  The grid is persistent: ~ 512 workgroups (two per CU),
// workgroup (tile, group g) walks steps [g, g + 1) x per_group of the linear order (n, strip, oy) and stores ONE partial tile
// [tap][co][ci] (128-byte rows); wgrad3_reduce_kernel adds the groups up into torch's (Cout, Cin, 3, 3) layout - no atomics, deterministic.
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct Wgrad3Args {
    const float *dy;   // [N][H][W][Cout]
    const float *x;    // [N][Hx][Wx][Cin]; the kernel sees the (H, W) image x'[r][c] = x[xs r + xr][xs c + xc] (zero outside x): xs = 1: x itself;
                       // xs = 2: one row / column parity class of a stride-2 convolution's input
    const unsigned *dy_amax, *x_amax;
    float *part;       // [groups][taps of this instantiation][Cout][Cin]
    int N, H, W, Cin, Cout, nstrips, groups;
    int Hx, Wx, xs, xr, xc;
    long steps, per_group;   // N * nstrips * H steps in all
    int abl;           // diagnostic (SD_W3_ABL): 1 skip the MFMA phase, 2 skip the LDS staging stores, 4 skip the global loads
};
constexpr int W3_PITCH = 72;                       // halfs per staged pixel (64 channels + 8: 144 bytes, an odd number of 16-byte units)
constexpr int W3_DY = 32 * W3_PITCH;               // halfs of one dY buffer
constexpr int W3_XROW = 36 * W3_PITCH;             // ... of one halo row of X (34 pixels + 2 that only the 12-pixel fragment reads touch)
constexpr int W3_PLANE = 2 * W3_DY + 4 * W3_XROW;  // one plane (hi or lo): two dY buffers, then the ring of four rows
__device__ __forceinline__ f16x8 w3_frag(const f16 *plane, int row0, int col0, int lane) {
    // as tns_frag of sd_train.hip: lane (l31, half) receives channel col0 + l31, pixels row0 + 0 .. 7 (row0 already holds 8 * half)
    const int q = (lane & 15) >> 2, p4 = (lane & 3) * 4, gc = ((lane >> 4) & 1) * 16;
    const f16 *at = plane + (row0 + q) * W3_PITCH + col0 + gc + p4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)at);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(at + 4 * W3_PITCH));
    return __builtin_shufflevector(__builtin_bit_cast(f16x4, lo), __builtin_bit_cast(f16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}
// 12 consecutive pixels row0 .. row0 + 11 of channel col0 + l31 (three transposing reads): the three taps of a kernel row use the windows
// [kx, kx + 8) of them - two LDS reads and four byte-permutes per shifted fragment instead of six LDS reads
struct Frag12 { f16x4 a, b, c; };
__device__ __forceinline__ Frag12 w3_frag12(const f16 *plane, int row0, int col0, int lane) {
    const int q = (lane & 15) >> 2, p4 = (lane & 3) * 4, gc = ((lane >> 4) & 1) * 16;
    const f16 *at = plane + (row0 + q) * W3_PITCH + col0 + gc + p4;
    Frag12 f;
    f.a = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)at));
    f.b = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(at + 4 * W3_PITCH)));
    f.c = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(at + 8 * W3_PITCH)));
    return f;
}
template <int KX>
__device__ __forceinline__ f16x8 w3_window(const Frag12 &f) {
    const f16x8 lo = __builtin_shufflevector(f.a, f.b, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(f.b, f.c, 0, 1, 2, 3, 4, 5, 6, 7);
    if constexpr (KX == 0) return lo;
    else if constexpr (KX == 1) return __builtin_shufflevector(lo, hi, 1, 2, 3, 4, 5, 6, 7, 12);   // hi = pixels 4 .. 11: element 12 of (lo, hi) = pixel 8
    else return __builtin_shufflevector(lo, hi, 2, 3, 4, 5, 6, 7, 12, 13);
}
// KYM / KXM: bit ky' / kx' set = the shift (ky' - 1, kx' - 1) between dY and x' is computed (all nine for the 3 x 3 / stride-1 convolution).  A 3 x 3 /
// stride-2 convolution is four launches, one per parity class (pr, pc) of x: tap ky reads row 2 oy + ky - 1 = row oy - 1 of the odd class (ky = 0), row oy
// of the even class (ky = 1) or row oy of the odd class (ky = 2) - the odd class computes shifts {-1, 0}, the even class shift {0}: 4 + 2 + 2 + 1 taps.
template <int KYM, int KXM>
__global__ __launch_bounds__(256, 2) void conv_wgrad3_kernel(Wgrad3Args a) {
    constexpr int NKX = ((KXM >> 0) & 1) + ((KXM >> 1) & 1) + ((KXM >> 2) & 1), NKY = ((KYM >> 0) & 1) + ((KYM >> 1) & 1) + ((KYM >> 2) & 1), NT = NKY * NKX;
    __shared__ __attribute__((aligned(16))) f16 sm[2 * W3_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int coh = wave & 1, cih = wave >> 1;     // this wave's 32-channel halves of the tile
    const int ct = a.Cout / 64, it = a.Cin / 64;
    long item = blockIdx.x;
    const int cit = (int)(item % it); item /= it;
    const int cot = (int)(item % ct); item /= ct;  // item = the group of steps
    const float sy = f16_scale_from_bits(*a.dy_amax), sx = f16_scale_from_bits(*a.x_amax);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // staging assignments: thread -> pixel (tid >> 4) + 16 v of a row, 4 channels c4: dY 32 pixels (v < 2), X 34 pixels (v < 3; v = 2: 2 pixels)
    const int c4 = (tid & 15) * 4, px = tid >> 4;
    const float *dyb = a.dy + cot * 64, *xb = a.x + cit * 64;   // wave-uniform bases: a load is base (SGPRs) + a 32-bit lane offset
    // the walk: (n, strip, oy) of the next RUN step, `pre` staging-only steps before it (2 at the head of a strip), `left` run steps to go
    const long u0 = item * a.per_group, u1 = u0 + a.per_group < a.steps ? u0 + a.per_group : a.steps;
    long left = u1 > u0 ? u1 - u0 : 0;
    int oy = (int)(u0 % a.H), strip = (int)((u0 / a.H) % a.nstrips), n = (int)(u0 / a.H / a.nstrips), pre = 2;
    struct Micro { int n, x0, xrow, oy, run; };
    auto next = [&]() __attribute__((always_inline)) {
        Micro m;
        m.n = n; m.x0 = strip * 32; m.oy = oy;
        if (pre > 0) {
            m.xrow = oy + 1 - pre;   // rows oy - 1, oy
            m.run = 0;
            --pre;
        } else {
            m.xrow = oy + 1;
            m.run = 1;
            --left;
            if (++oy == a.H) {
                oy = 0;
                pre = 2;
                if (++strip == a.nstrips) {
                    strip = 0;
                    ++n;
                }
            }
        }
        return m;
    };
    // Two register sets A / B hold the operands of the next two steps: while step m's MFMAs run, the loads of steps m + 1 AND m + 2 are in flight
    // (one step ahead = 20 KB per workgroup left the loads latency-bound: memory phase and MFMA phase added up, 0.35 + 0.53 ms at layer 1).
    // Every load is issued unconditionally - a lane outside the image reads the tensor's first bytes and its values are multiplied by a zero
    // scale at the split - so that hipcc's s_waitcnt for the older set counts exactly the younger set's five loads on every path.
    struct Stage { f32x4 y[2], x[3]; unsigned ok; Micro m; };
    auto load = [&](Stage &s) __attribute__((always_inline)) {
        const Micro &m = s.m;
        const bool live = !(a.abl & 4);
        const int rr = a.xs * m.xrow + a.xr;   // the row of x behind row m.xrow of x'
        const bool rowok = m.xrow >= 0 && rr < a.Hx && live;
        const float *xr = xb + (((long)m.n * a.Hx + rr) * a.Wx + (a.xs * (m.x0 - 1) + a.xc)) * (long)a.Cin;   // wave-uniform; only dereferenced inside the image
        const float *yr = dyb + (((long)m.n * a.H + m.oy) * a.W + m.x0) * (long)a.Cout;
        s.ok = 0u;
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int p = px + 16 * v, ix = m.x0 - 1 + p;
            const bool ok = rowok && p < 34 && ix >= 0 && a.xs * ix + a.xc < a.Wx;
            s.x[v] = *reinterpret_cast<const f32x4 *>(ok ? xr + (unsigned)(p * a.xs * a.Cin + c4) : a.x);
            s.ok |= (unsigned)ok << v;
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int p = px + 16 * v;
            const bool ok = m.run > 0 && m.x0 + p < a.W && live;
            s.y[v] = *reinterpret_cast<const f32x4 *>(ok ? yr + (unsigned)(p * a.Cout + c4) : a.dy);
            s.ok |= (unsigned)ok << (3 + v);
        }
    };
    int q = 0, nrun = 0;   // q: rows of X staged so far (row number q lives in slot q & 3); nrun: run steps so far (dY buffer nrun & 1)
    // one step: stage S's operands, refill S with the step after the other set's, barrier, MFMAs.  Returns false after the last step.
    auto step = [&](Stage &S) __attribute__((always_inline)) {
        const bool run = S.m.run > 0, filled = S.m.run >= 0;
        if (filled && !(a.abl & 2)) {
            f16 *xs = sm + 2 * W3_DY + (q & 3) * W3_XROW + c4;
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int p = px + 16 * v;
                if (p < 34) {
                    f16x4 h, l;
                    f16_split4(S.x[v], (S.ok >> v) & 1u ? sx : 0.f, h, l);
                    *reinterpret_cast<f16x4 *>(xs + p * W3_PITCH) = h;
                    *reinterpret_cast<f16x4 *>(xs + p * W3_PITCH + W3_PLANE) = l;
                }
            }
            if (run) {
                f16 *ys = sm + (nrun & 1) * W3_DY + c4;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    f16x4 h, l;
                    f16_split4(S.y[v], (S.ok >> (3 + v)) & 1u ? sy : 0.f, h, l);
                    *reinterpret_cast<f16x4 *>(ys + (px + 16 * v) * W3_PITCH) = h;
                    *reinterpret_cast<f16x4 *>(ys + (px + 16 * v) * W3_PITCH + W3_PLANE) = l;
                }
            }
        }
        // refill: the step after next (past the end of the walk: a staging-only step of nothing - its loads hit the tensors' first bytes)
        if (left > 0) S.m = next();   // (a strip's two staging-only steps come before its first run step: they never end a walk)
        else S.m = Micro{0, 0, -1, 0, -1};
        load(S);
        // LDS-only barrier (lgkmcnt(0) + s_barrier): __syncthreads() would also wait for the loads just issued (vmcnt(0))
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (run) {
            if (!(a.abl & 1)) {
                const f16 *ys = sm + (nrun & 1) * W3_DY;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    const int r0 = 16 * st + 8 * half;
                    const f16x8 ah = w3_frag(ys, r0, coh * 32, lane), al = w3_frag(ys + W3_PLANE, r0, coh * 32, lane);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        if (!((KYM >> ky) & 1)) continue;
                        constexpr int below[3] = {0, KYM & 1, (KYM & 1) + ((KYM >> 1) & 1)};   // computed rows before ky
                        const int t0 = below[ky] * NKX;
                        const f16 *xp = sm + 2 * W3_DY + ((q + 2 + ky) & 3) * W3_XROW;   // rows q - 2, q - 1, q = image rows oy - 1, oy, oy + 1
                        const Frag12 fh = w3_frag12(xp, r0, cih * 32, lane), fl = w3_frag12(xp + W3_PLANE, r0, cih * 32, lane);
                        auto tap = [&](f32x16 &c, const f16x8 &bh, const f16x8 &bl) __attribute__((always_inline)) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
                        };
                        if constexpr (KXM & 1) tap(acc[t0], w3_window<0>(fh), w3_window<0>(fl));
                        if constexpr (KXM & 2) tap(acc[t0 + (KXM & 1)], w3_window<1>(fh), w3_window<1>(fl));
                        if constexpr (KXM & 4) tap(acc[t0 + (KXM & 1) + ((KXM >> 1) & 1)], w3_window<2>(fh), w3_window<2>(fl));
                    }
                }
            }
            ++nrun;
        }
        q += filled;
    };
    // Both sets start EMPTY (run = -2) and are filled by step() itself: every global load of the kernel is issued from the one loop body, so
    // the vmcnt hipcc derives for "set A's five loads, with B's five younger ones in flight" is exact (with a prologue that loaded A and B, the
    // merge of the entry path with the back edge made it wait for everything: vmcnt(0) at the head of every step).  run = -1: end of the walk.
    Stage A, B;
    A.m = B.m = Micro{0, 0, -1, 0, left > 0 ? -2 : -1};
    A.ok = B.ok = 0u;
#pragma unroll
    for (int v = 0; v < 3; ++v) A.x[v] = B.x[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    A.y[0] = A.y[1] = B.y[0] = B.y[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    while (true) {
        if (A.m.run == -1) break;
        step(A);
        if (B.m.run == -1) break;
        step(B);
    }
    // the workgroup's partial tile [tap][co][ci]: a half-wave stores 32 consecutive ci = 128 bytes
    const float un = 1.0f / (sy * sx);
    const int ci = cit * 64 + cih * 32 + l31;
    float *out = a.part + item * ((long)NT * a.Cout * a.Cin);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cot * 64 + coh * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            out[((long)t * a.Cout + co) * a.Cin + ci] = acc[t][r] * un;
        }
}
// dw[co][ci][tap] = sum over the groups of part[g][t][co][ci], tap = nibble t of tapmap; kk taps per (co, ci) in dw (9 or 1)
__global__ __launch_bounds__(256) void wgrad3_reduce_kernel(const float *__restrict__ part, int groups, int Cout, int Cin, float *__restrict__ dw, int nt,
                                                            unsigned long tapmap, int kk) {
    const long n = (long)nt * Cout * Cin, cc = (long)Cout * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int g = 0;
        for (; g + 3 < groups; g += 4) {
            const float a0 = part[g * n + i], a1 = part[(g + 1) * n + i], a2 = part[(g + 2) * n + i], a3 = part[(g + 3) * n + i];
            s0 += a0;
            s1 += a1;
            s2 += a2;
            s3 += a3;
        }
        for (; g < groups; ++g) s0 += part[g * n + i];
        const long t = i / cc, rem = i - t * cc;
        dw[rem * kk + (long)((tapmap >> (4 * t)) & 15ul)] = (s0 + s1) + (s2 + s3);
    }
}
// out[slice][i] = sum of part[p][i] over the parts p of slice blockIdx.y (gridDim.y slices of `per` parts): called twice (parts -> 32 slices -> 1),
// so that no thread adds more than 32 values in sequence (one launch over 1024 parts took 0.28 ms for 38 MB)
__global__ void wgrad_reduce_kernel(const float *__restrict__ part, int n_parts, int per, long n, float *__restrict__ out) {
    const int p0 = blockIdx.y * per, p1 = p0 + per < n_parts ? p0 + per : n_parts;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int p = p0;
        for (; p + 3 < p1; p += 4) {
            const float a0 = part[p * n + i], a1 = part[(p + 1) * n + i], a2 = part[(p + 2) * n + i], a3 = part[(p + 3) * n + i];
            s0 += a0;
            s1 += a1;
            s2 += a2;
            s3 += a3;
        }
        for (; p < p1; ++p) s0 += part[p * n + i];
        out[(long)blockIdx.y * n + i] = (s0 + s1) + (s2 + s3);
    }
}

// ---- weight gradient of the ResNet stem: conv 7 x 7, stride 2, padding 3, 3 -> 64 channels, input frames NCHW.
// dW[co][ci][ky][kx] = sum over output pixels of dY[pixel][co] x[ci][2 oy + ky - 3][2 ox + kx - 3]: M = 64 output channels, N = 147 columns
// (ky, kx, ci) padded to 160, K = pixels.  A workgroup walks segments of 32 output pixels of one output row: it stages dY (32 x 64) and the
// TRANSPOSED im2col block of the 7 x 69 x 3 input tile, IM[column (ky, kx, ci)][pixel 0 .. 31], as fp16 hi | lo planes - an input value
// x[ky][cc][ci] is column (ky, kx, ci) of pixel (cc - kx) / 2 for the <= 4 kx of its parity: up to four 2-byte LDS stores per value and plane,
// offsets and masks fixed per thread - so that a lane's B fragment (its column's 8 consecutive pixels) is ONE 16-byte LDS read (80-byte row
// pitch: the rows of a ds_read_b128 group fall into distinct 16-byte slots).  Wave (co half, column-tile parity) takes the dY fragments from
// transposing reads.  (The first version kept the tile channel-interleaved and GATHERED the 8 pixels, 12 bytes apart, with 16-bit reads: 96
// reads + packing per lane and step for 18 MFMAs - 2.6 ms per 160 frames.)  Partial tiles per workgroup, summed by wgrad_reduce_kernel into
// torch's (64, 3, 7, 7) layout.
struct StemWgradArgs {
    const float *dy;     // [N][Hc][Wc][64]
    const float *x;      // [N][3][H][W]
    const unsigned *dy_amax, *x_amax;
    float *part;         // [n_items][64][3][7][7]
    int N, H, W, Hc, Wc, segs_per_row, steps_per_item;
    long n_steps;        // N * Hc * segs_per_row
};
constexpr int SW_RP = 40;                          // halfs per im2col row: 32 pixels + 8 (80 bytes)
constexpr int SW_DY = 32 * W3_PITCH;               // halfs of the dY plane (as conv_wgrad3_kernel)
constexpr int SW_PLANE = SW_DY + 147 * SW_RP;
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(StemWgradArgs a) {
    __shared__ __attribute__((aligned(16))) f16 sm[2 * SW_PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int coh = wave & 1, par = wave >> 1;     // output-channel half; column tiles par, par + 2 (, par + 4)
    const float sy = f16_scale_from_bits(*a.dy_amax), sx = f16_scale_from_bits(*a.x_amax);
    constexpr int NT = 3;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // this lane's column of every tile: c = 32 (par + 2 t) + l31 -> (ky, 3 kx + ci); columns >= 147 read the tile's first element with a zero flag
    int coff[NT];
    bool cok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c = 32 * (par + 2 * t) + l31;
        cok[t] = c < 147 && par + 2 * t < 5;
        coff[t] = SW_DY + (cok[t] ? c : 0) * SW_RP;
    }
    const long s0 = (long)blockIdx.x * a.steps_per_item, s1 = s0 + a.steps_per_item < a.n_steps ? s0 + a.steps_per_item : a.n_steps;
    const int c4 = (tid & 15) * 4;
    // this thread's six elements of the 7 x 3 x 69 input tile (the same for every step): row rr, channel ci, column cc; xo = the half index of
    // its first im2col copy (kx = cc & 1, pixel (cc - kx) / 2), the next kx of that parity is 6 rows further and one pixel back; xm = which of the
    // four copies exist (kx <= 6, pixel in 0 .. 31)
    constexpr int NX = (7 * 3 * 69 + 255) / 256;
    int xrr[NX], xci[NX], xcc[NX], xo[NX];
    unsigned xm[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int i = tid + 256 * k, rr = i / 207, rem = i - rr * 207;
        xrr[k] = i < 7 * 3 * 69 ? rr : -100;   // (-100: never inside the frame)
        xci[k] = rem / 69;
        xcc[k] = rem - xci[k] * 69;
        const int q = xcc[k] & 1, p0 = (xcc[k] - q) >> 1;
        xo[k] = SW_DY + (rr * 21 + q * 3 + xci[k]) * SW_RP + p0;
        xm[k] = 0u;
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (xrr[k] >= 0 && q + 2 * m <= 6 && p0 - m >= 0 && p0 - m < 32) xm[k] |= 1u << m;
    }
    // the operands of step st, in registers one step ahead: the global loads of step st + 1 are in flight during step st's gathers and MFMAs
    // (loaded at their point of use, every step paid an HBM round trip: 3.5 us per step for 0.24 us of MFMAs)
    f32x4 yv[2];
    float xv[NX];
    auto load = [&](long st) __attribute__((always_inline)) {
        const int seg = (int)(st % a.segs_per_row);
        const long row = st / a.segs_per_row;          // n * Hc + oy
        const int n = (int)(row / a.Hc), oy = (int)(row - (long)n * a.Hc), ox0 = 32 * seg;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int ox = ox0 + (tid >> 4) + 16 * v;
            yv[v] = ox < a.Wc ? *reinterpret_cast<const f32x4 *>(a.dy + (row * a.Wc + ox) * 64L + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int iy = 2 * oy - 3 + xrr[k], ix = 2 * ox0 - 3 + xcc[k];
            xv[k] = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? a.x[(((long)n * 3 + xci[k]) * a.H + iy) * a.W + ix] : 0.f;
        }
    };
    if (s0 < s1) load(s0);
    for (long st = s0; st < s1; ++st) {
        __syncthreads();   // the previous step's readers are done
        // dY: 32 pixels x 16 float4
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            f16x4 h, l;
            f16_split4(yv[v], sy, h, l);
            f16 *o = sm + ((tid >> 4) + 16 * v) * W3_PITCH + c4;
            *reinterpret_cast<f16x4 *>(o) = h;
            *reinterpret_cast<f16x4 *>(o + SW_PLANE) = l;
        }
        // input tile: rows 2 oy - 3 .. + 6, columns 2 ox0 - 3 .. + 68, 3 channels interleaved: half index 3 * col + ci
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const float v = xv[k] * sx;
            const f16 h = (f16)v, l = (f16)(v - (float)h);
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if ((xm[k] >> m) & 1u) {
                    f16 *o = sm + xo[k] + m * (6 * SW_RP - 1);
                    o[0] = h;
                    o[SW_PLANE] = l;
                }
        }
        if (st + 1 < s1) load(st + 1);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r0 = 16 * ks + 8 * half;
            const f16x8 ah = w3_frag(sm, r0, coh * 32, lane), al = w3_frag(sm + SW_PLANE, r0, coh * 32, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (par + 2 * t >= 5) continue;   // wave-uniform
                const f16x8 z8 = {(f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0, (f16)0};
                const f16 *at = sm + coff[t] + r0;   // this lane's column, pixels r0 .. r0 + 7
                const f16x8 bh = cok[t] ? *reinterpret_cast<const f16x8 *>(at) : z8, bl = cok[t] ? *reinterpret_cast<const f16x8 *>(at + SW_PLANE) : z8;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    }
    // column c = (ky, kx, ci) -> torch's W[co][ci][ky][kx]
    const float un = 1.0f / (sy * sx);
    float *out = a.part + (long)blockIdx.x * (64 * 147);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (par + 2 * t >= 5 || !cok[t]) continue;
        const int c = 32 * (par + 2 * t) + l31, ky = c / 21, kc = c - 21 * ky, kx = kc / 3, ci = kc - 3 * kx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = coh * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            out[((co * 3 + ci) * 7 + ky) * 7 + kx] = acc[t][r] * un;
        }
    }
}

}   // namespace cvt

static unsigned blocks_for(long n, long per_block, long cap) {
    long b = (n + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}
// grid of the per-channel reductions: ~ 64 K elements per workgroup (a workgroup of 1024 PIXELS left the 512-channel maps of layer 4 with 47
// workgroups), at most 2048
static unsigned reduce_blocks(long npix, int C) { return blocks_for(npix, 65536 / C, 2048); }
static bool bn_shape_ok(long npix, int C) { return npix > 0 && (C == 64 || C == 128 || C == 256 || C == 512 || C == 1024 || C == 2048); }

extern "C" size_t sd_bn_scratch_floats(int64_t npix, int C) { return bn_shape_ok(npix, C) ? (size_t)reduce_blocks(npix, C) * 2 * C : 0; }

extern "C" int sd_bn_train_fwd(const float *y, const float *gamma, const float *beta, const float *res, float *z, float *mean, float *rstd,
                               float *running_mean, float *running_var, double *acc, float *scratch, uint32_t *z_amax, int64_t npix, int C,
                               float eps, float momentum, int relu, void *stream) {
    if (!y || !gamma || !beta || !z || !mean || !rstd || !acc || !scratch || !bn_shape_ok(npix, C)) return fail(SD_E_BADARG, "sd_bn_train_fwd: null pointer or bad shape");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(gamma) |
         reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15)
        return fail(SD_E_BADARG, "sd_bn_train_fwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = reduce_blocks(npix, C);
    SD_LAUNCH(cvt::bn_stats_kernel, dim3(nb), dim3(256), 0, st, y, (long)npix, C, scratch);
    SD_CHECK_LAUNCH("bn_stats_kernel");
    SD_LAUNCH(cvt::partial_sum_kernel, dim3(2 * C / 32), dim3(256), 0, st, scratch, (int)nb, 2 * C, acc);
    SD_CHECK_LAUNCH("partial_sum_kernel");
    SD_LAUNCH(cvt::bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, acc, y, (long)npix, C, eps, momentum, mean, rstd, running_mean, running_var);
    SD_CHECK_LAUNCH("bn_finalize_kernel");
    const long n4 = npix * C / 4;
    SD_LAUNCH(cvt::bn_apply_kernel, dim3(blocks_for(n4, 1024, 4096)), dim3(256), 0, st, y, mean, rstd, gamma, beta, res, z, n4, C, relu, z_amax);
    SD_CHECK_LAUNCH("bn_apply_kernel");
    return 0;
}

extern "C" int sd_bn_train_bwd(const float *dz, const float *z, const float *y, const float *mean, const float *rstd, const float *gamma,
                               const float *beta, float *dy,
                               float *dres, float *dgamma, float *dbeta, double *acc, float *scratch, uint32_t *dy_amax, int64_t npix, int C, int relu,
                               void *stream) {
    if (!dz || !y || !mean || !rstd || !gamma || !dy || !dgamma || !dbeta || !acc || !scratch || (relu && !z && !beta) || !bn_shape_ok(npix, C))
        return fail(SD_E_BADARG, "sd_bn_train_bwd: null pointer or bad shape");
    if (relu && !z && dres) return fail(SD_E_BADARG, "sd_bn_train_bwd: a unit with a residual operand needs z for its ReLU mask");
    if ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) |
         reinterpret_cast<uintptr_t>(dres) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(mean) |
         reinterpret_cast<uintptr_t>(rstd)) & 15)
        return fail(SD_E_BADARG, "sd_bn_train_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = reduce_blocks(npix, C);
    SD_LAUNCH(cvt::bn_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, dz, z, y, mean, rstd, gamma, beta, (long)npix, C, relu, scratch);
    SD_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    SD_LAUNCH(cvt::partial_sum_kernel, dim3(2 * C / 32), dim3(256), 0, st, scratch, (int)nb, 2 * C, acc);
    SD_CHECK_LAUNCH("partial_sum_kernel");
    SD_LAUNCH(cvt::bn_bwd_apply_kernel, dim3(blocks_for(npix * C / 4, 1024, 4096)), dim3(256), 0, st, dz, z, y, mean, rstd, gamma, beta, acc, dy, dres, dgamma,
              dbeta, (long)npix, C, relu, dy_amax);
    SD_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return 0;
}

extern "C" int sd_bn_relu_pool_fwd(const float *y, const float *gamma, const float *beta, float *p, uint32_t *idx, float *mean, float *rstd,
                                   float *running_mean, float *running_var, double *acc, float *scratch, uint32_t *p_amax, int N, int Hc, int Wc,
                                   int C, float eps, float momentum, void *stream) {
    const long npix = (long)N * Hc * Wc;
    if (!y || !gamma || !beta || !p || !idx || !mean || !rstd || !acc || !scratch || N <= 0 || Hc <= 0 || Wc <= 0 || !bn_shape_ok(npix, C))
        return fail(SD_E_BADARG, "sd_bn_relu_pool_fwd: null pointer or bad shape");
    if (npix * (C / 4) >= (1l << 31)) return fail(SD_E_TOOBIG, "sd_bn_relu_pool_fwd: 2^31 or more 16-byte elements");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
         reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15)
        return fail(SD_E_BADARG, "sd_bn_relu_pool_fwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = reduce_blocks(npix, C);
    SD_LAUNCH(cvt::bn_stats_kernel, dim3(nb), dim3(256), 0, st, y, npix, C, scratch);
    SD_CHECK_LAUNCH("bn_stats_kernel");
    SD_LAUNCH(cvt::partial_sum_kernel, dim3(2 * C / 32), dim3(256), 0, st, scratch, (int)nb, 2 * C, acc);
    SD_CHECK_LAUNCH("partial_sum_kernel");
    SD_LAUNCH(cvt::bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, acc, y, npix, C, eps, momentum, mean, rstd, running_mean, running_var);
    SD_CHECK_LAUNCH("bn_finalize_kernel");
    const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
    SD_LAUNCH(cvt::bn_relu_pool_fwd_kernel, dim3(blocks_for((long)N * Hp * Wp * (C / 4), 512, 8192)), dim3(256), 0, st, y, mean, rstd, gamma, beta, p, idx, N, Hc,
              Wc, Hp, Wp, C, p_amax);
    SD_CHECK_LAUNCH("bn_relu_pool_fwd_kernel");
    return 0;
}
extern "C" int sd_bn_relu_pool_bwd(const float *dp, const uint32_t *idx, const float *y, const float *mean, const float *rstd, const float *gamma,
                                   float *dy, float *dgamma, float *dbeta, double *acc, float *scratch, uint32_t *dy_amax, int N, int Hc, int Wc, int C,
                                   void *stream) {
    const long npix = (long)N * Hc * Wc;
    if (!dp || !idx || !y || !mean || !rstd || !gamma || !dy || !dgamma || !dbeta || !acc || !scratch || N <= 0 || Hc <= 0 || Wc <= 0 || !bn_shape_ok(npix, C))
        return fail(SD_E_BADARG, "sd_bn_relu_pool_bwd: null pointer or bad shape");
    if (npix * (C / 4) >= (1l << 31)) return fail(SD_E_TOOBIG, "sd_bn_relu_pool_bwd: 2^31 or more 16-byte elements");
    if ((reinterpret_cast<uintptr_t>(dp) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(gamma) |
         reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15)
        return fail(SD_E_BADARG, "sd_bn_relu_pool_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const cvt::PoolGeom G{Hc, Wc, (Hc - 1) / 2 + 1, (Wc - 1) / 2 + 1, C, (Hc + 1) / 2, (Wc + 1) / 2};
    const long nitems = (long)N * G.Hb * G.Wb;   // 2 x 2 blocks of convolution pixels
    const unsigned nb = reduce_blocks(npix, C);    // (also what sd_bn_scratch_floats sized the scratch for)
    SD_LAUNCH(cvt::bn_pool_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, dp, idx, y, mean, rstd, nitems, G, scratch);
    SD_CHECK_LAUNCH("bn_pool_bwd_reduce_kernel");
    SD_LAUNCH(cvt::partial_sum_kernel, dim3(2 * C / 32), dim3(256), 0, st, scratch, (int)nb, 2 * C, acc);
    SD_CHECK_LAUNCH("partial_sum_kernel");
    SD_LAUNCH(cvt::bn_pool_bwd_apply_kernel, dim3(blocks_for(nitems * C / 4, 512, 4096)), dim3(256), 0, st, dp, idx, y, mean, rstd, gamma, acc, dy, dgamma, dbeta,
              npix, nitems, G, dy_amax);
    SD_CHECK_LAUNCH("bn_pool_bwd_apply_kernel");
    return 0;
}

// conv_wgrad3_kernel's persistent grid: ~ 512 workgroups (two per CU) = (co tile, ci tile) x groups of steps
static int wgrad3_groups(long steps, int Cin, int Cout) {
    const long tiles = (long)(Cout / 64) * (Cin / 64);
    long groups = 512 / tiles;
    if (groups > steps) groups = steps;
    if (groups < 1) groups = 1;
    return (int)groups;
}
extern "C" size_t sd_conv_wgrad_scratch_floats(int N, int H, int W, int Cin, int Cout, int ksize, int stride) {
    if (ksize != 3 || (stride != 1 && stride != 2) || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int nt = stride == 1 ? 9 : 4;   // the most taps one launch computes
    return (size_t)wgrad3_groups((long)N * Ho * ((Wo + 31) / 32), Cin, Cout) * nt * Cout * Cin;
}
// one launch of the strip-walking kernel + its reduction: shifts (KYM, KXM) of dY against the class (xs, xr, xc) of x; tapmap: compact tap t -> tap of dw
template <int KYM, int KXM>
static int wgrad3_launch(cvt::Wgrad3Args w, float *dw, unsigned long tapmap, int kk, hipStream_t st) {
    constexpr int NT = (((KYM >> 0) & 1) + ((KYM >> 1) & 1) + ((KYM >> 2) & 1)) * (((KXM >> 0) & 1) + ((KXM >> 1) & 1) + ((KXM >> 2) & 1));
    const long wg3 = (long)(w.Cout / 64) * (w.Cin / 64) * w.groups;
    if (wg3 > 0x7fffffffL) return fail(SD_E_TOOBIG, "sd_conv_wgrad: too many work items");
    SD_LAUNCH((cvt::conv_wgrad3_kernel<KYM, KXM>), dim3((unsigned)wg3), dim3(256), 0, st, w);
    SD_CHECK_LAUNCH("conv_wgrad3_kernel");
    SD_LAUNCH(cvt::wgrad3_reduce_kernel, dim3(grid_for((long)NT * w.Cout * w.Cin)), dim3(256), 0, st, w.part, w.groups, w.Cout, w.Cin, dw, NT, tapmap, kk);
    SD_CHECK_LAUNCH("wgrad3_reduce_kernel");
    return 0;
}
extern "C" int sd_conv_wgrad(const float *dy, const float *x, const uint32_t *dy_amax, const uint32_t *x_amax, float *dw, float *scratch, int N, int H,
                             int W, int Cin, int Cout, int ksize, int stride, void *stream) {
    if (!dy || !x || !dw || N <= 0 || H <= 0 || W <= 0) return fail(SD_E_BADARG, "sd_conv_wgrad: null pointer or empty shape");
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return fail(SD_E_BADARG, "sd_conv_wgrad: kernel size 1 or 3, stride 1 or 2");
    if (Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return fail(SD_E_BADDIM, "sd_conv_wgrad: channels must be positive multiples of 64");
    const int pad = ksize / 2, Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    static const char *simple = getenv("SD_WGRAD_SIMPLE");   // A/B runs: the per-wave kernel for every shape
    // (1 x 1 convolutions stay on the per-wave kernel below: one tap per staging does not pay - 0.17 against 0.19 ms for the stride-2 shortcuts)
    if (ksize == 3 && dy_amax && x_amax && scratch && !(simple && simple[0] == '1') && !((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15)) {
        // the strip-walking kernel over dY's (Ho, Wo) pixels; x through the class (xs, xr, xc)
        static const char *abl_env = getenv("SD_W3_ABL");
        hipStream_t st = (hipStream_t)stream;
        cvt::Wgrad3Args w{dy, x, dy_amax, x_amax, scratch, N, Ho, Wo, Cin, Cout, (Wo + 31) / 32, 0, H, W, stride, 0, 0, 0, 0, abl_env ? atoi(abl_env) : 0};
        w.steps = (long)N * Ho * w.nstrips;
        w.groups = wgrad3_groups(w.steps, Cin, Cout);
        w.per_group = (w.steps + w.groups - 1) / w.groups;
        if (stride == 1) return wgrad3_launch<7, 7>(w, dw, 0x876543210ul, 9, st);
        // stride 2: tap (ky, kx) of dw = shift (ky' - 1, kx' - 1) of class (pr, pc): odd class ky' = 0, 1 -> ky = 0, 2; even class ky' = 1 -> ky = 1
        w.xr = 1; w.xc = 1;
        if (int rc = wgrad3_launch<3, 3>(w, dw, 0x8620ul, 9, st)) return rc;                  // (0,0) (0,2) (2,0) (2,2)
        w.xr = 1; w.xc = 0;
        if (int rc = wgrad3_launch<3, 2>(w, dw, 0x71ul, 9, st)) return rc;                    // (0,1) (2,1)
        w.xr = 0; w.xc = 1;
        if (int rc = wgrad3_launch<2, 3>(w, dw, 0x53ul, 9, st)) return rc;                    // (1,0) (1,2)
        w.xr = 0; w.xc = 0;
        return wgrad3_launch<2, 2>(w, dw, 0x4ul, 9, st);                                      // (1,1)
    }
    cvt::WgradArgs a{dy, x, dy_amax, x_amax, dw, N, H, W, Ho, Wo, Cin, Cout, ksize, stride, pad, 0, 0};
    // ~4096 pixels per work item: long enough that a tile's 4096 atomics are a small part of its work, short enough to fill the chip
    a.rows_per_item = (4096 + Wo - 1) / Wo;
    const long rows = (long)N * Ho;
    while (a.rows_per_item > 1 && (rows + a.rows_per_item - 1) / a.rows_per_item * (Cout / 64) * (Cin / 64) * ksize * ksize < 2048) a.rows_per_item = (a.rows_per_item + 1) / 2;
    a.n_row_items = (int)((rows + a.rows_per_item - 1) / a.rows_per_item);
    const long items = (long)(Cout / 64) * (Cin / 64) * ksize * ksize * a.n_row_items;
    const long wgs = (items + 3) / 4;
    if (wgs > 0x7fffffffL) return fail(SD_E_TOOBIG, "sd_conv_wgrad: too many work items");
    if (dy_amax && x_amax) SD_LAUNCH(cvt::conv_wgrad_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    else SD_LAUNCH(cvt::conv_wgrad_kernel<false>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    SD_CHECK_LAUNCH("conv_wgrad_kernel");
    return 0;
}

static long stem_wgrad_items(long n_steps) { return n_steps < 512 ? n_steps : 512; }   // two workgroups per CU, one round
extern "C" size_t sd_stem_wgrad_scratch_floats(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;
    const long n_steps = (long)N * Hc * ((Wc + 31) / 32);
    return (size_t)(stem_wgrad_items(n_steps) + 32) * 64 * 147;   // the workgroups' partial tiles + 32 slices of the first reduction
}
extern "C" int sd_stem_wgrad(const float *dy, const float *x, const uint32_t *dy_amax, const uint32_t *x_amax, float *dw, float *scratch, int N, int H,
                             int W, void *stream) {
    if (!dy || !x || !dy_amax || !x_amax || !dw || !scratch || N <= 0 || H <= 0 || W <= 0) return fail(SD_E_BADARG, "sd_stem_wgrad: null pointer or empty shape");
    if (reinterpret_cast<uintptr_t>(dy) & 15) return fail(SD_E_BADARG, "sd_stem_wgrad: dy must be 16-byte aligned");
    cvt::StemWgradArgs a{};
    a.dy = dy; a.x = x; a.dy_amax = dy_amax; a.x_amax = x_amax; a.part = scratch;
    a.N = N; a.H = H; a.W = W; a.Hc = (H - 1) / 2 + 1; a.Wc = (W - 1) / 2 + 1;
    a.segs_per_row = (a.Wc + 31) / 32;
    a.n_steps = (long)N * a.Hc * a.segs_per_row;
    const long items = stem_wgrad_items(a.n_steps);
    a.steps_per_item = (int)((a.n_steps + items - 1) / items);
    const long wgs = (a.n_steps + a.steps_per_item - 1) / a.steps_per_item;
    SD_LAUNCH(cvt::stem_wgrad_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    SD_CHECK_LAUNCH("stem_wgrad_kernel");
    const long n = 64L * 147;
    const int per = (int)((wgs + 31) / 32), slices = (int)((wgs + per - 1) / per);
    float *tmp = scratch + items * n;
    SD_LAUNCH(cvt::wgrad_reduce_kernel, dim3(grid_for(n), slices), dim3(256), 0, (hipStream_t)stream, scratch, (int)wgs, per, n, tmp);
    SD_CHECK_LAUNCH("wgrad_reduce_kernel");
    SD_LAUNCH(cvt::wgrad_reduce_kernel, dim3(grid_for(n), 1), dim3(256), 0, (hipStream_t)stream, tmp, slices, slices, n, dw);
    SD_CHECK_LAUNCH("wgrad_reduce_kernel");
    return 0;
}
