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

namespace cvt {

__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

// ---- per-channel reductions over an NHWC tensor: thread -> 4 consecutive channels (one 16-byte load per pixel), a block walks its
// pixel range in steps of 256 * 4 / C pixels; lanes with the same channel group are combined through LDS, blocks through double atomics
template <class F>   // F(pixel index, channel group base, float4 of the tensor) -> accumulates into two f32x4
__device__ __forceinline__ void channel_reduce(long npix, int C, double *acc /* [C][2] */, F f) {
    __shared__ float red[256 * 8];
    const int cg = C / 4, tid = threadIdx.x;
    const int c4 = tid % cg, prow = tid / cg, pstep = 256 / cg;   // C in {64, 128, 256, 512}: cg in {16 .. 128} divides 256
    const long per = (npix + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
    for (long p = p0 + prow; p < p1; p += pstep) f(p, 4 * c4, s, q);
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
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomic_add_f64(acc + (4 * tid + e) * 2, (double)s[e]);
            atomic_add_f64(acc + (4 * tid + e) * 2 + 1, (double)q[e]);
        }
    }
}

// sums of (y - pivot) and (y - pivot)^2 per channel; pivot = the first pixel's value of the channel (keeps the variance formula
// q / n - (s / n)^2 free of the cancellation a large mean would cause)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ y, long npix, int C, double *acc) {
    channel_reduce(npix, C, acc, [&](long p, int c0, f32x4 &s, f32x4 &q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(y + p * C + c0) - *reinterpret_cast<const f32x4 *>(y + c0);
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
// z = relu?((y - mean) rstd gamma + beta (+ res)); abs-max of z -> word
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ y, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ res,
                                                       float *__restrict__ z, long n4, int C, int relu, unsigned *amax) {
    float mx = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)((4 * i) % C);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(rstd + c0) * *reinterpret_cast<const f32x4 *>(gamma + c0);
        f32x4 v = (*reinterpret_cast<const f32x4 *>(y + 4 * i) - *reinterpret_cast<const f32x4 *>(mean + c0)) * sc + *reinterpret_cast<const f32x4 *>(beta + c0);
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
// backward reductions: acc[c] = (sum g, sum g x_hat), g = dz (where z > 0 if relu), x_hat = (y - mean) rstd
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ dz, const float *__restrict__ z, const float *__restrict__ y,
                                                            const float *__restrict__ mean, const float *__restrict__ rstd, long npix, int C, int relu,
                                                            double *acc) {
    channel_reduce(npix, C, acc, [&](long p, int c0, f32x4 &s, f32x4 &q) {
        f32x4 g = *reinterpret_cast<const f32x4 *>(dz + p * C + c0);
        if (relu) {
            const f32x4 zz = *reinterpret_cast<const f32x4 *>(z + p * C + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        }
        const f32x4 xh = (*reinterpret_cast<const f32x4 *>(y + p * C + c0) - *reinterpret_cast<const f32x4 *>(mean + c0)) * *reinterpret_cast<const f32x4 *>(rstd + c0);
        s = s + g;
        q = q + g * xh;
    });
}
// dy = gamma rstd (g - s1 / n - x_hat s2 / n); dgamma = s2, dbeta = s1 (block 0 writes them); g -> dres (the residual branch's gradient) if asked
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ dz, const float *__restrict__ z, const float *__restrict__ y,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                           const double *__restrict__ acc, float *__restrict__ dy, float *__restrict__ dres,
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
        if (relu) {
            const f32x4 zz = *reinterpret_cast<const f32x4 *>(z + 4 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = zz[e] > 0.f ? g[e] : 0.f;
        }
        if (dres) *reinterpret_cast<f32x4 *>(dres + 4 * i) = g;
        const f32x4 rs = *reinterpret_cast<const f32x4 *>(rstd + c0);
        const f32x4 xh = (*reinterpret_cast<const f32x4 *>(y + 4 * i) - *reinterpret_cast<const f32x4 *>(mean + c0)) * rs;
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

// ---- convolution weight gradient.  One WAVE owns a 64 (co) x 64 (ci) tile of dW for ONE tap and a group of `rows_per_item` output image
// rows (n, oy); it walks their pixels 32 at a time: A[i = co][k = pixel] = dY[pixel][co0 + lane & 31 (+ 32)], B[k = pixel][j = ci] =
// X[pixel * stride + tap - pad][ci0 + lane & 31 (+ 32)] (zero outside the image), both read as they lie in memory (NHWC: a pixel's channels
// are one 128-byte segment per half-wave), split into fp16 hi / lo with one power-of-two scale per operand and 32 pixels (block floating
// point), 24 MFMAs into a sub-accumulator that is added un-scaled to the wave's fp32 accumulators; at the end fp32 atomics into
// dW (Cout, Cin, k, k).
struct WgradArgs {
    const float *dy;   // [N][Ho][Wo][Cout]
    const float *x;    // [N][H][W][Cin]
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
                const float v = acc[tm][tn][r];
                if (v != 0.f) atomicAdd(a.dw + ((long)co * a.Cin + ci) * taps + tap, v);
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
static bool bn_shape_ok(long npix, int C) { return npix > 0 && (C == 64 || C == 128 || C == 256 || C == 512 || C == 1024 || C == 2048); }

extern "C" int sd_bn_train_fwd(const float *y, const float *gamma, const float *beta, const float *res, float *z, float *mean, float *rstd,
                               float *running_mean, float *running_var, double *acc, uint32_t *z_amax, int64_t npix, int C, float eps,
                               float momentum, int relu, void *stream) {
    if (!y || !gamma || !beta || !z || !mean || !rstd || !acc || !bn_shape_ok(npix, C)) return fail(SD_E_BADARG, "sd_bn_train_fwd: null pointer or bad shape");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(gamma) |
         reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15)
        return fail(SD_E_BADARG, "sd_bn_train_fwd: tensors must be 16-byte aligned");
    if (C > 512) return fail(SD_E_BADDIM, "sd_bn_train_fwd: up to 512 channels");
    hipStream_t st = (hipStream_t)stream;
    SD_LAUNCH(cvt::bn_stats_kernel, dim3(blocks_for(npix, 1024, 2048)), dim3(256), 0, st, y, (long)npix, C, acc);
    SD_CHECK_LAUNCH("bn_stats_kernel");
    SD_LAUNCH(cvt::bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, acc, y, (long)npix, C, eps, momentum, mean, rstd, running_mean, running_var);
    SD_CHECK_LAUNCH("bn_finalize_kernel");
    const long n4 = npix * C / 4;
    SD_LAUNCH(cvt::bn_apply_kernel, dim3(blocks_for(n4, 1024, 4096)), dim3(256), 0, st, y, mean, rstd, gamma, beta, res, z, n4, C, relu, z_amax);
    SD_CHECK_LAUNCH("bn_apply_kernel");
    return 0;
}

extern "C" int sd_bn_train_bwd(const float *dz, const float *z, const float *y, const float *mean, const float *rstd, const float *gamma, float *dy,
                               float *dres, float *dgamma, float *dbeta, double *acc, uint32_t *dy_amax, int64_t npix, int C, int relu, void *stream) {
    if (!dz || !y || !mean || !rstd || !gamma || !dy || !dgamma || !dbeta || !acc || (relu && !z) || !bn_shape_ok(npix, C) || C > 512)
        return fail(SD_E_BADARG, "sd_bn_train_bwd: null pointer or bad shape");
    if ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) |
         reinterpret_cast<uintptr_t>(dres) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(rstd)) & 15)
        return fail(SD_E_BADARG, "sd_bn_train_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SD_LAUNCH(cvt::bn_bwd_reduce_kernel, dim3(blocks_for(npix, 1024, 2048)), dim3(256), 0, st, dz, z, y, mean, rstd, (long)npix, C, relu, acc);
    SD_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    SD_LAUNCH(cvt::bn_bwd_apply_kernel, dim3(blocks_for(npix * C / 4, 1024, 4096)), dim3(256), 0, st, dz, z, y, mean, rstd, gamma, acc, dy, dres, dgamma,
              dbeta, (long)npix, C, relu, dy_amax);
    SD_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return 0;
}

extern "C" int sd_conv_wgrad(const float *dy, const float *x, float *dw, int N, int H, int W, int Cin, int Cout, int ksize, int stride, void *stream) {
    if (!dy || !x || !dw || N <= 0 || H <= 0 || W <= 0) return fail(SD_E_BADARG, "sd_conv_wgrad: null pointer or empty shape");
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return fail(SD_E_BADARG, "sd_conv_wgrad: kernel size 1 or 3, stride 1 or 2");
    if (Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return fail(SD_E_BADDIM, "sd_conv_wgrad: channels must be positive multiples of 64");
    const int pad = ksize / 2, Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    cvt::WgradArgs a{dy, x, dw, N, H, W, Ho, Wo, Cin, Cout, ksize, stride, pad, 0, 0};
    // ~4096 pixels per work item: long enough that a tile's 4096 atomics are a small part of its work, short enough to fill the chip
    a.rows_per_item = (4096 + Wo - 1) / Wo;
    const long rows = (long)N * Ho;
    while (a.rows_per_item > 1 && (rows + a.rows_per_item - 1) / a.rows_per_item * (Cout / 64) * (Cin / 64) * ksize * ksize < 2048) a.rows_per_item = (a.rows_per_item + 1) / 2;
    a.n_row_items = (int)((rows + a.rows_per_item - 1) / a.rows_per_item);
    const long items = (long)(Cout / 64) * (Cin / 64) * ksize * ksize * a.n_row_items;
    const long wgs = (items + 3) / 4;
    if (wgs > 0x7fffffffL) return fail(SD_E_TOOBIG, "sd_conv_wgrad: too many work items");
    SD_LAUNCH(cvt::conv_wgrad_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a);
    SD_CHECK_LAUNCH("conv_wgrad_kernel");
    return 0;
}
