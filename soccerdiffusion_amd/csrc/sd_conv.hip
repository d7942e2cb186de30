// SoccerDiffusion image path: the 3 x 3, stride-1, padding-1 convolutions of ResNet-18's basic blocks with the
// inference-mode BatchNorm, the residual and the ReLU in their epilogue - reference: torchvision's BasicBlock as configured by
// soccer_diffusion/ml/model/encoder/image.py:55-83 (ResNetImageEncoder), 13 of the backbone's 20 convolutions and ~ 80 % of its FLOPs
// at 480 x 640.  Interface and citations: include/soccerdiffusion_hip.h (sd_conv3x3_*).
//
// Implicit GEMM on the fp16 matrix pipe with split operands: out[pixel][co] = sum over (tap, ci) of
// in[pixel + tap][ci] w[co][ci][tap] as three v_mfma_f32_32x32x16_f16 per product (lo.hi, hi.lo, hi.hi; fp32 accumulate), so the
// result is fp32-grade (parity with torch's fp32 CPU convolution at 1e-6), at the fp16 pipe's rate.
//   * Activations are NHWC fp32 in HBM: a pixel's channels are contiguous, so the contraction index of a tap is a 16-byte LDS
//     read and an output row is a 128-byte store.
//   * One workgroup (4 waves) owns an 8 x 16 pixel tile x 64 output channels.  Per 64-channel chunk of the input the 10 x 18
//     halo tile is staged ONCE through LDS as fp16 hi | lo planes (x scale: a power of two from the tensor's abs-max word, which the
//     producing launch left behind) and serves all 9 taps: 1.4 x the tile's own bytes instead of 9 x.  The planes are two arrays of
//     144-byte pixels (128 + 16: 9 sixteen-byte units, odd, so the 16 consecutive pixels of a ds_read_b128 lane group hit 16
//     distinct 16-byte bank groups).
//   * Wave (w & 1, w >> 1) = (pixel half: 4 x 16 pixels = two 32-row MFMA tiles, output-channel half: one 32-column tile).  Weight
//     fragments come straight from L2 in fragment-major planes ([co tile][tap][k-step][plane][lane][8 halfs], packed once per weight
//     update by conv3x3_pack_kernel), requested one k-step ahead; halo fragments likewise from LDS.
//   * Epilogue: acc / (s_in s_w) * bn_scale[co] + bn_shift[co] (+ residual) -> ReLU -> NHWC store, and the tile's abs-max into the
//     output's word for the next convolution.
// 52 KB of LDS: three workgroups per CU, so another workgroup's MFMAs cover a workgroup's halo staging.
#include "../../include/soccerdiffusion_hip.h"
#include "sd_common.h"

namespace cv {

constexpr int TH = 8, TW = 16, HH = TH + 2, HW = TW + 2;   // output tile and its halo
constexpr int CK = 64;                                      // input channels per chunk
constexpr int PIX = 2 * CK + 16;                            // bytes of a halo pixel in one fp16 plane (+ 16: bank spread)
constexpr int PLANE = HH * HW * PIX;                        // the lo plane follows the hi plane
constexpr int LDS_BYTES = 2 * PLANE;                        // 51 840
constexpr int COT = 64;                                     // output channels per workgroup
#ifndef SD_CONV_WD
#define SD_CONV_WD 4
#endif
constexpr int WD = SD_CONV_WD;                              // register stages of the weight-fragment ring (lookahead WD - 1 k-steps)

// row (0 / 1) of MFMA row i = 0..31 inside its 2 x 16 pixel block: columns 4 - 11 of the two rows swap lanes (see conv3x3_kernel)
__device__ __forceinline__ int row32(int i) { return (i >> 4) ^ ((((i & 15) + 4) >> 3) & 1); }
// XCD-aware work order of a 1-D grid over (tile, output-channel block): workgroup ids go round-robin over the 8 XCDs (id % 8), each
// with its own 4-MB L2.  The channel blocks of one tile read the same halo, so `g` of them (a divisor of yb whose packed weights fit
// half an L2 together: y_group) take CONSECUTIVE slots of ONE XCD's sequence: they run together and all but the first find the halo
// in that L2.  With the channel block as the grid's slow dimension every block fetched its input from HBM again (measured, 160 frames:
// layer 2 1.31 -> 0.79 GB per launch, layer 3 1.17 -> 0.51 GB; with all 8 blocks of layer 4 together their 9.4 MB of weights thrash
// the L2 instead: 1.17 -> 1.64 GB, hence the cap).  Time is unchanged either way: the kernels are not HBM-bound.  tile >= ntiles: idle.
__device__ __forceinline__ void xcd_tile(int yb, int g, int ntiles, int &tile, int &yblk) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, t8 = (ntiles + 7) >> 3;
    const int inner = k % g, rest = k / g;
    yblk = (rest / t8) * g + inner;
    tile = (rest % t8) * 8 + xcd;
}
inline int y_group(int yb, long weight_bytes_per_block) {
    int g = 1;
    while (g * 2 <= yb && yb % (g * 2) == 0 && (long)(g * 2) * weight_bytes_per_block <= (2L << 20)) g *= 2;
    return g;
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// abs-max of a tensor into ONE word (bits of a non-negative float: unsigned order = float order)
__global__ void absmax_word_kernel(const float *__restrict__ x, long n4, long n, unsigned *word) {
    float m = 0.f;
    if (blockIdx.x == 0 && 4 * n4 + threadIdx.x < n) m = fabsf(x[4 * n4 + threadIdx.x]);   // the tail of an n that is no multiple of 4
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(x + 4 * i);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(word, __builtin_bit_cast(unsigned, m));
}
// max |y| of a wave into the tensor's word.  Read first: after the first few workgroups nearly every wave finds the word at or above
// its value, and 10^5 .. 10^6 atomics on one address serialise in a single L2 channel (the stem kernel took 2x as long with them).
__device__ __forceinline__ void publish_amax(unsigned *word, float mx, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const unsigned b = __builtin_bit_cast(unsigned, mx);
    if (lane == 0 && b > __atomic_load_n(word, __ATOMIC_RELAXED)) atomicMax(word, b);
}
// (a hipMemsetAsync node replays garbage from a captured graph on this stack - NOTEBOOK.md 5.7: zero fills are kernels)
__global__ void zero_words_kernel(unsigned *p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0u;
}

// W (Cout, Cin, KS, KS) fp32 (torch layout; taps = KS^2 = 9 or 1) -> [co tile 32][tap][k-step Cin/16][plane][lane][8]: lane = 32 kg + j
// holds W[32 ct + j][16 ks + 8 kg + e][tap] * scale as hi / lo, e = 0..7.  scale: power of two from the weights' abs-max word.
__global__ void conv3x3_pack_kernel(const float *__restrict__ W, int Cout, int Cin, int taps, const unsigned *maxbits, f16 *__restrict__ dst, float *scale_out) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const int nks = Cin / 16;
    const long total = (long)Cout * taps * (Cin / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % (Cin / 8));
        const int tap = (int)((i / (Cin / 8)) % taps);
        const int co = (int)(i / (Cin / 8) / taps);
        f16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = W[((long)co * Cin + (c8 * 8 + e)) * taps + tap] * scale;
            hh[e] = (f16)v;
            ll[e] = (f16)(v - (float)hh[e]);
        }
        const int ct = co >> 5, j = co & 31, ks = c8 >> 1, kg = c8 & 1;
        f16 *o = dst + ((((long)ct * taps + tap) * nks + ks) * 2) * 512 + (32 * kg + j) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = hh[e];
            o[512 + e] = ll[e];
        }
    }
}

struct ConvArgs {
    const float *x;          // [N][H][W][Cin]
    const f16 *w;            // packed planes
    const float *w_scale;    // device word written by the pack kernel
    const unsigned *x_amax;  // abs-max word of x
    const float *bn_scale, *bn_shift;   // [Cout]
    const float *res;        // [N][H][W][Cout] or NULL
    float *y;                // [N][H][W][Cout]
    unsigned *y_amax;        // abs-max word of y, or NULL
    int N, H, W, Cin, Cout, relu, tiles_x, tiles_y;
    int ygroup;              // output-channel blocks that share a halo in one XCD's L2 (xcd_tile)
};

// KS = 3: the 3 x 3 / padding-1 convolution described above.  KS = 1: the same tile, epilogue and pipelines for a 1 x 1 convolution (ResNet-50's
// bottleneck projections): no halo, one tap, 4 k-steps per 64-channel chunk.
template <int KS>
__global__ __launch_bounds__(256, 3) void conv3x3_kernel(ConvArgs a) {
    constexpr int HH = TH + KS - 1, HW = TW + KS - 1, PAD = KS / 2, TAPS = KS * KS, NSTEP = TAPS * 4, PLANE = HH * HW * PIX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kg = lane >> 5;
    const int ph = w & 1, ch = w >> 1;                 // pixel half, output-channel half of this wave
    int t, yblk;
    xcd_tile(a.Cout / COT, a.ygroup, a.tiles_x * a.tiles_y * a.N, t, yblk);
    if (t >= a.tiles_x * a.tiles_y * a.N) return;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int co0 = yblk * COT + ch * 32;              // this wave's 32 output channels
    const float s_in = f16_scale_from_bits(*a.x_amax);
    const int nks = a.Cin / 16;
    const f16 *wbase = a.w + (long)(co0 >> 5) * TAPS * nks * 1024 + lane * 8;   // [tap][ks][plane]: 1024 halfs per (tap, ks)

    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    // this lane's A rows: pixel (row 4 ph + 2 m + row32(j), column j & 15) of the tile, as a halo offset for tap (0, 0).  row32 puts
    // the 16 lanes of each ds_read_b128 group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) on the 16 consecutive pixels of ONE halo
    // row (144-byte pitch: 16 distinct 16-byte slots); with row = j >> 4 a group straddled two rows and lost two of its slots (46 % of
    // the LDS cycles were conflicts)
    unsigned abase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) abase[m] = (unsigned)(((4 * ph + 2 * m + row32(j)) * HW + (j & 15)) * PIX + 16 * kg);

    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        if (c0 > 0) __syncthreads();   // every wave has consumed the previous chunk
        // ---- stage the halo tile of channels c0 .. c0 + 63: 180 pixels x 16 vectors of 4 channels.  All 12 loads of a thread are in
        // flight before the first split / LDS store (one HBM round trip per chunk, not twelve)
        constexpr int NST = (HH * HW * (CK / 4) + 255) / 256;
        f32x4 hv[NST];
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int i = tid + 256 * q, c4 = i & (CK / 4 - 1), p = i / (CK / 4);
            const int hy = p / HW, hx = p - hy * HW;
            const int gy = y0 + hy - PAD, gx = x0 + hx - PAD;
            hv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p < HH * HW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                hv[q] = *reinterpret_cast<const f32x4 *>(a.x + (((long)n * a.H + gy) * a.W + gx) * a.Cin + c0 + 4 * c4);
        }
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int i = tid + 256 * q, c4 = i & (CK / 4 - 1), p = i / (CK / 4);
            if (p >= HH * HW) continue;
            f16x4 h, l;
            f16_split4(hv[q], s_in, h, l);
            char *at = smem + p * PIX + 8 * c4;
            *reinterpret_cast<f16x4 *>(at) = h;
            *reinterpret_cast<f16x4 *>(at + PLANE) = l;
        }
        __syncthreads();
        // ---- 9 taps x 4 k-steps: weight fragments (global / L2) and halo fragments (LDS) one step ahead
        const f16 *wp = wbase + (long)(c0 / 16) * 1024;
        // a step is 6 MFMAs = 192 cycles of the pipe and a SIMD holds three waves: one step of lookahead is ~600 cycles against an L2
        // round trip of ~1 500 under load, so the weight fragments run WD - 1 steps ahead (a ring of WD register stages)
        f16x8 bw[WD][2], af[2][2][2];   // [ring stage][plane], [stage][tile][plane]
        auto load_w = [&](int s) __attribute__((always_inline)) {
            const int tap = s >> 2, ks = s & 3;
            const f16 *q = wp + ((long)tap * nks + ks) * 1024;
            bw[s % WD][0] = *reinterpret_cast<const f16x8 *>(q);
            bw[s % WD][1] = *reinterpret_cast<const f16x8 *>(q + 512);
        };
        auto load_a = [&](int s, int st) __attribute__((always_inline)) {
            const int tap = s >> 2, ks = s & 3;
            const unsigned toff = (unsigned)(((tap / KS) * HW + (tap % KS)) * PIX + 32 * ks);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                af[st][m][0] = *reinterpret_cast<const f16x8 *>(smem + abase[m] + toff);
                af[st][m][1] = *reinterpret_cast<const f16x8 *>(smem + abase[m] + toff + PLANE);
            }
        };
#pragma unroll
        for (int s = 0; s < WD - 1; ++s) load_w(s);
        load_a(0, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int st = s & 1;
            if (s + WD - 1 < NSTEP) load_w(s + WD - 1);
            if (s + 1 < NSTEP) load_a(s + 1, st ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                // weights as the A operand: the accumulator is the TRANSPOSED tile - lane = pixel, registers = output channels, four
                // consecutive channels in four consecutive registers, so the epilogue moves 16 bytes per lane
                acc[m] = mfma32(bw[s % WD][0], af[st][m][1], acc[m]);   // hi . lo
                acc[m] = mfma32(bw[s % WD][1], af[st][m][0], acc[m]);   // lo . hi
                acc[m] = mfma32(bw[s % WD][0], af[st][m][0], acc[m]);   // hi . hi
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- epilogue: un-scale, BatchNorm (inference: y = conv * s + t), residual, ReLU, NHWC store, abs-max.  Lane (j, kg) holds pixel
    // (row32(j), j & 15) of its 2 x 16 block; register r = channel (r & 3) + 8 (r >> 2) + 4 kg of the wave's 32: 16-byte accesses (with
    // pixels as the A operand a lane held ONE channel of 16 pixels: 4-byte accesses, and layer 1 took 1.07 instead of 0.89 ms; a 4 x 4
    // lane-quad transpose on top, which makes every access a whole 128-byte line, costs more in shuffles than it saves: 0.91 ms average)
    const float un = 1.0f / (s_in * *a.w_scale);
    f32x4 bs[4], bt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bs[q] = *reinterpret_cast<const f32x4 *>(a.bn_scale + co0 + 8 * q + 4 * kg) * un;
        bt[q] = *reinterpret_cast<const f32x4 *>(a.bn_shift + co0 + 8 * q + 4 * kg);
    }
    float mx = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int gy = y0 + 4 * ph + 2 * m + row32(j), gx = x0 + (j & 15);
        if (gy >= a.H || gx >= a.W) continue;
        const long at = (((long)n * a.H + gy) * a.W + gx) * a.Cout + co0 + 4 * kg;
        f32x4 rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[q] = a.res ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.res + at + 8 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = f32x4{acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]} * bs[q] + bt[q] + rv[q];
            if (a.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            *reinterpret_cast<f32x4 *>(a.y + at + 8 * q) = v;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    if (a.y_amax) publish_amax(a.y_amax, mx, lane);
}

// ---------------------------------------------------------------------------------------------------
// Data gradient of the 3 x 3 / stride-2 / padding-1 convolution (conv1 of layers 2 - 4 in training): the transposed convolution
//   dx[n][y][x][ci] = sum over (t, u) of dil[y + t - 1][x + u - 1] . wt[ci][:][t][u],   dil[2 i][2 j] = dy[i][j], zero elsewhere
// (wt = the flipped, transposed weights, packed like a forward convolution's).  Run on the dilated tensor, conv3x3_kernel multiplies three
// zeros for every value.  Here a launch takes ONE parity class (PY, PX) of the output pixels (y, x) = (2 i + PY, 2 j + PX): only the taps with
// t = PY + 1 (mod 2), u = PX + 1 (mod 2) meet a non-zero - 1, 2, 2 or 4 of the nine - and they read dy itself at (i + a, j + b), a, b in {0, 1}:
// the tile, the split-fp16 staging, the weight ring and the epilogue are conv3x3_kernel's with a (TH + PY) x (TW + PX) halo of dy, no
// padding, and an output pixel pitch of two.  Four launches = 9 tap-passes over the quarter-resolution map instead of 36.
template <int PY, int PX>
__global__ __launch_bounds__(256, 3) void convt3x3_s2_kernel(ConvArgs a, int Ho, int Wo, int wtaps) {
    constexpr int HHt = TH + PY, HWt = TW + PX, NTY = 1 + PY, NTX = 1 + PX, TAPS = NTY * NTX, NSTEP = TAPS * 4, PLANEt = HHt * HWt * PIX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kg = lane >> 5;
    const int ph = w & 1, ch = w >> 1;
    int t, yblk;
    xcd_tile(a.Cout / COT, a.ygroup, a.tiles_x * a.tiles_y * a.N, t, yblk);
    if (t >= a.tiles_x * a.tiles_y * a.N) return;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int i0 = ty * TH, j0 = tx * TW;              // the tile's first (i, j): output pixel (2 i + PY, 2 j + PX), dy pixels (i + a, j + b)
    const int co0 = yblk * COT + ch * 32;
    const float s_in = f16_scale_from_bits(*a.x_amax);
    const int nks = a.Cin / 16;
    const f16 *wbase = a.w + (long)(co0 >> 5) * wtaps * nks * 1024 + lane * 8;   // the forward layout: [tap 0 .. 8][ks][plane] (wtaps = 1: a 1 x 1 weight)
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    unsigned abase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) abase[m] = (unsigned)(((4 * ph + 2 * m + row32(j)) * HWt + (j & 15)) * PIX + 16 * kg);
    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        if (c0 > 0) __syncthreads();
        constexpr int NST = (HHt * HWt * (CK / 4) + 255) / 256;
        f32x4 hv[NST];
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int i = tid + 256 * q, c4 = i & (CK / 4 - 1), p = i / (CK / 4);
            const int hy = p / HWt, hx = p - hy * HWt;
            const int gy = i0 + hy, gx = j0 + hx;
            hv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p < HHt * HWt && gy < Ho && gx < Wo) hv[q] = *reinterpret_cast<const f32x4 *>(a.x + (((long)n * Ho + gy) * Wo + gx) * a.Cin + c0 + 4 * c4);
        }
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int i = tid + 256 * q, c4 = i & (CK / 4 - 1), p = i / (CK / 4);
            if (p >= HHt * HWt) continue;
            f16x4 h, l;
            f16_split4(hv[q], s_in, h, l);
            char *at = smem + p * PIX + 8 * c4;
            *reinterpret_cast<f16x4 *>(at) = h;
            *reinterpret_cast<f16x4 *>(at + PLANEt) = l;
        }
        __syncthreads();
        const f16 *wp = wbase + (long)(c0 / 16) * 1024;
        f16x8 bw[WD][2], af[2][2][2];
        // step s = (tap index q = s >> 2, k-step s & 3); tap q = (qy, qx): weight tap (t, u) = (PY ? 2 qy : 1, PX ? 2 qx : 1), dy offset (a, b) = (qy, qx)
        auto load_w = [&](int s) __attribute__((always_inline)) {
            const int q = s >> 2, ks = s & 3, qy = q / NTX, qx = q % NTX;
            const int tap = wtaps == 1 ? 0 : (PY ? 2 * qy : 1) * 3 + (PX ? 2 * qx : 1);
            const f16 *pq = wp + ((long)tap * nks + ks) * 1024;
            bw[s % WD][0] = *reinterpret_cast<const f16x8 *>(pq);
            bw[s % WD][1] = *reinterpret_cast<const f16x8 *>(pq + 512);
        };
        auto load_a = [&](int s, int st) __attribute__((always_inline)) {
            const int q = s >> 2, ks = s & 3, qy = q / NTX, qx = q % NTX;
            const unsigned toff = (unsigned)((qy * HWt + qx) * PIX + 32 * ks);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                af[st][m][0] = *reinterpret_cast<const f16x8 *>(smem + abase[m] + toff);
                af[st][m][1] = *reinterpret_cast<const f16x8 *>(smem + abase[m] + toff + PLANEt);
            }
        };
#pragma unroll
        for (int s = 0; s < WD - 1 && s < NSTEP; ++s) load_w(s);
        load_a(0, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int st = s & 1;
            if (s + WD - 1 < NSTEP) load_w(s + WD - 1);
            if (s + 1 < NSTEP) load_a(s + 1, st ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m] = mfma32(bw[s % WD][0], af[st][m][1], acc[m]);
                acc[m] = mfma32(bw[s % WD][1], af[st][m][0], acc[m]);
                acc[m] = mfma32(bw[s % WD][0], af[st][m][0], acc[m]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float un = 1.0f / (s_in * *a.w_scale);
    f32x4 bs[4], bt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bs[q] = *reinterpret_cast<const f32x4 *>(a.bn_scale + co0 + 8 * q + 4 * kg) * un;
        bt[q] = *reinterpret_cast<const f32x4 *>(a.bn_shift + co0 + 8 * q + 4 * kg);
    }
    float mx = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int gy = 2 * (i0 + 4 * ph + 2 * m + row32(j)) + PY, gx = 2 * (j0 + (j & 15)) + PX;
        if (gy >= a.H || gx >= a.W) continue;
        const long at = (((long)n * a.H + gy) * a.W + gx) * a.Cout + co0 + 4 * kg;
        f32x4 rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[q] = a.res ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.res + at + 8 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = f32x4{acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]} * bs[q] + bt[q] + rv[q];
            *reinterpret_cast<f32x4 *>(a.y + at + 8 * q) = v;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    if (a.y_amax) publish_amax(a.y_amax, mx, lane);
}

// ---------------------------------------------------------------------------------------------------
// The striding convolutions of a ResNet stage entry: 3 x 3 stride 2 padding 1 (conv1 of layers 2 - 4) and the 1 x 1 stride 2 shortcut,
// with the inference BatchNorm (+ ReLU) in the epilogue.  Same arithmetic and operand layouts as conv3x3_kernel; the output maps are
// small (60 x 80 ... 15 x 20), so a workgroup takes a 4 x 8 pixel tile (one 32-row MFMA tile) x 128 output channels, wave w = channel
// tile w; the 9 x 17 halo (KS = 3) or just the 4 x 8 sampled pixels (KS = 1) of a 64-channel chunk are staged as fp16 hi | lo planes.
// ---------------------------------------------------------------------------------------------------
constexpr int S2_TH = 4, S2_TW = 8, S2_COT = 128;
// KS = 3: the 9 x 17 halo is stored de-interleaved by column parity (a tap reads columns 2 ox + dx: one parity, consecutive indices),
// sub-plane [parity][row][9 pixels of 144 bytes, row pitch 84 slots].  With the lane map s2_pixel a 16-lane ds_read_b128 group covers
// two output rows x 8 columns: slots 9 ox + {0, 8} (the rows are 2 x 84 = 8 (mod 16) slots apart): 16 distinct.  (Reading columns
// 2 ox of ONE plane puts every lane on an even slot: 60 % of the LDS cycles were conflicts.)
template <int KS>
struct S2Cfg {
    static constexpr int HH = KS == 3 ? (S2_TH - 1) * 2 + 3 : S2_TH, HW = KS == 3 ? (S2_TW - 1) * 2 + 3 : S2_TW;
    static constexpr int RP = KS == 3 ? 84 * 16 : S2_TW * PIX;          // bytes of a sub-plane row
    static constexpr int SUB = HH * RP;                                  // bytes of a parity sub-plane (KS = 1: the plane)
    static constexpr int PLANE = KS == 3 ? 2 * SUB : SUB, LDS = 2 * PLANE;
    // byte offset of halo pixel (hy, hx) in a plane
    static __device__ __forceinline__ int at(int hy, int hx) { return KS == 3 ? (hx & 1) * SUB + hy * RP + (hx >> 1) * PIX : (hy * S2_TW + hx) * PIX; }
};
// MFMA row i = 0 .. 31 -> output pixel (oy, ox) of the 4 x 8 tile.  KS = 3: rows {0, 1} on lanes {0-3, 12-15, 20-27}, rows {2, 3} on
// {4-11, 16-19, 28-31} - the lane groups of ds_read_b128; KS = 1: row-major.
template <int KS>
__device__ __forceinline__ void s2_pixel(int i, int &oy, int &ox) {
    if (KS == 3) {
        const int q = i >> 2;
        oy = (0xD728 >> (2 * q)) & 3;                    // q = 0 .. 7 -> 0 2 2 0 3 1 1 3
        ox = i - 4 * ((065542110 >> (3 * q)) & 7);      // ... minus 0 4 4 8 16 20 20 24
    } else {
        oy = i >> 3;
        ox = i & 7;
    }
}

template <int KS>
__global__ __launch_bounds__(256, 3) void conv_s2_kernel(ConvArgs a) {
    using C = S2Cfg<KS>;
    constexpr int TAPS = KS * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kg = lane >> 5;
    const int Ho = (a.H + 1) / 2, Wo = (a.W + 1) / 2;   // (H + 2 pad - KS) / 2 + 1 for KS = 3 / pad 1 and KS = 1 / pad 0
    int t, yblk;
    xcd_tile(a.Cout / S2_COT, a.ygroup, a.tiles_x * a.tiles_y * a.N, t, yblk);
    if (t >= a.tiles_x * a.tiles_y * a.N) return;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int y0 = ty * S2_TH, x0 = tx * S2_TW;        // output tile origin
    const int co0 = yblk * S2_COT + w * 32;
    const float s_in = f16_scale_from_bits(*a.x_amax);
    const int nks = a.Cin / 16;
    const f16 *wbase = a.w + (long)(co0 >> 5) * TAPS * nks * 1024 + lane * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // this lane's A row: output pixel s2_pixel(j) of the tile; its halo offset for tap (0, 0)
    int oyj, oxj;
    s2_pixel<KS>(j, oyj, oxj);
    const unsigned abase = (unsigned)((KS == 3 ? C::at(2 * oyj, 2 * oxj) : C::at(oyj, oxj)) + 16 * kg);
    for (int c0 = 0; c0 < a.Cin; c0 += CK) {
        if (c0 > 0) __syncthreads();
        constexpr int NST = (C::HH * C::HW * (CK / 4) + 255) / 256;   // every load of the chunk in flight before the first LDS store
        f32x4 hv[NST];
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int i = tid + 256 * q, c4 = i & (CK / 4 - 1), p = i / (CK / 4);
            const int hy = p / C::HW, hx = p - hy * C::HW;
            const int gy = KS == 3 ? 2 * y0 + hy - 1 : 2 * (y0 + hy), gx = KS == 3 ? 2 * x0 + hx - 1 : 2 * (x0 + hx);
            hv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p < C::HH * C::HW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                hv[q] = *reinterpret_cast<const f32x4 *>(a.x + (((long)n * a.H + gy) * a.W + gx) * a.Cin + c0 + 4 * c4);
        }
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int i = tid + 256 * q, c4 = i & (CK / 4 - 1), p = i / (CK / 4);
            if (p >= C::HH * C::HW) continue;
            f16x4 h, l;
            f16_split4(hv[q], s_in, h, l);
            char *at = smem + C::at(p / C::HW, p % C::HW) + 8 * c4;
            *reinterpret_cast<f16x4 *>(at) = h;
            *reinterpret_cast<f16x4 *>(at + C::PLANE) = l;
        }
        __syncthreads();
        const f16 *wp = wbase + (long)(c0 / 16) * 1024;
        constexpr int SD = KS == 3 ? 2 * WD : 4;   // a step is 3 MFMAs here: twice the lookahead (KS = 1: all 4 steps at once)
        f16x8 bw[SD][2], af[2][2];
        auto load_w = [&](int s) __attribute__((always_inline)) {
            const int tap = s >> 2, ks = s & 3;
            const f16 *q = wp + ((long)tap * nks + ks) * 1024;
            bw[s % SD][0] = *reinterpret_cast<const f16x8 *>(q);
            bw[s % SD][1] = *reinterpret_cast<const f16x8 *>(q + 512);
        };
        auto load_a = [&](int s, int st) __attribute__((always_inline)) {
            const int tap = s >> 2, ks = s & 3;
            // tap (dy, dx): halo pixel (2 oy + dy, 2 ox + dx) = parity dx & 1, row + dy, index + (dx >> 1)
            const unsigned toff = (unsigned)((KS == 3 ? ((tap % 3) & 1) * C::SUB + (tap / 3) * C::RP + ((tap % 3) >> 1) * PIX : 0) + 32 * ks);
            af[st][0] = *reinterpret_cast<const f16x8 *>(smem + abase + toff);
            af[st][1] = *reinterpret_cast<const f16x8 *>(smem + abase + toff + C::PLANE);
        };
#pragma unroll
        for (int s = 0; s < SD - 1 && s < TAPS * 4; ++s) load_w(s);
        load_a(0, 0);
#pragma unroll
        for (int s = 0; s < TAPS * 4; ++s) {
            const int st = s & 1;
            if (s + SD - 1 < TAPS * 4) load_w(s + SD - 1);
            if (s + 1 < TAPS * 4) load_a(s + 1, st ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma32(bw[s % SD][0], af[st][1], acc);   // weights as A: the transposed tile (lane = pixel), 16-byte epilogue accesses
            acc = mfma32(bw[s % SD][1], af[st][0], acc);
            acc = mfma32(bw[s % SD][0], af[st][0], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float un = 1.0f / (s_in * *a.w_scale);
    float mx = 0.f;
    {   // lane (j, kg): output pixel s2_pixel(j); register r = channel (r & 3) + 8 (r >> 2) + 4 kg of the wave's 32
        const int gy = y0 + oyj, gx = x0 + oxj;
        if (gy < Ho && gx < Wo) {
            const long at = (((long)n * Ho + gy) * Wo + gx) * a.Cout + co0 + 4 * kg;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bs = *reinterpret_cast<const f32x4 *>(a.bn_scale + co0 + 8 * q + 4 * kg) * un;
                const f32x4 bt = *reinterpret_cast<const f32x4 *>(a.bn_shift + co0 + 8 * q + 4 * kg);
                f32x4 v = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]} * bs + bt;
                if (a.relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                *reinterpret_cast<f32x4 *>(a.y + at + 8 * q) = v;
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            }
        }
    }
    if (a.y_amax) publish_amax(a.y_amax, mx, lane);
}

// ---------------------------------------------------------------------------------------------------
// The ResNet stem in one launch: conv 7 x 7 stride 2 padding 3 (3 -> 64 channels) + inference BatchNorm + ReLU + max-pool 3 x 3 stride 2
// padding 1, NCHW frames in, NHWC map out (+ its abs-max word) - four library launches and a 3.1-GB intermediate map at 160 frames of
// 480 x 640 otherwise.  A workgroup (5 waves) owns a 4 x 8 tile of POOLED pixels: the 9 x 17 convolution outputs under it come from a
// 23 x 39 x 3 input tile, which is staged channel-interleaved as fp16 hi | lo planes.  No im2col: for a kernel row ky the 7 x 3 values
// (kx, ci) of an output pixel are 21 CONSECUTIVE halfs of an input row (from half 6 cx on), so K is ordered (ky, kx, ci) with each
// kernel row padded to 24 = three 8-half lane chunks (21 chunks + 1 of zero weights = 11 k-steps of 16) and an A fragment is one
// ds_read_b128 of the input row.  Its byte offset 12 cx + 16 q is 16-byte aligned only for cx = 0 mod 4, hence FOUR copies of the tile,
// copy s = cx mod 4 shifted by 4 s bytes; the copies start 12 sixteen-byte slots apart (mod 16) so that 16 consecutive cx read 16
// distinct slots.  (The first version wrote explicit im2col rows: 4.3 ms at 160 frames, its fill loop 3/4 of that.)
// Wave w = convolution pixels 32 w .. 32 w + 31 (row-major in the 9 x 17 tile; 153 of 160 rows real) x all 64 channels.  The results
// (BatchNorm, ReLU) pass through LDS (over the dead input planes) for the pooling.  50 KB of LDS: three workgroups per CU.
// ---------------------------------------------------------------------------------------------------
constexpr int ST_PH = 4, ST_PW = 8, ST_CH = 2 * ST_PH + 1, ST_CW = 2 * ST_PW + 1, ST_IH = 2 * ST_CH + 5, ST_IW = 2 * ST_CW + 5;
constexpr int ST_NPX = ST_CH * ST_CW;           // 153 convolution pixels per tile
constexpr int ST_WAVES = 5, ST_THREADS = 64 * ST_WAVES;
constexpr int ST_NKS = 11;                      // k-steps of 16: 7 kernel rows x 24 halfs = 21 chunks of 8 (+ 1 of zero weights)
constexpr int ST_ROWH = 3 * ST_IW;              // 117 halfs of an input row
constexpr int ST_RP = 272;                      // bytes of a copy's row: 12 (shift) + 234, and 17 slots (odd: kernel rows spread over the banks)
constexpr int ST_RD = ST_RP / 4;                // ... in dwords
constexpr int ST_CS = ((ST_IH * ST_RP / 16 + 15) / 16 * 16 + 12) * 16;   // copy stride: = 12 slots mod 16, >= the copy
constexpr int ST_PLANE = 4 * ST_CS;
constexpr int ST_CPITCH = 68;                   // floats per convolution pixel in the pooling buffer (64 + 4: bank spread)
constexpr int ST_LDS = 2 * ST_PLANE;
static_assert(ST_CS >= ST_IH * ST_RP && ST_NPX * ST_CPITCH * 4 <= ST_LDS, "copies fit their stride; the pooling buffer reuses the planes");
static_assert(12 + 12 * (ST_CW - 1) + 32 + 16 <= ST_RP && ST_NPX <= 32 * ST_WAVES, "fragment reads stay inside a row");

// W (64, 3, 7, 7) -> [co tile 2][k-step 11][plane][lane][8]: lane = 32 kg + j holds, for chunk c = 2 ks + kg (kernel row ky = c / 3,
// third q = c % 3), W[32 ct + j][ci][ky][kx] at e = 0..7 with 8 q + e = 3 kx + ci < 21; zero elsewhere (and for c = 21)
__global__ void stem_pack_kernel(const float *__restrict__ W, const unsigned *maxbits, f16 *__restrict__ dst, float *scale_out) {
    const float scale = f16_scale_from_bits(*maxbits);
    if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 * 2 * ST_NKS; i += gridDim.x * blockDim.x) {
        const int co = i / (2 * ST_NKS), c = i % (2 * ST_NKS);
        const int ky = c / 3, q = c - 3 * ky;
        f16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = 8 * q + e, kx = kk / 3, ci = kk - 3 * kx;
            const float v = (c < 21 && kk < 21) ? W[((co * 3 + ci) * 7 + ky) * 7 + kx] * scale : 0.f;
            hh[e] = (f16)v;
            ll[e] = (f16)(v - (float)hh[e]);
        }
        const int ct = co >> 5, j = co & 31, ks = c >> 1, kg = c & 1;
        f16 *o = dst + (((long)ct * ST_NKS + ks) * 2) * 512 + (32 * kg + j) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = hh[e];
            o[512 + e] = ll[e];
        }
    }
}

struct StemArgs {
    const float *x;           // [N][3][H][W]
    const f16 *w;
    const float *w_scale;
    const unsigned *x_amax;
    const float *bn_scale, *bn_shift;
    float *y;                 // [N][Hp][Wp][64]
    unsigned *y_amax;
    int N, H, W, Hc, Wc, Hp, Wp, tiles_x, tiles_y;
    float *y_raw;             // training: [N][Hc][Wc][64] receives the bare convolution (no BatchNorm, ReLU, pool; y unused), or NULL
};

__global__ __launch_bounds__(ST_THREADS) void stem_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kg = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int py0 = ty * ST_PH, px0 = tx * ST_PW;          // pooled tile origin
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;        // convolution tile origin (the pool's padding row / column)
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;        // input tile origin
    const float s_in = f16_scale_from_bits(*a.x_amax);
    // ---- input tile -> four shifted copies of its hi | lo planes.  An item is a dword of a copy's row: the half pair (2 p, 2 p + 1) of
    // the row (zeros outside the frame, left of the row (p < 0: the copies' shifts) and right of it) goes to dword p + s of copy s
    constexpr int ST_PAIRS = ST_RD + 3;                     // p = -3 .. ST_RD - 1
    constexpr int ST_ITEMS = ST_IH * ST_PAIRS, ST_NIT = (ST_ITEMS + ST_THREADS - 1) / ST_THREADS;
    float v0[ST_NIT], v1[ST_NIT];
#pragma unroll
    for (int q = 0; q < ST_NIT; ++q) {
        const int i = tid + ST_THREADS * q, iy = i / ST_PAIRS, p = i - iy * ST_PAIRS - 3;
        const int gy = iy0 + iy;
        v0[q] = v1[q] = 0.f;
        if (i < ST_ITEMS && p >= 0 && gy >= 0 && gy < a.H) {
            const int e0 = 2 * p, e1 = 2 * p + 1;
            const int x0 = e0 / 3, c0 = e0 - 3 * x0, x1 = e1 / 3, c1 = e1 - 3 * x1;
            const float *row = a.x + ((long)n * 3 * a.H + gy) * a.W;
            const long cstride = (long)a.H * a.W;
            if (e0 < ST_ROWH && ix0 + x0 >= 0 && ix0 + x0 < a.W) v0[q] = row[c0 * cstride + ix0 + x0];
            if (e1 < ST_ROWH && ix0 + x1 >= 0 && ix0 + x1 < a.W) v1[q] = row[c1 * cstride + ix0 + x1];
        }
    }
#pragma unroll
    for (int q = 0; q < ST_NIT; ++q) {
        const int i = tid + ST_THREADS * q, iy = i / ST_PAIRS, p = i - iy * ST_PAIRS - 3;
        if (i >= ST_ITEMS) continue;
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        const float s0 = v0[q] * s_in, s1 = v1[q] * s_in;
        const f16x2 h = {(f16)s0, (f16)s1};
        const f16x2 l = {(f16)(s0 - (float)h[0]), (f16)(s1 - (float)h[1])};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int d = p + s;
            if (d < 0 || d >= ST_RD) continue;
            char *at = smem + s * ST_CS + iy * ST_RP + 4 * d;
            *reinterpret_cast<f16x2 *>(at) = h;
            *reinterpret_cast<f16x2 *>(at + ST_PLANE) = l;
        }
    }
    __syncthreads();
    // ---- this wave's 32 convolution pixels x 64 channels: 11 k-steps x (2 channel tiles x 3 products)
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    {
        const int r0 = min(32 * w + j, ST_NPX - 1);          // rows past the tile repeat its last pixel; they are never stored
        const int cy = r0 / ST_CW, cx = r0 - cy * ST_CW;
        const char *arow = smem + (cx & 3) * ST_CS + 2 * cy * ST_RP + 4 * (cx & 3) + 12 * cx;
        const f16 *wp = a.w + lane * 8;
#pragma unroll
        for (int ks = 0; ks < ST_NKS; ++ks) {
            // chunk c = 2 ks + kg: kernel row c / 3, third c % 3; c = 21 has zero weights: it reads chunk 20 again
            const int ca = 2 * ks, cb = 2 * ks + 1 < 21 ? 2 * ks + 1 : 20;
            const int off = kg ? (cb / 3) * ST_RP + 16 * (cb % 3) : (ca / 3) * ST_RP + 16 * (ca % 3);
            const f16x8 ah = *reinterpret_cast<const f16x8 *>(arow + off), al = *reinterpret_cast<const f16x8 *>(arow + off + ST_PLANE);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f16 *q = wp + ((long)(m * ST_NKS + ks) * 2) * 512;
                const f16x8 bh = *reinterpret_cast<const f16x8 *>(q), bl = *reinterpret_cast<const f16x8 *>(q + 512);
                acc[m] = mfma32(al, bh, acc[m]);
                acc[m] = mfma32(ah, bl, acc[m]);
                acc[m] = mfma32(ah, bh, acc[m]);
            }
        }
    }
    const float un = 1.0f / (s_in * *a.w_scale);
    if (a.y_raw) {
        // training: the bare convolution output.  Neighbouring tiles share a row / column of convolution pixels (the pool's halo): a tile
        // stores rows / columns 1 .. of its 9 x 17 pixels
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * kg;
                if (row >= ST_NPX) continue;
                const int cy = row / ST_CW, cx = row - cy * ST_CW;
                const int gy = cy0 + cy, gx = cx0 + cx;
                if (cy >= 1 && cx >= 1 && gy < a.Hc && gx < a.Wc) a.y_raw[(((long)n * a.Hc + gy) * a.Wc + gx) * 64 + 32 * m + j] = acc[m][r] * un;
            }
        return;
    }
    __syncthreads();   // the input planes are consumed: the region becomes the pooling buffer [153 px][ST_CPITCH]
    float *cbuf = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int co = 32 * m + j;
        const float bs = a.bn_scale[co] * un, bt = a.bn_shift[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (row >= ST_NPX) continue;
            const int cy = row / ST_CW, cx = row - cy * ST_CW;
            const int gy = cy0 + cy, gx = cx0 + cx;
            // outside the convolution map = the pool's padding: every window holds its valid centre and ReLU outputs are >= 0, so 0 acts as -inf
            const bool ok = gy >= 0 && gy < a.Hc && gx >= 0 && gx < a.Wc;
            cbuf[row * ST_CPITCH + co] = ok ? fmaxf(acc[m][r] * bs + bt, 0.f) : 0.f;
        }
    }
    __syncthreads();
    float mx = 0.f;
    for (int i = tid; i < ST_PH * ST_PW * 64; i += ST_THREADS) {
        const int co = i & 63, pp = i >> 6, ppy = pp / ST_PW, ppx = pp - ppy * ST_PW;
        const int gy = py0 + ppy, gx = px0 + ppx;
        if (gy >= a.Hp || gx >= a.Wp) continue;
        float v = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v = fmaxf(v, cbuf[((2 * ppy + dy) * ST_CW + 2 * ppx + dx) * ST_CPITCH + co]);
        a.y[(((long)n * a.Hp + gy) * a.Wp + gx) * 64 + co] = v;
        mx = fmaxf(mx, v);
    }
    if (a.y_amax) publish_amax(a.y_amax, mx, lane);
}

}   // namespace cv

extern "C" int sd_absmax_word(const float *x, int64_t n, uint32_t *word, void *stream) {
    if (!x || !word || n <= 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return fail(SD_E_BADARG, "sd_absmax_word: x must be 16-byte aligned, n > 0");
    long blocks = (n / 4 + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    SD_LAUNCH(cv::absmax_word_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)(n / 4), (long)n, word);
    SD_CHECK_LAUNCH("absmax_word_kernel");
    return 0;
}

extern "C" size_t sd_conv3x3_packed_halfs(int Cout, int Cin) { return (size_t)Cout * Cin * 9 * 2; }
extern "C" size_t sd_conv_packed_halfs(int Cout, int Cin, int ksize) { return (size_t)Cout * Cin * ksize * ksize * 2; }

static int conv_pack(const float *w, int Cout, int Cin, int ksize, void *planes, float *scale, uint32_t *amax_word, void *stream);
extern "C" int sd_conv3x3_pack(const float *w, int Cout, int Cin, void *planes, float *scale, uint32_t *amax_word, void *stream) {
    return conv_pack(w, Cout, Cin, 3, planes, scale, amax_word, stream);
}
extern "C" int sd_conv_pack(const float *w, int Cout, int Cin, int ksize, void *planes, float *scale, uint32_t *amax_word, void *stream) {
    if (ksize != 1 && ksize != 3) return fail(SD_E_BADARG, "sd_conv_pack: kernel size 1 or 3");
    return conv_pack(w, Cout, Cin, ksize, planes, scale, amax_word, stream);
}
static int conv_pack(const float *w, int Cout, int Cin, int ksize, void *planes, float *scale, uint32_t *amax_word, void *stream) {
    if (!w || !planes || !scale || !amax_word || Cout <= 0 || Cin <= 0 || Cout % 64 || Cin % 64)
        return fail(SD_E_BADARG, "sd_conv3x3_pack: channels must be positive multiples of 64");
    hipStream_t st = (hipStream_t)stream;
    SD_LAUNCH(cv::zero_words_kernel, dim3(1), dim3(64), 0, st, amax_word, 1);
    SD_CHECK_LAUNCH("zero_words_kernel");
    const int taps = ksize * ksize;
    const long n = (long)Cout * Cin * taps;
    int rc = sd_absmax_word(w, n, amax_word, stream);
    if (rc) return rc;
    long blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    SD_LAUNCH(cv::conv3x3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, Cout, Cin, taps, amax_word, (f16 *)planes, scale);
    SD_CHECK_LAUNCH("conv3x3_pack_kernel");
    return 0;
}

static int conv_s1_bn_act(int ksize, const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                          const float *bn_shift, const float *res, float *y, uint32_t *y_amax, int N, int H, int W, int Cin, int Cout, int relu,
                          void *stream);
extern "C" int sd_conv3x3_bn_act(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                                 const float *bn_shift, const float *res, float *y, uint32_t *y_amax, int N, int H, int W, int Cin, int Cout,
                                 int relu, void *stream) {
    return conv_s1_bn_act(3, x, w_planes, w_scale, x_amax, bn_scale, bn_shift, res, y, y_amax, N, H, W, Cin, Cout, relu, stream);
}
extern "C" int sd_conv1x1_bn_act(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                                 const float *bn_shift, const float *res, float *y, uint32_t *y_amax, int N, int H, int W, int Cin, int Cout,
                                 int relu, void *stream) {
    return conv_s1_bn_act(1, x, w_planes, w_scale, x_amax, bn_scale, bn_shift, res, y, y_amax, N, H, W, Cin, Cout, relu, stream);
}
static int conv_s1_bn_act(int ksize, const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                          const float *bn_shift, const float *res, float *y, uint32_t *y_amax, int N, int H, int W, int Cin, int Cout, int relu,
                          void *stream) {
    if (!x || !w_planes || !w_scale || !x_amax || !bn_scale || !bn_shift || !y || N <= 0 || H <= 0 || W <= 0)
        return fail(SD_E_BADARG, "sd_conv3x3_bn_act: null pointer or empty shape");
    if (Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return fail(SD_E_BADDIM, "sd_conv3x3_bn_act: channels must be positive multiples of 64");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || x == y) return fail(SD_E_BADARG, "sd_conv3x3_bn_act: x must be 16-byte aligned and distinct from y");
    // the epilogue moves 16 bytes per lane: four consecutive output channels of y / res / the BatchNorm vectors
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(bn_scale) | reinterpret_cast<uintptr_t>(bn_shift)) & 15)
        return fail(SD_E_BADARG, "sd_conv3x3_bn_act / sd_conv1x1_bn_act: y, res, bn_scale and bn_shift must be 16-byte aligned");
    cv::ConvArgs a{x, (const f16 *)w_planes, w_scale, x_amax, bn_scale, bn_shift, res, y, y_amax, N, H, W, Cin, Cout, relu,
                   (W + cv::TW - 1) / cv::TW, (H + cv::TH - 1) / cv::TH, cv::y_group(Cout / cv::COT, (long)cv::COT * Cin * ksize * ksize * 4)};
    const long tiles = (long)a.tiles_x * a.tiles_y * N;
    const long wgs = (tiles + 7) / 8 * 8 * (Cout / cv::COT);   // cv::xcd_tile's order
    if (wgs > 0x7fffffffL) return fail(SD_E_TOOBIG, "sd_conv3x3_bn_act: too many tiles");
    if (ksize == 3) SD_LAUNCH(cv::conv3x3_kernel<3>, dim3((unsigned)wgs), dim3(256), (size_t)cv::LDS_BYTES, (hipStream_t)stream, a);
    else SD_LAUNCH(cv::conv3x3_kernel<1>, dim3((unsigned)wgs), dim3(256), (size_t)(2 * cv::TH * cv::TW * cv::PIX), (hipStream_t)stream, a);
    SD_CHECK_LAUNCH("conv3x3_kernel");
    return 0;
}

template <int PY, int PX>
static int convt_class(cv::ConvArgs a, int Ho, int Wo, int wtaps, hipStream_t st) {
    const int ni = (a.H - PY + 1) / 2, nj = (a.W - PX + 1) / 2;   // output pixels of this parity class per column / row
    if (ni <= 0 || nj <= 0) return 0;
    a.tiles_x = (nj + cv::TW - 1) / cv::TW;
    a.tiles_y = (ni + cv::TH - 1) / cv::TH;
    const long tiles = (long)a.tiles_x * a.tiles_y * a.N;
    const long wgs = (tiles + 7) / 8 * 8 * (a.Cout / cv::COT);
    if (wgs > 0x7fffffffL) return fail(SD_E_TOOBIG, "sd_convt3x3_s2: too many tiles");
    SD_LAUNCH((cv::convt3x3_s2_kernel<PY, PX>), dim3((unsigned)wgs), dim3(256), (size_t)(2 * (cv::TH + PY) * (cv::TW + PX) * cv::PIX), st, a, Ho, Wo, wtaps);
    SD_CHECK_LAUNCH("convt3x3_s2_kernel");
    return 0;
}
extern "C" int sd_convt3x3_s2(const float *dy, const void *w_planes, const float *w_scale, const uint32_t *dy_amax, const float *bn_scale,
                              const float *bn_shift, const float *res, float *dx, uint32_t *dx_amax, int N, int H, int W, int Cin, int Cout,
                              void *stream) {
    if (!dy || !w_planes || !w_scale || !dy_amax || !bn_scale || !bn_shift || !dx || N <= 0 || H <= 0 || W <= 0)
        return fail(SD_E_BADARG, "sd_convt3x3_s2: null pointer or empty shape");
    if (Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return fail(SD_E_BADDIM, "sd_convt3x3_s2: channels must be positive multiples of 64");
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(bn_scale) |
         reinterpret_cast<uintptr_t>(bn_shift)) & 15)
        return fail(SD_E_BADARG, "sd_convt3x3_s2: tensors must be 16-byte aligned");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    cv::ConvArgs a{dy, (const f16 *)w_planes, w_scale, dy_amax, bn_scale, bn_shift, res, dx, dx_amax, N, H, W, Cin, Cout, 0, 0, 0,
                   cv::y_group(Cout / cv::COT, (long)cv::COT * Cin * 9 * 4)};
    hipStream_t st = (hipStream_t)stream;
    if (int rc = convt_class<1, 1>(a, Ho, Wo, 9, st)) return rc;   // (the four-tap class first: the longest workgroups)
    if (int rc = convt_class<1, 0>(a, Ho, Wo, 9, st)) return rc;
    if (int rc = convt_class<0, 1>(a, Ho, Wo, 9, st)) return rc;
    return convt_class<0, 0>(a, Ho, Wo, 9, st);
}
// the 1 x 1 / stride-2 shortcut's data gradient: dx[2 i][2 j] = dy[i][j] . wt, every other pixel of dx stays as the caller left it (ZEROED)
extern "C" int sd_convt1x1_s2(const float *dy, const void *w_planes, const float *w_scale, const uint32_t *dy_amax, const float *bn_scale,
                              const float *bn_shift, float *dx, int N, int H, int W, int Cin, int Cout, void *stream) {
    if (!dy || !w_planes || !w_scale || !dy_amax || !bn_scale || !bn_shift || !dx || N <= 0 || H <= 0 || W <= 0)
        return fail(SD_E_BADARG, "sd_convt1x1_s2: null pointer or empty shape");
    if (Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return fail(SD_E_BADDIM, "sd_convt1x1_s2: channels must be positive multiples of 64");
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(bn_scale) | reinterpret_cast<uintptr_t>(bn_shift)) & 15)
        return fail(SD_E_BADARG, "sd_convt1x1_s2: tensors must be 16-byte aligned");
    cv::ConvArgs a{dy, (const f16 *)w_planes, w_scale, dy_amax, bn_scale, bn_shift, nullptr, dx, nullptr, N, H, W, Cin, Cout, 0, 0, 0,
                   cv::y_group(Cout / cv::COT, (long)cv::COT * Cin * 4)};
    return convt_class<0, 0>(a, (H + 1) / 2, (W + 1) / 2, 1, (hipStream_t)stream);
}

extern "C" int sd_conv_s2_bn_act(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                                 const float *bn_shift, float *y, uint32_t *y_amax, int N, int H, int W, int Cin, int Cout, int ksize, int relu,
                                 void *stream) {
    if (!x || !w_planes || !w_scale || !x_amax || !bn_scale || !bn_shift || !y || N <= 0 || H <= 0 || W <= 0)
        return fail(SD_E_BADARG, "sd_conv_s2_bn_act: null pointer or empty shape");
    if (ksize != 1 && ksize != 3) return fail(SD_E_BADARG, "sd_conv_s2_bn_act: kernel size 1 or 3");
    if (Cin <= 0 || Cin % 64 || Cout <= 0 || Cout % cv::S2_COT) return fail(SD_E_BADDIM, "sd_conv_s2_bn_act: Cin a multiple of 64, Cout a multiple of 128");
    if ((reinterpret_cast<uintptr_t>(x) & 15) || x == y) return fail(SD_E_BADARG, "sd_conv_s2_bn_act: x must be 16-byte aligned and distinct from y");
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bn_scale) | reinterpret_cast<uintptr_t>(bn_shift)) & 15)
        return fail(SD_E_BADARG, "sd_conv_s2_bn_act: y, bn_scale and bn_shift must be 16-byte aligned");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    cv::ConvArgs a{x, (const f16 *)w_planes, w_scale, x_amax, bn_scale, bn_shift, nullptr, y, y_amax, N, H, W, Cin, Cout, relu,
                   (Wo + cv::S2_TW - 1) / cv::S2_TW, (Ho + cv::S2_TH - 1) / cv::S2_TH,
                   cv::y_group(Cout / cv::S2_COT, (long)cv::S2_COT * Cin * ksize * ksize * 4)};
    const long tiles = (long)a.tiles_x * a.tiles_y * N;
    const long wgs = (tiles + 7) / 8 * 8 * (Cout / cv::S2_COT);   // cv::xcd_tile's order
    if (wgs > 0x7fffffffL) return fail(SD_E_TOOBIG, "sd_conv_s2_bn_act: too many tiles");
    const dim3 grid((unsigned)wgs);
    if (ksize == 3) SD_LAUNCH(cv::conv_s2_kernel<3>, grid, dim3(256), (size_t)cv::S2Cfg<3>::LDS, (hipStream_t)stream, a);
    else SD_LAUNCH(cv::conv_s2_kernel<1>, grid, dim3(256), (size_t)cv::S2Cfg<1>::LDS, (hipStream_t)stream, a);
    SD_CHECK_LAUNCH("conv_s2_kernel");
    return 0;
}

extern "C" size_t sd_stem_packed_halfs(void) { return (size_t)64 * cv::ST_NKS * 16 * 2; }

extern "C" int sd_stem_pack(const float *w, void *planes, float *scale, uint32_t *amax_word, void *stream) {
    if (!w || !planes || !scale || !amax_word) return fail(SD_E_BADARG, "sd_stem_pack: null pointer");
    hipStream_t st = (hipStream_t)stream;
    SD_LAUNCH(cv::zero_words_kernel, dim3(1), dim3(64), 0, st, amax_word, 1);
    SD_CHECK_LAUNCH("zero_words_kernel");
    int rc = sd_absmax_word(w, 64 * 3 * 49, amax_word, stream);
    if (rc) return rc;
    SD_LAUNCH(cv::stem_pack_kernel, dim3(6), dim3(256), 0, st, w, amax_word, (f16 *)planes, scale);
    SD_CHECK_LAUNCH("stem_pack_kernel");
    return 0;
}

static int stem_launch(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                       const float *bn_shift, float *y, uint32_t *y_amax, float *y_raw, int N, int H, int W, void *stream);
extern "C" int sd_stem_conv_bn_relu_pool(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                                         const float *bn_shift, float *y, uint32_t *y_amax, int N, int H, int W, void *stream) {
    if (!bn_scale || !bn_shift || !y) return fail(SD_E_BADARG, "sd_stem_conv_bn_relu_pool: null pointer or empty shape");
    return stem_launch(x, w_planes, w_scale, x_amax, bn_scale, bn_shift, y, y_amax, nullptr, N, H, W, stream);
}
extern "C" int sd_stem_conv_raw(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, float *y_raw, int N, int H, int W,
                                void *stream) {
    if (!y_raw) return fail(SD_E_BADARG, "sd_stem_conv_raw: null pointer");
    return stem_launch(x, w_planes, w_scale, x_amax, nullptr, nullptr, nullptr, nullptr, y_raw, N, H, W, stream);
}
static int stem_launch(const float *x, const void *w_planes, const float *w_scale, const uint32_t *x_amax, const float *bn_scale,
                       const float *bn_shift, float *y, uint32_t *y_amax, float *y_raw, int N, int H, int W, void *stream) {
    if (!x || !w_planes || !w_scale || !x_amax || N <= 0 || H <= 0 || W <= 0)
        return fail(SD_E_BADARG, "sd_stem_conv_bn_relu_pool: null pointer or empty shape");
    cv::StemArgs a;
    a.x = x; a.w = (const f16 *)w_planes; a.w_scale = w_scale; a.x_amax = x_amax; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.y = y; a.y_amax = y_amax;
    a.y_raw = y_raw;
    a.N = N; a.H = H; a.W = W;
    a.Hc = (H - 1) / 2 + 1; a.Wc = (W - 1) / 2 + 1;          // conv 7 x 7, stride 2, padding 3
    a.Hp = (a.Hc - 1) / 2 + 1; a.Wp = (a.Wc - 1) / 2 + 1;    // max-pool 3 x 3, stride 2, padding 1
    a.tiles_x = (a.Wp + cv::ST_PW - 1) / cv::ST_PW; a.tiles_y = (a.Hp + cv::ST_PH - 1) / cv::ST_PH;
    const long tiles = (long)a.tiles_x * a.tiles_y * N;
    if (tiles > 0x7fffffffL) return fail(SD_E_TOOBIG, "sd_stem_conv_bn_relu_pool: too many tiles");
    static DevFlag attr_set;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute((const void *)cv::stem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, cv::ST_LDS);
        if (e != hipSuccess) return fail((int)e, "stem_kernel: hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        attr_set = true;
    }
    SD_LAUNCH(cv::stem_kernel, dim3((unsigned)tiles), dim3(cv::ST_THREADS), (size_t)cv::ST_LDS, (hipStream_t)stream, a);
    SD_CHECK_LAUNCH("stem_kernel");
    return 0;
}
