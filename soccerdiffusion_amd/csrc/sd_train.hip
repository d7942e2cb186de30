// Training-side kernels (backward of the denoiser blocks, loss, optimizer) for gfx950.
// Interface: include/soccerdiffusion_hip.h ("training" section).  Same fragment maps and
// conventions as sd_kernels.hip; fp32 MFMA except the weight-gradient GEMM (gemm_tn16_kernel below); the forward
// and dX row GEMMs of a training step go through linear() and run on the split-fp16 kernel of sd_f16x3.h.
//
// Reference semantics: one training step of soccer_diffusion/ml/training/train.py:204-240
// (add_noise, forward, F.mse_loss, backward, AdamW.step, OneCycleLR.step) at dropout p=0.

#include <math.h>
#include <cstdlib>
#include <cstring>

#include "../../include/soccerdiffusion_hip.h"
#include "sd_common.h"
#include <algorithm>

// ======================================================================================
// gemm_tn:  dW[N,K] += dY[R,N]^T X[R,K]   and   db[N] += sum_r dY[r,:]
//
// The contraction runs over the R rows, so both MFMA operands are read exactly as they lie
// in memory (row-major, 32 consecutive columns per half-wave = one 128-B segment):
//   A[i = n][k = row] = dY[row][n0 + lane&31],  B[k = row][j = col] = X[row][k0 + lane&31],
// two rows per MFMA.  A workgroup owns a 128 x 128 tile of dW for a chunk of RC rows (4
// waves, each 64 x 64), keeps the partial tile in the accumulators and adds it to dW with
// fp32 atomics (each instruction = two 128-B row segments).  Columns past N / K and rows
// past R read as zero.  The bias gradient falls out of the A fragments.
// ======================================================================================
#ifndef TN_RC
#define TN_RC 256
#endif
// (the fp32-MFMA kernel that implemented this description in round 1 lost every A/B against the fp16 forms below and is gone;
//  the geometry - 128 x 128 tile of dW per workgroup, row chunks of TN_RC, fp32 atomics - is theirs too)

// --------------------------------------------------------------------------------------
// The same product on the fp16 pipe.  The contraction runs over the rows, so an operand scale must be
// constant over everything that is summed into one accumulator - and gradients have no a-priori magnitude.  Each wave
// therefore works in sub-chunks of 32 rows: abs-max of the 32 x 64 values of each operand it holds anyway (a wave-wide
// reduction), two power-of-two scales, split into fp16 hi + lo, 24 fp16 MFMAs into a zero-initialised sub-accumulator,
// which is then added - un-scaled - to the fp32 accumulator of the whole chunk (block floating point per 32 rows).
// --------------------------------------------------------------------------------------
// max inside each group of 16 consecutive lanes (one DPP row): 4 VALU-rate steps, no LDS round trips
__device__ __forceinline__ float row16_maxf(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false)));
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256, 2) void gemm_tn16_kernel(const float *__restrict__ dY, int ldy, const float *__restrict__ X, int ldx,
                                                            float *dW, int ldw, float *db, long R, int N, int K) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_k = (K + 127) / 128;
    const int n0 = (blockIdx.x / tiles_k) * 128 + wm * 64;
    const int k0 = (blockIdx.x % tiles_k) * 128 + wn * 64;
    const long rbeg = (long)blockIdx.y * TN_RC;
    long rend = rbeg + TN_RC;
    if (rend > R) rend = R;
    if (n0 >= N || k0 >= K) return;  // wave-uniform: this wave's 64 x 64 block is empty

    int ncol[2], kcol[2];
    bool nok[2], kok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + t * 32 + l31, k = k0 + t * 32 + l31;
        nok[t] = n < N;
        kok[t] = k < K;
        ncol[t] = nok[t] ? n : 0;
        kcol[t] = kok[t] ? k : 0;
    }
    const bool full_cols = n0 + 64 <= N && k0 + 64 <= K;   // wave-uniform
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[2] = {0.f, 0.f};
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (long r = rbeg; r < rend; r += 32) {
        // this lane's rows of step s (16 rows each): r + 16 s + 8 half + e
        float a[2][2][8], b[2][2][8];
        const bool full_rows = r + 32 <= rend;   // wave-uniform
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const long row = r + 16 * st + 8 * half + e;
                const bool ok = full_rows || row < rend;
                const long rr = ok ? row : rbeg;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[st][t][e] = dY[rr * ldy + ncol[t]];
                    b[st][t][e] = X[rr * ldx + kcol[t]];
                }
                if (!(full_rows && full_cols)) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (!(ok && nok[t])) a[st][t][e] = 0.f;
                        if (!(ok && kok[t])) b[st][t][e] = 0.f;
                    }
                }
            }
        float my = 0.f, mx = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    my = fmaxf(my, fabsf(a[st][t][e]));
                    mx = fmaxf(mx, fabsf(b[st][t][e]));
                    if (st == 0 || true) bsum[t] += a[st][t][e];
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
                    const float va = a[st][t][e] * sy, vb = b[st][t][e] * sx;
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
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int k = k0 + tn * 32 + l31;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < N) atomicAdd(dW + (long)n * ldw + k, acc[tm][tn][r]);
            }
        }
    if (db && (blockIdx.x % tiles_k) == 0 && wn == 0) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const float v = bsum[tm] + __shfl_xor(bsum[tm], 32, 64);
            const int n = n0 + tm * 32 + l31;
            if (half == 0 && n < N) atomicAdd(db + n, v);
        }
    }
}


// --------------------------------------------------------------------------------------
// gemm_tn16s_kernel: the same block-floating-point product with both operands staged through LDS.
// The kernel above reads every operand element with a 4-byte load (the contraction index is the memory row, so a lane's
// 8 k-values sit ldy floats apart): 64 load instructions per lane for 24 MFMAs, and each wave fetches its own 64 columns
// of both operands.  Here a workgroup (128 x 128 tile of dW, 4 waves of 64 x 64) loads a 32-row slab of dY and of X with
// 16-byte loads (8 per thread), splits it into fp16 hi / lo planes kept ROW-major in LDS ([32 rows][128 cols], 320-byte row
// pitch: rows q, q+1, .. land 16 banks apart), and the MFMA fragments - 8 consecutive rows of one column per lane - come
// out of ds_read_b64_tr_b16, gfx950's transposing LDS read (two per fragment).  One power-of-two scale per slab and
// operand (abs-max over 32 x 128, exchanged through LDS), sub-accumulator un-scaled into the fp32 accumulator per slab
// as before.  The next slab's global loads are issued before the MFMAs of the current one.
// Measured (B = 256 training step, rocprofv3): 53.5 -> 44.3 us per d x d weight gradient.  A variant with ONE scale per
// workgroup from an abs-max sweep over its row chunk (no per-slab exchange, no sub-accumulator, two slabs of loads in
// flight) took 53.6 us: the second sweep's reads cost more than the synchronisation they remove - the kernel moves
// 105 MB per launch (each operand is read by two column tiles) and sits at 2.4 TB/s with one workgroup wave on the chip.
// A second variant - 512-row chunks, one workgroup per CU, four slabs of loads in flight, two LDS buffers and one barrier per
// slab - took 58.6 us: with three slabs prefetched a slab still cost 6.9 k cycles, so the slab is bound by its own ~700 vector
// instructions around 24 MFMAs (abs-max, split, transposing-read addresses, un-scale; one wave per SIMD issues them at 5 - 7
// cycles each), not by memory latency.  Hence the running row pointers below instead of 64-bit row * stride products.
// Requires N % 128 == 0, K % 128 == 0, 16-byte aligned rows (the d x d / 3d x d / 2d x d weight gradients).
// --------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int TNS_PITCH = 160;   // halfs per LDS row (128 + 32: 320 bytes)
constexpr int TNS_PLANE = 32 * TNS_PITCH;

__device__ __forceinline__ f16x8 tns_frag(const f16 *plane, int row0, int col0, int lane) {
    // 16-lane group g reads the block rows row0 .. row0+3 (then +4 .. +7), columns col0 + 16 (g & 1) ...; lane 4q+p of the
    // group supplies row q, columns 4p .. 4p+3 and receives column (lane & 15), rows 0..3 of the block
    const int q = (lane & 15) >> 2, p4 = (lane & 3) * 4, gc = ((lane >> 4) & 1) * 16;
    const f16 *a = plane + (row0 + q) * TNS_PITCH + col0 + gc + p4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a + 4 * TNS_PITCH));
    const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
    return __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// (the 4-wave kernel described above was superseded by gemm_tn16d_kernel below, which keeps its slab loop; removed)


// --------------------------------------------------------------------------------------
// gemm_tn16d_kernel: two wave quartets per workgroup, half the atomics.
// What bounds the staged kernel is its epilogue: all 400 workgroups of a d x d gradient finish together and add 26 MB with
// float atomics, which run at ~1.3 TB/s chip-wide (MI355X_MICROARCH.md): ~20 of its 45 us (512-row chunks halve the atomics
// but double every workgroup's serial slab loop: 45.5 us).  Here quartet g = threadIdx.x >> 8 runs the same slab loop on its own
// TN_RC rows with its own LDS planes - the chip holds the same eight waves per CU as before - and the two partial 128 x 128
// tiles are added through LDS before ONE set of atomics per workgroup.  44.8 -> 41.6 us in the training step.
// Ablation, back to back at R = 25 600, N = K = 256 (tools/exp/tn_time.py with diagnostic switches): 36.9 us; without the
// atomics 28.2, without the MFMAs and transposing reads 31.1, without the slab loads 34.1, without all three 18.2 - the bare
// skeleton (abs-max, split into planes, two barriers per slab, the quartet exchange) is half the kernel.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_tn16d_kernel(const float *__restrict__ dY, int ldy, const float *__restrict__ X, int ldx,
                                                             float *dW, int ldw, float *db, long R, int N, int K) {
    __shared__ __attribute__((aligned(16))) f16 sT2[2][4 * TNS_PLANE];   // per quartet: dY hi, dY lo, X hi, X lo
    __shared__ __attribute__((aligned(16))) float sMax2[2][2][16];   // per quartet and operand: 4 waves x 4 DPP-row maxima
    __shared__ float sB[2][8][128];
    const int grp = threadIdx.x >> 8;
    f16 *sT = sT2[grp];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_k = K / 128;
    const int n0 = (blockIdx.x / tiles_k) * 128, k0 = (blockIdx.x % tiles_k) * 128;
    const long wbeg = (long)blockIdx.y * (2 * TN_RC);
    long rbeg = wbeg + (long)grp * TN_RC;
    if (rbeg > R) rbeg = R;
    long rend = rbeg + TN_RC;
    if (rend > R) rend = R;
    // both quartets run the SAME number of slabs (the barriers are workgroup-wide): that of the first, fuller one; rows past a
    // quartet's end read as zero
    const long rows0 = (R - wbeg) < TN_RC ? (R - wbeg) : TN_RC;
    const int n_it = (int)((rows0 + 31) / 32);
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    const float *yp = dY + (rbeg + srow) * (long)ldy + n0 + scol, *xp = X + (rbeg + srow) * (long)ldx + k0 + scol;
    const long ystep = 8L * ldy, xstep = 8L * ldx;
    f32x4 yv[4], xv[4];
    auto load = [&](long r) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const int left = (int)(rend - r) - srow;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            yv[v] = 8 * v < left ? *reinterpret_cast<const f32x4 *>(yp + v * ystep) : z;
            xv[v] = 8 * v < left ? *reinterpret_cast<const f32x4 *>(xp + v * xstep) : z;
        }
        yp += 4 * ystep;
        xp += 4 * xstep;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    load(rbeg);
    for (int it = 0; it < n_it; ++it) {
        const long r = rbeg + 32L * it;
        float my = 0.f, mx = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            bsum = bsum + yv[v];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                my = fmaxf(my, fabsf(yv[v][e]));
                mx = fmaxf(mx, fabsf(xv[v][e]));
            }
        }
        // 16 partial maxima per operand (4 waves x 4 DPP rows) go to LDS; everyone reduces them after the barrier: no
        // cross-row shuffles (12 dependent ds_bpermute round trips per slab before)
        my = row16_maxf(my);
        mx = row16_maxf(mx);
        if ((lane & 15) == 0) {
            sMax2[grp][0][wave * 4 + (lane >> 4)] = my;
            sMax2[grp][1][wave * 4 + (lane >> 4)] = mx;
        }
        __syncthreads();   // maxima visible; every wave is done reading the previous slab's planes
        float sy, sx;
        {
            const f32x4 *py = reinterpret_cast<const f32x4 *>(sMax2[grp][0]), *px = reinterpret_cast<const f32x4 *>(sMax2[grp][1]);
            f32x4 ay = py[0], ax = px[0];
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                const f32x4 ty = py[i], tx = px[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ay[e] = fmaxf(ay[e], ty[e]);
                    ax[e] = fmaxf(ax[e], tx[e]);
                }
            }
            sy = f16_scale_from_bits(__builtin_bit_cast(unsigned, fmaxf(fmaxf(ay[0], ay[1]), fmaxf(ay[2], ay[3]))));
            sx = f16_scale_from_bits(__builtin_bit_cast(unsigned, fmaxf(fmaxf(ax[0], ax[1]), fmaxf(ax[2], ax[3]))));
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            f16x4 h, l;
            f16 *o = sT + (srow + 8 * v) * TNS_PITCH + scol;
            f16_split4(yv[v], sy, h, l);
            *reinterpret_cast<f16x4 *>(o) = h;
            *reinterpret_cast<f16x4 *>(o + TNS_PLANE) = l;
            f16_split4(xv[v], sx, h, l);
            *reinterpret_cast<f16x4 *>(o + 2 * TNS_PLANE) = h;
            *reinterpret_cast<f16x4 *>(o + 3 * TNS_PLANE) = l;
        }
        if (it + 1 < n_it) load(r + 32);
        __syncthreads();
        f32x16 sub[2][2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f16x8 ah[2], al[2], bh[2], bl[2];
            const int row0 = 16 * st + 8 * half;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = tns_frag(sT, row0, wm * 64 + t * 32, lane);
                al[t] = tns_frag(sT + TNS_PLANE, row0, wm * 64 + t * 32, lane);
                bh[t] = tns_frag(sT + 2 * TNS_PLANE, row0, wn * 64 + t * 32, lane);
                bl[t] = tns_frag(sT + 3 * TNS_PLANE, row0, wn * 64 + t * 32, lane);
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
        // 1 / (sy sx) of two powers of two: exponent arithmetic instead of a division
        const float un = __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned, sy * sx));
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = acc[tm][tn] + sub[tm][tn] * un;
    }
    // ---- quartet 1 hands its tile to quartet 0 through the (now free) plane memory: [wave][tile][register][lane] floats ----
    __syncthreads();
    float *xch = reinterpret_cast<float *>(&sT2[0][0]);   // 2 x 40 KB >= 64 KB
    if (grp == 1) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((wave * 4 + tm * 2 + tn) * 16 + r) * 64 + lane] = acc[tm][tn][r];
        *reinterpret_cast<f32x4 *>(&sB[1][srow][scol]) = bsum;
    } else {
        *reinterpret_cast<f32x4 *>(&sB[0][srow][scol]) = bsum;
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int k = k0 + wn * 64 + tn * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    atomicAdd(dW + (long)n * ldw + k, acc[tm][tn][r] + xch[((wave * 4 + tm * 2 + tn) * 16 + r) * 64 + lane]);
                }
            }
        if (db && (blockIdx.x % tiles_k) == 0 && tid < 128) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) v += sB[0][g][tid] + sB[1][g][tid];
            atomicAdd(db + n0 + tid, v);
        }
    }
}

// --------------------------------------------------------------------------------------
// gemm_tn16g_kernel: several weight gradients in ONE launch, ONE power-of-two scale per operand TENSOR.
// What the block-floating-point kernels above pay per 32-row slab - abs-max over the slab, the exchange of the maxima through
// LDS, a sub-accumulator un-scaled into the fp32 accumulator - is half of their time, and a launch per d x d gradient leaves
// each workgroup 8 - 16 slabs between its prologue and 64 KB of atomics.  The fused training chains (sd_train_chain.hip) know
// the abs-max of every operand they produce or consume (their row passes compute it for the per-row scales anyway) and
// leave it in a device word; with it the whole K loop accumulates in the MFMA accumulators and the slab loop is load ->
// split -> LDS -> transposing reads -> 24 MFMAs.  hi + lo keep 22 bits of every element within 2^17 of the tensor's maximum
// and an absolute error of 2^-38 max below that: rows with small gradients lose relative precision only where they cannot
// matter to a sum that contains the large ones.  Grouping (up to 8 problems, e.g. the six d x d gradients of a layer) makes the
// row chunks long (~60 slabs for a decoder layer at B = 256) with ~2 workgroups per CU and a quarter of the atomics.
// Requires N % 4 == 0, K % 4 == 0, 16-byte aligned rows; ragged tiles (N or K not a multiple of 128) are masked.
// --------------------------------------------------------------------------------------
constexpr int TNG_MAX = 8;
struct TnProblem {
    const float *dY, *X;
    float *dW, *db;
    const unsigned *amax_y, *amax_x;
    long R;
    int N, K, ldy, ldx, ldw, tiles_k, tiles, chunks, chunk_rows, wg_begin;
};
struct TnGroup {
    TnProblem p[TNG_MAX];
    int n;
};

#ifndef SD_TNG_OCC
#define SD_TNG_OCC 2
#endif
#ifndef SD_TNG_SETS
#define SD_TNG_SETS 4
#endif
#ifndef SD_TNG_WGS
#define SD_TNG_WGS 512
#endif
// Software pipeline over 16-row slabs (one MFMA k-step): two LDS buffers, ONE barrier per slab, global loads two slabs ahead,
// and the split of slab s + 1 into the other buffer in the same basic block as the transposing reads and the 12 MFMAs of
// slab s, so that one wave's VALU work runs in the shadow of its own MFMAs.  The two-barrier form (load -> split -> barrier ->
// reads + MFMAs -> barrier) measured 147 us for a decoder layer's six gradients and its parts simply added up: without the
// MFMAs and reads 60, without the loads 94, without both 24 - nothing overlapped.
constexpr int TNP_ROWS = 16;
constexpr int TNP_PLANE = TNP_ROWS * TNS_PITCH;     // halfs per plane
__global__ __launch_bounds__(256, SD_TNG_OCC) void gemm_tn16g_kernel(TnGroup g) {
    __shared__ __attribute__((aligned(16))) f16 sT[2][4 * TNP_PLANE];   // per buffer: dY hi, dY lo, X hi, X lo
    __shared__ float sB[8][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroups go round-robin over the 8 XCDs (blockIdx % 8).  All tiles of one row chunk read the same rows of dY / X (a d x d
    // gradient has 2 x 2 tiles, in_proj 6 x 2): relabel so that they are consecutive on ONE XCD and meet in its L2.
    const unsigned nb = gridDim.x, q8 = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7;
    const int vid = (int)(xcd * q8 + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3));
    int pi = 0;
#pragma unroll
    for (int i = 1; i < TNG_MAX; ++i)
        if (i < g.n && vid >= g.p[i].wg_begin) pi = i;
    const TnProblem &P = g.p[pi];
    const int local = vid - P.wg_begin, chunk = local / P.tiles, tile = local - chunk * P.tiles;
    const int n0 = (tile / P.tiles_k) * 128, k0 = (tile % P.tiles_k) * 128;
    const long rbeg = (long)chunk * P.chunk_rows;
    long rend = rbeg + P.chunk_rows;
    if (rend > P.R) rend = P.R;
    const int n_slabs = (int)((rend - rbeg + TNP_ROWS - 1) / TNP_ROWS);
    static_assert(SD_AMAX_WORDS == 64, "one abs-max word per lane");
    const float sy = f16_scale_from_bits(__builtin_bit_cast(unsigned, wave_max(__builtin_bit_cast(float, P.amax_y[lane]))));
    const float sx = f16_scale_from_bits(__builtin_bit_cast(unsigned, wave_max(__builtin_bit_cast(float, P.amax_x[lane]))));
    // staging map: thread -> rows (tid >> 5) + 8 v, columns 4 (tid & 31) .. +3 of both operands
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    const float *yp = P.dY + (rbeg + srow) * (long)P.ldy + n0 + scol, *xp = P.X + (rbeg + srow) * (long)P.ldx + k0 + scol;
    const long ystep = 8L * P.ldy, xstep = 8L * P.ldx;
    int left = (int)(rend - rbeg) - srow;    // rows from this thread's first row of the NEXT slab to load to the chunk's end
#if SD_TNG_SETS == 2
    f32x4 y0[2], x0[2], y1[2], x1[2];
#else
    f32x4 y0[2], x0[2], y1[2], x1[2], y2[2], x2[2], y3[2], x3[2];   // four slabs of loads: three in flight behind the one being split
#endif
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    // Straight-line loads: rows past the chunk's end (the ragged last slab, the prefetches behind it) read the chunk's first
    // row instead and are zeroed when they are split.  Any control flow around the loads - per-lane predicates, or a
    // wave-uniform "is there another slab" - made the compiler keep the loaded registers in different places on the two paths
    // and copy them right behind the loads, with an s_waitcnt vmcnt(0) in front: the prefetch distance was zero.
    // Ragged N / K (the J = 20 sides of the embedding and fc_out gradients): a thread whose four columns lie past the operand's
    // width reads the tile's first columns instead and splits them with scale 0.
    const bool ycol_ok = n0 + scol < P.N, xcol_ok = k0 + scol < P.K;
    const float ycolf = ycol_ok ? 1.0f : 0.0f, xcolf = xcol_ok ? 1.0f : 0.0f;
    const float *ysafe = P.dY + rbeg * (long)P.ldy + n0 + (ycol_ok ? scol : 0), *xsafe = P.X + rbeg * (long)P.ldx + k0 + (xcol_ok ? scol : 0);
    int left_s = left;   // the same counter for the split
    auto load = [&](f32x4 (&yv)[2], f32x4 (&xv)[2]) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const bool ok = 8 * v < left;
#ifdef SD_TNG_ABL_NOLOAD
            yv[v] = f32x4{(float)left, 1.f, 2.f, 3.f}; xv[v] = f32x4{(float)v, 1.f, 2.f, (float)left};
#else
            yv[v] = *reinterpret_cast<const f32x4 *>(ok && ycol_ok ? yp + v * ystep : ysafe);
            xv[v] = *reinterpret_cast<const f32x4 *>(ok && xcol_ok ? xp + v * xstep : xsafe);
#endif
        }
        yp += 2 * ystep;
        xp += 2 * xstep;
        left -= TNP_ROWS;
    };
    auto split = [&](const f32x4 (&yv)[2], const f32x4 (&xv)[2], f16 *buf) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const float keep = 8 * v < left_s ? 1.0f : 0.0f;   // folded into the scale: no extra instruction per element
            bsum = bsum + yv[v] * (keep * ycolf);
            f16x4 h, l;
            f16 *o = buf + (srow + 8 * v) * TNS_PITCH + scol;
            f16_split4_pk(yv[v], sy * (keep * ycolf), h, l);
            *reinterpret_cast<f16x4 *>(o) = h;
            *reinterpret_cast<f16x4 *>(o + TNP_PLANE) = l;
            f16_split4_pk(xv[v], sx * (keep * xcolf), h, l);
            *reinterpret_cast<f16x4 *>(o + 2 * TNP_PLANE) = h;
            *reinterpret_cast<f16x4 *>(o + 3 * TNP_PLANE) = l;
        }
        left_s -= TNP_ROWS;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mfma = [&](const f16 *buf) {
        f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#ifdef SD_TNG_ABL_NOTR   // ablation builds (tools/ab_build.sh): wrong results, timings tell what bounds the kernel
            ah[t] = al[t] = bh[t] = bl[t] = f16x8{(f16)(float)lane, 1, 2, 3, 4, 5, 6, (f16)(float)t};
            asm volatile("" : "+v"(ah[t]), "+v"(al[t]), "+v"(bh[t]), "+v"(bl[t]));
#else
            ah[t] = tns_frag(buf, 8 * half, wm * 64 + t * 32, lane);
            al[t] = tns_frag(buf + TNP_PLANE, 8 * half, wm * 64 + t * 32, lane);
            bh[t] = tns_frag(buf + 2 * TNP_PLANE, 8 * half, wn * 64 + t * 32, lane);
            bl[t] = tns_frag(buf + 3 * TNP_PLANE, 8 * half, wn * 64 + t * 32, lane);
#endif
        }
        // product-major: four independent accumulators between two MFMAs on the same one
#ifndef SD_TNG_ABL_NOMFMA
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t == 0 ? al[tm] : ah[tm], t == 1 ? bl[tn] : bh[tn], acc[tm][tn], 0, 0, 0);
#else
        acc[0][0][0] += (float)ah[0][0] + (float)al[1][1] + (float)bh[0][2] + (float)bl[1][3] + (float)ah[1][4] + (float)al[0][5] + (float)bh[1][6] + (float)bl[0][7];
#endif
    };
#if SD_TNG_SETS == 2   // one slab of loads in flight behind the one being split: 56 registers fewer, three workgroups per CU
    load(y0, x0);
    load(y1, x1);
    split(y0, x0, sT[0]);
    for (int s = 0; s < n_slabs; s += 2) {   // slabs past the end are zeros: the loop body is branch-free
        __syncthreads();   // buffer 0 holds slab s; every wave is done reading buffer 1
        load(y0, x0);
        mfma(sT[0]);
        split(y1, x1, sT[1]);
        __syncthreads();   // buffer 1 holds slab s + 1; every wave is done reading buffer 0
        load(y1, x1);
        mfma(sT[1]);
        split(y0, x0, sT[0]);
    }
#else
    load(y0, x0);
    load(y1, x1);
    load(y2, x2);
    split(y0, x0, sT[0]);
    for (int s = 0; s < n_slabs; s += 4) {   // slabs past the end are zeros: the loop body is branch-free
        __syncthreads();   // buffer 0 holds slab s; every wave is done reading buffer 1
        load(y3, x3);
        mfma(sT[0]);
        split(y1, x1, sT[1]);
        __syncthreads();   // buffer 1 holds slab s + 1; every wave is done reading buffer 0
        load(y0, x0);
        mfma(sT[1]);
        split(y2, x2, sT[0]);
        __syncthreads();
        load(y1, x1);
        mfma(sT[0]);
        split(y3, x3, sT[1]);
        __syncthreads();
        load(y2, x2);
        mfma(sT[1]);
        split(y0, x0, sT[0]);
    }
#endif
    const float un = 1.0f / (sy * sx);
    float *dW = P.dW;
    const int ldw = P.ldw;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int k = k0 + wn * 64 + tn * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (n < P.N && k < P.K) atomicAdd(dW + (long)n * ldw + k, acc[tm][tn][r] * un);
            }
        }
    if (P.db && (tile % P.tiles_k) == 0) {   // column sums of dY: 8 row groups x 128 columns through LDS
        *reinterpret_cast<f32x4 *>(&sB[srow][scol]) = bsum;
        __syncthreads();
        if (tid < 128 && n0 + tid < P.N) {
            float v = 0.f;
#pragma unroll
            for (int gI = 0; gI < 8; ++gI) v += sB[gI][tid];
            atomicAdd(P.db + n0 + tid, v);
        }
    }
}

// bits of max |x| over a (rows x width) fp32 tensor with row stride ld -> the SD_AMAX_WORDS words at amax (atomic max: zero them first):
// the scale of an operand of sd_gemm_tn_grouped that no fused chain produced
__global__ __launch_bounds__(256) void absmax_words_kernel(const float *__restrict__ x, long rows, int width, int ld, unsigned *amax) {
    const int w4 = width >> 2;
    const long n4 = rows * w4;
    float m = 0.f;
    if (ld == width) {   // contiguous: a flat 16-byte stream, four loads in flight per thread
        const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
        const long stride = (long)gridDim.x * blockDim.x;
        long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(fmaxf(m, fmaxf(fabsf(a[e]), fabsf(b[e]))), fmaxf(fabsf(c[e]), fabsf(d[e])));
        }
        for (; i < n4; i += stride) {
            const f32x4 a = x4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(a[0]), fabsf(a[1]))), fmaxf(fabsf(a[2]), fabsf(a[3])));
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            const long r = i / w4;
            const int c = (int)(i - r * w4) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + r * ld + c);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(amax + (blockIdx.x & (SD_AMAX_WORDS - 1)), __builtin_bit_cast(unsigned, m));
}

extern "C" int sd_op_absmax(const float *x, int64_t rows, int width, int ld, uint32_t *amax, void *stream) {
    if (!x || !amax || rows <= 0 || width <= 0 || width % 4 || ld < width || ld % 4 || (reinterpret_cast<uintptr_t>(x) & 15))
        return fail(SD_E_BADARG, "sd_op_absmax: x must be 16-byte aligned with width and row stride multiples of 4");
    long blocks = (rows * (width / 4) + 1023) / 1024;   // ~4 x 16 bytes per thread
    if (blocks > 1024) blocks = 1024;
    SD_LAUNCH(absmax_words_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)rows, width, ld, amax);
    SD_CHECK_LAUNCH("absmax_words_kernel");
    return 0;
}

// workgroups of a grouped launch when every workgroup takes (at most) per_wg 32-row slabs of one 128 x 128 tile (even counts)
static long tng_total_wgs(const sd_gemm_tn_problem *pr, int cnt, long per_wg) {
    long total = 0;
    for (int i = 0; i < cnt; ++i) {
        const sd_gemm_tn_problem &q = pr[i];
        const long slabs = (q.R + 31) / 32;
        long cs = per_wg < slabs ? per_wg : slabs;
        cs += cs & 1;
        total += (long)((q.N + 127) / 128) * ((q.K + 127) / 128) * ((slabs + cs - 1) / cs);
    }
    return total;
}
// The 32-row slabs every workgroup of ONE launch (cnt <= TNG_MAX problems) takes: ~2 workgroups per CU (256 CUs), each with the same
// number of slabs, and no more workgroups than fit at once (chunks are rounded up per problem: nine extra workgroups behind a full
// chip made a second round, +30 % on the launch).  Once per_wg has reached every problem's slab count the total no longer shrinks:
// a group with more than SD_TNG_WGS output tiles (e.g. eight 1152 x 1152 gradients) then simply runs as more than one round of
// workgroups (this search used to spin forever on such a group).  Shared by the launch and by the exported plan function,
// so the CPU test of the latter exercises the loop the launch runs.
static long tng_per_wg(const sd_gemm_tn_problem *pr, int cnt) {
    long slab_tiles = 0, max_slabs = 1;
    for (int i = 0; i < cnt; ++i) {
        const sd_gemm_tn_problem &q = pr[i];
        slab_tiles += (long)((q.N + 127) / 128) * ((q.K + 127) / 128) * ((q.R + 31) / 32);
        max_slabs = std::max(max_slabs, (long)((q.R + 31) / 32));
    }
    long per_wg = (slab_tiles + SD_TNG_WGS - 1) / SD_TNG_WGS;
    if (per_wg < 8) per_wg = 8;
    for (;; ++per_wg)
        if (tng_total_wgs(pr, cnt, per_wg) <= SD_TNG_WGS || per_wg >= max_slabs) break;
    return per_wg;
}
// exported for the CPU-side unit test of the partitioning arithmetic: the per_wg the FIRST launch of sd_gemm_tn_grouped would use
// for `cnt` problems of R rows and N x K outputs each (the call splits its problems into launches of at most TNG_MAX, as
// sd_gemm_tn_grouped does; total_wgs: the workgroups of all launches); never launches anything
extern "C" long sd_gemm_tn_grouped_plan(const long *R, const long *N, const long *K, int cnt, long *total_wgs) {
    if (!R || !N || !K || cnt <= 0 || cnt > 64) return -1;
    sd_gemm_tn_problem pr[64] = {};
    for (int i = 0; i < cnt; ++i) {
        if (R[i] <= 0 || N[i] <= 0 || K[i] <= 0) return -1;
        pr[i].R = R[i]; pr[i].N = (int)N[i]; pr[i].K = (int)K[i];
    }
    long first_per_wg = 0, total = 0;
    for (int first = 0; first < cnt; first += TNG_MAX) {
        const int c = cnt - first < TNG_MAX ? cnt - first : TNG_MAX;
        const long per_wg = tng_per_wg(pr + first, c);
        if (first == 0) first_per_wg = per_wg;
        total += tng_total_wgs(pr + first, c, per_wg);
    }
    if (total_wgs) *total_wgs = total;
    return first_per_wg;
}

extern "C" int sd_gemm_tn_grouped(const sd_gemm_tn_problem *pr, int n, void *stream) {
    if (!pr || n <= 0) return fail(SD_E_BADARG, "sd_gemm_tn_grouped: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(SD_KCLASS_PANEL_GEMM, s);
    for (int first = 0; first < n; first += TNG_MAX) {
        const int cnt = n - first < TNG_MAX ? n - first : TNG_MAX;
        TnGroup g;
        g.n = cnt;
        for (int i = 0; i < cnt; ++i) {
            const sd_gemm_tn_problem &q = pr[first + i];
            if (!q.dY || !q.X || !q.dW || !q.amax_dy || !q.amax_x || q.R <= 0 || q.N <= 0 || q.K <= 0 || q.N % 4 || q.K % 4 || q.ldy < q.N ||
                q.ldx < q.K || q.ldw < q.K || q.ldy % 4 || q.ldx % 4 || (reinterpret_cast<uintptr_t>(q.dY) & 15) || (reinterpret_cast<uintptr_t>(q.X) & 15))
                return fail(SD_E_BADARG, "sd_gemm_tn_grouped: operands must be 16-byte aligned with N, K and the row strides multiples of 4, and carry their abs-max words");
        }
        const long per_wg = tng_per_wg(pr + first, cnt);
        int wgs = 0;
        for (int i = 0; i < cnt; ++i) {
            const sd_gemm_tn_problem &q = pr[first + i];
            TnProblem &P = g.p[i];
            P.dY = q.dY; P.X = q.X; P.dW = q.dW; P.db = q.db; P.amax_y = q.amax_dy; P.amax_x = q.amax_x; P.R = q.R;
            P.N = q.N; P.K = q.K;
            P.ldy = q.ldy; P.ldx = q.ldx; P.ldw = q.ldw; P.tiles_k = (q.K + 127) / 128; P.tiles = ((q.N + 127) / 128) * P.tiles_k;
            const long slabs = (q.R + 31) / 32;
            long chunk_slabs = per_wg < slabs ? per_wg : slabs;
            chunk_slabs += chunk_slabs & 1;   // chunks of a multiple of 64 rows: the kernel's loop takes four 16-row slabs per turn
            P.chunk_rows = (int)(chunk_slabs * 32);
            P.chunks = (int)((slabs + chunk_slabs - 1) / chunk_slabs);
            P.wg_begin = wgs;
            wgs += P.tiles * P.chunks;
        }
        SD_LAUNCH(gemm_tn16g_kernel, dim3((unsigned)wgs), dim3(256), 0, s, g);
        SD_CHECK_LAUNCH("gemm_tn16g_kernel");
    }
    return 0;
}

extern "C" int sd_op_gemm_tn(const float *dY, int ldy, const float *X, int ldx, float *dW, int ldw, float *db, long R,
                             int N, int K, void *stream) {
    if (!dY || !X || !dW || R <= 0 || N <= 0 || K <= 0 || ldy < N || ldx < K || ldw < K)
        return fail(SD_E_BADARG, "sd_op_gemm_tn: bad argument");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(SD_KCLASS_PANEL_GEMM, s);
    dim3 grid(((N + 127) / 128) * ((K + 127) / 128), (unsigned)((R + TN_RC - 1) / TN_RC));
    const bool staged = N % 128 == 0 && K % 128 == 0 && ldy % 4 == 0 && ldx % 4 == 0 &&
                        (reinterpret_cast<uintptr_t>(dY) & 15) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    if (staged) {
        dim3 gridd(((N + 127) / 128) * ((K + 127) / 128), (unsigned)((R + 2 * TN_RC - 1) / (2 * TN_RC)));
        SD_LAUNCH(gemm_tn16d_kernel, gridd, dim3(512), 0, s, dY, ldy, X, ldx, dW, ldw, db, R, N, K);
    } else SD_LAUNCH(gemm_tn16_kernel, grid, dim3(256), 0, s, dY, ldy, X, ldx, dW, ldw, db, R, N, K);   // ragged shapes (J = 20 columns, odd strides)
    SD_CHECK_LAUNCH("gemm_tn_kernel");
    return 0;
}

// ======================================================================================
// LayerNorm forward (materialised, saves mean / rstd) and backward.  One wave per row.
//   y = (x - mean) * rstd * g + b
//   dx = rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat))   [+ dres]
//   dg += sum_r dy * xhat ; db += sum_r dy    (per-workgroup partials -> fp32 atomics)
// ======================================================================================
// A lane owns PL = D / 64 CONSECUTIVE columns (one 16-byte access at D = 256): the dword-per-lane form of these two kernels
// took 11 / 35 us at R = 25 600, D = 256 for 52 / 78 MB of traffic.
template <int PL>
__device__ __forceinline__ void ln_load(float (&v)[PL], const float *p) {
    if constexpr (PL % 4 == 0) {
#pragma unroll
        for (int j = 0; j < PL; j += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(p + j);
            v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3];
        }
    } else if constexpr (PL == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2 *>(p);
        v[0] = t[0]; v[1] = t[1];
    } else {
        v[0] = p[0];
    }
}
template <int PL>
__device__ __forceinline__ void ln_store(float *p, const float (&v)[PL]) {
    if constexpr (PL % 4 == 0) {
#pragma unroll
        for (int j = 0; j < PL; j += 4) *reinterpret_cast<f32x4 *>(p + j) = f32x4{v[j], v[j + 1], v[j + 2], v[j + 3]};
    } else if constexpr (PL == 2) {
        *reinterpret_cast<f32x2 *>(p) = f32x2{v[0], v[1]};
    } else {
        p[0] = v[0];
    }
}

template <int D>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                             const float *__restrict__ b, float *__restrict__ y,
                                                             float *__restrict__ mean, float *__restrict__ rstd,
                                                             long R) {
    constexpr int PL = D / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c0 = lane * PL;
    float gw[PL], gb[PL];
    ln_load<PL>(gw, g + c0);
    ln_load<PL>(gb, b + c0);
    for (long row = (long)blockIdx.x * 4 + wave; row < R; row += (long)gridDim.x * 4) {
        float v[PL], s = 0.f;
        ln_load<PL>(v, x + row * D + c0);
#pragma unroll
        for (int j = 0; j < PL; ++j) s += v[j];
        const float mu = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < PL; ++j) {
            v[j] -= mu;
            q += v[j] * v[j];
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + SD_LN_EPS);
#pragma unroll
        for (int j = 0; j < PL; ++j) v[j] = v[j] * rs * gw[j] + gb[j];
        ln_store<PL>(y + row * D + c0, v);
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                             const float *__restrict__ mean,
                                                             const float *__restrict__ rstd,
                                                             const float *__restrict__ g, const float *dres, float *dx,
                                                             float *dg, float *db, long R) {
    constexpr int PL = D / 64;
    __shared__ float red[2][4][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c0 = lane * PL;
    float gw[PL], pg[PL], pb[PL];
    ln_load<PL>(gw, g + c0);
#pragma unroll
    for (int j = 0; j < PL; ++j) {
        pg[j] = 0.f;
        pb[j] = 0.f;
    }
    // NU rows of a wave in flight (the row loop is a chain of load -> wave reduction -> store) and FEW workgroups: every
    // workgroup ends with 2 D atomics on the same 2 D addresses, and same-address float atomics run at ~22 G/s chip-wide
    // (MI355X_MICROARCH.md) - 1 024 workgroups spent 23 of their 38 us there, 2 048 took 57 us
    constexpr int NU = 4;
    const long stride = (long)gridDim.x * 4;
    for (long row0 = (long)blockIdx.x * 4 + wave; row0 < R; row0 += NU * stride) {
        float d[NU][PL], xh[NU][PL], r[NU][PL], mu[NU], rs[NU];
        bool ok[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const long row = row0 + u * stride;
            ok[u] = row < R;
            const long rr = ok[u] ? row : row0;
            mu[u] = mean[rr];
            rs[u] = rstd[rr];
            ln_load<PL>(d[u], dy + rr * D + c0);
            ln_load<PL>(xh[u], x + rr * D + c0);
            if (dres) ln_load<PL>(r[u], dres + rr * D + c0);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (!ok[u]) continue;   // wave-uniform
            const long row = row0 + u * stride;
            float gd[PL], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < PL; ++j) {
                xh[u][j] = (xh[u][j] - mu[u]) * rs[u];
                gd[j] = d[u][j] * gw[j];
                s1 += gd[j];
                s2 += gd[j] * xh[u][j];
                pg[j] += d[u][j] * xh[u][j];
                pb[j] += d[u][j];
            }
            s1 = wave_sum(s1) * (1.0f / D);
            s2 = wave_sum(s2) * (1.0f / D);
            float o[PL];
#pragma unroll
            for (int j = 0; j < PL; ++j) {
                o[j] = rs[u] * (gd[j] - s1 - xh[u][j] * s2);
                if (dres) o[j] += r[u][j];
            }
            ln_store<PL>(dx + row * D + c0, o);
        }
    }
#pragma unroll
    for (int j = 0; j < PL; ++j) {
        red[0][wave][c0 + j] = pg[j];
        red[1][wave][c0 + j] = pb[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        atomicAdd(dg + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(db + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

extern "C" int sd_op_layernorm_fwd(const float *x, const float *g, const float *b, float *y, float *mean, float *rstd,
                                   long R, int d, void *stream) {
    if (!x || !g || !b || !y || R <= 0) return fail(SD_E_BADARG, "sd_op_layernorm_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((R + 3) / 4 > 4096 ? 4096 : (R + 3) / 4)), block(256);
    switch (d) {
        case 64: SD_LAUNCH(layernorm_fwd_kernel<64>, grid, block, 0, s, x, g, b, y, mean, rstd, R); break;
        case 128: SD_LAUNCH(layernorm_fwd_kernel<128>, grid, block, 0, s, x, g, b, y, mean, rstd, R); break;
        case 256: SD_LAUNCH(layernorm_fwd_kernel<256>, grid, block, 0, s, x, g, b, y, mean, rstd, R); break;
        case 512: SD_LAUNCH(layernorm_fwd_kernel<512>, grid, block, 0, s, x, g, b, y, mean, rstd, R); break;
        default: return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
    }
    SD_CHECK_LAUNCH("layernorm_fwd_kernel");
    return 0;
}

extern "C" int sd_op_layernorm_bwd(const float *dy, const float *x, const float *mean, const float *rstd, const float *g,
                                   const float *dres, float *dx, float *dg, float *db, long R, int d, void *stream) {
    if (!dy || !x || !mean || !rstd || !g || !dx || !dg || !db || R <= 0)
        return fail(SD_E_BADARG, "sd_op_layernorm_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((R + 3) / 4 > 256 ? 256 : (R + 3) / 4)), block(256);
    switch (d) {
        case 64: SD_LAUNCH(layernorm_bwd_kernel<64>, grid, block, 0, s, dy, x, mean, rstd, g, dres, dx, dg, db, R); break;
        case 128: SD_LAUNCH(layernorm_bwd_kernel<128>, grid, block, 0, s, dy, x, mean, rstd, g, dres, dx, dg, db, R); break;
        case 256: SD_LAUNCH(layernorm_bwd_kernel<256>, grid, block, 0, s, dy, x, mean, rstd, g, dres, dx, dg, db, R); break;
        case 512: SD_LAUNCH(layernorm_bwd_kernel<512>, grid, block, 0, s, dy, x, mean, rstd, g, dres, dx, dg, db, R); break;
        default: return fail(SD_E_BADDIM, "hidden_dim must be one of 64, 128, 256, 512");
    }
    SD_CHECK_LAUNCH("layernorm_bwd_kernel");
    return 0;
}

// ======================================================================================
// elementwise: GELU forward / backward, MSE loss, AdamW, column sums, small-K linear
// ======================================================================================
__global__ void gelu_fwd_kernel(const float *__restrict__ pre, float *__restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = gelu_erf(pre[i]);
}

__global__ void gelu_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ pre, float *__restrict__ dpre, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float u = pre[i];
        const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * u * u);
        dpre[i] = dy[i] * (cdf + u * pdf);
    }
}

extern "C" int sd_op_gelu_fwd(const float *pre, float *out, long n, void *stream) {
    if (!pre || !out || n <= 0) return fail(SD_E_BADARG, "sd_op_gelu_fwd: bad argument");
    SD_LAUNCH(gelu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pre, out, n);
    SD_CHECK_LAUNCH("gelu_fwd_kernel");
    return 0;
}

extern "C" int sd_op_gelu_bwd(const float *dy, const float *pre, float *dpre, long n, void *stream) {
    if (!dy || !pre || !dpre || n <= 0) return fail(SD_E_BADARG, "sd_op_gelu_bwd: bad argument");
    SD_LAUNCH(gelu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, pre, dpre, n);
    SD_CHECK_LAUNCH("gelu_bwd_kernel");
    return 0;
}

// --------------------------------------------------------------------------------------
// Dropout (sd_common.h: one Philox mask function for every kernel).  The elementwise forms: out = x o m (also the
// backward of a fused-epilogue site: dy o m), the mask itself (tests hand it to the oracle), and GELU with the
// dropout that follows it in the FFN (torch: linear2(dropout(activation(linear1(x))))).
// One thread per quad = 4 consecutive columns of one row = one Philox call.
// --------------------------------------------------------------------------------------
template <int MODE>   // 0: out = x o m;  1: mask only;  2: out = gelu(x) o m;  3: dpre = dy o m o gelu'(pre)
__global__ void dropout_quads_kernel(const float *__restrict__ x, const float *__restrict__ pre, float *__restrict__ out, long rows,
                                     int width, DropoutArgs da) {
    const int wq = (width + 3) >> 2;
    const long nq = rows * wq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (long)gridDim.x * blockDim.x) {
        const long row = i / wq;
        const int c0 = (int)(i - row * wq) * 4;
        const f32x4 m = dropout_quad(da, (unsigned long)i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c0 + e >= width) break;
            const long at = row * width + c0 + e;
            float v;
            if (MODE == 0) v = x[at] * m[e];
            else if (MODE == 1) v = m[e];
            else if (MODE == 2) v = gelu_erf(x[at]) * m[e];
            else {
                const float u = pre[at];
                const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
                const float pdf = 0.39894228040143267794f * expf(-0.5f * u * u);
                v = x[at] * m[e] * (cdf + u * pdf);
            }
            out[at] = v;
        }
    }
}

static int dropout_args_ok(float p, const char *what) {
    if (!(p >= 0.f) || !(p < 1.f)) return fail(SD_E_BADARG, what);
    return 0;
}

extern "C" int sd_op_dropout(const float *x, float *out, long rows, int width, float p, uint64_t seed, uint64_t site, void *stream) {
    if (!x || !out || rows <= 0 || width <= 0) return fail(SD_E_BADARG, "sd_op_dropout: bad argument");
    if (int rc = dropout_args_ok(p, "sd_op_dropout: p must be in [0, 1)")) return rc;
    SD_LAUNCH(dropout_quads_kernel<0>, dim3(grid_for(rows * ((width + 3) / 4))), dim3(256), 0, (hipStream_t)stream, x, nullptr, out, rows,
              width, make_dropout(p, seed, site));
    SD_CHECK_LAUNCH("dropout_kernel");
    return 0;
}

extern "C" int sd_op_dropout_mask(float *mask, long rows, int width, float p, uint64_t seed, uint64_t site, void *stream) {
    if (!mask || rows <= 0 || width <= 0) return fail(SD_E_BADARG, "sd_op_dropout_mask: bad argument");
    if (int rc = dropout_args_ok(p, "sd_op_dropout_mask: p must be in [0, 1)")) return rc;
    SD_LAUNCH(dropout_quads_kernel<1>, dim3(grid_for(rows * ((width + 3) / 4))), dim3(256), 0, (hipStream_t)stream, nullptr, nullptr, mask,
              rows, width, make_dropout(p, seed, site));
    SD_CHECK_LAUNCH("dropout_mask_kernel");
    return 0;
}

extern "C" int sd_op_gelu_dropout_fwd(const float *pre, float *out, long rows, int width, float p, uint64_t seed, uint64_t site,
                                      void *stream) {
    if (!pre || !out || rows <= 0 || width <= 0) return fail(SD_E_BADARG, "sd_op_gelu_dropout_fwd: bad argument");
    if (int rc = dropout_args_ok(p, "sd_op_gelu_dropout_fwd: p must be in [0, 1)")) return rc;
    SD_LAUNCH(dropout_quads_kernel<2>, dim3(grid_for(rows * ((width + 3) / 4))), dim3(256), 0, (hipStream_t)stream, pre, nullptr, out, rows,
              width, make_dropout(p, seed, site));
    SD_CHECK_LAUNCH("gelu_dropout_fwd_kernel");
    return 0;
}

extern "C" int sd_op_gelu_dropout_bwd(const float *dy, const float *pre, float *dpre, long rows, int width, float p, uint64_t seed,
                                      uint64_t site, void *stream) {
    if (!dy || !pre || !dpre || rows <= 0 || width <= 0) return fail(SD_E_BADARG, "sd_op_gelu_dropout_bwd: bad argument");
    if (int rc = dropout_args_ok(p, "sd_op_gelu_dropout_bwd: p must be in [0, 1)")) return rc;
    SD_LAUNCH(dropout_quads_kernel<3>, dim3(grid_for(rows * ((width + 3) / 4))), dim3(256), 0, (hipStream_t)stream, dy, pre, dpre, rows,
              width, make_dropout(p, seed, site));
    SD_CHECK_LAUNCH("gelu_dropout_bwd_kernel");
    return 0;
}

// F.mse_loss(pred, target) (mean) and its gradient 2 (pred - target) / n  (train.py:229)
__global__ void mse_kernel(const float *__restrict__ pred, const float *__restrict__ target, float *__restrict__ grad,
                           double *partial, long n, float scale) {
    __shared__ double red[4];
    double s = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        s += (double)d * d;
        if (grad) grad[i] = d * scale;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void mse_finish_kernel(const double *partial, int nblocks, float *loss, double inv_n) {
    double s = 0.0;
    for (int i = 0; i < nblocks; ++i) s += partial[i];  // fixed order: deterministic
    loss[0] = (float)(s * inv_n);
}

extern "C" int sd_mse_loss(const float *pred, const float *target, float *loss, float *grad, void *scratch256d, long n,
                           void *stream) {
    if (!pred || !target || !loss || !scratch256d || n <= 0) return fail(SD_E_BADARG, "sd_mse_loss: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = grid_for(n) > 256 ? 256 : grid_for(n);
    SD_LAUNCH(mse_kernel, dim3(nb), dim3(256), 0, s, pred, target, grad, (double *)scratch256d, n, 2.0f / (float)n);
    SD_CHECK_LAUNCH("mse_kernel");
    SD_LAUNCH(mse_finish_kernel, dim3(1), dim3(1), 0, s, (const double *)scratch256d, (int)nb, loss, 1.0 / (double)n);
    SD_CHECK_LAUNCH("mse_finish_kernel");
    return 0;
}

// torch.optim.AdamW single-tensor update order (decoupled weight decay, bias-corrected):
//   p *= 1 - lr*wd ; m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2) g^2
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void adamw_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                             float *__restrict__ v, long n, float decay, float one_minus_b1, float b2,
                             float one_minus_b2, float step_size, float bc2_sqrt, float eps) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * decay;
        float mi = m[i];
        mi = mi + one_minus_b1 * (gi - mi);
        const float vi = v[i] * b2 + one_minus_b2 * (gi * gi);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

// The same update with its seven scalars read from DEVICE memory (hyper[0..6] = decay, 1 - beta1, beta2, 1 - beta2,
// lr / bias_correction1, sqrt(bias_correction2), eps): the form a hipGraph-captured training step needs - the learning rate
// (OneCycleLR), beta1 (cycled momentum) and the bias corrections change every step, kernel arguments of a graph do not.
__global__ void adamw_dev_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long n,
                                 const float *__restrict__ hyper) {
    const float decay = hyper[0], one_minus_b1 = hyper[1], b2 = hyper[2], one_minus_b2 = hyper[3], step_size = hyper[4],
                bc2_sqrt = hyper[5], eps = hyper[6];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * decay;
        float mi = m[i];
        mi = mi + one_minus_b1 * (gi - mi);
        const float vi = v[i] * b2 + one_minus_b2 * (gi * gi);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" int sd_adamw_step_dev(float *p, const float *g, float *m, float *v, long n, const float *hyper7_dev, void *stream) {
    if (!p || !g || !m || !v || !hyper7_dev || n <= 0) return fail(SD_E_BADARG, "sd_adamw_step_dev: bad argument");
    SD_LAUNCH(adamw_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper7_dev);
    SD_CHECK_LAUNCH("adamw_dev_kernel");
    return 0;
}

/* host-side helper shared with the binding: the seven scalars of step `step` exactly as sd_adamw_step derives them */
extern "C" int sd_adamw_hyper(double lr, double beta1, double beta2, double eps, double weight_decay, long step, float *hyper7_host) {
    if (!hyper7_host || step <= 0) return fail(SD_E_BADARG, "sd_adamw_hyper: bad argument");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hyper7_host[0] = (float)(1.0 - lr * weight_decay);
    hyper7_host[1] = (float)(1.0 - beta1);
    hyper7_host[2] = (float)beta2;
    hyper7_host[3] = (float)(1.0 - beta2);
    hyper7_host[4] = (float)(lr / bc1);
    hyper7_host[5] = (float)sqrt(bc2);
    hyper7_host[6] = (float)eps;
    return 0;
}

extern "C" int sd_adamw_step(float *p, const float *g, float *m, float *v, long n, double lr, double beta1, double beta2,
                             double eps, double weight_decay, long step, void *stream) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) return fail(SD_E_BADARG, "sd_adamw_step: bad argument");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    SD_LAUNCH(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
              (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
              (float)(lr / bc1), (float)sqrt(bc2), (float)eps);
    SD_CHECK_LAUNCH("adamw_kernel");
    return 0;
}

// out[c] += sum_r src[r*row_stride + c]   (bias-like gradients over strided rows)
__global__ void colsum_kernel(const float *__restrict__ src, long row_stride, long rows, int width, float *out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    float s = 0.f;
    for (long r = blockIdx.y; r < rows; r += gridDim.y) s += src[r * row_stride + c];
    atomicAdd(out + c, s);
}

extern "C" int sd_op_colsum(const float *src, long row_stride, long rows, int width, float *out, void *stream) {
    if (!src || !out || rows <= 0 || width <= 0) return fail(SD_E_BADARG, "sd_op_colsum: bad argument");
    dim3 grid((width + 255) / 256, (unsigned)(rows < 64 ? rows : 64));
    SD_LAUNCH(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, row_stride, rows, width, out);
    SD_CHECK_LAUNCH("colsum_kernel");
    return 0;
}

// out[R,N] = A[R,K] @ B[K,N] for a small contraction (K <= 64: joints): backward of fc_out
// w.r.t. its input.  One thread per output element, A row cached in LDS.
__global__ __launch_bounds__(256) void small_k_matmul_kernel(const float *__restrict__ A, const float *__restrict__ Bm,
                                                              float *__restrict__ out, long R, int K, int N) {
    extern __shared__ float sAB[];  // [16][K] rows of A, then B [K][N]
    float *sArow = sAB;
    float *sB = sAB + 16 * K;
    for (int i = threadIdx.x; i < K * N; i += 256) sB[i] = Bm[i];
    const long r0 = (long)blockIdx.x * 16;
    for (int i = threadIdx.x; i < 16 * K; i += 256) {
        const long r = r0 + i / K;
        sArow[i] = r < R ? A[r * K + i % K] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * N; i += 256) {
        const int row = i / N, c = i - row * N;
        if (r0 + row >= R) continue;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(sArow[row * K + k], sB[k * N + c], acc);
        out[(r0 + row) * N + c] = acc;
    }
}

extern "C" int sd_op_small_k_matmul(const float *A, const float *Bm, float *out, long R, int K, int N, void *stream) {
    if (!A || !Bm || !out || R <= 0 || K <= 0 || K > 64 || N <= 0 || N > 512)
        return fail(SD_E_BADARG, "sd_op_small_k_matmul: bad argument (K <= 64, N <= 512)");
    const size_t lds = ((size_t)16 * K + (size_t)K * N) * sizeof(float);
    if (lds > 64 * 1024) {
        static DevFlag attr_set;       
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)small_k_matmul_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
    }
    SD_LAUNCH(small_k_matmul_kernel, dim3((unsigned)((R + 15) / 16)), dim3(256), lds, (hipStream_t)stream, A, Bm, out, R, K, N);
    SD_CHECK_LAUNCH("small_k_matmul_kernel");
    return 0;
}

// ======================================================================================
// Attention backward, one workgroup per (sample, head).
//
// Same transposed orientation as the forward: with keys on the accumulator rows and a
// query per lane column,  P^T = exp2(K Q^T * c - lse),  dP^T = V dO^T,
// dS^T = P^T o (dP^T - delta_q) / sqrt(hd)  are all in-lane, and  dQ^T += K^T dS^T  takes dS^T
// straight from the accumulator as its B operand.  dV = P^T dO and dK = dS^T Q contract over
// queries (the lane index), so the P^T / dS^T tiles of the 4 waves make one trip through
// LDS ([32 keys][QP queries]) and each wave then owns one (dV | dK, feature tile) job.
// Q and dO of the query pass sit in LDS row-major and serve both as B fragments of the
// first two products (16-B reads) and of the last two (4-B reads).
// ======================================================================================
template <int HD>
struct AttnBwdCfg {
    static constexpr int QP = (HD >= 128) ? 64 : 128;   // queries per pass
    static constexpr int KC = 64;                       // keys per LDS chunk
    static constexpr int FT = (HD + 31) / 32;
    static constexpr int FW = FT * 32;                  // feature width incl. zero padding
    static constexpr int LDF = FW + 4;                  // row stride of Q / dO / K / V images
    static constexpr int LDP = QP + 4;                  // row stride of P^T / dS^T tiles
    static constexpr int KSTEPS = HD / 8;
    static constexpr size_t LDS_FLOATS = (size_t)2 * QP * LDF + 2 * KC * LDF + 2 * 32 * LDP;
    static constexpr size_t LDS_BYTES = LDS_FLOATS * sizeof(float);
};

// DROP: the forward dropped probabilities (P_d = P o m, O = P_d V).  Then dV = P_d^T dO, dP = (V dO^T) o m,
// dS = P o (dP - delta) / sqrt(hd) with the same delta_q = sum_f dO O (= sum_k dP_d P_d); the mask is regenerated
// from (seed, site): row (b, h, q), column = key, 4 consecutive keys (registers 4g..4g+3) per Philox call.
template <int HD, bool DROP>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float *__restrict__ q, int ldq,
                                                             const float *__restrict__ k,
                                                             const float *__restrict__ v, int ldkv,
                                                             const float *__restrict__ o, int ldo,
                                                             const float *__restrict__ dO, int lddo,
                                                             const float *__restrict__ lse2, float *dq, int lddq,
                                                             float *dk, float *dv, int lddkv, int Tq, int S, int heads,
                                                             float scale, DropoutArgs da) {
    using C = AttnBwdCfg<HD>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sQ = smem;
    float *sdO = sQ + C::QP * C::LDF;
    float *sK = sdO + C::QP * C::LDF;
    float *sV = sK + C::KC * C::LDF;
    float *sP = sV + C::KC * C::LDF;
    float *sdS = sP + 32 * C::LDP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const float sl2e = scale * 1.44269504088896340736f;
    const float *qb = q + (long)b * Tq * ldq + h * HD;
    const float *ob = o + (long)b * Tq * ldo + h * HD;
    const float *dob = dO + (long)b * Tq * lddo + h * HD;
    const float *kb = k + (long)b * S * ldkv + h * HD;
    const float *vb = v + (long)b * S * ldkv + h * HD;
    float *dqb = dq + (long)b * Tq * lddq + h * HD;
    float *dkb = dk + (long)b * S * lddkv + h * HD;
    float *dvb = dv + (long)b * S * lddkv + h * HD;
    constexpr int F4 = C::FW / 4;

    for (int qpass = 0; qpass < Tq; qpass += C::QP) {
        __syncthreads();
        // ---- stage Q and dO rows of this pass (zero rows past Tq, zero feature padding) ---
        for (int i = tid; i < C::QP * F4; i += 256) {
            const int row = i / F4, c4 = i - row * F4;
            const int qi = qpass + row;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
            if (qi < Tq && c4 * 4 < HD) {
                a = *reinterpret_cast<const f32x4 *>(qb + (long)qi * ldq + c4 * 4);
                d = *reinterpret_cast<const f32x4 *>(dob + (long)qi * lddo + c4 * 4);
            }
            *reinterpret_cast<f32x4 *>(sQ + row * C::LDF + c4 * 4) = a;
            *reinterpret_cast<f32x4 *>(sdO + row * C::LDF + c4 * 4) = d;
        }
        const int q0 = wave * 32;                 // this wave's queries inside the pass
        const int qi = qpass + q0 + l31;
        const bool wave_active = q0 < C::QP && qpass + q0 < Tq;  // wave-uniform
        const bool q_ok = wave_active && qi < Tq;
        // delta_q = sum_f dO[q,f] O[q,f]; lse of the query
        float delta = 0.f, lse = 0.f;
        if (q_ok) {
            for (int f = half * (HD / 2); f < (half + 1) * (HD / 2); f += 4) {
                const f32x4 ov = *reinterpret_cast<const f32x4 *>(ob + (long)qi * ldo + f);
                const f32x4 dv4 = *reinterpret_cast<const f32x4 *>(dob + (long)qi * lddo + f);
                delta += ov[0] * dv4[0] + ov[1] * dv4[1] + ov[2] * dv4[2] + ov[3] * dv4[3];
            }
            lse = lse2[((long)b * heads + h) * Tq + qi];
        }
        delta += __shfl_xor(delta, 32, 64);
        f32x16 dqT[C::FT];
#pragma unroll
        for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqT[ft][r] = 0.f;
        __syncthreads();
        // Q / dO fragments of this wave's queries: B[k = feature][j = query]
        f32x4 qf[C::KSTEPS], dof[C::KSTEPS];
#pragma unroll
        for (int st = 0; st < C::KSTEPS; ++st) {
            const int row = wave_active ? q0 + l31 : 0;
            qf[st] = *reinterpret_cast<const f32x4 *>(sQ + row * C::LDF + st * 8 + 4 * half);
            dof[st] = *reinterpret_cast<const f32x4 *>(sdO + row * C::LDF + st * 8 + 4 * half);
        }

        for (int kc0 = 0; kc0 < S; kc0 += C::KC) {
            __syncthreads();
            for (int i = tid; i < C::KC * F4; i += 256) {
                const int row = i / F4, c4 = i - row * F4;
                const int key = kc0 + row;
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
                if (key < S && c4 * 4 < HD) {
                    a = *reinterpret_cast<const f32x4 *>(kb + (long)key * ldkv + c4 * 4);
                    d = *reinterpret_cast<const f32x4 *>(vb + (long)key * ldkv + c4 * 4);
                }
                *reinterpret_cast<f32x4 *>(sK + row * C::LDF + c4 * 4) = a;
                *reinterpret_cast<f32x4 *>(sV + row * C::LDF + c4 * 4) = d;
            }
            __syncthreads();
            const int tiles = (min(S - kc0, C::KC) + 31) / 32;
            for (int kt = 0; kt < tiles; ++kt) {
                f32x16 pT, dsT;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pT[r] = 0.f;
                    dsT[r] = 0.f;
                }
                if (wave_active) {
                    const float *kp = sK + (kt * 32 + l31) * C::LDF + 4 * half;
                    const float *vp = sV + (kt * 32 + l31) * C::LDF + 4 * half;
#pragma unroll
                    for (int st = 0; st < C::KSTEPS; ++st) {
                        const f32x4 kf = *reinterpret_cast<const f32x4 *>(kp + st * 8);
                        const f32x4 vf = *reinterpret_cast<const f32x4 *>(vp + st * 8);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            pT = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[st][j], pT, 0, 0, 0);     // S^T
                            dsT = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[j], dof[st][j], dsT, 0, 0, 0);  // dP^T
                        }
                    }
                    f32x4 dm[4];
                    if constexpr (DROP) {
                        const unsigned long mrow = ((unsigned long)b * heads + h) * Tq + (q_ok ? qi : 0);
                        const unsigned long wq = (unsigned long)((S + 3) >> 2);
#pragma unroll
                        for (int g = 0; g < 4; ++g) dm[g] = dropout_quad(da, mrow * wq + (unsigned long)((kc0 + kt * 32 + 8 * g + 4 * half) >> 2));
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kc0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const float p = (key < S && q_ok) ? exp2f(pT[r] * sl2e - lse) : 0.f;
                        if constexpr (DROP) {
                            const float mk = dm[r >> 2][r & 3];
                            pT[r] = p * mk;                                  // P_d: what dV contracts
                            dsT[r] = p * (dsT[r] * mk - delta) * scale;
                        } else {
                            pT[r] = p;
                            dsT[r] = p * (dsT[r] - delta) * scale;
                        }
                    }
                    // dQ^T += K^T dS^T : A = K^T[feature l31][key] read as K[key][feature]
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int krow = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
                        for (int ft = 0; ft < C::FT; ++ft) {
                            const float a = sK[krow * C::LDF + ft * 32 + l31];
                            dqT[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dsT[r], dqT[ft], 0, 0, 0);
                        }
                    }
                }
                // P^T / dS^T tiles -> LDS [32 keys][QP queries]
                if (q0 < C::QP) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int krow = (r & 3) + 8 * (r >> 2) + 4 * half;
                        sP[krow * C::LDP + q0 + l31] = pT[r];
                        sdS[krow * C::LDP + q0 + l31] = dsT[r];
                    }
                }
                __syncthreads();
                // dV tile (32 keys x 32 features) = P^T dO ; dK tile = dS^T Q ; 2*FT jobs over 4 waves
                for (int job = wave; job < 2 * C::FT; job += 4) {
                    const bool is_dk = job >= C::FT;
                    const int ft = is_dk ? job - C::FT : job;
                    const float *aT = is_dk ? sdS : sP;
                    const float *bM = is_dk ? sQ : sdO;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
                    for (int kq = 0; kq < C::QP; kq += 8) {
                        const f32x4 af = *reinterpret_cast<const f32x4 *>(aT + l31 * C::LDP + kq + 4 * half);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float bf = bM[(kq + 4 * half + j) * C::LDF + ft * 32 + l31];
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf, acc, 0, 0, 0);
                        }
                    }
                    float *dst = is_dk ? dkb : dvb;
                    const int f = ft * 32 + l31;
                    if (f < HD) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = kc0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (key < S) {
                                float *pd = dst + (long)key * lddkv + f;
                                // one workgroup owns this (sample, head): later query passes add in program order
                                *pd = (qpass == 0) ? acc[r] : *pd + acc[r];
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (q_ok) {
            float *op = dqb + (long)qi * lddq;
#pragma unroll
            for (int ft = 0; ft < C::FT; ++ft)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = ft * 32 + 8 * g + 4 * half;
                    if (f < HD) {
                        f32x4 t = {dqT[ft][4 * g], dqT[ft][4 * g + 1], dqT[ft][4 * g + 2], dqT[ft][4 * g + 3]};
                        *reinterpret_cast<f32x4 *>(op + f) = t;
                    }
                }
        }
    }
}


// ======================================================================================
// Attention backward on the fp16 pipe (head dim 64, Tq <= 128, S <= 128): one workgroup per (sample, head).
// Every product is three v_mfma_f32_32x32x16_f16 on operands split into fp16 hi + lo; the fp32-MFMA kernel
// above spends 41 k of its 55 k cycles per wave in 64-cycle fp32 MFMAs and holds one workgroup per CU.
// Both orientations of the score tile are computed, so that every later product takes its B operand straight from the
// accumulator (contraction index = accumulator rows) and nothing but the four input images goes through LDS:
//   A (a wave's 32 queries on the lanes):  S^T = K Q^T, dP^T = V dO^T, dS^T -> dQ^T += K^T dS^T      (K^T by transposing reads)
//   B (a wave's 32 keys on the lanes):     S = Q K^T,  dP = dO V^T,   P, dS -> dV^T += dO^T P, dK^T += Q^T dS
// Q, K, V, dO sit in LDS as row-major hi | lo planes (272-byte rows: conflict-free 16-byte row reads); the operands that need
// "8 consecutive rows of one column" come from ds_read_b64_tr_b16.  Scales: Q, K, V fixed 2^3 (as the forward), dO one power
// of two per workgroup (abs-max), P 2^10, dS one power of two per lane = per query (A) / per key (B) - constant over the
// contraction, as a scale must be.  With dropout, pass A leaves its masks in LDS as bits for pass B (one Philox call per
// 4 consecutive keys of a query; in B those sit on 4 different lanes).
// Measured at B = 256, T = S = 100 (tools/exp/attbwd_time.py): 160.6 -> 115 us per call with all four images resident (140 KB
// of LDS = ONE workgroup per CU: of the 115, staging the five arrays was 34 us with nothing to overlap it, pass A 32 us, pass B
// 51 us - the VALU work around the 336 MFMAs per wave is about as long as the MFMAs themselves).
// Only TWO images are resident at a time now (72 KB, two workgroups per CU, 256 registers): a pass needs the other two images
// only as "this lane's own row" fragments, which live in registers.  Q, dO are staged first and every lane takes its row
// fragments; K, V replace them for pass A; before pass B every lane takes its own K, V row fragments and the waves write Q, dO
// back from the fragments they kept.  An aliasing ablation had promised 116 -> 70 us; measured: see NOTEBOOK.md 5.10.
// ======================================================================================
constexpr int AB_P = 136;                 // halfs per image row: hi[64] | lo[64] | 8 pad
constexpr float AB_QKV = 8.0f, AB_PS = 1024.0f;
constexpr size_t AB_LDS = (size_t)2 * 128 * AB_P * sizeof(f16) + 3 * 128 * sizeof(float) + 128 * 4 * sizeof(unsigned);

__device__ __forceinline__ f16x8 ab_tr(const f16 *plane, int r16, int col0, int lane) {
    // 8 rows {r16 + 4 half + 0..3, r16 + 8 + 4 half + 0..3} (the k order of an accumulator used as the B operand) of column
    // col0 + (lane & 31)
    const int q = (lane & 15) >> 2, p4 = (lane & 3) * 4, gc = ((lane >> 4) & 1) * 16, half = lane >> 5;
    const f16 *a = plane + (r16 + 4 * half + q) * AB_P + col0 + gc + p4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a + 8 * AB_P));
    const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
    return __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
}

// acc (+)= A B with A = (ah, al), B = (bh, bl): lo.hi, hi.lo, hi.hi
__device__ __forceinline__ f32x16 ab_mfma3(const f16x8 &ah, const f16x8 &al, const f16x8 &bh, const f16x8 &bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}

__device__ __forceinline__ float ab_pow2_scale(float m) { return f16_scale_from_bits(__builtin_bit_cast(unsigned, m)); }

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attention_bwd16_kernel(const float *__restrict__ q, int ldq, const float *__restrict__ k,
                                                               const float *__restrict__ v, int ldkv, const float *__restrict__ o, int ldo,
                                                               const float *__restrict__ dO, int lddo, const float *__restrict__ lse2,
                                                               float *dq, int lddq, float *dk, float *dv, int lddkv, int Tq, int S,
                                                               int heads, float scale, DropoutArgs da) {
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) f16 ab_smem[];
    f16 *img0 = ab_smem, *img1 = img0 + 128 * AB_P;   // {Q, dO} -> {K, V} (pass A) -> {Q, dO} (pass B)
    f16 *sQ = img0, *sdO = img1, *sK = img0, *sV = img1;
    float *sLse = reinterpret_cast<float *>(img1 + 128 * AB_P), *sDelta = sLse + 128, *sRed = sDelta + 128;
    unsigned *sMask = reinterpret_cast<unsigned *>(sRed + 128);   // [128 queries][4 words]: bit = key kept
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const float *qb = q + (long)b * Tq * ldq + h * HD, *ob = o + (long)b * Tq * ldo + h * HD, *dob = dO + (long)b * Tq * lddo + h * HD;
    const float *kb = k + (long)b * S * ldkv + h * HD, *vb = v + (long)b * S * ldkv + h * HD;
    const float sl2e = scale * 1.44269504088896340736f;

    // ---- stage: 128 rows x 16 pieces of 4 floats per image; a row's 16 pieces sit in one DPP row of 16 lanes ----
    // every load of the workgroup is in flight before the first value is used: 40 x 16 bytes per thread, one HBM round trip
    // (with the delta reduction inside the load loop the compiler waited per iteration: 8 round trips, 92 us per launch even
    // for the 11-key cross-attention)
    f32x4 qv[8], kv[8], vv[8], dov[8], ovv[8];
    float dmax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i, row = idx >> 4, c4 = (idx & 15) * 4;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        qv[i] = row < Tq ? *reinterpret_cast<const f32x4 *>(qb + (long)row * ldq + c4) : z;
        dov[i] = row < Tq ? *reinterpret_cast<const f32x4 *>(dob + (long)row * lddo + c4) : z;
        ovv[i] = row < Tq ? *reinterpret_cast<const f32x4 *>(ob + (long)row * ldo + c4) : z;
        kv[i] = row < S ? *reinterpret_cast<const f32x4 *>(kb + (long)row * ldkv + c4) : z;
        vv[i] = row < S ? *reinterpret_cast<const f32x4 *>(vb + (long)row * ldkv + c4) : z;
    }
    float lsev[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (tid + 256 * i) >> 4;
        lsev[i] = ((lane & 15) == 0 && row < Tq) ? lse2[((long)b * heads + h) * Tq + row] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (tid + 256 * i) >> 4;
        float dl = dov[i][0] * ovv[i][0] + dov[i][1] * ovv[i][1] + dov[i][2] * ovv[i][2] + dov[i][3] * ovv[i][3];
        dl = row16_sum(dl);                       // delta_q = sum_f dO O
        if ((lane & 15) == 0) {
            sDelta[row] = dl;
            sLse[row] = lsev[i];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) dmax = fmaxf(dmax, fabsf(dov[i][e]));
    }
    dmax = wave_max(dmax);
    if (lane == 0) sRed[wave] = dmax;
    __syncthreads();
    const float s_do = ab_pow2_scale(fmaxf(fmaxf(sRed[0], sRed[1]), fmaxf(sRed[2], sRed[3])));
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // Q, dO first: every lane takes its own row out of them below
        const int idx = tid + 256 * i, row = idx >> 4, c4 = (idx & 15) * 4;
        f16x4 hh, ll;
        f16_split4(qv[i], AB_QKV, hh, ll);
        *reinterpret_cast<f16x4 *>(sQ + row * AB_P + c4) = hh;
        *reinterpret_cast<f16x4 *>(sQ + row * AB_P + HD + c4) = ll;
        f16_split4(dov[i], s_do, hh, ll);
        *reinterpret_cast<f16x4 *>(sdO + row * AB_P + c4) = hh;
        *reinterpret_cast<f16x4 *>(sdO + row * AB_P + HD + c4) = ll;
    }
    if (DROP)
        for (int i = tid; i < 128 * 4; i += 256) sMask[i] = 0u;
    __syncthreads();
    const int mine = wave * 32 + (lane & 31);            // this lane's query (pass A) / key (pass B)
    f16x8 qh[4], ql[4], doh[4], dol[4];                  // this lane's own Q / dO row: B operands of pass A, and what pass B's images
    {                                                    // are rebuilt from
        const f16 *pq = sQ + mine * AB_P + 8 * (lane >> 5), *pd = sdO + mine * AB_P + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qh[ks] = *reinterpret_cast<const f16x8 *>(pq + ks * 16);
            ql[ks] = *reinterpret_cast<const f16x8 *>(pq + HD + ks * 16);
            doh[ks] = *reinterpret_cast<const f16x8 *>(pd + ks * 16);
            dol[ks] = *reinterpret_cast<const f16x8 *>(pd + HD + ks * 16);
        }
    }
    __syncthreads();   // every lane has its row: K, V take the two images over
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 256 * i, row = idx >> 4, c4 = (idx & 15) * 4;
        f16x4 hh, ll;
        f16_split4(kv[i], AB_QKV, hh, ll);
        *reinterpret_cast<f16x4 *>(sK + row * AB_P + c4) = hh;
        *reinterpret_cast<f16x4 *>(sK + row * AB_P + HD + c4) = ll;
        f16_split4(vv[i], AB_QKV, hh, ll);
        *reinterpret_cast<f16x4 *>(sV + row * AB_P + c4) = hh;
        *reinterpret_cast<f16x4 *>(sV + row * AB_P + HD + c4) = ll;
    }
    __syncthreads();

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float c_s = sl2e / (AB_QKV * AB_QKV);          // raw score accumulator -> log2-domain score
    const float c_dp = 1.0f / (AB_QKV * s_do);           // raw dP accumulator -> dP
    const int n_kt = (S + 31) / 32, n_qt = (Tq + 31) / 32;

    // fragments of this lane's own row of two images: B operands (k = 8 half + j of k-step ks)
    auto row_frags = [&](const f16 *img, f16x8 (&fh)[4], f16x8 (&fl)[4]) {
        const f16 *p = img + mine * AB_P + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fh[ks] = *reinterpret_cast<const f16x8 *>(p + ks * 16);
            fl[ks] = *reinterpret_cast<const f16x8 *>(p + HD + ks * 16);
        }
    };
    // two 32 x 32 tiles at once: rows [t*32, t*32+32) of images A0 / A1 (lanes = their rows) times this lane's own rows of the
    // B images.  All 16 fragment reads are issued before the first MFMA (one LDS latency per tile pair: with one wave per
    // SIMD nothing else hides it)
    auto tile2 = [&](const f16 *imgA0, const f16 *imgA1, int t, const f16x8 (&b0h)[4], const f16x8 (&b0l)[4], const f16x8 (&b1h)[4],
                     const f16x8 (&b1l)[4], f32x16 &r0, f32x16 &r1) {
        const f16 *p0 = imgA0 + (t * 32 + l31) * AB_P + 8 * half, *p1 = imgA1 + (t * 32 + l31) * AB_P + 8 * half;
        f16x8 a0h[4], a0l[4], a1h[4], a1l[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a0h[ks] = *reinterpret_cast<const f16x8 *>(p0 + ks * 16);
            a0l[ks] = *reinterpret_cast<const f16x8 *>(p0 + HD + ks * 16);
            a1h[ks] = *reinterpret_cast<const f16x8 *>(p1 + ks * 16);
            a1l[ks] = *reinterpret_cast<const f16x8 *>(p1 + HD + ks * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        r0 = zero16;
        r1 = zero16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            r0 = ab_mfma3(a0h[ks], a0l[ks], b0h[ks], b0l[ks], r0);
            r1 = ab_mfma3(a1h[ks], a1l[ks], b1h[ks], b1l[ks], r1);
        }
    };
    auto split8 = [&](const f32x16 &t, int j2, float sc, f16x8 &hi, f16x8 &lo) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = t[8 * j2 + e] * sc;
            hi[e] = (f16)x;
            lo[e] = (f16)(x - (float)hi[e]);
        }
    };
    // out^T (64 features x this wave's 32 columns) += img^T B over the 16-row groups of `tiles` score tiles; the transposing
    // reads of a whole tile (2 groups x 2 feature tiles x 2 planes) are issued before its MFMAs
    auto contract = [&](const f16 *img, const f32x16 (&t)[4], int tiles, float sc, f32x16 (&out)[2]) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            if (tt >= tiles) break;
            f16x8 ah[2][2], al[2][2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    ah[j2][ft] = ab_tr(img, tt * 32 + j2 * 16, ft * 32, lane);
                    al[j2][ft] = ab_tr(img + HD, tt * 32 + j2 * 16, ft * 32, lane);
                }
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                f16x8 bh, bl;
                split8(t[tt], j2, sc, bh, bl);
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) out[ft] = ab_mfma3(ah[j2][ft], al[j2][ft], bh, bl, out[ft]);
            }
        }
    };
    auto store_T = [&](float *dst, int ld, int rows, const f32x16 (&acc)[2], float un) {
        if (mine < rows) {   // lane = row of the output, registers 4g..4g+3 = 4 consecutive features
            float *op = dst + (long)mine * ld;
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 t4 = {acc[ft][4 * g] * un, acc[ft][4 * g + 1] * un, acc[ft][4 * g + 2] * un, acc[ft][4 * g + 3] * un};
                    *reinterpret_cast<f32x4 *>(op + ft * 32 + 8 * g + 4 * half) = t4;
                }
        }
    };

    // ================= pass A: this wave's queries; dQ =================
    if (wave * 32 < Tq) {
        const bool q_ok = mine < Tq;
        const float lse = sLse[q_ok ? mine : 0], delta = sDelta[q_ok ? mine : 0];
        f32x16 ds[4];
        float dsmax = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt >= n_kt) { ds[kt] = zero16; continue; }
            f32x16 sT, dpT;
            tile2(sK, sV, kt, qh, ql, doh, dol, sT, dpT);
            f32x4 dm[4];
            if constexpr (DROP) {
                const unsigned long mrow = ((unsigned long)b * heads + h) * Tq + (q_ok ? mine : 0);
                const unsigned long wq = (unsigned long)((S + 3) >> 2);
#pragma unroll
                for (int g = 0; g < 4; ++g) dm[g] = dropout_quad(da, mrow * wq + (unsigned long)((kt * 32 + 8 * g + 4 * half) >> 2));
                if (q_ok) {   // leave the mask bits of (query mine, keys kt*32 ..) for pass B
                    unsigned bits = 0u;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (dm[r >> 2][r & 3] != 0.f) bits |= 1u << ((r & 3) + 8 * (r >> 2) + 4 * half);
                    atomicOr(&sMask[mine * 4 + kt], bits);   // the two halves of a query own disjoint bits
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float p = (key < S && q_ok) ? exp2f(sT[r] * c_s - lse) : 0.f;
                float dp = dpT[r] * c_dp;
                if constexpr (DROP) dp *= dm[r >> 2][r & 3];
                const float d = p * (dp - delta) * scale;
                ds[kt][r] = d;
                dsmax = fmaxf(dsmax, fabsf(d));
            }
        }
        dsmax = fmaxf(dsmax, __shfl_xor(dsmax, 32, 64));
        const float s_q = ab_pow2_scale(dsmax);       // one scale per query: constant over the keys it is contracted with
        f32x16 dqT[2] = {zero16, zero16};
        contract(sK, ds, n_kt, s_q, dqT);
        store_T(dq + (long)b * Tq * lddq + h * HD, lddq, Tq, dqT, 1.0f / (AB_QKV * s_q));
    }
    // ---- the images change hands: every lane takes its own K / V row, then Q / dO come back from the kept fragments ----
    f16x8 kh[4], kl[4], vh[4], vl[4];
    row_frags(sK, kh, kl);
    row_frags(sV, vh, vl);
    __syncthreads();   // pass A is done with K, V everywhere (and its mask bits are complete)
    {
        f16 *pq = sQ + mine * AB_P + 8 * half, *pd = sdO + mine * AB_P + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            *reinterpret_cast<f16x8 *>(pq + ks * 16) = qh[ks];
            *reinterpret_cast<f16x8 *>(pq + HD + ks * 16) = ql[ks];
            *reinterpret_cast<f16x8 *>(pd + ks * 16) = doh[ks];
            *reinterpret_cast<f16x8 *>(pd + HD + ks * 16) = dol[ks];
        }
    }
    __syncthreads();

    // ================= pass B: this wave's keys; dK, dV =================
    if (wave * 32 < S) {
        const bool k_ok = mine < S;
        f32x16 pd[4], ds[4];
        float dsmax = 0.f;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            if (qt >= n_qt) { pd[qt] = zero16; ds[qt] = zero16; continue; }
            f32x16 sN, dpN;                              // rows = queries, lane = key
            tile2(sQ, sdO, qt, kh, kl, vh, vl, sN, dpN);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const bool ok = k_ok && qi < Tq;
                const float p = ok ? exp2f(sN[r] * c_s - sLse[qi]) : 0.f;
                float dp = dpN[r] * c_dp, pm = p;
                if constexpr (DROP) {
                    const float m = ((sMask[qi * 4 + (mine >> 5)] >> (mine & 31)) & 1u) ? da.scale : 0.f;
                    dp *= m;
                    pm *= m;
                }
                const float d = p * (dp - sDelta[qi]) * scale;
                pd[qt][r] = pm;
                ds[qt][r] = d;
                dsmax = fmaxf(dsmax, fabsf(d));
            }
        }
        dsmax = fmaxf(dsmax, __shfl_xor(dsmax, 32, 64));
        const float s_k = ab_pow2_scale(dsmax);       // one scale per key
        f32x16 dvT[2] = {zero16, zero16}, dkT[2] = {zero16, zero16};
        contract(sdO, pd, n_qt, AB_PS, dvT);
        contract(sQ, ds, n_qt, s_k, dkT);
        store_T(dv + (long)b * S * lddkv + h * HD, lddkv, S, dvT, 1.0f / (s_do * AB_PS));
        store_T(dk + (long)b * S * lddkv + h * HD, lddkv, S, dkT, 1.0f / (AB_QKV * s_k));
    }
}

extern "C" int sd_op_attention_bwd(const float *q, int ldq, const float *k, const float *v, int ldkv, const float *o,
                                   int ldo, const float *dO, int lddo, const float *lse2, float *dq, int lddq, float *dk,
                                   float *dv, int lddkv, int B, int Tq, int S, int d, int heads, void *stream) {
    return sd_op_attention_bwd_dropout(q, ldq, k, v, ldkv, o, ldo, dO, lddo, lse2, dq, lddq, dk, dv, lddkv, B, Tq, S, d, heads, 0.f, 0, 0,
                                       stream);
}

extern "C" int sd_op_attention_bwd_dropout(const float *q, int ldq, const float *k, const float *v, int ldkv, const float *o,
                                           int ldo, const float *dO, int lddo, const float *lse2, float *dq, int lddq, float *dk,
                                           float *dv, int lddkv, int B, int Tq, int S, int d, int heads, float p, uint64_t seed,
                                           uint64_t site, void *stream) {
    if (int rc = dropout_args_ok(p, "sd_op_attention_bwd_dropout: p must be in [0, 1)")) return rc;
    const DropoutArgs da = make_dropout(p, seed, site);
    if (!q || !k || !v || !o || !dO || !lse2 || !dq || !dk || !dv || B <= 0 || Tq <= 0 || S <= 0 || heads <= 0)
        return fail(SD_E_BADARG, "sd_op_attention_bwd: bad argument");
    if (d % heads != 0) return fail(SD_E_BADDIM, "attention: d not divisible by heads");
    const int hd = d / heads;
    const float scale = 1.0f / sqrtf((float)hd);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(SD_KCLASS_ATTENTION, s);
    dim3 grid(B * heads), block(256);
    {   // head dim 64, <= 128 queries and keys, 16-byte aligned rows: the fp16-pipe kernel; anything else: the fp32-MFMA one
        auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        const bool lds_ok = ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && lddo % 4 == 0 && lddq % 4 == 0 && lddkv % 4 == 0;
        // (up to 32 keys - the cross-attention over 11 memory rows - only one wave owns keys in pass B; with one workgroup per CU
        // the fp32 kernel was faster there, 60 vs 83 us at B = 256; with two it is 54 vs 62)
        if (hd == 64 && Tq <= 128 && S <= 128 && lds_ok && al16(q) && al16(k) && al16(v) && al16(o) && al16(dO) && al16(dq) && al16(dk) &&
            al16(dv)) {
            static DevFlag attr_set;       
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void *)attention_bwd16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AB_LDS);
                (void)hipFuncSetAttribute((const void *)attention_bwd16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AB_LDS);
                attr_set = true;
            }
            if (da.thresh) SD_LAUNCH(attention_bwd16_kernel<true>, grid, block, AB_LDS, s, q, ldq, k, v, ldkv, o, ldo, dO, lddo, lse2, dq, lddq, dk, dv, lddkv, Tq, S, heads, scale, da);
            else SD_LAUNCH(attention_bwd16_kernel<false>, grid, block, AB_LDS, s, q, ldq, k, v, ldkv, o, ldo, dO, lddo, lse2, dq, lddq, dk, dv, lddkv, Tq, S, heads, scale, da);
            SD_CHECK_LAUNCH("attention_bwd16_kernel");
            return 0;
        }
    }
#define SD_ATTNB(HD_)                                                                                            \
    do {                                                                                                         \
        auto kfn = da.thresh ? attention_bwd_kernel<HD_, true> : attention_bwd_kernel<HD_, false>;               \
        const size_t lds = AttnBwdCfg<HD_>::LDS_BYTES;                                                           \
        static DevFlag attr_set[2];                                                                                \
        if (lds > 64 * 1024 && !attr_set[da.thresh ? 1 : 0]) {                                                   \
            (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            attr_set[da.thresh ? 1 : 0] = true;                                                                  \
        }                                                                                                        \
        SD_LAUNCH(kfn, grid, block, lds, s, q, ldq, k, v, ldkv, o, ldo, dO, lddo, lse2, dq, lddq, dk, dv, lddkv, Tq, S, \
                  heads, scale, da);                                                                             \
    } while (0)
    switch (hd) {
        case 16: SD_ATTNB(16); break;
        case 32: SD_ATTNB(32); break;
        case 64: SD_ATTNB(64); break;
        case 128: SD_ATTNB(128); break;
        default: return fail(SD_E_BADDIM, "attention: head dim must be 16, 32, 64 or 128");
    }
#undef SD_ATTNB
    SD_CHECK_LAUNCH("attention_bwd_kernel");
    return 0;
}
