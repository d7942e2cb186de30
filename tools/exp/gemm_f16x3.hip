// Prototype: fp32-grade GEMM from 3 fp16 MFMAs per product: a = ah + al, b = bh + bl (fp16 pairs of the power-of-two
// pre-scaled operands, 22 mantissa bits), a.b ~ ah.bh + ah.bl + al.bh accumulated in fp32.
// out[R,N] = A[R,256] W[N,256]^T.  The A panel is split ONCE into two fp16 planes in LDS; W is pre-split (2 planes).
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/gemm_f16x3.hip -o tools/exp/gemm_f16x3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", e, __FILE__, __LINE__); exit(1);} } while (0)
#ifndef WM_WAVES
#define WM_WAVES 1
#endif
#ifndef FRAG
#define FRAG 0
#endif
#ifndef NOB
#define NOB 0
#endif
constexpr int D = 256, BM = 64 * WM_WAVES, NT = 256 * WM_WAVES, LDH = D + 8;   // halfs per LDS row (528 B: 16 rows hit 16 distinct 4-bank groups)
#ifndef RING
#define RING 3
#endif
#ifndef STORE
#define STORE 1
#endif

// Wp: [2 planes][N][256] fp16
__global__ __launch_bounds__(NT, 2 / WM_WAVES) void gemm_f16x3_kernel(const float* __restrict__ A, const _Float16* __restrict__ Wp,
                                                            float* __restrict__ out, int R, int N, float a_scale, float out_scale) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sH[];
    _Float16* sL = sH + BM * LDH;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wm = tid >> 8;
    const long r0 = (long)blockIdx.x * BM;
    {
        f32x4 v[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const int i = tid + b * NT, row = i / (D / 4), c4 = i % (D / 4);
            v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (r0 + row < R) v[b] = *reinterpret_cast<const f32x4*>(A + (r0 + row) * D + c4 * 4);
        }
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const int i = tid + b * NT, row = i / (D / 4), c4 = i % (D / 4);
            f16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[b][e] * a_scale;
                h[e] = (_Float16)x;
                l[e] = (_Float16)(x - (float)h[e]);
            }
            *reinterpret_cast<f16x4*>(sH + row * LDH + c4 * 4) = h;
            *reinterpret_cast<f16x4*>(sL + row * LDH + c4 * 4) = l;
        }
    }
    __syncthreads();
    const int l31 = lane & 31, half = lane >> 5;
    const _Float16* aH = sH + (wm * 64 + l31) * LDH + 8 * half;
    const _Float16* aL = sL + (wm * 64 + l31) * LDH + 8 * half;
    const size_t plane = (size_t)N * D;
    constexpr int NK = D / 16;
    for (int n0 = 0; n0 < N; n0 += D) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#if FRAG
        // fragment-major: [pass][wave][ks][tn][plane][lane][8 halfs] -> every wave-load reads 1 KiB contiguous
        const _Float16* wBase = Wp + ((size_t)(n0 / D) * 4 + wave) * (NK * 2 * 2 * 512) + lane * 8;
#define WLOAD(tn, pl, ks) (*reinterpret_cast<const f16x8*>(wBase + (((ks) * 2 + (tn)) * 2 + (pl)) * 512))
#else
        const _Float16* wBase = Wp + (size_t)(n0 + wave * 64 + l31) * D + 8 * half;
#define WLOAD(tn, pl, ks) (*reinterpret_cast<const f16x8*>(wBase + (pl) * plane + (size_t)(tn) * 32 * D + (ks) * 16))
#endif
        f16x8 bq[RING][2][2];  // [slot][tn][plane]
#pragma unroll
        for (int s = 0; s < RING - 1; ++s)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    bq[s][tn][pl] = WLOAD(tn, pl, s);
        f16x8 ap[2][2][2];  // [buffer][tm][plane]
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            ap[0][tm][0] = *reinterpret_cast<const f16x8*>(aH + tm * 32 * LDH);
            ap[0][tm][1] = *reinterpret_cast<const f16x8*>(aL + tm * 32 * LDH);
        }
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int cur = ks % RING, fill = (ks + RING - 1) % RING;
            if (!NOB && ks + RING - 1 < NK) {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        bq[fill][tn][pl] = WLOAD(tn, pl, ks + RING - 1);
            }
            if (ks + 1 < NK) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    ap[(ks + 1) & 1][tm][0] = *reinterpret_cast<const f16x8*>(aH + tm * 32 * LDH + (ks + 1) * 16);
                    ap[(ks + 1) & 1][tm][1] = *reinterpret_cast<const f16x8*>(aL + tm * 32 * LDH + (ks + 1) * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // small terms first: al.bh, ah.bl, then ah.bh
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[ks & 1][tm][TA[t]], bq[cur][tn][TB[t]], acc[tm][tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STORE) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int col = n0 + wave * 64 + tn * 32 + l31;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long row = r0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (row < R) __builtin_nontemporal_store(acc[tm][tn][r] * out_scale, out + row * N + col);
                    }
            }
        } else {
            float s = 0.f;
            for (int tn = 0; tn < 2; ++tn) for (int tm = 0; tm < 2; ++tm) for (int r = 0; r < 16; ++r) s += acc[tm][tn][r];
            if (s == 1.2345f) out[0] = s;
        }
    }
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 409600, N = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<float> hA((size_t)R * D), hW((size_t)N * D);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) * 2 - 1) / 16;
    const float w_scale = 4096.f, a_scale = 256.f;   // powers of two: max |w| 1/16 -> 256, max |a| 1 -> 256
    std::vector<_Float16> hP((size_t)2 * N * D);
    for (size_t i = 0; i < hW.size(); ++i) {
        const float x = hW[i] * w_scale;
        const _Float16 h = (_Float16)x;
        hP[i] = h;
        hP[hW.size() + i] = (_Float16)(x - (float)h);
    }
#if FRAG
    {
        std::vector<_Float16> hF(hP.size());
        const int NKh = D / 16;
        for (int pass = 0; pass < N / D; ++pass) for (int wv = 0; wv < 4; ++wv) for (int ks = 0; ks < NKh; ++ks)
            for (int tn = 0; tn < 2; ++tn) for (int pl = 0; pl < 2; ++pl) for (int ln = 0; ln < 64; ++ln) for (int e = 0; e < 8; ++e) {
                const size_t n = (size_t)pass * D + wv * 64 + tn * 32 + (ln & 31), k = ks * 16 + 8 * (ln >> 5) + e;
                hF[((((((size_t)pass * 4 + wv) * NKh + ks) * 2 + tn) * 2 + pl) * 64 + ln) * 8 + e] = hP[pl * hW.size() + n * D + k];
            }
        hP = hF;
    }
#endif
    float *A, *o; _Float16* Wp;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&Wp, hP.size() * 2)); CK(hipMalloc(&o, (size_t)R * N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wp, hP.data(), hP.size() * 2, hipMemcpyHostToDevice));
    size_t lds = (size_t)2 * BM * LDH * 2;
    CK(hipFuncSetAttribute((const void*)gemm_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const float out_scale = 1.0f / (w_scale * a_scale);
    hipLaunchKernelGGL(gemm_f16x3_kernel, dim3((R + BM - 1) / BM), dim3(NT), lds, 0, A, Wp, o, R, N, a_scale, out_scale);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(gemm_f16x3_kernel, dim3((R + BM - 1) / BM), dim3(NT), lds, 0, A, Wp, o, R, N, a_scale, out_scale);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("f16x3 FRAG=%d WM=%d NOB=%d RING=%d STORE=%d R=%d N=%d: %.3f ms  %.1f TF fp32-equivalent\n", FRAG, WM_WAVES, NOB, RING, STORE, R, N, ms, 2.0 * R * N * D / ms / 1e9);
    if (STORE) {
        std::vector<float> ho((size_t)64 * N); CK(hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0, num32 = 0;
        for (int r = 0; r < 64; ++r) for (int n = 0; n < N; ++n) {
            double s = 0; float f = 0;
            for (int k = 0; k < D; ++k) { s += (double)hA[(size_t)r * D + k] * hW[(size_t)n * D + k]; f = fmaf(hA[(size_t)r * D + k], hW[(size_t)n * D + k], f); }
            const double d = ho[(size_t)r * N + n] - s; num += d * d; den += s * s; num32 += (f - s) * (f - s);
        }
        printf("rel L2 error vs double: f16x3 %.3e   (plain fp32 fma chain %.3e)\n", sqrt(num / den), sqrt(num32 / den));
    }
    return 0;
}
