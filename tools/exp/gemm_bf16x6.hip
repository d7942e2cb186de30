// Prototype: fp32-grade GEMM from 6 bf16 MFMAs per product (a = a1+a2+a3, b = b1+b2+b3 in bf16, terms i+j<=4).
// out[R,N] = A[R,256] W[N,256]^T.  A panel fp32 in LDS, split on the fly; W pre-split into 3 bf16 planes.
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/gemm_bf16x6.hip -o tools/exp/gemm_bf16x6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", e, __FILE__, __LINE__); exit(1);} } while (0)
constexpr int D = 256, BM = 64, LDA = D + 4;

// split 8 fp32 (two f32x4) into three packed-bf16 fragments by truncation: x = x1 + x2 + x3 exactly
__device__ __forceinline__ void split8(const f32x4& lo, const f32x4& hi, u32x4& p1, u32x4& p2, u32x4& p3) {
    float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    unsigned b1[8], b2[8], b3[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned u = __builtin_bit_cast(unsigned, x[i]);
        const float x1 = __builtin_bit_cast(float, u & 0xFFFF0000u);
        const float r = x[i] - x1;
        const unsigned ur = __builtin_bit_cast(unsigned, r);
        const float x2 = __builtin_bit_cast(float, ur & 0xFFFF0000u);
        const float r2 = r - x2;
        b1[i] = u; b2[i] = ur; b3[i] = __builtin_bit_cast(unsigned, r2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p1[i] = __builtin_amdgcn_perm(b1[2 * i + 1], b1[2 * i], 0x07060302u);
        p2[i] = __builtin_amdgcn_perm(b2[2 * i + 1], b2[2 * i], 0x07060302u);
        p3[i] = __builtin_amdgcn_perm(b3[2 * i + 1], b3[2 * i], 0x07060302u);
    }
}

// Wp: [3 planes][N][256] bf16 (as ushort)
__global__ __launch_bounds__(256) void gemm_bf16x6_kernel(const float* __restrict__ A, const unsigned short* __restrict__ Wp, float* __restrict__ out, int R, int N) {
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r0 = (long)blockIdx.x * BM;
    for (int i = tid; i < BM * (D / 4); i += 256) {
        const int row = i / (D / 4), c4 = i % (D / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < R) v = *reinterpret_cast<const f32x4*>(A + (r0 + row) * D + c4 * 4);
        *reinterpret_cast<f32x4*>(sA + row * LDA + c4 * 4) = v;
    }
    __syncthreads();
    const int l31 = lane & 31, half = lane >> 5;
    const float* aBase = sA + l31 * LDA + 8 * half;
    const size_t plane = (size_t)N * D;
    for (int n0 = 0; n0 < N; n0 += D) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const unsigned short* wBase = Wp + (size_t)(n0 + wave * 64 + l31) * D + 8 * half;
        constexpr int NK = D / 16;
        u32x4 bq[2][2][3];  // [buffer][tn][plane]
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bq[0][tn][pl] = *reinterpret_cast<const u32x4*>(wBase + pl * plane + (size_t)tn * 32 * D);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < NK) {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bq[nxt][tn][pl] = *reinterpret_cast<const u32x4*>(wBase + pl * plane + (size_t)tn * 32 * D + (ks + 1) * 16);
            }
            u32x4 ap[2][3];
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(aBase + tm * 32 * LDA + ks * 16);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(aBase + tm * 32 * LDA + ks * 16 + 4);
                split8(lo, hi, ap[tm][0], ap[tm][1], ap[tm][2]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // terms (i,j) with i+j <= 4, small ones first
            constexpr int TI[6] = {2, 1, 0, 1, 0, 0}, TJ[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[tm][TI[t]]), __builtin_bit_cast(bf16x8, bq[cur][tn][TJ[t]]), acc[tm][tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wave * 64 + tn * 32 + l31;
            for (int tm = 0; tm < 2; ++tm)
                for (int r = 0; r < 16; ++r) {
                    const long row = r0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < R) out[row * N + col] = acc[tm][tn][r];
                }
        }
    }
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 409600, N = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<float> hA((size_t)R * D), hW((size_t)N * D);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) * 2 - 1) / 16;
    std::vector<unsigned short> hP((size_t)3 * N * D);
    for (size_t i = 0; i < hW.size(); ++i) {
        float x = hW[i]; unsigned u; memcpy(&u, &x, 4);
        unsigned u1 = u & 0xFFFF0000u; float x1; memcpy(&x1, &u1, 4);
        float r = x - x1; unsigned ur; memcpy(&ur, &r, 4); unsigned u2 = ur & 0xFFFF0000u; float x2; memcpy(&x2, &u2, 4);
        float r2 = r - x2; unsigned u3; memcpy(&u3, &r2, 4);
        hP[i] = u >> 16; hP[hW.size() + i] = ur >> 16; hP[2 * hW.size() + i] = u3 >> 16;
    }
    float *A, *o; unsigned short* Wp;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&Wp, hP.size() * 2)); CK(hipMalloc(&o, (size_t)R * N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wp, hP.data(), hP.size() * 2, hipMemcpyHostToDevice));
    size_t lds = (size_t)BM * LDA * 4;
    CK(hipFuncSetAttribute((const void*)gemm_bf16x6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gemm_bf16x6_kernel, dim3((R + BM - 1) / BM), dim3(256), lds, 0, A, Wp, o, R, N);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(gemm_bf16x6_kernel, dim3((R + BM - 1) / BM), dim3(256), lds, 0, A, Wp, o, R, N);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("bf16x6 R=%d N=%d: %.3f ms  %.1f TF fp32-equivalent\n", R, N, ms, 2.0 * R * N * D / ms / 1e9);
    // accuracy vs double on a sample of rows
    std::vector<float> ho((size_t)64 * N); CK(hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0, maxrel = 0;
    for (int r = 0; r < 64; ++r) for (int n = 0; n < N; ++n) {
        double s = 0; for (int k = 0; k < D; ++k) s += (double)hA[(size_t)r * D + k] * hW[(size_t)n * D + k];
        double d = ho[(size_t)r * N + n] - s; num += d * d; den += s * s; if (fabs(s) > 1e-3) maxrel = fmax(maxrel, fabs(d / s));
    }
    // fp32 fma-chain reference error for comparison
    double num32 = 0;
    for (int r = 0; r < 64; ++r) for (int n = 0; n < N; ++n) {
        double s = 0; float f = 0; for (int k = 0; k < D; ++k) { s += (double)hA[(size_t)r * D + k] * hW[(size_t)n * D + k]; f = fmaf(hA[(size_t)r * D + k], hW[(size_t)n * D + k], f); }
        num32 += (f - s) * (f - s);
    }
    printf("rel L2 error vs double: bf16x6 %.3e   (plain fp32 fma chain %.3e)   max rel %.3e\n", sqrt(num / den), sqrt(num32 / den), maxrel);
    return 0;
}
