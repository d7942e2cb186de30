// Experimental harness for the fp32-MFMA panel GEMM inner loop (not product code).
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/gemm_exp.hip -o /tmp/gemm_exp && /tmp/gemm_exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", e, __FILE__, __LINE__); exit(1);} } while (0)

constexpr int D = 256, BM = 64, LDA = D + 4;

// VARIANT 0: baseline (as in product v1)   1: sched_barrier pinned prefetch depth 1
// 2: prefetch depth 2                       3: depth 1 + A frags double-buffered
template <int VARIANT>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out, int R, int N) {
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
    const float* aBase = sA + l31 * LDA + 4 * half;
    for (int n0 = 0; n0 < N; n0 += D) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const float* wBase = W + (long)(n0 + wave * 64 + l31) * D + 4 * half;
        if constexpr (VARIANT == 0) {
            f32x4 bcur[2], bnext[2];
            for (int tn = 0; tn < 2; ++tn) bcur[tn] = *reinterpret_cast<const f32x4*>(wBase + (long)tn * 32 * D);
#pragma unroll 4
            for (int k0 = 0; k0 < D; k0 += 8) {
                const int kn = (k0 + 8 < D) ? k0 + 8 : k0;
                for (int tn = 0; tn < 2; ++tn) bnext[tn] = *reinterpret_cast<const f32x4*>(wBase + (long)tn * 32 * D + kn);
                f32x4 a[2];
                for (int tm = 0; tm < 2; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(aBase + tm * 32 * LDA + k0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], bcur[tn][j], acc[tm][tn], 0, 0, 0);
                for (int tn = 0; tn < 2; ++tn) bcur[tn] = bnext[tn];
            }
        } else {
            constexpr int DEPTH = (VARIANT == 2) ? 2 : 1;
            constexpr int NK = D / 8;
            f32x4 b[DEPTH + 1][2];
            f32x4 a[2][2];
#pragma unroll
            for (int p = 0; p < DEPTH; ++p)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) b[p][tn] = *reinterpret_cast<const f32x4*>(wBase + (long)tn * 32 * D + p * 8);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) a[0][tm] = *reinterpret_cast<const f32x4*>(aBase + tm * 32 * LDA);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int cur = ks % (DEPTH + 1), nxt = (ks + DEPTH) % (DEPTH + 1);
                if (ks + DEPTH < NK) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) b[nxt][tn] = *reinterpret_cast<const f32x4*>(wBase + (long)tn * 32 * D + (ks + DEPTH) * 8);
                }
                if (ks + 1 < NK) {
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) a[(ks + 1) & 1][tm] = *reinterpret_cast<const f32x4*>(aBase + tm * 32 * LDA + (ks + 1) * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks & 1][tm][j], b[cur][tn][j], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
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

// raw MFMA rate: 4 accumulators per wave, operands in registers, NW waves per SIMD
__global__ __launch_bounds__(256) void mfma_only(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j & 3], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
float run(const float* A, const float* W, float* out, int R, int N, int iters) {
    auto k = gemm_kernel<V>;
    size_t lds = (size_t)BM * LDA * 4;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3((R + BM - 1) / BM), dim3(256), lds, 0, A, W, out, R, N);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3((R + BM - 1) / BM), dim3(256), lds, 0, A, W, out, R, N);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 409600, N = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<float> hA((size_t)R * D), hW((size_t)N * D);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
    for (auto& v : hW) v = ((rand() / (float)RAND_MAX) * 2 - 1) / 16;
    float *A, *W, *o0, *o1;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4));
    CK(hipMalloc(&o0, (size_t)R * N * 4)); CK(hipMalloc(&o1, (size_t)R * N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int wgs : {256, 512, 1024}) {
            const int iters = 4096;
            hipLaunchKernelGGL(mfma_only, dim3(wgs), dim3(256), 0, 0, o0, 16);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(mfma_only, dim3(wgs), dim3(256), 0, 0, o0, iters);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
            double fl = (double)wgs * 4 * iters * 16 * 4096.0;
            printf("mfma_only %d WGs (x4 waves): %.3f ms  %.1f TF\n", wgs, ms, fl / ms / 1e9);
        }
    }
    const double flops = 2.0 * R * N * D;
    float t0 = run<0>(A, W, o0, R, N, 10);
    printf("variant 0: %.3f ms  %.1f TF\n", t0, flops / t0 / 1e9);
    float t;
    t = run<1>(A, W, o1, R, N, 10); printf("variant 1: %.3f ms  %.1f TF\n", t, flops / t / 1e9);
    t = run<2>(A, W, o1, R, N, 10); printf("variant 2: %.3f ms  %.1f TF\n", t, flops / t / 1e9);
    // check variant 2 == variant 0 bitwise (same k order)
    std::vector<float> h0(1 << 16), h1(1 << 16);
    CK(hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost));
    double md = 0; for (size_t i = 0; i < h0.size(); ++i) md = fmax(md, fabs(h0[i] - h1[i]));
    printf("max diff v0 vs v2: %g\n", md);
    return 0;
}
