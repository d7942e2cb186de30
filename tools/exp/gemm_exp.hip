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
__device__ unsigned g_cu_counter[4096];

template <int VARIANT>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out, int R, int N) {
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r0 = (long)blockIdx.x * BM;
    if constexpr (VARIANT == 8 || VARIANT == 9) {
        __shared__ unsigned s_par;
        if (tid == 0) {
            unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
            unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11));
            unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            unsigned idx = (((xcc & 7) * 8 + se) * 2 + sh) * 16 + cu;
            s_par = atomicAdd(&g_cu_counter[idx], 1u) & 1u;
        }
        __syncthreads();
        if (s_par) __builtin_amdgcn_s_setprio(VARIANT == 9 ? 3 : 1);
    }
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
        const float* wBase = (VARIANT == 6) ? W + (long)(n0 + wave * 64) * D + lane * 4
                                            : W + (long)(n0 + wave * 64 + l31) * D + 4 * half;
        constexpr int KSS = (VARIANT == 6) ? 256 : 8;
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
                for (int tn = 0; tn < 2; ++tn) b[p][tn] = *reinterpret_cast<const f32x4*>(wBase + (long)tn * 32 * D + p * KSS);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) a[0][tm] = *reinterpret_cast<const f32x4*>(aBase + tm * 32 * LDA);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int cur = ks % (DEPTH + 1), nxt = (ks + DEPTH) % (DEPTH + 1);
                if (ks + DEPTH < NK) {
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        if constexpr (VARIANT == 3 || VARIANT == 5) b[nxt][tn] = b[cur][tn];
                        else b[nxt][tn] = *reinterpret_cast<const f32x4*>(wBase + (long)tn * 32 * D + (ks + DEPTH) * KSS);
                    }
                }
                if (ks + 1 < NK) {
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) {
                        if constexpr (VARIANT == 4 || VARIANT == 5) a[(ks + 1) & 1][tm] = a[ks & 1][tm];
                        else a[(ks + 1) & 1][tm] = *reinterpret_cast<const f32x4*>(aBase + tm * 32 * LDA + (ks + 1) * 8);
                    }
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
        if constexpr (VARIANT == 13) {
            const int i4 = lane & 3;
            for (int tn = 0; tn < 2; ++tn)
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float x0 = acc[tm][tn][4 * g], x1 = acc[tm][tn][4 * g + 1], x2 = acc[tm][tn][4 * g + 2], x3 = acc[tm][tn][4 * g + 3];
                        auto dpp = [](float v, int ctrl_is_b1) {
                            int iv = __builtin_bit_cast(int, v);
                            int r = ctrl_is_b1 ? __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, false) : __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xF, 0xF, false);
                            return __builtin_bit_cast(float, r);
                        };
                        const bool o1 = i4 & 1, o2 = i4 & 2;
                        float ta = o1 ? x0 : x1, tb = o1 ? x2 : x3;
                        ta = dpp(ta, 1); tb = dpp(tb, 1);
                        if (o1) { x0 = ta; x2 = tb; } else { x1 = ta; x3 = tb; }
                        float tc = o2 ? x0 : x2, td = o2 ? x1 : x3;
                        tc = dpp(tc, 0); td = dpp(td, 0);
                        if (o2) { x0 = tc; x1 = td; } else { x2 = tc; x3 = td; }
                        const long row = r0 + tm * 32 + 8 * g + 4 * half + i4;
                        const int col = n0 + wave * 64 + tn * 32 + (l31 & ~3);
                        if (row < R) { f32x4 v = {x0, x1, x2, x3}; *reinterpret_cast<f32x4*>(out + row * N + col) = v; }
                    }
        } else {
        const bool do_store = (VARIANT != 7) || (n0 + D >= N) || (R == 12345);
        if (do_store)
        for (int tn = 0; tn < 2; ++tn) {
            const int col = n0 + wave * 64 + tn * 32 + l31;
            for (int tm = 0; tm < 2; ++tm)
                for (int r = 0; r < 16; ++r) {
                    const long row = r0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < R) out[row * N + col] = acc[tm][tn][r];
                }
        }
        else { for (int tn = 0; tn < 2; ++tn) for (int tm = 0; tm < 2; ++tm) asm volatile("" :: "v"(acc[tm][tn])); }
        }
    }
}

// VARIANT 10/11: weight fragments prefetched PF k-steps ahead, CONTINUOUSLY across output passes: the tail of pass p
// loads the head of pass p+1, so those loads are older than pass p's epilogue stores in the in-order vmcnt queue.
template <int PF>
__global__ __launch_bounds__(256) void gemm_cont_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out, int R, int N) {
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
    constexpr int NK = D / 8;
    static_assert(NK % PF == 0, "ring must divide the k-steps");
    f32x4 b[PF][2];
    const float* wLane = W + (long)(wave * 64 + l31) * D + 4 * half;
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) b[p][tn] = *reinterpret_cast<const f32x4*>(wLane + (long)tn * 32 * D + p * 8);
    for (int n0 = 0; n0 < N; n0 += D) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const float* wCur = wLane + (long)n0 * D;
        const float* wNxt = wLane + (long)((n0 + D < N) ? n0 + D : n0) * D;
        f32x4 a[2][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) a[0][tm] = *reinterpret_cast<const f32x4*>(aBase + tm * 32 * LDA);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int slot = ks % PF;
            f32x4 bc[2] = {b[slot][0], b[slot][1]};
            if (ks + 1 < NK) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) a[(ks + 1) & 1][tm] = *reinterpret_cast<const f32x4*>(aBase + tm * 32 * LDA + (ks + 1) * 8);
            }
            // refill this slot with k-step ks+PF of this pass, or with the head of the next pass
            const float* src = (ks + PF < NK) ? wCur + (ks + PF) * 8 : wNxt + (ks + PF - NK) * 8;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) b[slot][tn] = *reinterpret_cast<const f32x4*>(src + (long)tn * 32 * D);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks & 1][tm][j], bc[tn][j], acc[tm][tn], 0, 0, 0);
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

template <int PF>
float run_cont(const float* A, const float* W, float* out, int R, int N, int iters) {
    auto k = gemm_cont_kernel<PF>;
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

// VARIANT 20: 32-row panels, 2 waves per workgroup (each wave 32 rows x 128 columns), 4 workgroups per CU
__global__ __launch_bounds__(128) void gemm_bm32_kernel(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out, int R, int N) {
    extern __shared__ __attribute__((aligned(16))) float sA[];
    constexpr int BM2 = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r0 = (long)blockIdx.x * BM2;
    for (int i = tid; i < BM2 * (D / 4); i += 128) {
        const int row = i / (D / 4), c4 = i % (D / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + row < R) v = *reinterpret_cast<const f32x4*>(A + (r0 + row) * D + c4 * 4);
        *reinterpret_cast<f32x4*>(sA + row * LDA + c4 * 4) = v;
    }
    __syncthreads();
    const int l31 = lane & 31, half = lane >> 5;
    const float* aBase = sA + l31 * LDA + 4 * half;
    constexpr int NK = D / 8;
    for (int n0 = 0; n0 < N; n0 += D) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float* wBase = W + (long)(n0 + wave * 128 + l31) * D + 4 * half;
        f32x4 b[2][4], a[2];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) b[0][tn] = *reinterpret_cast<const f32x4*>(wBase + (long)tn * 32 * D);
        a[0] = *reinterpret_cast<const f32x4*>(aBase);
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < NK) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) b[nxt][tn] = *reinterpret_cast<const f32x4*>(wBase + (long)tn * 32 * D + (ks + 1) * 8);
                a[nxt] = *reinterpret_cast<const f32x4*>(aBase + (ks + 1) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][j], b[cur][tn][j], acc[tn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int tn = 0; tn < 4; ++tn) {
            const int col = n0 + wave * 128 + tn * 32 + l31;
            for (int r = 0; r < 16; ++r) {
                const long row = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < R) out[row * N + col] = acc[tn][r];
            }
        }
    }
}

float run_bm32(const float* A, const float* W, float* out, int R, int N, int iters) {
    size_t lds = (size_t)32 * LDA * 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(gemm_bm32_kernel, dim3((R + 31) / 32), dim3(128), lds, 0, A, W, out, R, N);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_bm32_kernel, dim3((R + 31) / 32), dim3(128), lds, 0, A, W, out, R, N);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
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
        for (int wgs : {512}) {
            const int iters = 16384;
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
    float *o2; CK(hipMalloc(&o2, (size_t)R * N * 4));
    auto cmp = [&](const char* name) {
        std::vector<float> h2(1 << 16), h0b(1 << 16);
        CK(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h0b.data(), o0, h0b.size() * 4, hipMemcpyDeviceToHost));
        double md2 = 0; for (size_t i = 0; i < h2.size(); ++i) md2 = fmax(md2, fabs(h2[i] - h0b[i]));
        printf("   max diff v0 vs %s: %g\n", name, md2);
    };
    t = run<13>(A, W, o2, R, N, 10); printf("variant 13 (v1 + quad-transposed dwordx4 stores): %.3f ms  %.1f TF\n", t, flops / t / 1e9); cmp("v13");
    t = run_bm32(A, W, o2, R, N, 10); printf("variant 20 (32-row panels, 2 waves/WG, 4 WG/CU): %.3f ms  %.1f TF\n", t, flops / t / 1e9); cmp("v20");
    t = run_cont<2>(A, W, o2, R, N, 10); printf("variant 10 (continuous prefetch PF=2): %.3f ms  %.1f TF\n", t, flops / t / 1e9); cmp("v10");
    t = run_cont<4>(A, W, o2, R, N, 10); printf("variant 11 (continuous prefetch PF=4): %.3f ms  %.1f TF\n", t, flops / t / 1e9); cmp("v11");
    t = run_cont<8>(A, W, o2, R, N, 10); printf("variant 12 (continuous prefetch PF=8): %.3f ms  %.1f TF\n", t, flops / t / 1e9); cmp("v12");
    t = run<8>(A, W, o2, R, N, 10); printf("variant 8 (v1 + per-CU parity setprio 1): %.3f ms  %.1f TF\n", t, flops / t / 1e9);
    t = run<9>(A, W, o2, R, N, 10); printf("variant 9 (v1 + per-CU parity setprio 3): %.3f ms  %.1f TF\n", t, flops / t / 1e9);
    t = run<7>(A, W, o2, R, N, 10); printf("variant 7 (v1, stores only in last pass): %.3f ms  %.1f TF\n", t, flops / t / 1e9);
    t = run<3>(A, W, o2, R, N, 10); printf("variant 3 (no global B loads): %.3f ms  %.1f TF\n", t, flops / t / 1e9);
    t = run<4>(A, W, o2, R, N, 10); printf("variant 4 (no LDS A reads): %.3f ms  %.1f TF\n", t, flops / t / 1e9);
    t = run<5>(A, W, o2, R, N, 10); printf("variant 5 (neither): %.3f ms  %.1f TF\n", t, flops / t / 1e9);
    {   // fragment-major repack of W on the host: Wp[n_tile][ks][lane][4] = W[n_tile*32 + l31][ks*8 + 4*half + j]
        std::vector<float> hP((size_t)N * D);
        for (int nt = 0; nt < N / 32; ++nt) for (int ks = 0; ks < D / 8; ++ks) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j)
            hP[((size_t)(nt * (D / 8) + ks) * 64 + l) * 4 + j] = hW[(size_t)(nt * 32 + (l & 31)) * D + ks * 8 + 4 * (l >> 5) + j];
        float* Wp; CK(hipMalloc(&Wp, hP.size() * 4)); CK(hipMemcpy(Wp, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
        t = run<6>(A, Wp, o2, R, N, 10); printf("variant 6 (fragment-major W): %.3f ms  %.1f TF\n", t, flops / t / 1e9);
        std::vector<float> h2(1 << 16); CK(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost));
        double md2 = 0; std::vector<float> h0b(1 << 16); CK(hipMemcpy(h0b.data(), o0, h0b.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h2.size(); ++i) md2 = fmax(md2, fabs(h2[i] - h0b[i]));
        printf("max diff v0 vs v6: %g\n", md2);
    }
    // check variant 2 == variant 0 bitwise (same k order)
    std::vector<float> h0(1 << 16), h1(1 << 16);
    CK(hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost));
    double md = 0; for (size_t i = 0; i < h0.size(); ++i) md = fmax(md, fabs(h0[i] - h1[i]));
    printf("max diff v0 vs v2: %g\n", md);
    return 0;
}
