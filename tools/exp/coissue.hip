// Do VALU instructions of one wave issue beside the fp16 / fp32 MFMAs of another wave on the same SIMD?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 role B.  Each wave times its own loop
// with s_memtime; the roles are run alone and together.
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/coissue.hip -o tools/exp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", e, __FILE__, __LINE__); exit(1);} } while (0)

// mode bits: 1 = role A active (MFMA), 2 = role B active (VALU); kind: 0 fp16 MFMA, 1 fp32 MFMA; chain: B's VALU ops dependent (1) or 8 independent chains (0)
template <int KIND, int CHAIN>
__global__ __launch_bounds__(512) void coissue_kernel(unsigned long long *t, int mode, int iters, float seed, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool roleA = wave < 4;
    unsigned long long t0 = 0, t1 = 0;
    float out = 0.f;
    __syncthreads();
    if (roleA && (mode & 1)) {
        if (mode & 8) __builtin_amdgcn_s_setprio(0);
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = seed;
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed + e); b[e] = (_Float16)(seed - e); }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
                else acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed + 1.f, acc[u], 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 4; ++i) out += acc[i][lane & 15];
    } else if (!roleA && (mode & 2)) {
        if (mode & 4) __builtin_amdgcn_s_setprio(3);   // role B at high issue priority
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed + i + lane;
        t0 = __builtin_amdgcn_s_memtime();
        if constexpr (CHAIN) {
            for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[0] = __builtin_fmaf(v[0], 1.0001f, 0.5f);   // one dependent chain
            }
        } else {
            for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = __builtin_fmaf(v[u], 1.0001f, 0.5f);   // 8 independent chains
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) out += v[i];
    }
    if (lane == 0) t[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
    if (out == 12345.678f) sink[0] = out;
}

int main() {
    const int CUS = 256, iters = 4096;
    unsigned long long *t; float *sink;
    CK(hipMalloc(&t, CUS * 8 * sizeof(unsigned long long))); CK(hipMalloc(&sink, 4));
    std::vector<unsigned long long> h(CUS * 8);
    auto run = [&](int mode, int kind, int chain, double &a, double &b) {
        if (kind == 0 && chain == 0) hipLaunchKernelGGL((coissue_kernel<0, 0>), dim3(CUS), dim3(512), 0, 0, t, mode, iters, 1.0f, sink);
        else if (kind == 0) hipLaunchKernelGGL((coissue_kernel<0, 1>), dim3(CUS), dim3(512), 0, 0, t, mode, iters, 1.0f, sink);
        else if (chain == 0) hipLaunchKernelGGL((coissue_kernel<1, 0>), dim3(CUS), dim3(512), 0, 0, t, mode, iters, 1.0f, sink);
        else hipLaunchKernelGGL((coissue_kernel<1, 1>), dim3(CUS), dim3(512), 0, 0, t, mode, iters, 1.0f, sink);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), t, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> va, vb;
        for (int c = 0; c < CUS; ++c) for (int w = 0; w < 8; ++w) (w < 4 ? va : vb).push_back((double)h[c * 8 + w]);
        std::sort(va.begin(), va.end()); std::sort(vb.begin(), vb.end());
        a = va[va.size() / 2]; b = vb[vb.size() / 2];
    };
    for (int kind = 0; kind < 2; ++kind)
        for (int chain = 0; chain < 2; ++chain) {
            double a1, b1, a2, b2, a3, b3;
            run(1, kind, chain, a1, b1); run(2, kind, chain, a2, b2); run(3, kind, chain, a3, b3);
            double a4, b4; run(3 | 4, kind, chain, a4, b4);
            printf("   with s_setprio 3 on the VALU wave: together MFMA %.0f, VALU %.0f\n", a4, b4);
            printf("%s MFMA (%d per wave), VALU %s (%d fma per wave): MFMA alone %.0f cyc, VALU alone %.0f cyc; together MFMA %.0f, VALU %.0f\n",
                   kind ? "fp32 32x32x2" : "fp16 32x32x16", iters * 4, chain ? "one dependent chain" : "8 independent chains", iters * 32, a1, b2, a3, b3);
        }
    return 0;
}
