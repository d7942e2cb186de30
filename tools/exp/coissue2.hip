// Do an MFMA-bound wave and a VALU-bound wave that share a SIMD overlap?  (decides whether sd_traj.h should specialise its
// two wave quartets: one running row GEMMs, one running softmax / splits.)  512-thread workgroups, one per CU: waves 0..3 run
// `nm` v_mfma_f32_16x16x32_f16 (three per accumulator in a row, as the split-fp16 GEMM issues them, 7 accumulators), waves 4..7
// run `nv` VALU instructions (fma / cvt mix, 8 independent chains).  mode: 1 = MFMA quartet only, 2 = VALU quartet only, 3 = both;
// prio: s_setprio of the VALU quartet.
//   hipcc -O3 --offload-arch=gfx950 tools/exp/coissue2.hip -o tools/exp/coissue2 && tools/exp/coissue2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PRIO>
__global__ __launch_bounds__(512, 2) void k(int mode, int iters, float *out, unsigned long long *cyc) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (w < 4) {
        if (mode & 1) {
            f16x8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.01f + e); b[e] = (_Float16)(e * 0.5f - lane * 0.02f); }
            f32x4 acc[7];
            for (int i = 0; i < 7; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc[i], 0, 0, 0);
                }
            }
            float s = 0.f;
            for (int i = 0; i < 7; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    } else {
        if (mode & 2) {
            if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
            float x[8];
            for (int e = 0; e < 8; ++e) x[e] = lane * 0.001f + e;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const _Float16 h = (_Float16)x[e];               // cvt
                        x[e] = fmaf(x[e] - (float)h, 1.0009765625f, 0.25f);   // cvt, sub, fma: the split's instruction mix
                    }
            }
            float s = 0.f;
            for (int e = 0; e < 8; ++e) s += x[e];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <int PRIO>
static void run(int mode, int iters, float *out, unsigned long long *cyc, const char *what) {
    hipLaunchKernelGGL(k<PRIO>, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PRIO>, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %8.1f us   cycles: MFMA wave %8llu  VALU wave %8llu\n", what, ms * 1000.f, h[0], h[4]);
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;   // 42 000 MFMAs (672 k cycles at 16 per MFMA) / 48 000 x 4 VALU instructions per wave
    run<0>(1, iters, out, cyc, "MFMA quartet alone");
    run<0>(2, iters, out, cyc, "VALU quartet alone");
    run<0>(3, iters, out, cyc, "both, equal priority");
    run<1>(3, iters, out, cyc, "both, VALU quartet at s_setprio 1");
    run<3>(3, iters, out, cyc, "both, VALU quartet at s_setprio 3");
    return 0;
}
