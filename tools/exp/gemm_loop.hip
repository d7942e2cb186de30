// What limits the K = 256 GEMM loop of the trajectory kernel (gemm_pipe: 83 % of the MFMA rate in the step kernel)?  One workgroup
// per CU repeats the W1-shaped GEMM (8 waves x [2 n-tiles x 7 token tiles x 8 k-steps x 3 MFMAs]) REP times with
//   MODE 0: as in the kernel (A fragments from global / L2 per k-step, B fragments from the LDS panel per tile)
//   MODE 1: A fragments loaded once (resident), B from LDS        MODE 2: A from global, B loaded once (resident)
//   MODE 3: both resident (pure MFMA issue)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -I include -I soccerdiffusion_amd/csrc
//        tools/exp/gemm_loop.hip -o tools/exp/gemm_loop ; run once on the GPU box.
#include "sd_traj.h"
#include <cstdio>
#include <vector>

using namespace tj;
constexpr int REP = 64;

template <int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void k(const f16 *__restrict__ w, float *__restrict__ out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx c;
    ctx_init(c, smem, 100);
    for (int i = threadIdx.x; i < TMAX * XROW / 4; i += NTHREADS) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u + (i & 7);   // finite fp16 pairs
    __syncthreads();
    f32x4 U[2][NTT];
    for (int a = 0; a < 2; ++a)
        for (int tt = 0; tt < NTT; ++tt) U[a][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f16 *pa0 = w + (long)(2 * c.w) * (8 * 2 * 512), *pa1 = pa0 + 8 * 2 * 512;
    const char *X = c.smem + LDS_X;
    const unsigned lo = (unsigned)c.lane * 8;
    f16x8 ra0[2], ra1[2], rb[2];
    for (int pl = 0; pl < 2; ++pl) {
        ra0[pl] = *reinterpret_cast<const f16x8 *>(pa0 + lo + pl * 512);
        ra1[pl] = *reinterpret_cast<const f16x8 *>(pa1 + lo + pl * 512);
        rb[pl] = lds16(X + x_at(c, 0, pl, 0));
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int rep = 0; rep < REP; ++rep) {
        if (MODE == 0) {
            gemm_x2<0>(c, U, w);
        } else {
            f16x8 a0[2][2], a1[2][2], b[2][2];
            for (int pl = 0; pl < 2; ++pl) { a0[0][pl] = ra0[pl]; a1[0][pl] = ra1[pl]; a0[1][pl] = ra0[pl]; a1[1][pl] = ra1[pl]; b[0][pl] = rb[pl]; b[1][pl] = rb[pl]; }
            if (MODE == 2) for (int pl = 0; pl < 2; ++pl) { a0[0][pl] = *reinterpret_cast<const f16x8 *>(pa0 + lo + pl * 512); a1[0][pl] = *reinterpret_cast<const f16x8 *>(pa1 + lo + pl * 512); }
            if (MODE == 1) for (int pl = 0; pl < 2; ++pl) b[0][pl] = lds16(X + x_at(c, 0, pl, 0));
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    const int cur = (ks * NTT + tt) & 1, nxt = cur ^ 1;
                    if (MODE == 2 && tt == 0 && ks + 1 < 8) {
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) {
                            a0[(ks + 1) & 1][pl] = *reinterpret_cast<const f16x8 *>(pa0 + lo + ((ks + 1) * 2 + pl) * 512);
                            a1[(ks + 1) & 1][pl] = *reinterpret_cast<const f16x8 *>(pa1 + lo + ((ks + 1) * 2 + pl) * 512);
                        }
                    }
                    const int nks = tt + 1 < NTT ? ks : ks + 1, ntt = tt + 1 < NTT ? tt + 1 : 0;
                    if (MODE == 1 && nks < 8) {
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) b[nxt][pl] = lds16(X + x_at(c, ntt, pl, nks));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int ak = MODE == 2 ? (ks & 1) : 0, bk = MODE == 1 ? cur : 0;
                    mma3<0>(U[0][tt], a0[ak][0], a0[ak][1], b[bk][0], b[bk][1]);
                    mma3<0>(U[1][tt], a1[ak][0], a1[ak][1], b[bk][0], b[bk][1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int a = 0; a < 2; ++a)
        for (int tt = 0; tt < NTT; ++tt) acc += U[a][tt][0] + U[a][tt][3];
    out[(long)blockIdx.x * NTHREADS + threadIdx.x] = acc;
    if (blockIdx.x == 0 && c.lane == 0) cyc[c.w] = t1 - t0;
}

template <int MODE>
static void run(const char *name, const f16 *w, float *out, unsigned long long *cyc, int grid) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(NTHREADS), LDS_BYTES, 0, w, out, cyc);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s grid %4d:", name, grid);
    for (int wv : {0, 3, 4, 7}) printf("  wave %d %7.0f", wv, (double)h[wv] / REP);
    printf("   cycles per GEMM (MFMA issue of a SIMD's two waves: 10752)\n");
}

int main() {
    f16 *w;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&w, 256 * 256 * 2 * 2);
    std::vector<unsigned short> hw(256 * 256 * 2, 0x3400);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, 4096L * NTHREADS * 4);
    hipMalloc(&cyc, 64);
    for (int grid : {1, 256, 4096}) {
        run<0>("as in the kernel (A from L2, B from LDS)", w, out, cyc, grid);
        run<1>("A resident, B from LDS", w, out, cyc, grid);
        run<2>("A from L2, B resident", w, out, cyc, grid);
        run<3>("A and B resident (MFMA issue only)", w, out, cyc, grid);
    }
    return 0;
}
