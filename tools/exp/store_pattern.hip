// How fast does the chip take the row tensors a panel workgroup stores, by store pattern?  6 400 workgroups (the B = 4096
// sampler launch) x 256 threads, each workgroup writes `tensors` x 64 KB (64 rows x 256 floats, rows 1 KB apart) and,
// optionally, reads as much.  Patterns:
//   0  wave-contiguous: an instruction writes 1 KB = one row (lane x 16 bytes)              [accumulator-order h]
//   1  quad epilogue:   an instruction writes 8 rows x 128 bytes (the quad-transposed tile)  [chain epilogues]
//   2  row pass:        an instruction writes 4 rows x 256 bytes (16 lanes per row)          [LayerNorm row pass]
// Build: hipcc -O3 --offload-arch=gfx950 tools/exp/store_pattern.hip -o tools/exp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", e, __FILE__, __LINE__); exit(1);} } while (0)

template <int PAT, bool NT, bool READ>
__global__ __launch_bounds__(256, 2) void store_kernel(float *dst, const float *src, int tensors, long tensor_stride) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long base = (long)blockIdx.x * 64 * 256;
    f32x4 acc = {1.f, 2.f, 3.f, (float)tid};
    for (int t = 0; t < tensors; ++t) {
        float *d = dst + t * tensor_stride + base;
        const float *s = src + t * tensor_stride + base;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            long off;
            if (PAT == 0) off = (long)(wave * 16 + i) * 256 + lane * 4;                                  // row = wave*16+i
            else if (PAT == 1) {                                                                          // 8 rows x 128 B
                const int l31 = lane & 31, half = lane >> 5, i4 = lane & 3;
                const int row = (i & 3) * 8 + 4 * half + i4 + (i >> 2 & 1) * 32, c0 = wave * 64 + (i >> 3) * 32 + (l31 & ~3);
                off = (long)row * 256 + c0;
            } else {                                                                                      // 4 rows x 256 B
                const int sub = lane & 15, grp = lane >> 4;
                const int row = wave * 4 + grp + 16 * (i >> 2), c = 4 * (sub + 16 * (i & 3));
                off = (long)row * 256 + c;
            }
            if (READ) acc = acc + *reinterpret_cast<const f32x4 *>(s + off);
            if (NT) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4 *>(d + off));
            else *reinterpret_cast<f32x4 *>(d + off) = acc;
        }
    }
}

template <int PAT, bool NT, bool READ>
static void run(const char *name, float *dst, float *src, int wgs, int tensors) {
    const long stride = (long)wgs * 64 * 256;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((store_kernel<PAT, NT, READ>), dim3(wgs), dim3(256), 0, 0, dst, src, tensors, stride);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((store_kernel<PAT, NT, READ>), dim3(wgs), dim3(256), 0, 0, dst, src, tensors, stride);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)wgs * tensors * 65536.0;
    printf("%-44s wgs %5d tensors %d: %7.1f us  write %5.2f TB/s%s\n", name, wgs, tensors, ms / 5 * 1e3, bytes / (ms / 5 * 1e-3) / 1e12,
           READ ? " (+ the same read)" : "");
}

int main() {
    const int tensors = 6;
    for (int wgs : {400, 6400}) {
        float *dst, *src;
        const size_t n = (size_t)wgs * 64 * 256 * tensors;
        CK(hipMalloc(&dst, n * 4)); CK(hipMalloc(&src, n * 4));
        CK(hipMemset(src, 0, n * 4));
        run<0, false, false>("wave-contiguous 1 KB rows", dst, src, wgs, tensors);
        run<0, true, false>("wave-contiguous 1 KB rows, nontemporal", dst, src, wgs, tensors);
        run<1, false, false>("quad epilogue 8 x 128 B", dst, src, wgs, tensors);
        run<1, true, false>("quad epilogue 8 x 128 B, nontemporal", dst, src, wgs, tensors);
        run<2, false, false>("row pass 4 x 256 B", dst, src, wgs, tensors);
        run<2, true, false>("row pass 4 x 256 B, nontemporal", dst, src, wgs, tensors);
        run<0, true, true>("wave-contiguous, nontemporal, read + write", dst, src, wgs, tensors);
        run<1, true, true>("quad epilogue, nontemporal, read + write", dst, src, wgs, tensors);
        CK(hipFree(dst)); CK(hipFree(src));
    }
    return 0;
}
