// Issue cost of the vector instructions the trajectory kernel's VALU phases are made of (gfx950), in shader cycles per
// wave-instruction, for ONE and TWO waves per SIMD (one workgroup of 256 / 512 threads on one CU), independent and dependent
// chains.  Build: hipcc -O3 --offload-arch=gfx950 tools/exp/valu_rates.hip -o tools/exp/valu_rates ; run once on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 256, UNROLL = 8;

// BODY: one instruction on registers r[i] (independent over i) ; the chain variant feeds every instruction from the previous one
#define KERNEL(NAME, DECL, INDEP, DEP)                                                                        \
    __global__ void NAME##_indep(unsigned long long *out, float seed) {                                      \
        DECL;                                                                                                \
        __syncthreads();                                                                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int it = 0; it < ITERS; ++it) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) { INDEP; }                                    \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
        float acc = 0;                                                                                       \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) acc += sink(r[i]);                                \
        if (acc == 12345.678f) out[1000] = 1;                                                                \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                        \
    }                                                                                                        \
    __global__ void NAME##_dep(unsigned long long *out, float seed) {                                        \
        DECL;                                                                                                \
        __syncthreads();                                                                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
        for (int it = 0; it < ITERS; ++it) {                                                                 \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) { DEP; }                                      \
        }                                                                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
        float acc = 0;                                                                                       \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) acc += sink(r[i]);                                \
        if (acc == 12345.678f) out[1000] = 1;                                                                \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                        \
    }

__device__ __forceinline__ float sink(float v) { return v; }
__device__ __forceinline__ float sink(f32x2 v) { return v[0] + v[1]; }
__device__ __forceinline__ float sink(unsigned v) { return (float)v; }

#define DECL_F float r[UNROLL]; _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) r[i] = seed + i + threadIdx.x
#define DECL_F2 f32x2 r[UNROLL]; _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) r[i] = f32x2{seed + i, seed + threadIdx.x}
#define DECL_U unsigned r[UNROLL]; float x = seed + threadIdx.x; _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) r[i] = i + threadIdx.x

KERNEL(fma, DECL_F, asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i])), asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(pk_fma, DECL_F2, asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(r[i])), asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(pk_mul, DECL_F2, asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(r[i])), asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(pk_add, DECL_F2, asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(r[i])), asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(add, DECL_F, asm volatile("v_add_f32 %0, %0, %0" : "+v"(r[i])), asm volatile("v_add_f32 %0, %1, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(exp, DECL_F, asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])), asm volatile("v_exp_f32 %0, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(rcp, DECL_F, asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i])), asm volatile("v_rcp_f32 %0, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(cvt_pk, DECL_F, asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(r[i])), asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(mixlo, DECL_F, asm volatile("v_fma_mixlo_f16 %0, %0, 1.0, -%0 op_sel_hi:[0,0,1]" : "+v"(r[i])), asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%1 op_sel_hi:[0,0,1]" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(cvt_f32_f16, DECL_F, asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(r[i])), asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(max, DECL_F, asm volatile("v_max_f32 %0, %0, %0" : "+v"(r[i])), asm volatile("v_max_f32 %0, %1, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(mov, DECL_F, asm volatile("v_mov_b32 %0, %0" : "+v"(r[i])), asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(addu, DECL_U, asm volatile("v_add_u32 %0, %0, %0" : "+v"(r[i])), asm volatile("v_add_u32 %0, %1, %1" : "=v"(r[i]) : "v"(r[(i + UNROLL - 1) % UNROLL])))
KERNEL(perm32, DECL_F, asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 1) % UNROLL])), asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + UNROLL - 1) % UNROLL])))

// LDS stores / loads of the kernel's shapes (addresses conflict-free: consecutive lanes, consecutive 8 / 16 bytes)
__global__ void ds_write_b64_k(unsigned long long *out, float seed) {
    extern __shared__ char smem[];
    f32x2 v = {seed, seed + threadIdx.x};
    const unsigned at = threadIdx.x * 8;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(at), "v"(v), "n"(4096 * (i % 8)));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
__global__ void ds_read_b128_k(unsigned long long *out, float seed) {
    extern __shared__ char smem[];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v[UNROLL];
    const unsigned at = threadIdx.x * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"(at), "n"(8192 * (i % 4)));
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc += v[i][0];
    if (acc == 12345.678f) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <class K>
static int run(const char *name, K k, unsigned long long *d_out, size_t lds = 0) {
    printf("%-22s", name);
    for (int threads : {64, 256, 512}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), lds, 0, d_out, 1.0f);
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), lds, 0, d_out, 1.0f);
        CK(hipDeviceSynchronize());
        unsigned long long h[8];
        CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
        unsigned long long mx = 0;
        for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
        printf("  %2d waves: %6.2f", threads / 64, (double)mx / (ITERS * UNROLL));
    }
    printf("   (s_memtime ticks per wave-instruction; 1 / 4 / 8 waves = 1 wave on 1 SIMD / 1 per SIMD / 2 per SIMD)\n");
    return 0;
}

int main() {
    unsigned long long *d_out;
    CK(hipMalloc(&d_out, 2048 * 8));
    CK(hipFuncSetAttribute((const void *)ds_write_b64_k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void *)ds_read_b128_k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
#define RUN(N) run(#N " independent", N##_indep, d_out); run(#N " dependent", N##_dep, d_out)
    RUN(fma); RUN(pk_fma); RUN(pk_mul); RUN(pk_add); RUN(add); RUN(max); RUN(mov); RUN(addu); RUN(exp); RUN(rcp); RUN(cvt_pk); RUN(mixlo); RUN(cvt_f32_f16); RUN(perm32);
    run("ds_write_b64", ds_write_b64_k, d_out, 65536);
    run("ds_read_b128", ds_read_b128_k, d_out, 65536);
    return 0;
}
