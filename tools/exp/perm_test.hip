#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float rows4_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  v = a + b; a = v; b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(float* o) { o[threadIdx.x] = rows4_sum((float)(1 << (threadIdx.x >> 4)) + 100.f * (threadIdx.x & 15)); }
int main(){ float* d; hipMalloc(&d, 256); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
 int bad=0; for(int i=0;i<64;++i){ float want = 15.f + 400.f*(i&15); if (h[i]!=want) {bad++; printf("lane %d got %g want %g\n", i, h[i], want);} } printf(bad?"FAIL\n":"rows4_sum OK\n"); return bad; }
