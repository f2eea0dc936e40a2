// Do kernels whose waves need a SIMD's whole register file (512 VGPR + AGPR) start / end with a gap?  A X B sequences for
// rocprofv3 --kernel-trace (tools/exp_gaps-style post-processing): X = k_bigreg (touches v255 and a255) or k_plain.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k_small_a(float* p) { p[blockIdx.x * 256 + threadIdx.x] += 1.f; }
__global__ void __launch_bounds__(256) k_small_b(float* p) { p[blockIdx.x * 256 + threadIdx.x] += 2.f; }
__global__ void __launch_bounds__(256, 1) k_bigreg(float* p, int spin) {
  float x = p[blockIdx.x * 256 + threadIdx.x];
  asm volatile("v_mov_b32 v255, %0\n v_accvgpr_write_b32 a255, %0\n" : : "v"(x) : "v255", "a255");
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  p[blockIdx.x * 256 + threadIdx.x] = x;
}
__global__ void __launch_bounds__(256) k_plain(float* p, int spin) {
  float x = p[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  p[blockIdx.x * 256 + threadIdx.x] = x;
}
// ~190 KB of straight-line code (larger than the instruction cache): does the kernel boundary grow with the code size?
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
__global__ void __launch_bounds__(256) k_bigcode(float* p, int spin) {
  float x = p[blockIdx.x * 256 + threadIdx.x], y = x + 1.f, z = x + 2.f, w = x + 3.f;
  for (int i = 0; i < spin; ++i) {
    R256(R16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));))
  }
  p[blockIdx.x * 256 + threadIdx.x] = x + y + z + w;
}
int main() {
  float* p; hipMalloc(&p, 4096 * 256 * 4); hipMemset(p, 0, 4096 * 256 * 4);
  for (int rep = 0; rep < 60; ++rep) {
    hipLaunchKernelGGL(k_small_a, dim3(400), dim3(256), 0, 0, p);
    hipLaunchKernelGGL(k_bigreg, dim3(256), dim3(256), 0, 0, p, 20000);
    hipLaunchKernelGGL(k_small_b, dim3(400), dim3(256), 0, 0, p);
    hipLaunchKernelGGL(k_small_a, dim3(400), dim3(256), 0, 0, p);
    hipLaunchKernelGGL(k_plain, dim3(256), dim3(256), 0, 0, p, 20000);
    hipLaunchKernelGGL(k_small_b, dim3(400), dim3(256), 0, 0, p);
    hipLaunchKernelGGL(k_small_a, dim3(400), dim3(256), 0, 0, p);
    hipLaunchKernelGGL(k_bigcode, dim3(256), dim3(256), 0, 0, p, 4);
    hipLaunchKernelGGL(k_small_b, dim3(400), dim3(256), 0, 0, p);
  }
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
