// Flush patterns of float atomics on gfx950 (which lane adds which float): a wave flushes 64 float4 nodes (1 KB contiguous per
// wave and step, the blocks scattered over `range`) either node-per-lane (4 instructions, lanes 16 B apart) or
// float-per-lane (4 instructions, 64 consecutive floats each); and 8-float rows (a (tile, Gaussian) gradient row) either
// row-per-lane (8 instructions, lanes one row apart) or 8 rows x 8 floats per instruction, rows 9 or 16 floats apart.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_flush.hip -o tools/ubench_flush && tools/ubench_flush
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned hash(unsigned x) { x *= 2654435761u; x ^= x >> 13; x *= 2246822519u; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(256) k(float* buf, int steps, unsigned range /* floats */) {
  const unsigned lane = threadIdx.x & 63, wv = (blockIdx.x * 256 + threadIdx.x) >> 6;
  for (int s = 0; s < steps; ++s) {
    if (MODE == 0 || MODE == 1) {
      const unsigned blk = hash(wv * 977u + s) % (range / 256u);       // a 64-node block: 256 consecutive floats
      float* b = buf + (size_t)blk * 256u;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const unsigned a = MODE == 0 ? lane * 4 + c : 64 * c + lane;
        unsafeAtomicAdd(b + a, 1.0f);
      }
    } else {
      // 64 rows of 8 floats, each row somewhere else (stride RS floats)
      const int RS = (MODE == 2 || MODE == 3) ? 9 : 16;
      const unsigned nrow = range / RS;
      if (MODE == 2 || MODE == 4) {        // row per lane, 8 instructions
        const unsigned row = hash((wv * 64 + lane) * 31u + s) % nrow;
#pragma unroll
        for (int c = 0; c < 8; ++c) unsafeAtomicAdd(buf + (size_t)row * RS + c, 1.0f);
      } else {                             // 8 rows x 8 floats per instruction, 8 instructions
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const unsigned row = hash((wv * 64 + c * 8 + (lane >> 3)) * 31u + s) % nrow;
          unsafeAtomicAdd(buf + (size_t)row * RS + (lane & 7), 1.0f);
        }
      }
    }
  }
}
int main() {
  float* buf; size_t nbuf = (size_t)1 << 24; CK(hipMalloc(&buf, nbuf * 4)); CK(hipMemset(buf, 0, nbuf * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* nm[6] = {"node float4: lane = node (16 B apart)", "node float4: lane = float (64 consecutive)", "row of 8, stride 9: lane = row",
                       "row of 8, stride 9: 8 rows x 8 floats", "row of 8, stride 16: lane = row", "row of 8, stride 16: 8 rows x 8 floats"};
  for (unsigned range : {1u << 19, 1u << 21}) {
    for (int m = 0; m < 6; ++m) {
      const int gb = 2048, steps = 16;
      auto launch = [&] {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(gb), dim3(256), 0, 0, buf, steps, range);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(gb), dim3(256), 0, 0, buf, steps, range);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(gb), dim3(256), 0, 0, buf, steps, range);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(gb), dim3(256), 0, 0, buf, steps, range);
        if (m == 4) hipLaunchKernelGGL(k<4>, dim3(gb), dim3(256), 0, 0, buf, steps, range);
        if (m == 5) hipLaunchKernelGGL(k<5>, dim3(gb), dim3(256), 0, 0, buf, steps, range);
      };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double ops = (double)gb * 256 * steps * (m < 2 ? 4 : 8);
      printf("range %8u floats  %-46s %8.3f ms %8.1f G float atomics/s\n", range, nm[m], ms, ops / ms / 1e6);
    }
  }
  return 0;
}
