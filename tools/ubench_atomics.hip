// Micro-benchmarks: LDS float atomics and global float atomics on gfx950 (numbers feed DESIGN.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each lane does ITER ds_add_f32; address pattern: mode 0 distinct consecutive (stride 1), 1 stride 4 (AoS float4),
// 2 groups of 8 lanes share an address, 3 all lanes same address, 4 random distinct
template <int MODE>
__global__ void __launch_bounds__(256) k_lds(float* out, int iters) {
  __shared__ float s[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) s[i] = 0.f;
  __syncthreads();
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int a;
  if (MODE == 0) a = w * 1024 + lane;
  else if (MODE == 1) a = w * 1024 + lane * 4;
  else if (MODE == 2) a = w * 1024 + (lane >> 3);
  else if (MODE == 3) a = w * 1024;
  else a = w * 1024 + ((lane * 37) & 1023);
  for (int i = 0; i < iters; ++i) {
    unsafeAtomicAdd(&s[(a + i * 64) & 8191], 1.0f);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[0];
}

// plain LDS store+load equivalent for reference
__global__ void __launch_bounds__(256) k_lds_rw(float* out, int iters) {
  __shared__ float s[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) s[i] = 0.f;
  __syncthreads();
  int a = threadIdx.x;
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) { int q = (a + i * 64) & 8191; float v = s[q]; s[q] = v + 1.0f; acc += v; }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[0] + acc;
}

// global atomics: each thread does `per` atomics; mode 0: addresses unique per thread (stride 1), 1: 4 consecutive floats
// per thread (float4 node), 2: all threads of a block hit 256 shared addresses that 8 blocks also share, 3: random over `range`
__global__ void __launch_bounds__(256) k_gatom(float* buf, int per, int mode, unsigned range) {
  unsigned t = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < per; ++i) {
    unsigned a;
    if (mode == 0) a = (t + i * gridDim.x * 256u) % range;
    else if (mode == 1) a = ((t * 4 + (i & 3)) + (i >> 2) * gridDim.x * 1024u) % range;
    else if (mode == 2) a = ((blockIdx.x >> 3) * 1024u + threadIdx.x * 4 + (i & 3)) % range;
    else { unsigned h = t * 2654435761u + i * 40503u; h ^= h >> 13; a = (h * 2246822519u) % range; }
    unsafeAtomicAdd(&buf[a], 1.0f);
  }
}

int main() {
  float* out; CK(hipMalloc(&out, 1 << 20));
  float* buf; size_t nbuf = 64u << 20; CK(hipMalloc(&buf, nbuf * 4)); CK(hipMemset(buf, 0, nbuf * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * 4, iters = 2048;
  auto run = [&](const char* name, auto launch, double ops) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %9.3f ms  %8.2f Gop/s  (%.2f ops/clk/CU @2.4GHz)\n", name, ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    return 0;
  };
  double lops = (double)blocks * 256 * iters;
  run("lds ds_add_f32 distinct stride1", [&] { hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), 0, 0, out, iters); }, lops);
  run("lds ds_add_f32 stride4 (AoS float4)", [&] { hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 0, 0, out, iters); }, lops);
  run("lds ds_add_f32 8 lanes / address", [&] { hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(256), 0, 0, out, iters); }, lops);
  run("lds ds_add_f32 64 lanes / address", [&] { hipLaunchKernelGGL(k_lds<3>, dim3(blocks), dim3(256), 0, 0, out, iters); }, lops);
  run("lds ds_add_f32 scattered distinct", [&] { hipLaunchKernelGGL(k_lds<4>, dim3(blocks), dim3(256), 0, 0, out, iters); }, lops);
  run("lds read+write (non atomic)", [&] { hipLaunchKernelGGL(k_lds_rw, dim3(blocks), dim3(256), 0, 0, out, iters); }, lops);
  for (unsigned range : {1u << 16, 1u << 20, 1u << 24}) {
    for (int mode = 0; mode < 4; ++mode) {
      int gb = 2048, per = 64;
      char nm[128]; snprintf(nm, sizeof(nm), "global atomic_add_f32 mode %d range %u floats", mode, range);
      run(nm, [&] { hipLaunchKernelGGL(k_gatom, dim3(gb), dim3(256), 0, 0, buf, per, mode, range); }, (double)gb * 256 * per);
    }
  }
  return 0;
}
