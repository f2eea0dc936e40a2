// Micro-benchmark: LDS atomic throughput on gfx950 by data type - ds_add_u32 / ds_add_u64 / ds_add_f32 / ds_add_f64 - in the
// access pattern of a particle-to-grid scatter (lanes = particles, groups of PER_CELL consecutive lanes share a stencil
// origin, 27 nodes x 4 channels per particle into a workgroup tile).  Question it answers: the
// scatter kernels avoid ds_add_f32 (0.33 lanes/clk/CU) with a counting sort and ~45 barrier-separated phases; are the
// INTEGER LDS atomics fast enough to accumulate in fixed point instead?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class T> __device__ __forceinline__ void lds_add(T* p, T v);
template <> __device__ __forceinline__ void lds_add<unsigned>(unsigned* p, unsigned v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <> __device__ __forceinline__ void lds_add<unsigned long long>(unsigned long long* p, unsigned long long v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <> __device__ __forceinline__ void lds_add<float>(float* p, float v) { unsafeAtomicAdd(p, v); }
template <> __device__ __forceinline__ void lds_add<double>(double* p, double v) { unsafeAtomicAdd(p, v); }

// tile: NX x NY x NZ nodes x 4 channels of T; each thread = particle with cell (cx, cy, cz); 27 x 4 adds, REP times
template <class T, int PER_CELL, int LAYOUT, int PERM = 0>
__global__ void __launch_bounds__(256) k_scatter(T* out, int rep) {
  constexpr int NX = 8, NY = 12, NZ = 16;     // 1536 nodes
  __shared__ T tile[NX * NY * NZ * 4];
  for (int i = threadIdx.x; i < NX * NY * NZ * 4; i += 256) tile[i] = (T)0;
  __syncthreads();
  // PERM: thread t takes particle (t % 16) * 16 + t / 16 - the 16 lanes an LDS cycle serves then hold 16 different cells
  const int part = PERM ? (threadIdx.x & 15) * 16 + (threadIdx.x >> 4) : threadIdx.x;
  const int cell = part / PER_CELL;    // consecutive particles share a cell, cells walk along z then y then x
  const int cz = cell % (NZ - 2), cy = (cell / (NZ - 2)) % (NY - 2), cx = (cell / ((NZ - 2) * (NY - 2))) % (NX - 2);
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int node = ((cx + i) * NY + cy + j) * NZ + cz + k;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            // LAYOUT 0: AoS (node, channel); 1: SoA (channel planes)
            const int a = LAYOUT == 0 ? node * 4 + c : c * (NX * NY * NZ) + node;
            lds_add<T>(&tile[a], (T)(threadIdx.x + c + r + 1));
          }
        }
  }
  __syncthreads();
  T acc = (T)0;
  for (int i = threadIdx.x; i < NX * NY * NZ * 4; i += 256) acc += tile[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// plain pattern: each lane its own address (conflict-free), to read the raw instruction rate
template <class T>
__global__ void __launch_bounds__(256) k_plain(T* out, int rep) {
  __shared__ T s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (T)0;
  __syncthreads();
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int u = 0; u < 16; ++u) lds_add<T>(&s[(threadIdx.x + 256 * u) & 4095], (T)(r + 1));
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x];
}

int main() {
  void* out; CK(hipMalloc(&out, 64 << 20));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 512, rep = 64;
  auto run = [&](const char* name, auto launch, double ops) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-52s %9.3f ms  %9.2f Gop/s  (%6.2f lane-ops/clk/CU @2.4GHz)\n", name, ms, ops / ms / 1e6, ops / ms / 1e6 / 256 / 2.4);
    return 0;
  };
  const double pops = (double)blocks * 256 * rep * 16, sops = (double)blocks * 256 * rep * 108;
#define PLAIN(T, nm) run("plain distinct " nm, [&] { hipLaunchKernelGGL(k_plain<T>, dim3(blocks), dim3(256), 0, 0, (T*)out, rep); }, pops)
  PLAIN(unsigned, "ds_add_u32");
  PLAIN(unsigned long long, "ds_add_u64");
  PLAIN(float, "ds_add_f32");
  PLAIN(double, "ds_add_f64");
#define SCAT(T, PC, LAY, nm) run("scatter " nm, [&] { hipLaunchKernelGGL((k_scatter<T, PC, LAY>), dim3(blocks), dim3(256), 0, 0, (T*)out, rep); }, sops)
  SCAT(unsigned, 1, 0, "u32 1/cell AoS");
  SCAT(unsigned, 4, 0, "u32 4/cell AoS");
  SCAT(unsigned, 8, 0, "u32 8/cell AoS");
  SCAT(unsigned, 4, 1, "u32 4/cell SoA");
  SCAT(unsigned, 8, 1, "u32 8/cell SoA");
  SCAT(unsigned long long, 1, 0, "u64 1/cell AoS");
  SCAT(unsigned long long, 4, 0, "u64 4/cell AoS");
  SCAT(unsigned long long, 8, 0, "u64 8/cell AoS");
  SCAT(unsigned long long, 4, 1, "u64 4/cell SoA");
#define SCATP(T, PC, LAY, nm) run("scatter PERM " nm, [&] { hipLaunchKernelGGL((k_scatter<T, PC, LAY, 1>), dim3(blocks), dim3(256), 0, 0, (T*)out, rep); }, sops)
  SCATP(unsigned long long, 4, 0, "u64 4/cell AoS");
  SCATP(unsigned long long, 4, 1, "u64 4/cell SoA");
  SCATP(unsigned long long, 6, 1, "u64 6/cell SoA");
  SCATP(double, 4, 0, "f64 4/cell AoS");
  SCATP(double, 4, 1, "f64 4/cell SoA");
  SCATP(double, 6, 1, "f64 6/cell SoA");
  SCATP(double, 8, 1, "f64 8/cell SoA");
  SCAT(double, 1, 1, "f64 1/cell SoA");
  SCAT(double, 1, 0, "f64 1/cell AoS");
  SCAT(float, 4, 0, "f32 4/cell AoS");
  SCAT(float, 4, 1, "f32 4/cell SoA");
  SCAT(double, 4, 0, "f64 4/cell AoS");
  SCAT(double, 4, 1, "f64 4/cell SoA");
  return 0;
}
