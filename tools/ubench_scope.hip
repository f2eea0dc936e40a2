// Global float atomics by scope on gfx950: agent (what atomicAdd / unsafeAtomicAdd emit) against workgroup scope on a
// per-XCD private copy of the target (HW_REG_XCC_ID picks the copy: the atomic can then execute in that XCD's L2).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_scope.hip -o tools/ubench_scope && tools/ubench_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
// mode 1: four consecutive floats per thread and step (a grid node {mv, m}); mode 3: random over `range`
template <int SCOPE /* 0 agent, 1 workgroup on the XCD's private copy, 2 integer returning agent, 3 integer returning workgroup/XCD */>
__global__ void __launch_bounds__(256) k(float* buf, int per, int mode, unsigned range) {
  const unsigned t = blockIdx.x * 256 + threadIdx.x;
  float* base = buf;
  if (SCOPE == 1 || SCOPE == 3) base = buf + (size_t)xcc_id() * range;
  unsigned sink = 0;
  for (int i = 0; i < per; ++i) {
    unsigned a;
    if (mode == 1) a = ((t * 4 + (i & 3)) + (i >> 2) * gridDim.x * 1024u) % range;
    else { unsigned h = t * 2654435761u + i * 40503u; h ^= h >> 13; a = (h * 2246822519u) % range; }
    if (SCOPE == 0) __hip_atomic_fetch_add(&base[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (SCOPE == 1) __hip_atomic_fetch_add(&base[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (SCOPE == 2) sink += __hip_atomic_fetch_add((unsigned*)&base[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else sink += __hip_atomic_fetch_add((unsigned*)&base[a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (sink == 0xdeadbeefu) buf[0] = 1.f;
}
__global__ void k_sum(const float* buf, unsigned range, int copies, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)range * copies; i += (size_t)gridDim.x * blockDim.x) s += buf[i];
  atomicAdd(out, s);
}
int main() {
  float* buf; size_t nbuf = (size_t)8 << 24; CK(hipMalloc(&buf, nbuf * 4));
  double* tot; CK(hipMalloc(&tot, 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const char* names[4] = {"f32 add, agent scope", "f32 add, workgroup scope, per-XCD copy", "u32 add returning, agent scope",
                          "u32 add returning, workgroup scope, per-XCD copy"};
  for (unsigned range : {1u << 18, 1u << 22}) {
    for (int mode : {1, 3}) {
      for (int sc = 0; sc < 4; ++sc) {
        const int gb = 2048, per = 64;
        auto launch = [&] {
          if (sc == 0) hipLaunchKernelGGL(k<0>, dim3(gb), dim3(256), 0, 0, buf, per, mode, range);
          if (sc == 1) hipLaunchKernelGGL(k<1>, dim3(gb), dim3(256), 0, 0, buf, per, mode, range);
          if (sc == 2) hipLaunchKernelGGL(k<2>, dim3(gb), dim3(256), 0, 0, buf, per, mode, range);
          if (sc == 3) hipLaunchKernelGGL(k<3>, dim3(gb), dim3(256), 0, 0, buf, per, mode, range);
        };
        CK(hipMemset(buf, 0, nbuf * 4));
        launch(); CK(hipDeviceSynchronize());
        CK(hipMemset(buf, 0, nbuf * 4)); CK(hipMemset(tot, 0, 8)); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        double h = 0;
        if (sc < 2) { hipLaunchKernelGGL(k_sum, dim3(512), dim3(256), 0, 0, buf, range, 8, tot); CK(hipMemcpy(&h, tot, 8, hipMemcpyDeviceToHost)); }
        const double ops = (double)gb * 256 * per;
        printf("range %8u floats  %-26s %-50s %8.3f ms %8.1f Gop/s   sum %.0f (expect %.0f)\n", range, mode == 1 ? "4 consecutive per thread" : "random",
               names[sc], ms, ops / ms / 1e6, h, sc < 2 ? ops : 0.0);
      }
    }
  }
  return 0;
}
