// Micro-benchmark: how one wave per SIMD sustains v_mfma_f32_16x16x4_f32 with and without interleaved GELU VALU work
// and LDS-sourced A operands (numbers feed DESIGN.md: what bounds the constitutive-net kernels).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip -o tools/ubench_mfma && tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void gelu_both(float x, float& h, float& dh) {
  const float ax = fabsf(x);
  const float e = __expf(-0.5f * x * x);
  const float t = __frcp_rn(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float half_tail = 0.5f * poly * t * e;
  float Phi = x >= 0.f ? 1.0f - half_tail : half_tail;
  h = x * Phi;
  dh = fmaf(x, 0.3989422804014327f * e, Phi);
}

// MODE bit0: A operands from LDS; bit1: one gelu per 4 MFMAs on independent data; NACC accumulators in rotation
template <int MODE, int NACC, int VPER = 6>
__global__ void __launch_bounds__(256) k_mfma(float* out, long long* cyc, int reps) {
  __shared__ float sW[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) sW[i] = 0.001f * (i & 127);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  float b = 0.01f * lane, a = 0.02f * lane;
  float gx = 0.1f * lane, gh = 0.f, gd = 0.f;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      float av = (MODE & 1) ? sW[k * 64 + lane] : a;
      acc[k % NACC] = MFMA(av, b, acc[k % NACC]);
      if ((MODE & 2) && (k & 3) == 3) {
        float h, d;
        gelu_both(gx, h, d);
        gh += h; gd += d; gx += 0.001f;
      }
    }
    if (MODE & 4) {
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);
      }
    }
  }
  long long t1 = clock64();
  float s = gh + gd;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// bf16 matrix-core variant: v_mfma_f32_16x16x32_bf16, optional gelu per 4 MFMAs (MODE bit1), interleave hint (bit2)
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int MODE, int VPER = 6>
__global__ void __launch_bounds__(256) k_mfma_bf16(float* out, long long* cyc, int reps) {
  const int lane = threadIdx.x & 63;
  f4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
  float gx = 0.1f * lane, gh = 0.f, gd = 0.f;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      acc[k % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k % 4], 0, 0, 0);
      if ((MODE & 2) && (k & 3) == 3) {
        float h, d;
        gelu_both(gx, h, d);
        gh += h; gd += d; gx += 0.001f;
      }
    }
    if (MODE & 4) {
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);
      }
    }
  }
  long long t1 = clock64();
  float s = gh + gd;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// fast-math GELU (1-ulp rcp / exp2) and a two-at-a-time version written on float2 vectors (v_pk_*_f32)
__device__ __forceinline__ void gelu_fast(float x, float& h, float& dh) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float half_tail = 0.5f * poly * t * e;
  float Phi = x >= 0.f ? 1.0f - half_tail : half_tail;
  h = x * Phi;
  dh = fmaf(x, 0.3989422804014327f * e, Phi);
}
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_pk(f2 x, f2& h, f2& dh) {
  const f2 ax = {fabsf(x.x), fabsf(x.y)};
  const f2 xx = x * x * (-0.5f * 1.4426950408889634f);
  const f2 e = {__builtin_amdgcn_exp2f(xx.x), __builtin_amdgcn_exp2f(xx.y)};
  const f2 den = ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
  const f2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  f2 poly = t * 1.061405429f + (-1.453152027f);
  poly = poly * t + 1.421413741f;
  poly = poly * t + (-0.284496736f);
  poly = poly * t + 0.254829592f;
  const f2 ht = poly * t * e * 0.5f;
  const f2 one_m = 1.0f - ht;
  f2 Phi = {x.x >= 0.f ? one_m.x : ht.x, x.y >= 0.f ? one_m.y : ht.y};
  h = x * Phi;
  dh = x * (e * 0.3989422804014327f) + Phi;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_gelu2(float* out, long long* cyc, int reps) {
  const int lane = threadIdx.x & 63;
  float gx = 0.1f * lane, gh = 0.f, gd = 0.f;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      if (MODE == 0) {
        float h, d;
        gelu_fast(gx, h, d); gh += h; gd += d; gx += 0.001f;
        gelu_fast(gx, h, d); gh += h; gd += d; gx += 0.001f;
      } else {
        f2 x2 = {gx, gx + 0.001f}, h2, d2;
        gelu_pk(x2, h2, d2);
        gh += h2.x + h2.y; gd += d2.x + d2.y; gx += 0.002f;
      }
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = gh + gd;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// gelu only
__global__ void __launch_bounds__(256) k_gelu(float* out, long long* cyc, int reps) {
  const int lane = threadIdx.x & 63;
  float gx = 0.1f * lane, gh = 0.f, gd = 0.f;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float h, d;
      gelu_both(gx, h, d);
      gh += h; gd += d; gx += 0.001f;
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = gh + gd;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  float* out; CK(hipMalloc(&out, 4 << 20));
  long long* cyc; CK(hipMalloc(&cyc, 8 * 4096));
  const int reps = 16;
  std::vector<long long> h(4096);
  auto report = [&](const char* name, int blocks, double mfmas, double gelus) {
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, 8 * blocks * 4, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks * 4; ++i) s += h[i];
    s /= blocks * 4;
    printf("%-44s blocks %4d: %9.0f cycles/wave  %.1f cyc/MFMA  %.1f cyc/gelu\n", name, blocks, s, mfmas > 0 ? s / mfmas : 0.0, gelus > 0 ? s / gelus : 0.0);
    return 0;
  };
  for (int blocks : {256, 512}) {
    for (int it = 0; it < 2; ++it) {
      hipLaunchKernelGGL((k_mfma<0, 4>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("mfma regs, 4 acc", blocks, reps * 64.0, 0);
      hipLaunchKernelGGL((k_mfma<0, 16>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("mfma regs, 16 acc", blocks, reps * 64.0, 0);
      hipLaunchKernelGGL((k_mfma<1, 4>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("mfma A from LDS, 4 acc", blocks, reps * 64.0, 0);
      hipLaunchKernelGGL((k_mfma<2, 4>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("mfma regs + gelu/4, 4 acc", blocks, reps * 64.0, reps * 16.0);
      hipLaunchKernelGGL((k_mfma<3, 4>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("mfma LDS + gelu/4, 4 acc", blocks, reps * 64.0, reps * 16.0);
      hipLaunchKernelGGL((k_mfma<6, 4, 5>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("mfma regs + gelu/4, interleaved 1:5", blocks, reps * 64.0, reps * 16.0);
      hipLaunchKernelGGL((k_mfma<6, 4, 6>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("mfma regs + gelu/4, interleaved 1:6", blocks, reps * 64.0, reps * 16.0);
      hipLaunchKernelGGL((k_mfma<6, 4, 8>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("mfma regs + gelu/4, interleaved 1:8", blocks, reps * 64.0, reps * 16.0);
      hipLaunchKernelGGL((k_mfma<7, 4, 6>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("mfma LDS + gelu/4, interleaved 1:6", blocks, reps * 64.0, reps * 16.0);
      hipLaunchKernelGGL((k_mfma_bf16<0>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("bf16 16x16x32 mfma only", blocks, reps * 64.0, 0);
      hipLaunchKernelGGL((k_mfma_bf16<2>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("bf16 mfma + gelu/4", blocks, reps * 64.0, reps * 16.0);
      hipLaunchKernelGGL((k_mfma_bf16<6, 6>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("bf16 mfma + gelu/4 interleaved 1:6", blocks, reps * 64.0, reps * 16.0);
      hipLaunchKernelGGL((k_gelu2<0>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("gelu fast (rcp/exp2 1 ulp)", blocks, 0, reps * 16.0);
      hipLaunchKernelGGL((k_gelu2<1>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("gelu fast, packed float2", blocks, 0, reps * 16.0);
      hipLaunchKernelGGL(k_gelu, dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
      if (it) report("gelu only", blocks, 0, reps * 16.0);
    }
  }
  return 0;
}
