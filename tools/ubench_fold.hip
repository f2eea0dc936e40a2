// wave_fold8_swap against plain sums:  hipcc --offload-arch=gfx950 -O3 tools/ubench_fold.hip -o tools/ubench_fold && tools/ubench_fold
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_fold8_swap(const float* v, int lane) {
  float a[4], c[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
    a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 2]), false, false);
    c[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const bool b3 = lane & 8;
  const float keep = b3 ? c[1] : c[0], send = b3 ? c[0] : c[1];
  float d = keep + dpp_mov<0x128>(send);
  d += dpp_mov<0xB1>(d);
  d += dpp_mov<0x4E>(d);
  d += dpp_mov<0x141>(d);
  return d;
}
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = (float)((i + 1) * 1000 + l);
  out[l] = wave_fold8_swap(v, l);
}
int main() {
  float* d; hipMalloc(&d, 64 * 4);
  k<<<1, 64>>>(d);
  float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 8) printf("lane %2d: %.0f  (value %d expected %.0f)\n", l, h[l], l >> 3, 64.0 * ((l >> 3) + 1) * 1000 + 2016);
  return 0;
}
