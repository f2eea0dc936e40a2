// What v_permlane32_swap / v_permlane16_swap do to a register pair (x = lane, y = 100 + lane):
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_swap.hip -o tools/ubench_swap && tools/ubench_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  auto a = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
  out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"swap32 r[0]", "swap32 r[1]", "swap16 r[0]", "swap16 r[1]"};
  for (int q = 0; q < 4; ++q) { printf("%s:", nm[q]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 * q + l]); printf("\n"); }
  return 0;
}
