// Issue / latency cost of VALU instructions for ONE wave per SIMD on gfx950 (the constitutive kernels' situation).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int reps) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  float c = 0.999f, d = 1e-3f;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {          // 8 independent fma chains, round robin: 64 instructions
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    } else if (MODE == 1) {   // one dependent chain: 64 instructions
      REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(c), "v"(d));)
    } else if (MODE == 2) {   // two chains
      REP8(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                        "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                        : "+v"(a0), "+v"(a1) : "v"(c), "v"(d));)
    } else if (MODE == 3) {   // 8 independent v_exp
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 4) {   // dependent v_exp chain
      REP64(asm volatile("v_exp_f32 %0, %0\n" : "+v"(a0));)
    } else if (MODE == 5) {   // exp followed by dependent fma, 8 independent pairs: 32 exp + 32 fma
      REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %0, %0, %4, %5\n v_exp_f32 %1, %1\n v_fma_f32 %1, %1, %4, %5\n"
                        "v_exp_f32 %2, %2\n v_fma_f32 %2, %2, %4, %5\n v_exp_f32 %3, %3\n v_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));)
    } else if (MODE == 6) {   // v_cndmask + v_cmp pairs, independent
      REP8(asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_le_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n"
                        "v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_le_f32 vcc, %1, %0\n v_cndmask_b32 %5, %5, %4, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : : "vcc");)
    } else if (MODE == 7) {   // v_mul with literal constant (v_mul_f32 + 32-bit literal), independent
      REP8(asm volatile("v_mul_f32 %0, 0x3ecc422a, %0\n v_mul_f32 %1, 0x3ecc422a, %1\n v_mul_f32 %2, 0x3ecc422a, %2\n v_mul_f32 %3, 0x3ecc422a, %3\n"
                        "v_mul_f32 %4, 0x3ecc422a, %4\n v_mul_f32 %5, 0x3ecc422a, %5\n v_mul_f32 %6, 0x3ecc422a, %6\n v_mul_f32 %7, 0x3ecc422a, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 8) {   // v_fmaak (literal) independent
      REP8(asm volatile("v_fmaak_f32 %0, %0, %8, 0x3fb5f0e3\n v_fmaak_f32 %1, %1, %8, 0x3fb5f0e3\n v_fmaak_f32 %2, %2, %8, 0x3fb5f0e3\n v_fmaak_f32 %3, %3, %8, 0x3fb5f0e3\n"
                        "v_fmaak_f32 %4, %4, %8, 0x3fb5f0e3\n v_fmaak_f32 %5, %5, %8, 0x3fb5f0e3\n v_fmaak_f32 %6, %6, %8, 0x3fb5f0e3\n v_fmaak_f32 %7, %7, %8, 0x3fb5f0e3\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 9) {   // v_pk_fma_f32 independent (each = 2 fma)
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pc = {c, c}, pd = {d, d};
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc), "v"(pd));)
      a0 = p0[0] + p0[1]; a2 = p1[0] + p1[1]; a4 = p2[0] + p2[1]; a6 = p3[0] + p3[1];
    } else if (MODE == 10) {  // accvgpr write + read, independent
      REP8(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n"
                        "v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");)
    } else if (MODE == 11) {  // 8 independent v_rcp
      REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                        "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 12) {  // mfma 16x16x4 f32, 4 accumulators round robin, then 8 independent fma between each: sum or overlap?
      typedef float f4 __attribute__((ext_vector_type(4)));
      static __shared__ float dummy;
      f4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %12, %13, %0\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %12, %13\n"
                        "v_mfma_f32_16x16x4_f32 %1, %12, %13, %1\n v_fma_f32 %6, %6, %12, %13\n v_fma_f32 %7, %7, %12, %13\n"
                        "v_mfma_f32_16x16x4_f32 %2, %12, %13, %2\n v_fma_f32 %8, %8, %12, %13\n v_fma_f32 %9, %9, %12, %13\n"
                        "v_mfma_f32_16x16x4_f32 %3, %12, %13, %3\n v_fma_f32 %10, %10, %12, %13\n v_fma_f32 %11, %11, %12, %13\n"
                        : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                        : "v"(c), "v"(d));)
      a0 += acc0[0] + acc1[1] + acc2[2] + acc3[3];
      (void)dummy;
    } else if (MODE == 13) {  // mfma only, 4 accumulators (32 MFMAs)
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                        "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                        : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(c), "v"(d));)
      a0 += acc0[0] + acc1[1] + acc2[2] + acc3[3];
    } else if (MODE == 14) {  // mfma only, 8 accumulators, distinct A/B registers per MFMA (32 MFMAs)
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0, q4 = q0, q5 = q0, q6 = q0, q7 = q0;
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_mfma_f32_16x16x4_f32 %1, %10, %11, %1\n"
                        "v_mfma_f32_16x16x4_f32 %2, %12, %13, %2\n v_mfma_f32_16x16x4_f32 %3, %14, %15, %3\n"
                        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7)
                        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
      a0 += q0[0] + q1[1] + q2[2] + q3[3];
    } else if (MODE == 15) {  // mfma with accumulators in AGPRs ("a" constraint), 4 in rotation
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                        "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                        : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3) : "v"(c), "v"(d));)
      a0 += acc0[0] + acc1[1] + acc2[2] + acc3[3];
    } else if (MODE == 16) {  // 32x32x2 f32 (16 MFMAs = the flops of 32 16x16x4), 2 accumulators in rotation
      typedef float f16v __attribute__((ext_vector_type(16)));
      f16v r0 = {0}, r1 = {0};
      REP8(asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n v_mfma_f32_32x32x2_f32 %1, %2, %3, %1\n"
                        : "+v"(r0), "+v"(r1) : "v"(c), "v"(d));)
      a0 += r0[0] + r1[1];
    } else if (MODE == 17) {  // mfma 16x16x4 interleaved with one ds_read_b32 each (LDS in the shadow)
      typedef float f4 __attribute__((ext_vector_type(4)));
      __shared__ float lds[256];
      lds[threadIdx.x] = a0;
      f4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
      float l0, l1, l2, l3;
      const float* lp = lds + (threadIdx.x & 63);
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n ds_read_b32 %4, %10\n v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n ds_read_b32 %5, %10 offset:256\n"
                        "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n ds_read_b32 %6, %10 offset:512\n v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n ds_read_b32 %7, %10 offset:768\n s_waitcnt lgkmcnt(0)\n"
                        : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "=v"(l0), "=v"(l1), "=v"(l2), "=v"(l3)
                        : "v"(c), "v"(d), "v"((unsigned)(size_t)lp));)
      a0 += acc0[0] + acc1[1] + acc2[2] + acc3[3] + l0 + l1 + l2 + l3;
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static int run(const char* name, double ninstr, float* out, long long* cyc) {
  const int reps = 64, blocks = 256;
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
  CK(hipDeviceSynchronize());
  long long h[256];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double s = 0;
  for (int i = 0; i < blocks; ++i) s += h[i];
  s /= blocks;
  printf("%-64s %8.0f ticks/rep  %6.2f ticks/instr\n", name, s / reps, s / reps / ninstr);
  return 0;
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
  // clock64 tick vs wall: time a long kernel with events
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<1>), dim3(256), dim3(256), 0, 0, out, cyc, 20000);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<1>), dim3(256), dim3(256), 0, 0, out, cyc, 20000);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h0; CK(hipMemcpy(&h0, cyc, 8, hipMemcpyDeviceToHost));
  printf("clock64: %lld ticks in %.3f ms -> %.1f MHz tick rate (dependent fma chain, 1 wave/SIMD)\n", h0, ms, h0 / ms / 1e3);
  run<0>("v_fma_f32 x64, 8 independent chains", 64, out, cyc);
  run<2>("v_fma_f32 x64, 2 chains alternating", 64, out, cyc);
  run<1>("v_fma_f32 x64, ONE dependent chain", 64, out, cyc);
  run<7>("v_mul_f32 with 32-bit literal x64, independent", 64, out, cyc);
  run<8>("v_fmaak_f32 (literal) x64, independent", 64, out, cyc);
  run<9>("v_pk_fma_f32 x64 (=128 fma), 4 chains", 64, out, cyc);
  run<3>("v_exp_f32 x64, 8 independent", 64, out, cyc);
  run<11>("v_rcp_f32 x64, 8 independent", 64, out, cyc);
  run<4>("v_exp_f32 x64, ONE dependent chain", 64, out, cyc);
  run<5>("(v_exp ; dependent v_fma) x32, 4 independent pairs", 64, out, cyc);
  run<6>("(v_cmp ; v_cndmask) x32", 64, out, cyc);
  run<10>("v_accvgpr_write x32 + v_accvgpr_read x32", 64, out, cyc);
  run<13>("v_mfma_f32_16x16x4_f32 x32, 4 accumulators", 32, out, cyc);
  run<12>("(mfma ; 2 independent v_fma) x32  [96 instr]", 96, out, cyc);
  run<14>("v_mfma_f32_16x16x4_f32 x32, 4 acc, distinct A/B registers", 32, out, cyc);
  run<15>("v_mfma_f32_16x16x4_f32 x32, 4 accumulators in AGPRs", 32, out, cyc);
  run<16>("v_mfma_f32_32x32x2_f32 x16, 2 accumulators", 16, out, cyc);
  run<17>("(mfma ; ds_read_b32) x32, wait per 4", 32, out, cyc);
  return 0;
}
