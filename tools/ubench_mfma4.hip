// v_mfma_f32_4x4x1_16b_f32 on gfx950: register layout (incl. the cbsz / abid A-broadcast) and issue cost for ONE wave per
// SIMD and for two - the facts the 4-particle constitutive mini-tile is built on.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma4.hip -o /tmp/ubench_mfma4 && /tmp/ubench_mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------- layout
template <int CBSZ, int ABID>
__global__ void k_layout(const float* a, const float* b, float* d) {
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], c, CBSZ, ABID, 0);
  for (int i = 0; i < 4; ++i) d[threadIdx.x * 4 + i] = c[i];
}
template <int CBSZ, int ABID>
static int layout(const char* name, float* da, float* db, float* dd) {
  float a[64], b[64], d[256];
  for (int l = 0; l < 64; ++l) { a[l] = 1.f + l; b[l] = 100.f + l; }
  CK(hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b, sizeof(b), hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_layout<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
  CK(hipMemcpy(d, dd, sizeof(d), hipMemcpyDeviceToHost));
  // hypothesis: D[reg i] at lane (blk, j) = A[lane (src_blk, i)] * B[lane (blk, j)],
  // src_blk = blk (CBSZ 0) or (blk & ~(2^CBSZ - 1)) + ABID
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      int blk = l >> 2, src = CBSZ ? ((blk & ~((1 << CBSZ) - 1)) + ABID) : blk;
      float want = a[src * 4 + i] * b[l];
      if (d[l * 4 + i] != want) ++bad;
    }
  printf("layout %-24s cbsz %d abid %d : %s (%d mismatches)\n", name, CBSZ, ABID, bad ? "DIFFERENT" : "as assumed", bad);
  if (bad) {
    for (int l = 0; l < 8; ++l) printf("   lane %d: %g %g %g %g\n", l, d[l * 4], d[l * 4 + 1], d[l * 4 + 2], d[l * 4 + 3]);
  }
  return 0;
}

// ---------------------------------------------------------------- timing
#define REP4(x) x x x x
#define REP8(x) x x x x x x x x
#define REP16(x) REP4(REP4(x))

__device__ __forceinline__ float dpp_quad(float v, float old, const int sel, const int bank) {
  // lane l of every quad takes v from lane `sel` of its quad when (bank >> (l & 3)) & 1, else keeps old
  int r;
  switch (sel * 16 + bank) {
#define C(S, B) case S * 16 + B: r = __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), (S) | ((S) << 2) | ((S) << 4) | ((S) << 6), 0xf, B, false); break;
    C(0, 2) C(0, 4) C(0, 8) C(1, 1) C(1, 4) C(1, 8) C(2, 1) C(2, 2) C(2, 8) C(3, 1) C(3, 2) C(3, 4)
#undef C
    default: r = __float_as_int(old);
  }
  return __int_as_float(r);
}
// 4x4 transpose between the registers of an f4 and the lanes of every quad: out[r] at lane l = in[l] at lane r
__device__ __forceinline__ f4 quad_transpose(const f4 in) {
  f4 out = in;   // diagonal: out[r] at lane r = in[r] at lane r
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (l != r) out[r] = dpp_quad(in[l], out[r], r, 1 << l);
  return out;
}

__global__ void k_transpose_check(const float* in, float* out) {
  f4 v;
  for (int i = 0; i < 4; ++i) v[i] = in[threadIdx.x * 4 + i];
  f4 t = quad_transpose(v);
  for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = t[i];
}

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int reps, const float* wsrc) {
  float a0 = threadIdx.x * 1e-3f, b0 = 1.f - a0;
  f4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  float w[64];
  if (MODE >= 4) {
#pragma unroll
    for (int i = 0; i < 64; ++i) w[i] = wsrc[i * 64 + (threadIdx.x & 63)];
  }
  float sink = 0.f;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {          // 64 MFMAs, 4 accumulators round robin
      REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %5, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %4, %5, %1\n"
                         "v_mfma_f32_4x4x1_16b_f32 %2, %4, %5, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %4, %5, %3\n"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(a0), "v"(b0));)
    } else if (MODE == 1) {   // 64 MFMAs, 16 accumulators round robin
      REP4(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %16, %17, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %16, %17, %1\n"
                        "v_mfma_f32_4x4x1_16b_f32 %2, %16, %17, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %16, %17, %3\n"
                        "v_mfma_f32_4x4x1_16b_f32 %4, %16, %17, %4\n v_mfma_f32_4x4x1_16b_f32 %5, %16, %17, %5\n"
                        "v_mfma_f32_4x4x1_16b_f32 %6, %16, %17, %6\n v_mfma_f32_4x4x1_16b_f32 %7, %16, %17, %7\n"
                        "v_mfma_f32_4x4x1_16b_f32 %8, %16, %17, %8\n v_mfma_f32_4x4x1_16b_f32 %9, %16, %17, %9\n"
                        "v_mfma_f32_4x4x1_16b_f32 %10, %16, %17, %10\n v_mfma_f32_4x4x1_16b_f32 %11, %16, %17, %11\n"
                        "v_mfma_f32_4x4x1_16b_f32 %12, %16, %17, %12\n v_mfma_f32_4x4x1_16b_f32 %13, %16, %17, %13\n"
                        "v_mfma_f32_4x4x1_16b_f32 %14, %16, %17, %14\n v_mfma_f32_4x4x1_16b_f32 %15, %16, %17, %15\n"
                        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                          "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])
                        : "v"(a0), "v"(b0));)
    } else if (MODE == 2) {   // 64 MFMAs, ONE accumulator (dependent chain)
      REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n"
                         "v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n"
                         : "+v"(acc[0]) : "v"(a0), "v"(b0));)
    } else if (MODE == 3) {   // 64 MFMAs, two accumulators alternating
      REP16(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %2, %3, %1\n"
                         "v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %2, %3, %1\n"
                         : "+v"(acc[0]), "+v"(acc[1]) : "v"(a0), "v"(b0));)
    } else if (MODE == 4 || MODE == 5 || MODE == 8) {
      // a 64 x 64 layer on a 4-particle mini-tile as the kernels would run it: A = activations (4 registers, abid picks
      // the block = 16 k values per register), B = one weight register per k, 4 accumulators round robin.
      // MODE 5: accumulators (and so C/D) in AGPRs.  MODE 8: 8 accumulators.
      f4 t = {a0, b0, a0 + 1.f, b0 + 1.f};
#define L4(AB, K0)                                                                                                   \
      asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:4 abid:" #AB "\n"                                    \
                   "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:4 abid:" #AB "\n"                                    \
                   "v_mfma_f32_4x4x1_16b_f32 %2, %6, %10, %2 cbsz:4 abid:" #AB "\n"                                   \
                   "v_mfma_f32_4x4x1_16b_f32 %3, %7, %11, %3 cbsz:4 abid:" #AB "\n"                                   \
                   : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                           \
                   : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(w[K0]), "v"(w[K0 + 1]), "v"(w[K0 + 2]), "v"(w[K0 + 3]));
#define L4A(AB, K0)                                                                                                  \
      asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:4 abid:" #AB "\n"                                    \
                   "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:4 abid:" #AB "\n"                                    \
                   "v_mfma_f32_4x4x1_16b_f32 %2, %6, %10, %2 cbsz:4 abid:" #AB "\n"                                   \
                   "v_mfma_f32_4x4x1_16b_f32 %3, %7, %11, %3 cbsz:4 abid:" #AB "\n"                                   \
                   : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3])                                           \
                   : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(w[K0]), "v"(w[K0 + 1]), "v"(w[K0 + 2]), "v"(w[K0 + 3]));
#define L4B(AB, K0)                                                                                                  \
      asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:4 abid:" #AB "\n"                                    \
                   "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:4 abid:" #AB "\n"                                    \
                   "v_mfma_f32_4x4x1_16b_f32 %2, %6, %10, %2 cbsz:4 abid:" #AB "\n"                                   \
                   "v_mfma_f32_4x4x1_16b_f32 %3, %7, %11, %3 cbsz:4 abid:" #AB "\n"                                   \
                   : "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])                                           \
                   : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(w[K0]), "v"(w[K0 + 1]), "v"(w[K0 + 2]), "v"(w[K0 + 3]));
      if (MODE == 4) {
        L4(0, 0) L4(1, 4) L4(2, 8) L4(3, 12) L4(4, 16) L4(5, 20) L4(6, 24) L4(7, 28)
        L4(8, 32) L4(9, 36) L4(10, 40) L4(11, 44) L4(12, 48) L4(13, 52) L4(14, 56) L4(15, 60)
      } else if (MODE == 5) {
        L4A(0, 0) L4A(1, 4) L4A(2, 8) L4A(3, 12) L4A(4, 16) L4A(5, 20) L4A(6, 24) L4A(7, 28)
        L4A(8, 32) L4A(9, 36) L4A(10, 40) L4A(11, 44) L4A(12, 48) L4A(13, 52) L4A(14, 56) L4A(15, 60)
      } else {
        L4(0, 0) L4B(1, 4) L4(2, 8) L4B(3, 12) L4(4, 16) L4B(5, 20) L4(6, 24) L4B(7, 28)
        L4(8, 32) L4B(9, 36) L4(10, 40) L4B(11, 44) L4(12, 48) L4B(13, 52) L4(14, 56) L4B(15, 60)
      }
    } else if (MODE == 6) {   // baseline: 16 x v_mfma_f32_16x16x4_f32 (the same flops as 64 of the above), 4 accumulators
      REP4(asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                        "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(a0), "v"(b0));)
    } else if (MODE == 7) {   // 16 quad transposes (12 DPP moves each), dependent through the data
      f4 t = acc[0] + (f4){a0, b0, a0, b0};
      REP16(t = quad_transpose(t); asm volatile("" : "+v"(t));)
      acc[0] = t;
    } else if (MODE == 9) {   // mini-tile layer as a whole: transpose, 64 MFMAs (4 acc), combine, 2 packed GELU-sized VALU blocks
      f4 d = acc[0];
      f4 t = quad_transpose(d);
      f4 q[4] = {(f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}, (f4){0.f, 0.f, 0.f, 0.f}};
#define M9(AB, K0)                                                                                                   \
      q[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(t[0], w[K0], q[0], 4, AB, 0);                                         \
      q[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(t[1], w[K0 + 1], q[1], 4, AB, 0);                                     \
      q[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(t[2], w[K0 + 2], q[2], 4, AB, 0);                                     \
      q[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(t[3], w[K0 + 3], q[3], 4, AB, 0);
      M9(0, 0) M9(1, 4) M9(2, 8) M9(3, 12) M9(4, 16) M9(5, 20) M9(6, 24) M9(7, 28)
      M9(8, 32) M9(9, 36) M9(10, 40) M9(11, 44) M9(12, 48) M9(13, 52) M9(14, 56) M9(15, 60)
      f4 s = (q[0] + q[1]) + (q[2] + q[3]);
      // GELU-sized VALU work on 4 values (two packed pairs): ~21 instructions per pair
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        f2 x = {s[2 * pr], s[2 * pr + 1]};
        f2 u = x * x * (f2){-0.72f, -0.72f};
        f2 tt = {__builtin_amdgcn_rcpf(fmaf(0.23f, fabsf(x[0]), 1.f)), __builtin_amdgcn_rcpf(fmaf(0.23f, fabsf(x[1]), 1.f))};
        f2 e = {__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])};
        f2 p = __builtin_elementwise_fma(tt, (f2){0.53f, 0.53f}, (f2){-0.72f, -0.72f});
        p = __builtin_elementwise_fma(p, tt, (f2){0.71f, 0.71f});
        p = __builtin_elementwise_fma(p, tt, (f2){-0.14f, -0.14f});
        p = __builtin_elementwise_fma(p, tt, (f2){0.127f, 0.127f});
        f2 ht = p * tt * e;
        f2 qq = (f2){0.5f, 0.5f} - ht;
        f2 rr = {__builtin_copysignf(qq[0], x[0]), __builtin_copysignf(qq[1], x[1])};
        f2 P = rr + (f2){0.5f, 0.5f};
        f2 h = x * P;
        f2 dh = __builtin_elementwise_fma(x, e * (f2){0.3989f, 0.3989f}, P);
        s[2 * pr] = h[0] + 1e-3f * dh[0];
        s[2 * pr + 1] = h[1] + 1e-3f * dh[1];
      }
      acc[0] = s;
    }
  }
  long long t1 = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = sink;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static int run(const char* name, double ninstr, float* out, long long* cyc, const float* w, int blocks) {
  const int reps = 64;
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps, w);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps * 16, w);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  static long long h[4096];
  CK(hipMemcpy(h, cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost));
  double s = 0;
  for (int i = 0; i < blocks * 4; ++i) s += h[i];
  s /= blocks * 4;
  printf("%-72s wg/CU %d: %8.0f ticks/rep %7.2f ticks/instr   (wall %.1f us for %d reps)\n", name, blocks / 256, s / (reps * 16),
         s / (reps * 16) / ninstr, ms * 1e3, reps * 16);
  return 0;
}

int main() {
  float *da, *db, *dd, *out, *w; long long* cyc;
  CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dd, 1024));
  CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 4096 * 8)); CK(hipMalloc(&w, 64 * 64 * 4));
  CK(hipMemset(w, 0, 64 * 64 * 4));
  layout<0, 0>("no broadcast", da, db, dd);
  layout<4, 0>("A block 0 -> all", da, db, dd);
  layout<4, 5>("A block 5 -> all", da, db, dd);
  layout<4, 15>("A block 15 -> all", da, db, dd);
  layout<2, 1>("A block 4g+1 -> group g", da, db, dd);
  layout<2, 3>("A block 4g+3 -> group g", da, db, dd);
  layout<3, 6>("A block 8g+6 -> half g", da, db, dd);
  {   // transpose check
    float in[256], o[256];
    for (int i = 0; i < 256; ++i) in[i] = (float)i;
    CK(hipMemcpy(out, in, sizeof(in), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_transpose_check, dim3(1), dim3(64), 0, 0, out, out + 256);
    CK(hipMemcpy(o, out + 256, sizeof(o), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        int q = l & ~3, ll = l & 3;     // out[r] at lane (q, ll) = in[ll] at lane (q, r)
        if (o[l * 4 + r] != in[(q + r) * 4 + ll]) ++bad;
      }
    printf("quad transpose by 12 DPP moves: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
  }
  for (int blocks = 256; blocks <= 512; blocks += 256) {
    run<6>("16 x v_mfma_f32_16x16x4_f32, 4 accumulators (baseline)", 16, out, cyc, w, blocks);
    run<0>("64 x v_mfma_f32_4x4x1_16b_f32, 4 accumulators", 64, out, cyc, w, blocks);
    run<1>("64 x 4x4x1, 16 accumulators", 64, out, cyc, w, blocks);
    run<3>("64 x 4x4x1, 2 accumulators", 64, out, cyc, w, blocks);
    run<2>("64 x 4x4x1, ONE accumulator (dependent)", 64, out, cyc, w, blocks);
    run<4>("64 x 4x4x1 cbsz:4 abid 0..15, B = 64 distinct registers, 4 acc", 64, out, cyc, w, blocks);
    run<8>("the same, 8 accumulators", 64, out, cyc, w, blocks);
    run<5>("the same, 4 accumulators in AGPRs", 64, out, cyc, w, blocks);
    run<7>("16 quad transposes (192 DPP moves)", 192, out, cyc, w, blocks);
    run<9>("whole mini-tile layer: transpose + 64 MFMA + combine + 2 packed GELUs", 1, out, cyc, w, blocks);
  }
  return 0;
}
