// Shared host/device helpers for libneuma_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/neuma_hip.h"

// ---------------------------------------------------------------- error plumbing (host)
void nm_set_error(const char* fmt, ...);
#define NM_HIP_CHECK(expr)                                                                      \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      nm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
      return NM_ERR_HIP;                                                                        \
    }                                                                                           \
  } while (0)
#define NM_LAUNCH_CHECK()                                                                       \
  do {                                                                                          \
    hipError_t _e = hipGetLastError();                                                          \
    if (_e != hipSuccess) {                                                                     \
      nm_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return NM_ERR_HIP;                                                                        \
    }                                                                                           \
  } while (0)
#define NM_REQUIRE(cond, msg)                                                                   \
  do {                                                                                          \
    if (!(cond)) {                                                                              \
      nm_set_error("invalid argument: %s (%s:%d)", msg, __FILE__, __LINE__);                    \
      return NM_ERR_INVALID;                                                                    \
    }                                                                                           \
  } while (0)

// kernel timing hooks (nm_api.hip)
void nm_prof_begin(const char* name, hipStream_t s);
void nm_prof_end(const char* name, hipStream_t s);
extern int g_nm_prof_on;
// launch wrapper: NM_LAUNCH(kernel, grid, block, shmem, stream, args...)
#define NM_LAUNCH(kern, grid, block, shmem, stream, ...)                         \
  do {                                                                           \
    if (g_nm_prof_on) nm_prof_begin(#kern, (stream));                            \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);           \
    if (g_nm_prof_on) nm_prof_end(#kern, (stream));                              \
  } while (0)

// internal cross-file helpers
float nm_mpm_get_dt(const nm_mpm* h);
struct nm_mpm_view {   // what nm_shard.hip needs to see of a grid handle (valid until the next scatter)
  float4* gm;          // {mv.xyz, m} per node, 64 nodes per block
  float4* gg;          // adjoint scratch per node
  int* flags;          // per block: == epoch when the block is in the current active list
  int* list;           // current active-block list
  int* count;          // its length (device)
  int epoch, nblocks;
};
nm_mpm_view nm_mpm_get_view(nm_mpm* h);
void nm_mpm_set_fresh_rows(nm_mpm* h, int on);   // g2p writes a fresh state's rows for disabled particles (roll-out checkpoints)
int nm_mpm_shared_counters(nm_mpm* h, int** cnt, int** pos);
int nm_mpm_xchg_arrays(nm_mpm* h, int** slot, int** dil, int* new_tag);   // frame-level exchange: slot[] (-1 between frames), dil[]
int nm_mpm_grid_dims(const nm_mpm* h);                                    // blocks per axis
int nm_mpm_dil_tag(const nm_mpm* h);
int nm_shard_slots(nm_mpm* h, const int32_t* shared, int32_t cap_shared, int assign, void* stream);
int nm_shard_pack_fwd(nm_mpm* h, const int32_t* shared, int32_t cap_shared, float* buf, unsigned char* mine, int32_t* status,
                      void* stream);
int nm_shard_pack_bwd(nm_mpm* h, const int32_t* shared, int32_t cap_shared, float* buf, const unsigned char* mine, void* stream);   // [nblocks] each: 0 / INT_MAX between exchanges
int nm_material_prepare(const nm_mlp* w, float* wperm, void* stream);   // weights -> MFMA operand order, once per roll-out
int nm_material_prepare2(const nm_mlp* wa, float* wperm_a, const nm_mlp* wb, float* wperm_b, void* stream);   // two nets, one launch
size_t nm_material_prepared_floats();
int nm_material_wgrad_reduce(const float* wpart, int32_t n, float* gw0, float* gw1, float* gw2, int accumulate, void* stream);
int nm_material_wgrad_reduce2(const float* wpart_a, const float* wpart_b, int32_t n, float* gw_a, float* gw_b, void* stream);

static inline int nm_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- 3x3 helpers (host+device, row-major)
#define NM_HD __host__ __device__ __forceinline__
// torch.nan_to_num(x, nan=0, posinf=0, neginf=0) for one value (NaN fails the comparison, +-Inf exceed FLT_MAX)
__device__ __forceinline__ float nm_finite_or_zero(float v) { return fabsf(v) <= 3.402823466e+38f ? v : 0.f; }

struct M3 {
  float m[9];
  NM_HD float& operator()(int r, int c) { return m[r * 3 + c]; }
  NM_HD float operator()(int r, int c) const { return m[r * 3 + c]; }
};

NM_HD M3 m3_load(const float* __restrict__ p) {
  M3 a;
#pragma unroll
  for (int i = 0; i < 9; ++i) a.m[i] = p[i];
  return a;
}
NM_HD void m3_store(float* __restrict__ p, const M3& a) {
#pragma unroll
  for (int i = 0; i < 9; ++i) p[i] = a.m[i];
}
NM_HD M3 m3_zero() {
  M3 a;
#pragma unroll
  for (int i = 0; i < 9; ++i) a.m[i] = 0.f;
  return a;
}
NM_HD M3 m3_ident() {
  M3 a = m3_zero();
  a.m[0] = a.m[4] = a.m[8] = 1.f;
  return a;
}
// C = A B
NM_HD M3 m3_mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      C.m[r * 3 + c] = A.m[r * 3] * B.m[c] + A.m[r * 3 + 1] * B.m[3 + c] + A.m[r * 3 + 2] * B.m[6 + c];
  return C;
}
// C = A B^T
NM_HD M3 m3_mul_nt(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      C.m[r * 3 + c] = A.m[r * 3] * B.m[c * 3] + A.m[r * 3 + 1] * B.m[c * 3 + 1] + A.m[r * 3 + 2] * B.m[c * 3 + 2];
  return C;
}
// C = A^T B
NM_HD M3 m3_mul_tn(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      C.m[r * 3 + c] = A.m[r] * B.m[c] + A.m[3 + r] * B.m[3 + c] + A.m[6 + r] * B.m[6 + c];
  return C;
}
NM_HD M3 m3_transpose(const M3& A) {
  M3 C;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C.m[r * 3 + c] = A.m[c * 3 + r];
  return C;
}
NM_HD float m3_det(const M3& A) {
  return A.m[0] * (A.m[4] * A.m[8] - A.m[5] * A.m[7]) - A.m[1] * (A.m[3] * A.m[8] - A.m[5] * A.m[6]) +
         A.m[2] * (A.m[3] * A.m[7] - A.m[4] * A.m[6]);
}
// cofactor matrix: d det(A) / dA
NM_HD M3 m3_cofactor(const M3& A) {
  M3 C;
  C.m[0] = A.m[4] * A.m[8] - A.m[5] * A.m[7];
  C.m[1] = A.m[5] * A.m[6] - A.m[3] * A.m[8];
  C.m[2] = A.m[3] * A.m[7] - A.m[4] * A.m[6];
  C.m[3] = A.m[2] * A.m[7] - A.m[1] * A.m[8];
  C.m[4] = A.m[0] * A.m[8] - A.m[2] * A.m[6];
  C.m[5] = A.m[1] * A.m[6] - A.m[0] * A.m[7];
  C.m[6] = A.m[1] * A.m[5] - A.m[2] * A.m[4];
  C.m[7] = A.m[2] * A.m[3] - A.m[0] * A.m[5];
  C.m[8] = A.m[0] * A.m[4] - A.m[1] * A.m[3];
  return C;
}

// ---------------------------------------------------------------- 3x3 SVD (one-sided Jacobi)
// A = U diag(s) V^T with U, V in SO(3), s0 >= s1 >= |s2|, sign(s2) = sign(det A): the convention
// modules/nclaw/warp/svd.py:61-96 produces from wp.svd3 + its det fix-up.
// Hestenes rotations act on the columns of B = A V directly (no A^T A squaring), so small singular
// values keep full relative accuracy.
// fast 1-ulp device transcendentals (v_rcp_f32 / v_rsq_f32 / v_sqrt_f32): the IEEE-rounded forms expand to 10-15 VALU
// instructions each and the Jacobi iteration is self-correcting
#if defined(__HIP_DEVICE_COMPILE__)
#define NM_RCP(x) __builtin_amdgcn_rcpf(x)
#define NM_RSQ(x) __builtin_amdgcn_rsqf(x)
#define NM_SQRT(x) __builtin_amdgcn_sqrtf(x)
#else
#define NM_RCP(x) (1.f / (x))
#define NM_RSQ(x) (1.f / sqrtf(x))
#define NM_SQRT(x) sqrtf(x)
#endif
// one Hestenes rotation of columns p,q.  With a = |q|^2 - |p|^2, b = 2 p.q the tangent of the rotation angle is
//   t = sgn(a) b / (|a| + sqrt(a^2 + b^2))          (the smaller root of t^2 + 2 (a/b) t - 1 = 0, division-free form)
// Returns whether the pair was already orthogonal to fp32 rounding before the rotation: |p.q| <= 5e-7 |p| |q|.
NM_HD bool nm_jacobi_pair(float* __restrict__ B, float* __restrict__ V, int p, int q) {
  float bp0 = B[p], bp1 = B[3 + p], bp2 = B[6 + p];
  float bq0 = B[q], bq1 = B[3 + q], bq2 = B[6 + q];
  float alpha = bp0 * bp0 + bp1 * bp1 + bp2 * bp2;
  float beta = bq0 * bq0 + bq1 * bq1 + bq2 * bq2;
  float gamma = bp0 * bq0 + bp1 * bq1 + bp2 * bq2;
  float a = beta - alpha, b = gamma + gamma;
  float den = fabsf(a) + NM_SQRT(a * a + b * b);
  float t = b * NM_RCP(fmaxf(den, 1e-37f));
  t = a < 0.f ? -t : t;
  float c = NM_RSQ(1.f + t * t);
  float s = c * t;
  B[p] = c * bp0 - s * bq0; B[3 + p] = c * bp1 - s * bq1; B[6 + p] = c * bp2 - s * bq2;
  B[q] = s * bp0 + c * bq0; B[3 + q] = s * bp1 + c * bq1; B[6 + q] = s * bp2 + c * bq2;
  float vp0 = V[p], vp1 = V[3 + p], vp2 = V[6 + p];
  float vq0 = V[q], vq1 = V[3 + q], vq2 = V[6 + q];
  V[p] = c * vp0 - s * vq0; V[3 + p] = c * vp1 - s * vq1; V[6 + p] = c * vp2 - s * vq2;
  V[q] = s * vp0 + c * vq0; V[3 + q] = s * vp1 + c * vq1; V[6 + q] = s * vp2 + c * vq2;
  return gamma * gamma <= 2.5e-13f * alpha * beta;
}
// swap columns p,q of B and V keeping det(V) = +1 (negate the column that moves up)
NM_HD void nm_swap_cols(float* __restrict__ B, float* __restrict__ V, int p, int q, bool doit) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float bp = B[3 * r + p], bq = B[3 * r + q];
    B[3 * r + p] = doit ? bq : bp;
    B[3 * r + q] = doit ? -bp : bq;
    float vp = V[3 * r + p], vq = V[3 * r + q];
    V[3 * r + p] = doit ? vq : vp;
    V[3 * r + q] = doit ? -vp : vq;
  }
}
NM_HD void nm_svd3(const M3& A, M3& U, float s[3], M3& Vm) {
  float B[9], V[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { B[i] = A.m[i]; V[i] = (i % 4 == 0) ? 1.f : 0.f; }
  // at most 5 sweeps (enough for fp32 on any input); a particle stops as soon as a whole sweep found all three pairs
  // orthogonal to rounding, so the result of a particle never depends on its wave-mates
#pragma unroll 1
  for (int sweep = 0; sweep < 5; ++sweep) {
    bool c01 = nm_jacobi_pair(B, V, 0, 1);
    bool c02 = nm_jacobi_pair(B, V, 0, 2);
    bool c12 = nm_jacobi_pair(B, V, 1, 2);
    if (c01 && c02 && c12) break;
  }
  float n0 = B[0] * B[0] + B[3] * B[3] + B[6] * B[6];
  float n1 = B[1] * B[1] + B[4] * B[4] + B[7] * B[7];
  float n2 = B[2] * B[2] + B[5] * B[5] + B[8] * B[8];
  // sort descending by column norm (3-element network), proper-rotation preserving swaps
  bool sw = n0 < n1; nm_swap_cols(B, V, 0, 1, sw); { float a = sw ? n1 : n0, b = sw ? n0 : n1; n0 = a; n1 = b; }
  sw = n0 < n2;      nm_swap_cols(B, V, 0, 2, sw); { float a = sw ? n2 : n0, b = sw ? n0 : n2; n0 = a; n2 = b; }
  sw = n1 < n2;      nm_swap_cols(B, V, 1, 2, sw); { float a = sw ? n2 : n1, b = sw ? n1 : n2; n1 = a; n2 = b; }
  float s0 = NM_SQRT(n0), s1 = NM_SQRT(n1);
  float u00, u10, u20, u01, u11, u21;
  if (s0 > 1e-30f) { float r = NM_RSQ(n0); u00 = B[0] * r; u10 = B[3] * r; u20 = B[6] * r; }
  else { u00 = 1.f; u10 = 0.f; u20 = 0.f; }
  // second column: remove any residual component along u0 (also the rank-1 guard)
  float d = u00 * B[1] + u10 * B[4] + u20 * B[7];
  float c0 = B[1] - d * u00, c1 = B[4] - d * u10, c2 = B[7] - d * u20;
  float cn2 = c0 * c0 + c1 * c1 + c2 * c2;
  if (cn2 > 1e-37f && s1 > 1e-20f * s0) { float r = NM_RSQ(cn2); u01 = c0 * r; u11 = c1 * r; u21 = c2 * r; }
  else {  // pick any unit vector orthogonal to u0
    float ax = fabsf(u00), ay = fabsf(u10), az = fabsf(u20);
    float e0 = (ax <= ay && ax <= az) ? 1.f : 0.f, e1 = (e0 == 0.f && ay <= az) ? 1.f : 0.f, e2 = 1.f - e0 - e1;
    float dd = e0 * u00 + e1 * u10 + e2 * u20;
    c0 = e0 - dd * u00; c1 = e1 - dd * u10; c2 = e2 - dd * u20;
    float r = NM_RSQ(c0 * c0 + c1 * c1 + c2 * c2);
    u01 = c0 * r; u11 = c1 * r; u21 = c2 * r;
  }
  // third column = u0 x u1 (det U = +1 by construction); signed sigma2 = b2 . u2
  float u02 = u10 * u21 - u20 * u11, u12 = u20 * u01 - u00 * u21, u22 = u00 * u11 - u10 * u01;
  float s2 = B[2] * u02 + B[5] * u12 + B[8] * u22;
  U.m[0] = u00; U.m[1] = u01; U.m[2] = u02;
  U.m[3] = u10; U.m[4] = u11; U.m[5] = u12;
  U.m[6] = u20; U.m[7] = u21; U.m[8] = u22;
  s[0] = s0; s[1] = s1; s[2] = s2;
#pragma unroll
  for (int i = 0; i < 9; ++i) Vm.m[i] = V[i];
}

// wave64 sum via cross-lane shuffles (result valid in every lane)
__device__ __forceinline__ float nm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
