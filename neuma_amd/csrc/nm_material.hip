// Batched 3x3 SVD and the NeuMA neural constitutive nets (13 -> 64 -> 64 -> 9, GELU, no bias) for gfx950.
//
// Behaviour: /root/reference/modules/nclaw/warp/svd.py:61-96 (SVD convention),
//            /root/reference/modules/nclaw/material/meta.py:196-221 (elasticity), 468-489 (plasticity),
//            loralib.py:209-224 (LoRA enters as the merged weight W + (alpha/r) B A, computed by the host shim).
//
// Design: one kernel per direction.  A wave owns 64 particles: lane-per-particle VALU phases (SVD by
// one-sided Jacobi, invariants, R X F^T epilogue) sandwich an MFMA phase in which the three layers run as
// v_mfma_f32_16x16x4_f32 tiles over 16-particle column tiles.  Activations never leave registers between
// layers: the 16x16 accumulator layout (row = 4*(lane>>4)+reg, col = lane&15) is fed straight back as the
// next layer's B operand by permuting the K index of the weight operand instead (weights are stored in
// LDS once per workgroup in exactly the lane order each MFMA consumes, so every ds_read is linear and
// conflict free).  The backward kernel recomputes the forward, back-propagates in the same register
// layouts, and forms the weight gradients as MFMA outer products over the particle index through two
// small LDS transposes; per-workgroup partial sums are reduced by a second tiny kernel (deterministic).
#include "nm_common.h"
#include "nm_grid.h"
#include <cstddef>

typedef float f4 __attribute__((ext_vector_type(4)));
#ifdef NM_PHASES
__device__ long long g_nm_phase[3 * 8 * 2048];     // [0: elasticity adjoint / single forward, 1: plasticity adjoint, 2: forward pair][wave][phase]
extern "C" int nm_debug_phases(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nm_phase), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : -2;
}
#define NM_PH_DECL long long ph_t0 = clock64(); long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define NM_PH(i) { long long t1 = clock64(); ph[i] += t1 - ph_t0; ph_t0 = t1; }
#define NM_PH_STORE_K(kind) if ((threadIdx.x & 63) == 0) { int w = blockIdx.x * 4 + (threadIdx.x >> 6); if (w < 2048) for (int i = 0; i < 8; ++i) g_nm_phase[((kind) * 2048 + w) * 8 + i] = ph[i]; }
#define NM_PH_STORE NM_PH_STORE_K(0)
#else
#define NM_PH_STORE_K(kind)
#define NM_PH_DECL
#define NM_PH(i)
#define NM_PH_STORE
#endif
#define NM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define NM_SB_() __builtin_amdgcn_sched_barrier(0)

#define NM_BWD_GRID 256   // workgroups of the constitutive kernels = CUs
// A GridPrologue runs on NM_PRO_WGS extra workgroups (blockIdx >= pro.mat_grid) when the constitutive work leaves that many
// CUs idle - at 100k particles it fills 224 of 256 - so that its dependent-load chain delays nobody; otherwise the first
// workgroups do it before their own particles.  Evaluates to true for a workgroup that has nothing else to do.
#define NM_PRO_WGS 32
#define NM_PROLOGUE_SPLIT(pro) material_prologue(pro)
__device__ __forceinline__ bool material_prologue(const GridPrologue& pro) {
  if (pro.mode == 0) return false;
  const int mat = pro.mat_grid;
  if ((int)gridDim.x > mat) {
    if ((int)blockIdx.x < mat) return false;
    grid_prologue(pro, blockIdx.x - mat, gridDim.x - mat);
    return true;
  }
  grid_prologue(pro, blockIdx.x, gridDim.x);
  return false;
}
// forward pair kernel: mode 1 (clear + carry) only; the restore of mode 2 is not compiled in
__device__ __forceinline__ bool material_prologue_fwd(const GridPrologue& pro) {
  if (pro.mode == 0) return false;
  const int mat = pro.mat_grid;
  if ((int)gridDim.x > mat) {
    if ((int)blockIdx.x < mat) return false;
    grid_prologue_clear(pro, blockIdx.x - mat, gridDim.x - mat);
    return true;
  }
  grid_prologue_clear(pro, blockIdx.x, gridDim.x);
  return false;
}
#define NM_W0 (64 * 13)
#define NM_W1 (64 * 64)
#define NM_W2 (9 * 64)
#define NM_WTOT (NM_W0 + NM_W1 + NM_W2)
// weight-gradient partials in accumulator order: W1's sixteen 16x16 blocks | W0's four (13 of 16 columns used) | W2's four
// (9 of 16 rows used); element ((block * 64 + lane) * 4 + reg) = row 4 * (lane >> 4) + reg, column lane & 15 of the block
#define NM_WACC_W0 (64 * 64)
#define NM_WACC_W2 (64 * 64 + 16 * 64)
#define NM_WACC (64 * 64 + 16 * 64 + 16 * 64)
// accumulator-order index -> index in w0 | w1 | w2 ((out, in) order, back to back), -1 for the padding
__host__ __device__ inline int wacc_to_plain(int e) {
  const int reg = e & 3, lane = (e >> 2) & 63, blk = e >> 8, g = lane >> 4, j = lane & 15;
  if (blk < 16) return NM_W0 + (16 * (blk >> 2) + 4 * g + reg) * 64 + 16 * (blk & 3) + j;            // W1: block (rt, ctp)
  if (blk < 20) return j < 13 ? (16 * (blk - 16) + 4 * g + reg) * 13 + j : -1;                       // W0: block rt
  return 4 * g + reg < 9 ? NM_W0 + NM_W1 + (4 * g + reg) * 64 + 16 * (blk - 20) + j : -1;           // W2: block ctp
}

// Standard normal cdf Phi(x) and pdf phi(x) sharing ONE exponential: Abramowitz-Stegun 7.1.26 writes
// erf(z) = 1 - poly(t) e^{-z^2}, t = 1/(1 + p z) (|error| <= 1.5e-7, the fp32 rounding level), and with z = |x|/sqrt2
// the factor e^{-z^2} = e^{-x^2/2} is sqrt(2 pi) phi(x).  ~16 VALU instructions for GELU value AND derivative, against
// ~3 erff + 1 expf library calls (the constitutive kernels are VALU-bound on exactly this).
__device__ __forceinline__ void nm_phi(float x, float& Phi, float& phi) {
  const float ax = fabsf(x);
  // v_exp_f32 / v_rcp_f32 directly (1 ulp): the IEEE-rounded reciprocal costs ~11 VALU instructions per GELU
  const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float half_tail = 0.5f * poly * t * e;      // 0.5 * erfc(|x|/sqrt2)
  Phi = x >= 0.f ? 1.0f - half_tail : half_tail;
  phi = 0.3989422804014327f * e;
}
__device__ __forceinline__ float nm_gelu(float x) {   // the erf-form GELU of material/utils.py:16-17, erf to 1.5e-7 absolute (A&S 7.1.26)
  float P, p;
  nm_phi(x, P, p);
  return x * P;
}
__device__ __forceinline__ void nm_gelu_both(float x, float& h, float& dh) {
  float P, p;
  nm_phi(x, P, p);
  h = x * P;
  dh = fmaf(x, p, P);
}

// Two GELU value / derivative pairs with packed fp32 arithmetic.  Measured on gfx950 with one wave per SIMD
// (tools/ubench_valu.hip): a plain VALU instruction issues every 5.3 cycles, v_rcp / v_exp every 9, and a v_pk_fma_f32 /
// v_pk_mul_f32 every 6.2 - two results for little more than the price of one.  The constitutive kernels are issue-bound
// (their duration is the sum of their instructions' issue costs), so a pair of pairs drops from ~2 x 92 to ~132 cycles.
// Same polynomial as nm_phi with 0.5 folded into the coefficients (exact), and
// Phi = 0.5 + copysign(0.5 - half_tail, x) instead of the select (|difference| <= 1 ulp of 0.5).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nm_gelu_both2(const f2 x, f2& h, f2& dh) {
  const float k = -0.5f * 1.4426950408889634f, c = 0.3275911f * 0.70710678118654752f;
  f2 u = x * x;
  u = u * (f2){k, k};
  f2 t, e;
  t[0] = fmaf(c, fabsf(x[0]), 1.0f);
  t[1] = fmaf(c, fabsf(x[1]), 1.0f);
  e[0] = __builtin_amdgcn_exp2f(u[0]);
  e[1] = __builtin_amdgcn_exp2f(u[1]);
  t[0] = __builtin_amdgcn_rcpf(t[0]);
  t[1] = __builtin_amdgcn_rcpf(t[1]);
  const float c1 = 0.5f * 1.061405429f, c2 = 0.5f * -1.453152027f, c3 = 0.5f * 1.421413741f, c4 = 0.5f * -0.284496736f,
              c5 = 0.5f * 0.254829592f;
  f2 poly = __builtin_elementwise_fma(t, (f2){c1, c1}, (f2){c2, c2});
  poly = __builtin_elementwise_fma(poly, t, (f2){c3, c3});
  poly = __builtin_elementwise_fma(poly, t, (f2){c4, c4});
  poly = __builtin_elementwise_fma(poly, t, (f2){c5, c5});
  const f2 half_tail = poly * t * e;                       // 0.5 erfc(|x| / sqrt2)
  const f2 q = (f2){0.5f, 0.5f} - half_tail;
  f2 r;
  r[0] = __builtin_copysignf(q[0], x[0]);
  r[1] = __builtin_copysignf(q[1], x[1]);
  const f2 P = r + (f2){0.5f, 0.5f};
  const f2 phi = e * (f2){0.3989422804014327f, 0.3989422804014327f};
  h = x * P;
  dh = __builtin_elementwise_fma(x, phi, P);
}

// ---------------------------------------------------------------- standalone SVD operator
__global__ void __launch_bounds__(256) k_svd_fwd(int n, const float* __restrict__ F, float* __restrict__ U,
                                                 float* __restrict__ sig, float* __restrict__ Vh) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  M3 A = m3_load(F + 9 * p), Um, Vm;
  float s[3];
  nm_svd3(A, Um, s, Vm);
  m3_store(U + 9 * p, Um);
  sig[3 * p] = s[0]; sig[3 * p + 1] = s[1]; sig[3 * p + 2] = s[2];
  m3_store(Vh + 9 * p, m3_transpose(Vm));
}

// adjoint with the clamped denominators of warp's adj_svd3
__global__ void __launch_bounds__(256) k_svd_bwd(int n, const float* __restrict__ U, const float* __restrict__ sig,
                                                 const float* __restrict__ Vh, const float* __restrict__ gU,
                                                 const float* __restrict__ gs, const float* __restrict__ gVh,
                                                 float* __restrict__ gF) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  M3 Um = m3_load(U + 9 * p), Vhm = m3_load(Vh + 9 * p);
  M3 gUm = gU ? m3_load(gU + 9 * p) : m3_zero();
  M3 gVhm = gVh ? m3_load(gVh + 9 * p) : m3_zero();
  float s[3] = {sig[3 * p], sig[3 * p + 1], sig[3 * p + 2]};
  float g[3] = {0.f, 0.f, 0.f};
  if (gs) { g[0] = gs[3 * p]; g[1] = gs[3 * p + 1]; g[2] = gs[3 * p + 2]; }
  M3 UtgU = m3_mul_tn(Um, gUm);
  M3 VtgV = m3_mul_nt(Vhm, gVhm);  // V^T gV = Vh (gVh)^T
  M3 inner = m3_zero();
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (i == j) { inner.m[4 * i] = g[i]; continue; }
      int a = i < j ? i : j, b = i < j ? j : i;
      float e = 1.f / fminf(s[b] * s[b] - s[a] * s[a], -1e-6f);
      if (i > j) e = -e;
      float su = e * (UtgU.m[3 * i + j] - UtgU.m[3 * j + i]);
      float sv = e * (VtgV.m[3 * i + j] - VtgV.m[3 * j + i]);
      inner.m[3 * i + j] = su * s[j] + s[i] * sv;
    }
  m3_store(gF + 9 * p, m3_mul(m3_mul(Um, inner), Vhm));
}

extern "C" int nm_svd3_fwd(int32_t n, const float* F, float* U, float* sigma, float* Vh, void* stream) {
  NM_REQUIRE(n >= 0, "negative n");
  if (n == 0) return NM_OK;
  NM_REQUIRE(F && U && sigma && Vh, "null pointer");
  NM_LAUNCH(k_svd_fwd, dim3(nm_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, F, U, sigma, Vh);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" int nm_svd3_bwd(int32_t n, const float* U, const float* sigma, const float* Vh, const float* gU,
                           const float* gsigma, const float* gVh, float* gF, void* stream) {
  NM_REQUIRE(n >= 0, "negative n");
  if (n == 0) return NM_OK;
  NM_REQUIRE(U && sigma && Vh && gF, "null pointer");
  NM_LAUNCH(k_svd_bwd, dim3(nm_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, U, sigma, Vh, gU, gsigma,
                     gVh, gF);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

// ---------------------------------------------------------------- weight staging (LDS, MFMA operand order)
// forward operands
//   P0[(ks*4+rt)*64 + l]            = W0[16rt + (l&15)][4ks + (l>>4)]             (k >= 13 -> 0)
//   P1[((rtp*4+reg)*4+rt)*64 + l]   = W1[16rt + (l&15)][16rtp + 4(l>>4) + reg]
//   P2[(rtp*4+reg)*64 + l]          = W2[l&15][16rtp + 4(l>>4) + reg]             (row >= 9 -> 0)
// backward (transposed) operands
//   Q2[(ks*4+rt)*64 + l]            = W2[4ks + (l>>4)][16rt + (l&15)]             (row >= 9 -> 0), ks < 3
//   Q1[((rtp*4+reg)*4+rt)*64 + l]   = W1[16rtp + 4(l>>4) + reg][16rt + (l&15)]
//   Q0[(rtp*4+reg)*64 + l]          = W0[16rtp + 4(l>>4) + reg][l&15]             (col >= 13 -> 0)
// Raw weights are first copied to LDS with coalesced loads (w0 | w1 | w2 back to back, NM_WTOT floats), then permuted
// LDS -> LDS into MFMA operand order (a direct permuting gather from global costs ~12k cycles per workgroup: 64
// scattered 4-byte reads per wave-instruction).
#define NM_RLD 65
#define NM_RAW1 NM_W0
#define NM_RAW2 (NM_W0 + 64 * NM_RLD)
#define NM_RAWTOT (NM_RAW2 + 9 * NM_RLD)
__device__ __forceinline__ void stage_raw_weights(const float* __restrict__ w0, const float* __restrict__ w1,
                                                  const float* __restrict__ w2, float* raw) {
  // all 11 loads of a thread are issued before the first LDS write: a rolled copy loop pays one L2 round trip per
  // iteration (~1k cycles each, 11 of them), this pays one.  Requires blockDim.x == 256.
  const int tid = threadIdx.x;
  float a0[4], a2[3];
  float4 a1[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int i = tid + 256 * k;
    a0[k] = i < NM_W0 ? w0[i] : 0.f;
    a1[k] = reinterpret_cast<const float4*>(w1)[i];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int i = tid + 256 * k;
    a2[k] = i < NM_W2 ? w2[i] : 0.f;
  }
  // w1 / w2 rows are padded to NM_RLD floats so that the column-major gathers of the permute pass are conflict free
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int i = tid + 256 * k;
    if (i < NM_W0) raw[i] = a0[k];
    float* d = raw + NM_RAW1 + (i >> 4) * NM_RLD + 4 * (i & 15);
    d[0] = a1[k].x; d[1] = a1[k].y; d[2] = a1[k].z; d[3] = a1[k].w;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int i = tid + 256 * k;
    if (i < NM_W2) raw[NM_RAW2 + (i >> 6) * NM_RLD + (i & 63)] = a2[k];
  }
}
// Operand order in LDS: the A operands of four consecutive MFMAs sit side by side per lane - row (op / 4), lane, op % 4 -
// so that a lane fetches them with ONE 16-byte LDS read (an LDS instruction between MFMAs is not free: ~6 cycles of the
// wave's issue, tools/ubench_valu.hip).
#define NM_WIDX(idx) (((((idx) >> 6) >> 2) * 64 + ((idx) & 63)) * 4 + (((idx) >> 6) & 3))
__device__ __forceinline__ void stage_fwd_weights(const float* raw, float* P0, float* P1, float* P2) {
  const float *w0 = raw, *w1 = raw + NM_RAW1, *w2 = raw + NM_RAW2;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < 16 * 64; idx += blockDim.x) {
    int l = idx & 63, op = idx >> 6, ks = op >> 2, rt = op & 3, k = 4 * ks + (l >> 4);
    P0[NM_WIDX(idx)] = k < 13 ? w0[(16 * rt + (l & 15)) * 13 + k] : 0.f;
    int reg = op & 3, rtp = op >> 2, row = l & 15;
    P2[NM_WIDX(idx)] = row < 9 ? w2[row * NM_RLD + 16 * rtp + 4 * (l >> 4) + reg] : 0.f;
  }
  for (int idx = tid; idx < 64 * 64; idx += blockDim.x) {
    int l = idx & 63, op = idx >> 6, rt = op & 3, reg = (op >> 2) & 3, rtp = op >> 4;
    P1[NM_WIDX(idx)] = w1[(16 * rt + (l & 15)) * NM_RLD + 16 * rtp + 4 * (l >> 4) + reg];
  }
}
__device__ __forceinline__ void stage_bwd_weights(const float* raw, float* Q0, float* Q1, float* Q2) {
  const float *w0 = raw, *w1 = raw + NM_RAW1, *w2 = raw + NM_RAW2;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < 12 * 64; idx += blockDim.x) {
    int l = idx & 63, op = idx >> 6, ks = op >> 2, rt = op & 3, row = 4 * ks + (l >> 4);
    Q2[NM_WIDX(idx)] = row < 9 ? w2[row * NM_RLD + 16 * rt + (l & 15)] : 0.f;
  }
  for (int idx = tid; idx < 16 * 64; idx += blockDim.x) {
    int l = idx & 63, op = idx >> 6, reg = op & 3, rtp = op >> 2, col = l & 15;
    Q0[NM_WIDX(idx)] = col < 13 ? w0[(16 * rtp + 4 * (l >> 4) + reg) * 13 + col] : 0.f;
  }
  for (int idx = tid; idx < 64 * 64; idx += blockDim.x) {
    int l = idx & 63, op = idx >> 6, rt = op & 3, reg = (op >> 2) & 3, rtp = op >> 4;
    Q1[NM_WIDX(idx)] = w1[(16 * rtp + 4 * (l >> 4) + reg) * NM_RLD + 16 * rt + (l & 15)];
  }
}

// Weights already in MFMA operand order (nm_material_prepare): P0 | P1 | P2 | Q0 | Q1 | Q2, NM_PERM_FWD floats for the
// forward operands, NM_PERM_ALL with the transposed ones.  Staging is then a straight 16-byte copy.
#define NM_PERM_FWD (16 * 64 + 64 * 64 + 16 * 64)
#define NM_PERM_ALL (NM_PERM_FWD + 16 * 64 + 64 * 64 + 12 * 64)
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// `between` runs after the loads are issued and before the first LDS write waits for them: loads placed there are younger than
// the weights' and are not waited for here (vmcnt retires in order)
template <int NFLOAT, class HOOK = NoHook>
__device__ __forceinline__ void stage_permuted(const float* __restrict__ wperm, float* dst, HOOK between = HOOK()) {
  constexpr int NV = NFLOAT / 4, PER = (NV + 255) / 256;
  float4 v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int i = threadIdx.x + 256 * k;
    v[k] = i < NV ? reinterpret_cast<const float4*>(wperm)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  NM_SB_();
  between();
  NM_SB_();
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int i = threadIdx.x + 256 * k;
    if (i < NV) reinterpret_cast<float4*>(dst)[i] = v[k];
  }
}
// Two pieces staged with ONE round trip, and without waiting for anything but themselves.  The loads are inline assembly:
// written as C++ the compiler (a) put the loads of the last, partly filled slice under a lane predicate - a branch region,
// behind which its wait-count bookkeeping drains EVERY outstanding load, the particles' HBM requests in front included - and,
// with clamped indices instead of predicates, (b) sank each load next to its LDS write: seven round trips in a row.  Here
// the order is what is written: all weight loads, then `between` - which must issue exactly NHOOK vector-memory loads,
// unconditionally; they stay in flight - then one s_waitcnt that leaves those NHOOK outstanding (vmcnt retires in order),
// then the LDS writes.  Loads the compiler knows about and that were issued earlier are older and therefore complete too.
template <int NA, int NB, int NHOOK, class HOOK>
__device__ __forceinline__ void stage_permuted2(const float* __restrict__ srca, float* dsta, const float* __restrict__ srcb, float* dstb,
                                                HOOK between) {
  constexpr int VA = NA / 4, PA = (VA + 255) / 256, VB = NB / 4, PB = (VB + 255) / 256;
  f4 va[PA], vb[PB];
#pragma unroll
  for (int k = 0; k < PA; ++k) {
    const int i = threadIdx.x + 256 * k;
    const f4* ptr = reinterpret_cast<const f4*>(srca) + (i < VA ? i : VA - 1);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(va[k]) : "v"(ptr));
  }
#pragma unroll
  for (int k = 0; k < PB; ++k) {
    const int i = threadIdx.x + 256 * k;
    const f4* ptr = reinterpret_cast<const f4*>(srcb) + (i < VB ? i : VB - 1);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vb[k]) : "v"(ptr));
  }
  asm volatile("" ::: "memory");
  between();
  static_assert(NHOOK >= 0 && NHOOK < 48, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NHOOK) : "memory");
#pragma unroll
  for (int k = 0; k < PA; ++k) {
    const int i = threadIdx.x + 256 * k;
    asm volatile("" : "+v"(va[k]));      // (the value is only now defined as far as the compiler may assume)
    if (i < VA) *reinterpret_cast<f4*>(dsta + 4 * i) = va[k];
  }
#pragma unroll
  for (int k = 0; k < PB; ++k) {
    const int i = threadIdx.x + 256 * k;
    asm volatile("" : "+v"(vb[k]));
    if (i < VB) *reinterpret_cast<f4*>(dstb + 4 * i) = vb[k];
  }
}
// one workgroup: raw (out,in) weights -> operand order in global memory (once per roll-out and net)
__global__ void __launch_bounds__(256) k_permute_weights(const float* __restrict__ w0, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, float* __restrict__ wperm,
                                                         const float* __restrict__ w0b, const float* __restrict__ w1b,
                                                         const float* __restrict__ w2b, float* __restrict__ wpermb) {
  __shared__ float raw[NM_RAWTOT];
  if (blockIdx.x == 1) { w0 = w0b; w1 = w1b; w2 = w2b; wperm = wpermb; }     // (second net of a roll-out: same launch)
  __shared__ float perm[NM_PERM_ALL];
  stage_raw_weights(w0, w1, w2, raw);
  __syncthreads();
  stage_fwd_weights(raw, perm, perm + 16 * 64, perm + 16 * 64 + 64 * 64);
  stage_bwd_weights(raw, perm + NM_PERM_FWD, perm + NM_PERM_FWD + 16 * 64, perm + NM_PERM_FWD + 16 * 64 + 64 * 64);
  __syncthreads();
  for (int i = threadIdx.x; i < NM_PERM_ALL; i += 256) wperm[i] = perm[i];
}

// invariants of meta.py:197-213 for one particle: z[13], R = U V^T (also returns U, V, sigma)
// SVD cache of the fused roll-out (nm_rollout_cfg.svd_cache): the forward kernels leave U | sigma | V of every particle (21
// floats, component-major so that a wave's accesses are contiguous), the reverse sweep reads them back instead of running
// the Jacobi iteration again on the same matrix (5 k of the 7 k cycles a 64-particle round spends before its first tile).
__device__ __forceinline__ void svd_store(float* __restrict__ dst, int n, int p, const M3& U, const float s[3], const M3& V) {
#pragma unroll
  for (int c = 0; c < 9; ++c) __builtin_nontemporal_store(U.m[c], &dst[(size_t)c * n + p]);
#pragma unroll
  for (int c = 0; c < 3; ++c) __builtin_nontemporal_store(s[c], &dst[(size_t)(9 + c) * n + p]);
#pragma unroll
  for (int c = 0; c < 9; ++c) __builtin_nontemporal_store(V.m[c], &dst[(size_t)(12 + c) * n + p]);
}
__device__ __forceinline__ void svd_load(const float* __restrict__ src, int n, int p, M3& U, float s[3], M3& V) {
#pragma unroll
  for (int c = 0; c < 9; ++c) U.m[c] = __builtin_nontemporal_load(&src[(size_t)c * n + p]);
#pragma unroll
  for (int c = 0; c < 3; ++c) s[c] = __builtin_nontemporal_load(&src[(size_t)(9 + c) * n + p]);
#pragma unroll
  for (int c = 0; c < 9; ++c) V.m[c] = __builtin_nontemporal_load(&src[(size_t)(12 + c) * n + p]);
}
// HAVE_SVD: U, s, V are given (svd_load)
template <bool HAVE_SVD = false>
__device__ __forceinline__ void nm_features(const M3& F, float z[13], M3& R, M3& U, M3& V, float s[3]) {
  if (!HAVE_SVD) nm_svd3(F, U, s, V);
  R = m3_mul_nt(U, V);
  M3 G = m3_mul_tn(F, F);
  z[0] = s[0] - 1.f; z[1] = s[1] - 1.f; z[2] = s[2] - 1.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) z[3 + i] = G.m[i] - ((i % 4 == 0) ? 1.f : 0.f);
  z[12] = m3_det(F) - 1.f;
}

// MFMA chains whose A operands (weights in operand order, one 64-float row per MFMA) stream from LDS.  A wave is alone
// on its SIMD, so nobody hides an LDS round trip for it: the compiler's own schedule (4 reads, wait, 8 MFMAs, 4 reads,
// wait ...) exposed one LDS latency per 8 MFMAs (29 % of the wave's cycles were spent waiting, SQ_WAIT_INST_ANY in
// profiles/r02a).  Here the reads of group g+1 are issued BEFORE the MFMAs of group g (sched_barrier keeps the compiler
// from sinking them back), and the first group of a chain is fetched by the caller ahead of the phase in front of it.
// An LDS instruction between MFMAs costs its issue slot only (~6 cycles, nothing when it replaces a wait); VALU work
// does not hide at all, so everything that can be an LDS access in the shadow of a chain is placed there
// (`mid`), and the operands of four MFMAs come with one 16-byte read.
#define NM_SB() __builtin_amdgcn_sched_barrier(0)
template <int G>
struct AGroup {
  float a[G];
};
template <int G>
__device__ __forceinline__ void a_fetch(AGroup<G>& o, const float* __restrict__ base, int lane, int grp) {
  static_assert(G % 4 == 0, "groups of four operands (one 16-byte read each)");
#pragma unroll
  for (int q = 0; q < G / 4; ++q) {
    const f4 v = *reinterpret_cast<const f4*>(base + ((grp * (G / 4) + q) * 64 + lane) * 4);
    o.a[4 * q] = v[0]; o.a[4 * q + 1] = v[1]; o.a[4 * q + 2] = v[2]; o.a[4 * q + 3] = v[3];
  }
}
struct NoMid {
  __device__ __forceinline__ void operator()() const {}
};
// NOPS MFMAs: op -> acc[op & AM] += A(row op) x b[op >> BS];  mid() runs in the shadow of the first group
template <int NOPS, int G, int AM, int BS, class MID = NoMid>
__device__ __forceinline__ void a_chain(AGroup<G>& cur, const float* __restrict__ base, int lane, const float* b, f4* acc,
                                        MID mid = MID()) {
  static_assert(NOPS % G == 0, "whole groups");
#pragma unroll
  for (int g = 0; g < NOPS / G; ++g) {
    AGroup<G> nxt;
    if (g + 1 < NOPS / G) a_fetch<G>(nxt, base, lane, g + 1);
    NM_SB();
    if (g == 0) mid();
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int op = g * G + i;
      acc[op & AM] = NM_MFMA(cur.a[i], b[op >> BS], acc[op & AM]);
    }
    NM_SB();
    if (g + 1 < NOPS / G) cur = nxt;
  }
}

// three-layer MLP on one 16-particle column tile; B operand of layer 0 comes from zrow (LDS, [particle][17]).
// `first` holds the first A group of layer 0 (a_fetch<8>(first, P0, lane, 0), issued by the caller ahead of time).
// WITH_GRAD (reverse sweep): keeps the GELU derivatives, and the hidden activations go to LDS transposed
// ([feature][16 particles + pad], th1 / th2) in the shadow of the next layer's MFMAs - the weight-gradient products read
// them from there much later, so neither the writes nor the reads are ever waited for.
struct MlpFwd {
  f4 g1[4], g2[4];
  f4 y;
};
template <bool WITH_GRAD>
__device__ __forceinline__ void mlp_forward_tile(const float* __restrict__ P0, const float* __restrict__ P1,
                                                 const float* __restrict__ P2, const float* zin /* 4 B operands of layer 0 */,
                                                 int lane, AGroup<8>& first, MlpFwd& o, float* th1 = nullptr, float* th2 = nullptr,
                                                 f4* __restrict__ act = nullptr) {
  // act != NULL (activation cache of the fused roll-out, WITH_GRAD only): the second hidden layer's activations and GELU
  // derivatives and the output are written out in the accumulator layout itself - NM_ACT_SLOTS x 64 lanes x 16 B, slots
  // h2[0..3], g2[4..7], y[8] - in the shadow of the last MFMA chain; the reverse sweep loads them back and recomputes only
  // the first layer (16 of the forward pass's 96 MFMAs, half of its GELUs).  The whole record (17 slots) was measured too:
  // the reverse sweep's tile loops then ask for 5.3 TB/s and the forward kernels write twice as much - slower overall.
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  const int j = lane & 15, g = lane >> 4;
  f4 a1[4] = {zero, zero, zero, zero};
  a_chain<16, 8, 3, 2>(first, P0, lane, zin, a1);
  AGroup<8> w1g;
  a_fetch<8>(w1g, P1, lane, 0);      // in flight under the GELUs
  NM_SB();
  float hb[16];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      f2 h, dh;
      nm_gelu_both2((f2){a1[rt][2 * pr], a1[rt][2 * pr + 1]}, h, dh);
      hb[4 * rt + 2 * pr] = h[0];
      hb[4 * rt + 2 * pr + 1] = h[1];
      if (WITH_GRAD) {
        o.g1[rt][2 * pr] = dh[0];
        o.g1[rt][2 * pr + 1] = dh[1];
      }
    }
  f4 a2[4] = {zero, zero, zero, zero};
  a_chain<64, 8, 3, 2>(w1g, P1, lane, hb, a2, [&]() {
    if (WITH_GRAD && th1) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) th1[(16 * rt + 4 * g + r) * 17 + j] = hb[4 * rt + r];
      __builtin_amdgcn_wave_barrier();   // (other lanes read these words: keep the compiler from reordering around them)
    }
  });
  AGroup<8> w2g;
  a_fetch<8>(w2g, P2, lane, 0);
  NM_SB();
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      f2 h, dh;
      nm_gelu_both2((f2){a2[rt][2 * pr], a2[rt][2 * pr + 1]}, h, dh);
      hb[4 * rt + 2 * pr] = h[0];
      hb[4 * rt + 2 * pr + 1] = h[1];
      if (WITH_GRAD) {
        o.g2[rt][2 * pr] = dh[0];
        o.g2[rt][2 * pr + 1] = dh[1];
      }
    }
  f4 yy[2] = {zero, zero};
  a_chain<16, 8, 1, 0>(w2g, P2, lane, hb, yy, [&]() {
    if (WITH_GRAD && th2) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) th2[(16 * rt + 4 * g + r) * 17 + j] = hb[4 * rt + r];
      __builtin_amdgcn_wave_barrier();
    }
    if (WITH_GRAD && act) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        __builtin_nontemporal_store((f4){hb[4 * rt], hb[4 * rt + 1], hb[4 * rt + 2], hb[4 * rt + 3]}, &act[rt * 64 + lane]);
        __builtin_nontemporal_store(o.g2[rt], &act[(4 + rt) * 64 + lane]);
      }
    }
  });
  o.y = yy[0] + yy[1];
  if (WITH_GRAD && act) __builtin_nontemporal_store(o.y, &act[8 * 64 + lane]);
}
#define NM_ACT_SLOTS 9       // f4 slots per lane and 16-particle tile in the activation cache: h2[4], g2[4], y

// ---------------------------------------------------------------- forward
// Work split of the constitutive kernels: one workgroup (4 waves, one per SIMD) per CU, each wave owns q consecutive
// particles, q = the per-wave share rounded up to whole 16-particle MFMA tiles.  At 100k particles every wave gets
// 112 particles = 2 SVD rounds + 7 tiles, instead of a 64-particle batch granularity that leaves half the SIMDs with
// 2 batches (8 tiles) and the rest with 1.
static inline void nm_wave_quota(int n, int& grid, int& q) {
  const int waves = NM_BWD_GRID * 4;
  q = (n + waves - 1) / waves;
  q = ((q + 15) / 16) * 16;
  if (q < 16) q = 16;
  grid = nm_div_up(n, 4 * (int64_t)q);
  if (grid < 1) grid = 1;
}

// One round (up to 64 particles, one per lane) of a constitutive net on the wave's LDS slices: invariants of Fp -> MLP on the
// round's 16-particle tiles -> the net's output matrix (meta.py:219-221 / 486-488).  p = the lane's particle, c0 = the round's
// first particle (a multiple of 16: tile id = particle / 16), ntile wave-uniform.
template <int KIND, bool ACT>
__device__ __forceinline__ M3 material_fwd_round(const M3& Fp, bool valid, int n, int p, int c0, int ntile, int lane, float alpha,
                                                 const float* sP0, const float* sP1, const float* sP2, float* zb, float* yb,
                                                 float* __restrict__ svd_out, f4* __restrict__ act_out) {
  const int j = lane & 15, g = lane >> 4;
  M3 R, U, V;
  float z[13], s[3];
  nm_features(Fp, z, R, U, V, s);
  if (svd_out && valid) svd_store(svd_out, n, p, U, s, V);
#pragma unroll
  for (int c = 0; c < 13; ++c) zb[lane * 17 + c] = z[c];
  zb[lane * 17 + 13] = 0.f; zb[lane * 17 + 14] = 0.f; zb[lane * 17 + 15] = 0.f;
  __builtin_amdgcn_wave_barrier();
  // layer-0 B operands of all four column tiles first, results kept in registers: the tile loop itself then has
  // no LDS traffic besides the (hoisted) weights, so the scheduler may overlap one tile's GELU with another's MFMAs
  float zin[4][4];
  f4 yv[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) zin[ct][ks] = zb[(ct * 16 + j) * 17 + 4 * ks + g];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    if (ct < ntile) {     // wave-uniform
      MlpFwd m;
      AGroup<8> first;
      a_fetch<8>(first, sP0, lane, 0);
      if (ACT)
        mlp_forward_tile<true>(sP0, sP1, sP2, zin[ct], lane, first, m, nullptr, nullptr,
                               act_out + (size_t)((c0 >> 4) + ct) * NM_ACT_SLOTS * 64);
      else
        mlp_forward_tile<false>(sP0, sP1, sP2, zin[ct], lane, first, m);
      yv[ct] = m.y;
    } else {
      yv[ct] = (f4){0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = 4 * g + r;
      if (row < 9) yb[(ct * 16 + j) * 9 + row] = yv[ct][r];
    }
  __builtin_amdgcn_wave_barrier();
  M3 X;
#pragma unroll
  for (int i = 0; i < 9; ++i) X.m[i] = yb[lane * 9 + i];
  M3 Xs;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Xs.m[3 * r + c] = 0.5f * (X.m[3 * r + c] + X.m[3 * c + r]);
  M3 RX = m3_mul(R, Xs), o;
  if (KIND == NM_ELASTICITY) {
    o = m3_mul_nt(RX, Fp);  // R X F^T  (meta.py:219-221)
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) o.m[i] = alpha * RX.m[i] + Fp.m[i];  // meta.py:486-488
  }
  __builtin_amdgcn_wave_barrier();
  return o;
}

template <int KIND, bool ACT>
__global__ void __launch_bounds__(256) k_material_fwd(int n, int q, float alpha, const float* __restrict__ F,
                                                      const float* __restrict__ w0, const float* __restrict__ w1,
                                                      const float* __restrict__ w2, const float* __restrict__ wperm,
                                                      float* __restrict__ out, GridPrologue pro, G2pFuse gf,
                                                      float* __restrict__ svd_out, f4* __restrict__ act_out) {
  __shared__ __attribute__((aligned(16))) float sP[NM_PERM_FWD];
  float *sP0 = sP, *sP1 = sP + 16 * 64, *sP2 = sP + 16 * 64 + 64 * 64;
  // per-wave buffers (features 64x17, outputs 64x9); before the main loop the same memory holds the raw weights
  __shared__ __attribute__((aligned(16))) float sBuf[4 * 64 * 17 + 4 * 64 * 9];
  static_assert(4 * 64 * 17 + 4 * 64 * 9 >= NM_RAWTOT, "raw weights must fit the per-wave buffers");
  float (*sZ)[64 * 17] = reinterpret_cast<float (*)[64 * 17]>(sBuf);
  float (*sY)[64 * 9] = reinterpret_cast<float (*)[64 * 9]>(sBuf + 4 * 64 * 17);
  NM_PH_DECL
  // roll-out: the clear of the MPM substep that follows (nm_grid.h) - on workgroups of its own when CUs are to spare
  if (NM_PROLOGUE_SPLIT(pro)) return;
  if (wperm) {
    stage_permuted<NM_PERM_FWD>(wperm, sP);
  } else {
    stage_raw_weights(w0, w1, w2, sBuf);
    __syncthreads();
    stage_fwd_weights(sBuf, sP0, sP1, sP2);
  }
  __syncthreads();
  NM_PH(0)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* zb = sZ[wave];
  float* yb = sY[wave];
  // wave w owns particles [w*q, (w+1)*q), q a multiple of 16 (nm_wave_quota): every wave runs the same number of SVD
  // rounds and - because a round only visits the 16-particle tiles it actually holds - (almost) the same number of tiles
  const int pbeg = min(n, (blockIdx.x * 4 + wave) * q), pend = min(n, pbeg + q);
  for (int c0 = pbeg; c0 < pend; c0 += 64) {
    const int p = c0 + lane;
    const bool valid = p < pend;
    const int ntile = (min(64, pend - c0) + 15) >> 4;
    M3 Fp = m3_ident();
    if (gf.gv) {     // roll-out: this kernel performs the substep's g2p and takes the trial F straight from it
      // (a disabled particle's row of the checkpoint gets a fresh state's values, nm_grid.h: Fp = I goes through the net)
      if (valid) g2p_particle<true>(gf.K, p, gf.clip, gf.enabled, gf.x, gf.v, gf.C, gf.F, gf.gv, gf.xn, gf.vn, gf.Cn, Fp, nullptr, 0, true);
    } else if (valid) {
      Fp = m3_load(F + 9 * p);
    }
    NM_PH(1)
    const M3 o = material_fwd_round<KIND, ACT>(Fp, valid, n, p, c0, ntile, lane, alpha, sP0, sP1, sP2, zb, yb, svd_out, act_out);
    if (valid) m3_store(out + 9 * p, o);
    NM_PH(2)
  }
  NM_PH_STORE
}

// Roll-out forward, substeps t and t+1 in one launch (finetune.py:362-364 seen from the particle): g2p of substep t ->
// trial F -> plasticity net -> F_{t+1} (checkpointed) -> elasticity net -> stress_{t+1}.  F_{t+1} stays in registers between
// the nets, both nets' operands are staged once, and the launch carries the grid housekeeping of substep t+1 (GridPrologue
// mode 1 with keep_gv: the velocities are still being gathered here; the grid update of substep t+1 zeroes what drops out).
// Round 6: the kernel is held to 256 registers per lane (`__launch_bounds__(256, 2)`; it needed 231 + 16): with the whole
// budget addressable as VGPRs the compiler keeps the MFMA accumulators in VGPRs and the v_accvgpr_read / _write copies around
// every GELU go away (52.4 -> 50.1 us per launch, same box).  TWO workgroups per CU - which the bound also allows (74 KB of LDS
// each) - were built and measured too: 1920 waves of 3-4 tiles, two per SIMD, 51.5 us against 50.1: the SIMD's datapath, not a
// lone wave's issue rate, is what the lane-per-particle phases fill (f32 MFMA and VALU share it), so a second wave has nothing
// to run in; the launch stays one workgroup per CU.
template <bool ACT>
__global__ void __launch_bounds__(256, 2) k_material_fwd_pair(int n, int q, float alpha, const float* __restrict__ wperm_p,
                                                           const float* __restrict__ wperm_e, float* __restrict__ F_next,
                                                           float* __restrict__ stress_next, GridPrologue pro, G2pFuse gf,
                                                           float* __restrict__ svd_p, float* __restrict__ svd_e,
                                                           f4* __restrict__ act_p, f4* __restrict__ act_e) {
  __shared__ __attribute__((aligned(16))) float sPp[NM_PERM_FWD];
  __shared__ __attribute__((aligned(16))) float sPe[NM_PERM_FWD];
  __shared__ __attribute__((aligned(16))) float sBuf[4 * 64 * 17 + 4 * 64 * 9];
  float (*sZ)[64 * 17] = reinterpret_cast<float (*)[64 * 17]>(sBuf);
  float (*sY)[64 * 9] = reinterpret_cast<float (*)[64 * 9]>(sBuf + 4 * 64 * 17);
  if (material_prologue_fwd(pro)) return;
  NM_PH_DECL
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pbeg = min(n, (blockIdx.x * 4 + wave) * q), pend = min(n, pbeg + q);
  // the stencil-independent loads of a round's particles (enabled, x, clip, F) are issued one round ahead - the first round's in
  // front of the weight staging -: the wave is alone on its SIMD, so a round that starts with its own loads spends their whole
  // HBM round trip (~2 us) doing nothing
  // (unconditional, at a clamped index - a lane without a particle loads the last one's and ignores it: see G2pIn)
  G2pIn nxt = g2p_in_load(min(pbeg + lane, n - 1), gf.clip, gf.enabled, gf.x, gf.F);
  NM_SB();
  stage_permuted2<NM_PERM_FWD, NM_PERM_FWD, 0>(wperm_p, sPp, wperm_e, sPe, NoHook());     // both nets: one round trip
  __syncthreads();
  NM_PH(0)
  float* zb = sZ[wave];
  float* yb = sY[wave];
  for (int c0 = pbeg; c0 < pend; c0 += 64) {
    const int p = c0 + lane;
    const bool valid = p < pend;
    const int ntile = (min(64, pend - c0) + 15) >> 4;
    M3 Ftr = m3_ident();
    const G2pIn cur = nxt;
    nxt = g2p_in_load(min(p + 64, n - 1), gf.clip, gf.enabled, gf.x, gf.F);      // (also in the last round: nothing conditional)
    NM_SB();
    if (valid) g2p_particle<true>(gf.K, p, cur, gf.x, gf.gv, gf.xn, gf.vn, gf.Cn, Ftr, nullptr, 0, true);
    NM_PH(1)
    const M3 Fn = material_fwd_round<NM_PLASTICITY, ACT>(Ftr, valid, n, p, c0, ntile, lane, alpha, sPp, sPp + 16 * 64,
                                                         sPp + 16 * 64 + 64 * 64, zb, yb, svd_p, act_p);
    if (valid) m3_store(F_next + 9 * p, Fn);
    NM_PH(2)
    const M3 S = material_fwd_round<NM_ELASTICITY, ACT>(Fn, valid, n, p, c0, ntile, lane, 0.f, sPe, sPe + 16 * 64,
                                                        sPe + 16 * 64 + 64 * 64, zb, yb, svd_e, act_e);
    if (valid) m3_store(stress_next + 9 * p, S);
    NM_PH(3)
  }
  NM_PH_STORE_K(2)
}

// floats of one net's activation cache for n particles (one record per 16-particle tile)
size_t nm_material_act_floats(int32_t n) { return (size_t)nm_div_up(n > 0 ? n : 1, 16) * NM_ACT_SLOTS * 64 * 4; }

// internal (fused roll-out): wperm != NULL -> weights come pre-permuted from nm_material_prepare
int nm_material_fwd_launch(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w, const float* wperm, float* out,
                           const GridPrologue* pro, const G2pFuse* g2p, void* stream, float* svd_out, float* act_out) {
  int grid, q;
  nm_wave_quota(n, grid, q);
  hipStream_t s = (hipStream_t)stream;
  const float *w0 = w ? w->w0 : nullptr, *w1 = w ? w->w1 : nullptr, *w2 = w ? w->w2 : nullptr;
  GridPrologue gp;
  if (pro) gp = *pro; else { memset(&gp, 0, sizeof(gp)); }
  G2pFuse gf;
  if (g2p) gf = *g2p; else { memset(&gf, 0, sizeof(gf)); }
  gp.mat_grid = grid;
  const int launch = (pro && grid + NM_PRO_WGS <= NM_BWD_GRID) ? grid + NM_PRO_WGS : grid;
  f4* act4 = reinterpret_cast<f4*>(act_out);
  if (kind == NM_ELASTICITY && act4)
    NM_LAUNCH((k_material_fwd<NM_ELASTICITY, true>), dim3(launch), dim3(256), 0, s, n, q, alpha, F, w0, w1, w2, wperm, out, gp, gf, svd_out, act4);
  else if (kind == NM_ELASTICITY)
    NM_LAUNCH((k_material_fwd<NM_ELASTICITY, false>), dim3(launch), dim3(256), 0, s, n, q, alpha, F, w0, w1, w2, wperm, out, gp, gf, svd_out, act4);
  else if (act4)
    NM_LAUNCH((k_material_fwd<NM_PLASTICITY, true>), dim3(launch), dim3(256), 0, s, n, q, alpha, F, w0, w1, w2, wperm, out, gp, gf, svd_out, act4);
  else
    NM_LAUNCH((k_material_fwd<NM_PLASTICITY, false>), dim3(launch), dim3(256), 0, s, n, q, alpha, F, w0, w1, w2, wperm, out, gp, gf, svd_out, act4);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
int nm_material_fwd_pair_launch(int32_t n, float alpha_p, const float* wperm_p, const float* wperm_e, float* F_next,
                                float* stress_next, const GridPrologue* pro, const G2pFuse* g2p, void* stream, float* svd_p,
                                float* svd_e, float* act_p, float* act_e) {
  int grid, q;
  nm_wave_quota(n, grid, q);
  hipStream_t s = (hipStream_t)stream;
  GridPrologue gp;
  if (pro) gp = *pro; else { memset(&gp, 0, sizeof(gp)); }
  gp.mat_grid = grid;
  const int launch = (pro && grid + NM_PRO_WGS <= NM_BWD_GRID) ? grid + NM_PRO_WGS : grid;
  f4 *ap = reinterpret_cast<f4*>(act_p), *ae = reinterpret_cast<f4*>(act_e);
  if (ap && ae)
    NM_LAUNCH(k_material_fwd_pair<true>, dim3(launch), dim3(256), 0, s, n, q, alpha_p, wperm_p, wperm_e, F_next, stress_next, gp, *g2p,
              svd_p, svd_e, ap, ae);
  else
    NM_LAUNCH(k_material_fwd_pair<false>, dim3(launch), dim3(256), 0, s, n, q, alpha_p, wperm_p, wperm_e, F_next, stress_next, gp, *g2p,
              svd_p, svd_e, ap, ae);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
int nm_material_prepare(const nm_mlp* w, float* wperm, void* stream) {
  NM_LAUNCH(k_permute_weights, dim3(1), dim3(256), 0, (hipStream_t)stream, w->w0, w->w1, w->w2, wperm, (const float*)nullptr,
            (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
// both nets of a roll-out in one launch
int nm_material_prepare2(const nm_mlp* wa, float* wperm_a, const nm_mlp* wb, float* wperm_b, void* stream) {
  NM_LAUNCH(k_permute_weights, dim3(2), dim3(256), 0, (hipStream_t)stream, wa->w0, wa->w1, wa->w2, wperm_a, wb->w0, wb->w1, wb->w2,
            wperm_b);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
size_t nm_material_prepared_floats() { return NM_PERM_ALL; }

extern "C" int nm_material_fwd(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w, float* out,
                               void* stream) {
  NM_REQUIRE(n >= 0, "negative n");
  NM_REQUIRE(kind == NM_ELASTICITY || kind == NM_PLASTICITY, "kind must be NM_ELASTICITY or NM_PLASTICITY");
  if (n == 0) return NM_OK;
  NM_REQUIRE(F && out && w && w->w0 && w->w1 && w->w2, "null pointer");
  return nm_material_fwd_launch(n, kind, alpha, F, w, nullptr, out, nullptr, nullptr, stream);
}

// ---------------------------------------------------------------- backward
// optional fusions used by the roll-out's reverse sweep
struct BwdFuse {
  const float* trial_C;   // != NULL: F_in = (I + dt * trial_C) F for enabled particles (replaces a separate trial-F pass)
  const int* enabled;
  float dt;
  int add_to_gF;          // gF += result instead of gF = result
  int polar;              // 0: the reference's SVD adjoint (denominators clamped like warp's adj_svd3); 1: exact polar derivative
  const float* svd_in;    // != NULL: U | sigma | V of the input as the forward kernel left them (svd_store layout)
};
// per-wave buffers, one wave's side by side (round 6): every offset inside a wave's 26 KB is a 16-bit DS immediate on the wave's
// base register; as seven arrays of [4 waves] the ones beyond 64 KB cost a v_add_u32 per access (36 per tile, and VALU issue
// is what this kernel is made of)
struct BwdWave {
  float Z[64 * 17];     // features [particle][17]
  float GY[64 * 13];    // ybar     [particle][13] (rows 9..12 zero)
  float Y[64 * 9];      // forward y
  float GZ[64 * 13];    // zbar     [particle][13]
  float TA[64 * 17];    // [feature][16 particles + pad]: pre2bar
  float TB[64 * 17];    //   h2, then pre1bar
  float TC[64 * 17];    //   h1
};
struct BwdLds {
  float P0[16 * 64], P1[64 * 64], P2[16 * 64];
  float Q0[16 * 64], Q1[64 * 64], Q2[12 * 64];
  BwdWave W[4];
};

struct BwdArgs {
  int n, q;
  float alpha;
  const float *F, *w0, *w1, *w2, *wperm, *gout;
  float *gF, *wpart;
  int want_w;
  BwdFuse fz;
  const f4* act;      // activation cache written by the forward kernel (mlp_forward_tile), or NULL: recompute
};

// WW: the weight gradients are wanted, known at compile time (the roll-out's pair launches): as a run-time flag the test sat in
// front of every LDS write of the tile loop - a scalar branch per GELU pair, each the end of a basic block, so that no two
// GELU chains were ever interleaved (round 6)
template <int KIND, bool ACT, bool WW = false>
__device__ __forceinline__ void material_bwd_body(const BwdArgs& a, char* smem_raw) {
  const int n = a.n, q = a.q, wmode = a.want_w;
  const bool want_w = WW ? true : (wmode != 0);
  const float alpha = a.alpha;
  const float* __restrict__ F = a.F;
  const float *__restrict__ w0 = a.w0, *__restrict__ w1 = a.w1, *__restrict__ w2 = a.w2, *__restrict__ wperm = a.wperm;
  const float* __restrict__ gout = a.gout;
  float *__restrict__ gF = a.gF, *__restrict__ wpart = a.wpart;
  const BwdFuse fz = a.fz;
  BwdLds& L = *reinterpret_cast<BwdLds*>(smem_raw);
  NM_PH_DECL
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pbeg = min(n, (blockIdx.x * 4 + wave) * q), pend = min(n, pbeg + q);   // see nm_wave_quota
  // A round's loads - F, dL/dout, trial C', the SVD factors, the activation record of its first tile - are issued ONE ROUND AHEAD
  // (round 4): the wave is alone on its SIMD, and a round that started with its own loads sat through their whole HBM round trip
  // (the first record "took 8 k cycles to arrive", §5) four times per pair launch.  Loads are issued in the order they are
  // needed (vmcnt retires in order); `enabled` gates nothing any more - a disabled particle's values are replaced afterwards.
  // (the three 3x3 inputs stay in the shape their loads return them in - 16 + 16 + 4 bytes - until the round that uses them
  //  begins: as nine scalars the compiler re-packed the loaded registers right behind the load instructions, i.e. waited for
  //  the first round's HBM round trip before it had issued the rest of the loads, twice per body)
  struct M3Raw {
    f4 a, b;
    float c;
  };
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  auto raw_load = [](const float* __restrict__ ptr, M3Raw& r) {
    r.a = *reinterpret_cast<const f4u*>(ptr); r.b = *reinterpret_cast<const f4u*>(ptr + 4); r.c = ptr[8];
  };
  auto raw_m3 = [](const M3Raw& r) {
    M3 m;
    m.m[0] = r.a[0]; m.m[1] = r.a[1]; m.m[2] = r.a[2]; m.m[3] = r.a[3];
    m.m[4] = r.b[0]; m.m[5] = r.b[1]; m.m[6] = r.b[2]; m.m[7] = r.b[3]; m.m[8] = r.c;
    return m;
  };
  struct RoundIn {
    int en_p;
    M3Raw Fp, go, T;
    M3 U, V;
    float s[3];
  };
  // activation record of the tile about to be processed (ACT): requested one tile ahead - ACROSS the round boundaries too, a
  // wave's tiles are consecutive records - and never copied (round 5: as part of a round's inputs it was copied at the top of
  // the round, i.e. waited for right behind the weights in the first round)
  f4 nx[ACT ? NM_ACT_SLOTS : 1];
  auto load_first_act = [&](int c0_) {
    if (ACT) {
      const f4* at_ = a.act + (size_t)(c0_ >> 4) * NM_ACT_SLOTS * 64 + lane;
#pragma unroll
      for (int k = 0; k < NM_ACT_SLOTS; ++k) nx[k] = __builtin_nontemporal_load(&at_[k * 64]);
    }
  };
  auto load_round = [&](int c0_, RoundIn& o) {
    const int p_ = c0_ + lane;
    const bool valid_ = p_ < pend;
    o.en_p = (fz.trial_C && valid_) ? fz.enabled[p_] : 0;
    if (valid_) {
      raw_load(F + (size_t)9 * p_, o.Fp);
      raw_load(gout + (size_t)9 * p_, o.go);
      if (fz.trial_C) raw_load(fz.trial_C + (size_t)9 * p_, o.T);
    }
    if (fz.svd_in) {        // (workgroup-uniform)
      if (valid_) svd_load(fz.svd_in, n, p_, o.U, o.s, o.V);
      else { o.U = m3_ident(); o.V = m3_ident(); o.s[0] = o.s[1] = o.s[2] = 1.f; }
    }
  };
  // ... and the FIRST round's are issued here, in front of the weight staging: its round trip hides theirs.  Every wave of the
  // launch asks for its first round at the same moment (13 MB at once, and the staging cannot finish before the loads in front
  // of it have returned): the first tile's activation record - half of those bytes, not needed before the first tile - is
  // requested BEHIND the weights
  RoundIn ahead;
  if (pbeg < pend) load_round(pbeg, ahead);
  NM_SB();
  if (wperm) {
    static_assert(offsetof(BwdLds, W) == NM_PERM_ALL * sizeof(float), "P0..Q2 must be contiguous in operand order");
    if (ACT) {      // only the first layer is recomputed: its operands + the transposed ones
      stage_permuted2<16 * 64, NM_PERM_ALL - NM_PERM_FWD, NM_ACT_SLOTS>(wperm, L.P0, wperm + NM_PERM_FWD, L.Q0,
                                                          [&]() { load_first_act(pbeg < pend ? pbeg : 0); });      // (unconditional: NM_ACT_SLOTS loads)
    } else {
      stage_permuted<NM_PERM_ALL>(wperm, L.P0);
    }
    __syncthreads();
  } else {
    static_assert(sizeof(BwdWave) * 4 >= NM_RAWTOT * sizeof(float), "raw weights must fit the per-wave buffers");
    float* raw = &L.W[0].Z[0];   // the per-wave buffers (26 368 floats, contiguous) are free until the main loop
    stage_raw_weights(w0, w1, w2, raw);
    __syncthreads();
    stage_fwd_weights(raw, L.P0, L.P1, L.P2);
    stage_bwd_weights(raw, L.Q0, L.Q1, L.Q2);
    __syncthreads();
  }
  NM_PH(0)
  const int j = lane & 15, g = lane >> 4;
  float* zb = L.W[wave].Z;
  float* gyb = L.W[wave].GY;
  float* yb = L.W[wave].Y;
  float* gzb = L.W[wave].GZ;
  float* ta = L.W[wave].TA;
  float* tb = L.W[wave].TB;
  float* tc = L.W[wave].TC;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
  f4 gW1[4][4], gW0[4], gW2[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    gW0[a] = zero; gW2[a] = zero;
#pragma unroll
    for (int b = 0; b < 4; ++b) gW1[a][b] = zero;
  }
  for (int c0 = pbeg; c0 < pend; c0 += 64) {
    const int p = c0 + lane;
    const bool valid = p < pend;
    const int ntile = (min(64, pend - c0) + 15) >> 4;
    const f4* act_tile = ACT ? a.act + (size_t)(c0 >> 4) * NM_ACT_SLOTS * 64 + lane : nullptr;
    M3 Fp = valid ? raw_m3(ahead.Fp) : m3_ident(), go = valid ? raw_m3(ahead.go) : m3_zero();
    M3 T = (fz.trial_C && valid) ? raw_m3(ahead.T) : m3_zero();
    M3 R, U = ahead.U, V = ahead.V;
    float z[13], s[3] = {ahead.s[0], ahead.s[1], ahead.s[2]};
    const bool trial = ahead.en_p != 0;
    if (c0 + 64 < pend) load_round(c0 + 64, ahead);      // (wave-uniform)
    NM_SB();
    // roll-out: the forward pass fed the plasticity net I for a disabled particle (the fresh state of its next row, nm_grid.h)
    if (fz.trial_C && !trial) { Fp = m3_ident(); T = m3_zero(); }
    if (trial) {   // roll-out: input is the trial F = (I + dt C') F of mpm.py:489
#pragma unroll
      for (int i = 0; i < 9; ++i) T.m[i] *= fz.dt;
      T.m[0] += 1.f; T.m[4] += 1.f; T.m[8] += 1.f;
      Fp = m3_mul(T, Fp);
    }
    if (fz.svd_in) nm_features<true>(Fp, z, R, U, V, s);
    else nm_features(Fp, z, R, U, V, s);
#pragma unroll
    for (int c = 0; c < 13; ++c) zb[lane * 17 + c] = z[c];
    zb[lane * 17 + 13] = 0.f; zb[lane * 17 + 14] = 0.f; zb[lane * 17 + 15] = 0.f;
    // ybar = sym(Xbar):  elasticity Xbar = R^T g F ; plasticity Xbar = alpha R^T g
    M3 Rtg = m3_mul_tn(R, go);
    M3 Xb;
    if (KIND == NM_ELASTICITY) Xb = m3_mul(Rtg, Fp);
    else {
#pragma unroll
      for (int i = 0; i < 9; ++i) Xb.m[i] = alpha * Rtg.m[i];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) gyb[lane * 13 + 3 * r + c] = 0.5f * (Xb.m[3 * r + c] + Xb.m[3 * c + r]);
    gyb[lane * 13 + 9] = 0.f; gyb[lane * 13 + 10] = 0.f; gyb[lane * 13 + 11] = 0.f; gyb[lane * 13 + 12] = 0.f;
    __builtin_amdgcn_wave_barrier();
    NM_PH(1)

#pragma unroll 1
    for (int ct = 0; ct < ntile; ++ct) {
      // Order of the phases: every LDS operand is written at least one MFMA chain before it is read, and the first
      // operands of a phase are fetched before the chain in front of it starts - no LDS round trip is waited for.
      //   forward recompute (h1 -> TC under layer 1, h2 -> TB under layer 2)
      //   (b) h2bar = W2^T ybar, pre2bar = h2bar * gelu'(pre2)        [fetch (a)]
      //   (a) W2bar += ybar h2^T            (TB)                      [pre2bar -> TA ; fetch (d)]
      //   (d) h1bar = W1^T pre2bar, pre1bar = h1bar * gelu'(pre1)      [fetch (c)]
      //   (c) W1bar += pre2bar h1^T         (TA, TC)                  [pre1bar -> TB ; fetch (f)]
      //   (f) zbar = W0^T pre1bar                                      [fetch (e)]
      //   (e) W0bar += pre1bar z^T          (TB, Z)
      MlpFwd m;
      if (ACT) {
        // first layer recomputed (16 MFMAs + 16 GELU pairs), second layer's activations, derivatives and the output loaded
        f4 h2[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) { h2[rt] = nx[rt]; m.g2[rt] = nx[4 + rt]; }
        m.y = nx[8];
        const bool more = c0 + 16 * (ct + 1) < pend;      // (wave-uniform: the wave has another tile, in this round or the next)
        if (more) {
#pragma unroll
          for (int k = 0; k < 5; ++k) nx[k] = __builtin_nontemporal_load(&act_tile[((size_t)(ct + 1) * NM_ACT_SLOTS + k) * 64]);
        }
        float zin[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) zin[ks] = zb[(ct * 16 + j) * 17 + 4 * ks + g];
        AGroup<8> first;
        a_fetch<8>(first, L.P0, lane, 0);
        const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
        f4 a1[4] = {zero4, zero4, zero4, zero4};
        a_chain<16, 8, 3, 2>(first, L.P0, lane, zin, a1, [&]() {
          if (want_w) {       // h2 -> TB, transposed, as the recompute path leaves it (in the shadow of the layer-0 chain)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
              for (int r = 0; r < 4; ++r) tb[(16 * rt + 4 * g + r) * 17 + j] = h2[rt][r];
            __builtin_amdgcn_wave_barrier();
          }
        });
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            f2 h, dh;
            nm_gelu_both2((f2){a1[rt][2 * pr], a1[rt][2 * pr + 1]}, h, dh);
            m.g1[rt][2 * pr] = dh[0];
            m.g1[rt][2 * pr + 1] = dh[1];
            if (want_w) {     // h1 -> TC
              tc[(16 * rt + 4 * g + 2 * pr) * 17 + j] = h[0];
              tc[(16 * rt + 4 * g + 2 * pr + 1) * 17 + j] = h[1];
            }
          }
        __builtin_amdgcn_wave_barrier();
      } else {
        float zin[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) zin[ks] = zb[(ct * 16 + j) * 17 + 4 * ks + g];
        AGroup<8> first;
        a_fetch<8>(first, L.P0, lane, 0);
        mlp_forward_tile<true>(L.P0, L.P1, L.P2, zin, lane, first, m, want_w ? tc : nullptr, want_w ? tb : nullptr);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 4 * g + r;
        if (row < 9) yb[(ct * 16 + j) * 9 + row] = m.y[r];
      }
      NM_PH(2)
      const float* gyt = gyb + ct * 16 * 13;  // [particle][13]
      // ---- (b)
      f4 d2[4] = {zero, zero, zero, zero};
      float w2a[2], w2b[2][4];     // (a): A = ybar [yrow][particle], B = h2 [particle][feature] via TB
      {
        AGroup<4> q2g;
        a_fetch<4>(q2g, L.Q2, lane, 0);
        float b[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) b[ks] = gyt[j * 13 + 4 * ks + g];
        if (want_w) {
          w2a[0] = j < 9 ? gyt[g * 13 + j] : 0.f;
#pragma unroll
          for (int ctp = 0; ctp < 4; ++ctp) w2b[0][ctp] = tb[(16 * ctp + j) * 17 + g];
        }
        a_chain<12, 4, 3, 2>(q2g, L.Q2, lane, b, d2);
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) d2[rt] *= m.g2[rt];      // (vector form: packs into v_pk_mul_f32)
      NM_PH(4)
      AGroup<8> q1g;
      a_fetch<8>(q1g, L.Q1, lane, 0);
      NM_SB();
      // ---- (a)
      if (want_w) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) ta[(16 * rt + 4 * g + r) * 17 + j] = d2[rt][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int c = ks & 1, nx = c ^ 1;
          if (ks < 3) {
            w2a[nx] = j < 9 ? gyt[(4 * (ks + 1) + g) * 13 + j] : 0.f;
#pragma unroll
            for (int ctp = 0; ctp < 4; ++ctp) w2b[nx][ctp] = tb[(16 * ctp + j) * 17 + 4 * (ks + 1) + g];
          }
          NM_SB();
#pragma unroll
          for (int ctp = 0; ctp < 4; ++ctp) gW2[ctp] = NM_MFMA(w2a[c], w2b[c][ctp], gW2[ctp]);
          NM_SB();
        }
      }
      NM_PH(3)
      if (ACT && c0 + 16 * (ct + 1) < pend) {      // second half of the next tile's record
#pragma unroll
        for (int k = 5; k < NM_ACT_SLOTS; ++k) nx[k] = __builtin_nontemporal_load(&act_tile[((size_t)(ct + 1) * NM_ACT_SLOTS + k) * 64]);
      }
      // ---- (d)
      f4 d1[4] = {zero, zero, zero, zero};
      float w1a[2][4], w1b[2][4];  // (c): A = pre2bar via TA, B = h1 via TC
      {
        float b[16];
#pragma unroll
        for (int rtp = 0; rtp < 4; ++rtp)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) b[4 * rtp + reg] = d2[rtp][reg];
        if (want_w) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            w1b[0][t] = tc[(16 * t + j) * 17 + g];
            w1a[0][t] = ta[(16 * t + j) * 17 + g];
          }
        }
        a_chain<64, 8, 3, 2>(q1g, L.Q1, lane, b, d1);
      }
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) d1[rt] *= m.g1[rt];
      NM_PH(6)
      AGroup<8> q0g;
      a_fetch<8>(q0g, L.Q0, lane, 0);
      NM_SB();
      // ---- (c)
      if (want_w) {
        // TB (h2) was last read by (a), whose reads were issued long before these writes (one wave: LDS is in order)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) tb[(16 * rt + 4 * g + r) * 17 + j] = d1[rt][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int c = ks & 1, nx = c ^ 1;
          if (ks < 3) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              w1b[nx][t] = tc[(16 * t + j) * 17 + 4 * (ks + 1) + g];
              w1a[nx][t] = ta[(16 * t + j) * 17 + 4 * (ks + 1) + g];
            }
          }
          NM_SB();
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ctp = 0; ctp < 4; ++ctp) gW1[rt][ctp] = NM_MFMA(w1a[c][rt], w1b[c][ctp], gW1[rt][ctp]);
          NM_SB();
        }
      }
      NM_PH(5)
      // ---- (f)
      f4 dzz[2] = {zero, zero};
      float w0b[2], w0a[2][4];     // (e): A = pre1bar via TB, B = z [particle][z idx] straight from Z
      {
        float b[16];
#pragma unroll
        for (int rtp = 0; rtp < 4; ++rtp)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) b[4 * rtp + reg] = d1[rtp][reg];
        if (want_w) {
          w0b[0] = zb[(ct * 16 + g) * 17 + j];
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) w0a[0][rt] = tb[(16 * rt + j) * 17 + g];
        }
        a_chain<16, 8, 1, 0>(q0g, L.Q0, lane, b, dzz);
      }
      const f4 dza = dzz[0] + dzz[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = 4 * g + r;
        if (row < 13) gzb[(ct * 16 + j) * 13 + row] = dza[r];
      }
      // ---- (e)
      if (want_w) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int c = ks & 1, nx = c ^ 1;
          if (ks < 3) {
            w0b[nx] = zb[(ct * 16 + 4 * (ks + 1) + g) * 17 + j];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) w0a[nx][rt] = tb[(16 * rt + j) * 17 + 4 * (ks + 1) + g];
          }
          NM_SB();
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) gW0[rt] = NM_MFMA(w0a[c][rt], w0b[c], gW0[rt]);
          NM_SB();
        }
        // the next tile's forward pass overwrites TC / TB only after two of its own chains: in order behind these reads
      }
    }
    __builtin_amdgcn_wave_barrier();

    // per-particle epilogue: gradients through R X F^T / F + alpha R X and through the invariants
    // (roll-out: dL/dF of the sim step is already there and the elasticity path adds to it - asked for HERE, a few hundred
    //  instructions before it is needed: read at the point of use it was a round trip with nothing left to hide it)
    M3Raw prev_raw;
    const bool add_prev = fz.add_to_gF && valid;
    if (add_prev) raw_load(gF + (size_t)9 * p, prev_raw);
    M3 X;
#pragma unroll
    for (int i = 0; i < 9; ++i) X.m[i] = yb[lane * 9 + i];
    M3 Xs;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Xs.m[3 * r + c] = 0.5f * (X.m[3 * r + c] + X.m[3 * c + r]);
    float zbv[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) zbv[c] = gzb[lane * 13 + c];
    M3 Fb, Rb;
    if (KIND == NM_ELASTICITY) {
      M3 gFm = m3_mul(go, Fp);        // g F
      Rb = m3_mul(gFm, Xs);           // Rbar = g F X   (X symmetric)
      M3 RX = m3_mul(R, Xs);
      Fb = m3_mul_tn(go, RX);         // direct: g^T R X
    } else {
      M3 gX = m3_mul(go, Xs);
#pragma unroll
      for (int i = 0; i < 9; ++i) { Rb.m[i] = alpha * gX.m[i]; Fb.m[i] = go.m[i]; }
    }
    // sigma path + rotation path, both in the singular basis:  U [diag(sbar) + k_rc (At - At^T)_rc] V^T.
    // R = U V^T only: Ubar = Rbar V, Vbar = Rbar^T U, so U^T Ubar = At and V^T Vbar = At^T, and the general SVD adjoint
    // (E_ab = 1 / min(s_b^2 - s_a^2, -1e-6) for a < b, as warp's adj_svd3 clamps it) collapses to
    //   k_ab = (s_b - s_a) E_ab      - the reference's result: equals 1 / (s_a + s_b) away from the clamp and drops to
    //                                  1e6 (s_a - s_b) -> 0 where two singular values (nearly) coincide, e.g. at F = I;
    //   k_ab = 1 / (s_a + s_b)       - fz.polar: the exact derivative of the polar rotation (opt-in).
    M3 At = m3_mul(m3_mul_tn(U, Rb), V);
    M3 inner;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (r == c) { inner.m[4 * r] = zbv[r]; continue; }
        float kf;
        if (fz.polar) {
          float den = s[r] + s[c];
          den = fabsf(den) < 1e-6f ? copysignf(1e-6f, den) : den;
          kf = 1.f / den;
        } else {
          const int a = r < c ? r : c, b = r < c ? c : r;
          kf = (s[b] - s[a]) / fminf(s[b] * s[b] - s[a] * s[a], -1e-6f);
        }
        inner.m[3 * r + c] = (At.m[3 * r + c] - At.m[3 * c + r]) * kf;
      }
    M3 t = m3_mul_nt(m3_mul(U, inner), V);
    // G = F^T F:  F (Gbar + Gbar^T) ; det:  zbar12 * cof(F)
    M3 Gs;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Gs.m[3 * r + c] = zbv[3 + 3 * r + c] + zbv[3 + 3 * c + r];
    M3 FG = m3_mul(Fp, Gs);
    M3 cof = m3_cofactor(Fp);
#pragma unroll
    for (int i = 0; i < 9; ++i) Fb.m[i] += t.m[i] + FG.m[i] + zbv[12] * cof.m[i];
    if (valid) {
      if (add_prev) {
        const M3 prev = raw_m3(prev_raw);
#pragma unroll
        for (int i = 0; i < 9; ++i) Fb.m[i] += prev.m[i];
      }
      m3_store(gF + 9 * p, Fb);
    }
    __builtin_amdgcn_wave_barrier();
    NM_PH(7)
  }
  NM_PH_STORE_K(KIND == NM_PLASTICITY ? 1 : 0)

  if (!want_w) return;   // (workgroup-uniform)
  // combine the four waves' weight-gradient accumulators: every wave stores its own copy to a private LDS region (the weights
  // and per-wave buffers are dead by now), then the workgroup sums the four copies.
  // (LDS float atomics would serialise here: ds_add_f32 sustains ~0.3 lanes/clk/CU on gfx950, i.e. ~70k cycles for the
  // 24 576 lane-adds, against ~3k for this.)
  // The copies - and the per-workgroup partials in global memory - are in ACCUMULATOR order (NM_WACC floats: a lane's four
  // rows of one 16x16 block side by side, wacc_to_plain), not in (out, in) order: 24 16-byte LDS writes per lane instead of 96
  // scattered words, 24 16-byte reads and 6 + 6 16-byte global accesses per thread instead of 88 + 22 + 22 words; the one
  // place that needs (out, in) order is k_wgrad_reduce, once per roll-out.
  // want_w == 2: add to the partial this workgroup wrote in earlier launches (the roll-out sums over substeps and
  // reduces once); the order of additions is fixed, so the result stays deterministic.  All of a thread's old values are
  // requested HERE, in front of the LDS staging: the rolled read-add-write loop this replaces paid one L2 round trip per
  // iteration - 22 in a row at the end of every net, with nothing else left on the SIMD to hide them (round 5: the
  // ~15 k cycles per net that no phase counter covered).
  constexpr int W4 = NM_WACC / 4, WPER = W4 / 256;
  static_assert(W4 % 256 == 0, "whole 16-byte pieces per thread");
  f4* dst = reinterpret_cast<f4*>(wpart + (size_t)blockIdx.x * NM_WACC);
  f4 prev[WPER];
  if (wmode == 2) {
#pragma unroll
    for (int k = 0; k < WPER; ++k) prev[k] = dst[threadIdx.x + 256 * k];
  } else {
#pragma unroll
    for (int k = 0; k < WPER; ++k) prev[k] = zero;
  }
  NM_SB();
  __syncthreads();
  static_assert(sizeof(BwdLds) >= 4 * NM_WACC * sizeof(float), "four weight-gradient copies must fit");
  f4* red = reinterpret_cast<f4*>(smem_raw) + wave * W4;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
    for (int ctp = 0; ctp < 4; ++ctp) red[(rt * 4 + ctp) * 64 + lane] = gW1[rt][ctp];
    red[NM_WACC_W0 / 4 + rt * 64 + lane] = gW0[rt];
    red[NM_WACC_W2 / 4 + rt * 64 + lane] = gW2[rt];
  }
  __syncthreads();
  const f4* all = reinterpret_cast<const f4*>(smem_raw);
#pragma unroll
  for (int k = 0; k < WPER; ++k) {
    const int i = threadIdx.x + 256 * k;
    const f4 v = (all[i] + all[W4 + i]) + (all[2 * W4 + i] + all[3 * W4 + i]);
    dst[i] = prev[k] + v;      // (prev = 0 unless want_w == 2; x + 0 is exact)
  }
}

template <int KIND, bool ACT>
__global__ void __launch_bounds__(256, 1) k_material_bwd(BwdArgs a, GridPrologue pro) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // roll-out: grid restore + clear of the MPM adjoint that follows (nm_grid.h) - on workgroups of its own when CUs are to spare
  if (NM_PROLOGUE_SPLIT(pro)) return;
  material_bwd_body<KIND, ACT>(a, smem_raw);
}

// Roll-out reverse sweep: the elasticity adjoint of substep t and the plasticity adjoint of substep t-1 in ONE launch.  The
// second only needs the first's dL/dF of the same particle (a wave owns the same particles in both), so the pair saves a
// launch boundary - and k_material_bwd's boundaries are expensive: its waves own a SIMD's whole register file, so the
// neighbouring kernels cannot overlap its ramp-up / drain (~5 us).  `pro` is the grid prologue of substep t-1.
template <bool ACT, bool WW>
__global__ void __launch_bounds__(256, 1) k_material_bwd_pair(BwdArgs e, BwdArgs p, GridPrologue pro) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (NM_PROLOGUE_SPLIT(pro)) return;
  material_bwd_body<NM_ELASTICITY, ACT, WW>(e, smem_raw);
  __threadfence_block();     // dL/dF written by this workgroup's lanes is read back by the same lanes below
  __syncthreads();
  material_bwd_body<NM_PLASTICITY, ACT, WW>(p, smem_raw);
}

// sum the per-workgroup partials (accumulator order, NM_WACC floats each): a workgroup owns 64 consecutive elements, its four
// waves each sum a quarter of the partials (coalesced 256 B rows) and combine through LDS - deterministic - and the result
// goes to its place in (out, in) order
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float* __restrict__ wpart, int nparts, float* __restrict__ g0,
                                                      float* __restrict__ g1, float* __restrict__ g2, int accumulate,
                                                      const float* __restrict__ wpart_b, float* __restrict__ gb) {
  __shared__ float part[4][64];
  if (blockIdx.y == 1) {      // the second net of a roll-out in the same launch: its gradients are w0 | w1 | w2 back to back
    wpart = wpart_b; g0 = gb; g1 = gb + NM_W0; g2 = gb + NM_W0 + NM_W1;
  }
  const int li = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + li;      // (< NM_WACC: the grid is NM_WACC / 64 workgroups)
  const int i = wacc_to_plain(e);
  float acc = 0.f;
  if (i >= 0) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = sl;
    for (; b + 12 < nparts; b += 16) {
      a0 += wpart[(size_t)b * NM_WACC + e];
      a1 += wpart[(size_t)(b + 4) * NM_WACC + e];
      a2 += wpart[(size_t)(b + 8) * NM_WACC + e];
      a3 += wpart[(size_t)(b + 12) * NM_WACC + e];
    }
    for (; b < nparts; b += 4) a0 += wpart[(size_t)b * NM_WACC + e];
    acc = (a0 + a1) + (a2 + a3);
  }
  part[sl][li] = acc;
  __syncthreads();
  if (sl == 0 && i >= 0) {
    acc = (part[0][li] + part[1][li]) + (part[2][li] + part[3][li]);
    float* dst = i < NM_W0 ? g0 + i : (i < NM_W0 + NM_W1 ? g1 + (i - NM_W0) : g2 + (i - NM_W0 - NM_W1));
    *dst = accumulate ? *dst + acc : acc;
  }
}

// internal (also used by the fused roll-out): launch the backward kernel only.  wmode 0: no weight gradients,
// 1: write this launch's per-workgroup partial sums to wpart, 2: add them to wpart.
static int bwd_attr_once() {
  static bool attr_done[64] = {};       // (a function attribute belongs to the device it was set on)
  int devid = 0;
  NM_HIP_CHECK(hipGetDevice(&devid));
  const bool attr_set = devid >= 0 && devid < 64 && attr_done[devid];
  if (!attr_set) {
    NM_HIP_CHECK(hipFuncSetAttribute((const void*)k_material_bwd<NM_ELASTICITY, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(BwdLds)));
    NM_HIP_CHECK(hipFuncSetAttribute((const void*)k_material_bwd<NM_PLASTICITY, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(BwdLds)));
    NM_HIP_CHECK(hipFuncSetAttribute((const void*)k_material_bwd_pair<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(BwdLds)));
    NM_HIP_CHECK(hipFuncSetAttribute((const void*)k_material_bwd_pair<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(BwdLds)));
    NM_HIP_CHECK(hipFuncSetAttribute((const void*)k_material_bwd<NM_ELASTICITY, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(BwdLds)));
    NM_HIP_CHECK(hipFuncSetAttribute((const void*)k_material_bwd<NM_PLASTICITY, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(BwdLds)));
    NM_HIP_CHECK(hipFuncSetAttribute((const void*)k_material_bwd_pair<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(BwdLds)));
    NM_HIP_CHECK(hipFuncSetAttribute((const void*)k_material_bwd_pair<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(BwdLds)));
    if (devid >= 0 && devid < 64) attr_done[devid] = true;
  }
  return NM_OK;
}
static BwdArgs bwd_args(int32_t n, int q, float alpha, const float* F, const nm_mlp* w, const float* wperm, const float* gout,
                        float* gF, float* wpart, int wmode, const float* trial_C, const int* enabled, float dt, int flags,
                        const float* svd_in = nullptr, const float* act = nullptr) {
  BwdArgs a;
  a.n = n; a.q = q; a.alpha = alpha; a.F = F;
  a.w0 = w ? w->w0 : nullptr; a.w1 = w ? w->w1 : nullptr; a.w2 = w ? w->w2 : nullptr;
  a.wperm = wperm; a.gout = gout; a.gF = gF; a.wpart = wpart; a.want_w = wmode;
  a.fz.trial_C = trial_C; a.fz.enabled = enabled; a.fz.dt = dt; a.fz.add_to_gF = flags & 1; a.fz.polar = (flags >> 1) & 1;
  a.fz.svd_in = svd_in;
  a.act = reinterpret_cast<const f4*>(act);
  return a;
}

int nm_material_bwd_launch(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w, const float* wperm,
                           const float* gout, float* gF, float* wpart, int wmode, const float* trial_C, const int* enabled,
                           float dt, int add_to_gF, const GridPrologue* pro, void* stream, const float* svd_in, const float* act) {
  GridPrologue gp;
  if (pro) gp = *pro; else { memset(&gp, 0, sizeof(gp)); }
  hipStream_t s = (hipStream_t)stream;
  int grid, q;
  nm_wave_quota(n, grid, q);
  gp.mat_grid = grid;
  const int launch = (pro && grid + NM_PRO_WGS <= NM_BWD_GRID) ? grid + NM_PRO_WGS : grid;
  int rc = bwd_attr_once();
  if (rc) return rc;
  BwdArgs a = bwd_args(n, q, alpha, F, w, wperm, gout, gF, wpart, wmode, trial_C, enabled, dt, add_to_gF, svd_in, act);
  if (kind == NM_ELASTICITY && act)
    NM_LAUNCH((k_material_bwd<NM_ELASTICITY, true>), dim3(launch), dim3(256), sizeof(BwdLds), s, a, gp);
  else if (kind == NM_ELASTICITY)
    NM_LAUNCH((k_material_bwd<NM_ELASTICITY, false>), dim3(launch), dim3(256), sizeof(BwdLds), s, a, gp);
  else if (act)
    NM_LAUNCH((k_material_bwd<NM_PLASTICITY, true>), dim3(launch), dim3(256), sizeof(BwdLds), s, a, gp);
  else
    NM_LAUNCH((k_material_bwd<NM_PLASTICITY, false>), dim3(launch), dim3(256), sizeof(BwdLds), s, a, gp);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

// roll-out reverse sweep: elasticity adjoint (dL/dstress gS -> += gF) of one substep, then the plasticity adjoint of the
// substep before it (dL/dF = that gF -> gFtrial), one launch (k_material_bwd_pair)
int nm_material_bwd_pair_launch(int32_t n, const float* F_e, const nm_mlp* we, const float* wperm_e, const float* gS, float* gF,
                                float* wpart_e, int wmode_e, float alpha_p, const float* F_p, const nm_mlp* wp,
                                const float* wperm_p, float* gFtrial, float* wpart_p, int wmode_p, const float* trial_C,
                                const int* enabled, float dt, int polar, const GridPrologue* pro, void* stream,
                                const float* svd_in_e, const float* svd_in_p, const float* act_e, const float* act_p) {
  GridPrologue gp;
  if (pro) gp = *pro; else { memset(&gp, 0, sizeof(gp)); }
  hipStream_t s = (hipStream_t)stream;
  int grid, q;
  nm_wave_quota(n, grid, q);
  gp.mat_grid = grid;
  const int launch = (pro && grid + NM_PRO_WGS <= NM_BWD_GRID) ? grid + NM_PRO_WGS : grid;
  int rc = bwd_attr_once();
  if (rc) return rc;
  BwdArgs e = bwd_args(n, q, 0.f, F_e, we, wperm_e, gS, gF, wpart_e, wmode_e, nullptr, nullptr, 0.f, 1 | (polar ? 2 : 0), svd_in_e, act_e);
  BwdArgs p = bwd_args(n, q, alpha_p, F_p, wp, wperm_p, gF, gFtrial, wpart_p, wmode_p, trial_C, enabled, dt, polar ? 2 : 0, svd_in_p, act_p);
  const bool ww = wmode_e != 0 && wmode_p != 0;      // (the reverse sweep of a roll-out whose nets both train: the rule)
  if (act_e && act_p && ww)
    NM_LAUNCH((k_material_bwd_pair<true, true>), dim3(launch), dim3(256), sizeof(BwdLds), s, e, p, gp);
  else if (act_e && act_p)
    NM_LAUNCH((k_material_bwd_pair<true, false>), dim3(launch), dim3(256), sizeof(BwdLds), s, e, p, gp);
  else if (ww)
    NM_LAUNCH((k_material_bwd_pair<false, true>), dim3(launch), dim3(256), sizeof(BwdLds), s, e, p, gp);
  else
    NM_LAUNCH((k_material_bwd_pair<false, false>), dim3(launch), dim3(256), sizeof(BwdLds), s, e, p, gp);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
// internal: gw (+)= sum of the per-workgroup partials of a launch over n particles
int nm_material_wgrad_reduce(const float* wpart, int32_t n, float* gw0, float* gw1, float* gw2, int accumulate, void* stream) {
  int grid, q;
  nm_wave_quota(n, grid, q);
  NM_LAUNCH(k_wgrad_reduce, dim3(NM_WACC / 64), dim3(256), 0, (hipStream_t)stream, wpart, grid, gw0, gw1, gw2,
                     accumulate, (const float*)nullptr, (float*)nullptr);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
// both nets of a roll-out (gw_a, gw_b: w0 | w1 | w2 back to back, overwritten) in one launch
int nm_material_wgrad_reduce2(const float* wpart_a, const float* wpart_b, int32_t n, float* gw_a, float* gw_b, void* stream) {
  int grid, q;
  nm_wave_quota(n, grid, q);
  NM_LAUNCH(k_wgrad_reduce, dim3(NM_WACC / 64, 2), dim3(256), 0, (hipStream_t)stream, wpart_a, grid, gw_a, gw_a + NM_W0,
            gw_a + NM_W0 + NM_W1, 0, wpart_b, gw_b);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" size_t nm_material_bwd_workspace(int32_t n) {
  (void)n;
  return (size_t)NM_BWD_GRID * NM_WACC * sizeof(float);
}

extern "C" int nm_material_bwd(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w, const float* gout,
                               float* gF, float* gw0, float* gw1, float* gw2, void* workspace, size_t workspace_bytes,
                               void* stream) {
  return nm_material_bwd_ex(n, kind, alpha, F, w, gout, gF, gw0, gw1, gw2, 0, workspace, workspace_bytes, stream);
}

// flags: NM_BWD_ACCUMULATE (gw* += sum over particles instead of =), NM_BWD_POLAR_ADJOINT (exact polar derivative instead of
// the reference's clamped SVD adjoint)
extern "C" int nm_material_bwd_ex(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w, const float* gout,
                                  float* gF, float* gw0, float* gw1, float* gw2, int32_t flags, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  const int accumulate = flags & NM_BWD_ACCUMULATE;
  NM_REQUIRE(n >= 0, "negative n");
  NM_REQUIRE(kind == NM_ELASTICITY || kind == NM_PLASTICITY, "kind must be NM_ELASTICITY or NM_PLASTICITY");
  const int want_w = (gw0 && gw1 && gw2) ? 1 : 0;
  NM_REQUIRE(want_w || (!gw0 && !gw1 && !gw2), "weight gradient outputs must be all set or all NULL");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (want_w && !accumulate) {
      NM_HIP_CHECK(hipMemsetAsync(gw0, 0, NM_W0 * sizeof(float), s));
      NM_HIP_CHECK(hipMemsetAsync(gw1, 0, NM_W1 * sizeof(float), s));
      NM_HIP_CHECK(hipMemsetAsync(gw2, 0, NM_W2 * sizeof(float), s));
    }
    return NM_OK;
  }
  NM_REQUIRE(F && gout && gF && w && w->w0 && w->w1 && w->w2, "null pointer");
  if (want_w && (!workspace || workspace_bytes < nm_material_bwd_workspace(n))) {
    nm_set_error("material backward workspace too small: need %zu bytes, got %zu", nm_material_bwd_workspace(n),
                 workspace_bytes);
    return NM_ERR_WORKSPACE;
  }
  int rc = nm_material_bwd_launch(n, kind, alpha, F, w, nullptr, gout, gF, (float*)workspace, want_w, nullptr, nullptr, 0.f,
                                  (flags & NM_BWD_POLAR_ADJOINT) ? 2 : 0, nullptr, stream);
  if (rc) return rc;
  if (want_w) return nm_material_wgrad_reduce((const float*)workspace, n, gw0, gw1, gw2, accumulate, stream);
  return NM_OK;
}

// ---------------------------------------------------------------- LoRA merge (loralib.py:209-213)
// W_eff = W + scaling * B A for one dense layer, and its adjoint.  The matrices are tiny (<= 64x64, r = 16): one thread per
// output element; replaces a GEMM + scale + add chain of library kernels per layer and direction.
__global__ void __launch_bounds__(256) k_lora_merge(int out_f, int in_f, int r, float scaling, const float* __restrict__ W,
                                                    const float* __restrict__ B, const float* __restrict__ A,
                                                    float* __restrict__ Weff) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= out_f * in_f) return;
  int o = e / in_f, i = e - o * in_f;
  float acc = 0.f;
  for (int k = 0; k < r; ++k) acc = fmaf(B[o * r + k], A[k * in_f + i], acc);
  Weff[e] = fmaf(scaling, acc, W[e]);
}
__global__ void __launch_bounds__(256) k_lora_merge_bwd(int out_f, int in_f, int r, float scaling, const float* __restrict__ gW,
                                                        const float* __restrict__ B, const float* __restrict__ A,
                                                        float* __restrict__ gB, float* __restrict__ gA) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < out_f * r) {          // gB[o][k] = s * sum_i gW[o][i] A[k][i]
    int o = e / r, k = e - o * r;
    float acc = 0.f;
    for (int i = 0; i < in_f; ++i) acc = fmaf(gW[o * in_f + i], A[k * in_f + i], acc);
    gB[e] = scaling * acc;
    return;
  }
  e -= out_f * r;
  if (e < r * in_f) {           // gA[k][i] = s * sum_o B[o][k] gW[o][i]
    int k = e / in_f, i = e - k * in_f;
    float acc = 0.f;
    for (int o = 0; o < out_f; ++o) acc = fmaf(B[o * r + k], gW[o * in_f + i], acc);
    gA[e] = scaling * acc;
  }
}

extern "C" int nm_lora_merge(int32_t out_f, int32_t in_f, int32_t r, float scaling, const float* W, const float* B, const float* A,
                             float* Weff, void* stream) {
  NM_REQUIRE(out_f > 0 && in_f > 0 && r > 0, "bad LoRA shape");
  NM_REQUIRE(W && B && A && Weff, "null pointer");
  NM_LAUNCH(k_lora_merge, dim3(nm_div_up((int64_t)out_f * in_f, 256)), dim3(256), 0, (hipStream_t)stream, out_f, in_f, r, scaling,
                     W, B, A, Weff);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
extern "C" int nm_lora_merge_bwd(int32_t out_f, int32_t in_f, int32_t r, float scaling, const float* gW, const float* B,
                                 const float* A, float* gB, float* gA, void* stream) {
  NM_REQUIRE(out_f > 0 && in_f > 0 && r > 0, "bad LoRA shape");
  NM_REQUIRE(gW && B && A && gB && gA, "null pointer");
  NM_LAUNCH(k_lora_merge_bwd, dim3(nm_div_up((int64_t)(out_f + in_f) * r, 256)), dim3(256), 0, (hipStream_t)stream, out_f, in_f, r,
                     scaling, gW, B, A, gB, gA);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

// ---- all layers of a net in one launch (blockIdx.y = layer)
struct LoraJobs {
  nm_lora_layer l[NM_LORA_MAX_LAYERS];
};
__global__ void __launch_bounds__(256) k_lora_merge_layers(LoraJobs jobs) {
  const nm_lora_layer& L = jobs.l[blockIdx.y];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= L.out_f * L.in_f) return;
  const int o = e / L.in_f, i = e - o * L.in_f;
  // (the loads of a chunk are independent and issued together: a rolled loop pays one L2 round trip per term - 64 of them
  //  in the adjoint below - and these kernels were 9 / 20 us for a few kflop)
  const float* __restrict__ Bp = L.B + o * L.r;
  const float* __restrict__ Ap = L.A + i;
  float acc = 0.f;
  int k = 0;
  for (; k + 8 <= L.r; k += 8) {
    float b[8], a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { b[u] = Bp[k + u]; a[u] = Ap[(k + u) * L.in_f]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fmaf(b[u], a[u], acc);
  }
  for (; k < L.r; ++k) acc = fmaf(Bp[k], Ap[k * L.in_f], acc);
  L.o0[e] = fmaf(L.scaling, acc, L.W[e]);
}
__global__ void __launch_bounds__(256) k_lora_merge_layers_bwd(LoraJobs jobs) {
  const nm_lora_layer& L = jobs.l[blockIdx.y];
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < L.out_f * L.r) {          // gB[o][k] = s * sum_i gW[o][i] A[k][i]
    const int o = e / L.r, k = e - o * L.r;
    const float* __restrict__ Wp = L.W + o * L.in_f;
    const float* __restrict__ Ap = L.A + k * L.in_f;
    float acc = 0.f;
    int i = 0;
    for (; i + 8 <= L.in_f; i += 8) {
      float w[8], a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { w[u] = Wp[i + u]; a[u] = Ap[i + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(w[u], a[u], acc);
    }
    for (; i < L.in_f; ++i) acc = fmaf(Wp[i], Ap[i], acc);
    L.o0[e] = L.scaling * acc;
    return;
  }
  e -= L.out_f * L.r;
  if (e < L.r * L.in_f) {           // gA[k][i] = s * sum_o B[o][k] gW[o][i]
    const int k = e / L.in_f, i = e - k * L.in_f;
    const float* __restrict__ Bp = L.B + k;
    const float* __restrict__ Wp = L.W + i;
    float acc = 0.f;
    int o = 0;
    for (; o + 8 <= L.out_f; o += 8) {
      float b[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { b[u] = Bp[(o + u) * L.r]; w[u] = Wp[(o + u) * L.in_f]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(b[u], w[u], acc);
    }
    for (; o < L.out_f; ++o) acc = fmaf(Bp[o * L.r], Wp[o * L.in_f], acc);
    L.o1[e] = L.scaling * acc;
  }
}

static int lora_jobs(int32_t n, const nm_lora_layer* layers, bool bwd, LoraJobs* jobs, int* blocks) {
  NM_REQUIRE(layers && n >= 1 && n <= NM_LORA_MAX_LAYERS, "1..NM_LORA_MAX_LAYERS layers");
  memset(jobs, 0, sizeof(*jobs));
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    const nm_lora_layer& L = layers[i];
    NM_REQUIRE(L.out_f > 0 && L.in_f > 0 && L.r > 0, "bad LoRA shape");
    NM_REQUIRE(L.W && L.B && L.A && L.o0 && (!bwd || L.o1), "null pointer");
    jobs->l[i] = L;
    const int64_t work = bwd ? (int64_t)(L.out_f + L.in_f) * L.r : (int64_t)L.out_f * L.in_f;
    most = work > most ? work : most;
  }
  *blocks = nm_div_up(most, 256);
  return NM_OK;
}
extern "C" int nm_lora_merge_layers(int32_t n, const nm_lora_layer* layers, void* stream) {
  LoraJobs jobs;
  int blocks;
  int rc = lora_jobs(n, layers, false, &jobs, &blocks);
  if (rc) return rc;
  NM_LAUNCH(k_lora_merge_layers, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, jobs);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
extern "C" int nm_lora_merge_layers_bwd(int32_t n, const nm_lora_layer* layers, void* stream) {
  LoraJobs jobs;
  int blocks;
  int rc = lora_jobs(n, layers, true, &jobs, &blocks);
  if (rc) return rc;
  NM_LAUNCH(k_lora_merge_layers_bwd, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, jobs);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
