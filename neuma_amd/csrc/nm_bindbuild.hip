// Particle -> Gaussian binding construction (data preparation).
//
// Behaviour: /root/reference/modules/d3gs/utils/binding_utils.py:199-285 (gaussian_binding_with_clip_v1) and 123-196
// (gaussian_binding): a particle j is bound to Gaussian k iff its Mahalanobis distance
//     p = d^T inv(cov_k) d,  d = x_j - mean_k                                   (:105-121)
// is <= chi2.ppf(confidence, 3); when more than max_particles qualify the max_particles with the smallest p are kept
// (:253-262); every kept particle gets the same weight 1/n (softmax of -ones, :259-269).
//
// The reference evaluates all K x N pairs (one kernel launch per Gaussian over all particles) and materialises a dense
// K x N fp32 matrix.  Here the particles are binned once into a uniform grid (counting sort with rocPRIM), and a thread
// per Gaussian visits only the cells under the axis-aligned box of its confidence ellipsoid
// (half extent sqrt(chi2 * cov_ii) per axis), keeping the max_particles best candidates in registers.  Output is the
// padded candidate table (K x max_particles, columns ascending) + the number kept per Gaussian, i.e. CSR after one
// prefix sum - never a dense matrix.
#include "nm_common.h"

#include <rocprim/rocprim.hpp>

#define NM_BIND_MAXP 16

struct BindGrid {
  float o[3];      // origin of cell (0,0,0)
  float inv_h;     // 1 / cell edge
  int n[3];        // cells per axis
};

__device__ __forceinline__ int bind_cell_coord(float x, float o, float inv_h, int n) {
  int c = (int)floorf((x - o) * inv_h);
  return max(0, min(c, n - 1));
}

__global__ void __launch_bounds__(256) k_bind_cell_keys(int N, BindGrid g, const float* __restrict__ x, uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int cx = bind_cell_coord(x[3 * i], g.o[0], g.inv_h, g.n[0]);
  int cy = bind_cell_coord(x[3 * i + 1], g.o[1], g.inv_h, g.n[1]);
  int cz = bind_cell_coord(x[3 * i + 2], g.o[2], g.inv_h, g.n[2]);
  keys[i] = (uint32_t)((cx * g.n[1] + cy) * g.n[2] + cz);
  vals[i] = (uint32_t)i;
}

// cell_start[c] = first sorted position of cell c (cell_start[ncells] = N)
__global__ void __launch_bounds__(256) k_bind_cell_start(int N, int ncells, const uint32_t* __restrict__ keys_sorted,
                                                         int* __restrict__ cell_start) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > N) return;
  uint32_t cur = i < N ? keys_sorted[i] : (uint32_t)ncells;
  uint32_t prev = i > 0 ? keys_sorted[i - 1] : 0xffffffffu;
  if (i == 0) {
    for (uint32_t c = 0; c <= cur && c <= (uint32_t)ncells; ++c) cell_start[c] = 0;
  } else if (cur != prev) {
    for (uint32_t c = prev + 1; c <= cur; ++c) cell_start[c] = i;
  }
}

__global__ void __launch_bounds__(128) k_bind_build(int K, BindGrid g, const float* __restrict__ means, const float* __restrict__ cov6,
                                                    const float* __restrict__ x, const uint32_t* __restrict__ order,
                                                    const int* __restrict__ cell_start, float threshold, int maxp,
                                                    int* __restrict__ counts, int* __restrict__ n_inside, int* __restrict__ cols,
                                                    float* __restrict__ pvals) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const float m0 = means[3 * k], m1 = means[3 * k + 1], m2 = means[3 * k + 2];
  M3 S;
  S.m[0] = cov6[6 * k]; S.m[1] = cov6[6 * k + 1]; S.m[2] = cov6[6 * k + 2];
  S.m[3] = S.m[1];      S.m[4] = cov6[6 * k + 3]; S.m[5] = cov6[6 * k + 4];
  S.m[6] = S.m[2];      S.m[7] = S.m[5];          S.m[8] = cov6[6 * k + 5];
  // inverse by cofactors (wp.inverse of binding_utils.py:22-45)
  M3 cof = m3_cofactor(S);
  const float det = S.m[0] * cof.m[0] + S.m[1] * cof.m[1] + S.m[2] * cof.m[2];
  const float idet = 1.f / det;
  M3 A;   // inv(S) = cof^T / det
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) A.m[3 * r + c] = cof.m[3 * c + r] * idet;
  // box of the ellipsoid d^T A d <= threshold: half extent sqrt(threshold * S_ii) (slightly inflated against rounding)
  const float e0 = sqrtf(fmaxf(threshold * S.m[0], 0.f)) * 1.0001f + 1e-12f;
  const float e1 = sqrtf(fmaxf(threshold * S.m[4], 0.f)) * 1.0001f + 1e-12f;
  const float e2 = sqrtf(fmaxf(threshold * S.m[8], 0.f)) * 1.0001f + 1e-12f;
  const int x0 = bind_cell_coord(m0 - e0, g.o[0], g.inv_h, g.n[0]), x1 = bind_cell_coord(m0 + e0, g.o[0], g.inv_h, g.n[0]);
  const int y0 = bind_cell_coord(m1 - e1, g.o[1], g.inv_h, g.n[1]), y1 = bind_cell_coord(m1 + e1, g.o[1], g.inv_h, g.n[1]);
  const int z0 = bind_cell_coord(m2 - e2, g.o[2], g.inv_h, g.n[2]), z1 = bind_cell_coord(m2 + e2, g.o[2], g.inv_h, g.n[2]);
  float bp[NM_BIND_MAXP];
  int bi[NM_BIND_MAXP];
#pragma unroll
  for (int q = 0; q < NM_BIND_MAXP; ++q) { bp[q] = 3.0e38f; bi[q] = 0x7fffffff; }
  int inside = 0;
  // NaN / non-finite covariances bind nothing (the reference would assert further down)
  const bool ok = isfinite(idet) && det != 0.f;
  if (ok) {
    for (int cx = x0; cx <= x1; ++cx)
      for (int cy = y0; cy <= y1; ++cy) {
        // cells z0..z1 of this (cx,cy) column are contiguous in the sorted order
        const int c0 = (cx * g.n[1] + cy) * g.n[2] + z0, c1 = (cx * g.n[1] + cy) * g.n[2] + z1;
        const int s0 = cell_start[c0], s1 = cell_start[c1 + 1];
        for (int s = s0; s < s1; ++s) {
          const int j = (int)order[s];
          const float d0 = x[3 * j] - m0, d1 = x[3 * j + 1] - m1, d2 = x[3 * j + 2] - m2;
          // p = (d^T A) d in the reference's association (:106-112)
          const float p11 = d0 * A.m[0] + d1 * A.m[3] + d2 * A.m[6];
          const float p12 = d0 * A.m[1] + d1 * A.m[4] + d2 * A.m[7];
          const float p13 = d0 * A.m[2] + d1 * A.m[5] + d2 * A.m[8];
          const float p = p11 * d0 + p12 * d1 + p13 * d2;
          if (!(p <= threshold)) continue;
          ++inside;
          // insert (p, j) into the sorted best list (ascending p, ties by ascending index); the worst entry falls off
          float cp = p;
          int ci = j;
#pragma unroll
          for (int q = 0; q < NM_BIND_MAXP; ++q) {
            const bool better = q < maxp && (cp < bp[q] || (cp == bp[q] && ci < bi[q]));
            const float tp = bp[q];
            const int ti = bi[q];
            bp[q] = better ? cp : tp;
            bi[q] = better ? ci : ti;
            cp = better ? tp : cp;
            ci = better ? ti : ci;
          }
        }
      }
  }
  const int cnt = min(inside, maxp);
  counts[k] = cnt;
  if (n_inside) n_inside[k] = inside;
  // columns ascending within the row (the order of the reference's to_sparse_coo of a dense row)
#pragma unroll
  for (int a = 1; a < NM_BIND_MAXP; ++a) {
#pragma unroll
    for (int b = a; b > 0; --b) {
      const bool sw = b < cnt && bi[b] < bi[b - 1];
      const int ti = bi[b]; const float tp = bp[b];
      bi[b] = sw ? bi[b - 1] : ti; bp[b] = sw ? bp[b - 1] : tp;
      bi[b - 1] = sw ? ti : bi[b - 1]; bp[b - 1] = sw ? tp : bp[b - 1];
    }
  }
#pragma unroll
  for (int q = 0; q < NM_BIND_MAXP; ++q) {
    if (q < maxp) {
      cols[(size_t)k * maxp + q] = q < cnt ? bi[q] : -1;
      if (pvals) pvals[(size_t)k * maxp + q] = q < cnt ? bp[q] : 0.f;
    }
  }
}

static inline size_t bb_al(size_t x) { return (x + 255) & ~(size_t)255; }
struct BindWs { uint32_t *keys_in, *keys_out, *vals_in, *vals_out; int* cell_start; void* sort_tmp; size_t sort_bytes; size_t total; };
static BindWs carve_bind_ws(void* base, int N, int ncells) {
  BindWs w; char* p = (char*)base; size_t o = 0; size_t n = (size_t)(N > 0 ? N : 1);
  w.keys_in = (uint32_t*)(p + o); o += bb_al(n * 4);
  w.keys_out = (uint32_t*)(p + o); o += bb_al(n * 4);
  w.vals_in = (uint32_t*)(p + o); o += bb_al(n * 4);
  w.vals_out = (uint32_t*)(p + o); o += bb_al(n * 4);
  w.cell_start = (int*)(p + o); o += bb_al(((size_t)ncells + 2) * 4);
  size_t tb = 0;
  rocprim::radix_sort_pairs(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n, 0u, 32u,
                            (hipStream_t)0);
  w.sort_bytes = tb;
  w.sort_tmp = (void*)(p + o); o += bb_al(tb);
  w.total = o;
  return w;
}

extern "C" size_t nm_bind_build_workspace(int32_t n_particles, int32_t ncells) { return carve_bind_ws(nullptr, n_particles, ncells).total; }

extern "C" int nm_bind_build(int32_t K, int32_t N, const float* means, const float* cov6, const float* particles,
                             const float* grid_origin, float cell, const int32_t* grid_dims, float threshold, int32_t max_particles,
                             int32_t* counts, int32_t* n_inside, int32_t* cols, float* pvals, void* workspace, size_t workspace_bytes,
                             void* stream) {
  NM_REQUIRE(K >= 0 && N >= 0, "negative sizes");
  NM_REQUIRE(max_particles >= 1 && max_particles <= NM_BIND_MAXP, "max_particles must be in [1,16]");
  if (K == 0) return NM_OK;
  NM_REQUIRE(means && cov6 && counts && cols && grid_origin && grid_dims, "null pointer");
  NM_REQUIRE(cell > 0.f && grid_dims[0] > 0 && grid_dims[1] > 0 && grid_dims[2] > 0, "bad grid");
  const int64_t nc64 = (int64_t)grid_dims[0] * grid_dims[1] * grid_dims[2];
  NM_REQUIRE(nc64 < (int64_t)1 << 30, "binding grid too large");
  const int ncells = (int)nc64;
  hipStream_t s = (hipStream_t)stream;
  BindWs w = carve_bind_ws(workspace, N, ncells);
  if (!workspace || workspace_bytes < w.total) {
    nm_set_error("binding workspace too small: need %zu got %zu", w.total, workspace_bytes);
    return NM_ERR_WORKSPACE;
  }
  BindGrid g;
  for (int a = 0; a < 3; ++a) { g.o[a] = grid_origin[a]; g.n[a] = grid_dims[a]; }
  g.inv_h = 1.f / cell;
  if (N > 0) {
    NM_REQUIRE(particles, "null particles");
    NM_LAUNCH(k_bind_cell_keys, dim3(nm_div_up(N, 256)), dim3(256), 0, s, N, g, particles, w.keys_in, w.vals_in);
    NM_LAUNCH_CHECK();
    int bits = 1;
    while (((int64_t)1 << bits) < nc64) ++bits;
    size_t tb = w.sort_bytes;
    NM_HIP_CHECK(rocprim::radix_sort_pairs(w.sort_tmp, tb, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)N, 0u, (unsigned)bits, s));
  }
  NM_LAUNCH(k_bind_cell_start, dim3(nm_div_up((int64_t)N + 1, 256)), dim3(256), 0, s, N, ncells, w.keys_out, w.cell_start);
  NM_LAUNCH_CHECK();
  NM_LAUNCH(k_bind_build, dim3(nm_div_up(K, 128)), dim3(128), 0, s, K, g, means, cov6, particles, w.vals_out, w.cell_start, threshold,
                     max_particles, counts, n_inside, cols, pvals);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
