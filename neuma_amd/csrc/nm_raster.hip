// Particle-GS rasterizer (3DGS tile rasterizer, forward + backward) for gfx950.
//
// Replaces the un-vendored CUDA extension `diff_gaussian_rasterization` that the reference calls from
// modules/d3gs/gaussian_renderer/__init__.py:92-119 and modules/tune/utils.py:385-419.  Algorithm and
// constants follow the published 3DGS rasterizer (see SURVEY.md App. D and oracle/raster.py); the code is
// organised for 64-wide wavefronts:
//   * a 16x16 tile is four waves, each wave a 16x4 pixel strip; Gaussians of a tile are staged through LDS
//     in batches of 256 and read back as wave-uniform broadcasts;
//   * the backward pass never issues per-pixel atomics: each wave reduces a Gaussian's nine partial
//     gradients over its 64 lanes with DPP row operations (no LDS traffic), waves combine through
//     ds_add_f32 into a per-batch LDS table, and one thread per Gaussian flushes the tile total with global
//     atomics (one set per (tile, Gaussian) pair instead of one per (pixel, Gaussian));
//   * (tile, depth) ordering: Gaussians are depth-sorted once (K keys), pairs are emitted in that order and a STABLE
//     rocPRIM radix sort on the log2(tiles) tile bits alone finishes the job (2 passes over D instead of 6 over 64-bit keys).
#include "nm_common.h"

#include <rocprim/rocprim.hpp>

#define NM_TILE 16
#define NM_TPB 256

struct RK {
  int W, H, gx, gy, ty0, ty1, deg, M;
  float tanx, tany, fx, fy;
  float view[16], proj[16], cam[3], bg[3];
};

static int make_rk(const nm_raster_cfg* c, int m, RK& k) {
  NM_REQUIRE(c, "null raster cfg");
  NM_REQUIRE(c->image_width > 0 && c->image_height > 0, "bad image size");
  NM_REQUIRE(c->sh_degree >= 0 && c->sh_degree <= 3, "sh_degree must be 0..3");
  k.W = c->image_width; k.H = c->image_height;
  k.gx = (k.W + NM_TILE - 1) / NM_TILE; k.gy = (k.H + NM_TILE - 1) / NM_TILE;
  k.ty0 = 0; k.ty1 = k.gy;
  if (c->tile_y1 > c->tile_y0) {
    NM_REQUIRE(c->tile_y0 >= 0 && c->tile_y1 <= k.gy, "tile stripe out of range");
    k.ty0 = c->tile_y0; k.ty1 = c->tile_y1;
  }
  k.deg = c->sh_degree; k.M = m;
  k.tanx = c->tanfovx; k.tany = c->tanfovy;
  k.fx = k.W / (2.0f * c->tanfovx); k.fy = k.H / (2.0f * c->tanfovy);
  memcpy(k.view, c->viewmatrix, sizeof(k.view));
  memcpy(k.proj, c->projmatrix, sizeof(k.proj));
  memcpy(k.cam, c->campos, sizeof(k.cam));
  memcpy(k.bg, c->bg, sizeof(k.bg));
  return NM_OK;
}

// ---------------------------------------------------------------- buffer carving
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Geom {
  float2* xy; float* depth; float4* conop; float* rgb; uint32_t* clamped; uint32_t* tiles; uint32_t* offs; int* rad;
  uint32_t* order;                                 // Gaussian ids sorted by (depth, id); culled ones last
  uint32_t *dkey_in, *dkey_out, *dval_in, *tiles_sorted;
  void* scan_tmp; size_t scan_bytes; void* dsort_tmp; size_t dsort_bytes; size_t total;
};
static size_t scan_temp_bytes(int k) {
  size_t b = 0;
  (void)rocprim::inclusive_scan(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)(k > 0 ? k : 1),
                                rocprim::plus<uint32_t>());
  return b;
}
static size_t dsort_temp_bytes(int k) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (size_t)(k > 0 ? k : 1), 0u, 32u);
  return b;
}
static Geom carve_geom(void* base, int k) {
  Geom g; char* p = (char*)base; size_t o = 0; size_t K = (size_t)(k > 0 ? k : 1);
  g.xy = (float2*)(p + o); o += al256(K * sizeof(float2));
  g.depth = (float*)(p + o); o += al256(K * sizeof(float));
  g.conop = (float4*)(p + o); o += al256(K * sizeof(float4));
  g.rgb = (float*)(p + o); o += al256(K * 3 * sizeof(float));
  g.clamped = (uint32_t*)(p + o); o += al256(K * sizeof(uint32_t));
  g.tiles = (uint32_t*)(p + o); o += al256(K * sizeof(uint32_t));
  g.offs = (uint32_t*)(p + o); o += al256(K * sizeof(uint32_t));
  g.rad = (int*)(p + o); o += al256(K * sizeof(int));
  g.order = (uint32_t*)(p + o); o += al256(K * sizeof(uint32_t));
  g.dkey_in = (uint32_t*)(p + o); o += al256(K * sizeof(uint32_t));
  g.dkey_out = (uint32_t*)(p + o); o += al256(K * sizeof(uint32_t));
  g.dval_in = (uint32_t*)(p + o); o += al256(K * sizeof(uint32_t));
  g.tiles_sorted = (uint32_t*)(p + o); o += al256(K * sizeof(uint32_t));
  g.scan_bytes = scan_temp_bytes(k);
  g.scan_tmp = (void*)(p + o); o += al256(g.scan_bytes);
  g.dsort_bytes = dsort_temp_bytes(k);
  g.dsort_tmp = (void*)(p + o); o += al256(g.dsort_bytes);
  g.total = o;
  return g;
}
struct Binning { uint32_t* point_list; uint2* ranges; size_t total; };
static Binning carve_binning(void* base, int64_t D, int ntiles) {
  Binning b; char* p = (char*)base; size_t o = 0; size_t n = (size_t)(D > 0 ? D : 1);
  b.point_list = (uint32_t*)(p + o); o += al256(n * sizeof(uint32_t));
  b.ranges = (uint2*)(p + o); o += al256((size_t)(ntiles + 1) * sizeof(uint2));   // + one never-rendered dummy tile
  b.total = o;
  return b;
}
struct Scratch { uint32_t* keys_in; uint32_t* keys_out; uint32_t* vals_in; void* sort_tmp; size_t sort_bytes; size_t total; };
static Scratch carve_scratch(void* base, int64_t D) {
  Scratch s; char* p = (char*)base; size_t o = 0; size_t n = (size_t)(D > 0 ? D : 1);
  s.keys_in = (uint32_t*)(p + o); o += al256(n * sizeof(uint32_t));
  s.keys_out = (uint32_t*)(p + o); o += al256(n * sizeof(uint32_t));
  s.vals_in = (uint32_t*)(p + o); o += al256(n * sizeof(uint32_t));
  size_t b = 0, b16 = 0;     // the buffers are sized for 32-bit keys; 16-bit keys (<= 65535 tiles) use a prefix of them
  (void)rocprim::radix_sort_pairs(nullptr, b, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (uint32_t*)nullptr, n, 0u, 32u);
  (void)rocprim::radix_sort_pairs(nullptr, b16, (uint16_t*)nullptr, (uint16_t*)nullptr, (uint32_t*)nullptr,
                                  (uint32_t*)nullptr, n, 0u, 16u);
  s.sort_bytes = b > b16 ? b : b16;
  s.sort_tmp = (void*)(p + o); o += al256(b);
  s.total = o;
  return s;
}
struct Img { float* final_T; uint32_t* n_contrib; size_t total; };
static Img carve_img(void* base, int W, int H) {
  Img i; char* p = (char*)base; size_t o = 0; size_t n = (size_t)W * H;
  i.final_T = (float*)(p + o); o += al256(n * sizeof(float));
  i.n_contrib = (uint32_t*)(p + o); o += al256(n * sizeof(uint32_t));
  i.total = o;
  return i;
}

extern "C" size_t nm_raster_geom_bytes(int32_t k) { return carve_geom(nullptr, k).total; }
extern "C" size_t nm_raster_binning_bytes(int64_t D, const nm_raster_cfg* c) {
  int gx = (c->image_width + NM_TILE - 1) / NM_TILE, gy = (c->image_height + NM_TILE - 1) / NM_TILE;
  return carve_binning(nullptr, D, gx * gy).total;
}
extern "C" size_t nm_raster_scratch_bytes(int64_t D) { return carve_scratch(nullptr, D).total; }
extern "C" size_t nm_raster_image_bytes(const nm_raster_cfg* c) { return carve_img(nullptr, c->image_width, c->image_height).total; }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float3 xf43(const float* m, float x, float y, float z) {  // [p,1] * M, first 3 columns
  return make_float3(m[0] * x + m[4] * y + m[8] * z + m[12], m[1] * x + m[5] * y + m[9] * z + m[13],
                     m[2] * x + m[6] * y + m[10] * z + m[14]);
}
__device__ __forceinline__ float4 xf44(const float* m, float x, float y, float z) {
  return make_float4(m[0] * x + m[4] * y + m[8] * z + m[12], m[1] * x + m[5] * y + m[9] * z + m[13],
                     m[2] * x + m[6] * y + m[10] * z + m[14], m[3] * x + m[7] * y + m[11] * z + m[15]);
}
__device__ __forceinline__ void get_rect(const RK& k, float px, float py, int r, int& x0, int& y0, int& x1, int& y1, int ylo,
                                         int yhi) {
  x0 = min(k.gx, max(0, (int)((px - r) / NM_TILE)));
  y0 = min(yhi, max(ylo, (int)((py - r) / NM_TILE)));
  x1 = min(k.gx, max(0, (int)((px + r + NM_TILE - 1) / NM_TILE)));
  y1 = min(yhi, max(ylo, (int)((py + r + NM_TILE - 1) / NM_TILE)));
}


// Largest exponent `power` any pixel of the 16x16 tile (tx,ty) can see from a Gaussian at (mx,my) with conic
// (ca, cb, cc):  power = -0.5 q,  q(d) = ca dx^2 + 2 cb dx dy + cc dy^2 minimised over the tile's pixel rectangle
// (a convex quadratic over a box: zero if the centre is inside, else on one of the four edges).
struct TileCull {   // per-Gaussian constants of the tile test (the two divisions and the log are hoisted out of the tile loop)
  float mx, my, ca, cb, cc, cb_over_cc, cb_over_ca, qmax;
};
// fp contraction is switched off in the two functions below: the count pass (k_preprocess) and the emit pass
// (k_emit_keys) must take bit-identical decisions, and a multiply-add fused in one inlined copy but not in the other
// would let them disagree on a borderline tile.
__device__ __forceinline__ TileCull make_tile_cull(float mx, float my, const float4& co) {
#pragma clang fp contract(off)
  TileCull t;
  t.mx = mx; t.my = my; t.ca = co.x; t.cb = co.y; t.cc = co.z;
  t.cb_over_cc = co.y / co.z;
  t.cb_over_ca = co.y / co.x;
  // keep iff max power >= -log(255 o) - 0.01  <=>  min q <= 2 (log(255 o) + 0.01)
  t.qmax = 2.f * (__logf(255.f * co.w) + 0.01f);
  return t;
}
__device__ __forceinline__ float tile_min_q(const TileCull& t, int tx, int ty) {
#pragma clang fp contract(off)
  const float x0 = (float)(tx * NM_TILE), x1 = x0 + (float)(NM_TILE - 1);
  const float y0 = (float)(ty * NM_TILE), y1 = y0 + (float)(NM_TILE - 1);
  if (t.mx >= x0 && t.mx <= x1 && t.my >= y0 && t.my <= y1) return 0.f;
  float qmin = 3.0e38f;
#pragma unroll
  for (int e = 0; e < 2; ++e) {  // vertical edges x = x0 / x1: minimise over y
    float dx = (e ? x1 : x0) - t.mx;
    float y = fminf(fmaxf(t.my - t.cb_over_cc * dx, y0), y1);
    float dy = y - t.my;
    qmin = fminf(qmin, t.ca * dx * dx + 2.f * t.cb * dx * dy + t.cc * dy * dy);
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {  // horizontal edges y = y0 / y1: minimise over x
    float dy = (e ? y1 : y0) - t.my;
    float x = fminf(fmaxf(t.mx - t.cb_over_ca * dy, x0), x1);
    float dx = x - t.mx;
    qmin = fminf(qmin, t.ca * dx * dx + 2.f * t.cb * dx * dy + t.cc * dy * dy);
  }
  return qmin;
}
// A (Gaussian, tile) pair is kept iff some pixel of the tile can reach alpha >= 1/255 (the compositing kernels skip
// anything below, so dropping the pair leaves every pixel bit-identical).  0.01 of slack in the exponent keeps the
// test conservative against fp32 rounding of the per-pixel evaluation.  k_preprocess (count) and k_emit_keys (emit)
// evaluate this same function on the same inputs, so the two passes always agree.
__device__ __forceinline__ bool tile_contributes(const TileCull& t, int tx, int ty) {
  return !(tile_min_q(t, tx, ty) > t.qmax);
}

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                               0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// projected 2D covariance (a, b, c) + the 2x3 matrix T = J * Rv and the clamped view-space point
struct Cov2D { float a, b, c; float T[2][3]; float tx, ty, tz; float xmul, ymul; };
__device__ __forceinline__ Cov2D compute_cov2d(const RK& k, float mx, float my, float mz, const float* __restrict__ c6) {
  Cov2D o;
  float3 t = xf43(k.view, mx, my, mz);
  float limx = 1.3f * k.tanx, limy = 1.3f * k.tany;
  float txtz = t.x / t.z, tytz = t.y / t.z;
  o.xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  o.ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  o.tx = t.x; o.ty = t.y; o.tz = t.z;
  float j00 = k.fx / t.z, j02 = -(k.fx * t.x) / (t.z * t.z), j11 = k.fy / t.z, j12 = -(k.fy * t.y) / (t.z * t.z);
  // Rv[r][c] = view[4c + r]
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o.T[0][c] = j00 * k.view[4 * c + 0] + j02 * k.view[4 * c + 2];
    o.T[1][c] = j11 * k.view[4 * c + 1] + j12 * k.view[4 * c + 2];
  }
  float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  float ST0[3], ST1[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ST0[i] = S[i][0] * o.T[0][0] + S[i][1] * o.T[0][1] + S[i][2] * o.T[0][2];
    ST1[i] = S[i][0] * o.T[1][0] + S[i][1] * o.T[1][1] + S[i][2] * o.T[1][2];
  }
  o.a = o.T[0][0] * ST0[0] + o.T[0][1] * ST0[1] + o.T[0][2] * ST0[2] + 0.3f;
  o.b = o.T[0][0] * ST1[0] + o.T[0][1] * ST1[1] + o.T[0][2] * ST1[2];
  o.c = o.T[1][0] * ST1[0] + o.T[1][1] * ST1[1] + o.T[1][2] * ST1[2] + 0.3f;
  return o;
}

// ---------------------------------------------------------------- forward kernels
__global__ void __launch_bounds__(256) k_preprocess(RK k, int K, const float* __restrict__ means, const float* __restrict__ shs,
                                                    const float* __restrict__ colors, const float* __restrict__ opac,
                                                    const float* __restrict__ cov3D, int* __restrict__ radii, float2* __restrict__ xy,
                                                    float* __restrict__ depth, float4* __restrict__ conop, float* __restrict__ rgb,
                                                    uint32_t* __restrict__ clamped, uint32_t* __restrict__ tiles, int* __restrict__ grad_,
                                                    uint32_t* __restrict__ dkey, uint32_t* __restrict__ dval) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  radii[i] = 0;
  grad_[i] = 0;
  tiles[i] = 0;
  dkey[i] = 0xFFFFFFFFu;   // culled Gaussians sort last
  dval[i] = (uint32_t)i;
  float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
  float3 pv = xf43(k.view, mx, my, mz);
  if (!(pv.z > 0.2f)) return;  // near-plane cull
  float4 ph = xf44(k.proj, mx, my, mz);
  float pw = 1.0f / (ph.w + 0.0000001f);
  float ndx = ph.x * pw, ndy = ph.y * pw;
  float c6[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) c6[a] = cov3D[6 * i + a];
  Cov2D cv = compute_cov2d(k, mx, my, mz, c6);
  float det = cv.a * cv.c - cv.b * cv.b;
  if (det == 0.0f) return;
  float det_inv = 1.f / det;
  float mid = 0.5f * (cv.a + cv.c);
  float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  int r = (int)ceilf(3.f * sqrtf(lam));
  float px = ((ndx + 1.0f) * k.W - 1.0f) * 0.5f, py = ((ndy + 1.0f) * k.H - 1.0f) * 0.5f;
  int x0, y0, x1, y1;
  get_rect(k, px, py, r, x0, y0, x1, y1, 0, k.gy);
  if ((x1 - x0) * (y1 - y0) == 0) return;
  // colour
  uint32_t cl = 0;
  float col[3];
  if (colors) {
    col[0] = colors[3 * i]; col[1] = colors[3 * i + 1]; col[2] = colors[3 * i + 2];
  } else {
    float dx = mx - k.cam[0], dy = my - k.cam[1], dz = mz - k.cam[2];
    float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx * inv, y = dy * inv, z = dz * inv;
    const float* sh = shs + (size_t)i * k.M * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float res = SH_C0 * sh[ch];
      if (k.deg > 0) {
        res = res - SH_C1 * y * sh[3 + ch] + SH_C1 * z * sh[6 + ch] - SH_C1 * x * sh[9 + ch];
        if (k.deg > 1) {
          float xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
          res = res + SH_C2[0] * xy_ * sh[12 + ch] + SH_C2[1] * yz * sh[15 + ch] + SH_C2[2] * (2.f * zz - xx - yy) * sh[18 + ch] +
                SH_C2[3] * xz * sh[21 + ch] + SH_C2[4] * (xx - yy) * sh[24 + ch];
          if (k.deg > 2) {
            res = res + SH_C3[0] * y * (3.f * xx - yy) * sh[27 + ch] + SH_C3[1] * xy_ * z * sh[30 + ch] +
                  SH_C3[2] * y * (4.f * zz - xx - yy) * sh[33 + ch] + SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[36 + ch] +
                  SH_C3[4] * x * (4.f * zz - xx - yy) * sh[39 + ch] + SH_C3[5] * z * (xx - yy) * sh[42 + ch] +
                  SH_C3[6] * x * (xx - 3.f * yy) * sh[45 + ch];
          }
        }
      }
      res += 0.5f;
      if (res < 0.f) { cl |= (1u << ch); res = 0.f; }
      col[ch] = res;
    }
  }
  int sx0, sy0, sx1, sy1;
  get_rect(k, px, py, r, sx0, sy0, sx1, sy1, k.ty0, k.ty1);
  radii[i] = r;
  grad_[i] = r;
  xy[i] = make_float2(px, py);
  depth[i] = pv.z;
  conop[i] = make_float4(cv.c * det_inv, -cv.b * det_inv, cv.a * det_inv, opac[i]);
  rgb[3 * i] = col[0]; rgb[3 * i + 1] = col[1]; rgb[3 * i + 2] = col[2];
  clamped[i] = cl;
  {
    const float4 co = make_float4(cv.c * det_inv, -cv.b * det_inv, cv.a * det_inv, opac[i]);
    const TileCull tc = make_tile_cull(px, py, co);
    uint32_t cnt = 0;
    for (int y = sy0; y < sy1; ++y)
      for (int x = sx0; x < sx1; ++x) cnt += tile_contributes(tc, x, y) ? 1u : 0u;
    tiles[i] = cnt;
    if (cnt) dkey[i] = __float_as_uint(pv.z);   // positive floats order like their bit patterns
  }
}

__global__ void __launch_bounds__(256) k_gather_tiles(int K, const uint32_t* __restrict__ order, const uint32_t* __restrict__ tiles,
                                                      uint32_t* __restrict__ tiles_sorted) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < K) tiles_sorted[r] = tiles[order[r]];
}

// Pairs are emitted in depth-rank order, so the (tile, depth) order the compositor needs is obtained by a STABLE sort
// on the tile id alone (13 key bits instead of 45).  A lane loads the data of one rank; the tile rectangle of each
// Gaussian is then scanned by a group of 16 lanes (4 Gaussians per wave at a time, 16 rounds): the contributing tiles
// of a round are compacted with a ballot and written to consecutive slots, i.e. as full 64-byte segments.  (One
// thread per Gaussian writing its own pairs one at a time cost 4.6x the algorithmic write traffic in partial lines.)
template <typename KeyT>
__global__ void __launch_bounds__(256) k_emit_keys(RK k, int K, const uint32_t* __restrict__ order, const int* __restrict__ radii,
                                                   const float2* __restrict__ xy, const float4* __restrict__ conop,
                                                   const uint32_t* __restrict__ offs, const uint32_t* __restrict__ tiles,
                                                   KeyT* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  // ---- this lane's rank
  uint32_t my_i = 0, my_off = 0, my_end = 0;
  int rx0 = 0, ry0 = 0, rx1 = 0, ry1 = 0;
  TileCull my_tc = {0.f, 0.f, 1.f, 0.f, 1.f, 0.f, 0.f, -1.f};
  if (r < K) {
    my_i = order[r];
    if (tiles[my_i] != 0) {
      my_off = (r == 0) ? 0u : offs[r - 1];
      my_end = offs[r];
      const float2 p = xy[my_i];
      get_rect(k, p.x, p.y, radii[my_i], rx0, ry0, rx1, ry1, k.ty0, k.ty1);
      my_tc = make_tile_cull(p.x, p.y, conop[my_i]);
    }
  }
  // ---- 16 rounds: group g scans the rectangle of the rank held by lane (4 * round + g)
  for (int round = 0; round < 16; ++round) {
    const int src = 4 * round + grp;
    const uint32_t gi = __shfl(my_i, src, 64);
    uint32_t off = __shfl(my_off, src, 64);
    const uint32_t end = __shfl(my_end, src, 64);
    const int x0 = __shfl(rx0, src, 64), y0 = __shfl(ry0, src, 64), x1 = __shfl(rx1, src, 64), y1 = __shfl(ry1, src, 64);
    TileCull tc;
    tc.mx = __shfl(my_tc.mx, src, 64); tc.my = __shfl(my_tc.my, src, 64);
    tc.ca = __shfl(my_tc.ca, src, 64); tc.cb = __shfl(my_tc.cb, src, 64); tc.cc = __shfl(my_tc.cc, src, 64);
    tc.cb_over_cc = __shfl(my_tc.cb_over_cc, src, 64); tc.cb_over_ca = __shfl(my_tc.cb_over_ca, src, 64);
    tc.qmax = __shfl(my_tc.qmax, src, 64);
    const int w = x1 - x0, area = end > off ? w * (y1 - y0) : 0;
    const int steps = (area + 15) >> 4;
    for (int st = 0; st < steps; ++st) {            // group-uniform trip count
      const int t = st * 16 + sub;
      bool hit = false;
      int tx = 0, ty = 0;
      if (t < area) {
        ty = y0 + t / w; tx = x0 + t - (t / w) * w;
        hit = tile_contributes(tc, tx, ty);        // same predicate, same inputs as the count in k_preprocess
      }
      const uint32_t m16 = (uint32_t)((__ballot(hit) >> (16 * grp)) & 0xffffull);
      const uint32_t pos = off + (uint32_t)__popc(m16 & ((1u << sub) - 1u));
      if (hit && pos < end) {
        keys[pos] = (KeyT)(ty * k.gx + tx);
        vals[pos] = gi;
      }
      off += (uint32_t)__popc(m16);
    }
    // belt and braces: should the two passes ever disagree, no slot is left uninitialised - leftovers go to a dummy
    // tile (id gx*gy) that has a range entry but is never composited
    for (uint32_t q = off + (uint32_t)sub; q < end; q += 16u) {
      keys[q] = (KeyT)(k.gx * k.gy);
      vals[q] = gi;
    }
  }
}

template <typename KeyT>
__global__ void __launch_bounds__(256) k_tile_ranges(int64_t D, const KeyT* __restrict__ keys, uint2* __restrict__ ranges) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D) return;
  uint32_t t = keys[i];
  if (i == 0) ranges[t].x = 0;
  else {
    uint32_t tp = keys[i - 1];
    if (t != tp) { ranges[tp].y = (uint32_t)i; ranges[t].x = (uint32_t)i; }
  }
  if (i == D - 1) ranges[t].y = (uint32_t)D;
}

// front-to-back composite of one 16x16 tile (upstream renderCUDA forward)
__global__ void __launch_bounds__(NM_TPB) k_render(RK k, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
                                                   const float2* __restrict__ xy, const float* __restrict__ rgb,
                                                   const float4* __restrict__ conop, float* __restrict__ final_T,
                                                   uint32_t* __restrict__ n_contrib, float* __restrict__ out) {
  __shared__ float2 s_xy[NM_TPB];
  __shared__ float4 s_co[NM_TPB];
  __shared__ float s_rgb[NM_TPB * 3];
  const int tile_x = blockIdx.x, tile_y = blockIdx.y + k.ty0;
  const int tid = threadIdx.x;
  const int px = tile_x * NM_TILE + (tid & 15), py = tile_y * NM_TILE + (tid >> 4);
  const bool inside = px < k.W && py < k.H;
  const float fxp = (float)px, fyp = (float)py;
  const uint2 range = ranges[tile_y * k.gx + tile_x];
  int todo = (int)(range.y - range.x);
  const int rounds = (todo + NM_TPB - 1) / NM_TPB;
  bool done = !inside;
  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t contributor = 0, last = 0;
  for (int rd = 0; rd < rounds; ++rd, todo -= NM_TPB) {
    if (__syncthreads_count(done) == NM_TPB) break;
    int prog = rd * NM_TPB + tid;
    if (range.x + prog < range.y) {
      uint32_t id = plist[range.x + prog];
      s_xy[tid] = xy[id];
      s_co[tid] = conop[id];
      s_rgb[3 * tid] = rgb[3 * id]; s_rgb[3 * tid + 1] = rgb[3 * id + 1]; s_rgb[3 * tid + 2] = rgb[3 * id + 2];
    }
    __syncthreads();
    const int nb = min(NM_TPB, todo);
    for (int j = 0; !done && j < nb; ++j) {
      contributor++;
      float2 p = s_xy[j];
      float4 co = s_co[j];
      float dx = p.x - fxp, dy = p.y - fyp;
      float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
      if (power > 0.f) continue;
      float alpha = fminf(0.99f, co.w * __expf(power));
      if (alpha < 1.0f / 255.0f) continue;
      float test_T = T * (1.f - alpha);
      if (test_T < 0.0001f) { done = true; continue; }
      float w = alpha * T;
      C0 += s_rgb[3 * j] * w; C1 += s_rgb[3 * j + 1] * w; C2 += s_rgb[3 * j + 2] * w;
      T = test_T;
      last = contributor;
    }
  }
  if (inside) {
    size_t pix = (size_t)py * k.W + px, hw = (size_t)k.H * k.W;
    final_T[pix] = T;
    n_contrib[pix] = last;
    out[pix] = C0 + T * k.bg[0];
    out[hw + pix] = C1 + T * k.bg[1];
    out[2 * hw + pix] = C2 + T * k.bg[2];
  }
}

// ---------------------------------------------------------------- backward kernels
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// x + x(lane ^ 16) and x + x(lane ^ 32) on every lane with the gfx950 row / half swaps (two VALU instructions each) instead
// of ds_bpermute, whose LDS round trip sat on the dependent path of every evaluated (pixel, Gaussian) pair
__device__ __forceinline__ float add_xor16(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_xor32(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Sum 8 per-lane values over the 64 lanes of a wave with a folding butterfly: every step halves the number of
// live values per lane (lanes split on one index bit keep one half and hand the other half to their partner), so
// the whole reduction costs 4+2+1 DPP exchanges inside a row plus 3 single-value steps instead of 8 full
// reductions.  On return lane l (l < 8) holds the wave total of value  4*(l&1) + 2*((l>>1)&1) + ((l>>2)&1).
__device__ __forceinline__ float wave_fold8(const float* v, int lane) {
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  float a[4], c[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float keep = b0 ? v[i + 4] : v[i], send = b0 ? v[i] : v[i + 4];
    a[i] = keep + dpp_mov<0xB1>(send);   // quad_perm [1,0,3,2]: lane ^ 1
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float keep = b1 ? a[i + 2] : a[i], send = b1 ? a[i] : a[i + 2];
    c[i] = keep + dpp_mov<0x4E>(send);   // quad_perm [2,3,0,1]: lane ^ 2
  }
  float keep = b2 ? c[1] : c[0], send = b2 ? c[0] : c[1];
  float d = keep + dpp_mov<0x124>(send);  // row_ror:4 — partner has bit 2 flipped, bits 0,1 equal
  d += dpp_mov<0x128>(d);                 // row_ror:8 == lane ^ 8 within the row of 16
  return add_xor32(add_xor16(d));
}
__device__ __forceinline__ float wave_sum_dpp(float x) {
  x += dpp_mov<0xB1>(x);
  x += dpp_mov<0x4E>(x);
  x += dpp_mov<0x141>(x);  // row_half_mirror
  x += dpp_mov<0x140>(x);  // row_mirror
  return add_xor32(add_xor16(x));
}

#ifndef NM_RB_BATCH
#define NM_RB_BATCH 128
#endif
#define NM_NG 9  // per-Gaussian reduced quantities: ndc-mean(2) conic(3) colour(3) | opacity(1)
// slot order inside an accumulator row: [0..7] = values of wave_fold8 order, [8] = opacity
//   v[0]=d/dndc.x v[1]=d/dndc.y v[2]=d/dconic.x v[3]=d/dconic.y v[4]=d/dconic.z v[5..7]=d/drgb

template <bool WITH_OPACITY>
__global__ void __launch_bounds__(NM_TPB) k_render_bwd(RK k, const uint2* __restrict__ ranges, const uint32_t* __restrict__ plist,
                                                       const float2* __restrict__ xy, const float* __restrict__ rgb,
                                                       const float4* __restrict__ conop, const float* __restrict__ final_T,
                                                       const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                                                       float* __restrict__ acc /* (K, 9) */) {
  // Gaussians are staged NM_RB_BATCH at a time: the four per-wave tables are what limits the number of resident tiles
  // (LDS), and this loop lives on latency hiding - 128 per batch = 23 KB per tile = 6 waves per SIMD instead of 3
  __shared__ uint32_t s_id[NM_RB_BATCH];
  __shared__ float2 s_xy[NM_RB_BATCH];
  __shared__ float4 s_co[NM_RB_BATCH];
  __shared__ float s_rgb[NM_RB_BATCH * 3];
  __shared__ float s_acc[4][NM_RB_BATCH * NM_NG];   // one private table per wave: plain stores, no LDS atomics
  const int tile_x = blockIdx.x, tile_y = blockIdx.y + k.ty0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = tile_x * NM_TILE + (tid & 15), py = tile_y * NM_TILE + (tid >> 4);
  const bool inside = px < k.W && py < k.H;
  const float fxp = (float)px, fyp = (float)py;
  const uint2 range = ranges[tile_y * k.gx + tile_x];
  int todo = (int)(range.y - range.x);
  const int rounds = (todo + NM_RB_BATCH - 1) / NM_RB_BATCH;
  const size_t pix = (size_t)py * k.W + px, hw = (size_t)k.H * k.W;
  const float T_final = inside ? final_T[pix] : 0.f;
  float T = T_final;
  uint32_t contributor = (uint32_t)todo;
  const uint32_t last_contributor = inside ? n_contrib[pix] : 0u;
  float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f;
  if (inside) { dp0 = dL_dpix[pix]; dp1 = dL_dpix[hw + pix]; dp2 = dL_dpix[2 * hw + pix]; }
  const float bg_dot = k.bg[0] * dp0 + k.bg[1] * dp1 + k.bg[2] * dp2;
  float ar0 = 0.f, ar1 = 0.f, ar2 = 0.f, last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f;
  const float ddelx_dx = 0.5f * k.W, ddely_dy = 0.5f * k.H;
  float* my_acc = s_acc[wave];
  // lane l < 8 owns fold slot value index:
  const int slot = 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1);
  for (int i = lane; i < NM_RB_BATCH * NM_NG; i += 64) my_acc[i] = 0.f;
  // Gaussians behind every pixel's last contributor (the forward pass stopped compositing there) cannot
  // contribute: the wave skips them before doing any arithmetic, the tile skips whole batches of them
  uint32_t wave_last = last_contributor;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, o, 64));
  __shared__ uint32_t s_last[4];
  if (lane == 0) s_last[wave] = wave_last;
  __syncthreads();
  const uint32_t tile_last = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
  for (int rd = 0; rd < rounds; ++rd, todo -= NM_RB_BATCH) {
    // this batch covers list positions [todo - NM_RB_BATCH, todo): skip it entirely if all of them are >= tile_last
    if ((uint32_t)max(todo - NM_RB_BATCH, 0) >= tile_last) { contributor -= (uint32_t)min(NM_RB_BATCH, todo); continue; }
    __syncthreads();
    int prog = rd * NM_RB_BATCH + tid;
    if (tid < NM_RB_BATCH && range.x + prog < range.y) {
      uint32_t id = plist[range.y - prog - 1];  // back to front
      s_id[tid] = id;
      s_xy[tid] = xy[id];
      s_co[tid] = conop[id];
      s_rgb[3 * tid] = rgb[3 * id]; s_rgb[3 * tid + 1] = rgb[3 * id + 1]; s_rgb[3 * tid + 2] = rgb[3 * id + 2];
    }
    __syncthreads();
    const int nb = min(NM_RB_BATCH, todo);
    for (int j = 0; j < nb; ++j) {
      contributor--;
      if (contributor >= wave_last) continue;   // wave-uniform
      bool act = contributor < last_contributor;
      float2 p = s_xy[j];
      float4 co = s_co[j];
      float dx = p.x - fxp, dy = p.y - fyp;
      float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
      float G = __expf(power);
      float alpha = fminf(0.99f, co.w * G);
      act = act && !(power > 0.f) && !(alpha < 1.0f / 255.0f);
      if (__ballot(act) == 0ull) continue;  // whole wave skips this Gaussian
      float g[8], gop = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = 0.f;
      if (act) {
        // one hardware reciprocal (1 ulp) for both quotients below: the IEEE divisions were a quarter of the
        // instructions of an evaluated (pixel, Gaussian) pair; alpha <= 0.99 keeps the denominator >= 0.01
        const float inv1ma = __builtin_amdgcn_rcpf(1.f - alpha);
        T = T * inv1ma;
        float dch = alpha * T;
        float c0 = s_rgb[3 * j], c1 = s_rgb[3 * j + 1], c2 = s_rgb[3 * j + 2];
        ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0; lc0 = c0;
        ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1; lc1 = c1;
        ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2; lc2 = c2;
        float dL_dalpha = (c0 - ar0) * dp0 + (c1 - ar1) * dp1 + (c2 - ar2) * dp2;
        g[5] = dch * dp0; g[6] = dch * dp1; g[7] = dch * dp2;
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final * inv1ma) * bg_dot;
        float dL_dG = co.w * dL_dalpha;
        float gdx = G * dx, gdy = G * dy;
        float dG_ddelx = -gdx * co.x - gdy * co.y;
        float dG_ddely = -gdy * co.z - gdx * co.y;
        g[0] = dL_dG * dG_ddelx * ddelx_dx;   // d/d(ndc x)
        g[1] = dL_dG * dG_ddely * ddely_dy;
        g[2] = -0.5f * gdx * dx * dL_dG;       // d/d conic.x
        g[3] = -gdx * dy * dL_dG;              // d/d conic.y (full off-diagonal derivative)
        g[4] = -0.5f * gdy * dy * dL_dG;       // d/d conic.z
        gop = G * dL_dalpha;                   // d/d opacity
      }
#if defined(NM_RB_VARIANT) && NM_RB_VARIANT == 1
      float tot = g[0] + g[1] + g[2] + g[3] + g[4] + g[5] + g[6] + g[7];   // experiment: no cross-lane reduction
      if (tot == 12345.f) my_acc[j * NM_NG + slot] += tot;
#else
      float tot = wave_fold8(g, lane);
      if (lane < 8) my_acc[j * NM_NG + slot] += tot;
#endif
      if (WITH_OPACITY) {
        float to = wave_sum_dpp(gop);
        if (lane == 0) my_acc[j * NM_NG + 8] += to;
      }
    }
    __syncthreads();
    // one global atomic set per (tile, Gaussian): sum the four wave tables
    if (tid < nb) {
      uint32_t id = s_id[tid];
      float* dst = acc + (size_t)id * NM_NG;
#pragma unroll
      for (int q = 0; q < (WITH_OPACITY ? NM_NG : 8); ++q) {
        int o = tid * NM_NG + q;
        float v = (s_acc[0][o] + s_acc[1][o]) + (s_acc[2][o] + s_acc[3][o]);
        if (v != 0.f) unsafeAtomicAdd(dst + q, v);
      }
    }
    __syncthreads();
    for (int i = lane; i < nb * NM_NG; i += 64) my_acc[i] = 0.f;
  }
}

// adjoint of k_preprocess: per-Gaussian chain rule to means3D / cov3D / SH (upstream computeCov2DCUDA +
// preprocessCUDA backward)
__global__ void __launch_bounds__(256) k_preprocess_bwd(RK k, int K, const float* __restrict__ means, const float* __restrict__ shs,
                                                        const float* __restrict__ cov3D, const int* __restrict__ tiles_or_radii,
                                                        const uint32_t* __restrict__ clamped, const float* __restrict__ acc,
                                                        float* __restrict__ dmeans, float* __restrict__ dmeans2D,
                                                        float* __restrict__ dcov, float* __restrict__ dopac, float* __restrict__ dsh,
                                                        float* __restrict__ dcol, int has_sh) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K) return;
  float gm[3] = {0.f, 0.f, 0.f};
  float gc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool vis = tiles_or_radii[i] > 0;
  const float* a = acc + (size_t)i * NM_NG;
  if (vis) {
    float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    // ---- colour -> SH / view direction
    float dRGB[3] = {a[5], a[6], a[7]};
    if (has_sh) {
      uint32_t cl = clamped[i];
      float dx = mx - k.cam[0], dy = my - k.cam[1], dz = mz - k.cam[2];
      float len2 = dx * dx + dy * dy + dz * dz;
      float inv = 1.f / sqrtf(len2);
      float x = dx * inv, y = dy * inv, z = dz * inv;
      const float* sh = shs + (size_t)i * k.M * 3;
      float* gsh = dsh ? dsh + (size_t)i * k.M * 3 : nullptr;
      float ddir[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float dl = (cl & (1u << ch)) ? 0.f : dRGB[ch];
        float rx = 0.f, ry = 0.f, rz = 0.f;
        if (gsh) gsh[ch] = SH_C0 * dl;
        if (k.deg > 0) {
          float s1 = sh[3 + ch], s2 = sh[6 + ch], s3 = sh[9 + ch];
          if (gsh) { gsh[3 + ch] = -SH_C1 * y * dl; gsh[6 + ch] = SH_C1 * z * dl; gsh[9 + ch] = -SH_C1 * x * dl; }
          rx = -SH_C1 * s3; ry = -SH_C1 * s1; rz = SH_C1 * s2;
          if (k.deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
            float s4 = sh[12 + ch], s5 = sh[15 + ch], s6 = sh[18 + ch], s7 = sh[21 + ch], s8 = sh[24 + ch];
            if (gsh) {
              gsh[12 + ch] = SH_C2[0] * xy_ * dl; gsh[15 + ch] = SH_C2[1] * yz * dl;
              gsh[18 + ch] = SH_C2[2] * (2.f * zz - xx - yy) * dl; gsh[21 + ch] = SH_C2[3] * xz * dl;
              gsh[24 + ch] = SH_C2[4] * (xx - yy) * dl;
            }
            rx += SH_C2[0] * y * s4 + SH_C2[2] * 2.f * -x * s6 + SH_C2[3] * z * s7 + SH_C2[4] * 2.f * x * s8;
            ry += SH_C2[0] * x * s4 + SH_C2[1] * z * s5 + SH_C2[2] * 2.f * -y * s6 + SH_C2[4] * 2.f * -y * s8;
            rz += SH_C2[1] * y * s5 + SH_C2[2] * 4.f * z * s6 + SH_C2[3] * x * s7;
            if (k.deg > 2) {
              float s9 = sh[27 + ch], s10 = sh[30 + ch], s11 = sh[33 + ch], s12 = sh[36 + ch], s13 = sh[39 + ch],
                    s14 = sh[42 + ch], s15 = sh[45 + ch];
              if (gsh) {
                gsh[27 + ch] = SH_C3[0] * y * (3.f * xx - yy) * dl; gsh[30 + ch] = SH_C3[1] * xy_ * z * dl;
                gsh[33 + ch] = SH_C3[2] * y * (4.f * zz - xx - yy) * dl;
                gsh[36 + ch] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * dl;
                gsh[39 + ch] = SH_C3[4] * x * (4.f * zz - xx - yy) * dl; gsh[42 + ch] = SH_C3[5] * z * (xx - yy) * dl;
                gsh[45 + ch] = SH_C3[6] * x * (xx - 3.f * yy) * dl;
              }
              rx += SH_C3[0] * s9 * 6.f * xy_ + SH_C3[1] * s10 * yz + SH_C3[2] * s11 * -2.f * xy_ + SH_C3[3] * s12 * -6.f * xz +
                    SH_C3[4] * s13 * (4.f * zz - 3.f * xx - yy) + SH_C3[5] * s14 * 2.f * xz + SH_C3[6] * s15 * 3.f * (xx - yy);
              ry += SH_C3[0] * s9 * 3.f * (xx - yy) + SH_C3[1] * s10 * xz + SH_C3[2] * s11 * (4.f * zz - xx - 3.f * yy) +
                    SH_C3[3] * s12 * -6.f * yz + SH_C3[4] * s13 * -2.f * xy_ + SH_C3[5] * s14 * -2.f * yz + SH_C3[6] * s15 * -6.f * xy_;
              rz += SH_C3[1] * s10 * xy_ + SH_C3[2] * s11 * 8.f * yz + SH_C3[3] * s12 * 3.f * (2.f * zz - xx - yy) +
                    SH_C3[4] * s13 * 8.f * xz + SH_C3[5] * s14 * (xx - yy);
            }
          }
        }
        ddir[0] += rx * dl; ddir[1] += ry * dl; ddir[2] += rz * dl;
      }
      // through the normalisation  n = d / |d|
      float dot = x * ddir[0] + y * ddir[1] + z * ddir[2];
      gm[0] += (ddir[0] - x * dot) * inv;
      gm[1] += (ddir[1] - y * dot) * inv;
      gm[2] += (ddir[2] - z * dot) * inv;
    } else if (dcol) {
      dcol[3 * i] = dRGB[0]; dcol[3 * i + 1] = dRGB[1]; dcol[3 * i + 2] = dRGB[2];
    }
    // ---- conic -> cov2D -> cov3D and view-space point
    float c6[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) c6[q] = cov3D[6 * i + q];
    Cov2D cv = compute_cov2d(k, mx, my, mz, c6);
    float denom = cv.a * cv.c - cv.b * cv.b;
    float d2inv = 1.f / (denom * denom + 0.0000001f);
    if (denom * denom + 0.0000001f != 0.f) {
      float dA = a[2], dB = a[3], dC = a[4];
      float dL_da = d2inv * (-cv.c * cv.c * dA + cv.b * cv.c * dB - cv.b * cv.b * dC);
      float dL_dc = d2inv * (-cv.b * cv.b * dA + cv.a * cv.b * dB - cv.a * cv.a * dC);
      float dL_db = d2inv * (2.f * cv.b * cv.c * dA - (cv.a * cv.c + cv.b * cv.b) * dB + 2.f * cv.a * cv.b * dC);
      const float(*T)[3] = cv.T;
      gc[0] = T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
      gc[3] = T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
      gc[5] = T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
      gc[1] = 2.f * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2.f * T[1][0] * T[1][1] * dL_dc;
      gc[2] = 2.f * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2.f * T[1][0] * T[1][2] * dL_dc;
      gc[4] = 2.f * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2.f * T[1][1] * T[1][2] * dL_dc;
      // dL/dT
      float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
      float dT0[3], dT1[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        float st0 = S[q][0] * T[0][0] + S[q][1] * T[0][1] + S[q][2] * T[0][2];
        float st1 = S[q][0] * T[1][0] + S[q][1] * T[1][1] + S[q][2] * T[1][2];
        dT0[q] = 2.f * dL_da * st0 + dL_db * st1;
        dT1[q] = 2.f * dL_dc * st1 + dL_db * st0;
      }
      // dL/dJ[r][m] = sum_q dT[r][q] Rv[m][q],  Rv[m][q] = view[4q + m]
      float dJ00 = dT0[0] * k.view[0] + dT0[1] * k.view[4] + dT0[2] * k.view[8];
      float dJ02 = dT0[0] * k.view[2] + dT0[1] * k.view[6] + dT0[2] * k.view[10];
      float dJ11 = dT1[0] * k.view[1] + dT1[1] * k.view[5] + dT1[2] * k.view[9];
      float dJ12 = dT1[0] * k.view[2] + dT1[1] * k.view[6] + dT1[2] * k.view[10];
      float tz = 1.f / cv.tz, tz2 = tz * tz, tz3 = tz2 * tz;
      float dtx = cv.xmul * -k.fx * tz2 * dJ02;
      float dty = cv.ymul * -k.fy * tz2 * dJ12;
      float dtz = -k.fx * tz2 * dJ00 - k.fy * tz2 * dJ11 + (2.f * k.fx * cv.tx) * tz3 * dJ02 + (2.f * k.fy * cv.ty) * tz3 * dJ12;
      // t = Rv mu + tv  ->  dmu_c = sum_r Rv[r][c] dt_r = view[4c + r] dt_r
#pragma unroll
      for (int c = 0; c < 3; ++c) gm[c] += k.view[4 * c] * dtx + k.view[4 * c + 1] * dty + k.view[4 * c + 2] * dtz;
    }
    // ---- 2D mean -> 3D mean through the perspective divide
    float4 mh = xf44(k.proj, mx, my, mz);
    float mw = 1.0f / (mh.w + 0.0000001f);
    float mul1 = mh.x * mw * mw, mul2 = mh.y * mw * mw;
    float d0 = a[0], d1 = a[1];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      gm[c] += (k.proj[4 * c] * mw - k.proj[4 * c + 3] * mul1) * d0 + (k.proj[4 * c + 1] * mw - k.proj[4 * c + 3] * mul2) * d1;
  } else {
    if (has_sh && dsh) {
      float* gsh = dsh + (size_t)i * k.M * 3;
      for (int q = 0; q < k.M * 3; ++q) gsh[q] = 0.f;
    }
    if (!has_sh && dcol) { dcol[3 * i] = 0.f; dcol[3 * i + 1] = 0.f; dcol[3 * i + 2] = 0.f; }
  }
  if (vis && has_sh && dsh) {
    float* gsh = dsh + (size_t)i * k.M * 3;
    int used = (k.deg + 1) * (k.deg + 1);
    for (int q = used * 3; q < k.M * 3; ++q) gsh[q] = 0.f;
  }
  dmeans[3 * i] = gm[0]; dmeans[3 * i + 1] = gm[1]; dmeans[3 * i + 2] = gm[2];
  if (dmeans2D) { dmeans2D[3 * i] = vis ? a[0] : 0.f; dmeans2D[3 * i + 1] = vis ? a[1] : 0.f; dmeans2D[3 * i + 2] = 0.f; }
  if (dcov) {
#pragma unroll
    for (int q = 0; q < 6; ++q) dcov[6 * i + q] = gc[q];
  }
  if (dopac) dopac[i] = vis ? a[8] : 0.f;
}

// ---------------------------------------------------------------- host API
static int raster_preprocess_impl(bool wait, const nm_raster_cfg* cfg, int32_t K, int32_t m, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities, const float* cov3D, int32_t* radii,
                                    void* geom, size_t geom_bytes, int64_t* num_rendered, void* stream) {
  RK k;
  int rc = make_rk(cfg, m, k);
  if (rc) return rc;
  NM_REQUIRE(K >= 0 && num_rendered, "bad arguments");
  *num_rendered = 0;
  if (K == 0) return NM_OK;
  NM_REQUIRE((shs != nullptr) != (colors_precomp != nullptr), "provide exactly one of shs / colors_precomp");
  NM_REQUIRE(!shs || m >= (cfg->sh_degree + 1) * (cfg->sh_degree + 1), "shs has too few coefficients for sh_degree");
  NM_REQUIRE(means3D && opacities && cov3D && radii && geom, "null pointer");
  Geom g = carve_geom(geom, K);
  if (geom_bytes < g.total) { nm_set_error("geom buffer too small: need %zu got %zu", g.total, geom_bytes); return NM_ERR_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  NM_LAUNCH(k_preprocess, dim3(nm_div_up(K, 256)), dim3(256), 0, s, k, K, means3D, shs, colors_precomp, opacities, cov3D,
                     radii, g.xy, g.depth, g.conop, g.rgb, g.clamped, g.tiles, g.rad, g.dkey_in, g.dval_in);
  NM_LAUNCH_CHECK();
  // depth order of the Gaussians (stable: ties keep index order), then tile counts / offsets in that order
  size_t db = g.dsort_bytes;
  NM_HIP_CHECK(rocprim::radix_sort_pairs(g.dsort_tmp, db, g.dkey_in, g.dkey_out, g.dval_in, g.order, (size_t)K, 0u, 32u, s));
  NM_LAUNCH(k_gather_tiles, dim3(nm_div_up(K, 256)), dim3(256), 0, s, K, g.order, g.tiles, g.tiles_sorted);
  NM_LAUNCH_CHECK();
  size_t tb = g.scan_bytes;
  NM_HIP_CHECK(rocprim::inclusive_scan(g.scan_tmp, tb, g.tiles_sorted, g.offs, (size_t)K, rocprim::plus<uint32_t>(), s));
  if (!wait) {   // num_rendered is pinned host memory (pre-zeroed by the caller's *num_rendered = 0 above): low 32 bits arrive later
    NM_HIP_CHECK(hipMemcpyAsync(num_rendered, g.offs + (K - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return NM_OK;
  }
  uint32_t total = 0;
  NM_HIP_CHECK(hipMemcpyAsync(&total, g.offs + (K - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  NM_HIP_CHECK(hipStreamSynchronize(s));
  *num_rendered = (int64_t)total;
  return NM_OK;
}

extern "C" int nm_raster_preprocess(const nm_raster_cfg* cfg, int32_t K, int32_t m, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities, const float* cov3D, int32_t* radii,
                                    void* geom, size_t geom_bytes, int64_t* num_rendered, void* stream) {
  return raster_preprocess_impl(true, cfg, K, m, means3D, shs, colors_precomp, opacities, cov3D, radii, geom, geom_bytes, num_rendered, stream);
}
extern "C" int nm_raster_preprocess_async(const nm_raster_cfg* cfg, int32_t K, int32_t m, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities, const float* cov3D, int32_t* radii,
                                    void* geom, size_t geom_bytes, int64_t* num_rendered, void* stream) {
  return raster_preprocess_impl(false, cfg, K, m, means3D, shs, colors_precomp, opacities, cov3D, radii, geom, geom_bytes, num_rendered, stream);
}

extern "C" int nm_raster_render(const nm_raster_cfg* cfg, int32_t K, int64_t D, const void* geom, void* binning,
                                size_t binning_bytes, void* scratch, size_t scratch_bytes, void* image, size_t image_bytes,
                                float* out_color, void* stream) {
  RK k;
  int rc = make_rk(cfg, 0, k);
  if (rc) return rc;
  NM_REQUIRE(K >= 0 && D >= 0 && out_color && image && binning, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  Geom g = carve_geom((void*)geom, K);
  Binning b = carve_binning(binning, D, k.gx * k.gy);
  Img im = carve_img(image, k.W, k.H);
  if (binning_bytes < b.total) { nm_set_error("binning buffer too small: need %zu got %zu", b.total, binning_bytes); return NM_ERR_WORKSPACE; }
  if (image_bytes < im.total) { nm_set_error("image buffer too small: need %zu got %zu", im.total, image_bytes); return NM_ERR_WORKSPACE; }
  NM_HIP_CHECK(hipMemsetAsync(b.ranges, 0, ((size_t)k.gx * k.gy + 1) * sizeof(uint2), s));
  if (D > 0) {
    NM_REQUIRE(geom && scratch, "null geom/scratch");
    Scratch sc = carve_scratch(scratch, D);
    if (scratch_bytes < sc.total) { nm_set_error("scratch buffer too small: need %zu got %zu", sc.total, scratch_bytes); return NM_ERR_WORKSPACE; }
    int bits = 1;
    while ((1 << bits) <= k.gx * k.gy) ++bits;   // tile ids 0 .. gx*gy (the last one is the dummy tile)
    size_t tb = sc.sort_bytes;
    if (bits <= 16) {      // 16-bit tile keys: a quarter less sort traffic (6 instead of 8 bytes per pair and pass)
      uint16_t *k_in = (uint16_t*)sc.keys_in, *k_out = (uint16_t*)sc.keys_out;
      NM_LAUNCH(k_emit_keys<uint16_t>, dim3(nm_div_up(K, 256)), dim3(256), 0, s, k, K, g.order, (const int*)g.rad, g.xy, g.conop,
                g.offs, g.tiles, k_in, sc.vals_in);
      NM_LAUNCH_CHECK();
      NM_HIP_CHECK(rocprim::radix_sort_pairs(sc.sort_tmp, tb, k_in, k_out, sc.vals_in, b.point_list, (size_t)D, 0u, (unsigned)bits, s));
      NM_LAUNCH(k_tile_ranges<uint16_t>, dim3(nm_div_up(D, 256)), dim3(256), 0, s, D, (const uint16_t*)k_out, b.ranges);
      NM_LAUNCH_CHECK();
    } else {
      NM_LAUNCH(k_emit_keys<uint32_t>, dim3(nm_div_up(K, 256)), dim3(256), 0, s, k, K, g.order, (const int*)g.rad, g.xy, g.conop,
                g.offs, g.tiles, sc.keys_in, sc.vals_in);
      NM_LAUNCH_CHECK();
      NM_HIP_CHECK(rocprim::radix_sort_pairs(sc.sort_tmp, tb, sc.keys_in, sc.keys_out, sc.vals_in, b.point_list, (size_t)D, 0u,
                                             (unsigned)bits, s));
      NM_LAUNCH(k_tile_ranges<uint32_t>, dim3(nm_div_up(D, 256)), dim3(256), 0, s, D, (const uint32_t*)sc.keys_out, b.ranges);
      NM_LAUNCH_CHECK();
    }
  }
  NM_LAUNCH(k_render, dim3(k.gx, k.ty1 - k.ty0), dim3(NM_TPB), 0, s, k, b.ranges, b.point_list, g.xy, g.rgb, g.conop,
                     im.final_T, im.n_contrib, out_color);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" size_t nm_raster_bwd_workspace(int32_t K) { return al256((size_t)(K > 0 ? K : 1) * NM_NG * sizeof(float)); }

extern "C" int nm_raster_backward(const nm_raster_cfg* cfg, int32_t K, int32_t m, int64_t D, const float* means3D,
                                  const float* shs, const float* colors_precomp, const float* opacities, const float* cov3D,
                                  const void* geom, const void* binning, const void* image, const float* dL_dcolor,
                                  float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcov3D, float* dL_dopacity, float* dL_dshs,
                                  float* dL_dcolors, void* workspace, size_t workspace_bytes, void* stream) {
  (void)opacities; (void)colors_precomp;
  RK k;
  int rc = make_rk(cfg, m, k);
  if (rc) return rc;
  NM_REQUIRE(K >= 0 && D >= 0, "bad arguments");
  if (K == 0) return NM_OK;
  NM_REQUIRE(means3D && cov3D && geom && binning && image && dL_dcolor && dL_dmeans3D && workspace, "null pointer");
  if (workspace_bytes < nm_raster_bwd_workspace(K)) { nm_set_error("raster backward workspace too small"); return NM_ERR_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  Geom g = carve_geom((void*)geom, K);
  Binning b = carve_binning((void*)binning, D, k.gx * k.gy);
  Img im = carve_img((void*)image, k.W, k.H);
  float* acc = (float*)workspace;
  NM_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)K * NM_NG * sizeof(float), s));
  if (D > 0) {
    if (dL_dopacity)
      NM_LAUNCH(k_render_bwd<true>, dim3(k.gx, k.ty1 - k.ty0), dim3(NM_TPB), 0, s, k, b.ranges, b.point_list, g.xy, g.rgb, g.conop,
                im.final_T, im.n_contrib, dL_dcolor, acc);
    else
      NM_LAUNCH(k_render_bwd<false>, dim3(k.gx, k.ty1 - k.ty0), dim3(NM_TPB), 0, s, k, b.ranges, b.point_list, g.xy, g.rgb, g.conop,
                im.final_T, im.n_contrib, dL_dcolor, acc);
    NM_LAUNCH_CHECK();
  }
  NM_LAUNCH(k_preprocess_bwd, dim3(nm_div_up(K, 256)), dim3(256), 0, s, k, K, means3D, shs, cov3D, (const int*)g.tiles,
                     g.clamped, acc, dL_dmeans3D, dL_dmeans2D, dL_dcov3D, dL_dopacity, dL_dshs, dL_dcolors, shs ? 1 : 0);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
