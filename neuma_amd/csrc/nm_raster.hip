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
//   * (tile, depth) ordering without any global sort and without a host round trip: Gaussians are binned (integer global
//     atomics, ~3 per Gaussian) into cells = (64x64-pixel bin, one of 256 depth slabs between the view's nearest and farthest
//     visible Gaussian); every cell is sorted by (depth, index) in LDS by its own workgroup (bitonic network), which makes a
//     bin's cells, read in slab order, one depth-sorted list.  A 16x16 tile streams the list of its bin front to back in
//     batches of 256 candidates, keeps the ones that can reach one of its pixels (exact conic test) with an order-
//     preserving ballot compaction into LDS, composites them, and stops at saturation - on a dense scene that is after
//     ~5 % of the list, so the Gaussians behind an opaque surface are never duplicated per tile, sorted per tile or
//     read again.  The backward pass walks the same list in reverse from the last contributor.
#include "nm_common.h"

#define NM_TILE 16
#define NM_TPB 256

struct RK {
  int W, H, gx, gy, ty0, ty1, deg, M, K, items;
  float tanx, tany, fx, fy;
  float view[16], proj[16], cam[3], bg[3];
};

static int split_items_of(const nm_raster_cfg* c);
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
  k.deg = c->sh_degree; k.M = m; k.K = 0; k.items = split_items_of(c);
  k.tanx = c->tanfovx; k.tany = c->tanfovy;
  k.fx = k.W / (2.0f * c->tanfovx); k.fy = k.H / (2.0f * c->tanfovy);
  memcpy(k.view, c->viewmatrix, sizeof(k.view));
  memcpy(k.proj, c->projmatrix, sizeof(k.proj));
  memcpy(k.cam, c->campos, sizeof(k.cam));
  memcpy(k.bg, c->bg, sizeof(k.bg));
  return NM_OK;
}

// ---------------------------------------------------------------- state buffer (caller-owned, kept for the backward pass)
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

#define NM_BT 4      // a bin is NM_BT x NM_BT tiles (64 x 64 pixels)
#define NM_NS 256    // depth slabs per bin (one thread of a 256-thread workgroup each in the per-bin kernels)
#define NM_CELL_LDS 2048   // pairs a cell's workgroup sorts in LDS (bigger cells: same network on global memory)
#define NM_PAD 8           // words between two cell counters (32 bytes each: measured, 128-byte private lines bought nothing - 93.3 vs 92.7 us for k_bin_count - and cost 4x the memset / compact traffic)
#define NM_SPLIT_WORK 8192    // (tile, segment) work items a view may have (its segment records: 36 B per pixel each)
#define NM_SPLIT_BUSY 512     // a view with this many non-empty tiles (two per CU) fills the chip: no splitting (default of nm_raster_set_split)
#define NM_SPLIT_MINSEG 512   // shortest segment (list entries; default): ~50-90 us of one workgroup's walk
#define NM_SPLIT_WGS 4096     // segments aimed at (unhinted plan)
#define NM_HINT_WGS 12288     // segments a hinted plan may ask for (the caller sizes cfg.split_items from what it asked)
#define NM_PLAN_PER 8         // tiles per thread of the hinted plan (images up to 8192 tiles; larger ones are planned unhinted)
#define NM_HINT_FWD 0         // hinted plan: tiles planned longer than this are composited in parallel segments in the forward pass too
#define NM_HINT_MINSEG 256    // hinted plan: shortest segment
#define NM_SPLIT_TAU 0.01f    // a tile is split when, after its first segment, some pixel still has T above this
#define NM_SPLIT_FWD_MAX (2u << 20)   // forward splitting composites every segment of a split tile, also those behind the
                                      // point where its pixels stop: only when all candidate lists together are about one
                                      // chip-load of work (256 CUs x 8 workgroups x 1024 entries)

struct PairLog { uint32_t cell, rank, id, mask; };
#define NM_BC_OWN 2048       // k_bin_count: pairs of a wave whose owner lane is tabulated (beyond: binary search)
// Compositing record of one Gaussian (k_preprocess): everything the per-tile loops need, in one 64-byte line that a wave
// fetches with SCALAR loads - the operands are the same for all 64 lanes (one Gaussian against 64 pixels), so they belong in
// SGPRs, not in LDS: as wave-uniform LDS broadcasts (36 B per wave and Gaussian, 24 waves per CU on one LDS pipe) they were
// what bounded both compositing kernels.  a, b, c = the conic pre-multiplied so that the Gaussian's exponent comes out in base
// 2: log2(G) = a dx^2 + b dx dy + c dy^2  (a = -0.5 log2(e) conic.x, b = -log2(e) conic.y, c = -0.5 log2(e) conic.z).
struct __attribute__((aligned(64))) GRec { float x, y, a, b, c, lop, r, g, bl, op, pad[6]; };
#define NM_G 2       // Gaussians per scalar-fetch group of the forward composite (two groups per trip: 64 % (2 NM_G) == 0)
#define NM_LOG2E 1.4426950408889634f   // one (Gaussian, bin) pair as the count pass saw it

struct State {
  uint32_t* hdr;       // [2] pairs binned (may exceed cap) [3] overflow flag [4..5] exact pair count (stats) [6] largest cell
                       // [8] (tile, segment) work items of the split compositing [9] segment length [10] forward splitting allowed
                       // [11] hinted plan [12] work items the plan asked for [14..15] sum over the tiles of their lists' lengths
  float2* xy; float* depth; float4* conop; float* rgb; uint32_t* clamped; int* rad;
  GRec* recs;
  uint32_t* pad;       // per cell, one counter per 128-byte line (NM_PAD words apart): count, then fill cursor.  Neighbouring
                       // cells are hit by the same burst of atomics; packed they would serialise on one L2 channel
  uint32_t* cnt;       // per cell: count (compact copy of pad)
  uint32_t* off;       // per cell: exclusive offsets (ncell + 1)
  uint2* zrange;       // per k_preprocess workgroup: {min, max} depth bits of its visible Gaussians
  unsigned long long* keys;    // (depth bits << 32 | Gaussian id), cell-major; cap entries
  uint32_t* vals;              // tile mask of the pair: bit (ty % 4) * 4 + (tx % 4) set iff the Gaussian can reach that tile of the bin
  PairLog* log;                // the count pass's pair log (cap entries; dead after the fill pass)
  uint32_t* bin_total;
  // binning by depth-ordered chunks (k_bin_count2): the Gaussian sort's keys / dummy values, its slab counters | cursors |
  // header (one zeroed block) and offsets, per (chunk, bin) counts and offsets
  unsigned long long* gkeys; uint32_t* gvals; uint32_t* slab_blk; uint32_t* slab_off; uint32_t* hist; uint32_t* coff;
  int nchunk;
  float* final_T; uint32_t* n_contrib;
  // split compositing (k_split_plan): tiles of a view that leaves most of the chip idle are composited in list segments
  uint32_t* tile_rec;  // per tile: index of its first segment record, 0xFFFFFFFF = composited whole by k_render
  uint32_t* tile_ns;   // per tile: number of segments (the tile owns ns + 1 consecutive work items / records)
  uint32_t* tile_cnt;  // per tile: segments that have finished (arrival counter of the forward pass)
  uint32_t* tile_mode; // per tile: 1 = composited in segments (decided by the tile's first segment, k_render_seg stage 0)
  uint2* work;         // per (tile, segment) work item: {tile index ty * gx + tx, segment}
  float4* seg_raw;     // per work item and pixel: colour the segment composites from T = 1, and its transmittance (pass 1)
  float4* seg_fix;     // the same after pass 2: what the segment really contributes (pixels that stop inside / before it)
  float4* seg_ct;      // checkpoints: (C, T) of the pixel in front of the segment; the tile's extra item ns holds the final (C, T)
  uint32_t* seg_last;  // per work item and pixel: last contributor inside the segment (1-based list position), 0 = none
  uint32_t* seg_pos;   // per work item: list position its segment starts behind (segment s of a tile = positions
                       // (seg_pos[s], seg_pos[s + 1]]); s * seg by default, where the walk really stood for checkpoints
  int nbx, nby, ncell, items;
  size_t total, scratch_total;
};
// persistent part (read again by the reverse sweep) from `base`; forward-only scratch from `scratch`, or - scratch == NULL -
// behind the persistent part in the same buffer (the one-buffer layout of nm_raster_state_bytes / nm_raster_forward).
// items = capacity in (tile, segment) work items of the split compositing (cfg.split_items, 0 = NM_SPLIT_WORK).
static int split_items_of(const nm_raster_cfg* c) {
  int it = c->split_items > 0 ? c->split_items : NM_SPLIT_WORK;
  return it < 64 ? 64 : it;
}
static State carve_state(void* base, void* scratch, int W, int H, int k, int64_t cap, int items) {
  State t; char* p = (char*)base; size_t o = 0; size_t K = (size_t)(k > 0 ? k : 1), n = (size_t)W * H;
  const int gx = (W + NM_TILE - 1) / NM_TILE, gy = (H + NM_TILE - 1) / NM_TILE;
  t.nbx = (gx + NM_BT - 1) / NM_BT; t.nby = (gy + NM_BT - 1) / NM_BT; t.ncell = t.nbx * t.nby * NM_NS;
  t.items = items;
  const size_t cp = (size_t)(cap > 0 ? cap : 1), ntile = (size_t)gx * gy, it = (size_t)items;
  t.hdr = (uint32_t*)(p + o); o += 256;
  t.recs = (GRec*)(p + o); o += al256((K + 1) * sizeof(GRec));      // [K] = the null record
  t.clamped = (uint32_t*)(p + o); o += al256(K * sizeof(uint32_t));
  t.rad = (int*)(p + o); o += al256(K * sizeof(int));
  t.off = (uint32_t*)(p + o); o += al256(((size_t)t.ncell + 1) * sizeof(uint32_t));
  t.keys = (unsigned long long*)(p + o); o += al256(cp * sizeof(unsigned long long));
  t.vals = (uint32_t*)(p + o); o += al256(cp * sizeof(uint32_t));
  t.final_T = (float*)(p + o); o += al256(n * sizeof(float));
  t.n_contrib = (uint32_t*)(p + o); o += al256(n * sizeof(uint32_t));
  t.tile_rec = (uint32_t*)(p + o); o += al256(ntile * sizeof(uint32_t));
  t.tile_ns = (uint32_t*)(p + o); o += al256(ntile * sizeof(uint32_t));
  t.work = (uint2*)(p + o); o += al256(it * sizeof(uint2));
  t.seg_ct = (float4*)(p + o); o += al256(it * NM_TPB * sizeof(float4));
  t.seg_pos = (uint32_t*)(p + o); o += al256((it + 1) * sizeof(uint32_t));
  t.total = o;
  // ---- forward-only
  char* q = scratch ? (char*)scratch : p + o; size_t so = 0;
  t.xy = (float2*)(q + so); so += al256(K * sizeof(float2));
  t.depth = (float*)(q + so); so += al256(K * sizeof(float));
  t.conop = (float4*)(q + so); so += al256(K * sizeof(float4));
  t.rgb = (float*)(q + so); so += al256(K * 3 * sizeof(float));
  t.pad = (uint32_t*)(q + so); so += al256((size_t)t.ncell * NM_PAD * sizeof(uint32_t));
  t.cnt = (uint32_t*)(q + so); so += al256(((size_t)t.ncell + 1) * sizeof(uint32_t));
  t.zrange = (uint2*)(q + so); so += al256((K / 256 + 1) * sizeof(uint2));
  t.log = (PairLog*)(q + so); so += al256(cp * sizeof(PairLog));
  t.bin_total = (uint32_t*)(q + so); so += al256(((size_t)t.nbx * t.nby + 1) * sizeof(uint32_t));
  t.nchunk = (int)((K + 255) / 256);
  t.gkeys = (unsigned long long*)(q + so); so += al256(K * sizeof(unsigned long long));
  t.gvals = (uint32_t*)(q + so); so += al256(K * sizeof(uint32_t));
  t.slab_blk = (uint32_t*)(q + so); so += al256((2 * 1024 * 8 + 64) * sizeof(uint32_t));      // counts | cursors (NM_GS x NM_GSUB each) | hdr2
  t.slab_off = (uint32_t*)(q + so); so += al256((1024 + 1) * sizeof(uint32_t));
  t.hist = (uint32_t*)(q + so); so += al256((size_t)t.nchunk * t.nbx * t.nby * sizeof(uint32_t));
  t.coff = (uint32_t*)(q + so); so += al256(((size_t)t.nchunk * t.nbx * t.nby + 1) * sizeof(uint32_t));
  t.tile_cnt = (uint32_t*)(q + so); so += al256(ntile * sizeof(uint32_t));
  t.tile_mode = (uint32_t*)(q + so); so += al256(ntile * sizeof(uint32_t));
  t.seg_raw = (float4*)(q + so); so += al256(it * NM_TPB * sizeof(float4));
  t.seg_fix = (float4*)(q + so); so += al256(it * NM_TPB * sizeof(float4));
  t.seg_last = (uint32_t*)(q + so); so += al256(it * NM_TPB * sizeof(uint32_t));
  t.scratch_total = so;
  return t;
}

extern "C" size_t nm_raster_state_bytes(const nm_raster_cfg* c, int32_t k, int64_t cap_pairs) {
  if (!c) return 0;
  const State t = carve_state(nullptr, nullptr, c->image_width, c->image_height, k, cap_pairs, split_items_of(c));
  return t.total + t.scratch_total;
}
extern "C" int nm_raster_state_bytes_ex(const nm_raster_cfg* c, int32_t k, int64_t cap_pairs, size_t* state_bytes, size_t* scratch_bytes) {
  NM_REQUIRE(c && state_bytes && scratch_bytes, "null pointer");
  const State t = carve_state(nullptr, nullptr, c->image_width, c->image_height, k, cap_pairs, split_items_of(c));
  *state_bytes = t.total; *scratch_bytes = t.scratch_total;
  return NM_OK;
}

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
// fp contraction is switched off in the two functions below: the binning pass (k_bin_count) and the statistics pass
// (k_count_pairs) must take bit-identical decisions, and a multiply-add fused in one inlined copy but not in the other
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
// test conservative against fp32 rounding of the per-pixel evaluation.  k_bin_count (the pairs' tile masks) and
// k_count_pairs (statistics) evaluate this same function on the same inputs, so they always agree.
__device__ __forceinline__ bool tile_contributes(const TileCull& t, int tx, int ty) {
  return !(tile_min_q(t, tx, ty) > t.qmax);
}

// The same decision for the FOUR tiles of a tile row at once (round 5: the per-tile form above was 79 of the count pass's 105
// us - ~60 VALU operations per tile, 8.5 tiles per (Gaussian, bin) pair on average and as many loop trips as the wave's
// largest rectangle).  The region {q <= qmax} is an ellipse; cut by the row's strip y0 <= y <= y1 it stays convex, so the
// tiles of the row it touches are exactly those whose x range meets the x extent [xa, xb] of that cut.  The right end of the
// ellipse's horizontal chord at height dy, (-cb dy + sqrt(ca qmax - det dy^2)) / ca, is concave in dy: its maximum over the
// strip sits at the ellipse's right extreme point clamped into the strip (likewise the minimum of the left end) - two square
// roots per row instead of four edge minimisations per tile, the same number of operations for every lane.  The extent is
// widened by NM_ROW_EPS pixels (the per-pixel evaluation decides; a superset costs a candidate, a subset an error).
struct RowCull {
  float mx, my, ca, cb, cc, qmax, dye, cbcc_dxe;      // dye: half extent in y; cbcc_dxe = (cb / cc) * (half extent in x)
};
#define NM_ROW_EPS 0.01f
__device__ __forceinline__ RowCull make_row_cull(float mx, float my, const float4& co) {
#pragma clang fp contract(off)
  RowCull r;
  r.mx = mx; r.my = my; r.ca = co.x; r.cb = co.y; r.cc = co.z;
  r.qmax = 2.f * (__logf(255.f * co.w) + 0.01f);       // (as in make_tile_cull)
  const float det = co.x * co.z - co.y * co.y;
  const bool ok = det > 0.f && r.qmax > 0.f && co.x > 0.f && co.z > 0.f;
  const float dxe = ok ? __builtin_amdgcn_sqrtf(r.qmax * co.z / det) : 0.f;
  r.dye = ok ? __builtin_amdgcn_sqrtf(r.qmax * co.x / det) : -1.f;          // (-1: nothing passes)
  r.cbcc_dxe = ok ? (co.y / co.z) * dxe : 0.f;
  return r;
}
// tile mask (bit (ty % 4) * 4 + (tx % 4)) of bin (bx, by) for a Gaussian whose 3-sigma tile rectangle is [g.x, g.z) x [g.y, g.w)
__device__ __forceinline__ uint32_t bin_tile_mask(const RowCull& r, const int4& g, int bx, int by) {
#pragma clang fp contract(off)
  uint32_t m = 0u;
  const float det = r.ca * r.cc - r.cb * r.cb, inv_ca = 1.f / r.ca, caq = r.ca * r.qmax;
  const int c0 = bx * NM_BT, c_lo = max(g.x, c0), c_hi = min(g.z, c0 + NM_BT) - 1;      // columns of the bin inside the rectangle
#pragma unroll
  for (int q = 0; q < NM_BT; ++q) {
    const int ty = by * NM_BT + q;
    const float y0 = (float)(ty * NM_TILE) - r.my, y1 = y0 + (float)(NM_TILE - 1);
    const float lo = fmaxf(y0, -r.dye), hi = fminf(y1, r.dye);
    const bool row_ok = ty >= g.y && ty < g.w && !(lo > hi);          // (dye = -1 makes lo > hi)
    const float dyR = fminf(fmaxf(-r.cbcc_dxe, lo), hi), dyL = fminf(fmaxf(r.cbcc_dxe, lo), hi);
    const float sR = __builtin_amdgcn_sqrtf(fmaxf(0.f, caq - det * dyR * dyR));
    const float sL = __builtin_amdgcn_sqrtf(fmaxf(0.f, caq - det * dyL * dyL));
    const float xb = r.mx + (sR - r.cb * dyR) * inv_ca + NM_ROW_EPS;
    const float xa = r.mx - (sL + r.cb * dyL) * inv_ca - NM_ROW_EPS;
    // tile tx covers x in [16 tx, 16 tx + 15]: touched iff 16 tx <= xb and 16 tx + 15 >= xa
    const int ta = max(c_lo, (int)ceilf((xa - (float)(NM_TILE - 1)) * (1.f / NM_TILE)));
    const int tb = min(c_hi, (int)floorf(xb * (1.f / NM_TILE)));
    if (row_ok && ta <= tb) {
      const uint32_t bits = ((2u << (tb - c0)) - 1u) & ~((1u << (ta - c0)) - 1u);
      m |= bits << (q * NM_BT);
    }
  }
  return m;
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

// One Gaussian's SH coefficients into registers: 16-byte loads when the row allows it (degree 1 and 3: 12 / 48 floats per
// row; rows of 576 bytes) instead of 48 scalar loads 576 bytes apart between lanes
__device__ __forceinline__ void load_sh_row(const float* __restrict__ shs, size_t i, int M, float* sh) {
  const int nf = 3 * M;
  const float* row = shs + i * (size_t)nf;
  if ((nf & 3) == 0 && (reinterpret_cast<uintptr_t>(shs) & 15) == 0) {
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      if (4 * q < nf) {
        const float4 v = reinterpret_cast<const float4*>(row)[q];
        sh[4 * q] = v.x; sh[4 * q + 1] = v.y; sh[4 * q + 2] = v.z; sh[4 * q + 3] = v.w;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 48; ++q) if (q < nf) sh[q] = row[q];
  }
}

// ---------------------------------------------------------------- forward kernels
__global__ void __launch_bounds__(256) k_preprocess(RK k, int K, const float* __restrict__ means, const float* __restrict__ shs,
                                                    const float* __restrict__ colors, const float* __restrict__ opac,
                                                    const float* __restrict__ cov3D, int* __restrict__ radii, float2* __restrict__ xy,
                                                    float* __restrict__ depth, float4* __restrict__ conop, float* __restrict__ rgb,
                                                    uint32_t* __restrict__ clamped, int* __restrict__ grad_, uint2* __restrict__ zrange,
                                                    GRec* __restrict__ recs, uint32_t* __restrict__ hdr,
                                                    uint32_t* __restrict__ zero_words, int n_zero) {
  if (blockIdx.x == 0 && threadIdx.x < 64) hdr[threadIdx.x] = 0u;      // the view's header words (counters, flags): no memset launch for them
  // ... nor for the slab counters of the binning that follows (64 KB: a fill launch of its own was the first thing on a view's stream)
  for (int z = blockIdx.x * blockDim.x + threadIdx.x; z < n_zero; z += gridDim.x * blockDim.x) zero_words[z] = 0u;
  __shared__ uint32_t s_lo[4], s_hi[4];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  // depth range of the visible Gaussians (positive floats order like their bit patterns), reduced per workgroup; the
  // binning kernels fold the per-workgroup ranges themselves (a same-address atomic per wave cost more than this kernel)
  uint32_t zlo = 0x7f800000u, zhi = 0u;
  if (i < K) {
    radii[i] = 0;
    grad_[i] = 0;
  }
  if (i == 0) {      // the null record (alpha 0 everywhere): what the padding slots of a compositing trip point at - written
    float4* np = (float4*)(recs + K);      // whether or not Gaussian 0 itself is visible
    np[0] = make_float4(0.f, 0.f, -1.f, 0.f); np[1] = make_float4(-1.f, -__builtin_inff(), 0.f, 0.f); np[2] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  bool live = i < K;
  float mx = 0.f, my = 0.f, mz = 0.f, px = 0.f, py = 0.f;
  float3 pv = make_float3(0.f, 0.f, 0.f);
  Cov2D cv;
  float det_inv = 0.f;
  int r = 0;
  if (live) {
    mx = means[3 * i]; my = means[3 * i + 1]; mz = means[3 * i + 2];
    pv = xf43(k.view, mx, my, mz);
    live = pv.z > 0.2f;  // near-plane cull
  }
  if (live) {
    float4 ph = xf44(k.proj, mx, my, mz);
    float pw = 1.0f / (ph.w + 0.0000001f);
    float ndx = ph.x * pw, ndy = ph.y * pw;
    float c6[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) c6[a] = cov3D[6 * i + a];
    cv = compute_cov2d(k, mx, my, mz, c6);
    float det = cv.a * cv.c - cv.b * cv.b;
    live = det != 0.0f;
    if (live) {
      det_inv = 1.f / det;
      float mid = 0.5f * (cv.a + cv.c);
      float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
      r = (int)ceilf(3.f * sqrtf(lam));
      px = ((ndx + 1.0f) * k.W - 1.0f) * 0.5f; py = ((ndy + 1.0f) * k.H - 1.0f) * 0.5f;
      int x0, y0, x1, y1;
      get_rect(k, px, py, r, x0, y0, x1, y1, 0, k.gy);
      live = (x1 - x0) * (y1 - y0) != 0;
    }
  }
  if (live) {
    // colour
    uint32_t cl = 0;
    float col[3];
    if (colors) {
      col[0] = colors[3 * i]; col[1] = colors[3 * i + 1]; col[2] = colors[3 * i + 2];
    } else {
      float dx = mx - k.cam[0], dy = my - k.cam[1], dz = mz - k.cam[2];
      float inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
      float x = dx * inv, y = dy * inv, z = dz * inv;
      float sh[48];
      load_sh_row(shs, (size_t)i, k.M, sh);
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
    radii[i] = r;
    grad_[i] = r;
    xy[i] = make_float2(px, py);
    depth[i] = pv.z;
    conop[i] = make_float4(cv.c * det_inv, -cv.b * det_inv, cv.a * det_inv, opac[i]);
    rgb[3 * i] = col[0]; rgb[3 * i + 1] = col[1]; rgb[3 * i + 2] = col[2];
    {
      float4* rp = (float4*)(recs + i);
      const float op = opac[i];
      rp[0] = make_float4(px, py, -0.5f * NM_LOG2E * (cv.c * det_inv), NM_LOG2E * (cv.b * det_inv));
      rp[1] = make_float4(-0.5f * NM_LOG2E * (cv.a * det_inv), __log2f(op), col[0], col[1]);
      rp[2] = make_float4(col[2], op, 0.f, 0.f);
    }
    clamped[i] = cl;
    zlo = zhi = __float_as_uint(pv.z);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    zlo = min(zlo, (uint32_t)__shfl_xor((int)zlo, o, 64));
    zhi = max(zhi, (uint32_t)__shfl_xor((int)zhi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = zlo; s_hi[threadIdx.x >> 6] = zhi; }
  __syncthreads();
  if (threadIdx.x == 0)
    zrange[blockIdx.x] = make_uint2(min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])), max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])));
}

// depth slab of a view-space depth: monotone in z
__device__ __forceinline__ int slab_of(float z, uint32_t zmin_bits, uint32_t zmax_bits) {
  const float zmin = __uint_as_float(zmin_bits), zmax = __uint_as_float(zmax_bits);
  const float span = zmax - zmin;
  if (!(span > 0.f)) return 0;
  const int s = (int)((z - zmin) / span * (float)NM_NS);
  return min(NM_NS - 1, max(0, s));
}

#ifdef NM_PHASES
__device__ unsigned long long g_bc_phase[8];
extern "C" int nm_debug_bincount(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; return hipMemcpyToSymbol(HIP_SYMBOL(g_bc_phase), z, sizeof(z)) == hipSuccess ? 0 : -2; }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc_phase), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#define BC_T(i) { const long long t1_ = clock64(); bc_acc[i] += t1_ - bc_t0; bc_t0 = t1_; }
#else
#define BC_T(i)
#endif
// Count pass of the binning: one thread per Gaussian.  For every 64x64-pixel bin its 3-sigma tile rectangle (clipped to this
// rank's tile rows) overlaps, the exact conic test is run on the bin's 16 tiles; a pair with a non-empty tile mask takes a
// rank in its (bin, depth slab) cell (one integer atomic on the cell's counter) and is appended to the pair log (slots of a
// wave are reserved with ONE atomic), so that the fill pass needs neither atomics nor tile tests.
__global__ void __launch_bounds__(256) k_bin_count(RK k, int K, int nbx, const int* __restrict__ radii, const float2* __restrict__ xy,
                                                   const float* __restrict__ depth, const float4* __restrict__ conop,
                                                   const uint2* __restrict__ zrange, int nrange, uint32_t* __restrict__ pad,
                                                   PairLog* __restrict__ log, uint32_t* __restrict__ hdr, long long cap) {
  __shared__ uint32_t s_lo[4], s_hi[4];
  __shared__ int s_excl[4][64];
  __shared__ RowCull s_tc[4][64];
  __shared__ int4 s_geo[4][64], s_bin[4][64];
  const int lane = threadIdx.x & 63;
#ifdef NM_PHASES
  long long bc_t0 = clock64(), bc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  uint32_t zlo = 0x7f800000u, zhi = 0u;
  for (int q = threadIdx.x; q < nrange; q += 256) { const uint2 z = zrange[q]; zlo = min(zlo, z.x); zhi = max(zhi, z.y); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    zlo = min(zlo, (uint32_t)__shfl_xor((int)zlo, o, 64));
    zhi = max(zhi, (uint32_t)__shfl_xor((int)zhi, o, 64));
  }
  if (lane == 0) { s_lo[threadIdx.x >> 6] = zlo; s_hi[threadIdx.x >> 6] = zhi; }
  __syncthreads();
  zlo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
  zhi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
  BC_T(0)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0, slab = 0;
  RowCull tc = {0.f, 0.f, 1.f, 0.f, 1.f, -1.f, -1.f, 0.f};
  bool live = i < K && radii[i] > 0;
  if (live) {
    const float2 p = xy[i];
    get_rect(k, p.x, p.y, radii[i], x0, y0, x1, y1, k.ty0, k.ty1);
    live = (x1 - x0) * (y1 - y0) != 0;
    if (live) {
      tc = make_row_cull(p.x, p.y, conop[i]);
      slab = slab_of(depth[i], zlo, zhi);
    }
  }
  const int bx0 = x0 / NM_BT, bx1 = live ? (x1 - 1) / NM_BT : -1, by0 = y0 / NM_BT, by1 = live ? (y1 - 1) / NM_BT : -1;
  // ---- log slots: one per bin of the rectangle (an upper bound - bins no tile of which passes the conic test leave a dead
  //      entry), exclusive prefix over the lanes, ONE atomic for the wave
  const int nbw = bx1 - bx0 + 1;
  const int mine = live ? nbw * (by1 - by0 + 1) : 0;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
  const int wave_total = __shfl(incl, 63, 64);
  BC_T(1)
  // ONE reservation per workgroup: every returning atomic on this one word queues behind all the others (~12 ns each;
  // one per wave was 3 k of them per view)
  __shared__ uint32_t s_wtot[4], s_wgbase;
  if (lane == 0) s_wtot[threadIdx.x >> 6] = (uint32_t)wave_total;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t tot = s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
    s_wgbase = tot ? atomicAdd(&hdr[2], tot) : 0u;
  }
  __syncthreads();
  uint32_t base = s_wgbase;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += s_wtot[w];
  BC_T(2)
  // ---- the wave's pairs are spread over its lanes, one pair per lane and round: a thread that walked the bins of ITS
  //      Gaussian waited for one returning atomic per bin, one after the other (up to 25 round trips, and the wave waits
  //      for its longest lane: 146 us).  Now 64 atomics are in flight per round, the rounds are balanced, and the log is
  //      written in whole lines (it was 94 MB of HBM writes for 24 MB of entries).
  const int wv = threadIdx.x >> 6;
  s_excl[wv][lane] = incl - mine;
  s_tc[wv][lane] = tc;
  s_geo[wv][lane] = make_int4(x0, y0, x1, y1);
  s_bin[wv][lane] = make_int4(bx0, by0, nbw, slab);
  // Round 4 (phase counters, tools/exp_bincount_phases.py: owner search 23 % of a wave's cycles, waiting for the rank atomic 38 %):
  //  * the owner lane of every pair is tabulated once per wave (each lane writes its own index over its stretch of the wave's
  //    pair sequence) instead of being searched for per pair - six dependent LDS reads per pair and round;
  //  * a round's log entry is written one round LATER, behind the next round's atomic: the returning atomic's round trip (several
  //    us when a view's 1.15 M of them are in flight) then runs under the next round's tile tests instead of in front of them.
  __shared__ unsigned char s_owner[4][NM_BC_OWN];
  for (int q = 0; q < mine; ++q) {
    const int at_ = incl - mine + q;
    if (at_ < NM_BC_OWN) s_owner[wv][at_] = (unsigned char)lane;
  }
  __builtin_amdgcn_wave_barrier();
  BC_T(3)
  // (two entries in flight, alternating: a register COPY of an entry whose atomic has not returned would wait for it)
  auto round = [&](int p0, PairLog& e, long long& at) {
    const int pr = p0 + lane;
    at = -1;
    e.cell = 0xffffffffu; e.rank = 0u; e.id = 0u; e.mask = 0u;
    if (pr < wave_total) {
      int lo_;
      if (pr < NM_BC_OWN) {
        lo_ = (int)s_owner[wv][pr];
      } else {                                     // (a wave with more pairs than the table holds: search)
        lo_ = 0;
        int hi_ = 64;
#pragma unroll
        for (int it = 0; it < 6; ++it) { const int mid = (lo_ + hi_) >> 1; if (s_excl[wv][mid] <= pr) lo_ = mid; else hi_ = mid; }
      }
      const int q = pr - s_excl[wv][lo_];
      const int4 g = s_geo[wv][lo_], bn = s_bin[wv][lo_];
      const RowCull t = s_tc[wv][lo_];
      const int by = bn.y + q / bn.z, bx = bn.x + q % bn.z;
      const uint32_t m = bin_tile_mask(t, g, bx, by);
      e.id = (uint32_t)(blockIdx.x * blockDim.x + (wv << 6) + lo_); e.mask = m;
      if (m) {
        e.cell = (uint32_t)((by * nbx + bx) * NM_NS + bn.w);
        e.rank = atomicAdd(&pad[(size_t)e.cell * NM_PAD], 1u);
      }
      at = (long long)base + pr;
    }
  };
  PairLog ea, eb;
  long long ata = -1, atb = -1;
  for (int p0 = 0; p0 < wave_total; p0 += 128) {
    round(p0, ea, ata);
    BC_T(4)
    if (atb >= 0 && atb < cap) log[atb] = eb;      // the PREVIOUS round's entry: its atomic has had a round to return
    atb = -1;
    BC_T(5)
    if (p0 + 64 < wave_total) {
      round(p0 + 64, eb, atb);
      BC_T(4)
    }
    if (ata >= 0 && ata < cap) log[ata] = ea;
    BC_T(5)
  }
  if (atb >= 0 && atb < cap) log[atb] = eb;
  BC_T(6)
#ifdef NM_PHASES
  if (lane == 0) for (int q = 0; q < 7; ++q) atomicAdd(&g_bc_phase[q], (unsigned long long)bc_acc[q]);
  if (lane == 0) atomicAdd(&g_bc_phase[7], 1ull);
#endif
}

// padded counters -> compact array; per-bin totals (one workgroup per bin, one thread per depth slab)
__global__ void __launch_bounds__(NM_NS) k_bin_compact(int nbin, const uint32_t* __restrict__ pad, uint32_t* __restrict__ cnt,
                                                       uint32_t* __restrict__ bin_total, uint32_t* __restrict__ hdr) {
  __shared__ uint32_t s_sum[NM_NS / 64], s_big[NM_NS / 64];
  const int bin = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = bin * NM_NS + threadIdx.x;
  const uint32_t n = pad[(size_t)c * NM_PAD];
  cnt[c] = n;
  uint32_t sum = n, big = n;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sum += (uint32_t)__shfl_xor((int)sum, o, 64); big = max(big, (uint32_t)__shfl_xor((int)big, o, 64)); }
  if (lane == 0) { s_sum[wave] = sum; s_big[wave] = big; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t ts = 0, tb = 0;
    for (int w = 0; w < NM_NS / 64; ++w) { ts += s_sum[w]; tb = max(tb, s_big[w]); }
    bin_total[bin] = ts;
    if (tb) atomicMax(&hdr[6], tb);
  }
}

// cell offsets: bin offset + exclusive scan over the bin's depth slabs (one workgroup per bin).  The bin's own offset is the
// sum of the totals of the bins in front of it, which every workgroup adds up for itself (<= a few hundred words out of
// L2): cheaper than a scan kernel of its own between two 5-us launches.  The last bin's workgroup knows the grand total:
// it closes the offset array and sets the overflow flag (hdr[2] = pairs logged by the count pass, may exceed the capacity:
// the log and the lists then miss pairs).
__global__ void __launch_bounds__(NM_NS) k_cell_offsets(int nbin, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ bin_total,
                                                        uint32_t* __restrict__ off, uint32_t* __restrict__ hdr, long long cap) {
  __shared__ uint32_t s_w[NM_NS / 64], s_p[NM_NS / 64];
  const int bin = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t pre = 0u;
  for (int b = threadIdx.x; b < bin; b += NM_NS) pre += bin_total[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pre += (uint32_t)__shfl_xor((int)pre, o, 64);
  const int c = bin * NM_NS + threadIdx.x;
  const uint32_t n = cnt[c];
  uint32_t x = n;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64); if (lane >= o) x += y; }
  if (lane == 63) s_w[wave] = x;
  if (lane == 0) s_p[wave] = pre;
  __syncthreads();
  uint32_t before = 0u;
#pragma unroll
  for (int w = 0; w < NM_NS / 64; ++w) before += s_p[w];
  for (int w = 0; w < wave; ++w) before += s_w[w];
  off[c] = before + x - n;
  if (bin == nbin - 1 && threadIdx.x == NM_NS - 1) {
    off[c + 1] = before + x;
    hdr[3] = ((long long)hdr[2] > cap) ? 1u : 0u;
  }
}

// Fill pass: replay of the pair log - no atomics, no geometry: slot = cell offset + rank.  A thread takes its entries four at
// a time: all four log reads are requested together, then the eight dependent reads (cell offset, depth), then the stores - the
// rolled loop (one entry per trip: log -> offset / depth -> store) was two HBM round trips per entry, three to five entries per
// thread, one after the other (35 us per view for 28 MB of traffic).
__global__ void __launch_bounds__(256) k_bin_fill(const uint32_t* __restrict__ hdr, const PairLog* __restrict__ log,
                                                  const uint32_t* __restrict__ off, const float* __restrict__ depth,
                                                  unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, long long cap) {
  const long long n = min((long long)hdr[2], cap);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; e0 < n; e0 += 4 * stride) {
    PairLog q[4];
    uint32_t o[4], d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long e = e0 + u * stride;
      q[u].cell = 0xffffffffu;
      if (e < n) q[u] = log[e];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool liv = q[u].cell != 0xffffffffu;      // (dead entry: no tile of that bin passed the conic test)
      o[u] = liv ? off[q[u].cell] : 0u;
      d[u] = liv ? __float_as_uint(depth[q[u].id]) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (q[u].cell == 0xffffffffu) continue;
      const long long slot = (long long)o[u] + q[u].rank;
      if (slot < cap) {
        keys[slot] = ((unsigned long long)d[u] << 32) | (unsigned long long)q[u].id;
        vals[slot] = q[u].mask;
      }
    }
  }
}
// the same for the chunk binning's 8-byte entries (k_bin_count2): slot = coff[chunk, bin] + rank, key = gkeys[chunk * 256 + position]
__global__ void __launch_bounds__(256) k_bin_fill8(const uint32_t* __restrict__ hdr, const unsigned long long* __restrict__ log,
                                                   const uint32_t* __restrict__ coff, int nbin,
                                                   const unsigned long long* __restrict__ gkeys,
                                                   unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, long long cap) {
  const long long n = min((long long)hdr[2], cap);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; e0 < n; e0 += 4 * stride) {
    unsigned long long q[4], kk[4];
    uint32_t o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long e = e0 + u * stride;
      q[u] = log[min(e, n - 1)];            // (clamped: all four requests go out together, see the scans)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t chunk = (uint32_t)(q[u] >> 43), bin = (uint32_t)(q[u] >> 32) & 0x7ffu, pos = ((uint32_t)q[u] >> 16) & 0xffu;
      o[u] = coff[(size_t)chunk * nbin + bin];
      kk[u] = gkeys[(size_t)chunk * 256 + pos];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e0 + u * stride >= n) continue;
      const long long slot = (long long)o[u] + ((uint32_t)q[u] >> 24);
      if (slot < cap) {
        keys[slot] = kk[u];
        vals[slot] = (uint32_t)q[u] & 0xffffu;
      }
    }
  }
}

// Sort every cell by (depth, index), the tile masks riding along.  All compare-exchanges of this bitonic network put the
// smaller key at the lower index (each merge starts with a "flip" step i <-> block_end - 1 - i), so positions >= n act as
// +infinity padding and no power-of-two padding is stored.  SYNC: barrier among the NT cooperating threads.
template <int NT, class KeyP, class ValP, class Sync>
__device__ __forceinline__ void bitonic_sort_kv(KeyP key, ValP val, int n, int t, Sync sync) {
  int np = 1;
  while (np < n) np <<= 1;
  for (int kk = 2; kk <= np; kk <<= 1) {
    const int half = kk >> 1;
    for (int q = t; q < (np >> 1); q += NT) {
      const int blk = q / half, in = q - blk * half;
      const int i = blk * kk + in, j = blk * kk + kk - 1 - in;
      if (j < n) {
        const unsigned long long x = key[i], y = key[j];
        if (x > y) { key[i] = y; key[j] = x; const uint32_t u = val[i]; val[i] = val[j]; val[j] = u; }
      }
    }
    sync();
    for (int st = half >> 1; st > 0; st >>= 1) {
      for (int q = t; q < (np >> 1); q += NT) {
        const int i = 2 * st * (q / st) + (q % st), j = i + st;
        if (j < n) {
          const unsigned long long x = key[i], y = key[j];
          if (x > y) { key[i] = y; key[j] = x; const uint32_t u = val[i]; val[i] = val[j]; val[j] = u; }
        }
      }
      sync();
    }
  }
}

#define NM_CELL_WAVE 512   // cells up to this size are sorted by ONE wave (four cells per workgroup at a time, no block barriers)

// Rank sort of a cell held in LDS by NT cooperating threads: every element counts the keys smaller than its own (keys are
// unique: they end in the Gaussian index) - n broadcast reads per thread, no barrier, no data-dependent control flow - and
// goes straight to its final place in global memory.  O(n^2) compares, but for the few hundred pairs of a typical cell this
// is several times faster than a sorting network, whose ~log^2 n dependent LDS round trips dominate.
template <int NT, int PER>
__device__ __forceinline__ void rank_sort_to_global(const unsigned long long* s_key, const uint32_t* s_val, int n, int t,
                                                    unsigned long long* __restrict__ gkey, uint32_t* __restrict__ gval) {
  unsigned long long mine[PER];
  int rank[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) { const int i = t + NT * e; mine[e] = i < n ? s_key[i] : ~0ull; rank[e] = 0; }
  for (int j = 0; j < n; ++j) {
    const unsigned long long kj = s_key[j];
#pragma unroll
    for (int e = 0; e < PER; ++e) rank[e] += kj < mine[e] ? 1 : 0;
  }
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int i = t + NT * e;
    if (i < n) { gkey[rank[e]] = mine[e]; gval[rank[e]] = s_val[i]; }
  }
}

// elements per thread = ceil(n / NT), 1..8: the rank sort's cost is n x that, and a 330-pair cell is not a 512-pair one
template <int NT>
__device__ __forceinline__ void rank_sort_dispatch(const unsigned long long* s_key, const uint32_t* s_val, int n, int t,
                                                   unsigned long long* __restrict__ gkey, uint32_t* __restrict__ gval) {
  switch ((n + NT - 1) / NT) {
    case 1: rank_sort_to_global<NT, 1>(s_key, s_val, n, t, gkey, gval); break;
    case 2: rank_sort_to_global<NT, 2>(s_key, s_val, n, t, gkey, gval); break;
    case 3: rank_sort_to_global<NT, 3>(s_key, s_val, n, t, gkey, gval); break;
    case 4: rank_sort_to_global<NT, 4>(s_key, s_val, n, t, gkey, gval); break;
    case 5: rank_sort_to_global<NT, 5>(s_key, s_val, n, t, gkey, gval); break;
    case 6: rank_sort_to_global<NT, 6>(s_key, s_val, n, t, gkey, gval); break;
    case 7: rank_sort_to_global<NT, 7>(s_key, s_val, n, t, gkey, gval); break;
    default: rank_sort_to_global<NT, 8>(s_key, s_val, n, t, gkey, gval); break;
  }
}

// A workgroup takes four cells.  All four small: each wave sorts its own in a private LDS slice.  Otherwise the
// four are sorted one after the other by the whole workgroup (rank sort in LDS up to NM_CELL_LDS pairs; beyond that a bitonic
// network directly on global memory).
__global__ void __launch_bounds__(256) k_cell_sort(int ncell, const uint32_t* __restrict__ off, unsigned long long* __restrict__ keys,
                                                   uint32_t* __restrict__ vals, long long cap, const uint32_t* __restrict__ hdr) {
  if (hdr[3]) return;    // overflow: the lists are incomplete (some slots never written) and are not used
  __shared__ unsigned long long s_key[NM_CELL_LDS];
  __shared__ uint32_t s_val[NM_CELL_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // A group = four consecutive cells (neighbouring depth slabs of a bin are about equally big: the four waves finish together),
  // unless the view has cells the workgroup must sort as a whole (hdr[6] = the largest cell): four of those in one
  // workgroup are sorted one after the other (jd: four 536-pair neighbours were the kernel's critical path, 51 us), so the
  // group then takes its cells a quarter of the cell array apart
  const int ngroup = (ncell + 3) / 4;
  const bool spread = hdr[6] > NM_CELL_WAVE;
  for (int g0 = blockIdx.x; g0 < ngroup; g0 += gridDim.x) {
    long long lo[4];
    int n[4], biggest = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cq = spread ? g0 + q * ngroup : 4 * g0 + q;
      const int c = min(cq, ncell - 1);
      lo[q] = off[c];
      n[q] = (cq < ncell) ? (int)max(0ll, min((long long)off[c + 1], cap) - lo[q]) : 0;
      biggest = max(biggest, n[q]);
    }
    if (biggest < 2) continue;
    if (biggest <= NM_CELL_WAVE) {
      unsigned long long* wk = s_key + wave * NM_CELL_WAVE;
      uint32_t* wv = s_val + wave * NM_CELL_WAVE;
      int m = 0;
      long long l = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) if (q == wave) { m = n[q]; l = lo[q]; }
      if (m > 1) {
        for (int i = lane; i < m; i += 64) { wk[i] = keys[l + i]; wv[i] = vals[l + i]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rank_sort_dispatch<64>(wk, wv, m, lane, keys + l, vals + l);      // (wave-uniform m)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      auto bsync = [] { __syncthreads(); };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = n[q];
        if (m < 2) continue;        // workgroup-uniform
        __syncthreads();
        if (m <= NM_CELL_LDS) {
          for (int i = tid; i < m; i += 256) { s_key[i] = keys[lo[q] + i]; s_val[i] = vals[lo[q] + i]; }
          __syncthreads();
          rank_sort_dispatch<256>(s_key, s_val, m, tid, keys + lo[q], vals + lo[q]);
        } else {   // thousands of Gaussians of one bin in one depth slab: a sorting network on global memory
          bitonic_sort_kv<256>((volatile unsigned long long*)(keys + lo[q]), (volatile uint32_t*)(vals + lo[q]), m, tid, bsync);
        }
      }
      __syncthreads();
    }
  }
}


// ---------------------------------------------------------------- binning by depth-ordered chunks (round 5; default)
// The (bin, depth slab) cells above exist so that the per-bin depth sort decomposes into small pieces - at the price of one
// returning GLOBAL atomic per (Gaussian, bin) pair (1.15 M per metric view on 13 k cell counters), a 4 MB counter array to
// clear and read back, and a sort of 1.15 M pairs.  This path sorts the GAUSSIANS first (200 k keys instead of 1.15 M pairs):
//   k_slab_hist / k_slab_scatter   counting sort of the visible Gaussians into NM_GS depth slabs (keys = depth bits | id)
//   k_cell_sort (reused)           every slab sorted by (depth, id): the Gaussians are now in exact compositing order
//   k_bin_count2                   a workgroup takes 256 CONSECUTIVE Gaussians of that order.  Pairs are found as before (exact
//                                  tile masks), but ranked inside the workgroup: per bin a 256-bit membership mask in LDS (one
//                                  LDS atomic OR per pair), rank of a pair = set bits below its Gaussian - stable, no global
//                                  atomics, and the workgroup's pairs leave for the log already grouped by bin
//   k_col_sum / k_col_scan         per bin: exclusive scan of the chunks' counts (the (chunk, bin) segments of a bin follow each
//                                  other in depth order, and each is in depth order inside) + the bins' list offsets
//   k_bin_fill (reused)            slot = segment offset + rank: the bin lists come out SORTED - no sort of pairs at all.
// keys / vals / off[bin * NM_NS] / hdr[2], hdr[3] hold exactly what the cell path leaves there (same order: depth, then id).
#define NM_GS 1024            // depth slabs of the Gaussian sort
#define NM_GSUB 8             // counters per slab (by workgroup index): 200 k atomics on 1024 words queue up ~200 deep per word
#define NM_B2_MAXBIN 2048     // bins a view may have for this path (LDS: 32 B of mask per bin); larger images use the cells
#define NM_B2_STASH 384       // pairs per wave whose (bin, Gaussian, mask) are kept in LDS between the two walks
#define NM_B2_OWN 1024        // pairs per wave whose owner lane is tabulated (beyond: binary search)
__device__ __forceinline__ int gslab_of(float z, uint32_t zmin_bits, uint32_t zmax_bits) {
  const float zmin = __uint_as_float(zmin_bits), zmax = __uint_as_float(zmax_bits);
  const float span = zmax - zmin;
  if (!(span > 0.f)) return 0;
  const int sl = (int)((z - zmin) / span * (float)NM_GS);
  return min(NM_GS - 1, max(0, sl));
}
__device__ __forceinline__ void wg_depth_range(const uint2* __restrict__ zrange, int nrange, uint32_t* s_lo, uint32_t* s_hi,
                                               uint32_t& zlo, uint32_t& zhi) {
  const int lane = threadIdx.x & 63;
  zlo = 0x7f800000u; zhi = 0u;
  for (int q = threadIdx.x; q < nrange; q += 256) { const uint2 z = zrange[q]; zlo = min(zlo, z.x); zhi = max(zhi, z.y); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    zlo = min(zlo, (uint32_t)__shfl_xor((int)zlo, o, 64));
    zhi = max(zhi, (uint32_t)__shfl_xor((int)zhi, o, 64));
  }
  if (lane == 0) { s_lo[threadIdx.x >> 6] = zlo; s_hi[threadIdx.x >> 6] = zhi; }
  __syncthreads();
  zlo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
  zhi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
}
__device__ __forceinline__ bool gaussian_live(const RK& k, int i, int K, const int* __restrict__ radii, const float2* __restrict__ xy) {
  if (i >= K) return false;
  const int r = radii[i];
  if (r <= 0) return false;
  const float2 p = xy[i];
  int x0, y0, x1, y1;
  get_rect(k, p.x, p.y, r, x0, y0, x1, y1, k.ty0, k.ty1);
  return (x1 - x0) * (y1 - y0) != 0;
}
__global__ void __launch_bounds__(256) k_slab_hist(RK k, int K, const int* __restrict__ radii, const float2* __restrict__ xy,
                                                   const float* __restrict__ depth, const uint2* __restrict__ zrange, int nrange,
                                                   uint32_t* __restrict__ slab_cnt) {
  __shared__ uint32_t s_lo[4], s_hi[4];
  __shared__ uint32_t s_h[NM_GS];
  uint32_t zlo, zhi;
  for (int q = threadIdx.x; q < NM_GS; q += 256) s_h[q] = 0u;
  wg_depth_range(zrange, nrange, s_lo, s_hi, zlo, zhi);      // (its barrier also publishes the zeroed histogram)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (gaussian_live(k, i, K, radii, xy)) atomicAdd(&s_h[gslab_of(depth[i], zlo, zhi)], 1u);
  __syncthreads();
  for (int q = threadIdx.x; q < NM_GS; q += 256)
    if (s_h[q]) atomicAdd(&slab_cnt[q * NM_GSUB + (blockIdx.x & (NM_GSUB - 1))], s_h[q]);
}
// every workgroup scans the NM_GS counters for itself (cheaper than a launch in between); workgroup 0 publishes the offsets
__global__ void __launch_bounds__(256) k_slab_scatter(RK k, int K, const int* __restrict__ radii, const float2* __restrict__ xy,
                                                      const float* __restrict__ depth, const uint2* __restrict__ zrange, int nrange,
                                                      const uint32_t* __restrict__ slab_cnt, uint32_t* __restrict__ slab_cur,
                                                      uint32_t* __restrict__ slab_off, unsigned long long* __restrict__ gkeys,
                                                      uint32_t* __restrict__ hdr2) {
  __shared__ uint32_t s_lo[4], s_hi[4];
  __shared__ uint32_t s_off[NM_GS + 1], s_w[4], s_m[4];
  uint32_t zlo, zhi;
  wg_depth_range(zrange, nrange, s_lo, s_hi, zlo, zhi);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // thread t: slabs 4 t .. 4 t + 3, NM_GSUB counters each (32 consecutive words)
  uint32_t c[NM_GS / 256], sum = 0u, big = 0u;
  const int sub = blockIdx.x & (NM_GSUB - 1);
  uint32_t mine_before[NM_GS / 256];      // counters of this slab in front of this workgroup's own sub-counter
#pragma unroll
  for (int q = 0; q < NM_GS / 256; ++q) {
    const uint4* cp = reinterpret_cast<const uint4*>(slab_cnt + (size_t)(threadIdx.x * (NM_GS / 256) + q) * NM_GSUB);
    const uint4 a = cp[0], b = cp[1];
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t tot = 0u, bef = 0u;
#pragma unroll
    for (int u = 0; u < NM_GSUB; ++u) { bef += u < sub ? v[u] : 0u; tot += v[u]; }
    c[q] = tot; mine_before[q] = bef; sum += tot; big = max(big, tot);
  }
  uint32_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += y; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, o, 64));
  if (lane == 63) s_w[wave] = incl;
  if (lane == 0) s_m[wave] = big;
  __syncthreads();
  uint32_t before = incl - sum;
  for (int w = 0; w < wave; ++w) before += s_w[w];
  __shared__ uint32_t s_sub[NM_GS];       // where this workgroup's sub-counter starts inside the slab
#pragma unroll
  for (int q = 0; q < NM_GS / 256; ++q) {
    s_off[threadIdx.x * (NM_GS / 256) + q] = before;
    s_sub[threadIdx.x * (NM_GS / 256) + q] = before + mine_before[q];
    before += c[q];
  }
  if (threadIdx.x == 255) s_off[NM_GS] = before;
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int q = threadIdx.x; q <= NM_GS; q += 256) slab_off[q] = s_off[q];
    if (threadIdx.x == 0) { hdr2[0] = s_off[NM_GS]; hdr2[3] = 0u; hdr2[6] = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])); }
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (gaussian_live(k, i, K, radii, xy)) {
    const float z = depth[i];
    const int sl = gslab_of(z, zlo, zhi);
    const uint32_t pos = s_sub[sl] + atomicAdd(&slab_cur[sl * NM_GSUB + sub], 1u);
    gkeys[pos] = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)(uint32_t)i;
  }
}

// count pass over depth-ordered chunks: see the header of this section.  Dynamic LDS: nbin * 8 mask words | nbin + 1 prefix words.
__global__ void __launch_bounds__(256) k_bin_count2(RK k, int nbx, int nbin, const uint32_t* __restrict__ hdr2,
                                                    const unsigned long long* __restrict__ gkeys, const int* __restrict__ radii,
                                                    const float2* __restrict__ xy, const float4* __restrict__ conop,
                                                    uint32_t* __restrict__ hist, PairLog* __restrict__ log, uint32_t* __restrict__ hdr,
                                                    long long cap) {
  extern __shared__ uint32_t s_dyn[];
  uint32_t* s_mask = s_dyn;                    // [nbin][8]
  uint32_t* s_pref = s_dyn + (size_t)nbin * 8;   // [nbin + 1]
  __shared__ int s_excl[4][64];
  __shared__ RowCull s_tc[4][64];
  __shared__ int4 s_geo[4][64], s_bin[4][64];
  __shared__ uint32_t s_id[4][64];
  __shared__ unsigned char s_owner[4][NM_B2_OWN];
  __shared__ uint2 s_stash[4][NM_B2_STASH];
  __shared__ uint32_t s_wsum[4], s_base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int chunk = blockIdx.x;
  const uint32_t nvis = hdr2[0];
  for (int q = tid; q < nbin * 8; q += 256) s_mask[q] = 0u;
  const uint32_t at_ = (uint32_t)chunk * 256u + (uint32_t)tid;
  bool live = at_ < nvis;
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  uint32_t id = 0u;
  RowCull tc = {0.f, 0.f, 1.f, 0.f, 1.f, -1.f, -1.f, 0.f};
  if (live) {
    id = (uint32_t)gkeys[at_];
    const float2 p = xy[id];
    get_rect(k, p.x, p.y, radii[id], x0, y0, x1, y1, k.ty0, k.ty1);
    tc = make_row_cull(p.x, p.y, conop[id]);
  }
  const int bx0 = x0 / NM_BT, bx1 = live ? (x1 - 1) / NM_BT : -1, by0 = y0 / NM_BT, by1 = live ? (y1 - 1) / NM_BT : -1;
  const int nbw = bx1 - bx0 + 1;
  const int mine = live ? nbw * (by1 - by0 + 1) : 0;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
  const int wave_total = __shfl(incl, 63, 64);
  s_excl[wv][lane] = incl - mine;
  s_tc[wv][lane] = tc;
  s_geo[wv][lane] = make_int4(x0, y0, x1, y1);
  s_bin[wv][lane] = make_int4(bx0, by0, nbw, 0);
  s_id[wv][lane] = id;
  for (int q = 0; q < mine; ++q) {
    const int w_ = incl - mine + q;
    if (w_ < NM_B2_OWN) s_owner[wv][w_] = (unsigned char)lane;
  }
  __syncthreads();
  // one candidate pair (position pr of the wave's sequence): its owner lane, bin and exact tile mask
  auto pair_at = [&](int pr, int& owner, int& bin, uint32_t& m) {
    if (pr < NM_B2_OWN) owner = (int)s_owner[wv][pr];
    else {
      owner = 0;
      int hi_ = 64;
#pragma unroll
      for (int it = 0; it < 6; ++it) { const int mid = (owner + hi_) >> 1; if (s_excl[wv][mid] <= pr) owner = mid; else hi_ = mid; }
    }
    const int q = pr - s_excl[wv][owner];
    const int4 g = s_geo[wv][owner], bn = s_bin[wv][owner];
    const RowCull t = s_tc[wv][owner];
    const int by = bn.y + q / bn.z, bx = bn.x + q % bn.z;
    m = bin_tile_mask(t, g, bx, by);
    bin = by * nbx + bx;
  };
  // ---- walk 1: membership bits
  for (int p0 = 0; p0 < wave_total; p0 += 64) {
    const int pr = p0 + lane;
    if (pr < wave_total) {
      int owner, bin;
      uint32_t m;
      pair_at(pr, owner, bin, m);
      const int gl = 64 * wv + owner;
      if (m) atomicOr(&s_mask[bin * 8 + (gl >> 5)], 1u << (gl & 31));
      if (pr < NM_B2_STASH) s_stash[wv][pr] = make_uint2((uint32_t)bin | ((uint32_t)gl << 16), m);
    }
  }
  __syncthreads();
  // ---- per-bin counts of this chunk -> global row; exclusive prefix over the bins in LDS
  uint32_t run = 0u;
  const int per = (nbin + 255) / 256;
  const int b_lo = tid * per, b_hi = min(nbin, b_lo + per);
  for (int b = b_lo; b < b_hi; ++b) {
    uint32_t n = 0u;
#pragma unroll
    for (int w = 0; w < 8; ++w) n += (uint32_t)__popc(s_mask[b * 8 + w]);
    hist[(size_t)chunk * nbin + b] = n;
    s_pref[b] = run;          // (exclusive inside this thread's stretch; the stretch's offset is added below)
    run += n;
  }
  uint32_t inc2 = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc2, o, 64); if (lane >= o) inc2 += y; }
  if (lane == 63) s_wsum[wv] = inc2;
  __syncthreads();
  uint32_t before = inc2 - run;
  for (int w = 0; w < wv; ++w) before += s_wsum[w];
  for (int b = b_lo; b < b_hi; ++b) s_pref[b] += before;
  if (tid == 0) {
    const uint32_t tot = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    s_base = tot ? atomicAdd(&hdr[2], tot) : 0u;      // ONE reservation per workgroup
  }
  __syncthreads();
  const long long base = (long long)s_base;
  // ---- walk 2: ranks and log entries (grouped by bin, inside a bin in Gaussian = depth order)
  for (int p0 = 0; p0 < wave_total; p0 += 64) {
    const int pr = p0 + lane;
    if (pr < wave_total) {
      int bin, gl;
      uint32_t m;
      if (pr < NM_B2_STASH) {
        const uint2 e = s_stash[wv][pr];
        bin = (int)(e.x & 0xffffu); gl = (int)(e.x >> 16); m = e.y;
      } else {
        int owner;
        pair_at(pr, owner, bin, m);
        gl = 64 * wv + owner;
      }
      if (m) {
        uint32_t rank = 0u;
        const int wq = gl >> 5;
        for (int w = 0; w < wq; ++w) rank += (uint32_t)__popc(s_mask[bin * 8 + w]);
        rank += (uint32_t)__popc(s_mask[bin * 8 + wq] & ((1u << (gl & 31)) - 1u));
        const long long slot = base + (long long)s_pref[bin] + rank;
        if (slot < cap) {
          // 8-byte entries (k_bin_fill8): chunk | bin (11 bits) | rank (8) | position in the chunk (8) | tile mask (16).  The fill pass
          // finds (depth bits, id) at gkeys[chunk * 256 + position] - the chunk's 2 KB of the sorted list, read by the entries of one
          // chunk, which lie together in the log
          reinterpret_cast<unsigned long long*>(log)[slot] =
              ((unsigned long long)(uint32_t)chunk << 43) | ((unsigned long long)(uint32_t)bin << 32) |
              (unsigned long long)((rank << 24) | ((uint32_t)gl << 16) | (m & 0xffffu));
        }
      }
    }
  }
}
// per bin: pairs of all chunks
__global__ void __launch_bounds__(256) k_col_sum(int nbin, int nchunk, const uint32_t* __restrict__ hist, uint32_t* __restrict__ bin_total) {
  __shared__ uint32_t s_w[4];
  const int bin = blockIdx.x, lane = threadIdx.x & 63;
  uint32_t sum = 0u;
  for (int c = threadIdx.x; c < nchunk; c += 256) sum += hist[(size_t)c * nbin + bin];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += (uint32_t)__shfl_xor((int)sum, o, 64);
  if (lane == 0) s_w[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) bin_total[bin] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// per bin: offset of the bin's list (= the totals of the bins in front of it) + exclusive scan of its chunks' counts
__global__ void __launch_bounds__(256) k_col_scan(int nbin, int nchunk, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ bin_total,
                                                  uint32_t* __restrict__ coff, uint32_t* __restrict__ off, uint32_t* __restrict__ hdr,
                                                  long long cap) {
  __shared__ uint32_t s_p[4], s_w[4];
  const int bin = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t pre = 0u;
  for (int b = threadIdx.x; b < bin; b += 256) pre += bin_total[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pre += (uint32_t)__shfl_xor((int)pre, o, 64);
  const int per = (nchunk + 255) / 256;
  const int c_lo = threadIdx.x * per, c_hi = min(nchunk, c_lo + per);
  uint32_t run = 0u;
  for (int c = c_lo; c < c_hi; ++c) run += hist[(size_t)c * nbin + bin];
  uint32_t incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o, 64); if (lane >= o) incl += y; }
  if (lane == 63) s_w[wave] = incl;
  if (lane == 0) s_p[wave] = pre;
  __syncthreads();
  const uint32_t base = s_p[0] + s_p[1] + s_p[2] + s_p[3];
  uint32_t at = base + incl - run;
  for (int w = 0; w < wave; ++w) at += s_w[w];
  for (int c = c_lo; c < c_hi; ++c) { coff[(size_t)c * nbin + bin] = at; at += hist[(size_t)c * nbin + bin]; }
  if (threadIdx.x == 0) {
    off[(size_t)bin * NM_NS] = base;
    if (bin == nbin - 1) {
      const uint32_t total = base + s_w[0] + s_w[1] + s_w[2] + s_w[3];
      off[(size_t)nbin * NM_NS] = total;
      hdr[3] = ((long long)hdr[2] > cap) ? 1u : 0u;
      hdr[6] = 0u;
    }
  }
}

#ifdef NM_FIXDBG
__device__ unsigned long long* g_fixdbg;
extern "C" int nm_debug_fix_buffer(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_fixdbg), &p, sizeof(p)) == hipSuccess ? 0 : -2; }
#define FIXDBG(slot) do { if (threadIdx.x == 0 && g_fixdbg) g_fixdbg[(size_t)blockIdx.x * 8 + (slot)] = clock64(); } while (0)
#define FIXDBGV(slot, v) do { if (threadIdx.x == 0 && g_fixdbg) g_fixdbg[(size_t)blockIdx.x * 8 + (slot)] = (v); } while (0)
#define FIXACC(slot, t0) do { if (threadIdx.x == 0 && g_fixdbg) { const long long t1_ = clock64(); g_fixdbg[(size_t)blockIdx.x * 8 + (slot)] += t1_ - (t0); (t0) = t1_; } } while (0)
#define FIXCNT(slot, v) do { if (threadIdx.x == 0 && g_fixdbg) g_fixdbg[(size_t)blockIdx.x * 8 + (slot)] += (v); } while (0)
#else
#define FIXACC(slot, t0)
#define FIXCNT(slot, v)
#define FIXDBG(slot)
#define FIXDBGV(slot, v)
#endif
#define NM_SCAN 4096   // candidates a tile examines per round (16 per thread: four 16-byte loads of their tile masks)
struct CompLds {
  uint32_t hit[NM_SCAN];      // 1-based list positions of the candidates that touch this tile, in list order
  int wcnt[4];
  uint32_t wreach[4];
};
struct Pix {
  float T, C0, C1, C2;
  uint32_t last;      // last contributor, 1-based position in the bin's list
  bool done;          // stopped (T would fall below 1e-4) or not taking part
};
// front-to-back composite of one 16x16 tile (upstream renderCUDA forward) over entries [a, b) of the depth-sorted list of
// the tile's bin (the bin's list starts at lo; positions are counted from there)
__device__ __forceinline__ uint32_t composite_range(CompLds& L, long long lo, long long a, long long b, uint32_t bit,
                                                const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                const GRec* __restrict__ recs, uint32_t null_off, float fxp, float fyp, Pix& p,
                                                uint32_t seg = 0u, uint32_t ck_s = 0u, uint32_t ck_n = 0u,
                                                float4* __restrict__ ck = nullptr, uint32_t* __restrict__ ck_pos = nullptr) {
  // ck != NULL: checkpoints for the reverse sweep.  Whenever the walk has passed the nominal start ck_pos[s] of segment s
  // (looked at after every batch of hits and at the end of every round - never inside the compositing loop), the pixels'
  // (C, T) go to ck[s] and the position the walk stands at to ck_pos[s]: segment s starts exactly there.  s = ck_s .. ck_n-1.
  // Returns the list position (counted from lo) the walk had examined when it ended - every pixel stopped, or b reached.
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t reached = (uint32_t)(a - lo);
  // every wave writes the checkpoints of its own pixels (and the same boundary position: a benign race) - the waves of a
  // tile meet only between rounds
  auto checkpoint = [&](uint32_t at) {
    while (ck_s < ck_n && at >= ck_pos[ck_s]) {     // (the plan's boundary: q * seg, or the tile's own step in a hinted plan)
      ck[(size_t)ck_s * NM_TPB + tid] = make_float4(p.C0, p.C1, p.C2, p.T);
      if (lane == 0) ck_pos[ck_s] = at;
      ++ck_s;
    }
  };
  // A pixel that has stopped (or lies outside the image) is moved to x = +inf: every Gaussian's exponent is -inf there,
  // its alpha 0, and the loop body needs no `done` mask at all.
  const float kInf = __builtin_inff();
  float fx = p.done ? kInf : fxp;
  uint32_t wreach = reached;        // where this wave stands
  if (lane == 0) L.wreach[wave] = wreach;
#ifdef NM_FIXDBG
  long long tph = clock64();
#endif
  for (long long base = a & ~3ll; base < b; base += NM_SCAN) {     // rounds start 16-byte aligned; positions < a are masked out
    if (__syncthreads_count(!(fx < kInf)) == NM_TPB) break;
    FIXACC(5, tph);
    // ---- NM_SCAN candidates: which of them touch this tile (bit of their tile mask)?
    const long long c = base + 16 * tid;
    uint32_t m16 = 0;
    // All four loads are requested before any of them is looked at (round 5): written as `if (q0 < b) { load; test }` four
    // times, the compiler kept each load inside its own branch - four HBM / L2 round trips in a row per round and thread.  A
    // group past the end reads the range's last group instead (in bounds: the array is padded to 256 bytes) and is masked out.
    uint4 v[4];
    const long long qlast = (b - 1) & ~3ll;
#pragma unroll
    for (int v4 = 0; v4 < 4; ++v4) v[v4] = *(const uint4*)(vals + min(c + 4 * v4, qlast));
#pragma unroll
    for (int v4 = 0; v4 < 4; ++v4) {
      const long long q0 = c + 4 * v4;
      const uint32_t b4 = ((v[v4].x & bit) ? 1u : 0u) | ((v[v4].y & bit) ? 2u : 0u) | ((v[v4].z & bit) ? 4u : 0u) | ((v[v4].w & bit) ? 8u : 0u);
      uint32_t ok = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) ok |= (q0 + q >= a && q0 + q < b) ? (1u << q) : 0u;
      m16 |= (b4 & ok) << (4 * v4);
    }
    const int mine = __popc(m16);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
    if (lane == 63) L.wcnt[wave] = incl;
    __syncthreads();
    int before = 0, nh = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int n = L.wcnt[w]; before += w < wave ? n : 0; nh += n; }
    int slot = before + incl - mine;
    for (uint32_t mm = m16; mm; mm &= mm - 1u) L.hit[slot++] = (uint32_t)(c + (__ffs((int)mm) - 1) - lo) + 1u;
    __syncthreads();
    FIXACC(2, tph);
    FIXCNT(6, (unsigned long long)nh);
    // ---- composite the survivors, in list (= depth) order, 256 at a time - every wave by itself, no barrier until the round
    // is over.  No staging either: a wave keeps the batch's Gaussians (byte offsets of their records) in four registers
    // (lane l: hits l, l + 64, ...), pulls hit j's out with v_readlane and fetches the record with scalar loads into one of
    // two alternating register sets, NM_G Gaussians ahead of the arithmetic.  Slots past the batch's end point at the null
    // record (opacity 0), so a trip never needs a bound check.
    bool live = __ballot(fx < kInf) != 0ull;
    for (int h0 = 0; h0 < nh && live; h0 += NM_TPB) {
      const int nb = min(NM_TPB, nh - h0);
      uint32_t idr[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)        // all four keys requested at clamped slots ...
        idr[q] = (uint32_t)keys[lo + L.hit[h0 + min(64 * q + lane, nb - 1)] - 1];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // ... and pinned as unconditional (the empty asm): `t < nb ? load : null` - however it is spelled - is sunk into a branch
        // per load by the compiler, i.e. four round trips in a row per batch of 256 hits
        asm volatile("" : "+v"(idr[q]));
        idr[q] = 64 * q + lane < nb ? idr[q] * (uint32_t)sizeof(GRec) : null_off;
      }
      FIXACC(3, tph);
      int lastj = -1;
      // one Gaussian against this lane's pixel (upstream renderCUDA forward; `valid` / `upd` are lane masks)
      auto one = [&](const float4& g0, const float4& g1, float bl, int j) {
        const float dx = g0.x - fx, dy = g0.y - fyp;
        const float e2 = dx * (g0.z * dx + g0.w * dy) + (g1.x * dy) * dy;     // log2 of the Gaussian's falloff
        const float alpha = fminf(0.99f, __builtin_amdgcn_exp2f(e2 + g1.y));    // opacity folded into the exponent
        const bool valid = !(e2 > 0.f) && !(alpha < 1.0f / 255.0f);
        if (__ballot(valid) == 0ull) return;
        const float test_T = p.T * (1.f - alpha);
        const bool upd = valid && !(test_T < 0.0001f);
        const float w = upd ? alpha * p.T : 0.f;
        p.C0 += g1.z * w; p.C1 += g1.w * w; p.C2 += bl * w;
        p.T = upd ? test_T : p.T;
        lastj = upd ? j : lastj;
        fx = (valid && !upd) ? kInf : fx;            // T would fall below 1e-4: the pixel stops
      };
      struct Set { float4 a[NM_G], b[NM_G]; float c[NM_G]; };
      auto fetch = [&](Set& g, uint32_t offs, int j0) {
#pragma unroll
        for (int u = 0; u < NM_G; ++u) {
          const char* rp = (const char*)recs + (uint32_t)__builtin_amdgcn_readlane((int)offs, j0 + u);
          g.a[u] = *(const float4*)rp; g.b[u] = *(const float4*)(rp + 16); g.c[u] = *(const float*)(rp + 32);
        }
      };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nq = min(64, nb - 64 * q);          // (wave-uniform)
        if (nq <= 0) break;
        if (live) {
          Set A, B;
          fetch(A, idr[q], 0);
          for (int j0 = 0; j0 < nq; j0 += 2 * NM_G) {
            fetch(B, idr[q], j0 + NM_G);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NM_G; ++u) one(A.a[u], A.b[u], A.c[u], 64 * q + j0 + u);
            fetch(A, idr[q], (j0 + 2 * NM_G) & 63);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NM_G; ++u) one(B.a[u], B.b[u], B.c[u], 64 * q + j0 + NM_G + u);
            if (__ballot(fx < kInf) == 0ull) { live = false; break; }
          }
        }
        // checkpoints can be taken every 64 hits (the reverse sweep's segments are what lies between two of them).  Every wave
        // that entered the batch visits ALL of its group ends and its end, in order, whether or not its pixels have stopped on
        // the way (their state is final then): the waves must agree on the position a boundary is moved to, and a boundary
        // inside the batch that every wave dies in still has to be taken
        if (ck && 64 * q + nq < nb) checkpoint(L.hit[h0 + 64 * q + nq - 1]);
      }
      if (lastj >= 0) p.last = L.hit[h0 + lastj];
      wreach = L.hit[h0 + nb - 1];
      FIXACC(4, tph);
      if (ck) checkpoint(wreach);
    }
    if (live) {
      wreach = (uint32_t)(min(base + NM_SCAN, b) - lo);
      if (ck) checkpoint(wreach);
    }
    if (lane == 0) L.wreach[wave] = wreach;
  }
  __syncthreads();
  reached = max(max(L.wreach[0], L.wreach[1]), max(L.wreach[2], L.wreach[3]));       // the wave that went furthest
  __syncthreads();       // (L.wreach is written again by the caller's next range)
  p.done = !(fx < kInf);
  return reached;
}
__device__ __forceinline__ void write_pixel(const RK& k, int px, int py, const Pix& p, float* __restrict__ final_T,
                                            uint32_t* __restrict__ n_contrib, float* __restrict__ out) {
  size_t pix = (size_t)py * k.W + px, hw = (size_t)k.H * k.W;
  final_T[pix] = p.T;
  n_contrib[pix] = p.last;
  out[pix] = p.C0 + p.T * k.bg[0];
  out[hw + pix] = p.C1 + p.T * k.bg[1];
  out[2 * hw + pix] = p.C2 + p.T * k.bg[2];
}

__device__ __forceinline__ void render_whole(CompLds& L, const RK& k, int nbx, int tile_x, int tile_y, const uint32_t* __restrict__ off,
                                             const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                             long long cap, const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ tile_rec,
                                             const GRec* __restrict__ recs, float* __restrict__ final_T,
                                             uint32_t* __restrict__ n_contrib, float* __restrict__ out,
                                             uint32_t* __restrict__ hint) {
  if (tile_rec[tile_y * k.gx + tile_x] != 0xFFFFFFFFu) return;      // a candidate of the split compositing (render_seg)
  const int tid = threadIdx.x;
  const int px = tile_x * NM_TILE + (tid & 15), py = tile_y * NM_TILE + (tid >> 4);
  const bool inside = px < k.W && py < k.H;
  const int bin = (tile_y / NM_BT) * nbx + tile_x / NM_BT;
  const uint32_t bit = 1u << ((tile_y % NM_BT) * NM_BT + tile_x % NM_BT);
  const long long lo = off[bin * NM_NS];
  // capacity overflow (hdr[3]): slots of the lists were never written - render the background only, the caller re-runs
  const long long hi = hdr[3] ? lo : min((long long)off[(bin + 1) * NM_NS], cap);
  Pix p = {1.f, 0.f, 0.f, 0.f, 0u, !inside};
  const uint32_t reached = composite_range(L, lo, lo, hi, bit, keys, vals, recs, (uint32_t)k.K * (uint32_t)sizeof(GRec), (float)px, (float)py, p);
  if (inside) write_pixel(k, px, py, p, final_T, n_contrib, out);
  if (hint && tid == 0) hint[tile_y * k.gx + tile_x] = reached;
}

// ---- split compositing.  A view whose Gaussians all fall into a few dozen tiles (the single-object scenes: 36-240 busy
// tiles with 10-70 k overlapping Gaussians each) keeps a few waves per busy tile busy with one long sequential loop
// each while most of the chip idles, and the loop is longest where it cannot stop early: in tiles on the object's
// silhouette, whose barely covered pixels never saturate and walk the whole list.  Compositing is associative - a
// stretch of the list composited from T = 1 gives (C_s, T_s), and the pixel is C_0 + T_0 C_1 + T_0 T_1 C_2 ... - so
// such a tile's list is cut into segments, one workgroup each:
//   k_split_plan      candidate tiles (list longer than a segment) of a view with < busy_limit non-empty tiles; on the
//                     device, no host read-back.  Views that fill the chip are never split.
//   k_render_seg 0    a candidate tile's first segment.  If every pixel is at T <= NM_SPLIT_TAU afterwards (interior of an
//                     opaque object: it will stop soon) the same workgroup simply walks on and finishes the tile;
//                     otherwise the tile is split (tile_mode = 1) and
//   k_render_seg 1    composites its other segments from T = 1, in parallel;
//   k_render_fix      with the transmittance in front of a segment known, a pixel passes through it, has stopped before
//                     it, or stops inside it - then the segment is walked again for those pixels from the true T, so the
//                     reference's termination rule (stop when T would fall below 1e-4, forward.cu) is kept exactly;
//                     the tile's last workgroup to finish sums the records.
// The reverse sweep walks the segments of a split tile in parallel too (k_render_bwd_seg).
// The render's status words for the host, as three 64-bit values behind one another (one 24-byte copy instead of three
// 4-byte ones - each was a blit kernel of its own): pairs binned | overflow flag (NM_RASTER_DEBUG: high half = the largest cell) |
// work items the plan asked for.  Written by the last thread that knows them all.
// host_status (optional): the caller's pinned words as the device sees them - written from here, so that no device-to-host
// copy (a blit of its own, ~5 us plus its dependency bubbles) sits in the view's stream between the compositing and the loss
__device__ __forceinline__ void plan_status(uint32_t* __restrict__ hdr, int dbg, unsigned long long* __restrict__ host_status = nullptr,
                                            int host_words = 0) {
  hdr[16] = hdr[2]; hdr[17] = 0u; hdr[18] = hdr[3]; hdr[19] = dbg ? hdr[6] : 0u; hdr[20] = hdr[12]; hdr[21] = 0u;
  if (host_status) {
    host_status[0] = (unsigned long long)hdr[2];
    host_status[1] = (unsigned long long)hdr[3] | ((unsigned long long)(dbg ? hdr[6] : 0u) << 32);
    if (host_words > 2) host_status[2] = (unsigned long long)hdr[12];
  }
}
__global__ void __launch_bounds__(1024) k_split_plan(RK k, int nbx, uint32_t busy_limit, uint32_t min_seg, unsigned long long fwd_max,
                                                     const uint32_t* __restrict__ off, long long cap,
                                                     uint32_t* __restrict__ hdr, uint32_t* __restrict__ tile_rec,
                                                     uint32_t* __restrict__ tile_ns, uint32_t* __restrict__ tile_cnt,
                                                     uint32_t* __restrict__ tile_mode, uint2* __restrict__ work,
                                                     uint32_t* __restrict__ seg_pos, const uint32_t* __restrict__ hint,
                                                     uint32_t fwd_len, uint32_t hint_seg, int dbg,
                                                     unsigned long long* __restrict__ host_status, int host_words) {
  __shared__ unsigned long long s_total, s_lists;
  __shared__ uint32_t s_busy, s_base, s_used, s_scan[1024];
  const int tid = threadIdx.x, rows = k.ty1 - k.ty0, ntile = k.gx * rows;
  if (tid == 0) { s_total = 0ull; s_lists = 0ull; s_busy = 0u; s_base = 0u; s_used = 0u; }
  __syncthreads();
  auto list_len = [&](int i) -> uint32_t {
    const int tx = i % k.gx, ty = i / k.gx + k.ty0;
    const int bin = (ty / NM_BT) * nbx + tx / NM_BT;
    const long long lo = off[bin * NM_NS], hi = min((long long)off[(bin + 1) * NM_NS], cap);
    return hi > lo ? (uint32_t)(hi - lo) : 0u;
  };
  // ---- hinted plan.  hint[t] = list position the previous forward walk of tile t with this camera had reached when it ended
  // (every pixel saturated, or the end of the list for a tile with a pixel that never saturates); 0 = nothing known.  The
  // scene moves little between two renders of a camera, so the walk is planned to be about as long again: the tile is cut
  // into equal segments of ~seg entries up to 1.25 x the hint.  The reverse sweep walks all segments in parallel, whatever
  // the number of busy tiles.  The forward pass composites a tile's segments in parallel from T = 1 only where the walk is
  // long (> fwd_len entries: the view's critical path) - k_render_fix then restores the exact termination - and walks the
  // other tiles front to back as ever, leaving the (C, T) checkpoints in front of the segments on the way (no work is done
  // twice there).  Behind the planned stretch (record too short) the walk simply goes on.  A wrong hint costs time, never
  // accuracy.  Every thread owns NM_PLAN_PER consecutive tiles (all their loads in flight at once).
  if (hint && !hdr[3] && ntile <= 1024 * NM_PLAN_PER) {
    uint32_t ln[NM_PLAN_PER], ll[NM_PLAN_PER];
    int tt[NM_PLAN_PER];                 // tile index ty * gx + tx of the thread's tiles (-1: past the end)
    unsigned long long tot = 0ull; uint32_t any = 0u;
    {
      // thread t owns tiles t, t + 1024, ...: this kernel is ONE workgroup on one CU, and with eight consecutive tiles per
      // thread every load and store instruction of a wave touched 64 different cache lines (32 us; 9 us this way)
      // (all 3 x NM_PLAN_PER loads UNCONDITIONAL, at a clamped tile index, and issued before the first is used: under
      //  `if (i < ntile)` each tile's three loads sat in a branch region of their own and the compiler waited for them
      //  before it issued the next tile's - NM_PLAN_PER round trips in a row on the one CU this launch has)
      uint32_t hh[NM_PLAN_PER], lo_[NM_PLAN_PER], hi_[NM_PLAN_PER];
#pragma unroll
      for (int u = 0; u < NM_PLAN_PER; ++u) {
        const int i = u * 1024 + tid, ic = min(i, ntile - 1);
        const int ty = ic / k.gx, tx = ic - ty * k.gx, tyy = ty + k.ty0;
        const int t = tyy * k.gx + tx;
        tt[u] = i < ntile ? t : -1;
        const int bin = (tyy / NM_BT) * nbx + tx / NM_BT;
        hh[u] = hint[t]; lo_[u] = off[bin * NM_NS]; hi_[u] = off[(bin + 1) * NM_NS];
      }
#pragma unroll
      for (int u = 0; u < NM_PLAN_PER; ++u) {
        const long long lo = lo_[u], hi = min((long long)hi_[u], cap);
        const uint32_t h = tt[u] >= 0 ? hh[u] : 0u;
        ln[u] = (tt[u] >= 0 && hi > lo) ? (uint32_t)(hi - lo) : 0u;
        ll[u] = h ? min(ln[u], h + h / 4u + 64u) : 0u;
        tot += ll[u]; any += ll[u] ? 1u : 0u;
      }
    }
    {
      unsigned long long lt = 0ull;       // (statistics: list entries a tile-by-tile walk of the whole lists would examine)
#pragma unroll
      for (int u = 0; u < NM_PLAN_PER; ++u) lt += ln[u];
      // one LDS atomic per wave, not per thread (1024 64-bit LDS atomics in a row were most of this kernel's time)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        lt += (unsigned long long)__shfl_xor((long long)lt, o, 64);
        tot += (unsigned long long)__shfl_xor((long long)tot, o, 64);
        any += (uint32_t)__shfl_xor((int)any, o, 64);
      }
      if ((tid & 63) == 0) {
        if (lt) atomicAdd(&s_lists, lt);
        if (any) { atomicAdd(&s_total, tot); atomicAdd(&s_busy, any); }
      }
    }
    __syncthreads();
    if (s_busy > 0u) {
      uint32_t seg = (uint32_t)((s_total + NM_HINT_WGS - 1) / NM_HINT_WGS);      // (independent of the capacity: see hdr[12])
      seg = max(seg, hint_seg);
      seg = (seg + 15u) & ~15u;
      auto segments = [&](uint32_t l) -> uint32_t { return l > seg + seg / 2u ? (l + seg - 1) / seg : 0u; };
      uint32_t mine = 0u;
#pragma unroll
      for (int u = 0; u < NM_PLAN_PER; ++u) { const uint32_t ns = segments(ll[u]); mine += ns ? ns + 1u : 0u; }
      uint32_t incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o, 64); if ((tid & 63) >= o) incl += y; }
      if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
      __syncthreads();
      uint32_t rec = incl - mine;
      for (int w = 0; w < (tid >> 6); ++w) rec += s_scan[w];
      if (tid == 1023) s_base = rec + mine;
#pragma unroll
      for (int u = 0; u < NM_PLAN_PER; ++u) {
        if (tt[u] < 0) continue;
        const uint32_t l = ll[u], ns = segments(l), items = ns ? ns + 1u : 0u;
        const int t = tt[u];
        const bool ok = ns > 0u && rec + items <= (uint32_t)k.items;
        tile_rec[t] = ok ? rec : 0xFFFFFFFFu;
        tile_ns[t] = ok ? ns : 0u;
        tile_cnt[t] = ok ? (((l + ns - 1) / ns + 15u) & ~15u) : 0u;      // the tile's segment length (k_hint_fill)
        tile_mode[t] = ok && l > fwd_len ? 1u : 0u;
        if (ok) atomicMax(&s_used, rec + items);       // work items beyond the last assigned one do not exist (hdr[8])
        rec += items;
      }
      __syncthreads();
      if (tid == 0) { hdr[8] = s_used; hdr[9] = seg; hdr[10] = 1u; hdr[11] = 1u; hdr[12] = s_base; }
      if (tid == 0) { const unsigned long long lt = s_lists; hdr[14] = (uint32_t)lt; hdr[15] = (uint32_t)(lt >> 32); plan_status(hdr, dbg, host_status, host_words); }
      return;
    }
    // nothing known yet (the first render with this camera, or nothing in view): every tile is walked whole and leaves its
    // record - the next render is planned.  (A hinted call therefore never needs the second compositing stage: the host
    // does not launch it.)
#pragma unroll
    for (int u = 0; u < NM_PLAN_PER; ++u) {
      if (tt[u] < 0) continue;
      tile_rec[tt[u]] = 0xFFFFFFFFu; tile_ns[tt[u]] = 0u; tile_cnt[tt[u]] = 0u; tile_mode[tt[u]] = 0u;
    }
    if (tid == 0) {
      const unsigned long long lt = s_lists;
      hdr[8] = 0u; hdr[9] = (hint_seg + 15u) & ~15u; hdr[10] = 1u; hdr[11] = 1u; hdr[12] = 0u; hdr[14] = (uint32_t)lt; hdr[15] = (uint32_t)(lt >> 32);
      plan_status(hdr, dbg, host_status, host_words);
    }
    return;
  }
  unsigned long long tot = 0ull; uint32_t busy = 0u;
  if (!hdr[3]) {         // all tiles of a bin share its list: one thread per bin
    const int nby = (k.gy + NM_BT - 1) / NM_BT;
    for (int bin = tid; bin < nbx * nby; bin += 1024) {
      const int bx = bin % nbx, by = bin / nbx;
      const int nx = min(k.gx, (bx + 1) * NM_BT) - bx * NM_BT;
      const int ny = min(k.ty1, (by + 1) * NM_BT) - max(k.ty0, by * NM_BT);
      if (nx <= 0 || ny <= 0) continue;
      const long long lo = off[bin * NM_NS], hi = min((long long)off[(bin + 1) * NM_NS], cap);
      if (hi > lo) { tot += (unsigned long long)(hi - lo) * (unsigned)(nx * ny); busy += (uint32_t)(nx * ny); }
    }
  }
  if (busy) { atomicAdd(&s_total, tot); atomicAdd(&s_busy, busy); }
  __syncthreads();
  const bool split = s_busy > 0u && s_busy < busy_limit;
  uint32_t seg = (uint32_t)((s_total + NM_SPLIT_WGS - 1) / NM_SPLIT_WGS);
  seg = max(seg, min_seg);
  seg = (seg + 15u) & ~15u;
  if (!split) {                                  // the usual case (a view that fills the chip): no candidates
    for (int i = tid; i < ntile; i += 1024) {
      const int t = (i / k.gx + k.ty0) * k.gx + i % k.gx;
      tile_rec[t] = 0xFFFFFFFFu; tile_ns[t] = 0u; tile_cnt[t] = 0u; tile_mode[t] = 0u;
    }
    if (tid == 0) { hdr[8] = 0u; hdr[9] = seg; hdr[10] = 0u; hdr[14] = (uint32_t)s_total; hdr[15] = (uint32_t)(s_total >> 32); plan_status(hdr, dbg, host_status, host_words); }
    return;
  }
  for (int i0 = 0; i0 < ntile; i0 += 1024) {     // segments per tile, exclusive prefix in tile order
    const int i = i0 + tid;
    uint32_t ns = 0u;
    if (split && i < ntile) { const uint32_t n = list_len(i); ns = n > seg ? (n + seg - 1) / seg : 0u; }
    const uint32_t items = ns ? ns + 1u : 0u;      // one more record for the pixel's final (C, T)
    s_scan[tid] = items;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const uint32_t v = tid >= o ? s_scan[tid - o] : 0u;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const uint32_t rec = s_base + s_scan[tid] - items;
    if (i < ntile) {
      const int t = (i / k.gx + k.ty0) * k.gx + i % k.gx;
      const bool ok = ns > 0u && rec + items <= (uint32_t)k.items;    // (holds by construction of seg; kept as a guard)
      tile_rec[t] = ok ? rec : 0xFFFFFFFFu;
      tile_ns[t] = ok ? ns : 0u;
      tile_cnt[t] = 0u;
      tile_mode[t] = 0u;
      if (ok) {
        atomicMax(&s_used, rec + items);
        const uint32_t n = list_len(i);
        for (uint32_t q = 0; q <= ns; ++q) { work[rec + q] = make_uint2((uint32_t)t, q); seg_pos[rec + q] = min(q * seg, n); }
      }
    }
    __syncthreads();
    if (tid == 1023) s_base += s_scan[1023];
    __syncthreads();
  }
  if (tid == 0) {
    hdr[8] = s_used; hdr[9] = seg; hdr[10] = s_total <= fwd_max ? 1u : 0u; hdr[12] = s_base; hdr[14] = (uint32_t)s_total; hdr[15] = (uint32_t)(s_total >> 32);
    plan_status(hdr, dbg, host_status, host_words);
  }
}

// work items of a hinted plan: one wave per tile writes its (tile, segment) pairs and segment boundaries
__global__ void __launch_bounds__(256) k_hint_fill(RK k, int nbx, const uint32_t* __restrict__ off, long long cap,
                                                   const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ tile_rec,
                                                   const uint32_t* __restrict__ tile_ns, const uint32_t* __restrict__ tile_cnt,
                                                   uint2* __restrict__ work, uint32_t* __restrict__ seg_pos) {
  if (!hdr[11]) return;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, ntile = k.gx * (k.ty1 - k.ty0);
  if (i >= ntile) return;
  const int tx = i % k.gx, ty = i / k.gx + k.ty0, t = ty * k.gx + tx;
  const uint32_t ns = tile_ns[t];
  if (!ns) return;
  const uint32_t rec = tile_rec[t], step = tile_cnt[t];
  const int bin = (ty / NM_BT) * nbx + tx / NM_BT;
  const long long lo = off[bin * NM_NS], hi = min((long long)off[(bin + 1) * NM_NS], cap);
  const uint32_t n = hi > lo ? (uint32_t)(hi - lo) : 0u;
  for (uint32_t q = lane; q <= ns; q += 64) { work[rec + q] = make_uint2((uint32_t)t, q); seg_pos[rec + q] = min(q * step, n); }
}

// pass 1 (two launches: stage 0 = first segments, which decide; stage 1 = the other segments of the split tiles)
__device__ __forceinline__ void render_seg(CompLds& L, const RK& k, int nbx, int stage, uint32_t w, uint32_t* __restrict__ tile_mode,
                                           const uint32_t* __restrict__ tile_ns, const uint32_t* __restrict__ off,
                                           const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                           long long cap, const uint32_t* __restrict__ hdr, const uint2* __restrict__ work,
                                           float4* __restrict__ seg_raw, uint32_t* __restrict__ seg_last,
                                           float4* __restrict__ seg_ct, uint32_t* __restrict__ seg_pos,
                                           const GRec* __restrict__ recs, float* __restrict__ final_T,
                                           uint32_t* __restrict__ n_contrib, float* __restrict__ out,
                                           uint32_t* __restrict__ hint) {
  if (w >= hdr[8]) return;
  const uint2 wk = work[w];
  const uint32_t ns = tile_ns[wk.x];
  // hinted plan: everything in the first launch, nothing is decided on the way - the plan itself has chosen the tiles that
  // are composited in parallel segments (tile_mode 1); the other tiles are walked front to back by their first work item
  const bool hinted = hdr[11] != 0u;
  const bool walk_on = hinted && tile_mode[wk.x] != 1u;
  if (hinted ? (stage != 0 || wk.y >= ns || (walk_on && wk.y != 0u))
             : (stage == 0 ? wk.y != 0u : (wk.y == 0u || wk.y >= ns || tile_mode[wk.x] != 1u))) return;
  const int tile_x = wk.x % k.gx, tile_y = wk.x / k.gx;
  const long long seg = hdr[9];
  const int tid = threadIdx.x;
  const int px = tile_x * NM_TILE + (tid & 15), py = tile_y * NM_TILE + (tid >> 4);
  const bool inside = px < k.W && py < k.H;
  const float fxp = (float)px, fyp = (float)py;
  const int bin = (tile_y / NM_BT) * nbx + tile_x / NM_BT;
  const uint32_t bit = 1u << ((tile_y % NM_BT) * NM_BT + tile_x % NM_BT);
  const long long lo = off[bin * NM_NS], hi = min((long long)off[(bin + 1) * NM_NS], cap);
  const long long a = lo + seg_pos[w], b = min(hi, lo + (long long)seg_pos[w + 1]);     // (the plan's boundaries: q * seg by default)
  Pix p = {1.f, 0.f, 0.f, 0.f, 0u, !inside};
  composite_range(L, lo, a, b, bit, keys, vals, recs, (uint32_t)k.K * (uint32_t)sizeof(GRec), fxp, fyp, p);
  if (stage == 0 && (!hinted || walk_on)) {
    const bool far = walk_on ? false : __syncthreads_or(!p.done && p.T > NM_SPLIT_TAU);
    if (!(far && hdr[10])) {
      // nobody is far from stopping (or the view's lists are too much speculative work): this workgroup walks on as a whole
      // tile would, leaving a checkpoint (C, T) roughly every `seg` list entries - the reverse sweep starts its parallel
      // walks from them
      seg_ct[(size_t)w * NM_TPB + tid] = make_float4(0.f, 0.f, 0.f, 1.f);
      const uint32_t reached = composite_range(L, lo, b, hi, bit, keys, vals, recs, (uint32_t)k.K * (uint32_t)sizeof(GRec), fxp, fyp, p, (uint32_t)seg, 1u, ns,
                                               seg_ct + (size_t)w * NM_TPB, seg_pos + w);
      seg_ct[(size_t)(w + ns) * NM_TPB + tid] = make_float4(p.C0, p.C1, p.C2, p.T);
      if (inside) write_pixel(k, px, py, p, final_T, n_contrib, out);
      if (tid == 0) {
        if (hint) hint[wk.x] = reached;
        // a hinted plan may end in front of where this walk did: the reverse sweep's last segment reaches that far
        if (reached > seg_pos[w + ns]) seg_pos[w + ns] = reached;
      }
      return;
    }
    if (tid == 0) tile_mode[wk.x] = 1u;
  }
  // transmittance < 0 (-T: T >= 1e-4 whenever a pixel stops): the segment stopped by itself (its own T would have fallen below
  // 1e-4) - the pixel ends in it or earlier.  For the tile's first segment the record is already the truth (it starts at T = 1)
  seg_raw[(size_t)w * NM_TPB + tid] = make_float4(p.C0, p.C1, p.C2, (p.done && inside) ? -p.T : p.T);
  seg_last[(size_t)w * NM_TPB + tid] = p.last;
}
// one launch, 1-D grid: workgroups [0, ntile) = the tiles that are composited whole (stage 0 only), the others = the work
// items of the split compositing - the two kinds run side by side
__global__ void __launch_bounds__(NM_TPB) k_render(RK k, int nbx, int stage, int ntile, uint32_t* __restrict__ tile_mode,
                                                   const uint32_t* __restrict__ tile_rec, const uint32_t* __restrict__ tile_ns,
                                                   const uint32_t* __restrict__ off, const unsigned long long* __restrict__ keys,
                                                   const uint32_t* __restrict__ vals, long long cap, const uint32_t* __restrict__ hdr,
                                                   const uint2* __restrict__ work, float4* __restrict__ seg_raw,
                                                   uint32_t* __restrict__ seg_last, float4* __restrict__ seg_ct,
                                                   uint32_t* __restrict__ seg_pos,
                                                   const GRec* __restrict__ recs, float* __restrict__ final_T,
                                                   uint32_t* __restrict__ n_contrib, float* __restrict__ out,
                                                   uint32_t* __restrict__ hint) {
  __shared__ CompLds L;
  const int b = blockIdx.x;
#ifdef NM_FIXDBG
  if (stage == 0) FIXDBG(0);
#endif
  if (b < ntile)
    render_whole(L, k, nbx, b % k.gx, b / k.gx + k.ty0, off, keys, vals, cap, hdr, tile_rec, recs, final_T, n_contrib, out,
                 hint);
  else
    render_seg(L, k, nbx, stage, (uint32_t)(b - ntile), tile_mode, tile_ns, off, keys, vals, cap, hdr, work, seg_raw, seg_last, seg_ct,
               seg_pos, recs, final_T, n_contrib, out, hint);
#ifdef NM_FIXDBG
  if (stage == 0) FIXDBG(1);
#endif
}

// pass 2, again one workgroup per segment: with the transmittance in front of the segment known (product over the earlier
// segments' records) a pixel either passes through the segment (record kept), has stopped earlier (record cleared), or
// stops inside it - then the segment is walked again for those pixels from the true T, which reproduces the reference's
// termination exactly.  seg_fix receives the final records (what the segment really contributed, relative to its own
// start).  The last workgroup of a tile to finish sums the tile's records and leaves the checkpoints (C, T) in front of
// every segment for the reverse sweep.
__global__ void __launch_bounds__(NM_TPB) k_render_fix(RK k, int nbx, const uint32_t* __restrict__ off,
                                                       const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                       long long cap, const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ tile_rec,
                                                       const uint32_t* __restrict__ tile_ns, uint32_t* __restrict__ tile_cnt,
                                                       const uint32_t* __restrict__ tile_mode,
                                                       const uint2* __restrict__ work, const float4* __restrict__ seg_raw,
                                                       float4* __restrict__ seg_fix, float4* __restrict__ seg_ct,
                                                       uint32_t* __restrict__ seg_last, uint32_t* __restrict__ seg_pos,
                                                       const GRec* __restrict__ recs, float* __restrict__ final_T,
                                                       uint32_t* __restrict__ n_contrib, float* __restrict__ out,
                                                       uint32_t* __restrict__ hint) {
  __shared__ CompLds L;
  const uint32_t w = blockIdx.x;
  if (w >= hdr[8]) return;
  const uint2 wk = work[w];
  const uint32_t ns = tile_ns[wk.x], r0 = tile_rec[wk.x];
  if (tile_mode[wk.x] != 1u || wk.y >= ns) return;        // finished by its first segment's workgroup / the tile's extra record
  const int tile_x = wk.x % k.gx, tile_y = wk.x / k.gx;
  const int tid = threadIdx.x;
  const int px = tile_x * NM_TILE + (tid & 15), py = tile_y * NM_TILE + (tid >> 4);
  const bool inside = px < k.W && py < k.H;
  const float fxp = (float)px, fyp = (float)py;
  const int bin = (tile_y / NM_BT) * nbx + tile_x / NM_BT;
  const uint32_t bit = 1u << ((tile_y % NM_BT) * NM_BT + tile_x % NM_BT);
  const long long lo = off[bin * NM_NS], hi = min((long long)off[(bin + 1) * NM_NS], cap);
  FIXDBG(0);
  {
    float Tp = 1.f;
    bool stopped = !inside;
    for (uint32_t m0 = 0; m0 < wk.y && !stopped; m0 += 8) {        // (eight independent loads in flight, not a chain of round trips)
      float tw[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) tw[q] = m0 + q < wk.y ? seg_raw[(size_t)(r0 + m0 + q) * NM_TPB + tid].w : 1.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (stopped) continue;
        if (tw[q] < 0.f || Tp * tw[q] < 0.0001f) stopped = true;     // some Gaussian of segment m brings T below 1e-4
        else Tp *= tw[q];
      }
    }
    const size_t ri = (size_t)w * NM_TPB + tid;
    float4 ct = seg_raw[ri];
    uint32_t ls = seg_last[ri];
    // (segment 0 starts from the true T = 1: what pass 1 recorded stands, sign included)
    const bool need = wk.y != 0u && !stopped && (ct.w < 0.f || Tp * ct.w < 0.0001f);
    FIXDBG(1);
#ifdef NM_FIXDBG
    const int n_need = __syncthreads_count(need);
    FIXDBGV(5, (unsigned long long)n_need | ((unsigned long long)wk.y << 16) | ((unsigned long long)ns << 32));
#endif
    if (__syncthreads_or(need)) {
      const long long a = lo + seg_pos[w], b = min(hi, lo + (long long)seg_pos[w + 1]);
      Pix r = {Tp, 0.f, 0.f, 0.f, 0u, !need};
      composite_range(L, lo, a, b, bit, keys, vals, recs, (uint32_t)k.K * (uint32_t)sizeof(GRec), fxp, fyp, r);
      if (need) {
        const float inv = 1.f / Tp;
        // sign of the record's transmittance = "the pixel has stopped in this segment" (read by the summing workgroup below)
        ct = make_float4(r.C0 * inv, r.C1 * inv, r.C2 * inv, r.done ? -(r.T * inv) : r.T * inv);
        ls = r.last;
      }
    }
    if (stopped) { ct = make_float4(0.f, 0.f, 0.f, -1.f); ls = 0u; }
    seg_fix[ri] = ct;
    seg_last[ri] = ls;
  }
  FIXDBG(2);
}

// pass 3, one workgroup per split tile: sums the tile's records, leaves the checkpoints (C, T) in front of every segment for
// the reverse sweep, and walks on behind the planned stretch if a pixel is still alive there.  (A launch of its own: the
// kernel boundary orders it behind pass 2.  An arrival counter in pass 2 needed two device-scope fences per workgroup -
// on this chip each of them writes back and invalidates an XCD's L2, and a few thousand of them made every load of the
// kernel miss: 620 us for the metric view against 60 with the extra launch.)
__global__ void __launch_bounds__(NM_TPB) k_render_sum(RK k, int nbx, const uint32_t* __restrict__ off,
                                                       const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                       long long cap, const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ tile_rec,
                                                       const uint32_t* __restrict__ tile_ns, const uint32_t* __restrict__ tile_mode,
                                                       const float4* __restrict__ seg_fix, float4* __restrict__ seg_ct,
                                                       const uint32_t* __restrict__ seg_last, uint32_t* __restrict__ seg_pos,
                                                       const GRec* __restrict__ recs, float* __restrict__ final_T,
                                                       uint32_t* __restrict__ n_contrib, float* __restrict__ out,
                                                       uint32_t* __restrict__ hint) {
  __shared__ CompLds L;
  __shared__ uint32_t s_last;
  if (hdr[8] == 0u) return;
  const int tile_x = blockIdx.x % k.gx, tile_y = blockIdx.x / k.gx + k.ty0, t = tile_y * k.gx + tile_x;
  if (tile_mode[t] != 1u) return;
  const uint32_t ns = tile_ns[t], r0 = tile_rec[t];
  const int tid = threadIdx.x;
  const int px = tile_x * NM_TILE + (tid & 15), py = tile_y * NM_TILE + (tid >> 4);
  const bool inside = px < k.W && py < k.H;
  const float fxp = (float)px, fyp = (float)py;
  const int bin = (tile_y / NM_BT) * nbx + tile_x / NM_BT;
  const uint32_t bit = 1u << ((tile_y % NM_BT) * NM_BT + tile_x % NM_BT);
  const long long lo = off[bin * NM_NS], hi = min((long long)off[(bin + 1) * NM_NS], cap);
  if (tid == 0) s_last = 0u;
  Pix q = {1.f, 0.f, 0.f, 0.f, 0u, !inside};
  for (uint32_t s0 = 0; s0 < ns; s0 += 4) {          // (four records in flight)
    float4 ct[4]; uint32_t ls[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t ri = (size_t)(r0 + min(s0 + u, ns - 1u)) * NM_TPB + tid;
      ct[u] = seg_fix[ri]; ls[u] = seg_last[ri];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (s0 + u >= ns) break;
      seg_ct[(size_t)(r0 + s0 + u) * NM_TPB + tid] = make_float4(q.C0, q.C1, q.C2, q.T);
      q.C0 += q.T * ct[u].x; q.C1 += q.T * ct[u].y; q.C2 += q.T * ct[u].z;
      q.T *= fabsf(ct[u].w);
      q.done = q.done || __builtin_signbitf(ct[u].w);       // sign = the pixel has stopped in this segment or before
      if (ls[u]) q.last = ls[u];
    }
  }
  // list entries behind the planned stretch (hinted plans end at ~1.25 x the previous walk): walked here, from the true state
  const long long planned = lo + (long long)seg_pos[r0 + ns];
  uint32_t reached = (uint32_t)(planned - lo);
  const bool all = __syncthreads_and(q.done);
  if (!all && planned < hi) {
    reached = composite_range(L, lo, planned, hi, bit, keys, vals, recs, (uint32_t)k.K * (uint32_t)sizeof(GRec), fxp, fyp, q);
    if (tid == 0) seg_pos[r0 + ns] = reached;       // the reverse sweep's last segment ends where this walk did
  } else if (all && hint) {
    // where a sequential walk of this tile would have ended: just behind the last contributor
    if (q.last) atomicMax(&s_last, q.last);
    __syncthreads();
    reached = min(reached, s_last + 1u);
  }
  seg_ct[(size_t)(r0 + ns) * NM_TPB + tid] = make_float4(q.C0, q.C1, q.C2, q.T);
  if (inside) write_pixel(k, px, py, q, final_T, n_contrib, out);
  if (hint && tid == 0) hint[t] = reached;
}

// exact number of (Gaussian, tile) pairs of the view (what the reference's duplicateWithKeys would emit after the conic
// test): statistics for the byte accounting, not on the render path
__global__ void __launch_bounds__(256) k_count_pairs(RK k, int K, const int* __restrict__ radii, const float2* __restrict__ xy,
                                                     const float4* __restrict__ conop, unsigned long long* __restrict__ total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long cnt = 0;
  if (i < K && radii[i] > 0) {
    const float2 p = xy[i];
    int x0, y0, x1, y1;
    get_rect(k, p.x, p.y, radii[i], x0, y0, x1, y1, k.ty0, k.ty1);
    // (the binning passes' own function, bin by bin: the count is the number of mask bits they set)
    if ((x1 - x0) * (y1 - y0) != 0) {
      const RowCull rc = make_row_cull(p.x, p.y, conop[i]);
      const int4 g = make_int4(x0, y0, x1, y1);
      for (int by = y0 / NM_BT; by <= (y1 - 1) / NM_BT; ++by)
        for (int bx = x0 / NM_BT; bx <= (x1 - 1) / NM_BT; ++bx) cnt += (unsigned long long)__popc(bin_tile_mask(rc, g, bx, by));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += (unsigned long long)__shfl_xor((long long)cnt, o, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(total, cnt);
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
// The same sums with register-pair swaps for the two widest steps: v_permlane32_swap / v_permlane16_swap exchange halves
// (rows) BETWEEN two registers, which is exactly the keep / send exchange of a folding step - no selects, no DPP hazards.
// 8 -> 4 values over lane bit 5, 4 -> 2 over bit 4, 2 -> 1 over bit 3 (one select pair), then three single-value steps
// inside the groups of eight: 18 instructions instead of 28 + the s_nops of seven DPP adds in a row.
// On return the lanes with (lane & 7) == 0 hold the wave total of value  lane >> 3.
__device__ __forceinline__ float wave_fold8_swap(const float* v, int lane) {
  float a[4], c[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
    a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);      // lanes 0-31: value i, lanes 32-63: value i + 4
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 2]), false, false);
    c[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);      // even rows: a[i], odd rows: a[i + 2]
  }
  const bool b3 = lane & 8;
  const float keep = b3 ? c[1] : c[0], send = b3 ? c[0] : c[1];
  float d = keep + dpp_mov<0x128>(send);    // row_ror:8: the partner with bit 3 flipped
  d += dpp_mov<0xB1>(d);                    // lane ^ 1
  d += dpp_mov<0x4E>(d);                    // lane ^ 2
  d += dpp_mov<0x141>(d);                   // row_half_mirror: lane l reads lane 7 - l of its group of eight (the other quad)
  return d;
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
#define NM_RB_SCAN 2048
#define NM_NG 9  // per-Gaussian reduced quantities: ndc-mean(2) conic(3) colour(3) | opacity(1)
#define NM_NGS 16 // row stride of the per-Gaussian accumulator in global memory: a row's eight values in one aligned 32-byte
                  // piece of one line - the flush adds eight rows x eight consecutive floats per wave instruction, which is
                  // what the memory side's float atomics are fast at (tools/ubench_flush.hip: 166 G/s; 116 at stride 9; 20 with
                  // one row per lane and one instruction per column, which is what this kernel used to do)
// slot order inside an accumulator row: [0..7] = values of wave_fold8 order, [8] = opacity
//   v[0]=d/dndc.x v[1]=d/dndc.y v[2]=d/dconic.x v[3]=d/dconic.y v[4]=d/dconic.z v[5..7]=d/drgb

struct BwdLdsR {
  uint32_t hit[NM_RB_SCAN];
  uint32_t id[NM_RB_BATCH];
  float acc[4][NM_RB_BATCH * NM_NG];   // one private table per wave: plain stores, no LDS atomics
  int wcnt[4];
  uint32_t last[4];
};
// per-pixel state of the reverse walk
struct PixB {
  float T;                  // transmittance behind the Gaussian about to be visited
  float T_final;
  uint32_t last;            // positions > last do not contribute for this pixel
  float dp0, dp1, dp2;      // dL/dpixel
  float ar0, ar1, ar2;      // colour composited behind the Gaussian about to be visited (upstream renderCUDA backward: accum_rec)
};
// Reverse walk of one tile over list positions (floor, tile_top] of its bin (1-based, counted from lo): the bin's list is
// examined NM_RB_SCAN candidates at a time, back to front (tile-mask bit test); the survivors are staged in LDS NM_RB_BATCH
// at a time.  The four per-wave tables are what limits the number of resident tiles (LDS), and this loop lives on latency
// hiding - 128 per batch = 25 KB per tile = 6 waves per SIMD
template <bool WITH_OPACITY>
__device__ __forceinline__ void render_bwd_range(BwdLdsR& L, const RK& k, long long lo, uint32_t floor_pos, uint32_t bit,
                                                 const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                 const GRec* __restrict__ recs, float fxp, float fyp, PixB& P,
                                                 float* __restrict__ acc /* (K, 9) */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float bg_dot = k.bg[0] * P.dp0 + k.bg[1] * P.dp1 + k.bg[2] * P.dp2;
  const float kx = 0.5f * k.W / NM_LOG2E, ky = 0.5f * k.H / NM_LOG2E;
  float* my_acc = L.acc[wave];
  // fold slot (value index) of the lanes that end up holding a total:
  const int slot = lane >> 3;           // (wave_fold8_swap: lanes 0, 8, ..., 56 hold values 0..7)
  for (int i = lane; i < NM_RB_BATCH * NM_NG; i += 64) my_acc[i] = 0.f;
  // Gaussians behind every pixel's last contributor (the forward pass stopped compositing there) cannot
  // contribute: the wave skips them before doing any arithmetic, the tile skips whole batches of them
  uint32_t wave_last = P.last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, o, 64));
  wave_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_last);     // (uniform: lets `pos > wave_last` be a scalar branch)
  if (lane == 0) L.last[wave] = wave_last;
  __syncthreads();
  const uint32_t tile_last = max(max(L.last[0], L.last[1]), max(L.last[2], L.last[3]));
  const long long bottom = lo + (long long)floor_pos;       // absolute index of the first candidate of the range
  for (long long top = lo + (long long)tile_last; top > bottom; top -= NM_RB_SCAN) {
    __syncthreads();
    // ---- candidates top-1, top-2, ... (back to front): which of them touch this tile?  Thread t < 128 looks at sixteen.
    const long long c = top - 1 - 16 * tid;     // this thread's candidates: c, c-1, ..., c-15
    uint32_t m16 = 0;
    if (tid < NM_RB_SCAN / 16) {
      // (the sixteen masks are requested together, a candidate below the range reads the range's first entry and is masked
      //  out: `if (in range && (vals[..] & bit))` sixteen times was sixteen round trips in a row, round 5)
      uint32_t vv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) vv[q] = vals[max(c - q, bottom)];
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (c - q >= bottom && (vv[q] & bit)) m16 |= 1u << q;
    }
    const int mine = __popc(m16);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
    if (lane == 63) L.wcnt[wave] = incl;
    __syncthreads();
    int before = 0, nh = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int n = L.wcnt[w]; before += w < wave ? n : 0; nh += n; }
    {
      int hslot = before + incl - mine;
      for (uint32_t mm = m16; mm; mm &= mm - 1u) L.hit[hslot++] = (uint32_t)(c - (__ffs((int)mm) - 1) - lo) + 1u;   // 1-based, descending
    }
    for (int h0 = 0; h0 < nh; h0 += NM_RB_BATCH) {
    __syncthreads();
    const int nb = min(NM_RB_BATCH, nh - h0);
    // no staging of Gaussian data: every wave keeps the batch's record offsets and list positions in two registers each
    // (lane l: hits l and l + 64), pulls hit j's out with v_readlane and fetches the record with scalar loads (see GRec)
    uint32_t offr[NM_RB_BATCH / 64], posr[NM_RB_BATCH / 64];
#pragma unroll
    for (int q = 0; q < NM_RB_BATCH / 64; ++q) {      // both groups' keys requested at clamped slots ...
      posr[q] = L.hit[h0 + min(64 * q + lane, nb - 1)];
      offr[q] = (uint32_t)keys[lo + posr[q] - 1];
    }
#pragma unroll
    for (int q = 0; q < NM_RB_BATCH / 64; ++q) {
      const int t = 64 * q + lane;
      asm volatile("" : "+v"(offr[q]));      // ... and pinned as unconditional (the compiler sinks a conditional load into its branch)
      const uint32_t id = offr[q];
      offr[q] = t < nb ? id * (uint32_t)sizeof(GRec) : k.K * (uint32_t)sizeof(GRec);        // padding: the null record ...
      posr[q] = t < nb ? posr[q] : 0xFFFFFFFFu;                                             // ... behind everything
      if (t < nb && wave == 0) L.id[t] = id;
    }
    auto one = [&](const float4& g0, const float4& g1, const float2& g2, uint32_t pos, int j) {
      if (pos > wave_last) return;              // wave-uniform: behind every pixel's last contributor
      const float dx = g0.x - fxp, dy = g0.y - fyp;
      const float e2 = dx * (g0.z * dx + g0.w * dy) + (g1.x * dy) * dy;      // log2 G
      const float G = __builtin_amdgcn_exp2f(e2);
      const float alpha = fminf(0.99f, g2.y * G);
      const bool act = pos <= P.last && !(e2 > 0.f) && !(alpha < 1.0f / 255.0f);
      if (__ballot(act) == 0ull) return;      // whole wave skips this Gaussian
      // Mask-free recurrence: an inactive lane takes part with alpha = 0 - its T (x 1 / (1 - 0)) and its colour behind
      // (ar <- alpha c + (1 - alpha) ar) stay as they are and every gradient term carries a factor alpha or dL/dalpha, which
      // is zeroed for it.  Upstream keeps (last_alpha, last_colour) and applies them one Gaussian late; updating `ar` right
      // after its use is the same recurrence without the two extra state variables and their per-lane selects.
      const float al = act ? alpha : 0.f;
      // one hardware reciprocal (1 ulp) for both quotients below: the IEEE divisions were a quarter of the
      // instructions of an evaluated (pixel, Gaussian) pair; alpha <= 0.99 keeps the denominator >= 0.01
      const float inv1ma = __builtin_amdgcn_rcpf(1.f - al);
      P.T = P.T * inv1ma;
      const float dch = al * P.T;
      const float c0 = g1.z, c1 = g1.w, c2 = g2.x;
      float dL_dalpha = (c0 - P.ar0) * P.dp0 + (c1 - P.ar1) * P.dp1 + (c2 - P.ar2) * P.dp2;
      P.ar0 = al * c0 + (1.f - al) * P.ar0;
      P.ar1 = al * c1 + (1.f - al) * P.ar1;
      P.ar2 = al * c2 + (1.f - al) * P.ar2;
      float g[8];
      g[5] = dch * P.dp0; g[6] = dch * P.dp1; g[7] = dch * P.dp2;
      dL_dalpha *= P.T;
      dL_dalpha += (-P.T_final * inv1ma) * bg_dot;
      dL_dalpha = act ? dL_dalpha : 0.f;
      const float dL_dG = g2.y * dL_dalpha;
      const float Gm = act ? G : 0.f;        // (an inactive lane's G may be inf - e2 > 0 - and 0 x inf would poison the wave's sums)
      const float gdx = Gm * dx, gdy = Gm * dy;
      // dG/d(delta) = -G (conic . delta) with conic = -(2a, b, 2c) / log2(e): the 1/log2(e) sits in kx, ky
      g[0] = dL_dG * (2.f * g0.z * gdx + g0.w * gdy) * kx;   // d/d(ndc x)
      g[1] = dL_dG * (2.f * g1.x * gdy + g0.w * gdx) * ky;
      g[2] = -0.5f * gdx * dx * dL_dG;       // d/d conic.x
      g[3] = -gdx * dy * dL_dG;              // d/d conic.y (full off-diagonal derivative)
      g[4] = -0.5f * gdy * dy * dL_dG;       // d/d conic.z
      const float gop = Gm * dL_dalpha;      // d/d opacity
      // (each Gaussian of a batch is visited once per wave and the table starts from zero: a plain store, no read-modify-write)
      const float tot = wave_fold8_swap(g, lane);
      if ((lane & 7) == 0) my_acc[j * NM_NG + slot] = tot;
      if (WITH_OPACITY) {
        const float to = wave_sum_dpp(gop);
        if (lane == 0) my_acc[j * NM_NG + 8] = to;
      }
    };
    struct Set { float4 a[NM_G], b[NM_G]; float2 c[NM_G]; uint32_t pos[NM_G]; };
    auto fetch = [&](Set& g, uint32_t offs, uint32_t poss, int j0) {
#pragma unroll
      for (int u = 0; u < NM_G; ++u) {
        const char* rp = (const char*)recs + (uint32_t)__builtin_amdgcn_readlane((int)offs, j0 + u);
        g.a[u] = *(const float4*)rp; g.b[u] = *(const float4*)(rp + 16); g.c[u] = *(const float2*)(rp + 32);
        g.pos[u] = (uint32_t)__builtin_amdgcn_readlane((int)poss, j0 + u);
      }
    };
#pragma unroll
    for (int q = 0; q < NM_RB_BATCH / 64; ++q) {
      const int nq = min(64, nb - 64 * q);          // (wave-uniform)
      if (nq <= 0) break;
      Set A, B;
      fetch(A, offr[q], posr[q], 0);
      for (int j0 = 0; j0 < nq; j0 += 2 * NM_G) {
        fetch(B, offr[q], posr[q], j0 + NM_G);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NM_G; ++u) one(A.a[u], A.b[u], A.c[u], A.pos[u], 64 * q + j0 + u);
        fetch(A, offr[q], posr[q], (j0 + 2 * NM_G) & 63);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NM_G; ++u) one(B.a[u], B.b[u], B.c[u], B.pos[u], 64 * q + j0 + NM_G + u);
      }
    }
    __syncthreads();
    // one global atomic set per (tile, Gaussian): sum the four wave tables; a wave instruction covers eight Gaussians' rows
    for (int r = tid >> 3; r < nb; r += NM_TPB / 8) {
      const int o = r * NM_NG + (tid & 7);
      const float v = (L.acc[0][o] + L.acc[1][o]) + (L.acc[2][o] + L.acc[3][o]);
      if (v != 0.f) unsafeAtomicAdd(acc + (size_t)L.id[r] * NM_NGS + (tid & 7), v);
    }
    if (WITH_OPACITY && tid < nb) {
      const int o = tid * NM_NG + 8;
      const float v = (L.acc[0][o] + L.acc[1][o]) + (L.acc[2][o] + L.acc[3][o]);
      if (v != 0.f) unsafeAtomicAdd(acc + (size_t)L.id[tid] * NM_NGS + 8, v);
    }
    __syncthreads();
    for (int i = lane; i < nb * NM_NG; i += 64) my_acc[i] = 0.f;
    }
  }
}

template <bool WITH_OPACITY>
__device__ __forceinline__ void bwd_whole(BwdLdsR& L, const RK& k, int nbx, int tile_x, int tile_y, const uint32_t* __restrict__ off,
                                          const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                          const uint32_t* __restrict__ tile_rec,
                                          const GRec* __restrict__ recs, const float* __restrict__ final_T,
                                          const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                                          float* __restrict__ acc /* (K, 9) */) {
  if (tile_rec[tile_y * k.gx + tile_x] != 0xFFFFFFFFu) return;      // bwd_seg
  const int tid = threadIdx.x;
  const int px = tile_x * NM_TILE + (tid & 15), py = tile_y * NM_TILE + (tid >> 4);
  const bool inside = px < k.W && py < k.H;
  const int bin = (tile_y / NM_BT) * nbx + tile_x / NM_BT;
  const uint32_t bit = 1u << ((tile_y % NM_BT) * NM_BT + tile_x % NM_BT);
  const long long lo = off[bin * NM_NS];
  const size_t pix = (size_t)py * k.W + px, hw = (size_t)k.H * k.W;
  PixB P = {};
  if (inside) {
    P.T_final = final_T[pix];
    P.last = n_contrib[pix];                       // 1-based position in the bin's list, 0 = none
    P.dp0 = dL_dpix[pix]; P.dp1 = dL_dpix[hw + pix]; P.dp2 = dL_dpix[2 * hw + pix];
  }
  P.T = P.T_final;
  render_bwd_range<WITH_OPACITY>(L, k, lo, 0u, bit, keys, vals, recs, (float)px, (float)py, P, acc);
}

// reverse walk of one segment of a candidate tile (k_split_plan), all segments of a tile in parallel.  The state a pixel
// arrives with at the top of segment s comes from the forward pass's checkpoints: T behind the segment = T in front of
// segment s+1, colour composited behind it (seen from there) = (C_final - C in front of s+1) / T in front of s+1.  Feeding
// that colour as the recurrence's `colour behind` makes it start from it.
template <bool WITH_OPACITY>
__device__ __forceinline__ void bwd_seg(BwdLdsR& L, const RK& k, int nbx, uint32_t w, const uint32_t* __restrict__ off,
                                        const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                        const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ tile_rec,
                                        const uint32_t* __restrict__ tile_ns, const uint2* __restrict__ work,
                                        const float4* __restrict__ seg_ct, const uint32_t* __restrict__ seg_pos,
                                        const GRec* __restrict__ recs, const float* __restrict__ final_T,
                                        const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                                        float* __restrict__ acc /* (K, 9) */) {
  if (w >= hdr[8]) return;
  const uint2 wk = work[w];
  const uint32_t seg = hdr[9], sg = wk.y, ns = tile_ns[wk.x], r0 = tile_rec[wk.x];
  if (sg >= ns) return;            // the tile's extra record
  const int tile_x = wk.x % k.gx, tile_y = wk.x / k.gx;
  const int tid = threadIdx.x;
  const int px = tile_x * NM_TILE + (tid & 15), py = tile_y * NM_TILE + (tid >> 4);
  const bool inside = px < k.W && py < k.H;
  const int bin = (tile_y / NM_BT) * nbx + tile_x / NM_BT;
  const uint32_t bit = 1u << ((tile_y % NM_BT) * NM_BT + tile_x % NM_BT);
  const long long lo = off[bin * NM_NS];
  const size_t pix = (size_t)py * k.W + px, hw = (size_t)k.H * k.W;
  const uint32_t floor_pos = seg_pos[r0 + sg], ceil_pos = seg_pos[r0 + sg + 1];      // the segment holds positions floor_pos+1 .. ceil_pos
  if (ceil_pos <= floor_pos) return;                                                   // (an empty segment: the walk passed two nominal starts in one batch)
  PixB P = {};
  uint32_t last = 0u;
  if (inside) {
    P.T_final = final_T[pix];
    last = n_contrib[pix];
    P.dp0 = dL_dpix[pix]; P.dp1 = dL_dpix[hw + pix]; P.dp2 = dL_dpix[2 * hw + pix];
  }
  P.T = P.T_final;
  if (last > ceil_pos) {            // the pixel's last contributor lies in a later segment
    const float4 nx = seg_ct[(size_t)(r0 + sg + 1) * NM_TPB + tid], fin = seg_ct[(size_t)(r0 + ns) * NM_TPB + tid];
    const float inv = 1.f / nx.w;
    P.T = nx.w;
    P.ar0 = (fin.x - nx.x) * inv; P.ar1 = (fin.y - nx.y) * inv; P.ar2 = (fin.z - nx.z) * inv;
    P.last = ceil_pos;
  } else {
    P.last = last > floor_pos ? last : 0u;      // ends in this segment, or in an earlier one (nothing to do here)
  }
  render_bwd_range<WITH_OPACITY>(L, k, lo, floor_pos, bit, keys, vals, recs, (float)px, (float)py, P, acc);
}

// ---------------------------------------------------------------- reverse compositing, two pixels per lane
// The same walk with 128 threads per tile: lane (x, h) owns the pixels (x, 2h) and (x, 2h + 1) of the tile, a wave a 16 x 8
// block.  The kernel is instruction-issue bound (§5): with two pixels per lane the per-Gaussian overhead of a wave (record
// fetch, skip tests, the fold of the eight partial gradients over the lanes) is paid once for 128 pixels instead of twice,
// and the arithmetic in between runs on float2 operands (v_pk_fma_f32 / v_pk_mul_f32: two results for 6.2 cycles against
// 5.3 for one).  Per pixel the expressions are those of render_bwd_range, element by element.
typedef float f2r __attribute__((ext_vector_type(2)));
struct BwdLdsR2 {
  uint32_t hit[NM_RB_SCAN];
  uint32_t id[NM_RB_BATCH];
  float acc[2][NM_RB_BATCH * NM_NG];
  int wcnt[2];
  uint32_t last[2];
};
struct PixB2 {
  f2r T, T_final;
  uint32_t last[2];
  f2r dp0, dp1, dp2, ar0, ar1, ar2;
};
template <bool WITH_OPACITY>
__device__ __forceinline__ void render_bwd_range2(BwdLdsR2& L, const RK& k, long long lo, uint32_t floor_pos, uint32_t bit,
                                                  const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                  const GRec* __restrict__ recs, float fxp, float fy0, PixB2& P,
                                                  float* __restrict__ acc /* (K, 9) */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f2r bg_dot = k.bg[0] * P.dp0 + k.bg[1] * P.dp1 + k.bg[2] * P.dp2;
  const float kx = 0.5f * k.W / NM_LOG2E, ky = 0.5f * k.H / NM_LOG2E;
  const f2r fy = {fy0, fy0 + 1.f};
  float* my_acc = L.acc[wave];
  const int slot = lane >> 3;           // (wave_fold8_swap: lanes 0, 8, ..., 56 hold values 0..7)
  for (int i = lane; i < NM_RB_BATCH * NM_NG; i += 64) my_acc[i] = 0.f;
  uint32_t wave_last = max(P.last[0], P.last[1]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, o, 64));
  wave_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_last);
  if (lane == 0) L.last[wave] = wave_last;
  __syncthreads();
  const uint32_t tile_last = max(L.last[0], L.last[1]);
  const long long bottom = lo + (long long)floor_pos;
  for (long long top = lo + (long long)tile_last; top > bottom; top -= NM_RB_SCAN) {
    __syncthreads();
    const long long c = top - 1 - 16 * tid;     // this thread's candidates: c, c-1, ..., c-15 (128 threads x 16 = NM_RB_SCAN)
    uint32_t m16 = 0;
    uint32_t vv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) vv[q] = vals[max(c - q, bottom)];
#pragma unroll
    for (int q = 0; q < 16; ++q)
      if (c - q >= bottom && (vv[q] & bit)) m16 |= 1u << q;
    const int mine = __popc(m16);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
    if (lane == 63) L.wcnt[wave] = incl;
    __syncthreads();
    const int n0 = L.wcnt[0], nh = n0 + L.wcnt[1];
    {
      int hslot = (wave ? n0 : 0) + incl - mine;
      for (uint32_t mm = m16; mm; mm &= mm - 1u) L.hit[hslot++] = (uint32_t)(c - (__ffs((int)mm) - 1) - lo) + 1u;   // 1-based, descending
    }
    for (int h0 = 0; h0 < nh; h0 += NM_RB_BATCH) {
      __syncthreads();
      const int nb = min(NM_RB_BATCH, nh - h0);
      uint32_t offr[NM_RB_BATCH / 64], posr[NM_RB_BATCH / 64];
#pragma unroll
      for (int q = 0; q < NM_RB_BATCH / 64; ++q) {
        posr[q] = L.hit[h0 + min(64 * q + lane, nb - 1)];
        offr[q] = (uint32_t)keys[lo + posr[q] - 1];
      }
#pragma unroll
      for (int q = 0; q < NM_RB_BATCH / 64; ++q) {
        const int t = 64 * q + lane;
        asm volatile("" : "+v"(offr[q]));
        const uint32_t id = offr[q];
        offr[q] = t < nb ? id * (uint32_t)sizeof(GRec) : k.K * (uint32_t)sizeof(GRec);        // padding: the null record ...
        posr[q] = t < nb ? posr[q] : 0xFFFFFFFFu;                                             // ... behind everything
        if (t < nb && wave == 0) L.id[t] = id;
      }
      auto one = [&](const float4& g0, const float4& g1, const float2& g2, uint32_t pos, int j) {
        if (pos > wave_last) return;              // wave-uniform: behind every pixel's last contributor
        const float dxs = g0.x - fxp;
        const f2r dx = {dxs, dxs};
        const f2r dy = g0.y - fy;
        const f2r e2 = dx * (g0.z * dx + g0.w * dy) + (g1.x * dy) * dy;      // log2 G, element by element as in the forward pass
        f2r G;
        G[0] = __builtin_amdgcn_exp2f(e2[0]); G[1] = __builtin_amdgcn_exp2f(e2[1]);
        f2r alpha = g2.y * G;
        alpha[0] = fminf(0.99f, alpha[0]); alpha[1] = fminf(0.99f, alpha[1]);
        const bool act0 = pos <= P.last[0] && !(e2[0] > 0.f) && !(alpha[0] < 1.0f / 255.0f);
        const bool act1 = pos <= P.last[1] && !(e2[1] > 0.f) && !(alpha[1] < 1.0f / 255.0f);
        if (__ballot(act0 || act1) == 0ull) return;      // whole wave skips this Gaussian
        // mask-free recurrence (see render_bwd_range): an inactive pixel takes part with alpha = 0
        f2r al;
        al[0] = act0 ? alpha[0] : 0.f; al[1] = act1 ? alpha[1] : 0.f;
        const f2r oma = 1.f - al;
        f2r inv1ma;
        inv1ma[0] = __builtin_amdgcn_rcpf(oma[0]); inv1ma[1] = __builtin_amdgcn_rcpf(oma[1]);
        P.T = P.T * inv1ma;
        const f2r dch = al * P.T;
        const float c0 = g1.z, c1 = g1.w, c2 = g2.x;
        f2r dL_dalpha = (c0 - P.ar0) * P.dp0 + (c1 - P.ar1) * P.dp1 + (c2 - P.ar2) * P.dp2;
        P.ar0 = al * c0 + oma * P.ar0;
        P.ar1 = al * c1 + oma * P.ar1;
        P.ar2 = al * c2 + oma * P.ar2;
        const f2r q5 = dch * P.dp0, q6 = dch * P.dp1, q7 = dch * P.dp2;
        dL_dalpha *= P.T;
        dL_dalpha += (-P.T_final * inv1ma) * bg_dot;
        dL_dalpha[0] = act0 ? dL_dalpha[0] : 0.f; dL_dalpha[1] = act1 ? dL_dalpha[1] : 0.f;
        const f2r dL_dG = g2.y * dL_dalpha;
        f2r Gm;        // (an inactive pixel's G may be inf - e2 > 0 - and 0 x inf would poison the sums)
        Gm[0] = act0 ? G[0] : 0.f; Gm[1] = act1 ? G[1] : 0.f;
        const f2r gdx = Gm * dx, gdy = Gm * dy;
        const f2r q0 = dL_dG * (2.f * g0.z * gdx + g0.w * gdy) * kx;   // d/d(ndc x)
        const f2r q1 = dL_dG * (2.f * g1.x * gdy + g0.w * gdx) * ky;
        const f2r q2 = -0.5f * gdx * dx * dL_dG;       // d/d conic.x
        const f2r q3 = -gdx * dy * dL_dG;              // d/d conic.y
        const f2r q4 = -0.5f * gdy * dy * dL_dG;       // d/d conic.z
        float g[8];
        g[0] = q0[0] + q0[1]; g[1] = q1[0] + q1[1]; g[2] = q2[0] + q2[1]; g[3] = q3[0] + q3[1];
        g[4] = q4[0] + q4[1]; g[5] = q5[0] + q5[1]; g[6] = q6[0] + q6[1]; g[7] = q7[0] + q7[1];
        const float tot = wave_fold8_swap(g, lane);
        if ((lane & 7) == 0) my_acc[j * NM_NG + slot] = tot;
        if (WITH_OPACITY) {
          const f2r gop = Gm * dL_dalpha;      // d/d opacity
          const float to = wave_sum_dpp(gop[0] + gop[1]);
          if (lane == 0) my_acc[j * NM_NG + 8] = to;
        }
      };
      struct Set { float4 a[NM_G], b[NM_G]; float2 c[NM_G]; uint32_t pos[NM_G]; };
      auto fetch = [&](Set& g, uint32_t offs, uint32_t poss, int j0) {
#pragma unroll
        for (int u = 0; u < NM_G; ++u) {
          const char* rp = (const char*)recs + (uint32_t)__builtin_amdgcn_readlane((int)offs, j0 + u);
          g.a[u] = *(const float4*)rp; g.b[u] = *(const float4*)(rp + 16); g.c[u] = *(const float2*)(rp + 32);
          g.pos[u] = (uint32_t)__builtin_amdgcn_readlane((int)poss, j0 + u);
        }
      };
#pragma unroll
      for (int q = 0; q < NM_RB_BATCH / 64; ++q) {
        const int nq = min(64, nb - 64 * q);          // (wave-uniform)
        if (nq <= 0) break;
        Set A, B;
        fetch(A, offr[q], posr[q], 0);
        for (int j0 = 0; j0 < nq; j0 += 2 * NM_G) {
          fetch(B, offr[q], posr[q], j0 + NM_G);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < NM_G; ++u) one(A.a[u], A.b[u], A.c[u], A.pos[u], 64 * q + j0 + u);
          fetch(A, offr[q], posr[q], (j0 + 2 * NM_G) & 63);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < NM_G; ++u) one(B.a[u], B.b[u], B.c[u], B.pos[u], 64 * q + j0 + NM_G + u);
        }
      }
      __syncthreads();
      // one global atomic set per (tile, Gaussian): sum the two wave tables
      for (int r = tid >> 3; r < nb; r += 16) {
        const int o = r * NM_NG + (tid & 7);
        const float v = L.acc[0][o] + L.acc[1][o];
        if (v != 0.f) unsafeAtomicAdd(acc + (size_t)L.id[r] * NM_NGS + (tid & 7), v);
      }
      if (WITH_OPACITY && tid < nb) {
        const int o = tid * NM_NG + 8;
        const float v = L.acc[0][o] + L.acc[1][o];
        if (v != 0.f) unsafeAtomicAdd(acc + (size_t)L.id[tid] * NM_NGS + 8, v);
      }
      __syncthreads();
      for (int i = lane; i < nb * NM_NG; i += 64) my_acc[i] = 0.f;
    }
  }
}

// the state the two pixels of a lane arrive with: whole tile (segment == nullptr) or the segment [floor_pos, ceil_pos) of a
// split tile, from the forward pass's checkpoints (see bwd_seg)
template <bool WITH_OPACITY>
__global__ void __launch_bounds__(128) k_render_bwd2(RK k, int nbx, int ntile, const uint32_t* __restrict__ off,
                                                     const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                     const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ tile_rec,
                                                     const uint32_t* __restrict__ tile_ns, const uint2* __restrict__ work,
                                                     const float4* __restrict__ seg_ct, const uint32_t* __restrict__ seg_pos,
                                                     const GRec* __restrict__ recs, const float* __restrict__ final_T,
                                                     const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                                                     float* __restrict__ acc /* (K, 9) */) {
  __shared__ BwdLdsR2 L;
  if (hdr[3]) return;          // (overflowed render: no gradient, see k_render_bwd)
  const int b = blockIdx.x;
  int tile;
  bool seg = false;
  uint32_t sg = 0, ns = 0, r0 = 0;
  if (b < ntile) {
    tile = (b / k.gx + k.ty0) * k.gx + b % k.gx;
    if (tile_rec[tile] != 0xFFFFFFFFu) return;      // a split tile: its segments do the work
  } else {
    const uint32_t w = (uint32_t)(b - ntile);
    if (w >= hdr[8]) return;
    const uint2 wk = work[w];
    tile = (int)wk.x; sg = wk.y; ns = tile_ns[wk.x]; r0 = tile_rec[wk.x];
    if (sg >= ns) return;            // the tile's extra record
    seg = true;
  }
  const int tile_x = tile % k.gx, tile_y = tile / k.gx;
  const int tid = threadIdx.x;
  const int px = tile_x * NM_TILE + (tid & 15), py = tile_y * NM_TILE + 2 * (tid >> 4);
  const int bin = (tile_y / NM_BT) * nbx + tile_x / NM_BT;
  const uint32_t bit = 1u << ((tile_y % NM_BT) * NM_BT + tile_x % NM_BT);
  const long long lo = off[bin * NM_NS];
  const size_t hw = (size_t)k.H * k.W;
  uint32_t floor_pos = 0u, ceil_pos = 0xFFFFFFFFu;
  if (seg) {
    floor_pos = seg_pos[r0 + sg]; ceil_pos = seg_pos[r0 + sg + 1];      // the segment holds positions floor_pos+1 .. ceil_pos
    if (ceil_pos <= floor_pos) return;
  }
  PixB2 P = {};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const bool inside = px < k.W && py + e < k.H;
    uint32_t last = 0u;
    if (inside) {
      const size_t pix = (size_t)(py + e) * k.W + px;
      P.T_final[e] = final_T[pix];
      last = n_contrib[pix];                       // 1-based position in the bin's list, 0 = none
      P.dp0[e] = dL_dpix[pix]; P.dp1[e] = dL_dpix[hw + pix]; P.dp2[e] = dL_dpix[2 * hw + pix];
    }
    P.T[e] = P.T_final[e];
    if (!seg) {
      P.last[e] = last;
    } else if (last > ceil_pos) {            // the pixel's last contributor lies in a later segment
      const int pin = (2 * (tid >> 4) + e) * NM_TILE + (tid & 15);      // the pixel's index in its tile (the forward pass's thread)
      const float4 nx = seg_ct[(size_t)(r0 + sg + 1) * NM_TPB + pin], fin = seg_ct[(size_t)(r0 + ns) * NM_TPB + pin];
      const float inv = 1.f / nx.w;
      P.T[e] = nx.w;
      P.ar0[e] = (fin.x - nx.x) * inv; P.ar1[e] = (fin.y - nx.y) * inv; P.ar2[e] = (fin.z - nx.z) * inv;
      P.last[e] = ceil_pos;
    } else {
      P.last[e] = last > floor_pos ? last : 0u;      // ends in this segment, or in an earlier one (nothing to do here)
    }
  }
  render_bwd_range2<WITH_OPACITY>(L, k, lo, floor_pos, bit, keys, vals, recs, (float)px, (float)py, P, acc);
}

// one launch, 1-D grid: workgroups [0, ntile) = whole tiles, the others = segments of the candidate tiles, side by side
template <bool WITH_OPACITY>
__global__ void __launch_bounds__(NM_TPB) k_render_bwd(RK k, int nbx, int ntile, const uint32_t* __restrict__ off,
                                                       const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                       const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ tile_rec,
                                                       const uint32_t* __restrict__ tile_ns, const uint2* __restrict__ work,
                                                       const float4* __restrict__ seg_ct, const uint32_t* __restrict__ seg_pos,
                                                       const GRec* __restrict__ recs, const float* __restrict__ final_T,
                                                       const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix,
                                                       float* __restrict__ acc /* (K, 9) */) {
  __shared__ BwdLdsR L;
  const int b = blockIdx.x;
  // A render whose bin lists overflowed composited the background only and left lists / checkpoints unwritten: it contributes
  // no gradient (the accumulators stay zero).  The host learns of the overflow from the status words and raises; this guard
  // is what lets it do so without stalling the reverse sweep on the forward pass's completion.
  if (hdr[3]) return;
  if (b < ntile)
    bwd_whole<WITH_OPACITY>(L, k, nbx, b % k.gx, b / k.gx + k.ty0, off, keys, vals, tile_rec, recs, final_T, n_contrib, dL_dpix, acc);
  else
    bwd_seg<WITH_OPACITY>(L, k, nbx, (uint32_t)(b - ntile), off, keys, vals, hdr, tile_rec, tile_ns, work, seg_ct, seg_pos, recs,
                          final_T, n_contrib, dL_dpix, acc);
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
  const float* a = acc + (size_t)i * NM_NGS;
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
      float sh[48];
      load_sh_row(shs, (size_t)i, k.M, sh);
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
static int g_split_busy = NM_SPLIT_BUSY, g_split_minseg = NM_SPLIT_MINSEG;
static long long g_split_fwd = NM_SPLIT_FWD_MAX;
static int g_hint_fwd = NM_HINT_FWD, g_hint_seg = NM_HINT_MINSEG;
extern "C" int nm_raster_set_hinted(int32_t forward_split_length, int32_t min_segment) {
  NM_REQUIRE(forward_split_length >= 0 && min_segment >= 1, "forward_split_length >= 0, min_segment >= 1");
  g_hint_fwd = forward_split_length;
  g_hint_seg = (min_segment + 15) & ~15;
  return NM_OK;
}
// reverse compositing with two pixels per lane (k_render_bwd2): 0 off, 1 on.  NM_BWD_PX2 in the environment overrides (A/B runs)
static int g_bwd_px2 = 0;
extern "C" int nm_raster_set_reverse_px2(int32_t on) {
  g_bwd_px2 = on != 0;
  return NM_OK;
}
extern "C" int nm_raster_set_split(int32_t busy_tiles, int32_t min_segment, int64_t forward_budget) {
  NM_REQUIRE(busy_tiles >= 0 && min_segment >= 1 && forward_budget >= 0, "busy_tiles >= 0, min_segment >= 1, forward_budget >= 0");
  g_split_busy = busy_tiles;
  g_split_minseg = min_segment;
  g_split_fwd = forward_budget;
  return NM_OK;
}
// One view, forward: 6 launches, no host synchronisation.  status_host (optional, pinned host memory, 2 x int64): receives
// {pairs binned, overflow flag} by an asynchronous copy at the end - overflow != 0 means cap_pairs was too small and the image
// is incomplete (the caller re-runs with a capacity >= pairs).
static int raster_forward_impl(const nm_raster_cfg* cfg, int32_t K, int32_t m, const float* means3D, const float* shs,
                               const float* colors_precomp, const float* opacities, const float* cov3D, int32_t* radii,
                               void* state, size_t state_bytes, void* scratch, size_t scratch_bytes, int64_t cap_pairs,
                               float* out_color, int64_t* status_host, int status_words, uint32_t* tile_walk, void* stream);
extern "C" int nm_raster_forward(const nm_raster_cfg* cfg, int32_t K, int32_t m, const float* means3D, const float* shs,
                                 const float* colors_precomp, const float* opacities, const float* cov3D, int32_t* radii,
                                 void* state, size_t state_bytes, int64_t cap_pairs, float* out_color, int64_t* status_host,
                                 void* stream) {
  return raster_forward_impl(cfg, K, m, means3D, shs, colors_precomp, opacities, cov3D, radii, state, state_bytes, nullptr, 0,
                             cap_pairs, out_color, status_host, 2, nullptr, stream);
}
// tile_walk (optional, device, one uint32 per 16x16 tile of the image, zero-initialised by the caller once per camera): in =
// how far the previous forward walk of every tile with this camera went, out = the same for this render.  See k_split_plan.
// scratch (optional): the forward-only arrays live there instead of behind the persistent part of `state`.
extern "C" int nm_raster_forward_ex(const nm_raster_cfg* cfg, int32_t K, int32_t m, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities, const float* cov3D, int32_t* radii,
                                    void* state, size_t state_bytes, void* scratch, size_t scratch_bytes, int64_t cap_pairs,
                                    float* out_color, int64_t* status_host, uint32_t* tile_walk, void* stream) {
  return raster_forward_impl(cfg, K, m, means3D, shs, colors_precomp, opacities, cov3D, radii, state, state_bytes, scratch,
                             scratch_bytes, cap_pairs, out_color, status_host, 3, tile_walk, stream);
}
static int raster_forward_impl(const nm_raster_cfg* cfg, int32_t K, int32_t m, const float* means3D, const float* shs,
                               const float* colors_precomp, const float* opacities, const float* cov3D, int32_t* radii,
                               void* state, size_t state_bytes, void* scratch, size_t scratch_bytes, int64_t cap_pairs,
                               float* out_color, int64_t* status_host, int status_words, uint32_t* tile_walk, void* stream) {
  RK k;
  int rc = make_rk(cfg, m, k);
  if (rc) return rc;
  NM_REQUIRE(K >= 0 && cap_pairs >= 0 && out_color && state, "bad arguments");
  k.K = K;
  NM_REQUIRE(K == 0 || (shs != nullptr) != (colors_precomp != nullptr), "provide exactly one of shs / colors_precomp");
  NM_REQUIRE(!shs || m >= (cfg->sh_degree + 1) * (cfg->sh_degree + 1), "shs has too few coefficients for sh_degree");
  NM_REQUIRE(K == 0 || (means3D && opacities && cov3D && radii), "null pointer");
  State t = carve_state(state, scratch, k.W, k.H, K, cap_pairs, k.items);
  {
    const size_t need = scratch ? t.total : t.total + t.scratch_total;
    if (state_bytes < need) { nm_set_error("raster state buffer too small: need %zu got %zu", need, state_bytes); return NM_ERR_WORKSPACE; }
    if (scratch && scratch_bytes < t.scratch_total) {
      nm_set_error("raster scratch buffer too small: need %zu got %zu", t.scratch_total, scratch_bytes);
      return NM_ERR_WORKSPACE;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  const int nrange = nm_div_up(K, 256);
  const int nbin = t.nbx * t.nby;
  if (K == 0) NM_HIP_CHECK(hipMemsetAsync(t.hdr, 0, 256, s));      // (otherwise k_preprocess zeroes the header)
  static const int binning_cells = [] { const char* e = getenv("NEUMA_BINNING"); return (e && !strcmp(e, "cells")) ? 1 : 0; }();
  const size_t b2_lds = ((size_t)nbin * 9 + 1) * sizeof(uint32_t);
  const bool chunks = !binning_cells && nbin <= NM_B2_MAXBIN;       // (binning by depth-ordered chunks: see k_bin_count2)
  if (chunks) {
    if (K == 0) NM_HIP_CHECK(hipMemsetAsync(t.slab_blk, 0, (2 * NM_GS * NM_GSUB + 64) * sizeof(uint32_t), s));      // (otherwise k_preprocess zeroes them)
    uint32_t* hdr2 = t.slab_blk + 2 * NM_GS * NM_GSUB;
    if (K > 0) {
      NM_LAUNCH(k_preprocess, dim3(nrange), dim3(256), 0, s, k, K, means3D, shs, colors_precomp, opacities, cov3D, radii,
                t.xy, t.depth, t.conop, t.rgb, t.clamped, t.rad, t.zrange, t.recs, t.hdr, t.slab_blk, 2 * NM_GS * NM_GSUB + 64);
      NM_LAUNCH_CHECK();
      NM_LAUNCH(k_slab_hist, dim3(nrange), dim3(256), 0, s, k, K, (const int*)t.rad, (const float2*)t.xy, (const float*)t.depth,
                (const uint2*)t.zrange, nrange, t.slab_blk);
      NM_LAUNCH_CHECK();
      NM_LAUNCH(k_slab_scatter, dim3(nrange), dim3(256), 0, s, k, K, (const int*)t.rad, (const float2*)t.xy, (const float*)t.depth,
                (const uint2*)t.zrange, nrange, (const uint32_t*)t.slab_blk, t.slab_blk + NM_GS * NM_GSUB, t.slab_off, t.gkeys, hdr2);
      NM_LAUNCH_CHECK();
      NM_LAUNCH(k_cell_sort, dim3(NM_GS / 4), dim3(256), 0, s, NM_GS, (const uint32_t*)t.slab_off, t.gkeys, t.gvals, (long long)K,
                (const uint32_t*)hdr2);
      NM_LAUNCH_CHECK();
      static bool attr_set[64] = {};       // (a function attribute belongs to the device it was set on)
      int devid = 0;
      NM_HIP_CHECK(hipGetDevice(&devid));
      if (devid < 0 || devid >= 64 || !attr_set[devid]) {
        NM_HIP_CHECK(hipFuncSetAttribute((const void*)k_bin_count2, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(((size_t)NM_B2_MAXBIN * 9 + 1) * sizeof(uint32_t))));
        if (devid >= 0 && devid < 64) attr_set[devid] = true;
      }
      NM_LAUNCH(k_bin_count2, dim3(t.nchunk), dim3(256), b2_lds, s, k, t.nbx, nbin, (const uint32_t*)hdr2,
                (const unsigned long long*)t.gkeys, (const int*)t.rad, (const float2*)t.xy, (const float4*)t.conop, t.hist, t.log, t.hdr,
                (long long)cap_pairs);
      NM_LAUNCH_CHECK();
    } else {
      NM_HIP_CHECK(hipMemsetAsync(t.hist, 0, (size_t)t.nchunk * nbin * sizeof(uint32_t), s));
    }
    NM_LAUNCH(k_col_sum, dim3(nbin), dim3(256), 0, s, nbin, t.nchunk, (const uint32_t*)t.hist, t.bin_total);
    NM_LAUNCH_CHECK();
    NM_LAUNCH(k_col_scan, dim3(nbin), dim3(256), 0, s, nbin, t.nchunk, (const uint32_t*)t.hist, (const uint32_t*)t.bin_total, t.coff, t.off,
              t.hdr, (long long)cap_pairs);
    NM_LAUNCH_CHECK();
    if (K > 0) {
      NM_LAUNCH(k_bin_fill8, dim3(min(2048, nm_div_up((int)min((int64_t)cap_pairs, (int64_t)K * 64), 256) + 1)), dim3(256), 0, s,
                (const uint32_t*)t.hdr, (const unsigned long long*)t.log, (const uint32_t*)t.coff, nbin,
                (const unsigned long long*)t.gkeys, t.keys, t.vals, (long long)cap_pairs);
      NM_LAUNCH_CHECK();
    }
  } else {
  NM_HIP_CHECK(hipMemsetAsync(t.pad, 0, (size_t)t.ncell * NM_PAD * sizeof(uint32_t), s));
  if (K > 0) {
    NM_LAUNCH(k_preprocess, dim3(nrange), dim3(256), 0, s, k, K, means3D, shs, colors_precomp, opacities, cov3D, radii,
              t.xy, t.depth, t.conop, t.rgb, t.clamped, t.rad, t.zrange, t.recs, t.hdr, (uint32_t*)nullptr, 0);
    NM_LAUNCH_CHECK();
    NM_LAUNCH(k_bin_count, dim3(nrange), dim3(256), 0, s, k, K, t.nbx, (const int*)t.rad, t.xy, t.depth, t.conop, t.zrange, nrange,
              t.pad, t.log, t.hdr, (long long)cap_pairs);
    NM_LAUNCH_CHECK();
  }
  NM_LAUNCH(k_bin_compact, dim3(nbin), dim3(NM_NS), 0, s, nbin, (const uint32_t*)t.pad, t.cnt, t.bin_total, t.hdr);
  NM_LAUNCH_CHECK();
  NM_LAUNCH(k_cell_offsets, dim3(nbin), dim3(NM_NS), 0, s, nbin, (const uint32_t*)t.cnt, (const uint32_t*)t.bin_total, t.off, t.hdr,
            (long long)cap_pairs);
  NM_LAUNCH_CHECK();
  if (K > 0) {
    NM_LAUNCH(k_bin_fill, dim3(min(2048, nm_div_up((int)min((int64_t)cap_pairs, (int64_t)K * 64), 256) + 1)), dim3(256), 0, s,
              (const uint32_t*)t.hdr, (const PairLog*)t.log, (const uint32_t*)t.off, (const float*)t.depth, t.keys, t.vals,
              (long long)cap_pairs);
    NM_LAUNCH_CHECK();
    NM_LAUNCH(k_cell_sort, dim3(min(nm_div_up(t.ncell, 4), 8192)), dim3(256), 0, s, t.ncell, (const uint32_t*)t.off, t.keys, t.vals,
              (long long)cap_pairs, (const uint32_t*)t.hdr);
    NM_LAUNCH_CHECK();
  }
  }
  // the status words go straight to the caller's pinned memory if the device can address it (hipHostMalloc'ed memory can;
  // anything else gets the copy at the end)
  unsigned long long* status_dev = nullptr;
  static const bool status_direct = !(getenv("NM_RASTER_STATUS_DIRECT") && atoi(getenv("NM_RASTER_STATUS_DIRECT")) == 0);
  if (status_host && status_direct) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, status_host, 0) == hipSuccess && dp) status_dev = (unsigned long long*)dp;
    else (void)hipGetLastError();
  }
  NM_LAUNCH(k_split_plan, dim3(1), dim3(1024), 0, s, k, t.nbx, (uint32_t)g_split_busy, (uint32_t)g_split_minseg,
            (unsigned long long)g_split_fwd,
            (const uint32_t*)t.off, (long long)cap_pairs, t.hdr, t.tile_rec,
            t.tile_ns, t.tile_cnt, t.tile_mode, t.work, t.seg_pos, (const uint32_t*)tile_walk, (uint32_t)g_hint_fwd, (uint32_t)g_hint_seg,
            getenv("NM_RASTER_DEBUG") ? 1 : 0, status_dev, status_words < 3 ? status_words : 3);
  NM_LAUNCH_CHECK();
  if (tile_walk) {
    NM_LAUNCH(k_hint_fill, dim3(nm_div_up(k.gx * (k.ty1 - k.ty0), 4)), dim3(256), 0, s, k, t.nbx, (const uint32_t*)t.off,
              (long long)cap_pairs, (const uint32_t*)t.hdr, (const uint32_t*)t.tile_rec, (const uint32_t*)t.tile_ns,
              (const uint32_t*)t.tile_cnt, t.work, t.seg_pos);
    NM_LAUNCH_CHECK();
  }
  const int ntile = k.gx * (k.ty1 - k.ty0);
  // stage 0: whole tiles + first segments of the candidates (which decide how their tile goes on); stage 1: other segments
  // of the tiles that are split.  A view without candidates has no work items and those workgroups leave at once.
  const int stages = (tile_walk && ntile <= 1024 * NM_PLAN_PER) ? 1 : 2;      // (a hinted plan does everything in the first launch)
  for (int stage = 0; stage < stages; ++stage) {
    NM_LAUNCH(k_render, dim3((stage == 0 ? ntile : 0) + t.items), dim3(NM_TPB), 0, s, k, t.nbx, stage, stage == 0 ? ntile : 0,
              t.tile_mode, (const uint32_t*)t.tile_rec, (const uint32_t*)t.tile_ns, (const uint32_t*)t.off,
              (const unsigned long long*)t.keys, (const uint32_t*)t.vals, (long long)cap_pairs, (const uint32_t*)t.hdr,
              (const uint2*)t.work, t.seg_raw, t.seg_last, t.seg_ct, t.seg_pos, (const GRec*)t.recs, t.final_T, t.n_contrib,
              out_color, tile_walk);
    NM_LAUNCH_CHECK();
  }
  // A hinted plan splits the forward pass of a tile only if its planned walk is longer than g_hint_fwd entries; no list is
  // longer than the capacity, so with the threshold at or above it (the frame driver's setting when several views share the
  // chip: every tile walks front to back) the second and third pass have nothing to do and are not launched.
  const bool no_split_fwd = stages == 1 && (long long)g_hint_fwd >= (long long)cap_pairs;
  if (!no_split_fwd) {
  NM_LAUNCH(k_render_fix, dim3(t.items), dim3(NM_TPB), 0, s, k, t.nbx, (const uint32_t*)t.off, (const unsigned long long*)t.keys,
            (const uint32_t*)t.vals, (long long)cap_pairs, (const uint32_t*)t.hdr, (const uint32_t*)t.tile_rec,
            (const uint32_t*)t.tile_ns, t.tile_cnt, (const uint32_t*)t.tile_mode, (const uint2*)t.work, (const float4*)t.seg_raw,
            t.seg_fix, t.seg_ct, t.seg_last, t.seg_pos, (const GRec*)t.recs, t.final_T, t.n_contrib, out_color,
            tile_walk);
  NM_LAUNCH_CHECK();
  NM_LAUNCH(k_render_sum, dim3(ntile), dim3(NM_TPB), 0, s, k, t.nbx, (const uint32_t*)t.off, (const unsigned long long*)t.keys,
            (const uint32_t*)t.vals, (long long)cap_pairs, (const uint32_t*)t.hdr, (const uint32_t*)t.tile_rec,
            (const uint32_t*)t.tile_ns, (const uint32_t*)t.tile_mode, (const float4*)t.seg_fix, t.seg_ct,
            (const uint32_t*)t.seg_last, t.seg_pos, (const GRec*)t.recs, t.final_T, t.n_contrib, out_color, tile_walk);
  NM_LAUNCH_CHECK();
  }
  if (status_host && !status_dev)      // pairs | overflow | items wanted (plan_status), one copy
    NM_HIP_CHECK(hipMemcpyAsync(status_host, t.hdr + 16, sizeof(int64_t) * (size_t)(status_words < 3 ? status_words : 3), hipMemcpyDeviceToHost, s));
  return NM_OK;
}

// Exact (Gaussian, tile) pair count of the view held in `state` (what the reference's duplicateWithKeys emits after the
// conic test).  Synchronises the stream: statistics only.
extern "C" int nm_raster_count_pairs(const nm_raster_cfg* cfg, int32_t K, void* state, int64_t cap_pairs, int64_t* pairs_out,
                                     void* stream) {
  RK k;
  int rc = make_rk(cfg, 0, k);
  if (rc) return rc;
  NM_REQUIRE(K >= 0 && state && pairs_out, "bad arguments");
  State t = carve_state(state, nullptr, k.W, k.H, K, cap_pairs, k.items);
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* total = (unsigned long long*)(t.hdr + 4);
  NM_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(unsigned long long), s));
  if (K > 0) {
    NM_LAUNCH(k_count_pairs, dim3(nm_div_up(K, 256)), dim3(256), 0, s, k, K, (const int*)t.rad, t.xy, t.conop, total);
    NM_LAUNCH_CHECK();
  }
  unsigned long long h = 0;
  NM_HIP_CHECK(hipMemcpyAsync(&h, total, sizeof(h), hipMemcpyDeviceToHost, s));
  NM_HIP_CHECK(hipStreamSynchronize(s));
  *pairs_out = (int64_t)h;
  return NM_OK;
}

// debugging aid: copies the per-cell pair counts of `state` to host memory (ncell_out receives the number of cells)
extern "C" int nm_debug_raster_cells(const nm_raster_cfg* cfg, int32_t K, void* state, int64_t cap_pairs, uint32_t* counts_host,
                                     int32_t max_cells, int32_t* ncell_out, void* stream) {
  RK k;
  int rc = make_rk(cfg, 0, k);
  if (rc) return rc;
  State t = carve_state(state, nullptr, k.W, k.H, K, cap_pairs, k.items);
  *ncell_out = t.ncell;
  NM_HIP_CHECK(hipMemcpyAsync(counts_host, t.cnt, sizeof(uint32_t) * (size_t)(t.ncell < max_cells ? t.ncell : max_cells), hipMemcpyDeviceToHost,
                              (hipStream_t)stream));
  NM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return NM_OK;
}

// debugging aid: copies the per-pixel last-contributor positions (W*H) and the cell offsets (ncell+1) of `state` to host memory
extern "C" int nm_debug_raster_tiles(const nm_raster_cfg* cfg, int32_t K, void* state, int64_t cap_pairs, uint32_t* n_contrib_host,
                                     uint32_t* off_host, int32_t max_cells, int32_t* ncell_out, void* stream) {
  RK k;
  int rc = make_rk(cfg, 0, k);
  if (rc) return rc;
  State t = carve_state(state, nullptr, k.W, k.H, K, cap_pairs, k.items);
  *ncell_out = t.ncell;
  NM_HIP_CHECK(hipMemcpyAsync(n_contrib_host, t.n_contrib, sizeof(uint32_t) * (size_t)k.W * k.H, hipMemcpyDeviceToHost, (hipStream_t)stream));
  NM_HIP_CHECK(hipMemcpyAsync(off_host, t.off, sizeof(uint32_t) * (size_t)((t.ncell < max_cells ? t.ncell : max_cells) + 1), hipMemcpyDeviceToHost,
                              (hipStream_t)stream));
  NM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return NM_OK;
}

extern "C" size_t nm_raster_bwd_workspace(int32_t K) { return al256((size_t)(K > 0 ? K : 1) * NM_NGS * sizeof(float)); }

extern "C" int nm_raster_backward(const nm_raster_cfg* cfg, int32_t K, int32_t m, const float* means3D, const float* shs,
                                  const float* colors_precomp, const float* opacities, const float* cov3D, const void* state,
                                  int64_t cap_pairs, const float* dL_dcolor, float* dL_dmeans3D, float* dL_dmeans2D,
                                  float* dL_dcov3D, float* dL_dopacity, float* dL_dshs, float* dL_dcolors, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  (void)opacities; (void)colors_precomp;
  RK k;
  int rc = make_rk(cfg, m, k);
  if (rc) return rc;
  NM_REQUIRE(K >= 0 && cap_pairs >= 0, "bad arguments");
  if (K == 0) return NM_OK;
  k.K = K;
  NM_REQUIRE(means3D && cov3D && state && dL_dcolor && dL_dmeans3D && workspace, "null pointer");
  if (workspace_bytes < nm_raster_bwd_workspace(K)) { nm_set_error("raster backward workspace too small"); return NM_ERR_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  State t = carve_state((void*)state, nullptr, k.W, k.H, K, cap_pairs, k.items);
  float* acc = (float*)workspace;
  NM_HIP_CHECK(hipMemsetAsync(acc, 0, (size_t)K * NM_NGS * sizeof(float), s));
  const int ntile = k.gx * (k.ty1 - k.ty0);
  // Two pixels per lane (k_render_bwd2): less reverse-compositing work per pixel when several views share the chip (metric
  // frame, each view's adjoint behind its own forward pass: 186.7 -> 188.9 frames/s), 15 % MORE time for a view that has the
  // chip to itself (half the waves per tile: the long tiles' latency counts there).  The caller says which situation it is in
  // (nm_raster_set_reverse_px2; SceneRuntime sets it with its view streams); NM_BWD_PX2 overrides, read per call (tests, A/B).
  const char* px2_env = getenv("NM_BWD_PX2");
  const bool px2 = px2_env ? atoi(px2_env) != 0 : g_bwd_px2 != 0;
  if (px2 && dL_dopacity)
    NM_LAUNCH(k_render_bwd2<true>, dim3(ntile + t.items), dim3(128), 0, s, k, t.nbx, ntile, (const uint32_t*)t.off,
              (const unsigned long long*)t.keys, (const uint32_t*)t.vals, (const uint32_t*)t.hdr, (const uint32_t*)t.tile_rec,
              (const uint32_t*)t.tile_ns, (const uint2*)t.work, (const float4*)t.seg_ct, (const uint32_t*)t.seg_pos, (const GRec*)t.recs,
              t.final_T, t.n_contrib, dL_dcolor, acc);
  else if (px2)
    NM_LAUNCH(k_render_bwd2<false>, dim3(ntile + t.items), dim3(128), 0, s, k, t.nbx, ntile, (const uint32_t*)t.off,
              (const unsigned long long*)t.keys, (const uint32_t*)t.vals, (const uint32_t*)t.hdr, (const uint32_t*)t.tile_rec,
              (const uint32_t*)t.tile_ns, (const uint2*)t.work, (const float4*)t.seg_ct, (const uint32_t*)t.seg_pos, (const GRec*)t.recs,
              t.final_T, t.n_contrib, dL_dcolor, acc);
  else if (dL_dopacity)
    NM_LAUNCH(k_render_bwd<true>, dim3(ntile + t.items), dim3(NM_TPB), 0, s, k, t.nbx, ntile, (const uint32_t*)t.off,
              (const unsigned long long*)t.keys, (const uint32_t*)t.vals, (const uint32_t*)t.hdr, (const uint32_t*)t.tile_rec,
              (const uint32_t*)t.tile_ns, (const uint2*)t.work, (const float4*)t.seg_ct, (const uint32_t*)t.seg_pos, (const GRec*)t.recs,
              t.final_T, t.n_contrib, dL_dcolor, acc);
  else
    NM_LAUNCH(k_render_bwd<false>, dim3(ntile + t.items), dim3(NM_TPB), 0, s, k, t.nbx, ntile, (const uint32_t*)t.off,
              (const unsigned long long*)t.keys, (const uint32_t*)t.vals, (const uint32_t*)t.hdr, (const uint32_t*)t.tile_rec,
              (const uint32_t*)t.tile_ns, (const uint2*)t.work, (const float4*)t.seg_ct, (const uint32_t*)t.seg_pos, (const GRec*)t.recs,
              t.final_T, t.n_contrib, dL_dcolor, acc);
  NM_LAUNCH_CHECK();
  NM_LAUNCH(k_preprocess_bwd, dim3(nm_div_up(K, 256)), dim3(256), 0, s, k, K, means3D, shs, cov3D, (const int*)t.rad, t.clamped, acc,
            dL_dmeans3D, dL_dmeans2D, dL_dcov3D, dL_dopacity, dL_dshs, dL_dcolors, shs ? 1 : 0);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
