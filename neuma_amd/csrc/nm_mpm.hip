// MLS-MPM substep (APIC, quadratic B-spline) for gfx950: p2g / grid_op / g2p and their hand-written
// adjoints.  Behaviour follows /root/reference/modules/nclaw/sim/mpm.py:279-498 (see SURVEY.md App. A
// for the adjoint derivation); the structure does not:
//
//   * the grid lives in HBM as 4x4x4-node blocks of float4 {mv.xyz, m} / {v.xyz, -} (1 KiB per block),
//     and only blocks touched by a particle stencil are ever cleared or updated (active-block list built
//     by p2g with an epoch flag per block) — the reference sweeps all G^3 cells 12 times per step;
//   * scatters (p2g, g2p-adjoint) accumulate in a per-wave LDS tile covering the bounding box of the wave's
//     particles and flush each touched node once with global atomics.  ds_add_f32 is NOT used: measured on
//     MI355X it retires ~0.33 lanes/clk/CU (tools/ubench_atomics.hip), 20x slower than a plain LDS
//     read-modify-write.  Instead a workgroup is ONE wave that owns its tile; for a fixed stencil offset two
//     lanes collide only if they share a base cell, so lanes elect one owner per base cell (LDS ticket) and
//     the owners do plain ds_read_b128 / add / ds_write_b128; losers retry in the next round.  Lanes take
//     particles with stride PB, so cell-sorted inputs give ~1 round.  A wave whose particles are too spread
//     out falls back to direct global atomics: particle order affects speed only, never results beyond
//     fp32 summation order;
//   * stencil nodes with an index >= G (reference: out-of-bounds access when x > 1-1.5dx) land in
//     padding blocks whose velocity is defined as zero.
#include "nm_common.h"
#include <stdlib.h>


struct MpmK {
  int G, Gp, nb;
  float dt, dx, inv_dx, eps;
  float gdt[3];
  int bound, bc;
  int dbg;  // NM_DBG experiment switches (0 in production)
  long long* dbg_buf;
};

struct nm_mpm {
  nm_mpm_cfg cfg;
  MpmK k;
  int nblocks;
  float4* gm;  // {mv.xyz, m}
  float4* gv;  // {v.xyz, 0}
  float4* gg;  // adjoint scratch: {vbar.xyz,0} -> {mvbar.xyz, mbar}
  int* flags;
  int* list[2];
  int* count;  // [2] counters + [2] stats
  int cur;
  int epoch;
};

__device__ __forceinline__ int node_addr(int i, int j, int k, int nb) {
  return ((((i >> 2) * nb + (j >> 2)) * nb + (k >> 2)) << 6) | ((i & 3) << 4) | ((j & 3) << 2) | (k & 3);
}

struct Stencil {
  int b[3];
  float f[3];
  float w[3][3];   // w[axis][i]
  float dw[3][3];  // d w[axis][i] / d f
};

__device__ __forceinline__ void make_stencil(const MpmK& K, const float* __restrict__ xp, Stencil& s) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float px = xp[a] * K.inv_dx;
    int b = (int)(px - 0.5f);  // C cast: truncation toward zero (mpm.py:337-339)
    b = max(0, min(b, K.Gp - 3));
    float f = px - (float)b;
    s.b[a] = b;
    s.f[a] = f;
    float wa = 1.5f - f, wb = f - 1.0f, wc = f - 0.5f;
    s.w[a][0] = wa * wa * 0.5f;
    s.w[a][1] = 0.75f - wb * wb;
    s.w[a][2] = wc * wc * 0.5f;
    s.dw[a][0] = -wa;
    s.dw[a][1] = -2.f * wb;
    s.dw[a][2] = wc;
  }
}

__device__ __forceinline__ float sel3(const float* a, int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }

__device__ __forceinline__ void mark_block(int b, int* __restrict__ flags, int* __restrict__ list,
                                           int* __restrict__ count, int epoch) {
  if (flags[b] != epoch) {
    if (atomicExch(&flags[b], epoch) != epoch) {
      int pos = atomicAdd(count, 1);
      list[pos] = b;
    }
  }
}

// ---------------------------------------------------------------- per-wave LDS tile for scatters
#define NM_WT_CAP 1024   // nodes per wave tile: 16 KiB of float4 + 4 KiB of owner tickets
#define NM_WT_PB 8       // consecutive particles per lane (workgroup = one wave = 512 particles)
#define NM_WT_BOX 10     // edge of the fixed box used when a chunk's bounding box exceeds the tile (10^3 nodes)
#define NM_WT_MAXPASS 12 // boxes tried per chunk before the leftovers go to direct global atomics

struct TileGeom {
  int o[3];
  int n[3];
  int vol;
  bool use;
};

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ void base_cell(const MpmK& K, const float* __restrict__ xp, int* bb) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int q = (int)(xp[a] * K.inv_dx - 0.5f);
    bb[a] = max(0, min(q, K.Gp - 3));
  }
}
__device__ __forceinline__ void tile_zero(const TileGeom& g, float4* s_tile) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = threadIdx.x; i < g.vol; i += 64) s_tile[i] = z;
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ TileGeom tile_box(const MpmK& K, const int* anchor) {
  TileGeom g;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    g.o[a] = max(0, min(anchor[a] - 2, K.Gp - NM_WT_BOX));
    g.n[a] = NM_WT_BOX;
  }
  g.vol = NM_WT_BOX * NM_WT_BOX * NM_WT_BOX;
  g.use = true;
  return g;
}
__device__ __forceinline__ bool tile_holds(const TileGeom& g, const int* b) {
  return b[0] >= g.o[0] && b[0] + 3 <= g.o[0] + g.n[0] && b[1] >= g.o[1] && b[1] + 3 <= g.o[1] + g.n[1] && b[2] >= g.o[2] &&
         b[2] + 3 <= g.o[2] + g.n[2];
}

// flush the tile: one global atomic set per touched node; mark the grid blocks the tile overlaps
template <int NCH>
__device__ __forceinline__ void tile_flush(const TileGeom& g, const float4* s_tile, float4* __restrict__ grid, int nb, int* flags,
                                           int* list, int* count, int epoch) {
  __builtin_amdgcn_wave_barrier();
  const int lane = threadIdx.x;
  const int nyz = g.n[1] * g.n[2];
  for (int idx = lane; idx < g.vol; idx += 64) {
    float4 t = s_tile[idx];
    if (t.x != 0.f || t.y != 0.f || t.z != 0.f || t.w != 0.f) {
      int i = idx / nyz, r = idx - i * nyz;
      int j = r / g.n[2], k = r - j * g.n[2];
      float* dst = (float*)&grid[node_addr(g.o[0] + i, g.o[1] + j, g.o[2] + k, nb)];
      unsafeAtomicAdd(dst, t.x);
      unsafeAtomicAdd(dst + 1, t.y);
      unsafeAtomicAdd(dst + 2, t.z);
      if (NCH == 4) unsafeAtomicAdd(dst + 3, t.w);
    }
  }
  if (flags) {
    int b0 = g.o[0] >> 2, b1 = g.o[1] >> 2, b2 = g.o[2] >> 2;
    int m0 = ((g.o[0] + g.n[0] - 1) >> 2) - b0 + 1, m1 = ((g.o[1] + g.n[1] - 1) >> 2) - b1 + 1,
        m2 = ((g.o[2] + g.n[2] - 1) >> 2) - b2 + 1;
    for (int t = lane; t < m0 * m1 * m2; t += 64) {
      int i = t / (m1 * m2), r = t - i * (m1 * m2);
      int j = r / m2, k = r - j * m2;
      mark_block(((b0 + i) * nb + (b1 + j)) * nb + (b2 + k), flags, list, count, epoch);
    }
  }
}

#define NM_WT_CHUNK (64 * NM_WT_PB)
#define NM_WT_REC 16  // floats per staged particle record: x(3) + payload(13)

struct ScatterLds {
  float4 tile[NM_WT_CAP];
  int cnt[NM_WT_CAP];                 // particles per stencil origin (tile-local cell), then exclusive offsets
  float rec[NM_WT_CHUNK * NM_WT_REC];
  int key[NM_WT_CHUNK];               // global origin key per staged particle (-1: disabled / not staged)
  short rank[NM_WT_CHUNK];            // arrival order inside its cell
  short order[NM_WT_CHUNK];           // particle slots sorted by cell
  short run_cell[NM_WT_CAP];          // compacted list of non-empty cells
};

__device__ __forceinline__ int wave_excl_scan_i(int v, int lane, int& total) {
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  total = __shfl(x, 63, 64);
  return x - v;
}

// Scatter of one chunk of NM_WT_CHUNK consecutive particles into `grid` by one wave.
//   stage(p, rec)              loads particle p and writes x(3) + payload(13) to rec[16]      (coalesced phase)
//   contrib(st, rec, i, j, k)  contribution of a staged particle to its stencil node (i,j,k)
// A: every particle is staged once (coalesced global loads, all in flight together).
// B: the chunk is counting-sorted in LDS by stencil origin (tile-local cell): integer LDS atomics give each
//    particle its rank inside its cell, a wave scan gives the cell offsets.  So whatever order the caller keeps
//    its particles in (sorted at load time, gone stale since), every non-empty cell becomes exactly one run.
// C: lane r sums the 27 float4 contributions of run r in registers and adds them to the tile with plain
//    ds_read_b128 / ds_write_b128 — two lanes never share a cell, and for a fixed stencil offset distinct cells
//    hit distinct nodes, so no atomics and no retries are needed.
// D: the tile is flushed with one global atomic set per touched node.
// Chunks whose bounding box exceeds the tile are processed box by box (anchored at the first pending particle);
// after NM_WT_MAXPASS boxes the leftovers use per-particle global atomics.
template <int NCH, class StageF, class ContribF>
__device__ __forceinline__ void wave_scatter(const MpmK& K, int n, const int* __restrict__ enabled, const float* __restrict__ x,
                                             float4* __restrict__ grid, int* flags, int* list, int* count, int epoch,
                                             ScatterLds& L, StageF stage, ContribF contrib) {
  const int lane = threadIdx.x;
  const int c0 = blockIdx.x * NM_WT_CHUNK;
  const int GG = K.Gp * K.Gp;
  long long tm[6] = {0, 0, 0, 0, 0, 0};
  long long t0 = K.dbg_buf ? clock64() : 0;
#define NM_TICK(i) if (K.dbg_buf) { long long t1 = clock64(); tm[i] += t1 - t0; t0 = t1; }
  // ---- A: stage records + origin keys; bounding box of the origins
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
#pragma unroll
  for (int it = 0; it < NM_WT_PB; ++it) {
    const int t = it * 64 + lane, p = c0 + t;
    int key = -1;
    if (p < n && enabled[p] != 0) {
      float rec[NM_WT_REC];
      stage(p, rec);
      int bb[3];
      base_cell(K, rec, bb);
      key = (bb[0] * K.Gp + bb[1]) * K.Gp + bb[2];
#pragma unroll
      for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], bb[a]); hi[a] = max(hi[a], bb[a]); }
      float4* dst = reinterpret_cast<float4*>(&L.rec[t * NM_WT_REC]);
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[q] = make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
    }
    L.key[t] = key;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { lo[a] = wave_min_i(lo[a]); hi[a] = wave_max_i(hi[a]); }
  if (lo[0] == 0x7fffffff) return;  // nothing enabled in this chunk
  TileGeom g;
#pragma unroll
  for (int a = 0; a < 3; ++a) { g.o[a] = lo[a]; g.n[a] = hi[a] - lo[a] + 3; }
  g.vol = g.n[0] * g.n[1] * g.n[2];
  const bool single = g.vol <= NM_WT_CAP;
  __builtin_amdgcn_wave_barrier();
  NM_TICK(0)

  for (int pass = 0; pass <= NM_WT_MAXPASS; ++pass) {
    if (!single) {
      if (pass == NM_WT_MAXPASS) {   // last resort: per-particle global atomics for what is still pending
        for (int it = 0; it < NM_WT_PB; ++it) {
          const int t = it * 64 + lane;
          const int key = L.key[t];
          if (key < 0) continue;
          float rec[NM_WT_REC];
          const float4* src = reinterpret_cast<const float4*>(&L.rec[t * NM_WT_REC]);
#pragma unroll
          for (int q = 0; q < 4; ++q) { float4 v4 = src[q]; rec[4 * q] = v4.x; rec[4 * q + 1] = v4.y; rec[4 * q + 2] = v4.z; rec[4 * q + 3] = v4.w; }
          Stencil sp;
          make_stencil(K, rec, sp);
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int kk = 0; kk < 3; ++kk) {
                const float4 c = contrib(sp, rec, i, j, kk);
                float* dst = (float*)&grid[node_addr(sp.b[0] + i, sp.b[1] + j, sp.b[2] + kk, K.nb)];
                unsafeAtomicAdd(dst, c.x);
                unsafeAtomicAdd(dst + 1, c.y);
                unsafeAtomicAdd(dst + 2, c.z);
                if (NCH == 4) unsafeAtomicAdd(dst + 3, c.w);
              }
          if (flags) {
            for (int i = sp.b[0] >> 2; i <= (sp.b[0] + 2) >> 2; ++i)
              for (int j = sp.b[1] >> 2; j <= (sp.b[1] + 2) >> 2; ++j)
                for (int kk = sp.b[2] >> 2; kk <= (sp.b[2] + 2) >> 2; ++kk)
                  mark_block((i * K.nb + j) * K.nb + kk, flags, list, count, epoch);
          }
        }
        break;
      }
      // anchor a box at the first particle still pending
      int first = 0x7fffffff;
      for (int it = 0; it < NM_WT_PB; ++it)
        if (L.key[it * 64 + lane] >= 0) { first = it * 64 + lane; break; }
      first = wave_min_i(first);
      if (first == 0x7fffffff) break;
      const int key = L.key[first];
      int anchor[3] = {key / GG, (key / K.Gp) % K.Gp, key % K.Gp};
      g = tile_box(K, anchor);
    }
    // ---- B: counting sort of the pass's particles by tile-local cell
    for (int i = lane; i < g.vol; i += 64) { L.tile[i] = make_float4(0.f, 0.f, 0.f, 0.f); L.cnt[i] = 0; }
    __builtin_amdgcn_wave_barrier();
    int myci[NM_WT_PB];
#pragma unroll
    for (int it = 0; it < NM_WT_PB; ++it) {
      const int t = it * 64 + lane;
      const int key = L.key[t];
      myci[it] = -1;
      if (key >= 0) {
        int bb[3] = {key / GG, (key / K.Gp) % K.Gp, key % K.Gp};
        if (single || tile_holds(g, bb)) {
          const int ci = ((bb[0] - g.o[0]) * g.n[1] + (bb[1] - g.o[1])) * g.n[2] + (bb[2] - g.o[2]);
          myci[it] = ci;
          L.rank[t] = (short)atomicAdd(&L.cnt[ci], 1);
          L.key[t] = -1;   // consumed by this pass
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // exclusive scan of cnt over the tile cells (each lane owns a contiguous slice) + list of non-empty cells
    const int per = (g.vol + 63) >> 6;
    int nruns, tot;
    {
      int sum = 0, nz = 0;
      for (int i = 0; i < per; ++i) {
        int c = lane * per + i;
        if (c < g.vol) { int v = L.cnt[c]; sum += v; nz += v > 0; }
      }
      int off = wave_excl_scan_i(sum, lane, tot);
      int roff = wave_excl_scan_i(nz, lane, nruns);
      for (int i = 0; i < per; ++i) {
        int c = lane * per + i;
        if (c < g.vol) {
          int v = L.cnt[c];
          L.cnt[c] = off;
          off += v;
          if (v > 0) L.run_cell[roff++] = (short)c;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NM_WT_PB; ++it)
      if (myci[it] >= 0) L.order[L.cnt[myci[it]] + L.rank[it * 64 + lane]] = (short)(it * 64 + lane);
    __builtin_amdgcn_wave_barrier();
    NM_TICK(1)
    // ---- C: one lane per non-empty cell
    const int nyz = g.n[1] * g.n[2];
#pragma unroll 1
    for (int r0 = 0; r0 < nruns; r0 += 64) {
      const int r = r0 + lane;
      const bool has = r < nruns;
      float4 acc[27];
      int ci = 0;
      if (has) {
        ci = L.run_cell[r];
        const int start = L.cnt[ci];
        // cnt[] holds exclusive offsets: this run ends where the next non-empty cell starts (the last one at `tot`)
        const int end = (r + 1 < nruns) ? L.cnt[L.run_cell[r + 1]] : tot;
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s_ = start; s_ < end; ++s_) {
          const int t = L.order[s_];
          float rec[NM_WT_REC];
          const float4* src = reinterpret_cast<const float4*>(&L.rec[t * NM_WT_REC]);
#pragma unroll
          for (int q = 0; q < 4; ++q) { float4 v4 = src[q]; rec[4 * q] = v4.x; rec[4 * q + 1] = v4.y; rec[4 * q + 2] = v4.z; rec[4 * q + 3] = v4.w; }
          Stencil sp;
          make_stencil(K, rec, sp);
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int kk = 0; kk < 3; ++kk) {
                float4 c = contrib(sp, rec, i, j, kk);
                float4& a4 = acc[(i * 3 + j) * 3 + kk];
                a4.x += c.x; a4.y += c.y; a4.z += c.z; a4.w += c.w;
              }
        }
      }
      NM_TICK(2)
      if (has) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
              const int idx = ci + (i * g.n[1] + j) * g.n[2] + kk;
              const float4 c = acc[(i * 3 + j) * 3 + kk];
              float4 t4 = L.tile[idx];
              t4.x += c.x; t4.y += c.y; t4.z += c.z;
              if (NCH == 4) t4.w += c.w;
              L.tile[idx] = t4;
              // keep the wave-level order of LDS accesses between offsets: another lane's next offset may be this
              // lane's current node
              asm volatile("" ::: "memory");
            }
      }
      __builtin_amdgcn_wave_barrier();
      NM_TICK(3)
    }
    (void)nyz;
    tile_flush<NCH>(g, L.tile, grid, K.nb, flags, list, count, epoch);
    NM_TICK(4)
    if (single) break;
  }
  if (K.dbg_buf && NCH == 4 && lane == 0 && blockIdx.x < 512) {
    for (int i = 0; i < 5; ++i) K.dbg_buf[4096 * 2 + blockIdx.x * 8 + i] = tm[i];
    K.dbg_buf[4096 * 2 + blockIdx.x * 8 + 7] = single ? 1 : 0;
  }
}

// ---------------------------------------------------------------- kernels
// zero the blocks the previous substep touched (all three node arrays) and reset the counter p2g will fill
__global__ void __launch_bounds__(256) k_clear(float4* __restrict__ gm, float4* __restrict__ gv, float4* __restrict__ gg,
                                               const int* __restrict__ list, const int* __restrict__ count_prev,
                                               int* __restrict__ count_cur) {
  const int cnt = *count_prev;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int li = blockIdx.x * 4 + wave; li < cnt; li += gridDim.x * 4) {
    int node = (list[li] << 6) + lane;
    gm[node] = z;
    gv[node] = z;
    gg[node] = z;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *count_cur = 0;
}

// mpm.py:321-371.  One wave per workgroup and per chunk of NM_WT_CHUNK consecutive particles.
// staged record: x(3) | mom(3) pm(1) | A(9)
__global__ void __launch_bounds__(64) k_p2g(MpmK K, int n, const float* __restrict__ vol, const float* __restrict__ rho,
                                            const int* __restrict__ enabled, const float* __restrict__ x,
                                            const float* __restrict__ v, const float* __restrict__ C,
                                            const float* __restrict__ S, float4* __restrict__ gm, int* flags, int* list,
                                            int* count, int epoch) {
  __shared__ ScatterLds L;
  const long long t_start = K.dbg_buf ? clock64() : 0;
  auto stage = [&](int p, float* rec) {
    float vl = vol[p];
    float pm = vl * rho[p];
    float ks = -K.dt * vl * 4.0f * K.inv_dx * K.inv_dx;  // mpm.py:357
    M3 Sp = m3_load(S + 9 * p), Cp = m3_load(C + 9 * p);
#pragma unroll
    for (int a = 0; a < 3; ++a) { rec[a] = x[3 * p + a]; rec[3 + a] = pm * v[3 * p + a]; }
    rec[6] = pm;
#pragma unroll
    for (int i = 0; i < 9; ++i) rec[7 + i] = ks * Sp.m[i] + pm * Cp.m[i];
  };
  auto contrib = [&](const Stencil& st, const float* rec, int i, int j, int k) -> float4 {
    float d0 = ((float)i - st.f[0]) * K.dx, d1 = ((float)j - st.f[1]) * K.dx, d2 = ((float)k - st.f[2]) * K.dx;
    float w = st.w[0][i] * st.w[1][j] * st.w[2][k];
    return make_float4(w * (rec[3] + rec[7] * d0 + rec[8] * d1 + rec[9] * d2),
                       w * (rec[4] + rec[10] * d0 + rec[11] * d1 + rec[12] * d2),
                       w * (rec[5] + rec[13] * d0 + rec[14] * d1 + rec[15] * d2), w * rec[6]);
  };
  wave_scatter<4>(K, n, enabled, x, gm, flags, list, count, epoch, L, stage, contrib);
  if (K.dbg_buf && threadIdx.x == 0 && blockIdx.x < 4096) K.dbg_buf[blockIdx.x * 4] = clock64() - t_start;
}

__device__ __forceinline__ void block_coords(int b, int nb, int lane, int& i, int& j, int& k) {
  int bi = b / (nb * nb), r = b - bi * nb * nb;
  int bj = r / nb, bk = r - bj * nb;
  i = (bi << 2) | (lane >> 4);
  j = (bj << 2) | ((lane >> 2) & 3);
  k = (bk << 2) | (lane & 3);
}

// velocity before / after the boundary condition; returns the per-component pass mask
__device__ __forceinline__ void grid_velocity(const MpmK& K, int i, int j, int k, const float4& a, float u[3], float mask[3]) {
  if (a.w > 0.f) {  // mpm.py:382-385 / 411-414
    float inv = 1.f / (a.w + K.eps);
    u[0] = a.x * inv + K.gdt[0];
    u[1] = a.y * inv + K.gdt[1];
    u[2] = a.z * inv + K.gdt[2];
  } else {
    u[0] = K.gdt[0]; u[1] = K.gdt[1]; u[2] = K.gdt[2];
  }
  const int idx[3] = {i, j, k};
  bool hit[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
    hit[c] = (idx[c] < K.bound && u[c] < 0.f) || (idx[c] >= K.G - K.bound && u[c] > 0.f);
  if (K.bc == 0) {  // noslip: any hit zeroes the whole vector (sequential tests, mpm.py:416-427)
    float m = (hit[0] || hit[1] || hit[2]) ? 0.f : 1.f;
    mask[0] = mask[1] = mask[2] = m;
  } else {          // freeslip: only that component (mpm.py:387-398)
#pragma unroll
    for (int c = 0; c < 3; ++c) mask[c] = hit[c] ? 0.f : 1.f;
  }
}

// mpm.py:373-429 on the active blocks only
__global__ void __launch_bounds__(256) k_grid_op(MpmK K, const float4* __restrict__ gm, float4* __restrict__ gv,
                                                 const int* __restrict__ list, const int* __restrict__ count) {
  const int cnt = *count;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int li = blockIdx.x * 4 + wave; li < cnt; li += gridDim.x * 4) {
    int b = list[li];
    int i, j, k;
    block_coords(b, K.nb, lane, i, j, k);
    int node = (b << 6) + lane;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < K.G && j < K.G && k < K.G) {
      float u[3], mk[3];
      grid_velocity(K, i, j, k, gm[node], u, mk);
      out.x = u[0] * mk[0]; out.y = u[1] * mk[1]; out.z = u[2] * mk[2];
    }
    gv[node] = out;
  }
}

// adjoint of grid_op: gg {vbar} -> {mvbar, mbar}
__global__ void __launch_bounds__(256) k_grid_op_bwd(MpmK K, const float4* __restrict__ gm, float4* __restrict__ gg,
                                                     const int* __restrict__ list, const int* __restrict__ count) {
  const int cnt = *count;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int li = blockIdx.x * 4 + wave; li < cnt; li += gridDim.x * 4) {
    int b = list[li];
    int i, j, k;
    block_coords(b, K.nb, lane, i, j, k);
    int node = (b << 6) + lane;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < K.G && j < K.G && k < K.G) {
      float4 a = gm[node];
      if (a.w > 0.f) {
        float u[3], mk[3];
        grid_velocity(K, i, j, k, a, u, mk);
        float4 gb = gg[node];
        float inv = 1.f / (a.w + K.eps);
        float ux = gb.x * mk[0], uy = gb.y * mk[1], uz = gb.z * mk[2];
        out.x = ux * inv; out.y = uy * inv; out.z = uz * inv;
        out.w = -(ux * a.x + uy * a.y + uz * a.z) * inv * inv;
      }
    }
    gg[node] = out;
  }
}

// mpm.py:432-498
__global__ void __launch_bounds__(256, 4) k_g2p(MpmK K, int n, const float* __restrict__ clip, const int* __restrict__ enabled,
                                             const float* x, const float* v, const float* C, const float* F,
                                             const float4* __restrict__ gv, float* xn, float* vn, float* Cn, float* Fn) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  if (enabled[p] == 0) {  // reference skips the particle (mpm.py:443-444); pass the state through
    if (xn != x) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { xn[3 * p + a] = x[3 * p + a]; vn[3 * p + a] = v[3 * p + a]; }
#pragma unroll
      for (int a = 0; a < 9; ++a) { Cn[9 * p + a] = C[9 * p + a]; Fn[9 * p + a] = F[9 * p + a]; }
    }
    return;
  }
  float xp[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
  Stencil st;
  make_stencil(K, xp, st);
  float nv[3] = {0.f, 0.f, 0.f};
  M3 nC = m3_zero();
  const float kap = 4.0f * K.inv_dx * K.inv_dx;
#pragma unroll 1
  for (int i = 0; i < 3; ++i) {  // rolled: nine gathers in flight per trip keeps the kernel at 4+ waves/SIMD
    float d0 = ((float)i - st.f[0]) * K.dx;
    const float w0i = sel3(st.w[0], i);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float d1 = ((float)j - st.f[1]) * K.dx;
      float wij = w0i * st.w[1][j];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float d2 = ((float)k - st.f[2]) * K.dx;
        float w = wij * st.w[2][k];
        float4 g = gv[node_addr(st.b[0] + i, st.b[1] + j, st.b[2] + k, K.nb)];
        nv[0] += w * g.x; nv[1] += w * g.y; nv[2] += w * g.z;
        float kw = kap * w;  // mpm.py:479: (4 w inv_dx^2) outer(v, dpos)
        nC.m[0] += kw * g.x * d0; nC.m[1] += kw * g.x * d1; nC.m[2] += kw * g.x * d2;
        nC.m[3] += kw * g.y * d0; nC.m[4] += kw * g.y * d1; nC.m[5] += kw * g.y * d2;
        nC.m[6] += kw * g.z * d0; nC.m[7] += kw * g.z * d1; nC.m[8] += kw * g.z * d2;
      }
    }
  }
  M3 Fp = m3_load(F + 9 * p);
  M3 T = nC;
#pragma unroll
  for (int i = 0; i < 9; ++i) T.m[i] *= K.dt;
  T.m[0] += 1.f; T.m[4] += 1.f; T.m[8] += 1.f;
  M3 Fo = m3_mul(T, Fp);  // mpm.py:489
  float bnd = clip[p] * K.dx;
  float lo = 0.0f + bnd, hi = 1.0f - bnd;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float t = xp[a] + K.dt * nv[a];
    xn[3 * p + a] = fminf(fmaxf(t, lo), hi);  // wp.clamp, mpm.py:491-497
    vn[3 * p + a] = nv[a];
  }
  m3_store(Cn + 9 * p, nC);
  m3_store(Fn + 9 * p, Fo);
}

// adjoint of g2p: writes gx (direct part), gF; scatters vbar into gg (same wave-tile scheme as p2g)
struct G2pBwdP {  // per-particle quantities of the g2p adjoint
  Stencil st;
  float vt[3], xbar[3];
  M3 Ct, Fbar;
};
__device__ __forceinline__ bool g2p_bwd_particle(const MpmK& K, int n, int p, const float* __restrict__ clip,
                                                 const int* __restrict__ enabled, const float* __restrict__ x,
                                                 const float* __restrict__ F, const float* __restrict__ vnext,
                                                 const float* __restrict__ Cnext, const float* __restrict__ gxn,
                                                 const float* __restrict__ gvn, const float* __restrict__ gCn,
                                                 const float* __restrict__ gFn, G2pBwdP& q) {
  const bool active = p < n && enabled[p] != 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) { q.vt[a] = 0.f; q.xbar[a] = 0.f; }
  q.Ct = m3_zero();
  q.Fbar = m3_zero();
  if (active) {
    float xp[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
    make_stencil(K, xp, q.st);
    float bnd = clip[p] * K.dx;
    float lo = 0.0f + bnd, hi = 1.0f - bnd;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float t = xp[a] + K.dt * vnext[3 * p + a];
      float xe = (t >= lo && t <= hi) ? gxn[3 * p + a] : 0.f;  // clamp passes the gradient only inside
      q.xbar[a] = xe;
      q.vt[a] = gvn[3 * p + a] + K.dt * xe;
    }
    M3 Fp = m3_load(F + 9 * p), gFp = m3_load(gFn + 9 * p), Cn = m3_load(Cnext + 9 * p), gCp = m3_load(gCn + 9 * p);
    M3 T = Cn;
#pragma unroll
    for (int i = 0; i < 9; ++i) T.m[i] *= K.dt;
    T.m[0] += 1.f; T.m[4] += 1.f; T.m[8] += 1.f;
    q.Fbar = m3_mul_tn(T, gFp);           // (I + dt C')^T Fbar'
    M3 FF = m3_mul_nt(gFp, Fp);           // Fbar' F^T
#pragma unroll
    for (int i = 0; i < 9; ++i) q.Ct.m[i] = gCp.m[i] + K.dt * FF.m[i];
  } else {
    q.st.b[0] = q.st.b[1] = q.st.b[2] = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      q.st.f[a] = 0.f;
      q.st.w[a][0] = q.st.w[a][1] = q.st.w[a][2] = 0.f;
      q.st.dw[a][0] = q.st.dw[a][1] = q.st.dw[a][2] = 0.f;
    }
  }
  return active;
}

__global__ void __launch_bounds__(64) k_g2p_bwd(MpmK K, int n, const float* __restrict__ clip, const int* __restrict__ enabled,
                                                const float* __restrict__ x, const float* __restrict__ F,
                                                const float* __restrict__ vnext, const float* __restrict__ Cnext,
                                                const float* __restrict__ gxn, const float* __restrict__ gvn,
                                                const float* __restrict__ gCn, const float* __restrict__ gFn,
                                                const float4* __restrict__ gv, float4* __restrict__ gg,
                                                float* __restrict__ gx, float* __restrict__ gF) {
  __shared__ ScatterLds L;
  const float kap = 4.0f * K.inv_dx * K.inv_dx;
  // (1) per-particle outputs: gF and gx (direct + through weights/dpos, gathering the forward grid velocity)
#pragma unroll 1
  for (int b = 0; b < NM_WT_PB; ++b) {
    const int p = blockIdx.x * NM_WT_CHUNK + b * 64 + threadIdx.x;
    G2pBwdP q;
    const bool active = g2p_bwd_particle(K, n, p, clip, enabled, x, F, vnext, Cnext, gxn, gvn, gCn, gFn, q);
    if (active) {
#pragma unroll 1
      for (int i = 0; i < 3; ++i) {
        float d0 = ((float)i - q.st.f[0]) * K.dx;
        const float w0i = sel3(q.st.w[0], i), dw0i = sel3(q.st.dw[0], i);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float d1 = ((float)j - q.st.f[1]) * K.dx;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float d2 = ((float)k - q.st.f[2]) * K.dx;
            float w = w0i * q.st.w[1][j] * q.st.w[2][k];
            float4 gn = gv[node_addr(q.st.b[0] + i, q.st.b[1] + j, q.st.b[2] + k, K.nb)];
            const M3& Ct = q.Ct;
            float c0_ = Ct.m[0] * d0 + Ct.m[1] * d1 + Ct.m[2] * d2;
            float c1_ = Ct.m[3] * d0 + Ct.m[4] * d1 + Ct.m[5] * d2;
            float c2_ = Ct.m[6] * d0 + Ct.m[7] * d1 + Ct.m[8] * d2;
            float kw = kap * w;
            float dLdw = q.vt[0] * gn.x + q.vt[1] * gn.y + q.vt[2] * gn.z + kap * (gn.x * c0_ + gn.y * c1_ + gn.z * c2_);
            float gw0 = dw0i * q.st.w[1][j] * q.st.w[2][k] * K.inv_dx;
            float gw1 = w0i * q.st.dw[1][j] * q.st.w[2][k] * K.inv_dx;
            float gw2 = w0i * q.st.w[1][j] * q.st.dw[2][k] * K.inv_dx;
            float t0 = Ct.m[0] * gn.x + Ct.m[3] * gn.y + Ct.m[6] * gn.z;
            float t1 = Ct.m[1] * gn.x + Ct.m[4] * gn.y + Ct.m[7] * gn.z;
            float t2 = Ct.m[2] * gn.x + Ct.m[5] * gn.y + Ct.m[8] * gn.z;
            q.xbar[0] += dLdw * gw0 - kw * t0;
            q.xbar[1] += dLdw * gw1 - kw * t1;
            q.xbar[2] += dLdw * gw2 - kw * t2;
          }
        }
      }
    }
    if (p < n) {
#pragma unroll
      for (int a = 0; a < 3; ++a) gx[3 * p + a] = q.xbar[a];
      m3_store(gF + 9 * p, q.Fbar);
    }
  }
  // (2) scatter of the node-velocity adjoint; staged record: x(3) | vt(3) - | Ct(9)
  auto stage = [&](int p, float* rec) {
    G2pBwdP q;
    g2p_bwd_particle(K, n, p, clip, enabled, x, F, vnext, Cnext, gxn, gvn, gCn, gFn, q);
#pragma unroll
    for (int a = 0; a < 3; ++a) { rec[a] = x[3 * p + a]; rec[3 + a] = q.vt[a]; }
    rec[6] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) rec[7 + i] = q.Ct.m[i];
  };
  auto contrib = [&](const Stencil& st, const float* rec, int i, int j, int k) -> float4 {
    float d0 = ((float)i - st.f[0]) * K.dx, d1 = ((float)j - st.f[1]) * K.dx, d2 = ((float)k - st.f[2]) * K.dx;
    float w = st.w[0][i] * st.w[1][j] * st.w[2][k];
    float kw = kap * w;
    float c0_ = rec[7] * d0 + rec[8] * d1 + rec[9] * d2;
    float c1_ = rec[10] * d0 + rec[11] * d1 + rec[12] * d2;
    float c2_ = rec[13] * d0 + rec[14] * d1 + rec[15] * d2;
    return make_float4(w * rec[3] + kw * c0_, w * rec[4] + kw * c1_, w * rec[5] + kw * c2_, 0.f);
  };
  wave_scatter<3>(K, n, enabled, x, gg, nullptr, nullptr, nullptr, 0, L, stage, contrib);
}

// adjoint of p2g: gathers {mvbar, mbar}; writes gv, gC, gS and adds to gx
__global__ void __launch_bounds__(256, 2) k_p2g_bwd(MpmK K, int n, const float* __restrict__ vol, const float* __restrict__ rho,
                                                 const int* __restrict__ enabled, const float* __restrict__ x,
                                                 const float* __restrict__ v, const float* __restrict__ C,
                                                 const float* __restrict__ S, const float4* __restrict__ gg,
                                                 float* __restrict__ gx, float* __restrict__ gvp, float* __restrict__ gC,
                                                 float* __restrict__ gS) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  if (enabled[p] == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) gvp[3 * p + a] = 0.f;
    m3_store(gC + 9 * p, m3_zero());
    m3_store(gS + 9 * p, m3_zero());
    return;
  }
  float xp[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
  Stencil st;
  make_stencil(K, xp, st);
  float vl = vol[p];
  float pm = vl * rho[p];
  float ks = -K.dt * vl * 4.0f * K.inv_dx * K.inv_dx;
  M3 Sp = m3_load(S + 9 * p), Cp = m3_load(C + 9 * p), A;
#pragma unroll
  for (int i = 0; i < 9; ++i) A.m[i] = ks * Sp.m[i] + pm * Cp.m[i];
  float mom[3] = {pm * v[3 * p], pm * v[3 * p + 1], pm * v[3 * p + 2]};
  float vb[3] = {0.f, 0.f, 0.f}, xb[3] = {0.f, 0.f, 0.f};
  M3 Ab = m3_zero();
#pragma unroll 1
  for (int i = 0; i < 3; ++i) {
    float d0 = ((float)i - st.f[0]) * K.dx;
    const float w0i = sel3(st.w[0], i), dw0i = sel3(st.dw[0], i);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float d1 = ((float)j - st.f[1]) * K.dx;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float d2 = ((float)k - st.f[2]) * K.dx;
        float w = w0i * st.w[1][j] * st.w[2][k];
        float4 q = gg[node_addr(st.b[0] + i, st.b[1] + j, st.b[2] + k, K.nb)];
        vb[0] += w * q.x; vb[1] += w * q.y; vb[2] += w * q.z;
        Ab.m[0] += w * q.x * d0; Ab.m[1] += w * q.x * d1; Ab.m[2] += w * q.x * d2;
        Ab.m[3] += w * q.y * d0; Ab.m[4] += w * q.y * d1; Ab.m[5] += w * q.y * d2;
        Ab.m[6] += w * q.z * d0; Ab.m[7] += w * q.z * d1; Ab.m[8] += w * q.z * d2;
        float a0 = mom[0] + A.m[0] * d0 + A.m[1] * d1 + A.m[2] * d2;
        float a1 = mom[1] + A.m[3] * d0 + A.m[4] * d1 + A.m[5] * d2;
        float a2 = mom[2] + A.m[6] * d0 + A.m[7] * d1 + A.m[8] * d2;
        float dLdw = q.x * a0 + q.y * a1 + q.z * a2 + q.w * pm;
        float gw0 = dw0i * st.w[1][j] * st.w[2][k] * K.inv_dx;
        float gw1 = w0i * st.dw[1][j] * st.w[2][k] * K.inv_dx;
        float gw2 = w0i * st.w[1][j] * st.dw[2][k] * K.inv_dx;
        // A^T mvbar
        float t0 = A.m[0] * q.x + A.m[3] * q.y + A.m[6] * q.z;
        float t1 = A.m[1] * q.x + A.m[4] * q.y + A.m[7] * q.z;
        float t2 = A.m[2] * q.x + A.m[5] * q.y + A.m[8] * q.z;
        xb[0] += dLdw * gw0 - w * t0;
        xb[1] += dLdw * gw1 - w * t1;
        xb[2] += dLdw * gw2 - w * t2;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    gx[3 * p + a] += xb[a];
    gvp[3 * p + a] = pm * vb[a];
  }
  M3 o;
#pragma unroll
  for (int i = 0; i < 9; ++i) o.m[i] = pm * Ab.m[i];
  m3_store(gC + 9 * p, o);
#pragma unroll
  for (int i = 0; i < 9; ++i) o.m[i] = ks * Ab.m[i];
  m3_store(gS + 9 * p, o);
}

__global__ void k_grid_stats(const float4* __restrict__ gm, const int* __restrict__ list, const int* __restrict__ count,
                             int* __restrict__ out) {
  const int cnt = *count;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int local = 0;
  for (int li = blockIdx.x * 4 + wave; li < cnt; li += gridDim.x * 4) local += gm[(list[li] << 6) + lane].w > 0.f ? 1 : 0;
  if (local) atomicAdd(&out[1], local);
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = cnt;
}

__global__ void k_grid_export(MpmK K, const float4* __restrict__ gm, const float4* __restrict__ gv, float* mv, float* m,
                              float* v) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int G = K.G;
  if (t >= G * G * G) return;
  int i = t / (G * G), r = t - i * G * G, j = r / G, k = r - j * G;
  int a = node_addr(i, j, k, K.nb);
  float4 q = gm[a], u = gv[a];
  if (mv) { mv[3 * t] = q.x; mv[3 * t + 1] = q.y; mv[3 * t + 2] = q.z; }
  if (m) m[t] = q.w;
  if (v) { v[3 * t] = u.x; v[3 * t + 1] = u.y; v[3 * t + 2] = u.z; }
}

// ---------------------------------------------------------------- host API
extern "C" int nm_mpm_create(const nm_mpm_cfg* cfg, nm_mpm** out) {
  NM_REQUIRE(cfg && out, "null cfg/out");
  NM_REQUIRE(cfg->bc == 0 || cfg->bc == 1, "invalid boundary condition (0 = noslip, 1 = freeslip)");
  NM_REQUIRE(cfg->num_grids >= 4 && cfg->num_grids <= 1024, "num_grids out of range [4,1024]");
  nm_mpm* h = new nm_mpm();
  h->cfg = *cfg;
  MpmK& K = h->k;
  K.G = cfg->num_grids;
  K.Gp = ((cfg->num_grids + 2 + 3) / 4) * 4;
  K.nb = K.Gp / 4;
  K.dt = cfg->dt;
  K.dx = 1.0f / (float)cfg->num_grids;   // mpm.py:516-517
  K.inv_dx = (float)cfg->num_grids;
  K.eps = cfg->eps;
  for (int a = 0; a < 3; ++a) K.gdt[a] = cfg->gravity[a] * cfg->dt;
  K.bound = cfg->bound;
  K.bc = cfg->bc;
  K.dbg = getenv("NM_DBG") ? atoi(getenv("NM_DBG")) : 0;
  K.dbg_buf = nullptr;
  if (K.dbg & 8) { NM_HIP_CHECK(hipMalloc(&K.dbg_buf, 4096 * 4 * sizeof(long long))); NM_HIP_CHECK(hipMemset(K.dbg_buf, 0, 4096 * 4 * sizeof(long long))); }
  h->nblocks = K.nb * K.nb * K.nb;
  size_t nodes = (size_t)h->nblocks * 64;
  h->gm = h->gv = h->gg = nullptr;
  NM_HIP_CHECK(hipMalloc(&h->gm, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMalloc(&h->gv, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMalloc(&h->gg, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMalloc(&h->flags, h->nblocks * sizeof(int)));
  NM_HIP_CHECK(hipMalloc(&h->list[0], h->nblocks * sizeof(int)));
  NM_HIP_CHECK(hipMalloc(&h->list[1], h->nblocks * sizeof(int)));
  NM_HIP_CHECK(hipMalloc(&h->count, 4 * sizeof(int)));
  NM_HIP_CHECK(hipMemset(h->gm, 0, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMemset(h->gv, 0, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMemset(h->gg, 0, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMemset(h->flags, 0, h->nblocks * sizeof(int)));
  NM_HIP_CHECK(hipMemset(h->count, 0, 4 * sizeof(int)));
  NM_HIP_CHECK(hipDeviceSynchronize());
  h->cur = 0;
  h->epoch = 0;
  *out = h;
  return NM_OK;
}

float nm_mpm_get_dt(const nm_mpm* h) { return h->k.dt; }

extern "C" int nm_mpm_destroy(nm_mpm* h) {
  if (!h) return NM_OK;
  hipFree(h->gm); hipFree(h->gv); hipFree(h->gg); hipFree(h->flags);
  hipFree(h->list[0]); hipFree(h->list[1]); hipFree(h->count);
  delete h;
  return NM_OK;
}

static const int kSweepGrid = 512;  // workgroups for the active-block sweeps (grid-stride over the list)

// clear + p2g + grid_op (shared by forward, backward-recompute and forward_extra)
static int mpm_build_grid(nm_mpm* h, int n, const nm_statics* st, const nm_particles* cur, hipStream_t s) {
  const int prev = h->cur, now = prev ^ 1;
  h->epoch += 1;
  NM_LAUNCH(k_clear, dim3(kSweepGrid), dim3(256), 0, s, h->gm, h->gv, h->gg, h->list[prev], h->count + prev,
                     h->count + now);
  NM_LAUNCH_CHECK();
  if (n > 0) {
    NM_LAUNCH(k_p2g, dim3(nm_div_up(n, NM_WT_CHUNK)), dim3(64), 0, s, h->k, n, st->vol, st->rho, st->enabled, cur->x,
                       cur->v, cur->C, cur->stress, h->gm, h->flags, h->list[now], h->count + now, h->epoch);
    NM_LAUNCH_CHECK();
  }
  NM_LAUNCH(k_grid_op, dim3(kSweepGrid), dim3(256), 0, s, h->k, h->gm, h->gv, h->list[now], h->count + now);
  NM_LAUNCH_CHECK();
  h->cur = now;
  return NM_OK;
}

static int check_particles(const nm_statics* st, const nm_particles* p, bool need_stress) {
  NM_REQUIRE(st && st->vol && st->rho && st->clip_bound && st->enabled, "null statics");
  NM_REQUIRE(p && p->x && p->v && p->C && p->F, "null particle arrays");
  if (need_stress) NM_REQUIRE(p->stress, "null stress");
  return NM_OK;
}

extern "C" int nm_mpm_forward(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* next,
                              void* stream) {
  NM_REQUIRE(h, "null handle");
  NM_REQUIRE(n >= 0, "negative particle count");
  if (n == 0) return NM_OK;  // empty input: nothing to scatter or gather
  int rc = check_particles(st, cur, true);
  if (rc) return rc;
  rc = check_particles(st, next, false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  rc = mpm_build_grid(h, n, st, cur, s);
  if (rc) return rc;
  if (n > 0) {
    NM_LAUNCH(k_g2p, dim3(nm_div_up(n, 256)), dim3(256), 0, s, h->k, n, st->clip_bound, st->enabled, cur->x,
                       cur->v, cur->C, cur->F, h->gv, next->x, next->v, next->C, next->F);
    NM_LAUNCH_CHECK();
  }
  return NM_OK;
}

extern "C" int nm_mpm_forward_extra(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, int32_t n_extra,
                                    const nm_statics* st_extra, nm_particles* extra, void* stream) {
  NM_REQUIRE(h, "null handle");
  NM_REQUIRE(n >= 0 && n_extra >= 0, "negative particle count");
  int rc = NM_OK;
  if (n > 0) rc = check_particles(st, cur, true);
  if (rc) return rc;
  if (n_extra > 0) rc = check_particles(st_extra, extra, false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  rc = mpm_build_grid(h, n, st, cur, s);
  if (rc) return rc;
  if (n_extra > 0) {
    NM_LAUNCH(k_g2p, dim3(nm_div_up(n_extra, 256)), dim3(256), 0, s, h->k, n_extra, st_extra->clip_bound,
                       st_extra->enabled, extra->x, extra->v, extra->C, extra->F, h->gv, extra->x, extra->v, extra->C,
                       extra->F);
    NM_LAUNCH_CHECK();
  }
  return NM_OK;
}

extern "C" int nm_mpm_backward(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur,
                               const nm_particles* next, const nm_particles* gnext, nm_particles* gcur, void* stream) {
  NM_REQUIRE(h, "null handle");
  NM_REQUIRE(n >= 0, "negative particle count");
  if (n == 0) return NM_OK;
  int rc = check_particles(st, cur, true);
  if (rc) return rc;
  NM_REQUIRE(next && next->v && next->C, "next state (v, C) required");
  NM_REQUIRE(gnext && gnext->x && gnext->v && gnext->C && gnext->F, "null incoming gradients");
  NM_REQUIRE(gcur && gcur->x && gcur->v && gcur->C && gcur->F && gcur->stress, "null outgoing gradients");
  hipStream_t s = (hipStream_t)stream;
  rc = mpm_build_grid(h, n, st, cur, s);  // recompute, mpm.py:312-315
  if (rc) return rc;
  if (n == 0) return NM_OK;
  const int now = h->cur;
  const int nwg = nm_div_up(n, 256);
  NM_LAUNCH(k_g2p_bwd, dim3(nm_div_up(n, NM_WT_CHUNK)), dim3(64), 0, s, h->k, n, st->clip_bound, st->enabled, cur->x, cur->F, next->v,
                     next->C, gnext->x, gnext->v, gnext->C, gnext->F, h->gv, h->gg, gcur->x, gcur->F);
  NM_LAUNCH_CHECK();
  NM_LAUNCH(k_grid_op_bwd, dim3(kSweepGrid), dim3(256), 0, s, h->k, h->gm, h->gg, h->list[now], h->count + now);
  NM_LAUNCH_CHECK();
  NM_LAUNCH(k_p2g_bwd, dim3(nwg), dim3(256), 0, s, h->k, n, st->vol, st->rho, st->enabled, cur->x, cur->v, cur->C,
                     cur->stress, h->gg, gcur->x, gcur->v, gcur->C, gcur->stress);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" int nm_mpm_grid_stats(nm_mpm* h, int32_t* active_blocks, int32_t* nodes_with_mass, void* stream) {
  NM_REQUIRE(h, "null handle");
  hipStream_t s = (hipStream_t)stream;
  NM_HIP_CHECK(hipMemsetAsync(h->count + 2, 0, 2 * sizeof(int), s));
  NM_LAUNCH(k_grid_stats, dim3(kSweepGrid), dim3(256), 0, s, h->gm, h->list[h->cur], h->count + h->cur,
                     h->count + 2);
  NM_LAUNCH_CHECK();
  int host[2];
  NM_HIP_CHECK(hipMemcpyAsync(host, h->count + 2, sizeof(host), hipMemcpyDeviceToHost, s));
  NM_HIP_CHECK(hipStreamSynchronize(s));
  if (active_blocks) *active_blocks = host[0];
  if (nodes_with_mass) *nodes_with_mass = host[1];
  return NM_OK;
}

extern "C" int nm_mpm_debug_fetch(nm_mpm* h, long long* host, int n) {
  if (!h->k.dbg_buf) return NM_ERR_INVALID;
  NM_HIP_CHECK(hipDeviceSynchronize());
  NM_HIP_CHECK(hipMemcpy(host, h->k.dbg_buf, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
  return NM_OK;
}

extern "C" int nm_mpm_grid_export(nm_mpm* h, float* mv, float* m, float* v, void* stream) {
  NM_REQUIRE(h, "null handle");
  int G = h->k.G;
  NM_LAUNCH(k_grid_export, dim3(nm_div_up((int64_t)G * G * G, 256)), dim3(256), 0, (hipStream_t)stream, h->k,
                     h->gm, h->gv, mv, m, v);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
