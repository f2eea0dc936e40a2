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
#include "nm_grid.h"
#include <stdlib.h>
#include <string.h>


// experiment switches (tools/exp_*.py) exist only in -DNM_PHASES builds; in the shipped library they fold to constants
#ifdef NM_PHASES
#define NM_DBG_BIT(K, bit) (((K).dbg & (bit)) != 0)
#else
#define NM_DBG_BIT(K, bit) false
#endif

struct nm_mpm {
  nm_mpm_cfg cfg;
  MpmK k;
  int nblocks;
  float4* gm;  // {mv.xyz, m}
  float4* gv;  // {v.xyz, 0}
  float4* gg;  // adjoint scratch: {vbar.xyz,0} -> {mvbar.xyz, mbar}
  int* flags;
  int* list[3];   // active-block lists in rotation: previous / current / next substep
  int* count;     // [0..2] block counters in the same rotation, [4..5] stats
  int cur;
  int epoch;
  int* sh_cnt;    // sharded runs only (nm_shard.hip): per-block rank counter / first position, allocated on first use
  int* sh_pos;
  int* sh_slot;   // per block: slot of the block in the frame's exchange buffer, -1 = none (nm_mpm_xchg_arrays)
  int* sh_dil;    // per block: tag of the last frame whose negotiated neighbourhood holds the block
  int dil_tag;
  const void* resident_rec;   // the grid cache record whose substep the grid holds right now (forward pass just built it) ...
  int resident_epoch;         // ... as long as the epoch has not moved: the reverse sweep's first substep then restores nothing
  int gv_stale;   // the last clear left the velocity array alone (GridPrologue keep_gv): the next grid update zeroes what dropped out
  int fresh_rows; // g2p writes a fresh state's values into the rows of disabled particles (roll-out checkpoints, nm_grid.h)
};
void nm_mpm_set_fresh_rows(nm_mpm* h, int on) { h->fresh_rows = on; }

#ifdef NM_PHASES
__device__ int g_nm_markslow[4];
extern "C" int nm_debug_markslow(int* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nm_markslow), 16) == hipSuccess ? 0 : -2; }
#endif
// Keep a loaded value where it was loaded: the compiler sinks a load into the only branch that uses its result - behind the
// `enabled` test, i.e. behind another load's round trip.  An empty asm that "modifies" the register makes the value needed
// HERE; placed behind ALL of a particle's loads, the pins wait for the whole batch once.
__device__ __forceinline__ void nm_pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void nm_pin(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void nm_pin(M3& m) {
#pragma unroll
  for (int i = 0; i < 9; ++i) nm_pin(m.m[i]);
}
__device__ __forceinline__ void mark_block(int b, int* __restrict__ flags, int* __restrict__ list,
                                           int* __restrict__ count, int epoch) {
#ifdef NM_PHASES
  atomicAdd(&g_nm_markslow[0], 1);
#endif
  if (flags[b] != epoch) {
#ifdef NM_PHASES
    atomicAdd(&g_nm_markslow[1], 1);
#endif
    if (atomicExch(&flags[b], epoch) != epoch) {
#ifdef NM_PHASES
      atomicAdd(&g_nm_markslow[2], 1);
#endif
      int pos = atomicAdd(count, 1);
      list[pos] = b;
    }
  }
}

// ---------------------------------------------------------------- workgroup scatter (p2g, g2p adjoint)
#ifndef NM_SC_T
#define NM_SC_T 256      // threads = particles per workgroup
#endif
#ifndef NM_WT_CAP
#define NM_WT_CAP (8 * NM_SC_T)   // tile nodes a workgroup can own (NM_NPT per thread)
#endif
#define NM_NPT (NM_WT_CAP / NM_SC_T)
#define NM_SC_NW (NM_SC_T / 64)   // waves per workgroup
#define NM_WT_MAXPASS 12 // boxes a wave tries (wave_scatter) before its leftovers go to direct global atomics

#ifdef NM_PHASES
__device__ long long g_nm_scatter[8 * 4096];
extern "C" int nm_debug_scatter(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nm_scatter), (size_t)n * sizeof(long long)) == hipSuccess ? 0 : -2;
}
#define SC_DECL long long sc_t0 = clock64(); long long sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SC_PH(i) { long long t1 = clock64(); sc[i] += t1 - sc_t0; sc_t0 = t1; }
#define SC_STORE(npass) if (threadIdx.x == 0 && blockIdx.x < 4096) { sc[7] = (npass); for (int i = 0; i < 8; ++i) g_nm_scatter[blockIdx.x * 8 + i] = sc[i]; }
#else
#define SC_DECL
#define SC_PH(i)
#define SC_STORE(npass)
#endif
struct TileGeom {
  int o[3];
  int n[3];
  int vol;
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
struct ScatterLds {
  float4 C[NM_SC_T * 9];     // one i-slab (9 stencil nodes) of every particle's contributions, in cell-sorted order
  float4 tile[NM_WT_CAP];    // node sums of the workgroup's bounding box
  int cnt[NM_WT_CAP + 8];    // particles per stencil origin -> exclusive offsets (+ total as sentinel)
  short run_cell[NM_SC_T];   // compacted list of non-empty origin cells (<= one per particle)
  short ainv[3][NM_SC_T];    // axis compression: compressed coordinate -> grid coordinate
  int red[80];               // block reductions / broadcasts; fp64 path: 8 group rows of 8 ints + the jump list
};

// Fallback for a workgroup whose 256 particles neither fit one tile nor compress into one (the particle order left the
// body several times: up to ~7 disjoint runs of the curve in one chunk): every WAVE scatters its own 64 consecutive
// particles - a short, compact piece of the order - through a private quarter of the LDS buffers, with the same
// sort / sum / push / flush sequence as the workgroup pass but 8x8x8-node boxes and wave barriers only, so the four
// waves work on four different boxes at the same time instead of taking turns at workgroup-wide passes (7 passes at
// ~20k cycles each made one such workgroup the critical path of the whole launch).
#define NM_WV_BOX 8
#define NM_WV_VOL (NM_WV_BOX * NM_WV_BOX * NM_WV_BOX)
template <int NCH, class ContribF>
__device__ __forceinline__ void wave_scatter(const MpmK& K, bool en, const int* base, float4* __restrict__ grid, int* flags,
                                             int* list, int* count, int epoch, ScatterLds& L, ContribF contrib) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4* C = L.C + wave * 64 * 9;
  float4* tile = L.tile + wave * NM_WV_VOL;
  int* cnt = L.cnt + wave * (NM_WV_VOL + 1);
  short* run_cell = L.run_cell + wave * 64;
  bool pending = en;
  for (int pass = 0; pass <= K.maxpass; ++pass) {
    const unsigned long long pm = __ballot(pending);
    if (pm == 0ull) break;
    if (pass == K.maxpass) {   // last resort: per-particle global atomics
      if (pending) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const float4 c = contrib(i, j, k);
              float* dst = (float*)&grid[node_addr(base[0] + i, base[1] + j, base[2] + k, K.nb)];
              unsafeAtomicAdd(dst, c.x);
              unsafeAtomicAdd(dst + 1, c.y);
              unsafeAtomicAdd(dst + 2, c.z);
              if (NCH == 4) unsafeAtomicAdd(dst + 3, c.w);
            }
        if (flags) {
          for (int i = base[0] >> 2; i <= (base[0] + 2) >> 2; ++i)
            for (int j = base[1] >> 2; j <= (base[1] + 2) >> 2; ++j)
              for (int k = base[2] >> 2; k <= (base[2] + 2) >> 2; ++k)
                mark_block((i * K.nb + j) * K.nb + k, flags, list, count, epoch);
        }
      }
      break;
    }
    // box anchored at the first pending lane: node origin o = anchor - 1 (clamped), 8 nodes = 6 stencil origins per axis
    const int first = __ffsll((long long)pm) - 1;
    int o[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) o[a] = max(0, min(__shfl(base[a], first, 64) - 1, K.Gp - NM_WV_BOX));
    const int t0 = base[0] - o[0], t1 = base[1] - o[1], t2 = base[2] - o[2];
    const bool in = pending && t0 >= 0 && t0 + 3 <= NM_WV_BOX && t1 >= 0 && t1 + 3 <= NM_WV_BOX && t2 >= 0 && t2 + 3 <= NM_WV_BOX;
    const int ci = in ? (t0 * NM_WV_BOX + t1) * NM_WV_BOX + t2 : 0;
    for (int i = lane; i < NM_WV_VOL + 1; i += 64) cnt[i] = 0;
    for (int i = lane; i < NM_WV_VOL; i += 64) tile[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_wave_barrier();
    const int rank = in ? atomicAdd(&cnt[ci], 1) : 0;
    __builtin_amdgcn_wave_barrier();
    // exclusive scan of the 512 cell counts (8 consecutive cells per lane) + list of the non-empty cells
    int v[NM_WV_VOL / 64], sum = 0, nz = 0;
#pragma unroll
    for (int q = 0; q < NM_WV_VOL / 64; ++q) { v[q] = cnt[lane * (NM_WV_VOL / 64) + q]; sum += v[q]; nz += v[q] > 0; }
    int wtot, nruns;
    int off = wave_excl_scan_i(sum, lane, wtot);
    int roff = wave_excl_scan_i(nz, lane, nruns);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < NM_WV_VOL / 64; ++q) {
      const int c = lane * (NM_WV_VOL / 64) + q;
      cnt[c] = off;
      off += v[q];
      if (v[q] > 0) run_cell[roff++] = (short)c;
    }
    if (lane == 63) cnt[NM_WV_VOL] = off;
    __builtin_amdgcn_wave_barrier();
    const int slot = in ? cnt[ci] + rank : 0;
    const bool owner = lane < nruns;             // at most 64 non-empty cells: one per particle
    const int mycell = owner ? (int)run_cell[lane] : 0;
    const int s0 = owner ? cnt[mycell] : 0;
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
      if (in) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int k = 0; k < 3; ++k) C[slot * 9 + j * 3 + k] = contrib(i, j, k);
      }
      __builtin_amdgcn_wave_barrier();
      for (int u = lane; u < nruns * 9; u += 64) {
        const int r = u / 9, q = u - 9 * r;
        const int cell = (int)run_cell[r];
        const int a0 = cnt[cell], a1 = cnt[cell + 1];
        float4 acc = C[a0 * 9 + q];
        for (int s_ = a0 + 1; s_ < a1; ++s_) {
          const float4 t4 = C[s_ * 9 + q];
          acc.x += t4.x; acc.y += t4.y; acc.z += t4.z; acc.w += t4.w;
        }
        C[a0 * 9 + q] = acc;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (owner) {
            const int node = mycell + (i * NM_WV_BOX + j) * NM_WV_BOX + k;
            float4 t4 = tile[node];
            const float4 c4 = C[s0 * 9 + j * 3 + k];
            t4.x += c4.x; t4.y += c4.y; t4.z += c4.z; t4.w += c4.w;
            tile[node] = t4;
          }
          __builtin_amdgcn_wave_barrier();   // LDS operations of one wave complete in program order
        }
    }
    if (flags && lane < 27) {   // the box spans at most 3 blocks per axis
      const int bi = (o[0] >> 2) + lane / 9, bj = (o[1] >> 2) + (lane / 3) % 3, bk = (o[2] >> 2) + lane % 3;
      if (bi <= (o[0] + NM_WV_BOX - 1) >> 2 && bj <= (o[1] + NM_WV_BOX - 1) >> 2 && bk <= (o[2] + NM_WV_BOX - 1) >> 2)
        mark_block((bi * K.nb + bj) * K.nb + bk, flags, list, count, epoch);
    }
    for (int nidx = lane; nidx < NM_WV_VOL; nidx += 64) {
      const float4 t = tile[nidx];
      if (t.x != 0.f || t.y != 0.f || t.z != 0.f || t.w != 0.f) {
        const int a_ = nidx / (NM_WV_BOX * NM_WV_BOX), r = nidx - a_ * (NM_WV_BOX * NM_WV_BOX);
        const int b_ = r / NM_WV_BOX, c_ = r - b_ * NM_WV_BOX;
        float* dst = (float*)&grid[node_addr(o[0] + a_, o[1] + b_, o[2] + c_, K.nb)];
        unsafeAtomicAdd(dst, t.x);
        unsafeAtomicAdd(dst + 1, t.y);
        unsafeAtomicAdd(dst + 2, t.z);
        if (NCH == 4) unsafeAtomicAdd(dst + 3, t.w);
      }
    }
    pending = pending && !in;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- fp64-atomic scatter (round 4; K.smode == 1, the default).
// ds_add_f64 retires 6.5 lanes/clk/CU on gfx950 where ds_add_f32 retires 0.33 (tools/ubench_lds_int.hip: the f32 form is the
// slow one, not LDS atomics as such; ds_add_u64 8.3, ds_add_u32 9.2), so every particle adds its 27 x NCH contributions
// straight into the workgroup's tile - kept as NCH planes of doubles, node-contiguous.  With the thread -> particle
// permutation of the callers (scatter_particle: the 16 lanes an LDS cycle serves hold 16 different cells) the pattern runs at
// 2.6-3.6 lanes/clk/CU, against 1.3-1.4 in particle order and 0.9 with the four channels of a node side by side.  No sort, no
// per-offset barriers (6 workgroup barriers instead of ~45), and a node sum is rounded once (double -> float at the flush)
// instead of once per addend.
// A chunk of the particle list that the space-filling curve leaves and re-enters (bounding box > tile) is cut at the jumps -
// consecutive particles more than two cells apart - into up to eight GROUPS, each with a box of its own inside the same tile
// memory; what still does not fit (an arbitrary order) goes particle by particle to global atomics: any order is correct.
#define NM_F64_PS (NM_WT_CAP + 8)      // plane stride in doubles (+8: the four channels of a node sit in different banks at the flush)
#define NM_F64_MAXG 8
template <int NCH, class ContribF>
__device__ __forceinline__ void wg_scatter_f64(const MpmK& K, bool en, int lp, const int* base, float4* __restrict__ grid, int* flags,
                                               int* list, int* count, int epoch, ScatterLds& L, ContribF contrib) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  SC_DECL
  double* acc = reinterpret_cast<double*>(L.C);                 // C | tile are contiguous: 68 KiB >= 4 planes x 2056 x 8 B
  static_assert(sizeof(L.C) + sizeof(L.tile) >= (size_t)NM_F64_PS * 4 * sizeof(double) && offsetof(ScatterLds, tile) == sizeof(L.C),
                "the fp64 tile overlays the sort path's contribution buffer and tile");
  // ---- bounding box of the stencil origins
  int lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_min_i(en ? base[a] : 0x7fffffff);
    hi[a] = wave_max_i(en ? base[a] : -0x7fffffff);
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { L.red[wave * 6 + a] = lo[a]; L.red[wave * 6 + 3 + a] = hi[a]; }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = L.red[a]; hi[a] = L.red[3 + a];
#pragma unroll
    for (int w2 = 1; w2 < NM_SC_NW; ++w2) { lo[a] = min(lo[a], L.red[6 * w2 + a]); hi[a] = max(hi[a], L.red[6 * w2 + 3 + a]); }
  }
  __syncthreads();
  if (lo[0] == 0x7fffffff) return;  // nothing enabled in this workgroup
  if (NM_DBG_BIT(K, 4)) return;
  // geometry of this thread's group: origin, extents, first tile slot
  int go[3] = {lo[0], lo[1], lo[2]}, gn[3] = {hi[0] - lo[0] + 3, hi[1] - lo[1] + 3, hi[2] - lo[2] + 3}, goff = 0;
  int total = gn[0] * gn[1] * gn[2], ngroups = 1;
  bool direct = false;      // last resort: this workgroup's particles go to global memory one by one
  int* gtab = L.cnt + 3 * NM_SC_T;     // group boxes while they are being reduced: per group min[3], max[3] (8 ints)
  if (total > NM_WT_CAP) {
    // ---- cut the chunk at its jumps
    int* pk = L.cnt;                 // stencil origins by position in the chunk, three arrays of NM_SC_T (-1: disabled)
    int* jl = L.red + NM_F64_MAXG * 8;   // [0] number of jumps, [1..] their positions
#pragma unroll
    for (int a = 0; a < 3; ++a) pk[a * NM_SC_T + lp] = en ? base[a] : -1;
    if (tid == 0) jl[0] = 0;
    if (tid < NM_F64_MAXG * 8) gtab[tid] = (tid & 7) < 3 ? 0x7fffffff : ((tid & 7) < 6 ? -1 : 0);     // min | max | -
    __syncthreads();
    if (en && lp > 0) {
      const int q = pk[lp - 1];
      if (q >= 0) {
        const int d0 = abs(q - base[0]), d1 = abs(pk[NM_SC_T + lp - 1] - base[1]), d2 = abs(pk[2 * NM_SC_T + lp - 1] - base[2]);
        if (max(d0, max(d1, d2)) > 2) {
          const int pos = atomicAdd(&jl[0], 1);
          if (pos < NM_F64_MAXG - 1) jl[1 + pos] = lp;
        }
      }
    }
    __syncthreads();
    const int nj = jl[0];
    direct = nj > NM_F64_MAXG - 1;
    int grp = 0;
    if (!direct) {
      for (int q = 0; q < nj; ++q) grp += jl[1 + q] <= lp ? 1 : 0;
      if (en) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(&gtab[grp * 8 + a], base[a]); atomicMax(&gtab[grp * 8 + 3 + a], base[a]); }
      }
    }
    __syncthreads();
    if (!direct) {
      ngroups = nj + 1;
      int off = 0;
      for (int q = 0; q < ngroups; ++q) {
        int n_[3], v_ = 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) { n_[a] = gtab[q * 8 + 3 + a] < 0 ? 0 : gtab[q * 8 + 3 + a] - gtab[q * 8 + a] + 3; v_ *= n_[a]; }
        if (q == grp) {
#pragma unroll
          for (int a = 0; a < 3; ++a) { go[a] = gtab[q * 8 + a]; gn[a] = n_[a]; }
          goff = off;
        }
        off += v_;
      }
      total = off;
      direct = total > NM_WT_CAP;
    }
    __syncthreads();       // everybody has read the min / max words; the table is rewritten below in (origin, extent, offset) form
    if (!direct && tid < ngroups) {
      int n_[3], o_[3], off = 0;
      for (int q = 0; q <= tid; ++q) {
        int v_ = 1;
#pragma unroll
        for (int a = 0; a < 3; ++a) { o_[a] = gtab[q * 8 + a]; n_[a] = gtab[q * 8 + 3 + a] < 0 ? 0 : gtab[q * 8 + 3 + a] - o_[a] + 3; v_ *= n_[a]; }
        if (q < tid) off += v_;
      }
      // (written after the loop's reads of THIS thread; other threads read only rows <= their own id, row tid is ours)
      int* row = L.red;    // red[0..63]: 8 groups x (o[3], n[3], off, end)
#pragma unroll
      for (int a = 0; a < 3; ++a) { row[tid * 8 + a] = o_[a]; row[tid * 8 + 3 + a] = n_[a]; }
      row[tid * 8 + 6] = off;
      row[tid * 8 + 7] = off + n_[0] * n_[1] * n_[2];
    }
    // (L.red is published by the barrier that follows the tile clear)
  }
  SC_PH(0)
  if (direct) {
    if (en) {
#pragma unroll 1
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float4 c = contrib(i, j, k);
            float* dst = (float*)&grid[node_addr(base[0] + i, base[1] + j, base[2] + k, K.nb)];
            unsafeAtomicAdd(dst, c.x);
            unsafeAtomicAdd(dst + 1, c.y);
            unsafeAtomicAdd(dst + 2, c.z);
            if (NCH == 4) unsafeAtomicAdd(dst + 3, c.w);
          }
      if (flags) {
        for (int i = base[0] >> 2; i <= (base[0] + 2) >> 2; ++i)
          for (int j = base[1] >> 2; j <= (base[1] + 2) >> 2; ++j)
            for (int k = base[2] >> 2; k <= (base[2] + 2) >> 2; ++k)
              mark_block((i * K.nb + j) * K.nb + k, flags, list, count, epoch);
      }
    }
    SC_STORE(99)
    return;
  }
  const bool multi = ngroups > 1;
  // Block stamps of the bounding box are read NOW, while the memory system is quiet: at the end of the kernel the same
  // read queues behind every workgroup's atomic flush (10-25k cycles under load).
  int pre_flag = epoch;
  const int b0 = go[0] >> 2, b1 = go[1] >> 2, b2 = go[2] >> 2;
  const int m1 = ((go[1] + gn[1] - 1) >> 2) - b1 + 1, m2 = ((go[2] + gn[2] - 1) >> 2) - b2 + 1;
  const int nblk = (((go[0] + gn[0] - 1) >> 2) - b0 + 1) * m1 * m2;
  if (flags && !multi && tid < nblk) {
    const int i = tid / (m1 * m2), r = tid - i * (m1 * m2);
    pre_flag = flags[((b0 + i) * K.nb + (b1 + r / m2)) * K.nb + (b2 + r % m2)];
  }
  SC_PH(1)
  {
    const double2 z2 = make_double2(0.0, 0.0);
    double2* a2 = reinterpret_cast<double2*>(acc);
    const int half = (total + 1) >> 1;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      for (int i = tid; i < half; i += NM_SC_T) a2[c * (NM_F64_PS / 2) + i] = z2;
  }
  if (flags && multi && en && !NM_DBG_BIT(K, 2)) {      // several boxes: every particle stamps the blocks of its own stencil
    int bid[8], fl[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = (base[0] >> 2) + (q >> 2), j = (base[1] >> 2) + ((q >> 1) & 1), k = (base[2] >> 2) + (q & 1);
      const bool ok = i <= (base[0] + 2) >> 2 && j <= (base[1] + 2) >> 2 && k <= (base[2] + 2) >> 2;
      bid[q] = ok ? (i * K.nb + j) * K.nb + k : -1;
    }
    // (eight UNCONDITIONAL loads - q = 0 is always a block of the stencil and stands in for the ones that are not: a load under
    //  a lane predicate is a branch region of its own, and the eight stamps came back one round trip after the other)
#pragma unroll
    for (int q = 0; q < 8; ++q) fl[q] = flags[bid[q] >= 0 ? bid[q] : bid[0]];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (bid[q] >= 0 && fl[q] != epoch) mark_block(bid[q], flags, list, count, epoch);
  }
  __syncthreads();
  SC_PH(2)
  const int ny = gn[1], nz = gn[2];
  {
    // Lanes l and l ^ 1 hold two CONSECUTIVE particles (scatter_particle) - in a cell-ordered list usually two members of one
    // cell, i.e. the same 27 nodes.  Such a pair adds its values once: v + v(lane ^ 1) by a quad-permute DPP move (one or two
    // instructions; the row swap of the first version cost five with its register copies, and the merge was half of the
    // phase's instructions), issued by the even lane; a pair that straddles two cells (or holds a disabled particle) adds
    // separately.  Half the atomic lanes; the decision is per pair, so a stale order only loses the saving, never a contribution.
    const int ci = goff + ((base[0] - go[0]) * ny + (base[1] - go[1])) * nz + (base[2] - go[2]);
    const int key = en ? ci : -1 - lane;        // (never equal to the partner's)
    const int pkey = __builtin_amdgcn_update_dpp(0, key, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]: lane ^ 1
    const bool paired = en && !NM_DBG_BIT(K, 128) && pkey == key;      // both lanes of the pair hold the same cell
    const bool issue = en && !(paired && (lane & 1));
    double* a0 = acc + ci;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double* row = a0 + (i * ny + j) * nz;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 c = contrib(i, j, k);
          float v[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) {
            const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[ch]), 0xB1, 0xf, 0xf, true));
            v[ch] = paired ? v[ch] + other : v[ch];
          }
          if (issue) {
            unsafeAtomicAdd(row + k, (double)v[0]);
            unsafeAtomicAdd(row + NM_F64_PS + k, (double)v[1]);
            unsafeAtomicAdd(row + 2 * NM_F64_PS + k, (double)v[2]);
            if (NCH == 4) unsafeAtomicAdd(row + 3 * NM_F64_PS + k, (double)v[3]);
          }
        }
      }
  }
  __syncthreads();
  SC_PH(3)
  // node addresses (two integer divisions and the block / in-block split, once per node) and - one box - which blocks of the
  // bounding box received something (an untouched block stamped here would hold no mass, drop out at the next clear and take
  // the atomic path of mark_block again every substep)
  const bool mark = flags != nullptr && !multi && !NM_DBG_BIT(K, 2);
  int* naddr = L.cnt;
  int* touched = reinterpret_cast<int*>(&L.ainv[0][0]);
  static_assert(sizeof(L.ainv) >= 320 * sizeof(int), "block table of a 2048-node box (<= ~232 blocks)");
  if (mark) {
    for (int t = tid; t < nblk; t += NM_SC_T) touched[t] = 0;
    __syncthreads();
  }
  for (int nidx = tid; nidx < total; nidx += NM_SC_T) {
    int o_[3] = {go[0], go[1], go[2]}, n1 = gn[1], n2 = gn[2], loc = nidx;
    if (multi) {
      int q = 0;
      while (q + 1 < ngroups && nidx >= L.red[q * 8 + 7]) ++q;
      o_[0] = L.red[q * 8]; o_[1] = L.red[q * 8 + 1]; o_[2] = L.red[q * 8 + 2];
      n1 = L.red[q * 8 + 4]; n2 = L.red[q * 8 + 5];
      loc = nidx - L.red[q * 8 + 6];
    }
    const int a_ = loc / (n1 * n2), r = loc - a_ * (n1 * n2);
    const int b_ = r / n2, c_ = r - b_ * n2;
    const int x_ = o_[0] + a_, y_ = o_[1] + b_, z_ = o_[2] + c_;
    naddr[nidx] = 4 * node_addr(x_, y_, z_, K.nb);
    if (mark) {
      bool nzv = acc[nidx] != 0.0 || acc[NM_F64_PS + nidx] != 0.0 || acc[2 * NM_F64_PS + nidx] != 0.0;
      if (NCH == 4) nzv = nzv || acc[3 * NM_F64_PS + nidx] != 0.0;
      if (nzv) touched[(((x_ >> 2) - b0) * m1 + ((y_ >> 2) - b1)) * m2 + ((z_ >> 2) - b2)] = 1;
    }
  }
  __syncthreads();
  if (mark) {
    for (int t = tid; t < nblk; t += NM_SC_T) {
      if (touched[t] == 0 || (t == tid && pre_flag == epoch)) continue;     // already stamped when we looked
      const int i = t / (m1 * m2), r = t - i * (m1 * m2);
      const int j = r / m2, k = r - j * m2;
      mark_block(((b0 + i) * K.nb + (b1 + j)) * K.nb + (b2 + k), flags, list, count, epoch);
    }
  }
  SC_PH(4)
  // the flush: one global atomic set per touched node, the sums rounded to fp32 here.  Four channels: one FLOAT per lane
  // (four lanes cover a node, a wave instruction sixteen consecutive nodes of a z run: 314 against 77 G/s, tools/ubench_flush.hip);
  // three channels: one node per lane (the float-per-lane mapping leaves a quarter of the lanes idle there and measured slower)
  float* gridf = (float*)grid;
  if (NCH == 4) {
    for (int idx = tid; idx < 4 * total; idx += NM_SC_T) {
      const float v = (float)acc[(idx & 3) * NM_F64_PS + (idx >> 2)];
      if (!NM_DBG_BIT(K, 1) && v != 0.f) unsafeAtomicAdd(gridf + naddr[idx >> 2] + (idx & 3), v);
    }
  } else {
    for (int nidx = tid; nidx < total; nidx += NM_SC_T) {
      const float vx = (float)acc[nidx], vy = (float)acc[NM_F64_PS + nidx], vz = (float)acc[2 * NM_F64_PS + nidx];
      if (vx != 0.f || vy != 0.f || vz != 0.f) {
        float* dst = gridf + naddr[nidx];
        unsafeAtomicAdd(dst, vx);
        unsafeAtomicAdd(dst + 1, vy);
        unsafeAtomicAdd(dst + 2, vz);
      }
    }
  }
  SC_PH(6)
  SC_STORE(ngroups + 1)
}

// Scatter of the workgroup's 256 particles (one per thread) into `grid` WITHOUT floating-point atomics in LDS
// (ds_add_f32 retires ~0.33 lanes/clk/CU on gfx950) and without serial per-wave chains:
//   1. the particles are counting-sorted by stencil origin inside LDS (256 integer LDS atomics + a block scan), so
//      the members of every origin cell are contiguous whatever order the caller keeps its particles in;
//   2. for each of the three i-slabs of the 3x3x3 stencil every thread writes its 9 float4 contributions to its
//      sorted slot; then thread r sums the members of non-empty cell r (contiguous slots, independent reads) and the
//      cell sums are pushed into the LDS tile one stencil offset at a time: for a FIXED offset distinct cells hit
//      distinct nodes, so plain ds_read_b128 / ds_write_b128 never conflict; a workgroup barrier separates offsets;
//   3. every touched node is flushed with one global atomic set.
// A chunk whose bounding box exceeds the tile (the particle order left the body and re-entered it elsewhere: two or
// three compact clusters) first tries AXIS COMPRESSION: per axis, the coordinates no stencil touches are squeezed
// out (an occupancy array + a block scan give grid coordinate -> compressed coordinate; base, base+1, base+2 stay
// consecutive, so the stencil arithmetic is unchanged) and the whole chunk is still handled in ONE pass if the
// compressed box fits.  What is left goes to wave_scatter (each wave scatters its own 64 particles through small
// private boxes); after NM_WT_MAXPASS boxes the leftovers use per-particle global atomics, so any order is correct.
//   contrib(i, j, k) -> float4 contribution of THIS thread's particle to stencil node (i,j,k)
template <int NCH, class ContribF>
__device__ __forceinline__ void wg_scatter(const MpmK& K, bool en, int lp, const int* base, float4* __restrict__ grid, int* flags,
                                           int* list, int* count, int epoch, ScatterLds& L, ContribF contrib) {
  if (K.smode == 1) {     // (uniform over the launch)
    wg_scatter_f64<NCH>(K, en, lp, base, grid, flags, list, count, epoch, L, contrib);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  SC_DECL
  int sc_pass = 0;
  // ---- bounding box of the stencil origins
  int lo[3], hi[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_min_i(en ? base[a] : 0x7fffffff);
    hi[a] = wave_max_i(en ? base[a] : -0x7fffffff);
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { L.red[wave * 6 + a] = lo[a]; L.red[wave * 6 + 3 + a] = hi[a]; }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = L.red[a]; hi[a] = L.red[3 + a];
#pragma unroll
    for (int w2 = 1; w2 < NM_SC_NW; ++w2) { lo[a] = min(lo[a], L.red[6 * w2 + a]); hi[a] = max(hi[a], L.red[6 * w2 + 3 + a]); }
  }
  __syncthreads();
  if (lo[0] == 0x7fffffff) return;  // nothing enabled in this workgroup
  TileGeom g;
#pragma unroll
  for (int a = 0; a < 3; ++a) { g.o[a] = lo[a]; g.n[a] = hi[a] - lo[a] + 3; }
  g.vol = g.n[0] * g.n[1] * g.n[2];
  bool single = g.vol <= NM_WT_CAP;
  bool cmp = false;          // compressed tile coordinates in use
  int tb[3] = {base[0] - lo[0], base[1] - lo[1], base[2] - lo[2]};   // tile coordinates of this particle's stencil origin
  bool pending = en;
  if (NM_DBG_BIT(K, 4)) return;
  if (!single && !NM_DBG_BIT(K, 32)) {
    // ---- axis compression (scratch: the contribution buffer, not in use yet)
    const int Gp = K.Gp, tot = 3 * Gp;
    short* occ = reinterpret_cast<short*>(L.C);
    short* amap = occ + ((tot + 7) & ~7);
    for (int i = tid; i < tot; i += NM_SC_T) occ[i] = 0;
    __syncthreads();
    if (en) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { occ[a * Gp + base[a]] = 1; occ[a * Gp + base[a] + 1] = 1; occ[a * Gp + base[a] + 2] = 1; }
    }
    __syncthreads();
    const int E = (tot + NM_SC_T - 1) / NM_SC_T;
    int sum = 0;
    for (int e = 0; e < E; ++e) { int idx = tid * E + e; if (idx < tot) sum += occ[idx]; }
    int wtot;
    int off = wave_excl_scan_i(sum, lane, wtot);
    if (lane == 0) L.red[wave] = wtot;
    __syncthreads();
    for (int w2 = 0; w2 < wave; ++w2) off += L.red[w2];
    int total = 0;
#pragma unroll
    for (int w2 = 0; w2 < NM_SC_NW; ++w2) total += L.red[w2];
    for (int e = 0; e < E; ++e) { int idx = tid * E + e; if (idx < tot) { amap[idx] = (short)off; off += occ[idx]; } }
    __syncthreads();
    const int st1 = amap[Gp], st2 = amap[2 * Gp];
    const int cn[3] = {st1, st2 - st1, total - st2};
    const int cstart[3] = {0, st1, st2};
    if (cn[0] <= NM_SC_T && cn[1] <= NM_SC_T && cn[2] <= NM_SC_T && cn[0] * cn[1] * cn[2] <= NM_WT_CAP) {
      for (int idx = tid; idx < tot; idx += NM_SC_T) {
        if (occ[idx]) { int a = idx >= 2 * Gp ? 2 : (idx >= Gp ? 1 : 0); L.ainv[a][amap[idx] - cstart[a]] = (short)(idx - a * Gp); }
      }
      if (en) {
#pragma unroll
        for (int a = 0; a < 3; ++a) tb[a] = amap[a * Gp + base[a]] - cstart[a];
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) { g.o[a] = 0; g.n[a] = cn[a]; }
      g.vol = cn[0] * cn[1] * cn[2];
      cmp = true;
      single = true;
    }
    __syncthreads();   // scratch (= contribution buffer) free again, ainv visible
  }
  if (!single) {      // workgroup-uniform: no workgroup barrier is executed past this point
    SC_PH(0)
    wave_scatter<NCH>(K, en, base, grid, flags, list, count, epoch, L, contrib);
    SC_PH(6)
    SC_STORE(99)
    return;
  }

  SC_PH(0)
  sc_pass = 1;
  // Block stamps of the bounding box are read NOW, while the memory system is quiet: at the end of the kernel the same
  // read queues behind every workgroup's atomic flush (10-25k cycles under load).
  int pre_flag = epoch;
  if (flags && !cmp) {
    const int b0 = g.o[0] >> 2, b1 = g.o[1] >> 2, b2 = g.o[2] >> 2;
    const int m1 = ((g.o[1] + g.n[1] - 1) >> 2) - b1 + 1, m2 = ((g.o[2] + g.n[2] - 1) >> 2) - b2 + 1;
    const int nblk = (((g.o[0] + g.n[0] - 1) >> 2) - b0 + 1) * m1 * m2;
    if (tid < nblk) {
      const int i = tid / (m1 * m2), r = tid - i * (m1 * m2);
      pre_flag = flags[((b0 + i) * K.nb + (b1 + r / m2)) * K.nb + (b2 + r % m2)];
    }
  }
  {   // ---- the single workgroup-wide pass
    SC_PH(1)
    const bool in = pending;
    const int nyz = g.n[1] * g.n[2];
    const int ci = in ? (tb[0] * g.n[1] + tb[1]) * g.n[2] + tb[2] : 0;
    // ---- counting sort by origin cell (+ compacted list of the non-empty cells)
    for (int i = tid; i < g.vol + 4; i += NM_SC_T) L.cnt[i] = 0;
    for (int i = tid; i < g.vol; i += NM_SC_T) L.tile[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int rank = in ? atomicAdd(&L.cnt[ci], 1) : 0;
    __syncthreads();
    int nruns;
    {
      int v[NM_NPT], sum = 0, nz = 0;
#pragma unroll
      for (int q = 0; q < NM_NPT; ++q) {
        int c = tid * NM_NPT + q;
        v[q] = c < g.vol ? L.cnt[c] : 0;
        sum += v[q];
        nz += v[q] > 0;
      }
      int wtot, wnz;
      int off = wave_excl_scan_i(sum, lane, wtot);
      int roff = wave_excl_scan_i(nz, lane, wnz);
      if (lane == 0) { L.red[wave] = wtot; L.red[NM_SC_NW + wave] = wnz; }
      __syncthreads();
      for (int w2 = 0; w2 < wave; ++w2) { off += L.red[w2]; roff += L.red[NM_SC_NW + w2]; }
      nruns = 0;
#pragma unroll
      for (int w2 = 0; w2 < NM_SC_NW; ++w2) nruns += L.red[NM_SC_NW + w2];
#pragma unroll
      for (int q = 0; q < NM_NPT; ++q) {
        int c = tid * NM_NPT + q;
        if (c < g.vol) {
          L.cnt[c] = off;
          off += v[q];
          if (v[q] > 0) L.run_cell[roff++] = (short)c;
        }
      }
      if (tid == NM_SC_T - 1) L.cnt[g.vol] = off;   // last thread's running offset == total
    }
    __syncthreads();
    SC_PH(2)
    const int slot = in ? L.cnt[ci] + rank : 0;
    const bool owner = tid < nruns;                   // thread r owns non-empty cell r
    const int mycell = owner ? (int)L.run_cell[tid] : 0;
    const int s0 = owner ? L.cnt[mycell] : 0;
    if (flags && cmp && owner && !NM_DBG_BIT(K, 2)) {   // compressed tile: thread r stamps the blocks of cell r, early (see above)
      const int a_ = mycell / nyz, r_ = mycell - a_ * nyz;
      const int b_ = r_ / g.n[2], c_ = r_ - b_ * g.n[2];
      const int o0 = (int)L.ainv[0][a_], o1 = (int)L.ainv[1][b_], o2 = (int)L.ainv[2][c_];
      // the (up to) eight blocks of the cell's stencil: all their stamps are read first, in one round trip - nearly all are
      // current (k_clear carried them over), and one dependent global read per block was 8 k cycles of the two or three
      // workgroups per launch that take this path, which were the ones the launch waited for
      int bid[8], fl[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = (o0 >> 2) + (q >> 2), j = (o1 >> 2) + ((q >> 1) & 1), k = (o2 >> 2) + (q & 1);
        const bool ok = i <= (o0 + 2) >> 2 && j <= (o1 + 2) >> 2 && k <= (o2 + 2) >> 2;
        bid[q] = ok ? (i * K.nb + j) * K.nb + k : -1;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) fl[q] = flags[bid[q] >= 0 ? bid[q] : bid[0]];      // (unconditional: see wg_scatter_f64)
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (bid[q] >= 0 && fl[q] != epoch) mark_block(bid[q], flags, list, count, epoch);
    }
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
      if (in) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int k = 0; k < 3; ++k) L.C[slot * 9 + j * 3 + k] = contrib(i, j, k);
      }
      __syncthreads();
      SC_PH(3)
      // cell sums, in place: thread u handles (cell r, stencil offset q) = (u / 9, u % 9) and leaves the sum of the
      // cell's members in the first member's slot (only this thread touches column q of that cell's slots)
      for (int u = tid; u < nruns * 9; u += NM_SC_T) {
        const int r = u / 9, q = u - 9 * r;
        const int cell = (int)L.run_cell[r];
        const int a0 = L.cnt[cell], a1 = L.cnt[cell + 1];
        // members are read eight at a time with clamped addresses and 0/1 weights: eight independent LDS reads in flight
        // instead of a chain of dependent ones (the loop is LDS-latency bound, ~8 members per cell)
        float4 acc = L.C[a0 * 9 + q];
        for (int s_ = a0 + 1; s_ < a1; s_ += 8) {
          float4 t4[8];
#pragma unroll
          for (int m = 0; m < 8; ++m) t4[m] = L.C[min(s_ + m, a1 - 1) * 9 + q];
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const float wgt = s_ + m < a1 ? 1.f : 0.f;
            acc.x = fmaf(wgt, t4[m].x, acc.x); acc.y = fmaf(wgt, t4[m].y, acc.y);
            acc.z = fmaf(wgt, t4[m].z, acc.z); acc.w = fmaf(wgt, t4[m].w, acc.w);
          }
        }
        L.C[a0 * 9 + q] = acc;
      }
      __syncthreads();
      SC_PH(4)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (owner) {
            const int node = mycell + (i * g.n[1] + j) * g.n[2] + k;
            float4 t4 = L.tile[node];
            const float4 c4 = L.C[s0 * 9 + j * 3 + k];
            t4.x += c4.x; t4.y += c4.y; t4.z += c4.z; t4.w += c4.w;
            L.tile[node] = t4;
          }
          __syncthreads();   // next offset: another cell's target may be this cell's current node
        }
      SC_PH(5)
    }
    // ---- the blocks that receive something join the active list (almost always a single flag read per block: k_clear
    // carried the previous substep's blocks over).  Plain tile: a pass over the tile records in LDS which blocks of the
    // bounding box hold a non-zero node and one thread per such block stamps it - an untouched block stamped here would
    // hold no mass, drop out at the next k_clear and take the atomic path again every substep.  Compressed tile: thread r
    // stamps the blocks of cell r.  Then the flush: one global atomic set per touched node.
    const bool mark = flags != nullptr && !NM_DBG_BIT(K, 2);
    if (mark) {
      if (!cmp) {
        const int b0 = g.o[0] >> 2, b1 = g.o[1] >> 2, b2 = g.o[2] >> 2;
        const int m1 = ((g.o[1] + g.n[1] - 1) >> 2) - b1 + 1, m2 = ((g.o[2] + g.n[2] - 1) >> 2) - b2 + 1;
        const int nblk = (((g.o[0] + g.n[0] - 1) >> 2) - b0 + 1) * m1 * m2;
        int* touched = L.cnt;                     // free after the slab loop
        for (int t = tid; t < nblk; t += NM_SC_T) touched[t] = 0;
        __syncthreads();
        for (int nidx = tid; nidx < g.vol; nidx += NM_SC_T) {
          const float4 t = L.tile[nidx];
          if (t.x != 0.f || t.y != 0.f || t.z != 0.f || t.w != 0.f) {
            const int a_ = nidx / nyz, r = nidx - a_ * nyz;
            const int b_ = r / g.n[2], c_ = r - b_ * g.n[2];
            touched[((((g.o[0] + a_) >> 2) - b0) * m1 + (((g.o[1] + b_) >> 2) - b1)) * m2 + (((g.o[2] + c_) >> 2) - b2)] = 1;
          }
        }
        __syncthreads();
        for (int t = tid; t < nblk; t += NM_SC_T) {
          if (touched[t] == 0 || (t == tid && pre_flag == epoch)) continue;     // already stamped when we looked
          const int i = t / (m1 * m2), r = t - i * (m1 * m2);
          const int j = r / m2, k = r - j * m2;
          mark_block(((b0 + i) * K.nb + (b1 + j)) * K.nb + (b2 + k), flags, list, count, epoch);
        }
      }
    }
    // the atomics go last: nothing waits for them, they drain while other workgroups compute.  One FLOAT per lane, not one
    // node: four lanes cover a node's {mv, m}, a wave instruction sixteen consecutive nodes of the tile's z runs - the memory
    // side adds 64 consecutive floats 4x faster than 64 floats 16 bytes apart (tools/ubench_flush.hip: 314 against 77 G/s),
    // and this flush is what the scatter kernels' duration hangs on (§5)
    // (measured on one box, two builds back to back: k_p2g 24.4 -> 21.8 us; the three-channel adjoint scatter k_g2p_bwd leaves a
    //  quarter of the lanes idle in this mapping and got 1 us slower, so it keeps one node per lane)
    if (NCH != 4) {
    for (int nidx = tid; nidx < g.vol; nidx += NM_SC_T) {
      const float4 t = L.tile[nidx];
      if (t.x != 0.f || t.y != 0.f || t.z != 0.f || t.w != 0.f) {
        int a_ = nidx / nyz, r = nidx - a_ * nyz;
        int b_ = r / g.n[2], c_ = r - b_ * g.n[2];
        const int x_ = cmp ? (int)L.ainv[0][a_] : g.o[0] + a_, y_ = cmp ? (int)L.ainv[1][b_] : g.o[1] + b_,
                  z_ = cmp ? (int)L.ainv[2][c_] : g.o[2] + c_;
        float* dst = (float*)&grid[node_addr(x_, y_, z_, K.nb)];
        unsafeAtomicAdd(dst, t.x);
        unsafeAtomicAdd(dst + 1, t.y);
        unsafeAtomicAdd(dst + 2, t.z);
        if (NCH == 4) unsafeAtomicAdd(dst + 3, t.w);
      }
    }
    } else {
    // (the node's address - two integer divisions and the block / in-block split - once per node into LDS, not once per
    //  float: the largest boxes, 2048 nodes, are the workgroups the launch waits for)
    int* naddr = L.cnt;               // free since the slab loop (the block marking above was its last user)
    __syncthreads();
    for (int nidx = tid; nidx < g.vol; nidx += NM_SC_T) {
      int a_ = nidx / nyz, r = nidx - a_ * nyz;
      int b_ = r / g.n[2], c_ = r - b_ * g.n[2];
      const int x_ = cmp ? (int)L.ainv[0][a_] : g.o[0] + a_, y_ = cmp ? (int)L.ainv[1][b_] : g.o[1] + b_,
                z_ = cmp ? (int)L.ainv[2][c_] : g.o[2] + c_;
      naddr[nidx] = 4 * node_addr(x_, y_, z_, K.nb);
    }
    __syncthreads();
    const float* tilef = (const float*)L.tile;
    float* gridf = (float*)grid;
    for (int idx = tid; idx < 4 * g.vol; idx += NM_SC_T) {
      const float v = tilef[idx];
      if (!NM_DBG_BIT(K, 1) && v != 0.f) unsafeAtomicAdd(gridf + naddr[idx >> 2] + (idx & 3), v);
    }
    }
    SC_PH(6)
  }
  SC_STORE(sc_pass)
}

// Which particle of its workgroup a thread of the scatter kernels takes.  fp64-atomic mode: lanes 2m and 2m + 1 take two
// CONSECUTIVE particles (they usually share a cell and then add once, wg_scatter_f64), and the eight pairs inside a group of 16
// lanes - what one LDS cycle serves - are 32 particles apart in the (cell-ordered) list: different cells, no same-address
// serialisation inside the atomic instruction (2x on the atomic phase, tools/ubench_lds_int.hip).  The sort path re-orders
// the particles in LDS anyway and keeps thread = particle.
__device__ __forceinline__ int scatter_particle(const MpmK& K, int t) {
  static_assert(NM_SC_T == 256, "the permutation is written for 256-particle workgroups");
  return (K.smode == 1 && !NM_DBG_BIT(K, 64)) ? ((((t & 15) >> 1) << 5) | ((t >> 4) << 1) | (t & 1)) : t;
}

// ---------------------------------------------------------------- kernels
__global__ void __launch_bounds__(256) k_clear(float4* __restrict__ gm, float4* __restrict__ gv, float4* __restrict__ gg,
                                               const int* __restrict__ list_prev, const int* __restrict__ count_prev,
                                               int* __restrict__ list_now, int* __restrict__ count_now,
                                               int* __restrict__ count_next, int* __restrict__ flags, int epoch) {
  grid_clear_carry(gm, gv, gg, list_prev, count_prev, list_now, count_now, count_next, flags, epoch, blockIdx.x, gridDim.x);
}

// mpm.py:321-371.  One particle per thread, 256 per workgroup.
__global__ void __launch_bounds__(NM_SC_T) k_p2g(MpmK K, int n, const float* __restrict__ vol, const float* __restrict__ rho,
                                                 const int* __restrict__ enabled, const float* __restrict__ x,
                                                 const float* __restrict__ v, const float* __restrict__ C,
                                                 const float* __restrict__ S, float4* __restrict__ gm, int* flags, int* list,
                                                 int* count, int epoch, const int* __restrict__ skip_hdr) {
  __shared__ ScatterLds L;
  if (skip_hdr && *skip_hdr >= 0) return;   // the grid of this substep was restored from a cache record
  const int lp = scatter_particle(K, threadIdx.x);
  const int p = blockIdx.x * K.ppw + lp;
  // every load of the particle is issued before the first one is waited for - `enabled` included: a branch on it in front of
  // the others made two HBM round trips out of one (the kernel is a chain of latencies, not of bytes)
  const bool mine = lp < K.ppw && p < n;
  Stencil st;
  float pm = 0.f, mom[3] = {0.f, 0.f, 0.f};
  M3 A = m3_zero();
  bool en = false;
  if (mine) {
    const int e = enabled[p];
    const float vl = vol[p], rh = rho[p];
    const float xp[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
    const float vp[3] = {v[3 * p], v[3 * p + 1], v[3 * p + 2]};
    const M3 Sp = m3_load(S + 9 * p), Cp = m3_load(C + 9 * p);
    en = e != 0;
    make_stencil(K, xp, st);
    pm = vl * rh;
    const float ks = -K.dt * vl * 4.0f * K.inv_dx * K.inv_dx;  // mpm.py:357
#pragma unroll
    for (int i = 0; i < 9; ++i) A.m[i] = ks * Sp.m[i] + pm * Cp.m[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) mom[a] = pm * vp[a];
  }
  if (!en) {
    st.b[0] = st.b[1] = st.b[2] = 0;
    pm = 0.f;
    A = m3_zero();
#pragma unroll
    for (int a = 0; a < 3; ++a) { mom[a] = 0.f; st.f[a] = 0.f; st.w[a][0] = st.w[a][1] = st.w[a][2] = 0.f; }
  }
  auto contrib = [&](int i, int j, int k) -> float4 {
    float d0 = ((float)i - st.f[0]) * K.dx, d1 = ((float)j - st.f[1]) * K.dx, d2 = ((float)k - st.f[2]) * K.dx;
    float w = sel3(st.w[0], i) * st.w[1][j] * st.w[2][k];
    return make_float4(w * (mom[0] + A.m[0] * d0 + A.m[1] * d1 + A.m[2] * d2),
                       w * (mom[1] + A.m[3] * d0 + A.m[4] * d1 + A.m[5] * d2),
                       w * (mom[2] + A.m[6] * d0 + A.m[7] * d1 + A.m[8] * d2), w * pm);
  };
  wg_scatter<4>(K, en, lp, st.b, gm, flags, list, count, epoch, L, contrib);
}

static inline size_t gridrec_list_bytes(int cap) { return ((size_t)cap * sizeof(int) + 255) & ~(size_t)255; }
static inline size_t gridrec_bytes(int cap) { return 256 + gridrec_list_bytes(cap) + (size_t)cap * 64 * sizeof(float4); }
static inline GridRec gridrec_at(void* base, int cap) {
  GridRec r;
  char* p = (char*)base;
  r.hdr = (int*)p;
  r.list = (int*)(p + 256);
  r.gm = (float4*)(p + 256 + gridrec_list_bytes(cap));
  return r;
}

// blocks of the previous substep's list whose velocities the clear left in place (GridPrologue keep_gv); list == NULL: none
struct DroppedBlocks {
  const int* list;
  const int* count;
  const int* flags;
  int epoch;
};
// mpm.py:373-429 on the active blocks only; optionally saves the pre-grid-op node values into a cache record
__global__ void __launch_bounds__(256) k_grid_op(MpmK K, const float4* __restrict__ gm, float4* __restrict__ gv,
                                                 const int* __restrict__ list, const int* __restrict__ count, GridRec rec,
                                                 int cap, const int* __restrict__ skip_hdr, int* __restrict__ status,
                                                 const int* __restrict__ slot, const float4* __restrict__ xbuf,
                                                 DroppedBlocks dropped) {
  // slot / xbuf (sharded roll-out): a block with slot[b] >= 0 takes its {mv, m} - summed over the ranks - from the exchange
  // buffer instead of this rank's grid (the unpack step of the exchange, fused)
  if (skip_hdr && *skip_hdr >= 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // (the wave's first list entry is requested together with the count - the list holds one word per block of the grid, any
  //  index below nblocks is readable -: count -> list -> node was three dependent round trips for a 5 us kernel)
  const int b_first = (int)(blockIdx.x * 4 + wave) < K.nb * K.nb * K.nb ? list[blockIdx.x * 4 + wave] : 0;
  const int cnt = *count;
  if (dropped.list) {   // a block of the previous list that is not in this substep's (no stamp of this epoch) reads as empty again
    const int pc = *dropped.count;
    for (int li = blockIdx.x * 4 + wave; li < pc; li += gridDim.x * 4) {
      const int b = dropped.list[li];
      if (dropped.flags[b] != dropped.epoch) gv[(b << 6) + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const bool save = rec.hdr != nullptr && cnt <= cap;
  if (rec.hdr && blockIdx.x == 0 && threadIdx.x == 0) {
    rec.hdr[0] = save ? cnt : -1;
    if (!save && status) atomicOr(status, 4);   // sharded substeps cannot fall back to a recompute: tell the caller
  }
  for (int li = blockIdx.x * 4 + wave; li < cnt; li += gridDim.x * 4) {
    int b = li == blockIdx.x * 4 + wave ? b_first : list[li];
    int i, j, k;
    block_coords(b, K.nb, lane, i, j, k);
    int node = (b << 6) + lane;
    const int sl = slot ? slot[b] : -1;
    float4 a = sl >= 0 ? xbuf[(sl << 6) + lane] : gm[node];
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < K.G && j < K.G && k < K.G) {
      float u[3], mk[3];
      grid_velocity(K, i, j, k, a, u, mk);
      out.x = u[0] * mk[0]; out.y = u[1] * mk[1]; out.z = u[2] * mk[2];
    }
    gv[node] = out;
    if (save) {
      rec.gm[(li << 6) + lane] = a;
      if (lane == 0) rec.list[li] = b;
    }
  }
}

__global__ void __launch_bounds__(256) k_grid_restore(MpmK K, GridRec rec, float4* __restrict__ gm, float4* __restrict__ gv,
                                                      int* __restrict__ list, int* __restrict__ count) {
  grid_restore(K, rec, gm, gv, list, count, blockIdx.x, gridDim.x);
}

// adjoint of grid_op: gg {vbar} -> {mvbar, mbar}
// stamp != null (verified reverse sweep of a roll-out): also marks the blocks of the NEXT record of the sweep with the
// epoch its restore will run under, which is what lets that restore share one pass with the clear (GridPrologue mode 2)
// (the pointers the first loads go through come first: seven of them are what the dispatcher can hand over in registers -
//  -amdgpu-kernarg-preload-count - so that the list / count requests need not wait for the kernel-argument fetch)
__global__ void __launch_bounds__(256) k_grid_op_bwd(const int* __restrict__ list, const int* __restrict__ count,
                                                     const float4* __restrict__ gm, float4* __restrict__ gg,
                                                     int* __restrict__ flags, const int* __restrict__ slot,
                                                     const float4* __restrict__ xbuf, MpmK K, GridRec stamp, int stamp_epoch) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b_first = (int)(blockIdx.x * 4 + wave) < K.nb * K.nb * K.nb ? list[blockIdx.x * 4 + wave] : 0;      // (with the count, not behind it: see k_grid_op)
  const int cnt = *count;
  if (stamp.hdr) {
    const int sc = stamp.hdr[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < sc; i += gridDim.x * blockDim.x) flags[stamp.list[i]] = stamp_epoch;
  }
  for (int li = blockIdx.x * 4 + wave; li < cnt; li += gridDim.x * 4) {
    int b = li == blockIdx.x * 4 + wave ? b_first : list[li];
    int i, j, k;
    block_coords(b, K.nb, lane, i, j, k);
    int node = (b << 6) + lane;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < K.G && j < K.G && k < K.G) {
      float4 a = gm[node];
      if (a.w > 0.f) {
        float u[3], mk[3];
        grid_velocity(K, i, j, k, a, u, mk);
        const int sl = slot ? slot[b] : -1;
        float4 gb = sl >= 0 ? xbuf[(sl << 6) + lane] : gg[node];     // (summed over the ranks)
        float inv = 1.f / (a.w + K.eps);
        float ux = gb.x * mk[0], uy = gb.y * mk[1], uz = gb.z * mk[2];
        out.x = ux * inv; out.y = uy * inv; out.z = uz * inv;
        out.w = -(ux * a.x + uy * a.y + uz * a.z) * inv * inv;
      }
    }
    gg[node] = out;
  }
}

// mpm.py:432-498 (body: g2p_particle, nm_grid.h)
__global__ void __launch_bounds__(256, 3) k_g2p(MpmK K, int n, const float* __restrict__ clip, const int* __restrict__ enabled,
                                             const float* x, const float* v, const float* C, const float* F,
                                             const float4* __restrict__ gv, float* xn, float* vn, float* Cn, float* Fn, int fresh) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  M3 Fo;
  g2p_particle<false>(K, p, clip, enabled, x, v, C, F, gv, xn, vn, Cn, Fo, nullptr, 0, fresh != 0);
  if (enabled[p] != 0 || (fresh && Fn != F)) m3_store(Fn + 9 * p, Fo);      // a disabled row is left alone (mpm.py:443-444)
}

// g2p onto a passive particle set (MPMModel.forward_extra, mpm.py:260-277): in place, untouched blocks read as BC(g dt)
__global__ void __launch_bounds__(256, 3) k_g2p_extra(MpmK K, int n, const float* __restrict__ clip, const int* __restrict__ enabled,
                                                   float* x, float* v, float* C, float* F, const float4* __restrict__ gv,
                                                   const int* __restrict__ flags, int epoch) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  M3 Fo;
  g2p_particle<false, true>(K, p, clip, enabled, x, v, C, F, gv, x, v, C, Fo, flags, epoch);
  if (enabled[p] != 0) m3_store(F + 9 * p, Fo);
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
  // (all loads issued before the first is waited for, `enabled` included - see k_p2g)
  bool active = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) { q.vt[a] = 0.f; q.xbar[a] = 0.f; }
  q.Ct = m3_zero();
  q.Fbar = m3_zero();
  if (p < n) {
    int e = enabled[p];
    float xp[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
    float vn[3] = {vnext[3 * p], vnext[3 * p + 1], vnext[3 * p + 2]};
    float gxv[3] = {gxn[3 * p], gxn[3 * p + 1], gxn[3 * p + 2]};
    float gvv[3] = {gvn[3 * p], gvn[3 * p + 1], gvn[3 * p + 2]};
    float clp = clip[p];
    M3 Fp = m3_load(F + 9 * p), gFp = m3_load(gFn + 9 * p), Cn = m3_load(Cnext + 9 * p), gCp = m3_load(gCn + 9 * p);
    // (the comment above was not what the compiler made of it: every load but `enabled` had been sunk into `if (active)`)
    nm_pin(e); nm_pin(clp); nm_pin(Fp); nm_pin(gFp); nm_pin(Cn); nm_pin(gCp);
#pragma unroll
    for (int a = 0; a < 3; ++a) { nm_pin(xp[a]); nm_pin(vn[a]); nm_pin(gxv[a]); nm_pin(gvv[a]); }
    const float bnd = clp * K.dx;
    active = e != 0;
    if (active) {
      make_stencil(K, xp, q.st);
      const float lo = 0.0f + bnd, hi = 1.0f - bnd;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float t = xp[a] + K.dt * vn[a];
        float xe = (t >= lo && t <= hi) ? gxv[a] : 0.f;  // clamp passes the gradient only inside
        q.xbar[a] = xe;
        q.vt[a] = gvv[a] + K.dt * xe;
      }
      M3 T = Cn;
#pragma unroll
      for (int i = 0; i < 9; ++i) T.m[i] *= K.dt;
      T.m[0] += 1.f; T.m[4] += 1.f; T.m[8] += 1.f;
      q.Fbar = m3_mul_tn(T, gFp);           // (I + dt C')^T Fbar'
      M3 FF = m3_mul_nt(gFp, Fp);           // Fbar' F^T
#pragma unroll
      for (int i = 0; i < 9; ++i) q.Ct.m[i] = gCp.m[i] + K.dt * FF.m[i];
    }
  }
  if (!active) {
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

__global__ void __launch_bounds__(NM_SC_T) k_g2p_bwd(MpmK K, int n, const float* __restrict__ clip, const int* __restrict__ enabled,
                                                     const float* __restrict__ x, const float* __restrict__ F,
                                                     const float* __restrict__ vnext, const float* __restrict__ Cnext,
                                                     const float* __restrict__ gxn, const float* __restrict__ gvn,
                                                     const float* __restrict__ gCn, const float* __restrict__ gFn,
                                                     const float4* __restrict__ gv, float4* __restrict__ gg,
                                                     float* __restrict__ gx, float* __restrict__ gF) {
  __shared__ ScatterLds L;
  const float kap = 4.0f * K.inv_dx * K.inv_dx;
  const int lp = scatter_particle(K, threadIdx.x);
  const int p = lp < K.ppw ? blockIdx.x * K.ppw + lp : n;      // (a thread beyond the workgroup's share owns no particle)
  G2pBwdP q;
  const bool active = g2p_bwd_particle(K, n, p, clip, enabled, x, F, vnext, Cnext, gxn, gvn, gCn, gFn, q);
  // (1) per-particle outputs: gF and gx (direct + through weights/dpos, gathering the forward grid velocity)
  if (active) {
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {      // (27 gathers in flight at once cost an occupancy step: 20.7 -> 26.5 us, measured again in round 5)
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
    for (int a = 0; a < 3; ++a) gx[3 * p + a] = q.xbar[a];   // completed (and sanitised) by k_p2g_bwd
#pragma unroll
    for (int a = 0; a < 9; ++a) q.Fbar.m[a] = nm_finite_or_zero(q.Fbar.m[a]);   // interface.py:65-74
    m3_store(gF + 9 * p, q.Fbar);
  }
  // (2) scatter of the node-velocity adjoint
  auto contrib = [&](int i, int j, int k) -> float4 {
    float d0 = ((float)i - q.st.f[0]) * K.dx, d1 = ((float)j - q.st.f[1]) * K.dx, d2 = ((float)k - q.st.f[2]) * K.dx;
    float w = sel3(q.st.w[0], i) * q.st.w[1][j] * q.st.w[2][k];
    float kw = kap * w;
    float c0_ = q.Ct.m[0] * d0 + q.Ct.m[1] * d1 + q.Ct.m[2] * d2;
    float c1_ = q.Ct.m[3] * d0 + q.Ct.m[4] * d1 + q.Ct.m[5] * d2;
    float c2_ = q.Ct.m[6] * d0 + q.Ct.m[7] * d1 + q.Ct.m[8] * d2;
    return make_float4(w * q.vt[0] + kw * c0_, w * q.vt[1] + kw * c1_, w * q.vt[2] + kw * c2_, 0.f);
  };
  wg_scatter<3>(K, active, lp, q.st.b, gg, nullptr, nullptr, nullptr, 0, L, contrib);
}

// adjoint of p2g: gathers {mvbar, mbar}; writes gv, gC, gS and adds to gx.  These are the substep's returned gradients:
// non-finite values become zeros here, per substep, exactly where interface.py:65-74 applies nan_to_num_ - so that the
// fused roll-out (which never surfaces per-substep gradients to torch) behaves like the per-operator path.
__global__ void __launch_bounds__(256, 2) k_p2g_bwd(MpmK K, int n, const float* __restrict__ vol, const float* __restrict__ rho,
                                                 const int* __restrict__ enabled, const float* __restrict__ x,
                                                 const float* __restrict__ v, const float* __restrict__ C,
                                                 const float* __restrict__ S, const float4* __restrict__ gg,
                                                 float* __restrict__ gx, float* __restrict__ gvp, float* __restrict__ gC,
                                                 float* __restrict__ gS) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  // (all of the particle's loads are issued before `enabled` is looked at: one HBM round trip instead of two)
  int e_ = enabled[p];
  float xp[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
  float vp[3] = {v[3 * p], v[3 * p + 1], v[3 * p + 2]};
  float vl = vol[p];
  float rh = rho[p];
  M3 Sp = m3_load(S + 9 * p), Cp = m3_load(C + 9 * p), A;
  // (... which the compiler undid: every load but `enabled` sat behind the branch below - nm_pin keeps them here)
  nm_pin(e_); nm_pin(vl); nm_pin(rh); nm_pin(Sp); nm_pin(Cp);
#pragma unroll
  for (int a = 0; a < 3; ++a) { nm_pin(xp[a]); nm_pin(vp[a]); }
  if (e_ == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { gvp[3 * p + a] = 0.f; gx[3 * p + a] = nm_finite_or_zero(gx[3 * p + a]); }
    m3_store(gC + 9 * p, m3_zero());
    m3_store(gS + 9 * p, m3_zero());
    return;
  }
  Stencil st;
  make_stencil(K, xp, st);
  float pm = vl * rh;
  float ks = -K.dt * vl * 4.0f * K.inv_dx * K.inv_dx;
#pragma unroll
  for (int i = 0; i < 9; ++i) A.m[i] = ks * Sp.m[i] + pm * Cp.m[i];
  float mom[3] = {pm * vp[0], pm * vp[1], pm * vp[2]};
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
    gx[3 * p + a] = nm_finite_or_zero(gx[3 * p + a] + xb[a]);
    gvp[3 * p + a] = nm_finite_or_zero(pm * vb[a]);
  }
  M3 o;
#pragma unroll
  for (int i = 0; i < 9; ++i) o.m[i] = nm_finite_or_zero(pm * Ab.m[i]);
  m3_store(gC + 9 * p, o);
#pragma unroll
  for (int i = 0; i < 9; ++i) o.m[i] = nm_finite_or_zero(ks * Ab.m[i]);
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

// dense (G,G,G) view of the block-sparse grid, as the reference's arrays read after grid_op: blocks the last substep did
// not touch hold m = mv = 0 and v = BC(g dt) (the dense sweep of mpm.py:373-429 visits every node)
__global__ void k_grid_export(MpmK K, const float4* __restrict__ gm, const float4* __restrict__ gv, float* mv, float* m,
                              float* v, const int* __restrict__ flags, int epoch) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int G = K.G;
  if (t >= G * G * G) return;
  int i = t / (G * G), r = t - i * G * G, j = r / G, k = r - j * G;
  int a = node_addr(i, j, k, K.nb);
  float4 q = gm[a], u = gv[a];
  if (flags[((i >> 2) * K.nb + (j >> 2)) * K.nb + (k >> 2)] != epoch) {
    q = make_float4(0.f, 0.f, 0.f, 0.f);
    float w[3], mk[3];
    grid_velocity(K, i, j, k, q, w, mk);
    u = make_float4(w[0] * mk[0], w[1] * mk[1], w[2] * mk[2], 0.f);
  }
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
#ifdef NM_PHASES
  K.dbg = getenv("NM_DBG") ? atoi(getenv("NM_DBG")) : 0;
  K.maxpass = getenv("NM_MAXPASS") ? atoi(getenv("NM_MAXPASS")) : NM_WT_MAXPASS;
#else
  K.dbg = 0;
  K.maxpass = NM_WT_MAXPASS;
#endif
  {
    const char* sm = getenv("NEUMA_SCATTER");
    K.smode = (sm && (!strcmp(sm, "sort") || !strcmp(sm, "0"))) ? 0 : 1;
  }
  K.ppw = NM_SC_T;
  h->nblocks = K.nb * K.nb * K.nb;
  size_t nodes = (size_t)h->nblocks * 64;
  h->gm = h->gv = h->gg = nullptr;
  NM_HIP_CHECK(hipMalloc(&h->gm, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMalloc(&h->gv, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMalloc(&h->gg, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMalloc(&h->flags, h->nblocks * sizeof(int)));
  for (int i = 0; i < 3; ++i) NM_HIP_CHECK(hipMalloc(&h->list[i], h->nblocks * sizeof(int)));
  NM_HIP_CHECK(hipMalloc(&h->count, 8 * sizeof(int)));
  NM_HIP_CHECK(hipMemset(h->gm, 0, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMemset(h->gv, 0, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMemset(h->gg, 0, nodes * sizeof(float4)));
  NM_HIP_CHECK(hipMemset(h->flags, 0, h->nblocks * sizeof(int)));
  NM_HIP_CHECK(hipMemset(h->count, 0, 8 * sizeof(int)));
  NM_HIP_CHECK(hipDeviceSynchronize());
  h->cur = 0;
  h->epoch = 0;
  h->gv_stale = 0;
  h->resident_rec = nullptr; h->resident_epoch = -1;
  h->sh_cnt = h->sh_pos = nullptr;
  *out = h;
  return NM_OK;
}

float nm_mpm_get_dt(const nm_mpm* h) { return h->k.dt; }

extern "C" int nm_mpm_destroy(nm_mpm* h) {
  if (!h) return NM_OK;
  hipFree(h->gm); hipFree(h->gv); hipFree(h->gg); hipFree(h->flags);
  hipFree(h->list[0]); hipFree(h->list[1]); hipFree(h->list[2]); hipFree(h->count);
  if (h->sh_cnt) hipFree(h->sh_cnt);
  if (h->sh_pos) hipFree(h->sh_pos);
  if (h->sh_slot) hipFree(h->sh_slot);
  if (h->sh_dil) hipFree(h->sh_dil);
  delete h;
  return NM_OK;
}

// Particles per workgroup of the scatter kernels.  fp64-atomic mode: the LDS atomic unit of a CU is what bounds them, so the
// chunks are sized for a whole number of workgroups per CU (256 CUs) - 100 000 particles: 512 workgroups of 196 instead of 391
// of 256, i.e. two on every CU instead of two on 135 CUs and one on the rest.  The sort path keeps 256.
static int scatter_ppw(const nm_mpm* h, int n) {
  if (h->k.smode != 1) return NM_SC_T;
  const int per_cu = nm_div_up(n, 256 * NM_SC_T);
  int ppw = nm_div_up(n, 256 * (per_cu > 0 ? per_cu : 1));
  ppw = ((ppw + 3) / 4) * 4;
  return ppw < 128 ? 128 : (ppw > NM_SC_T ? NM_SC_T : ppw);
}
static int scatter_grid(const nm_mpm* h, int n) { return nm_div_up(n, scatter_ppw(h, n)); }
static MpmK scatter_k(const nm_mpm* h, int n) {
  MpmK K = h->k;
  K.ppw = scatter_ppw(h, n);
  return K;
}
#ifndef NM_SWEEP_GRID
#define NM_SWEEP_GRID 512
#endif
static const int kSweepGrid = NM_SWEEP_GRID;  // workgroups for the active-block sweeps (grid-stride over the list)

// clear + p2g + grid_op (shared by forward, backward-recompute and forward_extra).
// save != null: the forward pass also writes a grid cache record.  restore != null: the reverse sweep restores the
// grid from the record; p2g / grid_op are still enqueued but return at once unless the record is marked invalid.
static void mpm_rotate(nm_mpm* h, int& prev, int& now, int& next) {
  prev = h->cur; now = (prev + 1) % 3; next = (prev + 2) % 3;
  h->epoch += 1;
  h->cur = now;
}

// the previous substep's list, if the clear left its velocities in place (consumed by the grid update launched next)
static DroppedBlocks take_dropped(nm_mpm* h, int now) {
  DroppedBlocks d = {nullptr, nullptr, nullptr, 0};
  if (h->gv_stale) {
    const int before = (now + 2) % 3;
    d.list = h->list[before]; d.count = h->count + before; d.flags = h->flags; d.epoch = h->epoch;
    h->gv_stale = 0;
  }
  return d;
}

static int mpm_build_grid(nm_mpm* h, int n, const nm_statics* st, const nm_particles* cur, hipStream_t s, void* save = nullptr,
                          const void* restore = nullptr, int cap = 0, bool restore_verified = false, bool precleared = false) {
  int prev, now, next;
  if (precleared) {   // a GridPrologue (nm_mpm_prologue_forward) has rotated the lists and cleared the grid already
    now = h->cur;
  } else {
    mpm_rotate(h, prev, now, next);
    NM_LAUNCH(k_clear, dim3(NM_CLEAR_WGS), dim3(256), 0, s, h->gm, h->gv, h->gg, h->list[prev], h->count + prev, h->list[now],
                       h->count + now, h->count + next, h->flags, h->epoch);
    NM_LAUNCH_CHECK();
  }
  GridRec none = {nullptr, nullptr, nullptr};
  GridRec srec = save ? gridrec_at(save, cap) : none;
  const int* skip = nullptr;
  if (restore) {
    GridRec rrec = gridrec_at(const_cast<void*>(restore), cap);
    NM_LAUNCH(k_grid_restore, dim3(kSweepGrid), dim3(256), 0, s, h->k, rrec, h->gm, h->gv, h->list[now], h->count + now);
    NM_LAUNCH_CHECK();
    skip = rrec.hdr;
    if (restore_verified) return NM_OK;   // the host has seen this record's header: it is valid, nothing to fall back to
  }
  if (n > 0) {
    NM_LAUNCH(k_p2g, dim3(scatter_grid(h, n)), dim3(NM_SC_T), 0, s, scatter_k(h, n), n, st->vol, st->rho, st->enabled, cur->x,
                       cur->v, cur->C, cur->stress, h->gm, h->flags, h->list[now], h->count + now, h->epoch, skip);
    NM_LAUNCH_CHECK();
  }
  const DroppedBlocks dropped = take_dropped(h, now);
  NM_LAUNCH(k_grid_op, dim3(kSweepGrid), dim3(256), 0, s, h->k, h->gm, h->gv, h->list[now], h->count + now, srec, cap, skip,
            (int*)nullptr, (const int*)nullptr, (const float4*)nullptr, dropped);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

// ---- roll-out only: the clear / restore of a substep rides in the prologue of the constitutive kernel in front of it
int nm_mpm_prologue_forward(nm_mpm* h, GridPrologue* g, bool keep_gv) {
  int prev, now, next;
  mpm_rotate(h, prev, now, next);
  g->mode = 1;
  g->keep_gv = keep_gv ? 1 : 0;
  h->gv_stale = keep_gv ? 1 : 0;
  g->K = h->k;
  g->gm = h->gm; g->gv = h->gv; g->gg = h->gg;
  g->list_prev = h->list[prev]; g->count_prev = h->count + prev;
  g->list_now = h->list[now]; g->count_now = h->count + now; g->count_next = h->count + next;
  g->flags = h->flags; g->epoch = h->epoch;
  g->rec.hdr = nullptr; g->rec.list = nullptr; g->rec.gm = nullptr;
  return NM_OK;
}
int nm_mpm_prologue_backward(nm_mpm* h, const void* gridrec, int cap, GridPrologue* g) {
  NM_REQUIRE(gridrec && cap > 0, "prologue restore needs a grid cache record");
  nm_mpm_prologue_forward(h, g);
  g->mode = 2;
  g->rec = gridrec_at(const_cast<void*>(gridrec), cap);
  return NM_OK;
}

extern "C" size_t nm_mpm_gridcache_bytes(int32_t cap_blocks) { return cap_blocks > 0 ? gridrec_bytes(cap_blocks) : 0; }

static int check_particles(const nm_statics* st, const nm_particles* p, bool need_stress) {
  NM_REQUIRE(st && st->vol && st->rho && st->clip_bound && st->enabled, "null statics");
  NM_REQUIRE(p && p->x && p->v && p->C && p->F, "null particle arrays");
  if (need_stress) NM_REQUIRE(p->stress, "null stress");
  return NM_OK;
}

extern "C" int nm_mpm_forward(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* next,
                              void* stream) {
  return nm_mpm_forward_ex(h, n, st, cur, next, nullptr, 0, stream);
}

static int mpm_forward_impl(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* next,
                            void* gridrec, int32_t cap_blocks, bool precleared, void* stream) {
  NM_REQUIRE(h, "null handle");
  NM_REQUIRE(!gridrec || cap_blocks > 0, "grid cache record without capacity");
  NM_REQUIRE(n >= 0, "negative particle count");
  if (n == 0) return NM_OK;  // empty input: nothing to scatter or gather
  int rc = check_particles(st, cur, true);
  if (rc) return rc;
  if (next) {
    rc = check_particles(st, next, false);
    if (rc) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  rc = mpm_build_grid(h, n, st, cur, s, gridrec, nullptr, cap_blocks, false, precleared);
  if (rc) return rc;
  h->resident_rec = gridrec; h->resident_epoch = h->epoch;
  if (!next) return NM_OK;   // g2p is performed by the caller's next kernel (nm_mpm_g2p_fuse)
  NM_LAUNCH(k_g2p, dim3(nm_div_up(n, 256)), dim3(256), 0, s, h->k, n, st->clip_bound, st->enabled, cur->x,
                     cur->v, cur->C, cur->F, h->gv, next->x, next->v, next->C, next->F, h->fresh_rows);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" int nm_mpm_forward_ex(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* next,
                                 void* gridrec, int32_t cap_blocks, void* stream) {
  return mpm_forward_impl(h, n, st, cur, next, gridrec, cap_blocks, false, stream);
}

// roll-out: the grid was cleared by a GridPrologue (nm_mpm_prologue_forward) in the kernel launched just before
int nm_mpm_forward_prepared(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* next, void* gridrec,
                            int32_t cap_blocks, void* stream) {
  return mpm_forward_impl(h, n, st, cur, next, gridrec, cap_blocks, true, stream);
}

int nm_mpm_forward_prepared_nog2p(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, void* gridrec,
                                  int32_t cap_blocks, void* stream) {
  return mpm_forward_impl(h, n, st, cur, nullptr, gridrec, cap_blocks, true, stream);
}
int nm_mpm_g2p_fuse(nm_mpm* h, const nm_statics* st, const nm_particles* cur, nm_particles* next, G2pFuse* f) {
  int rc = check_particles(st, cur, true);
  if (rc) return rc;
  NM_REQUIRE(next && next->x && next->v && next->C, "null next state");
  f->gv = h->gv; f->K = h->k;
  f->clip = st->clip_bound; f->enabled = st->enabled;
  f->x = cur->x; f->v = cur->v; f->C = cur->C; f->F = cur->F;
  f->xn = next->x; f->vn = next->v; f->Cn = next->C;
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
    NM_LAUNCH(k_g2p_extra, dim3(nm_div_up(n_extra, 256)), dim3(256), 0, s, h->k, n_extra, st_extra->clip_bound,
                          st_extra->enabled, extra->x, extra->v, extra->C, extra->F, h->gv, h->flags, h->epoch);
    NM_LAUNCH_CHECK();
  }
  return NM_OK;
}

extern "C" int nm_mpm_backward(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur,
                               const nm_particles* next, const nm_particles* gnext, nm_particles* gcur, void* stream) {
  return nm_mpm_backward_ex(h, n, st, cur, next, gnext, gcur, nullptr, 0, stream);
}

extern "C" int nm_mpm_backward_ex(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur,
                                  const nm_particles* next, const nm_particles* gnext, nm_particles* gcur,
                                  const void* gridrec, int32_t cap_blocks, void* stream) {
  return nm_mpm_backward_cached(h, n, st, cur, next, gnext, gcur, gridrec, cap_blocks, false, false, nullptr, stream);
}

// verified: the host has seen the record's header (valid) - no fall-back launches.  prepared: a GridPrologue
// (nm_mpm_prologue_backward) restored the grid in the kernel launched just before.  stamp_rec: record of the substep the
// sweep visits next; its blocks get flagged for that substep's prologue.
int nm_mpm_backward_cached_begin(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, const nm_particles* next,
                                 const nm_particles* gnext, nm_particles* gcur, const void* gridrec, int32_t cap_blocks,
                                 bool verified, bool prepared, void* stream) {
  NM_REQUIRE(h, "null handle");
  NM_REQUIRE(!gridrec || cap_blocks > 0, "grid cache record without capacity");
  NM_REQUIRE(n >= 0, "negative particle count");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {      // (a rank without particles of a sharded roll-out: an empty grid, but the lists still rotate)
    if (!prepared) return mpm_build_grid(h, 0, st, cur, s, nullptr, gridrec, cap_blocks, verified && gridrec != nullptr);
    return NM_OK;
  }
  int rc = check_particles(st, cur, true);
  if (rc) return rc;
  NM_REQUIRE(next && next->v && next->C, "next state (v, C) required");
  NM_REQUIRE(gnext && gnext->x && gnext->v && gnext->C && gnext->F, "null incoming gradients");
  NM_REQUIRE(gcur && gcur->x && gcur->v && gcur->C && gcur->F && gcur->stress, "null outgoing gradients");
  // The reverse sweep usually starts right behind the forward one: the grid still holds the last substep - {mv, m}, velocities,
  // active list, a clean adjoint array (nothing but a reverse substep writes it, and each one cleans up behind the previous) -
  // exactly what clear + restore would rebuild from the (valid) record.  Two launches less per roll-out; for the one-substep
  // configurations that is all of them.
  const bool resident = !prepared && verified && gridrec != nullptr && gridrec == h->resident_rec && h->epoch == h->resident_epoch;
  if (!prepared && !resident) {
    rc = mpm_build_grid(h, n, st, cur, s, nullptr, gridrec, cap_blocks, verified && gridrec != nullptr);  // recompute (mpm.py:312-315) or restore
    if (rc) return rc;
  }
  h->resident_rec = nullptr;      // (the adjoint scatter below dirties the grid's adjoint array: resident no more)
  NM_LAUNCH(k_g2p_bwd, dim3(scatter_grid(h, n)), dim3(NM_SC_T), 0, s, scatter_k(h, n), n, st->clip_bound, st->enabled, cur->x, cur->F, next->v,
                     next->C, gnext->x, gnext->v, gnext->C, gnext->F, h->gv, h->gg, gcur->x, gcur->F);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
// xbuf != NULL (sharded roll-out): the node-velocity adjoint of the blocks with an exchange slot comes, summed over the
// ranks, from there
int nm_mpm_backward_cached_finish(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* gcur,
                                  const void* stamp_rec, int32_t cap_blocks, const float* xbuf, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int now = h->cur;
  GridRec stamp = {nullptr, nullptr, nullptr};
  if (stamp_rec) stamp = gridrec_at(const_cast<void*>(stamp_rec), cap_blocks);
  NM_LAUNCH(k_grid_op_bwd, dim3(kSweepGrid), dim3(256), 0, s, (const int*)h->list[now], (const int*)(h->count + now), (const float4*)h->gm, h->gg,
                     h->flags, (const int*)(xbuf ? h->sh_slot : nullptr), (const float4*)xbuf, h->k, stamp, h->epoch + 1);
  NM_LAUNCH_CHECK();
  if (n == 0) return NM_OK;
  NM_LAUNCH(k_p2g_bwd, dim3(nm_div_up(n, 256)), dim3(256), 0, s, h->k, n, st->vol, st->rho, st->enabled, cur->x, cur->v, cur->C,
                     cur->stress, h->gg, gcur->x, gcur->v, gcur->C, gcur->stress);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
int nm_mpm_backward_cached(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, const nm_particles* next,
                           const nm_particles* gnext, nm_particles* gcur, const void* gridrec, int32_t cap_blocks, bool verified,
                           bool prepared, const void* stamp_rec, void* stream) {
  if (n == 0) return NM_OK;
  int rc = nm_mpm_backward_cached_begin(h, n, st, cur, next, gnext, gcur, gridrec, cap_blocks, verified, prepared, stream);
  if (rc) return rc;
  return nm_mpm_backward_cached_finish(h, n, st, cur, gcur, stamp_rec, cap_blocks, nullptr, stream);
}

// sharded roll-out, forward: this rank's scatter alone (the grid was cleared by a GridPrologue), then - after the exchange -
// the grid update, which takes the blocks with an exchange slot from xbuf and writes the substep's cache record
int nm_mpm_forward_prepared_p2g(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, void* stream) {
  if (n == 0) return NM_OK;
  int rc = check_particles(st, cur, true);
  if (rc) return rc;
  const int now = h->cur;
  NM_LAUNCH(k_p2g, dim3(scatter_grid(h, n)), dim3(NM_SC_T), 0, (hipStream_t)stream, scatter_k(h, n), n, st->vol, st->rho, st->enabled, cur->x,
                     cur->v, cur->C, cur->stress, h->gm, h->flags, h->list[now], h->count + now, h->epoch, (const int*)nullptr);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
int nm_mpm_forward_gridop_x(nm_mpm* h, void* gridrec, int32_t cap_blocks, int32_t* status, const float* xbuf, void* stream) {
  NM_REQUIRE(!gridrec || cap_blocks > 0, "grid cache record without capacity");
  const int now = h->cur;
  GridRec none = {nullptr, nullptr, nullptr};
  const DroppedBlocks dropped = take_dropped(h, now);
  NM_LAUNCH(k_grid_op, dim3(kSweepGrid), dim3(256), 0, (hipStream_t)stream, h->k, h->gm, h->gv, h->list[now], h->count + now,
                     gridrec ? gridrec_at(gridrec, cap_blocks) : none, cap_blocks, (const int*)nullptr, (int*)status,
                     (const int*)(xbuf ? h->sh_slot : nullptr), (const float4*)xbuf, dropped);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
// clear + rotate for a rank without particles (no constitutive kernel carries the prologue there)
int nm_mpm_clear_only(nm_mpm* h, void* stream) {
  int prev, now, next;
  mpm_rotate(h, prev, now, next);
  NM_LAUNCH(k_clear, dim3(NM_CLEAR_WGS), dim3(256), 0, (hipStream_t)stream, h->gm, h->gv, h->gg, h->list[prev], h->count + prev,
                     h->list[now], h->count + now, h->count + next, h->flags, h->epoch);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

__global__ void k_fill_int(int* __restrict__ a, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}
// per-block arrays of the frame-level exchange: slot[b] (-1 between frames) and dil[b] (tag of the last negotiation whose
// neighbourhood holds b); *tag = a fresh tag for a new negotiation
int nm_mpm_xchg_arrays(nm_mpm* h, int** slot, int** dil, int* new_tag) {
  if (!h->sh_slot) {
    NM_HIP_CHECK(hipMalloc(&h->sh_slot, h->nblocks * sizeof(int)));
    NM_HIP_CHECK(hipMalloc(&h->sh_dil, h->nblocks * sizeof(int)));
    NM_LAUNCH(k_fill_int, dim3(nm_div_up(h->nblocks, 256)), dim3(256), 0, (hipStream_t)0, h->sh_slot, h->nblocks, -1);
    NM_LAUNCH(k_fill_int, dim3(nm_div_up(h->nblocks, 256)), dim3(256), 0, (hipStream_t)0, h->sh_dil, h->nblocks, 0);
    NM_LAUNCH_CHECK();
    NM_HIP_CHECK(hipDeviceSynchronize());
    h->dil_tag = 0;
  }
  if (slot) *slot = h->sh_slot;
  if (dil) *dil = h->sh_dil;
  if (new_tag) *new_tag = ++h->dil_tag;
  return NM_OK;
}
int nm_mpm_grid_dims(const nm_mpm* h) { return h->k.nb; }
int nm_mpm_dil_tag(const nm_mpm* h) { return h->dil_tag; }

// ---------------------------------------------------------------- particle-sharded substep: the same launches, cut at the
// two points where the caller sums the blocks that several ranks touch (nm_shard.hip has the exchange kernels)
nm_mpm_view nm_mpm_get_view(nm_mpm* h) {
  nm_mpm_view v;
  v.gm = h->gm; v.gg = h->gg; v.flags = h->flags; v.list = h->list[h->cur]; v.count = h->count + h->cur;
  v.epoch = h->epoch; v.nblocks = h->nblocks;
  return v;
}

__global__ void k_shared_init(int* __restrict__ cnt, int* __restrict__ pos, int nblocks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nblocks) { cnt[i] = 0; pos[i] = 0x7fffffff; }
}

int nm_mpm_shared_counters(nm_mpm* h, int** cnt, int** pos) {
  if (!h->sh_cnt) {
    NM_HIP_CHECK(hipMalloc(&h->sh_cnt, h->nblocks * sizeof(int)));
    NM_HIP_CHECK(hipMalloc(&h->sh_pos, h->nblocks * sizeof(int)));
    NM_LAUNCH(k_shared_init, dim3(nm_div_up(h->nblocks, 256)), dim3(256), 0, (hipStream_t)0, h->sh_cnt, h->sh_pos, h->nblocks);
    NM_LAUNCH_CHECK();
    NM_HIP_CHECK(hipDeviceSynchronize());
  }
  *cnt = h->sh_cnt;
  *pos = h->sh_pos;
  return NM_OK;
}

extern "C" int nm_mpm_p2g(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, void* stream) {
  NM_REQUIRE(h, "null handle");
  NM_REQUIRE(n >= 0, "negative particle count");
  if (n > 0) {
    int rc = check_particles(st, cur, true);
    if (rc) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  const int prev = h->cur, now = (prev + 1) % 3, next = (prev + 2) % 3;
  h->epoch += 1;
  NM_LAUNCH(k_clear, dim3(NM_CLEAR_WGS), dim3(256), 0, s, h->gm, h->gv, h->gg, h->list[prev], h->count + prev, h->list[now],
                     h->count + now, h->count + next, h->flags, h->epoch);
  NM_LAUNCH_CHECK();
  if (n > 0) {   // a rank without particles still takes part in the exchange with an empty list
    NM_LAUNCH(k_p2g, dim3(scatter_grid(h, n)), dim3(NM_SC_T), 0, s, scatter_k(h, n), n, st->vol, st->rho, st->enabled, cur->x,
                       cur->v, cur->C, cur->stress, h->gm, h->flags, h->list[now], h->count + now, h->epoch, (const int*)nullptr);
    NM_LAUNCH_CHECK();
  }
  h->cur = now;
  return NM_OK;
}

extern "C" int nm_mpm_forward_finish(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* next,
                                     void* gridrec, int32_t cap_blocks, int32_t* status, void* stream) {
  NM_REQUIRE(h, "null handle");
  NM_REQUIRE(!gridrec || cap_blocks > 0, "grid cache record without capacity");
  NM_REQUIRE(n >= 0, "negative particle count");
  hipStream_t s = (hipStream_t)stream;
  const int now = h->cur;
  GridRec none = {nullptr, nullptr, nullptr};
  NM_LAUNCH(k_grid_op, dim3(kSweepGrid), dim3(256), 0, s, h->k, h->gm, h->gv, h->list[now], h->count + now,
                     gridrec ? gridrec_at(gridrec, cap_blocks) : none, cap_blocks, (const int*)nullptr, (int*)status,
                     (const int*)nullptr, (const float4*)nullptr, DroppedBlocks{nullptr, nullptr, nullptr, 0});
  NM_LAUNCH_CHECK();
  if (n == 0) return NM_OK;
  int rc = check_particles(st, cur, true);
  if (rc) return rc;
  rc = check_particles(st, next, false);
  if (rc) return rc;
  NM_LAUNCH(k_g2p, dim3(nm_div_up(n, 256)), dim3(256), 0, s, h->k, n, st->clip_bound, st->enabled, cur->x, cur->v, cur->C,
                     cur->F, h->gv, next->x, next->v, next->C, next->F, h->fresh_rows);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" int nm_mpm_backward_begin(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, const nm_particles* next,
                                     const nm_particles* gnext, nm_particles* gcur, const void* gridrec, int32_t cap_blocks,
                                     void* stream) {
  NM_REQUIRE(h, "null handle");
  NM_REQUIRE(gridrec && cap_blocks > 0, "the sharded reverse sweep needs the substep's grid cache record");
  NM_REQUIRE(n >= 0, "negative particle count");
  hipStream_t s = (hipStream_t)stream;
  // restore {mv, m}, v and the block list of the forward substep (they already hold the sums over the ranks)
  int rc = mpm_build_grid(h, 0, st, cur, s, nullptr, gridrec, cap_blocks, true);
  if (rc) return rc;
  if (n == 0) return NM_OK;
  rc = check_particles(st, cur, true);
  if (rc) return rc;
  NM_REQUIRE(next && next->v && next->C, "next state (v, C) required");
  NM_REQUIRE(gnext && gnext->x && gnext->v && gnext->C && gnext->F, "null incoming gradients");
  NM_REQUIRE(gcur && gcur->x && gcur->v && gcur->C && gcur->F && gcur->stress, "null outgoing gradients");
  NM_LAUNCH(k_g2p_bwd, dim3(scatter_grid(h, n)), dim3(NM_SC_T), 0, s, scatter_k(h, n), n, st->clip_bound, st->enabled, cur->x, cur->F,
                     next->v, next->C, gnext->x, gnext->v, gnext->C, gnext->F, h->gv, h->gg, gcur->x, gcur->F);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" int nm_mpm_backward_finish(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* gcur,
                                      void* stream) {
  NM_REQUIRE(h, "null handle");
  NM_REQUIRE(n >= 0, "negative particle count");
  hipStream_t s = (hipStream_t)stream;
  const int now = h->cur;
  GridRec nostamp = {nullptr, nullptr, nullptr};
  NM_LAUNCH(k_grid_op_bwd, dim3(kSweepGrid), dim3(256), 0, s, (const int*)h->list[now], (const int*)(h->count + now), (const float4*)h->gm, h->gg,
                     h->flags, (const int*)nullptr, (const float4*)nullptr, h->k, nostamp, 0);
  NM_LAUNCH_CHECK();
  if (n == 0) return NM_OK;
  int rc = check_particles(st, cur, true);
  if (rc) return rc;
  NM_REQUIRE(gcur && gcur->x && gcur->v && gcur->C && gcur->F && gcur->stress, "null outgoing gradients");
  NM_LAUNCH(k_p2g_bwd, dim3(nm_div_up(n, 256)), dim3(256), 0, s, h->k, n, st->vol, st->rho, st->enabled, cur->x, cur->v, cur->C,
                     cur->stress, h->gg, gcur->x, gcur->v, gcur->C, gcur->stress);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" int nm_mpm_grid_stats(nm_mpm* h, int32_t* active_blocks, int32_t* nodes_with_mass, void* stream) {
  NM_REQUIRE(h, "null handle");
  hipStream_t s = (hipStream_t)stream;
  NM_HIP_CHECK(hipMemsetAsync(h->count + 4, 0, 2 * sizeof(int), s));
  NM_LAUNCH(k_grid_stats, dim3(kSweepGrid), dim3(256), 0, s, h->gm, h->list[h->cur], h->count + h->cur,
                     h->count + 4);
  NM_LAUNCH_CHECK();
  int host[2];
  NM_HIP_CHECK(hipMemcpyAsync(host, h->count + 4, sizeof(host), hipMemcpyDeviceToHost, s));
  NM_HIP_CHECK(hipStreamSynchronize(s));
  if (active_blocks) *active_blocks = host[0];
  if (nodes_with_mass) *nodes_with_mass = host[1];
  return NM_OK;
}

extern "C" int nm_mpm_grid_export(nm_mpm* h, float* mv, float* m, float* v, void* stream) {
  NM_REQUIRE(h, "null handle");
  int G = h->k.G;
  NM_LAUNCH(k_grid_export, dim3(nm_div_up((int64_t)G * G * G, 256)), dim3(256), 0, (hipStream_t)stream, h->k,
                     h->gm, h->gv, mv, m, v, h->flags, h->epoch);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
