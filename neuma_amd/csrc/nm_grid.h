// Block-sparse grid pieces shared by the MPM kernels (nm_mpm.hip) and the constitutive kernels (nm_material.hip), which in
// a roll-out perform the grid housekeeping of the following MPM substep in their prologue (GridPrologue below).
#pragma once
#include "nm_common.h"

struct MpmK {
  int G, Gp, nb;
  float dt, dx, inv_dx, eps;
  float gdt[3];
  int bound, bc;
  int dbg;
  int maxpass;  // NM_DBG experiment switches (0 in production)
  int ppw;      // scatter kernels: particles per workgroup (set per launch, scatter_k)
  int smode;    // scatter kernels: 1 = fp64 LDS atomics into the workgroup tile (default), 0 = counting sort + barrier-separated pushes (NEUMA_SCATTER=sort)
};

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


// mpm.py:432-498 for one particle: gather v and C' from the 27 stencil nodes, advance x (clamped), return the trial
// deformation gradient (I + dt C') F in Fo.  A disabled particle (mpm.py:443-444: the reference's g2p returns at once) leaves
// its row of the next state as the buffer held it - zeros / identity in the fresh model.state() that MPMDiffSim hands over
// (interface.py:101-105), the particle's own state when the step is in place (MPMForwardSim).  `fresh`: the next buffer is not
// such a state object (the roll-out's checkpoints live in a raw workspace), so those values are written here: x = v = 0, C = 0,
// Fo = I, exactly what a chain of MPMDiffSim calls produces.  UNROLL: all 27
// gathers in flight (for callers that run one wave per SIMD); otherwise nine per trip, which keeps k_g2p at 4+ waves/SIMD.
// FILL (passive extra sets, mpm.py:260-277): a stencil node in a block the scattering particles did not touch reads what the
// reference's dense grid_op sweep leaves there - BC(g dt) of an empty node (mpm.py:384-385 / 413-414) - instead of the
// block-sparse grid's zero; `flags[block] == epoch` marks the blocks this substep built.
// the loads of a particle's g2p that do not depend on its stencil (issued as a block; g2p_in_load of the NEXT round of a
// constitutive wave is issued before the current round's MLP, so that its HBM round trip hides behind it)
// (x and F stay in the shape their loads return them in - 12 and 16 + 16 + 4 bytes - until g2p_particle unpacks them: as
//  scalars under a lane predicate the compiler re-packed the loaded registers right behind the loads, which put an HBM round
//  trip - and, vmcnt counting stores too, the drain of every store of the round before - at the top of each round of the
//  constitutive kernels.  Callers that load ahead do so unconditionally, at a clamped particle index.)
typedef float nm_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float nm_f3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float nm_f4v __attribute__((ext_vector_type(4)));
typedef float nm_f3v __attribute__((ext_vector_type(3)));
struct G2pIn {
  int e;
  float clip;
  nm_f3v x;
  nm_f4v Fa, Fb;
  float Fc;
};
__device__ __forceinline__ G2pIn g2p_in_load(int p, const float* __restrict__ clip, const int* __restrict__ enabled, const float* x,
                                             const float* F) {
  G2pIn in;
  in.e = enabled[p];
  in.x = *reinterpret_cast<const nm_f3u*>(x + (size_t)3 * p);
  in.clip = clip[p];
  in.Fa = *reinterpret_cast<const nm_f4u*>(F + (size_t)9 * p);
  in.Fb = *reinterpret_cast<const nm_f4u*>(F + (size_t)9 * p + 4);
  in.Fc = F[(size_t)9 * p + 8];
  return in;
}
template <bool UNROLL, bool FILL = false>
__device__ __forceinline__ void g2p_particle(const MpmK& K, int p, const G2pIn& in, const float* x,
                                             const float4* __restrict__ gv, float* xn, float* vn, float* Cn, M3& Fo,
                                             const int* __restrict__ flags = nullptr, int epoch = 0, bool fresh = false);
template <bool UNROLL, bool FILL = false>
__device__ __forceinline__ void g2p_particle(const MpmK& K, int p, const float* __restrict__ clip, const int* __restrict__ enabled,
                                             const float* x, const float* v, const float* C, const float* F,
                                             const float4* __restrict__ gv, float* xn, float* vn, float* Cn, M3& Fo,
                                             const int* __restrict__ flags = nullptr, int epoch = 0, bool fresh = false) {
  // (every load that does not depend on the stencil is issued before `enabled` is looked at: a wave of the constitutive kernels
  //  is alone on its SIMD, and enabled -> x -> gathers -> F were four round trips in a row)
  const G2pIn in = g2p_in_load(p, clip, enabled, x, F);
  g2p_particle<UNROLL, FILL>(K, p, in, x, gv, xn, vn, Cn, Fo, flags, epoch, fresh);
}
template <bool UNROLL, bool FILL>
__device__ __forceinline__ void g2p_particle(const MpmK& K, int p, const G2pIn& in, const float* x,
                                             const float4* __restrict__ gv, float* xn, float* vn, float* Cn, M3& Fo,
                                             const int* __restrict__ flags, int epoch, bool fresh) {
  const int e_ = in.e;
  float xp[3] = {in.x[0], in.x[1], in.x[2]};
  const float clip_p = in.clip;
  M3 Fp;
  Fp.m[0] = in.Fa[0]; Fp.m[1] = in.Fa[1]; Fp.m[2] = in.Fa[2]; Fp.m[3] = in.Fa[3];
  Fp.m[4] = in.Fb[0]; Fp.m[5] = in.Fb[1]; Fp.m[6] = in.Fb[2]; Fp.m[7] = in.Fb[3]; Fp.m[8] = in.Fc;
  if (e_ == 0) {
    if (fresh && xn != x) {
      Fo = m3_ident();
#pragma unroll
      for (int a = 0; a < 3; ++a) { xn[3 * p + a] = 0.f; vn[3 * p + a] = 0.f; }
#pragma unroll
      for (int a = 0; a < 9; ++a) Cn[9 * p + a] = 0.f;
    } else {
      Fo = Fp;     // (not stored by the callers)
    }
    return;
  }
  Stencil st;
  make_stencil(K, xp, st);
  float nv[3] = {0.f, 0.f, 0.f};
  M3 nC = m3_zero();
  const float kap = 4.0f * K.inv_dx * K.inv_dx;
  auto slab = [&](int i) {
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
        const int ni = st.b[0] + i, nj = st.b[1] + j, nk = st.b[2] + k;
        float4 g = gv[node_addr(ni, nj, nk, K.nb)];
        if (FILL) {
          if (ni < K.G && nj < K.G && nk < K.G && flags[((ni >> 2) * K.nb + (nj >> 2)) * K.nb + (nk >> 2)] != epoch) {
            const float4 empty = {0.f, 0.f, 0.f, 0.f};
            float u[3], mk[3];
            grid_velocity(K, ni, nj, nk, empty, u, mk);
            g.x = u[0] * mk[0]; g.y = u[1] * mk[1]; g.z = u[2] * mk[2];
          }
        }
        nv[0] += w * g.x; nv[1] += w * g.y; nv[2] += w * g.z;
        float kw = kap * w;  // mpm.py:479: (4 w inv_dx^2) outer(v, dpos)
        nC.m[0] += kw * g.x * d0; nC.m[1] += kw * g.x * d1; nC.m[2] += kw * g.x * d2;
        nC.m[3] += kw * g.y * d0; nC.m[4] += kw * g.y * d1; nC.m[5] += kw * g.y * d2;
        nC.m[6] += kw * g.z * d0; nC.m[7] += kw * g.z * d1; nC.m[8] += kw * g.z * d2;
      }
    }
  };
  if (UNROLL) {
    slab(0); slab(1); slab(2);
  } else {
#pragma unroll 1
    for (int i = 0; i < 3; ++i) slab(i);
  }
  M3 T = nC;
#pragma unroll
  for (int i = 0; i < 9; ++i) T.m[i] *= K.dt;
  T.m[0] += 1.f; T.m[4] += 1.f; T.m[8] += 1.f;
  Fo = m3_mul(T, Fp);  // mpm.py:489
  float bnd = clip_p * K.dx;
  float lo = 0.0f + bnd, hi = 1.0f - bnd;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float t = xp[a] + K.dt * nv[a];
    xn[3 * p + a] = fminf(fmaxf(t, lo), hi);  // wp.clamp, mpm.py:491-497
    vn[3 * p + a] = nv[a];
  }
  m3_store(Cn + 9 * p, nC);
}

// g2p of the substep fused into the plasticity kernel that consumes its trial F (roll-out forward): gv == NULL -> off
struct G2pFuse {
  const float4* gv;      // node velocities
  MpmK K;
  const float *clip; const int* enabled;
  const float *x, *v, *C, *F;
  float *xn, *vn, *Cn;
};

// A grid cache record (optional, one per substep of a roll-out): the active-block list and the scattered node values
// {mv, m} of those blocks, so that the reverse sweep restores the grid instead of re-running p2g.
//   int hdr[4]  (hdr[0] = number of blocks, -1 = record invalid because the substep touched more than `cap` blocks)
//   int list[cap]   float4 gm[cap * 64]
struct GridRec {
  int* hdr;
  int* list;
  float4* gm;
};

// Zero the blocks the previous substep touched (all three node arrays) and CARRY the ones that still held mass over
// into the new active list, stamped with the new epoch: particles move a fraction of a cell per substep, so p2g finds
// nearly every block it touches already listed (one flag read) and the returning atomics of mark_block - which would
// otherwise all hit the same counter in a burst - are left to the few blocks that are genuinely new.  Blocks that
// lost their mass drop out here.  One returning atomic per wave reserves the list slots.  count_next is reset for the
// substep after this one (nobody reads it now).  Called by workgroups wg = 0..nwg-1 of 256 threads.
#define NM_CLEAR_WGS 64
__device__ __forceinline__ void grid_clear_carry(float4* __restrict__ gm, float4* __restrict__ gv, float4* __restrict__ gg,
                                                 const int* __restrict__ list_prev, const int* __restrict__ count_prev,
                                                 int* __restrict__ list_now, int* __restrict__ count_now,
                                                 int* __restrict__ count_next, int* __restrict__ flags, int epoch, int wg, int nwg,
                                                 bool keep_gv = false) {
  // keep_gv (forward pair kernel of the roll-out): the launch that carries this clear is still gathering the previous
  // substep's velocities - gv is left alone; the coming grid update overwrites the blocks that stay and zeroes those that
  // drop out (k_grid_op, `dropped`)
  const int cnt = *count_prev;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nw = nwg * 4, w = wg * 4 + wave;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (wg == 0 && threadIdx.x == 0) *count_next = 0;
  for (int li0 = w; li0 < cnt; li0 += nw * 64) {      // rounds of up to 64 blocks per wave (one keep-bit each)
    unsigned long long keep = 0ull;
    int it = 0;
    for (int li = li0; li < cnt && it < 64; li += nw, ++it) {
      const int node = (list_prev[li] << 6) + lane;
      const bool has = gm[node].w > 0.f;
      gm[node] = z;
      if (!keep_gv) gv[node] = z;
      gg[node] = z;
      if (__ballot(has) != 0ull) keep |= 1ull << it;
    }
    const int nkeep = __popcll(keep);
    if (nkeep == 0) continue;
    int pos = 0;
    if (lane == 0) pos = atomicAdd(count_now, nkeep);
    pos = __shfl(pos, 0, 64);
    it = 0;
    for (int li = li0; li < cnt && it < 64; li += nw, ++it) {
      if ((keep >> it) & 1ull) {
        if (lane == 0) {
          const int b = list_prev[li];
          list_now[pos] = b;
          flags[b] = epoch;
        }
        ++pos;
      }
    }
  }
}

// reverse sweep: rebuild {mv, m}, the post-grid-op velocities and the active list from a cache record
__device__ __forceinline__ void grid_restore(const MpmK& K, const GridRec& rec, float4* __restrict__ gm, float4* __restrict__ gv,
                                             int* __restrict__ list, int* __restrict__ count, int wg, int nwg) {
  const int cnt = rec.hdr[0];
  if (cnt < 0) return;   // invalid record: the p2g / grid_op launches that follow do the work
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wg == 0 && threadIdx.x == 0) *count = cnt;
  for (int li = wg * 4 + wave; li < cnt; li += nwg * 4) {
    int b = rec.list[li];
    int i, j, k;
    block_coords(b, K.nb, lane, i, j, k);
    int node = (b << 6) + lane;
    float4 a = rec.gm[(li << 6) + lane];
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < K.G && j < K.G && k < K.G) {
      float u[3], mk[3];
      grid_velocity(K, i, j, k, a, u, mk);
      out.x = u[0] * mk[0]; out.y = u[1] * mk[1]; out.z = u[2] * mk[2];
    }
    gm[node] = a;
    gv[node] = out;
    if (lane == 0) list[li] = b;
  }
}

// Grid housekeeping a constitutive kernel of the fused roll-out performs before its own work, so that the substep does
// not pay separate launches for it (each ~8 us of launch + dependent-load latency for ~1 MB of traffic):
//   mode 1 (forward, in front of p2g): grid_clear_carry.
//   mode 2 (verified reverse sweep, in front of the g2p adjoint): restore the substep's record AND zero what the
//           previous substep of the sweep left behind, in one pass without ordering between workgroups: a block of the
//           previous list that is also in the record (flags[b] == epoch - stamped by the previous substep's
//           k_grid_op_bwd, which knows the next record) only has its adjoint scratch zeroed, because its {mv, m} and v
//           are being overwritten by whichever workgroup restores it; a block that left the list is zeroed entirely.
struct GridPrologue {
  int mode;   // 0 = none
  int keep_gv;    // mode 1 only: leave the velocity array alone (see grid_clear_carry)
  int mat_grid;   // set by the constitutive launcher: workgroups that carry particles (the rest only run the prologue)
  MpmK K;
  float4 *gm, *gv, *gg;
  const int *list_prev, *count_prev;
  int *list_now, *count_now, *count_next;
  int* flags;
  int epoch;
  GridRec rec;
};

__device__ __forceinline__ void grid_prologue_clear(const GridPrologue& g, int wg, int nwg_all) {      // mode 1 alone
  const int nwg = min(nwg_all, NM_CLEAR_WGS);
  if (wg < nwg) grid_clear_carry(g.gm, g.gv, g.gg, g.list_prev, g.count_prev, g.list_now, g.count_now, g.count_next, g.flags,
                                 g.epoch, wg, nwg, g.keep_gv != 0);
}
__device__ __forceinline__ void grid_prologue(const GridPrologue& g, int wg, int nwg_all) {
  if (g.mode == 0) return;
  const int nwg = min(nwg_all, NM_CLEAR_WGS);
  if (g.mode == 1) {
    if (wg < nwg) grid_clear_carry(g.gm, g.gv, g.gg, g.list_prev, g.count_prev, g.list_now, g.count_now, g.count_next, g.flags,
                                   g.epoch, wg, nwg, g.keep_gv != 0);
    return;
  }
  // mode 2: the first nwg workgroups sweep the previous list, the next nwg restore the record (a launch with fewer
  // than 2*nwg workgroups lets every workgroup do a share of both)
  const bool split = nwg_all >= 2 * nwg;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (!split || wg < nwg) {
    const int sn = split ? nwg : nwg_all;
    const int cnt = *g.count_prev;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wg == 0 && threadIdx.x == 0) *g.count_next = 0;
    for (int li = wg * 4 + wave; li < cnt; li += sn * 4) {
      const int b = g.list_prev[li];
      const int node = (b << 6) + lane;
      g.gg[node] = z;
      if (g.flags[b] != g.epoch) { g.gm[node] = z; g.gv[node] = z; }
    }
  }
  if (!split) grid_restore(g.K, g.rec, g.gm, g.gv, g.list_now, g.count_now, wg, nwg_all);
  else if (wg >= nwg && wg < 2 * nwg) grid_restore(g.K, g.rec, g.gm, g.gv, g.list_now, g.count_now, wg - nwg, nwg);
}

// ---- internal cross-file entry points of the fused roll-out (nm_rollout.hip)
int nm_mpm_prologue_forward(nm_mpm* h, GridPrologue* g, bool keep_gv = false);
int nm_mpm_prologue_backward(nm_mpm* h, const void* gridrec, int cap, GridPrologue* g);
int nm_mpm_forward_prepared(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* next, void* gridrec,
                            int32_t cap_blocks, void* stream);
int nm_mpm_backward_cached(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, const nm_particles* next,
                           const nm_particles* gnext, nm_particles* gcur, const void* gridrec, int32_t cap_blocks, bool verified,
                           bool prepared, const void* stamp_rec, void* stream);
// sharded roll-out: the same launches with the exchange of the shared blocks in between (nm_shard.hip / nm_rollout.hip)
int nm_mpm_backward_cached_begin(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, const nm_particles* next,
                                 const nm_particles* gnext, nm_particles* gcur, const void* gridrec, int32_t cap_blocks,
                                 bool verified, bool prepared, void* stream);
int nm_mpm_backward_cached_finish(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* gcur,
                                  const void* stamp_rec, int32_t cap_blocks, const float* xbuf, void* stream);
int nm_mpm_forward_prepared_p2g(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, void* stream);
int nm_mpm_forward_gridop_x(nm_mpm* h, void* gridrec, int32_t cap_blocks, int32_t* status, const float* xbuf, void* stream);
int nm_mpm_clear_only(nm_mpm* h, void* stream);
// pro: grid housekeeping performed in the kernel's prologue (NULL = none)
int nm_material_bwd_launch(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w, const float* wperm,
                           const float* gout, float* gF, float* wpart, int wmode, const float* trial_C, const int* enabled,
                           float dt, int flags /* bit 0: gF += ; bit 1: polar SVD adjoint */, const GridPrologue* pro, void* stream,
                           const float* svd_in = nullptr /* U | sigma | V of the input as the forward kernel stored them */,
                           const float* act = nullptr /* the forward kernel's activation cache */);
int nm_material_bwd_pair_launch(int32_t n, const float* F_e, const nm_mlp* we, const float* wperm_e, const float* gS, float* gF,
                                float* wpart_e, int wmode_e, float alpha_p, const float* F_p, const nm_mlp* wp,
                                const float* wperm_p, float* gFtrial, float* wpart_p, int wmode_p, const float* trial_C,
                                const int* enabled, float dt, int polar, const GridPrologue* pro, void* stream,
                                const float* svd_in_e = nullptr, const float* svd_in_p = nullptr, const float* act_e = nullptr,
                                const float* act_p = nullptr);
int nm_material_fwd_launch(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w, const float* wperm, float* out,
                           const GridPrologue* pro, const G2pFuse* g2p, void* stream,
                           float* svd_out = nullptr /* roll-out: keep U | sigma | V for the reverse sweep (21 n floats) */,
                           float* act_out = nullptr /* roll-out: keep the hidden activations (nm_material_act_floats(n)) */);
int nm_material_fwd_pair_launch(int32_t n, float alpha_p, const float* wperm_p, const float* wperm_e, float* F_next,
                                float* stress_next, const GridPrologue* pro, const G2pFuse* g2p, void* stream, float* svd_p,
                                float* svd_e, float* act_p, float* act_e);
size_t nm_material_act_floats(int32_t n);
// g2p fused into the next constitutive kernel (roll-out forward): fills the descriptor / runs the substep without its g2p
int nm_mpm_g2p_fuse(nm_mpm* h, const nm_statics* st, const nm_particles* cur, nm_particles* next, G2pFuse* f);
int nm_mpm_forward_prepared_nog2p(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, void* gridrec,
                                  int32_t cap_blocks, void* stream);
