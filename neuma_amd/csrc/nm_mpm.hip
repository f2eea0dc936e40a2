// MLS-MPM substep (APIC, quadratic B-spline) for gfx950: p2g / grid_op / g2p and their hand-written
// adjoints.  Behaviour follows /root/reference/modules/nclaw/sim/mpm.py:279-498 (see SURVEY.md App. A
// for the adjoint derivation); the structure does not:
//
//   * the grid lives in HBM as 4x4x4-node blocks of float4 {mv.xyz, m} / {v.xyz, -} (1 KiB per block),
//     and only blocks touched by a particle stencil are ever cleared or updated (active-block list built
//     by p2g with an epoch flag per block) — the reference sweeps all G^3 cells 12 times per step;
//   * scatters (p2g, g2p-adjoint) accumulate in an LDS tile covering the bounding box of the
//     workgroup's 256 particles with ds_add_f32 and flush each touched node once with global atomics;
//     a workgroup whose particles are too spread out falls back to direct global atomics, so the
//     particle order only affects speed, never results beyond fp32 summation order;
//   * stencil nodes with an index >= G (reference: out-of-bounds access when x > 1-1.5dx) land in
//     padding blocks whose velocity is defined as zero.
#include "nm_common.h"

#define NM_TILE_CAP 2048  // LDS tile nodes (x16 B = 32 KiB)

struct MpmK {
  int G, Gp, nb;
  float dt, dx, inv_dx, eps;
  float gdt[3];
  int bound, bc;
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

// ---------------------------------------------------------------- workgroup LDS tile for scatters
struct TileGeom {
  int o[3];
  int n[3];
  int vol;
  bool any, use;
};

__device__ __forceinline__ TileGeom tile_setup(bool active, const int* b, int* s_mm, float* s_tile) {
  const int tid = threadIdx.x;
  if (tid < 3) { s_mm[tid] = 0x7fffffff; s_mm[3 + tid] = -0x7fffffff; }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { atomicMin(&s_mm[a], b[a]); atomicMax(&s_mm[3 + a], b[a]); }
  }
  __syncthreads();
  TileGeom g;
  g.any = s_mm[0] != 0x7fffffff;
#pragma unroll
  for (int a = 0; a < 3; ++a) { g.o[a] = s_mm[a]; g.n[a] = g.any ? s_mm[3 + a] - s_mm[a] + 3 : 0; }
  g.vol = g.n[0] * g.n[1] * g.n[2];
  g.use = g.any && g.vol <= NM_TILE_CAP;
  if (g.use) {
    for (int i = tid; i < g.vol * 4; i += blockDim.x) s_tile[i] = 0.f;
  }
  __syncthreads();
  return g;
}

// flush the LDS tile: one global atomic set per touched node; mark the blocks it overlaps
template <int NCH>
__device__ __forceinline__ void tile_flush(const TileGeom& g, const float* s_tile, float4* __restrict__ grid,
                                           int nb, int* flags, int* list, int* count, int epoch) {
  __syncthreads();
  const int tid = threadIdx.x;
  if (g.use) {
    const int nyz = g.n[1] * g.n[2];
    for (int idx = tid; idx < g.vol; idx += blockDim.x) {
      float t0 = s_tile[idx * 4], t1 = s_tile[idx * 4 + 1], t2 = s_tile[idx * 4 + 2], t3 = s_tile[idx * 4 + 3];
      if (t0 != 0.f || t1 != 0.f || t2 != 0.f || t3 != 0.f) {
        int i = idx / nyz, r = idx - i * nyz;
        int j = r / g.n[2], k = r - j * g.n[2];
        float* dst = (float*)&grid[node_addr(g.o[0] + i, g.o[1] + j, g.o[2] + k, nb)];
        unsafeAtomicAdd(dst, t0);
        unsafeAtomicAdd(dst + 1, t1);
        unsafeAtomicAdd(dst + 2, t2);
        if (NCH == 4) unsafeAtomicAdd(dst + 3, t3);
      }
    }
    if (flags) {
      int b0 = g.o[0] >> 2, b1 = g.o[1] >> 2, b2 = g.o[2] >> 2;
      int m0 = ((g.o[0] + g.n[0] - 1) >> 2) - b0 + 1, m1 = ((g.o[1] + g.n[1] - 1) >> 2) - b1 + 1,
          m2 = ((g.o[2] + g.n[2] - 1) >> 2) - b2 + 1;
      for (int t = tid; t < m0 * m1 * m2; t += blockDim.x) {
        int i = t / (m1 * m2), r = t - i * (m1 * m2);
        int j = r / m2, k = r - j * m2;
        mark_block(((b0 + i) * nb + (b1 + j)) * nb + (b2 + k), flags, list, count, epoch);
      }
    }
  }
}

template <int NCH>
__device__ __forceinline__ void scatter_node(const TileGeom& g, float* s_tile, float4* __restrict__ grid, int nb,
                                             int i, int j, int k, float a0, float a1, float a2, float a3) {
  if (g.use) {
    int idx = (((i - g.o[0]) * g.n[1] + (j - g.o[1])) * g.n[2] + (k - g.o[2])) * 4;
    unsafeAtomicAdd(&s_tile[idx], a0);
    unsafeAtomicAdd(&s_tile[idx + 1], a1);
    unsafeAtomicAdd(&s_tile[idx + 2], a2);
    if (NCH == 4) unsafeAtomicAdd(&s_tile[idx + 3], a3);
  } else {
    float* dst = (float*)&grid[node_addr(i, j, k, nb)];
    unsafeAtomicAdd(dst, a0);
    unsafeAtomicAdd(dst + 1, a1);
    unsafeAtomicAdd(dst + 2, a2);
    if (NCH == 4) unsafeAtomicAdd(dst + 3, a3);
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

// mpm.py:321-371
__global__ void __launch_bounds__(256) k_p2g(MpmK K, int n, const float* __restrict__ vol, const float* __restrict__ rho,
                                             const int* __restrict__ enabled, const float* __restrict__ x,
                                             const float* __restrict__ v, const float* __restrict__ C,
                                             const float* __restrict__ S, float4* __restrict__ gm, int* flags,
                                             int* list, int* count, int epoch) {
  __shared__ int s_mm[6];
  __shared__ float s_tile[NM_TILE_CAP * 4];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = p < n && enabled[p] != 0;
  Stencil st;
  float pm = 0.f, mom[3] = {0.f, 0.f, 0.f};
  M3 A = m3_zero();
  if (active) {
    make_stencil(K, x + 3 * p, st);
    float vl = vol[p];
    pm = vl * rho[p];
    float ks = -K.dt * vl * 4.0f * K.inv_dx * K.inv_dx;  // mpm.py:357
    M3 Sp = m3_load(S + 9 * p), Cp = m3_load(C + 9 * p);
#pragma unroll
    for (int i = 0; i < 9; ++i) A.m[i] = ks * Sp.m[i] + pm * Cp.m[i];
#pragma unroll
    for (int a = 0; a < 3; ++a) mom[a] = pm * v[3 * p + a];
  } else {
    st.b[0] = st.b[1] = st.b[2] = 0;
  }
  TileGeom g = tile_setup(active, st.b, s_mm, s_tile);
  if (!g.any) return;
  if (active) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float d0 = ((float)i - st.f[0]) * K.dx;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float d1 = ((float)j - st.f[1]) * K.dx;
        float wij = st.w[0][i] * st.w[1][j];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float d2 = ((float)k - st.f[2]) * K.dx;
          float w = wij * st.w[2][k];
          float m0 = w * (mom[0] + A.m[0] * d0 + A.m[1] * d1 + A.m[2] * d2);
          float m1 = w * (mom[1] + A.m[3] * d0 + A.m[4] * d1 + A.m[5] * d2);
          float m2 = w * (mom[2] + A.m[6] * d0 + A.m[7] * d1 + A.m[8] * d2);
          scatter_node<4>(g, s_tile, gm, K.nb, st.b[0] + i, st.b[1] + j, st.b[2] + k, m0, m1, m2, w * pm);
        }
      }
    }
    if (!g.use) {
      for (int i = st.b[0] >> 2; i <= (st.b[0] + 2) >> 2; ++i)
        for (int j = st.b[1] >> 2; j <= (st.b[1] + 2) >> 2; ++j)
          for (int k = st.b[2] >> 2; k <= (st.b[2] + 2) >> 2; ++k)
            mark_block((i * K.nb + j) * K.nb + k, flags, list, count, epoch);
    }
  }
  tile_flush<4>(g, s_tile, gm, K.nb, flags, list, count, epoch);
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

// adjoint of g2p: writes gx (direct part), gF; scatters vbar into gg
__global__ void __launch_bounds__(256, 2) k_g2p_bwd(MpmK K, int n, const float* __restrict__ clip, const int* __restrict__ enabled,
                                                 const float* __restrict__ x, const float* __restrict__ F,
                                                 const float* __restrict__ vnext, const float* __restrict__ Cnext,
                                                 const float* __restrict__ gxn, const float* __restrict__ gvn,
                                                 const float* __restrict__ gCn, const float* __restrict__ gFn,
                                                 const float4* __restrict__ gv, float4* __restrict__ gg,
                                                 float* __restrict__ gx, float* __restrict__ gF) {
  __shared__ int s_mm[6];
  __shared__ float s_tile[NM_TILE_CAP * 4];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = p < n && enabled[p] != 0;
  Stencil st;
  float vt[3] = {0.f, 0.f, 0.f}, xbar[3] = {0.f, 0.f, 0.f};
  M3 Ct = m3_zero(), Fbar = m3_zero();
  if (active) {
    float xp[3] = {x[3 * p], x[3 * p + 1], x[3 * p + 2]};
    make_stencil(K, xp, st);
    float bnd = clip[p] * K.dx;
    float lo = 0.0f + bnd, hi = 1.0f - bnd;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float t = xp[a] + K.dt * vnext[3 * p + a];
      float xe = (t >= lo && t <= hi) ? gxn[3 * p + a] : 0.f;  // clamp passes the gradient only inside
      xbar[a] = xe;
      vt[a] = gvn[3 * p + a] + K.dt * xe;
    }
    M3 Fp = m3_load(F + 9 * p), gFp = m3_load(gFn + 9 * p), Cn = m3_load(Cnext + 9 * p), gCp = m3_load(gCn + 9 * p);
    M3 T = Cn;
#pragma unroll
    for (int i = 0; i < 9; ++i) T.m[i] *= K.dt;
    T.m[0] += 1.f; T.m[4] += 1.f; T.m[8] += 1.f;
    Fbar = m3_mul_tn(T, gFp);           // (I + dt C')^T Fbar'
    M3 FF = m3_mul_nt(gFp, Fp);         // Fbar' F^T
#pragma unroll
    for (int i = 0; i < 9; ++i) Ct.m[i] = gCp.m[i] + K.dt * FF.m[i];
  } else {
    st.b[0] = st.b[1] = st.b[2] = 0;
  }
  TileGeom g = tile_setup(active, st.b, s_mm, s_tile);
  if (!g.any) {
    if (p < n) {
#pragma unroll
      for (int a = 0; a < 3; ++a) gx[3 * p + a] = 0.f;
      m3_store(gF + 9 * p, m3_zero());
    }
    return;
  }
  if (active) {
    const float kap = 4.0f * K.inv_dx * K.inv_dx;
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
          float4 gn = gv[node_addr(st.b[0] + i, st.b[1] + j, st.b[2] + k, K.nb)];
          // Ct dpos
          float c0 = Ct.m[0] * d0 + Ct.m[1] * d1 + Ct.m[2] * d2;
          float c1 = Ct.m[3] * d0 + Ct.m[4] * d1 + Ct.m[5] * d2;
          float c2 = Ct.m[6] * d0 + Ct.m[7] * d1 + Ct.m[8] * d2;
          float kw = kap * w;
          scatter_node<3>(g, s_tile, gg, K.nb, st.b[0] + i, st.b[1] + j, st.b[2] + k,
                          w * vt[0] + kw * c0, w * vt[1] + kw * c1, w * vt[2] + kw * c2, 0.f);
          float dLdw = vt[0] * gn.x + vt[1] * gn.y + vt[2] * gn.z + kap * (gn.x * c0 + gn.y * c1 + gn.z * c2);
          float gw0 = dw0i * st.w[1][j] * st.w[2][k] * K.inv_dx;
          float gw1 = w0i * st.dw[1][j] * st.w[2][k] * K.inv_dx;
          float gw2 = w0i * st.w[1][j] * st.dw[2][k] * K.inv_dx;
          // Ct^T v_i
          float t0 = Ct.m[0] * gn.x + Ct.m[3] * gn.y + Ct.m[6] * gn.z;
          float t1 = Ct.m[1] * gn.x + Ct.m[4] * gn.y + Ct.m[7] * gn.z;
          float t2 = Ct.m[2] * gn.x + Ct.m[5] * gn.y + Ct.m[8] * gn.z;
          xbar[0] += dLdw * gw0 - kw * t0;
          xbar[1] += dLdw * gw1 - kw * t1;
          xbar[2] += dLdw * gw2 - kw * t2;
        }
      }
    }
  }
  if (p < n) {
#pragma unroll
    for (int a = 0; a < 3; ++a) gx[3 * p + a] = xbar[a];
    m3_store(gF + 9 * p, Fbar);
  }
  tile_flush<3>(g, s_tile, gg, K.nb, nullptr, nullptr, nullptr, 0);
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
    NM_LAUNCH(k_p2g, dim3(nm_div_up(n, 256)), dim3(256), 0, s, h->k, n, st->vol, st->rho, st->enabled, cur->x,
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
  int rc = check_particles(st, cur, true);
  if (rc) return rc;
  rc = check_particles(st_extra, extra, false);
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
  NM_LAUNCH(k_g2p_bwd, dim3(nwg), dim3(256), 0, s, h->k, n, st->clip_bound, st->enabled, cur->x, cur->F, next->v,
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

extern "C" int nm_mpm_grid_export(nm_mpm* h, float* mv, float* m, float* v, void* stream) {
  NM_REQUIRE(h, "null handle");
  int G = h->k.G;
  NM_LAUNCH(k_grid_export, dim3(nm_div_up((int64_t)G * G * G, 256)), dim3(256), 0, (hipStream_t)stream, h->k,
                     h->gm, h->gv, mv, m, v);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
