// oracle/c/neuma_ref.cpp - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// C++17 / OpenMP restatement of the reference's hot path, with the reference's own launch structure (dense G^3 grid,
// three MPM kernels + recompute in the backward pass, per-particle constitutive nets, per-pixel tile rasterizer):
//   MPM substep            /root/reference/modules/nclaw/sim/mpm.py:279-319 (forward / backward), 321-371 p2g,
//                          373-429 grid_op_{freeslip,noslip}, 432-498 g2p; adjoints as derived in SURVEY.md App. A
//   SVD convention         modules/nclaw/warp/svd.py:61-96 (U, V in SO(3), sign on sigma_2), adjoint: SURVEY.md App. B
//   constitutive nets      modules/nclaw/material/meta.py:196-221, 468-489 (+ MLPBlock 20-42, exact-erf GELU)
//   bindings / covariance  modules/tune/utils.py:424-472, modules/d3gs/utils/simulation_utils.py:25-48
//   rasterizer             diff-gaussian-rasterization @ gaussian-splatting b17ded92 (un-vendored): algorithm and constants
//                          of SURVEY.md App. D (same restatement as oracle/raster.py)
//   pixel loss             modules/d3gs/utils/loss_utils.py:17-24
// Used (a) as the `cpu_baseline` of bench.py - timed on all host cores, labelled "CPU restatement of the reference
// algorithm" (the reference's Warp-CPU path cannot run here: warp-lang is absent) - and (b) as a second, independent
// checker in tests/test_oracle_cref.py, itself validated against the fixtures that tests/golden/gen_mpm_golden.py and
// gen_material_golden.py produced by executing the reference's code.  Only tests/, __graft_entry__ and bench.py's
// cpu_baseline leg may load it; neuma_amd/ never does.
//
// Arithmetic is fp32 like the reference (the 3x3 SVD iterates in fp64 for robustness and rounds its factors to fp32).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" {

struct ref_sim_cfg {
  int32_t G;
  float dt;
  int32_t bound;
  float gravity[3];
  float eps;
  int32_t bc;  // 0 noslip, 1 freeslip
};

int ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void ref_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
}

// ------------------------------------------------------------------------------------------------ MPM
namespace {

struct Stencil {
  int b[3];
  float f[3], w[3][3], dw[3][3];
};

inline void make_stencil(const ref_sim_cfg& c, const float* xp, Stencil& s) {
  const float inv_dx = (float)c.G;
  for (int a = 0; a < 3; ++a) {
    float px = xp[a] * inv_dx;
    int b = (int)(px - 0.5f);  // C cast, mpm.py:337-339
    float f = px - (float)b;
    s.b[a] = b;
    s.f[a] = f;
    float wa = 1.5f - f, wb = f - 1.0f, wc = f - 0.5f;  // mpm.py:346-355
    s.w[a][0] = wa * wa * 0.5f;
    s.w[a][1] = 0.75f - wb * wb;
    s.w[a][2] = wc * wc * 0.5f;
    s.dw[a][0] = -wa;
    s.dw[a][1] = -2.f * wb;
    s.dw[a][2] = wc;
  }
}

inline bool node_ok(const ref_sim_cfg& c, int i, int j, int k) {
  return i >= 0 && j >= 0 && k >= 0 && i < c.G && j < c.G && k < c.G;  // outside: reference UB, dropped (oracle/mpm.py)
}

inline void grid_velocity(const ref_sim_cfg& c, int i, int j, int k, const float* mv, float m, float u[3], float mask[3]) {
  if (m > 0.f) {
    float inv = 1.f / (m + c.eps);
    for (int a = 0; a < 3; ++a) u[a] = mv[a] * inv + c.gravity[a] * c.dt;
  } else {
    for (int a = 0; a < 3; ++a) u[a] = c.gravity[a] * c.dt;
  }
  const int idx[3] = {i, j, k};
  bool hit[3];
  for (int a = 0; a < 3; ++a) hit[a] = (idx[a] < c.bound && u[a] < 0.f) || (idx[a] >= c.G - c.bound && u[a] > 0.f);
  if (c.bc == 0) {
    float z = (hit[0] || hit[1] || hit[2]) ? 0.f : 1.f;  // sequential tests on the updated v: once zero, nothing fires
    mask[0] = mask[1] = mask[2] = z;
  } else {
    for (int a = 0; a < 3; ++a) mask[a] = hit[a] ? 0.f : 1.f;
  }
}

// Scatter without one lock-prefixed add per (particle, node): every OpenMP thread owns a contiguous range of the particle list
// (spatially coherent: the scenes keep their particles in cell order), sums its contributions in a private dense box around
// that range's stencils and flushes each touched box node to the shared grid ONCE (atomic: neighbouring threads' boxes
// overlap by a cell or two).  A range whose box would be large - an unordered particle list - adds to the grid directly.
struct ThreadBox {
  int lo[3], n[3];
  bool local;
  std::vector<float> acc;
  template <int NCH>
  void open(const ref_sim_cfg& c, const int32_t* en, const float* x, int p0, int p1) {
    int hi[3] = {-1, -1, -1};
    lo[0] = lo[1] = lo[2] = 1 << 30;
    for (int p = p0; p < p1; ++p) {
      if (en[p] == 0) continue;
      Stencil s;
      make_stencil(c, x + 3 * p, s);
      for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], s.b[a]); hi[a] = std::max(hi[a], s.b[a] + 2); }
    }
    size_t vol = 1;
    for (int a = 0; a < 3; ++a) { n[a] = hi[a] >= lo[a] ? hi[a] - lo[a] + 1 : 0; vol *= (size_t)n[a]; }
    local = vol > 0 && vol <= ((size_t)1 << 18);
    if (local) acc.assign(vol * NCH, 0.f);
  }
  inline float* at(int i, int j, int k, int nch) { return &acc[(((size_t)(i - lo[0]) * n[1] + (j - lo[1])) * n[2] + (k - lo[2])) * nch]; }
};

void p2g(const ref_sim_cfg& c, int N, const float* vol, const float* rho, const int32_t* en, const float* x, const float* v,
         const float* C, const float* S, float* gmv, float* gm) {
  const float inv_dx = (float)c.G, dx = 1.0f / (float)c.G;
  const size_t G = c.G;
#pragma omp parallel
  {
    const int T = omp_get_num_threads(), t = omp_get_thread_num();
    const int p0 = (int)((int64_t)N * t / T), p1 = (int)((int64_t)N * (t + 1) / T);
    static thread_local ThreadBox box;
    box.open<4>(c, en, x, p0, p1);
    for (int p = p0; p < p1; ++p) {
      if (en[p] == 0) continue;
      Stencil s;
      make_stencil(c, x + 3 * p, s);
      const float pm = vol[p] * rho[p];
      const float ks = -c.dt * vol[p] * 4.0f * inv_dx * inv_dx;
      float A[9];
      for (int i = 0; i < 9; ++i) A[i] = ks * S[9 * p + i] + pm * C[9 * p + i];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          for (int k = 0; k < 3; ++k) {
            const int ni = s.b[0] + i, nj = s.b[1] + j, nk = s.b[2] + k;
            if (!node_ok(c, ni, nj, nk)) continue;
            const float d[3] = {((float)i - s.f[0]) * dx, ((float)j - s.f[1]) * dx, ((float)k - s.f[2]) * dx};
            const float w = s.w[0][i] * s.w[1][j] * s.w[2][k];
            float val[4];
            for (int a = 0; a < 3; ++a) val[a] = w * (pm * v[3 * p + a] + A[3 * a] * d[0] + A[3 * a + 1] * d[1] + A[3 * a + 2] * d[2]);
            val[3] = w * pm;
            if (box.local) {
              float* q = box.at(ni, nj, nk, 4);
              for (int a = 0; a < 4; ++a) q[a] += val[a];
            } else {
              const size_t n = (ni * G + nj) * G + nk;
              for (int a = 0; a < 3; ++a) {
#pragma omp atomic
                gmv[3 * n + a] += val[a];
              }
#pragma omp atomic
              gm[n] += val[3];
            }
          }
    }
    if (box.local)
      for (int i = 0; i < box.n[0]; ++i)
        for (int j = 0; j < box.n[1]; ++j)
          for (int k = 0; k < box.n[2]; ++k) {
            const float* q = box.at(box.lo[0] + i, box.lo[1] + j, box.lo[2] + k, 4);
            if (q[3] == 0.f && q[0] == 0.f && q[1] == 0.f && q[2] == 0.f) continue;
            const size_t n = ((size_t)(box.lo[0] + i) * G + (box.lo[1] + j)) * G + (box.lo[2] + k);
            for (int a = 0; a < 3; ++a) {
#pragma omp atomic
              gmv[3 * n + a] += q[a];
            }
#pragma omp atomic
            gm[n] += q[3];
          }
  }
}

void grid_op(const ref_sim_cfg& c, const float* gmv, const float* gm, float* gv) {
  const int G = c.G;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < G; ++i)
    for (int j = 0; j < G; ++j)
      for (int k = 0; k < G; ++k) {
        size_t n = ((size_t)i * G + j) * G + k;
        float u[3], mk[3];
        grid_velocity(c, i, j, k, gmv + 3 * n, gm[n], u, mk);
        for (int a = 0; a < 3; ++a) gv[3 * n + a] = u[a] * mk[a];
      }
}

void g2p(const ref_sim_cfg& c, int N, const float* clip, const int32_t* en, const float* x, const float* F, const float* gv,
         float* xn, float* vn, float* Cn, float* Fn) {
  const float inv_dx = (float)c.G, dx = 1.0f / (float)c.G, kap = 4.0f * inv_dx * inv_dx;
  const size_t G = c.G;
#pragma omp parallel for schedule(static)
  for (int p = 0; p < N; ++p) {
    if (en[p] == 0) continue;  // mpm.py:443-444: the next state of a disabled particle is left as it is
    Stencil s;
    make_stencil(c, x + 3 * p, s);
    float nv[3] = {0, 0, 0}, nC[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) {
          const int ni = s.b[0] + i, nj = s.b[1] + j, nk = s.b[2] + k;
          if (!node_ok(c, ni, nj, nk)) continue;
          const float d[3] = {((float)i - s.f[0]) * dx, ((float)j - s.f[1]) * dx, ((float)k - s.f[2]) * dx};
          const float w = s.w[0][i] * s.w[1][j] * s.w[2][k];
          const float* g = gv + 3 * ((ni * G + nj) * G + nk);
          for (int a = 0; a < 3; ++a) {
            nv[a] += w * g[a];
            for (int b = 0; b < 3; ++b) nC[3 * a + b] += kap * w * g[a] * d[b];  // outer(v, dpos), mpm.py:480
          }
        }
    float T[9];
    for (int i = 0; i < 9; ++i) T[i] = c.dt * nC[i];
    T[0] += 1.f; T[4] += 1.f; T[8] += 1.f;
    float Fo[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b)
        Fo[3 * a + b] = T[3 * a] * F[9 * p + b] + T[3 * a + 1] * F[9 * p + 3 + b] + T[3 * a + 2] * F[9 * p + 6 + b];
    const float bnd = clip[p] * dx, lo = 0.0f + bnd, hi = 1.0f - bnd;
    float xo[3];
    for (int a = 0; a < 3; ++a) xo[a] = std::min(std::max(x[3 * p + a] + c.dt * nv[a], lo), hi);
    for (int a = 0; a < 3; ++a) { xn[3 * p + a] = xo[a]; vn[3 * p + a] = nv[a]; }
    for (int i = 0; i < 9; ++i) { Cn[9 * p + i] = nC[i]; Fn[9 * p + i] = Fo[i]; }
  }
}

}  // namespace

extern "C" {

// one substep, mpm.py:279-297.  gmv (G^3,3), gm (G^3), gv (G^3,3): caller-owned dense grid (cleared here).
void ref_mpm_forward(const ref_sim_cfg* c, int32_t N, const float* vol, const float* rho, const float* clip, const int32_t* en,
                     const float* x, const float* v, const float* C, const float* F, const float* S, float* xn, float* vn,
                     float* Cn, float* Fn, float* gmv, float* gm, float* gv) {
  const size_t cells = (size_t)c->G * c->G * c->G;
  memset(gmv, 0, cells * 3 * sizeof(float));
  memset(gm, 0, cells * sizeof(float));
  memset(gv, 0, cells * 3 * sizeof(float));
  p2g(*c, N, vol, rho, en, x, v, C, S, gmv, gm);
  grid_op(*c, gmv, gm, gv);
  g2p(*c, N, clip, en, x, F, gv, xn, vn, Cn, Fn);
}

// mpm.py:299-319: clear, recompute p2g + grid_op, then the adjoints of g2p, grid_op, p2g.
// vn, Cn: the next state computed by the forward pass.  g*n: incoming gradients of the next state.  Outputs g* of the
// current state (x, v, C, F, stress).  ggv (G^3,3), ggm (G^3,4): scratch adjoint grids.
void ref_mpm_backward(const ref_sim_cfg* cp, int32_t N, const float* vol, const float* rho, const float* clip, const int32_t* en,
                      const float* x, const float* v, const float* C, const float* F, const float* S, const float* vn,
                      const float* Cn, const float* gxn, const float* gvn, const float* gCn, const float* gFn, float* gx, float* gvo,
                      float* gC, float* gF, float* gS, float* gmv, float* gm, float* gv, float* ggv, float* ggm) {
  const ref_sim_cfg& c = *cp;
  const size_t G = c.G, cells = G * G * G;
  const float inv_dx = (float)c.G, dx = 1.0f / (float)c.G, kap = 4.0f * inv_dx * inv_dx;
  memset(gmv, 0, cells * 3 * sizeof(float));
  memset(gm, 0, cells * sizeof(float));
  memset(ggv, 0, cells * 3 * sizeof(float));
  p2g(c, N, vol, rho, en, x, v, C, S, gmv, gm);
  grid_op(c, gmv, gm, gv);
  // --- g2p adjoint (the scatter of the node-velocity adjoint through a private box per thread, like p2g)
#pragma omp parallel
  {
  const int T_ = omp_get_num_threads(), t_ = omp_get_thread_num();
  const int q0 = (int)((int64_t)N * t_ / T_), q1 = (int)((int64_t)N * (t_ + 1) / T_);
  static thread_local ThreadBox box;
  box.open<3>(c, en, x, q0, q1);
  for (int p = q0; p < q1; ++p) {
    for (int a = 0; a < 3; ++a) gx[3 * p + a] = 0.f;
    for (int i = 0; i < 9; ++i) gF[9 * p + i] = 0.f;
    if (en[p] == 0) continue;
    Stencil s;
    make_stencil(c, x + 3 * p, s);
    const float bnd = clip[p] * dx, lo = 0.0f + bnd, hi = 1.0f - bnd;
    float xb[3], vt[3];
    for (int a = 0; a < 3; ++a) {
      float t = x[3 * p + a] + c.dt * vn[3 * p + a];
      float xe = (t >= lo && t <= hi) ? gxn[3 * p + a] : 0.f;
      xb[a] = xe;
      vt[a] = gvn[3 * p + a] + c.dt * xe;
    }
    float T[9], Ct[9];
    for (int i = 0; i < 9; ++i) T[i] = c.dt * Cn[9 * p + i];
    T[0] += 1.f; T[4] += 1.f; T[8] += 1.f;
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        // Fbar = T^T gF' ;  Ct = gC' + dt gF' F^T
        gF[9 * p + 3 * a + b] = T[a] * gFn[9 * p + b] + T[3 + a] * gFn[9 * p + 3 + b] + T[6 + a] * gFn[9 * p + 6 + b];
        Ct[3 * a + b] = gCn[9 * p + 3 * a + b] + c.dt * (gFn[9 * p + 3 * a] * F[9 * p + 3 * b] + gFn[9 * p + 3 * a + 1] * F[9 * p + 3 * b + 1] +
                                                        gFn[9 * p + 3 * a + 2] * F[9 * p + 3 * b + 2]);
      }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) {
          const int ni = s.b[0] + i, nj = s.b[1] + j, nk = s.b[2] + k;
          if (!node_ok(c, ni, nj, nk)) continue;
          const float d[3] = {((float)i - s.f[0]) * dx, ((float)j - s.f[1]) * dx, ((float)k - s.f[2]) * dx};
          const float w = s.w[0][i] * s.w[1][j] * s.w[2][k];
          const size_t n = (ni * G + nj) * G + nk;
          const float* g = gv + 3 * n;
          float Cd[3], Ctg[3];
          for (int a = 0; a < 3; ++a) {
            Cd[a] = Ct[3 * a] * d[0] + Ct[3 * a + 1] * d[1] + Ct[3 * a + 2] * d[2];
            Ctg[a] = Ct[a] * g[0] + Ct[3 + a] * g[1] + Ct[6 + a] * g[2];
          }
          if (box.local) {
            float* q = box.at(ni, nj, nk, 3);
            for (int a = 0; a < 3; ++a) q[a] += w * vt[a] + kap * w * Cd[a];
          } else {
            for (int a = 0; a < 3; ++a) {
              float val = w * vt[a] + kap * w * Cd[a];
#pragma omp atomic
              ggv[3 * n + a] += val;
            }
          }
          const float dLdw = vt[0] * g[0] + vt[1] * g[1] + vt[2] * g[2] + kap * (g[0] * Cd[0] + g[1] * Cd[1] + g[2] * Cd[2]);
          const float gw[3] = {s.dw[0][i] * s.w[1][j] * s.w[2][k] * inv_dx, s.w[0][i] * s.dw[1][j] * s.w[2][k] * inv_dx,
                               s.w[0][i] * s.w[1][j] * s.dw[2][k] * inv_dx};
          for (int a = 0; a < 3; ++a) xb[a] += dLdw * gw[a] - kap * w * Ctg[a];
        }
    for (int a = 0; a < 3; ++a) gx[3 * p + a] = xb[a];
  }
  if (box.local)
    for (int i = 0; i < box.n[0]; ++i)
      for (int j = 0; j < box.n[1]; ++j)
        for (int k = 0; k < box.n[2]; ++k) {
          const float* q = box.at(box.lo[0] + i, box.lo[1] + j, box.lo[2] + k, 3);
          if (q[0] == 0.f && q[1] == 0.f && q[2] == 0.f) continue;
          const size_t n = ((size_t)(box.lo[0] + i) * G + (box.lo[1] + j)) * G + (box.lo[2] + k);
          for (int a = 0; a < 3; ++a) {
#pragma omp atomic
            ggv[3 * n + a] += q[a];
          }
        }
  }
  // --- grid_op adjoint: ggv {vbar} -> ggm {mvbar xyz, mbar}
#pragma omp parallel for schedule(static)
  for (int i = 0; i < (int)G; ++i)
    for (int j = 0; j < (int)G; ++j)
      for (int k = 0; k < (int)G; ++k) {
        size_t n = ((size_t)i * G + j) * G + k;
        float o[4] = {0, 0, 0, 0};
        if (gm[n] > 0.f) {
          float u[3], mk[3];
          grid_velocity(c, i, j, k, gmv + 3 * n, gm[n], u, mk);
          const float inv = 1.f / (gm[n] + c.eps);
          float ub[3] = {ggv[3 * n] * mk[0], ggv[3 * n + 1] * mk[1], ggv[3 * n + 2] * mk[2]};
          for (int a = 0; a < 3; ++a) o[a] = ub[a] * inv;
          o[3] = -(ub[0] * gmv[3 * n] + ub[1] * gmv[3 * n + 1] + ub[2] * gmv[3 * n + 2]) * inv * inv;
        }
        for (int a = 0; a < 4; ++a) ggm[4 * n + a] = o[a];
      }
  // --- p2g adjoint
#pragma omp parallel for schedule(static)
  for (int p = 0; p < N; ++p) {
    for (int a = 0; a < 3; ++a) gvo[3 * p + a] = 0.f;
    for (int i = 0; i < 9; ++i) { gC[9 * p + i] = 0.f; gS[9 * p + i] = 0.f; }
    if (en[p] == 0) continue;
    Stencil s;
    make_stencil(c, x + 3 * p, s);
    const float pm = vol[p] * rho[p];
    const float ks = -c.dt * vol[p] * 4.0f * inv_dx * inv_dx;
    float A[9], mom[3], vb[3] = {0, 0, 0}, xb[3] = {0, 0, 0}, Ab[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 9; ++i) A[i] = ks * S[9 * p + i] + pm * C[9 * p + i];
    for (int a = 0; a < 3; ++a) mom[a] = pm * v[3 * p + a];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) {
          const int ni = s.b[0] + i, nj = s.b[1] + j, nk = s.b[2] + k;
          if (!node_ok(c, ni, nj, nk)) continue;
          const float d[3] = {((float)i - s.f[0]) * dx, ((float)j - s.f[1]) * dx, ((float)k - s.f[2]) * dx};
          const float w = s.w[0][i] * s.w[1][j] * s.w[2][k];
          const float* q = ggm + 4 * ((ni * G + nj) * G + nk);
          float Ad[3], Atq[3];
          for (int a = 0; a < 3; ++a) {
            Ad[a] = mom[a] + A[3 * a] * d[0] + A[3 * a + 1] * d[1] + A[3 * a + 2] * d[2];
            Atq[a] = A[a] * q[0] + A[3 + a] * q[1] + A[6 + a] * q[2];
            vb[a] += w * q[a];
            for (int b = 0; b < 3; ++b) Ab[3 * a + b] += w * q[a] * d[b];
          }
          const float dLdw = q[0] * Ad[0] + q[1] * Ad[1] + q[2] * Ad[2] + q[3] * pm;
          const float gw[3] = {s.dw[0][i] * s.w[1][j] * s.w[2][k] * inv_dx, s.w[0][i] * s.dw[1][j] * s.w[2][k] * inv_dx,
                               s.w[0][i] * s.w[1][j] * s.dw[2][k] * inv_dx};
          for (int a = 0; a < 3; ++a) xb[a] += dLdw * gw[a] - w * Atq[a];
        }
    for (int a = 0; a < 3; ++a) { gx[3 * p + a] += xb[a]; gvo[3 * p + a] = pm * vb[a]; }
    for (int i = 0; i < 9; ++i) { gC[9 * p + i] = pm * Ab[i]; gS[9 * p + i] = ks * Ab[i]; }
  }
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ SVD + constitutive nets
namespace {

inline double det3(const double* a) {
  return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

// F = U diag(s) V^T, U, V in SO(3), s0 >= s1 >= |s2|, sign(s2) = sign(det F)   (svd.py:61-96 convention)
void svd3(const double* F, double* U, double* s, double* V) {
  double S[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) S[3 * i + j] = F[i] * F[j] + F[3 + i] * F[3 + j] + F[6 + i] * F[6 + j];
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 16; ++sweep) {
    double off = fabs(S[1]) + fabs(S[2]) + fabs(S[5]);
    if (off < 1e-300) break;
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (auto& pq : PQ) {
      const int p = pq[0], q = pq[1];
      const double apq = S[3 * p + q];
      if (fabs(apq) < 1e-300) continue;
      const double theta = (S[3 * q + q] - S[3 * p + p]) / (2.0 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
      for (int k = 0; k < 3; ++k) {  // S <- S J
        double skp = S[3 * k + p], skq = S[3 * k + q];
        S[3 * k + p] = cs * skp - sn * skq;
        S[3 * k + q] = sn * skp + cs * skq;
      }
      for (int k = 0; k < 3; ++k) {  // S <- J^T S
        double spk = S[3 * p + k], sqk = S[3 * q + k];
        S[3 * p + k] = cs * spk - sn * sqk;
        S[3 * q + k] = sn * spk + cs * sqk;
      }
      for (int k = 0; k < 3; ++k) {
        double vkp = V[3 * k + p], vkq = V[3 * k + q];
        V[3 * k + p] = cs * vkp - sn * vkq;
        V[3 * k + q] = sn * vkp + cs * vkq;
      }
    }
  }
  int ord[3] = {0, 1, 2};
  double lam[3] = {S[0], S[4], S[8]};
  std::sort(ord, ord + 3, [&](int a, int b) { return lam[a] > lam[b]; });
  double Vs[9];
  for (int k = 0; k < 3; ++k)
    for (int j = 0; j < 3; ++j) Vs[3 * k + j] = V[3 * k + ord[j]];
  if (det3(Vs) < 0)
    for (int k = 0; k < 3; ++k) Vs[3 * k + 2] = -Vs[3 * k + 2];
  memcpy(V, Vs, sizeof(Vs));
  double B[9];  // B = F V
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) B[3 * i + j] = F[3 * i] * V[j] + F[3 * i + 1] * V[3 + j] + F[3 * i + 2] * V[6 + j];
  double u0[3] = {B[0], B[3], B[6]}, u1[3] = {B[1], B[4], B[7]};
  double n0 = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
  if (n0 > 1e-150) { for (double& t : u0) t /= n0; } else { u0[0] = 1; u0[1] = u0[2] = 0; }
  double dp = u0[0] * u1[0] + u0[1] * u1[1] + u0[2] * u1[2];
  for (int k = 0; k < 3; ++k) u1[k] -= dp * u0[k];
  double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
  if (n1 > 1e-150 * std::max(1.0, n0)) { for (double& t : u1) t /= n1; }
  else {  // rank <= 1: any unit vector orthogonal to u0
    int m = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2) : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
    double e[3] = {0, 0, 0}; e[m] = 1;
    double d2 = u0[m];
    for (int k = 0; k < 3; ++k) u1[k] = e[k] - d2 * u0[k];
    n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    for (double& t : u1) t /= n1;
  }
  double u2[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2], u0[0] * u1[1] - u0[1] * u1[0]};
  for (int k = 0; k < 3; ++k) { U[3 * k] = u0[k]; U[3 * k + 1] = u1[k]; U[3 * k + 2] = u2[k]; }
  for (int j = 0; j < 3; ++j) s[j] = U[j] * B[j] + U[3 + j] * B[3 + j] + U[6 + j] * B[6 + j];
}

inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
inline float dgelu_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

struct NetFwd {
  float U[9], s[3], V[9], R[9], z[13], a1[64], h1[64], a2[64], h2[64], X[9];
};

inline void mat3_mul(const float* A, const float* B, float* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
inline void mat3_mul_nt(const float* A, const float* B, float* C) {  // A B^T
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
inline void mat3_mul_tn(const float* A, const float* B, float* C) {  // A^T B
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// meta.py:197-217: SVD, invariants, 13 -> 64 -> 64 -> 9 MLP (no bias, exact GELU), X <- sym(X)
void net_forward(const float* F, const float* W0, const float* W1, const float* W2, NetFwd& n) {
  double Fd[9], Ud[9], sd[3], Vd[9];
  for (int i = 0; i < 9; ++i) Fd[i] = F[i];
  svd3(Fd, Ud, sd, Vd);
  for (int i = 0; i < 9; ++i) { n.U[i] = (float)Ud[i]; n.V[i] = (float)Vd[i]; }
  for (int i = 0; i < 3; ++i) n.s[i] = (float)sd[i];
  mat3_mul_nt(n.U, n.V, n.R);  // R = U V^T
  float G[9];
  mat3_mul_tn(F, F, G);
  for (int i = 0; i < 3; ++i) n.z[i] = n.s[i] - 1.0f;
  for (int i = 0; i < 9; ++i) n.z[3 + i] = G[i] - ((i % 4 == 0) ? 1.0f : 0.0f);
  n.z[12] = (float)det3(Fd) - 1.0f;
  for (int o = 0; o < 64; ++o) {
    float acc = 0.f;
    for (int k = 0; k < 13; ++k) acc += W0[13 * o + k] * n.z[k];
    n.a1[o] = acc;
    n.h1[o] = gelu_f(acc);
  }
  for (int o = 0; o < 64; ++o) {
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc += W1[64 * o + k] * n.h1[k];
    n.a2[o] = acc;
    n.h2[o] = gelu_f(acc);
  }
  float y[9];
  for (int o = 0; o < 9; ++o) {
    float acc = 0.f;
    for (int k = 0; k < 64; ++k) acc += W2[64 * o + k] * n.h2[k];
    y[o] = acc;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) n.X[3 * i + j] = 0.5f * (y[3 * i + j] + y[3 * j + i]);
}

inline void net_output(int kind, float alpha, const float* F, const NetFwd& n, float* out) {
  float RX[9];
  mat3_mul(n.R, n.X, RX);
  if (kind == 0) mat3_mul_nt(RX, F, out);                         // stress = R X F^T   (meta.py:218-221)
  else for (int i = 0; i < 9; ++i) out[i] = F[i] + alpha * RX[i];  // F + alpha R X      (meta.py:486-489)
}

// adjoint of net_forward + net_output.  gW*: this thread's weight-gradient accumulators.
void net_backward(int kind, float alpha, const float* F, const float* W0, const float* W1, const float* W2, const NetFwd& n,
                  const float* gO, float* gF, float* gW0, float* gW1, float* gW2, float clampv) {
  float gR[9], gX[9], tmp[9];
  for (int i = 0; i < 9; ++i) gF[i] = 0.f;
  if (kind == 0) {
    mat3_mul(gO, F, tmp);        // gP F
    mat3_mul(tmp, n.X, gR);      // gR = gP F X^T (X symmetric)
    mat3_mul_tn(n.R, tmp, gX);   // gX = R^T gP F
    float RX[9];
    mat3_mul(n.R, n.X, RX);
    mat3_mul_tn(gO, RX, gF);     // gF = gP^T (R X)
  } else {
    mat3_mul(gO, n.X, gR);
    mat3_mul_tn(n.R, gO, gX);
    for (int i = 0; i < 9; ++i) { gR[i] *= alpha; gX[i] *= alpha; gF[i] = gO[i]; }
  }
  float gy[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) gy[3 * i + j] = 0.5f * (gX[3 * i + j] + gX[3 * j + i]);
  float gh2[64], ga2[64], gh1[64], ga1[64], gz[13];
  for (int k = 0; k < 64; ++k) {
    float acc = 0.f;
    for (int o = 0; o < 9; ++o) acc += W2[64 * o + k] * gy[o];
    gh2[k] = acc;
  }
  for (int o = 0; o < 9; ++o)
    for (int k = 0; k < 64; ++k) gW2[64 * o + k] += gy[o] * n.h2[k];
  for (int k = 0; k < 64; ++k) ga2[k] = gh2[k] * dgelu_f(n.a2[k]);
  for (int k = 0; k < 64; ++k) gh1[k] = 0.f;
  for (int o = 0; o < 64; ++o) {
    const float g = ga2[o];
    for (int k = 0; k < 64; ++k) { gh1[k] += W1[64 * o + k] * g; gW1[64 * o + k] += g * n.h1[k]; }
  }
  for (int k = 0; k < 64; ++k) ga1[k] = gh1[k] * dgelu_f(n.a1[k]);
  for (int k = 0; k < 13; ++k) gz[k] = 0.f;
  for (int o = 0; o < 64; ++o) {
    const float g = ga1[o];
    for (int k = 0; k < 13; ++k) { gz[k] += W0[13 * o + k] * g; gW0[13 * o + k] += g * n.z[k]; }
  }
  // G = F^T F: gF += F (gG + gG^T);  det: gF += gJ cof(F)
  float gGs[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) gGs[3 * i + j] = gz[3 + 3 * i + j] + gz[3 + 3 * j + i];
  mat3_mul(F, gGs, tmp);
  for (int i = 0; i < 9; ++i) gF[i] += tmp[i];
  const float cof[9] = {F[4] * F[8] - F[5] * F[7], F[5] * F[6] - F[3] * F[8], F[3] * F[7] - F[4] * F[6],
                        F[2] * F[7] - F[1] * F[8], F[0] * F[8] - F[2] * F[6], F[1] * F[6] - F[0] * F[7],
                        F[1] * F[5] - F[2] * F[4], F[2] * F[3] - F[0] * F[5], F[0] * F[4] - F[1] * F[3]};
  for (int i = 0; i < 9; ++i) gF[i] += gz[12] * cof[i];
  // SVD adjoint (SURVEY App. B / oracle.material.svd3_adjoint): R = U Vh -> gU = gR V, gVh = U^T gR -> gV = gR^T U
  float gU[9], gV[9], UtgU[9], VtgV[9];
  mat3_mul(gR, n.V, gU);
  mat3_mul_tn(gR, n.U, gV);
  mat3_mul_tn(n.U, gU, UtgU);
  mat3_mul_tn(n.V, gV, VtgV);
  const float s2[3] = {n.s[0] * n.s[0], n.s[1] * n.s[1], n.s[2] * n.s[2]};
  float E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j) {
      float e = 1.0f / std::min(s2[j] - s2[i], -clampv);
      E[3 * i + j] = e;
      E[3 * j + i] = -e;
    }
  float inner[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float a = E[3 * i + j] * (UtgU[3 * i + j] - UtgU[3 * j + i]) * n.s[j];
      float b = n.s[i] * E[3 * i + j] * (VtgV[3 * i + j] - VtgV[3 * j + i]);
      inner[3 * i + j] = a + b + ((i == j) ? gz[i] : 0.f);
    }
  float UI[9], add[9];
  mat3_mul(n.U, inner, UI);
  mat3_mul_nt(UI, n.V, add);   // U inner V^T
  for (int i = 0; i < 9; ++i) gF[i] += add[i];
}

}  // namespace

extern "C" {

// kind 0: elasticity (stress), 1: plasticity (F + alpha R X).  Weights row-major (64,13), (64,64), (9,64).
void ref_material_forward(int32_t kind, float alpha, int32_t N, const float* F, const float* W0, const float* W1, const float* W2,
                          float* out) {
#pragma omp parallel for schedule(static)
  for (int p = 0; p < N; ++p) {
    NetFwd n;
    net_forward(F + 9 * p, W0, W1, W2, n);
    net_output(kind, alpha, F + 9 * p, n, out + 9 * p);
  }
}

void ref_svd3(int32_t N, const float* F, float* U, float* s, float* Vh) {
#pragma omp parallel for schedule(static)
  for (int p = 0; p < N; ++p) {
    double Fd[9], Ud[9], sd[3], Vd[9];
    for (int i = 0; i < 9; ++i) Fd[i] = F[9 * p + i];
    svd3(Fd, Ud, sd, Vd);
    for (int i = 0; i < 3; ++i) {
      s[3 * p + i] = (float)sd[i];
      for (int j = 0; j < 3; ++j) { U[9 * p + 3 * i + j] = (float)Ud[3 * i + j]; Vh[9 * p + 3 * i + j] = (float)Vd[3 * j + i]; }
    }
  }
}

// recompute + adjoint (what torch autograd does for the reference, minus the stored activations).  gW* (64*13, 64*64, 9*64)
// are OVERWRITTEN with the sum over particles.  svd_clamp: denominator clamp of the SVD adjoint (1e-6).
void ref_material_backward(int32_t kind, float alpha, int32_t N, const float* F, const float* W0, const float* W1, const float* W2,
                           const float* gOut, float* gF, float* gW0, float* gW1, float* gW2, float svd_clamp) {
  const int NW = 64 * 13 + 64 * 64 + 9 * 64;
  int T = ref_num_threads();
  std::vector<float> acc((size_t)T * NW, 0.f);
#pragma omp parallel
  {
#ifdef _OPENMP
    int t = omp_get_thread_num();
#else
    int t = 0;
#endif
    float* a0 = acc.data() + (size_t)t * NW;
    float* a1 = a0 + 64 * 13;
    float* a2 = a1 + 64 * 64;
#pragma omp for schedule(static)
    for (int p = 0; p < N; ++p) {
      NetFwd n;
      net_forward(F + 9 * p, W0, W1, W2, n);
      net_backward(kind, alpha, F + 9 * p, W0, W1, W2, n, gOut + 9 * p, gF + 9 * p, a0, a1, a2, svd_clamp);
    }
  }
  for (int i = 0; i < NW; ++i) {
    float sum = 0.f;
    for (int t = 0; t < T; ++t) sum += acc[(size_t)t * NW + i];
    if (i < 64 * 13) gW0[i] = sum;
    else if (i < 64 * 13 + 64 * 64) gW1[i - 64 * 13] = sum;
    else gW2[i - 64 * 13 - 64 * 64] = sum;
  }
}

// ------------------------------------------------------------------------------------------------ bindings + covariance
// out (K,D) = base (K,D or NULL) + B (CSR K x N) @ in (N,D)          tune/utils.py:424-472
void ref_spmm_csr(int32_t K, int32_t D, const int32_t* rowptr, const int32_t* col, const float* val, const float* in,
                  const float* base, float* out) {
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k)
    for (int d = 0; d < D; ++d) {
      float acc = base ? base[(size_t)k * D + d] : 0.f;
      for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) acc += val[e] * in[(size_t)col[e] * D + d];
      out[(size_t)k * D + d] = acc;
    }
}

// gin (N,D) += B^T gout (K,D)
void ref_spmm_csr_t(int32_t K, int32_t N, int32_t D, const int32_t* rowptr, const int32_t* col, const float* val, const float* gout,
                    float* gin) {
  memset(gin, 0, (size_t)N * D * sizeof(float));
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k)
    for (int e = rowptr[k]; e < rowptr[k + 1]; ++e)
      for (int d = 0; d < D; ++d) {
        float v = val[e] * gout[(size_t)k * D + d];
#pragma omp atomic
        gin[(size_t)col[e] * D + d] += v;
      }
}

// simulation_utils.py:25-48
void ref_cov_deform(int32_t K, const float* cov6, const float* F, float* out) {
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k) {
    const float* c = cov6 + 6 * k;
    const float S[9] = {c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]};
    float FS[9], R[9];
    mat3_mul(F + 9 * k, S, FS);
    mat3_mul_nt(FS, F + 9 * k, R);
    float* o = out + 6 * k;
    o[0] = R[0]; o[1] = R[1]; o[2] = R[2]; o[3] = R[4]; o[4] = R[5]; o[5] = R[8];
  }
}

// loss_utils.py:17-24 over (3,H,W): kind 0 l1, 1 l2.  Returns the loss, writes weight * dL/dimg.
double ref_pixel_loss(int32_t kind, int64_t n, const float* img, const float* gt, float weight, float* gimg) {
  double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float d = img[i] - gt[i];
    if (kind == 0) { acc += fabsf(d); gimg[i] = weight * ((d > 0.f) - (d < 0.f)) / (float)n; }
    else { acc += (double)d * d; gimg[i] = weight * 2.0f * d / (float)n; }
  }
  return acc / (double)n;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ rasterizer
namespace {

const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                        -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// basis values b[16] and gradients db[16][3] w.r.t. the unit direction (x, y, z)
void sh_basis(int deg, float x, float y, float z, float* b, float (*db)[3]) {
  for (int i = 0; i < 16; ++i) { b[i] = 0.f; db[i][0] = db[i][1] = db[i][2] = 0.f; }
  b[0] = SH_C0;
  if (deg < 1) return;
  b[1] = -SH_C1 * y; db[1][1] = -SH_C1;
  b[2] = SH_C1 * z;  db[2][2] = SH_C1;
  b[3] = -SH_C1 * x; db[3][0] = -SH_C1;
  if (deg < 2) return;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  b[4] = SH_C2[0] * xy; db[4][0] = SH_C2[0] * y; db[4][1] = SH_C2[0] * x;
  b[5] = SH_C2[1] * yz; db[5][1] = SH_C2[1] * z; db[5][2] = SH_C2[1] * y;
  b[6] = SH_C2[2] * (2.f * zz - xx - yy); db[6][0] = SH_C2[2] * -2.f * x; db[6][1] = SH_C2[2] * -2.f * y; db[6][2] = SH_C2[2] * 4.f * z;
  b[7] = SH_C2[3] * xz; db[7][0] = SH_C2[3] * z; db[7][2] = SH_C2[3] * x;
  b[8] = SH_C2[4] * (xx - yy); db[8][0] = SH_C2[4] * 2.f * x; db[8][1] = SH_C2[4] * -2.f * y;
  if (deg < 3) return;
  b[9] = SH_C3[0] * y * (3.f * xx - yy); db[9][0] = SH_C3[0] * 6.f * xy; db[9][1] = SH_C3[0] * (3.f * xx - 3.f * yy);
  b[10] = SH_C3[1] * xy * z; db[10][0] = SH_C3[1] * yz; db[10][1] = SH_C3[1] * xz; db[10][2] = SH_C3[1] * xy;
  b[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
  db[11][0] = SH_C3[2] * -2.f * xy; db[11][1] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); db[11][2] = SH_C3[2] * 8.f * yz;
  b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
  db[12][0] = SH_C3[3] * -6.f * xz; db[12][1] = SH_C3[3] * -6.f * yz; db[12][2] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
  b[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
  db[13][0] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); db[13][1] = SH_C3[4] * -2.f * xy; db[13][2] = SH_C3[4] * 8.f * xz;
  b[14] = SH_C3[5] * z * (xx - yy); db[14][0] = SH_C3[5] * 2.f * xz; db[14][1] = SH_C3[5] * -2.f * yz; db[14][2] = SH_C3[5] * (xx - yy);
  b[15] = SH_C3[6] * x * (xx - 3.f * yy); db[15][0] = SH_C3[6] * (3.f * xx - 3.f * yy); db[15][1] = SH_C3[6] * -6.f * xy;
}

}  // namespace

extern "C" {

struct ref_cam {
  int32_t H, W;
  float tanfovx, tanfovy;
  float bg[3];
  float view[16];  // row-major 4x4 as torch holds it (cameras.py:54-56: already transposed, row-vector convention)
  float proj[16];
  int32_t sh_degree;
  float campos[3];
};

struct ref_raster_state {
  ref_cam cam;
  int K, M, gx, gy;
  const float *means3D, *shs, *colors, *opac, *cov6;
  std::vector<float> xy, depth, conic, rgb, txc;  // per Gaussian: 2, 1, 3, 3, (tx,ty,tz clamped: 3)
  std::vector<unsigned char> clamped, vis, xmul;  // 3 / 1 / 2 per Gaussian
  std::vector<int> rect;                          // 4 per Gaussian
  std::vector<int> tile_start, list;              // per tile ranges into `list` (Gaussian ids sorted by (depth, id))
  std::vector<float> Tfinal;                      // per pixel
  std::vector<int> ncontrib;                      // per pixel: index in the tile list after the last contributor
  int64_t D;
};

// preprocess + binning + composite.  image (3,H,W), radii (K).  Returns a state handle for ref_raster_backward.
ref_raster_state* ref_raster_forward(const ref_cam* cam, int32_t K, int32_t M, const float* means3D, const float* shs,
                                     const float* colors, const float* opac, const float* cov6, float* image, int32_t* radii) {
  ref_raster_state* st = new ref_raster_state();
  st->cam = *cam;
  st->K = K; st->M = M;
  st->means3D = means3D; st->shs = shs; st->colors = colors; st->opac = opac; st->cov6 = cov6;
  const int W = cam->W, H = cam->H;
  const int gx = (W + 15) / 16, gy = (H + 15) / 16;
  st->gx = gx; st->gy = gy;
  st->xy.assign(2 * (size_t)K, 0.f); st->depth.assign(K, 0.f); st->conic.assign(3 * (size_t)K, 0.f); st->rgb.assign(3 * (size_t)K, 0.f);
  st->txc.assign(3 * (size_t)K, 0.f); st->clamped.assign(3 * (size_t)K, 0); st->vis.assign(K, 0); st->xmul.assign(2 * (size_t)K, 1);
  st->rect.assign(4 * (size_t)K, 0);
  const float fx = W / (2.0f * cam->tanfovx), fy = H / (2.0f * cam->tanfovy);
  const float limx = 1.3f * cam->tanfovx, limy = 1.3f * cam->tanfovy;
  const float* V = cam->view; const float* P = cam->proj;
  std::vector<int> tiles_touched(K, 0);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k) {
    radii[k] = 0;
    const float* mu = means3D + 3 * k;
    float ph[4], t[3];
    for (int j = 0; j < 4; ++j) ph[j] = mu[0] * P[j] + mu[1] * P[4 + j] + mu[2] * P[8 + j] + P[12 + j];
    for (int j = 0; j < 3; ++j) t[j] = mu[0] * V[j] + mu[1] * V[4 + j] + mu[2] * V[8 + j] + V[12 + j];
    st->depth[k] = t[2];
    if (!(t[2] > 0.2f)) continue;
    const float pw = 1.0f / (ph[3] + 1e-7f);
    const float px = ph[0] * pw, py = ph[1] * pw;
    float rx = t[0] / t[2], ry = t[1] / t[2];
    const bool inx = rx >= -limx && rx <= limx, iny = ry >= -limy && ry <= limy;
    const float tx = inx ? t[0] : std::min(limx, std::max(-limx, rx)) * t[2];
    const float ty = iny ? t[1] : std::min(limy, std::max(-limy, ry)) * t[2];
    st->xmul[2 * k] = inx; st->xmul[2 * k + 1] = iny;
    st->txc[3 * k] = tx; st->txc[3 * k + 1] = ty; st->txc[3 * k + 2] = t[2];
    const float tz = t[2];
    const float J[6] = {fx / tz, 0.f, -(fx * tx) / (tz * tz), 0.f, fy / tz, -(fy * ty) / (tz * tz)};
    float Mm[6];  // M = J Rv, Rv[i][j] = V[j][i]
    for (int r = 0; r < 2; ++r)
      for (int j = 0; j < 3; ++j) Mm[3 * r + j] = J[3 * r] * V[4 * j] + J[3 * r + 1] * V[4 * j + 1] + J[3 * r + 2] * V[4 * j + 2];
    const float* c6 = cov6 + 6 * k;
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    float MS[6];
    for (int r = 0; r < 2; ++r)
      for (int j = 0; j < 3; ++j) MS[3 * r + j] = Mm[3 * r] * S[j] + Mm[3 * r + 1] * S[3 + j] + Mm[3 * r + 2] * S[6 + j];
    const float a = MS[0] * Mm[0] + MS[1] * Mm[1] + MS[2] * Mm[2] + 0.3f;
    const float b = MS[0] * Mm[3] + MS[1] * Mm[4] + MS[2] * Mm[5];
    const float cc = MS[3] * Mm[3] + MS[4] * Mm[4] + MS[5] * Mm[5] + 0.3f;
    const float det = a * cc - b * b;
    if (det == 0.f) continue;
    st->conic[3 * k] = cc / det; st->conic[3 * k + 1] = -b / det; st->conic[3 * k + 2] = a / det;
    const float mid = 0.5f * (a + cc);
    const float lam = mid + sqrtf(std::max(0.1f, mid * mid - det));
    const int radius = (int)ceilf(3.0f * sqrtf(lam));
    const float mx = ((px + 1.0f) * W - 1.0f) * 0.5f, my = ((py + 1.0f) * H - 1.0f) * 0.5f;
    st->xy[2 * k] = mx; st->xy[2 * k + 1] = my;
    auto clampi = [](int v, int lo, int hi) { return std::min(hi, std::max(lo, v)); };
    int r0x = clampi((int)((mx - radius) / 16.f), 0, gx), r0y = clampi((int)((my - radius) / 16.f), 0, gy);
    int r1x = clampi((int)((mx + radius + 15) / 16.f), 0, gx), r1y = clampi((int)((my + radius + 15) / 16.f), 0, gy);
    if ((r1x - r0x) * (r1y - r0y) == 0) continue;
    st->rect[4 * k] = r0x; st->rect[4 * k + 1] = r0y; st->rect[4 * k + 2] = r1x; st->rect[4 * k + 3] = r1y;
    if (colors) {
      for (int ch = 0; ch < 3; ++ch) st->rgb[3 * k + ch] = colors[3 * k + ch];
    } else {
      float d[3] = {mu[0] - cam->campos[0], mu[1] - cam->campos[1], mu[2] - cam->campos[2]};
      float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      float bas[16], dbas[16][3];
      sh_basis(cam->sh_degree, d[0] * inv, d[1] * inv, d[2] * inv, bas, dbas);
      const int nb = (cam->sh_degree + 1) * (cam->sh_degree + 1);
      for (int ch = 0; ch < 3; ++ch) {
        float acc = 0.f;
        for (int i = 0; i < nb; ++i) acc += bas[i] * shs[((size_t)k * M + i) * 3 + ch];
        acc += 0.5f;
        st->clamped[3 * k + ch] = acc < 0.f;
        st->rgb[3 * k + ch] = std::max(acc, 0.f);
      }
    }
    st->vis[k] = 1;
    radii[k] = radius;
    tiles_touched[k] = (r1x - r0x) * (r1y - r0y);
  }
  // binning: per-tile lists, sorted by (depth, Gaussian id)
  const int ntiles = gx * gy;
  std::vector<int> cnt(ntiles + 1, 0);
  for (int k = 0; k < K; ++k)
    if (st->vis[k])
      for (int ty = st->rect[4 * k + 1]; ty < st->rect[4 * k + 3]; ++ty)
        for (int tx = st->rect[4 * k]; tx < st->rect[4 * k + 2]; ++tx) cnt[ty * gx + tx + 1]++;
  for (int t = 0; t < ntiles; ++t) cnt[t + 1] += cnt[t];
  st->tile_start = cnt;
  st->D = cnt[ntiles];
  st->list.assign((size_t)st->D, 0);
  {
    std::vector<int> cur(cnt.begin(), cnt.end() - 1);
    for (int k = 0; k < K; ++k)
      if (st->vis[k])
        for (int ty = st->rect[4 * k + 1]; ty < st->rect[4 * k + 3]; ++ty)
          for (int tx = st->rect[4 * k]; tx < st->rect[4 * k + 2]; ++tx) st->list[cur[ty * gx + tx]++] = k;
  }
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < ntiles; ++t)
    std::stable_sort(st->list.begin() + st->tile_start[t], st->list.begin() + st->tile_start[t + 1],
                     [&](int a, int b) { return st->depth[a] < st->depth[b]; });
  // composite, one pixel at a time, front to back
  st->Tfinal.assign((size_t)W * H, 1.f);
  st->ncontrib.assign((size_t)W * H, 0);
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < ntiles; ++t) {
    const int tx0 = (t % gx) * 16, ty0 = (t / gx) * 16;
    const int b0 = st->tile_start[t], b1 = st->tile_start[t + 1];
    for (int py = ty0; py < std::min(ty0 + 16, H); ++py)
      for (int px = tx0; px < std::min(tx0 + 16, W); ++px) {
        float T = 1.f, Cc[3] = {0, 0, 0};
        int last = 0;
        for (int e = b0; e < b1; ++e) {
          const int k = st->list[e];
          const float dx = st->xy[2 * k] - (float)px, dy = st->xy[2 * k + 1] - (float)py;
          const float* co = &st->conic[3 * k];
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.f) continue;
          const float alpha = std::min(0.99f, opac[k] * expf(power));
          if (alpha < 1.0f / 255.0f) continue;
          const float Tn = T * (1.f - alpha);
          if (Tn < 0.0001f) break;
          for (int ch = 0; ch < 3; ++ch) Cc[ch] += st->rgb[3 * k + ch] * alpha * T;
          T = Tn;
          last = e - b0 + 1;
        }
        const size_t pix = (size_t)py * W + px;
        st->Tfinal[pix] = T;
        st->ncontrib[pix] = last;
        for (int ch = 0; ch < 3; ++ch) image[(size_t)ch * H * W + pix] = Cc[ch] + T * cam->bg[ch];
      }
  }
  return st;
}

int64_t ref_raster_pairs(const ref_raster_state* st) { return st->D; }
void ref_raster_free(ref_raster_state* st) { delete st; }

// dL/dimage (3,H,W) -> dL/dmeans3D (K,3) (the only gradient NeuMA consumes) and, optionally,
// dL/dcov6 (K,6), dL/dopacity (K), dL/dshs (K,M,3) or dL/dcolors (K,3) (NULL = not wanted)
void ref_raster_backward(ref_raster_state* st, const float* gimg, float* dmeans3D, float* dcov6, float* dopac, float* dsh,
                         float* dcol) {
  const ref_cam& cam = st->cam;
  const int W = cam.W, H = cam.H, K = st->K, gx = st->gx, ntiles = st->gx * st->gy;
  std::vector<float> dxy(2 * (size_t)K, 0.f), dcon(3 * (size_t)K, 0.f), drgb(3 * (size_t)K, 0.f), dop((size_t)K, 0.f);
  // A tile's 256 pixels are walked by ONE thread: their contributions to a Gaussian are summed in a tile-private table (one
  // row of 9 per list entry) and leave it with one atomic set per (tile, Gaussian) - not one per (pixel, Gaussian): the
  // lock-prefixed float adds were most of this function's time, and the only thing in it that did not scale with the cores.
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < ntiles; ++t) {
    const int tx0 = (t % gx) * 16, ty0 = (t / gx) * 16;
    const int b0 = st->tile_start[t];
    static thread_local std::vector<float> loc;
    const int nlist = st->tile_start[t + 1] - b0;
    loc.assign((size_t)9 * nlist, 0.f);
    for (int py = ty0; py < std::min(ty0 + 16, H); ++py)
      for (int px = tx0; px < std::min(tx0 + 16, W); ++px) {
        const size_t pix = (size_t)py * W + px;
        const float Tfin = st->Tfinal[pix];
        float T = Tfin;
        const float g[3] = {gimg[pix], gimg[(size_t)H * W + pix], gimg[2 * (size_t)H * W + pix]};
        float accum[3] = {0, 0, 0}, last_alpha = 0.f, last_c[3] = {0, 0, 0};
        for (int e = b0 + st->ncontrib[pix] - 1; e >= b0; --e) {
          const int k = st->list[e];
          const float dx = st->xy[2 * k] - (float)px, dy = st->xy[2 * k + 1] - (float)py;
          const float* co = &st->conic[3 * k];
          const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > 0.f) continue;
          const float Gv = expf(power);
          const float alpha = std::min(0.99f, st->opac[k] * Gv);
          if (alpha < 1.0f / 255.0f) continue;
          T = T / (1.f - alpha);
          float dL_dalpha = 0.f;
          for (int ch = 0; ch < 3; ++ch) {
            const float c = st->rgb[3 * k + ch];
            accum[ch] = last_alpha * last_c[ch] + (1.f - last_alpha) * accum[ch];
            last_c[ch] = c;
            dL_dalpha += (c - accum[ch]) * g[ch];
            loc[9 * (size_t)(e - b0) + ch] += alpha * T * g[ch];
          }
          dL_dalpha *= T;
          last_alpha = alpha;
          float bgdot = cam.bg[0] * g[0] + cam.bg[1] * g[1] + cam.bg[2] * g[2];
          dL_dalpha += (-Tfin / (1.f - alpha)) * bgdot;
          const float dL_dG = st->opac[k] * dL_dalpha;  // straight-through the 0.99 clamp, like upstream
          const float gdx = Gv * (-co[0] * dx - co[1] * dy), gdy = Gv * (-co[2] * dy - co[1] * dx);
          const float v0 = dL_dG * gdx, v1 = dL_dG * gdy;
          const float c0 = -0.5f * Gv * dx * dx * dL_dG, c1 = -Gv * dx * dy * dL_dG, c2 = -0.5f * Gv * dy * dy * dL_dG;
          const float o = Gv * dL_dalpha;
          float* row = &loc[9 * (size_t)(e - b0)];
          row[3] += v0; row[4] += v1; row[5] += c0; row[6] += c1; row[7] += c2; row[8] += o;
        }
      }
    for (int e = 0; e < nlist; ++e) {
      const float* row = &loc[9 * (size_t)e];
      bool any = false;
      for (int q = 0; q < 9; ++q) any = any || row[q] != 0.f;
      if (!any) continue;
      const int k = st->list[b0 + e];
      float* dst[9] = {&drgb[3 * k], &drgb[3 * k + 1], &drgb[3 * k + 2], &dxy[2 * k], &dxy[2 * k + 1],
                       &dcon[3 * k], &dcon[3 * k + 1], &dcon[3 * k + 2], &dop[k]};
      for (int q = 0; q < 9; ++q) {
        const float v = row[q];
#pragma omp atomic
        *dst[q] += v;
      }
    }
  }
  const float fx = W / (2.0f * cam.tanfovx), fy = H / (2.0f * cam.tanfovy);
  const float* V = cam.view; const float* P = cam.proj;
  const int M = st->M;
#pragma omp parallel for schedule(static)
  for (int k = 0; k < K; ++k) {
    float gm[3] = {0, 0, 0};
    if (dcov6) for (int i = 0; i < 6; ++i) dcov6[6 * k + i] = 0.f;
    if (dopac) dopac[k] = st->vis[k] ? dop[k] : 0.f;
    if (dcol) for (int ch = 0; ch < 3; ++ch) dcol[3 * k + ch] = st->vis[k] ? drgb[3 * k + ch] : 0.f;
    if (dsh) for (int i = 0; i < 3 * M; ++i) dsh[(size_t)k * 3 * M + i] = 0.f;
    if (st->vis[k]) {
      const float* mu = st->means3D + 3 * k;
      // (i) through the 2-D mean
      float ph[4];
      for (int j = 0; j < 4; ++j) ph[j] = mu[0] * P[j] + mu[1] * P[4 + j] + mu[2] * P[8 + j] + P[12 + j];
      const float pw = 1.0f / (ph[3] + 1e-7f);
      const float gpx = dxy[2 * k] * 0.5f * W, gpy = dxy[2 * k + 1] * 0.5f * H;
      const float gph[4] = {gpx * pw, gpy * pw, 0.f, -(gpx * ph[0] + gpy * ph[1]) * pw * pw};
      for (int i = 0; i < 3; ++i) gm[i] += P[4 * i] * gph[0] + P[4 * i + 1] * gph[1] + P[4 * i + 3] * gph[3];
      // (ii) through cov2D's dependence on the view-space position
      const float tx = st->txc[3 * k], ty = st->txc[3 * k + 1], tz = st->txc[3 * k + 2];
      const float J[6] = {fx / tz, 0.f, -(fx * tx) / (tz * tz), 0.f, fy / tz, -(fy * ty) / (tz * tz)};
      float Mm[6];
      for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 3; ++j) Mm[3 * r + j] = J[3 * r] * V[4 * j] + J[3 * r + 1] * V[4 * j + 1] + J[3 * r + 2] * V[4 * j + 2];
      const float* c6 = st->cov6 + 6 * k;
      const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
      float MS[6];
      for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 3; ++j) MS[3 * r + j] = Mm[3 * r] * S[j] + Mm[3 * r + 1] * S[3 + j] + Mm[3 * r + 2] * S[6 + j];
      const float a = MS[0] * Mm[0] + MS[1] * Mm[1] + MS[2] * Mm[2] + 0.3f;
      const float b = MS[0] * Mm[3] + MS[1] * Mm[4] + MS[2] * Mm[5];
      const float cc = MS[3] * Mm[3] + MS[4] * Mm[4] + MS[5] * Mm[5] + 0.3f;
      const float det = a * cc - b * b;
      const float d2 = 1.0f / (det * det + 1e-7f);
      const float gX = dcon[3 * k], gY = dcon[3 * k + 1], gZ = dcon[3 * k + 2];
      const float dLa = d2 * (-cc * cc * gX + b * cc * gY - b * b * gZ);
      const float dLb = d2 * (2.f * b * cc * gX - (a * cc + b * b) * gY + 2.f * a * b * gZ);
      const float dLc = d2 * (-b * b * gX + a * b * gY - a * a * gZ);
      const float Gc[4] = {dLa, 0.5f * dLb, 0.5f * dLb, dLc};
      float dM[6];  // dL/dM = 2 Gc M Sigma
      for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 3; ++j) dM[3 * r + j] = 2.f * (Gc[2 * r] * MS[j] + Gc[2 * r + 1] * MS[3 + j]);
      if (dcov6) {  // dL/dSigma = M^T Gc M (symmetric), packed with the off-diagonals counted twice
        float MtG[6];
        for (int i = 0; i < 3; ++i)
          for (int r = 0; r < 2; ++r) MtG[2 * i + r] = Mm[i] * Gc[r] + Mm[3 + i] * Gc[2 + r];
        float dS[9];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) dS[3 * i + j] = MtG[2 * i] * Mm[j] + MtG[2 * i + 1] * Mm[3 + j];
        float* o = dcov6 + 6 * k;
        o[0] = dS[0]; o[1] = dS[1] + dS[3]; o[2] = dS[2] + dS[6]; o[3] = dS[4]; o[4] = dS[5] + dS[7]; o[5] = dS[8];
      }
      float dJ[6];  // dL/dJ = dL/dM Rv^T, Rv^T[j][i] = V[i][j]... Rv[i][j] = V[j][i] => (Rv^T)[j][c] = V[j][c]... see below
      for (int r = 0; r < 2; ++r)
        for (int cidx = 0; cidx < 3; ++cidx)  // dJ[r][c] = sum_j dM[r][j] Rv[c][j] = sum_j dM[r][j] V[j][c]
          dJ[3 * r + cidx] = dM[3 * r] * V[cidx] + dM[3 * r + 1] * V[4 + cidx] + dM[3 * r + 2] * V[8 + cidx];
      const float tz2 = 1.0f / (tz * tz), tz3 = tz2 / tz;
      float gt[3];
      gt[0] = st->xmul[2 * k] ? -fx * tz2 * dJ[2] : 0.f;
      gt[1] = st->xmul[2 * k + 1] ? -fy * tz2 * dJ[5] : 0.f;
      gt[2] = -fx * tz2 * dJ[0] - fy * tz2 * dJ[4] + 2.f * fx * tx * tz3 * dJ[2] + 2.f * fy * ty * tz3 * dJ[5];
      for (int i = 0; i < 3; ++i) gm[i] += V[4 * i] * gt[0] + V[4 * i + 1] * gt[1] + V[4 * i + 2] * gt[2];
      // (iii) through the SH view direction
      if (!st->colors) {
        float d[3] = {mu[0] - cam.campos[0], mu[1] - cam.campos[1], mu[2] - cam.campos[2]};
        const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float dir[3] = {d[0] * inv, d[1] * inv, d[2] * inv};
        float bas[16], dbas[16][3];
        sh_basis(cam.sh_degree, dir[0], dir[1], dir[2], bas, dbas);
        const int nb = (cam.sh_degree + 1) * (cam.sh_degree + 1);
        float gdir[3] = {0, 0, 0};
        for (int ch = 0; ch < 3; ++ch) {
          if (st->clamped[3 * k + ch]) continue;
          const float gc = drgb[3 * k + ch];
          for (int i = 0; i < nb; ++i) {
            const float sv = st->shs[((size_t)k * M + i) * 3 + ch];
            for (int ax = 0; ax < 3; ++ax) gdir[ax] += gc * sv * dbas[i][ax];
            if (dsh) dsh[((size_t)k * M + i) * 3 + ch] = gc * bas[i];
          }
        }
        const float dp = dir[0] * gdir[0] + dir[1] * gdir[1] + dir[2] * gdir[2];
        for (int ax = 0; ax < 3; ++ax) gm[ax] += (gdir[ax] - dir[ax] * dp) * inv;
      }
    }
    for (int i = 0; i < 3; ++i) dmeans3D[3 * k + i] = gm[i];
  }
}

}  // extern "C"
