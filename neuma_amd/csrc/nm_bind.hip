// Particle -> Gaussian binding operators and the pixel loss (all HBM-bound, one pass each).
//   nm_spmm_csr    torch.sparse.mm(bindings, X) of modules/tune/utils.py:424-472 on a CSR copy
//   nm_cov_deform  modules/d3gs/utils/simulation_utils.py:25-48
//   nm_bind_frame  both spmm's + the covariance push-forward fused (F_k stays in registers)
//   nm_pixel_loss  modules/d3gs/utils/loss_utils.py:17-24 fused with its gradient
#include "nm_common.h"

// one 16-lane group per output row; lanes stride the D columns (D <= 16)
__global__ void __launch_bounds__(256) k_spmm_csr(int rows, int D, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const float* __restrict__ val, const float* __restrict__ in,
                                                  float* __restrict__ out) {
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int c = threadIdx.x & 15;
  if (r >= rows) return;
  const int b = rowptr[r], e = rowptr[r + 1];
  if (D <= 4) {
    // few columns (the frame's B^T dL/dmeans3D has three): the sixteen lanes share the row's ENTRIES instead of its columns -
    // with a lane per column 13 of 16 lanes idled and every lane walked the whole row through a chain of dependent loads
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int q = b + c; q < e; q += 16) {
      const float w = val[q];
      const float* src = in + (size_t)col[q] * D;
      a0 += w * src[0];
      if (D > 1) a1 += w * src[1];
      if (D > 2) a2 += w * src[2];
      if (D > 3) a3 += w * src[3];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      a0 += __shfl_xor(a0, o, 16); a1 += __shfl_xor(a1, o, 16); a2 += __shfl_xor(a2, o, 16); a3 += __shfl_xor(a3, o, 16);
    }
    if (c < D) out[(size_t)r * D + c] = c == 0 ? a0 : (c == 1 ? a1 : (c == 2 ? a2 : a3));
    return;
  }
  for (int cc = c; cc < D; cc += 16) {
    float acc = 0.f;
    for (int q = b; q < e; ++q) acc += val[q] * in[(size_t)col[q] * D + cc];
    out[(size_t)r * D + cc] = acc;
  }
}

extern "C" int nm_spmm_csr(int32_t rows, int32_t D, const int32_t* rowptr, const int32_t* col, const float* val,
                           const float* in, float* out, void* stream) {
  NM_REQUIRE(rows >= 0 && D > 0, "bad shape");
  if (rows == 0) return NM_OK;
  NM_REQUIRE(rowptr && col && val && in && out, "null pointer");
  NM_LAUNCH(k_spmm_csr, dim3(nm_div_up((int64_t)rows * 16, 256)), dim3(256), 0, (hipStream_t)stream, rows, D, rowptr,
                     col, val, in, out);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

// out = scale * A (in0 + in1 + in2 + in3), three columns: B^T of the views' summed dL/dmeans3D and the 1 / size of finetune.py:373
// in one pass (the frame's reverse sweep ran two adds, the product and a division as four launches between the renders' adjoints
// and the roll-out's).  Sixteen lanes share a row's entries, as in k_spmm_csr's D <= 4 branch.
__global__ void __launch_bounds__(256) k_spmm_csr_sum3(int rows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                       const float* __restrict__ val, const float* __restrict__ in0,
                                                       const float* __restrict__ in1, const float* __restrict__ in2,
                                                       const float* __restrict__ in3, float scale, float* __restrict__ out) {
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int c = threadIdx.x & 15;
  if (r >= rows) return;
  const int b = rowptr[r], e = rowptr[r + 1];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int q = b + c; q < e; q += 16) {
    const float w = val[q];
    const size_t o = (size_t)col[q] * 3;
    float s0 = in0[o], s1 = in0[o + 1], s2 = in0[o + 2];
    if (in1) { s0 += in1[o]; s1 += in1[o + 1]; s2 += in1[o + 2]; }
    if (in2) { s0 += in2[o]; s1 += in2[o + 1]; s2 += in2[o + 2]; }
    if (in3) { s0 += in3[o]; s1 += in3[o + 1]; s2 += in3[o + 2]; }
    a0 += w * s0; a1 += w * s1; a2 += w * s2;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o, 16); a1 += __shfl_xor(a1, o, 16); a2 += __shfl_xor(a2, o, 16);
  }
  if (c < 3) out[(size_t)r * 3 + c] = scale * (c == 0 ? a0 : (c == 1 ? a1 : a2));
}

extern "C" int nm_spmm_csr_sum3(int32_t rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* in0,
                                const float* in1, const float* in2, const float* in3, float scale, float* out, void* stream) {
  NM_REQUIRE(rows >= 0, "bad shape");
  if (rows == 0) return NM_OK;
  NM_REQUIRE(rowptr && col && val && in0 && out, "null pointer");
  NM_REQUIRE(in2 || !in3, "inputs must be given in order (in3 without in2)");
  NM_REQUIRE(in1 || !in2, "inputs must be given in order (in2 without in1)");
  NM_LAUNCH(k_spmm_csr_sum3, dim3(nm_div_up((int64_t)rows * 16, 256)), dim3(256), 0, (hipStream_t)stream, rows, rowptr, col, val,
                     in0, in1, in2, in3, scale, out);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

__device__ __forceinline__ void cov_push(const float* __restrict__ c6, const M3& Fm, float* __restrict__ o6) {
  M3 S;
  S.m[0] = c6[0]; S.m[1] = c6[1]; S.m[2] = c6[2];
  S.m[3] = c6[1]; S.m[4] = c6[3]; S.m[5] = c6[4];
  S.m[6] = c6[2]; S.m[7] = c6[4]; S.m[8] = c6[5];
  M3 R = m3_mul_nt(m3_mul(Fm, S), Fm);  // F S F^T
  o6[0] = R.m[0]; o6[1] = R.m[1]; o6[2] = R.m[2]; o6[3] = R.m[4]; o6[4] = R.m[5]; o6[5] = R.m[8];
}

__global__ void __launch_bounds__(256) k_cov_deform(int k, const float* __restrict__ cov6, const float* __restrict__ F,
                                                    float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  float c[6], o[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) c[a] = cov6[6 * i + a];
  cov_push(c, m3_load(F + 9 * i), o);
#pragma unroll
  for (int a = 0; a < 6; ++a) out[6 * i + a] = o[a];
}

extern "C" int nm_cov_deform(int32_t k, const float* cov6, const float* F, float* out_cov6, void* stream) {
  NM_REQUIRE(k >= 0, "negative k");
  if (k == 0) return NM_OK;
  NM_REQUIRE(cov6 && F && out_cov6, "null pointer");
  NM_LAUNCH(k_cov_deform, dim3(nm_div_up(k, 256)), dim3(256), 0, (hipStream_t)stream, k, cov6, F, out_cov6);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

__global__ void __launch_bounds__(256) k_bind_frame(int k, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                    const float* __restrict__ val, const float* __restrict__ pc,
                                                    const float* __restrict__ pp, const float* __restrict__ kp,
                                                    const float* __restrict__ F, const float* __restrict__ cov6,
                                                    float* __restrict__ means, float* __restrict__ cov_out,
                                                    float* __restrict__ F_out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  float d[3] = {0.f, 0.f, 0.f};
  M3 Fk = m3_zero();
  // four bound particles per trip, all their loads issued before the first use (a rolled loop pays the column-index round
  // trip and then the gather round trip for every particle, one after the other); summation order unchanged
  const int q1 = rowptr[i + 1];
  for (int q = rowptr[i]; q < q1; q += 4) {
    int c[4];
    float w[4], dp[4][3], Fv[4][9];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = q + u < q1;
      c[u] = col[ok ? q + u : q];
      w[u] = ok ? val[q + u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int a = 0; a < 3; ++a) dp[u][a] = pc[3 * c[u] + a] - pp[3 * c[u] + a];
      if (F) {
#pragma unroll
        for (int a = 0; a < 9; ++a) Fv[u][a] = F[9 * (size_t)c[u] + a];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (q + u < q1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) d[a] += w[u] * dp[u][a];
        if (F) {
#pragma unroll
          for (int a = 0; a < 9; ++a) Fk.m[a] += w[u] * Fv[u][a];
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) means[3 * i + a] = kp[3 * i + a] + d[a];
  if (F && cov_out) {
    float c6[6], o[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) c6[a] = cov6[6 * i + a];
    cov_push(c6, Fk, o);
#pragma unroll
    for (int a = 0; a < 6; ++a) cov_out[6 * i + a] = o[a];
  }
  if (F && F_out) m3_store(F_out + 9 * (size_t)i, Fk);
}

extern "C" int nm_bind_frame(int32_t k, const int32_t* rowptr, const int32_t* col, const float* val, const float* p_cur,
                             const float* p_prev, const float* k_prev, const float* F, const float* cov6, float* means3D,
                             float* cov6_out, float* F_out, void* stream) {
  NM_REQUIRE(k >= 0, "negative k");
  if (k == 0) return NM_OK;
  NM_REQUIRE(rowptr && col && val && p_cur && p_prev && k_prev && means3D, "null pointer");
  NM_REQUIRE(!cov6_out || (F && cov6), "cov6_out needs F and cov6");
  NM_LAUNCH(k_bind_frame, dim3(nm_div_up(k, 256)), dim3(256), 0, (hipStream_t)stream, k, rowptr, col, val, p_cur,
                     p_prev, k_prev, F, cov6, means3D, cov6_out, F_out);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

// loss over (3,H,W): rows [row0,row1) only; grad written for every pixel (zero outside the stripe).  One image row per
// loop iteration of a workgroup (no per-element index arithmetic), 16-byte accesses when the row length allows it.
template <bool VEC4>
__global__ void __launch_bounds__(256) k_pixel_loss(int kind, float scale, int h, int w, int row0, int row1,
                                                    const float* __restrict__ img, const float* __restrict__ gt,
                                                    float* __restrict__ loss, float* __restrict__ grad) {
  float acc = 0.f;
  for (int r = blockIdx.x; r < 3 * h; r += gridDim.x) {
    const int y = r % h;
    const bool in = y >= row0 && y < row1;
    const size_t base = (size_t)r * w;
    if (VEC4) {
      const float4* a4 = reinterpret_cast<const float4*>(img + base);
      typedef float f4v __attribute__((ext_vector_type(4)));
      const f4v* g4 = reinterpret_cast<const f4v*>(gt + base);      // the ground truth is read once per frame: streamed past the caches
      float4* o4 = grad ? reinterpret_cast<float4*>(grad + base) : nullptr;
      for (int i = threadIdx.x; i < w / 4; i += 256) {
        float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) {
          const float4 a = a4[i];
          const f4v g = __builtin_nontemporal_load(&g4[i]);
          const float d[4] = {a.x - g.x, a.y - g.y, a.z - g.z, a.w - g.w};
          float o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (kind == 0) { acc += fabsf(d[q]); o[q] = d[q] > 0.f ? scale : (d[q] < 0.f ? -scale : 0.f); }
            else { acc += d[q] * d[q]; o[q] = 2.f * d[q] * scale; }
          }
          gv = make_float4(o[0], o[1], o[2], o[3]);
        }
        if (o4) o4[i] = gv;
      }
    } else {
      for (int i = threadIdx.x; i < w; i += 256) {
        float gval = 0.f;
        if (in) {
          const float d = img[base + i] - gt[base + i];
          if (kind == 0) { acc += fabsf(d); gval = d > 0.f ? scale : (d < 0.f ? -scale : 0.f); }
          else { acc += d * d; gval = 2.f * d * scale; }
        }
        if (grad) grad[base + i] = gval;
      }
    }
  }
  acc = nm_wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * scale);
}

extern "C" int nm_pixel_loss(int32_t kind, float weight, int32_t h, int32_t w, int32_t row0, int32_t row1, const float* img,
                             const float* gt, float* loss_out, float* dL_dimg, void* stream) {
  NM_REQUIRE(kind == 0 || kind == 1, "kind must be 0 (l1) or 1 (l2)");
  NM_REQUIRE(h > 0 && w > 0 && img && gt && loss_out, "bad arguments");
  if (row1 <= row0) { row0 = 0; row1 = h; }
  float scale = weight / (3.0f * (float)h * (float)w);
  // every workgroup ends with one atomic on the SAME word (the loss): ~12 ns each, one after the other - 2048 workgroups
  // of one row each spent 25 us of a 30-us kernel queueing there.  About six rows per workgroup, 128..512 workgroups
  // (tools/exp_pixel_loss.py: 256^2 5.7 us, 800^2 ~10, 1080p 19; 12 / 29 / 34 us with 2048)
  int grid = (3 * h + 5) / 6;
  grid = grid < 128 ? (3 * h < 128 ? 3 * h : 128) : (grid > 512 ? 512 : grid);
  const bool vec4 = w % 4 == 0 && ((uintptr_t)img | (uintptr_t)gt | (uintptr_t)dL_dimg) % 16 == 0;
  if (vec4)
    NM_LAUNCH(k_pixel_loss<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, kind, scale, h, w, row0, row1, img, gt,
                       loss_out, dL_dimg);
  else
    NM_LAUNCH(k_pixel_loss<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, kind, scale, h, w, row0, row1, img, gt,
                       loss_out, dL_dimg);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
