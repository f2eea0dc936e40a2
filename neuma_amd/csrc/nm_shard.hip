// Exchange step of the particle-sharded MPM substep (one process per GPU; include/neuma_hip.h "Particle-sharded
// substep").  No reference counterpart: the reference is single-device.
//
// Every rank scatters its own particles into its own block-sparse grid.  A node can receive mass from two ranks only
// inside a 4x4x4-node block that both ranks list as touched, so the data-path collective is an all-reduce over exactly
// those blocks (1 KiB each), not over the grid.  The ranks find them without a host round trip:
//   1. each rank exports {count, ids} of its active list; the caller all-gathers the world lists, so every rank holds
//      the same `gathered` array;
//   2. k_shared_mark counts, per block, how many ranks list it and remembers the first position it appears at;
//   3. a block is selected at that first position iff >= 2 ranks list it; rocPRIM's order-preserving select compacts the
//      selection.  Same input + deterministic rule = the same shared list, in the same order, on all ranks;
//   4. pack copies the rank's values of those blocks into a dense buffer (zeros where the rank does not list the
//      block), the caller all-reduces the buffer, unpack writes the sums back into the blocks the rank lists.
// Blocks that only one rank touches never leave that rank: it alone gathers from them.
#include "nm_common.h"
#include <limits.h>
#include <rocprim/rocprim.hpp>

struct SharedWs {
  unsigned char* sel;  // [world * (1 + cap)]
  int* picked;         // [world * (1 + cap)] compacted selection
  int* npicked;        // [1]
  void* tmp;           // rocPRIM scratch
  size_t tmp_bytes;
};

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

static size_t shared_layout(void* base, int world, int cap, SharedWs* w) {
  const size_t total = (size_t)world * (1 + (size_t)cap);
  size_t tb = 0;
  (void)rocprim::select(nullptr, tb, (int*)nullptr, (unsigned char*)nullptr, (int*)nullptr, (int*)nullptr, total);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return base ? (char*)base + o : (char*)nullptr; };
  char* sel = take(total);
  char* picked = take(total * sizeof(int));
  char* npicked = take(sizeof(int));
  char* tmp = take(tb);
  if (w) { w->sel = (unsigned char*)sel; w->picked = (int*)picked; w->npicked = (int*)npicked; w->tmp = tmp; w->tmp_bytes = tb; }
  return off;
}

extern "C" size_t nm_mpm_shared_workspace(int32_t world, int32_t cap) {
  return (world > 0 && cap > 0) ? shared_layout(nullptr, world, cap, nullptr) : 0;
}

__global__ void k_list_export(const int* __restrict__ list, const int* __restrict__ count, int* __restrict__ out, int cap) {
  const int cnt = *count;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) out[0] = cnt;            // unclamped: the consumers flag cnt > cap as an overflow
  if (i < min(cnt, cap)) out[1 + i] = list[i];
}

__device__ __forceinline__ bool gathered_entry(const int* __restrict__ gathered, int i, int cap, int nblocks, int& b) {
  const int stride = 1 + cap;
  const int r = i / stride, j = i - r * stride - 1;
  if (j < 0) return false;                                   // the rank's count
  if (j >= min(gathered[r * stride], cap)) return false;
  b = gathered[i];
  return (unsigned)b < (unsigned)nblocks;
}

__global__ void k_shared_mark(const int* __restrict__ gathered, int total, int cap, int nblocks, int* __restrict__ cnt,
                              int* __restrict__ pos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int b;
  if (i >= total || !gathered_entry(gathered, i, cap, nblocks, b)) return;
  atomicAdd(&cnt[b], 1);
  atomicMin(&pos[b], i);
}

__global__ void k_shared_flag(const int* __restrict__ gathered, int total, int cap, int nblocks, const int* __restrict__ cnt,
                              const int* __restrict__ pos, unsigned char* __restrict__ sel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int b;
  bool s = false;
  if (gathered_entry(gathered, i, cap, nblocks, b)) s = pos[b] == i && cnt[b] >= 2;
  sel[i] = s ? 1 : 0;
}

__global__ void k_shared_finish(const int* __restrict__ gathered, int total, int world, int cap, int nblocks, int* __restrict__ cnt,
                                int* __restrict__ pos, const int* __restrict__ picked, const int* __restrict__ npicked,
                                int* __restrict__ shared, int cap_shared, const int* __restrict__ flags, int epoch,
                                int* __restrict__ status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int np = *npicked;
  if (i == 0) {
    int bits = np > cap_shared ? 2 : 0;
    for (int r = 0; r < world; ++r)
      if (gathered[r * (1 + cap)] > cap) bits |= 1;
    shared[0] = min(np, cap_shared);
    shared[1] = bits;
    if (bits && status) atomicOr(status, bits);
  }
  if (i < min(np, cap_shared)) {
    const int b = picked[i];
    shared[2 + i] = b;
    shared[2 + cap_shared + i] = flags[b] == epoch ? 1 : 0;
  }
  int b;
  if (i < total && gathered_entry(gathered, i, cap, nblocks, b)) { cnt[b] = 0; pos[b] = INT_MAX; }   // ready for the next call
}

// The same three steps in ONE launch of one 1024-thread workgroup, for the usual case of short lists (a 100k-particle
// body covers ~400 blocks, 1M particles ~3400): mark with global atomics, then an ordered compaction by block scan over
// chunks of 1024 entries, then the reset.  Saves four launches per substep on a path that is launch-latency bound.
#define NM_SHARED_FUSED_MAX 32768
__global__ void __launch_bounds__(1024) k_shared_fused(const int* __restrict__ gathered, int total, int world, int cap, int nblocks,
                                                       int* __restrict__ cnt, int* __restrict__ pos, int* __restrict__ shared,
                                                       int cap_shared, const int* __restrict__ flags, int epoch,
                                                       int* __restrict__ status) {
  __shared__ int wave_sum[16];
  __shared__ int running;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) running = 0;
  for (int i = tid; i < total; i += 1024) {
    int b;
    if (gathered_entry(gathered, i, cap, nblocks, b)) {
      atomicAdd(&cnt[b], 1);
      atomicMin(&pos[b], i);
    }
  }
  __threadfence();
  __syncthreads();
  for (int base = 0; base < total; base += 1024) {
    const int i = base + tid;
    int b = 0;
    bool s = false;
    if (i < total && gathered_entry(gathered, i, cap, nblocks, b))
      s = __hip_atomic_load(&pos[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i &&
          __hip_atomic_load(&cnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 2;
    const unsigned long long m = __ballot(s);
    if (lane == 0) wave_sum[wave] = __popcll(m);
    __syncthreads();
    int before = running, all = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wave) before += wave_sum[w];
      all += wave_sum[w];
    }
    const int idx = before + __popcll(m & ((1ull << lane) - 1ull));
    if (s && idx < cap_shared) {
      shared[2 + idx] = b;
      shared[2 + cap_shared + idx] = flags[b] == epoch ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) running += all;
    __syncthreads();
  }
  for (int i = tid; i < total; i += 1024) {
    int b;
    if (gathered_entry(gathered, i, cap, nblocks, b)) { cnt[b] = 0; pos[b] = INT_MAX; }
  }
  if (tid == 0) {
    const int np = running;
    int bits = np > cap_shared ? 2 : 0;
    for (int r = 0; r < world; ++r)
      if (gathered[r * (1 + cap)] > cap) bits |= 1;
    shared[0] = min(np, cap_shared);
    shared[1] = bits;
    if (bits && status) atomicOr(status, bits);
  }
}

extern "C" int nm_mpm_active_list(nm_mpm* h, int32_t* out, int32_t cap, void* stream) {
  NM_REQUIRE(h && out, "null handle / output");
  NM_REQUIRE(cap > 0, "list capacity must be positive");
  nm_mpm_view v = nm_mpm_get_view(h);
  NM_LAUNCH(k_list_export, dim3(nm_div_up(cap, 256)), dim3(256), 0, (hipStream_t)stream, v.list, v.count, out, cap);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" int nm_mpm_shared_blocks(nm_mpm* h, const int32_t* gathered, int32_t world, int32_t cap, int32_t* shared,
                                    int32_t cap_shared, int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
  NM_REQUIRE(h && gathered && shared && workspace, "null handle / buffer");
  NM_REQUIRE(world >= 1 && cap > 0 && cap_shared > 0, "world, cap and cap_shared must be positive");
  nm_mpm_view v = nm_mpm_get_view(h);
  SharedWs w;
  size_t need = shared_layout(workspace, world, cap, &w);
  NM_REQUIRE(workspace_bytes >= need, "shared-block workspace too small (nm_mpm_shared_workspace)");
  int *cnt = nullptr, *pos = nullptr;   // per-block counters, owned by the handle, clean between calls
  int rc = nm_mpm_shared_counters(h, &cnt, &pos);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int total = world * (1 + cap);
  if (total <= NM_SHARED_FUSED_MAX) {
    NM_LAUNCH(k_shared_fused, dim3(1), dim3(1024), 0, s, gathered, total, world, cap, v.nblocks, cnt, pos, shared, cap_shared,
                       v.flags, v.epoch, status);
    NM_LAUNCH_CHECK();
    return NM_OK;
  }
  const dim3 grid(nm_div_up(total, 256)), block(256);
  NM_LAUNCH(k_shared_mark, grid, block, 0, s, gathered, total, cap, v.nblocks, cnt, pos);
  NM_LAUNCH_CHECK();
  NM_LAUNCH(k_shared_flag, grid, block, 0, s, gathered, total, cap, v.nblocks, cnt, pos, w.sel);
  NM_LAUNCH_CHECK();
  size_t tb = w.tmp_bytes;
  NM_HIP_CHECK(rocprim::select(w.tmp, tb, gathered, w.sel, w.picked, w.npicked, (size_t)total, s));
  NM_LAUNCH(k_shared_finish, grid, block, 0, s, gathered, total, world, cap, v.nblocks, cnt, pos, w.picked, w.npicked,
                     shared, cap_shared, v.flags, v.epoch, status);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

// one wave per slot of the exchange buffer
__global__ void __launch_bounds__(256) k_blocks_pack(const float4* __restrict__ src, const int* __restrict__ shared, int cap_shared,
                                                     float4* __restrict__ buf) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cnt = shared[0];
  for (int i = blockIdx.x * 4 + wave; i < cap_shared; i += gridDim.x * 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < cnt && shared[2 + cap_shared + i]) v = src[(shared[2 + i] << 6) + lane];
    buf[(i << 6) + lane] = v;
  }
}

__global__ void __launch_bounds__(256) k_blocks_unpack(float4* __restrict__ dst, const int* __restrict__ shared, int cap_shared,
                                                       const float4* __restrict__ buf) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cnt = shared[0];
  for (int i = blockIdx.x * 4 + wave; i < cnt; i += gridDim.x * 4)
    if (shared[2 + cap_shared + i]) dst[(shared[2 + i] << 6) + lane] = buf[(i << 6) + lane];
}

static const int kXchgGrid = 512;

extern "C" int nm_mpm_blocks_pack(nm_mpm* h, int32_t which, const int32_t* shared, int32_t cap_shared, float* buf, void* stream) {
  NM_REQUIRE(h && shared && buf, "null handle / buffer");
  NM_REQUIRE(which == 0 || which == 1, "which: 0 = {mv, m}, 1 = grid adjoint");
  NM_REQUIRE(cap_shared > 0, "cap_shared must be positive");
  nm_mpm_view v = nm_mpm_get_view(h);
  NM_LAUNCH(k_blocks_pack, dim3(min(kXchgGrid, nm_div_up(cap_shared, 4))), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)(which ? v.gg : v.gm), shared, cap_shared, (float4*)buf);
  NM_LAUNCH_CHECK();
  return NM_OK;
}

extern "C" int nm_mpm_blocks_unpack(nm_mpm* h, int32_t which, const int32_t* shared, int32_t cap_shared, const float* buf,
                                    void* stream) {
  NM_REQUIRE(h && shared && buf, "null handle / buffer");
  NM_REQUIRE(which == 0 || which == 1, "which: 0 = {mv, m}, 1 = grid adjoint");
  NM_REQUIRE(cap_shared > 0, "cap_shared must be positive");
  nm_mpm_view v = nm_mpm_get_view(h);
  NM_LAUNCH(k_blocks_unpack, dim3(min(kXchgGrid, nm_div_up(cap_shared, 4))), dim3(256), 0, (hipStream_t)stream,
                     which ? v.gg : v.gm, shared, cap_shared, (const float4*)buf);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
