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

static const int kXchgGrid = 512;

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


// ---------------------------------------------------------------- frame-level negotiation (sharded roll-out in the library)
// The per-substep exchange above negotiates the shared-block list every substep (all-gather + selection + all-reduce).  A
// body moves a fraction of a block during the S substeps of a frame, so the roll-out negotiates ONCE per frame instead, with
// a superset: every rank lists the 27-neighbourhood (in blocks) of what it touches at the first substep - everything one of
// its particles can reach while it moves less than a block (4 grid cells) - and a block is exchanged if it lies in the
// neighbourhoods of two ranks.  Two ranks can only both touch a block that lies in both neighbourhoods, so the superset is
// complete as long as no rank leaves its own neighbourhood; every substep checks exactly that (status bit 8), so a wrong sum
// never goes unnoticed: the frame driver waits for the roll-out's status word before it hands the frame's gradients to its
// caller (GridExchange.check(wait="watched")) and raises - nothing re-runs the frame by itself.  A substep then costs one pack launch and one
// all-reduce per direction; the unpack is part of k_grid_op / k_grid_op_bwd (slot[] lookup).
__global__ void __launch_bounds__(256) k_dilate_export(const int* __restrict__ list, const int* __restrict__ count, int nb,
                                                       int* __restrict__ dil, int tag, int* __restrict__ out, int cap) {
  const int cnt = *count;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt * 27; i += gridDim.x * blockDim.x) {
    const int b = list[i / 27], o = i % 27;
    const int bk = b % nb, bj = (b / nb) % nb, bi = b / (nb * nb);
    const int ni = bi + o / 9 - 1, nj = bj + (o / 3) % 3 - 1, nk = bk + o % 3 - 1;
    if ((unsigned)ni >= (unsigned)nb || (unsigned)nj >= (unsigned)nb || (unsigned)nk >= (unsigned)nb) continue;
    const int q = (ni * nb + nj) * nb + nk;
    if (dil[q] != tag && atomicExch(&dil[q], tag) != tag) {
      const int pos = atomicAdd(&out[0], 1);       // unclamped count: the consumers flag > cap as an overflow
      if (pos < cap) out[1 + pos] = q;
    }
  }
}
__global__ void k_xslots(const int* __restrict__ shared, int cap_shared, int* __restrict__ slot, int assign) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < min(shared[0], cap_shared)) slot[shared[2 + i]] = assign ? i : -1;
}
// one wave per slot: this rank's {mv, m} of the block if it holds the block in this substep, zeros otherwise; the answer is
// remembered per slot for the reverse sweep (mine).  The same launch checks that the rank stayed inside its neighbourhood.
__global__ void __launch_bounds__(256) k_xpack_fwd(const float4* __restrict__ gm, const int* __restrict__ shared, int cap_shared,
                                                   const int* __restrict__ flags, int epoch, float4* __restrict__ buf,
                                                   unsigned char* __restrict__ mine, const int* __restrict__ list,
                                                   const int* __restrict__ count, const int* __restrict__ dil, int tag,
                                                   int* __restrict__ status) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cnt = min(shared[0], cap_shared);
  for (int i = blockIdx.x * 4 + wave; i < cap_shared; i += gridDim.x * 4) {
    bool m = false;
    int b = 0;
    if (i < cnt) { b = shared[2 + i]; m = flags[b] == epoch; }
    buf[(i << 6) + lane] = m ? gm[(b << 6) + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane == 0) mine[i] = m ? 1 : 0;
  }
  const int nl = *count;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += gridDim.x * blockDim.x)
    if (dil[list[i]] != tag) atomicOr(status, 8);
}
__global__ void __launch_bounds__(256) k_xpack_bwd(const float4* __restrict__ gg, const int* __restrict__ shared, int cap_shared,
                                                   const unsigned char* __restrict__ mine, float4* __restrict__ buf) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = blockIdx.x * 4 + wave; i < cap_shared; i += gridDim.x * 4)
    buf[(i << 6) + lane] = mine[i] ? gg[(shared[2 + i] << 6) + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// {count, ids} of the 27-neighbourhood of the blocks the rank's current grid holds (out[0] must be zero on entry; the count
// is not clamped).  Starts a new negotiation: the neighbourhood is what the following nm_shard_pack_fwd calls check against.
extern "C" int nm_mpm_dilated_list(nm_mpm* h, int32_t* out, int32_t cap, void* stream) {
  NM_REQUIRE(h && out, "null handle / output");
  NM_REQUIRE(cap > 0, "list capacity must be positive");
  nm_mpm_view v = nm_mpm_get_view(h);
  int *dil = nullptr, tag = 0;
  int rc = nm_mpm_xchg_arrays(h, nullptr, &dil, &tag);
  if (rc) return rc;
  NM_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(int32_t), (hipStream_t)stream));
  NM_LAUNCH(k_dilate_export, dim3(64), dim3(256), 0, (hipStream_t)stream, v.list, v.count, nm_mpm_grid_dims(h), dil, tag, out, cap);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
int nm_shard_slots(nm_mpm* h, const int32_t* shared, int32_t cap_shared, int assign, void* stream) {
  int* slot = nullptr;
  int rc = nm_mpm_xchg_arrays(h, &slot, nullptr, nullptr);
  if (rc) return rc;
  NM_LAUNCH(k_xslots, dim3(nm_div_up(cap_shared, 256)), dim3(256), 0, (hipStream_t)stream, shared, cap_shared, slot, assign);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
int nm_shard_pack_fwd(nm_mpm* h, const int32_t* shared, int32_t cap_shared, float* buf, unsigned char* mine, int32_t* status,
                      void* stream) {
  nm_mpm_view v = nm_mpm_get_view(h);
  int* dil = nullptr;
  int rc = nm_mpm_xchg_arrays(h, nullptr, &dil, nullptr);
  if (rc) return rc;
  NM_LAUNCH(k_xpack_fwd, dim3(min(kXchgGrid, nm_div_up(cap_shared, 4))), dim3(256), 0, (hipStream_t)stream, (const float4*)v.gm,
                     shared, cap_shared, (const int*)v.flags, v.epoch, (float4*)buf, mine, (const int*)v.list, (const int*)v.count,
                     (const int*)dil, nm_mpm_dil_tag(h), status);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
int nm_shard_pack_bwd(nm_mpm* h, const int32_t* shared, int32_t cap_shared, float* buf, const unsigned char* mine, void* stream) {
  nm_mpm_view v = nm_mpm_get_view(h);
  NM_LAUNCH(k_xpack_bwd, dim3(min(kXchgGrid, nm_div_up(cap_shared, 4))), dim3(256), 0, (hipStream_t)stream, (const float4*)v.gg,
                     shared, cap_shared, mine, (float4*)buf);
  NM_LAUNCH_CHECK();
  return NM_OK;
}


// ---------------------------------------------------------------- neighbour-only exchange (round 5; nm_comm.exchange_peers_f32)
// A shared block belongs to the 2-3 ranks whose particle ranges meet there.  With the frame's exchange buffer laid out by
// slot - the same on every rank, zeros where a rank does not hold a block - a rank only has to swap buffers with the ranks it
// shares at least one block with and add them up; ranks further away would contribute zeros to every slot it reads.
__global__ void k_peer_adj(const int* __restrict__ gathered, int total, int cap, int nblocks, const int* __restrict__ dil, int tag,
                           int rank, int* __restrict__ adj, unsigned peers, int* __restrict__ status) {
  // adj != NULL: adj[q] = 1 for every rank q that lists a block of this rank's neighbourhood;
  // status != NULL: bit 16 if such a rank is not in `peers`
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int b;
  if (i >= total || !gathered_entry(gathered, i, cap, nblocks, b)) return;
  const int q = i / (1 + cap);
  if (q == rank || dil[b] != tag) return;
  if (adj) adj[q] = 1;
  if (status && q < 32 && !((peers >> q) & 1u)) atomicOr(status, 16);
}
// buf[i] = sum over {this rank} + peers, in ascending rank order, of the ranks' buffers (the peers' lie behind one another in recv)
__global__ void __launch_bounds__(256) k_peer_sum(float4* __restrict__ buf, const float4* __restrict__ recv, size_t n4, unsigned peers,
                                                  int rank, int world) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (int q = 0; q < world && q < 32; ++q) {
      float4 v;
      if (q == rank) v = buf[i];
      else if ((peers >> q) & 1u) { v = recv[(size_t)k * n4 + i]; ++k; }
      else continue;
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    buf[i] = acc;
  }
}

extern "C" int nm_mpm_peer_ranks(nm_mpm* h, const int32_t* gathered, int32_t world, int32_t cap, int32_t rank, int32_t* adj,
                                 void* stream) {
  NM_REQUIRE(h && gathered && adj, "null handle / buffer");
  NM_REQUIRE(world >= 1 && cap > 0 && rank >= 0 && rank < world, "bad world / cap / rank");
  nm_mpm_view v = nm_mpm_get_view(h);
  int* dil = nullptr;
  int rc = nm_mpm_xchg_arrays(h, nullptr, &dil, nullptr);
  if (rc) return rc;
  NM_HIP_CHECK(hipMemsetAsync(adj, 0, (size_t)world * sizeof(int32_t), (hipStream_t)stream));
  const int total = world * (1 + cap);
  NM_LAUNCH(k_peer_adj, dim3(nm_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, gathered, total, cap, v.nblocks, (const int*)dil,
                     nm_mpm_dil_tag(h), rank, adj, 0u, (int*)nullptr);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
int nm_shard_peer_check(nm_mpm* h, const int32_t* gathered, int32_t world, int32_t cap, int32_t rank, uint32_t peers, int32_t* status,
                        void* stream) {
  nm_mpm_view v = nm_mpm_get_view(h);
  int* dil = nullptr;
  int rc = nm_mpm_xchg_arrays(h, nullptr, &dil, nullptr);
  if (rc) return rc;
  const int total = world * (1 + cap);
  NM_LAUNCH(k_peer_adj, dim3(nm_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, gathered, total, cap, v.nblocks, (const int*)dil,
                     nm_mpm_dil_tag(h), rank, (int*)nullptr, peers, status);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
int nm_shard_peer_sum(float* buf, const float* recv, size_t count, uint32_t peers, int32_t rank, int32_t world, void* stream) {
  const size_t n4 = count / 4;
  if (n4 == 0) return NM_OK;
  NM_LAUNCH(k_peer_sum, dim3((unsigned)min((size_t)1024, (n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4*)buf,
                     (const float4*)recv, n4, peers, rank, world);
  NM_LAUNCH_CHECK();
  return NM_OK;
}
