// Fused roll-out of S substeps with checkpointed BPTT.
// Operator order follows /root/reference/experiments/finetune.py:360-364 (stress = E(F); sim; F = P(F)),
// the reverse sweep follows interface.py:41-76 + mpm.py:299-319 per step.  Only (x,v,C,F,stress) per substep is
// kept (132 B/particle), plus - optionally - each substep's touched grid blocks (the grid cache, see nm_mpm_forward_ex);
// the trial deformation gradient and every MLP activation are recomputed.
#include "nm_common.h"
#include "nm_grid.h"

#define NM_WTOT_ (64 * 13 + 64 * 64 + 9 * 64)

static inline size_t al256r(size_t x) { return (x + 255) & ~(size_t)255; }

struct RolloutWs {
  float* stress;   // N*9
  float* ftrial;   // N*9   trial F out of g2p (input of plasticity)
  float* ga;       // N*24  gradient ping
  float* gb;       // N*24  gradient pong (+ stress grad N*9 appended)
  float* gS;       // N*9
  float* gFtr;     // N*9
  float* part_e;   // per-workgroup weight-gradient partials, summed over substeps, one buffer per net
  float* part_p;
  float* perm_e;   // weights in MFMA operand order (nm_material_prepare), shared by all substeps
  float* perm_p;
  size_t total;
};

static RolloutWs carve_ws(void* base, int n) {
  RolloutWs w;
  char* p = (char*)base;
  size_t o = 0, N = (size_t)(n > 0 ? n : 1);
  auto take = [&](size_t floats) { float* r = (float*)(p + o); o += al256r(floats * sizeof(float)); return r; };
  w.stress = take(N * 9);
  w.ftrial = take(N * 9);
  w.ga = take(N * 24);
  w.gb = take(N * 24);
  w.gS = take(N * 9);
  w.gFtr = take(N * 9);
  const size_t part = nm_material_bwd_workspace(n) / sizeof(float);
  w.part_e = take(part);
  w.part_p = take(part);
  w.perm_e = take(nm_material_prepared_floats());
  w.perm_p = take(nm_material_prepared_floats());
  w.total = o;
  return w;
}

extern "C" size_t nm_rollout_workspace(int32_t n, int32_t substeps) {
  (void)substeps;
  return carve_ws(nullptr, n).total;
}

#define NM_REC 33  // floats per particle per checkpoint record: x3 v3 C9 F9 stress9
static inline nm_particles rec(float* base, int n, int t) {
  float* r = base + (size_t)t * NM_REC * n;
  nm_particles p;
  p.x = r;
  p.v = r + 3 * (size_t)n;
  p.C = r + 6 * (size_t)n;
  p.F = r + 15 * (size_t)n;
  p.stress = r + 24 * (size_t)n;   // stress computed FROM this record's F (input of the step that leaves it)
  return p;
}

extern "C" size_t nm_rollout_gridcache_bytes(int32_t substeps, int32_t grid_cache_blocks) {
  if (substeps < 1 || grid_cache_blocks < 1) return 0;
  return (size_t)substeps * nm_mpm_gridcache_bytes(grid_cache_blocks);
}
static inline void* grid_rec(const void* gridcache, const nm_rollout_cfg* cfg, int t) {
  if (!gridcache || cfg->grid_cache_blocks < 1) return nullptr;
  return (char*)const_cast<void*>(gridcache) + (size_t)t * nm_mpm_gridcache_bytes(cfg->grid_cache_blocks);
}

// SVD cache (cfg->svd_cache, optional): U | sigma | V of both nets' inputs of every substep, 2 x 21 x N floats per substep
extern "C" size_t nm_rollout_svdcache_bytes(int32_t n, int32_t substeps) {
  if (n < 1 || substeps < 1) return 0;
  return (size_t)substeps * 2 * 21 * (size_t)n * sizeof(float);
}
// activation cache (cfg->act_cache, optional): the MLPs' hidden activations and GELU derivatives of every substep, in the
// kernels' own accumulator layout (17 x 16 B per lane and 16-particle tile = 1088 B per particle, substep and net)
extern "C" size_t nm_rollout_actcache_bytes(int32_t n, int32_t substeps) {
  if (n < 1 || substeps < 1) return 0;
  return (size_t)substeps * 2 * nm_material_act_floats(n) * sizeof(float);
}
static inline float* act_rec(const nm_rollout_cfg* cfg, int n, int t, int net) {
  if (!cfg->act_cache) return nullptr;
  return (float*)cfg->act_cache + ((size_t)t * 2 + net) * nm_material_act_floats(n);
}
static inline float* svd_rec(const nm_rollout_cfg* cfg, int n, int t, int net) {
  if (!cfg->svd_cache) return nullptr;
  return (float*)cfg->svd_cache + ((size_t)t * 2 + net) * 21 * (size_t)n;
}

// the S record headers, one word each at the records' stride, written by one wave straight into the caller's pinned words
__global__ void k_cache_status_out(const char* __restrict__ gridcache, size_t stride, int substeps, int32_t* __restrict__ host_words) {
  for (int t = threadIdx.x; t < substeps; t += blockDim.x) host_words[t] = *reinterpret_cast<const int32_t*>(gridcache + (size_t)t * stride);
}
extern "C" int nm_rollout_cache_status(const void* gridcache, const nm_rollout_cfg* cfg, int32_t* status_host, void* stream) {
  NM_REQUIRE(gridcache && cfg && status_host, "null pointer");
  NM_REQUIRE(cfg->substeps >= 1 && cfg->grid_cache_blocks >= 1, "no grid cache configured");
  // pinned memory the device can address (hipHostMalloc'ed: torch's pin_memory) is written directly - a strided device-to-host
  // copy is a blit of its own and a 20 us hole in the stream; anything else gets the copy
  void* dp = nullptr;
  static const bool direct = !(getenv("NEUMA_STATUS_DIRECT") && atoi(getenv("NEUMA_STATUS_DIRECT")) == 0);
  if (direct && hipHostGetDevicePointer(&dp, status_host, 0) == hipSuccess && dp) {
    NM_LAUNCH(k_cache_status_out, dim3(1), dim3(64), 0, (hipStream_t)stream, (const char*)gridcache,
              nm_mpm_gridcache_bytes(cfg->grid_cache_blocks), (int)cfg->substeps, (int32_t*)dp);
    NM_LAUNCH_CHECK();
    return NM_OK;
  }
  (void)hipGetLastError();
  NM_HIP_CHECK(hipMemcpy2DAsync(status_host, sizeof(int32_t), gridcache, nm_mpm_gridcache_bytes(cfg->grid_cache_blocks), sizeof(int32_t),
                                (size_t)cfg->substeps, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return NM_OK;
}

// forward sweep: plasticity(t) and elasticity(t+1) in one launch (default) or one launch per net (A/B measurements, tests)
static int g_forward_pair = 1;
// forward sweep: no k_grid_op in front of a pair launch (velocities formed inside its g2p, GridPrologue mode 3).  OFF by default:
// built, correct, measured 22 us per substep SLOWER at the metric size - kept as a switch with its test
// Cache buffers whose last-substep plasticity records the forward sweep did NOT write (nm_rollout_cfg.last_gF_zero given to
// nm_rollout_forward): a reverse sweep over the same buffers without the flag would read a previous frame's records from the
// pooled buffer and return wrong gradients with no error.  Host-side bookkeeping keyed by the cache pointers (a forward sweep
// that writes the records takes the entry out again); the reverse sweep refuses the mismatch.
#include <mutex>
#include <unordered_set>
static std::mutex g_skip_mu;
static std::unordered_set<const void*> g_skipped_last;
static void note_last_records(const nm_rollout_cfg* cfg, bool skipped) {
  std::lock_guard<std::mutex> lk(g_skip_mu);
  for (const void* key : {(const void*)cfg->svd_cache, (const void*)cfg->act_cache}) {
    if (!key) continue;
    if (skipped) g_skipped_last.insert(key); else g_skipped_last.erase(key);
  }
}
static bool last_records_missing(const nm_rollout_cfg* cfg) {
  std::lock_guard<std::mutex> lk(g_skip_mu);
  return (cfg->svd_cache && g_skipped_last.count(cfg->svd_cache)) || (cfg->act_cache && g_skipped_last.count(cfg->act_cache));
}

extern "C" int nm_rollout_set_forward_pair(int32_t on) {
  g_forward_pair = on ? 1 : 0;
  return NM_OK;
}

extern "C" int nm_rollout_forward(nm_mpm* h, int32_t n, const nm_rollout_cfg* cfg, const nm_statics* st, const nm_mlp* we,
                                  const nm_mlp* wp, float* states, void* gridcache, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  NM_REQUIRE(h && cfg && st && we && wp && states, "null pointer");
  NM_REQUIRE(n >= 0 && cfg->substeps >= 1, "bad sizes");
  if (n == 0) return NM_OK;
  RolloutWs w = carve_ws(workspace, n);
  if (!workspace || workspace_bytes < w.total) {
    nm_set_error("rollout workspace too small: need %zu got %zu", w.total, workspace_bytes);
    return NM_ERR_WORKSPACE;
  }
  int rc = nm_material_prepare2(we, w.perm_e, wp, w.perm_p, stream);
  if (rc) return rc;
  for (int t = 0; t < cfg->substeps; ++t) {
    nm_particles cur = rec(states, n, t), nxt = rec(states, n, t + 1);
    if (t == 0 || !g_forward_pair) {
      // the elasticity kernel also clears the grid for the substep that follows it (GridPrologue mode 1, nm_grid.h)
      GridPrologue pro;
      rc = nm_mpm_prologue_forward(h, &pro);
      if (rc) return rc;
      rc = nm_material_fwd_launch(n, NM_ELASTICITY, 0.f, cur.F, we, w.perm_e, cur.stress, &pro, nullptr, stream, svd_rec(cfg, n, t, 0),
                                  act_rec(cfg, n, t, 0));  // finetune.py:362
      if (rc) return rc;
    }
    // p2g + grid update here; the substep's g2p runs inside the plasticity kernel, which consumes its trial F from
    // registers (the reverse sweep recomputes the trial F from the checkpointed C', so it is never stored).
    // (round 5 also built a variant without the grid-update launch - the pair kernel's g2p forming the node velocities from
    //  {mv, m}, its prologue workgroups waiting for the gathers before they clear the grid: 56 -> 84 us for the 5 us saved, and a
    //  wait between workgroups of one launch; removed in round 6)
    rc = nm_mpm_forward_prepared_nog2p(h, n, st, &cur, grid_rec(gridcache, cfg, t), cfg->grid_cache_blocks, stream);  // finetune.py:363
    if (rc) return rc;
    G2pFuse g2p;
    rc = nm_mpm_g2p_fuse(h, st, &cur, &nxt, &g2p);
    if (rc) return rc;
    if (g_forward_pair && t + 1 < cfg->substeps) {
      // plasticity of this substep and elasticity of the next in one launch (F_{t+1} goes from one net to the other in
      // registers), which also carries the grid clear of substep t+1 - with the velocities left in place, because this very
      // launch gathers them (finetune.py:364 -> :362 of the next iteration)
      GridPrologue pro;
      rc = nm_mpm_prologue_forward(h, &pro, true);
      if (rc) return rc;
      rc = nm_material_fwd_pair_launch(n, cfg->plasticity_alpha, w.perm_p, w.perm_e, nxt.F, nxt.stress, &pro, &g2p, stream,
                                       svd_rec(cfg, n, t, 1), svd_rec(cfg, n, t + 1, 0), act_rec(cfg, n, t, 1), act_rec(cfg, n, t + 1, 0));
    } else {
      // (last_gF_zero: the reverse sweep will not visit the last substep's plasticity adjoint - nobody reads its SVD / activation
      //  records, 74 MB at the metric size; the pair launches' records are untouched by this)
      const bool unread = cfg->last_gF_zero != 0 && cfg->substeps >= 2 && t == cfg->substeps - 1;
      if (t == cfg->substeps - 1) note_last_records(cfg, unread);
      rc = nm_material_fwd_launch(n, NM_PLASTICITY, cfg->plasticity_alpha, nullptr, wp, w.perm_p, nxt.F, nullptr, &g2p, stream,
                                  unread ? nullptr : svd_rec(cfg, n, t, 1), unread ? nullptr : act_rec(cfg, n, t, 1));  // finetune.py:364
    }
    if (rc) return rc;
  }
  return NM_OK;
}

extern "C" int nm_rollout_backward(nm_mpm* h, int32_t n, const nm_rollout_cfg* cfg, const nm_statics* st, const nm_mlp* we,
                                   const nm_mlp* wp, const float* states, const void* gridcache, const float* gstate_last,
                                   float* gstate_first, float* gw_e, float* gw_p, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  NM_REQUIRE(h && cfg && st && we && wp && states && gstate_last && gstate_first && gw_e && gw_p, "null pointer");
  NM_REQUIRE(n >= 0 && cfg->substeps >= 1, "bad sizes");
  hipStream_t s = (hipStream_t)stream;
  (void)s;
  if (n == 0) {
    NM_HIP_CHECK(hipMemsetAsync(gw_e, 0, NM_WTOT_ * sizeof(float), s));
    NM_HIP_CHECK(hipMemsetAsync(gw_p, 0, NM_WTOT_ * sizeof(float), s));
    return NM_OK;
  }
  RolloutWs w = carve_ws(workspace, n);
  if (!workspace || workspace_bytes < w.total) {
    nm_set_error("rollout workspace too small: need %zu got %zu", w.total, workspace_bytes);
    return NM_ERR_WORKSPACE;
  }
  const size_t N = (size_t)n;
  float* states_m = const_cast<float*>(states);
  const float* gin = gstate_last;
  int rc = NM_OK;
  if (!cfg->weights_prepared) rc = nm_material_prepare2(we, w.perm_e, wp, w.perm_p, stream);
  if (rc) return rc;
  const bool verified = cfg->cache_verified != 0 && gridcache != nullptr && cfg->grid_cache_blocks > 0;
  const int polar = cfg->svd_adjoint == NM_SVD_ADJOINT_POLAR ? 1 : 0;
  const float dt = nm_mpm_get_dt(h);
  bool restored = false;   // the grid of the substep about to be visited was restored by the previous launch's prologue
  // (with one substep there is no pair launch that could write the plasticity partials in its place)
  const bool skip_last = cfg->last_gF_zero != 0 && cfg->substeps >= 2;
  if (!skip_last && cfg->substeps >= 2 && last_records_missing(cfg)) {
    nm_set_error("nm_rollout_backward without last_gF_zero on caches whose forward sweep ran with it: the last substep's "
                 "plasticity SVD / activation records were not written (give the flag to both calls or to neither)");
    return NM_ERR_INVALID;
  }
  for (int t = cfg->substeps - 1; t >= 0; --t) {
    nm_particles cur = rec(states_m, n, t), nxt = rec(states_m, n, t + 1);
    float* gout = (t == 0) ? gstate_first : ((gin == w.ga) ? w.gb : w.ga);
    const int wmode = (t == cfg->substeps - 1) ? 1 : 2;   // first visit writes the partials, later ones add
    if (t == cfg->substeps - 1) {
      // plasticity backward on the trial F of the last substep (recomputed in-kernel from the checkpoints):
      // dL/dF_{t+1} -> dL/dFtrial.  For every earlier substep it rides in the pair launch at the end of this loop body.
      if (skip_last) {
        // dL/dF of the last record is zero by the caller's word (a frame whose loss sees positions only): the adjoint of the
        // last plasticity step is zero too, its weight gradients as well - a 46 us launch that computed zeros every frame
        NM_HIP_CHECK(hipMemsetAsync(w.gFtr, 0, 9 * N * sizeof(float), s));
      } else {
        rc = nm_material_bwd_launch(n, NM_PLASTICITY, cfg->plasticity_alpha, cur.F, wp, w.perm_p, gin + 15 * N, w.gFtr, w.part_p, wmode,
                                    nxt.C, st->enabled, dt, polar ? 2 : 0, nullptr, stream, svd_rec(cfg, n, t, 1), act_rec(cfg, n, t, 1));
        if (rc) return rc;
      }
    }
    // sim backward (stress of this step was checkpointed by the forward pass).  Verified sweep: from the second substep
    // on the grid has been restored by the prologue of the preceding constitutive launch (GridPrologue mode 2; the block
    // flags it relies on were set by the previous substep's k_grid_op_bwd); otherwise the stand-alone launches do it.
    nm_particles gn, gc;
    gn.x = const_cast<float*>(gin); gn.v = const_cast<float*>(gin) + 3 * N; gn.C = const_cast<float*>(gin) + 6 * N;
    gn.F = w.gFtr; gn.stress = nullptr;
    gc.x = gout; gc.v = gout + 3 * N; gc.C = gout + 6 * N; gc.F = gout + 15 * N; gc.stress = w.gS;
    rc = nm_mpm_backward_cached(h, n, st, &cur, &nxt, &gn, &gc, grid_rec(gridcache, cfg, t), cfg->grid_cache_blocks,
                                cfg->cache_verified != 0, restored, (verified && t > 0) ? grid_rec(gridcache, cfg, t - 1) : nullptr,
                                stream);
    if (rc) return rc;
    restored = false;
    if (t == 0) {
      // elasticity backward: dL/dstress -> dL/dF (added to the sim's dL/dF)
      rc = nm_material_bwd_launch(n, NM_ELASTICITY, 0.f, cur.F, we, w.perm_e, w.gS, gc.F, w.part_e, wmode, nullptr, nullptr, 0.f,
                                  1 | (polar ? 2 : 0), nullptr, stream, svd_rec(cfg, n, t, 0), act_rec(cfg, n, t, 0));
      if (rc) return rc;
    } else {
      // elasticity backward of this substep and plasticity backward of the previous one (its input dL/dF_t is exactly
      // what the elasticity part leaves in gc.F) in one launch, which also carries the grid prologue of substep t-1
      nm_particles prev = rec(states_m, n, t - 1);
      GridPrologue pro;
      if (verified) {
        rc = nm_mpm_prologue_backward(h, grid_rec(gridcache, cfg, t - 1), cfg->grid_cache_blocks, &pro);
        if (rc) return rc;
        restored = true;
      }
      // (the plasticity partials: added to - written, if the skipped launch of the last substep has not done so)
      rc = nm_material_bwd_pair_launch(n, cur.F, we, w.perm_e, w.gS, gc.F, w.part_e, wmode, cfg->plasticity_alpha, prev.F, wp,
                                       w.perm_p, w.gFtr, w.part_p, (skip_last && t == cfg->substeps - 1) ? 1 : 2, cur.C, st->enabled, dt,
                                       polar, verified ? &pro : nullptr, stream,
                                       svd_rec(cfg, n, t, 0), svd_rec(cfg, n, t - 1, 1), act_rec(cfg, n, t, 0), act_rec(cfg, n, t - 1, 1));
      if (rc) return rc;
    }
    gin = gout;
  }
  // one deterministic reduction per net for the whole roll-out
  return nm_material_wgrad_reduce2(w.part_e, w.part_p, n, gw_e, gw_p, stream);
}

// ---------------------------------------------------------------- particle-sharded roll-out
// The same S-substep node for ONE rank's share of the particles: the substep's MPM part is cut at the two points where the
// grid blocks several ranks touch have to be summed (include/neuma_hip.h, "Particle-sharded substep"), and the loop -
// launches and collectives alike - runs here, in the library, on the caller's stream.  The collectives are the caller's
// (nm_comm: torch.distributed over RCCL in production, gloo in the tests); the library owns everything between them.
// The status word of a sharded roll-out is made THE SAME ON EVERY RANK before anybody reads it: bits 1 and 2 come from the
// gathered lists and already are, but bit 8 (a block left the neighbourhood its rank announced) and bit 4 (a grid cache
// record overflowed) are raised by the rank it happens to - and a rank that alone raises, resets its capacities and re-runs
// the frame enters collectives the others are not in.  One 4-float all-reduce at the end of the forward sweep (a flag per
// bit, summed) costs one small collective per roll-out; every rank then reports, and recovers, together.
__global__ void k_status_to_flags(const int32_t* __restrict__ status, float* __restrict__ flags) {
  if (threadIdx.x < 8) flags[threadIdx.x] = ((status[0] >> threadIdx.x) & 1) ? 1.f : 0.f;
}
__global__ void k_flags_to_status(const float* __restrict__ flags, int32_t* __restrict__ status) {
  if (threadIdx.x == 0) {
    int bits = status[0];
    for (int b = 0; b < 8; ++b) bits |= flags[b] > 0.f ? (1 << b) : 0;
    status[0] = bits;
  }
}

struct ShardWs {
  int32_t* status;   // capacity / neighbourhood bits of the roll-out (see nm_rollout_shard_status); [16..23] as floats: the flags
  int32_t* mine;     // this rank's neighbourhood list: [0] count, [1..cap] block ids
  int32_t* gathered; // world x (1 + cap)
  int32_t* shared;   // the frame's exchange list: 2 + 2 * cap_shared (kept for the reverse sweep)
  unsigned char* held;   // per substep and slot: the rank held the block in that substep (kept for the reverse sweep)
  float* buf;        // cap_shared x 64 float4: the all-reduced payload
  float* recv;       // (world - 1) x the same: the peers' buffers of the neighbour-only exchange
  void* sws;         // workspace of nm_mpm_shared_blocks
  size_t sws_bytes, held_stride, total;
};
static ShardWs carve_shard(void* base, int world, int cap, int cap_shared, int substeps) {
  ShardWs w;
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t bytes) { void* r = p + o; o += al256r(bytes); return r; };
  w.status = (int32_t*)take(256);
  w.mine = (int32_t*)take((size_t)(1 + cap) * sizeof(int32_t));
  w.gathered = (int32_t*)take((size_t)world * (1 + cap) * sizeof(int32_t));
  w.shared = (int32_t*)take((size_t)(2 + 2 * cap_shared) * sizeof(int32_t));
  w.held_stride = al256r((size_t)cap_shared);
  w.held = (unsigned char*)take((size_t)(substeps > 0 ? substeps : 1) * w.held_stride);
  w.buf = (float*)take((size_t)cap_shared * 64 * 4 * sizeof(float));
  w.recv = (float*)take((size_t)(world > 1 ? world - 1 : 1) * cap_shared * 64 * 4 * sizeof(float));
  w.sws_bytes = nm_mpm_shared_workspace(world, cap);
  w.sws = take(w.sws_bytes);
  w.total = o;
  return w;
}
extern "C" size_t nm_rollout_shard_workspace(int32_t world, int32_t cap, int32_t cap_shared, int32_t substeps) {
  if (world < 1 || cap < 1 || cap_shared < 1 || substeps < 1) return 0;
  return carve_shard(nullptr, world, cap, cap_shared, substeps).total;
}
static int shard_args_ok(const nm_comm* comm, int32_t cap, int32_t cap_shared, const nm_rollout_cfg* cfg, const void* gridcache,
                         const void* shard_ws, size_t shard_ws_bytes, ShardWs& sw) {
  NM_REQUIRE(comm && comm->all_gather_i32 && comm->all_reduce_sum_f32 && comm->world >= 1 && comm->rank >= 0 && comm->rank < comm->world,
             "nm_comm incomplete");
  NM_REQUIRE(cap >= 1 && cap_shared >= 1, "cap and cap_shared must be positive");
  NM_REQUIRE(gridcache && cfg->grid_cache_blocks >= 1,
             "the sharded roll-out needs the grid cache (the reverse sweep restores the summed grid from it; there is no recompute across ranks)");
  sw = carve_shard(const_cast<void*>(shard_ws), comm->world, cap, cap_shared, cfg->substeps);
  if (!shard_ws || shard_ws_bytes < sw.total) {
    nm_set_error("sharded roll-out workspace too small: need %zu got %zu", sw.total, shard_ws_bytes);
    return NM_ERR_WORKSPACE;
  }
  return NM_OK;
}
int nm_shard_peer_check(nm_mpm* h, const int32_t* gathered, int32_t world, int32_t cap, int32_t rank, uint32_t peers, int32_t* status,
                        void* stream);
int nm_shard_peer_sum(float* buf, const float* recv, size_t count, uint32_t peers, int32_t rank, int32_t world, void* stream);
static bool shard_by_peers(const nm_comm* comm) { return comm->exchange_peers_f32 != nullptr && comm->peers != NM_COMM_ALL_RANKS && comm->world <= 32; }
// the sum of the ranks' exchange buffers: all-reduce over the world, or buffers swapped with the neighbour ranks and added here
static int shard_all_reduce(const nm_comm* comm, const ShardWs& sw, int cap_shared, void* stream) {
  const int64_t count = (int64_t)cap_shared * 64 * 4;
  if (shard_by_peers(comm)) {
    uint32_t peers = comm->peers & ~(1u << comm->rank);
    if (comm->world < 32) peers &= (1u << comm->world) - 1u;
    if (peers && comm->exchange_peers_f32(comm->user, sw.buf, sw.recv, count, peers, stream)) {
      nm_set_error("nm_comm.exchange_peers_f32 failed");
      return NM_ERR_INVALID;
    }
    return peers ? nm_shard_peer_sum(sw.buf, sw.recv, (size_t)count, peers, comm->rank, comm->world, stream) : NM_OK;
  }
  if (comm->all_reduce_sum_f32(comm->user, sw.buf, count, stream)) {
    nm_set_error("nm_comm.all_reduce_sum_f32 failed");
    return NM_ERR_INVALID;
  }
  return NM_OK;
}

// Forward sweep of one rank.  Substep 0 negotiates the frame's exchange list (neighbourhoods all-gathered ONCE, nm_shard.hip);
// every substep is then the unsharded fused substep - grid clear in the elasticity kernel's prologue, g2p inside the
// plasticity kernel - with one pack launch and one all-reduce between the scatter and the grid update, which reads the
// summed blocks straight from the exchange buffer.
extern "C" int nm_rollout_forward_sharded(nm_mpm* h, int32_t n, const nm_rollout_cfg* cfg, const nm_statics* st, const nm_mlp* we,
                                          const nm_mlp* wp, float* states, void* gridcache, void* workspace, size_t workspace_bytes,
                                          const nm_comm* comm, int32_t cap, int32_t cap_shared, void* shard_ws, size_t shard_ws_bytes,
                                          void* stream) {
  NM_REQUIRE(h && cfg && st && we && wp && (states || n == 0), "null pointer");
  NM_REQUIRE(n >= 0 && cfg->substeps >= 1, "bad sizes");
  ShardWs sw;
  int rc = shard_args_ok(comm, cap, cap_shared, cfg, gridcache, shard_ws, shard_ws_bytes, sw);
  if (rc) return rc;
  RolloutWs w = carve_ws(workspace, n);
  if (!workspace || workspace_bytes < w.total) {
    nm_set_error("rollout workspace too small: need %zu got %zu", w.total, workspace_bytes);
    return NM_ERR_WORKSPACE;
  }
  NM_HIP_CHECK(hipMemsetAsync(sw.status, 0, sizeof(int32_t), (hipStream_t)stream));
  if (n > 0) {
    rc = nm_material_prepare2(we, w.perm_e, wp, w.perm_p, stream);
    if (rc) return rc;
  }
  nm_mpm_set_fresh_rows(h, 1);
  const int nrec = n > 0 ? n : 1;     // (a rank without particles still walks the exchange, with empty lists)
  for (int t = 0; t < cfg->substeps && !rc; ++t) {
    nm_particles cur = rec(states, nrec, t), nxt = rec(states, nrec, t + 1);
    if (n > 0) {
      if (t == 0 || !g_forward_pair) {
        GridPrologue pro;
        rc = nm_mpm_prologue_forward(h, &pro);
        if (rc) break;
        rc = nm_material_fwd_launch(n, NM_ELASTICITY, 0.f, cur.F, we, w.perm_e, cur.stress, &pro, nullptr, stream, svd_rec(cfg, n, t, 0),
                                    act_rec(cfg, n, t, 0));  // finetune.py:362
        if (rc) break;
      }
      rc = nm_mpm_forward_prepared_p2g(h, n, st, &cur, stream);        // this rank's scatter (mpm.py:281-290)
      if (rc) break;
    } else {
      rc = nm_mpm_clear_only(h, stream);
      if (rc) break;
    }
    if (t == 0) {
      rc = nm_mpm_dilated_list(h, sw.mine, cap, stream);
      if (rc) break;
      if (comm->all_gather_i32(comm->user, sw.mine, sw.gathered, (int64_t)(1 + cap), stream)) {
        nm_set_error("nm_comm.all_gather_i32 failed");
        rc = NM_ERR_INVALID;
        break;
      }
      rc = nm_mpm_shared_blocks(h, sw.gathered, comm->world, cap, sw.shared, cap_shared, sw.status, sw.sws, sw.sws_bytes, stream);
      if (rc) break;
      rc = nm_shard_slots(h, sw.shared, cap_shared, 1, stream);
      if (rc) break;
      if (shard_by_peers(comm)) {      // every rank this one shares a block with must be among its peers (status bit 16)
        rc = nm_shard_peer_check(h, sw.gathered, comm->world, cap, comm->rank, comm->peers, sw.status, stream);
        if (rc) break;
      }
    }
    rc = nm_shard_pack_fwd(h, sw.shared, cap_shared, sw.buf, sw.held + (size_t)t * sw.held_stride, sw.status, stream);
    if (rc) break;
    rc = shard_all_reduce(comm, sw, cap_shared, stream);
    if (rc) break;
    rc = nm_mpm_forward_gridop_x(h, grid_rec(gridcache, cfg, t), cfg->grid_cache_blocks, sw.status, sw.buf, stream);   // :291-297
    if (rc) break;
    if (n > 0) {
      G2pFuse g2p;
      rc = nm_mpm_g2p_fuse(h, st, &cur, &nxt, &g2p);
      if (rc) break;
      if (g_forward_pair && t + 1 < cfg->substeps) {     // plasticity(t) + elasticity(t+1) + the clear of substep t+1 (see nm_rollout_forward)
        GridPrologue pro;
        rc = nm_mpm_prologue_forward(h, &pro, true);
        if (rc) break;
        rc = nm_material_fwd_pair_launch(n, cfg->plasticity_alpha, w.perm_p, w.perm_e, nxt.F, nxt.stress, &pro, &g2p, stream,
                                         svd_rec(cfg, n, t, 1), svd_rec(cfg, n, t + 1, 0), act_rec(cfg, n, t, 1), act_rec(cfg, n, t + 1, 0));
      } else {
        rc = nm_material_fwd_launch(n, NM_PLASTICITY, cfg->plasticity_alpha, nullptr, wp, w.perm_p, nxt.F, nullptr, &g2p, stream,
                                    svd_rec(cfg, n, t, 1), act_rec(cfg, n, t, 1));  // finetune.py:364
      }
    }
  }
  nm_mpm_set_fresh_rows(h, 0);
  const int rc2 = nm_shard_slots(h, sw.shared, cap_shared, 0, stream);      // the handle's slot map is clean between roll-outs
  if (!rc && !rc2) {      // the status word, OR-ed over the ranks (see k_status_to_flags)
    float* flags = reinterpret_cast<float*>(sw.status + 16);
    NM_LAUNCH(k_status_to_flags, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int32_t*)sw.status, flags);
    NM_LAUNCH_CHECK();
    if (comm->all_reduce_sum_f32(comm->user, flags, 8, stream)) {
      nm_set_error("nm_comm.all_reduce_sum_f32 failed (status word)");
      return NM_ERR_INVALID;
    }
    NM_LAUNCH(k_flags_to_status, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)flags, sw.status);
    NM_LAUNCH_CHECK();
  }
  return rc ? rc : rc2;
}

extern "C" int nm_rollout_backward_sharded(nm_mpm* h, int32_t n, const nm_rollout_cfg* cfg, const nm_statics* st, const nm_mlp* we,
                                           const nm_mlp* wp, const float* states, const void* gridcache, const float* gstate_last,
                                           float* gstate_first, float* gw_e, float* gw_p, void* workspace, size_t workspace_bytes,
                                           const nm_comm* comm, int32_t cap, int32_t cap_shared, const void* shard_ws,
                                           size_t shard_ws_bytes, void* stream) {
  NM_REQUIRE(h && cfg && st && we && wp && gw_e && gw_p && ((states && gstate_last && gstate_first) || n == 0), "null pointer");
  NM_REQUIRE(n >= 0 && cfg->substeps >= 1, "bad sizes");
  hipStream_t s = (hipStream_t)stream;
  ShardWs sw;
  int rc = shard_args_ok(comm, cap, cap_shared, cfg, gridcache, shard_ws, shard_ws_bytes, sw);
  if (rc) return rc;
  RolloutWs w = carve_ws(workspace, n);
  if (!workspace || workspace_bytes < w.total) {
    nm_set_error("rollout workspace too small: need %zu got %zu", w.total, workspace_bytes);
    return NM_ERR_WORKSPACE;
  }
  const size_t N = (size_t)(n > 0 ? n : 1);
  const int nrec = (int)N;
  float* states_m = const_cast<float*>(states);
  const float* gin = gstate_last;
  if (n > 0) {
    rc = nm_material_prepare2(we, w.perm_e, wp, w.perm_p, stream);
    if (rc) return rc;
  }
  const int polar = cfg->svd_adjoint == NM_SVD_ADJOINT_POLAR ? 1 : 0;
  const float dt = nm_mpm_get_dt(h);
  rc = nm_shard_slots(h, sw.shared, cap_shared, 1, stream);      // the frame's slot map again
  if (rc) return rc;
  // The records of a sharded roll-out are never recomputed (an overflow is an error, status bit 4), so the sweep always runs
  // the way the unsharded one does once its records are verified: each substep's grid is restored in the prologue of the
  // constitutive launch in front of it.
  bool restored = false;
  for (int t = cfg->substeps - 1; t >= 0 && !rc; --t) {
    nm_particles cur = rec(states_m, nrec, t), nxt = rec(states_m, nrec, t + 1);
    float* gout = (t == 0) ? gstate_first : ((gin == w.ga) ? w.gb : w.ga);
    const int wmode = (t == cfg->substeps - 1) ? 1 : 2;
    if (n > 0 && t == cfg->substeps - 1) {
      rc = nm_material_bwd_launch(n, NM_PLASTICITY, cfg->plasticity_alpha, cur.F, wp, w.perm_p, gin + 15 * N, w.gFtr, w.part_p, wmode,
                                  nxt.C, st->enabled, dt, polar ? 2 : 0, nullptr, stream, svd_rec(cfg, n, t, 1), act_rec(cfg, n, t, 1));
      if (rc) break;
    }
    nm_particles gn, gc;
    gn.x = const_cast<float*>(gin); gn.v = const_cast<float*>(gin) + 3 * N; gn.C = const_cast<float*>(gin) + 6 * N;
    gn.F = w.gFtr; gn.stress = nullptr;
    gc.x = gout; gc.v = gout + 3 * N; gc.C = gout + 6 * N; gc.F = gout + 15 * N; gc.stress = w.gS;
    // restore the (summed) grid of substep t from its record, scatter this rank's g2p adjoint, sum the exchange blocks of the
    // node-velocity adjoint over the ranks, then the grid-update adjoint (reading them from the buffer) and the p2g adjoint
    rc = nm_mpm_backward_cached_begin(h, n, st, &cur, &nxt, &gn, &gc, grid_rec(gridcache, cfg, t), cfg->grid_cache_blocks, true,
                                      restored, stream);
    if (rc) break;
    restored = false;
    rc = nm_shard_pack_bwd(h, sw.shared, cap_shared, sw.buf, sw.held + (size_t)t * sw.held_stride, stream);
    if (rc) break;
    rc = shard_all_reduce(comm, sw, cap_shared, stream);
    if (rc) break;
    rc = nm_mpm_backward_cached_finish(h, n, st, &cur, &gc, (n > 0 && t > 0) ? grid_rec(gridcache, cfg, t - 1) : nullptr,
                                       cfg->grid_cache_blocks, sw.buf, stream);
    if (rc) break;
    if (n > 0) {
      if (t == 0) {
        rc = nm_material_bwd_launch(n, NM_ELASTICITY, 0.f, cur.F, we, w.perm_e, w.gS, gc.F, w.part_e, wmode, nullptr, nullptr, 0.f,
                                    1 | (polar ? 2 : 0), nullptr, stream, svd_rec(cfg, n, t, 0), act_rec(cfg, n, t, 0));
      } else {
        nm_particles prev = rec(states_m, nrec, t - 1);
        GridPrologue pro;
        rc = nm_mpm_prologue_backward(h, grid_rec(gridcache, cfg, t - 1), cfg->grid_cache_blocks, &pro);
        if (rc) break;
        restored = true;
        rc = nm_material_bwd_pair_launch(n, cur.F, we, w.perm_e, w.gS, gc.F, w.part_e, wmode, cfg->plasticity_alpha, prev.F, wp,
                                         w.perm_p, w.gFtr, w.part_p, 2, cur.C, st->enabled, dt, polar, &pro, stream,
                                         svd_rec(cfg, n, t, 0), svd_rec(cfg, n, t - 1, 1), act_rec(cfg, n, t, 0), act_rec(cfg, n, t - 1, 1));
      }
    }
    gin = gout;
  }
  const int rc2 = nm_shard_slots(h, sw.shared, cap_shared, 0, stream);
  if (rc || rc2) return rc ? rc : rc2;
  if (n == 0) {
    NM_HIP_CHECK(hipMemsetAsync(gw_e, 0, NM_WTOT_ * sizeof(float), s));
    NM_HIP_CHECK(hipMemsetAsync(gw_p, 0, NM_WTOT_ * sizeof(float), s));
    return NM_OK;
  }
  return nm_material_wgrad_reduce2(w.part_e, w.part_p, n, gw_e, gw_p, stream);
}

// status bits of the roll-out's exchanges (1: a rank's neighbourhood list exceeded cap, 2: more exchange blocks than cap_shared,
// 4: a grid cache record overflowed, 8: a particle left the neighbourhood its rank announced at the first substep) - asynchronous copy of one int32 to (pinned) host memory
extern "C" int nm_rollout_shard_status(const void* shard_ws, int32_t* status_host, void* stream) {
  NM_REQUIRE(shard_ws && status_host, "null pointer");
  NM_HIP_CHECK(hipMemcpyAsync(status_host, shard_ws, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return NM_OK;
}
