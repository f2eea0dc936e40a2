/*
 * neuma_hip.h — C ABI of libneuma_hip.so, the MI355X (gfx950) engine behind NeuMA's
 * simulator / constitutive-net / Particle-GS render operators.
 *
 * The reference (XJay18/NeuMA) has no C ABI of its own: it reaches the GPU through Warp's JIT
 * (`wp.launch`) and through the pybind11 torch extension `diff_gaussian_rasterization`.
 * Each entry point below names the reference interface it replaces (file:line relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a NeuMA maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (PyTorch); the library
 *     borrows it for the duration of the call and never frees or retains it.  Exceptions: the
 *     opaque handles, which own their scratch (grid, active-block lists), and `*_cfg` structs
 *     and `int* out` scalars, which are HOST pointers.
 *   - fp32, contiguous, array-of-structures exactly as torch lays them out:
 *     x,v (N,3); C,F,stress (N,3,3) row-major; statics float/int32 (N,); cov6 order
 *     xx,xy,xz,yy,yz,zz; SH (K, M, 3); view/proj matrices 16 floats in torch row-major order of
 *     the (already transposed, row-vector convention) tensors NeuMA passes.
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream).  All calls are
 *     asynchronous on that stream unless stated; no device-wide synchronisation is issued.
 *   - return 0 on success, negative on error; nm_last_error() gives the thread-local message.
 *     Non-finite gradients are not an error (the Python shim applies nan_to_num like
 *     modules/nclaw/sim/interface.py:65-74).
 *   - a handle is not thread-safe; different handles may be used concurrently.
 */
#ifndef NEUMA_HIP_H
#define NEUMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NM_OK 0
#define NM_ERR_INVALID (-1)
#define NM_ERR_HIP (-2)
#define NM_ERR_WORKSPACE (-3)

int nm_version(void);
const char* nm_last_error(void);

/* ------------------------------------------------------------------ in-library kernel timing (bench.py roofline) */

/* HIP-event timing of the library's own kernel launches, on the stream they are launched on.
 * nm_prof_enable(1, NULL) times every kernel; nm_prof_enable(1, "k_render_bwd") only that one (two event
 * records per launch); nm_prof_enable(n > 1, name) times every n-th matching launch only (a pair of event records costs
 * ~11 us of bubble around the launch: sampling keeps a timed region honest); nm_prof_enable(0, NULL) stops.  nm_prof_report synchronises the recorded events and
 * writes "name calls total_ms\n" lines (NUL-terminated, truncated to cap); nm_prof_reset drops the samples. */
int nm_prof_enable(int32_t on, const char* only_kernel);
int nm_prof_report(char* out, size_t cap);
int nm_prof_reset(void);

/* ------------------------------------------------------------------ MPM (modules/nclaw/sim) */

/* MPMConstant, modules/nclaw/sim/mpm.py:158-167; parse_cfg 507-528 (dx = 1/num_grids). */
typedef struct nm_mpm_cfg {
  int32_t num_grids;
  float dt;
  int32_t bound;
  float gravity[3];
  float eps;
  int32_t bc; /* 0 = noslip (mpm.py:402-429), 1 = freeslip (mpm.py:373-400) */
} nm_mpm_cfg;

/* MPMStatics, mpm.py:14-72 */
typedef struct nm_statics {
  const float* vol;
  const float* rho;
  const float* clip_bound;
  const int32_t* enabled;
} nm_statics;

/* MPMParticleData, mpm.py:75-128 (values or gradients; unused members may be NULL) */
typedef struct nm_particles {
  float* x;      /* (N,3)   */
  float* v;      /* (N,3)   */
  float* C;      /* (N,3,3) */
  float* F;      /* (N,3,3) */
  float* stress; /* (N,3,3) */
} nm_particles;

typedef struct nm_mpm nm_mpm; /* MPMModel (mpm.py:245-258): owns the grid */

/* MPMModelBuilder.finalize, mpm.py:543-551.  bc outside {0,1} -> NM_ERR_INVALID (ValueError there). */
int nm_mpm_create(const nm_mpm_cfg* cfg, nm_mpm** out);
int nm_mpm_destroy(nm_mpm* h);

/* MPMModel.forward, mpm.py:279-297: grid clear, p2g, grid_op, g2p.  `next` may alias `cur`
 * (MPMForwardSim, interface.py:131-135).  Writes next->x,v,C,F. */
int nm_mpm_forward(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur,
                   nm_particles* next, void* stream);

/* MPMModel.backward, mpm.py:299-319: recompute p2g + grid_op, then the adjoints of g2p,
 * grid_op, p2g.  gnext: incoming dL/d(x,v,C,F) of the next state (stress ignored);
 * gcur: outgoing dL/d(x,v,C,F,stress) of the current state (overwritten).
 * `next` holds the forward outputs of this step (v, C are read). */
int nm_mpm_backward(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur,
                    const nm_particles* next, const nm_particles* gnext, nm_particles* gcur,
                    void* stream);

/* Grid cache (no reference counterpart; removes the reference's p2g recompute of mpm.py:312-315 from the reverse
 * sweep).  nm_mpm_forward_ex additionally writes a record of the substep's touched 4x4x4-node blocks ({mv, m} of
 * those blocks + the block list; nm_mpm_gridcache_bytes(cap_blocks) bytes, caller-owned device memory);
 * nm_mpm_backward_ex restores the grid from it instead of re-scattering.  A substep that touches more than
 * cap_blocks blocks marks its record invalid and the backward transparently falls back to the recompute, so results
 * never depend on the capacity.  gridrec == NULL: identical to nm_mpm_forward / nm_mpm_backward. */
size_t nm_mpm_gridcache_bytes(int32_t cap_blocks);
int nm_mpm_forward_ex(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur,
                      nm_particles* next, void* gridrec, int32_t cap_blocks, void* stream);
int nm_mpm_backward_ex(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur,
                       const nm_particles* next, const nm_particles* gnext, nm_particles* gcur,
                       const void* gridrec, int32_t cap_blocks, void* stream);

/* MPMModel.forward_extra, mpm.py:260-277: p2g + grid_op from (st, cur), then g2p of a second,
 * passive particle set in place. */
int nm_mpm_forward_extra(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur,
                         int32_t n_extra, const nm_statics* st_extra, nm_particles* extra,
                         void* stream);

/* Particle-sharded substep (no reference counterpart - the reference is single-device).  Every rank
 * owns a fixed subset of the particles and a full grid handle; a grid node receives contributions from several ranks
 * only inside 4x4x4-node blocks that more than one rank touches.  The substep of mpm.py:279-297 / 299-319 is split
 * at the two points where those blocks have to be summed over the ranks, and the caller runs the collectives
 * (RCCL through torch.distributed) in between:
 *
 *   forward : nm_mpm_p2g                                     clear + scatter of the rank's particles (mpm.py:281-290)
 *             nm_mpm_active_list -> all-gather               the rank's touched-block list: out[0] = count, out[1..cap] = ids
 *             nm_mpm_shared_blocks                           blocks listed by >= 2 ranks, in an order all ranks agree on
 *             nm_mpm_blocks_pack(0) -> all-reduce(sum) -> nm_mpm_blocks_unpack(0)     {mv, m} of those blocks
 *             nm_mpm_forward_finish                          grid_op + g2p (mpm.py:291-297), writes the grid cache record
 *   backward: nm_mpm_backward_begin                          restore the grid from the record + scatter of the g2p adjoint
 *             nm_mpm_blocks_pack(1) -> all-reduce(sum) -> nm_mpm_blocks_unpack(1)     adjoint of the node velocities
 *             nm_mpm_backward_finish                         grid_op adjoint + p2g adjoint
 *
 * `shared` (device, int32[2 + 2*cap_shared], caller-owned, kept per substep for the backward): [0] number of shared
 * blocks, [1] status bits, [2..] block ids, [2+cap_shared..] 1 where this rank lists the block itself.  Status bits
 * (also OR-ed into *status, a device int32 the caller polls): 1 = some rank's list exceeded `cap`, 2 = more than
 * cap_shared shared blocks, 4 = the grid cache record overflowed (nm_mpm_forward_finish).  Any bit means the results
 * of that substep are incomplete: the caller must raise, enlarge the capacity and redo the roll-out.
 * pack writes cap_shared*64 float4 (zeros for blocks the rank does not list and for the unused tail), so the
 * all-reduce has a fixed size and needs no host synchronisation. */
int nm_mpm_p2g(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, void* stream);
int nm_mpm_active_list(nm_mpm* h, int32_t* out, int32_t cap, void* stream);
/* The same for the 27-neighbourhood (in blocks) of the rank's touched blocks: what the sharded roll-out announces once per
 * frame (nm_rollout_forward_sharded).  Exported so that the caller can size `cap` / `cap_shared` from a probe. */
int nm_mpm_dilated_list(nm_mpm* h, int32_t* out, int32_t cap, void* stream);
size_t nm_mpm_shared_workspace(int32_t world, int32_t cap);
int nm_mpm_shared_blocks(nm_mpm* h, const int32_t* gathered, int32_t world, int32_t cap, int32_t* shared,
                         int32_t cap_shared, int32_t* status, void* workspace, size_t workspace_bytes, void* stream);
int nm_mpm_blocks_pack(nm_mpm* h, int32_t which, const int32_t* shared, int32_t cap_shared, float* buf, void* stream);
int nm_mpm_blocks_unpack(nm_mpm* h, int32_t which, const int32_t* shared, int32_t cap_shared, const float* buf,
                         void* stream);
int nm_mpm_forward_finish(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* next,
                          void* gridrec, int32_t cap_blocks, int32_t* status, void* stream);
int nm_mpm_backward_begin(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, const nm_particles* next,
                          const nm_particles* gnext, nm_particles* gcur, const void* gridrec, int32_t cap_blocks,
                          void* stream);
int nm_mpm_backward_finish(nm_mpm* h, int32_t n, const nm_statics* st, const nm_particles* cur, nm_particles* gcur,
                           void* stream);

/* Introspection for tests / roofline accounting: number of touched 4x4x4-node blocks and nodes with
 * mass > 0 after the last p2g.  Synchronises the stream. */
int nm_mpm_grid_stats(nm_mpm* h, int32_t* active_blocks, int32_t* nodes_with_mass, void* stream);
/* Copies the dense grid of the last forward (mv (G,G,G,3), m (G,G,G), v (G,G,G,3)) for tests. */
int nm_mpm_grid_export(nm_mpm* h, float* mv, float* m, float* v, void* stream);

/* ------------------------------------------------------------------ SVD (modules/nclaw/warp/svd.py) */

/* SVDFunction.forward / batch_svd, svd.py:12-38, 61-96.  F (N,3,3) -> U (N,3,3), sigma (N,3), Vh (N,3,3);
 * U, V in SO(3), sigma0 >= sigma1 >= |sigma2|, sign(sigma2) = sign(det F). */
int nm_svd3_fwd(int32_t n, const float* F, float* U, float* sigma, float* Vh, void* stream);
/* SVDFunction.backward, svd.py:41-57 (adjoint of wp.svd3 with the 1e-6 denominator clamp). */
int nm_svd3_bwd(int32_t n, const float* U, const float* sigma, const float* Vh, const float* gU,
                const float* gsigma, const float* gVh, float* gF, void* stream);

/* ------------------------------------------------------------------ constitutive nets (modules/nclaw/material/meta.py) */

/* Effective (LoRA-merged, loralib.py:209-213) weights, row-major, no bias (meta.py:20-42):
 * w0 (64,13), w1 (64,64), w2 (9,64). */
typedef struct nm_mlp {
  const float* w0;
  const float* w1;
  const float* w2;
} nm_mlp;

#define NM_ELASTICITY 0 /* InvariantFullMetaElasticity.forward, meta.py:196-221: out = R sym(X) F^T */
#define NM_PLASTICITY 1 /* InvariantFullMetaPlasticity.forward, meta.py:468-489: out = F + alpha R sym(X) */

int nm_material_fwd(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w,
                    float* out, void* stream);
/* Backward: gF (N,3,3) overwritten; gw0/gw1/gw2 (same shapes as the weights) overwritten with the
 * sum over particles (may be NULL to skip weight gradients).  workspace: device scratch of at
 * least nm_material_bwd_workspace(n) bytes. */
size_t nm_material_bwd_workspace(int32_t n);
int nm_material_bwd(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w,
                    const float* gout, float* gF, float* gw0, float* gw1, float* gw2,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Same with flags: NM_BWD_ACCUMULATE adds into gw0/gw1/gw2 instead of overwriting (BPTT over substeps);
 * NM_BWD_POLAR_ADJOINT replaces the reference's SVD adjoint - warp's adj_svd3 behind warp/svd.py:41-57, whose denominators
 * 1/(s_j^2 - s_i^2) are clamped at 1e-6 and which therefore returns ~0 for the rotation path where two singular values
 * coincide (F = I) - by the exact derivative of the polar rotation.  nm_material_bwd == flags 0 == the reference's gradient. */
#define NM_BWD_ACCUMULATE 1
#define NM_BWD_POLAR_ADJOINT 2
int nm_material_bwd_ex(int32_t n, int32_t kind, float alpha, const float* F, const nm_mlp* w,
                       const float* gout, float* gF, float* gw0, float* gw1, float* gw2,
                       int32_t flags, void* workspace, size_t workspace_bytes, void* stream);

/* LinearLoRA merge, modules/nclaw/material/loralib.py:209-213: Weff (out,in) = W + scaling * B (out,r) @ A (r,in), and its
 * adjoint gB = scaling * gW A^T, gA = scaling * B^T gW (what autograd derives from loralib.py:216-224). */
int nm_lora_merge(int32_t out_f, int32_t in_f, int32_t r, float scaling, const float* W, const float* B,
                  const float* A, float* Weff, void* stream);
int nm_lora_merge_bwd(int32_t out_f, int32_t in_f, int32_t r, float scaling, const float* gW, const float* B,
                      const float* A, float* gB, float* gA, void* stream);

/* The same for up to NM_LORA_MAX_LAYERS layers in ONE launch each way (a constitutive net has three, a frame merges both nets' six; per-layer launches
 * of these tiny matrices are pure launch latency).  merge: o0 = W + scaling * B A.  merge_bwd: W holds dL/dW_eff,
 * o0 = dL/dB, o1 = dL/dA. */
#define NM_LORA_MAX_LAYERS 8
typedef struct nm_lora_layer {
  int32_t out_f, in_f, r;
  float scaling;
  const float* W;
  const float* B;
  const float* A;
  float* o0;
  float* o1;
} nm_lora_layer;
int nm_lora_merge_layers(int32_t n, const nm_lora_layer* layers, void* stream);
int nm_lora_merge_layers_bwd(int32_t n, const nm_lora_layer* layers, void* stream);

/* ------------------------------------------------------------------ fused roll-out (experiments/finetune.py:360-364) */

/* S substeps of   stress = E(F); (x,v,C,F) = sim(x,v,C,F,stress); F = P(F)   (finetune.py:362-364,
 * render.py:305-309) enqueued natively, so the host pays one call per video frame instead of three
 * autograd nodes per substep.  `states` is a caller-owned checkpoint buffer of (S+1) records; record t
 * holds x (N,3) | v (N,3) | C (N,9) | F (N,9) | stress (N,9) contiguously (33*N floats; stress = E(F) of that
 * record, written by the forward pass for t < S).  The caller fills x,v,C,F of record 0; records 1..S are
 * written.  Nothing else is kept: the backward pass recomputes the trial F, the grid and all MLP activations
 * from the checkpoints (132 B/particle/substep instead of the reference's ~3.5 KB). */
typedef struct nm_rollout_cfg {
  int32_t substeps;
  float plasticity_alpha;
  int32_t grid_cache_blocks; /* capacity (in 4x4x4-node blocks) of each substep's grid cache record; 0 = no cache */
  int32_t cache_verified;    /* backward only: the caller has read nm_rollout_cache_status and every record is valid, so the
                              * (early-exit) fallback launches of p2g / grid_op can be left out altogether */
  int32_t svd_adjoint;       /* backward only: NM_SVD_ADJOINT_REFERENCE (0, clamped like warp's adj_svd3) or NM_SVD_ADJOINT_POLAR */
  void* svd_cache;           /* optional device buffer of nm_rollout_svdcache_bytes(n, substeps): the forward pass keeps U, sigma, V of
                              * both nets' inputs of every substep (168 B/particle/substep) and the reverse sweep reads them back
                              * instead of repeating the Jacobi SVD; NULL = recompute (same results to rounding: the trial F the
                              * reverse sweep rebuilds from the checkpoints differs from the forward's in the last bit) */
  void* act_cache;           /* optional device buffer of nm_rollout_actcache_bytes(n, substeps): the forward pass keeps what the
                              * reverse sweep cannot cheaply rebuild - the second hidden layer's activations and GELU derivatives
                              * and the output of both nets, 1.2 KB/particle/substep - and the reverse sweep loads it, recomputing
                              * only the first layer (16 of the forward pass's 96 matrix instructions per tile); NULL = recompute
                              * everything.  Same arithmetic, same results. */
  int32_t weights_prepared;  /* nm_rollout_backward only: the workspace is the one the forward call of this node used and
                              * nothing has touched it since, so both nets' weights still sit there in operand order and the
                              * reverse sweep skips that launch.  0 (and every caller that does not know) = prepare them. */
  int32_t last_gF_zero;      /* (not the sharded forms) backward: the caller promises that dL/dF of the last record
                              * (gstate_last + 15 n) is zero everywhere - a loss that sees the final positions only, as the
                              * reference's does (the covariance push-forward is not differentiated, tune/utils.py:353-373).  The
                              * last substep's plasticity adjoint then has nothing to propagate (zero dL/dF, zero weight
                              * gradients) and is not launched.  Given to nm_rollout_forward as well, the last plasticity step
                              * leaves no SVD / activation records (nobody would read them); nm_rollout_backward WITHOUT the flag over
                              * caches whose forward sweep ran with it fails with NM_ERR_INVALID.  0 = make no assumption. */
} nm_rollout_cfg;
#define NM_SVD_ADJOINT_REFERENCE 0
#define NM_SVD_ADJOINT_POLAR 1
size_t nm_rollout_workspace(int32_t n, int32_t substeps);
size_t nm_rollout_svdcache_bytes(int32_t n, int32_t substeps);
size_t nm_rollout_actcache_bytes(int32_t n, int32_t substeps);
/* bytes of the optional `gridcache` buffer: substeps records of nm_mpm_gridcache_bytes(grid_cache_blocks) */
size_t nm_rollout_gridcache_bytes(int32_t substeps, int32_t grid_cache_blocks);
/* Asynchronous read-back of the S record headers of a grid cache into host memory (pinned recommended): status[t] = number
 * of cached blocks of substep t, or -1 if that substep outgrew the capacity.  Valid once the stream has passed this point. */
int nm_rollout_cache_status(const void* gridcache, const nm_rollout_cfg* cfg, int32_t* status_host, void* stream);
/* gridcache (may be NULL): written by the forward pass, handed unchanged to nm_rollout_backward, which then restores
 * each substep's grid instead of recomputing p2g (see nm_mpm_forward_ex). */
int nm_rollout_forward(nm_mpm* h, int32_t n, const nm_rollout_cfg* cfg, const nm_statics* st,
                       const nm_mlp* elasticity, const nm_mlp* plasticity, float* states, void* gridcache,
                       void* workspace, size_t workspace_bytes, void* stream);
/* Forward sweep layout (process-wide, default on): plasticity of substep t and elasticity of substep t+1 - the two calls
 * experiments/finetune.py:364 and :362 make back to back across the loop edge - run as ONE launch per substep boundary, F_{t+1}
 * passing from one net to the other in registers; off = one launch per net.  Results are bit-identical either way. */
int nm_rollout_set_forward_pair(int32_t on);
/* gstate_last: dL/d(x,v,C,F of record S) (24*N floats: x|v|C|F); gstate_first: dL/d(x,v,C,F of record 0) (written);
 * gw_e / gw_p: 5504 floats each = dL/d(w0 | w1 | w2) of the elasticity / plasticity nets, summed over
 * particles and substeps (overwritten). */
int nm_rollout_backward(nm_mpm* h, int32_t n, const nm_rollout_cfg* cfg, const nm_statics* st,
                        const nm_mlp* elasticity, const nm_mlp* plasticity, const float* states,
                        const void* gridcache, const float* gstate_last, float* gstate_first, float* gw_e,
                        float* gw_p, void* workspace, size_t workspace_bytes, void* stream);

/* Particle-sharded roll-out (no reference counterpart - the reference is single-device): the same S-substep
 * node for ONE rank's share of the particles.  The MPM part of every substep is cut at the two points where the grid
 * blocks that several ranks touch have to be summed ("Particle-sharded substep" above); the loop over substeps and phases
 * - launches and collectives alike - runs inside the library, on `stream`.  The collectives themselves are the caller's
 * (one process per GPU: torch.distributed over RCCL; any transport in tests): nm_comm is a table of two functions that
 * must enqueue, in stream order on `stream`,
 *   all_gather_i32      recv[r * count + i] = rank r's send[i]   (count = 1 + cap int32 per rank: the ranks' block lists)
 *   all_reduce_sum_f32  buf[i] = sum over ranks of buf[i]        (count = cap_shared * 256 floats: the shared blocks)
 * and return 0 on success.  send / recv / buf point into `shard_ws`.  Per roll-out: one all-gather (the ranks' block
 * neighbourhoods, negotiated at the first substep) and one 8-float all-reduce (the status bits, OR-ed over the ranks); per
 * substep: one all-reduce of the exchange buffer in the forward pass and one in the reverse sweep - or, with
 * exchange_peers_f32 and a peer set, one swap of the buffer with the neighbour ranks each.  gridcache is mandatory (grid_cache_blocks >= the blocks a rank
 * touches): the reverse sweep restores the already-summed grid from it - there is no recompute across ranks.
 * shard_ws: device memory of nm_rollout_shard_workspace(world, cap, cap_shared, substeps) bytes, written by the forward
 * pass (the shared-block list of every substep) and handed unchanged to the backward pass.  Capacity overflows do not
 * stop the roll-out: they set bits in a status word (nm_rollout_shard_status: 1 = a rank listed more than `cap` blocks,
 * 2 = more than `cap_shared` blocks are shared, 4 = a grid cache record overflowed, 8 = a particle left the neighbourhood its
 * rank announced, 16 = a block is shared with a rank outside nm_comm.peers) - any bit means the results are
 * incomplete and the caller must enlarge the capacity and redo the roll-out.  A rank with n = 0 particles takes part. */
typedef struct nm_comm {
  int32_t world, rank;
  int (*all_gather_i32)(void* user, const int32_t* send, int32_t* recv, int64_t count, void* stream);
  int (*all_reduce_sum_f32)(void* user, float* buf, int64_t count, void* stream);
  void* user;
  /* Optional neighbour-only exchange of the shared blocks (round 5).  A grid block is shared by the 2-3 ranks whose particle
   * ranges meet there, not by the world: instead of all-reducing the exchange buffer over every rank, a rank sends its buffer
   * to the ranks it shares at least one block with (`peers`, bit q = rank q; symmetric; sim/shard.py derives it from the same
   * all-gathered neighbourhood lists on every rank) and adds what they send, in ascending rank order - the same operands
   * in the same order on every owner of a block, so the owners still hold identical sums.
   *   exchange_peers_f32  send `count` floats at `send` to every rank in `peers` and receive that rank's `count` floats into
   *                       recv + k * count, k = the rank's position among the set bits of `peers` (ascending); stream-ordered.
   * NULL or peers == 0xFFFFFFFF: the all-reduce above.  A rank that turns out to share a block with a rank outside `peers`
   * sets status bit 16 (rank-uniform like the others): the caller re-derives `peers` and redoes the roll-out. */
  int (*exchange_peers_f32)(void* user, const float* send, float* recv, int64_t count, uint32_t peers, void* stream);
  uint32_t peers;
} nm_comm;
#define NM_COMM_ALL_RANKS 0xFFFFFFFFu
size_t nm_rollout_shard_workspace(int32_t world, int32_t cap, int32_t cap_shared, int32_t substeps);
int nm_rollout_forward_sharded(nm_mpm* h, int32_t n, const nm_rollout_cfg* cfg, const nm_statics* st,
                               const nm_mlp* elasticity, const nm_mlp* plasticity, float* states, void* gridcache,
                               void* workspace, size_t workspace_bytes, const nm_comm* comm, int32_t cap,
                               int32_t cap_shared, void* shard_ws, size_t shard_ws_bytes, void* stream);
int nm_rollout_backward_sharded(nm_mpm* h, int32_t n, const nm_rollout_cfg* cfg, const nm_statics* st,
                                const nm_mlp* elasticity, const nm_mlp* plasticity, const float* states,
                                const void* gridcache, const float* gstate_last, float* gstate_first, float* gw_e,
                                float* gw_p, void* workspace, size_t workspace_bytes, const nm_comm* comm, int32_t cap,
                                int32_t cap_shared, const void* shard_ws, size_t shard_ws_bytes, void* stream);
int nm_rollout_shard_status(const void* shard_ws, int32_t* status_host, void* stream);

/* Library-owned RCCL communicator (round 4; no reference counterpart - the reference is single-device).
 * With it the sharded roll-out's collectives are issued by the library itself, on the caller's stream, from the loop in
 * nm_rollout_forward_sharded / nm_rollout_backward_sharded: nm_rccl_comm fills an nm_comm whose two functions call
 * ncclAllGather / ncclAllReduce.  librccl is looked up with dlopen on first use (the copy already in the process - torch's -
 * first, then librccl.so.1): no link-time dependency; without RCCL these calls return NM_ERR_INVALID.
 *   nm_rccl_unique_id   rank 0 fills 128 bytes (ncclUniqueId) and hands them to the other ranks by any means
 *                       (neuma_amd/sim/shard.py: one torch.distributed broadcast)
 *   nm_rccl_create      collective over the group: ncclCommInitRank on the current HIP device
 *   nm_rccl_time_all_reduce  mean microseconds of `reps` in-place all-reduces of `count` floats (after `warm` untimed ones):
 *                       the start-up calibration of the shard cost model; collective, synchronises the stream
 *   nm_rccl_library     which librccl was bound ("" if none) */
typedef struct nm_rccl nm_rccl;
const char* nm_rccl_library(void);
int nm_rccl_unique_id(void* id128);
int nm_rccl_create(const void* id128, int32_t world, int32_t rank, nm_rccl** out);
int nm_rccl_destroy(nm_rccl* c);
int nm_rccl_comm(nm_rccl* c, nm_comm* out);
int nm_rccl_all_reduce_sum_f32(nm_rccl* c, float* buf, int64_t count, void* stream);
int nm_rccl_all_gather_i32(nm_rccl* c, const int32_t* send, int32_t* recv, int64_t count, void* stream);
/* nm_comm.exchange_peers_f32 on this communicator: ncclSend / ncclRecv to and from every rank in `peers` inside ONE group call */
int nm_rccl_exchange_peers_f32(nm_rccl* c, const float* send, float* recv, int64_t count, uint32_t peers, void* stream);
/* start-up calibration (like nm_rccl_time_all_reduce): mean microseconds of `reps` such exchanges of `count` floats */
int nm_rccl_time_exchange_peers(nm_rccl* c, const float* send, float* recv, int64_t count, uint32_t peers, int32_t warm, int32_t reps,
                                float* us_out, void* stream);
/* Which ranks share a block with this one: adj[q] = 1 iff a block of rank q's list in `gathered` (world x (1 + cap), the
 * all-gathered neighbourhood lists of nm_mpm_dilated_list) lies in THIS rank's current neighbourhood (device array of `world`
 * int32, adj[rank] = 0).  The predicate is symmetric, so the masks the ranks derive from it match pairwise. */
int nm_mpm_peer_ranks(nm_mpm* h, const int32_t* gathered, int32_t world, int32_t cap, int32_t rank, int32_t* adj, void* stream);
int nm_rccl_time_all_reduce(nm_rccl* c, float* buf, int64_t count, int32_t warm, int32_t reps, float* us_out, void* stream);

/* ------------------------------------------------------------------ Particle-GS binding (modules/tune/utils.py) */

/* torch.sparse.mm(bindings, X) of compute_bindings_xyz / compute_bindings_F, tune/utils.py:424-472,
 * on a CSR copy of the COO binding matrix: out[r,:] = sum_j val[j] * in[col[j],:], D columns.
 * The backward (B^T g) is the same call on the transposed CSR. */
int nm_spmm_csr(int32_t rows, int32_t D, const int32_t* rowptr, const int32_t* col,
                const float* val, const float* in, float* out, void* stream);
/* The reverse sweep's form of the same product (d loss / d particle positions, the adjoint of compute_bindings_xyz,
 * tune/utils.py:424-448, through finetune.py:373's (x - center) / size): out (rows x 3) = scale * A (in0 + in1 + in2 + in3),
 * the inputs being the views' dL/dmeans3D (in1..in3 may be NULL, in order), A the transposed CSR. */
int nm_spmm_csr_sum3(int32_t rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* in0,
                     const float* in1, const float* in2, const float* in3, float scale, float* out, void* stream);
/* Binding construction (data preparation): gaussian_binding_with_clip_v1 / gaussian_binding,
 * modules/d3gs/utils/binding_utils.py:199-285 / 123-196.  Particle j binds to Gaussian k iff (x_j - mean_k)^T inv(cov_k)
 * (x_j - mean_k) <= threshold (= chi2.ppf(confidence, 3)); at most max_particles (<= 16) per Gaussian, the ones with the
 * smallest distance.  The particles are binned into the uniform grid (origin, cell edge, dims) the caller chooses; a
 * Gaussian only visits the cells under its ellipsoid's bounding box.  Outputs (caller-allocated): counts (K) = kept per
 * Gaussian, n_inside (K, may be NULL) = qualifying before the clip, cols (K x max_particles, ascending, -1 padded),
 * pvals (same shape, may be NULL) = their Mahalanobis distances.  Every kept particle has weight 1/counts[k]
 * (softmax of -ones, binding_utils.py:259-269).  No dense K x N matrix is ever formed. */
size_t nm_bind_build_workspace(int32_t n_particles, int32_t ncells);
int nm_bind_build(int32_t K, int32_t N, const float* means, const float* cov6, const float* particles,
                  const float* grid_origin, float cell, const int32_t* grid_dims, float threshold,
                  int32_t max_particles, int32_t* counts, int32_t* n_inside, int32_t* cols, float* pvals,
                  void* workspace, size_t workspace_bytes, void* stream);

/* deform_cov_by_F, modules/d3gs/utils/simulation_utils.py:25-48. */
int nm_cov_deform(int32_t k, const float* cov6, const float* F, float* out_cov6, void* stream);
/* Fused per-frame binding: means3D = k_prev + B (p_cur - p_prev);  F_k = B F;  cov' = F_k cov F_k^T
 * (tune/utils.py:441-471 + simulation_utils.py:25-48 in one pass; F_k never reaches HBM unless F_out != NULL). */
int nm_bind_frame(int32_t k, const int32_t* rowptr, const int32_t* col, const float* val,
                  const float* p_cur, const float* p_prev, const float* k_prev, const float* F,
                  const float* cov6, float* means3D, float* cov6_out, float* F_out, void* stream);

/* ------------------------------------------------------------------ rasterizer (diff_gaussian_rasterization) */

/* GaussianRasterizationSettings as filled by modules/d3gs/gaussian_renderer/__init__.py:103-116.
 * tile_y0/tile_y1: half-open range of 16-pixel tile rows this call renders (0,0 = whole image);
 * used to shard one view across GPUs. */
typedef struct nm_raster_cfg {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  float bg[3];
  float scale_modifier;
  float viewmatrix[16];
  float projmatrix[16];
  int32_t sh_degree;
  float campos[3];
  int32_t prefiltered;
  int32_t debug;
  int32_t tile_y0;
  int32_t tile_y1;
  int32_t split_items;   /* capacity, in (tile, list segment) work items, of the split compositing's records inside the
                          * state buffer (256 x 16 B kept for the backward pass + 256 x 36 B of forward scratch per item);
                          * 0 = 8192.  A view that wants more composites its remaining tiles whole: slower, same result;
                          * nm_raster_forward_ex reports what the view asked for */
} nm_raster_cfg;

/* One caller-allocated state buffer per rendered view replaces the geomBuffer / binningBuffer / imgBuffer tensors of the
 * reference extension (diff_gaussian_rasterization rasterize_points.cu, called from gaussian_renderer/__init__.py:103-119);
 * it is what the backward pass needs and nothing else.  cap_pairs = capacity, in (Gaussian, 64x64-pixel bin) pairs, of the
 * depth-sorted bin lists inside it (a Gaussian of the usual size touches 1-4 bins; 8 * k is a generous first guess). */
size_t nm_raster_state_bytes(const nm_raster_cfg* cfg, int32_t k, int64_t cap_pairs);

/* GaussianRasterizer.forward: preprocess, binning into (bin, depth slab) cells, per-cell LDS depth sort, per-tile
 * streaming composite -> out_color (3,H,W) and radii (K).  shs (K,M,3) or colors_precomp (K,3): exactly one non-NULL.
 * cov3D (K,6) required.  Rows outside the cfg tile stripe are left untouched.  Entirely asynchronous on `stream` (the
 * reference extension synchronises once per view to size its sort buffers).
 * status_host: NULL, or PINNED host memory of two zero-initialised int64 that receive, in stream order, {number of
 * pairs binned, overflow flag}.  overflow != 0: cap_pairs was too small, the image is incomplete - re-run with a state
 * buffer of capacity >= the reported number of pairs. */
int nm_raster_forward(const nm_raster_cfg* cfg, int32_t k, int32_t m, const float* means3D,
                      const float* shs, const float* colors_precomp, const float* opacities,
                      const float* cov3D, int32_t* radii, void* state, size_t state_bytes, int64_t cap_pairs,
                      float* out_color, int64_t* status_host, void* stream);
/* The same with (a) the forward-only arrays in a buffer of their own and (b) a per-camera walk record.
 * (a) nm_raster_state_bytes_ex splits what nm_raster_state_bytes adds up: *state_bytes = what the backward pass reads again
 * (keep it until then), *scratch_bytes = what only the forward pass uses (pair log, cell counters, per-segment scratch: the
 * larger half) - free it, or hand it to the next render on the same stream, as soon as the call has been enqueued.
 * scratch == NULL: one-buffer layout as in nm_raster_forward (state_bytes >= the sum).  nm_raster_count_pairs needs that
 * layout.
 * (b) tile_walk: NULL, or device memory of one uint32 per 16x16 tile of the image (ceil(W/16) * ceil(H/16), row-major),
 * zeroed by the caller when the camera is created and then handed to every render with that camera (the reference keeps
 * no state between two calls of the rasterizer, gaussian_renderer/__init__.py:103-119; this is an addition underneath
 * it).  The forward pass leaves in it how far along its depth-sorted list each tile had to walk, and plans the NEXT render
 * with the same camera from it: tiles with a long walk are cut into segments that are composited in parallel (forward and
 * reverse sweep), however many busy tiles the view has.  A stale or wrong record costs time, never accuracy: image,
 * termination rule and gradients are those of nm_raster_forward.
 * status_host: NULL or pinned memory of THREE zero-initialised int64: {pairs binned, overflow flag, work items the split
 * compositing asked for (compare with cfg.split_items)}. */
int nm_raster_state_bytes_ex(const nm_raster_cfg* cfg, int32_t k, int64_t cap_pairs, size_t* state_bytes,
                             size_t* scratch_bytes);
int nm_raster_forward_ex(const nm_raster_cfg* cfg, int32_t k, int32_t m, const float* means3D,
                         const float* shs, const float* colors_precomp, const float* opacities,
                         const float* cov3D, int32_t* radii, void* state, size_t state_bytes, void* scratch,
                         size_t scratch_bytes, int64_t cap_pairs, float* out_color, int64_t* status_host,
                         uint32_t* tile_walk, void* stream);
/* Tuning of the hinted plan (process-wide).  forward_split_length: tiles whose planned walk is longer than this many list
 * entries are composited in parallel segments in the forward pass as well; shorter ones are walked front to back (leaving
 * checkpoints) and only their reverse sweep runs in segments.  min_segment: shortest segment (rounded up to a multiple of
 * 16).  Defaults 0 (every planned tile: measured best for forward + backward together) and 256. */
int nm_raster_set_hinted(int32_t forward_split_length, int32_t min_segment);
/* Reverse compositing with two pixels per lane (process-wide; read by the following nm_raster_backward calls): pays when
 * several views' reverse sweeps share the chip, costs latency for a view that has it to itself.  Default 0.  Same gradients.
 * (Underneath the reference's stateless rasterizer interface, like the two knobs around it: no counterpart there.) */
int nm_raster_set_reverse_px2(int32_t on);
/* Tuning of the split compositing (process-wide; read by the following nm_raster_forward calls).  A view with fewer than
 * `busy_tiles` non-empty tiles leaves most of the chip idle; its tiles whose depth-sorted list is longer than a segment
 * (>= `min_segment` entries, ~4096 segments per view at most) are walked segment by segment on separate workgroups:
 * always in the reverse sweep (from checkpoints the forward pass leaves in front of every segment), and in the forward
 * pass for the tiles that still have a barely covered pixel after their first segment, provided the view's candidate
 * lists hold no more than `forward_budget` entries together (every segment of such a tile is composited, also those
 * behind the point where its pixels stop).  Same image, same termination rule (forward.cu: stop when T would fall below
 * 1e-4), same gradients.  Defaults 512 (two tiles per CU) / 512 / 2^21; busy_tiles = 0 switches the splitting off,
 * forward_budget = 0 keeps the forward walk sequential.  min_segment is rounded up to a multiple of 16. */
int nm_raster_set_split(int32_t busy_tiles, int32_t min_segment, int64_t forward_budget);
/* Exact number of (Gaussian, 16x16 tile) pairs of the view held in `state` - the `num_rendered` the reference extension
 * returns.  Statistics (byte accounting); synchronises `stream`. */
int nm_raster_count_pairs(const nm_raster_cfg* cfg, int32_t k, void* state, int64_t cap_pairs,
                          int64_t* pairs_out, void* stream);
/* GaussianRasterizer.backward.  dL_dcolor (3,H,W) in; outputs (any may be NULL except dL_dmeans3D):
 * dL_dmeans3D (K,3), dL_dmeans2D (K,3; screen-space mean gradient as the reference returns it),
 * dL_dcov3D (K,6), dL_dopacity (K,1), dL_dshs (K,M,3) or dL_dcolors (K,3).
 * state / cap_pairs: as passed to nm_raster_forward.  workspace: device scratch of nm_raster_bwd_workspace(k) bytes. */
size_t nm_raster_bwd_workspace(int32_t k);
int nm_raster_backward(const nm_raster_cfg* cfg, int32_t k, int32_t m, const float* means3D,
                       const float* shs, const float* colors_precomp, const float* opacities,
                       const float* cov3D, const void* state, int64_t cap_pairs, const float* dL_dcolor,
                       float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcov3D, float* dL_dopacity,
                       float* dL_dshs, float* dL_dcolors, void* workspace, size_t workspace_bytes,
                       void* stream);

/* Fused pixel loss on the rendered image (modules/d3gs/utils/loss_utils.py:17-24):
 * kind 0 = l1 (mean |a-b|), 1 = l2 (mean (a-b)^2); *loss_out (device float) += weight * loss;
 * dL_dimg (3,H,W) = weight * dloss/dimg.  Rows outside [row0,row1) contribute nothing (sharded render). */
int nm_pixel_loss(int32_t kind, float weight, int32_t h, int32_t w, int32_t row0, int32_t row1,
                  const float* img, const float* gt, float* loss_out, float* dL_dimg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUMA_HIP_H */
