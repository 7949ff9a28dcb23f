/* trajadmm.h -- C ABI of the MI355X-native ADMM inner loop (libtrajadmm.so).
 *
 * Drop-in boundary for the hot path of ruiqini/traj-opt-admm.  The reference has no plugin or
 * FFI layer; its narrowest seam is one static call per ADMM iteration from the two mains plus
 * ~30 namespace-scope globals (HighOrderCCD/Utils/CCDUtils.cpp:5-44).  Each entry point below
 * names the reference interface it replaces.  Plain pointers and sizes only; no C++ or torch
 * types.  All matrices use the reference's Eigen layout: column-major, i.e. a T x 3 control net
 * is stored as [x_0..x_{T-1}, y_0.., z_0..].
 *
 * A context owns all device memory (state, BVH, scratch) on one GPU; state stays resident in
 * HBM between calls.  Calls on one context must be serialised by the caller.  Every function
 * returns TJ_OK or a negative TJ_ERR_* code; tj_last_error() gives the message.
 * The library has NO CPU fallback: without a usable HIP device tj_create fails.
 */
#ifndef TRAJADMM_H
#define TRAJADMM_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tj_ctx tj_ctx;

enum {
  TJ_OK = 0,
  TJ_ERR_INVALID = -1,      /* bad argument / call order */
  TJ_ERR_DEVICE = -2,       /* HIP runtime failure or no device */
  TJ_ERR_CAPACITY = -3,     /* a device-side list overflowed (raise cap_* in tj_params) */
  TJ_ERR_NO_PROGRESS = -4,  /* a back-off / Newton / Armijo loop reached the point where the reference's own loop can no longer end (the
                               fixed point of step *= 0.8 after 3332 factors; 4000 Newton rounds): infeasible state -- the reference
                               would spin forever there (Step.h:83-97, Optimal_plane.h:23) */
  TJ_ERR_UNSUPPORTED = -5
};

enum { TJ_MODE_SINGLE = 0,      /* Optimization3D_admm::optimization            (Optimization3D_admm.h:29-33)  */
       TJ_MODE_MULTI_DECOUPLE = 1, /* Optimization3D_multi::optimization_decouple (Optimization3D_multi.h:29-33) */
       TJ_MODE_MULTI_COUPLED = 2   /* Optimization3D_multi::optimization ("decouple":0, one piece_time shared by all robots;
                                      Optimization3D_multi.h:120-174, update_spline :508-639, Step::couple_self_step Step.h:112-182);
                                      tj_get_state returns the shared piece_time for every robot */ };

/* Replaces the parameter globals of CCDUtils.cpp:5-44 that the mains fill from Config_File/3D.json
 * (Main/admmPathPlanning3D.cpp:368-397, Main/multiPathPlanning3D.cpp:478-511) and hard-code
 * (ks, kt: admmPathPlanning3D.cpp:477-478, multiPathPlanning3D.cpp:596-597). */
typedef struct tj_params {
  int mode;            /* TJ_MODE_* */
  int uav_num;         /* global `uav_num` */
  int piece_num;       /* global `piece_num` (= waypoints - 1) */
  int res;             /* "res": segments per piece */
  double lambda;       /* "lambda" */
  double margin;       /* "margin" */
  double offset;       /* "offset" */
  double mu;           /* "mu" */
  double vel_limit;    /* "vel_limit" */
  double acc_limit;    /* "acc_limit" */
  double ks;           /* 1e-8 single / 1e-3 multi */
  double kt;           /* 1 */
  double stop;         /* "stop": device-side stop test iter>1 && gnorm<stop; <=0 disables it */
  int device;          /* HIP device ordinal */
  int rank, world;     /* robot sharding: this context owns robots [rank*U/world, (rank+1)*U/world) */
  int cap_obs;         /* max obstacle planes per (robot, segment); 0 = default 256 */
  int cap_self;        /* max inter-robot planes per (robot, segment); 0 = default min(uav_num - 1, 64) */
  int cap_pairs;       /* max inter-robot CCD candidate pairs per segment; 0 = default */
  int optimal_plane;   /* "optimal_plane" (global is_optimal_plane): 1 = separating planes persist across iterations and are refined
                          by Optimal_plane::optimal_cd (single UAV, obstacle planes: Optimization3D_admm.h:120-192) /
                          self_optimal_cd (multi UAV, robot-pair planes: Optimization3D_multi.h:276-338) instead of being
                          rebuilt by GJK every iteration */
} tj_params;

/* Fills *p with the shipped 3D.json values ("Config File/3D.json") and the mode's ks/kt. */
void tj_default_params(tj_params* p, int mode, int uav_num, int piece_num);

/* The constant tables the mains precompute into globals (init_variable, Main/admmPathPlanning3D.cpp:249-353):
 * convert[P][36] = convert_list (CCDUtils.h:137-170), mdyn[36] = M_dynamic (:172-227), basis[P*res][36] = the
 * subdivide_tree bases blossom(k/res,(k+1)/res) * convert_list[i] (:229-315), kdop[49][3] = normalised k-DOP axes
 * (CCDUtils.cpp:56-119).  Row-major; any pointer may be NULL.  Host only -- needs no context and no GPU. */
int tj_host_tables(int piece_num, int res, double* convert, double* mdyn, double* basis, double* kdop);

/* LIMITS the reference does not have (it sizes everything from the init file, Main/multiPathPlanning3D.cpp:342-467); each is checked
 * and REPORTED, never silently different:
 *   uav_num <= 2048                 TJ_ERR_UNSUPPORTED from tj_create (11-bit robot fields in packed pair keys).  The robot-pair plane tables
 *                                   are dense [segments][uav_num][uav_num] (84 MB at 256 robots, 6 GB at 2048).
 *   piece_num * res <= 511, res <= 16   TJ_ERR_UNSUPPORTED from tj_create.
 *   order-dependent robot-pair clamp (Step.h:213-251: two acting pairs of one segment share a robot): replayed in the reference's
 *                                   tree order for up to 256 acting pairs per segment and 512-1024 (folded replay) / 4096 acting pairs per
 *                                   iteration; beyond that tj_iterate FAILS (TJ_ERR_UNSUPPORTED / TJ_ERR_CAPACITY, error bit 8).
 *   obstacle CCD clamp              the reference's result depends on the order in which its dynamic tree emits the candidates once GJK's
 *                                   `<= offset` decision is not monotone in the step (swept hulls > 1e4 long: directions 1e5 x a real
 *                                   iteration's); there this library returns the largest first-clear exponent over the candidates
 *                                   (tests/golden/backoff_kat.npz pins the regime boundary).
 *   back-off loops                  followed to where the reference's own loop ends (step *= 0.8 to its fixed point 1e-323 after 3332 factors);
 *                                   TJ_ERR_NO_PROGRESS only where the reference would spin forever -- in all three modes since round 5 (the coupled search
 *                                   beyond 0.8^30 is continued by one block, tests/golden/coupled_long_kat.npz).  A SHARDED coupled context (world > 1)
 *                                   exchanges 31 steps at a time: tj_group follows the search by itself, a caller that drives the phases uses
 *                                   tj_set_coupled_follow / tj_coupled_search_pending (below); without them TJ_ERR_NO_PROGRESS (detail bit 32) beyond 0.8^30.
 *   cap_obs / cap_self / cap_pairs  list capacities of tj_params; an overflow is TJ_ERR_CAPACITY with the bit that says which.
 *   several processes per GPU       one context (world == 1) of a fleet up to about one robot per compute unit enqueues its Newton solve and the next iteration's
 *                                   k_front on a SECOND stream of its own, and "optimal_plane":1 its stored planes' refinement on a third: kernels
 *                                   of one queue sleep on words kernels of the other write.  Streams of ONE process run side by side; two PROCESSES that both do this on
 *                                   one GPU shut each other out (the device runs one process's waves at a time) until a 2 s limit fires.  Not an error since round 6:
 *                                   the library restores the state the batch started from, runs the batch again on ONE queue and keeps the one-queue chain for the
 *                                   life of the context (tj_stats.async_fallbacks counts it; same results bit for bit; the incident costs its 2 s once).  TJ_HEAL=0
 *                                   restores round 5's report (TJ_ERR_NO_PROGRESS, error bit 2048); TJ_XS_ASYNC=0 TJ_KEEP_ASYNC=0 avoid the stall up front.  Under
 *                                   rocprofv3's counter collection (which serialises dispatches across queues) the library keeps one queue by itself.  Several
 *                                   contexts in ONE process: HIP lets streams beyond GPU_MAX_HW_QUEUES (4) share hardware queues, where a sleeping kernel would keep another
 *                                   context's kernels back; the contexts that sleep across queues therefore claim their streams out of a per-device budget of that many
 *                                   minus one, and a context that does not fit keeps the one-queue chain from the start (same bits). */
int tj_create(const tj_params* p, tj_ctx** out);
void tj_destroy(tj_ctx* c);
const char* tj_last_error(const tj_ctx* c);

/* Replaces BVH::InitPointcloud (HighOrderCCD/BVH/BVH.cpp:53-93) + vertex_list: uploads the
 * obstacle cloud (row-major n x 3) and builds the static device BVH.  n may be 0 ("init_ob":0). */
int tj_set_cloud(tj_ctx* c, const double* xyz, int n);

/* Obstacles as a TRIANGLE mesh instead of a point cloud (BASELINE config 5).  Replaces BVH::InitObstacle(V, F)
 * (HighOrderCCD/BVH/BVH.cpp:15-51) -- a path the reference ships but never calls: its OBJ reader keeps `v` lines only
 * (CCDUtils.h:320-390) and its live narrow phase hard-wires one-vertex obstacle bodies.  Semantics here = that path with
 * the body-2 loops the reference left commented out enabled (Separate.h:123-131: d0 = min over the triangle's vertices;
 * CCD.h:448-458: k-DOP interval over its vertices; GJK / GJKDCD / KDOPDCD already take the body size from their arguments):
 * broad phase on the triangle's box (BVH.cpp:26-46), then k-DOP, then GJK hull-vs-triangle, CCD clamp like Step::mix_step
 * (Step.h:380-404).  A triangle with three equal vertices behaves bit for bit like the cloud point.
 * vertices is row-major n_vertices x 3, faces row-major n_faces x 3 (0-based).  Replaces any cloud set before. */
int tj_set_mesh(tj_ctx* c, const double* vertices, int n_vertices, const int* faces, int n_faces);

/* Replaces init_variable (Main/admmPathPlanning3D.cpp:249-353 single,
 * Main/multiPathPlanning3D.cpp:342-467 multi): waypoints is [uav_num][piece_num+1][3], already
 * in solver units; builds spline, p_slack = C x, zero duals, t_slack = piece_time = piece_time0,
 * and resets the iteration counter. */
int tj_init_state(tj_ctx* c, const double* waypoints, double piece_time0);

/* State of robot u: the six by-reference arguments of the reference call
 * (spline T x 3, p_slack / p_lambda 6P x 3 column-major, t_slack / t_lambda P, piece_time). */
int tj_get_state(tj_ctx* c, int u, double* spline, double* p_slack, double* p_lambda, double* t_slack, double* t_lambda, double* piece_time);
int tj_set_state(tj_ctx* c, int u, const double* spline, const double* p_slack, const double* p_lambda, const double* t_slack, const double* t_lambda, double piece_time);

/* The hot path.  Runs up to n_iters ADMM iterations entirely on the device (one call of
 * Optimization3D_admm::optimization / Optimization3D_multi::optimization_decouple each), with the
 * mains' stop test evaluated on the device before every iteration.  Outputs (any may be NULL):
 * gnorm = reference global `gnorm` after the last executed iteration, iters_total = reference
 * global `iter`, converged = stop test fired. */
int tj_iterate(tj_ctx* c, int n_iters, double* gnorm, int* iters_total, int* converged);

/* Same work, asynchronous: enqueue only (no host sync, no read-back).  tj_sync waits. */
int tj_iterate_async(tj_ctx* c, int n_iters);
int tj_sync(tj_ctx* c);
/* Stream the context enqueues on (hipStream_t as void*), for event timing by the caller. */
void* tj_stream(tj_ctx* c);
/* Enqueue on a caller-owned stream instead (e.g. the stream a collective library orders against).  (The asynchronous solve's second stream, where it is
 * in use, stays the context's own; whatever is enqueued on the caller's stream behind tj_iterate_async sees the iterations' results as on one stream --
 * the chain's last kernels wait for the second stream's inside the kernels.) */
int tj_set_stream(tj_ctx* c, void* hip_stream);
/* Runs n_iters iterations with a hipEvent pair around EVERY KERNEL on the context's stream and
 * returns the summed device time per kernel in milliseconds (ms[tj_kernel_count()]) and how often each
 * kernel was launched (launches[...], may be NULL).  Same work as tj_iterate; kernel i is tj_kernel_name(i),
 * the name rocprofv3 reports (tj::<name>). */
int tj_profile_kernels(tj_ctx* c, int n_iters, double* ms, int* launches);
int tj_kernel_count(void);
/* kernels the iteration schedules of this context have enqueued so far (tj_iterate*, tj_iterate_phase*, tj_group_iterate incl. the exchange
 * kernels / collectives of its transports): launches per iteration of a schedule = the difference over a batch / its iterations */
long long tj_launch_count(tj_ctx* c);
const char* tj_kernel_name(int i);

/* ---- stage-level access (teacher-forced parity tests, profiling) ---------------------------- */
enum { TJ_STAGE_BEGIN = 0, TJ_STAGE_PLANES_OBS = 1, TJ_STAGE_PLANES_SELF = 2, TJ_STAGE_GRAD = 3, TJ_STAGE_XSOLVE = 4,
       TJ_STAGE_CCD_PREP = 5, TJ_STAGE_CCD_OBS = 6, TJ_STAGE_CCD_SELF = 7, TJ_STAGE_LINESEARCH = 8, TJ_STAGE_SLACK = 9, TJ_STAGE_END = 10 };
int tj_run_stage(tj_ctx* c, int stage);
/* planes of robot u: counts[S] (obstacle planes first, then inter-robot), planes[total][4] = (cx,cy,cz,d);
 * returns total (>=0) or an error; planes may be NULL to query the size. */
int tj_get_planes(tj_ctx* c, int u, int* counts_obs, int* counts_self, double* planes, int cap);
/* broad phase of the last plane stage for (robot u, segment seg): ids[<= cap] = obstacle primitives (indices into the cloud /
 * face list given to tj_set_cloud / tj_set_mesh, BVH traversal order) that passed BVH::DCDCollision (BVH.cpp:149-193) AND the
 * k-DOP cull (CCD::KDOPDCD); *n_broad = how many the box query alone returned for this segment SINCE tj_init_state
 * (a running total: read it after exactly one plane stage).  Returns the number of ids. */
int tj_get_candidates(tj_ctx* c, int u, int seg, int cap, int* ids, int* n_broad);
/* teacher forcing: overwrite robot u's plane lists (obstacle list only is used; self list emptied) */
int tj_set_planes(tj_ctx* c, int u, const int* counts, const double* planes);
int tj_get_direction(tj_ctx* c, int u, double* direction, double* t_direction, double* wolfe, double* gn);
int tj_get_local_grad(tj_ctx* c, int u, int piece, double* g19, double* h361);
int tj_get_steps(tj_ctx* c, double* step_self, double* step_obs, double* step_armijo);
/* Energy_admm::spline_energy (HighOrderCCD/Energy_admm.h:16-44) of every OWNED robot at the current state, against the
 * separating planes of the last iteration (the lists the last tj_iterate built): energy[uav_num], robots of other ranks 0.
 * The value the line search calls E(x); the mains do not print it, parity tests compare it with the reference's. */
int tj_get_energy(tj_ctx* c, double* energy);
/* teacher forcing of the CCD / line-search stages: overwrite robot u's search direction record (direction T x 3 column-major) */
int tj_set_direction(tj_ctx* c, int u, const double* direction, double t_direction, double wolfe, double gn);

/* ---- "optimal_plane":1 : the persistent plane tables (the reference globals is_seperate / seperate_c / seperate_d and
 * is_self_seperate / self_seperate_c / self_seperate_d, CCDUtils.cpp:30-36), for teacher-forced tests and checkpoints ---- */
/* single UAV: stored obstacle planes of (robot u, segment seg): ids = indices into the cloud given to tj_set_cloud,
 * cd[.][4] = (c, d); returns the number stored (may exceed cap; only min(n, cap) are written) */
int tj_get_obs_cache(tj_ctx* c, int u, int seg, int cap, int* ids, double* cd);
int tj_set_obs_cache(tj_ctx* c, int u, int seg, int n, const int* ids, const double* cd);
/* multi UAV: flags[S][U][U] (only p0 < p1 is used) and cd[S][U][U][4] = (c, d) of the plane between robots p0 and p1
 * before it is split into (c, d - offset/2) and (-c, -d - offset/2).  A sharded context (world > 1) tracks, returns and
 * accepts only the pairs that touch one of its own robots. */
int tj_get_pair_cache(tj_ctx* c, int* flags, double* cd);
int tj_set_pair_cache(tj_ctx* c, const int* flags, const double* cd);

/* counters since tj_init_state, for the algorithmic-byte model */
typedef struct tj_stats {
  unsigned long long iters, nodes_dcd, nodes_ccd, cand_dcd, cand_ccd, planes_obs, planes_self, energy_evals, pair_tests;
  unsigned long long llt_fail_piece, llt_fail_robot; /* PSD repairs taken: per-piece 19x19 blocks, per-robot reduced systems */
  unsigned long long newton_iters, pair_solves;      /* Optimal_plane::optimal_d iterations, robot pairs solved */
  int order_ambiguous; /* segments whose inter-robot clamp depended on pair order (two acting pairs sharing a robot): replayed in the
                          order of the reference's per-segment dynamic AABB tree (Step.h:213-251, AABB.cc:669-734) */
  int error_bits;      /* 1 plane list overflow, 2 BVH frontier overflow, 4 a loop hit its cap (detail: 32 coupled Armijo range, 64 plane
                          refinement, 128 CCD contact at every step = state in collision, 256 slack Armijo, 1024 a wait for passed-on pairs timed out, 2048 a wait of the asynchronous Newton solve timed out), 8 pair list overflow,
                          16 coupled Newton system not SPD, 512 tj_group: a peer's slice did not arrive */
  int order_unresolved; /* such segments for which the tree order could NOT be established (result may differ from the reference's;
                           tj_iterate returns TJ_ERR_UNSUPPORTED) -- 0 unless uav_num is in the thousands */
  int head_starts;      /* robot-pair GJK queries whose first iterations ran inside the broad-phase kernel and were continued by the solve
                           kernel (pairs that were slow in the previous iteration; same bits either way) */
  unsigned long long gjk_max_sum; /* sum over the iterations of the longest robot-pair GJK (iterations of openGJK's main loop; pairs below 6
                                     do not report): / iters = unit count of the pair stage's critical path */
  int ls_giveups;         /* line search, helper blocks: primaries that found a helper's post missing after 10 us and searched on alone (same result) */
  int ls_helper_timeouts; /* ... helper blocks that left after 5 ms without a word from their primary.  Both 0 on a GPU of the solver's own */
  int async_fallbacks;    /* batches that were run again on ONE hardware queue because a wait between the context's queues had run out (error bit 2048: a GPU shared with
                             another process).  The context keeps the one-queue chain from then on; the results are the same bits, the incident costs its 2 s limit once */
} tj_stats;
int tj_get_stats(tj_ctx* c, tj_stats* s);
/* the obstacle BVH of the last tj_set_cloud / tj_set_mesh: device time of the build (Morton keys, radix sort, box pyramid;
 * upload excluded) -- the counterpart of the reference's tree construction (BVH.cpp:53-93: 95 ms for 20k points) */
int tj_get_build_info(tj_ctx* c, double* bvh_build_ms, int* built_on_device);

/* (The known-answer hooks of the parity tests -- tj_kat_* -- are declared in trajadmm_kat.h and exist only in the TEST build
 * libtrajadmm_kat.so (-DTJ_KAT): the product library carries no test surface.) */

/* ---- initial-trajectory planner (replaces ompl_init + simplify_path + edge_collision, Main/multiPathPlanning3D.cpp:123-340
 * and HighOrderCCD/OMPL/OMPL.cpp; OMPL itself is not needed) -------------------------------------------------------------- */
/* The reference's motion validator for a batch of straight edges (OMPL.cpp:36-98, multiPathPlanning3D.cpp:123-160):
 * hit[i] = 1 if edge i = edges[i][2][3] comes within d of a cloud point (BVH::EdgeCollision + CCD::GJKDCD) or of one of
 * the prior edges (GJKDCD edge-edge).  The mains use d = offset + margin/2. */
int tj_edge_collision(tj_ctx* c, int n, const double* edges, int n_prior, const double* prior, double d, int* hit);
/* Plans way points for n_robots one after the other (a later robot treats the earlier robots' paths as obstacles, like
 * ompl_init): a deterministic roadmap -- start, goal and `nodes` Halton samples of bound_scale * bounding box of the cloud
 * (0 = the mains' 1.2 single / 1.5 multi), all-pairs visibility evaluated on the device with tj_edge_collision, shortest
 * path -- followed by the reference's simplify_path, a corner check (the hull of the solver's initial control net cuts every
 * corner; a fan of chords across the cut is validated and the edges at a failing corner are halved) and padding to a common
 * way-point count (>= min_waypoints, 0 = 6) by splitting the longest edges.  The reference plans with OMPL's randomised RRTConnect, so paths are not comparable point by point; what is kept
 * is the validity predicate, the post-processing and the output contract.  starts/goals are [n_robots][3]; waypoints is
 * [n_robots][cap_waypoints][3], the first *n_waypoints rows of each robot are written.  Any context with the cloud set
 * will do (tj_create with piece_num = 2 before the number of pieces is known). */
int tj_plan_init(tj_ctx* c, int n_robots, const double* starts, const double* goals, double bound_scale, int nodes, int min_waypoints, int cap_waypoints, double* waypoints, int* n_waypoints);

/* ---- robot sharding across GPUs (one context per rank) -------------------------------------- */
/* Device pointers + element counts of the buffers that must be all-gathered per iteration (robot-major, so a rank's owned
 * robots are one contiguous slice of doubles):
 *   what = 0  control points (spline, 3T doubles per robot)
 *   what = 1  search direction records (3T + 4: direction, t_direction, wolfe, |g|, the robot's share of the time gradient)
 * coupled mode ("decouple":0) only:
 *   what = 2  Schur-corner contributions of the shared piece_time (4 per robot)  -- the arrowhead system of update_spline,
 *             Optimization3D_multi.h:519-557: every rank eliminates its robots' blocks, the corner is their sum
 *   what = 3  obstacle CCD exponent of every robot (1)  -- Step::couple_self_step takes ONE step for all (Step.h:112-182)
 *   what = 4  energies of the Armijo candidates (4 rounds x 8 per robot)  -- the test is on the SUM over robots (:605-636) */
int tj_exchange_buffer(tj_ctx* c, int what, void** dev_ptr, int* doubles_per_robot, int* first_owned, int* n_owned);
/* Iteration split for external collectives.  Decoupled / single-UAV, 3 phases:
 *   phase 0 stop test | gather 0 | phase 1 planes, gradient, Newton direction | gather 1 | phase 2 CCD clamps, line search
 * coupled, 6 phases:
 *   phase 0 | gather 0 | phase 1 planes, gradient, per-robot elimination | gather 2 | phase 2 corner pivot + back substitution |
 *   gather 1 | phase 3 CCD clamps, shared step, gnorm | gather 3 | phase 4 Armijo candidates (all rounds) | gather 4 | phase 5 commit
 * Results are bitwise those of one unsharded context. */
int tj_phase_count(tj_ctx* c);
int tj_iterate_phase(tj_ctx* c, int phase);
/* The same with the caller saying whether another iteration follows in this batch (more != 0).  Decoupled / single-UAV schedules then run
 * the FUSED chain of one context inside the phases: phase 2's line search also does the next iteration's stop test and counter resets, so
 * that iteration's phase 0 launches nothing -- 6 kernels + the caller's 2 collectives per iteration (round 4: 10 + 2).  A begin that was
 * folded for an iteration the caller never enqueues is taken back by the next tj_sync / state access.  tj_iterate_phase(c, p) is
 * tj_iterate_phase_chained(c, p, 0).  The caches of the robots other ranks own (hull cache, swept-hull cache) are rebuilt from the gathered
 * buffers by extra units at the head of k_front / k_ccd (csrc/kernels_step.h), bitwise what the owner holds. */
int tj_iterate_phase_chained(tj_ctx* c, int phase, int more);

/* ---- direct exchange between sharded contexts (decoupled mode, world > 1; csrc/kernels_step.h) ---------------------------------------------
 * Instead of a collective between the phases, the PRODUCING kernels store an owned robot's slice straight into every peer's receive block
 * (k_linesearch / k_begin: control points for Optimization3D_multi.h:246-259 separate_self; k_xsolve: the direction record for Step.h:196-208
 * self_step) and bump an arrival counter there; the (foreign robot, segment) units at the head of the peers' k_front / k_ccd wait for the
 * count, read the slice and rebuild that robot's cache records.  With it a sharded iteration is the six-kernel chain of one context
 * (tj_iterate_async / tj_iterate work on the sharded context) -- nothing on the host, no launch for the exchange.  tj_group's "flag"
 * transport is this, wired inside one process; two or more PROCESSES (one per GPU, e.g. under torchrun) wire it through hipIpc:
 *   every rank:  tj_xch_ipc_export(c, handle)            64-byte handle of its receive block (allocated on first use, uncached memory)
 *                ... all-gather the handles with any host-side collective ...
 *                tj_xch_ipc_open(c, handle_of_peer, &base) for every other rank
 *                tj_xch_attach(c, world - 1, peer_ranks, peer_bases);  tj_xch_enable(c, 1, wait_mode)
 *   then tj_iterate_async / tj_iterate on every rank; every rank must run the same number of iterations per batch, and a host-side barrier
 *   must separate "every rank has drained its batch" from tj_init_state (which restarts the counters) and tj_init_state from the next batch.
 * wait_mode = 1: the foreign units poll the arrival counters themselves (ranks on distinct devices); 0: a one-wave launch in front of
 * k_front / k_ccd polls instead (ranks SHARING a device: polling units would hold the LDS the peer's producing kernel needs); 2: nobody polls --
 * the CALLER orders the streams (tj_group's event transport: an event recorded behind the producing kernel, waited for by the consumer's stream).
 * A push that does not arrive within 2 s fails the batch (TJ_ERR_DEVICE, error bit 512).  Results are bitwise those of one context.  UNVERIFIED ACROSS xGMI, like
 * tj_group on distinct devices: the tests run ranks and processes on one device. */
int tj_xch_block(tj_ctx* c, void** base, size_t* bytes);                 /* this rank's receive block (same-process wiring: hand `base` to the peers' tj_xch_attach) */
int tj_xch_ipc_export(tj_ctx* c, void* handle64);                        /* hipIpcGetMemHandle of the block */
int tj_xch_ipc_open(tj_ctx* c, const void* handle64, void** base);       /* hipIpcOpenMemHandle of a peer's block (closed by tj_destroy) */
int tj_xch_attach(tj_ctx* c, int n_peers, const int* peer_ranks, void* const* peer_bases);
int tj_xch_enable(tj_ctx* c, int on, int wait_mode);

/* ---- several GPUs under one process (csrc/tj_group.h) ------------------------------------------------------------------
 * What a maintainer of Main/multiPathPlanning3D.cpp would call instead of tj_create / tj_iterate to use N devices: the robots
 * of the per-robot loops (Optimization3D_multi.h:29-118, :120-174) are block-partitioned over n_ranks contexts, rank r on HIP
 * device devices[r] (NULL: device r; entries may repeat -- several ranks on one device, which is how the tests run it on a
 * one-GPU box).  tj_group_iterate runs the phase schedule above on every rank (one host thread per rank) and exchanges the
 * tj_exchange_buffer slices through one of three transports (csrc/tj_group.h):
 *   "event"  hipEventRecord behind the producing kernel / hipStreamWaitEvent in front of the consuming one: plain HIP stream semantics; THE DEFAULT.
 *            Decoupled mode: the slices travel by the direct exchange's in-kernel pushes (six kernels per iteration and rank, nothing launched for
 *            the exchange); coupled mode: a push kernel and an unpack kernel per exchange
 *   "flag"   decoupled mode: the DIRECT exchange above (tj_xch_*): the producing kernels push, the consuming kernels wait -- the fused
 *            six-kernel chain per rank, no launch and no host work for the exchange (ranks sharing a device: two one-wave wait launches);
 *            coupled mode: peer stores + a sequence flag polled by a small unpack kernel.  Opt-in until it has run across xGMI (a push
 *            that does not arrive within 2 s fails the batch with TJ_ERR_DEVICE / error bit 512)
 *   "rccl"   ncclCommInitAll + one in-place ncclAllGather per exchange on each rank's solver stream (the collective
 *            Optimization3D_multi's sharding would use over xGMI); librccl.so is opened at run time, only for this transport;
 *            needs distinct devices; uav_num not divisible by n_ranks: one grouped ncclBroadcast per owner instead
 * chosen by TJ_GROUP_TRANSPORT at tj_group_create or by tj_group_set_transport between batches.  Results are bitwise those of
 * one context.  tj_params.rank / world / device are ignored (set per rank).  Single-UAV mode has nothing to shard (n_ranks
 * must be 1).  UNVERIFIED ON HARDWARE: a group whose devices are all distinct (the configuration the feature exists for) has
 * not run yet -- no multi-GPU box was available to rounds 1-3; same-device groups are tested bitwise against one context. */
typedef struct tj_group tj_group;
/* Coupled mode ("decouple":0) on SHARDED contexts: the Armijo search on the summed energy has no bound in the reference (Optimization3D_multi.h:605-636); one exchange of
 * buffer 4 carries the candidates 0.8^0 .. 0.8^30.  Default: a search that needs more ends in TJ_ERR_NO_PROGRESS (error bit 32), as in round 5.  With follow = 1 phase 5
 * commits nothing in that case and the caller -- after phase 5 of every iteration -- asks tj_coupled_search_pending (it drains the context's stream: the one host look of the
 * schedule); while it answers 1: run phase 4, exchange buffer 4, phase 5 again (they evaluate, carry and decide the next 32 candidates) and ask again.  Every rank reads
 * the same answer.  tj_group does this by itself: a batch that runs into error bit 32 is run again from its first state with the followed search.  One context needs
 * neither call (its deciding block goes on alone). */
int tj_set_coupled_follow(tj_ctx* c, int on);
int tj_coupled_search_pending(tj_ctx* c, int* pending);
int tj_group_create(const tj_params* p, int n_ranks, const int* devices, tj_group** out);
void tj_group_destroy(tj_group* g);
int tj_group_size(tj_group* g);
tj_ctx* tj_group_ctx(tj_group* g, int rank);          /* rank's context: stats, planes, caches of the robots it owns */
const char* tj_group_last_error(tj_group* g);         /* g == NULL: why the last tj_group_create failed */
int tj_group_set_cloud(tj_group* g, const double* xyz, int n);   /* the obstacle BVH is replicated on every device */
int tj_group_set_mesh(tj_group* g, const double* vertices, int n_vertices, const int* faces, int n_faces);
int tj_group_init_state(tj_group* g, const double* waypoints, double piece_time0);
int tj_group_iterate(tj_group* g, int n_iters, double* gnorm, int* iters_total, int* converged);   /* like tj_iterate */
int tj_group_get_state(tj_group* g, int u, double* spline, double* p_slack, double* p_lambda, double* t_slack, double* t_lambda, double* piece_time);   /* from u's owner */
const char* tj_group_transport(tj_group* g);          /* "flag", "event" or "rccl" */
int tj_group_set_transport(tj_group* g, const char* name);   /* between batches; restarts the exchange sequence numbers */
/* event-timed cost of one exchange of each buffer kind (microseconds, slowest rank's average over `reps`): us[5], kinds 2..4
 * are zero outside coupled mode; all zero for one rank */
int tj_group_profile_exchange(tj_group* g, int reps, double* us);
/* 1 if librccl.so can be opened and exports the entry points the "rccl" transport binds (needs no GPU) */
int tj_rccl_available(void);
/* ranks the group's RCCL communicator reports (ncclCommCount); 0 unless the "rccl" transport is selected */
int tj_group_rccl_ranks(tj_group* g);
/* After a rank failed inside tj_group_iterate the ranks' exchange counts disagree: every tj_group_* call except
 * tj_group_init_state (which drains the streams and restarts the counters) and tj_group_destroy then returns TJ_ERR_DEVICE. */

#ifdef __cplusplus
}
#endif
#endif
