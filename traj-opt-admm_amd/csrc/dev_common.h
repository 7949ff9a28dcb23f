// dev_common.h -- device-side view of the solver (all pointers are HBM-resident), wave64
// helpers and small geometry routines shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_gjk.h"

namespace tj {

constexpr int WAVE = 64;
// Bounds of the data-dependent device loops.  The reference has none; each bound below is placed where the reference's own loop
// can no longer end, so reaching it is true non-termination (reported, never silently different):
//  * back-off loops (`step *= 0.8`: Armijo Optimization3D_multi.h:792 / :486, CCD Step.h:89,229): 0.8^k by repeated
//    multiplication reaches its fixed point -- two denormal units, 1e-323 -- at k = 3332; an Armijo test that has not passed by
//    then (e - 1e-4*wolfe*step < E with step, hence E, no longer changing) never passes, a CCD contact still found there is a
//    contact of the state itself.  Dev::pow08 holds all 3333 values, so pow08[min(k, STEP_CAP)] IS the reference's step for any k.
//  * Newton on a pair plane's offset (Optimal_plane.h:23, exit |grad| < 1e-2 only): quadratically convergent on feasible
//    input; beyond NEWTON_CAP rounds the iterate is NaN or cycling (the reference spins).
constexpr int STEP_CAP = 3332;
constexpr int NEWTON_CAP = 4000;
constexpr int MAX_LEVELS = 12;     // 8-ary levels of the static BVH: 8^12 leaves
constexpr int FRONT_CAP = 1024;    // BFS frontier capacity per wave (LDS)

// error bits (Ctl::error)
enum : int {
  ERR_PLANE_OVERFLOW = 1,    // more separating planes for one segment than the configured capacity
  ERR_FRONT_OVERFLOW = 2,    // BVH frontier overflow (query box far larger than expected)
  ERR_LOOP_CAP = 4,          // a back-off / Newton / Armijo loop reached STEP_CAP / NEWTON_CAP: the reference's loop would not end (infeasible state)
  ERR_PAIR_OVERFLOW = 8,     // inter-robot CCD survivor list overflow
  ERR_NOT_SPD = 16,          // coupled mode: the arrowhead Newton system is not positive definite (the reference's
                             // SimplicialLLT has no fallback there either, Optimization3D_multi.h:553-557)
  // detail bits, set together with ERR_LOOP_CAP
  ERR_LS_RANGE = 32,         // coupled mode, SHARDED context: no step among the 8 * LSC_ROUNDS Armijo candidates its exchange carries (0.8^0 .. 0.8^30) was accepted
                             // (one context goes on to the reference's own end: kernels_ls.h lsc_continue)
  ERR_CCD_STUCK = 128,       // a CCD clamp found a contact at every step down to 0.8^STEP_CAP (the fixed point of step *= 0.8): the state itself is in collision (the
                             // reference spins forever there, Step.h:83-97)
  ERR_SLACK_ARMIJO = 256,    // the slack update's Armijo loop
  ERR_PLANE_REFINE = 64,     // "optimal_plane":1 -- a plane's Newton refinement hit PLANE_NEWTON_CAP / PLANE_BACKOFF_CAP
  ERR_PEER_TIMEOUT = 512,    // tj_group, flag transport: a peer's push did not arrive within 2 s (tj_group.h)
  ERR_PASS_TIMEOUT = 1024,   // large fleets: a wave waiting for passed-on robot pairs gave up after 5 ms (the queue was descheduled, e.g. several
                             // processes on one GPU); the pairs it would have taken are unsolved -- set together with ERR_LOOP_CAP
  ERR_XS_TIMEOUT = 2048,     // asynchronous Newton solve: a wait between the two queues of the context (tickets, gate, flags, the records k_ccd's units build) ran out (2 s)
};

#ifdef TJ_NO_DONE_CHECK
#define TJ_DONE(D) false
#else
// ... or a wait between the queues of the context has run out (ERR_XS_TIMEOUT: a GPU shared with another process): what is still enqueued of the batch then returns at
// once instead of running into one 2 s limit after the other -- the host restores the batch's first state and runs it again on one queue (tj_api.hip: heal_check)
#define TJ_DONE(D) ((D).ctl->done | ((D).ctl->error & ERR_XS_TIMEOUT))
#endif
struct Ctl {
  int iter;            // completed iterations (reference global `iter`)
  int done;            // stop test fired: iter>1 && gnorm<stop (Main/multiPathPlanning3D.cpp:633)
  int error;           // ERR_* bits
  int order_ambiguous; // segments whose inter-robot CCD result depended on pair order and were replayed in the reference's tree order (k_ccd_self_seq)
  int pending;         // an iteration was started by k_begin and is not yet counted in `iter`
  int epoch;           // bumped by every k_begin: stamp that marks this iteration's pair-plane slots as valid
  int slack_now;       // the slack/dual update of the PREVIOUS iteration is due (deferred so it overlaps the next planes)
  int slack_next;      // the iteration that k_begin just started still owes its slack/dual update
  int any_pair;        // robot pairs within `offset` at full step this iteration (entries of Dev::pair_list): work of the sequential CCD replay
  int order_unresolved; // segments whose pair order mattered but could not be replayed in the reference's tree order
  int ticket;          // k_linesearch blocks that have finished (the last one does the next iteration's k_begin work); + 65536 per robot that backed off
  int ls_quiet;        // iterations in a row in which every robot accepted the full step (k_linesearch: helpers stay home from LS_QUIET_ITERS on)
  int gjk_prev;        // gjk_max of the iteration before the running one (k_begin): the GJK head start's threshold follows it

  double gnorm;        // reference global `gnorm`
  // statistics for the algorithmic-byte model; accumulated over iterations
  unsigned long long llt_fail_piece, llt_fail_robot;  // PSD repairs taken (per-piece 19x19, per-robot reduced system)
  unsigned long long energy_evals;
  unsigned long long newton_iters, pair_solves;  // unused since the counters moved to Dev::pair_stats (kept for the layout)
  double wolfe_c;      // coupled mode: the global `wolfe` of update_spline (Optimization3D_multi.h:558)
  // coupled mode, one context: the Armijo decision, taken once by the last block of an evaluation round (kernels_ls.h)
  int lsf_epoch, lsf_r, lsf_c, ls_ticket;   // epoch it belongs to / accepted round and slot / arrival counter of the round's blocks
  double lsf_step;
  // length of the longest robot-pair GJK of the running iteration (only pairs with >= 6 iterations report: a handful per launch)
  // and the sum of these maxima over the iterations begun so far: the unit count of k_mid's critical path (bench.py critical_path)
  int gjk_max, spec_taken;   // spec_taken: GJK head starts k_mid continued from (kernels_pairs.h), summed over the iterations
  unsigned long long gjk_max_sum;
  // k_ccd's pair-selection blocks that have finished, counted in two levels (sixteen sub-counters, then this one: hundreds of returning
  // atomics on ONE address serialise at ~13 ns each); the last one runs the sequential pair replay + gnorm (Dev::seq_fold)
  int ccd_ticket, pad3;
  int ccd_sub[16];
  // direct exchange (Dev::xch): robots whose control points [0] / direction records [1] THIS rank has pushed to its peers so far -- every rank
  // pushes once per iteration, so a consumer expects xpush / owned rounds from every peer
  int xpush[2];
  // k_linesearch's helper protocol: primaries that gave up waiting for a post (10 us) and searched on alone / helpers that left after 5 ms without a word.
  // Both are zero on a GPU of the solver's own; a regression to the always-timeout path shows here (tj_stats) instead of only as a slower run.
  int ls_giveups, ls_helper_timeouts;
  int c2_cnt;          // coupled chain with the corner solve folded into k_xsolve (Dev::c2_fold): robots whose Schur-corner terms have been written (zeroed by begin_body)
  // coupled mode, SHARDED context whose caller follows the Armijo search beyond the candidates one exchange carries (Dev::lsc_follow, kernels_ls.h k_ls_commit):
  // 1 = none of the candidates gathered so far passes, nothing has been committed -- the caller evaluates, gathers and decides the next LSC_ROUNDS rounds;
  // lsc_e0: the summed E(x) of the search (formed with the first table, needed by the later ones)
  int lsc_pending, pad4;
  double lsc_e0;
};

// kernels of one iteration, in stream order (unit of tj_profile_kernels and of the phase stamps)
// Union kernels (K_FRONT, K_MID, K_CCD; kernels_step.h) replace their constituents in the single-GPU iteration graph.
enum { K_BEGIN = 0, K_HULLINFO, K_FRONT, K_SEP_OBS /* k_obs_query */, K_SEP_SELF_ROWS, K_MID, K_OBS_SOLVE, K_SEP_SELF_SOLVE,
       K_KEEP,                      // "optimal_plane":1 only: persistent planes, refined every iteration (kernels_keep.h)
       K_SEP_SELF_COMPACT, K_GRAD, K_XSOLVE,
       K_XSOLVE_C2,                 // coupled mode only ("decouple":0)
       K_CCD_PREP, K_CCD, K_CCD_OBS, K_CCD_SELF_PAIRS, K_CCD_SELF_SEQ, K_LINESEARCH,
       K_LS_COUPLED, K_LS_COMMIT,   // coupled mode only
       K_SLACK, K_COUNT };
constexpr int LS_HELP_MAX = 8;                 // k_linesearch: at most this many blocks per robot (16 candidates per super-round)
constexpr int LS_TAB_STRIDE = 3 * LS_HELP_MAX * 2;   // Dev::ls_tab per robot: three rotating sets (super-round % 3) of LS_HELP_MAX blocks x 2 candidates
constexpr unsigned LS_WORD_DONE = 0x7fffffffu;
constexpr int LS_QUIET_ITERS = 8;
constexpr int LS_NARROW_MIN_PLANES = 160;      // k_linesearch: a robot with at least this many planes evaluates a narrow first round (kernels_ls.h)
constexpr unsigned long long LS_TAB_EMPTY = ~0ull;   // a NaN no evaluation produces (and a false "empty" only sends the primary to its own evaluation)
constexpr int LSC_ROUNDS = 4;  // coupled Armijo search: rounds of 8 candidates evaluated per launch (steps 0.8^0 .. 0.8^30)

// Phase stamps for kernel tuning: compiled in only by `make timing` (-DTJ_PHASE_TIMING); thread 0 of
// a block stores the constant-rate wall clock at a phase boundary.  The product build has none.
constexpr int TJ_TIC_BLOCKS = 65536, TJ_TIC_SLOTS = 8;   // blocks per kernel that leave stamps (timing builds only)
#ifdef TJ_PHASE_TIMING
#define TJ_STAMP_(D, kid, slot, thr) do { if (threadIdx.x == (thr) && blockIdx.x < TJ_TIC_BLOCKS) (D).dbg[((size_t)(kid) * TJ_TIC_BLOCKS + blockIdx.x) * TJ_TIC_SLOTS + (slot)] = wall_clock64(); } while (0)
#define TJ_ORDER(v) asm volatile("" :: "v"(v))   /* the value is computed before the next stamp is taken */
#ifdef TJ_PHASE_LIGHT
/* A stamp costs its wave ~0.3 us (s_memrealtime + wait): three per GJK iteration made a slow pair look 60 % slower than it is.
   The light build (`make timing_light`) keeps what cannot distort a chain: the union kernels' block start / end, and for the block
   kernels the entry (TJ_TIC_ENTRY: first instruction, slot 7), the first phase stamp and the last one. */
#define TJ_LIGHT_KEEP(kid, slot) ((kid) == K_MID || (kid) == K_FRONT || (kid) == K_CCD || ((kid) == K_GRAD && ((slot) == 0 || (slot) == 6)) || ((kid) == K_XSOLVE && ((slot) == 0 || (slot) == 6)) || ((kid) == K_LINESEARCH && ((slot) == 0 || (slot) == 5 || (slot) == 6)))
#define TJ_TIC(D, kid, slot) do { if (TJ_LIGHT_KEEP(kid, slot)) TJ_STAMP_(D, kid, slot, 0); } while (0)
#define TJ_TIC_ENTRY(D, kid) TJ_STAMP_(D, kid, 7, 0)
#define TJ_TICB(D, kid, slot) do {} while (0)
#else
#define TJ_TIC(D, kid, slot) TJ_STAMP_(D, kid, slot, 0)
#define TJ_TIC_ENTRY(D, kid) do {} while (0)
#define TJ_TICB(D, kid, slot) TJ_STAMP_(D, kid, slot, 192)   /* first thread of k_grad's second wave group */
#endif
#else
#define TJ_TIC(D, kid, slot) do {} while (0)
#define TJ_TIC_ENTRY(D, kid) do {} while (0)
#define TJ_TICB(D, kid, slot) do {} while (0)
#define TJ_ORDER(v) do {} while (0)
#endif

// direct exchange between sharded contexts (tj_group "flag" transport, or processes that mapped each other's blocks through hipIpc): what a rank needs
// to know about its peers.  Lives in device memory (Dev::xp).
constexpr int XCH_MAX = 16;                    // ranks of a group
constexpr int XF_SEG_STRIDE = 32;              // ints between two segments' completion counters (Dev::xf_seg): a 128-byte line each
struct XchPeers {
  int n;                                       // peers (world - 1)
  int rank[XCH_MAX];                           // their ranks
  double* rx[XCH_MAX][2];                      // their receive buffers: [0] control points [U][3T], [1] direction records [U][xs]
  unsigned long long* cnt[XCH_MAX];            // their arrival counters [2][XCH_MAX] (kind, source rank)
};

struct Dev {
  // ---- parameters (3D.json + hard-coded constants of the mains) ----
  int mode, U, P, res, S, T, N;
  int u0, u1;  // robots owned by this rank: [u0,u1)
  int rank, world;
  // Sharded contexts: the hull cache / swept-hull cache of the robots OTHER ranks own is rebuilt on this rank from their control points / directions by
  // extra one-wave units at the head of k_front / k_ccd (kernels_step.h: xf_hull_body, xf_ccd_body); the pair tiles of the same launch wait for them.
  // xch = 0: the foreign slices are in place when the kernel starts (an all-gather by the caller, or tj_group's event / rccl transports, ran between the launches).
  // xch = 1: DIRECT exchange -- the producing kernels (k_linesearch / k_begin, k_xsolve) store an owned robot's slice straight into every peer's receive
  //          buffer and count it there; the foreign units wait for the owner's count, read the slice from the receive buffer and put it in place.
  int xf;                        // 1: cache units lead k_front / k_ccd: the other ranks' robots (world > 1 and Dev::fuse), or -- xf_all -- ALL robots
  int c2_fold;                   // coupled mode, one context, every robot's k_xsolve block resident at once: the block waits for all corner terms and finishes the arrowhead solve itself (no k_xsolve_c2 launch)
  int xf_all;                    // coupled mode, one context: nobody publishes a hull / swept-hull cache there (one block commits every robot, the direction comes from k_xsolve_c2),
                                 // so the obstacle units of k_front / k_ccd -- one per (robot, segment) -- publish that record themselves (write-through + the segment's counter) before
                                 // they walk, and the k_hullinfo / k_ccd_prep launches drop out of the chain
  __host__ __device__ int xf_want() const { return (xf_all || xs_async || fa_units) ? U : U - (u1 - u0); }                           // robots the units cover (= units per segment)
  __host__ __device__ int xf_units() const { return (xf && !xf_all) ? xf_want() * S : 0; }   // extra one-wave units in the grid (xf_all: the obstacle units of k_front / k_ccd publish their own robot's record instead)
  __host__ __device__ int xf_robot(int i) const { return xf_all ? i : (i < u0 ? i : i + (u1 - u0)); }
  int xch, xch_poll;             // xch_poll = 1: the foreign units poll the arrival counters themselves; 0: a k_xch_wait launch in front of the kernel has (ranks sharing a device)
  const XchPeers* xp;
  // completion counters of the foreign units, one per (kind, SEGMENT), each on a 128-byte line of its own ([2][S][XF_SEG_STRIDE] ints; zeroed by
  // begin_body): the units of segment tr add to word tr, fire and forget, and what reads that segment's records in the same launch -- its pair tiles, a
  // head start on it -- polls that word only.  (One set of sixteen words for everything, as k_ccd's selection blocks use, was a hot line here: 1 280 adds
  // and 450 polling waves on it made k_front 35 us long.)
  int* xf_seg;
  double* rx[2];                 // this rank's receive buffers (uncached memory, written by the peers)
  unsigned long long* xcnt;      // this rank's arrival counters [2][XCH_MAX]: robots of rank r whose slice of kind k has arrived, cumulative
  __host__ __device__ int n_foreign() const { return U - (u1 - u0); }
  __host__ __device__ int foreign_robot(int f) const { return f < u0 ? f : f + (u1 - u0); }            // f-th robot this rank does not own
  __host__ __device__ int owner_of(int u) const { int r = (int)(((long long)(u + 1) * world - 1) / U); while ((long long)r * U / world > u) r--; while ((long long)(r + 1) * U / world <= u) r++; return r; }
  __host__ __device__ int owned_by(int r) const { return (int)((long long)(r + 1) * U / world) - (int)((long long)r * U / world); }
  // ASYNCHRONOUS Newton solve (round 5; single-GPU chain, decoupled / single-UAV modes): k_xsolve runs on a SECOND hardware queue, released when k_mid has finished,
  // i.e. next to k_grad.  Its block of robot u sleeps until the robot's P piece blocks of k_grad have stored their 19 x 19 blocks (write-through) and taken a ticket,
  // solves, leaves direction record and swept-hull cache with write-through stores and raises the robot's flag; k_ccd -- next on the FIRST queue behind k_grad --
  // waits for the flags in its units.  The two kernel boundaries grad -> xsolve -> ccd (each ~5 us between the last useful instruction of one kernel and the first
  // of the next) disappear from the chain.  xs_sync: ints [U][32] tickets | [U][32] flags | [32] robots done | [32] xs_go -- a 128-byte line each; all but the last zeroed by begin_body.
  // The second queue's launch is released by a one-wave gate kernel in front of it (k_xs_gate: it sleeps on the word xs_go until k_grad's first block has stored this
  // chain link's sequence number xs_seq there -- an event recorded between k_mid and k_grad instead costs a 7 us marker on the first queue, measured).
  int xs_async, xs_seq; int* xs_sync;
  __host__ __device__ int* xs_go() const { return xs_sync + ((size_t)2 * U + 1) * 32; }
  __host__ __device__ int* xs_ticket(int u) const { return xs_sync + (size_t)u * 32; }
  __host__ __device__ int* xs_flag(int u) const { return xs_sync + ((size_t)U + u) * 32; }
  __host__ __device__ int* xs_done() const { return xs_sync + (size_t)2 * U * 32; }
  // "optimal_plane":1, multi-UAV, one context: the refinement of the planes stored before this iteration (k_keep part 2) needs nothing of this iteration's broad
  // phase -- only the committed control points -- and is as long as its slowest plane (tens of Newton rounds).  It runs on a THIRD queue from the start of the
  // iteration (gate: k_front's first block has started), next to k_front and k_mid; k_grad (its compaction reads the planes) waits for the waves' sixteen completion
  // counters.  keep_sync: ints [16][32] counters (zeroed by begin_body) | [32] go word.
  int keep_async, keep_seq, keep_waves; int* keep_sync;
  __host__ __device__ int* keep_go() const { return keep_sync + 16 * 32; }
  // ASYNCHRONOUS FRONT (round 6; one context, all three modes, every block of the line-search kernel resident at once): inside a batch the NEXT iteration's k_front
  // runs on the second hardware queue next to this iteration's k_linesearch (coupled mode: the one-launch k_ls_coupled).  Its launch is held back by a one-wave gate
  // (k_fa_gate) until every block of the line search has started (residency counters: all of them are resident then, so a k_front block that sleeps on a flag can never
  // keep a line-search block off a compute unit); the primary block of robot u stores the accepted control net written through and raises the robot's COMMIT FLAG;
  // k_front's obstacle unit of (u, segment) prefetches what does not depend on the net, waits for the flag, forms hull / box / k-DOP intervals itself (the expressions of
  // k_hullinfo: same bits), publishes the record written through and counts itself on the segment's completion counter (xf_seg, kind 0) -- the pair tiles of the launch
  // wait for that count, a GJK head start for its two robots' flags.  The paired line search writes no hull cache (976 B x S per robot and launch: the units' job now).
  // Everything k_front leaves for later kernels goes out written through and every block counts itself done behind its acknowledged stores; the LAST block of the line
  // search -- the one that begins the next iteration -- waits for k_front (its end, or with fa_mid only its blocks' start), so the queue order k_linesearch -> k_mid still
  // holds what k_mid needs of the line search, and the two kernel boundaries k_linesearch -> k_front -> k_mid overlap k_front's work.
  //   fa      the context is eligible (tj_create)
  //   fa_units (per launch) k_linesearch: publish no hull cache -- the k_front that follows forms the records in its units; k_front: do so.  Set for the two launches of a
  //           pairing, and for the one-queue emulation of the schedule (TJ_FRONT_ASYNC_ONE_QUEUE=1: what the counter passes of tools/profile_round.sh run).  An unpaired
  //           k_linesearch -- the last of a batch, every one of a caller that asks for one iteration at a time -- publishes the cache as before and the next k_front reads it
  //   fa_seq  > 0: THIS launch is part of pairing number fa_seq (k_linesearch(i) <-> k_front(i + 1) [<-> k_mid(i + 1)]); all words below are monotonic in it -- nothing is reset
  //   fa_mid  (with fa_seq; grids of k_front that are resident all at once next to the last k_linesearch block): k_linesearch only waits until every k_front block has
  //           STARTED -- so that k_mid's waves, which then follow at once, can never keep a k_front block off the device -- and k_mid<FA> waits for k_front's end ITSELF: its
  //           first block (the watcher) polls the done counters and raises 64 go words, the pair / obstacle solve waves sleep on one of them and then read what k_front
  //           left past the caches; the slack blocks need nothing of k_front and start at once.  The second boundary (k_linesearch -> k_mid) then overlaps k_front's tail.
  //   fa_sync ints, a 128-byte line per word: [16] k_linesearch blocks started | [16] k_front blocks done | [16] k_front blocks started | [64] go words of k_mid |
  //           [U] commit flags | record {-, epoch, done} of the begun iteration
  int fa, fa_units, fa_seq, fa_mid, fa_nls, fa_nfront; int* fa_sync;
  __host__ __device__ int* fa_res(int i) const { return fa_sync + (size_t)(i & 15) * 32; }
  __host__ __device__ int* fa_fdone(int i) const { return fa_sync + (size_t)(16 + (i & 15)) * 32; }
  __host__ __device__ int* fa_fstart(int i) const { return fa_sync + (size_t)(32 + (i & 15)) * 32; }
  __host__ __device__ int* fa_go(int i) const { return fa_sync + (size_t)(48 + (i & 63)) * 32; }
  __host__ __device__ int* fa_commit(int u) const { return fa_sync + (size_t)(112 + u) * 32; }
  __host__ __device__ int* fa_rec() const { return fa_sync + (size_t)(112 + U) * 32; }
  __host__ __device__ static size_t fa_sync_ints(int U_) { return (size_t)(113 + U_) * 32; }
  int lsc_follow;   // coupled mode, sharded context: the caller follows the Armijo search to the reference's end (tj_coupled_search_pending); 0: ERR_LS_RANGE beyond 0.8^30 as in round 5
  int xs_band;  // long trajectories (piece_num > 10): the Newton solve runs on band storage (k_xsolve_band) and the swept-hull
                // cache comes from k_ccd_prep again
  int seq_tree; // k_ccd_self_seq has LDS for the reference's per-segment dynamic tree (dev_dyntree.h)
  int fuse;    // single-GPU iteration graph: k_linesearch leaves the next iteration's hull cache, so k_hullinfo is not
               // launched (the sharded schedule needs the cache for ALL robots after its all-gather and keeps the kernel;
               // folding k_ccd_prep into k_xsolve the same way was measured slower: 10 dependent segments per wave)
  double lambda, margin, offset, mu, vel_limit, acc_limit, ks, kt, stop;
  int cap_obs, cap_self, cap_pairs;
  int optimal_plane;  // "optimal_plane":1 (Optimization3D_admm.h:126-192 obstacle planes in mode 0, Optimization3D_multi.h:276-338 pair planes)
  // ---- tables (row-major 6x6) ----
  const double* basis;    // [S][36]
  const double* convert;  // [P][36]
  const double* mdyn;     // [36]
  const double* kdop;     // [49][3]
  const double* pow08;    // [STEP_CAP+1]  0.8^k by repeated multiplication, up to its fixed point
  // ---- obstacles: Morton-sorted primitives + implicit 8-ary box hierarchy ----
  // prim = 1: a point cloud (the reference's live path, BVH::InitPointcloud BVH.cpp:53-93); prim = 3: triangles (the
  // reference's dormant BVH::InitObstacle / Step::mix_step path, BVH.cpp:15-51, Step.h:313-411: BASELINE config 5)
  int prim;
  const double *px, *py, *pz;   // [N] sorted points (prim == 1)
  const double* tri;            // [N][9] sorted triangles, three vertices row-major (prim == 3)
  const float* leafbox;         // [N][6] outward-rounded fp32 box of each triangle: pre-filter before the exact fp64 test (prim == 3)
  int nlevels;                  // level 0 = boxes over 8 consecutive primitives; top level has <= 64 boxes
  int bvh_skip;                 // 1: the walk takes two levels per step while its frontier is small (kernels_sep.h bvh_query; TJ_BVH_SKIP=0: off -- same candidates, same order)
  int lvl_off[MAX_LEVELS], lvl_n[MAX_LEVELS];
  // Inner boxes are fp32, rounded OUTWARD (lo down, hi up): 24 B instead of 48 B per box visited.  A conservative box can
  // only add visits, never lose a primitive, and the leaf predicate is evaluated in fp64 on the primitive itself
  // (AABB.cc:131-148, touching counts) -- so the candidate SET and its order are exactly those of fp64 boxes.
  const float* boxes;           // [total][6] lo.xyz hi.xyz, padded slots are empty (lo=+inf, hi=-inf)
  // ---- ADMM state, column-major per robot like the reference's Eigen matrices ----
  double *spline;      // [U][3][T]
  double *p_slack;     // [U][3][6P]
  double *p_lambda;    // [U][3][6P]
  double *t_slack;     // [U][P]
  double *t_lambda;    // [U][P]
  double *piece_time;  // [U]
  // ---- per-iteration intermediates ----
  double *oplanes; int *ocount;   // obstacle planes  [U][S][cap_obs][4], [U][S]
  double *splanes; int *scount;   // inter-robot planes [U][S][cap_self][4], [U][S]
  // obstacle-plane pipeline (kernels_sep.h): candidate points per segment after the k-DOP cull, the segment's hull, one
  // (segment, slot) work item per candidate, and per-slot plane + epoch stamp before compaction into oplanes
  int *ocand, *ocand_n;           // [U][S][cap_obs], [U][S]
  double *ohull;                  // [U][S][18]
  int *obs_work, *obs_work_n;     // [U*S*cap_obs][2], [1]
  double *oraw; int *ostamp;      // [U][S][cap_obs][4], [U][S][cap_obs]
  double *hullinfo;               // [U][S][HULL_STRIDE] hull, AABB, k-DOP intervals of the current control net
  // the same AABBs / the swept pair boxes again, robot-minor [S][6][U] (lo.xyz, hi.xyz): the all-pairs box tests read
  // one component of 64 partners with one coalesced load instead of 64 strided 8-byte loads
  double *hbox, *cbox;
  double *pairplane; int *pairstamp;  // [S][U][U][4] plane of robot a against partner b, [S][U][U] epoch stamp
  // index of the stamped slots: bit b of word [S][U][ceil(U/64)] = "the slot of partner b may carry this iteration's stamp" -- set (atomicOr) by whoever stamps a
  // slot, read AND cleared by the one wave that compacts the row, which therefore looks at the stamps and planes of the set bits only (the dense row scan was
  // 1 KB of stamps + 2 KB of speculative planes per segment: 16.8 MB per launch at 256 robots).  A superset is enough: the stamp still decides.
  unsigned long long* pairbits;
  __device__ __forceinline__ void pair_mark(int tr, int a, int b) const { atomicOr(&pairbits[((size_t)tr * U + a) * ((U + 63) >> 6) + (b >> 6)], 1ull << (b & 63)); }
  int* ccd_found;                  // [64] obstacle primitives the CCD stage found inside swept boxes, cumulative, spread over 64 counters (block & 63) so that no
                                   // address is hot; the host sums them when it reads the control block and picks k_ccd's build from the rate
  int pair_rows;                   // rows per tile of the robot-pair broad phase (kernels_pairs.h)
  int mid_order;      // k_mid's grid: 0 slack | pair waves | obstacle solves; 1 (hundreds of robots) pair waves | obstacle solves | slack (kernels_step.h; TJ_MID_ORDER)
  const int* xs_gather;   // k_xsolve's overlap-add as a table (tj_create): [n*n][2] piece-block entries per entry of the reduced system, then [n][2] for the gradient; -1 none, -2 every piece (time entry)
  int pair_prio;      // large fleets: producer waves of k_mid run at wave priority 3 (TJ_PAIR_PRIO=0: off)
  int pair_lpw;       // large fleets, one pair per lane: pairs a producer wave of k_mid takes per pass (64 = every lane; TJ_PAIR_LPW)
  int pair_pass_on;   // 1 (default): large fleets pass long pair solves on to idle waves; TJ_PAIR_PASS_ON=0 keeps every pair on its lane (test hook: same bits either way)
  int* pair_ovf; unsigned long long* pair_ovf_list;   // large fleets: [0] pairs passed on by the lane solve, [1] producer waves done, [2] consumer cursor; entries (epoch << 32 | q << 20 | p0 << 9 | segment), cap_work of them
  int *pair_work; int *pair_work_n; int cap_work;  // (segment, p0, p1) triples that passed box + k-DOP this iteration
  // GJK head start (kernels_pairs.h: spec_pair_body): pairs whose GJK was long in one iteration get the first iterations of the
  // next one's query inside k_front, next to the broad phase; k_mid continues from the saved state.  0 = off (TJ_PAIR_HEAD_START=0: same bits).
  int seq_fold; double* seq_gmem_d; int* seq_gmem_i;   // > 0: k_ccd's last block finishes with k_ccd_self_seq's work (kernels_step.h); the value = acting pairs its sort holds
  int spec, spec_budget, spec_min; int* spec_n; int* spec_list; unsigned long long* spec_tag; double* spec_state; int* spec_sti;
  // "optimal_plane":1 -- planes that persist across iterations (the reference's is_seperate / seperate_c / seperate_d and
  // is_self_seperate / self_seperate_c / self_seperate_d tables, CCDUtils.cpp:30-36).  Obstacle planes (mode 0): a list per
  // (robot, segment) in insertion order, keyed by the sorted point index.  Pair planes (modes 1, 2): dense [S][U][U] table,
  // p0 < p1 only, plus the list of switched-on slots that k_keep strides over.
  int *kobs_id, *kobs_n; double *kobs_cd;       // [U][S][cap_obs], [U][S], [U][S][cap_obs][4]
  int *kpair_on, *kpair_list, *kpair_n; double *kpair_cd;  // [S][U][U], [S*U*U], [2] = {count, count at iteration start}, [S][U][U][4]
  int grad_npl; double* grad_scr;  // k_grad: planes its LDS buffer holds; per-block HBM staging [owned*P][16*(cap_obs+cap_self)] for larger segments
  double *lg, *lh;                // per-piece gradient [U][P][19] and Hessian [U][P][361] (after PSD repair)
  // search direction record per robot, robot-major so a rank's robots are one slice for the
  // all-gather: [U][xs], xs = 3T+4 : direction (T x 3 col-major), t_direction, wolfe, |g|, pad
  double *xdir; int xs;
  __host__ __device__ double* dirp(int u) const { return xdir + (size_t)u * xs; }
  __host__ __device__ double& tdir(int u) const { return xdir[(size_t)u * xs + 3 * T]; }
  __host__ __device__ double& wolfe(int u) const { return xdir[(size_t)u * xs + 3 * T + 1]; }
  __host__ __device__ double& gn(int u) const { return xdir[(size_t)u * xs + 3 * T + 2]; }
  __host__ __device__ bool multi() const { return mode >= 1; }     // robot-pair stages exist
  __host__ __device__ bool coupled() const { return mode == 2; }   // one piece_time for all robots
  // coupled mode (shared piece_time): per-robot Cholesky factor of the reduced block incl. the arrow row, the
  // forward-substituted right-hand side, the raw gradient, {Schur corner, Schur rhs, g_t} contributions,
  // and the per-(round, robot, candidate) energies of the Armijo search on the summed objective
  double *xL, *xy, *xg, *xcorner, *ls_e;
  double *k_obs_f;                // coupled + sharded: every robot's obstacle CCD exponent as a double (exchange buffer 3)
  double *xs_scr;                 // k_xsolve_band: per-robot dense scratch [owned][n*n + 4n] for the eigenvalue fallback
  int *k_obs, *k_self;            // [U] exponents: step = 0.8^k
  double *step_out;               // [U] accepted Armijo step (diagnostics)
  int *ls_hist; int ls_fast;      // [U] Armijo exponent each robot accepted in the previous iteration (-1: none yet) -- k_linesearch evaluates round 0 in the
                                  // team shape for a robot that took the full step last time (kernels_ls.h: x_energy_team); ls_fast = 0 (TJ_LS_FAST=0): never (same bits)
  // helper workgroups of k_linesearch (kernels_ls.h, "super-rounds"): with fewer robots than compute units the launch carries ls_help blocks per
  // robot; block h of a robot evaluates candidates 2h-1 and 2h of a super-round in the team shape on a CU of its own and posts the two energies.
  // k_grad's launch order (kernels_newton.h, grad_order_body): with more (robot, piece) blocks than compute units but fewer than twice as many, the
  // blocks beyond the first `num_cu` share a CU with an older block and run ~20 % slower (the SIMD issues its oldest wave first) -- they set the kernel's
  // length.  Every block leaves the wall-clock ticks it took; the next iteration's k_front ranks them and hands the late positions (and their
  // CU mates) to the cheapest items.  Which block computes which item changes no bit of any item.
  int grad_bal, num_cu;           // 1: k_grad maps blockIdx -> item through grad_perm; compute units of the device
  int *grad_cost, *grad_perm;     // [owned * P] ticks of the item's last block; [owned * P] item of launch position b (always a permutation: identity at the start)
  int ls_help_late;               // TJ_LS_HELP_LATE=<us> (test hook, same bits): helper blocks idle that long before they stage -- they then start AFTER the primary's commit, the case the late-start guard exists for
  int ls_help, ls_help_mute;      // blocks per robot (1 = none); host: compute units / owned robots, at most LS_HELP_MAX.  ls_help_mute (TJ_LS_HELP_MUTE=1, test hook): helpers leave at once
  double *ls_tab;                 // [U][3][LS_HELP_MAX][2] posted energies (set = super-round % 3) (all-ones = not there yet; reset by begin_body and, between super-rounds, by the robot's primary block)
  unsigned long long *ls_word;    // [U] (epoch << 32) | super-round the primary asks for (LS_WORD_DONE: the search is over, helpers leave)
  double *ccdinfo;                // [U][S][CCD_STRIDE] swept-hull cache of the current direction
  int *pair_list;                 // [ACT_CAP] inter-robot CCD: keys (segment, p0, p1) of the pairs within `offset` at full step; Ctl::any_pair counts them
  // per-(robot, segment) statistics slots {nodes_dcd, cand_dcd, nodes_ccd, cand_ccd, planes_obs, planes_self}:
  // each slot is only ever touched by the one wave that owns (robot, segment), so plain += suffices
  // (a shared counter would serialise ~10^4 atomics per iteration on one L2 word)
  unsigned long long* seg_stats;
  unsigned long long* blk_stats;    // [U*P] PSD repairs per piece (k_grad), then [U] energy evaluations per robot (line search): single-writer words, no atomics
  unsigned long long* pair_stats;   // [U*S][2] Optimal_plane::optimal_d iterations / robot pairs solved, spread over (lower robot, segment)
  Ctl* ctl;
  long long* dbg;  // phase stamps (TJ_PHASE_TIMING builds only, else null)
};
// P, D, obstacle box, pair box, 49 k-DOP intervals (lo,hi): 146 values in a record of 160 doubles = ten 128-byte lines of its own.  No line is shared between
// two records, so a record that another block of the SAME launch writes (sharded contexts: foreign units, then pair tiles) cannot sit half-stale in an L2
// that fetched the neighbouring record earlier in the launch.
constexpr int CCD_STRIDE = 160;
constexpr int CCD_REC = 18 + 18 + 6 + 6 + 98;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- the signalling idiom of every cross-block / cross-queue protocol in this library ("Synchronisation protocols") ----
// PRODUCER: the records go out with write-through stores (agent / system scope: `global_store ... sc1`), the wave then waits until every one of them has been
// ACKNOWLEDGED (`s_waitcnt vmcnt(0)`: the counter is per wave; a barrier hands it to the other waves of a block) and only then issues the signal -- a relaxed atomic on a word
// of its own cache line.  CONSUMER: polls the word with a relaxed atomic load (`global_load ... sc1`) and reads the records afterwards, either past the caches (sc1 loads) or
// from lines its XCD's L2 cannot hold yet.  No fence instruction on either side (a release is `buffer_wbl2`: it writes back the whole L2 of the XCD, measured +14 us when
// every block of a large grid does it).  What the HIP memory model does not promise for relaxed atomics the ISA does; so that a compiler change cannot silently take it
// away, the sites are marked with `s_nop` immediates no compiler emits and tests/test_abi_and_host.py::test_signalling_sites_keep_their_order checks, in the disassembly of
// the shipped code object, that between "acknowledged" (0x2a1 + the s_waitcnt) and "sent" (0x2a2) there is no store but the signal itself, and that a wait loop (0x2b1 ...
// 0x2b2) contains no load that is not a poll.
#define TJ_MARK_(imm) asm volatile("s_nop " #imm ::: "memory")
__device__ __forceinline__ void sig_acked() { TJ_MARK_(0x2a1); __builtin_amdgcn_s_waitcnt(0); asm volatile("" ::: "memory"); }   // every store this wave has issued is acknowledged
__device__ __forceinline__ void sig_sent() { TJ_MARK_(0x2a2); }                                                                   // the signal has been issued
__device__ __forceinline__ void wait_begin() { TJ_MARK_(0x2b1); }
__device__ __forceinline__ void wait_end() { TJ_MARK_(0x2b2); }
__device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int prefix_count(unsigned long long m) { return __popcll(m & ((1ull << lane_id()) - 1ull)); }

// a / b for 0 <= a < 2^20, b >= 1, exact: (a + 0.5) / b is at least 0.5 / b away from an integer and the float product is off by less than
// 0.13 / b.  The integer division by a run-time value is ~40 VALU instructions, and several per-lane index computations (segment ->
// piece, segment -> sub-segment) sit in front of the first load of a block.
__device__ __forceinline__ int div_small(int a, int b) { return (int)(((float)a + 0.5f) * (1.0f / (float)b)); }
// hull of segment tr of a T x 3 column-major control net: out[j*3+a] = sum_k basis[j][k] * net[3*piece+k][a]
// accumulated in k order from zero, as every variant in the reference does (e.g. Energy_admm.h:116-129)
__device__ __forceinline__ double hull_entry(const Dev& D, const double* net, int tr, int j, int a) {
  const double* B = D.basis + (size_t)tr * 36 + j * 6;
  const double* col = net + div_small(tr, D.res) * 3 + D.T * a;
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) acc += B[k] * col[k];
  return acc;
}
// ---- direct exchange between sharded contexts (Dev::xch): producer side ---------------------------------------------------------------------
// Ordering without fences: a slice is stored with system-scope (write-through, uncached at the destination) stores, the wave then waits until every one of
// them has been ACKNOWLEDGED (s_waitcnt vmcnt(0): the counter is per wave), and only after that -- and after a barrier if several waves stored -- the
// arrival counter on the peer is bumped.  The consumer reads the counter and then the slice with system-scope loads (kernels_step.h: xch_wait_owner).
// A release fence instead would write back the whole L2 of the XCD from every block (measured in k_linesearch: +14 us).
__device__ __forceinline__ void xch_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double xch_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// robot u's slice of kind k (src[0, count), final; `per` doubles per robot in the receive buffer) to every peer, by the nth threads of one block (or one wave)
template <bool ONE_WAVE>
__device__ __forceinline__ void xch_push_robot(const Dev& D, int kind, int u, int per, const double* src, int count, int tid, int nth) {
  const XchPeers* xp = D.xp;
  const int np = xp->n;
  for (int q = 0; q < np; q++) {
    double* dst = xp->rx[q][kind] + (size_t)u * per;
    for (int i = tid; i < count; i += nth) xch_store(dst + i, src[i]);
  }
  sig_acked();
  if constexpr (!ONE_WAVE) __syncthreads();
  asm volatile("" ::: "memory");
  if (tid < np) __hip_atomic_fetch_add(xp->cnt[tid] + kind * XCH_MAX + D.rank, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  sig_sent();
  if (tid == 0) atomicAdd(&D.ctl->xpush[kind], 1);
}

// ---- consumer side ----
constexpr long long XCH_TIMEOUT_TICKS = 200000000ll;   // 2 s of the 100 MHz wall clock: a lost peer must not hang the device
// All `need` robots of rank r have pushed their slice of kind k for the launch that is running?  (Every rank pushes once per iteration, so the rounds
// THIS rank has pushed -- final when the kernel started -- are the rounds it may expect of a peer.)  Wave-uniform; false = timed out (error bit set).
__device__ __forceinline__ bool xch_wait_owner(const Dev& D, int kind, int r) {
  const unsigned long long need = (unsigned long long)(D.ctl->xpush[kind] / (D.u1 - D.u0)) * (unsigned long long)D.owned_by(r);
  const unsigned long long* w = D.xcnt + kind * XCH_MAX + r;
  wait_begin();
  bool ok = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= need;
  if (!ok) {
    const long long t_end = wall_clock64() + XCH_TIMEOUT_TICKS;
    for (;;) {
      __builtin_amdgcn_s_sleep(4);
      if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= need) { ok = true; break; }
      if (wall_clock64() > t_end) { atomicOr(&D.ctl->error, ERR_PEER_TIMEOUT); break; }
    }
  }
  wait_end();
  return ok;
}
// Foreign-robot units (kernels_step.h) leave their records with write-through stores, wait for the acknowledgements and count themselves done; the pair
// tiles of the same launch -- later in the grid, so every unit is resident or finished when a tile starts -- wait for the count.  Records are line
// aligned (HULL_INFO_STRIDE, CCD_STRIDE), so what a tile then loads cannot have been fetched by its XCD's L2 earlier in the launch.
__device__ __forceinline__ void xf_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double xf_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int* xf_word(const Dev& D, int kind, int tr) { return D.xf_seg + ((size_t)kind * D.S + tr) * XF_SEG_STRIDE; }
__device__ __forceinline__ void xf_signal(const Dev& D, int kind, int tr) {
  sig_acked();
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(xf_word(D, kind, tr), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  sig_sent();
}
// every foreign unit of segment tr has left its record?  one wave; uniform
__device__ __forceinline__ bool xf_wait_seg(const Dev& D, int kind, int tr) {
  const int want = D.xf_want();
  const int* w = xf_word(D, kind, tr);
  wait_begin();
  bool ok = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
  if (!ok) {
    const long long t_end = wall_clock64() + ((D.xch || D.xs_async) ? XCH_TIMEOUT_TICKS + 10000000ll : 500000ll + 100ll * gridDim.x);   // (direct exchange: the units themselves may wait 2 s for a peer)
    for (;;) {
      __builtin_amdgcn_s_sleep(2);
      if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = true; break; }
      if (wall_clock64() > t_end) { if ((threadIdx.x & 63) == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_PASS_TIMEOUT); break; }
    }
  }
  wait_end();
  return ok;
}

// asynchronous Newton solve (Dev::xs_async): wait until word *w has reached `want`.  One wave, uniform; false = timed out (error bit set).
template <int SLEEP = 2>
__device__ __forceinline__ bool xs_wait(const Dev& D, const int* w, int want) {
  wait_begin();
  bool ok = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
  if (!ok) {
    const long long t_end = wall_clock64() + XCH_TIMEOUT_TICKS;   // 2 s: a logic error must not hang the device -- but a GPU shared with another process may leave the
                                                                  // other queue of this context off the hardware for whole time slices (5 ms was too short for that)
    for (;;) {
      __builtin_amdgcn_s_sleep(SLEEP);
      if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = true; break; }
      if (wall_clock64() > t_end) { if ((threadIdx.x & 63) == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_XS_TIMEOUT); break; }
    }
  }
  wait_end();
  return ok;
}
// asynchronous plane refinement (Dev::keep_async): every wave of the refinement launch (third queue) has left its planes?  Called by all threads of a block; uniform.
__device__ __forceinline__ void keep_wait(const Dev& D) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const long long t_end = wall_clock64() + XCH_TIMEOUT_TICKS;   // 2 s (a refinement of thousands of rounds is legitimate: the reference spins on such planes as well)
    wait_begin();
    for (;;) {
      int v = lane < 16 ? __hip_atomic_load(D.keep_sync + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
      v = __shfl(v, 0);
      if (v >= D.keep_waves) break;
      if (wall_clock64() > t_end) { if (lane == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_XS_TIMEOUT); break; }
      __builtin_amdgcn_s_sleep(8);
    }
    wait_end();
  }
  __syncthreads();
  asm volatile("" ::: "memory");
}
// asynchronous front (Dev::fa_seq): the sixteen words at `base` (a 128-byte line each) sum to at least `want`?  One wave, uniform; false = timed out (error bit set).
__device__ __forceinline__ bool fa_wait16(const Dev& D, const int* base, int want) {
  const int lane = threadIdx.x & 63;
  const long long t_end = wall_clock64() + XCH_TIMEOUT_TICKS;
  bool ok = false;
  wait_begin();
  for (;;) {
    int v = lane < 16 ? __hip_atomic_load(base + (size_t)lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    v = __shfl(v, 0);
    if (v - want >= 0) { ok = true; break; }
    if (wall_clock64() > t_end) { if (lane == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_XS_TIMEOUT); break; }
    __builtin_amdgcn_s_sleep(4);
  }
  wait_end();
  return ok;
}
// ... one word has reached `want` (a robot's commit flag)
__device__ __forceinline__ bool fa_wait_flag(const Dev& D, const int* w, int want) {
  wait_begin();
  bool ok = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want >= 0;
  if (!ok) {
    const long long t_end = wall_clock64() + XCH_TIMEOUT_TICKS;
    for (;;) {
      __builtin_amdgcn_s_sleep(4);
      if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want >= 0) { ok = true; break; }
      if (wall_clock64() > t_end) { if ((threadIdx.x & 63) == 0) atomicOr(&D.ctl->error, ERR_LOOP_CAP | ERR_XS_TIMEOUT); break; }
    }
  }
  wait_end();
  return ok;
}
// a block's stores (write-through) have been acknowledged -> it counts itself on one of sixteen words (fire and forget)
__device__ __forceinline__ void fa_count(int* w) {
  sig_acked();
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  sig_sent();
}
__device__ __forceinline__ void xf_store_i(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xf_load_i(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a value that a kernel running at the same time on the other queue will read: written through when the solve is asynchronous
__device__ __forceinline__ void xs_out(bool wt, double* p, double v) { if (wt) xf_store(p, v); else *p = v; }

__device__ __forceinline__ double seg_weight(const Dev& D, int tr) {
  int k = tr - D.res * div_small(tr, D.res);
  return (k + 1) / double(D.res) - k / double(D.res);
}
__device__ __forceinline__ double norm3(double x, double y, double z) { return sqrt(x * x + y * y + z * z); }

// barrier b(d) = -(d-m)^2 ln(d/m) and its derivatives (Energy_admm.h:84-88, Gradient_admm.h:380-384)
__device__ __forceinline__ double barrier(double w, double d, double m) { return -w * (d - m) * (d - m) * log(d / m); }
__device__ __forceinline__ void barrier_d(double w, double d, double m, double& e1, double& e2) {
  e1 = -w * (2 * (d - m) * log(d / m) + (d - m) * (d - m) / d);
  e2 = -w * (2 * log(d / m) + 4 * (d - m) / d - (d - m) * (d - m) / (d * d));
}

// Integer powers x^3 .. x^6 evaluated in double-double and rounded once.  The reference calls
// std::pow(x, k) (glibc, < 0.53 ulp); a once-rounded product agrees with it except on near-ties,
// and costs ~10 instructions where the device pow() costs several hundred.
struct DD { double h, l; };
__device__ __forceinline__ DD dd_sq(double x) { const double h = x * x; return DD{h, fma(x, x, -h)}; }
__device__ __forceinline__ DD dd_mul_d(DD a, double x) { const double h = a.h * x; return DD{h, fma(a.h, x, -h) + a.l * x}; }
__device__ __forceinline__ DD dd_sq(DD a) { const double h = a.h * a.h; return DD{h, fma(a.h, a.h, -h) + 2 * (a.h * a.l)}; }
__device__ __forceinline__ double pow3(double x) { const DD r = dd_mul_d(dd_sq(x), x); return r.h + r.l; }
__device__ __forceinline__ double pow4(double x) { const DD r = dd_sq(dd_sq(x)); return r.h + r.l; }
__device__ __forceinline__ double pow5(double x) { DD q = dd_sq(dd_sq(x)); const double h = q.h + q.l; q = DD{h, (q.h - h) + q.l}; const DD r = dd_mul_d(q, x); return r.h + r.l; }
__device__ __forceinline__ double pow6(double x) { DD c = dd_mul_d(dd_sq(x), x); const double h = c.h + c.l; c = DD{h, (c.h - h) + c.l}; const DD r = dd_sq(c); return r.h + r.l; }

// Sum in the association order of Eigen's 2-wide vectorised reduction (Redux.h) -- used for the
// few scalars that feed Armijo / stop decisions (wolfe, |g|, consensus norms).
__device__ __forceinline__ double esum(const double* e, int n) {
  if (n == 0) return 0;
  int a2 = (n / 4) * 4, a1 = (n / 2) * 2;
  double r;
  if (a1) {
    double r0a = e[0], r0b = e[1];
    if (a1 > 2) {
      double r1a = e[2], r1b = e[3];
      for (int i = 4; i < a2; i += 4) { r0a += e[i]; r0b += e[i + 1]; r1a += e[i + 2]; r1b += e[i + 3]; }
      r0a += r1a; r0b += r1b;
      if (a1 > a2) { r0a += e[a2]; r0b += e[a2 + 1]; }
    }
    r = r0a + r0b;
    for (int i = a1; i < n; i++) r += e[i];
  } else {
    r = e[0];
    for (int i = 1; i < n; i++) r += e[i];
  }
  return r;
}

}  // namespace tj
