// tj_api.hip -- host side of libtrajadmm.so: context, device memory, launch sequence of one ADMM
// iteration, and the C ABI declared in include/trajadmm.h.
//
// One iteration = the stage sequence of Optimization3D_multi::optimization_decouple
// (Optimization3D_multi.h:29-118) / Optimization3D_admm::optimization (Optimization3D_admm.h:29-67),
// enqueued with plain launches and no host synchronisation inside or between iterations: a linear chain on the context's
// stream, plus -- one context -- the Newton solve on a second stream next to the gradient kernel (Dev::xs_async, dev_common.h)
// and, inside a batch, the NEXT iteration's k_front on that stream next to the line search (Dev::fa; TJ_USE_GRAPH=1 replays a
// captured hipGraph of the one-queue chain instead).  A wait between the queues that runs out is healed, not reported (heal_check).  The stop test of the mains
// runs on the device (begin_body), so a converged problem turns the remaining launches into early-exit kernels.
//
// There is deliberately no CPU path in this file: every entry point either runs HIP kernels or
// fails with TJ_ERR_DEVICE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/trajadmm.h"
#include "dev_common.h"
#include "host_tables.h"
#include "kernels_sep.h"
#include "kernels_pairs.h"
#include "kernels_keep.h"
#include "kernels_newton.h"
#include "kernels_step.h"
#ifdef TJ_KAT
#include "../../include/trajadmm_kat.h"
#include "kernels_debug.h"   // known-answer kernels: test build only (libtrajadmm_kat.so)
#endif
#include "kernels_plan.h"
#include "kernels_bvh.h"

using namespace tj;

#include <atomic>
// streams claimed by the contexts of this process whose kernels sleep across queues, per device (tj_create)
static std::atomic<int> g_async_queues[64];
static int hw_queue_budget() { const char* e = getenv("GPU_MAX_HW_QUEUES"); const int n = e ? atoi(e) : 4; return std::max(n, 2) - 1; }

struct tj_ctx {
  tj_params prm;
  Dev d;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  bool maybe_deferred = false;                    // a graph iteration ran since the last flush
  std::vector<void*> allocs;
  std::string err;
  bool have_cloud = false, have_state = false;
  // asynchronous Newton solve (Dev::xs_async): k_xsolve goes to a second hardware queue, behind a one-wave gate kernel that the iteration's k_grad opens (xs_seq: the
  // sequence number of that pairing; xs_seq_gated: the last one a gate was launched for; xs_same_queue_now: tj_profile_kernels keeps everything on one queue)
  hipStream_t stream3 = nullptr; int keep_seq = 0; bool keep_two_queues = false;   // asynchronous plane refinement (Dev::keep_async)
  hipStream_t stream2 = nullptr; int xs_seq = 0, xs_seq_gated = 0; bool xs_two_queues = false, xs_same_queue_now = false;
  // asynchronous front (Dev::fa): fa_seq = pairings k_linesearch(i) <-> k_front(i + 1) launched so far (the device's words are monotonic in it); fa_armed: the last
  // k_linesearch enqueued belongs to pairing fa_seq and the k_front that follows goes to the second queue behind k_fa_gate
  int fa_seq = 0; bool fa_armed = false;
  int hwq_claim = 0; bool hwq_refused = false;   // hardware queues this context claimed out of the process's budget for contexts that sleep across queues (tj_create)
  bool hull_from_units = false, fa_emulate = false;   // the last k_linesearch enqueued published no hull cache (the next k_front forms the records in its units) / TJ_FRONT_ASYNC_ONE_QUEUE=1: the schedule's data flow on one queue
  // Self-healing of the cross-queue schedules: a wait between the queues that runs out (ERR_XS_TIMEOUT -- in practice a GPU shared with another process, whose time slices
  // keep one of the queues off the hardware) must not fail a run.  The first tj_iterate_async after a point at which the host has looked at the device takes a snapshot of the
  // state (one launch); when the host next looks and finds the bit, it latches every two-queue schedule off, restores the snapshot, enqueues the same iterations again on
  // the one queue and counts the incident (tj_stats.async_fallbacks).  TJ_HEAL=0: off (the bit is reported as TJ_ERR_NO_PROGRESS, as in round 5).
  bool heal = false, heal_busy = false, snap_in_begin = false; long long snap_iters = 0; int async_fallbacks = 0, xs_fault = 0;
  SnapRegion* snap_tab = nullptr; int snap_n = 0; Ctl* ctl_snap = nullptr;
  bool fa_mid_ok = false, fa_mid_now = false;   // Dev::fa_mid: k_front's whole grid is resident at once next to one k_linesearch block (tj_create) / the k_mid about to be enqueued waits for k_front itself
  bool use_graph = false;    // TJ_USE_GRAPH=1: replay a captured hipGraph per iteration instead of plain launches
  bool hull_valid = false;   // Dev::fuse: the hull cache matches the control points (else k_hullinfo runs before the next iteration)
  bool ccd_valid = false;    // Dev::fuse: the swept-hull cache of the owned robots matches their direction records (k_xsolve's tail wrote it; tj_set_direction / tj_set_state clear it)
  long long launches = 0;    // kernels enqueued by the iteration schedules so far (tj_launch_count)
  bool begin_folded = false; // the last k_linesearch enqueued has already begun the next iteration (begin_next): the next phase 0 / iteration launches no k_begin
  // direct exchange between sharded contexts (Dev::xch): this rank's receive block (uncached), the peers' blocks as mapped here, the device table
  void* xch_block = nullptr; size_t xch_bytes = 0; bool xch_ipc_exported = false;
  std::vector<void*> xch_ipc_opened;
  XchPeers* xch_table = nullptr;
  bool xf_used[2] = {false, false};   // the cache units of k_front [0] / k_ccd [1] have counted themselves done since the last begin: a repeat of that launch before the next begin first zeroes the counters
  bool xch_wait_kernel = false;   // direct exchange, wait mode 0: a one-wave k_xch_wait launch in front of k_front / k_ccd
  // graph of one full iteration
  // hipGraphs: [0..2] the three phases of a sharded iteration, [3] one full iteration
  hipGraph_t graph[4] = {nullptr, nullptr, nullptr, nullptr};
  hipGraphExec_t gexec[4] = {nullptr, nullptr, nullptr, nullptr};
  bool graph_ok[4] = {false, false, false, false};
  bool graph_failed[4] = {false, false, false, false};
  size_t lds_grad = 0, lds_xs = 0, lds_xs2 = 0, lds_ls = 0, lds_seq = 0;
  bool lsc_wide = false;     // coupled mode: k_ls_coupled evaluates all LSC_ROUNDS rounds in one launch (kernels_ls.h)
  int lsc_base = 0;          // coupled mode, sharded context that follows the Armijo search (Dev::lsc_follow): first round of the table the next phases 4 / 5 evaluate and decide (0 at every iteration's start)
  bool ccd_lean = true;        // which build of k_ccd the chain launches (kernels_step.h); re-decided whenever the control block is read
  unsigned ccd_found_seen = 0; long long iters_enqueued = 0, iters_seen = 0;
  bool grad_fold = true;       // k_grad compacts its own segments (one launch less); TJ_GRAD_FOLD=0 keeps k_sep_self_compact + the 192-thread k_grad
  int n_solve_env = 0;         // TJ_N_SOLVE: pair-solve waves of k_mid (launch-shape switch)
  bool split_unions = false;   // k_front / k_ccd as two launches each (hundreds of robots), see launch_kernel
  LsLayout lsl;
  // cloud-dependent allocations (rebuilt by tj_set_cloud)
  std::vector<void*> cloud_allocs;
  std::vector<int> cloud_order;   // sorted position -> index in the caller's cloud (ids of tj_get/set_obs_cache)
  double cloud_lo[3] = {0, 0, 0}, cloud_hi[3] = {0, 0, 0};   // bounding box of the cloud (planner bounds, Main/multiPathPlanning3D.cpp:211-218)
  double bvh_build_ms = 0;   // device time of the last BVH build (tj_get_build_info)
  int bvh_on_device = 0;
};

// EVERY environment switch of the library is read through this one function (tj_group.h included): TJ_TUNE="KEY=value,KEY=value" or, equivalently, TJ_KEY=value
// (an entry of TJ_TUNE wins over the single variable).  INTEGRATION.md lists them all with their kind -- feature switch, launch-shape switch (same bits), test hook --, and
// tests/test_abi_and_host.py::test_every_environment_switch_is_documented compares that table with the keys that appear here.
static const char* tune(const char* key) {
  static thread_local std::string hold;
  if (const char* t = getenv("TJ_TUNE")) {
    const std::string all(t), k(key);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t end = all.find(',', pos); if (end == std::string::npos) end = all.size();
      const size_t eq = all.find('=', pos);
      if (eq != std::string::npos && eq < end && all.compare(pos, eq - pos, k) == 0) { hold = all.substr(eq + 1, end - eq - 1); return hold.c_str(); }
      pos = end + 1;
    }
  }
  return getenv((std::string("TJ_") + key).c_str());
}

namespace {

#define HIPCHK(c, call)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return TJ_ERR_DEVICE;                                                                      \
    }                                                                                            \
  } while (0)

template <class T>
int dalloc(tj_ctx* c, T** p, size_t n, std::vector<void*>* list = nullptr) {
  void* q = nullptr;
  HIPCHK(c, hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
  HIPCHK(c, hipMemsetAsync(q, 0, std::max<size_t>(n, 1) * sizeof(T), c->stream));  // ordered on the solver's stream like every kernel and copy that follows
  (list ? *list : c->allocs).push_back(q);
  *p = (T*)q;
  return TJ_OK;
}

int upload(tj_ctx* c, const void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return TJ_OK;
  // on the solver's stream (a non-blocking stream has no implicit ordering against the null stream), then waited for:
  // the host buffer may be reused as soon as this returns
  HIPCHK(c, hipMemcpyAsync((void*)dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TJ_OK;
}

void drop_graph(tj_ctx* c) {
  for (int i = 0; i < 4; i++) {
    if (c->gexec[i]) { hipGraphExecDestroy(c->gexec[i]); c->gexec[i] = nullptr; }
    if (c->graph[i]) { hipGraphDestroy(c->graph[i]); c->graph[i] = nullptr; }
    c->graph_ok[i] = false; c->graph_failed[i] = false;
  }
}

// every kernel the iteration schedules enqueue is counted (tj_launch_count: launches per iteration of a schedule, bench.py / tests)
#define TJ_LAUNCH(...) do { c->launches++; hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// ---- kernels of one iteration, in stream order (also the unit of tj_profile_kernels) ----
const char* const kKernelNames[K_COUNT] = {"k_begin", "k_hullinfo", "k_front", "k_obs_query", "k_sep_self_rows", "k_mid", "k_obs_solve", "k_sep_self_solve", "k_keep", "k_sep_self_compact",
                                           "k_grad", "k_xsolve", "k_xsolve_c2", "k_ccd_prep", "k_ccd", "k_ccd_obs", "k_ccd_self_pairs", "k_ccd_self_seq",
                                           "k_linesearch", "k_ls_coupled", "k_ls_commit", "k_slack"};

// launch exactly one kernel (returns false for kernels that do not exist in this mode / schedule).
// in_graph: the launch belongs to the single-GPU iteration graph, a linear chain on one queue in which independent
// stages share a launch (union kernels k_front / k_mid / k_ccd instead of their constituents), the slack/dual update
// is the deferred one inside k_mid, and -- except in coupled mode -- the hull cache comes from k_linesearch (Dev::fuse).
// chain_pos (single-GPU chain only): 0 = an iteration on its own (k_begin launched, k_linesearch plain); bit 1 = this iteration's
// begin work was done by the previous iteration's k_linesearch (no k_begin launch); bit 2 = this iteration's k_linesearch also
// does the next iteration's begin work.
bool launch_kernel(tj_ctx* c, int kid, hipStream_t s, int slack_deferred = 0, bool in_graph = false, bool in_phase = false, int chain_pos = 0) {
  Dev d_ = c->d;
  if (!in_graph) d_.xs_async = 0;   // the asynchronous solve's tickets and flags belong to the single-GPU chain (begin -> k_grad -> k_xsolve -> k_ccd, every iteration); stage API and phases: plain
  d_.xs_seq = 0; d_.keep_seq = 0;
  const bool keep2q = in_graph && c->keep_two_queues && !c->xs_same_queue_now && !c->use_graph;
  if (!keep2q) d_.keep_async = 0;
  if (kid == K_GRAD && d_.xs_async && c->xs_two_queues && !c->xs_same_queue_now && !c->use_graph) d_.xs_seq = ++c->xs_seq;   // this k_grad opens the gate of its k_xsolve
  d_.fa_seq = 0; d_.fa_mid = 0; d_.fa_units = 0;
  if (!in_graph) d_.fa = 0;   // (the context switch belongs to the single-GPU chain like xs_async; fa contexts are never sharded, and the stage API rebuilds the hull cache itself)
  const bool fa2q = in_graph && d_.fa && c->xs_two_queues && !c->xs_same_queue_now && !c->use_graph;
  const Dev& d = d_;
  const int owned = d.u1 - d.u0;
  const bool multi = d.mode >= 1, coupled = d.mode == 2, tri = d.prim == 3;
  // Waves striding over the two device-built work lists.  k_mid holds ~1 wave per SIMD (VGPR bound), i.e. 1024 resident
  // waves: a larger grid adds no parallelism, only dispatch time for blocks that find no work (measured: with
  // 4096 + 1024 blocks the last ones started 50 us into a 60 us kernel).
  // Large fleets (one pair per lane, long solves passed on to idle waves -- sep_self_solve_body): half as many waves again, they
  // are the consumers of the passed-on pairs (SCN-D: k_mid 62 us with 1024, 58 with 1536, 62 with 2048).
  // Small fleets (a wave per pair, two or three pairs per wave): 1 728 = what is left of k_mid's 2 048 resident waves beside SCN-C's
  // 320 slack blocks (1 024: k_mid 31.5 us, 1 536: 28.2, 1 728: 27.6, 2 048: 27.6 before the static assignment; alike after it).
  const int n_solve = (multi && !d.optimal_plane) ? std::min(d.cap_work, c->n_solve_env > 0 ? c->n_solve_env : (d.U >= 192 ? 1536 : 1728)) : 0;  // "optimal_plane":1 -- k_keep finds and refines the pair planes
  const int n_obs_solve = d.N > 0 ? 1024 : 0;   // (512: SCN-E's k_mid 43.7 us, 1024: 38.3, 2048: 37.8; SCN-C indifferent)
  const int n_rows = multi ? d.S * pair_units(d.U, d.pair_rows) : 0;   // one wave per (segment, tile of pair_rows lower robots x 64 partners)
  const int n_xf = d.xf_units();   // sharded contexts: one wave per (foreign robot, segment) at the head of k_front / k_ccd (kernels_step.h); coupled chain: per (robot, segment)
  const int n_ccd = owned * d.S + n_rows, n_front = n_ccd + n_xf + (d.spec ? SPEC_CAP : 0) + (d.grad_bal ? (owned * d.P + 63) / 64 : 0);
  const bool chained = in_graph || in_phase;          // an iteration chain (one context, or the phases of a sharded schedule) as opposed to the stage API
  const int n_mid_slack = owned * d.P;
  d_.fa_nfront = n_front; d_.fa_nls = coupled ? owned * LSC_ROUNDS : owned * d.ls_help;
  switch (kid) {
    case K_BEGIN: if (chain_pos & 1) return false;
      if (c->snap_in_begin) { c->snap_in_begin = false; TJ_LAUNCH(k_begin, dim3(1 + 128), dim3(256), 0, s, d, c->snap_tab, c->snap_n, c->ctl_snap); }   // (self-healing: the batch's snapshot rides in this launch)
      else TJ_LAUNCH(k_begin, dim3(1), dim3(256), 0, s, d, nullptr, 0, nullptr);
      c->xf_used[0] = c->xf_used[1] = false; return true;
    case K_HULLINFO: if ((chained && (d.fuse || d.xf_all)) || !multi) return false; TJ_LAUNCH(k_hullinfo, dim3(d.U * d.S), dim3(64), 0, s, d); return true;  // unfused sharded phases (coupled mode): always (all robots, after the gather)
    case K_FRONT: if (!in_graph && !in_phase) return false;
      if (c->split_unions && multi) {
        // hundreds of robots: the union is bound by how many one-wave blocks are resident (LDS of the BVH frontier: 14 per CU), and
        // the pair rows need none of that LDS -- two launches, the second one at full occupancy, beat one boundary saved
        if (owned * d.S > 0) { if (tri) TJ_LAUNCH((k_obs_query<3>), dim3(owned * d.S), dim3(64), 0, s, d); else TJ_LAUNCH((k_obs_query<1>), dim3(owned * d.S), dim3(64), 0, s, d); }
        TJ_LAUNCH(k_sep_self_rows, dim3(n_rows), dim3(64), 0, s, d);
        return true;
      }
      if (d.xf) { if (c->xf_used[0]) (void)hipMemsetAsync(d.xf_seg, 0, (size_t)d.S * XF_SEG_STRIDE * sizeof(int), s); c->xf_used[0] = true; }
      if (d.xch && c->xch_wait_kernel) TJ_LAUNCH(k_xch_wait, dim3(1), dim3(64), 0, s, d, 0);   // ranks sharing a device: the wait for the peers' control points is a launch of its own
      if (keep2q) d_.keep_seq = ++c->keep_seq;   // this k_front opens the gate of the iteration's plane refinement (third queue)
      if (c->fa_armed && fa2q && (chain_pos & 1)) {   // asynchronous front: on the second queue, next to the k_linesearch just enqueued (pairing fa_seq), behind the residency gate
        c->fa_armed = false;
        d_.fa_seq = c->fa_seq; d_.fa_units = 1; d_.fa_mid = c->fa_mid_ok ? 1 : 0; c->fa_mid_now = c->fa_mid_ok;
        TJ_LAUNCH(k_fa_gate, dim3(1), dim3(64), 0, c->stream2, d, (int)((unsigned)c->fa_seq * (unsigned)d.fa_nls));
        if (tri) TJ_LAUNCH((k_front<3, true>), dim3(n_front), dim3(64), 0, c->stream2, d); else TJ_LAUNCH((k_front<1, true>), dim3(n_front), dim3(64), 0, c->stream2, d);
        return true;
      }
      d_.fa_units = (in_graph && c->hull_from_units) ? 1 : 0;   // (the k_linesearch before it published no hull cache: one-queue emulation of the asynchronous front)
      if (tri) TJ_LAUNCH((k_front<3>), dim3(n_front), dim3(64), 0, s, d); else TJ_LAUNCH((k_front<1>), dim3(n_front), dim3(64), 0, s, d);
      if (keep2q) {
        d_.keep_seq = 0;
        TJ_LAUNCH(k_keep_gate, dim3(1), dim3(64), 0, c->stream3, d, c->keep_seq);
        TJ_LAUNCH(k_keep, dim3(d.keep_waves), dim3(64), 0, c->stream3, d, 2);
      }
      return true;
    case K_SEP_OBS: if (in_graph || in_phase) return false;  // stage API and sharded phase 0
      if (tri) TJ_LAUNCH((k_obs_query<3>), dim3(owned * d.S), dim3(64), 0, s, d); else TJ_LAUNCH((k_obs_query<1>), dim3(owned * d.S), dim3(64), 0, s, d);
      return true;
    case K_OBS_SOLVE: if (in_graph || !n_obs_solve) return false;
      if (tri) TJ_LAUNCH((k_obs_solve<3>), dim3(n_obs_solve), dim3(64), 0, s, d); else TJ_LAUNCH((k_obs_solve<1>), dim3(n_obs_solve), dim3(64), 0, s, d);
      return true;
    case K_SEP_SELF_ROWS: if (in_graph || in_phase || !multi) return false; TJ_LAUNCH(k_sep_self_rows, dim3(n_rows), dim3(64), 0, s, d); return true;
    case K_MID: if (!in_graph && !in_phase) return false;
      if (in_graph && c->fa_mid_now) {   // asynchronous front, small grids: this launch starts while the iteration's k_front (pairing fa_seq) still runs -- its solve waves wait for it themselves (+ the watcher block)
        c->fa_mid_now = false;
        d_.fa_seq = c->fa_seq; d_.fa_mid = 1;
        if (tri) TJ_LAUNCH((k_mid<3, true>), dim3(1 + n_mid_slack + n_solve + n_obs_solve), dim3(64), 0, s, d, n_solve, n_obs_solve); else TJ_LAUNCH((k_mid<1, true>), dim3(1 + n_mid_slack + n_solve + n_obs_solve), dim3(64), 0, s, d, n_solve, n_obs_solve);
        return true;
      }
      if (tri) TJ_LAUNCH((k_mid<3>), dim3(n_mid_slack + n_solve + n_obs_solve), dim3(64), 0, s, d, n_solve, n_obs_solve); else TJ_LAUNCH((k_mid<1>), dim3(n_mid_slack + n_solve + n_obs_solve), dim3(64), 0, s, d, n_solve, n_obs_solve);
      return true;
    case K_SEP_SELF_SOLVE: if (in_graph || !n_solve) return false; TJ_LAUNCH(k_sep_self_solve, dim3(n_solve), dim3(64), 0, s, d); return true;
    case K_KEEP:  // "optimal_plane":1 only; single UAV: a wave per segment, multi UAV: lanes over the switched-on pair slots
      if (!d.optimal_plane || (multi ? false : d.N == 0)) return false;
      TJ_LAUNCH(k_keep, dim3(multi ? 1024 : owned * d.S), dim3(64), 0, s, d, keep2q ? 1 : 0); return true;   // (asynchronous refinement: the new pairs only)
    case K_SEP_SELF_COMPACT: if ((in_graph || in_phase) && c->grad_fold) return false;   // iteration chains (single GPU and sharded phases): folded into k_grad
      TJ_LAUNCH(k_sep_self_compact, dim3(owned * d.S), dim3(64), 0, s, d); return true;
    case K_GRAD:
      if ((in_graph || in_phase) && c->grad_fold) TJ_LAUNCH((k_grad<true>), dim3(owned * d.P), dim3(GRAD_FOLD_THREADS), c->lds_grad + grad_fold_extra_doubles(d.res) * sizeof(double), s, d);
      else TJ_LAUNCH((k_grad<false>), dim3(owned * d.P), dim3(GRAD_THREADS), c->lds_grad, s, d);
      return true;
    case K_XSOLVE: {
      Dev dx = d;
      if (!chained) dx.c2_fold = 0;   // (the stage API's solve is followed by k_xsolve_c2)
      const Dev& d = dx;
      if (c->xs_seq_gated != c->xs_seq) {   // asynchronous solve: on the second queue, behind a gate that this iteration's k_grad opens (profiling, graphs: it simply follows k_grad on this queue)
        c->xs_seq_gated = c->xs_seq;
        s = c->stream2;
        TJ_LAUNCH(k_xs_gate, dim3(1), dim3(64), 0, s, d, c->xs_seq, (c->xs_fault > 0 && c->xs_seq == c->xs_fault) ? 1 : 0);
      }
      if (d.xs_band) TJ_LAUNCH(k_xsolve_band, dim3(owned), dim3(XB_THREADS), c->lds_xs, s, d);
      else switch (9 * d.P - 2) {   // the register factorisation is inlined per size (kernels_newton.h); 61 rows and the LDS forms: the generic kernel
        case 16: TJ_LAUNCH((k_xsolve<16>), dim3(owned), dim3(XS_LOAD_THREADS), c->lds_xs, s, d); break;
        case 25: TJ_LAUNCH((k_xsolve<25>), dim3(owned), dim3(XS_LOAD_THREADS), c->lds_xs, s, d); break;
        case 34: TJ_LAUNCH((k_xsolve<34>), dim3(owned), dim3(XS_LOAD_THREADS), c->lds_xs, s, d); break;
        case 43: TJ_LAUNCH((k_xsolve<43>), dim3(owned), dim3(XS_LOAD_THREADS), c->lds_xs, s, d); break;
        case 52: TJ_LAUNCH((k_xsolve<52>), dim3(owned), dim3(XS_LOAD_THREADS), c->lds_xs, s, d); break;
        default: TJ_LAUNCH((k_xsolve<0>), dim3(owned), dim3(XS_LOAD_THREADS), c->lds_xs, s, d); break;
      }
      if (chained && d.fuse && !d.xs_band) c->ccd_valid = true;   // its tail has left the owned robots' swept-hull cache
      return true;
    }
    case K_XSOLVE_C2:
      if (coupled && chained && d.c2_fold) return false;   // k_xsolve has finished the arrowhead solve itself
      if (coupled) { if (d.xs_band) TJ_LAUNCH(k_xsolve_c2_band, dim3(owned), dim3(XS_THREADS), c->lds_xs2, s, d); else TJ_LAUNCH(k_xsolve_c2, dim3(owned), dim3(XS_THREADS), c->lds_xs2, s, d); }
      return coupled;
    case K_CCD_PREP: if (chained && d.xf_all) return false;                                // coupled chain: k_ccd's units build every robot's record
      if (chained && d.fuse && !d.xs_band && c->ccd_valid) return false;  // fused chains: k_xsolve's tail leaves the swept-hull cache
      if (chained && d.xf) { if (owned > 0) TJ_LAUNCH(k_ccd_prep, dim3(owned * d.S), dim3(64), 0, s, d, d.u0); }   // (the other ranks' robots: k_ccd's foreign units)
      else TJ_LAUNCH(k_ccd_prep, dim3(d.U * d.S), dim3(64), 0, s, d, 0);
      return true;
    case K_CCD: if (!in_graph && !in_phase) return false;
      if (c->split_unions && multi) {
        if (owned * d.S > 0) { if (tri) TJ_LAUNCH((k_ccd_obs<3>), dim3(owned * d.S), dim3(64), 0, s, d); else TJ_LAUNCH((k_ccd_obs<1>), dim3(owned * d.S), dim3(64), 0, s, d); }
        TJ_LAUNCH(k_ccd_self_pairs, dim3(n_rows), dim3(64), 0, s, d);
        return true;
      }
      if (d.xf) { if (c->xf_used[1]) (void)hipMemsetAsync(d.xf_seg + (size_t)d.S * XF_SEG_STRIDE, 0, (size_t)d.S * XF_SEG_STRIDE * sizeof(int), s); c->xf_used[1] = true; }
      if (d.xch && c->xch_wait_kernel) TJ_LAUNCH(k_xch_wait, dim3(1), dim3(64), 0, s, d, 1);   // ranks sharing a device: the wait for the peers' direction records is a launch of its own
      {
        const int g = n_ccd + n_xf + (d.seq_fold ? 1 : 0);   // + the finisher of the folded pair replay (kernels_step.h)
        if (c->ccd_lean) { if (tri) TJ_LAUNCH((k_ccd_lean<3>), dim3(g), dim3(64), 0, s, d); else TJ_LAUNCH((k_ccd_lean<1>), dim3(g), dim3(64), 0, s, d); }
        else { if (tri) TJ_LAUNCH((k_ccd<3>), dim3(g), dim3(64), 0, s, d); else TJ_LAUNCH((k_ccd<1>), dim3(g), dim3(64), 0, s, d); }
      }
      return true;
    case K_CCD_OBS: if (in_graph) return false;
      if (tri) TJ_LAUNCH((k_ccd_obs<3>), dim3(owned * d.S), dim3(64), 0, s, d); else TJ_LAUNCH((k_ccd_obs<1>), dim3(owned * d.S), dim3(64), 0, s, d);
      return true;
    case K_CCD_SELF_PAIRS: if (in_graph || !multi) return false; TJ_LAUNCH(k_ccd_self_pairs, dim3(n_rows), dim3(64), 0, s, d); return true;
    case K_CCD_SELF_SEQ:
      if ((in_graph || in_phase) && d.seq_fold && !(c->split_unions && multi)) return false;   // the last block of k_ccd has done it
      if (!multi && in_graph) return false;   // single UAV: no pairs to replay, and k_xsolve has left gnorm = |g| itself -- one launch less in the chain
      TJ_LAUNCH(k_ccd_self_seq, dim3(1), dim3(64), c->lds_seq, s, d); return true;
    case K_LINESEARCH: if (!coupled) { c->hull_from_units = false;
      if (fa2q && (chain_pos & 2)) { d_.fa_seq = ++c->fa_seq; d_.fa_units = 1; d_.fa_mid = c->fa_mid_ok ? 1 : 0; c->fa_armed = true; c->hull_from_units = true; }
      else if (in_graph && d_.fa && c->fa_emulate && (chain_pos & 2)) { d_.fa_units = 1; c->hull_from_units = true; }   // the next iteration of the batch follows: its k_front runs next to this launch
      TJ_LAUNCH(k_linesearch, dim3(owned * d.ls_help), dim3(LS_THREADS), c->lds_ls, s, d, c->lsl, (chain_pos & 2) ? 1 : 0); if (chain_pos & 2) c->xf_used[0] = c->xf_used[1] = false; }   // (its last block runs begin_body)
      return !coupled;
    // coupled mode ("decouple":0): evaluation rounds of the summed-energy Armijo search, commit
    case K_LS_COUPLED:
      if (coupled) {
        const int base = (owned != d.U && d.lsc_follow) ? c->lsc_base : 0;   // (a followed search of a sharded context: the rounds beyond the first table)
        c->hull_from_units = false;
        if (fa2q && c->lsc_wide && owned == d.U && (chain_pos & 2)) { d_.fa_seq = ++c->fa_seq; d_.fa_units = 1; d_.fa_mid = c->fa_mid_ok ? 1 : 0; c->fa_armed = true; }   // asynchronous front: the next k_front runs next to this launch
        if (c->lsc_wide) { TJ_LAUNCH(k_ls_coupled, dim3(owned * LSC_ROUNDS), dim3(LS_THREADS), c->lds_ls, s, d, c->lsl, 0, LSC_ROUNDS, (chain_pos & 2) ? 1 : 0, base); if (chain_pos & 2) c->xf_used[0] = c->xf_used[1] = false; }   // all rounds at once, one block per (robot, round)
        else for (int r = 0; r < LSC_ROUNDS; r++) TJ_LAUNCH(k_ls_coupled, dim3(owned), dim3(LS_THREADS), c->lds_ls, s, d, c->lsl, r, 1, 0, base);
      }
      return coupled;
    case K_LS_COMMIT: if (coupled && !(c->lsc_wide && owned == d.U)) TJ_LAUNCH(k_ls_commit, dim3(owned), dim3(64), 0, s, d, (owned != d.U && d.lsc_follow) ? c->lsc_base : 0); return coupled;   // (one context, all rounds in one launch: its last block commits)
    case K_SLACK: if (in_graph) return false; TJ_LAUNCH(k_slack, dim3(owned * d.P), dim3(64), 0, s, d, slack_deferred); return true;
  }
  return false;
}

// enqueue one stage (= one or more kernels) on a stream
int enqueue_stage(tj_ctx* c, int stage, hipStream_t s = nullptr, bool in_graph = false) {
  if (!s) s = c->stream;
  switch (stage) {
    case TJ_STAGE_BEGIN: launch_kernel(c, K_BEGIN, s); break;
    case TJ_STAGE_PLANES_OBS: launch_kernel(c, K_SEP_OBS, s); launch_kernel(c, K_OBS_SOLVE, s); if (c->d.mode == 0) launch_kernel(c, K_KEEP, s); launch_kernel(c, K_SEP_SELF_COMPACT, s); break;
    case TJ_STAGE_PLANES_SELF: launch_kernel(c, K_HULLINFO, s); launch_kernel(c, K_SEP_SELF_ROWS, s); launch_kernel(c, K_SEP_SELF_SOLVE, s); launch_kernel(c, K_KEEP, s); launch_kernel(c, K_SEP_SELF_COMPACT, s); break;
    case TJ_STAGE_GRAD: launch_kernel(c, K_GRAD, s); break;
    case TJ_STAGE_XSOLVE: launch_kernel(c, K_XSOLVE, s); launch_kernel(c, K_XSOLVE_C2, s); break;
    case TJ_STAGE_CCD_PREP: launch_kernel(c, K_CCD_PREP, s); break;
    case TJ_STAGE_CCD_OBS: launch_kernel(c, K_CCD_OBS, s); break;
    case TJ_STAGE_CCD_SELF: launch_kernel(c, K_CCD_SELF_PAIRS, s); launch_kernel(c, K_CCD_SELF_SEQ, s); break;
    case TJ_STAGE_LINESEARCH: launch_kernel(c, K_LINESEARCH, s); launch_kernel(c, K_LS_COUPLED, s); launch_kernel(c, K_LS_COMMIT, s); break;
    case TJ_STAGE_SLACK: launch_kernel(c, K_SLACK, s, 0); break;
    case TJ_STAGE_END: hipLaunchKernelGGL(k_end, dim3(1), dim3(1), 0, s, c->d); break;
    default: c->err = "unknown stage"; return TJ_ERR_INVALID;
  }
  HIPCHK(c, hipGetLastError());
  return TJ_OK;
}

// One iteration of the single-GPU schedule: a LINEAR chain on one stream / hardware queue (captured into the hipGraph)
//   begin -> [hullinfo] -> front{obstacle planes | pair rows} -> mid{slack+dual of the PREVIOUS iteration | pair solves}
//         -> compact -> grad -> xsolve [-> xsolve_c2] -> ccd_prep -> ccd{obstacle CCD | pair CCD selection} -> seq -> line search
// Same-queue successors start back to back, so concurrency between independent stages comes from sharing a launch
// (union kernels), not from parallel streams.  The plane builders only read control points, which the previous line
// search already committed, so that iteration's slack/dual update (touches z, Lambda, t_z, tau only) is deferred into
// k_mid; flush_deferred() pays the last one before anything on the host looks at the state.  The iteration counter is
// committed by the next k_begin.
int enqueue_iteration(tj_ctx* c, int chain_pos = 0) {
  for (int k = 0; k < K_COUNT; k++) launch_kernel(c, k, c->stream, 0, true, false, chain_pos);
  HIPCHK(c, hipGetLastError());
  c->maybe_deferred = true;
  return TJ_OK;
}

int ensure_hull_cache(tj_ctx* c);

// Pay a deferred slack/dual update (no-op kernels if nothing is owed).
int flush_deferred(tj_ctx* c) {
  if (!c->maybe_deferred) return TJ_OK;
  const Dev& d = c->d;
  TJ_LAUNCH(k_flush, dim3(1), dim3(1), 0, c->stream, d, c->begin_folded ? 1 : 0);   // (a begin folded into the last line search whose iteration was never enqueued is taken back)
  c->begin_folded = false;
  launch_kernel(c, K_SLACK, c->stream, 1);
  HIPCHK(c, hipGetLastError());
  c->maybe_deferred = false;
  return TJ_OK;
}

int heal_check(tj_ctx* c, int err_known);
// drain: pay a deferred slack/dual update, then wait for every queue of the context
#define QUIESCE_NOHEAL(c)                                       \
  do {                                                          \
    int qr_ = flush_deferred(c);                                \
    if (qr_) return qr_;                                        \
    HIPCHK(c, hipStreamSynchronize((c)->stream));               \
    if ((c)->stream2) HIPCHK(c, hipStreamSynchronize((c)->stream2)); \
    if ((c)->stream3) HIPCHK(c, hipStreamSynchronize((c)->stream3)); \
  } while (0)
// every host-visible read or write of solver state first drains the context -- and, where a batch ran on several queues, looks whether it has to be run again (heal_check:
// one 4-byte read-back; tj_sync alone does not look -- whoever reads a result afterwards does; tj_iterate uses the control block it reads anyway)
#define QUIESCE(c)                                              \
  do {                                                          \
    QUIESCE_NOHEAL(c);                                          \
    if ((c)->snap_iters > 0 && !(c)->heal_busy) { int hr_ = heal_check(c, -1); if (hr_) return hr_; } \
  } while (0)

// Work of graph slot `which`: 0,1,2 = the phases of a sharded iteration (split at the two all-gathers), 3 = one full
// iteration.  The phases are linear chains on the context's stream as well and reuse the union kernels where the
// stages they join fall into the same phase (k_mid, k_ccd); the slack/dual update is the deferred one inside k_mid.
//   phase 0: begin (stop test)                                                            -> all-gather control points
//   phase 1: hull cache (ALL robots), k_front {obstacle query | pair rows}, k_mid, compaction, gradient, Newton solve -> all-gather directions
//   phase 2: swept-hull cache (ALL robots), k_ccd, sequential pair clamp + gnorm, line search
int enqueue_body(tj_ctx* c, int which, int chain_pos = 0, bool whole_iteration = false) {
  if (whole_iteration) return enqueue_iteration(c, chain_pos);
  hipStream_t m = c->stream;
  // decoupled / single: 3 phases.  coupled ("decouple":0): 6 phases -- the arrowhead system, the shared CCD step and the
  // Armijo test on the summed energy each need something from every robot (tj_iterate_phase, trajadmm.h)
  static const int ph0[] = {K_BEGIN}, ph1[] = {K_HULLINFO, K_FRONT, K_MID, K_KEEP, K_SEP_SELF_COMPACT, K_GRAD, K_XSOLVE},
                   ph2[] = {K_CCD_PREP, K_CCD, K_CCD_SELF_SEQ, K_LINESEARCH},
                   pc2[] = {K_XSOLVE_C2}, pc3[] = {K_CCD_PREP, K_CCD, K_CCD_SELF_SEQ}, pc4[] = {K_LS_COUPLED}, pc5[] = {K_LS_COMMIT};
  const bool cpl = c->d.mode == TJ_MODE_MULTI_COUPLED;
  const int* list = nullptr; int n = 0;
  switch (which) {
    case 0: list = ph0; n = 1; break;
    case 1: list = ph1; n = 7; break;
    case 2: if (cpl) { list = pc2; n = 1; } else { list = ph2; n = 4; } break;
    case 3: list = pc3; n = 3; break;
    case 4: list = pc4; n = 1; break;
    case 5: list = pc5; n = 1; break;
  }
  if (which == 1 && c->d.fuse) { int hr = ensure_hull_cache(c); if (hr) return hr; }   // fused phases: k_linesearch keeps the owned robots' hull cache; after a host write it is rebuilt once
  for (int i = 0; i < n; i++) launch_kernel(c, list[i], m, 0, false, true, chain_pos);
  HIPCHK(c, hipGetLastError());
  // this iteration's slack/dual update is owed to the next k_mid (or the flush) -- in a followed coupled search only once the search is over (tj_coupled_search_pending says so):
  // a flush between two tables of candidates would pay the update on the uncommitted control net
  if (which == (cpl ? 5 : 2) && !(cpl && c->d.lsc_follow && c->d.u1 - c->d.u0 != c->d.U)) c->maybe_deferred = true;
  return TJ_OK;
}

// Capture the body once into a hipGraph and replay it; fall back to eager launches if capture is
// not possible on this stream.
int launch_graph_or_eager(tj_ctx* c, int which, int chain_pos = 0) {
  // Default: plain launches.  The iteration is a linear chain on one queue and the host enqueues far ahead of the device
  // (10 launches ~ 35 us of host time per ~200 us iteration), so consecutive kernels already start back to back; a
  // hipGraph replay of the same chain measured 4 us SLOWER per iteration (~8 us between consecutive graph launches),
  // and ten iterations per graph 6 us slower still.  TJ_USE_GRAPH=1 selects the captured-graph replay.
  if (!c->use_graph) {
    int r = enqueue_body(c, which, chain_pos, which == 3);
    if (r == TJ_OK && which >= 3) c->maybe_deferred = true;
    return r;
  }

  if (!c->graph_ok[which] && !c->graph_failed[which]) {
    hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
      int r = enqueue_body(c, which, 0, which == 3);
      hipGraph_t g = nullptr;
      e = hipStreamEndCapture(c->stream, &g);
      if (r == TJ_OK && e == hipSuccess && g && hipGraphInstantiate(&c->gexec[which], g, nullptr, nullptr, 0) == hipSuccess) { c->graph[which] = g; c->graph_ok[which] = true; }
      else { if (g) hipGraphDestroy(g); (void)hipGetLastError(); c->graph_failed[which] = true; }
    } else { (void)hipGetLastError(); c->graph_failed[which] = true; }
  }
  if (c->graph_ok[which]) {
    HIPCHK(c, hipGraphLaunch(c->gexec[which], c->stream));
    if (which == 3) c->maybe_deferred = true;  // a replayed iteration leaves its slack/dual update owed
    return TJ_OK;
  }
  return enqueue_body(c, which, 0, which == 3);
}

// obstacle-CCD candidates per iteration since the last look -> build of k_ccd for what comes next (kernels_step.h)
void choose_builds(tj_ctx* c, const int* found64) {
  unsigned tot = 0;
  for (int i = 0; i < 64; i++) tot += (unsigned)found64[i];
  const long long di = c->iters_enqueued - c->iters_seen;
  // (rounds 2 - 4 picked the per-lane build of k_ccd where many candidates reach the GJK; since the lean build takes them cooperatively (round 5: no spills) it is the
  //  faster one on every scene measured -- hard 8-robot fleet 43.4 -> 34.3 us, 64 robots through 1 M points 31.3 -> 29.5, SCN-A 17.4 -> 16.6 -- and stays selected;
  //  TJ_CCD_LEAN=0 launches the per-lane build)
  (void)di; (void)tot;
  c->ccd_found_seen = tot; c->iters_seen = c->iters_enqueued;
}

int check_device_errors(tj_ctx* c, Ctl* out = nullptr) {
  Ctl h; int found64[64];
  const int fb0 = c->async_fallbacks;
  { int fr_ = flush_deferred(c); if (fr_) return fr_; }   // (in front of the copies: they are to see the state behind the last slack / dual update)
  HIPCHK(c, hipMemcpyAsync(&h, c->d.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(found64, c->d.ccd_found, sizeof(found64), hipMemcpyDeviceToHost, c->stream));
  QUIESCE_NOHEAL(c);
  if (c->snap_iters > 0 && !c->heal_busy) { int hr_ = heal_check(c, h.error); if (hr_) return hr_; }   // (the error word has come with the control block: no extra read-back)
  if (c->async_fallbacks != fb0) {   // the batch was run again on one queue (heal_check): what was copied above belongs to the abandoned attempt
    HIPCHK(c, hipMemcpy(&h, c->d.ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(found64, c->d.ccd_found, sizeof(found64), hipMemcpyDeviceToHost));
  }
  if (out) *out = h;
  choose_builds(c, found64);
  if (h.error & ERR_PEER_TIMEOUT) { c->err = "tj_group: a peer rank's slice did not arrive within 2 s (flag transport); the group must be re-initialised"; return TJ_ERR_DEVICE; }
  if (h.error & (ERR_PLANE_OVERFLOW | ERR_FRONT_OVERFLOW | ERR_PAIR_OVERFLOW)) {
    c->err = "device list overflow (error bits " + std::to_string(h.error) + "): raise cap_obs/cap_self/cap_pairs";
    return TJ_ERR_CAPACITY;
  }
  if (h.error & ERR_LOOP_CAP) {
    c->err = "a device back-off/Newton/Armijo loop hit its cap (infeasible state)";
    if (h.error & ERR_LS_RANGE) c->err += ": coupled mode on a SHARDED context, no acceptable step among the 31 Armijo back-offs its exchange carries (one context follows the search to the reference's own end)";
    if (h.error & ERR_CCD_STUCK) c->err += ": a CCD clamp found contact at every step (the state itself is in collision; the reference loops forever here)";
    if (h.error & ERR_SLACK_ARMIJO) c->err += ": the slack update's Armijo search";
    if (h.error & ERR_PLANE_REFINE) c->err += ": optimal_plane, a plane refinement did not terminate within its caps";
    if (h.error & ERR_XS_TIMEOUT) c->err += ": NOT an infeasible state -- a wait between the queues of the context (asynchronous Newton solve / plane refinement) ran out after 2 s (GPU shared with other processes?); TJ_XS_ASYNC=0 TJ_KEEP_ASYNC=0 keep everything on the chain's queue";
    if (h.error & ERR_PASS_TIMEOUT) c->err += ": NOT an infeasible state -- a wave waiting for passed-on robot pairs timed out after 5 ms (GPU queue descheduled / shared with other processes); re-run the iteration or set TJ_PAIR_PASS_ON=0";
    return TJ_ERR_NO_PROGRESS;
  }
  if (h.order_unresolved) { c->err = "inter-robot CCD clamp: two acting pairs of a segment share a robot and the reference's pair order could not be replayed (fleet too large for the LDS-resident tree)"; return TJ_ERR_UNSUPPORTED; }
  if (h.error & ERR_NOT_SPD) { c->err = "coupled mode: the arrowhead Newton system is not positive definite (the reference has no fallback either)"; return TJ_ERR_NO_PROGRESS; }
  return TJ_OK;
}

// Dev::fuse: the iteration graph has no k_hullinfo (k_linesearch leaves the next iteration's hull cache); after the
// control points were set from the host the cache is rebuilt once here.
int ensure_hull_cache(tj_ctx* c) {
  if (!c->d.fuse || c->hull_valid) return TJ_OK;
  launch_kernel(c, K_HULLINFO, c->stream);
  HIPCHK(c, hipGetLastError());
  c->hull_valid = true; c->hull_from_units = false;
  return TJ_OK;
}

// The batch enqueued since the last snapshot is through (every queue drained).  No incident: forget the snapshot.  ERR_XS_TIMEOUT: one queue from now on, the snapshot's
// state back in place, the same iterations again.
int heal_check(tj_ctx* c, int err_known) {
  int err = err_known;
  if (err < 0) HIPCHK(c, hipMemcpy(&err, &c->d.ctl->error, sizeof(int), hipMemcpyDeviceToHost));   // (every queue has drained: QUIESCE)
  const long long n = c->snap_iters;
  c->snap_iters = 0;
  if (!(err & ERR_XS_TIMEOUT)) return TJ_OK;
  c->heal_busy = true;
  c->async_fallbacks++;
  if (c->hwq_claim) { g_async_queues[std::min(std::max(c->prm.device, 0), 63)].fetch_sub(c->hwq_claim); c->hwq_claim = 0; }   // (one queue from now on: the budget is free for another context)
  c->xs_two_queues = false; c->keep_two_queues = false; c->fa_armed = false; c->fa_mid_now = false; c->xs_fault = 0;   // (the tickets / flags of the asynchronous solve work on one queue as well: TJ_XS_ONE_QUEUE's schedule)
  Dev& d = c->d;
  hipLaunchKernelGGL(k_snapshot, dim3(64, std::max(c->snap_n, 1)), dim3(256), 0, c->stream, c->snap_tab, c->snap_n, 1, d.ctl, c->ctl_snap);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemsetAsync(d.xs_sync, 0, ((size_t)2 * d.U + 2) * 32 * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(d.fa_sync, 0, Dev::fa_sync_ints(d.U) * sizeof(int), c->stream)); c->fa_seq = 0;
  HIPCHK(c, hipMemsetAsync(d.keep_sync, 0, 17 * 32 * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(d.xf_seg, 0, (size_t)2 * d.S * XF_SEG_STRIDE * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(d.spec_n, 0, 8, c->stream));   // (head-start lists of the abandoned iterations: launch shape only, dropped)
  c->hull_valid = false; c->ccd_valid = false; c->begin_folded = false; c->maybe_deferred = false;   // (what the snapshot's control block owes is owed again: the next k_mid / flush pays it)
  c->xs_seq = c->xs_seq_gated = 0; c->keep_seq = 0;
  int r = tj_iterate_async(c, (int)n);
  if (r == TJ_OK) r = flush_deferred(c);
  if (r == TJ_OK) { HIPCHK(c, hipStreamSynchronize(c->stream)); if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2)); if (c->stream3) HIPCHK(c, hipStreamSynchronize(c->stream3)); }
  c->snap_iters = 0;
  c->heal_busy = false;
  return r;
}

bool ready(tj_ctx* c) {
  if (!c->have_cloud) { c->err = "tj_set_cloud has not been called"; return false; }
  if (!c->have_state) { c->err = "tj_init_state has not been called"; return false; }
  return true;
}

}  // namespace

extern "C" {

void tj_default_params(tj_params* p, int mode, int uav_num, int piece_num) {
  memset(p, 0, sizeof(*p));
  p->mode = mode; p->uav_num = uav_num; p->piece_num = piece_num; p->res = 8;
  p->lambda = 10.0; p->margin = 0.1; p->offset = 0.1; p->mu = 0.1; p->vel_limit = 2.0; p->acc_limit = 2.0;
  p->ks = mode == TJ_MODE_SINGLE ? 1e-8 : 1e-3;  /* Main/admmPathPlanning3D.cpp:477, Main/multiPathPlanning3D.cpp:596 */ p->kt = 1.0; p->stop = 1e-2;
  p->device = 0; p->rank = 0; p->world = 1;
}

#ifdef TJ_PHASE_TIMING
// development build only (make timing): wall-clock stamps (10 ns ticks) the kernels left at their
// phase boundaries during the most recent launch; out is [K_COUNT][TJ_TIC_BLOCKS][TJ_TIC_SLOTS]
int tj_debug_phase_times(tj_ctx* c, long long* out) {
  if (!c || !out) return TJ_ERR_INVALID;
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(out, c->d.dbg, sizeof(long long) * K_COUNT * TJ_TIC_BLOCKS * TJ_TIC_SLOTS, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemsetAsync(c->d.dbg, 0, sizeof(long long) * K_COUNT * TJ_TIC_BLOCKS * TJ_TIC_SLOTS, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return K_COUNT;
}
#endif

int tj_host_tables(int piece_num, int res, double* convert, double* mdyn, double* basis, double* kdop) {
  if (piece_num < 1 || res < 1) return TJ_ERR_INVALID;
  HostTables t;
  build_tables(piece_num, res, 1, t);
  if (convert) memcpy(convert, t.convert.data(), t.convert.size() * 8);
  if (mdyn) memcpy(mdyn, t.mdyn, 36 * 8);
  if (basis) memcpy(basis, t.basis.data(), t.basis.size() * 8);
  if (kdop) memcpy(kdop, t.kdop, 147 * 8);
  return TJ_OK;
}

const char* tj_last_error(const tj_ctx* c) { return c ? c->err.c_str() : "null context"; }

int tj_create(const tj_params* p, tj_ctx** out) {
  if (!p || !out) return TJ_ERR_INVALID;
  *out = nullptr;
  tj_ctx* c = new tj_ctx();
  *out = c;  // returned even on failure so the caller can read tj_last_error()
  c->prm = *p;
  if (p->uav_num < 1 || p->piece_num < 2 || p->res < 1 || p->mode < 0 || p->mode > 2 || p->world < 1 || p->rank < 0 || p->rank >= p->world) {
    c->err = "invalid tj_params (need uav_num>=1, piece_num>=2, res>=1, mode 0/1/2, 0<=rank<world)";
    return TJ_ERR_INVALID;
  }
  if (p->mode == TJ_MODE_SINGLE && p->uav_num != 1) { c->err = "TJ_MODE_SINGLE requires uav_num == 1"; return TJ_ERR_INVALID; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { c->err = "no HIP device available (this library has no CPU fallback)"; return TJ_ERR_DEVICE; }
  HIPCHK(c, hipSetDevice(p->device));
  HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  Dev& d = c->d;
  memset(&d, 0, sizeof(d));
  d.mode = p->mode; d.U = p->uav_num; d.P = p->piece_num; d.res = p->res; d.S = d.P * d.res; d.T = 3 * d.P + 3; d.N = 0; d.prim = 1;
  c->use_graph = tune("USE_GRAPH") != nullptr;
  // the fused chain (k_linesearch leaves the hull cache, k_xsolve's tail the swept-hull cache) -- sharded contexts too since round 5: the other ranks' robots
  // are handled by foreign units inside k_front / k_ccd (Dev::xf).  Coupled mode keeps its own kernels (and, sharded, k_hullinfo / k_ccd_prep for all robots).
  const bool split_env = tune("SPLIT_UNIONS") && atoi(tune("SPLIT_UNIONS")) != 0;
  d.fuse = (p->mode != TJ_MODE_MULTI_COUPLED && !(p->world > 1 && split_env)) ? 1 : 0;
  d.rank = p->rank; d.world = p->world;
  d.xf = (p->world > 1 && d.fuse && p->mode == TJ_MODE_MULTI_DECOUPLE) ? 1 : 0;
  // coupled mode, one context: every robot's cache records by units inside k_front / k_ccd (two launches less per iteration; TJ_COUPLED_UNITS=0: k_hullinfo / k_ccd_prep)
  if (p->mode == TJ_MODE_MULTI_COUPLED && p->world == 1 && !split_env && !(tune("COUPLED_UNITS") && atoi(tune("COUPLED_UNITS")) == 0)) { d.xf = 1; d.xf_all = 1; }
  d.u0 = (int)((long long)p->rank * d.U / p->world); d.u1 = (int)((long long)(p->rank + 1) * d.U / p->world);
  d.lambda = p->lambda; d.margin = p->margin; d.offset = p->offset; d.mu = p->mu; d.vel_limit = p->vel_limit; d.acc_limit = p->acc_limit;
  d.ks = p->ks; d.kt = p->kt; d.stop = p->stop;
  d.cap_obs = p->cap_obs > 0 ? p->cap_obs : 256;
  d.cap_self = p->cap_self > 0 ? p->cap_self : std::max(1, std::min(d.U - 1, 64));  // neighbours within offset + 2 margin of ONE segment; k_grad's LDS grows with it
  d.cap_pairs = p->cap_pairs > 0 ? p->cap_pairs : d.U;
  d.optimal_plane = p->optimal_plane ? 1 : 0;
  d.pair_rows = d.U <= 128 ? 8 : 16;   // tile height: 64 robots -- 8 rows: k_front 13.3 -> 12.5 us, k_ccd 9.6 -> 8.7 (with eight interval records in flight); 256 robots -- 16 rows (8: +1.3 us, 4: +13)
  if (const char* e = tune("PAIR_ROWS")) { const int r = atoi(e); if (r == 2 || r == 4 || r == 8 || r == 16) d.pair_rows = r; }
  d.cap_work = d.mode >= 1 ? (int)std::min<long long>((long long)d.S * d.U * (d.U - 1) / 2 + 1, 1 << 22) : 1;  // robot pairs per iteration
  d.xs = 3 * d.T + 4;
  const int n = 9 * d.P - 2;
  d.grad_npl = std::min(d.cap_obs + d.cap_self, 64);   // what a batch of segments really carries (SCN-C: <= 40); more goes through grad_scr.  64: five workgroups per CU (96: four)
  if (const char* e = tune("GRAD_NPL")) { const int r = atoi(e); if (r >= 8 && r <= 4096) d.grad_npl = std::min(d.cap_obs + d.cap_self, r); }
  c->lds_grad = grad_lds_doubles(d.grad_npl, d.res) * sizeof(double);
  const size_t lds_max = 160 * 1024 - 1024;
  // long trajectories: the dense per-robot system no longer fits LDS -> band storage (decoupled / single-UAV modes)
  d.xs_band = (xsolve_lds_doubles(n) * sizeof(double) > lds_max || tune("XS_BAND")) ? 1 : 0;
  c->lds_xs = (d.xs_band ? xsolve_band_lds_doubles(n) : xsolve_lds_doubles(n)) * sizeof(double);
  c->lds_xs2 = (d.xs_band ? (size_t)(n - 1) * BAND_BS + 5 * (size_t)n : (size_t)n * n + 4 * (size_t)n) * sizeof(double);   // k_xsolve_c2 / k_xsolve_c2_band
  c->lsl = ls_layout(d.S, d.T, d.P, 120 * 1024);
  c->lds_ls = c->lsl.total * sizeof(double);
#ifdef TJ_PHASE_TIMING
  { int r_ = dalloc(c, &d.dbg, (size_t)K_COUNT * TJ_TIC_BLOCKS * TJ_TIC_SLOTS); if (r_) return r_; }
#endif
  if (d.U > 2048) { c->err = "more than 2048 robots are not supported (pair keys pack robot ids into 11 bits; the dense [S][U][U] plane tables are 5.4 GB + 0.7 GB there)"; return TJ_ERR_UNSUPPORTED; }
  if (d.S > 511) { c->err = "more than 511 segments per robot are not supported by the line-search kernel"; return TJ_ERR_UNSUPPORTED; }
  if (d.res > GRAD_MAXRES) { c->err = "res > 16 segments per piece is not supported by the gradient kernel"; return TJ_ERR_UNSUPPORTED; }
  d.seq_tree = (d.mode == TJ_MODE_MULTI_DECOUPLE && seq_lds_bytes(d.U, d.S, true) <= lds_max) ? 1 : 0;
  if (tune("NO_SEQ_TREE")) d.seq_tree = 0;  // test hook: behave like a fleet too large for the LDS-resident tree
  c->lds_seq = seq_lds_bytes(d.U, d.S, d.seq_tree != 0);
  c->split_unions = false;
  if (const char* e = tune("CCD_LEAN")) c->ccd_lean = atoi(e) != 0;
  // hundreds of robots: the 512-thread folded k_grad is limited to ~2 workgroups per CU by wave slots; the 192-thread one (5 per CU)
  // plus a separate compaction launch is faster once there are more pieces than that (SCN-D: k_grad 109 -> 72 + 14 us)
  c->grad_fold = (d.u1 - d.u0) * d.P <= 512;
  if (const char* e = tune("GRAD_FOLD")) c->grad_fold = atoi(e) != 0;
  d.bvh_skip = 0;    // decided when the obstacle set is known (set_obstacles); TJ_BVH_SKIP=0 / 1 forces it (launch-shape switch, same bits)
  if (const char* e = tune("BVH_SKIP")) d.bvh_skip = atoi(e) != 0;
  d.pair_prio = 1;
  d.mid_order = (d.mode >= 1 && d.U >= 192) ? 1 : 0;   // k_mid's grid order (kernels_step.h): config 5 -15 us; small fleets: nothing or slightly worse
  if (const char* e = tune("MID_ORDER")) d.mid_order = atoi(e) != 0;   // launch-shape switch (same bits)
  if (const char* e = tune("PAIR_PRIO")) d.pair_prio = atoi(e) != 0;   // launch-shape switch (same bits)
  d.pair_lpw = 64;
  if (const char* e = tune("PAIR_LPW")) { const int r = atoi(e); if (r == 8 || r == 16 || r == 32 || r == 64) d.pair_lpw = r; }   // launch-shape switch (same bits)
  d.pair_pass_on = 1;
  if (const char* e = tune("PAIR_PASS_ON")) d.pair_pass_on = atoi(e) != 0;
  if (const char* e = tune("SPLIT_UNIONS")) c->split_unions = atoi(e) != 0;
  // GJK head start for last iteration's slow robot pairs (kernels_pairs.h: spec_pair_body); TJ_PAIR_HEAD_START=0 switches it off (test hook: same bits)
  d.spec = (d.mode >= 1 && !d.optimal_plane && !c->split_unions) ? 1 : 0;
  if (const char* e = tune("PAIR_HEAD_START")) d.spec = d.spec && atoi(e) != 0;
  // k_ccd's last block finishes with the sequential pair replay + gnorm (kernels_step.h): decoupled mode, when the replay's small
  // arrays fit k_ccd's static LDS buffer with room for at least 256 acting-pair keys (the value is that capacity)
  d.seq_fold = 0;
  if ((d.mode == TJ_MODE_MULTI_DECOUPLE || (d.mode == TJ_MODE_MULTI_COUPLED && p->world == 1)) && !c->split_unions) {   // (coupled: one context only -- a sharded one exports its obstacle-CCD exponents from k_ccd_self_seq)
    const size_t buf = sizeof(double) * (size_t)(CCD_LDS_DOUBLES > PAIR_LDS_DOUBLES ? CCD_LDS_DOUBLES : PAIR_LDS_DOUBLES);
    int cap = 4096;
    while (cap >= 256 && seq_fold_lds_bytes(d.U, cap) > buf) cap >>= 1;
    if (cap >= 256 && (size_t)d.S * pair_units(d.U, d.pair_rows) < 65536) d.seq_fold = cap;   // (the finisher counts the selection blocks in 16 bits)
  }
  if (const char* e = tune("SEQ_FOLD")) if (atoi(e) == 0) d.seq_fold = 0;   // launch-shape switch (same bits)
  c->n_solve_env = 0;
  if (const char* e = tune("N_SOLVE")) c->n_solve_env = std::max(1, atoi(e));
  d.spec_budget = SPEC_GJK_BUDGET; d.spec_min = SPEC_GJK_MIN;
  if (const char* e = tune("HS_BUDGET")) d.spec_budget = std::max(1, atoi(e));   // development hooks (same bits for any value)
  if (const char* e = tune("HS_MIN")) d.spec_min = std::max(1, atoi(e));
  d.ls_fast = 1;
  if (const char* e = tune("LS_FAST")) d.ls_fast = atoi(e) != 0;   // launch-shape switch (same bits): round 0 of k_linesearch in the team shape
  {   // helper blocks of k_linesearch: one CU each, so as many per robot as the device has compute units to spare (64 robots on 256 CUs: 4)
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, p->device));
    const int owned = std::max(1, d.u1 - d.u0);
    d.num_cu = prop.multiProcessorCount;
    // coupled mode: the four evaluation rounds of the Armijo search in one launch where a block per (robot, round) gets a compute unit of its own
    c->lsc_wide = p->mode == TJ_MODE_MULTI_COUPLED && owned * LSC_ROUNDS <= d.num_cu;
    // ... and the corner solve inside k_xsolve where every robot's block is resident at once (one block per compute unit: 242 registers x 8 waves), dense storage
    d.c2_fold = (p->mode == TJ_MODE_MULTI_COUPLED && p->world == 1 && d.U <= d.num_cu && !d.xs_band) ? 1 : 0;
    if (const char* e = tune("C2_FOLD")) d.c2_fold = d.c2_fold && atoi(e) != 0;   // launch-shape switch (same bits)
    if (const char* e = tune("LSC_WIDE")) c->lsc_wide = atoi(e) != 0;   // launch-shape switch (same bits)
    // k_grad's launch order follows the items' last durations where blocks outnumber the compute units (kernels_newton.h: grad_order_body)
    // -- between one and two blocks per unit, the case it was measured on: SCN-C -1.5 us per iteration, the 64 hard robots -1.4; at five blocks per unit
    // (256 robots) longest-first ordering bought nothing in k_grad and the run was 1.5 % slower, so larger fleets keep the identity
    d.grad_bal = (owned * d.P > d.num_cu && owned * d.P < 2 * d.num_cu) ? 1 : 0;
    // asynchronous Newton solve (dev_common.h, Dev::xs_async): one context, decoupled / single-UAV chain with the swept-hull tail in k_xsolve.  TJ_XS_ASYNC=0: the
    // solve stays a link of the one-queue chain (launch-shape switch: same bits)
    d.xs_async = (p->world == 1 && !d.xs_band && (p->mode != TJ_MODE_MULTI_COUPLED ? d.fuse != 0 : (d.c2_fold && d.xf_all))) ? 1 : 0;   // (coupled chain: with the corner solve in k_xsolve and k_ccd's units building the records already;
                                                                                                                                               //  sharded contexts keep the one-queue chain: tried in round 5, a tj_group of two ranks aborted -- not pursued)
    if (d.xs_async) {
      // Liveness: k_xsolve's blocks hold registers and LDS while they sleep on their tickets, and the k_grad blocks that hand the tickets out may still be waiting
      // for a compute unit.  Safe when the sleepers can never shut k_grad out: at most half as many robots as compute units (half the device stays free whatever
      // the dispatcher does), or at most one robot per unit AND a k_grad block fits a unit next to one k_xsolve block (a unit with two sleepers then implies a
      // unit with none).  Larger fleets keep the solve on the chain's queue (1 500 robots: the sleepers filled the device and every wait ran into its 5 ms limit).
      bool fits = false;
      {
        const void* fx = nullptr;
        switch (9 * d.P - 2) {
          case 16: fx = (const void*)k_xsolve<16>; break; case 25: fx = (const void*)k_xsolve<25>; break; case 34: fx = (const void*)k_xsolve<34>; break;
          case 43: fx = (const void*)k_xsolve<43>; break; case 52: fx = (const void*)k_xsolve<52>; break; default: fx = (const void*)k_xsolve<0>; break;
        }
        const void* fg = c->grad_fold ? (const void*)k_grad<true> : (const void*)k_grad<false>;
        hipFuncAttributes ax, ag;
        if (hipFuncGetAttributes(&ax, fx) == hipSuccess && hipFuncGetAttributes(&ag, fg) == hipSuccess) {
          auto gran = [](int r) { return (r + 7) / 8 * 8; };
          const int wx = XS_LOAD_THREADS / 64, wg = (c->grad_fold ? GRAD_FOLD_THREADS : GRAD_THREADS) / 64;
          const size_t lx = c->lds_xs + ax.sharedSizeBytes, lg = c->lds_grad + (c->grad_fold ? grad_fold_extra_doubles(d.res) * sizeof(double) : 0) + ag.sharedSizeBytes;
          fits = ((wx + 3) / 4) * gran(ax.numRegs) + ((wg + 3) / 4) * gran(ag.numRegs) <= 512 && lx + lg <= (size_t)160 * 1024 && (wx + 3) / 4 + (wg + 3) / 4 <= 8;
        } else (void)hipGetLastError();
      }
      if (!(2 * owned <= d.num_cu || (owned <= d.num_cu && fits))) d.xs_async = 0;
    }
    // rocprofv3's counter collection (--pmc) serialises the dispatches of ALL queues, in an order of its own: a gate held back behind the kernel it waits for would run every
    // wait into its 2 s limit.  Under it the context keeps everything on the one queue (an explicit TJ_XS_ASYNC=1 / TJ_KEEP_ASYNC=1 overrides).
    const char* cc_ = getenv("ROCPROF_COUNTER_COLLECTION");
    const bool counters_on = cc_ && cc_[0] && strcmp(cc_, "0") != 0 && strcasecmp(cc_, "false") != 0;
    if (counters_on && !tune("XS_ASYNC")) d.xs_async = 0;
    if (const char* e = tune("XS_ASYNC")) d.xs_async = d.xs_async && atoi(e) != 0;
    // Hardware queues.  HIP maps a process's streams onto GPU_MAX_HW_QUEUES (4) hardware queues per device and lets further streams SHARE them; a gate kernel that sleeps at the
    // head of a shared queue keeps back whatever another context put behind it -- possibly the very kernel a gate of THAT context, asleep on a queue of this one, waits for:
    // measured with three default contexts in one process, two of them ran into the 2 s limit (and healed themselves).  So the contexts of a process that sleep across queues
    // claim their streams (main + second + third) out of a per-device budget of GPU_MAX_HW_QUEUES - 1 (the null stream has one); a context that does not fit keeps the one-queue
    // chain (same bits).  An explicit TJ_XS_ASYNC=1 / TJ_KEEP_ASYNC=1 overrides; self-healing stays the net under it.
    {
      const bool want_keep = d.optimal_plane && p->mode == TJ_MODE_MULTI_DECOUPLE && p->world == 1 && !c->split_unions && !(counters_on && !tune("KEEP_ASYNC")) && !(tune("KEEP_ASYNC") && atoi(tune("KEEP_ASYNC")) == 0);
      const int need = 1 + (d.xs_async ? 1 : 0) + (want_keep ? 1 : 0);
      const bool forced = (tune("XS_ASYNC") && atoi(tune("XS_ASYNC")) != 0) || (tune("KEEP_ASYNC") && atoi(tune("KEEP_ASYNC")) != 0);
      if (need > 1) {
        std::atomic<int>& g = g_async_queues[std::min(std::max(p->device, 0), 63)];
        const int had = g.fetch_add(need);
        if (had + need > hw_queue_budget() && !forced) { g.fetch_sub(need); d.xs_async = 0; c->hwq_refused = true; }
        else c->hwq_claim = need;
      }
    }
    if (d.xs_async) {
      const bool ok = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); c->stream2 = nullptr; }   // (the tickets and flags work on one queue as well)
      c->xs_two_queues = ok && tune("XS_ONE_QUEUE") == nullptr;
    }
    // asynchronous plane refinement ("optimal_plane":1, multi-UAV decoupled mode, one context; TJ_KEEP_ASYNC=0: k_keep stays one launch between k_mid and k_grad -- same bits)
    d.keep_async = (d.optimal_plane && p->mode == TJ_MODE_MULTI_DECOUPLE && p->world == 1 && !c->split_unions && !c->hwq_refused) ? 1 : 0;
    if (counters_on && !tune("KEEP_ASYNC")) d.keep_async = 0;
    if (const char* e = tune("KEEP_ASYNC")) d.keep_async = d.keep_async && atoi(e) != 0;
    d.keep_waves = 1024;
    if (d.keep_async) {
      if (hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) == hipSuccess) c->keep_two_queues = true;
      else { (void)hipGetLastError(); c->stream3 = nullptr; d.keep_async = 0; }
    }
    if (const char* e = tune("GRAD_BALANCE")) d.grad_bal = (atoi(e) != 0 && owned * d.P <= 65536) ? 1 : 0;   // launch-shape switch (same bits)
    d.ls_help = (d.ls_fast && p->mode != TJ_MODE_MULTI_COUPLED) ? std::max(1, std::min(LS_HELP_MAX, prop.multiProcessorCount / owned)) : 1;
    if (const char* e = tune("LS_HELP")) {   // launch-shape switch (same bits); 1 = no helpers.  More blocks per robot than the compute units hold at once would leave helpers waiting for a
                                                  // unit while every primary runs into its 10 us give-up per super-round: clamped to the units the device has
      d.ls_help = std::max(1, std::min(LS_HELP_MAX, atoi(e)));
      d.ls_help = std::min(d.ls_help, std::max(1, prop.multiProcessorCount / owned));
    }
    if (const char* e = tune("LS_HELP_LATE")) d.ls_help_late = std::max(0, std::min(4000, atoi(e)));   // test hook (same bits): helper blocks idle that many microseconds before staging
    if (const char* e = tune("LS_HELP_MUTE")) d.ls_help_mute = atoi(e) != 0;                          // test hook (same bits): the helpers never post, the primaries time out
    // asynchronous front (dev_common.h, Dev::fa): one context, decoupled mode, the asynchronous solve's second queue, and a k_linesearch grid that is resident all at once
    // (one block per compute unit at most -- the residency gate's premise).  TJ_FRONT_ASYNC=0: k_linesearch publishes the hull cache and k_front follows it on the chain's queue (same bits)
    d.fa = (d.xs_async && p->world == 1 && !d.optimal_plane && !c->split_unions && !c->use_graph &&
            (p->mode != TJ_MODE_MULTI_COUPLED ? (d.fuse && owned * d.ls_help <= d.num_cu) : (c->lsc_wide && d.xf_all && owned * LSC_ROUNDS <= d.num_cu))) ? 1 : 0;   // (coupled: the one-launch search, whose last block commits every robot)
    if (const char* e = tune("FRONT_ASYNC")) d.fa = d.fa && atoi(e) != 0;
    c->fa_emulate = tune("FRONT_ASYNC_ONE_QUEUE") && atoi(tune("FRONT_ASYNC_ONE_QUEUE")) != 0;   // the asynchronous front's data flow (k_front's units form the records, k_linesearch publishes none) on the chain's queue: counter passes
    if (d.fa) {
      // Dev::fa_mid: k_mid may start while k_front still runs only if k_front's whole grid is resident before k_mid's first wave is -- the last k_linesearch block waits
      // until every k_front block has started, so the grid must fit the device next to that one block: blocks per compute unit by LDS, registers and wave slots
      const int n_rows = d.S * pair_units(d.U, d.pair_rows);
      const int n_front = owned * d.S + n_rows + (d.spec ? SPEC_CAP : 0) + (d.grad_bal ? (owned * d.P + 63) / 64 : 0);
      hipFuncAttributes af;
      if (hipFuncGetAttributes(&af, (const void*)k_front<1, true>) == hipSuccess) {
        hipFuncAttributes a3; if (hipFuncGetAttributes(&a3, (const void*)k_front<3, true>) == hipSuccess) { af.numRegs = std::max(af.numRegs, a3.numRegs); af.sharedSizeBytes = std::max(af.sharedSizeBytes, a3.sharedSizeBytes); } else (void)hipGetLastError();
        const int by_lds = (int)(((size_t)160 * 1024) / std::max<size_t>(af.sharedSizeBytes, 1)), by_regs = 4 * (512 / std::max((af.numRegs + 7) / 8 * 8, 8)), per_cu = std::min(std::min(by_lds, by_regs), 32);
        c->fa_mid_ok = (long long)n_front <= (long long)(d.num_cu - 1) * per_cu;
      } else (void)hipGetLastError();
      if (const char* e = tune("FRONT_ASYNC_MID")) c->fa_mid_ok = c->fa_mid_ok && atoi(e) != 0;   // launch-shape switch (same bits): 0 = k_linesearch waits for k_front's end, k_mid follows plainly
    }
  }
  if (c->lds_grad + grad_fold_extra_doubles(d.res) * sizeof(double) > lds_max || c->lds_xs > lds_max || c->lds_ls > lds_max || c->lds_seq > lds_max) {
    c->err = "problem does not fit the 160 KB LDS of one CU (segments per robot / fleet size too large for this version)";
    return TJ_ERR_UNSUPPORTED;
  }
  HIPCHK(c, hipFuncSetAttribute((const void*)k_grad<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(c->lds_grad + grad_fold_extra_doubles(d.res) * sizeof(double))));
  HIPCHK(c, hipFuncSetAttribute((const void*)k_grad<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_grad));
  if (d.xs_band) HIPCHK(c, hipFuncSetAttribute((const void*)k_xsolve_band, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_xs));
  else {
    const void* kx = (const void*)k_xsolve<0>;
    switch (n) { case 16: kx = (const void*)k_xsolve<16>; break; case 25: kx = (const void*)k_xsolve<25>; break; case 34: kx = (const void*)k_xsolve<34>; break;
                 case 43: kx = (const void*)k_xsolve<43>; break; case 52: kx = (const void*)k_xsolve<52>; break; }   // 61 rows: the inlined form would spill, the generic kernel calls it out of line
    HIPCHK(c, hipFuncSetAttribute(kx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_xs));
  }
  HIPCHK(c, hipFuncSetAttribute((const void*)k_linesearch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_ls));
  HIPCHK(c, hipFuncSetAttribute((const void*)k_ls_coupled, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_ls));
  if (!d.xs_band) HIPCHK(c, hipFuncSetAttribute((const void*)k_xsolve_c2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_xs2));
  else HIPCHK(c, hipFuncSetAttribute((const void*)k_xsolve_c2_band, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_xs2));
  HIPCHK(c, hipFuncSetAttribute((const void*)k_ccd_self_seq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_seq));

  HostTables t;
  build_tables(d.P, d.res, STEP_CAP, t);
  double *basis, *convert, *mdyn, *kdop, *pow08;
  int r;
  if ((r = dalloc(c, &basis, t.basis.size())) || (r = dalloc(c, &convert, t.convert.size())) || (r = dalloc(c, &mdyn, 36)) ||
      (r = dalloc(c, &kdop, 147)) || (r = dalloc(c, &pow08, t.pow08.size()))) return r;
  if ((r = upload(c, basis, t.basis.data(), t.basis.size() * 8)) || (r = upload(c, convert, t.convert.data(), t.convert.size() * 8)) ||
      (r = upload(c, mdyn, t.mdyn, 36 * 8)) || (r = upload(c, kdop, t.kdop, 147 * 8)) || (r = upload(c, pow08, t.pow08.data(), t.pow08.size() * 8))) return r;
  d.basis = basis; d.convert = convert; d.mdyn = mdyn; d.kdop = kdop; d.pow08 = pow08;
  {   // k_xsolve's overlap-add as a table: per entry of the reduced system the (at most two) piece-block entries that cover it, in piece order; -2: the time-time entry (every piece)
    const int Pn = d.P, m = 9 * Pn - 3, n = m + 1;
    std::vector<int> gt((size_t)2 * n * n + 2 * n, -1);
    auto cover = [&](int g, int& lo, int& hi) { if (g >= 0) { lo = std::max(lo, (g - 17 + 8) / 9); hi = std::min(hi, g / 9); } };
    for (int idx = 0; idx < n * n; idx++) {
      const int ra = idx / n, rb = idx % n, ga = ra == m ? -1 : ra + 6, gb = rb == m ? -1 : rb + 6;
      if (ga < 0 && gb < 0) { gt[2 * (size_t)idx] = -2; continue; }
      int lo = 0, hi = Pn - 1, k = 0;
      cover(ga, lo, hi); cover(gb, lo, hi);
      for (int sp = std::max(lo, 0); sp <= hi; sp++) {
        const int a = ga < 0 ? 18 : ga - 9 * sp, b = gb < 0 ? 18 : gb - 9 * sp;
        if (k < 2) gt[2 * (size_t)idx + k] = sp * 361 + a * 19 + b;
        k++;
      }
      if (k > 2) { c->err = "internal: an entry of the reduced system is covered by more than two piece blocks"; return TJ_ERR_INVALID; }
    }
    for (int ra = 0; ra < n; ra++) {
      const int ga = ra == m ? -1 : ra + 6;
      if (ga < 0) { gt[(size_t)2 * n * n + 2 * ra] = -2; continue; }
      int lo = 0, hi = Pn - 1, k = 0;
      cover(ga, lo, hi);
      for (int sp = std::max(lo, 0); sp <= hi; sp++) { if (k < 2) gt[(size_t)2 * n * n + 2 * ra + k] = sp * 19 + (ga - 9 * sp); k++; }
      if (k > 2) { c->err = "internal: a row of the reduced system is covered by more than two piece blocks"; return TJ_ERR_INVALID; }
    }
    int* gtd = nullptr;
    if ((r = dalloc(c, &gtd, gt.size())) || (r = upload(c, gtd, gt.data(), gt.size() * sizeof(int)))) return r;
    d.xs_gather = gtd;
  }
  const size_t U = d.U, S = d.S, P = d.P, T = d.T;
  if ((r = dalloc(c, &d.spline, U * 3 * T)) || (r = dalloc(c, &d.p_slack, U * 18 * P)) || (r = dalloc(c, &d.p_lambda, U * 18 * P)) ||
      (r = dalloc(c, &d.t_slack, U * P)) || (r = dalloc(c, &d.t_lambda, U * P)) || (r = dalloc(c, &d.piece_time, U)) ||
      (r = dalloc(c, &d.oplanes, U * S * d.cap_obs * 4)) || (r = dalloc(c, &d.ocount, U * S)) ||
      (r = dalloc(c, &d.splanes, U * S * d.cap_self * 4)) || (r = dalloc(c, &d.scount, U * S)) ||
      (r = dalloc(c, &d.lg, U * P * 19)) || (r = dalloc(c, &d.lh, U * P * 361)) || (r = dalloc(c, &d.xdir, U * d.xs)) ||
      (r = dalloc(c, &d.k_obs, U)) || (r = dalloc(c, &d.k_self, U)) || (r = dalloc(c, &d.step_out, U)) || (r = dalloc(c, &d.ls_hist, U)) || (r = dalloc(c, &d.grad_cost, (d.u1 - d.u0) * P)) || (r = dalloc(c, &d.grad_perm, (d.u1 - d.u0) * P)) || (r = dalloc(c, &d.ls_tab, U * LS_TAB_STRIDE)) || (r = dalloc(c, &d.ls_word, U)) ||
      (r = dalloc(c, &d.ccdinfo, U * S * CCD_STRIDE)) || (r = dalloc(c, &d.pair_list, ACT_CAP)) ||
      (r = dalloc(c, &d.seg_stats, U * S * 6)) || (r = dalloc(c, &d.pair_stats, U * S * 2)) || (r = dalloc(c, &d.blk_stats, U * P + U)) ||
      (r = dalloc(c, &d.hullinfo, U * S * HULL_STRIDE)) || (r = dalloc(c, &d.hbox, S * 6 * U)) || (r = dalloc(c, &d.cbox, S * 6 * U)) || (r = dalloc(c, &d.pairplane, d.mode >= 1 ? S * U * U * 4 : 1)) ||
      (r = dalloc(c, &d.pairstamp, d.mode >= 1 ? S * U * U : 1)) || (r = dalloc(c, &d.pairbits, d.mode >= 1 ? S * U * ((U + 63) / 64) : 1)) ||
      (r = dalloc(c, &d.pair_work, 3 * (size_t)d.cap_work)) || (r = dalloc(c, &d.pair_work_n, (size_t)d.S + 1)) || (r = dalloc(c, &d.pair_ovf, 4)) || (r = dalloc(c, &d.seq_gmem_d, seq_fold_gmem_doubles(d.U))) || (r = dalloc(c, &d.seq_gmem_i, seq_fold_gmem_ints(d.U))) || (r = dalloc(c, &d.spec_n, 2)) || (r = dalloc(c, &d.spec_list, 2 * SPEC_CAP)) || (r = dalloc(c, &d.spec_tag, SPEC_CAP)) || (r = dalloc(c, &d.spec_state, SPEC_CAP * SPEC_STATE_DOUBLES)) || (r = dalloc(c, &d.spec_sti, SPEC_CAP * SPEC_STATE_INTS)) || (r = dalloc(c, &d.pair_ovf_list, (size_t)d.cap_work + PAIR_CONSUMERS_MAX)) || (r = dalloc(c, &d.ccd_found, 64)) || (r = dalloc(c, &d.ctl, 1)) ||
      (r = dalloc(c, &d.ocand, U * S * d.cap_obs)) || (r = dalloc(c, &d.ocand_n, U * S)) || (r = dalloc(c, &d.ohull, U * S * 18)) ||
      (r = dalloc(c, &d.obs_work, 2 * U * S * d.cap_obs)) || (r = dalloc(c, &d.obs_work_n, 1)) ||
      (r = dalloc(c, &d.oraw, U * S * d.cap_obs * 4)) || (r = dalloc(c, &d.ostamp, U * S * d.cap_obs)) ||
      (r = dalloc(c, &d.grad_scr, (size_t)(d.u1 - d.u0) * P * 16 * (size_t)(d.cap_obs + d.cap_self))) ||
      (r = dalloc(c, &d.xs_scr, d.xs_band ? (size_t)(d.u1 - d.u0) * ((size_t)n * n + 4 * n) : 1)) ||
      (r = dalloc(c, &d.xf_seg, 2 * S * XF_SEG_STRIDE)) || (r = dalloc(c, &d.xs_sync, (2 * U + 2) * 32)) || (r = dalloc(c, &d.keep_sync, 17 * 32)) || (r = dalloc(c, &d.fa_sync, Dev::fa_sync_ints(d.U)))) return r;
  if (d.optimal_plane) {
    const bool m0 = d.mode == 0;
    if ((r = dalloc(c, &d.kobs_id, m0 ? U * S * d.cap_obs : 1)) || (r = dalloc(c, &d.kobs_n, U * S)) || (r = dalloc(c, &d.kobs_cd, m0 ? U * S * d.cap_obs * 4 : 1)) ||
        (r = dalloc(c, &d.kpair_on, m0 ? 1 : S * U * U)) || (r = dalloc(c, &d.kpair_list, m0 ? 1 : S * U * U)) || (r = dalloc(c, &d.kpair_n, 2)) ||
        (r = dalloc(c, &d.kpair_cd, m0 ? 1 : S * U * U * 4))) return r;
  }
  {   // self-healing: what a batch's first state consists of (everything an iteration reads that an earlier iteration wrote and that is not rebuilt or re-stamped anyway)
    c->heal = !(tune("HEAL") && atoi(tune("HEAL")) == 0);
    if (const char* e = tune("XS_FAULT")) c->xs_fault = atoi(e);   // test hook: the n-th gate of the asynchronous solve reports a time-out
    std::vector<std::pair<void*, size_t>> reg = {
      {d.spline, U * 3 * T * 8}, {d.p_slack, U * 18 * P * 8}, {d.p_lambda, U * 18 * P * 8}, {d.t_slack, U * P * 8}, {d.t_lambda, U * P * 8}, {d.piece_time, U * 8},
      {d.xdir, U * d.xs * 8}, {d.ls_hist, U * 4}, {d.step_out, U * 8}, {d.seg_stats, U * S * 6 * 8}, {d.pair_stats, U * S * 2 * 8}, {d.blk_stats, (U * P + U) * 8}, {d.ccd_found, 64 * 4}};
    if (d.optimal_plane) {
      const bool m0 = d.mode == 0;
      if (m0) { reg.push_back({d.kobs_id, U * S * d.cap_obs * 4}); reg.push_back({d.kobs_n, U * S * 4}); reg.push_back({d.kobs_cd, U * S * d.cap_obs * 32}); }
      else { reg.push_back({d.kpair_on, S * U * U * 4}); reg.push_back({d.kpair_list, S * U * U * 4}); reg.push_back({d.kpair_n, 8}); reg.push_back({d.kpair_cd, S * U * U * 32}); }
    }
    std::vector<SnapRegion> tab;
    for (auto& pr : reg) {
      char* snap = nullptr;
      if ((r = dalloc(c, &snap, (pr.second + 15) / 16 * 16))) return r;
      tab.push_back(SnapRegion{(char*)pr.first, snap, (unsigned long long)pr.second});
    }
    c->snap_n = (int)tab.size();
    if ((r = dalloc(c, &c->snap_tab, tab.size())) || (r = dalloc(c, &c->ctl_snap, 1)) || (r = upload(c, c->snap_tab, tab.data(), tab.size() * sizeof(SnapRegion)))) return r;
  }
  if (d.mode == TJ_MODE_MULTI_COUPLED &&
      ((r = dalloc(c, &d.xL, U * (d.xs_band ? (size_t)(n - 1) * BAND_BS + n : (size_t)n * n))) || (r = dalloc(c, &d.xy, U * (size_t)n)) || (r = dalloc(c, &d.xg, U * (size_t)n)) ||
       (r = dalloc(c, &d.xcorner, U * 4)) || (r = dalloc(c, &d.k_obs_f, U)) || (r = dalloc(c, &d.ls_e, (size_t)LSC_ROUNDS * U * LS_GROUPS)))) return r;
  return TJ_OK;
}

void tj_destroy(tj_ctx* c) {
  if (!c) return;
  drop_graph(c);
  if (c->hwq_claim) { g_async_queues[std::min(std::max(c->prm.device, 0), 63)].fetch_sub(c->hwq_claim); c->hwq_claim = 0; }
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
  if (c->stream3) { (void)hipStreamSynchronize(c->stream3); (void)hipStreamDestroy(c->stream3); }
  for (void* p : c->xch_ipc_opened) (void)hipIpcCloseMemHandle(p);
  if (c->xch_block) (void)hipFree(c->xch_block);
  for (void* p : c->allocs) hipFree(p);
  for (void* p : c->cloud_allocs) hipFree(p);
  if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
  delete c;
}

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
};
int to_dev(tj_ctx* c, DevBuf& b, const void* src, size_t bytes) {
  HIPCHK(c, hipMalloc(&b.p, std::max<size_t>(bytes, 8)));
  if (src && bytes) return upload(c, b.p, src, bytes);
  return TJ_OK;
}
}  // namespace

namespace {
struct EvPair {   // the two timing events of the device build, destroyed on every exit
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EvPair() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); }
};
// verts: [n][prim][3] in the caller's order.  Everything that can be refused is checked BEFORE the current obstacle set is
// touched; from the first free on the context counts as "no obstacles" (have_cloud = false, N = 0) until a new set is
// completely built, so a failed call can never leave tj_iterate with a half-built hierarchy.
int set_obstacles(tj_ctx* c, const double* verts, int n, int prim) {
  Dev& d = c->d;
  if (prim == 3 && d.optimal_plane && d.mode == TJ_MODE_SINGLE) {
    c->err = "triangle obstacles with optimal_plane:1 in single-UAV mode are not supported (Optimal_plane::optimal_cd is defined for obstacle points only, Optimal_plane.h:160)";
    return TJ_ERR_UNSUPPORTED;
  }
  // pyramid geometry: level 0 = boxes over 8 consecutive primitives, up to a top level of <= 64 boxes
  std::vector<int> lvl_off, lvl_n;
  if (n > 0) { int cnt = (n + 7) / 8, off = 0; for (;;) { lvl_off.push_back(off); lvl_n.push_back(cnt); off += cnt; if (cnt <= 64) break; cnt = (cnt + 7) / 8; } }
  if ((int)lvl_n.size() > MAX_LEVELS) { c->err = "too many obstacle primitives for MAX_LEVELS"; return TJ_ERR_UNSUPPORTED; }
  QUIESCE(c);
  drop_graph(c);
  c->have_cloud = false;
  d.N = 0; d.nlevels = 0; if (!tune("BVH_SKIP")) d.bvh_skip = 0; d.px = d.py = d.pz = d.tri = nullptr; d.boxes = d.leafbox = nullptr;
  for (void* p : c->cloud_allocs) hipFree(p);
  c->cloud_allocs.clear();
  c->cloud_order.clear();
  if (n > 0) {
    for (int k = 0; k < 3; k++) { c->cloud_lo[k] = INFINITY; c->cloud_hi[k] = -INFINITY; }
    for (size_t i = 0; i < (size_t)n * prim; i++) for (int k = 0; k < 3; k++) { c->cloud_lo[k] = std::min(c->cloud_lo[k], verts[3 * i + k]); c->cloud_hi[k] = std::max(c->cloud_hi[k], verts[3 * i + k]); }
    const size_t nbox = (size_t)lvl_off.back() + lvl_n.back();
    float* boxes; int r;
    if ((r = dalloc(c, &boxes, nbox * 6, &c->cloud_allocs))) return r;
    double *px = nullptr, *py = nullptr, *pz = nullptr, *tri = nullptr; float* lb = nullptr;
    if (prim == 1) { if ((r = dalloc(c, &px, n, &c->cloud_allocs)) || (r = dalloc(c, &py, n, &c->cloud_allocs)) || (r = dalloc(c, &pz, n, &c->cloud_allocs))) return r; }
    else if ((r = dalloc(c, &tri, (size_t)n * 9, &c->cloud_allocs)) || (r = dalloc(c, &lb, (size_t)n * 6, &c->cloud_allocs))) return r;
    std::vector<int> order(n);
    c->bvh_on_device = tune("BVH_HOST") ? 0 : 1;   // TJ_BVH_HOST=1: the host build of host_tables.h (the checker of the device build)
    if (!c->bvh_on_device) {
      HostBvh b;
      build_bvh(verts, n, prim, b);
      if ((r = upload(c, boxes, b.boxes.data(), b.boxes.size() * 4))) return r;
      if (prim == 1) { if ((r = upload(c, px, b.px.data(), (size_t)n * 8)) || (r = upload(c, py, b.py.data(), (size_t)n * 8)) || (r = upload(c, pz, b.pz.data(), (size_t)n * 8))) return r; }
      else if ((r = upload(c, tri, b.tri.data(), (size_t)n * 72)) || (r = upload(c, lb, b.leafbox.data(), (size_t)n * 24))) return r;
      order = b.order;
      c->bvh_build_ms = 0;
    } else {
      // device build (kernels_bvh.h): bounds -> Morton keys -> stable radix sort -> gather -> box pyramid
      DevBuf dv, dpart, dlohi, dkA, dkB, dvA, dvB, dhist, d64a, d64b;
      const int nb_red = std::min(1024, (n + 255) / 256), nblocks = (n + RS_TILE - 1) / RS_TILE;
      if ((r = to_dev(c, dv, verts, (size_t)n * prim * 24)) || (r = to_dev(c, dpart, nullptr, (size_t)nb_red * 48)) || (r = to_dev(c, dlohi, nullptr, 48)) ||
          (r = to_dev(c, dkA, nullptr, (size_t)n * 8)) || (r = to_dev(c, dkB, nullptr, (size_t)n * 8)) || (r = to_dev(c, dvA, nullptr, (size_t)n * 4)) || (r = to_dev(c, dvB, nullptr, (size_t)n * 4)) ||
          (r = to_dev(c, dhist, nullptr, (size_t)256 * nblocks * 4)) || (r = to_dev(c, d64a, nullptr, (size_t)lvl_n[0] * 48)) || (r = to_dev(c, d64b, nullptr, (size_t)(lvl_n.size() > 1 ? lvl_n[1] : 1) * 48))) return r;
      EvPair ev;
      HIPCHK(c, hipEventCreate(&ev.e0)); HIPCHK(c, hipEventCreate(&ev.e1));
      hipStream_t s = c->stream;
      HIPCHK(c, hipEventRecord(ev.e0, s));
      const double* V = (const double*)dv.p;
      hipLaunchKernelGGL(k_bvh_bounds, dim3(nb_red), dim3(256), 0, s, V, n, prim, (double*)dpart.p);
      hipLaunchKernelGGL(k_bvh_bounds_final, dim3(1), dim3(64), 0, s, (const double*)dpart.p, nb_red, (double*)dlohi.p);
      unsigned long long *kA = (unsigned long long*)dkA.p, *kB = (unsigned long long*)dkB.p; int *vA = (int*)dvA.p, *vB = (int*)dvB.p;
      hipLaunchKernelGGL(k_bvh_keys, dim3((n + 255) / 256), dim3(256), 0, s, V, n, prim, (const double*)dlohi.p, kA, vA);
      for (int pass = 0; pass < 8; pass++) {
        hipLaunchKernelGGL(k_rsort_hist, dim3(nblocks), dim3(RS_THREADS), 0, s, kA, n, 8 * pass, nblocks, (int*)dhist.p);
        hipLaunchKernelGGL(k_rsort_scan, dim3(1), dim3(1024), 0, s, (int*)dhist.p, 256 * nblocks);
        hipLaunchKernelGGL(k_rsort_scatter, dim3(nblocks), dim3(RS_THREADS), 0, s, kA, vA, n, 8 * pass, nblocks, (const int*)dhist.p, kB, vB);
        std::swap(kA, kB); std::swap(vA, vB);
      }
      hipLaunchKernelGGL(k_bvh_gather, dim3((n + 255) / 256), dim3(256), 0, s, V, vA, n, prim, px, py, pz, tri, lb);
      double *cur = (double*)d64a.p, *prev = (double*)d64b.p;
      for (size_t lv = 0; lv < lvl_n.size(); lv++) {
        const int nchild = lv == 0 ? n : lvl_n[lv - 1];
        hipLaunchKernelGGL(k_bvh_level, dim3((lvl_n[lv] + 255) / 256), dim3(256), 0, s, (int)lv, lvl_n[lv], nchild, prim, px, py, pz, tri, prev, cur, boxes + (size_t)lvl_off[lv] * 6);
        std::swap(cur, prev);
      }
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipEventRecord(ev.e1, s));
      HIPCHK(c, hipStreamSynchronize(s));
      float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, ev.e0, ev.e1)); c->bvh_build_ms = ms;
      HIPCHK(c, hipMemcpy(order.data(), vA, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    // the build succeeded: publish it
    c->cloud_order.swap(order);
    d.boxes = boxes; d.px = px; d.py = py; d.pz = pz; d.tri = tri; d.leafbox = lb;
    d.nlevels = (int)lvl_n.size();
    // two levels per step at the top of the walk (kernels_sep.h): pays where the pyramid is deep AND the query waves outnumber the resident slots, i.e. where a
    // query's latency is the launch's throughput (256 robots x 1 M primitives: k_front 40.2 -> 36.3 us, k_ccd 34.0 -> 31.0); 64 robots through 1 M points: no change
    if (!tune("BVH_SKIP")) d.bvh_skip = (d.nlevels >= 5 && (d.u1 - d.u0) * d.S > 3584) ? 1 : 0;
    for (int i = 0; i < d.nlevels; i++) { d.lvl_off[i] = lvl_off[i]; d.lvl_n[i] = lvl_n[i]; }
    d.N = n;
  }
  d.prim = prim;
  c->have_cloud = true;
  return TJ_OK;
}
}  // namespace

int tj_set_cloud(tj_ctx* c, const double* xyz, int n) {
  if (!c || n < 0 || (n > 0 && !xyz)) return TJ_ERR_INVALID;
  return set_obstacles(c, xyz, n, 1);
}

int tj_set_mesh(tj_ctx* c, const double* vertices, int n_vertices, const int* faces, int n_faces) {
  if (!c || n_vertices < 0 || n_faces < 0 || (n_faces > 0 && (!vertices || !faces))) return TJ_ERR_INVALID;
  std::vector<double> tri((size_t)n_faces * 9);
  for (int f = 0; f < n_faces; f++)
    for (int j = 0; j < 3; j++) {
      const int v = faces[3 * (size_t)f + j];
      if (v < 0 || v >= n_vertices) { c->err = "tj_set_mesh: face " + std::to_string(f) + " refers to vertex " + std::to_string(v); return TJ_ERR_INVALID; }
      for (int k = 0; k < 3; k++) tri[(size_t)f * 9 + 3 * j + k] = vertices[3 * (size_t)v + k];
    }
  return set_obstacles(c, tri.data(), n_faces, 3);
}

int tj_init_state(tj_ctx* c, const double* wp, double pt0) {
  if (!c || !wp) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  const int U = d.U, P = d.P, T = d.T;
  HostTables t;
  build_tables(P, d.res, 1, t);
  std::vector<double> spline((size_t)U * 3 * T), p_slack((size_t)U * 18 * P), zeros((size_t)U * 18 * P, 0.0), ts((size_t)U * P, pt0), tz((size_t)U * P, 0.0), ptv(U, pt0);
  for (int u = 0; u < U; u++) {
    double* s = &spline[(size_t)u * 3 * T];
    for (int a = 0; a < 3; a++) {
      auto W = [&](int k) { return wp[((size_t)u * (P + 1) + k) * 3 + a]; };
      double* col = s + T * a;
      col[0] = W(0);
      if (d.mode == TJ_MODE_SINGLE) {  // Main/admmPathPlanning3D.cpp:258-275
        for (int i = 0; i < P; i++) {
          const double head = 0.9 * W(i) + 0.1 * W(i + 1), tail = 0.9 * W(i + 1) + 0.1 * W(i);
          col[3 * i + 1] = W(i);
          for (int j = 1; j < 3; j++) col[j + 3 * i + 1] = double(2 - j) / 1 * head + (double)(j - 1) / 1 * tail;
          col[3 * (i + 1) + 1] = W(i + 1);
        }
      } else {  // Main/multiPathPlanning3D.cpp:363-375
        for (int k = 0; k < P; k++)
          for (int j = 0; j <= 3; j++) col[j + 3 * k + 1] = double(3 - j) / 3 * W(k) + (double)j / 3 * W(k + 1);
      }
      col[T - 1] = W(P);
      col[1] = col[0];
      col[T - 2] = col[T - 1];
    }
    for (int sp = 0; sp < P; sp++)
      for (int a = 0; a < 3; a++)
        for (int j = 0; j < 6; j++) {
          double acc = 0;
          for (int k = 0; k < 6; k++) acc += t.convert[sp * 36 + j * 6 + k] * s[sp * 3 + k + T * a];
          p_slack[(size_t)u * 18 * P + sp * 6 + j + 6 * P * a] = acc;
        }
  }
  QUIESCE(c);
  int r;
  if ((r = upload(c, d.spline, spline.data(), spline.size() * 8)) || (r = upload(c, d.p_slack, p_slack.data(), p_slack.size() * 8)) ||
      (r = upload(c, d.p_lambda, zeros.data(), zeros.size() * 8)) || (r = upload(c, d.t_slack, ts.data(), ts.size() * 8)) ||
      (r = upload(c, d.t_lambda, tz.data(), tz.size() * 8)) || (r = upload(c, d.piece_time, ptv.data(), ptv.size() * 8))) return r;
  // direct exchange: the push / arrival counts restart (the caller has every rank drained before any rank calls this, and a barrier after:
  // tj_group_init_state does; processes use their collective's barrier)
  if (c->xch_block) HIPCHK(c, hipMemsetAsync(d.xcnt, 0, 2 * XCH_MAX * sizeof(unsigned long long), c->stream));
  if (d.xf) HIPCHK(c, hipMemsetAsync(d.xf_seg, 0, (size_t)2 * d.S * XF_SEG_STRIDE * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(d.xs_sync, 0, ((size_t)2 * d.U + 1) * 32 * sizeof(int), c->stream));
  HIPCHK(c, hipMemsetAsync(d.fa_sync, 0, Dev::fa_sync_ints(d.U) * sizeof(int), c->stream)); c->fa_seq = 0; c->fa_armed = false;   // (the words are monotonic in the pairing number, which restarts here)
  Ctl h;
  memset(&h, 0, sizeof(h));
  h.gnorm = 1.0;  // Main/multiPathPlanning3D.cpp:594
  if ((r = upload(c, d.ctl, &h, sizeof(h)))) return r;
  c->hull_valid = false; c->ccd_valid = false;
  HIPCHK(c, hipMemsetAsync(d.xdir, 0, (size_t)U * d.xs * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(d.ocount, 0, (size_t)U * d.S * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(d.scount, 0, (size_t)U * d.S * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(d.ocand_n, 0, (size_t)U * d.S * 4, c->stream));
  if (d.optimal_plane) {  // the mains start with empty persistent tables (Main/admmPathPlanning3D.cpp:343-351, Main/multiPathPlanning3D.cpp:450-464)
    HIPCHK(c, hipMemsetAsync(d.kobs_n, 0, (size_t)U * d.S * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(d.kpair_n, 0, 8, c->stream));
    if (d.mode != 0) HIPCHK(c, hipMemsetAsync(d.kpair_on, 0, (size_t)d.S * U * U * 4, c->stream));
  }
  HIPCHK(c, hipMemsetAsync(d.ostamp, 0, (size_t)U * d.S * d.cap_obs * 4, c->stream));  // epochs restart at 1
  HIPCHK(c, hipMemsetAsync(d.seg_stats, 0, (size_t)U * d.S * 6 * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(d.pair_stats, 0, (size_t)U * d.S * 2 * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(d.blk_stats, 0, ((size_t)U * d.P + U) * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(d.ls_hist, 0xff, (size_t)U * 4, c->stream));   // -1: no line search yet
  HIPCHK(c, hipMemsetAsync(d.ls_tab, 0xff, (size_t)U * LS_TAB_STRIDE * 8, c->stream));   // LS_TAB_EMPTY
  HIPCHK(c, hipMemsetAsync(d.ls_word, 0, (size_t)U * 8, c->stream));                        // (the words carry the epoch, which restarts at 1)
  {   // k_grad's launch order: no history, identity
    std::vector<int> idp((size_t)(d.u1 - d.u0) * d.P);
    for (size_t i = 0; i < idp.size(); i++) idp[i] = (int)i;
    HIPCHK(c, hipMemsetAsync(d.grad_cost, 0, idp.size() * 4, c->stream));
    HIPCHK(c, hipMemcpyAsync(d.grad_perm, idp.data(), idp.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (d.mode >= 1) HIPCHK(c, hipMemsetAsync(d.pairstamp, 0, (size_t)d.S * U * U * 4, c->stream));  // epochs restart at 1
  if (d.mode >= 1) HIPCHK(c, hipMemsetAsync(d.pairbits, 0, (size_t)d.S * U * ((U + 63) / 64) * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(d.pair_ovf_list, 0, ((size_t)d.cap_work + PAIR_CONSUMERS_MAX) * 8, c->stream));                  // (entries are tagged with the epoch)
  HIPCHK(c, hipMemsetAsync(d.pair_ovf, 0, 16, c->stream));
  HIPCHK(c, hipMemsetAsync(d.spec_n, 0, 8, c->stream));
  HIPCHK(c, hipMemsetAsync(d.spec_tag, 0, SPEC_CAP * 8, c->stream));                                                          // (tags carry the epoch)
  c->have_state = true;
  return TJ_OK;
}

int tj_get_state(tj_ctx* c, int u, double* spline, double* p_slack, double* p_lambda, double* t_slack, double* t_lambda, double* piece_time) {
  if (!c || u < 0 || u >= c->d.U) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  if (spline) HIPCHK(c, hipMemcpy(spline, d.spline + (size_t)u * 3 * d.T, 3 * d.T * 8, hipMemcpyDeviceToHost));
  if (p_slack) HIPCHK(c, hipMemcpy(p_slack, d.p_slack + (size_t)u * 18 * d.P, 18 * d.P * 8, hipMemcpyDeviceToHost));
  if (p_lambda) HIPCHK(c, hipMemcpy(p_lambda, d.p_lambda + (size_t)u * 18 * d.P, 18 * d.P * 8, hipMemcpyDeviceToHost));
  if (t_slack) HIPCHK(c, hipMemcpy(t_slack, d.t_slack + (size_t)u * d.P, d.P * 8, hipMemcpyDeviceToHost));
  if (t_lambda) HIPCHK(c, hipMemcpy(t_lambda, d.t_lambda + (size_t)u * d.P, d.P * 8, hipMemcpyDeviceToHost));
  if (piece_time) HIPCHK(c, hipMemcpy(piece_time, d.piece_time + u, 8, hipMemcpyDeviceToHost));
  return TJ_OK;
}

int tj_set_state(tj_ctx* c, int u, const double* spline, const double* p_slack, const double* p_lambda, const double* t_slack, const double* t_lambda, double piece_time) {
  if (!c || u < 0 || u >= c->d.U) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  int r;
  if (spline && (r = upload(c, d.spline + (size_t)u * 3 * d.T, spline, 3 * d.T * 8))) return r;
  if (p_slack && (r = upload(c, d.p_slack + (size_t)u * 18 * d.P, p_slack, 18 * d.P * 8))) return r;
  if (p_lambda && (r = upload(c, d.p_lambda + (size_t)u * 18 * d.P, p_lambda, 18 * d.P * 8))) return r;
  if (t_slack && (r = upload(c, d.t_slack + (size_t)u * d.P, t_slack, d.P * 8))) return r;
  if (t_lambda && (r = upload(c, d.t_lambda + (size_t)u * d.P, t_lambda, d.P * 8))) return r;
  if ((r = upload(c, d.piece_time + u, &piece_time, 8))) return r;
  c->hull_valid = false; c->ccd_valid = false;
  c->have_state = true;
  return TJ_OK;
}

int tj_iterate_async(tj_ctx* c, int n_iters) {
  if (!c || n_iters < 0) return TJ_ERR_INVALID;
  if (!ready(c)) return TJ_ERR_INVALID;
  if (c->heal && n_iters > 0 && (c->xs_two_queues || c->keep_two_queues)) {   // self-healing: the state this batch starts from (one launch), unless iterations the host has not looked at yet are already outstanding
    if (c->snap_iters == 0 && !c->heal_busy) {
      if (!c->begin_folded) c->snap_in_begin = true;   // the batch's first launch is k_begin: the snapshot rides in it
      else { hipLaunchKernelGGL(k_snapshot, dim3(64, std::max(c->snap_n, 1)), dim3(256), 0, c->stream, c->snap_tab, c->snap_n, 0, c->d.ctl, c->ctl_snap); HIPCHK(c, hipGetLastError()); }
    }
    if (!c->heal_busy) c->snap_iters += n_iters;
  }
  if (n_iters > 0) { int r = ensure_hull_cache(c); if (r) return r; }
  // inside a batch the begin work of iteration i+1 rides on iteration i's k_linesearch (not in coupled mode, whose line search
  // is several kernels, and not in the captured-graph replay, which is one fixed iteration)
  // (coupled mode: only where the whole search is ONE launch whose last block commits -- lsc_wide, one context)
  const bool chain = !c->use_graph && (c->d.mode != TJ_MODE_MULTI_COUPLED || (c->lsc_wide && c->d.u1 - c->d.u0 == c->d.U));
  for (int i = 0; i < n_iters; i++) {
    const int pos = chain ? (((i > 0 || c->begin_folded) ? 1 : 0) | (i + 1 < n_iters ? 2 : 0)) : 0;
    c->begin_folded = false;
    int r = launch_graph_or_eager(c, 3, pos); if (r) return r;
  }
  c->iters_enqueued += n_iters;
  return TJ_OK;
}

int tj_sync(tj_ctx* c) {
  if (!c) return TJ_ERR_INVALID;
  { int r = flush_deferred(c); if (r) return r; }
  QUIESCE_NOHEAL(c);   // (no read-back here: a batch that has to be run again -- heal_check -- is noticed by the next call that reads a result or the statistics)
  return TJ_OK;
}

void* tj_stream(tj_ctx* c) { return c ? (void*)c->stream : nullptr; }

int tj_set_stream(tj_ctx* c, void* hip_stream) {
  if (!c) return TJ_ERR_INVALID;
  QUIESCE(c);
  drop_graph(c);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  c->stream = (hipStream_t)hip_stream;
  c->own_stream = false;
  return TJ_OK;
}

int tj_profile_kernels(tj_ctx* c, int n_iters, double* ms, int* launches) {
  if (!c || !ms || n_iters < 0) return TJ_ERR_INVALID;
  if (!ready(c)) return TJ_ERR_INVALID;
  { int fr = flush_deferred(c); if (fr) return fr; }
  { int hr = ensure_hull_cache(c); if (hr) return hr; }
  std::vector<hipEvent_t> ev((size_t)n_iters * (K_COUNT + 1));
  for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
  std::vector<int> ran(K_COUNT, 0);
  c->xs_same_queue_now = true;   // per-kernel events on one queue: the asynchronous solve follows k_grad there (its own time is then what the events show)
  for (int it = 0; it < n_iters; it++) {
    hipEvent_t* e = &ev[(size_t)it * (K_COUNT + 1)];
    HIPCHK(c, hipEventRecord(e[0], c->stream));
    const int pos = (c->d.mode != TJ_MODE_MULTI_COUPLED) ? ((it > 0 ? 1 : 0) | (it + 1 < n_iters ? 2 : 0)) : 0;
    for (int k = 0; k < K_COUNT; k++) {
      if (launch_kernel(c, k, c->stream, 0, true, false, pos)) ran[k]++;
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipEventRecord(e[k + 1], c->stream));
    }
  }
  c->xs_same_queue_now = false;
  if (n_iters > 0) c->maybe_deferred = true;  // the last iteration's slack/dual update is still owed (paid by the flush below)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < K_COUNT; k++) ms[k] = 0;
  for (int it = 0; it < n_iters; it++)
    for (int k = 0; k < K_COUNT; k++) {
      float t = 0;
      HIPCHK(c, hipEventElapsedTime(&t, ev[(size_t)it * (K_COUNT + 1) + k], ev[(size_t)it * (K_COUNT + 1) + k + 1]));
      if (ran[k]) ms[k] += t;
    }
  for (auto& e : ev) hipEventDestroy(e);
  if (launches) for (int k = 0; k < K_COUNT; k++) launches[k] = ran[k];
  return check_device_errors(c);
}

long long tj_launch_count(tj_ctx* c) { return c ? c->launches : -1; }
int tj_kernel_count(void) { return K_COUNT; }
const char* tj_kernel_name(int i) { return (i >= 0 && i < K_COUNT) ? kKernelNames[i] : ""; }

int tj_iterate(tj_ctx* c, int n_iters, double* gnorm, int* iters_total, int* converged) {
  int r = tj_iterate_async(c, n_iters);
  if (r) return r;
  if ((r = flush_deferred(c))) return r;
  Ctl h;
  r = check_device_errors(c, &h);
  // the iteration counter of the last started iteration is committed by the next k_begin
  const int it = h.iter + h.pending;
  if (gnorm) *gnorm = h.gnorm;
  if (iters_total) *iters_total = it;
  if (converged) *converged = h.done || (c->d.stop > 0 && it > 1 && h.gnorm < c->d.stop);
  return r;
}

int tj_run_stage(tj_ctx* c, int stage) {
  if (!c) return TJ_ERR_INVALID;
  if (!ready(c)) return TJ_ERR_INVALID;
  int r = flush_deferred(c);
  if (r) return r;
  if ((r = enqueue_stage(c, stage))) return r;
  return check_device_errors(c);
}

int tj_phase_count(tj_ctx* c) { return c ? (c->d.mode == TJ_MODE_MULTI_COUPLED ? 6 : 3) : TJ_ERR_INVALID; }

int tj_iterate_phase_chained(tj_ctx* c, int phase, int more) {
  if (!c || phase < 0 || phase >= tj_phase_count(c)) return TJ_ERR_INVALID;
  if (!ready(c)) return TJ_ERR_INVALID;
  // eager launches: measured faster than three graph replays per iteration (a replay costs ~10-16 us of host time, a
  // plain launch ~3.5 us, and a phase has only 2-7 kernels).  No flush here: the slack/dual update an iteration owes is
  // paid by k_mid of the next iteration's phase 1 (or by tj_sync / any state access).
  // Decoupled / single-UAV schedules fold the next iteration's begin into the last phase's k_linesearch when the caller says one follows
  // (more != 0): that iteration's phase 0 then launches nothing.  A begin that was folded for an iteration the caller never enqueues is
  // taken back by the next flush (tj_sync, any state access).
  int pos = 0;
  if (c->d.mode != TJ_MODE_MULTI_COUPLED && c->d.fuse) {
    if (phase == 0) { pos = c->begin_folded ? 1 : 0; c->begin_folded = false; }
    if (phase == 2 && more) { pos = 2; c->begin_folded = true; }
  }
  if (phase == 0) { c->iters_enqueued += 1; c->lsc_base = 0; }
  return enqueue_body(c, phase, pos);
}

// Coupled mode, sharded contexts: the Armijo search on the summed energy to the reference's own end (Optimization3D_multi.h:605-636: no bound).  One exchange carries the
// candidates of LSC_ROUNDS rounds (steps 0.8^0 .. 0.8^30 in the first table).  With `follow` on, phase 5 commits nothing when none of them passes and the caller asks here
// (the stream is drained: this is the one host look of the schedule, paid only by callers that want the exact loop): pending = 1 -> run phase 4, exchange buffer 4, phase 5
// again -- they evaluate, carry and decide the NEXT rounds -- and ask again.  Every rank of the sharded run reads the same answer (same gathered table, same decision).
int tj_set_coupled_follow(tj_ctx* c, int on) {
  if (!c) return TJ_ERR_INVALID;
  if (c->d.mode != TJ_MODE_MULTI_COUPLED) { c->err = "tj_set_coupled_follow: coupled mode only"; return TJ_ERR_INVALID; }
  c->d.lsc_follow = on ? 1 : 0; c->lsc_base = 0;
  return TJ_OK;
}
int tj_coupled_search_pending(tj_ctx* c, int* pending) {
  if (!c || !pending) return TJ_ERR_INVALID;
  *pending = 0;
  if (!c->d.lsc_follow || c->d.u1 - c->d.u0 == c->d.U) return TJ_OK;   // (one context follows the search inside its own launch: lsc_continue)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int p = 0;
  HIPCHK(c, hipMemcpy(&p, &c->d.ctl->lsc_pending, sizeof(int), hipMemcpyDeviceToHost));
  *pending = p ? 1 : 0;
  c->lsc_base = p ? c->lsc_base + LSC_ROUNDS : 0;
  if (!p) c->maybe_deferred = true;   // the step is committed: the iteration's slack/dual update is owed from here on
  return TJ_OK;
}
int tj_iterate_phase(tj_ctx* c, int phase) { return tj_iterate_phase_chained(c, phase, 0); }

int tj_exchange_buffer(tj_ctx* c, int what, void** dev_ptr, int* doubles_per_robot, int* first_owned, int* n_owned) {
  if (!c || what < 0 || what > 4) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  if (what >= 2 && d.mode != TJ_MODE_MULTI_COUPLED) { c->err = "tj_exchange_buffer: buffers 2..4 exist in coupled mode only"; return TJ_ERR_INVALID; }
  void* p = nullptr; int per = 0;
  switch (what) {
    case 0: p = d.spline; per = 3 * d.T; break;
    case 1: p = d.xdir; per = d.xs; break;
    case 2: p = d.xcorner; per = 4; break;                       // Schur-corner contributions of the shared piece_time
    case 3: p = d.k_obs_f; per = 1; break;                       // obstacle CCD exponent of every robot (as a double)
    case 4: p = d.ls_e; per = LSC_ROUNDS * LS_GROUPS; break;     // energies of the Armijo candidates
  }
  if (dev_ptr) *dev_ptr = p;
  if (doubles_per_robot) *doubles_per_robot = per;
  if (first_owned) *first_owned = d.u0;
  if (n_owned) *n_owned = d.u1 - d.u0;
  return TJ_OK;
}

int tj_get_planes(tj_ctx* c, int u, int* counts_obs, int* counts_self, double* planes, int cap) {
  if (!c || u < 0 || u >= c->d.U) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  std::vector<int> co(d.S), cs(d.S, 0);
  HIPCHK(c, hipMemcpy(co.data(), d.ocount + (size_t)u * d.S, d.S * 4, hipMemcpyDeviceToHost));
  if (d.mode >= 1) HIPCHK(c, hipMemcpy(cs.data(), d.scount + (size_t)u * d.S, d.S * 4, hipMemcpyDeviceToHost));
  int total = 0;
  for (int tr = 0; tr < d.S; tr++) total += co[tr] + cs[tr];
  if (counts_obs) memcpy(counts_obs, co.data(), d.S * 4);
  if (counts_self) memcpy(counts_self, cs.data(), d.S * 4);
  if (planes) {
    if (cap < total) { c->err = "tj_get_planes: buffer too small"; return TJ_ERR_INVALID; }
    size_t w = 0;
    for (int tr = 0; tr < d.S; tr++) {
      if (co[tr]) HIPCHK(c, hipMemcpy(planes + 4 * w, d.oplanes + ((size_t)u * d.S + tr) * d.cap_obs * 4, (size_t)co[tr] * 32, hipMemcpyDeviceToHost));
      w += co[tr];
      if (cs[tr]) HIPCHK(c, hipMemcpy(planes + 4 * w, d.splanes + ((size_t)u * d.S + tr) * d.cap_self * 4, (size_t)cs[tr] * 32, hipMemcpyDeviceToHost));
      w += cs[tr];
    }
  }
  return total;
}

int tj_get_candidates(tj_ctx* c, int u, int seg, int cap, int* ids, int* n_broad) {
  if (!c || u < 0 || u >= c->d.U || seg < 0 || seg >= c->d.S || cap < 0) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  const size_t s = (size_t)u * d.S + seg;
  int n = 0;
  HIPCHK(c, hipMemcpy(&n, d.ocand_n + s, 4, hipMemcpyDeviceToHost));
  const int m = std::min(n, cap);
  if (m > 0 && ids) {
    std::vector<int> tmp(m);
    HIPCHK(c, hipMemcpy(tmp.data(), d.ocand + s * d.cap_obs, (size_t)m * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < m; i++) ids[i] = c->cloud_order[tmp[i]];
  }
  if (n_broad) {
    unsigned long long st[6];
    HIPCHK(c, hipMemcpy(st, d.seg_stats + s * 6, sizeof(st), hipMemcpyDeviceToHost));
    *n_broad = (int)st[1];
  }
  return n;
}

int tj_set_planes(tj_ctx* c, int u, const int* counts, const double* planes) {
  if (!c || u < 0 || u >= c->d.U || !counts) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  size_t w = 0;
  std::vector<int> zero(d.S, 0);
  for (int tr = 0; tr < d.S; tr++) {
    if (counts[tr] > d.cap_obs) { c->err = "tj_set_planes: more planes than cap_obs"; return TJ_ERR_CAPACITY; }
    if (counts[tr]) { int ur = upload(c, d.oplanes + ((size_t)u * d.S + tr) * d.cap_obs * 4, planes + 4 * w, (size_t)counts[tr] * 32); if (ur) return ur; }
    w += counts[tr];
  }
  { int ur; if ((ur = upload(c, d.ocount + (size_t)u * d.S, counts, d.S * 4)) || (ur = upload(c, d.scount + (size_t)u * d.S, zero.data(), d.S * 4))) return ur; }
  return TJ_OK;
}

int tj_get_direction(tj_ctx* c, int u, double* direction, double* t_direction, double* wolfe, double* gn) {
  if (!c || u < 0 || u >= c->d.U) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  std::vector<double> rec(d.xs);
  HIPCHK(c, hipMemcpy(rec.data(), d.xdir + (size_t)u * d.xs, d.xs * 8, hipMemcpyDeviceToHost));
  if (direction) memcpy(direction, rec.data(), 3 * d.T * 8);
  if (t_direction) *t_direction = rec[3 * d.T];
  if (wolfe) *wolfe = rec[3 * d.T + 1];
  if (gn) *gn = rec[3 * d.T + 2];
  if (d.mode == TJ_MODE_MULTI_COUPLED) {  // one Newton system for all robots: report its global wolfe and gnorm
    Ctl h;
    HIPCHK(c, hipMemcpy(&h, d.ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    if (wolfe) *wolfe = h.wolfe_c;
    if (gn) *gn = h.gnorm;
  }
  return TJ_OK;
}

int tj_set_direction(tj_ctx* c, int u, const double* direction, double t_direction, double wolfe, double gn) {
  if (!c || u < 0 || u >= c->d.U || !direction) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  std::vector<double> rec(d.xs, 0.0);
  memcpy(rec.data(), direction, 3 * d.T * 8);
  rec[3 * d.T] = t_direction; rec[3 * d.T + 1] = wolfe; rec[3 * d.T + 2] = gn;
  c->ccd_valid = false;   // the swept-hull cache no longer matches: the next chained CCD stage rebuilds it (k_ccd_prep)
  return upload(c, d.xdir + (size_t)u * d.xs, rec.data(), d.xs * 8);
}

int tj_get_local_grad(tj_ctx* c, int u, int piece, double* g19, double* h361) {
  if (!c || u < 0 || u >= c->d.U || piece < 0 || piece >= c->d.P) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  if (g19) HIPCHK(c, hipMemcpy(g19, d.lg + ((size_t)u * d.P + piece) * 19, 19 * 8, hipMemcpyDeviceToHost));
  if (h361) HIPCHK(c, hipMemcpy(h361, d.lh + ((size_t)u * d.P + piece) * 361, 361 * 8, hipMemcpyDeviceToHost));
  return TJ_OK;
}

int tj_get_energy(tj_ctx* c, double* energy) {
  if (!c || !energy) return TJ_ERR_INVALID;
  if (!ready(c)) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  DevBuf out;
  { int r = to_dev(c, out, nullptr, (size_t)d.U * 8); if (r) return r; }
  HIPCHK(c, hipMemsetAsync(out.p, 0, (size_t)d.U * 8, c->stream));
  HIPCHK(c, hipFuncSetAttribute((const void*)k_energy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_ls));
  hipLaunchKernelGGL(k_energy, dim3(d.u1 - d.u0), dim3(LS_THREADS), c->lds_ls, c->stream, d, c->lsl, (double*)out.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(energy, out.p, (size_t)d.U * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return TJ_OK;
}

int tj_get_steps(tj_ctx* c, double* step_self, double* step_obs, double* step_armijo) {
  if (!c) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  QUIESCE(c);
  std::vector<int> ko(d.U), ks(d.U);
  HIPCHK(c, hipMemcpy(ko.data(), d.k_obs, d.U * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(ks.data(), d.k_self, d.U * 4, hipMemcpyDeviceToHost));
  auto p = [](int k) { double s = 1.0; for (int i = 0; i < k; i++) s *= 0.8; return s; };
  for (int u = 0; u < d.U; u++) { if (step_self) step_self[u] = p(ks[u]); if (step_obs) step_obs[u] = p(ko[u]); }
  if (step_armijo) HIPCHK(c, hipMemcpy(step_armijo, d.step_out, d.U * 8, hipMemcpyDeviceToHost));
  return TJ_OK;
}

#ifdef TJ_KAT
// ---- known-answer hooks ------------------------------------------------------------------------

int tj_kat_gjk(tj_ctx* c, int n, int n1, const double* a, int n2, const double* b, double* v) {
  if (!c || n < 0 || !a || !b || !v) return TJ_ERR_INVALID;
  DevBuf da, db, dv; int r;
  if ((r = to_dev(c, da, a, (size_t)n * n1 * 24)) || (r = to_dev(c, db, b, (size_t)n * n2 * 24)) || (r = to_dev(c, dv, nullptr, (size_t)n * 24))) return r;
  dim3 g((n + 63) / 64), t(64);
  const double *A = (const double*)da.p, *B = (const double*)db.p; double* V = (double*)dv.p;
  if (n1 == 6 && n2 == 1) hipLaunchKernelGGL((k_dbg_gjk<6, 1>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 6 && n2 == 6) hipLaunchKernelGGL((k_dbg_gjk<6, 6>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 12 && n2 == 1) hipLaunchKernelGGL((k_dbg_gjk<12, 1>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 12 && n2 == 12) hipLaunchKernelGGL((k_dbg_gjk<12, 12>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 6 && n2 == 3) hipLaunchKernelGGL((k_dbg_gjk<6, 3>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 12 && n2 == 3) hipLaunchKernelGGL((k_dbg_gjk<12, 3>), g, t, 0, c->stream, n, A, B, V);
  else { c->err = "tj_kat_gjk: body sizes must be 6v1, 6v3, 6v6, 12v1, 12v3 or 12v12"; return TJ_ERR_INVALID; }
  HIPCHK(c, hipGetLastError());
  QUIESCE(c);
  HIPCHK(c, hipMemcpy(v, dv.p, (size_t)n * 24, hipMemcpyDeviceToHost));
  return TJ_OK;
}

int tj_kat_gjk_wave(tj_ctx* c, int n, int n1, const double* a, int n2, const double* b, double* v) {
  if (!c || n < 0 || !a || !b || !v) return TJ_ERR_INVALID;
  DevBuf da, db, dv; int r;
  if ((r = to_dev(c, da, a, (size_t)n * n1 * 24)) || (r = to_dev(c, db, b, (size_t)n * n2 * 24)) || (r = to_dev(c, dv, nullptr, (size_t)n * 24))) return r;
  dim3 g(std::max(n, 1)), t(64);
  const double *A = (const double*)da.p, *B = (const double*)db.p; double* V = (double*)dv.p;
  if (n1 == 6 && n2 == 6) hipLaunchKernelGGL((k_dbg_gjk_wave<6, 6>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 12 && n2 == 12) hipLaunchKernelGGL((k_dbg_gjk_wave<12, 12>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 6 && n2 == 1) hipLaunchKernelGGL((k_dbg_gjk_wave<6, 1>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 12 && n2 == 1) hipLaunchKernelGGL((k_dbg_gjk_wave<12, 1>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 6 && n2 == 3) hipLaunchKernelGGL((k_dbg_gjk_wave<6, 3>), g, t, 0, c->stream, n, A, B, V);
  else if (n1 == 12 && n2 == 3) hipLaunchKernelGGL((k_dbg_gjk_wave<12, 3>), g, t, 0, c->stream, n, A, B, V);
  else { c->err = "tj_kat_gjk_wave: body sizes must be 6v1, 6v3, 6v6, 12v1, 12v3 or 12v12"; return TJ_ERR_INVALID; }
  HIPCHK(c, hipGetLastError());
  QUIESCE(c);
  HIPCHK(c, hipMemcpy(v, dv.p, (size_t)n * 24, hipMemcpyDeviceToHost));
  return TJ_OK;
}

int tj_kat_gjk_wave_split(tj_ctx* c, int n, int n1, const double* a, int n2, const double* b, int k_stop, double* v_iters) {
  if (!c || n < 0 || !a || !b || !v_iters || k_stop < 1) return TJ_ERR_INVALID;
  DevBuf da, db, ds, dv; int r;
  if ((r = to_dev(c, da, a, (size_t)n * n1 * 24)) || (r = to_dev(c, db, b, (size_t)n * n2 * 24)) || (r = to_dev(c, ds, nullptr, (size_t)n * 256)) || (r = to_dev(c, dv, nullptr, (size_t)n * 32))) return r;
  dim3 g(std::max(n, 1)), t(64);
  const double *A = (const double*)da.p, *B = (const double*)db.p; double* S = (double*)ds.p; double* V = (double*)dv.p;
  if (n1 == 6 && n2 == 6) hipLaunchKernelGGL((k_dbg_gjk_wave_split<6, 6>), g, t, 0, c->stream, n, A, B, k_stop, S, V);
  else if (n1 == 12 && n2 == 12) hipLaunchKernelGGL((k_dbg_gjk_wave_split<12, 12>), g, t, 0, c->stream, n, A, B, k_stop, S, V);
  else { c->err = "tj_kat_gjk_wave_split: body sizes must be 6v6 or 12v12"; return TJ_ERR_INVALID; }
  HIPCHK(c, hipGetLastError());
  QUIESCE(c);
  HIPCHK(c, hipMemcpy(v_iters, dv.p, (size_t)n * 32, hipMemcpyDeviceToHost));
  return TJ_OK;
}

int tj_kat_planes(tj_ctx* c, int what, int n, const double* P, const double* Q, double dist, double* out) {
  if (!c || n < 0 || what < 0 || what > 7 || !P || !Q || !out) return TJ_ERR_INVALID;
  const size_t qbytes = (what == 0 || what == 2 || what == 5) ? (size_t)n * 24 : (size_t)n * 144;  // what 1, 3, 4, 6: hull vs hull
  DevBuf dp, dq, dout; int r;
  if ((r = to_dev(c, dp, P, (size_t)n * 144)) || (r = to_dev(c, dq, Q, qbytes)) || (r = to_dev(c, dout, nullptr, (size_t)n * 40))) return r;
  HIPCHK(c, hipMemsetAsync(dout.p, 0, std::max<size_t>((size_t)n * 40, 8), c->stream));
  if (what >= 5 && n > 0) { int ur = upload(c, dout.p, out, (size_t)n * 40); if (ur) return ur; }  // in/out: the plane to refine
  if (what == 7) hipLaunchKernelGGL(k_dbg_optpair_wave, dim3(std::max(n, 1)), dim3(64), 0, c->stream, c->d, n, (const double*)dp.p, (const double*)dq.p, (double*)dout.p);
  else if (what == 4) hipLaunchKernelGGL(k_dbg_pair_wave, dim3(std::max(n, 1)), dim3(64), 0, c->stream, c->d, n, (const double*)dp.p, (const double*)dq.p, dist, (double*)dout.p);
  else hipLaunchKernelGGL(k_dbg_planes, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->d, what, n, (const double*)dp.p, (const double*)dq.p, dist, (double*)dout.p);
  HIPCHK(c, hipGetLastError());
  QUIESCE(c);
  HIPCHK(c, hipMemcpy(out, dout.p, (size_t)n * 40, hipMemcpyDeviceToHost));
  return TJ_OK;
}

#endif  // TJ_KAT

// ---- "optimal_plane":1 : host access to the persistent plane tables (teacher-forced parity tests, checkpointing) ----
int tj_get_obs_cache(tj_ctx* c, int u, int seg, int cap, int* ids, double* cd) {
  if (!c || u < 0 || u >= c->d.U || seg < 0 || seg >= c->d.S || cap < 0) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  if (!d.optimal_plane || d.mode != 0) { c->err = "tj_get_obs_cache: needs optimal_plane and TJ_MODE_SINGLE"; return TJ_ERR_INVALID; }
  QUIESCE(c);
  const size_t s = (size_t)u * d.S + seg;
  int n = 0;
  HIPCHK(c, hipMemcpy(&n, d.kobs_n + s, 4, hipMemcpyDeviceToHost));
  const int m = std::min(n, cap);
  if (m > 0 && ids) {
    std::vector<int> tmp(m);
    HIPCHK(c, hipMemcpy(tmp.data(), d.kobs_id + s * d.cap_obs, (size_t)m * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < m; i++) ids[i] = c->cloud_order[tmp[i]];
  }
  if (m > 0 && cd) HIPCHK(c, hipMemcpy(cd, d.kobs_cd + s * d.cap_obs * 4, (size_t)m * 32, hipMemcpyDeviceToHost));
  return n;
}
int tj_set_obs_cache(tj_ctx* c, int u, int seg, int n, const int* ids, const double* cd) {
  if (!c || u < 0 || u >= c->d.U || seg < 0 || seg >= c->d.S || n < 0 || (n > 0 && (!ids || !cd))) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  if (!d.optimal_plane || d.mode != 0) { c->err = "tj_set_obs_cache: needs optimal_plane and TJ_MODE_SINGLE"; return TJ_ERR_INVALID; }
  if (n > d.cap_obs) { c->err = "tj_set_obs_cache: more planes than cap_obs"; return TJ_ERR_CAPACITY; }
  QUIESCE(c);
  std::vector<int> inv(c->cloud_order.size());
  for (size_t i = 0; i < inv.size(); i++) inv[c->cloud_order[i]] = (int)i;
  std::vector<int> tmp(std::max(n, 1));
  for (int i = 0; i < n; i++) {
    if (ids[i] < 0 || ids[i] >= d.N) { c->err = "tj_set_obs_cache: obstacle id out of range"; return TJ_ERR_INVALID; }
    tmp[i] = inv[ids[i]];
  }
  const size_t s = (size_t)u * d.S + seg;
  int r;
  if ((r = upload(c, d.kobs_id + s * d.cap_obs, tmp.data(), (size_t)n * 4)) || (r = upload(c, d.kobs_cd + s * d.cap_obs * 4, cd, (size_t)n * 32)) || (r = upload(c, d.kobs_n + s, &n, 4))) return r;
  return TJ_OK;
}
int tj_get_pair_cache(tj_ctx* c, int* flags, double* cd) {
  if (!c || !flags || !cd) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  if (!d.optimal_plane || d.mode == 0) { c->err = "tj_get_pair_cache: needs optimal_plane and a multi-UAV mode"; return TJ_ERR_INVALID; }
  QUIESCE(c);
  const size_t n = (size_t)d.S * d.U * d.U;
  HIPCHK(c, hipMemcpy(flags, d.kpair_on, n * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(cd, d.kpair_cd, n * 32, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; i++) if (!flags[i]) cd[4 * i] = cd[4 * i + 1] = cd[4 * i + 2] = cd[4 * i + 3] = 0.0;
  return TJ_OK;
}
int tj_set_pair_cache(tj_ctx* c, const int* flags, const double* cd) {
  if (!c || !flags || !cd) return TJ_ERR_INVALID;
  const Dev& d = c->d;
  if (!d.optimal_plane || d.mode == 0) { c->err = "tj_set_pair_cache: needs optimal_plane and a multi-UAV mode"; return TJ_ERR_INVALID; }
  QUIESCE(c);
  const size_t n = (size_t)d.S * d.U * d.U;
  std::vector<int> on(n, 0), list;
  for (int tr = 0; tr < d.S; tr++) for (int a = 0; a < d.U; a++) for (int b = a + 1; b < d.U; b++) {
    const size_t i = ((size_t)tr * d.U + a) * d.U + b;
    const bool mine = (a >= d.u0 && a < d.u1) || (b >= d.u0 && b < d.u1);   // a rank only tracks pairs that touch its robots
    if (flags[i] && mine) { on[i] = 1; list.push_back((int)i); }
  }
  const int cnt = (int)list.size();
  int r;
  if ((r = upload(c, d.kpair_on, on.data(), n * 4)) || (r = upload(c, d.kpair_cd, cd, n * 32)) || (r = upload(c, d.kpair_list, list.data(), (size_t)cnt * 4)) || (r = upload(c, d.kpair_n, &cnt, 4))) return r;  // [1] is re-snapshot by the next k_begin
  return TJ_OK;
}

// ---- initial-trajectory planner --------------------------------------------------------------------------
namespace {
// edge_collision for a batch of edges: cut into pieces no longer than `piece_len`, one wavefront per piece
int edges_hit(tj_ctx* c, int n, const double* edges, int n_prior, const double* prior, double d, double piece_len, std::vector<int>& hit) {
  hit.assign(n, 0);
  if (n == 0) return TJ_OK;
  std::vector<double> pieces; std::vector<int> owner;
  for (int e = 0; e < n; e++) {
    const double* a = edges + 6 * (size_t)e; const double* b = a + 3;
    const double len = std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
    const int k = piece_len > 0 ? std::max(1, (int)std::ceil(len / piece_len)) : 1;
    for (int i = 0; i < k; i++) {
      for (int t = 0; t < 2; t++) {
        const double s = double(i + t) / k;
        for (int x = 0; x < 3; x++) pieces.push_back(i + t == 0 ? a[x] : (i + t == k ? b[x] : a[x] + s * (b[x] - a[x])));
      }
      owner.push_back(e);
    }
  }
  const int np = (int)owner.size();
  DevBuf dp, dow, dpr, dh; int r;
  if ((r = to_dev(c, dp, pieces.data(), pieces.size() * 8)) || (r = to_dev(c, dow, owner.data(), (size_t)np * 4)) || (r = to_dev(c, dpr, prior, (size_t)n_prior * 48)) ||
      (r = to_dev(c, dh, hit.data(), (size_t)n * 4))) return r;
  if (c->d.prim == 3) hipLaunchKernelGGL((k_edge_hit<3>), dim3(np), dim3(64), 0, c->stream, c->d, np, (const double*)dp.p, (const int*)dow.p, n_prior, (const double*)dpr.p, d, (int*)dh.p);
  else hipLaunchKernelGGL((k_edge_hit<1>), dim3(np), dim3(64), 0, c->stream, c->d, np, (const double*)dp.p, (const int*)dow.p, n_prior, (const double*)dpr.p, d, (int*)dh.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(hit.data(), dh.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return check_device_errors(c);
}
double halton(unsigned i, unsigned base) { double f = 1, r = 0; while (i) { f /= base; r += f * (i % base); i /= base; } return r; }
}  // namespace

int tj_edge_collision(tj_ctx* c, int n, const double* edges, int n_prior, const double* prior, double d, int* hit) {
  if (!c || n < 0 || n_prior < 0 || (n > 0 && (!edges || !hit)) || (n_prior > 0 && !prior)) return TJ_ERR_INVALID;
  if (!c->have_cloud) { c->err = "tj_edge_collision: call tj_set_cloud first"; return TJ_ERR_INVALID; }
  QUIESCE(c);
  double ext = 0;
  for (int k = 0; k < 3; k++) ext = std::max(ext, c->cloud_hi[k] - c->cloud_lo[k]);
  std::vector<int> h;
  int r = edges_hit(c, n, edges, n_prior, prior, d, c->d.N > 0 ? ext / 16 : 0.0, h);
  if (r) return r;
  for (int i = 0; i < n; i++) hit[i] = h[i];
  return TJ_OK;
}

int tj_plan_init(tj_ctx* c, int n_robots, const double* starts, const double* goals, double bound_scale, int nodes, int min_waypoints, int cap_waypoints, double* waypoints, int* n_waypoints) {
  if (!c || n_robots < 1 || !starts || !goals || !waypoints || !n_waypoints || cap_waypoints < 3) return TJ_ERR_INVALID;
  if (!c->have_cloud) { c->err = "tj_plan_init: call tj_set_cloud first"; return TJ_ERR_INVALID; }
  QUIESCE(c);
  const double d = c->d.offset + 0.5 * c->d.margin;   // the planner's clearance (OMPL.cpp:74, multiPathPlanning3D.cpp:127)
  if (bound_scale <= 0) bound_scale = c->d.mode == TJ_MODE_SINGLE ? 1.2 : 1.5;   // admmPathPlanning3D.cpp:203-204, multiPathPlanning3D.cpp:216-217
  if (min_waypoints < 3) min_waypoints = 6;
  int K0 = nodes > 0 ? nodes : 254;
  double lo[3], hi[3], ext = 0;
  for (int k = 0; k < 3; k++) {
    lo[k] = bound_scale * c->cloud_lo[k]; hi[k] = bound_scale * c->cloud_hi[k];
    if (c->d.N == 0) { lo[k] = -10; hi[k] = 10; }
    for (int u = 0; u < n_robots; u++) { lo[k] = std::min(lo[k], std::min(starts[3 * u + k], goals[3 * u + k])); hi[k] = std::max(hi[k], std::max(starts[3 * u + k], goals[3 * u + k])); }
    ext = std::max(ext, hi[k] - lo[k]);
  }
  const double piece_len = c->d.N > 0 ? ext / 16 : 0.0;
  std::vector<double> prior;                 // edges of the robots planned so far [.][6]
  std::vector<std::vector<double>> paths(n_robots);
  std::vector<int> hit;
  int r;
  for (int u = 0; u < n_robots; u++) {
    std::vector<double> path;
    for (int K = K0;; K = 2 * K + 2) {
      // roadmap nodes: start, goal, K Halton points of the bounds (deterministic; the reference samples with OMPL's RNG)
      const int V = K + 2;
      std::vector<double> P((size_t)V * 3);
      for (int k = 0; k < 3; k++) { P[k] = starts[3 * u + k]; P[3 + k] = goals[3 * u + k]; }
      for (int i = 0; i < K; i++) { const unsigned h = (unsigned)(i + 1 + 409 * u); P[3 * (i + 2)] = lo[0] + (hi[0] - lo[0]) * halton(h, 2); P[3 * (i + 2) + 1] = lo[1] + (hi[1] - lo[1]) * halton(h, 3); P[3 * (i + 2) + 2] = lo[2] + (hi[2] - lo[2]) * halton(h, 5); }
      // all-pairs visibility with the reference's motion validator, one device batch
      std::vector<double> E; std::vector<int> ea, eb;
      E.reserve((size_t)V * (V - 1) * 3);
      for (int a = 0; a < V; a++) for (int b = a + 1; b < V; b++) { for (int k = 0; k < 3; k++) E.push_back(P[3 * a + k]); for (int k = 0; k < 3; k++) E.push_back(P[3 * b + k]); ea.push_back(a); eb.push_back(b); }
      if ((r = edges_hit(c, (int)ea.size(), E.data(), (int)(prior.size() / 6), prior.data(), d, piece_len, hit))) return r;
      // Dijkstra by Euclidean length, node 0 -> node 1 (dense: V is a few hundred)
      std::vector<double> W((size_t)V * V, INFINITY), dist(V, INFINITY);
      for (size_t e = 0; e < ea.size(); e++) if (!hit[e]) {
        const double* a = &P[3 * ea[e]]; const double* b = &P[3 * eb[e]];
        const double w = std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
        W[(size_t)ea[e] * V + eb[e]] = W[(size_t)eb[e] * V + ea[e]] = w;
      }
      std::vector<int> prev(V, -1); std::vector<char> done(V, 0);
      dist[0] = 0;
      for (int it = 0; it < V; it++) {
        int best = -1;
        for (int v = 0; v < V; v++) if (!done[v] && dist[v] < INFINITY && (best < 0 || dist[v] < dist[best])) best = v;
        if (best < 0 || best == 1) break;
        done[best] = 1;
        for (int v = 0; v < V; v++) if (!done[v] && dist[best] + W[(size_t)best * V + v] < dist[v]) { dist[v] = dist[best] + W[(size_t)best * V + v]; prev[v] = best; }
      }
      if (dist[1] < INFINITY) {
        std::vector<int> idx;
        for (int v = 1; v != -1; v = prev[v]) idx.push_back(v);
        for (auto it = idx.rbegin(); it != idx.rend(); ++it) for (int k = 0; k < 3; k++) path.push_back(P[3 * *it + k]);
        break;
      }
      if (K > 1100) { c->err = "tj_plan_init: no collision-free path for robot " + std::to_string(u) + " (start or goal inside the clearance of an obstacle?)"; return TJ_ERR_NO_PROGRESS; }
    }
    // simplify_path (Main/multiPathPlanning3D.cpp:162-203): greedy shortcutting with the same predicate, one edge at a time
    {
      const int n = (int)path.size() / 3;
      std::vector<char> rm(n, 0);
      int prev = 0, next = 2;
      for (int i = 1; i < n - 1; i++) {
        double e6[6];
        for (int k = 0; k < 3; k++) { e6[k] = path[3 * prev + k]; e6[3 + k] = path[3 * next + k]; }
        if ((r = edges_hit(c, 1, e6, (int)(prior.size() / 6), prior.data(), d, piece_len, hit))) return r;
        if (hit[0]) { prev = i; next += 1; } else { next += 1; rm[i] = 1; }
      }
      std::vector<double> kept;
      for (int i = 0; i < n; i++) if (!rm[i]) for (int k = 0; k < 3; k++) kept.push_back(path[3 * i + k]);
      path.swap(kept);
    }
    // Corner rounding (ours).  The solver's initial control net puts control points at thirds of the polyline edges
    // (init_variable, Main/multiPathPlanning3D.cpp:363-375), so the hull of a piece cuts each corner W by the triangle
    // (W + (A-W)/3, W, W + (B-W)/3).  A fan of chords across that triangle is validated with the same predicate; where it
    // fails the two edges at W are halved by collinear way points, which shrinks the triangle by 2 per round.
    for (int round = 0; round < 6; round++) {
      const int n = (int)path.size() / 3;
      std::vector<double> fan; std::vector<int> corner;
      for (int i = 1; i < n - 1; i++) {
        const double* A = &path[3 * (i - 1)]; const double* W = &path[3 * i]; const double* B = &path[3 * (i + 1)];
        double a[3], b[3], cr[3];
        for (int k = 0; k < 3; k++) { a[k] = W[k] + (A[k] - W[k]) / 3; b[k] = W[k] + (B[k] - W[k]) / 3; }
        cr[0] = (a[1] - W[1]) * (b[2] - W[2]) - (a[2] - W[2]) * (b[1] - W[1]); cr[1] = (a[2] - W[2]) * (b[0] - W[0]) - (a[0] - W[0]) * (b[2] - W[2]); cr[2] = (a[0] - W[0]) * (b[1] - W[1]) - (a[1] - W[1]) * (b[0] - W[0]);
        if (cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2] < 1e-24) continue;   // straight through W: nothing is cut
        for (int sdiv = 1; sdiv <= 4; sdiv++) {
          for (int k = 0; k < 3; k++) fan.push_back(a[k]);
          for (int k = 0; k < 3; k++) fan.push_back(W[k] + (b[k] - W[k]) * sdiv / 4.0);
          corner.push_back(i);
        }
      }
      if (corner.empty()) break;
      if ((r = edges_hit(c, (int)corner.size(), fan.data(), (int)(prior.size() / 6), prior.data(), d, piece_len, hit))) return r;
      std::vector<char> bad(n, 0);
      bool any = false;
      for (size_t e = 0; e < corner.size(); e++) if (hit[e]) { bad[corner[e]] = 1; any = true; }
      if (!any) break;
      std::vector<double> np_;
      for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) np_.push_back(path[3 * i + k]);
        if (i + 1 < n && (bad[i] || bad[i + 1])) for (int k = 0; k < 3; k++) np_.push_back(0.5 * (path[3 * i + k] + path[3 * (i + 1) + k]));
      }
      path.swap(np_);
    }
    for (size_t i = 0; i + 5 < path.size(); i += 3) for (int k = 0; k < 6; k++) prior.push_back(path[i + k]);   // this robot's edges are obstacles for the next
    paths[u] = path;
  }
  // Equal way-point counts.  The reference pads a shorter path with interpolated points between its LAST two way points
  // (Main/multiPathPlanning3D.cpp:297-322), which leaves a cluster of very short pieces that all get the same piece time;
  // here the extra points go to the edges with the longest sub-segments and sit uniformly inside an edge (collinear points:
  // the polyline and its validity are unchanged, the pieces come out as even as the corners allow).
  int max_size = min_waypoints;
  for (auto& p : paths) max_size = std::max(max_size, (int)p.size() / 3);
  if (max_size > cap_waypoints) { c->err = "tj_plan_init: path needs more way points than cap_waypoints"; return TJ_ERR_CAPACITY; }
  for (int u = 0; u < n_robots; u++) {
    std::vector<double>& p = paths[u];
    const int n = (int)p.size() / 3, extra = max_size - n;
    if (extra > 0) {
      // give the extra points to the edges greedily by largest sub-segment length, then place them uniformly inside each edge
      std::vector<double> len(n - 1); std::vector<int> parts(n - 1, 1);
      for (int i = 0; i + 1 < n; i++) len[i] = std::sqrt((p[3 * i] - p[3 * i + 3]) * (p[3 * i] - p[3 * i + 3]) + (p[3 * i + 1] - p[3 * i + 4]) * (p[3 * i + 1] - p[3 * i + 4]) + (p[3 * i + 2] - p[3 * i + 5]) * (p[3 * i + 2] - p[3 * i + 5]));
      for (int k = 0; k < extra; k++) {
        int best = 0;
        for (int i = 1; i + 1 < n; i++) if (len[i] / parts[i] > len[best] / parts[best]) best = i;
        parts[best]++;
      }
      std::vector<double> q;
      for (int i = 0; i + 1 < n; i++)
        for (int j = 0; j < parts[i]; j++)
          for (int k = 0; k < 3; k++) q.push_back(j == 0 ? p[3 * i + k] : p[3 * i + k] + (p[3 * i + 3 + k] - p[3 * i + k]) * (double(j) / parts[i]));
      for (int k = 0; k < 3; k++) q.push_back(p[3 * (n - 1) + k]);
      p.swap(q);
    }
    for (int i = 0; i < max_size * 3; i++) waypoints[(size_t)u * cap_waypoints * 3 + i] = p[i];
  }
  *n_waypoints = max_size;
  return TJ_OK;
}

#ifdef TJ_KAT
int tj_kat_ccd(tj_ctx* c, int n, const double* P, const double* D, const double* Q, const double* E, const double* q, const double* tu, double d, double* out) {
  if (!c || n < 0 || !P || !D || !Q || !E || !q || !tu || !out) return TJ_ERR_INVALID;
  DevBuf b[6], dout; int r;
  const void* src[6] = {P, D, Q, E, q, tu};
  const size_t sz[6] = {(size_t)n * 144, (size_t)n * 144, (size_t)n * 144, (size_t)n * 144, (size_t)n * 24, (size_t)n * 16};
  for (int i = 0; i < 6; i++) if ((r = to_dev(c, b[i], src[i], sz[i]))) return r;
  if ((r = to_dev(c, dout, nullptr, (size_t)n * 16))) return r;
  hipLaunchKernelGGL(k_dbg_ccd, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, (const double*)b[0].p, (const double*)b[1].p, (const double*)b[2].p,
                     (const double*)b[3].p, (const double*)b[4].p, (const double*)b[5].p, d, (double*)dout.p);
  HIPCHK(c, hipGetLastError());
  QUIESCE(c);
  HIPCHK(c, hipMemcpy(out, dout.p, (size_t)n * 16, hipMemcpyDeviceToHost));
  return TJ_OK;
}

int tj_kat_query(tj_ctx* c, int nq, const double* boxes, double margin, int cap, int* counts, int* ids) {
  if (!c || nq < 0 || cap < 1 || !boxes || !counts || !ids) return TJ_ERR_INVALID;
  if (!c->have_cloud) { c->err = "tj_kat_query: set the obstacles first"; return TJ_ERR_INVALID; }
  DevBuf db, di, dn; int r;
  if ((r = to_dev(c, db, boxes, (size_t)nq * 48)) || (r = to_dev(c, di, nullptr, (size_t)nq * cap * 4)) || (r = to_dev(c, dn, nullptr, (size_t)nq * 4))) return r;
  if (c->d.prim == 3) hipLaunchKernelGGL((k_dbg_query<3>), dim3(std::max(nq, 1)), dim3(64), 0, c->stream, c->d, nq, (const double*)db.p, margin, cap, (int*)di.p, (int*)dn.p);
  else hipLaunchKernelGGL((k_dbg_query<1>), dim3(std::max(nq, 1)), dim3(64), 0, c->stream, c->d, nq, (const double*)db.p, margin, cap, (int*)di.p, (int*)dn.p);
  HIPCHK(c, hipGetLastError());
  QUIESCE(c);
  if (nq == 0) return TJ_OK;
  HIPCHK(c, hipMemcpy(counts, dn.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(ids, di.p, (size_t)nq * cap * 4, hipMemcpyDeviceToHost));
  for (int q = 0; q < nq; q++) {
    if (counts[q] > cap) { c->err = "tj_kat_query: a query returned more candidates than cap"; return TJ_ERR_CAPACITY; }
    for (int i = 0; i < counts[q]; i++) ids[(size_t)q * cap + i] = c->cloud_order[ids[(size_t)q * cap + i]];
  }
  return check_device_errors(c);
}

int tj_kat_tri(tj_ctx* c, int n, const double* P, const double* D, const double* tri, const double* t, double dist, double off, double* out) {
  if (!c || n < 0 || !P || !D || !tri || !t || !out) return TJ_ERR_INVALID;
  DevBuf b[4], dout; int r;
  const void* src[4] = {P, D, tri, t};
  const size_t sz[4] = {(size_t)n * 144, (size_t)n * 144, (size_t)n * 72, (size_t)n * 8};
  for (int i = 0; i < 4; i++) if ((r = to_dev(c, b[i], src[i], sz[i]))) return r;
  if ((r = to_dev(c, dout, nullptr, (size_t)n * 64))) return r;
  hipLaunchKernelGGL(k_dbg_tri, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->d, n, (const double*)b[0].p, (const double*)b[1].p, (const double*)b[2].p, (const double*)b[3].p, dist, off, (double*)dout.p);
  HIPCHK(c, hipGetLastError());
  QUIESCE(c);
  HIPCHK(c, hipMemcpy(out, dout.p, (size_t)n * 64, hipMemcpyDeviceToHost));
  return TJ_OK;
}

int tj_kat_linalg(tj_ctx* c, int nmat, int n, const double* mats, double* out) {
  if (!c || nmat < 0 || n < 1 || n > 64 || !mats || !out) return TJ_ERR_INVALID;
  DevBuf dm, dout; int r;
  if ((r = to_dev(c, dm, mats, (size_t)nmat * n * n * 8)) || (r = to_dev(c, dout, nullptr, (size_t)nmat * 16))) return r;
  const size_t lds = (2 * (size_t)n * n + 4 * n) * 8;
  hipLaunchKernelGGL(k_dbg_linalg, dim3(nmat), dim3(64), lds, c->stream, nmat, n, (const double*)dm.p, (double*)dout.p);
  HIPCHK(c, hipGetLastError());
  QUIESCE(c);
  HIPCHK(c, hipMemcpy(out, dout.p, (size_t)nmat * 16, hipMemcpyDeviceToHost));
  return TJ_OK;
}

#endif  // TJ_KAT

int tj_get_build_info(tj_ctx* c, double* bvh_build_ms, int* built_on_device) {
  if (!c) return TJ_ERR_INVALID;
  if (bvh_build_ms) *bvh_build_ms = c->bvh_build_ms;
  if (built_on_device) *built_on_device = c->bvh_on_device;
  return TJ_OK;
}

int tj_get_stats(tj_ctx* c, tj_stats* s) {
  if (!c || !s) return TJ_ERR_INVALID;
  Ctl h;
  QUIESCE(c);
  HIPCHK(c, hipMemcpy(&h, c->d.ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
  const Dev& d = c->d;
  std::vector<unsigned long long> seg((size_t)d.U * d.S * 6);
  HIPCHK(c, hipMemcpy(seg.data(), d.seg_stats, seg.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long tot[6] = {0, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < seg.size(); i++) tot[i % 6] += seg[i];
  s->iters = (unsigned long long)(h.iter + h.pending);
  s->nodes_dcd = tot[0]; s->cand_dcd = tot[1]; s->nodes_ccd = tot[2]; s->cand_ccd = tot[3]; s->planes_obs = tot[4]; s->planes_self = tot[5];
  std::vector<unsigned long long> ps((size_t)d.U * d.S * 2);
  HIPCHK(c, hipMemcpy(ps.data(), d.pair_stats, ps.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long pt[2] = {0, 0};
  for (size_t i = 0; i < ps.size(); i++) pt[i % 2] += ps[i];
  std::vector<unsigned long long> bs((size_t)d.U * d.P + d.U);
  HIPCHK(c, hipMemcpy(bs.data(), d.blk_stats, bs.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long fails = 0, evals = 0;
  for (size_t i = 0; i < (size_t)d.U * d.P; i++) fails += bs[i];
  for (size_t i = (size_t)d.U * d.P; i < bs.size(); i++) evals += bs[i];
  s->energy_evals = evals; s->llt_fail_piece = fails; s->llt_fail_robot = h.llt_fail_robot; s->newton_iters = pt[0]; s->pair_solves = pt[1];
  s->pair_tests = d.mode >= 1 ? s->iters * (unsigned long long)(d.u1 - d.u0) * d.S * d.U : 0;
  s->order_ambiguous = h.order_ambiguous; s->error_bits = h.error; s->order_unresolved = h.order_unresolved;
  s->gjk_max_sum = h.gjk_max_sum + (unsigned long long)h.gjk_max;
  s->ls_giveups = h.ls_giveups; s->ls_helper_timeouts = h.ls_helper_timeouts;
  s->head_starts = h.spec_taken;
  s->async_fallbacks = c->async_fallbacks;
  return TJ_OK;
}

}  // extern "C"

// ---- direct exchange between sharded contexts (Dev::xch; include/trajadmm.h "tj_xch_*") ------------------------------------------------------------
// Each rank owns ONE uncached block: [U][3T] control points | [U][xs] direction records | [2][XCH_MAX] arrival counters.  Peers store into it (same
// process: plain peer access; other processes: hipIpc) from inside their producing kernels; this rank's k_front / k_ccd read it.
namespace {
size_t xch_rx1_off(const Dev& d) { return (size_t)d.U * 3 * d.T; }
size_t xch_cnt_off(const Dev& d) { return xch_rx1_off(d) + (size_t)d.U * d.xs; }
size_t xch_block_bytes(const Dev& d) { return (xch_cnt_off(d) + 2 * XCH_MAX) * sizeof(double); }
}  // namespace

int tj_xch_block(tj_ctx* c, void** base, size_t* bytes) {
  if (!c) return TJ_ERR_INVALID;
  Dev& d = c->d;
  if (!d.xf) { c->err = "tj_xch_block: the direct exchange exists for sharded decoupled contexts (world > 1) only"; return TJ_ERR_UNSUPPORTED; }
  if (d.u1 - d.u0 < 1) { c->err = "tj_xch_block: this rank owns no robot (more ranks than robots)"; return TJ_ERR_UNSUPPORTED; }
  if (d.world > XCH_MAX) { c->err = "tj_xch_block: more than 16 ranks"; return TJ_ERR_UNSUPPORTED; }
  if (!c->xch_block) {
    HIPCHK(c, hipSetDevice(c->prm.device));
    const size_t nb = xch_block_bytes(d);
    void* q = nullptr;
    // uncached: written by a remote GPU (or another process) while this rank's kernels run, read by them with system-scope loads
    if (hipExtMallocWithFlags(&q, nb, hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      if (hipExtMallocWithFlags(&q, nb, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); c->err = "tj_xch_block: hipExtMallocWithFlags (uncached / fine-grained) failed"; return TJ_ERR_DEVICE; }
    }
    HIPCHK(c, hipMemset(q, 0, nb));
    HIPCHK(c, hipDeviceSynchronize());
    c->xch_block = q; c->xch_bytes = nb;
    d.rx[0] = (double*)q; d.rx[1] = (double*)q + xch_rx1_off(d); d.xcnt = (unsigned long long*)((double*)q + xch_cnt_off(d));
  }
  if (base) *base = c->xch_block;
  if (bytes) *bytes = c->xch_bytes;
  return TJ_OK;
}

int tj_xch_ipc_export(tj_ctx* c, void* handle64) {
  if (!c || !handle64) return TJ_ERR_INVALID;
  { int r = tj_xch_block(c, nullptr, nullptr); if (r) return r; }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI hands the handle over as 64 bytes");
  hipIpcMemHandle_t h;
  HIPCHK(c, hipSetDevice(c->prm.device));
  HIPCHK(c, hipIpcGetMemHandle(&h, c->xch_block));
  memcpy(handle64, &h, 64);
  c->xch_ipc_exported = true;
  return TJ_OK;
}

int tj_xch_ipc_open(tj_ctx* c, const void* handle64, void** base) {
  if (!c || !handle64 || !base) return TJ_ERR_INVALID;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  HIPCHK(c, hipSetDevice(c->prm.device));
  void* p = nullptr;
  HIPCHK(c, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  c->xch_ipc_opened.push_back(p);
  *base = p;
  return TJ_OK;
}

int tj_xch_attach(tj_ctx* c, int n_peers, const int* peer_ranks, void* const* peer_bases) {
  if (!c || n_peers < 0 || n_peers >= XCH_MAX || (n_peers > 0 && (!peer_ranks || !peer_bases))) return TJ_ERR_INVALID;
  { int r = tj_xch_block(c, nullptr, nullptr); if (r) return r; }
  Dev& d = c->d;
  if (n_peers != d.world - 1) { c->err = "tj_xch_attach: every other rank of the world must be attached"; return TJ_ERR_INVALID; }
  XchPeers t;
  memset(&t, 0, sizeof(t));
  t.n = n_peers;
  for (int q = 0; q < n_peers; q++) {
    if (peer_ranks[q] < 0 || peer_ranks[q] >= d.world || peer_ranks[q] == d.rank || !peer_bases[q]) { c->err = "tj_xch_attach: bad peer"; return TJ_ERR_INVALID; }
    t.rank[q] = peer_ranks[q];
    double* b = (double*)peer_bases[q];
    t.rx[q][0] = b; t.rx[q][1] = b + xch_rx1_off(d); t.cnt[q] = (unsigned long long*)(b + xch_cnt_off(d));
  }
  QUIESCE(c);
  if (!c->xch_table) { int r = dalloc(c, &c->xch_table, 1); if (r) return r; }
  { int r = upload(c, c->xch_table, &t, sizeof(t)); if (r) return r; }
  d.xp = c->xch_table;
  return TJ_OK;
}

int tj_xch_enable(tj_ctx* c, int on, int wait_mode) {
  if (!c || wait_mode < 0 || wait_mode > 2) return TJ_ERR_INVALID;
  Dev& d = c->d;
  if (on && (!d.xp || !c->xch_block)) { c->err = "tj_xch_enable: tj_xch_attach has not been called"; return TJ_ERR_INVALID; }
  QUIESCE(c);
  drop_graph(c);
  d.xch = on ? 1 : 0; d.xch_poll = (on && wait_mode == 1) ? 1 : 0; c->xch_wait_kernel = on && wait_mode == 0;
  return TJ_OK;
}

#include "tj_group.h"
