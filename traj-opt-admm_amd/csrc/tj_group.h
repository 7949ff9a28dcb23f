// tj_group.h -- several GPUs under ONE host process, inside the library.  Included at the end of tj_api.hip.
//
// The reference is single-process, single-thread; its per-robot loops (Optimization3D_multi.h:29-118) are what shards.  A group
// is one sharded context per rank (robots block-partitioned exactly like tj_params.rank/world), each on its own device and
// stream, driven by one host thread per rank so that the enqueue cost (~3.5 us per launch) does not add up over the ranks.  An
// iteration is the same phase list tj_iterate_phase exposes; after a phase that produces something every rank needs
// (tj_exchange_buffer: control points, direction records, and in coupled mode the Schur-corner terms, CCD exponents and
// Armijo energies) the exchange is
//     k_group_push     the owner writes its slice straight into a receive buffer on EVERY peer (peer-mapped device memory:
//                      stores over xGMI, which is point-to-point -- N-1 small messages leave in parallel on N-1 links; there
//                      is no ring and no collective library on the path), then records an event on its stream;
//     hipStreamWaitEvent on the N-1 peers' events;
//     k_group_unpack   foreign slices from the receive buffer into the buffer the kernels read.
// Receive buffers are double-buffered by exchange parity: a peer can run at most one exchange of the same kind ahead, because
// its next push follows its own wait on everybody's current one.  The payload is <= 30 KB per rank, so the exchange is
// latency, not bandwidth.  Results are bitwise those of one context (tests run ranks on the same device, which the design
// allows: `devices` may repeat).
#pragma once
#include <atomic>
#include <thread>

namespace tj {

constexpr int GROUP_MAX = 16;
struct GroupPeers { double* p[GROUP_MAX]; };

__global__ __launch_bounds__(256) void k_group_push(const double* src, GroupPeers dst, int n_peers, size_t off, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const double v = src[off + i];
    for (int q = 0; q < n_peers; q++) dst.p[q][off + i] = v;
  }
}
// everything outside [own_off, own_off + own_count) of a buffer of `total` doubles
__global__ __launch_bounds__(256) void k_group_unpack(double* dst, const double* rx, size_t own_off, size_t own_count, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    if (i < own_off || i >= own_off + own_count) dst[i] = rx[i];
}

}  // namespace tj

struct tj_group {
  int n = 0;
  std::vector<tj_ctx*> ctx;
  std::vector<int> dev;
  double* rx[tj::GROUP_MAX][5][2] = {};
  hipEvent_t ev[tj::GROUP_MAX][5][2] = {};
  std::atomic<long> recorded[tj::GROUP_MAX][5];   // pushes of buffer `what` rank r has recorded so far
  long issued[5] = {0, 0, 0, 0, 0};               // exchanges of each kind completed by earlier tj_group_iterate calls
  std::atomic<int> abort_flag{0};
  std::string err;
};

namespace {

struct GroupExchangeInfo { double* buf; size_t per; };
GroupExchangeInfo group_buffer(tj_ctx* c, int what) {
  void* p = nullptr; int per = 0;
  tj_exchange_buffer(c, what, &p, &per, nullptr, nullptr);
  return {(double*)p, (size_t)per};
}

// one exchange of buffer `what`, issued by rank r's host thread; s = how many exchanges of this kind came before
int group_exchange(tj_group* g, int r, int what, long s) {
  tj_ctx* c = g->ctx[r];
  const Dev& d = c->d;
  if (g->n == 1) return TJ_OK;   // nothing is foreign
  const int par = (int)(s & 1);
  const GroupExchangeInfo b = group_buffer(c, what);
  const size_t off = (size_t)d.u0 * b.per, cnt = (size_t)(d.u1 - d.u0) * b.per, total = (size_t)d.U * b.per;
  GroupPeers peers; int np = 0;
  for (int q = 0; q < g->n; q++) if (q != r) peers.p[np++] = g->rx[q][what][par];
  if (cnt > 0) hipLaunchKernelGGL(k_group_push, dim3((unsigned)std::min<size_t>((cnt + 255) / 256, 64)), dim3(256), 0, c->stream, b.buf, peers, np, off, cnt);
  HIPCHK(c, hipEventRecord(g->ev[r][what][par], c->stream));
  g->recorded[r][what].store(s + 1, std::memory_order_release);
  for (int q = 0; q < g->n; q++) {
    if (q == r) continue;
    while (g->recorded[q][what].load(std::memory_order_acquire) < s + 1) {   // the peer's host thread has not recorded this push yet
      if (g->abort_flag.load(std::memory_order_relaxed)) { c->err = "group: a peer rank failed"; return TJ_ERR_DEVICE; }
      std::this_thread::yield();
    }
    HIPCHK(c, hipStreamWaitEvent(c->stream, g->ev[q][what][par], 0));
  }
  hipLaunchKernelGGL(k_group_unpack, dim3((unsigned)std::min<size_t>((total + 255) / 256, 64)), dim3(256), 0, c->stream, b.buf, g->rx[r][what][par], off, cnt, total);
  HIPCHK(c, hipGetLastError());
  return TJ_OK;
}

// (phase, buffer exchanged after it or -1): the schedules of traj-opt-admm_amd/sharding.py
const int kGroupDecoupled[][2] = {{0, 0}, {1, 1}, {2, -1}};
const int kGroupCoupled[][2] = {{0, 0}, {1, 2}, {2, 1}, {3, 3}, {4, 4}, {5, -1}};

int group_rank_loop(tj_group* g, int r, int n_iters) {
  tj_ctx* c = g->ctx[r];
  HIPCHK(c, hipSetDevice(g->dev[r]));
  if (!ready(c)) return TJ_ERR_INVALID;
  const bool cpl = c->d.mode == TJ_MODE_MULTI_COUPLED;
  const int (*sched)[2] = cpl ? kGroupCoupled : kGroupDecoupled;
  const int nph = cpl ? 6 : 3;
  long s[5];
  for (int w = 0; w < 5; w++) s[w] = g->issued[w];
  for (int it = 0; it < n_iters; it++)
    for (int k = 0; k < nph; k++) {
      int rc = enqueue_body(c, sched[k][0]);
      if (rc) return rc;
      const int what = sched[k][1];
      if (what >= 0) { rc = group_exchange(g, r, what, s[what]++); if (rc) return rc; }
    }
  return TJ_OK;
}

int group_fail(tj_group* g, int rc, const std::string& m) { g->err = m; return rc; }
std::string g_group_create_err;   // why the last tj_group_create failed (there is no group to ask then)

}  // namespace

extern "C" {

int tj_group_create(const tj_params* p, int n_ranks, const int* devices, tj_group** out) {
  if (!p || !out || n_ranks < 1 || n_ranks > tj::GROUP_MAX) return TJ_ERR_INVALID;
  if (p->mode == TJ_MODE_SINGLE && n_ranks > 1) return TJ_ERR_INVALID;   // one robot does not shard
  tj_group* g = new tj_group();
  g->n = n_ranks;
  for (int r = 0; r < n_ranks; r++) for (int w = 0; w < 5; w++) g->recorded[r][w].store(0);
  auto bail = [&](int rc, const std::string& m) { g_group_create_err = m; tj_group_destroy(g); return rc; };
  for (int r = 0; r < n_ranks; r++) g->dev.push_back(devices ? devices[r] : r);
  // peers write into each other's receive buffers
  for (int a = 0; a < n_ranks; a++) for (int b = 0; b < n_ranks; b++) {
    if (g->dev[a] == g->dev[b]) continue;
    int can = 0;
    if (hipSetDevice(g->dev[a]) != hipSuccess || hipDeviceCanAccessPeer(&can, g->dev[a], g->dev[b]) != hipSuccess || !can) return bail(TJ_ERR_UNSUPPORTED, "no peer access between the devices of the group");
    hipError_t e = hipDeviceEnablePeerAccess(g->dev[b], 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bail(TJ_ERR_DEVICE, "hipDeviceEnablePeerAccess failed");
    (void)hipGetLastError();
  }
  for (int r = 0; r < n_ranks; r++) {
    tj_params q = *p;
    q.rank = r; q.world = n_ranks; q.device = g->dev[r];
    tj_ctx* c = nullptr;
    const int rc = tj_create(&q, &c);
    if (rc) { const std::string m = std::string("rank ") + std::to_string(r) + ": " + tj_last_error(c); if (c) tj_destroy(c); return bail(rc, m); }
    g->ctx.push_back(c);
    const int nwhat = c->d.mode == TJ_MODE_MULTI_COUPLED ? 5 : 2;
    for (int w = 0; w < nwhat; w++) {
      const GroupExchangeInfo b = group_buffer(c, w);
      for (int par = 0; par < 2; par++) {
        if (dalloc(c, &g->rx[r][w][par], (size_t)c->d.U * b.per)) return bail(TJ_ERR_DEVICE, "receive buffer allocation failed");
        if (hipEventCreateWithFlags(&g->ev[r][w][par], hipEventDisableTiming) != hipSuccess) return bail(TJ_ERR_DEVICE, "hipEventCreate failed");
      }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return bail(TJ_ERR_DEVICE, "stream synchronisation failed");
  }
  *out = g;
  return TJ_OK;
}

void tj_group_destroy(tj_group* g) {
  if (!g) return;
  for (size_t r = 0; r < g->ctx.size(); r++) {
    (void)hipSetDevice(g->dev[r]);
    (void)hipStreamSynchronize(g->ctx[r]->stream);
  }
  for (size_t r = 0; r < g->ctx.size(); r++) {
    (void)hipSetDevice(g->dev[r]);
    for (int w = 0; w < 5; w++) for (int par = 0; par < 2; par++) if (g->ev[r][w][par]) (void)hipEventDestroy(g->ev[r][w][par]);
    tj_destroy(g->ctx[r]);
  }
  delete g;
}

int tj_group_size(tj_group* g) { return g ? g->n : TJ_ERR_INVALID; }
tj_ctx* tj_group_ctx(tj_group* g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }
const char* tj_group_last_error(tj_group* g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }

#define GROUP_EACH(g, call)                                                                                     \
  for (int r_ = 0; r_ < (g)->n; r_++) {                                                                         \
    tj_ctx* c = (g)->ctx[r_];                                                                                   \
    if (hipSetDevice((g)->dev[r_]) != hipSuccess) return group_fail(g, TJ_ERR_DEVICE, "hipSetDevice failed");   \
    const int rc_ = (call);                                                                                     \
    if (rc_ < 0) return group_fail(g, rc_, std::string("rank ") + std::to_string(r_) + ": " + tj_last_error(c)); \
  }

int tj_group_set_cloud(tj_group* g, const double* points, int n) { if (!g) return TJ_ERR_INVALID; GROUP_EACH(g, tj_set_cloud(c, points, n)); return TJ_OK; }
int tj_group_set_mesh(tj_group* g, const double* verts, int n_verts, const int* tris, int n_tris) { if (!g) return TJ_ERR_INVALID; GROUP_EACH(g, tj_set_mesh(c, verts, n_verts, tris, n_tris)); return TJ_OK; }
int tj_group_init_state(tj_group* g, const double* waypoints, double piece_time) { if (!g) return TJ_ERR_INVALID; GROUP_EACH(g, tj_init_state(c, waypoints, piece_time)); return TJ_OK; }

int tj_group_iterate(tj_group* g, int n_iters, double* gnorm, int* iters_total, int* converged) {
  if (!g || n_iters < 0) return TJ_ERR_INVALID;
  g->abort_flag.store(0);
  std::vector<int> rc(g->n, TJ_OK);
  if (g->n == 1) rc[0] = group_rank_loop(g, 0, n_iters);
  else {
    std::vector<std::thread> th;
    for (int r = 0; r < g->n; r++) th.emplace_back([g, r, n_iters, &rc]() { rc[r] = group_rank_loop(g, r, n_iters); if (rc[r]) g->abort_flag.store(1); });
    for (auto& t : th) t.join();
  }
  {  // what the schedule exchanged, for the parity of the next call
    const bool cpl = g->ctx[0]->d.mode == TJ_MODE_MULTI_COUPLED;
    for (int w = 0; w < (cpl ? 5 : 2); w++) g->issued[w] += n_iters;
  }
  for (int r = 0; r < g->n; r++) if (rc[r]) return group_fail(g, rc[r], std::string("rank ") + std::to_string(r) + ": " + tj_last_error(g->ctx[r]));
  Ctl h0; memset(&h0, 0, sizeof(h0));
  for (int r = 0; r < g->n; r++) {
    tj_ctx* c = g->ctx[r];
    if (hipSetDevice(g->dev[r]) != hipSuccess) return group_fail(g, TJ_ERR_DEVICE, "hipSetDevice failed");
    int e = flush_deferred(c);
    Ctl h;
    if (!e) e = check_device_errors(c, &h);
    if (e) return group_fail(g, e, std::string("rank ") + std::to_string(r) + ": " + tj_last_error(c));
    if (r == 0) h0 = h;   // gnorm, the iteration counter and the stop flag are formed identically on every rank
  }
  const int it = h0.iter + h0.pending;
  if (gnorm) *gnorm = h0.gnorm;
  if (iters_total) *iters_total = it;
  if (converged) *converged = h0.done || (g->ctx[0]->d.stop > 0 && it > 1 && h0.gnorm < g->ctx[0]->d.stop);
  return TJ_OK;
}

int tj_group_get_state(tj_group* g, int u, double* spline, double* p_slack, double* p_lambda, double* t_slack, double* t_lambda, double* piece_time) {
  if (!g) return TJ_ERR_INVALID;
  for (int r = 0; r < g->n; r++) {
    tj_ctx* c = g->ctx[r];
    if (u >= c->d.u0 && u < c->d.u1) {
      if (hipSetDevice(g->dev[r]) != hipSuccess) return group_fail(g, TJ_ERR_DEVICE, "hipSetDevice failed");
      const int rc = tj_get_state(c, u, spline, p_slack, p_lambda, t_slack, t_lambda, piece_time);
      if (rc < 0) g->err = tj_last_error(c);
      return rc;
    }
  }
  return group_fail(g, TJ_ERR_INVALID, "tj_group_get_state: no such robot");
}

}  // extern "C"
