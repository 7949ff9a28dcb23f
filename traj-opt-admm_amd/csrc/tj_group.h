// tj_group.h -- several GPUs under ONE host process, inside the library.  Included at the end of tj_api.hip.
//
// The reference is single-process, single-thread; its per-robot loops (Optimization3D_multi.h:29-118) are what shards.  A group
// is one sharded context per rank (robots block-partitioned exactly like tj_params.rank/world), each on its own device and
// stream, driven by one host thread per rank so that the enqueue cost (~3.5 us per launch) does not add up over the ranks.  An
// iteration is the same phase list tj_iterate_phase exposes; after a phase that produces something every rank needs
// (tj_exchange_buffer: control points, direction records, and in coupled mode the Schur-corner terms, CCD exponents and
// Armijo energies) the slices are exchanged by one of three TRANSPORTS (TJ_GROUP_TRANSPORT=flag|event|rccl, or
// tj_group_set_transport):
//
//   flag   (opt-in until it has run across xGMI; the fastest with ranks on one device) device-to-device, no host on the path, no event, no collective
//          library.  k_group_push: the owner stores its slice straight into a receive buffer on EVERY peer (peer-mapped memory:
//          N-1 point-to-point xGMI transfers leave in parallel, there is no ring), fences at system scope and then stores the
//          exchange's sequence number into its flag word on every peer.  k_group_wait_unpack, next on the consumer's own stream:
//          one lane per peer polls that peer's flag word (system-scope acquire; wall-clock timeout -> ERR_PEER_TIMEOUT, never a
//          hang), then the block moves the foreign slices into the buffer the kernels read.  Two small launches per exchange.
//   event  (the default: plain HIP stream / event semantics; ranks sharing a GPU may share a hardware queue, where the flag
//          transport's polling kernel could sit in front of the very push it waits for) k_group_push, hipEventRecord; the peers' host threads wait for the record to
//          exist and make their streams wait on the event (hipStreamWaitEvent), then k_group_unpack.
//   rccl   the collective north_star names, driven from host C++: ncclCommInitAll over the group's devices, one in-place
//          ncclAllGather per exchange on each rank's solver stream (called by that rank's host thread).  librccl.so is opened
//          at run time (dlopen) only when this transport is asked for -- the library has no link-time dependency on it.
//          Needs distinct devices and U divisible by the number of ranks (equal slices); refused otherwise.
//
// Receive buffers and flag words are UNCACHED device memory (hipExtMallocWithFlags(hipDeviceMallocUncached)): they are
// written by a remote GPU and read by a later local kernel, and the same-parity buffer was read two exchanges earlier, so
// ordinary cached memory could serve stale lines; the events of the event transport release to system scope.  Buffers are
// double-buffered by exchange parity: a peer can run at most one exchange of the same kind ahead, because its next push
// follows its own wait on everybody's current one.  The payload is <= 30 KB per rank, so the exchange is latency, not
// bandwidth.  Results are bitwise those of one context (tests run ranks on the same device, which the design allows: `devices`
// may repeat).  A group whose devices are all distinct has NOT run on hardware yet (no multi-GPU box was available in rounds
// 1-3): tests/test_gpu_group.py::test_group_on_two_devices runs wherever two devices are visible.
#pragma once
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <dlfcn.h>

namespace tj {

constexpr int GROUP_MAX = 16;
struct GroupPeers { double* p[GROUP_MAX]; unsigned long long* flag[GROUP_MAX]; };

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

// flag transport.  ONE block: its barrier is what orders "every thread's stores are fenced" before the flags go out.
constexpr int GROUP_FLAG_THREADS = 1024;
__global__ __launch_bounds__(GROUP_FLAG_THREADS) void k_group_push_flag(const double* src, GroupPeers dst, int n_peers, size_t off, size_t count, unsigned long long seq) {
  for (size_t i = threadIdx.x; i < count; i += GROUP_FLAG_THREADS) {
    const double v = src[off + i];
    for (int q = 0; q < n_peers; q++) dst.p[q][off + i] = v;
  }
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < n_peers) __hip_atomic_store(dst.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr long long GROUP_FLAG_TIMEOUT_TICKS = 200000000ll;   // 2 s of the 100 MHz wall clock: a lost peer must not hang the device
// flags: this rank's words, one per peer (already offset to the exchange's kind and parity)
__global__ __launch_bounds__(GROUP_FLAG_THREADS) void k_group_wait_unpack(double* dst, const double* rx, GroupPeers mine, int n_peers, unsigned long long seq, size_t own_off, size_t own_count, size_t total, Ctl* ctl) {
  __shared__ int s_lost;
  if (threadIdx.x == 0) s_lost = 0;
  __syncthreads();
  if ((int)threadIdx.x < n_peers) {
    const long long t_end = wall_clock64() + GROUP_FLAG_TIMEOUT_TICKS;
    while (__hip_atomic_load(mine.flag[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      if (wall_clock64() > t_end) { atomicOr(&ctl->error, ERR_PEER_TIMEOUT); s_lost = 1; break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (s_lost) return;   // a peer's slice never arrived: leave the buffer as it is (the error bit fails the batch) rather than unpack stale data
  for (size_t i = threadIdx.x; i < total; i += GROUP_FLAG_THREADS)
    if (i < own_off || i >= own_off + own_count) dst[i] = __builtin_nontemporal_load(rx + i);
}

}  // namespace tj

// ---- RCCL, resolved at run time ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;                                         // ncclCommInitAll
  int (*CommDestroy)(void* comm) = nullptr;                                                                          // ncclCommDestroy
  int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;  // ncclAllGather
  const char* (*GetErrorString)(int) = nullptr;                                                                      // ncclGetErrorString
  int (*CommCount)(void* comm, int* count) = nullptr;                                                                // ncclCommCount (optional: reporting only)
  int (*Broadcast)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t stream) = nullptr;   // ncclBroadcast  } optional: unequal slices
  int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;                                                        // ncclGroupStart / ncclGroupEnd  } (uav_num % ranks != 0)
  bool uneven_ok() const { return Broadcast && GroupStart && GroupEnd; }
  std::string err;                                                                                                   // why the library could not be bound (taken once, where dlopen failed)
  bool ok() const { return lib && CommInitAll && CommDestroy && AllGather && GetErrorString; }
};
constexpr int kNcclFloat64 = 8;   // ncclDouble / ncclFloat64 (rccl.h ncclDataType_t)
RcclApi& rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
      const char* e = dlerror();   // one call: dlerror() clears the pending message
      if (!api.err.empty()) api.err += "; ";
      api.err += e ? e : (std::string(n) + ": dlopen failed");
    }
    if (!api.lib) return;
    api.err.clear();
    api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
    api.Broadcast = (decltype(api.Broadcast))dlsym(api.lib, "ncclBroadcast");
    api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
    if (!api.ok()) api.err = "librccl.so lacks ncclCommInitAll / ncclCommDestroy / ncclAllGather / ncclGetErrorString";
  });
  return api;
}
}  // namespace

enum { TJ_TRANSPORT_EVENT = 0, TJ_TRANSPORT_FLAG = 1, TJ_TRANSPORT_RCCL = 2 };

struct tj_group {
  int n = 0;
  int lsc_followed = 0;                           // coupled mode: batches that were run again with the Armijo search followed beyond the exchanged candidates (tj_group_iterate)
  int transport = TJ_TRANSPORT_EVENT;
  bool distinct = false;                          // every rank on its own device
  std::vector<tj_ctx*> ctx;
  std::vector<int> dev;
  double* rx[tj::GROUP_MAX][5][2] = {};           // uncached receive buffers
  unsigned long long* flags[tj::GROUP_MAX] = {};  // uncached [5][2][GROUP_MAX]: sequence number of the last push of (what, parity) by each peer
  hipEvent_t ev[tj::GROUP_MAX][5][2] = {};
  std::atomic<long> recorded[tj::GROUP_MAX][5];   // pushes of buffer `what` rank r has recorded so far (event transport)
  long issued[5] = {0, 0, 0, 0, 0};               // exchanges of each kind completed by earlier tj_group_iterate calls
  std::atomic<int> abort_flag{0};
  bool poisoned = false;                          // a rank failed mid-batch: the ranks' exchange counts no longer agree
  std::vector<void*> comms;                       // ncclComm_t per rank (rccl transport)
  std::vector<void*> uncached;                    // hipExtMallocWithFlags allocations, with their device
  std::vector<int> uncached_dev;
  std::string err;
};

namespace {

struct GroupExchangeInfo { double* buf; size_t per; };
GroupExchangeInfo group_buffer(tj_ctx* c, int what) {
  void* p = nullptr; int per = 0;
  tj_exchange_buffer(c, what, &p, &per, nullptr, nullptr);
  return {(double*)p, (size_t)per};
}
inline unsigned long long* group_flag(tj_group* g, int rank, int what, int par, int src) { return g->flags[rank] + ((size_t)(what * 2 + par) * tj::GROUP_MAX + src); }

// one exchange of buffer `what`, issued by rank r's host thread; s = how many exchanges of this kind came before
int group_exchange(tj_group* g, int r, int what, long s) {
  tj_ctx* c = g->ctx[r];
  const Dev& d = c->d;
  if (g->n == 1) return TJ_OK;   // nothing is foreign
  const int par = (int)(s & 1);
  const GroupExchangeInfo b = group_buffer(c, what);
  const size_t off = (size_t)d.u0 * b.per, cnt = (size_t)(d.u1 - d.u0) * b.per, total = (size_t)d.U * b.per;
  if (g->transport == TJ_TRANSPORT_RCCL) {   // in place: this rank's slice already sits at its offset of the full buffer
    RcclApi& api = rccl_api();
    c->launches++;   // (the collective's kernel)
    if (d.U % g->n == 0) {
      const int rc = api.AllGather(b.buf + off, b.buf, cnt, kNcclFloat64, g->comms[r], c->stream);
      if (rc != 0) { c->err = std::string("ncclAllGather: ") + api.GetErrorString(rc); return TJ_ERR_DEVICE; }
      return TJ_OK;
    }
    // unequal slices (uav_num not divisible by the ranks): one in-place broadcast per owner, fused into one group call
    int rc = api.GroupStart();
    for (int q = 0; q < g->n && rc == 0; q++) {
      const size_t qo = (size_t)((long long)q * d.U / g->n) * b.per, qc = (size_t)((long long)(q + 1) * d.U / g->n) * b.per - qo;
      if (qc > 0) rc = api.Broadcast(b.buf + qo, b.buf + qo, qc, kNcclFloat64, q, g->comms[r], c->stream);
    }
    const int rc2 = api.GroupEnd();
    if (rc == 0) rc = rc2;
    if (rc != 0) { c->err = std::string("ncclBroadcast (unequal slices): ") + api.GetErrorString(rc); return TJ_ERR_DEVICE; }
    return TJ_OK;
  }
  GroupPeers peers; int np = 0;
  for (int q = 0; q < g->n; q++) if (q != r) { peers.p[np] = g->rx[q][what][par]; peers.flag[np] = group_flag(g, q, what, par, r); np++; }
  if (g->transport == TJ_TRANSPORT_FLAG) {
    GroupPeers mine; int nm = 0;
    for (int q = 0; q < g->n; q++) if (q != r) { mine.p[nm] = nullptr; mine.flag[nm] = group_flag(g, r, what, par, q); nm++; }
    TJ_LAUNCH(k_group_push_flag, dim3(1), dim3(GROUP_FLAG_THREADS), 0, c->stream, b.buf, peers, np, off, cnt, (unsigned long long)(s + 1));
    TJ_LAUNCH(k_group_wait_unpack, dim3(1), dim3(GROUP_FLAG_THREADS), 0, c->stream, b.buf, g->rx[r][what][par], mine, nm, (unsigned long long)(s + 1), off, cnt, total, d.ctl);
    HIPCHK(c, hipGetLastError());
    return TJ_OK;
  }
  const bool direct = d.xch != 0;   // decoupled mode: the producing kernel itself has stored this rank's slice into the peers' receive blocks (kernels_step.h) -- the event
                                    // only orders the consumer's stream behind it, and the foreign units of its next kernel put the slices in place
  if (cnt > 0 && !direct) TJ_LAUNCH(k_group_push, dim3((unsigned)std::min<size_t>((cnt + 255) / 256, 64)), dim3(256), 0, c->stream, b.buf, peers, np, off, cnt);
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
  if (!direct) TJ_LAUNCH(k_group_unpack, dim3((unsigned)std::min<size_t>((total + 255) / 256, 64)), dim3(256), 0, c->stream, b.buf, g->rx[r][what][par], off, cnt, total);
  HIPCHK(c, hipGetLastError());
  return TJ_OK;
}

// (phase, buffer exchanged after it or -1): the schedules of traj-opt-admm_amd/sharding.py
const int kGroupDecoupled[][2] = {{0, 0}, {1, 1}, {2, -1}};
const int kGroupCoupled[][2] = {{0, 0}, {1, 2}, {2, 1}, {3, 3}, {4, 4}, {5, -1}};

// careful (coupled mode, the re-run of a batch whose Armijo search left the candidates one exchange carries): the rank follows the search -- after phase 5 it asks
// tj_coupled_search_pending (one host look per iteration) and repeats phase 4 / exchange 4 / phase 5 while the answer is yes; *extra4 = the additional exchanges of
// buffer 4 it issued (the same number on every rank: same gathered tables, same decisions)
int group_rank_loop(tj_group* g, int r, int n_iters, bool careful = false, long* extra4 = nullptr) {
  tj_ctx* c = g->ctx[r];
  HIPCHK(c, hipSetDevice(g->dev[r]));
  if (!ready(c)) return TJ_ERR_INVALID;
  const bool cpl = c->d.mode == TJ_MODE_MULTI_COUPLED;
  const int (*sched)[2] = cpl ? kGroupCoupled : kGroupDecoupled;
  const int nph = cpl ? 6 : 3;
  // direct exchange (flag transport, decoupled mode): nothing between the launches -- the producing kernels push, the consuming kernels wait:
  // the fused six-kernel chain of one context
  if (g->n > 1 && g->transport == TJ_TRANSPORT_FLAG && c->d.xch) return tj_iterate_async(c, n_iters);
  long s[5];
  for (int w = 0; w < 5; w++) s[w] = g->issued[w];
  for (int it = 0; it < n_iters; it++) {
    for (int k = 0; k < nph; k++) {
      int rc = tj_iterate_phase_chained(c, sched[k][0], it + 1 < n_iters);   // (decoupled: the next iteration's begin rides in this one's k_linesearch)
      if (rc) return rc;
      const int what = sched[k][1];
      if (what >= 0) { rc = group_exchange(g, r, what, s[what]++); if (rc) return rc; }
    }
    if (careful && cpl)
      for (;;) {
        int pending = 0;
        int rc = tj_coupled_search_pending(c, &pending); if (rc) return rc;
        if (!pending) break;
        if ((rc = tj_iterate_phase_chained(c, 4, 0)) || (rc = group_exchange(g, r, 4, s[4]++)) || (rc = tj_iterate_phase_chained(c, 5, 0))) return rc;
        if (extra4) (*extra4)++;
      }
  }
  return TJ_OK;
}

int group_fail(tj_group* g, int rc, const std::string& m) { g->err = m; return rc; }
thread_local std::string g_group_create_err;   // why this thread's last tj_group_create failed (there is no group to ask then)

const char* transport_name(int t) { return t == TJ_TRANSPORT_FLAG ? "flag" : (t == TJ_TRANSPORT_RCCL ? "rccl" : "event"); }

// (re)select the transport; rccl communicators are created on first selection
int group_select_transport(tj_group* g, int t) {
  if (t == TJ_TRANSPORT_RCCL) {
    if (!g->distinct) return group_fail(g, TJ_ERR_UNSUPPORTED, "transport rccl needs every rank on its own device (RCCL refuses two ranks on one GPU)");
    RcclApi& api = rccl_api();
    if (!api.ok()) return group_fail(g, TJ_ERR_DEVICE, std::string("transport rccl: librccl.so could not be bound: ") + (api.err.empty() ? std::string("symbols missing") : api.err));
    if (g->ctx[0]->d.U % g->n != 0 && !api.uneven_ok()) return group_fail(g, TJ_ERR_UNSUPPORTED, "transport rccl: the robot count is not divisible by the number of ranks and this librccl.so lacks ncclBroadcast / ncclGroupStart / ncclGroupEnd (unequal slices go out as one grouped broadcast per owner)");
    if (g->comms.empty()) {
      g->comms.assign(g->n, nullptr);
      const int rc = api.CommInitAll(g->comms.data(), g->n, g->dev.data());
      if (rc != 0) { g->comms.clear(); return group_fail(g, TJ_ERR_DEVICE, std::string("ncclCommInitAll: ") + api.GetErrorString(rc)); }
    }
  }
  // flag transport, decoupled mode: the DIRECT exchange (in-kernel pushes and waits, csrc/kernels_step.h).  Ranks that share a device do not poll inside
  // k_front / k_ccd (their waiting waves would hold the LDS the peer's producing kernel needs): a one-wave wait launch in front instead.
  // TJ_XCH_POLL=1 / 0 overrides (test hook: small fleets can poll on a shared device).
  for (int r = 0; r < g->n; r++) {
    tj_ctx* c = g->ctx[r];
    if (!c->d.xf || !c->d.xp) continue;
    (void)hipSetDevice(g->dev[r]);
    int sharers = 0;
    for (int b = 0; b < g->n; b++) sharers += g->dev[b] == g->dev[r] ? 1 : 0;
    // ranks sharing a device: k_linesearch's helper blocks (one compute unit each) only where the ranks run in lockstep (flag: the kernels of the ranks overlap and every
    // helper is resident; measured with 2 x 32 robots on one device: 0.147 -> 0.137 ms) -- under the event transport, which orders whole kernels, the helpers of one rank
    // queue behind the other rank's blocks and the primaries run into their 10 us give-up (0.166 -> 0.279 ms)
    if (sharers > 1 && !tune("LS_HELP") && c->d.ls_fast && c->d.mode != TJ_MODE_MULTI_COUPLED) {
      const int owned = std::max(1, c->d.u1 - c->d.u0);
      c->d.ls_help = t == TJ_TRANSPORT_FLAG ? std::max(1, std::min(LS_HELP_MAX, c->d.num_cu / (owned * sharers))) : 1;
    }
    int wait_mode = sharers == 1 ? 1 : 0;
    if (const char* e = tune("XCH_POLL")) wait_mode = atoi(e) != 0 ? 1 : 0;
    if (t == TJ_TRANSPORT_EVENT) wait_mode = 2;   // events order the streams
    const int rc = tj_xch_enable(c, t != TJ_TRANSPORT_RCCL ? 1 : 0, wait_mode);
    if (rc) return group_fail(g, rc, std::string("rank ") + std::to_string(r) + ": " + tj_last_error(c));
  }
  g->transport = t;
  return TJ_OK;
}

}  // namespace

extern "C" {

int tj_rccl_available(void) { return rccl_api().ok() ? 1 : 0; }   // librccl.so opens and exports the four entry points the rccl transport binds (no GPU needed)

int tj_group_create(const tj_params* p, int n_ranks, const int* devices, tj_group** out) {
  if (!p || !out || n_ranks < 1 || n_ranks > tj::GROUP_MAX) return TJ_ERR_INVALID;
  if (p->mode == TJ_MODE_SINGLE && n_ranks > 1) return TJ_ERR_INVALID;   // one robot does not shard
  tj_group* g = new tj_group();
  g->n = n_ranks;
  for (int r = 0; r < n_ranks; r++) for (int w = 0; w < 5; w++) g->recorded[r][w].store(0);
  auto bail = [&](int rc, const std::string& m) { g_group_create_err = m; tj_group_destroy(g); return rc; };
  for (int r = 0; r < n_ranks; r++) g->dev.push_back(devices ? devices[r] : r);
  g->distinct = true;
  for (int a = 0; a < n_ranks; a++) for (int b = a + 1; b < n_ranks; b++) if (g->dev[a] == g->dev[b]) g->distinct = false;
  // peers write into each other's receive buffers
  for (int a = 0; a < n_ranks; a++) for (int b = 0; b < n_ranks; b++) {
    if (g->dev[a] == g->dev[b]) continue;
    int can = 0;
    if (hipSetDevice(g->dev[a]) != hipSuccess || hipDeviceCanAccessPeer(&can, g->dev[a], g->dev[b]) != hipSuccess || !can) return bail(TJ_ERR_UNSUPPORTED, "no peer access between the devices of the group");
    hipError_t e = hipDeviceEnablePeerAccess(g->dev[b], 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bail(TJ_ERR_DEVICE, "hipDeviceEnablePeerAccess failed");
    (void)hipGetLastError();
  }
  auto ualloc = [&](int dev, size_t bytes, void** out_p) -> bool {   // uncached device memory, zeroed
    if (hipSetDevice(dev) != hipSuccess) return false;
    void* q = nullptr;
    if (hipExtMallocWithFlags(&q, std::max<size_t>(bytes, 8), hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); return false; }
    g->uncached.push_back(q); g->uncached_dev.push_back(dev);
    if (hipMemset(q, 0, std::max<size_t>(bytes, 8)) != hipSuccess) return false;
    *out_p = q;
    return true;
  };
  for (int r = 0; r < n_ranks; r++) {
    tj_params q = *p;
    q.rank = r; q.world = n_ranks; q.device = g->dev[r];
    tj_ctx* c = nullptr;
    const int rc = tj_create(&q, &c);
    if (rc) { const std::string m = std::string("rank ") + std::to_string(r) + ": " + tj_last_error(c); if (c) tj_destroy(c); return bail(rc, m); }
    g->ctx.push_back(c);
    {   // ranks that share a device (a test arrangement) share its compute units too: k_linesearch's helper blocks (one CU each) and k_grad's launch order assume
        // a device of their own -- off for those ranks (same bits either way)
      int sharers = 0;
      for (int b = 0; b < n_ranks; b++) sharers += g->dev[b] == g->dev[r] ? 1 : 0;
      if (sharers > 1) { c->d.ls_help = 1; c->d.grad_bal = 0; c->lsc_wide = false; }   // (ls_help: re-decided per transport, group_select_transport)
    }
    const int nwhat = c->d.mode == TJ_MODE_MULTI_COUPLED ? 5 : 2;
    for (int w = 0; w < nwhat; w++) {
      const GroupExchangeInfo b = group_buffer(c, w);
      for (int par = 0; par < 2; par++) {
        if (!ualloc(g->dev[r], (size_t)c->d.U * b.per * sizeof(double), (void**)&g->rx[r][w][par])) return bail(TJ_ERR_DEVICE, "receive buffer allocation failed (hipExtMallocWithFlags, uncached)");
        if (hipEventCreateWithFlags(&g->ev[r][w][par], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) return bail(TJ_ERR_DEVICE, "hipEventCreate failed");
      }
    }
    if (!ualloc(g->dev[r], sizeof(unsigned long long) * 5 * 2 * tj::GROUP_MAX, (void**)&g->flags[r])) return bail(TJ_ERR_DEVICE, "flag allocation failed (hipExtMallocWithFlags, uncached)");
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return bail(TJ_ERR_DEVICE, "stream synchronisation failed");
  }
  // direct exchange (decoupled mode): every rank's receive block, attached to every other rank (peer access is enabled above)
  if (n_ranks > 1 && g->ctx[0]->d.xf) {
    bool ok = true;
    std::vector<void*> base(n_ranks, nullptr);
    for (int r = 0; r < n_ranks && ok; r++) { (void)hipSetDevice(g->dev[r]); ok = tj_xch_block(g->ctx[r], &base[r], nullptr) == TJ_OK; }
    for (int r = 0; r < n_ranks && ok; r++) {
      std::vector<int> pr; std::vector<void*> pb;
      for (int q = 0; q < n_ranks; q++) if (q != r) { pr.push_back(q); pb.push_back(base[q]); }
      (void)hipSetDevice(g->dev[r]);
      ok = tj_xch_attach(g->ctx[r], (int)pr.size(), pr.data(), pb.data()) == TJ_OK;
    }
    // (a group that cannot have it -- more ranks than robots -- keeps the legacy flag kernels; the other transports do not need it)
    if (!ok) for (int r = 0; r < n_ranks; r++) g->ctx[r]->d.xp = nullptr;
  }
  // Default: event, on distinct devices too.  The flag transport (no host, no event on the path) is the fastest one measured with the ranks
  // on one device, but it has not crossed xGMI yet: it is opt-in (TJ_GROUP_TRANSPORT=flag / tj_group_set_transport) until
  // test_group_on_two_devices[flag] has passed on a multi-GPU box; bench.py tries flag -> event -> rccl and validates each bitwise.
  int t = TJ_TRANSPORT_EVENT;
  if (const char* e = tune("GROUP_TRANSPORT")) {
    if (!strcmp(e, "flag")) t = TJ_TRANSPORT_FLAG; else if (!strcmp(e, "event")) t = TJ_TRANSPORT_EVENT; else if (!strcmp(e, "rccl")) t = TJ_TRANSPORT_RCCL;
    else return bail(TJ_ERR_INVALID, std::string("TJ_GROUP_TRANSPORT=") + e + ": expected flag, event or rccl");
  }
  if (n_ranks > 1) { const int rc = group_select_transport(g, t); if (rc) { const std::string m = g->err; return bail(rc, m); } }
  else g->transport = t == TJ_TRANSPORT_RCCL ? TJ_TRANSPORT_EVENT : t;
  *out = g;
  return TJ_OK;
}

void tj_group_destroy(tj_group* g) {
  if (!g) return;
  for (size_t r = 0; r < g->ctx.size(); r++) {
    (void)hipSetDevice(g->dev[r]);
    (void)hipStreamSynchronize(g->ctx[r]->stream);
  }
  for (void* cm : g->comms) if (cm) (void)rccl_api().CommDestroy(cm);
  for (size_t r = 0; r < g->ctx.size(); r++) {
    (void)hipSetDevice(g->dev[r]);
    for (int w = 0; w < 5; w++) for (int par = 0; par < 2; par++) if (g->ev[r][w][par]) (void)hipEventDestroy(g->ev[r][w][par]);
    tj_destroy(g->ctx[r]);
  }
  for (size_t i = 0; i < g->uncached.size(); i++) { (void)hipSetDevice(g->uncached_dev[i]); (void)hipFree(g->uncached[i]); }
  delete g;
}

int tj_group_size(tj_group* g) { return g ? g->n : TJ_ERR_INVALID; }
tj_ctx* tj_group_ctx(tj_group* g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }
const char* tj_group_last_error(tj_group* g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }
const char* tj_group_transport(tj_group* g) { return g ? transport_name(g->transport) : ""; }
// ranks the group's RCCL communicator reports (ncclCommCount of rank 0's comm); 0 unless the rccl transport is selected and initialised
int tj_group_rccl_ranks(tj_group* g) {
  if (!g) return TJ_ERR_INVALID;
  if (g->transport != TJ_TRANSPORT_RCCL || g->comms.empty() || !g->comms[0] || !rccl_api().CommCount) return 0;
  int n = 0;
  return rccl_api().CommCount(g->comms[0], &n) == 0 ? n : 0;
}
int tj_group_set_transport(tj_group* g, const char* name) {
  if (!g || !name) return TJ_ERR_INVALID;
  if (g->poisoned) return group_fail(g, TJ_ERR_DEVICE, "group is poisoned by an earlier failure: " + g->err);
  int t;
  if (!strcmp(name, "flag")) t = TJ_TRANSPORT_FLAG; else if (!strcmp(name, "event")) t = TJ_TRANSPORT_EVENT; else if (!strcmp(name, "rccl")) t = TJ_TRANSPORT_RCCL;
  else return group_fail(g, TJ_ERR_INVALID, std::string("unknown transport ") + name);
  // the exchange counters are shared by the transports, but flag words / events of exchanges done under another transport
  // were never written: only switch between batches, and restart the sequence numbers so that parity and flags agree
  for (size_t r = 0; r < g->ctx.size(); r++) { (void)hipSetDevice(g->dev[r]); (void)hipStreamSynchronize(g->ctx[r]->stream); }
  for (int w = 0; w < 5; w++) { g->issued[w] = 0; for (int r = 0; r < g->n; r++) g->recorded[r][w].store(0); }
  for (int r = 0; r < g->n; r++) { (void)hipSetDevice(g->dev[r]); if (g->flags[r]) (void)hipMemset(g->flags[r], 0, sizeof(unsigned long long) * 5 * 2 * tj::GROUP_MAX); (void)hipDeviceSynchronize(); }
  return g->n > 1 ? group_select_transport(g, t) : TJ_OK;
}

#define GROUP_EACH(g, call)                                                                                     \
  for (int r_ = 0; r_ < (g)->n; r_++) {                                                                         \
    tj_ctx* c = (g)->ctx[r_];                                                                                   \
    if (hipSetDevice((g)->dev[r_]) != hipSuccess) return group_fail(g, TJ_ERR_DEVICE, "hipSetDevice failed");   \
    const int rc_ = (call);                                                                                     \
    if (rc_ < 0) return group_fail(g, rc_, std::string("rank ") + std::to_string(r_) + ": " + tj_last_error(c)); \
  }
#define GROUP_LIVE(g) do { if (!(g)) return TJ_ERR_INVALID; if ((g)->poisoned) return TJ_ERR_DEVICE; } while (0)   /* the stored error says which rank failed and why */

int tj_group_set_cloud(tj_group* g, const double* points, int n) { GROUP_LIVE(g); GROUP_EACH(g, tj_set_cloud(c, points, n)); return TJ_OK; }
int tj_group_set_mesh(tj_group* g, const double* verts, int n_verts, const int* tris, int n_tris) { GROUP_LIVE(g); GROUP_EACH(g, tj_set_mesh(c, verts, n_verts, tris, n_tris)); return TJ_OK; }
// also the way out of a poisoned group: every stream is drained, the exchange counters, flag words and recorded counts restart
int tj_group_init_state(tj_group* g, const double* waypoints, double piece_time) {
  if (!g) return TJ_ERR_INVALID;
  for (size_t r = 0; r < g->ctx.size(); r++) { (void)hipSetDevice(g->dev[r]); (void)hipStreamSynchronize(g->ctx[r]->stream); }
  for (int w = 0; w < 5; w++) { g->issued[w] = 0; for (int r = 0; r < g->n; r++) g->recorded[r][w].store(0); }
  for (int r = 0; r < g->n; r++) { (void)hipSetDevice(g->dev[r]); if (g->flags[r]) (void)hipMemset(g->flags[r], 0, sizeof(unsigned long long) * 5 * 2 * tj::GROUP_MAX); (void)hipDeviceSynchronize(); }
  g->poisoned = false;
  GROUP_EACH(g, tj_init_state(c, waypoints, piece_time));   // (also restarts the direct exchange's arrival counters: every stream is drained here)
  return TJ_OK;
}

int tj_group_iterate(tj_group* g, int n_iters, double* gnorm, int* iters_total, int* converged) {
  if (!g || n_iters < 0) return TJ_ERR_INVALID;
  if (g->poisoned) return TJ_ERR_DEVICE;
  g->abort_flag.store(0);
  std::vector<int> rc(g->n, TJ_OK);
  const bool cpl_sharded = g->n > 1 && g->ctx[0]->d.mode == TJ_MODE_MULTI_COUPLED;
  if (cpl_sharded)   // the state the batch starts from: a search that leaves the exchanged candidates (error bit 32) is followed in a second, careful run of the batch (below)
    for (int r = 0; r < g->n; r++) {
      tj_ctx* c = g->ctx[r];
      if (hipSetDevice(g->dev[r]) != hipSuccess) return group_fail(g, TJ_ERR_DEVICE, "hipSetDevice failed");
      if (int fr = flush_deferred(c)) return group_fail(g, fr, tj_last_error(c));   // (the update the previous batch still owes belongs to the state)
      hipLaunchKernelGGL(k_snapshot, dim3(64, std::max(c->snap_n, 1)), dim3(256), 0, c->stream, c->snap_tab, c->snap_n, 0, c->d.ctl, c->ctl_snap);
    }
  auto run_batch = [&](bool careful, std::vector<long>* extra) {
    if (g->n == 1) rc[0] = group_rank_loop(g, 0, n_iters, careful, extra ? &(*extra)[0] : nullptr);
    else {
      std::vector<std::thread> th;
      for (int r = 0; r < g->n; r++) th.emplace_back([g, r, n_iters, &rc, careful, extra]() { rc[r] = group_rank_loop(g, r, n_iters, careful, extra ? &(*extra)[r] : nullptr); if (rc[r]) g->abort_flag.store(1); });
      for (auto& t : th) t.join();
    }
  };
  run_batch(false, nullptr);
  if (cpl_sharded && std::all_of(rc.begin(), rc.end(), [](int x) { return x == TJ_OK; })) {
    bool range = false;
    for (int r = 0; r < g->n; r++) {
      tj_ctx* c = g->ctx[r];
      int err = 0;
      if (hipSetDevice(g->dev[r]) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&err, &c->d.ctl->error, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return group_fail(g, TJ_ERR_DEVICE, "tj_group_iterate: reading a rank's error word failed");
      range = range || (err & ERR_LS_RANGE) != 0;
    }
    if (range) {
      // the Armijo search of some iteration needed more candidates than one exchange carries (never met outside constructed states): every rank back to the batch's
      // first state, then the same iterations with the search followed to the reference's own end -- one host look per iteration, slow and exact
      for (int w = 0; w < 5; w++) g->issued[w] += n_iters;   // (the abandoned run's exchanges were issued: the sequence numbers and the buffer parity go on from there)
      for (int r = 0; r < g->n; r++) {
        tj_ctx* c = g->ctx[r];
        (void)hipSetDevice(g->dev[r]);
        hipLaunchKernelGGL(k_snapshot, dim3(64, std::max(c->snap_n, 1)), dim3(256), 0, c->stream, c->snap_tab, c->snap_n, 1, c->d.ctl, c->ctl_snap);
        c->hull_valid = false; c->ccd_valid = false; c->maybe_deferred = false; c->begin_folded = false;
        tj_set_coupled_follow(c, 1);
      }
      std::vector<long> extra(g->n, 0);
      run_batch(true, &extra);
      for (int r = 0; r < g->n; r++) tj_set_coupled_follow(g->ctx[r], 0);
      g->issued[4] += extra[0];
      g->lsc_followed++;
    }
  }
  for (int r = 0; r < g->n; r++) if (rc[r]) {
    // the ranks have enqueued different numbers of phases and pushes: nothing but tj_group_init_state / destroy may follow
    g->poisoned = true;
    return group_fail(g, rc[r], std::string("rank ") + std::to_string(r) + ": " + tj_last_error(g->ctx[r]));
  }
  {  // what the schedule exchanged, for the parity of the next call
    const bool cpl = g->ctx[0]->d.mode == TJ_MODE_MULTI_COUPLED;
    for (int w = 0; w < (cpl ? 5 : 2); w++) g->issued[w] += n_iters;
  }
  Ctl h0; memset(&h0, 0, sizeof(h0));
  for (int r = 0; r < g->n; r++) {
    tj_ctx* c = g->ctx[r];
    if (hipSetDevice(g->dev[r]) != hipSuccess) return group_fail(g, TJ_ERR_DEVICE, "hipSetDevice failed");
    int e = flush_deferred(c);
    Ctl h;
    if (!e) e = check_device_errors(c, &h);
    if (e) { if (e == TJ_ERR_DEVICE) g->poisoned = true; return group_fail(g, e, std::string("rank ") + std::to_string(r) + ": " + tj_last_error(c)); }
    if (r == 0) h0 = h;   // gnorm, the iteration counter and the stop flag are formed identically on every rank
  }
  const int it = h0.iter + h0.pending;
  if (gnorm) *gnorm = h0.gnorm;
  if (iters_total) *iters_total = it;
  if (converged) *converged = h0.done || (g->ctx[0]->d.stop > 0 && it > 1 && h0.gnorm < g->ctx[0]->d.stop);
  return TJ_OK;
}

// Event-timed cost of ONE exchange of each buffer kind: every rank's host thread issues `reps` exchanges of kind w back to back
// (the slices are what they are: re-sending them changes nothing) between two events on its stream; us[w] = the slowest rank's
// average.  us has 5 entries (kinds 2..4 exist in coupled mode only; 0 otherwise).  One rank: all zeros (nothing is foreign).
int tj_group_profile_exchange(tj_group* g, int reps, double* us) {
  if (!g || !us || reps < 1) return TJ_ERR_INVALID;
  if (g->poisoned) return TJ_ERR_DEVICE;
  for (int w = 0; w < 5; w++) us[w] = 0;
  if (g->n == 1) return TJ_OK;
  const int nwhat = g->ctx[0]->d.mode == TJ_MODE_MULTI_COUPLED ? 5 : 2;
  for (int w = 0; w < nwhat; w++) {
    std::vector<int> rc(g->n, TJ_OK);
    std::vector<float> ms(g->n, 0.f);
    g->abort_flag.store(0);
    std::vector<std::thread> th;
    for (int r = 0; r < g->n; r++) th.emplace_back([g, r, w, reps, &rc, &ms]() {
      tj_ctx* c = g->ctx[r];
      if (hipSetDevice(g->dev[r]) != hipSuccess) { rc[r] = TJ_ERR_DEVICE; g->abort_flag.store(1); return; }
      hipEvent_t e0, e1;
      if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc[r] = TJ_ERR_DEVICE; g->abort_flag.store(1); return; }
      long s = g->issued[w];
      int e = group_exchange(g, r, w, s++);   // one untimed exchange first (module load, first-touch of the peer mapping)
      (void)hipEventRecord(e0, c->stream);
      for (int i = 0; i < reps && !e; i++) e = group_exchange(g, r, w, s++);
      (void)hipEventRecord(e1, c->stream);
      if (!e && hipStreamSynchronize(c->stream) != hipSuccess) e = TJ_ERR_DEVICE;
      if (!e) (void)hipEventElapsedTime(&ms[r], e0, e1);
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      rc[r] = e; if (e) g->abort_flag.store(1);
    });
    for (auto& t : th) t.join();
    for (int r = 0; r < g->n; r++) if (rc[r]) { g->poisoned = true; return group_fail(g, rc[r], std::string("rank ") + std::to_string(r) + ": " + tj_last_error(g->ctx[r])); }
    g->issued[w] += reps + 1;
    float worst = 0; for (int r = 0; r < g->n; r++) worst = std::max(worst, ms[r]);
    us[w] = 1e3 * worst / reps;
  }
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
