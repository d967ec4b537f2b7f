// multi_gpu.hip - ONE process driving several GPUs: what an unchanged mapper.cc reaches through bundle_adjustment()
// -> mavba_solve when MAVBA_GPUS=N is set (reference call site: src/sfm/sequential_mapper.cc:1074-1080, one call from
// one thread of one process).
//
// The 3-D points (with all their observations) are sharded over N ranks exactly as the one-process-per-GPU path does
// (SURVEY.md 8(e)): contiguous point ranges balanced by observation count, cameras replicated, rotation priors on rank 0.
// Every rank is a host thread with its own session on its own device; the exchange of the reduced camera system is an
// in-process all-reduce over xGMI peer access:
//
//   one-shot, owner-computes-slice: rank r reduces slice r of the vector by reading that slice from every rank's buffer
//   (direct peer loads, fixed rank order: every rank ends with bit-identical sums) and writes the result back into every
//   rank's buffer (direct peer stores). Every link of the full mesh carries 1/N of the vector in each direction - the
//   point-to-point topology of the MI355X node is used as what it is, no ring.
//
// Where peer access is not available the slices travel by hipMemcpyPeer through a staging buffer. Host threads meet at
// a barrier before and after the kernel (the sessions' hook protocol is host-synchronised anyway).
#include "session.h"

namespace mavba {
namespace {

constexpr int kMaxRanks = 16;

struct PeerTable { double* p[kMaxRanks]; };

// out = op over ranks (fixed order 0..world-1) of bufs[r][i] for i in [begin, end), written back to every rank.
// op 0 = sum, 1 = max, 2 = sum for all but the LAST element of the whole vector (index count - 1), max for that one.
__global__ void __launch_bounds__(256) k_inproc_allreduce(PeerTable T, int world, long long begin, long long end, long long count,
                                                          int op) {
  for (long long i = begin + (long long)blockIdx.x * 256 + threadIdx.x; i < end; i += (long long)gridDim.x * 256) {
    const bool mx = op == 1 || (op == 2 && i == count - 1);
    double acc = T.p[0][i];
    for (int r = 1; r < world; ++r) {
      const double x = T.p[r][i];
      acc = mx ? fmax(acc, x) : acc + x;
    }
    for (int r = 0; r < world; ++r) T.p[r][i] = acc;
  }
}

struct InProcGroup {
  int world = 1;
  std::vector<int> device;            // device of every rank
  bool direct = true;                 // every pair of devices can map each other's memory
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long long generation = 0;
  bool failed = false;
  std::vector<double*> ptr;           // this exchange: every rank's buffer
  std::vector<long long> count;
  std::vector<hipStream_t> stream;    // one per rank, on its device
  std::vector<double*> stage;         // memcpy path: staging block on every rank's device (world slices of the largest exchange)
  std::vector<size_t> stage_doubles;

  // false: another rank failed - give up (nobody may wait for a rank that is gone)
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (failed) return false;
    const unsigned long long g = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); return true; }
    cv.wait(lk, [&] { return generation != g || failed; });
    return !failed;
  }
  void fail() {
    std::lock_guard<std::mutex> lk(m);
    failed = true;
    cv.notify_all();
  }
};

struct RankCtx { InProcGroup* g; int rank; };

int inproc_allreduce(void* vctx, void* device_ptr, int64_t count, int32_t op) {
  RankCtx* c = static_cast<RankCtx*>(vctx);
  InProcGroup& G = *c->g;
  const int r = c->rank, W = G.world;
  G.ptr[r] = static_cast<double*>(device_ptr);
  G.count[r] = count;
  if (!G.barrier()) return 1;  // every rank's buffer is complete (its stream is idle) and published
  for (int q = 0; q < W; ++q) if (G.count[q] != count) { G.fail(); return 1; }
  const long long b = count * r / W, e = count * (r + 1) / W;
  hipStream_t st = G.stream[r];
  bool ok = true;
  if (G.direct) {
    if (e > b) {
      PeerTable T;
      for (int q = 0; q < W; ++q) T.p[q] = G.ptr[q];
      const int grid = (int)std::min<long long>(1024, (e - b + 255) / 256);
      hipLaunchKernelGGL(k_inproc_allreduce, dim3(grid), dim3(256), 0, st, T, W, b, e, (long long)count, (int)op);
    }
    ok = hipStreamSynchronize(st) == hipSuccess && hipGetLastError() == hipSuccess;
  } else {
    // staged: gather the peers' copies of my slice, reduce locally, scatter the result
    const size_t n = (size_t)(e - b);
    if (n > 0) {
      if (G.stage_doubles[r] < n * W) {
        if (G.stage[r]) device_free(G.stage[r]);
        G.stage[r] = nullptr;
        ok = device_alloc(reinterpret_cast<void**>(&G.stage[r]), n * W * sizeof(double)) == hipSuccess;
        G.stage_doubles[r] = ok ? n * W : 0;
      }
      PeerTable T;
      for (int q = 0; q < W && ok; ++q) {
        T.p[q] = G.stage[r] + (size_t)q * n - b;  // (indexed with the global element number)
        ok = hipMemcpyPeerAsync(G.stage[r] + (size_t)q * n, G.device[r], G.ptr[q] + b, G.device[q], n * sizeof(double), st) == hipSuccess;
      }
      if (ok) {
        const int grid = (int)std::min<long long>(1024, (e - b + 255) / 256);
        hipLaunchKernelGGL(k_inproc_allreduce, dim3(grid), dim3(256), 0, st, T, W, b, e, (long long)count, (int)op);
        for (int q = 0; q < W && ok; ++q)
          ok = hipMemcpyPeerAsync(G.ptr[q] + b, G.device[q], G.stage[r], G.device[r], n * sizeof(double), st) == hipSuccess;
      }
    }
    ok = hipStreamSynchronize(st) == hipSuccess && ok;
  }
  if (!ok) { G.fail(); return 1; }
  if (!G.barrier()) return 1;  // every slice is written everywhere
  return 0;
}

// contiguous point ranges balanced by observation count (the rule of BAProblem.shard_by_point / bench.py)
std::vector<int> shard_bounds(const mavba_problem* P, int world) {
  // (the same answers as the single-GPU path gives for a malformed problem - before anything is indexed)
  if (P->num_points < 0 || P->num_obs < 0 || (P->num_obs > 0 && (!P->obs_point || !P->obs_image || !P->obs_uv)) || (P->num_points > 0 && !P->points))
    throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null array in the problem");
  for (long long o = 0; o < P->num_obs; ++o)
    if (P->obs_point[o] < 0 || P->obs_point[o] >= P->num_points) throw Failure(MAVBA_ERR_BAD_INDEX, "observation index out of range");
  std::vector<long long> csum((size_t)P->num_points + 1, 0);
  for (long long o = 0; o < P->num_obs; ++o) csum[P->obs_point[o] + 1]++;
  for (int p = 0; p < P->num_points; ++p) csum[p + 1] += csum[p];
  const long long total = csum[P->num_points];
  std::vector<int> bounds(world + 1, P->num_points);
  bounds[0] = 0;
  for (int r = 1; r < world; ++r) {
    const double target = (double)total * r / world;
    // first p with csum(obs of points 0..p) >= target
    int lo = 0, hi = P->num_points;
    while (lo < hi) { const int mid = (lo + hi) / 2; if ((double)csum[mid + 1] >= target) hi = mid; else lo = mid + 1; }
    bounds[r] = std::max(bounds[r - 1], std::min(lo, P->num_points));
  }
  return bounds;
}

}  // namespace

// ---- the RCCL communicators of the in-process ranks (round 5) -------------------------------------------------------
// Round 4 created them inside every mavba_solve (one ncclCommInitRank per rank thread per call): tens of milliseconds in
// front of a 15 ms solve. Now they belong to the process: created once per (world, devices) by one thread per rank - RCCL
// binds a communicator to the device current at creation -, handed to the sessions as BORROWED handles, aborted as a group
// when a rank fails (so that nobody is left inside a collective a dead rank will never join) and then rebuilt on demand.
namespace {
struct CommGroup {
  std::mutex m;
  int world = 0;
  std::vector<int> device;
  std::vector<void*> comm;
  std::shared_ptr<RcclGuard> guard;  // of the CURRENT communicators; a session keeps its group's guard alive
};
CommGroup& comm_group() { static CommGroup* g = new CommGroup; return *g; }
// One multi-GPU solve of the process at a time on the RCCL path: the group is a process-wide singleton - two
// concurrent solves would interleave their collectives on the same communicators in rank-inconsistent order, a solve with
// other devices would destroy the group under a running one, one solve's abort would free the other's handles.
std::mutex& rccl_solve_mutex() { static std::mutex* m = new std::mutex; return *m; }
}  // namespace

bool inproc_comms_acquire(int world, const std::vector<int>& device, std::vector<void*>& out, std::shared_ptr<RcclGuard>& guard,
                          std::string& error) {
  CommGroup& C = comm_group();
  std::lock_guard<std::mutex> lk(C.m);
  if (C.world == world && C.device == device && (int)C.comm.size() == world) {
    // a cached group is reused only while every communicator is healthy (one in an asynchronous error state would fail or hang
    // every later collective)
    bool healthy = true;
    for (void* c : C.comm) healthy = healthy && rccl_comm_async_error(c) == 0;
    if (healthy) { out = C.comm; guard = C.guard; return true; }
    for (void* c : C.comm) rccl_comm_abort(c);
    C.comm.clear(); C.world = 0; C.device.clear(); C.guard.reset();
  }
  for (void* c : C.comm) rccl_comm_destroy(c);
  C.comm.clear(); C.world = 0; C.device.clear(); C.guard.reset();
  unsigned char uid[128];
  try { rccl_unique_id(uid); } catch (const std::exception& e) { error = e.what(); return false; }
  // everything that can fail without the peers goes first: a creator that threw before ncclCommInitRank would leave the other
  // ranks waiting in the rendezvous for good (and this thread in join() with the group's mutex held)
  {
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int r = 0; r < world; ++r)
      if (device[r] >= 0 && hipSetDevice(device[r]) != hipSuccess) {
        (void)hipGetLastError();
        if (prev >= 0) (void)hipSetDevice(prev);
        error = "rank " + std::to_string(r) + ": device " + std::to_string(device[r]) + " cannot be selected";
        return false;
      }
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  std::vector<void*> comm(world, nullptr);
  std::vector<std::string> err(world);
  auto make = [&](int r) {
    int prev = -1;
    try {
      if (device[r] >= 0) { (void)hipGetDevice(&prev); HIP_OK(hipSetDevice(device[r])); }
      comm[r] = rccl_comm_create(uid, r, world);
    } catch (const std::exception& e) { err[r] = e.what(); }
    if (prev >= 0) (void)hipSetDevice(prev);
  };
  std::vector<std::thread> th;
  for (int r = 1; r < world; ++r) th.emplace_back(make, r);
  make(0);
  for (auto& t : th) t.join();
  for (int r = 0; r < world; ++r)
    if (!comm[r]) {
      error = "rank " + std::to_string(r) + ": " + (err[r].empty() ? std::string("ncclCommInitRank failed") : err[r]);
      for (void* c : comm) rccl_comm_abort(c);
      return false;
    }
  C.world = world; C.device = device; C.comm = comm; C.guard = std::make_shared<RcclGuard>();
  out = comm; guard = C.guard;
  return true;
}

// Called by a rank that failed: first nobody may START another collective on the handles (the guard, exclusively - waits
// for the peers that are inside an enqueue call), then the communicators are aborted (what is in flight on the devices is
// given up, the peers' read-backs end) and freed. Peers find `aborted` set before they touch a handle again.
void inproc_comms_abort() {
  CommGroup& C = comm_group();
  std::lock_guard<std::mutex> lk(C.m);
  if (C.guard) {
    std::unique_lock<std::shared_mutex> g(C.guard->m);
    C.guard->aborted = true;
    for (void* c : C.comm) rccl_comm_abort(c);
  } else {
    for (void* c : C.comm) rccl_comm_abort(c);
  }
  C.comm.clear(); C.world = 0; C.device.clear(); C.guard.reset();
}

int multi_gpu_ranks() {
  const char* e = std::getenv("MAVBA_GPUS");
  if (!e) return 1;
  int n = std::atoi(e);
  if (n <= 1) return 1;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 1; }
  if (std::getenv("MAVBA_GPUS_SAME_DEVICE")) return std::min(n, kMaxRanks);  // tests: every rank on the current device
  return std::max(1, std::min(std::min(n, ndev), kMaxRanks));
}

// mavba_solve over `world` devices of this process. Same contract: parameters written back in place (unless the solve
// ends in NUMERICAL_FAILURE), `result` summarises the (global) solve, `point_error` as mavba_solve.
int solve_multi_gpu(const mavba_problem* P, const mavba_options* options, mavba_result* result, double* point_error, int world) {
  const bool same_device = std::getenv("MAVBA_GPUS_SAME_DEVICE") != nullptr;
  int cur = 0;
  HIP_OK(hipGetDevice(&cur));
  InProcGroup G;
  G.world = world;
  G.device.resize(world);
  for (int r = 0; r < world; ++r) G.device[r] = same_device ? cur : r;
  G.ptr.assign(world, nullptr); G.count.assign(world, 0); G.stream.assign(world, nullptr);
  G.stage.assign(world, nullptr); G.stage_doubles.assign(world, 0);
  G.direct = std::getenv("MAVBA_GPUS_STAGED") == nullptr;
  const std::vector<int> bounds = shard_bounds(P, world);  // (validates the indices: nothing is created before that)
  // the per-rank streams and staging buffers leave with this scope, whatever throws below
  struct Cleanup {
    InProcGroup& G; int cur;
    ~Cleanup() {
      for (size_t r = 0; r < G.stream.size(); ++r) {
        if (!G.stream[r] && !G.stage[r]) continue;
        (void)hipSetDevice(G.device[r]);
        if (G.stream[r]) { (void)hipStreamSynchronize(G.stream[r]); stream_release(G.stream[r], G.device[r]); }  // (back to the process's stream cache)
        if (G.stage[r]) device_free(G.stage[r]);
        G.stream[r] = nullptr; G.stage[r] = nullptr;
      }
      (void)hipSetDevice(cur);
    }
  } cleanup{G, cur};
  for (int a = 0; a < world && G.direct; ++a)
    for (int b = 0; b < world && G.direct; ++b) {
      if (G.device[a] == G.device[b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, G.device[a], G.device[b]) != hipSuccess || !can) { (void)hipGetLastError(); G.direct = false; }
    }
  for (int a = 0; a < world; ++a) {
    HIP_OK(hipSetDevice(G.device[a]));
    HIP_OK(stream_acquire(&G.stream[a]));  // (the exchange kernel's stream, from the cache the sessions use: created once per device and process)
    if (G.direct)
      for (int b = 0; b < world; ++b)
        if (G.device[a] != G.device[b]) {
          const hipError_t e = hipDeviceEnablePeerAccess(G.device[b], 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) G.direct = false;
          (void)hipGetLastError();
        }
  }
  HIP_OK(hipSetDevice(cur));

  struct Shard {
    std::vector<double> points, uv, perr;
    std::vector<uint8_t> pconst;
    std::vector<int32_t> oimg, opt;
    std::vector<double> poses, intr;
    mavba_problem prob;
    mavba_result res;
    int rc = MAVBA_OK, term = MAVBA_TERM_RUNNING;
    std::string error;
  };
  std::vector<Shard> sh(world);
  // observations to their rank, input order kept: count, size once, fill (no per-element growth)
  std::vector<int> rank_of_point((size_t)std::max(P->num_points, 1), 0);
  for (int r = 0; r < world; ++r) for (int p = bounds[r]; p < bounds[r + 1]; ++p) rank_of_point[p] = r;
  {
    std::vector<long long> cnt(world, 0);
    for (long long o = 0; o < P->num_obs; ++o) cnt[rank_of_point[P->obs_point[o]]]++;  // (indices validated by shard_bounds)
    for (int r = 0; r < world; ++r) { sh[r].uv.resize(2 * (size_t)cnt[r]); sh[r].oimg.resize((size_t)cnt[r]); sh[r].opt.resize((size_t)cnt[r]); cnt[r] = 0; }
    for (long long o = 0; o < P->num_obs; ++o) {
      const int p = P->obs_point[o], r = rank_of_point[p];
      Shard& S = sh[r];
      const size_t at = (size_t)cnt[r]++;
      S.uv[2 * at] = P->obs_uv[2 * o]; S.uv[2 * at + 1] = P->obs_uv[2 * o + 1];
      S.oimg[at] = P->obs_image[o]; S.opt[at] = p - bounds[r];
    }
  }
  for (int r = 0; r < world; ++r) {
    Shard& S = sh[r];
    const int lo = bounds[r], hi = bounds[r + 1];
    S.points.assign(P->points + 3 * (size_t)lo, P->points + 3 * (size_t)hi);
    if (P->point_const) S.pconst.assign(P->point_const + lo, P->point_const + hi);
    S.poses.assign(P->poses, P->poses + 6 * (size_t)P->num_images);
    S.intr.assign(P->intrinsics, P->intrinsics + 9 * (size_t)P->num_cameras);
    S.perr.assign((size_t)std::max(hi - lo, 1), std::numeric_limits<double>::quiet_NaN());
    S.prob = *P;
    S.prob.num_points = hi - lo; S.prob.num_obs = (int64_t)S.oimg.size();
    S.prob.points = S.points.data(); S.prob.point_const = P->point_const ? S.pconst.data() : nullptr;
    S.prob.obs_uv = S.uv.data(); S.prob.obs_image = S.oimg.data(); S.prob.obs_point = S.opt.data();
    S.prob.poses = S.poses.data(); S.prob.intrinsics = S.intr.data();
    if (r != 0) { S.prob.num_rot_priors = 0; S.prob.rot_prior_image = nullptr; S.prob.rot_prior_rvec = nullptr; }  // camera-only residuals: counted once
    std::memset(&S.res, 0, sizeof(S.res));
  }

  // The exchange. Ranks on DISTINCT devices: one RCCL communicator per rank inside this process (ncclCommInitRank from the
  // rank's own thread, one unique id) - the all-reduce of the reduced camera system is enqueued on the session's stream
  // like in the process-per-GPU launch, no host barrier, no stream synchronisation per collective. Ranks that share a
  // device (MAVBA_GPUS_SAME_DEVICE: tests on one GPU - RCCL refuses duplicates), no librccl, or MAVBA_GPUS_EXCHANGE=inproc:
  // the peer-access kernel behind host barriers.
  bool use_rccl = !same_device;
  if (const char* e = std::getenv("MAVBA_GPUS_EXCHANGE")) use_rccl = use_rccl && std::string(e) != "inproc";
  std::vector<void*> comms;
  std::shared_ptr<RcclGuard> guard;
  std::unique_lock<std::mutex> one_rccl_solve;  // held from here to the end of the call on the RCCL path
  if (use_rccl) {
    one_rccl_solve = std::unique_lock<std::mutex>(rccl_solve_mutex());
    std::string why;
    if (!inproc_comms_acquire(world, G.device, comms, guard, why)) { use_rccl = false; one_rccl_solve.unlock(); }  // (librccl.so not loadable, ...: the peer-access exchange)
  }
  std::vector<RankCtx> ctx(world);
  auto worker = [&](int r) {
    Shard& S = sh[r];
    mavba_session* s = nullptr;
    mavba_options o = *options;
    o.device = G.device[r];
    ctx[r] = RankCtx{&G, r};
    S.rc = mavba_session_create(&S.prob, &o, &s);
    if (use_rccl) {
      // every rank must reach ncclCommInitRank or none: a rank whose set-up failed would leave the others waiting in it
      if (S.rc != MAVBA_OK) { S.error = g_last_error; G.fail(); }
      if (!G.barrier() && S.rc == MAVBA_OK) { S.rc = MAVBA_ERR_HIP; g_last_error = "another rank failed during set-up"; }
    }
    if (S.rc == MAVBA_OK) S.rc = use_rccl ? session_borrow_rccl(s, comms[r], guard, r, world) : mavba_session_set_allreduce(s, inproc_allreduce, &ctx[r], r, world);
    int done = 0;
    if (S.rc == MAVBA_OK) S.rc = mavba_session_iterate(s, options->max_num_iterations + 1, &done, &S.term);
    // (stream-ordered collectives: a peer that failed aborts the group, this rank's collectives then complete with
    // whatever was in the buffers - its numbers mean nothing)
    if (S.rc == MAVBA_OK && use_rccl && G.failed) { S.rc = MAVBA_ERR_HIP; g_last_error = "another rank failed during the solve"; }
    if (S.rc == MAVBA_OK) S.rc = mavba_session_result(s, &S.res);
    // ceres leaves the user's parameter blocks untouched after NUMERICAL_FAILURE
    if (S.rc == MAVBA_OK && S.term != MAVBA_TERM_NUMERICAL_FAILURE)
      S.rc = mavba_session_get_params(s, S.poses.data(), S.intr.data(), S.points.data());
    if (S.rc == MAVBA_OK && point_error && options->update_point_errors) {
      // problem.Evaluate runs on the user's blocks (bundle_adjustment.cc:583-588): after NUMERICAL_FAILURE those still hold
      // the parameters the call started from - as mavba_solve does on one GPU
      if (S.term == MAVBA_TERM_NUMERICAL_FAILURE) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        try { HIP_OK(hipSetDevice(s->device)); s->restore_initial_params(); } catch (const std::exception& e) { g_last_error = e.what(); S.rc = MAVBA_ERR_HIP; }
        if (prev >= 0) (void)hipSetDevice(prev);
      }
      if (S.rc == MAVBA_OK) S.rc = mavba_session_point_errors(s, S.perr.data());
    }
    if (S.rc != MAVBA_OK) {
      S.error = g_last_error; G.fail();
      // nobody may be left inside a collective this rank will never join: abort the group's communicators (the peers'
      // ncclAllReduce calls return or complete, their read-backs end, the threads join); the next call builds a new group
      if (use_rccl) inproc_comms_abort();
    }
    if (s) mavba_session_destroy(s);
  };
  std::vector<std::thread> threads;
  for (int r = 1; r < world; ++r) threads.emplace_back(worker, r);
  worker(0);
  for (auto& t : threads) t.join();
  for (int r = 0; r < world; ++r)
    if (sh[r].rc != MAVBA_OK && !sh[r].error.empty()) throw Failure(sh[r].rc, "rank " + std::to_string(r) + ": " + sh[r].error);
  for (int r = 0; r < world; ++r)
    if (sh[r].rc != MAVBA_OK) throw Failure(sh[r].rc, "rank " + std::to_string(r) + " failed");
  // every rank holds the same cameras and the same (global) summary
  if (result) *result = sh[0].res;
  if (sh[0].term != MAVBA_TERM_NUMERICAL_FAILURE) {
    std::memcpy(P->poses, sh[0].poses.data(), sizeof(double) * 6 * (size_t)P->num_images);
    std::memcpy(P->intrinsics, sh[0].intr.data(), sizeof(double) * 9 * (size_t)P->num_cameras);
    for (int r = 0; r < world; ++r)
      std::memcpy(P->points + 3 * (size_t)bounds[r], sh[r].points.data(), sizeof(double) * 3 * (size_t)(bounds[r + 1] - bounds[r]));
  }
  if (point_error && options->update_point_errors)
    for (int r = 0; r < world; ++r)
      for (int p = bounds[r]; p < bounds[r + 1]; ++p) {
        const double e = sh[r].perr[p - bounds[r]];
        if (!std::isnan(e)) point_error[p] = e;  // (points without observations are left untouched, as mavba_solve leaves them)
      }
  return MAVBA_OK;
}

}  // namespace mavba
