// session.h - shared by the session_*.hip / api.hip translation units: host-side helpers and the session object.
// (Split of the former session.hip: set-up in session_build.hip, the LM loop in session_lm.hip, the C ABI in api.hip,
// the process-wide device pool / stream cache / host worker threads in host_util.hip.)
#ifndef MAVBA_SESSION_H_
#define MAVBA_SESSION_H_
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mavba.h"
#include "ba_math.h"
#include "internal.h"

namespace mavba {

extern thread_local std::string g_last_error;  // api.hip

struct Failure : std::runtime_error {
  int code;
  Failure(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#define HIP_OK(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      throw Failure(e_ == hipErrorOutOfMemory ? MAVBA_ERR_OUT_OF_MEMORY : MAVBA_ERR_HIP, \
                    std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) device_free(p); }
  void alloc(size_t count) {
    if (p) { device_free(p); p = nullptr; }
    n = count;
    if (count) HIP_OK(device_alloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
  }
  // from PAGEABLE host memory (std::vector, caller arrays): staged, see copy_h2d_staged
  void upload(const std::vector<T>& h, hipStream_t st) {
    alloc(std::max<size_t>(h.size(), 1));
    if (!h.empty()) HIP_OK(copy_h2d_staged(p, h.data(), h.size() * sizeof(T), st));
  }
  void upload(const T* h, size_t count, hipStream_t st) {
    alloc(std::max<size_t>(count, 1));
    if (count) HIP_OK(copy_h2d_staged(p, h, count * sizeof(T), st));
  }
  // from a page-locked block (PinnedBuf): the plain asynchronous copy
  void upload_pinned(const T* h, size_t count, hipStream_t st) {
    alloc(std::max<size_t>(count, 1));
    if (count) HIP_OK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, st));
  }
  void zero(hipStream_t st) { if (n) HIP_OK(zero_async(p, n * sizeof(T), st)); }
};

// session_build.hip: elimination tree of the reduced camera system (host only)
struct ElimNode { std::vector<int> imgs; int parent; };
std::vector<ElimNode> elimination_tree(int NI, int NC, const std::vector<std::vector<int>>& lower, int forced, int max_depth);

// device_setup.hip: a problem whose observations and points ALREADY live in HBM (the device-resident scene hands these over)
struct DeviceRaw { const double* uv; const int* img; const int* pt; const double* points; };
long long device_scan_scratch(long long n);
void device_scan_exclusive(hipStream_t st, unsigned* data, long long n, unsigned* scratch);  // in place, any length; scratch >= device_scan_scratch(n) values

// multi_gpu.hip: one process, several devices (MAVBA_GPUS)
struct RcclGuard;
int session_borrow_rccl(mavba_session* s, void* comm, const std::shared_ptr<RcclGuard>& guard, int rank, int world_size);  // (api.hip)
// the RCCL communicators of the in-process ranks, kept for the life of the process: ncclCommInitRank costs tens of
// milliseconds - more than the C3 solve it would serve. device[r] < 0: do not bind the creating thread (host-only tests)
bool inproc_comms_acquire(int world, const std::vector<int>& device, std::vector<void*>& out, std::shared_ptr<RcclGuard>& guard,
                          std::string& error);
void inproc_comms_abort();  // a rank failed: ncclCommAbort on every communicator (peers blocked in a collective return), group dropped
int multi_gpu_ranks();
int solve_multi_gpu(const mavba_problem* P, const mavba_options* options, mavba_result* result, double* point_error, int world);

// pose_refine.hip
void pose_refine_batch(int count, mavba_pose_refine_item* items, const mavba_options& opt, mavba_result* results);

// host_util.hip
void rccl_unique_id(void* out128);
void* rccl_comm_create(const void* id128, int rank, int world);
void rccl_comm_destroy(void* comm);
void rccl_comm_abort(void* comm);
void rccl_allreduce(void* comm, double* p, long long count, int op, hipStream_t stream);
int rccl_comm_async_error(void* comm);  // 0: healthy (or the library has no ncclCommGetAsyncError)
// The communicators of the in-process ranks are BORROWED by their sessions (multi_gpu.hip). A rank that fails aborts the whole
// group - ncclCommAbort frees the handles - while its peers may still be enqueueing collectives on them: every use of a
// borrowed handle holds this guard shared and looks at `aborted` first; the abort takes it exclusively.
struct RcclGuard {
  std::shared_mutex m;
  bool aborted = false;  // (written under the exclusive lock, read under the shared one)
};
hipError_t pinned_alloc(void** out, size_t bytes);
void pinned_free(void* p);
hipError_t stream_acquire(hipStream_t* st);
void stream_release(hipStream_t st, int dev);
double* lm_pub_alloc();          // host_util.hip: a 256-byte host-mapped coherent slot (null: none left / not supported)
void lm_pub_free(double* p);
bool persistent_allowed_now();   // host_util.hip: false while the process is in the cool-down after a time-out (counts one session of it)
bool persistent_in_cooldown();   // the same question without counting a session (sharded sessions: the ranks agree in join_ranks)
void persistent_timed_out();
int host_threads();
void host_run(int T, const std::function<void(int)>& body);

// The three big per-observation / per-point host arrays of a session are handed from a dying session to the next one
// (capacity reuse): freeing and re-faulting ~30 MB per bundle_adjustment() call cost 14 ms of tear-down at C3.
template <typename T>
struct HostSpare {
  static std::mutex& mtx() { static std::mutex m; return m; }
  static std::vector<std::vector<T>>& slots() { static std::vector<std::vector<T>>* s = new std::vector<std::vector<T>>; return *s; }
  static void give(std::vector<T>& v) {
    if (v.capacity() < (1u << 16)) return;
    std::lock_guard<std::mutex> g(mtx());
    if (slots().size() < 4) { slots().emplace_back(); slots().back().swap(v); }
  }
  static void take(std::vector<T>& v, size_t want) {  // v gets the spare with the largest capacity (if any)
    std::lock_guard<std::mutex> g(mtx());
    int best = -1;
    for (size_t i = 0; i < slots().size(); ++i)
      if (best < 0 || slots()[i].capacity() > slots()[best].capacity()) best = (int)i;
    if (best >= 0 && slots()[best].capacity() >= want / 2) { v.swap(slots()[best]); slots().erase(slots().begin() + best); }
  }
};

// Host scratch array WITHOUT value-initialisation (std::vector<T>(n) clears the memory first: ~1 ms per 10 MB,
// and the set-up shuffles ~100 MB of such arrays that are fully overwritten anyway).
void* host_scratch_alloc(size_t bytes);        // host_util.hip: process-wide cache of big host blocks
void host_scratch_free(void* p, size_t bytes);
template <typename T>
struct HostBuf {
  // Blocks come from (and go back to) a process-wide cache: a fresh 16-32 MB allocation is an mmap whose pages are
  // faulted in one by one while 16 threads write it, and an munmap when it dies - at C3 that was a third of the set-up.
  T* p = nullptr;
  size_t n = 0;
  explicit HostBuf(size_t count) : p(static_cast<T*>(host_scratch_alloc(std::max<size_t>(count, 1) * sizeof(T)))), n(count) {}
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  ~HostBuf() { host_scratch_free(p, std::max<size_t>(n, 1) * sizeof(T)); }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
};

// The same for arrays that are uploaded: page-locked blocks (from the pinned pool), so hipMemcpyAsync really is
// asynchronous and the transfer runs while the host builds the block structure.
template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  size_t n = 0;
  bool pinned = true;  // false: page-locking failed (locked-memory limit of the process) - an ordinary scratch block, the copy is then staged by the runtime
  explicit PinnedBuf(size_t count) : n(count) {
    void* q = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (pinned_alloc(&q, bytes) != hipSuccess) {
      (void)hipGetLastError();
      pinned = false;
      q = host_scratch_alloc(bytes);
    }
    p = static_cast<T*>(q);
  }
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { if (pinned) pinned_free(p); else host_scratch_free(p, std::max<size_t>(n, 1) * sizeof(T)); }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* data() { return p; }
};

// Stable counting sort of items 0..n-1 by key(i) in [0, nkeys) on a few threads (per-thread histograms turned
// into per-thread cursors): start[k] = first position of key k, emit(i, position) is called once per item.
// The result does not depend on the number of threads.
template <typename KeyFn, typename EmitFn>
static void counting_sort_parallel(long long n, int nkeys, KeyFn key, std::vector<int>& start, EmitFn emit) {
  int T = n >= 200000 ? host_threads() : 1;
  while (T > 1 && (size_t)T * nkeys > ((size_t)64 << 20)) T /= 2;
  HostBuf<int> hist_store((size_t)T * nkeys);  // (cached block: no page faults after the first session)
  std::vector<int*> hist(T);
  for (int t = 0; t < T; ++t) hist[t] = hist_store.data() + (size_t)t * nkeys;
  auto run = [&](const std::function<void(int)>& body) { host_run(T, body); };
  run([&](int t) {
    std::memset(hist[t], 0, (size_t)nkeys * sizeof(int));
    for (long long i = n * t / T; i < n * (t + 1) / T; ++i) hist[t][key(i)]++;
  });
  // per-thread counts -> per-thread cursors (key-major, thread-minor), over key ranges in parallel: the serial version
  // of this loop was 3 ms of the C3 set-up (200 000 keys x 16 threads)
  start.assign((size_t)nkeys + 1, 0);
  std::vector<long long> range_sum((size_t)T + 1, 0);
  run([&](int r) {
    long long sum = 0;
    for (long long k = (long long)nkeys * r / T; k < (long long)nkeys * (r + 1) / T; ++k) {
      int within = 0;
      for (int t = 0; t < T; ++t) { const int c = hist[t][k]; hist[t][k] = within; within += c; }
      start[k] = within;  // (the key's total, turned into its first position below)
      sum += within;
    }
    range_sum[r + 1] = sum;
  });
  for (int r = 0; r < T; ++r) range_sum[r + 1] += range_sum[r];
  run([&](int r) {
    int pos = (int)range_sum[r];
    for (long long k = (long long)nkeys * r / T; k < (long long)nkeys * (r + 1) / T; ++k) {
      const int tot = start[k];
      start[k] = pos;
      for (int t = 0; t < T; ++t) hist[t][k] += pos;
      pos += tot;
    }
  });
  start[nkeys] = (int)range_sum[T];
  run([&](int t) {
    for (long long i = n * t / T; i < n * (t + 1) / T; ++i) emit(i, hist[t][key(i)]++);
  });
}

static inline int model_k(int m) { return m == MAVBA_MODEL_PINHOLE ? 4 : m == MAVBA_MODEL_OPENCV ? 8 : 9; }
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct KernelTimer { std::string name; long long launches = 0; double total_ms = 0.0; };

// Run body(begin, end) over [0, n) on a few host threads (set-up work only).
template <typename F>
static void parallel_ranges(long long n, F&& body, long long min_parallel = 200000) {
  int T = (int)std::min<long long>(host_threads(), std::max<long long>(n, 1));
  if (n < min_parallel) T = 1;
  if (T == 1) { body(0ll, n); return; }
  host_run(T, [&](int t) { body(n * t / T, n * (t + 1) / T); });
}

}  // namespace mavba

// (internal header: mavba_session is the global C-ABI handle type, assembled from mavba:: pieces)
using namespace mavba;

struct mavba_session {
  mavba_options opt;
  int device = 0;
  hipStream_t st = nullptr;
  // sizes
  int NI = 0, NC = 0, NP = 0, N = 0, Nstride = 32, NPs = 32, KMAX = 4, n_full = 0, n_pad = 64, Q = 0;
  long long NO_all = 0;
  bool any_intr_free = false;
  // host-side copies
  std::vector<double> h_poses0, h_intr0, h_points0;
  std::vector<int> h_cam_model, h_img_cam, h_pt_start, h_oimg;
  std::vector<long long> perm;      // point-major position -> caller observation index
  std::vector<int> h_pt_count_all;  // observations per point in the caller's problem
  std::vector<double> h_dropped_rnorm;  // caller's point index -> sum |r_raw| of its all-constant (dropped) observations; empty if none
  std::vector<double> h_dropped_cost;   // caller's point index -> their part of the fixed cost; empty if none
  double fixed_cost_priors = 0.0;       // the constant rotation priors' part of the fixed cost
  int num_priors_all = 0;               // rotation priors in the caller's problem
  // Points filtered out of the resident problem (mavba_session_filter_points): internal order, empty = none.
  // Their observations stay in the arrays with zero weight, their blocks are not free.
  std::vector<unsigned char> h_pt_removed;
  DevBuf<unsigned char> d_pt_active;
  // Points are renumbered at session creation (sorted by their image lists, so that neighbours in the order
  // see the same images: the Schur-complement clusters rely on it). h_pt_orig[internal] = caller's index.
  std::vector<int> h_pt_orig;
  std::vector<unsigned char> h_pose_const, h_intr_const_in, h_pt_const_in;
  std::vector<unsigned char> h_img_used, h_cam_used, h_pt_used;
  std::vector<unsigned char> h_prior_on_img;  // image carries a (non-constant) rotation prior
  std::vector<unsigned char> h_pose_free, h_intr_free, h_pt_free;
  double fixed_cost = 0.0;
  long long num_residuals = 0, num_residuals_reduced = 0, num_parameters_reduced = 0;
  int num_priors = 0;
  double prior_weight = 0.0;
  double setup_seconds = 0.0, solve_seconds = 0.0;

  // ---- device: static problem data ----
  DevBuf<double2> d_uv, d_im_uv;
  DevBuf<int> d_obs_img, d_obs_pt, d_pt_start, d_im_pt, d_img_cam, d_cam_model, d_img_chunk_start,
      d_cam_img_start, d_cam_imgs, d_prior_img, d_prior_start, d_q_pt, d_q_cam, d_q_start, d_pt_count;
  // J-free front end (k_point_front): tiles of consecutive points; front_ok false = a point has more intrinsics entries
  // than a tile holds (or MAVBA_FRONT_PLANES is set): the Jacobian-plane kernels are used instead
  DevBuf<FrontTile> d_front_tiles;
  int num_front_tiles = 0;
  DevBuf<FrontTile> d_tail_tiles;  // tiles over the points [tail_begin, NP): what the fused kernel does not cover
  int num_tail_tiles = 0, tail_begin = 0;
  bool front_ok = false;
  bool fused_ok = false;        // every observed point before tail_begin is clustered: their front end runs inside the cluster kernel (k_schur_rows), the tail's in k_point_front
  int eval_rows = 0;            // cost partials the last evaluation pass wrote
  bool front_valid = false;     // Cu, gu, Gi, h and the entry records match the current x, scales and front_radius
  double front_radius = 0.0;
  bool fail_slot_clean = false; // SC_FAIL was cleared (with SC_FAIL_FRONT, by the front end) and no solve has run since
  bool planes_ready = false;    // the Jacobian planes exist (probe path / plane kernels only)
  DevBuf<SweepChunk> d_sweep_chunks;
  int num_sweep_chunks = 0;
  DevBuf<unsigned char> d_pose_free, d_intr_free, d_pt_free;
  DevBuf<double> d_prior_R0;
  DevBuf<SchurBlock> d_blocks;
  DevBuf<SchurChunk> d_chunks[3];
  DevBuf<SchurCluster> d_clusters;
  // k_schur_rows (round 4): the clusters with their row counts, sorted by row class then length; rows_ok: the set-up built
  // them and the clusters take k_schur_rows (otherwise: k_point_front + k_schur_clusters)
  DevBuf<SchurRowsCluster> d_rows_clusters;
  DevBuf<int> d_rows_lists;                 // k_schur_rows: kRowsLists ints per cluster (internal.h)
  DevBuf<unsigned> d_rows_emit;             // k_schur_rows: the emit maps of the cluster shapes present (rows_emit_map)
  DevBuf<unsigned long long> d_rows_lanes;  // k_schur_rows: per point, which observation each lane of its row takes (internal.h)
  int rows_class_count[kRowsClasses] = {};  // (statistics)
  bool rows_ok = false;
  bool rows_generic = false;  // some cluster has three camera slots: the general form of the intrinsics entries
  DevBuf<PartialReduce> d_reduce_tasks;
  int num_reduce_tasks = 0;
  DevBuf<int> d_cl_tab, d_cl_lists;  // per cluster: slot table of its blocks; its image and camera lists (-1 padded)
  DevBuf<unsigned short> d_obs_meta, d_q_meta;
  DevBuf<unsigned char> d_pt_clustered;
  int num_clusters = 0, num_slots[3] = {0, 0, 0};
  ClusterShape cl_shape{16, 3};
  long long clustered_points = 0, cluster_partials = 0;
  double cluster_flops = 0.0;
  DevBuf<int2> d_terms[3];
  int num_blocks = 0, num_chunks[3] = {0, 0, 0};
  long long num_terms[3] = {0, 0, 0};
  // ---- device: parameters (current x, candidate, initial) ----
  DevBuf<double> d_poses, d_intr, d_points, d_cposes, d_cintr, d_cpoints, d_poses0, d_intr0, d_points0;
  DevBuf<double> d_camrec, d_ccamrec;
  bool camrec_current = false;  // d_camrec holds the records of d_poses (an accepted step swaps in the candidate's)
  // ---- device: linearisation ----
  DevBuf<double> d_R, d_Jp, d_Jc, d_Jk, d_Cu, d_gu, d_Gi, d_h, d_scale_cam, d_scale_pt;
  DevBuf<double> d_sweep_partial, d_camsum /* img_rec | cam_rec */, d_img_intr_tmp, d_cam_partial;
  DevBuf<double> d_prior_res, d_prior_jac, d_prior_cost;
  DevBuf<double> d_Epose, d_Eintr, d_Wk, d_part[3];
  DevBuf<double> d_M, d_L, d_y, d_diag_ws, d_delta_cam, d_delta_pts, d_norm_partial, d_step_partial, d_scal;
  DevBuf<double> d_rnorm, d_perr;
  DevBuf<int> d_pt_orig;      // internal point -> caller's index (read-backs are permuted on the device)
  DevBuf<int> d_perm32;       // device set-up: point-major position -> caller's observation index (`perm` on demand)
  DevBuf<double> d_pts_out;   // [NP][3] staging in the caller's order
  double* d_img_rec = nullptr;
  double* d_cam_rec = nullptr;
  CholStructure chol_struct;
  bool prereduced = false;       // the long runs of the current block partials are pre-reduced (k_partial_reduce's tasks rode in the evaluation: PartialRide)
  bool sweep_rode_along = false; // the last launch_front carried the camera sweep's chunks (k_schur_rows<.., SWEEP>): no separate sweep launch
  bool setup_batched = false;    // build() collects the set-up's small uploads (upload_batch_begin): finish_structure does not synchronise
  bool M_is_clean = false;       // d_M holds zeros outside the entries the assembly writes
  // cleared when a persistent factorisation launch had to give up; a session created within the next
  // kPersistCooldownSessions sessions of the process starts without it (a device shared with another tenant would
  // otherwise pay the 0.3 s time-out once per bundle_adjustment() call). Decided in start(): only a single-rank session
  // that HAS a persistent schedule consumes a session of the cool-down; the ranks of a sharded solve agree on one value
  // in join_ranks (a rank that re-solves after a time-out issues a collective - every rank must take that branch).
  bool allow_persistent = true;
  bool persist_decided = false;
  bool speculate_on = true;   // MAVBA_SPECULATE, read by start() for every solve (tests compare both ways in one process)
  void decide_persistent();
  // The reduced camera matrix is assembled in the factorisation's elimination order: n_mat (multiple of 64)
  // columns, image i's pose block at h_off_img[i], camera c's intrinsics block at h_off_cam[c]; col_var maps a
  // matrix column back to the variable (index into the length-n_pad camera vectors), -1 for padding.
  int n_mat = 64, nd_parts = 0;
  std::vector<int> h_off_img, h_off_cam, h_col_var;
  DevBuf<int> d_off, d_col_var;
  DevBuf<double> d_ymat;
  // multi-rank: the structurally non-zero lower tiles of the matrix, packed for the all-reduce
  DevBuf<int> d_ar_tiles;   // slots (tile store) of the tiles that travel
  DevBuf<double> d_ar_buf;
  int num_ar_tiles = 0;

  // ---- LM state (Ceres TrustRegionMinimizer / LevenbergMarquardtStrategy) ----
  bool evaluated = false, scales_ready = false, started = false, assembled = false;
  double radius = 1e4, decrease_factor = 2.0;
  double cost = 0.0, x_norm = 0.0, grad_max = 0.0, abs_gtol = 0.0, initial_cost = 0.0;
  int iteration = 0, invalid_steps = 0, n_success = 0, n_fail = 0;
  int termination = MAVBA_TERM_RUNNING;

  // ---- multi-GPU ----
  mavba_allreduce_fn ar_fn = nullptr;  // hook: the caller's collective (host-synchronised)
  void* ar_ctx = nullptr;
  void* rccl_comm = nullptr;           // native: ncclAllReduce enqueued on the session's stream, no host synchronisation
  bool rccl_comm_owned = true;         // false: borrowed from the process-wide group of the in-process ranks (multi_gpu.hip)
  std::shared_ptr<RcclGuard> rccl_guard;  // borrowed handles only: the group's abort guard
  bool sharded() const { return (ar_fn || rccl_comm) && (world > 1 || force_exchange); }
  bool force_exchange = false;         // run the multi-rank protocol even with one rank (tests of the native path)
  int rank = 0, world = 1;

  // ---- profiling ----
  std::vector<KernelTimer> timers;
  struct Pending { int idx; hipEvent_t a, b; bool own_a = true; };  // own_a false: `a` is the previous bracket's `b`
  std::vector<Pending> pending;
  std::vector<hipEvent_t> ev_pool;

  ~mavba_session() {
    // the buffers go back to the process-wide pool: nothing may still be running on them
    if (st) (void)hipStreamSynchronize(st);
    if (rccl_comm && rccl_comm_owned) rccl_comm_destroy(rccl_comm);
    if (lm_pub) lm_pub_free(lm_pub);
    HostSpare<long long>::give(perm); HostSpare<int>::give(h_oimg); HostSpare<double>::give(h_points0);
    for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& e : ev_pool) (void)hipEventDestroy(e);
    if (st) stream_release(st, device);
  }

  int timer_index(const char* name) {
    for (size_t i = 0; i < timers.size(); ++i) if (timers[i].name == name) return (int)i;
    timers.push_back(KernelTimer{name, 0, 0.0});
    return (int)timers.size() - 1;
  }
  hipEvent_t get_event() {
    if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
    hipEvent_t e; HIP_OK(hipEventCreate(&e)); return e;
  }
  template <typename F>
  void timed(const char* name, F&& f) {
    if (!opt.profile_kernels) { f(); return; }
    Pending p; p.idx = timer_index(name); p.a = get_event(); p.b = get_event();
    HIP_OK(hipEventRecord(p.a, st));
    f();
    HIP_OK(hipEventRecord(p.b, st));
    pending.push_back(p);
  }
  // Two consecutive brackets around ONE call: f(mid) records `mid` on the stream where its first part ends.
  template <typename F>
  void timed_split(const char* first, const char* second, F&& f) {
    if (!opt.profile_kernels) { f(nullptr); return; }
    Pending p, q;
    p.idx = timer_index(first); q.idx = timer_index(second);
    p.a = get_event(); p.b = get_event(); q.a = p.b; q.b = get_event(); q.own_a = false;
    HIP_OK(hipEventRecord(p.a, st));
    f(p.b);
    HIP_OK(hipEventRecord(q.b, st));
    pending.push_back(p); pending.push_back(q);
  }
  void flush_timers() {  // only after a stream synchronisation
    for (auto& p : pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { timers[p.idx].launches++; timers[p.idx].total_ms += ms; }
      if (p.own_a) ev_pool.push_back(p.a);
      ev_pool.push_back(p.b);
    }
    pending.clear();
  }
  void sync() { HIP_OK(hipStreamSynchronize(st)); HIP_OK(hipGetLastError()); release_staged(st); flush_timers(); }
  // device -> pageable host memory (synchronises the stream)
  void download(void* dst, const void* src, size_t bytes) { HIP_OK(copy_d2h_staged_sync(dst, src, bytes, st)); release_staged(st); }

  void allreduce(double* dptr, long long count, int op) {
    if (!sharded()) return;
    if (rccl_comm) {
      if (rccl_guard) {
        std::shared_lock<std::shared_mutex> lk(rccl_guard->m);
        if (rccl_guard->aborted) throw Failure(MAVBA_ERR_HIP, "the communicator group was aborted (another rank failed)");
        rccl_allreduce(rccl_comm, dptr, count, op, st);
      } else {
        rccl_allreduce(rccl_comm, dptr, count, op, st);
      }
      return;
    }
    sync();
    if (ar_fn(ar_ctx, dptr, count, op) != 0) throw Failure(MAVBA_ERR_HIP, "all-reduce hook failed");
  }

  SweepArgs sweep_args(const double* camrec, const double* intr, const double* points) {
    SweepArgs a;
    a.N = N; a.Nstride = Nstride; a.NI = NI; a.NC = NC; a.KMAX = KMAX;
    a.uv = d_uv.p; a.obs_img = d_obs_img.p; a.obs_pt = d_obs_pt.p;
    a.camrec = camrec; a.intr = intr; a.img_cam = d_img_cam.p; a.cam_model = d_cam_model.p;
    a.points = points;
    a.pt_active = h_pt_removed.empty() ? nullptr : d_pt_active.p;
    a.loss_b = opt.loss_scale_factor * opt.loss_scale_factor; a.loss_inv_b = 1.0 / a.loss_b;
    a.R = d_R.p; a.Jp = d_Jp.p; a.Jc = d_Jc.p; a.Jk = d_Jk.p; a.cost_partial = d_sweep_partial.p;
    return a;
  }
  void read_scalars(double* h) {
    HIP_OK(hipMemcpyAsync(h, d_scal.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, st));
    sync();
  }

  // raw != null: obs_uv / obs_image / obs_point / points of the problem are DEVICE arrays given by `raw` (P's own pointers to
  // them are not read); needs a problem without dropped all-constant blocks
  void build(const mavba_problem* P, const DeviceRaw* raw = nullptr);
void intr_entries_on_device(const std::vector<unsigned char>& cam_active, std::vector<int>& q_start, std::vector<int>& q_cam);
    void order_on_device(const mavba_problem* P, std::vector<int>& img_start, const DeviceRaw* raw);  // device_setup.hip
  void order_on_host(const mavba_problem* P, const long long* keptp, std::vector<int>& img_start,
                     std::unique_ptr<PinnedBuf<double2>>& uv_h, std::unique_ptr<PinnedBuf<int>>& opt_h,
                     std::unique_ptr<PinnedBuf<double2>>& im_uv_h, std::unique_ptr<PinnedBuf<int>>& im_pt_h);
  void ensure_perm_host();
  void derive_free_flags();
  void finish_structure();
  void choose_elimination_order(const std::vector<SchurBlock>& blocks);
  void reset_state();
  // next_radius > 0: the trust-region radius of the linear solve that follows (the front end then writes the Schur entry
  // records in the same pass); <= 0: unknown
  void evaluate(double next_radius = -1.0);
  // the launches of evaluate() without reading the scalars back. spec.dec != null: the SPECULATIVE evaluation at the
  // candidate point - every kernel returns at once unless k_lm_snapshot accepted the step, the front end takes the radius
  // from its decision (next_radius only says "with entries")
  void evaluate_enqueue(double next_radius = -1.0, const LmSpec& spec = lm_spec_off());
  void launch_front(double r, bool entries, const LmSpec& spec = lm_spec_off(), const CamSweepArgs* with_sweep = nullptr);
  // ---- speculative evaluation (lm_decide.h) ----
  DevBuf<double> d_lm_dec;       // {code, radius} of k_lm_snapshot's decision
  double* lm_pub = nullptr;      // host-mapped slot the snapshot kernel publishes to (null: the copy + synchronise read-back)
  double lm_seq = 0.0;           // sequence number of the last publication asked for
  LmSpec lm_spec(bool pending_eval) const {
    LmSpec sp{};
    sp.dec = nullptr; sp.scal = d_scal.p;
    sp.radius = radius; sp.decrease_factor = decrease_factor;
    sp.ptol = opt.parameter_tolerance; sp.ftol = opt.function_tolerance; sp.min_rel_dec = opt.min_relative_decrease;
    sp.max_radius = opt.max_trust_region_radius; sp.abs_gtol = abs_gtol; sp.pending_eval = pending_eval ? 1 : 0;
    return sp;
  }
  bool wait_publication(double* h);  // polls lm_pub for lm_seq; false: timed out (the caller synchronises the stream instead)
  void build_front_tiles(const std::vector<int>& q_start);
  void build_tiles(const std::vector<int>& q_start, int first, DevBuf<FrontTile>& out, int& count);
  void ensure_planes();
  // (the front end runs inside the cluster kernel - k_schur_rows, which masks filtered points itself)
  bool fused_now() const { return fused_ok && rows_ok; }
  int eval_cost_rows() const { return front_ok ? eval_rows : (N > 0 ? jacobian_sweep_grid(N) : 0); }
  void take_evaluation(const double* h) { cost = h[SC_COST]; grad_max = h[SC_GRAD_MAX]; x_norm = std::sqrt(h[SC_XNORM2]); }
  void assemble(double r);
  void solve_linear(double r);
  void linear_step(double r, double* h_scal);
  void candidate(double r, double* h_scal);
  // tail != null: the candidate's scalar reductions are NOT launched - the caller gets them (and their count) to run them in
  // one work-group with the decision (launch_lm_tail)
  void candidate_enqueue(double r, ReduceTasks* tail = nullptr, int* tail_count = nullptr);
  // Launch merging for problems whose reduced system one work-group solves (a local window): the evaluation's reductions in
  // one launch (k_eval_small) and the camera update inside the solve's launch (k_chol_small<true>). MAVBA_MERGE=0: off.
  bool merge_small() const;
  bool merge_on = true;             // MAVBA_MERGE (read in start())
  double pub_wait_seconds = 0.0;    // host time spent polling for the device's decision (MAVBA_SETUP_TIMING prints it)
  bool cameras_updated = false;     // the solve's launch has applied the camera update for ...
  double cameras_updated_r = 0.0;   // ... this radius
  void start();
  int iterate(int max_iters, int* done);
  void point_errors(double* out);
  void restart();  // a new solve from the current parameters (LM state, Jacobi scales; the structure stays)
  long long filter_points(double max_error, const unsigned char* keep, unsigned char* removed_out, double* errors_out);
  void restore_initial_params() {  // x <- the parameters the session was built with (what ceres leaves the user with after NUMERICAL_FAILURE)
    HIP_OK(hipMemcpyAsync(d_poses.p, d_poses0.p, (size_t)NI * 6 * 8, hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(d_intr.p, d_intr0.p, (size_t)NC * 9 * 8, hipMemcpyDeviceToDevice, st));
    HIP_OK(hipMemcpyAsync(d_points.p, d_points0.p, (size_t)NP * 3 * 8, hipMemcpyDeviceToDevice, st));
    camrec_current = false; evaluated = false; front_valid = false;
  }
  void apply_filter_state();  // counts, used / free flags, fixed cost from h_pt_removed (empty = the problem as built)
  void to_caller_points(const double* internal, double* out, int width) const {
    for (int q = 0; q < NP; ++q)
      for (int e = 0; e < width; ++e) out[(size_t)h_pt_orig[q] * width + e] = internal[(size_t)q * width + e];
  }
  void fill_result(mavba_result* r);
};
#endif  // MAVBA_SESSION_H_
