// session.hip — host side of the MI355X bundle-adjustment backend: problem indexing,
// device state, the Levenberg-Marquardt trust-region loop, and the C ABI of include/mavba.h.
//
// The LM loop runs on the host in C++ and drives the HIP kernels of kernels.hip /
// dense_chol.hip through one stream; one 128-byte scalar read-back per evaluation and per
// candidate step is the only device->host traffic inside the loop.
//
// Semantics restated (reference file:line, /root/reference):
//   src/base3d/bundle_adjustment.cc:553-569  ceres::Solve, LM + SPARSE_SCHUR, options
//   Ceres 1.8 trust_region_minimizer.cc / levenberg_marquardt_strategy.cc  
//   src/base3d/bundle_adjustment.cc:575-598  point3D_errors
//   src/base3d/bundle_adjustment.cc:139-225  pose_refinement
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
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/mavba.h"
#include "ba_math.h"
#include "internal.h"

namespace mavba {

static thread_local std::string g_last_error;

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
  void upload(const std::vector<T>& h, hipStream_t st) {
    alloc(std::max<size_t>(h.size(), 1));
    if (!h.empty()) HIP_OK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st));
  }
  void upload(const T* h, size_t count, hipStream_t st) {
    alloc(std::max<size_t>(count, 1));
    if (count) HIP_OK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, st));
  }
  void zero(hipStream_t st) { if (n) HIP_OK(hipMemsetAsync(p, 0, n * sizeof(T), st)); }
};

// ---- caching device allocator (declared in internal.h) ----------------------------------------------------
namespace {
struct DevicePool {
  std::mutex m;
  std::multimap<std::pair<int, size_t>, void*> free_blocks;      // (device, class size) -> block
  std::unordered_map<void*, std::pair<int, size_t>> live;        // every block handed out by device_alloc
  std::multimap<int, hipStream_t> streams;                        // idle streams per device
  size_t cached = 0, cap = 0;
  DevicePool() {
    const char* e = std::getenv("MAVBA_POOL_MB");
    cap = (size_t)(e ? std::atoll(e) : 16384) << 20;
  }
  ~DevicePool() {}  // blocks are left to the driver at process exit (the HIP runtime may already be gone)
  // size classes 1, 1.25, 1.5, 1.75 x 2^k (>= 256 B): at most 25 % slack, few distinct sizes
  static size_t size_class(size_t bytes) {
    size_t c = 256;
    while (c < bytes) c <<= 1;
    if (c >= 1024) {
      const size_t q = c >> 3;
      for (int k = 5; k <= 7; ++k) if (bytes <= (size_t)k * q) return (size_t)k * q;
    }
    return c;
  }
};
DevicePool& pool() { static DevicePool* p = new DevicePool; return *p; }
}  // namespace

hipError_t device_alloc(void** out, size_t bytes) {
  DevicePool& P = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t cls = DevicePool::size_class(bytes);
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.free_blocks.find({dev, cls});
    if (it != P.free_blocks.end()) {
      *out = it->second;
      P.free_blocks.erase(it);
      P.cached -= cls;
      P.live[*out] = {dev, cls};
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, cls);
  if (e != hipSuccess) {
    // out of memory: give the cache back and try once more
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> g(P.m);
      for (auto& kv : P.free_blocks) drop.push_back(kv.second);
      P.free_blocks.clear();
      P.cached = 0;
    }
    for (void* q : drop) (void)hipFree(q);
    (void)hipGetLastError();
    e = hipMalloc(out, cls);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> g(P.m);
  P.live[*out] = {dev, cls};
  return hipSuccess;
}

void device_free(void* p) {
  if (!p) return;
  DevicePool& P = pool();
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.live.find(p);
    if (it != P.live.end()) {
      const auto key = it->second;
      P.live.erase(it);
      if (P.cached + key.second <= P.cap) {
        P.free_blocks.insert({key, p});
        P.cached += key.second;
        return;
      }
    }
  }
  (void)hipFree(p);
}

// Streams are cached too (hipStreamCreate + hipStreamDestroy cost ~2 ms per session, more than a local-BA solve).
hipError_t stream_acquire(hipStream_t* st) {
  DevicePool& P = pool();
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.streams.find(dev);
    if (it != P.streams.end()) { *st = it->second; P.streams.erase(it); return hipSuccess; }
  }
  return hipStreamCreate(st);
}
void stream_release(hipStream_t st, int dev) {  // the caller has synchronised it
  if (!st) return;
  DevicePool& P = pool();
  std::lock_guard<std::mutex> g(P.m);
  if (P.streams.count(dev) < 8) { P.streams.insert({dev, st}); return; }
  (void)hipStreamDestroy(st);
}

// Persistent host worker threads for the set-up passes: creating and joining 16 threads costs ~0.5-1 ms on a
// 256-thread host and build() has ~20 parallel passes. host_run(T, body) runs body(0..T-1) on the workers and
// returns when all are done; calls are serialised (sessions may be created from several user threads).
class HostWorkers {
  std::vector<std::thread> workers;
  std::mutex m, run_m;
  std::condition_variable cv_start, cv_done;
  const std::function<void(int)>* job = nullptr;
  int job_T = 0, remaining = 0;
  unsigned long long generation = 0;

  void loop(int id) {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_start.wait(lk, [&] { return generation != seen; });
      seen = generation;
      if (id < job_T) {
        const std::function<void(int)>* j = job;
        lk.unlock();
        (*j)(id);
        lk.lock();
        if (--remaining == 0) cv_done.notify_one();
      }
    }
  }

 public:
  explicit HostWorkers(int n) {
    for (int i = 0; i < n; ++i) { workers.emplace_back([this, i] { loop(i); }); workers.back().detach(); }
  }
  int size() const { return (int)workers.size(); }
  void run(int T, const std::function<void(int)>& body) {
    std::lock_guard<std::mutex> one(run_m);
    std::unique_lock<std::mutex> lk(m);
    job = &body; job_T = T; remaining = T; ++generation;
    cv_start.notify_all();
    cv_done.wait(lk, [&] { return remaining == 0; });
    job = nullptr;
  }
};
static int host_threads() {
  static const int n = (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
  return n;
}
static void host_run(int T, const std::function<void(int)>& body) {
  if (T <= 1) { body(0); return; }
  static HostWorkers* W = new HostWorkers(host_threads());  // never destroyed: the workers outlive static destruction
  W->run(std::min(T, W->size()), body);
}

// Host scratch array WITHOUT value-initialisation (std::vector<T>(n) clears the memory first: ~1 ms per 10 MB,
// and the set-up shuffles ~100 MB of such arrays that are fully overwritten anyway).
template <typename T>
struct HostBuf {
  std::unique_ptr<T[]> p;
  size_t n = 0;
  explicit HostBuf(size_t count) : p(new T[std::max<size_t>(count, 1)]), n(count) {}
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* data() { return p.get(); }
};

// Stable counting sort of items 0..n-1 by key(i) in [0, nkeys) on a few threads (per-thread histograms turned
// into per-thread cursors): start[k] = first position of key k, emit(i, position) is called once per item.
// The result does not depend on the number of threads.
template <typename KeyFn, typename EmitFn>
static void counting_sort_parallel(long long n, int nkeys, KeyFn key, std::vector<int>& start, EmitFn emit) {
  int T = n >= 200000 ? host_threads() : 1;
  while (T > 1 && (size_t)T * nkeys > ((size_t)64 << 20)) T /= 2;
  std::vector<std::vector<int>> hist(T);
  auto run = [&](const std::function<void(int)>& body) { host_run(T, body); };
  run([&](int t) {
    hist[t].assign((size_t)nkeys, 0);
    for (long long i = n * t / T; i < n * (t + 1) / T; ++i) hist[t][key(i)]++;
  });
  start.assign((size_t)nkeys + 1, 0);
  int pos = 0;
  for (int k = 0; k < nkeys; ++k) {
    start[k] = pos;
    for (int t = 0; t < T; ++t) { const int c = hist[t][k]; hist[t][k] = pos; pos += c; }
  }
  start[nkeys] = pos;
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
  // Points are renumbered at session creation (sorted by their image lists, so that neighbours in the order
  // see the same images: the Schur-complement clusters rely on it). h_pt_orig[internal] = caller's index.
  std::vector<int> h_pt_orig;
  std::vector<unsigned char> h_pose_const, h_intr_const_in, h_pt_const_in;
  std::vector<unsigned char> h_img_used, h_cam_used, h_pt_used;
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
  DevBuf<SweepChunk> d_sweep_chunks;
  int num_sweep_chunks = 0;
  DevBuf<unsigned char> d_pose_free, d_intr_free, d_pt_free;
  DevBuf<double> d_prior_R0;
  DevBuf<SchurBlock> d_blocks;
  DevBuf<SchurChunk> d_chunks[3];
  DevBuf<SchurCluster> d_clusters;
  DevBuf<PartialReduce> d_reduce_tasks;
  int num_reduce_tasks = 0;
  DevBuf<int> d_cl_tab;
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
  double* d_img_rec = nullptr;
  double* d_cam_rec = nullptr;
  CholStructure chol_struct;
  // The reduced camera matrix is assembled in the factorisation's elimination order: n_mat (multiple of 64)
  // columns, image i's pose block at h_off_img[i], camera c's intrinsics block at h_off_cam[c]; col_var maps a
  // matrix column back to the variable (index into the length-n_pad camera vectors), -1 for padding.
  int n_mat = 64, nd_parts = 0;
  std::vector<int> h_off_img, h_off_cam, h_col_var;
  DevBuf<int> d_off, d_col_var;
  DevBuf<double> d_ymat;
  // multi-rank: the structurally non-zero lower tiles of the matrix, packed for the all-reduce
  DevBuf<int2> d_ar_tiles;
  DevBuf<double> d_ar_buf;
  int num_ar_tiles = 0;

  // ---- LM state (Ceres TrustRegionMinimizer / LevenbergMarquardtStrategy) ----
  bool evaluated = false, scales_ready = false, started = false, assembled = false;
  double radius = 1e4, decrease_factor = 2.0;
  double cost = 0.0, x_norm = 0.0, grad_max = 0.0, abs_gtol = 0.0, initial_cost = 0.0;
  int iteration = 0, invalid_steps = 0, n_success = 0, n_fail = 0;
  int termination = MAVBA_TERM_RUNNING;

  // ---- multi-GPU ----
  mavba_allreduce_fn ar_fn = nullptr;
  void* ar_ctx = nullptr;
  int rank = 0, world = 1;

  // ---- profiling ----
  std::vector<KernelTimer> timers;
  struct Pending { int idx; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> ev_pool;

  ~mavba_session() {
    // the buffers go back to the process-wide pool: nothing may still be running on them
    if (st) (void)hipStreamSynchronize(st);
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
  void flush_timers() {  // only after a stream synchronisation
    for (auto& p : pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { timers[p.idx].launches++; timers[p.idx].total_ms += ms; }
      ev_pool.push_back(p.a); ev_pool.push_back(p.b);
    }
    pending.clear();
  }
  void sync() { HIP_OK(hipStreamSynchronize(st)); HIP_OK(hipGetLastError()); flush_timers(); }

  void allreduce(double* dptr, long long count, int op) {
    if (!ar_fn || world <= 1) return;
    sync();
    if (ar_fn(ar_ctx, dptr, count, op) != 0) throw Failure(MAVBA_ERR_HIP, "all-reduce hook failed");
  }

  SweepArgs sweep_args(const double* camrec, const double* intr, const double* points) {
    SweepArgs a;
    a.N = N; a.Nstride = Nstride; a.NI = NI; a.NC = NC; a.KMAX = KMAX;
    a.uv = d_uv.p; a.obs_img = d_obs_img.p; a.obs_pt = d_obs_pt.p;
    a.camrec = camrec; a.intr = intr; a.img_cam = d_img_cam.p; a.cam_model = d_cam_model.p;
    a.points = points;
    a.loss_b = opt.loss_scale_factor * opt.loss_scale_factor; a.loss_inv_b = 1.0 / a.loss_b;
    a.R = d_R.p; a.Jp = d_Jp.p; a.Jc = d_Jc.p; a.Jk = d_Jk.p; a.cost_partial = d_sweep_partial.p;
    return a;
  }
  void read_scalars(double* h) {
    HIP_OK(hipMemcpyAsync(h, d_scal.p, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, st));
    sync();
  }

  void build(const mavba_problem* P);
  void derive_free_flags();
  void finish_structure();
  void choose_elimination_order(const std::vector<SchurBlock>& blocks);
  void reset_state();
  void evaluate();
  void evaluate_enqueue();  // the launches of evaluate() without reading the scalars back
  void take_evaluation(const double* h) { cost = h[SC_COST]; grad_max = h[SC_GRAD_MAX]; x_norm = std::sqrt(h[SC_XNORM2]); }
  void assemble(double r);
  void solve_linear(double r);
  void candidate(double r, double* h_scal);
  void start();
  int iterate(int max_iters, int* done);
  void point_errors(double* out);
  void to_caller_points(const double* internal, double* out, int width) const {
    for (int q = 0; q < NP; ++q)
      for (int e = 0; e < width; ++e) out[(size_t)h_pt_orig[q] * width + e] = internal[(size_t)q * width + e];
  }
  void fill_result(mavba_result* r);
};

// ===========================================================================
// Problem indexing (host) — the device-side counterpart of
// _bundle_adjustment_extract_data / _fill_problem (bundle_adjustment.cc:228-387): the shim
// has already selected images and observations; here they are re-ordered point-major and
// the block structure of the reduced camera system is enumerated.
// ===========================================================================
void mavba_session::derive_free_flags() {
  h_pose_free.assign((size_t)NI * 6, 0);
  h_intr_free.assign((size_t)NC * 9, 0);
  h_pt_free.assign(NP, 0);
  for (int i = 0; i < NI; ++i) {
    if (!h_img_used[i]) continue;
    const unsigned m = h_pose_const[i];
    for (int e = 0; e < 6; ++e) {
      const bool c = e < 3 ? (m & MAVBA_CONST_RVEC) != 0 : (m & (MAVBA_CONST_TX << (e - 3))) != 0;
      h_pose_free[(size_t)i * 6 + e] = c ? 0 : 1;
    }
  }
  any_intr_free = false;
  for (int c = 0; c < NC; ++c) {
    if (!h_cam_used[c] || h_intr_const_in[c]) continue;
    for (int k = 0; k < model_k(h_cam_model[c]); ++k) h_intr_free[(size_t)c * 9 + k] = 1;
    any_intr_free = true;
  }
  for (int p = 0; p < NP; ++p) h_pt_free[p] = (h_pt_used[p] && !h_pt_const_in[p]) ? 1 : 0;
  long long np = 0;
  for (unsigned char f : h_pose_free) np += f;
  for (unsigned char f : h_intr_free) np += f;
  for (unsigned char f : h_pt_free) np += 3 * f;
  num_parameters_reduced = np;
}

void mavba_session::build(const mavba_problem* P) {
  const double t0 = now_s();
  const bool tt = std::getenv("MAVBA_SETUP_TIMING") != nullptr;
  double tl = t0;
  auto lap = [&](const char* what) { if (tt) { const double t = now_s(); std::fprintf(stderr, "[setup] %-28s %8.2f ms\n", what, 1e3 * (t - tl)); tl = t; } };
  NI = P->num_images; NC = P->num_cameras; NP = P->num_points; NO_all = P->num_obs;
  if (NI < 0 || NC < 0 || NP < 0 || NO_all < 0) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "negative size");
  if (NO_all >= (1ll << 31) - 64) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "more than 2^31 observations per session");
  if (NI > 16000) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "more than 16000 images per session (dense block index)");
  if (!(opt.loss_scale_factor > 0.0)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "loss_scale_factor must be > 0");
  if (NI > 0 && (!P->poses || !P->image_camera)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null pose arrays");
  if (NC > 0 && (!P->intrinsics || !P->camera_model)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null camera arrays");
  if (NP > 0 && !P->points) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null point array");
  if (NO_all > 0 && (!P->obs_uv || !P->obs_image || !P->obs_point)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null observation arrays");
  h_cam_model.assign(P->camera_model, P->camera_model + NC);
  h_img_cam.assign(P->image_camera, P->image_camera + NI);
  int kmax = 4;
  for (int c = 0; c < NC; ++c) {
    if (h_cam_model[c] < 1 || h_cam_model[c] > 3) throw Failure(MAVBA_ERR_BAD_MODEL, "camera model code not in {1,2,3}");
    kmax = std::max(kmax, model_k(h_cam_model[c]));
  }
  KMAX = kmax;  // 4, 8 or 9: number of intrinsics columns the Jacobian planes carry
  for (int i = 0; i < NI; ++i)
    if (h_img_cam[i] < 0 || h_img_cam[i] >= NC) throw Failure(MAVBA_ERR_BAD_INDEX, "image_camera out of range");
  {
    int bad = 0;
    parallel_ranges(NO_all, [&](long long b0, long long b1) {
      int local = 0;
      for (long long o = b0; o < b1; ++o)
        local |= (P->obs_image[o] < 0) | (P->obs_image[o] >= NI) | (P->obs_point[o] < 0) | (P->obs_point[o] >= NP);
      if (local) __atomic_store_n(&bad, 1, __ATOMIC_RELAXED);
    });
    if (bad) throw Failure(MAVBA_ERR_BAD_INDEX, "observation index out of range");
  }
  for (int q = 0; q < P->num_rot_priors; ++q)
    if (P->rot_prior_image[q] < 0 || P->rot_prior_image[q] >= NI) throw Failure(MAVBA_ERR_BAD_INDEX, "rot_prior_image out of range");

  h_poses0.assign(P->poses, P->poses + (size_t)NI * 6);
  h_intr0.assign(P->intrinsics, P->intrinsics + (size_t)NC * 9);
  h_points0.assign(P->points, P->points + (size_t)NP * 3);
  h_pose_const.assign(NI, 0); h_intr_const_in.assign(NC, 0); h_pt_const_in.assign(NP, 0);
  if (P->pose_const) h_pose_const.assign(P->pose_const, P->pose_const + NI);
  if (P->intr_const) h_intr_const_in.assign(P->intr_const, P->intr_const + NC);
  if (P->point_const) h_pt_const_in.assign(P->point_const, P->point_const + NP);

  // Residual blocks without a free parameter block leave the program (ceres
  // RemoveFixedBlocksFromProgram); their cost is the fixed cost.
  h_pt_count_all.assign(NP, 0);
  std::vector<long long> kept;
  kept.reserve((size_t)NO_all);
  fixed_cost = 0.0;
  const double b = opt.loss_scale_factor * opt.loss_scale_factor;
  h_img_used.assign(NI, 0); h_cam_used.assign(NC, 0); h_pt_used.assign(NP, 0);
  // Only an observation whose image is entirely constant (pose AND its camera's intrinsics) on a constant point
  // can leave the program. Without such a combination (the normal case: global BA has no constant points, local
  // BA windows only a few) every observation is kept and the pass is a parallel histogram.
  bool any_const_img = false, any_const_pt = false;
  for (int i = 0; i < NI; ++i) any_const_img |= (h_pose_const[i] & 15u) == 15u && h_intr_const_in[h_img_cam[i]];
  for (int p = 0; p < NP; ++p) any_const_pt |= h_pt_const_in[p] != 0;
  const bool all_kept = !(any_const_img && any_const_pt);
  if (all_kept) {
    // (the per-point counts and the used flags then fall out of the counting sorts below)
    kept.resize((size_t)NO_all);
    parallel_ranges(NO_all, [&](long long b0, long long b1) { for (long long o = b0; o < b1; ++o) kept[o] = o; });
  } else {
    for (long long o = 0; o < NO_all; ++o) {
      const int i = P->obs_image[o], p = P->obs_point[o], c = h_img_cam[i];
      h_pt_count_all[p]++;
      if ((h_pose_const[i] & 15u) == 15u && h_intr_const_in[c] && h_pt_const_in[p]) {
        double rec[9], r[2], w, hr;
        cam_prepare(&h_poses0[(size_t)i * 6], rec);
        obs_residual(h_cam_model[c], rec, &h_intr0[(size_t)c * 9], &h_points0[(size_t)p * 3], P->obs_uv[2 * o],
                     P->obs_uv[2 * o + 1], r);
        cauchy_weight(r[0] * r[0] + r[1] * r[1], b, 1.0 / b, w, hr);
        fixed_cost += hr;
        continue;
      }
      kept.push_back(o);
      h_img_used[i] = 1; h_cam_used[c] = 1; h_pt_used[p] = 1;
    }
  }
  lap("validate + fixed blocks");
  N = (int)kept.size();
  Nstride = std::max(32, round_up(N, 32));
  NPs = std::max(32, round_up(NP, 32));

  // rotation priors: kept when the image's rvec block is free, sorted by image
  std::vector<std::pair<int, int>> pri;  // (image, index)
  for (int q = 0; q < P->num_rot_priors; ++q) {
    const int i = P->rot_prior_image[q];
    if (h_pose_const[i] & MAVBA_CONST_RVEC) {
      double R0[9], r, j[3];
      rot_matrix_colmajor(&P->rot_prior_rvec[3 * q], R0);
      rot_prior_eval(&h_poses0[(size_t)i * 6], R0, P->rot_prior_weight, r, j);
      fixed_cost += 0.5 * r * r;
      continue;
    }
    pri.push_back({i, q});
    h_img_used[i] = 1;
  }
  std::stable_sort(pri.begin(), pri.end());
  num_priors = (int)pri.size();
  prior_weight = P->rot_prior_weight;
  num_residuals = 2 * NO_all + P->num_rot_priors;
  num_residuals_reduced = 2ll * N + num_priors;

  // ---- internal point order: lexicographic by the sorted list of images that see the point ----
  // One stable counting sort of the kept observations by (caller's) point gives every point's bucket; the
  // point-major order is the buckets concatenated in the new point order.
  std::vector<int> pt_new(NP);
  std::vector<int> cstart;
  HostBuf<long long> bucket(N);  // kept observation ids, grouped by caller's point, input order inside
  {
    HostBuf<int> simg(N);
    counting_sort_parallel(N, NP, [&](long long k) { return P->obs_point[kept[k]]; }, cstart,
                           [&](long long k, int at) { bucket[at] = kept[k]; simg[at] = P->obs_image[kept[k]]; });
    lap("  buckets by point");
    if (all_kept)
      for (int p = 0; p < NP; ++p) { h_pt_count_all[p] = cstart[p + 1] - cstart[p]; h_pt_used[p] = h_pt_count_all[p] > 0; }
    parallel_ranges(NP, [&](long long b0, long long b1) {
      for (long long p = b0; p < b1; ++p) std::sort(simg.data() + cstart[p], simg.data() + cstart[p + 1]);
    });
    // Order: by the first four images packed into one 64-bit key (cache-friendly sort of (key, point) pairs),
    // ties by the full list, then by point id - a strict total order, so the result does not depend on the
    // number of threads.
    auto before_full = [&](int a, int b) {
      const int* xa = simg.data() + cstart[a]; const int* xb = simg.data() + cstart[b];
      const int na = cstart[a + 1] - cstart[a], nb = cstart[b + 1] - cstart[b];
      const int n = std::min(na, nb);
      for (int i = 0; i < n; ++i) if (xa[i] != xb[i]) return xa[i] < xb[i];
      if (na != nb) return na < nb;
      return a < b;
    };
    lap("  per-point image lists");
    typedef std::pair<unsigned long long, int> KeyId;
    HostBuf<KeyId> keyed(NP);
    const bool packable = NI < 65535;
    parallel_ranges(NP, [&](long long b0, long long b1) {
      for (long long p = b0; p < b1; ++p) {
        unsigned long long key = 0;
        const int n = cstart[p + 1] - cstart[p];
        for (int i = 0; i < 4; ++i) key = key << 16 | (unsigned long long)(packable && i < n ? simg[cstart[p] + i] : 0xFFFF);
        keyed[p] = KeyId(packable ? key : 0ull, (int)p);
      }
    });
    auto before = [&](const KeyId& a, const KeyId& b) {
      if (a.first != b.first) return a.first < b.first;
      return before_full(a.second, b.second);
    };
    // sorted runs on a few threads, then pairwise merges
    const int T = NP >= 100000 ? host_threads() : 1;
    std::vector<int> cut(T + 1);
    for (int t = 0; t <= T; ++t) cut[t] = (int)((long long)NP * t / T);
    host_run(T, [&](int t) { std::sort(keyed.data() + cut[t], keyed.data() + cut[t + 1], before); });
    for (int w = 1; w < T; w *= 2) {
      const int pairs = (T + 2 * w - 1) / (2 * w);
      host_run(pairs, [&](int q) {
        const int t = q * 2 * w;
        if (t + w < T)
          std::inplace_merge(keyed.data() + cut[t], keyed.data() + cut[t + w], keyed.data() + cut[std::min(t + 2 * w, T)], before);
      });
    }
    lap("  sort points");
    h_pt_orig.resize(NP);
    for (int q = 0; q < NP; ++q) h_pt_orig[q] = keyed[q].second;
    for (int q = 0; q < NP; ++q) pt_new[h_pt_orig[q]] = q;
    auto permute = [&](auto& v, int width) {
      auto old = v;
      parallel_ranges(NP, [&](long long b0, long long b1) {
        for (long long q = b0; q < b1; ++q)
          for (int e = 0; e < width; ++e) v[(size_t)q * width + e] = old[(size_t)h_pt_orig[q] * width + e];
      });
    };
    permute(h_points0, 3); permute(h_pt_const_in, 1); permute(h_pt_count_all, 1); permute(h_pt_used, 1);
  }
  lap("point order");

  // ---- point-major order: the buckets in the new point order ----
  h_pt_start.assign(NP + 1, 0);
  for (int q = 0; q < NP; ++q) h_pt_start[q + 1] = h_pt_start[q] + (cstart[h_pt_orig[q] + 1] - cstart[h_pt_orig[q]]);
  perm.assign(N, 0);
  HostBuf<double2> uv(N);
  HostBuf<int> opt_(N);
  h_oimg.assign(N, 0);
  parallel_ranges(NP, [&](long long q0, long long q1) {
    for (long long q = q0; q < q1; ++q) {
      const int src = cstart[h_pt_orig[q]], cnt = h_pt_start[q + 1] - h_pt_start[q];
      for (int j = 0; j < cnt; ++j) {
        const long long o = bucket[src + j];
        const int a = h_pt_start[q] + j;
        perm[a] = o;
        uv[a] = make_double2(P->obs_uv[2 * o], P->obs_uv[2 * o + 1]);
        h_oimg[a] = P->obs_image[o]; opt_[a] = (int)q;
      }
    }
  });

  lap("point-major sort");
  // ---- image-major view for the camera sweep ----
  std::vector<int> img_start;
  HostBuf<double2> im_uv(N);
  HostBuf<int> im_pt(N);
  counting_sort_parallel(N, NI, [&](long long a) { return h_oimg[a]; }, img_start,
                         [&](long long a, int at) { im_uv[at] = uv[a]; im_pt[at] = opt_[a]; });
  if (all_kept)
    for (int i = 0; i < NI; ++i)
      if (img_start[i + 1] > img_start[i]) { h_img_used[i] = 1; h_cam_used[h_img_cam[i]] = 1; }
  const int kSweepChunk = 2048;
  std::vector<SweepChunk> sweep_chunks;
  std::vector<int> img_chunk_start(NI + 1, 0);
  for (int i = 0; i < NI; ++i) {
    img_chunk_start[i] = (int)sweep_chunks.size();
    for (int b0 = img_start[i]; b0 < img_start[i + 1]; b0 += kSweepChunk)
      sweep_chunks.push_back(SweepChunk{i, b0, std::min(b0 + kSweepChunk, img_start[i + 1])});
  }
  img_chunk_start[NI] = (int)sweep_chunks.size();
  num_sweep_chunks = (int)sweep_chunks.size();
  std::vector<int> cam_img_start(NC + 1, 0), cam_imgs(std::max(NI, 1));
  for (int i = 0; i < NI; ++i) cam_img_start[h_img_cam[i] + 1]++;
  for (int c = 0; c < NC; ++c) cam_img_start[c + 1] += cam_img_start[c];
  {
    std::vector<int> cur(cam_img_start.begin(), cam_img_start.end() - 1);
    for (int i = 0; i < NI; ++i) cam_imgs[cur[h_img_cam[i]]++] = i;
  }
  std::vector<int> prior_img(num_priors), prior_start(NI + 1, 0);
  std::vector<double> prior_R0((size_t)num_priors * 9);
  for (int k = 0; k < num_priors; ++k) {
    prior_img[k] = pri[k].first;
    prior_start[pri[k].first + 1]++;
    rot_matrix_colmajor(&P->rot_prior_rvec[3 * pri[k].second], &prior_R0[(size_t)k * 9]);
  }
  for (int i = 0; i < NI; ++i) prior_start[i + 1] += prior_start[i];

  n_full = 6 * NI + 9 * NC;
  n_pad = std::max(64, round_up(n_full, 64));

  lap("image-major view");
  // ---- uploads of the static data ----
  d_uv.upload(uv.data(), (size_t)N, st); d_obs_img.upload(h_oimg, st); d_obs_pt.upload(opt_.data(), (size_t)N, st); d_pt_start.upload(h_pt_start, st);
  d_im_uv.upload(im_uv.data(), (size_t)N, st); d_im_pt.upload(im_pt.data(), (size_t)N, st);
  d_img_cam.upload(h_img_cam, st); d_cam_model.upload(h_cam_model, st);
  d_sweep_chunks.upload(sweep_chunks, st); d_img_chunk_start.upload(img_chunk_start, st);
  d_cam_img_start.upload(cam_img_start, st); d_cam_imgs.upload(cam_imgs, st);
  d_prior_img.upload(prior_img, st); d_prior_start.upload(prior_start, st); d_prior_R0.upload(prior_R0, st);
  d_pt_count.upload(h_pt_count_all, st);
  d_poses0.upload(h_poses0, st); d_intr0.upload(h_intr0, st); d_points0.upload(h_points0, st);
  const size_t nI = std::max(NI, 1), nC = std::max(NC, 1), nP = std::max(NP, 1);
  d_poses.alloc(nI * 6); d_intr.alloc(nC * 9); d_points.alloc(nP * 3);
  d_cposes.alloc(nI * 6); d_cintr.alloc(nC * 9); d_cpoints.alloc(nP * 3);
  d_camrec.alloc(nI * 9); d_ccamrec.alloc(nI * 9);
  d_R.alloc((size_t)2 * Nstride); d_Jp.alloc((size_t)6 * Nstride); d_Jc.alloc((size_t)12 * Nstride);
  d_Jk.alloc((size_t)2 * KMAX * Nstride);
  d_Cu.alloc((size_t)6 * NPs); d_gu.alloc((size_t)3 * NPs); d_Gi.alloc((size_t)6 * NPs); d_h.alloc((size_t)3 * NPs);
  d_scale_cam.alloc((size_t)n_pad); d_scale_pt.alloc((size_t)3 * NPs);
  d_scale_cam.zero(st); d_scale_pt.zero(st);
  d_sweep_partial.alloc((size_t)jacobian_sweep_grid(std::max(N, 1)) + 8);
  d_camsum.alloc(nI * kImgRec + nC * kCamRec);
  d_camsum.zero(st);
  d_img_rec = d_camsum.p; d_cam_rec = d_camsum.p + (size_t)NI * kImgRec;
  d_img_intr_tmp.alloc(nI * kCamRec);
  d_cam_partial.alloc((size_t)std::max(num_sweep_chunks, 1) * kSweepAcc);
  d_prior_res.alloc(std::max(num_priors, 1)); d_prior_jac.alloc((size_t)std::max(num_priors, 1) * 3);
  d_prior_cost.alloc(std::max(num_priors, 1));
  d_Epose.alloc((size_t)std::max(N, 1) * kPoseRec);
  d_y.alloc(n_pad); d_y.zero(st);  // the matrix-sized buffers follow the elimination order chosen in finish_structure
  d_delta_cam.alloc(n_pad); d_delta_pts.alloc(nP * 3);
  d_norm_partial.alloc((size_t)(512 + 2) * 2); d_step_partial.alloc((size_t)(1024 + 2) * 3);
  d_scal.alloc(SC_COUNT); d_scal.zero(st);
  d_rnorm.alloc(std::max(N, 1)); d_perr.alloc(nP);

  lap("alloc + upload");
  derive_free_flags();
  finish_structure();
  lap("finish_structure total");
  reset_state();
  sync();
  lap("reset + sync");
  setup_seconds = now_s() - t0;
}

// Everything that depends on which parameter blocks are free: flags on the device, the
// intrinsics entries (one per free point x free camera seen by it) and the term / chunk /
// block lists of the Schur complement.
// Elimination order of the reduced camera system + the factorisation's tile structure.
//
// The dependent chain of the blocked Cholesky is one 64-column panel after the other, ~20 us each, and
// at BA sizes that chain - not the flops - is the cost of the solve. Images are connected through the
// points they share; in acquisition order that graph is banded, so a band partition is a nested
// dissection: cut the order into P runs, move every image that has a neighbour in an EARLIER run into the
// separator S, and the remaining parts A_1..A_P are mutually uncoupled. Ordered [A_1 | .. | A_P | S |
// intrinsics] their panels are factorised concurrently and the chain is max|A_p| + |S| instead of the sum.
// P (and for P = 2 the cut position) is chosen to minimise that chain, P = 1 (no dissection) included.
void mavba_session::choose_elimination_order(const std::vector<SchurBlock>& blocks) {
  const int tiles0 = std::max(1, round_up(n_full, 64) / 64);
  // Tree of image sets in elimination order (children before parents, root last); the root also carries the
  // intrinsics blocks. One node = no dissection.
  struct TNode { std::vector<int> imgs; int parent; };
  std::vector<TNode> tn;
  int forced = -1, max_depth = 2;
  if (const char* e = std::getenv("MAVBA_ND_PARTS")) forced = std::atoi(e);  // 0/1 = off, n = force n flat parts
  if (const char* e = std::getenv("MAVBA_ND_DEPTH")) max_depth = std::atoi(e);  // recursion depth of the automatic choice
  const bool can_dissect = NI >= 16 && tiles0 >= 8 && forced != 0 && forced != 1 && max_depth > 0 && (world == 1 || NI <= 4096);
  if (can_dissect) {
    // image adjacency (lower: col < row) from the pose-pose blocks; with shards, the union over ranks
    std::vector<std::vector<int>> lower(NI);
    if (world > 1 && ar_fn) {
      std::vector<double> a((size_t)NI * NI, 0.0);
      for (const SchurBlock& B : blocks)
        if (B.kind == BLK_PP && B.row_ent != B.col_ent) a[(size_t)std::max(B.row_ent, B.col_ent) * NI + std::min(B.row_ent, B.col_ent)] = 1.0;
      DevBuf<double> d;
      d.upload(a, st);
      allreduce(d.p, (long long)NI * NI, 1);
      HIP_OK(hipMemcpyAsync(a.data(), d.p, a.size() * 8, hipMemcpyDeviceToHost, st));
      sync();
      for (int r = 0; r < NI; ++r)
        for (int c = 0; c < r; ++c) if (a[(size_t)r * NI + c] != 0.0) lower[r].push_back(c);
    } else {
      for (const SchurBlock& B : blocks)
        if (B.kind == BLK_PP && B.row_ent != B.col_ent) lower[std::max(B.row_ent, B.col_ent)].push_back(std::min(B.row_ent, B.col_ent));
    }
    const int tail = 9 * NC;
    auto tiles_of = [](int cols) { return (cols + 63) / 64; };
    if (forced > 1) {
      // flat dissection into `forced` runs of the natural order (kept for tests and experiments)
      std::vector<int> cut(forced);
      for (int q = 0; q < forced; ++q) cut[q] = (int)((long long)q * NI / forced);
      std::vector<std::vector<int>> part(forced);
      std::vector<int> sep;
      int run = 0;
      for (int i = 0; i < NI; ++i) {
        while (run + 1 < forced && i >= cut[run + 1]) ++run;
        int m = i;
        for (int c : lower[i]) m = std::min(m, c);
        if (m < cut[run]) sep.push_back(i); else part[run].push_back(i);
      }
      int np = 0;
      for (auto& pr : part) if (!pr.empty()) { tn.push_back(TNode{std::move(pr), -1}); ++np; }
      if (np >= 2) {
        for (auto& t : tn) t.parent = np;
        tn.push_back(TNode{std::move(sep), -1});
      } else {
        tn.clear();
      }
    } else {
      // recursive bisection: M (ascending) -> [A | B | S], S = the members of the second run that have a neighbour
      // in the first; the cut is scanned for the shortest chain max(A, B) + S, and a split must pay
      std::vector<int> pos(NI, -1);
      std::function<std::pair<int, int>(std::vector<int>&&, int, int)> rec = [&](std::vector<int>&& M, int depth, int tl) {
        const int n = (int)M.size();
        const int leaf_tiles = tiles_of(6 * n + tl);
        auto make_leaf = [&]() { tn.push_back(TNode{std::move(M), -1}); return std::make_pair((int)tn.size() - 1, leaf_tiles); };
        if (depth >= max_depth || n < 32 || leaf_tiles < 6) return make_leaf();
        for (int t = 0; t < n; ++t) pos[M[t]] = t;
        std::vector<int> mnp(n);
        for (int t = 0; t < n; ++t) {
          int m = t;
          for (int c : lower[M[t]]) if (pos[c] >= 0) m = std::min(m, pos[c]);
          mnp[t] = m;
        }
        for (int t = 0; t < n; ++t) pos[M[t]] = -1;
        int best = leaf_tiles, best_c = -1;
        for (int c = n / 4; c <= 3 * n / 4; c += std::max(1, n / 64)) {
          int ns = 0;
          for (int t = c; t < n; ++t) ns += mnp[t] < c;
          if (ns == 0 && tl == 0) continue;  // (a separator node needs at least one column)
          const int est = std::max(tiles_of(6 * c), tiles_of(6 * (n - c - ns))) + tiles_of(6 * ns + tl);
          if (est < best) { best = est; best_c = c; }
        }
        if (best_c < 0 || best > leaf_tiles - std::max(2, leaf_tiles / 8)) return make_leaf();
        std::vector<int> A(M.begin(), M.begin() + best_c), B, S;
        for (int t = best_c; t < n; ++t) (mnp[t] < best_c ? S : B).push_back(M[t]);
        if (A.size() < 8 || B.size() < 8) return make_leaf();
        const auto ra = rec(std::move(A), depth + 1, 0);
        const auto rb = rec(std::move(B), depth + 1, 0);
        const int sep_tiles = tiles_of(6 * (int)S.size() + tl);
        tn.push_back(TNode{std::move(S), -1});
        const int me = (int)tn.size() - 1;
        tn[ra.first].parent = me; tn[rb.first].parent = me;
        return std::make_pair(me, std::max(ra.second, rb.second) + sep_tiles);
      };
      std::vector<int> all(NI);
      for (int i = 0; i < NI; ++i) all[i] = i;
      rec(std::move(all), 0, tail);
      if (tn.size() < 3) tn.clear();
    }
  }
  // column offsets in tree order (every node padded to whole tiles), intrinsics at the end of the root
  h_off_img.assign(NI, 0); h_off_cam.assign(NC, 0);
  std::vector<CholNode> tree;
  int col = 0;
  if (tn.size() >= 3) {
    for (size_t t = 0; t < tn.size(); ++t) {
      const int begin = col;
      for (int i : tn[t].imgs) { h_off_img[i] = col; col += 6; }
      if (t + 1 == tn.size()) for (int c = 0; c < NC; ++c) { h_off_cam[c] = col; col += 9; }
      col = std::max(round_up(col, 64), begin + 64);
      tree.push_back(CholNode{begin / 64, col / 64, tn[t].parent});
    }
  } else {
    for (int i = 0; i < NI; ++i) { h_off_img[i] = col; col += 6; }
    for (int c = 0; c < NC; ++c) { h_off_cam[c] = col; col += 9; }
  }
  n_mat = std::max(64, round_up(col, 64));
  h_col_var.assign(n_mat, -1);
  for (int i = 0; i < NI; ++i) for (int e = 0; e < 6; ++e) h_col_var[h_off_img[i] + e] = 6 * i + e;
  for (int c = 0; c < NC; ++c) for (int k = 0; k < 9; ++k) h_col_var[h_off_cam[c] + k] = 6 * NI + 9 * c + k;
  {
    std::vector<int> off(h_off_img);
    off.insert(off.end(), h_off_cam.begin(), h_off_cam.end());
    d_off.upload(off, st);
    d_col_var.upload(h_col_var, st);
  }
  // structurally non-zero tiles (lower) of the permuted matrix
  const int nbt = n_mat / 64;
  std::vector<unsigned char> mark((size_t)nbt * nbt, 0);
  for (const SchurBlock& B : blocks) {
    const int r0 = B.kind == BLK_PP ? h_off_img[B.row_ent] : h_off_cam[B.row_ent];
    const int r1 = r0 + (B.kind == BLK_PP ? 5 : 8);
    const int c0 = B.kind == BLK_II ? h_off_cam[B.col_ent] : h_off_img[B.col_ent];
    const int c1 = c0 + (B.kind == BLK_II ? 8 : 5);
    for (int tr = r0 / 64; tr <= r1 / 64; ++tr)
      for (int tc = c0 / 64; tc <= c1 / 64; ++tc) mark[(size_t)std::max(tr, tc) * nbt + std::min(tr, tc)] = 1;
  }
  if (world > 1 && ar_fn) {
    // The matrix that gets factorised is the SUM over ranks: its structure is the union of the ranks'.
    std::vector<double> h(mark.begin(), mark.end());
    DevBuf<double> d;
    d.upload(h, st);
    allreduce(d.p, (long long)h.size(), 1);
    HIP_OK(hipMemcpyAsync(h.data(), d.p, h.size() * 8, hipMemcpyDeviceToHost, st));
    sync();
    for (size_t t = 0; t < h.size(); ++t) mark[t] = h[t] != 0.0;
  }
  std::vector<std::pair<int, int>> tile_pairs;
  for (int tr = 0; tr < nbt; ++tr)
    for (int tc = 0; tc <= tr; ++tc) if (mark[(size_t)tr * nbt + tc]) tile_pairs.emplace_back(tr, tc);
  HIP_OK(chol_struct.build(nbt, tile_pairs, tree, st));
  nd_parts = chol_struct.nseg > 1 ? chol_struct.num_fronts_max : 0;
  if (world > 1) {
    std::vector<int2> tl;
    std::vector<unsigned char> have((size_t)nbt * nbt, 0);
    for (int t = 0; t < nbt; ++t) { tl.push_back(make_int2(t, t)); have[(size_t)t * nbt + t] = 1; }
    for (const auto& pr : tile_pairs)
      if (!have[(size_t)pr.first * nbt + pr.second]) { have[(size_t)pr.first * nbt + pr.second] = 1; tl.push_back(make_int2(pr.first, pr.second)); }
    num_ar_tiles = (int)tl.size();
    d_ar_tiles.upload(tl, st);
    d_ar_buf.alloc((size_t)num_ar_tiles * 4096 + n_mat);
  }
  d_M.alloc((size_t)(n_mat + 64) * n_mat); d_L.alloc((size_t)(n_mat + 64) * n_mat);
  d_ymat.alloc(n_mat); d_diag_ws.alloc((size_t)n_mat * 64);
}

void mavba_session::finish_structure() {
  const bool tt = std::getenv("MAVBA_SETUP_TIMING") != nullptr;
  double tl = now_s();
  auto lap = [&](const char* what) { if (tt) { const double t = now_s(); std::fprintf(stderr, "[setup]   %-26s %8.2f ms\n", what, 1e3 * (t - tl)); tl = t; } };
  d_pose_free.upload(h_pose_free, st); d_intr_free.upload(h_intr_free, st); d_pt_free.upload(h_pt_free, st);
  std::vector<unsigned char> img_active(NI, 0), cam_active(NC, 0);
  for (int i = 0; i < NI; ++i) for (int e = 0; e < 6; ++e) img_active[i] |= h_pose_free[(size_t)i * 6 + e];
  for (int c = 0; c < NC; ++c) for (int k = 0; k < 9; ++k) cam_active[c] |= h_intr_free[(size_t)c * 9 + k];

  // intrinsics entries
  std::vector<int> q_start(NP + 1, 0), q_pt, q_cam;
  {
    std::vector<int> seen;
    for (int p = 0; p < NP; ++p) {
      q_start[p] = (int)q_pt.size();
      if (!h_pt_free[p]) continue;
      seen.clear();
      for (int a = h_pt_start[p]; a < h_pt_start[p + 1]; ++a) {
        const int c = h_img_cam[h_oimg[a]];
        if (cam_active[c] && std::find(seen.begin(), seen.end(), c) == seen.end()) seen.push_back(c);
      }
      std::sort(seen.begin(), seen.end());
      for (int c : seen) { q_pt.push_back(p); q_cam.push_back(c); }
    }
    q_start[NP] = (int)q_pt.size();
  }
  Q = (int)q_pt.size();
  d_q_start.upload(q_start, st); d_q_pt.upload(q_pt, st); d_q_cam.upload(q_cam, st);
  d_Eintr.alloc((size_t)std::max(Q, 1) * kIntrRec);
  d_Wk.alloc((size_t)std::max(Q, 1) * 27);

  lap("flags + intr entries");
  // ---- point clusters (k_schur_clusters): consecutive points whose images / cameras fit one local list ----
  // pt_mode: 0 = contributes nothing, 1 = clustered, 2 = generic term lists (long tracks, an image seen twice,
  // more shared cameras than a cluster holds)
  std::vector<unsigned char> pt_mode(NP, 0);
  std::vector<unsigned short> obs_meta((size_t)std::max(N, 1), 0xFFFFu), q_meta((size_t)std::max(Q, 1), 0xFFFFu);
  std::vector<SchurCluster> clusters;
  std::vector<int> cl_imgs, cl_cams;  // [cluster][sh.images] / [cluster][sh.cams], ascending, -1 padded
  {
    // Cluster shape: 16 images x 3 cameras (128 rows, 36 tiles). MAVBA_CLUSTER_SHAPE=12 selects 12 x 2 (96 rows,
    // 21 tiles): 42 % fewer matrix instructions per batch, but the smaller image list closes clusters earlier (C3:
    // 2851 clusters of ~70 points instead of 1799 of ~110) and the per-cluster costs eat the gain - same 0.37 ms.
    cl_shape = ClusterShape{16, 3};
    if (const char* e = std::getenv("MAVBA_CLUSTER_SHAPE")) cl_shape = std::atoi(e) == 12 ? ClusterShape{12, 2} : ClusterShape{16, 3};
  }
  const ClusterShape sh = cl_shape;
  const int kClTab = sh.tab(), kClTabPP = sh.tab_pp(), kClTabIP = sh.tab_ip(), kClTabII = sh.tab_ii();
  const int kClImages = sh.images, kClCams = sh.cams;
  {
    bool use_clusters = true;
    if (const char* e = std::getenv("MAVBA_CLUSTERS")) use_clusters = std::atoi(e) != 0;
    // 128 points per cluster amortise the per-cluster costs; small problems get smaller clusters so that there
    // are at least ~2 per CU (a cluster is one work-group; C2: 235 clusters of 128 would leave CUs idle)
    long long nfree = 0;
    for (int p = 0; p < NP; ++p) nfree += h_pt_free[p] != 0;
    int kMaxPoints = (int)std::min<long long>(128, std::max<long long>(kClBatch, round_up((int)(nfree / 512), kClBatch)));
    if (const char* e = std::getenv("MAVBA_CLUSTER_POINTS")) kMaxPoints = std::max(1, std::atoi(e));
    // Greedy over consecutive points, run independently on fixed ranges of points (NOT on "one range per
    // thread": the clusters - and with them the order in which partials are added - must not depend on the
    // machine's core count).
    const int kRange = 4096;
    const int nranges = (NP + kRange - 1) / kRange;
    std::vector<std::vector<SchurCluster>> r_clusters(nranges);
    std::vector<std::vector<int>> r_imgs(nranges), r_cams(nranges);
    auto do_range = [&](int rg) {
      const int r0 = rg * kRange, r1 = std::min(NP, r0 + kRange);
      std::vector<int> cur_i, cur_c, mi, mc, pi;
      int cur_p0 = r0, cur_n = 0;
      auto close = [&](int p_end) {
        if (cur_n > 0) {
          r_clusters[rg].push_back(SchurCluster{cur_p0, p_end});
          for (int k = 0; k < kClImages; ++k) r_imgs[rg].push_back(k < (int)cur_i.size() ? cur_i[k] : -1);
          for (int k = 0; k < kClCams; ++k) r_cams[rg].push_back(k < (int)cur_c.size() ? cur_c[k] : -1);
        }
        cur_i.clear(); cur_c.clear(); cur_n = 0; cur_p0 = p_end;
      };
      for (int p = r0; p < r1; ++p) {
        if (!h_pt_free[p]) continue;
        pi.clear();
        for (int a = h_pt_start[p]; a < h_pt_start[p + 1]; ++a) if (img_active[h_oimg[a]]) pi.push_back(h_oimg[a]);
        const int nq = q_start[p + 1] - q_start[p];
        if (pi.empty() && nq == 0) continue;
        std::sort(pi.begin(), pi.end());
        const bool dup = std::adjacent_find(pi.begin(), pi.end()) != pi.end();
        if (!use_clusters || dup || (int)pi.size() > kClImages || nq > kClCams) { pt_mode[p] = 2; continue; }
        auto merged_sizes = [&]() {
          mi.clear(); mc.clear();
          std::set_union(cur_i.begin(), cur_i.end(), pi.begin(), pi.end(), std::back_inserter(mi));
          std::set_union(cur_c.begin(), cur_c.end(), q_cam.begin() + q_start[p], q_cam.begin() + q_start[p + 1], std::back_inserter(mc));
        };
        merged_sizes();
        if ((int)mi.size() > kClImages || (int)mc.size() > kClCams || cur_n >= kMaxPoints || p - cur_p0 >= kClMaxBatches * kClBatch - 1) {
          close(p);
          merged_sizes();
        }
        if (cur_n == 0) cur_p0 = p;
        cur_i.swap(mi); cur_c.swap(mc);
        ++cur_n;
        pt_mode[p] = 1;
      }
      close(r1);
    };
    parallel_ranges(nranges, [&](long long g0, long long g1) { for (long long g = g0; g < g1; ++g) do_range((int)g); }, 2);
    for (int rg = 0; rg < nranges; ++rg) {
      clusters.insert(clusters.end(), r_clusters[rg].begin(), r_clusters[rg].end());
      cl_imgs.insert(cl_imgs.end(), r_imgs[rg].begin(), r_imgs[rg].end());
      cl_cams.insert(cl_cams.end(), r_cams[rg].begin(), r_cams[rg].end());
    }
  }
  num_clusters = (int)clusters.size();
  cluster_flops = 0.0;
  for (const SchurCluster& c : clusters)  // batches x k-steps x 36 lower tiles x 2*16*16*4
    cluster_flops += (double)((c.p1 - c.p0 + kClBatch - 1) / kClBatch) * (3 * kClBatch / 4) * (double)((sh.rows() / 16) * (sh.rows() / 16 + 1) / 2) * 2048.0;
  // local indices of every clustered observation / intrinsics entry, and which blocks a cluster touches
  std::vector<unsigned char> cl_present((size_t)std::max(num_clusters, 1) * kClTab, 0);
  parallel_ranges(num_clusters, [&](long long c0, long long c1) {
    int loc[kClImagesMax];
    for (long long cl = c0; cl < c1; ++cl) {
      const int* imgs = &cl_imgs[(size_t)cl * kClImages];
      const int* cams = &cl_cams[(size_t)cl * kClCams];
      int ni = 0, nc = 0;
      while (ni < kClImages && imgs[ni] >= 0) ++ni;
      while (nc < kClCams && cams[nc] >= 0) ++nc;
      unsigned char* pres = &cl_present[(size_t)cl * kClTab];
      for (int p = clusters[cl].p0; p < clusters[cl].p1; ++p) {
        if (pt_mode[p] != 1) continue;
        int n = 0;
        for (int a = h_pt_start[p]; a < h_pt_start[p + 1]; ++a) {
          if (!img_active[h_oimg[a]]) continue;
          const int l = (int)(std::lower_bound(imgs, imgs + ni, h_oimg[a]) - imgs);
          obs_meta[a] = (unsigned short)(l << 8 | ((p - clusters[cl].p0) % kClBatch));
          loc[n++] = l;
        }
        for (int x = 0; x < n; ++x)
          for (int y = 0; y < n; ++y)
            if (loc[x] >= loc[y]) pres[kClTabPP + loc[x] * (loc[x] + 1) / 2 + loc[y]] = 1;
        for (int q = q_start[p]; q < q_start[p + 1]; ++q) {
          const int lc = (int)(std::lower_bound(cams, cams + nc, q_cam[q]) - cams);
          q_meta[q] = (unsigned short)(lc << 8 | ((p - clusters[cl].p0) % kClBatch));
          for (int x = 0; x < n; ++x) pres[kClTabIP + lc * kClImages + loc[x]] = 1;
          for (int q2 = q_start[p]; q2 <= q; ++q2) {
            const int lc2 = (int)(std::lower_bound(cams, cams + nc, q_cam[q2]) - cams);
            pres[kClTabII + lc * (lc + 1) / 2 + lc2] = 1;
          }
        }
      }
    }
  }, 64);
  clustered_points = 0;
  for (int p = 0; p < NP; ++p) clustered_points += pt_mode[p] == 1;
  lap("point clusters");
  // term enumeration over a range of points: f(kind, row_ent, col_ent, x, y)
  auto enumerate = [&](int p_begin, int p_end, auto&& f) {
    for (int p = p_begin; p < p_end; ++p) {
      if (pt_mode[p] != 2) continue;  // clustered points never become terms
      const int a0 = h_pt_start[p], a1 = h_pt_start[p + 1], q0 = q_start[p], q1 = q_start[p + 1];
      for (int a = a0; a < a1; ++a) {
        const int i = h_oimg[a];
        if (!img_active[i]) continue;
        for (int bq = a0; bq < a1; ++bq) {
          const int j = h_oimg[bq];
          if (img_active[j] && i >= j) f(BLK_PP, i, j, a, bq);
        }
      }
      for (int q = q0; q < q1; ++q) {
        for (int a = a0; a < a1; ++a)
          if (img_active[h_oimg[a]]) f(BLK_IP, q_cam[q], h_oimg[a], q, a);
        for (int q2 = q0; q2 <= q; ++q2) f(BLK_II, q_cam[q], q_cam[q2], q, q2);
      }
    }
  };
  const long long ncols[3] = {NI, NI, NC};
  const long long nrows[3] = {NI, NC, NC};
  size_t nkeys[3], nkeys_tot = 0;
  for (int k = 0; k < 3; ++k) { nkeys[k] = (size_t)(nrows[k] * ncols[k]); nkeys_tot += nkeys[k]; }
  // Host threads own contiguous point ranges (balanced by observations). Per-thread counts turn
  // into per-thread cursors, so the term order inside a block (by point) does not depend on the
  // number of threads: the device sums stay bit-reproducible.
  int T = host_threads();
  if (N < 50000) T = 1;
  while (T > 1 && (size_t)T * nkeys_tot > (size_t)48 << 20) T /= 2;
  std::vector<int> range(T + 1, NP);
  range[0] = 0;
  for (int t = 1; t < T; ++t) {
    const long long target = (long long)N * t / T;
    range[t] = (int)(std::upper_bound(h_pt_start.begin(), h_pt_start.end(), (int)target) - h_pt_start.begin()) - 1;
    range[t] = std::max(range[t - 1], std::min(range[t], NP));
  }
  std::vector<std::vector<int>> tcount(T * 3);
  auto run_threads = [&](const std::function<void(int)>& body) { host_run(T, body); };
  run_threads([&](int t) {
    for (int k = 0; k < 3; ++k) tcount[t * 3 + k].assign(nkeys[k], 0);
    enumerate(range[t], range[t + 1], [&](int kind, int r, int c, int, int) { tcount[t * 3 + kind][(size_t)r * ncols[kind] + c]++; });
  });
  std::vector<int> count[3];
  std::vector<unsigned char> mandatory[3];
  long long tot[3] = {0, 0, 0};
  for (int k = 0; k < 3; ++k) {
    count[k].assign(nkeys[k], 0);
    mandatory[k].assign(nkeys[k], 0);
    for (int t = 0; t < T; ++t)
      for (size_t key = 0; key < nkeys[k]; ++key) count[k][key] += tcount[t * 3 + k][key];
    for (size_t key = 0; key < nkeys[k]; ++key) tot[k] += count[k][key];
  }
  for (int i = 0; i < NI; ++i) {
    if (!img_active[i]) continue;
    mandatory[BLK_PP][(size_t)i * NI + i] = 1;
    if (cam_active[h_img_cam[i]]) mandatory[BLK_IP][(size_t)h_img_cam[i] * NI + i] = 1;
  }
  for (int c = 0; c < NC; ++c) if (cam_active[c]) mandatory[BLK_II][(size_t)c * NC + c] = 1;
  // blocks a cluster touches get one partial slot per cluster
  std::vector<int> cref[3];
  for (int k = 0; k < 3; ++k) cref[k].assign(nkeys[k], 0);
  auto cluster_key = [&](long long cl, int slot, int& kind) -> size_t {
    const int* imgs = &cl_imgs[(size_t)cl * kClImages];
    const int* cams = &cl_cams[(size_t)cl * kClCams];
    if (slot < kClTabIP) {
      int la = 0;
      while ((la + 1) * (la + 2) / 2 <= slot) ++la;
      const int lb = slot - la * (la + 1) / 2;
      kind = BLK_PP;
      return (size_t)imgs[la] * NI + imgs[lb];
    }
    if (slot < kClTabII) {
      const int lc = (slot - kClTabIP) / kClImages, la = (slot - kClTabIP) % kClImages;
      kind = BLK_IP;
      return (size_t)cams[lc] * NI + imgs[la];
    }
    int lc = 0;
    const int sl = slot - kClTabII;
    while ((lc + 1) * (lc + 2) / 2 <= sl) ++lc;
    kind = BLK_II;
    return (size_t)cams[lc] * NC + cams[sl - lc * (lc + 1) / 2];
  };
  cluster_partials = 0;
  for (long long cl = 0; cl < num_clusters; ++cl)
    for (int sl = 0; sl < kClTab; ++sl)
      if (cl_present[(size_t)cl * kClTab + sl]) { int kind; const size_t key = cluster_key(cl, sl, kind); cref[kind][key]++; ++cluster_partials; }
  lap("count terms");
  for (int k = 0; k < 3; ++k)
    if (tot[k] >= (1ll << 31) - 1) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "more than 2^31 Schur terms of one kind");
  // One wave per chunk. A block gets ceil(terms / 1024) chunks but never more than 256, so the
  // finalize pass (which adds a block's chunk partials in order) stays short even for the
  // intrinsics-intrinsics block, whose term list has one entry per point.
  auto block_chunk_terms = [](int cnt) {
    int nch = (cnt + 1023) / 1024;
    nch = std::max(1, std::min(nch, 256));
    return std::max(1, (cnt + nch - 1) / nch);
  };
  std::vector<SchurBlock> blocks;
  std::vector<SchurChunk> chunks[3];
  std::vector<int> cursor[3];
  // Block order = launch order of the chunk kernels. Pose-pose blocks are visited in 2-D tiles of
  // kTile x kTile images so that the entry records of ~2*kTile images (a few MB) stay in one XCD's
  // L2 while all blocks among them are accumulated; the other kinds are ordered by image.
  int kTile = 4;
  if (const char* e = std::getenv("MAVBA_PP_TILE")) kTile = std::max(1, std::atoi(e));  // tuning knob
  for (int k = 0; k < 3; ++k) {
    cursor[k].assign(count[k].size(), 0);
    std::vector<size_t> keys;
    for (size_t key = 0; key < count[k].size(); ++key)
      if (count[k][key] != 0 || mandatory[k][key] || cref[k][key] != 0) keys.push_back(key);
    if (k == BLK_PP) {
      const long long nc = ncols[k];
      std::stable_sort(keys.begin(), keys.end(), [&](size_t a, size_t b) {
        const long long ia = a / nc, ja = a % nc, ib = b / nc, jb = b % nc;
        if (ia / kTile != ib / kTile) return ia / kTile < ib / kTile;
        if (ja / kTile != jb / kTile) return ja / kTile < jb / kTile;
        return a < b;
      });
    } else if (k == BLK_IP) {
      const long long nc = ncols[k];
      std::stable_sort(keys.begin(), keys.end(), [&](size_t a, size_t b) { return a % nc < b % nc; });
    }
    int off = 0, slot = 0;
    for (size_t key : keys) {
      const int cnt = count[k][key];
      SchurBlock B;
      B.kind = k; B.row_ent = (int)(key / ncols[k]); B.col_ent = (int)(key % ncols[k]);
      // the block's partials: its term-list chunks, then one slot per cluster that touches it
      B.chunk_begin = slot;
      const int ct = block_chunk_terms(cnt);
      for (int b0 = off; b0 < off + cnt; b0 += ct)
        chunks[k].push_back(SchurChunk{b0, std::min(b0 + ct, off + cnt), slot++});
      const int nref = cref[k][key];
      cref[k][key] = slot;  // from here on: the next free cluster slot of this block
      slot += nref;
      B.chunk_end = slot;
      blocks.push_back(B);
      cursor[k][key] = off;
      off += cnt;
    }
    num_slots[k] = slot;
  }
  // long partial runs are pre-reduced in groups of 32 into extra slots; the block then points at those
  std::vector<PartialReduce> reduce_tasks;
  for (SchurBlock& B : blocks) {
    const int n = B.chunk_end - B.chunk_begin;
    if (n <= 64) continue;
    const int first = num_slots[B.kind];
    for (int b0 = B.chunk_begin; b0 < B.chunk_end; b0 += 32)
      reduce_tasks.push_back(PartialReduce{B.kind, b0, std::min(b0 + 32, B.chunk_end), num_slots[B.kind]++});
    B.chunk_begin = first; B.chunk_end = num_slots[B.kind];
  }
  num_reduce_tasks = (int)reduce_tasks.size();
  d_reduce_tasks.upload(reduce_tasks, st);
  // slot tables of the clusters (clusters in order -> a block's partials are added in a fixed order)
  std::vector<int> cl_tab((size_t)std::max(num_clusters, 1) * kClTab, -1);
  for (long long cl = 0; cl < num_clusters; ++cl)
    for (int sl = 0; sl < kClTab; ++sl)
      if (cl_present[(size_t)cl * kClTab + sl]) { int kind; const size_t key = cluster_key(cl, sl, kind); cl_tab[(size_t)cl * kClTab + sl] = cref[kind][key]++; }
  lap("order blocks + chunks");
  std::unique_ptr<int2[]> terms[3];  // uninitialised on purpose: first touched by the filling threads
  for (int k = 0; k < 3; ++k) terms[k].reset(new int2[std::max<size_t>((size_t)tot[k], 1)]);
  // per-thread cursors: block offset + what the threads owning earlier points put into the block
  for (int k = 0; k < 3; ++k)
    for (size_t key = 0; key < nkeys[k]; ++key) {
      int run = cursor[k][key];
      for (int t = 0; t < T; ++t) { const int c = tcount[t * 3 + k][key]; tcount[t * 3 + k][key] = run; run += c; }
    }
  run_threads([&](int t) {
    enumerate(range[t], range[t + 1], [&](int kind, int r, int c, int x, int y) {
      terms[kind][(size_t)tcount[t * 3 + kind][(size_t)r * ncols[kind] + c]++] = make_int2(x, y);
    });
  });
  lap("fill terms");
  choose_elimination_order(blocks);
  lap("elimination order");
  num_blocks = (int)blocks.size();
  d_blocks.upload(blocks, st);
  for (int k = 0; k < 3; ++k) {
    num_chunks[k] = (int)chunks[k].size();
    num_terms[k] = tot[k];
    d_chunks[k].upload(chunks[k], st);
    d_terms[k].alloc(std::max<size_t>((size_t)tot[k], 1));
    if (tot[k]) HIP_OK(hipMemcpyAsync(d_terms[k].p, terms[k].get(), (size_t)tot[k] * sizeof(int2), hipMemcpyHostToDevice, st));
    d_part[k].alloc((size_t)std::max(num_slots[k], 1) * schur_partial_stride(k));
  }
  {
    std::vector<unsigned char> ptc(std::max(NP, 1), 0);
    for (int p = 0; p < NP; ++p) ptc[p] = pt_mode[p] == 1;
    d_clusters.upload(clusters, st); d_cl_tab.upload(cl_tab, st);
    d_obs_meta.upload(obs_meta, st); d_q_meta.upload(q_meta, st); d_pt_clustered.upload(ptc, st);
  }
  sync();
  lap("upload terms");
}

void mavba_session::reset_state() {
  HIP_OK(hipMemcpyAsync(d_poses.p, d_poses0.p, (size_t)NI * 6 * 8, hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(d_intr.p, d_intr0.p, (size_t)NC * 9 * 8, hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(d_points.p, d_points0.p, (size_t)NP * 3 * 8, hipMemcpyDeviceToDevice, st));
  evaluated = scales_ready = started = assembled = false;
  camrec_current = false;
  radius = opt.initial_trust_region_radius; decrease_factor = 2.0;
  cost = x_norm = grad_max = abs_gtol = initial_cost = 0.0;
  iteration = invalid_steps = n_success = n_fail = 0;
  termination = MAVBA_TERM_RUNNING;
  solve_seconds = 0.0;
}

// ===========================================================================
// Evaluation at the current x: residuals, Jacobian, cost, gradient norm (ceres
// Evaluator::Evaluate with jacobian != NULL).
// ===========================================================================
void mavba_session::evaluate_enqueue() {
  if (!camrec_current) timed("cam_prepare", [&] { launch_cam_prepare(st, NI, d_poses.p, d_camrec.p); });
  camrec_current = true;
  SweepArgs a = sweep_args(d_camrec.p, d_intr.p, d_points.p);
  timed("jacobian_sweep", [&] { launch_jacobian_sweep(st, a); });
  timed("point_reduce", [&] {
    launch_point_reduce(st, NP, NPs, Nstride, KMAX, d_pt_start.p, d_q_start.p, d_q_cam.p, d_obs_img.p, d_img_cam.p,
                        d_R.p, d_Jp.p, d_Jk.p, d_Cu.p, d_gu.p, d_Wk.p);
  });
  CamSweepArgs c;
  c.NI = NI; c.NC = NC; c.chunks = d_sweep_chunks.p; c.num_chunks = num_sweep_chunks;
  c.im_uv = d_im_uv.p; c.im_pt = d_im_pt.p; c.camrec = d_camrec.p; c.intr = d_intr.p;
  c.img_cam = d_img_cam.p; c.cam_model = d_cam_model.p; c.points = d_points.p;
  c.loss_b = a.loss_b; c.loss_inv_b = a.loss_inv_b; c.partial = d_cam_partial.p;
  timed("camera_sweep", [&] { launch_camera_sweep(st, c, KMAX, any_intr_free); });
  if (num_priors > 0)
    timed("rot_prior", [&] {
      launch_rot_prior(st, num_priors, d_prior_img.p, d_prior_R0.p, prior_weight, d_poses.p, d_prior_res.p,
                       d_prior_jac.p, d_prior_cost.p);
    });
  timed("camera_reduce", [&] {
    launch_camera_reduce(st, NI, NC, d_img_chunk_start.p, d_cam_partial.p, num_priors > 0 ? d_prior_start.p : nullptr,
                         d_prior_res.p, d_prior_jac.p, d_cam_img_start.p, d_cam_imgs.p, d_img_rec, d_cam_rec,
                         d_img_intr_tmp.p);
  });
  allreduce(d_camsum.p, (long long)NI * kImgRec + (long long)NC * kCamRec, 0);
  if (!scales_ready) {
    timed("scales", [&] {
      launch_scales(st, NI, NC, NP, NPs, opt.jacobi_scaling, d_pose_free.p, d_intr_free.p, d_pt_free.p, d_img_rec,
                    d_cam_rec, d_Cu.p, d_scale_cam.p, d_scale_pt.p);
    });
    scales_ready = true;
  }
  int rows = 0;
  timed("state_norms", [&] {
    launch_state_norms(st, NI, NC, NP, NPs, rank == 0, d_pose_free.p, d_intr_free.p, d_pt_free.p, d_poses.p,
                       d_intr.p, d_points.p, d_img_rec, d_cam_rec, d_gu.p, d_norm_partial.p, &rows);
  });
  timed("reduce", [&] {
    const int nsweep = N > 0 ? jacobian_sweep_grid(N) : 0;
    ReduceTasks T;
    T.t[0] = ReduceTask{d_norm_partial.p, rows, 2, 1, nullptr, 0, d_scal.p + SC_GRAD_MAX};
    T.t[1] = ReduceTask{d_norm_partial.p + 1, rows, 2, 0, nullptr, 0, d_scal.p + SC_XNORM2};
    T.t[2] = ReduceTask{d_sweep_partial.p, nsweep, 1, 0, d_prior_cost.p, num_priors, d_scal.p + SC_COST};
    launch_reduce_tasks(st, T, 3);
  });
  if (world > 1) allreduce(d_scal.p, SC_NUM_SUMS + 1, 2);  // sums, then max|g| in the last slot
  evaluated = true; assembled = false;
}
void mavba_session::evaluate() {
  evaluate_enqueue();
  double h[SC_COUNT];
  read_scalars(h);
  take_evaluation(h);
}

// Schur complement for the current Jacobian at trust-region radius r:
// rows [0, n_pad) of d_M <- S, row n_pad <- v   (SchurEliminator::Eliminate).
void mavba_session::assemble(double r) {
  const double dmin = opt.min_lm_diagonal, dmax = opt.max_lm_diagonal;
  HIP_OK(hipMemsetAsync(d_scal.p + SC_FAIL, 0, sizeof(double), st));
  timed("point_factor", [&] {
    launch_point_factor(st, NP, NPs, r, dmin, dmax, d_pt_free.p, d_Cu.p, d_gu.p, d_scale_pt.p, d_Gi.p, d_h.p,
                        d_scal.p + SC_FAIL);
  });
  timed("entries_pose", [&] {
    launch_entries_pose(st, N, Nstride, NPs, d_obs_img.p, d_obs_pt.p, d_pt_free.p, d_Jc.p, d_Jp.p, d_scale_cam.p,
                        d_scale_pt.p, d_Gi.p, d_h.p, d_Epose.p);
  });
  timed("entries_intr", [&] {
    launch_entries_intr(st, Q, NI, NPs, d_q_pt.p, d_q_cam.p, d_Wk.p, d_scale_cam.p, d_scale_pt.p, d_Gi.p, d_h.p,
                        d_Eintr.p);
  });
  timed("memset_S", [&] { HIP_OK(hipMemsetAsync(d_M.p, 0, (size_t)(n_mat + 64) * n_mat * sizeof(double), st)); });
  timed("schur_clusters", [&] {
    launch_schur_clusters(st, cl_shape, num_clusters, d_clusters.p, d_cl_tab.p, d_pt_start.p, d_q_start.p, d_obs_meta.p,
                          d_q_meta.p, d_pt_clustered.p, d_Epose.p, d_Eintr.p, d_h.p, NPs, d_part[0].p, d_part[1].p,
                          d_part[2].p);
  });
  if (num_chunks[0] > 0) timed("schur_chunks_pp", [&] { launch_schur_chunks(st, BLK_PP, num_chunks[0], d_chunks[0].p, d_terms[0].p, d_Epose.p, d_Eintr.p, d_part[0].p); });
  if (num_chunks[1] > 0) timed("schur_chunks_ip", [&] { launch_schur_chunks(st, BLK_IP, num_chunks[1], d_chunks[1].p, d_terms[1].p, d_Epose.p, d_Eintr.p, d_part[1].p); });
  if (num_chunks[2] > 0) timed("schur_chunks_ii", [&] { launch_schur_chunks(st, BLK_II, num_chunks[2], d_chunks[2].p, d_terms[2].p, d_Epose.p, d_Eintr.p, d_part[2].p); });
  double* v = d_M.p + (size_t)n_mat * n_mat;
  timed("schur_finalize", [&] {
    launch_partial_reduce(st, num_reduce_tasks, d_reduce_tasks.p, d_part[0].p, d_part[1].p, d_part[2].p);
    launch_schur_finalize(st, num_blocks, d_blocks.p, d_part[0].p, d_part[1].p, d_part[2].p, NI, NC, n_mat, rank == 0,
                          r, dmin, dmax, d_img_cam.p, d_img_rec, d_cam_rec, d_scale_cam.p, d_off.p, d_off.p + NI, d_M.p, v);
    launch_fix_diag(st, n_mat, n_mat, rank == 0, d_col_var.p, d_scale_cam.p, d_M.p);
  });
  if (world > 1 && ar_fn) {
    // only the tiles the factorisation reads (lower, inside the structure) and the right-hand side travel
    double* rhs = d_ar_buf.p + (size_t)num_ar_tiles * 4096;
    launch_tiles_copy(st, num_ar_tiles, d_ar_tiles.p, d_M.p, n_mat, d_ar_buf.p, true);
    HIP_OK(hipMemcpyAsync(rhs, v, (size_t)n_mat * 8, hipMemcpyDeviceToDevice, st));
    allreduce(d_ar_buf.p, (long long)num_ar_tiles * 4096 + n_mat, 0);
    launch_tiles_copy(st, num_ar_tiles, d_ar_tiles.p, d_M.p, n_mat, d_ar_buf.p, false);
    HIP_OK(hipMemcpyAsync(v, rhs, (size_t)n_mat * 8, hipMemcpyDeviceToDevice, st));
  }
  assembled = true;
}

void mavba_session::solve_linear(double r) {
  assemble(r);
  timed("dense_cholesky", [&] { dense_spd_solve_device(st, d_M.p, n_mat, d_ymat.p, d_scal.p + SC_FAIL, d_diag_ws.p, d_L.p, chol_struct, d_col_var.p, d_y.p); });
  assembled = false;  // the factorisation overwrote S
}

// Back-substitution, candidate x + delta, and its cost. Leaves the scalars on the host.
void mavba_session::candidate(double r, double* h) {
  const double dmin = opt.min_lm_diagonal, dmax = opt.max_lm_diagonal;
  int rows = 0;
  timed("backsub_points", [&] {
    launch_backsub_points(st, NP, NPs, NI, r, dmin, dmax, d_pt_start.p, d_obs_img.p, d_q_start.p, d_q_cam.p,
                          d_pt_free.p, d_Epose.p, d_Eintr.p, d_y.p, d_Gi.p, d_h.p, d_Cu.p, d_gu.p, d_scale_pt.p,
                          d_points.p, d_cpoints.p, d_delta_pts.p, d_step_partial.p, &rows);
  });
  timed("update_cameras", [&] {
    launch_update_cameras(st, NI, NC, rank == 0, r, dmin, dmax, d_y.p, d_scale_cam.p, d_img_rec, d_cam_rec,
                          d_poses.p, d_intr.p, d_cposes.p, d_cintr.p, d_delta_cam.p, d_step_partial.p + 3 * (size_t)rows);
  });
  timed("cam_prepare", [&] { launch_cam_prepare(st, NI, d_cposes.p, d_ccamrec.p); });
  SweepArgs a = sweep_args(d_ccamrec.p, d_cintr.p, d_cpoints.p);
  timed("cost_only", [&] { launch_cost_only(st, a); });
  if (num_priors > 0)
    timed("rot_prior", [&] {
      launch_rot_prior(st, num_priors, d_prior_img.p, d_prior_R0.p, prior_weight, d_cposes.p, d_prior_res.p,
                       d_prior_jac.p, d_prior_cost.p);
    });
  timed("reduce", [&] {
    const int nsweep = N > 0 ? jacobian_sweep_grid(N) : 0;
    ReduceTasks T;
    T.t[0] = ReduceTask{d_step_partial.p, rows + 1, 3, 0, nullptr, 0, d_scal.p + SC_STEP_NORM2};
    T.t[1] = ReduceTask{d_step_partial.p + 1, rows + 1, 3, 0, nullptr, 0, d_scal.p + SC_MODEL_CHANGE};
    T.t[2] = ReduceTask{d_step_partial.p + 2, rows + 1, 3, 0, nullptr, 0, d_scal.p + SC_CAND_XNORM2};
    T.t[3] = ReduceTask{d_sweep_partial.p, nsweep, 1, 0, d_prior_cost.p, num_priors, d_scal.p + SC_NEW_COST};
    launch_reduce_tasks(st, T, 4);
  });
  if (world > 1) allreduce(d_scal.p, SC_NUM_SUMS, 0);
  read_scalars(h);
}

void mavba_session::start() {
  evaluate();
  initial_cost = cost + fixed_cost;
  const double g0 = std::max(grad_max, std::numeric_limits<double>::epsilon());
  abs_gtol = opt.gradient_tolerance * g0;
  started = true;
  if (num_residuals_reduced == 0 || num_parameters_reduced == 0) { termination = MAVBA_TERM_FUNCTION_TOLERANCE; return; }
  if (grad_max <= abs_gtol) { termination = MAVBA_TERM_GRADIENT_TOLERANCE; return; }
  if (opt.print_progress) {
    std::printf("%4s %14s %12s %10s %10s %10s %10s\n", "iter", "cost", "cost_change", "|gradient|", "|step|", "tr_ratio", "tr_radius");
    std::printf("%4d %14.6e %12.2e %10.2e %10.2e %10.2e %10.2e\n", 0, cost + fixed_cost, 0.0, grad_max, 0.0, 0.0, radius);
  }
}

// TrustRegionMinimizer::Minimize main loop (Ceres 1.8), one pass per LM iteration.
int mavba_session::iterate(int max_iters, int* done) {
  const double t0 = now_s();
  int n = 0;
  if (!started) start();
  // Single process: the scalars of the evaluation at an accepted point are read back together with those of
  // the NEXT candidate (one host synchronisation per iteration instead of two): the next linear solve is
  // enqueued right behind the evaluation, and the tests that follow an evaluation in Ceres' loop (gradient
  // tolerance) are applied when its scalars arrive - before anything of the speculative iteration counts.
  const bool defer = world == 1 && !opt.print_progress;
  bool pending_eval = false;
  while (termination == MAVBA_TERM_RUNNING && n < max_iters) {
    if (iteration >= opt.max_num_iterations) { termination = MAVBA_TERM_NO_CONVERGENCE; break; }
    ++iteration; ++n;
    solve_linear(radius);
    double h[SC_COUNT];
    candidate(radius, h);
    if (pending_eval) {
      pending_eval = false;
      take_evaluation(h);
      if (grad_max <= abs_gtol) {  // the previous iteration ended the solve: this one never happened
        termination = MAVBA_TERM_GRADIENT_TOLERANCE;
        --iteration; --n;
        break;
      }
    }
    const double mcc = h[SC_MODEL_CHANGE];
    const bool solved = h[SC_FAIL] == 0.0 && std::isfinite(mcc) && std::isfinite(h[SC_STEP_NORM2]);
    const bool valid = solved && !(mcc < 0.0);
    bool successful = false;
    double rel = 0.0, step_norm = 0.0, cost_change = 0.0;
    if (!valid) {
      if (++invalid_steps >= opt.max_num_consecutive_invalid_steps) { termination = MAVBA_TERM_NUMERICAL_FAILURE; ++n_fail; break; }
    } else {
      invalid_steps = 0;
      step_norm = std::sqrt(h[SC_STEP_NORM2]);
      const double new_cost = h[SC_NEW_COST];
      if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) { termination = MAVBA_TERM_PARAMETER_TOLERANCE; break; }
      cost_change = cost - new_cost;
      if (std::fabs(cost_change) < opt.function_tolerance * cost) { termination = MAVBA_TERM_FUNCTION_TOLERANCE; break; }
      rel = cost_change / mcc;
      successful = rel > opt.min_relative_decrease;
    }
    if (successful) {
      ++n_success;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
      radius = std::min(opt.max_trust_region_radius, radius);
      decrease_factor = 2.0;
      std::swap(d_poses.p, d_cposes.p); std::swap(d_intr.p, d_cintr.p); std::swap(d_points.p, d_cpoints.p);
      std::swap(d_camrec.p, d_ccamrec.p);  // the candidate's camera records are the new point's (camrec_current stays true)
      if (defer) {
        evaluate_enqueue();
        pending_eval = true;
      } else {
        evaluate();
        if (grad_max <= abs_gtol) termination = MAVBA_TERM_GRADIENT_TOLERANCE;
      }
    } else {
      ++n_fail;
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
    }
    if (opt.print_progress)
      std::printf("%4d %14.6e %12.2e %10.2e %10.2e %10.2e %10.2e\n", iteration, cost + fixed_cost, successful ? cost_change : 0.0,
                  grad_max, step_norm, rel, radius);
    if (termination == MAVBA_TERM_RUNNING && radius < opt.min_trust_region_radius) termination = MAVBA_TERM_PARAMETER_TOLERANCE;
  }
  if (pending_eval) {
    // the evaluation's own test comes before whatever ended the loop after it was enqueued
    double h[SC_COUNT];
    read_scalars(h);
    take_evaluation(h);
    if (grad_max <= abs_gtol) termination = MAVBA_TERM_GRADIENT_TOLERANCE;
  }
  if (termination == MAVBA_TERM_RUNNING && iteration >= opt.max_num_iterations) termination = MAVBA_TERM_NO_CONVERGENCE;
  if (done) *done = n;
  solve_seconds += now_s() - t0;
  return MAVBA_OK;
}

void mavba_session::point_errors(double* out) {
  launch_cam_prepare(st, NI, d_poses.p, d_camrec.p);
  SweepArgs a = sweep_args(d_camrec.p, d_intr.p, d_points.p);
  launch_raw_residual_norm(st, a, d_rnorm.p);
  launch_point_errors(st, NP, d_pt_start.p, d_rnorm.p, d_pt_count.p, d_perr.p);
  std::vector<double> h(NP);
  if (NP) HIP_OK(hipMemcpyAsync(h.data(), d_perr.p, (size_t)NP * 8, hipMemcpyDeviceToHost, st));
  sync();
  evaluated = false;  // camrec still matches x, but keep the contract simple
  // Only points that have observations in the problem are touched (bundle_adjustment.cc:578-581);
  // observations dropped as all-constant blocks still count (they are residual blocks there).
  for (int p = 0; p < NP; ++p)
    if (h_pt_count_all[p] > 0) out[h_pt_orig[p]] = h[p];
}

void mavba_session::fill_result(mavba_result* r) {
  std::memset(r, 0, sizeof(*r));
  r->initial_cost = initial_cost;
  r->final_cost = cost + fixed_cost;
  r->fixed_cost = fixed_cost;
  r->num_residuals = num_residuals;
  r->num_residuals_reduced = num_residuals_reduced;
  r->num_parameters_reduced = num_parameters_reduced;
  r->num_successful_steps = n_success;
  r->num_unsuccessful_steps = n_fail;
  r->termination = termination;
  r->final_gradient_max_norm = grad_max;
  r->final_trust_region_radius = radius;
  r->setup_seconds = setup_seconds;
  r->solve_seconds = solve_seconds;
}

// ===========================================================================
// C ABI
// ===========================================================================
#define MAVBA_TRY try {
#define MAVBA_CATCH                                                              \
  }                                                                              \
  catch (const Failure& f) { g_last_error = f.what(); return f.code; }           \
  catch (const std::bad_alloc&) { g_last_error = "host out of memory"; return MAVBA_ERR_OUT_OF_MEMORY; } \
  catch (const std::exception& e) { g_last_error = e.what(); return MAVBA_ERR_HIP; }

extern "C" {

void mavba_options_init(mavba_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 100;       // bundle_adjustment.h:40
  o->function_tolerance = 1e-4;      // :41
  o->gradient_tolerance = 1e-8;      // :42
  o->loss_scale_factor = 1.0;        // :45
  o->update_point_errors = 0;        // :43
  o->print_progress = 0;             // :49
  o->parameter_tolerance = 1e-8;     // Ceres 1.8 Solver::Options defaults below
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 10;  // bundle_adjustment.cc:559
  o->jacobi_scaling = 1;
  o->device = -1;
  o->profile_kernels = 0;
}

const char* mavba_last_error(void) { return g_last_error.c_str(); }

int mavba_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int mavba_session_create(const mavba_problem* problem, const mavba_options* options, mavba_session** out) {
  if (!problem || !options || !out) { g_last_error = "null argument"; return MAVBA_ERR_INVALID_ARGUMENT; }
  *out = nullptr;
  if (mavba_device_count() <= 0) {
    g_last_error = "no HIP device: the mavba backend has no CPU path";
    return MAVBA_ERR_NO_DEVICE;
  }
  mavba_session* s = nullptr;
  MAVBA_TRY
  s = new mavba_session();
  s->opt = *options;
  if (options->device >= 0) HIP_OK(hipSetDevice(options->device));
  HIP_OK(hipGetDevice(&s->device));
  HIP_OK(stream_acquire(&s->st));
  s->build(problem);
  *out = s;
  return MAVBA_OK;
  }
  catch (const Failure& f) { g_last_error = f.what(); delete s; return f.code; }
  catch (const std::bad_alloc&) { g_last_error = "host out of memory"; delete s; return MAVBA_ERR_OUT_OF_MEMORY; }
  catch (const std::exception& e) { g_last_error = e.what(); delete s; return MAVBA_ERR_HIP; }
}

void mavba_session_destroy(mavba_session* s) { delete s; }

int mavba_session_reset(mavba_session* s) {
  MAVBA_TRY
  s->reset_state();
  s->sync();
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_iterate(mavba_session* s, int32_t max_iters, int32_t* iters_done, int32_t* termination) {
  MAVBA_TRY
  int done = 0;
  s->iterate(max_iters, &done);
  if (iters_done) *iters_done = done;
  if (termination) *termination = s->termination;
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_result(mavba_session* s, mavba_result* result) {
  MAVBA_TRY
  s->fill_result(result);
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_get_params(mavba_session* s, double* poses, double* intrinsics, double* points) {
  MAVBA_TRY
  if (poses && s->NI) HIP_OK(hipMemcpyAsync(poses, s->d_poses.p, (size_t)s->NI * 48, hipMemcpyDeviceToHost, s->st));
  if (intrinsics && s->NC) HIP_OK(hipMemcpyAsync(intrinsics, s->d_intr.p, (size_t)s->NC * 72, hipMemcpyDeviceToHost, s->st));
  std::vector<double> hp;
  if (points && s->NP) { hp.resize((size_t)s->NP * 3); HIP_OK(hipMemcpyAsync(hp.data(), s->d_points.p, (size_t)s->NP * 24, hipMemcpyDeviceToHost, s->st)); }
  s->sync();
  if (points) s->to_caller_points(hp.data(), points, 3);
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_point_errors(mavba_session* s, double* point_error) {
  MAVBA_TRY
  if (!point_error) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null point_error");
  s->point_errors(point_error);
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_set_allreduce(mavba_session* s, mavba_allreduce_fn fn, void* ctx, int32_t rank, int32_t world_size) {
  MAVBA_TRY
  if (s->started) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "set_allreduce must precede the first iteration");
  s->ar_fn = fn; s->ar_ctx = ctx; s->rank = rank; s->world = world_size;
  if (fn && world_size > 1) {
    // A camera block is in the problem if ANY rank has a residual block on it; the counts
    // reported in mavba_result become global.
    const size_t n = (size_t)s->NI + s->NC + 4;
    std::vector<double> h(n, 0.0);
    for (int i = 0; i < s->NI; ++i) h[i] = s->h_img_used[i];
    for (int c = 0; c < s->NC; ++c) h[s->NI + c] = s->h_cam_used[c];
    DevBuf<double> d;
    d.upload(h, s->st);
    s->allreduce(d.p, (long long)s->NI + s->NC, 1);
    std::vector<double> g(4, 0.0);
    long long free_pts = 0;
    for (unsigned char f : s->h_pt_free) free_pts += f;
    g[0] = s->fixed_cost; g[1] = (double)s->num_residuals; g[2] = (double)s->num_residuals_reduced; g[3] = (double)free_pts;
    HIP_OK(hipMemcpyAsync(d.p + s->NI + s->NC, g.data(), 32, hipMemcpyHostToDevice, s->st));
    s->allreduce(d.p + s->NI + s->NC, 4, 0);
    HIP_OK(hipMemcpyAsync(h.data(), d.p, n * 8, hipMemcpyDeviceToHost, s->st));
    s->sync();
    for (int i = 0; i < s->NI; ++i) s->h_img_used[i] = h[i] != 0.0;
    for (int c = 0; c < s->NC; ++c) s->h_cam_used[c] = h[s->NI + c] != 0.0;
    s->derive_free_flags();
    long long cam_params = 0;
    for (unsigned char f : s->h_pose_free) cam_params += f;
    for (unsigned char f : s->h_intr_free) cam_params += f;
    s->fixed_cost = h[s->NI + s->NC];
    s->num_residuals = (long long)h[s->NI + s->NC + 1];
    s->num_residuals_reduced = (long long)h[s->NI + s->NC + 2];
    s->num_parameters_reduced = cam_params + 3 * (long long)h[s->NI + s->NC + 3];
    s->finish_structure();
  }
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_eval_jacobian(mavba_session* s, double* cost, double* r, double* Jc, double* Jp, double* Jk) {
  MAVBA_TRY
  s->evaluate();
  if (cost) *cost = s->cost + s->fixed_cost;
  const size_t S = s->Nstride, N = s->N;
  auto pull = [&](const double* dev, int planes, std::vector<double>& h) {
    h.resize((size_t)planes * S);
    HIP_OK(hipMemcpyAsync(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost, s->st));
  };
  std::vector<double> hR, hJp, hJc, hJk;
  pull(s->d_R.p, 2, hR); pull(s->d_Jp.p, 6, hJp); pull(s->d_Jc.p, 12, hJc); pull(s->d_Jk.p, 2 * s->KMAX, hJk);
  s->sync();
  const size_t NO = (size_t)s->NO_all;
  if (r) std::memset(r, 0, NO * 2 * 8);
  if (Jc) std::memset(Jc, 0, NO * 12 * 8);
  if (Jp) std::memset(Jp, 0, NO * 6 * 8);
  if (Jk) std::memset(Jk, 0, NO * 18 * 8);
  for (size_t a = 0; a < N; ++a) {
    const size_t o = (size_t)s->perm[a];
    if (r) for (int e = 0; e < 2; ++e) r[o * 2 + e] = hR[e * S + a];
    if (Jp) for (int e = 0; e < 6; ++e) Jp[o * 6 + e] = hJp[e * S + a];
    if (Jc) for (int e = 0; e < 12; ++e) Jc[o * 12 + e] = hJc[e * S + a];
    if (Jk)
      for (int row = 0; row < 2; ++row)
        for (int k = 0; k < s->KMAX; ++k) Jk[o * 18 + row * 9 + k] = hJk[(size_t)(row * s->KMAX + k) * S + a];
  }
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_reduced_dim(mavba_session* s) { return s ? s->n_full : 0; }

int mavba_session_reduced_system(mavba_session* s, double radius, double* Sout, double* vout) {
  MAVBA_TRY
  if (!s->evaluated) s->evaluate();
  s->assemble(radius);
  // the device matrix is in elimination order; hand it out in the variables' order
  const int n = s->n_full, m = s->n_mat;
  std::vector<double> h((size_t)(m + 1) * m);
  HIP_OK(hipMemcpyAsync(h.data(), s->d_M.p, h.size() * 8, hipMemcpyDeviceToHost, s->st));
  s->sync();
  std::vector<int> var_col(n, 0);
  for (int t = 0; t < m; ++t) if (s->h_col_var[t] >= 0) var_col[s->h_col_var[t]] = t;
  if (Sout)
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) {  // the lower triangle is the one that is factorised (and, with shards, all-reduced)
        const int a = std::max(var_col[r], var_col[c]), b = std::min(var_col[r], var_col[c]);
        Sout[(size_t)r * n + c] = h[(size_t)a * m + b];
      }
  if (vout)
    for (int r = 0; r < n; ++r) vout[r] = h[(size_t)m * m + var_col[r]];
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_linear_step(mavba_session* s, double radius, double* d_poses, double* d_intr, double* d_points,
                              double* model_cost_change) {
  MAVBA_TRY
  if (!s->evaluated) s->evaluate();
  s->solve_linear(radius);
  double h[SC_COUNT];
  s->candidate(radius, h);
  if (model_cost_change) *model_cost_change = h[SC_MODEL_CHANGE];
  if (d_poses && s->NI) HIP_OK(hipMemcpyAsync(d_poses, s->d_delta_cam.p, (size_t)s->NI * 48, hipMemcpyDeviceToHost, s->st));
  if (d_intr && s->NC) HIP_OK(hipMemcpyAsync(d_intr, s->d_delta_cam.p + 6 * (size_t)s->NI, (size_t)s->NC * 72, hipMemcpyDeviceToHost, s->st));
  std::vector<double> hdp;
  if (d_points && s->NP) { hdp.resize((size_t)s->NP * 3); HIP_OK(hipMemcpyAsync(hdp.data(), s->d_delta_pts.p, (size_t)s->NP * 24, hipMemcpyDeviceToHost, s->st)); }
  s->sync();
  if (d_points) s->to_caller_points(hdp.data(), d_points, 3);
  if (h[SC_FAIL] != 0.0) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "linear solve failed (matrix not positive definite)");
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_time_jacobian(mavba_session* s, int32_t reps, float* ms_avg) {
  MAVBA_TRY
  if (reps < 1) reps = 1;
  launch_cam_prepare(s->st, s->NI, s->d_poses.p, s->d_camrec.p);
  SweepArgs a = s->sweep_args(s->d_camrec.p, s->d_intr.p, s->d_points.p);
  launch_jacobian_sweep(s->st, a);  // warm-up
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, s->st));
  for (int i = 0; i < reps; ++i) launch_jacobian_sweep(s->st, a);
  HIP_OK(hipEventRecord(e1, s->st));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (ms_avg) *ms_avg = ms / reps;
  s->sync();
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_get_info(mavba_session* s, mavba_session_info* out) {
  MAVBA_TRY
  if (!s || !out) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  std::memset(out, 0, sizeof(*out));
  out->num_obs_kept = s->N;
  out->reduced_dim = s->n_full; out->padded_dim = s->n_pad;
  for (int k = 0; k < 3; ++k) out->schur_terms[k] = s->num_terms[k];
  out->schur_blocks = s->num_blocks; out->intr_entries = s->Q;
  const CholStructure& cs = s->chol_struct;
  const long long nb = cs.nb;
  out->dense_tiles = nb * (nb + 1) / 2;
  out->envelope_tiles = cs.envelope_tiles;
  out->factor_flops = cs.factor_flops;
  out->matrix_dim = s->n_mat;
  out->nd_parts = s->nd_parts;
  out->chain_steps = cs.chain_steps;
  out->num_clusters = s->num_clusters;
  out->clustered_points = s->clustered_points;
  out->cluster_partials = s->cluster_partials;
  out->cluster_flops = s->cluster_flops;
  const double n = (double)s->n_full;
  out->dense_factor_flops = n * n * n / 3.0 + 2.0 * n * n;
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_kernel_stats(mavba_session* s, mavba_kernel_stat* out, int32_t cap) {
  if (!s) return 0;
  const int n = (int)s->timers.size();
  for (int i = 0; i < n && i < cap; ++i) {
    std::memset(&out[i], 0, sizeof(out[i]));
    std::strncpy(out[i].name, s->timers[i].name.c_str(), sizeof(out[i].name) - 1);
    out[i].launches = s->timers[i].launches;
    out[i].total_ms = s->timers[i].total_ms;
  }
  return n;
}

int mavba_solve(const mavba_problem* problem, const mavba_options* options, mavba_result* result, double* point_error) {
  mavba_session* s = nullptr;
  int rc = mavba_session_create(problem, options, &s);
  if (rc != MAVBA_OK) return rc;
  int done = 0, term = 0;
  rc = mavba_session_iterate(s, options->max_num_iterations + 1, &done, &term);
  if (rc == MAVBA_OK && result) rc = mavba_session_result(s, result);
  // ceres leaves the user's parameter blocks untouched after NUMERICAL_FAILURE
  if (rc == MAVBA_OK && term != MAVBA_TERM_NUMERICAL_FAILURE)
    rc = mavba_session_get_params(s, problem->poses, problem->intrinsics, problem->points);
  if (rc == MAVBA_OK && point_error && options->update_point_errors) rc = mavba_session_point_errors(s, point_error);
  mavba_session_destroy(s);
  return rc;
}

int mavba_pose_refine(double rvec[3], double tvec[3], const double* intrinsics, int32_t camera_model,
                      const double* uv, const double* xyz, const uint8_t* inlier_mask, int64_t n,
                      const mavba_options* options, mavba_result* result) {
  if (!rvec || !tvec || !intrinsics || !options || n < 0 || (n > 0 && (!uv || !xyz))) {
    g_last_error = "null argument"; return MAVBA_ERR_INVALID_ARGUMENT;
  }
  if (camera_model < 1 || camera_model > 3) { g_last_error = "bad camera model"; return MAVBA_ERR_BAD_MODEL; }
  // One image, one (constant) camera, every inlier a constant point: pose_refinement(),
  // bundle_adjustment.cc:160-193.
  std::vector<double> pose = {rvec[0], rvec[1], rvec[2], tvec[0], tvec[1], tvec[2]};
  std::vector<double> intr(9, 0.0), pts, obs;
  for (int k = 0; k < model_k(camera_model); ++k) intr[k] = intrinsics[k];
  std::vector<int32_t> oi, op;
  for (int64_t i = 0; i < n; ++i) {
    if (inlier_mask && !inlier_mask[i]) continue;
    op.push_back((int32_t)(pts.size() / 3)); oi.push_back(0);
    pts.insert(pts.end(), xyz + 3 * i, xyz + 3 * i + 3);
    obs.insert(obs.end(), uv + 2 * i, uv + 2 * i + 2);
  }
  const int32_t np = (int32_t)(pts.size() / 3);
  std::vector<uint8_t> pconst(std::max(np, 1), 1);
  uint8_t pose_const = 0, intr_const = 1;
  int32_t img_cam = 0, model = camera_model;
  mavba_problem P;
  std::memset(&P, 0, sizeof(P));
  P.num_images = 1; P.num_cameras = 1; P.num_points = np; P.num_obs = np;
  P.poses = pose.data(); P.pose_const = &pose_const; P.image_camera = &img_cam;
  P.intrinsics = intr.data(); P.camera_model = &model; P.intr_const = &intr_const;
  P.points = pts.data(); P.point_const = pconst.data();
  P.obs_uv = obs.data(); P.obs_image = oi.data(); P.obs_point = op.data();
  mavba_result local;
  const int rc = mavba_solve(&P, options, result ? result : &local, nullptr);
  if (rc == MAVBA_OK) { for (int k = 0; k < 3; ++k) { rvec[k] = pose[k]; tvec[k] = pose[3 + k]; } }
  return rc;
}

int mavba_dense_spd_solve(int32_t n, const double* A, const double* b, double* x, int32_t device) {
  if (n <= 0 || !A || !b || !x) { g_last_error = "bad argument"; return MAVBA_ERR_INVALID_ARGUMENT; }
  if (mavba_device_count() <= 0) { g_last_error = "no HIP device: the mavba backend has no CPU path"; return MAVBA_ERR_NO_DEVICE; }
  MAVBA_TRY
  if (device >= 0) HIP_OK(hipSetDevice(device));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  const int n_pad = std::max(64, round_up(n, 64));
  std::vector<double> M((size_t)(n_pad + 64) * n_pad, 0.0);
  for (int i = 0; i < n_pad; ++i) M[(size_t)i * n_pad + i] = 1.0;
  for (int i = 0; i < n; ++i) std::memcpy(&M[(size_t)i * n_pad], &A[(size_t)i * n], (size_t)n * 8);
  std::memcpy(&M[(size_t)n_pad * n_pad], b, (size_t)n * 8);
  int rc = MAVBA_OK;
  {
    DevBuf<double> dM, dL, dy, dws, dfail;
    dM.upload(M, st); dL.alloc(M.size()); dy.alloc(n_pad); dws.alloc((size_t)2 * n_pad * 64); dfail.alloc(1); dfail.zero(st);
    CholStructure cs;
    HIP_OK(cs.build_dense(n_pad / 64));
    dense_spd_solve_device(st, dM.p, n_pad, dy.p, dfail.p, dws.p, dL.p, cs);
    std::vector<double> y(n_pad);
    double fail = 0.0;
    HIP_OK(hipMemcpyAsync(y.data(), dy.p, (size_t)n_pad * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&fail, dfail.p, 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    std::memcpy(x, y.data(), (size_t)n * 8);
    if (fail != 0.0) { g_last_error = "matrix is not positive definite"; rc = MAVBA_ERR_INVALID_ARGUMENT; }
  }
  (void)hipStreamDestroy(st);
  return rc;
  MAVBA_CATCH
}

}  // extern "C"
