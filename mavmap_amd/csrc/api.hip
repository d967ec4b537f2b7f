// api.hip - the C ABI of include/mavba.h over the session object (no exception crosses it).
#include "session.h"
#include "lm_decide.h"

namespace mavba { thread_local std::string g_last_error; }

using namespace mavba;


// ===========================================================================
// C ABI
// ===========================================================================
#define MAVBA_TRY try {
// Entry points that work on a session run on the SESSION's device whatever the calling thread's current device is
// (two sessions on two GPUs driven from one process, or from different threads), and leave the caller's device as it was.
namespace {
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    if (prev != dev) HIP_OK(hipSetDevice(dev)); else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
}  // namespace
#define MAVBA_SESSION_TRY(s)                                                    \
  try {                                                                         \
    if (!(s)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null session");        \
    DeviceGuard device_guard_((s)->device);
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
  int dev = options->device;
  if (dev < 0) HIP_OK(hipGetDevice(&dev));  // the calling thread's current device
  DeviceGuard device_guard_(dev);            // (the caller's current device is restored on return)
  s->device = dev;
  HIP_OK(stream_acquire(&s->st));
  s->build(problem);
  *out = s;
  return MAVBA_OK;
  }
  catch (const Failure& f) { g_last_error = f.what(); delete s; return f.code; }
  catch (const std::bad_alloc&) { g_last_error = "host out of memory"; delete s; return MAVBA_ERR_OUT_OF_MEMORY; }
  catch (const std::exception& e) { g_last_error = e.what(); delete s; return MAVBA_ERR_HIP; }
}

void mavba_session_destroy(mavba_session* s) {
  if (!s) return;
  int prev = -1;
  if (hipGetDevice(&prev) == hipSuccess && prev != s->device) (void)hipSetDevice(s->device); else prev = -1;
  struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
  if (!std::getenv("MAVBA_SETUP_TIMING")) { delete s; return; }
  // phase timing of the tear-down (profiling aid)
  double t0 = now_s();
  auto lap = [&](const char* what) { const double t = now_s(); std::fprintf(stderr, "[destroy] %-26s %8.2f ms\n", what, 1e3 * (t - t0)); t0 = t; };
  if (s->st) (void)hipStreamSynchronize(s->st);
  lap("stream sync");
  s->chol_struct.release();
  lap("factorisation structure");
  delete s;
  lap("rest (device buffers -> pool)");
}

int mavba_session_reset(mavba_session* s) {
  MAVBA_SESSION_TRY(s)
  s->reset_state();
  s->sync();
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_iterate(mavba_session* s, int32_t max_iters, int32_t* iters_done, int32_t* termination) {
  MAVBA_SESSION_TRY(s)
  int done = 0;
  s->iterate(max_iters, &done);
  if (iters_done) *iters_done = done;
  if (termination) *termination = s->termination;
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_result(mavba_session* s, mavba_result* result) {
  MAVBA_SESSION_TRY(s)
  s->fill_result(result);
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_get_params(mavba_session* s, double* poses, double* intrinsics, double* points) {
  MAVBA_SESSION_TRY(s)
  // one pinned staging block: poses | intrinsics | points (already in the caller's order, permuted on the device)
  const size_t nI = (size_t)s->NI * 6, nC = (size_t)s->NC * 9, nP = (size_t)s->NP * 3;
  double* stage = nullptr;
  HIP_OK(pinned_alloc(reinterpret_cast<void**>(&stage), (nI + nC + nP + 1) * 8));
  struct Give { void* p; ~Give() { pinned_free(p); } } give{stage};
  if (poses && nI) HIP_OK(hipMemcpyAsync(stage, s->d_poses.p, nI * 8, hipMemcpyDeviceToHost, s->st));
  if (intrinsics && nC) HIP_OK(hipMemcpyAsync(stage + nI, s->d_intr.p, nC * 8, hipMemcpyDeviceToHost, s->st));
  if (points && nP) {
    launch_points_to_caller(s->st, s->NP, 3, s->d_pt_orig.p, s->d_points.p, s->d_pts_out.p);
    HIP_OK(hipMemcpyAsync(stage + nI + nC, s->d_pts_out.p, nP * 8, hipMemcpyDeviceToHost, s->st));
  }
  s->sync();
  if (poses && nI) std::memcpy(poses, stage, nI * 8);
  if (intrinsics && nC) std::memcpy(intrinsics, stage + nI, nC * 8);
  if (points && nP) std::memcpy(points, stage + nI + nC, nP * 8);
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_point_errors(mavba_session* s, double* point_error) {
  MAVBA_SESSION_TRY(s)
  if (!point_error) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null point_error");
  s->point_errors(point_error);
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_set_params(mavba_session* s, const double* poses, const double* intrinsics, const double* points) {
  MAVBA_SESSION_TRY(s)
  if (poses && s->NI) HIP_OK(copy_h2d_staged(s->d_poses.p, poses, (size_t)s->NI * 48, s->st));
  if (intrinsics && s->NC) HIP_OK(copy_h2d_staged(s->d_intr.p, intrinsics, (size_t)s->NC * 72, s->st));
  std::vector<double> hp;
  if (points && s->NP) {
    hp.resize((size_t)s->NP * 3);
    for (int q = 0; q < s->NP; ++q)
      for (int e = 0; e < 3; ++e) hp[(size_t)q * 3 + e] = points[(size_t)s->h_pt_orig[q] * 3 + e];
    HIP_OK(copy_h2d_staged(s->d_points.p, hp.data(), hp.size() * 8, s->st));
  }
  s->camrec_current = false; s->evaluated = false; s->front_valid = false;
  s->sync();
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_restart(mavba_session* s) {
  MAVBA_SESSION_TRY(s)
  s->restart();
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_filter_points(mavba_session* s, double max_error, const uint8_t* keep, uint8_t* removed, double* errors,
                                int64_t* num_removed) {
  MAVBA_SESSION_TRY(s)
  const long long n = s->filter_points(max_error, keep, removed, errors);
  if (num_removed) *num_removed = n;
  return MAVBA_OK;
  MAVBA_CATCH
}

// After the collective is in place: what every rank must agree on before the first iteration.
static void join_ranks(mavba_session* s) {
  if (!s->sharded()) return;
  // A camera block is in the problem if ANY rank has a residual block on it; the counts
  // reported in mavba_result become global. One more max-reduced flag: "this rank's process is in the cool-down after a
  // persistent-launch time-out" - every rank then starts on the launch-per-panel schedule (the fallback inside
  // linear_step contains a collective, so the ranks must take it together; a sharded session does not count against
  // the cool-down).
  const size_t nmax = (size_t)s->NI + s->NC + 1;
  const size_t n = nmax + 4;
  std::vector<double> h(n, 0.0);
  for (int i = 0; i < s->NI; ++i) h[i] = s->h_img_used[i];
  for (int c = 0; c < s->NC; ++c) h[s->NI + c] = s->h_cam_used[c];
  h[nmax - 1] = persistent_in_cooldown() ? 1.0 : 0.0;
  DevBuf<double> d;
  d.upload(h, s->st);
  s->allreduce(d.p, (long long)nmax, 1);
  std::vector<double> g(4, 0.0);
  long long free_pts = 0;
  for (unsigned char f : s->h_pt_free) free_pts += f;
  g[0] = s->fixed_cost; g[1] = (double)s->num_residuals; g[2] = (double)s->num_residuals_reduced; g[3] = (double)free_pts;
  HIP_OK(hipMemcpyAsync(d.p + nmax, g.data(), 32, hipMemcpyHostToDevice, s->st));
  s->allreduce(d.p + nmax, 4, 0);
  s->download(h.data(), d.p, n * 8);
  for (int i = 0; i < s->NI; ++i) s->h_img_used[i] = h[i] != 0.0;
  for (int c = 0; c < s->NC; ++c) s->h_cam_used[c] = h[s->NI + c] != 0.0;
  s->allow_persistent = h[nmax - 1] == 0.0;
  s->persist_decided = true;
  s->derive_free_flags();
  long long cam_params = 0;
  for (unsigned char f : s->h_pose_free) cam_params += f;
  for (unsigned char f : s->h_intr_free) cam_params += f;
  s->fixed_cost = h[nmax];
  s->num_residuals = (long long)h[nmax + 1];
  s->num_residuals_reduced = (long long)h[nmax + 2];
  s->num_parameters_reduced = cam_params + 3 * (long long)h[nmax + 3];
  s->finish_structure();
}

int mavba_session_set_allreduce(mavba_session* s, mavba_allreduce_fn fn, void* ctx, int32_t rank, int32_t world_size) {
  MAVBA_SESSION_TRY(s)
  if (s->started) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "set_allreduce must precede the first iteration");
  s->ar_fn = fn; s->ar_ctx = ctx; s->rank = rank; s->world = world_size;
  s->force_exchange = world_size == 1 && std::getenv("MAVBA_FORCE_EXCHANGE") != nullptr;
  join_ranks(s);
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_rccl_unique_id(void* out128) {
  MAVBA_TRY
  if (!out128) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  rccl_unique_id(out128);
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_set_rccl(mavba_session* s, const void* unique_id128, int32_t rank, int32_t world_size) {
  MAVBA_SESSION_TRY(s)
  if (!unique_id128) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
  if (s->started) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "set_rccl must precede the first iteration");
  if (rank < 0 || rank >= world_size) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "rank out of range");
  s->rccl_comm = rccl_comm_create(unique_id128, rank, world_size);
  s->ar_fn = nullptr; s->rank = rank; s->world = world_size;
  s->force_exchange = world_size == 1 && std::getenv("MAVBA_FORCE_EXCHANGE") != nullptr;
  join_ranks(s);
  return MAVBA_OK;
  MAVBA_CATCH
}

// Host-only exercise of the in-process communicator group (tests/test_abi.py against tests/stubs/mock_rccl.c through
// MAVBA_RCCL_LIB; no device is touched): `calls` acquisitions of a group of `world` ranks, then - abort_after != 0 - an abort of
// the group and one more acquisition. out[0] = communicators of the last acquisition, out[1] = 1 when acquisitions 2..calls
// returned the handles of the first (the group is kept), out[2] = 1 when the acquisition after the abort returned NEW ones.
int mavba_debug_inproc_comms(int32_t world, int32_t calls, int32_t abort_after, int64_t* out) {
  MAVBA_TRY
  if (world < 1 || world > 16 || calls < 1 || !out) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "bad argument");
  const std::vector<int> dev((size_t)world, -1);
  std::vector<void*> first, cur;
  std::shared_ptr<RcclGuard> guard, guard0;
  std::string why;
  bool same = true;
  for (int c = 0; c < calls; ++c) {
    if (!inproc_comms_acquire(world, dev, cur, guard, why)) throw Failure(MAVBA_ERR_HIP, why);
    if (c == 0) guard0 = guard;
    if (c == 0) first = cur; else same = same && cur == first;
  }
  out[0] = (int64_t)cur.size(); out[1] = same ? 1 : 0; out[2] = 0;
  if (abort_after) {
    inproc_comms_abort();
    if (!inproc_comms_acquire(world, dev, cur, guard, why)) throw Failure(MAVBA_ERR_HIP, why);
    out[0] = (int64_t)cur.size();
    // (handles are heap pointers of the library: freed ones may be handed out again - the counts tell, see the test.) The old
    // group's guard says "aborted" to whoever still holds it, the new group has a guard of its own:
    out[2] = (guard0 && guard0->aborted && guard && guard != guard0 && !guard->aborted) ? 1 : 0;
  }
  return MAVBA_OK;
  MAVBA_CATCH
}

// (multi_gpu.hip) a communicator of the process-wide group of in-process ranks: used, not owned - it outlives the session
extern "C++" {
namespace mavba {
int session_borrow_rccl(mavba_session* s, void* comm, const std::shared_ptr<RcclGuard>& guard, int rank, int world_size) {
  MAVBA_SESSION_TRY(s)
  if (!comm || rank < 0 || rank >= world_size) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "bad communicator / rank");
  if (s->started) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "set_rccl must precede the first iteration");
  s->rccl_comm = comm; s->rccl_comm_owned = false; s->rccl_guard = guard;
  s->ar_fn = nullptr; s->rank = rank; s->world = world_size;
  s->force_exchange = false;
  join_ranks(s);
  return MAVBA_OK;
  MAVBA_CATCH
}
}  // namespace mavba
}  // extern "C++"

int mavba_session_eval_jacobian(mavba_session* s, double* cost, double* r, double* Jc, double* Jp, double* Jk) {
  MAVBA_SESSION_TRY(s)
  s->evaluate();
  if (cost) *cost = s->cost + s->fixed_cost;
  // probe path: the Jacobian planes are not part of the solve any more (J-free front end); materialise them here
  s->ensure_planes();
  launch_jacobian_sweep(s->st, s->sweep_args(s->d_camrec.p, s->d_intr.p, s->d_points.p));
  const size_t S = s->Nstride, N = s->N;
  auto pull = [&](const double* dev, int planes, std::vector<double>& h) {
    h.resize((size_t)planes * S);
    s->download(h.data(), dev, h.size() * 8);
  };
  std::vector<double> hR, hJp, hJc, hJk;
  pull(s->d_R.p, 2, hR); pull(s->d_Jp.p, 6, hJp); pull(s->d_Jc.p, 12, hJc); pull(s->d_Jk.p, 2 * s->KMAX, hJk);
  s->sync();
  s->ensure_perm_host();
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

// Debug / test entry (no device needed): the elimination tree the session set-up would choose for an image graph given
// as `npairs` coupled image pairs. node_of_image [NI] receives every image's tree node, node_parent [cap] the parents
// (-1: root); returns the number of nodes (0: no dissection) or a negative error code.
int mavba_debug_elimination_tree(int32_t NI, int32_t NC, int64_t npairs, const int32_t* pair_a, const int32_t* pair_b, int32_t max_depth,
                                 int32_t* node_of_image, int32_t* node_parent, int32_t cap) {
  MAVBA_TRY
  if (NI < 0 || NC < 0 || npairs < 0 || (npairs > 0 && (!pair_a || !pair_b)) || !node_of_image || !node_parent)
    throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "bad argument");
  std::vector<std::vector<int>> lower(NI);
  for (int64_t k = 0; k < npairs; ++k) {
    const int a = pair_a[k], b = pair_b[k];
    if (a < 0 || a >= NI || b < 0 || b >= NI) throw Failure(MAVBA_ERR_BAD_INDEX, "image pair out of range");
    if (a != b) lower[std::max(a, b)].push_back(std::min(a, b));
  }
  const std::vector<ElimNode> tn = elimination_tree(NI, NC, lower, -1, max_depth);
  if ((int)tn.size() > cap) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "node_parent too small");
  for (int i = 0; i < NI; ++i) node_of_image[i] = -1;
  for (size_t t = 0; t < tn.size(); ++t) {
    node_parent[t] = tn[t].parent;
    for (int i : tn[t].imgs) node_of_image[i] = (int)t;
  }
  return (int)tn.size();
  MAVBA_CATCH
}

int mavba_session_reduced_system(mavba_session* s, double radius, double* Sout, double* vout) {
  MAVBA_SESSION_TRY(s)
  if (!s->evaluated) s->evaluate();
  s->assemble(radius);
  // the device matrix is in elimination order; hand it out in the variables' order
  const int n = s->n_full, m = s->n_mat, nbt = m / 64;
  // the device keeps the envelope's tiles only (tile store, internal.h): spread them into the dense (m + 1) x m form
  std::vector<double> store(s->chol_struct.store_doubles());
  s->download(store.data(), s->d_M.p, store.size() * 8);
  s->sync();
  std::vector<double> h((size_t)(m + 1) * m, 0.0);
  const std::vector<int>& slot = s->chol_struct.tile_slot;
  for (int tr = 0; tr <= nbt; ++tr)
    for (int tc = 0; tc < nbt; ++tc) {
      const int sl = slot[(size_t)tr * nbt + tc];
      if (sl < 0) continue;
      const double* T = &store[(size_t)sl << 12];
      const int rows = tr < nbt ? 64 : 1;  // (tile row nbt: the right-hand side, first row)
      for (int r = 0; r < rows; ++r) std::memcpy(&h[(size_t)(64 * tr + r) * m + 64 * tc], T + 64 * r, 64 * 8);
    }
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
  MAVBA_SESSION_TRY(s)
  if (!s->evaluated) s->evaluate();
  double h[SC_COUNT];
  s->linear_step(radius, h);
  if (model_cost_change) *model_cost_change = h[SC_MODEL_CHANGE];
  if (d_poses && s->NI) s->download(d_poses, s->d_delta_cam.p, (size_t)s->NI * 48);
  if (d_intr && s->NC) s->download(d_intr, s->d_delta_cam.p + 6 * (size_t)s->NI, (size_t)s->NC * 72);
  std::vector<double> hdp;
  if (d_points && s->NP) { hdp.resize((size_t)s->NP * 3); s->download(hdp.data(), s->d_delta_pts.p, (size_t)s->NP * 24); }
  s->sync();
  if (d_points) s->to_caller_points(hdp.data(), d_points, 3);
  if (h[SC_FAIL] != 0.0 || h[SC_FAIL_FRONT] != 0.0) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "linear solve failed (matrix not positive definite)");
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_session_time_jacobian(mavba_session* s, int32_t reps, float* ms_avg) {
  MAVBA_SESSION_TRY(s)
  if (reps < 1) reps = 1;
  s->ensure_planes();  // (the materialising sweep is a probe: SURVEY 8(d) prices the Jacobian kernel with J written out)
  launch_cam_prepare(s->st, s->NI, s->d_poses.p, s->d_camrec.p);
  SweepArgs a = s->sweep_args(s->d_camrec.p, s->d_intr.p, s->d_points.p);
  // warm-up: ~25 ms of the same kernel, so that the timed launches run at the clocks of a busy device (a cold burst of 20
  // launches measured 134 us at C3 where the same kernel took 111 us inside a running solve)
  for (int i = 0; i < 200; ++i) launch_jacobian_sweep(s->st, a);
  std::vector<hipEvent_t> ev((size_t)reps + 1);
  for (auto& e : ev) HIP_OK(hipEventCreate(&e));
  // one event between consecutive launches, everything enqueued at once: the device stays busy, every launch has its own bracket
  HIP_OK(hipEventRecord(ev[0], s->st));
  for (int i = 0; i < reps; ++i) {
    launch_jacobian_sweep(s->st, a);
    HIP_OK(hipEventRecord(ev[(size_t)i + 1], s->st));
  }
  HIP_OK(hipStreamSynchronize(s->st));
  float ms = 0.f;
  for (int i = 0; i < reps; ++i) {
    float one = 0.f;
    HIP_OK(hipEventElapsedTime(&one, ev[(size_t)i], ev[(size_t)i + 1]));
    ms += one;
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (ms_avg) *ms_avg = ms / reps;
  s->sync();
  return MAVBA_OK;
  MAVBA_CATCH
}

// Test entry: the tile structure and the persistent schedule of a factorisation WITHOUT a device - what CholStructure::build
// computes on the host (tree validation, envelope, update lists ordered by the timing model, the helpers' queues from list
// scheduling). The CPU test-suite plays the queues against random task durations and checks that every launch completes.
//   out[8]: schedule exists | modelled forward us | launch-per-panel estimate us | grid | chain work-groups | tiles | updates | nodes
//   tasks_out: 6 ints per task in queue order (work-group, kind, i, j, first update, end update); upd_out: the update lists
int mavba_debug_chol_schedule(int32_t nb, int32_t num_nodes, const int32_t* node_begin, const int32_t* node_end, const int32_t* node_parent,
                              int64_t num_pairs, const int32_t* pair_row, const int32_t* pair_col, int32_t cus, double* out,
                              int32_t* tasks_out, int64_t tasks_cap, int64_t* num_tasks, int32_t* upd_out, int64_t upd_cap,
                              int64_t* num_upd, int32_t* chain_info_out) {
  MAVBA_TRY
  if (nb < 1 || num_nodes < 0 || num_pairs < 0 || !out || !num_tasks || !num_upd || cus < 2) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "bad argument");
  std::vector<CholNode> tree;
  for (int n = 0; n < num_nodes; ++n) tree.push_back(CholNode{node_begin[n], node_end[n], node_parent[n]});
  std::vector<std::pair<int, int>> pairs;
  for (int64_t q = 0; q < num_pairs; ++q) {
    if (pair_row[q] < 0 || pair_row[q] >= nb || pair_col[q] < 0 || pair_col[q] > pair_row[q]) throw Failure(MAVBA_ERR_BAD_INDEX, "tile pair out of range");
    pairs.emplace_back(pair_row[q], pair_col[q]);
  }
  CholStructure cs;
  cs.host_only = true;
  cs.host_only_cus = cus;
  if (cs.build(nb, pairs, tree, nullptr) != hipSuccess) throw Failure(MAVBA_ERR_HIP, "structure build failed");
  out[0] = cs.persist_ok ? 1.0 : 0.0; out[1] = cs.predicted_forward_us; out[2] = cs.lpp_estimate_us; out[3] = cs.persist_grid;
  out[4] = cs.persist_chain_wgs; out[5] = (double)cs.persist_tiles; out[6] = (double)cs.persist_updates; out[7] = cs.nseg;
  *num_tasks = (int64_t)cs.h_tasks.size();
  *num_upd = (int64_t)cs.h_upd.size();
  if (tasks_out && (int64_t)cs.h_tasks.size() <= tasks_cap) {
    int wg = 0;
    for (size_t t = 0; t < cs.h_tasks.size(); ++t) {
      while (wg + 1 < (int)cs.h_wg_begin.size() && cs.h_wg_begin[wg + 1] <= (int)t) ++wg;
      const CholTask& T = cs.h_tasks[t];
      int32_t* o = tasks_out + 6 * t;
      o[0] = wg; o[1] = T.kind; o[2] = T.i; o[3] = T.j; o[4] = T.ub; o[5] = T.ue;
    }
  }
  if (upd_out && (int64_t)cs.h_upd.size() <= upd_cap) std::copy(cs.h_upd.begin(), cs.h_upd.end(), upd_out);
  if (chain_info_out) for (int j = 0; j < nb; ++j) chain_info_out[j] = j < (int)cs.h_chain_info.size() ? cs.h_chain_info[j] : 0;
  return MAVBA_OK;
  MAVBA_CATCH
}

// Test entry: the LM decision function (lm_decide.h) evaluated by the host build and by the device build on the same
// `n` cases (SC_COUNT scalars + 8 parameters each); 6 doubles per case out of each. The speculative evaluation relies on
// the two agreeing bit for bit.
int mavba_debug_lm_decide(int32_t n, const double* cases, double* out_host, double* out_device, int32_t device) {
  MAVBA_TRY
  if (n < 0 || (n > 0 && (!cases || !out_host || !out_device))) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "bad argument");
  for (int c = 0; c < n; ++c) {
    const double* q = cases + (size_t)c * (SC_COUNT + 8);
    LmSpec sp = lm_spec_off();
    sp.radius = q[SC_COUNT]; sp.decrease_factor = q[SC_COUNT + 1]; sp.ptol = q[SC_COUNT + 2]; sp.ftol = q[SC_COUNT + 3];
    sp.min_rel_dec = q[SC_COUNT + 4]; sp.max_radius = q[SC_COUNT + 5]; sp.abs_gtol = q[SC_COUNT + 6]; sp.pending_eval = q[SC_COUNT + 7] != 0.0;
    const LmDecision d = lm_decide(q, sp);
    double* o = out_host + (size_t)c * 6;
    o[0] = (double)d.code; o[1] = d.radius; o[2] = d.decrease_factor; o[3] = d.rel; o[4] = d.step_norm; o[5] = d.cost_change;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); throw Failure(MAVBA_ERR_NO_DEVICE, "no HIP device: the mavba backend has no CPU path"); }
  if (n == 0) return MAVBA_OK;
  DeviceGuard g(device >= 0 ? device : 0);
  double *din = nullptr, *dout = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&din), (size_t)n * (SC_COUNT + 8) * 8));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&dout), (size_t)n * 6 * 8));
  HIP_OK(hipMemcpy(din, cases, (size_t)n * (SC_COUNT + 8) * 8, hipMemcpyHostToDevice));
  launch_lm_decide_cases(nullptr, n, din, dout);
  HIP_OK(hipMemcpy(out_device, dout, (size_t)n * 6 * 8, hipMemcpyDeviceToHost));
  (void)hipFree(din); (void)hipFree(dout);
  return MAVBA_OK;
  MAVBA_CATCH
}

// Event brackets around the kernels on / off for the iterations that follow (options.profile_kernels): a bracket costs ~10 us
// of queue drain per kernel, so the bench times its steps without them and collects the kernel timers in a second pass.
int mavba_session_set_profiling(mavba_session* s, int32_t on) {
  MAVBA_SESSION_TRY(s)
  s->sync();
  s->opt.profile_kernels = on ? 1 : 0;
  return MAVBA_OK;
  MAVBA_CATCH
}

// Probe: the front end of a linear solve (k_schur_rows / k_schur_fused, or k_point_front) for trust-region radius
// `radius`, `reps` launches back to back at the current parameters, average milliseconds per pass (HIP events on the
// session's stream). Leaves the session as an evaluation would.
int mavba_session_time_front(mavba_session* s, double radius, int32_t reps, float* ms_avg) {
  MAVBA_SESSION_TRY(s)
  if (reps < 1) reps = 1;
  if (!s->front_ok) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "time_front: the session runs on the plane kernels");
  if (!s->evaluated || !s->scales_ready) s->evaluate();
  for (int i = 0; i < 3; ++i) s->launch_front(radius, true);
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, s->st));
  for (int i = 0; i < reps; ++i) s->launch_front(radius, true);
  HIP_OK(hipEventRecord(e1, s->st));
  HIP_OK(hipStreamSynchronize(s->st));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (ms_avg) *ms_avg = ms / reps;
  s->sync();
  return MAVBA_OK;
  MAVBA_CATCH
}

// Test entry of the batched small uploads (host_util.hip): `n` buffers of sizes[i] bytes get a pattern, a clear (sizes[i] < 0:
// cleared, -sizes[i] bytes) or - every third one - a second upload over the first (the overlap rule), all inside ONE batch with an
// arena of `arena_bytes` (small: the arena-full path); everything is read back and compared. Returns the number of wrong bytes.
int64_t mavba_debug_upload_batch(int32_t n, const int64_t* sizes, int64_t arena_bytes, int32_t device) {
  if (mavba_device_count() <= 0) { g_last_error = "no HIP device: the mavba backend has no CPU path"; return -1; }
  int64_t wrong = -1;
  try {
    if (device >= 0) HIP_OK(hipSetDevice(device));
    hipStream_t st;
    HIP_OK(stream_acquire(&st));
    int st_dev = 0;
    (void)hipGetDevice(&st_dev);
    struct Release { hipStream_t st; int dev; ~Release() { (void)hipStreamSynchronize(st); release_staged(st); stream_release(st, dev); upload_batch_debug_arena(0); } } rel{st, st_dev};
    upload_batch_debug_arena((size_t)arena_bytes);
    std::vector<std::unique_ptr<DevBuf<unsigned char>>> bufs;
    std::vector<std::vector<unsigned char>> want;
    if (!upload_batch_begin(st)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "a batch is already open on this thread");
    struct End { bool open = true; ~End() { if (open) (void)upload_batch_end(false); } } guard;
    for (int i = 0; i < n; ++i) {
      const bool clear = sizes[i] < 0;
      const size_t bytes = (size_t)(clear ? -sizes[i] : sizes[i]);
      bufs.emplace_back(new DevBuf<unsigned char>);
      std::vector<unsigned char> h(bytes);
      for (size_t k = 0; k < bytes; ++k) h[k] = (unsigned char)(1 + (k * 7 + (size_t)i * 13) % 251);
      if (clear) {
        bufs.back()->upload(std::vector<unsigned char>(std::max<size_t>(bytes, 1), 0xEE), st);  // dirty first, then the clear: same destination twice
        bufs.back()->n = bytes;
        bufs.back()->zero(st);
        std::fill(h.begin(), h.end(), 0);
      } else {
        bufs.back()->upload(h, st);
        if (i % 3 == 2) {  // once more with other content (pointer-stable: DevBuf::upload frees and re-allocates the same class)
          for (size_t k = 0; k < bytes; ++k) h[k] = (unsigned char)(255 - h[k]);
          bufs.back()->upload(h, st);
        }
      }
      want.push_back(std::move(h));
    }
    guard.open = false;
    HIP_OK(upload_batch_end(true));
    wrong = 0;
    for (int i = 0; i < n; ++i) {
      std::vector<unsigned char> got(want[i].size());
      if (!got.empty()) HIP_OK(copy_d2h_staged_sync(got.data(), bufs[i]->p, got.size(), st));
      for (size_t k = 0; k < got.size(); ++k) wrong += got[k] != want[i][k];
    }
  } catch (const Failure& f) {
    g_last_error = f.what();
    return -1;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
  return wrong;
}

int mavba_session_get_info(mavba_session* s, mavba_session_info* out) {
  MAVBA_SESSION_TRY(s)
  if (!out) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null argument");
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
  out->chol_model_forward_us = cs.persist_ok ? cs.predicted_forward_us : 0.0;
  out->reduced_store_bytes = (int64_t)(cs.store_doubles() * sizeof(double));
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
  // MAVBA_GPUS=N: this one call - the one an unchanged mapper.cc makes (sequential_mapper.cc:1074-1080) - shards the points
  // over N devices of this process (multi_gpu.hip). Small problems stay on one GPU: the exchange would cost more than it saves.
  const char* min_obs_env = std::getenv("MAVBA_GPUS_MIN_OBS");
  if (problem && options && options->device < 0 && problem->num_obs >= (min_obs_env ? std::atoll(min_obs_env) : 200000ll)) {
    const int world = multi_gpu_ranks();
    if (world > 1) {
      MAVBA_TRY
      return solve_multi_gpu(problem, options, result, point_error, world);
      MAVBA_CATCH
    }
  }
  const bool tt = std::getenv("MAVBA_SETUP_TIMING") != nullptr;
  double t0 = now_s();
  auto lap = [&](const char* what) { if (tt) { const double t = now_s(); std::fprintf(stderr, "[solve] %-28s %8.2f ms\n", what, 1e3 * (t - t0)); t0 = t; } };
  mavba_session* s = nullptr;
  int rc = mavba_session_create(problem, options, &s);
  if (rc != MAVBA_OK) return rc;
  lap("session create");
  int done = 0, term = 0;
  rc = mavba_session_iterate(s, options->max_num_iterations + 1, &done, &term);
  lap("iterate");
  if (tt) std::fprintf(stderr, "[solve]   of it waiting for the device  %8.2f ms (%d iterations)\n", 1e3 * s->pub_wait_seconds, done);
  if (rc == MAVBA_OK && result) rc = mavba_session_result(s, result);
  // ceres leaves the user's parameter blocks untouched after NUMERICAL_FAILURE
  if (rc == MAVBA_OK && term != MAVBA_TERM_NUMERICAL_FAILURE)
    rc = mavba_session_get_params(s, problem->poses, problem->intrinsics, problem->points);
  lap("write-back");
  if (rc == MAVBA_OK && point_error && options->update_point_errors) {
    // problem.Evaluate runs on the user's blocks (bundle_adjustment.cc:583-588): after NUMERICAL_FAILURE those still hold
    // the parameters the call started from
    if (term == MAVBA_TERM_NUMERICAL_FAILURE) {
      try { DeviceGuard g(s->device); s->restore_initial_params(); } catch (const std::exception& e) { g_last_error = e.what(); rc = MAVBA_ERR_HIP; }
    }
    if (rc == MAVBA_OK) rc = mavba_session_point_errors(s, point_error);
  }
  lap("point errors");
  mavba_session_destroy(s);
  lap("session destroy");
  return rc;
}

// The second problem of mavba_solve_filter_solve built afresh on the host: the filtered points' observations leave, and
// the constancy flags follow the reference's "> 1 residual block" rule (bundle_adjustment.cc:361-385) - an image left
// with one observation is free, a camera is constant only if an image with more than one observation holds it so.
static int solve_filtered_afresh(const mavba_problem* problem, const mavba_options* options, const uint8_t* removed,
                                 mavba_result* second, double* point_error) {
  const int NI = problem->num_images, NC = problem->num_cameras;
  const long long NO = problem->num_obs;
  std::vector<double> uv;
  std::vector<int32_t> oi, op;
  std::vector<int> per_img((size_t)std::max(NI, 1), 0);
  for (long long o = 0; o < NO; ++o) {
    if (removed[problem->obs_point[o]]) continue;
    uv.push_back(problem->obs_uv[2 * o]); uv.push_back(problem->obs_uv[2 * o + 1]);
    oi.push_back(problem->obs_image[o]); op.push_back(problem->obs_point[o]);
    per_img[problem->obs_image[o]]++;
  }
  std::vector<uint8_t> pose_const((size_t)std::max(NI, 1), 0), intr_const((size_t)std::max(NC, 1), 0);
  for (int i = 0; i < NI; ++i) {
    if (per_img[i] <= 1) continue;
    if (problem->pose_const) pose_const[i] = problem->pose_const[i];
    if (problem->intr_const && problem->intr_const[problem->image_camera[i]]) intr_const[problem->image_camera[i]] = 1;
  }
  mavba_problem P = *problem;
  P.num_obs = (int64_t)oi.size();
  P.obs_uv = uv.data(); P.obs_image = oi.data(); P.obs_point = op.data();
  P.pose_const = pose_const.data(); P.intr_const = intr_const.data();
  return mavba_solve(&P, options, second, point_error);
}

int mavba_solve_filter_solve(const mavba_problem* problem, const mavba_options* options, double filter_max_error,
                             const uint8_t* keep, mavba_result* first, mavba_result* second, double* point_error,
                             uint8_t* removed, int64_t* num_removed) {
  mavba_session* s = nullptr;
  int rc = mavba_session_create(problem, options, &s);
  if (rc != MAVBA_OK) return rc;
  int done = 0, term = 0;
  rc = mavba_session_iterate(s, options->max_num_iterations + 1, &done, &term);
  if (rc == MAVBA_OK && first) rc = mavba_session_result(s, first);
  // (after NUMERICAL_FAILURE the filter and the second solve start from the parameters the call came with, as the
  // reference's second bundle_adjustment() call would: mavba_session::filter_points restores them)
  std::vector<double> errors((size_t)std::max(problem->num_points, 1), 0.0);
  if (rc == MAVBA_OK) rc = mavba_session_filter_points(s, filter_max_error, keep, removed, errors.data(), num_removed);
  if (rc == MAVBA_ERR_NEEDS_REBUILD) {
    // the filtered problem has another block structure: write the first solve back, then solve the second problem
    // the way a second mavba_solve call would
    std::vector<uint8_t> rem((size_t)std::max(problem->num_points, 1), 0);
    long long n = 0;
    for (int p = 0; p < problem->num_points; ++p) {
      rem[p] = !(keep && keep[p]) && errors[p] > filter_max_error;  // (NaN: no observations, never filtered)
      n += rem[p];
    }
    if (removed) std::memcpy(removed, rem.data(), (size_t)problem->num_points);
    if (num_removed) *num_removed = n;
    rc = MAVBA_OK;
    if (term != MAVBA_TERM_NUMERICAL_FAILURE) rc = mavba_session_get_params(s, problem->poses, problem->intrinsics, problem->points);
    mavba_session_destroy(s);
    if (rc != MAVBA_OK) return rc;
    mavba_options o2 = *options;
    o2.update_point_errors = point_error ? 1 : 0;
    return solve_filtered_afresh(problem, &o2, rem.data(), second, point_error);
  }
  if (rc == MAVBA_OK) rc = mavba_session_iterate(s, options->max_num_iterations + 1, &done, &term);
  if (rc == MAVBA_OK && second) rc = mavba_session_result(s, second);
  if (rc == MAVBA_OK && term != MAVBA_TERM_NUMERICAL_FAILURE)
    rc = mavba_session_get_params(s, problem->poses, problem->intrinsics, problem->points);
  if (rc == MAVBA_OK && point_error) {
    if (term == MAVBA_TERM_NUMERICAL_FAILURE) {
      try { DeviceGuard g(s->device); s->restore_initial_params(); } catch (const std::exception& e) { g_last_error = e.what(); rc = MAVBA_ERR_HIP; }
    }
    if (rc == MAVBA_OK) rc = mavba_session_point_errors(s, point_error);
  }
  mavba_session_destroy(s);
  return rc;
}

int mavba_pose_refine_batch(int32_t count, mavba_pose_refine_item* items, const mavba_options* options, mavba_result* results) {
  if (count < 0 || (count > 0 && !items) || !options) { g_last_error = "null argument"; return MAVBA_ERR_INVALID_ARGUMENT; }
  if (count == 0) return MAVBA_OK;
  if (mavba_device_count() <= 0) { g_last_error = "no HIP device: the mavba backend has no CPU path"; return MAVBA_ERR_NO_DEVICE; }
  if (!(options->loss_scale_factor > 0.0)) { g_last_error = "loss_scale_factor must be > 0"; return MAVBA_ERR_INVALID_ARGUMENT; }
  MAVBA_TRY
  if (options->device >= 0) HIP_OK(hipSetDevice(options->device));
  pose_refine_batch(count, items, *options, results);
  return MAVBA_OK;
  MAVBA_CATCH
}

int mavba_pose_refine(double rvec[3], double tvec[3], const double* intrinsics, int32_t camera_model,
                      const double* uv, const double* xyz, const uint8_t* inlier_mask, int64_t n,
                      const mavba_options* options, mavba_result* result) {
  if (!rvec || !tvec || !intrinsics || !options || n < 0 || (n > 0 && (!uv || !xyz))) {
    g_last_error = "null argument"; return MAVBA_ERR_INVALID_ARGUMENT;
  }
  if (camera_model < 1 || camera_model > 3) { g_last_error = "bad camera model"; return MAVBA_ERR_BAD_MODEL; }
  if (!options->print_progress && !std::getenv("MAVBA_POSE_REFINE_SESSION")) {
    // the on-device trust-region loop (pose_refine.hip); the per-iteration table needs the host loop below
    mavba_pose_refine_item it;
    for (int k = 0; k < 3; ++k) { it.rvec[k] = rvec[k]; it.tvec[k] = tvec[k]; }
    it.intrinsics = intrinsics; it.camera_model = camera_model; it.uv = uv; it.xyz = xyz; it.inlier_mask = inlier_mask; it.n = n;
    const int rc = mavba_pose_refine_batch(1, &it, options, result);
    if (rc == MAVBA_OK) for (int k = 0; k < 3; ++k) { rvec[k] = it.rvec[k]; tvec[k] = it.tvec[k]; }
    return rc;
  }
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
  HIP_OK(stream_acquire(&st));  // (the cached stream the sessions use: a process that solves one thing at a time stays on one stream)
  int st_dev = 0;
  (void)hipGetDevice(&st_dev);
  struct Release { hipStream_t st; int dev; ~Release() { (void)hipStreamSynchronize(st); release_staged(st); stream_release(st, dev); } } rel{st, st_dev};
  const int n_pad = std::max(64, round_up(n, 64));
  int rc = MAVBA_OK;
  {
    CholStructure cs;
    HIP_OK(cs.build_dense(n_pad / 64));
    // the caller's dense matrix into the tile store of a dense structure (every lower tile + the right-hand-side row)
    const int nbt = n_pad / 64;
    std::vector<double> M(cs.store_doubles(), 0.0);
    auto at = [&](int r, int c) -> double& { return M[((size_t)cs.tile_slot[(size_t)(r >> 6) * nbt + (c >> 6)] << 12) + (r & 63) * 64 + (c & 63)]; };
    for (int i = 0; i < n_pad; ++i) at(i, i) = 1.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        if ((i >> 6) >= (j >> 6)) at(i, j) = A[(size_t)i * n + j];
    for (int j = 0; j < n; ++j) at(n_pad, j) = b[j];
    DevBuf<double> dM, dL, dy, dws, dfail;
    dM.upload(M, st); dL.alloc(M.size()); dy.alloc(n_pad); dws.alloc((size_t)2 * n_pad * 64); dfail.alloc(1); dfail.zero(st);
    std::vector<double> y(n_pad);
    double fail = 0.0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      dense_spd_solve_device(st, dM.p, n_pad, dy.p, dfail.p, dws.p, dL.p, cs, nullptr, nullptr, attempt == 0);
      HIP_OK(hipMemcpyAsync(&fail, dfail.p, 8, hipMemcpyDeviceToHost, st));
      HIP_OK(copy_d2h_staged_sync(y.data(), dy.p, (size_t)n_pad * 8, st));
      release_staged(st);
      if (fail < 1e29) break;
      // the persistent launch gave up (it leaves M untouched): once more with the launch-per-panel schedule
      dfail.zero(st);
    }
    std::memcpy(x, y.data(), (size_t)n * 8);
    if (fail != 0.0) { g_last_error = "matrix is not positive definite"; rc = MAVBA_ERR_INVALID_ARGUMENT; }
  }
  return rc;
  MAVBA_CATCH
}

}  // extern "C"

