// session_lm.hip — the Levenberg-Marquardt trust-region loop of the MI355X bundle-adjustment backend
// (evaluation, Schur assembly, reduced solve, candidate step, accept / reject).
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
#include "session.h"
#include "lm_decide.h"
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define MAVBA_CPU_RELAX() _mm_pause()
#else
#include <thread>
#define MAVBA_CPU_RELAX() std::this_thread::yield()
#endif

using namespace mavba;


// ===========================================================================
// Evaluation at the current x: residuals, Jacobian, cost, gradient norm (ceres
// Evaluator::Evaluate with jacobian != NULL).
// ===========================================================================
void mavba_session::ensure_planes() {
  if (planes_ready) return;
  d_R.alloc((size_t)2 * Nstride); d_Jp.alloc((size_t)6 * Nstride); d_Jc.alloc((size_t)12 * Nstride);
  d_Jk.alloc((size_t)2 * KMAX * Nstride);
  d_Wk.alloc((size_t)std::max(Q, 1) * 27);
  planes_ready = true;
}

// The J-free front end at the current x: cost partials, Cu, gu and - with `entries` - the points' factors and the Schur
// entry records for trust-region radius r.
void mavba_session::launch_front(double r, bool entries, const LmSpec& spec, const CamSweepArgs* with_sweep) {
  sweep_rode_along = false;
  prereduced = false;  // (new block partials: their long runs have to be pre-reduced again)
  FrontArgs f;
  f.spec = spec;
  f.sw = sweep_args(d_camrec.p, d_intr.p, d_points.p);
  f.num_tiles = num_front_tiles; f.NPs = NPs; f.tiles = d_front_tiles.p;
  f.pt_start = d_pt_start.p; f.q_start = d_q_start.p; f.q_cam = d_q_cam.p; f.q_pt = d_q_pt.p;
  f.pt_free = d_pt_free.p; f.scale_cam = d_scale_cam.p; f.scale_pt = d_scale_pt.p;
  f.radius = r; f.dmin = opt.min_lm_diagonal; f.dmax = opt.max_lm_diagonal;
  f.Cu = d_Cu.p; f.gu = d_gu.p; f.Gi = d_Gi.p; f.h = d_h.p; f.Epose = d_Epose.p; f.Eintr = d_Eintr.p;
  f.fail = d_scal.p + SC_FAIL_FRONT;
  f.trace = nullptr;
  // (one fill for the two failure slots - they are neighbours: the solve that follows this front end finds SC_FAIL clean)
  static_assert(SC_FAIL_FRONT == SC_FAIL + 1, "the two failure slots are cleared together");
  // (the speculative front end runs right behind k_lm_snapshot, which has cleared both slots on the device)
  if (entries) { if (!spec.dec) HIP_OK(hipMemsetAsync(d_scal.p + SC_FAIL, 0, 2 * sizeof(double), st)); fail_slot_clean = true; }
  if (entries && fused_now()) {
    // every observed point sits in a cluster: the cluster kernel evaluates the Jacobians itself and leaves the block
    // partials of S for this radius (no entry records in HBM)
    // (a local window with constant intrinsics: the camera sweep's chunks ride along as extra work-groups - one launch less)
    // (the sweep's form must be the one launch_camera_sweep would pick: constant intrinsics everywhere, or the Gram form of KMAX)
    const bool forms_agree = (Q == 0 && !any_intr_free) || (Q > 0 && any_intr_free);
    const CamSweepArgs* ride = (with_sweep && forms_agree) ? with_sweep : nullptr;
    timed("schur_fused", [&] {
      launch_schur_rows(st, f, Q > 0 ? KMAX : 0, rows_generic, num_clusters, d_rows_clusters.p, d_cl_tab.p, d_rows_lists.p, d_obs_meta.p,
                        d_rows_lanes.p, d_rows_emit.p, d_part[0].p, d_part[1].p, d_part[2].p, ride);
    });
    sweep_rode_along = ride != nullptr;
    eval_rows = num_clusters;
    if (num_tail_tiles > 0) {
      // the points behind the clusters (long tracks with their generic term lists, constant points): sums, factors and
      // entry records by the separate front end, its cost partials behind the clusters' ones
      FrontArgs t = f;
      t.tiles = d_tail_tiles.p; t.num_tiles = num_tail_tiles;
      t.sw.cost_partial = f.sw.cost_partial + num_clusters;
      timed("point_front", [&] { launch_point_front(st, t, Q > 0 ? KMAX : 0, true); });
      eval_rows += point_front_grid(num_tail_tiles);
    }
  } else {
    timed(entries ? "point_front" : "point_front_sums", [&] { launch_point_front(st, f, Q > 0 ? KMAX : 0, entries); });
    eval_rows = point_front_grid(num_front_tiles);
  }
  front_valid = entries;
  front_radius = r;
}

void mavba_session::evaluate_enqueue(double next_radius, const LmSpec& spec) {
  if (!camrec_current) timed("cam_prepare", [&] { launch_cam_prepare(st, NI, d_poses.p, d_camrec.p); });
  camrec_current = true;
  SweepArgs a = sweep_args(d_camrec.p, d_intr.p, d_points.p);
  CamSweepArgs c;
  c.NI = NI; c.NC = NC; c.chunks = d_sweep_chunks.p; c.num_chunks = num_sweep_chunks;
  c.im_uv = d_im_uv.p; c.im_pt = d_im_pt.p; c.camrec = d_camrec.p; c.intr = d_intr.p;
  c.img_cam = d_img_cam.p; c.cam_model = d_cam_model.p; c.points = d_points.p;
  c.pt_active = a.pt_active;
  c.loss_b = a.loss_b; c.loss_inv_b = a.loss_inv_b; c.partial = d_cam_partial.p;
  c.spec = spec;
  sweep_rode_along = false;
  if (front_ok) {
    // one pass: the evaluation's sums and (once the Jacobi scales exist and the next radius is known) the entries
    // Beyond local windows, up to MAVBA_SWEEP_RIDE_MAX_OBS observations (default 1 000 000): the sweep's chunks fill the tail of the
    // cluster launch - C2 4 095 -> 4 184 LM iterations/s, A/B/A/B, profiles/r06_ab_sweep_ride.txt. At C3 the merged launch is 19 us
    // shorter than the two (1 285 -> 1 310 iter/s) and it stays OFF there by default: the cluster kernel is that configuration's
    // dominant kernel and its roofline figures (time, flops, counters) are only meaningful for the kernel on its own; C5: no gain.
    static const long long ride_max_obs = [] { const char* e = std::getenv("MAVBA_SWEEP_RIDE_MAX_OBS"); return e ? std::atoll(e) : 1000000ll; }();
    const bool ride = scales_ready && num_sweep_chunks > 0 && (merge_small() || (merge_on && !sharded() && N <= ride_max_obs));
    launch_front(next_radius, scales_ready && next_radius > 0.0, spec, ride ? &c : nullptr);
  } else {
    ensure_planes();
    front_valid = false;
    timed("jacobian_sweep", [&] { launch_jacobian_sweep(st, a); });
    timed("point_reduce", [&] {
      launch_point_reduce(st, NP, NPs, Nstride, KMAX, d_pt_start.p, d_q_start.p, d_q_cam.p, d_obs_img.p, d_img_cam.p,
                          d_R.p, d_Jp.p, d_Jk.p, d_Cu.p, d_gu.p, d_Wk.p);
    });
  }
  if (!sweep_rode_along) timed("camera_sweep", [&] { launch_camera_sweep(st, c, KMAX, any_intr_free); });
  // The pre-reduction of the block partials the cluster kernel has just written (k_partial_reduce's tasks) rides in the next
  // kernel of the evaluation that does not depend on them - one launch less per linear solve (round 6; C3 778 -> 773 us). Only
  // when every partial of a pre-reduced run comes from the cluster kernel (no generic term lists: their chunk kernels run in
  // assemble()), on one rank, and outside the profiled pass (the timers attribute the work to the kernel it belongs to).
  PartialRide pride;
  static const bool pride_on = [] { const char* e = std::getenv("MAVBA_PARTIAL_RIDE"); return !e || std::atoi(e) != 0; }();
  if (pride_on && merge_on && front_ok && front_valid && fused_now() && num_reduce_tasks > 0 && num_chunks[0] + num_chunks[1] + num_chunks[2] == 0 &&
      num_tail_tiles == 0 && !sharded() && !opt.profile_kernels) {
    pride.n = num_reduce_tasks; pride.tasks = d_reduce_tasks.p; pride.pp = d_part[0].p; pride.ip = d_part[1].p; pride.ii = d_part[2].p;
  }
  if (num_priors > 0)
    timed("rot_prior", [&] {
      launch_rot_prior(st, num_priors, d_prior_img.p, d_prior_R0.p, prior_weight, d_poses.p, d_prior_res.p,
                       d_prior_jac.p, d_prior_cost.p, spec);
    });
  if (scales_ready && merge_small()) {
    // a local window: per-image and per-camera sums, norms and the evaluation's three scalars by ONE work-group
    timed("eval_small", [&] {
      EvalSmallArgs e;
      e.NI = NI; e.NC = NC; e.NP = NP; e.NPs = NPs; e.with_cams = any_intr_free ? 1 : 0; e.cam_part = rank == 0 ? 1 : 0;
      state_norms_grid(NI, NC, NP, &e.gp, &e.gc);
      e.img_chunk_start = d_img_chunk_start.p; e.cam_partial = d_cam_partial.p;
      e.prior_start = num_priors > 0 ? d_prior_start.p : nullptr; e.prior_res = d_prior_res.p; e.prior_jac = d_prior_jac.p;
      e.cam_img_start = d_cam_img_start.p; e.cam_imgs = d_cam_imgs.p; e.img_rec = d_img_rec; e.cam_rec = d_cam_rec;
      e.img_intr_tmp = d_img_intr_tmp.p;
      e.pose_free = d_pose_free.p; e.intr_free = d_intr_free.p; e.pt_free = d_pt_free.p;
      e.poses = d_poses.p; e.intr = d_intr.p; e.points = d_points.p; e.gu = d_gu.p; e.norm_partial = d_norm_partial.p;
      const int rows = e.gp + e.gc;
      e.T.t[0] = ReduceTask{d_norm_partial.p, rows, 2, 1, nullptr, 0, d_scal.p + SC_GRAD_MAX};
      e.T.t[1] = ReduceTask{d_norm_partial.p + 1, rows, 2, 0, nullptr, 0, d_scal.p + SC_XNORM2};
      e.T.t[2] = ReduceTask{d_sweep_partial.p, eval_cost_rows(), 1, 0, d_prior_cost.p, num_priors, d_scal.p + SC_COST};
      e.num_tasks = 3;
      e.spec = spec;
      launch_eval_small(st, e);
    });
    evaluated = true; assembled = false;
    return;
  }
  if (scales_ready && merge_on && !sharded() && NI <= 160) {
    // a medium problem on one rank: the evaluation's tail as two launches instead of four (k_eval_head / k_eval_tail, kernels.hip).
    // Measured: C2 (100 images) 24.0 -> 16.1 us; at C3 (500 images, two cameras of 250) the ONE work-group of the second launch
    // takes 34.5 us against 28.8 for the four launches - large problems keep those.
    EvalSmallArgs e;
    e.NI = NI; e.NC = NC; e.NP = NP; e.NPs = NPs; e.with_cams = any_intr_free ? 1 : 0; e.cam_part = rank == 0 ? 1 : 0;
    state_norms_grid(NI, NC, NP, &e.gp, &e.gc);
    if (eval_head_tail_fits(e.gc)) {
      timed("eval_tail", [&] {
        e.img_chunk_start = d_img_chunk_start.p; e.cam_partial = d_cam_partial.p;
        e.prior_start = num_priors > 0 ? d_prior_start.p : nullptr; e.prior_res = d_prior_res.p; e.prior_jac = d_prior_jac.p;
        e.cam_img_start = d_cam_img_start.p; e.cam_imgs = d_cam_imgs.p; e.img_rec = d_img_rec; e.cam_rec = d_cam_rec;
        e.img_intr_tmp = d_img_intr_tmp.p;
        e.pose_free = d_pose_free.p; e.intr_free = d_intr_free.p; e.pt_free = d_pt_free.p;
        e.poses = d_poses.p; e.intr = d_intr.p; e.points = d_points.p; e.gu = d_gu.p; e.norm_partial = d_norm_partial.p;
        const int rows = e.gp + e.gc;
        e.T.t[0] = ReduceTask{d_norm_partial.p, rows, 2, 1, nullptr, 0, d_scal.p + SC_GRAD_MAX};
        e.T.t[1] = ReduceTask{d_norm_partial.p + 1, rows, 2, 0, nullptr, 0, d_scal.p + SC_XNORM2};
        e.T.t[2] = ReduceTask{d_sweep_partial.p, eval_cost_rows(), 1, 0, d_prior_cost.p, num_priors, d_scal.p + SC_COST};
        e.num_tasks = 3;
        e.spec = spec;
        e.ride = pride;
        launch_eval_head_tail(st, e);
      });
      prereduced = pride.n > 0;
      evaluated = true; assembled = false;
      return;
    }
  }
  timed("camera_reduce", [&] {
    launch_camera_reduce(st, NI, NC, d_img_chunk_start.p, d_cam_partial.p, num_priors > 0 ? d_prior_start.p : nullptr,
                         d_prior_res.p, d_prior_jac.p, d_cam_img_start.p, d_cam_imgs.p, d_img_rec, d_cam_rec,
                         d_img_intr_tmp.p, any_intr_free, spec, pride);
  });
  prereduced = pride.n > 0;
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
                       d_intr.p, d_points.p, d_img_rec, d_cam_rec, d_gu.p, d_norm_partial.p, &rows, spec);
  });
  timed("reduce", [&] {
    ReduceTasks T;
    T.t[0] = ReduceTask{d_norm_partial.p, rows, 2, 1, nullptr, 0, d_scal.p + SC_GRAD_MAX};
    T.t[1] = ReduceTask{d_norm_partial.p + 1, rows, 2, 0, nullptr, 0, d_scal.p + SC_XNORM2};
    T.t[2] = ReduceTask{d_sweep_partial.p, eval_cost_rows(), 1, 0, d_prior_cost.p, num_priors, d_scal.p + SC_COST};
    launch_reduce_tasks(st, T, 3, spec);
  });
  if (sharded()) allreduce(d_scal.p + SC_COST, SC_EVAL_COUNT, 2);  // the evaluation's two sums, then max|g| (never the candidate's slots)
  evaluated = true; assembled = false;
}
void mavba_session::evaluate(double next_radius) {
  evaluate_enqueue(next_radius);
  double h[SC_COUNT];
  read_scalars(h);
  take_evaluation(h);
}

// Schur complement for the current Jacobian at trust-region radius r:
// rows [0, n_pad) of d_M <- S, row n_pad <- v   (SchurEliminator::Eliminate).
void mavba_session::assemble(double r) {
  const double dmin = opt.min_lm_diagonal, dmax = opt.max_lm_diagonal;
  if (front_ok) {
    // the entry records of this (x, radius) may already be there (written together with the evaluation); a rejected
    // step comes back with a smaller radius: the front end runs again, Jacobians recomputed, nothing was stored
    if (!(front_valid && front_radius == r)) launch_front(r, true);
    else if (!fail_slot_clean) HIP_OK(hipMemsetAsync(d_scal.p + SC_FAIL, 0, sizeof(double), st));  // (a repeated solve of the same system)
  } else {
    HIP_OK(hipMemsetAsync(d_scal.p + SC_FAIL, 0, sizeof(double), st));
    HIP_OK(hipMemsetAsync(d_scal.p + SC_FAIL_FRONT, 0, sizeof(double), st));
    // (the points' 3x3 factors are computed inside the entries kernel: one launch; points without observations are not free)
    timed("entries_pose", [&] {
      launch_factor_entries_pose(st, N, Nstride, NPs, d_obs_img.p, d_obs_pt.p, d_pt_free.p, d_Jc.p, d_Jp.p, d_scale_cam.p,
                                 d_scale_pt.p, d_Gi.p, d_h.p, d_Epose.p, d_pt_start.p, d_Cu.p, d_gu.p, r, dmin, dmax,
                                 d_scal.p + SC_FAIL);
    });
    timed("entries_intr", [&] {
      launch_entries_intr(st, Q, NI, NPs, d_q_pt.p, d_q_cam.p, d_Wk.p, d_scale_cam.p, d_scale_pt.p, d_Gi.p, d_h.p,
                          d_Eintr.p);
    });
  }
  // The assembly writes the same (structural) entries every time and the persistent factorisation only reads the
  // matrix: its zeros survive from one linear solve to the next. Only the launch-per-panel schedule, which factorises
  // in place, makes a fresh clear necessary.
  if (!M_is_clean) {
    timed("memset_S", [&] {
      // The store holds the envelope's tiles and nothing else (round 6; rounds 1-5: a dense (n + 64) n array of which the
      // envelope was 11 % at C5): what the in-place factorisation of the launch-per-panel schedule - or, with shards, the
      // exchange - left behind is cleared as one block.
      HIP_OK(hipMemsetAsync(d_M.p, 0, chol_struct.store_doubles() * sizeof(double), st));
      // unit diagonal of the columns no block of S covers (padding, entirely constant blocks); the finalize pass
      // writes the diagonal of the constant parameters inside its blocks itself
      launch_fix_diag(st, n_mat, chol_struct.d_tile_slot, rank == 0, d_col_var.p, d_scale_cam.p, d_M.p);
    });
    M_is_clean = true;
  }
  if (!(front_ok && fused_now())) timed("schur_clusters", [&] {
    launch_schur_clusters(st, cl_shape, num_clusters, d_clusters.p, d_cl_tab.p, d_pt_start.p, d_q_start.p, d_obs_meta.p,
                          d_q_meta.p, d_pt_clustered.p, d_Epose.p, d_Eintr.p, d_h.p, NPs, d_part[0].p, d_part[1].p,
                          d_part[2].p);
  });
  if (num_chunks[0] > 0) timed("schur_chunks_pp", [&] { launch_schur_chunks(st, BLK_PP, num_chunks[0], d_chunks[0].p, d_terms[0].p, d_Epose.p, d_Eintr.p, d_part[0].p); });
  if (num_chunks[1] > 0) timed("schur_chunks_ip", [&] { launch_schur_chunks(st, BLK_IP, num_chunks[1], d_chunks[1].p, d_terms[1].p, d_Epose.p, d_Eintr.p, d_part[1].p); });
  if (num_chunks[2] > 0) timed("schur_chunks_ii", [&] { launch_schur_chunks(st, BLK_II, num_chunks[2], d_chunks[2].p, d_terms[2].p, d_Epose.p, d_Eintr.p, d_part[2].p); });
  const int nbt = n_mat / 64;
  timed("schur_finalize", [&] {
    if (!(prereduced && front_valid && front_radius == r)) launch_partial_reduce(st, num_reduce_tasks, d_reduce_tasks.p, d_part[0].p, d_part[1].p, d_part[2].p);
    launch_schur_finalize(st, num_blocks, d_blocks.p, d_part[0].p, d_part[1].p, d_part[2].p, NI, NC, chol_struct.d_tile_slot, nbt, rank == 0,
                          r, dmin, dmax, d_img_cam.p, d_img_rec, d_cam_rec, d_scale_cam.p, d_off.p, d_off.p + NI, d_M.p);
  });
  if (sharded()) {
    // only the tiles the factorisation reads (lower, inside the structure) and the right-hand side travel
    launch_tiles_copy(st, num_ar_tiles, d_ar_tiles.p, chol_struct.d_tile_slot, nbt, d_M.p, d_ar_buf.p, true);
    allreduce(d_ar_buf.p, (long long)num_ar_tiles * 4096 + n_mat, 0);
    launch_tiles_copy(st, num_ar_tiles, d_ar_tiles.p, chol_struct.d_tile_slot, nbt, d_M.p, d_ar_buf.p, false);
  }
  assembled = true;
}

// Persistent launch or launch-per-panel schedule for this session: decided once, at the first linear solve - whichever entry
// point reaches it (start(), mavba_session_linear_step, the timing probes) -, so the process-wide cool-down after a
// time-out is honoured by all of them. Sharded sessions decided in join_ranks, for all ranks together.
void mavba_session::decide_persistent() {
  if (persist_decided) return;
  if (chol_struct.persist_ok && !sharded()) allow_persistent = persistent_allowed_now();
  persist_decided = true;
}

void mavba_session::solve_linear(double r) {
  decide_persistent();
  assemble(r);
  // (two timers: the forward factorisation - ONE kernel, k_chol_persist, on the persistent schedule - and the backward substitution)
  CamUpdateArgs u;
  const bool with_update = merge_small();
  if (with_update) {
    // (launch_update_cameras' arguments, as candidate_enqueue passes them)
    u.NI = NI; u.NC = NC; u.cam_part = rank == 0 ? 1 : 0; u.radius = r; u.dmin = opt.min_lm_diagonal; u.dmax = opt.max_lm_diagonal;
    u.scale_cam = d_scale_cam.p; u.img_rec = d_img_rec; u.cam_rec = d_cam_rec; u.poses = d_poses.p; u.intr = d_intr.p;
    u.cand_poses = d_cposes.p; u.cand_intr = d_cintr.p; u.delta_cam = d_delta_cam.p;
    u.partial3 = d_step_partial.p + 3 * (size_t)backsub_points_grid(NP); u.cand_camrec = d_ccamrec.p;
  }
  timed_split("chol_factor", "chol_backsolve", [&](hipEvent_t mid) {
    cameras_updated = dense_spd_solve_device(st, d_M.p, n_mat, d_ymat.p, d_scal.p + SC_FAIL, d_diag_ws.p, d_L.p, chol_struct, d_col_var.p, d_y.p,
                                             allow_persistent, mid, with_update ? &u : nullptr);
    cameras_updated_r = r;
  });
  assembled = false;  // the factorisation overwrote S
  fail_slot_clean = false;
  // (in place; and with shards the exchange leaves the SUM over ranks in tiles this rank's assembly does not rewrite)
  if (!(allow_persistent && chol_struct.persist_ok) || sharded()) M_is_clean = false;
}

// One LM linear step: reduced solve + candidate. The persistent factorisation bounds every wait; should a launch ever
// give up (its work-groups were not all resident: another persistent launch on the device, CU masking), the solve is
// repeated once with the launch-per-panel schedule and the session stays on that schedule.
void mavba_session::linear_step(double r, double* h) {
  solve_linear(r);
  candidate(r, h);
  if (h[SC_FAIL] >= 1e29 && allow_persistent) {
    std::fprintf(stderr, "mavba: persistent factorisation timed out, falling back to the launch-per-panel schedule\n");
    allow_persistent = false;
    persistent_timed_out();
    solve_linear(r);
    candidate(r, h);
  }
}

// Back-substitution, candidate x + delta, and its cost. Leaves the scalars on the host.
void mavba_session::candidate(double r, double* h) {
  candidate_enqueue(r);
  read_scalars(h);
}
// (the launches of candidate() without the read-back)
void mavba_session::candidate_enqueue(double r, ReduceTasks* tail, int* tail_count) {
  const double dmin = opt.min_lm_diagonal, dmax = opt.max_lm_diagonal;
  // cameras first: the point back-substitution reads their step (delta_cam)
  const int rows = backsub_points_grid(NP), ugroups = update_cameras_groups(NI);
  if (!(cameras_updated && cameras_updated_r == r))  // (a small system: the work-group that solved it has done this already)
    timed("update_cameras", [&] {
      launch_update_cameras(st, NI, NC, rank == 0, r, dmin, dmax, d_y.p, d_scale_cam.p, d_img_rec, d_cam_rec,
                            d_poses.p, d_intr.p, d_cposes.p, d_cintr.p, d_delta_cam.p, d_step_partial.p + 3 * (size_t)rows,
                            d_ccamrec.p);  // (+ the candidate's camera records: no separate cam_prepare launch)
    });
  cameras_updated = false;
  // (round 4: for launch-bound problems the candidate's cost is summed by the back-substitution kernel itself - the observations
  // of a block's points are in its caches, the new points in its LDS -, one launch less: a 10-image window 2.30 -> 2.19 ms. At
  // C3 / C5 the streaming k_cost_only is the faster way to do that pass (0.091 + 0.023 against 0.123 ms), so large problems keep
  // it. MAVBA_COST_FUSE_MAX_OBS moves the switch, 0 = never fuse)
  static const long long fuse_max_obs = [] { const char* e = std::getenv("MAVBA_COST_FUSE_MAX_OBS"); return e ? std::atoll(e) : 500000ll; }();  // (round 6: 200 000 -> 500 000 - C2, 300 000 observations: 4 058 -> 4 109 iter/s fused, A/B/A/B)
  const bool cost_separate = (long long)N > fuse_max_obs;
  timed("backsub_points", [&] {
    launch_backsub_points_jvp(st, NP, NPs, NI, r, dmin, dmax, sweep_args(d_camrec.p, d_intr.p, d_points.p), d_pt_start.p,
                              d_delta_cam.p, d_pt_free.p, d_Gi.p, d_h.p, d_Cu.p, d_gu.p, d_scale_pt.p, d_cpoints.p,
                              d_delta_pts.p, d_step_partial.p, d_ccamrec.p, d_cintr.p, cost_separate ? nullptr : d_sweep_partial.p);
  });
  SweepArgs a = sweep_args(d_ccamrec.p, d_cintr.p, d_cpoints.p);
  if (cost_separate) timed("cost_only", [&] { launch_cost_only(st, a); });
  if (num_priors > 0)
    timed("rot_prior", [&] {
      launch_rot_prior(st, num_priors, d_prior_img.p, d_prior_R0.p, prior_weight, d_cposes.p, d_prior_res.p,
                       d_prior_jac.p, d_prior_cost.p);
    });
  {
    const int nsweep = N > 0 ? (cost_separate ? jacobian_sweep_grid(N) : rows) : 0;
    ReduceTasks T;
    T.t[0] = ReduceTask{d_step_partial.p, rows + ugroups, 3, 0, nullptr, 0, d_scal.p + SC_STEP_NORM2};
    T.t[1] = ReduceTask{d_step_partial.p + 1, rows + ugroups, 3, 0, nullptr, 0, d_scal.p + SC_MODEL_CHANGE};
    T.t[2] = ReduceTask{d_step_partial.p + 2, rows + ugroups, 3, 0, nullptr, 0, d_scal.p + SC_CAND_XNORM2};
    T.t[3] = ReduceTask{d_sweep_partial.p, nsweep, 1, 0, d_prior_cost.p, num_priors, d_scal.p + SC_NEW_COST};
    if (tail) { *tail = T; *tail_count = 4; }
    else timed("reduce", [&] { launch_reduce_tasks(st, T, 4); });
  }
  if (sharded()) allreduce(d_scal.p + SC_CAND_BEGIN, SC_CAND_COUNT, 0);  // only what the candidate wrote: an evaluation enqueued before it keeps its (already global) sums
}

bool mavba_session::merge_small() const {
  return merge_on && !sharded() && NI <= 64 && NC <= 8 && NP <= 8192 && dense_spd_solve_is_small(n_mat, chol_struct);
}

void mavba_session::start() {
  { const char* e = std::getenv("MAVBA_MERGE"); merge_on = !e || std::atoi(e) != 0; }  // (read per solve: the tests compare both ways)
  { const char* e = std::getenv("MAVBA_SPECULATE"); speculate_on = !e || std::atoi(e) != 0; }  // (per solve, like MAVBA_MERGE)
  decide_persistent();
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

// Polls the host-mapped slot for the publication with sequence number lm_seq (k_lm_snapshot writes it last).
bool mavba_session::wait_publication(double* h) {
  volatile double* pub = lm_pub;
  const double t_end = now_s() + 2.0;
  for (long long spin = 0;; ++spin) {
    if (pub[SC_COUNT + 7] == lm_seq) break;
    MAVBA_CPU_RELAX();
    if ((spin & 0xFFFF) == 0xFFFF && now_s() > t_end) return false;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for (int i = 0; i < SC_COUNT; ++i) h[i] = pub[i];
  return true;
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
  // (with the hook every collective synchronises the host anyway; RCCL collectives are stream-ordered)
  // MAVBA_DEFER_WITH_HOOK (tests): run the deferred protocol over the hook too, so that in-process ranks on ONE GPU
  // exercise the exact sequence of collectives the RCCL path issues (evaluation group, then candidate group, one read-back)
  const bool defer = (!sharded() || rccl_comm || std::getenv("MAVBA_DEFER_WITH_HOOK") != nullptr) && !opt.print_progress;
  // Round 4: the evaluation at the candidate point is enqueued BEFORE the candidate's scalars are read (lm_decide.h): a
  // one-lane kernel decides on the device, the evaluation's kernels return at once unless it accepted. The host reads the
  // scalars that kernel published, runs the same decision function and either keeps the evaluation (accepted: what
  // `defer` enqueued AFTER the read-back before - the device no longer idles through the host's turn-around) or forgets it.
  // Not with shards (the evaluation's collective would run on a rejected step's stale sums) or the plane kernels.
  const bool speculate = speculate_on && defer && !sharded() && front_ok && fused_now() && rows_ok && scales_ready;
  if (speculate && !lm_pub) { lm_pub = lm_pub_alloc(); if (lm_pub) lm_pub[SC_COUNT + 7] = -1.0; }
  if (speculate && lm_pub && !d_lm_dec.p) d_lm_dec.alloc(2);
  const bool merge_tail = merge_on;
  bool pending_eval = false;
  while (termination == MAVBA_TERM_RUNNING && n < max_iters) {
    if (iteration >= opt.max_num_iterations) { termination = MAVBA_TERM_NO_CONVERGENCE; break; }
    ++iteration; ++n;
    double h[SC_COUNT];
    const LmSpec sp = lm_spec(pending_eval);
    bool speculated = false;
    // what the speculative evaluation changes in the session's books (restored if the step is not accepted)
    struct Books { bool evaluated, assembled, front_valid, fail_slot_clean; double front_radius; int eval_rows; bool prereduced; } books{};
    if (speculate && lm_pub) {
      solve_linear(radius);
      lm_seq += 1.0;
      if (merge_tail) {
        // the candidate's four reductions and the decision in one work-group
        ReduceTasks T; int nt = 0;
        candidate_enqueue(radius, &T, &nt);
        timed("lm_tail", [&] { launch_lm_tail(st, T, nt, sp, d_lm_dec.p, lm_pub, lm_seq, d_scal.p + SC_FAIL); });
      } else {
        candidate_enqueue(radius);
        timed("lm_snapshot", [&] { launch_lm_snapshot(st, sp, d_lm_dec.p, lm_pub, lm_seq, d_scal.p + SC_FAIL); });
      }
      books = Books{evaluated, assembled, front_valid, fail_slot_clean, front_radius, eval_rows, prereduced};
      std::swap(d_poses.p, d_cposes.p); std::swap(d_intr.p, d_cintr.p); std::swap(d_points.p, d_cpoints.p);
      std::swap(d_camrec.p, d_ccamrec.p);
      LmSpec ks = sp;
      ks.dec = d_lm_dec.p;
      evaluate_enqueue(1.0 /* "with entries": the front end takes the radius from the decision */, ks);
      speculated = true;
      const double tw = now_s();
      const bool published = wait_publication(h);
      pub_wait_seconds += now_s() - tw;
      if (!published) {  // (never seen; keeps the loop alive if the mapping is not coherent on some system)
        sync();
        for (int i = 0; i < SC_COUNT; ++i) h[i] = lm_pub[i];
      }
      if (opt.profile_kernels) sync();  // (flushes the event timers: a profiled pass gives up the overlap)
      if (h[SC_FAIL] >= 1e29 && allow_persistent) {
        // the persistent factorisation gave up: the device's decision was "invalid", the evaluation did not run; forget
        // it and repeat the solve on the launch-per-panel schedule (linear_step does, and reads back the plain way)
        std::swap(d_poses.p, d_cposes.p); std::swap(d_intr.p, d_cintr.p); std::swap(d_points.p, d_cpoints.p);
        std::swap(d_camrec.p, d_ccamrec.p);
        evaluated = books.evaluated; assembled = books.assembled;
        front_valid = false;  // (k_lm_snapshot has cleared SC_FAIL_FRONT too: the front end runs again and rewrites it)
        fail_slot_clean = false; front_radius = books.front_radius; eval_rows = books.eval_rows; prereduced = false;
        speculated = false;
        std::fprintf(stderr, "mavba: persistent factorisation timed out, falling back to the launch-per-panel schedule\n");
        allow_persistent = false;
        persistent_timed_out();
        solve_linear(radius);
        candidate(radius, h);
      }
    } else {
      linear_step(radius, h);
    }
    if (pending_eval) take_evaluation(h);  // (the evaluation at the current point, enqueued behind the previous accepted step)
    const LmDecision dec = lm_decide(h, sp);
    auto forget_speculation = [&] {
      if (!speculated) return;
      std::swap(d_poses.p, d_cposes.p); std::swap(d_intr.p, d_cintr.p); std::swap(d_points.p, d_cpoints.p);
      std::swap(d_camrec.p, d_ccamrec.p);
      evaluated = books.evaluated; assembled = books.assembled; front_valid = books.front_valid;
      fail_slot_clean = false;  // (the speculative front end's fill did run)
      front_radius = books.front_radius; eval_rows = books.eval_rows; prereduced = books.prereduced;
      speculated = false;
    };
    if (dec.code == LM_TERM_GTOL) {  // the previous iteration ended the solve: this one never happened
      forget_speculation();
      pending_eval = false;
      termination = MAVBA_TERM_GRADIENT_TOLERANCE;
      --iteration; --n;
      break;
    }
    pending_eval = false;
    if (dec.code == LM_INVALID) {
      forget_speculation();
      if (++invalid_steps >= opt.max_num_consecutive_invalid_steps) { termination = MAVBA_TERM_NUMERICAL_FAILURE; ++n_fail; break; }
    } else {
      invalid_steps = 0;
      if (dec.code == LM_TERM_PTOL) { forget_speculation(); termination = MAVBA_TERM_PARAMETER_TOLERANCE; break; }
      if (dec.code == LM_TERM_FTOL) { forget_speculation(); termination = MAVBA_TERM_FUNCTION_TOLERANCE; break; }
    }
    const bool successful = dec.code == LM_ACCEPTED;
    radius = dec.radius;
    decrease_factor = dec.decrease_factor;
    if (successful) {
      ++n_success;
      if (speculated) {
        // the evaluation at the accepted point is already running (pointers swapped, books written by evaluate_enqueue)
        front_radius = radius;
        pending_eval = true;
      } else {
        std::swap(d_poses.p, d_cposes.p); std::swap(d_intr.p, d_cintr.p); std::swap(d_points.p, d_cpoints.p);
        std::swap(d_camrec.p, d_ccamrec.p);  // the candidate's camera records are the new point's (camrec_current stays true)
        if (defer) {
          evaluate_enqueue(radius);
          pending_eval = true;
        } else {
          evaluate(radius);
          if (grad_max <= abs_gtol) termination = MAVBA_TERM_GRADIENT_TOLERANCE;
        }
      }
    } else {
      forget_speculation();
      ++n_fail;
    }
    if (opt.print_progress)
      std::printf("%4d %14.6e %12.2e %10.2e %10.2e %10.2e %10.2e\n", iteration, cost + fixed_cost, successful ? dec.cost_change : 0.0,
                  grad_max, dec.step_norm, dec.rel, radius);
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
  if (NP) download(h.data(), d_perr.p, (size_t)NP * 8);
  sync();
  evaluated = false;  // camrec still matches x, but keep the contract simple
  // Only points that have observations in the problem are touched (bundle_adjustment.cc:578-581);
  // observations dropped as all-constant blocks still count (they are residual blocks there).
  for (int p = 0; p < NP; ++p)
    if (h_pt_count_all[p] > 0 && (h_pt_removed.empty() || !h_pt_removed[p])) {
      const int po = h_pt_orig[p];
      out[po] = h_dropped_rnorm.empty() ? h[p] : h[p] + h_dropped_rnorm[po] / (double)h_pt_count_all[p];
    }
}

// A new solve from the current parameters: what a second bundle_adjustment() call on the same data does (ceres::Solve
// starts with a fresh trust region and re-estimates the Jacobi scaling), without any of the set-up.
void mavba_session::restart() {
  evaluated = scales_ready = started = assembled = false;
  front_valid = false;
  cameras_updated = false;  // (a camera update done inside a solve's launch belongs to that solve's candidate only)
  // The Jacobi scales are re-estimated: which columns are constant (unit diagonal) may differ from the previous solve
  // (filter_points), so the matrix is cleared and its constant / padding diagonal rewritten once (k_fix_diag).
  M_is_clean = false;
  radius = opt.initial_trust_region_radius; decrease_factor = 2.0;
  cost = x_norm = grad_max = abs_gtol = initial_cost = 0.0;
  iteration = invalid_steps = n_success = n_fail = 0;
  termination = MAVBA_TERM_RUNNING;
  solve_seconds = 0.0;
}

// filter_point_cloud (reference src/mapper.cc:382-402): points whose mean raw reprojection error exceeds max_error leave
// the problem (unless kept). On the resident session their observations get zero weight and their blocks stop being
// free - the index structure of the set-up is a superset of what the smaller problem needs and stays as it is - then
// the counts, the used / free flags of images and cameras and the fixed cost are re-derived and the solve restarts.
long long mavba_session::filter_points(double max_error, const unsigned char* keep, unsigned char* removed_out,
                                       double* errors_out) {
  if (sharded()) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "filter_points on a sharded session is not supported");
  const double t0 = now_s();
  // ceres::Solve leaves the user's parameter blocks untouched after NUMERICAL_FAILURE: the errors the reference's
  // filter would see are those at the parameters the failed solve STARTED from, and its second solve starts there too
  if (termination == MAVBA_TERM_NUMERICAL_FAILURE) restore_initial_params();
  std::vector<double> err((size_t)std::max(NP, 1), 0.0);
  // (NaN marks points without observations in the problem: never filtered, like points BA never reported on)
  std::fill(err.begin(), err.end(), std::numeric_limits<double>::quiet_NaN());
  point_errors(err.data());
  if (errors_out) std::memcpy(errors_out, err.data(), (size_t)NP * 8);
  std::vector<unsigned char> removed_now(h_pt_removed);
  if (removed_now.empty()) removed_now.assign((size_t)std::max(NP, 1), 0);
  long long removed = 0;
  for (int q = 0; q < NP; ++q) {
    const int po = h_pt_orig[q];
    const bool out = !removed_now[q] && !(keep && keep[po]) && err[po] > max_error;
    if (out) { removed_now[q] = 1; ++removed; }
  }
  // The caller's constancy flags came out of the reference's rule "a pose state (and, without refine_camera_params, the
  // constant intrinsics) applies only to an image that contributed MORE THAN ONE residual block"
  // (bundle_adjustment.cc:361-385). An image that the filter leaves with exactly one would be free in the reference's
  // second call: this session's structure has no blocks for it, so the caller has to build the second problem afresh.
  if (removed > 0) {
    std::vector<int> per_img((size_t)std::max(NI, 1), 0);
    for (int q = 0; q < NP; ++q)
      if (!removed_now[q])
        for (int a = h_pt_start[q]; a < h_pt_start[q + 1]; ++a) per_img[h_oimg[a]]++;
    for (int i = 0; i < NI; ++i)
      if (per_img[i] == 1 && (h_pose_const[i] != 0 || h_intr_const_in[h_img_cam[i]]))
        throw Failure(MAVBA_ERR_NEEDS_REBUILD, "filter_points: an image with constant blocks is left with one residual block "
                                               "(the reference frees it, bundle_adjustment.cc:361): build the filtered problem afresh");
  }
  if (removed_out) for (int q = 0; q < NP; ++q) removed_out[h_pt_orig[q]] = removed_now[q];
  if (removed == 0) { restart(); return 0; }  // (nothing filtered so far: the kernels keep their unmasked instantiations)
  h_pt_removed.swap(removed_now);
  apply_filter_state();
  restart();
  setup_seconds += now_s() - t0;
  return removed;
}

// Re-derive everything that depends on the set of residual blocks from h_pt_removed (empty: the problem as built).
void mavba_session::apply_filter_state() {
  std::vector<unsigned char> active((size_t)std::max(NP, 1), 1);
  std::fill(h_img_used.begin(), h_img_used.end(), 0);
  std::fill(h_cam_used.begin(), h_cam_used.end(), 0);
  const bool any = !h_pt_removed.empty();
  long long n_all = 0, n_kept = 0;
  fixed_cost = fixed_cost_priors;
  for (int q = 0; q < NP; ++q) {
    if (any && h_pt_removed[q]) { active[q] = 0; h_pt_used[q] = 0; continue; }
    n_all += h_pt_count_all[q];
    n_kept += h_pt_start[q + 1] - h_pt_start[q];
    h_pt_used[q] = h_pt_start[q + 1] > h_pt_start[q];
    if (!h_dropped_cost.empty()) fixed_cost += h_dropped_cost[h_pt_orig[q]];
    for (int a = h_pt_start[q]; a < h_pt_start[q + 1]; ++a) { h_img_used[h_oimg[a]] = 1; h_cam_used[h_img_cam[h_oimg[a]]] = 1; }
  }
  for (int i = 0; i < NI; ++i) if (h_prior_on_img[i]) h_img_used[i] = 1;
  num_residuals = 2 * n_all + num_priors_all;
  num_residuals_reduced = 2 * n_kept + num_priors;
  derive_free_flags();
  d_pose_free.upload(h_pose_free, st); d_intr_free.upload(h_intr_free, st); d_pt_free.upload(h_pt_free, st);
  if (any) d_pt_active.upload(active, st);
  // (constant / unused columns get their unit diagonal back when restart() clears the matrix: M_is_clean = false)
  sync();
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
