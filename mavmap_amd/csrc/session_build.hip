// session_build.hip - session set-up: validation, internal point / observation order, point clusters, Schur block
// structure, elimination order of the reduced system (reference semantics: src/base3d/bundle_adjustment.cc:228-549
// for what the caller's flat problem means; everything here is indexing for the device kernels).
#include "session.h"
#include <array>
#include <map>

using namespace mavba;


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

// The ordering block of build() on the host (problems with dropped all-constant blocks, small problems): the same
// definitions as order_on_device - point order = the 8 smallest images that see the point, then the caller's index.
void mavba_session::order_on_host(const mavba_problem* P, const long long* keptp, std::vector<int>& img_start,
                                  std::unique_ptr<PinnedBuf<double2>>& uv_h, std::unique_ptr<PinnedBuf<int>>& opt_h,
                                  std::unique_ptr<PinnedBuf<double2>>& im_uv_h, std::unique_ptr<PinnedBuf<int>>& im_pt_h) {
  const bool all_kept = keptp == nullptr;
  auto kept_at = [keptp](long long k) { return keptp ? keptp[k] : k; };
  const bool tt = std::getenv("MAVBA_SETUP_TIMING") != nullptr;
  double tl = now_s();
  auto lap = [&](const char* what) { if (tt) { const double t = now_s(); std::fprintf(stderr, "[setup]   host: %-20s %8.3f ms\n", what, 1e3 * (t - tl)); tl = t; } };
  // ---- internal point order ----
  // One stable counting sort of the kept observations by (caller's) point gives every point's bucket; the
  // point-major order is the buckets concatenated in the new point order.
  std::vector<int> pt_new(NP);
  std::vector<int> cstart;
  // Small problems (serial anyway; round 6, a local window: 0.19 -> 0.15 ms): no buckets - ONE pass over the observations counts
  // them per point and keeps every point's key up to date (the eight smallest distinct images and the hash do not depend on
  // the order in which a point's observations arrive), and the point-major arrays are written by a second pass through
  // per-point cursors (caller's order inside a point, as the stable counting sort gives it).
  const bool streaming = N < 50000 && !std::getenv("MAVBA_ORDER_GENERAL");
  HostBuf<long long> bucket(streaming ? 1 : N);  // kept observation ids, grouped by caller's point, input order inside
  // (pixel and image travel with it: the caller's arrays are read once, in order, instead of being gathered again)
  HostBuf<double2> buv(streaming ? 1 : N);
  HostBuf<int> bimg(streaming ? 1 : N);
  {
    HostBuf<int> simg(streaming ? 1 : N);
    HostBuf<unsigned short> k8(streaming ? (size_t)NP * 8 : 1);
    std::vector<unsigned> hsum;
    if (streaming) {
      cstart.assign((size_t)NP + 1, 0);
      hsum.assign(NP, 0u);
      std::memset(k8.data(), 0xFF, (size_t)NP * 8 * sizeof(unsigned short));
      for (long long k = 0; k < N; ++k) {
        const long long o = kept_at(k);
        const int p = P->obs_point[o];
        unsigned x = (unsigned)P->obs_image[o];
        cstart[(size_t)p + 1]++;
        hsum[p] += image_set_mix(x);
        unsigned short* kk = &k8[(size_t)p * 8];
        bool dup = false;
        for (int t = 0; t < 8; ++t) dup = dup || kk[t] == x;
        if (dup) continue;
        for (int t = 0; t < 8; ++t) if (x < kk[t]) { const unsigned y = kk[t]; kk[t] = (unsigned short)x; x = y; }
      }
      for (int p = 0; p < NP; ++p) cstart[(size_t)p + 1] += cstart[p];
    } else {
    counting_sort_parallel(N, NP, [&](long long k) { return P->obs_point[kept_at(k)]; }, cstart,
                           [&](long long k, int at) {
                             const long long o = kept_at(k);
                             bucket[at] = o; simg[at] = bimg[at] = P->obs_image[o];
                             buv[at] = make_double2(P->obs_uv[2 * o], P->obs_uv[2 * o + 1]);
                           });
    }
      if (all_kept)
      for (int p = 0; p < NP; ++p) { h_pt_count_all[p] = cstart[p + 1] - cstart[p]; h_pt_used[p] = h_pt_count_all[p] > 0; }
    lap("buckets by point");
    // Order: the 8 smallest DISTINCT images that see the point, 16 bits each in one 128-bit key (0xFFFF padded: a point
    // nobody sees sorts last), ties by the caller's point index - a strict total order, so the result does not depend on
    // the number of threads, and the same definition as k_point_keys + the radix sort of the device path.
    // ... behind the `tail` flag (most significant): points that can never be clustered - more than kTailObs observations, or
    // constant - follow all the others (the same definition as k_point_keys)
    // Round 4: a 32-bit hash of ALL the point's images follows the eight smallest (before the caller's index): points with
    // the same image SET become neighbours even when their lists are longer than eight - the clusters of k_schur_rows
    // (80 rows: 10 images + 2 cameras) are runs of such points. A sum of mixed image numbers: independent of the order
    // of the point's observations, the same on the host and in k_point_keys.
    struct KeyId { unsigned long long hi, lo; unsigned hash; int id; int tail; };
    HostBuf<KeyId> keyed(NP);
    parallel_ranges(NP, [&](long long b0, long long b1) {
      for (long long p = b0; p < b1; ++p) {
        unsigned k[8] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
        unsigned hash = 0;
        if (streaming) { for (int t = 0; t < 8; ++t) k[t] = k8[(size_t)p * 8 + t]; hash = hsum[p]; }
        else for (int a2 = cstart[p]; a2 < cstart[p + 1]; ++a2) {
          unsigned x = (unsigned)simg[a2];
          hash += image_set_mix(x);
          bool dup = false;
          for (int t = 0; t < 8; ++t) dup = dup || k[t] == x;
          if (dup) continue;
          for (int t = 0; t < 8; ++t) if (x < k[t]) std::swap(x, k[t]);
        }
        KeyId kk;
        kk.hi = (unsigned long long)k[0] << 48 | (unsigned long long)k[1] << 32 | (unsigned long long)k[2] << 16 | k[3];
        kk.lo = (unsigned long long)k[4] << 48 | (unsigned long long)k[5] << 32 | (unsigned long long)k[6] << 16 | k[7];
        kk.hash = hash;
        kk.id = (int)p;
        kk.tail = (cstart[p + 1] - cstart[p] > kTailObs || (P->point_const && P->point_const[p])) ? 1 : 0;
        keyed[p] = kk;
      }
    });
    lap("point keys");
    auto before = [&](const KeyId& x, const KeyId& y) {
      if (x.tail != y.tail) return x.tail < y.tail;
      if (x.hi != y.hi) return x.hi < y.hi;
      if (x.lo != y.lo) return x.lo < y.lo;
      if (x.hash != y.hash) return x.hash < y.hash;
      return x.id < y.id;
    };
    // Points are first dealt into buckets by their FIRST image (the key's leading 16 bits; a stable counting sort),
    // then every bucket is sorted on its own - no merge passes.
    if (NP >= 20000) {
      std::vector<int> fstart;
      HostBuf<KeyId> dealt(NP);
      // (bucket NI: points without observations, whose key starts with 0xFFFF)
      counting_sort_parallel(NP, 2 * (NI + 1), [&](long long p) { return keyed[p].tail * (NI + 1) + std::min((int)(keyed[p].hi >> 48), NI); }, fstart,
                             [&](long long p, int at) { dealt[at] = keyed[p]; });
      std::vector<int> nonempty;
      for (int k = 0; k < 2 * (NI + 1); ++k) if (fstart[k + 1] > fstart[k]) nonempty.push_back(k);
      parallel_ranges((long long)nonempty.size(), [&](long long k0, long long k1) {
        for (long long k = k0; k < k1; ++k) std::sort(dealt.data() + fstart[nonempty[k]], dealt.data() + fstart[nonempty[k] + 1], before);
      }, 2);
      parallel_ranges(NP, [&](long long b0, long long b1) { for (long long q = b0; q < b1; ++q) keyed[q] = dealt[q]; }, 20000);
    } else if (NI <= 31 && !std::getenv("MAVBA_ORDER_GENERAL")) {  // (the switch: tests compare the two sorts)
      // A local window (round 6): with at most 31 images the eight smallest images of a point are a 31-bit set (image 0 = the
      // most significant bit) and the lexicographic order of the 0xFFFF-padded ascending lists is the DESCENDING order of
      // those sets - at the first position where two lists differ, the smaller image is a bit one set has and the other has
      // not, and every higher bit is common. So (tail, hi, lo, hash) packs into one 64-bit integer and the sort moves 16-byte
      // pairs with a trivial comparison instead of 40-byte records with a five-level one (0.15 of a 0.3 ms ordering block at
      // 2 500 points); the order is the same, ties by the caller's index as before.
      std::vector<std::pair<unsigned long long, int>> packed(NP);
      for (int p = 0; p < NP; ++p) {
        const KeyId& kk = keyed[p];
        unsigned set = 0;
        for (int t = 0; t < 4; ++t) {
          const unsigned a = (unsigned)(kk.hi >> (48 - 16 * t)) & 0xFFFFu, b2 = (unsigned)(kk.lo >> (48 - 16 * t)) & 0xFFFFu;
          if (a != 0xFFFFu) set |= 1u << (30 - a);
          if (b2 != 0xFFFFu) set |= 1u << (30 - b2);
        }
        packed[p] = {(unsigned long long)kk.tail << 63 | (unsigned long long)(~set & 0x7FFFFFFFu) << 32 | kk.hash, kk.id};
      }
      if (NI <= 12) {
        // ... and with at most 12 images (tail, set) is one of 8 192 values: a stable counting sort puts the points of one
        // image set together in the caller's order; only a set whose points differ in the hash - an image seen twice by some
        // of them, more than eight images - needs a sort of its own (0.125 -> 0.03 ms at 2 500 points)
        const int shift = 31 - NI, nbuckets = 2 << NI;
        std::vector<int> start((size_t)nbuckets + 1, 0);
        auto bucket_of = [&](unsigned long long key) { return (int)(key >> 63) << NI | (int)((unsigned)(key >> 32) & 0x7FFFFFFFu) >> shift; };
        for (int p = 0; p < NP; ++p) start[(size_t)bucket_of(packed[p].first) + 1]++;
        for (int b2 = 0; b2 < nbuckets; ++b2) start[b2 + 1] += start[b2];
        std::vector<std::pair<unsigned long long, int>> dealt(NP);
        {
          std::vector<int> cur(start.begin(), start.end() - 1);
          for (int p = 0; p < NP; ++p) dealt[cur[bucket_of(packed[p].first)]++] = packed[p];  // (ids ascend inside a bucket: keyed[] is in the caller's order)
        }
        for (int b2 = 0; b2 < nbuckets; ++b2) {
          const int q0 = start[b2], q1 = start[b2 + 1];
          bool same = true;
          for (int q = q0 + 1; q < q1 && same; ++q) same = dealt[q].first == dealt[q0].first;
          if (!same) std::sort(dealt.begin() + q0, dealt.begin() + q1);
        }
        packed.swap(dealt);
      } else {
        std::sort(packed.begin(), packed.end());
      }
      for (int q = 0; q < NP; ++q) keyed[q].id = packed[q].second;
    } else {
      std::sort(keyed.data(), keyed.data() + NP, before);
    }
    lap("sort");
    h_pt_orig.resize(NP);
    for (int q = 0; q < NP; ++q) h_pt_orig[q] = keyed[q].id;
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

  lap("permute point arrays");
  // ---- point-major order: the buckets in the new point order ----
  h_pt_start.assign(NP + 1, 0);
  for (int q = 0; q < NP; ++q) h_pt_start[q + 1] = h_pt_start[q] + (cstart[h_pt_orig[q] + 1] - cstart[h_pt_orig[q]]);
  HostSpare<long long>::take(perm, (size_t)N);
  HostSpare<int>::take(h_oimg, (size_t)N);
  perm.resize(N);    // (every element is written by the pass below)
  uv_h.reset(new PinnedBuf<double2>(N));   // (uploaded by build(), asynchronously: they live until its final sync)
  opt_h.reset(new PinnedBuf<int>(N));
  PinnedBuf<double2>& uv = *uv_h;
  PinnedBuf<int>& opt_ = *opt_h;
  h_oimg.resize(N);
  if (streaming) {
    std::vector<int> cursor(h_pt_start.begin(), h_pt_start.end() - 1);
    for (long long k = 0; k < N; ++k) {
      const long long o = kept_at(k);
      const int q = pt_new[P->obs_point[o]], a = cursor[q]++;
      perm[a] = o;
      uv[a] = make_double2(P->obs_uv[2 * o], P->obs_uv[2 * o + 1]);
      h_oimg[a] = P->obs_image[o]; opt_[a] = q;
    }
  } else
  parallel_ranges(NP, [&](long long q0, long long q1) {
    for (long long q = q0; q < q1; ++q) {
      const int src = cstart[h_pt_orig[q]], cnt = h_pt_start[q + 1] - h_pt_start[q];
      for (int j = 0; j < cnt; ++j) {
        const int a = h_pt_start[q] + j;
        perm[a] = bucket[src + j];
        uv[a] = buv[src + j];
        h_oimg[a] = bimg[src + j]; opt_[a] = (int)q;
      }
    }
  });

  lap("point-major order");
  // ---- image-major view for the camera sweep ----
  im_uv_h.reset(new PinnedBuf<double2>(N));
  im_pt_h.reset(new PinnedBuf<int>(N));
  PinnedBuf<double2>& im_uv = *im_uv_h;
  PinnedBuf<int>& im_pt = *im_pt_h;
  counting_sort_parallel(N, NI, [&](long long a) { return h_oimg[a]; }, img_start,
                         [&](long long a, int at) { im_uv[at] = uv[a]; im_pt[at] = opt_[a]; });
  lap("image-major order");
}

void mavba_session::build(const mavba_problem* P, const DeviceRaw* raw) {
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
  if (!raw && NP > 0 && !P->points) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null point array");
  if (!raw && NO_all > 0 && (!P->obs_uv || !P->obs_image || !P->obs_point)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null observation arrays");
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
  // (the observation indices are checked below: on the host, or by the device set-up's counting kernel)
  if (P->num_rot_priors < 0) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "negative num_rot_priors");
  if (P->num_rot_priors > 0 && (!P->rot_prior_image || !P->rot_prior_rvec)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "null rotation-prior arrays");
  for (int q = 0; q < P->num_rot_priors; ++q)
    if (P->rot_prior_image[q] < 0 || P->rot_prior_image[q] >= NI) throw Failure(MAVBA_ERR_BAD_INDEX, "rot_prior_image out of range");

  h_poses0.assign(P->poses, P->poses + (size_t)NI * 6);
  h_intr0.assign(P->intrinsics, P->intrinsics + (size_t)NC * 9);
  if (!raw) {
    HostSpare<double>::take(h_points0, (size_t)NP * 3);
    h_points0.assign(P->points, P->points + (size_t)NP * 3);
  }
  h_pose_const.assign(NI, 0); h_intr_const_in.assign(NC, 0); h_pt_const_in.assign(NP, 0);
  if (P->pose_const) h_pose_const.assign(P->pose_const, P->pose_const + NI);
  if (P->intr_const) h_intr_const_in.assign(P->intr_const, P->intr_const + NC);
  if (P->point_const) h_pt_const_in.assign(P->point_const, P->point_const + NP);

  // Residual blocks without a free parameter block leave the program (ceres
  // RemoveFixedBlocksFromProgram); their cost is the fixed cost.
  h_pt_count_all.assign(NP, 0);
  h_dropped_rnorm.clear(); h_dropped_cost.clear(); h_pt_removed.clear();
  std::vector<long long> kept;
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
  // where the ordering block will run (device_setup.hip unless all-constant residual blocks are dropped - the host walks
  // the kept list then - or the problem is small); MAVBA_SETUP=device | host forces one implementation (tests, timing)
  const char* setup_env = std::getenv("MAVBA_SETUP");
  const int setup_mode = !setup_env ? 0 : (std::string(setup_env) == "device" ? 1 : (std::string(setup_env) == "host" ? 2 : 0));
  const bool device_order = raw || (all_kept && NO_all > 0 && NP > 0 && setup_mode != 2 && (setup_mode == 1 || NO_all >= 50000));
  if (raw && !(all_kept && NO_all > 0 && NP > 0)) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "device-resident problem with dropped residual blocks (or none at all)");
  if (!device_order) {
    int bad = 0;
    parallel_ranges(NO_all, [&](long long b0, long long b1) {
      int local = 0;
      for (long long o = b0; o < b1; ++o)
        local |= (P->obs_image[o] < 0) | (P->obs_image[o] >= NI) | (P->obs_point[o] < 0) | (P->obs_point[o] >= NP);
      if (local) __atomic_store_n(&bad, 1, __ATOMIC_RELAXED);
    });
    if (bad) throw Failure(MAVBA_ERR_BAD_INDEX, "observation index out of range");
  }
  if (all_kept) {
    // (the per-point counts and the used flags then fall out of the counting sorts below; `kept` stays empty = identity)
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
        // its raw residual never changes (every block is constant) but it still counts in the point's error
        // (problem.Evaluate covers all residual blocks, bundle_adjustment.cc:583-596)
        if (h_dropped_rnorm.empty()) { h_dropped_rnorm.assign(NP, 0.0); h_dropped_cost.assign(NP, 0.0); }
        h_dropped_rnorm[p] += std::sqrt(r[0] * r[0] + r[1] * r[1]);
        h_dropped_cost[p] += hr;
        continue;
      }
      kept.push_back(o);
      h_img_used[i] = 1; h_cam_used[c] = 1; h_pt_used[p] = 1;
    }
  }
  lap("validate + fixed blocks");
  N = all_kept ? (int)NO_all : (int)kept.size();
  const long long* keptp = all_kept ? nullptr : kept.data();
  Nstride = std::max(32, round_up(N, 32));
  NPs = std::max(32, round_up(NP, 32));

  // rotation priors: kept when the image's rvec block is free, sorted by image
  std::vector<std::pair<int, int>> pri;  // (image, index)
  h_prior_on_img.assign(NI, 0);
  fixed_cost_priors = 0.0;
  num_priors_all = P->num_rot_priors;
  for (int q = 0; q < P->num_rot_priors; ++q) {
    const int i = P->rot_prior_image[q];
    if (h_pose_const[i] & MAVBA_CONST_RVEC) {
      double R0[9], r, j[3];
      rot_matrix_colmajor(&P->rot_prior_rvec[3 * q], R0);
      rot_prior_eval(&h_poses0[(size_t)i * 6], R0, P->rot_prior_weight, r, j);
      fixed_cost += 0.5 * r * r;
      fixed_cost_priors += 0.5 * r * r;
      continue;
    }
    pri.push_back({i, q});
    h_img_used[i] = 1;
    h_prior_on_img[i] = 1;
  }
  std::stable_sort(pri.begin(), pri.end());
  num_priors = (int)pri.size();
  prior_weight = P->rot_prior_weight;
  num_residuals = 2 * NO_all + P->num_rot_priors;
  num_residuals_reduced = 2ll * N + num_priors;

  // ---- internal point order and the two observation orders: on the device (device_setup.hip) unless all-constant
  // residual blocks were dropped (the host walks the kept list then) or the problem is small ----
  // Small problems on the host path: nothing of the set-up below runs on the device, so its ~50 small uploads and clears are
  // collected and leave as ONE copy + ONE kernel before reset_state (upload_batch_begin, host_util.hip; MAVBA_UPLOAD_BATCH=0: off)
  struct BatchScope { bool open = false; ~BatchScope() { if (open) (void)upload_batch_end(false); } } batch;
  {
    const char* e = std::getenv("MAVBA_UPLOAD_BATCH");
    if (!device_order && N < 50000 && !(e && std::atoi(e) == 0)) batch.open = upload_batch_begin(st);
  }
  setup_batched = batch.open;
  std::vector<int> img_start;
  std::unique_ptr<PinnedBuf<double2>> uv_h, im_uv_h;   // host path: page-locked staging of the arrays uploaded below
  std::unique_ptr<PinnedBuf<int>> opt_h, im_pt_h;
  if (device_order) {
    order_on_device(P, img_start, raw);
    lap("order on device");
  } else {
    order_on_host(P, keptp, img_start, uv_h, opt_h, im_uv_h, im_pt_h);
    lap("order on host");
  }
  if (all_kept)
    for (int i = 0; i < NI; ++i)
      if (img_start[i + 1] > img_start[i]) { h_img_used[i] = 1; h_cam_used[h_img_cam[i]] = 1; }
  // observations per camera-sweep work-group: 2048 for large problems, down to 256 for small ones (a local window has
  // ~10 images: one work-group per image took 13 us of a 170 us iteration)
  const int kSweepChunk = std::min(2048, std::max(256, round_up(N / 512, 256)));
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
  if (!device_order) {
    d_uv.upload_pinned(uv_h->data(), (size_t)N, st); d_obs_img.upload(h_oimg, st); d_obs_pt.upload_pinned(opt_h->data(), (size_t)N, st); d_pt_start.upload(h_pt_start, st);
    d_im_uv.upload_pinned(im_uv_h->data(), (size_t)N, st); d_im_pt.upload_pinned(im_pt_h->data(), (size_t)N, st);
  }
  d_img_cam.upload(h_img_cam, st); d_cam_model.upload(h_cam_model, st);
  d_sweep_chunks.upload(sweep_chunks, st); d_img_chunk_start.upload(img_chunk_start, st);
  d_cam_img_start.upload(cam_img_start, st); d_cam_imgs.upload(cam_imgs, st);
  d_prior_img.upload(prior_img, st); d_prior_start.upload(prior_start, st); d_prior_R0.upload(prior_R0, st);
  d_pt_count.upload(h_pt_count_all, st);
  if (!device_order) { d_pt_orig.upload(h_pt_orig, st); d_points0.upload(h_points0, st); }
  d_pts_out.alloc((size_t)std::max(NP, 1) * 3);
  d_poses0.upload(h_poses0, st); d_intr0.upload(h_intr0, st);
  const size_t nI = std::max(NI, 1), nC = std::max(NC, 1), nP = std::max(NP, 1);
  d_poses.alloc(nI * 6); d_intr.alloc(nC * 9); d_points.alloc(nP * 3);
  d_cposes.alloc(nI * 6); d_cintr.alloc(nC * 9); d_cpoints.alloc(nP * 3);
  d_camrec.alloc(nI * 9); d_ccamrec.alloc(nI * 9);
  // (the Jacobian planes R / Jp / Jc / Jk are allocated on demand: ensure_planes - the solve itself is J-free)
  d_Cu.alloc((size_t)6 * NPs); d_gu.alloc((size_t)3 * NPs); d_Gi.alloc((size_t)6 * NPs); d_h.alloc((size_t)3 * NPs);
  d_Gi.zero(st); d_h.zero(st);  // (written per linear solve for the points that have observations; the others stay 0)
  d_scale_cam.alloc((size_t)n_pad); d_scale_pt.alloc((size_t)3 * NPs);
  d_scale_cam.zero(st); d_scale_pt.zero(st);
  d_sweep_partial.alloc((size_t)std::max(jacobian_sweep_grid(std::max(N, 1)), kFrontMaxGrid) + 8);
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
  d_norm_partial.alloc((size_t)(512 + kStateNormsCamBlocks) * 2); d_step_partial.alloc((size_t)(1024 + update_cameras_groups(NI) + 2) * 3);
  d_scal.alloc(SC_COUNT); d_scal.zero(st);
  d_rnorm.alloc(std::max(N, 1)); d_perr.alloc(nP);

  lap("alloc + upload");
  derive_free_flags();
  finish_structure();
  lap("finish_structure total");
  setup_batched = false;
  if (batch.open) { batch.open = false; HIP_OK(upload_batch_end(true)); }
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
// Elimination tree of the reduced camera system over the image graph (lower[r] = images c < r that share a point with
// r): nodes in elimination order, children before parents, the root last; empty = no dissection. Pure host code (no
// device): also exported for tests as mavba_debug_elimination_tree.
namespace mavba {
std::vector<ElimNode> elimination_tree(int NI, int NC, const std::vector<std::vector<int>>& lower, int forced, int max_depth) {
  std::vector<ElimNode> tn;
  typedef ElimNode TNode;
  {
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
      // recursive bisection: M (ascending) -> [A | B | S] with A and B uncoupled. Two candidates per set, the one with
      // the shorter chain max(A, B) + S wins (and a split must pay):
      //  * by acquisition order: S = the members of the second run that have a neighbour in the first (a flight strip is
      //    banded in that order); the cut is scanned;
      //  * by a level structure: breadth-first levels from a pseudo-peripheral image - edges only join adjacent
      //    levels, so one level (thinned to its members with a neighbour in the next level) separates what lies before
      //    from what lies behind; on a 2-D block of images this cuts ACROSS the short side, where the order-based cut
      //    can only take whole rows (C3: separators of 4 instead of 8 tile columns).
      std::vector<std::vector<int>> adj(NI);
      for (int r = 0; r < NI; ++r)
        for (int c : lower[r]) { adj[r].push_back(c); adj[c].push_back(r); }
      for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
      bool use_levels = true;
      if (const char* e = std::getenv("MAVBA_ND_LEVELS")) use_levels = std::atoi(e) != 0;
      std::vector<int> pos(NI, -1), level(NI, -1);
      std::function<std::pair<int, int>(std::vector<int>&&, int, int)> rec = [&](std::vector<int>&& M, int depth, int tl) {
        const int n = (int)M.size();
        const int leaf_tiles = tiles_of(6 * n + tl);
        auto make_leaf = [&]() { tn.push_back(TNode{std::move(M), -1}); return std::make_pair((int)tn.size() - 1, leaf_tiles); };
        if (depth >= max_depth || n < 32 || leaf_tiles < 6) return make_leaf();
        for (int t = 0; t < n; ++t) pos[M[t]] = t;
        // ---- candidate 1: cut of the acquisition order
        std::vector<int> mnp(n);
        for (int t = 0; t < n; ++t) {
          int m = t;
          for (int c : lower[M[t]]) if (pos[c] >= 0) m = std::min(m, pos[c]);
          mnp[t] = m;
        }
        int best = leaf_tiles, best_c = -1;
        for (int c = n / 4; c <= 3 * n / 4; c += std::max(1, n / 64)) {
          int ns = 0;
          for (int t = c; t < n; ++t) ns += mnp[t] < c;
          if (ns == 0 && tl == 0) continue;  // (a separator node needs at least one column)
          const int est = std::max(tiles_of(6 * c), tiles_of(6 * (n - c - ns))) + tiles_of(6 * ns + tl);
          if (est < best) { best = est; best_c = c; }
        }
        // ---- candidate 2: a level of a breadth-first level structure
        int best_lv = leaf_tiles, best_m = -1, nlev = 0;
        if (use_levels) {
          auto bfs = [&](int start) {  // levels of the sub-graph induced by M; a further component starts two levels on
            for (int t = 0; t < n; ++t) level[M[t]] = -1;
            std::vector<int> queue;
            queue.reserve(n);
            size_t head = 0;
            int seed = start, base = 0, scan = 0;
            for (;;) {
              level[seed] = base;
              queue.push_back(seed);
              while (head < queue.size()) {
                const int v = queue[head++];
                for (int w : adj[v]) if (pos[w] >= 0 && level[w] < 0) { level[w] = level[v] + 1; queue.push_back(w); }
              }
              if ((int)queue.size() == n) break;
              base = level[queue.back()] + 2;
              while (level[M[scan]] >= 0) ++scan;
              seed = M[scan];
            }
            return queue;
          };
          // pseudo-peripheral start: the lowest-degree image of the last level, twice
          int start = M[0];
          for (int round = 0; round < 2; ++round) {
            const std::vector<int> q = bfs(start);
            const int last = level[q.back()];
            int pick = -1;
            for (int v : q)
              if (level[v] == last && (pick < 0 || adj[v].size() < adj[pick].size() || (adj[v].size() == adj[pick].size() && v < pick))) pick = v;
            start = pick;
          }
          bfs(start);
          for (int t = 0; t < n; ++t) nlev = std::max(nlev, level[M[t]] + 1);
          // per level: size, and how many members have a neighbour in the NEXT level (the thinned separator)
          std::vector<int> lsize(nlev, 0), lsep(nlev, 0);
          for (int t = 0; t < n; ++t) {
            const int v = M[t];
            lsize[level[v]]++;
            bool fwd = false;
            for (int w : adj[v]) if (pos[w] >= 0 && level[w] == level[v] + 1) { fwd = true; break; }
            lsep[level[v]] += fwd;
          }
          int before = 0;
          for (int m = 0; m < nlev; ++m) {
            const int ns = lsep[m], na = before + lsize[m] - ns, nbh = n - before - lsize[m];
            before += lsize[m];
            if (na < 8 || nbh < 8 || (ns == 0 && tl == 0)) continue;
            const int est = std::max(tiles_of(6 * na), tiles_of(6 * nbh)) + tiles_of(6 * ns + tl);
            if (est < best_lv) { best_lv = est; best_m = m; }
          }
        }
        std::vector<int> A, B, S;
        if (best_m >= 0 && best_lv < best) {
          for (int t = 0; t < n; ++t) {
            const int v = M[t];
            if (level[v] < best_m) A.push_back(v);
            else if (level[v] > best_m) B.push_back(v);
            else {
              bool fwd = false;
              for (int w : adj[v]) if (pos[w] >= 0 && level[w] == best_m + 1) { fwd = true; break; }
              (fwd ? S : A).push_back(v);
            }
          }
          best = best_lv;
        } else if (best_c >= 0) {
          A.assign(M.begin(), M.begin() + best_c);
          for (int t = best_c; t < n; ++t) (mnp[t] < best_c ? S : B).push_back(M[t]);
        }
        for (int t = 0; t < n; ++t) pos[M[t]] = -1;
        if (A.empty() || best > leaf_tiles - std::max(2, leaf_tiles / 8)) return make_leaf();
        if (A.size() < 8 || B.size() < 8) return make_leaf();
        const auto ra = rec(std::move(A), depth + 1, 0);
        const auto rb = rec(std::move(B), depth + 1, 0);
        const int sep_tiles = tiles_of(6 * (int)S.size() + tl);
        tn.push_back(TNode{std::move(S), -1});
        const int me = (int)tn.size() - 1;
        tn[ra.first].parent = me; tn[rb.first].parent = me;
        return std::make_pair(me, std::max(ra.second, rb.second) + sep_tiles);
      };
      // Images without a neighbour (entirely constant poses: the datum image of a global BA) couple to nothing. As a
      // component of their own they would distort the level structures (C3: one branch stayed unsplit, 24 instead of 19
      // dependent panel steps), so they stay out of the bisection and join the leaf with the most padding to spare.
      std::vector<int> all, isolated;
      for (int i = 0; i < NI; ++i) (adj[i].empty() ? isolated : all).push_back(i);
      rec(std::move(all), 0, tail);
      if (tn.size() < 3) tn.clear();
      if (!tn.empty()) {
        std::vector<char> is_leaf(tn.size(), 1);
        for (const TNode& t : tn) if (t.parent >= 0) is_leaf[t.parent] = 0;
        for (int i : isolated) {
          int best = -1, best_slack = -1;
          for (size_t t = 0; t < tn.size(); ++t) {
            if (!is_leaf[t]) continue;
            const int cols = 6 * (int)tn[t].imgs.size();
            const int slack = std::max(64, (cols + 63) / 64 * 64) - cols;
            if (slack > best_slack) { best_slack = slack; best = (int)t; }
          }
          tn[best].imgs.push_back(i);
        }
        for (TNode& t : tn) std::sort(t.imgs.begin(), t.imgs.end());
      }
    }
  }
  return tn;
}
}  // namespace mavba

void mavba_session::choose_elimination_order(const std::vector<SchurBlock>& blocks) {
  const int tiles0 = std::max(1, round_up(n_full, 64) / 64);
  // Tree of image sets in elimination order (children before parents, root last); the root also carries the
  // intrinsics blocks. One node = no dissection.
  std::vector<ElimNode> tn;
  int forced = -1, max_depth = 2;
  if (const char* e = std::getenv("MAVBA_ND_PARTS")) forced = std::atoi(e);  // 0/1 = off, n = force n flat parts
  // Depth of the recursive bisection. The persistent factorisation has no barrier between tree levels, so what counts is
  // the longest root-to-leaf path and a third level pays (C3: 0.44 -> 0.41 ms, round 3). Round 3 kept depth 2 above 96 tile
  // columns because such systems ran the launch-per-panel schedule, whose per-level launches made depth 3 slower (C5: 3.46 ->
  // 3.58 ms); since round 4 they run the persistent launch too and the rule was never revisited - round 6,
  // profiles/r06_nd_depth_sweep.txt: C5 at depth 3 has 8 concurrent fronts instead of 4 and 4 104 envelope tiles instead of
  // 5 041, k_chol_persist 1.96 -> 1.53 ms, 186 -> 203 LM iterations/s (a fourth level changes nothing at C5, C3 or C2: a split
  // must pay). Depth 2 stays where the launch-per-panel schedule is certain (more than MAVBA_CHOL_PERSIST_MAX_NB = 1024 tile columns).
  {
    const char* e = std::getenv("MAVBA_CHOL_PERSIST_MAX_NB");
    if (tiles0 <= (e ? std::atoi(e) : 1024)) max_depth = 3;
  }
  if (const char* e = std::getenv("MAVBA_ND_DEPTH")) max_depth = std::atoi(e);
  const bool can_dissect = NI >= 16 && tiles0 >= 8 && forced != 0 && forced != 1 && max_depth > 0 && (world == 1 || NI <= 4096);
  if (can_dissect) {
    // image adjacency (lower: col < row) from the pose-pose blocks; with shards, the union over ranks
    std::vector<std::vector<int>> lower(NI);
    if (sharded()) {
      std::vector<double> a((size_t)NI * NI, 0.0);
      for (const SchurBlock& B : blocks)
        if (B.kind == BLK_PP && B.row_ent != B.col_ent) a[(size_t)std::max(B.row_ent, B.col_ent) * NI + std::min(B.row_ent, B.col_ent)] = 1.0;
      DevBuf<double> d;
      d.upload(a, st);
      allreduce(d.p, (long long)NI * NI, 1);
      download(a.data(), d.p, a.size() * 8);
      for (int r = 0; r < NI; ++r)
        for (int c = 0; c < r; ++c) if (a[(size_t)r * NI + c] != 0.0) lower[r].push_back(c);
    } else {
      for (const SchurBlock& B : blocks)
        if (B.kind == BLK_PP && B.row_ent != B.col_ent) lower[std::max(B.row_ent, B.col_ent)].push_back(std::min(B.row_ent, B.col_ent));
    }
    tn = elimination_tree(NI, NC, lower, forced, max_depth);
  }
  // column offsets in tree order (every node padded to whole tiles), intrinsics at the end of the root
  h_off_img.assign(NI, 0); h_off_cam.assign(NC, 0);
  std::vector<CholNode> tree;
  int col = 0, active_cols = -1;  // active_cols: columns before the first entirely constant block (-1: unknown order)
  if (tn.size() >= 3) {
    for (size_t t = 0; t < tn.size(); ++t) {
      const int begin = col;
      for (int i : tn[t].imgs) { h_off_img[i] = col; col += 6; }
      if (t + 1 == tn.size()) for (int c = 0; c < NC; ++c) { h_off_cam[c] = col; col += 9; }
      col = std::max(round_up(col, 64), begin + 64);
      tree.push_back(CholNode{begin / 64, col / 64, tn[t].parent});
    }
  } else {
    // No dissection (small systems): blocks with a free parameter first, entirely constant ones (unit diagonal, zero
    // row and right-hand side) last - the one-work-group solve of small systems then stops at the last tile column
    // that holds a free parameter (a 10-image window with 2 fixed images and fixed intrinsics: 1 tile instead of 2).
    // (With shards the free flags were agreed between the ranks before this runs: join_ranks.)
    auto img_free = [&](int i) { for (int e = 0; e < 6; ++e) if (h_pose_free[(size_t)i * 6 + e]) return true; return false; };
    auto cam_free = [&](int c) { for (int k = 0; k < 9; ++k) if (h_intr_free[(size_t)c * 9 + k]) return true; return false; };
    for (int i = 0; i < NI; ++i) if (img_free(i)) { h_off_img[i] = col; col += 6; }
    for (int c = 0; c < NC; ++c) if (cam_free(c)) { h_off_cam[c] = col; col += 9; }
    active_cols = col;
    for (int i = 0; i < NI; ++i) if (!img_free(i)) { h_off_img[i] = col; col += 6; }
    for (int c = 0; c < NC; ++c) if (!cam_free(c)) { h_off_cam[c] = col; col += 9; }
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
  if (sharded()) {
    // The matrix that gets factorised is the SUM over ranks: its structure is the union of the ranks'.
    std::vector<double> h(mark.begin(), mark.end());
    DevBuf<double> d;
    d.upload(h, st);
    allreduce(d.p, (long long)h.size(), 1);
    download(h.data(), d.p, h.size() * 8);
    for (size_t t = 0; t < h.size(); ++t) mark[t] = h[t] != 0.0;
  }
  std::vector<std::pair<int, int>> tile_pairs;
  for (int tr = 0; tr < nbt; ++tr)
    for (int tc = 0; tc <= tr; ++tc) if (mark[(size_t)tr * nbt + tc]) tile_pairs.emplace_back(tr, tc);
  HIP_OK(chol_struct.build(nbt, tile_pairs, tree, st));
  chol_struct.active_tiles = active_cols >= 0 ? std::max(1, (active_cols + 63) / 64) : nbt;
  nd_parts = chol_struct.nseg > 1 ? chol_struct.num_fronts_max : 0;
  if (sharded()) {
    // (slots of the tile store: diagonal tiles, then the structurally non-zero ones)
    std::vector<int> tl;
    std::vector<unsigned char> have((size_t)nbt * nbt, 0);
    auto slot_of = [&](int tr, int tc) {
      const int sl = chol_struct.tile_slot[(size_t)tr * nbt + tc];
      if (sl < 0) throw Failure(MAVBA_ERR_HIP, "a structurally non-zero tile lies outside the factorisation's envelope");
      return sl;
    };
    for (int t = 0; t < nbt; ++t) { tl.push_back(slot_of(t, t)); have[(size_t)t * nbt + t] = 1; }
    for (const auto& pr : tile_pairs)
      if (!have[(size_t)pr.first * nbt + pr.second]) { have[(size_t)pr.first * nbt + pr.second] = 1; tl.push_back(slot_of(pr.first, pr.second)); }
    num_ar_tiles = (int)tl.size();
    d_ar_tiles.upload(tl, st);
    d_ar_buf.alloc((size_t)num_ar_tiles * 4096 + n_mat);
  }
  // every block of S must land in a tile of the store
  for (const auto& pr : tile_pairs)
    if (chol_struct.tile_slot[(size_t)pr.first * nbt + pr.second] < 0)
      throw Failure(MAVBA_ERR_HIP, "a structurally non-zero tile lies outside the factorisation's envelope");
  d_M.alloc(chol_struct.store_doubles()); d_L.alloc(chol_struct.store_doubles());
  M_is_clean = false;  // (fresh memory: the first assembly clears it)
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

  // intrinsics entries: one per (free point, free camera that sees it), cameras ascending; two parallel passes
  // over the points (count, then fill at the scanned offsets)
  std::vector<int> q_start(NP + 1, 0), q_pt, q_cam;
  bool q_on_device = false;
  {
    bool any_cam_active = false;
    for (int c = 0; c < NC; ++c) any_cam_active |= cam_active[c] != 0;
    auto cams_of = [&](int p, int* out) {  // distinct active cameras of point p, ascending; returns their number (<= NC)
      int n = 0;
      for (int a = h_pt_start[p]; a < h_pt_start[p + 1]; ++a) {
        const int c = h_img_cam[h_oimg[a]];
        if (!cam_active[c]) continue;
        int k = 0;
        while (k < n && out[k] != c) ++k;
        if (k == n) out[n++] = c;
      }
      std::sort(out, out + n);
      return n;
    };
    // large problems: on the device, from the point-major arrays that are already there (MAVBA_SETUP=host keeps it here)
    const char* setup_env = std::getenv("MAVBA_SETUP");  // (read per session: the tests switch it)
    const bool q_host = setup_env && std::string(setup_env) == "host";
    q_on_device = any_cam_active && N >= 50000 && !q_host;
    if (q_on_device) {
      intr_entries_on_device(cam_active, q_start, q_cam);
    } else if (any_cam_active) {
      // (two passes that both walk the observations. Keeping the first pass's cameras in a 16 B / point side buffer saved
      // 0.3 ms here - and made the NEXT call's 48 MB upload from page-locked memory take 15-25 ms instead of 1 ms every
      // other time, reproducibly, A/B on one box; cause not understood, so the buffer is gone)
      parallel_ranges(NP, [&](long long p0, long long p1) {
        std::vector<int> tmp(std::max(NC, 1));
        for (long long p = p0; p < p1; ++p) q_start[p + 1] = h_pt_free[p] ? cams_of((int)p, tmp.data()) : 0;
      }, 20000);
      for (int p = 0; p < NP; ++p) q_start[p + 1] += q_start[p];
      q_pt.resize((size_t)q_start[NP]); q_cam.resize((size_t)q_start[NP]);
      parallel_ranges(NP, [&](long long p0, long long p1) {
        std::vector<int> tmp(std::max(NC, 1));
        for (long long p = p0; p < p1; ++p) {
          if (q_start[p + 1] == q_start[p]) continue;
          const int n = cams_of((int)p, tmp.data());
          for (int k = 0; k < n; ++k) { q_pt[(size_t)q_start[p] + k] = (int)p; q_cam[(size_t)q_start[p] + k] = tmp[k]; }
        }
      }, 20000);
    }
  }
  Q = (int)q_cam.size();
  if (!q_on_device) { d_q_start.upload(q_start, st); d_q_pt.upload(q_pt, st); d_q_cam.upload(q_cam, st); }
  d_Eintr.alloc((size_t)std::max(Q, 1) * kIntrRec);
  if (planes_ready) d_Wk.alloc((size_t)std::max(Q, 1) * 27);
  build_front_tiles(q_start);

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
  // The point order puts the points that can never be clustered (more than kTailObs observations, constant) behind all the
  // others (order_on_host / k_point_keys): clusters end where that tail begins.
  tail_begin = NP;
  while (tail_begin > 0 && (h_pt_start[tail_begin] - h_pt_start[tail_begin - 1] > kTailObs || h_pt_const_in[tail_begin - 1])) --tail_begin;
  const ClusterShape sh = cl_shape;
  const int kClTab = sh.tab(), kClTabPP = sh.tab_pp(), kClTabIP = sh.tab_ip(), kClTabII = sh.tab_ii();
  const int kClImages = sh.images, kClCams = sh.cams;
  // Round 4: the clusters of k_schur_rows. The number of ROWS of a cluster's entry matrix decides what a batch costs (80
  // rows = 10 images + 2 cameras + h: 15 tiles of matrix instructions; 96 rows: 21; 128 rows: 36), so clusters are formed
  // from runs of points with one image set by estimated cost (do_range_rows) instead of "until 16 images are full". The
  // point order keeps equal image sets together (the hash in the key). Chosen when the front end can run inside the
  // cluster kernel at all; otherwise the clusters of rounds 1-3 (k_point_front + k_schur_clusters).
  std::vector<int> cl_ni, cl_nc, cl_kr;  // image slots, camera slots, camera rows (the slots' model parameters) of every cluster
  auto cam_rows_of = [&](const std::vector<int>& cams) { int k = 0; for (int c : cams) k += model_k(h_cam_model[c]); return k; };
  const bool no_fuse_env = std::getenv("MAVBA_NO_FUSE") != nullptr;  // (read per session: the tests switch paths)
  auto build_clusters = [&](bool rows_mode) {
    clusters.clear(); cl_imgs.clear(); cl_cams.clear(); cl_ni.clear(); cl_nc.clear(); cl_kr.clear();
    std::fill(pt_mode.begin(), pt_mode.end(), 0);
    bool use_clusters = true;
    if (const char* e = std::getenv("MAVBA_CLUSTERS")) use_clusters = std::atoi(e) != 0;
    // 128 points per cluster amortise the per-cluster costs; small problems get smaller clusters so that there
    // are at least ~2 per CU (a cluster is one work-group; C2: 235 clusters of 128 would leave CUs idle)
    long long nfree = 0;
    for (int p = 0; p < NP; ++p) nfree += h_pt_free[p] != 0;
    // points per cluster: small problems still give every CU ~2 clusters; large ones up to 256 (fewer partials and emits:
    // C3 0.297 -> 0.283 ms for the cluster kernel, 0.037 -> 0.031 for the finalize pass)
    int kMaxPoints = (int)std::min<long long>(256, std::max<long long>(kClBatch, round_up((int)(nfree / 512), kClBatch)));
    // k_schur_rows: several 256-lane work-groups share a CU - about 1024 clusters fill the device; a cluster holds whole
    // 16-point batches
    // (a local window - up to 4096 free points - gets single-batch clusters: every cluster is one work-group on a CU of its own and
    // the kernel's time is the longest cluster's; 10-image window: 15.6 -> ~12 us, the call 2.16 -> 2.12 ms)
    if (rows_mode) kMaxPoints = (int)std::min<long long>(kRowsMaxPoints, std::max<long long>(nfree <= 4096 ? kRowsBatch : 2 * kRowsBatch, round_up((int)(nfree / 1024), kRowsBatch)));
    if (const char* e = std::getenv("MAVBA_CLUSTER_POINTS")) kMaxPoints = std::max(1, std::atoi(e));
    if (rows_mode) kMaxPoints = std::min(kMaxPoints, kRowsMaxPoints);
    // cost model of the run-based greedy (cycles of one CU: per cluster, per 16-point batch of each row class; measured on
    // C3, MAVBA_ROWS_COST="F,B0,B1,B2" overrides)
    double kCostF = 20000.0, kCostB[kRowsClasses] = {15400.0, 17500.0, 27500.0};  // (F also stands for what a cluster costs downstream: its block partials in the finalize pass - sweep in profiles/r04_cluster_cost_sweep.txt)
    if (const char* e = std::getenv("MAVBA_ROWS_COST")) std::sscanf(e, "%lf,%lf,%lf,%lf", &kCostF, &kCostB[0], &kCostB[1], &kCostB[2]);
    // Greedy over consecutive points, run independently on fixed ranges of points (NOT on "one range per
    // thread": the clusters - and with them the order in which partials are added - must not depend on the
    // machine's core count).
    const int kRange = 2048;
    const int nranges = (NP + kRange - 1) / kRange;
    std::vector<std::vector<SchurCluster>> r_clusters(nranges);
    std::vector<std::vector<int>> r_imgs(nranges), r_cams(nranges), r_ni(nranges), r_nc(nranges), r_kr(nranges);
    // Membership of an image in the open cluster / in the current point is an epoch tag per image (one table per host
    // thread's range, reused): a point costs its observations, not a set union (2.6 -> ~1 ms at C3). Same greedy rule as
    // before - the cluster closes when the UNION of its images and the point's would not fit - so the clusters are the same.
    auto do_range = [&](int rg, std::vector<int>& in_cluster, std::vector<int>& in_point) {
      const int r0 = rg * kRange, r_end = std::min(NP, r0 + kRange);
      // (tail points are generic by definition; no cluster's span reaches into the tail)
      for (int p = std::max(r0, tail_begin); p < r_end; ++p)
        if (h_pt_free[p] && (h_pt_start[p + 1] > h_pt_start[p] || q_start[p + 1] > q_start[p])) pt_mode[p] = 2;
      const int r1 = std::min(r_end, std::max(r0, tail_begin));
      std::vector<int> cur_i, cur_c, pi;
      // (epoch tags, never reset: cluster serials are unique over all ranges - a range closes at most kRange + 1 clusters -
      // and so are point indices)
      int cl_serial = rg * (kRange + 1);
      int cur_p0 = r0, cur_n = 0;
      auto close = [&](int p_end) {
        if (cur_n > 0) {
          r_ni[rg].push_back((int)cur_i.size()); r_nc[rg].push_back((int)cur_c.size()); r_kr[rg].push_back(cam_rows_of(cur_c));
          std::sort(cur_i.begin(), cur_i.end());
          std::sort(cur_c.begin(), cur_c.end());
          r_clusters[rg].push_back(SchurCluster{cur_p0, p_end});
          for (int k = 0; k < kClImages; ++k) r_imgs[rg].push_back(k < (int)cur_i.size() ? cur_i[k] : -1);
          for (int k = 0; k < kClCams; ++k) r_cams[rg].push_back(k < (int)cur_c.size() ? cur_c[k] : -1);
        }
        cur_i.clear(); cur_c.clear(); cur_n = 0; cur_p0 = p_end; ++cl_serial;
      };
      for (int p = r0; p < r1; ++p) {
        if (!h_pt_free[p]) continue;
        pi.clear();
        bool dup = false;
        for (int a = h_pt_start[p]; a < h_pt_start[p + 1]; ++a) {
          const int i = h_oimg[a];
          if (!img_active[i]) continue;
          if (in_point[i] == p) dup = true;
          in_point[i] = p;
          pi.push_back(i);
        }
        const int nq = q_start[p + 1] - q_start[p];
        if (pi.empty() && nq == 0) continue;
        if (!use_clusters || dup || (int)pi.size() > kClImages || nq > kClCams) { pt_mode[p] = 2; continue; }
        auto grown = [&](int& ni, int& nc) {  // sizes of the unions with the open cluster
          ni = (int)cur_i.size(); nc = (int)cur_c.size();
          for (int i : pi) ni += in_cluster[i] != cl_serial;
          for (int q = q_start[p]; q < q_start[p + 1]; ++q) nc += std::find(cur_c.begin(), cur_c.end(), q_cam[q]) == cur_c.end();
        };
        int ni, nc;
        grown(ni, nc);
        if (ni > kClImages || nc > kClCams || cur_n >= kMaxPoints || p - cur_p0 >= kClMaxBatches * kClBatch - 1) {
          close(p);
          grown(ni, nc);
        }
        if (cur_n == 0) cur_p0 = p;
        for (int i : pi) if (in_cluster[i] != cl_serial) { in_cluster[i] = cl_serial; cur_i.push_back(i); }
        for (int q = q_start[p]; q < q_start[p + 1]; ++q) if (std::find(cur_c.begin(), cur_c.end(), q_cam[q]) == cur_c.end()) cur_c.push_back(q_cam[q]);
        ++cur_n;
        pt_mode[p] = 1;
      }
      close(r1);
    };
    // k_schur_rows: the walk goes over RUNS of consecutive points with one image / camera set (the point order keeps them
    // together). A run is appended to the open cluster when that is cheaper - in estimated cycles - than a cluster of its own:
    // a batch costs more the more rows the cluster's entry matrix has (15 / 21 / 36 tiles), a cluster costs its tables and its
    // emit, and a batch that is not full costs a whole one.
    auto do_range_rows = [&](int rg, std::vector<int>& in_cluster, std::vector<int>& in_point) {
      const int r0 = rg * kRange, r_end = std::min(NP, r0 + kRange);
      for (int p = std::max(r0, tail_begin); p < r_end; ++p)
        if (h_pt_free[p] && (h_pt_start[p + 1] > h_pt_start[p] || q_start[p + 1] > q_start[p])) pt_mode[p] = 2;
      const int r1 = std::min(r_end, std::max(r0, tail_begin));
      // (a run's image / camera lists live in two flat arrays of the range: one heap block per list was 80 000 allocations on 16
      // threads at C3)
      struct Run {
        int p0, p1, n, io, ni, co, nc;
        const std::vector<int>* fi; const std::vector<int>* fc;
        struct Span { const int* b; const int* e; const int* begin() const { return b; } const int* end() const { return e; } size_t size() const { return (size_t)(e - b); } };
        Span imgs_span() const { return Span{fi->data() + io, fi->data() + io + ni}; }
        Span cams_span() const { return Span{fc->data() + co, fc->data() + co + nc}; }
      };
      std::vector<Run> runs;
      std::vector<int> flat_i, flat_c;
      flat_i.reserve((size_t)4 * kRange); flat_c.reserve((size_t)kRange);
      std::vector<int> pi, pc;
      for (int p = r0; p < r1; ++p) {
        if (!h_pt_free[p]) continue;
        pi.clear(); pc.clear();
        bool dup = false;
        for (int a = h_pt_start[p]; a < h_pt_start[p + 1]; ++a) {
          const int i = h_oimg[a];
          if (!img_active[i]) continue;
          if (in_point[i] == p) dup = true;
          in_point[i] = p;
          pi.push_back(i);
        }
        const int nq = q_start[p + 1] - q_start[p];
        if (pi.empty() && nq == 0) continue;
        if (!use_clusters || dup || (int)pi.size() > kClImages || nq > kClCams) { pt_mode[p] = 2; continue; }
        std::sort(pi.begin(), pi.end());
        for (int q = q_start[p]; q < q_start[p + 1]; ++q) pc.push_back(q_cam[q]);  // (ascending)
        pt_mode[p] = 1;
        bool same_set = !runs.empty() && runs.back().ni == (int)pi.size() && runs.back().nc == (int)pc.size();
        if (same_set) {
          const Run& B = runs.back();
          same_set = std::equal(pi.begin(), pi.end(), flat_i.begin() + B.io) && std::equal(pc.begin(), pc.end(), flat_c.begin() + B.co);
        }
        if (same_set) { runs.back().p1 = p + 1; runs.back().n++; }
        else {
          runs.push_back(Run{p, p + 1, 1, (int)flat_i.size(), (int)pi.size(), (int)flat_c.size(), (int)pc.size(), &flat_i, &flat_c});
          flat_i.insert(flat_i.end(), pi.begin(), pi.end());
          flat_c.insert(flat_c.end(), pc.begin(), pc.end());
        }
      }
      auto batches = [](int n) { return (n + kRowsBatch - 1) / kRowsBatch; };
      auto cost = [&](int n, int ni, int kr) { return kCostF + batches(n) * kCostB[rows_class_of_rows(rows_count(ni, kr))]; };
      std::vector<int> cur_i, cur_c;
      int cl_serial = rg * (kRange + 1), cur_p0 = r0, cur_n = 0;
      auto close = [&](int p_end) {
        if (cur_n > 0) {
          r_ni[rg].push_back((int)cur_i.size()); r_nc[rg].push_back((int)cur_c.size()); r_kr[rg].push_back(cam_rows_of(cur_c));
          std::sort(cur_i.begin(), cur_i.end());
          std::sort(cur_c.begin(), cur_c.end());
          r_clusters[rg].push_back(SchurCluster{cur_p0, p_end});
          for (int k = 0; k < kClImages; ++k) r_imgs[rg].push_back(k < (int)cur_i.size() ? cur_i[k] : -1);
          for (int k = 0; k < kClCams; ++k) r_cams[rg].push_back(k < (int)cur_c.size() ? cur_c[k] : -1);
        }
        cur_i.clear(); cur_c.clear(); cur_n = 0; cur_p0 = p_end; ++cl_serial;
      };
      auto cam_rows_span = [&](const Run::Span& cams) { int k = 0; for (int c : cams) k += model_k(h_cam_model[c]); return k; };
      auto add = [&](const Run& R, int p_begin, int n) {
        if (cur_n == 0) cur_p0 = p_begin;
        for (int i : R.imgs_span()) if (in_cluster[i] != cl_serial) { in_cluster[i] = cl_serial; cur_i.push_back(i); }
        for (int c : R.cams_span()) if (std::find(cur_c.begin(), cur_c.end(), c) == cur_c.end()) cur_c.push_back(c);
        cur_n += n;
      };
      for (const Run& R : runs) {
        const int rni = R.ni;
        bool join = false;
        if (cur_n > 0 && cur_n + R.n <= kMaxPoints && R.p1 - cur_p0 <= kRowsMaxPoints) {
          int ni = (int)cur_i.size(), nc = (int)cur_c.size(), kr = cam_rows_of(cur_c);
          for (int i : R.imgs_span()) ni += in_cluster[i] != cl_serial;
          for (int c : R.cams_span()) if (std::find(cur_c.begin(), cur_c.end(), c) == cur_c.end()) { ++nc; kr += model_k(h_cam_model[c]); }
          join = ni <= kClImages && nc <= kClCams &&
                 cost(cur_n + R.n, ni, kr) <= cost(cur_n, (int)cur_i.size(), cam_rows_of(cur_c)) + cost(R.n, rni, cam_rows_span(R.cams_span()));
        }
        if (join) { add(R, R.p0, R.n); continue; }
        close(R.p0);
        if (R.n <= kMaxPoints && R.p1 - R.p0 <= kRowsMaxPoints) { add(R, R.p0, R.n); continue; }
        // a run longer than a cluster: equal parts of whole batches (a part's SPAN - the run may contain skipped points -
        // must fit the kernel's table of point starts as well)
        const int parts = (R.n + kMaxPoints - 1) / kMaxPoints;
        const int per = std::min(kMaxPoints, round_up((R.n + parts - 1) / parts, kRowsBatch));
        int p = R.p0, left = R.n;
        while (left > 0) {
          int cnt = 0, q = p;
          while (cnt < per && q < R.p1 && q - p < kRowsMaxPoints) { cnt += pt_mode[q] == 1; ++q; }
          add(R, p, cnt);
          left -= cnt;
          if (left > 0) close(q);
          p = q;
        }
      }
      close(r1);
    };
    parallel_ranges(nranges, [&](long long g0, long long g1) {
      std::vector<int> in_cluster(NI, -1), in_point(NI, -1);
      for (long long g = g0; g < g1; ++g) {
        if (rows_mode) do_range_rows((int)g, in_cluster, in_point);
        else do_range((int)g, in_cluster, in_point);
      }
    }, 8);  // (a handful of ranges: waking the worker threads costs more than the walk)
    for (int rg = 0; rg < nranges; ++rg) {
      clusters.insert(clusters.end(), r_clusters[rg].begin(), r_clusters[rg].end());
      cl_imgs.insert(cl_imgs.end(), r_imgs[rg].begin(), r_imgs[rg].end());
      cl_cams.insert(cl_cams.end(), r_cams[rg].begin(), r_cams[rg].end());
      cl_ni.insert(cl_ni.end(), r_ni[rg].begin(), r_ni[rg].end());
      cl_nc.insert(cl_nc.end(), r_nc[rg].begin(), r_nc[rg].end());
      cl_kr.insert(cl_kr.end(), r_kr[rg].begin(), r_kr[rg].end());
    }
  };
  // can the front end run inside the cluster kernel? every observed point before the tail clustered, nothing else clustered
  auto fusable = [&]() {
    long long head_observed = 0, head_clustered = 0, all_clustered = 0;
    for (int p = 0; p < tail_begin; ++p) { head_observed += h_pt_start[p + 1] > h_pt_start[p]; head_clustered += pt_mode[p] == 1; }
    for (int p = 0; p < NP; ++p) all_clustered += pt_mode[p] == 1;
    return front_ok && !no_fuse_env && cl_shape.images == 16 && !clusters.empty() && head_clustered == head_observed &&
           all_clustered == head_clustered && (long long)clusters.size() <= kFrontMaxGrid;
  };
  rows_ok = false;
  if (front_ok && !no_fuse_env && cl_shape.images == 16) {
    build_clusters(true);
    rows_ok = fusable();
  }
  if (!rows_ok) build_clusters(false);
  num_clusters = (int)clusters.size();
  lap("  clusters: greedy");
  cluster_flops = 0.0;
  for (size_t c = 0; c < clusters.size(); ++c) {  // batches x k-steps x lower tiles x 2*16*16*4
    const int np = clusters[c].p1 - clusters[c].p0;
    if (rows_ok) {
      const int nt = kRowsClassNT[rows_class_of_rows(rows_count(cl_ni[c], cl_kr[c]))];
      cluster_flops += (double)((np + kRowsBatch - 1) / kRowsBatch) * (3 * kRowsBatch / 4) * (double)(nt * (nt + 1) / 2) * 2048.0;
    } else {
      cluster_flops += (double)((np + kClBatch - 1) / kClBatch) * (3 * kClBatch / 4) * (double)((sh.rows() / 16) * (sh.rows() / 16 + 1) / 2) * 2048.0;
    }
  }
  // local indices of every clustered observation / intrinsics entry, and which blocks a cluster touches
  std::vector<unsigned char> cl_present((size_t)std::max(num_clusters, 1) * kClTab, 0);
  // k_schur_rows, round 6: the lane every observation of a point takes in the point's 16-lane row. In a cluster with two camera
  // slots the observations of slot 0 go to lanes 0-7 and those of slot 1 to lanes 8-15 (observations of cameras with constant
  // intrinsics fill what is left), so that the kernel's reduce-scatter of the intrinsics rows starts inside the halves. A point
  // with more than 8 observations of one slot cannot be placed: its cluster is flagged (kRowsUnplaced) and keeps the selects.
  std::vector<unsigned long long> rows_lanes;
  std::vector<unsigned char> cl_unplaced((size_t)std::max(num_clusters, 1), 0);
  if (rows_ok) rows_lanes.assign((size_t)std::max(NP, 1), kRowsLanesIdentity);
  parallel_ranges(num_clusters, [&](long long c0, long long c1) {
    int loc[kClImagesMax], prev_loc[kClImagesMax];
    std::vector<int> slot_of(NI, 0);  // image -> its place in the cluster's list (only read for the cluster's own images)
    for (long long cl = c0; cl < c1; ++cl) {
      const int* imgs = &cl_imgs[(size_t)cl * kClImages];
      const int* cams = &cl_cams[(size_t)cl * kClCams];
      int ni = 0, nc = 0;
      while (ni < kClImages && imgs[ni] >= 0) { slot_of[imgs[ni]] = ni; ++ni; }
      while (nc < kClCams && cams[nc] >= 0) ++nc;
      unsigned char* pres = &cl_present[(size_t)cl * kClTab];
      int prev_n = -1;
      if (rows_ok && nc == 2) {
        for (int p = clusters[cl].p0; p < clusters[cl].p1; ++p) {
          const int ob = h_pt_start[p], cnt = h_pt_start[p + 1] - ob;
          if (cnt > kRowsBatch) { cl_unplaced[cl] = 1; continue; }
          int half_n[2] = {0, 0}, lane_of[kRowsBatch];
          bool fits = true;
          for (int a = 0; a < cnt; ++a) {  // observations of the two slots first, in their order
            const int c = h_img_cam[h_oimg[ob + a]];
            const int lc = c == cams[0] ? 0 : c == cams[1] ? 1 : -1;
            lane_of[a] = -1;
            if (lc < 0) continue;
            if (half_n[lc] == 8) { fits = false; break; }
            lane_of[a] = 8 * lc + half_n[lc]++;
          }
          if (!fits) { cl_unplaced[cl] = 1; continue; }
          unsigned used = 0;
          for (int a = 0; a < cnt; ++a) if (lane_of[a] >= 0) used |= 1u << lane_of[a];
          for (int a = 0, l = 0; a < cnt; ++a) {  // the others: the free lanes, ascending
            if (lane_of[a] >= 0) continue;
            while (used & (1u << l)) ++l;
            lane_of[a] = l; used |= 1u << l;
          }
          unsigned long long lm = cnt < kRowsBatch ? ~0ull : 0ull;  // (nibble 15 = no observation; a point with 16 has none empty)
          for (int a = 0; a < cnt; ++a) lm = (lm & ~(15ull << (4 * lane_of[a]))) | ((unsigned long long)a << (4 * lane_of[a]));
          rows_lanes[p] = lm;
        }
      }
      for (int p = clusters[cl].p0; p < clusters[cl].p1; ++p) {
        if (pt_mode[p] != 1) continue;
        int n = 0;
        for (int a = h_pt_start[p]; a < h_pt_start[p + 1]; ++a) {
          if (!img_active[h_oimg[a]]) continue;
          const int l = slot_of[h_oimg[a]];
          obs_meta[a] = (unsigned short)(l << 8 | ((p - clusters[cl].p0) % kClBatch));
          loc[n++] = l;
        }
        // (neighbours in the point order usually see the same images: their block pairs are already marked)
        const bool same = n == prev_n && std::equal(loc, loc + n, prev_loc);
        if (!same) {
          for (int x = 0; x < n; ++x)
            for (int y = 0; y < n; ++y)
              if (loc[x] >= loc[y]) pres[kClTabPP + loc[x] * (loc[x] + 1) / 2 + loc[y]] = 1;
          std::copy(loc, loc + n, prev_loc);
          prev_n = n;
        }
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
  }, 512);
  clustered_points = 0;
  for (int p = 0; p < NP; ++p) clustered_points += pt_mode[p] == 1;
  if (std::getenv("MAVBA_CLUSTER_STATS")) {  // debugging aid: how many 16-row blocks of the pose rows a 32-point batch really touches
    long long hist[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tiles_needed = 0, tiles_all = 0, obs_hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int cl = 0; cl < num_clusters; ++cl)
      for (int b0 = clusters[cl].p0; b0 < clusters[cl].p1; b0 += kClBatch) {
        unsigned mask = 0;
        const int b1 = std::min(b0 + kClBatch, clusters[cl].p1);
        for (int p = b0; p < b1; ++p)
          for (int a = h_pt_start[p]; a < h_pt_start[p + 1]; ++a)
            if (obs_meta[a] != 0xFFFFu) { const int l = obs_meta[a] >> 8; mask |= 1u << (6 * l / 16); mask |= 1u << ((6 * l + 5) / 16); }
        const int nb = __builtin_popcount(mask);
        hist[std::min(nb, 7)]++;
        const int nt = nb + 2;  // + the two row blocks of the camera / h rows
        tiles_needed += nt * (nt + 1) / 2; tiles_all += 36;
        obs_hist[std::min(8, (h_pt_start[b1] - h_pt_start[b0]) / 64)]++;
      }
    std::fprintf(stderr, "[cluster stats] batches by touched pose row blocks:");
    for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %d:%lld", i, hist[i]);
    std::fprintf(stderr, "  tiles needed %lld of %lld; batches by observations/64:", tiles_needed, tiles_all);
    for (int i = 0; i < 9; ++i) std::fprintf(stderr, " %d:%lld", i, obs_hist[i]);
    std::fprintf(stderr, "\n");
  }
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
  // (global BA of short tracks: every point sits in a cluster and there is nothing to enumerate - one thread, and the
  // per-thread count tables below stay empty)
  long long generic_points = 0;
  for (int p = 0; p < NP; ++p) generic_points += pt_mode[p] == 2;
  int T = host_threads();
  if (N < 50000 || generic_points == 0) T = 1;
  while (T > 1 && (size_t)T * nkeys_tot > (size_t)48 << 20) T /= 2;
  // contiguous point ranges of equal TERM work (a generic point with n observations makes ~n^2 / 2 terms; the generic points
  // sit together at the end of the point order, so ranges of equal observation counts would leave all of it to one thread)
  std::vector<int> range(T + 1, NP);
  range[0] = 0;
  if (T > 1) {
    std::vector<double> work((size_t)NP + 1, 0.0);
    for (int p = 0; p < NP; ++p) {
      const double n = pt_mode[p] == 2 ? (double)(h_pt_start[p + 1] - h_pt_start[p] + q_start[p + 1] - q_start[p]) : 0.0;
      work[p + 1] = work[p] + 0.5 * n * (n + 1.0);
    }
    for (int t = 1; t < T; ++t) {
      const double target = work[NP] * t / T;
      range[t] = (int)(std::lower_bound(work.begin(), work.end(), target) - work.begin());
      range[t] = std::max(range[t - 1], std::min(range[t], NP));
    }
  }
  std::vector<std::vector<int>> tcount(T * 3);
  auto run_threads = [&](const std::function<void(int)>& body) { host_run(T, body); };
  if (generic_points > 0)
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
    if (generic_points > 0) {
      // (dense tables of NI x NI keys per thread: at C5 - 4 M keys, 8 threads - summing them on one thread was ~15 ms of the set-up)
      std::mutex tot_m;
      parallel_ranges((long long)nkeys[k], [&](long long b0, long long b1) {
        long long sum = 0;
        for (long long key = b0; key < b1; ++key) {
          int c = 0;
          for (int t = 0; t < T; ++t) c += tcount[t * 3 + k][(size_t)key];
          count[k][(size_t)key] = c;
          sum += c;
        }
        std::lock_guard<std::mutex> g(tot_m);
        tot[k] += sum;
      }, 1 << 18);
    }
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
  // slot of a cluster's table -> the local pair it stands for (looked up ~300 000 times per set-up at C3)
  std::vector<unsigned char> slot_a(kClTab), slot_b(kClTab);
  for (int la = 0, sl = kClTabPP; la < kClImages; ++la) for (int lb = 0; lb <= la; ++lb, ++sl) { slot_a[sl] = (unsigned char)la; slot_b[sl] = (unsigned char)lb; }
  for (int sl = kClTabIP; sl < kClTabII; ++sl) { slot_a[sl] = (unsigned char)((sl - kClTabIP) / kClImages); slot_b[sl] = (unsigned char)((sl - kClTabIP) % kClImages); }
  for (int lc = 0, sl = kClTabII; lc < kClCams; ++lc) for (int ld = 0; ld <= lc; ++ld, ++sl) { slot_a[sl] = (unsigned char)lc; slot_b[sl] = (unsigned char)ld; }
  auto cluster_key = [&](long long cl, int slot, int& kind) -> size_t {
    const int* imgs = &cl_imgs[(size_t)cl * kClImages];
    const int* cams = &cl_cams[(size_t)cl * kClCams];
    if (slot < kClTabIP) { kind = BLK_PP; return (size_t)imgs[slot_a[slot]] * NI + imgs[slot_b[slot]]; }
    if (slot < kClTabII) { kind = BLK_IP; return (size_t)cams[slot_a[slot]] * NI + imgs[slot_b[slot]]; }
    kind = BLK_II;
    return (size_t)cams[slot_a[slot]] * NC + cams[slot_b[slot]];
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
    {
      // (ascending keys; fixed slices gathered by the host threads and joined in order)
      const long long nk = (long long)count[k].size();
      const int slices = nk >= (1 << 18) ? 64 : 1;
      std::vector<std::vector<size_t>> part(slices);
      parallel_ranges(slices, [&](long long s0, long long s1) {
        for (long long sl = s0; sl < s1; ++sl) {
          const long long b0 = nk * sl / slices, b1 = nk * (sl + 1) / slices;
          for (long long key = b0; key < b1; ++key)
            if (count[k][(size_t)key] != 0 || mandatory[k][(size_t)key] || cref[k][(size_t)key] != 0) part[sl].push_back((size_t)key);
        }
      }, 2);
      size_t total = 0;
      for (const auto& v : part) total += v.size();
      keys.reserve(total);
      for (const auto& v : part) keys.insert(keys.end(), v.begin(), v.end());
    }
    if (k == BLK_PP || k == BLK_IP) {
      // (the order as one precomputed integer per block: the comparator with its divisions was 1 ms of the C3 set-up)
      const unsigned long long nc = (unsigned long long)ncols[k];
      std::vector<std::pair<unsigned long long, size_t>> ranked(keys.size());
      for (size_t q = 0; q < keys.size(); ++q) {
        const unsigned long long a = keys[q], ia = a / nc, ja = a % nc;
        // pose-pose: (row tile, column tile, key); intrinsics-pose: (image, position in the ascending key list)
        ranked[q] = {k == BLK_PP ? ((ia / kTile) << 42 | (ja / kTile) << 21 | 0) : ja, q};
      }
      std::stable_sort(ranked.begin(), ranked.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
      std::vector<size_t> sorted(keys.size());
      for (size_t q = 0; q < keys.size(); ++q) sorted[q] = keys[ranked[q].second];
      keys.swap(sorted);
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
  // long partial runs are pre-reduced in groups of 32 into extra slots; the block then points at those (launch-bound
  // problems - a local window has ~80 clusters - keep runs of up to 256 for the finalize pass itself: one launch less)
  std::vector<PartialReduce> reduce_tasks;
  const int kPreReduceFrom = N < 200000 ? 256 : 64;  // (flat between 64 and 256 at C2 / C3: profiles/r06_ab_small_knobs.txt, item 3)
  for (SchurBlock& B : blocks) {
    const int n = B.chunk_end - B.chunk_begin;
    if (n <= kPreReduceFrom) continue;
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
  if (generic_points > 0) {
    for (int k = 0; k < 3; ++k)
      parallel_ranges((long long)nkeys[k], [&](long long b0, long long b1) {
        for (long long key = b0; key < b1; ++key) {
          int run = cursor[k][(size_t)key];
          for (int t = 0; t < T; ++t) { const int c = tcount[t * 3 + k][(size_t)key]; tcount[t * 3 + k][(size_t)key] = run; run += c; }
        }
      }, 1 << 18);
    run_threads([&](int t) {
      enumerate(range[t], range[t + 1], [&](int kind, int r, int c, int x, int y) {
        terms[kind][(size_t)tcount[t * 3 + kind][(size_t)r * ncols[kind] + c]++] = make_int2(x, y);
      });
    });
  }
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
    if (tot[k]) HIP_OK(copy_h2d_staged(d_terms[k].p, terms[k].get(), (size_t)tot[k] * sizeof(int2), st));
    d_part[k].alloc((size_t)std::max(num_slots[k], 1) * schur_partial_stride(k));
  }
  {
    std::vector<unsigned char> ptc(std::max(NP, 1), 0);
    for (int p = 0; p < NP; ++p) ptc[p] = pt_mode[p] == 1;
    // launch order: longest clusters first (work-groups are handed out in index order, so the tail of the launch is
    // made of short clusters); the slot tables travel with their clusters, the partial order inside a block is
    // the slot order and does not change
    std::vector<int> order(num_clusters);
    for (int c = 0; c < num_clusters; ++c) order[c] = c;
    auto row_class = [&](int c) { return rows_ok ? rows_class_of_rows(rows_count(cl_ni[c], cl_kr[c])) : 0; };
    // (k_schur_rows: longest = most matrix instructions first - batches x tiles of the cluster's row class)
    auto weight = [&](int c) {
      const int np = clusters[c].p1 - clusters[c].p0;
      if (!rows_ok) return (long long)np;
      const int nt = kRowsClassNT[row_class(c)];
      return (long long)((np + kRowsBatch - 1) / kRowsBatch) * (60 + nt * (nt + 1) / 2);
    };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weight(a) > weight(b); });
    if (rows_ok) {
      std::vector<SchurRowsCluster> rc(clusters.size());
      for (int k = 0; k < kRowsClasses; ++k) rows_class_count[k] = 0;
      rows_generic = false;
      for (int c = 0; c < num_clusters; ++c) rows_generic = rows_generic || cl_nc[c] > 2;
      d_rows_lanes.upload(rows_lanes, st);
      // the emit maps of the shapes present: one per (image slots, camera slots' rows), concatenated
      std::vector<unsigned> emit_all, e0, e1;
      std::map<std::pair<int, int>, std::array<int, 3>> emit_of;
      for (int c = 0; c < num_clusters; ++c) {
        const int o = order[c];
        int kslot[kClCamsMax] = {0, 0, 0};
        for (int k = 0; k < cl_nc[o] && k < kClCamsMax; ++k) kslot[k] = model_k(h_cam_model[cl_cams[(size_t)o * kClCams + k]]);  // (slot order = the cluster's sorted camera list)
        const int flags = rows_pack_flags(cl_unplaced[o] != 0, kslot, cl_nc[o]);
        auto it = emit_of.find({cl_ni[o], flags >> 8});
        if (it == emit_of.end()) {
          rows_emit_map(cl_ni[o], cl_nc[o], flags, e0, e1);
          it = emit_of.emplace(std::make_pair(cl_ni[o], flags >> 8), std::array<int, 3>{(int)emit_all.size(), (int)e0.size(), (int)e1.size()}).first;
          emit_all.insert(emit_all.end(), e0.begin(), e0.end());
          emit_all.insert(emit_all.end(), e1.begin(), e1.end());
        }
        rc[c] = SchurRowsCluster{clusters[o].p0, clusters[o].p1, cl_ni[o], cl_nc[o], it->second[0], {it->second[1], it->second[2]}, flags};
        rows_class_count[row_class(o)]++;
      }
      d_rows_clusters.upload(rc, st);
      d_rows_emit.upload(emit_all, st);
      std::vector<int> rl((size_t)std::max(num_clusters, 1) * kRowsLists, -1);
      for (int c = 0; c < num_clusters; ++c) {
        const int o = order[c];
        int* L = &rl[(size_t)c * kRowsLists];
        for (int k = 0; k < kClImages; ++k) { const int img = cl_imgs[(size_t)o * kClImages + k]; L[k] = img; L[kRowsListsCam + k] = img >= 0 ? h_img_cam[img] : -1; }
        for (int k = 0; k < kClCams; ++k) L[kRowsListsCams + k] = cl_cams[(size_t)o * kClCams + k];
      }
      d_rows_lists.upload(rl, st);
      if (std::getenv("MAVBA_CLUSTER_STATS")) {
        long long pts[kRowsClasses] = {}, bat[kRowsClasses] = {};
        for (int c = 0; c < num_clusters; ++c) { const int k = row_class(order[c]); pts[k] += rc[c].p1 - rc[c].p0; bat[k] += (rc[c].p1 - rc[c].p0 + kRowsBatch - 1) / kRowsBatch; }
        for (int k = 0; k < kRowsClasses; ++k)
          std::fprintf(stderr, "[cluster stats] k_schur_rows class %d (%d rows): %d clusters, %lld points, %lld batches\n", k, 16 * kRowsClassNT[k], rows_class_count[k], pts[k], bat[k]);
        long long hist[kClImagesMax + 1][kClCamsMax + 1][2] = {}, unplaced = 0;
        for (int c = 0; c < num_clusters; ++c) { hist[rc[c].ni][rc[c].nc][0]++; hist[rc[c].ni][rc[c].nc][1] += (rc[c].p1 - rc[c].p0 + kRowsBatch - 1) / kRowsBatch; unplaced += (rc[c].flags & kRowsUnplaced) != 0; }
        std::fprintf(stderr, "[cluster stats] k_schur_rows clusters / batches by (images, cameras):");
        for (int i = 0; i <= kClImagesMax; ++i) for (int k = 0; k <= kClCamsMax; ++k) if (hist[i][k][0]) std::fprintf(stderr, " (%d,%d) %lld/%lld", i, k, hist[i][k][0], hist[i][k][1]);
        std::fprintf(stderr, "; clusters that keep the selects: %lld\n", unplaced);
      }
    }
    std::vector<SchurCluster> cl_sorted(clusters.size());
    std::vector<int> tab_sorted(cl_tab.size(), -1);
    std::vector<int> lists_sorted((size_t)std::max(num_clusters, 1) * (kClImages + kClCams), -1);  // images, then cameras of a cluster
    for (int c = 0; c < num_clusters; ++c) {
      cl_sorted[c] = clusters[order[c]];
      std::copy(cl_tab.begin() + (size_t)order[c] * kClTab, cl_tab.begin() + (size_t)(order[c] + 1) * kClTab, tab_sorted.begin() + (size_t)c * kClTab);
    }
    for (int c = 0; c < num_clusters; ++c) {
      std::copy(cl_imgs.begin() + (size_t)order[c] * kClImages, cl_imgs.begin() + (size_t)(order[c] + 1) * kClImages, lists_sorted.begin() + (size_t)c * (kClImages + kClCams));
      std::copy(cl_cams.begin() + (size_t)order[c] * kClCams, cl_cams.begin() + (size_t)(order[c] + 1) * kClCams, lists_sorted.begin() + (size_t)c * (kClImages + kClCams) + kClImages);
    }
    d_clusters.upload(cl_sorted, st); d_cl_tab.upload(tab_sorted, st); d_cl_lists.upload(lists_sorted, st);
    d_obs_meta.upload(obs_meta, st); d_q_meta.upload(q_meta, st); d_pt_clustered.upload(ptc, st);
  }
  // the front end can run inside the cluster kernel when every observed point is clustered (no generic term lists, no
  // constant points with observations) and the clusters have the 16 x 3 shape
  {
    // If every observed point BEFORE the tail is clustered, the clusters take the fused kernel and the tail - with the generic
    // term lists of its free points - the separate front end, on tiles of its own.
    long long head_observed = 0, head_clustered = 0;
    for (int p = 0; p < tail_begin; ++p) { head_observed += h_pt_start[p + 1] > h_pt_start[p]; head_clustered += pt_mode[p] == 1; }
    const bool no_fuse = std::getenv("MAVBA_NO_FUSE") != nullptr;  // (read per session: the tests switch paths)
    fused_ok = front_ok && !no_fuse && cl_shape.images == 16 && num_clusters > 0 && head_clustered == head_observed &&
               clustered_points == head_clustered && num_clusters <= kFrontMaxGrid;
    num_tail_tiles = 0;
    if (fused_ok && h_pt_start[NP] > h_pt_start[tail_begin]) {
      build_tiles(q_start, tail_begin, d_tail_tiles, num_tail_tiles);
      if ((long long)num_clusters + point_front_grid(num_tail_tiles) > kFrontMaxGrid) { fused_ok = false; num_tail_tiles = 0; }
    }
    rows_ok = rows_ok && fused_ok;  // (if not: the clusters built for k_schur_rows are valid - only shorter - clusters of k_schur_clusters)
  }
  if (!setup_batched) sync();  // (a batched set-up has nothing in flight yet: build() flushes and synchronises once)
  lap("upload terms");
}

// Tiles of the J-free front end: consecutive points, at most kFrontObs observations / kFrontPts points / kFrontQ intrinsics
// entries each; a point with more observations than that is a tile of its own (walked in windows by the kernel).
void mavba_session::build_front_tiles(const std::vector<int>& q_start) {
  const bool planes_only = std::getenv("MAVBA_FRONT_PLANES") != nullptr;
  front_ok = !planes_only;
  for (int p = 0; p < NP && front_ok; ++p)
    if (q_start[p + 1] - q_start[p] > kFrontQ) front_ok = false;  // a point seen by that many refined cameras: plane kernels
  num_front_tiles = 0;
  if (front_ok) build_tiles(q_start, 0, d_front_tiles, num_front_tiles);
  else d_front_tiles.upload(std::vector<FrontTile>(), st);
  front_valid = false;
  if (!front_ok) ensure_planes();
}
// tiles over the points [first, NP)
void mavba_session::build_tiles(const std::vector<int>& q_start, int first, DevBuf<FrontTile>& out, int& count) {
  std::vector<FrontTile> tiles;
  int p0 = first, obs = 0, qs = 0;
  for (int p = first; p < NP; ++p) {
    const int c = h_pt_start[p + 1] - h_pt_start[p], nq = q_start[p + 1] - q_start[p];
    if (p > p0 && (obs + c > kFrontObs || p - p0 >= kFrontPts || qs + nq > kFrontQ)) {
      tiles.push_back(FrontTile{p0, p, h_pt_start[p0], h_pt_start[p], q_start[p0], q_start[p]});
      p0 = p; obs = 0; qs = 0;
    }
    obs += c; qs += nq;
  }
  if (NP > p0) tiles.push_back(FrontTile{p0, NP, h_pt_start[p0], h_pt_start[NP], q_start[p0], q_start[NP]});
  count = (int)tiles.size();
  out.upload(tiles, st);
}

void mavba_session::reset_state() {
  front_valid = false;
  // back to the problem as built: points filtered out of the resident problem return with their initial coordinates
  if (!h_pt_removed.empty()) {
    h_pt_removed.clear();
    apply_filter_state();
    M_is_clean = false;
  }
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
