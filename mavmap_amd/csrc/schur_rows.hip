// schur_rows.hip - k_schur_rows: Jacobians, per-point sums, 3x3 factors and the Schur complement of a point cluster in ONE
// kernel, laid out so that NOTHING between the Jacobian and the entry matrix goes through LDS (round 4).
//
// What it replaces (reference file:line, /root/reference): the evaluation of BACostFunction<Model> over all residual blocks
// (src/base3d/bundle_adjustment.h:131-159 through ceres' AutoDiff) and the point elimination of ceres' SPARSE_SCHUR
// (bundle_adjustment.cc:554-569) for the clustered points - the same job as k_schur_fused
// (kernels.hip), which it supersedes as the default.
//
// Why another kernel. scripts/_dbg/pipe_bench.hip settled the pipe question on the MI355X: FP64 matrix and FP64 vector
// instructions of different waves on one SIMD do NOT overlap (a vector-only wave next to a matrix-only wave makes no
// progress at all until the matrix wave is done), and one wave issues v_mfma_f64_16x16x4_f64 every 64.6 cycles whatever
// it does. So the only levers are (1) fewer FP64 instructions and (2) fewer cycles in which no FP64 instruction issues.
//   (1) k_schur_fused multiplied structural zeros: a cluster's local list has 16 image slots (128 rows, 36 tiles), a point
//       touches 10 of them. Here the ROW COUNT IS A PROPERTY OF THE CLUSTER: the set-up (finish_structure) closes a
//       cluster when its rows would pass 80 (= 10 images + 2 cameras + the h row: 15 tiles) once it is long enough, and
//       the point order keeps points with the same image set together - 82 % of C3's points sit in such clusters, the
//       matrix-instruction count halves. Clusters that need more rows take the 128-row instantiation.
//   (2) Half of k_schur_fused's time went to phases that park products in LDS, barrier, re-read them with one lane per
//       sum, barrier, factorise on 32 owner lanes, barrier ... with one 512-lane work-group per CU. Here a point owns a
//       16-lane DPP row (lane 16 r + i = observation i of the batch's point r; a clusterable point has at most 16):
//       the 9 sums of the point block are an all-reduce over the row (mirror butterflies: both partners add the same two
//       numbers, so all 16 lanes end with bit-identical totals and EVERY lane factorises the 3x3 block itself - no owner
//       phase, no broadcast), the 3K sums per camera of the intrinsics entries are a reduce-scatter over the row (48
//       values halve four times; lane j ends with the three sums of entry row j). 256-lane work-groups, 37 KB of LDS
//       (80 rows) or 56 KB (128 rows): several work-groups share a CU and fill each other's latencies.
// Two barriers per 16-point batch (entry matrix free / entry matrix written); a row group clears and writes only ITS
// three columns of the entry matrix, in program order of one wave.
#include "internal.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "ba_math.h"
#include "dev_reduce.h"
#include "lm_decide.h"
#include "sweep_body.h"

// MAVBA_ROWS_SKIP (debug builds only, scripts/_dbg/rows_variants.sh): bit mask of parts left out to time the rest - wrong results.
#ifndef MAVBA_ROWS_SKIP
#define MAVBA_ROWS_SKIP 0
#endif

namespace mavba {
namespace {
typedef double f2_d4 __attribute__((ext_vector_type(4)));
constexpr int kF2Threads = 256, kF2Waves = 4;
constexpr int kF2K = 3 * kRowsBatch, kF2Pitch = kF2K + 2;  // (2 * pitch) mod 64 dwords = 36: the 32 lanes of a half-wave's operand read hit 64 different banks
static_assert(kRowsBatch == 16, "one point per 16-lane DPP row");
static_assert((2 * kF2Pitch) % 8 == 4, "pitch must spread the 16 operand rows over all banks");
// slot tables of a cluster: the 16 x 3 layout of ClusterShape{16, 3} for every row count
constexpr int kF2TabIP = 136, kF2TabII = 184, kF2Tab = 190;

// ---- DPP within a 16-lane row ----
// (bound_ctrl: every control used here reads a lane of the row, so the "old" value is never kept - with bound_ctrl = 0 the
// compiler materialised it all the same: one v_mov_b32 0 in front of every DPP move, 162 of the 1 400 vector instructions
// of a batch, profiles/r06_isa_by_line_k_schur_rows.txt)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int kDppMirror = 0x140;      // lane i <-> 15 - i
constexpr int kDppHalfMirror = 0x141;  // i <-> 7 - i inside each half row
constexpr int kDppQuadRev = 0x1B;      // quad_perm [3,2,1,0]: i <-> 3 - i inside each quad
constexpr int kDppQuadSwap = 0xB1;     // quad_perm [1,0,3,2]: i <-> i ^ 1
// (Round 6, measured and dropped - scripts/_dbg/issue_bench.hip, profiles/r06_issue_bench.txt, profiles/r06_ab_swizzle.txt: the
// same exchanges as ds_swizzle_b32 are LDS-pipe instructions that issue next to the other wave's vector or matrix
// instructions, and with both halves fetched the reduce-scatter's selects become adds under the execution mask - 17 % fewer
// vector instructions per batch, and the kernel was 4 % SLOWER (C3 0.2731 against 0.2625 ms, A/B/A/B in one visit): seven
// dependent LDS round trips per batch that the second wave of the SIMD does not cover.)
template <int CTRL>
__device__ __forceinline__ double xlane_f64(double v) { return dpp_f64<CTRL>(v); }
// Sum over the row, the same bits in all 16 lanes: at every step both partners add the same two numbers.
template <int NV>
__device__ __forceinline__ void row16_allsum(double (&v)[NV]) {
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e] += dpp_f64<kDppMirror>(v[e]);
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e] += dpp_f64<kDppHalfMirror>(v[e]);
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e] += dpp_f64<kDppQuadRev>(v[e]);
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e] += dpp_f64<kDppQuadSwap>(v[e]);
}
// One halving step of the reduce-scatter: NU values in v[0, NU) -> the pair's sums of the half this lane keeps in v[0, NU/2).
template <int CTRL, int NU, int N>
__device__ __forceinline__ void rs_step(double (&v)[N], bool upper) {
#pragma unroll
  for (int u = 0; u < NU / 2; ++u) {
    const double lo = v[u], hi = v[NU / 2 + u];
    const double keep = upper ? hi : lo, send = upper ? lo : hi;
    v[u] = keep + dpp_f64<CTRL>(send);
  }
}
// Reduce-scatter over the row of the 48 values val(u), u = 3 * entry row + column: lane i of the row ends with the row's
// sums of values 3 i .. 3 i + 2 in out[0..2]. The first halving (lane i <-> 15 - i: lanes 0-7 keep values 0-23, lanes 8-15
// values 24-47) evaluates the values on the fly - the 48 never exist at the same time (96 registers).
template <class F>
__device__ __forceinline__ void row16_reduce_scatter48(F&& val, int i, double (&out)[3]) {
  double v[24];
  const bool up1 = i >= 8;
#pragma unroll
  for (int u = 0; u < 24; ++u) {
    const double lo = val(u), hi = val(u + 24);
    const double keep = up1 ? hi : lo, send = up1 ? lo : hi;
    v[u] = keep + dpp_f64<kDppMirror>(send);
    if (u % 3 == 2) __builtin_amdgcn_sched_barrier(0);  // (one entry row at a time: without it the scheduler forms all 48 products first and spills)
  }
  rs_step<kDppHalfMirror, 24>(v, (i & 7) >= 4);
  rs_step<kDppQuadRev, 12>(v, (i & 3) >= 2);
  rs_step<kDppQuadSwap, 6>(v, (i & 1) != 0);
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
}

// ---- E E^T on the matrix cores: lower 16x16 tiles dealt round-robin to the 4 waves ----
template <int NT>
constexpr bool f2_row_used(int W, int row) {
  int t = 0;
  for (int i = 0; i < NT; ++i)
    for (int j = 0; j <= i; ++j) {
      if (t % kF2Waves == W && (i == row || j == row)) return true;
      ++t;
    }
  return false;
}
template <int NT>
struct F2Shape { static constexpr int rows = 16 * NT, tiles = NT * (NT + 1) / 2, acc = (tiles + kF2Waves - 1) / kF2Waves; };
template <int NT, int W>
__device__ __forceinline__ void f2_mfma(const double* __restrict__ E, int lane, f2_d4 (&acc)[F2Shape<NT>::acc]) {
  const int li = lane & 15, lk = lane >> 4;
  const double* base = E + li * kF2Pitch + lk;
  auto load = [&](double (&x)[NT], int kk) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
      if (f2_row_used<NT>(W, i)) x[i] = base[16 * i * kF2Pitch + kk];
  };
  auto mma = [&](const double (&x)[NT]) {
    int t = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        if (t % kF2Waves == W) acc[t / kF2Waves] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[i], x[j], acc[t / kF2Waves], 0, 0, 0);
        ++t;
      }
  };
  static_assert(kF2K % 8 == 0, "two k-steps per trip");
  double a[NT], b[NT];
  load(a, 0);
#pragma unroll 1
  for (int kk = 0; kk < kF2K; kk += 8) {
    load(b, kk + 4);
    __builtin_amdgcn_sched_barrier(0);  // reads first, then the matrix instructions they hide behind
    mma(a);
    __builtin_amdgcn_sched_barrier(0);
    if (kk + 8 < kF2K) load(a, kk + 8);
    __builtin_amdgcn_sched_barrier(0);
    mma(b);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// ---- the cluster's product leaves as block partials: staged through LDS, written block by block ----
// The accumulators sit in the matrix instruction's D layout (a lane holds 4 elements of a 16x16 tile); a block partial is
// 42 / 54 / 90 consecutive doubles somewhere else. Written straight from the registers every store instruction touched
// up to 64 different cache lines (8 useful bytes in each) - 0.13 of 0.31 ms at C3, half the kernel at C2 - and the index
// arithmetic per element was unrolled 16-36 times per lane. Now: the lower triangle goes to LDS packed by rows (the entry
// matrix is free after the last batch), then ONE lane per output element walks the blocks in order - consecutive lanes
// write consecutive doubles of a partial.
__host__ __device__ constexpr int f2_tri(int R) { return (R * (R + 1)) >> 1; }
// rows [RLO, RHI) of the lower triangle, packed: element (R, C) at tri(R) - tri(RLO) + C.
// Round 6: R = 16 i + 4 r + lk, so tri(R) = tri(16 i + 4 r) + (16 i + 4 r) lk + tri(lk): per lane ONE base (tri(lk) + li) and
// ONE multiple of lk per accumulator row; everything else is an immediate offset of the store. (Round 5 evaluated
// tri(R) - tri(RLO) + C per element: 1 400 integer instructions in the three instantiations.)
template <int NT, int W, int RLO, int RHI>
__device__ __forceinline__ void f2_stage(double* __restrict__ T, int lane, const f2_d4 (&acc)[F2Shape<NT>::acc]) {
  const int li = lane & 15, lk = lane >> 4;
  const double* Tl = T + (f2_tri(lk) + li);
  int t = 0;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      if (t % kF2Waves == W && 16 * i >= RLO && 16 * i < RHI) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int R0 = 16 * i + 4 * r;  // the row without the lane's part: R = R0 + lk, C = 16 j + li (D layout of the matrix instruction)
          double* dst = const_cast<double*>(Tl) + R0 * lk + (f2_tri(R0) - f2_tri(RLO) + 16 * j);
          if (i > j || li <= lk + 4 * r) *dst = acc[t / kF2Waves][r];
        }
      }
      ++t;
    }
}
// One lane per output element: the cluster's emit map (built per (ni, nc) at set-up, rows_emit_map) says where the element
// sits in the staged triangle and where it goes - slot-table index and offset inside the partial; s_dst holds the partials'
// addresses (null: the cluster does not touch that block). Consecutive lanes write consecutive doubles of a partial.
// (Round 5 derived all of that per element: two divisions, two table look-ups and the triangle index in a dependent chain,
// 600 ticks per trip of the loop.)
__device__ __forceinline__ void f2_write_blocks(const double* __restrict__ T, int tid, const unsigned* __restrict__ map, int n,
                                                double* const* __restrict__ s_dst) {
  // eight elements per lane and trip: the map entries are requested together, then the table / triangle reads, then the stores
  // (one element per trip was a chain of a global load, two LDS reads and a store - ~800 ticks each, 19 trips per cluster)
  constexpr int U = 8;
  for (int e0 = tid; e0 < n; e0 += U * kF2Threads) {
    unsigned m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) m[u] = e0 + u * kF2Threads < n ? map[e0 + u * kF2Threads] : 0u;
    double* d[U];
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      d[u] = e0 + u * kF2Threads < n ? s_dst[m[u] >> 20] : nullptr;
      const unsigned src = m[u] & 8191u;
      v[u] = src == kRowsEmitZero ? 0.0 : T[src];  // (kRowsEmitZero: a parameter the camera slot's model does not have)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) if (d[u]) d[u][(m[u] >> 13) & 127u] = v[u];
  }
}
}  // namespace

// One work-group per cluster. a.sw.cost_partial[cluster index] gets the cluster's cost; Cu, gu, Gi, h per point go out for
// the back-substitution and the gradient norm; the cluster's block partials through its slot table.
// GENERIC: the intrinsics entries by the general reduce-scatter (three camera slots in a cluster, or the 9-parameter model);
// otherwise every cluster of the launch has at most two camera slots and KMAX is 4 or 8 (the in-place form: 48 registers
// fewer - with both forms in one kernel the common one spilt ~70 registers per lane).
// One launch for all row classes: the cluster's row count selects the instantiation of the batch loop (the registers and
// the LDS of the kernel are those of the largest class either way: two work-groups per CU).
// SWEEP (round 6, local windows with constant intrinsics): work-groups behind the clusters' run the camera sweep's chunks
// (camera_sweep_body<0>, sweep_body.h) - the evaluation's other pass over the observations, independent of this one - so that
// a 65 us iteration has one launch less; the sweep's LDS scratch is this kernel's entry matrix.
struct RowsSweep { CamSweepArgs cs; int first; };  // first: number of cluster work-groups in front of the sweep's
template <int KMAX, bool GENERIC, bool TRACE = false, bool SWEEP = false>
__global__ void __launch_bounds__(kF2Threads, 2) k_schur_rows(
    FrontArgs a, const SchurRowsCluster* __restrict__ clusters, const int* __restrict__ tabs, const int* __restrict__ cl_lists,
    const unsigned short* __restrict__ obs_meta, const unsigned long long* __restrict__ lanemap,
    const unsigned* __restrict__ emit_map, double* __restrict__ part_pp, double* __restrict__ part_ip, double* __restrict__ part_ii,
    RowsSweep sweep) {
  using SHMAX = F2Shape<kRowsClassNT[kRowsClasses - 1]>;
  if (!lm_spec_go(a.spec, &a.radius)) return;  // (speculative evaluation: only behind an accepted step, with the radius it leaves)
  long long t_entry = 0, t_loop0 = 0, t_loop1 = 0;  // (TRACE: the cluster's time line - entry, tables ready, batches done, end)
  if constexpr (TRACE) t_entry = (long long)__builtin_amdgcn_s_memtime();
  // per-cluster tables (cl_lists, kRowsLists ints per cluster: its image slots, the camera of every image slot, its camera
  // slots; -1 padded): camera records, intrinsics and column scales are read from memory once per cluster
  __shared__ double s_rec[kClImagesMax][9], s_kin[kClImagesMax][9], s_sc[kClImagesMax][6], s_ksc[kClCamsMax][9];
  __shared__ int s_icam[kClImagesMax], s_model[kClImagesMax], s_lc[kClImagesMax], s_clcam[4];
  // (SWEEP with free intrinsics: the Gram form's rows need 70 KB - still two work-groups per CU)
  constexpr int kELen = (SWEEP && KMAX > 0 && kCsLdsDoubles > SHMAX::rows * kF2Pitch) ? kCsLdsDoubles : SHMAX::rows * kF2Pitch;
  __shared__ __attribute__((aligned(16))) double E[kELen];
  if constexpr (SWEEP) {
    static_assert(kF2Threads == 256 && SHMAX::rows * kF2Pitch >= 4 * kSweepAcc, "the sweep's chunk layout and scratch");
    if ((int)blockIdx.x >= sweep.first) {
      if constexpr (KMAX == 0) camera_sweep_body<0>(sweep.cs, (int)blockIdx.x - sweep.first, E);
      else camera_sweep_gram_body<KMAX>(sweep.cs, (int)blockIdx.x - sweep.first, E);
      return;
    }
  }
  __shared__ double s_red[kF2Waves];
  __shared__ double* s_dst[kF2Tab];  // the cluster's block partials by slot-table index (null: block not touched)
  __shared__ int s_pstart[kRowsMaxPoints + 1];
  __shared__ unsigned long long s_lanes[kRowsMaxPoints];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, r = tid >> 4, i = tid & 15;
  const SweepArgs& w = a.sw;
  const int NPs = a.NPs;
  const int cidx = blockIdx.x;
  const SchurRowsCluster cl = clusters[cidx];
  const int npts = cl.p1 - cl.p0, nbatch = (npts + kRowsBatch - 1) / kRowsBatch;
  const int P0 = 6 * cl.ni, H = P0 + rows_cam_rows(cl.flags);  // first camera row, the h row
  for (int j = tid; j <= npts; j += kF2Threads) s_pstart[j] = a.pt_start[cl.p0 + j];
  for (int j = tid; j < npts; j += kF2Threads) s_lanes[j] = lanemap[cl.p0 + j];
  if (tid < kF2Tab) {
    const int slot = tabs[(size_t)cidx * kF2Tab + tid];
    s_dst[tid] = slot < 0 ? nullptr : tid < kF2TabIP ? part_pp + (size_t)slot * 42 : tid < kF2TabII ? part_ip + (size_t)slot * 54 : part_ii + (size_t)slot * 90;
  }
  {
    const int* lists = cl_lists + (size_t)cidx * kRowsLists;
    if (tid < kClImagesMax * 9) {
      const int sl = tid / 9, e = tid - 9 * sl, img = lists[sl];
      if (img >= 0) {
        const int cam = lists[kRowsListsCam + sl];  // (= img_cam[img], noted at set-up: one dependent load less per cluster)
        s_rec[sl][e] = w.camrec[9 * img + e];
        s_kin[sl][e] = w.intr[9 * cam + e];
        if (e < 6) s_sc[sl][e] = a.scale_cam[6 * img + e];
        if (e == 0) {
          s_icam[sl] = cam; s_model[sl] = w.cam_model[cam];
          s_lc[sl] = cam == lists[kRowsListsCams] ? 0 : cam == lists[kRowsListsCams + 1] ? 1 : cam == lists[kRowsListsCams + 2] ? 2 : -1;  // the camera's slot (-1: constant intrinsics)
        }
      }
    } else if (tid < kClImagesMax * 9 + kClCamsMax * 9) {
      const int t = tid - kClImagesMax * 9, c = t / 9, k = t - 9 * c, cam = lists[kRowsListsCams + c];
      if (cam >= 0) s_ksc[c][k] = a.scale_cam[6 * w.NI + 9 * cam + k];
      if (k == 0) s_clcam[c] = cam;
    }
  }
  double cost = 0.0;
  auto run = [&](auto nt_const) {
  constexpr int NT = decltype(nt_const)::value;
  using SH = F2Shape<NT>;
  f2_d4 acc[SH::acc];
#pragma unroll
  for (int t = 0; t < SH::acc; ++t) acc[t] = (f2_d4){0.0, 0.0, 0.0, 0.0};
  long long stamp[TRACE ? 8 : 1];
  int nstamp = 0;
  bool tracing = false;
  auto mark = [&]() { if constexpr (TRACE) { if (tracing && nstamp < 8) stamp[nstamp++] = (long long)__builtin_amdgcn_s_memtime(); } };
  // A batch's loads - every lane's observation, the row's point - are requested before the PREVIOUS batch's matrix
  // instructions start and travel under them; the first batch's are requested HERE, next to the tables' (its point starts and
  // lane maps straight from memory: behind the tables' barrier they were one more exposed round trip per cluster).
  bool act_n = false, free_n = false, seen_n = false;
  int im_n = 0;
  double2 m_n = make_double2(0.0, 0.0);
  unsigned meta_n = 0xFFFFu;
  double X_n[3] = {0.0, 0.0, 0.0}, sp_n[3] = {0.0, 0.0, 0.0};
  auto request_point = [&](int pj, int ob, int cnt, unsigned long long lm) {
    const int p = cl.p0 + pj;
    const bool on = w.pt_active == nullptr || w.pt_active[p] != 0;  // (a point filtered out of the resident problem has no residual blocks)
    // the observation this lane takes: nibble i of the point's lane map (15 in an empty lane of a point with fewer than 16)
    const int oi = (int)(((i & 8) ? (unsigned)(lm >> 32) : (unsigned)lm) >> (4 * (i & 7))) & 15;
    seen_n = cnt > 0;
    act_n = on && oi < cnt;
    if (act_n) { im_n = w.obs_img[ob + oi]; m_n = w.uv[ob + oi]; meta_n = obs_meta[ob + oi]; }
    X_n[0] = w.points[3 * (size_t)p]; X_n[1] = w.points[3 * (size_t)p + 1]; X_n[2] = w.points[3 * (size_t)p + 2];
    free_n = a.pt_free[p] != 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) sp_n[k] = a.scale_pt[(size_t)k * NPs + p];
  };
  auto request_batch = [&](int bj) {
    const int pj = bj * kRowsBatch + r;
    act_n = free_n = seen_n = false;
    if (pj < npts) { const int ob = s_pstart[pj]; request_point(pj, ob, s_pstart[pj + 1] - ob, s_lanes[pj]); }
  };
  if (r < npts) { const int ob = a.pt_start[cl.p0 + r]; request_point(r, ob, a.pt_start[cl.p0 + r + 1] - ob, lanemap[cl.p0 + r]); }
  __syncthreads();
  if constexpr (TRACE) t_loop0 = (long long)__builtin_amdgcn_s_memtime();
  for (int bi = 0; bi < nbatch; ++bi) {
    const int pj = bi * kRowsBatch + r;
    const bool valid = pj < npts, act = act_n, own_free = free_n, seen = seen_n;
    const int im = im_n;
    const double2 m = m_n;
    const unsigned meta = act ? meta_n : 0xFFFFu;
    const double X[3] = {X_n[0], X_n[1], X_n[2]};
    const double own_sp[3] = {sp_n[0], sp_n[1], sp_n[2]};
    if constexpr (TRACE) tracing = bi == 1;
    lds_barrier();  // every wave's matrix instructions of the previous batch have read E: it is free from here on
    mark();  // 0: top of the batch
    // sp: the point's column scales (zero for a point that is not free: its entries vanish); jps = Jp' sp; P = Jk'^T jps, the
    // observation's contribution to the intrinsics entries of ITS camera
    constexpr int K3 = KMAX > 0 ? 3 * KMAX : 1;
    const double sp[3] = {own_free ? own_sp[0] : 0.0, own_free ? own_sp[1] : 0.0, own_free ? own_sp[2] : 0.0};
    double jc[12], jps[6], s9[9], P[K3];  // (jc, jps: written and read by the lanes with an observation only - no zeros for the others)
#pragma unroll
    for (int e = 0; e < 9; ++e) s9[e] = 0.0;
#pragma unroll
    for (int u = 0; u < K3; ++u) P[u] = 0.0;
    int mylc = -1;
    if (act) {
      int model, cam;
      double rec[9], kin[9];
      if (meta != 0xFFFFu) {  // an image of the cluster's list: everything but the point comes from LDS
        const int sl = (int)(meta >> 8);
        cam = s_icam[sl];
        model = s_model[sl];
        mylc = s_lc[sl];
#pragma unroll
        for (int k = 0; k < 9; ++k) { rec[k] = s_rec[sl][k]; kin[k] = s_kin[sl][k]; }
      } else {  // (constant pose: not in the list)
        cam = w.img_cam[im];
        model = w.cam_model[cam];
#pragma unroll
        for (int k = 0; k < 9; ++k) rec[k] = w.camrec[9 * im + k];
#pragma unroll
        for (int k = 0; k < 9; ++k) kin[k] = w.intr[9 * cam + k];
        if constexpr (KMAX > 0) mylc = cam == s_clcam[0] ? 0 : cam == s_clcam[1] ? 1 : cam == s_clcam[2] ? 2 : -1;
      }
      double res[2], Jc[12], Jp[6], Jk[18];
      double wgt, half_rho;
#if MAVBA_ROWS_SKIP & 4
      res[0] = m.x * rec[0] + kin[0]; res[1] = m.y * rec[1] + X[0];
      for (int e = 0; e < 12; ++e) Jc[e] = res[0] + e;
      for (int e = 0; e < 6; ++e) Jp[e] = res[1] + e;
      for (int e = 0; e < 18; ++e) Jk[e] = res[1] - e + model;
      wgt = 0.5; half_rho = res[0];
#else
      obs_jacobian(model, rec, kin, X, m.x, m.y, res, Jc, Jp, Jk);
      cauchy_weight(res[0] * res[0] + res[1] * res[1], w.loss_b, w.loss_inv_b, wgt, half_rho);
#endif
      cost += half_rho;
      const double rr0 = wgt * res[0], rr1 = wgt * res[1];
#pragma unroll
      for (int e = 0; e < 12; ++e) jc[e] = wgt * Jc[e];
      double jp[6];
#pragma unroll
      for (int e = 0; e < 6; ++e) jp[e] = wgt * Jp[e];
      s9[0] = jp[0] * jp[0] + jp[3] * jp[3]; s9[1] = jp[0] * jp[1] + jp[3] * jp[4]; s9[2] = jp[0] * jp[2] + jp[3] * jp[5];
      s9[3] = jp[1] * jp[1] + jp[4] * jp[4]; s9[4] = jp[1] * jp[2] + jp[4] * jp[5]; s9[5] = jp[2] * jp[2] + jp[5] * jp[5];
      s9[6] = jp[0] * rr0 + jp[3] * rr1; s9[7] = jp[1] * rr0 + jp[4] * rr1; s9[8] = jp[2] * rr0 + jp[5] * rr1;
#pragma unroll
      for (int e = 0; e < 6; ++e) jps[e] = jp[e] * sp[e % 3];
      // (an observation whose camera's intrinsics are constant contributes nothing to the intrinsics rows: its weight there is 0 -
      // the reduce-scatter below then needs no masks. GENERIC keeps the plain weight: it selects by camera slot.)
      const double wgt_k = (GENERIC || mylc >= 0) ? wgt : 0.0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const double k0 = wgt_k * Jk[k], k1 = wgt_k * Jk[9 + k];
#pragma unroll
        for (int t = 0; t < 3; ++t) P[3 * k + t] = k0 * jps[t] + k1 * jps[3 + t];
      }
    }
    mark();  // 1: Jacobian + products
    // ---- the point block's sums: all-reduce over the row, bit-identical in its 16 lanes ----
#if !(MAVBA_ROWS_SKIP & 16)
    row16_allsum(s9);
#endif
    // ---- every lane factorises its point's damped 3x3 block (the same arithmetic on the same bits in the 16 lanes) ----
    double G[6] = {0, 0, 0, 0, 0, 0}, hh[3] = {0, 0, 0};
    bool fin = true;
    if (own_free) {
      double C[6];
      C[0] = sp[0] * sp[0] * s9[0]; C[1] = sp[0] * sp[1] * s9[1]; C[2] = sp[0] * sp[2] * s9[2];
      C[3] = sp[1] * sp[1] * s9[3]; C[4] = sp[1] * sp[2] * s9[4]; C[5] = sp[2] * sp[2] * s9[5];
      const double inv_radius = 1.0 / a.radius;
      C[0] = __builtin_fma(clampd(C[0], a.dmin, a.dmax), inv_radius, C[0]);
      C[3] = __builtin_fma(clampd(C[3], a.dmin, a.dmax), inv_radius, C[3]);
      C[5] = __builtin_fma(clampd(C[5], a.dmin, a.dmax), inv_radius, C[5]);
#if MAVBA_ROWS_SKIP & 16
      for (int k = 0; k < 6; ++k) G[k] = C[k];
#else
      fin = chol3_inv_fast(C, G);
#endif
      const double gs[3] = {sp[0] * s9[6], sp[1] * s9[7], sp[2] * s9[8]};
      gi_mul(G, gs, hh);
#pragma unroll
      for (int k = 0; k < 6; ++k) fin = fin && isfinite(G[k]);
    }
    if (valid && i == 0) {  // the row's first lane writes the point's planes
      const int p = cl.p0 + pj;
#pragma unroll
      for (int k = 0; k < 6; ++k) a.Cu[(size_t)k * NPs + p] = s9[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) a.gu[(size_t)k * NPs + p] = s9[6 + k];
      if (seen) {
#pragma unroll
        for (int k = 0; k < 6; ++k) a.Gi[(size_t)k * NPs + p] = G[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) a.h[(size_t)k * NPs + p] = hh[k];
      }
      if (!fin) atomicAdd(a.fail, 1.0);
    }
    mark();  // 2: sums + factors
    // ---- the row group's three columns of the entry matrix: cleared, then written, by the lanes of ONE wave in program
    // order. The pose rows first: the observation's Jacobian is dead after them (register pressure). ----
    double* col = E + 3 * r;
#if !(MAVBA_ROWS_SKIP & 32)
#pragma unroll
    for (int rb = 0; rb < NT; ++rb) {
      double* e = col + (16 * rb + i) * kF2Pitch;
      e[0] = 0.0; e[1] = 0.0; e[2] = 0.0;
    }
#endif
    if (act && meta != 0xFFFFu && !(MAVBA_ROWS_SKIP & 8)) {  // (0xFFFF: the image's pose is constant - it has no rows; the sums above included it)
      double* Eo = col + 6 * (int)(meta >> 8) * kF2Pitch;
      const double* scl = s_sc[meta >> 8];
      // U = (Jc'^T Jp') Gi^T = Jc'^T (Jp' Gi^T): the 2 x 3 factor once (12 multiply-adds), then two per element - 60 FP64
      // instructions per observation instead of 84 for the 6 x 3 product followed by Gi^T
      const double m00 = jps[0] * G[0], m01 = jps[0] * G[1] + jps[1] * G[2], m02 = jps[0] * G[3] + jps[1] * G[4] + jps[2] * G[5];
      const double m10 = jps[3] * G[0], m11 = jps[3] * G[1] + jps[4] * G[2], m12 = jps[3] * G[3] + jps[4] * G[4] + jps[5] * G[5];
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const double sc = scl[e];
        const double j0 = jc[e] * sc, j1 = jc[6 + e] * sc;
        Eo[e * kF2Pitch] = j0 * m00 + j1 * m10;
        Eo[e * kF2Pitch + 1] = j0 * m01 + j1 * m11;
        Eo[e * kF2Pitch + 2] = j0 * m02 + j1 * m12;
      }
    }
    if (i < 3 && own_free && seen) col[H * kF2Pitch + i] = hh[i];
    // ---- intrinsics entries: Wk = sum over the camera's observations of P, reduce-scattered over the row; the lane that ends
    // with the three sums of an entry row (camera slot lc, parameter k) writes U = (s_k Wk) Gi^T ----
    if constexpr (KMAX > 0 && !(MAVBA_ROWS_SKIP & 2)) {
      // (slot lc's rows: one per parameter of ITS model - a lane whose parameter the model does not have writes nothing)
      auto write_row = [&](int lc, int k, const double* W) {
        const int r0 = rows_cam_off(cl.flags, lc);
        if (k >= rows_cam_off(cl.flags, lc + 1) - r0) return;
        const double sk = s_ksc[lc][k];
        const double w0 = W[0] * sk, w1 = W[1] * sk, w2 = W[2] * sk;
        double* Eo = col + (P0 + r0 + k) * kF2Pitch;
        Eo[0] = w0 * G[0];
        Eo[1] = w0 * G[1] + w1 * G[2];
        Eo[2] = w0 * G[3] + w1 * G[4] + w2 * G[5];
      };
      if constexpr (!GENERIC) {
        static_assert(KMAX == 8 || KMAX == 4, "the in-place form is written for 4 and 8 parameters");
        // IN PLACE: lanes 0-7 of the row end with camera slot 0's sums, lanes 8-15 with slot 1's, then the 3 K values halve
        // inside the 8 lanes. Round 6: with two camera slots the SET-UP has put every observation into a lane of its
        // camera's half (the point's lane map; a cluster where that is impossible - more than 8 observations of one camera
        // in a point - is flagged and keeps the selects), so the first halving - i <-> 15 - i with a keep and a send
        // select per value, 24 x 7 instructions - is gone; with one slot it is a plain sum of the mirrored lanes.
        if (cl.nc < 2) {
#pragma unroll
          for (int u = 0; u < K3; ++u) P[u] += xlane_f64<kDppMirror>(P[u]);
        } else if (cl.flags & kRowsUnplaced) {  // (a point of the cluster has more than 8 observations of one slot: keep / send by camera, as in round 4)
          const bool up1 = i >= 8;
          const bool keepm = mylc == (up1 ? 1 : 0), sendm = mylc == (up1 ? 0 : 1);
#pragma unroll
          for (int u = 0; u < K3; ++u) {
            const double keep = keepm ? P[u] : 0.0, send = sendm ? P[u] : 0.0;
            P[u] = keep + xlane_f64<kDppMirror>(send);
          }
        }
        rs_step<kDppHalfMirror, K3>(P, (i & 7) >= 4);
        rs_step<kDppQuadRev, K3 / 2>(P, (i & 3) >= 2);
        if constexpr (KMAX == 8) {
          rs_step<kDppQuadSwap, K3 / 4>(P, (i & 1) != 0);                 // lane i: parameter i & 7
          if ((i >> 3) < cl.nc) write_row(i >> 3, i & 7, P);
        } else {
#pragma unroll
          for (int t = 0; t < 3; ++t) P[t] += xlane_f64<kDppQuadSwap>(P[t]);  // 3 values left for 2 lanes: both add, the even one writes
          if ((i >> 3) < cl.nc && (i & 1) == 0) write_row(i >> 3, (i & 7) >> 1, P);
        }
      } else {
        constexpr int NROUND = (kClCamsMax * KMAX + 15) / 16;
#pragma unroll
        for (int rd = 0; rd < NROUND; ++rd) {
          if (16 * rd < cl.nc * KMAX) {  // (uniform over the work-group)
            auto val = [&](int u) {
              const int jj = 16 * rd + u / 3, t = u % 3, lc = jj / KMAX, k = jj % KMAX;
              return (lc < kClCamsMax && mylc == lc) ? P[3 * k + t] : 0.0;
            };
            double W3[3];
            row16_reduce_scatter48(val, i, W3);
            const int jj = 16 * rd + i, lc = jj / KMAX, k = jj - KMAX * lc;
            if (lc < cl.nc) write_row(lc, k, W3);
          }
        }
      }
    }
    lds_barrier();
    mark();  // 3: entry matrix written
    if (bi + 1 < nbatch) {
      request_batch(bi + 1);
      __builtin_amdgcn_sched_barrier(0);  // (keep the loads here: the scheduler would sink them to their use)
    }
    if (!(MAVBA_ROWS_SKIP & 1))
    switch (wv) {
      case 0: f2_mfma<NT, 0>(E, lane, acc); break;
      case 1: f2_mfma<NT, 1>(E, lane, acc); break;
      case 2: f2_mfma<NT, 2>(E, lane, acc); break;
      default: f2_mfma<NT, 3>(E, lane, acc); break;
    }
    mark();  // 4: matrix instructions issued
  }
  if constexpr (TRACE) {
    t_loop1 = (long long)__builtin_amdgcn_s_memtime();
    if (a.trace && lane == 0 && blockIdx.x < 4096) {
      long long* out = a.trace + ((size_t)blockIdx.x * kF2Waves + wv) * 16;
      out[0] = nstamp;
      for (int t = 0; t < nstamp; ++t) out[1 + t] = stamp[t];
    }
  }
  if constexpr (!(MAVBA_ROWS_SKIP & 64)) {
    // passes over row ranges of the packed lower triangle that fit the (free) entry matrix buffer
    auto pass = [&](auto lo_c, auto hi_c) {
      constexpr int RLO = decltype(lo_c)::value, RHI = decltype(hi_c)::value;
      static_assert((RHI * (RHI + 1) - RLO * (RLO + 1)) / 2 <= SHMAX::rows * kF2Pitch, "a pass must fit the entry matrix buffer");
      lds_barrier();  // every wave's last matrix instructions (or the previous pass's readers) are done with the buffer
      switch (wv) {
        case 0: f2_stage<NT, 0, RLO, RHI>(E, lane, acc); break;
        case 1: f2_stage<NT, 1, RLO, RHI>(E, lane, acc); break;
        case 2: f2_stage<NT, 2, RLO, RHI>(E, lane, acc); break;
        default: f2_stage<NT, 3, RLO, RHI>(E, lane, acc); break;
      }
      lds_barrier();
      constexpr int PASS = RLO == 0 ? 0 : 1;
      f2_write_blocks(E, tid, emit_map + cl.emit_off + (PASS ? cl.emit_n[0] : 0), cl.emit_n[PASS], s_dst);
    };
    if constexpr (NT <= 6) pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 16 * NT>{});
    else { pass(std::integral_constant<int, 0>{}, std::integral_constant<int, kRowsPassSplit>{}); pass(std::integral_constant<int, kRowsPassSplit>{}, std::integral_constant<int, 16 * NT>{}); }
  }
  };  // run
  static_assert(kRowsClasses == 3, "one instantiation of the batch loop per row class");
  switch (rows_class_of_rows(H + 1)) {  // (uniform over the work-group)
    case 0: run(std::integral_constant<int, kRowsClassNT[0]>{}); break;
    case 1: run(std::integral_constant<int, kRowsClassNT[1]>{}); break;
    default: run(std::integral_constant<int, kRowsClassNT[2]>{}); break;
  }
  // cost partial of the cluster (fixed tree: lanes -> waves -> work-group)
  const double wsum = wave_sum(cost);
  __syncthreads();
  if (lane == 0) s_red[wv] = wsum;
  __syncthreads();
  if (tid == 0) w.cost_partial[cidx] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  if constexpr (TRACE) {
    if (a.trace && lane == 0 && blockIdx.x < 4096) {
      long long* out = a.trace + ((size_t)blockIdx.x * kF2Waves + wv) * 16;
      out[9] = t_entry; out[10] = t_loop0; out[11] = t_loop1; out[12] = (long long)__builtin_amdgcn_s_memtime();
      out[13] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);  // HW_ID | XCC_ID << 32: where the wave ran
      out[14] = ((long long)nbatch << 8) | rows_class_of_rows(H + 1);
    }
  }
}

void rows_emit_map(int ni, int nc, int flags, std::vector<unsigned>& pass0, std::vector<unsigned>& pass1) {
  pass0.clear(); pass1.clear();
  const int P0 = 6 * ni, H = P0 + rows_cam_rows(flags);
  const bool two = kRowsClassNT[rows_class_of_rows(H + 1)] > 6;  // (the kernel stages a product of more than 96 rows in two passes)
  auto koff = [&](int lc) { return rows_cam_off(flags, lc); };
  auto kcnt = [&](int lc) { return koff(lc + 1) - koff(lc); };
  auto put = [&](int R, int C, int o, int tab_index) {
    const int pass = two && R >= kRowsPassSplit ? 1 : 0;
    (pass ? pass1 : pass0).push_back(rows_emit_entry(f2_tri(R) - (pass ? f2_tri(kRowsPassSplit) : 0) + C, o, tab_index));
  };
  // (an element without a source row: written as zero, in the pass that writes the slot's other elements of that row block)
  auto put_zero = [&](int R_near, int o, int tab_index) {
    const int pass = two && R_near >= kRowsPassSplit ? 1 : 0;
    (pass ? pass1 : pass0).push_back(rows_emit_entry((int)kRowsEmitZero, o, tab_index));
  };
  for (int la = 0; la < ni; ++la)  // pose x pose (+ the h row's part of the diagonal blocks)
    for (int lb = 0; lb <= la; ++lb)
      for (int o = 0; o < 42; ++o) {
        if (o < 36) { const int r = o / 6, c = o % 6; if (la == lb && r < c) continue; put(6 * la + r, 6 * lb + c, o, f2_tri(la) + lb); }
        else if (la == lb) put(H, 6 * la + (o - 36), o, f2_tri(la) + lb);
      }
  for (int lc = 0; lc < nc; ++lc)  // intrinsics x pose
    for (int la = 0; la < ni; ++la)
      for (int o = 0; o < 54; ++o) {
        const int r = o / 6;
        if (r < kcnt(lc)) put(P0 + koff(lc) + r, 6 * la + o % 6, o, kF2TabIP + lc * 16 + la);
        else put_zero(P0 + koff(lc), o, kF2TabIP + lc * 16 + la);
      }
  for (int lc = 0; lc < nc; ++lc)  // intrinsics x intrinsics (+ the h row's part)
    for (int lc2 = 0; lc2 <= lc; ++lc2)
      for (int o = 0; o < 90; ++o) {
        const int t = kF2TabII + f2_tri(lc) + lc2;
        if (o < 81) {
          const int r = o / 9, c = o % 9;
          if (lc == lc2 && r < c) continue;
          if (r < kcnt(lc) && c < kcnt(lc2)) put(P0 + koff(lc) + r, P0 + koff(lc2) + c, o, t);
          else put_zero(P0 + koff(lc), o, t);
        } else if (lc == lc2) {
          if (o - 81 < kcnt(lc)) put(H, P0 + koff(lc) + (o - 81), o, t);
          else put_zero(H, o, t);
        }
      }
}

// generic: some cluster has three camera slots (the 9-parameter model always takes the general form of the intrinsics entries).
void launch_schur_rows(hipStream_t st, const FrontArgs& a, int kmax_intr, bool generic, int num_clusters,
                       const SchurRowsCluster* clusters, const int* tab, const int* cl_lists, const unsigned short* obs_meta,
                       const unsigned long long* lanemap, const unsigned* emit_map, double* part_pp, double* part_ip, double* part_ii,
                       const CamSweepArgs* with_sweep) {
  if (num_clusters <= 0) return;
  RowsSweep sweep{};
  sweep.first = num_clusters;
  if (with_sweep) {
    // (the caller asks for it on local windows; the sweep's form follows the widest model like launch_camera_sweep's: the vector
    // kernel's body without free intrinsics, the Gram form's with)
    sweep.cs = *with_sweep;
#define MAVBA_ROWS_SWEEP(K, G) hipLaunchKernelGGL((k_schur_rows<K, G, false, true>), dim3(num_clusters + with_sweep->num_chunks), dim3(kF2Threads), 0, st, a, \
                                                 clusters, tab, cl_lists, obs_meta, lanemap, emit_map, part_pp, part_ip, part_ii, sweep)
    if (kmax_intr <= 0) MAVBA_ROWS_SWEEP(0, true);
    else if (kmax_intr <= 4) { if (generic) MAVBA_ROWS_SWEEP(4, true); else MAVBA_ROWS_SWEEP(4, false); }
    else if (kmax_intr <= 8) { if (generic) MAVBA_ROWS_SWEEP(8, true); else MAVBA_ROWS_SWEEP(8, false); }
    else MAVBA_ROWS_SWEEP(9, true);
#undef MAVBA_ROWS_SWEEP
    return;
  }
  // MAVBA_ROWS_TRACE=<file>: the 5th launch of the process (widest model <= 8) records s_memtime stamps per wave
  static const char* trace_file = std::getenv("MAVBA_ROWS_TRACE");
  static int trace_calls = 0;
  if (trace_file && !generic && kmax_intr > 4 && kmax_intr <= 8 && ++trace_calls == 5) {
    const size_t trace_n = (size_t)4096 * kF2Waves * 16;
    long long* tr = nullptr;
    (void)hipMalloc(reinterpret_cast<void**>(&tr), trace_n * 8);
    (void)hipMemsetAsync(tr, 0, trace_n * 8, st);
    FrontArgs b = a;
    b.trace = tr;
    hipLaunchKernelGGL((k_schur_rows<8, false, true>), dim3(num_clusters), dim3(kF2Threads), 0, st, b, clusters, tab, cl_lists, obs_meta, lanemap, emit_map, part_pp, part_ip, part_ii, sweep);
    std::vector<long long> hst(trace_n);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(hst.data(), tr, trace_n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(tr);
    if (FILE* fp = std::fopen(trace_file, "w")) {
      for (int g = 0; g < std::min(num_clusters, 4096); ++g)
        for (int wv = 0; wv < kF2Waves; ++wv) {
          const long long* rr = hst.data() + ((size_t)g * kF2Waves + wv) * 16;
          if (rr[12] <= 0) continue;
          // cluster, wave, number of phase stamps (second batch; 0 for a one-batch cluster), 8 phase stamps, then the cluster's
          // time line: entry, tables ready, batches done, end, HW_ID, batches << 8 | row class
          std::fprintf(fp, "%d %d", g, wv);
          for (int t = 0; t < 15; ++t) std::fprintf(fp, " %lld", rr[t]);
          std::fprintf(fp, "\n");
        }
      std::fclose(fp);
    }
    return;
  }
#define MAVBA_ROWS(K, G) hipLaunchKernelGGL((k_schur_rows<K, G>), dim3(num_clusters), dim3(kF2Threads), 0, st, a, clusters, tab, cl_lists, obs_meta, lanemap, emit_map, part_pp, part_ip, part_ii, sweep)
  if (kmax_intr <= 0) MAVBA_ROWS(0, true);
  else if (kmax_intr <= 4) { if (generic) MAVBA_ROWS(4, true); else MAVBA_ROWS(4, false); }
  else if (kmax_intr <= 8) { if (generic) MAVBA_ROWS(8, true); else MAVBA_ROWS(8, false); }
  else MAVBA_ROWS(9, true);
#undef MAVBA_ROWS
}

}  // namespace mavba
