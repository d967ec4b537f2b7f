// device_setup.hip - the O(observations) part of a session's set-up on the device (the flatten /
// re-index work of _bundle_adjustment_extract_data / _fill_problem, reference src/base3d/bundle_adjustment.cc:228-387,
// stops being a host cost once the solve is fast).
//
// Input: the caller's flat problem as it is (observations in the reference's residual order). Output, all in HBM:
//   * the internal point order: points sorted by the 8 smallest images that see them (one 128-bit key, ties by the caller's
//     point index) - neighbours in the order see the same images, which the Schur clusters rely on;
//   * observations point-major in that order (uv, image, point, caller index), the points' first observations;
//   * the image-major view (uv, point) for the camera sweep and the images' first observations.
// The host then builds what depends on the block STRUCTURE (clusters, term lists, elimination order) from three small
// downloads: 4 B per point and per observation, nothing of the 48 B per observation that used to be shuffled on the host.
//
// Building block: a hand-written stable LSD radix sort of element indices by an indirect 32-bit key word, 8-bit digits
// (histogram per work-group -> exclusive scan -> ranked scatter; ranks from wave ballots, no atomics: the result does not
// depend on scheduling).
#include "session.h"

namespace mavba {
namespace {

constexpr int kRsThreads = 256, kRsItems = 8, kRsChunk = kRsThreads * kRsItems;  // elements per work-group

// hist[bin * nblocks + block] = number of the block's elements whose digit is `bin`
__global__ void __launch_bounds__(kRsThreads) k_radix_hist(const int* __restrict__ vals, int n, const unsigned* __restrict__ word,
                                                           int shift, unsigned* __restrict__ hist, int nblocks) {
  __shared__ unsigned s_h[256];
  s_h[threadIdx.x] = 0u;
  __syncthreads();
  const int base = blockIdx.x * kRsChunk;
#pragma unroll
  for (int j = 0; j < kRsItems; ++j) {
    const int i = base + j * kRsThreads + threadIdx.x;
    if (i < n) {
      const int v = vals ? vals[i] : i;
      atomicAdd(&s_h[(word[v] >> shift) & 255u], 1u);  // (integer counts: order-independent)
    }
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_h[threadIdx.x];
}

// exclusive prefix sum of `total` unsigned values in place; ONE work-group of 1024 threads (short arrays: per-point /
// per-image counts). out_total (may be null) receives the grand total.
__global__ void __launch_bounds__(1024) k_scan_exclusive(unsigned* __restrict__ data, long long total, unsigned* __restrict__ out_total) {
  __shared__ unsigned s_part[1024];
  const int t = threadIdx.x;
  const long long per = (total + 1023) / 1024;
  const long long b = (long long)t * per, e = b + per < total ? b + per : total;
  unsigned sum = 0u;
  for (long long i = b; i < e; ++i) sum += data[i];
  s_part[t] = sum;
  __syncthreads();
  // Hillis-Steele over the 1024 partials
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned add = t >= off ? s_part[t - off] : 0u;
    __syncthreads();
    s_part[t] += add;
    __syncthreads();
  }
  unsigned run = t > 0 ? s_part[t - 1] : 0u;
  for (long long i = b; i < e; ++i) { const unsigned x = data[i]; data[i] = run; run += x; }
  if (out_total && t == 1023) *out_total = s_part[1023];
}

// The radix histogram hist[bin][block] is scanned in two parallel steps instead of by one work-group: row totals (one
// work-group per bin, coalesced over the blocks), then every row scans itself starting from the total of the bins before it.
__device__ __forceinline__ unsigned block_sum_u256(unsigned v, unsigned* s4) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return s4[0] + s4[1] + s4[2] + s4[3];
}
__global__ void __launch_bounds__(256) k_hist_row_totals(const unsigned* __restrict__ hist, int nblocks, unsigned* __restrict__ totals) {
  __shared__ unsigned s4[4];
  const unsigned* row = hist + (size_t)blockIdx.x * nblocks;
  unsigned sum = 0u;
  for (int i = threadIdx.x; i < nblocks; i += 256) sum += row[i];
  const unsigned tot = block_sum_u256(sum, s4);
  if (threadIdx.x == 0) totals[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) k_hist_row_scan(unsigned* __restrict__ hist, int nblocks, const unsigned* __restrict__ totals) {
  __shared__ unsigned s4[4];
  __shared__ unsigned s_scan[256];
  unsigned* row = hist + (size_t)blockIdx.x * nblocks;
  // base = total of the bins before this one
  const unsigned before = (int)threadIdx.x < (int)blockIdx.x ? totals[threadIdx.x] : 0u;
  unsigned run = block_sum_u256(before, s4);
  for (int b0 = 0; b0 < nblocks; b0 += 256) {
    const int i = b0 + threadIdx.x;
    const unsigned x = i < nblocks ? row[i] : 0u;
    s_scan[threadIdx.x] = x;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const unsigned add = (int)threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0u;
      __syncthreads();
      s_scan[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nblocks) row[i] = run + s_scan[threadIdx.x] - x;  // exclusive
    run += s_scan[255];
    __syncthreads();
  }
}

// Stable scatter: element i of the input goes to  hist[digit][block] (scanned) + its rank among the block's earlier
// elements with the same digit. A wave owns 512 consecutive elements (8 rounds of 64): in a round the lanes with equal
// digits find each other with 8 ballots, the wave's running count per digit sits in LDS.
__global__ void __launch_bounds__(kRsThreads) k_radix_scatter(const int* __restrict__ vals_in, int* __restrict__ vals_out, int n,
                                                              const unsigned* __restrict__ word, int shift,
                                                              const unsigned* __restrict__ hist, int nblocks) {
  __shared__ unsigned s_cnt[4][256];
  __shared__ unsigned s_off[4][256];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 4 * 256; i += kRsThreads) (&s_cnt[0][0])[i] = 0u;
  __syncthreads();
  const int wbase = blockIdx.x * kRsChunk + wv * (64 * kRsItems);
  int v[kRsItems];
  unsigned dg[kRsItems], rk[kRsItems];
#pragma unroll
  for (int j = 0; j < kRsItems; ++j) {
    const int i = wbase + j * 64 + lane;
    const bool valid = i < n;
    v[j] = valid ? (vals_in ? vals_in[i] : i) : 0;
    dg[j] = valid ? ((word[v[j]] >> shift) & 255u) : 0u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (dg[j] >> bit) & 1u;
      const unsigned long long bb = __ballot(valid && one);
      peers &= one ? bb : ~bb;
    }
    const unsigned below = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
    const unsigned prev = s_cnt[wv][dg[j]];           // (every peer reads the same word before the leader updates it)
    rk[j] = prev + below;
    if (valid && below == 0u) s_cnt[wv][dg[j]] = prev + (unsigned)__popcll(peers);
    if (!valid) rk[j] = 0u;
  }
  __syncthreads();
  {
    unsigned run = hist[(size_t)tid * nblocks + blockIdx.x];  // one digit per thread
#pragma unroll
    for (int w = 0; w < 4; ++w) { s_off[w][tid] = run; run += s_cnt[w][tid]; }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kRsItems; ++j) {
    const int i = wbase + j * 64 + lane;
    if (i < n) vals_out[s_off[wv][dg[j]] + rk[j]] = v[j];
  }
}

// ---- exclusive scan of any length: work-group chunks of 2048 (local scan + chunk total), scan of the totals, add ----
__global__ void __launch_bounds__(256) k_scan_chunks(unsigned* __restrict__ data, long long n, unsigned* __restrict__ totals) {
  __shared__ unsigned s_scan[256];
  const long long base = (long long)blockIdx.x * 2048 + (long long)threadIdx.x * 8;
  unsigned x[8], sum = 0u;
#pragma unroll
  for (int k = 0; k < 8; ++k) { x[k] = base + k < n ? data[base + k] : 0u; sum += x[k]; }
  s_scan[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const unsigned add = (int)threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0u;
    __syncthreads();
    s_scan[threadIdx.x] += add;
    __syncthreads();
  }
  unsigned run = s_scan[threadIdx.x] - sum;
#pragma unroll
  for (int k = 0; k < 8; ++k) { if (base + k < n) data[base + k] = run; run += x[k]; }
  if (threadIdx.x == 255) totals[blockIdx.x] = s_scan[255];
}
__global__ void __launch_bounds__(256) k_scan_add(unsigned* __restrict__ data, long long n, const unsigned* __restrict__ offsets) {
  const unsigned off = offsets[blockIdx.x];
  const long long base = (long long)blockIdx.x * 2048 + (long long)threadIdx.x * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) if (base + k < n) data[base + k] += off;
}

struct RadixScratch {
  DevBuf<unsigned> hist, totals;
  DevBuf<int> tmp;
};

// Stable sort of the indices `idx` (n of them; null = 0..n-1) by key word[idx], least significant pass first:
// passes[k] = (word array, shift). The result is left in `out` (idx may alias neither out nor scratch.tmp).
void radix_sort_indices(hipStream_t st, int n, const int* idx, int* out, RadixScratch& S,
                        const std::vector<std::pair<const unsigned*, int>>& passes) {
  if (n <= 0) return;
  const int nblocks = (n + kRsChunk - 1) / kRsChunk;
  if (S.hist.n < (size_t)256 * nblocks) S.hist.alloc((size_t)256 * nblocks);
  if (S.tmp.n < (size_t)n) S.tmp.alloc((size_t)n);
  if (S.totals.n < 256) S.totals.alloc(256);
  // ping-pong so that the LAST pass writes `out`
  const int np = (int)passes.size();
  const int* src = idx;
  for (int k = 0; k < np; ++k) {
    int* dst = ((np - 1 - k) % 2 == 0) ? out : S.tmp.p;
    hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(kRsThreads), 0, st, src, n, passes[k].first, passes[k].second, S.hist.p, nblocks);
    hipLaunchKernelGGL(k_hist_row_totals, dim3(256), dim3(256), 0, st, S.hist.p, nblocks, S.totals.p);
    hipLaunchKernelGGL(k_hist_row_scan, dim3(256), dim3(256), 0, st, S.hist.p, nblocks, S.totals.p);
    hipLaunchKernelGGL(k_radix_scatter, dim3(nblocks), dim3(kRsThreads), 0, st, src, dst, n, passes[k].first, passes[k].second, S.hist.p, nblocks);
    src = dst;
  }
  if (np == 0) {
    if (idx) HIP_OK(hipMemcpyAsync(out, idx, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    else throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "radix sort without passes needs an index array");
  }
}

}  // namespace
// scratch: at least device_scan_scratch(n) unsigned values, alive until the stream has run the scan
long long device_scan_scratch(long long n) {
  const long long n1 = (n + 2047) / 2048;
  return n <= 4096 ? 1 : n1 + (n1 + 2047) / 2048 + 1;
}
void device_scan_exclusive(hipStream_t st, unsigned* data, long long n, unsigned* scratch) {
  if (n <= 0) return;
  if (n <= 4096) { hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, st, data, n, (unsigned*)nullptr); return; }
  const long long nchunks = (n + 2047) / 2048;
  hipLaunchKernelGGL(k_scan_chunks, dim3((unsigned)nchunks), dim3(256), 0, st, data, n, scratch);
  device_scan_exclusive(st, scratch, nchunks, scratch + nchunks);
  hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nchunks), dim3(256), 0, st, data, n, scratch);
}
namespace {
int bytes_for(long long max_value) {  // key bytes needed for values in [0, max_value]
  int b = 1;
  while (b < 4 && (max_value >> (8 * b)) != 0) ++b;
  return b;
}

// ---- set-up kernels ----------------------------------------------------------------------------------------------
// Observations per point (global atomics: ~10 increments per address) and per image (thousands per address: first an LDS
// histogram per work-group of 4096 observations, then one global increment per image the group saw); index check.
constexpr int kCountImgLds = 8192;
__global__ void __launch_bounds__(256) k_count_obs(int n, int NI, int NP, const int* __restrict__ oimg, const int* __restrict__ opt,
                                                   unsigned* __restrict__ cnt_pt, unsigned* __restrict__ cnt_img, int* __restrict__ bad) {
  extern __shared__ unsigned s_img[];  // [min(NI, kCountImgLds)] or nothing
  const bool lds = NI <= kCountImgLds;
  if (lds) {
    for (int i = threadIdx.x; i < NI; i += 256) s_img[i] = 0u;
    __syncthreads();
  }
  const int base = blockIdx.x * 4096;
  for (int k = 0; k < 16; ++k) {
    const int o = base + k * 256 + threadIdx.x;
    if (o >= n) break;
    const int i = oimg[o], p = opt[o];
    if (i < 0 || i >= NI || p < 0 || p >= NP) { *bad = 1; continue; }
    atomicAdd(&cnt_pt[p], 1u);
    if (lds) atomicAdd(&s_img[i], 1u); else atomicAdd(&cnt_img[i], 1u);
  }
  if (lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < NI; i += 256) { const unsigned c = s_img[i]; if (c) atomicAdd(&cnt_img[i], c); }
  }
}

// Per caller point: the 8 smallest images that see it (repeats count once), packed 16 bits each into four key words,
// 0xFFFF padded (a point nobody sees sorts last); and the most significant key, `tail`: 1 for a point that cannot sit in a
// cluster whatever its neighbours are (more than kTailObs observations, or constant) - those go behind all the others, so
// that the clustered part of the problem can take the fused kernel and only this tail the separate front end.
__global__ void k_point_keys(int NP, const unsigned* __restrict__ cstart, const int* __restrict__ byp, const int* __restrict__ oimg,
                             const unsigned char* __restrict__ pconst, unsigned* __restrict__ w0, unsigned* __restrict__ w1,
                             unsigned* __restrict__ w2, unsigned* __restrict__ w3, unsigned* __restrict__ w4, unsigned* __restrict__ tail) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= NP) return;
  tail[p] = (cstart[p + 1] - cstart[p] > (unsigned)kTailObs || (pconst && pconst[p])) ? 1u : 0u;
  unsigned k[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) k[t] = 0xFFFFu;
  unsigned hash = 0;  // of the whole image set (order_on_host: the same sum)
  for (unsigned a = cstart[p]; a < cstart[p + 1]; ++a) {
    unsigned x = (unsigned)oimg[byp[a]];
    hash += image_set_mix(x);
    bool dup = false;
#pragma unroll
    for (int t = 0; t < 8; ++t) dup = dup || k[t] == x;
    if (dup) continue;
    // insert x into the sorted 8 (the largest falls out)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (x < k[t]) { const unsigned y = k[t]; k[t] = x; x = y; }
    }
  }
  w0[p] = k[0] << 16 | k[1]; w1[p] = k[2] << 16 | k[3]; w2[p] = k[4] << 16 | k[5]; w3[p] = k[6] << 16 | k[7];
  w4[p] = hash;
}

__global__ void k_new_counts(int NP, const int* __restrict__ orig, const unsigned* __restrict__ cstart, unsigned* __restrict__ cnt_new) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= NP) return;
  const int p = orig[q];
  cnt_new[q] = cstart[p + 1] - cstart[p];
}

// Internal point q: its parameters and its observations (caller order inside the point) to their point-major places.
__global__ void k_gather_point_major(int NP, const int* __restrict__ orig, const unsigned* __restrict__ cstart,
                                     const unsigned* __restrict__ pstart, const int* __restrict__ byp,
                                     const double* __restrict__ raw_uv, const int* __restrict__ raw_img,
                                     const double* __restrict__ raw_pts, const unsigned char* __restrict__ raw_pconst,
                                     double2* __restrict__ uv, int* __restrict__ oimg, int* __restrict__ opt, int* __restrict__ perm,
                                     double* __restrict__ pts, unsigned char* __restrict__ pconst, int* __restrict__ pt_start_i) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > NP) return;
  if (q == NP) { pt_start_i[NP] = (int)pstart[NP]; return; }
  const int p = orig[q];
  const unsigned src0 = cstart[p], n = cstart[p + 1] - src0, dst0 = pstart[q];
  pt_start_i[q] = (int)dst0;
  pts[3 * (size_t)q] = raw_pts[3 * (size_t)p]; pts[3 * (size_t)q + 1] = raw_pts[3 * (size_t)p + 1]; pts[3 * (size_t)q + 2] = raw_pts[3 * (size_t)p + 2];
  pconst[q] = raw_pconst ? raw_pconst[p] : 0;
  for (unsigned j = 0; j < n; ++j) {
    const int o = byp[src0 + j];
    uv[dst0 + j] = make_double2(raw_uv[2 * (size_t)o], raw_uv[2 * (size_t)o + 1]);
    oimg[dst0 + j] = raw_img[o];
    opt[dst0 + j] = q;
    perm[dst0 + j] = o;
  }
}

__global__ void k_gather_image_major(int n, const int* __restrict__ order, const double2* __restrict__ uv, const int* __restrict__ opt,
                                     double2* __restrict__ im_uv, int* __restrict__ im_pt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int a = order[t];
  im_uv[t] = uv[a];
  im_pt[t] = opt[a];
}

}  // namespace
// ---- intrinsics entries: one per (free point, free camera that sees it), cameras ascending (finish_structure) ----
// Distinct cameras by repeated minimum: k rounds over the point's observations (k = its number of refined cameras, 1-3
// in practice), no scratch. COUNT: q_count[p]; else fills q_pt / q_cam at q_start[p].
template <bool COUNT>
__global__ void __launch_bounds__(256) k_intr_entries(int NP, const int* __restrict__ pt_start, const int* __restrict__ obs_img,
                                                      const int* __restrict__ img_cam, const unsigned char* __restrict__ cam_active,
                                                      const unsigned char* __restrict__ pt_free, unsigned* __restrict__ q_count,
                                                      const unsigned* __restrict__ q_start, int* __restrict__ q_pt, int* __restrict__ q_cam) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= NP) return;
  int n = 0;
  if (pt_free[p]) {
    const int a0 = pt_start[p], a1 = pt_start[p + 1];
    int last = -1;
    for (;;) {
      int best = 0x7fffffff;
      for (int a = a0; a < a1; ++a) {
        const int c = img_cam[obs_img[a]];
        if (c > last && c < best && cam_active[c]) best = c;
      }
      if (best == 0x7fffffff) break;
      if (!COUNT) { q_pt[q_start[p] + n] = p; q_cam[q_start[p] + n] = best; }
      last = best;
      ++n;
    }
  }
  if (COUNT) q_count[p] = (unsigned)n;
}

}  // namespace mavba

using namespace mavba;

// finish_structure's intrinsics entries on the device (large problems: the host version walks the observations twice on 16
// threads, 1.1 ms at C3). d_q_start / d_q_pt / d_q_cam stay where the kernels want them; the host gets q_start and q_cam for
// the cluster construction. Same definition as the host version: cameras ascending inside a point.
void mavba_session::intr_entries_on_device(const std::vector<unsigned char>& cam_active, std::vector<int>& q_start, std::vector<int>& q_cam) {
  DevBuf<unsigned char> d_active;
  d_active.upload(cam_active, st);
  DevBuf<unsigned> cnt, scratch;
  cnt.alloc((size_t)NP + 1);
  scratch.alloc((size_t)device_scan_scratch((long long)NP + 1) + 8);
  HIP_OK(hipMemsetAsync(cnt.p + NP, 0, 4, st));
  hipLaunchKernelGGL((k_intr_entries<true>), dim3((NP + 255) / 256), dim3(256), 0, st, NP, d_pt_start.p, d_obs_img.p, d_img_cam.p, d_active.p,
                     d_pt_free.p, cnt.p, (const unsigned*)nullptr, (int*)nullptr, (int*)nullptr);
  device_scan_exclusive(st, cnt.p, (long long)NP + 1, scratch.p);
  q_start.resize((size_t)NP + 1);
  download(q_start.data(), cnt.p, ((size_t)NP + 1) * 4);
  const int total = q_start[NP];
  d_q_start.alloc((size_t)NP + 1);
  HIP_OK(hipMemcpyAsync(d_q_start.p, cnt.p, ((size_t)NP + 1) * 4, hipMemcpyDeviceToDevice, st));
  d_q_pt.alloc((size_t)std::max(total, 1)); d_q_cam.alloc((size_t)std::max(total, 1));
  q_cam.resize((size_t)total);
  if (total > 0) {
    hipLaunchKernelGGL((k_intr_entries<false>), dim3((NP + 255) / 256), dim3(256), 0, st, NP, d_pt_start.p, d_obs_img.p, d_img_cam.p, d_active.p,
                       d_pt_free.p, (unsigned*)nullptr, cnt.p, d_q_pt.p, d_q_cam.p);
    download(q_cam.data(), d_q_cam.p, (size_t)total * 4);
  }
  sync();  // (cnt, scratch and d_active are freed behind this)
}

// The ordering block of build() on the device. Fills: d_uv, d_obs_img, d_obs_pt, d_pt_start, d_im_uv, d_im_pt, d_pt_orig,
// d_points0, d_perm32; host: h_pt_orig, h_pt_start, h_oimg, h_pt_const_in (internal order), h_pt_count_all, h_pt_used,
// img_start. Throws MAVBA_ERR_BAD_INDEX for an observation index out of range.
void mavba_session::order_on_device(const mavba_problem* P, std::vector<int>& img_start, const DeviceRaw* raw) {
  const int n = N;
  RadixScratch S;
  const bool tt = std::getenv("MAVBA_SETUP_TIMING") != nullptr;
  double tl = now_s();
  auto lap = [&](const char* what) {  // (timing mode synchronises the stream at every lap)
    if (!tt) return;
    (void)hipStreamSynchronize(st);
    const double t = now_s();
    std::fprintf(stderr, "[setup]   dev: %-22s %8.2f ms\n", what, 1e3 * (t - tl));
    tl = t;
  };
  // raw problem -> device (page-locked staging blocks filled by a few host threads: the copies are asynchronous) - unless
  // it is there already (device-resident scene)
  DevBuf<double> r_uv_own, r_pts_own;
  DevBuf<int> r_img_own, r_pt_own;
  DevBuf<unsigned char> r_pconst;
  std::unique_ptr<PinnedBuf<double>> s_uv, s_pts;
  std::unique_ptr<PinnedBuf<int>> s_img, s_pt;
  struct { const double* p; } r_uv, r_pts;
  struct { const int* p; } r_img, r_pt;
  if (raw) {
    r_uv.p = raw->uv; r_img.p = raw->img; r_pt.p = raw->pt; r_pts.p = raw->points;
  } else {
    s_uv.reset(new PinnedBuf<double>((size_t)2 * n)); s_img.reset(new PinnedBuf<int>(n)); s_pt.reset(new PinnedBuf<int>(n));
    s_pts.reset(new PinnedBuf<double>((size_t)3 * std::max(NP, 1)));
    // Staged and uploaded in slices (round 6): while slice k travels (page-locked source: the copy is asynchronous) the host threads
    // copy slice k + 1 into the staging blocks - at C3 the 0.7-0.9 ms of staging used to sit in FRONT of the 1.0 ms of transfer.
    r_uv_own.alloc((size_t)2 * std::max<long long>(n, 1)); r_img_own.alloc((size_t)std::max<long long>(n, 1)); r_pt_own.alloc((size_t)std::max<long long>(n, 1));
    r_pts_own.alloc((size_t)3 * std::max(NP, 1));
    const int slices = n >= (1 << 20) ? 6 : 1;
    for (int sl = 0; sl < slices; ++sl) {
      const long long a0 = n * sl / slices, a1 = n * (sl + 1) / slices;
      parallel_ranges(a1 - a0, [&](long long b0, long long b1) {
        b0 += a0; b1 += a0;
        std::memcpy(s_uv->data() + 2 * b0, P->obs_uv + 2 * b0, (size_t)(b1 - b0) * 16);
        std::memcpy(s_img->data() + b0, P->obs_image + b0, (size_t)(b1 - b0) * 4);
        std::memcpy(s_pt->data() + b0, P->obs_point + b0, (size_t)(b1 - b0) * 4);
      });
      if (a1 > a0) {
        HIP_OK(hipMemcpyAsync(r_uv_own.p + 2 * a0, s_uv->data() + 2 * a0, (size_t)(a1 - a0) * 16, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(r_img_own.p + a0, s_img->data() + a0, (size_t)(a1 - a0) * 4, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(r_pt_own.p + a0, s_pt->data() + a0, (size_t)(a1 - a0) * 4, hipMemcpyHostToDevice, st));
      }
    }
    std::memcpy(s_pts->data(), P->points, (size_t)NP * 24);
    lap("stage (host memcpy) + upload of the observations");
    if (NP > 0) HIP_OK(hipMemcpyAsync(r_pts_own.p, s_pts->data(), (size_t)NP * 24, hipMemcpyHostToDevice, st));
    r_uv.p = r_uv_own.p; r_img.p = r_img_own.p; r_pt.p = r_pt_own.p; r_pts.p = r_pts_own.p;
  }
  if (P->point_const) r_pconst.upload(h_pt_const_in, st);  // (still in the caller's order here)
  lap("upload");
  // counts per point / per image, index check
  DevBuf<unsigned> cstart, pstart, istart, w[5];
  DevBuf<int> bad;
  cstart.alloc((size_t)NP + 1); pstart.alloc((size_t)NP + 1); istart.alloc((size_t)NI + 1);
  bad.alloc(1);
  cstart.zero(st); istart.zero(st); bad.zero(st);
  hipLaunchKernelGGL(k_count_obs, dim3((n + 4095) / 4096), dim3(256), NI <= kCountImgLds ? (size_t)NI * 4 : 0, st, n, NI, NP, r_img.p, r_pt.p,
                     cstart.p, istart.p, bad.p);
  DevBuf<unsigned> scan_scratch;  // (alive until this function's final synchronisation)
  scan_scratch.alloc((size_t)device_scan_scratch((long long)std::max(NP, NI) + 1) + 8);
  device_scan_exclusive(st, cstart.p, (long long)NP + 1, scan_scratch.p);
  device_scan_exclusive(st, istart.p, (long long)NI + 1, scan_scratch.p);
  int h_bad = 0;
  HIP_OK(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st));
  sync();
  if (h_bad) throw Failure(MAVBA_ERR_BAD_INDEX, "observation index out of range");
  lap("counts");
  // observations grouped by (caller's) point, caller order inside a point
  DevBuf<int> byp;
  byp.alloc((size_t)n);
  {
    std::vector<std::pair<const unsigned*, int>> passes;
    for (int b = 0; b < bytes_for(std::max(NP - 1, 0)); ++b) passes.push_back({reinterpret_cast<const unsigned*>(r_pt.p), 8 * b});
    radix_sort_indices(st, n, nullptr, byp.p, S, passes);
  }
  lap("sort obs by point");
  // point order: 128-bit key of the 8 smallest images, then the 32-bit hash of the whole image set, ties by the caller's index
  for (int k = 0; k < 5; ++k) w[k].alloc((size_t)std::max(NP, 1));
  DevBuf<unsigned> wtail;
  wtail.alloc((size_t)std::max(NP, 1));
  hipLaunchKernelGGL(k_point_keys, dim3((NP + 255) / 256), dim3(256), 0, st, NP, cstart.p, byp.p, r_img.p,
                     P->point_const ? r_pconst.p : (const unsigned char*)nullptr, w[0].p, w[1].p, w[2].p, w[3].p, w[4].p, wtail.p);
  d_pt_orig.alloc((size_t)std::max(NP, 1));
  {
    std::vector<std::pair<const unsigned*, int>> passes;
    for (int k = 4; k >= 0; --k)  // (least significant first: the image-set hash, then the eight smallest images)
      for (int b = 0; b < 4; ++b) passes.push_back({w[k].p, 8 * b});
    passes.push_back({wtail.p, 0});  // (most significant: the tail flag)
    radix_sort_indices(st, NP, nullptr, d_pt_orig.p, S, passes);
  }
  lap("point keys + sort");
  // point-major arrays in the new order
  hipLaunchKernelGGL(k_new_counts, dim3((NP + 255) / 256), dim3(256), 0, st, NP, d_pt_orig.p, cstart.p, pstart.p);
  HIP_OK(hipMemsetAsync(pstart.p + NP, 0, 4, st));
  device_scan_exclusive(st, pstart.p, (long long)NP + 1, scan_scratch.p);
  d_uv.alloc((size_t)std::max(n, 1)); d_obs_img.alloc((size_t)std::max(n, 1)); d_obs_pt.alloc((size_t)std::max(n, 1));
  d_perm32.alloc((size_t)std::max(n, 1)); d_pt_start.alloc((size_t)NP + 1); d_points0.alloc((size_t)std::max(NP, 1) * 3);
  DevBuf<unsigned char> pconst_new;
  pconst_new.alloc((size_t)std::max(NP, 1));
  hipLaunchKernelGGL(k_gather_point_major, dim3((NP + 1 + 255) / 256), dim3(256), 0, st, NP, d_pt_orig.p, cstart.p, pstart.p, byp.p,
                     r_uv.p, r_img.p, r_pts.p, P->point_const ? r_pconst.p : nullptr, d_uv.p, d_obs_img.p, d_obs_pt.p, d_perm32.p,
                     d_points0.p, pconst_new.p, d_pt_start.p);
  lap("gather point-major");
  // image-major view: positions sorted by image, point-major order inside an image
  DevBuf<int> im_order;
  im_order.alloc((size_t)std::max(n, 1));
  {
    std::vector<std::pair<const unsigned*, int>> passes;
    for (int b = 0; b < bytes_for(std::max(NI - 1, 0)); ++b) passes.push_back({reinterpret_cast<const unsigned*>(d_obs_img.p), 8 * b});
    radix_sort_indices(st, n, nullptr, im_order.p, S, passes);
  }
  d_im_uv.alloc((size_t)std::max(n, 1)); d_im_pt.alloc((size_t)std::max(n, 1));
  hipLaunchKernelGGL(k_gather_image_major, dim3((n + 255) / 256), dim3(256), 0, st, n, im_order.p, d_uv.p, d_obs_pt.p, d_im_uv.p, d_im_pt.p);
  lap("image-major view");
  // what the host's structure pass needs: 4 B per point and per observation
  h_pt_orig.resize(NP); h_pt_start.resize((size_t)NP + 1); img_start.resize((size_t)NI + 1);
  HostSpare<int>::take(h_oimg, (size_t)n);
  h_oimg.resize(n);
  std::vector<unsigned char> pc((size_t)std::max(NP, 1));
  // (through page-locked staging blocks, never into the vectors directly: see copy_h2d_staged in host_util.hip)
  if (NP) download(h_pt_orig.data(), d_pt_orig.p, (size_t)NP * 4);
  download(h_pt_start.data(), d_pt_start.p, ((size_t)NP + 1) * 4);
  download(img_start.data(), istart.p, ((size_t)NI + 1) * 4);
  if (n) download(h_oimg.data(), d_obs_img.p, (size_t)n * 4);
  if (NP) download(pc.data(), pconst_new.p, (size_t)NP);
  sync();
  h_pt_const_in.assign(pc.begin(), pc.begin() + NP);
  h_pt_count_all.resize(NP); h_pt_used.resize(NP);
  for (int q = 0; q < NP; ++q) { h_pt_count_all[q] = h_pt_start[q + 1] - h_pt_start[q]; h_pt_used[q] = h_pt_count_all[q] > 0; }
  lap("downloads");
  h_points0.clear();  // (the device holds the initial points; nothing on the host reads them on this path)
  perm.clear();
}

// perm (point-major position -> caller's observation index) on the host, for the probe paths that need it
void mavba_session::ensure_perm_host() {
  if (!perm.empty() || N == 0 || !d_perm32.p) return;
  std::vector<int> p32((size_t)N);
  download(p32.data(), d_perm32.p, (size_t)N * 4);
  perm.assign(p32.begin(), p32.end());
}

// Test entry: stable sort of 0..n-1 by keys[i] (32-bit, `key_bytes` significant bytes) on the device.
extern "C" int mavba_debug_radix_sort(int32_t n, const uint32_t* keys, int32_t key_bytes, int32_t* order_out, int32_t device) {
  try {
    if (n < 0 || (n > 0 && (!keys || !order_out)) || key_bytes < 1 || key_bytes > 4) throw Failure(MAVBA_ERR_INVALID_ARGUMENT, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); throw Failure(MAVBA_ERR_NO_DEVICE, "no HIP device: the mavba backend has no CPU path"); }
    if (device >= 0) HIP_OK(hipSetDevice(device));
    if (n == 0) return MAVBA_OK;
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    int rc = MAVBA_OK;
    {
      DevBuf<unsigned> k;
      DevBuf<int> out;
      RadixScratch S;
      k.upload(keys, (size_t)n, st);
      out.alloc((size_t)n);
      std::vector<std::pair<const unsigned*, int>> passes;
      for (int b = 0; b < key_bytes; ++b) passes.push_back({k.p, 8 * b});
      radix_sort_indices(st, n, nullptr, out.p, S, passes);
      HIP_OK(copy_d2h_staged_sync(order_out, out.p, (size_t)n * 4, st));
      release_staged(st);
    }
    (void)hipStreamDestroy(st);
    return rc;
  } catch (const Failure& f) { g_last_error = f.what(); return f.code; }
  catch (const std::exception& e) { g_last_error = e.what(); return MAVBA_ERR_HIP; }
}
