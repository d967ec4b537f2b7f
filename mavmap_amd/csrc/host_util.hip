// host_util.hip - process-wide helpers of the host side: caching device allocator, stream cache, worker threads.
#include "session.h"
#include <atomic>
#include <cstdint>
#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace mavba {

// ---- RCCL, loaded on first use (a single-GPU process never maps the 570 MB library) --------------------------------
namespace {
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;
  std::string error;
};
RcclApi& rccl() {
  static RcclApi* api = [] {
    RcclApi* a = new RcclApi;
    // MAVBA_RCCL_LIB: another library with the same entry points (tests/stubs/mock_rccl.c: the in-process rank protocol on
    // a box without GPUs)
    const char* names[] = {std::getenv("MAVBA_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) if (n && *n && (a->lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!a->lib) { a->error = std::string("librccl.so not found: ") + dlerror(); return a; }
    a->GetUniqueId = reinterpret_cast<decltype(a->GetUniqueId)>(dlsym(a->lib, "ncclGetUniqueId"));
    a->CommInitRank = reinterpret_cast<decltype(a->CommInitRank)>(dlsym(a->lib, "ncclCommInitRank"));
    a->AllReduce = reinterpret_cast<decltype(a->AllReduce)>(dlsym(a->lib, "ncclAllReduce"));
    a->CommDestroy = reinterpret_cast<decltype(a->CommDestroy)>(dlsym(a->lib, "ncclCommDestroy"));
    a->CommAbort = reinterpret_cast<decltype(a->CommAbort)>(dlsym(a->lib, "ncclCommAbort"));
    a->GetErrorString = reinterpret_cast<decltype(a->GetErrorString)>(dlsym(a->lib, "ncclGetErrorString"));
    a->CommGetAsyncError = reinterpret_cast<decltype(a->CommGetAsyncError)>(dlsym(a->lib, "ncclCommGetAsyncError"));
    if (!a->GetUniqueId || !a->CommInitRank || !a->AllReduce || !a->CommDestroy) a->error = "librccl.so lacks an expected symbol";
    return a;
  }();
  return *api;
}
void rccl_ok(ncclResult_t r, const char* what) {
  if (r != ncclSuccess)
    throw Failure(MAVBA_ERR_HIP, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error"));
}
}  // namespace

void rccl_unique_id(void* out128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  if (!rccl().error.empty()) throw Failure(MAVBA_ERR_HIP, rccl().error);
  ncclUniqueId id;
  rccl_ok(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out128, &id, sizeof(id));
}
void* rccl_comm_create(const void* id128, int rank, int world) {
  if (!rccl().error.empty()) throw Failure(MAVBA_ERR_HIP, rccl().error);
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  rccl_ok(rccl().CommInitRank(&comm, world, id, rank), "ncclCommInitRank");
  return comm;
}
void rccl_comm_destroy(void* comm) { if (comm && rccl().CommDestroy) (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm)); }
// gives up whatever the communicator has in flight (a peer is gone) and frees it; plain destroy where the library has no abort
void rccl_comm_abort(void* comm) {
  if (!comm) return;
  if (rccl().CommAbort) (void)rccl().CommAbort(static_cast<ncclComm_t>(comm));
  else rccl_comm_destroy(comm);
}
int rccl_comm_async_error(void* comm) {
  if (!comm || !rccl().CommGetAsyncError) return 0;
  ncclResult_t e = ncclSuccess;
  if (rccl().CommGetAsyncError(static_cast<ncclComm_t>(comm), &e) != ncclSuccess) return 1;
  return e == ncclSuccess ? 0 : (int)e;
}
// in place on `stream`; op 0 = sum, 1 = max, 2 = sum over the first count - 1 doubles and max over the last
void rccl_allreduce(void* comm, double* p, long long count, int op, hipStream_t stream) {
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  if (op == 2) {
    if (count > 1) rccl_ok(rccl().AllReduce(p, p, (size_t)(count - 1), ncclDouble, ncclSum, c, stream), "ncclAllReduce");
    rccl_ok(rccl().AllReduce(p + count - 1, p + count - 1, 1, ncclDouble, ncclMax, c, stream), "ncclAllReduce");
  } else {
    rccl_ok(rccl().AllReduce(p, p, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, c, stream), "ncclAllReduce");
  }
}

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

// Pinned host staging for the read-backs (a pageable 4.8 MB device->host copy costs ~3 ms through the runtime's own
// staging; pinned it is ~0.2 ms + a memcpy): blocks are cached like the device blocks; at most kPinnedParkedBytes of
// page-locked memory stay parked (MAVBA_PINNED_POOL_MB, default 1024), what comes back beyond that is unlocked and freed.
namespace {
struct PinnedPool { std::mutex m; std::multimap<size_t, void*> free_blocks; std::unordered_map<void*, size_t> live; size_t parked = 0; };
PinnedPool& pinned_pool() { static PinnedPool* p = new PinnedPool; return *p; }
size_t pinned_cap() {
  static const size_t v = [] { const char* e = std::getenv("MAVBA_PINNED_POOL_MB"); return (size_t)(e ? std::atoll(e) : 1024) << 20; }();
  return v;
}
}  // namespace
hipError_t pinned_alloc(void** out, size_t bytes) {
  PinnedPool& P = pinned_pool();
  const size_t cls = DevicePool::size_class(bytes);
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.free_blocks.find(cls);
    if (it != P.free_blocks.end()) { *out = it->second; P.free_blocks.erase(it); P.parked -= cls; P.live[*out] = cls; return hipSuccess; }
  }
  const hipError_t e = hipHostMalloc(out, cls, hipHostMallocDefault);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> g(P.m);
  P.live[*out] = cls;
  return hipSuccess;
}
void pinned_free(void* p) {
  if (!p) return;
  PinnedPool& P = pinned_pool();
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.live.find(p);
    if (it == P.live.end()) return;
    const size_t cls = it->second;
    P.live.erase(it);
    if (P.parked + cls <= pinned_cap()) { P.free_blocks.insert({cls, p}); P.parked += cls; return; }
  }
  (void)hipHostFree(p);
}

// Host-mapped, COHERENT slots (256 B) for the LM scalars a session's k_lm_snapshot publishes and its host loop polls: one
// slab per process, slots handed out and returned (a local-BA mapper creates a session per call).
namespace {
struct PubPool { std::mutex m; char* slab = nullptr; std::vector<int> free_slots; bool tried = false; };
PubPool& pub_pool() { static PubPool* p = new PubPool; return *p; }
constexpr int kPubSlots = 256, kPubSlotBytes = 256;
}  // namespace
double* lm_pub_alloc() {
  PubPool& P = pub_pool();
  std::lock_guard<std::mutex> g(P.m);
  if (!P.tried) {
    P.tried = true;
    void* q = nullptr;
    if (hipHostMalloc(&q, (size_t)kPubSlots * kPubSlotBytes, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
      P.slab = static_cast<char*>(q);
      std::memset(P.slab, 0, (size_t)kPubSlots * kPubSlotBytes);
      for (int i = kPubSlots - 1; i >= 0; --i) P.free_slots.push_back(i);
    } else {
      (void)hipGetLastError();
    }
  }
  if (!P.slab || P.free_slots.empty()) return nullptr;  // (the caller falls back to the copy + synchronise read-back)
  const int sl = P.free_slots.back();
  P.free_slots.pop_back();
  return reinterpret_cast<double*>(P.slab + (size_t)sl * kPubSlotBytes);
}
void lm_pub_free(double* p) {
  if (!p) return;
  PubPool& P = pub_pool();
  std::lock_guard<std::mutex> g(P.m);
  P.free_slots.push_back((int)((reinterpret_cast<char*>(p) - P.slab) / kPubSlotBytes));
}

// Copies between PAGEABLE host memory and the device. Handed a pageable buffer of a megabyte or more, the HIP runtime
// page-locks it in place (a userptr registration with the kernel driver); when that memory is later freed or trimmed by
// the allocator, the driver's MMU notifier quiesces ALL queues of the process until a worker has revalidated them -
// measured here as a 15-25 ms stall of the NEXT call's first transfer, every few calls, depending on what malloc did with
// the set-up's temporaries. So nothing pageable ever reaches hipMemcpy: transfers of kStagedCopyMin bytes or more go through
// page-locked blocks of the pool above (smaller ones use the runtime's own staging buffers, which pin nothing).
namespace {
constexpr size_t kStagedCopyMin = (size_t)128 << 10, kStagedChunk = (size_t)32 << 20;
struct Parked { std::mutex m; std::unordered_map<hipStream_t, std::vector<void*>> by_stream; };
Parked& parked() { static Parked* p = new Parked; return *p; }
}  // namespace
// ---- batched small uploads (round 6) ---------------------------------------------------------------------------------
// A local-window session uploads ~50 small tables and clears ~8 small arrays while it is set up; every one of them was a
// hipMemcpyAsync / hipMemsetAsync of its own - 4-8 us of host time each, a third of a 0.8 ms set-up. While a batch is open on
// the calling thread, copies below kStagedCopyMin and clears below kBatchZeroMax on the batch's stream are only RECORDED
// (payload copied into one page-locked arena); the flush enqueues ONE host-to-device copy of the arena and ONE kernel
// that deals the segments to their destinations (a work-group per 16 KB piece). Stream order is kept as long as nothing
// that reads the destinations is enqueued before the flush - the host-only set-up path of small problems (session_build.hip).
namespace {
struct UploadSeg { unsigned long long dst; long long src; unsigned bytes; unsigned pad; };  // src < 0: zeros
constexpr size_t kBatchArenaDefault = (size_t)2 << 20, kBatchZeroMax = (size_t)1 << 20, kBatchPiece = (size_t)16 << 10;
size_t g_batch_arena = kBatchArenaDefault;  // (mavba_debug_upload_batch shrinks it: the arena-full path)
#define kBatchArena g_batch_arena
struct UploadBatch {
  hipStream_t st = nullptr;
  char* host = nullptr;
  size_t used = 0;
  std::vector<UploadSeg> segs;
};
thread_local UploadBatch* g_batch = nullptr;
struct ParkedDev { std::mutex m; std::unordered_map<hipStream_t, std::vector<void*>> by_stream; };
ParkedDev& parked_dev() { static ParkedDev* p = new ParkedDev; return *p; }

__global__ void __launch_bounds__(256) k_upload_scatter(const char* __restrict__ arena, const UploadSeg* __restrict__ segs) {
  const UploadSeg s = segs[blockIdx.x];
  unsigned* d = reinterpret_cast<unsigned*>(s.dst);
  const unsigned words = s.bytes >> 2;
  if (s.src < 0) {
    for (unsigned i = threadIdx.x; i < words; i += 256) d[i] = 0u;
    if (threadIdx.x < (s.bytes & 3u)) reinterpret_cast<unsigned char*>(d + words)[threadIdx.x] = 0;
  } else {
    const unsigned* a = reinterpret_cast<const unsigned*>(arena + s.src);
    for (unsigned i = threadIdx.x; i < words; i += 256) d[i] = a[i];
    if (threadIdx.x < (s.bytes & 3u))
      reinterpret_cast<unsigned char*>(d + words)[threadIdx.x] = reinterpret_cast<const unsigned char*>(a + words)[threadIdx.x];
  }
}

// enqueue what the batch holds (the batch stays open, empty)
hipError_t batch_emit(UploadBatch& B) {
  if (B.segs.empty()) return hipSuccess;
  const size_t table_off = (B.used + 15) & ~(size_t)15, table_bytes = B.segs.size() * sizeof(UploadSeg);
  // (the arena has kBatchArena of payload + room for the table of the pieces that fit into it)
  std::memcpy(B.host + table_off, B.segs.data(), table_bytes);
  void* dev = nullptr;
  hipError_t e = device_alloc(&dev, table_off + table_bytes);
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(dev, B.host, table_off + table_bytes, hipMemcpyHostToDevice, B.st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_upload_scatter, dim3((unsigned)B.segs.size()), dim3(256), 0, B.st, static_cast<const char*>(dev),
                       reinterpret_cast<const UploadSeg*>(static_cast<const char*>(dev) + table_off));
    e = hipGetLastError();
  }
  // both arenas are in use until the stream has passed the kernel: parked, freed by release_staged
  { std::lock_guard<std::mutex> g(parked().m); parked().by_stream[B.st].push_back(B.host); }
  { std::lock_guard<std::mutex> g(parked_dev().m); parked_dev().by_stream[B.st].push_back(dev); }
  B.host = nullptr; B.used = 0; B.segs.clear();
  return e;
}
size_t batch_table_room() { return (kBatchArena / 256 + kBatchArena / kBatchPiece + 64) * sizeof(UploadSeg); }
// records one copy (src != nullptr) or clear; false = not taken (no arena): the caller issues it directly
bool batch_add(UploadBatch& B, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return true;
  if ((reinterpret_cast<uintptr_t>(dst) & 3u) != 0) return false;
  const size_t need = src ? ((bytes + 15) & ~(size_t)15) : 0;
  if (need > kBatchArena) return false;  // (does not fit an arena at all)
  const size_t max_segs = batch_table_room() / sizeof(UploadSeg);
  const size_t pieces = (bytes + kBatchPiece - 1) / kBatchPiece;
  // a destination written twice (a buffer freed and handed out again inside the batch): the pieces of one flush run side by
  // side, so the earlier write has to leave first
  bool overlap = false;
  const unsigned long long d0 = reinterpret_cast<unsigned long long>(dst), d1 = d0 + bytes;
  for (const UploadSeg& s : B.segs) overlap = overlap || (s.dst < d1 && d0 < s.dst + s.bytes);
  if (B.host && (overlap || B.used + need > kBatchArena || B.segs.size() + pieces > max_segs))
    if (batch_emit(B) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (!B.host) {
    void* h = nullptr;
    if (pinned_alloc(&h, kBatchArena + 16 + batch_table_room()) != hipSuccess) { (void)hipGetLastError(); return false; }
    B.host = static_cast<char*>(h);
    B.used = 0;
  }
  long long at = -1;
  if (src) { at = (long long)B.used; std::memcpy(B.host + B.used, src, bytes); B.used += need; }
  for (size_t off = 0; off < bytes; off += kBatchPiece)
    B.segs.push_back(UploadSeg{reinterpret_cast<unsigned long long>(static_cast<char*>(dst) + off), src ? at + (long long)off : -1,
                               (unsigned)std::min(kBatchPiece, bytes - off), 0u});
  return true;
}
}  // namespace
bool upload_batch_begin(hipStream_t st) {
  if (g_batch) return false;
  g_batch = new UploadBatch;
  g_batch->st = st;
  return true;
}
hipError_t upload_batch_end(bool emit) {
  if (!g_batch) return hipSuccess;
  UploadBatch* B = g_batch;
  g_batch = nullptr;
  hipError_t e = hipSuccess;
  if (emit) e = batch_emit(*B);
  if (B->host) pinned_free(B->host);  // (aborted, or nothing recorded)
  delete B;
  return e;
}
void upload_batch_debug_arena(size_t bytes) { g_batch_arena = bytes ? bytes : kBatchArenaDefault; }
// An operation the open batch does not take is issued directly - BEHIND what the batch holds (stream order as the caller wrote it).
static void batch_flush_before_direct(hipStream_t st) {
  if (g_batch && g_batch->st == st) (void)batch_emit(*g_batch);
}
hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
  if (g_batch && g_batch->st == st && bytes <= kBatchZeroMax && batch_add(*g_batch, p, nullptr, bytes)) return hipSuccess;
  batch_flush_before_direct(st);
  return hipMemsetAsync(p, 0, bytes, st);
}

hipError_t copy_h2d_staged(void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (bytes < kStagedCopyMin) {
    if (g_batch && g_batch->st == st && batch_add(*g_batch, dst, src, bytes)) return hipSuccess;
    batch_flush_before_direct(st);
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
  }
  batch_flush_before_direct(st);
  for (size_t off = 0; off < bytes; off += kStagedChunk) {
    const size_t len = std::min(kStagedChunk, bytes - off);
    void* stage = nullptr;
    hipError_t e = pinned_alloc(&stage, len);
    if (e != hipSuccess) { (void)hipGetLastError(); return hipMemcpyAsync((char*)dst + off, (const char*)src + off, bytes - off, hipMemcpyHostToDevice, st); }
    std::memcpy(stage, (const char*)src + off, len);
    e = hipMemcpyAsync((char*)dst + off, stage, len, hipMemcpyHostToDevice, st);
    { std::lock_guard<std::mutex> g(parked().m); parked().by_stream[st].push_back(stage); }  // (freed by release_staged, behind a synchronisation)
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
void release_staged(hipStream_t st) {
  std::vector<void*> blocks;
  {
    std::lock_guard<std::mutex> g(parked().m);
    auto it = parked().by_stream.find(st);
    if (it != parked().by_stream.end()) blocks.swap(it->second);
  }
  for (void* b : blocks) pinned_free(b);
  std::vector<void*> dblocks;
  {
    std::lock_guard<std::mutex> g(parked_dev().m);
    auto it = parked_dev().by_stream.find(st);
    if (it != parked_dev().by_stream.end()) dblocks.swap(it->second);
  }
  for (void* b : dblocks) device_free(b);
}
hipError_t copy_d2h_staged_sync(void* dst, const void* src, size_t bytes, hipStream_t st) {
  if (bytes < kStagedCopyMin) {
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    return e != hipSuccess ? e : hipStreamSynchronize(st);
  }
  // all chunks in flight, one synchronisation, then the host copies
  std::vector<std::pair<void*, size_t>> stages;
  hipError_t e = hipSuccess;
  for (size_t off = 0; off < bytes && e == hipSuccess; off += kStagedChunk) {
    const size_t len = std::min(kStagedChunk, bytes - off);
    void* stage = nullptr;
    e = pinned_alloc(&stage, len);
    if (e != hipSuccess) break;
    stages.push_back({stage, len});
    e = hipMemcpyAsync(stage, (const char*)src + off, len, hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  size_t off = 0;
  for (auto& sg : stages) {
    if (e == hipSuccess) std::memcpy((char*)dst + off, sg.first, sg.second);
    off += sg.second;
    pinned_free(sg.first);
  }
  if (e != hipSuccess && stages.empty()) {  // no page-locked memory to be had: the plain copy
    (void)hipGetLastError();
    e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  return e;
}

// Big host scratch blocks (the set-up's per-observation temporaries) are cached as well: at most kScratchCacheBytes
// stay parked, blocks under 256 KiB go straight to malloc (its own free lists handle those without page faults).
namespace {
struct ScratchCache {
  std::mutex m;
  std::multimap<size_t, void*> free_blocks;  // capacity -> block
  std::unordered_map<void*, size_t> live;     // block -> capacity (only cached-class blocks)
  size_t parked = 0;
};
ScratchCache& scratch_cache() { static ScratchCache* c = new ScratchCache; return *c; }
constexpr size_t kScratchMin = (size_t)256 << 10, kScratchCacheBytes = (size_t)1 << 30;
}  // namespace
void* host_scratch_alloc(size_t bytes) {
  if (bytes < kScratchMin) { void* p = std::malloc(bytes); if (!p) throw std::bad_alloc(); return p; }
  ScratchCache& C = scratch_cache();
  {
    std::lock_guard<std::mutex> g(C.m);
    auto it = C.free_blocks.lower_bound(bytes);
    if (it != C.free_blocks.end() && it->first <= 2 * bytes + ((size_t)1 << 20)) {
      void* p = it->second;
      C.parked -= it->first;
      C.live[p] = it->first;
      C.free_blocks.erase(it);
      return p;
    }
  }
  const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
  void* p = std::malloc(cap);
  if (!p) throw std::bad_alloc();
  std::lock_guard<std::mutex> g(C.m);
  C.live[p] = cap;
  return p;
}
void host_scratch_free(void* p, size_t bytes) {
  if (!p) return;
  if (bytes < kScratchMin) { std::free(p); return; }
  ScratchCache& C = scratch_cache();
  {
    std::lock_guard<std::mutex> g(C.m);
    auto it = C.live.find(p);
    if (it != C.live.end()) {
      const size_t cap = it->second;
      C.live.erase(it);
      if (C.parked + cap <= kScratchCacheBytes) { C.free_blocks.insert({cap, p}); C.parked += cap; return; }
    }
  }
  std::free(p);
}

// The persistent factorisation needs all its work-groups resident; when a launch gives up (0.3 s), the session falls back
// to the launch-per-panel schedule - and so do the next kPersistCooldownSessions sessions of the process, without trying.
namespace {
constexpr int kPersistCooldownSessions = 64;
std::atomic<int> g_persist_cooldown{0};
}  // namespace
bool persistent_allowed_now() {
  int c = g_persist_cooldown.load(std::memory_order_relaxed);
  while (c > 0 && !g_persist_cooldown.compare_exchange_weak(c, c - 1, std::memory_order_relaxed)) {}
  return c <= 0;
}
bool persistent_in_cooldown() { return g_persist_cooldown.load(std::memory_order_relaxed) > 0; }  // (does not count a session)
void persistent_timed_out() { g_persist_cooldown.store(kPersistCooldownSessions, std::memory_order_relaxed); }

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
  release_staged(st);
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
  std::exception_ptr error;  // the first exception a body threw in the current run()

  void loop(int id) {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_start.wait(lk, [&] { return generation != seen; });
      seen = generation;
      if (id < job_T) {
        const std::function<void(int)>* j = job;
        lk.unlock();
        std::exception_ptr e;
        try { (*j)(id); } catch (...) { e = std::current_exception(); }  // (rethrown by run() on the calling thread)
        lk.lock();
        if (e && !error) error = e;
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
    std::exception_ptr e = error;
    error = nullptr;
    lk.unlock();
    if (e) std::rethrow_exception(e);  // (a Failure thrown inside a parallel pass reaches the C ABI's catch like any other)
  }
};
int host_threads() {
  static const int n = [] {
    const char* e = std::getenv("MAVBA_HOST_THREADS");
    const int cap = e ? std::max(1, std::atoi(e)) : 16;
    return (int)std::min<size_t>((size_t)cap, std::max(1u, std::thread::hardware_concurrency()));
  }();
  return n;
}
void host_run(int T, const std::function<void(int)>& body) {
  if (T <= 1) { body(0); return; }
  static HostWorkers* W = new HostWorkers(host_threads());  // never destroyed: the workers outlive static destruction
  W->run(std::min(T, W->size()), body);
}

}  // namespace mavba
