// host_util.hip - process-wide helpers of the host side: caching device allocator, stream cache, worker threads.
#include "session.h"

namespace mavba {

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
int host_threads() {
  static const int n = (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
  return n;
}
void host_run(int T, const std::function<void(int)>& body) {
  if (T <= 1) { body(0); return; }
  static HostWorkers* W = new HostWorkers(host_threads());  // never destroyed: the workers outlive static destruction
  W->run(std::min(T, W->size()), body);
}

}  // namespace mavba
