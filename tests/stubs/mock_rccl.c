/* A stand-in for librccl.so with the five entry points the library loads (tests only, MAVBA_RCCL_LIB): lets the in-process
 * rank protocol of csrc/multi_gpu.hip - one creating thread per rank, the process-wide communicator group, the abort of
 * the group when a rank fails - run on a box without GPUs. ncclCommInitRank blocks until all ranks of the id have arrived
 * (as the real one does: a caller that created the communicators one after the other would time out here). */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct MockComm { int rank, world; long long id; int aborted; } MockComm;
typedef MockComm* ncclComm_t;
typedef int ncclResult_t;

static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
static long long next_id = 1, n_init = 0, n_destroy = 0, n_abort = 0, n_allreduce = 0, n_timeout = 0;
static long long arrived_id = 0;
static int arrived = 0;
static unsigned long long generation = 0;

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  pthread_mutex_lock(&mu);
  memset(id, 0, sizeof(*id));
  memcpy(id->internal, &next_id, sizeof(next_id));
  ++next_id;
  pthread_mutex_unlock(&mu);
  return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  long long v;
  memcpy(&v, id.internal, sizeof(v));
  pthread_mutex_lock(&mu);
  if (arrived == 0) arrived_id = v;
  if (arrived_id != v) { pthread_mutex_unlock(&mu); return 5; }
  const unsigned long long g = generation;
  if (++arrived == nranks) { arrived = 0; ++generation; pthread_cond_broadcast(&cv); }
  else {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += 5;
    while (generation == g)
      if (pthread_cond_timedwait(&cv, &mu, &ts) != 0) { ++n_timeout; --arrived; pthread_mutex_unlock(&mu); return 1; }
  }
  ++n_init;
  pthread_mutex_unlock(&mu);
  MockComm* c = (MockComm*)calloc(1, sizeof(MockComm));
  c->rank = rank; c->world = nranks; c->id = v;
  *comm = c;
  return 0;
}

ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, int dt, int op, ncclComm_t c, void* stream) {
  (void)s; (void)r; (void)n; (void)dt; (void)op; (void)stream;
  pthread_mutex_lock(&mu); ++n_allreduce; pthread_mutex_unlock(&mu);
  return c && !c->aborted ? 0 : 1;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { pthread_mutex_lock(&mu); ++n_destroy; pthread_mutex_unlock(&mu); free(c); return 0; }
ncclResult_t ncclCommAbort(ncclComm_t c) { pthread_mutex_lock(&mu); ++n_abort; pthread_mutex_unlock(&mu); free(c); return 0; }
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "ok" : "mock rccl error"; }

/* init, destroy, abort, allreduce, time-outs */
void mock_rccl_counts(long long* out5) {
  pthread_mutex_lock(&mu);
  out5[0] = n_init; out5[1] = n_destroy; out5[2] = n_abort; out5[3] = n_allreduce; out5[4] = n_timeout;
  pthread_mutex_unlock(&mu);
}
