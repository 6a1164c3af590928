// Context, communicator, mesh/table upload, device vectors and matrices of libpynama_hip.so.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstring>

#include "pyn_internal.h"

// layout of the shared-memory TEST transport (see pyn_comm_init_shm below)
struct pyn_shm_hdr {
  std::atomic<int> count;    // barrier of the all-reduces (host thread of every rank)
  std::atomic<int> sense;
  int nranks;
  int64_t cap;   // outbox capacity per rank, bytes
  std::atomic<int> hcount;   // barrier of the halo exchanges (stream callbacks): its own object, so that an exchange in flight on the
  std::atomic<int> hsense;   // communication stream and an all-reduce on the main stream can never pair up with each other
};
struct pyn_shm_comm {
  unsigned char* base = nullptr;
  size_t size = 0;
  int rank = 0, nranks = 1, local_sense = 0, halo_sense = 0;
  int64_t cap = 0;
  // asynchronous exchange: pinned staging per stream slot (0: main stream, 1: communication stream); a failure inside a stream
  // callback is parked here and reported by the next call that can return it
  double* stage_out[2] = {nullptr, nullptr};
  double* stage_in[2] = {nullptr, nullptr};
  size_t stage_in_cap[2] = {0, 0};
  std::atomic<int> failed{0};
  char fail_msg[256] = "";
  pyn_shm_hdr* hdr() const { return reinterpret_cast<pyn_shm_hdr*>(base); }
  double* ar(int r) const { return reinterpret_cast<double*>(base + 4096) + (size_t)r * 64; }
  int64_t* dir(int src) const { return reinterpret_cast<int64_t*>(base + 4096 + (size_t)nranks * 512) + (size_t)src * nranks * 2; }
  unsigned char* outbox(int r) const { return base + 4096 + (size_t)nranks * 512 + (size_t)nranks * nranks * 16 + (size_t)r * cap; }
};


// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void pyn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pyn_last_error(void) { return g_err; }
extern "C" int pyn_version(void) { return 101; }
#ifndef PYN_SRC_HASH
#define PYN_SRC_HASH "unknown"
#endif
extern "C" const char* pyn_source_hash(void) { return PYN_SRC_HASH; }

extern "C" int pyn_device_count(int* count) {
  PYN_CHECK(count, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return PYN_OK;
}

template <typename T>
static int dev_upload(T** dst, const T* src, size_t n, hipStream_t s) {
  if (*dst) {
    PYN_HIP(hipFree(*dst));
    *dst = nullptr;
  }
  if (n == 0) return PYN_OK;
  PYN_HIP(hipMalloc((void**)dst, n * sizeof(T)));
  if (src) PYN_HIP(hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, s));
  return PYN_OK;
}

extern "C" int pyn_ctx_create(int device, pyn_ctx** out) {
  PYN_CHECK(out, "out is NULL");
  int n = 0;
  PYN_TRY(pyn_device_count(&n));
  if (n <= 0) {
    pyn_set_error("no HIP device visible: libpynama_hip.so has no CPU fallback");
    return PYN_ENOGPU;
  }
  PYN_CHECK(device >= 0 && device < n, "device %d out of range [0,%d)", device, n);
  PYN_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  PYN_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    pyn_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return PYN_ENOGPU;
  }
  pyn_ctx* c = new pyn_ctx();
  c->device = device;
  PYN_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  PYN_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  PYN_HIP(hipEventCreate(&c->ev0));
  PYN_HIP(hipEventCreate(&c->ev1));
  PYN_HIP(hipEventCreateWithFlags(&c->ev_vec, hipEventDisableTiming));
  PYN_HIP(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
  PYN_HIP(hipMalloc((void**)&c->d_part, 8 * PYN_MAX_PARTIALS * sizeof(double)));
  PYN_HIP(hipMalloc((void**)&c->d_scal, 64 * sizeof(double)));
  PYN_HIP(hipMalloc((void**)&c->d_flag, 8 * sizeof(int)));
  PYN_HIP(hipMemset(c->d_scal, 0, 64 * sizeof(double)));
  PYN_HIP(hipMemset(c->d_flag, 0, 8 * sizeof(int)));
  PYN_HIP(hipHostMalloc((void**)&c->h_scal, 64 * sizeof(double), hipHostMallocDefault));
  PYN_HIP(hipHostMalloc((void**)&c->h_flag, 8 * sizeof(int), hipHostMallocDefault));
  *out = c;
  return PYN_OK;
}

static void free_quad(QuadTab& q) {
  (void)hipFree(q.w);
  (void)hipFree(q.H);
  (void)hipFree(q.Hrs);
  (void)hipFree(q.HrsCoo);
  q = QuadTab();
}

extern "C" int pyn_ctx_destroy(pyn_ctx* c) {
  if (!c) return PYN_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm_halo && c->comm_halo != c->comm) ncclCommDestroy(c->comm_halo);
  if (c->comm) ncclCommDestroy(c->comm);
  if (c->shm) {
    (void)hipStreamSynchronize(c->comm_stream);
    for (int k = 0; k < 2; ++k) {
      (void)hipHostFree(c->shm->stage_out[k]);
      (void)hipHostFree(c->shm->stage_in[k]);
    }
    munmap(c->shm->base, c->shm->size);
    delete c->shm;
    c->shm = nullptr;
  }
  for (auto& m : c->mats) {
    (void)hipFree(m.val);
    (void)hipFree(m.sell_val);
    (void)hipFree(m.dinv);
    m.release_lu();
    pyn_rhs_release(m);
  }
  (void)hipFree(c->d_esel);
  pyn_sell_drop_structure(c);
  for (auto& v : c->vecs) (void)hipFree(v.d);
  for (int k = 0; k < 3; ++k) (void)hipFree(c->mf_mask[k]);
  for (auto& q : c->quad) free_quad(q);
  (void)hipFree(c->d_conn);
  (void)hipFree(c->d_xyz);
  (void)hipFree(c->d_aff);
  (void)hipFree(c->lat.d_P);
  (void)hipFree(c->lat.d_zord);
  pyn_ho3_release(c);
  (void)hipFree(c->d_ho3_tabs);
  (void)hipFree(c->d_ho3_t1d);
  (void)hipFree(c->d_bcmask);
  (void)hipFree(c->d_rowptr);
  (void)hipFree(c->d_colidx);
  (void)pyn_patch_plan_set_kind(c, 0, 0, nullptr, nullptr);
  (void)pyn_patch_plan_set_kind(c, 1, 0, nullptr, nullptr);
  (void)hipFree(c->d_send_idx);
  (void)hipFree(c->d_send_buf);
  (void)hipFree(c->d_part);
  (void)hipFree(c->d_scal);
  (void)hipFree(c->d_flag);
  (void)hipFree(c->d_work);
  (void)hipFree(c->d_eloc);
  (void)hipFree(c->d_kle_lel);
  (void)hipHostFree(c->h_scal);
  (void)hipHostFree(c->h_flag);
  for (auto e : c->prof_ev) (void)hipEventDestroy(e);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipEventDestroy(c->ev_vec);
  (void)hipEventDestroy(c->ev_halo);
  (void)hipStreamDestroy(c->comm_stream);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return PYN_OK;
}

extern "C" int pyn_sync(pyn_ctx* c) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

int pyn_ensure_work(pyn_ctx* c, size_t bytes) {
  if (bytes <= c->work_bytes) return PYN_OK;
  if (c->d_work) PYN_HIP(hipFree(c->d_work));
  c->d_work = nullptr;
  PYN_HIP(hipMalloc((void**)&c->d_work, bytes));
  c->work_bytes = bytes;
  return PYN_OK;
}

// ---------------------------------------------------------------------------------------------
// communicator
// Shared-memory TEST transport: ranks are processes that may share one GPU; every exchange is staged through a POSIX
// shared-memory file (device -> host copy, process barrier, host -> device copy).  Slow by construction; it exists so that
// the distributed solver can be run end to end with world_size > 1 on a one-GPU box, where RCCL refuses duplicate devices.
static bool shm_barrier_on(std::atomic<int>& count, std::atomic<int>& sense, int& local, int nranks) {
  local ^= 1;
  if (count.fetch_add(1, std::memory_order_acq_rel) == nranks - 1) {
    count.store(0, std::memory_order_relaxed);
    sense.store(local, std::memory_order_release);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (sense.load(std::memory_order_acquire) != local) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
    sched_yield();
  }
  return true;
}

static int shm_barrier(pyn_shm_comm* m) {
  pyn_shm_hdr* h = m->hdr();
  PYN_CHECK(shm_barrier_on(h->count, h->sense, m->local_sense, m->nranks), "shared-memory barrier: a rank did not arrive within 120 s");
  return PYN_OK;
}

// One halo exchange of the test transport, host side, run as a STREAM CALLBACK (hipLaunchHostFunc) between the device -> host copy
// of the packed send buffer and the host -> device copies of the ghosts: the stream that carries the exchange stalls in here until
// every rank has published its outbox, while the other stream of the process keeps computing -- the overlapped CG really races its
// interior product against the exchange, as it does over RCCL.  No HIP call in here.
struct ShmExchange {
  pyn_shm_comm* m;
  int slot, bs;
  size_t send_bytes;
  std::vector<int> neigh;
  std::vector<int64_t> send_ptr, recv_ptr;
};

static void shm_exchange_host(void* arg) {
  ShmExchange* e = static_cast<ShmExchange*>(arg);
  pyn_shm_comm* m = e->m;
  pyn_shm_hdr* h = m->hdr();
  auto fail = [&](const char* what) {
    if (!m->failed.exchange(1)) snprintf(m->fail_msg, sizeof(m->fail_msg), "shared-memory halo exchange: %s", what);
  };
  if (e->send_bytes) memcpy(m->outbox(m->rank), m->stage_out[e->slot], e->send_bytes);
  for (size_t k = 0; k < e->neigh.size(); ++k) {
    int64_t* d = m->dir(m->rank) + (size_t)e->neigh[k] * 2;
    d[0] = e->send_ptr[k] * e->bs;
    d[1] = (e->send_ptr[k + 1] - e->send_ptr[k]) * e->bs;
  }
  if (!shm_barrier_on(h->hcount, h->hsense, m->halo_sense, m->nranks)) fail("a rank did not arrive within 120 s");
  for (size_t k = 0; k < e->neigh.size(); ++k) {
    const int src = e->neigh[k];
    const int64_t* d = m->dir(src) + (size_t)m->rank * 2;
    const int64_t nr = (e->recv_ptr[k + 1] - e->recv_ptr[k]) * e->bs;
    if (d[1] != nr) {
      fail("halo plan mismatch between two ranks");
      continue;
    }
    if (nr) memcpy(m->stage_in[e->slot] + e->recv_ptr[k] * e->bs, reinterpret_cast<const double*>(m->outbox(src)) + d[0], (size_t)nr * sizeof(double));
  }
  if (!shm_barrier_on(h->hcount, h->hsense, m->halo_sense, m->nranks)) fail("a rank did not arrive within 120 s");   // outboxes may be rewritten
  delete e;
}

extern "C" int pyn_comm_init_shm(pyn_ctx* c, int rank, int nranks, const char* path, int64_t cap_bytes) {
  PYN_CHECK(c && path, "NULL argument");
  PYN_CHECK(nranks >= 1 && rank >= 0 && rank < nranks && nranks <= 16, "bad rank %d / %d", rank, nranks);
  PYN_CHECK(cap_bytes > 0, "outbox capacity must be positive");
  const size_t size = 4096 + (size_t)nranks * 512 + (size_t)nranks * nranks * 16 + (size_t)nranks * cap_bytes;
  const int fd = open(path, O_RDWR);
  PYN_CHECK(fd >= 0, "cannot open %s (rank 0's launcher creates and sizes it)", path);
  struct stat st;
  PYN_CHECK(fstat(fd, &st) == 0 && (size_t)st.st_size >= size, "%s is smaller than %zu bytes", path, size);
  void* base = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  PYN_CHECK(base != MAP_FAILED, "mmap of %s failed", path);
  auto* m = new pyn_shm_comm();
  m->base = static_cast<unsigned char*>(base);
  m->size = size;
  m->rank = rank;
  m->nranks = nranks;
  m->cap = cap_bytes;
  c->shm = m;
  c->rank = rank;
  c->nranks = nranks;
  return shm_barrier(m);   // the file arrives zero-filled: count = sense = 0
}

int pyn_allreduce_dev(pyn_ctx* c, double* dbuf, int n, int op, hipStream_t st) {
  if (c->comm) {
    PYN_NCCL(ncclAllReduce(dbuf, dbuf, n, ncclDouble, op == 1 ? ncclMax : ncclSum, c->comm, st));
    return PYN_OK;
  }
  if (!c->shm) return PYN_OK;
  pyn_shm_comm* m = c->shm;
  PYN_CHECK(!m->failed.load(), "%s", m->fail_msg);
  PYN_CHECK(n <= 64, "shared-memory all-reduce: at most 64 values");
  double loc[64];
  PYN_HIP(hipMemcpyAsync(loc, dbuf, n * sizeof(double), hipMemcpyDeviceToHost, st));
  PYN_HIP(hipStreamSynchronize(st));
  memcpy(m->ar(m->rank), loc, n * sizeof(double));
  PYN_TRY(shm_barrier(m));
  for (int i = 0; i < n; ++i) {   // same order on every rank: identical results
    double r = m->ar(0)[i];
    for (int k = 1; k < m->nranks; ++k) r = op == 1 ? fmax(r, m->ar(k)[i]) : r + m->ar(k)[i];
    loc[i] = r;
  }
  PYN_TRY(shm_barrier(m));
  PYN_HIP(hipMemcpyAsync(dbuf, loc, n * sizeof(double), hipMemcpyHostToDevice, st));
  PYN_HIP(hipStreamSynchronize(st));
  return PYN_OK;
}

extern "C" int pyn_comm_unique_id(void* out, int nbytes) {
  PYN_CHECK(out && nbytes >= (int)sizeof(ncclUniqueId), "need >= %d bytes", (int)sizeof(ncclUniqueId));
  ncclUniqueId id;
  PYN_NCCL(ncclGetUniqueId(&id));
  memcpy(out, &id, sizeof(id));
  return PYN_OK;
}

extern "C" int pyn_comm_init(pyn_ctx* c, int rank, int nranks, const void* uid, int nbytes) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d / %d", rank, nranks);
  c->rank = rank;
  c->nranks = nranks;
  if (nranks == 1 && !uid) return PYN_OK;  // serial: no communicator at all
  if (!uid) {  // detached: this rank's slab is processed in isolation (tests, staged pipelines)
    c->detached = true;
    return PYN_OK;
  }
  PYN_CHECK(nbytes >= (int)sizeof(ncclUniqueId), "unique id too short");
  ncclUniqueId id;
  memcpy(&id, uid, sizeof(id));
  PYN_HIP(hipSetDevice(c->device));
  PYN_NCCL(ncclCommInitRank(&c->comm, nranks, id, rank));
  // a SECOND communicator for the halo exchanges: the overlapped CG issues them on the communication stream while an all-reduce of
  // the previous step may still be queued on the main stream -- with their own communicator nothing rests on how RCCL orders the
  // operations of ONE communicator across streams
  // (a collective: it succeeds or fails on every rank alike.  Should this RCCL refuse it, the exchanges share the first communicator as
  // they did in round 2 -- correct as long as RCCL keeps one communicator's operations in issue order -- and say so on stderr)
  c->comm_halo = nullptr;
  const ncclResult_t sr = getenv("PYNAMA_NO_COMM_SPLIT") ? ncclInvalidUsage : ncclCommSplit(c->comm, 0, rank, &c->comm_halo, nullptr);
  if (sr != ncclSuccess || !c->comm_halo) {
    fprintf(stderr, "[pynama_hip] rank %d: no second communicator for the halo exchanges (%s): sharing the first one\n", rank,
            sr != ncclSuccess ? ncclGetErrorString(sr) : "ncclCommSplit returned none");
    c->comm_halo = c->comm;
  }
  return PYN_OK;
}

extern "C" int pyn_comm_allreduce_f64(pyn_ctx* c, double* inout, int n, int op) {
  PYN_CHECK(c && inout && n > 0 && n <= 32, "bad arguments");
  if (c->nranks == 1 && !pyn_has_comm(c)) return PYN_OK;
  PYN_CHECK(!c->detached, "detached communicator: no collectives");
  PYN_HIP(hipMemcpyAsync(c->d_scal + 32, inout, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  PYN_TRY(pyn_allreduce_dev(c, c->d_scal + 32, n, op, c->stream));
  PYN_HIP(hipMemcpyAsync(inout, c->d_scal + 32, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

extern "C" int pyn_comm_barrier(pyn_ctx* c) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_HIP(hipStreamSynchronize(c->stream));
  double one = 1.0;
  return pyn_comm_allreduce_f64(c, &one, 1, 0);
}

__global__ void selftest_fill_kernel(double* x, int64_t n_owned, int64_t n_all, double stamp) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = i < n_owned ? stamp : -1.0;
}

// Start-up self-test of the communicator (bench.py --gpus N runs it before anything is timed, under a host watchdog that
// names the phase and exits if a collective never returns): (1) what RCCL thinks the world is, (2) all-reduce of 1 and of
// the rank, (3) one halo exchange of a rank-stamped vector on the main stream, checked on the receiver, (4) the same on the
// communication stream with the event ordering the overlapped CG uses.  info: [0] ranks seen by RCCL, [1] sum of ones,
// [2] sum of ranks, [3] ghosts checked (main stream), [4] ghosts checked (communication stream).
extern "C" int pyn_comm_selftest(pyn_ctx* c, double* info, int ninfo) {
  PYN_CHECK(c && info && ninfo >= 5, "bad arguments");
  for (int i = 0; i < ninfo; ++i) info[i] = 0.0;
  PYN_CHECK(pyn_has_comm(c), "self-test needs a communicator");
  PYN_HIP(hipSetDevice(c->device));
  int seen = c->nranks;
  if (c->comm) {
    PYN_NCCL(ncclCommCount(c->comm, &seen));
    int seen_h = 0, rank_h = -1;
    PYN_NCCL(ncclCommCount(c->comm_halo, &seen_h));
    PYN_NCCL(ncclCommUserRank(c->comm_halo, &rank_h));
    PYN_CHECK(seen_h == seen && rank_h == c->rank, "halo communicator: %d ranks / rank %d, want %d / %d", seen_h, rank_h, seen, c->rank);
  }
  info[0] = c->comm ? seen : 0;      // ranks counted by RCCL itself; 0 = the shared-memory test transport is in use
  if (ninfo >= 7) info[6] = c->comm ? (c->comm_halo != c->comm ? 1.0 : 2.0) : 0.0;   // halo exchanges: 1 own communicator, 2 shared
  PYN_CHECK(seen == c->nranks, "RCCL communicator has %d ranks, the launcher declared %d", seen, c->nranks);
  double v[2] = {1.0, (double)c->rank};
  PYN_TRY(pyn_comm_allreduce_f64(c, v, 2, 0));
  info[1] = v[0];
  info[2] = v[1];
  const double want = 0.5 * c->nranks * (c->nranks - 1);
  PYN_CHECK(v[0] == (double)c->nranks && v[1] == want, "all-reduce self-test: sum(1) = %g (want %d), sum(rank) = %g (want %g)", v[0],
            c->nranks, v[1], want);
  if (c->neigh.empty() || !c->halo_set) return PYN_OK;
  const int64_t na = c->n_owned + c->n_ghost;
  DevTmp x;
  PYN_HIP(x.alloc((size_t)na * sizeof(double)));
  std::vector<double> ghosts((size_t)c->n_ghost);
  for (int pass = 0; pass < 3; ++pass) {
    const double base = 1.0 + 100.0 * pass;   // stamp = base + rank
    selftest_fill_kernel<<<1024, 256, 0, c->stream>>>(x.as<double>(), c->n_owned, na, base + c->rank);
    if (pass == 0) {
      PYN_TRY(pyn_halo_exchange(c, x.as<double>(), 1));
    } else {   // as in solve_cg_sr: vector ready -> exchange on the communication stream -> main stream waits for the ghosts
      PYN_HIP(hipEventRecord(c->ev_vec, c->stream));
      PYN_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_vec, 0));
      PYN_TRY(pyn_halo_exchange_on(c, x.as<double>(), 1, c->comm_stream));
      PYN_HIP(hipEventRecord(c->ev_halo, c->comm_stream));
      if (pass == 2) {   // one overlapped step: an all-reduce queued on the MAIN stream while the exchange is in flight on the other
        const double mine[2] = {1.0, (double)(c->rank + 1)};
        PYN_HIP(hipMemcpyAsync(c->d_scal + 40, mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
        PYN_TRY(pyn_allreduce_dev(c, c->d_scal + 40, 2, 0, c->stream));
      }
      PYN_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
      if (pass == 2) {
        double got[2] = {0.0, 0.0};
        PYN_HIP(hipMemcpyAsync(got, c->d_scal + 40, sizeof(got), hipMemcpyDeviceToHost, c->stream));
        PYN_HIP(hipStreamSynchronize(c->stream));
        const double w2 = 0.5 * c->nranks * (c->nranks + 1);
        PYN_CHECK(got[0] == (double)c->nranks && got[1] == w2, "all-reduce beside an exchange in flight: (%g, %g), want (%d, %g)", got[0], got[1],
                  c->nranks, w2);
        if (ninfo >= 6) info[5] = got[1];
      }
    }
    PYN_HIP(hipMemcpyAsync(ghosts.data(), x.as<double>() + c->n_owned, (size_t)c->n_ghost * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PYN_HIP(hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < c->neigh.size(); ++k)
      for (int64_t j = c->recv_ptr[k]; j < c->recv_ptr[k + 1]; ++j)
        PYN_CHECK(ghosts[(size_t)j] == base + c->neigh[k], "halo self-test (%s): ghost %lld from rank %d holds %g, want %g",
                  pass == 0 ? "main stream" : pass == 1 ? "communication stream" : "communication stream beside an all-reduce", (long long)j,
                  c->neigh[k], ghosts[(size_t)j], base + c->neigh[k]);
    if (pass < 2) info[3 + pass] = (double)c->n_ghost;
  }
  return PYN_OK;
}

extern "C" int pyn_halo_set(pyn_ctx* c, int64_t n_owned, int64_t n_ghost, int n_neigh, const int32_t* neigh,
                            const int64_t* send_ptr, const int32_t* send_idx, const int64_t* recv_ptr) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_CHECK(n_owned >= 0 && n_ghost >= 0 && n_neigh >= 0, "negative sizes");
  PYN_CHECK(c->n_node == 0, "pyn_halo_set must precede pyn_mesh_set");
  c->n_owned = n_owned;
  c->n_ghost = n_ghost;
  c->neigh.assign(neigh, neigh + n_neigh);
  c->send_ptr.assign(send_ptr, send_ptr + n_neigh + 1);
  c->recv_ptr.assign(recv_ptr, recv_ptr + n_neigh + 1);
  PYN_CHECK(c->recv_ptr[n_neigh] == n_ghost, "recv_ptr does not cover the ghosts");
  c->n_send = c->send_ptr[n_neigh];
  for (int k = 0; k < n_neigh; ++k) PYN_CHECK(neigh[k] >= 0 && neigh[k] < c->nranks, "bad neighbour");
  for (int64_t i = 0; i < c->n_send; ++i) PYN_CHECK(send_idx[i] >= 0 && send_idx[i] < n_owned, "send_idx out of range");
  PYN_TRY(dev_upload(&c->d_send_idx, send_idx, (size_t)c->n_send, c->stream));
  if (c->d_send_buf) PYN_HIP(hipFree(c->d_send_buf));
  c->d_send_buf = nullptr;
  if (c->n_send) PYN_HIP(hipMalloc((void**)&c->d_send_buf, (size_t)c->n_send * 6 * sizeof(double)));
  PYN_HIP(hipStreamSynchronize(c->stream));
  c->halo_set = true;
  return PYN_OK;
}

__global__ void pack_send_kernel(const double* __restrict__ x, const int32_t* __restrict__ idx, double* __restrict__ buf,
                                 int64_t n, int bs) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < n * bs; t += stride) {
    int64_t i = t / bs;
    int q = (int)(t - i * bs);
    buf[t] = x[(int64_t)idx[i] * bs + q];
  }
}

int pyn_halo_exchange(pyn_ctx* c, double* x, int bs) { return pyn_halo_exchange_on(c, x, bs, c->stream); }

int pyn_halo_exchange_on(pyn_ctx* c, double* x, int bs, hipStream_t st) {
  if (c->neigh.empty()) return PYN_OK;
  if (c->detached) return PYN_OK;  // ghost entries were written by the caller (pyn_vec_set_local_host)
  PYN_CHECK(pyn_has_comm(c), "halo exchange needs a communicator (pyn_comm_init with a unique id)");
  PYN_CHECK(bs <= 6, "block size too large for the halo buffer");
  if (c->n_send) {
    int64_t tot = c->n_send * bs;
    int grid = (int)std::min<int64_t>((tot + 255) / 256, 1024);
    pack_send_kernel<<<grid, 256, 0, st>>>(x, c->d_send_idx, c->d_send_buf, c->n_send, bs);
  }
  if (c->shm) {   // test transport, stream ordered like the RCCL one: device -> pinned host, host callback (outbox, barrier, inbox), host -> device
    pyn_shm_comm* m = c->shm;
    PYN_CHECK(!m->failed.load(), "%s", m->fail_msg);
    const size_t bytes = (size_t)c->n_send * bs * sizeof(double);
    PYN_CHECK((int64_t)bytes <= m->cap, "halo (%zu bytes) exceeds the shared-memory outbox", bytes);
    const int slot = st == c->comm_stream ? 1 : 0;
    if (!m->stage_out[slot]) PYN_HIP(hipHostMalloc((void**)&m->stage_out[slot], (size_t)m->cap, hipHostMallocDefault));
    const size_t in_bytes = (size_t)c->n_ghost * bs * sizeof(double);
    if (in_bytes > m->stage_in_cap[slot]) {
      PYN_HIP(hipStreamSynchronize(st));   // an earlier exchange of this slot may still be copying from the old buffer
      if (m->stage_in[slot]) PYN_HIP(hipHostFree(m->stage_in[slot]));
      m->stage_in[slot] = nullptr;
      PYN_HIP(hipHostMalloc((void**)&m->stage_in[slot], (size_t)c->n_ghost * 6 * sizeof(double), hipHostMallocDefault));
      m->stage_in_cap[slot] = (size_t)c->n_ghost * 6 * sizeof(double);
    }
    if (bytes) PYN_HIP(hipMemcpyAsync(m->stage_out[slot], c->d_send_buf, bytes, hipMemcpyDeviceToHost, st));
    ShmExchange* e = new ShmExchange{m, slot, bs, bytes, c->neigh, c->send_ptr, c->recv_ptr};
    PYN_HIP(hipLaunchHostFunc(st, shm_exchange_host, e));
    if (in_bytes)
      PYN_HIP(hipMemcpyAsync(x + c->n_owned * bs, m->stage_in[slot], in_bytes, hipMemcpyHostToDevice, st));
    return PYN_OK;
  }
  PYN_NCCL(ncclGroupStart());
  for (size_t k = 0; k < c->neigh.size(); ++k) {
    int64_t ns = c->send_ptr[k + 1] - c->send_ptr[k];
    int64_t nr = c->recv_ptr[k + 1] - c->recv_ptr[k];
    if (ns) PYN_NCCL(ncclSend(c->d_send_buf + c->send_ptr[k] * bs, (size_t)ns * bs, ncclDouble, c->neigh[k], c->comm_halo, st));
    if (nr) PYN_NCCL(ncclRecv(x + (c->n_owned + c->recv_ptr[k]) * bs, (size_t)nr * bs, ncclDouble, c->neigh[k], c->comm_halo, st));
  }
  PYN_NCCL(ncclGroupEnd());
  return PYN_OK;
}

// ---------------------------------------------------------------------------------------------
// mesh / tables / bc
// first connectivity entry outside [0, n_node) (INT64_MAX: none)
__global__ void conn_range_kernel(const int32_t* __restrict__ conn, int64_t n, int32_t n_node, unsigned long long* __restrict__ first_bad) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; t < n; t += stride) {
    const int32_t v = conn[t];
    if (v < 0 || v >= n_node) atomicMin(first_bad, (unsigned long long)t);
  }
}

// box mesh in closed form: element e = (ex, ey[, el]) of the local block, local node a -> local id of the lattice node
struct BoxArgs {
  int dim, nn, m;
  int64_t n_elem, n_node, PS;          // PS: nodes per plane (3-D) / per x-line (2-D) = one step of the slow axis
  int E[3], N[3];                      // elements per axis of the local block, global lattice per axis
  int64_t layer0;                      // first global element layer of the block along the slow axis
  const int32_t* loc;                  // [nn * dim] lattice offsets of the local nodes
  const int32_t* lplane;               // [m * E[slow] + 1]: LOCAL plane index of plane j of the block (owned first, then ghosts)
  const int64_t* planes;               // [n_node / PS]: global slow-axis index of local plane k
  const double* axes;                  // coordinates of the lattice lines: N[0] + N[1] (+ N[2]) doubles
};

__global__ void box_conn_kernel(BoxArgs B, int32_t* __restrict__ conn) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B.n_elem * B.nn) return;
  const int64_t e = t / B.nn;
  const int a = (int)(t - e * B.nn);
  const int ex = (int)(e % B.E[0]);
  const int64_t r = e / B.E[0];
  const int ey = B.dim == 3 ? (int)(r % B.E[1]) : 0;
  const int64_t el = B.dim == 3 ? r / B.E[1] : r;
  const int32_t* l = B.loc + a * B.dim;
  int64_t id = (int64_t)B.lplane[B.m * el + l[B.dim - 1]] * B.PS + B.m * ex + l[0];
  if (B.dim == 3) id += (int64_t)(B.m * ey + l[1]) * B.N[0];
  conn[t] = (int32_t)id;
}

__global__ void box_xyz_kernel(BoxArgs B, double* __restrict__ xyz) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= B.n_node) return;
  const int64_t k = n / B.PS, in = n - k * B.PS;
  const int64_t g = B.planes[k];
  if (B.dim == 3) {
    const int iy = (int)(in / B.N[0]), ix = (int)(in - (int64_t)iy * B.N[0]);
    xyz[n * 3] = B.axes[ix];
    xyz[n * 3 + 1] = B.axes[B.N[0] + iy];
    xyz[n * 3 + 2] = B.axes[B.N[0] + B.N[1] + g];
  } else {
    xyz[n * 2] = B.axes[in];
    xyz[n * 2 + 1] = B.axes[B.N[0] + g];
  }
}

// what pyn_mesh_set / pyn_mesh_box share once c->d_conn and c->d_xyz hold the local mesh
static int mesh_installed(pyn_ctx* c, const ConnAt& at) {
  {
    unsigned long long* d_bad = nullptr;
    unsigned long long bad = ~0ull;
    const int64_t n = c->n_elem * c->nn;
    PYN_HIP(hipMalloc((void**)&d_bad, sizeof(bad)));
    PYN_HIP(hipMemcpyAsync(d_bad, &bad, sizeof(bad), hipMemcpyHostToDevice, c->stream));
    conn_range_kernel<<<(unsigned)std::min<int64_t>((n + 255) / 256, 4096), 256, 0, c->stream>>>(c->d_conn, n, (int32_t)c->n_node, d_bad);
    PYN_HIP(hipGetLastError());
    PYN_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
    PYN_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(d_bad);
    if (bad != ~0ull) {
      const long long i = (long long)bad;
      c->n_elem = c->n_node = 0;
      PYN_CHECK(false, "conn[%lld]=%d out of range", i, (int)at(i));
    }
  }
  c->mesh_affine = -1;
  for (int k = 0; k < 3; ++k) {   // matrix-free operators belong to the mesh
    (void)hipFree(c->mf_mask[k]);
    c->mf_mask[k] = nullptr;
    c->mf_set[k] = false;
  }
  PYN_TRY(pyn_lattice_detect(c, at));
  PYN_TRY(pyn_ho3_detect(c, at));
  // graph + matrices depend on the mesh
  (void)hipFree(c->d_rowptr);
  (void)hipFree(c->d_colidx);
  c->d_rowptr = nullptr;
  c->d_colidx = nullptr;
  c->nnzb = 0;
  return PYN_OK;
}

static int mesh_sizes(pyn_ctx* c, int dim, int nn, int64_t n_elem, int64_t n_node) {
  PYN_CHECK(dim == 2 || dim == 3, "dim must be 2 or 3");
  int ngl = 0;
  for (int g = 2; g <= 32; ++g) {
    int p = 1;
    for (int d = 0; d < dim; ++d) p *= g;
    if (p == nn) ngl = g;
  }
  // nn == dim + 1: linear simplex (triangle / tetrahedron) -- geometry nodes = the element's nodes
  const bool simplex = nn == dim + 1;
  PYN_CHECK(ngl >= 2 || simplex, "nn=%d is neither ngl^dim nor a linear simplex (dim+1)", nn);
  PYN_CHECK(n_elem > 0 && n_node > 0 && n_node < (int64_t)INT32_MAX, "bad sizes");
  PYN_CHECK(n_elem * nn < ((int64_t)1 << 40), "bad sizes");
  if (!c->halo_set) {
    c->n_owned = n_node;
    c->n_ghost = 0;
  }
  PYN_CHECK(c->n_owned + c->n_ghost == n_node, "n_node %lld != owned+ghost %lld", (long long)n_node,
            (long long)(c->n_owned + c->n_ghost));
  c->dim = dim;
  c->nn = nn;
  c->nc = simplex ? nn : 1 << dim;
  c->ngl = simplex ? 0 : ngl;
  c->n_elem = n_elem;
  c->n_node = n_node;
  PYN_HIP(hipSetDevice(c->device));
  return PYN_OK;
}

extern "C" int pyn_mesh_set(pyn_ctx* c, int dim, int nn, int64_t n_elem, int64_t n_node, const int32_t* conn,
                            const double* xyz) {
  PYN_CHECK(c && conn && xyz, "NULL argument");
  PYN_TRY(mesh_sizes(c, dim, nn, n_elem, n_node));
  PYN_TRY(dev_upload(&c->d_conn, conn, (size_t)n_elem * nn, c->stream));
  PYN_TRY(dev_upload(&c->d_xyz, xyz, (size_t)n_node * dim, c->stream));
  return mesh_installed(c, [conn](int64_t i) { return conn[i]; });
}

extern "C" int pyn_mesh_box(pyn_ctx* c, int dim, int ngl, const int64_t* nel_local, int64_t layer0, const int64_t* lattice,
                            const int32_t* loc, int64_t n_planes, const int64_t* planes, const double* axes) {
  PYN_CHECK(c && nel_local && lattice && loc && planes && axes, "NULL argument");
  PYN_CHECK(dim == 2 || dim == 3, "dim must be 2 or 3");
  PYN_CHECK(ngl >= 2 && ngl <= 32, "ngl out of range");
  const int m = ngl - 1, slow = dim - 1;
  int nn = 1;
  int64_t n_elem = 1, PS = 1;
  for (int d = 0; d < dim; ++d) {
    nn *= ngl;
    PYN_CHECK(nel_local[d] >= 1 && lattice[d] >= 2 && lattice[d] < INT32_MAX && nel_local[d] < INT32_MAX, "bad box sizes");
    PYN_CHECK(d == slow || lattice[d] == m * nel_local[d] + 1, "only the slowest axis may be cut into slabs");
    n_elem *= nel_local[d];
    if (d != slow) PS *= lattice[d];
  }
  PYN_CHECK(layer0 >= 0 && m * (layer0 + nel_local[slow]) + 1 <= lattice[slow], "element layers outside the lattice");
  PYN_CHECK(n_planes == m * nel_local[slow] + 1, "a block of %lld element layers has %lld planes, not %lld", (long long)nel_local[slow],
            (long long)(m * nel_local[slow] + 1), (long long)n_planes);
  for (int a = 0; a < nn; ++a)
    for (int d = 0; d < dim; ++d) PYN_CHECK(loc[a * dim + d] >= 0 && loc[a * dim + d] <= m, "loc[%d][%d] outside the element", a, d);
  // local plane index of every plane of the block
  std::vector<int32_t> lplane((size_t)n_planes, -1);
  const int64_t g0 = (int64_t)m * layer0;
  for (int64_t k = 0; k < n_planes; ++k) {
    const int64_t j = planes[k] - g0;
    PYN_CHECK(j >= 0 && j < n_planes && lplane[j] < 0, "planes[] is not a permutation of the block's planes");
    lplane[j] = (int32_t)k;
  }
  PYN_TRY(mesh_sizes(c, dim, nn, n_elem, PS * n_planes));
  PYN_TRY(dev_upload(&c->d_conn, (const int32_t*)nullptr, (size_t)n_elem * nn, c->stream));
  PYN_TRY(dev_upload(&c->d_xyz, (const double*)nullptr, (size_t)c->n_node * dim, c->stream));
  BoxArgs B;
  B.dim = dim;
  B.nn = nn;
  B.m = m;
  B.n_elem = n_elem;
  B.n_node = c->n_node;
  B.PS = PS;
  B.layer0 = layer0;
  int64_t n_axes = 0;
  for (int d = 0; d < 3; ++d) {
    B.E[d] = d < dim ? (int)nel_local[d] : 1;
    B.N[d] = d < dim ? (int)lattice[d] : 1;
    if (d < dim) n_axes += lattice[d];
  }
  int32_t *d_loc = nullptr, *d_lplane = nullptr;
  int64_t* d_planes = nullptr;
  double* d_axes = nullptr;
  PYN_TRY(dev_upload(&d_loc, loc, (size_t)nn * dim, c->stream));
  PYN_TRY(dev_upload(&d_lplane, lplane.data(), (size_t)n_planes, c->stream));
  PYN_TRY(dev_upload(&d_planes, planes, (size_t)n_planes, c->stream));
  PYN_TRY(dev_upload(&d_axes, axes, (size_t)n_axes, c->stream));
  B.loc = d_loc;
  B.lplane = d_lplane;
  B.planes = d_planes;
  B.axes = d_axes;
  box_conn_kernel<<<(unsigned)((n_elem * nn + 255) / 256), 256, 0, c->stream>>>(B, c->d_conn);
  box_xyz_kernel<<<(unsigned)((c->n_node + 255) / 256), 256, 0, c->stream>>>(B, c->d_xyz);
  PYN_HIP(hipGetLastError());
  PYN_HIP(hipStreamSynchronize(c->stream));   // the uploads above read host memory of this frame
  (void)hipFree(d_loc);
  (void)hipFree(d_lplane);
  (void)hipFree(d_planes);
  (void)hipFree(d_axes);
  // the same closed form for the handful of entries the topology detection asks the host for
  const int E0 = B.E[0], E1 = B.E[1], N0 = B.N[0];
  std::vector<int32_t> locv(loc, loc + (size_t)nn * dim);
  return mesh_installed(c, [=](int64_t t) -> int32_t {
    const int64_t e = t / nn;
    const int a = (int)(t - e * nn);
    const int ex = (int)(e % E0);
    const int64_t r = e / E0;
    const int ey = dim == 3 ? (int)(r % E1) : 0;
    const int64_t el = dim == 3 ? r / E1 : r;
    const int32_t* l = locv.data() + (size_t)a * dim;
    int64_t id = (int64_t)lplane[m * el + l[dim - 1]] * PS + m * ex + l[0];
    if (dim == 3) id += (int64_t)(m * ey + l[1]) * N0;
    return (int32_t)id;
  });
}

extern "C" int pyn_mesh_get(pyn_ctx* c, int32_t* conn, double* xyz) {
  PYN_CHECK(c && c->n_elem > 0, "pyn_mesh_set first");
  PYN_HIP(hipSetDevice(c->device));
  if (conn) PYN_HIP(hipMemcpyAsync(conn, c->d_conn, (size_t)c->n_elem * c->nn * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (xyz) PYN_HIP(hipMemcpyAsync(xyz, c->d_xyz, (size_t)c->n_node * c->dim * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

extern "C" int pyn_mesh_topology(pyn_ctx* c, int* kind, int* nx, int* ny, int* nz) {
  PYN_CHECK(c && c->n_elem > 0, "pyn_mesh_set first");
  if (c->ho3.valid && (c->ho3.ngl == 3 || c->ho3.dim == 2)) {   // second-order lattice (kind 2) / first-order quadrilaterals (kind 3):
                                                                // nodes per x-line, x-lines per plane, planes (2-D: ny = x-lines, nz = 1)
    if (kind) *kind = c->ho3.ngl == 3 ? 2 : 3;
    if (nx) *nx = c->ho3.NX;
    if (ny) *ny = c->ho3.dim == 3 ? c->ho3.NY : c->ho3.npl;
    if (nz) *nz = c->ho3.dim == 3 ? c->ho3.npl : 1;
    return PYN_OK;
  }
  if (kind) *kind = c->lat.valid ? 1 : 0;
  if (nx) *nx = c->lat.nx;
  if (ny) *ny = c->lat.ny;
  if (nz) *nz = c->lat.npl;
  return PYN_OK;
}

extern "C" int pyn_elem_tables_set(pyn_ctx* c, int which, int ngp, const double* w, const double* H, const double* Hrs,
                                   const double* HrsCoo) {
  PYN_CHECK(c && w && H && Hrs && HrsCoo, "NULL argument");
  PYN_CHECK(which >= 0 && which < 3, "bad table slot");
  PYN_CHECK(c->nn > 0, "pyn_mesh_set first");
  PYN_CHECK(ngp > 0, "ngp must be positive");
  QuadTab& q = c->quad[which];
  q.ngp = ngp;
  PYN_TRY(dev_upload(&q.w, w, (size_t)ngp, c->stream));
  PYN_TRY(dev_upload(&q.H, H, (size_t)ngp * c->nn, c->stream));
  PYN_TRY(dev_upload(&q.Hrs, Hrs, (size_t)ngp * c->dim * c->nn, c->stream));
  PYN_TRY(dev_upload(&q.HrsCoo, HrsCoo, (size_t)ngp * c->dim * c->nc, c->stream));
  q.wsum = 0.0;
  for (int g = 0; g < ngp; ++g) q.wsum += w[g];
  PYN_TRY(pyn_ho3_tables(c, which, ngp, w, H, Hrs));   // ngl = 3: reference matrices of the closed-form blocks
  q.const_grad = c->nc == c->nn;
  for (int g = 0; g < ngp && q.const_grad; ++g)
    for (int t = 0; t < c->dim * c->nn; ++t)
      if (Hrs[(size_t)g * c->dim * c->nn + t] != Hrs[t] || HrsCoo[(size_t)g * c->dim * c->nn + t] != Hrs[t]) {
        q.const_grad = false;
        break;
      }
  if (which == PYN_Q_FULL && c->dim == 3 && c->nn == 8 && ngp == 8) {
    // Tables of the affine shortcut of the tiled Q1-hex kernel (exact for parallelepipeds, where
    // J is constant):  L_ab = detJ * sum_{r<=s} Q_rs T_rs[ab],  Q = J^-T J^-1 (reference axes),
    // T_rr = sum_g w hr[r][a] hr[r][b],  T_rs = sum_g w (hr[r][a] hr[s][b] + hr[s][a] hr[r][b]).
    double aff[6 * 36 + 4 * 8 + 9];
    const int RS[6][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 1}, {0, 2}, {1, 2}};
    for (int t = 0; t < 6; ++t) {
      const int r = RS[t][0], s2 = RS[t][1];
      int idx = 0;
      for (int a = 0; a < 8; ++a)
        for (int b = a; b < 8; ++b) {
          double acc = 0.0;
          for (int g = 0; g < 8; ++g) {
            const double* hr = Hrs + (size_t)g * 24;
            acc += w[g] * (r == s2 ? hr[r * 8 + a] * hr[r * 8 + b] : hr[r * 8 + a] * hr[s2 * 8 + b] + hr[s2 * 8 + a] * hr[r * 8 + b]);
          }
          aff[t * 36 + idx++] = acc;
        }
    }
    // signs of the non-affine trilinear monomials rs, rt, st, rst at the corners (reference
    // coordinates of corner a = sign of dN_a/d(axis) anywhere inside the element)
    for (int a = 0; a < 8; ++a) {
      const double xr = HrsCoo[0 * 8 + a] > 0 ? 1.0 : -1.0, xs = HrsCoo[1 * 8 + a] > 0 ? 1.0 : -1.0,
                   xt = HrsCoo[2 * 8 + a] > 0 ? 1.0 : -1.0;
      aff[216 + 0 * 8 + a] = xr * xs;
      aff[216 + 1 * 8 + a] = xr * xt;
      aff[216 + 2 * 8 + a] = xs * xt;
      aff[216 + 3 * 8 + a] = xr * xs * xt;
    }
    // S[d][m] = sum_c hcoo[d][c] C_m[c]: d(reference axis d) of the lattice coordinate m (corner offsets in the
    // closure order of SURVEY.md A.2) -- for a parallelepiped J = S . (edge vectors along x, y, z)
    const int CO[3][8] = {{0, 0, 1, 1, 0, 1, 1, 0}, {0, 1, 1, 0, 0, 0, 1, 1}, {0, 0, 0, 0, 1, 1, 1, 1}};
    for (int d = 0; d < 3; ++d)
      for (int m = 0; m < 3; ++m) {
        double sacc = 0.0;
        for (int cc = 0; cc < 8; ++cc) sacc += HrsCoo[d * 8 + cc] * CO[m][cc];
        aff[248 + d * 3 + m] = sacc;
      }
    c->aff_standard = pyn_q1_affine_tables_standard(aff);
    c->aff_rw_standard = pyn_q1_mixed_tables_standard(w, H, Hrs);
    c->q1_gauss_standard = pyn_q1_gauss_tables_standard(w, H, Hrs, HrsCoo);
    PYN_TRY(dev_upload(&c->d_aff, aff, (size_t)(6 * 36 + 32 + 9), c->stream));
  }
  if (which == PYN_Q_RED && c->dim == 3 && c->nn == 8) {
    // one point at the centroid, weight 8, gradients s_d(a) / 8, values 1 / 8 (corner order of SURVEY.md A.2)?
    const int CO[3][8] = {{0, 0, 1, 1, 0, 1, 1, 0}, {0, 1, 1, 0, 0, 0, 1, 1}, {0, 0, 0, 0, 1, 1, 1, 1}};
    bool ok = ngp == 1 && fabs(w[0] - 8.0) < 1e-13;
    for (int a = 0; a < 8 && ok; ++a) {
      ok = fabs(H[a] - 0.125) < 1e-14;
      for (int d = 0; d < 3 && ok; ++d)
        ok = fabs(Hrs[d * 8 + a] - (2 * CO[d][a] - 1) * 0.125) < 1e-14 && fabs(HrsCoo[d * 8 + a] - (2 * CO[d][a] - 1) * 0.125) < 1e-14;
    }
    c->q1_red_standard = ok;
  }
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

extern "C" int pyn_bc_set(pyn_ctx* c, int ndof, const uint8_t* mask) {
  PYN_CHECK(c, "ctx is NULL");
  c->bc_stamp++;
  PYN_CHECK(c->n_node > 0, "pyn_mesh_set first");
  if (!mask) {
    (void)hipFree(c->d_bcmask);
    c->d_bcmask = nullptr;
    c->bc_ndof = 0;
    return PYN_OK;
  }
  PYN_CHECK(ndof >= 1 && ndof <= 3, "ndof must be 1..3");
  PYN_TRY(dev_upload(&c->d_bcmask, mask, (size_t)c->n_node * ndof, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  c->bc_ndof = ndof;
  return PYN_OK;
}

// ---------------------------------------------------------------------------------------------
// matrices / vectors
int pyn_check_mat(pyn_ctx* c, int id, const char* what) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_CHECK(id >= 0 && id < (int)c->mats.size() && c->mats[id].live, "%s: invalid matrix handle %d", what, id);
  return PYN_OK;
}
int pyn_check_vec(pyn_ctx* c, int id, const char* what) {
  PYN_CHECK(c, "ctx is NULL");
  PYN_CHECK(id >= 0 && id < (int)c->vecs.size() && c->vecs[id].live, "%s: invalid vector handle %d", what, id);
  return PYN_OK;
}

extern "C" int pyn_mat_create(pyn_ctx* c, int br, int bc, int* mat_id) {
  PYN_CHECK(c && mat_id, "NULL argument");
  PYN_CHECK(c->d_rowptr, "pyn_csr_symbolic first");
  PYN_CHECK(br >= 1 && br <= 6 && bc >= 1 && bc <= 6, "block shape out of range");
  DMat m;
  m.br = br;
  m.bc = bc;
  size_t n = (size_t)c->nnzb * br * bc;
  PYN_HIP(hipMalloc((void**)&m.val, n * sizeof(double)));
  PYN_HIP(hipMemsetAsync(m.val, 0, n * sizeof(double), c->stream));
  if (getenv("PYNAMA_DEBUG_ALLOC")) fprintf(stderr, "[pynama] matrix %d: %zu bytes at %p\n", (int)c->mats.size(), n * sizeof(double), (void*)m.val);
  m.rhs_clean = PYN_RHS_ANY;
  m.live = true;
  c->mats.push_back(m);
  *mat_id = (int)c->mats.size() - 1;
  return PYN_OK;
}

extern "C" int pyn_mat_destroy(pyn_ctx* c, int id) {
  PYN_TRY(pyn_check_mat(c, id, "pyn_mat_destroy"));
  DMat& m = c->mats[id];
  PYN_HIP(hipStreamSynchronize(c->stream));   // no kernel in flight may still read the arrays
  (void)hipFree(m.val);
  (void)hipFree(m.sell_val);
  (void)hipFree(m.dinv);
  m.release_lu();
  pyn_rhs_release(m);
  m = DMat();   // live = false: the handle is dead, its slot is not reused (handles stay stable)
  return PYN_OK;
}

// Host insertion path: a dense block of values into the device matrix (PETSc MatSetValues with ADD_VALUES / INSERT_VALUES).
// One thread per (row, column) entry: the column's node is located in the row's sorted column list by bisection; entries
// outside the graph raise the flag (PETSc: "new nonzero caused a malloc"); rows this rank does not own are dropped (the
// device path is owner-computes: there is no off-process stash).
__global__ void mat_add_values_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, double* __restrict__ val,
                                      int64_t n_owned, int br, int bc, int nr, const int32_t* __restrict__ rows, int nc,
                                      const int32_t* __restrict__ cols, const double* __restrict__ v, int insert, int* __restrict__ bad,
                                      const int32_t* __restrict__ crow) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nr * nc; e += gridDim.x * blockDim.x) {
    const int r = rows[e / nc], cidx = cols[e % nc];
    if (r < 0 || cidx < 0) continue;                 // PETSc: negative indices are ignored
    const int64_t i = r / br;
    const int p = r - (int)i * br, j = cidx / bc, q = cidx - j * bc;
    if (i >= n_owned) continue;
    const int lo = rowptr[i], len = rowptr[i + 1] - lo;
    int l = 0, h = len;
    while (l < h) {
      const int m = (l + h) >> 1;
      if (colidx[lo + m] < j) l = m + 1;
      else h = m;
    }
    if (l >= len || colidx[lo + l] != j) {
      atomicExch(bad, 1);
      continue;
    }
    const int vlo = crow ? crow[i] : lo;      // compact imposed-column matrix: the row may not be stored
    if (vlo < 0) {
      if (v[e] != 0.0) atomicExch(bad, 2);
      continue;
    }
    double* dst = val + ((int64_t)vlo * br + (int64_t)p * len + l) * bc + q;
    if (insert) *dst = v[e];
    else atomicAdd(dst, v[e]);
  }
}

extern "C" int pyn_mat_add_values(pyn_ctx* c, int id, int nr, const int32_t* rows, int nc, const int32_t* cols, const double* vals,
                                  int insert) {
  PYN_TRY(pyn_check_mat(c, id, "pyn_mat_add_values"));
  PYN_CHECK(rows && cols && vals && nr > 0 && nc > 0 && (int64_t)nr * nc <= (1 << 24), "bad block");
  DMat& m = c->mats[id];
  PYN_HIP(hipSetDevice(c->device));
  const size_t bytes = (size_t)(nr + nc) * sizeof(int32_t) + (size_t)nr * nc * sizeof(double) + 16;
  DevTmp buf;
  PYN_HIP(buf.alloc(bytes));
  double* dv = buf.as<double>();
  int32_t* dr = reinterpret_cast<int32_t*>(dv + (size_t)nr * nc);
  int32_t* dc = dr + nr;
  int* dbad = reinterpret_cast<int*>(dc + nc);
  PYN_HIP(hipMemcpyAsync(dv, vals, (size_t)nr * nc * sizeof(double), hipMemcpyHostToDevice, c->stream));
  PYN_HIP(hipMemcpyAsync(dr, rows, nr * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  PYN_HIP(hipMemcpyAsync(dc, cols, nc * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  PYN_HIP(hipMemsetAsync(dbad, 0, sizeof(int), c->stream));
  PYN_TRY(pyn_rhs_ensure(c, m));
  m.touch();
  mat_add_values_kernel<<<std::min((nr * nc + 255) / 256, 1024), 256, 0, c->stream>>>(c->d_rowptr, c->d_colidx, m.val, c->n_owned, m.br, m.bc, nr, dr,
                                                                                nc, dc, dv, insert, dbad, m.rhs_compact ? m.c_crow : nullptr);
  int bad = 0;
  PYN_HIP(hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  PYN_CHECK(bad != 2, "pyn_mat_add_values: a nonzero entry lies in a row that the compact imposed-column matrix does not store (no "
                      "imposed node next to it under the Dirichlet set the matrix was laid out for)");
  PYN_CHECK(!bad, "pyn_mat_add_values: an entry lies outside the node graph of the mesh (no new nonzeros can be allocated)");
  return PYN_OK;
}

extern "C" int pyn_mat_zero(pyn_ctx* c, int id) {
  PYN_TRY(pyn_check_mat(c, id, "pyn_mat_zero"));
  DMat& m = c->mats[id];
  PYN_TRY(pyn_rhs_ensure(c, m));
  PYN_HIP(hipMemsetAsync(m.val, 0, (size_t)pyn_mat_blocks(c, m) * m.br * m.bc * sizeof(double), c->stream));
  m.touch();
  m.rhs_clean = PYN_RHS_ANY;
  return PYN_OK;
}

extern "C" int pyn_mat_get_values(pyn_ctx* c, int id, double* val) {
  PYN_TRY(pyn_check_mat(c, id, "pyn_mat_get_values"));
  PYN_CHECK(val, "val is NULL");
  DMat& m = c->mats[id];
  const size_t bytes = (size_t)c->nnzb * m.br * m.bc * sizeof(double);
  if (m.rhs_compact) {      // the caller sees the graph's layout: zeros in the rows the matrix does not store
    DevTmp full;
    PYN_HIP(full.alloc(bytes));
    PYN_TRY(pyn_rhs_expand(c, m, full.as<double>()));
    PYN_HIP(hipMemcpyAsync(val, full.p, bytes, hipMemcpyDeviceToHost, c->stream));
    PYN_HIP(hipStreamSynchronize(c->stream));
    return PYN_OK;
  }
  PYN_HIP(hipMemcpyAsync(val, m.val, bytes, hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

extern "C" int pyn_vec_create(pyn_ctx* c, int bs, int* vec_id) {
  PYN_CHECK(c && vec_id, "NULL argument");
  PYN_CHECK(c->n_node > 0, "pyn_mesh_set first");
  PYN_CHECK(bs >= 1 && bs <= 6, "block size out of range");
  DVec v;
  v.bs = bs;
  size_t n = (size_t)n_local(c) * bs;
  PYN_HIP(hipMalloc((void**)&v.d, n * sizeof(double)));
  PYN_HIP(hipMemsetAsync(v.d, 0, n * sizeof(double), c->stream));
  v.live = true;
  for (size_t i = 0; i < c->vecs.size(); ++i)
    if (!c->vecs[i].live) {
      c->vecs[i] = v;
      *vec_id = (int)i;
      return PYN_OK;
    }
  c->vecs.push_back(v);
  *vec_id = (int)c->vecs.size() - 1;
  return PYN_OK;
}

extern "C" int pyn_vec_destroy(pyn_ctx* c, int id) {
  PYN_TRY(pyn_check_vec(c, id, "pyn_vec_destroy"));
  PYN_HIP(hipStreamSynchronize(c->stream));
  PYN_HIP(hipFree(c->vecs[id].d));
  c->vecs[id] = DVec();
  return PYN_OK;
}

extern "C" int pyn_vec_set_host(pyn_ctx* c, int id, const double* src) {
  PYN_TRY(pyn_check_vec(c, id, "pyn_vec_set_host"));
  PYN_CHECK(src, "src is NULL");
  DVec& v = c->vecs[id];
  PYN_HIP(hipMemcpyAsync(v.d, src, (size_t)c->n_owned * v.bs * sizeof(double), hipMemcpyHostToDevice, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

extern "C" int pyn_vec_set_local_host(pyn_ctx* c, int id, const double* src) {
  PYN_TRY(pyn_check_vec(c, id, "pyn_vec_set_local_host"));
  PYN_CHECK(src, "src is NULL");
  DVec& v = c->vecs[id];
  PYN_HIP(hipMemcpyAsync(v.d, src, (size_t)n_local(c) * v.bs * sizeof(double), hipMemcpyHostToDevice, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

extern "C" int pyn_vec_get_host(pyn_ctx* c, int id, double* dst) {
  PYN_TRY(pyn_check_vec(c, id, "pyn_vec_get_host"));
  PYN_CHECK(dst, "dst is NULL");
  DVec& v = c->vecs[id];
  PYN_HIP(hipMemcpyAsync(dst, v.d, (size_t)c->n_owned * v.bs * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

static inline int ew_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 511) / 512, PYN_MAX_PARTIALS)); }

__global__ void fill_kernel(double* __restrict__ x, int64_t n, double v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = v;
}

extern "C" int pyn_vec_fill(pyn_ctx* c, int id, double value) {
  PYN_TRY(pyn_check_vec(c, id, "pyn_vec_fill"));
  DVec& v = c->vecs[id];
  int64_t n = n_local(c) * v.bs;
  fill_kernel<<<ew_grid(n), 256, 0, c->stream>>>(v.d, n, value);
  return PYN_OK;
}

__global__ void scatter_kernel(double* __restrict__ x, const int32_t* __restrict__ idx, const double* __restrict__ vals,
                               int64_t n, int add) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (add)
      atomicAdd(&x[idx[i]], vals[i]);
    else
      x[idx[i]] = vals[i];
  }
}

extern "C" int pyn_vec_scatter_host(pyn_ctx* c, int id, int64_t n, const int32_t* idx, const double* vals, int add) {
  PYN_TRY(pyn_check_vec(c, id, "pyn_vec_scatter_host"));
  if (n == 0) return PYN_OK;
  PYN_CHECK(idx && vals, "NULL argument");
  DVec& v = c->vecs[id];
  int64_t lim = c->n_owned * v.bs;
  for (int64_t i = 0; i < n; ++i) PYN_CHECK(idx[i] >= 0 && idx[i] < lim, "index %d out of range [0,%lld)", idx[i], (long long)lim);
  size_t bytes = (size_t)n * (sizeof(int32_t) + sizeof(double)) + 16;
  PYN_TRY(pyn_ensure_work(c, bytes));
  double* dv = c->d_work;
  int32_t* di = (int32_t*)(dv + n);
  PYN_HIP(hipMemcpyAsync(dv, vals, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  PYN_HIP(hipMemcpyAsync(di, idx, n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  scatter_kernel<<<ew_grid(n), 256, 0, c->stream>>>(v.d, di, dv, n, add);
  PYN_HIP(hipStreamSynchronize(c->stream));
  return PYN_OK;
}

__global__ void axpby_kernel(double* __restrict__ w, double a, const double* __restrict__ x, double b,
                             const double* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double r = 0.0;
    if (a != 0.0) r += a * x[i];
    if (b != 0.0) r += b * y[i];
    w[i] = r;
  }
}

extern "C" int pyn_vec_axpby(pyn_ctx* c, int w, double a, int x, double b, int y) {
  PYN_TRY(pyn_check_vec(c, w, "axpby w"));
  PYN_TRY(pyn_check_vec(c, x, "axpby x"));
  PYN_TRY(pyn_check_vec(c, y, "axpby y"));
  PYN_CHECK(c->vecs[w].bs == c->vecs[x].bs && c->vecs[w].bs == c->vecs[y].bs, "block size mismatch");
  int64_t n = c->n_owned * c->vecs[w].bs;
  axpby_kernel<<<ew_grid(n), 256, 0, c->stream>>>(c->vecs[w].d, a, c->vecs[x].d, b, c->vecs[y].d, n);
  return PYN_OK;
}

__global__ void pmult_kernel(double* __restrict__ w, const double* __restrict__ x, const double* __restrict__ y, int64_t n,
                             int recip) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    w[i] = recip ? 1.0 / x[i] : x[i] * y[i];
}

extern "C" int pyn_vec_pointwise_mult(pyn_ctx* c, int w, int x, int y) {
  PYN_TRY(pyn_check_vec(c, w, "pmult w"));
  PYN_TRY(pyn_check_vec(c, x, "pmult x"));
  PYN_TRY(pyn_check_vec(c, y, "pmult y"));
  PYN_CHECK(c->vecs[w].bs == c->vecs[x].bs && c->vecs[w].bs == c->vecs[y].bs, "block size mismatch");
  int64_t n = c->n_owned * c->vecs[w].bs;
  pmult_kernel<<<ew_grid(n), 256, 0, c->stream>>>(c->vecs[w].d, c->vecs[x].d, c->vecs[y].d, n, 0);
  return PYN_OK;
}

extern "C" int pyn_vec_reciprocal(pyn_ctx* c, int x) {
  PYN_TRY(pyn_check_vec(c, x, "reciprocal"));
  int64_t n = c->n_owned * c->vecs[x].bs;
  pmult_kernel<<<ew_grid(n), 256, 0, c->stream>>>(c->vecs[x].d, c->vecs[x].d, c->vecs[x].d, n, 1);
  return PYN_OK;
}

// v (x) v in the reference's component order (base_problem.py:234-252):
// 2D [vx vx, vx vy, vy vy] ; 3D [vx vx, vx vy, vy vy, vy vz, vz vz, vz vx]
__global__ void vtensv_kernel(const double* __restrict__ v, double* __restrict__ out, int64_t n_nodes, int dim) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += (int64_t)gridDim.x * blockDim.x) {
    const double vx = v[i * dim], vy = v[i * dim + 1];
    if (dim == 2) {
      out[i * 3] = vx * vx;
      out[i * 3 + 1] = vx * vy;
      out[i * 3 + 2] = vy * vy;
    } else {
      const double vz = v[i * 3 + 2];
      out[i * 6] = vx * vx;
      out[i * 6 + 1] = vx * vy;
      out[i * 6 + 2] = vy * vy;
      out[i * 6 + 3] = vy * vz;
      out[i * 6 + 4] = vz * vz;
      out[i * 6 + 5] = vz * vx;
    }
  }
}

extern "C" int pyn_vec_vtensv(pyn_ctx* c, int v, int out) {
  PYN_TRY(pyn_check_vec(c, v, "vtensv v"));
  PYN_TRY(pyn_check_vec(c, out, "vtensv out"));
  const int dim = c->vecs[v].bs;
  PYN_CHECK((dim == 2 && c->vecs[out].bs == 3) || (dim == 3 && c->vecs[out].bs == 6), "block sizes must be (2,3) or (3,6)");
  vtensv_kernel<<<ew_grid(c->n_owned), 256, 0, c->stream>>>(c->vecs[v].d, c->vecs[out].d, c->n_owned, dim);
  return PYN_OK;
}

// ---- reductions ------------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// mode 0: sum x*y ; 1: sum |x| ; 2: max |x|
__global__ void __launch_bounds__(256) reduce_kernel(const double* __restrict__ x, const double* __restrict__ y, int64_t n,
                                                     int mode, double* __restrict__ part) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double a = x[i];
    if (mode == 0)
      acc += a * y[i];
    else if (mode == 1)
      acc += fabs(a);
    else
      acc = fmax(acc, fabs(a));
  }
  acc = mode == 2 ? wave_max(acc) : wave_sum(acc);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sm[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = sm[0];
    for (int k = 1; k < 4; ++k) r = mode == 2 ? fmax(r, sm[k]) : r + sm[k];
    part[blockIdx.x] = r;
  }
}

// one block: out[s] = reduce(part[s*PYN_MAX_PARTIALS + 0..nblocks))
__global__ void __launch_bounds__(256) finish_kernel(const double* __restrict__ part, int nslots, int nblocks, int op,
                                                     double* __restrict__ out) {
  __shared__ double sm[4];
  for (int s = 0; s < nslots; ++s) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
      double v = part[s * PYN_MAX_PARTIALS + i];
      acc = op == 1 ? fmax(acc, v) : acc + v;
    }
    acc = op == 1 ? wave_max(acc) : wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double r = sm[0];
      for (int k = 1; k < 4; ++k) r = op == 1 ? fmax(r, sm[k]) : r + sm[k];
      out[s] = r;
    }
    __syncthreads();
  }
}

int pyn_reduce_host(pyn_ctx* c, int nslots, int nblocks, int op, double* out) {
  finish_kernel<<<1, 256, 0, c->stream>>>(c->d_part, nslots, nblocks, op, c->d_scal + 40);
  PYN_TRY(pyn_allreduce_dev(c, c->d_scal + 40, nslots, op, c->stream));
  PYN_HIP(hipMemcpyAsync(c->h_scal, c->d_scal + 40, nslots * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PYN_HIP(hipStreamSynchronize(c->stream));
  for (int s = 0; s < nslots; ++s) out[s] = c->h_scal[s];
  return PYN_OK;
}

extern "C" int pyn_vec_dot(pyn_ctx* c, int x, int y, double* out) {
  PYN_TRY(pyn_check_vec(c, x, "dot x"));
  PYN_TRY(pyn_check_vec(c, y, "dot y"));
  PYN_CHECK(out && c->vecs[x].bs == c->vecs[y].bs, "bad arguments");
  int64_t n = c->n_owned * c->vecs[x].bs;
  int g = ew_grid(n);
  reduce_kernel<<<g, 256, 0, c->stream>>>(c->vecs[x].d, c->vecs[y].d, n, 0, c->d_part);
  return pyn_reduce_host(c, 1, g, 0, out);
}

extern "C" int pyn_vec_norm(pyn_ctx* c, int x, int type, double* out) {
  PYN_TRY(pyn_check_vec(c, x, "norm"));
  PYN_CHECK(out && (type == 1 || type == 2 || type == 3), "norm type must be 1, 2 or 3 (inf)");
  int64_t n = c->n_owned * c->vecs[x].bs;
  int g = ew_grid(n);
  int mode = type == 2 ? 0 : (type == 1 ? 1 : 2);
  reduce_kernel<<<g, 256, 0, c->stream>>>(c->vecs[x].d, c->vecs[x].d, n, mode, c->d_part);
  PYN_TRY(pyn_reduce_host(c, 1, g, type == 3 ? 1 : 0, out));
  if (type == 2) *out = sqrt(*out);
  return PYN_OK;
}

extern "C" int pyn_timers_get(pyn_ctx* c, double* ms, int n) {
  PYN_CHECK(c && ms && n > 0, "bad arguments");
  for (int i = 0; i < n && i < PYN_T_COUNT; ++i) ms[i] = c->timers[i];
  return PYN_OK;
}
