// comm.cpp -- the one exchange step of the sharded path: sum of the raw accumulators over RCCL (xGMI).
//
// New functionality: the reference is single-GPU.  Views are independent and every aggregator is a sum in some
// domain (python/semantic_meshes/src/Fusion.cu:46-92), so each rank fuses its own views and ONE ncclAllReduce(float32, sum) of the
// raw float32[P * row_stride] buffer, in place in HBM, precedes get().  The collective is enqueued on the library's own stream
// right behind the last fusion kernel: no host synchronisation between the two, no PyTorch.
//
// RCCL is loaded at run time (dlopen): single-GPU users of libsmesh_hip.so do not need it, and a process that already holds a
// librccl (e.g. the one bundled with PyTorch) keeps using that one.
#include "common.hpp"

#include <dlfcn.h>
// RCCL is loaded with dlopen at run time; its header only supplies a handful of types.  A ROCm installation without the RCCL
// development files still builds the (single-GPU) library: the same few declarations are written out below.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef enum { ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
}
#endif

#include <cstring>
#include <new>
#include <vector>

using namespace smesh;

// from fusion.hip
struct smesh_aggregator;
DeviceCtx* smesh_aggregator_ctx(smesh_aggregator* a);
std::mutex& smesh_aggregator_mutex(smesh_aggregator* a);
int smesh_aggregator_acc(smesh_aggregator* a, float** acc, uint64_t* num_floats, uint32_t* row_stride, uint64_t* rows);
uint64_t smesh_aggregator_primitives(smesh_aggregator* a);
int smesh_aggregator_join_exchange(smesh_aggregator* a);
int smesh_aggregator_refuse_scattered(smesh_aggregator* a, const char* what);
void smesh_aggregator_mark_scattered(smesh_aggregator* a, uint64_t lo, uint64_t hi);
int smesh_aggregator_exchange_begin(smesh_aggregator* a, uint64_t lo, uint64_t hi, hipStream_t st, void** buf, uint64_t* count, int* is_f64);
int smesh_aggregator_exchange_end(smesh_aggregator* a, uint64_t lo, uint64_t hi, hipStream_t st);
int smesh_aggregator_exchange_events(smesh_aggregator* a, hipEvent_t* ev_part, hipEvent_t* ev_xchg);
void smesh_aggregator_exchange_pending(smesh_aggregator* a);

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl* r = [] {
    auto* x = new Rccl();
    const char* names[] = {getenv("SMESH_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      x->handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (x->handle) break;
    }
    if (!x->handle) { x->error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found"); return x; }
    auto sym = [&](const char* s) -> void* {
      void* p = dlsym(x->handle, s);
      if (!p && x->error.empty()) x->error = std::string("librccl lacks ") + s;
      return p;
    };
    x->GetUniqueId = reinterpret_cast<decltype(x->GetUniqueId)>(sym("ncclGetUniqueId"));
    x->CommInitRank = reinterpret_cast<decltype(x->CommInitRank)>(sym("ncclCommInitRank"));
    x->CommInitAll = reinterpret_cast<decltype(x->CommInitAll)>(sym("ncclCommInitAll"));
    x->CommDestroy = reinterpret_cast<decltype(x->CommDestroy)>(sym("ncclCommDestroy"));
    x->AllReduce = reinterpret_cast<decltype(x->AllReduce)>(sym("ncclAllReduce"));
    x->ReduceScatter = reinterpret_cast<decltype(x->ReduceScatter)>(sym("ncclReduceScatter"));
    x->GroupStart = reinterpret_cast<decltype(x->GroupStart)>(sym("ncclGroupStart"));
    x->GroupEnd = reinterpret_cast<decltype(x->GroupEnd)>(sym("ncclGroupEnd"));
    x->GetErrorString = reinterpret_cast<decltype(x->GetErrorString)>(sym("ncclGetErrorString"));
    return x;
  }();
  return r;
}

int need_rccl(Rccl** out) {
  Rccl* r = rccl();
  if (!r->error.empty()) return fail(SMESH_ERR_RUNTIME, r->error);
  *out = r;
  return SMESH_OK;
}

int fail_rccl(Rccl* r, ncclResult_t e, const char* what) {
  return fail(SMESH_ERR_RUNTIME, std::string("RCCL error in ") + what + ": " + (r->GetErrorString ? r->GetErrorString(e) : "?"));
}

#define SMESH_RCCL(r, expr)                                   \
  do {                                                        \
    ncclResult_t _e = (expr);                                 \
    if (_e != ncclSuccess) return fail_rccl(r, _e, #expr);    \
  } while (0)

}  // namespace

struct smesh_comm {
  DeviceCtx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
  double* d_small = nullptr;   // device scratch of the small host-value reductions
  static constexpr int kSmall = 64;
};

extern "C" {

int smesh_comm_unique_id(uint8_t id[SMESH_COMM_ID_BYTES]) {
  if (!id) return fail(SMESH_ERR_INVALID, "id is NULL");
  static_assert(SMESH_COMM_ID_BYTES == sizeof(ncclUniqueId), "unique id size");
  Rccl* r;
  SMESH_TRY(need_rccl(&r));
  ncclUniqueId u;
  SMESH_RCCL(r, r->GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return SMESH_OK;
}

static int finish_comm(DeviceCtx* ctx, ncclComm_t c, int nranks, int rank, smesh_comm_t** out) {
  auto* m = new (std::nothrow) smesh_comm();
  if (!m) return fail(SMESH_ERR_RUNTIME, "out of memory");
  m->ctx = ctx; m->comm = c; m->nranks = nranks; m->rank = rank;
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_HIP(dev_malloc(reinterpret_cast<void**>(&m->d_small), smesh_comm::kSmall * sizeof(double)));
  *out = m;
  return SMESH_OK;
}

int smesh_comm_create(int device, int nranks, int rank, const uint8_t id[SMESH_COMM_ID_BYTES], smesh_comm_t** out) {
  if (!out || !id) return fail(SMESH_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(SMESH_ERR_INVALID, "bad rank / nranks");
  Rccl* r;
  SMESH_TRY(need_rccl(&r));
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  SMESH_HIP(hipSetDevice(device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t c = nullptr;
  SMESH_RCCL(r, r->CommInitRank(&c, nranks, u, rank));
  return finish_comm(ctx, c, nranks, rank, out);
}

int smesh_comm_create_all(const int* devices, int ndev, smesh_comm_t** out) {
  if (!out || !devices || ndev < 1) return fail(SMESH_ERR_INVALID, "bad argument");
  for (int i = 0; i < ndev; i++) out[i] = nullptr;
  Rccl* r;
  SMESH_TRY(need_rccl(&r));
  std::vector<DeviceCtx*> ctx((size_t)ndev);
  for (int i = 0; i < ndev; i++) SMESH_TRY(get_ctx(devices[i], &ctx[(size_t)i]));
  std::vector<ncclComm_t> c((size_t)ndev, nullptr);
  SMESH_RCCL(r, r->CommInitAll(c.data(), ndev, devices));
  for (int i = 0; i < ndev; i++) SMESH_TRY(finish_comm(ctx[(size_t)i], c[(size_t)i], ndev, i, &out[i]));
  return SMESH_OK;
}

int smesh_comm_destroy(smesh_comm_t* c) {
  if (!c) return SMESH_OK;
  Rccl* r = rccl();
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->comm && r->CommDestroy) (void)r->CommDestroy(c->comm);
  if (c->d_small) (void)dev_free(c->d_small);
  delete c;
  return SMESH_OK;
}

int smesh_comm_rank(const smesh_comm_t* c, int* rank, int* nranks) {
  if (!c) return fail(SMESH_ERR_INVALID, "NULL communicator");
  if (rank) *rank = c->rank;
  if (nranks) *nranks = c->nranks;
  return SMESH_OK;
}

// Sum the raw accumulators of all ranks in place.  One process per GPU: n == 1.  One process driving several GPUs
// (smesh_comm_create_all): all of its (communicator, aggregator) pairs in one call -- the collectives are grouped.
// Asynchronous: enqueued on each device's library stream behind the fusion kernels already queued there; get() /
// smesh_synchronize() order after it.
int smesh_allreduce(smesh_comm_t* const* comms, smesh_aggregator_t* const* aggs, int n) {
  if (n < 1 || !comms || !aggs) return fail(SMESH_ERR_INVALID, "bad argument");
  Rccl* r;
  SMESH_TRY(need_rccl(&r));
  for (int i = 0; i < n; i++) {
    if (!comms[i] || !aggs[i]) return fail(SMESH_ERR_INVALID, "NULL communicator / aggregator");
    if (smesh_aggregator_ctx(aggs[i]) != comms[i]->ctx) return fail(SMESH_ERR_INVALID, "aggregator and communicator live on different devices");
  }
  if (n > 1) SMESH_RCCL(r, r->GroupStart());
  int status = SMESH_OK;
  std::vector<char> reduced((size_t)n, 0);   // pair i's exchange was begun (Mul: its rows now sit in the float64 image): its epilogue is due whatever happens next
  for (int i = 0; i < n && status == SMESH_OK; i++) {
    std::lock_guard<std::mutex> g(smesh_aggregator_mutex(aggs[i]));
    DeviceCtx* ctx = comms[i]->ctx;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { status = fail(SMESH_ERR_RUNTIME, "hipSetDevice failed"); break; }
    uint64_t count = 0;
    void* buf = nullptr;
    int f64 = 0;
    // a failure here must surface: a rank that skipped the collective would leave its peers blocked inside theirs.  An EMPTY
    // accumulator (P == 0) is empty on every rank (same num_primitives is a precondition), so skipping it is collective-safe.
    // Mul: the (hi, lo) pairs travel as float64 (smesh_aggregator_exchange_begin), twice the bytes, exact to 1e-16.
    status = smesh_aggregator_refuse_scattered(aggs[i], "allreduce");
    if (status == SMESH_OK) status = smesh_aggregator_join_exchange(aggs[i]);
    const uint64_t P = smesh_aggregator_primitives(aggs[i]);
    if (status == SMESH_OK) status = smesh_aggregator_exchange_begin(aggs[i], 0, P, ctx->stream, &buf, &count, &f64);
    if (status != SMESH_OK || count == 0) continue;
    reduced[(size_t)i] = 1;
    ProfScope prof(ctx, SMESH_PROF_EXCHANGE);
    const ncclResult_t e = r->AllReduce(buf, buf, (size_t)count, f64 ? ncclFloat64 : ncclFloat32, ncclSum, comms[i]->comm, ctx->stream);
    if (e != ncclSuccess) status = fail_rccl(r, e, "ncclAllReduce");
  }
  if (n > 1) {
    const ncclResult_t e = r->GroupEnd();
    if (e != ncclSuccess && status == SMESH_OK) status = fail_rccl(r, e, "ncclGroupEnd");
  }
  // The epilogues go on the streams only now: inside a group a collective is put on its stream by ncclGroupEnd, so an epilogue
  // launched beside its ncclAllReduce call would run AHEAD of the reduction and write this device's own partial sums back
  // (a grouped Mul all-reduce returned unreduced data without an error).  They run for EVERY pair whose exchange was
  // begun, also when another pair or ncclGroupEnd failed: a Mul aggregator left in its float64 image would be unusable;
  // the first error is the one reported.
  for (int i = 0; i < n; i++) {
    if (!reduced[(size_t)i]) continue;
    std::lock_guard<std::mutex> g(smesh_aggregator_mutex(aggs[i]));
    DeviceCtx* ctx = comms[i]->ctx;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (hipSetDevice(ctx->device) != hipSuccess) { if (status == SMESH_OK) status = fail(SMESH_ERR_RUNTIME, "hipSetDevice failed"); continue; }
    const std::string first = smesh_last_error();
    const int e = smesh_aggregator_exchange_end(aggs[i], 0, smesh_aggregator_primitives(aggs[i]), ctx->stream);
    if (status == SMESH_OK) status = e;
    else (void)fail(status, first);          // (keep the first failure's message)
  }
  return status;
}

// Opt-in alternative to smesh_allreduce (half the bytes on every xGMI link): an in-place ncclReduceScatter.  Afterwards rank r
// holds the sum over all ranks in rows [*row_lo, *row_hi) of ITS accumulator -- q = floor(P / nranks / 4) * 4 rows per rank, so that
// every slice starts on a 16-byte boundary whatever the class count -- and the few rows beyond nranks * q (fewer than 5 * nranks)
// are all-reduced and belong to the last rank's range.  The other rows of a rank keep its own partial sums: only
// smesh_aggregator_get_rows(a, *row_lo, *row_hi, ...) is meaningful afterwards, each rank normalising its own slice.
int smesh_reduce_scatter(smesh_comm_t* c, smesh_aggregator_t* a, uint64_t* row_lo, uint64_t* row_hi) {
  if (!c || !a || !row_lo || !row_hi) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (smesh_aggregator_ctx(a) != c->ctx) return fail(SMESH_ERR_INVALID, "aggregator and communicator live on different devices");
  Rccl* r;
  SMESH_TRY(need_rccl(&r));
  std::lock_guard<std::mutex> g(smesh_aggregator_mutex(a));
  DeviceCtx* ctx = c->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  uint64_t count = 0, P = 0;
  uint32_t S = 0;
  float* acc = nullptr;
  SMESH_TRY(smesh_aggregator_acc(a, &acc, &count, &S, &P));
  const uint64_t q = (P / (uint64_t)c->nranks) & ~(uint64_t)3;
  *row_lo = q * (uint64_t)c->rank;
  *row_hi = c->rank == c->nranks - 1 ? P : q * (uint64_t)(c->rank + 1);
  if (count == 0) return SMESH_OK;
  ProfScope prof(ctx, SMESH_PROF_EXCHANGE);
  if (q) SMESH_RCCL(r, r->ReduceScatter(acc, acc + *row_lo * S, (size_t)(q * S), ncclFloat32, ncclSum, c->comm, ctx->stream));
  const uint64_t tail = q * (uint64_t)c->nranks;
  if (tail < P) SMESH_RCCL(r, r->AllReduce(acc + tail * S, acc + tail * S, (size_t)((P - tail) * S), ncclFloat32, ncclSum, c->comm, ctx->stream));
  if (c->nranks > 1) smesh_aggregator_mark_scattered(a, *row_lo, *row_hi);   // the other rows hold partial sums: get() / add() / a second exchange are refused until reset()
  return SMESH_OK;
}

// smesh_allreduce for the rows [row_lo, row_hi) only, on the EXCHANGE stream: it starts when everything queued so far on the main
// stream has finished (an event, no host wait) and runs beside whatever the main stream is given next -- the fusion of the next
// triangle range (smesh_fuse_views_continue), which does not touch these rows.  The main stream picks the result up at the next
// entry point that uses the accumulator (smesh_aggregator_join_exchange: a device-side wait).  Collectives of successive calls run
// in call order; every rank must issue the same sequence of ranges.
int smesh_allreduce_rows(smesh_comm_t* c, smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi) {
  if (!c || !a) return fail(SMESH_ERR_INVALID, "NULL communicator / aggregator");
  if (smesh_aggregator_ctx(a) != c->ctx) return fail(SMESH_ERR_INVALID, "aggregator and communicator live on different devices");
  Rccl* r;
  SMESH_TRY(need_rccl(&r));
  std::lock_guard<std::mutex> g(smesh_aggregator_mutex(a));
  DeviceCtx* ctx = c->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_TRY(smesh_aggregator_refuse_scattered(a, "allreduce_rows"));
  if (row_lo > row_hi || row_hi > smesh_aggregator_primitives(a)) return fail(SMESH_ERR_INVALID, "bad row range");
  if (row_lo == row_hi) return SMESH_OK;   // (empty on every rank: the ranges are a function of P and the part alone)
  hipEvent_t ev_part, ev_xchg;
  SMESH_TRY(smesh_aggregator_exchange_events(a, &ev_part, &ev_xchg));
  hipStream_t xs = ctx->exchange_stream;
  SMESH_HIP(hipEventRecord(ev_part, ctx->stream));
  SMESH_HIP(hipStreamWaitEvent(xs, ev_part, 0));
  void* buf = nullptr;
  uint64_t count = 0;
  int f64 = 0;
  SMESH_TRY(smesh_aggregator_exchange_begin(a, row_lo, row_hi, xs, &buf, &count, &f64));
  {
    ProfScope prof(ctx, SMESH_PROF_EXCHANGE, xs);
    SMESH_RCCL(r, r->AllReduce(buf, buf, (size_t)count, f64 ? ncclFloat64 : ncclFloat32, ncclSum, c->comm, xs));
  }
  SMESH_TRY(smesh_aggregator_exchange_end(a, row_lo, row_hi, xs));
  SMESH_HIP(hipEventRecord(ev_xchg, xs));
  smesh_aggregator_exchange_pending(a);
  return SMESH_OK;
}

// The main stream of the aggregator's device waits (device-side) for the row exchanges queued so far.  Every entry point that
// touches the accumulator does this itself; a harness calls it to place a time stamp behind the exchange (smesh_stream_mark).
int smesh_exchange_join(smesh_aggregator_t* a) {
  if (!a) return fail(SMESH_ERR_INVALID, "NULL aggregator");
  std::lock_guard<std::mutex> g(smesh_aggregator_mutex(a));
  DeviceCtx* ctx = smesh_aggregator_ctx(a);
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  return smesh_aggregator_join_exchange(a);
}

// Small reduction of host values over the same communicator (op: 0 = sum, 2 = max), blocking: what a benchmark harness needs
// for its barrier and its max-over-ranks clock without a second communication library.
int smesh_comm_allreduce_f64(smesh_comm_t* c, double* values, int n, int op) {
  if (!c || !values || n < 1 || n > smesh_comm::kSmall) return fail(SMESH_ERR_INVALID, "bad argument");
  if (op != 0 && op != 2) return fail(SMESH_ERR_INVALID, "op must be 0 (sum) or 2 (max)");
  Rccl* r;
  SMESH_TRY(need_rccl(&r));
  DeviceCtx* ctx = c->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_HIP(hipMemcpyAsync(c->d_small, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  SMESH_RCCL(r, r->AllReduce(c->d_small, c->d_small, (size_t)n, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->comm, ctx->stream));
  SMESH_HIP(hipMemcpyAsync(values, c->d_small, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

}  // extern "C"
