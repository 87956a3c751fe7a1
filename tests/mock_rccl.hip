// mock_rccl.hip -- a stand-in for librccl with RCCL's GROUP SEMANTICS, for the one-GPU test box (test infrastructure only).
//
// RCCL refuses two ranks on one device, so the grouped collectives of a single process driving several GPUs
// (smesh_comm_create_all / smesh_allreduce with n > 1) cannot run on the box that runs `pytest -m gpu`.  This library exports
// the handful of entry points comm.cpp loads (SMESH_RCCL_LIB points at it) and keeps the one property that matters for stream
// ordering: between ncclGroupStart and ncclGroupEnd an ncclAllReduce call only RECORDS the operation -- the reduction is put on
// the streams by ncclGroupEnd.  Code that queues a consumer of the result right behind its ncclAllReduce call, inside the
// group, therefore reads unreduced data here exactly as it does with the real library (comm.cpp's Mul epilogue).
// "Ranks" are the communicators of ONE ncclCommInitAll call, all on the device the test uses.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {
typedef struct mock_comm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclRedOp_t;
typedef int ncclDataType_t;
}

struct mock_comm { int rank, nranks; };

namespace {
struct Op { const void* send; void* recv; size_t count; int dtype; hipStream_t st; mock_comm* c; };
std::vector<Op> g_ops;
int g_depth = 0;
int g_groups_with_work = 0;

template <typename T>
__global__ void k_sum(const T* const* __restrict__ in, int n, T* __restrict__ out, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  T s = (T)0;
  for (int r = 0; r < n; r++) s += in[r][i];
  out[i] = s;
}

int flush() {
  if (g_ops.empty()) return 0;
  g_groups_with_work++;
  const size_t count = g_ops[0].count;
  const int dtype = g_ops[0].dtype;
  const size_t esize = dtype == 8 ? 8 : 4;
  for (const Op& o : g_ops)
    if (o.count != count || o.dtype != dtype || (int)g_ops.size() != o.c->nranks) { g_ops.clear(); return 1; }
  const int n = (int)g_ops.size();
  void* tmp = nullptr;
  const void** d_in = nullptr;
  if (hipMalloc(&tmp, count * esize) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d_in), n * sizeof(void*)) != hipSuccess) return 1;
  std::vector<const void*> in((size_t)n);
  for (int r = 0; r < n; r++) in[(size_t)r] = g_ops[(size_t)r].send;
  hipStream_t st = g_ops[0].st;
  // every participant's stream has to reach this point before the sum is taken (the test's communicators share one stream; for
  // distinct streams a host wait does: this is a mock)
  for (const Op& o : g_ops)
    if (o.st != st) (void)hipStreamSynchronize(o.st);
  (void)hipMemcpyAsync(d_in, in.data(), n * sizeof(void*), hipMemcpyHostToDevice, st);
  const dim3 grid((unsigned)((count + 255) / 256)), block(256);
  if (dtype == 8) hipLaunchKernelGGL(k_sum<double>, grid, block, 0, st, reinterpret_cast<const double* const*>(d_in), n, static_cast<double*>(tmp), count);
  else hipLaunchKernelGGL(k_sum<float>, grid, block, 0, st, reinterpret_cast<const float* const*>(d_in), n, static_cast<float*>(tmp), count);
  for (const Op& o : g_ops) (void)hipMemcpyAsync(o.recv, tmp, count * esize, hipMemcpyDeviceToDevice, st);
  (void)hipStreamSynchronize(st);     // (tmp / d_in are freed here; the CALLER's later launches are still ordered behind the copies)
  (void)hipFree(tmp);
  (void)hipFree(d_in);
  g_ops.clear();
  return 0;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 7, sizeof *id); return 0; }
ncclResult_t ncclCommInitRank(ncclComm_t* c, int nranks, ncclUniqueId, int rank) {
  if (nranks != 1) return 5;
  *c = new mock_comm{rank, nranks};
  return 0;
}
ncclResult_t ncclCommInitAll(ncclComm_t* c, int n, const int*) {
  for (int i = 0; i < n; i++) c[i] = new mock_comm{i, n};
  return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return 0; }
ncclResult_t ncclGroupStart() { g_depth++; return 0; }
ncclResult_t ncclGroupEnd() {
  if (--g_depth > 0) return 0;
  return flush() ? 5 : 0;
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
  if (op != 0 || (dtype != 7 && dtype != 8)) return 5;
  g_ops.push_back(Op{send, recv, count, dtype, st, c});
  if (g_depth == 0) return flush() ? 5 : 0;     // outside a group: a communicator of one rank (an identity)
  return 0;
}
ncclResult_t ncclReduceScatter(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) { return 5; }
const char* ncclGetErrorString(ncclResult_t) { return "mock_rccl: unsupported call or mismatched group"; }
int mock_rccl_groups_with_work() { return g_groups_with_work; }
}
