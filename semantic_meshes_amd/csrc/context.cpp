// context.cpp -- device contexts, error strings, memory helpers and the timing hooks of the C ABI.
#include "common.hpp"

#include <algorithm>
#include <atomic>

#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <unordered_map>
#include <vector>

namespace smesh {

static thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

int fail_hip(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  g_err = buf;
  // clear the sticky error so that later calls report their own failures
  (void)hipGetLastError();
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? SMESH_ERR_NODEVICE : SMESH_ERR_RUNTIME;
}

static std::atomic<int> g_group_pipeline{-1};   // -1: not decided yet (the environment, else on)
bool opt_group_pipeline() {
  int v = g_group_pipeline.load();
  if (v < 0) {
    const char* e = getenv("SMESH_GROUP_PIPELINE");
    v = e ? (atoi(e) != 0 ? 1 : 0) : 1;
    g_group_pipeline.store(v);
  }
  return v != 0;
}

static std::mutex g_ctx_mu;
static std::vector<std::unique_ptr<DeviceCtx>> g_ctx;

// The context of `device` if one exists already (never creates one).
DeviceCtx* peek_ctx(int device) {
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  return (device >= 0 && device < (int)g_ctx.size()) ? g_ctx[device].get() : nullptr;
}

int get_ctx(int device, DeviceCtx** out) {
  std::lock_guard<std::mutex> lock(g_ctx_mu);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(SMESH_ERR_NODEVICE, "no HIP device available (libsmesh_hip has no CPU fallback)");
  }
  if (device < 0 || device >= n) return fail(SMESH_ERR_INVALID, "device index out of range");
  if ((int)g_ctx.size() < n) g_ctx.resize(n);
  if (!g_ctx[device]) {
    SMESH_HIP(hipSetDevice(device));
    auto ctx = std::make_unique<DeviceCtx>();
    ctx->device = device;
    hipDeviceProp_t prop;
    SMESH_HIP(hipGetDeviceProperties(&prop, device));
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // (Round 6 measured the two streams confined to disjoint sets of CUs -- hipExtStreamCreateWithCUMask, 4 or 5 of every 8 CUs for the
    // rasteriser -- so that the group pipeline's two kernels would not share CUs: 15 442 / 15 499 views/s against 15 370 with masks
    // that interleave inside an XCD, 9 000 - 11 000 with masks that give whole XCDs away.  Not kept.)
    SMESH_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    SMESH_HIP(hipStreamCreateWithFlags(&ctx->raster_stream, hipStreamNonBlocking));
    {
      // the exchange stream gets the greatest priority there is: a collective's few workgroups must not queue behind the tens of
      // thousands of one-wave workgroups of the fusion launch it runs beside
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
      if (hipStreamCreateWithPriority(&ctx->exchange_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
        (void)hipGetLastError();
        SMESH_HIP(hipStreamCreateWithFlags(&ctx->exchange_stream, hipStreamNonBlocking));
      }
    }
    g_ctx[device] = std::move(ctx);
  }
  *out = g_ctx[device].get();
  return SMESH_OK;
}

ProfScope::ProfScope(DeviceCtx* c, int s, hipStream_t stream) : ctx(c), slot(s), st(stream ? stream : c->stream) {
  if (!(ctx->profiling & (1u << slot))) return;
  ProfSlot& ps = ctx->slots[slot];
  if (ps.depth++ > 0) { counted = true; return; }   // inside an open region of this slot
  counted = true;
  // (sampling applies to the per-view slots; get() and the exchange are rare and always bracketed)
  if (ps.seen++ % (slot <= SMESH_PROF_RASTER ? std::max(1u, ctx->profile_every) : 1u) != 0) return;
  if (!ps.pool.empty()) {
    start = ps.pool.back().first;
    stop = ps.pool.back().second;
    ps.pool.pop_back();
  } else {
    if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) {
      start = stop = nullptr;
      return;
    }
  }
  (void)hipEventRecord(start, st);
  ps.open_timed = true;
}

void prof_note(DeviceCtx* ctx, int slot, uint64_t launches, uint64_t views) {
  ProfSlot& ps = ctx->slots[slot];
  if (ps.depth > 0 && ps.open_timed) { ps.timed_launches += launches; ps.timed_views += views; }
}

ProfScope::~ProfScope() {
  if (counted && --ctx->slots[slot].depth == 0 && !start) ctx->slots[slot].open_timed = false;
  if (!start) return;
  ctx->slots[slot].open_timed = false;
  (void)hipEventRecord(stop, st);
  ctx->slots[slot].pending.emplace_back(start, stop);
}

// ---- device memory: blocks are kept mapped and handed out again ------------------------------------------------------------
// Why the library does not hipFree what it may need again (round 5; tests/flake_hunt.py, tools/alloc_churn_repro.hip,
// profiles/r05_lost_writes_*): on a GPU shared by several processes with a few busy streams each, a kernel's writes into a buffer
// that hipMalloc has JUST mapped are, about once in 20 000 allocations, missing for every workgroup that ran on one of the eight
// XCDs -- the buffer keeps the zeros it came with, no error is raised; a stand-alone HIP program shows it (no libsmesh involved),
// one process alone never does, and memory that stays mapped never does.  That was round 4's "unexplained failure": an eighth of a
// class-vector image read as don't-care pixels.  So device blocks freed by the library's handles and by smesh_device_free go to a
// per-process cache (still mapped) and are handed out again for requests of the same size class; smesh_device_trim() or the cap
// (SMESH_ALLOC_CACHE_MB per device, default a sixteenth of the device's memory, at most 8 GiB; 0 = plain hipMalloc / hipFree) unmaps them.
// dev_free keeps hipFree's ordering contract for the work that can touch the block: nothing queued still uses it on return.
namespace {
struct DevBlock { int device; size_t bytes; };
struct CachedBlock { void* ptr; int device; unsigned long long stamp; };
std::mutex g_alloc_mu;
std::unordered_map<void*, DevBlock> g_live;                 // handed out: size class of the block
std::multimap<size_t, CachedBlock> g_cached;                // free, still mapped: size class -> block
std::map<int, size_t> g_cached_bytes;                       // per device: bytes held in g_cached
unsigned long long g_alloc_stamp = 0;

size_t size_class(size_t n) {
  if (n == 0) n = 1;
  const size_t g = n <= (1u << 20) ? (size_t)4096 : (size_t)2 << 20;
  return (n + g - 1) / g * g;
}
// The cap is PER DEVICE (one cap sized from whichever device was current first was shared by all of them): what the
// library reallocates again and again -- record sets, fragment queues, scratch, index planes, a cfg2-sized accumulator -- fits in a
// few GiB; a destroyed 12 GB cfg5 accumulator is larger than the cap and really freed, so that other allocators of the process
// (torch, cupy) find the memory.  smesh_device_trim() gives back the rest on request.
size_t cache_cap(int device) {
  static std::map<int, size_t> caps;          // (g_alloc_mu held)
  auto it = caps.find(device);
  if (it != caps.end()) return it->second;
  size_t cap;
  if (const char* e = getenv("SMESH_ALLOC_CACHE_MB")) cap = (size_t)std::max(0ll, atoll(e)) << 20;
  else {
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    size_t free_b = 0, total = 0;
    if (hipMemGetInfo(&free_b, &total) != hipSuccess) { (void)hipGetLastError(); total = (size_t)64 << 30; }
    if (cur != device) (void)hipSetDevice(cur);
    cap = std::min<size_t>(total / 16, (size_t)8 << 30);
  }
  caps[device] = cap;
  return cap;
}
void really_free_locked(std::multimap<size_t, CachedBlock>::iterator it) {
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != it->second.device) (void)hipSetDevice(it->second.device);
  (void)hipFree(it->second.ptr);
  if (cur != it->second.device) (void)hipSetDevice(cur);
  g_cached_bytes[it->second.device] -= it->first;
  g_cached.erase(it);
}
// (g_alloc_mu held) really frees cached blocks of `device` (-1: any device), oldest first, until `need` more bytes fit under `cap`
void evict_locked(int device, size_t need, size_t cap) {
  for (;;) {
    size_t held = 0;
    if (device >= 0) held = g_cached_bytes[device];
    else for (auto& kv : g_cached_bytes) held += kv.second;
    if (held == 0 || held + need <= cap) return;
    auto oldest = g_cached.end();
    for (auto it = g_cached.begin(); it != g_cached.end(); ++it)
      if ((device < 0 || it->second.device == device) && (oldest == g_cached.end() || it->second.stamp < oldest->second.stamp)) oldest = it;
    if (oldest == g_cached.end()) return;
    really_free_locked(oldest);
  }
}
}  // namespace

hipError_t dev_malloc(void** out, size_t bytes) {
  *out = nullptr;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const size_t sz = size_class(bytes);
  std::lock_guard<std::mutex> lock(g_alloc_mu);
  for (auto it = g_cached.lower_bound(sz); it != g_cached.end() && it->first <= sz + sz / 8; ++it) {
    if (it->second.device != dev) continue;
    *out = it->second.ptr;
    g_live[*out] = DevBlock{dev, it->first};
    g_cached_bytes[dev] -= it->first;
    g_cached.erase(it);
    return hipSuccess;
  }
  e = hipMalloc(out, sz);
  if (e != hipSuccess) {      // out of memory: give back what the cache holds on this device and try once more
    (void)hipGetLastError();
    evict_locked(dev, 0, 0);
    e = hipMalloc(out, sz);
  }
  if (e == hipSuccess) g_live[*out] = DevBlock{dev, sz};
  return e;
}

// `foreign`: the block was visible to the caller (smesh_device_malloc, exported planes) and may have been used on streams the
// library knows nothing about -- wait for the whole device, as hipFree would.  Blocks only the library's handles ever touched
// (scratch, queues, records, accumulators) are used on the library's three streams only: those are waited for, not other
// frameworks' work on the device.  Either way it is the BLOCK'S device that is waited for, whichever is current.
hipError_t dev_free(void* p, bool foreign) {
  if (!p) return hipSuccess;
  int owner = -1;
  {
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    auto it = g_live.find(p);
    if (it != g_live.end()) owner = it->second.device;
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (owner >= 0 && owner != cur) (void)hipSetDevice(owner);
  DeviceCtx* ctx = (owner >= 0 && !foreign) ? peek_ctx(owner) : nullptr;
  hipError_t e = hipSuccess;
  if (ctx) {
    for (hipStream_t st : {ctx->raster_stream, ctx->stream, ctx->exchange_stream})
      if (st && e == hipSuccess) e = hipStreamSynchronize(st);
  } else {
    e = hipDeviceSynchronize();
  }
  if (e != hipSuccess) (void)hipGetLastError();
  hipError_t ret = hipSuccess;
  {
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) ret = hipFree(p);      // (not one of ours)
    else {
      const DevBlock b = it->second;
      g_live.erase(it);
      const size_t cap = cache_cap(b.device);
      if (b.bytes > cap) ret = hipFree(p);
      else {
        evict_locked(b.device, b.bytes, cap);
        g_cached.emplace(b.bytes, CachedBlock{p, b.device, g_alloc_stamp++});
        g_cached_bytes[b.device] += b.bytes;
      }
    }
  }
  if (owner >= 0 && owner != cur) (void)hipSetDevice(cur);
  return ret;
}

void dev_trim(int device) {
  std::lock_guard<std::mutex> lock(g_alloc_mu);
  for (auto it = g_cached.begin(); it != g_cached.end();) {
    if (device >= 0 && it->second.device != device) { ++it; continue; }
    auto victim = it++;
    really_free_locked(victim);
  }
}

void dev_cache_stats(uint64_t* live_blocks, uint64_t* cached_blocks, uint64_t* cached_bytes) {
  std::lock_guard<std::mutex> lock(g_alloc_mu);
  if (live_blocks) *live_blocks = g_live.size();
  if (cached_blocks) *cached_blocks = g_cached.size();
  if (cached_bytes) { *cached_bytes = 0; for (auto& kv : g_cached_bytes) *cached_bytes += kv.second; }
}
uint64_t dev_cached_bytes(int device) {
  std::lock_guard<std::mutex> lock(g_alloc_mu);
  uint64_t n = 0;
  for (auto& kv : g_cached_bytes) if (device < 0 || kv.first == device) n += kv.second;
  return n;
}

int Scratch::reserve(size_t need) {
  if (need <= bytes) return SMESH_OK;
  if (ptr) SMESH_HIP(dev_free(ptr));
  ptr = nullptr;
  bytes = 0;
  size_t want = need + need / 8 + 256;  // slack so that slowly growing images do not reallocate every call
  SMESH_HIP(dev_malloc(&ptr, want));
  bytes = want;
  return SMESH_OK;
}

void Scratch::release() {
  if (ptr) (void)dev_free(ptr);
  ptr = nullptr;
  bytes = 0;
}

}  // namespace smesh

using namespace smesh;

extern "C" {

const char* smesh_backend(void) { return "hip-gfx950"; }

const char* smesh_last_error(void) { return g_err.c_str(); }

// Run-time options by name.  "group_pipeline" (0 / 1; default 1, or SMESH_GROUP_PIPELINE): smesh_fuse_views rasterises the next
// group of views on a second stream beside the fusion of the current one -- same kernels, same inputs, same results; a harness
// that wants one kernel's own duration (a roofline) turns it off for that measurement.
int smesh_set_option(const char* name, int64_t value) {
  if (!name) return fail(SMESH_ERR_INVALID, "option name is NULL");
  if (!strcmp(name, "group_pipeline")) { g_group_pipeline.store(value != 0 ? 1 : 0); return SMESH_OK; }
  return fail(SMESH_ERR_INVALID, std::string("unknown option: ") + name);
}
int smesh_get_option(const char* name, int64_t* value) {
  if (!name || !value) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (!strcmp(name, "group_pipeline")) { *value = opt_group_pipeline() ? 1 : 0; return SMESH_OK; }
  return fail(SMESH_ERR_INVALID, std::string("unknown option: ") + name);
}

int smesh_device_count(int* count) {
  if (!count) return fail(SMESH_ERR_INVALID, "count is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return SMESH_OK;
}

int smesh_synchronize(int device) {
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  SMESH_HIP(hipSetDevice(device));
  SMESH_HIP(hipStreamSynchronize(ctx->raster_stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->exchange_stream));
  return SMESH_OK;
}

// Orders the library's streams after everything queued so far on `producer_stream` (NULL = the legacy default stream, on
// which PyTorch-ROCm runs unless told otherwise): the library's streams are non-blocking, so device buffers written by
// another framework are NOT implicitly ordered before the kernels that read them (the reference is: synchronous cudaMemcpy
// on the null stream, Fusion.h:35-37).  Costs an event record + two stream waits, no host synchronisation.
int smesh_stream_wait(int device, void* producer_stream) {
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(device));
  hipStream_t ps = static_cast<hipStream_t>(producer_stream);
  if (ps == ctx->raster_stream) return SMESH_OK;
  if (!ctx->ev_order) SMESH_HIP(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
  SMESH_HIP(hipEventRecord(ctx->ev_order, ps));
  // the library's MAIN stream as producer: a DLPack producer was handed that stream (`__dlpack__(stream=...)`) and ordered its
  // writes on it -- the raster stream (content checksums of foreign images run there) still has to be put behind them
  if (ps != ctx->stream) SMESH_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_order, 0));
  SMESH_HIP(hipStreamWaitEvent(ctx->raster_stream, ctx->ev_order, 0));
  return SMESH_OK;
}

// The other direction: `consumer_stream` (NULL = the legacy default stream) waits for everything this library has queued so far on
// `device`.  After an asynchronous entry point (smesh_fuse_view, smesh_aggregator_add_rendered ...) read DEVICE buffers that another
// framework will overwrite or free on that stream, this keeps the framework's later work behind the library's reads without
// blocking the host.
int smesh_stream_release(int device, void* consumer_stream) {
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(device));
  hipStream_t cs = static_cast<hipStream_t>(consumer_stream);
  if (cs == ctx->stream) return SMESH_OK;
  if (!ctx->ev_release) SMESH_HIP(hipEventCreateWithFlags(&ctx->ev_release, hipEventDisableTiming));
  SMESH_HIP(hipEventRecord(ctx->ev_release, ctx->stream));
  SMESH_HIP(hipStreamWaitEvent(cs, ctx->ev_release, 0));
  return SMESH_OK;
}

// The library's main stream (a hipStream_t) of `device`: consumers that want to order their own work after the library's
// without a host synchronisation (`__cuda_array_interface__` v3 "stream", `__dlpack__(stream=...)`).
int smesh_stream_handle(int device, void** stream) {
  if (!stream) return fail(SMESH_ERR_INVALID, "stream is NULL");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  *stream = ctx->stream;
  return SMESH_OK;
}

int smesh_stream_mark(int device, int id) {
  if (id < 0 || id >= SMESH_STREAM_MARKS) return fail(SMESH_ERR_INVALID, "bad mark id");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(device));
  if (!ctx->marks[id]) SMESH_HIP(hipEventCreate(&ctx->marks[id]));
  SMESH_HIP(hipEventRecord(ctx->marks[id], ctx->stream));
  return SMESH_OK;
}

int smesh_stream_mark_elapsed(int device, int from, int to, double* ms) {
  if (from < 0 || from >= SMESH_STREAM_MARKS || to < 0 || to >= SMESH_STREAM_MARKS || !ms) return fail(SMESH_ERR_INVALID, "bad mark id");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  hipEvent_t a, b;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    a = ctx->marks[from];
    b = ctx->marks[to];
  }
  if (!a || !b) return fail(SMESH_ERR_INVALID, "mark was never recorded");
  SMESH_HIP(hipSetDevice(device));
  SMESH_HIP(hipEventSynchronize(b));
  float f = 0.f;
  SMESH_HIP(hipEventElapsedTime(&f, a, b));
  *ms = (double)f;
  return SMESH_OK;
}

// Completion tokens: an event behind everything queued so far on the library's streams of `device`.  What the asynchronous entry
// points need from their caller -- "keep the DEVICE images valid until the kernels have read them" -- becomes checkable without a
// host wait: the Python layer holds its references to the inputs of add() / fuse_view(s)() until their token is done.
int smesh_token_record(int device, uint64_t* token) {
  if (!token) return fail(SMESH_ERR_INVALID, "token is NULL");
  *token = 0;
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(device));
  hipEvent_t ev = nullptr;
  if (!ctx->token_pool.empty()) { ev = ctx->token_pool.back(); ctx->token_pool.pop_back(); }
  else SMESH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  // the main stream also stands for the exchange stream's reads of the accumulator; images are only ever read on the main stream
  // (and, for content checksums of foreign index images, on the raster stream, which add_matched waits for itself)
  SMESH_HIP(hipEventRecord(ev, ctx->stream));
  *token = (uint64_t)reinterpret_cast<uintptr_t>(ev);
  return SMESH_OK;
}

int smesh_token_done(int device, uint64_t token, int* done) {
  if (!done) return fail(SMESH_ERR_INVALID, "done is NULL");
  *done = 1;
  if (!token) return SMESH_OK;
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  hipEvent_t ev = reinterpret_cast<hipEvent_t>((uintptr_t)token);
  const hipError_t e = hipEventQuery(ev);
  if (e == hipErrorNotReady) { (void)hipGetLastError(); *done = 0; return SMESH_OK; }
  if (e != hipSuccess) return fail_hip(e, "hipEventQuery", __FILE__, __LINE__);
  ctx->token_pool.push_back(ev);     // done: the token is spent
  return SMESH_OK;
}

int smesh_profile_enable(int device, int enabled) {
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  ctx->profiling = enabled < 0 ? 0xFFu : (unsigned)enabled;
  return SMESH_OK;
}

int smesh_profile_sample_every(int device, uint32_t n) {
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  ctx->profile_every = n ? n : 1u;
  return SMESH_OK;
}

static int drain(DeviceCtx* ctx) {
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_HIP(hipStreamSynchronize(ctx->raster_stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->exchange_stream));
  for (auto& ps : ctx->slots) {
    for (auto& ev : ps.pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
        ps.total_ms += ms;
        ps.launches += 1;
      }
      ps.pool.push_back(ev);
    }
    ps.pending.clear();
  }
  return SMESH_OK;
}

int smesh_profile_read(int device, int slot, double* total_ms, uint64_t* launches) {
  if (slot < 0 || slot >= SMESH_PROF_SLOTS) return fail(SMESH_ERR_INVALID, "bad profile slot");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_TRY(drain(ctx));
  if (total_ms) *total_ms = ctx->slots[slot].total_ms;
  if (launches) *launches = ctx->slots[slot].launches;
  return SMESH_OK;
}

int smesh_profile_read_ex(int device, int slot, double* total_ms, uint64_t* regions, uint64_t* launches, uint64_t* views) {
  if (slot < 0 || slot >= SMESH_PROF_SLOTS) return fail(SMESH_ERR_INVALID, "bad profile slot");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_TRY(drain(ctx));
  const ProfSlot& ps = ctx->slots[slot];
  if (total_ms) *total_ms = ps.total_ms;
  if (regions) *regions = ps.launches;
  if (launches) *launches = ps.timed_launches;
  if (views) *views = ps.timed_views;
  return SMESH_OK;
}

int smesh_profile_regions(int device, int slot, uint64_t* entered) {
  if (slot < 0 || slot >= SMESH_PROF_SLOTS) return fail(SMESH_ERR_INVALID, "bad profile slot");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (entered) *entered = ctx->slots[slot].seen;
  return SMESH_OK;
}

int smesh_profile_reset(int device) {
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_TRY(drain(ctx));
  for (auto& ps : ctx->slots) {
    ps.total_ms = 0.0;
    ps.launches = 0;
    ps.seen = 0;
    ps.timed_launches = 0;
    ps.timed_views = 0;
  }
  return SMESH_OK;
}

int smesh_device_malloc(int device, uint64_t bytes, void** out) {
  if (!out) return fail(SMESH_ERR_INVALID, "out is NULL");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  SMESH_HIP(hipSetDevice(device));
  SMESH_HIP(dev_malloc(out, bytes ? bytes : 1));
  return SMESH_OK;
}

int smesh_device_free(int device, void* ptr) {
  if (!ptr) return SMESH_OK;
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  SMESH_HIP(hipSetDevice(device));
  SMESH_HIP(dev_free(ptr, /*foreign=*/true));      // (waits for the device, like hipFree; the block stays mapped for the next request of its size)
  return SMESH_OK;
}

// Unmaps the device blocks the library keeps for reuse on `device` (-1: every device).  `cached_bytes` (may be NULL) receives what
// was being held.  Nothing else changes: live handles keep their memory.
int smesh_device_trim(int device, uint64_t* cached_bytes) {
  if (cached_bytes) *cached_bytes = dev_cached_bytes(device);
  if (device >= 0) {
    DeviceCtx* ctx;
    SMESH_TRY(get_ctx(device, &ctx));
    SMESH_HIP(hipSetDevice(device));
  }
  (void)hipDeviceSynchronize();
  dev_trim(device);
  return SMESH_OK;
}

// Page-locked host memory: host images handed to add() / fuse_view() from such a buffer cross PCIe by DMA at link speed
// (pageable memory is staged through the driver's bounce buffers).
int smesh_host_malloc(uint64_t bytes, void** out) {
  if (!out) return fail(SMESH_ERR_INVALID, "out is NULL");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(0, &ctx));   // needs a HIP runtime with a device
  SMESH_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return SMESH_OK;
}

int smesh_host_free(void* ptr) {
  if (!ptr) return SMESH_OK;
  SMESH_HIP(hipHostFree(ptr));
  return SMESH_OK;
}

int smesh_memcpy(void* dst, const void* src, uint64_t bytes, int dst_memkind, int src_memkind, int device) {
  if (bytes == 0) return SMESH_OK;
  if (!dst || !src) return fail(SMESH_ERR_INVALID, "NULL pointer");
  if (dst_memkind == SMESH_MEM_HOST && src_memkind == SMESH_MEM_HOST) {
    memmove(dst, src, bytes);
    return SMESH_OK;
  }
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(device));
  hipMemcpyKind kind = dst_memkind == SMESH_MEM_HOST ? hipMemcpyDeviceToHost
                       : (src_memkind == SMESH_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice);
  // ordered after everything already queued on the library stream
  SMESH_HIP(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

}  // extern "C"
