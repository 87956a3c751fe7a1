// common.hpp -- shared host-side plumbing of libsmesh_hip.so (device contexts, errors, timing).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/smesh.h"

namespace smesh {

int fail(int code, const std::string& msg);
int fail_hip(hipError_t e, const char* what, const char* file, int line);

#define SMESH_HIP(expr)                                                         \
  do {                                                                          \
    hipError_t _e = (expr);                                                     \
    if (_e != hipSuccess) return ::smesh::fail_hip(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define SMESH_TRY(expr)              \
  do {                               \
    int _s = (expr);                 \
    if (_s != SMESH_OK) return _s;   \
  } while (0)

struct ProfSlot {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;  // recorded, not yet read
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;     // free event pairs
  double total_ms = 0.0;
  uint64_t launches = 0;
  uint64_t seen = 0;   // regions entered while the slot was enabled (only every `profile_every`-th is bracketed)
  int depth = 0;       // > 0 inside a region of this slot: nested regions belong to it (back-to-back launches timed together)
  bool open_timed = false;   // the region that is open right now is one of the bracketed ones
  uint64_t timed_launches = 0;   // launches of the slot's dominant kernel inside bracketed regions (prof_note) ...
  uint64_t timed_views = 0;      // ... and the views those launches fused
};

// One per GPU: a compute stream all of this library's kernels and copies are ordered on.
struct DeviceCtx {
  int device = -1;
  hipStream_t stream = nullptr;         // main stream: fusion kernels, copies, everything by default
  hipStream_t raster_stream = nullptr;  // smesh_fuse_view rasterises view k+1 here while view k is being fused
  hipStream_t exchange_stream = nullptr;   // smesh_allreduce_rows: the collective of a finished row range runs here, beside the fusion of the next
  hipEvent_t ev_order = nullptr;        // smesh_stream_wait: marks the producer's stream
  hipEvent_t ev_release = nullptr;      // smesh_stream_release: marks the library's stream
  int num_cus = 256;
  unsigned profiling = 0;   // bit s set = bracket slot s with HIP events
  unsigned profile_every = 1;   // ... every n-th region of the slot (an event pair costs ~4 us of stream time)
  ProfSlot slots[SMESH_PROF_SLOTS];
  hipEvent_t marks[SMESH_STREAM_MARKS] = {};   // smesh_stream_mark
  std::vector<hipEvent_t> token_pool;          // smesh_token_record / _done: events waiting for their next use
  std::recursive_mutex mu;
};

// Returns the context of `device`, creating it (and its stream) on first use.
int get_ctx(int device, DeviceCtx** out);

// RAII: brackets a region of the context's stream with HIP events when profiling is enabled.
struct ProfScope {
  DeviceCtx* ctx;
  int slot;
  hipEvent_t start = nullptr, stop = nullptr;
  hipStream_t st;
  bool counted = false;   // this scope opened a region (as opposed to: profiling off, or nested in an open region)
  ProfScope(DeviceCtx* c, int s, hipStream_t stream = nullptr);
  ~ProfScope();
};

// Called where a slot's dominant kernel is launched: counts the launch and the views it processes if (and only if) it falls
// inside a region that is being timed, so that total_ms / timed_launches is that kernel's average duration whatever the
// grouping of views into calls was.
void prof_note(DeviceCtx* ctx, int slot, uint64_t launches, uint64_t views);

// Device memory of the library: hipMalloc / hipFree with a cache of still-mapped blocks in between (context.cpp: why).
// dev_malloc allocates on the CURRENT device; the contents of a block are unspecified; dev_free returns when the device is idle.
hipError_t dev_malloc(void** out, size_t bytes);
hipError_t dev_free(void* p, bool foreign = false);   // foreign: the caller could see the block -- wait for the whole device
uint64_t dev_cached_bytes(int device);
void dev_trim(int device);
void dev_cache_stats(uint64_t* live_blocks, uint64_t* cached_blocks, uint64_t* cached_bytes);

// Run-time options (smesh_set_option; the environment supplies the defaults).  group_pipeline: smesh_fuse_views rasterises group
// g + 1 on the raster stream beside the fusion of group g (default on; SMESH_GROUP_PIPELINE=0).
bool opt_group_pipeline();

// Grow-only device scratch buffer.
struct Scratch {
  void* ptr = nullptr;
  size_t bytes = 0;
  int reserve(size_t need);
  void release();
};

inline uint64_t div_up(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// Development ablations ("what does the kernel cost without its atomics / its stores / its class-vector loads": WRONG results)
// exist only in builds with -DSMESH_ABLATION (make ABLATION=1, tools/*ablation*.sh).  In the product build SMESH_ABL() is the
// constant 0 -- the branches are compiled out of the kernels -- and the SMESH_DBG / SMESH_FDBG / SMESH_RDBG / SMESH_REC_DBG
// environment variables are not read (tests/test_abi.py looks for their names in the shared library).
#ifdef SMESH_ABLATION
#define SMESH_ABL(bits) (bits)
#define SMESH_ABL_ENV(name) (getenv(name) ? atoi(getenv(name)) : 0)
#else
#define SMESH_ABL(bits) 0
#define SMESH_ABL_ENV(name) 0
#endif

// Per-triangle record the rasteriser leaves for the triangle-order fusion (smesh_fuse_view):
//   kind 1 (small): bounding box at (x0, y0) of at most 8 x 8 pixels; bit (dx * 8 + dy) of `mask` is set when
//                   the triangle emitted a fragment at (x0 + dx, y0 + dy) (it may still have lost the depth test);
//   kind 2 (big):   bounding box (x0, y0) .. (x1, y1) with x1 = mask & 0xFFFF, y1 = (mask >> 16) & 0xFFFF;
//   kind 0:         culled / nothing emitted.
struct TriFrag {
  uint16_t x0, y0;
  uint16_t kind;
  uint16_t pad;
  unsigned long long mask;
};
static_assert(sizeof(TriFrag) == 16, "TriFrag must be 16 bytes");

// Arguments of the triangle-order fusion kernels (fuse_tri.inc.hpp, fusion.hip).
struct TriFuseArgs {
  const TriFrag* frags;
  const uint32_t* idx;
  const float* probs;
  const float* weights;       // may be null
  float* acc;                 // [P][C] dense
  float* acc_lo;              // Mul only: second float32 plane, a row's value is acc + acc_lo (fuse_tri.inc.hpp, "Mul state"); else null
  uint64_t F;
  uint32_t C, W, H;
  float iew;
  const uint32_t* big_queue;
  const uint32_t* big_len;    // queue length of this render (emptied by the next render's vertex kernel)
  uint32_t big_capacity;
  uint32_t tri_blocks;        // blocks 0 .. tri_blocks-1 walk the triangles, the next big_blocks the big-triangle queue, the rest (k_fuse_tri with `mid`) the lists of medium triangles
  uint32_t big_blocks;
  int dbg;                    // development ablation (SMESH_FDBG): 1 stop after pass 1, 2 no stores, 4 no row loads, 8 no probs loads
  const uint32_t* prim_id;    // [F] primitive id of triangle f when the renderer re-ordered its triangles (null: id == f)
  // texel primitives (k_fuse_texel) only
  const uint32_t* tex_first;  // [F] first texel id of each triangle
  const uint32_t* tex_res;    // [F] texel resolution r: the triangle owns r (r + 1) / 2 consecutive texels
  const uint8_t* tex_kinds;   // [F] TriFrag::kind of every triangle, a byte each: a texel renderer leaves a record only where it is nonzero
  uint32_t* count;            // [P] scratch histogram, all zero between launches (big triangles only)
  double* acc_d;              // Mul + texel primitives: [P][C] sums of ONE view's terms, all zero between launches (big triangles only)
  int mid;                    // k_fuse_tri: nonzero = the launch's last workgroups (fuse_mid_entries) take the queued triangles with at most kMidBox pixels per view (the tail waves skip them)
  uint32_t ps0, ps1;          // k_fuse_tri / fuse_box only: element strides of x and y of the class-vector image (dense: H * C and C); the class stride is 1
  // Fusion by triangle RANGE (smesh_fuse_views_begin / _continue: the rows of a finished range are exchanged between GPUs while the
  // next range is fused).  Main block b of the launch takes the triangles of block blk_first + b; the waves that walk the queues of
  // big triangles take only triangles at positions [f_lo, f_hi).  A launch over everything: blk_first = 0, f_lo = 0, f_hi = F.
  uint32_t blk_first;
  uint32_t f_lo, f_hi;
  uint32_t lds_pad;           // host side only: dynamic LDS of the eight-view k_fuse_tri launch (caps its workgroups per CU beside the rasteriser, fusion_multi8.hip)
  uint32_t xcd_chunk = 0;     // k_fuse_tri_wide: nonzero = the triangle blocks are dealt to the XCDs (block b runs on XCD b % 8) in runs of xcd_chunk
                              // consecutive blocks (an experiment knob, SMESH_WIDE_XCD); 0 = blocks in dispatch order
};

// What k_fuse_tri needs to know about ONE of the views it fuses in a launch (the per-view part of TriFuseArgs), and NV of them.
struct TriView {
  const TriFrag* frags;
  const uint32_t* idx;
  const float* probs;
  const float* weights;       // may be null
  const uint32_t* big_queue;
  const uint32_t* big_len;    // [0] queue length, [1] "check the masks against the index plane" flag of the render
  uint32_t W, H;
  uint32_t ps0, ps1;          // k_fuse_tri only: element strides of x and y of `probs` (dense: H * C and C; a (H,W,C) tensor seen as (W,H,C): C and W * C)
  const uint8_t* kinds;       // k_fuse_texel_multi only: TriFrag::kind of every triangle, a byte each (null: read the records)
};
template <int NV>
struct TriViews {
  TriView v[NV];
};

// One rendered view as the triangle-order fusion consumes it (raster.hip -> fusion.hip).
struct RenderedView {
  const TriFrag* frags;         // per-triangle fragment records of the render
  const uint32_t* big_queue;    // triangles with a bounding box over 8 x 8 pixels ...
  const uint32_t* big_len;      // ... and how many (device counter)
  const uint32_t* idx;          // index plane [W][H]
  const float* probs;           // [W][H][C], device
  const float* weights;         // [W][H] or null, device
  uint64_t W, H;
  int64_t ps0 = 0, ps1 = 0;     // element strides of x and y of `probs` when it is not the dense (W,H,C) image (class stride 1); 0, 0: dense
  bool mid_queue = false;       // big_queue[big_capacity ...] lists the medium triangles of this view, big_len[3] of them (the rasteriser's renders)
  bool no_big = false;          // PROVEN on the host (raster.hip no_big_possible): no triangle of this view has a box over 8 x 8 pixels -- big_queue is empty
  const uint8_t* kinds = nullptr;   // texel renderers: TriFrag::kind of every triangle again, a byte each (RasterArgs::kinds)
  bool fine = false;            // bounded on the host (raster.hip box_extent_bound <= 48 pixels): a finely tessellated mesh seen from outside -- few or no queued triangles
};

// Medium triangles: a bounding box over 8 x 8 pixels of at most kMidBox pixels (16 x 16: beyond that a whole wave per triangle --
// fuse_box -- is faster than sixteen lanes, tools/mesh_density_sweep.py).  The rasteriser lists them (push_mid), the last workgroups of the k_fuse_tri launch fuse them (fuse_mid.inc.hpp).
constexpr int kMidBox = 256;

// Per-primitive records built from an arbitrary index image (image_records.hip): what lets MeshAggregator::add() run the
// triangle-order fusion on images the library did not render.  Scratch of one aggregator, all zero between calls.
struct ImageRecords {
  TriFrag* frags = nullptr;        // [P] (start of the one allocation)
  uint32_t* cand = nullptr;        // [P] ~(x << 16 | y) of the primitive's first pixel in (x, y) order; 0 = none
  uint4* big4 = nullptr;           // [P] primitives with runs outside the 8 x 8 box: largest x + 1, largest y + 1, pixels in those
                                   //     runs, 65536 - smallest y; zero otherwise
  uint32_t* big_queue = nullptr;   // [P] those primitives
  uint32_t* big_count = nullptr;   // [4] [0] queue length, [1] stays 0: the masks are exact, [2] nonzero: there are sparse primitives
  unsigned long long* mom = nullptr;   // [P] moments of the current image (count | sum x << 24 | sum y << 44), zero between calls
  uint64_t P = 0;
  bool clean = false;              // frags / cand / counters are all zero (what passes A and B start from)
  bool moments = false;            // the last build took passes M / R (image_records.hip)
  uint32_t tag = 2;                // passes M / R: what marks a record written by THIS call's pass M (alternates 2, 4)
  void release();
};
int image_records_build(DeviceCtx* ctx, ImageRecords& r, const uint32_t* d_idx, uint64_t W, uint64_t H, uint64_t P);
int image_records_pending(DeviceCtx* ctx, ImageRecords& r, const uint32_t* d_idx, uint64_t W, uint64_t H, hipStream_t st);
int image_records_scatter_sparse(DeviceCtx* ctx, ImageRecords& r, int kind, const uint32_t* d_idx, const float* d_probs, const float* d_w,
                                 uint64_t W, uint64_t H, uint32_t C, float iew, float* acc, float* acc_lo, double* acc_d, hipStream_t st);
int image_records_clear(DeviceCtx* ctx, ImageRecords& r, const uint32_t* d_idx, uint64_t W, uint64_t H, hipStream_t st);
bool image_records_build_group(DeviceCtx* ctx, ImageRecords* const* recs, const uint32_t* const* d_idx, int n, uint64_t W, uint64_t H, uint64_t P,
                               int* status);
int image_records_scatter_sparse_group(DeviceCtx* ctx, ImageRecords* const* recs, int n, int kind, const uint32_t* const* d_idx,
                                       const float* const* d_probs, const float* const* d_w, uint64_t W, uint64_t H, uint32_t C, float iew,
                                       float* acc, float* acc_lo, double* acc_d, hipStream_t st);

}  // namespace smesh
