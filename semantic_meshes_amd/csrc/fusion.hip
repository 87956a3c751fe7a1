// fusion.hip -- MeshAggregator on MI355X (gfx950): per-view histogram, segmented scatter-add, finalize.
//
// Replaces the reference's host-side fusion (citations relative to /root/reference):
//   include/semantic_meshes/fusion/Mesh.h:90-93   serial std::map histogram        -> k_hist_strip
//   include/semantic_meshes/fusion/Mesh.h:94-106  OpenMP loop + per-primitive mutex -> k_scatter_strip
//   python/semantic_meshes/src/Fusion.cu:46-92    Summax / Sum / Mul aggregators    -> KIND template
//   python/semantic_meshes/include/Fusion.h:26-40 TensorConstructor copy+cast       -> k_gather_* (only
//                                                  when the caller's layout is not already contiguous)
//   Fusion.h:79-104 + Fusion.cu:47-49,67-69,79-82 get() functor chain               -> k_finalize_tile
//
// Data layout in HBM: accumulator float32[P][S] with the row stride S = C rounded up to 16 floats, so that
// every primitive's row starts on a 64-byte boundary (C = 19 -> one 128-byte line per row); the padding
// stays zero.  get()/get_raw() return dense [P][C].  Per-view histogram uint32[P], zero between add() calls.
//
// The scatter-add is HBM-bound (no MFMA): per view it must read 4*N (indices) + 4*N*C (probs) bytes and
// read-modify-write 2*4*C*T accumulator bytes (T = distinct primitives touched).  Measured on MI355X
// (tools/atomic_bench.hip, tools/flush_replay.hip): float atomics execute memory-side and cost
// roughly 30 ps per request + 0.5 ps per byte, LDS float atomics are an order of magnitude too slow, and
// workgroup barriers / dependent LDS round trips dominate a tile-per-workgroup design.  Hence:
//   * one WAVE = one strip of 4 columns x 16 rows (images are y-fastest: 4 column segments of 16
//     consecutive pixels); no workgroup barriers, every wave is independent,
//   * the strip's probs (4 x 16*C contiguous floats) are streamed with 16-byte coalesced loads into LDS and
//     read back one pixel row per lane; the don't-care test and the weight are applied in registers,
//   * a wave ballot splits each column into same-primitive runs; a segmented suffix scan over the 16-lane
//     rows folds each run into its head lane; runs of the same primitive in neighbouring columns are
//     linked (mutual first match, 8-connectivity) into chains = groups,
//   * lanes then own (group, class) elements, add up the <= 4 run totals of the chain out of LDS and issue
//     ONE float atomic each: 19 consecutive lanes cover one 128-byte accumulator row = one request.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <new>
#include <cstring>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <type_traits>

using namespace smesh;

// fusion_pair.hip: k_fuse_tri<CT, KIND, EXACT, 2> for the class-count slot `tri_ct` chosen below
void smesh_launch_fuse_tri_multi(int kind, int tri_ct, int nviews, dim3 grid, hipStream_t st, const TriFuseArgs& t, const TriViews<8>& tv);

namespace {

#include "fuse_tri.inc.hpp"


// ------------------------------------------------------------------------------------------------
// layout normalisation (Fusion.h:26-40): arbitrary dtype/strides -> contiguous uint32[N] / float[N*C]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_gather_idx(const T* __restrict__ in, int64_t s0, int64_t s1, uint32_t* __restrict__ out,
                             uint64_t N, uint32_t H) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint64_t x = i / H, y = i - x * H;
  out[i] = (uint32_t)in[x * s0 + y * s1];  // the reference casts to uint32 (Fusion.h:45): -1 -> 0xFFFFFFFF
}

__global__ void k_gather_f32_2d(const float* __restrict__ in, int64_t s0, int64_t s1, float* __restrict__ out,
                                uint64_t N, uint32_t H) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint64_t x = i / H, y = i - x * H;
  out[i] = in[x * s0 + y * s1];
}

// One workgroup = 256 consecutive pixels of the (W,H) output order: the input offset of each is computed once (thread t: pixel t), the
// 256 C output floats are then written 256 at a time, a thread's next (pixel, class) following from the previous one by adding
// 256 / C and 256 % C.  The usual strided input is a C-contiguous (H,W,C) array seen as (W,H,C) -- what a network hands over --
// whose pixels are contiguous runs of C floats: consecutive lanes read consecutive floats.  (First version: two 64-bit divisions
// per float.)
__global__ __launch_bounds__(256) void k_gather_probs(const float* __restrict__ in, int64_t s0, int64_t s1, int64_t s2,
                                                      float* __restrict__ out, uint64_t N, uint32_t H, uint32_t C) {
  __shared__ int64_t s_off[256];
  const uint32_t t = threadIdx.x;
  const uint64_t pix0 = (uint64_t)blockIdx.x * 256u;
  const uint32_t npix = (uint32_t)min((uint64_t)256u, N - pix0);
  if (t < npix) {
    const uint64_t pix = pix0 + t, x = pix / H, y = pix - x * H;
    s_off[t] = (int64_t)x * s0 + (int64_t)y * s1;
  }
  __syncthreads();
  const uint32_t q = 256u / C, r = 256u % C;
  uint32_t pl = t / C, c = t % C;
  float* __restrict__ o = out + pix0 * C;
  const uint32_t total = npix * C;
  for (uint32_t e = t; e < total; e += 256u) {
    o[e] = in[s_off[pl] + (int64_t)c * s2];
    pl += q; c += r;
    if (c >= C) { c -= C; pl++; }
  }
}

// The usual strided case -- a C-contiguous (H,W,C) tensor seen as (W,H,C): s0 == C, s2 == 1 -- as a tiled transpose of class vectors
// through LDS: a workgroup takes 32 x 8 pixels, reads them in the INPUT's order (eight runs of 32 C contiguous floats) and writes
// them in the OUTPUT's (32 runs of 8 C contiguous floats); the kernel above reads every class vector from a row of its own.  C <= 40
// (LDS: 256 C floats); render + add per cfg2 view with such a tensor: 0.191 -> 0.173 ms (0.102 with a dense one).
constexpr int kGTX = 32, kGTY = 8, kGatherTiledMaxC = 40;
__global__ __launch_bounds__(256) void k_gather_probs_hwc(const float* __restrict__ in, int64_t s1, float* __restrict__ out,
                                                          uint32_t W, uint32_t H, uint32_t C) {
  extern __shared__ float s_tile[];               // [xl][yl][c]: the output's order inside the tile
  const uint32_t t = threadIdx.x;
  const uint32_t x0 = blockIdx.x * kGTX, y0 = blockIdx.y * kGTY;
  const uint32_t nx = min((uint32_t)kGTX, W - x0), ny = min((uint32_t)kGTY, H - y0);
  // load: row yl of the tile = nx * C contiguous floats of the input
  {
    const uint32_t row = nx * C, q = 256u / C, r = 256u % C;
    const float* __restrict__ src = in + (int64_t)y0 * s1 + (int64_t)x0 * C;
    uint32_t xl = t / C, c = t % C;
    for (uint32_t j = t; j < row; j += 256u) {          // the same (xl, c) in every row of the tile: all rows' loads in flight together
      float v[kGTY];
#pragma unroll
      for (int yl = 0; yl < kGTY; yl++) v[yl] = (uint32_t)yl < ny ? src[(int64_t)yl * s1 + j] : 0.0f;
#pragma unroll
      for (int yl = 0; yl < kGTY; yl++) s_tile[(xl * kGTY + yl) * C + c] = v[yl];
      xl += q; c += r;
      if (c >= C) { c -= C; xl++; }
    }
  }
  __syncthreads();
  // store: column xl of the tile = ny * C contiguous floats of the output
  {
    const uint32_t seg = ny * C, q = 256u / seg, r = 256u % seg;
    uint32_t xl = t / seg, k = t % seg;
    const uint32_t total = nx * seg;
    for (uint32_t e = t; e < total; e += 256u) {
      out[((uint64_t)(x0 + xl) * H + y0) * C + k] = s_tile[xl * kGTY * C + k];
      xl += q; k += r;
      if (k >= seg) { k -= seg; xl++; }
    }
  }
}

// k_gather_probs, or its tiled form when the input is a dense (H,W,C) tensor seen as (W,H,C)
void launch_gather_probs(const float* d_probs, const int64_t ps[3], float* out, uint64_t W, uint64_t H, uint32_t C, hipStream_t st) {
  const uint64_t N = W * H;
  if (ps[2] == 1 && ps[0] == (int64_t)C && ps[1] >= (int64_t)(W * C) && C <= (uint32_t)kGatherTiledMaxC)
    hipLaunchKernelGGL(k_gather_probs_hwc, dim3((uint32_t)div_up(W, kGTX), (uint32_t)div_up(H, kGTY)), dim3(256), 256 * C * sizeof(float), st,
                       d_probs, ps[1], out, (uint32_t)W, (uint32_t)H, C);
  else
    hipLaunchKernelGGL(k_gather_probs, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, d_probs, ps[0], ps[1], ps[2], out, N, (uint32_t)H, C);
}

// Opaque to the optimiser: the value must sit in VGPRs here, so the load that produced it cannot be sunk
// into a later conditional block (which would serialise the strip's loads one s_waitcnt at a time).
__device__ __forceinline__ void pin(float4& v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

struct ScatterArgs {
  const uint32_t* idx;
  const float* probs;
  const float* weights;   // may be null
  const float* pw;        // per-pixel weight image from k_pixel_weights; null = every weight is 1
  uint32_t* count;        // per-view histogram; null when images_equal_weight == 0
  float* acc;             // [P][S]
  uint64_t N;
  uint32_t P;
  uint32_t C;
  uint32_t S;             // accumulator row stride in floats
  uint32_t W, H;
  uint32_t strips_y;      // strips per image column
  uint32_t nstrips;
  uint32_t strips_per_xcd;  // ceil(nstrips / 8)
  uint32_t strips_per_wave; // scatter kernel: contiguous strips walked by one persistent wave
  int vec_ok;             // every full strip's column segments start 16-byte aligned
  int dbg;                // development ablation switches (SMESH_DBG): 2 = no global atomics
  float iew;
};

__device__ __forceinline__ float pixel_weight(const ScatterArgs& a, uint32_t v, float wt) {
  // Mesh.h:100-103, evaluated in float32 in the reference's order
  float image_weight = 1.0f;
  if (a.count) image_weight = 1.0f / ((float)a.count[v]);
  const float pixel_w = 1.0f;
  const float image_pixel_weight = a.iew * image_weight + (1 - a.iew) * pixel_w;
  return image_pixel_weight * wt;
}

// ------------------------------------------------------------------------------------------------
// Strip machinery shared by the histogram and the scatter-add.
//
// A strip is 4 columns x 16 rows = one wave; lane l owns pixel (cx = l / 16, ty = l % 16).  Consecutive
// strip ids walk down an image column (their probs segments abut in memory); ids are dealt to the 8 XCDs
// in contiguous ranges so that neighbouring strips share an L2.
// ------------------------------------------------------------------------------------------------
#include "strip.inc.hpp"

struct StripGeom {
  uint32_t x0, y0;
  int nx, ny;
  bool valid;
};

__device__ __forceinline__ StripGeom strip_geom(const ScatterArgs& a) {
  StripGeom g;
  // block b runs on XCD b % 8 (observed, MI355X_MICROARCH.md); only speed depends on it
  const uint32_t b = blockIdx.x;
  const uint32_t L = (b & 7u) * a.strips_per_xcd + (b >> 3);
  g.valid = L < a.nstrips;
  const uint32_t bx = L / a.strips_y, by = L - bx * a.strips_y;
  g.x0 = bx * kSX;
  g.y0 = by * kTY;
  g.nx = g.valid ? min((int)(a.W - g.x0), kSX) : 0;
  g.ny = g.valid ? min((int)(a.H - g.y0), kTY) : 0;
  return g;
}

// ------------------------------------------------------------------------------------------------
// F1: per-view histogram count[v] = #pixels with index v (Mesh.h:90-93): one atomic per group.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_hist_strip(ScatterArgs a) {
  __shared__ StripLists L;
  const StripGeom g = strip_geom(a);
  if (!g.valid) return;
  const int l = threadIdx.x;
  const int cx = l / kTY, ty = l - cx * kTY;
  const bool in = cx < g.nx && ty < g.ny;
  const uint32_t v = in ? a.idx[(uint64_t)(g.x0 + cx) * a.H + g.y0 + ty] : 0xFFFFFFFFu;
  const StripRuns r = build_strip(L, v, a.P, l);
  uint32_t n = 0;
  if (r.root) {
    int q = l;
    for (int hop = 0; hop < kSX && q != kNone; hop++) {
      n += L.slen[q];
      q = L.child[q];
    }
  }
  if (SMESH_ABL(a.dbg) & 2) return;
  // (sorting the groups by primitive id first, as the scatter kernel does, was measured slower here:
  // 19.1 vs 15.4 us -- the network costs more than the better-coalesced 4-byte atomics save)
  if (r.root) atomicAdd(&a.count[v], n);
}

// Per-pixel weight image (Mesh.h:100-103) in pixel order: w = (iew / count[idx] + (1 - iew)) * weight.
// Doing this gather in its own pass lets the scatter kernel prefetch a strip with independent loads only.
__global__ void k_pixel_weights(ScatterArgs a, float* __restrict__ pw) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N) return;
  const uint32_t v = a.idx[i];
  float w = 0.0f;
  if (v < a.P) w = pixel_weight(a, v, a.weights ? a.weights[i] : 1.0f);
  pw[i] = w;
}

// Zero only the touched histogram entries (cheaper than a memset when P >> N).
__global__ void k_hist_clear(const uint32_t* __restrict__ idx, uint32_t* __restrict__ count, uint64_t N, uint32_t P) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint32_t v = idx[i];
  if (v < P) count[v] = 0u;
}

// ------------------------------------------------------------------------------------------------
// F2: segmented scatter-add (Mesh.h:94-106).  Persistent waves: each wave walks a contiguous range of
// 4 x 16 strips.  The float atomics are acknowledged memory-side microseconds after they are issued, and
// a wave that ends (or waits on a later load) right after issuing them holds its slot for that whole
// drain -- measured, that made the atomics' time ADD to the streaming time instead of hiding under it.
// So the loop is software-pipelined around the in-order vmcnt counter:
//     park strip s (prefetched registers -> LDS) | issue loads for strip s+1 | compute strip s |
//     wait for the s+1 loads (long since landed)  | issue the atomics of strip s, do not wait
// Nothing issued after the atomics is waited on before the next iteration has done a full strip of
// LDS/VALU work, so the drain overlaps compute.
// ------------------------------------------------------------------------------------------------
// Lane i reads lane i+D of its 16-lane DPP row (0 beyond the row): `row_shl:D`, bound_ctrl.  Columns of a
// strip are exactly one DPP row, so "the pixel D rows further down" costs a VALU operand modifier, not LDS.
template <int D>
__device__ __forceinline__ float row_down(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x100 + D, 0xF, 0xF, true));
}
template <int D>
__device__ __forceinline__ int row_down(int x) {
  return __builtin_amdgcn_update_dpp(0, x, 0x100 + D, 0xF, 0xF, true);
}

template <int CT>
struct Chunk {  // classes held in registers at a time
  static constexpr int value = (CT > 0 && CT <= 32) ? CT : 16;
};

template <int CT>
struct PrefetchVecs {  // float4 registers holding the next strip's probs (0 = no register prefetch)
  static constexpr int value = (CT > 0 && CT <= 40) ? (CT * 4 * kSX + kWave - 1) / kWave : 0;
};

__device__ __forceinline__ void pin(f4& v) { asm volatile("" : "+v"(v)); }
template <bool NT>
__device__ __forceinline__ f4 load_stream(const f4* p) {
  // NT: streamed-once data should not displace the accumulator rows from the memory-side cache
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(uint32_t& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ StripGeom strip_at(const ScatterArgs& a, uint32_t s) {
  StripGeom g;
  g.valid = s < a.nstrips;
  const uint32_t bx = s / a.strips_y, by = s - bx * a.strips_y;
  g.x0 = bx * kSX;
  g.y0 = by * kTY;
  g.nx = g.valid ? min((int)(a.W - g.x0), kSX) : 0;
  g.ny = g.valid ? min((int)(a.H - g.y0), kTY) : 0;
  return g;
}

// Flush order of one strip: its groups sorted by primitive id.
struct FlushLists {
  uint32_t sprim[kWave];    // primitive at sorted position s
  uint8_t sorder[kWave];    // root lane (= LDS row holding the group total) at sorted position s
};

// Bitonic sort of the first N lanes' keys (N = 16, 32 or 64; other lanes unchanged garbage).
template <int N>
__device__ __forceinline__ uint32_t wave_sort(uint32_t key, int l) {
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t other = (uint32_t)__shfl_xor((int)key, j);
      const bool up = (l & k) == 0;          // ascending block
      const bool lower = (l & j) == 0;       // the lower lane of a pair keeps the smaller key when ascending
      const uint32_t lo = min(key, other), hi = max(key, other);
      key = (up == lower) ? lo : hi;
    }
  }
  return key;
}

template <int CT, int KIND, bool NT>
__global__ __launch_bounds__(kWave) void k_scatter_strip(ScatterArgs a) {
  constexpr int KV = PrefetchVecs<CT>::value;
  constexpr int CH = Chunk<CT>::value;
  constexpr bool ONE_CHUNK = CT > 0 && CT <= 32;   // the whole class vector of a pixel lives in registers
  const int C = CT > 0 ? CT : (int)a.C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int pfloats = (kWave * C + 3) & ~3;
  float* sp = reinterpret_cast<float*>(smem);  // [64*C] probs strip: pixel l at sp[l*C ..)
  StripLists& L = *reinterpret_cast<StripLists*>(smem + (size_t)pfloats * 4);
  FlushLists& F = *reinterpret_cast<FlushLists*>(smem + (size_t)pfloats * 4 + ((sizeof(StripLists) + 15) & ~(size_t)15));

  // contiguous strip range of this wave; ranges are dealt to the XCDs in contiguous blocks
  // (block b runs on XCD b % 8 -- observed, MI355X_MICROARCH.md; only speed depends on it)
  const uint32_t b = blockIdx.x;
  const uint32_t chunk = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
  const uint32_t s_begin = chunk * a.strips_per_wave;
  const uint32_t s_end = min(s_begin + a.strips_per_wave, a.nstrips);
  if (s_begin >= s_end) return;

  const int l = threadIdx.x;
  const int cx = l / kTY, ty = l - cx * kTY;
  const int seg_vecs = 4 * C;           // float4 per column segment (16 pixels)
  const int nvec = kSX * seg_vecs;
  const f4* __restrict__ base4 = reinterpret_cast<const f4*>(a.probs);

  // Strip geometry is carried incrementally (one division per wave, none per strip): strip (bx, by) covers
  // columns 4*bx.., rows 16*by..; the next strip is one step down the image column.
  struct Pos { uint32_t bx, by; };
  auto geom = [&](const Pos& p) {
    StripGeom g;
    g.valid = true;
    g.x0 = p.bx * kSX;
    g.y0 = p.by * kTY;
    g.nx = min((int)(a.W - g.x0), kSX);
    g.ny = min((int)(a.H - g.y0), kTY);
    return g;
  };
  auto advance = [&](Pos p) {
    if (++p.by == a.strips_y) { p.by = 0; ++p.bx; }
    return p;
  };
  auto is_fast = [&](const StripGeom& g) { return g.nx == kSX && g.ny == kTY && a.vec_ok; };
  // per-lane float4 offsets of a full strip relative to its first float4 (vec_ok: H*C is a multiple of 4)
  uint32_t voff[KV > 0 ? KV : 1];
  if constexpr (KV > 0) {
    const uint32_t col_vecs = (uint32_t)(((uint64_t)a.H * (uint32_t)C) >> 2);
#pragma unroll
    for (int k = 0; k < KV; k++) {
      const int q = min(l + k * kWave, nvec - 1);   // clamped: the surplus lanes of the last vector re-read it
      const int seg = q / seg_vecs;
      voff[k] = (uint32_t)seg * col_vecs + (uint32_t)(q - seg * seg_vecs);
    }
  }
  auto strip_vec0 = [&](const StripGeom& g) -> uint64_t { return (((uint64_t)g.x0 * a.H + g.y0) * (uint64_t)C) >> 2; };
  auto src_index = [&](const StripGeom& g, int q) -> uint64_t {   // generic (non-prefetched) path
    const int seg = q / seg_vecs;
    const int within = q - seg * seg_vecs;
    return (((uint64_t)(g.x0 + seg) * a.H + g.y0) * (uint64_t)C) / 4 + within;
  };
  // this lane's pixel, clamped into the image so that the per-pixel loads are unconditional
  auto pixel_of = [&](const StripGeom& g) -> uint64_t {
    return (uint64_t)(g.x0 + min(cx, g.nx - 1)) * a.H + g.y0 + min(ty, g.ny - 1);
  };

  Pos pos;
  pos.bx = s_begin / a.strips_y;
  pos.by = s_begin - pos.bx * a.strips_y;

  // ---- prologue: load the first strip -----------------------------------------------------------
  // All prefetch loads are unconditional and land directly in the loop-carried registers: a load inside
  // a branch gets copied at the join, and that copy would wait for it (and for every older atomic).
  f4 r[KV > 0 ? KV : 1];
  {
    const StripGeom g0 = geom(pos);
    if constexpr (KV > 0) {
      // a partial (edge) strip is parked with dword loads instead; prefetch a harmless in-bounds address
      const f4* p0 = base4 + (is_fast(g0) ? strip_vec0(g0) : 0);
#pragma unroll
      for (int k = 0; k < KV; k++) r[k] = load_stream<NT>(p0 + (is_fast(g0) ? voff[k] : 0u));
    }
    (void)g0;
  }
  uint32_t v_next = a.idx[pixel_of(geom(pos))];
  float pw_next = a.pw ? a.pw[pixel_of(geom(pos))] : 1.0f;
  // wait for the prologue loads here, so that inside the loop the only pending memory operations at the
  // loop head are the previous strip's atomics (which no register depends on)
  if constexpr (KV > 0) {
#pragma unroll
    for (int k = 0; k < KV; k++) pin(r[k]);
  }
  pin(v_next);
  pin(pw_next);

  // flush ownership: lane -> (row slot, class) with 64 / C rows per pass; no division inside the loop
  const int rows_per_pass = C <= kWave ? kWave / C : 0;
  const int f_slot = rows_per_pass ? l / C : 0;
  const int f_c = rows_per_pass ? l - f_slot * C : 0;
  const bool f_active = rows_per_pass && f_slot < rows_per_pass;

  for (uint32_t s = s_begin; s < s_end; s++) {
    const StripGeom g = geom(pos);
    const uint32_t v_raw = v_next;
    const float pw_raw = pw_next;

    // ---- 1. park this strip's probs in LDS ------------------------------------------------------
    if (KV > 0 && is_fast(g)) {
      f4* sp4 = reinterpret_cast<f4*>(sp);
#pragma unroll
      for (int k = 0; k < (KV > 0 ? KV : 1); k++) {
        const int q = l + k * kWave;
        if (q < nvec) sp4[q] = r[k];
      }
    } else if (is_fast(g)) {
      f4* sp4 = reinterpret_cast<f4*>(sp);
      for (int base = 0; base < nvec; base += 8 * kWave) {
        f4 t8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int q = base + l + k * kWave;
          t8[k] = load_stream<NT>(base4 + src_index(g, q < nvec ? q : nvec - 1));
        }
#pragma unroll
        for (int k = 0; k < 8; k++) pin(t8[k]);
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int q = base + l + k * kWave;
          if (q < nvec) sp4[q] = t8[k];
        }
      }
    } else {
      // edge strips / unaligned images: dword loads, zero fill outside the image
      for (int q = l; q < kWave * C; q += kWave) {
        const int px = q / C, c = q - px * C;
        const int pcx = px / kTY, pty = px - pcx * kTY;
        float val = 0.0f;
        if (pcx < g.nx && pty < g.ny) val = a.probs[((uint64_t)(g.x0 + pcx) * a.H + g.y0 + pty) * C + c];
        sp[q] = val;
      }
    }

    // ---- 2. issue the loads of the next strip (consumed one iteration later) --------------------
    {
      const Pos pos1 = (s + 1 < s_end) ? advance(pos) : pos;
      const StripGeom g1 = geom(pos1);
      if constexpr (KV > 0) {
        const bool f1 = is_fast(g1);
        const f4* p1 = base4 + (f1 ? strip_vec0(g1) : 0);
#pragma unroll
        for (int k = 0; k < KV; k++) r[k] = load_stream<NT>(p1 + (f1 ? voff[k] : 0u));
      }
      v_next = a.idx[pixel_of(g1)];
      pw_next = a.pw ? a.pw[pixel_of(g1)] : 1.0f;
      pos = pos1;
    }

    // ---- 3. compute: runs, links, groups (LDS + VALU only) ---------------------------------------
    const bool in = cx < g.nx && ty < g.ny;
    const uint32_t v = in ? v_raw : 0xFFFFFFFFu;
    float w = (in && v < a.P) ? pw_raw : 0.0f;          // Mesh.h:95,100-103 (k_pixel_weights)
    const StripRuns rr = build_strip(L, v, a.P, l);     // wave syncs inside: the probs strip is complete in LDS
    // the per-view histogram was consumed by k_pixel_weights: restore its all-zero state on the way
    // (one plain store per run; replaces a memset launch per view)
    if (a.count && rr.head && v < a.P) a.count[v] = 0u;

    // ---- segmented reduction, all in registers -------------------------------------------------------
    // (a) down the column: suffix scan over the 16-lane DPP row folds every run into its head lane;
    // (b) along the chain: two pointer-doubling hops fold the <= 4 linked runs into the chain's root.
    float* row = sp + l * C;
    const int hid = rr.hl + 1;  // 0 is reserved for "no lane"
    float samef[4];
    samef[0] = (row_down<1>(hid) == hid) ? 1.0f : 0.0f;
    samef[1] = (row_down<2>(hid) == hid) ? 1.0f : 0.0f;
    samef[2] = (row_down<4>(hid) == hid) ? 1.0f : 0.0f;
    samef[3] = (row_down<8>(hid) == hid) ? 1.0f : 0.0f;
    const bool long_runs = __ballot(samef[2] != 0.0f) != 0ull;    // any run longer than 4 pixels in this strip?
    const int child1 = rr.head ? (int)L.child[l] : kNone;
    const bool has1 = child1 != kNone;
    const int c2 = has1 ? (int)L.child[child1] : kNone;
    const bool has2 = c2 != kNone;
    const bool any1 = __ballot(has1) != 0ull, any2 = __ballot(has2) != 0ull;
    auto fold = [&](float mine, float other, float sf) -> float {
      // Mul contributions can be -inf: select instead of multiplying by 0
      if (KIND == SMESH_AGG_MUL) return mine + (sf != 0.0f ? other : 0.0f);
      return fmaf(other, sf, mine);
    };
    auto reduce_chunk = [&](float (&val)[CH], int c0) {
#pragma unroll
      for (int k = 0; k < CH; k++) val[k] = fold(val[k], row_down<1>(val[k]), samef[0]);
#pragma unroll
      for (int k = 0; k < CH; k++) val[k] = fold(val[k], row_down<2>(val[k]), samef[1]);
      if (long_runs) {
#pragma unroll
        for (int k = 0; k < CH; k++) val[k] = fold(val[k], row_down<4>(val[k]), samef[2]);
#pragma unroll
        for (int k = 0; k < CH; k++) val[k] = fold(val[k], row_down<8>(val[k]), samef[3]);
      }
      if (any1) {
        // hop 1: every head adds its child's run total; hop 2: adds its grandchild's (which by then holds
        // grandchild + great-grandchild), so chains of up to 4 runs end up in their root
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const float o = __shfl(val[k], has1 ? child1 : l);
          val[k] += has1 ? o : 0.0f;
        }
        if (any2) {
#pragma unroll
          for (int k = 0; k < CH; k++) {
            const float o = __shfl(val[k], has2 ? c2 : l);
            val[k] += has2 ? o : 0.0f;
          }
        }
      }
      if (rr.root) {   // park the group total in the root pixel's own LDS row
#pragma unroll
        for (int k = 0; k < CH; k++)
          if (c0 + k < C) row[c0 + k] = val[k];
      }
    };
    if constexpr (ONE_CHUNK) {
      // read the pixel's class vector once: don't-care test on the float32 sequential class sum (Mesh.h:98),
      // then the aggregator's input map in place
      float val[CH];
      float sum = 0.0f;
      int amax = 0;
#pragma unroll
      for (int k = 0; k < CH; k++) {
        val[k] = row[k];
        sum = sum + val[k];
      }
      if (KIND == SMESH_AGG_SUMMAX) {
        float best = val[0];
#pragma unroll
        for (int k = 1; k < CH; k++)
          if (val[k] > best) { best = val[k]; amax = k; }   // first max (Fusion.cu:53)
      }
      if (!(sum > 0.5f)) w = 0.0f;
      const bool live = w != 0.0f;
#pragma unroll
      for (int k = 0; k < CH; k++) {
        float x = 0.0f;
        if (KIND == SMESH_AGG_SUMMAX) x = (live && k == amax) ? val[k] * w : 0.0f;
        else x = live ? contribution<KIND>(val[k], w) : 0.0f;
        val[k] = x;
      }
      reduce_chunk(val, 0);
    } else {
      int amax = 0;
      {
        float sum = 0.0f;
        float best = row[0];
        for (int c = 0; c < C; c++) {
          const float p = row[c];
          sum = sum + p;
          if (KIND == SMESH_AGG_SUMMAX && p > best) { best = p; amax = c; }
        }
        if (!(sum > 0.5f)) w = 0.0f;
      }
      for (int c0 = 0; c0 < C; c0 += CH) {
        float val[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
          const int c = c0 + k;
          float x = 0.0f;
          if (c < C && w != 0.0f) {
            const float p = row[c];
            if (KIND == SMESH_AGG_SUMMAX) x = (c == amax) ? p * w : 0.0f;
            else x = contribution<KIND>(p, w);
          }
          val[k] = x;
        }
        reduce_chunk(val, c0);
      }
    }

    // ---- 5. order the strip's groups by primitive id (bitonic network, key = prim << 6 | root lane):
    // neighbouring primitives then sit on neighbouring lanes and share cache lines in one atomic instruction
    {
      const bool sortable = a.P <= (1u << 26) && !(SMESH_ABL(a.dbg) & 1);
      if (rr.root) F.sprim[rr.gidx] = (sortable ? (v << 6) : ((uint32_t)rr.gidx << 6)) | (uint32_t)l;   // compact the keys
      wave_sync();
      uint32_t key = l < rr.G ? F.sprim[l] : 0xFFFFFFFFu;
      if (rr.G <= 16) key = wave_sort<16>(key, l);
      else if (rr.G <= 32) key = wave_sort<32>(key, l);
      else key = wave_sort<64>(key, l);
      wave_sync();
      if (l < rr.G) {
        const int lane = (int)(key & 63u);
        F.sorder[l] = (uint8_t)lane;
        F.sprim[l] = L.sv[lane];
      }
      wave_sync();
      // the next strip's loads have had a whole compute phase to land: wait for them NOW, before the
      // atomics go out, so that nothing younger than the atomics is ever waited on
      if constexpr (KV > 0) {
#pragma unroll
        for (int k = 0; k < KV; k++) pin(r[k]);
      }
      pin(v_next);
      pin(pw_next);
      if (!(SMESH_ABL(a.dbg) & 2)) {
        if (rows_per_pass) {
          // lane -> (row slot, class): 64 / C sorted groups per pass, C consecutive lanes per accumulator row
          if (f_active) {
            for (int si = f_slot; si < rr.G; si += rows_per_pass) {
              const float x = sp[(int)F.sorder[si] * C + f_c];
              if (x != 0.0f) unsafeAtomicAdd(&a.acc[(uint64_t)F.sprim[si] * a.S + f_c], x);
            }
          }
        } else {
          const int total = rr.G * C;
          for (int e = l; e < total; e += kWave) {
            const int si = e / C;
            const int c = e - si * C;
            const float x = sp[(int)F.sorder[si] * C + c];
            if (x != 0.0f) unsafeAtomicAdd(&a.acc[(uint64_t)F.sprim[si] * a.S + c], x);
          }
        }
      }
    }
    wave_sync();  // the next iteration overwrites the LDS strip
  }
}

// ------------------------------------------------------------------------------------------------
// Triangle-order fusion for ANY class count (run-time C).  Same ownership scheme as k_fuse_tri, but a row is
// split over a GROUP of G adjacent lanes (G = 1, 2, 4 .. 64: the smallest power of two with C / G <= kSliceAny), each
// lane holding a contiguous slice of at most kSliceAny classes of the accumulator row and of the pixel's class vector
// in registers: one pass over the probabilities whatever C is.  The don't-care test needs the float32 row sum in
// class order (Mesh.h:98, tt::sum): the running sum (and the running arg-max for Summax) is handed from lane to
// lane through the group, each lane adding its own classes one by one, so the order of operations -- and the
// result -- is the single-threaded reference's.  Loads are 16 bytes wide at 4-byte alignment (rows are only
// float-aligned).  Big triangles: tail blocks, chunked over kSlice classes (fuse_big_triangles_any).
// ------------------------------------------------------------------------------------------------
constexpr int kSlice = 40;      // texel kernel (C <= 40) and the class chunks of the big-triangle waves
constexpr int kSliceAny = 32;   // k_fuse_tri_any: classes per lane (40 cost 170 VGPRs = 2 waves per SIMD)
constexpr uint32_t kSkipPixel = 0x7FC00001u;   // NaN payload in `pw`: pixel dropped by the don't-care test


// Loads n <= N floats at src into dst; 16-byte loads as long as four floats remain.
template <int N>
__device__ __forceinline__ void load_slice(const float* __restrict__ src, int n, float (&dst)[N]) {
#pragma unroll
  for (int j = 0; j < N; j += 4) {
    if (j + 4 <= n) {
      const fvec4 q = *reinterpret_cast<const fvec4_a4*>(src + j);
      dst[j] = q.x; dst[j + 1] = q.y; dst[j + 2] = q.z; dst[j + 3] = q.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++) dst[j + t] = (j + t < n) ? src[j + t] : 0.0f;
    }
  }
}

template <int N>
__device__ __forceinline__ void store_slice(float* __restrict__ dst, int n, const float (&src)[N]) {
#pragma unroll
  for (int j = 0; j < N; j += 4) {
    if (j + 4 <= n) {
      fvec4 q;
      q.x = src[j]; q.y = src[j + 1]; q.z = src[j + 2]; q.w = src[j + 3];
      *reinterpret_cast<fvec4_a4*>(dst + j) = q;
    } else {
#pragma unroll
      for (int t = 0; t < 4; t++) if (j + t < n) dst[j + t] = src[j + t];
    }
  }
}

// acc slice += contribution of one pixel's class slice
template <int KIND, int N>
__device__ __forceinline__ void accumulate_slice(float (&accr)[N], const float (&p)[N], int cw, float w, int am_local) {
  if (KIND == SMESH_AGG_SUMMAX) {
#pragma unroll
    for (int j = 0; j < N; j++) accr[j] = (j == am_local) ? accr[j] + p[j] * w : accr[j];   // select, not an indexed update
  } else {
#pragma unroll
    for (int j = 0; j < N; j++) if (j < cw) accr[j] = accr[j] + contribution<KIND>(p[j], w);
  }
}

// Big triangles for any C: one wave per queued triangle, lanes over its pixels.  First sweep: per pixel row sum /
// arg-max / weight into the per-view scratch images (each lane re-reads only what it wrote itself); then one sweep
// per kSlice-class chunk with per-lane partial sums and a butterfly over the wave.
// One view of one such triangle: the pixels of primitive f inside the box [x0, x1] x [y0, y1] of view `a`.
template <int KIND>
__device__ __attribute__((noinline)) void fuse_box_any(const TriFuseArgs& a, const uint32_t f, const int x0, const int y0, const int x1, const int y1,
                                                       float* __restrict__ pw, uint32_t* __restrict__ amax) {
  const int l = threadIdx.x;
  const uint32_t C = a.C;
  const int bh = y1 - y0 + 1;
  const long long npx = (long long)(x1 - x0 + 1) * bh;
  uint32_t n = 0;
  for (long long i = l; i < npx; i += kWave) {
    const int x = x0 + (int)(i / bh), y = y0 + (int)(i % bh);
    n += a.idx[(uint64_t)x * a.H + y] == f ? 1u : 0u;
  }
  n = wave_sum_u(n);
  if (n == 0) return;
  const float w0 = a.iew * (1.0f / (float)n) + (1 - a.iew) * 1.0f;
  for (long long i = l; i < npx; i += kWave) {
    const int x = x0 + (int)(i / bh), y = y0 + (int)(i % bh);
    const uint64_t pix = (uint64_t)x * a.H + y;
    if (a.idx[pix] != f) continue;
    const float* __restrict__ pr = a.probs + pix * C;
    float sum = 0.0f, best = 0.0f;
    uint32_t am = 0;
    for (uint32_t c = 0; c < C; c++) {
      const float p = pr[c];
      sum = sum + p;
      if (KIND == SMESH_AGG_SUMMAX && (c == 0 || p > best)) { best = p; am = c; }
    }
    pw[pix] = sum > 0.5f ? w0 * (a.weights ? a.weights[pix] : 1.0f) : __uint_as_float(kSkipPixel);
    if (KIND == SMESH_AGG_SUMMAX) amax[pix] = am;
  }
  if constexpr (KIND == SMESH_AGG_MUL) {
    // Mul ("Mul state", fuse_tri.inc.hpp; as fuse_box does it for k_fuse_tri): the view's contributions are summed from zero in
    // double and meet the (hi, lo) row, re-centred on its largest finite element, once -- in chunks of kMulSlice classes
    constexpr int kMulSlice = 16;
    float m = -INFINITY;
    for (uint32_t c = (uint32_t)l; c < C; c += kWave) {
      const float h = a.acc[(uint64_t)f * C + c];
      if (h > m && h < INFINITY) m = h;
    }
    m = wave_max(m);
    const float centre = m > -INFINITY ? m : 0.0f;
    for (uint32_t c0 = 0; c0 < C; c0 += kMulSlice) {
      const int cw = (int)min((uint32_t)kMulSlice, C - c0);
      double part[kMulSlice];
#pragma unroll
      for (int j = 0; j < kMulSlice; j++) part[j] = 0.0;
      for (long long i = l; i < npx; i += kWave) {
        const int x = x0 + (int)(i / bh), y = y0 + (int)(i % bh);
        const uint64_t pix = (uint64_t)x * a.H + y;
        if (a.idx[pix] != f) continue;
        const float w = pw[pix];
        if (__float_as_uint(w) == kSkipPixel) continue;
        float p[kMulSlice];
        load_slice(a.probs + pix * C + c0, cw, p);
#pragma unroll
        for (int j = 0; j < kMulSlice; j++) if (j < cw) part[j] += (double)contribution<KIND>(p[j], w);
      }
      double mine = 0.0;
#pragma unroll
      for (int j = 0; j < kMulSlice; j++) {
        const double v = wave_sum_d(part[j]);
        if (l == j) mine = v;
      }
      if (l < cw) {   // this wave owns the row
        const uint64_t at = (uint64_t)f * C + c0 + l;
        float hi = a.acc[at], lo = a.acc_lo[at];
        mul_fold(hi, lo, centre, mine);
        a.acc[at] = hi;
        a.acc_lo[at] = lo;
      }
    }
    return;
  }
  for (uint32_t c0 = 0; c0 < C; c0 += kSlice) {
    const int cw = (int)min((uint32_t)kSlice, C - c0);
    float part[kSlice];
#pragma unroll
    for (int j = 0; j < kSlice; j++) part[j] = 0.0f;
    for (long long i = l; i < npx; i += kWave) {
      const int x = x0 + (int)(i / bh), y = y0 + (int)(i % bh);
      const uint64_t pix = (uint64_t)x * a.H + y;
      if (a.idx[pix] != f) continue;
      const float w = pw[pix];
      if (__float_as_uint(w) == kSkipPixel) continue;
      float p[kSlice];
      load_slice(a.probs + pix * C + c0, cw, p);
      const int am_local = KIND == SMESH_AGG_SUMMAX ? (int)amax[pix] - (int)c0 : 0;
      accumulate_slice<KIND>(part, p, cw, w, am_local);
    }
    float mine = 0.0f;
#pragma unroll
    for (int j = 0; j < kSlice; j++) {
      const float v = wave_sum(part[j]);
      if (l == j) mine = v;
    }
    if (l < cw) a.acc[(uint64_t)f * C + c0 + l] += mine;   // this wave owns the row: plain read-modify-write
  }
}

// The queues of the `nv` views of a launch, walked as one: a triangle that is big in ANY of the views is left to one wave for
// ALL of them (first view first, as separate launches would do it; the views in which it is small are scanned as 8 x 8 boxes), so
// that nobody else touches its row; a triangle queued by several views is taken from the queue of the first of them only
// (fuse_big_triangles in fuse_tri.inc.hpp: same rule).  View v parks its per-pixel weights in its own scratch image.
template <int KIND>
__device__ __forceinline__ void fuse_big_triangles_any(const TriFuseArgs& a, const TriViews<8>& vw, const int nv, uint32_t worker, uint32_t nworkers,
                                                       float* __restrict__ pw, uint32_t* __restrict__ amax, const uint64_t scratch_stride) {
  uint32_t len[8], total = 0u;
#pragma unroll
  for (int v = 0; v < 8; v++) { len[v] = v < nv ? min(*vw.v[v].big_len, a.big_capacity) : 0u; total += len[v]; }
  for (uint32_t q = worker; q < total; q += nworkers) {
    uint32_t fi = 0u;
    bool take = false;
    {
      uint32_t qq = q;
      bool located = false;
#pragma unroll
      for (int j = 0; j < 8; j++) {   // wave-uniform: q is
        if (!located) {
          if (qq < len[j]) {
            fi = vw.v[j].big_queue[qq];
            take = vw.v[j].frags[fi].kind == 2 && fi >= a.f_lo && fi < a.f_hi;   // (f_lo, f_hi: fusion by triangle range)
#pragma unroll
            for (int i = 0; i < 8; i++) if (i < j && vw.v[i].frags[fi].kind == 2) take = false;   // an earlier view's queue has it
            located = true;
          } else {
            qq -= len[j];
          }
        }
      }
    }
    if (!take) continue;
    const uint32_t f = a.prim_id ? a.prim_id[fi] : fi;        // primitive id = value in the index image = accumulator row
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (j >= nv) continue;
      const TriFrag rec = vw.v[j].frags[fi];
      if (rec.kind == 0) continue;
      const TriFuseArgs x = with_view(a, vw.v[j]);
      int x1, y1;
      if (rec.kind == 2) { x1 = (int)(rec.mask & 0xFFFFu); y1 = (int)((rec.mask >> 16) & 0xFFFFu); }
      else { x1 = min((int)rec.x0 + 7, (int)x.W - 1); y1 = min((int)rec.y0 + 7, (int)x.H - 1); }
      fuse_box_any<KIND>(x, f, rec.x0, rec.y0, x1, y1, pw + (uint64_t)j * scratch_stride, amax ? amax + (uint64_t)j * scratch_stride : nullptr);
    }
  }
}

// The big triangles of the any-C paths run as their own launch: inlined into the kernels below, their kSlice-wide
// partial sums set the register allocation (148 VGPRs) of waves that never execute them.
template <int KIND>
__global__ __launch_bounds__(kWave) void k_fuse_big_any(TriFuseArgs a, TriViews<8> vw, int nv, float* __restrict__ pw, uint32_t* __restrict__ amax,
                                                        uint64_t scratch_stride) {
  fuse_big_triangles_any<KIND>(a, vw, nv, blockIdx.x, gridDim.x, pw, amax, scratch_stride);
}

// Per-view state of the wave's lanes, parked in LDS so that the view loop of the multi-view kernels is a run-time loop over
// LDS reads instead of eight copies of the code (k_fuse_tri_any: every lane reads its own entries back; k_fuse_tri_wide: the
// owner lane's entries are broadcast).
struct ViewState {
  TriView view[8];
  uint32_t org[8][kWave], lo[8][kWave], hi[8][kWave];
};

// Up to eight views per launch (`nv`, a run-time count), in order: the row slices stay in registers from the first view's first
// pixel to the last view's last, one load and one store of the row for all of them -- the additions are those of one launch per
// view, in their order.  A triangle that is big in any of the views belongs to k_fuse_big_any for all of them.
template <int KIND, int G>
__global__ __launch_bounds__(kWave) void k_fuse_tri_any(TriFuseArgs a, TriViews<8> vw, int nv) {
  constexpr int TPW = kWave / G;   // triangles per wave
  __shared__ ViewState S;
  const int l = threadIdx.x;
  const uint32_t C = a.C;
  const int g = l % G;                                   // rank inside the group
  const uint64_t fi = ((uint64_t)a.blk_first + blockIdx.x) * TPW + (uint32_t)(l / G);   // position in the renderer's triangle order (blk_first: fusion by range)
  const uint64_t f = (a.prim_id && fi < a.F) ? (uint64_t)a.prim_id[fi] : fi;   // primitive id (index image value, accumulator row)
  const uint32_t SL = G == 1 ? C : (((C + G - 1) / G + 3u) & ~3u);   // classes per lane (whole float4s when the row is split)
  const uint32_t c_lo = (uint32_t)g * SL;
  const int cw = c_lo < C ? (int)min(SL, C - c_lo) : 0;
  // pass 1 per view (as k_fuse_tri; the lanes of a group do it redundantly -- same addresses, one request)
  unsigned long long win[8];
  bool big = false;
#pragma unroll
  for (int v = 0; v < 8; v++) {
    if (l == 0) S.view[v] = vw.v[v];
    win[v] = 0ull;
    uint32_t origin = 0u;
    if (v < nv) {
      TriFrag rec;
      rec.x0 = 0; rec.y0 = 0; rec.kind = 0; rec.pad = 0; rec.mask = 0ull;
      if (fi < a.F) rec = vw.v[v].frags[fi];
      origin = (uint32_t)rec.x0 | ((uint32_t)rec.y0 << 16);
      big = big || rec.kind == 2;
      unsigned long long m = rec.kind == 1 ? rec.mask : 0ull;
      if (!a.prim_id && vw.v[v].big_len[1] == 0u) { win[v] = m; m = 0ull; }   // the tile resolve has cleared the losers (k_fuse_tri)
      const uint32_t* __restrict__ idx = vw.v[v].idx;
      const uint32_t Hv = vw.v[v].H;
      while (__ballot(m != 0ull) != 0ull) {
        int k[4];
        uint32_t got[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          k[j] = -1;
          if (m) { k[j] = __ffsll((long long)m) - 1; m &= m - 1ull; }
          got[j] = idx[k[j] >= 0 ? (uint64_t)(rec.x0 + (k[j] >> 3)) * Hv + rec.y0 + (k[j] & 7) : 0];
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (k[j] >= 0 && got[j] == (uint32_t)f) win[v] |= 1ull << k[j];
      }
    }
    S.org[v][l] = origin;
  }
  unsigned long long any_win = 0ull;
#pragma unroll
  for (int v = 0; v < 8; v++) {
    if (big) win[v] = 0ull;
    any_win |= win[v];
    S.lo[v][l] = (uint32_t)win[v];
    S.hi[v][l] = (uint32_t)(win[v] >> 32);
  }
  if (__ballot(any_win != 0ull) == 0ull) return;
  wave_sync();
  float* __restrict__ row = a.acc + f * C + c_lo;
  float accr[kSliceAny];
  if (any_win) load_slice(row, cw, accr);
  else {
#pragma unroll
    for (int j = 0; j < kSliceAny; j++) accr[j] = 0.0f;
  }
  for (int v = 0; v < nv; v++) {
    const unsigned long long wv = (unsigned long long)S.lo[v][l] | ((unsigned long long)S.hi[v][l] << 32);
    if (__ballot(wv != 0ull) == 0ull) continue;   // (wave-uniform)
    const uint32_t org = S.org[v][l];
    const float* __restrict__ probs = S.view[v].probs;
    const float* __restrict__ weights = S.view[v].weights;
    const uint32_t Hv = S.view[v].H;
    const uint32_t n = (uint32_t)__popcll(wv);
    const float w0 = n ? a.iew * (1.0f / ((float)n)) + (1 - a.iew) * 1.0f : 0.0f;      // Mesh.h:100-102
    // Mul ("Mul state", fuse_tri.inc.hpp): the view's contributions are summed separately, from zero, and meet the re-centred
    // (hi, lo) row once per view, in double -- as in k_fuse_tri
    constexpr int PT = KIND == SMESH_AGG_MUL ? kSliceAny : 1;
    double part[PT];
#pragma unroll
    for (int j = 0; j < PT; j++) part[j] = 0.0;
    for (unsigned long long m = wv; __ballot(m != 0ull) != 0ull; m &= m - 1ull) {   // wave-uniform trip count: shuffles inside
      const bool have = m != 0ull;
      const int k = have ? __ffsll((long long)m) - 1 : 0;
      const uint64_t pix = have ? (uint64_t)((org & 0xFFFFu) + (uint32_t)(k >> 3)) * Hv + (org >> 16) + (uint32_t)(k & 7) : 0;
      float p[kSliceAny];
      load_slice(probs + pix * C + c_lo, have ? cw : 0, p);
      const float wt = (have && weights) ? weights[pix] : 1.0f;
      // row sum (and arg-max) in class order: the running values travel through the group's lanes
      float s = 0.0f, best = 0.0f;
      uint32_t am = 0;
#pragma unroll
      for (int ph = 0; ph < G; ph++) {
        if (ph > 0) {   // from the lane below (wave_shr:1)
          const float s_in = dpp_f<kDppWaveShr1>(0.0f, s);
          const float b_in = KIND == SMESH_AGG_SUMMAX ? dpp_f<kDppWaveShr1>(0.0f, best) : 0.0f;
          const uint32_t a_in = KIND == SMESH_AGG_SUMMAX ? dpp_u<kDppWaveShr1>(0u, am) : 0u;
          if (g == ph) { s = s_in; best = b_in; am = a_in; }
        }
        if (g == ph) {
#pragma unroll
          for (int j = 0; j < kSliceAny; j++)
            if (j < cw) {
              s = s + p[j];
              if (KIND == SMESH_AGG_SUMMAX && ((ph == 0 && j == 0) || p[j] > best)) { best = p[j]; am = c_lo + (uint32_t)j; }
            }
        }
      }
      if (G == 2) {          // the group's last lane holds the totals: quad_perm [1,1,3,3]
        s = dpp_f<0xF5>(0.0f, s);
        if (KIND == SMESH_AGG_SUMMAX) am = dpp_u<0xF5>(0u, am);
      } else if (G == 4) {   // quad_perm [3,3,3,3]
        s = dpp_f<0xFF>(0.0f, s);
        if (KIND == SMESH_AGG_SUMMAX) am = dpp_u<0xFF>(0u, am);
      } else if (G > 4) {
        const int last = (l / G) * G + (G - 1);
        s = __shfl(s, last);
        if (KIND == SMESH_AGG_SUMMAX) am = (uint32_t)__shfl((int)am, last);
      }
      if (have && s > 0.5f) {
        if constexpr (KIND == SMESH_AGG_MUL) {
#pragma unroll
          for (int j = 0; j < PT; j++) if (j < cw) part[j] += (double)contribution<KIND>(p[j], w0 * wt);
        } else {
          accumulate_slice<KIND>(accr, p, cw, w0 * wt, (int)am - (int)c_lo);
        }
      }
    }
    if constexpr (KIND == SMESH_AGG_MUL) {
      // the row's largest finite element, over the G lanes that share it (every lane takes part in the shuffles)
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < kSliceAny; j++) if (j < cw && accr[j] > mx && accr[j] < INFINITY) mx = accr[j];
#pragma unroll
      for (int o = 1; o < G; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float centre = mx > -INFINITY ? mx : 0.0f;
      if (wv != 0ull) {
        float* __restrict__ lo_row = a.acc_lo + f * C + c_lo;
        float lo[kSliceAny];
        load_slice(lo_row, cw, lo);
#pragma unroll
        for (int j = 0; j < kSliceAny; j++) if (j < cw) mul_fold(accr[j], lo[j], centre, part[j]);
        store_slice(lo_row, cw, lo);
      }
    }
  }
  if (any_win) store_slice(row, cw, accr);
}

// ------------------------------------------------------------------------------------------------
// Triangle-order fusion for WIDE rows (128 <= C <= 1024): a row no longer belongs to a lane but to the wave.
// Lane l holds classes 4 (l + 64 k) .. + 3 (k < NCH) of the current accumulator row and of the current pixel's class
// vector, so every row moves as one coalesced run of 16-byte pieces.  The wave first finds, lane = triangle, the
// visible pixels of its 64 triangles (pass 1 of k_fuse_tri); then it walks the visible triangles a few at a time
// (wave-uniform control flow, the pixel set of a triangle read from its owner lane into scalar registers): load the
// row, add its pixels in image order, store the row.  Per class the additions happen in the reference's order, so the result is still
// bit-identical.  The don't-care test needs the float32 row sum in class order (Mesh.h:98): the wave first
// tree-reduces the row and its absolute values; only if that estimate is within its own error bound of 0.5
// does it replay the additions one class at a time (never for probability rows, whose sums are ~1 or 0).
// ------------------------------------------------------------------------------------------------
// One lane's share of a row: NCH chunks of four classes, chunk k covering classes 4 (l + 64 k) .. + 3.
// (The row's tail -- C = 150 leaves two classes to lane 37 -- costs that lane two or three single-dword instructions.  Round 6 moved the
// row's last four classes with ONE 16-byte instruction instead, the lane keeping / composing what is its own: k_fuse_tri_wide 1 654 ->
// 1 872 us per cfg5 view.  Overlapping, unaligned pieces are worse than three instructions; not kept.)
template <int NCH>
__device__ __forceinline__ void load_wide(const float* __restrict__ src, uint32_t C, int l, fvec4 (&v)[NCH]) {
#pragma unroll
  for (int k = 0; k < NCH; k++) {
    const uint32_t c = 4u * ((uint32_t)l + 64u * k);
    if (c + 4u <= C) v[k] = *reinterpret_cast<const fvec4_a4*>(src + c);
    else { v[k].x = c < C ? src[c] : 0.0f; v[k].y = c + 1 < C ? src[c + 1] : 0.0f; v[k].z = c + 2 < C ? src[c + 2] : 0.0f; v[k].w = 0.0f; }
  }
}

template <int NCH>
__device__ __forceinline__ void store_wide(float* __restrict__ dst, uint32_t C, int l, const fvec4 (&v)[NCH]) {
#pragma unroll
  for (int k = 0; k < NCH; k++) {
    const uint32_t c = 4u * ((uint32_t)l + 64u * k);
    if (c + 4u <= C) *reinterpret_cast<fvec4_a4*>(dst + c) = v[k];
    else {
      if (c < C) dst[c] = v[k].x;
      if (c + 1 < C) dst[c + 1] = v[k].y;
      if (c + 2 < C) dst[c + 2] = v[k].z;
    }
  }
}

// Mesh.h:94-106 for one pixel whose class vector is spread over the wave (wave-uniform control flow).
typedef double dvec4 __attribute__((ext_vector_type(4)));
template <int KIND, int NCH, typename ACC4>   // ACC4: fvec4 (the accumulator row itself), or dvec4 (Mul: a view's partial sums)
__device__ __forceinline__ void fuse_pixel_wide(ACC4 (&ac)[NCH], const fvec4 (&p)[NCH], uint32_t C, int l, float w) {
  typedef decltype(ac[0].x + ac[0].x) acc_t;
  float ps = 0.0f, pa = 0.0f;
  bool neg = false;
#pragma unroll
  for (int k = 0; k < NCH; k++) {
    ps += (p[k].x + p[k].y) + (p[k].z + p[k].w);
    pa += (fabsf(p[k].x) + fabsf(p[k].y)) + (fabsf(p[k].z) + fabsf(p[k].w));
    neg = neg || !(fminf(fminf(p[k].x, p[k].y), fminf(p[k].z, p[k].w)) >= 0.0f);   // (a NaN counts as negative: fminf drops it, the sum keeps it)
  }
  neg = neg || !(ps >= 0.0f);
  ps = wave_sum(ps);
  // the sum of the absolute values -- the scale of the tree sum's error -- is the sum itself when no element is negative
  // (every probability row): the second wave-wide reduction only runs for rows with negative entries
  pa = __ballot(neg) != 0ull ? wave_sum(pa) : ps;
  bool counted = ps > 0.5f;
  if (!(fabsf(ps - 0.5f) > 1e-4f * (pa + 1.0f))) {
    // too close to call from a tree sum (or not finite): replay tt::sum, one class at a time
    float sq = 0.0f;
    for (uint32_t c = 0; c < C; c++) {
      const uint32_t q = c >> 2, comp = c & 3u;
      float v = 0.0f;
#pragma unroll
      for (int k = 0; k < NCH; k++)
        if ((int)(q >> 6) == k) v = comp == 0 ? p[k].x : comp == 1 ? p[k].y : comp == 2 ? p[k].z : p[k].w;
      sq = sq + __shfl(v, (int)(q & 63u));
    }
    counted = sq > 0.5f;
  }
  if (!counted) return;
  if (KIND == SMESH_AGG_SUMMAX) {
    // first maximum in class order: per lane in order, then across lanes (greater, or equal at a lower class)
    float best = -INFINITY;
    uint32_t am = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < NCH; k++) {
      const uint32_t c = 4u * ((uint32_t)l + 64u * k);
      const float v[4] = {p[k].x, p[k].y, p[k].z, p[k].w};
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (c + e < C && (am == 0xFFFFFFFFu || v[e] > best)) { best = v[e]; am = c + e; }
    }
#define SMESH_ARGMAX_STEP(CTRL, ROWS)                                                                      \
    {                                                                                                      \
      const float ob = dpp_f<CTRL, ROWS>(-INFINITY, best);                                                 \
      const uint32_t oa = dpp_u<CTRL, ROWS>(0xFFFFFFFFu, am);                                              \
      const bool take = oa != 0xFFFFFFFFu && (am == 0xFFFFFFFFu || ob > best || (ob == best && oa < am));  \
      if (take) { best = ob; am = oa; }                                                                    \
    }
    SMESH_ARGMAX_STEP(kDppRowShr1, 0xF) SMESH_ARGMAX_STEP(kDppRowShr2, 0xF) SMESH_ARGMAX_STEP(kDppRowShr4, 0xF)
    SMESH_ARGMAX_STEP(kDppRowShr8, 0xF) SMESH_ARGMAX_STEP(kDppRowBcast15, 0xA) SMESH_ARGMAX_STEP(kDppRowBcast31, 0xC)
#undef SMESH_ARGMAX_STEP
    am = (uint32_t)__builtin_amdgcn_readlane((int)am, 63);   // lane 63 has seen every lane
#pragma unroll
    for (int k = 0; k < NCH; k++) {
      const uint32_t c = 4u * ((uint32_t)l + 64u * k);
      ac[k].x = (am == c) ? ac[k].x + (acc_t)(p[k].x * w) : ac[k].x;
      ac[k].y = (am == c + 1) ? ac[k].y + (acc_t)(p[k].y * w) : ac[k].y;
      ac[k].z = (am == c + 2) ? ac[k].z + (acc_t)(p[k].z * w) : ac[k].z;
      ac[k].w = (am == c + 3) ? ac[k].w + (acc_t)(p[k].w * w) : ac[k].w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < NCH; k++) {
      ac[k].x = ac[k].x + (acc_t)contribution<KIND>(p[k].x, w);
      ac[k].y = ac[k].y + (acc_t)contribution<KIND>(p[k].y, w);
      ac[k].z = ac[k].z + (acc_t)contribution<KIND>(p[k].z, w);
      ac[k].w = ac[k].w + (acc_t)contribution<KIND>(p[k].w, w);
    }
  }
}

// A pointer read from LDS is the same in every lane: said so, it lives in scalar registers.
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return reinterpret_cast<const float*>((uint64_t)lo | ((uint64_t)hi << 32));
}

// Up to eight views of the same mesh in one launch (`nv`, a run-time count: one instance per row width), in order: a triangle's
// accumulator row -- 600 bytes each way at C = 150, two thirds of what a cfg5 view moves -- makes ONE round trip for all of them, and
// the additions happen in the order `nv` launches would have made them (a store and a load of a float32 row change nothing).  The
// per-view state of the wave's 64 triangles (box origin, mask of visible pixels) and the views' pointers are parked in LDS, so
// that the view loop is a run-time loop over broadcast reads instead of eight copies of the code.  A triangle that is big in any
// of the views belongs to k_fuse_big_any for all of them.
template <int KIND, int NCH, int B>   // B: visible triangles whose rows and next pixels are in flight together
__global__ __launch_bounds__(kWave) void k_fuse_tri_wide(TriFuseArgs a, TriViews<8> vw, int nv) {
  __shared__ ViewState S;
  const int l = threadIdx.x;
  const uint32_t C = a.C;
  uint32_t blk = blockIdx.x;
  if (a.xcd_chunk) {                   // (TriFuseArgs::xcd_chunk: runs of that many consecutive triangle blocks per XCD)
    const uint32_t sq = blk >> 3, q = sq / a.xcd_chunk;
    blk = (q * 8u + (blk & 7u)) * a.xcd_chunk + (sq - q * a.xcd_chunk);
    if (blk >= a.tri_blocks) return;   // block-uniform
  }
  const uint64_t f0 = ((uint64_t)a.blk_first + blk) * kWave;   // (blk_first: fusion by triangle range; 0 otherwise)
  const uint64_t f = f0 + l;
  const uint32_t pid = (a.prim_id && f < a.F) ? a.prim_id[f] : (uint32_t)f;   // primitive id (index image value, accumulator row)
  unsigned long long win[8];
  bool big = false;
#pragma unroll
  for (int v = 0; v < 8; v++) {
    if (l == 0) S.view[v] = vw.v[v];
    win[v] = 0ull;
    uint32_t origin = 0u;
    if (v < nv) {
      TriFrag rec;
      rec.x0 = 0; rec.y0 = 0; rec.kind = 0; rec.pad = 0; rec.mask = 0ull;
      if (f < a.F) rec = vw.v[v].frags[f];
      origin = (uint32_t)rec.x0 | ((uint32_t)rec.y0 << 16);
      big = big || rec.kind == 2;
      unsigned long long m = rec.kind == 1 ? rec.mask : 0ull;
      // pass 1, lane = triangle, normally skipped: the tile resolve has cleared the losers out of the masks (k_fuse_tri)
      if (!a.prim_id && vw.v[v].big_len[1] == 0u) { win[v] = m; m = 0ull; }
      const uint32_t* __restrict__ idx = vw.v[v].idx;
      const uint32_t Hv = vw.v[v].H;
      while (__ballot(m != 0ull) != 0ull) {
        int k[4];
        uint32_t got[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          k[j] = -1;
          if (m) { k[j] = __ffsll((long long)m) - 1; m &= m - 1ull; }
          got[j] = idx[k[j] >= 0 ? (uint64_t)(rec.x0 + (k[j] >> 3)) * Hv + rec.y0 + (k[j] & 7) : 0];
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (k[j] >= 0 && got[j] == pid) win[v] |= 1ull << k[j];
      }
    }
    S.org[v][l] = origin;
  }
  unsigned long long any_win = 0ull;
#pragma unroll
  for (int v = 0; v < 8; v++) {
    if (big) win[v] = 0ull;
    any_win |= win[v];
    S.lo[v][l] = (uint32_t)win[v];
    S.hi[v][l] = (uint32_t)(win[v] >> 32);
  }
  unsigned long long vis = __ballot(any_win != 0ull);
  if (vis == 0ull || (SMESH_ABL(a.dbg) & 1)) return;
  wave_sync();

  // From here on the wave works on a few triangles' rows at a time; a triangle's pixel sets are read from its owner lane's LDS
  // entries into scalar registers, so the control flow is wave-uniform.
  auto pix_of = [&](uint32_t o, int k, uint32_t Hv) -> uint64_t { return (uint64_t)((o & 0xFFFFu) + (uint32_t)(k >> 3)) * Hv + (o >> 16) + (uint32_t)(k & 7); };
  while (vis) {
    int t[B];
    uint32_t rowid[B];
    fvec4 ac[B][NCH];
#pragma unroll
    for (int b = 0; b < B; b++) {
      t[b] = -1; rowid[b] = 0u;
      if (vis) {
        t[b] = __ffsll((long long)vis) - 1;
        vis &= vis - 1ull;
        rowid[b] = (uint32_t)__builtin_amdgcn_readlane((int)pid, t[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < B; b++)
      if (t[b] >= 0 && !(SMESH_ABL(a.dbg) & 4)) load_wide<NCH>(a.acc + (uint64_t)rowid[b] * C, C, l, ac[b]);
    if constexpr (KIND != SMESH_AGG_MUL) {
      // Sum / Summax: every triangle of the batch is a STREAM of pixels -- its views in order, its pixels of a view in image order --
      // and a round takes the next pixel of every stream whatever view it belongs to.  (Round 4 walked the views in lock step:
      // a round loaded the pixels the B triangles have in ONE view.  At cfg5 a visible triangle has a pixel in three or four of a
      // launch's eight views, so a round had one or two of its B loads in flight and a batch took eight dependent round trips; the
      // counters put the kernel at 2.6 TB/s of scattered 600-byte rows: short of requests in flight, not of bandwidth.)  Per row the
      // additions are the same, in the same order.
      int sv[B];                       // the view a stream is in (wave-uniform, like everything below)
      unsigned long long pm[B];        // its pixels of that view that are still to come
      uint32_t org[B], Hs[B];
      float w0[B];
      const float* pr[B];
      const float* wg[B];
#pragma unroll
      for (int b = 0; b < B; b++) { sv[b] = -1; pm[b] = 0ull; org[b] = 0u; Hs[b] = 0u; w0[b] = 0.0f; pr[b] = nullptr; wg[b] = nullptr; }
      while (true) {
        bool have[B];
        bool any = false;
#pragma unroll
        for (int b = 0; b < B; b++) {
          if (t[b] >= 0) {
            while (pm[b] == 0ull && sv[b] + 1 < nv) {      // on to the stream's next view that holds pixels of the triangle
              sv[b]++;
              const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.lo[sv[b]][t[b]]);
              const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.hi[sv[b]][t[b]]);
              pm[b] = (unsigned long long)lo | ((unsigned long long)hi << 32);
              if (pm[b]) {
                org[b] = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.org[sv[b]][t[b]]);
                const uint32_t nt = (uint32_t)__popcll(pm[b]);                      // this primitive's pixels in this view (Mesh.h:90-93)
                w0[b] = a.iew * (1.0f / ((float)nt)) + (1 - a.iew) * 1.0f;           // Mesh.h:100-102
                pr[b] = uniform_ptr(S.view[sv[b]].probs);
                wg[b] = uniform_ptr(S.view[sv[b]].weights);
                Hs[b] = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.view[sv[b]].H);
              }
            }
          }
          have[b] = pm[b] != 0ull;
          any = any || have[b];
        }
        if (!any) break;
        fvec4 p[B][NCH];
        float wt[B];
#pragma unroll
        for (int b = 0; b < B; b++) {
          wt[b] = 1.0f;
          if (have[b]) {
            const uint64_t pix = pix_of(org[b], __ffsll((long long)pm[b]) - 1, Hs[b]);
            pm[b] &= pm[b] - 1ull;
            if (!(SMESH_ABL(a.dbg) & 8)) load_wide<NCH>(pr[b] + pix * C, C, l, p[b]);
            if (wg[b]) wt[b] = wg[b][pix];
          }
        }
#pragma unroll
        for (int b = 0; b < B; b++)
          if (have[b]) fuse_pixel_wide<KIND, NCH>(ac[b], p[b], C, l, w0[b] * wt[b]);     // :103
      }
    } else
    for (int v = 0; v < nv; v++) {
      const float* __restrict__ probs = S.view[v].probs;
      const float* __restrict__ weights = S.view[v].weights;
      const uint32_t Hv = S.view[v].H;
      unsigned long long pm[B];
      uint32_t org[B];
      float w0[B];
      unsigned long long left = 0ull;
      // Mul: the view's contributions are summed separately, from zero, and folded into the re-centred (hi, lo) row once per view
      constexpr int PB = KIND == SMESH_AGG_MUL ? B : 1;
      dvec4 part[PB][NCH];
      bool touched[B];
#pragma unroll
      for (int b = 0; b < PB; b++)
#pragma unroll
        for (int k = 0; k < NCH; k++) part[b][k] = dvec4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int b = 0; b < B; b++) {
        touched[b] = false;
        pm[b] = 0ull; org[b] = 0u; w0[b] = 0.0f;
        if (t[b] >= 0) {
          const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.lo[v][t[b]]);
          const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.hi[v][t[b]]);
          pm[b] = (unsigned long long)lo | ((unsigned long long)hi << 32);
          org[b] = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.org[v][t[b]]);
          const uint32_t nt = (uint32_t)__popcll(pm[b]);                                    // this primitive's pixels in this view (Mesh.h:90-93)
          if (nt) w0[b] = a.iew * (1.0f / ((float)nt)) + (1 - a.iew) * 1.0f;               // Mesh.h:100-102
          touched[b] = nt != 0u;
        }
        left |= pm[b];
      }
      while (left) {   // one pixel of each of the B triangles per round, in image order per triangle
        fvec4 p[B][NCH];
        float wt[B];
        bool have[B];
#pragma unroll
        for (int b = 0; b < B; b++) {
          have[b] = pm[b] != 0ull;
          wt[b] = 1.0f;
          if (have[b]) {
            const uint64_t pix = pix_of(org[b], __ffsll((long long)pm[b]) - 1, Hv);
            pm[b] &= pm[b] - 1ull;
            if (!(SMESH_ABL(a.dbg) & 8)) load_wide<NCH>(probs + pix * C, C, l, p[b]);
            if (weights) wt[b] = weights[pix];
          }
        }
        left = 0ull;
#pragma unroll
        for (int b = 0; b < B; b++) {
          if (have[b]) {
            if constexpr (KIND == SMESH_AGG_MUL) fuse_pixel_wide<KIND, NCH>(part[b], p[b], C, l, w0[b] * wt[b]);
            else fuse_pixel_wide<KIND, NCH>(ac[b], p[b], C, l, w0[b] * wt[b]);     // :103
          }
          left |= pm[b];
        }
      }
      if constexpr (KIND == SMESH_AGG_MUL) {
#pragma unroll
        for (int b = 0; b < B; b++) {
          if (!touched[b]) continue;   // (wave-uniform)
          float mx = -INFINITY;
#pragma unroll
          for (int k = 0; k < NCH; k++) {
            const uint32_t c = 4u * ((uint32_t)l + 64u * k);
            const float e[4] = {ac[b][k].x, ac[b][k].y, ac[b][k].z, ac[b][k].w};
#pragma unroll
            for (int q = 0; q < 4; q++) if (c + q < C && e[q] > mx && e[q] < INFINITY) mx = e[q];
          }
          mx = wave_max(mx);
          const float centre = mx > -INFINITY ? mx : 0.0f;
          fvec4 lo[NCH];
          float* __restrict__ lo_row = a.acc_lo + (uint64_t)rowid[b] * C;
          load_wide<NCH>(lo_row, C, l, lo);
#pragma unroll
          for (int k = 0; k < NCH; k++) {
            float h[4] = {ac[b][k].x, ac[b][k].y, ac[b][k].z, ac[b][k].w};
            float r[4] = {lo[k].x, lo[k].y, lo[k].z, lo[k].w};
            const double pt[4] = {part[b][k].x, part[b][k].y, part[b][k].z, part[b][k].w};
#pragma unroll
            for (int q = 0; q < 4; q++) mul_fold(h[q], r[q], centre, pt[q]);
            ac[b][k] = fvec4{h[0], h[1], h[2], h[3]};
            lo[k] = fvec4{r[0], r[1], r[2], r[3]};
          }
          store_wide<NCH>(lo_row, C, l, lo);
        }
      }
    }
#pragma unroll
    for (int b = 0; b < B; b++)
      if (t[b] >= 0 && !(SMESH_ABL(a.dbg) & 2)) store_wide<NCH>(a.acc + (uint64_t)rowid[b] * C, C, l, ac[b]);
  }
}

// ------------------------------------------------------------------------------------------------
// k_fuse_tri_wide for Sum as a walk over a PIXEL LIST (round 6).  A replay of a cfg5 launch's REAL address stream with nothing around it
// (tools/stream_bench p, profiles/r06_stream_replay_cfg5.txt: the class vectors triangle after triangle, every visible triangle's row
// read and written once) takes 8.6 ms; the kernel above took 13.1, and its development build without any row or class-vector access
// still 6.2: what it executes per pixel beside the two memory instructions -- finding the stream's next view with pixels of the
// triangle, bit scans of 64-bit masks, the pixel's address, the view's pointers read back from LDS: ~80 scalar and ~70 vector
// instructions per 600-byte row -- did not hide under the memory time.  Here the 64 lanes first write, lane = triangle and all at once,
// the pixels of their triangles -- views in order, a view's pixels in image order: the order of the additions -- into a list in LDS; then
// the wave walks the list with nothing left to decide, through a RING of K loads: entry i is consumed from slot i % K, which is
// refilled with entry i + K at once, and a slot also carries the accumulator row of the triangle that starts at its entry.  K class
// vectors (and the rows of the triangles ahead) are in flight at every moment, whatever triangle or view they belong to.
// cfg5: 1 651 -> 1 125 us per view (replay: 1 075), 504 -> 681 views/s.  Per accumulator row the same float32 additions in the same
// order as the kernel above: bit-equal raw accumulators (tests/test_gpu_parity.py).  Seven waves per SIMD (72 registers, 46 spilled
// around the records phase): 6 / 7 / 8 waves 1 200 / 1 125 / 1 231 us.  A chunk of the list holds whole triangles (at most 8 x 64
// entries each); a wave with more pixels than kWideCap walks several chunks, reloading its records for each.
// ------------------------------------------------------------------------------------------------
constexpr int kWideCap = 512;      // entries of a chunk of the list (a triangle has at most 8 views x 64 pixels = 512)
struct WideList {
  TriView view[8];
  float w0[64];            // iew / n + (1 - iew) for n = 1 .. 64 pixels of a triangle in a view (Mesh.h:90-93, 100-102)
  uint32_t px[kWideCap];   // pixel (x * H + y, < 2^29: check_camera) | view << 29
  uint16_t nl[kWideCap];   // n - 1 | the triangle's lane << 6
};                         // (3.9 KB: the LDS must not be what limits the waves per SIMD)

template <int KIND, int NCH>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_fuse_tri_wide_list(TriFuseArgs a, TriViews<8> vw, int nv) {
  static_assert(KIND != SMESH_AGG_MUL, "Mul folds a view's terms once per view: k_fuse_tri_wide");
  constexpr int K = NCH == 1 ? 4 : 2;      // class vectors in flight per wave: a ring of K slots, refilled as they are consumed
  __shared__ WideList S;
  const int l = threadIdx.x;
  const uint32_t C = a.C;
  uint32_t blk = blockIdx.x;
  if (a.xcd_chunk) {
    const uint32_t sq = blk >> 3, q = sq / a.xcd_chunk;
    blk = (q * 8u + (blk & 7u)) * a.xcd_chunk + (sq - q * a.xcd_chunk);
    if (blk >= a.tri_blocks) return;   // block-uniform
  }
  const uint64_t f0 = ((uint64_t)a.blk_first + blk) * kWave;
  const uint64_t f = f0 + l;
  const uint32_t pid = (a.prim_id && f < a.F) ? a.prim_id[f] : (uint32_t)f;
  if (l == 0) {
#pragma unroll
    for (int v = 0; v < 8; v++) S.view[v] = vw.v[v];
  }
  S.w0[l] = a.iew * (1.0f / ((float)(l + 1))) + (1 - a.iew) * 1.0f;              // Mesh.h:100-102 for l + 1 pixels
  uint32_t cnt = 0u, inc = 0u, exc = 0u;   // this triangle's entries (its visible pixels over the launch's views), and their running sums over the lanes
  unsigned long long vis = 0ull;
  uint32_t base = 0u;                // entries of the chunks before this one
  int s = 0;                         // first lane (triangle) of this chunk
  for (bool first = true;; first = false) {
    // The triangles' records of every view -> the masks of their visible pixels.  Per CHUNK (a wave almost always has one): kept across the
    // walk below, the eight masks and origins would cost the kernel a wave per SIMD.
    unsigned long long win[8];
    uint32_t org[8];
    bool big = false;
#pragma unroll
    for (int v = 0; v < 8; v++) {
      win[v] = 0ull;
      org[v] = 0u;
      if (v < nv) {
        TriFrag rec;
        rec.x0 = 0; rec.y0 = 0; rec.kind = 0; rec.pad = 0; rec.mask = 0ull;
        if (f < a.F) rec = vw.v[v].frags[f];
        org[v] = (uint32_t)rec.x0 | ((uint32_t)rec.y0 << 16);
        big = big || rec.kind == 2;
        unsigned long long m = rec.kind == 1 ? rec.mask : 0ull;
        // pass 1, lane = triangle, normally skipped: the tile resolve has cleared the losers out of the masks (k_fuse_tri)
        if (!a.prim_id && vw.v[v].big_len[1] == 0u) { win[v] = m; m = 0ull; }
        const uint32_t* __restrict__ idx = vw.v[v].idx;
        const uint32_t Hv = vw.v[v].H;
        while (__ballot(m != 0ull) != 0ull) {
          int k[4];
          uint32_t got[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            k[j] = -1;
            if (m) { k[j] = __ffsll((long long)m) - 1; m &= m - 1ull; }
            got[j] = idx[k[j] >= 0 ? (uint64_t)(rec.x0 + (k[j] >> 3)) * Hv + rec.y0 + (k[j] & 7) : 0];
          }
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (k[j] >= 0 && got[j] == pid) win[v] |= 1ull << k[j];
        }
      }
    }
#pragma unroll
    for (int v = 0; v < 8; v++)
      if (big) win[v] = 0ull;          // (a triangle that is big in any of the views belongs to k_fuse_big_any for all of them)
    if (first) {
#pragma unroll
      for (int v = 0; v < 8; v++) cnt += (uint32_t)__popcll(win[v]);
      vis = __ballot(cnt != 0u);
      if (vis == 0ull || (SMESH_ABL(a.dbg) & 1)) return;
      inc = wave_scan_incl_u(cnt);
      exc = inc - cnt;
      wave_sync();                     // (S.view is in LDS)
    }
    // the chunk: the longest run of lanes from s on whose entries fit the list (inc is monotone: the lanes that fit are a run)
    const unsigned long long fit = __ballot(l >= s && inc - base <= (uint32_t)kWideCap);
    const int e1 = s + (int)__popcll(fit);     // one past the chunk's last lane (>= s + 1: a triangle has at most 512 entries)
    if (l >= s && l < e1 && cnt != 0u) {
      uint32_t pos = exc - base;
#pragma unroll
      for (int v = 0; v < 8; v++) {
        if (win[v] == 0ull) continue;
        const uint32_t nt = (uint32_t)__popcll(win[v]);                                   // this primitive's pixels in this view (Mesh.h:90-93)
        const uint32_t Hv = S.view[v].H, ox = org[v] & 0xFFFFu, oy = org[v] >> 16;
        for (unsigned long long m = win[v]; m; m &= m - 1ull) {                          // ascending bits = image order (x, then y)
          const int k = __ffsll((long long)m) - 1;
          S.px[pos] = ((ox + (uint32_t)(k >> 3)) * Hv + oy + (uint32_t)(k & 7)) | ((uint32_t)v << 29);
          S.nl[pos] = (uint16_t)((nt - 1u) | ((uint32_t)l << 6));
          pos++;
        }
      }
    }
    wave_sync();
    // ---- the walk: entry i of the chunk is consumed from ring slot i % K, which is refilled with entry i + K at once.  A slot also
    // carries the accumulator row of the triangle that STARTS at its entry, requested together with the entry's class vector, so the
    // row is there when the walk arrives.  Everything about an entry is wave-uniform (scalar registers).
    const uint32_t n = (e1 >= 64 ? (uint32_t)__builtin_amdgcn_readlane((int)inc, 63) : (uint32_t)__builtin_amdgcn_readlane((int)inc, e1 - 1)) - base;
    fvec4 p[K][NCH], rw[K][NCH], ac[NCH];
    float w[K];
    uint32_t rid[K];
    bool start[K];
    uint32_t prev_issued = 0xFFFFFFFFu;      // row of the entry issued last
    auto issue = [&](const int k, const uint32_t j) {
      const uint32_t ex = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.px[j]);
      const uint32_t en = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.nl[j]);
      const uint32_t ew = (uint32_t)__builtin_amdgcn_readfirstlane(__float_as_int(S.w0[en & 63u]));
      rid[k] = (uint32_t)__builtin_amdgcn_readlane((int)pid, (int)(en >> 6));
      const uint32_t v = ex >> 29;
      const uint64_t pix = ex & 0x1FFFFFFFu;
      const float* probs = uniform_ptr(S.view[v].probs);
      const float* weights = uniform_ptr(S.view[v].weights);
      if (!(SMESH_ABL(a.dbg) & 8)) load_wide<NCH>(probs + pix * C, C, l, p[k]);
      w[k] = __uint_as_float(ew) * (weights ? weights[pix] : 1.0f);
      start[k] = rid[k] != prev_issued;
      if (start[k] && !(SMESH_ABL(a.dbg) & 4)) load_wide<NCH>(a.acc + (uint64_t)rid[k] * C, C, l, rw[k]);
      prev_issued = rid[k];
    };
#pragma unroll
    for (int k = 0; k < K; k++) { w[k] = 0.0f; rid[k] = 0u; start[k] = false; if ((uint32_t)k < n) issue(k, (uint32_t)k); }
    uint32_t cur = 0xFFFFFFFFu;
    for (uint32_t i0 = 0; i0 < n; i0 += K) {
#pragma unroll
      for (int k = 0; k < K; k++) {
        const uint32_t i = i0 + (uint32_t)k;
        if (i < n) {
          if (start[k]) {
            if (cur != 0xFFFFFFFFu && !(SMESH_ABL(a.dbg) & 2)) store_wide<NCH>(a.acc + (uint64_t)cur * C, C, l, ac);
            cur = rid[k];
#pragma unroll
            for (int c = 0; c < NCH; c++) ac[c] = rw[k][c];
          }
          fuse_pixel_wide<KIND, NCH>(ac, p[k], C, l, w[k]);            // Mesh.h:103
          if (i + K < n) issue(k, i + K);
        }
      }
    }
    if (cur != 0xFFFFFFFFu && !(SMESH_ABL(a.dbg) & 2)) store_wide<NCH>(a.acc + (uint64_t)cur * C, C, l, ac);
    if (e1 >= 64 || (vis >> e1) == 0ull) break;
    base += n;
    s = e1;
    wave_sync();                     // (the list is rewritten)
  }
}

// ------------------------------------------------------------------------------------------------
// Triangle-order fusion for TEXEL primitives (render.texels + fuse_view, C <= kSlice).  A triangle owns a contiguous
// range of texel rows and nobody else writes them, so the lane that owns the triangle can read-modify-write the
// row of each of its visible pixels in plain memory operations, in image order (= the reference's order per row).
// The per-view count of a texel is the number of the triangle's visible pixels that carry the same texel id.
// Big triangles (tail blocks, one wave each): lanes stride the pixels, so rows are shared inside the wave -> the
// aggregator's scratch histogram and float atomics, like the generic path, but confined to the triangle's own rows.
// ------------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void fuse_big_texel_triangles(const TriFuseArgs& a, uint32_t worker, uint32_t nworkers) {
  const int l = threadIdx.x;
  const uint32_t C = a.C;
  const uint32_t nbig = min(*a.big_len, a.big_capacity);
  for (uint32_t q = worker; q < nbig; q += nworkers) {
    const uint32_t f = a.big_queue[q];
    const TriFrag rec = a.frags[f];
    if (rec.kind != 2) continue;
    const uint32_t first = a.tex_first[f], res = a.tex_res[f], cnt = res * (res + 1u) / 2u;
    const int x0 = rec.x0, y0 = rec.y0, x1 = (int)(rec.mask & 0xFFFFu), y1 = (int)((rec.mask >> 16) & 0xFFFFu);
    const int bh = y1 - y0 + 1;
    const long long npx = (long long)(x1 - x0 + 1) * bh;
    for (long long i = l; i < npx; i += kWave) {
      const uint32_t v = a.idx[(uint64_t)(x0 + (int)(i / bh)) * a.H + (y0 + (int)(i % bh))];
      if (v - first < cnt) atomicAdd(&a.count[v], 1u);
    }
    __threadfence();
    for (long long i = l; i < npx; i += kWave) {
      const uint64_t pix = (uint64_t)(x0 + (int)(i / bh)) * a.H + (y0 + (int)(i % bh));
      const uint32_t v = a.idx[pix];
      if (!(v - first < cnt)) continue;
      const uint32_t n = __hip_atomic_load(&a.count[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float* __restrict__ pr = a.probs + pix * C;
      float sum = 0.0f, best = 0.0f;
      uint32_t am = 0;
      for (uint32_t c = 0; c < C; c++) {
        const float pc = pr[c];
        sum = sum + pc;
        if (KIND == SMESH_AGG_SUMMAX && (c == 0 || pc > best)) { best = pc; am = c; }
      }
      if (!(sum > 0.5f)) continue;
      const float w = (a.iew * (1.0f / ((float)n)) + (1 - a.iew) * 1.0f) * (a.weights ? a.weights[pix] : 1.0f);
      float* row = a.acc + (uint64_t)v * C;
      if (KIND == SMESH_AGG_SUMMAX) unsafeAtomicAdd(&row[am], best * w);
      else if (KIND == SMESH_AGG_MUL) {
        // a coarse texel of a big triangle receives thousands of pixels: this view's terms are summed in DOUBLE, from zero, in the
        // aggregator's scratch rows (all zero between launches), and folded into the (hi, lo) row below
        double* drow = a.acc_d + (uint64_t)v * C;
        for (uint32_t c = 0; c < C; c++) unsafeAtomicAdd(&drow[c], (double)contribution<KIND>(pr[c], w));
      } else {
        for (uint32_t c = 0; c < C; c++) unsafeAtomicAdd(&row[c], contribution<KIND>(pr[c], w));
      }
    }
    __threadfence();
    if constexpr (KIND == SMESH_AGG_MUL) {
      // Mul ("Mul state", fuse_tri.inc.hpp): the texels that received terms (one lane each: the triangle owns its rows) fold their
      // double sums into the (hi, lo) row, re-centred on its largest finite element, and leave the scratch row zero again
      for (uint32_t tx = first + (uint32_t)l; tx - first < cnt; tx += kWave) {
        if (__hip_atomic_load(&a.count[tx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) continue;
        float* hi = a.acc + (uint64_t)tx * C;
        float* lo = a.acc_lo + (uint64_t)tx * C;
        double* drow = a.acc_d + (uint64_t)tx * C;
        float m = -INFINITY;
        for (uint32_t c = 0; c < C; c++) { const float h = hi[c]; if (h > m && h < INFINITY) m = h; }
        const float centre = m > -INFINITY ? m : 0.0f;
        for (uint32_t c = 0; c < C; c++) {
          float h = hi[c], r = lo[c];
          mul_fold(h, r, centre, __hip_atomic_load(&drow[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          hi[c] = h; lo[c] = r; drow[c] = 0.0;
        }
      }
      __threadfence();
    }
    for (long long i = l; i < npx; i += kWave) {
      const uint32_t v = a.idx[(uint64_t)(x0 + (int)(i / bh)) * a.H + (y0 + (int)(i % bh))];
      if (v - first < cnt) a.count[v] = 0u;
    }
    __threadfence();
  }
}

template <int KIND>
__global__ __launch_bounds__(kWave) void k_fuse_texel_big(TriFuseArgs a) { fuse_big_texel_triangles<KIND>(a, blockIdx.x, gridDim.x); }

// kTexelBlock lanes per workgroup.  Texel meshes are dense meshes seen from afar: at BASELINE cfg4 seven triangles out of eight emit
// no fragment at all, so a wave that walked its own 64 triangles to the end issued every one of the ~30 memory instructions of a
// pixel (class vector, accumulator row in, row out) for a handful of active lanes -- and a memory instruction costs the texture
// path the same whether 8 or 64 lanes take part.  So the workgroup first collects its visible pixels in LDS (the ones that are
// the only pixel of their texel in this view: nobody else touches that row) and then deals them out one per lane: the rows move
// with full waves.  Pixels that share their texel with another pixel of the triangle stay with the triangle's own lane, in image
// order, as before (the reference's order of additions; a row with a single contribution has no order to keep).
// 256 lanes (round 4; 512 until then): k_fuse_texel_multi takes 136 registers, i.e. three waves per SIMD -- of an eight-wave workgroup
// only ONE fits a CU (two waves per SIMD), of a four-wave one three.  cfg4 / cfg4t, us per view in the kernel, same box:
// 512 lanes 112.5 / 150.0, 256 lanes 103.5 / 143.0, 128 lanes 103.7 / 144.1, 64 lanes 107.0 / 148.4; a stated budget of four waves per SIMD
// (128 registers, 14 spilled) 106.3 / 149.6 at 512 lanes and 102.0 / 146.5 at 256 (profiles/r04_texel_block_sweep.txt).
constexpr int kTexelBlock = 256;

// Mul ("Mul state", fuse_tri.inc.hpp): a texel receives a pixel or two per view, so every term is folded into the (hi, lo) row on its
// own, in double, the row re-centred on its largest finite element.
template <int N>
__device__ __forceinline__ void mul_fold_pixel(float (&hi)[N], float (&lo)[N], const float (&p)[N], int cw, float w) {
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < N; j++) if (j < cw && hi[j] > m && hi[j] < INFINITY) m = hi[j];
  const float centre = m > -INFINITY ? m : 0.0f;
#pragma unroll
  for (int j = 0; j < N; j++) if (j < cw) mul_fold(hi[j], lo[j], centre, (double)contribution<SMESH_AGG_MUL>(p[j], w));
}

template <int KIND>
__device__ __forceinline__ void fuse_texel_pixel(const TriFuseArgs& a, const uint32_t C, const uint64_t pix, const uint32_t v, const uint32_t n) {
  float p[kSlice];
  load_slice(a.probs + pix * C, (int)C, p);
  // the row is requested together with the class vector (wasted only for the rare don't-care pixel)
  float* row = a.acc + (uint64_t)v * C;
  float accr[kSlice];
  load_slice(row, (int)C, accr);
  constexpr int NL = KIND == SMESH_AGG_MUL ? kSlice : 1;
  float lo[NL];
  if constexpr (KIND == SMESH_AGG_MUL) load_slice(a.acc_lo + (uint64_t)v * C, (int)C, lo);
  const float wt = a.weights ? a.weights[pix] : 1.0f;
  float sum = 0.0f, best = p[0];
  int am = 0;
#pragma unroll
  for (int j = 0; j < kSlice; j++)
    if (j < (int)C) {
      sum = sum + p[j];
      if (KIND == SMESH_AGG_SUMMAX && p[j] > best) { best = p[j]; am = j; }
    }
  if (!(sum > 0.5f)) return;                          // Mesh.h:98
  const float w = (a.iew * (1.0f / ((float)n)) + (1 - a.iew) * 1.0f) * wt;   // :100-103
  if constexpr (KIND == SMESH_AGG_MUL) {
    mul_fold_pixel(accr, lo, p, (int)C, w);
    store_slice(a.acc_lo + (uint64_t)v * C, (int)C, lo);
  } else {
    accumulate_slice<KIND>(accr, p, (int)C, w, am);
  }
  store_slice(row, (int)C, accr);
}

template <int KIND>
__global__ __launch_bounds__(kTexelBlock) void k_fuse_texel(TriFuseArgs a) {
  __shared__ uint32_t s_pix[kTexelBlock], s_tex[kTexelBlock];
  __shared__ uint32_t s_items;
  const uint32_t C = a.C;
  const uint64_t f = (uint64_t)blockIdx.x * kTexelBlock + threadIdx.x;
  if (threadIdx.x == 0) s_items = 0u;
  __syncthreads();
  TriFrag rec;
  rec.x0 = 0; rec.y0 = 0; rec.kind = 0; rec.pad = 0; rec.mask = 0ull;
  uint32_t first = 0, cnt = 0;
  if (f < a.F && (!a.tex_kinds || a.tex_kinds[f] != 0)) rec = a.frags[f];   // (no record where the kind byte says "nothing")
  if (rec.kind == 1) {   // (the tables are not read for the triangles that emitted nothing)
    first = a.tex_first[f];
    const uint32_t res = a.tex_res[f];
    cnt = res * (res + 1u) / 2u;
  }
  auto pixel = [&](int k) -> uint64_t { return (uint64_t)(rec.x0 + (k >> 3)) * a.H + rec.y0 + (k & 7); };
  // pass 1: which emitted fragments won the depth test (the pixel then holds one of this triangle's texels)
  unsigned long long m = rec.kind == 1 ? rec.mask : 0ull;
  unsigned long long win = 0ull;
  while (__ballot(m != 0ull) != 0ull) {
    int k[4];
    uint32_t got[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      k[j] = -1;
      if (m) { k[j] = __ffsll((long long)m) - 1; m &= m - 1ull; }
      got[j] = a.idx[k[j] >= 0 ? pixel(k[j]) : 0];
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (k[j] >= 0 && got[j] - first < cnt) win |= 1ull << k[j];
  }
  for (m = win; m; m &= m - 1ull) {
    const uint64_t pix = pixel(__ffsll((long long)m) - 1);
    const uint32_t v = a.idx[pix];
    uint32_t n = 0;                                   // Mesh.h:90-93 restricted to this triangle's pixels
    for (unsigned long long m2 = win; m2; m2 &= m2 - 1ull) n += a.idx[pixel(__ffsll((long long)m2) - 1)] == v ? 1u : 0u;
    if (n == 1u) {                                    // the texel's only pixel in this view: any lane may fuse it
      const uint32_t slot = atomicAdd(&s_items, 1u);
      if (slot < (uint32_t)kTexelBlock) { s_pix[slot] = (uint32_t)pix; s_tex[slot] = v; continue; }   // (pixel offsets fit 32 bits: W, H <= 65536)
    }
    fuse_texel_pixel<KIND>(a, C, pix, v, n);          // shares its row with a later pixel of this lane: in image order, here
  }
  __syncthreads();
  const uint32_t items = min(s_items, (uint32_t)kTexelBlock);
  if (threadIdx.x < items) fuse_texel_pixel<KIND>(a, C, (uint64_t)s_pix[threadIdx.x], s_tex[threadIdx.x], 1u);
}

// Up to eight views of a texel renderer in one launch (smesh_fuse_views).  The workgroup first compacts, in order, the triangles
// that emitted fragments in ANY of the views (one in eight per view at cfg4); then every lane takes one of them through all the
// views: view by view, its visible pixels in image order, the accumulator row of the current texel kept in registers for as long
// as consecutive pixels -- of the same view or the next -- carry the same texel.  A triangle seen in several views (most are:
// the cameras of a group look at the same surface) then reads and writes its 160-byte rows once instead of once per view,
// which is half of what a cfg4 view moves.  Per row the additions happen in the order of one launch per view: view order,
// then image order.  Triangles with a box over 8 x 8 in a view: k_fuse_texel_big of that view, launched afterwards.
template <int KIND>
__global__ __launch_bounds__(kTexelBlock) void k_fuse_texel_multi(TriFuseArgs a, TriViews<8> vw, int nv) {
  __shared__ TriView s_view[8];
  __shared__ uint32_t s_tri[kTexelBlock];
  __shared__ uint8_t s_in[kTexelBlock];      // bit v: the triangle emitted fragments in view v
  __shared__ uint32_t s_wave_count[kTexelBlock / kWave];
  const uint32_t C = a.C;
  const int t = threadIdx.x, l = t & (kWave - 1), wv = t / kWave;
  const uint64_t f = (uint64_t)blockIdx.x * kTexelBlock + t;
  if (t == 0) {
#pragma unroll
    for (int v = 0; v < 8; v++) s_view[v] = vw.v[v];
  }
  __syncthreads();
  // (round 3) Every step that can be issued for all views at once IS: the kernel is a chain of dependent memory round trips per
  // lane (record -> index -> row / class vector, view after view), and its TCPs sat waiting on pending requests 63 % of the time.
  // The eight records of a triangle are requested together (the short-circuit `seen = seen || ...` made them eight dependent
  // loads for the seven triangles out of eight that no view sees), again after the compaction, and so is each view's first
  // candidate pixel.
  bool seen = false;
  uint32_t in_views = 0u;
  {
    uint16_t kind[8];
#pragma unroll
    for (int v = 0; v < 8; v++)   // (a byte per triangle where the renderer left them: seven triangles in eight emitted nothing at cfg4)
      kind[v] = (v < nv && f < a.F) ? (s_view[v].kinds ? (uint16_t)s_view[v].kinds[f] : s_view[v].frags[f].kind) : (uint16_t)0;
#pragma unroll
    for (int v = 0; v < 8; v++) in_views |= kind[v] == 1 ? 1u << v : 0u;
    seen = in_views != 0u;
  }
  const unsigned long long ballot = __ballot(seen);
  if (l == 0) s_wave_count[wv] = (uint32_t)__popcll(ballot);
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kTexelBlock / kWave; w++) {
    const uint32_t c = s_wave_count[w];
    if (w < wv) base += c;
    total += c;
  }
  if (seen) {
    const uint32_t slot = base + (uint32_t)__popcll(ballot & ((1ull << l) - 1ull));
    s_tri[slot] = (uint32_t)f;
    s_in[slot] = (uint8_t)in_views;
  }
  __syncthreads();
  if ((uint32_t)t >= total) return;
  const uint32_t g = s_tri[t];
  in_views = s_in[t];       // (a triangle seen at all is seen in one or two of the eight views at cfg4: only those records are read)
  // this triangle in all views: box origin, mask of emitted fragments, and the index under the first of them -- all in flight together
  uint32_t org[8], t0[8];
  unsigned long long msk[8];
#pragma unroll
  for (int v = 0; v < 8; v++) {
    org[v] = 0u; msk[v] = 0ull;
    if (in_views >> v & 1u) {
      const TriFrag rec = s_view[v].frags[g];
      org[v] = (uint32_t)rec.x0 | ((uint32_t)rec.y0 << 16);
      msk[v] = rec.mask;
    }
  }
  const uint32_t first = a.tex_first[g], res = a.tex_res[g], cnt = res * (res + 1u) / 2u;
#pragma unroll
  for (int v = 0; v < 8; v++) {
    t0[v] = 0xFFFFFFFFu;
    if (v < nv && msk[v]) {
      const int k = __ffsll((long long)msk[v]) - 1;
      t0[v] = s_view[v].idx[(uint64_t)((org[v] & 0xFFFFu) + (uint32_t)(k >> 3)) * s_view[v].H + (org[v] >> 16) + (uint32_t)(k & 7)];
    }
  }
  float accr[kSlice];
  constexpr int NL = KIND == SMESH_AGG_MUL ? kSlice : 1;
  float lo[NL];                 // Mul: the row's second plane
  uint32_t cur = 0xFFFFFFFFu;   // texel whose row is in accr
  bool dirty = false;
#pragma unroll
  for (int v = 0; v < 8; v++) {
    if (v >= nv || msk[v] == 0ull) continue;
    const uint32_t* __restrict__ idx = s_view[v].idx;
    const float* __restrict__ probs = s_view[v].probs;
    const float* __restrict__ weights = s_view[v].weights;
    const uint32_t Hv = s_view[v].H;
    const uint32_t ox = org[v] & 0xFFFFu, oy = org[v] >> 16;
    auto pixel = [&](int k) -> uint64_t { return (uint64_t)(ox + (uint32_t)(k >> 3)) * Hv + oy + (uint32_t)(k & 7); };
    // which emitted fragments won the depth test (the pixel then holds one of this triangle's texels)
    const int kfirst = __ffsll((long long)msk[v]) - 1;
    unsigned long long win = (t0[v] - first < cnt) ? 1ull << kfirst : 0ull;
    for (unsigned long long m = msk[v] & (msk[v] - 1ull); m; m &= m - 1ull) {
      const int k = __ffsll((long long)m) - 1;
      if (idx[pixel(k)] - first < cnt) win |= 1ull << k;
    }
    const bool single = win == (1ull << kfirst);   // the first candidate is the only visible pixel: its texel is t0, its count 1 -- no further index loads
    for (unsigned long long m = win; m; m &= m - 1ull) {
      const uint64_t pix = pixel(__ffsll((long long)m) - 1);
      uint32_t tex = t0[v], n = 1u;
      if (!single) {
        tex = idx[pix];
        n = 0u;                                           // Mesh.h:90-93 restricted to this triangle's pixels
        for (unsigned long long m2 = win; m2; m2 &= m2 - 1ull) n += idx[pixel(__ffsll((long long)m2) - 1)] == tex ? 1u : 0u;
      }
      // the class vector and (if it is not the one in registers) the texel's row: requested together
      float p[kSlice];
      load_slice(probs + pix * C, (int)C, p);
      if (tex != cur) {
        if (dirty) {
          store_slice(a.acc + (uint64_t)cur * C, (int)C, accr);
          if constexpr (KIND == SMESH_AGG_MUL) store_slice(a.acc_lo + (uint64_t)cur * C, (int)C, lo);
        }
        load_slice(a.acc + (uint64_t)tex * C, (int)C, accr);
        if constexpr (KIND == SMESH_AGG_MUL) load_slice(a.acc_lo + (uint64_t)tex * C, (int)C, lo);
        cur = tex;
        dirty = false;
      }
      const float wt = weights ? weights[pix] : 1.0f;
      float sum = 0.0f, best = p[0];
      int am = 0;
#pragma unroll
      for (int j = 0; j < kSlice; j++)
        if (j < (int)C) {
          sum = sum + p[j];
          if (KIND == SMESH_AGG_SUMMAX && p[j] > best) { best = p[j]; am = j; }
        }
      if (!(sum > 0.5f)) continue;                        // Mesh.h:98
      const float w = (a.iew * (1.0f / ((float)n)) + (1 - a.iew) * 1.0f) * wt;   // :100-103
      if constexpr (KIND == SMESH_AGG_MUL) mul_fold_pixel(accr, lo, p, (int)C, w);
      else accumulate_slice<KIND>(accr, p, (int)C, w, am);
      dirty = true;
    }
  }
  if (dirty) {
    store_slice(a.acc + (uint64_t)cur * C, (int)C, accr);
    if constexpr (KIND == SMESH_AGG_MUL) store_slice(a.acc_lo + (uint64_t)cur * C, (int)C, lo);
  }
}

// ------------------------------------------------------------------------------------------------
// Fallback for class counts whose strip does not fit LDS: per-pixel weights, then a flat scatter.
// ------------------------------------------------------------------------------------------------
template <int KIND>
__global__ void k_pixel_weight(ScatterArgs a, float* __restrict__ wpix, uint32_t* __restrict__ amax) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N) return;
  const uint32_t v = a.idx[i];
  float w = 0.0f;
  uint32_t m = 0;
  if (v < a.P) {
    const float* row = a.probs + i * a.C;
    float s = 0.0f, best = row[0];
    for (uint32_t c = 0; c < a.C; c++) {
      const float p = row[c];
      s = s + p;
      if (KIND == SMESH_AGG_SUMMAX && p > best) { best = p; m = c; }
    }
    if (s > 0.5f) w = pixel_weight(a, v, a.weights ? a.weights[i] : 1.0f);
  }
  wpix[i] = w;
  if (KIND == SMESH_AGG_SUMMAX) amax[i] = m;
}

template <int KIND>
__global__ void k_scatter_flat(ScatterArgs a, const float* __restrict__ wpix, const uint32_t* __restrict__ amax) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.N * a.C) return;
  const uint64_t i = e / a.C;
  const uint32_t c = (uint32_t)(e - i * a.C);
  const float w = wpix[i];
  if (w == 0.0f) return;
  if (KIND == SMESH_AGG_SUMMAX && amax[i] != c) return;
  unsafeAtomicAdd(&a.acc[(uint64_t)a.idx[i] * a.S + c], contribution<KIND>(a.probs[e], w));
}

// Mul on the generic path: one float64 atomic per (pixel, class) into the aggregator's scratch rows (all zero between calls), then
// every row that received terms folds them into its (hi, lo) pair (k_fold_rows) -- "Mul state", fuse_tri.inc.hpp.  Float32 atomics on
// the hi plane (rounds 1-3) missed 1e-5 on get() by two orders of magnitude; this path is what is left when the image records do
// not apply (padded rows, SMESH_ADD_RECORDS=0, SMESH_FUSE=strip), so it is sized for exactness, not speed.
__global__ void k_scatter_flat_mul(ScatterArgs a, const float* __restrict__ wpix, double* __restrict__ acc_d) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.N * a.C) return;
  const uint64_t i = e / a.C;
  const uint32_t c = (uint32_t)(e - i * a.C);
  const float w = wpix[i];
  if (w == 0.0f) return;
  unsafeAtomicAdd(&acc_d[(uint64_t)a.idx[i] * a.C + c], (double)contribution<SMESH_AGG_MUL>(a.probs[e], w));
}
__global__ void k_fold_rows(float* __restrict__ acc, float* __restrict__ acc_lo, double* __restrict__ acc_d, const uint32_t* __restrict__ count,
                            uint64_t P, uint32_t C, uint32_t S) {
  const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= P || (count && count[v] == 0u)) return;     // (count: this view's histogram, when there is one)
  double* drow = acc_d + v * C;
  bool any = false;
  for (uint32_t c = 0; c < C; c++) any = any || drow[c] != 0.0;
  if (!any) return;
  float* hi = acc + v * S;
  float* lo = acc_lo + v * S;
  float m = -INFINITY;
  for (uint32_t c = 0; c < C; c++) { const float h = hi[c]; if (h > m && h < INFINITY) m = h; }
  const float centre = m > -INFINITY ? m : 0.0f;
  for (uint32_t c = 0; c < C; c++) {
    float h = hi[c], r = lo[c];
    mul_fold(h, r, centre, drow[c]);
    hi[c] = h; lo[c] = r; drow[c] = 0.0;
  }
}

// ------------------------------------------------------------------------------------------------
// get(): load -> [Mul: / max element] -> L1 normalise -> NaN/Inf -> 0   (Fusion.h:79-104)
// Reads padded rows [P][S], writes dense [P][C].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float nan_inf_to_zero(float v) { return (isnan(v) || isinf(v)) ? 0.0f : v; }

template <int KIND>
__device__ __forceinline__ void finalize_row(float* row, int C) {
  if (KIND == SMESH_AGG_MUL) {
    float m = row[0];
    for (int c = 1; c < C; c++) if (row[c] > m) m = row[c];
    for (int c = 0; c < C; c++) row[c] = expf(row[c] - m);  // logprob_normalize, cast to float
  }
  float n = 0.0f;
  for (int c = 0; c < C; c++) n = n + fabsf(row[c]);        // l1_norm, sequential float32
  for (int c = 0; c < C; c++) row[c] = nan_inf_to_zero(row[c] / n);
}

// One workgroup of 256 threads = TP consecutive rows (TP C floats of LDS): all threads move the tile in and out with coalesced
// 16-byte accesses, threads 0 .. TP-1 then apply the functor chain to one row each, in the reference's sequential order.  (Round 1
// launched TP threads per workgroup -- a single wave at 150 classes, four of them per CU -- and left rows beyond 220 classes to a
// thread-per-row kernel with uncoalesced accesses: 0.19 TB/s at C = 300.)
template <int KIND>
__global__ __launch_bounds__(256) void k_finalize_tile(const float* __restrict__ acc, const float* __restrict__ acc_lo, float* __restrict__ out,
                                                       uint64_t P, int C, int S, int TP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sp = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x;
  const uint64_t r0 = (uint64_t)blockIdx.x * TP;
  const int nrows = (int)((P - r0) < (uint64_t)TP ? (P - r0) : (uint64_t)TP);
  const int nfl = nrows * C;
  const float* __restrict__ src = acc + r0 * (uint64_t)S;
  float* __restrict__ dst = out + r0 * (uint64_t)C;
  if (S == C) {        // dense rows (the layout in use): the tile is one contiguous run, no division per float
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const float4* src4 = reinterpret_cast<const float4*>(src);
      float4* sp4w = reinterpret_cast<float4*>(sp);
      for (int e = t; e < (nfl >> 2); e += 256) sp4w[e] = src4[e];
      for (int e = (nfl & ~3) + t; e < nfl; e += 256) sp[e] = src[e];
    } else {
      for (int e = t; e < nfl; e += 256) sp[e] = src[e];
    }
  } else {
    for (int e = t; e < nfl; e += 256) {
      const int rr = e / C, c = e - rr * C;
      sp[e] = src[(uint64_t)rr * S + c];
    }
  }
  if (KIND == SMESH_AGG_MUL && acc_lo) {
    // Mul ("Mul state", fuse_tri.inc.hpp): a row is hi + lo.  The lo tile sits behind the hi tile in LDS; the row's thread centres
    // the pair on its largest finite element in double and leaves the float32 values k_mul_normalise would have left in the hi
    // plane -- without writing the state back (get() used to run that kernel over the whole accumulator first: a thread per row,
    // 0.11-0.15 TB/s).
    float* sl = sp + (((size_t)TP * C + 3) & ~(size_t)3);
    const float* __restrict__ lsrc = acc_lo + r0 * (uint64_t)S;
    for (int e = t; e < nfl; e += 256) {
      const int rr = S == C ? 0 : e / C;
      sl[e] = S == C ? lsrc[e] : lsrc[(uint64_t)rr * S + (e - rr * C)];
    }
    __syncthreads();
    if (t < nrows) {
      float* hi = sp + t * C;
      const float* lo = sl + t * C;
      double m = -INFINITY;
      for (int c = 0; c < C; c++) { const double v = (double)hi[c] + (double)lo[c]; if (v > m && v < INFINITY) m = v; }
      if (!(m > -INFINITY)) m = 0.0;
      for (int c = 0; c < C; c++) hi[c] = (float)(((double)hi[c] + (double)lo[c]) - m);
    }
  }
  __syncthreads();
  if (t < nrows) finalize_row<KIND>(sp + t * C, C);
  __syncthreads();
  if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    const float4* sp4 = reinterpret_cast<const float4*>(sp);
    float4* dst4 = reinterpret_cast<float4*>(dst);
    for (int e = t; e < (nfl >> 2); e += 256) dst4[e] = sp4[e];
    for (int e = (nfl & ~3) + t; e < nfl; e += 256) dst[e] = sp[e];
  } else {
    for (int e = t; e < nfl; e += 256) dst[e] = sp[e];
  }
}

template <int KIND>
__global__ void k_finalize_rows(const float* __restrict__ acc, float* __restrict__ out, uint64_t P, int C, int S) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* src = acc + p * S;
  float* dst = out + p * C;
  for (int c = 0; c < C; c++) dst[c] = src[c];
  finalize_row<KIND>(dst, C);
}

// ModelRenderer::render (Mesh.h:25-42): per-pixel gather of the fused annotation rows
// One workgroup = 256 consecutive pixels = 256 C consecutive output floats, written 256 at a time (coalesced; the rows read are 4 C
// contiguous bytes each).  The pixel and class of a thread's next element follow from the previous one by adding 256 / C and
// 256 % C: no division per element (the first version divided a 64-bit element index by C for every float: 85 us per cfg2 image).
__global__ __launch_bounds__(256) void k_gather_annotations(const uint32_t* __restrict__ idx, const float* __restrict__ ann,
                                                            const float* __restrict__ background, float* __restrict__ out,
                                                            uint64_t N, uint32_t P, uint32_t C) {
  __shared__ uint32_t s_idx[256];
  const uint32_t t = threadIdx.x;
  const uint64_t pix0 = (uint64_t)blockIdx.x * 256u;
  const uint32_t npix = (uint32_t)min((uint64_t)256u, N - pix0);
  s_idx[t] = t < npix ? idx[pix0 + t] : 0xFFFFFFFFu;
  __syncthreads();
  const uint32_t q = 256u / C, r = 256u % C;
  uint32_t pl = t / C, c = t % C;                 // element t of the block: pixel pl, class c
  float* __restrict__ o = out + pix0 * C;
  const uint32_t total = npix * C;
  for (uint32_t e = t; e < total; e += 256u) {
    const uint32_t v = s_idx[pl];
    o[e] = v < P ? ann[(uint64_t)v * C + c] : background[c];
    pl += q; c += r;
    if (c >= C) { c -= C; pl++; }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// Dense rows: padding rows to whole cache lines was measured SLOWER (tools/flush_replay.hip: memory-side
// atomics cost per line touched, and dense neighbours share lines), so the stride is C.
inline uint32_t row_stride(uint32_t C) {
  static const int pad = getenv("SMESH_ROW_PAD") ? atoi(getenv("SMESH_ROW_PAD")) : 0;   // experiment knob
  return pad > 1 ? (uint32_t)((C + pad - 1) / pad * pad) : C;
}

// the strip path keeps 64 pixel rows of C floats in LDS (<= 64 KiB without opting into more: C <= 250)

inline size_t strip_lds_bytes(uint32_t C) {
  const size_t pfloats = ((size_t)kWave * C + 3) & ~(size_t)3;
  return pfloats * 4 + ((sizeof(StripLists) + 15) & ~(size_t)15) + sizeof(FlushLists) + 16;
}

inline bool strip_path(uint32_t C) { return strip_lds_bytes(C) <= 64 * 1024; }

// rows per finalize tile
// Rows per workgroup of k_finalize_tile: a multiple of four (16-byte alignment of every tile), at most 256, about 32 KB of LDS
// (64 KB for the widest rows); 0: rows too wide for LDS tiles.
inline int tile_pixels(uint32_t C) {
  if (C > 4000) return 0;
  const int tp = (int)((8192u / C) & ~3u);
  return tp >= 4 ? std::min(tp, 256) : 4;
}

void set_tiling(ScatterArgs& a) {
  const uint32_t strips_x = (uint32_t)div_up(a.W, kSX);
  a.strips_y = (uint32_t)div_up(a.H, kTY);
  a.nstrips = strips_x * a.strips_y;
  a.strips_per_xcd = (uint32_t)div_up(a.nstrips, 8);
  a.strips_per_wave = 1;
  // a full strip's column segment starts at float offset ((x0+seg)*H + y0)*C with y0 a multiple of 16:
  // 16-byte aligned for every column iff H*C is a multiple of 4 and the base pointer is aligned
  a.vec_ok = ((uint64_t)a.H * a.C) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.probs) & 15) == 0;
}

template <int KIND>
int launch_strip(const ScatterArgs& a0, int num_cus, hipStream_t st) {
  ScatterArgs a = a0;
  const size_t lds = strip_lds_bytes(a.C);
  // persistent waves: as many as the LDS lets a CU hold (at most 32), each walking a contiguous strip range
  static const int env_wpc = getenv("SMESH_WAVES_PER_CU") ? atoi(getenv("SMESH_WAVES_PER_CU")) : 0;
  int waves_per_cu = (int)std::min<size_t>(16, (160 * 1024) / lds);   // measured best on cfg2 (8 strips per wave)
  if (env_wpc > 0) waves_per_cu = env_wpc;
  if (waves_per_cu < 1) waves_per_cu = 1;
  uint32_t waves = (uint32_t)num_cus * (uint32_t)waves_per_cu;
  waves = (waves + 7u) & ~7u;
  a.strips_per_wave = (uint32_t)div_up(a.nstrips, waves);
  if (a.strips_per_wave < 1) a.strips_per_wave = 1;
  waves = ((uint32_t)div_up(a.nstrips, a.strips_per_wave) + 7u) & ~7u;
  const dim3 grid(waves), block(kWave);
  // class counts of the benchmark configs get compile-time loops; everything else runs the same
  // kernel with a run-time C (the reference needs a rebuild with -DCLASSES_NUMS for each count)
  if (SMESH_ABL(a.dbg) & 32) {
    switch (a.C) {
      case 19: hipLaunchKernelGGL((k_scatter_strip<19, KIND, true>), grid, block, lds, st, a); break;
      default: hipLaunchKernelGGL((k_scatter_strip<0, KIND, true>), grid, block, lds, st, a); break;
    }
  } else {
    switch (a.C) {
      case 5:  hipLaunchKernelGGL((k_scatter_strip<5, KIND, false>), grid, block, lds, st, a); break;
      case 19: hipLaunchKernelGGL((k_scatter_strip<19, KIND, false>), grid, block, lds, st, a); break;
      case 40: hipLaunchKernelGGL((k_scatter_strip<40, KIND, false>), grid, block, lds, st, a); break;
      default: hipLaunchKernelGGL((k_scatter_strip<0, KIND, false>), grid, block, lds, st, a); break;
    }
  }
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

int launch_hist(const ScatterArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_hist_strip, dim3(a.strips_per_xcd * 8), dim3(kWave), 0, st, a);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

}  // namespace

// Mul only (fuse_tri.inc.hpp, "Mul state"): re-centre every row of the (hi, lo) accumulator pair on its largest finite element and
// renormalise the pair (hi = the float32 nearest to the value, lo = the remainder; `fold`: lo := 0, the value rounded to the
// hi plane alone -- what leaves the library as "the raw accumulator").  Runs ahead of the fusion kernels that only know the hi
// plane, in get(), and before the raw accumulator is read or all-reduced.
namespace {
__global__ void k_mul_normalise(float* __restrict__ acc, float* __restrict__ acc_lo, uint64_t P, uint32_t C, uint32_t S, int fold) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float* __restrict__ hi = acc + p * S;
  float* __restrict__ lo = acc_lo + p * S;
  double m = -INFINITY;
  for (uint32_t c = 0; c < C; c++) { const double t = (double)hi[c] + (double)lo[c]; if (t > m && t < INFINITY) m = t; }
  if (!(m > -INFINITY)) m = 0.0;
  for (uint32_t c = 0; c < C; c++) {
    const double t = ((double)hi[c] + (double)lo[c]) - m;
    const float h = (float)t;
    hi[c] = h;
    lo[c] = (fold || !(h > -INFINITY && h < INFINITY)) ? 0.0f : (float)(t - (double)h);
  }
}
}  // namespace

struct smesh_aggregator {
  DeviceCtx* ctx = nullptr;
  uint64_t P = 0;
  uint32_t C = 0;
  int kind = 0;
  float iew = 0.5f;
  uint32_t S = 0;             // accumulator row stride in floats (C rounded up to 16)
  float* acc = nullptr;       // float32[P*S]
  float* acc_lo = nullptr;    // Mul only: float32[P*S], a row's value is acc + acc_lo (fuse_tri.inc.hpp, "Mul state")
  double* acc_d = nullptr;    // Mul with texel renderers only (allocated on first use): float64[P*C], one view's terms of the texels of
                              // big triangles, all zero between launches (k_fuse_texel_big)
  uint32_t* count = nullptr;  // uint32[P], all zero between add() calls
  Scratch st_idx, st_probs, st_w;        // host->device staging
  Scratch nm_idx, nm_probs, nm_w;        // normalised (contiguous) copies
  Scratch fb_w, fb_amax;                 // fallback path scratch
  Scratch pw;                            // per-pixel weight image of the current view
  hipEvent_t ev_staged = nullptr;        // host inputs have been copied into the staging buffers
  Scratch out_tmp;                       // get(): normalised result before the D2H copy
  ImageRecords rec;                      // add() on an image the library did not render: per-primitive records (image_records.hip)
  ImageRecords rec_many[7];              // add_many(): the record sets of the further images of a group of up to eight (rec is the first)
  // Exchange of row ranges beside the fusion (smesh_allreduce_rows, comm.cpp): ev_part marks the main stream where the range became
  // final, ev_xchg the exchange stream behind the range's collective; xchg_pending: the main stream has not yet waited for ev_xchg.
  hipEvent_t ev_part = nullptr, ev_xchg = nullptr;
  bool xchg_pending = false;
  Scratch xchg_stage;                    // Mul: float64 image of the rows being exchanged ((hi, lo) pairs summed exactly)
  // After smesh_reduce_scatter only rows [owned_lo, owned_hi) hold the sum over the ranks: everything but get_rows inside that
  // range and reset() is refused until reset() (scattered == true).
  bool scattered = false;
  uint64_t owned_lo = 0, owned_hi = 0;
  std::mutex mu;
};

int smesh_aggregator_join_exchange(smesh_aggregator* a);
int smesh_aggregator_refuse_scattered(smesh_aggregator* a, const char* what);
static int ensure_acc_d(smesh_aggregator* a);
static void copy_ring_prepare(DeviceCtx* ctx);

bool smesh_aggregator_can_fuse_triangles(smesh_aggregator* a, uint64_t F);
const char* smesh_aggregator_fuse_kernel_name(smesh_aggregator* a, bool reordered);
int smesh_aggregator_fuse_triangles(smesh_aggregator* a, uint64_t F, const uint32_t* prim_id, uint32_t big_capacity,
                                    const RenderedView* views, int nviews, int part = 0, int nparts = 1);
void smesh_fuse_part_rows(uint64_t F, int part, int nparts, uint64_t* f_lo, uint64_t* f_hi);
void smesh_note_fuse(const char* kernel, const char* path);   // raster.hip: what smesh_last_fuse_kernel / smesh_last_add_path report

namespace {

int mul_normalise(smesh_aggregator* a, bool fold) {
  if (a->kind != SMESH_AGG_MUL || a->P == 0) return SMESH_OK;
  hipLaunchKernelGGL(k_mul_normalise, dim3((uint32_t)div_up(a->P, 256)), dim3(256), 0, a->ctx->stream, a->acc, a->acc_lo,
                     a->P, a->C, a->S, fold ? 1 : 0);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}
int mul_recentre(smesh_aggregator* a) { return mul_normalise(a, false); }

int stage_in(DeviceCtx* ctx, Scratch& st, const void* host, size_t bytes, const void** dev) {
  SMESH_TRY(st.reserve(bytes));
  SMESH_HIP(hipMemcpyAsync(st.ptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  *dev = st.ptr;
  return SMESH_OK;
}

size_t idx_itemsize(int dt) { return (dt == SMESH_IDX_U64 || dt == SMESH_IDX_I64) ? 8 : 4; }

// Class count from which add() on a foreign image rebuilds per-primitive records and fuses in triangle order (image_records.hip).
// Round 2 (passes A / B, ~3 integer atomics per run of pixels, ~100 us per cfg2-sized image) set it to 32: ms per view records /
// scatter at cfg2's geometry C = 19 0.149 / 0.122, 32 0.196 / 0.391.  Round 3 (passes M / R: one atomic per (primitive, strip) group,
// asynchronous add): 2 0.053 / 0.093, 5 0.053 / 0.063, 13 0.066 / 0.105, 19 0.075 / 0.110, 27 0.108 / 0.144, 32 0.121 / 0.374,
// 48 0.157, 64 0.294, 150 0.717 (tools/generic_add_sweep.py, profiles/r03_foreign_images_class_sweep.txt): the records win at every
// class count, and they are deterministic with the reference's order of additions.  SMESH_ADD_RECORDS_MIN_C moves the threshold,
// SMESH_ADD_RECORDS=0 turns the records off.
constexpr uint32_t kAddRecordsMinC = 0;

// Core of add() once every buffer is in device memory.
int add_device(smesh_aggregator* a, const void* d_idx, int idx_dtype, const int64_t is[2],
               const float* d_probs, const int64_t ps[3], const float* d_w, const int64_t ws[2],
               uint64_t W, uint64_t H) {
  DeviceCtx* ctx = a->ctx;
  hipStream_t st = ctx->stream;
  const uint64_t N = W * H;
  const uint32_t C = a->C;

  // ---- normalise layouts only when needed -------------------------------------------------
  const uint32_t* idx = nullptr;
  const bool idx_contig = is[0] == (int64_t)H && is[1] == 1;
  if (idx_contig && (idx_dtype == SMESH_IDX_U32 || idx_dtype == SMESH_IDX_I32)) {
    idx = static_cast<const uint32_t*>(d_idx);  // int32 -1 and uint32 0xFFFFFFFF share a bit pattern
  } else {
    SMESH_TRY(a->nm_idx.reserve(N * 4));
    uint32_t* out = static_cast<uint32_t*>(a->nm_idx.ptr);
    const dim3 g((uint32_t)div_up(N, 256)), b(256);
    switch (idx_dtype) {
      case SMESH_IDX_U32: hipLaunchKernelGGL(k_gather_idx<uint32_t>, g, b, 0, st, (const uint32_t*)d_idx, is[0], is[1], out, N, (uint32_t)H); break;
      case SMESH_IDX_I32: hipLaunchKernelGGL(k_gather_idx<int32_t>, g, b, 0, st, (const int32_t*)d_idx, is[0], is[1], out, N, (uint32_t)H); break;
      case SMESH_IDX_U64: hipLaunchKernelGGL(k_gather_idx<uint64_t>, g, b, 0, st, (const uint64_t*)d_idx, is[0], is[1], out, N, (uint32_t)H); break;
      default:            hipLaunchKernelGGL(k_gather_idx<int64_t>, g, b, 0, st, (const int64_t*)d_idx, is[0], is[1], out, N, (uint32_t)H); break;
    }
    SMESH_HIP(hipGetLastError());
    idx = out;
  }
  const float* probs = d_probs;
  const bool probs_contig = ps[0] == (int64_t)(H * C) && ps[1] == (int64_t)C && ps[2] == 1;
  if (!probs_contig || (reinterpret_cast<uintptr_t>(d_probs) & 15)) {
    SMESH_TRY(a->nm_probs.reserve(N * C * 4));
    launch_gather_probs(d_probs, ps, static_cast<float*>(a->nm_probs.ptr), W, H, C, st);
    SMESH_HIP(hipGetLastError());
    probs = static_cast<const float*>(a->nm_probs.ptr);
  }
  const float* weights = d_w;
  if (d_w && !(ws[0] == (int64_t)H && ws[1] == 1)) {
    SMESH_TRY(a->nm_w.reserve(N * 4));
    hipLaunchKernelGGL(k_gather_f32_2d, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, d_w, ws[0], ws[1],
                       static_cast<float*>(a->nm_w.ptr), N, (uint32_t)H);
    SMESH_HIP(hipGetLastError());
    weights = static_cast<const float*>(a->nm_w.ptr);
  }

  // ---- triangle-order fusion on records built from the image (image_records.hip) ------------------
  // Every accumulator row gets one owner and the reference's order of additions: deterministic, Sum / Summax bit-equal to the
  // single-threaded float32 reference loop, and faster than the atomic scatter-add below, which remains for what the triangle-order
  // kernels do not take (padded rows, SMESH_FUSE=strip, SMESH_ADD_RECORDS=0).
  static const bool records_off = getenv("SMESH_ADD_RECORDS") && atoi(getenv("SMESH_ADD_RECORDS")) == 0;
  const char* min_c_env = getenv("SMESH_ADD_RECORDS_MIN_C");      // (read per call: tests and tools move the threshold at run time)
  const uint32_t records_min_c = min_c_env ? (uint32_t)atoi(min_c_env) : kAddRecordsMinC;
  if (!records_off && C >= records_min_c && a->P > 0 && W <= 65535 && H <= 65535 && smesh_aggregator_can_fuse_triangles(a, a->P)) {
    ImageRecords& rec = a->rec;
    {
      ProfScope prof(ctx, SMESH_PROF_FUSE_HIST);
      SMESH_TRY(image_records_build(ctx, rec, idx, W, H, a->P));
    }
    SMESH_TRY(image_records_pending(ctx, rec, idx, W, H, st));
    const RenderedView rv{rec.frags, rec.big_queue, rec.big_count, idx, probs, weights, W, H};
    SMESH_TRY(smesh_aggregator_fuse_triangles(a, a->P, nullptr, (uint32_t)a->P, &rv, 1));
    if (a->kind == SMESH_AGG_MUL) SMESH_TRY(ensure_acc_d(a));   // (float64 scratch rows of the sparse primitives' terms; all zero between calls)
    SMESH_TRY(image_records_scatter_sparse(ctx, rec, a->kind, idx, probs, weights, W, H, C, a->iew, a->acc, a->acc_lo, a->acc_d, st));
    SMESH_TRY(image_records_clear(ctx, rec, idx, W, H, st));
    smesh_note_fuse(smesh_aggregator_fuse_kernel_name(a, false), "image-records");
    return SMESH_OK;
  }
  smesh_note_fuse(strip_path(C) && a->kind != SMESH_AGG_MUL ? "k_scatter_strip" : "k_scatter_flat", "scatter");

  ScatterArgs args;
  args.idx = idx; args.probs = probs; args.weights = weights; args.pw = nullptr;
  const bool need_hist = a->iew != 0.0f;
  args.count = need_hist ? a->count : nullptr;
  args.acc = a->acc; args.N = N; args.P = (uint32_t)a->P; args.C = C; args.S = a->S; args.iew = a->iew;
  args.W = (uint32_t)W; args.H = (uint32_t)H;
  args.dbg = SMESH_ABL_ENV("SMESH_DBG");
  set_tiling(args);

  // ---- F1 histogram (skipped when the weight does not depend on it) -------------------------
  if (need_hist) {
    ProfScope prof(ctx, SMESH_PROF_FUSE_HIST);
    SMESH_TRY(launch_hist(args, st));
  }

  // ---- F2 scatter-add -------------------------------------------------------------------------
  SMESH_TRY(mul_recentre(a));
  const bool mul = a->kind == SMESH_AGG_MUL;      // Mul: float64 atomics + one fold per touched row (k_scatter_flat_mul)
  if (strip_path(C) && !mul) {
    args.pw = nullptr;
    if (need_hist || weights) {
      SMESH_TRY(a->pw.reserve(N * 4));
      ProfScope prof(ctx, SMESH_PROF_FUSE_HIST);
      hipLaunchKernelGGL(k_pixel_weights, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, args, static_cast<float*>(a->pw.ptr));
      SMESH_HIP(hipGetLastError());
      args.pw = static_cast<const float*>(a->pw.ptr);
    }
    ProfScope prof(ctx, SMESH_PROF_FUSE_SCATTER);
    prof_note(ctx, SMESH_PROF_FUSE_SCATTER, 1, 1);
    switch (a->kind) {
      case SMESH_AGG_SUM:    SMESH_TRY(launch_strip<SMESH_AGG_SUM>(args, ctx->num_cus, st)); break;
      case SMESH_AGG_SUMMAX: SMESH_TRY(launch_strip<SMESH_AGG_SUMMAX>(args, ctx->num_cus, st)); break;
      default:               SMESH_TRY(launch_strip<SMESH_AGG_MUL>(args, ctx->num_cus, st)); break;
    }
  } else {
    SMESH_TRY(a->fb_w.reserve(N * 4));
    SMESH_TRY(a->fb_amax.reserve(N * 4));
    float* wpix = static_cast<float*>(a->fb_w.ptr);
    uint32_t* amax = static_cast<uint32_t*>(a->fb_amax.ptr);
    ProfScope prof(ctx, SMESH_PROF_FUSE_SCATTER);
    prof_note(ctx, SMESH_PROF_FUSE_SCATTER, 1, 1);
    const dim3 g1((uint32_t)div_up(N, 256)), g2((uint32_t)div_up(N * C, 256)), b(256);
    switch (a->kind) {
      case SMESH_AGG_SUM:
        hipLaunchKernelGGL(k_pixel_weight<SMESH_AGG_SUM>, g1, b, 0, st, args, wpix, amax);
        hipLaunchKernelGGL(k_scatter_flat<SMESH_AGG_SUM>, g2, b, 0, st, args, wpix, amax);
        break;
      case SMESH_AGG_SUMMAX:
        hipLaunchKernelGGL(k_pixel_weight<SMESH_AGG_SUMMAX>, g1, b, 0, st, args, wpix, amax);
        hipLaunchKernelGGL(k_scatter_flat<SMESH_AGG_SUMMAX>, g2, b, 0, st, args, wpix, amax);
        break;
      default:
        SMESH_TRY(ensure_acc_d(a));
        hipLaunchKernelGGL(k_pixel_weight<SMESH_AGG_MUL>, g1, b, 0, st, args, wpix, amax);
        hipLaunchKernelGGL(k_scatter_flat_mul, g2, b, 0, st, args, wpix, a->acc_d);
        hipLaunchKernelGGL(k_fold_rows, dim3((uint32_t)div_up(a->P, 256)), b, 0, st, a->acc, a->acc_lo, a->acc_d, args.count, a->P, C, a->S);
        break;
    }
    SMESH_HIP(hipGetLastError());
  }

  // ---- restore the all-zero histogram (the strip kernel does it itself) ---------------------------
  if (need_hist && !(strip_path(C) && !mul)) {
    if (a->P <= 2 * N) {
      SMESH_HIP(hipMemsetAsync(a->count, 0, a->P * 4, st));
    } else {
      hipLaunchKernelGGL(k_hist_clear, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, idx, a->count, N, (uint32_t)a->P);
      SMESH_HIP(hipGetLastError());
    }
  }
  return SMESH_OK;
}

// Contiguous uint32 view of an index image already in device memory (gathers/casts only when needed).
int normalize_idx(DeviceCtx* ctx, Scratch& scratch, const void* d_idx, int idx_dtype, const int64_t is[2],
                  uint64_t W, uint64_t H, const uint32_t** out) {
  const uint64_t N = W * H;
  if (is[0] == (int64_t)H && is[1] == 1 && (idx_dtype == SMESH_IDX_U32 || idx_dtype == SMESH_IDX_I32)) {
    *out = static_cast<const uint32_t*>(d_idx);
    return SMESH_OK;
  }
  SMESH_TRY(scratch.reserve(N * 4));
  uint32_t* o = static_cast<uint32_t*>(scratch.ptr);
  const dim3 g((uint32_t)div_up(N, 256)), b(256);
  hipStream_t st = ctx->stream;
  switch (idx_dtype) {
    case SMESH_IDX_U32: hipLaunchKernelGGL(k_gather_idx<uint32_t>, g, b, 0, st, (const uint32_t*)d_idx, is[0], is[1], o, N, (uint32_t)H); break;
    case SMESH_IDX_I32: hipLaunchKernelGGL(k_gather_idx<int32_t>, g, b, 0, st, (const int32_t*)d_idx, is[0], is[1], o, N, (uint32_t)H); break;
    case SMESH_IDX_U64: hipLaunchKernelGGL(k_gather_idx<uint64_t>, g, b, 0, st, (const uint64_t*)d_idx, is[0], is[1], o, N, (uint32_t)H); break;
    default:            hipLaunchKernelGGL(k_gather_idx<int64_t>, g, b, 0, st, (const int64_t*)d_idx, is[0], is[1], o, N, (uint32_t)H); break;
  }
  SMESH_HIP(hipGetLastError());
  *out = o;
  return SMESH_OK;
}

int check_strides(const int64_t* s, int n, const char* what) {
  if (!s) return fail(SMESH_ERR_INVALID, std::string(what) + ": strides are NULL");
  for (int i = 0; i < n; i++)
    if (s[i] < 0) return fail(SMESH_ERR_INVALID, std::string(what) + ": negative strides are not supported");
  return SMESH_OK;
}

}  // namespace

// Used by raster.hip's smesh_fuse_view.
int smesh_aggregator_add_device_contig(smesh_aggregator* a, const uint32_t* d_idx, const float* d_probs,
                                       const float* d_w, uint64_t W, uint64_t H) {
  const int64_t is[2] = {(int64_t)H, 1};
  const int64_t ps[3] = {(int64_t)(H * a->C), (int64_t)a->C, 1};
  return add_device(a, d_idx, SMESH_IDX_U32, is, d_probs, ps, d_w, is, W, H);
}

// Triangle-order fusion entry used by raster.hip's smesh_fuse_view; see k_fuse_tri.
bool smesh_aggregator_can_fuse_triangles(smesh_aggregator* a, uint64_t F) {
  static const bool off = getenv("SMESH_FUSE") && std::string(getenv("SMESH_FUSE")) == "strip";
  return !off && a->P == F && a->S == a->C && a->C <= 64u * (uint32_t)kSliceAny;   // 64 lanes x kSliceAny classes per row
}

// Largest class count the LDS-block kernel k_fuse_tri takes (a 64-slot instance needs 292 VGPRs and is no faster than
// k_fuse_tri_any: C = 64 0.311 vs 0.292 ms/view; 48 slots: C = 48 0.184 vs 0.252).
static const uint32_t kFuseTriMaxC = getenv("SMESH_FUSE_TRI_MAXC") ? (uint32_t)std::min(48, std::max(1, atoi(getenv("SMESH_FUSE_TRI_MAXC")))) : 48u;   // (experiment knob)

// Which kernel smesh_aggregator_fuse_triangles dispatches for this aggregator (reporting only).
static bool fuse_wide_enabled() {
  static const bool off = getenv("SMESH_FUSE_WIDE") && atoi(getenv("SMESH_FUSE_WIDE")) == 0;
  return !off;
}
const char* smesh_aggregator_fuse_kernel_name(smesh_aggregator* a, bool reordered) {
  if (a->C <= kFuseTriMaxC) return "k_fuse_tri";
  (void)reordered;
  if (fuse_wide_enabled() && a->C >= 128 && a->C <= 1024) return "k_fuse_tri_wide";
  return "k_fuse_tri_any";
}

int smesh_aggregator_max_fused_views(smesh_aggregator* a);
bool smesh_aggregator_fuses_small_views_by_mask(smesh_aggregator* a);
bool smesh_aggregator_takes_strided_probs(smesh_aggregator* a, int64_t ps0, int64_t ps1, int nviews);
bool smesh_aggregator_can_fuse_pair(smesh_aggregator* a) { return smesh_aggregator_max_fused_views(a) >= 2; }

// How many views one k_fuse_tri launch takes for this aggregator: 8 for class counts up to 40 (the per-view state is 3 registers:
// 149 VGPRs at C = 19, 206 at C = 40 -- the occupancy of the two-view instance or one wave less), 2 up to kFuseTriMaxC (the 48-slot
// instance has no registers left), else 1.  Instances exist for 1, 2, 4 and 8 views.
int smesh_aggregator_max_fused_views(smesh_aggregator* a) {
  static const int cap = getenv("SMESH_FUSE_VIEWS") ? std::max(1, atoi(getenv("SMESH_FUSE_VIEWS"))) : 8;
  int m = a->C <= 40u ? 8 : (a->C <= (uint32_t)kFuseTriMaxC ? 2 : 1);
  if (a->C > (uint32_t)kFuseTriMaxC) m = 8;   // k_fuse_tri_any / k_fuse_tri_wide: any count up to eight
  // 41 .. 48: four or eight views go through k_fuse_tri_any (0.124-0.133 vs 0.138-0.146 ms per view at cfg2's geometry), one or two
  // through the 48-slot k_fuse_tri; Mul stays there (the Mul instances of k_fuse_tri_any carry a view's partial sums in double: 220 VGPRs)
  if (a->C > 40u && a->C <= (uint32_t)kFuseTriMaxC && a->kind != SMESH_AGG_MUL) m = 8;
  return std::min(m, cap);
}

// Is EVERY triangle-order launch of this aggregator k_fuse_tri -- whose waves never read the index plane of a view that has neither queued
// triangles nor masks to check (fuse_box by_mask, the per-view verify pass)?  Up to 40 classes yes, whatever the number of views; 41 .. 48 only
// for Mul (two views at most); Sum / Summax there take k_fuse_tri_any + k_fuse_big_any for more than two views, and those scan the planes.
// raster.hip asks before it lets each view of a group decide on its own plane (plane_optional_level).
bool smesh_aggregator_fuses_small_views_by_mask(smesh_aggregator* a) {
  return a->C <= 40u || (a->C <= (uint32_t)kFuseTriMaxC && a->kind == SMESH_AGG_MUL && smesh_aggregator_max_fused_views(a) <= 2);
}

// Can the triangle-order fusion of this aggregator read class vectors in place at element strides (ps0, ps1, 1) -- the (H,W,C)
// output of a network seen as (W,H,C), colorize_cityscapes_mesh.py:65-67 -- instead of a gathered copy?  k_fuse_tri (C <= 48)
// addresses every pixel's vector on its own; the wide-row kernels keep their dense layout.
bool smesh_aggregator_takes_strided_probs(smesh_aggregator* a, int64_t ps0, int64_t ps1, int nviews) {
  static const bool off = getenv("SMESH_STRIDED_PROBS") && atoi(getenv("SMESH_STRIDED_PROBS")) == 0;
  if (off || ps0 <= 0 || ps1 <= 0 || ps0 > 0xFFFFFFFFll || ps1 > 0xFFFFFFFFll) return false;
  return a->C <= kFuseTriMaxC && !(a->C > 40u && nviews > 2);
}

// Fusion by triangle range (smesh_fuse_views_begin / _continue): part `part` of `nparts` covers the triangles at positions
// [*f_lo, *f_hi) of the renderer's order -- whole 64-triangle blocks, so that every kernel's workgroups (64, 32 .. 1 triangles each) and
// the 64-row accumulator blocks of k_fuse_tri fall inside one part.  Without a re-ordered mesh these are also the accumulator rows
// the part leaves final.
void smesh_fuse_part_rows(uint64_t F, int part, int nparts, uint64_t* f_lo, uint64_t* f_hi) {
  const uint64_t B = div_up(F, 64);
  *f_lo = std::min(F, 64 * (B * (uint64_t)part / (uint64_t)nparts));
  *f_hi = std::min(F, 64 * (B * (uint64_t)(part + 1) / (uint64_t)nparts));
}

// Sum rows of 128 .. 255 classes: the pixel-list kernel (k_fuse_tri_wide_list; SMESH_WIDE_LIST=0: k_fuse_tri_wide as until round 5).
// false: not launched.  Where it is used and where not is measured (cfg2's mesh and resolution, eight views per call, ms per view, old /
// list kernel; profiles/r06_wide_rows_sweep.txt): Sum C = 128 0.405 / 0.381, 150 0.467 / 0.406, 192 0.438 / 0.407, 240 0.454 / 0.439 --
// 256 (rows of whole aligned lines) 0.371 / 0.454, 300 0.528 / 0.558, 1024 1.361 / 1.441; Summax (its arg-max needs registers the ring
// has taken: spills) 150 0.638 / 0.694.  cfg5 (20 M sub-pixel triangles, C = 150): 1 651 / 1 125 us per view.
template <int KIND>
static bool launch_fuse_wide_list(int wide_chunks, uint32_t C, dim3 wgrid, hipStream_t st, const TriFuseArgs& t, const TriViews<8>& tv, int nviews) {
  static const bool off = getenv("SMESH_WIDE_LIST") && atoi(getenv("SMESH_WIDE_LIST")) == 0;
  if constexpr (KIND != SMESH_AGG_SUM) return false;
  else {
    if (off || wide_chunks != 1 || C >= 256u) return false;
    hipLaunchKernelGGL((k_fuse_tri_wide_list<KIND, 1>), wgrid, dim3(kWave), 0, st, t, tv, nviews);
    return true;
  }
}

// `nviews` = 1, 2, 4 or 8 (smesh_aggregator_max_fused_views): views[0], views[1] ... of the same renderer in one launch.
// `part` / `nparts`: only the triangles of smesh_fuse_part_rows(F, part, nparts) -- the queued medium triangles (fuse_mid_entries, float
// atomics) all go with part 0, the queued big ones with the part their position falls into.
int smesh_aggregator_fuse_triangles(smesh_aggregator* a, uint64_t F, const uint32_t* prim_id, uint32_t big_capacity,
                                    const RenderedView* views, int nviews, int part, int nparts) {
  DeviceCtx* ctx = a->ctx;
  hipStream_t st = ctx->stream;
  if (F == 0) return SMESH_OK;
  if (nparts < 1 || part < 0 || part >= nparts) return fail(SMESH_ERR_INVALID, "fuse_triangles: bad part");
  SMESH_TRY(smesh_aggregator_refuse_scattered(a, "fuse_view()"));
  uint64_t f_lo, f_hi;
  smesh_fuse_part_rows(F, part, nparts, &f_lo, &f_hi);
  if (f_lo >= f_hi && part != 0) return SMESH_OK;   // (part 0 still owns the medium triangles)
  if ((nviews != 1 && nviews != 2 && nviews != 4 && nviews != 8) || nviews > smesh_aggregator_max_fused_views(a))
    return fail(SMESH_ERR_INVALID, "fuse_triangles: unsupported view count");
  const uint64_t N = views[0].W * views[0].H;
  TriFuseArgs t;
  TriViews<8> tv;
  for (int v = 0; v < 8; v++) {
    const RenderedView& rv = views[v < nviews ? v : 0];
    tv.v[v] = TriView{rv.frags, rv.idx, rv.probs, rv.weights, rv.big_queue, rv.big_len, (uint32_t)rv.W, (uint32_t)rv.H,
                      rv.ps0 ? (uint32_t)rv.ps0 : (uint32_t)(rv.H * a->C), rv.ps1 ? (uint32_t)rv.ps1 : a->C};
  }
  for (int v = 0; v < nviews; v++)
    if ((views[v].ps0 || views[v].ps1) && !smesh_aggregator_takes_strided_probs(a, views[v].ps0, views[v].ps1, nviews))
      return fail(SMESH_ERR_INVALID, "fuse_triangles: strided class vectors need the narrow-row kernel");
  for (int v = 0; v < 1; v++) {
    const RenderedView& rv = views[0];
    TriFuseArgs& x = t;
    x.frags = rv.frags; x.idx = rv.idx; x.probs = rv.probs; x.weights = rv.weights; x.acc = a->acc; x.acc_lo = a->acc_lo; x.F = F; x.C = a->C;
    x.W = (uint32_t)rv.W; x.H = (uint32_t)rv.H; x.iew = a->iew; x.big_queue = rv.big_queue; x.big_len = rv.big_len;
    x.ps0 = tv.v[0].ps0; x.ps1 = tv.v[0].ps1;
    x.big_capacity = big_capacity;
    x.tri_blocks = (uint32_t)div_up(F, kWave);
    { static const int fdbg = SMESH_ABL_ENV("SMESH_FDBG"); x.dbg = fdbg; }
    x.tex_first = nullptr; x.tex_res = nullptr; x.count = nullptr; x.acc_d = nullptr;
    x.prim_id = prim_id;
    x.mid = 0;
    x.lds_pad = 0u;
    x.blk_first = (uint32_t)(f_lo / kWave);
    x.tri_blocks = (uint32_t)div_up(f_hi - f_lo, kWave);
    x.f_lo = (uint32_t)f_lo; x.f_hi = (uint32_t)f_hi;
  }
  // k_fuse_tri (row in registers; the wave's 64-row block staged through LDS unless the mesh was re-ordered) takes C <= 48: exact instances for 5 / 13 / 19 / 20 / 21 / 40, run-time-C instances sized 8 .. 48 for the rest
  // (tri_ct 41 = the run-time instance with 40 slots).
  int tri_ct = 0;
  if (a->C <= kFuseTriMaxC && !(a->C > 40u && nviews > 2)) {
    if (a->C == 5 || a->C == 13 || a->C == 19 || a->C == 20 || a->C == 21 || a->C == 40) tri_ct = (int)a->C;   // common label sets
    else tri_ct = a->C <= 8 ? 8 : a->C <= 16 ? 16 : a->C <= 24 ? 24 : a->C <= 32 ? 32 : a->C <= 40 ? 41 : 48;
  }
  const bool specialised = tri_ct != 0;
  float* pw = nullptr;
  uint32_t* amax = nullptr;
  uint64_t scratch_stride = N;   // per-view scratch images of the big-triangle waves
  int G = 1;
  const int wide_chunks = (fuse_wide_enabled() && a->C >= 128 && a->C <= 1024) ? (a->C <= 256 ? 1 : a->C <= 512 ? 2 : 4) : 0;   // k_fuse_tri_wide
  // rows (and their next pixels) in flight per wave: 2 (Sum / Summax at any width: cfg5 views/s with 2 rows and 8 waves per SIMD 509, 4 rows
  // and 5 waves 496, 8 rows -- 284 registers, one wave -- 130); Mul with rows of up to 256 classes: 4 (measured in round 4); Mul beyond:
  // 2 (its per-view partial sums in double over two chunks are 64 more registers -- ADVICE r5: the dispatch used to pick <K, 2, 4>).
  // SMESH_WIDE_B = 2 / 4 / 8 forces (experiment knob; other values are ignored).
  static const int wide_b_env = [] { const int v = getenv("SMESH_WIDE_B") ? atoi(getenv("SMESH_WIDE_B")) : 0; return (v == 2 || v == 4 || v == 8) ? v : 0; }();
  const int wide_b = wide_b_env ? wide_b_env : ((a->kind == SMESH_AGG_MUL && wide_chunks == 1) ? 4 : 2);
  if (!specialised) {
    while ((((a->C + G - 1) / G + 3u) & ~3u) > (uint32_t)kSliceAny) G *= 2;   // lanes per accumulator row (can_fuse_triangles: G <= 64)
    // the big-triangle waves park per-pixel weights (and arg-max) here
    for (int v = 0; v < nviews; v++) scratch_stride = std::max<uint64_t>(scratch_stride, views[v].W * views[v].H);
    SMESH_TRY(a->pw.reserve(scratch_stride * (uint64_t)nviews * 4));
    pw = static_cast<float*>(a->pw.ptr);
    if (a->kind == SMESH_AGG_SUMMAX) {
      SMESH_TRY(a->fb_amax.reserve(scratch_stride * (uint64_t)nviews * 4));
      amax = static_cast<uint32_t*>(a->fb_amax.ptr);
    }
    const uint64_t tpb = wide_chunks ? (uint64_t)kWave : (uint64_t)(kWave / G);   // triangles per workgroup
    t.blk_first = (uint32_t)(f_lo / tpb);
    t.tri_blocks = (uint32_t)div_up(f_hi - f_lo, tpb);
  }
  static const uint32_t big_per_cu = getenv("SMESH_BIG_WAVES") ? (uint32_t)std::max(1, atoi(getenv("SMESH_BIG_WAVES"))) : 16u;
  // one wave per queued big triangle at a time; they exit at once if the queue is empty.  A launch over one of `nparts` triangle
  // ranges takes its share of them (every one of its waves still walks the whole queue and keeps the triangles of its range).
  bool no_big = true;     // every view of the launch PROVEN free of triangles with a box over 8 x 8 (RenderedView::no_big): no tail waves at all
  for (int v = 0; v < nviews; v++) no_big = no_big && views[v].no_big;
  const uint32_t big_waves = no_big ? 0u : std::max((uint32_t)std::max(1, ctx->num_cus), big_per_cu * (uint32_t)std::max(1, ctx->num_cus) / (uint32_t)nparts);
  // Medium triangles (a box over 8 x 8 of at most kMidBox pixels; k_fuse_tri's class counts, Sum / Summax -- Mul's (hi, lo) rows cannot
  // take atomics): their queue entries go to further one-wave workgroups at the end of the SAME launch (fuse_mid_entries; a launch of
  // its own until round 4), gone at once when the lists are empty.  They add with float atomics, so the main waves leave the rows of
  // such triangles alone (`mixed` lanes) and the tail waves skip what t.mid hands over.  With fusion by triangle range all of them go
  // with part 0.
  uint32_t mid_waves = 0;
  if (specialised) {
    static const int mid_mode = getenv("SMESH_FUSE_MID") ? atoi(getenv("SMESH_FUSE_MID")) : 1;   // 0: the tail waves take the medium triangles too
    bool listed = true;                              // (records rebuilt from a foreign image carry no list of medium primitives)
    for (int v = 0; v < nviews; v++) listed = listed && views[v].mid_queue;
    if (mid_mode && listed && a->kind != SMESH_AGG_MUL) {
      t.mid = 1;
      static const uint32_t mid_per_cu = getenv("SMESH_MID_WAVES") ? (uint32_t)std::max(1, atoi(getenv("SMESH_MID_WAVES"))) : 32u;   // (90 000 triangles at 1080p, ms per view: 8 -> 0.111, 16 -> 0.108, 32 -> 0.104, 64 -> 0.103; cfg2 0.062 throughout)
      if (part == 0 && !no_big) mid_waves = mid_per_cu * (uint32_t)std::max(1, ctx->num_cus);
    }
  }
  t.big_blocks = big_waves;
  {
    // Group pipeline: the rasteriser of the next group runs beside this launch.  On a finely tessellated mesh (every view `fine`: the
    // queues of medium and big triangles empty or nearly) the main waves are all there is, five of them per CU saturate the memory
    // system, and an LDS pad that caps them there leaves the registers to the rasteriser's waves.  Where the launch has medium or
    // big triangles to fuse (their waves are workgroups of the same launch, 16 - 32 per CU wanted) the cap starves them: 90 000
    // triangles at 1080p 0.102 -> 0.112 ms per view with it.
    bool fine = true;
    for (int v = 0; v < nviews; v++) fine = fine && views[v].fine;
    t.lds_pad = (fine && opt_group_pipeline() && nparts == 1) ? 24576u : 0u;
  }
  const dim3 grid(t.tri_blocks + big_waves + mid_waves), block(kWave);
  const dim3 tgrid(t.tri_blocks), bgrid(big_waves);                        // any-C paths: big triangles in a second launch
  {   // k_fuse_tri_wide: the triangle blocks dealt to the XCDs in runs of SMESH_WIDE_XCD consecutive blocks (default 0: dispatch order).
      // Measured at cfg5 (round 6): one run per XCD -- an eighth of the mesh each -- 506 -> 268 views/s: the eighths are not equally
      // visible in a view and the launch waits for the XCDs that hold the visible ones.
    static const uint32_t wide_xcd = getenv("SMESH_WIDE_XCD") ? (uint32_t)std::max(0, atoi(getenv("SMESH_WIDE_XCD"))) : 0u;
    t.xcd_chunk = (wide_chunks && t.tri_blocks >= 64u) ? wide_xcd : 0u;
  }
  const dim3 wgrid(t.xcd_chunk ? 8u * (uint32_t)div_up(t.tri_blocks, 8u * t.xcd_chunk) * t.xcd_chunk : t.tri_blocks);
  if (!specialised && part == 0) SMESH_TRY(mul_recentre(a));   // (a pass over ALL rows: never beside the exchange of a finished range)
  {
    ProfScope prof(ctx, SMESH_PROF_FUSE_SCATTER);
    prof_note(ctx, SMESH_PROF_FUSE_SCATTER, 1, part == 0 ? (uint64_t)nviews : 0);
#define SMESH_FA(K)                                                                            \
    switch (G) {                                                                               \
      case 1:  hipLaunchKernelGGL((k_fuse_tri_any<K, 1>), tgrid, block, 0, st, t, tv, nviews); break;  \
      case 2:  hipLaunchKernelGGL((k_fuse_tri_any<K, 2>), tgrid, block, 0, st, t, tv, nviews); break;  \
      case 4:  hipLaunchKernelGGL((k_fuse_tri_any<K, 4>), tgrid, block, 0, st, t, tv, nviews); break;  \
      case 8:  hipLaunchKernelGGL((k_fuse_tri_any<K, 8>), tgrid, block, 0, st, t, tv, nviews); break;  \
      case 16: hipLaunchKernelGGL((k_fuse_tri_any<K, 16>), tgrid, block, 0, st, t, tv, nviews); break; \
      case 32: hipLaunchKernelGGL((k_fuse_tri_any<K, 32>), tgrid, block, 0, st, t, tv, nviews); break; \
      default: hipLaunchKernelGGL((k_fuse_tri_any<K, 64>), tgrid, block, 0, st, t, tv, nviews); break; \
    }
#define SMESH_FW(K)                                                                            \
    switch (wide_chunks) {                                                                     \
      case 1:  if (wide_b == 8) hipLaunchKernelGGL((k_fuse_tri_wide<K, 1, 8>), wgrid, block, 0, st, t, tv, nviews);   \
               else if (wide_b == 2) hipLaunchKernelGGL((k_fuse_tri_wide<K, 1, 2>), wgrid, block, 0, st, t, tv, nviews); \
               else hipLaunchKernelGGL((k_fuse_tri_wide<K, 1, 4>), wgrid, block, 0, st, t, tv, nviews); break; \
      case 2:  if (wide_b >= 4) hipLaunchKernelGGL((k_fuse_tri_wide<K, 2, 4>), wgrid, block, 0, st, t, tv, nviews);   \
               else hipLaunchKernelGGL((k_fuse_tri_wide<K, 2, 2>), wgrid, block, 0, st, t, tv, nviews); break; \
      default: hipLaunchKernelGGL((k_fuse_tri_wide<K, 4, 2>), wgrid, block, 0, st, t, tv, nviews); break; \
    }
#define SMESH_FT(K)                                                                           \
    switch (tri_ct) {                                                                         \
      case 5:  hipLaunchKernelGGL((k_fuse_tri<5, K, true, 1>), grid, block, 0, st, t, tv1); break;     \
      case 13: hipLaunchKernelGGL((k_fuse_tri<13, K, true, 1>), grid, block, 0, st, t, tv1); break;    \
      case 19: hipLaunchKernelGGL((k_fuse_tri<19, K, true, 1>), grid, block, 0, st, t, tv1); break;    \
      case 20: hipLaunchKernelGGL((k_fuse_tri<20, K, true, 1>), grid, block, 0, st, t, tv1); break;    \
      case 21: hipLaunchKernelGGL((k_fuse_tri<21, K, true, 1>), grid, block, 0, st, t, tv1); break;    \
      case 40: hipLaunchKernelGGL((k_fuse_tri<40, K, true, 1>), grid, block, 0, st, t, tv1); break;    \
      case 8:  hipLaunchKernelGGL((k_fuse_tri<8, K, false, 1>), grid, block, 0, st, t, tv1); break;    \
      case 16: hipLaunchKernelGGL((k_fuse_tri<16, K, false, 1>), grid, block, 0, st, t, tv1); break;   \
      case 24: hipLaunchKernelGGL((k_fuse_tri<24, K, false, 1>), grid, block, 0, st, t, tv1); break;   \
      case 32: hipLaunchKernelGGL((k_fuse_tri<32, K, false, 1>), grid, block, 0, st, t, tv1); break;   \
      case 41: hipLaunchKernelGGL((k_fuse_tri<40, K, false, 1>), grid, block, 0, st, t, tv1); break;   \
      case 48: hipLaunchKernelGGL((k_fuse_tri<48, K, false, 1>), grid, block, 0, st, t, tv1); break;   \
      default:                                                                                \
        if (!t.tri_blocks) { }                                                                \
        else if (wide_chunks && launch_fuse_wide_list<K>(wide_chunks, a->C, wgrid, st, t, tv, nviews)) { }   \
        else if (wide_chunks) { SMESH_FW(K); } else { SMESH_FA(K); }                          \
        if (!no_big) hipLaunchKernelGGL((k_fuse_big_any<K>), bgrid, block, 0, st, t, tv, nviews, pw, amax, scratch_stride); \
        break;                                                                                \
    }
    TriViews<1> tv1;
    tv1.v[0] = tv.v[0];
    if (nviews >= 2 && specialised) {   // the several-view instances of k_fuse_tri live in fusion_pair.hip / fusion_multi*.hip
      smesh_launch_fuse_tri_multi(a->kind, tri_ct, nviews, grid, st, t, tv);
    } else
    switch (a->kind) {
      case SMESH_AGG_SUM: SMESH_FT(SMESH_AGG_SUM); break;
      case SMESH_AGG_SUMMAX: SMESH_FT(SMESH_AGG_SUMMAX); break;
      default: SMESH_FT(SMESH_AGG_MUL); break;
    }
#undef SMESH_FT
#undef SMESH_FW
#undef SMESH_FA
  }
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

bool smesh_aggregator_can_fuse_texels(smesh_aggregator* a, uint64_t P) {
  static const bool off = getenv("SMESH_FUSE") && std::string(getenv("SMESH_FUSE")) == "strip";
  return !off && a->P == P && a->S == a->C && a->C <= (uint32_t)kSlice;
}

// Mul: the double scratch of k_fuse_texel_big, on first use.
static int ensure_acc_d(smesh_aggregator* a) {
  if (a->kind != SMESH_AGG_MUL || a->acc_d) return SMESH_OK;
  const size_t bytes = (size_t)a->P * a->C * sizeof(double);
  SMESH_HIP(dev_malloc(reinterpret_cast<void**>(&a->acc_d), bytes ? bytes : 16));
  SMESH_HIP(hipMemsetAsync(a->acc_d, 0, bytes, a->ctx->stream));
  return SMESH_OK;
}

int smesh_aggregator_fuse_texels(smesh_aggregator* a, const TriFrag* frags, const uint8_t* kinds, uint64_t F, const uint32_t* tex_first,
                                 const uint32_t* tex_res, const uint32_t* big_queue, const uint32_t* big_len,
                                 uint32_t big_capacity, const uint32_t* d_idx, const float* d_probs, const float* d_w, uint64_t H) {
  DeviceCtx* ctx = a->ctx;
  hipStream_t st = ctx->stream;
  if (F == 0) return SMESH_OK;
  TriFuseArgs t;
  t.frags = frags; t.idx = d_idx; t.probs = d_probs; t.weights = d_w; t.acc = a->acc; t.acc_lo = a->acc_lo; t.F = F; t.C = a->C;
  t.H = (uint32_t)H; t.iew = a->iew; t.big_queue = big_queue; t.big_len = big_len; t.big_capacity = big_capacity;
  t.tri_blocks = (uint32_t)div_up(F, kWave);
  t.big_blocks = 0;
  t.blk_first = 0u; t.f_lo = 0u; t.f_hi = (uint32_t)F;
  t.dbg = 0; t.prim_id = nullptr; t.lds_pad = 0u;
  SMESH_TRY(ensure_acc_d(a));
  t.tex_first = tex_first; t.tex_res = tex_res; t.count = a->count; t.acc_d = a->acc_d; t.tex_kinds = kinds;
  const dim3 tgrid((uint32_t)div_up(F, kTexelBlock)), bgrid(12u * (uint32_t)std::max(1, ctx->num_cus)), block(kWave), tblock(kTexelBlock);
  // (no pass over the whole accumulator for Mul here: every texel kernel re-centres the rows it touches)
  {
    ProfScope prof(ctx, SMESH_PROF_FUSE_SCATTER);
    prof_note(ctx, SMESH_PROF_FUSE_SCATTER, 1, 1);
    switch (a->kind) {
      case SMESH_AGG_SUM:
        hipLaunchKernelGGL((k_fuse_texel<SMESH_AGG_SUM>), tgrid, tblock, 0, st, t);
        hipLaunchKernelGGL((k_fuse_texel_big<SMESH_AGG_SUM>), bgrid, block, 0, st, t);
        break;
      case SMESH_AGG_SUMMAX:
        hipLaunchKernelGGL((k_fuse_texel<SMESH_AGG_SUMMAX>), tgrid, tblock, 0, st, t);
        hipLaunchKernelGGL((k_fuse_texel_big<SMESH_AGG_SUMMAX>), bgrid, block, 0, st, t);
        break;
      default:
        hipLaunchKernelGGL((k_fuse_texel<SMESH_AGG_MUL>), tgrid, tblock, 0, st, t);
        hipLaunchKernelGGL((k_fuse_texel_big<SMESH_AGG_MUL>), bgrid, block, 0, st, t);
        break;
    }
  }
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

// The same for up to eight views of the renderer in one launch (k_fuse_texel_multi), the big triangles of each view in a launch of
// their own behind it (their scratch histogram serves one view at a time).
int smesh_aggregator_fuse_texels_multi(smesh_aggregator* a, uint64_t F, const uint32_t* tex_first, const uint32_t* tex_res, uint32_t big_capacity,
                                       const RenderedView* views, int nviews) {
  DeviceCtx* ctx = a->ctx;
  hipStream_t st = ctx->stream;
  if (F == 0 || nviews <= 0) return SMESH_OK;
  if (nviews > 8) return fail(SMESH_ERR_INVALID, "fuse_texels_multi: at most eight views per launch");
  TriViews<8> tv;
  for (int v = 0; v < 8; v++) {
    const RenderedView& rv = views[v < nviews ? v : 0];
    tv.v[v] = TriView{rv.frags, rv.idx, rv.probs, rv.weights, rv.big_queue, rv.big_len, (uint32_t)rv.W, (uint32_t)rv.H, 0u, 0u, rv.kinds};
  }
  TriFuseArgs t;
  t.frags = views[0].frags; t.idx = views[0].idx; t.probs = views[0].probs; t.weights = views[0].weights;
  t.acc = a->acc; t.acc_lo = a->acc_lo; t.F = F; t.C = a->C; t.W = (uint32_t)views[0].W; t.H = (uint32_t)views[0].H; t.iew = a->iew;
  t.big_queue = views[0].big_queue; t.big_len = views[0].big_len; t.big_capacity = big_capacity;
  t.tri_blocks = (uint32_t)div_up(F, kWave);
  t.big_blocks = 0;
  t.blk_first = 0u; t.f_lo = 0u; t.f_hi = (uint32_t)F;
  t.dbg = 0; t.prim_id = nullptr; t.lds_pad = 0u;
  SMESH_TRY(ensure_acc_d(a));
  t.tex_first = tex_first; t.tex_res = tex_res; t.count = a->count; t.acc_d = a->acc_d; t.tex_kinds = nullptr;
  const dim3 tgrid((uint32_t)div_up(F, kTexelBlock)), bgrid(12u * (uint32_t)std::max(1, ctx->num_cus)), block(kWave), tblock(kTexelBlock);
  // (no pass over the whole accumulator for Mul here: every texel kernel re-centres the rows it touches)
  {
    ProfScope prof(ctx, SMESH_PROF_FUSE_SCATTER);
    prof_note(ctx, SMESH_PROF_FUSE_SCATTER, 1, (uint64_t)nviews);
    switch (a->kind) {
      case SMESH_AGG_SUM:    hipLaunchKernelGGL((k_fuse_texel_multi<SMESH_AGG_SUM>), tgrid, tblock, 0, st, t, tv, nviews); break;
      case SMESH_AGG_SUMMAX: hipLaunchKernelGGL((k_fuse_texel_multi<SMESH_AGG_SUMMAX>), tgrid, tblock, 0, st, t, tv, nviews); break;
      default:               hipLaunchKernelGGL((k_fuse_texel_multi<SMESH_AGG_MUL>), tgrid, tblock, 0, st, t, tv, nviews); break;
    }
    for (int v = 0; v < nviews; v++) {
      if (views[v].no_big) continue;   // (proven on the host: this view's queue of big triangles is empty -- eight empty launches per cfg4 group)
      TriFuseArgs x = t;
      x.frags = views[v].frags; x.idx = views[v].idx; x.probs = views[v].probs; x.weights = views[v].weights;
      x.W = (uint32_t)views[v].W; x.H = (uint32_t)views[v].H; x.big_queue = views[v].big_queue; x.big_len = views[v].big_len;
      switch (a->kind) {
        case SMESH_AGG_SUM:    hipLaunchKernelGGL((k_fuse_texel_big<SMESH_AGG_SUM>), bgrid, block, 0, st, x); break;
        case SMESH_AGG_SUMMAX: hipLaunchKernelGGL((k_fuse_texel_big<SMESH_AGG_SUMMAX>), bgrid, block, 0, st, x); break;
        default:               hipLaunchKernelGGL((k_fuse_texel_big<SMESH_AGG_MUL>), bgrid, block, 0, st, x); break;
      }
    }
  }
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

DeviceCtx* smesh_aggregator_ctx(smesh_aggregator* a) { return a->ctx; }
// The accumulator as one float32 buffer on the library stream (Mul: the (hi, lo) pair folded into the hi plane first).
// A failure (the Mul fold could not be launched) is reported as such: a rank that silently skipped the collective would leave its
// peers blocked inside it.  `row_stride` = floats per accumulator row, `rows` = P.
int smesh_aggregator_acc(smesh_aggregator* a, float** acc, uint64_t* num_floats, uint32_t* row_stride, uint64_t* rows) {
  if (acc) *acc = nullptr;
  if (num_floats) *num_floats = 0;
  SMESH_TRY(smesh_aggregator_refuse_scattered(a, "exchange"));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  SMESH_TRY(mul_normalise(a, true));
  if (acc) *acc = a->acc;
  if (num_floats) *num_floats = a->P * a->S;
  if (row_stride) *row_stride = a->S;
  if (rows) *rows = a->P;
  return SMESH_OK;
}
uint32_t smesh_aggregator_classes(smesh_aggregator* a) { return a->C; }
uint64_t smesh_aggregator_primitives(smesh_aggregator* a) { return a->P; }

// The main stream waits (on the device, not the host) for the collectives that smesh_allreduce_rows put on the exchange stream.
// Called by every entry point that reads or writes the accumulator on the main stream, with the aggregator and context locked.
int smesh_aggregator_join_exchange(smesh_aggregator* a) {
  if (!a->xchg_pending) return SMESH_OK;
  SMESH_HIP(hipStreamWaitEvent(a->ctx->stream, a->ev_xchg, 0));
  a->xchg_pending = false;
  return SMESH_OK;
}

// A reduce-scatter leaves the rows outside [owned_lo, owned_hi) holding this rank's partial sums only.
int smesh_aggregator_refuse_scattered(smesh_aggregator* a, const char* what) {
  if (!a->scattered) return SMESH_OK;
  return fail(SMESH_ERR_INVALID, std::string(what) + ": the accumulator was reduce-scattered -- only get_rows() inside the owned rows and "
                                                     "reset() are meaningful until reset()");
}
void smesh_aggregator_mark_scattered(smesh_aggregator* a, uint64_t lo, uint64_t hi) { a->scattered = true; a->owned_lo = lo; a->owned_hi = hi; }

namespace {
// Mul: the exact value of an accumulator element is hi + lo (two float32 planes, fuse_tri.inc.hpp "Mul state").  Across ranks the
// pairs are summed in float64: a float32 all-reduce of a folded hi plane loses the difference between a row's leading classes to the
// ulp of the sums (get() was off by up to 2e-4 relative with eight ranks; 1e-5 is the bar).
__global__ void k_mul_rows_to_f64(const float* __restrict__ hi, const float* __restrict__ lo, double* __restrict__ out, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (double)hi[i] + (double)lo[i];
}
__global__ void k_mul_rows_from_f64(const double* __restrict__ in, float* __restrict__ hi, float* __restrict__ lo, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = in[i];
  const float h = (float)v;
  hi[i] = h;
  lo[i] = (h > -INFINITY && h < INFINITY) ? (float)(v - (double)h) : 0.0f;   // (-inf / NaN: no remainder, as mul_fold)
}
}  // namespace

// What a sum over ranks of rows [lo, hi) operates on, prepared on stream `st`: Sum / Summax -- the float32 rows in place; Mul -- a
// float64 image of the (hi, lo) pairs in the aggregator's staging buffer (*is_f64 = 1), written back by smesh_aggregator_exchange_end.
int smesh_aggregator_exchange_begin(smesh_aggregator* a, uint64_t lo, uint64_t hi, hipStream_t st, void** buf, uint64_t* count, int* is_f64) {
  *buf = nullptr; *count = 0; *is_f64 = 0;
  if (lo > hi || hi > a->P) return fail(SMESH_ERR_INVALID, "bad row range");
  const uint64_t n = (hi - lo) * a->S;
  if (n == 0) return SMESH_OK;
  *count = n;
  if (a->kind != SMESH_AGG_MUL) { *buf = a->acc + lo * a->S; return SMESH_OK; }
  if (a->xchg_stage.bytes < n * 8) {
    SMESH_HIP(hipStreamSynchronize(st));   // growing the staging buffer frees the old one: the previous range's collective must be through
    SMESH_TRY(a->xchg_stage.reserve(n * 8));
  }
  hipLaunchKernelGGL(k_mul_rows_to_f64, dim3((uint32_t)div_up(n, 256)), dim3(256), 0, st, a->acc + lo * a->S, a->acc_lo + lo * a->S,
                     static_cast<double*>(a->xchg_stage.ptr), n);
  SMESH_HIP(hipGetLastError());
  *buf = a->xchg_stage.ptr; *is_f64 = 1;
  return SMESH_OK;
}
int smesh_aggregator_exchange_end(smesh_aggregator* a, uint64_t lo, uint64_t hi, hipStream_t st) {
  const uint64_t n = (hi - lo) * a->S;
  if (n == 0 || a->kind != SMESH_AGG_MUL) return SMESH_OK;
  hipLaunchKernelGGL(k_mul_rows_from_f64, dim3((uint32_t)div_up(n, 256)), dim3(256), 0, st, static_cast<const double*>(a->xchg_stage.ptr),
                     a->acc + lo * a->S, a->acc_lo + lo * a->S, n);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}
// The events that hand a row range from the main stream to the exchange stream and back (comm.cpp: smesh_allreduce_rows).
int smesh_aggregator_exchange_events(smesh_aggregator* a, hipEvent_t* ev_part, hipEvent_t* ev_xchg) {
  if (!a->ev_part) SMESH_HIP(hipEventCreateWithFlags(&a->ev_part, hipEventDisableTiming));
  if (!a->ev_xchg) SMESH_HIP(hipEventCreateWithFlags(&a->ev_xchg, hipEventDisableTiming));
  *ev_part = a->ev_part; *ev_xchg = a->ev_xchg;
  return SMESH_OK;
}
void smesh_aggregator_exchange_pending(smesh_aggregator* a) { a->xchg_pending = true; }
std::mutex& smesh_aggregator_mutex(smesh_aggregator* a) { return a->mu; }
// Contiguous (W,H,C) copy, in the aggregator's scratch, of class vectors that sit in DEVICE memory with other (non-negative) strides
// -- a (H,W,C) tensor seen as (W,H,C), the view a framework's transpose / permute returns -- or at an address the 16-byte loads of
// the kernels cannot take.  On the library's stream; *out = d_probs itself when it is dense already.  (raster.hip: add_rendered.)
int smesh_aggregator_dense_probs(smesh_aggregator* a, const float* d_probs, const int64_t ps[3], uint64_t W, uint64_t H, const float** out) {
  const uint64_t N = W * H;
  const uint32_t C = a->C;
  *out = d_probs;
  if (ps[0] == (int64_t)(H * C) && ps[1] == (int64_t)C && ps[2] == 1 && !(reinterpret_cast<uintptr_t>(d_probs) & 15)) return SMESH_OK;
  SMESH_TRY(check_strides(ps, 3, "probs"));
  SMESH_TRY(a->nm_probs.reserve(N * C * 4));
  launch_gather_probs(d_probs, ps, static_cast<float*>(a->nm_probs.ptr), W, H, C, a->ctx->stream);
  SMESH_HIP(hipGetLastError());
  *out = static_cast<const float*>(a->nm_probs.ptr);
  return SMESH_OK;
}

Scratch& smesh_aggregator_stage_probs(smesh_aggregator* a) { return a->st_probs; }
Scratch& smesh_aggregator_stage_w(smesh_aggregator* a) { return a->st_w; }

extern "C" {

int smesh_aggregator_create(uint64_t P, uint32_t C, int kind, float iew, int device, smesh_aggregator_t** out) {
  if (!out) return fail(SMESH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (C == 0) return fail(SMESH_ERR_INVALID, "classes must be > 0");
  if (C > 65535) return fail(SMESH_ERR_INVALID, "classes must be <= 65535");
  if (kind < 0 || kind > 2) return fail(SMESH_ERR_INVALID, "unknown aggregator kind");
  if (P >= 0xFFFFFFFFull) return fail(SMESH_ERR_INVALID, "too many primitives for uint32 indices");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(device));
  auto* a = new (std::nothrow) smesh_aggregator();
  if (!a) return fail(SMESH_ERR_RUNTIME, "out of memory");
  a->ctx = ctx; a->P = P; a->C = C; a->S = row_stride(C); a->kind = kind; a->iew = iew;
  const size_t acc_bytes = (size_t)P * a->S * 4;
  hipError_t e = dev_malloc(reinterpret_cast<void**>(&a->acc), acc_bytes ? acc_bytes : 16);
  if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&a->count), P ? P * 4 : 16);
  if (e == hipSuccess) e = hipMemsetAsync(a->acc, 0, acc_bytes, ctx->stream);   // Sum/Summax: 0; Mul: log 1 = 0
  if (e == hipSuccess && kind == SMESH_AGG_MUL) e = dev_malloc(reinterpret_cast<void**>(&a->acc_lo), acc_bytes ? acc_bytes : 16);
  if (e == hipSuccess && kind == SMESH_AGG_MUL) e = hipMemsetAsync(a->acc_lo, 0, acc_bytes, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(a->count, 0, P * 4, ctx->stream);
  if (e != hipSuccess) {
    if (a->acc) (void)dev_free(a->acc);
    if (a->acc_lo) (void)dev_free(a->acc_lo);
    if (a->count) (void)dev_free(a->count);
    delete a;
    return fail_hip(e, "aggregator allocation", __FILE__, __LINE__);
  }
  if (acc_bytes >= (4u << 20)) copy_ring_prepare(ctx);
  *out = a;
  return SMESH_OK;
}

int smesh_aggregator_destroy(smesh_aggregator_t* a) {
  if (!a) return SMESH_OK;
  (void)hipSetDevice(a->ctx->device);
  (void)hipStreamSynchronize(a->ctx->exchange_stream);   // (a row exchange may still be reading or writing the accumulator)
  (void)hipStreamSynchronize(a->ctx->stream);
  (void)dev_free(a->acc);
  if (a->acc_lo) (void)dev_free(a->acc_lo);
  if (a->acc_d) (void)dev_free(a->acc_d);
  (void)dev_free(a->count);
  if (a->ev_staged) (void)hipEventDestroy(a->ev_staged);
  if (a->ev_part) (void)hipEventDestroy(a->ev_part);
  if (a->ev_xchg) (void)hipEventDestroy(a->ev_xchg);
  a->xchg_stage.release();
  for (Scratch* s : {&a->st_idx, &a->st_probs, &a->st_w, &a->nm_idx, &a->nm_probs, &a->nm_w, &a->fb_w, &a->fb_amax, &a->pw, &a->out_tmp})
    s->release();
  a->rec.release();
  for (auto& r : a->rec_many) r.release();
  delete a;
  return SMESH_OK;
}

int smesh_aggregator_reset(smesh_aggregator_t* a) {
  if (!a) return fail(SMESH_ERR_INVALID, "NULL aggregator");
  std::lock_guard<std::mutex> g(a->mu);
  std::lock_guard<std::recursive_mutex> lock(a->ctx->mu);
  SMESH_HIP(hipSetDevice(a->ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  a->scattered = false;
  SMESH_HIP(hipMemsetAsync(a->acc, 0, (size_t)a->P * a->S * 4, a->ctx->stream));
  if (a->acc_lo) SMESH_HIP(hipMemsetAsync(a->acc_lo, 0, (size_t)a->P * a->S * 4, a->ctx->stream));
  return SMESH_OK;
}

static int aggregator_add(smesh_aggregator_t* a, const void* indices, int idx_dtype, const int64_t is[2], int imem,
                          const float* probs, const int64_t ps[3], int pmem,
                          const float* weights, const int64_t ws[2], int wmem, uint64_t W, uint64_t H, bool wait_for_device_inputs) {
  if (!a || !indices || !probs) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (idx_dtype < 0 || idx_dtype > 3) return fail(SMESH_ERR_INVALID, "bad index dtype");
  SMESH_TRY(check_strides(is, 2, "indices"));
  SMESH_TRY(check_strides(ps, 3, "probs"));
  if (weights) SMESH_TRY(check_strides(ws, 2, "weights"));
  if (W == 0 || H == 0) return SMESH_OK;
  if (W * H >= 0x7FFFFFFFull / 4) return fail(SMESH_ERR_INVALID, "image too large");
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  SMESH_TRY(smesh_aggregator_refuse_scattered(a, "add()"));

  const void* d_idx = indices;
  const float* d_probs = probs;
  const float* d_w = weights;
  if (imem == SMESH_MEM_HOST) {
    const size_t span = 1 + (W - 1) * is[0] + (H - 1) * is[1];
    SMESH_TRY(stage_in(ctx, a->st_idx, indices, span * idx_itemsize(idx_dtype), &d_idx));
  }
  if (pmem == SMESH_MEM_HOST) {
    const size_t span = 1 + (W - 1) * ps[0] + (H - 1) * ps[1] + (size_t)(a->C - 1) * ps[2];
    const void* p;
    SMESH_TRY(stage_in(ctx, a->st_probs, probs, span * 4, &p));
    d_probs = static_cast<const float*>(p);
  }
  if (weights && wmem == SMESH_MEM_HOST) {
    const size_t span = 1 + (W - 1) * ws[0] + (H - 1) * ws[1];
    const void* p;
    SMESH_TRY(stage_in(ctx, a->st_w, weights, span * 4, &p));
    d_w = static_cast<const float*>(p);
  }
  const bool any_host = imem == SMESH_MEM_HOST || pmem == SMESH_MEM_HOST || (weights && wmem == SMESH_MEM_HOST);
  if (any_host) {
    // inputs are never retained after return (Fusion.h:45-47): the caller may overwrite its host arrays as
    // soon as we return, so wait for the staging copies (only the copies -- the kernels run on)
    if (!a->ev_staged) SMESH_HIP(hipEventCreateWithFlags(&a->ev_staged, hipEventDisableTiming));
    SMESH_HIP(hipEventRecord(a->ev_staged, ctx->stream));
  }
  SMESH_TRY(add_device(a, d_idx, idx_dtype, is, d_probs, ps, d_w, ws, W, H));
  if (any_host) SMESH_HIP(hipEventSynchronize(a->ev_staged));
  // device inputs must stay valid until the kernels that read them have run
  if (wait_for_device_inputs && (imem == SMESH_MEM_DEVICE || pmem == SMESH_MEM_DEVICE || (weights && wmem == SMESH_MEM_DEVICE))) {
    // inputs are never retained after return (Fusion.h:45-47): wait for the kernels that read them
    SMESH_HIP(hipStreamSynchronize(ctx->stream));
  }
  return SMESH_OK;
}

int smesh_aggregator_add(smesh_aggregator_t* a, const void* indices, int idx_dtype, const int64_t is[2], int imem,
                         const float* probs, const int64_t ps[3], int pmem,
                         const float* weights, const int64_t ws[2], int wmem, uint64_t W, uint64_t H) {
  return aggregator_add(a, indices, idx_dtype, is, imem, probs, ps, pmem, weights, ws, wmem, W, H, true);
}

int smesh_aggregator_add_async(smesh_aggregator_t* a, const void* indices, int idx_dtype, const int64_t is[2], int imem,
                               const float* probs, const int64_t ps[3], int pmem,
                               const float* weights, const int64_t ws[2], int wmem, uint64_t W, uint64_t H) {
  return aggregator_add(a, indices, idx_dtype, is, imem, probs, ps, pmem, weights, ws, wmem, W, H, false);
}


// add() for a batch of images, in order (new functionality; the reference adds one image per call, Mesh.h:65-107).  The same sums
// as `n` smesh_aggregator_add_async calls -- per accumulator row the same float32 additions in the same order (rows of sparse
// primitives, whose pixels are added by float atomics, excepted) -- but images that are dense uint32 / int32 planes in DEVICE memory with
// dense device class vectors share their kernel launches in groups of up to eight: ONE launch per record pass for the group
// (image_records.hip), ONE triangle-order fusion launch (k_fuse_tri<.., 8>: a row makes one round trip for the group).  Everything
// else goes through smesh_aggregator_add_async image by image.  `weights` may be NULL (no image has weights) or hold one pointer per
// image.  Asynchronous for device images, like smesh_aggregator_add_async.
int smesh_aggregator_add_many(smesh_aggregator_t* a, uint64_t n, const void* const* indices, int idx_dtype, const int64_t is[2], int imem,
                              const float* const* probs, const int64_t ps[3], int pmem,
                              const float* const* weights, const int64_t ws[2], int wmem, uint64_t W, uint64_t H) {
  if (!a || (n && (!indices || !probs || !is || !ps))) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (weights && !ws) return fail(SMESH_ERR_INVALID, "weights without strides");
  for (uint64_t i = 0; i < n; i++)
    if (!indices[i] || !probs[i] || (weights && !weights[i])) return fail(SMESH_ERR_INVALID, "NULL image in the batch");
  if (n == 0 || W == 0 || H == 0) return SMESH_OK;
  bool grouped;
  {
    std::lock_guard<std::mutex> g(a->mu);
    static const bool records_off = getenv("SMESH_ADD_RECORDS") && atoi(getenv("SMESH_ADD_RECORDS")) == 0;
    const char* min_c_env = getenv("SMESH_ADD_RECORDS_MIN_C");
    const uint32_t records_min_c = min_c_env ? (uint32_t)atoi(min_c_env) : kAddRecordsMinC;
    grouped = n >= 2 && imem == SMESH_MEM_DEVICE && pmem == SMESH_MEM_DEVICE && (!weights || wmem == SMESH_MEM_DEVICE) &&
              (idx_dtype == SMESH_IDX_U32 || idx_dtype == SMESH_IDX_I32) && is[0] == (int64_t)H && is[1] == 1 &&
              ps[0] == (int64_t)(H * a->C) && ps[1] == (int64_t)a->C && ps[2] == 1 && (!weights || (ws[0] == (int64_t)H && ws[1] == 1)) &&
              !records_off && a->C >= records_min_c && a->P > 0 && W <= 65535 && H <= 65535 && W * H < 0x7FFFFFFFull / 4 &&
              smesh_aggregator_can_fuse_triangles(a, a->P) && smesh_aggregator_max_fused_views(a) >= 2;
    for (uint64_t i = 0; i < n && grouped; i++) grouped = !(reinterpret_cast<uintptr_t>(probs[i]) & 15);
  }
  uint64_t i = 0;
  while (i < n) {
    int nv = 1;
    if (grouped) {
      std::lock_guard<std::mutex> g(a->mu);
      const int max_nv = smesh_aggregator_max_fused_views(a);
      while (nv * 2 <= std::min<uint64_t>((uint64_t)max_nv, n - i)) nv *= 2;
    }
    if (nv == 1) {
      SMESH_TRY(aggregator_add(a, indices[i], idx_dtype, is, imem, probs[i], ps, pmem, weights ? weights[i] : nullptr, ws, wmem, W, H, false));
      i += 1;
      continue;
    }
    std::lock_guard<std::mutex> g(a->mu);
    DeviceCtx* ctx = a->ctx;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    SMESH_HIP(hipSetDevice(ctx->device));
    SMESH_TRY(smesh_aggregator_join_exchange(a));
    SMESH_TRY(smesh_aggregator_refuse_scattered(a, "add_many()"));
    ImageRecords* recs[8];
    const uint32_t* idx[8];
    const float* pr[8];
    const float* wt[8];
    for (int v = 0; v < nv; v++) {
      recs[v] = v == 0 ? &a->rec : &a->rec_many[v - 1];
      idx[v] = static_cast<const uint32_t*>(indices[i + v]);      // (int32 -1 and uint32 0xFFFFFFFF share a bit pattern)
      pr[v] = probs[i + v];
      wt[v] = weights ? weights[i + v] : nullptr;
    }
    int status = SMESH_OK;
    bool built;
    {
      ProfScope prof(ctx, SMESH_PROF_FUSE_HIST);
      built = image_records_build_group(ctx, recs, idx, nv, W, H, a->P, &status);
    }
    SMESH_TRY(status);
    if (!built) {      // images that take round 2's record passes: one by one (the fusion launch is still shared)
      for (int v = 0; v < nv; v++) {
        SMESH_TRY(image_records_build(ctx, *recs[v], idx[v], W, H, a->P));
        SMESH_TRY(image_records_pending(ctx, *recs[v], idx[v], W, H, ctx->stream));
      }
    }
    RenderedView rv[8];
    for (int v = 0; v < nv; v++) rv[v] = RenderedView{recs[v]->frags, recs[v]->big_queue, recs[v]->big_count, idx[v], pr[v], wt[v], W, H};
    {
      ProfScope fuse_region(ctx, SMESH_PROF_FUSE_SCATTER);
      SMESH_TRY(smesh_aggregator_fuse_triangles(a, a->P, nullptr, (uint32_t)a->P, rv, nv));
    }
    if (a->kind == SMESH_AGG_MUL) SMESH_TRY(ensure_acc_d(a));
    SMESH_TRY(image_records_scatter_sparse_group(ctx, recs, nv, a->kind, idx, pr, weights ? wt : nullptr, W, H, a->C, a->iew, a->acc, a->acc_lo,
                                                 a->acc_d, ctx->stream));
    for (int v = 0; v < nv; v++) SMESH_TRY(image_records_clear(ctx, *recs[v], idx[v], W, H, ctx->stream));
    smesh_note_fuse(smesh_aggregator_fuse_kernel_name(a, false), "image-records");
    i += (uint64_t)nv;
  }
  return SMESH_OK;
}


// rows [row_lo, row_hi) (row_lo a multiple of 4: the tiles move whole 16-byte pieces) normalised into d_out[(row_hi - row_lo) * C]
static int finalize_into(smesh_aggregator* a, float* d_out, uint64_t row_lo = 0, uint64_t row_hi = ~(uint64_t)0) {
  DeviceCtx* ctx = a->ctx;
  row_hi = std::min<uint64_t>(row_hi, a->P);
  if (row_lo >= row_hi) return SMESH_OK;
  const uint64_t P = row_hi - row_lo;
  float* const acc = a->acc + row_lo * a->S;
  float* const acc_lo = a->acc_lo ? a->acc_lo + row_lo * a->S : nullptr;
  ProfScope prof(ctx, SMESH_PROF_FINALIZE);
  int TP = tile_pixels(a->C);
  const int C = (int)a->C;
  // Mul rows of up to 512 classes are centred inside the tile kernel (hi and lo tiles in LDS); wider ones -- a handful of rows per
  // tile, whose threads would walk them alone -- by k_mul_normalise over the whole accumulator first, as in round 1
  const bool mul = a->kind == SMESH_AGG_MUL && C <= 512;
  if (mul) TP = std::max(4, (TP / 2) & ~3);
  if (a->kind == SMESH_AGG_MUL && !mul) SMESH_TRY(mul_normalise(a, false));
  if (TP) {
    const size_t lds = (((size_t)TP * C + 3) & ~(size_t)3) * 4 * (mul ? 2 : 1);
    const dim3 g((uint32_t)div_up(P, TP));
    switch (a->kind) {
      case SMESH_AGG_SUM:    hipLaunchKernelGGL(k_finalize_tile<SMESH_AGG_SUM>, g, dim3(256), lds, ctx->stream, acc, nullptr, d_out, P, C, (int)a->S, TP); break;
      case SMESH_AGG_SUMMAX: hipLaunchKernelGGL(k_finalize_tile<SMESH_AGG_SUMMAX>, g, dim3(256), lds, ctx->stream, acc, nullptr, d_out, P, C, (int)a->S, TP); break;
      default:               hipLaunchKernelGGL(k_finalize_tile<SMESH_AGG_MUL>, g, dim3(256), lds, ctx->stream, acc, mul ? acc_lo : nullptr, d_out, P, C, (int)a->S, TP); break;
    }
  } else {
    const dim3 g((uint32_t)div_up(P, 256)), b(256);
    switch (a->kind) {
      case SMESH_AGG_SUM: hipLaunchKernelGGL(k_finalize_rows<SMESH_AGG_SUM>, g, b, 0, ctx->stream, acc, d_out, P, C, (int)a->S); break;
      case SMESH_AGG_SUMMAX: hipLaunchKernelGGL(k_finalize_rows<SMESH_AGG_SUMMAX>, g, b, 0, ctx->stream, acc, d_out, P, C, (int)a->S); break;
      default: hipLaunchKernelGGL(k_finalize_rows<SMESH_AGG_MUL>, g, b, 0, ctx->stream, acc, d_out, P, C, (int)a->S); break;
    }
  }
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

// Device -> pageable host memory at link speed.  hipMemcpy into pageable memory is staged by the runtime at ~16 GB/s (76 MB of a
// cfg2 result: 4.7 ms for a 34 us kernel); here the result crosses PCIe by DMA into a ring of page-locked chunks (~55 GB/s) and a
// few host threads move each chunk on into the caller's array while the next ones are in flight.
namespace {
struct PinnedRing {
  static constexpr int kChunks = 4;
  static constexpr size_t kChunkBytes = 8u << 20;
  void* buf[kChunks] = {};
  hipEvent_t ev[kChunks] = {};
  bool ok = false;
  bool init() {
    if (ok) return true;
    for (int i = 0; i < kChunks; i++) {
      if (hipHostMalloc(&buf[i], kChunkBytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
      if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    ok = true;
    return true;
  }
};
void parallel_copy(char* dst, const char* src, size_t n) {
  constexpr int kThreads = 12;     // (6 until round 5: first-touch page faults of a fresh pageable array, not the copy, are what a thread spends its time on)
  if (n < (1u << 20)) { memcpy(dst, src, n); return; }
  std::thread th[kThreads];
  const size_t per = ((n + kThreads - 1) / kThreads + 63) & ~(size_t)63;   // (rounded UP: n / kThreads rounded down and already a multiple of 64 left the
                                                                            //  last n % kThreads bytes of a chunk uncopied -- found by tools/soup_sweep.py in round 4)
  for (int t = 0; t < kThreads; t++) {
    const size_t lo = std::min(n, per * (size_t)t), hi = std::min(n, per * (size_t)(t + 1));
    th[t] = std::thread([=] { if (hi > lo) memcpy(dst + lo, src + lo, hi - lo); });
  }
  for (auto& x : th) x.join();
}
}  // namespace

// `d_src` (device) -> `out` (pageable or page-locked host memory), ordered behind what is queued on the context's stream; returns
// when the bytes are in `out`.  Serialised by the context lock (one ring per device context).
static PinnedRing g_rings[64];
// (called when an aggregator with a result of 4 MB and more is created: the ring is not allocated inside somebody's first get())
static void copy_ring_prepare(DeviceCtx* ctx) { (void)g_rings[ctx->device & 63].init(); }

static int copy_to_host(DeviceCtx* ctx, void* out, const void* d_src, size_t bytes) {
  static const bool off = getenv("SMESH_GET_STAGING") && atoi(getenv("SMESH_GET_STAGING")) == 0;
  PinnedRing& ring = g_rings[ctx->device & 63];
  if (!off && bytes >= (4u << 20)) {
    // a fresh numpy array is untouched memory: 19 000 first-touch page faults for a cfg2 result cost more than the transfer.  Ask
    // for transparent huge pages on its page-aligned interior (a no-op where THP is off or the range is already populated).
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(out) + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(out) + bytes) & ~(uintptr_t)((2u << 20) - 1);
    if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
  }
  bool pinned = false;     // page-locked destination (semantic_meshes_amd/device.py: result_empty): DMA straight into it
  {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, out) == hipSuccess) pinned = attr.type == hipMemoryTypeHost;
    else (void)hipGetLastError();
  }
  if (off || pinned || bytes < (4u << 20) || !ring.init()) {
    SMESH_HIP(hipMemcpyAsync(out, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SMESH_HIP(hipStreamSynchronize(ctx->stream));
    return SMESH_OK;
  }
  const size_t CB = PinnedRing::kChunkBytes;
  const size_t nchunks = (bytes + CB - 1) / CB;
  for (size_t k = 0; k < nchunks + PinnedRing::kChunks - 1; k++) {
    if (k < nchunks) {     // chunk k: DMA into its ring slot (the slot's previous tenant, chunk k - kChunks, was drained below)
      const size_t off_b = k * CB, n = std::min(CB, bytes - off_b);
      SMESH_HIP(hipMemcpyAsync(ring.buf[k % PinnedRing::kChunks], static_cast<const char*>(d_src) + off_b, n, hipMemcpyDeviceToHost, ctx->stream));
      SMESH_HIP(hipEventRecord(ring.ev[k % PinnedRing::kChunks], ctx->stream));
    }
    if (k + 1 >= (size_t)PinnedRing::kChunks) {     // drain chunk j = k - (kChunks - 1)
      const size_t j = k + 1 - PinnedRing::kChunks;
      if (j < nchunks) {
        const size_t off_b = j * CB, n = std::min(CB, bytes - off_b);
        SMESH_HIP(hipEventSynchronize(ring.ev[j % PinnedRing::kChunks]));
        parallel_copy(static_cast<char*>(out) + off_b, static_cast<const char*>(ring.buf[j % PinnedRing::kChunks]), n);
      }
    }
  }
  return SMESH_OK;
}

int smesh_aggregator_get(smesh_aggregator_t* a, float* out, int memkind) {
  if (!a || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  SMESH_TRY(smesh_aggregator_refuse_scattered(a, "get()"));
  const size_t bytes = (size_t)a->P * a->C * 4;
  if (bytes == 0) return SMESH_OK;
  if (memkind == SMESH_MEM_DEVICE) {
    SMESH_TRY(finalize_into(a, out));
    SMESH_HIP(hipStreamSynchronize(ctx->stream));
    return SMESH_OK;
  }
  SMESH_TRY(a->out_tmp.reserve(bytes));
  SMESH_TRY(finalize_into(a, static_cast<float*>(a->out_tmp.ptr)));
  return copy_to_host(ctx, out, a->out_tmp.ptr, bytes);
}

int smesh_aggregator_get_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, float* out, int memkind) {
  if (!a || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (row_lo > row_hi || row_hi > a->P || (row_lo & 3)) return fail(SMESH_ERR_INVALID, "bad row range (row_lo must be a multiple of 4, row_hi <= P)");
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  if (a->scattered && (row_lo < a->owned_lo || row_hi > a->owned_hi))
    return fail(SMESH_ERR_INVALID, "get_rows(): after a reduce-scatter only the owned rows hold the fused result");
  const size_t bytes = (size_t)(row_hi - row_lo) * a->C * 4;
  if (bytes == 0) return SMESH_OK;
  float* d_out = out;
  if (memkind != SMESH_MEM_DEVICE) {
    SMESH_TRY(a->out_tmp.reserve(bytes));
    d_out = static_cast<float*>(a->out_tmp.ptr);
  }
  SMESH_TRY(finalize_into(a, d_out, row_lo, row_hi));
  if (memkind != SMESH_MEM_DEVICE) return copy_to_host(ctx, out, d_out, bytes);
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

int smesh_aggregator_get_raw(smesh_aggregator_t* a, float* out, int memkind) {
  if (!a || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  SMESH_TRY(smesh_aggregator_refuse_scattered(a, "get_raw()"));
  const size_t bytes = (size_t)a->P * a->C * 4;
  if (!bytes) return SMESH_OK;
  SMESH_TRY(mul_normalise(a, true));
  // padded rows [P][S] -> dense [P][C]
  SMESH_HIP(hipMemcpy2DAsync(out, (size_t)a->C * 4, a->acc, (size_t)a->S * 4, (size_t)a->C * 4, a->P,
                             memkind == SMESH_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

int smesh_aggregator_set_raw(smesh_aggregator_t* a, const float* in, int memkind) {
  if (!a || !in) return fail(SMESH_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  a->scattered = false;   // every row is being overwritten
  const size_t bytes = (size_t)a->P * a->C * 4;
  if (!bytes) return SMESH_OK;
  if (a->acc_lo) SMESH_HIP(hipMemsetAsync(a->acc_lo, 0, (size_t)a->P * a->S * 4, ctx->stream));
  // dense [P][C] -> padded rows [P][S]; the padding stays zero
  SMESH_HIP(hipMemcpy2DAsync(a->acc, (size_t)a->S * 4, in, (size_t)a->C * 4, (size_t)a->C * 4, a->P,
                             memkind == SMESH_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

// Rows [row_lo, row_hi) of one PLANE of the raw state, dense float32[row_hi - row_lo, C]: plane 0 = the accumulator as it is stored
// (Mul: the hi plane, NOT folded), plane 1 = Mul's lo plane.  What a host-side exchange of a row range moves (distributed.py).
static int raw_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, int plane, float* buf, int memkind, bool set) {
  if (!a || !buf) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (row_lo > row_hi || row_hi > a->P) return fail(SMESH_ERR_INVALID, "bad row range");
  if (plane != 0 && !(plane == 1 && a->kind == SMESH_AGG_MUL)) return fail(SMESH_ERR_INVALID, "plane must be 0 (or 1 for the Mul aggregator's lo plane)");
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  SMESH_TRY(smesh_aggregator_refuse_scattered(a, set ? "set_raw_rows()" : "get_raw_rows()"));
  if (row_lo == row_hi) return SMESH_OK;
  float* rows = (plane ? a->acc_lo : a->acc) + row_lo * a->S;
  if (set)
    SMESH_HIP(hipMemcpy2DAsync(rows, (size_t)a->S * 4, buf, (size_t)a->C * 4, (size_t)a->C * 4, row_hi - row_lo,
                               memkind == SMESH_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
  else
    SMESH_HIP(hipMemcpy2DAsync(buf, (size_t)a->C * 4, rows, (size_t)a->S * 4, (size_t)a->C * 4, row_hi - row_lo,
                               memkind == SMESH_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}
int smesh_aggregator_get_raw_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, int plane, float* out, int memkind) {
  return raw_rows(a, row_lo, row_hi, plane, out, memkind, false);
}
int smesh_aggregator_set_raw_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, int plane, const float* in, int memkind) {
  return raw_rows(a, row_lo, row_hi, plane, const_cast<float*>(in), memkind, true);
}

int smesh_aggregator_raw_pointer(smesh_aggregator_t* a, void** ptr, uint64_t* n) {
  if (!a || !ptr) return fail(SMESH_ERR_INVALID, "NULL argument");
  // callers (the RCCL all-reduce) use this from another stream: make sure our work is done first
  std::lock_guard<std::mutex> g(a->mu);
  std::lock_guard<std::recursive_mutex> lock(a->ctx->mu);
  SMESH_HIP(hipSetDevice(a->ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  SMESH_TRY(mul_normalise(a, true));
  SMESH_HIP(hipStreamSynchronize(a->ctx->stream));
  *ptr = a->acc;
  if (n) *n = a->P * a->S;
  return SMESH_OK;
}

int smesh_aggregator_row_stride(smesh_aggregator_t* a, uint32_t* stride) {
  if (!a || !stride) return fail(SMESH_ERR_INVALID, "NULL argument");
  *stride = a->S;
  return SMESH_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// annotation renderer (Mesh.h:25-42, 124-129)
// ------------------------------------------------------------------------------------------------
struct smesh_annotation_renderer {
  DeviceCtx* ctx = nullptr;
  uint64_t P = 0;
  uint32_t C = 0;
  float* ann = nullptr;      // snapshot of get(): float32[P*C]
  float* bg = nullptr;       // background vector [C]
  Scratch st_idx, nm_idx, out_tmp;
  std::mutex mu;
};

extern "C" {

int smesh_aggregator_renderer(smesh_aggregator_t* a, smesh_annotation_renderer_t** out) {
  if (!a || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  *out = nullptr;
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));
  SMESH_TRY(smesh_aggregator_refuse_scattered(a, "renderer()"));
  auto* r = new (std::nothrow) smesh_annotation_renderer();
  if (!r) return fail(SMESH_ERR_RUNTIME, "out of memory");
  r->ctx = ctx; r->P = a->P; r->C = a->C;
  const size_t bytes = (size_t)a->P * a->C * 4;
  hipError_t e = dev_malloc(reinterpret_cast<void**>(&r->ann), bytes ? bytes : 16);
  if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&r->bg), (size_t)a->C * 4);
  if (e != hipSuccess) { if (r->ann) (void)dev_free(r->ann); delete r; return fail_hip(e, "annotation renderer allocation", __FILE__, __LINE__); }
  int st = bytes ? finalize_into(a, r->ann) : SMESH_OK;   // m_annotations = elwise(get)  (Mesh.h:127)
  if (st == SMESH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = fail(SMESH_ERR_RUNTIME, "stream sync failed");
  if (st) { (void)dev_free(r->ann); (void)dev_free(r->bg); delete r; return st; }
  *out = r;
  return SMESH_OK;
}

int smesh_annotation_renderer_render(smesh_annotation_renderer_t* r, const void* indices, int idx_dtype, const int64_t is[2],
                                     int imem, const float* background, float* out, int omem, uint64_t W, uint64_t H) {
  if (!r || !indices || !background || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (idx_dtype < 0 || idx_dtype > 3) return fail(SMESH_ERR_INVALID, "bad index dtype");
  SMESH_TRY(check_strides(is, 2, "indices"));
  if (W == 0 || H == 0) return SMESH_OK;
  std::lock_guard<std::mutex> g(r->mu);
  DeviceCtx* ctx = r->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  const uint64_t N = W * H, total = N * r->C;
  const void* d_idx = indices;
  if (imem == SMESH_MEM_HOST) {
    const size_t span = 1 + (W - 1) * is[0] + (H - 1) * is[1];
    SMESH_TRY(stage_in(ctx, r->st_idx, indices, span * idx_itemsize(idx_dtype), &d_idx));
  }
  const uint32_t* idx = nullptr;
  SMESH_TRY(normalize_idx(ctx, r->nm_idx, d_idx, idx_dtype, is, W, H, &idx));
  SMESH_HIP(hipMemcpyAsync(r->bg, background, (size_t)r->C * 4, hipMemcpyHostToDevice, ctx->stream));
  float* d_out = out;
  if (omem == SMESH_MEM_HOST) {
    SMESH_TRY(r->out_tmp.reserve(total * 4));
    d_out = static_cast<float*>(r->out_tmp.ptr);
  }
  hipLaunchKernelGGL(k_gather_annotations, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, ctx->stream, idx, r->ann, r->bg,
                     d_out, N, (uint32_t)r->P, r->C);
  SMESH_HIP(hipGetLastError());
  if (omem == SMESH_MEM_HOST) SMESH_HIP(hipMemcpyAsync(out, d_out, total * 4, hipMemcpyDeviceToHost, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

int smesh_annotation_renderer_destroy(smesh_annotation_renderer_t* r) {
  if (!r) return SMESH_OK;
  (void)hipSetDevice(r->ctx->device);
  (void)hipStreamSynchronize(r->ctx->stream);
  (void)dev_free(r->ann);
  (void)dev_free(r->bg);
  r->st_idx.release(); r->nm_idx.release(); r->out_tmp.release();
  delete r;
  return SMESH_OK;
}

}  // extern "C"
