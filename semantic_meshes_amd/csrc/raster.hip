// raster.hip -- triangle / texel rasteriser for MI355X (gfx950).
//
// Replaces (citations relative to /root/reference):
//   include/semantic_meshes/render/TriangleRenderer.h:30-39   mesh upload            -> smesh_renderer_create_triangles
//   include/semantic_meshes/render/TriangleRenderer.h:75-78   clear {+inf, -1}       -> fused into k_resolve (keys are re-armed while being read)
//   include/semantic_meshes/render/TriangleRenderer.h:81-88   DeviceMutexRasterizer  -> k_project_vertices + k_raster_*
//   python/semantic_meshes/include/Renderer.h:32-35           AoS -> SoA split       -> k_resolve writes both planes
//
// The rasteriser's arithmetic is NOT in the reference tree (template-tensors submodule is empty), so the
// sampling / fill / depth / tie rules are this project's documented decisions (DESIGN.md "Raster spec",
// SURVEY.md Appendix B).  This file implements that spec independently of oracle/smesh_oracle.cpp; the two
// must agree bit-for-bit on the index image, which is why the translation unit is compiled with
// -ffp-contract=off and every expression below has a fixed evaluation order.
//
// Depth test: a pixel holds one 64-bit key (float bits of z << 32 | primitive id); nearer z wins and equal
// z resolves to the lower id, so the result does not depend on the order triangles are processed in.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#pragma clang fp contract(off)

using namespace smesh;

// from fusion.hip
struct smesh_aggregator;
int smesh_aggregator_add_device_contig(smesh_aggregator* a, const uint32_t* d_idx, const float* d_probs,
                                       const float* d_w, uint64_t W, uint64_t H);
bool smesh_aggregator_can_fuse_triangles(smesh_aggregator* a, uint64_t F);
int smesh_aggregator_fuse_triangles(smesh_aggregator* a, const TriFrag* frags, uint64_t F, const uint32_t* big_queue,
                                    const uint32_t* big_len, uint32_t big_capacity, const uint32_t* d_idx,
                                    const float* d_probs, const float* d_w, uint64_t H);
DeviceCtx* smesh_aggregator_ctx(smesh_aggregator* a);
uint32_t smesh_aggregator_classes(smesh_aggregator* a);
std::mutex& smesh_aggregator_mutex(smesh_aggregator* a);
Scratch& smesh_aggregator_stage_probs(smesh_aggregator* a);
Scratch& smesh_aggregator_stage_w(smesh_aggregator* a);

namespace {

constexpr float kNear = 1e-6f;
constexpr unsigned long long kBackgroundKey = (0x7F800000ull << 32) | 0xFFFFFFFFull;  // {+inf, -1}

struct ScreenVertex {
  double u, v;  // pixel coordinates
  double iz;    // 1/z_c, 0 = unusable vertex
};

struct CameraArgs {
  float R[9];
  float t[3];
  double fx, fy, cx, cy;
  uint32_t W, H;
};

// ---- vertex stage: float32 rigid transform, double pinhole projection (render/Camera.h:9-13) --------
__global__ void k_project_vertices(const float* __restrict__ verts, uint64_t V, CameraArgs cam,
                                   ScreenVertex* __restrict__ sv) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V) return;
  const float X = verts[3 * i + 0], Y = verts[3 * i + 1], Z = verts[3 * i + 2];
  const float xc = ((cam.R[0] * X + cam.R[1] * Y) + cam.R[2] * Z) + cam.t[0];
  const float yc = ((cam.R[3] * X + cam.R[4] * Y) + cam.R[5] * Z) + cam.t[1];
  const float zc = ((cam.R[6] * X + cam.R[7] * Y) + cam.R[8] * Z) + cam.t[2];
  ScreenVertex s;
  s.u = 0.0; s.v = 0.0; s.iz = 0.0;
  if (zc > kNear && isfinite(xc) && isfinite(yc) && isfinite(zc)) {
    const double zd = (double)zc;
    const double u = cam.fx * ((double)xc / zd) + cam.cx;
    const double v = cam.fy * ((double)yc / zd) + cam.cy;
    if (isfinite(u) && isfinite(v)) {
      s.u = u; s.v = v; s.iz = 1.0 / zd;
    }
  }
  sv[i] = s;
}

// ---- triangle setup --------------------------------------------------------------------------------
// Edge through A,B evaluated from canonically ordered endpoints: both triangles sharing the edge get the
// same |value|, so a sample is claimed by exactly one of them (with the tie rule below).
struct EdgeEq {
  double lx, ly, dx, dy, sign;
};

__device__ __forceinline__ EdgeEq make_edge(double ax, double ay, double bx, double by) {
  EdgeEq e;
  const bool sw = (bx < ax) || (bx == ax && by < ay);
  e.lx = sw ? bx : ax;
  e.ly = sw ? by : ay;
  const double hx = sw ? ax : bx;
  const double hy = sw ? ay : by;
  e.dx = hx - e.lx;
  e.dy = hy - e.ly;
  e.sign = sw ? -1.0 : 1.0;
  return e;
}

__device__ __forceinline__ double eval_edge(const EdgeEq& e, double px, double py) {
  const double a = e.dx * (py - e.ly);
  const double b = e.dy * (px - e.lx);
  return e.sign * (a - b);
}

struct Tri {
  EdgeEq e0, e1, e2;   // e_i is opposite vertex i
  double s;            // orientation sign
  bool own0, own1, own2;
  double iz0, iz1, iz2;
  int x0, x1, y0, y1;
};

__device__ __forceinline__ bool owns(double s, const EdgeEq& e) {
  const double A = s * (e.sign * (-e.dy));
  const double B = s * (e.sign * e.dx);
  return (A > 0.0) || (A == 0.0 && B > 0.0);
}

__device__ __forceinline__ bool setup_tri(const ScreenVertex& a, const ScreenVertex& b, const ScreenVertex& c,
                                          uint32_t W, uint32_t H, Tri& t) {
  if (a.iz == 0.0 || b.iz == 0.0 || c.iz == 0.0) return false;
  const double minu = fmin(a.u, fmin(b.u, c.u)), maxu = fmax(a.u, fmax(b.u, c.u));
  const double minv = fmin(a.v, fmin(b.v, c.v)), maxv = fmax(a.v, fmax(b.v, c.v));
  double fx0 = ceil(minu - 0.5), fx1 = floor(maxu - 0.5);
  double fy0 = ceil(minv - 0.5), fy1 = floor(maxv - 0.5);
  if (fx0 < 0.0) fx0 = 0.0;
  if (fy0 < 0.0) fy0 = 0.0;
  if (fx1 > (double)(W - 1)) fx1 = (double)(W - 1);
  if (fy1 > (double)(H - 1)) fy1 = (double)(H - 1);
  if (!(fx0 <= fx1) || !(fy0 <= fy1)) return false;
  t.x0 = (int)fx0; t.x1 = (int)fx1; t.y0 = (int)fy0; t.y1 = (int)fy1;
  t.e0 = make_edge(b.u, b.v, c.u, c.v);
  t.e1 = make_edge(c.u, c.v, a.u, a.v);
  t.e2 = make_edge(a.u, a.v, b.u, b.v);
  const double area2 = eval_edge(t.e2, c.u, c.v);
  if (!(area2 != 0.0) || !isfinite(area2)) return false;
  t.s = area2 > 0.0 ? 1.0 : -1.0;
  t.own0 = owns(t.s, t.e0);
  t.own1 = owns(t.s, t.e1);
  t.own2 = owns(t.s, t.e2);
  t.iz0 = a.iz; t.iz1 = b.iz; t.iz2 = c.iz;
  return true;
}

// Coverage + depth of one sample; returns false if the sample is not covered.
__device__ __forceinline__ bool shade(const Tri& t, int x, int y, float* z, double* b1, double* b2) {
  const double px = (double)x + 0.5, py = (double)y + 0.5;
  const double w0 = t.s * eval_edge(t.e0, px, py);
  if (!(w0 > 0.0 || (w0 == 0.0 && t.own0))) return false;
  const double w1 = t.s * eval_edge(t.e1, px, py);
  if (!(w1 > 0.0 || (w1 == 0.0 && t.own1))) return false;
  const double w2 = t.s * eval_edge(t.e2, px, py);
  if (!(w2 > 0.0 || (w2 == 0.0 && t.own2))) return false;
  const double num = (w0 + w1) + w2;
  const double den = (w0 * t.iz0 + w1 * t.iz1) + w2 * t.iz2;
  const float zf = (float)(num / den);
  if (!(zf > 0.0f) || !isfinite(zf)) return false;
  *z = zf;
  if (b1) { *b1 = w1 / num; *b2 = w2 / num; }
  return true;
}

__device__ __forceinline__ uint32_t texel_of(uint32_t res, double b1, double b2) {
  // TexturedTriangleRenderer.h:34-38; toIndex bijection decided in DESIGN.md (B-5)
  const float u = (float)b1, v = (float)b2;
  const float fu = (u - 1e-6f) * (float)res, fv = (v - 1e-6f) * (float)res;
  int tu = (int)fu, tv = (int)fv;
  if (tu < 0) tu = 0;
  if (tv < 0) tv = 0;
  const int r1 = (int)res - 1;
  if (tu > r1) tu = r1;
  if (tv > r1 - tu) tv = r1 - tu;
  const int row = tu + tv;
  return (uint32_t)(row * (row + 1) / 2 + tu);
}

struct RasterArgs {
  const int32_t* faces;
  const ScreenVertex* sv;
  const uint32_t* tex_res;    // null for triangle primitives
  const uint32_t* tex_first;
  unsigned long long* keys;
  uint64_t F, V;
  uint32_t W, H;
  uint32_t* big_queue;        // triangles deferred to the cooperative kernel
  uint32_t* big_count;
  uint32_t big_capacity;
  TriFrag* frags;             // per-triangle fragment records for the triangle-order fusion (may be null)
  int dbg;                    // development ablation (SMESH_RDBG): 1 = no atomics, 2 = setup only
};

// Key image layout: same (W,H) y-fastest order as the output planes.  (A 4 x 4 blocked layout was tried to
// make neighbouring fragments share cache lines: no change in k_raster_small, slower resolve.)
__device__ __forceinline__ uint64_t key_index(uint32_t x, uint32_t y, uint32_t H) { return (uint64_t)x * H + y; }

__device__ __forceinline__ bool load_tri(const RasterArgs& a, uint64_t f, Tri& t) {
  const int32_t i0 = a.faces[3 * f + 0], i1 = a.faces[3 * f + 1], i2 = a.faces[3 * f + 2];
  if (i0 < 0 || i1 < 0 || i2 < 0) return false;
  if ((uint64_t)i0 >= a.V || (uint64_t)i1 >= a.V || (uint64_t)i2 >= a.V) return false;
  if (a.tex_res && a.tex_res[f] == 0) return false;
  return setup_tri(a.sv[i0], a.sv[i1], a.sv[i2], a.W, a.H, t);
}

__device__ __forceinline__ void emit(const RasterArgs& a, uint64_t f, const Tri& t, int x, int y) {
  float z;
  double b1, b2;
  if (!shade(t, x, y, &z, a.tex_res ? &b1 : nullptr, &b2)) return;
  uint32_t prim = (uint32_t)f;
  if (a.tex_res) prim = a.tex_first[f] + texel_of(a.tex_res[f], b1, b2);
  const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | prim;
  if (a.dbg & 1) { if (key == 12345ull) a.keys[0] = key; return; }
  atomicMin(&a.keys[key_index((uint32_t)x, (uint32_t)y, a.H)], key);
}

__device__ __forceinline__ bool shade_key(const RasterArgs& a, uint64_t f, const Tri& t, int x, int y,
                                          unsigned long long* key) {
  float z;
  double b1, b2;
  if (!shade(t, x, y, &z, a.tex_res ? &b1 : nullptr, &b2)) return false;
  uint32_t prim = (uint32_t)f;
  if (a.tex_res) prim = a.tex_first[f] + texel_of(a.tex_res[f], b1, b2);
  *key = ((unsigned long long)__float_as_uint(z) << 32) | prim;
  return true;
}

// One lane per triangle; triangles whose bounding box exceeds 8 x 8 pixels are queued for k_raster_big.
// Besides the depth-tested keys, each triangle leaves a TriFrag record (which pixels it emitted).
__global__ void k_raster_small(RasterArgs a) {
  const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= a.F) return;
  TriFrag rec;
  rec.x0 = 0; rec.y0 = 0; rec.kind = 0; rec.pad = 0; rec.mask = 0ull;
  Tri t;
  if (load_tri(a, f, t)) {
    const int bw = t.x1 - t.x0 + 1, bh = t.y1 - t.y0 + 1;
    rec.x0 = (uint16_t)t.x0; rec.y0 = (uint16_t)t.y0;
    if (bw > 8 || bh > 8) {
      const uint32_t slot = atomicAdd(a.big_count, 1u);
      if (slot < a.big_capacity) a.big_queue[slot] = (uint32_t)f;
      rec.kind = 2;
      rec.mask = (unsigned long long)(uint32_t)t.x1 | ((unsigned long long)(uint32_t)t.y1 << 16);
    } else if (!(a.dbg & 2)) {
      unsigned long long mask = 0ull;
      for (int dx = 0; dx < bw; dx++)
        for (int dy = 0; dy < bh; dy++) {
          unsigned long long key;
          if (shade_key(a, f, t, t.x0 + dx, t.y0 + dy, &key)) {
            mask |= 1ull << (dx * 8 + dy);
            if (!(a.dbg & 1)) atomicMin(&a.keys[key_index((uint32_t)(t.x0 + dx), (uint32_t)(t.y0 + dy), a.H)], key);
          }
        }
      rec.kind = mask ? 1 : 0;
      rec.mask = mask;
    }
  }
  if (a.frags) a.frags[f] = rec;
}

// One workgroup per (queued triangle, 64x64 pixel chunk of its bounding box); lanes run down columns.
__global__ __launch_bounds__(256) void k_raster_big(RasterArgs a, uint32_t chunks_budget) {
  const uint32_t nbig = min(*a.big_count, a.big_capacity);
  // blocks walk the queue; each triangle is split into 64x64 chunks processed by successive blocks
  for (uint32_t q = blockIdx.x; q < nbig; q += gridDim.x) {
    const uint64_t f = a.big_queue[q];
    Tri t;
    if (!load_tri(a, f, t)) continue;
    const int bw = t.x1 - t.x0 + 1, bh = t.y1 - t.y0 + 1;
    const int ty = threadIdx.x & 63, tx = threadIdx.x >> 6;  // 64 rows x 4 columns per pass
    for (int cx = 0; cx < bw; cx += 4) {
      const int x = t.x0 + cx + tx;
      if (x > t.x1) continue;
      for (int cy = 0; cy < bh; cy += 64) {
        const int y = t.y0 + cy + ty;
        if (y <= t.y1) emit(a, f, t, x, y);
      }
    }
  }
  (void)chunks_budget;
}

// ------------------------------------------------------------------------------------------------
// Tiled path: triangles are binned to screen tiles of 16 columns x 32 rows (a tile column is one 128-byte
// line of each output plane); one workgroup per tile resolves depth in LDS (4 KiB of 64-bit keys,
// ds_min_u64) and writes both planes coalesced.  No global atomics per fragment, no clear, no split pass.
// ------------------------------------------------------------------------------------------------
constexpr int kTW = 16, kTH = 32, kTilePixels = kTW * kTH;
constexpr uint32_t kSkipCode = 0xFFFFFFFFu;
constexpr int kMaxTilesPerTri = 16;   // triangles overlapping more tiles take the cooperative big-triangle path

struct BinArgs {
  RasterArgs r;
  uint32_t tiles_x, tiles_y, ntiles;
  uint32_t* tile_count;    // [ntiles]   triangles per tile; reused as the fill cursor
  uint32_t* tile_offset;   // [ntiles+1] exclusive scan
  uint32_t* tri_code;      // [F] packed tile range of each triangle (tx0:12 | ty0:12 | nx-1:4 | ny-1:4)
  uint32_t* tile_list;     // [<= 16 F] triangle ids grouped by tile
  uint32_t* idx_out;
  float* depth_out;
};

// Adds `active ? 1 : 0` to counter[tile] with one atomic per distinct tile in the wave; returns this
// lane's slot (old value + rank among the lanes of the same tile).  Must be called by the whole wave.
// The distinct tiles are found with ballots only (no memory traffic); then ALL leaders issue their atomics
// in one instruction, so a wave pays one atomic round trip, not one per distinct tile.
__device__ __forceinline__ uint32_t wave_claim(uint32_t* counter, uint32_t tile, bool active, bool want_slot) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(active);
  unsigned long long mine = 0ull;   // lanes holding the same tile as this lane
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const uint32_t t0 = (uint32_t)__shfl((int)tile, first);
    const unsigned long long same = __ballot(active && tile == t0);
    if (active && tile == t0) mine = same;
    todo &= ~same;
  }
  if (!active) return 0u;
  const int leader = __ffsll((long long)mine) - 1;
  const uint32_t cnt = (uint32_t)__popcll(mine);
  uint32_t base = 0;
  if (lane == leader) {
    if (want_slot) base = atomicAdd(&counter[tile], cnt);
    else atomicAdd(&counter[tile], cnt);
  }
  if (!want_slot) return 0u;
  base = (uint32_t)__shfl((int)base, leader);
  return base + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
}

// Pass A: one lane per triangle: setup, cull, tile range; count triangles per tile.
__global__ __launch_bounds__(256) void k_bin_count(BinArgs b) {
  const RasterArgs& a = b.r;
  const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t code = kSkipCode;
  int tx0 = 0, ty0 = 0, nx = 0, n = 0;
  Tri t;
  if (f < a.F && load_tri(a, f, t)) {
    tx0 = t.x0 / kTW; ty0 = t.y0 / kTH;
    nx = t.x1 / kTW - tx0 + 1;
    const int ny = t.y1 / kTH - ty0 + 1;
    if (nx * ny > kMaxTilesPerTri) {
      const uint32_t slot = atomicAdd(a.big_count, 1u);
      if (slot < a.big_capacity) a.big_queue[slot] = (uint32_t)f;
    } else {
      n = nx * ny;
      code = ((uint32_t)tx0 << 20) | ((uint32_t)ty0 << 8) | ((uint32_t)(nx - 1) << 4) | (uint32_t)(ny - 1);
    }
  }
  if (f < a.F) b.tri_code[f] = code;
  for (int k = 0; __ballot(k < n) != 0ull; k++) {
    const bool active = k < n;
    const uint32_t tile = active ? (uint32_t)(tx0 + k % nx) * b.tiles_y + (uint32_t)(ty0 + k / nx) : 0u;
    wave_claim(b.tile_count, tile, active, false);
  }
}

// Exclusive scan of the tile counts (one workgroup); the counts become zeroed fill cursors.
__global__ __launch_bounds__(1024) void k_bin_scan(BinArgs b) {
  __shared__ uint32_t part[1024];
  const int t = threadIdx.x;
  const uint32_t per = (b.ntiles + 1023u) / 1024u;
  const uint32_t lo = min((uint32_t)t * per, b.ntiles), hi = min(lo + per, b.ntiles);
  uint32_t sum = 0;
  for (uint32_t i = lo; i < hi; i++) sum += b.tile_count[i];
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = t >= d ? part[t - d] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = part[t] - sum;   // exclusive prefix of this thread's chunk
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t c = b.tile_count[i];
    b.tile_offset[i] = run;
    b.tile_count[i] = 0;
    run += c;
  }
  if (t == 1023) b.tile_offset[b.ntiles] = part[1023];
}

// Pass B: write each triangle id into the lists of the tiles it overlaps.
__global__ __launch_bounds__(256) void k_bin_fill(BinArgs b) {
  const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t code = f < b.r.F ? b.tri_code[f] : kSkipCode;
  int tx0 = 0, ty0 = 0, nx = 1, n = 0;
  if (code != kSkipCode) {
    tx0 = (int)(code >> 20); ty0 = (int)((code >> 8) & 0xFFFu);
    nx = (int)((code >> 4) & 0xFu) + 1;
    n = nx * ((int)(code & 0xFu) + 1);
  }
  for (int k = 0; __ballot(k < n) != 0ull; k++) {
    const bool active = k < n;
    const uint32_t tile = active ? (uint32_t)(tx0 + k % nx) * b.tiles_y + (uint32_t)(ty0 + k / nx) : 0u;
    const uint32_t slot = wave_claim(b.tile_count, tile, active, true);
    if (active) b.tile_list[b.tile_offset[tile] + slot] = (uint32_t)f;
  }
}

// One workgroup per tile: depth resolve in LDS, then both planes written once.
__global__ __launch_bounds__(256) void k_raster_tile(BinArgs b) {
  const RasterArgs& a = b.r;
  __shared__ unsigned long long skeys[kTilePixels];
  const int t = threadIdx.x;
  const uint32_t tile = blockIdx.x;
  const uint32_t tx = tile / b.tiles_y, ty = tile - tx * b.tiles_y;
  const int x0 = (int)tx * kTW, y0 = (int)ty * kTH;
  const uint32_t off = b.tile_offset[tile];
  const uint32_t n = b.tile_offset[tile + 1] - off;
  const bool merge_big = *a.big_count != 0u;   // big triangles went through global atomics into a.keys
  for (int p = t; p < kTilePixels; p += 256) {
    const int gx = x0 + (p >> 5), gy = y0 + (p & 31);
    unsigned long long k = kBackgroundKey;
    if (merge_big && gx < (int)a.W && gy < (int)a.H) {
      const uint64_t g = key_index((uint32_t)gx, (uint32_t)gy, a.H);
      k = a.keys[g];
      a.keys[g] = kBackgroundKey;   // re-armed for the next render
    }
    skeys[p] = k;
  }
  __syncthreads();
  for (uint32_t base = 0; base < n; base += 256) {
    const uint32_t i = base + t;
    if (i < n) {
      const uint64_t f = b.tile_list[off + i];
      Tri tr;
      if (load_tri(a, f, tr)) {
        const int xa = max(tr.x0, x0), xb = min(tr.x1, x0 + kTW - 1);
        const int ya = max(tr.y0, y0), yb = min(tr.y1, y0 + kTH - 1);
        for (int x = xa; x <= xb; x++)
          for (int y = ya; y <= yb; y++) {
            unsigned long long key;
            if (shade_key(a, f, tr, x, y, &key)) atomicMin(&skeys[(x - x0) * kTH + (y - y0)], key);
          }
      }
    }
  }
  __syncthreads();
  for (int p = t; p < kTilePixels; p += 256) {
    const int gx = x0 + (p >> 5), gy = y0 + (p & 31);
    if (gx < (int)a.W && gy < (int)a.H) {
      const unsigned long long k = skeys[p];
      const uint64_t g = (uint64_t)gx * a.H + gy;
      b.idx_out[g] = (uint32_t)(k & 0xFFFFFFFFull);
      b.depth_out[g] = __uint_as_float((uint32_t)(k >> 32));
    }
  }
}

// Split the key image into the two output planes and re-arm the keys for the next render.
__global__ void k_resolve(unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx, float* __restrict__ depth,
                          uint32_t W, uint32_t H, uint32_t* __restrict__ big_count) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {   // the queue of this render has been consumed by the rasteriser: re-arm it, keep its length
    big_count[1] = big_count[0];
    big_count[0] = 0u;
  }
  if (i >= (uint64_t)W * H) return;
  const uint32_t x = (uint32_t)(i / H), y = (uint32_t)(i - (uint64_t)x * H);
  const uint64_t g = key_index(x, y, H);
  const unsigned long long k = keys[g];
  keys[g] = kBackgroundKey;
  idx[i] = (uint32_t)(k & 0xFFFFFFFFull);
  depth[i] = __uint_as_float((uint32_t)(k >> 32));
}

__global__ void k_fill_keys(unsigned long long* __restrict__ keys, uint64_t N) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) keys[i] = kBackgroundKey;
}

// ---- texel constructor: max projected area per triangle over all cameras (TexturedTriangleRenderer.h:94-127)
__global__ void k_texel_area(const float* __restrict__ verts, const int32_t* __restrict__ faces, uint64_t F, uint64_t V,
                             const smesh_camera_t* __restrict__ cams, uint32_t K, float* __restrict__ best_area) {
  const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const int32_t id[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
  float best = 0.0f;
  bool valid = true;
  for (int k = 0; k < 3; k++) if (id[k] < 0 || (uint64_t)id[k] >= V) valid = false;
  if (valid) {
    float P[3][3];
    for (int k = 0; k < 3; k++)
      for (int d = 0; d < 3; d++) P[k][d] = verts[3 * (uint64_t)id[k] + d];
    for (uint32_t ci = 0; ci < K; ci++) {
      const smesh_camera_t& cam = cams[ci];
      float pu[3], pv[3];
      bool in_front = false;
      for (int k = 0; k < 3; k++) {
        const float xc = ((cam.rotation[0] * P[k][0] + cam.rotation[1] * P[k][1]) + cam.rotation[2] * P[k][2]) + cam.translation[0];
        const float yc = ((cam.rotation[3] * P[k][0] + cam.rotation[4] * P[k][1]) + cam.rotation[5] * P[k][2]) + cam.translation[1];
        const float zc = ((cam.rotation[6] * P[k][0] + cam.rotation[7] * P[k][1]) + cam.rotation[8] * P[k][2]) + cam.translation[2];
        in_front |= zc > 0.0f;
        pu[k] = (float)(cam.focal[0] * ((double)xc / (double)zc) + cam.principal[0]);
        pv[k] = (float)(cam.focal[1] * ((double)yc / (double)zc) + cam.principal[1]);
      }
      const float border = 0.5f;
      const float rw = (float)(int)cam.width, rh = (float)(int)cam.height;
      bool inside = in_front;
      for (int k = 0; k < 3; k++)
        inside = inside && (-border * rw <= pu[k]) && (pu[k] < (1 + border) * rw) &&
                 (-border * rh <= pv[k]) && (pv[k] < (1 + border) * rh);
      if (inside) {
        const float area = 0.5f * fabsf((pu[0] * (pv[1] - pv[2]) + pu[1] * (pv[2] - pv[0])) + pu[2] * (pv[0] - pv[1]));
        if (area > best) best = area;
      }
    }
  }
  best_area[f] = best;
}

}  // namespace

struct ImagePair {
  uint32_t* idx;
  float* depth;
  uint64_t pixels;
  bool idx_out, depth_out;  // handed to the caller and not yet released
};

struct smesh_renderer {
  DeviceCtx* ctx = nullptr;
  uint64_t V = 0, F = 0, num_primitives = 0;
  float* verts = nullptr;          // float32[V*3]
  int32_t* faces = nullptr;        // int32[F*3]
  ScreenVertex* sv = nullptr;      // per-view projected vertices [V]
  bool texels = false;
  uint32_t* tex_res = nullptr;     // [F]
  uint32_t* tex_first = nullptr;   // [F]
  std::vector<int32_t> h_faces;    // texel renderers: re-ordered faces
  std::vector<uint32_t> h_res, h_first;
  unsigned long long* keys = nullptr;
  uint64_t keys_pixels = 0;
  uint32_t* big_queue = nullptr;
  uint32_t* big_count = nullptr;
  uint32_t big_capacity = 0;
  TriFrag* frags = nullptr;        // [F] per-triangle fragment records (direct path)
  uint32_t* tri_code = nullptr;    // [F] tiled path: packed tile range per triangle
  uint32_t* tile_list = nullptr;   // [16 F]
  Scratch tile_tables;             // tile_count[ntiles] + tile_offset[ntiles+1]
  std::vector<ImagePair> images;   // pooled output planes
  Scratch own_idx;                 // for the host-output entry point
  // smesh_fuse_view pipeline: two index/depth slots, rasterised on ctx->raster_stream
  Scratch fused[2];
  hipEvent_t ev_rendered[2] = {nullptr, nullptr};   // raster stream: slot is complete
  hipEvent_t ev_consumed[2] = {nullptr, nullptr};   // main stream: the fusion kernels have read the slot
  uint64_t fused_seq = 0;
  bool raster_pending = false;     // work queued on the raster stream since the last synchronisation
  bool main_pending = false;       // renderer state (keys, scratch) used on the main stream since then
  std::mutex mu;
};

namespace {

int ensure_keys(smesh_renderer* r, uint64_t W, uint64_t H, hipStream_t st) {
  const uint64_t N = div_up(W, 4) * div_up(H, 4) * 16;   // 4 x 4 blocked layout, padded
  if (N <= r->keys_pixels) return SMESH_OK;
  if (r->keys) SMESH_HIP(hipFree(r->keys));
  r->keys = nullptr;
  r->keys_pixels = 0;
  SMESH_HIP(hipMalloc(reinterpret_cast<void**>(&r->keys), N * 8));
  r->keys_pixels = N;
  hipLaunchKernelGGL(k_fill_keys, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, r->keys, N);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

// Rasterise into caller-provided device planes (both required).
int render_into(smesh_renderer* r, const smesh_camera_t* cam, uint32_t* d_idx, float* d_depth, hipStream_t st = nullptr) {
  DeviceCtx* ctx = r->ctx;
  if (!st) {
    // plain render()/render_device(): main stream, after whatever smesh_fuse_view left on the raster stream
    st = ctx->stream;
    if (r->raster_pending) {
      SMESH_HIP(hipStreamSynchronize(ctx->raster_stream));
      r->raster_pending = false;
    }
    r->main_pending = true;
  }
  const uint64_t W = cam->width, H = cam->height, N = W * H;
  SMESH_TRY(ensure_keys(r, W, H, st));
  ProfScope prof(ctx, SMESH_PROF_RASTER, st);
  CameraArgs ca;
  memcpy(ca.R, cam->rotation, sizeof ca.R);
  memcpy(ca.t, cam->translation, sizeof ca.t);
  ca.fx = cam->focal[0]; ca.fy = cam->focal[1]; ca.cx = cam->principal[0]; ca.cy = cam->principal[1];
  ca.W = (uint32_t)W; ca.H = (uint32_t)H;
  if (r->V) {
    hipLaunchKernelGGL(k_project_vertices, dim3((uint32_t)div_up(r->V, 256)), dim3(256), 0, st, r->verts, r->V, ca, r->sv);
    SMESH_HIP(hipGetLastError());
  }
  if (r->F) {
    RasterArgs a;
    a.faces = r->faces; a.sv = r->sv; a.tex_res = r->texels ? r->tex_res : nullptr; a.tex_first = r->tex_first;
    a.keys = r->keys; a.F = r->F; a.V = r->V; a.W = (uint32_t)W; a.H = (uint32_t)H;
    a.big_queue = r->big_queue; a.big_count = r->big_count; a.big_capacity = r->big_capacity;
    a.frags = r->frags;
    { static const int rdbg = getenv("SMESH_RDBG") ? atoi(getenv("SMESH_RDBG")) : 0; a.dbg = rdbg; }
    static const bool direct = !(getenv("SMESH_RASTER") && std::string(getenv("SMESH_RASTER")) == "tiled");
    const uint32_t tiles_x = (uint32_t)div_up(W, kTW), tiles_y = (uint32_t)div_up(H, kTH);
    const uint64_t ntiles = (uint64_t)tiles_x * tiles_y;
    const uint32_t big_grid = (uint32_t)std::min<uint64_t>(r->F, (uint64_t)ctx->num_cus);
    if (!direct && ntiles <= (1u << 20)) {
      SMESH_HIP(hipMemsetAsync(r->big_count, 0, 4, st));
      // tiled path: bin (count, scan, fill), big triangles through global atomics, then one workgroup per tile
      SMESH_TRY(r->tile_tables.reserve((2 * ntiles + 1) * 4));
      BinArgs b;
      b.r = a;
      b.tiles_x = tiles_x; b.tiles_y = tiles_y; b.ntiles = (uint32_t)ntiles;
      b.tile_count = static_cast<uint32_t*>(r->tile_tables.ptr);
      b.tile_offset = b.tile_count + ntiles;
      b.tri_code = r->tri_code; b.tile_list = r->tile_list;
      b.idx_out = d_idx; b.depth_out = d_depth;
      SMESH_HIP(hipMemsetAsync(b.tile_count, 0, ntiles * 4, st));
      const dim3 gtri((uint32_t)div_up(r->F, 256));
      hipLaunchKernelGGL(k_bin_count, gtri, dim3(256), 0, st, b);
      hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, st, b);
      hipLaunchKernelGGL(k_bin_fill, gtri, dim3(256), 0, st, b);
      hipLaunchKernelGGL(k_raster_big, dim3(big_grid), dim3(256), 0, st, a, 0u);
      hipLaunchKernelGGL(k_raster_tile, dim3((uint32_t)ntiles), dim3(256), 0, st, b);
      SMESH_HIP(hipGetLastError());
      return SMESH_OK;
    }
    hipLaunchKernelGGL(k_raster_small, dim3((uint32_t)div_up(r->F, 256)), dim3(256), 0, st, a);
    SMESH_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_raster_big, dim3(big_grid), dim3(256), 0, st, a, 0u);
    SMESH_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_resolve, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, r->keys, d_idx, d_depth, (uint32_t)W, (uint32_t)H, r->big_count);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

int acquire_image(smesh_renderer* r, uint64_t N, ImagePair** out) {
  for (auto& im : r->images)
    if (!im.idx_out && !im.depth_out && im.pixels == N) { *out = &im; return SMESH_OK; }
  // drop idle planes of other sizes (resolution changed)
  for (size_t i = 0; i < r->images.size();) {
    ImagePair& im = r->images[i];
    if (!im.idx_out && !im.depth_out) {
      SMESH_HIP(hipStreamSynchronize(r->ctx->stream));
      (void)hipFree(im.idx);
      (void)hipFree(im.depth);
      r->images.erase(r->images.begin() + i);
    } else {
      i++;
    }
  }
  ImagePair im{nullptr, nullptr, N, false, false};
  SMESH_HIP(hipMalloc(reinterpret_cast<void**>(&im.idx), N * 4));
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&im.depth), N * 4);
  if (e != hipSuccess) { (void)hipFree(im.idx); return fail_hip(e, "hipMalloc depth plane", __FILE__, __LINE__); }
  r->images.push_back(im);
  *out = &r->images.back();
  return SMESH_OK;
}

int check_camera(const smesh_camera_t* cam) {
  if (!cam) return fail(SMESH_ERR_INVALID, "camera is NULL");
  if (cam->width == 0 || cam->height == 0 || cam->width > 65536 || cam->height > 65536)
    return fail(SMESH_ERR_INVALID, "camera resolution must be in [1, 65536]");
  return SMESH_OK;
}

int create_common(const float* vertices, uint64_t V, const int32_t* faces, uint64_t F, int device,
                  smesh_renderer** out) {
  if (!out) return fail(SMESH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if ((V && !vertices) || (F && !faces)) return fail(SMESH_ERR_INVALID, "NULL mesh arrays");
  if (F >= 0xFFFFFFFFull || V >= 0x7FFFFFFFull) return fail(SMESH_ERR_INVALID, "mesh too large for 32-bit indices");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  SMESH_HIP(hipSetDevice(device));
  auto* r = new (std::nothrow) smesh_renderer();
  if (!r) return fail(SMESH_ERR_RUNTIME, "out of memory");
  r->ctx = ctx; r->V = V; r->F = F; r->num_primitives = F;
  r->big_capacity = (uint32_t)std::max<uint64_t>(F, 1);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&r->verts), std::max<uint64_t>(V * 12, 16));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->faces), std::max<uint64_t>(F * 12, 16));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->sv), std::max<uint64_t>(V * sizeof(ScreenVertex), 16));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->big_queue), (size_t)r->big_capacity * 4);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->big_count), 16);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->frags), std::max<uint64_t>(F * sizeof(TriFrag), 16));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->tri_code), std::max<uint64_t>(F * 4, 16));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->tile_list), std::max<uint64_t>(F * 4 * kMaxTilesPerTri, 16));
  if (e == hipSuccess) e = hipMemsetAsync(r->big_count, 0, 16, ctx->stream);
  if (e == hipSuccess && V) e = hipMemcpyAsync(r->verts, vertices, V * 12, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && F) e = hipMemcpyAsync(r->faces, faces, F * 12, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) {
    smesh_renderer_destroy(r);
    return fail_hip(e, "renderer allocation/upload", __FILE__, __LINE__);
  }
  *out = r;
  return SMESH_OK;
}

}  // namespace

static thread_local const char* g_last_fuse_kernel = "none";

extern "C" {

const char* smesh_last_fuse_kernel(void) { return g_last_fuse_kernel; }

int smesh_renderer_create_triangles(const float* vertices, uint64_t V, const int32_t* faces, uint64_t F, int device,
                                    smesh_renderer_t** out) {
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  return create_common(vertices, V, faces, F, device, out);
}

int smesh_renderer_create_texels(const float* vertices, uint64_t V, const int32_t* faces, uint64_t F,
                                 const smesh_camera_t* cameras, uint64_t K, float tpp, int device,
                                 smesh_renderer_t** out) {
  if (K && !cameras) return fail(SMESH_ERR_INVALID, "NULL cameras");
  if (K > 0xFFFFFFFFull) return fail(SMESH_ERR_INVALID, "too many cameras");
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);

  // "Optimally order face indices" (TexturedTriangleRenderer.h:129-146) runs on the host like the
  // reference's OpenMP loop: it is O(F), needs acos, and feeds the vertex order of every later render.
  std::vector<int32_t> hf(faces, faces + 3 * F);
  for (uint64_t f = 0; f < F; f++) {
    int32_t* face = &hf[3 * f];
    bool valid = true;
    for (int k = 0; k < 3; k++) if (face[k] < 0 || (uint64_t)face[k] >= V) valid = false;
    if (!valid) continue;
    float diffs[3];
    for (int k = 0; k < 3; k++) {
      const float* p0 = vertices + 3 * (size_t)face[k];
      const float* p1 = vertices + 3 * (size_t)face[(k + 1) % 3];
      const float* p2 = vertices + 3 * (size_t)face[(k + 2) % 3];
      const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
      const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
      const float dot = ax * bx + ay * by + az * bz;
      const float la = std::sqrt(ax * ax + ay * ay + az * az), lb = std::sqrt(bx * bx + by * by + bz * bz);
      float cosv = dot / (la * lb);
      cosv = cosv > 1.0f ? 1.0f : (cosv < -1.0f ? -1.0f : cosv);
      diffs[k] = std::fabs(std::acos(cosv) - 1.57079632679489661923f);
    }
    int best = 0;
    for (int k = 1; k < 3; k++) if (diffs[k] < diffs[best]) best = k;
    if (best != 0) { std::swap(face[0], face[best]); std::swap(diffs[0], diffs[best]); }
    if (diffs[1] >= diffs[2]) std::swap(face[1], face[2]);
  }

  smesh_renderer* r = nullptr;
  SMESH_TRY(create_common(vertices, V, hf.data(), F, device, &r));
  r->texels = true;
  r->h_faces = std::move(hf);
  r->h_res.assign(F, 0);
  r->h_first.assign(F, 0);

  // max projected area over all cameras: an F x K reduction, done on the GPU
  std::vector<float> best(F, 0.0f);
  if (F && K) {
    smesh_camera_t* d_cams = nullptr;
    float* d_best = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_cams), K * sizeof(smesh_camera_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_best), F * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cams, cameras, K * sizeof(smesh_camera_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_texel_area, dim3((uint32_t)div_up(F, 128)), dim3(128), 0, ctx->stream, r->verts, r->faces, F, V,
                         d_cams, (uint32_t)K, d_best);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(best.data(), d_best, F * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_cams) (void)hipFree(d_cams);
    if (d_best) (void)hipFree(d_best);
    if (e != hipSuccess) { smesh_renderer_destroy(r); return fail_hip(e, "texel area reduction", __FILE__, __LINE__); }
  }
  uint64_t total = 0;
  for (uint64_t f = 0; f < F; f++) {
    const uint32_t res = (uint32_t)std::ceil(tpp * std::sqrt(best[f]));   // :127
    r->h_res[f] = res;
    r->h_first[f] = (uint32_t)total;                                        // :149-162
    total += (uint64_t)res * (res + 1) / 2;                                 // getTexelNum, :43-47
  }
  if (total >= 0xFFFFFFFFull) { smesh_renderer_destroy(r); return fail(SMESH_ERR_INVALID, "texel count overflows uint32"); }
  r->num_primitives = total;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&r->tex_res), std::max<uint64_t>(F * 4, 16));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&r->tex_first), std::max<uint64_t>(F * 4, 16));
  if (e == hipSuccess && F) e = hipMemcpyAsync(r->tex_res, r->h_res.data(), F * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && F) e = hipMemcpyAsync(r->tex_first, r->h_first.data(), F * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { smesh_renderer_destroy(r); return fail_hip(e, "texel table upload", __FILE__, __LINE__); }
  *out = r;
  return SMESH_OK;
}

int smesh_renderer_destroy(smesh_renderer_t* r) {
  if (!r) return SMESH_OK;
  (void)hipSetDevice(r->ctx->device);
  (void)hipStreamSynchronize(r->ctx->raster_stream);
  (void)hipStreamSynchronize(r->ctx->stream);
  for (void* p : {(void*)r->verts, (void*)r->faces, (void*)r->sv, (void*)r->tex_res, (void*)r->tex_first, (void*)r->keys,
                  (void*)r->big_queue, (void*)r->big_count, (void*)r->tri_code, (void*)r->tile_list, (void*)r->frags})
    if (p) (void)hipFree(p);
  for (auto& im : r->images) { (void)hipFree(im.idx); (void)hipFree(im.depth); }
  r->own_idx.release();
  r->tile_tables.release();
  for (int i = 0; i < 2; i++) {
    r->fused[i].release();
    if (r->ev_rendered[i]) (void)hipEventDestroy(r->ev_rendered[i]);
    if (r->ev_consumed[i]) (void)hipEventDestroy(r->ev_consumed[i]);
  }
  delete r;
  return SMESH_OK;
}

int smesh_renderer_num_primitives(const smesh_renderer_t* r, uint64_t* out) {
  if (!r || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  *out = r->num_primitives;
  return SMESH_OK;
}

int smesh_renderer_texel_layout(const smesh_renderer_t* r, int32_t* faces_out, uint32_t* res_out, uint32_t* first_out) {
  if (!r) return fail(SMESH_ERR_INVALID, "NULL renderer");
  if (!r->texels) return fail(SMESH_ERR_INVALID, "not a texel renderer");
  if (faces_out && r->F) memcpy(faces_out, r->h_faces.data(), r->F * 12);
  if (res_out && r->F) memcpy(res_out, r->h_res.data(), r->F * 4);
  if (first_out && r->F) memcpy(first_out, r->h_first.data(), r->F * 4);
  return SMESH_OK;
}

int smesh_renderer_render_device(smesh_renderer_t* r, const smesh_camera_t* cam, uint32_t** indices_dev, float** depth_dev) {
  if (!r || !indices_dev || !depth_dev) return fail(SMESH_ERR_INVALID, "NULL argument");
  SMESH_TRY(check_camera(cam));
  std::lock_guard<std::mutex> g(r->mu);
  std::lock_guard<std::recursive_mutex> lock(r->ctx->mu);
  SMESH_HIP(hipSetDevice(r->ctx->device));
  ImagePair* im;
  SMESH_TRY(acquire_image(r, cam->width * cam->height, &im));
  SMESH_TRY(render_into(r, cam, im->idx, im->depth));
  im->idx_out = im->depth_out = true;
  *indices_dev = im->idx;
  *depth_dev = im->depth;
  return SMESH_OK;
}

int smesh_renderer_release_image(smesh_renderer_t* r, void* indices_dev, void* depth_dev) {
  if (!r) return fail(SMESH_ERR_INVALID, "NULL renderer");
  std::lock_guard<std::mutex> g(r->mu);
  for (auto& im : r->images) {
    if (indices_dev && im.idx == indices_dev) im.idx_out = false;
    if (depth_dev && im.depth == depth_dev) im.depth_out = false;
  }
  return SMESH_OK;
}

int smesh_renderer_render(smesh_renderer_t* r, const smesh_camera_t* cam, uint32_t* indices_out, float* depth_out) {
  if (!r || !indices_out) return fail(SMESH_ERR_INVALID, "NULL argument");
  SMESH_TRY(check_camera(cam));
  std::lock_guard<std::mutex> g(r->mu);
  DeviceCtx* ctx = r->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  const uint64_t N = cam->width * cam->height;
  SMESH_TRY(r->own_idx.reserve(N * 8));
  uint32_t* d_idx = static_cast<uint32_t*>(r->own_idx.ptr);
  float* d_depth = reinterpret_cast<float*>(d_idx + N);
  SMESH_TRY(render_into(r, cam, d_idx, d_depth));
  SMESH_HIP(hipMemcpyAsync(indices_out, d_idx, N * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (depth_out) SMESH_HIP(hipMemcpyAsync(depth_out, d_depth, N * 4, hipMemcpyDeviceToHost, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

// One iteration of the driver loop (colorize_cityscapes_mesh.py:54-67): render + add, indices never leave HBM.
// Asynchronous for DEVICE probs: they must stay valid until smesh_synchronize().
int smesh_fuse_view(smesh_renderer_t* r, smesh_aggregator_t* a, const smesh_camera_t* cam, const float* probs,
                    const float* weights, int memkind) {
  if (!r || !a || !probs) return fail(SMESH_ERR_INVALID, "NULL argument");
  SMESH_TRY(check_camera(cam));
  DeviceCtx* ctx = r->ctx;
  if (smesh_aggregator_ctx(a) != ctx) return fail(SMESH_ERR_INVALID, "renderer and aggregator live on different devices");
  std::lock_guard<std::mutex> g(r->mu);
  std::lock_guard<std::mutex> g2(smesh_aggregator_mutex(a));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  const uint64_t W = cam->width, H = cam->height, N = W * H;
  // Optional two-stage pipeline over two HIP streams (SMESH_FUSE_PIPELINE=1): the rasteriser of this view
  // runs on the raster stream while the main stream is still fusing the previous view; events hand the index
  // image over and back.  Measured on cfg2 it gains only ~2 % (5358 vs 5240 views/s): the rasteriser's 64-bit
  // atomics and the scatter-add's float atomics contend for the same memory-side path and each kernel gets
  // slower by about what the overlap saves, so it is off by default (and kernel timings stay clean).
  static const bool pipelined = getenv("SMESH_FUSE_PIPELINE") && atoi(getenv("SMESH_FUSE_PIPELINE")) != 0;
  const int slot = pipelined ? (int)(r->fused_seq & 1u) : 0;
  hipStream_t rst = pipelined ? ctx->raster_stream : ctx->stream;
  if (pipelined) {
    for (int i = 0; i < 2; i++) {
      if (!r->ev_rendered[i]) SMESH_HIP(hipEventCreateWithFlags(&r->ev_rendered[i], hipEventDisableTiming));
      if (!r->ev_consumed[i]) SMESH_HIP(hipEventCreateWithFlags(&r->ev_consumed[i], hipEventDisableTiming));
    }
    if (r->main_pending) {   // a plain render() used the renderer's scratch on the main stream
      SMESH_HIP(hipStreamSynchronize(ctx->stream));
      r->main_pending = false;
    }
  } else if (r->raster_pending) {
    SMESH_HIP(hipStreamSynchronize(ctx->raster_stream));
    r->raster_pending = false;
  }
  if (r->fused[slot].bytes < N * 8) {
    // growing a slot frees the old buffer: nothing may still be reading it
    SMESH_HIP(hipStreamSynchronize(ctx->raster_stream));
    SMESH_HIP(hipStreamSynchronize(ctx->stream));
    SMESH_TRY(r->fused[slot].reserve(N * 8));
  } else if (pipelined && r->fused_seq >= 2) {
    SMESH_HIP(hipStreamWaitEvent(ctx->raster_stream, r->ev_consumed[slot], 0));   // view k-2 has been fused
  }
  uint32_t* d_idx = static_cast<uint32_t*>(r->fused[slot].ptr);
  float* d_depth = reinterpret_cast<float*>(d_idx + N);
  SMESH_TRY(render_into(r, cam, d_idx, d_depth, rst));
  if (pipelined) {
    r->raster_pending = true;
    SMESH_HIP(hipEventRecord(r->ev_rendered[slot], ctx->raster_stream));
    SMESH_HIP(hipStreamWaitEvent(ctx->stream, r->ev_rendered[slot], 0));
  }
  const float* d_probs = probs;
  const float* d_w = weights;
  if (memkind == SMESH_MEM_HOST) {
    const uint32_t C = smesh_aggregator_classes(a);
    Scratch& sp = smesh_aggregator_stage_probs(a);
    SMESH_TRY(sp.reserve(N * C * 4));
    SMESH_HIP(hipMemcpyAsync(sp.ptr, probs, N * C * 4, hipMemcpyHostToDevice, ctx->stream));
    d_probs = static_cast<const float*>(sp.ptr);
    if (weights) {
      Scratch& sw = smesh_aggregator_stage_w(a);
      SMESH_TRY(sw.reserve(N * 4));
      SMESH_HIP(hipMemcpyAsync(sw.ptr, weights, N * 4, hipMemcpyHostToDevice, ctx->stream));
      d_w = static_cast<const float*>(sw.ptr);
    }
    SMESH_HIP(hipStreamSynchronize(ctx->stream));   // the caller may reuse its host arrays once we return
  }
  static const bool tiled_raster = getenv("SMESH_RASTER") && std::string(getenv("SMESH_RASTER")) == "tiled";
  if (!r->texels && !tiled_raster && smesh_aggregator_can_fuse_triangles(a, r->F)) {
    // triangle primitives: every accumulator row is owned by its triangle's lane -- no atomics, no histogram
    SMESH_TRY(smesh_aggregator_fuse_triangles(a, r->frags, r->F, r->big_queue, r->big_count + 1, r->big_capacity,
                                              d_idx, d_probs, d_w, H));
    g_last_fuse_kernel = "k_fuse_tri";
  } else {
    g_last_fuse_kernel = "k_scatter_strip";
    SMESH_TRY(smesh_aggregator_add_device_contig(a, d_idx, d_probs, d_w, W, H));
  }
  if (pipelined) SMESH_HIP(hipEventRecord(r->ev_consumed[slot], ctx->stream));
  r->fused_seq++;
  return SMESH_OK;
}

}  // extern "C"
