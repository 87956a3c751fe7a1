// raster.hip -- triangle / texel rasteriser for MI355X (gfx950).
//
// Replaces (citations relative to /root/reference):
//   include/semantic_meshes/render/TriangleRenderer.h:30-39   mesh upload            -> smesh_renderer_create_triangles
//   include/semantic_meshes/render/TriangleRenderer.h:75-78   clear {+inf, -1}       -> nothing to clear: k_tile_resolve starts every tile from the background key
//   include/semantic_meshes/render/TriangleRenderer.h:81-88   DeviceMutexRasterizer  -> k_project_vertices + k_raster_frag + k_tile_resolve
//   python/semantic_meshes/include/Renderer.h:32-35           AoS -> SoA split       -> k_tile_resolve writes the two planes directly
//
// Default path: per-tile fragment queues + depth test in LDS (k_raster_frag / k_tile_resolve, see the banner further
// down).  Kept beside it: the direct path (k_raster_small / k_raster_big / k_resolve: one global 64-bit atomicMin per
// fragment on a key image), used for images with too many tiles for the queues and with SMESH_RASTER=direct.
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
#include <utility>
#include <vector>

#pragma clang fp contract(off)

using namespace smesh;

// from fusion.hip
struct smesh_aggregator;
int smesh_aggregator_add_device_contig(smesh_aggregator* a, const uint32_t* d_idx, const float* d_probs,
                                       const float* d_w, uint64_t W, uint64_t H);
bool smesh_aggregator_can_fuse_triangles(smesh_aggregator* a, uint64_t F);
bool smesh_aggregator_can_fuse_pair(smesh_aggregator* a);
int smesh_aggregator_max_fused_views(smesh_aggregator* a);
bool smesh_aggregator_takes_strided_probs(smesh_aggregator* a, int64_t ps0, int64_t ps1, int nviews);
int smesh_aggregator_dense_probs(smesh_aggregator* a, const float* d_probs, const int64_t ps[3], uint64_t W, uint64_t H, const float** out);
bool smesh_aggregator_fuses_small_views_by_mask(smesh_aggregator* a);
int smesh_aggregator_fuse_triangles(smesh_aggregator* a, uint64_t F, const uint32_t* prim_id, uint32_t big_capacity,
                                    const RenderedView* views, int nviews, int part = 0, int nparts = 1);
void smesh_fuse_part_rows(uint64_t F, int part, int nparts, uint64_t* f_lo, uint64_t* f_hi);
uint64_t smesh_aggregator_primitives(smesh_aggregator* a);
int smesh_aggregator_join_exchange(smesh_aggregator* a);
const char* smesh_aggregator_fuse_kernel_name(smesh_aggregator* a, bool reordered);
bool smesh_aggregator_can_fuse_texels(smesh_aggregator* a, uint64_t P);
int smesh_aggregator_fuse_texels(smesh_aggregator* a, const TriFrag* frags, const uint8_t* kinds, uint64_t F, const uint32_t* tex_first,
                                 const uint32_t* tex_res, const uint32_t* big_queue, const uint32_t* big_len,
                                 uint32_t big_capacity, const uint32_t* d_idx, const float* d_probs, const float* d_w, uint64_t H);
int smesh_aggregator_fuse_texels_multi(smesh_aggregator* a, uint64_t F, const uint32_t* tex_first, const uint32_t* tex_res, uint32_t big_capacity,
                                       const RenderedView* views, int nviews);
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
  double iz;    // 1/z_c > 0: in front of the near plane; kBehind: finite, at or behind it; 0 = unusable vertex
};
constexpr double kBehind = -1.0;

struct CameraArgs {
  float R[9];
  float t[3];
  double fx, fy, cx, cy;
  uint32_t W, H;
};

// ---- vertex stage: float32 rigid transform, double pinhole projection (render/Camera.h:9-13) --------
// Also opens the render: empties the big-triangle queue (its length stays readable until the next render).
__device__ __forceinline__ ScreenVertex project_point(const CameraArgs& cam, const float X, const float Y, const float Z) {
  const float xc = ((cam.R[0] * X + cam.R[1] * Y) + cam.R[2] * Z) + cam.t[0];
  const float yc = ((cam.R[3] * X + cam.R[4] * Y) + cam.R[5] * Z) + cam.t[1];
  const float zc = ((cam.R[6] * X + cam.R[7] * Y) + cam.R[8] * Z) + cam.t[2];
  ScreenVertex s;
  s.u = 0.0; s.v = 0.0; s.iz = 0.0;
  const bool fin = isfinite(xc) && isfinite(yc) && isfinite(zc);
  if (zc > kNear && fin) {
    const double zd = (double)zc;
    const double u = cam.fx * ((double)xc / zd) + cam.cx;
    const double v = cam.fy * ((double)yc / zd) + cam.cy;
    if (isfinite(u) && isfinite(v)) {
      s.u = u; s.v = v; s.iz = 1.0 / zd;
    }
  } else if (fin) {
    s.iz = kBehind;   // at or behind the near plane: triangles that also have a vertex in front are clipped (clip_piece)
  }
  return s;
}

__device__ __forceinline__ void project_vertex(const float* __restrict__ verts, uint64_t V, const CameraArgs& cam,
                                               ScreenVertex* __restrict__ sv, uint32_t* __restrict__ big_count, const uint64_t i) {
  if (i == 0) { big_count[0] = 0u; big_count[1] = 0u; big_count[2] = 0u; big_count[3] = 0u; big_count[4] = 0u; big_count[5] = 0u; }   // [1]: "the per-triangle masks are not final" flag
  if (i >= V) return;
  sv[i] = project_point(cam, verts[3 * i + 0], verts[3 * i + 1], verts[3 * i + 2]);
}

__global__ void k_project_vertices(const float* __restrict__ verts, uint64_t V, CameraArgs cam,
                                   ScreenVertex* __restrict__ sv, uint32_t* __restrict__ big_count) {
  project_vertex(verts, V, cam, sv, big_count, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Up to kMaxGroup views of the same mesh in one launch (smesh_fuse_views): a thread reads its vertex once and projects it for
// every camera of the group.
constexpr int kMaxGroup = 8;
struct ProjectGroup {
  const float* verts;
  uint64_t V;
  CameraArgs cam[kMaxGroup];
  ScreenVertex* sv[kMaxGroup];
  uint32_t* big_count[kMaxGroup];
  uint32_t n;
};
__global__ void k_project_vertices_group(ProjectGroup g) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0)
    for (uint32_t v = 0; v < g.n; v++)
      for (int k = 0; k < 6; k++) g.big_count[v][k] = 0u;      // ([4]: length of the view's list of medium triangles, RasterArgs::med_queue)
  if (i >= g.V) return;
  const float X = g.verts[3 * i + 0], Y = g.verts[3 * i + 1], Z = g.verts[3 * i + 2];
  for (uint32_t v = 0; v < g.n; v++) g.sv[v][i] = project_point(g.cam[v], X, Y, Z);
}

// ---- triangle setup --------------------------------------------------------------------------------
// Edge through two screen points as the linear form E(p) = A px + B py + C, the coefficients taken from canonically ordered
// endpoints (both triangles sharing the edge get the same |value|, so a sample is claimed by exactly one of them, with the
// tie rule below) and evaluated as fma(A, px, fma(B, py, C)): two FP64 instructions per edge and sample, each one IEEE
// rounding -- the oracle's Edge::setup / Edge::eval, operation for operation.
struct EdgeEq {
  double A, B, C;
};

__device__ __forceinline__ EdgeEq make_edge(double ax, double ay, double bx, double by) {
  // The oracle's Edge::setup with its selects folded away.  With l the canonically lower endpoint, h the other and
  // sign = -1 if they were swapped: h - l = sign * (b - a) exactly (IEEE subtraction is antisymmetric), and sign changes pass
  // through every rounding, so
  //   A = sign * -(hy - ly) = ay - by,   B = sign * (hx - lx) = bx - ax,
  //   C = sign * fma(hy - ly, lx, -((hx - lx) * ly)) = fma(by - ay, lx, -((bx - ax) * ly)):
  // only the lower endpoint itself has to be selected.  Bit-identical to the oracle (the parity tests compare every pixel).
  EdgeEq e;
  const bool sw = (bx < ax) || (bx == ax && by < ay);
  const double lx = sw ? bx : ax, ly = sw ? by : ay;
  e.A = ay - by;
  e.B = bx - ax;
  const double c0 = e.B * ly;
  e.C = __builtin_fma(-e.A, lx, -c0);
  return e;
}

__device__ __forceinline__ double eval_edge(const EdgeEq& e, double px, double py) {
  return __builtin_fma(e.A, px, __builtin_fma(e.B, py, e.C));
}

__device__ __forceinline__ void flip_edge(EdgeEq& e) { e.A = -e.A; e.B = -e.B; e.C = -e.C; }

struct Tri {
  EdgeEq e0, e1, e2;   // e_i is opposite vertex i; oriented so that interior samples have positive values
  int cls0, cls1, cls2;   // v_cmp_class masks of the accepted edge values: positive, plus the zeros on an owned edge
  double iz0, iz1, iz2;
  int x0, x1, y0, y1;
};

// Tie rule: a sample exactly on the edge belongs to the triangle whose (oriented) edge normal satisfies A > 0 || (A == 0 && B > 0).
__device__ __forceinline__ bool owns(const EdgeEq& e) { return e.A > 0.0 || (e.A == 0.0 && e.B > 0.0); }
// The coverage test of one edge, w > 0 || (w == 0 && owned), as ONE instruction: v_cmp_class_f64 with the set of accepted
// classes (+normal, +denormal, +inf; with ownership also +0 and -0).  NaN is in no set, as it fails both comparisons.
constexpr int kClsPositive = 0x380, kClsZero = 0x060;
__device__ __forceinline__ int edge_classes(const EdgeEq& e) { return owns(e) ? (kClsPositive | kClsZero) : kClsPositive; }
__device__ __forceinline__ bool edge_accepts(double w, int cls) { return __builtin_amdgcn_class(w, cls); }

__device__ __forceinline__ bool setup_tri(const ScreenVertex& a, const ScreenVertex& b, const ScreenVertex& c,
                                          uint32_t W, uint32_t H, Tri& t) {
  if (!(a.iz > 0.0) || !(b.iz > 0.0) || !(c.iz > 0.0)) return false;
  const double minu = fmin(a.u, fmin(b.u, c.u)), maxu = fmax(a.u, fmax(b.u, c.u));
  const double minv = fmin(a.v, fmin(b.v, c.v)), maxv = fmax(a.v, fmax(b.v, c.v));
  // first / last sample column and row, clamped to the image (the conversions saturate: coordinates are finite here)
  t.x0 = max((int)ceil(minu - 0.5), 0); t.x1 = min((int)floor(maxu - 0.5), (int)W - 1);
  t.y0 = max((int)ceil(minv - 0.5), 0); t.y1 = min((int)floor(maxv - 0.5), (int)H - 1);
  if (t.x0 > t.x1 || t.y0 > t.y1) return false;
  t.e0 = make_edge(b.u, b.v, c.u, c.v);
  t.e1 = make_edge(c.u, c.v, a.u, a.v);
  t.e2 = make_edge(a.u, a.v, b.u, b.v);
  const double area2 = eval_edge(t.e2, c.u, c.v);
  if (!(area2 != 0.0) || !isfinite(area2)) return false;
  if (area2 < 0.0) { flip_edge(t.e0); flip_edge(t.e1); flip_edge(t.e2); }   // no back-face culling: orient instead
  t.cls0 = edge_classes(t.e0);
  t.cls1 = edge_classes(t.e1);
  t.cls2 = edge_classes(t.e2);
  t.iz0 = a.iz; t.iz1 = b.iz; t.iz2 = c.iz;
  return true;
}

// Coverage + depth of one sample; returns false if the sample is not covered.
// Values (no out-pointers: address-taken locals ended up in scratch memory).
struct Shaded { bool ok; float z; double b1, b2; };
__device__ __forceinline__ Shaded shade(const Tri& t, int x, int y, bool want_bary) {
  Shaded r; r.ok = false; r.z = 0.0f; r.b1 = 0.0; r.b2 = 0.0;
  const double px = (double)x + 0.5, py = (double)y + 0.5;
  const double w0 = eval_edge(t.e0, px, py);
  if (!edge_accepts(w0, t.cls0)) return r;
  const double w1 = eval_edge(t.e1, px, py);
  if (!edge_accepts(w1, t.cls1)) return r;
  const double w2 = eval_edge(t.e2, px, py);
  if (!edge_accepts(w2, t.cls2)) return r;
  const double num = (w0 + w1) + w2;
  const double den = __builtin_fma(w2, t.iz2, __builtin_fma(w1, t.iz1, w0 * t.iz0));
  const float zf = (float)(num / den);
  if (!(zf > 0.0f) || !isfinite(zf)) return r;
  r.ok = true; r.z = zf;
  if (want_bary) { r.b1 = w1 / num; r.b2 = w2 / num; }
  return r;
}

// The coverage part of shade() alone (same expressions, same order).
__device__ __forceinline__ bool covered(const Tri& t, int x, int y) {
  const double px = (double)x + 0.5, py = (double)y + 0.5;
  const double w0 = eval_edge(t.e0, px, py);
  if (!edge_accepts(w0, t.cls0)) return false;
  const double w1 = eval_edge(t.e1, px, py);
  if (!edge_accepts(w1, t.cls1)) return false;
  const double w2 = eval_edge(t.e2, px, py);
  return edge_accepts(w2, t.cls2);
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

// Fragment queues of the default path: one queue per 32 x 64 pixel screen tile (see k_raster_frag / k_tile_resolve).
constexpr int kQW = 32, kQH = 64, kQPixels = kQW * kQH;
constexpr int kQSub = 4;      // sub-queues per tile: a wave appends to sub-queue (wave id % kQSub).  Returning atomics on ONE address are
                              // served one after the other: with a single counter per tile the ~70 reservations a cfg2 tile receives per
                              // view cost k_raster_frag 19 of its 44 us.  Measured 2 / 4 / 8 / 16 sub-queues: k_raster_frag 34.5 / 33.1 /
                              // 32.7 / 32.0 us, k_tile_resolve (which reads them back) 8.2 / 9.0 / 10.7 / 15.1 us.
constexpr int kMedium = 64;   // boxes up to kMedium x kMedium (and larger than 8 x 8) are rasterised by a whole wave inside k_raster_frag
constexpr unsigned long long kNullKey = ~0ull;   // loses every depth test, including against the background key

struct FragQueues {
  unsigned long long* key = nullptr;   // [ntiles * kQSub * cap] depth-test keys
  uint16_t* pix = nullptr;             // [ntiles * kQSub * cap] pixel inside the tile: (x - tile_x0) * kQH + (y - tile_y0)
  uint32_t* count = nullptr;           // [ntiles * kQSub] fragments queued per sub-queue (may exceed cap: the excess went to the key image)
  uint32_t* flag = nullptr;            // [ntiles] nonzero: the global key image holds fragments of this tile
  uint32_t cap = 0, tiles_y = 0;       // cap: slots per SUB-queue
};

struct RasterArgs {
  const int32_t* faces;
  const float* verts;         // [V][3] world-space vertices: re-read by the (rare) triangles that cross the near plane
  CameraArgs cam;             // ... together with the camera, to cut them along it (clip_piece)
  const ScreenVertex* sv;
  const uint32_t* prim_id;    // [F] primitive id of the triangle at position f (null: id == f; the renderer re-orders badly ordered meshes)
  const uint32_t* tex_res;    // null for triangle primitives
  const uint32_t* tex_first;
  unsigned long long* keys;
  uint64_t F, V;
  uint32_t W, H;
  uint32_t* big_queue;        // triangles with a box larger than 8 x 8 (the fusion's cooperative waves walk it)
  uint32_t* huge_queue;       // ... of those, the ones larger than kMedium x kMedium: rasterised by the tile workgroups
  uint32_t* big_count;        // [0] length of big_queue, [2] length of huge_queue, [1] nonzero: the masks of the per-triangle
                              //     records still hold fragments that lost the depth test (the fusion must check them)
  uint32_t big_capacity;
  TriFrag* frags;             // per-triangle fragment records for the triangle-order fusion (may be null)
  uint8_t* kinds;             // texel renderers: the records' `kind`, a byte per triangle -- the fusion finds the one triangle in eight that
                              // emitted anything from these instead of from sixteen bytes per triangle and view, and a record is
                              // written only where the byte is nonzero; else null
  FragQueues q;               // fragment-queue path only
  uint32_t tpw;               // k_raster_frag: triangles per group of a wave (power of two <= 64)
  uint32_t groups;            // k_raster_frag: groups per wave (1; 2 with tpw = 64 when the launch has waves to spare)
  uint32_t* med_queue;        // k_raster_frag, balanced (below): the view's list of triangles for the cooperative loop, [big_capacity];
                              // its length: big_count[4]
  uint32_t balance;           // k_raster_frag (wg_push instances): nonzero = the triangles a wave would walk one at a time (boxes beyond the
                              // lanes' reach, up to kMedium) go to med_queue and k_raster_medium walks that list with the whole
                              // chip: a view from inside the scene has them all in a few waves (the near part of the mesh);
                              // 2 (one view per launch) = so do the triangles the lanes would walk in rounds (boxes of 9 ..
                              // kLaneBox pixels; from the far end of med_queue, count in big_count[5]): seven per wave and turn
  uint32_t wg_push;           // k_raster_frag: nonzero = the workgroup's four waves share each atomic on the view's queues (views of mostly medium
                              // triangles: every wave pushes, and increments of ONE counter pass the L2 at ~5 ns each); 0 = one per wave
  uint32_t spread;            // k_raster_frag: nonzero = a triangle takes kSpread consecutive lanes, one per 8 x 8 sub-box of its (at most
                              // kLaneBox x kLaneBox) box; tpw = 64 / kSpread (raster_frag_wave)
  uint32_t idx_optional;      // k_tile_resolve: nonzero = the index plane is wanted only by the fusion's fallback paths (smesh_fuse_view(s) of a
                              // triangle renderer in its own face order: k_fuse_tri* read the plane for queued triangles -- big_count[0] -- and
                              // when the masks need checking -- big_count[1]): a view with neither does not write it (8.3 MB per 1080p view).
                              // 2 = each view of a group for itself, 1 = one decision for the group's launch (plane_optional_level)
  int dbg;                    // development ablation (SMESH_RDBG) of k_raster_frag: 1 = no stores, 2 = setup only, 4 = + coverage,
                              // 8 = + slot reservation, 16 = grouping without the reservation atomics
};

// Key image layout: same (W,H) y-fastest order as the output planes.  (A 4 x 4 blocked layout was tried to
// make neighbouring fragments share cache lines: no change in k_raster_small, slower resolve.)
__device__ __forceinline__ uint64_t key_index(uint32_t x, uint32_t y, uint32_t H) { return (uint64_t)x * H + y; }

__device__ __forceinline__ bool load_tri(const RasterArgs& a, uint64_t f, Tri& t, const int32_t i0, const int32_t i1, const int32_t i2);
__device__ __forceinline__ bool load_tri(const RasterArgs& a, uint64_t f, Tri& t) {
  return load_tri(a, f, t, a.faces[3 * f + 0], a.faces[3 * f + 1], a.faces[3 * f + 2]);
}
// (vertex indices already in registers: k_raster_frag requests those of its next 64 triangles ahead of time)
__device__ __forceinline__ bool load_tri(const RasterArgs& a, uint64_t f, Tri& t, const int32_t i0, const int32_t i1, const int32_t i2) {
  // 0 <= i < V for all three, as one unsigned comparison (negative indices are huge as uint32; V <= 2^31 since indices are int32)
  if ((uint64_t)max((uint32_t)i0, max((uint32_t)i1, (uint32_t)i2)) >= a.V) return false;
  if (a.tex_res && a.tex_res[f] == 0) return false;
  // all three vertices are fetched before any of them is tested: one memory round trip instead of three
  const ScreenVertex va = a.sv[i0], vb = a.sv[i1], vc = a.sv[i2];
  return setup_tri(va, vb, vc, a.W, a.H, t);
}

// ---- near-plane clipping (round 3; DESIGN.md "Raster spec" 2; oracle: clip_edge / clip_triangle, operation for operation) --------
// A triangle with vertices in front of AND at / behind the near plane z_c = kNear is cut along it.  On every edge from a front
// vertex F to a behind vertex B the point I = F + t (B - F), t = (zF - n) / (zF - zB), is computed in double FROM THE FRONT VERTEX
// (the two triangles sharing the edge compute the same I: the cut edge F-I stays watertight) and projected with z = n exactly.
// One front vertex (cyclic order F, N, P): the piece (F, I_FN, I_FP).  Two front vertices (cyclic order B, N, P): the pieces
// (N, P, I_PB) and (N, I_PB, I_NB).  Pieces are ordinary triangles for setup_tri / the coverage rule / the depth formula and write
// the id of the triangle they came from; texel primitives interpolate the ORIGINAL triangle's (b1, b2), carried at the pieces'
// vertices, perspective-correctly.  Such triangles are rare (the ring of floor / wall triangles around a camera inside a room):
// they take the path of the largest triangles -- the fusion's big-triangle queue and the tile workgroups' queue -- whatever
// their size, and re-read their world-space vertices instead of widening the per-vertex record of every vertex.
struct CamPoint { float x, y, z; };
__device__ __forceinline__ CamPoint camera_point(const CameraArgs& cam, const float* __restrict__ verts, const int32_t i) {
  const float X = verts[3 * (uint64_t)i + 0], Y = verts[3 * (uint64_t)i + 1], Z = verts[3 * (uint64_t)i + 2];
  CamPoint c;
  c.x = ((cam.R[0] * X + cam.R[1] * Y) + cam.R[2] * Z) + cam.t[0];
  c.y = ((cam.R[3] * X + cam.R[4] * Y) + cam.R[5] * Z) + cam.t[1];
  c.z = ((cam.R[6] * X + cam.R[7] * Y) + cam.R[8] * Z) + cam.t[2];
  return c;
}
struct ClipVertex { ScreenVertex s; double b1, b2; };
// The point where the edge from front vertex F to behind vertex B meets the near plane (false: it does not project).
__device__ __forceinline__ bool clip_edge(const CameraArgs& cam, const CamPoint& F, const double fb1, const double fb2,
                                          const CamPoint& B, const double bb1, const double bb2, ClipVertex& out) {
  const double zn = (double)kNear;
  const double t = ((double)F.z - zn) / ((double)F.z - (double)B.z);
  const double x = __builtin_fma(t, (double)B.x - (double)F.x, (double)F.x);
  const double y = __builtin_fma(t, (double)B.y - (double)F.y, (double)F.y);
  const double u = cam.fx * (x / zn) + cam.cx;
  const double v = cam.fy * (y / zn) + cam.cy;
  out.s.u = u; out.s.v = v; out.s.iz = 1.0 / zn;
  out.b1 = __builtin_fma(t, bb1 - fb1, fb1);
  out.b2 = __builtin_fma(t, bb2 - fb2, fb2);
  return isfinite(u) && isfinite(v);
}

// 0: all three vertices in front (an ordinary triangle); 1 / 2: crossing with that many front vertices; -1: nothing to draw
__device__ __forceinline__ int clip_class(const ScreenVertex& va, const ScreenVertex& vb, const ScreenVertex& vc) {
  if (va.iz == 0.0 || vb.iz == 0.0 || vc.iz == 0.0) return -1;
  const int nfront = (va.iz > 0.0 ? 1 : 0) + (vb.iz > 0.0 ? 1 : 0) + (vc.iz > 0.0 ? 1 : 0);
  return nfront == 3 ? 0 : (nfront == 0 ? -1 : nfront);
}

// The three vertices (with the original triangle's barycentric coordinates) of piece `p` (0 or 1) of a crossing triangle whose
// vertices are (i0, i1, i2) -> (va, vb, vc); nfront = clip_class() in {1, 2}.  False: there is no such piece.
__device__ __forceinline__ bool clip_piece(const RasterArgs& a, const int32_t i0, const int32_t i1, const int32_t i2,
                                           const ScreenVertex& va, const ScreenVertex& vb, const ScreenVertex& vc,
                                           const int nfront, const int p, ClipVertex& q0, ClipVertex& q1, ClipVertex& q2) {
  if (nfront == 1 && p != 0) return false;
  // rotate so that the special vertex (the only front one / the only behind one) comes first: cyclic order S, N, P
  const int k = nfront == 1 ? (va.iz > 0.0 ? 0 : (vb.iz > 0.0 ? 1 : 2)) : (!(va.iz > 0.0) ? 0 : (!(vb.iz > 0.0) ? 1 : 2));
  const int32_t is = k == 0 ? i0 : (k == 1 ? i1 : i2), in = k == 0 ? i1 : (k == 1 ? i2 : i0), ip = k == 0 ? i2 : (k == 1 ? i0 : i1);
  const CamPoint cs = camera_point(a.cam, a.verts, is), cn = camera_point(a.cam, a.verts, in), cp = camera_point(a.cam, a.verts, ip);
  // (b1, b2) of the original vertices a, b, c: (0,0), (1,0), (0,1)
  const double s1 = k == 1 ? 1.0 : 0.0, s2 = k == 2 ? 1.0 : 0.0;
  const double n1 = k == 0 ? 1.0 : 0.0, n2 = k == 1 ? 1.0 : 0.0;
  const double p1 = k == 2 ? 1.0 : 0.0, p2 = k == 0 ? 1.0 : 0.0;
  const ScreenVertex& ss = k == 0 ? va : (k == 1 ? vb : vc);
  const ScreenVertex& sn = k == 0 ? vb : (k == 1 ? vc : va);
  const ScreenVertex& sp = k == 0 ? vc : (k == 1 ? va : vb);
  if (nfront == 1) {
    q0.s = ss; q0.b1 = s1; q0.b2 = s2;
    const bool ok1 = clip_edge(a.cam, cs, s1, s2, cn, n1, n2, q1);
    const bool ok2 = clip_edge(a.cam, cs, s1, s2, cp, p1, p2, q2);
    return ok1 && ok2;
  }
  ClipVertex inb, ipb;
  const bool ok1 = clip_edge(a.cam, cn, n1, n2, cs, s1, s2, inb);
  const bool ok2 = clip_edge(a.cam, cp, p1, p2, cs, s1, s2, ipb);
  if (!ok1 || !ok2) return false;
  q0.s = sn; q0.b1 = n1; q0.b2 = n2;
  if (p == 0) { q1.s = sp; q1.b1 = p1; q1.b2 = p2; q2 = ipb; }
  else        { q1 = ipb; q2 = inb; }
  return true;
}

// Screen bounding box of what is left of a crossing triangle in front of the near plane (the box of its pieces' vertices, clamped
// to the image like setup_tri's): false if that is empty.
__device__ __forceinline__ bool clip_bbox(const RasterArgs& a, const int32_t i0, const int32_t i1, const int32_t i2,
                                          const ScreenVertex& va, const ScreenVertex& vb, const ScreenVertex& vc, const int nfront,
                                          int& x0, int& y0, int& x1, int& y1) {
  ClipVertex q0, q1, q2;
  if (!clip_piece(a, i0, i1, i2, va, vb, vc, nfront, nfront == 2 ? 1 : 0, q0, q1, q2)) return false;
  // one front vertex: the piece itself; two: piece 1 = (N, I_PB, I_NB) plus P
  double minu = fmin(q0.s.u, fmin(q1.s.u, q2.s.u)), maxu = fmax(q0.s.u, fmax(q1.s.u, q2.s.u));
  double minv = fmin(q0.s.v, fmin(q1.s.v, q2.s.v)), maxv = fmax(q0.s.v, fmax(q1.s.v, q2.s.v));
  if (nfront == 2) {
    const int k = !(va.iz > 0.0) ? 0 : (!(vb.iz > 0.0) ? 1 : 2);
    const ScreenVertex& sp = k == 0 ? vc : (k == 1 ? va : vb);
    minu = fmin(minu, sp.u); maxu = fmax(maxu, sp.u); minv = fmin(minv, sp.v); maxv = fmax(maxv, sp.v);
  }
  // (the conversions saturate; coordinates are finite here)
  x0 = max((int)ceil(minu - 0.5), 0); x1 = min((int)floor(maxu - 0.5), (int)a.W - 1);
  y0 = max((int)ceil(minv - 0.5), 0); y1 = min((int)floor(maxv - 0.5), (int)a.H - 1);
  return x0 <= x1 && y0 <= y1;
}

// A triangle as the big-triangle paths shade it: piece `p` of triangle f (piece 0 of an ordinary triangle is the triangle).
struct Piece {
  Tri t;
  double b1[3], b2[3];   // clipped pieces: the original triangle's barycentric coordinates at the piece's vertices
  bool clipped;
};
__device__ __forceinline__ bool load_piece(const RasterArgs& a, const uint64_t f, const int p, Piece& pc) {
  const int32_t i0 = a.faces[3 * f + 0], i1 = a.faces[3 * f + 1], i2 = a.faces[3 * f + 2];
  if ((uint64_t)max((uint32_t)i0, max((uint32_t)i1, (uint32_t)i2)) >= a.V) return false;
  if (a.tex_res && a.tex_res[f] == 0) return false;
  const ScreenVertex va = a.sv[i0], vb = a.sv[i1], vc = a.sv[i2];
  const int cls = clip_class(va, vb, vc);
  pc.clipped = cls > 0;
  if (cls < 0) return false;
  if (cls == 0) return p == 0 && setup_tri(va, vb, vc, a.W, a.H, pc.t);
  ClipVertex q0, q1, q2;
  if (!clip_piece(a, i0, i1, i2, va, vb, vc, cls, p, q0, q1, q2)) return false;
  pc.b1[0] = q0.b1; pc.b1[1] = q1.b1; pc.b1[2] = q2.b1;
  pc.b2[0] = q0.b2; pc.b2[1] = q1.b2; pc.b2[2] = q2.b2;
  return setup_tri(q0.s, q1.s, q2.s, a.W, a.H, pc.t);
}

// Depth-test key of piece `pc` of triangle f at sample (x, y), or kNullKey if the sample is not covered (shade() / shade_key() with
// the texel of a clipped piece taken from the perspective-correct barycentric coordinates).
__device__ __forceinline__ unsigned long long shade_piece_key(const RasterArgs& a, const uint64_t f, const Piece& pc, const int x, const int y) {
  const Tri& t = pc.t;
  const double px = (double)x + 0.5, py = (double)y + 0.5;
  const double w0 = eval_edge(t.e0, px, py);
  if (!edge_accepts(w0, t.cls0)) return kNullKey;
  const double w1 = eval_edge(t.e1, px, py);
  if (!edge_accepts(w1, t.cls1)) return kNullKey;
  const double w2 = eval_edge(t.e2, px, py);
  if (!edge_accepts(w2, t.cls2)) return kNullKey;
  const double num = (w0 + w1) + w2;
  const double den = __builtin_fma(w2, t.iz2, __builtin_fma(w1, t.iz1, w0 * t.iz0));
  const float zf = (float)(num / den);
  if (!(zf > 0.0f) || !isfinite(zf)) return kNullKey;
  uint32_t prim = a.prim_id ? a.prim_id[f] : (uint32_t)f;
  if (a.tex_res) {
    double b1, b2;
    if (!pc.clipped) { b1 = w1 / num; b2 = w2 / num; }
    else {
      const double q0 = w0 * t.iz0, q1 = w1 * t.iz1, q2 = w2 * t.iz2;
      b1 = __builtin_fma(q2, pc.b1[2], __builtin_fma(q1, pc.b1[1], q0 * pc.b1[0])) / den;
      b2 = __builtin_fma(q2, pc.b2[2], __builtin_fma(q1, pc.b2[1], q0 * pc.b2[0])) / den;
    }
    prim = a.tex_first[f] + texel_of(a.tex_res[f], b1, b2);
  }
  return ((unsigned long long)__float_as_uint(zf) << 32) | prim;
}

// load_tri for the one-lane-per-triangle kernels: 1 = an ordinary triangle, set up in `t`; 2 = a triangle that crosses the near
// plane, t.x0 .. t.y1 = the screen box of what is left of it (to be shaded piece by piece, load_piece); 0 = nothing to draw.
__device__ __forceinline__ int load_tri_ex(const RasterArgs& a, uint64_t f, Tri& t, const int32_t i0, const int32_t i1, const int32_t i2) {
  if ((uint64_t)max((uint32_t)i0, max((uint32_t)i1, (uint32_t)i2)) >= a.V) return 0;
  if (a.tex_res && a.tex_res[f] == 0) return 0;
  const ScreenVertex va = a.sv[i0], vb = a.sv[i1], vc = a.sv[i2];
  const int cls = clip_class(va, vb, vc);
  if (cls == 0) return setup_tri(va, vb, vc, a.W, a.H, t) ? 1 : 0;
  if (cls < 0) return 0;
  return clip_bbox(a, i0, i1, i2, va, vb, vc, cls, t.x0, t.y0, t.x1, t.y1) ? 2 : 0;
}

// Depth-test key of triangle f at sample (x, y), or kNullKey if the sample is not covered.
__device__ __forceinline__ unsigned long long shade_key(const RasterArgs& a, uint64_t f, const Tri& t, int x, int y) {
  const Shaded sh = shade(t, x, y, a.tex_res != nullptr);
  if (!sh.ok) return kNullKey;
  uint32_t prim = a.prim_id ? a.prim_id[f] : (uint32_t)f;
  if (a.tex_res) prim = a.tex_first[f] + texel_of(a.tex_res[f], sh.b1, sh.b2);
  return ((unsigned long long)__float_as_uint(sh.z) << 32) | prim;
}

__device__ __forceinline__ void emit(const RasterArgs& a, uint64_t f, const Tri& t, int x, int y) {
  const unsigned long long key = shade_key(a, f, t, x, y);
  if (key == kNullKey) return;
  if (SMESH_ABL(a.dbg) & 1) { if (key == 12345ull) a.keys[0] = key; return; }
  atomicMin(&a.keys[key_index((uint32_t)x, (uint32_t)y, a.H)], key);
}

// Medium triangles (a box over 8 x 8 of at most kMidBox pixels) are listed a second time, in the upper half of the big-triangle queue
// (entries big_capacity ..., count in big_count[3]): fuse_mid_entries (fuse_mid.inc.hpp, the last workgroups of a k_fuse_tri launch) walks that list and nothing else.
__device__ __forceinline__ void push_mid(const RasterArgs& a, const uint64_t f, const int box_pixels) {
  if (box_pixels > kMidBox) return;
  const uint32_t slot = atomicAdd(a.big_count + 3, 1u);
  if (slot < a.big_capacity) a.big_queue[(uint64_t)a.big_capacity + slot] = (uint32_t)f;
}

// Direct path.  One lane per triangle; triangles whose bounding box exceeds 8 x 8 pixels are queued for k_raster_big.
// Besides the depth-tested keys, each triangle leaves a TriFrag record (which pixels it emitted).
__global__ void k_raster_small(RasterArgs a) {
  const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= a.F) return;
  TriFrag rec;
  rec.x0 = 0; rec.y0 = 0; rec.kind = 0; rec.pad = 0; rec.mask = 0ull;
  Tri t;
  const int have = load_tri_ex(a, f, t, a.faces[3 * f + 0], a.faces[3 * f + 1], a.faces[3 * f + 2]);
  if (have) {
    const int bw = t.x1 - t.x0 + 1, bh = t.y1 - t.y0 + 1;
    rec.x0 = (uint16_t)t.x0; rec.y0 = (uint16_t)t.y0;
    if (bw > 8 || bh > 8 || have == 2) {
      const uint32_t slot = atomicAdd(a.big_count, 1u);
      if (slot < a.big_capacity) a.big_queue[slot] = (uint32_t)f;
      push_mid(a, f, bw * bh);
      rec.kind = 2;
      rec.mask = (unsigned long long)(uint32_t)t.x1 | ((unsigned long long)(uint32_t)t.y1 << 16);
    } else if (!(SMESH_ABL(a.dbg) & 2)) {
      unsigned long long mask = 0ull;
      for (int dx = 0; dx < bw; dx++)
        for (int dy = 0; dy < bh; dy++) {
          const unsigned long long key = shade_key(a, f, t, t.x0 + dx, t.y0 + dy);
          if (key != kNullKey) {
            mask |= 1ull << (dx * 8 + dy);
            if (!(SMESH_ABL(a.dbg) & 1)) atomicMin(&a.keys[key_index((uint32_t)(t.x0 + dx), (uint32_t)(t.y0 + dy), a.H)], key);
          }
        }
      rec.kind = mask ? 1 : 0;
      rec.mask = mask;
    }
  }
  if (a.frags && (rec.kind != 0 || !a.kinds)) a.frags[f] = rec;
  if (a.kinds) a.kinds[f] = (uint8_t)rec.kind;
}

// One workgroup per (queued triangle, 64x64 pixel chunk of its bounding box); lanes run down columns.
__global__ __launch_bounds__(256) void k_raster_big(RasterArgs a, uint32_t chunks_budget) {
  const uint32_t nbig = min(*a.big_count, a.big_capacity);
  // blocks walk the queue; each triangle is split into 64x64 chunks processed by successive blocks
  for (uint32_t q = blockIdx.x; q < nbig; q += gridDim.x) {
    const uint64_t f = a.big_queue[q];
    for (int p = 0; p < 2; p++) {   // the triangle, or the one or two pieces of a triangle that crosses the near plane
      Piece pc;
      if (!load_piece(a, f, p, pc)) continue;
      const Tri& t = pc.t;
      const int bw = t.x1 - t.x0 + 1, bh = t.y1 - t.y0 + 1;
      const int ty = threadIdx.x & 63, tx = threadIdx.x >> 6;  // 64 rows x 4 columns per pass
      for (int cx = 0; cx < bw; cx += 4) {
        const int x = t.x0 + cx + tx;
        if (x > t.x1) continue;
        for (int cy = 0; cy < bh; cy += 64) {
          const int y = t.y0 + cy + ty;
          if (y > t.y1) continue;
          const unsigned long long key = shade_piece_key(a, f, pc, x, y);
          if (key != kNullKey && !(SMESH_ABL(a.dbg) & 1)) atomicMin(&a.keys[key_index((uint32_t)x, (uint32_t)y, a.H)], key);
        }
      }
    }
  }
  (void)chunks_budget;
}

// ------------------------------------------------------------------------------------------------
// Fragment-queue path (default).  Measured on MI355X (tools/min_bench.hip): a 64-bit atomicMin costs ~39 ps per
// distinct 64-byte segment a wave instruction touches, i.e. 55 us for the 1.4 M scattered fragments of a cfg2
// view but 10 us when the lanes of an instruction hit consecutive keys.  One lane per triangle scatters by
// construction, so the depth test is moved out of global memory: k_raster_frag appends each fragment
// (key, pixel-in-tile) to the queue of its 32 x 64 pixel screen tile -- slots are reserved with ONE atomic per
// (wave, distinct tile), the stores of neighbouring lanes land next to each other -- and k_tile_resolve runs
// the depth test of a tile in LDS (ds_min_u64) and writes the output planes once, coalesced: no global atomic
// per fragment, no key image to clear or split.  Triangles with a box larger than 8 x 8 are rasterised by the whole
// wave (boxes up to 64 x 64) or by the tile workgroups they overlap (larger).  Only fragments beyond a queue's
// capacity still go through the global key image: the tiles they touch are flagged and merge (and re-arm) their
// part of it.
// ------------------------------------------------------------------------------------------------

// wave64 inclusive scan in registers: DPP row_shr 1,2,4,8 inside the 16-lane rows, then row_bcast:15 / row_bcast:31
// carry the row totals (semantics checked on MI355X by tools/dpp_test.hip).
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
  int s = (int)v;
  s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xF, 0xF, true);
  s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xF, 0xF, true);
  s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xF, 0xF, true);
  s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xF, 0xF, true);
  s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xA, 0xF, false);
  s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xC, 0xF, false);
  return (uint32_t)s;
}

// First half of a slot reservation.  A lane's fragments fall into the (at most 2 x 2) tiles T0, T0 + tiles_y, T0 + 1, T0 + tiles_y + 1,
// so lanes with the same T0 share all four: the lanes are grouped by T0 and every lane learns, per quadrant, its offset inside
// its group and the group's total, and the group's leader lane.  The four counts travel through TWO wave scans, packed in pairs
// of 16-bit halves (a wave holds at most 64 x 64 fragments per quadrant: no carry).  No memory traffic.  Whole wave must call.
struct Claim4 { uint32_t pre01, pre23, tot01, tot23; int leader; };
__device__ __forceinline__ Claim4 wave_claim_prepare4(uint32_t T0, uint32_t n01, uint32_t n23) {
  Claim4 c; c.pre01 = 0u; c.pre23 = 0u; c.tot01 = 0u; c.tot23 = 0u; c.leader = 0;
  const bool active = (n01 | n23) != 0u;
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)T0, first);
    const bool in = active && T0 == t0;
    const unsigned long long same = __ballot(in);
    const uint32_t v01 = in ? n01 : 0u, v23 = in ? n23 : 0u;
    const uint32_t s01 = wave_scan_incl(v01), s23 = wave_scan_incl(v23);
    const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane((int)s01, 63), t23 = (uint32_t)__builtin_amdgcn_readlane((int)s23, 63);
    if (in) { c.pre01 = s01 - v01; c.pre23 = s23 - v23; c.tot01 = t01; c.tot23 = t23; c.leader = first; }
    todo &= ~same;
  }
  return c;
}

// Coverage of the (at most 8 x 8) box at (X0, Y0), bw x bh samples, by triangle t: bit dx * 8 + dy per covered sample.  One flattened
// loop (trip count bw * bh, not max bw x max bh over the wave's lanes).  The edge functions are shade()'s, expression for expression.
__device__ __forceinline__ unsigned long long walk_box(const Tri& t, const int X0, const int Y0, const int bw, const int bh) {
  unsigned long long cover = 0ull;
  const int area = bw * bh;
  int dx = 0, dy = 0;
  const double py0 = (double)Y0 + 0.5;
  double px = (double)X0 + 0.5, py = py0;   // advanced by exact steps of 1.0
  for (int k = 0; k < area; k++) {
    const double w0 = eval_edge(t.e0, px, py);
    const double w1 = eval_edge(t.e1, px, py);
    const double w2 = eval_edge(t.e2, px, py);
    const bool in = edge_accepts(w0, t.cls0) && edge_accepts(w1, t.cls1) && edge_accepts(w2, t.cls2);
    if (in) cover |= 1ull << (dx * 8 + dy);
    py += 1.0;
    if (++dy == bh) { dy = 0; dx++; py = py0; px += 1.0; }
  }
  return cover;
}

// The covered samples `cover` (bit dx * 8 + dy) of the 8 x 8 box at (X0, Y0) go to the fragment queues: slot reservation in the (at
// most 2 x 2) tiles the box overlaps -- lanes grouped by their first tile, ONE atomic per (wave, tile) -- then depth per covered
// sample and the stores.  WHOLE WAVE must call (lanes without samples pass cover = 0).  Returns the mask of the samples that got a key.
// TEX (compile time, = a.tex_res != nullptr): with a run-time test the compiler computed the two divisions of the texel's barycentric
// coordinates for every fragment of every renderer and selected afterwards -- three FP64 divisions per fragment instead of one,
// a tenth of k_raster_frag's vector instructions at cfg2 (round 6).
template <bool TEX>
__device__ __forceinline__ unsigned long long emit_cover(const RasterArgs& a, const Tri& t, const uint64_t f, const uint32_t pid,
                                                         const int X0, const int Y0, unsigned long long cover, const uint32_t sub) {
  const int lane = threadIdx.x & 63;
  // split the box at the tile borders: columns dx < bx / rows dy < by belong to tile (tx0, ty0)
  const uint32_t tx0 = (uint32_t)X0 / kQW, ty0 = (uint32_t)Y0 / kQH;
  const int bx = (int)(tx0 + 1) * kQW - X0, by = (int)(ty0 + 1) * kQH - Y0;
  const unsigned long long lox = bx >= 8 ? ~0ull : ((1ull << (8 * bx)) - 1ull);
  const unsigned long long loy = by >= 8 ? ~0ull : (0x0101010101010101ull * ((1ull << by) - 1ull));
  const uint32_t T0 = tx0 * a.q.tiles_y + ty0;
  // quadrant q: bit 0 = next tile in x, bit 1 = next tile in y.  Group, then ALL reservations of the wave in
  // four back-to-back atomics per group leader (one round trip), then fetch the bases from the leaders.
  const uint32_t n0 = (uint32_t)__popcll(cover & lox & loy), n1 = (uint32_t)__popcll(cover & ~lox & loy);
  const uint32_t n2 = (uint32_t)__popcll(cover & lox & ~loy), n3 = (uint32_t)__popcll(cover & ~lox & ~loy);
  const Claim4 c = wave_claim_prepare4(T0, n0 | (n1 << 16), n2 | (n3 << 16));
  const uint32_t tot0 = c.tot01 & 0xFFFFu, tot1 = c.tot01 >> 16, tot2 = c.tot23 & 0xFFFFu, tot3 = c.tot23 >> 16;
  uint32_t g0 = 0u, g1 = 0u, g2 = 0u, g3 = 0u;
  if (!(SMESH_ABL(a.dbg) & 16) && cover != 0ull && lane == c.leader) {   // (ablation bit 16: grouping without the reservations)
    if (tot0) g0 = atomicAdd(&a.q.count[T0 * kQSub + sub], tot0);
    if (tot1) g1 = atomicAdd(&a.q.count[(T0 + a.q.tiles_y) * kQSub + sub], tot1);
    if (tot2) g2 = atomicAdd(&a.q.count[(T0 + 1u) * kQSub + sub], tot2);
    if (tot3) g3 = atomicAdd(&a.q.count[(T0 + a.q.tiles_y + 1u) * kQSub + sub], tot3);
  }
  // entry index (into key[] / pix[]) of this lane's next fragment in each quadrant, and the end of that sub-queue
  const uint32_t sq0 = (T0 * kQSub + sub) * a.q.cap, sq1 = ((T0 + a.q.tiles_y) * kQSub + sub) * a.q.cap,
                 sq2 = ((T0 + 1u) * kQSub + sub) * a.q.cap, sq3 = ((T0 + a.q.tiles_y + 1u) * kQSub + sub) * a.q.cap;
  uint32_t e0 = sq0 + (uint32_t)__shfl((int)g0, c.leader) + (c.pre01 & 0xFFFFu);
  uint32_t e1 = sq1 + (uint32_t)__shfl((int)g1, c.leader) + (c.pre01 >> 16);
  uint32_t e2 = sq2 + (uint32_t)__shfl((int)g2, c.leader) + (c.pre23 & 0xFFFFu);
  uint32_t e3 = sq3 + (uint32_t)__shfl((int)g3, c.leader) + (c.pre23 >> 16);
  unsigned long long mask = 0ull;
  if (SMESH_ABL(a.dbg) & 8) cover = 0ull;   // ablation: setup + coverage + slot reservation, no depth / stores
  for (unsigned long long m = cover; m; m &= m - 1ull) {
    const int bit = __ffsll((long long)m) - 1;
    const int dx = bit >> 3, dy = bit & 7;
    const int x = X0 + dx, y = Y0 + dy;
    const double px = (double)x + 0.5, py = (double)y + 0.5;
    const double w0 = eval_edge(t.e0, px, py);
    const double w1 = eval_edge(t.e1, px, py);
    const double w2 = eval_edge(t.e2, px, py);
    const double num = (w0 + w1) + w2;
    const double den = __builtin_fma(w2, t.iz2, __builtin_fma(w1, t.iz1, w0 * t.iz0));
    const float zf = (float)(num / den);
    unsigned long long key = kNullKey;
    if (zf > 0.0f && isfinite(zf)) {
      uint32_t prim = pid;
      if constexpr (TEX) prim = a.tex_first[f] + texel_of(a.tex_res[f], w1 / num, w2 / num);
      key = ((unsigned long long)__float_as_uint(zf) << 32) | prim;
      mask |= 1ull << bit;
    }
    const bool hx = dx >= bx, hy = dy >= by;
    // this fragment's entry, and one past the last entry of its sub-queue (selects, not branches: the four cases diverge in every wave)
    const uint32_t e = hy ? (hx ? e3 : e2) : (hx ? e1 : e0);
    const uint32_t lim = (hy ? (hx ? sq3 : sq2) : (hx ? sq1 : sq0)) + a.q.cap;
    e0 += (!hx && !hy) ? 1u : 0u; e1 += (hx && !hy) ? 1u : 0u; e2 += (!hx && hy) ? 1u : 0u; e3 += (hx && hy) ? 1u : 0u;
    if (SMESH_ABL(a.dbg) & 1) continue;
    // pixel inside its tile: (x mod 32) * 64 + (y mod 64)
    const uint16_t pin = (uint16_t)((((uint32_t)x & (kQW - 1)) * kQH) | ((uint32_t)y & (kQH - 1)));
    if (e < lim) {
      a.q.key[e] = key;
      a.q.pix[e] = pin;
    } else if (key != kNullKey) {
      atomicMin(&a.keys[key_index((uint32_t)x, (uint32_t)y, a.H)], key);
      atomicOr(&a.q.flag[(tx0 + (hx ? 1u : 0u)) * a.q.tiles_y + ty0 + (hy ? 1u : 0u)], 1u);
      a.big_count[1] = 1u;      // "check the masks" is known BEFORE the tile resolve starts (it decides there whether the index plane is written)
    }
  }
  return mask;
}

// One lane per triangle.  Box <= 8 x 8: coverage walk (walk_box) and the fragments (emit_cover), and the record's mask.  Box up to
// kLaneBox x kLaneBox (round 5): the SAME lane walks the box as up to (kLaneBox / 8)^2 sub-boxes of 8 x 8, one per wave-uniform round
// -- its set-up stays in registers, every round is the small-triangle procedure, and only covered samples reach the queues.  (Until
// round 5 every box over 8 x 8 went through the cooperative loop below: one triangle at a time per wave, ~100 instructions per 64
// samples plus ~40 to broadcast the triangle, null keys for the uncovered samples -- a mesh of ~20-pixel triangles spent 41 us per
// 1080p view there.)  Larger boxes, up to kMedium x kMedium: the cooperative loop.
// `f`: this lane's triangle (a.F: none; ANY triangle: lanes need not hold consecutive ones), `i0 .. i2` its vertex indices (valid
// when f < a.F); `sub`: the wave's sub-queue in every tile.
constexpr int kLaneBox = 24;
constexpr int kLaneBoxMin = 20;
constexpr int kSpread = (kLaneBox / 8) * (kLaneBox / 8);   // lanes per triangle in a spread wave: one per sub-box
constexpr int kSpreadTris = 64 / kSpread;                  // triangles per spread wave
// `part`: -1, or (spread waves, raster_frag_wave) which of the triangle's sub-boxes this lane walks; the lane with part 0 is the triangle's
// owner (its record, its queue entries, the small box, the cooperative loop), the others only walk.
// The cooperative walk of one triangle by a whole wave, in two pieces used by raster_frag_64 (the triangle's owner lane broadcasts it)
// and by k_raster_medium (every lane set the triangle up for itself).
// Queue bases of the (at most 3 x 2) tile sub-rectangles of the box of `t`: the whole sub-rectangle is reserved with one atomic per tile.
__device__ __forceinline__ void coop_reserve(const RasterArgs& a, const Tri& t, const uint32_t sub, uint32_t (&rb)[6]) {
  const int tX0 = t.x0 / kQW, tY0 = t.y0 / kQH;
  const int ntx = t.x1 / kQW - tX0 + 1, nty = t.y1 / kQH - tY0 + 1;
  auto reserve = [&](int j) -> uint32_t {
    if (j >= ntx * nty) return 0u;
    const int jx = tX0 + j / nty, jy = tY0 + j % nty;
    const int wx = min(t.x1, jx * kQW + kQW - 1) - max(t.x0, jx * kQW) + 1, hy = min(t.y1, jy * kQH + kQH - 1) - max(t.y0, jy * kQH) + 1;
    return atomicAdd(&a.q.count[((uint32_t)jx * a.q.tiles_y + (uint32_t)jy) * kQSub + sub], (uint32_t)(wx * hy));
  };
#pragma unroll
  for (int j = 0; j < 6; j++) rb[j] = reserve(j);     // in flight together
}
// The triangle of lane `src` (its set-up `t`, its reservations `rb`, triangle number f / index value pid in that lane), all 64 lanes on its box.
template <bool TEX>
__device__ __forceinline__ void coop_walk(const RasterArgs& a, const Tri& t, const uint32_t (&rb)[6], const int src, const uint64_t fsrc,
                                          const uint32_t psrc, const uint32_t sub) {
  const int lane = threadIdx.x & 63;
  const uint32_t rb0 = rb[0], rb1 = rb[1], rb2 = rb[2], rb3 = rb[3], rb4 = rb[4], rb5 = rb[5];
    auto bi = [&](int v) -> int { return __builtin_amdgcn_readlane(v, src); };
    auto bd = [&](double v) -> double { return __hiloint2double(bi(__double2hiint(v)), bi(__double2loint(v))); };
    const int X0 = bi(t.x0), X1 = bi(t.x1), Y0 = bi(t.y0), Y1 = bi(t.y1);
    const double e0A = bd(t.e0.A), e0B = bd(t.e0.B), e0C = bd(t.e0.C);
    const double e1A = bd(t.e1.A), e1B = bd(t.e1.B), e1C = bd(t.e1.C);
    const double e2A = bd(t.e2.A), e2B = bd(t.e2.B), e2C = bd(t.e2.C);
    const double iz0 = bd(t.iz0), iz1 = bd(t.iz1), iz2 = bd(t.iz2);
    const int c0 = bi(t.cls0), c1 = bi(t.cls1), c2 = bi(t.cls2);
    const uint32_t fb = (uint32_t)bi((int)(uint32_t)fsrc);      // (the owner lane's triangle: lanes need not hold consecutive ones, see raster_frag_wave)
    const uint32_t fbid = (uint32_t)bi((int)psrc);
    uint32_t tfirst = 0u, tres = 0u;
    if constexpr (TEX) { tfirst = a.tex_first[fb]; tres = a.tex_res[fb]; }
    const int tX0 = X0 / kQW, tY0 = Y0 / kQH, nty = Y1 / kQH - tY0 + 1;
    const uint32_t sb0 = (uint32_t)bi((int)rb0), sb1 = (uint32_t)bi((int)rb1), sb2 = (uint32_t)bi((int)rb2),
                   sb3 = (uint32_t)bi((int)rb3), sb4 = (uint32_t)bi((int)rb4), sb5 = (uint32_t)bi((int)rb5);
    const int bh = Y1 - Y0 + 1, area = (X1 - X0 + 1) * bh;
    // i / bh as a multiplication: exact for bh <= kMedium = 64 and i < 2^26 (i * (m - 2^32 / bh) / 2^32 < 1 / bh); the two integer
    // divisions per sample were a third of the loop's instructions
    const uint32_t bh_rcp = bh > 1 ? 0xFFFFFFFFu / (uint32_t)bh + 1u : 0u;   // (bh = 1: the multiplier would be 2^32)
    for (int i = lane; i < area + lane; i += 64) {   // every lane runs the same number of steps (readlane inside)
      const bool in_box = i < area;
      const int col = bh > 1 ? (int)__umulhi((uint32_t)i, bh_rcp) : i;
      const int x = X0 + (in_box ? col : 0), y = Y0 + (in_box ? i - col * bh : 0);
      const double px = (double)x + 0.5, py = (double)y + 0.5;
      const double w0 = __builtin_fma(e0A, px, __builtin_fma(e0B, py, e0C));
      const double w1 = __builtin_fma(e1A, px, __builtin_fma(e1B, py, e1C));
      const double w2 = __builtin_fma(e2A, px, __builtin_fma(e2B, py, e2C));
      const bool cov = edge_accepts(w0, c0) && edge_accepts(w1, c1) && edge_accepts(w2, c2);
      unsigned long long key = kNullKey;
      if (in_box && cov) {
        const double num = (w0 + w1) + w2;
        const double den = __builtin_fma(w2, iz2, __builtin_fma(w1, iz1, w0 * iz0));
        const float zf = (float)(num / den);
        if (zf > 0.0f && isfinite(zf)) {
          uint32_t prim = fbid;
          if constexpr (TEX) prim = tfirst + texel_of(tres, w1 / num, w2 / num);
          key = ((unsigned long long)__float_as_uint(zf) << 32) | prim;
        }
      }
      // slot of this sample inside its tile's sub-rectangle
      const int jx = x / kQW, jy = y / kQH;
      const int xlo = max(X0, jx * kQW), ylo = max(Y0, jy * kQH), hy = min(Y1, jy * kQH + kQH - 1) - ylo + 1;
      const int jt = (jx - tX0) * nty + (jy - tY0);
      const uint32_t tb = jt == 0 ? sb0 : jt == 1 ? sb1 : jt == 2 ? sb2 : jt == 3 ? sb3 : jt == 4 ? sb4 : sb5;
      const uint32_t slot = tb + (uint32_t)((x - xlo) * hy + (y - ylo));
      const uint32_t tile = (uint32_t)jx * a.q.tiles_y + (uint32_t)jy;
      if (!in_box || (SMESH_ABL(a.dbg) & 1)) continue;
      if (slot < a.q.cap) {
        const uint64_t e = ((uint64_t)tile * kQSub + sub) * a.q.cap + slot;
        a.q.key[e] = key;
        a.q.pix[e] = (uint16_t)((x - jx * kQW) * kQH + (y - jy * kQH));
      } else if (key != kNullKey) {
        atomicMin(&a.keys[key_index((uint32_t)x, (uint32_t)y, a.H)], key);
        atomicOr(&a.q.flag[tile], 1u);
        a.big_count[1] = 1u;
      }
    }
}

// MODE (an instance of the kernels per mode: the code of the other modes costs the small-triangle instance registers it does not
// have -- k_raster_frag_group is held to 96 for five waves per SIMD): bit 0 = RasterArgs::wg_push, bit 1 = RasterArgs::spread,
// bit 2 = texel primitives (RasterArgs::tex_res != nullptr; emit_cover).  (Round 6 also built a bit 3: an 8-byte PIXEL CELL per vertex -- first / last
// sample column and row a box ending / beginning there can reach -- read first, so that the seven in eight sub-pixel triangles of cfg4 that hold no
// sample centre are dropped on 24 bytes instead of 72: k_raster_frag_group 667 -> 728 us per eight cfg4 views, 2 342 -> 2 579 at cfg5, the vertex stage
// 97 -> 125: a second dependent round trip for every wave costs more than the bytes.  Not kept.)  (Round 6 also built instances whose lanes project their
// triangles' vertices themselves -- no vertex stage, no 24-byte record per vertex and view: k_raster_frag_group 195 -> 242 us per eight
// cfg2 views for 24 us of vertex stage saved, cfg4 668 -> 814 for 106; not kept.)
template <int MODE>
__device__ __forceinline__ void raster_frag_64(const RasterArgs& a, const uint64_t f, const int32_t i0, const int32_t i1, const int32_t i2,
                                               const uint32_t sub, const int part_in = -1, const bool listed = false) {
  constexpr bool kWgPush = (MODE & 1) != 0, kSpreadMode = (MODE & 2) != 0, kTex = (MODE & 4) != 0;
  const int lane = threadIdx.x & 63;
  const int part = kSpreadMode ? part_in : -1;
  const bool owner = part <= 0 && !listed;    // (listed: a triangle from the view's list, k_raster_medium -- its owner lane left its record and queue entries)
  TriFrag rec;
  rec.x0 = 0; rec.y0 = 0; rec.kind = 0; rec.pad = 0; rec.mask = 0ull;
  Tri t;
  t.x0 = 0; t.y0 = 0; t.x1 = -1; t.y1 = -1;
  unsigned long long cover = 0ull;
  int CX0 = 0, CY0 = 0;          // origin of the (at most 8 x 8) box `cover` describes
  bool medium = false, lanebox = false, small = false;
  bool q_big = false, q_mid = false, q_huge = false;   // this lane's triangle goes into the view's queues (below, one atomic per wave and queue)
  const uint32_t pid = (a.prim_id && f < a.F) ? a.prim_id[f] : (uint32_t)f;   // value written to the index image
  const int have = f < a.F ? load_tri_ex(a, f, t, i0, i1, i2) : 0;   // 2: crosses the near plane (t holds only its screen box)
  if (have) {
    const int bw = t.x1 - t.x0 + 1, bh = t.y1 - t.y0 + 1;
    rec.x0 = (uint16_t)t.x0; rec.y0 = (uint16_t)t.y0;
    CX0 = t.x0; CY0 = t.y0;
    if (bw > 8 || bh > 8 || have == 2) {
      q_big = owner;                                             // every such triangle: the fusion walks this queue
      q_mid = owner && bw * bh <= kMidBox;                       // (push_mid's list)
      rec.kind = 2;
      rec.mask = (unsigned long long)(uint32_t)t.x1 | ((unsigned long long)(uint32_t)t.y1 << 16);
      if (bw <= kLaneBox && bh <= kLaneBox && have == 1) {       // rasterised by lanes, sub-box by sub-box
        if (part < 0) lanebox = true;                            // ... this lane, one sub-box per round (below)
        else {                                                   // ... the triangle's kSpread lanes, one sub-box each, now
          constexpr int kSub = kLaneBox / 8;
          CX0 = t.x0 + 8 * (part / kSub); CY0 = t.y0 + 8 * (part % kSub);
          const int sw = min(t.x1 - CX0 + 1, 8), sh = min(t.y1 - CY0 + 1, 8);
          if (sw > 0 && sh > 0 && !(SMESH_ABL(a.dbg) & 2)) cover = walk_box(t, CX0, CY0, sw, sh);
        }
      } else if (owner) {                                        // (the other lanes of a spread triangle have nothing to do with it)
        if (bw <= kMedium && bh <= kMedium && have == 1) medium = true;   // rasterised below by the whole wave
        else q_huge = true;                                      // rasterised by the tile workgroups it overlaps
      }
    } else if (owner && !(SMESH_ABL(a.dbg) & 2)) {
      small = true;
      cover = walk_box(t, t.x0, t.y0, bw, bh);
    }
  }
  // The queues of the view: ONE atomic per wave and queue -- or per workgroup (a.wg_push), where most triangles are queued: increments
  // of one counter pass the L2 one by one at ~5 ns each, and one view of a 40 000-triangle mesh in spread waves spent 73 of its
  // 132 us on 11 000 of them.
  {
    __shared__ uint32_t s_push[kWgPush ? 3 : 1][6];     // per queue: the counts of the four waves, then the workgroup's base
    const int wv = (int)(threadIdx.x >> 6);
    (void)wv; (void)s_push;
    uint32_t* const counter[3] = {a.big_count, a.big_count + 3, a.big_count + 2};
    const bool want[3] = {q_big, q_mid, q_huge};
    uint32_t slot[3];
    if constexpr (kWgPush) {
      unsigned long long m[3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        m[q] = __ballot(want[q]);
        if (lane == 0) s_push[q][wv] = (uint32_t)__popcll(m[q]);
      }
      __syncthreads();
      if (threadIdx.x == 0u) {            // (three independent atomics, in flight together; their addresses are scalars)
        uint32_t total[3], base[3] = {0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 3; q++) total[q] = s_push[q][0] + s_push[q][1] + s_push[q][2] + s_push[q][3];
        if (total[0]) base[0] = atomicAdd(a.big_count, total[0]);
        if (total[1]) base[1] = atomicAdd(a.big_count + 3, total[1]);
        if (total[2]) base[2] = atomicAdd(a.big_count + 2, total[2]);
#pragma unroll
        for (int q = 0; q < 3; q++) s_push[q][4] = base[q];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 3; q++) {
        uint32_t before = s_push[q][4];
        for (int w = 0; w < wv; w++) before += s_push[q][w];
        slot[q] = want[q] ? before + (uint32_t)__popcll(m[q] & ((1ull << lane) - 1ull)) : 0xFFFFFFFFu;
      }
      __syncthreads();                    // (the next group of a looping wave writes its counts)
    } else {
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const unsigned long long m = __ballot(want[q]);
        slot[q] = 0xFFFFFFFFu;
        if (m == 0ull) continue;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0u;
        if (lane == leader) base = atomicAdd(counter[q], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader);
        if (want[q]) slot[q] = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      }
    }
    if (q_big && slot[0] < a.big_capacity) a.big_queue[slot[0]] = (uint32_t)f;
    if (q_mid && slot[1] < a.big_capacity) a.big_queue[(uint64_t)a.big_capacity + slot[1]] = (uint32_t)f;
    if (q_huge && slot[2] < a.big_capacity) a.huge_queue[slot[2]] = (uint32_t)f;
  }
  if (SMESH_ABL(a.dbg) & 4) { if (a.frags && f < a.F && owner) { rec.mask = cover; a.frags[f] = rec; } return; }   // ablation: setup + coverage only
  {
    const unsigned long long mask = emit_cover<kTex>(a, t, f, pid, CX0, CY0, cover, sub);
    if (mask && small) rec.kind = 1;
    if (rec.kind == 1) rec.mask = mask;
  }
  if (owner) {
    // texel renderers (a.kinds): seven triangles in eight emit nothing in a cfg4 view -- those leave their kind byte only, and every
    // reader of a texel renderer's records asks the byte first (cfg4: 5 050 -> 5 210 views/s)
    if (a.frags && f < a.F && (rec.kind != 0 || !a.kinds)) a.frags[f] = rec;
    if (a.kinds && f < a.F) a.kinds[f] = (uint8_t)rec.kind;
  }
  // ---- boxes over 8 x 8 up to kLaneBox x kLaneBox: this lane's sub-boxes, one per round (wave-uniform trip count) -- where the wave
  // holds enough of them: a round costs what a whole small-triangle pass costs (~2 200 instructions) however many lanes take part,
  // the cooperative loop ~350 - 450 per triangle, so a wave with fewer than kLaneBoxMin such triangles hands them to that loop
  if constexpr (kWgPush && !kSpreadMode) {
    if (a.balance >= 2u) {
      // balanced, one view per launch: these triangles join the view's list too (from its far end; count in big_count[5]) and k_raster_medium walks them
      // seven to a wave, a lane per sub-box -- the waves of the middle distance of a view from inside held 64 of them each
      const unsigned long long m = __ballot(lanebox);
      if (m != 0ull && !(SMESH_ABL(a.dbg) & 1)) {
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0u;
        if (lane == leader) base = atomicAdd(a.big_count + 5, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader);
        if (lanebox) a.med_queue[a.big_capacity - 1u - (base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)))] = (uint32_t)f;
      }
      lanebox = false;
    }
  }
  if ((int)__popcll(__ballot(lanebox)) < kLaneBoxMin) { medium = medium || lanebox; lanebox = false; }
  if (__ballot(lanebox) != 0ull && !(SMESH_ABL(a.dbg) & 1)) {
    constexpr int kSub = kLaneBox / 8;
    for (int k = 0; k < kSub * kSub; k++) {
      const int X0 = t.x0 + 8 * (k / kSub), Y0 = t.y0 + 8 * (k % kSub);
      const int bw = lanebox ? min(t.x1 - X0 + 1, 8) : 0, bh = lanebox ? min(t.y1 - Y0 + 1, 8) : 0;
      const bool on = bw > 0 && bh > 0;
      if (__ballot(on) == 0ull) continue;
      const unsigned long long cv = on ? walk_box(t, X0, Y0, bw, bh) : 0ull;
      (void)emit_cover<kTex>(a, t, f, pid, on ? X0 : 0, on ? Y0 : 0, cv, sub);
    }
  }

  // ---- medium triangles (box up to kMedium x kMedium): one at a time, all 64 lanes on its bounding box.  The
  // triangle is broadcast from its owner lane (readlane -> scalar registers).  Queue slots: the box overlaps at most
  // 3 x 2 tiles; in each the whole sub-rectangle is reserved with one atomic (a sample's slot is its position in
  // the sub-rectangle) and samples the triangle does not cover store the null key -- no per-fragment bookkeeping.
  // All reservations of the wave are issued first, by the owner lanes: one memory round trip, not one per triangle.
  // (balanced instances: those triangles go to the view's list instead and ANY wave walks them, below)
  uint32_t rb[6] = {0u, 0u, 0u, 0u, 0u, 0u};   // queue bases of this lane's (<= 3 x 2) sub-rectangles
  if constexpr (kWgPush) {
    if (a.balance) {
      // balanced: the triangle joins the view's list instead (one reservation per wave), k_raster_medium walks the list with the
      // whole chip -- a view from inside the scene has all its medium triangles in a few waves (the near part of the mesh)
      const unsigned long long m = __ballot(medium);
      if (m != 0ull && !(SMESH_ABL(a.dbg) & 1)) {
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0u;
        if (lane == leader) base = atomicAdd(a.big_count + 4, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader);
        if (medium) a.med_queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)f;    // (< big_capacity: a triangle is listed once)
      }
      return;
    }
  }
  if (medium && !(SMESH_ABL(a.dbg) & 1)) coop_reserve(a, t, sub, rb);     // all reservations of the wave first: one memory round trip
  unsigned long long todo = __ballot(medium);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    coop_walk<kTex>(a, t, rb, src, f, pid, sub);
  }
}

// The view's list of medium triangles (RasterArgs::med_queue, filled by the balanced instances of k_raster_frag), a triangle per wave
// and turn: every lane sets the triangle up for itself -- the same loads and arithmetic as its owner lane's, the same bits -- lane 0
// reserves its tiles' sub-rectangles, the wave walks its box.
template <bool TEX>
__device__ __forceinline__ void raster_medium_wave(const RasterArgs& a, const uint32_t wave, const uint32_t nwaves) {
  const int lane = threadIdx.x & 63;
  const uint32_t n = min(a.big_count[4], a.big_capacity);
  for (uint32_t i = wave; i < n; i += nwaves) {
    const uint32_t g = a.med_queue[i];
    if (g >= a.F) continue;
    const int32_t g0 = a.faces[3 * (uint64_t)g + 0], g1 = a.faces[3 * (uint64_t)g + 1], g2 = a.faces[3 * (uint64_t)g + 2];
    Tri t;
    if (load_tri_ex(a, g, t, g0, g1, g2) != 1) continue;            // (never: its owner lane found it drawable)
    uint32_t rb[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    if (lane == 0) coop_reserve(a, t, i & (kQSub - 1), rb);
    coop_walk<TEX>(a, t, rb, 0, g, a.prim_id ? a.prim_id[g] : g, i & (kQSub - 1));
  }
  // ... and the listed triangles with boxes of 9 .. kLaneBox pixels (from the far end of the list), seven per wave and turn: a spread wave
  constexpr int kLanes = kSpread, kTris = kSpreadTris;
  const uint32_t nl = min(a.big_count[5], a.big_capacity);
  const uint32_t turns = (nl + (uint32_t)kTris - 1u) / (uint32_t)kTris;
  for (uint32_t i = wave; i < turns; i += nwaves) {
    const int tri = lane / kLanes;
    const uint32_t e = i * (uint32_t)kTris + (uint32_t)tri;
    uint64_t fs = a.F;
    if (tri < kTris && e < nl) fs = a.med_queue[a.big_capacity - 1u - e];
    int32_t s0 = 0, s1 = 0, s2 = 0;
    if (fs < a.F) { s0 = a.faces[3 * fs + 0]; s1 = a.faces[3 * fs + 1]; s2 = a.faces[3 * fs + 2]; }
    raster_frag_64<TEX ? 6 : 2>(a, fs, s0, s1, s2, i & (kQSub - 1), lane % kLanes, true);
  }
}
template <bool TEX>
__global__ __launch_bounds__(256) void k_raster_medium(RasterArgs a) {
  raster_medium_wave<TEX>(a, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, (gridDim.x * blockDim.x) >> 6);
}

// One wave = a.groups x a.tpw consecutive triangles (tpw: 64 for large meshes; fewer for small ones, so that the cooperative
// medium-triangle loop has enough waves to spread over the chip).  With several groups per wave (LOOP: the grouped launches of
// meshes of millions of triangles) the next group's vertex indices are requested before the current group is shaded (three
// registers), so that it starts one memory round trip ahead: cfg4 666 -> 627 us per eight-view launch with four groups; nothing at
// cfg2 (169 / 171 / 175 / 182 us with 1 / 2 / 4 / 8 groups, tools/raster_groups.sh), where one group per wave stays.
// (Lane compaction -- cull a group by bounding box first, park the triangles that can emit in an LDS ring, shade full waves of
// them -- removes a third (cfg2) to three quarters (cfg4) of the wave instructions and was measured SLOWER in rounds 2 and 5:
// cfg2 33.1 -> 38.0 us per view for the raster stage, cfg4 4 777 -> 4 615 views/s.  The kernel waits for memory round trips, not for
// issue slots; NOTES/round5.md.)
template <bool LOOP, int MODE>
__device__ __forceinline__ void raster_frag_wave(const RasterArgs& a, const uint64_t wave_id) {
  const int lane = threadIdx.x & 63;
  if constexpr ((MODE & 2) != 0) {
    // Spread waves (round 5): where most boxes are 9 .. kLaneBox pixels wide and the launch has few triangles (one view of a decimated
    // scan: 40 000 triangles are 625 waves of 64), a triangle takes kSpread consecutive lanes, one per 8 x 8 sub-box -- seven triangles a
    // wave, every lane setting up its triangle for itself, ONE coverage walk and one pass of reservations and stores per wave instead
    // of up to nine rounds.  Same fragments.
    const int tri = lane / kSpread;
    uint64_t fs = wave_id * (uint64_t)kSpreadTris + (uint64_t)tri;
    if (tri >= kSpreadTris || fs >= a.F) fs = a.F;
    int32_t s0 = 0, s1 = 0, s2 = 0;
    if (fs < a.F) { s0 = a.faces[3 * fs + 0]; s1 = a.faces[3 * fs + 1]; s2 = a.faces[3 * fs + 2]; }
    raster_frag_64<MODE>(a, fs, s0, s1, s2, (uint32_t)wave_id & (kQSub - 1), lane % kSpread);
    return;
  } else {
  const uint32_t G = LOOP ? a.groups : 1u;
  uint64_t f = lane < (int)a.tpw ? wave_id * G * a.tpw + lane : a.F;
  int32_t n0 = 0, n1 = 0, n2 = 0;
  if (f < a.F) { n0 = a.faces[3 * f + 0]; n1 = a.faces[3 * f + 1]; n2 = a.faces[3 * f + 2]; }
  for (uint32_t g = 0; g < G; g++) {
    const int32_t i0 = n0, i1 = n1, i2 = n2;
    const uint64_t fn = f + 64u;          // (G > 1 only with 64 triangles per group)
    if (g + 1 < G && fn < a.F) { n0 = a.faces[3 * fn + 0]; n1 = a.faces[3 * fn + 1]; n2 = a.faces[3 * fn + 2]; }
    raster_frag_64<MODE>(a, f, i0, i1, i2, (uint32_t)(wave_id * G + g) & (kQSub - 1));
    f = fn < a.F ? fn : a.F;
  }
  }
}

// Block -> triangles, XCD-aware (round 6).  The dispatcher places block b on XCD b % 8 and every XCD has its own L2, so with blocks
// taking consecutive triangles in dispatch order, the four or so blocks that share a line of projected vertices (neighbouring
// triangles) sat on four different XCDs and each L2 fetched the line for itself: 25 MB of vertex reads per cfg2 view for a 12 MB
// array (profiles/r06_pmc_raster_cfg2.txt).  Now the triangle blocks are dealt to the XCDs in RUNS of `run` consecutive blocks: the
// s-th block an XCD x is given (s = b / 8) is triangle block (s / run) * 8 * run + x * run + s % run.  (One run per XCD -- an eighth
// of the mesh each -- was measured slower, 198 -> 239 us per eight cfg2 views: the eighths of a mesh are not equally expensive in a
// view, and the launch then waits for the slowest XCD.)  Blocks beyond the last triangle block leave at once.  A speed choice only:
// any placement rasterises every triangle once.
__device__ __forceinline__ uint32_t xcd_block(const uint32_t xcd, const uint32_t s, const uint32_t run) {
  const uint32_t q = s / run;
  return (q * 8u + xcd) * run + (s - q * run);
}
// blocks per XCD (its sequence length) for `nblocks` triangle blocks in runs of `run`
inline uint32_t xcd_slots(uint32_t nblocks, uint32_t run) { return (uint32_t)div_up(nblocks, 8u * run) * run; }

// (Five waves per SIMD -- 96 registers -- for the small-triangle instances; built for six (80 registers, four of them spilled) the
// grouped launch took 195.7 against 200.3 us per eight cfg2 views alone and LOST beside the fusion launch, 14 910 against 15 492 views/s;
// for seven (72 registers, 48 spilled) 323 us.  Round 6.)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MODE & 3) ? 4 : 5, 5))) void k_raster_frag(RasterArgs a, uint32_t nblocks, uint32_t chunk) {
  const uint32_t tb = chunk ? xcd_block(blockIdx.x & 7u, blockIdx.x >> 3, chunk) : blockIdx.x;
  if (tb >= nblocks) return;     // (block-uniform)
  raster_frag_wave<false, MODE>(a, ((uint64_t)tb * blockDim.x + threadIdx.x) >> 6);
}
int raster_mode(const RasterArgs& a) { return (a.spread ? 3 : a.wg_push ? 1 : 0) | (a.tex_res ? 4 : 0); }   // (spread views are views of medium triangles: wg_push too)   // (spread views are views of medium triangles: wg_push too)

// Several views in one launch: blocks [v * blocks_per_view, (v + 1) * blocks_per_view) rasterise view v.
struct RasterGroup {
  RasterArgs view[kMaxGroup];
  uint32_t* idx[kMaxGroup];       // k_tile_resolve_group: index plane of view v ...
  uint32_t tile_end[kMaxGroup];   // ... whose tiles are blocks [tile_end[v-1], tile_end[v])
  uint32_t n, blocks_per_view;
  uint32_t chunk;                 // k_raster_frag_group: nonzero = XCD-aware placement (below): consecutive triangle blocks per run of an XCD
};
// (five waves per SIMD -- 96 registers -- for the small-triangle instance; the instances for views of medium triangles may take 128: at 96
// they spill, and their waves are long)
// Placement (g.chunk != 0): the triangle blocks go to the XCDs in runs (xcd_block), for EVERY view alike, and the views of a launch
// take turns on an XCD -- blocks 8 k + x, k = 0, 1, 2 ... of the grid are (the XCD's s-th triangle block, view v) for k = s * n + v:
// the n views' blocks over the same 256 triangles are dispatched back to back on the same XCD, so the triangles' vertex indices
// (12 MB per cfg2 view) are fetched into that L2 once per launch, not once per view.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MODE & 3) ? 4 : 5, 5))) void k_raster_frag_group(RasterGroup g) {
  uint32_t v, tb;
  if (g.chunk) {
    const uint32_t k = blockIdx.x >> 3, sq = k / g.n;
    v = k - sq * g.n;
    tb = xcd_block(blockIdx.x & 7u, sq, g.chunk);
    if (tb >= g.blocks_per_view) return;               // block-uniform
  } else {
    v = blockIdx.x / g.blocks_per_view;                // block-uniform
    tb = blockIdx.x - v * g.blocks_per_view;
  }
  raster_frag_wave<true, MODE>(g.view[v], ((uint64_t)tb * blockDim.x + threadIdx.x) >> 6);
}
template <bool TEX>
__global__ __launch_bounds__(256) void k_raster_medium_group(RasterGroup g, uint32_t blocks_per_view) {
  const uint32_t v = blockIdx.x / blocks_per_view;   // block-uniform
  raster_medium_wave<TEX>(g.view[v], ((blockIdx.x - v * blocks_per_view) * blockDim.x + threadIdx.x) >> 6, (blocks_per_view * blockDim.x) >> 6);
}
// The largest triangles (box over kMedium x kMedium) and the triangles that cross the near plane (clip_piece), between k_raster_frag
// and k_tile_resolve: one workgroup per screen tile scans their queue, keeps those whose box overlaps the tile and shades the
// overlap, 256 samples at a time, piece by piece.  The fragments go through the global key image (64-bit atomicMin; the lanes of
// an instruction run down a column, consecutive keys) and the tile is flagged (bit 1) so that its resolve merges them.  With an
// empty queue -- every BASELINE config -- a workgroup reads one counter and leaves.  (Round 2 did this inside k_tile_resolve,
// against the LDS keys; with the clipping code in it the resolve went from 76 to 110 VGPRs and from 57 to 159 us per eight cfg2
// views, so it moved out.)
__device__ __forceinline__ void raster_huge_block(const RasterArgs& a, const uint32_t tile) {
  __shared__ uint32_t s_hits[256];
  __shared__ uint32_t s_nhits;
  const uint32_t nbig = min(a.big_count[2], a.big_capacity);
  if (nbig == 0u) return;
  const FragQueues& q = a.q;
  const int t = threadIdx.x;
  const uint32_t tx = tile / q.tiles_y, ty = tile - tx * q.tiles_y;
  const uint32_t x0 = tx * kQW, y0 = ty * kQH;
  const int tx1 = (int)min(x0 + kQW, a.W) - 1, ty1 = (int)min(y0 + kQH, a.H) - 1;   // last pixel of the tile inside the image
  bool any = false;
  if (t == 0) s_nhits = 0u;
  __syncthreads();
  for (uint32_t qb = 0; qb < nbig; qb += 256u) {
    const uint32_t qi = qb + (uint32_t)t;
    if (qi < nbig) {
      const uint32_t f = a.huge_queue[qi];
      const TriFrag rec = a.frags[f];
      const int bx1 = (int)(rec.mask & 0xFFFFu), by1 = (int)((rec.mask >> 16) & 0xFFFFu);
      if (rec.kind == 2 && (int)rec.x0 <= tx1 && bx1 >= (int)x0 && (int)rec.y0 <= ty1 && by1 >= (int)y0)
        s_hits[atomicAdd(&s_nhits, 1u)] = f;
    }
    __syncthreads();
    const uint32_t nh = s_nhits;
    for (uint32_t h = 0; h < nh; h++) {
      const uint64_t f = s_hits[h];
      for (int p = 0; p < 2; p++) {   // the triangle, or the one or two pieces of a triangle that crosses the near plane
        Piece pc;
        if (!load_piece(a, f, p, pc)) continue;
        const Tri& tr = pc.t;
        const int xa = max(tr.x0, (int)x0), xb = min(tr.x1, tx1), ya = max(tr.y0, (int)y0), yb = min(tr.y1, ty1);
        const int hh = yb - ya + 1, area = (xb - xa + 1) * hh;
        if (hh <= 0) continue;
        for (int i = t; i < area; i += 256) {
          const int x = xa + i / hh, y = ya + i % hh;
          const unsigned long long key = shade_piece_key(a, f, pc, x, y);
          if (key != kNullKey) { atomicMin(&a.keys[key_index((uint32_t)x, (uint32_t)y, a.H)], key); any = true; }
        }
      }
    }
    __syncthreads();
    if (t == 0) s_nhits = 0u;
    __syncthreads();
  }
  if (any) atomicOr(&q.flag[tile], 2u);
}

// (a grid of a few hundred workgroups walking the tiles: with the queue empty -- the usual case -- the launch is over in the time
// it takes that many waves to read one counter)
__global__ __launch_bounds__(256) void k_raster_huge(RasterArgs a, uint32_t ntiles) {
  if (min(a.big_count[2], a.big_capacity) == 0u) return;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) { raster_huge_block(a, tile); __syncthreads(); }
}

// One workgroup per tile: depth test in LDS over the tile's fragment queue, then the big triangles (bounding box
// > 8 x 8: the workgroup scans their queue, keeps those whose box overlaps the tile and shades the overlap, 256
// samples at a time), then the output planes are written once.  "Big" here means larger than kMedium x kMedium.
// `planes_wanted()`: false = nobody will read this view's planes (RasterArgs::idx_optional; decided by the kernel per view or for the whole launch, asked
// only where the planes would be written: its loads are not on the block's way in).
template <typename PlanesWanted>
__device__ __forceinline__ void tile_resolve_block(const RasterArgs& a, uint32_t* __restrict__ idx_out, float* __restrict__ depth_out,
                                                   const uint32_t tile, PlanesWanted planes_wanted) {
  __shared__ unsigned long long skeys[kQPixels];
  __shared__ uint32_t scount[kQSub];
  const FragQueues& q = a.q;
  const int t = threadIdx.x;
  const uint32_t W = a.W, H = a.H;
  const uint32_t tx = tile / q.tiles_y, ty = tile - tx * q.tiles_y;
  const uint32_t x0 = tx * kQW, y0 = ty * kQH;
  // 256 / kQSub threads per sub-queue (the sub-queues of a tile fill evenly: waves pick theirs by wave id).  The first
  // kSpec entries of every thread are requested together with the fill count (they exist in memory whether or not they
  // are valid): one memory round trip instead of two for the typical tile.
  constexpr uint32_t kGroup = 256u / kQSub, kSpec = 6;
  const uint32_t sq = (uint32_t)t / kGroup, sq_lane = (uint32_t)t % kGroup;
  const uint64_t qbase = ((uint64_t)tile * kQSub + sq) * q.cap;
  unsigned long long spec_key[kSpec];
  uint16_t spec_pix[kSpec];
#pragma unroll
  for (uint32_t k = 0; k < kSpec; k++) {
    const uint32_t i = min(sq_lane + k * kGroup, q.cap - 1u);
    spec_key[k] = q.key[qbase + i];
    spec_pix[k] = q.pix[qbase + i];
  }
  const uint32_t n = min(q.count[tile * kQSub + sq], q.cap);
  if (sq_lane == 0u) scount[sq] = n;
  // bit 0: fragments of small triangles that did not fit the queue went through the global key image; bit 1: k_raster_huge left the
  // fragments of triangles larger than kMedium x kMedium (and of triangles clipped at the near plane) there
  const uint32_t tile_flag = q.flag[tile];
  const bool merge = tile_flag != 0u;
  for (int p = t; p < kQPixels; p += 256) {
    const uint32_t gx = x0 + (uint32_t)(p >> 6), gy = y0 + (uint32_t)(p & 63);
    unsigned long long k = kBackgroundKey;
    if (merge && gx < W && gy < H) {   // fragments that did not fit the queue went through the global key image
      const uint64_t g = key_index(gx, gy, H);
      k = a.keys[g];
      a.keys[g] = kBackgroundKey;   // re-armed for the next render
    }
    skeys[p] = k;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < kSpec; k++)
    if (sq_lane + k * kGroup < n) atomicMin(&skeys[spec_pix[k]], spec_key[k]);
  // (skipping the null keys of the cooperative loop's sub-rectangles here was measured slower: 10 000 triangles 0.083 -> 0.090 ms per view)
  // What the speculative loads did not cover -- entries kHead ... of every sub-queue -- is ONE list for all 256 threads (sub-queue after
  // sub-queue), four entries of a thread in flight: the sub-queues of a tile fill evenly only where many waves reach it.  A medium
  // triangle's whole sub-rectangle sits in its wave's sub-queue, and a tile of a coarse mesh hears from three or four waves: a quarter
  // of the threads then walked most of the tile's entries (round 6: profiles/r06_coarse_and_inside_traces.txt).
  constexpr uint32_t kHead = kSpec * kGroup;
  uint32_t rest[kQSub], total = 0u;
#pragma unroll
  for (int s = 0; s < kQSub; s++) { rest[s] = scount[s] > kHead ? scount[s] - kHead : 0u; total += rest[s]; }
  auto entry_of = [&](uint32_t j) -> uint64_t {      // j < total: the j-th of those entries
    uint32_t sub = 0u;
#pragma unroll
    for (int s = 0; s < kQSub - 1; s++) if (sub == (uint32_t)s && j >= rest[s]) { j -= rest[s]; sub = (uint32_t)s + 1u; }
    return ((uint64_t)tile * kQSub + sub) * q.cap + kHead + j;
  };
  for (uint32_t j0 = (uint32_t)t; j0 < total; j0 += 4u * 256u) {
    unsigned long long k4[4];
    uint16_t p4[4];
#pragma unroll
    for (uint32_t u = 0; u < 4u; u++) {
      const uint32_t j = j0 + u * 256u;
      const uint64_t e = j < total ? entry_of(j) : qbase;
      k4[u] = q.key[e];
      p4[u] = q.pix[e];
    }
#pragma unroll
    for (uint32_t u = 0; u < 4u; u++)
      if (j0 + u * 256u < total) atomicMin(&skeys[p4[u]], k4[u]);
  }
  __syncthreads();
  // The depth test is decided: tell the triangle-order fusion which fragments of the small triangles LOST it, by clearing their
  // bit in the triangle's record (about one fragment in twenty at cfg2; the winners need no memory traffic at all).  The record
  // masks are then exactly the visible pixels of every triangle, and k_fuse_tri no longer reads the index plane to find them --
  // a quarter of its scattered loads and one memory round trip of every wave.  Not for texel primitives (the key holds a texel
  // id) nor re-ordered meshes (it holds the caller's face id, the records are indexed by position); a tile that merged
  // overflowed fragments from the key image cannot name their losers and raises the "check them" flag for the whole view.
  if (a.frags && !a.prim_id && !a.tex_res) {
    if (tile_flag & 1u) {
      if (t == 0) a.big_count[1] = 1u;
    } else {
      auto lost = [&](const unsigned long long key, const uint32_t pin) {
        if (key == kNullKey || skeys[pin] == key) return;
        const uint32_t f = (uint32_t)(key & 0xFFFFFFFFull);
        const TriFrag rec = a.frags[f];
        if (rec.kind != 1) return;   // (fragments of triangles with a box over 8 x 8: their record holds the box, not a mask)
        const int bit = (int)(x0 + (pin >> 6) - rec.x0) * 8 + (int)(y0 + (pin & 63u) - rec.y0);
        atomicAnd(&a.frags[f].mask, ~(1ull << bit));
      };
#pragma unroll
      for (uint32_t k = 0; k < kSpec; k++)
        if (sq_lane + k * kGroup < n) lost(spec_key[k], spec_pix[k]);
      for (uint32_t j0 = (uint32_t)t; j0 < total; j0 += 4u * 256u) {
        unsigned long long k4[4];
        uint16_t p4[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
          const uint32_t j = j0 + u * 256u;
          const uint64_t e = j < total ? entry_of(j) : qbase;
          k4[u] = q.key[e];
          p4[u] = q.pix[e];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++)
          if (j0 + u * 256u < total) lost(k4[u], p4[u]);
      }
    }
  }
  if (planes_wanted())
  for (int p = t; p < kQPixels; p += 256) {
    const uint32_t gx = x0 + (uint32_t)(p >> 6), gy = y0 + (uint32_t)(p & 63);
    if (gx < W && gy < H) {
      const unsigned long long k = skeys[p];
      const uint64_t g = (uint64_t)gx * H + gy;
      idx_out[g] = (uint32_t)(k & 0xFFFFFFFFull);
      if (depth_out) depth_out[g] = __uint_as_float((uint32_t)(k >> 32));
    }
  }
  if (t < kQSub) q.count[tile * kQSub + t] = 0u;
  if (t == 0) q.flag[tile] = 0u;
}

// Is the index plane of a view read by anybody?  Always, unless the launch's caller is fuse_view(s) with the triangle-order kernels
// (idx_optional) -- and then only if a view FUSED TOGETHER with this one has queued triangles (big_count[0]: their waves scan the plane,
// and for a triangle that is big in one view of a launch they scan the planes of the views in which it is small as well) or masks that
// need checking (big_count[1]).  Both counters are final: every push and every overflow happened in the launches before the resolve.
__device__ __forceinline__ bool view_needs_planes(const RasterArgs& a) { return !a.idx_optional || a.big_count[0] != 0u || a.big_count[1] != 0u; }
// idx_optional == 2: every fusion launch of these views is k_fuse_tri, whose big-triangle waves read a triangle's SMALL views from their
// records' masks (fuse_box by_mask) and whose main waves check only the views that ask for it: the decision is each view's own
// (k_tile_resolve_group 41.6 against 51 us per eight cfg2 views, where a view or two of a group usually has a handful of queued triangles).
// == 1: k_fuse_tri_any / _wide, whose k_fuse_big_any scans the planes of such views: one decision for the raster launch.

__global__ __launch_bounds__(256) void k_tile_resolve(RasterArgs a, uint32_t* __restrict__ idx_out, float* __restrict__ depth_out) {
  tile_resolve_block(a, idx_out, depth_out, blockIdx.x, [&] { return view_needs_planes(a); });     // (idx_optional here: a view that is fused alone, smesh_fuse_view)
}

__global__ __launch_bounds__(256) void k_raster_huge_group(RasterGroup g) {
  bool any = false;
  for (uint32_t v = 0; v < g.n; v++) any = any || min(g.view[v].big_count[2], g.view[v].big_capacity) != 0u;
  if (!any) return;
  for (uint32_t tile = blockIdx.x; tile < g.tile_end[g.n - 1]; tile += gridDim.x) {
    uint32_t v = 0;
    while (v + 1 < g.n && tile >= g.tile_end[v]) v++;   // block-uniform, n <= kMaxGroup
    raster_huge_block(g.view[v], tile - (v ? g.tile_end[v - 1] : 0u));
    __syncthreads();
  }
}

// Several views in one launch (index planes only).
__global__ __launch_bounds__(256) void k_tile_resolve_group(RasterGroup g) {
  uint32_t v = 0;
  while (v + 1 < g.n && blockIdx.x >= g.tile_end[v]) v++;   // block-uniform, n <= kMaxGroup
  // the views of a raster launch are fused together (in launches of 8 / 4 / 2 / 1 of them): one decision for all of them (level 1) -- a view
  // without queued triangles of its own is still scanned for the triangles that are big in ANOTHER view of the launch (round 6: 2 of 7 500
  // random soups caught a per-view decision there, profiles/r06_differential_sweeps.txt) -- unless the fusion kernel does not do that (level 2)
  tile_resolve_block(g.view[v], g.idx[v], nullptr, blockIdx.x - (v ? g.tile_end[v - 1] : 0u), [&] {
    if (g.view[v].idx_optional == 2u) return view_needs_planes(g.view[v]);
    bool planes = false;
    for (uint32_t u = 0; u < g.n; u++) planes = planes || view_needs_planes(g.view[u]);
    return planes;
  });
}

// Split the key image into the two output planes and re-arm the keys for the next render.
__global__ void k_resolve(unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx, float* __restrict__ depth,
                          uint32_t W, uint32_t H) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)W * H) return;
  const uint32_t x = (uint32_t)(i / H), y = (uint32_t)(i - (uint64_t)x * H);
  const uint64_t g = key_index(x, y, H);
  const unsigned long long k = keys[g];
  keys[g] = kBackgroundKey;
  idx[i] = (uint32_t)(k & 0xFFFFFFFFull);
  if (depth) depth[i] = __uint_as_float((uint32_t)(k >> 32));
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

constexpr int kSlots = 2 * kMaxGroup;   // two banks of view slots: smesh_fuse_views rasterises group g+1 into one while group g is fused from the other
constexpr int kHeld = 32;          // smesh_fuse_views_begin: views whose records and index planes stay resident until the job's last part
constexpr int kSides = kSlots + kHeld;   // record sets / index planes: one per view slot, then the held ones (they need no rasteriser scratch)
constexpr int kRecordSides = 6;   // render_device() rotates over this many sets of per-triangle records (<= kMaxGroup)

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
  // Host-side summary of the mesh (create_common): the box of its vertices and its longest edge -- what no_huge_possible() needs to
  // prove, per camera, that the queue of k_raster_huge stays empty.  valid = every vertex finite, at least one usable face.
  struct Bounds { bool valid = false; double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, max_edge = 0, mean_edge = 0; } bounds;
  uint32_t* prim_id = nullptr;     // [F] primitive id per triangle position, when the triangles were re-ordered (else null)
  // What a render in flight needs besides the per-triangle records, per view slot: smesh_fuse_views rasterises up to
  // kMaxGroup views in the same launches (slots beyond 0 are allocated on first use); every other entry point uses slot 0.
  struct ViewScratch {
    ScreenVertex* sv = nullptr;      // projected vertices [V]
    uint32_t* huge_queue = nullptr;  // [big_capacity] triangles larger than kMedium x kMedium
    uint32_t* med_queue = nullptr;   // [big_capacity] RasterArgs::med_queue
    unsigned long long* keys = nullptr;   // global key image (direct path; overflow of the fragment queues)
    uint64_t keys_pixels = 0;
    FragQueues fq;                   // fragment-queue path: per-tile queues, sized for the largest image seen
    uint64_t fq_tiles = 0;
  } vs[kSlots];
  bool texels = false;
  uint32_t* tex_res = nullptr;     // [F]
  uint32_t* tex_first = nullptr;   // [F]
  std::vector<int32_t> h_faces;    // texel renderers: re-ordered faces
  std::vector<uint32_t> h_res, h_first;
  // What a render leaves behind for the triangle-order fusion: one set per view slot (two banks of eight: the group pipeline
  // rasterises into one bank while the other is being fused), then the held ones; render_device() rotates over the first six.
  struct Side {
    uint32_t* big_queue = nullptr;   // [big_capacity] triangles with a bounding box > 8 x 8
    uint32_t* big_count = nullptr;   // [0] length of the queue; emptied by the next render's vertex kernel
    TriFrag* frags = nullptr;        // [F] per-triangle fragment records
    uint8_t* kinds = nullptr;        // [F] texel renderers: TriFrag::kind, a byte per triangle
  } side[kSides];
  uint32_t big_capacity = 0;
  std::vector<ImagePair> images;   // pooled output planes
  Scratch own_idx;                 // for the host-output entry point
  // smesh_fuse_view pipeline: two index/depth slots, rasterised on ctx->raster_stream
  Scratch fused[kSides];   // index planes of fuse_view (slots 0, 1) / fuse_views (one per view of a group) / held views
  // smesh_fuse_views_begin / _continue: the job whose views sit in side[kSlots ...] / fused[kSlots ...]
  struct HeldJob {
    int n = 0, nparts = 0, next_part = 0;
    bool ranged = false;             // false: everything went with part 0 (the later parts are empty)
    smesh_aggregator* agg = nullptr;
    const float* probs[kHeld] = {};
    const float* weights[kHeld] = {};
    uint64_t W[kHeld] = {}, H[kHeld] = {};
  } held;
  // smesh_fuse_views, group pipeline: bank b (slots b * kMaxGroup ...) has been rasterised / its fusion has been queued
  hipEvent_t ev_bank_rendered[2] = {nullptr, nullptr}, ev_bank_consumed[2] = {nullptr, nullptr}, ev_main_fence = nullptr;
  hipEvent_t ev_raster_done = nullptr;   // main_after_raster()
  uint64_t group_seq = 0;
  bool bank_used[2] = {false, false};
  uint64_t fused_seq = 0;
  // the index planes handed out by the last TWO smesh_renderer_render_device() calls (they alternate between the two
  // sets of per-triangle records): last_idx[s] is the plane side[s] still describes, or null
  // (kRecordSides of them since round 2: the reference's harness queues up to three rendered views between its render
  // and its add thread, eval_scannet.py:189-238; smesh_renderer_find_render searches them by CONTENT)
  const uint32_t* last_idx[kRecordSides] = {};
  uint64_t last_W[kRecordSides] = {}, last_H[kRecordSides] = {};
  uint64_t render_seq = 0;
  // smesh_aggregator_add_matched: rec_hash[s] != null-state while side[s]'s records are intact -- the 64-bit content checksum
  // of the plane they describe lives in d_hash[s] (d_hash[kRecordSides]: the checksum of the image being looked up)
  bool rec_valid[kRecordSides] = {};
  bool hash_valid[kRecordSides] = {};   // d_hash[s] holds the checksum of the plane (smesh_renderer_seal_render)
  unsigned long long* d_hash = nullptr;
  hipEvent_t hash_ev[kRecordSides] = {};   // recorded on the main stream behind the kernel that writes d_hash[s]
  Scratch match_stage;             // device copy of a host index image that is being looked up
  int last_render_side = 0;        // the side render_into() filled last (smesh_renderer_render_stats)
  bool raster_pending = false;     // work queued on the raster stream since the last synchronisation
  bool main_pending = false;       // renderer state (keys, scratch) used on the main stream since then
  std::mutex mu;
};

// 64-bit content checksum of an index plane: sum over the pixels of a bijective mix of (position, value).  Changing any single
// pixel changes it with certainty, any other difference with probability 1 - 2^-64; the sum makes it independent of the
// order in which blocks finish.
namespace {
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {   // splitmix64 finaliser (a bijection)
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_plane_checksum(const uint32_t* __restrict__ img, uint64_t N, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long part[4];
  unsigned long long h = 0ull;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (uint64_t)gridDim.x * blockDim.x)
    h += mix64((i << 32) | (unsigned long long)img[i]);
  for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
  __syncthreads();
  // one atomic per workgroup: atomics on ONE address are served one after the other (8192 of them took 99 us)
  if (threadIdx.x == 0) atomicAdd(out, ((part[0] + part[1]) + part[2]) + part[3]);
}
int plane_checksum(DeviceCtx* ctx, hipStream_t st, const uint32_t* d_img, uint64_t N, unsigned long long* d_out) {
  SMESH_HIP(hipMemsetAsync(d_out, 0, sizeof(unsigned long long), st));
  const uint32_t blocks = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(div_up(N, 256 * 16), (uint64_t)ctx->num_cus));
  hipLaunchKernelGGL(k_plane_checksum, dim3(blocks), dim3(256), 0, st, d_img, N, d_out);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}
}  // namespace

namespace {
bool no_huge_possible(const smesh_renderer* r, const smesh_camera_t* cam);
bool no_big_possible(const smesh_renderer* r, const smesh_camera_t* cam);
void mesh_bounds(const float* v, uint64_t V, const int32_t* f, uint64_t F, smesh_renderer::Bounds& b);

int ensure_keys(smesh_renderer::ViewScratch& vs, uint64_t W, uint64_t H, hipStream_t st) {
  const uint64_t N = div_up(W, 4) * div_up(H, 4) * 16;   // 4 x 4 blocked layout, padded
  if (N <= vs.keys_pixels) return SMESH_OK;
  if (vs.keys) SMESH_HIP(dev_free(vs.keys));
  vs.keys = nullptr;
  vs.keys_pixels = 0;
  SMESH_HIP(dev_malloc(reinterpret_cast<void**>(&vs.keys), N * 8));
  vs.keys_pixels = N;
  hipLaunchKernelGGL(k_fill_keys, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, vs.keys, N);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

// The main stream takes over renderer state that the raster stream may still be using (the last group of a pipelined
// smesh_fuse_views call): a device-side wait on an event, never a host synchronisation -- the group pipeline is the default since
// round 5, and a plain render() or a ranged job right behind a fuse_views call must not stall the host for it.
int main_after_raster(smesh_renderer* r) {
  if (!r->raster_pending) return SMESH_OK;
  if (!r->ev_raster_done) SMESH_HIP(hipEventCreateWithFlags(&r->ev_raster_done, hipEventDisableTiming));
  SMESH_HIP(hipEventRecord(r->ev_raster_done, r->ctx->raster_stream));
  SMESH_HIP(hipStreamWaitEvent(r->ctx->stream, r->ev_raster_done, 0));
  r->raster_pending = false;
  return SMESH_OK;
}

enum class RasterPath { Frag, Direct };

RasterPath raster_path() {
  static const RasterPath p = [] {
    const char* e = getenv("SMESH_RASTER");
    const std::string v = e ? e : "";
    if (v == "direct") return RasterPath::Direct;
    return RasterPath::Frag;
  }();
  return p;
}

// May a fuse_view(s) call leave out the index plane of a view that has no queued triangles and no overflow (RasterArgs::idx_optional)?
// Only where the tile resolve clears the losers out of the records' masks itself: triangle primitives in the caller's face order,
// fragment-queue path.  SMESH_RASTER_SKIP_PLANE=0 turns it off.
bool plane_optional_allowed(const smesh_renderer* r) {
  static const bool off = getenv("SMESH_RASTER_SKIP_PLANE") && atoi(getenv("SMESH_RASTER_SKIP_PLANE")) == 0;
  return !off && !r->texels && !r->prim_id && raster_path() == RasterPath::Frag;
}

// RasterArgs::idx_optional for a fuse_view(s) call of this renderer into this aggregator: 0 = the planes are read (not the triangle-order
// kernels), 1 = they are read only for queued triangles and masks that need checking -- one decision per raster launch (view_needs_planes),
// 2 = the same, each view for itself: aggregators whose EVERY triangle-order launch is k_fuse_tri (smesh_aggregator_fuses_small_views_by_mask:
// not the name of the kernel -- 41 .. 48 classes go to k_fuse_tri_any when a launch holds more than two views, which is what the first
// attempt at level 2 tripped over: seeds 553071 / 702926 of tools/soup_sweep.py, C = 47 / 48).  SMESH_PLANE_LEVEL=1 forces level 1.
int plane_optional_level(smesh_renderer* r, smesh_aggregator* a) {
  if (r->texels || !smesh_aggregator_can_fuse_triangles(a, r->F)) return 0;
  static const int forced = getenv("SMESH_PLANE_LEVEL") ? atoi(getenv("SMESH_PLANE_LEVEL")) : 0;
  if (forced == 1) return 1;
  return smesh_aggregator_fuses_small_views_by_mask(a) ? 2 : 1;
}

// Consecutive triangle blocks per run of an XCD (xcd_block).  SMESH_RASTER_XCD=n sets it; 0: blocks take consecutive triangles in
// dispatch order, as until round 5.  cfg2, eight views per launch (profiles/r06_raster_experiments.txt): k_raster_frag_group fetches
// 149.6 MB with 0, 79.1 MB with 8 (x 2: gfx950's FETCH_SIZE unit) -- 562 -> 422 MB of traffic per launch -- and takes 195 -> 191 us; runs of
// 2 / 32 / 128: 195.5 / 201 / 211 us.
uint32_t raster_xcd_run() {
  static const uint32_t run = getenv("SMESH_RASTER_XCD") ? (uint32_t)std::max(0, atoi(getenv("SMESH_RASTER_XCD"))) : 8u;
  return run;
}

// Per-tile fragment queues (kQSub sub-queues each).  Capacity per tile: 16 fragments per pixel of the tile
// (SMESH_FRAG_CAP overrides the capacity of a sub-queue; fragments beyond it fall back to the global key image), shrunk
// if the whole set would exceed 8 GiB.  Returns false (queues unusable -> direct path) for images with so many tiles
// that a sub-queue would drop under 256 slots.
bool ensure_queues(smesh_renderer* r, smesh_renderer::ViewScratch& vs, uint64_t W, uint64_t H, hipStream_t st, int* status) {
  *status = SMESH_OK;
  const uint64_t tiles_x = div_up(W, kQW), tiles_y = div_up(H, kQH), ntiles = tiles_x * tiles_y;
  uint64_t cap = 16ull * kQPixels / kQSub;   // per sub-queue
  if (const char* e = getenv("SMESH_FRAG_CAP")) cap = std::max<uint64_t>(1, (uint64_t)atoll(e));
  else {
    const uint64_t budget = 8ull << 30;
    if (ntiles * kQSub * cap * 10 > budget) cap = budget / (ntiles * kQSub * 10);
    if (cap < 256) return false;
  }
  if (ntiles > vs.fq_tiles || (uint32_t)cap != vs.fq.cap) {
    (void)hipStreamSynchronize(r->ctx->stream);
    (void)hipStreamSynchronize(r->ctx->raster_stream);
    for (void* p : {(void*)vs.fq.key, (void*)vs.fq.pix, (void*)vs.fq.count, (void*)vs.fq.flag})
      if (p) (void)dev_free(p);
    vs.fq = FragQueues();
    vs.fq_tiles = 0;
    hipError_t e = dev_malloc(reinterpret_cast<void**>(&vs.fq.key), ntiles * kQSub * cap * 8);
    if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&vs.fq.pix), ntiles * kQSub * cap * 2);
    if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&vs.fq.count), ntiles * kQSub * 4);
    if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&vs.fq.flag), ntiles * 4);
    if (e == hipSuccess) e = hipMemsetAsync(vs.fq.count, 0, ntiles * kQSub * 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(vs.fq.flag, 0, ntiles * 4, st);
    if (e != hipSuccess) { *status = fail_hip(e, "fragment queue allocation", __FILE__, __LINE__); return false; }
    vs.fq.cap = (uint32_t)cap;
    vs.fq_tiles = ntiles;
  }
  vs.fq.tiles_y = (uint32_t)tiles_y;
  return true;
}

CameraArgs camera_args(const smesh_camera_t* cam) {
  CameraArgs ca;
  memcpy(ca.R, cam->rotation, sizeof ca.R);
  memcpy(ca.t, cam->translation, sizeof ca.t);
  ca.fx = cam->focal[0]; ca.fy = cam->focal[1]; ca.cx = cam->principal[0]; ca.cy = cam->principal[1];
  ca.W = (uint32_t)cam->width; ca.H = (uint32_t)cam->height;
  return ca;
}

// Kernel arguments of a render of a W x H view with scratch set `vs`, leaving its records in side `side` (queues: a.q).
// An ESTIMATE (a tuning input, never a proof: every path rasterises every triangle correctly) of the screen size in pixels of a typical
// edge of the mesh for this camera: the mean edge at the depth of the centre of the mesh's box.  0: no estimate.
double typical_edge_pixels(const smesh_renderer* r, const smesh_camera_t* cam) {
  const smesh_renderer::Bounds& b = r->bounds;
  if (!b.valid || !(b.mean_edge > 0.0)) return 0.0;
  double z = cam->translation[2];
  for (int d = 0; d < 3; d++) z += (double)cam->rotation[6 + d] * 0.5 * (b.lo[d] + b.hi[d]);
  if (!(z > 1e-3)) return 0.0;
  const double e = b.mean_edge * std::max(std::fabs(cam->focal[0]), std::fabs(cam->focal[1])) / z;
  return std::isfinite(e) ? e : 0.0;
}
// How unevenly the mesh's triangles are sized in this view: farthest over nearest camera-space depth of the corners of the mesh's box
// (+inf: the camera is inside or beside it).  An estimate like typical_edge_pixels: it only chooses between paths.  A ring camera
// around BASELINE's meshes: 3 - 4; a camera above the surface looking along it: +inf.  1080p, one render(): 90 000 triangles from inside
// 0.433 ms with the triangles a wave walks one at a time left to their waves, 0.205 through k_raster_medium; the same mesh from
// outside 0.092 / 0.092, 10 000 triangles from outside 0.100 / 0.103 (eight views per launch: 0.079 / 0.087).
double depth_spread(const smesh_renderer* r, const smesh_camera_t* cam) {
  const smesh_renderer::Bounds& b = r->bounds;
  if (!b.valid) return 1.0;
  double zmin = INFINITY, zmax = -INFINITY;
  for (int c = 0; c < 8; c++) {
    double z = cam->translation[2];
    for (int d = 0; d < 3; d++) z += (double)cam->rotation[6 + d] * ((c >> d) & 1 ? b.hi[d] : b.lo[d]);
    zmin = std::min(zmin, z); zmax = std::max(zmax, z);
  }
  if (!(zmax > 0.0)) return 1.0;          // (behind the camera altogether)
  return zmin > 1e-3 * zmax ? zmax / zmin : INFINITY;
}
// Spread waves (raster_frag_wave) for this launch?  SMESH_RASTER_SPREAD=0 / 1 forces; else for ONE view whose typical edge measures
// 10.5 .. 19 pixels (boxes of 9 .. 24).  1080p, one render() at a time, ms (lanes walking their own boxes in rounds / spread waves;
// NOTES/round5.md 11): 25 600 triangles 0.129 / 0.092, 40 000 0.146 / 0.084, 62 500 0.128 / 0.095 -- 19 600 (a fifth of the boxes
// beyond 24 pixels) 0.100 / 0.122, 90 000 (half of them within 8) 0.092 / 0.100.  The views of a group fill the chip with waves of 32
// or 64 triangles: 40 000 triangles 0.080 / 0.085 per view, 90 000 0.077 / 0.113.
bool want_spread(const smesh_renderer* r, const smesh_camera_t* cam, int nviews) {
  static const int knob = getenv("SMESH_RASTER_SPREAD") ? atoi(getenv("SMESH_RASTER_SPREAD")) : -1;
  if (knob == 0 || !cam) return false;
  if (knob == 1) return true;
  if (nviews > 1) return false;
  if (depth_spread(r, cam) > 6.0) return false;     // (a view from inside: the estimate says nothing about most of its triangles)
  const double e = typical_edge_pixels(r, cam);
  return e >= 10.5 && e <= 19.0;
}

RasterArgs raster_args(smesh_renderer* r, smesh_renderer::ViewScratch& vs, int side, uint64_t W, uint64_t H, int nviews = 1,
                       const smesh_camera_t* cam = nullptr) {
  RasterArgs a;
  a.faces = r->faces; a.verts = r->verts; a.sv = vs.sv; a.tex_res = r->texels ? r->tex_res : nullptr; a.tex_first = r->tex_first;
  a.prim_id = r->prim_id;
  a.keys = vs.keys; a.F = r->F; a.V = r->V; a.W = (uint32_t)W; a.H = (uint32_t)H;
  a.big_queue = r->side[side].big_queue; a.big_count = r->side[side].big_count; a.big_capacity = r->big_capacity;
  a.huge_queue = vs.huge_queue;
  a.med_queue = vs.med_queue;
  {   // SMESH_RASTER_BALANCE=0 / 1 forces; else for views from inside or close to the scene (depth_spread)
    static const int knob = getenv("SMESH_RASTER_BALANCE") ? atoi(getenv("SMESH_RASTER_BALANCE")) : -1;
    a.balance = (vs.med_queue && (knob >= 0 ? knob != 0 : (cam && depth_spread(r, cam) > 6.0))) ? (nviews > 1 ? 1u : 2u) : 0u;
  }
  a.frags = r->side[side].frags;
  a.kinds = r->side[side].kinds;
  { static const int rdbg = SMESH_ABL_ENV("SMESH_RDBG"); a.dbg = rdbg; }
  a.q = FragQueues();
  a.idx_optional = 0u;
  a.tpw = 64;   // small meshes: fewer triangles per wave, at least ~kMinWaves waves
  // Meshes of 32 768 triangles and more count the waves of the whole launch (`nviews` views) and are content with 1 024 of them: their
  // boxes over 8 x 8 are mostly the ones a lane walks itself, sub-box by sub-box (raster_frag_64), and a round of that costs the same
  // for 16 lanes as for 64 -- 1080p, fuse_views, ms per view: 90 000 triangles 0.086 -> 0.077, 40 000 triangles 0.100 -> 0.081.
  // Coarser meshes need the waves for the cooperative loop (10 000 triangles: 0.084 with 2 048 waves per view, 0.091 with 1 024).
  static const uint64_t min_waves_env = getenv("SMESH_RASTER_MIN_WAVES") ? (uint64_t)std::max(1, atoi(getenv("SMESH_RASTER_MIN_WAVES"))) : 0u;
  const bool fine = r->F >= 32768u;
  const uint64_t min_waves = min_waves_env ? min_waves_env : (fine ? 1024u : 2048u);
  const uint64_t launch_views = fine ? (uint64_t)std::max(1, nviews) : 1u;
  while (a.tpw > 1 && launch_views * r->F / a.tpw < min_waves) a.tpw >>= 1;
  a.groups = 1;  // (frag_groups() decides per launch)
  a.spread = 0u;
  {   // (SMESH_RASTER_WG_PUSH=0 / 1 forces)
    static const int knob = getenv("SMESH_RASTER_WG_PUSH") ? atoi(getenv("SMESH_RASTER_WG_PUSH")) : -1;
    a.wg_push = knob >= 0 ? (uint32_t)(knob != 0) : (cam && typical_edge_pixels(r, cam) >= 5.0 ? 1u : 0u);
  }
  if (want_spread(r, cam, nviews)) { a.spread = (uint32_t)kSpread; a.tpw = (uint32_t)kSpreadTris; }
  return a;
}

// Rasterise into caller-provided device planes (d_depth may be null: index plane only).
// Groups of 64 triangles per wave in the grouped launches (k_raster_frag_group; the next group's vertex indices prefetched): four for
// meshes of four million triangles and more, when the launch still has four waves for every slot of the chip (1024 SIMDs x 5); else one.
// SMESH_RASTER_GROUPS=n forces n.
uint32_t frag_groups(uint64_t F, int views, uint32_t tpw) {
  static const int knob = getenv("SMESH_RASTER_GROUPS") ? atoi(getenv("SMESH_RASTER_GROUPS")) : 0;
  if (tpw != 64u) return 1u;
  if (knob >= 1) return (uint32_t)knob;
  return (F >= 4000000u && (uint64_t)views * div_up(F, 256u) >= 20480u) ? 4u : 1u;
}

int render_into(smesh_renderer* r, const smesh_camera_t* cam, uint32_t* d_idx, float* d_depth, hipStream_t st = nullptr,
                int side = 0, int idx_optional = 0) {
  DeviceCtx* ctx = r->ctx;
  r->last_render_side = side;
  if (!st) {
    // plain render()/render_device(): main stream, after whatever smesh_fuse_view left on the raster stream
    st = ctx->stream;
    SMESH_TRY(main_after_raster(r));
    r->main_pending = true;
  }
  const uint64_t W = cam->width, H = cam->height, N = W * H;
  smesh_renderer::ViewScratch& vs = r->vs[0];
  SMESH_TRY(ensure_keys(vs, W, H, st));
  ProfScope prof(ctx, SMESH_PROF_RASTER, st);
  const CameraArgs ca = camera_args(cam);
  RasterArgs a = raster_args(r, vs, side, W, H, 1, cam);
  a.cam = ca;
  if (r->V) {
    hipLaunchKernelGGL(k_project_vertices, dim3((uint32_t)div_up(r->V, 256)), dim3(256), 0, st, r->verts, r->V, ca, vs.sv,
                       r->side[side].big_count);
    SMESH_HIP(hipGetLastError());
  } else {
    SMESH_HIP(hipMemsetAsync(r->side[side].big_count, 0, 32, st));
  }
  if (r->F) {
    a.idx_optional = (idx_optional && plane_optional_allowed(r)) ? (uint32_t)idx_optional : 0u;
    const uint32_t big_grid = (uint32_t)std::min<uint64_t>(r->F, (uint64_t)ctx->num_cus);
    int qs = SMESH_OK;
    if (raster_path() == RasterPath::Frag && ensure_queues(r, vs, W, H, st, &qs)) {
      a.q = vs.fq;
      {
        const uint32_t nblocks = (uint32_t)div_up(div_up(r->F, a.tpw), 4);
        const uint32_t chunk = nblocks >= 64u ? raster_xcd_run() : 0u;
        const dim3 grid(chunk ? 8u * xcd_slots(nblocks, chunk) : nblocks);
        switch (raster_mode(a)) {
          case 7:  hipLaunchKernelGGL(k_raster_frag<7>, grid, dim3(256), 0, st, a, nblocks, chunk); break;
          case 5:  hipLaunchKernelGGL(k_raster_frag<5>, grid, dim3(256), 0, st, a, nblocks, chunk); break;
          case 4:  hipLaunchKernelGGL(k_raster_frag<4>, grid, dim3(256), 0, st, a, nblocks, chunk); break;
          case 3:  hipLaunchKernelGGL(k_raster_frag<3>, grid, dim3(256), 0, st, a, nblocks, chunk); break;
          case 1:  hipLaunchKernelGGL(k_raster_frag<1>, grid, dim3(256), 0, st, a, nblocks, chunk); break;
          default: hipLaunchKernelGGL(k_raster_frag<0>, grid, dim3(256), 0, st, a, nblocks, chunk); break;
        }
        if ((raster_mode(a) & 3) != 0 && a.balance) {    // the list the balanced instances left: four waves per SIMD's worth of waves walk it
          const dim3 mgrid(4u * (uint32_t)std::max(1, ctx->num_cus));
          if (a.tex_res) hipLaunchKernelGGL(k_raster_medium<true>, mgrid, dim3(256), 0, st, a);
          else hipLaunchKernelGGL(k_raster_medium<false>, mgrid, dim3(256), 0, st, a);
        }
      }
      SMESH_HIP(hipGetLastError());
      const uint32_t ntiles = (uint32_t)(div_up(W, kQW) * div_up(H, kQH));
      if (!no_huge_possible(r, cam)) hipLaunchKernelGGL(k_raster_huge, dim3(std::min<uint32_t>(ntiles, 2u * (uint32_t)ctx->num_cus)), dim3(256), 0, st, a, ntiles);
      SMESH_HIP(hipGetLastError());
      hipLaunchKernelGGL(k_tile_resolve, dim3((uint32_t)(div_up(W, kQW) * div_up(H, kQH))), dim3(256), 0, st, a, d_idx, d_depth);
      SMESH_HIP(hipGetLastError());
      return SMESH_OK;
    }
    SMESH_TRY(qs);
    hipLaunchKernelGGL(k_raster_small, dim3((uint32_t)div_up(r->F, 256)), dim3(256), 0, st, a);
    SMESH_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_raster_big, dim3(big_grid), dim3(256), 0, st, a, 0u);
    SMESH_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_resolve, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, vs.keys, d_idx, d_depth, (uint32_t)W, (uint32_t)H);
  SMESH_HIP(hipGetLastError());
  // the direct path leaves the emitted fragments in the records, winners and losers alike: the fusion checks them
  SMESH_HIP(hipMemsetAsync(r->side[side].big_count + 1, 0xFF, 4, st));
  return SMESH_OK;
}

hipError_t alloc_side(smesh_renderer* r, int i);

// Scratch of view slot `i` (smesh_fuse_views rasterises several views per launch).
hipError_t alloc_scratch(smesh_renderer* r, int i) {
  smesh_renderer::ViewScratch& vs = r->vs[i];
  hipError_t e = hipSuccess;
  if (!vs.sv) e = dev_malloc(reinterpret_cast<void**>(&vs.sv), std::max<uint64_t>(r->V * sizeof(ScreenVertex), 16));
  if (e == hipSuccess && !vs.huge_queue) e = dev_malloc(reinterpret_cast<void**>(&vs.huge_queue), (size_t)r->big_capacity * 4);
  if (e == hipSuccess && !vs.med_queue) e = dev_malloc(reinterpret_cast<void**>(&vs.med_queue), (size_t)r->big_capacity * 4);
  return e;
}

// Can a W x H view go through fragment queues of a size that kMaxGroup view slots can afford?  (ensure_queues' sizing rule.)
bool queues_fit_group(uint64_t W, uint64_t H) {
  if (raster_path() != RasterPath::Frag) return false;
  if (getenv("SMESH_FRAG_CAP")) return true;
  const uint64_t ntiles = div_up(W, kQW) * div_up(H, kQH);
  const uint64_t cap = 16ull * kQPixels / kQSub;
  return ntiles * kQSub * cap * 10 <= (2ull << 30);
}

// n <= kMaxGroup views (cams[v] -> view slot v: its own projected vertices, fragment queues, key image, queues of large
// triangles, per-triangle records and index plane r->fused[v]) through the fragment-queue rasteriser with ONE launch per
// stage: the vertex stage and the tile resolve are short kernels that fill a fraction of the chip, and every launch has a
// ramp and a tail, so n views' worth of blocks take less than n launches.  Index planes only.
// Everything a group of views needs in view slots base .. base + n - 1 (records, scratch, fragment queues, index planes), allocated
// and initialised on `st` -- what render_group_into does on first use, callable ahead of time for the OTHER bank of the group
// pipeline so that no allocation falls between two groups.
int prepare_group_slots(smesh_renderer* r, const smesh_camera_t* cams, int n, hipStream_t st, int base) {
  for (int v = 0; v < n; v++) {
    const uint64_t W = cams[v].width, H = cams[v].height, N = W * H;
    SMESH_HIP(alloc_side(r, base + v));
    SMESH_HIP(alloc_scratch(r, base + v));
    smesh_renderer::ViewScratch& vs = r->vs[base + v];
    SMESH_TRY(ensure_keys(vs, W, H, st));
    int qs = SMESH_OK;
    if (!ensure_queues(r, vs, W, H, st, &qs)) return qs != SMESH_OK ? qs : fail(SMESH_ERR_RUNTIME, "fragment queues unavailable");
    if (r->fused[base + v].bytes < N * 8) {
      SMESH_HIP(hipStreamSynchronize(st));   // growing a slot frees the old buffer: nothing may still be reading it
      SMESH_HIP(hipStreamSynchronize(r->ctx->stream));
      SMESH_TRY(r->fused[base + v].reserve(N * 8));
    }
  }
  return SMESH_OK;
}

// `side_base`: where the records and index planes go (side[side_base + v], fused[side_base + v]) when that is not the view slots'
// own set (held views: the rasteriser scratch of slots base .. base + n - 1 is free again after the launches, the records stay).
int render_group_into(smesh_renderer* r, const smesh_camera_t* cams, int n, hipStream_t st, int base = 0, int side_base = -1,
                      int idx_optional = 0) {
  if (side_base < 0) side_base = base;
  if (!plane_optional_allowed(r)) idx_optional = 0;
  DeviceCtx* ctx = r->ctx;
  ProjectGroup pg;
  RasterGroup rg;
  memset(&pg, 0, sizeof pg);
  memset(&rg, 0, sizeof rg);
  uint32_t tiles = 0;
  for (int v = 0; v < n; v++) {
    const uint64_t W = cams[v].width, H = cams[v].height, N = W * H;
    SMESH_HIP(alloc_side(r, side_base + v));
    SMESH_HIP(alloc_scratch(r, base + v));
    smesh_renderer::ViewScratch& vs = r->vs[base + v];
    SMESH_TRY(ensure_keys(vs, W, H, st));
    int qs = SMESH_OK;
    if (!ensure_queues(r, vs, W, H, st, &qs)) return qs != SMESH_OK ? qs : fail(SMESH_ERR_RUNTIME, "fragment queues unavailable");
    if (r->fused[side_base + v].bytes < N * 8) {
      SMESH_HIP(hipStreamSynchronize(st));   // growing a slot frees the old buffer: nothing may still be reading it
      SMESH_HIP(hipStreamSynchronize(r->ctx->stream));
      SMESH_TRY(r->fused[side_base + v].reserve(N * 8));
    }
    if (side_base + v < kRecordSides) { r->last_idx[side_base + v] = nullptr; r->rec_valid[side_base + v] = false; }   // the records of a render_device() on this side are being overwritten
    pg.cam[v] = camera_args(&cams[v]);
    pg.sv[v] = vs.sv;
    pg.big_count[v] = r->side[side_base + v].big_count;
    rg.view[v] = raster_args(r, vs, side_base + v, W, H, n);
    { const RasterArgs own = raster_args(r, vs, side_base + v, W, H, 1, &cams[v]); rg.view[v].wg_push = own.wg_push; rg.view[v].balance = own.balance; }
    rg.view[v].cam = pg.cam[v];
    rg.view[v].q = vs.fq;
    rg.view[v].idx_optional = (uint32_t)idx_optional;
    rg.idx[v] = static_cast<uint32_t*>(r->fused[side_base + v].ptr);
    tiles += (uint32_t)(div_up(W, kQW) * div_up(H, kQH));
    rg.tile_end[v] = tiles;
  }
  pg.verts = r->verts; pg.V = r->V;
  pg.n = (uint32_t)n;
  rg.n = (uint32_t)n;
  {   // spread waves: one decision for the launch (its views share the grid): where at least half of the views ask for them
    int votes = 0;
    for (int v = 0; v < n; v++) votes += want_spread(r, &cams[v], n) ? 1 : 0;
    if (2 * votes >= n && votes > 0)
      for (int v = 0; v < n; v++) { rg.view[v].spread = (uint32_t)kSpread; rg.view[v].tpw = (uint32_t)kSpreadTris; }
    int push_votes = 0;
    for (int v = 0; v < n; v++) push_votes += rg.view[v].wg_push ? 1 : 0;
    for (int v = 0; v < n; v++) rg.view[v].wg_push = 2 * push_votes >= n && push_votes > 0 ? 1u : 0u;
    int balance_votes = 0;
    for (int v = 0; v < n; v++) balance_votes += rg.view[v].balance ? 1 : 0;
    for (int v = 0; v < n; v++) rg.view[v].balance = 2 * balance_votes >= n && balance_votes > 0 ? (n > 1 ? 1u : 2u) : 0u;
  }
  for (int v = 0; v < n; v++) rg.view[v].groups = frag_groups(r->F, n, rg.view[0].tpw);
  rg.blocks_per_view = (uint32_t)div_up(div_up(r->F, rg.view[0].tpw * rg.view[0].groups), 4);
  ProfScope prof(ctx, SMESH_PROF_RASTER, st);
  hipLaunchKernelGGL(k_project_vertices_group, dim3((uint32_t)div_up(r->V, 256)), dim3(256), 0, st, pg);
  SMESH_HIP(hipGetLastError());
  // (experiment knob: an LDS pad caps the rasteriser's workgroups per CU when it runs beside a fusion launch -- group pipeline)
  static const unsigned raster_pad = getenv("SMESH_RASTER_LDS_PAD") ? (unsigned)atoi(getenv("SMESH_RASTER_LDS_PAD")) : 0u;
  {
    rg.chunk = rg.blocks_per_view >= 64u ? raster_xcd_run() : 0u;
    const dim3 grid(rg.chunk ? 8u * xcd_slots(rg.blocks_per_view, rg.chunk) * (uint32_t)n : (uint32_t)n * rg.blocks_per_view);
    const unsigned pad = st == ctx->raster_stream ? raster_pad : 0u;
    switch (raster_mode(rg.view[0])) {     // (one mode per launch: render_group_into made the views agree)
      case 7:  hipLaunchKernelGGL(k_raster_frag_group<7>, grid, dim3(256), pad, st, rg); break;
      case 5:  hipLaunchKernelGGL(k_raster_frag_group<5>, grid, dim3(256), pad, st, rg); break;
      case 4:  hipLaunchKernelGGL(k_raster_frag_group<4>, grid, dim3(256), pad, st, rg); break;
      case 3:  hipLaunchKernelGGL(k_raster_frag_group<3>, grid, dim3(256), pad, st, rg); break;
      case 1:  hipLaunchKernelGGL(k_raster_frag_group<1>, grid, dim3(256), pad, st, rg); break;
      default: hipLaunchKernelGGL(k_raster_frag_group<0>, grid, dim3(256), pad, st, rg); break;
    }
    if ((raster_mode(rg.view[0]) & 3) != 0 && rg.view[0].balance) {
      const uint32_t per_view = std::max(64u, 4u * (uint32_t)std::max(1, ctx->num_cus) / (uint32_t)n);
      if (rg.view[0].tex_res) hipLaunchKernelGGL(k_raster_medium_group<true>, dim3((uint32_t)n * per_view), dim3(256), 0, st, rg, per_view);
      else hipLaunchKernelGGL(k_raster_medium_group<false>, dim3((uint32_t)n * per_view), dim3(256), 0, st, rg, per_view);
    }
  }
  SMESH_HIP(hipGetLastError());
  bool huge_needed = false;   // (no_huge_possible: a proof that the queue of every view of the group stays empty)
  for (int v = 0; v < n; v++) huge_needed = huge_needed || !no_huge_possible(r, &cams[v]);
  if (huge_needed) hipLaunchKernelGGL(k_raster_huge_group, dim3(std::min<uint32_t>(tiles, 4u * (uint32_t)ctx->num_cus)), dim3(256), 0, st, rg);
  SMESH_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_tile_resolve_group, dim3(tiles), dim3(256), 0, st, rg);
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
      (void)dev_free(im.idx);
      (void)dev_free(im.depth);
      r->images.erase(r->images.begin() + i);
    } else {
      i++;
    }
  }
  ImagePair im{nullptr, nullptr, N, false, false};
  SMESH_HIP(dev_malloc(reinterpret_cast<void**>(&im.idx), N * 4));
  hipError_t e = dev_malloc(reinterpret_cast<void**>(&im.depth), N * 4);
  if (e != hipSuccess) { (void)dev_free(im.idx); return fail_hip(e, "hipMalloc depth plane", __FILE__, __LINE__); }
  r->images.push_back(im);
  *out = &r->images.back();
  return SMESH_OK;
}

int check_camera(const smesh_camera_t* cam) {
  if (!cam) return fail(SMESH_ERR_INVALID, "camera is NULL");
  if (cam->width == 0 || cam->height == 0 || cam->width > 65536 || cam->height > 65536)
    return fail(SMESH_ERR_INVALID, "camera resolution must be in [1, 65536]");
  // the kernels pack pixel offsets into 32 bits and size their grids with 32-bit block counts (same bound as smesh_aggregator_add)
  if (cam->width * cam->height >= 0x7FFFFFFFull / 4) return fail(SMESH_ERR_INVALID, "image too large");
  return SMESH_OK;
}

hipError_t alloc_side(smesh_renderer* r, int i) {
  smesh_renderer::Side& sd = r->side[i];
  if (sd.frags) return hipSuccess;
  hipError_t e = dev_malloc(reinterpret_cast<void**>(&sd.big_queue), (size_t)r->big_capacity * 8);   // (upper half: the medium triangles again, push_mid)
  if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&sd.big_count), 32);
  if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&sd.frags), std::max<uint64_t>(r->F * sizeof(TriFrag), 16));
  if (e == hipSuccess && r->texels) e = dev_malloc(reinterpret_cast<void**>(&sd.kinds), std::max<uint64_t>(r->F, 16));
  if (e == hipSuccess) e = hipMemsetAsync(sd.big_count, 0, 32, r->ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(r->ctx->stream);
  return e;
}

// Box of the vertices and longest edge of the faces, on the host, once per renderer (a few ms per million faces).
void mesh_bounds(const float* v, uint64_t V, const int32_t* f, uint64_t F, smesh_renderer::Bounds& b) {
  b.valid = false;
  if (!V || !F) return;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  bool finite = true;
  for (uint64_t i = 0; i < V; i++)
    for (int d = 0; d < 3; d++) {
      const float x = v[3 * i + d];
      finite = finite && std::isfinite(x);
      lo[d] = std::min(lo[d], x); hi[d] = std::max(hi[d], x);
    }
  if (!finite) return;
  double m2 = 0.0, sum_edges = 0.0;
  uint64_t edges = 0;
  for (uint64_t i = 0; i < F; i++) {
    const int32_t a = f[3 * i], c = f[3 * i + 1], e = f[3 * i + 2];
    if (a < 0 || c < 0 || e < 0 || (uint64_t)a >= V || (uint64_t)c >= V || (uint64_t)e >= V) continue;   // (load_tri drops such faces)
    const float* p[3] = {v + 3 * (uint64_t)a, v + 3 * (uint64_t)c, v + 3 * (uint64_t)e};
    for (int k = 0; k < 3; k++) {
      const float* q = p[k]; const float* w = p[(k + 1) % 3];
      const double dx = (double)q[0] - w[0], dy = (double)q[1] - w[1], dz = (double)q[2] - w[2];
      m2 = std::max(m2, dx * dx + dy * dy + dz * dz);
      sum_edges += std::sqrt(dx * dx + dy * dy + dz * dz);
      edges++;
    }
  }
  b.mean_edge = edges ? sum_edges / (double)edges : 0.0;
  for (int d = 0; d < 3; d++) { b.lo[d] = lo[d]; b.hi[d] = hi[d]; }
  b.max_edge = std::sqrt(m2);
  b.valid = std::isfinite(b.max_edge);
}

// An upper bound, PROVEN from the mesh's box and its longest edge, on the screen extent (pixels, either axis) of every triangle that
// reaches the image of this camera -- or +inf when there is no proof (a vertex at or behind the near plane is possible, coarse
// triangles, no valid summary).  It lets the host leave out launches whose work lists are then empty by construction (at cfg2 an
// empty k_raster_huge_group cost 7 us of kernel time plus a launch gap per group of eight views).  Never a guess:
// whenever the bound does not hold the launches happen as before.  With z_lo the smallest camera-space depth of the box (float32
// rounding of the vertex stage subtracted) and L the longest edge as the vertex stage sees it, r = L / z_lo: two vertices P, Q of one
// triangle project
//   |u_P - u_Q| = |fx| |x_P / z_P - x_Q / z_Q| <= (|fx| |x_P - x_Q| + |u_Q - cx| |z_P - z_Q|) / z_P <= (|fx| + |u_Q - cx|) r,
// and a triangle that reaches the image has its leftmost vertex Q at -D - 1 <= u_Q <= W + 1 (D: its extent), so
//   D <= (|fx| + max(|cx|, |W - cx|) + 1) r / (1 - r); likewise in v.
double box_extent_bound(const smesh_renderer::Bounds& b, const smesh_camera_t* cam) {
  if (!b.valid) return INFINITY;
  const double W = (double)cam->width, H = (double)cam->height;
  double zmin = INFINITY, mag = 0.0, frob2 = 0.0;
  for (int row = 0; row < 3; row++) {
    double m = std::fabs((double)cam->translation[row]);
    for (int d = 0; d < 3; d++) {
      const double R = cam->rotation[3 * row + d];
      m += std::fabs(R) * std::max(std::fabs(b.lo[d]), std::fabs(b.hi[d]));
      frob2 += R * R;
    }
    mag = std::max(mag, m);
  }
  for (int c = 0; c < 8; c++) {
    double z = cam->translation[2];
    for (int d = 0; d < 3; d++) z += (double)cam->rotation[6 + d] * ((c >> d) & 1 ? b.hi[d] : b.lo[d]);
    zmin = std::min(zmin, z);
  }
  const double eps = 32.0 * 5.9604644775390625e-08 * mag;            // float32 evaluation of R X + t (project_point): < 4 ulp of `mag`
  const double z_lo = zmin - eps;
  const double L = std::sqrt(frob2) * b.max_edge + 4.0 * eps;       // |R (P - Q)| <= ||R||_F |P - Q|, plus the rounding of both points
  if (!(z_lo > 1e-3) || !std::isfinite(mag) || !std::isfinite(L)) return INFINITY;   // (kNear = 1e-6: nothing behind or on the near plane)
  const double ratio = L / z_lo;
  if (!(ratio < 0.25)) return INFINITY;
  const double fx = std::fabs(cam->focal[0]), fy = std::fabs(cam->focal[1]);
  const double ax = std::max(std::fabs(cam->principal[0]), std::fabs(W - cam->principal[0])) + 1.0;
  const double ay = std::max(std::fabs(cam->principal[1]), std::fabs(H - cam->principal[1])) + 1.0;
  const double du = (fx + ax) * ratio / (1.0 - ratio), dv = (fy + ay) * ratio / (1.0 - ratio);
  if (!std::isfinite(du) || !std::isfinite(dv)) return INFINITY;
  return std::max(du, dv);
}
// A box of n pixel centres needs an extent of at least n - 1: no triangle crosses the near plane or has a box of more than kMedium
// pixels a side (the queue of k_raster_huge stays empty) ...
double box_extent_bound(const smesh_renderer* r, const smesh_camera_t* cam) {
  static const bool off = getenv("SMESH_HUGE_ALWAYS") && atoi(getenv("SMESH_HUGE_ALWAYS")) != 0;
  return off ? INFINITY : box_extent_bound(r->bounds, cam);
}
bool no_huge_possible(const smesh_renderer* r, const smesh_camera_t* cam) { return box_extent_bound(r, cam) <= (double)kMedium - 4.0; }
// ... or of more than 8 pixels a side: the big-triangle queue of the view (and with it the list of medium triangles) stays empty -- a
// mesh of millions of triangles seen from outside (cfg4: 3.5 pixels, cfg5: 5.6; not cfg2: 13 - 24 against boxes that do stay under 8).
bool no_big_possible(const smesh_renderer* r, const smesh_camera_t* cam) { return box_extent_bound(r, cam) <= 6.5; }

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
  mesh_bounds(vertices, V, faces, F, r->bounds);
  hipError_t e = dev_malloc(reinterpret_cast<void**>(&r->verts), std::max<uint64_t>(V * 12, 16));
  if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&r->faces), std::max<uint64_t>(F * 12, 16));
  if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&r->vs[0].sv), std::max<uint64_t>(V * sizeof(ScreenVertex), 16));
  if (e == hipSuccess) e = alloc_side(r, 0);
  if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&r->vs[0].huge_queue), (size_t)r->big_capacity * 4);
  if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&r->vs[0].med_queue), (size_t)r->big_capacity * 4);
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

// Decides whether the face order is spatially scattered and, if so, returns the Morton order of the face centroids
// (position -> original face).  Metric: summed extent of the centroids of every 64 consecutive faces, against what
// neighbours would span (a few mean edge lengths); SMESH_REORDER=0 / 1 forces the answer.
static bool spatial_order(const float* v, uint64_t V, const int32_t* f, uint64_t F, std::vector<uint32_t>& order) {
  const char* env = getenv("SMESH_REORDER");
  if (env && atoi(env) == 0) return false;
  const bool force = env && atoi(env) == 1;
  if (F < 4096 && !force) return false;
  std::vector<float> cx(F), cy(F), cz(F);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, edge_sum = 0.0;
  uint64_t edges = 0;
  for (uint64_t i = 0; i < F; i++) {
    const int32_t a = f[3 * i], b = f[3 * i + 1], c = f[3 * i + 2];
    if (a < 0 || b < 0 || c < 0 || (uint64_t)a >= V || (uint64_t)b >= V || (uint64_t)c >= V) { cx[i] = cy[i] = cz[i] = 0.f; continue; }
    const float* pa = v + 3 * (uint64_t)a; const float* pb = v + 3 * (uint64_t)b; const float* pc = v + 3 * (uint64_t)c;
    cx[i] = (pa[0] + pb[0] + pc[0]) / 3.f; cy[i] = (pa[1] + pb[1] + pc[1]) / 3.f; cz[i] = (pa[2] + pb[2] + pc[2]) / 3.f;
    if (!std::isfinite(cx[i]) || !std::isfinite(cy[i]) || !std::isfinite(cz[i])) { cx[i] = cy[i] = cz[i] = 0.f; continue; }
    const double c3[3] = {cx[i], cy[i], cz[i]};
    for (int d = 0; d < 3; d++) { lo[d] = std::min(lo[d], c3[d]); hi[d] = std::max(hi[d], c3[d]); }
    if ((i & 15) == 0) { edge_sum += std::sqrt((double)(pa[0] - pb[0]) * (pa[0] - pb[0]) + (double)(pa[1] - pb[1]) * (pa[1] - pb[1]) + (double)(pa[2] - pb[2]) * (pa[2] - pb[2])); edges++; }
  }
  if (!(lo[0] <= hi[0]) || !edges) return false;
  auto spread = [&](const uint32_t* ord) {   // sum over groups of 64 consecutive faces of the extent of their centroids
    double total = 0.0;
    for (uint64_t g = 0; g < F; g += 64) {
      float l0 = 1e30f, l1 = 1e30f, l2 = 1e30f, h0 = -1e30f, h1 = -1e30f, h2 = -1e30f;
      const uint64_t e = std::min<uint64_t>(F, g + 64);
      for (uint64_t i = g; i < e; i++) {
        const uint64_t j = ord ? ord[i] : i;
        l0 = std::min(l0, cx[j]); h0 = std::max(h0, cx[j]); l1 = std::min(l1, cy[j]); h1 = std::max(h1, cy[j]);
        l2 = std::min(l2, cz[j]); h2 = std::max(h2, cz[j]);
      }
      total += (double)(h0 - l0) + (double)(h1 - l1) + (double)(h2 - l2);
    }
    return total;
  };
  const double before = spread(nullptr);
  const double neighbours = (double)((F + 63) / 64) * 16.0 * (edge_sum / (double)edges);   // ~8 x 8 triangles, two tangential axes
  if (!force && before <= 6.0 * neighbours) return false;
  // Morton code of the centroid, 21 bits per axis
  std::vector<std::pair<uint64_t, uint32_t>> keyed(F);
  auto spread_bits = [](uint64_t x) {
    x &= 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull; x = (x | x << 16) & 0x1F0000FF0000FFull; x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull; x = (x | x << 2) & 0x1249249249249249ull;
    return x;
  };
  const double sx = hi[0] > lo[0] ? 2097151.0 / (hi[0] - lo[0]) : 0.0, sy = hi[1] > lo[1] ? 2097151.0 / (hi[1] - lo[1]) : 0.0,
               sz = hi[2] > lo[2] ? 2097151.0 / (hi[2] - lo[2]) : 0.0;
  for (uint64_t i = 0; i < F; i++) {
    const uint64_t qx = (uint64_t)std::min(2097151.0, std::max(0.0, (cx[i] - lo[0]) * sx)), qy = (uint64_t)std::min(2097151.0, std::max(0.0, (cy[i] - lo[1]) * sy)),
                   qz = (uint64_t)std::min(2097151.0, std::max(0.0, (cz[i] - lo[2]) * sz));
    keyed[i] = {spread_bits(qx) | (spread_bits(qy) << 1) | (spread_bits(qz) << 2), (uint32_t)i};
  }
  std::sort(keyed.begin(), keyed.end());
  order.resize(F);
  for (uint64_t i = 0; i < F; i++) order[i] = keyed[i].second;
  if (!force && spread(order.data()) * 2.0 > before) { order.clear(); return false; }   // not worth it
  return true;
}

static thread_local const char* g_last_fuse_kernel = "none";
static thread_local const char* g_last_add_path = "none";   // "render-records" (the rasteriser's per-triangle records), or what add_device reports
void smesh_note_fuse(const char* kernel, const char* path) { g_last_fuse_kernel = kernel; g_last_add_path = path; }

// The fusion half of smesh_fuse_view / smesh_aggregator_add_rendered: `d_idx` is the index plane of the render
// whose per-triangle records sit in r->side[slot].
static int fuse_rendered(smesh_renderer* r, smesh_aggregator* a, int slot, const uint32_t* d_idx, const float* probs,
                         const float* weights, int memkind, uint64_t W, uint64_t H, int64_t ps0 = 0, int64_t ps1 = 0) {
  DeviceCtx* ctx = r->ctx;
  const uint64_t N = W * H;
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
  if (!r->texels && smesh_aggregator_can_fuse_triangles(a, r->F)) {
    // triangle primitives: every accumulator row is owned by its triangle's lane -- no atomics, no histogram
    // (ps0, ps1: DEVICE class vectors with element strides (ps0, ps1, 1), read in place by k_fuse_tri; 0, 0: the dense image)
    const RenderedView rv{r->side[slot].frags, r->side[slot].big_queue, r->side[slot].big_count, d_idx, d_probs, d_w, W, H, ps0, ps1, true};
    SMESH_TRY(smesh_aggregator_fuse_triangles(a, r->F, r->prim_id, r->big_capacity, &rv, 1));
    smesh_note_fuse(smesh_aggregator_fuse_kernel_name(a, r->prim_id != nullptr), "render-records");
  } else if (r->texels && smesh_aggregator_can_fuse_texels(a, r->num_primitives)) {
    // texel primitives: a triangle owns its texel rows, its lane read-modify-writes them without atomics
    SMESH_TRY(smesh_aggregator_fuse_texels(a, r->side[slot].frags, r->side[slot].kinds, r->F, r->tex_first, r->tex_res, r->side[slot].big_queue,
                                           r->side[slot].big_count, r->big_capacity, d_idx, d_probs, d_w, H));
    smesh_note_fuse("k_fuse_texel", "render-records");
  } else {
    SMESH_TRY(smesh_aggregator_add_device_contig(a, d_idx, d_probs, d_w, W, H));   // (reports its path itself)
  }
  return SMESH_OK;
}

extern "C" {

const char* smesh_last_fuse_kernel(void) { return g_last_fuse_kernel; }
const char* smesh_last_add_path(void) { return g_last_add_path; }

int smesh_renderer_create_triangles(const float* vertices, uint64_t V, const int32_t* faces, uint64_t F, int device,
                                    smesh_renderer_t** out) {
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if ((V && !vertices) || (F && !faces)) return fail(SMESH_ERR_INVALID, "NULL mesh arrays");
  // The kernels take 64 consecutive triangles per wave and rely on them being neighbours in space (shared cache lines,
  // few screen tiles per wave).  A mesh whose faces come in a scattered order is processed in Morton order instead;
  // the index image and the accumulator rows keep the caller's numbering through the position -> id table.
  std::vector<uint32_t> order;
  if (!spatial_order(vertices, V, faces, F, order)) return create_common(vertices, V, faces, F, device, out);
  std::vector<int32_t> hf(3 * F);
  for (uint64_t i = 0; i < F; i++)
    for (int k = 0; k < 3; k++) hf[3 * i + k] = faces[3 * (uint64_t)order[i] + k];
  smesh_renderer* r = nullptr;
  SMESH_TRY(create_common(vertices, V, hf.data(), F, device, &r));
  hipError_t e = dev_malloc(reinterpret_cast<void**>(&r->prim_id), F * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(r->prim_id, order.data(), F * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { smesh_renderer_destroy(r); return fail_hip(e, "primitive id table upload", __FILE__, __LINE__); }
  *out = r;
  return SMESH_OK;
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
  // (side 0 was made before the renderer knew what it was: its kind bytes now)
  if (!r->side[0].kinds && dev_malloc(reinterpret_cast<void**>(&r->side[0].kinds), std::max<uint64_t>(F, 16)) != hipSuccess) {
    smesh_renderer_destroy(r);
    return fail(SMESH_ERR_RUNTIME, "texel renderer: allocation of the kind bytes failed");
  }
  r->h_faces = std::move(hf);
  r->h_res.assign(F, 0);
  r->h_first.assign(F, 0);

  // max projected area over all cameras: an F x K reduction, done on the GPU
  std::vector<float> best(F, 0.0f);
  if (F && K) {
    // The area is taken from the faces in the CALLER's vertex order, as the reference does (:100-127 come before the
    // re-ordering of :129-146): the float32 area expression is not invariant under permutations of the three vertices, and
    // ceil() turns a last-bit difference into one more texel row.
    smesh_camera_t* d_cams = nullptr;
    float* d_best = nullptr;
    int32_t* d_faces = nullptr;
    hipError_t e = dev_malloc(reinterpret_cast<void**>(&d_cams), K * sizeof(smesh_camera_t));
    if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&d_best), F * 4);
    if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&d_faces), F * 12);
    if (e == hipSuccess) e = hipMemcpyAsync(d_cams, cameras, K * sizeof(smesh_camera_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_faces, faces, F * 12, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_texel_area, dim3((uint32_t)div_up(F, 128)), dim3(128), 0, ctx->stream, r->verts, d_faces, F, V,
                         d_cams, (uint32_t)K, d_best);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(best.data(), d_best, F * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_cams) (void)dev_free(d_cams);
    if (d_best) (void)dev_free(d_best);
    if (d_faces) (void)dev_free(d_faces);
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
  hipError_t e = dev_malloc(reinterpret_cast<void**>(&r->tex_res), std::max<uint64_t>(F * 4, 16));
  if (e == hipSuccess) e = dev_malloc(reinterpret_cast<void**>(&r->tex_first), std::max<uint64_t>(F * 4, 16));
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
  for (void* p : {(void*)r->prim_id, (void*)r->verts, (void*)r->faces, (void*)r->tex_res, (void*)r->tex_first})
    if (p) (void)dev_free(p);
  for (auto& sd : r->side)
    for (void* p : {(void*)sd.big_queue, (void*)sd.big_count, (void*)sd.frags, (void*)sd.kinds})
      if (p) (void)dev_free(p);
  for (auto& vs : r->vs)
    for (void* p : {(void*)vs.sv, (void*)vs.huge_queue, (void*)vs.med_queue, (void*)vs.keys, (void*)vs.fq.key, (void*)vs.fq.pix, (void*)vs.fq.count, (void*)vs.fq.flag})
      if (p) (void)dev_free(p);
  for (auto& im : r->images) { (void)dev_free(im.idx); (void)dev_free(im.depth); }
  r->own_idx.release();
  r->match_stage.release();
  if (r->d_hash) (void)dev_free(r->d_hash);
  for (int sd = 0; sd < kRecordSides; sd++)
    if (r->hash_ev[sd]) (void)hipEventDestroy(r->hash_ev[sd]);
  for (auto& f : r->fused) f.release();
  for (int i = 0; i < 2; i++) {
    if (i == 0 && r->ev_main_fence) (void)hipEventDestroy(r->ev_main_fence);
    if (r->ev_bank_rendered[i]) (void)hipEventDestroy(r->ev_bank_rendered[i]);
    if (r->ev_bank_consumed[i]) (void)hipEventDestroy(r->ev_bank_consumed[i]);
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
  // alternate between the two sides: add(idx) still finds the records of the render BEFORE the latest one (the reference's
  // harness adds view k on a worker thread while the main thread already renders view k+1, eval_scannet.py:189-238)
  const int side = (int)(r->render_seq++ % (uint64_t)kRecordSides);
  if (side != 0) SMESH_HIP(alloc_side(r, side));
  r->last_idx[side] = nullptr; r->rec_valid[side] = false;
  SMESH_TRY(render_into(r, cam, im->idx, im->depth, nullptr, side));
  for (int sd = 0; sd < kRecordSides; sd++)
    if (r->last_idx[sd] == im->idx) r->last_idx[sd] = nullptr;   // that plane went back to the pool and is being reused (the records
                                                                  // of its side stay valid: copies of it are still recognised by content)
  r->last_idx[side] = im->idx; r->last_W[side] = cam->width; r->last_H[side] = cam->height;
  r->rec_valid[side] = true;
  r->hash_valid[side] = false;   // computed when the plane's content first leaves the library (smesh_renderer_seal_render)
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
  r->last_idx[0] = nullptr; r->rec_valid[0] = false;   // side[0] is about to describe this render, whose planes stay private
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
  SMESH_TRY(smesh_aggregator_join_exchange(a));   // (rows still being exchanged on the exchange stream: smesh_allreduce_rows)
  const uint64_t W = cam->width, H = cam->height, N = W * H;
  // One view: everything on the main stream (the rasteriser of view k + 1 on a second stream beside the fusion of view k was an
  // opt-in until round 5 -- SMESH_FUSE_PIPELINE, +0.6 % -- and is gone: batches overlap through smesh_fuse_views' group pipeline).
  const int slot = 0;
  SMESH_TRY(main_after_raster(r));          // (a pipelined smesh_fuse_views call may still be rasterising into the view slots)
  r->main_pending = true;                   // the next pipelined group must wait for what this call does to slot 0 on the main stream
  if (r->fused[slot].bytes < N * 8) {
    // growing a slot frees the old buffer: nothing may still be reading it
    SMESH_HIP(hipStreamSynchronize(ctx->raster_stream));
    SMESH_HIP(hipStreamSynchronize(ctx->stream));
    SMESH_TRY(r->fused[slot].reserve(N * 8));
  }
  uint32_t* d_idx = static_cast<uint32_t*>(r->fused[slot].ptr);
  r->last_idx[slot] = nullptr; r->rec_valid[slot] = false;   // the records of a render_device() on this side are being overwritten
  // (the fusion only consumes the index plane -- and the triangle-order kernels not even that, where the view has no queued triangles)
  const int tri_path = plane_optional_level(r, a);
  SMESH_TRY(render_into(r, cam, d_idx, /*d_depth=*/nullptr, ctx->stream, slot, tri_path));
  SMESH_TRY(fuse_rendered(r, a, slot, d_idx, probs, weights, memkind, W, H));
  r->fused_seq++;
  return SMESH_OK;
}

// A batch of views: smesh_fuse_view for each of them, in order -- except that, with device-resident class vectors, up to
// kMaxGroup views share each rasteriser launch, and two consecutive views of a triangle renderer are fused by ONE launch
// (k_fuse_tri<.., 2>: every 64-row accumulator block makes one round trip for both views; same additions in the same order
// as two calls).  Asynchronous like smesh_fuse_view.
int smesh_fuse_views(smesh_renderer_t* r, smesh_aggregator_t* a, const smesh_camera_t* cams, uint64_t n,
                     const float* const* probs, const float* const* weights, int memkind) {
  if (!r || !a || (n && (!cams || !probs))) return fail(SMESH_ERR_INVALID, "NULL argument");
  for (uint64_t i = 0; i < n; i++) {
    SMESH_TRY(check_camera(&cams[i]));
    if (!probs[i]) return fail(SMESH_ERR_INVALID, "NULL probs image");
  }
  DeviceCtx* ctx = r->ctx;
  if (smesh_aggregator_ctx(a) != ctx) return fail(SMESH_ERR_INVALID, "renderer and aggregator live on different devices");
  static const bool pairs_off = getenv("SMESH_FUSE_PAIRS") && atoi(getenv("SMESH_FUSE_PAIRS")) == 0;
  bool pairable;
  int tri_path;
  {
    std::lock_guard<std::mutex> g2(smesh_aggregator_mutex(a));
    tri_path = plane_optional_level(r, a);   // (fuse_rendered / the group's fusion take k_fuse_tri*: RasterArgs::idx_optional)
    pairable = !pairs_off && memkind == SMESH_MEM_DEVICE && !r->texels && r->F != 0 && smesh_aggregator_can_fuse_triangles(a, r->F) &&
               smesh_aggregator_can_fuse_pair(a);
  }
  static const bool raster_pairs_off = getenv("SMESH_RASTER_PAIRS") && atoi(getenv("SMESH_RASTER_PAIRS")) == 0;
  static const int group_max = getenv("SMESH_RASTER_GROUP") ? std::min(kMaxGroup, std::max(2, atoi(getenv("SMESH_RASTER_GROUP")))) : kMaxGroup;
  uint64_t i = 0;
  while (i < n) {
    // Rasterise a GROUP of the remaining views with one launch per stage (any renderer, any aggregator, device images), then
    // fuse them two by two where the aggregator has the two-view kernel, else one by one.
    int gn = (int)std::min<uint64_t>((uint64_t)group_max, n - i);
    bool grouped = !raster_pairs_off && memkind == SMESH_MEM_DEVICE && r->F != 0 && r->V != 0 && gn >= 2;
    for (int v = 0; v < gn && grouped; v++) grouped = queues_fit_group(cams[i + v].width, cams[i + v].height);
    if ((!pairable && !grouped) || i + 1 >= n) {
      SMESH_TRY(smesh_fuse_view(r, a, &cams[i], probs[i], weights ? weights[i] : nullptr, memkind));
      i += 1;
      continue;
    }
    std::lock_guard<std::mutex> g(r->mu);
    std::lock_guard<std::mutex> g2(smesh_aggregator_mutex(a));
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    SMESH_HIP(hipSetDevice(ctx->device));
    SMESH_TRY(smesh_aggregator_join_exchange(a));   // (rows still being exchanged on the exchange stream: smesh_allreduce_rows)
    const bool group_pipeline_on = opt_group_pipeline();     // (default on since round 5; SMESH_GROUP_PIPELINE=0 / smesh_set_option)
    const bool use_pipeline = grouped && group_pipeline_on;
    if (!use_pipeline) {
      SMESH_TRY(main_after_raster(r));
      r->main_pending = true;   // the renderer's scratch is in use on the main stream
    }
    // Group pipeline (SMESH_GROUP_PIPELINE=1; off by default): the rasteriser launches of this group go to the raster stream and
    // into the view-slot bank the previous group is NOT using, so they run beside the previous group's fusion launches; two
    // events per group hand the bank over and back.  Measured on cfg2: 12 950-13 010 views/s against 12 660-12 760 without
    // (+2 %): k_fuse_tri stretches from 83 to 95 us per pair launch and the rasteriser stage from 35 to ~70 us per view while they
    // share the CUs -- both are bound by how many of their waves a CU holds (96 / 126 VGPRs), not by two different units.
    const bool group_pipeline = group_pipeline_on;
    int base = 0;
    if (use_pipeline) {
      const int bank = (int)(r->group_seq++ & 1u);
      base = bank * kMaxGroup;
      for (int b = 0; b < 2; b++) {
        if (!r->ev_bank_rendered[b]) SMESH_HIP(hipEventCreateWithFlags(&r->ev_bank_rendered[b], hipEventDisableTiming));
        if (!r->ev_bank_consumed[b]) SMESH_HIP(hipEventCreateWithFlags(&r->ev_bank_consumed[b], hipEventDisableTiming));
      }
      // the bank's previous tenant must have been fused ...
      if (r->bank_used[bank]) SMESH_HIP(hipStreamWaitEvent(ctx->raster_stream, r->ev_bank_consumed[bank], 0));
      // ... and whatever the main stream did to the renderer's state OUTSIDE this pipeline (plain renders, fuse_view, slot growth)
      // comes first -- a full fence, paid once after such a call, not per group
      if (r->main_pending || !r->bank_used[bank]) {
        if (!r->ev_main_fence) SMESH_HIP(hipEventCreateWithFlags(&r->ev_main_fence, hipEventDisableTiming));
        SMESH_HIP(hipEventRecord(r->ev_main_fence, ctx->stream));
        SMESH_HIP(hipStreamWaitEvent(ctx->raster_stream, r->ev_main_fence, 0));
        r->main_pending = false;
      }
      if (!r->bank_used[bank ^ 1]) SMESH_TRY(prepare_group_slots(r, &cams[i], gn, ctx->raster_stream, (bank ^ 1) * kMaxGroup));
      SMESH_TRY(render_group_into(r, &cams[i], gn, ctx->raster_stream, base, -1, tri_path));
      SMESH_HIP(hipEventRecord(r->ev_bank_rendered[bank], ctx->raster_stream));
      SMESH_HIP(hipStreamWaitEvent(ctx->stream, r->ev_bank_rendered[bank], 0));
      r->bank_used[bank] = true;
      r->raster_pending = true;
    } else if (grouped) {
      SMESH_TRY(render_group_into(r, &cams[i], gn, ctx->stream, 0, -1, tri_path));
    } else {   // images too large for kMaxGroup sets of fragment queues, direct rasteriser, or SMESH_RASTER_PAIRS=0: one view at a time
      gn = 2;
      SMESH_HIP(alloc_side(r, 1));
      for (int v = 0; v < 2; v++) {
        const uint64_t N = (uint64_t)cams[i + v].width * cams[i + v].height;
        if (r->fused[v].bytes < N * 8) {
          SMESH_HIP(hipStreamSynchronize(ctx->stream));   // growing a slot frees the old buffer: nothing may still be reading it
          SMESH_TRY(r->fused[v].reserve(N * 8));
        }
        r->last_idx[v] = nullptr; r->rec_valid[v] = false;   // the records of a render_device() on this side are being overwritten
        SMESH_TRY(render_into(r, &cams[i + v], static_cast<uint32_t*>(r->fused[v].ptr), /*d_depth=*/nullptr, ctx->stream, v));   // (fused as a PAIR below: each needs the other's plane decision -- always written)
      }
    }
    // texel renderers: the views of the group in ONE fusion launch (a triangle's texel rows make one round trip for all of them)
    static const bool texel_multi_off = getenv("SMESH_TEXEL_MULTI") && atoi(getenv("SMESH_TEXEL_MULTI")) == 0;
    bool texel_multi = false;
    if (grouped && !pairable && r->texels && !texel_multi_off) texel_multi = smesh_aggregator_can_fuse_texels(a, r->num_primitives);
    if (texel_multi) {
      ProfScope fuse_region(ctx, SMESH_PROF_FUSE_SCATTER);
      RenderedView rv[kMaxGroup];
      for (int j = 0; j < gn; j++) {
        const smesh_renderer::Side& sd = r->side[base + j];
        const uint64_t k = i + (uint64_t)j;
        rv[j] = RenderedView{sd.frags, sd.big_queue, sd.big_count, static_cast<const uint32_t*>(r->fused[base + j].ptr), probs[k],
                             weights ? weights[k] : nullptr, cams[k].width, cams[k].height};
        rv[j].no_big = no_big_possible(r, &cams[k]);
        rv[j].kinds = sd.kinds;
      }
      SMESH_TRY(smesh_aggregator_fuse_texels_multi(a, r->F, r->tex_first, r->tex_res, r->big_capacity, rv, gn));
      smesh_note_fuse("k_fuse_texel", "render-records");
    }
    for (int j = 0; j < gn && !pairable && !texel_multi; j++) {   // class counts beyond k_fuse_tri, foreign primitive counts
      const uint64_t k = i + (uint64_t)j;
      SMESH_TRY(fuse_rendered(r, a, base + j, static_cast<const uint32_t*>(r->fused[base + j].ptr), probs[k], weights ? weights[k] : nullptr,
                              SMESH_MEM_DEVICE, cams[k].width, cams[k].height));
    }
    // the fusion launches of a group are back to back: one timed region for all of them (smesh_profile_*: a HIP event pair around a
    // single launch adds the dispatch latency that back-to-back launches hide)
    if (pairable) {
      ProfScope fuse_region(ctx, SMESH_PROF_FUSE_SCATTER);
      const int max_nv = smesh_aggregator_max_fused_views(a);
      for (int j = 0; j < gn;) {
        // the largest of 8 / 4 / 2 / 1 views that fits what is left of the group and what the kernel takes for this class count
        int nv = 1;
        while (nv * 2 <= std::min(max_nv, gn - j)) nv *= 2;
        RenderedView rv[kMaxGroup];
        for (int v = 0; v < nv; v++) {
          const smesh_renderer::Side& sd = r->side[base + j + v];
          const uint64_t k = i + (uint64_t)(j + v);
          rv[v] = RenderedView{sd.frags, sd.big_queue, sd.big_count, static_cast<const uint32_t*>(r->fused[base + j + v].ptr), probs[k],
                               weights ? weights[k] : nullptr, cams[k].width, cams[k].height, 0, 0, true};
          rv[v].no_big = no_big_possible(r, &cams[k]);
          rv[v].fine = box_extent_bound(r, &cams[k]) <= 48.0;      // (the bound is ~4 x the largest box: cfg2 13 - 33, boxes under 8 pixels; a 250 000-triangle mesh at 1080p 27 - 66, boxes of ~12)
        }
        SMESH_TRY(smesh_aggregator_fuse_triangles(a, r->F, r->prim_id, r->big_capacity, rv, nv));
        j += nv;
      }
    }
    if (pairable) smesh_note_fuse(smesh_aggregator_fuse_kernel_name(a, r->prim_id != nullptr), "render-records");
    if (grouped && group_pipeline) SMESH_HIP(hipEventRecord(r->ev_bank_consumed[base / kMaxGroup], ctx->stream));
    r->fused_seq += (uint64_t)gn;
    i += (uint64_t)gn;
  }
  return SMESH_OK;
}

}  // extern "C"

// ---- sharded jobs: a rank's last views fused by triangle range (smesh.h) -----------------------------------
// One launch per (group of up to eight held views, part): the records of all held views stay in side[kSlots ...], their index
// planes in fused[kSlots ...]; part p reads them again.  Per accumulator row the additions are those of smesh_fuse_views, in
// its order (rows of queued medium triangles excepted: their float atomics all go with part 0).
static int fuse_held_part(smesh_renderer* r, smesh_aggregator* a, int part) {
  DeviceCtx* ctx = r->ctx;
  const smesh_renderer::HeldJob& h = r->held;
  const int max_nv = smesh_aggregator_max_fused_views(a);
  static const int group_max = getenv("SMESH_RASTER_GROUP") ? std::min(kMaxGroup, std::max(2, atoi(getenv("SMESH_RASTER_GROUP")))) : kMaxGroup;
  for (int i = 0; i < h.n; i += group_max) {
    const int gn = std::min(group_max, h.n - i);
    ProfScope fuse_region(ctx, SMESH_PROF_FUSE_SCATTER);
    for (int j = 0; j < gn;) {
      int nv = 1;
      while (nv * 2 <= std::min(max_nv, gn - j)) nv *= 2;
      RenderedView rv[kMaxGroup];
      for (int v = 0; v < nv; v++) {
        const int k = i + j + v;
        const smesh_renderer::Side& sd = r->side[kSlots + k];
        rv[v] = RenderedView{sd.frags, sd.big_queue, sd.big_count, static_cast<const uint32_t*>(r->fused[kSlots + k].ptr), h.probs[k],
                             h.weights[k], h.W[k], h.H[k], 0, 0, true};
      }
      SMESH_TRY(smesh_aggregator_fuse_triangles(a, r->F, nullptr, r->big_capacity, rv, nv, part, h.nparts));
      j += nv;
    }
  }
  return SMESH_OK;
}

extern "C" {

int smesh_fuse_views_begin(smesh_renderer_t* r, smesh_aggregator_t* a, const smesh_camera_t* cams, uint64_t n,
                           const float* const* probs, const float* const* weights, int memkind, int nparts,
                           uint64_t* row_lo, uint64_t* row_hi) {
  if (!r || !a || !row_lo || !row_hi || (n && (!cams || !probs))) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (nparts < 1 || nparts > 64) return fail(SMESH_ERR_INVALID, "nparts must be in [1, 64]");
  for (uint64_t i = 0; i < n; i++) {
    SMESH_TRY(check_camera(&cams[i]));
    if (!probs[i]) return fail(SMESH_ERR_INVALID, "NULL probs image");
  }
  DeviceCtx* ctx = r->ctx;
  if (smesh_aggregator_ctx(a) != ctx) return fail(SMESH_ERR_INVALID, "renderer and aggregator live on different devices");
  const uint64_t P = smesh_aggregator_primitives(a);
  static const bool ranges_off = getenv("SMESH_FUSE_RANGES") && atoi(getenv("SMESH_FUSE_RANGES")) == 0;
  bool ranged;
  {
    std::lock_guard<std::mutex> g2(smesh_aggregator_mutex(a));
    ranged = !ranges_off && nparts > 1 && n >= 1 && n <= (uint64_t)kHeld && memkind == SMESH_MEM_DEVICE && !r->texels && r->F != 0 &&
             r->V != 0 && r->prim_id == nullptr && smesh_aggregator_can_fuse_triangles(a, r->F);
  }
  for (uint64_t i = 0; i < n && ranged; i++) ranged = queues_fit_group(cams[i].width, cams[i].height);
  if (!ranged) {
    // Texel renderers, re-ordered meshes (rows are not in triangle order), host images, more views than can be held: the whole job
    // goes with part 0, whose rows are then all of them; the later parts are empty.
    SMESH_TRY(smesh_fuse_views(r, a, cams, n, probs, weights, memkind));
    std::lock_guard<std::mutex> g(r->mu);
    r->held = smesh_renderer::HeldJob();
    r->held.nparts = nparts; r->held.next_part = 1; r->held.agg = a; r->held.ranged = false;
    *row_lo = 0; *row_hi = P;
    return SMESH_OK;
  }
  std::lock_guard<std::mutex> g(r->mu);
  std::lock_guard<std::mutex> g2(smesh_aggregator_mutex(a));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  SMESH_TRY(smesh_aggregator_join_exchange(a));   // (an exchange of an earlier job still in flight on the exchange stream)
  SMESH_TRY(main_after_raster(r));
  r->main_pending = true;
  smesh_renderer::HeldJob& h = r->held;
  h = smesh_renderer::HeldJob();
  h.n = (int)n; h.nparts = nparts; h.next_part = 1; h.agg = a; h.ranged = true;
  for (uint64_t i = 0; i < n; i++) {
    h.probs[i] = probs[i]; h.weights[i] = weights ? weights[i] : nullptr; h.W[i] = cams[i].width; h.H[i] = cams[i].height;
  }
  static const int group_max = getenv("SMESH_RASTER_GROUP") ? std::min(kMaxGroup, std::max(2, atoi(getenv("SMESH_RASTER_GROUP")))) : kMaxGroup;
  for (int i = 0; i < h.n; i += group_max)
    SMESH_TRY(render_group_into(r, &cams[i], std::min(group_max, h.n - i), ctx->stream, 0, kSlots + i, /*idx_optional: ranged => k_fuse_tri* */ plane_optional_level(r, a)));
  SMESH_TRY(fuse_held_part(r, a, 0));
  smesh_note_fuse(smesh_aggregator_fuse_kernel_name(a, false), "render-records");
  r->fused_seq += n;
  smesh_fuse_part_rows(r->F, 0, nparts, row_lo, row_hi);
  if (nparts == 1) h = smesh_renderer::HeldJob();
  return SMESH_OK;
}

int smesh_fuse_views_continue(smesh_renderer_t* r, smesh_aggregator_t* a, int part, uint64_t* row_lo, uint64_t* row_hi) {
  if (!r || !a || !row_lo || !row_hi) return fail(SMESH_ERR_INVALID, "NULL argument");
  DeviceCtx* ctx = r->ctx;
  if (smesh_aggregator_ctx(a) != ctx) return fail(SMESH_ERR_INVALID, "renderer and aggregator live on different devices");
  std::lock_guard<std::mutex> g(r->mu);
  std::lock_guard<std::mutex> g2(smesh_aggregator_mutex(a));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  smesh_renderer::HeldJob& h = r->held;
  if (h.agg != a || h.nparts < 2 || part != h.next_part || part >= h.nparts)
    return fail(SMESH_ERR_INVALID, "smesh_fuse_views_continue: no such part of a job begun with smesh_fuse_views_begin (parts go in order)");
  SMESH_HIP(hipSetDevice(ctx->device));
  if (h.ranged) {
    SMESH_TRY(fuse_held_part(r, a, part));
    smesh_fuse_part_rows(r->F, part, h.nparts, row_lo, row_hi);
  } else {
    *row_lo = *row_hi = smesh_aggregator_primitives(a);
  }
  if (++h.next_part == h.nparts) h = smesh_renderer::HeldJob();   // the job is over: its records may be overwritten
  return SMESH_OK;
}

}  // extern "C"

extern "C" {
// add() for an index image that is the unmodified device output of `r`'s most recent smesh_renderer_render_device():
// the reference's two-call convention (render, then add: colorize_cityscapes_mesh.py:65-67) at the speed of
// smesh_fuse_view.  Anything else is forwarded to smesh_aggregator_add.
int smesh_aggregator_add_rendered(smesh_aggregator_t* a, smesh_renderer_t* r, const uint32_t* idx_dev,
                                  const float* probs, const int64_t probs_strides[3], int probs_mem,
                                  const float* weights, const int64_t w_strides[2], int w_mem, uint64_t W, uint64_t H) {
  if (!a || !r || !idx_dev || !probs || !probs_strides) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (weights && !w_strides) return fail(SMESH_ERR_INVALID, "weights without strides");
  DeviceCtx* ctx = r->ctx;
  const int64_t C = (int64_t)smesh_aggregator_classes(a);
  const bool dense = probs_strides[0] == (int64_t)H * C && probs_strides[1] == C && probs_strides[2] == 1;
  // class vectors in device memory with other strides (the (H,W,C) tensor of a network seen as (W,H,C): what a framework's
  // transpose / permute returns) are gathered into the aggregator's scratch first and take the same path
  const bool strided_dev = !dense && probs_mem == SMESH_MEM_DEVICE && probs_strides[0] >= 0 && probs_strides[1] >= 0 && probs_strides[2] >= 0;
  bool fast = smesh_aggregator_ctx(a) == ctx && W != 0 && H != 0 && (dense || strided_dev) &&
              (!weights || (w_mem == probs_mem && w_strides[0] == (int64_t)H && w_strides[1] == 1));
  if (fast) {
    std::lock_guard<std::mutex> g(r->mu);
    std::lock_guard<std::mutex> g2(smesh_aggregator_mutex(a));
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    SMESH_TRY(smesh_aggregator_join_exchange(a));   // (rows still being exchanged on the exchange stream: smesh_allreduce_rows)
    int side = -1;
    for (int sd = 0; sd < kRecordSides; sd++)
      if (idx_dev == r->last_idx[sd] && W == r->last_W[sd] && H == r->last_H[sd]) side = sd;
    fast = side >= 0 && ((!r->texels && smesh_aggregator_can_fuse_triangles(a, r->F)) ||
                         (r->texels && smesh_aggregator_can_fuse_texels(a, r->num_primitives)));
    if (fast) {
      SMESH_HIP(hipSetDevice(ctx->device));
      // Asynchronous for DEVICE probs / weights, like smesh_fuse_view (host images were waited for inside fuse_rendered): a
      // host synchronisation per view would serialise the reference's two-call loop on launch latencies (0.10 -> 0.21 ms per
      // cfg2 view).  Callers order the buffers' next use after the library with smesh_stream_release (no host wait) or
      // smesh_synchronize; the Python layer does the former for every array that is not the library's own.
      // strided class vectors (the (H,W,C) output of a network seen as (W,H,C), colorize_cityscapes_mesh.py:65-67): k_fuse_tri
      // reads them where they are; the wide-row and texel kernels get a gathered copy (315 MB more traffic per cfg2-sized view)
      if (strided_dev && !r->texels && probs_strides[2] == 1 && !(reinterpret_cast<uintptr_t>(probs) & 3) &&
          smesh_aggregator_takes_strided_probs(a, probs_strides[0], probs_strides[1], 1))
        return fuse_rendered(r, a, side, idx_dev, probs, weights, probs_mem, W, H, probs_strides[0], probs_strides[1]);
      if (strided_dev) SMESH_TRY(smesh_aggregator_dense_probs(a, probs, probs_strides, W, H, &probs));
      return fuse_rendered(r, a, side, idx_dev, probs, weights, probs_mem, W, H);
    }
  }
  const int64_t is[2] = {(int64_t)H, 1};
  return smesh_aggregator_add(a, idx_dev, SMESH_IDX_U32, is, SMESH_MEM_DEVICE, probs, probs_strides, probs_mem, weights, w_strides,
                              w_mem, W, H);
}

// add() for an index image that has lost its identity but not its content (DLPack -> TF -> numpy -> add in the reference's harness,
// eval-scannet/eval_scannet.py:211-238): if the image is a dense (W,H) uint32 / int32 array whose 64-bit content checksum equals that
// of a plane one of r's last kRecordSides smesh_renderer_render_device() calls produced, and that render's per-triangle records are
// still there, the view is fused in triangle order -- reading the GIVEN image -- and *matched = 1.  Otherwise nothing happens and
// *matched = 0: the caller falls back to smesh_aggregator_add.  One pass over the image + one 56-byte read-back.
int smesh_aggregator_add_matched(smesh_aggregator_t* a, smesh_renderer_t* r,
                                 const void* indices, int idx_dtype, const int64_t idx_strides[2], int idx_mem,
                                 const float* probs, const int64_t probs_strides[3], int probs_mem,
                                 const float* weights, const int64_t w_strides[2], int w_mem, uint64_t W, uint64_t H, int* matched) {
  if (!a || !r || !indices || !idx_strides || !probs || !probs_strides || !matched) return fail(SMESH_ERR_INVALID, "NULL argument");
  *matched = 0;
  smesh_note_fuse("none", "none");   // (reporting: unless a render matches below, the caller goes on to smesh_aggregator_add, which reports itself)
  if (weights && !w_strides) return fail(SMESH_ERR_INVALID, "weights without strides");
  DeviceCtx* ctx = r->ctx;
  const int64_t C = (int64_t)smesh_aggregator_classes(a);
  if ((idx_dtype != SMESH_IDX_U32 && idx_dtype != SMESH_IDX_I32) || idx_strides[0] != (int64_t)H || idx_strides[1] != 1 || W == 0 || H == 0 ||
      smesh_aggregator_ctx(a) != ctx || (weights && (w_mem != probs_mem || w_strides[0] != (int64_t)H || w_strides[1] != 1)))
    return SMESH_OK;   // only dense index / weight images can take the triangle-order kernels
  const bool dense_probs = probs_strides[0] == (int64_t)H * C && probs_strides[1] == C && probs_strides[2] == 1;
  // ... and dense class vectors, or DEVICE ones that k_fuse_tri can read in place at their strides
  const bool strided_ok = !dense_probs && probs_mem == SMESH_MEM_DEVICE && !r->texels && probs_strides[2] == 1 &&
                          !(reinterpret_cast<uintptr_t>(probs) & 3) && smesh_aggregator_takes_strided_probs(a, probs_strides[0], probs_strides[1], 1);
  if (!dense_probs && !strided_ok) return SMESH_OK;
  std::lock_guard<std::mutex> g(r->mu);
  std::lock_guard<std::mutex> g2(smesh_aggregator_mutex(a));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_TRY(smesh_aggregator_join_exchange(a));   // (rows still being exchanged on the exchange stream: smesh_allreduce_rows)
  if (!((!r->texels && smesh_aggregator_can_fuse_triangles(a, r->F)) || (r->texels && smesh_aggregator_can_fuse_texels(a, r->num_primitives))))
    return SMESH_OK;
  bool any = false;
  for (int sd = 0; sd < kRecordSides; sd++) any = any || (r->rec_valid[sd] && r->hash_valid[sd] && r->last_W[sd] == W && r->last_H[sd] == H);
  if (!any || !r->d_hash) return SMESH_OK;
  SMESH_HIP(hipSetDevice(ctx->device));
  const uint64_t N = W * H;
  const uint32_t* d_img = static_cast<const uint32_t*>(indices);
  if (idx_mem == SMESH_MEM_HOST) {
    SMESH_TRY(r->match_stage.reserve(N * 4));
    SMESH_HIP(hipMemcpyAsync(r->match_stage.ptr, indices, N * 4, hipMemcpyHostToDevice, ctx->stream));
    d_img = static_cast<const uint32_t*>(r->match_stage.ptr);
  }
  // The look-up must not drain the main stream (the previous views' fusion kernels are still queued there: waiting for them made an
  // exported-and-returned plane cost 0.34 ms per cfg2 view).  An image in DEVICE memory is checksummed on the raster stream, which
  // the caller has ordered behind the image's producer like the main stream (smesh_stream_wait) and which only has to wait for the
  // kernels that sealed the candidate planes (hash_ev); the host then waits for that stream alone.  HOST images are staged on the
  // main stream (the staging buffer is read again by the fusion).
  hipStream_t ps = ctx->stream;
  if (idx_mem != SMESH_MEM_HOST) {
    ps = ctx->raster_stream;
    for (int sd = 0; sd < kRecordSides; sd++)
      if (r->hash_valid[sd] && r->hash_ev[sd]) SMESH_HIP(hipStreamWaitEvent(ps, r->hash_ev[sd], 0));
  }
  SMESH_TRY(plane_checksum(ctx, ps, d_img, N, r->d_hash + kRecordSides));
  unsigned long long h[kRecordSides + 1];
  SMESH_HIP(hipMemcpyAsync(h, r->d_hash, sizeof h, hipMemcpyDeviceToHost, ps));
  SMESH_HIP(hipStreamSynchronize(ps));
  int side = -1;
  for (int sd = 0; sd < kRecordSides; sd++)
    if (r->rec_valid[sd] && r->hash_valid[sd] && r->last_W[sd] == W && r->last_H[sd] == H && h[sd] == h[kRecordSides]) side = sd;
  if (side < 0) return SMESH_OK;
  SMESH_TRY(fuse_rendered(r, a, side, d_img, probs, weights, probs_mem, W, H, dense_probs ? 0 : probs_strides[0],
                          dense_probs ? 0 : probs_strides[1]));   // asynchronous for DEVICE images, see smesh_aggregator_add_rendered
  *matched = 1;
  return SMESH_OK;
}

// Computes (once) the content checksum of the plane `indices_dev` that one of r's last smesh_renderer_render_device() calls returned,
// which makes copies of it recognisable by smesh_aggregator_add_matched.  Call it when the plane's content first leaves the library's
// hands -- before exporting the pointer to another framework or copying the image to the host -- i.e. while it is still what the
// rasteriser wrote.  Not needed for (and not paid by) the plain render -> add loop.
int smesh_renderer_seal_render(smesh_renderer_t* r, const uint32_t* indices_dev) {
  if (!r || !indices_dev) return fail(SMESH_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(r->mu);
  DeviceCtx* ctx = r->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  for (int sd = 0; sd < kRecordSides; sd++) {
    if (r->last_idx[sd] != indices_dev || !r->rec_valid[sd] || r->hash_valid[sd]) continue;
    SMESH_HIP(hipSetDevice(ctx->device));
    if (!r->d_hash) {
      SMESH_HIP(dev_malloc(reinterpret_cast<void**>(&r->d_hash), (kRecordSides + 1) * sizeof(unsigned long long)));
      SMESH_HIP(hipMemsetAsync(r->d_hash, 0, (kRecordSides + 1) * sizeof(unsigned long long), ctx->stream));
    }
    SMESH_TRY(plane_checksum(ctx, ctx->stream, indices_dev, r->last_W[sd] * r->last_H[sd], r->d_hash + sd));
    if (!r->hash_ev[sd]) SMESH_HIP(hipEventCreateWithFlags(&r->hash_ev[sd], hipEventDisableTiming));
    SMESH_HIP(hipEventRecord(r->hash_ev[sd], ctx->stream));
    r->hash_valid[sd] = true;
  }
  return SMESH_OK;
}

int smesh_box_extent_bound(const float* vertices, uint64_t V, const int32_t* faces, uint64_t F, const smesh_camera_t* cam, double* bound) {
  if (!cam || !bound || (V && !vertices) || (F && !faces)) return fail(SMESH_ERR_INVALID, "NULL argument");
  smesh_renderer::Bounds b;
  mesh_bounds(vertices, V, faces, F, b);
  *bound = box_extent_bound(b, cam);
  return SMESH_OK;
}

int smesh_renderer_render_stats(smesh_renderer_t* r, const smesh_camera_t* cam, int* huge_stage_needed, uint32_t queue_lengths[4]) {
  if (!r || !cam) return fail(SMESH_ERR_INVALID, "NULL argument");
  SMESH_TRY(check_camera(cam));
  std::lock_guard<std::mutex> g(r->mu);
  if (huge_stage_needed) *huge_stage_needed = (no_huge_possible(r, cam) ? 0 : 1) | (no_big_possible(r, cam) ? 0 : 2);
  if (queue_lengths) {
    DeviceCtx* ctx = r->ctx;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    SMESH_HIP(hipSetDevice(ctx->device));
    SMESH_HIP(hipStreamSynchronize(ctx->raster_stream));
    SMESH_HIP(hipMemcpyAsync(queue_lengths, r->side[r->last_render_side].big_count, 16, hipMemcpyDeviceToHost, ctx->stream));
    SMESH_HIP(hipStreamSynchronize(ctx->stream));
  }
  return SMESH_OK;
}

}  // extern "C"
