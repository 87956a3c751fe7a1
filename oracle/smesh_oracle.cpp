// smesh_oracle.cpp -- CPU restatement of the semantic-meshes project-and-fuse hot path.
//
// *** TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT. ***
// Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library.
// The product (semantic_meshes_amd/) never links, imports or falls back to anything in oracle/.
//
// PARITY STATUS: "parity unpinned" for the rasteriser.  The reference has no tests, golden vectors
// or fixtures, cannot be compiled here (needs nvcc + the absent template-tensors,
// tinyply, dlpack submodules + Boost.Python), and the rasteriser arithmetic lives
// in the un-vendored, un-pinned submodule extern/template-tensors (.gitmodules:7-9).  The fusion
// arithmetic (Sum/Summax) IS fully determined by in-tree reference lines and is restated 1:1 below.
// Behaviours the reference leaves to template-tensors are decided in DESIGN.md ("Raster spec") and
// implemented here and, independently, in semantic_meshes_amd/csrc/raster.hip.
//
// Citations are relative to /root/reference.
//
// Build: see oracle/Makefile (g++ -O2 -fopenmp -ffp-contract=off).  -ffp-contract=off matters: the
// GPU kernels must execute the same IEEE-754 operation sequence to be bit-exact on indices.

#include "../include/smesh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include <omp.h>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

int g_threads = 1;            // deterministic by default; bench raises it for the CPU baseline
bool g_accum_double = false;  // float64 accumulators for tolerance tests (reference uses float32)
bool g_fast_histogram = false;  // "optimised CPU" baseline variant: dense parallel histogram instead of the reference's serial std::map
// Independent yardsticks for the two places where the spec was tuned together with the kernels: kept so that the
// spec'd forms are compared against a restatement that did NOT move with the implementation (tests/test_oracle.py,
// tests/test_gpu_parity.py state the tolerances and the known divergences; INTEGRATION.md lists them).
bool g_mul_literal = false;     // Mul term as the literal reading of Fusion.cu:83-87: logf(powf(p, w)), p^w rounded to float32 first
bool g_edge_five_op = false;    // edge function in its round-1 form sign * (dx * (py - ly) - dy * (px - lx)) (five operations)

// ---------------------------------------------------------------------------------------------
// Raster spec (DESIGN.md "Raster spec"); protocol from TriangleRenderer.h:30-39,46-61,63-89.
// ---------------------------------------------------------------------------------------------
constexpr float kNear = 1e-6f;  // near plane z_c = kNear: triangles crossing it are CLIPPED against it (round 3; B-3), see clip_triangle

struct ScreenVertex {
  double u, v;  // pixel coordinates: f * (Xc.xy / Xc.z) + c  (render/Camera.h:10-11 -- double intrinsics)
  double iz;    // 1 / z_c ; 0 marks an unusable vertex
};

inline ScreenVertex project_vertex(const smesh_camera_t& cam, const float* p) {
  const float* R = cam.rotation;
  const float* t = cam.translation;
  // Rigid<float,3>::transformPoint: float32 (render/Camera.h:13), fixed left-to-right order
  const float xc = ((R[0] * p[0] + R[1] * p[1]) + R[2] * p[2]) + t[0];
  const float yc = ((R[3] * p[0] + R[4] * p[1]) + R[5] * p[2]) + t[1];
  const float zc = ((R[6] * p[0] + R[7] * p[1]) + R[8] * p[2]) + t[2];
  ScreenVertex s;
  s.u = 0.0; s.v = 0.0; s.iz = 0.0;
  if (!(zc > kNear) || !std::isfinite(xc) || !std::isfinite(yc) || !std::isfinite(zc)) return s;
  const double zd = (double)zc;
  const double u = cam.focal[0] * ((double)xc / zd) + cam.principal[0];
  const double v = cam.focal[1] * ((double)yc / zd) + cam.principal[1];
  if (!std::isfinite(u) || !std::isfinite(v)) return s;
  s.u = u; s.v = v; s.iz = 1.0 / zd;
  return s;
}

// Edge function through two screen points as a linear form E(p) = A px + B py + C, its coefficients taken from the endpoints in
// a canonical order so that the two triangles sharing an edge compute bit-identical magnitudes (watertightness, B-2), evaluated
// with two fused multiply-adds -- fma(A, px, fma(B, py, C)): every step one IEEE rounding, the same on the host and on the GPU.
// (Round 1's form dx (py - ly) - dy (px - lx) cost five operations per edge and sample; the rasteriser is bound by them.)
struct Edge {
  double A, B, C;
  inline void setup(double ax, double ay, double bx, double by) {
    const bool sw = (bx < ax) || (bx == ax && by < ay);
    const double lx = sw ? bx : ax, ly = sw ? by : ay;
    const double hx = sw ? ax : bx, hy = sw ? ay : by;
    const double dx = hx - lx, dy = hy - ly;
    const double c0 = dx * ly;
    const double c = std::fma(dy, lx, -c0);          // dy lx - dx ly
    const double sign = sw ? -1.0 : 1.0;             // -1 if the endpoints were swapped into canonical order
    A = sign * (-dy); B = sign * dx; C = sign * c;   // (multiplications by +-1: exact)
  }
  // five-operation form (g_edge_five_op): the canonical endpoints themselves
  double lx5 = 0, ly5 = 0, dx5 = 0, dy5 = 0, sign5 = 1;
  inline void setup_any(double ax, double ay, double bx, double by) {
    setup(ax, ay, bx, by);
    const bool sw = (bx < ax) || (bx == ax && by < ay);
    lx5 = sw ? bx : ax; ly5 = sw ? by : ay;
    dx5 = (sw ? ax : bx) - lx5; dy5 = (sw ? ay : by) - ly5;
    sign5 = sw ? -1.0 : 1.0;
  }
  inline void flip() { A = -A; B = -B; C = -C; sign5 = -sign5; }
  inline double eval(double px, double py) const {
    if (g_edge_five_op) return sign5 * (dx5 * (py - ly5) - dy5 * (px - lx5));
    return std::fma(A, px, std::fma(B, py, C));
  }
};

struct TriSetup {
  Edge e[3];        // e[i] is the edge opposite vertex i
  double s;         // orientation sign so that interior weights are positive
  bool own[3];      // tie-break: does a sample exactly on edge i belong to this triangle?
  double iz[3];
  int x0, x1, y0, y1;
  bool ok;
};

inline TriSetup setup_triangle(const ScreenVertex& a, const ScreenVertex& b, const ScreenVertex& c,
                               uint64_t W, uint64_t H) {
  TriSetup t;
  t.ok = false;
  if (a.iz == 0.0 || b.iz == 0.0 || c.iz == 0.0) return t;
  const double minu = std::min(a.u, std::min(b.u, c.u)), maxu = std::max(a.u, std::max(b.u, c.u));
  const double minv = std::min(a.v, std::min(b.v, c.v)), maxv = std::max(a.v, std::max(b.v, c.v));
  // samples sit at pixel centres (x + 0.5, y + 0.5)  (B-1)
  double fx0 = std::ceil(minu - 0.5), fx1 = std::floor(maxu - 0.5);
  double fy0 = std::ceil(minv - 0.5), fy1 = std::floor(maxv - 0.5);
  if (fx0 < 0.0) fx0 = 0.0;
  if (fy0 < 0.0) fy0 = 0.0;
  if (fx1 > (double)(W - 1)) fx1 = (double)(W - 1);
  if (fy1 > (double)(H - 1)) fy1 = (double)(H - 1);
  if (!(fx0 <= fx1) || !(fy0 <= fy1)) return t;
  t.x0 = (int)fx0; t.x1 = (int)fx1; t.y0 = (int)fy0; t.y1 = (int)fy1;
  t.e[0].setup_any(b.u, b.v, c.u, c.v);
  t.e[1].setup_any(c.u, c.v, a.u, a.v);
  t.e[2].setup_any(a.u, a.v, b.u, b.v);
  const double area2 = t.e[2].eval(c.u, c.v);
  if (!(area2 != 0.0) || !std::isfinite(area2)) return t;  // degenerate; no back-face culling (a3)
  t.s = area2 > 0.0 ? 1.0 : -1.0;
  for (int i = 0; i < 3; i++) {
    if (t.s < 0.0) t.e[i].flip();                           // interior weights positive whatever the orientation
    t.own[i] = (t.e[i].A > 0.0) || (t.e[i].A == 0.0 && t.e[i].B > 0.0);
  }
  t.iz[0] = a.iz; t.iz[1] = b.iz; t.iz[2] = c.iz;
  t.ok = true;
  return t;
}

// Returns true and the weights if sample (px,py) is covered.
inline bool cover(const TriSetup& t, double px, double py, double w[3]) {
  for (int i = 0; i < 3; i++) {
    w[i] = t.e[i].eval(px, py);
    if (!(w[i] > 0.0 || (w[i] == 0.0 && t.own[i]))) return false;
  }
  return true;
}

// perspective-correct camera-space depth (B-3)
inline bool depth_at(const TriSetup& t, const double w[3], float* z) {
  const double num = (w[0] + w[1]) + w[2];
  const double den = std::fma(w[2], t.iz[2], std::fma(w[1], t.iz[1], w[0] * t.iz[0]));
  const float zf = (float)(num / den);
  if (!(zf > 0.0f) || !std::isfinite(zf)) return false;
  *z = zf;
  return true;
}

inline uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bits_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

constexpr uint64_t kBackgroundKey = ((uint64_t)0x7F800000u << 32) | 0xFFFFFFFFull;  // {+inf, -1}

inline void key_min(uint64_t* slot, uint64_t key, bool parallel) {
  if (!parallel) { if (key < *slot) *slot = key; return; }
  uint64_t cur = __atomic_load_n(slot, __ATOMIC_RELAXED);
  while (key < cur && !__atomic_compare_exchange_n(slot, &cur, key, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

// SymmetricMatrixLowerTriangleRowMajor::toIndex is out of tree (B-5); decided bijection:
// row = tu + tv, col = tu  ->  row*(row+1)/2 + col, with (tu,tv) clamped into the triangle.
inline uint32_t texel_index(uint32_t res, double b1, double b2) {
  // TexturedTriangleRenderer.h:34-38: uv = b1*(1,0) + b2*(0,1) in float; t = (int)((uv - 1e-6) * res)
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

// ---- near-plane clipping (round 3; DESIGN.md "Raster spec" 2) ---------------------------------------------------------------
// A vertex is FRONT when its float32 camera-space z is > kNear and everything about it is finite (project_vertex gives it screen
// coordinates), BEHIND when its camera-space coordinates are finite and z <= kNear, UNUSABLE otherwise.  A triangle with an unusable
// vertex, or with no front vertex, emits nothing.  A triangle with front AND behind vertices is cut along the plane z_c = kNear:
// on every edge from a front vertex F to a behind vertex B the point I = F + t (B - F), t = (zF - n) / (zF - zB), is computed in
// double FROM THE FRONT VERTEX (both triangles sharing the edge compute the same I bit for bit, so the cut edge F-I stays
// watertight), projected with z = n exactly.  One front vertex F (cyclic order F, N, P): the triangle (F, I_FN, I_FP).  Two front
// vertices (cyclic order B, N, P): the quadrilateral I_NB, N, P, I_PB as the two triangles (N, P, I_PB) and (N, I_PB, I_NB).
// The pieces are rasterised like any triangle (same edge functions, fill rule, depth formula with the pieces' own 1/z) and write the
// id of the triangle they came from.  Texel primitives: the pieces carry the original triangle's barycentric coordinates (b1, b2) at
// their vertices and interpolate them perspective-correctly (the screen-space weights of an unclipped triangle are not defined for
// a triangle with a vertex behind the camera).
struct CamVertex { float x, y, z; int status; };   // status: 0 front, 1 behind, 2 unusable
inline CamVertex camera_vertex(const smesh_camera_t& cam, const float* p, const ScreenVertex& s) {
  const float* R = cam.rotation;
  const float* t = cam.translation;
  CamVertex c;
  c.x = ((R[0] * p[0] + R[1] * p[1]) + R[2] * p[2]) + t[0];
  c.y = ((R[3] * p[0] + R[4] * p[1]) + R[5] * p[2]) + t[1];
  c.z = ((R[6] * p[0] + R[7] * p[1]) + R[8] * p[2]) + t[2];
  if (s.iz != 0.0) c.status = 0;
  else if (std::isfinite(c.x) && std::isfinite(c.y) && std::isfinite(c.z) && !(c.z > kNear)) c.status = 1;
  else c.status = 2;
  return c;
}

struct ClipVertex { ScreenVertex s; double b1, b2; };   // screen vertex + barycentric coordinates in the original triangle

// The point where the edge from front vertex F to behind vertex B meets the near plane.  Returns false if it does not project.
inline bool clip_edge(const smesh_camera_t& cam, const CamVertex& F, double fb1, double fb2, const CamVertex& B, double bb1, double bb2,
                      ClipVertex* out) {
  const double zn = (double)kNear;
  const double t = ((double)F.z - zn) / ((double)F.z - (double)B.z);
  const double x = std::fma(t, (double)B.x - (double)F.x, (double)F.x);
  const double y = std::fma(t, (double)B.y - (double)F.y, (double)F.y);
  const double u = cam.focal[0] * (x / zn) + cam.principal[0];
  const double v = cam.focal[1] * (y / zn) + cam.principal[1];
  if (!std::isfinite(u) || !std::isfinite(v)) return false;
  out->s.u = u; out->s.v = v; out->s.iz = 1.0 / zn;
  out->b1 = std::fma(t, bb1 - fb1, fb1);
  out->b2 = std::fma(t, bb2 - fb2, fb2);
  return true;
}

// Pieces of a triangle that crosses the near plane: 0, 1 or 2 triangles of ClipVertex.
inline int clip_triangle(const smesh_camera_t& cam, const CamVertex cv[3], const ScreenVertex sv[3], ClipVertex out[2][3]) {
  static const double B1[3] = {0.0, 1.0, 0.0}, B2[3] = {0.0, 0.0, 1.0};   // (b1, b2) of the original vertices a, b, c
  int nfront = 0;
  for (int k = 0; k < 3; k++) {
    if (cv[k].status == 2) return 0;
    nfront += cv[k].status == 0;
  }
  if (nfront == 0 || nfront == 3) return 0;
  auto orig = [&](int k) { ClipVertex c; c.s = sv[k]; c.b1 = B1[k]; c.b2 = B2[k]; return c; };
  if (nfront == 1) {
    int f = 0;
    while (cv[f].status != 0) f++;
    const int n = (f + 1) % 3, p = (f + 2) % 3;
    out[0][0] = orig(f);
    if (!clip_edge(cam, cv[f], B1[f], B2[f], cv[n], B1[n], B2[n], &out[0][1])) return 0;
    if (!clip_edge(cam, cv[f], B1[f], B2[f], cv[p], B1[p], B2[p], &out[0][2])) return 0;
    return 1;
  }
  int b = 0;
  while (cv[b].status != 1) b++;
  const int n = (b + 1) % 3, p = (b + 2) % 3;
  ClipVertex inb, ipb;
  if (!clip_edge(cam, cv[n], B1[n], B2[n], cv[b], B1[b], B2[b], &inb)) return 0;
  if (!clip_edge(cam, cv[p], B1[p], B2[p], cv[b], B1[b], B2[b], &ipb)) return 0;
  out[0][0] = orig(n); out[0][1] = orig(p); out[0][2] = ipb;
  out[1][0] = orig(n); out[1][1] = ipb;     out[1][2] = inb;
  return 2;
}

}  // namespace

struct smesh_renderer {
  std::vector<float> verts;
  std::vector<int32_t> faces;
  uint64_t V = 0, F = 0;
  bool texels = false;
  std::vector<uint32_t> tex_res, tex_first;
  uint64_t num_primitives = 0;
  std::vector<uint64_t> keys;
  std::vector<ScreenVertex> sv;
  std::vector<CamVertex> cv;
};

struct smesh_aggregator {
  uint64_t P = 0;
  uint32_t C = 0;
  int kind = 0;
  float iew = 0.5f;
  std::vector<float> acc;    // float32 state, as the reference (Fusion.cu:58,71,85)
  std::vector<double> accd;  // used instead when g_accum_double
  std::vector<std::mutex> locks;  // atomic::op::Lock<std::mutex> per primitive (Fusion.cu:58,71,85)
  smesh_aggregator(uint64_t P_, uint32_t C_) : P(P_), C(C_), locks(P_) {}
};

extern "C" {

const char* smesh_backend(void) { return "oracle-cpu"; }
const char* smesh_last_error(void) { return g_err.c_str(); }
int smesh_device_count(int* count) { if (count) *count = 0; return SMESH_OK; }
int smesh_synchronize(int) { return SMESH_OK; }

// oracle-only knobs (not part of smesh.h)
int smesh_oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; return SMESH_OK; }
int smesh_oracle_get_threads(void) { return g_threads; }
int smesh_oracle_set_accum_double(int on) { g_accum_double = on != 0; return SMESH_OK; }
int smesh_oracle_set_fast_histogram(int on) { g_fast_histogram = on != 0; return SMESH_OK; }
int smesh_oracle_set_mul_literal(int on) { g_mul_literal = on != 0; return SMESH_OK; }
int smesh_oracle_set_edge_five_op(int on) { g_edge_five_op = on != 0; return SMESH_OK; }

// --------------------------------------------------------------------------------------------
// renderer
// --------------------------------------------------------------------------------------------
static int make_renderer(const float* vertices, uint64_t V, const int32_t* faces, uint64_t F,
                         smesh_renderer_t** out) {
  if (!out) return fail(SMESH_ERR_INVALID, "out is NULL");
  if ((V && !vertices) || (F && !faces)) return fail(SMESH_ERR_INVALID, "NULL mesh arrays");
  auto* r = new (std::nothrow) smesh_renderer();
  if (!r) return fail(SMESH_ERR_RUNTIME, "out of memory");
  r->V = V; r->F = F;
  r->verts.assign(vertices, vertices + 3 * V);   // TriangleRenderer.h:36
  r->faces.assign(faces, faces + 3 * F);         // TriangleRenderer.h:37-38
  r->num_primitives = F;                         // TriangleRenderer.h:41-44
  *out = r;
  return SMESH_OK;
}

int smesh_renderer_create_triangles(const float* vertices, uint64_t V, const int32_t* faces, uint64_t F,
                                    int, smesh_renderer_t** out) {
  return make_renderer(vertices, V, faces, F, out);
}

// TexturedTriangleRenderer.h:87-182
int smesh_renderer_create_texels(const float* vertices, uint64_t V, const int32_t* faces, uint64_t F,
                                 const smesh_camera_t* cameras, uint64_t K, float tpp, int,
                                 smesh_renderer_t** out) {
  if (K && !cameras) return fail(SMESH_ERR_INVALID, "NULL cameras");
  int st = make_renderer(vertices, V, faces, F, out);
  if (st) return st;
  smesh_renderer* r = *out;
  r->texels = true;
  r->tex_res.assign(F, 0);
  r->tex_first.assign(F, 0);
  for (uint64_t f = 0; f < F; f++) {
    int32_t* face = &r->faces[3 * f];
    bool valid = true;
    for (int k = 0; k < 3; k++) if (face[k] < 0 || (uint64_t)face[k] >= V) valid = false;
    if (!valid) continue;
    auto vert = [&](int k) { return &r->verts[3 * (size_t)face[k % 3]]; };
    float best = 0.0f;  // aggregator::max<float>(0)  (:92)
    for (uint64_t ci = 0; ci < K; ci++) {
      const smesh_camera_t& cam = cameras[ci];
      float pu[3], pv[3];
      bool in_front = false;
      for (int k = 0; k < 3; k++) {
        const float* p = vert(k);
        const float* R = cam.rotation; const float* t = cam.translation;
        const float xc = ((R[0] * p[0] + R[1] * p[1]) + R[2] * p[2]) + t[0];
        const float yc = ((R[3] * p[0] + R[4] * p[1]) + R[5] * p[2]) + t[1];
        const float zc = ((R[6] * p[0] + R[7] * p[1]) + R[8] * p[2]) + t[2];
        in_front |= zc > 0.0f;  // :108
        // Project -> Vector2f (:73-82,109)
        pu[k] = (float)(cam.focal[0] * ((double)xc / (double)zc) + cam.principal[0]);
        pv[k] = (float)(cam.focal[1] * ((double)yc / (double)zc) + cam.principal[1]);
      }
      const float border = 0.5f;  // :114
      const float rw = (float)(int)cam.width, rh = (float)(int)cam.height;
      bool inside = in_front;
      for (int k = 0; k < 3 && inside; k++) {
        inside = (-border * rw <= pu[k]) && (pu[k] < (1 + border) * rw) &&
                 (-border * rh <= pv[k]) && (pv[k] < (1 + border) * rh);  // :115-119
      }
      if (inside) {
        const float area = 0.5f * std::fabs(pu[0] * (pv[1] - pv[2]) + pu[1] * (pv[2] - pv[0]) + pu[2] * (pv[0] - pv[1]));  // :121-123
        if (area > best) best = area;
      }
    }
    r->tex_res[f] = (uint32_t)std::ceil(tpp * std::sqrt(best));  // :127

    // "Optimally order face indices" (:129-146): vertex 0 gets the angle closest to 90 degrees
    float diffs[3];
    for (int k = 0; k < 3; k++) {
      const float* p0 = vert(k); const float* p1 = vert(k + 1); const float* p2 = vert(k + 2);
      const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
      const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
      const float dot = ax * bx + ay * by + az * bz;
      const float la = std::sqrt(ax * ax + ay * ay + az * az), lb = std::sqrt(bx * bx + by * by + bz * bz);
      float cosv = dot / (la * lb);
      if (cosv > 1.0f) cosv = 1.0f;
      if (cosv < -1.0f) cosv = -1.0f;
      const float angle = std::acos(cosv);
      diffs[k] = std::fabs(angle - 1.57079632679489661923f);
    }
    int best_k = 0;
    for (int k = 1; k < 3; k++) if (diffs[k] < diffs[best_k]) best_k = k;
    if (best_k != 0) { std::swap(face[0], face[best_k]); std::swap(diffs[0], diffs[best_k]); }
    if (diffs[1] >= diffs[2]) std::swap(face[1], face[2]);
  }
  // prefix sum (:149-162); getTexelNum = r(r+1)/2 (:43-47)
  uint64_t total = 0;
  for (uint64_t f = 0; f < F; f++) {
    r->tex_first[f] = (uint32_t)total;
    const uint64_t res = r->tex_res[f];
    total += res * (res + 1) / 2;
  }
  if (total >= 0xFFFFFFFFull) { delete r; *out = nullptr; return fail(SMESH_ERR_INVALID, "texel count overflows uint32"); }
  r->num_primitives = total;
  return SMESH_OK;
}

int smesh_renderer_destroy(smesh_renderer_t* r) { delete r; return SMESH_OK; }

int smesh_renderer_num_primitives(const smesh_renderer_t* r, uint64_t* out) {
  if (!r || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  *out = r->num_primitives;
  return SMESH_OK;
}

int smesh_renderer_texel_layout(const smesh_renderer_t* r, int32_t* faces_out, uint32_t* res_out, uint32_t* first_out) {
  if (!r) return fail(SMESH_ERR_INVALID, "NULL renderer");
  if (!r->texels) return fail(SMESH_ERR_INVALID, "not a texel renderer");
  if (faces_out) std::memcpy(faces_out, r->faces.data(), r->faces.size() * sizeof(int32_t));
  if (res_out) std::memcpy(res_out, r->tex_res.data(), r->tex_res.size() * sizeof(uint32_t));
  if (first_out) std::memcpy(first_out, r->tex_first.data(), r->tex_first.size() * sizeof(uint32_t));
  return SMESH_OK;
}

// Renderer<T>::render (Renderer.h:25-43) = clear (TriangleRenderer.h:75-78) + raster (:81-88) + split (Renderer.h:32-35)
int smesh_renderer_render(smesh_renderer_t* r, const smesh_camera_t* cam, uint32_t* idx_out, float* depth_out) {
  if (!r || !cam || !idx_out) return fail(SMESH_ERR_INVALID, "NULL argument");
  const uint64_t W = cam->width, H = cam->height;
  if (W == 0 || H == 0 || W > 65536 || H > 65536) return fail(SMESH_ERR_INVALID, "bad resolution");
  const uint64_t N = W * H;
  r->keys.assign(N, kBackgroundKey);  // {z=+inf, primitive_index=-1}
  r->sv.resize(r->V);
  r->cv.resize(r->V);
  const bool par = g_threads > 1;
#pragma omp parallel for num_threads(g_threads) schedule(static) if (par)
  for (int64_t v = 0; v < (int64_t)r->V; v++) {
    r->sv[v] = project_vertex(*cam, &r->verts[3 * v]);
    r->cv[v] = camera_vertex(*cam, &r->verts[3 * v], r->sv[v]);
  }
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 1024) if (par)
  for (int64_t f = 0; f < (int64_t)r->F; f++) {
    const int32_t* face = &r->faces[3 * f];
    if (face[0] < 0 || face[1] < 0 || face[2] < 0) continue;
    if ((uint64_t)face[0] >= r->V || (uint64_t)face[1] >= r->V || (uint64_t)face[2] >= r->V) continue;
    if (r->texels && r->tex_res[f] == 0) continue;  // a triangle without texels has no primitive to write
    // the triangle itself, or the one or two pieces left of it in front of the near plane
    ClipVertex piece[2][3];
    int npieces = 1;
    bool clipped = false;
    const ScreenVertex tsv[3] = {r->sv[face[0]], r->sv[face[1]], r->sv[face[2]]};
    if (tsv[0].iz == 0.0 || tsv[1].iz == 0.0 || tsv[2].iz == 0.0) {
      const CamVertex tcv[3] = {r->cv[face[0]], r->cv[face[1]], r->cv[face[2]]};
      npieces = clip_triangle(*cam, tcv, tsv, piece);
      clipped = true;
    } else {
      piece[0][0].s = tsv[0]; piece[0][1].s = tsv[1]; piece[0][2].s = tsv[2];
    }
    for (int pc = 0; pc < npieces; pc++) {
      const ClipVertex* q = piece[pc];
      const TriSetup t = setup_triangle(q[0].s, q[1].s, q[2].s, W, H);
      if (!t.ok) continue;
      for (int x = t.x0; x <= t.x1; x++) {
        for (int y = t.y0; y <= t.y1; y++) {
          double w[3];
          if (!cover(t, (double)x + 0.5, (double)y + 0.5, w)) continue;
          float z;
          if (!depth_at(t, w, &z)) continue;
          uint32_t prim = (uint32_t)f;  // Shader: primitive ordinal (TriangleRenderer.h:57-60)
          if (r->texels) {
            double b1, b2;
            if (!clipped) {
              const double num = (w[0] + w[1]) + w[2];
              b1 = w[1] / num; b2 = w[2] / num;                         // TexturedTriangleRenderer.h:193-196
            } else {
              // perspective-correct interpolation of the original triangle's (b1, b2) over the piece
              const double q0 = w[0] * t.iz[0], q1 = w[1] * t.iz[1], q2 = w[2] * t.iz[2];
              const double den = std::fma(w[2], t.iz[2], std::fma(w[1], t.iz[1], w[0] * t.iz[0]));
              b1 = std::fma(q2, q[2].b1, std::fma(q1, q[1].b1, q0 * q[0].b1)) / den;
              b2 = std::fma(q2, q[2].b2, std::fma(q1, q[1].b2, q0 * q[0].b2)) / den;
            }
            prim = r->tex_first[f] + texel_index(r->tex_res[f], b1, b2);
          }
          const uint64_t key = ((uint64_t)float_bits(z) << 32) | prim;
          key_min(&r->keys[(uint64_t)x * H + y], key, par);  // nearest z wins; ties -> lower id (B-4)
        }
      }
    }
  }
  for (uint64_t i = 0; i < N; i++) {
    idx_out[i] = (uint32_t)(r->keys[i] & 0xFFFFFFFFull);
    if (depth_out) depth_out[i] = bits_float((uint32_t)(r->keys[i] >> 32));
  }
  return SMESH_OK;
}

int smesh_renderer_render_device(smesh_renderer_t*, const smesh_camera_t*, uint32_t**, float**) {
  return fail(SMESH_ERR_NODEVICE, "oracle has no device memory");
}
int smesh_renderer_release_image(smesh_renderer_t*, void*, void*) { return SMESH_OK; }

// --------------------------------------------------------------------------------------------
// aggregator
// --------------------------------------------------------------------------------------------
static void agg_fill_zero(smesh_aggregator* a) {
  // Sum/Summax start at the 0-vector; Mul at LogProb of 1 == log-domain 0 (Mesh.h:57-63; A.2)
  std::fill(a->acc.begin(), a->acc.end(), 0.0f);
  std::fill(a->accd.begin(), a->accd.end(), 0.0);
}

int smesh_aggregator_create(uint64_t P, uint32_t C, int kind, float iew, int, smesh_aggregator_t** out) {
  if (!out) return fail(SMESH_ERR_INVALID, "out is NULL");
  if (C == 0) return fail(SMESH_ERR_INVALID, "classes must be > 0");
  if (kind < 0 || kind > 2) return fail(SMESH_ERR_INVALID, "unknown aggregator kind");
  if (P >= 0xFFFFFFFFull) return fail(SMESH_ERR_INVALID, "too many primitives for uint32 indices");
  auto* a = new (std::nothrow) smesh_aggregator(P, C);
  if (!a) return fail(SMESH_ERR_RUNTIME, "out of memory");
  a->kind = kind; a->iew = iew;
  if (g_accum_double) a->accd.assign(P * C, 0.0); else a->acc.assign(P * C, 0.0f);
  *out = a;
  return SMESH_OK;
}

int smesh_aggregator_destroy(smesh_aggregator_t* a) { delete a; return SMESH_OK; }

int smesh_aggregator_reset(smesh_aggregator_t* a) {  // Mesh.h:119-122
  if (!a) return fail(SMESH_ERR_INVALID, "NULL aggregator");
  agg_fill_zero(a);
  return SMESH_OK;
}

static inline uint32_t load_idx(const void* base, int dtype, int64_t off) {
  // FromPrimitiveImage accepts u32/i32/u64/i64 (Common.h:5-12); TensorConstructor casts to uint32 (Fusion.h:45)
  switch (dtype) {
    case SMESH_IDX_U32: return ((const uint32_t*)base)[off];
    case SMESH_IDX_I32: return (uint32_t)((const int32_t*)base)[off];
    case SMESH_IDX_U64: return (uint32_t)((const uint64_t*)base)[off];
    default:            return (uint32_t)((const int64_t*)base)[off];
  }
}

// Mul's logarithm is part of the spec (the reference's pow / log live in the absent template-tensors): a fixed
// sequence of IEEE float32 operations -- Cephes' logf polynomial with every multiply-add a correctly rounded fma -- that the HIP
// kernels execute identically (fuse_tri.inc.hpp, log_spec), so that both sides produce bit-identical terms.  Within 2 ulp of ln.
static inline float log_spec(float x) {
  if (!(x > 0.0f)) return x == 0.0f ? -std::numeric_limits<float>::infinity() : std::numeric_limits<float>::quiet_NaN();
  if (std::isinf(x)) return x;
  int e_adj = 0;
  uint32_t ix;
  std::memcpy(&ix, &x, 4);
  if (ix < 0x00800000u) { x = x * 8388608.0f; e_adj = -23; std::memcpy(&ix, &x, 4); }
  int e = (int)(ix >> 23) - 127 + e_adj;
  uint32_t im = (ix & 0x007FFFFFu) | 0x3F800000u;
  if ((ix & 0x007FFFFFu) > 0x003504F3u) { im -= 0x00800000u; e += 1; }
  float m;
  std::memcpy(&m, &im, 4);
  const float f = m - 1.0f;
  const float z = f * f;
  float y = 7.0376836292e-2f;
  y = std::fmaf(y, f, -1.1514610310e-1f);
  y = std::fmaf(y, f, 1.1676998740e-1f);
  y = std::fmaf(y, f, -1.2420140846e-1f);
  y = std::fmaf(y, f, 1.4249322787e-1f);
  y = std::fmaf(y, f, -1.6668057665e-1f);
  y = std::fmaf(y, f, 2.0000714765e-1f);
  y = std::fmaf(y, f, -2.4999993993e-1f);
  y = std::fmaf(y, f, 3.3333331174e-1f);
  y = (y * f) * z;
  const float fe = (float)e;
  y = std::fmaf(-2.12194440e-4f, fe, y);
  y = std::fmaf(-0.5f, z, y);
  return std::fmaf(0.693359375f, fe, f + y);
}

extern "C" int smesh_oracle_log_spec(const float* in, float* out, uint64_t n) {   // test hook (tests/test_oracle.py)
  for (uint64_t i = 0; i < n; i++) out[i] = log_spec(in[i]);
  return SMESH_OK;
}

// ModelAggregator::add1/add2 (Fusion.h:42-64) -> ModelAggregator::add (Mesh.h:65-107)
int smesh_aggregator_add(smesh_aggregator_t* a, const void* indices, int idx_dtype, const int64_t is[2], int imem,
                         const float* probs, const int64_t ps[3], int pmem,
                         const float* weights, const int64_t ws[2], int wmem,
                         uint64_t W, uint64_t H) {
  if (!a || !indices || !probs || !is || !ps) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (imem != SMESH_MEM_HOST || pmem != SMESH_MEM_HOST || (weights && wmem != SMESH_MEM_HOST))
    return fail(SMESH_ERR_INVALID, "oracle only reads host memory");
  if (idx_dtype < 0 || idx_dtype > 3) return fail(SMESH_ERR_INVALID, "bad index dtype");
  if (weights && !ws) return fail(SMESH_ERR_INVALID, "weights without strides");
  const uint64_t N = W * H;
  const uint32_t C = a->C;
  const uint64_t P = a->P;
  const bool par = g_threads > 1;

  // Fusion.h:45-47 -- TensorConstructor: OpenMP element copy (+cast) into fresh row-major host tensors
  std::vector<uint32_t> idx(N);
  std::vector<float> pr(N * C);
  std::vector<float> wt(weights ? N : 0);
#pragma omp parallel for num_threads(g_threads) schedule(static) if (par)
  for (int64_t x = 0; x < (int64_t)W; x++)
    for (uint64_t y = 0; y < H; y++) {
      const uint64_t i = (uint64_t)x * H + y;
      idx[i] = load_idx(indices, idx_dtype, x * is[0] + (int64_t)y * is[1]);
      for (uint32_t c = 0; c < C; c++) pr[i * C + c] = probs[x * ps[0] + (int64_t)y * ps[1] + (int64_t)c * ps[2]];
      if (weights) wt[i] = weights[x * ws[0] + (int64_t)y * ws[1]];
    }

  // Mesh.h:90-93 -- serial std::map histogram over ALL pixels of this image
  std::map<size_t, size_t> pixels_per_face;
  std::vector<uint32_t> dense_count;   // optimised variant (not the reference's algorithm): same counts, dense and parallel
  if (g_fast_histogram) {
    dense_count.assign(P + 1, 0u);     // slot P collects the background / out-of-range values
#pragma omp parallel for num_threads(g_threads) schedule(static) if (par)
    for (int64_t i = 0; i < (int64_t)N; i++) {
      const size_t v = idx[i] < P ? (size_t)idx[i] : (size_t)P;
#pragma omp atomic
      dense_count[v] += 1u;
    }
  } else {
    for (uint64_t i = 0; i < N; i++) pixels_per_face.insert({(size_t)idx[i], 0}).first->second += 1;
  }

  // Mesh.h:94-106 -- OpenMP 2-D loop, one lock per primitive
  const float iew = a->iew;
  const int kind = a->kind;
#pragma omp parallel for num_threads(g_threads) schedule(static) if (par)
  for (int64_t i = 0; i < (int64_t)N; i++) {
    const size_t primitive_index = idx[i];
    if (!(primitive_index < P)) continue;                       // Mesh.h:95
    const float* next = &pr[(uint64_t)i * C];
    float sum = 0.0f;
    for (uint32_t c = 0; c < C; c++) sum = sum + next[c];        // tt::sum, sequential float32
    if (!(sum > 0.5f)) continue;                                // Mesh.h:98 "Not the don't-care class"
    const size_t npix = g_fast_histogram ? (size_t)dense_count[primitive_index] : pixels_per_face.find(primitive_index)->second;
    const float image_weight = 1.0f / ((float)npix);                                         // :100
    const float pixel_weight = 1.0f;                                                        // :101
    const float image_pixel_weight = iew * image_weight + (1 - iew) * pixel_weight;           // :102
    const float w = image_pixel_weight * (weights ? wt[i] : 1.0f);                           // :103
    std::unique_lock<std::mutex> guard(a->locks[primitive_index], std::defer_lock);
    if (par) guard.lock();
    if (kind == SMESH_AGG_SUM) {
      // aggregator::weighted::sum (Fusion.cu:70-73): s += p * w
      if (g_accum_double) { double* s = &a->accd[primitive_index * C]; for (uint32_t c = 0; c < C; c++) s[c] += (double)(next[c] * w); }
      else { float* s = &a->acc[primitive_index * C]; for (uint32_t c = 0; c < C; c++) s[c] = s[c] + next[c] * w; }
    } else if (kind == SMESH_AGG_SUMMAX) {
      // Fusion.cu:51-56: only the (first) arg-max class contributes prob*weight
      uint32_t m = 0;
      for (uint32_t c = 1; c < C; c++) if (next[c] > next[m]) m = c;
      if (g_accum_double) a->accd[primitive_index * C + m] += (double)(next[m] * w);
      else a->acc[primitive_index * C + m] = a->acc[primitive_index * C + m] + next[m] * w;
    } else {
      // Fusion.cu:83-87: map_input(pow) -> prod of LogProb<float>: L += log(p^w).  Whether p^w is rounded to float before
      // the log is decided inside the absent template-tensors; the spec (the same formula in the HIP
      // kernels) is w * log(p) in float32 with p^0 = 1 for every p.
      for (uint32_t c = 0; c < C; c++) {
        // g_mul_literal: pow() rounds p^w to float32 -- it underflows to 0 (log: -inf, the class is eliminated for good) where
        // w * log(p) < log(FLT_TRUE_MIN) ~ -103.3, e.g. p = 1e-30 with w = 2; the spec'd form stays finite there
        const float l = g_mul_literal ? std::log(std::pow(next[c], w)) : (w == 0.0f ? 0.0f : w * log_spec(next[c]));
        if (g_accum_double) a->accd[primitive_index * C + c] += (double)l;
        else a->acc[primitive_index * C + c] = a->acc[primitive_index * C + c] + l;
      }
    }
  }
  return SMESH_OK;
}

static inline float nan_inf_to_zero(float v) { return (std::isnan(v) || std::isinf(v)) ? 0.0f : v; }  // Fusion.h:79-95

// ModelAggregator::get (Fusion.h:72-76, Mesh.h:131-132) + functor chains (Fusion.cu:47-49,67-69,79-82)
static int get_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, float* out, int memkind);
int smesh_aggregator_get(smesh_aggregator_t* a, float* out, int memkind) {
  if (!a) return fail(SMESH_ERR_INVALID, "NULL argument");
  return get_rows(a, 0, a->P, out, memkind);
}
int smesh_aggregator_get_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, float* out, int memkind) {
  if (!a || row_lo > row_hi || row_hi > a->P || (row_lo & 3)) return fail(SMESH_ERR_INVALID, "bad row range");
  return get_rows(a, row_lo, row_hi, out, memkind);
}
static int get_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, float* out, int memkind) {
  if (!a || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (memkind != SMESH_MEM_HOST) return fail(SMESH_ERR_INVALID, "oracle only writes host memory");
  const uint32_t C = a->C;
  std::vector<float> row(C);
  out -= row_lo * C;
  for (uint64_t p = row_lo; p < row_hi; p++) {
    for (uint32_t c = 0; c < C; c++) row[c] = g_accum_double ? (float)a->accd[p * C + c] : a->acc[p * C + c];
    if (a->kind == SMESH_AGG_MUL && g_accum_double) {
      // the float64 yardstick keeps its precision through the division by the largest element: casting a log-sum of magnitude
      // 1e4 to float first (as the float32 state of the reference does) would cost 5e-4 relative after exp()
      double m = a->accd[p * C];
      for (uint32_t c = 1; c < C; c++) if (a->accd[p * C + c] > m) m = a->accd[p * C + c];
      for (uint32_t c = 0; c < C; c++) row[c] = (float)std::exp(a->accd[p * C + c] - m);
    } else if (a->kind == SMESH_AGG_MUL) {
      // logprob_normalize (Fusion.h:97-104): p / max_el(p) in the log domain, then cast to float
      float m = row[0];
      for (uint32_t c = 1; c < C; c++) if (row[c] > m) m = row[c];
      for (uint32_t c = 0; c < C; c++) row[c] = std::exp(row[c] - m);
    }
    float n = 0.0f;  // normalize<l1_norm> (Fusion.cu:48,68,80)
    for (uint32_t c = 0; c < C; c++) n = n + std::fabs(row[c]);
    for (uint32_t c = 0; c < C; c++) out[p * C + c] = nan_inf_to_zero(row[c] / n);
  }
  return SMESH_OK;
}

int smesh_aggregator_get_raw(smesh_aggregator_t* a, float* out, int memkind) {
  if (!a || !out || memkind != SMESH_MEM_HOST) return fail(SMESH_ERR_INVALID, "bad argument");
  const uint64_t n = a->P * a->C;
  if (g_accum_double) for (uint64_t i = 0; i < n; i++) out[i] = (float)a->accd[i];
  else std::memcpy(out, a->acc.data(), n * sizeof(float));
  return SMESH_OK;
}

int smesh_aggregator_set_raw(smesh_aggregator_t* a, const float* in, int memkind) {
  if (!a || !in || memkind != SMESH_MEM_HOST) return fail(SMESH_ERR_INVALID, "bad argument");
  const uint64_t n = a->P * a->C;
  if (g_accum_double) for (uint64_t i = 0; i < n; i++) a->accd[i] = in[i];
  else std::memcpy(a->acc.data(), in, n * sizeof(float));
  return SMESH_OK;
}

int smesh_aggregator_raw_pointer(smesh_aggregator_t* a, void** ptr, uint64_t* n) {
  if (!a || !ptr) return fail(SMESH_ERR_INVALID, "bad argument");
  if (g_accum_double) return fail(SMESH_ERR_INVALID, "raw pointer unavailable with float64 accumulators");
  *ptr = a->acc.data();
  if (n) *n = a->P * a->C;
  return SMESH_OK;
}

int smesh_aggregator_row_stride(smesh_aggregator_t* a, uint32_t* stride) {
  if (!a || !stride) return fail(SMESH_ERR_INVALID, "bad argument");
  *stride = a->C;
  return SMESH_OK;
}

// ModelAggregator::renderer (Mesh.h:124-129) + ModelRenderer::render (Mesh.h:25-42)
struct smesh_annotation_renderer {
  uint64_t P; uint32_t C;
  std::vector<float> ann;
};

int smesh_aggregator_renderer(smesh_aggregator_t* a, smesh_annotation_renderer_t** out) {
  if (!a || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  auto* r = new (std::nothrow) smesh_annotation_renderer();
  if (!r) return fail(SMESH_ERR_RUNTIME, "out of memory");
  r->P = a->P; r->C = a->C;
  r->ann.resize(a->P * a->C);
  int st = smesh_aggregator_get(a, r->ann.data(), SMESH_MEM_HOST);   // m_annotations = elwise(get) (:127)
  if (st) { delete r; return st; }
  *out = r;
  return SMESH_OK;
}

int smesh_annotation_renderer_render(smesh_annotation_renderer_t* r, const void* indices, int idx_dtype, const int64_t is[2],
                                     int imem, const float* background, float* out, int omem, uint64_t W, uint64_t H) {
  if (!r || !indices || !is || !background || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  if (imem != SMESH_MEM_HOST || omem != SMESH_MEM_HOST) return fail(SMESH_ERR_INVALID, "oracle only handles host memory");
  if (idx_dtype < 0 || idx_dtype > 3) return fail(SMESH_ERR_INVALID, "bad index dtype");
  for (uint64_t x = 0; x < W; x++)
    for (uint64_t y = 0; y < H; y++) {
      const size_t primitive_index = load_idx(indices, idx_dtype, (int64_t)x * is[0] + (int64_t)y * is[1]);
      float* p = out + (x * H + y) * r->C;
      if (primitive_index < r->P) std::memcpy(p, &r->ann[primitive_index * r->C], r->C * sizeof(float));   // :34-36
      else std::memcpy(p, background, r->C * sizeof(float));                                                // :38-40
    }
  return SMESH_OK;
}

int smesh_annotation_renderer_destroy(smesh_annotation_renderer_t* r) { delete r; return SMESH_OK; }

int smesh_fuse_view(smesh_renderer_t* r, smesh_aggregator_t* a, const smesh_camera_t* cam,
                    const float* probs, const float* weights, int memkind) {
  if (!r || !a || !cam || !probs) return fail(SMESH_ERR_INVALID, "NULL argument");
  const uint64_t W = cam->width, H = cam->height;
  std::vector<uint32_t> idx(W * H);
  int st = smesh_renderer_render(r, cam, idx.data(), nullptr);
  if (st) return st;
  const int64_t is[2] = {(int64_t)H, 1};
  const int64_t ps[3] = {(int64_t)H * a->C, (int64_t)a->C, 1};
  return smesh_aggregator_add(a, idx.data(), SMESH_IDX_U32, is, SMESH_MEM_HOST, probs, ps, memkind,
                              weights, is, memkind, W, H);
}

int smesh_fuse_views(smesh_renderer_t* r, smesh_aggregator_t* a, const smesh_camera_t* cams, uint64_t n,
                     const float* const* probs, const float* const* weights, int memkind) {
  if (n && (!cams || !probs)) return fail(SMESH_ERR_INVALID, "NULL argument");
  for (uint64_t i = 0; i < n; i++) {   // the reference's loop: one view after the other
    const int st = smesh_fuse_view(r, a, &cams[i], probs[i], weights ? weights[i] : nullptr, memkind);
    if (st) return st;
  }
  return SMESH_OK;
}

// --------------------------------------------------------------------------------------------
int smesh_aggregator_add_matched(smesh_aggregator_t*, smesh_renderer_t*, const void*, int, const int64_t*, int, const float*, const int64_t*, int,
                                 const float*, const int64_t*, int, uint64_t, uint64_t, int* matched) {
  if (matched) *matched = 0;   // the oracle keeps no per-render records: every add() is the reference's scatter
  return SMESH_OK;
}
int smesh_renderer_seal_render(smesh_renderer_t*, const uint32_t*) { return SMESH_OK; }
int smesh_box_extent_bound(const float*, uint64_t, const int32_t*, uint64_t, const smesh_camera_t*, double* bound) {   // (no proofs on the oracle's side)
  if (bound) *bound = INFINITY;
  return SMESH_OK;
}
int smesh_renderer_render_stats(smesh_renderer_t*, const smesh_camera_t*, int* needed, uint32_t q[4]) {   // (the oracle has one loop for every triangle)
  if (needed) *needed = 1;
  if (q) q[0] = q[1] = q[2] = q[3] = 0u;
  return SMESH_OK;
}
const char* smesh_last_fuse_kernel(void) { return "oracle"; }
const char* smesh_last_add_path(void) { return "oracle"; }
int smesh_aggregator_add_rendered(smesh_aggregator_t* a, smesh_renderer_t*, const uint32_t* idx, const float* probs,
                                  const int64_t ps[3], int pmem, const float* weights, const int64_t ws[2], int wmem,
                                  uint64_t W, uint64_t H) {
  const int64_t is[2] = {(int64_t)H, 1};   // the oracle has one add(): the reference's
  return smesh_aggregator_add(a, idx, SMESH_IDX_U32, is, SMESH_MEM_HOST, probs, ps, pmem, weights, ws, wmem, W, H);
}
int smesh_aggregator_add_async(smesh_aggregator_t* a, const void* indices, int idx_dtype, const int64_t is[2], int imem,
                               const float* probs, const int64_t ps[3], int pmem,
                               const float* weights, const int64_t ws[2], int wmem, uint64_t W, uint64_t H) {
  return smesh_aggregator_add(a, indices, idx_dtype, is, imem, probs, ps, pmem, weights, ws, wmem, W, H);   // (synchronous like everything here)
}
int smesh_aggregator_add_many(smesh_aggregator_t* a, uint64_t n, const void* const* indices, int idx_dtype, const int64_t is[2], int imem,
                              const float* const* probs, const int64_t ps[3], int pmem,
                              const float* const* weights, const int64_t ws[2], int wmem, uint64_t W, uint64_t H) {
  if (n && (!indices || !probs)) return fail(SMESH_ERR_INVALID, "NULL argument");
  for (uint64_t i = 0; i < n; i++) {      // the reference's add(), image by image (Mesh.h:65-107)
    const int st = smesh_aggregator_add(a, indices[i], idx_dtype, is, imem, probs[i], ps, pmem, weights ? weights[i] : nullptr, ws, wmem, W, H);
    if (st != SMESH_OK) return st;
  }
  return SMESH_OK;
}
int smesh_stream_wait(int, void*) { return SMESH_OK; }   // the oracle has no streams: everything is synchronous
int smesh_stream_release(int, void*) { return SMESH_OK; }
int smesh_stream_handle(int, void** s) { if (s) *s = nullptr; return SMESH_OK; }
int smesh_token_record(int, uint64_t* t) { if (t) *t = 0; return SMESH_OK; }
int smesh_token_done(int, uint64_t, int* d) { if (d) *d = 1; return SMESH_OK; }
int smesh_stream_mark(int, int) { return SMESH_OK; }
int smesh_stream_mark_elapsed(int, int, int, double* ms) { if (ms) *ms = 0.0; return SMESH_OK; }
// multi-GPU exchange: not part of the CPU restatement (tests sum the shards' raw accumulators themselves)
int smesh_comm_unique_id(uint8_t*) { return fail(SMESH_ERR_NODEVICE, "oracle has no communicator"); }
int smesh_comm_create(int, int, int, const uint8_t*, smesh_comm_t**) { return fail(SMESH_ERR_NODEVICE, "oracle has no communicator"); }
int smesh_comm_create_all(const int*, int, smesh_comm_t**) { return fail(SMESH_ERR_NODEVICE, "oracle has no communicator"); }
int smesh_comm_destroy(smesh_comm_t*) { return SMESH_OK; }
int smesh_comm_rank(const smesh_comm_t*, int*, int*) { return fail(SMESH_ERR_NODEVICE, "oracle has no communicator"); }
int smesh_allreduce(smesh_comm_t* const*, smesh_aggregator_t* const*, int) { return fail(SMESH_ERR_NODEVICE, "oracle has no communicator"); }
int smesh_reduce_scatter(smesh_comm_t*, smesh_aggregator_t*, uint64_t*, uint64_t*) { return fail(SMESH_ERR_NODEVICE, "oracle has no communicator"); }
int smesh_comm_allreduce_f64(smesh_comm_t*, double*, int, int) { return fail(SMESH_ERR_NODEVICE, "oracle has no communicator"); }
int smesh_allreduce_rows(smesh_comm_t*, smesh_aggregator_t*, uint64_t, uint64_t) { return fail(SMESH_ERR_NODEVICE, "oracle has no communicator"); }
int smesh_exchange_join(smesh_aggregator_t*) { return SMESH_OK; }
// fusion by row range: the oracle fuses everything with part 0 (rows [0, P)); the later parts are empty
int smesh_fuse_views_begin(smesh_renderer_t* r, smesh_aggregator_t* a, const smesh_camera_t* cams, uint64_t n, const float* const* probs,
                           const float* const* weights, int memkind, int nparts, uint64_t* row_lo, uint64_t* row_hi) {
  if (!a || !row_lo || !row_hi || nparts < 1) return fail(SMESH_ERR_INVALID, "bad argument");
  const int st = smesh_fuse_views(r, a, cams, n, probs, weights, memkind);
  *row_lo = 0; *row_hi = a->P;
  return st;
}
int smesh_fuse_views_continue(smesh_renderer_t*, smesh_aggregator_t* a, int, uint64_t* row_lo, uint64_t* row_hi) {
  if (!a || !row_lo || !row_hi) return fail(SMESH_ERR_INVALID, "bad argument");
  *row_lo = *row_hi = a->P;
  return SMESH_OK;
}
// plane 0: the state rounded to float32 (the product's hi plane); plane 1 (Mul, float64 yardstick only): what the rounding left, so
// that hi + lo carries the state across a host-side exchange as the product's (hi, lo) pairs do (set: plane 0 first, then plane 1)
int smesh_aggregator_get_raw_rows(smesh_aggregator_t* a, uint64_t lo, uint64_t hi, int plane, float* out, int memkind) {
  if (!a || !out || memkind != SMESH_MEM_HOST || lo > hi || hi > a->P || plane < 0 || plane > 1) return fail(SMESH_ERR_INVALID, "bad argument");
  if (plane == 1 && a->kind != SMESH_AGG_MUL) return fail(SMESH_ERR_INVALID, "plane 1 is the Mul aggregator's");
  for (uint64_t i = lo * a->C; i < hi * a->C; i++) {
    const float h = g_accum_double ? (float)a->accd[i] : a->acc[i];
    out[i - lo * a->C] = plane == 0 ? h : ((g_accum_double && std::isfinite(h)) ? (float)(a->accd[i] - (double)h) : 0.0f);
  }
  return SMESH_OK;
}
int smesh_aggregator_set_raw_rows(smesh_aggregator_t* a, uint64_t lo, uint64_t hi, int plane, const float* in, int memkind) {
  if (!a || !in || memkind != SMESH_MEM_HOST || lo > hi || hi > a->P || plane < 0 || plane > 1) return fail(SMESH_ERR_INVALID, "bad argument");
  if (plane == 1 && a->kind != SMESH_AGG_MUL) return fail(SMESH_ERR_INVALID, "plane 1 is the Mul aggregator's");
  for (uint64_t i = lo * a->C; i < hi * a->C; i++) {
    const float v = in[i - lo * a->C];
    if (plane == 0) { if (g_accum_double) a->accd[i] = v; else a->acc[i] = v; }
    else if (g_accum_double) a->accd[i] += (double)v;
  }
  return SMESH_OK;
}
int smesh_profile_enable(int, int) { return SMESH_OK; }
int smesh_profile_sample_every(int, uint32_t) { return SMESH_OK; }
int smesh_profile_read(int, int, double* ms, uint64_t* n) { if (ms) *ms = 0; if (n) *n = 0; return SMESH_OK; }
int smesh_profile_read_ex(int, int, double* ms, uint64_t* r, uint64_t* l, uint64_t* v) { if (ms) *ms = 0; if (r) *r = 0; if (l) *l = 0; if (v) *v = 0; return SMESH_OK; }
int smesh_profile_regions(int, int, uint64_t* n) { if (n) *n = 0; return SMESH_OK; }
int smesh_profile_reset(int) { return SMESH_OK; }

// Synthetic probabilities.  Arithmetic is chosen so that CPU and GPU produce the same
// bits: hashed uniforms u in [0,1) -> q = u^4 (peaky, softmax-like) -> p = q / sum(q).
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

int smesh_synth_probs(float* out, uint64_t N, uint32_t C, uint64_t seed, float zero_fraction, int, int memkind) {
  if (!out || memkind != SMESH_MEM_HOST) return fail(SMESH_ERR_INVALID, "bad argument");
  const uint32_t zthr = (uint32_t)(zero_fraction * 16777216.0f);
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
  for (int64_t i = 0; i < (int64_t)N; i++) {
    float* row = out + (uint64_t)i * C;
    const uint64_t hz = splitmix64(seed ^ (0xD1B54A32D192ED03ull * ((uint64_t)i + 1)));
    if ((uint32_t)(hz >> 40) < zthr) { for (uint32_t c = 0; c < C; c++) row[c] = 0.0f; continue; }
    float s = 0.0f;
    for (uint32_t c = 0; c < C; c++) {
      const uint64_t h = splitmix64(seed + 0x632BE59BD9B4E019ull * ((uint64_t)i * C + c + 1));
      const float u = (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
      const float u2 = u * u;
      const float q = u2 * u2 + 1e-4f;
      row[c] = q;
      s = s + q;
    }
    for (uint32_t c = 0; c < C; c++) row[c] = row[c] / s;
  }
  return SMESH_OK;
}

int smesh_host_malloc(uint64_t bytes, void** out) { if (!out) return fail(SMESH_ERR_INVALID, "out is NULL"); *out = malloc(bytes ? bytes : 1); return *out ? SMESH_OK : fail(SMESH_ERR_RUNTIME, "out of memory"); }
int smesh_host_free(void* p) { free(p); return SMESH_OK; }
int smesh_device_malloc(int, uint64_t, void**) { return fail(SMESH_ERR_NODEVICE, "oracle has no device memory"); }
int smesh_device_free(int, void*) { return SMESH_OK; }
int smesh_device_trim(int, uint64_t* cached_bytes) { if (cached_bytes) *cached_bytes = 0; return SMESH_OK; }
int smesh_set_option(const char* name, int64_t) { return name ? SMESH_OK : fail(SMESH_ERR_INVALID, "option name is NULL"); }   // (the oracle has no streams)
int smesh_get_option(const char* name, int64_t* value) { if (!name || !value) return fail(SMESH_ERR_INVALID, "NULL argument"); *value = 0; return SMESH_OK; }
int smesh_memcpy(void* dst, const void* src, uint64_t bytes, int dk, int sk, int) {
  if (dk != SMESH_MEM_HOST || sk != SMESH_MEM_HOST) return fail(SMESH_ERR_NODEVICE, "oracle has no device memory");
  std::memmove(dst, src, bytes);
  return SMESH_OK;
}

}  // extern "C"
