// image_records.hip -- per-primitive records built from an ARBITRARY index image, so that MeshAggregator::add() on an image the
// library did not render (the reference's harness reloads its renders from an .npz cache, eval-scannet/eval_scannet.py:168-185;
// other renderers; images a caller edited) can take the triangle-order fusion kernels (fusion.hip, k_fuse_tri and friends) instead
// of the atomic scatter-add: every accumulator row then has one owner, the additions happen in the reference's pixel order
// (Mesh.h:94-106 run single-threaded) and the result no longer depends on the order in which float atomics land.
//
// What the rasteriser leaves per triangle (common.hpp, TriFrag) is rebuilt per PRIMITIVE from the image alone.  Round 3's passes
// (M, R, E / C' / D: one atomic per (primitive, strip) group, further down) serve every image of 16 .. 16383 pixels a side with fewer
// than 2^24 pixels; round 2's passes serve the rest and stay behind SMESH_REC_MOMENTS=0:
//   pass A  k_rec_origin   per run of equal indices in a column: ONE 32-bit atomic that leaves the primitive's first pixel in
//                          (x, y) order -- its smallest x, and the smallest y of that column (atomic max on the complement of
//                          x << 16 | y: the cleared state, 0, means "no pixel")
//   pass B  k_rec_mask     per run: inside the 8 x 8 box whose origin is (that x, that y - 3) -> its bits OR-ed into the 64-bit
//                          mask (kind 1: the mask is exactly the primitive's pixels, its population count the histogram entry of
//                          Mesh.h:90-93); outside -> the primitive is "big": extent and pixel count by atomics, queued once
//   pass C  k_rec_big      per queued primitive: one that fits 8 x 8 after all (just not at that origin) gets its mask at the true
//                          origin; else a kind 2 record with its bounding box (the fusion scans the box in the image: one
//                          wave per primitive) -- unless the box is much larger than the primitive's pixel count (an index image
//                          that is not a rendering: scattered pixels), in which case scanning boxes would cost O(P N): such
//                          "sparse" primitives are taken out of the triangle-order launch (kind 0) and
//   (fusion launch: smesh_aggregator_fuse_triangles)
//   pass D  k_scatter_sparse  adds their pixels in pixel order with float atomics (a no-op launch for renderings)
//   clear                  one memset when P <= N, else per pixel: the scratch is all zero between calls
// Atomics are what this costs (MI355X: ~30 us per million requests, whatever their width), hence one per run and pass.  Runs: a
// wave covers 64 consecutive pixels of the y-fastest image; neighbouring lanes with the same index in the same column form a run
// and only its first lane issues atomics -- a primitive covering thousands of pixels would otherwise serialise thousands of
// same-address atomics.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

using namespace smesh;

namespace {

#include "fuse_tri.inc.hpp"   // contribution<KIND>() (the Mul logarithm is part of the spec)
#include "strip.inc.hpp"      // runs / chains / groups of a 4 x 16-pixel strip (shared with the histogram kernel)

constexpr int kBlock = 256;
constexpr unsigned long long kDenseFactor = 16ull, kDenseSlack = 256ull;   // a box scan may cost 16 x the pixel count + 256 reads

struct RunInfo {
  uint32_t v;        // primitive id of the lane's pixel (0xFFFFFFFF: none / past the end)
  uint32_t x, y;     // pixel
  uint32_t len;      // leader lanes: pixels in the run (consecutive y); other lanes: 0
};

// One lane per pixel, 64 consecutive pixels per wave.
__device__ __forceinline__ RunInfo find_run(const uint32_t* __restrict__ idx, uint64_t N, uint32_t H, uint32_t P) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  const int l = threadIdx.x & 63;
  RunInfo r;
  r.v = i < N ? idx[i] : 0xFFFFFFFFu;
  if (r.v >= P) r.v = 0xFFFFFFFFu;
  r.x = (uint32_t)(i / H);
  r.y = (uint32_t)(i - (uint64_t)r.x * H);
  const uint32_t prev = (uint32_t)__shfl_up((int)r.v, 1);
  const bool leader = l == 0 || prev != r.v || r.y == 0u;
  const unsigned long long L = __ballot(leader);
  const unsigned long long after = l == 63 ? 0ull : (L >> (l + 1));
  const uint32_t next = after ? (uint32_t)l + 1u + (uint32_t)__builtin_ctzll(after) : 64u;
  r.len = (leader && r.v != 0xFFFFFFFFu) ? next - (uint32_t)l : 0u;
  return r;
}

// (x0, y0) of the 8 x 8 box of a primitive whose first pixel in (x, y) order is `first` = ~cand
__device__ __forceinline__ void box_origin(uint32_t cand, uint32_t& x0, uint32_t& y0) {
  const uint32_t first = ~cand;
  x0 = first >> 16;
  const uint32_t yf = first & 0xFFFFu;
  y0 = yf > 3u ? yf - 3u : 0u;
}

__global__ __launch_bounds__(kBlock) void k_rec_origin(const uint32_t* __restrict__ idx, uint64_t N, uint32_t H, uint32_t P,
                                                       uint32_t* __restrict__ cand) {
  const RunInfo r = find_run(idx, N, H, P);
  if (r.len == 0u) return;
  atomicMax(&cand[r.v], ~((r.x << 16) | r.y));     // x, y <= 65534: never 0
}

__global__ __launch_bounds__(kBlock) void k_rec_mask(const uint32_t* __restrict__ idx, uint64_t N, uint32_t H, uint32_t P,
                                                     const uint32_t* __restrict__ cand, uint4* __restrict__ big4,
                                                     TriFrag* __restrict__ frags, uint32_t* __restrict__ big_queue,
                                                     uint32_t* __restrict__ big_count) {
  const RunInfo r = find_run(idx, N, H, P);
  if (r.len == 0u) return;
  uint32_t x0, y0;
  box_origin(cand[r.v], x0, y0);
  const uint32_t dx = r.x - x0, dy0 = r.y - y0, dy1 = dy0 + r.len - 1u;    // (a run above the box: dy0 wraps around, dy1 >= 8 or < dy0)
  if (dx < 8u && dy0 < 8u && dy1 < 8u) {
    const unsigned long long bits = ((1ull << r.len) - 1ull) << (dx * 8u + dy0);    // bit dx * 8 + dy (common.hpp, TriFrag)
    atomicOr(&frags[r.v].mask, bits);
    // the record's head: every run of the primitive stores the same two words (pass C overrules them for big primitives)
    *reinterpret_cast<uint2*>(&frags[r.v]) = make_uint2(x0 | (y0 << 16), 1u);
  } else {
    uint32_t* b = reinterpret_cast<uint32_t*>(&big4[r.v]);
    const uint32_t old = atomicMax(&b[0], r.x + 1u);   // largest x + 1
    atomicMax(&b[1], r.y + r.len);                     // largest y + 1
    atomicAdd(&b[2], r.len);                           // pixels outside the box
    atomicMax(&b[3], 65536u - r.y);                    // 65536 - smallest y
    if (old == 0u) {                                   // first run outside the box: queue the primitive, once
      const uint32_t slot = atomicAdd(big_count, 1u);
      big_queue[slot] = r.v;                           // capacity P: a primitive is queued at most once
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_rec_big(const uint32_t* __restrict__ idx, uint32_t H, const uint32_t* __restrict__ cand,
                                                    uint4* __restrict__ big4, TriFrag* __restrict__ frags,
                                                    const uint32_t* __restrict__ big_queue, uint32_t* __restrict__ big_count) {
  const uint32_t nbig = *big_count;
  for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < nbig; q += gridDim.x * kBlock) {
    const uint32_t v = big_queue[q];
    const uint4 b = big4[v];
    uint32_t x0, y0;
    box_origin(cand[v], x0, y0);                 // x0 is the primitive's smallest x
    uint32_t x1 = b.x - 1u, y1 = b.y - 1u, ytop = 65536u - b.w;
    // pixels inside the 8 x 8 box went into the mask: their extent counts too
    const unsigned long long mask = frags[v].mask;
    if (mask) {
      const uint32_t maxdx = (63u - (uint32_t)__builtin_clzll(mask)) >> 3;
      uint32_t rows = (uint32_t)(mask | (mask >> 32));
      rows |= rows >> 16;
      rows |= rows >> 8;
      rows &= 0xFFu;
      x1 = max(x1, x0 + maxdx);
      y1 = max(y1, y0 + 31u - (uint32_t)__builtin_clz(rows));
      ytop = min(ytop, y0 + (uint32_t)__builtin_ctz(rows));
    }
    const unsigned long long n = (unsigned long long)b.z + (unsigned long long)__popcll(mask);   // Mesh.h:90-93 for this primitive
    const unsigned long long area = (unsigned long long)(x1 - x0 + 1u) * (unsigned long long)(y1 - ytop + 1u);
    TriFrag rec;
    rec.x0 = (uint16_t)x0; rec.y0 = (uint16_t)ytop; rec.pad = 0;
    if (x1 - x0 < 8u && y1 - ytop < 8u) {
      // fits 8 x 8 after all, just not at the origin pass B had to guess from the first pixel: the mask again, at the true origin --
      // every primitive of up to 8 x 8 pixels is a kind 1 record (one lane, the reference's order of additions)
      unsigned long long m = 0ull;
      for (uint32_t dx = 0; dx <= x1 - x0; dx++)
        for (uint32_t dy = 0; dy <= y1 - ytop; dy++)
          if (idx[(uint64_t)(x0 + dx) * H + ytop + dy] == v) m |= 1ull << (dx * 8u + dy);
      rec.kind = 1;
      rec.mask = m;
    } else if (area <= kDenseFactor * n + kDenseSlack) {
      rec.kind = 2;
      rec.mask = (unsigned long long)x1 | ((unsigned long long)y1 << 16);
    } else {            // sparse: not for the triangle-order kernels (kind 0 = nothing emitted); k_scatter_sparse finds it by `pad`
      rec.kind = 0; rec.pad = 1;
      rec.mask = n;
      big_count[2] = 1u;
    }
    frags[v] = rec;
    big4[v] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// ================================================================================================================================
// Round 3: records from MOMENTS.  Passes A and B above cost two atomics per run of pixels (1.5 M runs in a cfg2 image: ~90 us),
// and atomics are all they cost.  This variant issues ONE atomic per (primitive, 4 x 16 strip) group -- 0.8 M for the same image:
//   pass M  k_rec_moments  one wave per strip (strip.inc.hpp: runs linked into chains across the strip's four columns); the root
//                          lane of each group adds the group's pixel count, sum of x and sum of y to the primitive's 64-bit moment
//                          word (count: 24 bits, sums: 20 bits each -- exact for up to 64 pixels below 16384 x 16384; beyond, only
//                          the count is used) and STORES the group's own 8 x 8 record (plainly, tagged with the call's tag).
//   pass R  k_rec_resolve  one lane per primitive: a stored record whose population count equals the primitive's total count IS the
//                          primitive (a group that holds every pixel has no rival writer) -- 85 % of cfg2's primitives.  Otherwise
//                          the lane scans the index image around the centroid (4 x 4, 8 x 8, then 16 x 16 pixels): all `count` pixels found
//                          -> the exact record (mask at the true origin, or a kind 2 box); else (more than 64 pixels, or pixels
//                          further than 7 from the centroid) the primitive is queued as "pending", and
//   passes E, C', D  k_rec_tail  (one launch, grid barriers; leaves at once when nothing is pending) E: per run of a pending primitive
//                          its extent by atomics; C': per pending primitive the kind 2 record, or "sparse" exactly as k_rec_big
//                          decides it; D: the sparse primitives' pixels by float atomics, ahead of the fusion launch.
// No clear pass: R rewrites every record that differs and zeroes the moment words it consumed, C' its extent words, M the counters.
// ================================================================================================================================
constexpr uint32_t kPadPending = 3u;    // (1: sparse, k_rec_big; 2 and 4: the alternating tags of pass M's own records)
constexpr int kMomCountBits = 24, kMomSumBits = 20;

typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment (a window row of the index image)

__device__ __forceinline__ unsigned long long shfl64(unsigned long long x, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)x, src), hi = (uint32_t)__shfl((int)(uint32_t)(x >> 32), src);
  return (unsigned long long)lo | ((unsigned long long)hi << 32);
}

// Pass M.  The strip's runs, links and groups are those of strip.inc.hpp (build_strip: the histogram and scatter-add kernels), but
// worked out in registers: equal neighbours across columns are one wave shuffle, two DPP row shifts and three ballots, a run's first match in the next
// column is bit arithmetic on them, and the (at most four) runs of a chain are folded into its root by two pointer-doubling hops --
// no LDS, no loops over rows.
__device__ __forceinline__ void rec_moments_block(const uint32_t* __restrict__ idx, uint32_t W, uint32_t H, uint32_t P, uint32_t strips_y,
                                                  uint32_t nstrips, uint32_t strips_per_xcd, uint32_t tag, int dbg,
                                                  unsigned long long* __restrict__ mom, TriFrag* __restrict__ frags,
                                                  uint32_t* __restrict__ big_count) {
  const uint32_t b = blockIdx.x;
  const int l = threadIdx.x;
  if (b == 0u && l == 0) { big_count[0] = 0u; big_count[2] = 0u; big_count[3] = 0u; }   // (R, later on the stream, is the first to touch them)
  // block b runs on XCD b % 8 (observed, MI355X_MICROARCH.md): neighbouring strips share an L2; only speed depends on it
  const uint32_t s = (b & 7u) * strips_per_xcd + (b >> 3);
  if (s >= nstrips) return;
  const uint32_t bx = s / strips_y, by = s - bx * strips_y;
  const uint32_t x0 = bx * kSX, y0 = by * kTY;
  const int cx = l / kTY, ty = l - cx * kTY;
  const bool in = x0 + (uint32_t)cx < W && y0 + (uint32_t)ty < H;
  uint32_t v = in ? idx[(uint64_t)(x0 + cx) * H + y0 + ty] : 0xFFFFFFFFu;
  if (v >= P) v = 0xFFFFFFFFu;
  const bool valid = v != 0xFFFFFFFFu;
  // ---- runs down the columns
  const uint32_t prev = (uint32_t)__shfl_up((int)v, 1);
  const bool head = ty == 0 || v != prev;
  const unsigned long long heads = __ballot(head);
  const unsigned long long upto = (2ull << l) - 1ull;          // bits 0 .. l (l = 63: all ones)
  const unsigned long long later = heads & ~upto;
  const int len = (later ? (__ffsll((long long)later) - 1) : kWave) - l;     // (head lanes: at most 16, a head starts every column)
  const unsigned long long run = (((1ull << (len & 31)) - 1ull) | (len >= 32 ? ~0ull : 0ull)) << l;   // this lane .. the end of its run
  // ---- equal pixels in the column to the right, one row up / level / one row down (8-connectivity): the pixel level with this one by
  // a wave shuffle, its neighbours above and below from the neighbouring lanes' copies (DPP row shifts: a column is one 16-lane row)
  const uint32_t o_level = (uint32_t)__shfl((int)v, (l + kTY) & (kWave - 1));
  const uint32_t o_below = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o_level, 0x101, 0xF, 0xF, true);   // row_shl:1 -- lane i reads lane i + 1
  const uint32_t o_above = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o_level, 0x111, 0xF, 0xF, true);   // row_shr:1 -- lane i reads lane i - 1
  auto right_equal = [&](int d) -> bool {
    const uint32_t o = d == 0 ? o_level : (d > 0 ? o_below : o_above);
    return valid && cx < kSX - 1 && (unsigned)(ty + d) < (unsigned)kTY && o == v;
  };
  const unsigned long long B0 = __ballot(right_equal(0)), Bp = __ballot(right_equal(1)), Bm = __ballot(right_equal(-1));
  // the same relation seen from the right-hand pixel: its left neighbours
  const unsigned long long G0 = B0 << kTY, Gp = Bm << (kTY - 1), Gm = Bp << (kTY + 1);
  auto head_of = [&](int q) -> int { return 63 - __clzll((long long)(heads & ((2ull << q) - 1ull))); };
  // first row of the neighbouring column, from one above the run to one below it, that holds the same primitive -> that run's head
  int rc = kNone, lp = kNone;
  {
    const unsigned long long rows_r = (B0 & run) | ((Bp & run) << 1) | ((Bm & run) >> 1);
    const unsigned long long rows_l = (G0 & run) | ((Gp & run) << 1) | ((Gm & run) >> 1);
    if (rows_r) rc = head_of(__builtin_ctzll(rows_r) + kTY);
    if (rows_l) lp = head_of(__builtin_ctzll(rows_l) - kTY);
  }
  // ---- a link exists only when both runs chose each other, so chains never fork
  const int lp_of_rc = __shfl(lp, rc != kNone ? rc : l), rc_of_lp = __shfl(rc, lp != kNone ? lp : l);
  const bool hv = head && valid;
  const int child = (hv && rc != kNone && lp_of_rc == l) ? rc : kNone;
  const bool root = hv && !(lp != kNone && rc_of_lp == l);
  // ---- fold the chain into its root: count | sum of columns | sum of rows, strip-relative in one 32-bit word (7 + 8 + 10 bits), and the
  // strip-relative pixel mask (bit = lane)
  const uint32_t ulen = (uint32_t)len;
  uint32_t wm = 0u;
  unsigned long long sm = 0ull;
  if (hv) {
    wm = ulen | (__umul24(ulen, (uint32_t)cx) << 7) | ((__umul24(ulen, (uint32_t)ty) + (__umul24(ulen, ulen - 1u) >> 1)) << 15);
    sm = run;
  }
  const bool has1 = child != kNone;
  const int c2raw = __shfl(child, has1 ? child : l);
  const bool has2 = has1 && c2raw != kNone;
  if (__ballot(has1) != 0ull) {
    // hop 1: every head adds its child's run; hop 2: its grandchild's, which by then holds grandchild + great-grandchild
    const uint32_t owm = (uint32_t)__shfl((int)wm, has1 ? child : l);
    const unsigned long long osm = shfl64(sm, has1 ? child : l);
    if (has1) { wm += owm; sm |= osm; }
    if (__ballot(has2) != 0ull) {
      const uint32_t owm2 = (uint32_t)__shfl((int)wm, has2 ? c2raw : l);
      const unsigned long long osm2 = shfl64(sm, has2 ? c2raw : l);
      if (has2) { wm += owm2; sm |= osm2; }
    }
  }
  if (!root) return;
  {
    const uint32_t n = wm & 127u, scx = (wm >> 7) & 255u, sty = wm >> 15;
    const unsigned long long pk = (unsigned long long)n | ((unsigned long long)(__umul24(n, x0) + scx) << kMomCountBits) |
                                  ((unsigned long long)(__umul24(n, y0) + sty) << (kMomCountBits + kMomSumBits));
    if (!(SMESH_ABL(dbg) & 4)) atomicAdd(&mom[v], pk);
  }
  uint32_t rows = (uint32_t)(sm | (sm >> 32));
  rows = (rows | (rows >> 16)) & 0xFFFFu;
  const int ymin = __builtin_ctz(rows), ymax = 31 - __builtin_clz(rows);
  if (ymax - ymin < 8 && !(SMESH_ABL(dbg) & 2)) {
    unsigned long long mask = 0ull;               // bit dx * 8 + dy (common.hpp, TriFrag); the chain's columns are cx, cx + 1, ...
#pragma unroll
    for (int j = 0; j < kSX; j++)
      if (cx + j < kSX) mask |= (unsigned long long)(((uint32_t)(sm >> (kTY * (cx + j))) & 0xFFFFu) >> ymin & 0xFFu) << (8 * j);
    *reinterpret_cast<uint4*>(&frags[v]) = make_uint4((x0 + (uint32_t)cx) | ((y0 + (uint32_t)ymin) << 16), 1u | (tag << 16),
                                                      (uint32_t)mask, (uint32_t)(mask >> 32));
  }
}

__global__ __launch_bounds__(kWave) void k_rec_moments(const uint32_t* __restrict__ idx, uint32_t W, uint32_t H, uint32_t P, uint32_t strips_y,
                                                       uint32_t nstrips, uint32_t strips_per_xcd, uint32_t tag, int dbg,
                                                       unsigned long long* __restrict__ mom, TriFrag* __restrict__ frags,
                                                       uint32_t* __restrict__ big_count) {
  rec_moments_block(idx, W, H, P, strips_y, nstrips, strips_per_xcd, tag, dbg, mom, frags, big_count);
}

// Up to eight images of the same size in ONE launch per pass (smesh_aggregator_add_many): image blockIdx.y with its own record set.
// Same blocks, same work per image as the one-image kernels above and below; what goes away is seven of every eight launches -- the
// passes that find nothing to do (E and D on a rendering) cost a launch each, ~5 us of a 75 us call.
struct RecImage {
  const uint32_t* idx;
  const float* probs;
  const float* weights;
  unsigned long long* mom;
  TriFrag* frags;
  uint4* big4;
  uint32_t* big_queue;
  uint32_t* big_count;
  uint32_t tag;
};
struct RecGroup {
  RecImage im[8];
};
__global__ __launch_bounds__(kWave) void k_rec_moments_group(RecGroup g, uint32_t W, uint32_t H, uint32_t P, uint32_t strips_y, uint32_t nstrips,
                                                             uint32_t strips_per_xcd, int dbg) {
  const RecImage& r = g.im[blockIdx.y];
  rec_moments_block(r.idx, W, H, P, strips_y, nstrips, strips_per_xcd, r.tag, dbg, r.mom, r.frags, r.big_count);
}

// Bits of the pixels equal to v in an 8-column x 8-row window at (xs, ys), bit dx * 8 + dy.  The window lies inside the image.
__device__ __forceinline__ unsigned long long scan8(const uint32_t* __restrict__ idx, uint32_t H, uint32_t xs, uint32_t ys, uint32_t v) {
  unsigned long long mask = 0ull;
#pragma unroll
  for (int dx = 0; dx < 8; dx++) {
    const uint32_t* col = idx + (uint64_t)(xs + dx) * H + ys;
    const u32x4u a = *reinterpret_cast<const u32x4u*>(col), c = *reinterpret_cast<const u32x4u*>(col + 4);
    const uint32_t bits = (a.x == v ? 1u : 0u) | (a.y == v ? 2u : 0u) | (a.z == v ? 4u : 0u) | (a.w == v ? 8u : 0u) |
                          (c.x == v ? 16u : 0u) | (c.y == v ? 32u : 0u) | (c.z == v ? 64u : 0u) | (c.w == v ? 128u : 0u);
    mask |= (unsigned long long)bits << (dx * 8);
  }
  return mask;
}

// A kind 1 record from the bits of an 8 x 8 window at (xs, ys): origin moved to the first occupied column and row.
__device__ __forceinline__ uint4 record_from_window(unsigned long long mask, uint32_t xs, uint32_t ys) {
  const uint32_t jx = (uint32_t)__builtin_ctzll(mask) >> 3;
  uint32_t rows = (uint32_t)(mask | (mask >> 32));
  rows |= rows >> 16;
  rows |= rows >> 8;
  const uint32_t jy = (uint32_t)__builtin_ctz(rows & 0xFFu);
  mask >>= jx * 8u + jy;                 // every set bit has dy >= jy: nothing crosses a column
  return make_uint4((xs + jx) | ((ys + jy) << 16), 1u, (uint32_t)mask, (uint32_t)(mask >> 32));
}

__device__ __forceinline__ void rec_resolve_block(const uint32_t* __restrict__ idx, uint32_t W, uint32_t H, uint32_t P, uint32_t tag,
                                                  int dbg, unsigned long long* __restrict__ mom, TriFrag* __restrict__ frags,
                                                  uint32_t* __restrict__ big_queue, uint32_t* __restrict__ big_count) {
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= P) return;
  const unsigned long long m = mom[v];
  const uint4 raw = *reinterpret_cast<const uint4*>(&frags[v]);
  if (m == 0ull) {                       // not in this image: no record (whatever the last image left is cleared)
    if (raw.x | raw.y | raw.z | raw.w) *reinterpret_cast<uint4*>(&frags[v]) = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  mom[v] = 0ull;
  const uint32_t n = (uint32_t)(m & ((1ull << kMomCountBits) - 1ull));       // Mesh.h:90-93 for this primitive
  const unsigned long long smask = (unsigned long long)raw.z | ((unsigned long long)raw.w << 32);
  // One group holds every pixel: its record stands as it is.  (`tag` alternates between calls, and every record this pass meets was
  // left by the previous call or written by this call's pass M: a stale record never carries this call's tag.)
  // (Compacting the others -- a sixth of cfg2's primitives, spread over every wave -- through LDS so that one wave of the workgroup
  // repairs them was measured slower: 13.8 -> 15.4 us.)
  if (((raw.y >> 16) == tag && (uint32_t)__popcll(smask) == n) || (SMESH_ABL(dbg) & 1)) return;
  uint4 rec = make_uint4(0u, kPadPending << 16, n, 0u);                      // pending: kind 0, the pixel count parked in the mask
  bool queue = true;
  if (n <= 64u) {
    const uint32_t fx = (uint32_t)((m >> kMomCountBits) & ((1ull << kMomSumBits) - 1ull)) / n;
    const uint32_t fy = (uint32_t)(m >> (kMomCountBits + kMomSumBits)) / n;
    // a primitive of at most 8 x 8 pixels lies within 7 pixels of (the floor of) its centroid; most lie within [-3, +4], and the
    // typical one here -- two or three pixels cut by a strip border -- within [-1, +2]: four loads instead of sixteen
    unsigned long long m4 = 0ull;
    const uint32_t xq = min(fx > 1u ? fx - 1u : 0u, W - 4u), yq = min(fy > 1u ? fy - 1u : 0u, H - 4u);
    if (n <= 16u) {
#pragma unroll
      for (int dx = 0; dx < 4; dx++) {
        const u32x4u a = *reinterpret_cast<const u32x4u*>(idx + (uint64_t)(xq + dx) * H + yq);
        m4 |= (unsigned long long)((a.x == v ? 1u : 0u) | (a.y == v ? 2u : 0u) | (a.z == v ? 4u : 0u) | (a.w == v ? 8u : 0u)) << (dx * 8);
      }
    }
    const bool found4 = n <= 16u && (uint32_t)__popcll(m4) == n;
    const uint32_t xs = min(fx > 3u ? fx - 3u : 0u, W - 8u), ys = min(fy > 3u ? fy - 3u : 0u, H - 8u);
    const unsigned long long m8 = found4 ? 0ull : scan8(idx, H, xs, ys, v);
    if (found4) {
      rec = record_from_window(m4, xq, yq);
      queue = false;
    } else if ((uint32_t)__popcll(m8) == n) {
      rec = record_from_window(m8, xs, ys);
      queue = false;
    } else {
      const uint32_t xw = min(fx > 7u ? fx - 7u : 0u, W - 16u), yw = min(fy > 7u ? fy - 7u : 0u, H - 16u);
      uint32_t found = 0, xlo = 16u, xhi = 0u, ylo = 16u, yhi = 0u;
      for (uint32_t dx = 0; dx < 16u; dx++) {
        const uint32_t* col = idx + (uint64_t)(xw + dx) * H + yw;
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u32x4u a = *reinterpret_cast<const u32x4u*>(col + 4 * k);
          bits |= ((a.x == v ? 1u : 0u) | (a.y == v ? 2u : 0u) | (a.z == v ? 4u : 0u) | (a.w == v ? 8u : 0u)) << (4 * k);
        }
        if (bits) {
          found += (uint32_t)__popc(bits);
          xlo = min(xlo, dx); xhi = dx;
          ylo = min(ylo, (uint32_t)__builtin_ctz(bits));
          yhi = max(yhi, 31u - (uint32_t)__builtin_clz(bits));
        }
      }
      if (found == n) {
        if (xhi - xlo < 8u && yhi - ylo < 8u) {
          // (the window at the box's own origin may reach past the image: clamp it, the record's origin moves back to the pixels)
          const uint32_t bxs = min(xw + xlo, W - 8u), bys = min(yw + ylo, H - 8u);
          rec = record_from_window(scan8(idx, H, bxs, bys, v), bxs, bys);
          queue = false;
        } else {     // a box of at most 16 x 16 pixels: dense by k_rec_big's rule (area <= 16 n + 256)
          rec = make_uint4((xw + xlo) | ((yw + ylo) << 16), 2u, (xw + xhi) | ((yw + yhi) << 16), 0u);
        }
      }
    }
  }
  *reinterpret_cast<uint4*>(&frags[v]) = rec;
  if (queue) big_queue[atomicAdd(big_count, 1u)] = v;                        // capacity P: a primitive is queued at most once
}

__global__ __launch_bounds__(kBlock) void k_rec_resolve(const uint32_t* __restrict__ idx, uint32_t W, uint32_t H, uint32_t P, uint32_t tag,
                                                        int dbg, unsigned long long* __restrict__ mom, TriFrag* __restrict__ frags,
                                                        uint32_t* __restrict__ big_queue, uint32_t* __restrict__ big_count) {
  rec_resolve_block(idx, W, H, P, tag, dbg, mom, frags, big_queue, big_count);
}
__global__ __launch_bounds__(kBlock) void k_rec_resolve_group(RecGroup g, uint32_t W, uint32_t H, uint32_t P, int dbg) {
  const RecImage& r = g.im[blockIdx.y];
  rec_resolve_block(r.idx, W, H, P, r.tag, dbg, r.mom, r.frags, r.big_queue, r.big_count);
}

// Pixels of the sparse primitives (k_rec_big), in pixel order: Mesh.h:94-106 with one float atomic per class.  One thread per pixel;
// an image without sparse primitives -- every rendering -- leaves at the first test.  Mul adds on the hi plane (hi + lo is the value).
// One pixel of pass D: Mesh.h:94-106 for a pixel of a sparse primitive, one float atomic per class.
template <int KIND>
__device__ __forceinline__ void scatter_sparse_pixel(uint64_t i, const uint32_t* __restrict__ idx, const float* __restrict__ probs,
                                                     const float* __restrict__ weights, uint32_t P, uint32_t C, float iew,
                                                     const TriFrag* __restrict__ frags, float* __restrict__ acc, double* __restrict__ acc_d) {
  const uint32_t v = idx[i];
  if (v >= P) return;
  const TriFrag rec = frags[v];
  if (!(rec.kind == 0 && rec.pad == 1)) return;
  const float* __restrict__ pr = probs + i * C;
  float sum = 0.0f, best = 0.0f;
  uint32_t am = 0;
  for (uint32_t c = 0; c < C; c++) {
    const float p = pr[c];
    sum = sum + p;                                                        // Mesh.h:98 (tt::sum, class order)
    if (KIND == SMESH_AGG_SUMMAX && (c == 0 || p > best)) { best = p; am = c; }
  }
  if (!(sum > 0.5f)) return;
  const float w = (iew * (1.0f / (float)(uint32_t)rec.mask) + (1 - iew) * 1.0f) * (weights ? weights[i] : 1.0f);   // Mesh.h:100-102
  float* __restrict__ row = acc + (uint64_t)v * C;
  if (KIND == SMESH_AGG_SUMMAX) {
    atomicAdd(&row[am], best * w);
  } else if (KIND == SMESH_AGG_MUL) {
    // Mul: this image's terms of a sparse primitive are summed in DOUBLE in the aggregator's scratch rows (all zero between calls;
    // MI355X adds float64 in memory natively) and folded into the (hi, lo) row by k_fold_sparse -- float32 atomics on the hi plane
    // missed 1e-5 on get() by three orders of magnitude for primitives of thousands of pixels
    double* __restrict__ drow = acc_d + (uint64_t)v * C;
    for (uint32_t c = 0; c < C; c++) unsafeAtomicAdd(&drow[c], (double)contribution<KIND>(pr[c], w));
  } else {
    for (uint32_t c = 0; c < C; c++) atomicAdd(&row[c], contribution<KIND>(pr[c], w));
  }
}

// Pass D: the pixels of the sparse primitives, a grid-stride loop of a small persistent grid (an image without sparse primitives --
// every rendering -- leaves at the first test, and a few hundred workgroups doing so cost less than one per 256 pixels).
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_scatter_sparse(const uint32_t* __restrict__ idx, const float* __restrict__ probs,
                                                           const float* __restrict__ weights, uint64_t N, uint32_t P, uint32_t C, float iew,
                                                           const TriFrag* __restrict__ frags, const uint32_t* __restrict__ big_count,
                                                           float* __restrict__ acc, double* __restrict__ acc_d) {
  if (big_count[2] == 0u) return;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (uint64_t)gridDim.x * kBlock)
    scatter_sparse_pixel<KIND>(i, idx, probs, weights, P, C, iew, frags, acc, acc_d);
}

template <int KIND>
__global__ __launch_bounds__(kBlock) void k_scatter_sparse_group(RecGroup g, uint64_t N, uint32_t P, uint32_t C, float iew, float* __restrict__ acc,
                                                                 double* __restrict__ acc_d) {
  const RecImage& r = g.im[blockIdx.y];
  if (r.big_count[2] == 0u) return;
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (uint64_t)gridDim.x * kBlock)
    scatter_sparse_pixel<KIND>(i, r.idx, r.probs, r.weights, P, C, iew, r.frags, acc, acc_d);
}

// Mul, behind pass D: every sparse primitive (they are all in the queue) folds the float64 sums of this image's terms into its (hi, lo)
// row, re-centred on the row's largest finite element ("Mul state", fuse_tri.inc.hpp), and leaves the scratch row zero again.
__global__ __launch_bounds__(kBlock) void k_fold_sparse(const TriFrag* __restrict__ frags, const uint32_t* __restrict__ big_queue,
                                                        const uint32_t* __restrict__ big_count, uint32_t C, float* __restrict__ acc,
                                                        float* __restrict__ acc_lo, double* __restrict__ acc_d) {
  if (big_count[2] == 0u) return;
  const uint32_t nbig = big_count[0];
  for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < nbig; q += gridDim.x * kBlock) {
    const uint32_t v = big_queue[q];
    const TriFrag rec = frags[v];
    if (!(rec.kind == 0 && rec.pad == 1)) continue;
    float* hi = acc + (uint64_t)v * C;
    float* lo = acc_lo + (uint64_t)v * C;
    double* drow = acc_d + (uint64_t)v * C;
    float m = -INFINITY;
    for (uint32_t c = 0; c < C; c++) { const float h = hi[c]; if (h > m && h < INFINITY) m = h; }
    const float centre = m > -INFINITY ? m : 0.0f;
    for (uint32_t c = 0; c < C; c++) {
      float h = hi[c], r = lo[c];
      mul_fold(h, r, centre, __hip_atomic_load(&drow[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // (written by atomics: read where they landed)
      hi[c] = h; lo[c] = r; drow[c] = 0.0;
    }
  }
}

// Pending primitives (more than 64 pixels, or pixels far from the centroid): passes E and C' in ONE launch of a small persistent grid;
// every workgroup leaves at once when nothing is pending -- every rendering of a finely tessellated mesh.  E: extent by atomics, one
// set per run of a column.  C': the kind 2 record, or "sparse" as k_rec_big decides -- by the workgroup that finishes pass E LAST
// (a ticket counter, big_count[3]: it has seen every other workgroup's fence, so their atomics have landed).  No workgroup ever
// waits for another one: nothing here depends on how many of them are resident at once (round 3's grid barrier did).
__device__ __forceinline__ void rec_extent_block(const uint32_t* __restrict__ idx, uint64_t N, uint32_t H, uint32_t P,
                                                 TriFrag* __restrict__ frags, uint4* __restrict__ big4,
                                                 const uint32_t* __restrict__ big_queue, uint32_t* __restrict__ big_count) {
  const uint32_t nbig = big_count[0];       // final: pass R ran before this launch
  if (nbig == 0u) return;
  for (uint64_t base = (uint64_t)blockIdx.x * kBlock; base < N; base += (uint64_t)gridDim.x * kBlock) {
    const uint64_t i = base + threadIdx.x;
    const int l = threadIdx.x & 63;
    uint32_t v = i < N ? idx[i] : 0xFFFFFFFFu;
    if (v >= P) v = 0xFFFFFFFFu;
    const uint32_t x = (uint32_t)(i / H), y = (uint32_t)(i - (uint64_t)x * H);
    const uint32_t prev = (uint32_t)__shfl_up((int)v, 1);
    const bool leader = l == 0 || prev != v || y == 0u;
    const unsigned long long Lm = __ballot(leader);
    const unsigned long long after = l == 63 ? 0ull : (Lm >> (l + 1));
    const uint32_t next = after ? (uint32_t)l + 1u + (uint32_t)__builtin_ctzll(after) : 64u;
    if (!leader || v == 0xFFFFFFFFu) continue;
    const uint32_t len = next - (uint32_t)l;
    if (frags[v].pad != kPadPending) continue;
    uint32_t* b = reinterpret_cast<uint32_t*>(&big4[v]);
    atomicMax(&b[0], x + 1u);            // largest x + 1
    atomicMax(&b[1], y + len);           // largest y + 1
    atomicMax(&b[2], 65536u - x);        // 65536 - smallest x
    atomicMax(&b[3], 65536u - y);        // 65536 - smallest y
  }
  __shared__ uint32_t s_ticket;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_ticket = atomicAdd(&big_count[3], 1u);   // (reset by the next call's pass M)
  }
  __syncthreads();
  if (s_ticket != gridDim.x - 1u) return;
  __threadfence();
  for (uint32_t q = threadIdx.x; q < nbig; q += kBlock) {
    const uint32_t v = big_queue[q];
    TriFrag rec = frags[v];
    if (rec.pad != kPadPending) continue;          // a kind 2 record k_rec_resolve finished itself
    uint32_t* bw = reinterpret_cast<uint32_t*>(&big4[v]);
    uint32_t b[4];
    for (int k = 0; k < 4; k++) b[k] = __hip_atomic_load(&bw[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (written by atomics: read where they landed)
    const uint32_t x0 = 65536u - b[2], x1 = b[0] - 1u, ytop = 65536u - b[3], y1 = b[1] - 1u;
    const unsigned long long n = rec.mask;
    const unsigned long long area = (unsigned long long)(x1 - x0 + 1u) * (unsigned long long)(y1 - ytop + 1u);
    rec.x0 = (uint16_t)x0; rec.y0 = (uint16_t)ytop; rec.pad = 0;
    if (x1 - x0 < 8u && y1 - ytop < 8u) {          // (cannot happen for a pending primitive; kept so that the record is right if it does)
      unsigned long long m = 0ull;
      for (uint32_t dx = 0; dx <= x1 - x0; dx++)
        for (uint32_t dy = 0; dy <= y1 - ytop; dy++)
          if (idx[(uint64_t)(x0 + dx) * H + ytop + dy] == v) m |= 1ull << (dx * 8u + dy);
      rec.kind = 1;
      rec.mask = m;
    } else if (area <= kDenseFactor * n + kDenseSlack) {
      rec.kind = 2;
      rec.mask = (unsigned long long)x1 | ((unsigned long long)y1 << 16);
    } else {            // sparse, as in k_rec_big
      rec.kind = 0; rec.pad = 1;
      rec.mask = n;
      big_count[2] = 1u;
    }
    frags[v] = rec;
    big4[v] = make_uint4(0u, 0u, 0u, 0u);
  }
}

__global__ __launch_bounds__(kBlock) void k_rec_extent(const uint32_t* __restrict__ idx, uint64_t N, uint32_t H, uint32_t P,
                                                       TriFrag* __restrict__ frags, uint4* __restrict__ big4,
                                                       const uint32_t* __restrict__ big_queue, uint32_t* __restrict__ big_count) {
  rec_extent_block(idx, N, H, P, frags, big4, big_queue, big_count);
}
__global__ __launch_bounds__(kBlock) void k_rec_extent_group(RecGroup g, uint64_t N, uint32_t H, uint32_t P) {
  const RecImage& r = g.im[blockIdx.y];
  rec_extent_block(r.idx, N, H, P, r.frags, r.big4, r.big_queue, r.big_count);     // (the ticket counts the blocks of THIS image: gridDim.x)
}

// After the fusion, when the image is much smaller than the primitive count: only the records the image touched.
__global__ __launch_bounds__(kBlock) void k_rec_clear(const uint32_t* __restrict__ idx, uint64_t N, uint32_t P, uint32_t* __restrict__ cand,
                                                      TriFrag* __restrict__ frags, uint32_t* __restrict__ big_count) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i == 0) { big_count[0] = 0u; big_count[2] = 0u; }
  if (i >= N) return;
  const uint32_t v = idx[i];
  if (v >= P) return;
  TriFrag z;
  z.x0 = 0; z.y0 = 0; z.kind = 0; z.pad = 0; z.mask = 0ull;
  frags[v] = z;
  cand[v] = 0u;
}

}  // namespace

namespace smesh {

// One allocation: [frags 16 P][cand 4 P][big_count 16] -- what a call dirties and one memset clears -- then [big4 16 P][queue 4 P]
// (rounded up to 16 bytes: big4 behind it is read and written with 128-bit accesses)
static size_t block_bytes(uint64_t P) { return (((size_t)P * (sizeof(TriFrag) + 4) + 16) + 15) & ~(size_t)15; }

void ImageRecords::release() {
  if (frags) (void)dev_free(frags);
  frags = nullptr; cand = nullptr; big4 = nullptr; big_queue = nullptr; big_count = nullptr; mom = nullptr;
  P = 0; clean = false;
}

// Moments (passes M, R, E, C') when the packed sums are exact: fewer than 2^24 pixels, sides of 16 .. 16383 pixels -- and when pass R's
// sweep over all P primitives (24 bytes each) is not what the call costs: up to eight primitives per pixel (cfg5: 2.3); beyond,
// passes A / B and their per-pixel clear stay O(pixels).
static bool use_moments(uint64_t W, uint64_t H, uint64_t P) {
  static const bool off = getenv("SMESH_REC_MOMENTS") && atoi(getenv("SMESH_REC_MOMENTS")) == 0;
  return !off && W >= 16 && H >= 16 && W < 16384 && H < 16384 && W * H < (1ull << kMomCountBits) && P <= 8 * W * H;
}

int image_records_build(DeviceCtx* ctx, ImageRecords& r, const uint32_t* d_idx, uint64_t W, uint64_t H, uint64_t P) {
  if (W > 65535 || H > 65535) return fail(SMESH_ERR_INVALID, "image records: image sides are limited to 65535 pixels");
  hipStream_t st = ctx->stream;
  if (r.P != P || !r.frags) {
    r.release();
    const size_t n = (size_t)(P ? P : 1);
    char* base = nullptr;
    const size_t total = block_bytes(n) + n * (sizeof(uint4) + 8 + 4);
    hipError_t e = dev_malloc(reinterpret_cast<void**>(&base), total);
    if (e != hipSuccess) return fail_hip(e, "image records allocation", __FILE__, __LINE__);
    r.frags = reinterpret_cast<TriFrag*>(base);
    r.cand = reinterpret_cast<uint32_t*>(base + n * sizeof(TriFrag));
    r.big_count = reinterpret_cast<uint32_t*>(base + n * (sizeof(TriFrag) + 4));
    r.big4 = reinterpret_cast<uint4*>(base + block_bytes(n));
    r.mom = reinterpret_cast<unsigned long long*>(base + block_bytes(n) + n * sizeof(uint4));
    r.big_queue = reinterpret_cast<uint32_t*>(base + block_bytes(n) + n * (sizeof(uint4) + 8));
    r.P = P;
    SMESH_HIP(hipMemsetAsync(base, 0, total, st));    // once: every pass leaves big4 / mom zero behind it
    r.clean = true;
  }
  const uint64_t N = W * H;
  r.moments = use_moments(W, H, P);
  if (r.moments) {
    // frags may hold the last image's records: pass R rewrites whatever differs
    r.clean = false;
    static const int dbg = SMESH_ABL_ENV("SMESH_REC_DBG");   // development ablation (timing only: wrong results; -DSMESH_ABLATION builds)
    const uint32_t strips_y = (uint32_t)div_up(H, kTY), nstrips = (uint32_t)div_up(W, kSX) * strips_y;
    const uint32_t strips_per_xcd = (uint32_t)div_up(nstrips, 8);
    hipLaunchKernelGGL(k_rec_moments, dim3(strips_per_xcd * 8), dim3(kWave), 0, st, d_idx, (uint32_t)W, (uint32_t)H, (uint32_t)P, strips_y, nstrips,
                       strips_per_xcd, r.tag, dbg, r.mom, r.frags, r.big_count);
    hipLaunchKernelGGL(k_rec_resolve, dim3((uint32_t)div_up(P ? P : 1, kBlock)), dim3(kBlock), 0, st, d_idx, (uint32_t)W, (uint32_t)H, (uint32_t)P,
                       r.tag, dbg, r.mom, r.frags, r.big_queue, r.big_count);
    r.tag ^= 6u;      // 2 <-> 4
    SMESH_HIP(hipGetLastError());
    return SMESH_OK;
  }
  if (!r.clean) SMESH_HIP(hipMemsetAsync(r.frags, 0, block_bytes(P ? P : 1), st));
  r.clean = false;   // until image_records_clear has run
  const dim3 grid((uint32_t)div_up(N, kBlock)), block(kBlock);
  hipLaunchKernelGGL(k_rec_origin, grid, block, 0, st, d_idx, N, (uint32_t)H, (uint32_t)P, r.cand);
  hipLaunchKernelGGL(k_rec_mask, grid, block, 0, st, d_idx, N, (uint32_t)H, (uint32_t)P, r.cand, r.big4, r.frags, r.big_queue, r.big_count);
  hipLaunchKernelGGL(k_rec_big, dim3(64), block, 0, st, d_idx, (uint32_t)H, r.cand, r.big4, r.frags, r.big_queue, r.big_count);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

// smesh_aggregator_add_many: the records of `n` <= 8 images of the same size, each into its own set, with ONE launch per pass
// (M, R, E).  Returns false (nothing launched) when the images do not take the moments passes: the caller then builds them one by one.
bool image_records_build_group(DeviceCtx* ctx, ImageRecords* const* recs, const uint32_t* const* d_idx, int n, uint64_t W, uint64_t H, uint64_t P,
                               int* status) {
  *status = SMESH_OK;
  if (n < 1 || n > 8 || W > 65535 || H > 65535 || !use_moments(W, H, P)) return false;
  hipStream_t st = ctx->stream;
  RecGroup g;
  memset(&g, 0, sizeof g);
  for (int i = 0; i < n; i++) {
    ImageRecords& r = *recs[i];
    if (r.P != P || !r.frags) {
      r.release();
      const size_t np = (size_t)(P ? P : 1);
      char* base = nullptr;
      const size_t total = block_bytes(np) + np * (sizeof(uint4) + 8 + 4);
      hipError_t e = dev_malloc(reinterpret_cast<void**>(&base), total);
      if (e != hipSuccess) { *status = fail_hip(e, "image records allocation", __FILE__, __LINE__); return true; }
      r.frags = reinterpret_cast<TriFrag*>(base);
      r.cand = reinterpret_cast<uint32_t*>(base + np * sizeof(TriFrag));
      r.big_count = reinterpret_cast<uint32_t*>(base + np * (sizeof(TriFrag) + 4));
      r.big4 = reinterpret_cast<uint4*>(base + block_bytes(np));
      r.mom = reinterpret_cast<unsigned long long*>(base + block_bytes(np) + np * sizeof(uint4));
      r.big_queue = reinterpret_cast<uint32_t*>(base + block_bytes(np) + np * (sizeof(uint4) + 8));
      r.P = P;
      if (hipMemsetAsync(base, 0, total, st) != hipSuccess) { *status = fail(SMESH_ERR_RUNTIME, "image records: clearing a new record set failed"); return true; }
      r.clean = true;
    } else if (!r.moments && !r.clean) {
      // the set was last used by passes A / B and not cleared: start from zero
      if (hipMemsetAsync(r.frags, 0, block_bytes(P ? P : 1), st) != hipSuccess) { *status = fail(SMESH_ERR_RUNTIME, "image records: clear failed"); return true; }
    }
  }
  // every record set exists and is cleared: only now do the sets change state (a failed allocation for image i used to
  // leave images 0 .. i-1 with flipped tags and nothing launched, so that the next build met stale records carrying its own tag)
  for (int i = 0; i < n; i++) {
    ImageRecords& r = *recs[i];
    r.moments = true;
    r.clean = false;
    g.im[i] = RecImage{d_idx[i], nullptr, nullptr, r.mom, r.frags, r.big4, r.big_queue, r.big_count, r.tag};
    r.tag ^= 6u;      // 2 <-> 4
  }
  static const int dbg = SMESH_ABL_ENV("SMESH_REC_DBG");
  const uint64_t N = W * H;
  const uint32_t strips_y = (uint32_t)div_up(H, kTY), nstrips = (uint32_t)div_up(W, kSX) * strips_y;
  const uint32_t strips_per_xcd = (uint32_t)div_up(nstrips, 8);
  hipLaunchKernelGGL(k_rec_moments_group, dim3(strips_per_xcd * 8, (uint32_t)n), dim3(kWave), 0, st, g, (uint32_t)W, (uint32_t)H, (uint32_t)P, strips_y,
                     nstrips, strips_per_xcd, dbg);
  hipLaunchKernelGGL(k_rec_resolve_group, dim3((uint32_t)div_up(P ? P : 1, kBlock), (uint32_t)n), dim3(kBlock), 0, st, g, (uint32_t)W, (uint32_t)H,
                     (uint32_t)P, dbg);
  const uint32_t eg = (uint32_t)std::min<uint64_t>(div_up(N, kBlock), (uint64_t)std::max(1, ctx->num_cus / n));
  hipLaunchKernelGGL(k_rec_extent_group, dim3(eg, (uint32_t)n), dim3(kBlock), 0, st, g, N, (uint32_t)H, (uint32_t)P);
  if (hipGetLastError() != hipSuccess) *status = fail(SMESH_ERR_RUNTIME, "image records: group launch failed");
  return true;
}

// Pass D for the images of a group (behind their fusion launch), one launch; Mul: the fold of each image behind it, image by image.
int image_records_scatter_sparse_group(DeviceCtx* ctx, ImageRecords* const* recs, int n, int kind, const uint32_t* const* d_idx,
                                       const float* const* d_probs, const float* const* d_w, uint64_t W, uint64_t H, uint32_t C, float iew,
                                       float* acc, float* acc_lo, double* acc_d, hipStream_t st) {
  if (kind == SMESH_AGG_MUL) {     // (the float64 scratch rows serve one image at a time)
    for (int i = 0; i < n; i++)
      SMESH_TRY(image_records_scatter_sparse(ctx, *recs[i], kind, d_idx[i], d_probs[i], d_w ? d_w[i] : nullptr, W, H, C, iew, acc, acc_lo, acc_d, st));
    return SMESH_OK;
  }
  RecGroup g;
  memset(&g, 0, sizeof g);
  for (int i = 0; i < n; i++) {
    const ImageRecords& r = *recs[i];
    g.im[i] = RecImage{d_idx[i], d_probs[i], d_w ? d_w[i] : nullptr, r.mom, r.frags, r.big4, r.big_queue, r.big_count, 0u};
  }
  const uint64_t N = W * H;
  const dim3 grid((uint32_t)std::min<uint64_t>(div_up(N, kBlock), (uint64_t)std::max(1, 2 * ctx->num_cus / n)), (uint32_t)n), block(kBlock);
  if (kind == SMESH_AGG_SUM) hipLaunchKernelGGL(k_scatter_sparse_group<SMESH_AGG_SUM>, grid, block, 0, st, g, N, (uint32_t)recs[0]->P, C, iew, acc, acc_d);
  else hipLaunchKernelGGL(k_scatter_sparse_group<SMESH_AGG_SUMMAX>, grid, block, 0, st, g, N, (uint32_t)recs[0]->P, C, iew, acc, acc_d);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

// Passes E and C' of the moments variant (between the build and the fusion launch); nothing for passes A / B.
int image_records_pending(DeviceCtx* ctx, ImageRecords& r, const uint32_t* d_idx, uint64_t W, uint64_t H, hipStream_t st) {
  if (!r.moments) return SMESH_OK;
  const uint64_t N = W * H;
  const dim3 grid((uint32_t)std::min<uint64_t>(div_up(N, kBlock), (uint32_t)std::max(1, ctx->num_cus))), block(kBlock);
  hipLaunchKernelGGL(k_rec_extent, grid, block, 0, st, d_idx, N, (uint32_t)H, (uint32_t)r.P, r.frags, r.big4, r.big_queue, r.big_count);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

// Pass D, behind the fusion launch (whose waves write whole blocks of rows back: the float atomics must not run beside it): the pixels
// of the sparse primitives.  `acc_lo` / `acc_d`: the Mul aggregator's lo plane and its float64 scratch rows (else null).
int image_records_scatter_sparse(DeviceCtx* ctx, ImageRecords& r, int kind, const uint32_t* d_idx, const float* d_probs, const float* d_w,
                                 uint64_t W, uint64_t H, uint32_t C, float iew, float* acc, float* acc_lo, double* acc_d, hipStream_t st) {
  const uint64_t N = W * H;
  const dim3 grid((uint32_t)std::min<uint64_t>(div_up(N, kBlock), 2u * (uint32_t)std::max(1, ctx->num_cus))), block(kBlock);
  switch (kind) {
    case SMESH_AGG_SUM:
      hipLaunchKernelGGL(k_scatter_sparse<SMESH_AGG_SUM>, grid, block, 0, st, d_idx, d_probs, d_w, N, (uint32_t)r.P, C, iew, r.frags,
                         r.big_count, acc, acc_d);
      break;
    case SMESH_AGG_SUMMAX:
      hipLaunchKernelGGL(k_scatter_sparse<SMESH_AGG_SUMMAX>, grid, block, 0, st, d_idx, d_probs, d_w, N, (uint32_t)r.P, C, iew, r.frags,
                         r.big_count, acc, acc_d);
      break;
    default:
      if (!acc_lo || !acc_d) return fail(SMESH_ERR_RUNTIME, "image records: the Mul aggregator's float64 scratch rows are missing");
      hipLaunchKernelGGL(k_scatter_sparse<SMESH_AGG_MUL>, grid, block, 0, st, d_idx, d_probs, d_w, N, (uint32_t)r.P, C, iew, r.frags,
                         r.big_count, acc, acc_d);
      hipLaunchKernelGGL(k_fold_sparse, dim3((uint32_t)std::max(1, ctx->num_cus)), block, 0, st, r.frags, r.big_queue, r.big_count, C, acc, acc_lo, acc_d);
      break;
  }
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

int image_records_clear(DeviceCtx* ctx, ImageRecords& r, const uint32_t* d_idx, uint64_t W, uint64_t H, hipStream_t st) {
  const uint64_t N = W * H;
  if (r.moments) return SMESH_OK;     // nothing to clear: the next build's pass R rewrites the records
  if (r.P <= N) {
    SMESH_HIP(hipMemsetAsync(r.frags, 0, block_bytes(r.P ? r.P : 1), st));
  } else {
    hipLaunchKernelGGL(k_rec_clear, dim3((uint32_t)div_up(N, kBlock)), dim3(kBlock), 0, st, d_idx, N, (uint32_t)r.P, r.cand, r.frags,
                       r.big_count);
    SMESH_HIP(hipGetLastError());
  }
  r.clean = true;
  return SMESH_OK;
}

}  // namespace smesh
