// image_records.hip -- per-primitive records built from an ARBITRARY index image, so that MeshAggregator::add() on an image the
// library did not render (the reference's harness reloads its renders from an .npz cache, eval-scannet/eval_scannet.py:168-185;
// other renderers; images a caller edited) can take the triangle-order fusion kernels (fusion.hip, k_fuse_tri and friends) instead
// of the atomic scatter-add: every accumulator row then has one owner, the additions happen in the reference's pixel order
// (Mesh.h:94-106 run single-threaded) and the result no longer depends on the order in which float atomics land.
//
// What the rasteriser leaves per triangle (common.hpp, TriFrag) is rebuilt per PRIMITIVE from the image alone:
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

#include <cmath>
#include <type_traits>

using namespace smesh;

namespace {

#include "fuse_tri.inc.hpp"   // contribution<KIND>() (the Mul logarithm is part of the spec)

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

// Pixels of the sparse primitives (k_rec_big), in pixel order: Mesh.h:94-106 with one float atomic per class.  One thread per pixel;
// an image without sparse primitives -- every rendering -- leaves at the first test.  Mul adds on the hi plane (hi + lo is the value).
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_scatter_sparse(const uint32_t* __restrict__ idx, const float* __restrict__ probs,
                                                           const float* __restrict__ weights, uint64_t N, uint32_t P, uint32_t C, float iew,
                                                           const TriFrag* __restrict__ frags, const uint32_t* __restrict__ big_count,
                                                           float* __restrict__ acc) {
  if (big_count[2] == 0u) return;
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
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
  } else {
    for (uint32_t c = 0; c < C; c++) atomicAdd(&row[c], contribution<KIND>(pr[c], w));
  }
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
  if (frags) (void)hipFree(frags);
  frags = nullptr; cand = nullptr; big4 = nullptr; big_queue = nullptr; big_count = nullptr;
  P = 0; clean = false;
}

int image_records_build(DeviceCtx* ctx, ImageRecords& r, const uint32_t* d_idx, uint64_t W, uint64_t H, uint64_t P) {
  if (W > 65535 || H > 65535) return fail(SMESH_ERR_INVALID, "image records: image sides are limited to 65535 pixels");
  hipStream_t st = ctx->stream;
  if (r.P != P || !r.frags) {
    r.release();
    const size_t n = (size_t)(P ? P : 1);
    char* base = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), block_bytes(n) + n * (sizeof(uint4) + 4));
    if (e != hipSuccess) return fail_hip(e, "image records allocation", __FILE__, __LINE__);
    r.frags = reinterpret_cast<TriFrag*>(base);
    r.cand = reinterpret_cast<uint32_t*>(base + n * sizeof(TriFrag));
    r.big_count = reinterpret_cast<uint32_t*>(base + n * (sizeof(TriFrag) + 4));
    r.big4 = reinterpret_cast<uint4*>(base + block_bytes(n));
    r.big_queue = reinterpret_cast<uint32_t*>(base + block_bytes(n) + n * sizeof(uint4));
    r.P = P;
    r.clean = false;
  }
  if (!r.clean) SMESH_HIP(hipMemsetAsync(r.frags, 0, block_bytes(P ? P : 1) + (size_t)(P ? P : 1) * sizeof(uint4), st));
  r.clean = false;   // until image_records_clear has run
  const uint64_t N = W * H;
  const dim3 grid((uint32_t)div_up(N, kBlock)), block(kBlock);
  hipLaunchKernelGGL(k_rec_origin, grid, block, 0, st, d_idx, N, (uint32_t)H, (uint32_t)P, r.cand);
  hipLaunchKernelGGL(k_rec_mask, grid, block, 0, st, d_idx, N, (uint32_t)H, (uint32_t)P, r.cand, r.big4, r.frags, r.big_queue, r.big_count);
  hipLaunchKernelGGL(k_rec_big, dim3(64), block, 0, st, d_idx, (uint32_t)H, r.cand, r.big4, r.frags, r.big_queue, r.big_count);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

int image_records_scatter_sparse(DeviceCtx* ctx, ImageRecords& r, int kind, const uint32_t* d_idx, const float* d_probs, const float* d_w,
                                 uint64_t W, uint64_t H, uint32_t C, float iew, float* acc) {
  const uint64_t N = W * H;
  const dim3 grid((uint32_t)div_up(N, kBlock)), block(kBlock);
  switch (kind) {
    case SMESH_AGG_SUM:
      hipLaunchKernelGGL(k_scatter_sparse<SMESH_AGG_SUM>, grid, block, 0, ctx->stream, d_idx, d_probs, d_w, N, (uint32_t)r.P, C, iew, r.frags,
                         r.big_count, acc);
      break;
    case SMESH_AGG_SUMMAX:
      hipLaunchKernelGGL(k_scatter_sparse<SMESH_AGG_SUMMAX>, grid, block, 0, ctx->stream, d_idx, d_probs, d_w, N, (uint32_t)r.P, C, iew, r.frags,
                         r.big_count, acc);
      break;
    default:
      hipLaunchKernelGGL(k_scatter_sparse<SMESH_AGG_MUL>, grid, block, 0, ctx->stream, d_idx, d_probs, d_w, N, (uint32_t)r.P, C, iew, r.frags,
                         r.big_count, acc);
      break;
  }
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

int image_records_clear(DeviceCtx* ctx, ImageRecords& r, const uint32_t* d_idx, uint64_t W, uint64_t H) {
  const uint64_t N = W * H;
  if (r.P <= N) {
    SMESH_HIP(hipMemsetAsync(r.frags, 0, block_bytes(r.P ? r.P : 1), ctx->stream));
  } else {
    hipLaunchKernelGGL(k_rec_clear, dim3((uint32_t)div_up(N, kBlock)), dim3(kBlock), 0, ctx->stream, d_idx, N, (uint32_t)r.P, r.cand, r.frags,
                       r.big_count);
    SMESH_HIP(hipGetLastError());
  }
  r.clean = true;
  return SMESH_OK;
}

}  // namespace smesh
